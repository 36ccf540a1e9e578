// Device-side small-matrix camera maths: Rodrigues, Brown distortion (forward / inverse), planar-homography start and
// Levenberg-Marquardt pose refinement. Everything stays in registers / private memory of one lane (6 parameters,
// 2n residuals accumulated on the fly into J^T J and J^T e) — there is no dense contraction worth an MFMA here.
//
// Reference call sites: cv::solvePnP(obj, img, K, dist, rvec, tvec) default ITERATIVE
//   /root/reference/src/markerdetector.cpp:458, src/marker.cpp:118, src/boarddetector.cpp:157,193
// cv::undistortPoints / cv::projectPoints: src/markerdetector.cpp:959,152; rotateXAxis: src/utils.cpp:16-30.
#pragma once
#include <float.h>

#include "internal.h"

namespace ah {

// Gaussian elimination with partial pivoting, same operation order as the CPU oracle. N is a template parameter and every
// loop is unrolled, the pivot row is exchanged by selects, so A and b are indexed statically and stay in registers (a
// run-time indexed private array would live in scratch memory: one memory round trip per element access).
template <int N>
__device__ inline bool solve_static(double* A, double* b) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < N; c++) {
        int piv = c;
        double best = fabs(A[c * N + c]);
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const double v = fabs(A[r * N + c]);
            if (v > best) best = v, piv = r;
        }
        if (best == 0) ok = false;
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const bool sw = piv == r;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const double t = A[c * N + k], u = A[r * N + k];
                A[c * N + k] = sw ? u : t;
                A[r * N + k] = sw ? t : u;
            }
            const double t = b[c], u = b[r];
            b[c] = sw ? u : t;
            b[r] = sw ? t : u;
        }
        const double inv = 1.0 / A[c * N + c];
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const double f = A[r * N + c] * inv;
#pragma unroll
            for (int k = c; k < N; k++) A[r * N + k] -= f * A[c * N + k];
            b[r] -= f * b[c];
        }
    }
#pragma unroll
    for (int r = N - 1; r >= 0; r--) {
        double s = b[r];
#pragma unroll
        for (int k = r + 1; k < N; k++) s -= A[r * N + k] * b[k];
        b[r] = s / A[r * N + r];
    }
    return ok;
}

// Symmetric positive definite N x N system by LDL^T (no pivoting, no selects): the normal equations of the homography and the
// damped J^T J of Levenberg-Marquardt are SPD. cv::solve / CvLevMarq use pivoted eliminations (SVD / LU); both are backward
// stable on these systems and the results agree far inside the 1e-4 pose tolerance — the batched kernels trade the 1500
// v_cndmask of the pivoted form for a chain a third as long. Returns false when a pivot is not positive (degenerate input).
template <int N>
__device__ inline bool solve_spd(double* A, double* b) {
    double d[N];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double dj = A[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= A[j * N + k] * A[j * N + k] * d[k];   // A[j][k], k < j holds L[j][k]
        if (!(dj > 0)) ok = false;
        d[j] = dj;
        const double inv = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double v = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= A[i * N + k] * A[j * N + k] * d[k];
            A[i * N + j] = v * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {   // L y = b
#pragma unroll
        for (int k = 0; k < i; k++) b[i] -= A[i * N + k] * b[k];
    }
#pragma unroll
    for (int i = 0; i < N; i++) b[i] /= d[i];
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {   // L^T x = z
#pragma unroll
        for (int k = i + 1; k < N; k++) b[i] -= A[k * N + i] * b[k];
    }
    return ok;
}

__device__ inline bool solve_n(double* A, double* b, int n) {
    if (n == 6) return solve_static<6>(A, b);
    return solve_static<8>(A, b);
}

__device__ inline void mat3_mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = t[i];
}

__device__ inline double mat3_det(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// projection onto the rotation group (polar factor), Newton iteration
__device__ inline void orthonormalise(double* R) {
    for (int it = 0; it < 30; it++) {
        double d = mat3_det(R);
        if (d == 0) return;
        double c[9];
        c[0] = (R[4] * R[8] - R[5] * R[7]) / d;
        c[1] = (R[5] * R[6] - R[3] * R[8]) / d;
        c[2] = (R[3] * R[7] - R[4] * R[6]) / d;
        c[3] = (R[2] * R[7] - R[1] * R[8]) / d;
        c[4] = (R[0] * R[8] - R[2] * R[6]) / d;
        c[5] = (R[1] * R[6] - R[0] * R[7]) / d;
        c[6] = (R[1] * R[5] - R[2] * R[4]) / d;
        c[7] = (R[2] * R[3] - R[0] * R[5]) / d;
        c[8] = (R[0] * R[4] - R[1] * R[3]) / d;
        double diff = 0;
        for (int k = 0; k < 9; k++) {
            double n = 0.5 * (R[k] + c[k]);
            diff = fmax(diff, fabs(n - R[k]));
            R[k] = n;
        }
        if (diff < 1e-16) break;
    }
}

// Rodrigues vector -> matrix, optional dR/dr (J[j*9+k] = dR[k]/dr[j])
__device__ inline void rodrigues_vec2mat(const double* r, double* R, double* J) {
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double dRX[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = I[k];
        if (J)
            for (int k = 0; k < 27; k++) J[k] = dRX[k];
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    rx *= itheta, ry *= itheta, rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx_[k] + a4 * dRX[i * 9 + k];
        }
    }
}

__device__ inline void rodrigues_mat2vec(const double* Rin, double* r) {
    double R[9];
    for (int k = 0; k < 9; k++) R[k] = Rin[k];
    orthonormalise(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta, ry *= theta, rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth, ry *= vth, rz *= vth;
    }
    r[0] = rx, r[1] = ry, r[2] = rz;
}

// inverse Brown model: pixel -> normalised coordinates, 5 fixed-point iterations (cvUndistortPoints)
__device__ inline void undistort_point(double px, double py, const float* K, const double* k, double* ox, double* oy) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double ifx = 1. / fx, ify = 1. / fy;
    double x = (px - cx) * ifx, y = (py - cy) * ify;
    double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = x, *oy = y;
}

// forward model for one point; optional derivative rows (2x3 each) w.r.t. rvec and tvec
__device__ inline void project_point(double X, double Y, double Z, const double* R, const double* dRdr, const double* t,
                                     const float* K, const double* k, double* mx, double* my, double* dr, double* dt) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z, y *= z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
    double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
    *mx = xd * fx + cx;
    *my = yd * fy + cy;
    if (!dr) return;
    for (int j = 0; j < 6; j++) {
        double dxd, dyd;
        if (j < 3) {
            const double* d = dRdr + j * 9;
            double dx0 = X * d[0] + Y * d[1] + Z * d[2];
            double dy0 = X * d[3] + Y * d[4] + Z * d[5];
            double dz0 = X * d[6] + Y * d[7] + Z * d[8];
            dxd = z * (dx0 - x * dz0), dyd = z * (dy0 - y * dz0);
        } else {
            int q = j - 3;
            dxd = q == 0 ? z : (q == 1 ? 0 : -x * z);
            dyd = q == 0 ? 0 : (q == 1 ? z : -y * z);
        }
        double dr2 = 2 * x * dxd + 2 * y * dyd;
        double dcdist = k[0] * dr2 + 2 * k[1] * r2 * dr2 + 3 * k[4] * r4 * dr2;
        double dicdist2 = -icdist2 * icdist2 * (k[5] * dr2 + 2 * k[6] * r2 * dr2 + 3 * k[7] * r4 * dr2);
        double da1 = 2 * (x * dyd + y * dxd);
        double omx = fx * (dxd * cdist * icdist2 + x * dcdist * icdist2 + x * cdist * dicdist2 + k[2] * da1 + k[3] * (dr2 + 2 * x * dxd));
        double omy = fy * (dyd * cdist * icdist2 + y * dcdist * icdist2 + y * cdist * dicdist2 + k[2] * (dr2 + 2 * y * dyd) + k[3] * da1);
        if (j < 3)
            dr[j] = omx, dr[3 + j] = omy;
        else
            dt[j - 3] = omx, dt[j] = omy;
    }
}

// CvLevMarq's damping factor 10^k, k = -16 .. 16 (the reference evaluates exp(k * log(10.)) every iteration; the two agree to
// an ulp and lambda only scales the diagonal by 1 + lambda)
__device__ inline double lm_lambda(int k) {
    const double t[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.,
                          1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
    double v = t[16];
#pragma unroll
    for (int i = 0; i < 33; i++) v = (k == i - 16) ? t[i] : v;   // selects, not an indexed private array
    return v;
}

// Point accessor: obj xyz (float, stride 3) and image xy (float, stride 2) in global or private memory.
// solvePnP(ITERATIVE) for planar (z == 0) object points: homography start + <= 20 LM iterations, eps FLT_EPSILON.
__device__ inline bool solve_pnp_planar(const float* obj, const float* img, int n, const CamModel& cam, double* rvec, double* tvec) {
    if (n < 4) return false;
    const float* K = cam.K;
    const double* k = cam.k;
    double Mc[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        if (obj[3 * i + 2] != 0.f) return false;
        Mc[0] += (double)obj[3 * i], Mc[1] += (double)obj[3 * i + 1];
    }
    Mc[0] /= n, Mc[1] /= n;
    // ---- homography plane -> normalised image (inputs rounded to float as cv::findHomography does)
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        double ux, uy;
        undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
        cM[0] += (double)(float)((double)obj[3 * i] - Mc[0]), cM[1] += (double)(float)((double)obj[3 * i + 1] - Mc[1]);
        cm[0] += (double)(float)ux, cm[1] += (double)(float)uy;
    }
    cM[0] /= n, cM[1] /= n, cm[0] /= n, cm[1] /= n;
    for (int i = 0; i < n; i++) {
        double ux, uy;
        undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
        sM[0] += fabs((double)(float)((double)obj[3 * i] - Mc[0]) - cM[0]);
        sM[1] += fabs((double)(float)((double)obj[3 * i + 1] - Mc[1]) - cM[1]);
        sm[0] += fabs((double)(float)ux - cm[0]), sm[1] += fabs((double)(float)uy - cm[1]);
    }
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    bool hok = !(fabs(sM[0]) < DBL_EPSILON || fabs(sM[1]) < DBL_EPSILON || fabs(sm[0]) < DBL_EPSILON || fabs(sm[1]) < DBL_EPSILON);
    double H[9];
    if (hok) {
        sM[0] = n / sM[0], sM[1] = n / sM[1], sm[0] = n / sm[0], sm[1] = n / sm[1];
        double A[64], b[8];
        for (int i = 0; i < 64; i++) A[i] = 0;
        for (int i = 0; i < 8; i++) b[i] = 0;
        for (int i = 0; i < n; i++) {
            double ux, uy;
            undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
            double x = ((double)(float)ux - cm[0]) * sm[0], y = ((double)(float)uy - cm[1]) * sm[1];
            double X = ((double)(float)((double)obj[3 * i] - Mc[0]) - cM[0]) * sM[0];
            double Y = ((double)(float)((double)obj[3 * i + 1] - Mc[1]) - cM[1]) * sM[1];
            double Lx[8] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y};
            double Ly[8] = {0, 0, 0, X, Y, 1, -y * X, -y * Y};
            for (int j = 0; j < 8; j++) {
                for (int q = 0; q < 8; q++) A[j * 8 + q] += Lx[j] * Lx[q] + Ly[j] * Ly[q];
                b[j] += Lx[j] * x + Ly[j] * y;
            }
        }
        hok = solve_n(A, b, 8);
        if (hok) {
            double H0[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
            double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
            double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
            double T[9];
            mat3_mul(invHnorm, H0, T);
            mat3_mul(T, Hnorm2, H);
            double s = 1. / H[8];
            for (int i = 0; i < 9; i++) {
                H[i] *= s;
                if (!isfinite(H[i])) hok = false;
            }
        }
    }
    if (hok) {
        double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
        double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
        double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), st = 2. / fmax(h1n + h2n, DBL_EPSILON);
        double h1[3] = {H[0] * s1, H[3] * s1, H[6] * s1}, h2[3] = {H[1] * s2, H[4] * s2, H[7] * s2};
        t[0] = H[2] * st, t[1] = H[5] * st, t[2] = H[8] * st;
        double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
        double R[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
        rodrigues_mat2vec(R, r);
        rodrigues_vec2mat(r, R, nullptr);
        for (int i = 0; i < 3; i++) t[i] += R[i * 3] * (-Mc[0]) + R[i * 3 + 1] * (-Mc[1]);
        rodrigues_mat2vec(R, r);
    }
    // ---- CvLevMarq
    double param[6] = {r[0], r[1], r[2], t[0], t[1], t[2]}, prev[6];
    double JtJ[36], JtErr[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    for (;;) {
        double R[9], dRdr[27];
        rodrigues_vec2mat(param, R, dRdr);
        for (int i = 0; i < 36; i++) JtJ[i] = 0;
        for (int i = 0; i < 6; i++) JtErr[i] = 0;
        double e2 = 0;
        for (int i = 0; i < n; i++) {
            double mx, my, dr[6], dt[6];
            project_point(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], R, dRdr, param + 3, K, k, &mx, &my, dr, dt);
            double ex = mx - (double)img[2 * i], ey = my - (double)img[2 * i + 1];
            double jx[6] = {dr[0], dr[1], dr[2], dt[0], dt[1], dt[2]};
            double jy[6] = {dr[3], dr[4], dr[5], dt[3], dt[4], dt[5]};
            for (int a = 0; a < 6; a++) {
                for (int c = 0; c < 6; c++) JtJ[a * 6 + c] += jx[a] * jx[c] + jy[a] * jy[c];
                JtErr[a] += jx[a] * ex + jy[a] * ey;
            }
            e2 += ex * ex + ey * ey;
        }
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        if (iters == 0) prevErrNorm = sqrt(e2);
        double errNorm;
        bool converged = false;
        for (bool first = true;; first = false) {
            if (!first) {
                if (!(errNorm > prevErrNorm && ++lambdaLg10 <= 16)) break;
            }
            double lambda = lm_lambda(lambdaLg10);
            double A[36], b[6];
            for (int i = 0; i < 36; i++) A[i] = JtJ[i];
            for (int i = 0; i < 6; i++) b[i] = JtErr[i];
            for (int i = 0; i < 6; i++) A[i * 7] *= 1. + lambda;
            if (!solve_n(A, b, 6))
                for (int i = 0; i < 6; i++) b[i] = 0;
            for (int i = 0; i < 6; i++) param[i] = prev[i] - b[i];
            {
                // A step below the solver's own stopping threshold (relative change < FLT_EPSILON) ends the solve here. CvLevMarq would
                // still evaluate the error and, where rounding keeps it from dropping, walk lambda up to 1e16 — about 19 further
                // solve / project rounds that only shrink this step further: the result stays within FLT_EPSILON (1.2e-7) relative of
                // the reference's, three orders below the 1e-4 pose tolerance, at a fifth of the dependent fp64 chain.
                double sn = 0, sd = 0;
                for (int i = 0; i < 6; i++) sn += b[i] * b[i], sd += prev[i] * prev[i];
                if (sqrt(sn) < FLT_EPSILON * sqrt(sd)) {
                    converged = true;
                    break;
                }
            }
            rodrigues_vec2mat(param, R, nullptr);
            e2 = 0;
            for (int i = 0; i < n; i++) {
                double mx, my;
                project_point(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], R, nullptr, param + 3, K, k, &mx, &my, nullptr, nullptr);
                double ex = mx - (double)img[2 * i], ey = my - (double)img[2 * i + 1];
                e2 += ex * ex + ey * ey;
            }
            errNorm = sqrt(e2);
        }
        if (converged) break;
        lambdaLg10 = max(lambdaLg10 - 1, -16);
        double num = 0, den = 0;
        for (int i = 0; i < 6; i++) num += (param[i] - prev[i]) * (param[i] - prev[i]), den += prev[i] * prev[i];
        double change = sqrt(num) / sqrt(den);
        if (++iters >= 20 || change < FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    for (int i = 0; i < 3; i++) rvec[i] = param[i], tvec[i] = param[3 + i];
    return true;
}

// Wave-parallel form of solve_pnp_planar for many points (board pose): the 64 lanes of one wave split the points, every
// accumulation is followed by a butterfly sum so that all lanes hold the same totals and run the small solves redundantly.
// G = lanes that share one problem (a power of two: 64 = the whole wave for a board, 4 = one lane per marker corner)
template <int G>
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int G>
__device__ inline void wave_sum_arr(double* a, int n) {
    for (int i = 0; i < n; i++) a[i] = wave_sum_d<G>(a[i]);
}

template <int G>
__device__ inline bool solve_pnp_planar_wave(const float* obj, const float* img, int n, const CamModel& cam, double* rvec, double* tvec, int lane) {
    if (n < 4) return false;
    const float* K = cam.K;
    const double* k = cam.k;
    double Mc[2] = {0, 0};
    bool nonplanar = false;
    for (int i = lane; i < n; i += G) {
        if (obj[3 * i + 2] != 0.f) nonplanar = true;
        Mc[0] += (double)obj[3 * i], Mc[1] += (double)obj[3 * i + 1];
    }
    if (wave_sum_d<G>(nonplanar ? 1.0 : 0.0) > 0) return false;
    wave_sum_arr<G>(Mc, 2);
    Mc[0] /= n, Mc[1] /= n;
    // ---- homography plane -> normalised image (inputs rounded to float as cv::findHomography does)
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = lane; i < n; i += G) {
        double ux, uy;
        undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
        cM[0] += (double)(float)((double)obj[3 * i] - Mc[0]), cM[1] += (double)(float)((double)obj[3 * i + 1] - Mc[1]);
        cm[0] += (double)(float)ux, cm[1] += (double)(float)uy;
    }
    wave_sum_arr<G>(cM, 2), wave_sum_arr<G>(cm, 2);
    cM[0] /= n, cM[1] /= n, cm[0] /= n, cm[1] /= n;
    for (int i = lane; i < n; i += G) {
        double ux, uy;
        undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
        sM[0] += fabs((double)(float)((double)obj[3 * i] - Mc[0]) - cM[0]);
        sM[1] += fabs((double)(float)((double)obj[3 * i + 1] - Mc[1]) - cM[1]);
        sm[0] += fabs((double)(float)ux - cm[0]), sm[1] += fabs((double)(float)uy - cm[1]);
    }
    wave_sum_arr<G>(sM, 2), wave_sum_arr<G>(sm, 2);
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    bool hok = !(fabs(sM[0]) < DBL_EPSILON || fabs(sM[1]) < DBL_EPSILON || fabs(sm[0]) < DBL_EPSILON || fabs(sm[1]) < DBL_EPSILON);
    double H[9];
    if (hok) {
        sM[0] = n / sM[0], sM[1] = n / sM[1], sm[0] = n / sm[0], sm[1] = n / sm[1];
        double A[64], b[8];
        for (int i = 0; i < 64; i++) A[i] = 0;
        for (int i = 0; i < 8; i++) b[i] = 0;
        for (int i = lane; i < n; i += G) {
            double ux, uy;
            undistort_point(img[2 * i], img[2 * i + 1], K, k, &ux, &uy);
            double x = ((double)(float)ux - cm[0]) * sm[0], y = ((double)(float)uy - cm[1]) * sm[1];
            double X = ((double)(float)((double)obj[3 * i] - Mc[0]) - cM[0]) * sM[0];
            double Y = ((double)(float)((double)obj[3 * i + 1] - Mc[1]) - cM[1]) * sM[1];
            double Lx[8] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y};
            double Ly[8] = {0, 0, 0, X, Y, 1, -y * X, -y * Y};
            for (int j = 0; j < 8; j++) {
                for (int q = 0; q < 8; q++) A[j * 8 + q] += Lx[j] * Lx[q] + Ly[j] * Ly[q];
                b[j] += Lx[j] * x + Ly[j] * y;
            }
        }
        wave_sum_arr<G>(A, 64), wave_sum_arr<G>(b, 8);
        hok = solve_spd<8>(A, b);
        if (hok) {
            double H0[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
            double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
            double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
            double T[9];
            mat3_mul(invHnorm, H0, T);
            mat3_mul(T, Hnorm2, H);
            double s = 1. / H[8];
            for (int i = 0; i < 9; i++) {
                H[i] *= s;
                if (!isfinite(H[i])) hok = false;
            }
        }
    }
    if (hok) {
        double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
        double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
        double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), st = 2. / fmax(h1n + h2n, DBL_EPSILON);
        double h1[3] = {H[0] * s1, H[3] * s1, H[6] * s1}, h2[3] = {H[1] * s2, H[4] * s2, H[7] * s2};
        t[0] = H[2] * st, t[1] = H[5] * st, t[2] = H[8] * st;
        double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
        double R[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
        rodrigues_mat2vec(R, r);
        rodrigues_vec2mat(r, R, nullptr);
        for (int i = 0; i < 3; i++) t[i] += R[i * 3] * (-Mc[0]) + R[i * 3 + 1] * (-Mc[1]);
        rodrigues_mat2vec(R, r);
    }
    // ---- CvLevMarq
    double param[6] = {r[0], r[1], r[2], t[0], t[1], t[2]}, prev[6];
    double JtJ[36], JtErr[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    for (;;) {
        double R[9], dRdr[27];
        rodrigues_vec2mat(param, R, dRdr);
        for (int i = 0; i < 36; i++) JtJ[i] = 0;
        for (int i = 0; i < 6; i++) JtErr[i] = 0;
        double e2 = 0;
        for (int i = lane; i < n; i += G) {
            double mx, my, dr[6], dt[6];
            project_point(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], R, dRdr, param + 3, K, k, &mx, &my, dr, dt);
            double ex = mx - (double)img[2 * i], ey = my - (double)img[2 * i + 1];
            double jx[6] = {dr[0], dr[1], dr[2], dt[0], dt[1], dt[2]};
            double jy[6] = {dr[3], dr[4], dr[5], dt[3], dt[4], dt[5]};
            for (int a = 0; a < 6; a++) {
                for (int c = 0; c < 6; c++) JtJ[a * 6 + c] += jx[a] * jx[c] + jy[a] * jy[c];
                JtErr[a] += jx[a] * ex + jy[a] * ey;
            }
            e2 += ex * ex + ey * ey;
        }
        wave_sum_arr<G>(JtJ, 36), wave_sum_arr<G>(JtErr, 6);
        e2 = wave_sum_d<G>(e2);
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        if (iters == 0) prevErrNorm = sqrt(e2);
        double errNorm;
        bool converged = false;
        for (bool first = true;; first = false) {
            if (!first) {
                if (!(errNorm > prevErrNorm && ++lambdaLg10 <= 16)) break;
            }
            double lambda = lm_lambda(lambdaLg10);
            double A[36], b[6];
            for (int i = 0; i < 36; i++) A[i] = JtJ[i];
            for (int i = 0; i < 6; i++) b[i] = JtErr[i];
            for (int i = 0; i < 6; i++) A[i * 7] *= 1. + lambda;
            if (!solve_spd<6>(A, b))
                for (int i = 0; i < 6; i++) b[i] = 0;
            for (int i = 0; i < 6; i++) param[i] = prev[i] - b[i];
            {
                // A step below the solver's own stopping threshold (relative change < FLT_EPSILON) ends the solve here. CvLevMarq would
                // still evaluate the error and, where rounding keeps it from dropping, walk lambda up to 1e16 — about 19 further
                // solve / project rounds that only shrink this step further: the result stays within FLT_EPSILON (1.2e-7) relative of
                // the reference's, three orders below the 1e-4 pose tolerance, at a fifth of the dependent fp64 chain.
                double sn = 0, sd = 0;
                for (int i = 0; i < 6; i++) sn += b[i] * b[i], sd += prev[i] * prev[i];
                if (sqrt(sn) < FLT_EPSILON * sqrt(sd)) {
                    converged = true;
                    break;
                }
            }
            rodrigues_vec2mat(param, R, nullptr);
            e2 = 0;
            for (int i = lane; i < n; i += G) {
                double mx, my;
                project_point(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], R, nullptr, param + 3, K, k, &mx, &my, nullptr, nullptr);
                double ex = mx - (double)img[2 * i], ey = my - (double)img[2 * i + 1];
                e2 += ex * ex + ey * ey;
            }
            e2 = wave_sum_d<G>(e2);
            errNorm = sqrt(e2);
        }
        if (converged) break;
        lambdaLg10 = max(lambdaLg10 - 1, -16);
        double num = 0, den = 0;
        for (int i = 0; i < 6; i++) num += (param[i] - prev[i]) * (param[i] - prev[i]), den += prev[i] * prev[i];
        double change = sqrt(num) / sqrt(den);
        if (++iters >= 20 || change < FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    for (int i = 0; i < 3; i++) rvec[i] = param[i], tvec[i] = param[3 + i];
    return true;
}

// aruco::rotateXAxis — rotation in float (cv::Matx33f), result kept at float precision
__device__ inline void rotate_x_axis(double* rvec) {
    double Rd[9];
    rodrigues_vec2mat(rvec, Rd, nullptr);
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
    float ang = (float)(3.14159265358979323846 / 2);
    float cs = (float)cos((double)ang), sn = (float)sin((double)ang);
    float RX[9] = {1, 0, 0, 0, cs, -sn, 0, sn, cs};
    double Qd[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float q = R[i * 3] * RX[j] + R[i * 3 + 1] * RX[3 + j] + R[i * 3 + 2] * RX[6 + j];
            Qd[i * 3 + j] = q;
        }
    double r[3];
    rodrigues_mat2vec(Qd, r);
    for (int i = 0; i < 3; i++) rvec[i] = (double)(float)r[i];
}

}  // namespace ah
