// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Restatements of the OpenCV 3.0 calib3d primitives used on the reference's hot path:
// Rodrigues, undistortPoints, projectPoints and solvePnP(SOLVEPNP_ITERATIVE)
// (call sites: src/markerdetector.cpp:458,959,152; src/marker.cpp:118; src/boarddetector.cpp:157,174,193;
//  src/utils.cpp:16-30). Algorithm outline: SURVEY.md Appendix A.8/A.9.
#include "orc.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace orc {

static bool solve_sym(double* A, double* b, int n) {  // Gaussian elimination, partial pivoting
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (std::fabs(A[r * n + c]) > best) best = std::fabs(A[r * n + c]), piv = r;
        if (best == 0) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        double inv = 1.0 / A[c * n + c];
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] * inv;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

static void mat3_mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, t, sizeof(t));
}

static double mat3_det(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// Orthogonal polar factor U*Vt of a near-rotation 3x3 (what cvRodrigues2 obtains with an SVD), by Newton iteration.
static void orthonormalise(double* R) {
    for (int it = 0; it < 30; it++) {
        double d = mat3_det(R);
        if (d == 0) return;
        double inv_t[9];  // (R^-1)^T = cofactor / det
        inv_t[0] = (R[4] * R[8] - R[5] * R[7]) / d;
        inv_t[1] = (R[5] * R[6] - R[3] * R[8]) / d;
        inv_t[2] = (R[3] * R[7] - R[4] * R[6]) / d;
        inv_t[3] = (R[2] * R[7] - R[1] * R[8]) / d;
        inv_t[4] = (R[0] * R[8] - R[2] * R[6]) / d;
        inv_t[5] = (R[1] * R[6] - R[0] * R[7]) / d;
        inv_t[6] = (R[1] * R[5] - R[2] * R[4]) / d;
        inv_t[7] = (R[2] * R[3] - R[0] * R[5]) / d;
        inv_t[8] = (R[0] * R[4] - R[1] * R[3]) / d;
        double diff = 0;
        for (int k = 0; k < 9; k++) {
            double n = 0.5 * (R[k] + inv_t[k]);
            diff = std::max(diff, std::fabs(n - R[k]));
            R[k] = n;
        }
        if (diff < 1e-16) break;
    }
}

// cv::Rodrigues, vector -> matrix with optional derivative dRdr[j*9+k] = dR[k]/dr[j].
void rodrigues_to_mat(const double r[3], double R[9], double dRdr[27]) {
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(R, I, sizeof(I));
        if (dRdr) {
            static const double J0[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
            std::memcpy(dRdr, J0, sizeof(J0));
        }
        return;
    }
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = 1. / theta;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    if (dRdr) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        static const double d_rx_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                dRdr[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx_[k] + a4 * d_rx_[i * 9 + k];
        }
    }
}

// cv::Rodrigues, matrix -> vector (input is first projected onto the rotation group).
void rodrigues_to_vec(const double Rin[9], double r[3]) {
    double R[9];
    std::memcpy(R, Rin, sizeof(R));
    orthonormalise(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

static void load_dist(const float* dist, int ndist, double k[8]) {
    for (int i = 0; i < 8; i++) k[i] = (dist && i < ndist) ? (double)dist[i] : 0.0;
}

// Inverse Brown model, 5 fixed-point iterations (cvUndistortPoints, OpenCV 3.0), double precision.
void undistort_points_d(const double* src, int n, const double K[9], const double k[8], double* dst) {
    double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double x = (src[2 * i] - cx) * ifx, y = (src[2 * i + 1] - cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        dst[2 * i] = x;
        dst[2 * i + 1] = y;
    }
}

// cv::undistortPoints(src, dst, K, dist, R = I, P) on float points (src/markerdetector.cpp:959 passes P = K).
void undistort_points(const Pt2f* src, int n, const float K[9], const float* dist, int ndist, const float* P, Pt2f* dst) {
    double Kd[9], k[8];
    for (int i = 0; i < 9; i++) Kd[i] = K[i];
    load_dist(dist, ndist, k);
    for (int i = 0; i < n; i++) {
        double s[2] = {src[i].x, src[i].y}, d[2];
        undistort_points_d(s, 1, Kd, k, d);
        double x = d[0], y = d[1];
        if (P) {
            double xx = (double)P[0] * x + (double)P[1] * y + (double)P[2];
            double yy = (double)P[3] * x + (double)P[4] * y + (double)P[5];
            double ww = 1. / ((double)P[6] * x + (double)P[7] * y + (double)P[8]);
            x = xx * ww;
            y = yy * ww;
        }
        dst[i].x = (float)x;
        dst[i].y = (float)y;
    }
}

// cvProjectPoints2 (forward Brown model k1,k2,p1,p2,k3,k4,k5,k6) with derivatives w.r.t. rvec and tvec.
void project_points(const double* M, int n, const double r[3], const double t[3], const double K[9], const double k[8],
                    double* m, double* dpdr, double* dpdt) {
    double R[9], dRdr[27];
    rodrigues_to_mat(r, R, dpdr ? dRdr : nullptr);
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
        double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
        m[2 * i] = xd * fx + cx;
        m[2 * i + 1] = yd * fy + cy;
        if (!dpdr && !dpdt) continue;
        auto chain = [&](double dxd, double dyd, double& omx, double& omy) {
            double dr2 = 2 * x * dxd + 2 * y * dyd;
            double dcdist = k[0] * dr2 + 2 * k[1] * r2 * dr2 + 3 * k[4] * r4 * dr2;
            double dicdist2 = -icdist2 * icdist2 * (k[5] * dr2 + 2 * k[6] * r2 * dr2 + 3 * k[7] * r4 * dr2);
            double da1 = 2 * (x * dyd + y * dxd);
            omx = fx * (dxd * cdist * icdist2 + x * dcdist * icdist2 + x * cdist * dicdist2 + k[2] * da1 +
                        k[3] * (dr2 + 2 * x * dxd));
            omy = fy * (dyd * cdist * icdist2 + y * dcdist * icdist2 + y * cdist * dicdist2 + k[2] * (dr2 + 2 * y * dyd) +
                        k[3] * da1);
        };
        if (dpdt) {
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) chain(dxdt[j], dydt[j], dpdt[(2 * i) * 3 + j], dpdt[(2 * i + 1) * 3 + j]);
        }
        if (dpdr) {
            for (int j = 0; j < 3; j++) {
                const double* d = dRdr + j * 9;
                double dx0 = X * d[0] + Y * d[1] + Z * d[2];
                double dy0 = X * d[3] + Y * d[4] + Z * d[5];
                double dz0 = X * d[6] + Y * d[7] + Z * d[8];
                double dxdr = z * (dx0 - x * dz0), dydr = z * (dy0 - y * dz0);
                chain(dxdr, dydr, dpdr[(2 * i) * 3 + j], dpdr[(2 * i + 1) * 3 + j]);
            }
        }
    }
}

// Homography plane(X,Y) -> normalised image (x,y) by normalised DLT (cv::findHomography, method 0; for 4 points
// the solution is exact, for more points OpenCV polishes an algebraic estimate — only used as the LM start).
static bool find_homography(const double* Mxy, const double* mxy, int n, double H[9]) {
    // findHomography converts both point sets to float first
    std::vector<double> M(2 * n), m(2 * n);
    for (int i = 0; i < 2 * n; i++) M[i] = (double)(float)Mxy[i], m[i] = (double)(float)mxy[i];
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++) cM[0] += M[2 * i], cM[1] += M[2 * i + 1], cm[0] += m[2 * i], cm[1] += m[2 * i + 1];
    cM[0] /= n, cM[1] /= n, cm[0] /= n, cm[1] /= n;
    for (int i = 0; i < n; i++) {
        sM[0] += std::fabs(M[2 * i] - cM[0]), sM[1] += std::fabs(M[2 * i + 1] - cM[1]);
        sm[0] += std::fabs(m[2 * i] - cm[0]), sm[1] += std::fabs(m[2 * i + 1] - cm[1]);
    }
    if (std::fabs(sM[0]) < DBL_EPSILON || std::fabs(sM[1]) < DBL_EPSILON || std::fabs(sm[0]) < DBL_EPSILON ||
        std::fabs(sm[1]) < DBL_EPSILON)
        return false;
    sM[0] = n / sM[0], sM[1] = n / sM[1], sm[0] = n / sm[0], sm[1] = n / sm[1];
    // least squares for h (h9 = 1) in normalised coordinates: normal equations 8x8
    double A[64], b[8];
    std::memset(A, 0, sizeof(A));
    std::memset(b, 0, sizeof(b));
    for (int i = 0; i < n; i++) {
        double x = (m[2 * i] - cm[0]) * sm[0], y = (m[2 * i + 1] - cm[1]) * sm[1];
        double X = (M[2 * i] - cM[0]) * sM[0], Y = (M[2 * i + 1] - cM[1]) * sM[1];
        double Lx[8] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y};
        double Ly[8] = {0, 0, 0, X, Y, 1, -y * X, -y * Y};
        for (int j = 0; j < 8; j++) {
            for (int q = 0; q < 8; q++) A[j * 8 + q] += Lx[j] * Lx[q] + Ly[j] * Ly[q];
            b[j] += Lx[j] * x + Ly[j] * y;
        }
    }
    if (!solve_sym(A, b, 8)) return false;
    double H0[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
    double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
    double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double T[9];
    mat3_mul(invHnorm, H0, T);
    mat3_mul(T, Hnorm2, H);
    double s = 1. / H[8];
    for (int i = 0; i < 9; i++) H[i] *= s;
    for (int i = 0; i < 9; i++)
        if (!std::isfinite(H[i])) return false;
    return true;
}

// cv::solvePnP(obj, img, K, dist, rvec, tvec) default flags = ITERATIVE, no extrinsic guess
// (cvFindExtrinsicCameraParams2, OpenCV 3.0): planar-homography start + CvLevMarq (<=20 iterations, eps FLT_EPSILON).
bool solve_pnp_iterative(const Pt3f* obj, const Pt2f* img, int n, const float Kf[9], const float* dist, int ndist,
                         double rvec[3], double tvec[3]) {
    if (n < 4) return false;
    double K[9], k[8];
    for (int i = 0; i < 9; i++) K[i] = Kf[i];
    load_dist(dist, ndist, k);
    std::vector<double> M(3 * n), m(2 * n), mn(2 * n);
    for (int i = 0; i < n; i++) {
        M[3 * i] = obj[i].x, M[3 * i + 1] = obj[i].y, M[3 * i + 2] = obj[i].z;
        m[2 * i] = img[i].x, m[2 * i + 1] = img[i].y;
    }
    undistort_points_d(m.data(), n, K, k, mn.data());

    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) Mc[0] += M[3 * i], Mc[1] += M[3 * i + 1], Mc[2] += M[3 * i + 2];
    Mc[0] /= n, Mc[1] /= n, Mc[2] /= n;
    // The reference only ever passes z == 0 object points (marker and board corners): the planar branch with
    // R_transform = identity (third right-singular vector is +-z). Non-planar input is outside this path.
    for (int i = 0; i < n; i++)
        if (M[3 * i + 2] != 0) return false;
    double Tt[3] = {-Mc[0], -Mc[1], -Mc[2]};
    std::vector<double> Mxy(2 * n);
    for (int i = 0; i < n; i++) Mxy[2 * i] = M[3 * i] + Tt[0], Mxy[2 * i + 1] = M[3 * i + 1] + Tt[1];

    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    double H[9];
    if (find_homography(Mxy.data(), mn.data(), n, H)) {
        double h1n = std::sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
        double h2n = std::sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
        double s1 = 1. / std::max(h1n, DBL_EPSILON), s2 = 1. / std::max(h2n, DBL_EPSILON);
        double st = 2. / std::max(h1n + h2n, DBL_EPSILON);
        double h1[3] = {H[0] * s1, H[3] * s1, H[6] * s1}, h2[3] = {H[1] * s2, H[4] * s2, H[7] * s2};
        t[0] = H[2] * st, t[1] = H[5] * st, t[2] = H[8] * st;
        double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
        double R[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
        rodrigues_to_vec(R, r);
        rodrigues_to_mat(r, R, nullptr);
        for (int i = 0; i < 3; i++) t[i] += R[i * 3] * Tt[0] + R[i * 3 + 1] * Tt[1] + R[i * 3 + 2] * Tt[2];
        rodrigues_to_vec(R, r);
    }

    // CvLevMarq(6 params, 2n residuals, max_iter 20, eps FLT_EPSILON, completeSymm)
    double param[6] = {r[0], r[1], r[2], t[0], t[1], t[2]}, prev[6];
    std::vector<double> J(2 * n * 6), err(2 * n), dpdr(2 * n * 3), dpdt(2 * n * 3), proj(2 * n);
    double JtJ[36], JtErr[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX;
    auto calc_err = [&](bool with_j) {
        project_points(M.data(), n, param, param + 3, K, k, proj.data(), with_j ? dpdr.data() : nullptr,
                       with_j ? dpdt.data() : nullptr);
        for (int i = 0; i < 2 * n; i++) err[i] = proj[i] - m[i];
        if (with_j)
            for (int i = 0; i < 2 * n; i++)
                for (int j = 0; j < 3; j++) J[i * 6 + j] = dpdr[i * 3 + j], J[i * 6 + 3 + j] = dpdt[i * 3 + j];
    };
    auto norm = [&](const std::vector<double>& v) {
        double s = 0;
        for (double e : v) s += e * e;
        return std::sqrt(s);
    };
    auto step = [&]() {
        double lambda = std::exp(lambdaLg10 * std::log(10.));
        double A[36], b[6];
        std::memcpy(A, JtJ, sizeof(A));
        std::memcpy(b, JtErr, sizeof(b));
        for (int i = 0; i < 6; i++) A[i * 7] *= 1. + lambda;
        if (!solve_sym(A, b, 6)) std::memset(b, 0, sizeof(b));
        for (int i = 0; i < 6; i++) param[i] = prev[i] - b[i];
    };
    for (;;) {
        // state CALC_J
        calc_err(true);
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int q = 0; q < 2 * n; q++) s += J[q * 6 + i] * J[q * 6 + j];
                JtJ[i * 6 + j] = s;
            }
            double s = 0;
            for (int q = 0; q < 2 * n; q++) s += J[q * 6 + i] * err[q];
            JtErr[i] = s;
        }
        std::memcpy(prev, param, sizeof(prev));
        if (iters == 0) prevErrNorm = norm(err);
        step();
        // state CHECK_ERR (possibly repeated with growing lambda)
        double errNorm;
        for (;;) {
            calc_err(false);
            errNorm = norm(err);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
                step();
                continue;
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double num = 0, den = 0;
        for (int i = 0; i < 6; i++) num += (param[i] - prev[i]) * (param[i] - prev[i]), den += prev[i] * prev[i];
        double change = std::sqrt(num) / std::sqrt(den);
        if (++iters >= 20 || change < FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    for (int i = 0; i < 3; i++) rvec[i] = param[i], tvec[i] = param[3 + i];
    return true;
}

// aruco::rotateXAxis (src/utils.cpp:16-30): R computed and multiplied in float (Matx33f), Rodrigues back.
void rotate_x_axis(double rvec[3]) {
    double Rd[9];
    rodrigues_to_mat(rvec, Rd, nullptr);
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
    float ang = (float)(M_PI / 2);
    float RX[9] = {1, 0, 0, 0, (float)std::cos(ang), (float)-std::sin(ang), 0, (float)std::sin(ang), (float)std::cos(ang)};
    float Q[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Q[i * 3 + j] = R[i * 3] * RX[j] + R[i * 3 + 1] * RX[3 + j] + R[i * 3 + 2] * RX[6 + j];
    double Qd[9];
    for (int i = 0; i < 9; i++) Qd[i] = Q[i];
    double r[3];
    rodrigues_to_vec(Qd, r);
    // the reference stores the result as CV_32F inside the Mat_<double> member: float precision
    for (int i = 0; i < 3; i++) rvec[i] = (double)(float)r[i];
}

}  // namespace orc
