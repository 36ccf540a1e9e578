// Kernel 5 — perspective warp + Otsu + 5x5 Hamming decode, one wavefront per candidate.
//
// Reference: MarkerDetector::warp (/root/reference/src/markerdetector.cpp:684-697: getPerspectiveTransform +
// warpPerspective INTER_NEAREST from the GRAY frame with the integer corners) and FiducidalMarkers::detect
// (/root/reference/src/arucofidmarkers.cpp:438-452 -> analyzeMarkerImage :100-137, checkBorders :168-184,
// getMarkerCode :189-204, hammDistMarker :74-98, rotate :63-72).
// The 8x8 homography system and the Otsu sweep run on lane 0 in double with a fixed operation order (the library is
// built with -ffp-contract=off), the 56x56 gather, histogram and cell counts use all 64 lanes through LDS.
#include <float.h>
#include <limits.h>

#include "internal.h"

namespace ah {

constexpr int MAX_WARP = 128;

// 8x8 dense solve, partial pivoting, on LDS arrays (lane 0).
__device__ static bool solve8(double* A, double* b) {
    const int n = 8;
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) {
            double v = fabs(A[r * n + c]);
            if (v > best) best = v, piv = r;
        }
        if (best == 0) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) {
                double t = A[c * n + k];
                A[c * n + k] = A[piv * n + k];
                A[piv * n + k] = t;
            }
            double t = b[c];
            b[c] = b[piv];
            b[piv] = t;
        }
        double inv = 1.0 / A[c * n + c];
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] * inv;
            if (f == 0) continue;
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

// inverse map of cv::getPerspectiveTransform(quad -> (0,0),(s-1,0),(s-1,s-1),(0,s-1)) into iM (lane 0, LDS scratch)
__device__ static void inverse_homography(const float* quad, int size, double* A, double* b, double* iM) {
    const double d = (double)(float)(size - 1);
    const double dxs[4] = {0, d, d, 0}, dys[4] = {0, 0, d, d};
    for (int i = 0; i < 64; i++) A[i] = 0;
    for (int i = 0; i < 4; i++) {
        double sx = quad[2 * i], sy = quad[2 * i + 1], dx = dxs[i], dy = dys[i];
        double* r0 = A + i * 8;
        double* r1 = A + (i + 4) * 8;
        r0[0] = r1[3] = sx;
        r0[1] = r1[4] = sy;
        r0[2] = r1[5] = 1;
        r0[6] = -sx * dx;
        r0[7] = -sy * dx;
        r1[6] = -sx * dy;
        r1[7] = -sy * dy;
        b[i] = dx;
        b[i + 4] = dy;
    }
    if (!solve8(A, b))
        for (int i = 0; i < 8; i++) b[i] = 0;
    double m[9];
    for (int i = 0; i < 8; i++) m[i] = b[i];
    m[8] = 1.0;
    double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    det = det != 0 ? 1. / det : 0;
    iM[0] = (m[4] * m[8] - m[5] * m[7]) * det;
    iM[1] = (m[2] * m[7] - m[1] * m[8]) * det;
    iM[2] = (m[1] * m[5] - m[2] * m[4]) * det;
    iM[3] = (m[5] * m[6] - m[3] * m[8]) * det;
    iM[4] = (m[0] * m[8] - m[2] * m[6]) * det;
    iM[5] = (m[2] * m[3] - m[0] * m[5]) * det;
    iM[6] = (m[3] * m[7] - m[4] * m[6]) * det;
    iM[7] = (m[1] * m[6] - m[0] * m[7]) * det;
    iM[8] = (m[0] * m[4] - m[1] * m[3]) * det;
}

// nearest-neighbour gather of one patch pixel (BORDER_CONSTANT 0), cvRound = round-half-even
__device__ __forceinline__ uint8_t warp_pixel(const uint8_t* src, int W, int H, size_t stride, const double* iM, int x, int y) {
    double X0 = iM[1] * y + iM[2];
    double Y0 = iM[4] * y + iM[5];
    double W0 = iM[7] * y + iM[8];
    double Wd = W0 + iM[6] * x;
    Wd = Wd != 0 ? 1. / Wd : 0;
    double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + iM[0] * x) * Wd));
    double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + iM[3] * x) * Wd));
    long long X = __double2ll_rn(fX), Y = __double2ll_rn(fY);
    if (X >= 0 && X < W && Y >= 0 && Y < H) return src[(size_t)Y * stride + X];
    return 0;
}

__device__ static int hamm_dist(const uint8_t b[5][5]) {
    const uint8_t words[4] = {0x10, 0x17, 0x09, 0x0E};
    int dist = 0;
    for (int y = 0; y < 5; y++) {
        int row = 0;
        for (int x = 0; x < 5; x++) row |= b[y][x] << (4 - x);
        int best = 100000;
        for (int p = 0; p < 4; p++) best = min(best, __popc((unsigned)(row ^ words[p])));
        dist += best;
    }
    return dist;
}

struct DecodeArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height, ws;
    Cand* cands;
    const int32_t* ncands;
    int cap_cands;
};

__global__ __launch_bounds__(64) void decode_kernel(DecodeArgs a) {
    extern __shared__ __align__(16) uint8_t patch[];   // ws*ws bytes
    __shared__ int hist[256];
    __shared__ double sA[64], sb[8], siM[9];
    __shared__ int s_thr;
    __shared__ uint8_t s_cell[49];
    const int frame = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= a.ncands[frame]) return;
    Cand* cand = a.cands + (size_t)frame * a.cap_cands + ci;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int ws = a.ws, npx = ws * ws;
    for (int i = lane; i < 256; i += WAVE) hist[i] = 0;
    if (lane == 0) {
        float q[8];
        for (int k = 0; k < 4; k++) q[2 * k] = (float)cand->qx[k], q[2 * k + 1] = (float)cand->qy[k];
        inverse_homography(q, ws, sA, sb, siM);
    }
    __syncthreads();
    for (int i = lane; i < npx; i += WAVE) {
        int y = i / ws, x = i - y * ws;
        uint8_t v = warp_pixel(src, a.width, a.height, a.row_stride, siM, x, y);
        patch[i] = v;
        atomicAdd(&hist[v], 1);
    }
    __syncthreads();
    if (lane == 0) {  // getThreshVal_Otsu_8u
        double mu = 0, scale = 1. / npx;
        for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
        mu *= scale;
        double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
        for (int i = 0; i < 256; i++) {
            double p_i = hist[i] * scale;
            mu1 *= q1;
            q1 += p_i;
            double q2 = 1. - q1;
            if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) continue;
            mu1 = (mu1 + i * p_i) / q1;
            double mu2 = (mu - q1 * mu1) / q2;
            double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
            if (sigma > max_sigma) {
                max_sigma = sigma;
                max_val = i;
            }
        }
        s_thr = (int)max_val;
    }
    __syncthreads();
    const int thr = s_thr, sw = ws / 7;
    if (lane < 49) {  // 7x7 cells: white iff more than half of the pixels exceed the Otsu threshold
        int cy = lane / 7, cx = lane - cy * 7, cnt = 0;
        for (int y = 0; y < sw; y++)
            for (int x = 0; x < sw; x++) cnt += patch[(cy * sw + y) * ws + cx * sw + x] > thr;
        s_cell[lane] = cnt > (sw * sw) / 2;
    }
    __syncthreads();
    if (lane == 0) {
        int id = -1, nrot = 0;
        bool border_ok = true;
        for (int y = 0; y < 7 && border_ok; y++) {
            int inc = (y == 0 || y == 6) ? 1 : 6;
            for (int x = 0; x < 7; x += inc)
                if (s_cell[y * 7 + x]) {
                    border_ok = false;
                    break;
                }
        }
        if (border_ok) {
            uint8_t rot[2][5][5];
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) rot[0][y][x] = s_cell[(y + 1) * 7 + x + 1];
            int min_dist = hamm_dist(rot[0]);
            uint8_t best[5][5];
            for (int y = 0; y < 5; y++)
                for (int x = 0; x < 5; x++) best[y][x] = rot[0][y][x];
            int cur = 0;
            for (int r = 1; r < 4; r++) {
                int nxt = cur ^ 1;
                for (int i = 0; i < 5; i++)
                    for (int j = 0; j < 5; j++) rot[nxt][i][j] = rot[cur][5 - j - 1][i];
                cur = nxt;
                int d = hamm_dist(rot[cur]);
                if (d < min_dist) {
                    min_dist = d, nrot = r;
                    for (int y = 0; y < 5; y++)
                        for (int x = 0; x < 5; x++) best[y][x] = rot[cur][y][x];
                }
            }
            if (min_dist == 0) {
                id = 0;
                for (int y = 0; y < 5; y++) id |= (best[y][1] << 1 | best[y][3]) << 2 * (4 - y);
            }
        }
        cand->id = id;
        cand->nrot = nrot;
    }
}

void launch_decode(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    DecodeArgs a;
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride, a.width = g.width, a.height = g.height;
    a.ws = p.warp_size, a.cands = b.cands, a.ncands = b.ncands, a.cap_cands = b.cap_cands;
    hipLaunchKernelGGL(decode_kernel, dim3(b.cap_cands, nframes), dim3(64), (size_t)((p.warp_size * p.warp_size + 15) & ~15), s, a);
}

// MarkerDetector::warp as a stage entry point: one patch from one quad
__global__ __launch_bounds__(64) void warp_only_kernel(const uint8_t* gray, int W, int H, size_t stride, const float* quad, int ws, uint8_t* dst) {
    __shared__ double sA[64], sb[8], siM[9];
    if (threadIdx.x == 0) inverse_homography(quad, ws, sA, sb, siM);
    __syncthreads();
    for (int i = threadIdx.x; i < ws * ws; i += WAVE) {
        int y = i / ws, x = i - y * ws;
        dst[i] = warp_pixel(gray, W, H, stride, siM, x, y);
    }
}

void launch_warp_only(hipStream_t s, const uint8_t* gray, const FrameGeom& g, const float* quad_dev, int size, uint8_t* dst_dev) {
    hipLaunchKernelGGL(warp_only_kernel, dim3(1), dim3(64), 0, s, gray, g.width, g.height, g.row_stride, quad_dev, size, dst_dev);
}

}  // namespace ah
