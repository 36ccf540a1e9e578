// Device pieces of the decode stage that two kernels share (round 3: fewer, fatter dispatches in the chain).
//   homography_lane   : inverse of cv::getPerspectiveTransform(quad -> patch corners), one candidate per LANE (the 8x8 system of every
//                       lane lives in LDS, element-major) — the tail of frame_candidates_kernel since round 3, its own kernel before
//   cells_decode_wave : FiducidalMarkers::detect on the Otsu-thresholded patch, one candidate per WAVEFRONT — the head of
//                       refine_lines_kernel for the built-in 5x5 decoder since round 3, its own kernel for the other entry points
// Reference: MarkerDetector::warp src/markerdetector.cpp:684-697, FiducidalMarkers::detect src/arucofidmarkers.cpp:438-452 with
// analyzeMarkerImage :100-137, checkBorders :168-184, getMarkerCode :189-204, hammDistMarker :74-98, rotate :63-72.
#pragma once
#include "internal.h"

namespace ah {

// The 8x8 systems of HOMOGRAPHY_GROUP lanes at a time live in LDS, element-major: element k of a lane's matrix at k * GROUP + (lane % GROUP), so the lanes
// of a group never collide on a bank. A wave solves its 64 candidates group after group (round 3: all 64 at once took 37 KB, and with the batches in
// flight a workgroup that wants a quarter of a CU's LDS waits for it: 0.61 ms in the stream against 0.07 alone).
// One frame per call (round 4): a single workgroup has the CU's LDS to itself, so all 64 lanes solve at once (frame_candidates_kernel<64>: 46 -> 2x us for the
// 48 candidates of a 1080p frame).
constexpr int HOMOGRAPHY_GROUP = 16;
template <int G>
struct LaneMatT {
    double* base;
    int lane;
    __device__ __forceinline__ double& operator[](int k) const { return base[k * G + (lane & (G - 1))]; }
};
using LaneMat = LaneMatT<HOMOGRAPHY_GROUP>;
template <int G>
constexpr int homography_lds_doubles() { return 64 * G + 8 * G; }   // A (8 x 8) and b (8) of a group
constexpr int HOMOGRAPHY_LDS_DOUBLES = homography_lds_doubles<HOMOGRAPHY_GROUP>();

// qx / qy: the candidate's integer corners; iM: 9 doubles. Gaussian elimination with partial pivoting, same operation order as the CPU
// restatement of cv::getPerspectiveTransform + the inversion cv::warpPerspective starts with.
template <typename Mat>
__device__ __forceinline__ void homography_lane(const int16_t* qx, const int16_t* qy, int ws, const Mat& A, const Mat& b, double* iM) {
    const double d = (double)(float)(ws - 1);
    const double dxs[4] = {0, d, d, 0}, dys[4] = {0, 0, d, d};
    for (int i = 0; i < 64; i++) A[i] = 0;
    for (int i = 0; i < 4; i++) {
        const double sx = (double)(float)qx[i], sy = (double)(float)qy[i], dx = dxs[i], dy = dys[i];
        const int r0 = i * 8, r1 = (i + 4) * 8;
        A[r0 + 0] = sx, A[r1 + 3] = sx;
        A[r0 + 1] = sy, A[r1 + 4] = sy;
        A[r0 + 2] = 1, A[r1 + 5] = 1;
        A[r0 + 6] = -sx * dx;
        A[r0 + 7] = -sy * dx;
        A[r1 + 6] = -sx * dy;
        A[r1 + 7] = -sy * dy;
        b[i] = dx;
        b[i + 4] = dy;
    }
    bool ok = true;
    for (int c = 0; c < 8 && ok; c++) {
        int piv = c;
        double best = fabs(A[c * 8 + c]);
        for (int r = c + 1; r < 8; r++) {
            double v = fabs(A[r * 8 + c]);
            if (v > best) best = v, piv = r;
        }
        if (best == 0) {
            ok = false;
            break;
        }
        if (piv != c) {
            for (int k = 0; k < 8; k++) {
                double t = A[c * 8 + k];
                A[c * 8 + k] = A[piv * 8 + k];
                A[piv * 8 + k] = t;
            }
            double t = b[c];
            b[c] = b[piv];
            b[piv] = t;
        }
        const double inv = 1.0 / A[c * 8 + c];
        for (int r = c + 1; r < 8; r++) {
            const double f = A[r * 8 + c] * inv;
            if (f == 0) continue;
            for (int k = c; k < 8; k++) A[r * 8 + k] -= f * A[c * 8 + k];
            b[r] -= f * b[c];
        }
    }
    double m[9];
    if (ok) {
        for (int r = 7; r >= 0; r--) {
            double sacc = b[r];
            for (int k = r + 1; k < 8; k++) sacc -= A[r * 8 + k] * b[k];
            b[r] = sacc / A[r * 8 + r];
        }
        for (int i = 0; i < 8; i++) m[i] = b[i];
    } else {
        for (int i = 0; i < 8; i++) m[i] = 0;
    }
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

// 5d: one wavefront per candidate — 7x7 cell votes on the binarised patch and the 5x5 Hamming decode
// 5x5 code as five 5-bit rows, bit x = column x (so the dictionary words of hammDistMarker are used bit-reversed)
__device__ __forceinline__ int hamm_rows(const uint32_t v[5]) {
    const uint32_t words[4] = {0x01, 0x1D, 0x12, 0x0E};   // 10000, 10111, 01001, 01110 reversed
    int dist = 0;
#pragma unroll
    for (int y = 0; y < 5; y++) {
        int best = 100000;
#pragma unroll
        for (int p = 0; p < 4; p++) best = min(best, __popc(v[y] ^ words[p]));
        dist += best;
    }
    return dist;
}

// all 64 lanes call it; *id / *nrot are valid on lane 0 (id = -1: not a marker)
__device__ __forceinline__ void cells_decode_wave(const uint8_t* patch, int ws, int thr, int lane, int* id_out, int* nrot_out) {
    const int sw = ws / 7;
    const int half = (sw * sw) / 2;
    bool white = false;
    if (lane < 49) {   // cell (cy,cx): white iff more than half of its pixels exceed the Otsu threshold
        const int cy = lane / 7, cx = lane - cy * 7;
        int cnt = 0;
        if (sw == 8 && (((size_t)patch | (size_t)ws) & 7) == 0) {   // default 56x56 patch: a cell row is one aligned 8-byte load
#pragma unroll
            for (int y = 0; y < 8; y++) {
                const uint2 w = *(const uint2*)(patch + (cy * 8 + y) * ws + cx * 8);
#pragma unroll
                for (int b = 0; b < 4; b++) cnt += (int)((w.x >> (8 * b)) & 0xFFu) > thr, cnt += (int)((w.y >> (8 * b)) & 0xFFu) > thr;
            }
        } else {
            for (int y = 0; y < sw; y++)
                for (int x = 0; x < sw; x++) cnt += patch[(cy * sw + y) * ws + cx * sw + x] > thr;
        }
        white = cnt > half;
    }
    const unsigned long long m = __ballot(white);   // bit cy*7+cx
    int id = -1, nrot = 0;
    if (lane == 0) {
        // checkBorders: all 24 frame cells must be black
        unsigned long long border = 0x7Full | (0x7Full << 42);
#pragma unroll
        for (int y = 1; y < 6; y++) border |= (1ull << (7 * y)) | (1ull << (7 * y + 6));
        if ((m & border) == 0) {
            uint32_t cur[5], best[5];
#pragma unroll
            for (int y = 0; y < 5; y++) cur[y] = (uint32_t)(m >> (7 * (y + 1) + 1)) & 31u, best[y] = cur[y];
            int min_dist = hamm_rows(cur);
#pragma unroll
            for (int r = 1; r < 4; r++) {
                uint32_t nxt[5];   // rotate: new[i][j] = old[4-j][i]
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    nxt[i] = 0;
#pragma unroll
                    for (int j = 0; j < 5; j++) nxt[i] |= ((cur[4 - j] >> i) & 1u) << j;
                }
#pragma unroll
                for (int i = 0; i < 5; i++) cur[i] = nxt[i];
                const int dd = hamm_rows(cur);
                if (dd < min_dist) {
                    min_dist = dd, nrot = r;
#pragma unroll
                    for (int i = 0; i < 5; i++) best[i] = cur[i];
                }
            }
            if (min_dist == 0) {
                id = 0;
#pragma unroll
                for (int y = 0; y < 5; y++) id |= (int)((((best[y] >> 1) & 1u) << 1) | ((best[y] >> 3) & 1u)) << (2 * (4 - y));
            }
        }
    }
    *id_out = id, *nrot_out = nrot;
}

}  // namespace ah
