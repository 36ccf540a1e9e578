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

#include <algorithm>

#include "internal.h"
#include "decode_device.h"

namespace ah {

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

// The decode stage is four small kernels so that the serial double-precision parts (8x8 homography solve, 256-bin Otsu
// sweep) run one candidate per LANE while the pixel work (56x56 gather, histogram, cell counts) runs one candidate per
// WAVEFRONT. Candidates of all frames are addressed through the flat list the frame kernel appended to.
struct DecodeArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height, ws;
    Cand* cands;
    int cap_cands;
    const uint32_t* cand_list;   // frame << 16 | index
    const uint32_t* counters;    // [CNT_NCAND] = entries in cand_list
    uint32_t cap_flat;
    double* iM;                  // [cap_flat][9]
    uint16_t* hist;              // [cap_flat][256]
    int32_t* othr;               // [cap_flat]
    uint8_t* patches;            // [cap_flat][ws*ws]
};

// 5a (inverse homography, one candidate per lane) is the tail of frame_candidates_kernel since round 3: decode_device.h, k_contours.hip

// 5b: one wavefront per candidate — gather the ws x ws patch and build its 256-bin histogram.
// The wave takes the patch in 8x8-pixel blocks, lane = pixel of the block, GQ = 7 blocks (a block row of a 56x56 patch) in flight so that the gathers
// overlap (round 1 mapped lanes to the 56 pixels of a patch row: for a rotated marker every lane then reads a different
// image row; round 2: 0.75 -> 0.56 ms per 1024 frames). Every pixel forms exactly the products and sums cv::warpPerspective
// forms (iM[0]*x + (iM[1]*y + iM[2]) ...).
// Histogram of a patch: a marker patch is two-valued, so most lanes hit one of two bins and LDS atomics on one
// histogram serialise 50-fold. 32 private copies (one per lane pair) with byte counters packed four to a word cost the same
// 8 KB as 8 word-sized copies: a copy sees at most 2 * (ws / 8)^2 <= 128 pixels when ws <= 64, so a byte never
// overflows into its neighbour; wider patches use word counters in 8 copies.
#ifndef HCOPIES_N
#define HCOPIES_N 32
#endif
constexpr int HCOPIES = HCOPIES_N, HPITCH = 65;   // words per copy: 64 (bins 4w .. 4w+3) + 1 so that copies start in different banks
constexpr int HWCOPIES = HCOPIES / 4, HWPITCH = 257;
constexpr int HLANES = WAVE / HCOPIES;            // lanes that share a copy
#ifndef GQ_N
#define GQ_N 7
#endif
#ifndef XCD_RUN_N
#define XCD_RUN_N 48
#endif
constexpr uint32_t XCD_RUN = XCD_RUN_N;           // consecutive candidates per XCD in warp_hist_kernel (1: plain round-robin)
constexpr int GQ = GQ_N;                          // 8x8 patch blocks (wave-gathers) in flight
static_assert(HWCOPIES * HWPITCH <= HCOPIES * HPITCH, "both layouts share the array");
// One pixel of the patch: the source position exactly as cv::warpPerspective forms it from the row terms (X0, Y0, W0) and the column terms
// (ax, bx, cx), nearest neighbour, BORDER_CONSTANT 0, cvRound = round-half-even. cv::warpPerspective forms Wd = 1/W (correctly rounded),
// fX = X*Wd, cvRound(fX). The full IEEE division is the most expensive part of the pixel, so the reciprocal is first taken from
// v_rcp_f64 + Newton steps; the rounded integers can only differ from the reference's when fX or fY lies within the reciprocal's error
// (times the coordinate) of a rounding boundary, and every pixel within 1e-6 of one takes the exact path (so does a NaN: the compares
// are false). FAST (the 56 x 56 patch of a frame below 4 GB): one Newton step (v_rcp_f64 is good to 2^-24 or so, one step squares that: 1e-11 of
// a pixel at coordinate 4000), no range test (|f| >= 2^31 saturates the conversion and lands outside the frame on both paths) and 32-bit offsets.
template <bool FAST>
__device__ __forceinline__ uint8_t warp_gather(const uint8_t* src, size_t row_stride, int W, int H, double X0, double Y0, double W0, double ax, double bx,
                                               double cx, bool inside) {
    const double Wq = W0 + cx, nx = X0 + ax, ny = Y0 + bx;
    double r = __builtin_amdgcn_rcp(Wq);
    r = __builtin_fma(__builtin_fma(-Wq, r, 1.0), r, r);
    if (!FAST) r = __builtin_fma(__builtin_fma(-Wq, r, 1.0), r, r);
    double fX = nx * r, fY = ny * r;
    bool sure = (int)(fabs(__builtin_amdgcn_fract(fX) - 0.5) > 1e-6) & (int)(fabs(__builtin_amdgcn_fract(fY) - 0.5) > 1e-6);   // no short circuit: two compares, one scalar and
    if (!FAST) sure = sure && fabs(fX) < 1e9 && fabs(fY) < 1e9;
    if (!sure) {
        const double Wd = Wq != 0 ? 1. / Wq : 0;
        fX = fmax((double)INT_MIN, fmin((double)INT_MAX, nx * Wd));
        fY = fmax((double)INT_MIN, fmin((double)INT_MAX, ny * Wd));
    }
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    uint8_t v = 0;
    if (FAST) {
        if ((int)((uint32_t)X < (uint32_t)W) & (int)((uint32_t)Y < (uint32_t)H)) v = src[(uint32_t)Y * (uint32_t)row_stride + (uint32_t)X];
    } else {
        if (inside && X >= 0 && X < W && Y >= 0 && Y < H) v = src[(size_t)Y * row_stride + X];
    }
    return v;
}

// WS = 56 (the reference's default markerWarpSize, src/markerdetector.cpp:246): a patch is 7 x 7 blocks, a step of the loop is one block row, so the
// column terms of the seven blocks are formed once per candidate and the row terms once per block row, and no pixel needs a bounds test (round 3:
// the stream is bound by vector-instruction issue, and these were a third of the kernel's). WS = 0: any size, everything per pixel.
template <int WS, int ROWS>
__device__ __forceinline__ void warp_hist_candidate(const DecodeArgs& a, const uint32_t idx, uint32_t* hist, const double* siM, const int lane) {
    const uint32_t e = a.cand_list[idx];
    const uint8_t* src = a.gray + (size_t)(e >> 16) * a.frame_stride;
    const int W = a.width, H = a.height;
    const int ws = WS ? WS : a.ws, npx = ws * ws;
    const bool bytes = HLANES * ((ws + 7) / 8) * ((ws + 7) / 8) < 256;   // pixels a copy can see: a byte counter must hold them
    uint8_t* patch = a.patches + (size_t)idx * npx;
    uint32_t* myhist = bytes ? hist + (lane / HLANES) * HPITCH : hist + (lane & (HWCOPIES - 1)) * HWPITCH;
    const double m0 = siM[0], m1 = siM[1], m2 = siM[2], m3 = siM[3], m4 = siM[4], m5 = siM[5], m6 = siM[6], m7 = siM[7], m8 = siM[8];
    // The 64 lanes of a gather take an 8x8 block of patch pixels (lane = 8 * row + column inside the block), GQ blocks in
    // flight. A gather costs about as much as the number of distinct 128-byte lines it touches: a patch ROW of a rotated
    // marker crosses 56 image rows, an 8x8 block of the patch covers about 21 x 21 source pixels whatever the rotation.
    const int bxl = lane & 7, byl = lane >> 3;
    uint32_t psum = 0;   // this lane's share of the sum of the patch's pixels = sum of i * hist[i]: the numerator of Otsu's mean, exact in integers
    if (WS == 8 * GQ) {
        double ax[GQ], bx[GQ], cx[GQ];
#pragma unroll
        for (int q = 0; q < GQ; q++) {
            const int x = q * 8 + bxl;
            ax[q] = m0 * x, bx[q] = m3 * x, cx[q] = m6 * x;
        }
        uint8_t* prow = patch + byl * WS + bxl;
        // ROWS block rows of gathers are issued before any of them is consumed: 1 for batches (registers = resident waves), 2 for one frame per call,
        // where a candidate's wave has the SIMD to itself and the kernel's time is the chain of its gather round trips (7 -> 4 of them)
        for (int BY0 = 0; BY0 < GQ; BY0 += ROWS) {
            uint8_t v[ROWS][GQ];
#pragma unroll
            for (int rr = 0; rr < ROWS; rr++) {
                const int BY = min(BY0 + rr, GQ - 1);   // the odd last round repeats the last row's addresses (their values are not used twice)
                const int y = BY * 8 + byl;
                const double X0 = m1 * y + m2, Y0 = m4 * y + m5, W0 = m7 * y + m8;
#pragma unroll
                for (int q = 0; q < GQ; q++) v[rr][q] = warp_gather<true>(src, a.row_stride, W, H, X0, Y0, W0, ax[q], bx[q], cx[q], true);
            }
#pragma unroll
            for (int rr = 0; rr < ROWS; rr++) {
                const int BY = BY0 + rr;
                if (BY < GQ) {
#pragma unroll
                    for (int q = 0; q < GQ; q++) {
                        prow[BY * 8 * WS + q * 8] = v[rr][q];
                        psum += v[rr][q];
                        atomicAdd(&myhist[v[rr][q] >> 2], 1u << (8 * (v[rr][q] & 3)));
                    }
                }
            }
        }
    } else {
        const int nb = (ws + 7) >> 3, nblocks = nb * nb;
        for (int b0 = 0; b0 < nblocks; b0 += GQ) {
            uint8_t v[GQ];
            int px[GQ], py[GQ];
#pragma unroll
            for (int q = 0; q < GQ; q++) {
                const int bi = b0 + q, BY = bi / nb, BX = bi - BY * nb;
                const int x = BX * 8 + bxl, y = BY * 8 + byl;
                px[q] = x, py[q] = (bi < nblocks && x < ws && y < ws) ? y : -1;
                v[q] = warp_gather<false>(src, a.row_stride, W, H, m1 * y + m2, m4 * y + m5, m7 * y + m8, m0 * x, m3 * x, m6 * x, py[q] >= 0);
            }
#pragma unroll
            for (int q = 0; q < GQ; q++) {
                if (py[q] >= 0) {
                    patch[py[q] * ws + px[q]] = v[q];
                    psum += v[q];
                    if (bytes)
                        atomicAdd(&myhist[v[q] >> 2], 1u << (8 * (v[q] & 3)));
                    else
                        atomicAdd(&myhist[v[q]], 1u);
                }
            }
        }
    }
    __syncthreads();
    // four bins per lane: the candidate's 512-byte histogram row is written as whole cache lines
    uint32_t hsum[4] = {0, 0, 0, 0};
    if (bytes) {
        uint32_t even = 0, odd = 0;   // bins 4*lane, 4*lane+2 and 4*lane+1, 4*lane+3 as halfword pairs
#pragma unroll
        for (int c = 0; c < HCOPIES; c++) {
            const uint32_t w = hist[c * HPITCH + lane];
            even += w & 0x00FF00FFu, odd += (w >> 8) & 0x00FF00FFu;
        }
        hsum[0] = even & 0xFFFFu, hsum[2] = even >> 16, hsum[1] = odd & 0xFFFFu, hsum[3] = odd >> 16;
    } else {
#pragma unroll
        for (int c = 0; c < HWCOPIES; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) hsum[q] += hist[c * HWPITCH + 4 * lane + q];
    }
    ((uint2*)(a.hist + (size_t)idx * 256))[lane] = make_uint2(hsum[0] | (hsum[1] << 16), hsum[2] | (hsum[3] << 16));
    // the pixel sum travels in the candidate's threshold slot: otsu_kernel reads it there and puts the threshold in its place
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) psum += (uint32_t)__shfl_xor((int)psum, o, 64);
    if (lane == 0) a.othr[idx] = (int32_t)psum;
}

#ifndef WARP_WAVES_N
#define WARP_WAVES_N 2   // waves per SIMD the register allocator aims at (170 VGPRs unconstrained = 2)
#endif
template <int ROWS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WARP_WAVES_N, WARP_WAVES_N))) void warp_hist_kernel(DecodeArgs a) {
    throughput_bound_priority();
    __shared__ uint32_t hist[HCOPIES * HPITCH];
    __shared__ double siM[9];
    const uint32_t n = min(a.counters[CNT_NCAND], a.cap_flat);
    const int lane = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs. The candidate list is ordered by frame, and the candidates of a marker (its border, the
    // outline of its quiet zone) sample the same pixels: XCD_RUN consecutive list entries go to one XCD, so that its L2 serves the second
    // candidate's lines instead of HBM.
    const uint32_t nslots = ((n + 8 * XCD_RUN - 1) / (8 * XCD_RUN)) * (8 * XCD_RUN);
    for (uint32_t b = blockIdx.x; b < nslots; b += gridDim.x) {
        uint32_t idx = b;
        if (XCD_RUN > 1) {
            const uint32_t xcd = b & 7u, r = b >> 3;
            idx = ((r / XCD_RUN) * 8u + xcd) * XCD_RUN + (r % XCD_RUN);
        }
        if (idx >= n) continue;
        __syncthreads();
        for (int i = lane; i < HCOPIES * HPITCH; i += WAVE) hist[i] = 0;
        if (lane < 9) siM[lane] = a.iM[(size_t)idx * 9 + lane];
        __syncthreads();
        if (a.ws == 8 * GQ && a.row_stride * (size_t)a.height < ((size_t)1 << 32))
            warp_hist_candidate<8 * GQ, ROWS>(a, idx, hist, siM, lane);
        else
            warp_hist_candidate<0, 1>(a, idx, hist, siM, lane);
    }
}

// 5c: one lane per candidate — getThreshVal_Otsu_8u, strictly sequential in double like the reference. Its first sweep (mu = sum of i * h[i],
// all integers, exact in double in any order) is the sum of the patch's pixels, which warp_hist_kernel left in the candidate's threshold slot, so
// only the sigma sweep remains. The histograms of the workgroup's 64 candidates are brought into LDS a quarter (64 bins) at a time with coalesced
// reads (bin-major, so that the lanes' sweeps read neighbouring halfwords): 8.4 KB instead of the 34 KB of all bins at once. Round 3: with the batches
// in flight a workgroup that needs a fifth of a CU's LDS waits for it, this kernel ran 0.41 ms in the stream against 0.06 alone (overlap trace).
constexpr int OTSU_PITCH = 66;   // halfwords per bin row: 64 candidates + padding against bank conflicts
constexpr int OTSU_BINS = 64;    // bins staged at a time
// Round 4 measured a sweep that skips runs of empty bins once mu1 has reached its fixed point (bit-identical, a third of the divisions): the lane's
// branches cost more than the divisions save - 93-106 us against 50 us for the 15 candidates of a 640x480 still, 0.43 against 0.41 ms per 1024-frame batch. Not kept.
__global__ __launch_bounds__(64) void otsu_kernel(DecodeArgs a) {
    latency_bound_priority();
    __shared__ uint16_t sh[OTSU_BINS * OTSU_PITCH];
    const uint32_t n = min(a.counters[CNT_NCAND], a.cap_flat);
    const uint32_t base = blockIdx.x * 64;
    if (base >= n) return;
    const int lane = threadIdx.x;
    const int cnt = (int)min(64u, n - base);
    const uint32_t idx = base + min(lane, cnt - 1);
    const int npx = a.ws * a.ws;
    const double scale = 1. / npx;
    const double mu = (double)(uint32_t)a.othr[idx] * scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    constexpr int DW = OTSU_BINS / 2;          // dwords of a candidate's row per stage
    constexpr int CPL = 64 / DW;               // candidates a load instruction covers
    for (int stage = 0; stage < 256 / OTSU_BINS; stage++) {
        __syncthreads();
        // 8 loads in flight per step: lane -> candidate c0 + CPL * j + lane / DW, dword stage * DW + lane % DW of its row
        for (int c0 = 0; c0 < cnt; c0 += 8 * CPL) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = min(c0 + CPL * j + lane / DW, cnt - 1);
                v[j] = ((const uint32_t*)(a.hist + (size_t)(base + c) * 256))[stage * DW + (lane % DW)];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = c0 + CPL * j + lane / DW;
                if (c < cnt) {
                    const int k = 2 * (lane % DW);
                    sh[k * OTSU_PITCH + c] = (uint16_t)(v[j] & 0xFFFFu);
                    sh[(k + 1) * OTSU_PITCH + c] = (uint16_t)(v[j] >> 16);
                }
            }
        }
        __syncthreads();
        if (lane < cnt) {
            const uint16_t* h = sh + lane;
            // Sixteen bins at a time, in three passes, so that only what MUST be sequential sits on the chain of dependent operations. The reference's
            // recurrence per bin is  mu1 = (mu1 * q1_old + i * p_i) / q1  (or mu1 * q1_old alone when a class is empty): a product, a sum and a
            // division, each rounded. q1 (a running sum of the p_i) and the skip test do not depend on mu1, so pass A forms them and the
            // denominator's part of the division ahead of the chain - the reciprocal of q1 refined by two Newton steps, exactly the first half of
            // the fp64 division sequence the compiler emits (rcp, fma, fma, fma, fma); pass B is the chain: product, sum, and the numerator's half
            // of that same sequence (q = a * r, e = fma(-q1, q, a), fma(e, r, q)) - the operands are far from the exponent range where the
            // hardware sequence rescales (q1 in [1e-7, 1], a in [0, 255]), so this IS the compiler's division, bit for bit; pass C (mu2, sigma,
            // the running maximum) follows off the chain. 5 dependent operations per bin instead of ~14.
            constexpr int G = 16;
#pragma unroll 1
            for (int g0 = 0; g0 < OTSU_BINS; g0 += G) {
                double q1v[G], rv[G], ipv[G], m1v[G];
                uint32_t skip = 0;
                const double q1_in = q1;
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const int i = stage * OTSU_BINS + g0 + j;
                    const double p_i = h[(g0 + j) * OTSU_PITCH] * scale;
                    q1 += p_i;
                    const double q2 = 1. - q1;
                    if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) skip |= 1u << j;
                    q1v[j] = q1, ipv[j] = i * p_i;
                    const double r0 = __builtin_amdgcn_rcp(q1);
                    const double r1 = __builtin_fma(r0, __builtin_fma(-q1, r0, 1.), r0);
                    rv[j] = __builtin_fma(r1, __builtin_fma(-q1, r1, 1.), r1);
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const double t = mu1 * (j ? q1v[j - 1] : q1_in);
                    const double a_ = t + ipv[j];
                    const double q = a_ * rv[j];
                    const double e = __builtin_fma(-q1v[j], q, a_);
                    const double d = __builtin_fma(e, rv[j], q);
                    mu1 = ((skip >> j) & 1u) ? t : d;
                    m1v[j] = mu1;
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if ((skip >> j) & 1u) continue;
                    const int i = stage * OTSU_BINS + g0 + j;
                    const double q2 = 1. - q1v[j];
                    const double mu2 = (mu - q1v[j] * m1v[j]) / q2;
                    const double sigma = q1v[j] * q2 * (m1v[j] - mu2) * (m1v[j] - mu2);
                    if (sigma > max_sigma) {
                        max_sigma = sigma;
                        max_val = i;
                    }
                }
            }
        }
    }
    if (lane < cnt) a.othr[idx] = (int)max_val;
}

// 5d: one wavefront per candidate — 7x7 cell votes on the binarised patch and the 5x5 Hamming decode (decode_device.h). Since round 3 the
// default pipeline runs it as the head of refine_lines_kernel; this kernel remains for callers that want the ids without the refinement.
__global__ __launch_bounds__(64) void cells_decode_kernel(DecodeArgs a) {
    latency_bound_priority();
    const uint32_t n = min(a.counters[CNT_NCAND], a.cap_flat);
    const int lane = threadIdx.x;
    for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
        const uint32_t e = a.cand_list[idx];
        Cand* cand = a.cands + (size_t)(e >> 16) * a.cap_cands + (e & 0xFFFFu);
        int id, nrot;
        cells_decode_wave(a.patches + (size_t)idx * a.ws * a.ws, a.ws, a.othr[idx], lane, &id, &nrot);
        if (lane == 0) cand->id = id, cand->nrot = nrot;
    }
}

// 5d' — highly reliable markers: HighlyReliableMarkers::detect, src/highlyreliablemarkers.cpp:332-383.
// One wavefront per candidate: lane = inner cell (n*n <= 64), majority vote on the Otsu-binarised patch, the code and its
// three rotations (MarkerCode::set :113-142) through ballots, then the lanes share the dictionary: nearest entry over the
// four rotations with the reference's first-minimum order (entries ascending, then rotations), accepted if its Hamming
// distance is at most the correction distance; an exact match is the distance-0 case (the reference finds it through
// 32-bit ids that are unique for n <= 5 and overflow beyond; the nearest-entry search is the intended behaviour for every n).
struct HrmArgs {
    int n, count;
    uint32_t correction;
    const uint64_t* codes;
};

__global__ __launch_bounds__(64) void hrm_decode_kernel(DecodeArgs a, HrmArgs d) {
    latency_bound_priority();
    const uint32_t ncand = min(a.counters[CNT_NCAND], a.cap_flat);
    const int lane = threadIdx.x, n = d.n, nn = n * n;
    for (uint32_t idx = blockIdx.x; idx < ncand; idx += gridDim.x) {
        const uint32_t e = a.cand_list[idx];
        Cand* cand = a.cands + (size_t)(e >> 16) * a.cap_cands + (e & 0xFFFFu);
        const int ws = a.ws, cell = ws / (n + 2), thr = a.othr[idx];
        const uint8_t* patch = a.patches + (size_t)idx * ws * ws;
        const int y = lane / n, x = lane - y * n;
        bool white = false;
        if (lane < nn) {   // getMarkerCode: inner cell (y, x) is white iff more than half of its pixels exceed the Otsu threshold
            int cnt = 0;
            for (int py = 0; py < cell; py++)
                for (int px = 0; px < cell; px++) cnt += patch[((y + 1) * cell + py) * ws + (x + 1) * cell + px] > thr;
            white = cnt > (cell * cell) / 2;
        }
        // rotation r puts cell (y, x) at (ry, rx): r=1 (x, n-y-1), r=2 (n-y-1, n-x-1), r=3 (n-x-1, y); a ballot needs the
        // value at the destination lane, so every lane fetches the vote of its source cell
        unsigned long long rot[4];
        rot[0] = __ballot(white);
#pragma unroll
        for (int r = 1; r < 4; r++) {
            // destination (y, x) <- source (sy, sx): invert the mapping above
            const int sy = r == 1 ? n - x - 1 : r == 2 ? n - y - 1 : x;
            const int sx = r == 1 ? y : r == 2 ? n - x - 1 : n - y - 1;
            const int srcl = lane < nn ? sy * n + sx : lane;
            const int v = __shfl((int)white, srcl, 64);
            rot[r] = __ballot(lane < nn && v != 0);
        }
        // nearest dictionary entry: key = distance << 16 | entry << 2 | rotation, smallest key wins (= first minimum)
        uint32_t best = 0xFFFFFFFFu;
        for (int i = lane; i < d.count; i += WAVE) {
            const unsigned long long c = d.codes[i];
            uint32_t dm = (uint32_t)nn, rm = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t hd = (uint32_t)__popcll(c ^ rot[r]);
                if (hd < dm) dm = hd, rm = (uint32_t)r;
            }
            if (dm < (uint32_t)nn) best = min(best, (dm << 16) | ((uint32_t)i << 2) | rm);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
        if (lane == 0) {
            int id = -1, nrot = 0;
            if (best != 0xFFFFFFFFu && (best >> 16) <= d.correction) id = (int)((best >> 2) & 0x3FFFu), nrot = (int)(best & 3u);
            cand->id = id;
            cand->nrot = nrot;
        }
    }
}

void launch_decode(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, bool fused_cells) {
    DecodeArgs a;
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride, a.width = g.width, a.height = g.height;
    a.ws = p.warp_size, a.cands = b.cands, a.cap_cands = b.cap_cands;
    a.cand_list = b.cand_list, a.counters = b.counters, a.cap_flat = b.cap_flat, a.iM = b.iM, a.hist = b.hist, a.othr = b.othr, a.patches = b.patches;
    const int lane_blocks = (int)((b.cap_flat + 63) / 64);
    const int wave_blocks = (int)std::min<uint32_t>(b.cap_flat, (uint32_t)nframes * 48u);
    // the inverse homographies are there already: frame_candidates_kernel solved them (b.iM)
    // one frame per call: the XCD-run unpacking sends list entries 0..47 to the slots of XCD 0 only - with 48 workgroups six of them took all the
    // candidates, eight one after the other (47 us for a 1080p frame); a workgroup per slot of the first run instead
    const int warp_blocks = nframes <= 2 ? (int)std::min<uint32_t>(b.cap_flat, 8u * XCD_RUN) : wave_blocks;
    if (nframes <= 2)
        hipLaunchKernelGGL(warp_hist_kernel<2>, dim3(warp_blocks), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(warp_hist_kernel<1>, dim3(warp_blocks), dim3(64), 0, s, a);
    if (p.decoder == ARUCOHIP_DECODER_USER) return;   // the host callback decodes the patches (capi.hip: user_decode_stage)
    hipLaunchKernelGGL(otsu_kernel, dim3(lane_blocks), dim3(64), 0, s, a);
    if (p.decoder == 1) {
        HrmArgs d{p.hrm_n, p.hrm_count, p.hrm_correction, p.hrm_codes};
        hipLaunchKernelGGL(hrm_decode_kernel, dim3(wave_blocks), dim3(64), 0, s, a, d);
    } else if (!fused_cells) {
        hipLaunchKernelGGL(cells_decode_kernel, dim3(wave_blocks), dim3(64), 0, s, a);
    }   // else: refine_lines_kernel decodes the cells of a candidate before it refines it (launch_refine_lines)
}

// ids / rotations a host decoder returned for the first n entries of the flat candidate list
__global__ void set_decoded_kernel(Cand* cands, int cap_cands, const uint32_t* cand_list, uint32_t n, const int2* dec) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = cand_list[i];
    Cand* c = cands + (size_t)(e >> 16) * cap_cands + (e & 0xFFFFu);
    c->id = dec[i].x, c->nrot = dec[i].y;
}
void launch_set_decoded(hipStream_t s, const Buffers& b, uint32_t n, const int2* dec) {
    if (n) hipLaunchKernelGGL(set_decoded_kernel, dim3((n + 255) / 256), dim3(256), 0, s, b.cands, b.cap_cands, b.cand_list, n, dec);
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
