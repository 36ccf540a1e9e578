// Kernel 6 — corner refinement of decoded candidates, one wavefront per candidate (LINES) or per corner (SUBPIX/HARRIS).
//
// Reference:
//   LINES  : MarkerDetector::refineCandidateLines (/root/reference/src/markerdetector.cpp:931-997) with
//            interpolate2Dline :83-130, getCrossPoint :132-139, distortPoints :141-153; applied before the corner
//            rotation std::rotate(begin, begin + 4 - nRotations, end) :364-366.
//   SUBPIX : cv::cornerSubPix(grey, corners, Size(p1,p1), Size(-1,-1), {MAX_ITER|EPS, 8, 0.005}) :402-405
//   HARRIS : SubPixelCorner::RefineCorner (/root/reference/src/subpixelcorner.cpp:70-189), one iteration, quirks kept.
// The reference fits each side with a float32 SVD least squares; here the 2x2 normal equations are accumulated in
// double across the 64 lanes (deviation from the goldens 2.5e-4 px).
#include <float.h>

#include "internal.h"
#include "decode_device.h"
#include "pnp_device.h"

namespace ah {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_maxi(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

struct LinesArgs {
    Cand* cands;
    const int32_t* ncands;
    const ContourDesc* cdesc;
    const short2* pool;
    int cap_cands;
    int do_lines;
    CamModel cam;
    const uint32_t* cand_list;   // frame << 16 | index, counters[CNT_NCAND] entries
    const uint32_t* counters;
    uint32_t cap_flat;
    // round 3: the built-in 5x5 decoder as the head of this kernel (decode_device.h: cells_decode_wave)
    int fused_cells, ws;
    const uint8_t* patches;      // [cap_flat][ws * ws]
    const int32_t* othr;         // [cap_flat] Otsu thresholds
};

struct SideSums {
    double n, sx, sy, sxx, syy, sxy;
    float minx, maxx, miny, maxy;
    float x0, y0;
};

__device__ __forceinline__ void contour_point(const LinesArgs& a, const short2* C, int n, int swapped, int j, bool undist, float* px, float* py) {
    short2 p = C[swapped ? n - 1 - j : j];
    float x = (float)p.x, y = (float)p.y;
    if (undist) {
        double ux, uy;
        undistort_point((double)x, (double)y, a.cam.K, a.cam.k, &ux, &uy);
        const float* P = a.cam.K;
        double xx = (double)P[0] * ux + (double)P[1] * uy + (double)P[2];
        double yy = (double)P[3] * ux + (double)P[4] * uy + (double)P[5];
        double ww = 1. / ((double)P[6] * ux + (double)P[7] * uy + (double)P[8]);
        x = (float)(xx * ww), y = (float)(yy * ww);
    }
    *px = x, *py = y;
}

// least-squares line through the accumulated side; same branches as interpolate2Dline
__device__ static void fit_line(const SideSums& s, float line[3]) {
    bool yx = (s.maxx - s.minx > s.maxy - s.miny);
    double su = yx ? s.sx : s.sy, sv = yx ? s.sy : s.sx, suu = yx ? s.sxx : s.syy, suv = s.sxy;
    double u0 = yx ? s.x0 : s.y0, v0 = yx ? s.y0 : s.x0;
    double det = s.n * suu - su * su;
    double A, C;
    if (fabs(det) > 0) {
        A = (s.n * suv - su * sv) / det;
        C = (sv - A * su) / s.n;
    } else {
        A = 0;
        C = sv / s.n;
    }
    C = C + v0 - A * u0;
    if (yx)
        line[0] = (float)A, line[1] = -1.f, line[2] = (float)C;
    else
        line[0] = -1.f, line[1] = (float)A, line[2] = (float)C;
}

__device__ __forceinline__ void refine_one(const LinesArgs& a, uint32_t e, int lane, uint32_t li) {
    const int frame = (int)(e >> 16), ci = (int)(e & 0xFFFFu);
    Cand* cand = a.cands + (size_t)frame * a.cap_cands + ci;
    int nrot_fused = 0;
    if (a.fused_cells) {
        // FiducidalMarkers::detect on this candidate's Otsu-thresholded patch; lane 0 knows the result and publishes it
        int id, nrot;
        cells_decode_wave(a.patches + (size_t)li * a.ws * a.ws, a.ws, a.othr[li], lane, &id, &nrot);
        if (lane == 0) cand->id = id, cand->nrot = nrot;
        id = __builtin_amdgcn_readfirstlane(id);
        nrot_fused = __builtin_amdgcn_readfirstlane(nrot);   // the other lanes take it from the register, not from the store above
        if (id < 0) return;
    } else if (cand->id < 0) {
        return;
    }
    float out[8];
    for (int k = 0; k < 8; k++) out[k] = cand->c[k];
    if (a.do_lines) {
        const ContourDesc cd = a.cdesc[cand->cdesc];
        const int n = cd.n, swapped = cand->swapped;
        const short2* C = a.pool + cd.pool_off;
        const bool undist = a.cam.has_K && a.cam.has_dist;
        // corner positions in the contour (last match wins)
        int cidx[4] = {-1, -1, -1, -1};
        for (int j = lane; j < n; j += WAVE) {
            short2 p = C[swapped ? n - 1 - j : j];
            for (int k = 0; k < 4; k++)
                if (p.x == cand->qx[k] && p.y == cand->qy[k]) cidx[k] = j;
        }
        for (int k = 0; k < 4; k++) cidx[k] = max(wave_maxi(cidx[k]), 0);
        bool inverse;
        if ((cidx[1] > cidx[0]) && (cidx[2] > cidx[1] || cidx[2] < cidx[0]))
            inverse = false;
        else if (cidx[2] > cidx[1] && cidx[2] < cidx[0])
            inverse = false;
        else
            inverse = true;
        float lines[4][3];
        for (int l = 0; l < 4; l++) {
            const int start = cidx[l], end = cidx[(l + 1) & 3];
            SideSums s;
            s.n = 0, s.sx = s.sy = s.sxx = s.syy = s.sxy = 0;
            s.minx = s.miny = FLT_MAX, s.maxx = s.maxy = -FLT_MAX;
            contour_point(a, C, n, swapped, start, undist, &s.x0, &s.y0);
            if (!inverse) {
                int cnt = end - start;
                if (cnt < 0) cnt += n;
                for (int q = lane; q < cnt; q += WAVE) {
                    int j = start + q;
                    if (j >= n) j -= n;
                    float x, y;
                    contour_point(a, C, n, swapped, j, undist, &x, &y);
                    double u = (double)x - (double)s.x0, v = (double)y - (double)s.y0;
                    s.sx += u, s.sy += v, s.sxx += u * u, s.syy += v * v, s.sxy += u * v;
                    s.minx = fminf(s.minx, x), s.maxx = fmaxf(s.maxx, x), s.miny = fminf(s.miny, y), s.maxy = fmaxf(s.maxy, y);
                }
                s.n = cnt;
                if (cnt == 1 && lane == 0) {  // :974-976 — a side with a single point also takes the next corner
                    float x, y;
                    contour_point(a, C, n, swapped, end, undist, &x, &y);
                    double u = (double)x - (double)s.x0, v = (double)y - (double)s.y0;
                    s.sx += u, s.sy += v, s.sxx += u * u, s.syy += v * v, s.sxy += u * v;
                    s.minx = fminf(s.minx, x), s.maxx = fmaxf(s.maxx, x), s.miny = fminf(s.miny, y), s.maxy = fmaxf(s.maxy, y);
                }
                if (cnt == 1) s.n = 2;
            } else if (lane == 0) {
                // backward walk with the reference's size_t modulo (:967): j = (uint64)(j - 1) % n, bounded like the oracle
                int j = start, guard = 0, cnt = 0;
                while (j != end && guard++ <= 2 * n) {
                    float x, y;
                    contour_point(a, C, n, swapped, j, undist, &x, &y);
                    double u = (double)x - (double)s.x0, v = (double)y - (double)s.y0;
                    s.sx += u, s.sy += v, s.sxx += u * u, s.syy += v * v, s.sxy += u * v;
                    s.minx = fminf(s.minx, x), s.maxx = fmaxf(s.maxx, x), s.miny = fminf(s.miny, y), s.maxy = fmaxf(s.maxy, y);
                    cnt++;
                    j = (int)((unsigned long long)(long long)(j - 1) % (unsigned long long)n);
                }
                if (cnt == 1) {
                    float x, y;
                    contour_point(a, C, n, swapped, end, undist, &x, &y);
                    double u = (double)x - (double)s.x0, v = (double)y - (double)s.y0;
                    s.sx += u, s.sy += v, s.sxx += u * u, s.syy += v * v, s.sxy += u * v;
                    s.minx = fminf(s.minx, x), s.maxx = fmaxf(s.maxx, x), s.miny = fminf(s.miny, y), s.maxy = fmaxf(s.maxy, y);
                    cnt = 2;
                }
                s.n = cnt;
            }
            s.sx = wave_sum(s.sx), s.sy = wave_sum(s.sy), s.sxx = wave_sum(s.sxx), s.syy = wave_sum(s.syy), s.sxy = wave_sum(s.sxy);
            s.minx = wave_minf(s.minx), s.maxx = wave_maxf(s.maxx), s.miny = wave_minf(s.miny), s.maxy = wave_maxf(s.maxy);
            if (inverse) s.n = __shfl(s.n, 0, 64);
            fit_line(s, lines[l]);
        }
        for (int i = 0; i < 4; i++) {  // getCrossPoint(lines[i], lines[i-1])
            const float* l1 = lines[i];
            const float* l2 = lines[(i + 3) & 3];
            double A = l1[0], B = l1[1], Cc = l2[0], D = l2[1], E = -l1[2], F = -l2[2];
            double det = A * D - B * Cc;
            float cx = (float)((E * D - B * F) / det), cy = (float)((A * F - E * Cc) / det);
            if (undist) {  // distortPoints: normalise with the float K, forward Brown model
                const float* K = a.cam.K;
                float X = (cx - K[2]) / K[0], Y = (cy - K[5]) / K[4];
                const double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0, 0, 0};
                double mx, my;
                project_point((double)X, (double)Y, 1.0, R, nullptr, t0, K, a.cam.k, &mx, &my, nullptr, nullptr);
                cx = (float)mx, cy = (float)my;
            }
            out[2 * i] = cx, out[2 * i + 1] = cy;
        }
    }
    // canonical corner order: std::rotate(begin, begin + 4 - nRotations, end)
    if (lane == 0) {
        const int nrot = a.fused_cells ? nrot_fused : cand->nrot;
        float r[8];
        for (int i = 0; i < 4; i++) {
            int srci = (i + 4 - nrot) & 3;
            r[2 * i] = out[2 * srci], r[2 * i + 1] = out[2 * srci + 1];
        }
        for (int k = 0; k < 8; k++) cand->c[k] = r[k];
    }
}

// one wave per decoded candidate, taken from the flat candidate list of the batch (a grid over every candidate slot of
// every frame would be 90 % empty workgroups)
#ifndef LINES_WAVES_N
#define LINES_WAVES_N 4   // waves per SIMD the register allocator is held to: 127 VGPRs without a spill instead of 129 = four resident waves instead of three
                          // (round 4: 0.183 -> 0.153 ms alone, +1.8 % frames/s; the same for warp_hist - 170 -> 168 VGPRs, three waves - spills and gains nothing)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LINES_WAVES_N, LINES_WAVES_N))) void refine_lines_kernel(LinesArgs a) {
    latency_bound_priority();
    const int lane = threadIdx.x;
    const uint32_t nlist = min(a.counters[CNT_NCAND], a.cap_flat);
    for (uint32_t li = blockIdx.x; li < nlist; li += gridDim.x) refine_one(a, a.cand_list[li], lane, li);
}

void launch_refine_lines(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const CamModel& cam, const Buffers& b, bool fused_cells) {
    LinesArgs a;
    a.fused_cells = fused_cells ? 1 : 0, a.ws = p.warp_size, a.patches = b.patches, a.othr = b.othr;
    a.cands = b.cands, a.ncands = b.ncands, a.cdesc = b.cdesc, a.pool = b.pool, a.cap_cands = b.cap_cands;
    a.do_lines = p.corner_method == ARUCOHIP_CORNER_LINES;
    a.cam = cam;
    a.cand_list = b.cand_list, a.counters = b.counters, a.cap_flat = b.cap_flat;
    const int blocks = (int)std::min<uint32_t>(b.cap_flat, (uint32_t)nframes * 48u);
    hipLaunchKernelGGL(refine_lines_kernel, dim3(blocks), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// SUBPIX / HARRIS: one wavefront per corner of a decoded candidate
// ---------------------------------------------------------------------------------------------
struct PixArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    Cand* cands;
    const int32_t* ncands;
    int cap_cands, method, win;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(64) void refine_pixels_kernel(PixArgs a) {
    latency_bound_priority();
    __shared__ float buf[33 * 33];
    __shared__ uint8_t loc[17 * 17];
    const int frame = blockIdx.z, ci = blockIdx.y, corner = blockIdx.x, lane = threadIdx.x;
    if (ci >= a.ncands[frame]) return;
    Cand* cand = a.cands + (size_t)frame * a.cap_cands + ci;
    if (cand->id < 0) return;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int W = a.width, H = a.height;
    const size_t st = a.row_stride;
    const float cTx = cand->c[2 * corner], cTy = cand->c[2 * corner + 1];
    float rx = cTx, ry = cTy;
    if (a.method == ARUCOHIP_CORNER_SUBPIX) {
        const int win = a.win, ww = 2 * win + 1, pw = ww + 2;
        float cIx = cTx, cIy = cTy;
        int iter = 0;
        double err = 0;
        const double eps = 0.005 * 0.005;
        do {
            // getRectSubPix 8u -> 32f, (ww+2)^2 patch around cI
            float ox = cIx - (pw - 1) * 0.5f, oy = cIy - (pw - 1) * 0.5f;
            int ix = (int)floorf(ox), iy = (int)floorf(oy);
            float fa = ox - ix, fb = oy - iy;
            float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
            __syncthreads();
            for (int i = lane; i < pw * pw; i += WAVE) {
                int r = i / pw, c = i - r * pw;
                int y0 = clampi(iy + r, 0, H - 1), y1 = clampi(iy + r + 1, 0, H - 1);
                int x0 = clampi(ix + c, 0, W - 1), x1 = clampi(ix + c + 1, 0, W - 1);
                float s0 = src[y0 * st + x0] * a11 + src[y0 * st + x1] * a12 + src[y1 * st + x0] * a21 + src[y1 * st + x1] * a22;
                buf[i] = s0;
            }
            __syncthreads();
            double A = 0, B = 0, C = 0, bb1 = 0, bb2 = 0;
            for (int k = lane; k < ww * ww; k += WAVE) {
                int i = k / ww, j = k - i * ww;
                float y = (float)(i - win) / win, x = (float)(j - win) / win;
                float vy = expf(-y * y);
                double m = (double)(float)(vy * expf(-x * x));
                const float* sp = buf + (i + 1) * pw + (j + 1);
                double tgx = (double)sp[1] - (double)sp[-1];
                double tgy = (double)sp[pw] - (double)sp[-pw];
                double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                double px = j - win, py = i - win;
                A += gxx, B += gxy, C += gyy;
                bb1 += gxx * px + gxy * py;
                bb2 += gxy * px + gyy * py;
            }
            A = wave_sum(A), B = wave_sum(B), C = wave_sum(C), bb1 = wave_sum(bb1), bb2 = wave_sum(bb2);
            double det = A * C - B * B;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            float nx = (float)(cIx + C * scale * bb1 - B * scale * bb2);
            float ny = (float)(cIy - B * scale * bb1 + A * scale * bb2);
            err = (nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy);
            cIx = nx, cIy = ny;
            if (cIx < 0 || cIx >= W || cIy < 0 || cIy >= H) break;
        } while (++iter < 8 && err > eps);
        if (fabsf(cIx - cTx) > win || fabsf(cIy - cTy) > win) cIx = cTx, cIy = cTy;
        rx = cIx, ry = cIy;
    } else {  // HARRIS (SubPixelCorner)
        const int win = 15, ps = 17;
        bool skip = cTx < 0 || cTy < 0 || cTy > H || cTy > W;
        if (!skip) {
            float ox = cTx - (ps - 1) * 0.5f, oy = cTy - (ps - 1) * 0.5f;
            int ix = (int)floorf(ox), iy = (int)floorf(oy);
            float fa = ox - ix, fb = oy - iy;
            int a11 = __float2int_rn((1.f - fa) * (1.f - fb) * 65536.f), a12 = __float2int_rn(fa * (1.f - fb) * 65536.f);
            int a21 = __float2int_rn((1.f - fa) * fb * 65536.f), a22 = __float2int_rn(fa * fb * 65536.f);
            for (int i = lane; i < ps * ps; i += WAVE) {
                int r = i / ps, c = i - r * ps;
                int y0 = clampi(iy + r, 0, H - 1), y1 = clampi(iy + r + 1, 0, H - 1);
                int x0 = clampi(ix + c, 0, W - 1), x1 = clampi(ix + c + 1, 0, W - 1);
                int s0 = src[y0 * st + x0] * a11 + src[y0 * st + x1] * a12 + src[y1 * st + x0] * a21 + src[y1 * st + x1] * a22;
                loc[i] = (uint8_t)((s0 + (1 << 15)) >> 16);
            }
            __syncthreads();
            const double coeff = 1. / (win * win);
            double A = 0, B = 0, C = 0, E = 0, F = 0;
            for (int k = lane; k < win * win; k += WAVE) {
                int i = k / win + 1, j = k - (k / win) * win + 1;   // rows/cols 1..15 of the 17x17 patch
                auto P = [&](int yy, int xx) { return (int)loc[yy * ps + xx]; };
                float gx = (float)((P(i - 1, j + 1) + 2 * P(i, j + 1) + P(i + 1, j + 1)) - (P(i - 1, j - 1) + 2 * P(i, j - 1) + P(i + 1, j - 1)));
                float gy = (float)((P(i + 1, j - 1) + 2 * P(i + 1, j) + P(i + 1, j + 1)) - (P(i - 1, j - 1) + 2 * P(i - 1, j) + P(i - 1, j + 1)));
                int ly = i - 8, lx = j - 8;
                float mxv = (float)exp(-(double)(lx * lx) * coeff), myv = (float)exp(-(double)(ly * ly) * coeff);
                double val = (double)(float)(mxv * myv);
                double dxx = (double)(gx * gx) * val, dyy = (double)(gy * gy) * val, dxy = (double)(gx * gy) * val;
                A += dxx, B += dxy, E += dyy;
                C += dxx * lx + dxy * ly;
                F += dxy * lx + dyy * ly;
            }
            A = wave_sum(A), B = wave_sum(B), C = wave_sum(C), E = wave_sum(E), F = wave_sum(F);
            double det = A * E - B * B;
            float ex = cTx, ey = cTy;
            if (fabs(det) > DBL_EPSILON * DBL_EPSILON) {
                det = 1.0 / det;
                ex = (float)(cTx + ((C * E) - (B * F)) * det);
                ey = (float)(cTy + (A * F) * det);   // the reference's (A*F - C*D) with D == 0
            }
            if (fabsf(cTx - ex) > win || fabsf(cTy - ey) > win) ex = cTx, ey = cTy;
            rx = ex, ry = ey;
        }
    }
    if (lane == 0) cand->c[2 * corner] = rx, cand->c[2 * corner + 1] = ry;
}

// ---------------------------------------------------------------------------------------------
// Locked-corner pre-pass: findCornerMaxima (/root/reference/src/markerdetector.cpp:157-199, enabled by :291-295, called
// at :398-399 before HARRIS / SUBPIX). One wavefront per corner of a decoded candidate: cv::cornerHarris(block 3, aperture 3,
// k 0.04) on the window of +-wsize pixels around the corner (the derivatives at the window's rim read the image around
// it; REFLECT_101 at the image border), cv::integral in double, every interior response replaced by the sum of the 4x4
// block that starts at it, then the arg max of the response weighted by 1 - (L1 distance to the window centre) / (half
// width + half height): first maximum in raster order, strictly positive, else (-1, -1) + window origin like the reference.
// ---------------------------------------------------------------------------------------------
constexpr int LOCK_MAXW = 62;   // window side: 2 * wsize, wsize <= 31

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

struct LockArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    Cand* cands;
    const int32_t* ncands;
    int cap_cands, wsize;
};

__global__ __launch_bounds__(64) void locked_corners_kernel(LockArgs a) {
    latency_bound_priority();
    // 61.5 KB: the response, then either the three derivative products or (once they are summed) the integral image
    constexpr int N = LOCK_MAXW * LOCK_MAXW;
    static_assert((LOCK_MAXW + 1) * (LOCK_MAXW + 1) * sizeof(double) <= 3 * N * sizeof(float), "the integral reuses the product arrays");
    __shared__ float harr[N];
    __shared__ __align__(8) float prod[3 * N];
    float *sxx = prod, *sxy = prod + N, *syy = prod + 2 * N;
    double* I = reinterpret_cast<double*>(prod);
    const int frame = blockIdx.z, ci = blockIdx.y, corner = blockIdx.x, lane = threadIdx.x;
    if (ci >= a.ncands[frame]) return;
    Cand* cand = a.cands + (size_t)frame * a.cap_cands + ci;
    if (cand->id < 0) return;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int W = a.width, H = a.height;
    const size_t st = a.row_stride;
    const float cx0 = cand->c[2 * corner], cy0 = cand->c[2 * corner + 1];
    const int x0 = max(0, (int)(cx0 - (float)a.wsize)), y0 = max(0, (int)(cy0 - (float)a.wsize));
    const int x1 = min(W, (int)(cx0 + (float)a.wsize)), y1 = min(H, (int)(cy0 + (float)a.wsize));
    const int rw = x1 - x0, rh = y1 - y0;
    if (rw <= 0 || rh <= 0 || rw > LOCK_MAXW || rh > LOCK_MAXW) {
        if (lane == 0) cand->c[2 * corner] = -1.f + (float)x0, cand->c[2 * corner + 1] = -1.f + (float)y0;
        return;
    }
    const float scale = (float)(1. / ((double)(1 << 2) * 3 * 255.));
    auto G = [&](int x, int y) -> float { return (float)src[(size_t)reflect101(y, H) * st + reflect101(x, W)]; };
    for (int i = lane; i < rw * rh; i += WAVE) {
        const int y = i / rw, x = i - y * rw, gx = x0 + x, gy = y0 + y;
        const float dxm = G(gx + 1, gy - 1) - G(gx - 1, gy - 1), dxc = G(gx + 1, gy) - G(gx - 1, gy), dxp = G(gx + 1, gy + 1) - G(gx - 1, gy + 1);
        const float dx = (dxm + dxp) * scale + dxc * (2.f * scale);
        const float sm = (G(gx - 1, gy - 1) + G(gx + 1, gy - 1)) * scale + G(gx, gy - 1) * (2.f * scale);
        const float sp = (G(gx - 1, gy + 1) + G(gx + 1, gy + 1)) * scale + G(gx, gy + 1) * (2.f * scale);
        const float dy = sp - sm;
        sxx[i] = dx * dx, sxy[i] = dx * dy, syy[i] = dy * dy;
    }
    __syncthreads();
    for (int i = lane; i < rw * rh; i += WAVE) {
        const int y = i / rw, x = i - y * rw;
        float A = 0.f, B = 0.f, C = 0.f;
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = reflect101(y + dy, rh);
            float ra = 0.f, rb = 0.f, rc = 0.f;
            for (int dx = -1; dx <= 1; dx++) {
                const int q = yy * rw + reflect101(x + dx, rw);
                ra += sxx[q], rb += sxy[q], rc += syy[q];
            }
            A += ra, B += rb, C += rc;
        }
        harr[i] = (float)((double)A * C - (double)B * B - 0.04 * ((double)A + C) * ((double)A + C));
    }
    __syncthreads();
    // cv::integral in double with the reference's addition order: running sum along each row, then down the columns
    const int iw = rw + 1;
    for (int i = lane; i < iw * (rh + 1); i += WAVE) I[i] = 0.0;
    __syncthreads();
    for (int y = lane; y < rh; y += WAVE) {          // lane = row: I[y+1][x+1] temporarily holds the row's running sum
        double row = 0;
        for (int x = 0; x < rw; x++) {
            row += (double)harr[y * rw + x];
            I[(y + 1) * iw + x + 1] = row;
        }
    }
    __syncthreads();
    for (int x = lane; x < rw; x += WAVE)            // lane = column: I[y+1][x+1] = I[y][x+1] + row sum
        for (int y = 0; y < rh; y++) I[(y + 1) * iw + x + 1] = I[y * iw + x + 1] + I[(y + 1) * iw + x + 1];
    __syncthreads();
    const int bls = 4;
    for (int i = lane; i < rw * rh; i += WAVE) {
        const int y = i / rw, x = i - y * rw;
        if (y >= bls && y < rh - bls && x >= bls && x < rw - bls)
            harr[i] = (float)(I[(y + bls) * iw + x + bls] - I[(y + bls) * iw + x] - I[y * iw + x + bls] + I[y * iw + x]);
    }
    __syncthreads();
    const float ccx = (float)(rw / 2), ccy = (float)(rh / 2), den = (float)(rw / 2 + rh / 2);
    double best = 0;
    int besti = 0x7FFFFFFF;
    for (int i = lane; i < rw * rh; i += WAVE) {
        const int y = i / rw, x = i - y * rw;
        const float d = (float)(fabsf(ccx - (float)x) + fabsf(ccy - (float)y)) / den;
        const float wgt = (float)(1. - (double)d);
        const double v = (double)(wgt * harr[i]);
        if (v > best) best = v, besti = i;          // ascending i per lane: the lane's first maximum
    }
    // wave: largest value, lowest raster index among equals (the sequential scan keeps the first)
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) best = ob, besti = oi;
    }
    if (lane == 0) {
        float bx = -1.f, by = -1.f;
        if (best > 0 && besti != 0x7FFFFFFF) by = (float)(besti / rw), bx = (float)(besti - (besti / rw) * rw);
        cand->c[2 * corner] = bx + (float)x0, cand->c[2 * corner + 1] = by + (float)y0;
    }
}

void launch_locked_corners(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    LockArgs a;
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride, a.width = g.width, a.height = g.height;
    a.cands = b.cands, a.ncands = b.ncands, a.cap_cands = b.cap_cands, a.wsize = p.locked_wsize;
    hipLaunchKernelGGL(locked_corners_kernel, dim3(4, b.cap_cands, nframes), dim3(64), 0, s, a);
}

void launch_refine_pixels(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    PixArgs a;
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride, a.width = g.width, a.height = g.height;
    a.cands = b.cands, a.ncands = b.ncands, a.cap_cands = b.cap_cands, a.method = p.corner_method, a.win = p.subpix_win;
    hipLaunchKernelGGL(refine_pixels_kernel, dim3(4, b.cap_cands, nframes), dim3(64), 0, s, a);
}

}  // namespace ah
