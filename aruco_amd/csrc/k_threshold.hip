// Kernel 1 — adaptive threshold + neighbour masks + border-start candidates, one pass over the gray frame.
//
// Reference: MarkerDetector::thresHold -> cv::adaptiveThreshold(MEAN_C, BINARY_INV, b, C)
//            (/root/reference/src/markerdetector.cpp:643-677) and the raster scan of cv::findContours (:511).
// Per 64x32 output tile a 256-thread workgroup stages the gray tile (+halo) in LDS, forms the separable integer
// box sum, rounds the mean exactly like the u8 box filter, thresholds, and from the binary tile (1-px frame zeroed,
// as findContours does) derives for every pixel the 8-neighbour occupancy byte and the two local start rules:
//   outer border start : first pixel of a horizontal run of set pixels with no set pixel 8-adjacent in the row above
//                        (the raster-first pixel of an 8-connected component is such a pixel)
//   hole  border start : first pixel of a horizontal run of clear pixels whose left neighbour is set and which has no
//                        clear pixel directly above (the raster-first pixel of a 4-connected background hole is such)
// Runs are followed inside the LDS tile only (a run leaving the tile keeps its candidate). Candidates are verified by
// the walkers (k_contours.hip). Each workgroup collects its candidates in LDS and reserves list space with one atomic
// on its plane's counter. HBM traffic: read W*H, write 2*W*H (+ sparse list).
#include "internal.h"

namespace ah {

constexpr int TW = 64, TH = 32, NT = 256;

enum ThrMode { MODE_ADPT = 0, MODE_FIXED = 1, MODE_BINARY = 2 };

struct ThrArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    int nthr, t;          // planes per frame, plane handled by this launch
    int R;                // box radius
    int idelta;           // ADPT: floor(C); FIXED: floor(threshold)
    uint32_t magic;       // ceil(2^28 / n)
    int n_half;           // n / 2
    uint8_t* thres;
    uint8_t* nbr;
    uint2* trig;
    uint32_t* trig_cnt;
    uint32_t* counters;
    uint32_t cap_trig;
};

constexpr int LOCAL_TRIG = 192;   // candidates a workgroup can stage in LDS; more go straight to the global list

template <int RT, int MODE>
__global__ __launch_bounds__(NT) void threshold_kernel(ThrArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int R = (RT >= 0) ? RT : a.R;
    const int GW = TW + 2 * R + 2;              // gray tile width
    const int GP = (GW + 3) & ~3;               // pitch
    const int GH = TH + 2 * R + 2;
    const int HP = TW + 2;                      // hsum pitch (u16)
    uint8_t* g = lds;
    uint16_t* hs = (uint16_t*)(lds + ((GP * GH + 15) & ~15));
    uint8_t* bn = (uint8_t*)(hs + ((HP * GH + 7) & ~7));   // (TH+2) x (TW+2), pitch BP
    const int BP = TW + 4;
    __shared__ uint32_t s_trig[LOCAL_TRIG];
    __shared__ uint32_t s_ntrig, s_base;
    if (threadIdx.x == 0) s_ntrig = 0;

    const int tid = threadIdx.x;
    const int frame = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int W = a.width, H = a.height;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int gx0 = x0 - R - 1, gy0 = y0 - R - 1;

    // ---- stage gray tile (BORDER_REPLICATE by clamping)
    const bool interior = gx0 >= 0 && gx0 + GP <= W && ((a.row_stride | (size_t)gx0 | (size_t)src) & 3) == 0;
    if (interior) {
        const int dw = GP / 4;
        for (int i = tid; i < GH * dw; i += NT) {
            int r = i / dw, c = i - r * dw;
            int y = min(max(gy0 + r, 0), H - 1);
            uint32_t v = *(const uint32_t*)(src + (size_t)y * a.row_stride + gx0 + c * 4);
            *(uint32_t*)(g + r * GP + c * 4) = v;
        }
    } else {
        for (int i = tid; i < GH * GW; i += NT) {
            int r = i / GW, c = i - r * GW;
            int y = min(max(gy0 + r, 0), H - 1), x = min(max(gx0 + c, 0), W - 1);
            g[r * GP + c] = src[(size_t)y * a.row_stride + x];
        }
    }
    __syncthreads();

    // ---- horizontal box sums for columns x0-1 .. x0+TW
    if (MODE == MODE_ADPT) {
        for (int i = tid; i < GH * HP; i += NT) {
            int r = i / HP, c = i - r * HP;
            const uint8_t* p = g + r * GP + c;
            int s = 0;
            if (RT >= 0) {
#pragma unroll
                for (int k = 0; k <= 2 * RT; k++) s += p[k];
            } else {
                for (int k = 0; k <= 2 * R; k++) s += p[k];
            }
            hs[r * HP + c] = (uint16_t)s;
        }
        __syncthreads();
    }

    // ---- vertical sums, mean, threshold for rows y0-1 .. y0+TH, cols x0-1 .. x0+TW
    for (int i = tid; i < (TH + 2) * HP; i += NT) {
        int r = i / HP, c = i - r * HP;
        int v = g[(r + R) * GP + c + R];
        int thr;
        if (MODE == MODE_ADPT) {
            int s = 0;
            if (RT >= 0) {
#pragma unroll
                for (int k = 0; k <= 2 * RT; k++) s += hs[(r + k) * HP + c];
            } else {
                for (int k = 0; k <= 2 * R; k++) s += hs[(r + k) * HP + c];
            }
            int mean = (int)(((uint64_t)(uint32_t)(s + a.n_half) * a.magic) >> 28);
            thr = (v + a.idelta <= mean);
        } else if (MODE == MODE_FIXED) {
            thr = !(v > a.idelta);
        } else {
            thr = v != 0;
        }
        int x = x0 - 1 + c, y = y0 - 1 + r;
        int inside = (x >= 1) & (x <= W - 2) & (y >= 1) & (y <= H - 2);
        bn[r * BP + c] = (uint8_t)((thr & inside) | (thr << 1));
    }
    __syncthreads();

    // ---- outputs: 4 pixels per thread-iteration
    const int plane = frame * a.nthr + a.t;
    uint8_t* tdst = a.thres + (size_t)plane * W * H;
    uint8_t* ndst = a.nbr + (size_t)plane * W * H;
    const bool can_dword = (W & 3) == 0;
    for (int d = tid; d < TH * (TW / 4); d += NT) {
        int r = d / (TW / 4), c4 = (d - r * (TW / 4)) * 4;
        int y = y0 + r;
        if (y >= H) continue;
        uint32_t tpack = 0, npack = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int c = c4 + q, x = x0 + c;
            const uint8_t* p = bn + (r + 1) * BP + (c + 1);
            int e = p[1] & 1, ne = p[-BP + 1] & 1, n = p[-BP] & 1, nw = p[-BP - 1] & 1;
            int w = p[-1] & 1, sw = p[BP - 1] & 1, s = p[BP] & 1, se = p[BP + 1] & 1;
            int self = p[0];
            uint32_t m = e | (ne << 1) | (n << 2) | (nw << 3) | (w << 4) | (sw << 5) | (s << 6) | (se << 7);
            tpack |= ((self >> 1) ? 255u : 0u) << (8 * q);
            npack |= m << (8 * q);
            if (x < W) {
                int b = self & 1;
                int inside = (x >= 1) & (x <= W - 2) & (y >= 1) & (y <= H - 2);
                int outer = b & !(w | nw | n | ne);
                int hole = (!b) & inside & w & n;
                if (outer) {
                    // follow the run to the right: any set pixel 8-adjacent in the row above joins an earlier pixel
                    for (int j = 1; c + 1 + j + 1 <= TW + 1; j++) {
                        if (!(p[j] & 1)) break;
                        if (p[-BP + j + 1] & 1) {
                            outer = 0;
                            break;
                        }
                    }
                } else if (hole) {
                    // follow the background run: a clear pixel directly above joins an earlier background pixel
                    for (int j = 1; c + 1 + j <= TW + 1; j++) {
                        if (p[j] & 1) break;
                        if (!(p[-BP + j] & 1)) {
                            hole = 0;
                            break;
                        }
                    }
                }
                if (outer | hole) {
                    uint32_t e = ((uint32_t)hole << 31) | ((uint32_t)y << 16) | (uint32_t)x;
                    uint32_t ls = atomicAdd(&s_ntrig, 1u);
                    if (ls < LOCAL_TRIG) {
                        s_trig[ls] = e;
                    } else {
                        uint32_t slot = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE], 1u);
                        if (slot < a.cap_trig)
                            a.trig[(size_t)plane * a.cap_trig + slot] = make_uint2(e >> 31, e & 0x7FFFFFFFu);
                        else
                            atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_TRIG_OVERFLOW);
                    }
                }
            }
        }
        int x = x0 + c4;
        size_t off = (size_t)y * W + x;
        if (can_dword && x + 3 < W) {
            if (MODE != MODE_BINARY) *(uint32_t*)(tdst + off) = tpack;
            *(uint32_t*)(ndst + off) = npack;
        } else {
            for (int q = 0; q < 4 && x + q < W; q++) {
                if (MODE != MODE_BINARY) tdst[off + q] = (uint8_t)(tpack >> (8 * q));
                ndst[off + q] = (uint8_t)(npack >> (8 * q));
            }
        }
    }
    // ---- flush the staged start candidates: one atomic per workgroup on the plane's own counter
    __syncthreads();
    const uint32_t nl = min(s_ntrig, (uint32_t)LOCAL_TRIG);
    if (nl == 0) return;
    if (tid == 0) s_base = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE], nl);
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = tid; i < nl; i += NT) {
        uint32_t e = s_trig[i];
        if (base + i < a.cap_trig)
            a.trig[(size_t)plane * a.cap_trig + base + i] = make_uint2(e >> 31, e & 0x7FFFFFFFu);
        else
            atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_TRIG_OVERFLOW);
    }
}

static size_t lds_bytes(int R) {
    int GW = TW + 2 * R + 2, GP = (GW + 3) & ~3, GH = TH + 2 * R + 2, HP = TW + 2;
    size_t a = (size_t)((GP * GH + 15) & ~15);
    size_t b = (size_t)((HP * GH + 7) & ~7) * 2;
    size_t c = (size_t)(TH + 2) * (TW + 4);
    return a + b + c;
}

template <int MODE>
static void launch_mode(hipStream_t s, ThrArgs& a, dim3 grid) {
    size_t sh = lds_bytes(a.R);
    if (MODE == MODE_ADPT && a.R == 3)
        hipLaunchKernelGGL((threshold_kernel<3, MODE>), grid, dim3(NT), sh, s, a);
    else if (MODE != MODE_ADPT)
        hipLaunchKernelGGL((threshold_kernel<0, MODE>), grid, dim3(NT), sh, s, a);
    else
        hipLaunchKernelGGL((threshold_kernel<-1, MODE>), grid, dim3(NT), sh, s, a);
}

void launch_threshold(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    dim3 grid((g.width + TW - 1) / TW, (g.height + TH - 1) / TH, nframes);
    for (int t = 0; t < p.nthr; t++) {
        ThrArgs a;
        a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride;
        a.width = g.width, a.height = g.height, a.nthr = p.nthr, a.t = t;
        a.thres = b.thres, a.nbr = b.nbr, a.trig = b.trig, a.trig_cnt = b.trig_cnt, a.counters = b.counters, a.cap_trig = b.cap_trig;
        if (p.thres_method == ARUCOHIP_THRES_FIXED) {
            a.R = 0, a.idelta = (int)floor(p.p1[t]), a.magic = 0, a.n_half = 0;
            launch_mode<MODE_FIXED>(s, a, grid);
        } else {
            int n = p.block[t] * p.block[t];
            a.R = p.block[t] / 2, a.idelta = p.idelta, a.n_half = n / 2;
            a.magic = (uint32_t)(((1ull << 28) + n - 1) / n);
            launch_mode<MODE_ADPT>(s, a, grid);
        }
    }
}

// detectRectangles on a caller-supplied thresholded image (markerdetector.h:261): only masks + candidates.
void launch_binary_planes(hipStream_t s, const uint8_t* thres_in, const FrameGeom& g, int nframes, const Buffers& b) {
    dim3 grid((g.width + TW - 1) / TW, (g.height + TH - 1) / TH, nframes);
    ThrArgs a;
    a.gray = thres_in, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride;
    a.width = g.width, a.height = g.height, a.nthr = 1, a.t = 0;
    a.thres = b.thres, a.nbr = b.nbr, a.trig = b.trig, a.trig_cnt = b.trig_cnt, a.counters = b.counters, a.cap_trig = b.cap_trig;
    a.R = 0, a.idelta = 0, a.magic = 0, a.n_half = 0;
    launch_mode<MODE_BINARY>(s, a, grid);
}

}  // namespace ah
