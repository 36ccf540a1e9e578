// Kernel 1 — adaptive threshold + bit-packed binary image + border-start candidates, one streaming pass per frame.
//
// Reference: MarkerDetector::thresHold -> cv::adaptiveThreshold(MEAN_C, BINARY_INV, b, C)
//            (/root/reference/src/markerdetector.cpp:643-677) and the raster scan of cv::findContours (:511).
//
// One wavefront owns a vertical strip: lane l holds 4 horizontally adjacent pixels (one dword of the gray row), the wave
// spans 256 px of which the middle 224 (lanes 4..59 = seven 32-px words) are outputs and 16 px per side are halo.
// The wave walks down its row segment keeping, in registers only,
//   * a ring of the last 2R+1 gray rows and of their horizontal box sums (two u16 per dword),
//   * the running vertical sums V (added row in, row out),
// so every gray byte is read from HBM/L2 once (plus the halo overlap), there is no LDS tile and no re-read.
// mean = round(S / b^2) is never formed: src - mean <= -C  <=>  (src + C) * b^2 <= S + b^2/2 (exact in integers).
// Per output row the wave stores 224 threshold bytes (one dword per lane); every 8 rows it stores 28 tiles (8x8 px, one
// uint64 each) of the binary image (1-px frame cleared, the image cv::findContours binarises) and stages the local border-start candidates
//   outer: pixel set,  W, NW, N, NE clear         hole: pixel clear, W and N set
// in LDS; one atomic per flush reserves space in the plane's raw candidate list. k_contours.hip filters (run rule)
// and verifies them. HBM traffic per frame: read W*H, write W*H + W*H/8 (+ sparse lists).
#include "bits_tiles.h"
#include "internal.h"

namespace ah {

enum ThrMode { MODE_ADPT = 0, MODE_FIXED = 1, MODE_BINARY = 2 };

constexpr int STRIP_OUT = 224;   // output pixels per strip (7 words)
constexpr int STRIP_HALO = 16;   // halo pixels per side (4 lanes) — covers box radius <= 15 plus the 1-px neighbourhood
constexpr int SEG = 128;         // output rows per wave
constexpr int LOCAL_TRIG = 256;

struct ThrArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    int nthr, t;          // planes per frame, plane handled by this launch
    int idelta;           // ADPT: floor(C); FIXED: floor(threshold)
    int n, n_half;        // b*b and b*b/2
    int tnx, tny;         // tiles per row / column of the tiled binary image
    uint8_t* thres;
    uint64_t* tiles;
    uint2* raw;           // raw candidate list of the plane
    uint32_t* raw_cnt;
    uint32_t* counters;
    uint32_t cap_raw;
    int seg_mode;         // 0: raw start candidates for the walkers, 1: waypoint cracks for the segment pipeline
    int grid_mask;        // waypoint grid spacing - 1 (spacing is a power of two)
};

__device__ __forceinline__ uint32_t byte_of(uint32_t v, int i) { return (v >> (8 * i)) & 0xFFu; }

// value of the neighbouring lane through the DPP wave shift (a VALU operand modifier on gfx9, no LDS crossbar trip):
// from_left = lane i receives lane i-1 (wave_shr:1), from_right = lane i receives lane i+1 (wave_shl:1); lane 0 / 63 get 0
__device__ __forceinline__ uint32_t from_left(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t from_right(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }

template <int R, int MODE>
__global__ __launch_bounds__(64) void threshold_strip_kernel(ThrArgs a) {
    constexpr int RING = 2 * R + 1;
    constexpr int NL = (R + 3) / 4;            // neighbour dwords needed on each side for the horizontal sums
    constexpr bool PACKV = RING * RING * 255 < 65536;
    __shared__ uint32_t s_trig[LOCAL_TRIG], s_trig2[LOCAL_TRIG];
    __shared__ uint32_t s_ntrig, s_base;
    const int lane = threadIdx.x;
    const int frame = blockIdx.z;
    const int W = a.width, H = a.height;
    const int x = (int)blockIdx.x * STRIP_OUT - STRIP_HALO + 4 * lane;   // first pixel of this lane
    const int ys = (int)blockIdx.y * SEG, ye = min(ys + SEG, H);
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int plane = frame * a.nthr + a.t;
    uint8_t* tdst = a.thres + (size_t)plane * W * H;
    uint64_t* bdst = a.tiles + (size_t)plane * a.tnx * a.tny;
    uint64_t tile_acc = 0;                                                     // 8 rows x 8 px of this even lane's tile
    if (lane == 0) s_ntrig = 0;
    __syncthreads();

    const bool whole = x >= 0 && x + 3 < W;                                   // all 4 pixels inside the image
    const bool aligned = ((a.row_stride | (size_t)src) & 3) == 0;             // dword loads allowed
    const bool out_lane = lane >= 4 && lane < 60 && x < W;
    const bool tile_lane = (lane & 1) == 0 && lane >= 4 && lane < 60 && x < W;   // x is a multiple of 8 on even lanes
    uint32_t insx = 0;                                                         // pixels with 1 <= x <= W-2
#pragma unroll
    for (int j = 0; j < 4; j++) insx |= (uint32_t)((x + j >= 1) && (x + j <= W - 2)) << j;
    uint32_t xsel = 0;                                                         // pixels on a waypoint grid column
#pragma unroll
    for (int j = 0; j < 4; j++) xsel |= (uint32_t)(((x + j) & a.grid_mask) == 0) << j;
    const int xc0 = min(max(x, 0), W - 1), xc1 = min(max(x + 1, 0), W - 1), xc2 = min(max(x + 2, 0), W - 1), xc3 = min(max(x + 3, 0), W - 1);

    auto load_row = [&](int r) -> uint32_t {
        const uint8_t* row = src + (size_t)min(max(r, 0), H - 1) * a.row_stride;   // BORDER_REPLICATE in y
        if (whole && aligned) return *(const uint32_t*)(row + x);
        return (uint32_t)row[xc0] | ((uint32_t)row[xc1] << 8) | ((uint32_t)row[xc2] << 16) | ((uint32_t)row[xc3] << 24);   // and in x
    };

    uint32_t G[RING], H01[RING], H23[RING];
#pragma unroll
    for (int k = 0; k < RING; k++) G[k] = 0, H01[k] = 0, H23[k] = 0;
    uint32_t V01 = 0, V23 = 0;          // packed u16 pairs (PACKV) ...
    uint32_t V0 = 0, V1 = 0, V2 = 0, V3 = 0;   // ... or four 32-bit sums
    uint32_t Eup = 0;                   // 6-bit pattern (left px, own 4, right px) of the binary row above

    const int r_begin = ys - 1 - R;     // first gray row; the first complete window is centred on row ys-1
    const int r_end = ye - 1 + R;       // last gray row
    for (int r0 = r_begin; r0 <= r_end; r0 += RING) {
#pragma unroll
        for (int k = 0; k < RING; k++) {
            const int r = r0 + k;
            if (r > r_end) break;
            const uint32_t D0 = load_row(r);
            uint32_t tbits = 0;         // raw threshold of the centre row, 4 bits
            const int c = r - R;        // centre row of the window that ends at r
            if (MODE == MODE_ADPT) {
                // ---- horizontal box sums of the 4 pixels
                uint32_t Dn[2 * NL + 1];
                Dn[NL] = D0;
#pragma unroll
                for (int q = 1; q <= NL; q++) {
                    Dn[NL - q] = from_left(Dn[NL - q + 1]);     // lane i - q
                    Dn[NL + q] = from_right(Dn[NL + q - 1]);    // lane i + q
                }
                auto px = [&](int i) -> uint32_t {   // byte i of the row relative to this lane's first pixel, i in [-4NL, 4NL+3]
                    const int q = i + 4 * NL;
                    return byte_of(Dn[q >> 2], q & 3);
                };
                uint32_t s0 = 0;
#pragma unroll
                for (int i = -R; i <= R; i++) s0 += px(i);
                const uint32_t s1 = s0 - px(-R) + px(R + 1);
                const uint32_t s2 = s1 - px(1 - R) + px(R + 2);
                const uint32_t s3 = s2 - px(2 - R) + px(R + 3);
                const uint32_t h01 = s0 | (s1 << 16), h23 = s2 | (s3 << 16);
                // ---- vertical running sums: row r enters, row r-RING leaves (it sits in the slot being overwritten)
                if (PACKV) {
                    V01 = V01 + h01 - H01[k];
                    V23 = V23 + h23 - H23[k];
                } else {
                    V0 += (h01 & 0xFFFFu) - (H01[k] & 0xFFFFu), V1 += (h01 >> 16) - (H01[k] >> 16);
                    V2 += (h23 & 0xFFFFu) - (H23[k] & 0xFFFFu), V3 += (h23 >> 16) - (H23[k] >> 16);
                }
                H01[k] = h01, H23[k] = h23, G[k] = D0;
                if (r < r_begin + 2 * R) continue;   // window not complete yet
                const uint32_t Gc = G[(k + RING - R) % RING];
                const int v0 = PACKV ? (int)(V01 & 0xFFFFu) : (int)V0, v1 = PACKV ? (int)(V01 >> 16) : (int)V1;
                const int v2 = PACKV ? (int)(V23 & 0xFFFFu) : (int)V2, v3 = PACKV ? (int)(V23 >> 16) : (int)V3;
                tbits = (uint32_t)(((int)byte_of(Gc, 0) + a.idelta) * a.n <= v0 + a.n_half) |
                        ((uint32_t)(((int)byte_of(Gc, 1) + a.idelta) * a.n <= v1 + a.n_half) << 1) |
                        ((uint32_t)(((int)byte_of(Gc, 2) + a.idelta) * a.n <= v2 + a.n_half) << 2) |
                        ((uint32_t)(((int)byte_of(Gc, 3) + a.idelta) * a.n <= v3 + a.n_half) << 3);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (int)byte_of(D0, j);
                    tbits |= (uint32_t)(MODE == MODE_FIXED ? !(v > a.idelta) : (v != 0)) << j;
                }
            }
            // ---- binary row for contour purposes: frame cleared
            const uint32_t B = (c >= 1 && c <= H - 2) ? (tbits & insx) : 0u;
            const uint32_t Bnext = from_right(B);
            const uint32_t left = (from_left(B) >> 3) & 1u, right = Bnext & 1u;
            const uint32_t Emid = left | (B << 1) | (right << 5);
            if (c >= ys) {   // c < ye by construction
                if (MODE != MODE_BINARY && out_lane) {
                    const uint32_t t4 = ((tbits * 0x00204081u) & 0x01010101u) * 255u;   // 4 bits -> 4 bytes of 0/255
                    uint8_t* tp = tdst + (size_t)c * W + x;
                    if (whole && (W & 3) == 0) {
                        *(uint32_t*)tp = t4;
                    } else {
                        for (int j = 0; j < 4 && x + j < W; j++) tp[j] = (uint8_t)(t4 >> (8 * j));
                    }
                }
                // 8x8 tiles: an even lane and its right neighbour hold the 8 pixels of one tile row; a tile is stored
                // every 8 rows (segments start on multiples of 8)
                tile_acc |= (unsigned long long)(B | (Bnext << 4)) << (8 * (c & 7));
                if ((c & 7) == 7 || c == ye - 1) {
                    if (tile_lane) bdst[(size_t)(c >> 3) * a.tnx + (x >> 3)] = tile_acc;
                    tile_acc = 0;
                }
                // border-start candidates of row c (rows c-1 and c)
                const uint32_t self4 = B, w4 = Emid & 15u, nw4 = Eup & 15u, n4 = (Eup >> 1) & 15u, ne4 = (Eup >> 2) & 15u;
                uint32_t outer4 = self4 & ~(w4 | nw4 | n4 | ne4);
                uint32_t hole4 = ~self4 & w4 & n4 & insx;   // row c is inside (c >= ys >= 0; c <= H-2 checked below)
                if (c > H - 2 || c < 1) hole4 = 0;
                auto stage = [&](uint32_t lo, uint32_t hi) {
                    const uint32_t ls = atomicAdd(&s_ntrig, 1u);
                    if (ls < LOCAL_TRIG) {
                        s_trig[ls] = lo, s_trig2[ls] = hi;
                    } else {
                        const uint32_t slot = atomicAdd(&a.raw_cnt[plane * TRIG_CNT_STRIDE], 1u);
                        if (slot < a.cap_raw)
                            a.raw[(size_t)plane * a.cap_raw + slot] = make_uint2(hi, lo);
                        else
                            atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_TRIG_OVERFLOW);
                    }
                };
                if (!a.seg_mode) {
                    uint32_t cand = out_lane ? (outer4 | (hole4 << 4)) : 0u;
                    while (cand) {
                        const int b = __builtin_ctz(cand);
                        cand &= cand - 1;
                        stage(((uint32_t)c << 16) | (uint32_t)(x + (b & 3)), (uint32_t)(b >> 2));
                    }
                } else if (out_lane) {
                    // waypoint cracks (pixel p set, 4-neighbour clear), see k_segments.hip:
                    //   W / E cracks on rows y % S == 0, N / S cracks on columns x % S == 0, plus every start-candidate crack.
                    // Record: lo = (pos(p) << 2) | code (E=0,N=1,W=2,S=3), hi = 1 if the crack is a border-start candidate.
                    const uint32_t e4 = (Emid >> 2) & 15u;
                    const uint32_t rowsel = ((c & a.grid_mask) == 0) ? 15u : 0u;
                    const uint32_t crW = self4 & ~w4, crN = self4 & ~n4 & xsel, crS = n4 & ~self4 & xsel;
                    const uint32_t crEz = ~self4 & w4 & 15u;          // seen from the clear pixel z = p + 1
                    uint32_t ev = ((crW & (rowsel | outer4))) | ((crEz & (rowsel | hole4)) << 4) | (crN << 8) | (crS << 12);
                    (void)e4;
                    while (ev) {
                        const int b = __builtin_ctz(ev);
                        ev &= ev - 1;
                        const int j = b & 3, kind = b >> 2;
                        const uint32_t px = (uint32_t)(x + j);
                        uint32_t pos, code, cand = 0;
                        if (kind == 0) pos = ((uint32_t)c << 16) | px, code = 2u, cand = (outer4 >> j) & 1u;
                        else if (kind == 1) pos = (((uint32_t)c << 16) | px) - 1u, code = 0u, cand = (hole4 >> j) & 1u;
                        else if (kind == 2) pos = ((uint32_t)c << 16) | px, code = 1u;
                        else pos = ((uint32_t)(c - 1) << 16) | px, code = 3u;
                        stage((pos << 2) | code, cand);
                    }
                }
            }
            Eup = Emid;
        }
    }
    // ---- flush the staged candidates: one atomic per wave on the plane's own counter
    __syncthreads();
    const uint32_t nl = min(s_ntrig, (uint32_t)LOCAL_TRIG);
    if (nl == 0) return;
    if (lane == 0) s_base = atomicAdd(&a.raw_cnt[plane * TRIG_CNT_STRIDE], nl);
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = lane; i < nl; i += 64) {
        if (base + i < a.cap_raw)
            a.raw[(size_t)plane * a.cap_raw + base + i] = make_uint2(s_trig2[i], s_trig[i]);
        else
            atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_TRIG_OVERFLOW);
    }
}

template <int R>
static void launch_adpt(hipStream_t s, const ThrArgs& a, dim3 grid) {
    hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT>), grid, dim3(64), 0, s, a);
}

static void fill_args(ThrArgs& a, const uint8_t* gray, const FrameGeom& g, const Buffers& b, int nthr, int t) {
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride;
    a.width = g.width, a.height = g.height, a.nthr = nthr, a.t = t;
    a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height);
    a.thres = b.thres, a.tiles = b.tiles, a.raw = b.raw, a.raw_cnt = b.raw_cnt, a.counters = b.counters, a.cap_raw = b.cap_raw;
    a.idelta = 0, a.n = 1, a.n_half = 0;
    a.seg_mode = b.seg_mode, a.grid_mask = b.grid_mask;
}

void launch_threshold(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    dim3 grid((g.width + STRIP_OUT - 1) / STRIP_OUT, (g.height + SEG - 1) / SEG, nframes);
    for (int t = 0; t < p.nthr; t++) {
        ThrArgs a;
        fill_args(a, gray, g, b, p.nthr, t);
        if (p.thres_method == ARUCOHIP_THRES_FIXED) {
            a.idelta = (int)floor(p.p1[t]);
            hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_FIXED>), grid, dim3(64), 0, s, a);
            continue;
        }
        a.n = p.block[t] * p.block[t], a.n_half = a.n / 2, a.idelta = p.idelta;
        switch (p.block[t] / 2) {
            case 1: launch_adpt<1>(s, a, grid); break;
            case 2: launch_adpt<2>(s, a, grid); break;
            case 3: launch_adpt<3>(s, a, grid); break;
            case 4: launch_adpt<4>(s, a, grid); break;
            case 5: launch_adpt<5>(s, a, grid); break;
            case 6: launch_adpt<6>(s, a, grid); break;
            case 7: launch_adpt<7>(s, a, grid); break;
            case 8: launch_adpt<8>(s, a, grid); break;
            case 9: launch_adpt<9>(s, a, grid); break;
            case 10: launch_adpt<10>(s, a, grid); break;
            case 11: launch_adpt<11>(s, a, grid); break;
            case 12: launch_adpt<12>(s, a, grid); break;
            case 13: launch_adpt<13>(s, a, grid); break;
            case 14: launch_adpt<14>(s, a, grid); break;
            default: launch_adpt<15>(s, a, grid); break;
        }
    }
}

// detectRectangles on a caller-supplied thresholded image (markerdetector.h:261): only the bit image + candidates.
void launch_binary_planes(hipStream_t s, const uint8_t* thres_in, const FrameGeom& g, int nframes, const Buffers& b) {
    dim3 grid((g.width + STRIP_OUT - 1) / STRIP_OUT, (g.height + SEG - 1) / SEG, nframes);
    ThrArgs a;
    fill_args(a, thres_in, g, b, 1, 0);
    hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_BINARY>), grid, dim3(64), 0, s, a);
}

}  // namespace ah
