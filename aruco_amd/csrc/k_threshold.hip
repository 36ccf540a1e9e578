// Kernel 1 — adaptive threshold + tiled binary image, one streaming pass per frame.
//
// Reference: MarkerDetector::thresHold -> cv::adaptiveThreshold(MEAN_C, BINARY_INV, b, C)
//            (/root/reference/src/markerdetector.cpp:643-677); the binary image feeds cv::findContours (:511).
//
// One wavefront owns a vertical strip of 256 pixels: lane l holds 4 horizontally adjacent pixels (one dword of the gray
// row), so a row is one aligned 256-byte load and one aligned 256-byte store per wave; lanes 0 and 63 also fetch the
// ceil(R/4) dwords left / right of the strip that the horizontal sums need (they enter the DPP lane shifts as the value
// shifted into the wave). The wave walks down its row segment keeping, in registers only,
//   * a ring of the last 2R+1 gray rows and of their horizontal box sums (two u16 per dword, pixel pairs (0,2) and (1,3)),
//   * the running vertical sums V (added row in, row out),
// so every gray byte is read from HBM/L2 once (plus the halo overlap); there is no LDS tile and no re-read. The slot of
// the ring that held the centre row is refilled at once with the row R+1 ahead, so R+1 loads are always in flight.
// mean = round(S / b^2) is never formed: src - mean <= -C  <=>  (src + C) * b^2 <= S + b^2/2 (exact in integers); for
// blocks up to 11x11 both sides fit 16 bits and the four comparisons are two packed 16-bit multiply-adds and subtracts.
// Per output row the wave stores one dword of threshold bytes per lane; every 8 rows it stores the 8x8-pixel tiles
// (one uint64 each) of the binary image with the 1-px frame cleared — the image cv::findContours binarises.
// HBM traffic per frame: read W*H, write W*H + W*H/8. Border-start candidates come from the tiles (k_contours.hip).
#include <stdlib.h>

#include "bits_tiles.h"
#include "internal.h"

namespace ah {

enum ThrMode { MODE_ADPT = 0, MODE_FIXED = 1, MODE_BINARY = 2 };

constexpr int SEG = 128;         // output rows per wave (a multiple of 8: tiles never straddle two waves)

struct ThrArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    int nthr, t;          // planes per frame, plane handled by this launch
    int idelta;           // ADPT: floor(C); FIXED: floor(threshold)
    int n, n_half;        // b*b and b*b/2
    int tnx, tny;         // tiles per row / column of the tiled binary image
    int fast;             // width, strides and base address are multiples of 4: dword loads and stores (template FAST)
    uint8_t* thres;
    uint64_t* tiles;
};

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t byte_of(uint32_t v, int i) { return (v >> (8 * i)) & 0xFFu; }

// value of the neighbouring lane through the DPP wave shift (a VALU operand modifier on gfx9, no LDS crossbar trip):
// from_right = lane i receives lane i+1 (wave_shl:1), lane 63 gets 0; the halo-aware variants live in the kernel
__device__ __forceinline__ uint32_t from_right(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }

constexpr int STRIP = 256;       // pixels per wave and row: one aligned 256-byte load and store per row

// P16: (255 + |C| + 1) * b^2 + b^2/2 < 32768, checked by the host. FAST: width, strides and base address are multiples
// of 4 (dword loads and stores).
template <int R, int MODE, bool P16, bool FAST>
__global__ __launch_bounds__(64) void threshold_strip_kernel(ThrArgs a) {
    constexpr int RING = 2 * R + 1;
    constexpr int NL = MODE == MODE_ADPT ? (R + 3) / 4 : 0;   // neighbour dwords needed on each side for the horizontal sums
    constexpr int NLA = NL > 0 ? NL : 1;
    constexpr bool HRING = NL == 1;            // the halo dword travels through a ring like the rows (prefetched R+1 rows ahead)
    constexpr bool PACKV = (RING + 1) * RING * 255 < 65536;
    static_assert(!P16 || PACKV, "packed compare needs packed sums");
    const int lane = threadIdx.x;
    const int frame = blockIdx.z;
    const int W = a.width, H = a.height;
    const int x = (int)blockIdx.x * STRIP + 4 * lane;                          // first pixel of this lane
    const int ys = (int)blockIdx.y * SEG, ye = min(ys + SEG, H);
    const int ye8 = (ye + 7) & ~7;                                             // virtual rows complete the last tile row
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int plane = frame * a.nthr + a.t;
    uint8_t* tdst = a.thres + (size_t)plane * W * H;
    uint64_t* bdst = a.tiles + (size_t)plane * a.tnx * a.tny;

    const bool out_lane = x < W;
    const bool even = (lane & 1) == 0;                                         // x is a multiple of 8 on these lanes
    const bool edge_lane = lane == 0 || lane == 63;                            // these two fetch the halo of the strip
    const bool tile_lane = even && out_lane;
    // byte mask of the pixels with 1 <= x <= W-2, restricted to the bit this lane contributes to the tile row byte:
    // even lanes own bits 0..3, odd lanes bits 4..7
    uint32_t rowsel = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (x + j >= 1 && x + j <= W - 2) rowsel |= (even ? 1u : 16u) << (9 * j);
    const int xc0 = min(max(x, 0), W - 1), xc1 = min(max(x + 1, 0), W - 1), xc2 = min(max(x + 2, 0), W - 1), xc3 = min(max(x + 3, 0), W - 1);
    // fast loads: the dword at the clamped position, BORDER_REPLICATE in x through a byte permute
    const uint32_t xa = (uint32_t)min(max(x, 0), (W - 4) & ~3);
    const uint32_t lsel = x < 0 ? 0x00000000u : (x >= W ? 0x03030303u : 0x03020100u);

    auto load_row = [&](int r) -> uint32_t {
        const uint8_t* row = src + (size_t)min(max(r, 0), H - 1) * a.row_stride;   // BORDER_REPLICATE in y
        if (FAST) return __builtin_amdgcn_perm(0u, *(const uint32_t*)(row + xa), lsel);
        return (uint32_t)row[xc0] | ((uint32_t)row[xc1] << 8) | ((uint32_t)row[xc2] << 16) | ((uint32_t)row[xc3] << 24);
    };
    // halo: lane 0 fetches the q-th dword left of the strip, lane 63 the q-th dword right of it (q = 1..NL)
    uint32_t hxa[NLA], hsel[NLA];
#pragma unroll
    for (int q = 1; q <= NL; q++) {
        const int xp = lane == 0 ? x - 4 * q : x + 4 * q;
        hxa[q - 1] = (uint32_t)min(max(xp, 0), (W - 4) & ~3);
        hsel[q - 1] = xp < 0 ? 0x00000000u : (xp >= W ? 0x03030303u : 0x03020100u);
    }
    auto load_halo = [&](int r, int q) -> uint32_t {
        uint32_t v = 0;
        if (edge_lane) {
            const uint8_t* row = src + (size_t)min(max(r, 0), H - 1) * a.row_stride;
            if (FAST) {
                v = __builtin_amdgcn_perm(0u, *(const uint32_t*)(row + hxa[q - 1]), hsel[q - 1]);
            } else {
                const int xp = lane == 0 ? x - 4 * q : x + 4 * q;
                v = (uint32_t)row[min(max(xp, 0), W - 1)] | ((uint32_t)row[min(max(xp + 1, 0), W - 1)] << 8) |
                    ((uint32_t)row[min(max(xp + 2, 0), W - 1)] << 16) | ((uint32_t)row[min(max(xp + 3, 0), W - 1)] << 24);
            }
        }
        return v;
    };
    // neighbour lane's value; lane 0 / lane 63 take the halo dword they hold in `edge` instead
    auto from_left_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false); };
    auto from_right_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false); };

    uint32_t G[RING], H02[RING], H13[RING], GH[HRING ? RING : 1];
#pragma unroll
    for (int k = 0; k < RING; k++) G[k] = 0, H02[k] = 0, H13[k] = 0;
    uint32_t V02 = 0, V13 = 0;                 // packed u16 pairs (PACKV) ...
    uint32_t V0 = 0, V1 = 0, V2 = 0, V3 = 0;   // ... or four 32-bit sums
    uint32_t acc = 0, tile_lo = 0;             // tile rows of this lane pair: 4 rows per dword
    const int cstp = a.idelta * a.n - a.n_half - 1;

    // One gray row enters the window: k = its ring slot (compile-time after unrolling), r = its row number.
    // STEADY = the window is complete: threshold the centre row r - R, store it, extend the tiles.
    auto row_step = [&](const int k, const int r, const bool steady) {
        const int kc = (k + RING - R) % RING;   // slot of the centre row r-R == slot of row r+R+1
        const uint32_t D0 = G[k];
        uint32_t t4 = 0;                        // threshold bytes (0 / 255) of the centre row
        const int c = r - R;                    // centre row of the window that ends at r
        if (MODE == MODE_ADPT) {
            // ---- horizontal box sums of the 4 pixels as pairs: s02 = s0 | s2 << 16, s13 = s1 | s3 << 16
            uint32_t Dn[2 * NL + 1];
            Dn[NL] = D0;
#pragma unroll
            for (int q = 1; q <= NL; q++) {
                const uint32_t hv = HRING ? GH[HRING ? k : 0] : load_halo(r, q);
                Dn[NL - q] = from_left_h(Dn[NL - q + 1], hv);     // lane i - q
                Dn[NL + q] = from_right_h(Dn[NL + q - 1], hv);    // lane i + q
            }
            // P(i) = byte i | byte i+2 << 16 of the row, relative to this lane's first pixel
            auto P = [&](int i) -> uint32_t {
                const int q = i + 4 * NL, q2 = q + 2;
                const int wa = q >> 2, wb = q2 >> 2;
                const uint32_t sel = 0x0C000C00u | (uint32_t)(q & 3) | ((uint32_t)((wb == wa ? 0 : 4) + (q2 & 3)) << 16);
                return __builtin_amdgcn_perm(Dn[wb], Dn[wa], sel);
            };
            uint32_t s02 = 0;
#pragma unroll
            for (int i = -R; i <= R; i++) s02 += P(i);
            const uint32_t s13 = s02 - P(-R) + P(R + 1);
            // ---- vertical running sums: row r enters, row r-RING leaves (it sits in the slot being overwritten)
            if (PACKV) {
                V02 = V02 - H02[k] + s02;
                V13 = V13 - H13[k] + s13;
            } else {
                V0 += (s02 & 0xFFFFu) - (H02[k] & 0xFFFFu), V2 += (s02 >> 16) - (H02[k] >> 16);
                V1 += (s13 & 0xFFFFu) - (H13[k] & 0xFFFFu), V3 += (s13 >> 16) - (H13[k] >> 16);
            }
            H02[k] = s02, H13[k] = s13;
            if (steady) {
                const uint32_t Gc = G[kc];
                if (P16) {
                    const short2v n2 = {(short)a.n, (short)a.n}, c2 = {(short)cstp, (short)cstp};
                    const short2v e = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, Gc, 0x0C020C00u));   // g0, g2
                    const short2v o = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, Gc, 0x0C030C01u));   // g1, g3
                    // (g + C) * n - n/2 - 1 - V < 0  <=>  (g + C) * n <= V + n/2
                    const short2v d02 = e * n2 + c2 - __builtin_bit_cast(short2v, V02);
                    const short2v d13 = o * n2 + c2 - __builtin_bit_cast(short2v, V13);
                    const uint32_t m02 = __builtin_bit_cast(uint32_t, d02 >> 15), m13 = __builtin_bit_cast(uint32_t, d13 >> 15);
                    t4 = __builtin_amdgcn_perm(m13, m02, 0x06020400u);
                } else {
                    const int v0 = PACKV ? (int)(V02 & 0xFFFFu) : (int)V0, v2 = PACKV ? (int)(V02 >> 16) : (int)V2;
                    const int v1 = PACKV ? (int)(V13 & 0xFFFFu) : (int)V1, v3 = PACKV ? (int)(V13 >> 16) : (int)V3;
                    const uint32_t tbits = (uint32_t)(((int)byte_of(Gc, 0) + a.idelta) * a.n <= v0 + a.n_half) |
                                           ((uint32_t)(((int)byte_of(Gc, 1) + a.idelta) * a.n <= v1 + a.n_half) << 1) |
                                           ((uint32_t)(((int)byte_of(Gc, 2) + a.idelta) * a.n <= v2 + a.n_half) << 2) |
                                           ((uint32_t)(((int)byte_of(Gc, 3) + a.idelta) * a.n <= v3 + a.n_half) << 3);
                    t4 = ((tbits * 0x00204081u) & 0x01010101u) * 255u;   // 4 bits -> 4 bytes of 0/255
                }
            }
        } else {
            uint32_t tbits = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = (int)byte_of(D0, j);
                tbits |= (uint32_t)(MODE == MODE_FIXED ? !(v > a.idelta) : (v != 0)) << j;
            }
            t4 = ((tbits * 0x00204081u) & 0x01010101u) * 255u;
        }
        G[kc] = load_row(r + R + 1);            // refill the slot the centre row just left: R+1 rows ahead
        if (HRING) GH[HRING ? kc : 0] = load_halo(r + R + 1, 1);
        if (!steady) return;
        // ---- threshold bytes
        if (MODE != MODE_BINARY && c < ye && out_lane) {
            uint8_t* trow = tdst + (size_t)c * W;
            if (FAST) {
                __builtin_nontemporal_store(t4, (uint32_t*)(trow + xa));   // written once, read by nobody here (1.43 -> 1.34 ms)
            } else {
                for (int j = 0; j < 4 && x + j < W; j++) trow[x + j] = (uint8_t)(t4 >> (8 * j));
            }
        }
        // ---- binary image for contour purposes, frame cleared: one byte per tile row on the even lanes
        const uint32_t rs = (c >= 1 && c <= H - 2) ? rowsel : 0u;
        const uint32_t nib = __builtin_amdgcn_sad_u8(t4 & rs, 0u, 0u);          // disjoint bits: byte sum == or
        const uint32_t rowbyte = nib | from_right(nib);
        acc = __builtin_amdgcn_alignbyte(rowbyte, acc, 1);                       // rows enter at the top byte
        if ((c & 3) == 3) {
            if (c & 4) {
                if (tile_lane) bdst[(size_t)(c >> 3) * a.tnx + (x >> 3)] = (uint64_t)tile_lo | ((uint64_t)acc << 32);
            } else {
                tile_lo = acc;
            }
        }
    };

    const int r_begin = ys - R;         // first gray row; the first complete window is centred on row ys
    const int r_last = ye8 - 1 + R;     // last gray row
    // prologue: rows r_begin .. r_begin+R are in flight before the first one is used
#pragma unroll
    for (int k = 0; k <= R; k++) {
        G[k] = load_row(r_begin + k);
        if (HRING) GH[HRING ? k : 0] = load_halo(r_begin + k, 1);
    }
    // warm-up: the first 2R rows only build the sums (slots 0 .. 2R-1)
#pragma unroll
    for (int k = 0; k < 2 * R; k++) row_step(k, r_begin + k, false);
    // first complete window: slot 2R
    row_step(2 * R, r_begin + 2 * R, true);
    // whole turns of the ring: straight-line code, no row tests
    int r0 = r_begin + RING;
    for (; r0 + RING - 1 <= r_last; r0 += RING) {
#pragma unroll
        for (int k = 0; k < RING; k++) row_step(k, r0 + k, true);
    }
    // the rest of the segment
#pragma unroll
    for (int k = 0; k < RING - 1; k++) {
        if (r0 + k > r_last) return;
        row_step(k, r0 + k, true);
    }
}

template <int R>
static void launch_adpt(hipStream_t s, const ThrArgs& a, int nframes) {
    dim3 grid((a.width + STRIP - 1) / STRIP, (a.height + SEG - 1) / SEG, nframes);
    constexpr bool CAN16 = R <= 5;
    const long lim = (long)(256 + abs(a.idelta)) * a.n + a.n_half;
    if (!a.fast)
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, false, false>), grid, dim3(64), 0, s, a);
    else if (CAN16 && lim < 32768)
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, CAN16, true>), grid, dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, false, true>), grid, dim3(64), 0, s, a);
}

static void fill_args(ThrArgs& a, const uint8_t* gray, const FrameGeom& g, const Buffers& b, int nthr, int t) {
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride;
    a.width = g.width, a.height = g.height, a.nthr = nthr, a.t = t;
    a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height);
    a.thres = b.thres, a.tiles = b.tiles;
    a.idelta = 0, a.n = 1, a.n_half = 0;
    a.fast = ((g.width | (int)(g.row_stride & 3) | (int)(g.frame_stride & 3) | (int)((uintptr_t)gray & 3)) & 3) == 0;
}

void launch_threshold(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    for (int t = 0; t < p.nthr; t++) {
        ThrArgs a;
        fill_args(a, gray, g, b, p.nthr, t);
        if (p.thres_method == ARUCOHIP_THRES_FIXED) {
            a.idelta = (int)floor(p.p1[t]);
            dim3 grid((g.width + STRIP - 1) / STRIP, (g.height + SEG - 1) / SEG, nframes);
            if (a.fast)
                hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_FIXED, false, true>), grid, dim3(64), 0, s, a);
            else
                hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_FIXED, false, false>), grid, dim3(64), 0, s, a);
            continue;
        }
        a.n = p.block[t] * p.block[t], a.n_half = a.n / 2, a.idelta = p.idelta;
        switch (p.block[t] / 2) {
            case 1: launch_adpt<1>(s, a, nframes); break;
            case 2: launch_adpt<2>(s, a, nframes); break;
            case 3: launch_adpt<3>(s, a, nframes); break;
            case 4: launch_adpt<4>(s, a, nframes); break;
            case 5: launch_adpt<5>(s, a, nframes); break;
            case 6: launch_adpt<6>(s, a, nframes); break;
            case 7: launch_adpt<7>(s, a, nframes); break;
            case 8: launch_adpt<8>(s, a, nframes); break;
            case 9: launch_adpt<9>(s, a, nframes); break;
            case 10: launch_adpt<10>(s, a, nframes); break;
            case 11: launch_adpt<11>(s, a, nframes); break;
            case 12: launch_adpt<12>(s, a, nframes); break;
            case 13: launch_adpt<13>(s, a, nframes); break;
            case 14: launch_adpt<14>(s, a, nframes); break;
            default: launch_adpt<15>(s, a, nframes); break;
        }
    }
}

// Optional erosion (north_star; off by default, no reference counterpart in this snapshot): 3x3 minimum of the thresholded
// image, pixels outside the image do not erode (cv::erode's default border). One thread per 4 pixels of a row.
__global__ __launch_bounds__(256) void erode3x3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= W) return;
    const size_t plane = (size_t)blockIdx.z * W * H;
    const uint8_t* s = src + plane;
    for (int j = 0; j < 4 && x0 + j < W; j++) {
        const int x = x0 + j;
        int v = 255;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int xx = x + dx, yy = y + dy;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) v = min(v, (int)s[(size_t)yy * W + xx]);
            }
        dst[plane + (size_t)y * W + x] = (uint8_t)v;
    }
}

// thres planes -> eroded planes in `tmp`, tiles rebuilt from them, eroded planes copied back (the thresholded image the API
// hands out is the eroded one)
void launch_erode(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, uint8_t* tmp) {
    const int W = g.width, H = g.height;
    hipLaunchKernelGGL(erode3x3_kernel, dim3((W / 4 + 256) / 256, H, nplanes), dim3(256), 0, s, b.thres, tmp, W, H);
    FrameGeom tg = g;
    tg.row_stride = (size_t)W, tg.frame_stride = (size_t)W * H;
    launch_binary_planes(s, tmp, tg, nplanes, b);
    (void)hipMemcpyAsync(b.thres, tmp, (size_t)nplanes * W * H, hipMemcpyDeviceToDevice, s);
}

// detectRectangles on a caller-supplied thresholded image (markerdetector.h:261): only the tiled binary image.
void launch_binary_planes(hipStream_t s, const uint8_t* thres_in, const FrameGeom& g, int nframes, const Buffers& b) {
    dim3 grid((g.width + STRIP - 1) / STRIP, (g.height + SEG - 1) / SEG, nframes);
    ThrArgs a;
    fill_args(a, thres_in, g, b, 1, 0);
    if (a.fast)
        hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_BINARY, false, true>), grid, dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_BINARY, false, false>), grid, dim3(64), 0, s, a);
}

}  // namespace ah
