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
    int fast16;           // ... multiples of 16 and width >= 16: the 16-pixel-per-lane kernel applies
    int wide_ok;          // Tuning::threshold_wide
    int eo_ok;            // Tuning::threshold_eo
    int eo_strips, eo_segs, eo_frames;   // the round-3 kernel's 1-D grid: strips per row, row segments per frame, frames
    uint8_t* thres;
    uint64_t* tiles;
    uint64_t* tile_bits;  // non-empty-tile bitmap (internal.h), written by the wide kernel; launch_tile_bitmap for the other paths
    int nstrips;          // 128-tile strips per tile row
    uint64_t* stamps;     // timing only (else null): first / last device-clock reading of every wave of the wide kernel
    uint8_t* edge;        // lazy byte image (thres == null): the four border lines of every plane [P][2 W + 2 H], see expand_thres_kernel
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

// ---------------------------------------------------------------------------------------------
// The same pass with 16 pixels per lane (blocks up to 9x9, rows 16-byte aligned: the default configuration). A wave owns a
// 1024-pixel strip: one 16-byte load and one 16-byte non-temporal store per lane and row. The narrow kernel above spends
// 74 % of its wave time waiting for memory (SQ_WAIT_ANY, profiles/r02_sq_counters_b.txt) with 256 bytes per load in flight and
// 35 scalar instructions per 256 pixels; here a wave has PF KiB of loads in flight and a quarter of the scalar work per pixel.
//
// Vertical first: V = column sums of the last 2R+1 rows, kept per dword as two u16 pairs E = (V[0], V[2]) and O = (V[1], V[3])
// (the byte pairs `& 0x00FF00FF` and `>> 8 & 0x00FF00FF` of a gray dword), updated with the row that enters and the row that
// leaves (V + new - old never borrows across the halves). The horizontal 7-sum of a pixel pair is then a sum of pair-aligned
// neighbours: with XE_d = (E_d.hi, E_d+1.lo) and XO_d likewise (one v_alignbyte each),
//   C_d = E_d + O_d + XE_d + XO_d + XE_d-1 + XO_d-1,   S(E_d) = C_d + O_d-1,   S(O_d) = C_d + E_d+1        (R = 3)
// and for other R the window is assembled from the same pieces. Only the gray rows live in a register ring (window + PF
// rows of prefetch: no ring of horizontal sums), neighbour lanes contribute E / O of their first / last dword through the
// DPP wave shift, the strip's halo dword keeps its own column sums on the two edge lanes. A lane covers two tile columns.
// ---------------------------------------------------------------------------------------------
#ifndef THR_PF_N
#define THR_PF_N 3
#endif
constexpr int THR_PF = THR_PF_N;   // gray rows of prefetch per wave
constexpr int WSTRIP = 1024;    // pixels per wave and row; strips start on 1-KiB boundaries of the row (a 992-pixel strip with
                                // two halo lanes instead of halo loads was measured: the misaligned 16-byte accesses cost more)
constexpr uint32_t SEL02 = 0x0C020C00u, SEL13 = 0x0C030C01u;   // v_perm selectors: bytes (0, 2) / (1, 3) of a dword as u16 pairs

template <int R, int PF, int SEGW>
__device__ __forceinline__ void threshold_wide_body(const ThrArgs& a) {
    static_assert(R >= 1 && R <= 4 && PF >= 1 && PF <= 7, "one halo dword per side; the shortest segment has 8 rows");
    constexpr int RING = 2 * R + 1, N = RING + PF;
    const int lane = threadIdx.x;
    const int frame = blockIdx.z;
    const int W = a.width, H = a.height;
    const int x = (int)blockIdx.x * WSTRIP + 16 * lane;                        // first pixel of this lane
    const int ys = (int)blockIdx.y * SEGW, ye = min(ys + SEGW, H);
    const int ye8 = (ye + 7) & ~7;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int plane = frame * a.nthr + a.t;
    uint8_t* tdst = a.thres ? a.thres + (size_t)plane * W * H : nullptr;
    uint64_t* bdst = a.tiles + (size_t)plane * a.tnx * a.tny;

    const bool out_lane = x < W;
    const bool edge_lane = lane == 0 || lane == 63;
    const uint32_t xa = (uint32_t)min(x, W - 16);                              // W is a multiple of 16
    // lanes right of the image replicate the row's last pixel (BORDER_REPLICATE in x): one v_perm per dword takes either the
    // dword itself or that pixel
    const uint32_t rep_sel = x >= W ? 0x07070707u : 0x03020100u;
    // tile-row bits: dwords 0 / 2 give bits 0..3 of their tile's row byte, dwords 1 / 3 bits 4..7; pixels with 1 <= x <= W-2
    uint32_t rowsel[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        rowsel[d] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x + 4 * d + j >= 1 && x + 4 * d + j <= W - 2) rowsel[d] |= ((d & 1) ? 16u : 1u) << (9 * j);
    }
    auto load_row = [&](int r) -> uint4 {
        const uint8_t* row = src + (size_t)min(max(r, 0), H - 1) * a.row_stride;   // BORDER_REPLICATE in y
        const uint4 v = *(const uint4*)(row + xa);
        return make_uint4(__builtin_amdgcn_perm(v.w, v.x, rep_sel), __builtin_amdgcn_perm(v.w, v.y, rep_sel), __builtin_amdgcn_perm(v.w, v.z, rep_sel),
                          __builtin_amdgcn_perm(v.w, v.w, rep_sel));
    };
    // halo: lane 0 needs the dword left of the strip, lane 63 the dword right of it. Every lane loads a dword (the others their own first
    // one) so that the load sits in straight-line code and is waited for PF rows later (round 3: under `if (edge lane)` it had a wait of
    // its own that drained the prefetch queue every row); BORDER_REPLICATE in x is part of the unpack selectors.
    const int xp = lane == 0 ? x - 4 : x + 16;
    const uint32_t hxa = edge_lane ? (uint32_t)min(max(xp, 0), W - 4) : xa;
    const uint32_t hsel02 = xp < 0 ? 0x0C000C00u : (xp >= W ? 0x0C030C03u : SEL02), hsel13 = xp < 0 ? 0x0C000C00u : (xp >= W ? 0x0C030C03u : SEL13);
    auto load_halo = [&](int r) -> uint32_t {
        const uint8_t* row = src + (size_t)min(max(r, 0), H - 1) * a.row_stride;
        return *(const uint32_t*)(row + hxa);
    };
    auto from_left_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false); };
    auto from_right_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false); };

    uint4 G[N];
    uint32_t GH[N];
#pragma unroll
    for (int k = 0; k < N; k++) G[k] = make_uint4(0, 0, 0, 0), GH[k] = 0;
    uint32_t VE[4] = {0, 0, 0, 0}, VO[4] = {0, 0, 0, 0}, VEh = 0, VOh = 0;
    uint32_t accA = 0, accB = 0, loA = 0, loB = 0;     // tile rows of the lane's two tile columns, 4 rows per dword
    const int cstp = a.idelta * a.n - a.n_half - 1;
    const short2v n2 = {(short)a.n, (short)a.n}, c2 = {(short)cstp, (short)cstp};
    const int r_begin = ys - R;

    // Step s: gray row r_begin + s enters the window (ring slot s % N), row s - RING leaves it (slot (s + PF) % N, refilled with
    // row s + PF right after); OLD = there is a row to leave; STEADY = the window is complete, its centre row s - R is thresholded.
    auto step = [&](const int sn, const int s, const bool has_old, const bool steady) {
        const int so = (sn + PF) % N, sc = (sn + N - R) % N;
        const int r = r_begin + s, c = r - R;
        {
            const uint32_t nw[4] = {G[sn].x, G[sn].y, G[sn].z, G[sn].w};
            const uint32_t od[4] = {G[so].x, G[so].y, G[so].z, G[so].w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                VE[d] += __builtin_amdgcn_perm(0u, nw[d], SEL02), VO[d] += __builtin_amdgcn_perm(0u, nw[d], SEL13);
                if (has_old) VE[d] -= __builtin_amdgcn_perm(0u, od[d], SEL02), VO[d] -= __builtin_amdgcn_perm(0u, od[d], SEL13);
            }
            VEh += __builtin_amdgcn_perm(0u, GH[sn], hsel02), VOh += __builtin_amdgcn_perm(0u, GH[sn], hsel13);
            if (has_old) VEh -= __builtin_amdgcn_perm(0u, GH[so], hsel02), VOh -= __builtin_amdgcn_perm(0u, GH[so], hsel13);
        }
        G[so] = load_row(r + PF);
        GH[so] = load_halo(r + PF);
        if (!steady) return;
        // ---- horizontal sums of the column sums: E[d + 1] / O[d + 1] = pairs of dword d, d = -1 (left lane) .. 4 (right lane)
        uint32_t E[6], O[6];
        E[0] = from_left_h(VE[3], VEh), O[0] = from_left_h(VO[3], VOh);
        E[5] = from_right_h(VE[0], VEh), O[5] = from_right_h(VO[0], VOh);
#pragma unroll
        for (int d = 0; d < 4; d++) E[d + 1] = VE[d], O[d + 1] = VO[d];
        // pair (x, x + 2) for every pixel offset: at[4 * (d + 1) + j] = pair that starts at pixel j of dword d
        auto at = [&](int q) -> uint32_t {   // q = 4 * (dword + 1) + pixel, pixel 0..3
            const int w = q >> 2, j = q & 3;
            if (j == 0) return E[w];
            if (j == 1) return O[w];
            if (j == 2) return __builtin_amdgcn_alignbyte(E[w + 1], E[w], 2);
            return __builtin_amdgcn_alignbyte(O[w + 1], O[w], 2);
        };
        const uint4 Gc4 = G[sc];
        const uint32_t Gc[4] = {Gc4.x, Gc4.y, Gc4.z, Gc4.w};
        uint32_t t4[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            // S(E_d) = pairs starting at pixels -R .. R of the dword, S(O_d) = those starting at 1 - R .. R + 1 (the compiler
            // shares the common terms and the v_alignbyte results between the two sums and between neighbouring dwords)
            uint32_t sE = 0, sO = 0;
#pragma unroll
            for (int i = -R; i <= R; i++) sE += at(4 * (d + 1) + i), sO += at(4 * (d + 1) + i + 1);
            const short2v e = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, Gc[d], SEL02));   // g0, g2
            const short2v o = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, Gc[d], SEL13));   // g1, g3
            // (g + C) * n - n/2 - 1 - S < 0  <=>  (g + C) * n <= S + n/2
            const short2v d02 = e * n2 + c2 - __builtin_bit_cast(short2v, sE);
            const short2v d13 = o * n2 + c2 - __builtin_bit_cast(short2v, sO);
            // v_perm selectors 8..11 replicate the sign bit of a source's 16-bit halves: 255 where the difference is negative
            t4[d] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, d13), __builtin_bit_cast(uint32_t, d02), 0x0B090A08u);
        }
        if (c < ye && out_lane) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 tv = {t4[0], t4[1], t4[2], t4[3]};
            if (a.thres) {
                __builtin_nontemporal_store(tv, (u32x4*)(tdst + (size_t)c * W + xa));
            } else {
                // The byte image is not written: the tiles below hold it as bits except for its four border lines (cleared there,
                // as cv::findContours does), which go to a side array; expand_thres_kernel rebuilds the bytes when they are asked for.
                const size_t Wp = thres_edge_wp(W), Hp = thres_edge_hp(H);
                uint8_t* e = a.edge + (size_t)plane * thres_edge_stride(W, H);
                if (c == 0) *(u32x4*)(e + xa) = tv;
                if (c == H - 1) *(u32x4*)(e + Wp + xa) = tv;
                if (x == 0) e[2 * Wp + c] = (uint8_t)(t4[0] & 0xFFu);
                if (x == W - 16) e[2 * Wp + Hp + c] = (uint8_t)(t4[3] >> 24);
            }
        }
        // ---- binary image for contour purposes, frame cleared: one byte per tile row and tile column
        const bool rs_on = c >= 1 && c <= H - 2;
        uint32_t rbA = __builtin_amdgcn_sad_u8(t4[0] & rowsel[0], 0u, __builtin_amdgcn_sad_u8(t4[1] & rowsel[1], 0u, 0u));
        uint32_t rbB = __builtin_amdgcn_sad_u8(t4[2] & rowsel[2], 0u, __builtin_amdgcn_sad_u8(t4[3] & rowsel[3], 0u, 0u));
        rbA = rs_on ? rbA : 0u, rbB = rs_on ? rbB : 0u;
        accA = __builtin_amdgcn_alignbyte(rbA, accA, 1);                       // rows enter at the top byte
        accB = __builtin_amdgcn_alignbyte(rbB, accB, 1);
        if ((c & 3) == 3) {
            if (c & 4) {
                if (out_lane) {
                    uint64_t* t = bdst + (size_t)(c >> 3) * a.tnx + (x >> 3);
                    t[0] = (uint64_t)loA | ((uint64_t)accA << 32);
                    t[1] = (uint64_t)loB | ((uint64_t)accB << 32);
                }
                // which of the strip's 128 tiles hold a pixel: the start-candidate kernel only visits those (and their
                // right / lower neighbours)
                const unsigned long long balA = __ballot(out_lane && (loA | accA) != 0u), balB = __ballot(out_lane && (loB | accB) != 0u);
                if (lane == 0) {
                    uint64_t* bw = a.tile_bits + ((size_t)plane * a.tny + (c >> 3)) * (2 * a.nstrips) + 2 * blockIdx.x;
                    bw[0] = balA, bw[1] = balB;
                }
            } else {
                loA = accA, loB = accB;
            }
        }
    };

    const int s_last = ye8 - 1 + R - r_begin;   // >= 7 + 2R >= N - 1
    // prologue: PF rows in flight before the first one is used
#pragma unroll
    for (int k = 0; k < PF; k++) {
        G[k] = load_row(r_begin + k);
        GH[k] = load_halo(r_begin + k);
    }
#pragma unroll
    for (int s = 0; s < 2 * R; s++) step(s % N, s, false, false);      // the first 2R rows only build the sums
    step((2 * R) % N, 2 * R, false, true);                               // first complete window
#pragma unroll
    for (int s = RING; s < N; s++) step(s % N, s, true, true);          // up to the first whole turn of the ring
    int s0 = N;
    for (; s0 + N - 1 <= s_last; s0 += N) {                              // whole turns: straight-line code, slots are constants
#pragma unroll
        for (int k = 0; k < N; k++) step(k, s0 + k, true, true);
    }
#pragma unroll
    for (int k = 0; k < N - 1; k++) {
        if (s0 + k > s_last) return;
        step(k, s0 + k, true, true);
    }
}

// ---------------------------------------------------------------------------------------------
// Round 3: the 16-pixel-per-lane pass for 7x7 blocks rebuilt around what the instructions cost on this part (tools/mb_isa.hip,
// profiles/r03_isa_rates.txt: a 32-bit-encoded VOP2 issues in 1.1 ns per SIMD with four waves per SIMD, everything in a 64-bit encoding —
// v_perm, v_alignbyte, v_pk_*, v_add3, DPP — in 1.85 ns, v_mqsad_pk_u16_u8 in 6.9 ns). Same memory pattern as above (one 16-byte
// load per lane and row, PF rows in flight); what changed is the arithmetic per 16 pixels, 140 -> 78 vector instructions:
//   * every gray row is unpacked ONCE into u16 pairs E = (g0, g2), O = (g1, g3) per dword and stays in a register ring for the seven
//     rows it is in the window: the column sums take the leaving row from there (no second unpack) and the compare takes the centre row
//     from there (no third);
//   * all constants of the test ride along: the unpack writes g + 256 (byte 1 of each pair is the constant 1 from the v_perm's other
//     source), the column sums start from 7 * 256 plus an offset per column that repeats with period 7 and sums to n/2 - C n over any
//     seven neighbouring columns, so the horizontal sums come out as S' = S + 49 * 256 + n/2 - C n without a single extra addition, all
//     sums stay non-negative 16-bit fields of plain 32-bit adds, and  e = S' - (g + 256) n  is one v_pk_mad_u16 with the multiplier -n:
//     its sign bit says "black";
//   * the horizontal 7-sums are built from pair sums T = E + O: U_d = (T_d.hi, T_d+1.lo) is one v_alignbyte per dword (round 2 shifted E
//     and O separately), C_d = T_d + U_d + U_d-1 one v_add3, S(E_d) = C_d + O_d-1 and S(O_d) = C_d + E_d+1 one add each;
//   * a tile-row byte is two v_dot4_i32_i8 of the 0 / -1 "black" bytes with the weights 1, 2, 4 ... 64, -128 (zero for the image's first
//     and last column) on top of the mask of the allowed pixels: modulo 256 that is the byte of the WHITE pixels; the first and the last
//     image row are cleared when the tile is stored, not per row;
//   * the border columns of the lazy byte image are collected four rows at a time (round 2 stored a byte per row from a divergent branch).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SEL_E256 = 0x04020400u, SEL_O256 = 0x04030401u;   // v_perm(0x01010101, x, .): (x.b0 | 256, x.b2 | 256) and (x.b1 | 256, x.b3 | 256)

template <int PF, int SEGW, bool LAZY>
__device__ __forceinline__ void threshold_eo_body(const ThrArgs& a) {
    constexpr int R = 3, RING = 7;
    static_assert(PF >= 1 && PF <= 7, "rows of prefetch");
    constexpr int UNROLL = (RING % PF == 0) ? RING : RING * PF;   // ring slots and prefetch slots are compile-time constants inside a turn
    const int lane = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one). The strips of one row segment read the same 128-byte lines
    // where they meet (the halo dwords): the 1-D grid is unpacked so that they are 8 apart — same XCD, same L2, dispatched back to back.
    const int L = blockIdx.x;
    const int strip = (L >> 3) % a.eo_strips, rest = (L & 7) | (((L >> 3) / a.eo_strips) << 3);
    const int seg = rest % a.eo_segs, frame = rest / a.eo_segs;
    if (frame >= a.eo_frames) return;
    const int W = a.width, H = a.height;
    const int x = strip * WSTRIP + 16 * lane;                                  // first pixel of this lane
    const int ys = seg * SEGW, ye = min(ys + SEGW, H);
    const int ye8 = (ye + 7) & ~7;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    const int plane = frame * a.nthr + a.t;
    uint8_t* tdst = LAZY ? nullptr : a.thres + (size_t)plane * W * H;
    uint64_t* bdst = a.tiles + (size_t)plane * a.tnx * a.tny;
    const size_t Wp = thres_edge_wp(W), Hp = thres_edge_hp(H);
    uint8_t* edge = LAZY ? a.edge + (size_t)plane * thres_edge_stride(W, H) : nullptr;

    const bool out_lane = x < W;
    const bool last_lane = x == W - 16;                                        // holds the image's last column
    const uint32_t xa = (uint32_t)min(x, W - 16);                              // W is a multiple of 16
    // a lane right of the image loads the row's last 16 pixels; its first dword becomes that row's last pixel four times
    // (BORDER_REPLICATE in x) — the only dword of it its left neighbour, the last lane of the image, ever looks at
    const uint32_t rep_sel = x >= W ? 0x07070707u : 0x03020100u;
    // halo: lane 0 needs the dword left of the strip, lane 63 the dword right of it. EVERY lane loads a dword (the others their own first
    // one: the same line as their row load), so the load sits in the straight-line code next to the row load and is waited for PF rows
    // later like it; a load under `if (edge lane)` has its own wait right behind it, which drains the whole prefetch queue every row.
    // BORDER_REPLICATE in x is part of the unpack selectors.
    const int xp = lane == 0 ? x - 4 : x + 16;
    const bool edge_lane = lane == 0 || lane == 63;
    const uint32_t hxa = edge_lane ? (uint32_t)min(max(xp, 0), W - 4) : xa;
    const uint32_t hselE = xp < 0 ? 0x04000400u : (xp >= W ? 0x04030403u : SEL_E256), hselO = xp < 0 ? 0x04000400u : (xp >= W ? 0x04030403u : SEL_O256);
    auto from_left_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false); };
    auto from_right_h = [](uint32_t v, uint32_t edge) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false); };

    // ---- constants of the test folded into the sums (see the header): column offsets with period 7
    const int off = a.n_half - a.idelta * a.n;                                 // white <=> S + off - g n >= 0
    const int q = (off >= 0 ? off : off - (RING - 1)) / RING, rem = off - RING * q;   // floor division; 0 <= rem < 7
    auto col_init = [&](int col) -> uint32_t {                                 // start value of column col's sum: seven virtual rows of g = 0
        const int m = ((col % RING) + RING) % RING;
        return (uint32_t)(RING * 256 + q + (m < rem ? 1 : 0));                 // >= 0: the host only takes this kernel for |C| <= 200
    };
    uint32_t VE[4], VO[4], VEh, VOh;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        VE[d] = col_init(x + 4 * d) | (col_init(x + 4 * d + 2) << 16);
        VO[d] = col_init(x + 4 * d + 1) | (col_init(x + 4 * d + 3) << 16);
    }
    VEh = col_init(xp) | (col_init(xp + 2) << 16), VOh = col_init(xp + 1) | (col_init(xp + 3) << 16);
    // window rows as u16 pairs (+ 256), the strip's halo dword raw; the virtual rows they start with are g = 0
    uint32_t RE[RING][4], RO[RING][4], RH[RING];
#pragma unroll
    for (int k = 0; k < RING; k++) {
        RH[k] = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) RE[k][d] = 0x01000100u, RO[k][d] = 0x01000100u;
    }
    uint4 G[PF];
    uint32_t GH[PF];
    const uint32_t K1 = 0x01010101u;
    const uint32_t negn = (uint32_t)((65536 - a.n) & 0xFFFF) * 0x00010001u;
    // tile-row weights: dwords 0 / 2 carry bits 0..3 of their tile's row byte, dwords 1 / 3 bits 4..7; the image's first and last column
    // stay clear (cv::findContours' frame); wsum = the byte of all allowed pixels
    uint32_t wq[4], wsum[2] = {0, 0};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        wq[d] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int bit = 4 * (d & 1) + j, px = x + 4 * d + j;
            if (px >= 1 && px <= W - 2) wq[d] |= (bit == 7 ? 0x80u : (1u << bit)) << (8 * j), wsum[d >> 1] |= 1u << bit;
        }
    }
    uint32_t tA0 = 0, tA1 = 0, tB0 = 0, tB1 = 0;        // the last 8 tile-row bytes of the lane's two tile columns (64-bit shift registers)
    uint32_t colL = 0, colR = 0;                        // black bytes of the image's first / last column, the last 4 rows
    const int r_begin = ys - R;
    // address of the row that is loaded next: rows above / below the image repeat its first / last row (BORDER_REPLICATE in y)
    int r_next = r_begin;
    const uint8_t* p_next = src + (size_t)min(max(r_next, 0), H - 1) * a.row_stride;
    auto issue_loads = [&](const int kp) {
        // (round 4: the same loads as non-temporal loads - so that the gray rows, read once, would not evict the tiles and pools the gather kernels re-read -
        // changed nothing in the stream and cost this kernel 5 %: profiles/r04_experiments.txt 7)
        G[kp] = *(const uint4*)(p_next + xa);
        GH[kp] = *(const uint32_t*)(p_next + hxa);
        if (r_next >= 0 && r_next < H - 1) p_next += a.row_stride;
        r_next++;
    };

    // Step s: gray row r_begin + s enters the window (ring slot k = s % 7; it was loaded PF steps ago into prefetch slot kp = s % PF, which
    // takes row s + PF right after); STEADY = the window is complete, its centre row s - R is thresholded.
    auto step = [&](const int k, const int kp, const int s, const bool steady) {
        const int c = r_begin + s - R;
        {
            const uint4 Dn = G[kp];
            const uint32_t nw[4] = {__builtin_amdgcn_perm(Dn.w, Dn.x, rep_sel), Dn.y, Dn.z, Dn.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t en = __builtin_amdgcn_perm(K1, nw[d], SEL_E256), on = __builtin_amdgcn_perm(K1, nw[d], SEL_O256);
                VE[d] = VE[d] + en - RE[k][d], VO[d] = VO[d] + on - RO[k][d];
                RE[k][d] = en, RO[k][d] = on;
            }
            const uint32_t hn = GH[kp], ho = RH[k];
            VEh = VEh + __builtin_amdgcn_perm(K1, hn, hselE) - __builtin_amdgcn_perm(K1, ho, hselE);
            VOh = VOh + __builtin_amdgcn_perm(K1, hn, hselO) - __builtin_amdgcn_perm(K1, ho, hselO);
            RH[k] = hn;
        }
        issue_loads(kp);
        if (!steady) return;
        // ---- horizontal sums of the column sums
        uint32_t T[6];                                   // T[d + 1] = pair sums of dword d, d = -1 (left lane) .. 4 (right lane)
#pragma unroll
        for (int d = 0; d < 4; d++) T[d + 1] = VE[d] + VO[d];
        const uint32_t Th = VEh + VOh;
        T[0] = from_left_h(T[4], Th), T[5] = from_right_h(T[1], Th);
        const uint32_t Om1 = from_left_h(VO[3], VOh), E4 = from_right_h(VE[0], VEh);
        uint32_t U[5];                                   // U[d + 1] = (T_d.hi, T_d+1.lo)
#pragma unroll
        for (int d = 0; d < 5; d++) U[d] = __builtin_amdgcn_alignbyte(T[d + 1], T[d], 2);
        const int kc = (k + RING - R) % RING;            // slot of the centre row
        uint32_t tb[4];                                  // "black" bytes of the centre row: 255 = black
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t C = T[d + 1] + U[d + 1] + U[d];
            const uint32_t sE = C + (d == 0 ? Om1 : VO[d - 1]), sO = C + (d == 3 ? E4 : VE[d + 1]);
            typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
            const ushort2v n2 = __builtin_bit_cast(ushort2v, negn);
            const ushort2v eE = __builtin_bit_cast(ushort2v, RE[kc][d]) * n2 + __builtin_bit_cast(ushort2v, sE);
            const ushort2v eO = __builtin_bit_cast(ushort2v, RO[kc][d]) * n2 + __builtin_bit_cast(ushort2v, sO);
            tb[d] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, eO), __builtin_bit_cast(uint32_t, eE), 0x0B090A08u);
        }
        // ---- the thresholded bytes, where somebody wants them
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        if (!LAZY) {
            if (c < ye && out_lane) {
                const u32x4 tv = {~tb[0], ~tb[1], ~tb[2], ~tb[3]};
                __builtin_nontemporal_store(tv, (u32x4*)(tdst + (size_t)c * W + xa));
            }
        } else {
            // the byte image is not written: the tiles below hold it as bits except for its four border lines (cleared there, as
            // cv::findContours does), which go to a side array; expand_thres_kernel rebuilds the bytes when they are asked for
            if (c == 0 || c == H - 1) {
                if (out_lane) {
                    const u32x4 tv = {~tb[0], ~tb[1], ~tb[2], ~tb[3]};
                    *(u32x4*)(edge + (c == 0 ? 0 : Wp) + xa) = tv;
                }
            }
            colL = __builtin_amdgcn_perm(tb[0], colL, 0x04030201u);   // drop the oldest row, append byte 0 of dword 0 ...
            colR = __builtin_amdgcn_perm(tb[3], colR, 0x07030201u);   // ... and byte 3 of dword 3
            if ((c & 3) == 3) {
                if (x == 0) *(uint32_t*)(edge + 2 * Wp + (c - 3)) = ~colL;
                if (last_lane) *(uint32_t*)(edge + 2 * Wp + Hp + (c - 3)) = ~colR;
            }
        }
        // ---- binary image for contour purposes: one byte per tile row and tile column. black bytes are -1: the dot products take the
        // black pixels' weights off the allowed mask (the -128 of bit 7 is +128 modulo 256, and only the low byte is kept)
        const uint32_t rbA = (uint32_t)__builtin_amdgcn_sdot4((int)tb[1], (int)wq[1], __builtin_amdgcn_sdot4((int)tb[0], (int)wq[0], (int)wsum[0], false), false);
        const uint32_t rbB = (uint32_t)__builtin_amdgcn_sdot4((int)tb[3], (int)wq[3], __builtin_amdgcn_sdot4((int)tb[2], (int)wq[2], (int)wsum[1], false), false);
        tA0 = __builtin_amdgcn_alignbyte(tA1, tA0, 1), tA1 = __builtin_amdgcn_alignbyte(rbA, tA1, 1);   // rows enter at the top byte
        tB0 = __builtin_amdgcn_alignbyte(tB1, tB0, 1), tB1 = __builtin_amdgcn_alignbyte(rbB, tB1, 1);
        if ((c & 7) == 7) {
            // rows that are not part of the contour image: the image's first row, its last row and the virtual rows below it
            const int ty = c >> 3;
            uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
            if (ty == 0) mlo = 0xFFFFFF00u;
            if (ty == ((H - 1) >> 3)) {
                const int b = (H - 1) & 7;           // rows b .. 7 of this tile row go
                const uint64_t keep = (1ull << (8 * b)) - 1ull;
                mlo &= (uint32_t)keep, mhi &= (uint32_t)(keep >> 32);
            }
            const uint32_t a0 = tA0 & mlo, a1 = tA1 & mhi, b0 = tB0 & mlo, b1 = tB1 & mhi;
            if (out_lane) {
                uint64_t* t = bdst + (size_t)ty * a.tnx + (x >> 3);   // 8-byte aligned (the tile row length is odd): two stores, not one 16-byte vector
                t[0] = (uint64_t)a0 | ((uint64_t)a1 << 32);
                t[1] = (uint64_t)b0 | ((uint64_t)b1 << 32);
            }
            // which of the strip's 128 tiles hold a pixel: the start-candidate kernel only visits those (and their right / lower neighbours)
            const unsigned long long balA = __ballot(out_lane && (a0 | a1) != 0u), balB = __ballot(out_lane && (b0 | b1) != 0u);
            if (lane == 0) {
                uint64_t* bw = a.tile_bits + ((size_t)plane * a.tny + ty) * (2 * a.nstrips) + 2 * strip;
                bw[0] = balA, bw[1] = balB;
            }
        }
    };

    const int s_last = ye8 - 1 + R - r_begin;   // >= 7 + 2R
    // prologue: PF rows in flight before the first one is used
#pragma unroll
    for (int k = 0; k < PF; k++) issue_loads(k);
#pragma unroll
    for (int s = 0; s < 2 * R; s++) step(s % RING, s % PF, s, false);   // the first 2R rows only replace virtual rows
    // up to the first whole turn
#pragma unroll
    for (int s = 2 * R; s < UNROLL; s++) {
        if (s > s_last) return;
        step(s % RING, s % PF, s, true);
    }
    int s0 = UNROLL;
    for (; s0 + UNROLL - 1 <= s_last; s0 += UNROLL) {                   // whole turns: straight-line code, every slot is a constant
#pragma unroll
        for (int k = 0; k < UNROLL; k++) step(k % RING, k % PF, s0 + k, true);
    }
#pragma unroll
    for (int k = 0; k < UNROLL - 1; k++) {
        if (s0 + k > s_last) return;
        step(k % RING, k % PF, s0 + k, true);
    }
}

#ifndef THR_EO_WAVES
#define THR_EO_WAVES 3
#endif
template <int PF, int SEGW, bool LAZY>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(THR_EO_WAVES, THR_EO_WAVES))) void threshold_eo_kernel(ThrArgs a) {
    uint64_t t0 = 0;
    if (a.stamps) t0 = wall_clock64();
    threshold_eo_body<PF, SEGW, LAZY>(a);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // the wave's stores have left
        const uint64_t t1 = wall_clock64();
        // same unpacking as the body; the grid is padded to whole groups of 8 x strips and the surplus workgroups own no stamp
        const int L = blockIdx.x;
        const int strip = (L >> 3) % a.eo_strips, rest = (L & 7) | (((L >> 3) / a.eo_strips) << 3);
        const int seg = rest % a.eo_segs, frame = rest / a.eo_segs;
        if (threadIdx.x == 0 && frame < a.eo_frames) {
            const size_t w = ((size_t)frame * a.eo_segs + seg) * a.eo_strips + strip;
            a.stamps[2 * w] = t0, a.stamps[2 * w + 1] = t1;
        }
    }
}

// With hipEvent timing on, every wave also leaves the device clock (constant rate, hipDeviceAttributeWallClockRate) at its start and
// after its last store; stamp_reduce_kernel turns them into the launch's execution span. With several batches in flight the hipEvent
// interval around the launch also contains the time the dispatch waits for wave slots other batches' kernels hold; the span from the
// first wave's start to the last wave's end is what rocprofv3 reports per dispatch, and what the roofline fraction is about.
template <int R, int PF, int SEGW>
__global__ __launch_bounds__(64) void threshold_wide_kernel(ThrArgs a) {
    uint64_t t0 = 0;
    if (a.stamps) t0 = wall_clock64();
    threshold_wide_body<R, PF, SEGW>(a);
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // the wave's stores have left
        const uint64_t t1 = wall_clock64();
        if (threadIdx.x == 0) {
            const size_t w = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            a.stamps[2 * w] = t0, a.stamps[2 * w + 1] = t1;
        }
    }
}

// acc[0] += last end - first start of the launch's nwaves stamps, acc[1] += 1
__global__ __launch_bounds__(1024) void stamp_reduce_kernel(const uint64_t* __restrict__ stamps, size_t nwaves, unsigned long long* acc) {
    __shared__ unsigned long long smin[1024], smax[1024];
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = threadIdx.x; i < nwaves; i += 1024) lo = min(lo, (unsigned long long)stamps[2 * i]), hi = max(hi, (unsigned long long)stamps[2 * i + 1]);
    smin[threadIdx.x] = lo, smax[threadIdx.x] = hi;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + st]), smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && smax[0] > smin[0]) acc[0] += smax[0] - smin[0], acc[1] += 1;
}

// Lazy byte image: one plane of the thresholded image (0 / 255 bytes) from its tiles and its four border lines. A thread per 8 pixels.
__global__ __launch_bounds__(256) void expand_thres_kernel(const uint64_t* __restrict__ tiles, const uint8_t* __restrict__ edge, int W, int H, int tnx,
                                                           uint8_t* __restrict__ dst) {
    const int x8 = 8 * (int)(blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
    if (x8 >= W) return;
    const uint32_t bits = (uint32_t)(tiles[(size_t)(y >> 3) * tnx + (x8 >> 3)] >> (8 * (y & 7))) & 0xFFu;
    const size_t Wp = thres_edge_wp(W), Hp = thres_edge_hp(H);
    for (int j = 0; j < 8 && x8 + j < W; j++) {
        const int x = x8 + j;
        uint8_t v = (uint8_t)(((bits >> j) & 1u) * 255u);
        if (y == 0) v = edge[x];
        else if (y == H - 1) v = edge[Wp + x];
        else if (x == 0) v = edge[2 * Wp + y];
        else if (x == W - 1) v = edge[2 * Wp + Hp + y];
        dst[(size_t)y * W + x] = v;
    }
}

void launch_expand_thres(hipStream_t s, const FrameGeom& g, int plane, const Buffers& b) {
    const int W = g.width, H = g.height, tnx = tiles_x(W), tny = tiles_y(H);
    hipLaunchKernelGGL(expand_thres_kernel, dim3((W / 8 + 256) / 256, H), dim3(256), 0, s, b.tiles + (size_t)plane * tnx * tny,
                       b.thres_edge + (size_t)plane * thres_edge_stride(W, H), W, H, tnx, b.thres + (size_t)plane * W * H);
}

// the non-empty-tile bitmap for the paths whose kernel does not write it (narrow / FIXED / caller-supplied binary image): one
// wave per (strip, tile row, plane) reads the strip's 128 tiles
__global__ __launch_bounds__(64) void tile_bitmap_kernel(const uint64_t* __restrict__ tiles, uint64_t* __restrict__ tile_bits, int tnx, int tny, int nstrips) {
    const int strip = blockIdx.x, ty = blockIdx.y, plane = blockIdx.z, lane = threadIdx.x;
    const uint64_t* row = tiles + ((size_t)plane * tny + ty) * tnx;
    const int tx = 128 * strip + 2 * lane;
    const uint64_t A = tx < tnx ? row[tx] : 0ull, B = tx + 1 < tnx ? row[tx + 1] : 0ull;
    const unsigned long long balA = __ballot(A != 0ull), balB = __ballot(B != 0ull);
    if (lane == 0) {
        uint64_t* bw = tile_bits + ((size_t)plane * tny + ty) * (2 * nstrips) + 2 * strip;
        bw[0] = balA, bw[1] = balB;
    }
}

void launch_tile_bitmap(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b) {
    const int tnx = tiles_x(g.width), tny = tiles_y(g.height), ns = tile_strips(g.width);
    hipLaunchKernelGGL(tile_bitmap_kernel, dim3(ns, tny, nplanes), dim3(64), 0, s, b.tiles, b.tile_bits, tnx, tny, ns);
}

// returns true if the kernel that ran also wrote the non-empty-tile bitmap
template <int R>
static bool launch_adpt(hipStream_t s, const ThrArgs& a, int nframes, unsigned long long* stamp_acc) {
    dim3 grid((a.width + STRIP - 1) / STRIP, (a.height + SEG - 1) / SEG, nframes);
    constexpr bool CAN16 = R <= 5;
    const long lim = (long)(256 + abs(a.idelta)) * a.n + a.n_half;
    if constexpr (R <= 4) {
        if (a.fast16 && lim < 32768 && a.wide_ok) {
            // prefetch depth 3 rows, 128-row segments: the best of the sweep (PF 2..5, segments 64 / 128 / 256, forced register
            // budgets; profiles/r02_threshold_sweep.txt: 0.53 ms per 512 frames, everything else 0.54 .. 1.6)
            // A wave walks down its segment row by row: a launch of few frames gets shorter segments so that it still spreads over
            // the chip (one 640x480 frame: 4 waves of 128 rows took 91 us; 2R extra rows per segment are re-read, which only small
            // launches can afford)
            const int strips = (a.width + WSTRIP - 1) / WSTRIP;
            const long waves128 = (long)strips * ((a.height + 127) / 128) * nframes;
            int segs;
            if constexpr (R == 3) {
                if (a.eo_ok && abs(a.idelta) <= 200) {   // the round-3 form of this pass (7x7 blocks; its folded constants want a moderate C)
                    const dim3 blk(64);
#define EO_LAUNCH(SEG_)                                                                                                               \
    do {                                                                                                                              \
        segs = (a.height + (SEG_) - 1) / (SEG_);                                                                                      \
        ThrArgs e = a;                                                                                                                \
        e.eo_strips = strips, e.eo_segs = segs, e.eo_frames = nframes;                                                                \
        const long nb = (((long)strips * segs * nframes + 8L * strips - 1) / (8L * strips)) * (8L * strips);                          \
        if (a.thres)                                                                                                                  \
            hipLaunchKernelGGL((threshold_eo_kernel<THR_PF, SEG_, false>), dim3((unsigned)nb), blk, 0, s, e);                         \
        else                                                                                                                          \
            hipLaunchKernelGGL((threshold_eo_kernel<THR_PF, SEG_, true>), dim3((unsigned)nb), blk, 0, s, e);                          \
    } while (0)
                    if (waves128 >= 512)
                        EO_LAUNCH(128);
                    else if (waves128 * 4 >= 512)
                        EO_LAUNCH(32);
                    else
                        EO_LAUNCH(16);
#undef EO_LAUNCH
                    if (a.stamps) hipLaunchKernelGGL(stamp_reduce_kernel, dim3(1), dim3(1024), 0, s, a.stamps, (size_t)strips * segs * nframes, stamp_acc);
                    return true;
                }
            }
            if (waves128 >= 512) {
                segs = (a.height + 127) / 128;
                hipLaunchKernelGGL((threshold_wide_kernel<R, THR_PF, 128>), dim3(strips, segs, nframes), dim3(64), 0, s, a);
            } else if (waves128 * 4 >= 512) {
                segs = (a.height + 31) / 32;
                hipLaunchKernelGGL((threshold_wide_kernel<R, THR_PF, 32>), dim3(strips, segs, nframes), dim3(64), 0, s, a);
            } else {
                segs = (a.height + 15) / 16;
                hipLaunchKernelGGL((threshold_wide_kernel<R, THR_PF, 16>), dim3(strips, segs, nframes), dim3(64), 0, s, a);
            }
            if (a.stamps) hipLaunchKernelGGL(stamp_reduce_kernel, dim3(1), dim3(1024), 0, s, a.stamps, (size_t)strips * segs * nframes, stamp_acc);
            return true;
        }
    }
    if (!a.fast)
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, false, false>), grid, dim3(64), 0, s, a);
    else if (CAN16 && lim < 32768)
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, CAN16, true>), grid, dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((threshold_strip_kernel<R, MODE_ADPT, false, true>), grid, dim3(64), 0, s, a);
    return false;
}

static void fill_args(ThrArgs& a, const uint8_t* gray, const FrameGeom& g, const Buffers& b, int nthr, int t) {
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride;
    a.width = g.width, a.height = g.height, a.nthr = nthr, a.t = t;
    a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height);
    a.thres = b.thres, a.tiles = b.tiles, a.tile_bits = b.tile_bits, a.nstrips = tile_strips(g.width), a.wide_ok = b.tune.threshold_wide, a.eo_ok = b.tune.threshold_eo;
    a.idelta = 0, a.n = 1, a.n_half = 0;
    a.stamps = b.thr_stamp_on ? b.thr_stamps : nullptr;
    a.edge = nullptr;
    a.fast = ((g.width | (int)(g.row_stride & 3) | (int)(g.frame_stride & 3) | (int)((uintptr_t)gray & 3)) & 3) == 0;
    a.fast16 = g.width >= 16 && ((g.width | (int)(g.row_stride & 15) | (int)(g.frame_stride & 15) | (int)((uintptr_t)gray & 15) | (int)((uintptr_t)b.thres & 15)) & 15) == 0;
}

// lazy: the caller does not need the byte image now. Returns true if it was left out (every plane ran the 16-pixel-per-lane kernel,
// which keeps the border lines instead: launch_expand_thres rebuilds a plane on demand).
bool launch_threshold(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, bool lazy) {
    bool bitmap_done = true;
    if (lazy) {   // all planes or none
        ThrArgs a;
        fill_args(a, gray, g, b, p.nthr, 0);
        lazy = p.thres_method == ARUCOHIP_THRES_ADPT && a.fast16 && a.wide_ok && b.thres_edge != nullptr;
        for (int t = 0; t < p.nthr && lazy; t++) {
            const long n = (long)p.block[t] * p.block[t];
            lazy = p.block[t] / 2 >= 1 && p.block[t] / 2 <= 4 && (long)(256 + abs(p.idelta)) * n + n / 2 < 32768;
        }
    }
    for (int t = 0; t < p.nthr; t++) {
        ThrArgs a;
        fill_args(a, gray, g, b, p.nthr, t);
        if (lazy) a.thres = nullptr, a.edge = b.thres_edge;
        if (p.thres_method == ARUCOHIP_THRES_FIXED) {
            a.idelta = (int)floor(p.p1[t]);
            dim3 grid((g.width + STRIP - 1) / STRIP, (g.height + SEG - 1) / SEG, nframes);
            if (a.fast)
                hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_FIXED, false, true>), grid, dim3(64), 0, s, a);
            else
                hipLaunchKernelGGL((threshold_strip_kernel<0, MODE_FIXED, false, false>), grid, dim3(64), 0, s, a);
            bitmap_done = false;
            continue;
        }
        a.n = p.block[t] * p.block[t], a.n_half = a.n / 2, a.idelta = p.idelta;
        switch (p.block[t] / 2) {
            case 1: bitmap_done &= launch_adpt<1>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 2: bitmap_done &= launch_adpt<2>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 3: bitmap_done &= launch_adpt<3>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 4: bitmap_done &= launch_adpt<4>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 5: bitmap_done &= launch_adpt<5>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 6: bitmap_done &= launch_adpt<6>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 7: bitmap_done &= launch_adpt<7>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 8: bitmap_done &= launch_adpt<8>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 9: bitmap_done &= launch_adpt<9>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 10: bitmap_done &= launch_adpt<10>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 11: bitmap_done &= launch_adpt<11>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 12: bitmap_done &= launch_adpt<12>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 13: bitmap_done &= launch_adpt<13>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            case 14: bitmap_done &= launch_adpt<14>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
            default: bitmap_done &= launch_adpt<15>(s, a, nframes, (unsigned long long*)b.thr_acc); break;
        }
    }
    if (!bitmap_done) launch_tile_bitmap(s, g, nframes * p.nthr, b);
    return lazy;
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

// The same erosion on the lazy byte image (round 3): the thresholded planes exist as 8x8-pixel bit tiles (frame cleared) + their four border
// lines, and a 3x3 minimum of a binary image is an AND of nine shifted copies - 64 pixels per 64-bit operation, nothing leaves the bit domain and the byte
// image stays unwritten. One thread per tile: the tile and its eight neighbours with the frame pixels put back from the border lines, pixels outside the
// image read as set (they do not erode); out: the eroded tile with the frame cleared again and, from the threads on the frame, the eroded border lines.
struct ErodeArgs {
    const uint64_t* tiles;
    const uint8_t* edge;
    uint64_t* out_tiles;
    uint8_t* out_edge;
    int W, H, tnx, tny;
};
__device__ __forceinline__ uint64_t erode_full_tile(const ErodeArgs& a, const uint64_t* tiles, const uint8_t* e, int tx, int ty) {
    const int ntx = (a.W + 7) >> 3, nty = (a.H + 7) >> 3;
    if (tx < 0 || ty < 0 || tx >= ntx || ty >= nty) return ~0ull;           // outside the image: does not erode
    const uint64_t COL0 = 0x0101010101010101ull;
    uint64_t t = tiles[(size_t)ty * a.tnx + tx];
    const size_t Wp = thres_edge_wp(a.W), Hp = thres_edge_hp(a.H);
    // the frame pixels (cleared in the tiles) from the border lines: a line byte is 0 or 255
    auto row_bits = [&](const uint8_t* line) -> uint64_t {                  // 8 bytes -> 8 bits
        const uint64_t v = *(const uint64_t*)(line + 8 * tx) & COL0;
        return (v * 0x0102040810204080ull) >> 56;
    };
    auto col_bits = [&](const uint8_t* line, int c) -> uint64_t {           // 8 bytes (one per row) -> bit c of every row byte
        return (*(const uint64_t*)(line + 8 * ty) & COL0) << c;
    };
    if (ty == 0) t |= row_bits(e);
    if (ty == (a.H - 1) >> 3) t |= row_bits(e + Wp) << (8 * ((a.H - 1) & 7));
    if (tx == 0) t |= col_bits(e + 2 * Wp, 0);
    if (tx == (a.W - 1) >> 3) t |= col_bits(e + 2 * Wp + Hp, (a.W - 1) & 7);
    // pixels of the tile beyond the image's last column / row: set
    const int vc = min(8, a.W - 8 * tx), vr = min(8, a.H - 8 * ty);
    const uint64_t valid = (((1ull << vc) - 1ull) & 0xFFull) * COL0 & (vr >= 8 ? ~0ull : ((1ull << (8 * vr)) - 1ull));
    return t | ~valid;
}
__global__ __launch_bounds__(256) void erode_tiles_kernel(ErodeArgs a) {
    const int tx = blockIdx.x * blockDim.x + threadIdx.x, ty = blockIdx.y, plane = blockIdx.z;
    const int ntx = (a.W + 7) >> 3, nty = (a.H + 7) >> 3;
    if (tx >= a.tnx) return;
    const size_t tplane = (size_t)plane * a.tnx * a.tny, eplane = (size_t)plane * thres_edge_stride(a.W, a.H);
    uint64_t* out = a.out_tiles + tplane + (size_t)ty * a.tnx + tx;
    if (tx >= ntx || ty >= nty) {   // the pad column / row of the tile array
        *out = 0ull;
        return;
    }
    const uint64_t* tiles = a.tiles + tplane;
    const uint8_t* e = a.edge + eplane;
    const uint64_t COL0 = 0x0101010101010101ull, COL7 = 0x8080808080808080ull;
    auto hmin = [&](int y) -> uint64_t {   // tile (tx, y) ANDed with its left and right neighbours, column-wise
        const uint64_t X = erode_full_tile(a, tiles, e, tx, y), L = erode_full_tile(a, tiles, e, tx - 1, y), R = erode_full_tile(a, tiles, e, tx + 1, y);
        return X & (((X << 1) & ~COL0) | ((L >> 7) & COL0)) & (((X >> 1) & ~COL7) | ((R << 7) & COL7));
    };
    const uint64_t hC = hmin(ty), hN = hmin(ty - 1), hS = hmin(ty + 1);
    uint64_t E = hC & ((hC << 8) | (hN >> 56)) & ((hC >> 8) | (hS << 56));
    const int vc = min(8, a.W - 8 * tx), vr = min(8, a.H - 8 * ty);
    E &= (((1ull << vc) - 1ull) & 0xFFull) * COL0 & (vr >= 8 ? ~0ull : ((1ull << (8 * vr)) - 1ull));
    // the eroded border lines, then the frame cleared for the contour stage
    const size_t Wp = thres_edge_wp(a.W), Hp = thres_edge_hp(a.H);
    uint8_t* oe = a.out_edge + eplane;
    uint64_t frame = 0;
    if (ty == 0) {
        for (int c = 0; c < vc; c++) oe[8 * tx + c] = (uint8_t)(((E >> c) & 1ull) * 255u);
        frame |= 0xFFull;
    }
    if (ty == (a.H - 1) >> 3) {
        const int r = (a.H - 1) & 7;
        for (int c = 0; c < vc; c++) oe[Wp + 8 * tx + c] = (uint8_t)(((E >> (8 * r + c)) & 1ull) * 255u);
        frame |= 0xFFull << (8 * r);
    }
    if (tx == 0) {
        for (int r = 0; r < vr; r++) oe[2 * Wp + 8 * ty + r] = (uint8_t)(((E >> (8 * r)) & 1ull) * 255u);
        frame |= COL0;
    }
    if (tx == (a.W - 1) >> 3) {
        const int c = (a.W - 1) & 7;
        for (int r = 0; r < vr; r++) oe[2 * Wp + Hp + 8 * ty + r] = (uint8_t)(((E >> (8 * r + c)) & 1ull) * 255u);
        frame |= COL0 << c;
    }
    *out = E & ~frame;
}

// tiles + border lines -> eroded tiles + border lines (through `tmp`, which holds both), bitmap rebuilt; the byte image stays lazy
size_t erode_tiles_tmp_bytes(const FrameGeom& g, int nplanes) {
    return (size_t)nplanes * ((size_t)tiles_x(g.width) * tiles_y(g.height) * sizeof(uint64_t) + thres_edge_stride(g.width, g.height)) + 64;
}
void launch_erode_tiles(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, uint8_t* tmp) {
    ErodeArgs a;
    a.W = g.width, a.H = g.height, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height);
    const size_t tile_bytes = (size_t)nplanes * a.tnx * a.tny * sizeof(uint64_t), edge_bytes = (size_t)nplanes * thres_edge_stride(g.width, g.height);
    a.tiles = b.tiles, a.edge = b.thres_edge, a.out_tiles = (uint64_t*)tmp, a.out_edge = tmp + tile_bytes;
    (void)hipMemsetAsync(a.out_edge, 0, edge_bytes, s);
    hipLaunchKernelGGL(erode_tiles_kernel, dim3((a.tnx + 255) / 256, a.tny, nplanes), dim3(256), 0, s, a);
    (void)hipMemcpyAsync(b.tiles, a.out_tiles, tile_bytes, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(b.thres_edge, a.out_edge, edge_bytes, hipMemcpyDeviceToDevice, s);
    launch_tile_bitmap(s, g, nplanes, b);
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
    launch_tile_bitmap(s, g, nframes, b);
}

}  // namespace ah
