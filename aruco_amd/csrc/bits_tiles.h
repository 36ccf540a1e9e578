// Tiled binary image + per-lane block cache for border following (device side).
//
// The image cv::findContours binarises is kept as 8x8-pixel tiles, one uint64 per tile (bit (y&7)*8 + (x&7)), row-major
// over tiles with one zero pad tile column and row: tiles_x = ceil(W/8)+1, tiles_y = ceil(H/8)+1. A border follower needs
// the 3x3 neighbourhood of one pixel per step; instead of three scattered reads per step every lane keeps a 32x32-pixel
// block (4x4 tiles, 128 B = eight 16-byte loads) in LDS as 32 row words and all lanes of the wave re-centre their blocks
// together, whenever one of them gets within a pixel of its block's edge. After a reload every lane is at least 11 pixels
// from every edge, so the loads (and their latency) are paid once per ~11 steps instead of every step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ah {

// at least 4 x 4 tiles: the followers load 4 x 4-tile blocks, and a frame narrower or lower than 25 pixels still needs one (the extra tiles stay zero)
__host__ __device__ inline int tiles_x(int width) { const int t = (width + 7) / 8 + 1; return t < 4 ? 4 : t; }
__host__ __device__ inline int tiles_y(int height) { const int t = (height + 7) / 8 + 1; return t < 4 ? 4 : t; }
// 128-tile strips of a tile row (one wave of the wide threshold kernel covers one: 64 lanes x 2 tiles)
__host__ __device__ inline int tile_strips(int width) { return (width + 1023) / 1024; }

// packed pixel position y << 16 | x; a step in direction d (0=E,1=NE,2=N,3=NW,4=W,5=SW,6=S,7=SE, y down) adds tb_dpos(d)
__device__ __forceinline__ uint32_t tb_dpos(int d) {
    const uint32_t nib = (0xA9840126u >> (4 * d)) & 15u;   // nibble d = (dy+1) << 2 | (dx+1)
    return ((nib >> 2) << 16) + (nib & 3u) - 65537u;
}

// up = NW,N,NE  mid = W,self,E  dn = SW,S,SE (bit 0..2)  ->  E,NE,N,NW,W,SW,S,SE (bit 0..7)
__device__ __forceinline__ uint32_t tb_assemble(uint32_t up, uint32_t mid, uint32_t dn) {
    return (mid >> 2) | ((up >> 2) << 1) | ((up & 2u) << 1) | ((up & 1u) << 3) | ((mid & 1u) << 4) | (dn << 5);
}

// 3x3 neighbourhood straight from the tiles (no cache): used where only a few steps are walked
__device__ __forceinline__ uint32_t tb_mask_direct(const uint64_t* __restrict__ tiles, int tnx, uint32_t pos) {
    const int x = (int)(pos & 0xFFFFu), y = (int)(pos >> 16);
    const int tx = (x - 1) >> 3, sh = (x - 1) & 7;
    uint32_t rows[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int yy = y - 1 + k;
        const uint64_t* t = tiles + (size_t)(yy >> 3) * tnx + tx;
        const uint32_t b16 = ((uint32_t)(t[0] >> (8 * (yy & 7))) & 0xFFu) | (((uint32_t)(t[1] >> (8 * (yy & 7))) & 0xFFu) << 8);
        rows[k] = (b16 >> sh) & 7u;
    }
    return tb_assemble(rows[0], rows[1], rows[2]);
}

// up to 64 pixels of row y starting at x (bit 0 = pixel x), assembled from 9 tile bytes
__device__ __forceinline__ uint64_t tb_row64(const uint64_t* __restrict__ tiles, int tnx, int x, int y, int* avail) {
    const int tx = x >> 3, sh = x & 7;
    const uint64_t* t = tiles + (size_t)(y >> 3) * tnx + tx;
    const int by = 8 * (y & 7);
    const int ntile = min(8, tnx - tx);          // stay inside the row of tiles (the pad column is zero)
    uint64_t v = 0;
    for (int k = 0; k < ntile; k++) v |= (uint64_t)((uint32_t)(t[k] >> by) & 0xFFu) << (8 * k);
    *avail = 8 * ntile - sh;
    return v >> sh;
}

// ---- per-lane 32x32 block in LDS: rows[r * LANES + lane], r = 0..31
constexpr int TB_ROWS = 32;

struct TileBlock {
    int bx, by;   // pixel origin of the block (multiples of 8)
};

// A block in two halves: the 16 tile loads (4 x 4 tiles of 8 bytes) into registers, and the transposition into the lane's 32 row words in LDS. A follower
// issues the loads of the block it will need NEXT while it still walks in the current one and transposes them when it gets there (k_contours.hip,
// walk_run): the load latency (1-2 us under load, once per 4-8 steps of a wave) was what the walkers spent their time on.
struct TileRegs {
    uint64_t t[16];
};
// the caller keeps the block inside the tile array (0 <= tx0 <= tnx-4, same in y)
__device__ __forceinline__ void tb_fetch_at(const uint64_t* __restrict__ tiles, int tnx, int tx0, int ty0, TileRegs& T, TileBlock& b) {
    b.bx = tx0 * 8, b.by = ty0 * 8;
    const uint64_t* p = tiles + (size_t)ty0 * tnx + tx0;
#pragma unroll
    for (int tr = 0; tr < 4; tr++) {
#pragma unroll
        for (int k = 0; k < 4; k++) T.t[tr * 4 + k] = p[k];
        p += tnx;
    }
}
template <int LANES>
__device__ __forceinline__ void tb_store(const TileRegs& T, uint32_t* rows, int lane) {
#pragma unroll
    for (int tr = 0; tr < 4; tr++) {
        const uint64_t t0 = T.t[tr * 4], t1 = T.t[tr * 4 + 1], t2 = T.t[tr * 4 + 2], t3 = T.t[tr * 4 + 3];
#pragma unroll
        for (int h = 0; h < 2; h++) {            // low / high half of the tiles: rows 4h .. 4h+3 of this tile row
            const uint32_t a0 = (uint32_t)(t0 >> (32 * h)), a1 = (uint32_t)(t1 >> (32 * h));
            const uint32_t a2 = (uint32_t)(t2 >> (32 * h)), a3 = (uint32_t)(t3 >> (32 * h));
            // row word k = byte k of a0 | byte k of a1 << 8 | byte k of a2 << 16 | byte k of a3 << 24: a 4x4 byte transpose in two rounds of
            // v_perm (8 instead of the 12 operations of "two selects and an or" per word)
            const uint32_t p01a = __builtin_amdgcn_perm(a1, a0, 0x05010400u), p01b = __builtin_amdgcn_perm(a1, a0, 0x07030602u);   // [a0.0 a1.0 a0.1 a1.1], [a0.2 a1.2 a0.3 a1.3]
            const uint32_t p23a = __builtin_amdgcn_perm(a3, a2, 0x05010400u), p23b = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
            uint32_t* r4 = rows + (tr * 8 + h * 4) * LANES + lane;
            r4[0] = __builtin_amdgcn_perm(p23a, p01a, 0x05040100u);
            r4[LANES] = __builtin_amdgcn_perm(p23a, p01a, 0x07060302u);
            r4[2 * LANES] = __builtin_amdgcn_perm(p23b, p01b, 0x05040100u);
            r4[3 * LANES] = __builtin_amdgcn_perm(p23b, p01b, 0x07060302u);
        }
    }
}
template <int LANES>
__device__ __forceinline__ void tb_load_at(const uint64_t* __restrict__ tiles, int tnx, int tx0, int ty0, uint32_t* rows, int lane, TileBlock& b) {
    TileRegs T;
    tb_fetch_at(tiles, tnx, tx0, ty0, T, b);
    tb_store<LANES>(T, rows, lane);
}

// block centred on pos: after the load pos is at least 11 pixels from every edge of the block (image borders aside)
template <int LANES>
__device__ __forceinline__ void tb_load(const uint64_t* __restrict__ tiles, int tnx, int tny, uint32_t pos, uint32_t* rows, int lane, TileBlock& b) {
    const int x = (int)(pos & 0xFFFFu), y = (int)(pos >> 16);
    const int tx0 = min(max((x - 12) >> 3, 0), tnx - 4), ty0 = min(max((y - 12) >> 3, 0), tny - 4);
    tb_load_at<LANES>(tiles, tnx, tx0, ty0, rows, lane, b);
}

// The same for a walk in progress: s = direction that points at the PREVIOUS border pixel, so the walk is heading the other way. M = steps the
// follower takes between two looks at the block edge (it needs M steps of room in every direction). The block is placed with the pixel
// M+1 .. M+8 pixels from the edge it comes from and 23-M .. 30-M pixels from the edge it is heading for (M = 8: 9..16 and 15..22, instead of 12..19
// from both): a border that keeps its direction travels 6..13 pixels before the next re-centring instead of 3..10 (M = 4: 14..21).
// Purely a placement: which block a lane holds never changes what it reads.
template <int M>
__device__ __forceinline__ void tb_place_dir(int tnx, int tny, int x, int y, int s, int* tx0, int* ty0) {
    // heading = -step(s): dx of direction s is +1 for E, NE, SE (0, 1, 7), -1 for NW, W, SW (3, 4, 5); dy is -1 for NE, N, NW (1, 2, 3), +1 for SW, S, SE
    const int sdx = (0x83u >> s) & 1 ? 1 : ((0x38u >> s) & 1 ? -1 : 0), sdy = (0x0Eu >> s) & 1 ? -1 : ((0xE0u >> s) & 1 ? 1 : 0);
    const int offx = sdx > 0 ? 23 - M : (sdx < 0 ? M + 1 : 12), offy = sdy > 0 ? 23 - M : (sdy < 0 ? M + 1 : 12);   // previous pixel to the east: heading west: far from the west edge
    *tx0 = min(max((x - offx) >> 3, 0), tnx - 4), *ty0 = min(max((y - offy) >> 3, 0), tny - 4);
}
template <int LANES, int M = 8>
__device__ __forceinline__ void tb_load_dir(const uint64_t* __restrict__ tiles, int tnx, int tny, uint32_t pos, int s, uint32_t* rows, int lane, TileBlock& b) {
    int tx0, ty0;
    tb_place_dir<M>(tnx, tny, (int)(pos & 0xFFFFu), (int)(pos >> 16), s, &tx0, &ty0);
    tb_load_at<LANES>(tiles, tnx, tx0, ty0, rows, lane, b);
}

__device__ __forceinline__ bool tb_inside(const TileBlock& b, uint32_t pos) {
    const int lx = (int)(pos & 0xFFFFu) - b.bx, ly = (int)(pos >> 16) - b.by;
    return lx >= 1 && lx <= 30 && ly >= 1 && ly <= 30;
}

template <int LANES>
__device__ __forceinline__ uint32_t tb_mask(const uint32_t* rows, int lane, const TileBlock& b, uint32_t pos) {
    const int lx = (int)(pos & 0xFFFFu) - b.bx, ly = (int)(pos >> 16) - b.by;
    const uint32_t* r = rows + (ly - 1) * LANES + lane;
    const int sh = lx - 1;
    return tb_assemble((r[0] >> sh) & 7u, (r[LANES] >> sh) & 7u, (r[2 * LANES] >> sh) & 7u);
}

// ---- a walk inside its block in the fewest vector instructions (round 3: the stream is bound by their issue; 39 -> 22 per step)
// lp  = (y - by) << 16 | (x - bx - 1): its low 5 bits are the shift that brings the pixel's 3x3 window down to bit 0 (the hardware uses no more of
//       a shift amount or bit-field offset), its upper half is the row inside the block;
// s1c = ((d + 5) & 7) | 0x0C0C0C00 with d = direction of the last step: the low bits are the rotation the next search starts at ((s + 1) & 7 with
//       s = (d + 4) & 7 pointing back), and the word as a whole is the v_perm selector that fetches the step's displacement from an 8-byte table
//       (byte (d + 5) & 7 = (dy + 1) << 4 | (dx + 1); selector bytes 0x0C read as zero).
constexpr uint32_t TB_S1C = 0x0C0C0C00u;
__device__ __forceinline__ uint32_t tb_s1c(int s) { return ((uint32_t)(s + 1) & 7u) | TB_S1C; }
__device__ __forceinline__ int tb_s_of(uint32_t s1c) { return (int)((s1c + 7u) & 7u); }
__device__ __forceinline__ uint32_t tb_base1(const TileBlock& b) { return (((uint32_t)b.by << 16) | (uint32_t)b.bx) + 1u; }   // pos = lp + tb_base1
// rb = rows + lane - LANES (row ly - 1 of the lane's block sits at rb[ly * LANES])
template <int LANES>
__device__ __forceinline__ void tb_step(const uint32_t* rb, uint32_t& lp, uint32_t& s1c) {
    const uint32_t* r = rb + (lp >> 16) * LANES;
    const uint32_t up = __builtin_amdgcn_ubfe(r[0], lp, 3u), mid = __builtin_amdgcn_ubfe(r[LANES], lp, 3u), dn = __builtin_amdgcn_ubfe(r[2 * LANES], lp, 3u);
    // ring E,NE,N,NW,W,SW,S,SE: the reversed `up` lands on bits 1..3
    const uint32_t m = (__builtin_bitreverse32(up) >> 28) | (mid >> 2) | ((mid & 1u) << 4) | (dn << 5);
    const uint32_t rot = (m | (m << 8)) >> (s1c & 31u);
    const uint32_t k = (uint32_t)__builtin_ctz(rot);   // a border pixel has a neighbour
    s1c = ((s1c + k + 5u) & 7u) | TB_S1C;
    const uint32_t b = __builtin_amdgcn_perm(0x01021222u, 0x21201000u, s1c);
    lp += ((b | (b << 12)) & 0x000F000Fu) - 65537u;
}

}  // namespace ah
