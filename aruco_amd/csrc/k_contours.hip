// Kernels 2-4 — border following, polygon approximation, quad filtering (detectRectangles).
//
// Reference: MarkerDetector::detectRectangles (/root/reference/src/markerdetector.cpp:496-635):
//   cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) :511, size filter :517, cv::approxPolyDP(eps = 0.05 n) :522,
//   4 vertices :526, cv::isContourConvex :535, min side :542-552, orientation :566-581, near-duplicates :586-627.
//
// cv::findContours is a sequential raster scan that relabels pixels while it follows borders. The point sequence
// of a border depends only on the binary image, its start pixel and whether it is a hole border, so the scan is
// replaced by (a) local start candidates found on the tiled image, thinned by the run rule (candidates_kernel: the start of a
// border is the first pixel of a horizontal run with nothing connected in the row above) and (b) one walker per candidate that follows the border
// with OpenCV's step rule and drops itself as soon as it proves it is not the scan's start:
//   outer border: a visited pixel precedes the start in raster order
//   hole border : a 4-neighbour background pixel examined during the walk precedes the trigger pixel
// or as soon as the border is longer than the size filter admits. Survivors are exactly the borders
// detectRectangles keeps, with the same start and direction.
#include <algorithm>

#include "bits_tiles.h"
#include "internal.h"
#include "decode_device.h"

namespace ah {

// Initial clockwise search of icvFetchContour: returns direction to the predecessor pixel, or -1 for an isolated pixel.
__device__ __forceinline__ int first_dir(uint32_t m, int s_end) {
    int s = s_end;
    do {
        s = (s - 1) & 7;
        if ((m >> s) & 1) break;
    } while (s != s_end);
    return s == s_end ? -1 : s;
}

// Run rule on the tiled image: can the crack's pixel be the raster-first pixel of its component (outer: pixel (x,y) set)
// or of its background hole (hole: pixel (x,y) clear, left neighbour set)?
//   outer - no set pixel 8-adjacent to the run of set pixels starting at x in the row above
//   hole  - no clear pixel directly above the run of clear pixels starting at x
// Runs are followed for at most 64 pixels; a longer run keeps the candidate (the walk decides).
__device__ __forceinline__ bool run_rule_tiles(const uint64_t* __restrict__ tiles, int tnx, int x, int y, int hole) {
    int avail, avail_up;
    const uint64_t mid = tb_row64(tiles, tnx, x, y, &avail), up = tb_row64(tiles, tnx, x, y - 1, &avail_up);
    if (!hole) {
        int L = (~mid) ? __builtin_ctzll(~mid) : 64;      // run of set pixels starting at x
        L = min(L, avail);
        const int hi = min(L, avail - 1);                  // blockers: row above, columns x+2 .. x+L
        const uint64_t m = hi >= 2 ? ((hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~3ull) : 0ull;
        return (up & m) == 0;
    }
    int L = mid ? __builtin_ctzll(mid) : 64;               // run of clear pixels starting at x
    L = min(L, avail);
    const int hi = min(L - 1, avail - 1);                  // row above, columns x+1 .. x+L-1 must all be set
    const uint64_t m = hi >= 1 ? ((hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~1ull) : 0ull;
    return (~up & m) == 0;
}

// ---------------------------------------------------------------------------------------------
// Kernel 1b: border-start candidates from the tiled binary image. One lane per 8x8 tile evaluates the 3x3 start rule
// of cv::findContours' raster scan for its 64 pixels with 64-bit logic on the tile and its W / N / NW / NE neighbours
//   outer: pixel set,  W, NW, N, NE clear         hole: pixel clear, W and N set (pixel inside the 1-px frame)
// and then the run rule for every hit: a candidate survives only if it can be the first pixel of its component / hole
// in raster order (outer - no set pixel 8-adjacent to its run of set pixels in the row above; hole - no clear pixel
// directly above its run of clear pixels). Survivors go to two lists per plane (outer starts in the first half of
// trig[plane], hole starts in the second half) so that a walker wavefront follows only one kind of border.
// Segment mode (k_segments.hip) instead gets the waypoint cracks of every tile: W / E cracks on grid rows, N / S cracks
// on grid columns plus every start-candidate crack, as records {candidate flag, pos << 2 | code}.
// ---------------------------------------------------------------------------------------------
struct CandArgs {
    const uint64_t* tiles;
    int tnx, tny;
    int width, height;
    uint2* trig;
    uint32_t* trig_cnt;
    uint2* raw;
    uint32_t* raw_cnt;
    uint32_t* counters;
    uint32_t cap_raw, cap_trig;
    int seg_mode, grid_mask;
    uint32_t* hash;          // segment mode: the plane's key -> node table, cleared here (segment_kernel fills it: one memset node less per call)
    uint32_t hash_size;
};

constexpr int CAND_THREADS = 256;
constexpr int CAND_STAGE = 512;   // staged records per list and round

__global__ __launch_bounds__(CAND_THREADS) void candidates_kernel(CandArgs a) {
    throughput_bound_priority();
    __shared__ uint2 s_keep[2][CAND_STAGE];
    __shared__ uint2 s_long[CAND_STAGE];
    __shared__ uint32_t s_n[2], s_base[2], s_nlong;
    const int plane = blockIdx.y;
    const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * a.tnx * a.tny;
    const int ntx = a.tnx - 1, nty = a.tny - 1;          // real tiles (the pad column / row holds no pixel)
    const int ntiles = ntx * nty;
    const uint64_t COL0 = 0x0101010101010101ull, COL7 = 0x8080808080808080ull;
    const uint32_t half = a.seg_mode ? a.cap_raw : a.cap_trig / 2;
    if (a.hash)
        for (uint32_t i = blockIdx.x * CAND_THREADS + threadIdx.x; i < a.hash_size; i += gridDim.x * CAND_THREADS) a.hash[(size_t)plane * a.hash_size + i] = 0xFFFFFFFFu;
    uint2* const out = a.seg_mode ? a.raw + (size_t)plane * a.cap_raw : a.trig + (size_t)plane * a.cap_trig;
    uint32_t* const out_cnt = (a.seg_mode ? a.raw_cnt : a.trig_cnt) + plane * TRIG_CNT_STRIDE;

    // a record either enters the staging list of its kind or, when the round's list is full, goes straight to the plane's list
    auto stage = [&](int kind, uint2 rec) {
        const uint32_t ls = atomicAdd(&s_n[kind], 1u);
        if (ls < CAND_STAGE) {
            s_keep[kind][ls] = rec;
        } else {
            const uint32_t slot = atomicAdd(&out_cnt[kind], 1u);
            if (slot < half)
                out[(size_t)kind * half + slot] = rec;
            else
                flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
        }
    };

    // a candidate whose run outlasts the pixels at hand waits for the 64-pixel test (or takes it at once if the list is full)
    auto defer = [&](uint2 rec) {
        const uint32_t ls = atomicAdd(&s_nlong, 1u);
        if (ls < CAND_STAGE)
            s_long[ls] = rec;
        else if (run_rule_tiles(tiles, a.tnx, (int)(rec.y & 0xFFFFu), (int)(rec.y >> 16), (int)rec.x))
            stage((int)rec.x, rec);
    };

    for (int i0 = blockIdx.x * CAND_THREADS; i0 < ntiles; i0 += gridDim.x * CAND_THREADS) {
        if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
        if (threadIdx.x == 2) s_nlong = 0;
        __syncthreads();
        const int i = i0 + threadIdx.x;
        if (i < ntiles) {
            const int ty = i / ntx, tx = i - ty * ntx;
            const uint64_t* t = tiles + (size_t)ty * a.tnx + tx;
            const uint64_t T = t[0];
            const uint64_t L = tx > 0 ? t[-1] : 0ull;
            const uint64_t U = ty > 0 ? t[-a.tnx] : 0ull, UL = (ty > 0 && tx > 0) ? t[-a.tnx - 1] : 0ull, UR = ty > 0 ? t[-a.tnx + 1] : 0ull;
            const uint64_t Rt = t[1];
            if (T | L | U) {     // a start needs a set pixel in the tile (outer) or set W and N neighbours (hole)
                const uint64_t Wn = ((T << 1) & ~COL0) | ((L >> 7) & COL0);
                const uint64_t N = (T << 8) | (U >> 56), NLt = (L << 8) | (UL >> 56), NRt = (Rt << 8) | (UR >> 56);
                const uint64_t NW = ((N << 1) & ~COL0) | ((NLt >> 7) & COL0);
                const uint64_t NE = ((N >> 1) & ~COL7) | ((NRt << 7) & COL7);
                // pixels with x <= W-2 and y <= H-2 (x >= 1 and y >= 1 follow from the set W / N neighbours)
                const int jmax = a.width - 2 - 8 * tx, imax = a.height - 2 - 8 * ty;
                const uint64_t colm = (jmax >= 7 ? 0xFFull : ((1ull << (jmax + 1)) - 1ull)) * COL0;
                const uint64_t rowm = imax >= 7 ? ~0ull : ((1ull << (8 * (imax + 1))) - 1ull);
                const uint64_t outer = T & ~(Wn | NW | N | NE);
                const uint64_t hole = ~T & Wn & N & colm & rowm;
                const uint32_t base = ((uint32_t)(8 * ty) << 16) | (uint32_t)(8 * tx);
                if (!a.seg_mode) {
                    // run rule on the 16 - (x & 7) pixels at hand (this tile and its right neighbour, rows y and y-1); only a
                    // run that leaves them undecided takes the 64-pixel test after the tile pass
                    const uint64_t both = outer | hole;
                    uint64_t m = both;
                    while (m) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1;
                        const int q = b >> 3, j = b & 7;
                        const uint32_t kind = (uint32_t)(hole >> b) & 1u;
                        const uint32_t mid = ((((uint32_t)(T >> (8 * q)) & 0xFFu) | (((uint32_t)(Rt >> (8 * q)) & 0xFFu) << 8)) >> j);
                        const uint32_t up = ((((uint32_t)(N >> (8 * q)) & 0xFFu) | (((uint32_t)(NRt >> (8 * q)) & 0xFFu) << 8)) >> j);
                        const int avail = 16 - j;
                        const uint32_t pos = base + ((uint32_t)q << 16) + (uint32_t)j;
                        // outer: run of set pixels from x, blockers = set pixels above columns x+2 .. x+L
                        // hole : run of clear pixels from x, blockers = clear pixels above columns x+1 .. x+L-1
                        const uint32_t runbits = (kind ? mid : ~mid) | (1u << avail);
                        const int Lr = __builtin_ctz(runbits);                       // run length inside the window
                        const int hi = kind ? min(Lr - 1, avail - 1) : min(Lr, avail - 1);
                        const uint32_t span = ((2u << hi) - 1u) & (kind ? ~1u : ~3u);
                        const bool blocked = ((kind ? ~up : up) & span) != 0;
                        if (blocked) continue;
                        if (Lr < avail)
                            stage((int)kind, make_uint2(kind, pos));
                        else
                            defer(make_uint2(kind, pos));
                    }
                } else {
                    // waypoint cracks (pixel p set, 4-neighbour clear). Record: y = (pos(p) << 2) | code (E=0,N=1,W=2,S=3),
                    // x = 1 if the crack is a border-start candidate.
                    uint64_t rsel = 0, csel = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (((8 * ty + q) & a.grid_mask) == 0) rsel |= 0xFFull << (8 * q);
                        if (((8 * tx + q) & a.grid_mask) == 0) csel |= COL0 << q;
                    }
                    const uint64_t crW = T & ~Wn & (rsel | outer);          // W crack of the set pixel
                    const uint64_t crE = ~T & Wn & (rsel | hole);           // E crack of p, seen from the clear pixel z = p + 1
                    const uint64_t crN = T & ~N & csel;                     // N crack of the set pixel
                    const uint64_t crS = N & ~T & csel;                     // S crack of the pixel above, seen from the clear pixel
                    for (int kind = 0; kind < 4; kind++) {
                        uint64_t m = kind == 0 ? crW : kind == 1 ? crE : kind == 2 ? crN : crS;
                        while (m) {
                            const int b = __builtin_ctzll(m);
                            m &= m - 1;
                            const uint32_t z = base + ((uint32_t)(b >> 3) << 16) + (uint32_t)(b & 7);
                            uint32_t pos, code, cand = 0;
                            if (kind == 0) pos = z, code = 2u, cand = (uint32_t)(outer >> b) & 1u;
                            else if (kind == 1) pos = z - 1u, code = 0u, cand = (uint32_t)(hole >> b) & 1u;
                            else if (kind == 2) pos = z, code = 1u;
                            else pos = z - 65536u, code = 3u;
                            stage(0, make_uint2(cand, (pos << 2) | code));
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            const uint32_t nlong = min(s_nlong, (uint32_t)CAND_STAGE);
            for (uint32_t j = threadIdx.x; j < nlong; j += CAND_THREADS) {
                const uint2 rec = s_long[j];
                if (run_rule_tiles(tiles, a.tnx, (int)(rec.y & 0xFFFFu), (int)(rec.y >> 16), (int)rec.x)) stage((int)rec.x, rec);
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 && s_n[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&out_cnt[threadIdx.x], min(s_n[threadIdx.x], (uint32_t)CAND_STAGE));
        __syncthreads();
        for (int kind = 0; kind < 2; kind++) {
            const uint32_t n = min(s_n[kind], (uint32_t)CAND_STAGE);
            for (uint32_t j = threadIdx.x; j < n; j += CAND_THREADS) {
                const uint32_t slot = s_base[kind] + j;
                if (slot < half)
                    out[(size_t)kind * half + slot] = s_keep[kind][j];
                else
                    flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1b, sparse form (walker mode): nine tiles in ten are empty and cannot hold a start (a start needs a set pixel in the
// tile, or in its left or upper neighbour). The threshold kernel leaves a non-empty-tile bitmap (two words per 128-tile
// strip: even tiles, odd tiles); a wave walks its share of the tile rows on those words alone, queues the tiles that can
// hold a start and evaluates 64 queued tiles at a time with all lanes busy: the start rule, the run rule on the 16 pixels at
// hand and — for the few runs that leave them — the 64-pixel run rule at once. Survivors are staged per wave and appended to
// the plane's lists with one atomic per list and flush.
// ---------------------------------------------------------------------------------------------
constexpr int CQ_CAP = 192;       // queued tile ids per wave (a round consumes 64; at most 128 join per strip)
constexpr int CS_CAP = 256;       // staged survivors per list and wave

struct SparseArgs {
    const uint64_t* tiles;
    const uint64_t* tile_bits;
    int tnx, tny, nstrips;
    int width, height;
    uint2* trig;
    uint32_t* trig_cnt;
    uint32_t* counters;
    uint32_t cap_trig;
    int drop_single;   // the size filter keeps no 1-point border (min contour >= 1): isolated pixels are not worth a start candidate
};

__global__ __launch_bounds__(64) void candidates_sparse_kernel(SparseArgs a) {
    throughput_bound_priority();
    __shared__ uint32_t s_q[CQ_CAP];
    __shared__ uint2 s_keep[2][CS_CAP];
    __shared__ uint32_t s_n[2];
    const int plane = blockIdx.y, lane = threadIdx.x;
    const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * a.tnx * a.tny;
    const uint64_t* __restrict__ bits = a.tile_bits + (size_t)plane * a.tny * (2 * a.nstrips);
    const int ntx = a.tnx - 1, nty = a.tny - 1;          // real tiles
    const uint32_t half = a.cap_trig / 2;
    uint2* const out = a.trig + (size_t)plane * a.cap_trig;
    uint32_t* const out_cnt = a.trig_cnt + plane * TRIG_CNT_STRIDE;
    const uint64_t COL0 = 0x0101010101010101ull, COL7 = 0x8080808080808080ull;
    if (lane < 2) s_n[lane] = 0;
    __syncthreads();

    // staged survivors -> the plane's lists (all lanes call it)
    auto flush = [&]() {
        __syncthreads();
        for (int kind = 0; kind < 2; kind++) {
            const uint32_t n = min(s_n[kind], (uint32_t)CS_CAP);
            if (n == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&out_cnt[kind], n);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            for (uint32_t j = lane; j < n; j += WAVE) {
                if (base + j < half)
                    out[(size_t)kind * half + base + j] = s_keep[kind][j];
                else
                    flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
            }
        }
        __syncthreads();
        if (lane < 2) s_n[lane] = 0;
        __syncthreads();
    };
    auto stage = [&](int kind, uint2 rec) {
        const uint32_t ls = atomicAdd(&s_n[kind], 1u);
        if (ls < CS_CAP) {
            s_keep[kind][ls] = rec;
        } else {   // the staging list is full: straight to the plane's list
            const uint32_t slot = atomicAdd(&out_cnt[kind], 1u);
            if (slot < half)
                out[(size_t)kind * half + slot] = rec;
            else
                flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
        }
    };
    // one queued tile per lane: the tile and its five neighbours (loads only; two sets are put in flight before either is used)
    struct TileSet {
        uint64_t T, L, U, UL, UR, Rt;
        int tx, ty;
    };
    auto fetch = [&](bool live, uint32_t id) -> TileSet {
        TileSet q;
        q.ty = (int)(id >> 16), q.tx = (int)(id & 0xFFFFu);
        q.T = q.L = q.U = q.UL = q.UR = q.Rt = 0ull;
        if (live) {
            const uint64_t* t = tiles + (size_t)q.ty * a.tnx + q.tx;
            q.T = t[0];
            q.L = q.tx > 0 ? t[-1] : 0ull;
            q.U = q.ty > 0 ? t[-a.tnx] : 0ull, q.UL = (q.ty > 0 && q.tx > 0) ? t[-a.tnx - 1] : 0ull, q.UR = q.ty > 0 ? t[-a.tnx + 1] : 0ull;
            q.Rt = t[1];
        }
        return q;
    };
    auto evaluate = [&](bool live, const TileSet& q) {
        if (live) {
            const int ty = q.ty, tx = q.tx;
            const uint64_t T = q.T, L = q.L, U = q.U, UL = q.UL, UR = q.UR, Rt = q.Rt;
            const uint64_t Wn = ((T << 1) & ~COL0) | ((L >> 7) & COL0);
            const uint64_t N = (T << 8) | (U >> 56), NLt = (L << 8) | (UL >> 56), NRt = (Rt << 8) | (UR >> 56);
            const uint64_t NW = ((N << 1) & ~COL0) | ((NLt >> 7) & COL0);
            const uint64_t NE = ((N >> 1) & ~COL7) | ((NRt << 7) & COL7);
            const int jmax = a.width - 2 - 8 * tx, imax = a.height - 2 - 8 * ty;
            const uint64_t colm = (jmax >= 7 ? 0xFFull : (jmax < 0 ? 0ull : ((1ull << (jmax + 1)) - 1ull))) * COL0;
            const uint64_t rowm = imax >= 7 ? ~0ull : (imax < 0 ? 0ull : ((1ull << (8 * (imax + 1))) - 1ull));
            uint64_t outer = T & ~(Wn | NW | N | NE);
            const uint64_t hole = ~T & Wn & N & colm & rowm;
            if (a.drop_single) {
                // An outer start with no set neighbour at all is a border of one point, which the size filter drops: in noisy frames
                // these are a large share of the candidates and each would cost the walker a block load. Rows 0..6 of the tile see
                // their lower neighbours inside T / L / Rt (row 7 would need the tiles below and stays a candidate).
                const uint64_t E = ((T >> 1) & ~COL7) | ((Rt << 7) & COL7);
                const uint64_t lower = (T >> 8) | (Wn >> 8) | (E >> 8);            // S, SW, SE of rows 0..6
                outer &= ~(~(E | lower) & 0x00FFFFFFFFFFFFFFull);
            }
            const uint32_t base = ((uint32_t)(8 * ty) << 16) | (uint32_t)(8 * tx);
            uint64_t m = outer | hole;
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int qr = b >> 3, j = b & 7;
                const uint32_t kind = (uint32_t)(hole >> b) & 1u;
                const uint32_t mid = ((((uint32_t)(T >> (8 * qr)) & 0xFFu) | (((uint32_t)(Rt >> (8 * qr)) & 0xFFu) << 8)) >> j);
                const uint32_t up = ((((uint32_t)(N >> (8 * qr)) & 0xFFu) | (((uint32_t)(NRt >> (8 * qr)) & 0xFFu) << 8)) >> j);
                const int avail = 16 - j;
                const uint32_t pos = base + ((uint32_t)qr << 16) + (uint32_t)j;
                const uint32_t runbits = (kind ? mid : ~mid) | (1u << avail);
                const int Lr = __builtin_ctz(runbits);
                const int hi = kind ? min(Lr - 1, avail - 1) : min(Lr, avail - 1);
                const uint32_t span = ((2u << hi) - 1u) & (kind ? ~1u : ~3u);
                if (((kind ? ~up : up) & span) != 0) continue;
                // a run that leaves the 16 pixels at hand takes the 64-pixel test at once (rare)
                if (Lr < avail || run_rule_tiles(tiles, a.tnx, (int)(pos & 0xFFFFu), (int)(pos >> 16), (int)kind)) stage((int)kind, make_uint2(kind, pos));
            }
        }
        __syncthreads();
        if (max(s_n[0], s_n[1]) > CS_CAP - 64) flush();   // wave-uniform: room for another round's typical yield
    };

    // The wave's tile rows are ty = blockIdx.x, + gridDim.x, ...; the bitmap words of a row and of the row above it (2 * nstrips each)
    // are one load by 4 * nstrips lanes, issued one row ahead; the strips read them with v_readlane.
    const int nw = 2 * a.nstrips;                       // words per tile row; 2 * nw <= 64 (tile_strips() <= 16 for the widths a handle accepts)
    auto row_words = [&](int ty) -> uint64_t {
        const int which = lane / nw, idx = lane - which * nw, row = ty - which;
        return (ty < nty && which < 2 && row >= 0) ? bits[(size_t)row * nw + idx] : 0ull;
    };
    auto word_of = [&](uint64_t v, int l) -> uint64_t {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
    };
    uint32_t qn = 0;   // queued tiles (wave-uniform)
    uint64_t nextw = row_words(blockIdx.x);
    for (int ty = blockIdx.x; ty < nty; ty += gridDim.x) {
        const uint64_t curw = nextw;
        nextw = row_words(ty + gridDim.x);
        for (int st = 0; st < a.nstrips; st++) {
            const uint64_t A = word_of(curw, 2 * st), Bm = word_of(curw, 2 * st + 1);
            const uint64_t UA = word_of(curw, nw + 2 * st), UB = word_of(curw, nw + 2 * st + 1);
            const uint64_t carry = st > 0 ? (word_of(curw, 2 * st - 1) >> 63) : 0ull;          // the previous strip's last (odd) tile
            // a tile can hold a start if it, its left or its upper neighbour holds a pixel
            const uint64_t needA = A | (Bm << 1) | carry | UA, needB = Bm | A | UB;
#pragma unroll
            for (int odd = 0; odd < 2; odd++) {
                const uint64_t need = odd ? needB : needA;
                const int tx = 128 * st + 2 * lane + odd;
                const bool act = ((need >> lane) & 1ull) && tx < ntx;
                const unsigned long long bal = __ballot(act);
                if (bal) {
                    if (act) s_q[qn + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)ty << 16) | (uint32_t)tx;
                    qn += (uint32_t)__popcll(bal);
                    __syncthreads();
                    if (qn >= 128) {   // keep room for the next 64 joins; two rounds' loads are in flight together
                        qn -= 128;
                        const TileSet q0 = fetch(true, s_q[qn + 64 + lane]), q1 = fetch(true, s_q[qn + lane]);
                        evaluate(true, q0);
                        evaluate(true, q1);
                    }
                }
            }
        }
    }
    {
        const bool l0 = qn >= 64 ? true : (uint32_t)lane < qn;
        const bool l1 = qn >= 64 && (uint32_t)lane < qn - 64;
        const TileSet q0 = fetch(l0, l0 ? s_q[lane] : 0u), q1 = fetch(l1, l1 ? s_q[64 + lane] : 0u);
        if (qn > 0) evaluate(l0, q0);
        if (qn > 64) evaluate(l1, q1);
    }
    flush();
}

void launch_start_candidates(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, int min_contour) {
    if (!b.seg_mode && b.tune.cand_sparse) {
        SparseArgs a;
        a.drop_single = min_contour >= 1;
        a.tiles = b.tiles, a.tile_bits = b.tile_bits, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height), a.nstrips = tile_strips(g.width);
        a.width = g.width, a.height = g.height, a.trig = b.trig, a.trig_cnt = b.trig_cnt, a.counters = b.counters, a.cap_trig = b.cap_trig;
        const int nty = a.tny - 1;
        const int waves = b.tune.cand_waves;
        hipLaunchKernelGGL(candidates_sparse_kernel, dim3(std::min(waves, nty), nplanes), dim3(64), 0, s, a);
        return;
    }
    CandArgs a;
    a.tiles = b.tiles, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height);
    a.width = g.width, a.height = g.height;
    a.trig = b.trig, a.trig_cnt = b.trig_cnt, a.raw = b.raw, a.raw_cnt = b.raw_cnt, a.counters = b.counters;
    a.cap_raw = b.cap_raw, a.cap_trig = b.cap_trig;
    a.seg_mode = b.seg_mode, a.grid_mask = b.grid_mask;
    a.hash = b.seg_mode ? b.hash : nullptr, a.hash_size = b.hash_mask + 1;
    const int ntiles = (a.tnx - 1) * (a.tny - 1);
    // a single frame has the chip to itself: a tile per thread; a batch of planes: four tiles per thread, at most cand_chunks workgroups per plane
    const int maxchunks = nplanes <= 2 ? 256 : b.tune.cand_chunks;
    const int per = nplanes <= 2 ? CAND_THREADS : 4 * CAND_THREADS;
    const int chunks = std::max(1, std::min(maxchunks, (ntiles + per - 1) / per));
    hipLaunchKernelGGL(candidates_kernel, dim3(chunks, nplanes), dim3(CAND_THREADS), 0, s, a);
}

constexpr int CK = 16;        // border steps between two checkpoints
constexpr int LEASH_MAX = 96;     // most steps a candidate can get in the first pass (sizes the LDS checkpoint array)
constexpr int LEASH_DEFAULT = 64; // a multiple of CK; ARUCOHIP_LEASH overrides for tuning
constexpr int PROBE = 10;     // steps of the reverse probe: stays inside the 32x32 block loaded around the start
constexpr int GEN_MAX = 30;           // generations of the long walks (kernel launches after the first pass)
constexpr int GEN_CNT_STRIDE = 32;    // uint32 words between two generation counters (one 128-byte line each)

struct WalkArgs {
    const uint64_t* tiles;
    int tnx, tny, nplanes;
    const uint2* trig;
    uint32_t* trig_cnt;    // per plane line: start candidates (outer, hole), descriptors, points
    ContourDesc* cdesc;
    uint32_t* counters;
    uint32_t cap_trig, cap_cdesc, cap_pool;
    int width, height;
    int min_contour, max_contour;
    // long walks: dense lists per kind and generation parity, {tkey, pos, pos1, n | s << 16} + ring id
    uint4* gen_state;      // [2 kinds][2 parities][gen_cap]
    uint32_t* gen_ring;    // [2 kinds][2 parities][gen_cap]
    uint32_t* gen_cnt;     // [(kind * (GEN_MAX + 2) + generation) * GEN_CNT_STRIDE] entries of each list
    uint32_t gen_cap;      // entries per list
    int gen_nx;            // sublists per list: 8 = one per XCD (a walk stays on the XCD its plane's first pass, tiles and contour_quad live on), 1 = one list
    int leash;             // steps every candidate gets in the first pass
    int gen, gen_steps;    // generation this launch processes (>= 1) and the steps it may take per walk
    int gen_blocks;        // 64-lane workgroups per kind in this launch (each loops over its share of the list)
    uint32_t long_cap;     // rings per plane and kind
    uint32_t* ring_cnt;   // per plane line: rings handed out (outer, hole)
    uint32_t* scratch;     // [P][2][long_cap][maxck] checkpoint ring of every long walk
    short2* pool;          // points of the kept borders; a short border keeps its checkpoints in front of its points
    int maxck;
};

enum WalkResult { WR_BAD = 0, WR_CLOSED = 1, WR_LIMIT = 2 };

// Generation lists (round 4): [kind][parity][gen_nx sublists][gen_cap / gen_nx] and one counter line per (kind, generation, sublist). With gen_nx = 8 sublist x holds
// the walks of the planes with plane % 8 == x - the planes whose first pass ran on XCD x (walker_kernel's unpacking) - and the generation workgroups with
// blockIdx % 8 == x, which the hardware deals to XCD x, take exactly that sublist: a plane's tiles are then only ever read through ONE XCD's L2, by the first
// pass, every generation and contour_quad alike (until round 4 a generation wave held walks of 64 different planes, whichever XCD it ran on).
constexpr int GEN_NX_MAX = 8;
__device__ __forceinline__ uint32_t gen_cnt_index(int kind, int gen, int x) { return (uint32_t)(((kind * (GEN_MAX + 2) + gen) * GEN_NX_MAX + x) * GEN_CNT_STRIDE); }
__device__ __forceinline__ size_t gen_list_base(const WalkArgs& a, int kind, int parity, int x) {
    return ((size_t)(kind * 2 + parity) * a.gen_nx + x) * (a.gen_cap / a.gen_nx);
}

// A checkpoint lets any lane resume the walk at step k*CK: pixel and the direction that points at the previous pixel.
__device__ __forceinline__ uint32_t pack_ck(uint32_t pos, int s) { return (pos & 0x3FFFu) | ((pos >> 16) << 14) | ((uint32_t)s << 28); }

// One step of OpenCV's border follower at pixel `pos`, previous pixel in direction s, neighbourhood m (bit d = neighbour in
// direction d set). Returns the direction of the next border pixel; *bad = the walk has proven that `tkey` is not the
// scan's start: outer border - the pixel precedes the start in raster order; hole border - a background 4-neighbour
// examined from this pixel precedes the trigger pixel. MIRROR: m, s and the result are directions of the x-mirrored image
// (the reverse probe follows the same border the other way round); pos and tkey are always real coordinates.
template <bool HOLE, bool MIRROR>
__device__ __forceinline__ int walk_step(uint32_t m, int s, uint32_t pos, uint32_t tkey, uint32_t pos0, bool* bad) {
    const uint32_t sh = (uint32_t)(s + 1) & 7u;
    const uint32_t rot = ((m | (m << 8)) >> sh);
    const int k = __builtin_ctz(rot | 0x100u);
    if (HOLE) {
        // examined 4-neighbours z = pos + off precede tkey iff delta = pos - tkey < -off; N < W < E < S in raster order
        const int delta = (int)(pos - tkey);
        bool b = false;
        if (delta < 65536) {                       // only near or above the trigger row: rare
            uint32_t ex = ((1u << k) - 1u) << sh;
            ex |= ex >> 8;
            const uint32_t bitW = MIRROR ? 1u : 16u, bitE = MIRROR ? 16u : 1u;
            const uint32_t cm = 4u | (delta < 1 ? bitW : 0u) | (delta < -1 ? bitE : 0u) | (delta < -65536 ? 64u : 0u);
            b = (ex & cm) != 0;
        }
        *bad = b;
    } else {
        *bad = pos < pos0;
    }
    return (int)((sh + (uint32_t)k) & 7u);
}

__device__ __forceinline__ uint32_t tb_assemble_mirror(uint32_t up, uint32_t mid, uint32_t dn) {
    // directions of the x-mirrored image: E' = W, NE' = NW, N' = N, NW' = NE, W' = E, SW' = SE, S' = S, SE' = SW
    return (mid & 1u) | ((up & 1u) << 1) | ((up & 2u) << 1) | ((up >> 2) << 3) | ((mid >> 2) << 4) | ((dn >> 2) << 5) | ((dn & 2u) << 5) | ((dn & 1u) << 7);
}
template <int LANES>
__device__ __forceinline__ uint32_t tb_mask_mirror(const uint32_t* rows, int lane, const TileBlock& b, uint32_t pos) {
    const int lx = (int)(pos & 0xFFFFu) - b.bx, ly = (int)(pos >> 16) - b.by;
    const uint32_t* r = rows + (ly - 1) * LANES + lane;
    const int sh = lx - 1;
    return tb_assemble_mirror((r[0] >> sh) & 7u, (r[LANES] >> sh) & 7u, (r[2 * LANES] >> sh) & 7u);
}

#ifndef PER_LANE_RELOAD
#define PER_LANE_RELOAD 1
#endif
#ifndef TB_DIRECTED
#define TB_DIRECTED 1   // re-centre a walk's block ahead of the walk instead of around it (bits_tiles.h: tb_load_dir)
#endif
#ifndef CHUNK_N
#define CHUNK_N 8
#endif
constexpr int CHUNK = CHUNK_N;   // steps taken between two looks at the block edge (divides CK, LEASH and every generation length). Round 3 measured 4: a lane then
                                 // needs 4 steps of room, a directed block serves 14..21 pixels of travel instead of 6..13 and the generations fetch a fifth fewer
                                 // bytes - +1.2 % on the bench's stream, -17 % on the cluttered one (148k against 178k frames/s: many short, wiggly borders pay
                                 // the edge tests twice as often). 8 stays.

// CHUNK border steps of every walking lane inside its 32x32 block, without edge tests (the caller has made room). The walk lives in block
// coordinates (bits_tiles.h: lp, s1c); p0b / p1b / tkb are the start pixel, the start's predecessor and the trigger in the same coordinates. n is a
// multiple of CHUNK at a chunk start, so the checkpoint of every CK-th step is stored there: ck_at(n / CK) is where this lane's goes.
template <bool HOLE, int LANES, typename CkAt>
__device__ __forceinline__ void walk_chunk(const uint32_t* rb, uint32_t base1, uint32_t p0b, uint32_t p1b, uint32_t tkb, uint32_t pos0, uint32_t lim, uint32_t& lp,
                                           uint32_t& s1c, uint32_t& n, bool& walking, bool& was_bad, bool& was_closed, CkAt ck_at) {
#pragma unroll
    for (int j = 0; j < CHUNK; j++) {
        if (walking) {
            if (j == 0 && (n & (CK - 1)) == 0) *ck_at(n / CK) = pack_ck(lp + base1, tb_s_of(s1c));
            const uint32_t* r = rb + (lp >> 16) * LANES;
            const uint32_t up = __builtin_amdgcn_ubfe(r[0], lp, 3u), mid = __builtin_amdgcn_ubfe(r[LANES], lp, 3u), dn = __builtin_amdgcn_ubfe(r[2 * LANES], lp, 3u);
            // ring E,NE,N,NW,W,SW,S,SE: the reversed `up` lands on bits 1..3
            const uint32_t m = (__builtin_bitreverse32(up) >> 28) | (mid >> 2) | ((mid & 1u) << 4) | (dn << 5);
            // walk_step<HOLE, false> on the selector-shaped direction state: the search starts at rotation (s + 1) & 7 = the state's low bits
            const uint32_t rot = (m | (m << 8)) >> (s1c & 31u);
            const uint32_t k = (uint32_t)__builtin_ctz(rot);   // a pixel of a border that is being followed has a neighbour
            bool bad;
            if (HOLE) {
                // examined 4-neighbours z = pos + off precede tkey iff delta = pos - tkey < -off; N < W < E < S in raster order
                const int delta = (int)(lp - tkb);
                uint32_t hit = 0;   // an integer, not a flag, crosses the branch: a flag would be turned into one and back
                if (delta < 65536) {                       // only near or above the trigger row
                    uint32_t ex;                           // ((1 << k) - 1) << sh: directions examined and found empty
                    asm("v_bfm_b32 %0, %1, %2" : "=v"(ex) : "v"(k), "v"(s1c));   // uses the low 5 bits of both, and the state's are (s + 1) & 7
                    const uint32_t cm = 4u | (delta < 1 ? 16u : 0u) | (delta < -1 ? 1u : 0u) | (delta < -65536 ? 64u : 0u);
                    hit = (ex | (ex >> 8)) & cm;
                }
                asm volatile("" : "+v"(hit));   // keeps the compare below the branch, where its result is a lane mask
                bad = hit != 0;
            } else {
                bad = lp + base1 < pos0;
            }
            const uint32_t ns1c = ((s1c + k + 5u) & 7u) | TB_S1C;   // d = (sh + k) & 7, back direction (d + 4) & 7, next rotation (d + 5) & 7
            const uint32_t nb = __builtin_amdgcn_perm(0x01021222u, 0x21201000u, ns1c);
            const uint32_t nlp = lp + ((nb | (nb << 12)) & 0x000F000Fu) - 65537u;
            ++n;
            const bool closed = nlp == p0b && lp == p1b;
            was_bad |= bad, was_closed |= closed;
            walking = !(bad | closed | (n >= lim));
            lp = nlp, s1c = ns1c;   // also when the walk has ended: position and direction only matter to a walk that goes on (WR_LIMIT)
        }
    }
}

// Follows one border per lane until every lane's walk has ended: proven not to be the scan's start (WR_BAD), closed
// (WR_CLOSED) or n == lim (WR_LIMIT, state advanced so that the walk can be resumed). All lanes of the wave step together.
// The per-lane 32x32 block is re-centred (by all lanes at once) when a walking lane is within CHUNK pixels of its edge,
// then CHUNK steps run without any edge test; n is a multiple of CHUNK at every chunk start, so checkpoints (every CK
// steps) are only stored there. ck_at(n / CK) is where this lane's checkpoint goes.
template <bool HOLE, int LANES, typename CkAt>
__device__ __forceinline__ int walk_run(const uint64_t* __restrict__ tiles, int tnx, int tny, uint32_t* rows, int lane, bool live, uint32_t tkey,
                                        uint32_t pos0, uint32_t pos1, uint32_t lim, uint32_t nmax, uint32_t& pos, uint32_t& n, int& s, CkAt ck_at,
                                        const TileBlock* preloaded = nullptr) {
    TileBlock blk;
    if (preloaded)
        blk = *preloaded;   // the caller's block is centred on pos already
    else
        tb_load<LANES>(tiles, tnx, tny, pos, rows, lane, blk);
    bool walking = live, was_bad = false, was_closed = false;   // the verdict is read off the two sticky flags after the loop
    const int maxbx = (tnx - 4) * 8, maxby = (tny - 4) * 8;
    // Inside the loop the walk lives in block coordinates (bits_tiles.h: lp, s1c); everything a step compares against is brought into them when the
    // block changes (differences of packed positions are exact modulo 2^32, so equality and the signed distance to the trigger survive).
    uint32_t base1 = tb_base1(blk), lp = pos - base1, s1c = tb_s1c(s);
    uint32_t p0b = pos0 - base1, p1b = pos1 - base1, tkb = tkey - base1;
    const uint32_t* rb = rows + lane - LANES;
    while (__any(walking)) {
        {
            // a side that is clamped to the image needs no margin: the border cannot leave the image
            const int lxm = (int)(lp & 0xFFFFu), ly = (int)(lp >> 16);   // lxm = x - bx - 1
            const bool near = (lxm < CHUNK && blk.bx > 0) || (lxm > 29 - CHUNK && blk.bx < maxbx) || (ly < 1 + CHUNK && blk.by > 0) ||
                              (ly > 30 - CHUNK && blk.by < maxby);
            // A closed 8-connected border that reaches Chebyshev distance d from its start has at least 2 d points (every step moves
            // at most one pixel in that metric, and the border returns): at 2 d >= nmax the size filter's verdict is known and the
            // walk ends here instead of after nmax steps.
            const uint32_t at = lp + base1;
            const int x = (int)(at & 0xFFFFu), y = (int)(at >> 16);
            {
                const int ax = abs(x - (int)(pos0 & 0xFFFFu)), ay = abs(y - (int)(pos0 >> 16));
                if (walking && 2u * (uint32_t)max(ax, ay) >= nmax) walking = false, was_bad = true;
            }
            if (PER_LANE_RELOAD ? (walking && near) : __any(walking && near)) {
                if (TB_DIRECTED)
                    tb_load_dir<LANES, CHUNK>(tiles, tnx, tny, at, tb_s_of(s1c), rows, lane, blk);
                else
                    tb_load<LANES>(tiles, tnx, tny, at, rows, lane, blk);
                base1 = tb_base1(blk), lp = at - base1;
                p0b = pos0 - base1, p1b = pos1 - base1, tkb = tkey - base1;
            }
        }
        walk_chunk<HOLE, LANES>(rb, base1, p0b, p1b, tkb, pos0, lim, lp, s1c, n, walking, was_bad, was_closed, ck_at);
    }
    const int res = was_bad ? WR_BAD : was_closed ? WR_CLOSED : WR_LIMIT;
    pos = lp + base1, s = tb_s_of(s1c);
    return res;
}

// A closed border that passes the size filter: descriptor slot and point range from the plane's own counters (a global
// counter would serialise every kept border). Returns the descriptor's pool offset for the checkpoints / points.
__device__ __forceinline__ bool keep_border(const WalkArgs& a, int plane, bool hole, uint32_t tkey, uint32_t pos0, uint32_t n, uint32_t nck_in_pool,
                                            uint32_t ck_off, uint32_t* pool_at) {
    const uint32_t slot = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_CDESC], 1u);
    uint32_t off = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_POOL], n + nck_in_pool);
    if (slot >= a.cap_cdesc) {
        flag_overflow(a.counters, a.trig_cnt, plane, ST_CDESC_OVERFLOW);
        return false;
    }
    bool ok = true;
    if (off + n + nck_in_pool > a.cap_pool) {
        flag_overflow(a.counters, a.trig_cnt, plane, ST_POOL_OVERFLOW);
        n = 0, ok = false;  // keeps the list consistent; a zero-length contour is ignored downstream
    }
    *pool_at = (uint32_t)plane * a.cap_pool + off;
    ContourDesc cd;
    cd.plane = plane, cd.x0 = (int16_t)(pos0 & 0xFFFFu), cd.y0 = (int16_t)(pos0 >> 16), cd.hole = hole ? 1 : 0, cd.n = (int)n;
    cd.key = (tkey >> 16) * (uint32_t)a.width + (tkey & 0xFFFFu);
    cd.pool_off = *pool_at + nck_in_pool;   // points follow the checkpoints
    cd.ck_off = ck_off;                     // 0xFFFFFFFF: checkpoints sit in front of the points in the pool
    a.cdesc[(size_t)plane * a.cap_cdesc + slot] = cd;
    return ok;
}

// Kernel 2a: one lane per start candidate, at most LEASH steps. Workgroups are dealt round-robin over the 8 XCDs, so the
// linear block id is unpacked such that all workgroups of one plane share an XCD and its L2 keeps that plane's tiles.
// The first half of a plane's workgroups follows outer candidates, the second half hole candidates. Before the walk a
// short probe follows the border the other way round: many false starts whose forward walk would take hundreds of steps
// to meet an earlier pixel are exposed within a few steps backwards. Walks that outlast the leash are queued with their
// state for kernel 2b; most walks (small borders, false starts) end here.
template <bool HOLE>
__device__ __forceinline__ void walk_short(const WalkArgs& a, int plane, int chunk, int nchunks, uint32_t* rows, uint32_t* ck0) {
    const uint32_t half = a.cap_trig / 2;
    const uint32_t ntrig = min(a.trig_cnt[plane * TRIG_CNT_STRIDE + (HOLE ? 1 : 0)], half);
    const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * a.tnx * a.tny;
    const uint2* list = a.trig + (size_t)plane * a.cap_trig + (HOLE ? half : 0);
    const int lane = threadIdx.x;
    const uint32_t nmax = (uint32_t)a.max_contour;
    const uint32_t lim = min((uint32_t)a.leash, nmax);
    for (uint32_t i0 = chunk * blockDim.x; i0 < ntrig; i0 += nchunks * blockDim.x) {
        const uint32_t i = i0 + lane;
        bool live = i < ntrig;
        const uint32_t tkey = live ? list[i].y : 0x00200020u;   // y << 16 | x of the scan transition
        const uint32_t pos0 = tkey - (HOLE ? 1u : 0u);
        TileBlock blk;
        tb_load<64>(tiles, a.tnx, a.tny, pos0, rows, lane, blk);
        uint32_t m = live ? tb_mask<64>(rows, lane, blk, pos0) : 0u;
        int s = live ? first_dir(m, HOLE ? 0 : 4) : -1;
        if (s < 0) live = false;  // isolated pixel: 1 point, never passes the size filter
        // ---- reverse probe (directions of the mirrored image; the block is not re-centred: PROBE steps stay inside it)
        {
            uint32_t rm = live ? tb_mask_mirror<64>(rows, lane, blk, pos0) : 0u;
            int rs = live ? first_dir(rm, HOLE ? 4 : 0) : -1;
            bool rlive = live && rs >= 0;
            uint32_t rpos = pos0;
            for (int it = 0; it < PROBE && __any(rlive); it++) {
                if (rlive) {
                    bool bad;
                    const int d = walk_step<HOLE, true>(rm, rs, rpos, tkey, pos0, &bad);
                    if (bad) {
                        live = false, rlive = false;
                    } else {
                        rpos += tb_dpos((4 - d) & 7);
                        rs = (d + 4) & 7;
                        rm = tb_mask_mirror<64>(rows, lane, blk, rpos);
                    }
                }
            }
        }
        const uint32_t pos1 = pos0 + tb_dpos(s & 7);
        uint32_t pos = pos0, n = 0;
        // all lanes of the wave step together; a lane that finished idles until the longest walk of the wave ends
        int res = walk_run<HOLE, 64>(tiles, a.tnx, a.tny, rows, lane, live, tkey, pos0, pos1, lim, nmax, pos, n, s,
                                     [&](uint32_t q) { return ck0 + q * 64 + lane; }, &blk);
        if (!live) res = WR_BAD;
        // ---- walks that outlast the leash join generation 1 with their state; their checkpoints move to a ring in HBM
        {
            bool longw = live && res == WR_LIMIT && n < nmax;
            uint32_t ring = 0;
            if (longw) {
                const uint32_t slot = atomicAdd(&a.ring_cnt[plane * TRIG_CNT_STRIDE + (HOLE ? 1 : 0)], 1u);
                if (slot < a.long_cap) {
                    ring = (uint32_t)(((size_t)plane * 2 + (HOLE ? 1 : 0)) * a.long_cap + slot);
                    uint32_t* ck = a.scratch + (size_t)ring * a.maxck;
                    for (uint32_t c = 0; c < n / CK; c++) ck[c] = ck0[c * 64 + lane];
                } else {
                    flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
                    longw = false;
                }
            }
            const unsigned long long bal = __ballot(longw);
            if (bal) {   // one atomic per wave on the list's counter
                uint32_t base = 0;
                const int gx = a.gen_nx > 1 ? (plane & (GEN_NX_MAX - 1)) : 0;
                if (lane == 0) base = atomicAdd(&a.gen_cnt[gen_cnt_index(HOLE ? 1 : 0, 1, gx)], (uint32_t)__popcll(bal));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (longw) {
                    const uint32_t at = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (at < a.gen_cap / a.gen_nx) {
                        const size_t li = gen_list_base(a, HOLE ? 1 : 0, 1, a.gen_nx > 1 ? (plane & (GEN_NX_MAX - 1)) : 0) + at;   // parity of generation 1
                        a.gen_state[li] = make_uint4(tkey, pos, pos1, n | ((uint32_t)s << 16));
                        a.gen_ring[li] = ring;
                    } else {
                        flag_overflow(a.counters, a.trig_cnt, plane, ST_TRIG_OVERFLOW);
                    }
                }
            }
        }
        if (!live || res != WR_CLOSED || n >= nmax) continue;
        if ((int)n <= a.min_contour) continue;
        const uint32_t ncp = (n + CK - 1) / CK;
        uint32_t at;
        if (keep_border(a, plane, HOLE, tkey, pos0, n, ncp, 0xFFFFFFFFu, &at)) {
            uint32_t* dst = (uint32_t*)(a.pool + at);
            for (uint32_t c = 0; c < ncp; c++) dst[c] = ck0[c * 64 + lane];
        }
    }
}

__global__ __launch_bounds__(64) void walker_kernel(WalkArgs a) {
    latency_bound_priority();
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int chunk = rest % WALK_BLOCKS, plane = (rest / WALK_BLOCKS) * 8 + xcd;
    if (plane >= a.nplanes) return;
    __shared__ uint32_t rows[TB_ROWS * 64];   // one 32x32-pixel block per lane
    __shared__ uint32_t ck0[(LEASH_MAX / CK + 1) * 64];
    if (chunk < WALK_BLOCKS / 2)
        walk_short<false>(a, plane, chunk, WALK_BLOCKS / 2, rows, ck0);
    else
        walk_short<true>(a, plane, chunk - WALK_BLOCKS / 2, WALK_BLOCKS / 2, rows, ck0);
}

// Kernel 2b, one launch per generation: the walks that are still open after the previous generation, from all planes, as
// one dense list per kind. Every wave takes 64 of them, walks at most gen_steps steps and appends the survivors (with
// their state) to the next generation's list, so wavefronts are full apart from the walks that end inside a generation;
// the generations grow from 64 steps (many walks alive) to thousands (a handful of very long borders).
// Round 4 measured the alternative "a wave owns 64 q walks and its lanes take over the next one when theirs ends" (lane utilisation of the generations
// 60 % -> 69 % / 80 % for q = 2 / 4, and the stage 0.64 -> 0.84 / 1.07 ms: every generation wave is resident at once, the time is the iterations of the
// longest walk times the latency of an iteration, not lane-step slots). Step counts and the A/B: profiles/r04_walker_steps.txt. Not kept.
template <bool HOLE>
__device__ __forceinline__ void walk_generation(const WalkArgs& a, int chunk, uint32_t* rows) {
    const int kind = HOLE ? 1 : 0, lane = threadIdx.x;
    // gen_blocks is a multiple of 8: workgroup b of this kind runs on XCD b % 8 and takes sublist b % 8 (gen_nx = 8) with the other gen_blocks / 8 - 1 of its XCD
    const int gx = a.gen_nx > 1 ? (chunk & (GEN_NX_MAX - 1)) : 0;
    const uint32_t sub = a.gen_nx > 1 ? (uint32_t)chunk >> 3 : (uint32_t)chunk, nsub = a.gen_nx > 1 ? (uint32_t)a.gen_blocks >> 3 : (uint32_t)a.gen_blocks;
    const uint32_t count = min(a.gen_cnt[gen_cnt_index(kind, a.gen, gx)], a.gen_cap / a.gen_nx);
    const size_t src = gen_list_base(a, kind, a.gen & 1, gx), dst = gen_list_base(a, kind, (a.gen + 1) & 1, gx);
    const uint32_t nmax = (uint32_t)a.max_contour;
    const size_t plane_tiles = (size_t)a.tnx * a.tny;
    for (uint32_t base = sub * 64u; base < count; base += nsub * 64u) {
        const bool live = base + lane < count;
        const uint4 st = live ? a.gen_state[src + base + lane] : make_uint4(0x00200021u, 0x00200020u, 0u, 0u);
        const uint32_t ring = live ? a.gen_ring[src + base + lane] : 0u;
        const uint32_t tkey = st.x, pos1 = st.z, pos0 = tkey - (HOLE ? 1u : 0u);
        uint32_t pos = st.y, n = st.w & 0xFFFFu;
        int s = (int)(st.w >> 16);
        const int plane = (int)(ring / (2u * a.long_cap));
        const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * plane_tiles;
        uint32_t* ck = a.scratch + (size_t)ring * a.maxck;
        const uint32_t lim = min(n + (uint32_t)a.gen_steps, nmax);
        const int res = walk_run<HOLE, 64>(tiles, a.tnx, a.tny, rows, lane, live, tkey, pos0, pos1, lim, nmax, pos, n, s, [&](uint32_t q) { return ck + q; });
        if (live && res == WR_CLOSED && n < nmax && (int)n > a.min_contour) {
            uint32_t at;
            keep_border(a, plane, HOLE, tkey, pos0, n, 0u, (uint32_t)((size_t)ring * a.maxck), &at);
        }
        const bool again = live && res == WR_LIMIT && n < nmax;
        const unsigned long long bal = __ballot(again);
        if (bal) {
            uint32_t at0 = 0;
            if (lane == 0) at0 = atomicAdd(&a.gen_cnt[gen_cnt_index(kind, a.gen + 1, gx)], (uint32_t)__popcll(bal));
            at0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)at0);
            if (again) {
                const size_t li = dst + at0 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));   // at most as many entries as this list had
                a.gen_state[li] = make_uint4(tkey, pos, pos1, n | ((uint32_t)s << 16));
                a.gen_ring[li] = ring;
            }
        }
    }
}

__global__ __launch_bounds__(64) void walker_long_kernel(WalkArgs a) {
    latency_bound_priority();
    __shared__ uint32_t rows[TB_ROWS * 64];   // one 32x32-pixel block per lane
    if ((int)blockIdx.x < a.gen_blocks)
        walk_generation<false>(a, blockIdx.x, rows);
    else
        walk_generation<true>(a, blockIdx.x - a.gen_blocks, rows);
}

size_t walk_scratch_words(int nplanes, const DetectParams& p, uint32_t long_cap) {
    return (size_t)((nplanes + 7) / 8) * 8 * 2 * long_cap * ((p.max_contour + CK - 1) / CK);
}

__global__ void snapshot_kernel(uint32_t* trig_cnt, int nplanes) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nplanes) trig_cnt[p * TRIG_CNT_STRIDE + TC_SNAP] = trig_cnt[p * TRIG_CNT_STRIDE + TC_CDESC];
}

// returns true if the late generations were forked to fk.side: the caller runs launch_contour_quads pass 1, waits for
// fk.joined on its stream and runs pass 2
bool launch_walkers(hipStream_t s, const WalkFork& fk, const FrameGeom& g, int nplanes, const DetectParams& p, const Buffers& b) {
    WalkArgs a;
    a.tiles = b.tiles, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height), a.nplanes = nplanes;
    a.trig = b.trig, a.trig_cnt = b.trig_cnt, a.cdesc = b.cdesc, a.counters = b.counters;
    a.cap_trig = b.cap_trig, a.cap_cdesc = b.cap_cdesc, a.cap_pool = b.cap_pool;
    a.width = g.width, a.height = g.height, a.min_contour = p.min_contour, a.max_contour = p.max_contour;
    a.scratch = b.walk_scratch, a.pool = b.pool, a.maxck = (p.max_contour + CK - 1) / CK;
    a.ring_cnt = b.ring_cnt, a.gen_cnt = b.gen_cnt;
    // generation lists: [2 kinds][2 parities][gen_cap] states (16 B) followed by the ring ids (4 B); a list can never
    // hold more walks than rings exist
    a.long_cap = b.long_cap;
    // one sublist per XCD when the batch gives every XCD planes to work on; a few planes (one frame per call with the walkers) keep one list for the whole chip
    a.gen_nx = (nplanes >= 64 && b.tune.gen_xcd) ? GEN_NX_MAX : 1;
    a.gen_cap = (uint32_t)((size_t)((nplanes + 7) / 8 * 8) * b.long_cap);
    a.gen_state = (uint4*)b.gen_buf;
    a.gen_ring = (uint32_t*)(a.gen_state + 4 * (size_t)a.gen_cap);
    a.gen = 0, a.gen_steps = 0;
    {
        const int v = b.tune.leash > 0 ? b.tune.leash : LEASH_DEFAULT;
        a.leash = (v >= CK && v <= LEASH_MAX && v % CK == 0) ? v : LEASH_DEFAULT;
    }
    const int planes8 = ((nplanes + 7) / 8) * 8;
    // a 64-thread workgroup per wave keeps the divergent walks of one wave from holding other waves' slots
    hipLaunchKernelGGL(walker_kernel, dim3(planes8 * WALK_BLOCKS), dim3(64), 0, s, a);
    if (fk.after_first) (void)hipEventRecord(fk.after_first, s);
    // Generations: 64-step ones while many walks are alive, then doubling. The late generations hold a handful of very long
    // walks and are pure latency (a border of n pixels is a chain of n dependent steps), so they run on the side stream
    // while the main stream already turns the borders found so far into quads (launch_contour_quads pass 1); the per-plane
    // descriptor counts at the fork are snapshotted for that.
    // schedule: ARUCOHIP_GENS="64,64,128,..." overrides for tuning; after the listed generations the length stays 1024
    int kSteps[GEN_MAX], nsched = 0;
    {
        for (int i = 0; i < b.tune.ngens && nsched < GEN_MAX; i++) {   // ARUCOHIP_GENS="64,64,128,..."; after the listed generations the length stays 1024
            const int v = b.tune.gens[i];
            if (v >= CHUNK && v % CHUNK == 0 && v % CK == 0) kSteps[nsched++] = v;
        }
        // round 2 (every lane re-centres its own block): few long generations beat many short ones; the side stream takes over
        // after 128 + 256 + 512 steps, borders of up to 960 points are in contour_quad's first pass (profiles/r02_kernel_experiments.txt)
        static const int kDefault[] = {128, 256, 512, 1024};
        if (nsched == 0)
            for (int v : kDefault) kSteps[nsched++] = v;
    }
    const int kForkAfter = b.tune.fork_after;
    int done = a.leash;
    bool forked = false;
    hipStream_t cur = s;
    for (int g = 1; g <= GEN_MAX && done < p.max_contour && RUN_STAGE(b.tune, 3); g++) {
        if (g == kForkAfter + 1 && fk.side) {
            hipLaunchKernelGGL(snapshot_kernel, dim3((nplanes + 255) / 256), dim3(256), 0, s, b.trig_cnt, nplanes);
            (void)hipEventRecord(fk.forked, s);
            (void)hipStreamWaitEvent(fk.side, fk.forked, 0);
            cur = fk.side, forked = true;
        }
        a.gen = g;
        a.gen_steps = g <= nsched ? kSteps[g - 1] : 1024;
        if (g == GEN_MAX) a.gen_steps = p.max_contour;   // whatever is left
        done += a.gen_steps;
        // enough workgroups that every wave-load of walks runs at once while many walks are alive (about a fifth of a
        // plane's ~1000 candidates reach generation 1), fewer for the thin late generations; surplus workgroups exit at once
        const int before = done - a.gen_steps;   // steps every walk of this generation has behind it
        const int per_plane_x16 = before < 200 ? 64 : before < 450 ? 40 : before < 1100 ? 24 : 4;   // walks per plane and kind / 4, rough upper bounds
        a.gen_blocks = (std::max(64, std::min(8192, (nplanes * per_plane_x16 * 4 + 63) / 64 / 2)) + 7) / 8 * 8;
        hipLaunchKernelGGL(walker_long_kernel, dim3(2 * a.gen_blocks), dim3(64), 0, cur, a);
    }
    if (forked) (void)hipEventRecord(fk.joined, fk.side);
    return forked;
}

// ---------------------------------------------------------------------------------------------
// Kernel 3: one wave per surviving border: emit points, approxPolyDP, convexity, min side -> Quad
// ---------------------------------------------------------------------------------------------
// Wave reductions on the DPP network (no LDS crossbar trips): butterfly inside each row of 16 lanes, then row
// broadcasts; the result is read from lane 63.
// max / min of the wave on the DPP network with the shuffle folded into the operation (v_max_u32_dpp: one instruction per step; written as
// "move with DPP, then max" the compiler keeps two). Rows that a row_bcast step does not address keep their value.
#define DPP_FOLD(op, v)                                                                                         \
    asm volatile("s_nop 4\n\t" /* the compiler does not see a DPP read here: cover the VGPR-write (2) and VALU-EXEC-write (5) wait states */ \
                 op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                            \
                 "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"             \
                 "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                 \
                 "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                      \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                    \
                 "s_nop 1"                                                                                      \
                 : "+v"(v))
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    DPP_FOLD("v_max_u32_dpp", v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    DPP_FOLD("v_min_u32_dpp", v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// argmax with first-maximum tie break: every lane brings its best (value, position), positions are unique; a lane whose elements are all zero
// (or that has none) brings (0, 0xFFFFFFFF), so a maximum of zero comes back with position 0xFFFFFFFF
__device__ __forceinline__ void wave_argmax_first(uint32_t val, uint32_t idx, uint32_t* max_val, uint32_t* first_idx) {
    const uint32_t mv = wave_max_u32(val);
    *max_val = mv;
    *first_idx = wave_min_u32(val == mv ? idx : 0xFFFFFFFFu);
}

// The same reductions for the two halves of a wave at once (round 4: two borders per wave, one per half): the butterfly inside the rows of 16 lanes and
// the first row broadcast leave the maximum of lanes 0..31 in row 1 and that of lanes 32..63 in row 3; no step reads across the halves.
#define DPP_FOLD_HALF(op, v)                                                                                    \
    asm volatile("s_nop 4\n\t"                                                                                  \
                 op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                            \
                 "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"             \
                 "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                 \
                 "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                      \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
                 "s_nop 1"                                                                                      \
                 : "+v"(v))
__device__ __forceinline__ uint32_t half_max_u32(uint32_t v, bool upper) {
    DPP_FOLD_HALF("v_max_u32_dpp", v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return upper ? hi : lo;
}
__device__ __forceinline__ uint32_t half_min_u32(uint32_t v, bool upper) {
    DPP_FOLD_HALF("v_min_u32_dpp", v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return upper ? hi : lo;
}
__device__ __forceinline__ void half_argmax_first(uint32_t val, uint32_t idx, bool upper, uint32_t* max_val, uint32_t* first_idx) {
    const uint32_t mv = half_max_u32(val, upper);
    *max_val = mv;
    *first_idx = half_min_u32(val == mv ? idx : 0xFFFFFFFFu, upper);
}

// The "farthest point" scans of approxPolyDP over the cyclic index range first, first + 1, ... (len entries, first < count, len <= count) as two
// linear runs (up to the end of the border, then from its beginning), so that no entry needs a wrap test; every lane keeps the first maximum of its
// own entries (only a strictly greater value replaces it, its entries come in scan order), the key is the scan position. A point is one 32-bit word
// x | y << 16: the difference to the anchor is one v_pk_sub_i16 and the squared distance / the cross product one v_dot2_i32_i16 (coordinates are
// below 2^14, so differences fit 16 bits and the sums 31).
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
template <bool CROSS>
__device__ __forceinline__ uint32_t scan_value(uint32_t pt, uint32_t anchor, uint32_t w) {
    const short2v d = as_s2(pt) - as_s2(anchor);
    if (!CROSS) return (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
    const int cr = __builtin_amdgcn_sdot2(d, as_s2(w), 0, false);
    return (uint32_t)max(cr, -cr);
}
template <bool CROSS, int STRIDE = WAVE>
__device__ __forceinline__ void scan_cyclic(const uint32_t* P32, int count, int first, int len, int lane, uint32_t anchor, uint32_t w, uint32_t& bd, uint32_t& bk) {
    bd = 0, bk = 0xFFFFFFFFu;
    const int lenA = min(len, count - first);
    const uint32_t* pa = P32 + first;
    for (int k = lane; k < lenA; k += STRIDE) {
        const uint32_t v = scan_value<CROSS>(pa[k], anchor, w);
        if (v > bd) bd = v, bk = (uint32_t)k;
    }
    const uint32_t* pb = P32 - lenA;
    for (int k = lenA + lane; k < len; k += STRIDE) {
        const uint32_t v = scan_value<CROSS>(pb[k], anchor, w);
        if (v > bd) bd = v, bk = (uint32_t)k;
    }
}

#ifndef EMIT_LANES_N
#define EMIT_LANES_N 64   // round 3: the whole wave emits. Round 2 kept 32 (less LDS, more waves: 0.58 against 0.67 ms one batch at a time); with the
                          // batches in flight the vector-instruction count is what counts: 446.3k / 447.6k against 443.2k / 442.5k fps, same box, alternating
#endif
constexpr int EMIT_LANES = EMIT_LANES_N;   // lanes that emit points at a time (each needs a 32x32 block in LDS)
#ifndef QP_LDS_N
#define QP_LDS_N 1024
#endif
constexpr int QP_LDS = QP_LDS_N;   // points of a border kept in LDS; longer borders are scanned in HBM (their pool range)

struct QuadArgs {
    const uint64_t* tiles;
    int tnx, tny;
    int from_pool;     // segment pipeline: the points are already in the pool
    const ContourDesc* cdesc;
    short2* pool;
    Quad* quads;
    uint32_t* counters;
    uint32_t* trig_cnt;
    const uint32_t* walk_scratch;   // checkpoint rings of the long walks
    uint32_t cap_cdesc;
    int cap_quads, nthr;
    int width, height;
    int pass;
    int qblocks, nplanes;   // workgroups per plane, planes
    int dual;               // two short borders per wave (ARUCOHIP_QUAD_DUAL, default on)
};

// The vertices approxPolyDP's recursion left (<= 8, in s_out) -> its clean-up pass, 4 vertices, cv::isContourConvex, min side -> Quad. One lane.
__device__ __forceinline__ void polygon_to_quad(const QuadArgs& a, const ContourDesc& cd, const uint32_t ci, short2* s_out, const int outn, const double eps) {
    int new_count = outn;
    const int cnt = outn;
    int p2 = cnt - 1;
    short2 start_pt = s_out[p2];
    if (++p2 >= cnt) p2 = 0;
    int wpos = p2;
    short2 pt = s_out[p2];
    if (++p2 >= cnt) p2 = 0;
    for (int i = 0; i < cnt && new_count > 2; i++) {
        short2 end_pt = s_out[p2];
        if (++p2 >= cnt) p2 = 0;
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = fabs((double)(pt.x - start_pt.x) * dy - (double)(pt.y - start_pt.y) * dx);
        double sip = (double)(pt.x - start_pt.x) * (double)(end_pt.x - pt.x) +
                     (double)(pt.y - start_pt.y) * (double)(end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            s_out[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            pt = s_out[p2];
            if (++p2 >= cnt) p2 = 0;
            i++;
            continue;
        }
        s_out[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    bool ok = new_count == 4;
    if (ok) {  // cv::isContourConvex on 4 int points
        short2 prev = s_out[2], cur = s_out[3];
        int dx0 = cur.x - prev.x, dy0 = cur.y - prev.y, orientation = 0;
        for (int i = 0; i < 4 && ok; i++) {
            prev = cur;
            cur = s_out[i];
            int dx = cur.x - prev.x, dy = cur.y - prev.y;
            int dxdy0 = dx * dy0, dydx0 = dy * dx0;
            orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
            if (orientation == 3) ok = false;
            dx0 = dx, dy0 = dy;
        }
    }
    if (ok) {  // minimum side > 10 px (intended form of markerdetector.cpp:542-552)
        int mind2 = 0x7FFFFFFF;
        for (int j = 0; j < 4; j++) {
            int dx = s_out[j].x - s_out[(j + 1) & 3].x, dy = s_out[j].y - s_out[(j + 1) & 3].y;
            mind2 = min(mind2, dx * dx + dy * dy);
        }
        ok = mind2 > 100;
    }
    if (ok) {
        const int frame = cd.plane / a.nthr, t = cd.plane - frame * a.nthr;
        uint32_t slot = atomicAdd(&a.counters[CNT_FIXED + frame], 1u);
        if (slot < (uint32_t)a.cap_quads) {
            Quad q;
            for (int j = 0; j < 4; j++) q.x[j] = s_out[j].x, q.y[j] = s_out[j].y;
            q.cdesc = (int)ci;
            q.key = ((uint32_t)t << 26) | (0x3FFFFFFu - cd.key);
            q.pad_ = 0;
            a.quads[(size_t)frame * a.cap_quads + slot] = q;
        } else {
            flag_overflow(a.counters, a.trig_cnt, cd.plane, ST_QUAD_OVERFLOW);
        }
    }
}

// One border -> at most one quad. P holds the border's points: LDS (LDSP, up to QP_LDS points) or the border's own pool range.
template <bool LDSP>
__device__ __forceinline__ void border_to_quad(const QuadArgs& a, const ContourDesc& cd, const uint32_t ci, short2* P, int (*s_stack)[2], short2* s_out,
                                               int& s_outn, uint32_t* rows) {
    const int lane = threadIdx.x;
    const int count = cd.n;
    // ---- points: already emitted (segment pipeline) or every lane resumes the walk at one checkpoint and records CK points
    if (a.from_pool) {
        if (LDSP)
            for (int i = lane; i < count; i += WAVE) P[i] = a.pool[cd.pool_off + i];
    } else {
        const uint64_t* tiles = a.tiles + (size_t)cd.plane * a.tnx * a.tny;
        const int ncp = (count + CK - 1) / CK;
        const uint32_t* ckp = cd.ck_off == 0xFFFFFFFFu ? (const uint32_t*)(a.pool + cd.pool_off) - ncp : a.walk_scratch + cd.ck_off;
        // Every lane resumes the walk at one checkpoint and records CK = 16 points. The neighbourhoods come from the lane's own
        // 32x32-pixel block in LDS (one load of eight 16-byte pieces instead of six dependent tile reads per step). ONE block serves the
        // whole stretch because both of its ends are known - this checkpoint and the next one (the border's start for the last stretch): a path
        // of L steps from A to B cannot leave their bounding box by more than (L - |dx|) / 2 columns or (L - |dy|) / 2 rows (a pixel further
        // out would need more than L steps to be reached from A and left towards B), so with the ring of neighbours the stretch spans at most
        // 19 pixels each way and a tile-aligned block of 32 that holds it always exists. (Until round 3 the block was placed by the heading at
        // the checkpoint and re-centred after 8 steps where needed - almost always for some lane: a second load per round, a fifth of the
        // kernel's vector instructions.)
        // EMIT_LANES lanes at a time: the kernel's speed follows its occupancy (12 KB of LDS per wave = 12 waves per CU measured
        // 0.67 ms, 24 KB 1.08 ms), and the blocks of a half wave cost 4 KB instead of 8
        uint32_t* P32 = (uint32_t*)P;
        const uint32_t* rb = rows + lane - EMIT_LANES;
        for (int k0 = 0; k0 < ncp; k0 += EMIT_LANES) {
            const int k = k0 + lane;
            if (lane >= EMIT_LANES || k >= ncp) continue;
            const uint32_t c = ckp[k];
            const uint32_t pos = (c & 0x3FFFu) | (((c >> 14) & 0x3FFFu) << 16);
            const int n0 = k * CK, n1 = min(n0 + CK, count);
            int bx, by;   // where the stretch ends
            if (k + 1 < ncp) {
                const uint32_t cn = ckp[k + 1];
                bx = (int)(cn & 0x3FFFu), by = (int)((cn >> 14) & 0x3FFFu);
            } else {
                bx = cd.x0, by = cd.y0;
            }
            const int ax = (int)(pos & 0xFFFFu), ay = (int)(pos >> 16), L = n1 - n0;
            const int ex = (L - abs(bx - ax)) >> 1, ey = (L - abs(by - ay)) >> 1;
            const int tx0 = min(max((min(ax, bx) - ex - 1) >> 3, 0), a.tnx - 4), ty0 = min(max((min(ay, by) - ey - 1) >> 3, 0), a.tny - 4);
            TileBlock blk;
            tb_load_at<EMIT_LANES>(tiles, a.tnx, tx0, ty0, rows, lane, blk);
            const uint32_t base1 = tb_base1(blk);
            uint32_t lp = pos - base1, s1c = tb_s1c((int)(c >> 28));
#pragma unroll
            for (int j = 0; j < CK; j++) {
                if (n0 + j < n1) {
                    P32[n0 + j] = lp + base1;
                    tb_step<EMIT_LANES>(rb, lp, s1c);
                }
            }
        }
    }
    __syncthreads();
    if (LDSP && !a.from_pool)
        for (int i = lane; i < count; i += WAVE) a.pool[cd.pool_off + i] = P[i];

    // ---- cv::approxPolyDP(closed), restated for a wavefront: every "farthest point" scan is a 64-lane argmax with
    // first-maximum tie break (lowest scan position), control flow is wave-uniform.
    double eps = (double)count * 0.05;
    eps *= eps;
    int pos = 0, rs_start = 0;
    bool le_eps = false;
    const uint32_t* PW = (const uint32_t*)P;
    for (int it = 0; it < 3; it++) {
        pos = (pos + rs_start) % count;
        uint32_t bd, bk, maxd, kmax;
        scan_cyclic<false>(PW, count, pos + 1 == count ? 0 : pos + 1, count - 1, lane, PW[pos], 0u, bd, bk);   // j = 1 .. count-1 at scan position j - 1
        wave_argmax_first(bd, bk, &maxd, &kmax);
        if (maxd > 0) rs_start = (int)kmax + 1;
        le_eps = (double)maxd <= eps;
    }
    int top = 0, outn = 0;
    bool reject = false;
    if (lane == 0) s_outn = 0;
    if (!le_eps) {
        int sl_start = pos % count;
        int sl_end = (rs_start + sl_start) % count;
        if (lane == 0) {
            s_stack[0][0] = sl_end, s_stack[0][1] = sl_start;   // right_slice
            s_stack[1][0] = sl_start, s_stack[1][1] = sl_end;   // slice
        }
        top = 2;
    } else {
        if (lane == 0) s_out[0] = P[pos];
        outn = 1;
    }
    __syncthreads();
    while (top > 0) {
        if (outn + top > 8) {  // cannot end as 4 vertices (clean-up removes at most every other vertex)
            reject = true;
            break;
        }
        --top;
        const int sl_start = s_stack[top][0], sl_end = s_stack[top][1];
        __syncthreads();
        const short2 ep = P[sl_end], sp = P[sl_start];
        int len = sl_end - sl_start;
        if (len <= 0) len += count;
        bool small;
        int split = 0;
        if (len > 1) {
            const int dx = ep.x - sp.x, dy = ep.y - sp.y;
            // |(pt.y - sp.y) * dx - (pt.x - sp.x) * dy| = |(pt - sp) . (-dy, dx)|
            const uint32_t w = ((uint32_t)(-dy) & 0xFFFFu) | ((uint32_t)dx << 16);
            uint32_t bd, bq, maxd_u, qmax;
            scan_cyclic<true>(PW, count, sl_start + 1 == count ? 0 : sl_start + 1, len - 1, lane, PW[sl_start], w, bd, bq);
            wave_argmax_first(bd, bq, &maxd_u, &qmax);
            double maxd = (double)maxd_u;
            int q = qmax == 0xFFFFFFFFu ? 0 : (int)qmax;
            split = sl_start + 1 + q;
            if (split >= count) split -= count;
            small = maxd * maxd <= eps * ((double)dx * (double)dx + (double)dy * (double)dy);
        } else {
            small = true;
        }
        if (small) {
            if (lane == 0) s_out[outn] = sp;
            outn++;
        } else {
            if (lane == 0) {
                s_stack[top][0] = split, s_stack[top][1] = sl_end;
                s_stack[top + 1][0] = sl_start, s_stack[top + 1][1] = split;
            }
            top += 2;
        }
        __syncthreads();
    }
    if (reject || outn < 4) return;
    if (lane == 0) polygon_to_quad(a, cd, ci, s_out, outn, eps);
}

// Round 4: TWO borders per wave, one per half (lanes 0..31 / 32..63), for borders of at most DUAL_MAX points - the median kept border has 460. One
// border per wave left half the emitting lanes idle (29 checkpoints for 64 lanes) and paid the fixed cost of every approxPolyDP iteration - stack
// traffic, two reductions on the DPP network, the fp64 test, the barriers; about 150 instructions against 10 per 64 points scanned - per border; here
// the two halves run the same instruction stream on their own border. Every quantity that was wave-uniform is uniform per half; `valid` is false
// for a half without a border. Same arithmetic, same scan order, same tie breaks as border_to_quad.
#ifndef DUAL_MAX_N
#define DUAL_MAX_N 512
#endif
constexpr int DUAL_MAX = DUAL_MAX_N;   // 1024 (with QP_LDS_N=2048) measured: see profiles/r04_experiments.txt
__device__ __forceinline__ void border_pair_to_quads(const QuadArgs& a, const ContourDesc& cd, const uint32_t ci, const bool valid, short2* Pbase, int (*s_stack)[16][2],
                                                     short2 (*s_out)[12], uint32_t* rows) {
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const bool upper = h != 0;
    short2* P = Pbase + h * DUAL_MAX;
    uint32_t* P32 = (uint32_t*)P;
    const int count = valid ? cd.n : 1;
    if (a.from_pool) {
        if (valid)
            for (int i = hl; i < count; i += 32) P[i] = a.pool[cd.pool_off + i];
    } else {
        // a lane per checkpoint (at most 32 per border): resume the walk there and record CK points from the lane's own block (border_to_quad)
        const uint64_t* tiles = a.tiles + (size_t)cd.plane * a.tnx * a.tny;
        const int ncp = (count + CK - 1) / CK;
        const uint32_t* ckp = cd.ck_off == 0xFFFFFFFFu ? (const uint32_t*)(a.pool + cd.pool_off) - ncp : a.walk_scratch + cd.ck_off;
        const uint32_t* rb = rows + lane - 64;
        for (int k = hl; k < (DUAL_MAX + CK - 1) / CK; k += 32) {   // one round for borders of up to 512 points
            if (!(valid && k < ncp)) continue;
            const uint32_t c = ckp[k];
            const uint32_t pos = (c & 0x3FFFu) | (((c >> 14) & 0x3FFFu) << 16);
            const int n0 = k * CK, n1 = min(n0 + CK, count);
            int bx, by;   // where the stretch ends
            if (k + 1 < ncp) {
                const uint32_t cn = ckp[k + 1];
                bx = (int)(cn & 0x3FFFu), by = (int)((cn >> 14) & 0x3FFFu);
            } else {
                bx = cd.x0, by = cd.y0;
            }
            const int ax = (int)(pos & 0xFFFFu), ay = (int)(pos >> 16), L = n1 - n0;
            const int ex = (L - abs(bx - ax)) >> 1, ey = (L - abs(by - ay)) >> 1;
            const int tx0 = min(max((min(ax, bx) - ex - 1) >> 3, 0), a.tnx - 4), ty0 = min(max((min(ay, by) - ey - 1) >> 3, 0), a.tny - 4);
            TileBlock blk;
            tb_load_at<64>(tiles, a.tnx, tx0, ty0, rows, lane, blk);
            const uint32_t base1 = tb_base1(blk);
            uint32_t lp = pos - base1, s1c = tb_s1c((int)(c >> 28));
#pragma unroll
            for (int j = 0; j < CK; j++) {
                if (n0 + j < n1) {
                    P32[n0 + j] = lp + base1;
                    tb_step<64>(rb, lp, s1c);
                }
            }
        }
    }
    __syncthreads();
    if (!a.from_pool && valid)
        for (int i = hl; i < count; i += 32) a.pool[cd.pool_off + i] = P[i];

    // ---- cv::approxPolyDP(closed) per half
    double eps = (double)count * 0.05;
    eps *= eps;
    int pos = 0, rs_start = 0;
    bool le_eps = false;
    const uint32_t* PW = (const uint32_t*)P;
    for (int it = 0; it < 3; it++) {
        pos = (pos + rs_start) % count;
        uint32_t bd, bk, maxd, kmax;
        scan_cyclic<false, 32>(PW, count, pos + 1 == count ? 0 : pos + 1, count - 1, hl, PW[pos], 0u, bd, bk);
        half_argmax_first(bd, bk, upper, &maxd, &kmax);
        if (maxd > 0) rs_start = (int)kmax + 1;
        le_eps = (double)maxd <= eps;
    }
    int top = 0, outn = 0;
    bool reject = !valid;
    if (!le_eps) {
        const int sl_start = pos % count;
        const int sl_end = (rs_start + sl_start) % count;
        if (hl == 0) {
            s_stack[h][0][0] = sl_end, s_stack[h][0][1] = sl_start;   // right_slice
            s_stack[h][1][0] = sl_start, s_stack[h][1][1] = sl_end;   // slice
        }
        top = 2;
    } else {
        if (hl == 0) s_out[h][0] = P[pos];
        outn = 1;
    }
    __syncthreads();
    for (;;) {
        bool act = !reject && top > 0;
        if (act && outn + top > 8) reject = true, act = false;   // cannot end as 4 vertices (clean-up removes at most every other vertex)
        if (!__any(act)) break;
        int sl_start = 0, sl_end = 0, len = 0;
        short2 ep = make_short2(0, 0), sp = make_short2(0, 0);
        uint32_t bd = 0, bq = 0xFFFFFFFFu;
        int dx = 0, dy = 0;
        if (act) {
            --top;
            sl_start = s_stack[h][top][0], sl_end = s_stack[h][top][1];
            ep = P[sl_end], sp = P[sl_start];
            len = sl_end - sl_start;
            if (len <= 0) len += count;
            if (len > 1) {
                dx = ep.x - sp.x, dy = ep.y - sp.y;
                const uint32_t w = ((uint32_t)(-dy) & 0xFFFFu) | ((uint32_t)dx << 16);
                scan_cyclic<true, 32>(PW, count, sl_start + 1 == count ? 0 : sl_start + 1, len - 1, hl, PW[sl_start], w, bd, bq);
            }
        }
        __syncthreads();
        uint32_t maxd_u, qmax;
        half_argmax_first(bd, bq, upper, &maxd_u, &qmax);   // every lane takes part: the DPP steps never run under a half's mask
        if (act) {
            bool small = true;
            int split = 0;
            if (len > 1) {
                const double maxd = (double)maxd_u;
                const int q = qmax == 0xFFFFFFFFu ? 0 : (int)qmax;
                split = sl_start + 1 + q;
                if (split >= count) split -= count;
                small = maxd * maxd <= eps * ((double)dx * (double)dx + (double)dy * (double)dy);
            }
            if (small) {
                if (hl == 0) s_out[h][outn] = sp;
                outn++;
            } else {
                if (hl == 0) {
                    s_stack[h][top][0] = split, s_stack[h][top][1] = sl_end;
                    s_stack[h][top + 1][0] = sl_start, s_stack[h][top + 1][1] = split;
                }
                top += 2;
            }
        }
        __syncthreads();
    }
    if (hl == 0 && !reject && outn >= 4) polygon_to_quad(a, cd, ci, s_out[h], outn, eps);
}

__global__ __launch_bounds__(64) void contour_quad_kernel(QuadArgs a) {
    throughput_bound_priority();
    __shared__ __align__(16) short2 Plds[QP_LDS];   // contour points (two borders of up to DUAL_MAX points, or one of up to QP_LDS)
    __shared__ uint32_t rows[TB_ROWS * EMIT_LANES];  // one 32x32-pixel block per emitting lane
    __shared__ int s_stack[2][16][2];
    __shared__ short2 s_out[2][12];
    __shared__ int s_outn;
    static_assert(QP_LDS >= 2 * DUAL_MAX && EMIT_LANES == 64, "the two-borders-per-wave path uses the whole point buffer and a block per lane");
    // 1-D grid dealt round-robin over the 8 XCDs: all workgroups of a plane land on one XCD, whose L2 then serves the plane's tiles,
    // descriptors and checkpoints to all of them (same unpacking as walker_kernel)
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int chunk = rest % a.qblocks, plane = (rest / a.qblocks) * 8 + xcd;
    if (plane >= a.nplanes) return;
    // pass 0: all borders; pass 1: those that existed when the late walker generations were forked; pass 2: the rest
    const uint32_t nall = min(a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_CDESC], a.cap_cdesc);
    const uint32_t nsnap = min(a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_SNAP], a.cap_cdesc);
    const uint32_t lo = a.pass == 2 ? nsnap : 0u, ncd = a.pass == 1 ? nsnap : nall;
    const bool upper = threadIdx.x >= 32;
    ContourDesc pend;   // a short border waiting for a partner
    uint32_t pend_ci = 0;
    bool has_pend = false;
    for (uint32_t cslot = lo + (uint32_t)chunk; cslot < ncd; cslot += (uint32_t)a.qblocks) {
        const uint32_t ci = (uint32_t)plane * a.cap_cdesc + cslot;
        const ContourDesc cd = a.cdesc[ci];
        if (cd.n <= 0) continue;
        if (a.dual && cd.n <= DUAL_MAX) {
            if (!has_pend) {
                pend = cd, pend_ci = ci, has_pend = true;
                continue;
            }
            __syncthreads();
            border_pair_to_quads(a, upper ? cd : pend, upper ? ci : pend_ci, true, Plds, s_stack, s_out, rows);
            has_pend = false;
            continue;
        }
        __syncthreads();
        if (cd.n <= QP_LDS)
            border_to_quad<true>(a, cd, ci, Plds, s_stack[0], s_out[0], s_outn, rows);
        else
            border_to_quad<false>(a, cd, ci, a.pool + cd.pool_off, s_stack[0], s_out[0], s_outn, rows);
    }
    if (has_pend) {   // no partner came: all 64 lanes on the one border
        __syncthreads();
        border_to_quad<true>(a, pend, pend_ci, Plds, s_stack[0], s_out[0], s_outn, rows);
    }
}

void launch_contour_quads(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, int pass) {
    QuadArgs a;
    a.pass = pass;
    a.tiles = b.tiles, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height), a.from_pool = b.seg_mode, a.cdesc = b.cdesc, a.pool = b.pool, a.quads = b.quads, a.counters = b.counters;
    a.cap_cdesc = b.cap_cdesc, a.cap_quads = b.cap_quads, a.nthr = p.nthr, a.width = g.width, a.height = g.height;
    a.trig_cnt = b.trig_cnt, a.walk_scratch = b.walk_scratch;
    a.nplanes = nframes * p.nthr;
    a.dual = b.tune.quad_dual && a.nplanes > 8;   // a single frame has a wave per kept border anyway: all 64 lanes on one border are faster there
    // a handful of planes (one detect() per frame): a wave per kept border instead of a few waves that take the borders one after the other
    const int qb = a.nplanes <= 2 ? 128 : a.nplanes <= 8 ? 48 : b.tune.quad_blocks;
    a.qblocks = pass == 2 ? std::max(1, qb / 2) : qb;
    const int planes8 = ((a.nplanes + 7) / 8) * 8;
    hipLaunchKernelGGL(contour_quad_kernel, dim3(planes8 * a.qblocks), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// Kernel 4: per frame — reference order, orientation, near-duplicate removal  (markerdetector.cpp:562-627)
// ---------------------------------------------------------------------------------------------
struct FrameArgs {
    double* iM;           // [cap_flat][9] inverse homographies of the flat candidate list (round 3: solved here, one candidate per lane)
    int ws;               // patch size of MarkerDetector::warp
    const Quad* quads;
    Cand* cands;
    uint32_t* cand_list;
    uint32_t cap_flat;
    int32_t* ncands;
    uint32_t* counters;
    uint32_t* trig_cnt;   // per-plane counter lines: the frame's first plane takes the frame-level overflow bits
    int nthr;
    int cap_quads, cap_cands;
};

constexpr int MAXQ = 512;

__device__ __forceinline__ float quad_perimeter_i(const int16_t* x, const int16_t* y) {
    float sum = 0;
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        double dx = (double)(x[i] - x[j]), dy = (double)(y[i] - y[j]);
        sum = (float)((double)sum + sqrt(dx * dx + dy * dy));
    }
    return sum;
}

template <int HG>   // lanes that solve their candidate's homography at a time (decode_device.h): 16 for batches, 64 for one frame per call
__global__ __launch_bounds__(64) void frame_candidates_kernel(FrameArgs a) {
    latency_bound_priority();
    // one LDS region, two lives: the quads of the frame while they are ranked and thinned out, then the 8x8 systems of the homography solve
    // (decode_device.h) of up to 64 candidates at a time
    constexpr int PHASE1_BYTES = MAXQ * (8 + 8 + 4 + 4 + 2 + 1 + 1);
    constexpr int BIG_DOUBLES = homography_lds_doubles<HG>() > (PHASE1_BYTES + 7) / 8 ? homography_lds_doubles<HG>() : (PHASE1_BYTES + 7) / 8;
    __shared__ double s_big[BIG_DOUBLES];
    int16_t(*sx)[4] = (int16_t(*)[4])s_big;
    int16_t(*sy)[4] = sx + MAXQ;
    int* s_cdesc = (int*)(sy + MAXQ);
    float* s_perim = (float*)(s_cdesc + MAXQ);
    int16_t* s_map = (int16_t*)(s_perim + MAXQ);      // candidate k of the frame = ranked quad s_map[k]
    uint8_t* s_swapped = (uint8_t*)(s_map + MAXQ);
    uint8_t* s_rem = s_swapped + MAXQ;
    __shared__ int s_n;
    __shared__ uint32_t s_base;
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int nq = min((int)a.counters[CNT_FIXED + frame], min(a.cap_quads, MAXQ));
    const Quad* Q = a.quads + (size_t)frame * a.cap_quads;
    // rank by key (keys are unique: one border per scan transition)
    for (int i = lane; i < nq; i += WAVE) {
        const uint32_t key = Q[i].key;
        int rank = 0;
        for (int j = 0; j < nq; j++) rank += Q[j].key < key;
        Quad q = Q[i];
        // orientation: cross((c1-c0),(c2-c0)) < 0 -> swap corners 1 and 3
        int d1x = q.x[1] - q.x[0], d1y = q.y[1] - q.y[0], d2x = q.x[2] - q.x[0], d2y = q.y[2] - q.y[0];
        float o = ((float)d1x * (float)d2y) - ((float)d1y * (float)d2x);
        bool sw = o < 0.0f;
        if (sw) {
            int16_t tx = q.x[1], ty = q.y[1];
            q.x[1] = q.x[3], q.y[1] = q.y[3];
            q.x[3] = tx, q.y[3] = ty;
        }
        for (int k = 0; k < 4; k++) sx[rank][k] = q.x[k], sy[rank][k] = q.y[k];
        s_cdesc[rank] = q.cdesc;
        s_swapped[rank] = sw;
        s_rem[rank] = 0;
        s_perim[rank] = quad_perimeter_i(q.x, q.y);
    }
    __syncthreads();
    // pairs (i<j) whose four same-index corners are all closer than 6 px: drop the smaller perimeter
    for (int i = 0; i < nq; i++) {
        for (int j = i + 1 + lane; j < nq; j += WAVE) {
            bool near = true;
            for (int c = 0; c < 4; c++) {
                int dx = sx[i][c] - sx[j][c], dy = sy[i][c] - sy[j][c];
                near = near && (dx * dx + dy * dy < 36);
            }
            if (near) {
                if (s_perim[i] > s_perim[j])
                    s_rem[j] = 1;
                else
                    s_rem[i] = 1;
            }
        }
    }
    __syncthreads();
    if (lane == 0) {
        int n = 0;
        Cand* C = a.cands + (size_t)frame * a.cap_cands;
        for (int i = 0; i < nq; i++) {
            if (s_rem[i]) continue;
            if (n >= a.cap_cands) {
                flag_overflow(a.counters, a.trig_cnt, frame * a.nthr, ST_CAND_OVERFLOW);
                break;
            }
            Cand c;
            for (int k = 0; k < 4; k++) {
                c.c[2 * k] = (float)sx[i][k], c.c[2 * k + 1] = (float)sy[i][k];
                c.qx[k] = sx[i][k], c.qy[k] = sy[i][k];
            }
            c.cdesc = s_cdesc[i], c.swapped = s_swapped[i], c.id = -1, c.nrot = 0;
            s_map[n] = (int16_t)i;
            C[n++] = c;
        }
        a.ncands[frame] = n;
        // flat list for the decode kernels (order across frames is irrelevant)
        uint32_t base = 0;
        if (n > 0) {
            base = atomicAdd(&a.counters[CNT_NCAND], (uint32_t)n);
            for (int i = 0; i < n; i++) {
                if (base + i < a.cap_flat)
                    a.cand_list[base + i] = ((uint32_t)frame << 16) | (uint32_t)i;
                else
                    flag_overflow(a.counters, a.trig_cnt, frame * a.nthr, ST_CAND_OVERFLOW);
            }
        }
        s_n = n, s_base = base;
    }
    __syncthreads();
    // ---- MarkerDetector::warp's getPerspectiveTransform, inverted: one candidate per lane, 64 at a time. A lane first takes the corners
    // of ALL its candidates into registers (at most MAXQ / 64 of them): the solve reuses the LDS they sit in.
    const int n = s_n;
    const uint32_t base = s_base;
    constexpr int PER_LANE = MAXQ / WAVE;
    int16_t qx[PER_LANE][4], qy[PER_LANE][4];
#pragma unroll
    for (int r = 0; r < PER_LANE; r++) {
        const int k = r * WAVE + lane;
        if (k < n) {
            const int i = s_map[k];
            for (int c = 0; c < 4; c++) qx[r][c] = sx[i][c], qy[r][c] = sy[i][c];
        } else {
            for (int c = 0; c < 4; c++) qx[r][c] = 0, qy[r][c] = 0;
        }
    }
    __syncthreads();
    const LaneMatT<HG> A{s_big, lane}, b{s_big + 64 * HG, lane};
#pragma unroll
    for (int r = 0; r < PER_LANE; r++) {
        if (r * WAVE >= n) break;                       // wave-uniform
        const int k = r * WAVE + lane;
        for (int g = 0; g < WAVE / HG; g++) {   // one group of lanes at a time: they share the LDS the systems sit in
            if (r * WAVE + g * HG >= n) break;
            if (lane / HG == g && k < n && base + (uint32_t)k < a.cap_flat)
                homography_lane(qx[r], qy[r], a.ws, A, b, a.iM + (size_t)(base + (uint32_t)k) * 9);
        }
    }
}

void launch_frame_candidates(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b) {
    FrameArgs a;
    a.quads = b.quads, a.cands = b.cands, a.ncands = b.ncands, a.counters = b.counters, a.trig_cnt = b.trig_cnt, a.nthr = p.nthr;
    a.cand_list = b.cand_list, a.cap_flat = b.cap_flat;
    a.cap_quads = b.cap_quads, a.cap_cands = b.cap_cands;
    a.iM = b.iM, a.ws = p.warp_size;
    if (nframes <= 2)
        hipLaunchKernelGGL(frame_candidates_kernel<64>, dim3(nframes), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(frame_candidates_kernel<HOMOGRAPHY_GROUP>, dim3(nframes), dim3(64), 0, s, a);
}

}  // namespace ah
