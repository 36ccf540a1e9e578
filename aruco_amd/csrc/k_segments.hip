// Parallel border extraction ("waypoint segments") — replaces the sequential scan of cv::findContours
// (/root/reference/src/markerdetector.cpp:511) without long sequential walks.
//
// A border, as OpenCV follows it, is a cyclic sequence of VISITS (pixel p, direction s towards the previous pixel);
// during a visit the follower examines, counter-clockwise from s+1, the clear neighbours up to the next set one. Every
// CRACK (set pixel p, clear 4-neighbour in direction e) is examined in exactly one visit of exactly one border, and that
// visit's state is a function of the 3x3 neighbourhood alone: s = first set neighbour clockwise from e.
//
//  1. candidates_kernel (k_contours.hip) emits WAYPOINTS: the cracks on a sparse grid (W/E cracks on rows y % S == 0, N/S cracks on columns
//     x % S == 0) plus every crack that can start a border (the 3x3 start rule). Every border that spans more than S rows
//     or columns crosses a grid line and therefore carries waypoints every few pixels.
//  2. segment_kernel (one lane per waypoint): follow the border from the waypoint's visit until the next waypoint crack is
//     examined (a handful of steps). Records the next waypoint's key, the number of visits moved and the smallest
//     start key met (W crack: pos(p); E crack: pos(p)+1 — the candidates for OpenCV's scan transition).
//  3. link_kernel: next key -> node index through a per-plane hash table.
//  4. cycle_kernel (one lane per start-candidate waypoint that passes the run rule): hop from node to node around the
//     cycle, summing lengths; drop out when a smaller start key exists on the cycle or the border exceeds the size filter.
//     The survivor of a cycle is OpenCV's start; a second lap stamps every node with its offset in the output.
//  5. emit_kernel (one lane per stamped node): re-follow the node's few visits and write the points at their final
//     positions. Point order, start pixel and direction equal cv::findContours(RETR_LIST, CHAIN_APPROX_NONE).
#include "bits_tiles.h"
#include "internal.h"

namespace ah {

// first set neighbour clockwise from direction e (exclusive); mask must be non-zero
__device__ __forceinline__ int cw_first(uint32_t m, int e) {
    const uint32_t r = ((m | (m << 8)) >> e) & 0xFFu;   // bit t <-> direction e + t
    return (e + (31 - __builtin_clz(r))) & 7;            // highest set bit = first one met going clockwise from e
}
// first set neighbour counter-clockwise from direction s (exclusive); *k = number of clear neighbours passed
__device__ __forceinline__ int ccw_first(uint32_t m, int s, int* k) {
    const uint32_t sh = (uint32_t)(s + 1) & 7u;
    const uint32_t rot = ((m | (m << 8)) >> sh) & 0xFFu;
    *k = __builtin_ctz(rot | 0x100u);
    return (int)((sh + (uint32_t)*k) & 7u);
}

struct SegArgs {
    const uint64_t* tiles;
    int tnx, tny, width, height, nplanes;
    int grid_mask;
    const uint2* raw;           // waypoints {cand, key}
    const uint32_t* raw_cnt;
    uint32_t cap_raw;           // nodes per plane
    // node records [P][cap_raw]: x = key of the next waypoint (later its node index), y = smallest start key among the
    // cracks examined in this segment, z = visits moved | flag << 16 (flag 1: start candidate after the run rule, 2: dead)
    uint4* node;
    // stamp of the smallest start key that lapped over the node: key(30) << 34 | start node(20) << 14 | visits from the start(14)
    unsigned long long* stamp;
    uint32_t* hash;             // [P][hash_size] node index by key
    uint32_t hash_mask;
    ContourDesc* cdesc;
    short2* pool;
    uint32_t* counters;
    uint32_t cap_cdesc, cap_pool;   // per plane
    uint32_t* trig_cnt;             // per-plane counter lines (TC_CDESC, TC_POOL)
    int min_contour, max_contour;
    int seg_chunks, cyc_chunks;     // workgroups per plane of the per-node kernels (256 threads) / of the lap kernel (64 threads)
    uint4* skipn;                   // non-null: laps take SKIP segments per hop (one frame per call)
};

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr unsigned long long NONE64 = ~0ull;

// 1-D grids: workgroups are dealt round-robin over the 8 XCDs; unpack the block id so that all workgroups of one plane
// share an XCD (its L2 then holds that plane's bit image, node records and hash table).
__device__ __forceinline__ bool plane_of_block(int nplanes, int chunks, int* plane, int* chunk) {
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    *chunk = rest % chunks;
    *plane = (rest / chunks) * 8 + xcd;
    return *plane < nplanes;
}
// Workgroups per plane: 16 x 256 threads in the per-node kernels and 32 x 64 in the lap kernel when a batch of planes fills the chip. A single
// frame (the reference's call shape: one detect() per frame) has a few thousand waypoints and nothing else to run: with 16 workgroups every
// thread looped over several waypoints one after the other (segment_kernel 119 us of a 440 us call, round 3); it gets enough workgroups that
// every waypoint has a lane of its own.
static void seg_chunks_for(int nplanes, int* seg, int* cyc) {
    *seg = nplanes <= 2 ? 128 : nplanes <= 8 ? 48 : 16;
    *cyc = nplanes <= 2 ? 256 : nplanes <= 8 ? 96 : 32;
}

__device__ __forceinline__ uint32_t hash_key(uint32_t key, uint32_t mask) { return (key * 2654435761u >> 7) & mask; }

// run rule for a start-candidate crack (same test as candidates_kernel in k_contours.hip)
__device__ __forceinline__ bool run_rule(const uint64_t* __restrict__ tiles, int tnx, uint32_t pos, int e) {
    const int hole = e == 0;
    const uint32_t zpos = pos + (hole ? 1u : 0u);            // outer: the pixel itself; hole: the clear pixel right of p
    const int x = (int)(zpos & 0xFFFFu), y = (int)(zpos >> 16);
    int avail, avail_up;
    const uint64_t mid = tb_row64(tiles, tnx, x, y, &avail), up = tb_row64(tiles, tnx, x, y - 1, &avail_up);
    if (!hole) {
        int L = (~mid) ? __builtin_ctzll(~mid) : 64;
        L = min(L, avail);
        const int hi = min(L, avail - 1);
        const uint64_t mm = hi >= 2 ? ((hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~3ull) : 0ull;
        return (up & mm) == 0;
    }
    int L = mid ? __builtin_ctzll(mid) : 64;
    L = min(L, avail);
    const int hi = min(L - 1, avail - 1);
    const uint64_t mm = hi >= 1 ? ((hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~1ull) : 0ull;
    return (~up & mm) == 0;
}

// Is crack (pos, direction e) a waypoint? Grid cracks always; start-candidate cracks (3x3 rule of kernel 1) only when
// they also pass the run rule — the segment kernel discards the other candidates kernel 1 emitted, with this same test.
__device__ __forceinline__ bool on_grid(uint32_t pos, int e, int grid_mask) {
    return (e & 2) ? ((pos & 0xFFFFu) & grid_mask) == 0 : ((pos >> 16) & grid_mask) == 0;   // N,S: column; W,E: row
}
// The run rule from the lane's own 32x32 block in LDS (round 4): the thresholded image is made of 3-pixel bands, a run almost always ends within the block's
// columns and the test is two row words; only a run that leaves the block goes back to the tiles in memory (nine tile reads per row - inside the step
// loop those reads were what segment_kernel spent its time on: 116 us of a 430 us single-frame call). *decided = false: the block cannot tell.
template <int LANES>
__device__ __forceinline__ bool run_rule_block(const uint32_t* rows, int lane, const TileBlock& blk, uint32_t pos, int e, bool* decided) {
    const int hole = e == 0;
    const uint32_t zpos = pos + (hole ? 1u : 0u);
    const int lx = (int)(zpos & 0xFFFFu) - blk.bx, ly = (int)(zpos >> 16) - blk.by;
    *decided = false;
    if (lx < 0 || lx > 31 || ly < 1 || ly > 31) return true;
    const uint32_t mid = rows[ly * LANES + lane] >> lx, up = rows[(ly - 1) * LANES + lane] >> lx;
    const int avail = 32 - lx;
    // a blocker inside the block's columns settles the rule whatever the run does beyond them (the interior of a hole is a clear run of tens of
    // pixels - with a clear pixel above it after one or two); without one the run has to end inside the block
    if (!hole) {
        const int L = (~mid) ? __builtin_ctz(~mid) : 32;          // run of set pixels starting at x
        const int hi = min(L, avail - 1);                          // blockers: row above, columns x+2 .. x+L
        const uint32_t mm = hi >= 2 ? (((2u << hi) - 1u) & ~3u) : 0u;
        if (up & mm) return *decided = true, false;
        if (L >= avail) return true;                               // no blocker so far and the run leaves the block: the tiles decide
        *decided = true;
        return true;
    }
    const int L = mid ? __builtin_ctz(mid) : 32;                   // run of clear pixels starting at x
    const int hi = min(L - 1, avail - 1);                          // row above, columns x+1 .. x+L-1 must all be set
    const uint32_t mm = hi >= 1 ? (((2u << hi) - 1u) & ~1u) : 0u;
    if (~up & mm) return *decided = true, false;
    if (L >= avail) return true;
    *decided = true;
    return true;
}
template <int LANES>
__device__ __forceinline__ bool run_rule_any(const uint64_t* tiles, int tnx, const uint32_t* rows, int lane, const TileBlock& blk, uint32_t pos, int e) {
    bool decided;
    const bool r = run_rule_block<LANES>(rows, lane, blk, pos, e, &decided);
    return decided ? r : run_rule(tiles, tnx, pos, e);
}
template <int LANES>
__device__ __forceinline__ bool is_waypoint(const uint64_t* tiles, int tnx, const uint32_t* rows, int lane, const TileBlock& blk, uint32_t pos, int e, uint32_t m,
                                            int grid_mask, int width) {
    if (on_grid(pos, e, grid_mask)) return true;
    if (e == 4 && (m & 0x1Eu) == 0) return run_rule_any<LANES>(tiles, tnx, rows, lane, blk, pos, 4);                                               // outer start: W,NW,N,NE clear
    if (e == 0 && ((m >> 1) & 1u) && (int)(pos & 0xFFFFu) + 1 <= width - 2) return run_rule_any<LANES>(tiles, tnx, rows, lane, blk, pos, 0);    // hole start: z inside, N(z) set
    return false;
}

// Kernel S2: one lane per waypoint; the lanes of a wave step together and re-centre their 32x32 blocks together.
__global__ __launch_bounds__(256) void segment_kernel(SegArgs a) {
    __shared__ uint32_t rows[TB_ROWS * 256];
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.seg_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * a.tnx * a.tny;
    const size_t nb = (size_t)plane * a.cap_raw;
    uint32_t* hash = a.hash + (size_t)plane * (a.hash_mask + 1);
    const int tid = threadIdx.x;
    // a border that avoids the grid is confined to one S x S cell: at most 4 visits per pixel
    const int max_steps = 4 * (a.grid_mask + 1) * (a.grid_mask + 1) + 8;
    for (uint32_t i0 = chunk * blockDim.x; i0 < n; i0 += a.seg_chunks * blockDim.x) {
        const uint32_t i = i0 + tid;
        bool live = i < n;
        const uint2 rec = live ? a.raw[nb + i] : make_uint2(0u, (0x00010001u << 2) | 2u);
        const uint32_t key = rec.y, pos0 = key >> 2;
        const int e0 = (int)(key & 3u) * 2;
        TileBlock blk;
        tb_load<256>(tiles, a.tnx, a.tny, pos0, rows, tid, blk);
        uint32_t m = tb_mask<256>(rows, tid, blk, pos0);
        uint8_t flag = 0;
        if (live) {
            a.stamp[nb + i] = NONE64;
            if (rec.x & 1u) flag = run_rule_any<256>(tiles, a.tnx, rows, tid, blk, pos0, e0) ? 1 : 0;
            // not a node: isolated pixel (a one-point border, never kept) or a start candidate the run rule rejects off the grid
            if (m == 0 || (!flag && !on_grid(pos0, e0, a.grid_mask))) {
                a.node[nb + i] = make_uint4(key, NONE32, 2u << 16, NONE32);
                live = false;
            } else {
                // publish this node under its key
                for (uint32_t slot = hash_key(key, a.hash_mask);; slot = (slot + 1) & a.hash_mask)
                    if (atomicCAS(&hash[slot], NONE32, i) == NONE32) break;
            }
        }
        const bool is_node = live;
        int s = live ? cw_first(m, e0) : 0;
        uint32_t pos = pos0, mn = NONE32, len = 0, next_key = NONE32;
        int from = e0;   // cracks after direction `from` (counter-clockwise) are still to come in this visit
        int step = 0;
        while (__any(live)) {
            if (live) {
                int k;
                const int d = ccw_first(m, s, &k);
                // Clear neighbours examined in this visit: directions s+1 .. s+k, of which those after `from` are still to come; `from` is the t-th
                // examined direction (t >= 1), or 0 at a fresh visit. Round 4: the visit's cracks as bit sets instead of a loop over them (the loop
                // with the waypoint test inside ran up to seven times per step for the slowest lane of the wave). Relative bit r <-> direction
                // s + 1 + r, so that the lowest set bit is the first crack in examination order.
                const int t = (from - s) & 7;
                const uint32_t sh = (uint32_t)(s + 1) & 7u;
                const uint32_t rel = ((1u << k) - 1u) & ~((1u << t) - 1u);                       // examined and still to come
                const uint32_t ex = ((rel << sh) | (rel << sh >> 8)) & 0x55u;                    // absolute directions, cracks only (E, N, W, S)
                if (ex) {
                    const uint32_t x = pos & 0xFFFFu, y = pos >> 16;
                    uint32_t way = (((y & (uint32_t)a.grid_mask) == 0) ? 0x11u : 0u) | (((x & (uint32_t)a.grid_mask) == 0) ? 0x44u : 0u);   // W, E cracks on grid rows; N, S on grid columns
                    // start-candidate cracks off the grid (3x3 rule of kernel 1) are waypoints when they pass the run rule
                    if ((ex & ~way & 0x10u) && (m & 0x1Eu) == 0 && run_rule_any<256>(tiles, a.tnx, rows, tid, blk, pos, 4)) way |= 0x10u;
                    if ((ex & ~way & 0x01u) && ((m >> 1) & 1u) && (int)x + 1 <= a.width - 2 && run_rule_any<256>(tiles, a.tnx, rows, tid, blk, pos, 0)) way |= 0x01u;
                    const uint32_t hit = ex & way;
                    uint32_t seen = ex;                                                           // cracks examined up to and including the first waypoint
                    if (hit) {
                        const uint32_t hr = ((hit | (hit << 8)) >> sh) & 0xFFu;                   // back to examination order
                        const uint32_t r = (uint32_t)__builtin_ctz(hr);
                        const uint32_t dir = (sh + r) & 7u;
                        next_key = (pos << 2) | (dir >> 1);
                        const uint32_t upto = (2u << r) - 1u;
                        seen = ex & ((upto << sh) | (upto << sh >> 8));
                    }
                    if (seen & 0x10u) mn = min(mn, pos);
                    if (seen & 0x01u) mn = min(mn, pos + 1u);
                }
                if (next_key != NONE32 || ++step > max_steps) {
                    live = false;
                } else {
                    pos += tb_dpos(d);
                    s = (d + 4) & 7;
                    from = s;
                    len++;
                }
            }
            if (__any(live && !tb_inside(blk, pos))) tb_load<256>(tiles, a.tnx, a.tny, pos, rows, tid, blk);
            if (live) m = tb_mask<256>(rows, tid, blk, pos);
        }
        if (!is_node) continue;
        if (next_key == NONE32) {   // no waypoint within the bound: cannot happen for borders that cross the grid
            flag_overflow(a.counters, a.trig_cnt, plane, ST_SEGMENT_ERROR);
            flag = 2, next_key = key, len = 0;
        }
        a.node[nb + i] = make_uint4(next_key, mn, len | ((uint32_t)flag << 16), NONE32);
    }
}

// Kernel S3: next key -> node index
__global__ __launch_bounds__(256) void link_kernel(SegArgs a) {
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.seg_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const size_t nb = (size_t)plane * a.cap_raw;
    const uint32_t* hash = a.hash + (size_t)plane * (a.hash_mask + 1);
    for (uint32_t i = chunk * blockDim.x + threadIdx.x; i < n; i += a.seg_chunks * blockDim.x) {
        const uint4 nd = a.node[nb + i];
        const uint32_t key = nd.x;
        uint32_t found = i;   // a dead node points at itself
        if (!((nd.z >> 16) & 2u)) {
            found = NONE32;
            for (uint32_t slot = hash_key(key, a.hash_mask), probes = 0; probes <= a.hash_mask; slot = (slot + 1) & a.hash_mask, probes++) {
                const uint32_t j = hash[slot];
                if (j == NONE32) break;
                if (a.raw[nb + j].y == key) {
                    found = j;
                    break;
                }
            }
            if (found == NONE32) {
                flag_overflow(a.counters, a.trig_cnt, plane, ST_SEGMENT_ERROR);
                found = i;
            }
        }
        a.node[nb + i].x = found;
    }
}

// Kernel S4: one lane per start candidate: ONE lap around the cycle of nodes. Every node passed is stamped (atomicMin) with
// (start key, start node, visits from the start): the true start has the smallest key of its cycle and completes its
// lap, so afterwards every node of a kept border carries the true start's stamp; false starts drop out at the first
// node whose segment holds a smaller key and can only leave larger stamps behind.
// One frame per call (round 4): a lap is a chain of dependent node reads, 0.24 us each - 48 us for a 640x480 photograph, 95 us for the board still, a
// fifth of the call. skip_kernel gives every node its SKIP-th successor with the smallest key and the visits of the SKIP segments in between; a lap
// then takes SKIP segments per hop while the smallest key of the stretch stays above its own, stamps only the nodes it lands on, and walks single
// segments once the stretch holds its own key (the start's crack is examined in the segment before the start node: the lap is about to close).
// fill_kernel afterwards carries every stamp SKIP - 1 nodes forward (atomicMin like the laps: the true start's stamps - the smallest key, and per
// node the smallest offset - win), so every node of a kept border ends with the stamp the single-segment lap gives it.
constexpr int SKIP = 8;
__global__ __launch_bounds__(256) void skip_kernel(SegArgs a) {
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.seg_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const size_t nb = (size_t)plane * a.cap_raw;
    for (uint32_t i = chunk * blockDim.x + threadIdx.x; i < n; i += a.seg_chunks * blockDim.x) {
        uint32_t j = i, mn = NONE32, len = 0;
        for (int h = 0; h < SKIP; h++) {
            const uint4 nd = a.node[nb + j];
            mn = min(mn, nd.y), len += nd.z & 0xFFFFu, j = nd.x;
        }
        a.skipn[nb + i] = make_uint4(j, mn, len, 0u);
    }
}
__global__ __launch_bounds__(256) void fill_kernel(SegArgs a) {
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.seg_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const size_t nb = (size_t)plane * a.cap_raw;
    for (uint32_t i = chunk * blockDim.x + threadIdx.x; i < n; i += a.seg_chunks * blockDim.x) {
        const unsigned long long st = a.stamp[nb + i];
        if (st == NONE64) continue;
        const unsigned long long tag = st & ~0x3FFFull;
        uint32_t off = (uint32_t)(st & 0x3FFFull), j = i;
        for (int h = 1; h < SKIP; h++) {
            const uint4 nd = a.node[nb + j];
            off += nd.z & 0xFFFFu, j = nd.x;
            if (off > 0x3FFFu) break;
            atomicMin(&a.stamp[nb + j], tag | off);
        }
    }
}

__global__ __launch_bounds__(64) void cycle_kernel(SegArgs a) {
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.cyc_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const size_t nb = (size_t)plane * a.cap_raw;
    for (uint32_t i = chunk * blockDim.x + threadIdx.x; i < n; i += a.cyc_chunks * blockDim.x) {
        if ((a.node[nb + i].z >> 16) != 1u) continue;
        const uint32_t key = a.raw[nb + i].y, pos0 = key >> 2;
        const int hole = (key & 3u) == 0;
        const uint32_t k0 = pos0 + (hole ? 1u : 0u);     // the scan transition: outer -> the pixel, hole -> clear pixel right of it
        const unsigned long long tag = ((unsigned long long)k0 << 34) | ((unsigned long long)i << 14);
        uint32_t total = 0, j = i, hops = 0;
        bool ok = true;
        if (a.skipn) {   // SKIP segments per hop until the stretch ahead holds this start's own key
            for (;; hops++) {
                const uint4 sk = a.skipn[nb + j];
                if (sk.y == k0) break;
                if (sk.y < k0 || hops > 4u * (uint32_t)a.max_contour) {
                    ok = false;
                    break;
                }
                atomicMin(&a.stamp[nb + j], tag | total);
                total += sk.z;
                if (total >= (uint32_t)a.max_contour) {
                    ok = false;
                    break;
                }
                j = sk.x;
            }
        }
        for (; ok; hops++) {
            const uint4 nd = a.node[nb + j];
            if (nd.y < k0 || hops > 4u * (uint32_t)a.max_contour) {
                ok = false;
                break;
            }
            atomicMin(&a.stamp[nb + j], tag | total);
            total += nd.z & 0xFFFFu;
            if (total >= (uint32_t)a.max_contour) {
                ok = false;
                break;
            }
            j = nd.x;
            if (j == i) break;
        }
        if (!ok || (int)total <= a.min_contour) continue;
        const uint32_t slot = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_CDESC], 1u);
        const uint32_t off = atomicAdd(&a.trig_cnt[plane * TRIG_CNT_STRIDE + TC_POOL], total);
        if (slot >= a.cap_cdesc) {
            flag_overflow(a.counters, a.trig_cnt, plane, ST_CDESC_OVERFLOW);
            continue;
        }
        ContourDesc cd;
        cd.plane = plane, cd.x0 = (int16_t)(pos0 & 0xFFFFu), cd.y0 = (int16_t)(pos0 >> 16), cd.hole = hole, cd.n = (int)total;
        cd.key = (k0 >> 16) * (uint32_t)a.width + (k0 & 0xFFFFu);
        cd.pool_off = (uint32_t)plane * a.cap_pool + off;
        cd.ck_off = 0xFFFFFFFFu, cd.pad_ = 0;
        const uint32_t gslot = (uint32_t)plane * a.cap_cdesc + slot;
        if (off + total > a.cap_pool) {
            flag_overflow(a.counters, a.trig_cnt, plane, ST_POOL_OVERFLOW);
            cd.n = 0;
            a.cdesc[gslot] = cd;
            continue;
        }
        a.cdesc[gslot] = cd;
        a.node[nb + i].w = gslot;   // emit_kernel finds the descriptor through the start node
    }
}
// Round 4 measured the laps hopping through a copy of the plane's node table in LDS (one workgroup per frame): a hop costs what it costs through the
// L2 (0.24 us: the chain is the atomic's address, the compare and the loop, not the read), and one CU then takes every lap of the frame - 48.8 against 48.6 us for a
// 640x480 still, 219 against 60 us for the board still. Not kept.

// Kernel S5: one lane per node of a kept border: write the points of its visits
__global__ __launch_bounds__(256) void emit_kernel(SegArgs a) {
    __shared__ uint32_t rows[TB_ROWS * 256];
    int plane, chunk;
    if (!plane_of_block(a.nplanes, a.seg_chunks, &plane, &chunk)) return;
    const uint32_t n = min(a.raw_cnt[plane * TRIG_CNT_STRIDE], a.cap_raw);
    const uint64_t* __restrict__ tiles = a.tiles + (size_t)plane * a.tnx * a.tny;
    const size_t nb = (size_t)plane * a.cap_raw;
    const int tid = threadIdx.x;
    for (uint32_t i0 = chunk * blockDim.x; i0 < n; i0 += a.seg_chunks * blockDim.x) {
        const uint32_t i = i0 + tid;
        bool live = i < n;
        unsigned long long st = NONE64;
        uint32_t len = 0, ci = NONE32;
        if (live) {
            st = a.stamp[nb + i];
            len = a.node[nb + i].z & 0xFFFFu;
            if (st != NONE64 && len) ci = a.node[nb + (uint32_t)((st >> 14) & 0xFFFFFu)].w;   // descriptor of the stamping start, if kept
        }
        live = live && ci != NONE32;
        if (!__any(live)) continue;
        const uint32_t key = live ? a.raw[nb + i].y : ((0x00010001u << 2) | 2u);
        uint32_t pos = key >> 2;
        TileBlock blk;
        tb_load<256>(tiles, a.tnx, a.tny, pos, rows, tid, blk);
        uint32_t m = tb_mask<256>(rows, tid, blk, pos);
        int s = live ? cw_first(m, (int)(key & 3u) * 2) : 0;
        short2* out = a.pool;
        uint32_t idx = (uint32_t)(st & 0x3FFFu), total = 1;
        if (live) {
            const ContourDesc cd = a.cdesc[ci];
            out = a.pool + cd.pool_off;
            total = (uint32_t)cd.n;
        }
        uint32_t t = 0;
        while (__any(live)) {
            if (live) {
                int k;
                const int d = ccw_first(m, s, &k);
                pos += tb_dpos(d);
                s = (d + 4) & 7;
                if (++idx >= total) idx -= total;
                out[idx] = make_short2((short)(pos & 0xFFFFu), (short)(pos >> 16));
                if (++t >= len) live = false;
            }
            if (__any(live && !tb_inside(blk, pos))) tb_load<256>(tiles, a.tnx, a.tny, pos, rows, tid, blk);
            if (live) m = tb_mask<256>(rows, tid, blk, pos);
        }
    }
}

static void fill_seg_args(SegArgs& a, const FrameGeom& g, int nplanes, const DetectParams& p, const Buffers& b) {
    a.tiles = b.tiles, a.tnx = tiles_x(g.width), a.tny = tiles_y(g.height), a.width = g.width, a.height = g.height, a.nplanes = nplanes;
    a.grid_mask = b.grid_mask;
    a.raw = b.raw, a.raw_cnt = b.raw_cnt, a.cap_raw = b.cap_raw;
    a.node = b.node, a.stamp = b.stamp;
    a.hash = b.hash, a.hash_mask = b.hash_mask;
    a.cdesc = b.cdesc, a.pool = b.pool, a.counters = b.counters, a.cap_cdesc = b.cap_cdesc, a.cap_pool = b.cap_pool, a.trig_cnt = b.trig_cnt;
    a.min_contour = p.min_contour, a.max_contour = p.max_contour;
    seg_chunks_for(nplanes, &a.seg_chunks, &a.cyc_chunks);
    a.skipn = (nplanes <= 2 && b.tune.seg_skip) ? b.skipn : nullptr;
}

void launch_segments(hipStream_t s, const FrameGeom& g, int nplanes, const DetectParams& p, const Buffers& b) {
    SegArgs a;
    fill_seg_args(a, g, nplanes, p, b);
    const int groups = (nplanes + 7) / 8 * 8;
    hipLaunchKernelGGL(segment_kernel, dim3(groups * a.seg_chunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(link_kernel, dim3(groups * a.seg_chunks), dim3(256), 0, s, a);
    if (a.skipn) hipLaunchKernelGGL(skip_kernel, dim3(groups * a.seg_chunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(cycle_kernel, dim3(groups * a.cyc_chunks), dim3(64), 0, s, a);
    if (a.skipn) hipLaunchKernelGGL(fill_kernel, dim3(groups * a.seg_chunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(emit_kernel, dim3(groups * a.seg_chunks), dim3(256), 0, s, a);
}

}  // namespace ah
