// Internal device-side data layout of libarucohip (MI355X / gfx950). Not part of the C ABI.
//
// HBM layout per handle (F = frames in the batch, T = threshold planes per frame, P = F*T planes):
//   thres [P][H][W] u8   thresholded image (API-visible product, MarkerDetector::getThresholdedImage)
//   tiles [P][H/8+1][W/8+1] u64 binary image cv::findContours works on, 8x8-pixel tiles (1-px frame cleared, zero pad row/col)
//   raw   [P][capR] u32x2 segment mode only: waypoint cracks {start candidate, pos << 2 | code}
//   trig  [P][capT] u32x2 candidates that pass the run rule
//   cdesc [P][capC]      borders that passed the size filter {plane, start, hole, n, key, pool offset}
//   pool  [P][capPts] short2 contour points (checkpoints in front of each border's points)
//   quads [F][capQ]      4-vertex convex polygons
//   cands [F][capC]      ordered candidates with decode result and refined corners
//   markers [F][capM]    arucohip_marker_t
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/arucohip.h"

namespace ah {

constexpr int WAVE = 64;
constexpr int TRIG_CNT_STRIDE = 32;   // uint32 words between per-plane counters (one 128-byte line per plane)
constexpr int TC_CDESC = 2;           // words of a plane's trig_cnt line: 0 / 1 = outer / hole start candidates,
constexpr int TC_POOL = 3;
constexpr int TC_STATUS = 5;          // 5 = overflow bits (StatusBits) of THIS plane: a frame whose planes carry any is reported with n = -1
constexpr int TC_SNAP = 4;            // 4 = descriptors that existed when the late walker generations were forked            // 2 = contour descriptors, 3 = contour points allocated (per plane: no global hot counter)
constexpr int WALK_BLOCKS = 16;       // 64-lane walker workgroups per plane


enum Counter {
    CNT_NMARK = 0,     // entries of the flat marker list (all frames)
    CNT_UNUSED1 = 1,
    CNT_UNUSED2 = 2,
    CNT_STATUS = 3,    // overflow bit flags
    CNT_NCAND = 5,     // entries of the flat candidate list (all frames)
    CNT_FIXED = 8      // per-frame counters follow: [CNT_FIXED + f] = quads of frame f
};

enum StatusBits {
    ST_TRIG_OVERFLOW = 1,
    ST_CDESC_OVERFLOW = 2,
    ST_POOL_OVERFLOW = 4,
    ST_QUAD_OVERFLOW = 8,
    ST_CAND_OVERFLOW = 16,
    ST_MARKER_OVERFLOW = 32,
    ST_SEGMENT_ERROR = 64,
};

// a device list overflowed while working on `plane`: the batch-wide status word (the call's return code) and the plane's own word (which
// frame to give up on: the other frames' results stay valid; the reference has no limits, src/markerdetector.cpp:496-635)
__device__ __forceinline__ void flag_overflow(uint32_t* counters, uint32_t* trig_cnt, int plane, uint32_t bit) {
    atomicOr(&counters[CNT_STATUS], bit);
    atomicOr(&trig_cnt[(size_t)plane * TRIG_CNT_STRIDE + TC_STATUS], bit);
}

struct ContourDesc {
    int32_t plane;     // frame*T + t
    int16_t x0, y0;    // start pixel
    int32_t hole;
    int32_t n;         // number of points
    uint32_t key;      // raster index of the scan transition (y*W + x_trigger)
    uint32_t pool_off; // first point in pool
    uint32_t ck_off;   // checkpoints of the border: word offset into walk_scratch, or 0xFFFFFFFF = in front of the points
    uint32_t pad_;
};

struct Quad {
    int16_t x[4], y[4];
    int32_t cdesc;     // index into cdesc list
    uint32_t key;      // ordering key inside the frame: t << 26 | (0x3FFFFFF - raster)  (ascending = reference order)
    int32_t pad_;
};

struct Cand {
    float c[8];        // current corners (x0,y0,...)
    int16_t qx[4], qy[4];  // integer corners after orientation normalisation (detectRectangles output)
    int32_t cdesc;
    int32_t swapped;   // contour must be read reversed
    int32_t id;        // -1 = not a marker
    int32_t nrot;
};

struct FrameGeom {
    int width, height;
    size_t row_stride, frame_stride;   // of the input gray frames
};

struct CamModel {
    int has_K, has_dist;
    float K[9];
    double k[8];       // k1,k2,p1,p2,k3,k4,k5,k6 (zero padded)
    float marker_size;
    int y_perp;
};

struct DetectParams {
    int thres_method;
    int block[16];      // per threshold plane: block size (ADPT) (already fixed up to odd >= 3)
    double p1[16];      // per plane param1 (FIXED uses it as the threshold)
    int idelta;         // floor(param2)
    int nthr;           // planes per frame
    int corner_method;
    int warp_size;
    int min_contour, max_contour;   // contour length bounds (exclusive)
    int bx0, by0, bx1, by1;         // valid region of the border filter [bx0,bx1) x [by0,by1)
    int subpix_win;
    int locked, locked_wsize;       // _useLockedCorners and the window of findCornerMaxima (int(_thresParam1))
    // decoder: 0 = 5x5 fiducial, 1 = highly reliable markers with the handle's dictionary, 2 = host callback (the device
    // only warps; ids come back through launch_set_decoded)
    int decoder;
    int hrm_n, hrm_count;
    uint32_t hrm_correction;        // largest Hamming distance that is still corrected
    const uint64_t* hrm_codes;      // device
};

// Tuning / A-B knobs. They are environment variables (INTEGRATION.md lists them), read ONCE when a handle is created and
// kept with it: a handle behaves the same for its whole life whatever the environment does later, and no kernel launcher
// touches the environment.
struct Tuning {
    int walk_fork = 1;         // ARUCOHIP_WALK_FORK: late walker generations on a side stream
    int chain = 0;             // ARUCOHIP_CHAIN: threshold kernels of chunk streams one after the other
    int cand_sparse = 1;       // ARUCOHIP_CAND_SPARSE: bitmap-driven start candidates
    int cand_waves = 32;       // ARUCOHIP_CAND_WAVES: waves per plane of that kernel (4 0.25, 8 0.18, 16 0.13, 32 0.13 ms per 512 frames)
    int cand_chunks = 16;      // ARUCOHIP_CAND_CHUNKS: workgroups per plane of the lane-per-tile kernel
    int leash = 0;             // ARUCOHIP_LEASH: steps of the first walker pass (0 = default)
    int gens[32] = {};         // ARUCOHIP_GENS: steps per generation of long walks
    int ngens = 0;
    int fork_after = 3;        // ARUCOHIP_FORK_AFTER: generations on the main stream
    int gen_xcd = 1;           // ARUCOHIP_GEN_XCD=0: one generation list for the whole chip (rounds 1-3) instead of one per XCD
    int seg_skip = 1;          // ARUCOHIP_SEG_SKIP=0: the laps of the segment pipeline take one segment per hop also for one frame per call
    int quad_dual = 1;         // ARUCOHIP_QUAD_DUAL=0: one border per wave in contour_quad (round 3), 1: two borders of <= 512 points per wave
    int quad_blocks = 12;      // ARUCOHIP_QUAD_BLOCKS: workgroups per plane of contour_quad. One border per wave (rounds 1-3): 8: 0.93 ms, 16: 0.68, 24: 0.60, 32: 0.72 -> 24.
                               // Two borders per wave (round 4): a workgroup needs an even number of short borders to pair them all, so fewer, longer lists:
                               // 24 / 16 / 12 = 467.7k / 471.1k / 480.0k frames/s (flat stream), 273.1k / 276.2k / 276.2k (cluttered), same box
    int thres_lazy = 1;        // ARUCOHIP_THRES_BYTES=1 clears it: the threshold kernel always writes the byte image
    int threshold_wide = 1;    // ARUCOHIP_THRESHOLD_WIDE: 16-pixel-per-lane threshold kernel where it applies
#ifdef ARUCOHIP_STAGE_EXPERIMENT
    int stop_after = 99;       // stage-cost experiment (tools/stage_cost.sh builds a variant library with this flag and reads ARUCOHIP_STOP_AFTER):
                               // 1 threshold, 2 start candidates, 3 first walker pass, 4 generations, 5 contour_quad, 6 frame_candidates, 7 warp + Otsu, 8 LINES
#endif
    int threshold_eo = 1;      // ARUCOHIP_THRESHOLD_EO: its round-3 form for 7x7 blocks (unpacked row ring, folded constants); 0 = the round-2 kernel
};
Tuning read_tuning();          // capi.hip
// RUN_STAGE(tune, n): does the pipeline run past stage n? Always, except in the stage-cost experiment's variant build, where the pipeline is cut
// behind the stage ARUCOHIP_STOP_AFTER names (results are then meaningless; arucohip_build_info() names the flag and bench.py prints no headline).
#ifdef ARUCOHIP_STAGE_EXPERIMENT
#define RUN_STAGE(tune, n) ((tune).stop_after > (n))
#else
#define RUN_STAGE(tune, n) true
#endif

// Border lines of a thresholded plane kept beside the bit tiles (lazy byte image): row 0 at 0, row H-1 at Wp, column 0 at 2 Wp, column W-1
// at 2 Wp + Hp, with Wp / Hp = width / height rounded up to 16 so that every line starts 16-byte aligned (the kernels store 16 and 4 bytes
// at a time) and the virtual rows that complete the last tile row have somewhere to go.
__host__ __device__ inline size_t thres_edge_wp(int W) { return (size_t)((W + 15) & ~15); }
__host__ __device__ inline size_t thres_edge_hp(int H) { return (size_t)((H + 15) & ~15); }
__host__ __device__ inline size_t thres_edge_stride(int W, int H) { return 2 * thres_edge_wp(W) + 2 * thres_edge_hp(H); }

// device pointers + capacities handed to kernels
struct Buffers {
    Tuning tune;
    uint8_t* thres;
    uint64_t* tiles;       // [P][tiles_y(H)][tiles_x(W)] binary image in 8x8-pixel tiles (bits_tiles.h)
    uint64_t* tile_bits;   // [P][tiles_y(H)][2 * tile_strips(W)] non-empty-tile bitmap: per 128-tile strip one word for the even
                           // tiles (bit i = tile 128 s + 2 i) and one for the odd ones (tile 128 s + 2 i + 1)
    uint2* raw;            // [P][cap_raw] waypoint cracks per plane (segment mode)
    uint32_t* raw_cnt;     // [P * TRIG_CNT_STRIDE] fill level of each plane's raw list (one counter per 128-byte line)
    uint2* trig;           // [P][cap_trig] candidates that pass the run rule
    uint32_t* trig_cnt;    // [P * TRIG_CNT_STRIDE]
    uint2* gen_buf;        // [P][cap_trig] storage of the long-walk generation lists (states + ring ids)
    uint32_t* ring_cnt;    // [P * TRIG_CNT_STRIDE] checkpoint rings handed out per plane (outer, hole)
    uint32_t* gen_cnt;     // counters of the long-walk generation lists (k_contours.hip), zeroed per batch
    ContourDesc* cdesc;
    short2* pool;
    uint8_t* thres_edge;    // [P][thres_edge_stride(W, H)] border lines of the thresholded planes when the byte image is left out (k_threshold.hip)
    uint64_t* thr_stamps;   // timing: per-wave device-clock stamps of the wide threshold kernel [2 * waves]
    uint64_t* thr_acc;      // timing: {clock ticks, launches} accumulated by stamp_reduce_kernel
    int thr_stamp_on;       // stamps are taken (arucohip_enable_timing)
    uint32_t* walk_scratch; // checkpoint rings of the long walks [P][2][LONG_CAP][max_contour/16]
    uint4* node;            // [P][cap_raw] waypoint records of the segment pipeline
    uint4* skipn;           // [P][cap_raw] a node's 8th successor, smallest key and visits of the 8 segments up to it (one frame per call: k_segments.hip)
    unsigned long long* stamp; // [P][cap_raw] (start key, start node, offset) of the start that owns the node
    uint32_t* hash;         // [P][hash_mask+1] node index by waypoint key
    uint32_t hash_mask;
    Quad* quads;
    Cand* cands;
    int32_t* ncands;       // [F]
    uint32_t* cand_list;   // flat list over all frames: frame << 16 | index, counters[CNT_NCAND] entries
    double* iM;            // [cap_flat][9] inverse homographies
    uint16_t* hist;        // [cap_flat][256] patch histograms
    int32_t* othr;         // [cap_flat] Otsu thresholds
    uint8_t* patches;      // [cap_flat][warp_size^2] canonical patches
    uint32_t cap_flat;
    arucohip_marker_t* markers;
    int32_t* nmarkers;     // [F]
    uint32_t* marker_list; // flat list over all frames: frame << 16 | index, counters[CNT_NMARK] entries (the pose kernel's work list)
    uint32_t* counters;
    uint32_t cap_raw, cap_trig;   // per plane
    uint32_t long_cap;            // checkpoint rings (long walks) per plane and kind
    int seg_mode, grid_mask;      // contour pipeline: 0 = walkers, 1 = waypoint segments (grid spacing = grid_mask + 1)
    uint32_t cap_cdesc, cap_pool;   // per plane: cdesc [P][cap_cdesc], pool [P][cap_pool]
    int cap_quads, cap_cands, cap_markers;   // per frame
};

// Every kernel after the threshold pass is a chain of dependent steps on few wavefronts. With batches in flight its waves
// share SIMDs with the next batch's streaming threshold waves, which are always ready to issue; at equal priority the
// dependent chain only gets every n-th issue slot. Raised priority lets the sparse chains issue whenever they can — they leave
// most slots to the streaming waves anyway.
#ifndef LBP_PRIO
#define LBP_PRIO 3
#endif
__device__ __forceinline__ void latency_bound_priority() { if (LBP_PRIO > 0) __builtin_amdgcn_s_setprio(LBP_PRIO); }
// The kernels with many busy waves (contour_quad, warp_hist, start candidates): between the streaming pass and the sparse chains
#ifndef LBP_PRIO_HEAVY
#define LBP_PRIO_HEAVY LBP_PRIO
#endif
__device__ __forceinline__ void throughput_bound_priority() { if (LBP_PRIO_HEAVY > 0) __builtin_amdgcn_s_setprio(LBP_PRIO_HEAVY); }

// ---- kernel launchers (host side, defined in the .hip files)
void launch_bgr2gray(hipStream_t s, const uint8_t* bgr, size_t row_stride, size_t frame_stride, int width, int height, int nframes, uint8_t* gray);
bool launch_threshold(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, bool lazy);
void launch_expand_thres(hipStream_t s, const FrameGeom& g, int plane, const Buffers& b);
void launch_undist_map(hipStream_t s, int W, int H, const float* K, const float* dist, int ndist, short2* xy, uint16_t* fxy);
void launch_remap(hipStream_t s, const uint8_t* src, size_t row_stride, size_t frame_stride, int W, int H, int cn, int nframes, const short2* xy,
                  const uint16_t* fxy, uint8_t* dst);
int launch_canny(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, int nthr, const Buffers& b, uint64_t* surv, uint64_t* edge, uint32_t* changed);
void launch_erode(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, uint8_t* tmp);
size_t erode_tiles_tmp_bytes(const FrameGeom& g, int nplanes);
void launch_erode_tiles(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, uint8_t* tmp);   // on the lazy byte image (tiles + border lines)
void launch_tile_bitmap(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b);
void launch_binary_planes(hipStream_t s, const uint8_t* thres_in, const FrameGeom& g, int nframes, const Buffers& b);
void launch_start_candidates(hipStream_t s, const FrameGeom& g, int nplanes, const Buffers& b, int min_contour = 0);
struct WalkFork {
    hipStream_t side;            // stream of the late walker generations (nullptr: everything on the main stream)
    hipEvent_t forked, joined;
    hipEvent_t after_first;      // recorded behind the first pass (per-kernel timing), may be null
};
bool launch_walkers(hipStream_t s, const WalkFork& fk, const FrameGeom& g, int nplanes, const DetectParams& p, const Buffers& b);
size_t walk_scratch_words(int nplanes, const DetectParams& p, uint32_t long_cap);   // capacity launch_walkers needs in Buffers::walk_scratch
constexpr size_t GEN_CNT_WORDS = 2 * 32 * 32 * 8;                 // words of Buffers::gen_cnt: [2 kinds][GEN_MAX + 2 generations][8 sublists] lines of 32 words
void launch_segments(hipStream_t s, const FrameGeom& g, int nplanes, const DetectParams& p, const Buffers& b);
void launch_contour_quads(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, int pass = 0);
void launch_frame_candidates(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b);
// fused_cells: the built-in 5x5 decoder runs as the head of refine_lines_kernel (launch_refine_lines with the same flag) instead of a kernel of its own
void launch_decode(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b, bool fused_cells = false);
void launch_set_decoded(hipStream_t s, const Buffers& b, uint32_t n, const int2* id_nrot_dev);
void launch_refine_lines(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const CamModel& cam, const Buffers& b, bool fused_cells = false);
void launch_locked_corners(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b);
void launch_refine_pixels(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, const DetectParams& p, const Buffers& b);
void launch_finalize(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const CamModel& cam, const Buffers& b, arucohip_marker_t* out = nullptr,
                     int out_cap = 0, int32_t* n_out = nullptr);   // out / n_out: the caller's device arrays, written as well (no pose to add later)
void launch_pose(hipStream_t s, int nframes, const CamModel& cam, const Buffers& b);
void launch_warp_only(hipStream_t s, const uint8_t* gray, const FrameGeom& g, const float* quad_dev, int size, uint8_t* dst_dev);
void launch_pnp_points(hipStream_t s, const float* obj, const float* img, int npts, const CamModel& cam, double* rt_out, int* ok_out);
void launch_project_points(hipStream_t s, const float* obj, int npts, const double* rt, const CamModel& cam, float* img_out);
void launch_gl_modelview(hipStream_t s, int nframes, int cap, const Buffers& b, double* out_dev);
void launch_marker_pose(hipStream_t s, arucohip_marker_t* markers, int n, const CamModel& cam);
void launch_board_pose(hipStream_t s, int nframes, const Buffers& b, const int32_t* ids, const float* obj, int nboard, int info_type,
                       float marker_size, float repj_thres, const CamModel& cam, arucohip_board_t* out, float* prob);

}  // namespace ah
