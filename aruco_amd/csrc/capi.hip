// Host side of libarucohip: handle, device buffers, launch order. Implements include/arucohip.h.
//
// Launch order of one batch (all on the handle's stream, no host round trip until the final D2H of the markers):
//   memset counters -> threshold(+masks+start candidates) -> walkers -> contour/quad -> frame candidates ->
//   warp+decode -> corner refinement (+rotation) -> finalize -> pose
// which is the stage order of MarkerDetector::detect (/root/reference/src/markerdetector.cpp:302-478).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bits_tiles.h"
#include "internal.h"

namespace ah {
void launch_rotate_x(hipStream_t s, double* rt);
}

namespace ah {
Tuning read_tuning() {
    Tuning t;
    auto geti = [](const char* name, int dflt) {
        const char* e = getenv(name);
        return (e && *e) ? atoi(e) : dflt;
    };
    t.walk_fork = geti("ARUCOHIP_WALK_FORK", 1) != 0;
    t.chain = geti("ARUCOHIP_CHAIN", 0) != 0;
    t.cand_sparse = geti("ARUCOHIP_CAND_SPARSE", 1) != 0;
    t.cand_waves = std::max(1, geti("ARUCOHIP_CAND_WAVES", 32));
    t.cand_chunks = std::max(1, geti("ARUCOHIP_CAND_CHUNKS", 16));
    t.leash = geti("ARUCOHIP_LEASH", 0);
    t.fork_after = geti("ARUCOHIP_FORK_AFTER", 3);
    t.quad_blocks = std::max(1, geti("ARUCOHIP_QUAD_BLOCKS", 12));
    t.quad_dual = geti("ARUCOHIP_QUAD_DUAL", 1) != 0;
    t.seg_skip = geti("ARUCOHIP_SEG_SKIP", 1) != 0;
    t.gen_xcd = geti("ARUCOHIP_GEN_XCD", 1) != 0;
    t.threshold_wide = geti("ARUCOHIP_THRESHOLD_WIDE", 1) != 0;
    t.threshold_eo = geti("ARUCOHIP_THRESHOLD_EO", 1) != 0;
#ifdef ARUCOHIP_STAGE_EXPERIMENT
    t.stop_after = geti("ARUCOHIP_STOP_AFTER", 99);   // truncates the pipeline: results are meaningless, only the step time is
#endif
    t.thres_lazy = geti("ARUCOHIP_THRES_BYTES", 0) == 0;
    if (const char* e = getenv("ARUCOHIP_GENS")) {
        for (const char* q = e; *q && t.ngens < 32;) {
            const int v = atoi(q);
            if (v > 0) t.gens[t.ngens++] = v;
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    }
    return t;
}
}  // namespace ah

using namespace ah;

enum { STAGE_THRESHOLD = 0, STAGE_RECTANGLES, STAGE_IDENTIFY, STAGE_SUBPIXEL, STAGE_FILTERING, STAGE_COUNT };
static const char* kStageNames[STAGE_COUNT] = {"Threshold", "Rectangles", "Identify", "Subpixel", "Filtering"};
// one event after every kernel of a batch; a ring of TSETS batches so that asynchronous steps can be averaged
// slot k = the interval between mark k and mark k + 1. walker_long = the generations of long walks up to the fork of the side
// stream; contour_quad = both passes including the wait for the side stream's late generations.
enum { K_THRESHOLD = 0, K_FILTER, K_WALKERS, K_WALKERS_LONG, K_CONTOUR_QUADS, K_FRAME_CANDS, K_DECODE, K_REFINE_LINES, K_REFINE_PIXELS, K_FINALIZE, K_POSE, K_COUNT };
static const char* kKernelNames[K_COUNT] = {"threshold_kernel", "candidates_kernel", "walker_kernel", "walker_long_kernel", "contour_quad_kernel",
                                            "frame_candidates_kernel", "decode_kernel", "refine_lines_kernel", "refine_pixels_kernel", "finalize_kernel",
                                            "pose_kernel"};
static const int kKernelStage[K_COUNT] = {STAGE_THRESHOLD, STAGE_RECTANGLES, STAGE_RECTANGLES, STAGE_RECTANGLES, STAGE_RECTANGLES, STAGE_RECTANGLES,
                                          STAGE_IDENTIFY, STAGE_IDENTIFY, STAGE_SUBPIXEL, STAGE_FILTERING, STAGE_FILTERING};
constexpr int TSETS = 32;

struct arucohip_handle {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    arucohip_params_t params;
    arucohip_limits_t lim;
    Buffers buf{};
    uint8_t* d_gray = nullptr;        // staging for host frames (gray) and the converted BGR frames
    size_t gray_bytes = 0;
    uint8_t* d_bgr = nullptr;         // staging for host BGR frames
    size_t bgr_bytes = 0;
    arucohip_marker_t* wt_out = nullptr;   // set around detect_core by chunk_enqueue: finalize_kernel writes the results there too
    int32_t* wt_n = nullptr;
    int wt_cap = 0;
    uint8_t* d_erode = nullptr;       // eroded planes (params.erode)
    size_t erode_bytes = 0;
    uint8_t* d_canny = nullptr;       // CANNY: survivor tiles, edge tiles, changed flag
    size_t canny_bytes = 0;
    // frame undistortion (arucohip_undistort): the map of the last camera is kept
    short2* d_umap_xy = nullptr;
    uint16_t* d_umap_f = nullptr;
    size_t umap_px = 0;
    int umap_w = 0, umap_h = 0, umap_nd = -1;
    float umap_K[9] = {}, umap_d[8] = {};
    uint8_t* d_undist = nullptr;      // undistorted frames when the caller wants them on the host
    size_t undist_bytes = 0;
    // highly reliable markers (arucohip_set_dictionary)
    uint64_t* d_hrm = nullptr;
    int hrm_n = 0, hrm_count = 0, hrm_tau0 = 0;
    float hrm_rate = 1.f;
    // caller's own decoder (arucohip_set_decoder_callback)
    arucohip_decoder_fn decoder_fn = nullptr;
    void* decoder_user = nullptr;
    int2* d_user_dec = nullptr;       // [cap_flat] {id, nRotations} returned by the callback
    uint32_t* hu_list = nullptr;      // pinned staging of the callback path: candidate list, decoder results, call order
    size_t hu_list_bytes = 0;
    uint8_t* hu_patches = nullptr;    // pinned staging: the canonical patches handed to the callback (+ one scratch patch)
    size_t hu_patch_bytes = 0;
    size_t scratch_words = 0;         // capacity of buf.walk_scratch
    size_t bits_bytes = 0;
    size_t patch_bytes = 0;           // capacity of buf.patches
    int bits_w = 0, bits_h = 0;       // geometry the bit image was last written with (pad words depend on it)
    // pinned host staging
    arucohip_marker_t* h_markers = nullptr;
    int32_t* h_n = nullptr;
    uint32_t* h_counters = nullptr;
    // small device scratch for the stage-level calls
    float* d_small_f = nullptr;       // 4096 floats
    double* d_small_d = nullptr;      // 64 doubles
    int* d_small_i = nullptr;
    uint8_t* d_patch = nullptr;       // MAX_WARP^2
    void* d_board = nullptr;          // batched board results + ids
    uint32_t* zero_block = nullptr;   // counters, gen_cnt, trig_cnt, raw_cnt, ring_cnt: zeroed together at the start of a batch
    size_t zero_words = 0;
    double* d_gl = nullptr;           // batched GL modelview matrices
    size_t gl_bytes = 0;
    // last call
    int last_w = 0, last_h = 0, last_frames = 0, last_nthr = 1;
    const uint8_t* last_gray = nullptr;
    FrameGeom last_geom{};
    bool timing = false;
    hipEvent_t ev[TSETS][K_COUNT + 1] = {};
    int tsets = 0;                       // batches recorded since the last reset
    float kernel_ms[K_COUNT] = {};       // averages over the recorded batches
    // Sub-batch pipelining: a batch larger than cap_frames is cut into up to nsub chunks; chunk 0 runs on this handle and
    // the caller's stream, chunk i on kids[i-1] and its own stream, so the latency-bound kernels of one chunk (border
    // following, Otsu) overlap the bandwidth-bound ones of another and host frames are copied while earlier chunks compute.
    int nsub = 1, cap_frames = 1;        // workers, frames each worker's buffers hold
    bool is_child = false;
    std::vector<arucohip_handle*> kids;
    hipEvent_t ev_fork = nullptr, ev_join[8] = {};
    hipStream_t side_stream = nullptr;   // late walker generations (k_contours.hip)
    hipEvent_t ev_wfork = nullptr, ev_wjoin = nullptr;
    hipEvent_t ev_thr = nullptr;         // this worker's threshold kernel has finished (staggers the chunks, see detect_batch)
    bool thres_bytes = true;             // buf.thres holds the last batch's byte image (else: tiles + buf.thres_edge, expanded on demand)
    hipEvent_t wait_thr = nullptr;       // set by detect_batch: event the next threshold kernel waits for
    int last_chunks = 1, last_per = 0;   // chunks and frames per chunk of the last batch
    // One frame per call (the reference's call shape, arucohip_detect): the chain of ~20 dependent dispatches of a frame is captured once per
    // (geometry, parameters, camera) into a hipGraph and replayed with ONE launch per call; the frame's H2D copy stays outside (its source
    // pointer changes with every call), the results land in the handle's pinned staging inside the graph.
    struct FrameGraph {
        hipGraphExec_t exec = nullptr;
        uint64_t key = 0;          // digest of everything the captured launches carry by value
        uint64_t seen = 0;         // key of the previous eager call: the second call with the same key captures (buffers are sized by then)
        int disabled = 0;          // ARUCOHIP_GRAPH=0, or a capture failed once
    } fgraph;
    // Batches in flight (arucohip_set_pipeline_depth / _submit / _wait): every pipeline lane is a complete worker (own
    // buffers, own stream); ticket t runs on lane t mod depth, so the latency-bound tail of batch t (border following,
    // decoding) overlaps the bandwidth-bound head of batch t+1.
    std::vector<arucohip_handle*> lanes;
    int next_ticket = 0;
    arucohip_handle* cur = nullptr;      // lane whose results the getters / board pose address (last waited ticket)
    arucohip_handle* retry = nullptr;    // one-frame handle with larger lists for frames that overflowed (arucohip_detect_batch_retry_overflowed)
    int retry_mult = 0;
    hipEvent_t ev_submit = nullptr;
    struct Pending {
        bool active = false;
        int ticket = -1, nframes = 0, cap = 0, out_on_device = 0;
        arucohip_marker_t* out = nullptr;
        int32_t* n_out = nullptr;
    } pend;
    std::string err;
};

// the worker that holds the results of the last completed batch
static arucohip_handle* active(arucohip_handle* h) { return (h && h->cur) ? h->cur : h; }

// worker that holds frame `frame` of the last batch (and the frame's index inside that worker)
static arucohip_handle* route(arucohip_handle* h, int frame, int* local) {
    if (h->last_chunks <= 1 || h->last_per <= 0) {
        *local = frame;
        return h;
    }
    const int c = std::min(frame / h->last_per, h->last_chunks - 1);
    *local = frame - c * h->last_per;
    return c == 0 ? h : h->kids[c - 1];
}

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char buf_[256];                                                                     \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (h)->err = buf_;                                                                    \
            return ARUCOHIP_E_HIP;                                                              \
        }                                                                                       \
    } while (0)

static int fail(arucohip_handle* h, int code, const char* msg) {
    if (h) h->err = msg;
    return code;
}

extern "C" {

int arucohip_version(void) { return ARUCOHIP_VERSION; }

#ifndef ARUCOHIP_SRC_HASH
#define ARUCOHIP_SRC_HASH "unknown"
#endif
#ifndef ARUCOHIP_EXTRA_FLAGS
#define ARUCOHIP_EXTRA_FLAGS ""
#endif
const char* arucohip_build_info(void) { return "src=" ARUCOHIP_SRC_HASH " flags=[" ARUCOHIP_EXTRA_FLAGS "]"; }

void arucohip_default_params(arucohip_params_t* p) {
    std::memset(p, 0, sizeof(*p));
    p->thres_method = ARUCOHIP_THRES_ADPT;
    p->thres_param1 = 7, p->thres_param2 = 7, p->thres_param1_range = 0;
    p->corner_method = ARUCOHIP_CORNER_LINES;
    p->warp_size = 56;
    p->min_size = 0.04f, p->max_size = 0.5f;
    p->border_dist = 0.025f;
    p->use_locked_corners = 0;
    p->decoder_kind = ARUCOHIP_DECODER_FIDUCIAL_5X5;
}

void arucohip_default_limits(arucohip_limits_t* l, int max_width, int max_height, int max_batch) {
    l->max_width = max_width, l->max_height = max_height, l->max_batch = std::max(max_batch, 1);
    l->max_thres_planes = 1;
    long px = (long)max_width * max_height;
    l->triggers_per_frame = (int)std::min<long>(std::max<long>(px / 16, 16384), 1 << 20);
    l->contours_per_frame = (int)std::min<long>(std::max<long>(px / 256, 1024), 16384);   // per threshold plane
    l->points_per_frame = (int)std::min<long>(std::max<long>(px / 8, 65536), 1 << 21);
    l->candidates_per_frame = 256;
    l->markers_per_frame = 128;
    // a synthetic 1080p frame has ~200 long walks per plane, a cluttered one several times that; small batches can afford
    // more rings (a ring is max contour length / 16 words)
    l->long_walks_per_plane = l->max_batch <= 16 ? 8192 : l->max_batch <= 128 ? 2048 : 1024;
}

static int validate_params(arucohip_handle* h, const arucohip_params_t* p) {
    // CV_Assert of setMinMaxSize (markerdetector.cpp:1032-1034) and setWarpSize (:1048)
    if (!(p->min_size > 0 && p->min_size <= 1) || !(p->max_size > 0 && p->max_size <= 1) || !(p->min_size < p->max_size))
        return fail(h, ARUCOHIP_E_INVALID, "setMinMaxSize: need 0 < min < max <= 1");
    if (p->warp_size < 10) return fail(h, ARUCOHIP_E_INVALID, "setWarpSize: need >= 10");
    if (p->warp_size > 128) return fail(h, ARUCOHIP_E_UNSUPPORTED, "warp size > 128 not supported");
    if (p->thres_method < ARUCOHIP_THRES_FIXED || p->thres_method > ARUCOHIP_THRES_CANNY) return fail(h, ARUCOHIP_E_INVALID, "bad threshold method");
    if (p->corner_method < ARUCOHIP_CORNER_NONE || p->corner_method > ARUCOHIP_CORNER_LINES) return fail(h, ARUCOHIP_E_INVALID, "bad corner method");
    if (p->use_locked_corners && (p->corner_method == ARUCOHIP_CORNER_HARRIS || p->corner_method == ARUCOHIP_CORNER_SUBPIX) &&
        ((int)p->thres_param1 < 1 || (int)p->thres_param1 > 31))
        return fail(h, ARUCOHIP_E_UNSUPPORTED, "locked corners: window (thres_param1) outside 1..31");
    if (p->decoder_kind < ARUCOHIP_DECODER_FIDUCIAL_5X5 || p->decoder_kind > ARUCOHIP_DECODER_USER) return fail(h, ARUCOHIP_E_INVALID, "bad decoder kind");
    if (p->thres_param1_range < 0 || 2 * p->thres_param1_range + 1 > 16) return fail(h, ARUCOHIP_E_UNSUPPORTED, "threshold range too large");
    if (p->corner_method == ARUCOHIP_CORNER_SUBPIX && (int)p->thres_param1 > 15) return fail(h, ARUCOHIP_E_UNSUPPORTED, "SUBPIX window > 15");
    if (p->corner_method == ARUCOHIP_CORNER_SUBPIX && (int)p->thres_param1 < 1) return fail(h, ARUCOHIP_E_INVALID, "SUBPIX window < 1");
    return ARUCOHIP_OK;
}

extern "C" void arucohip_destroy(arucohip_handle* h);
static void free_all(arucohip_handle* h) {
    hipSetDevice(h->device);
    for (auto* k : h->kids) arucohip_destroy(k);
    h->kids.clear();
    for (auto* l : h->lanes) arucohip_destroy(l);
    h->lanes.clear();
    if (h->retry) arucohip_destroy(h->retry);
    h->retry = nullptr;
    if (h->fgraph.exec) hipGraphExecDestroy(h->fgraph.exec);
    h->fgraph.exec = nullptr;
    if (h->ev_submit) hipEventDestroy(h->ev_submit);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_thr) hipEventDestroy(h->ev_thr);
    if (h->ev_wfork) hipEventDestroy(h->ev_wfork);
    if (h->ev_wjoin) hipEventDestroy(h->ev_wjoin);
    if (h->side_stream) hipStreamDestroy(h->side_stream);
    for (auto& e : h->ev_join)
        if (e) hipEventDestroy(e);
    hipFree(h->buf.thr_stamps), hipFree(h->buf.thr_acc), hipFree(h->buf.thres_edge);
    hipFree(h->buf.thres), hipFree(h->buf.tiles), hipFree(h->buf.tile_bits), hipFree(h->buf.raw), hipFree(h->buf.trig), hipFree(h->buf.gen_buf), hipFree(h->zero_block), hipFree(h->buf.cdesc), hipFree(h->buf.pool);
    hipFree(h->buf.quads), hipFree(h->buf.cands), hipFree(h->buf.ncands), hipFree(h->buf.cand_list), hipFree(h->buf.iM), hipFree(h->buf.hist), hipFree(h->buf.othr), hipFree(h->buf.markers), hipFree(h->buf.nmarkers), hipFree(h->buf.marker_list);
    hipFree(h->buf.walk_scratch), hipFree(h->buf.node), hipFree(h->buf.skipn), hipFree(h->buf.stamp), hipFree(h->buf.hash), hipFree(h->buf.patches), hipFree(h->d_gray), hipFree(h->d_bgr), hipFree(h->d_erode), hipFree(h->d_canny), hipFree(h->d_umap_xy), hipFree(h->d_umap_f), hipFree(h->d_undist), hipFree(h->d_hrm), hipFree(h->d_user_dec), hipFree(h->d_small_f), hipFree(h->d_small_d), hipFree(h->d_small_i), hipFree(h->d_patch), hipFree(h->d_board), hipFree(h->d_gl);
    if (h->hu_list) hipHostFree(h->hu_list);
    if (h->hu_patches) hipHostFree(h->hu_patches);
    if (h->h_markers) hipHostFree(h->h_markers);
    if (h->h_n) hipHostFree(h->h_n);
    if (h->h_counters) hipHostFree(h->h_counters);
    for (auto& set : h->ev)
        for (auto& e : set)
            if (e) hipEventDestroy(e);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
}

static thread_local bool g_creating_child = false;

// The one-frame handle of arucohip_detect_batch_retry_overflowed copies parameters, dictionary and decoder callback when it is made: whenever
// one of them changes it is dropped and the next retry builds a fresh one (a stale copy would decode retried frames with the old dictionary).
static void drop_retry(arucohip_handle* h) {
    if (h->retry) arucohip_destroy(h->retry);
    h->retry = nullptr, h->retry_mult = 0;
}

int arucohip_create_ex(const arucohip_params_t* params, int device, const arucohip_limits_t* lim, arucohip_handle** out) {
    if (!out || !lim) return ARUCOHIP_E_INVALID;
    *out = nullptr;
    if (lim->max_width < 32 || lim->max_height < 32 || lim->max_width > 16383 || lim->max_height > 16383 || lim->max_batch < 1 ||
        lim->max_thres_planes < 1 || lim->max_thres_planes > 16 || lim->candidates_per_frame > 512 || lim->markers_per_frame > 256 ||
        (long)lim->max_width * lim->max_height > (1L << 26) /* Quad::key holds a 26-bit raster index */)
        return ARUCOHIP_E_INVALID;
    arucohip_handle* h = new arucohip_handle();
    h->device = device;
    h->lim = *lim;
    if (params)
        h->params = *params;
    else
        arucohip_default_params(&h->params);
    int rc = validate_params(h, &h->params);
    if (rc != ARUCOHIP_OK) {
        delete h;
        return rc;
    }
    auto bail = [&](hipError_t e) {
        fprintf(stderr, "arucohip_create: %s\n", hipGetErrorString(e));
        free_all(h);
        delete h;
        return ARUCOHIP_E_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
    h->stream = h->own_stream;
    {
        // workers: one by default; ARUCOHIP_STREAMS = 2..8 cuts large batches into chunks on separate streams (copies of host
        // frames then overlap the kernels). With the late walker generations on their own side stream a second chunk stream
        // no longer gains anything for device-resident frames (1 stream 208 k fps, 2 streams 208 k at 1024 1080p frames).
        const double mpx = (double)lim->max_batch * lim->max_width * lim->max_height / (128.0 * 1024 * 1024);
        int ns = 1;
        (void)mpx;
        if (const char* es = getenv("ARUCOHIP_STREAMS")) ns = std::min(8, std::max(1, atoi(es)));
        if (g_creating_child) ns = 1;
        h->is_child = g_creating_child;
        h->nsub = std::min(ns, lim->max_batch);
        h->cap_frames = (lim->max_batch + h->nsub - 1) / h->nsub;
    }
    const size_t F = h->cap_frames, P = F * lim->max_thres_planes, px = (size_t)lim->max_width * lim->max_height;
    Buffers& b = h->buf;
    b.tune = read_tuning();
    {
        const char* eg = getenv("ARUCOHIP_GRAPH");
        h->fgraph.disabled = (eg && *eg && atoi(eg) == 0) ? 1 : 0;
    }
    b.cap_raw = (uint32_t)lim->triggers_per_frame;
    b.cap_trig = (uint32_t)std::max(lim->triggers_per_frame, 8192);   // two halves: outer starts, hole starts
    b.long_cap = (uint32_t)std::min(std::max(lim->long_walks_per_plane, 64), 1 << 16);
    h->lim.long_walks_per_plane = (int32_t)b.long_cap;
    b.cap_cdesc = (uint32_t)lim->contours_per_frame;   // per plane
    b.cap_pool = (uint32_t)lim->points_per_frame;       // per plane
    // pool offsets are 32-bit (ContourDesc::pool_off = plane * cap_pool + offset)
    if (P * (size_t)b.cap_pool > 0xFFFFFFF0ull) b.cap_pool = (uint32_t)(0xFFFFFFF0ull / P);
    b.cap_quads = std::min(lim->candidates_per_frame * 2, 512);
    b.cap_cands = lim->candidates_per_frame;
    b.cap_markers = lim->markers_per_frame;
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return bail(e)
    ALLOC(b.thres, P * px);
    ALLOC(b.thres_edge, P * thres_edge_stride(lim->max_width, lim->max_height));
    {   // timing stamps of the wide threshold kernel: its finest grid is one wave per 1024-px strip and 16 rows
        const size_t waves = (size_t)tile_strips(lim->max_width) * ((lim->max_height + 15) / 16) * F;
        ALLOC(b.thr_stamps, 2 * waves * sizeof(uint64_t));
        ALLOC(b.thr_acc, 2 * sizeof(uint64_t));
        b.thr_stamp_on = 0;
    }
    const size_t bits_bytes = P * (size_t)tiles_x(lim->max_width) * tiles_y(lim->max_height) * sizeof(uint64_t) + 64;
    ALLOC(b.tiles, bits_bytes);
    if ((e = hipMemset(b.tiles, 0, bits_bytes)) != hipSuccess) return bail(e);   // pad tiles must read as zero
    ALLOC(b.tile_bits, P * (size_t)tiles_y(lim->max_height) * 2 * tile_strips(lim->max_width) * sizeof(uint64_t));
    h->bits_bytes = bits_bytes;
    ALLOC(b.trig, P * (size_t)b.cap_trig * sizeof(uint2));
    {
        // contour pipeline: ARUCOHIP_CONTOURS = walkers | segments; default by handle shape. The per-candidate walkers win on
        // batches; a single small frame is a chain of up to max-contour dependent border steps for them, which the waypoint
        // segments cut (bench.py latency leg, 1000 calls: 640x480 stills 0.67-0.72 ms vs 0.49-0.52 ms; one 1080p frame 0.80 vs 0.51 since round 4:
        // per-plane workgroup counts scaled for one frame, the run rule read from the lane's block)
        const char* mode = getenv("ARUCOHIP_CONTOURS");
        b.seg_mode = mode ? std::string(mode) == "segments" : (lim->max_batch == 1 && (long)lim->max_width * lim->max_height <= 2048L * 1536L);
        const char* gs = getenv("ARUCOHIP_GRID");
        int grid = gs ? atoi(gs) : 16;
        if (grid != 1 && grid != 2 && grid != 4 && grid != 8 && grid != 16 && grid != 32) grid = 8;
        b.grid_mask = grid - 1;
        uint32_t hs = 1;
        while (hs < 2u * b.cap_raw) hs <<= 1;
        b.hash_mask = hs - 1;
    }
    if (b.seg_mode) {   // waypoint-segment pipeline only: a walker handle of 1024 frames would carry 5 GB of these for nothing
        ALLOC(b.raw, P * (size_t)b.cap_raw * sizeof(uint2));
        ALLOC(b.node, P * (size_t)b.cap_raw * sizeof(uint4));
        if (lim->max_batch <= 2) ALLOC(b.skipn, P * (size_t)b.cap_raw * sizeof(uint4));
        ALLOC(b.stamp, P * (size_t)b.cap_raw * sizeof(unsigned long long));
        ALLOC(b.hash, P * (size_t)(b.hash_mask + 1) * sizeof(uint32_t));
    }
    ALLOC(b.gen_buf, ((P + 7) / 8 * 8) * (size_t)b.long_cap * 4 * 20);   // [2 kinds][2 parities][planes rounded up to 8 * long_cap] walk states (16 B) + ring ids (4 B)
    ALLOC(b.cdesc, P * (size_t)b.cap_cdesc * sizeof(ContourDesc));
    ALLOC(b.pool, P * (size_t)b.cap_pool * sizeof(short2));
    ALLOC(b.quads, F * b.cap_quads * sizeof(Quad));
    ALLOC(b.cands, F * b.cap_cands * sizeof(Cand));
    ALLOC(b.ncands, F * sizeof(int32_t));
    b.cap_flat = (uint32_t)std::min<size_t>(F * (size_t)std::min(b.cap_cands, 96), 65535u * 16u);
    ALLOC(b.cand_list, (size_t)b.cap_flat * sizeof(uint32_t));
    ALLOC(b.iM, (size_t)b.cap_flat * 9 * sizeof(double));
    ALLOC(b.hist, (size_t)b.cap_flat * 256 * sizeof(uint16_t));
    ALLOC(b.othr, (size_t)b.cap_flat * sizeof(int32_t));
    ALLOC(b.markers, (F * b.cap_markers + 1) * sizeof(arucohip_marker_t));   // + the header slot of a one-frame call (k_finalize.hip: write_hdr)
    ALLOC(b.nmarkers, F * sizeof(int32_t));
    ALLOC(b.marker_list, F * (size_t)b.cap_markers * sizeof(uint32_t));
    {
        // every counter a batch starts from zero with lives in one block: one memset per batch instead of five (a single frame's
        // call is a chain of ~25 short operations, each memset was 6 us of it)
        const size_t w_cnt = (CNT_FIXED + F + 31) & ~(size_t)31, w_plane = P * TRIG_CNT_STRIDE;
        h->zero_words = w_cnt + GEN_CNT_WORDS + 3 * w_plane;
        ALLOC(h->zero_block, h->zero_words * sizeof(uint32_t));
        b.counters = h->zero_block;
        b.gen_cnt = b.counters + w_cnt;
        b.trig_cnt = b.gen_cnt + GEN_CNT_WORDS;
        b.raw_cnt = b.trig_cnt + w_plane;
        b.ring_cnt = b.raw_cnt + w_plane;
    }
    ALLOC(h->d_small_f, 8192 * sizeof(float));
    ALLOC(h->d_small_d, 64 * sizeof(double));
    ALLOC(h->d_small_i, 64 * sizeof(int));
    ALLOC(h->d_patch, 128 * 128);
#undef ALLOC
    if ((e = hipHostMalloc((void**)&h->h_markers, (F * b.cap_markers + 1) * sizeof(arucohip_marker_t))) != hipSuccess) return bail(e);
    if ((e = hipHostMalloc((void**)&h->h_n, F * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipHostMalloc((void**)&h->h_counters, (CNT_FIXED + F) * sizeof(uint32_t))) != hipSuccess) return bail(e);
    for (auto& set : h->ev)
        for (auto& ev : set)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e);
    if ((e = hipEventCreateWithFlags(&h->ev_thr, hipEventDisableTiming)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
    if ((e = hipEventCreateWithFlags(&h->ev_wfork, hipEventDisableTiming)) != hipSuccess) return bail(e);
    if ((e = hipEventCreateWithFlags(&h->ev_wjoin, hipEventDisableTiming)) != hipSuccess) return bail(e);
    if (h->nsub > 1) {
        if ((e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail(e);
        for (int i = 0; i < h->nsub - 1; i++) {
            if ((e = hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming)) != hipSuccess) return bail(e);
            arucohip_limits_t kl = *lim;
            kl.max_batch = h->cap_frames;
            arucohip_handle* kid = nullptr;
            g_creating_child = true;
            int krc = arucohip_create_ex(&h->params, device, &kl, &kid);
            g_creating_child = false;
            if (krc != ARUCOHIP_OK) {
                free_all(h);
                delete h;
                return krc;
            }
            h->kids.push_back(kid);
        }
    }
    *out = h;
    return ARUCOHIP_OK;
}

int arucohip_create(const arucohip_params_t* params, int device, int max_width, int max_height, int max_batch, arucohip_handle** out) {
    arucohip_limits_t l;
    arucohip_default_limits(&l, max_width, max_height, max_batch);
    if (params) l.max_thres_planes = std::max(1, 2 * params->thres_param1_range + 1);
    return arucohip_create_ex(params, device, &l, out);
}

void arucohip_destroy(arucohip_handle* h) {
    if (!h) return;
    free_all(h);
    delete h;
}

int arucohip_set_params(arucohip_handle* h, const arucohip_params_t* p) {
    if (!h || !p) return ARUCOHIP_E_INVALID;
    int rc = validate_params(h, p);
    if (rc != ARUCOHIP_OK) return rc;
    if (2 * p->thres_param1_range + 1 > h->lim.max_thres_planes)
        return fail(h, ARUCOHIP_E_INVALID, "threshold range exceeds the planes this handle was created with");
    h->params = *p;
    for (auto* k : h->kids) k->params = *p;
    for (auto* l : h->lanes) {
        l->params = *p;
        for (auto* k : l->kids) k->params = *p;
    }
    drop_retry(h);
    return ARUCOHIP_OK;
}

int arucohip_get_params(const arucohip_handle* h, arucohip_params_t* p) {
    if (!h || !p) return ARUCOHIP_E_INVALID;
    *p = h->params;
    return ARUCOHIP_OK;
}

const char* arucohip_last_error_string(const arucohip_handle* h) { return h ? h->err.c_str() : "null handle"; }

int arucohip_set_stream(arucohip_handle* h, void* s) {
    if (!h) return ARUCOHIP_E_INVALID;
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return ARUCOHIP_OK;
}
void* arucohip_get_stream(arucohip_handle* h) { return h ? (void*)h->stream : nullptr; }

int arucohip_synchronize(arucohip_handle* h) {
    if (!h) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ARUCOHIP_OK;
}

int arucohip_wait_event(arucohip_handle* h, void* ev) {
    if (!h || !ev) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    // a submit forks its lane from the handle's stream (ev_submit), so one wait here orders both forms behind the producer
    HIPCHK(h, hipStreamWaitEvent(h->stream, (hipEvent_t)ev, 0));
    return ARUCOHIP_OK;
}

int arucohip_enable_timing(arucohip_handle* h, int on) {
    if (!h) return ARUCOHIP_E_INVALID;
    h->timing = on == 1;      // on == 2: only the threshold kernel's device-clock stamps, no hipEvents between the kernels (the launches then overlap
                              // with the other batches exactly as in an uninstrumented run: bench.py's replica pass)
    h->tsets = 0;
    for (auto* k : h->kids) k->timing = h->timing, k->tsets = 0;
    // device-clock stamps of the wide threshold kernel (k_threshold.hip): taken while timing is on, accumulators restart with it
    auto arm = [&](arucohip_handle* x) {
        x->buf.thr_stamp_on = on != 0 && x->buf.thr_stamps && x->buf.thr_acc;
        if (on && x->buf.thr_acc) {
            hipSetDevice(x->device);
            (void)hipStreamSynchronize(x->stream);
            (void)hipMemset(x->buf.thr_acc, 0, 2 * sizeof(uint64_t));
        }
    };
    arm(h);
    for (auto* k : h->kids) arm(k);
    for (auto* l : h->lanes) arucohip_enable_timing(l, on);
    return ARUCOHIP_OK;
}
// synchronises the stream and averages the per-kernel event intervals of the batches since enable/reset
// (with sub-batch pipelining: the average over the launches of all workers, each launch covering one chunk)
static void collect_times(arucohip_handle* h) {
    for (int k = 0; k < K_COUNT; k++) h->kernel_ms[k] = 0;
    hipSetDevice(h->device);
    int total = 0;
    auto add = [&](arucohip_handle* w) {
        int n = std::min(w->tsets, TSETS);
        if (n <= 0) return;
        if (hipStreamSynchronize(w->stream) != hipSuccess) return;
        for (int s = 0; s < n; s++)
            for (int k = 0; k < K_COUNT; k++) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, w->ev[s][k], w->ev[s][k + 1]) == hipSuccess) h->kernel_ms[k] += ms;
            }
        total += n;
    };
    add(h);
    for (auto* k : h->kids) add(k);
    for (auto* l : h->lanes) {
        add(l);
        for (auto* k : l->kids) add(k);
    }
    if (total > 0)
        for (int k = 0; k < K_COUNT; k++) h->kernel_ms[k] /= total;
}
const char* arucohip_stage_name(int i) { return (i >= 0 && i < STAGE_COUNT) ? kStageNames[i] : ""; }
int arucohip_stage_times(arucohip_handle* h, float* ms, int cap) {
    if (!h) return 0;
    collect_times(h);
    for (int i = 0; i < STAGE_COUNT && i < cap; i++) ms[i] = 0;
    for (int k = 0; k < K_COUNT; k++)
        if (kKernelStage[k] < cap) ms[kKernelStage[k]] += h->kernel_ms[k];
    return STAGE_COUNT;
}
const char* arucohip_kernel_name(int i) { return (i >= 0 && i < K_COUNT) ? kKernelNames[i] : ""; }
int arucohip_threshold_exec_ms(arucohip_handle* h, double* total_ms, int* launches) {
    if (!h || !total_ms || !launches) return ARUCOHIP_E_INVALID;
    hipSetDevice(h->device);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) return fail(h, ARUCOHIP_E_HIP, "no wall clock rate");
    unsigned long long ticks = 0, n = 0;
    auto add = [&](arucohip_handle* w) -> int {
        if (!w->buf.thr_acc) return ARUCOHIP_OK;
        unsigned long long v[2] = {0, 0};
        HIPCHK(h, hipStreamSynchronize(w->stream));
        HIPCHK(h, hipMemcpy(v, w->buf.thr_acc, sizeof(v), hipMemcpyDeviceToHost));
        ticks += v[0], n += v[1];
        return ARUCOHIP_OK;
    };
    int rc = add(h);
    for (auto* k : h->kids) if (rc == ARUCOHIP_OK) rc = add(k);
    for (auto* l : h->lanes) {
        if (rc == ARUCOHIP_OK) rc = add(l);
        for (auto* k : l->kids) if (rc == ARUCOHIP_OK) rc = add(k);
    }
    if (rc != ARUCOHIP_OK) return rc;
    *total_ms = (double)ticks / (double)khz, *launches = (int)n;
    return ARUCOHIP_OK;
}

int arucohip_kernel_times(arucohip_handle* h, float* ms, int cap) {
    if (!h) return 0;
    collect_times(h);
    for (int i = 0; i < K_COUNT && i < cap; i++) ms[i] = h->kernel_ms[i];
    return K_COUNT;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
static int make_detect_params(arucohip_handle* h, int W, int H, DetectParams* dp) {
    const arucohip_params_t& p = h->params;
    std::memset(dp, 0, sizeof(*dp));
    dp->thres_method = p.thres_method;
    dp->nthr = 2 * p.thres_param1_range + 1;
    if (dp->nthr > h->lim.max_thres_planes) return fail(h, ARUCOHIP_E_INVALID, "threshold range exceeds handle planes");
    for (int i = 0; i < dp->nthr; i++) {
        // markerdetector.cpp:325-333 (step is the range itself) and the odd/>=3 fix-up of :657-660
        double p1 = dp->nthr == 1 ? p.thres_param1 : p.thres_param1 - p.thres_param1_range + (double)p.thres_param1_range * i;
        if (p.thres_method == ARUCOHIP_THRES_ADPT) {
            if (p1 < 3)
                p1 = 3;
            else if (((int)p1) % 2 != 1)
                p1 = (int)(p1 + 1);
            dp->block[i] = (int)p1;
            if (dp->block[i] > 31) return fail(h, ARUCOHIP_E_UNSUPPORTED, "adaptive threshold block size > 31");
        }
        dp->p1[i] = p1;
    }
    dp->idelta = (int)std::floor(p.thres_param2);
    dp->corner_method = p.corner_method;
    dp->warp_size = p.warp_size;
    dp->min_contour = (int)(p.min_size * std::max(W, H) * 4);   // :500-501, float arithmetic
    dp->max_contour = (int)(p.max_size * std::max(W, H) * 4);
    if (dp->max_contour > 16383) dp->max_contour = 16383;       // offsets inside a border are 14-bit fields
    // :433-434 Rect(Point(size)*t, Point(size)*(1-t)) with cvRound
    int x1 = (int)lrintf((float)W * p.border_dist), y1 = (int)lrintf((float)H * p.border_dist);
    int x2 = (int)lrintf((float)W * (1.0f - p.border_dist)), y2 = (int)lrintf((float)H * (1.0f - p.border_dist));
    dp->bx0 = std::min(x1, x2), dp->by0 = std::min(y1, y2);
    dp->bx1 = std::max(x1, x2), dp->by1 = std::max(y1, y2);
    dp->subpix_win = (int)p.thres_param1;
    dp->locked = p.use_locked_corners != 0, dp->locked_wsize = (int)p.thres_param1;   // findCornerMaxima(Corners, grey, _thresParam1)
    dp->decoder = p.decoder_kind;
    if (p.decoder_kind == ARUCOHIP_DECODER_HRM) {
        if (!h->d_hrm || h->hrm_count <= 0) return fail(h, ARUCOHIP_E_INVALID, "decoder HRM without a dictionary (arucohip_set_dictionary)");
        if (p.warp_size < 2 * (h->hrm_n + 2)) return fail(h, ARUCOHIP_E_INVALID, "warp size too small for the dictionary's markers");
        dp->hrm_n = h->hrm_n, dp->hrm_count = h->hrm_count, dp->hrm_codes = h->d_hrm;
        dp->hrm_correction = (uint32_t)(h->hrm_rate * (float)((h->hrm_tau0 - 1) / 2));   // highlyreliablemarkers.cpp:318
    }
    if (p.decoder_kind == ARUCOHIP_DECODER_USER && !h->decoder_fn)
        return fail(h, ARUCOHIP_E_INVALID, "decoder USER without a callback (arucohip_set_decoder_callback)");
    return ARUCOHIP_OK;
}

static int make_cam(arucohip_handle* h, const float* K, const float* dist, int ndist, float marker_size, int y_perp, CamModel* cam) {
    std::memset(cam, 0, sizeof(*cam));
    if (ndist < 0 || ndist > 8) return fail(h, ARUCOHIP_E_INVALID, "ndist must be 0..8");
    cam->has_K = K != nullptr;
    if (K)
        for (int i = 0; i < 9; i++) cam->K[i] = K[i];
    cam->has_dist = dist != nullptr && ndist > 0;
    if (cam->has_dist)
        for (int i = 0; i < ndist; i++) cam->k[i] = (double)dist[i];
    cam->marker_size = marker_size;
    cam->y_perp = y_perp;
    return ARUCOHIP_OK;
}

static int check_status(arucohip_handle* h, uint32_t st) {
    if (!st) return ARUCOHIP_OK;
    char msg[256];
    snprintf(msg, sizeof(msg), "device list overflow:%s%s%s%s%s%s", (st & ST_TRIG_OVERFLOW) ? " triggers" : "",
             (st & ST_CDESC_OVERFLOW) ? " contours" : "", (st & ST_POOL_OVERFLOW) ? " points" : "",
             (st & ST_QUAD_OVERFLOW) ? " quads" : "", (st & ST_CAND_OVERFLOW) ? " candidates" : "",
             (st & ST_MARKER_OVERFLOW) ? " markers" : "");
    if (st & ST_SEGMENT_ERROR) snprintf(msg + strlen(msg), sizeof(msg) - strlen(msg), " segment-link");
    h->err = msg;
    return ARUCOHIP_E_OVERFLOW;
}

// the pad word of every bit-image row must be zero; its position depends on the frame width
static int ensure_bits_geometry(arucohip_handle* h, int W, int H) {
    if (h->bits_w == W && h->bits_h == H) return ARUCOHIP_OK;
    HIPCHK(h, hipMemsetAsync(h->buf.tiles, 0, h->bits_bytes, h->stream));
    h->bits_w = W, h->bits_h = H;
    return ARUCOHIP_OK;
}

// canonical patches of the decode stage: cap_flat * warp_size^2 bytes
static int ensure_patches(arucohip_handle* h, const DetectParams& dp) {
    size_t need = (size_t)h->buf.cap_flat * dp.warp_size * dp.warp_size;
    if (need <= h->patch_bytes) return ARUCOHIP_OK;
    if (h->buf.patches) HIPCHK(h, hipFree(h->buf.patches));
    h->buf.patches = nullptr, h->patch_bytes = 0;
    HIPCHK(h, hipMalloc((void**)&h->buf.patches, need));
    h->patch_bytes = need;
    return ARUCOHIP_OK;
}

// the long walks keep their checkpoint rings in HBM; (re)size the space for this batch
static int ensure_walk_scratch(arucohip_handle* h, int nplanes, const DetectParams& dp) {
    size_t need = walk_scratch_words(nplanes, dp, h->buf.long_cap);
    if (need > 0xFFFFFFF0ull) return fail(h, ARUCOHIP_E_CAPACITY, "batch too large for 32-bit checkpoint offsets: fewer frames per batch or a smaller max size");
    if (need <= h->scratch_words) return ARUCOHIP_OK;
    if (h->buf.walk_scratch) HIPCHK(h, hipFree(h->buf.walk_scratch));
    h->buf.walk_scratch = nullptr, h->scratch_words = 0;
    HIPCHK(h, hipMalloc((void**)&h->buf.walk_scratch, need * sizeof(uint32_t)));
    h->scratch_words = need;
    return ARUCOHIP_OK;
}

// Plugin boundary (markerdetector.h:65-78, :243-245): the caller's decoder runs on the host between the device's warp and
// the rest of the pipeline. Candidates are decoded frame by frame in detectRectangles order like the loop at
// markerdetector.cpp:350-368.
static int user_decode_stage(arucohip_handle* h, const DetectParams& dp) {
    hipStream_t s = h->stream;
    const Buffers& b = h->buf;
    // pinned staging owned by the handle, grown on demand: the steady state of a stream of calls allocates nothing
    const size_t npx = (size_t)dp.warp_size * dp.warp_size;
    auto pinned = [&](void** p, size_t* have, size_t need) -> int {
        if (need <= *have) return ARUCOHIP_OK;
        if (*p) HIPCHK(h, hipHostFree(*p));
        *p = nullptr, *have = 0;
        HIPCHK(h, hipHostMalloc(p, need));
        *have = need;
        return ARUCOHIP_OK;
    };
    int rc;
    if ((rc = pinned((void**)&h->hu_list, &h->hu_list_bytes, (size_t)b.cap_flat * (sizeof(uint32_t) + sizeof(int2) + sizeof(uint32_t)) + sizeof(uint32_t)))) return rc;
    uint32_t* list = h->hu_list;                                   // [cap_flat] frame << 16 | index
    int2* dec = (int2*)(list + b.cap_flat);                        // [cap_flat] {id, nRotations}
    uint32_t* order = (uint32_t*)(dec + b.cap_flat);               // [cap_flat] + the candidate count behind it
    uint32_t* ncand_p = order + b.cap_flat;
    HIPCHK(h, hipMemcpyAsync(ncand_p, b.counters + CNT_NCAND, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    const uint32_t n = std::min(*ncand_p, b.cap_flat);
    if (!n) return ARUCOHIP_OK;
    if ((rc = pinned((void**)&h->hu_patches, &h->hu_patch_bytes, (size_t)n * npx + npx))) return rc;
    uint8_t* patches = h->hu_patches;
    uint8_t* scratch = patches + (size_t)n * npx;
    HIPCHK(h, hipMemcpyAsync(list, b.cand_list, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(patches, b.patches, (size_t)n * npx, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::sort(order, order + n, [&](uint32_t x, uint32_t y) { return list[x] < list[y]; });   // frame << 16 | index
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t i = order[k];
        std::memcpy(scratch, patches + (size_t)i * npx, npx);
        int nrot = 0;   // the reference leaves it uninitialised (markerdetector.cpp:354); 0 is the intent
        const int id = h->decoder_fn(h->decoder_user, scratch, dp.warp_size, &nrot);
        dec[i] = make_int2(id < 0 ? -1 : id, nrot & 3);
    }
    if (!h->d_user_dec) HIPCHK(h, hipMalloc((void**)&h->d_user_dec, (size_t)b.cap_flat * sizeof(int2)));   // once per handle
    HIPCHK(h, hipMemcpyAsync(h->d_user_dec, dec, n * sizeof(int2), hipMemcpyHostToDevice, s));
    launch_set_decoded(s, b, n, h->d_user_dec);
    // no synchronise: the staging belongs to the handle, and the next call that touches it synchronises the stream first (the count above)
    return ARUCOHIP_OK;
}

static void run_walkers_and_quads(arucohip_handle* h, hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& dp) {
    WalkFork fk{h->buf.tune.walk_fork ? h->side_stream : nullptr, h->ev_wfork, h->ev_wjoin, nullptr};
    const bool forked = launch_walkers(s, fk, g, nframes * dp.nthr, dp, h->buf);
    launch_contour_quads(s, g, nframes, dp, h->buf, forked ? 1 : 0);
    if (forked) {
        (void)hipStreamWaitEvent(s, h->ev_wjoin, 0);
        launch_contour_quads(s, g, nframes, dp, h->buf, 2);
    }
}

// runs kernels 2..8 after the masks and start candidates exist
static void run_rectangles(arucohip_handle* h, const FrameGeom& g, int nframes, const DetectParams& dp) {
    if (h->buf.seg_mode) {
        launch_start_candidates(h->stream, g, nframes * dp.nthr, h->buf);   // also clears the planes' key -> node tables
        launch_segments(h->stream, g, nframes * dp.nthr, dp, h->buf);
    } else {
        launch_start_candidates(h->stream, g, nframes * dp.nthr, h->buf, dp.min_contour);
        run_walkers_and_quads(h, h->stream, g, nframes, dp);
    }
    if (h->buf.seg_mode) launch_contour_quads(h->stream, g, nframes, dp, h->buf);
    launch_frame_candidates(h->stream, g, nframes, dp, h->buf);
}

static int grow(arucohip_handle* h, uint8_t** buf, size_t* have, size_t need);

// threshold stage of any method into buf.thres / buf.tiles (+ bitmap). CANNY (markerdetector.cpp:667-676) blocks the host while its
// hysteresis converges.
// want_bytes: the caller reads buf.thres right away (stage entry point, erosion); otherwise the byte image may be left as tiles + border
// lines (h->thres_bytes says which) and arucohip_get_thresholded expands the plane it is asked for.
static int run_threshold(arucohip_handle* h, hipStream_t s, const uint8_t* gray_dev, const FrameGeom& g, int nframes, const DetectParams& dp, bool want_bytes) {
    const Buffers& b = h->buf;
    h->thres_bytes = true;
    if (dp.thres_method != ARUCOHIP_THRES_CANNY) {
        const bool lazy = launch_threshold(s, gray_dev, g, nframes, dp, b, b.tune.thres_lazy && !want_bytes);
        h->thres_bytes = !lazy;
        return ARUCOHIP_OK;
    }
    const size_t ntiles = (size_t)nframes * dp.nthr * ((g.width + 7) / 8) * ((g.height + 7) / 8);
    int rc = grow(h, &h->d_canny, &h->canny_bytes, 2 * ntiles * sizeof(uint64_t) + 64);
    if (rc) return rc;
    uint64_t* surv = (uint64_t*)h->d_canny;
    uint64_t* edge = surv + ntiles;
    if (launch_canny(s, gray_dev, g, nframes, dp.nthr, b, surv, edge, (uint32_t*)(edge + ntiles))) return fail(h, ARUCOHIP_E_HIP, "CANNY kernels failed");
    FrameGeom tg = g;
    tg.row_stride = (size_t)g.width, tg.frame_stride = (size_t)g.width * g.height;
    launch_binary_planes(s, b.thres, tg, nframes * dp.nthr, b);   // contour tiles + bitmap from the edge image
    return ARUCOHIP_OK;
}

static int detect_core(arucohip_handle* h, const uint8_t* gray_dev, const FrameGeom& g, int nframes, const DetectParams& dp, const CamModel& cam) {
    hipStream_t s = h->stream;
    Buffers& b = h->buf;
    {
        int rc_ = ensure_walk_scratch(h, nframes * dp.nthr, dp);
        if (rc_) return rc_;
        if ((rc_ = ensure_bits_geometry(h, g.width, g.height))) return rc_;
        if ((rc_ = ensure_patches(h, dp))) return rc_;
    }
    HIPCHK(h, hipMemsetAsync(h->zero_block, 0, h->zero_words * sizeof(uint32_t), s));
    hipEvent_t* ev = h->ev[h->tsets % TSETS];
    const bool tm = h->timing;
#define MARK(i) do { if (tm) (void)hipEventRecord(ev[i], s); } while (0)
    if (h->wait_thr) HIPCHK(h, hipStreamWaitEvent(s, h->wait_thr, 0));   // threshold kernels of the lanes run one after the other
    MARK(K_THRESHOLD);   // after that wait: the interval is this batch's own threshold kernel
    {
        const int rc_ = run_threshold(h, s, gray_dev, g, nframes, dp, false);
        if (rc_) return rc_;
    }
    if (h->params.erode) {
        // on the bit tiles where the byte image was left out (the default path), on the bytes otherwise
        const bool on_tiles = !h->thres_bytes;
        int rc_ = grow(h, &h->d_erode, &h->erode_bytes, on_tiles ? erode_tiles_tmp_bytes(g, nframes * dp.nthr) : (size_t)nframes * dp.nthr * g.width * g.height);
        if (rc_) return rc_;
        if (on_tiles)
            launch_erode_tiles(s, g, nframes * dp.nthr, b, h->d_erode);
        else
            launch_erode(s, g, nframes * dp.nthr, b, h->d_erode);
    }
    if (h->ev_thr) HIPCHK(h, hipEventRecord(h->ev_thr, s));
    MARK(K_FILTER);
    if (b.seg_mode) {
        launch_start_candidates(s, g, nframes * dp.nthr, b);   // also clears the planes' key -> node tables
        MARK(K_WALKERS);
        launch_segments(s, g, nframes * dp.nthr, dp, b);
        MARK(K_WALKERS_LONG);
        MARK(K_CONTOUR_QUADS);
        launch_contour_quads(s, g, nframes, dp, b);
    } else {
        if (RUN_STAGE(b.tune, 1)) launch_start_candidates(s, g, nframes * dp.nthr, b, dp.min_contour);
        MARK(K_WALKERS);
        // walkers; their late generations run on the side stream under the first quad pass (the contour_quad mark sits at the fork)
        WalkFork fk{b.tune.walk_fork ? h->side_stream : nullptr, h->ev_wfork, h->ev_wjoin, tm ? ev[K_WALKERS_LONG] : nullptr};
        const bool forked = RUN_STAGE(b.tune, 2) ? launch_walkers(s, fk, g, nframes * dp.nthr, dp, b) : false;
        MARK(K_CONTOUR_QUADS);
        if (RUN_STAGE(b.tune, 4)) launch_contour_quads(s, g, nframes, dp, b, forked ? 1 : 0);
        if (forked) {
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_wjoin, 0));
            if (RUN_STAGE(b.tune, 4)) launch_contour_quads(s, g, nframes, dp, b, 2);
        }
    }
    MARK(K_FRAME_CANDS);
    if (RUN_STAGE(b.tune, 5)) launch_frame_candidates(s, g, nframes, dp, b);
    MARK(K_DECODE);
    // built-in 5x5 decoder: the cell votes and the Hamming decode of a candidate are the head of its refinement wave (one dispatch less)
    const bool fused_cells = dp.decoder == ARUCOHIP_DECODER_FIDUCIAL_5X5;
    if (RUN_STAGE(b.tune, 6)) launch_decode(s, gray_dev, g, nframes, dp, b, fused_cells);
    if (dp.decoder == ARUCOHIP_DECODER_USER) {
        const int rc_ = user_decode_stage(h, dp);
        if (rc_) return rc_;
    }
    MARK(K_REFINE_LINES);
    if (RUN_STAGE(b.tune, 7)) launch_refine_lines(s, g, nframes, dp, cam, b, fused_cells);
    MARK(K_REFINE_PIXELS);
    if (dp.corner_method == ARUCOHIP_CORNER_HARRIS || dp.corner_method == ARUCOHIP_CORNER_SUBPIX) {
        if (dp.locked) launch_locked_corners(s, gray_dev, g, nframes, dp, b);   // markerdetector.cpp:398-399
        launch_refine_pixels(s, gray_dev, g, nframes, dp, b);
    }
    MARK(K_FINALIZE);
    if (RUN_STAGE(b.tune, 8)) launch_finalize(s, g, nframes, dp, cam, b, h->wt_out, h->wt_cap, h->wt_n);
    MARK(K_POSE);
    if (RUN_STAGE(b.tune, 8) && cam.has_K && cam.marker_size > 0) launch_pose(s, nframes, cam, b);
    MARK(K_COUNT);
#undef MARK
    if (tm) h->tsets++;
    HIPCHK(h, hipGetLastError());
    h->last_w = g.width, h->last_h = g.height, h->last_frames = nframes, h->last_nthr = dp.nthr;
    h->last_gray = gray_dev, h->last_geom = g;
    return ARUCOHIP_OK;
}

static int grow(arucohip_handle* h, uint8_t** buf, size_t* have, size_t need) {
    if (need <= *have) return ARUCOHIP_OK;
    if (*buf) HIPCHK(h, hipFree(*buf));
    *buf = nullptr, *have = 0;
    HIPCHK(h, hipMalloc((void**)buf, need));
    *have = need;
    return ARUCOHIP_OK;
}

// channels = 1: gray frames (device frames are used in place); channels = 3: B,G,R interleaved, converted into d_gray
static int stage_frames(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                        int on_device, int channels, const uint8_t** gray_dev, FrameGeom* g) {
    g->width = W, g->height = H;
    int rc;
    if (channels == 3) {
        const uint8_t* bgr = frames;
        size_t rs = row_stride, fs = frame_stride;
        if (!on_device) {
            if ((rc = grow(h, &h->d_bgr, &h->bgr_bytes, (size_t)nframes * W * H * 3))) return rc;
            for (int f = 0; f < nframes; f++)
                HIPCHK(h, hipMemcpy2DAsync(h->d_bgr + (size_t)f * W * H * 3, (size_t)W * 3, frames + (size_t)f * frame_stride, row_stride, (size_t)W * 3, H,
                                           hipMemcpyHostToDevice, h->stream));
            bgr = h->d_bgr, rs = (size_t)W * 3, fs = (size_t)W * H * 3;
        }
        if ((rc = grow(h, &h->d_gray, &h->gray_bytes, (size_t)nframes * W * H))) return rc;
        launch_bgr2gray(h->stream, bgr, rs, fs, W, H, nframes, h->d_gray);
        HIPCHK(h, hipGetLastError());
        *gray_dev = h->d_gray;
        g->row_stride = W, g->frame_stride = (size_t)W * H;
        return ARUCOHIP_OK;
    }
    if (on_device) {
        *gray_dev = frames;
        g->row_stride = row_stride, g->frame_stride = frame_stride;
        return ARUCOHIP_OK;
    }
    if ((rc = grow(h, &h->d_gray, &h->gray_bytes, (size_t)nframes * W * H))) return rc;
    if (row_stride == (size_t)W && frame_stride == (size_t)W * H) {
        // tightly packed frames (a pinned ring of camera frames): ONE copy for the batch instead of one 2-D copy per frame
        HIPCHK(h, hipMemcpyAsync(h->d_gray, frames, (size_t)nframes * W * H, hipMemcpyHostToDevice, h->stream));
    } else {
        for (int f = 0; f < nframes; f++)
            HIPCHK(h, hipMemcpy2DAsync(h->d_gray + (size_t)f * W * H, W, frames + (size_t)f * frame_stride, row_stride, W, H,
                                       hipMemcpyHostToDevice, h->stream));
    }
    *gray_dev = h->d_gray;
    g->row_stride = W, g->frame_stride = (size_t)W * H;
    return ARUCOHIP_OK;
}

static int check_geometry(arucohip_handle* h, int nframes, int W, int H, size_t row_stride, int channels = 1) {
    if (nframes < 1 || nframes > h->lim.max_batch) return fail(h, ARUCOHIP_E_INVALID, "nframes outside 1..max_batch");
    // every device array and packed field is sized per dimension (tile rows, 14-bit checkpoint coordinates, raster keys)
    if (W < 1 || H < 1 || W > h->lim.max_width || H > h->lim.max_height) return fail(h, ARUCOHIP_E_INVALID, "frame wider or taller than the handle was created for");
    if (row_stride < (size_t)W * channels) return fail(h, ARUCOHIP_E_INVALID, "row_stride < width * channels");
    return ARUCOHIP_OK;
}

extern "C" {

// enqueue one chunk on worker w (its buffers, its stream); results go to device memory or to w's pinned staging
static int chunk_enqueue(arucohip_handle* w, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                         int frames_on_device, int channels, const DetectParams& dp, const CamModel& cam, arucohip_marker_t* out, int cap, int32_t* n_out,
                         int out_on_device) {
    int rc;
    const uint8_t* gray_dev;
    FrameGeom g;
    if ((rc = stage_frames(w, frames, nframes, W, H, row_stride, frame_stride, frames_on_device, channels, &gray_dev, &g))) return rc;
    // results for device memory without poses: finalize_kernel stores them there itself
    const bool write_through = out_on_device && !(cam.has_K && cam.marker_size > 0) && cap > 0;
    w->wt_out = write_through ? out : nullptr, w->wt_cap = write_through ? cap : 0, w->wt_n = write_through ? n_out : nullptr;
    rc = detect_core(w, gray_dev, g, nframes, dp, cam);
    w->wt_out = nullptr, w->wt_cap = 0, w->wt_n = nullptr;
    if (rc) return rc;
    if (write_through) return ARUCOHIP_OK;
    const Buffers& b = w->buf;
    const int ncopy = std::min(cap, b.cap_markers);
    if (out_on_device) {
        if (ncopy > 0)
            HIPCHK(w, hipMemcpy2DAsync(out, (size_t)cap * sizeof(arucohip_marker_t), b.markers, (size_t)b.cap_markers * sizeof(arucohip_marker_t),
                                       (size_t)ncopy * sizeof(arucohip_marker_t), nframes, hipMemcpyDeviceToDevice, w->stream));
        HIPCHK(w, hipMemcpyAsync(n_out, b.nmarkers, nframes * sizeof(int32_t), hipMemcpyDeviceToDevice, w->stream));
        return ARUCOHIP_OK;
    }
    HIPCHK(w, hipMemcpyAsync(w->h_markers, b.markers, (size_t)nframes * b.cap_markers * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipMemcpyAsync(w->h_n, b.nmarkers, nframes * sizeof(int32_t), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipMemcpyAsync(w->h_counters, b.counters, CNT_FIXED * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
    return ARUCOHIP_OK;
}

// after the worker's stream has drained: copy the chunk's markers from the pinned staging to the caller's arrays
static int chunk_collect_host(arucohip_handle* h, arucohip_handle* w, int nframes, arucohip_marker_t* out, int cap, int32_t* n_out) {
    int ret = check_status(h, w->h_counters[CNT_STATUS] & ~(uint32_t)ST_MARKER_OVERFLOW);
    const Buffers& b = w->buf;
    for (int f = 0; f < nframes; f++) {
        int n = w->h_n[f];
        n_out[f] = n;
        if (n > cap) {
            if (ret == ARUCOHIP_OK) ret = fail(h, ARUCOHIP_E_CAPACITY, "marker output array too small");
            n = cap;
        }
        n = std::min(n, b.cap_markers);
        if (n > 0) std::memcpy(out + (size_t)f * cap, w->h_markers + (size_t)f * b.cap_markers, (size_t)n * sizeof(arucohip_marker_t));
    }
    return ret;
}

// fork: the workers' streams wait for what the caller's stream has queued so far
static int fork_workers(arucohip_handle* h, int chunks) {
    if (chunks <= 1) return ARUCOHIP_OK;
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    for (int c = 1; c < chunks; c++) HIPCHK(h, hipStreamWaitEvent(h->kids[c - 1]->stream, h->ev_fork, 0));
    return ARUCOHIP_OK;
}
// join: the caller's stream waits for the workers
static int join_workers(arucohip_handle* h, int chunks) {
    for (int c = 1; c < chunks; c++) {
        HIPCHK(h, hipEventRecord(h->ev_join[c - 1], h->kids[c - 1]->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[c - 1], 0));
    }
    return ARUCOHIP_OK;
}

// the batch's stream has been filled: wait for it and copy every chunk's markers from the pinned staging to the caller
static int collect_batch_host(arucohip_handle* h, int nframes, arucohip_marker_t* out, int cap, int32_t* n_out) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int chunks = std::max(h->last_chunks, 1), per = chunks > 1 ? h->last_per : nframes;
    int ret = ARUCOHIP_OK;
    for (int c = 0; c < chunks; c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        const int off = c * per, cnt = std::min(per, nframes - off);
        if (cnt <= 0) break;
        int r = chunk_collect_host(h, w, cnt, out + (size_t)off * cap, cap, n_out + off);
        if (ret == ARUCOHIP_OK) ret = r;
    }
    return ret;
}

// FNV-1a over the bytes of everything a captured launch carries by value
static uint64_t digest(uint64_t hsh, const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) hsh = (hsh ^ c[i]) * 1099511628211ull;
    return hsh;
}

// arucohip_detect on one host frame through a captured graph. *handled = false: the caller takes the eager path (first call of a
// configuration, timing on, a decoder or threshold method that blocks the host, a failed capture). The frame's copy to the device is issued
// eagerly in front of the graph; inside it: the counters' memset, every kernel of detect_core (with the fork to the side stream of the late
// walker generations) and the three copies of the results into the handle's pinned staging.
static int detect_one_graphed(arucohip_handle* h, const uint8_t* frame, int W, int H, size_t row_stride, int channels, const DetectParams& dp, const CamModel& cam,
                              arucohip_marker_t* out, int cap, int32_t* n_out, bool* handled) {
    *handled = false;
    if (h->fgraph.disabled || h->timing || dp.decoder == ARUCOHIP_DECODER_USER || dp.thres_method == ARUCOHIP_THRES_CANNY || h->params.erode) return ARUCOHIP_OK;
    uint64_t key = 1469598103934665603ull;
    const int geo[4] = {W, H, channels, (int)h->buf.seg_mode};
    key = digest(key, geo, sizeof(geo));
    key = digest(key, &dp, sizeof(dp));
    key = digest(key, &cam, sizeof(cam));
    key = digest(key, &h->stream, sizeof(h->stream));
    if (h->fgraph.exec && h->fgraph.key != key) {   // another configuration: start over
        (void)hipGraphExecDestroy(h->fgraph.exec);
        h->fgraph.exec = nullptr, h->fgraph.seen = 0;
    }
    if (!h->fgraph.exec && h->fgraph.seen != key) {   // first call with this configuration: eager (it sizes every buffer), remember it
        h->fgraph.seen = key;
        return ARUCOHIP_OK;
    }
    int rc;
    const uint8_t* gray_dev;
    FrameGeom g;
    if ((rc = stage_frames(h, frame, 1, W, H, row_stride, (size_t)H * row_stride, 0, channels, &gray_dev, &g))) return rc;   // H2D (+ BGR conversion), eager
    h->last_chunks = 1, h->last_per = 1;
    if (!h->fgraph.exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->fgraph.disabled = 1;
            return ARUCOHIP_OK;   // the frame is staged; the eager path stages it again, which is harmless
        }
        rc = detect_core(h, gray_dev, g, 1, dp, cam);
        const Buffers& b = h->buf;
        hipError_t e = hipSuccess;
        if (rc == ARUCOHIP_OK) {
            // ONE copy: the markers and, in the slot behind them, the count and the status word finalize_kernel left there
            e = hipMemcpyAsync(h->h_markers, b.markers, ((size_t)b.cap_markers + 1) * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost, h->stream);
        }
        const hipError_t e2 = hipStreamEndCapture(h->stream, &graph);
        if (rc != ARUCOHIP_OK || e != hipSuccess || e2 != hipSuccess || !graph || hipGraphInstantiate(&h->fgraph.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            h->fgraph.exec = nullptr, h->fgraph.disabled = 1;   // this handle stays on the eager path
            return ARUCOHIP_OK;
        }
        (void)hipGraphDestroy(graph);
        h->fgraph.key = key;
    }
    *handled = true;
    HIPCHK(h, hipGraphLaunch(h->fgraph.exec, h->stream));
    h->last_w = W, h->last_h = H, h->last_frames = 1, h->last_nthr = dp.nthr, h->last_gray = gray_dev, h->last_geom = g;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int32_t* hdr = (const int32_t*)(h->h_markers + h->buf.cap_markers);
    h->h_n[0] = hdr[0], h->h_counters[CNT_STATUS] = (uint32_t)hdr[1];
    return collect_batch_host(h, 1, out, cap, n_out);
}

static int detect_batch_impl(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                             int frames_on_device, int channels, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                             arucohip_marker_t* out, int cap, int32_t* n_out, int out_on_device, bool defer = false) {
    if (!h || !frames || !n_out || (cap > 0 && !out) || cap < 0) return ARUCOHIP_E_INVALID;
    int rc = check_geometry(h, nframes, W, H, row_stride, channels);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    DetectParams dp;
    CamModel cam;
    if ((rc = make_detect_params(h, W, H, &dp))) return rc;
    if ((rc = make_cam(h, K, dist, ndist, marker_size, y_perp, &cam))) return rc;
    if (nframes == 1 && !frames_on_device && !out_on_device && !defer && h->nsub == 1) {
        bool handled = false;
        rc = detect_one_graphed(h, frames, W, H, row_stride, channels, dp, cam, out, cap, n_out, &handled);
        if (handled) return rc;
    }
    // chunks of equal size, as few as the workers' buffers allow but one per worker when the batch is large enough to share
    int chunks = (nframes + h->cap_frames - 1) / h->cap_frames;
    if (h->nsub > 1 && (size_t)nframes * W * H >= (size_t)h->nsub * 32 * 1024 * 1024) chunks = std::max(chunks, std::min(h->nsub, nframes));
    const int per = (nframes + chunks - 1) / chunks;
    chunks = (nframes + per - 1) / per;
    h->last_chunks = chunks, h->last_per = per;
    if ((rc = fork_workers(h, chunks))) return rc;
    for (int c = 0; c < chunks; c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        const int off = c * per, cnt = std::min(per, nframes - off);
        // optional stagger (ARUCOHIP_CHAIN=1): the bandwidth-bound threshold kernels of the chunks run one after the other,
        // so that chunk c's threshold overlaps the latency-bound border following / decoding of chunk c-1. Helps with 4
        // streams on some boxes and hurts on others, hence off by default.
        const bool chain = h->buf.tune.chain != 0;
        w->wait_thr = (chain && c > 0) ? (c == 1 ? h : h->kids[c - 2])->ev_thr : nullptr;
        rc = chunk_enqueue(w, frames + (size_t)off * frame_stride, cnt, W, H, row_stride, frame_stride, frames_on_device, channels, dp, cam,
                           out ? out + (size_t)off * cap : nullptr, cap, n_out + off, out_on_device);
        if (rc) {
            if (w != h) h->err = w->err;
            // the workers that already have queued work still have to rejoin the caller's stream
            const std::string keep = h->err;
            (void)join_workers(h, chunks);
            for (int k = 0; k < chunks; k++) (k == 0 ? h : h->kids[k - 1])->wait_thr = nullptr;
            h->err = keep;
            return rc;
        }
    }
    if ((rc = join_workers(h, chunks))) return rc;
    if (out_on_device || defer) return ARUCOHIP_OK;
    return collect_batch_host(h, nframes, out, cap, n_out);
}

int arucohip_detect_batch(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                          int frames_on_device, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                          arucohip_marker_t* out, int cap, int32_t* n_out, int out_on_device) {
    return detect_batch_impl(h, frames, nframes, W, H, row_stride, frame_stride, frames_on_device, 1, K, dist, ndist, marker_size, y_perp, out, cap,
                             n_out, out_on_device);
}

int arucohip_detect_batch_bgr(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                              int frames_on_device, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                              arucohip_marker_t* out, int cap, int32_t* n_out, int out_on_device) {
    return detect_batch_impl(h, frames, nframes, W, H, row_stride, frame_stride, frames_on_device, 3, K, dist, ndist, marker_size, y_perp, out, cap,
                             n_out, out_on_device);
}

int arucohip_detect_bgr(arucohip_handle* h, const uint8_t* bgr, int W, int H, size_t row_stride, const float* K, const float* dist, int ndist,
                        float marker_size, int y_perp, arucohip_marker_t* out, int cap, int* n_out) {
    int32_t n = 0;
    int rc = arucohip_detect_batch_bgr(h, bgr, 1, W, H, row_stride, (size_t)H * row_stride, 0, K, dist, ndist, marker_size, y_perp, out, cap, &n, 0);
    if (n_out) *n_out = n;
    return rc;
}

int arucohip_bgr_to_gray(arucohip_handle* h, const uint8_t* bgr, int W, int H, size_t row_stride, uint8_t* gray) {
    if (!h || !bgr || !gray) return ARUCOHIP_E_INVALID;
    int rc = check_geometry(h, 1, W, H, row_stride, 3);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    const uint8_t* dev;
    FrameGeom g;
    if ((rc = stage_frames(h, bgr, 1, W, H, row_stride, (size_t)H * row_stride, 0, 3, &dev, &g))) return rc;
    HIPCHK(h, hipMemcpyAsync(gray, dev, (size_t)W * H, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ARUCOHIP_OK;
}

// cv::undistort(src, dst, K, dist) as the reference's GL apps call it before detect()
// (utils/aruco_test_gl.cpp:237-240, utils/aruco_test_board_gl.cpp:265-268)
int arucohip_undistort(arucohip_handle* h, const uint8_t* src, int nframes, int W, int H, size_t row_stride, size_t frame_stride, int channels,
                       int src_on_device, const float* K, const float* dist, int ndist, uint8_t* dst, int dst_on_device) {
    if (!h || !src || !dst || !K || (channels != 1 && channels != 3) || ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return ARUCOHIP_E_INVALID;
    int rc = check_geometry(h, nframes, W, H, row_stride, channels);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // map of this camera: recomputed only when size, K or dist change
    bool same = h->d_umap_xy && h->umap_w == W && h->umap_h == H && h->umap_nd == ndist && std::memcmp(h->umap_K, K, sizeof(h->umap_K)) == 0 &&
                (ndist == 0 || std::memcmp(h->umap_d, dist, ndist * sizeof(float)) == 0);
    if (!same) {
        const size_t px = (size_t)W * H;
        if (px > h->umap_px) {
            if (h->d_umap_xy) HIPCHK(h, hipFree(h->d_umap_xy));
            if (h->d_umap_f) HIPCHK(h, hipFree(h->d_umap_f));
            h->d_umap_xy = nullptr, h->d_umap_f = nullptr, h->umap_px = 0;
            HIPCHK(h, hipMalloc((void**)&h->d_umap_xy, px * sizeof(short2)));
            HIPCHK(h, hipMalloc((void**)&h->d_umap_f, px * sizeof(uint16_t)));
            h->umap_px = px;
        }
        launch_undist_map(s, W, H, K, dist, ndist, h->d_umap_xy, h->d_umap_f);
        HIPCHK(h, hipGetLastError());
        h->umap_w = W, h->umap_h = H, h->umap_nd = ndist;
        std::memcpy(h->umap_K, K, sizeof(h->umap_K));
        if (ndist) std::memcpy(h->umap_d, dist, ndist * sizeof(float));
    }
    const size_t fbytes = (size_t)W * H * channels;
    const uint8_t* sdev = src;
    size_t rs = row_stride, fs = frame_stride;
    if (!src_on_device) {
        if ((rc = grow(h, &h->d_bgr, &h->bgr_bytes, (size_t)nframes * fbytes))) return rc;
        for (int f = 0; f < nframes; f++)
            HIPCHK(h, hipMemcpy2DAsync(h->d_bgr + (size_t)f * fbytes, (size_t)W * channels, src + (size_t)f * frame_stride, row_stride, (size_t)W * channels, H,
                                       hipMemcpyHostToDevice, s));
        sdev = h->d_bgr, rs = (size_t)W * channels, fs = fbytes;
    }
    uint8_t* ddev = dst;
    if (!dst_on_device) {
        if ((rc = grow(h, &h->d_undist, &h->undist_bytes, (size_t)nframes * fbytes))) return rc;
        ddev = h->d_undist;
    }
    launch_remap(s, sdev, rs, fs, W, H, channels, nframes, h->d_umap_xy, h->d_umap_f, ddev);
    HIPCHK(h, hipGetLastError());
    if (!dst_on_device) {
        HIPCHK(h, hipMemcpyAsync(dst, ddev, (size_t)nframes * fbytes, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return ARUCOHIP_OK;
}

int arucohip_set_dictionary(arucohip_handle* h, int n, int count, const uint64_t* codes, int tau0, float correction_rate) {
    if (!h) return ARUCOHIP_E_INVALID;
    drop_retry(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_hrm) HIPCHK(h, hipFree(h->d_hrm));
    h->d_hrm = nullptr, h->hrm_count = 0, h->hrm_n = 0;
    if (count > 0) {
        if (!codes || n < 2 || n > 8 || count > 4096) return fail(h, ARUCOHIP_E_UNSUPPORTED, "dictionary: 2 <= n <= 8, count <= 4096");
        HIPCHK(h, hipMalloc((void**)&h->d_hrm, (size_t)count * sizeof(uint64_t)));
        HIPCHK(h, hipMemcpy(h->d_hrm, codes, (size_t)count * sizeof(uint64_t), hipMemcpyHostToDevice));
        h->hrm_n = n, h->hrm_count = count, h->hrm_tau0 = tau0, h->hrm_rate = correction_rate;
    }
    for (auto* k : h->kids) {
        int rc = arucohip_set_dictionary(k, n, count, codes, tau0, correction_rate);
        if (rc) return rc;
    }
    for (auto* l : h->lanes) {
        int rc = arucohip_set_dictionary(l, n, count, codes, tau0, correction_rate);
        if (rc) return rc;
    }
    return ARUCOHIP_OK;
}

int arucohip_set_decoder_callback(arucohip_handle* h, arucohip_decoder_fn fn, void* user) {
    if (!h) return ARUCOHIP_E_INVALID;
    drop_retry(h);
    h->decoder_fn = fn, h->decoder_user = user;
    for (auto* k : h->kids) k->decoder_fn = fn, k->decoder_user = user;
    for (auto* l : h->lanes) arucohip_set_decoder_callback(l, fn, user);
    if (!fn && h->params.decoder_kind == ARUCOHIP_DECODER_USER) {
        h->params.decoder_kind = ARUCOHIP_DECODER_FIDUCIAL_5X5;
        for (auto* k : h->kids) k->params.decoder_kind = ARUCOHIP_DECODER_FIDUCIAL_5X5;
    }
    return ARUCOHIP_OK;
}

int arucohip_batch_status(arucohip_handle* h) {
    if (!h) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    uint32_t st = 0;
    for (int c = 0; c < std::max(h->last_chunks, 1); c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        HIPCHK(h, hipMemcpyAsync(w->h_counters, w->buf.counters, CNT_FIXED * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(h, hipStreamSynchronize(w->stream));
        st |= w->h_counters[CNT_STATUS];
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_status(h, st);
}

int arucohip_batch_chunks(arucohip_handle* h, int* frames_per_chunk) {
    if (!h) return 0;
    h = active(h);
    if (frames_per_chunk) *frames_per_chunk = h->last_chunks > 1 ? h->last_per : h->last_frames;
    return std::max(h->last_chunks, 1);
}

int arucohip_detect(arucohip_handle* h, const uint8_t* gray, int W, int H, size_t row_stride, const float* K, const float* dist, int ndist,
                    float marker_size, int y_perp, arucohip_marker_t* out, int cap, int* n_out) {
    int32_t n = 0;
    int rc = arucohip_detect_batch(h, gray, 1, W, H, row_stride, (size_t)H * row_stride, 0, K, dist, ndist, marker_size, y_perp, out, cap, &n, 0);
    if (n_out) *n_out = n;
    return rc;
}

int arucohip_get_thresholded(arucohip_handle* h0, int frame, uint8_t* dst) {
    if (!h0 || !dst || frame < 0) return ARUCOHIP_E_INVALID;
    h0 = active(h0);
    arucohip_handle* h = route(h0, frame, &frame);
    if (frame >= h->last_frames) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    size_t px = (size_t)h->last_w * h->last_h;
    int plane = frame * h->last_nthr + h->last_nthr / 2;   // thres = thres_images[n_param1 / 2]
    if (!h->thres_bytes) {   // the batch kept the image as tiles + border lines: rebuild this plane's bytes
        FrameGeom g;
        g.width = h->last_w, g.height = h->last_h, g.row_stride = (size_t)h->last_w, g.frame_stride = px;
        launch_expand_thres(h->stream, g, plane, h->buf);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(dst, h->buf.thres + plane * px, px, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ARUCOHIP_OK;
}

static int fetch_cands(arucohip_handle* h0, int frame, std::vector<Cand>* v) {
    if (!h0 || frame < 0) return ARUCOHIP_E_INVALID;
    h0 = active(h0);
    arucohip_handle* h = route(h0, frame, &frame);
    if (frame >= h->last_frames) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    int32_t n = 0;
    HIPCHK(h, hipMemcpyAsync(&n, h->buf.ncands + frame, sizeof(n), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    v->resize(std::max(n, 0));
    if (n > 0) {
        HIPCHK(h, hipMemcpyAsync(v->data(), h->buf.cands + (size_t)frame * h->buf.cap_cands, n * sizeof(Cand), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return ARUCOHIP_OK;
}

int arucohip_get_candidates(arucohip_handle* h, int frame, float* quads, int cap, int* n) {
    std::vector<Cand> v;
    int rc = fetch_cands(h, frame, &v);
    if (rc) return rc;
    int k = 0;
    for (auto& c : v) {
        if (c.id != -1) continue;
        if (k < cap)
            for (int i = 0; i < 8; i++) quads[k * 8 + i] = c.c[i];
        k++;
    }
    if (n) *n = k;
    return k > cap ? ARUCOHIP_E_CAPACITY : ARUCOHIP_OK;
}

// Otsu threshold of every candidate of a frame (candidate order of arucohip_debug_candidates): what otsu_kernel left in the flat list's threshold slots
int arucohip_debug_otsu(arucohip_handle* h0, int frame, int32_t* thr, int cap, int* n) {
    if (!h0 || frame < 0 || !thr || !n) return ARUCOHIP_E_INVALID;
    h0 = active(h0);
    arucohip_handle* h = route(h0, frame, &frame);
    if (frame >= h->last_frames) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    uint32_t cnt[CNT_FIXED];
    int32_t nc = 0;
    HIPCHK(h, hipMemcpyAsync(cnt, h->buf.counters, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&nc, h->buf.ncands + frame, sizeof(nc), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t nflat = std::min(cnt[CNT_NCAND], h->buf.cap_flat);
    std::vector<uint32_t> list(nflat);
    std::vector<int32_t> othr(nflat);
    if (nflat) {
        HIPCHK(h, hipMemcpyAsync(list.data(), h->buf.cand_list, nflat * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(othr.data(), h->buf.othr, nflat * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    *n = std::max(nc, 0);
    for (int i = 0; i < std::min(*n, cap); i++) thr[i] = -1;
    for (uint32_t i = 0; i < nflat; i++)
        if ((int)(list[i] >> 16) == frame && (int)(list[i] & 0xFFFFu) < cap) thr[list[i] & 0xFFFFu] = othr[i];
    return *n > cap ? ARUCOHIP_E_CAPACITY : ARUCOHIP_OK;
}

int arucohip_debug_candidates(arucohip_handle* h, int frame, float* quads0, int32_t* ids, int32_t* nrot, int cap, int* n) {
    std::vector<Cand> v;
    int rc = fetch_cands(h, frame, &v);
    if (rc) return rc;
    int k = 0;
    for (auto& c : v) {
        if (k < cap) {
            for (int i = 0; i < 4; i++) quads0[k * 8 + 2 * i] = c.qx[i], quads0[k * 8 + 2 * i + 1] = c.qy[i];
            if (ids) ids[k] = c.id;
            if (nrot) nrot[k] = c.nrot;
        }
        k++;
    }
    if (n) *n = k;
    return k > cap ? ARUCOHIP_E_CAPACITY : ARUCOHIP_OK;
}

// contours of one frame in reference (RETR_LIST) order: planes ascending, raster key descending
static int fetch_contours(arucohip_handle* h0, int frame, std::vector<ContourDesc>* out, arucohip_handle** owner = nullptr) {
    if (!h0 || frame < 0) return ARUCOHIP_E_INVALID;
    h0 = active(h0);
    arucohip_handle* h = route(h0, frame, &frame);
    if (owner) *owner = h;
    if (frame >= h->last_frames) return ARUCOHIP_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    // the frame's planes are consecutive; every plane owns cap_cdesc descriptor slots
    std::vector<ContourDesc> all;
    for (int t = 0; t < h->last_nthr; t++) {
        const int plane = frame * h->last_nthr + t;
        uint32_t n = 0;
        HIPCHK(h, hipMemcpyAsync(&n, h->buf.trig_cnt + (size_t)plane * TRIG_CNT_STRIDE + TC_CDESC, sizeof(n), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        n = std::min(n, h->buf.cap_cdesc);
        if (!n) continue;
        const size_t at = all.size();
        all.resize(at + n);
        HIPCHK(h, hipMemcpyAsync(all.data() + at, h->buf.cdesc + (size_t)plane * h->buf.cap_cdesc, n * sizeof(ContourDesc), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    out->clear();
    for (auto& c : all)
        if (c.n > 0) out->push_back(c);
    std::sort(out->begin(), out->end(), [](const ContourDesc& a, const ContourDesc& b) {
        if (a.plane != b.plane) return a.plane < b.plane;
        return a.key > b.key;
    });
    return ARUCOHIP_OK;
}

int arucohip_debug_num_contours(arucohip_handle* h, int frame, int* n) {
    std::vector<ContourDesc> v;
    int rc = fetch_contours(h, frame, &v);
    if (rc) return rc;
    *n = (int)v.size();
    return ARUCOHIP_OK;
}

int arucohip_debug_contour(arucohip_handle* h0, int frame, int index, int* is_hole, int* sx, int* sy, int16_t* xy, int cap_points, int* n_points) {
    std::vector<ContourDesc> v;
    arucohip_handle* h = h0;
    int rc = fetch_contours(h0, frame, &v, &h);
    if (rc) return rc;
    if (index < 0 || index >= (int)v.size()) return ARUCOHIP_E_INVALID;
    const ContourDesc& c = v[index];
    if (is_hole) *is_hole = c.hole;
    if (sx) *sx = c.x0;
    if (sy) *sy = c.y0;
    if (n_points) *n_points = c.n;
    if (xy && cap_points >= c.n) {
        HIPCHK(h, hipMemcpyAsync(xy, h->buf.pool + c.pool_off, (size_t)c.n * sizeof(short2), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else if (xy) {
        return ARUCOHIP_E_CAPACITY;
    }
    return ARUCOHIP_OK;
}

int arucohip_debug_counters(arucohip_handle* h, uint32_t* out8) {
    if (!h || !out8) return ARUCOHIP_E_INVALID;
    h = active(h);
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t acc[CNT_FIXED] = {};
    uint64_t ntrig = 0, nraw = 0, nlong = 0;
    for (int c = 0; c < std::max(h->last_chunks, 1); c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        uint32_t cnt[CNT_FIXED];
        HIPCHK(h, hipMemcpyAsync(cnt, w->buf.counters, sizeof(cnt), hipMemcpyDeviceToHost, w->stream));
        int planes = std::max(w->last_frames * w->last_nthr, 1);
        std::vector<uint32_t> tc((size_t)planes * TRIG_CNT_STRIDE), rc_((size_t)planes * TRIG_CNT_STRIDE), rg((size_t)planes * TRIG_CNT_STRIDE);
        HIPCHK(h, hipMemcpyAsync(tc.data(), w->buf.trig_cnt, tc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(h, hipMemcpyAsync(rc_.data(), w->buf.raw_cnt, rc_.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(h, hipMemcpyAsync(rg.data(), w->buf.ring_cnt, rg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(h, hipStreamSynchronize(w->stream));
        for (int i = 0; i < CNT_FIXED; i++)
            if (i != 1 && i != 2) acc[i] = (i == CNT_STATUS) ? (acc[i] | cnt[i]) : acc[i] + cnt[i];   // [1], [2] come from the planes' own counters below
        for (int p = 0; p < planes; p++) {
            ntrig += tc[(size_t)p * TRIG_CNT_STRIDE] + tc[(size_t)p * TRIG_CNT_STRIDE + 1];
            acc[1] += tc[(size_t)p * TRIG_CNT_STRIDE + TC_CDESC], acc[2] += tc[(size_t)p * TRIG_CNT_STRIDE + TC_POOL];
            nraw += rc_[(size_t)p * TRIG_CNT_STRIDE];
            nlong += rg[(size_t)p * TRIG_CNT_STRIDE] + rg[(size_t)p * TRIG_CNT_STRIDE + 1];
        }
    }
    for (int i = 0; i < CNT_FIXED; i++) out8[i] = (uint32_t)std::min<uint64_t>(acc[i], 0xFFFFFFFFu);
    out8[0] = (uint32_t)std::min<uint64_t>(ntrig, 0xFFFFFFFFu);   // start candidates after the run rule (all planes)
    out8[4] = (uint32_t)std::min<uint64_t>(h->buf.seg_mode ? nraw : nlong, 0xFFFFFFFFu);   // waypoint records (segment mode) / long walks = checkpoint rings handed out
    return ARUCOHIP_OK;
}

// ---- stage entry points (markerdetector.h:255-280)
int arucohip_threshold(arucohip_handle* h, int method, const uint8_t* gray, int W, int H, size_t row_stride, double param1, double param2, uint8_t* dst) {
    if (!h || !gray || !dst) return ARUCOHIP_E_INVALID;
    int rc = check_geometry(h, 1, W, H, row_stride);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    arucohip_params_t saved = h->params;
    arucohip_params_t p = saved;
    p.thres_method = method;
    if (param1 != -1) p.thres_param1 = param1;   // thresHold(): -1 selects the configured value (:646-649)
    if (param2 != -1) p.thres_param2 = param2;
    p.thres_param1_range = 0;
    if ((rc = validate_params(h, &p))) return rc;
    h->params = p;
    DetectParams dp;
    rc = make_detect_params(h, W, H, &dp);
    h->params = saved;
    if (rc) return rc;
    const uint8_t* gray_dev;
    FrameGeom g;
    if ((rc = stage_frames(h, gray, 1, W, H, row_stride, (size_t)H * row_stride, 0, 1, &gray_dev, &g))) return rc;
    HIPCHK(h, hipMemsetAsync(h->zero_block, 0, h->zero_words * sizeof(uint32_t), h->stream));
    if ((rc = ensure_bits_geometry(h, W, H))) return rc;
    if ((rc = run_threshold(h, h->stream, gray_dev, g, 1, dp, true))) return rc;
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(dst, h->buf.thres, (size_t)W * H, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->last_w = W, h->last_h = H, h->last_frames = 1, h->last_nthr = 1, h->last_chunks = 1;
    return ARUCOHIP_OK;
}

int arucohip_detect_rectangles(arucohip_handle* h, const uint8_t* thres, int W, int H, size_t row_stride, float* quads, int cap, int* n) {
    if (!h || !thres || !n) return ARUCOHIP_E_INVALID;
    int rc = check_geometry(h, 1, W, H, row_stride);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    DetectParams dp;
    arucohip_params_t saved = h->params;
    h->params.thres_param1_range = 0;
    rc = make_detect_params(h, W, H, &dp);
    h->params = saved;
    if (rc) return rc;
    const uint8_t* dev;
    FrameGeom g;
    if ((rc = stage_frames(h, thres, 1, W, H, row_stride, (size_t)H * row_stride, 0, 1, &dev, &g))) return rc;
    HIPCHK(h, hipMemsetAsync(h->zero_block, 0, h->zero_words * sizeof(uint32_t), h->stream));
    if ((rc = ensure_walk_scratch(h, 1, dp))) return rc;
    if ((rc = ensure_bits_geometry(h, W, H))) return rc;
    launch_binary_planes(h->stream, dev, g, 1, h->buf);
    run_rectangles(h, g, 1, dp);
    HIPCHK(h, hipGetLastError());
    h->last_w = W, h->last_h = H, h->last_frames = 1, h->last_nthr = 1, h->last_chunks = 1;
    std::vector<Cand> v;
    if ((rc = fetch_cands(h, 0, &v))) return rc;
    HIPCHK(h, hipMemcpy(h->h_counters, h->buf.counters, CNT_FIXED * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if ((rc = check_status(h, h->h_counters[CNT_STATUS]))) return rc;
    *n = (int)v.size();
    for (int k = 0; k < (int)v.size() && k < cap; k++)
        for (int i = 0; i < 8; i++) quads[k * 8 + i] = v[k].c[i];
    return (int)v.size() > cap ? ARUCOHIP_E_CAPACITY : ARUCOHIP_OK;
}

int arucohip_warp(arucohip_handle* h, const uint8_t* gray, int W, int H, size_t row_stride, const float quad[8], int size, uint8_t* dst) {
    if (!h || !gray || !quad || !dst) return ARUCOHIP_E_INVALID;
    if (size < 1 || size > 128) return fail(h, ARUCOHIP_E_INVALID, "warp size outside 1..128");
    int rc = check_geometry(h, 1, W, H, row_stride);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    const uint8_t* dev;
    FrameGeom g;
    if ((rc = stage_frames(h, gray, 1, W, H, row_stride, (size_t)H * row_stride, 0, 1, &dev, &g))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_small_f, quad, 8 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    launch_warp_only(h->stream, dev, g, h->d_small_f, size, h->d_patch);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(dst, h->d_patch, (size_t)size * size, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ARUCOHIP_OK;
}

// MarkerDetector::refineCandidateLines (markerdetector.cpp:931-997) on a caller-supplied contour: the contour becomes border 0 of plane 0, the
// corners candidate 0 of frame 0, and one wave of refine_lines_kernel does what it does for a decoded candidate of a batch.
int arucohip_refine_candidate_lines(arucohip_handle* h, const int32_t* contour_xy, int npoints, float corners[8], const float* K, const float* dist, int ndist) {
    if (!h || !contour_xy || !corners || npoints < 1) return ARUCOHIP_E_INVALID;
    if ((uint32_t)npoints > h->buf.cap_pool) return fail(h, ARUCOHIP_E_CAPACITY, "contour longer than the handle's point list (points_per_frame)");
    HIPCHK(h, hipSetDevice(h->device));
    CamModel cam;
    int rc = make_cam(h, K, dist, ndist, -1.f, 0, &cam);
    if (rc) return rc;
    std::vector<short2> pts((size_t)npoints);
    for (int i = 0; i < npoints; i++) {
        const int32_t x = contour_xy[2 * i], y = contour_xy[2 * i + 1];
        if (x < 0 || y < 0 || x > 32767 || y > 32767) return fail(h, ARUCOHIP_E_INVALID, "contour point outside 0..32767");
        pts[i] = make_short2((short)x, (short)y);
    }
    ContourDesc cd{};
    cd.plane = 0, cd.x0 = pts[0].x, cd.y0 = pts[0].y, cd.hole = 0, cd.n = npoints, cd.key = 0, cd.pool_off = 0, cd.ck_off = 0xFFFFFFFFu;
    Cand c{};
    for (int k = 0; k < 4; k++) {
        c.c[2 * k] = corners[2 * k], c.c[2 * k + 1] = corners[2 * k + 1];
        // Point(candidate[k]): cv::Point2f -> cv::Point rounds to nearest, ties to even (saturate_cast<int>(float) = cvRound)
        const long qx = lrintf(corners[2 * k]), qy = lrintf(corners[2 * k + 1]);
        c.qx[k] = (int16_t)std::min<long>(std::max<long>(qx, -32768), 32767), c.qy[k] = (int16_t)std::min<long>(std::max<long>(qy, -32768), 32767);
    }
    c.cdesc = 0, c.swapped = 0, c.id = 0, c.nrot = 0;
    DetectParams dp;
    std::memset(&dp, 0, sizeof(dp));
    dp.nthr = 1, dp.corner_method = ARUCOHIP_CORNER_LINES, dp.warp_size = h->params.warp_size, dp.decoder = ARUCOHIP_DECODER_USER;   // ids are given: no cell decode
    hipStream_t s = h->stream;
    const Buffers& b = h->buf;
    const uint32_t one = 1, entry = 0;
    HIPCHK(h, hipMemsetAsync(h->zero_block, 0, h->zero_words * sizeof(uint32_t), s));
    HIPCHK(h, hipMemcpyAsync(b.pool, pts.data(), pts.size() * sizeof(short2), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(b.cdesc, &cd, sizeof(cd), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(b.cands, &c, sizeof(c), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(b.cand_list, &entry, sizeof(entry), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(b.counters + CNT_NCAND, &one, sizeof(one), hipMemcpyHostToDevice, s));
    FrameGeom g{};
    g.width = h->lim.max_width, g.height = h->lim.max_height;
    launch_refine_lines(s, g, 1, dp, cam, b, false);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(&c, b.cands, sizeof(c), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int k = 0; k < 8; k++) corners[k] = c.c[k];
    h->last_frames = 0;   // the lists no longer hold a batch
    return ARUCOHIP_OK;
}

int arucohip_board_detect_batch(arucohip_handle* h, int nframes, const int32_t* ids, const float* obj, int nboard, int info_type, const float* K,
                                const float* dist, int ndist, float marker_size, float repj_err_thres, int y_perp, arucohip_board_t* out, float* prob) {
    if (!h || !out || !prob) return ARUCOHIP_E_INVALID;
    if (nboard <= 0 || !ids || !obj) return fail(h, ARUCOHIP_E_BOARD_CONFIG, "invalid BoardConfig that is empty");
    h = active(h);   // with batches in flight: the lane of the last ticket waited for
    const int chunks = std::max(h->last_chunks, 1), per = chunks > 1 ? h->last_per : h->last_frames;
    int have = 0;
    for (int c = 0; c < chunks; c++) have += (c == 0 ? h : h->kids[c - 1])->last_frames;
    if (nframes < 1 || nframes > have) return fail(h, ARUCOHIP_E_INVALID, "nframes exceeds the last batch");
    if (nboard * 12 > 8192) return fail(h, ARUCOHIP_E_CAPACITY, "board with too many markers");
    HIPCHK(h, hipSetDevice(h->device));
    float zeros[4] = {0, 0, 0, 0};
    if (!dist || ndist == 0) dist = zeros, ndist = 4;
    CamModel cam;
    int rc = make_cam(h, K, dist, ndist, marker_size, y_perp, &cam);
    if (rc) return rc;
    // every worker solves the boards of the frames it detected, on its own stream
    if ((rc = fork_workers(h, chunks))) return rc;
    for (int c = 0; c < chunks; c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        const int off = c * per, cnt = std::min(per, nframes - off);
        if (cnt <= 0) break;
        if (!w->d_board)
            HIPCHK(h, hipMalloc((void**)&w->d_board, (size_t)w->cap_frames * (sizeof(arucohip_board_t) + sizeof(float)) + 8192 * sizeof(int32_t)));
        arucohip_board_t* d_out = (arucohip_board_t*)w->d_board;
        float* d_prob = (float*)(d_out + w->cap_frames);
        int32_t* d_ids = (int32_t*)(d_prob + w->cap_frames);
        HIPCHK(h, hipMemcpyAsync(d_ids, ids, (size_t)nboard * sizeof(int32_t), hipMemcpyHostToDevice, w->stream));
        HIPCHK(h, hipMemcpyAsync(w->d_small_f, obj, (size_t)nboard * 12 * sizeof(float), hipMemcpyHostToDevice, w->stream));
        launch_board_pose(w->stream, cnt, w->buf, d_ids, w->d_small_f, nboard, info_type, marker_size, repj_err_thres, cam, d_out, d_prob);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out + off, d_out, (size_t)cnt * sizeof(arucohip_board_t), hipMemcpyDeviceToHost, w->stream));
        HIPCHK(h, hipMemcpyAsync(prob + off, d_prob, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost, w->stream));
    }
    if ((rc = join_workers(h, chunks))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // a frame with more member markers than the kernel's correspondence array holds is reported, not truncated silently
    for (int c = 0; c < chunks; c++) {
        arucohip_handle* w = c == 0 ? h : h->kids[c - 1];
        HIPCHK(h, hipMemcpy(w->h_counters, w->buf.counters, CNT_FIXED * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (w->h_counters[CNT_STATUS] & ST_MARKER_OVERFLOW) return fail(h, ARUCOHIP_E_CAPACITY, "a frame has more than 128 board markers");
    }
    return ARUCOHIP_OK;
}

// SURVEY §8 row f4, batched: Marker::glGetModelViewMatrix (src/marker.h:90) for every marker of the last batch in one launch
int arucohip_gl_modelview_batch(arucohip_handle* h, int nframes, int cap, double* modelview, int32_t* n_out) {
    if (!h || !modelview || !n_out || cap < 1) return ARUCOHIP_E_INVALID;
    h = active(h);
    if (h->last_chunks > 1) return fail(h, ARUCOHIP_E_UNSUPPORTED, "not available for batches split over chunk streams");
    if (nframes < 1 || nframes > h->last_frames) return fail(h, ARUCOHIP_E_INVALID, "nframes exceeds the last batch");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t need = (size_t)nframes * cap * 16 * sizeof(double);
    if (need > h->gl_bytes) {
        if (h->d_gl) HIPCHK(h, hipFree(h->d_gl));
        h->d_gl = nullptr, h->gl_bytes = 0;
        HIPCHK(h, hipMalloc((void**)&h->d_gl, need));
        h->gl_bytes = need;
    }
    launch_gl_modelview(h->stream, nframes, cap, h->buf, h->d_gl);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(modelview, h->d_gl, need, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(n_out, h->buf.nmarkers, (size_t)nframes * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int f = 0; f < nframes; f++) n_out[f] = std::min(std::min(n_out[f], cap), h->buf.cap_markers);
    return ARUCOHIP_OK;
}

int arucohip_calculate_extrinsics(arucohip_handle* h, arucohip_marker_t* markers, int n, const float* K, const float* dist, int ndist,
                                  float marker_size, int y_perp) {
    if (!h || !markers || n < 0 || !K) return ARUCOHIP_E_INVALID;
    if (!(marker_size > 0)) return fail(h, ARUCOHIP_E_INVALID, "marker size must be positive");   // marker.cpp:114
    if (n == 0) return ARUCOHIP_OK;
    if ((size_t)n > (size_t)h->cap_frames * h->buf.cap_markers) return fail(h, ARUCOHIP_E_CAPACITY, "too many markers for this handle");
    HIPCHK(h, hipSetDevice(h->device));
    CamModel cam;
    int rc = make_cam(h, K, dist, ndist, marker_size, y_perp, &cam);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->buf.markers, markers, (size_t)n * sizeof(arucohip_marker_t), hipMemcpyHostToDevice, h->stream));
    launch_marker_pose(h->stream, h->buf.markers, n, cam);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(markers, h->buf.markers, (size_t)n * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ARUCOHIP_OK;
}

// BoardDetector::detect (boarddetector.cpp:90-205). The id filter and point gathering are a few hundred bytes of host
// glue; both solvePnP calls and the reprojection run on the device.
int arucohip_board_detect(arucohip_handle* h, const arucohip_marker_t* markers, int n, const int32_t* ids, const float* obj, int nboard,
                          int info_type, const float* K, const float* dist, int ndist, float marker_size, float repj_err_thres, int y_perp,
                          arucohip_marker_t* out_markers, arucohip_board_t* out, float* prob) {
    if (!h || !out || !prob || n < 0 || (n > 0 && (!markers || !out_markers))) return ARUCOHIP_E_INVALID;
    if (nboard <= 0 || !ids || !obj) return fail(h, ARUCOHIP_E_BOARD_CONFIG, "invalid BoardConfig that is empty");
    std::memset(out, 0, sizeof(*out));
    *prob = 0;
    auto onorm = [&](int a, int b) {
        // Point3f difference in float, cv::norm in double
        float dx = obj[3 * a] - obj[3 * b], dy = obj[3 * a + 1] - obj[3 * b + 1], dz = obj[3 * a + 2] - obj[3 * b + 2];
        return std::sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
    };
    float ssize = -1;
    if (info_type == ARUCOHIP_BOARD_PIX && marker_size > 0)
        ssize = marker_size;
    else if (info_type == ARUCOHIP_BOARD_METERS)
        ssize = (float)onorm(0, 1);
    std::vector<int> slot;
    int nb = 0;
    for (int i = 0; i < n; i++) {
        const int32_t* f = std::find(ids, ids + nboard, markers[i].id);
        if (f == ids + nboard) continue;
        out_markers[nb] = markers[i];
        out_markers[nb].ssize = ssize;
        slot.push_back((int)(f - ids));
        nb++;
    }
    out->n_markers = nb;
    if (nb == 0 || !K) return ARUCOHIP_OK;
    bool enough = (marker_size > 0 && info_type == ARUCOHIP_BOARD_PIX) || info_type == ARUCOHIP_BOARD_METERS;
    if (!enough) return ARUCOHIP_OK;
    double mpp = info_type == ARUCOHIP_BOARD_PIX ? marker_size / onorm(0, 1) : 1;
    std::vector<float> o3, i2;
    for (int i = 0; i < nb; i++)
        for (int p = 0; p < 4; p++) {
            i2.push_back(out_markers[i].corners[2 * p]), i2.push_back(out_markers[i].corners[2 * p + 1]);
            const float* q = obj + ((size_t)slot[i] * 4 + p) * 3;
            for (int c = 0; c < 3; c++) o3.push_back((float)(q[c] * mpp));
        }
    int npts = nb * 4;
    // d_small_f holds obj[3 npts] + img[2 npts] + the reprojected points [2 npts]
    if (npts * 7 > 8192) return fail(h, ARUCOHIP_E_CAPACITY, "board with too many points");
    HIPCHK(h, hipSetDevice(h->device));
    float zeros[4] = {0, 0, 0, 0};
    if (!dist || ndist == 0) dist = zeros, ndist = 4;
    CamModel cam;
    int rc = make_cam(h, K, dist, ndist, marker_size, y_perp, &cam);
    if (rc) return rc;
    float* d_obj = h->d_small_f;
    float* d_img = h->d_small_f + 3 * npts;
    double rt[6];
    int ok = 0;
    auto solve = [&](int m) -> int {
        HIPCHK(h, hipMemcpyAsync(d_obj, o3.data(), 3 * m * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_img, i2.data(), 2 * m * sizeof(float), hipMemcpyHostToDevice, h->stream));
        launch_pnp_points(h->stream, d_obj, d_img, m, cam, h->d_small_d, h->d_small_i);
        HIPCHK(h, hipGetLastError());
        return ARUCOHIP_OK;
    };
    if ((rc = solve(npts))) return rc;
    if (repj_err_thres > 0) {
        std::vector<float> rp(2 * npts);
        launch_project_points(h->stream, d_obj, npts, h->d_small_d, cam, d_img + 2 * npts);
        HIPCHK(h, hipMemcpyAsync(rp.data(), d_img + 2 * npts, 2 * npts * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<float> o3f, i2f;
        for (int i = 0; i < npts; i++) {
            float dx = rp[2 * i] - i2[2 * i], dy = rp[2 * i + 1] - i2[2 * i + 1];
            float err = (float)std::sqrt((double)dx * dx + (double)dy * dy);
            if (err < repj_err_thres) {
                for (int c = 0; c < 3; c++) o3f.push_back(o3[3 * i + c]);
                i2f.push_back(i2[2 * i]), i2f.push_back(i2[2 * i + 1]);
            }
        }
        o3.swap(o3f), i2.swap(i2f);
        // fewer than 4 surviving points: the reference's second cv::solvePnP would throw; like the batched kernel the
        // board then has no pose
        if (i2.size() / 2 < 4) {
            *prob = float(nb) / float(nboard);
            return ARUCOHIP_OK;
        }
        if ((rc = solve((int)(i2.size() / 2)))) return rc;
    }
    if (y_perp) launch_rotate_x(h->stream, h->d_small_d);
    HIPCHK(h, hipMemcpyAsync(rt, h->d_small_d, sizeof(rt), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&ok, h->d_small_i, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out->has_pose = ok;
    for (int k = 0; k < 3; k++) out->rvec[k] = rt[k], out->tvec[k] = rt[3 + k];
    *prob = float(nb) / float(nboard);
    return ARUCOHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Batches in flight. No reference counterpart (MarkerDetector::detect is synchronous); this is how a stream of batches
// keeps the GPU busy: the tail of a batch is a chain of dependent border steps that a handful of wavefronts work on, the
// head of the next batch is a streaming kernel that wants the whole chip.
// ---------------------------------------------------------------------------------------------
extern "C" {

int arucohip_set_pipeline_depth(arucohip_handle* h, int depth) {
    if (!h || depth < 0 || depth > 8) return ARUCOHIP_E_INVALID;
    for (auto* l : h->lanes)
        if (l->pend.active) return fail(h, ARUCOHIP_E_INVALID, "a submitted batch has not been waited for");
    HIPCHK(h, hipSetDevice(h->device));
    for (auto* l : h->lanes) arucohip_destroy(l);
    h->lanes.clear();
    h->cur = nullptr, h->next_ticket = 0;
    if (depth == 0) return ARUCOHIP_OK;
    if (!h->ev_submit) HIPCHK(h, hipEventCreateWithFlags(&h->ev_submit, hipEventDisableTiming));
    for (int i = 0; i < depth; i++) {
        arucohip_handle* l = nullptr;
        int rc = arucohip_create_ex(&h->params, h->device, &h->lim, &l);
        if (rc != ARUCOHIP_OK) {   // all or nothing: a later submit must not run at a smaller depth than the caller asked for
            for (auto* made : h->lanes) arucohip_destroy(made);
            h->lanes.clear();
            return fail(h, rc, "creating a pipeline lane failed (no lanes kept)");
        }
        l->decoder_fn = h->decoder_fn, l->decoder_user = h->decoder_user;
        for (auto* k : l->kids) k->decoder_fn = h->decoder_fn, k->decoder_user = h->decoder_user;
        l->timing = h->timing;
        h->lanes.push_back(l);
        if (h->d_hrm && h->hrm_count > 0) {
            std::vector<uint64_t> codes(h->hrm_count);
            HIPCHK(h, hipMemcpy(codes.data(), h->d_hrm, codes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            if ((rc = arucohip_set_dictionary(l, h->hrm_n, h->hrm_count, codes.data(), h->hrm_tau0, h->hrm_rate))) {
                for (auto* made : h->lanes) arucohip_destroy(made);
                h->lanes.clear();
                return rc;
            }
        }
    }
    return ARUCOHIP_OK;
}

int arucohip_detect_batch_submit(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                                 int frames_on_device, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                                 arucohip_marker_t* out, int cap, int32_t* n_out, int out_on_device, int* ticket) {
    if (!h || !ticket) return ARUCOHIP_E_INVALID;
    if (h->lanes.empty()) return fail(h, ARUCOHIP_E_INVALID, "arucohip_set_pipeline_depth first");
    arucohip_handle* l = h->lanes[h->next_ticket % (int)h->lanes.size()];
    if (l->pend.active) return fail(h, ARUCOHIP_E_CAPACITY, "pipeline full: wait for the oldest ticket first");
    HIPCHK(h, hipSetDevice(h->device));
    // what the caller's stream has queued so far (the frames) is visible to the lane
    HIPCHK(h, hipEventRecord(h->ev_submit, h->stream));
    HIPCHK(h, hipStreamWaitEvent(l->stream, h->ev_submit, 0));
    int rc = detect_batch_impl(l, frames, nframes, W, H, row_stride, frame_stride, frames_on_device, 1, K, dist, ndist, marker_size, y_perp, out, cap, n_out,
                               out_on_device, true);
    if (rc) {
        h->err = l->err;
        return rc;
    }
    l->pend.active = true, l->pend.ticket = h->next_ticket, l->pend.nframes = nframes, l->pend.cap = cap, l->pend.out_on_device = out_on_device;
    l->pend.out = out, l->pend.n_out = n_out;
    *ticket = h->next_ticket++;
    return ARUCOHIP_OK;
}

// One bad frame must not void a batch (the reference has no limits at all, src/markerdetector.cpp:496-635): a frame whose lists overflowed
// comes back with n = -1 and everything else is valid. This call runs exactly those frames again, one at a time, on a one-frame handle
// whose per-frame lists are 4x (then 16x, 64x) the batch handle's, and patches their results into the caller's arrays.
int arucohip_detect_batch_retry_overflowed(arucohip_handle* h, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride,
                                           int frames_on_device, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                                           arucohip_marker_t* out, int cap, int32_t* n_out, int out_on_device, int* n_retried) {
    if (!h || !frames || !n_out || (cap > 0 && !out) || cap < 0 || nframes < 1) return ARUCOHIP_E_INVALID;
    if (n_retried) *n_retried = 0;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int32_t> n(nframes);
    if (out_on_device)
        HIPCHK(h, hipMemcpy(n.data(), n_out, (size_t)nframes * sizeof(int32_t), hipMemcpyDeviceToHost));
    else
        std::memcpy(n.data(), n_out, (size_t)nframes * sizeof(int32_t));
    std::vector<arucohip_marker_t> tmp((size_t)std::max(cap, 1));
    int ret = ARUCOHIP_OK;
    for (int f = 0; f < nframes; f++) {
        if (n[f] >= 0) continue;
        int32_t got = 0;
        int rc = ARUCOHIP_E_OVERFLOW;
        for (int attempt = 0; attempt < 3 && rc == ARUCOHIP_E_OVERFLOW; attempt++) {
            // 4x, 16x, 64x the batch handle's lists and never more: a cached handle is reused at its size, a frame that overflows 64x is reported
            const int want = h->retry ? std::min(64, attempt == 0 ? h->retry_mult : h->retry_mult * 4) : 4;
            if (attempt > 0 && h->retry && want == h->retry_mult) break;   // already at the cap
            if (!h->retry || want != h->retry_mult) {
                if (h->retry) arucohip_destroy(h->retry);
                h->retry = nullptr;
                arucohip_limits_t l = h->lim;
                l.max_batch = 1;
                auto grow = [&](int32_t v, long top) { return (int32_t)std::min<long>((long)v * want, top); };
                l.triggers_per_frame = grow(l.triggers_per_frame, 1L << 22), l.contours_per_frame = grow(l.contours_per_frame, 1L << 18);
                l.points_per_frame = grow(l.points_per_frame, 1L << 24), l.long_walks_per_plane = grow(l.long_walks_per_plane, 1L << 16);
                l.candidates_per_frame = std::min(512, l.candidates_per_frame * 2);
                int crc = arucohip_create_ex(&h->params, h->device, &l, &h->retry);
                if (crc != ARUCOHIP_OK) return fail(h, crc, "creating the retry handle failed");
                h->retry_mult = want;
                h->retry->decoder_fn = h->decoder_fn, h->retry->decoder_user = h->decoder_user;
                if (h->d_hrm && h->hrm_count > 0) {
                    std::vector<uint64_t> codes(h->hrm_count);
                    HIPCHK(h, hipMemcpy(codes.data(), h->d_hrm, codes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                    if ((crc = arucohip_set_dictionary(h->retry, h->hrm_n, h->hrm_count, codes.data(), h->hrm_tau0, h->hrm_rate))) return crc;
                }
            }
            h->retry->params = h->params;
            rc = arucohip_detect_batch(h->retry, frames + (size_t)f * frame_stride, 1, W, H, row_stride, frame_stride, frames_on_device, K, dist, ndist, marker_size,
                                       y_perp, tmp.data(), cap, &got, 0);
        }
        if (rc != ARUCOHIP_OK && rc != ARUCOHIP_E_CAPACITY) {
            if (ret == ARUCOHIP_OK) ret = fail(h, rc, h->retry ? h->retry->err.c_str() : "retry failed");
            continue;
        }
        if (rc == ARUCOHIP_E_CAPACITY && ret == ARUCOHIP_OK) ret = fail(h, rc, "marker output array too small");
        const int ncopy = std::min<int>(std::max<int>(got, 0), cap);
        if (out_on_device) {
            if (ncopy > 0) HIPCHK(h, hipMemcpy(out + (size_t)f * cap, tmp.data(), (size_t)ncopy * sizeof(arucohip_marker_t), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(n_out + f, &got, sizeof(int32_t), hipMemcpyHostToDevice));
        } else {
            if (ncopy > 0) std::memcpy(out + (size_t)f * cap, tmp.data(), (size_t)ncopy * sizeof(arucohip_marker_t));
            n_out[f] = got;
        }
        if (n_retried) (*n_retried)++;
    }
    return ret;
}

int arucohip_detect_batch_wait(arucohip_handle* h, int ticket) {
    if (!h || h->lanes.empty() || ticket < 0) return ARUCOHIP_E_INVALID;
    arucohip_handle* l = h->lanes[ticket % (int)h->lanes.size()];
    if (!l->pend.active || l->pend.ticket != ticket) return fail(h, ARUCOHIP_E_INVALID, "no such batch in flight");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = l->pend.out_on_device ? arucohip_batch_status(l) : collect_batch_host(l, l->pend.nframes, l->pend.out, l->pend.cap, l->pend.n_out);
    l->pend.active = false;
    h->cur = l;
    if (rc) h->err = l->err;
    return rc;
}

}  // extern "C"
