/*
 * arucohip — C ABI of the MI355X-native ArUco marker-detection hot path.
 *
 * This is the drop-in boundary for ONE path of the reference library (paroj/aruco, ArUco 1.3):
 *   aruco::MarkerDetector::detect()   /root/reference/src/markerdetector.h:102-120, .cpp:302-478
 *   aruco::BoardDetector::detect()    /root/reference/src/boarddetector.h:103-108, .cpp:90-205
 * plus the public stage entry points the reference keeps callable (markerdetector.h:255-280).
 * Everything is plain C: pointers, sizes, PODs. No OpenCV, no torch types. The header-only C++ shim in
 * include/aruco_hip_shim.hpp rebuilds the reference's classes on top of these calls.
 *
 * Threading: like the reference object (markerdetector.cpp:334,372-380 mutate members) a handle is not
 * re-entrant. One handle owns one HIP stream and all device buffers; use one handle per host thread.
 */
#ifndef ARUCOHIP_H
#define ARUCOHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARUCOHIP_VERSION 100

/* status codes; mirror the reference's failure modes (CV_Assert -> cv::Exception), SURVEY.md §8b "Errors" */
enum {
    ARUCOHIP_OK = 0,
    ARUCOHIP_E_INVALID = 1,      /* bad argument: what the reference rejects with CV_Assert (markerdetector.cpp:644,685,1032-1034,1048) */
    ARUCOHIP_E_CAPACITY = 2,     /* output array too small; *n_out holds the required count */
    ARUCOHIP_E_UNSUPPORTED = 3,  /* a parameter value outside what the device kernels are built for (warp size > 128, adaptive block > 31,
                                    SUBPIX window > 15, locked-corner window > 31, dictionary markers beyond 8x8 / 4096 entries) */
    ARUCOHIP_E_HIP = 4,          /* HIP runtime failure, see arucohip_last_error_string */
    ARUCOHIP_E_OVERFLOW = 5,     /* an internal device list overflowed for some frame(s): those have n_out = -1, the others are valid
                                    (arucohip_detect_batch_retry_overflowed, or raise the limits with arucohip_create_ex) */
    ARUCOHIP_E_BOARD_CONFIG = 6  /* empty board configuration (boarddetector.cpp:93) */
};

/* MarkerDetector::ThresholdMethods (markerdetector.h:125) */
enum { ARUCOHIP_THRES_FIXED = 0, ARUCOHIP_THRES_ADPT = 1, ARUCOHIP_THRES_CANNY = 2 };
/* MarkerDetector::CornerRefinementMethod (markerdetector.h:192) */
enum { ARUCOHIP_CORNER_NONE = 0, ARUCOHIP_CORNER_HARRIS = 1, ARUCOHIP_CORNER_SUBPIX = 2, ARUCOHIP_CORNER_LINES = 3 };
/* BoardConfiguration::mInfoType (board.h:64) */
enum { ARUCOHIP_BOARD_NONE = -1, ARUCOHIP_BOARD_PIX = 0, ARUCOHIP_BOARD_METERS = 1 };
/* decoder behind MarkerDetector::setMakerDetectorFunction (markerdetector.h:243) */
enum { ARUCOHIP_DECODER_FIDUCIAL_5X5 = 0, ARUCOHIP_DECODER_HRM = 1, ARUCOHIP_DECODER_USER = 2 };

/* 1:1 image of MarkerDetector's private configuration members (markerdetector.h:283-306; defaults .cpp:235-249). */
typedef struct arucohip_params {
    int32_t thres_method;       /* _thresMethod        default ADPT */
    int32_t thres_param1_range; /* _thresParam1_range  default 0    */
    double thres_param1;        /* _thresParam1        default 7    */
    double thres_param2;        /* _thresParam2        default 7    */
    int32_t corner_method;      /* _cornerMethod       default LINES */
    int32_t warp_size;          /* _markerWarpSize     default 56 (>= 10, multiple of 7 recommended) */
    float min_size;             /* _minSize            default 0.04 */
    float max_size;             /* _maxSize            default 0.5  */
    float border_dist;          /* _borderDistThres    default 0.025 */
    int32_t use_locked_corners; /* _useLockedCorners   default 0; with HARRIS / SUBPIX: findCornerMaxima before the refinement */
    int32_t decoder_kind;       /* markerIdDetectorFunc: ARUCOHIP_DECODER_FIDUCIAL_5X5, ARUCOHIP_DECODER_HRM after
                                   arucohip_set_dictionary, ARUCOHIP_DECODER_USER after arucohip_set_decoder_callback */
    int32_t erode;              /* north_star's "optional erosion": != 0 erodes the thresholded image with a 3x3 structuring
                                   element (cv::erode(thres, thres, Mat()) semantics, outside pixels do not erode) before the
                                   contour stage. Default 0: this snapshot of the reference dropped its enableErosion() as a
                                   no-op (PortingManual.md:5-7), so there is no reference result to match */
} arucohip_params_t;

/* aruco::Marker (marker.h:46-53): id, 4 corners (x0,y0,..), ssize, Rvec/Tvec as double. 96 bytes. */
typedef struct arucohip_marker {
    int32_t id;
    float corners[8];
    float ssize;
    int32_t has_pose;
    int32_t pad_;
    double rvec[3];
    double tvec[3];
} arucohip_marker_t;

/* aruco::Board pose part (board.h:99-104); the member markers are returned in a caller array. */
typedef struct arucohip_board {
    int32_t n_markers;
    int32_t has_pose;
    double rvec[3];
    double tvec[3];
} arucohip_board_t;

/* device-side limits of one handle */
typedef struct arucohip_limits {
    int32_t max_width, max_height;
    int32_t max_batch;             /* frames per arucohip_detect_batch call */
    int32_t max_thres_planes;      /* 2*range+1 supported per frame */
    int32_t triggers_per_frame;    /* border-start candidates (average per frame) */
    int32_t contours_per_frame;    /* borders that pass the size filter */
    int32_t points_per_frame;      /* contour-point pool (average per frame) */
    int32_t candidates_per_frame;  /* quads per frame */
    int32_t markers_per_frame;     /* device-side marker slots per frame */
    int32_t long_walks_per_plane;  /* borders followed beyond the first 64 steps, per threshold plane and kind (outer /
                                      hole); each holds a checkpoint ring of max contour length / 16 words in HBM */
} arucohip_limits_t;

typedef struct arucohip_handle arucohip_handle;

int arucohip_version(void);
/* "src=<digest of the library's sources> flags=[<extra compiler flags of a variant build>]": which build is loaded. An experiment build
 * (tools/stage_cost.sh: -DARUCOHIP_STAGE_EXPERIMENT truncates the pipeline) says so here, and bench.py refuses to print a headline for it. */
const char* arucohip_build_info(void);
void arucohip_default_params(arucohip_params_t* p);
void arucohip_default_limits(arucohip_limits_t* l, int max_width, int max_height, int max_batch);

/* Create a detector on HIP device `device` able to take frames up to max_width x max_height, `max_batch` at a time.
 * params may be NULL (reference defaults). Geometry: a handle is at least 32 x 32 and at most 16383 x 16383 (14-bit coordinates in the
 * border checkpoints) with max_width * max_height <= 2^26 (raster keys); a frame is 1 x 1 up to the handle's size, the limits hold per
 * dimension (a frame may be smaller than the handle, not wider or taller). Anything outside is ARUCOHIP_E_INVALID; the reference has no
 * upper limits. */
int arucohip_create(const arucohip_params_t* params, int device, int max_width, int max_height, int max_batch,
                    arucohip_handle** out);
int arucohip_create_ex(const arucohip_params_t* params, int device, const arucohip_limits_t* limits, arucohip_handle** out);
void arucohip_destroy(arucohip_handle* h);
/* setters of MarkerDetector (markerdetector.h:129-245) collapse into one call; validates like setMinMaxSize /
 * setWarpSize (markerdetector.cpp:1031-1051). */
int arucohip_set_params(arucohip_handle* h, const arucohip_params_t* p);
int arucohip_get_params(const arucohip_handle* h, arucohip_params_t* p);
const char* arucohip_last_error_string(const arucohip_handle* h);

/* Run the handle's work on an existing HIP stream (hipStream_t passed as void*); NULL restores the handle's own. */
int arucohip_set_stream(arucohip_handle* h, void* hip_stream);
void* arucohip_get_stream(arucohip_handle* h);
int arucohip_synchronize(arucohip_handle* h);
/* Frames produced on ANOTHER stream: everything the handle enqueues from now on (the next arucohip_detect_batch with device frames, the
 * lane that takes the next arucohip_detect_batch_submit) first waits for `hip_event` (hipEvent_t passed as void*), which the producer
 * recorded on its stream behind the last write to the frames. This is the stream-ordered alternative to synchronising the producer. */
int arucohip_wait_event(arucohip_handle* h, void* hip_event);

/* MarkerDetector::detect (markerdetector.h:102-103) for one 8-bit gray frame in HOST memory.
 * K: 9 floats row-major or NULL (no pose); dist: ndist (0,4,5,8) floats or NULL; marker_size <= 0 -> no pose.
 * out/cap: caller array; *n_out = number of markers (sorted by id). */
int arucohip_detect(arucohip_handle* h, const uint8_t* gray, int width, int height, size_t row_stride, const float* K,
                    const float* dist, int ndist, float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap,
                    int* n_out);

/* Same for a batch of nframes equally sized frames (frame f at frames + f*frame_stride).
 * frames_on_device != 0: `frames` is a device pointer (frames already resident in HBM). The kernels read it on the
 *                       handle's stream: frames produced on another stream must be ordered before the call — record an event
 *                       behind the producer and pass it to arucohip_wait_event, run the handle on the producer's stream
 *                       (arucohip_set_stream), or synchronise the producer.
 * out_on_device   != 0: `out` (nframes*cap markers) and `n_out` (nframes int32) are device pointers, the call is
 *                       asynchronous on the handle's stream and reports only launch errors; otherwise host arrays and
 *                       the call returns when the results are there. */
int arucohip_detect_batch(arucohip_handle* h, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                          size_t frame_stride, int frames_on_device, const float* K, const float* dist, int ndist,
                          float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out,
                          int out_on_device);
/* SURVEY §8 row f1 — highly reliable markers: HighlyReliableMarkers::loadDictionary (src/highlyreliablemarkers.cpp:311-329).
 * codes[i] = marker i of the dictionary as n*n bits, bit y*n+x = cell (y, x) ('1' in aruco::Dictionary's bit strings),
 * n <= 8, count <= 4096; tau0 = Dictionary::tau0,
 * correction_rate = correctionDistanceRate (the reference's default is 1). With params.decoder_kind = ARUCOHIP_DECODER_HRM
 * the candidates are then decoded like HighlyReliableMarkers::detect (:332-383) — id = position in the dictionary — which
 * is what MarkerDetector::setMakerDetectorFunction(HighlyReliableMarkers::detect) selects in the reference; the warp size
 * should be a multiple of n + 2 (the reference's apps use (n + 2) * 8). count = 0 drops the dictionary.
 * For n >= 6 the reference's exact-match shortcut compares 32-bit ids built with `2 << bit` (:137-138), which overflow;
 * the nearest-entry search it falls back to — the intended behaviour, identical whenever the ids are unique — is what runs
 * here for every n. */
int arucohip_set_dictionary(arucohip_handle* h, int n, int count, const uint64_t* codes, int tau0, float correction_rate);

/* SURVEY §8b, plugin boundary — MarkerDetector::setMakerDetectorFunction (src/markerdetector.h:243-245) with a function of
 * the caller's own: typedef int (*MarkerdetectorFunc)(const cv::Mat& in, int& nRotations) (:78; contract :65-77: `in` is
 * the square canonical view of a candidate, the return value is the marker id or -1, nRotations the number of 90-degree
 * clockwise turns that bring the candidate to its canonical orientation). With params.decoder_kind =
 * ARUCOHIP_DECODER_USER the device still warps every candidate (MarkerDetector::warp, :353), the size x size patches come
 * back to the host, `fn` is called once per candidate in the reference's order (frame by frame, candidates in
 * detectRectangles order, *n_rotations preset to 0) and the rest of the pipeline — corner refinement, rotation of the
 * corners, sorting, duplicate and border filters, pose — continues on the device with the ids it returned. `patch` is a
 * scratch copy the function may overwrite (FiducidalMarkers::detect thresholds its input in place, arucofidmarkers.cpp:446).
 * A batch call then blocks until the decoders have run, also with out_on_device. fn = NULL removes the callback. */
typedef int (*arucohip_decoder_fn)(void* user, uint8_t* patch, int size, int* n_rotations);
int arucohip_set_decoder_callback(arucohip_handle* h, arucohip_decoder_fn fn, void* user);

/* SURVEY §8 row f3 — frames with three interleaved 8-bit channels in B,G,R order (what cv::imread / cv::VideoCapture
 * deliver; row_stride >= 3*width): MarkerDetector::detect converts them with cv::cvtColor(CV_BGR2GRAY)
 * (src/markerdetector.cpp:307-310); here the conversion runs on the device, bit-identical to OpenCV's 8-bit fixed-point
 * form (B*1868 + G*9617 + R*4899 + 8192) >> 14, and the gray frames then take the path of arucohip_detect_batch. */
int arucohip_detect_bgr(arucohip_handle* h, const uint8_t* bgr, int width, int height, size_t row_stride, const float* K,
                        const float* dist, int ndist, float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap,
                        int* n_out);
int arucohip_detect_batch_bgr(arucohip_handle* h, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                              size_t frame_stride, int frames_on_device, const float* K, const float* dist, int ndist,
                              float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out,
                              int out_on_device);
/* The conversion alone: one host BGR frame -> host gray frame (width*height bytes). */
int arucohip_bgr_to_gray(arucohip_handle* h, const uint8_t* bgr, int width, int height, size_t row_stride, uint8_t* gray);
/* SURVEY §8 row f3 — lens undistortion of the frames on the device: cv::undistort(src, dst, CameraMatrix, Distorsion), which
 * the reference's GL apps run on every frame before detect() (utils/aruco_test_gl.cpp:237-240, utils/aruco_test_board_gl.cpp:
 * 265-268; detect is then called with an empty distortion vector). 8-bit frames with `channels` = 1 or 3 interleaved channels;
 * dst is tightly packed (nframes x height x width x channels), in host or device memory; with dst_on_device the call is
 * asynchronous on the handle's stream, so arucohip_detect_batch(frames_on_device = 1) can follow without a round trip. The
 * fixed-point map (OpenCV's CV_16SC2 form: bilinear, 5 fractional bits, constant 0 outside) is computed once per (size, K, dist)
 * and kept in the handle. */
int arucohip_undistort(arucohip_handle* h, const uint8_t* src, int nframes, int width, int height, size_t row_stride, size_t frame_stride,
                       int channels, int src_on_device, const float* K, const float* dist, int ndist, uint8_t* dst, int dst_on_device);
/* After an asynchronous batch: synchronise and report device-side overflow / capacity conditions. */
int arucohip_batch_status(arucohip_handle* h);
/* Device lists are finite (arucohip_limits_t), the reference's vectors are not (src/markerdetector.cpp:496-635). When a list overflows
 * the call returns ARUCOHIP_E_OVERFLOW, but only the frames it happened to are given up: their n_out is -1, every other frame of the batch
 * holds its complete result. arucohip_detect_batch_retry_overflowed takes the arguments of the batch call that returned the code (after it
 * has completed: arucohip_batch_status / _wait for device outputs) and runs the frames with n_out = -1 again, one at a time, on an internal
 * one-frame handle whose per-frame lists are 4x (16x, 64x) larger, patching out / n_out in place; *n_retried = frames redone. */
int arucohip_detect_batch_retry_overflowed(arucohip_handle* h, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                                           size_t frame_stride, int frames_on_device, const float* K, const float* dist, int ndist,
                                           float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out,
                                           int out_on_device, int* n_retried);
/* With the environment variable ARUCOHIP_STREAMS = 2..8 a large batch is processed as that many chunks of consecutive
 * frames on separate HIP streams (default 1) that fork from and join the handle's stream, so the caller sees one
 * stream-ordered call; host frames of chunk i+1 are copied while chunk i computes. Returns the number of chunks of the
 * last batch and, if not NULL, the frames per chunk — every kernel launch covers one chunk. No reference counterpart. */
int arucohip_batch_chunks(arucohip_handle* h, int* frames_per_chunk);

/* Batches in flight (no reference counterpart: MarkerDetector::detect is synchronous). The tail of a batch — border
 * following of the longest contours, decoding — is latency bound and leaves most of the chip idle, the head of the next
 * batch is a streaming kernel: with depth >= 2 a stream of batches overlaps them. arucohip_set_pipeline_depth creates
 * `depth` complete workers (own buffers, own stream; 0 removes them); arucohip_detect_batch_submit takes the arguments
 * of arucohip_detect_batch, enqueues the batch on worker ticket mod depth behind everything queued on the handle's stream so
 * far and returns at once; arucohip_detect_batch_wait(ticket) returns when that batch is complete (its status code: overflow /
 * capacity conditions as arucohip_detect_batch / arucohip_batch_status would report them), host outputs are filled then.
 * At most `depth` tickets can be outstanding (ARUCOHIP_E_CAPACITY otherwise); getters and arucohip_board_detect_batch
 * address the batch of the last ticket waited for. Input frames and output arrays of a ticket must stay untouched until
 * its wait returns. */
int arucohip_set_pipeline_depth(arucohip_handle* h, int depth);
int arucohip_detect_batch_submit(arucohip_handle* h, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                                 size_t frame_stride, int frames_on_device, const float* K, const float* dist, int ndist,
                                 float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out,
                                 int out_on_device, int* ticket);
int arucohip_detect_batch_wait(arucohip_handle* h, int ticket);

/* MarkerDetector::getThresholdedImage (markerdetector.h:183): thresholded image of frame `frame` of the last call
 * (the middle one when thres_param1_range > 0), copied to host `dst` (width*height bytes, tightly packed). The hot path keeps the image
 * as bit tiles plus its four border lines (what cv::findContours works on); this call expands the requested plane to the reference's
 * 0 / 255 bytes. ARUCOHIP_THRES_BYTES=1 (environment, read at handle creation) writes the bytes during detection instead. */
int arucohip_get_thresholded(arucohip_handle* h, int frame, uint8_t* dst);
/* MarkerDetector::getCandidates (markerdetector.h:266): quads that were rectangles but not markers. quads: cap*8 floats. */
int arucohip_get_candidates(arucohip_handle* h, int frame, float* quads, int cap, int* n);

/* Stage entry points the reference keeps public (markerdetector.h:255-280), on host buffers. */
int arucohip_threshold(arucohip_handle* h, int method, const uint8_t* gray, int width, int height, size_t row_stride,
                       double param1, double param2, uint8_t* dst);
int arucohip_detect_rectangles(arucohip_handle* h, const uint8_t* thres, int width, int height, size_t row_stride,
                               float* quads, int cap, int* n);
int arucohip_warp(arucohip_handle* h, const uint8_t* gray, int width, int height, size_t row_stride, const float quad[8],
                  int size, uint8_t* dst);

/* MarkerDetector::refineCandidateLines(MarkerCandidate&, camMatrix, distCoeff) (markerdetector.h:280, .cpp:931-997), the LINES
 * corner refinement as a stage of its own: contour_xy = the candidate's contour (MarkerCandidate::contour, markerdetector.h:60: npoints
 * cv::Point = int32 pairs x0,y0,x1,y1,... in the order cv::findContours produced them, reversed if detectRectangles swapped the corners),
 * corners = the candidate's four corners on input (they must be contour points: every corner is looked up in the contour after rounding
 * to int like Point(candidate[k]), the last match wins) and the four intersections of the sides' least-squares lines on output. With K
 * (9 floats) and dist (ndist > 0) the contour is undistorted before the fit and the corners are distorted again, as the reference does
 * when both matrices are non-empty. Coordinates are image pixels, 0 <= x, y <= 32767; npoints at most the handle's
 * points_per_frame. Uses the handle's candidate and point lists: results of the last batch are gone afterwards (like the other stage calls). */
int arucohip_refine_candidate_lines(arucohip_handle* h, const int32_t* contour_xy, int npoints, float corners[8], const float* K,
                                    const float* dist, int ndist);

/* Stage inspection for parity tests (results of the last detect/detect_batch/detect_rectangles call).
 * Contours that passed the size filter, in the reference's cv::findContours(RETR_LIST) relative order. */
int arucohip_debug_num_contours(arucohip_handle* h, int frame, int* n);
int arucohip_debug_contour(arucohip_handle* h, int frame, int index, int* is_hole, int* start_x, int* start_y, int16_t* xy,
                           int cap_points, int* n_points);
/* Candidates after detectRectangles in reference order: integer quad, decoded id (-1 none), nRotations. */
int arucohip_debug_candidates(arucohip_handle* h, int frame, float* quads0, int32_t* ids, int32_t* nrot, int cap, int* n);
/* Otsu threshold (cv::threshold THRESH_OTSU inside the decoders, arucofidmarkers.cpp:169 / highlyreliablemarkers.cpp:346) of every candidate's patch,
 * same order; -1 where the decode stage did not run for the candidate. */
int arucohip_debug_otsu(arucohip_handle* h, int frame, int32_t* thr, int cap, int* n);

/* Device list fill levels of the last batch: [0] border-start candidates, [1] borders kept, [2] contour points,
 * [3] overflow bits. For sizing arucohip_limits_t. */
int arucohip_debug_counters(arucohip_handle* h, uint32_t* out8);

/* BoardDetector::detect (boarddetector.h:103-108). markers: output of arucohip_detect; ids/obj: BoardConfiguration
 * (board.h:56-69) as nboard ids and nboard*4*3 floats; returns likelihood in *prob (found / total).
 * out_markers (cap n) receives the board's member markers (Board : vector<Marker>). */
int arucohip_board_detect(arucohip_handle* h, const arucohip_marker_t* markers, int n, const int32_t* ids, const float* obj,
                          int nboard, int info_type, const float* K, const float* dist, int ndist, float marker_size,
                          float repj_err_thres, int y_perpendicular, arucohip_marker_t* out_markers, arucohip_board_t* out,
                          float* prob);

/* BoardDetector::detect for every frame of the LAST arucohip_detect_batch call, on its device-resident markers (one
 * wavefront per frame, wave-parallel solvePnP over all board corners). out / prob: host arrays of nframes entries. The
 * board's member markers are the detected markers whose id is in `ids`, in the same order. */
int arucohip_board_detect_batch(arucohip_handle* h, int nframes, const int32_t* ids, const float* obj, int nboard, int info_type,
                                const float* K, const float* dist, int ndist, float marker_size, float repj_err_thres,
                                int y_perpendicular, arucohip_board_t* out, float* prob);

/* Marker::calculateExtrinsics (marker.h:98-104 / marker.cpp:112-124) for n markers at once (batched solvePnP). */
int arucohip_calculate_extrinsics(arucohip_handle* h, arucohip_marker_t* markers, int n, const float* K, const float* dist,
                                  int ndist, float marker_size, int y_perpendicular);

/* Execution time of the dominant streaming kernel (the 16-pixel-per-lane adaptive threshold kernel) from the device's constant-rate
 * clock: every wave leaves its first and last reading, *total_ms = sum over the launches since arucohip_enable_timing(h, 1) of
 * (last wave's end - first wave's start), *launches = their number (0 when another threshold kernel ran: use the event times).
 * Unlike the hipEvent interval of arucohip_kernel_times this excludes the time a dispatch queues behind other batches' kernels
 * when several batches are in flight; it is what rocprofv3 --kernel-trace reports per dispatch. No reference counterpart. */
int arucohip_threshold_exec_ms(arucohip_handle* h, double* total_ms, int* launches);

/* Per-stage device time per batch in milliseconds (hipEvent pairs on the handle's stream), the reference's
 * ARUCO_MARKER_BENCHMARK stages (markerdetector.cpp:472-476): names via arucohip_stage_name. Returns count. */
int arucohip_stage_times(arucohip_handle* h, float* ms, int cap);
const char* arucohip_stage_name(int i);
/* on = 1 starts recording (and resets the average); up to 32 batches are averaged. on = 2: no hipEvents, only the device-clock stamps behind
 * arucohip_threshold_exec_ms (an otherwise uninstrumented run). on = 0 stops both. */
int arucohip_enable_timing(arucohip_handle* h, int on);
/* Per-kernel average device time (ms per batch) since arucohip_enable_timing; names via arucohip_kernel_name. */
int arucohip_kernel_times(arucohip_handle* h, float* ms, int cap);
const char* arucohip_kernel_name(int i);

/* ---- Frame sharding over the GPUs of one node (no reference counterpart — the reference is single-process
 * CPU code; the caller shape is the frame loop of utils/aruco_test.cpp:140-160 with one detector per GPU behind one call).
 * Frames are independent units: frame f goes to device slot f mod G, there is no collective on the data path. The per-frame
 * marker blocks {n, arucohip_marker_t[cap]} are gathered once per call: each slot copies its block to pinned host memory, or
 * with ARUCOHIP_MGPU_GATHER_PEER device-to-device (xGMI) into one buffer on the first device that a single copy brings to
 * the host. One host thread per slot keeps the devices busy concurrently. (Across PROCESSES, one rank per GPU, the same
 * blocks are gathered with RCCL: bench.py / aruco_amd/dist.py.) */
enum { ARUCOHIP_MGPU_GATHER_HOST = 0, ARUCOHIP_MGPU_GATHER_PEER = 1 };
typedef struct arucohip_mgpu arucohip_mgpu;
int arucohip_mgpu_device_count(void);
/* devices: ndevices HIP device ids (NULL: 0..ndevices-1; an id may repeat, every slot gets its own handle and stream).
 * Every slot takes up to max_frames_per_device frames per call; cap = marker slots per frame in the gathered blocks. */
int arucohip_mgpu_create(const arucohip_params_t* params, const int* devices, int ndevices, int max_width, int max_height,
                         int max_frames_per_device, int cap, int flags, arucohip_mgpu** out);
void arucohip_mgpu_destroy(arucohip_mgpu* m);
int arucohip_mgpu_size(const arucohip_mgpu* m);
arucohip_handle* arucohip_mgpu_handle(arucohip_mgpu* m, int slot);   /* a slot's own handle (dictionary, callback, timing) */
int arucohip_mgpu_set_params(arucohip_mgpu* m, const arucohip_params_t* p);
const char* arucohip_mgpu_last_error_string(const arucohip_mgpu* m);
/* nframes host frames (frame f at frames + f*frame_stride), frame f -> slot f mod G; out[f*cap ..], n_out[f] in frame order. */
int arucohip_mgpu_detect_batch(arucohip_mgpu* m, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                               size_t frame_stride, const float* K, const float* dist, int ndist, float marker_size,
                               int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out);
/* One camera stream per slot, frames already resident in that slot's HBM (BASELINE config 5): frames_dev[g] = device pointer
 * on slot g's device, nframes[g] <= max_frames_per_device. Results camera-major: frame j of slot g at index
 * g*max_frames_per_device + j of out (x cap) and n_out. */
int arucohip_mgpu_detect_streams(arucohip_mgpu* m, const uint8_t* const* frames_dev, const int* nframes, int width, int height,
                                 size_t row_stride, size_t frame_stride, const float* K, const float* dist, int ndist,
                                 float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out);

/* Asynchronous form (round 3): every device slot keeps `depth` batches in flight (arucohip_set_pipeline_depth on its handle) and has one
 * persistent host thread, created with the detector, that submits a ticket's sub-batch as soon as a lane of its handle is free and waits
 * for the oldest otherwise. arucohip_mgpu_set_depth(m, d): 1 <= d <= 8 tickets outstanding (default 1; rebuilds the lanes, nothing may be
 * in flight). arucohip_mgpu_submit_batch / _submit_streams take the arguments of arucohip_mgpu_detect_batch / _detect_streams and return
 * at once with a ticket; arucohip_mgpu_wait(ticket) blocks until every slot's sub-batch is complete and the gathered blocks are in `out` /
 * `n_out` (same layout as the synchronous calls, which are submit + wait). Frames and output arrays of a ticket stay untouched until
 * its wait returns; tickets are waited for in the order they were submitted. */
int arucohip_mgpu_set_depth(arucohip_mgpu* m, int depth);   /* transactional: on failure the previous depth keeps running */
/* The gather in use: ARUCOHIP_MGPU_GATHER_PEER only if it was asked for AND every device can reach the first one (hipDeviceCanAccessPeer);
 * otherwise the blocks go through pinned host memory (ARUCOHIP_MGPU_GATHER_HOST). */
int arucohip_mgpu_gather_mode(const arucohip_mgpu* m);
int arucohip_mgpu_submit_batch(arucohip_mgpu* m, const uint8_t* frames, int nframes, int width, int height, size_t row_stride,
                               size_t frame_stride, const float* K, const float* dist, int ndist, float marker_size,
                               int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out, int* ticket);
int arucohip_mgpu_submit_streams(arucohip_mgpu* m, const uint8_t* const* frames_dev, const int* nframes, int width, int height,
                                 size_t row_stride, size_t frame_stride, const float* K, const float* dist, int ndist,
                                 float marker_size, int y_perpendicular, arucohip_marker_t* out, int cap, int32_t* n_out, int* ticket);
int arucohip_mgpu_wait(arucohip_mgpu* m, int ticket);

/* Across PROCESSES (one rank per GPU, RCCL): the block a rank contributes to the gather. arucohip_compact_markers packs the device arrays a
 * batch left with out_on_device (blocks_dev: nframes x cap markers, counts_dev: nframes int32) into ONE contiguous device block
 *   { int32 total, nframes, cap_total, overflow;  int32 counts[nframes] (padded to 16 bytes);  arucohip_marker_t markers[cap_total] }
 * with the frames' markers back to back (frame f at offset sum over j < f of min(max(counts[j], 0), cap)); overflow != 0 says that total
 * exceeded cap_total and the tail is missing. One kernel on `hip_stream` (the current device), no handle, no host round trip;
 * arucohip_compact_bytes gives the size of the block. At the bench's config 2 the block is a third of the fixed-capacity arrays.
 * No reference counterpart. */
size_t arucohip_compact_bytes(int nframes, int cap_total);
int arucohip_compact_markers(const arucohip_marker_t* blocks_dev, const int32_t* counts_dev, int nframes, int cap, void* dst_dev,
                             int cap_total, void* hip_stream);

/* ---- OpenGL / Ogre conversions of the pose results (host arithmetic, no handle, no device work).
 * GetGLModelViewMatrix (src/utils.cpp:32-69; Marker::glGetModelViewMatrix src/marker.h:90, Board:: src/board.h:109):
 * column-major 4x4 from rvec / tvec (3 doubles each). */
int arucohip_gl_modelview(const double* rvec, const double* tvec, double* modelview16);
/* The same for n markers that carry a pose (has_pose), modelview16: n*16 doubles. */
int arucohip_gl_modelview_n(const arucohip_marker_t* markers, int n, double* modelview16);
/* ... and for every marker of the LAST batch on the device (one launch, a lane per marker): modelview = host array of
 * nframes*cap*16 doubles (frame-major, zero matrices for slots without a posed marker), n_out[f] = markers of frame f. */
int arucohip_gl_modelview_batch(arucohip_handle* h, int nframes, int cap, double* modelview, int32_t* n_out);
/* GetOgrePoseParameters (src/utils.cpp:71-147): position[3], orientation[4] = quaternion (w, x, y, z). */
int arucohip_ogre_pose(const double* rvec, const double* tvec, double* position3, double* orientation4);
/* CameraParameters::glGetProjectionMatrix (src/cameraparameters.cpp:226-266) including the CameraParameters::resize
 * (:166-179) to width x height it starts with. K: 9 floats row-major, valid for cam_width x cam_height (= CamSize, which
 * the reference keeps using for the right / bottom planes after the resize). */
int arucohip_gl_projection(const float* K, int cam_width, int cam_height, int width, int height, double gnear, double gfar,
                           int invert, double* proj16);
/* CameraParameters::OgreGetProjectionMatrix (src/cameraparameters.cpp:271-295). */
int arucohip_ogre_projection(const float* K, int cam_width, int cam_height, int width, int height, double gnear, double gfar,
                             int invert, double* proj16);

#ifdef __cplusplus
}
#endif
#endif /* ARUCOHIP_H */
