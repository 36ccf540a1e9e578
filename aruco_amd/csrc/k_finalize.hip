// Kernels 7-8 — per-frame marker list (sort by id, same-id dedupe, border filter) and batched pose.
//
// Reference: MarkerDetector::detect tail (/root/reference/src/markerdetector.cpp:371-382 split valid / rejected,
// :417 std::sort by id, :421-430 same-id dedupe by perimeter, :433-447 border filter, :450-467 per-marker solvePnP),
// perimeter (src/utils.h:39-46), getObjectPoints (src/marker.cpp:91-108), rotateXAxis (src/utils.cpp:16-30),
// BoardDetector pose (src/boarddetector.cpp:157-198).
#include "internal.h"
#include "pnp_device.h"

namespace ah {

constexpr int MAXM = 512;   // = the largest candidates_per_frame arucohip_create_ex accepts

__device__ __forceinline__ float perimeter_f(const float* c) {
    float sum = 0;
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        float dx = c[2 * i] - c[2 * j], dy = c[2 * i + 1] - c[2 * j + 1];
        sum = (float)((double)sum + sqrt((double)dx * dx + (double)dy * dy));
    }
    return sum;
}

struct FinalArgs {
    const Cand* cands;
    const int32_t* ncands;
    arucohip_marker_t* markers;
    int32_t* nmarkers;
    uint32_t* counters;
    uint32_t* marker_list;
    const uint32_t* trig_cnt;   // per-plane counter lines: TC_STATUS = overflow bits of the plane
    int nthr;
    int cap_cands, cap_markers;
    int bx0, by0, bx1, by1;
    // write-through (round 3): results that go to the caller's device memory and need no pose are stored there by this kernel as well,
    // instead of two copies queued behind it (each a dispatch that waits its turn on a busy chip: 0.1 ms apiece with five batches in flight)
    arucohip_marker_t* out;   // [frames][out_cap] or null
    int32_t* n_out;           // [frames] or null
    int out_cap;
    int write_hdr;            // one frame per call: count and status word also go to the slot behind the marker array, so ONE copy brings everything to the host
};

__global__ __launch_bounds__(64) void finalize_kernel(FinalArgs a) {
    latency_bound_priority();
    __shared__ int s_src[MAXM];      // candidate index, sorted by (id, candidate order)
    __shared__ int s_id[MAXM];
    __shared__ uint8_t s_rem[MAXM];
    __shared__ int s_nvalid;
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int nc = a.ncands[frame];
    const Cand* C = a.cands + (size_t)frame * a.cap_cands;
    // stable rank by id among decoded candidates
    if (lane == 0) s_nvalid = 0;
    __syncthreads();
    for (int i = lane; i < nc; i += WAVE) {
        int id = C[i].id;
        if (id < 0) continue;
        int rank = 0;
        for (int j = 0; j < nc; j++) {
            int idj = C[j].id;
            if (idj < 0) continue;
            rank += (idj < id) || (idj == id && j < i);
        }
        if (rank < MAXM) s_src[rank] = i, s_id[rank] = id, s_rem[rank] = 0;
        atomicAdd(&s_nvalid, 1);
    }
    __syncthreads();
    const int nv = min(s_nvalid, MAXM);
    if (lane == 0 && s_nvalid > MAXM) atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_MARKER_OVERFLOW);
    if (lane == 0) {
        for (int i = 0; i < nv - 1; i++) {
            if (s_id[i] == s_id[i + 1] && !s_rem[i + 1]) {
                if (perimeter_f(C[s_src[i]].c) > perimeter_f(C[s_src[i + 1]].c))
                    s_rem[i + 1] = 1;
                else
                    s_rem[i] = 1;
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < nv; i += WAVE) {  // corners outside Rect(size*t, size*(1-t)) -> drop
        const float* c = C[s_src[i]].c;
        for (int k = 0; k < 4; k++) {
            int px = __float2int_rn(c[2 * k]), py = __float2int_rn(c[2 * k + 1]);
            if (!(a.bx0 <= px && px < a.bx1 && a.by0 <= py && py < a.by1)) s_rem[i] = 1;
        }
    }
    __syncthreads();
    if (lane == 0) {
        int n = 0;
        arucohip_marker_t* M = a.markers + (size_t)frame * a.cap_markers;
        for (int i = 0; i < nv; i++) {
            if (s_rem[i]) continue;
            if (n < a.cap_markers) {
                arucohip_marker_t m;
                m.id = s_id[i];
                for (int k = 0; k < 8; k++) m.corners[k] = C[s_src[i]].c[k];
                m.ssize = -1.f, m.has_pose = 0, m.pad_ = 0;
                for (int k = 0; k < 3; k++) m.rvec[k] = m.tvec[k] = 0;
                M[n] = m;
                if (a.out && n < a.out_cap) a.out[(size_t)frame * a.out_cap + n] = m;
            }
            n++;
        }
        if (n > a.cap_markers) atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_MARKER_OVERFLOW);
        // a device list overflowed while one of this frame's planes was worked on: its result may be incomplete and is given up (n = -1);
        // every other frame of the batch is unaffected (the lists are per plane / per frame, or name the frame that did not fit)
        uint32_t fst = 0;
        for (int t = 0; t < a.nthr; t++) fst |= a.trig_cnt[(size_t)(frame * a.nthr + t) * TRIG_CNT_STRIDE + TC_STATUS];
        if (fst) n = -1;
        a.nmarkers[frame] = n;   // required count; the host clamps and reports ARUCOHIP_E_CAPACITY
        if (a.write_hdr && frame == 0) {
            int32_t* hdr = (int32_t*)(a.markers + (size_t)gridDim.x * a.cap_markers);
            hdr[0] = n, hdr[1] = (int32_t)atomicOr(&a.counters[CNT_STATUS], 0u);   // every earlier kernel's overflow bits and this one's
        }
        if (a.n_out) a.n_out[frame] = n;
        // work list of the pose kernel (order across frames is irrelevant); never more than F * cap_markers entries
        const int kept = min(n, a.cap_markers);
        if (kept > 0) {
            const uint32_t base = atomicAdd(&a.counters[CNT_NMARK], (uint32_t)kept);
            for (int i = 0; i < kept; i++) a.marker_list[base + i] = ((uint32_t)frame << 16) | (uint32_t)i;
        }
    }
}

void launch_finalize(hipStream_t s, const FrameGeom& g, int nframes, const DetectParams& p, const CamModel& cam, const Buffers& b, arucohip_marker_t* out, int out_cap,
                     int32_t* n_out) {
    FinalArgs a;
    a.out = out, a.out_cap = out_cap, a.n_out = n_out;
    a.write_hdr = nframes == 1;
    a.cands = b.cands, a.ncands = b.ncands, a.markers = b.markers, a.nmarkers = b.nmarkers, a.counters = b.counters, a.marker_list = b.marker_list;
    a.trig_cnt = b.trig_cnt, a.nthr = p.nthr;
    a.cap_cands = b.cap_cands, a.cap_markers = b.cap_markers;
    a.bx0 = p.bx0, a.by0 = p.by0, a.bx1 = p.bx1, a.by1 = p.by1;
    hipLaunchKernelGGL(finalize_kernel, dim3(nframes), dim3(64), 0, s, a);
}

// Per-marker pose (markerdetector.cpp:450-467, Marker::calculateExtrinsics marker.cpp:112-124): solvePnP of the 4 corners
// against the marker's own square. Four lanes share a marker — one corner each for the homography sums, the Jacobian rows
// and the reprojection errors, butterfly sums inside the group of four, the small solves redundantly on every lane — and
// the markers come from the flat list finalize_kernel wrote, so a wave carries 16 live markers instead of the few slots of
// one frame that happen to be filled (round 1: one lane per marker slot, 2.0 ms per 20 k markers).
constexpr int POSE_G = 4;

__device__ inline void marker_pose4(arucohip_marker_t* m, const CamModel& cam, float* s_obj, float* s_img, int sub) {
    const float hs = (float)((double)cam.marker_size / 2.);
    // getObjectPoints (marker.cpp:91-108): (-,-), (-,+), (+,+), (+,-)
    s_obj[3 * sub] = (sub < 2) ? -hs : hs, s_obj[3 * sub + 1] = (sub == 1 || sub == 2) ? hs : -hs, s_obj[3 * sub + 2] = 0.f;
    s_img[2 * sub] = m->corners[2 * sub], s_img[2 * sub + 1] = m->corners[2 * sub + 1];
    double r[3], t[3];
    const bool ok = solve_pnp_planar_wave<POSE_G>(s_obj, s_img, 4, cam, r, t, sub);
    if (ok && cam.y_perp) rotate_x_axis(r);
    if (sub == 0) {
        for (int k = 0; k < 3; k++) m->rvec[k] = ok ? r[k] : 0, m->tvec[k] = ok ? t[k] : 0;
        m->has_pose = ok ? 1 : 0;
        m->ssize = cam.marker_size;
    }
}

// list != nullptr: the batch's flat marker list; else markers[0 .. n_direct)
__global__ __launch_bounds__(64) void pose_kernel(arucohip_marker_t* markers, const uint32_t* list, const uint32_t* counters, uint32_t cap_list,
                                                  int cap_markers, int n_direct, CamModel cam) {
    latency_bound_priority();
    __shared__ float s_obj[16][12], s_img[16][8];
    const int grp = threadIdx.x / POSE_G, sub = threadIdx.x % POSE_G;
    const uint32_t gid = blockIdx.x * (64 / POSE_G) + grp;
    const uint32_t n = list ? min(counters[CNT_NMARK], cap_list) : (uint32_t)n_direct;
    if (gid >= n) return;   // uniform within the group of four
    arucohip_marker_t* m = markers + gid;
    if (list) {
        const uint32_t e = list[gid];
        m = markers + (size_t)(e >> 16) * cap_markers + (e & 0xFFFFu);
    }
    marker_pose4(m, cam, s_obj[grp], s_img[grp], sub);
}

void launch_pose(hipStream_t s, int nframes, const CamModel& cam, const Buffers& b) {
    // the list's length is only known on the device: the grid covers its capacity, surplus workgroups exit at once
    const uint32_t cap_list = (uint32_t)nframes * (uint32_t)b.cap_markers;
    hipLaunchKernelGGL(pose_kernel, dim3((cap_list + 15) / 16), dim3(64), 0, s, b.markers, b.marker_list, b.counters, cap_list, b.cap_markers, 0, cam);
}

void launch_marker_pose(hipStream_t s, arucohip_marker_t* markers, int n, const CamModel& cam) {
    hipLaunchKernelGGL(pose_kernel, dim3((n + 15) / 16), dim3(64), 0, s, markers, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0, n, cam);
}

// generic planar PnP over npts correspondences (board pose of one arucohip_board_detect call): one wavefront, the lanes share the
// points (wave sums for the homography and J^T J) like board_pose_kernel. A single lane took 1.7 ms for the 96 points of the
// reference's board still (ArucoPerf.Board), more than the rest of the call.
__global__ __launch_bounds__(64) void pnp_points_kernel(const float* obj, const float* img, int npts, CamModel cam, double* rt, int* ok_out) {
    if (blockIdx.x != 0) return;
    double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    const bool ok = solve_pnp_planar_wave<64>(obj, img, npts, cam, r, t, (int)threadIdx.x);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) rt[k] = r[k], rt[3 + k] = t[k];
        *ok_out = ok ? 1 : 0;
    }
}

void launch_pnp_points(hipStream_t s, const float* obj, const float* img, int npts, const CamModel& cam, double* rt_out, int* ok_out) {
    hipLaunchKernelGGL(pnp_points_kernel, dim3(1), dim3(64), 0, s, obj, img, npts, cam, rt_out, ok_out);
}

// cv::projectPoints(obj, rvec, tvec, K, dist) -> float image points (BoardDetector reprojection filter)
__global__ void project_points_kernel(const float* obj, int npts, const double* rt, CamModel cam, float* img) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    double R[9];
    rodrigues_vec2mat(rt, R, nullptr);
    double mx, my;
    project_point(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], R, nullptr, rt + 3, cam.K, cam.k, &mx, &my, nullptr, nullptr);
    img[2 * i] = (float)mx, img[2 * i + 1] = (float)my;
}

void launch_project_points(hipStream_t s, const float* obj, int npts, const double* rt, const CamModel& cam, float* img_out) {
    hipLaunchKernelGGL(project_points_kernel, dim3((npts + 63) / 64), dim3(64), 0, s, obj, npts, rt, cam, img_out);
}

// ---------------------------------------------------------------------------------------------
// Batched BoardDetector::detect (boarddetector.cpp:90-205): one wavefront per frame on the device-resident markers of the
// last batch. Lane 0 filters the markers by board id and lays out the 3-D / 2-D correspondences in LDS, then the 64 lanes
// share the points of the planar solvePnP (wave sums for J^T J); optional reprojection filter and second solve.
// ---------------------------------------------------------------------------------------------
constexpr int MAX_BOARD_POINTS = 512;

struct BoardArgs {
    const arucohip_marker_t* markers;
    const int32_t* nmarkers;
    int cap_markers;
    const int32_t* ids;
    const float* obj;      // nboard * 4 * 3
    int nboard, info_type;
    float marker_size, repj_thres;
    CamModel cam;
    arucohip_board_t* out;
    float* prob;
    uint32_t* counters;
};

__global__ __launch_bounds__(64) void board_pose_kernel(BoardArgs a) {
    latency_bound_priority();
    __shared__ float s_obj[MAX_BOARD_POINTS * 3], s_img[MAX_BOARD_POINTS * 2], s_obj2[MAX_BOARD_POINTS * 3], s_img2[MAX_BOARD_POINTS * 2];
    __shared__ int s_npts, s_nmark, s_n2;
    const int frame = blockIdx.x, lane = threadIdx.x;
    const arucohip_marker_t* M = a.markers + (size_t)frame * a.cap_markers;
    const int nm = min(a.nmarkers[frame], a.cap_markers);
    if (lane == 0) {
        const float dx = a.obj[0] - a.obj[3], dy = a.obj[1] - a.obj[4], dz = a.obj[2] - a.obj[5];
        const double side = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
        const double mpp = a.info_type == ARUCOHIP_BOARD_PIX ? (double)a.marker_size / side : 1.0;
        int np = 0, nk = 0;
        for (int i = 0; i < nm; i++) {
            int slot = -1;
            for (int j = 0; j < a.nboard; j++)
                if (a.ids[j] == M[i].id) {
                    slot = j;
                    break;
                }
            if (slot < 0) continue;
            nk++;
            if (np + 4 > MAX_BOARD_POINTS) {   // more correspondences than the kernel holds: reported, never silent
                atomicOr(&a.counters[CNT_STATUS], (uint32_t)ST_MARKER_OVERFLOW);
                continue;
            }
            for (int p = 0; p < 4; p++, np++) {
                s_img[2 * np] = M[i].corners[2 * p], s_img[2 * np + 1] = M[i].corners[2 * p + 1];
                const float* q = a.obj + ((size_t)slot * 4 + p) * 3;
                for (int c = 0; c < 3; c++) s_obj[3 * np + c] = (float)(q[c] * mpp);
            }
        }
        s_npts = np, s_nmark = nk;
    }
    __syncthreads();
    const int np = s_npts, nk = s_nmark;
    arucohip_board_t res;
    res.n_markers = nk, res.has_pose = 0;
    for (int k = 0; k < 3; k++) res.rvec[k] = res.tvec[k] = 0;
    float prob = 0;
    const bool enough = (a.marker_size > 0 && a.info_type == ARUCOHIP_BOARD_PIX) || a.info_type == ARUCOHIP_BOARD_METERS;
    if (nk > 0 && a.cam.has_K && enough) {
        double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
        bool ok = solve_pnp_planar_wave<64>(s_obj, s_img, np, a.cam, r, t, lane);
        if (a.repj_thres > 0 && ok) {
            double R[9];
            rodrigues_vec2mat(r, R, nullptr);
            if (lane == 0) s_n2 = 0;
            __syncthreads();
            for (int base = 0; base < np; base += 64) {   // order-preserving compaction of the points that pass
                const int i = base + lane;
                bool keep = false;
                if (i < np) {
                    double mx, my;
                    project_point(s_obj[3 * i], s_obj[3 * i + 1], s_obj[3 * i + 2], R, nullptr, t, a.cam.K, a.cam.k, &mx, &my, nullptr, nullptr);
                    const float ex = (float)mx - s_img[2 * i], ey = (float)my - s_img[2 * i + 1];
                    keep = (float)sqrt((double)ex * ex + (double)ey * ey) < a.repj_thres;
                }
                const unsigned long long bal = __ballot(keep);
                const int dst = s_n2 + __popcll(bal & ((1ull << lane) - 1ull));
                if (keep) {
                    for (int c = 0; c < 3; c++) s_obj2[3 * dst + c] = s_obj[3 * i + c];
                    s_img2[2 * dst] = s_img[2 * i], s_img2[2 * dst + 1] = s_img[2 * i + 1];
                }
                __syncthreads();
                if (lane == 0) s_n2 += __popcll(bal);
                __syncthreads();
            }
            // fewer than 4 surviving points: the reference's second solvePnP would throw; keep the first pose, flag no pose
            ok = s_n2 >= 4 && solve_pnp_planar_wave<64>(s_obj2, s_img2, s_n2, a.cam, r, t, lane);
        }
        if (ok && a.cam.y_perp) rotate_x_axis(r);
        res.has_pose = ok ? 1 : 0;
        for (int k = 0; k < 3; k++) res.rvec[k] = r[k], res.tvec[k] = t[k];
        prob = (float)nk / (float)a.nboard;
    }
    if (lane == 0) a.out[frame] = res, a.prob[frame] = prob;
}

void launch_board_pose(hipStream_t s, int nframes, const Buffers& b, const int32_t* ids, const float* obj, int nboard, int info_type,
                       float marker_size, float repj_thres, const CamModel& cam, arucohip_board_t* out, float* prob) {
    BoardArgs a;
    a.markers = b.markers, a.nmarkers = b.nmarkers, a.cap_markers = b.cap_markers;
    a.ids = ids, a.obj = obj, a.nboard = nboard, a.info_type = info_type, a.marker_size = marker_size, a.repj_thres = repj_thres;
    a.cam = cam, a.out = out, a.prob = prob, a.counters = b.counters;
    hipLaunchKernelGGL(board_pose_kernel, dim3(nframes), dim3(64), 0, s, a);
}

// GetGLModelViewMatrix (src/utils.cpp:32-69) for every marker of the batch, one lane per marker slot: column-major 4x4
// from the marker's rvec / tvec (third row negated: OpenGL looks down -z); markers without a pose give a zero matrix.
__global__ void gl_modelview_kernel(const arucohip_marker_t* markers, const int32_t* nmarkers, int cap_markers, int nframes, int cap, double* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = gid / cap, i = gid - frame * cap;
    if (frame >= nframes) return;
    double* m = out + (size_t)gid * 16;
    const bool live = i < min(nmarkers[frame], cap_markers) && markers[(size_t)frame * cap_markers + i].has_pose;
    if (!live) {
        for (int k = 0; k < 16; k++) m[k] = 0.0;
        return;
    }
    const arucohip_marker_t& mk = markers[(size_t)frame * cap_markers + i];
    double R[9];
    rodrigues_vec2mat(mk.rvec, R, nullptr);
    for (int col = 0; col < 3; col++) {
        m[0 + col * 4] = R[0 * 3 + col];
        m[1 + col * 4] = R[1 * 3 + col];
        m[2 + col * 4] = -R[2 * 3 + col];
        m[3 + col * 4] = 0.0;
    }
    m[12] = mk.tvec[0], m[13] = mk.tvec[1], m[14] = -mk.tvec[2], m[15] = 1.0;
}
void launch_gl_modelview(hipStream_t s, int nframes, int cap, const Buffers& b, double* out_dev) {
    const int total = nframes * cap;
    hipLaunchKernelGGL(gl_modelview_kernel, dim3((total + 255) / 256), dim3(256), 0, s, b.markers, b.nmarkers, b.cap_markers, nframes, cap, out_dev);
}

// rotateXAxis on a pose stored as rt[0..2]
__global__ void rotate_x_kernel(double* rt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) rotate_x_axis(rt);
}
void launch_rotate_x(hipStream_t s, double* rt) { hipLaunchKernelGGL(rotate_x_kernel, dim3(1), dim3(64), 0, s, rt); }

}  // namespace ah
