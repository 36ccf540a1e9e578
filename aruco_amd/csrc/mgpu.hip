// Frame sharding over the GPUs of one node behind the C ABI: frames are independent units, frame f goes
// to device slot f mod G, there is no exchange on the data path. The only collective step is the gather of the per-frame
// marker blocks {int32 n, arucohip_marker_t[cap]}: every device writes its block either straight to pinned host memory or
// — ARUCOHIP_MGPU_GATHER_PEER — over xGMI into one buffer on the first device (hipMemcpyPeerAsync, device to device), from
// where a single copy brings all blocks to the host. The reference has no multi-GPU code; the caller shape it serves is the
// frame loop of /root/reference/utils/aruco_test.cpp:140-160 with G detectors behind one call.
//
// Round 3: one PERSISTENT host thread per device slot (created with the detector, not per call) and an asynchronous form,
// arucohip_mgpu_submit_* / arucohip_mgpu_wait: every slot's handle runs `depth` batches in flight (arucohip_set_pipeline_depth),
// a worker submits the next ticket's sub-batch before it waits for the oldest, so each device sees the same pipelined stream of
// batches as a single-GPU caller of arucohip_detect_batch_submit. The synchronous calls are submit + wait.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arucohip.h"

namespace {

struct Job {
    int ticket = -1;
    const uint8_t* base = nullptr;   // slot's first frame (host or device)
    int count = 0;
    int on_device = 0;
    int W = 0, H = 0;
    size_t row_stride = 0, frame_stride = 0;
    bool has_K = false, has_dist = false;
    float K[9] = {}, dist[8] = {};
    int ndist = 0;
    float marker_size = -1;
    int y_perp = 0;
    int lane = 0;                    // staging set (ticket mod depth)
};

struct Slot {
    int device = 0;
    arucohip_handle* h = nullptr;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> queue;
    bool stop = false;
    // per lane: results of the sub-batch. Host gather: the handle writes them to h_out / h_n. Peer gather: the handle leaves them in
    // d_out / d_n on this device and the worker sends them to the first device's g_out / g_n.
    std::vector<arucohip_marker_t*> h_out, d_out;
    std::vector<int32_t*> h_n, d_n;
    // completion of tickets (guarded by the detector's mutex)
    std::vector<int> done_rc;        // per lane: rc of the sub-batch of the ticket that lane last carried
    std::vector<int> done_ticket;    // per lane: which ticket that was (-1 none)
    std::vector<std::string> done_msg;
};

}  // namespace

struct arucohip_mgpu {
    std::vector<Slot*> slots;
    int per_device = 0;          // frames one device takes per call
    int cap = 0;                 // marker slots per frame in the gather blocks
    int flags = 0;
    int depth = 1;
    int max_w = 0, max_h = 0;
    arucohip_params_t params;
    // peer gather: [lane][G][per_device][cap] on the first device, pinned host copies of it
    std::vector<arucohip_marker_t*> g_out, hg_out;
    std::vector<int32_t*> g_n, hg_n;
    // tickets
    struct Pending {
        bool active = false;
        int ticket = -1, kind = 0;   // kind 0: frame f -> slot f mod G (host frames), 1: one camera stream per slot
        std::vector<int> counts;
        arucohip_marker_t* out = nullptr;
        int cap = 0;
        int32_t* n_out = nullptr;
    };
    std::vector<Pending> pend;       // per lane
    int next_ticket = 0;
    bool broken = false;             // a rebuild of the lanes failed and the previous depth could not be restored: every call reports it
    bool peer_fallback = false;      // ARUCOHIP_MGPU_GATHER_PEER was asked for but a device cannot reach the first one: host gather in use
    std::mutex mu;                   // completion records of all slots
    std::condition_variable cv;
    std::string err;
};

namespace {

int mg_fail(arucohip_mgpu* m, int code, const std::string& msg) {
    if (m) m->err = msg;
    return code;
}

#define MGCHK(m, expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return mg_fail(m, ARUCOHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// the worker of one device slot: submits a job as soon as a lane of its handle is free, waits for the oldest otherwise
void worker_main(arucohip_mgpu* m, int g) {
    Slot* s = m->slots[g];
    (void)hipSetDevice(s->device);
    const bool peer = (m->flags & ARUCOHIP_MGPU_GATHER_PEER) != 0;
    const size_t blk = (size_t)m->per_device * m->cap;
    struct Flight {
        Job job;
        int hticket;
        int rc;
    };
    std::deque<Flight> flight;
    auto finish_oldest = [&]() {
        Flight f = flight.front();
        flight.pop_front();
        int rc = f.rc;
        std::string msg;
        if (rc == ARUCOHIP_OK) {
            rc = arucohip_detect_batch_wait(s->h, f.hticket);   // host outputs are filled / device outputs are complete
            if (rc != ARUCOHIP_OK) msg = arucohip_last_error_string(s->h);
        } else {
            msg = arucohip_last_error_string(s->h);
        }
        if (peer && (rc == ARUCOHIP_OK || rc == ARUCOHIP_E_OVERFLOW)) {
            // the block travels device to device (xGMI) into the first device's gather buffer of this lane
            hipStream_t st = (hipStream_t)arucohip_get_stream(s->h);
            const int l = f.job.lane;
            hipError_t e = hipMemcpyPeerAsync(m->g_out[l] + (size_t)g * blk, m->slots[0]->device, s->d_out[l], s->device,
                                              (size_t)f.job.count * m->cap * sizeof(arucohip_marker_t), st);
            if (e == hipSuccess)
                e = hipMemcpyPeerAsync(m->g_n[l] + (size_t)g * m->per_device, m->slots[0]->device, s->d_n[l], s->device, (size_t)f.job.count * sizeof(int32_t), st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) rc = ARUCOHIP_E_HIP, msg = hipGetErrorString(e);
        }
        {
            std::lock_guard<std::mutex> lk(m->mu);
            s->done_rc[f.job.lane] = rc, s->done_ticket[f.job.lane] = f.job.ticket, s->done_msg[f.job.lane] = msg;
        }
        m->cv.notify_all();
    };
    for (;;) {
        Job job;
        bool have = false;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            if (flight.empty())
                s->cv.wait(lk, [&] { return s->stop || !s->queue.empty(); });
            if (!s->queue.empty() && (int)flight.size() < m->depth) {
                job = s->queue.front();
                s->queue.pop_front();
                have = true;
            } else if (s->stop && s->queue.empty() && flight.empty()) {
                return;
            }
        }
        if (have) {
            Flight f;
            f.job = job, f.hticket = -1;
            const int l = job.lane;
            if (peer)
                f.rc = arucohip_detect_batch_submit(s->h, job.base, job.count, job.W, job.H, job.row_stride, job.frame_stride, job.on_device, job.has_K ? job.K : nullptr,
                                                    job.has_dist ? job.dist : nullptr, job.ndist, job.marker_size, job.y_perp, s->d_out[l], m->cap, s->d_n[l], 1, &f.hticket);
            else
                f.rc = arucohip_detect_batch_submit(s->h, job.base, job.count, job.W, job.H, job.row_stride, job.frame_stride, job.on_device, job.has_K ? job.K : nullptr,
                                                    job.has_dist ? job.dist : nullptr, job.ndist, job.marker_size, job.y_perp, s->h_out[l], m->cap, s->h_n[l], 0, &f.hticket);
            flight.push_back(f);
        } else if (!flight.empty()) {
            finish_oldest();   // nothing to submit (or the pipeline is full): the oldest batch completes
        }
    }
}

void stop_workers(arucohip_mgpu* m) {
    for (Slot* s : m->slots) {
        if (!s->worker.joinable()) continue;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->stop = true;
        }
        s->cv.notify_all();
        s->worker.join();
    }
}

void free_staging(arucohip_mgpu* m) {
    for (Slot* s : m->slots) {
        (void)hipSetDevice(s->device);
        for (auto* p : s->h_out) if (p) (void)hipHostFree(p);
        for (auto* p : s->h_n) if (p) (void)hipHostFree(p);
        for (auto* p : s->d_out) if (p) (void)hipFree(p);
        for (auto* p : s->d_n) if (p) (void)hipFree(p);
        s->h_out.clear(), s->h_n.clear(), s->d_out.clear(), s->d_n.clear();
    }
    if (!m->slots.empty()) {
        (void)hipSetDevice(m->slots[0]->device);
        for (auto* p : m->g_out) if (p) (void)hipFree(p);
        for (auto* p : m->g_n) if (p) (void)hipFree(p);
        for (auto* p : m->hg_out) if (p) (void)hipHostFree(p);
        for (auto* p : m->hg_n) if (p) (void)hipHostFree(p);
    }
    m->g_out.clear(), m->g_n.clear(), m->hg_out.clear(), m->hg_n.clear();
}

// Fault injection for the tests of the two recovery paths below (tests/test_gpu_boundary.py): ARUCOHIP_MGPU_INJECT = "fail_depth:3,1"
// makes every build of 3 or 1 lanes fail behind its first slot, "nopeer" makes hipDeviceCanAccessPeer read as false. Read per call.
bool inject_fail_depth(int depth) {
    const char* e = getenv("ARUCOHIP_MGPU_INJECT");
    const char* q = e ? strstr(e, "fail_depth:") : nullptr;
    if (!q) return false;
    for (q += 11; *q;) {
        if (atoi(q) == depth) return true;
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
    }
    return false;
}
bool inject_nopeer() {
    const char* e = getenv("ARUCOHIP_MGPU_INJECT");
    return e && strstr(e, "nopeer");
}

// can every device slot write into the first device's memory (xGMI peer access)? Slots on the first device itself need nothing.
bool peers_reachable(arucohip_mgpu* m) {
    if (inject_nopeer()) return false;
    for (size_t g = 1; g < m->slots.size(); g++) {
        if (m->slots[g]->device == m->slots[0]->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->slots[g]->device, m->slots[0]->device) != hipSuccess || !can) {
            (void)hipGetLastError();
            return false;
        }
    }
    return true;
}

// one attempt to (re)build the lanes: handles' pipelines, staging per lane, workers. On failure nothing runs (the workers start last) but the
// staging may be partly allocated: build_lanes cleans up.
int build_lanes_once(arucohip_mgpu* m, int depth) {
    stop_workers(m);
    free_staging(m);
    const int G = (int)m->slots.size();
    const size_t blk = (size_t)m->per_device * m->cap;
    // the peer gather needs every device to reach the first one; where one cannot, the blocks go through pinned host memory instead
    if ((m->flags & ARUCOHIP_MGPU_GATHER_PEER) && !peers_reachable(m)) m->flags &= ~ARUCOHIP_MGPU_GATHER_PEER, m->peer_fallback = true;
    const bool peer = (m->flags & ARUCOHIP_MGPU_GATHER_PEER) != 0;
    m->depth = depth;
    m->pend.assign(depth, arucohip_mgpu::Pending());
    int nslot = 0;
    for (Slot* s : m->slots) {
        MGCHK(m, hipSetDevice(s->device));
        if (nslot++ == 1 % G && inject_fail_depth(depth)) return mg_fail(m, ARUCOHIP_E_HIP, "injected failure while building the lanes");
        int rc = arucohip_set_pipeline_depth(s->h, depth);
        if (rc) return mg_fail(m, rc, arucohip_last_error_string(s->h));
        s->h_out.assign(depth, nullptr), s->h_n.assign(depth, nullptr), s->d_out.assign(depth, nullptr), s->d_n.assign(depth, nullptr);
        s->done_rc.assign(depth, ARUCOHIP_OK), s->done_ticket.assign(depth, -1), s->done_msg.assign(depth, std::string());
        for (int l = 0; l < depth; l++) {
            if (peer) {
                MGCHK(m, hipMalloc((void**)&s->d_out[l], blk * sizeof(arucohip_marker_t)));
                MGCHK(m, hipMalloc((void**)&s->d_n[l], (size_t)m->per_device * sizeof(int32_t)));
            } else {
                MGCHK(m, hipHostMalloc((void**)&s->h_out[l], blk * sizeof(arucohip_marker_t)));
                MGCHK(m, hipHostMalloc((void**)&s->h_n[l], (size_t)m->per_device * sizeof(int32_t)));
            }
        }
    }
    if (peer) {
        MGCHK(m, hipSetDevice(m->slots[0]->device));
        m->g_out.assign(depth, nullptr), m->g_n.assign(depth, nullptr), m->hg_out.assign(depth, nullptr), m->hg_n.assign(depth, nullptr);
        for (int l = 0; l < depth; l++) {
            MGCHK(m, hipMalloc((void**)&m->g_out[l], (size_t)G * blk * sizeof(arucohip_marker_t)));
            MGCHK(m, hipMalloc((void**)&m->g_n[l], (size_t)G * m->per_device * sizeof(int32_t)));
            MGCHK(m, hipHostMalloc((void**)&m->hg_out[l], (size_t)G * blk * sizeof(arucohip_marker_t)));
            MGCHK(m, hipHostMalloc((void**)&m->hg_n[l], (size_t)G * m->per_device * sizeof(int32_t)));
        }
        // the writing device maps the first device's memory so that the copy travels over xGMI (peers_reachable has checked that it can;
        // "already enabled" is not an error)
        for (int g = 1; g < G; g++) {
            if (m->slots[g]->device == m->slots[0]->device) continue;
            if (hipSetDevice(m->slots[g]->device) == hipSuccess) (void)hipDeviceEnablePeerAccess(m->slots[0]->device, 0);
            (void)hipGetLastError();
        }
    }
    for (int g = 0; g < G; g++) {
        m->slots[g]->stop = false;
        m->slots[g]->worker = std::thread(worker_main, m, g);
    }
    return ARUCOHIP_OK;
}

// Transactional: either the detector runs at `depth` afterwards, or — when that build fails, e.g. out of device memory for the lanes — at the
// depth it had before, and the call reports the failure. Only when the previous depth cannot be restored either is the detector marked
// broken: every later submit / wait / detect then returns ARUCOHIP_E_HIP at once instead of queueing work that no thread would take.
int build_lanes(arucohip_mgpu* m, int depth, int previous) {
    int rc = build_lanes_once(m, depth);
    if (rc == ARUCOHIP_OK) {
        m->broken = false;
        return rc;
    }
    const std::string why = m->err;
    if (previous > 0 && previous != depth && build_lanes_once(m, previous) == ARUCOHIP_OK) {
        m->broken = false;
        m->err = why + " (the previous depth " + std::to_string(previous) + " was restored)";
        return rc;
    }
    stop_workers(m);
    free_staging(m);
    for (Slot* s : m->slots) {
        (void)hipSetDevice(s->device);
        (void)arucohip_set_pipeline_depth(s->h, 0);
    }
    m->depth = 1, m->pend.assign(1, arucohip_mgpu::Pending());
    m->broken = true;
    m->err = why + " (no lanes could be built: the detector is unusable, destroy it)";
    return rc;
}

}  // namespace

extern "C" {

int arucohip_mgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void arucohip_mgpu_destroy(arucohip_mgpu* m) {
    if (!m) return;
    stop_workers(m);
    free_staging(m);
    for (Slot* s : m->slots) {
        (void)hipSetDevice(s->device);
        if (s->h) arucohip_destroy(s->h);
        delete s;
    }
    delete m;
}

int arucohip_mgpu_create(const arucohip_params_t* params, const int* devices, int ndevices, int max_width, int max_height,
                         int max_frames_per_device, int cap, int flags, arucohip_mgpu** out) {
    if (!out || ndevices < 1 || ndevices > 64 || max_frames_per_device < 1 || cap < 1) return ARUCOHIP_E_INVALID;
    *out = nullptr;
    const int have = arucohip_mgpu_device_count();
    arucohip_mgpu* m = new arucohip_mgpu();
    m->per_device = max_frames_per_device, m->cap = cap, m->flags = flags, m->max_w = max_width, m->max_h = max_height;
    if (params)
        m->params = *params;
    else
        arucohip_default_params(&m->params);
    for (int g = 0; g < ndevices; g++) {
        const int dev = devices ? devices[g] : g;   // the same device may be listed more than once (separate handles)
        if (dev < 0 || dev >= have) {
            arucohip_mgpu_destroy(m);
            return ARUCOHIP_E_INVALID;
        }
        Slot* s = new Slot();
        s->device = dev;
        m->slots.push_back(s);
    }
    for (Slot* s : m->slots) {
        int rc = arucohip_create(params, s->device, max_width, max_height, max_frames_per_device, &s->h);
        if (rc != ARUCOHIP_OK) {
            arucohip_mgpu_destroy(m);
            return rc;
        }
    }
    int rc = build_lanes(m, 1, 0);
    if (rc != ARUCOHIP_OK) {
        arucohip_mgpu_destroy(m);
        return rc;
    }
    *out = m;
    return ARUCOHIP_OK;
}

int arucohip_mgpu_size(const arucohip_mgpu* m) { return m ? (int)m->slots.size() : 0; }
arucohip_handle* arucohip_mgpu_handle(arucohip_mgpu* m, int slot) { return (m && slot >= 0 && slot < (int)m->slots.size()) ? m->slots[slot]->h : nullptr; }
const char* arucohip_mgpu_last_error_string(const arucohip_mgpu* m) { return m ? m->err.c_str() : "null multi-GPU detector"; }

int arucohip_mgpu_set_params(arucohip_mgpu* m, const arucohip_params_t* p) {
    if (!m || !p) return ARUCOHIP_E_INVALID;
    for (auto& pd : m->pend)
        if (pd.active) return mg_fail(m, ARUCOHIP_E_INVALID, "a submitted batch has not been waited for");
    for (Slot* s : m->slots) {
        int rc = arucohip_set_params(s->h, p);
        if (rc) return mg_fail(m, rc, arucohip_last_error_string(s->h));
    }
    m->params = *p;
    return ARUCOHIP_OK;
}

int arucohip_mgpu_set_depth(arucohip_mgpu* m, int depth) {
    if (!m || depth < 1 || depth > 8) return ARUCOHIP_E_INVALID;
    for (auto& pd : m->pend)
        if (pd.active) return mg_fail(m, ARUCOHIP_E_INVALID, "a submitted batch has not been waited for");
    if (depth == m->depth && !m->broken) return ARUCOHIP_OK;
    return build_lanes(m, depth, m->broken ? 0 : m->depth);
}

int arucohip_mgpu_gather_mode(const arucohip_mgpu* m) { return (m && (m->flags & ARUCOHIP_MGPU_GATHER_PEER)) ? ARUCOHIP_MGPU_GATHER_PEER : ARUCOHIP_MGPU_GATHER_HOST; }

}  // extern "C"

namespace {

int submit_common(arucohip_mgpu* m, int kind, const std::vector<const uint8_t*>& bases, const std::vector<int>& counts, int on_device, int W, int H, size_t row_stride,
                  size_t fstride, const float* K, const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out, int cap, int32_t* n_out,
                  int* ticket) {
    if (ndist < 0 || ndist > 8) return mg_fail(m, ARUCOHIP_E_INVALID, "ndist must be 0..8");
    if (m->broken) return mg_fail(m, ARUCOHIP_E_HIP, "the detector has no lanes (an earlier arucohip_mgpu_set_depth failed and could not be rolled back)");
    const int lane = m->next_ticket % m->depth;
    if (m->pend[lane].active) return mg_fail(m, ARUCOHIP_E_CAPACITY, "pipeline full: wait for the oldest ticket first");
    auto& pd = m->pend[lane];
    pd.active = true, pd.ticket = m->next_ticket, pd.kind = kind, pd.counts = counts, pd.out = out, pd.cap = cap, pd.n_out = n_out;
    const int G = (int)m->slots.size();
    for (int g = 0; g < G; g++) {
        Slot* s = m->slots[g];
        if (counts[g] <= 0) {   // nothing for this slot: complete at once
            std::lock_guard<std::mutex> lk(m->mu);
            s->done_rc[lane] = ARUCOHIP_OK, s->done_ticket[lane] = pd.ticket, s->done_msg[lane].clear();
            continue;
        }
        Job j;
        j.ticket = pd.ticket, j.base = bases[g], j.count = counts[g], j.on_device = on_device, j.W = W, j.H = H, j.row_stride = row_stride, j.frame_stride = fstride;
        j.has_K = K != nullptr, j.has_dist = dist != nullptr && ndist > 0, j.ndist = j.has_dist ? ndist : 0;
        if (K) std::memcpy(j.K, K, sizeof(j.K));
        if (j.has_dist) std::memcpy(j.dist, dist, (size_t)ndist * sizeof(float));
        j.marker_size = marker_size, j.y_perp = y_perp, j.lane = lane;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->queue.push_back(j);
        }
        s->cv.notify_all();
    }
    *ticket = m->next_ticket++;
    return ARUCOHIP_OK;
}

}  // namespace

extern "C" {

int arucohip_mgpu_submit_batch(arucohip_mgpu* m, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride, const float* K,
                               const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out, int cap, int32_t* n_out, int* ticket) {
    if (!m || !frames || !out || !n_out || !ticket || nframes < 1 || cap < 1) return ARUCOHIP_E_INVALID;
    const int G = (int)m->slots.size();
    if (nframes > G * m->per_device) return mg_fail(m, ARUCOHIP_E_INVALID, "more frames than devices x frames per device");
    // frame f -> slot f mod G: a slot's frames are a strided batch of the caller's array
    std::vector<const uint8_t*> bases(G);
    std::vector<int> counts(G);
    for (int g = 0; g < G; g++) bases[g] = frames + (size_t)g * frame_stride, counts[g] = g < nframes ? (nframes - g + G - 1) / G : 0;
    return submit_common(m, 0, bases, counts, 0, W, H, row_stride, (size_t)G * frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out, ticket);
}

int arucohip_mgpu_submit_streams(arucohip_mgpu* m, const uint8_t* const* frames_dev, const int* nframes, int W, int H, size_t row_stride, size_t frame_stride,
                                 const float* K, const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out, int cap, int32_t* n_out,
                                 int* ticket) {
    if (!m || !frames_dev || !nframes || !out || !n_out || !ticket || cap < 1) return ARUCOHIP_E_INVALID;
    const int G = (int)m->slots.size();
    std::vector<const uint8_t*> bases(G);
    std::vector<int> counts(G);
    for (int g = 0; g < G; g++) {
        if (nframes[g] < 0 || nframes[g] > m->per_device || (nframes[g] > 0 && !frames_dev[g])) return mg_fail(m, ARUCOHIP_E_INVALID, "bad per-device frame count");
        bases[g] = frames_dev[g], counts[g] = nframes[g];
    }
    return submit_common(m, 1, bases, counts, 1, W, H, row_stride, frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out, ticket);
}

int arucohip_mgpu_wait(arucohip_mgpu* m, int ticket) {
    if (!m || ticket < 0) return ARUCOHIP_E_INVALID;
    if (m->broken) return mg_fail(m, ARUCOHIP_E_HIP, "the detector has no lanes (an earlier arucohip_mgpu_set_depth failed and could not be rolled back)");
    const int lane = ticket % m->depth;
    auto& pd = m->pend[lane];
    if (!pd.active || pd.ticket != ticket) return mg_fail(m, ARUCOHIP_E_INVALID, "no such batch in flight");
    const int G = (int)m->slots.size();
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv.wait(lk, [&] {
            for (int g = 0; g < G; g++)
                if (m->slots[g]->done_ticket[lane] != ticket) return false;
            return true;
        });
    }
    pd.active = false;
    int ret = ARUCOHIP_OK;
    for (int g = 0; g < G; g++) {
        const int rc = m->slots[g]->done_rc[lane];
        // a list overflow leaves the other frames' results valid (n = -1 marks the overflowed ones): keep collecting, report it at the end
        if (rc != ARUCOHIP_OK && rc != ARUCOHIP_E_OVERFLOW) return mg_fail(m, rc, std::string("device slot ") + std::to_string(g) + ": " + m->slots[g]->done_msg[lane]);
        if (rc == ARUCOHIP_E_OVERFLOW && ret == ARUCOHIP_OK) ret = mg_fail(m, rc, std::string("device slot ") + std::to_string(g) + ": " + m->slots[g]->done_msg[lane]);
    }
    const bool peer = (m->flags & ARUCOHIP_MGPU_GATHER_PEER) != 0;
    const size_t blk = (size_t)m->per_device * m->cap;
    if (peer) {   // one copy brings every slot's block from the first device to the host
        MGCHK(m, hipSetDevice(m->slots[0]->device));
        MGCHK(m, hipMemcpy(m->hg_out[lane], m->g_out[lane], (size_t)G * blk * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost));
        MGCHK(m, hipMemcpy(m->hg_n[lane], m->g_n[lane], (size_t)G * m->per_device * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    const int mcap = m->cap, cap = pd.cap;
    for (int g = 0; g < G; g++) {
        const arucohip_marker_t* src = peer ? m->hg_out[lane] + (size_t)g * blk : m->slots[g]->h_out[lane];
        const int32_t* sn = peer ? m->hg_n[lane] + (size_t)g * m->per_device : m->slots[g]->h_n[lane];
        for (int j = 0; j < pd.counts[g]; j++) {
            const size_t f = pd.kind == 0 ? (size_t)j * G + g : (size_t)g * m->per_device + j;
            int n = sn[j];
            pd.n_out[f] = n;
            if (n > std::min(cap, mcap)) {
                if (ret == ARUCOHIP_OK) ret = mg_fail(m, ARUCOHIP_E_CAPACITY, "marker output array too small");
                n = std::min(cap, mcap);
            }
            if (n > 0) std::memcpy(pd.out + f * cap, src + (size_t)j * mcap, (size_t)n * sizeof(arucohip_marker_t));
        }
    }
    return ret;
}

int arucohip_mgpu_detect_batch(arucohip_mgpu* m, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride, const float* K,
                               const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out, int cap, int32_t* n_out) {
    int t = -1;
    int rc = arucohip_mgpu_submit_batch(m, frames, nframes, W, H, row_stride, frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out, &t);
    if (rc) return rc;
    return arucohip_mgpu_wait(m, t);
}

int arucohip_mgpu_detect_streams(arucohip_mgpu* m, const uint8_t* const* frames_dev, const int* nframes, int W, int H, size_t row_stride,
                                 size_t frame_stride, const float* K, const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out,
                                 int cap, int32_t* n_out) {
    int t = -1;
    int rc = arucohip_mgpu_submit_streams(m, frames_dev, nframes, W, H, row_stride, frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out, &t);
    if (rc) return rc;
    return arucohip_mgpu_wait(m, t);
}

}  // extern "C"
