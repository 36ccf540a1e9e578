// Frame sharding over the GPUs of one node behind the C ABI: frames are independent units, frame f goes
// to device slot f mod G, there is no exchange on the data path. The only collective step is the gather of the per-frame
// marker blocks {int32 n, arucohip_marker_t[cap]}: every device writes its block either straight to pinned host memory or
// — ARUCOHIP_MGPU_GATHER_PEER — over xGMI into one buffer on the first device (hipMemcpyPeerAsync, device to device), from
// where a single copy brings all blocks to the host. One host thread per device keeps the devices' copies and launches
// concurrent. The reference has no multi-GPU code; the caller shape it serves is the frame loop of
// /root/reference/utils/aruco_test.cpp:140-160 with G detectors behind one call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arucohip.h"

struct arucohip_mgpu {
    std::vector<int> devices;
    std::vector<arucohip_handle*> handles;
    int per_device = 0;          // frames one device takes per call
    int cap = 0;                 // marker slots per frame in the gather blocks
    int flags = 0;
    // per device: results in its own HBM, pinned host staging (host gather) or a slice of the first device's buffer (peer gather)
    std::vector<arucohip_marker_t*> d_out;
    std::vector<int32_t*> d_n;
    std::vector<arucohip_marker_t*> h_out;
    std::vector<int32_t*> h_n;
    arucohip_marker_t* g_out = nullptr;   // [G][per_device][cap] on devices[0] (peer gather)
    int32_t* g_n = nullptr;
    std::vector<hipEvent_t> done;
    std::string err;
};

static int mg_fail(arucohip_mgpu* m, int code, const std::string& msg) {
    if (m) m->err = msg;
    return code;
}

#define MGCHK(m, expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return mg_fail(m, ARUCOHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int arucohip_mgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void arucohip_mgpu_destroy(arucohip_mgpu* m) {
    if (!m) return;
    for (size_t g = 0; g < m->handles.size(); g++) {
        (void)hipSetDevice(m->devices[g]);
        if (g < m->d_out.size() && m->d_out[g]) (void)hipFree(m->d_out[g]);
        if (g < m->d_n.size() && m->d_n[g]) (void)hipFree(m->d_n[g]);
        if (g < m->h_out.size() && m->h_out[g]) (void)hipHostFree(m->h_out[g]);
        if (g < m->h_n.size() && m->h_n[g]) (void)hipHostFree(m->h_n[g]);
        if (g < m->done.size() && m->done[g]) (void)hipEventDestroy(m->done[g]);
        arucohip_destroy(m->handles[g]);
    }
    if (!m->devices.empty()) {
        (void)hipSetDevice(m->devices[0]);
        if (m->g_out) (void)hipFree(m->g_out);
        if (m->g_n) (void)hipFree(m->g_n);
    }
    delete m;
}

int arucohip_mgpu_create(const arucohip_params_t* params, const int* devices, int ndevices, int max_width, int max_height,
                         int max_frames_per_device, int cap, int flags, arucohip_mgpu** out) {
    if (!out || ndevices < 1 || ndevices > 64 || max_frames_per_device < 1 || cap < 1) return ARUCOHIP_E_INVALID;
    *out = nullptr;
    const int have = arucohip_mgpu_device_count();
    arucohip_mgpu* m = new arucohip_mgpu();
    m->per_device = max_frames_per_device, m->cap = cap, m->flags = flags;
    for (int g = 0; g < ndevices; g++) {
        const int dev = devices ? devices[g] : g;   // the same device may be listed more than once (separate handles)
        if (dev < 0 || dev >= have) {
            arucohip_mgpu_destroy(m);
            return ARUCOHIP_E_INVALID;
        }
        m->devices.push_back(dev);
    }
    const size_t blk = (size_t)max_frames_per_device * cap;
    m->d_out.assign(ndevices, nullptr), m->d_n.assign(ndevices, nullptr), m->h_out.assign(ndevices, nullptr), m->h_n.assign(ndevices, nullptr);
    m->done.assign(ndevices, nullptr);
    for (int g = 0; g < ndevices; g++) {
        arucohip_handle* h = nullptr;
        int rc = arucohip_create(params, m->devices[g], max_width, max_height, max_frames_per_device, &h);
        if (rc != ARUCOHIP_OK) {
            arucohip_mgpu_destroy(m);
            return rc;
        }
        m->handles.push_back(h);
        bool ok = hipSetDevice(m->devices[g]) == hipSuccess && hipMalloc((void**)&m->d_out[g], blk * sizeof(arucohip_marker_t)) == hipSuccess &&
                  hipMalloc((void**)&m->d_n[g], (size_t)max_frames_per_device * sizeof(int32_t)) == hipSuccess &&
                  hipEventCreateWithFlags(&m->done[g], hipEventDisableTiming) == hipSuccess;
        if (ok && !(flags & ARUCOHIP_MGPU_GATHER_PEER))
            ok = hipHostMalloc((void**)&m->h_out[g], blk * sizeof(arucohip_marker_t)) == hipSuccess &&
                 hipHostMalloc((void**)&m->h_n[g], (size_t)max_frames_per_device * sizeof(int32_t)) == hipSuccess;
        if (!ok) {
            arucohip_mgpu_destroy(m);
            return ARUCOHIP_E_HIP;
        }
    }
    if (flags & ARUCOHIP_MGPU_GATHER_PEER) {
        bool ok = hipSetDevice(m->devices[0]) == hipSuccess && hipMalloc((void**)&m->g_out, (size_t)ndevices * blk * sizeof(arucohip_marker_t)) == hipSuccess &&
                  hipMalloc((void**)&m->g_n, (size_t)ndevices * max_frames_per_device * sizeof(int32_t)) == hipSuccess &&
                  hipHostMalloc((void**)&m->h_out[0], (size_t)ndevices * blk * sizeof(arucohip_marker_t)) == hipSuccess &&
                  hipHostMalloc((void**)&m->h_n[0], (size_t)ndevices * max_frames_per_device * sizeof(int32_t)) == hipSuccess;
        // the writing device needs direct access to the first device's memory for the copy to travel over xGMI; without it
        // hipMemcpyPeerAsync still works (staged by the runtime), so this is best effort
        for (int g = 1; ok && g < ndevices; g++) {
            if (m->devices[g] == m->devices[0]) continue;
            int can = 0;
            if (hipSetDevice(m->devices[g]) == hipSuccess && hipDeviceCanAccessPeer(&can, m->devices[g], m->devices[0]) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(m->devices[0], 0);
            (void)hipGetLastError();
        }
        if (!ok) {
            arucohip_mgpu_destroy(m);
            return ARUCOHIP_E_HIP;
        }
    }
    *out = m;
    return ARUCOHIP_OK;
}

int arucohip_mgpu_size(const arucohip_mgpu* m) { return m ? (int)m->handles.size() : 0; }
arucohip_handle* arucohip_mgpu_handle(arucohip_mgpu* m, int slot) { return (m && slot >= 0 && slot < (int)m->handles.size()) ? m->handles[slot] : nullptr; }
const char* arucohip_mgpu_last_error_string(const arucohip_mgpu* m) { return m ? m->err.c_str() : "null multi-GPU detector"; }

int arucohip_mgpu_set_params(arucohip_mgpu* m, const arucohip_params_t* p) {
    if (!m || !p) return ARUCOHIP_E_INVALID;
    for (auto* h : m->handles) {
        int rc = arucohip_set_params(h, p);
        if (rc) return mg_fail(m, rc, arucohip_last_error_string(h));
    }
    return ARUCOHIP_OK;
}

// common part: slot g detects `counts[g]` frames starting at bases[g] (stride `fstride`), all slots concurrently, then the gather.
// place(g, j) = index of slot g's j-th frame in the caller's out / n_out arrays.
static int run_sharded(arucohip_mgpu* m, const std::vector<const uint8_t*>& bases, const std::vector<int>& counts, int frames_on_device, int W, int H,
                       size_t row_stride, size_t fstride, const float* K, const float* dist, int ndist, float marker_size, int y_perp,
                       arucohip_marker_t* out, int cap, int32_t* n_out, const std::function<size_t(int, int)>& place) {
    const int G = (int)m->handles.size();
    const int mcap = m->cap;
    const size_t blk = (size_t)m->per_device * mcap;
    const bool peer = (m->flags & ARUCOHIP_MGPU_GATHER_PEER) != 0;
    std::vector<int> rcs(G, ARUCOHIP_OK);
    auto work = [&](int g) {
        if (counts[g] <= 0) return;
        arucohip_handle* h = m->handles[g];
        if (hipSetDevice(m->devices[g]) != hipSuccess) {
            rcs[g] = ARUCOHIP_E_HIP;
            return;
        }
        int rc = arucohip_detect_batch(h, bases[g], counts[g], W, H, row_stride, fstride, frames_on_device, K, dist, ndist, marker_size, y_perp, m->d_out[g],
                                       mcap, m->d_n[g], 1);
        if (rc == ARUCOHIP_OK) rc = arucohip_batch_status(h);   // waits for the slot's stream; overflow conditions surface here
        hipStream_t s = (hipStream_t)arucohip_get_stream(h);
        if (rc == ARUCOHIP_OK) {
            hipError_t e;
            if (peer) {   // the block travels device to device (xGMI) into the first device's gather buffer
                e = hipMemcpyPeerAsync(m->g_out + (size_t)g * blk, m->devices[0], m->d_out[g], m->devices[g], (size_t)counts[g] * mcap * sizeof(arucohip_marker_t), s);
                if (e == hipSuccess)
                    e = hipMemcpyPeerAsync(m->g_n + (size_t)g * m->per_device, m->devices[0], m->d_n[g], m->devices[g], (size_t)counts[g] * sizeof(int32_t), s);
            } else {
                e = hipMemcpyAsync(m->h_out[g], m->d_out[g], (size_t)counts[g] * mcap * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipMemcpyAsync(m->h_n[g], m->d_n[g], (size_t)counts[g] * sizeof(int32_t), hipMemcpyDeviceToHost, s);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) rc = ARUCOHIP_E_HIP;
        }
        rcs[g] = rc;
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; g++) th.emplace_back(work, g);
    work(0);
    for (auto& t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[g]) return mg_fail(m, rcs[g], std::string("device slot ") + std::to_string(g) + ": " + arucohip_last_error_string(m->handles[g]));
    if (peer) {   // one copy brings every slot's block from the first device to the host
        MGCHK(m, hipSetDevice(m->devices[0]));
        MGCHK(m, hipMemcpy(m->h_out[0], m->g_out, (size_t)G * blk * sizeof(arucohip_marker_t), hipMemcpyDeviceToHost));
        MGCHK(m, hipMemcpy(m->h_n[0], m->g_n, (size_t)G * m->per_device * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    int ret = ARUCOHIP_OK;
    for (int g = 0; g < G; g++) {
        const arucohip_marker_t* src = peer ? m->h_out[0] + (size_t)g * blk : m->h_out[g];
        const int32_t* sn = peer ? m->h_n[0] + (size_t)g * m->per_device : m->h_n[g];
        for (int j = 0; j < counts[g]; j++) {
            const size_t f = place(g, j);
            int n = sn[j];
            n_out[f] = n;
            if (n > std::min(cap, mcap)) {
                if (ret == ARUCOHIP_OK) ret = mg_fail(m, ARUCOHIP_E_CAPACITY, "marker output array too small");
                n = std::min(cap, mcap);
            }
            if (n > 0) std::memcpy(out + f * cap, src + (size_t)j * mcap, (size_t)n * sizeof(arucohip_marker_t));
        }
    }
    return ret;
}

int arucohip_mgpu_detect_batch(arucohip_mgpu* m, const uint8_t* frames, int nframes, int W, int H, size_t row_stride, size_t frame_stride, const float* K,
                               const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out, int cap, int32_t* n_out) {
    if (!m || !frames || !out || !n_out || nframes < 1 || cap < 1) return ARUCOHIP_E_INVALID;
    const int G = (int)m->handles.size();
    if (nframes > G * m->per_device) return mg_fail(m, ARUCOHIP_E_INVALID, "more frames than devices x frames per device");
    // frame f -> slot f mod G: a slot's frames are a strided batch of the caller's array
    std::vector<const uint8_t*> bases(G);
    std::vector<int> counts(G);
    for (int g = 0; g < G; g++) bases[g] = frames + (size_t)g * frame_stride, counts[g] = g < nframes ? (nframes - g + G - 1) / G : 0;
    return run_sharded(m, bases, counts, 0, W, H, row_stride, (size_t)G * frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out,
                       [G](int g, int j) { return (size_t)j * G + g; });
}

int arucohip_mgpu_detect_streams(arucohip_mgpu* m, const uint8_t* const* frames_dev, const int* nframes, int W, int H, size_t row_stride,
                                 size_t frame_stride, const float* K, const float* dist, int ndist, float marker_size, int y_perp, arucohip_marker_t* out,
                                 int cap, int32_t* n_out) {
    if (!m || !frames_dev || !nframes || !out || !n_out || cap < 1) return ARUCOHIP_E_INVALID;
    const int G = (int)m->handles.size();
    std::vector<const uint8_t*> bases(G);
    std::vector<int> counts(G);
    for (int g = 0; g < G; g++) {
        if (nframes[g] < 0 || nframes[g] > m->per_device || (nframes[g] > 0 && !frames_dev[g])) return mg_fail(m, ARUCOHIP_E_INVALID, "bad per-device frame count");
        bases[g] = frames_dev[g], counts[g] = nframes[g];
    }
    const int per = m->per_device;
    return run_sharded(m, bases, counts, 1, W, H, row_stride, frame_stride, K, dist, ndist, marker_size, y_perp, out, cap, n_out,
                       [per](int g, int j) { return (size_t)g * per + j; });
}

}  // extern "C"
