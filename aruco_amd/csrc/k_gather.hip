// Compaction of a batch's fixed-capacity marker blocks for the gather step of frame sharding.
//
// A batch leaves {int32 n; arucohip_marker_t m[cap]} per frame (arucohip_detect_batch with out_on_device): at the bench's config 2 that is
// 1024 x 64 x 96 B = 6.3 MB per rank and step of which a third is used. The gather sends one packed block instead:
//
//   int32 total        markers in the block (sum of the clipped per-frame counts)
//   int32 nframes
//   int32 cap_total    marker slots the block was packed for
//   int32 overflow     != 0: total > cap_total, the markers beyond cap_total are missing (the counts are still all there)
//   int32 counts[nframes]   per-frame counts as the batch left them (-1 = the frame overflowed a device list), padded to 16 bytes
//   arucohip_marker_t m[cap_total]   the frames' markers back to back, frame f at offset sum over j < f of clip(counts[j])
//
// clip(n) = min(max(n, 0), cap). The receiver rebuilds the per-frame arrays from counts alone (aruco_amd/dist.py::unpack_block).
// The reference has no counterpart (single process, /root/reference/src/markerdetector.cpp:302); the marker layout is aruco::Marker
// (src/marker.h:46-53) as arucohip_marker_t.
#include <hip/hip_runtime.h>

#include "../../include/arucohip.h"

namespace {

// one wave per frame: the wave sums the clipped counts of the frames before its own (at most nframes / 64 rounds of a wave reduction)
// and copies its frame's markers as 16-byte pieces
__global__ __launch_bounds__(64) void compact_markers_kernel(const arucohip_marker_t* __restrict__ blocks, const int32_t* __restrict__ counts, int nframes, int cap,
                                                             int32_t* __restrict__ dst, int cap_total, size_t marker_off_bytes) {
    const int f = blockIdx.x, lane = threadIdx.x;
    int before = 0;
    for (int j = lane; j < f; j += 64) before += min(max(counts[j], 0), cap);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    const int raw = counts[f];
    const int n = min(max(raw, 0), cap);
    if (lane == 0) dst[4 + f] = raw;
    const int fits = max(0, min(n, cap_total - before));
    const uint4* src = (const uint4*)(blocks + (size_t)f * cap);
    uint4* out = (uint4*)((char*)dst + marker_off_bytes) + (size_t)before * 6;   // 96 bytes = six 16-byte pieces
    for (int i = lane; i < fits * 6; i += 64) out[i] = src[i];
    if (f == nframes - 1 && lane == 0) {
        dst[0] = before + n, dst[1] = nframes, dst[2] = cap_total, dst[3] = (before + n > cap_total) ? 1 : 0;
    }
}

}  // namespace

extern "C" {

size_t arucohip_compact_bytes(int nframes, int cap_total) {
    if (nframes < 0 || cap_total < 0) return 0;
    const size_t head = (16 + (size_t)nframes * 4 + 15) & ~(size_t)15;
    return head + (size_t)cap_total * sizeof(arucohip_marker_t);
}

int arucohip_compact_markers(const arucohip_marker_t* blocks_dev, const int32_t* counts_dev, int nframes, int cap, void* dst_dev, int cap_total,
                             void* hip_stream) {
    if (!blocks_dev || !counts_dev || !dst_dev || nframes < 1 || cap < 1 || cap_total < 0) return ARUCOHIP_E_INVALID;
    const size_t head = (16 + (size_t)nframes * 4 + 15) & ~(size_t)15;
    hipLaunchKernelGGL(compact_markers_kernel, dim3(nframes), dim3(64), 0, (hipStream_t)hip_stream, blocks_dev, counts_dev, nframes, cap, (int32_t*)dst_dev,
                       cap_total, head);
    return hipGetLastError() == hipSuccess ? ARUCOHIP_OK : ARUCOHIP_E_HIP;
}

}  // extern "C"
