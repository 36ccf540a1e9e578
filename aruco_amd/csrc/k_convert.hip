// Row f3 of SURVEY §8 — input side: 8-bit BGR frames converted to gray on the device.
//
// Reference: MarkerDetector::detect converts 3-channel input with cv::cvtColor(input, grey, CV_BGR2GRAY)
// (/root/reference/src/markerdetector.cpp:307-310). For 8-bit data that is the fixed-point form
//   gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14
// (coefficients 0.114, 0.587, 0.299 in 14 bits), restated in oracle/orc_imgproc.cpp and used by tests/golden/make_fixtures.py.
// Streaming kernel: a lane turns 4 pixels (three dwords in, one dword out), so a wave reads 768 and writes 256 contiguous bytes.
#include "internal.h"

namespace ah {

__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) { return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14; }

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t* __restrict__ bgr, size_t row_stride, size_t frame_stride, int width, int height,
                                                       uint8_t* __restrict__ gray, int fast) {
    const int frame = blockIdx.z, y = blockIdx.y;
    const uint8_t* src = bgr + (size_t)frame * frame_stride + (size_t)y * row_stride;
    uint8_t* dst = gray + ((size_t)frame * height + y) * width;
    const int quads = (width + 3) / 4;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
        const int x = 4 * q;
        if (fast && x + 3 < width) {   // rows and frames start on dword boundaries, width is a multiple of 4
            const uint32_t* p = (const uint32_t*)(src + 3 * x);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            const uint32_t g0 = gray_of(w0 & 0xFFu, (w0 >> 8) & 0xFFu, (w0 >> 16) & 0xFFu);
            const uint32_t g1 = gray_of(w0 >> 24, w1 & 0xFFu, (w1 >> 8) & 0xFFu);
            const uint32_t g2 = gray_of((w1 >> 16) & 0xFFu, w1 >> 24, w2 & 0xFFu);
            const uint32_t g3 = gray_of((w2 >> 8) & 0xFFu, (w2 >> 16) & 0xFFu, w2 >> 24);
            *(uint32_t*)(dst + x) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else {
            for (int j = 0; j < 4 && x + j < width; j++) {
                const uint8_t* p = src + 3 * (x + j);
                dst[x + j] = (uint8_t)gray_of(p[0], p[1], p[2]);
            }
        }
    }
}

void launch_bgr2gray(hipStream_t s, const uint8_t* bgr, size_t row_stride, size_t frame_stride, int width, int height, int nframes, uint8_t* gray) {
    const int fast = ((width & 3) == 0) && ((row_stride & 3) == 0) && ((frame_stride & 3) == 0) && (((uintptr_t)bgr & 3) == 0) && (((uintptr_t)gray & 3) == 0);
    const int quads = (width + 3) / 4;
    dim3 grid((quads + 255) / 256, height, nframes);
    hipLaunchKernelGGL(bgr2gray_kernel, grid, dim3(256), 0, s, bgr, row_stride, frame_stride, width, height, gray, fast);
}

}  // namespace ah
