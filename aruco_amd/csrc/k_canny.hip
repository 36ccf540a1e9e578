// The CANNY threshold method: cv::Canny(grey, out, 10, 220) (/root/reference/src/markerdetector.cpp:667-676; OpenCV 3.0
// imgproc/src/canny.cpp: 3x3 Sobel with replicated borders, L1 magnitude with a zero rim, non-maximum suppression along the
// quantised gradient direction, hysteresis). Non-default method of row a2: built for completeness, not tuned.
//
//   canny_nms_kernel   one workgroup (64 lanes) per 8x8-pixel tile: gradients of the tile and its 1-pixel apron into LDS,
//                      suppression per pixel, two ballots -> the tile's survivors (mag > low) and seeds (mag > high) as
//                      uint64 tiles — the representation the contour stage uses for its binary image.
//   canny_hyst_kernel  hysteresis = the survivors that are 8-connected to a seed: a workgroup owns 16x16 tiles (128x128
//                      pixels), grows the edge set to its fixed point inside the block with 64-bit tile logic (3x3
//                      dilation of a tile from its 8 neighbours, AND survivors) and reports whether anything changed;
//                      the host repeats the launch until no block changes (an edge crosses a block per launch at worst).
//   canny_expand_kernel edge tiles -> the 0 / 255 bytes of the thresholded image; the contour tiles are then built from
//                      those bytes by the BINARY pass of k_threshold.hip like for a caller-supplied thresholded image.
#include "bits_tiles.h"
#include "internal.h"

namespace ah {

struct CannyArgs {
    const uint8_t* gray;
    size_t row_stride, frame_stride;
    int width, height;
    int ctx, cty;          // tiles per row / column of the canny tile arrays (no pad tile)
    int low, high;
    uint64_t* surv;        // [P][cty][ctx] survivors of the suppression
    uint64_t* edge;        // [P][cty][ctx] seeds, then edges
    uint32_t* changed;     // one flag
    int nthr, t;
};

__global__ __launch_bounds__(64) void canny_nms_kernel(CannyArgs a) {
    __shared__ int s_dx[10][10], s_dy[10][10], s_mag[10][10];
    const int bx = blockIdx.x, by = blockIdx.y, frame = blockIdx.z, lane = threadIdx.x;
    const int W = a.width, H = a.height;
    const uint8_t* src = a.gray + (size_t)frame * a.frame_stride;
    auto G = [&](int x, int y) -> int { return src[(size_t)min(max(y, 0), H - 1) * a.row_stride + min(max(x, 0), W - 1)]; };
    for (int i = lane; i < 100; i += WAVE) {
        const int ay = i / 10, ax = i - ay * 10;
        const int x = bx * 8 - 1 + ax, y = by * 8 - 1 + ay;
        int gx = 0, gy = 0, m = 0;
        if (x >= 0 && x < W && y >= 0 && y < H) {   // outside the image the magnitude is zero
            gx = (G(x + 1, y - 1) + 2 * G(x + 1, y) + G(x + 1, y + 1)) - (G(x - 1, y - 1) + 2 * G(x - 1, y) + G(x - 1, y + 1));
            gy = (G(x - 1, y + 1) + 2 * G(x, y + 1) + G(x + 1, y + 1)) - (G(x - 1, y - 1) + 2 * G(x, y - 1) + G(x + 1, y - 1));
            m = abs(gx) + abs(gy);
        }
        s_dx[ay][ax] = gx, s_dy[ay][ax] = gy, s_mag[ay][ax] = m;
    }
    __syncthreads();
    const int lx = lane & 7, ly = lane >> 3, x = bx * 8 + lx, y = by * 8 + ly;
    bool keep = false, strong = false;
    if (x < W && y < H) {
        const int cx = lx + 1, cy = ly + 1;
        const int v = s_mag[cy][cx];
        if (v > a.low) {
            const int xs = s_dx[cy][cx], ys = s_dy[cy][cx];
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int TG22 = 13573;   // (int)(0.41421356... * 2^15 + 0.5)
            const int tg22x = ax * TG22;
            if (ay < tg22x) {
                keep = v > s_mag[cy][cx - 1] && v >= s_mag[cy][cx + 1];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x)
                    keep = v > s_mag[cy - 1][cx] && v >= s_mag[cy + 1][cx];
                else {
                    const int s = (xs ^ ys) < 0 ? -1 : 1;
                    keep = v > s_mag[cy - 1][cx - s] && v > s_mag[cy + 1][cx + s];
                }
            }
            strong = keep && v > a.high;
        }
    }
    const unsigned long long bk = __ballot(keep), bs = __ballot(strong);
    if (lane == 0) {
        const size_t ti = ((size_t)(frame * a.nthr + a.t) * a.cty + by) * a.ctx + bx;
        a.surv[ti] = bk, a.edge[ti] = bs;
    }
}

constexpr int HB = 16;   // tiles per side of a hysteresis block

__device__ __forceinline__ uint64_t hdil(uint64_t T, uint64_t L, uint64_t R) {
    const uint64_t COL0 = 0x0101010101010101ull, COL7 = 0x8080808080808080ull;
    return T | ((T << 1) & ~COL0) | ((L >> 7) & COL0) | ((T >> 1) & ~COL7) | ((R << 7) & COL7);
}

__global__ __launch_bounds__(HB* HB) void canny_hyst_kernel(CannyArgs a) {
    __shared__ uint64_t sE[HB + 2][HB + 2];
    const int plane = blockIdx.z, lx = threadIdx.x % HB, ly = threadIdx.x / HB;
    const int tx = blockIdx.x * HB + lx, ty = blockIdx.y * HB + ly;
    const size_t base = (size_t)plane * a.cty * a.ctx;
    auto ld = [&](const uint64_t* arr, int x, int y) -> uint64_t { return (x >= 0 && x < a.ctx && y >= 0 && y < a.cty) ? arr[base + (size_t)y * a.ctx + x] : 0ull; };
    // the block's tiles and a ring of neighbours (the ring stays as it is during this launch)
    for (int i = threadIdx.x; i < (HB + 2) * (HB + 2); i += HB * HB) {
        const int yy = i / (HB + 2), xx = i - yy * (HB + 2);
        sE[yy][xx] = ld(a.edge, blockIdx.x * HB + xx - 1, blockIdx.y * HB + yy - 1);
    }
    const uint64_t C = ld(a.surv, tx, ty);
    __syncthreads();
    const uint64_t E0 = sE[ly + 1][lx + 1];
    uint64_t E = E0;
    int again;
    do {
        const uint64_t Hs = hdil(E, sE[ly + 1][lx], sE[ly + 1][lx + 2]);
        const uint64_t Hu = hdil(sE[ly][lx + 1], sE[ly][lx], sE[ly][lx + 2]);
        const uint64_t Hd = hdil(sE[ly + 2][lx + 1], sE[ly + 2][lx], sE[ly + 2][lx + 2]);
        uint64_t En = E | ((Hs | (Hs << 8) | (Hu >> 56) | (Hs >> 8) | (Hd << 56)) & C);
        for (int it = 0; it < 64; it++) {   // fixed point inside the tile
            const uint64_t h = hdil(En, 0ull, 0ull);
            const uint64_t g = En | ((h | (h << 8) | (h >> 8)) & C);
            if (g == En) break;
            En = g;
        }
        const int ch = En != E;
        __syncthreads();          // everybody has read its neighbours
        E = En;
        sE[ly + 1][lx + 1] = E;
        again = __syncthreads_or(ch);
    } while (again);
    if (E != E0) {
        a.edge[base + (size_t)ty * a.ctx + tx] = E;   // C is zero outside the tile array, so (tx, ty) is inside
        atomicOr(a.changed, 1u);
    }
}

__global__ __launch_bounds__(256) void canny_expand_kernel(const uint64_t* __restrict__ edge, int ctx, int cty, int W, int H, uint8_t* __restrict__ thres) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, plane = blockIdx.z;
    if (x >= W) return;
    const uint64_t t = edge[((size_t)plane * cty + (y >> 3)) * ctx + (x >> 3)];
    thres[((size_t)plane * H + y) * W + x] = ((t >> ((y & 7) * 8 + (x & 7))) & 1ull) ? 255 : 0;
}

// thresholded planes of `nframes` frames into b.thres (plane = frame * nthr + t); blocks the host while the hysteresis converges
int launch_canny(hipStream_t s, const uint8_t* gray, const FrameGeom& g, int nframes, int nthr, const Buffers& b, uint64_t* surv, uint64_t* edge, uint32_t* changed) {
    CannyArgs a;
    a.gray = gray, a.row_stride = g.row_stride, a.frame_stride = g.frame_stride, a.width = g.width, a.height = g.height;
    a.ctx = (g.width + 7) / 8, a.cty = (g.height + 7) / 8, a.low = 10, a.high = 220;
    a.surv = surv, a.edge = edge, a.changed = changed, a.nthr = nthr;
    for (int t = 0; t < nthr; t++) {   // MarkerDetector::thresHold ignores its parameters for CANNY: every plane of a range is the same image
        a.t = t;
        hipLaunchKernelGGL(canny_nms_kernel, dim3(a.ctx, a.cty, nframes), dim3(64), 0, s, a);
    }
    const int planes = nframes * nthr;
    for (int it = 0; it < 4096; it++) {
        if (hipMemsetAsync(changed, 0, sizeof(uint32_t), s) != hipSuccess) return 1;
        hipLaunchKernelGGL(canny_hyst_kernel, dim3((a.ctx + HB - 1) / HB, (a.cty + HB - 1) / HB, planes), dim3(HB * HB), 0, s, a);
        uint32_t ch = 0;
        if (hipMemcpyAsync(&ch, changed, sizeof(ch), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
        if (!ch) break;
    }
    hipLaunchKernelGGL(canny_expand_kernel, dim3((g.width + 255) / 256, g.height, planes), dim3(256), 0, s, edge, a.ctx, a.cty, g.width, g.height, b.thres);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace ah
