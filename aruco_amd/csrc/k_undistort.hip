// Row f3 of SURVEY §8 — lens undistortion of the input frame on the device.
//
// Reference call sites: the GL apps undistort every frame before detect() — cv::undistort(TheInputImage, TheUndInputImage,
// CameraMatrix, Distorsion) at /root/reference/utils/aruco_test_gl.cpp:237-240, utils/aruco_test_board_gl.cpp:265-268 — and then
// detect with an empty distortion vector. cv::undistort (OpenCV 3.0 imgproc/src/undistort.cpp) builds CV_16SC2 fixed-point
// maps stripe by stripe (initUndistortRectifyMap with the principal point shifted by the stripe's first row, the source
// position accumulated along each row in double) and remaps with INTER_LINEAR / BORDER_CONSTANT 0 (5 fractional bits, 15-bit
// weights). The map depends on (size, K, dist) only: it is computed once per camera — one thread per image row, because the
// position of column j is the running sum of j additions in the reference — and cached in the handle; per frame the
// remap kernel is a streaming gather (4 source bytes per channel and pixel, coalesced stores).
#include "internal.h"

namespace ah {

struct UndistArgs {
    int width, height, stripe;   // stripe = rows of one cv::undistort stripe
    double A[9], k[8];
    short2* xy;
    uint16_t* fxy;
};

__global__ __launch_bounds__(64) void undist_map_kernel(UndistArgs a) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= a.height) return;
    const int y0 = (y / a.stripe) * a.stripe, i = y - y0;
    // ir = inverse of the new camera matrix with cy - y0 (cv::invert's 3x3 closed form: determinant and adjugate)
    double S[9], ir[9];
    for (int q = 0; q < 9; q++) S[q] = a.A[q];
    S[5] = a.A[5] - y0;
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d != 0.) {
        d = 1. / d;
        ir[0] = (S[4] * S[8] - S[5] * S[7]) * d;
        ir[1] = (S[2] * S[7] - S[1] * S[8]) * d;
        ir[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        ir[3] = (S[5] * S[6] - S[3] * S[8]) * d;
        ir[4] = (S[0] * S[8] - S[2] * S[6]) * d;
        ir[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        ir[6] = (S[3] * S[7] - S[4] * S[6]) * d;
        ir[7] = (S[1] * S[6] - S[0] * S[7]) * d;
        ir[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    } else {
        for (int q = 0; q < 9; q++) ir[q] = 0;
    }
    const double u0 = a.A[2], v0 = a.A[5], fx = a.A[0], fy = a.A[4];
    const double k1 = a.k[0], k2 = a.k[1], p1 = a.k[2], p2 = a.k[3], k3 = a.k[4], k4 = a.k[5], k5 = a.k[6], k6 = a.k[7];
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    short2* xy = a.xy + (size_t)y * a.width;
    uint16_t* fxy = a.fxy + (size_t)y * a.width;
    for (int j = 0; j < a.width; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double ww = 1. / _w, x = _x * ww, yy = _y * ww;
        const double x2 = x * x, y2 = yy * yy;
        const double r2 = x2 + y2, _2xy = 2 * x * yy;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
        const double v = fy * (yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
        // saturate_cast<int>(double) = round half to even, saturated
        const int iu = __double2int_rn(fmax(-2147483648.0, fmin(2147483647.0, u * 32)));
        const int iv = __double2int_rn(fmax(-2147483648.0, fmin(2147483647.0, v * 32)));
        xy[j] = make_short2((short)(iu >> 5), (short)(iv >> 5));
        fxy[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
    }
}

struct RemapArgs {
    const uint8_t* src;
    size_t row_stride, frame_stride;
    int width, height, cn;
    const short2* xy;
    const uint16_t* fxy;
    uint8_t* dst;   // tightly packed [frame][H][W][cn]
};

__global__ __launch_bounds__(256) void remap_linear_kernel(RemapArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= a.width) return;
    const size_t pix = (size_t)y * a.width + x;
    const short2 s = a.xy[pix];
    const int fq = a.fxy[pix], fx = fq & 31, fy = fq >> 5;
    const int sx = s.x, sy = s.y, W = a.width, H = a.height, cn = a.cn;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const uint8_t* src = a.src + (size_t)f * a.frame_stride;
    uint8_t* d = a.dst + ((size_t)f * W * H + pix) * cn;
    const bool outside = sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0;
    const bool x0in = sx >= 0 && sx < W, x1in = sx + 1 >= 0 && sx + 1 < W, y0in = sy >= 0 && sy < H, y1in = sy + 1 >= 0 && sy + 1 < H;
    for (int c = 0; c < cn; c++) {
        int v = 0;
        if (!outside) {
            const int p00 = (x0in && y0in) ? src[(size_t)sy * a.row_stride + (size_t)sx * cn + c] : 0;
            const int p01 = (x1in && y0in) ? src[(size_t)sy * a.row_stride + (size_t)(sx + 1) * cn + c] : 0;
            const int p10 = (x0in && y1in) ? src[(size_t)(sy + 1) * a.row_stride + (size_t)sx * cn + c] : 0;
            const int p11 = (x1in && y1in) ? src[(size_t)(sy + 1) * a.row_stride + (size_t)(sx + 1) * cn + c] : 0;
            v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        }
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

void launch_undist_map(hipStream_t s, int W, int H, const float* K, const float* dist, int ndist, short2* xy, uint16_t* fxy) {
    UndistArgs a;
    a.width = W, a.height = H;
    a.stripe = min(max(1, (1 << 12) / max(W, 1)), H);
    for (int i = 0; i < 9; i++) a.A[i] = (double)K[i];
    for (int i = 0; i < 8; i++) a.k[i] = (dist && i < ndist) ? (double)dist[i] : 0.0;
    a.xy = xy, a.fxy = fxy;
    hipLaunchKernelGGL(undist_map_kernel, dim3((H + 63) / 64), dim3(64), 0, s, a);
}

void launch_remap(hipStream_t s, const uint8_t* src, size_t row_stride, size_t frame_stride, int W, int H, int cn, int nframes, const short2* xy,
                  const uint16_t* fxy, uint8_t* dst) {
    RemapArgs a{src, row_stride, frame_stride, W, H, cn, xy, fxy, dst};
    hipLaunchKernelGGL(remap_linear_kernel, dim3((W + 255) / 256, H, nframes), dim3(256), 0, s, a);
}

}  // namespace ah
