// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h). Restatements added in round 2:
//   * cv::undistort (8-bit, 1 or 3 channels) as the reference's GL apps call it before detect()
//     (/root/reference/utils/aruco_test_gl.cpp:237-240, utils/aruco_test_board_gl.cpp:265-268): OpenCV 3.0
//     imgproc/src/undistort.cpp (cv::undistort -> initUndistortRectifyMap per stripe, CV_16SC2 fixed-point maps) and
//     imgwarp.cpp (remap INTER_LINEAR, BORDER_CONSTANT 0, 5 fractional bits, 15-bit weights). OpenCV is not vendored in
//     /root/reference and the reference holds no fixture for this call: PARITY UNPINNED at the OpenCV level; pinned
//     independently by tests/test_oracle_crosschecks.py (float bilinear remap within 1 grey level).
//   * findCornerMaxima (/root/reference/src/markerdetector.cpp:157-199), the pre-pass of the locked-corner method
//     (:291-295, :398-399): cv::cornerHarris(block 3, aperture 3, k 0.04) on the window around every corner, 4x4 block sums
//     through cv::integral, L1-weighted arg max. The reference holds no fixture: PARITY UNPINNED, cross-checked the same way.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "orc.h"

namespace orc {

static inline int cv_round(double v) { return (int)std::lrint(v); }

// 3x3 inverse as cv::invert takes it for small matrices (determinant and adjugate, double)
static bool invert3(const double* S, double* t) {
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.) return false;
    d = 1. / d;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    return true;
}

// maps of one stripe (rows y0 .. y0+rows-1): integer source position and the 5+5 fractional bits
void undistort_maps(int w, int y0, int rows, const double A[9], const double k[8], int16_t* xy, uint16_t* fxy) {
    double Ar[9], ir[9];
    for (int i = 0; i < 9; i++) Ar[i] = A[i];
    Ar[5] = A[5] - y0;                       // undistort.cpp: Ar(1, 2) = v0 - y
    if (!invert3(Ar, ir)) std::memset(ir, 0, sizeof(ir));
    const double u0 = A[2], v0 = A[5], fx = A[0], fy = A[4];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    for (int i = 0; i < rows; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double ww = 1. / _w, x = _x * ww, y = _y * ww;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
            const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
            const int iu = cv_round(u * 32), iv = cv_round(v * 32);
            xy[((size_t)i * w + j) * 2] = (int16_t)(iu >> 5);
            xy[((size_t)i * w + j) * 2 + 1] = (int16_t)(iv >> 5);
            fxy[(size_t)i * w + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

// remap INTER_LINEAR / BORDER_CONSTANT(0) of one pixel, cn interleaved channels
static inline void remap_pixel(const uint8_t* src, int w, int h, size_t stride, int cn, int sx, int sy, int fxy, uint8_t* d) {
    const int fx = fxy & 31, fy = fxy >> 5;
    // 15-bit weights (1 - fx)(1 - fy), fx (1 - fy), (1 - fx) fy, fx fy of the 1/32 steps: exact integers
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    for (int c = 0; c < cn; c++) {
        auto px = [&](int x, int y) -> int { return (x >= 0 && x < w && y >= 0 && y < h) ? src[(size_t)y * stride + (size_t)x * cn + c] : 0; };
        int v;
        if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0)
            v = 0;
        else
            v = (px(sx, sy) * w00 + px(sx + 1, sy) * w01 + px(sx, sy + 1) * w10 + px(sx + 1, sy + 1) * w11 + (1 << 14)) >> 15;
        d[c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

void undistort_8u(const uint8_t* src, int w, int h, size_t stride, int cn, const float K[9], const float* dist, int ndist, uint8_t* dst) {
    double A[9], k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 9; i++) A[i] = (double)K[i];
    for (int i = 0; i < ndist && i < 8; i++) k[i] = (double)dist[i];
    const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(w, 1)), h);
    std::vector<int16_t> xy((size_t)stripe0 * w * 2);
    std::vector<uint16_t> fxy((size_t)stripe0 * w);
    for (int y = 0; y < h; y += stripe0) {
        const int rows = std::min(stripe0, h - y);
        undistort_maps(w, y, rows, A, k, xy.data(), fxy.data());
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < w; j++)
                remap_pixel(src, w, h, stride, cn, xy[((size_t)i * w + j) * 2], xy[((size_t)i * w + j) * 2 + 1], fxy[(size_t)i * w + j],
                            dst + ((size_t)(y + i) * w + j) * cn);
    }
}

// ---------------------------------------------------------------------------------------------
// findCornerMaxima
// ---------------------------------------------------------------------------------------------
static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// cv::cornerHarris(reg, harr, 3, 3, 0.04) where reg is the window [x0, x1) x [y0, y1) of the gray image: the Sobel
// derivatives at the window's rim read the pixels around the window (a filter on a sub-matrix sees its parent image;
// BORDER_REFLECT_101 only at the image's own border), the 3x3 box sums of the products reflect at the window's rim.
// Float arithmetic in a fixed order: derivative = (difference) then smoothing taps scaled by 1 / (4 * 3 * 255).
void corner_harris_window(const uint8_t* gray, int w, int h, int stride, int x0, int y0, int x1, int y1, std::vector<float>& harr) {
    const int rw = x1 - x0, rh = y1 - y0;
    harr.assign((size_t)std::max(rw, 0) * std::max(rh, 0), 0.f);
    if (rw <= 0 || rh <= 0) return;
    const float scale = (float)(1. / ((double)(1 << 2) * 3 * 255.));
    auto G = [&](int x, int y) -> float { return (float)gray[(size_t)reflect101(y, h) * stride + reflect101(x, w)]; };
    std::vector<float> xx((size_t)rw * rh), xy((size_t)rw * rh), yy((size_t)rw * rh);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            const int gx = x0 + x, gy = y0 + y;
            // row pass then column pass of the separable Sobel kernels; the smoothing taps carry the scale
            const float dxm = G(gx + 1, gy - 1) - G(gx - 1, gy - 1), dx0 = G(gx + 1, gy) - G(gx - 1, gy), dxp = G(gx + 1, gy + 1) - G(gx - 1, gy + 1);
            const float dx = (dxm + dxp) * scale + dx0 * (2.f * scale);
            const float sm = (G(gx - 1, gy - 1) + G(gx + 1, gy - 1)) * scale + G(gx, gy - 1) * (2.f * scale);
            const float sp = (G(gx - 1, gy + 1) + G(gx + 1, gy + 1)) * scale + G(gx, gy + 1) * (2.f * scale);
            const float dy = sp - sm;
            xx[(size_t)y * rw + x] = dx * dx, xy[(size_t)y * rw + x] = dx * dy, yy[(size_t)y * rw + x] = dy * dy;
        }
    auto box = [&](const std::vector<float>& a, int x, int y) -> float {   // unnormalised 3x3 sum, rows then columns
        float s = 0.f;
        for (int dy = -1; dy <= 1; dy++) {
            const int yy_ = reflect101(y + dy, rh);
            float r = 0.f;
            for (int dx = -1; dx <= 1; dx++) r += a[(size_t)yy_ * rw + reflect101(x + dx, rw)];
            s += r;
        }
        return s;
    };
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            const float a = box(xx, x, y), b = box(xy, x, y), c = box(yy, x, y);
            harr[(size_t)y * rw + x] = (float)((double)a * c - (double)b * b - 0.04 * ((double)a + c) * ((double)a + c));
        }
}

void find_corner_maxima(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n, int wsize) {
    for (int i = 0; i < n; i++) {
        const int x0 = std::max(0, (int)(corners[i].x - (float)wsize)), y0 = std::max(0, (int)(corners[i].y - (float)wsize));
        const int x1 = std::min(w, (int)(corners[i].x + (float)wsize)), y1 = std::min(h, (int)(corners[i].y + (float)wsize));
        const int rw = x1 - x0, rh = y1 - y0;
        std::vector<float> harr;
        corner_harris_window(gray, w, h, stride, x0, y0, x1, y1, harr);
        if (rw <= 0 || rh <= 0) {
            corners[i] = Pt2f{-1.f + (float)x0, -1.f + (float)y0};
            continue;
        }
        // cv::integral (double sums), then every interior response becomes the sum of the 4x4 block that starts at it
        std::vector<double> I((size_t)(rw + 1) * (rh + 1), 0.0);
        for (int y = 0; y < rh; y++) {
            double row = 0;
            for (int x = 0; x < rw; x++) {
                row += (double)harr[(size_t)y * rw + x];
                I[(size_t)(y + 1) * (rw + 1) + x + 1] = I[(size_t)y * (rw + 1) + x + 1] + row;
            }
        }
        const int bls = 4;
        for (int y = bls; y < rh - bls; y++)
            for (int x = bls; x < rw - bls; x++)
                harr[(size_t)y * rw + x] = (float)(I[(size_t)(y + bls) * (rw + 1) + x + bls] - I[(size_t)(y + bls) * (rw + 1) + x] -
                                                   I[(size_t)y * (rw + 1) + x + bls] + I[(size_t)y * (rw + 1) + x]);
        float bx = -1.f, by = -1.f;
        const float cx = (float)(rw / 2), cy = (float)(rh / 2);
        double maxv = 0;
        for (int y = 0; y < rh; y++)
            for (int x = 0; x < rw; x++) {
                const float d = (float)(std::fabs(cx - (float)x) + std::fabs(cy - (float)y)) / (float)(rw / 2 + rh / 2);
                const float wgt = (float)(1. - (double)d);
                if ((double)(wgt * harr[(size_t)y * rw + x]) > maxv) maxv = (double)(wgt * harr[(size_t)y * rw + x]), bx = (float)x, by = (float)y;
            }
        corners[i] = Pt2f{bx + (float)x0, by + (float)y0};
    }
}

// ---------------------------------------------------------------------------------------------
// cv::Canny(grey, out, 10, 220) — the CANNY threshold method (/root/reference/src/markerdetector.cpp:667-676): aperture 3, L1
// gradient magnitude. OpenCV 3.0 imgproc/src/canny.cpp: Sobel dx, dy (CV_16S, BORDER_REPLICATE), mag = |dx| + |dy| with a
// zero rim around the image, non-maximum suppression along the quantised gradient direction (tan 22.5 deg in 15-bit fixed
// point; the asymmetric > / >= comparisons), then hysteresis: every suppression survivor (mag > low) that is 8-connected
// to a survivor with mag > high is an edge. (The scan's `prev_flag` / "pixel above already an edge" shortcuts only skip
// seeds that a neighbouring seed reaches anyway, so the edge set is this connectivity closure.) The reference holds no
// fixture for this method: PARITY UNPINNED at the OpenCV level.
// ---------------------------------------------------------------------------------------------
void canny_3x3_l1(const uint8_t* src, int w, int h, int stride, int low, int high, uint8_t* dst) {
    auto G = [&](int x, int y) -> int { return src[(size_t)std::min(std::max(y, 0), h - 1) * stride + std::min(std::max(x, 0), w - 1)]; };
    std::vector<int> dx((size_t)w * h), dy((size_t)w * h), mag((size_t)(w + 2) * (h + 2), 0);
    const int ms = w + 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int gx = (G(x + 1, y - 1) + 2 * G(x + 1, y) + G(x + 1, y + 1)) - (G(x - 1, y - 1) + 2 * G(x - 1, y) + G(x - 1, y + 1));
            const int gy = (G(x - 1, y + 1) + 2 * G(x, y + 1) + G(x + 1, y + 1)) - (G(x - 1, y - 1) + 2 * G(x, y - 1) + G(x + 1, y - 1));
            dx[(size_t)y * w + x] = gx, dy[(size_t)y * w + x] = gy;
            mag[(size_t)(y + 1) * ms + x + 1] = std::abs(gx) + std::abs(gy);
        }
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    std::vector<uint8_t> state((size_t)w * h, 0);   // 0 suppressed, 1 survivor, 2 edge
    std::vector<int> stack;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* m = &mag[(size_t)(y + 1) * ms + x + 1];
            const int v = m[0];
            bool keep = false;
            if (v > low) {
                const int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
                const int ax = std::abs(xs);
                const int ay = std::abs(ys) << 15;
                const int tg22x = ax * TG22;
                if (ay < tg22x) {
                    keep = v > m[-1] && v >= m[1];
                } else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x)
                        keep = v > m[-ms] && v >= m[ms];
                    else {
                        const int s = (xs ^ ys) < 0 ? -1 : 1;
                        keep = v > m[-ms - s] && v > m[ms + s];
                    }
                }
            }
            if (keep) {
                state[(size_t)y * w + x] = 1;
                if (v > high) {
                    state[(size_t)y * w + x] = 2;
                    stack.push_back(y * w + x);
                }
            }
        }
    while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int px = p % w, py = p / w;
        for (int oy = -1; oy <= 1; oy++)
            for (int ox = -1; ox <= 1; ox++) {
                const int qx = px + ox, qy = py + oy;
                if (qx < 0 || qx >= w || qy < 0 || qy >= h) continue;
                uint8_t& st = state[(size_t)qy * w + qx];
                if (st == 1) st = 2, stack.push_back(qy * w + qx);
            }
    }
    for (size_t i = 0; i < (size_t)w * h; i++) dst[i] = state[i] == 2 ? 255 : 0;
}

}  // namespace orc
