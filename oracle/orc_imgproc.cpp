// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Restatements of the OpenCV 3.0 imgproc primitives used on the reference's hot path
// (call sites: SURVEY.md §2.2). OpenCV itself is absent; these follow its published algorithms.
#include "orc.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace orc {

// cv::cvtColor(BGR2GRAY) for 8u, OpenCV 3.x fixed-point form (14-bit coefficients).
// Call site: src/markerdetector.cpp:307-310.
void bgr2gray(const uint8_t* bgr, int npix, uint8_t* gray) {
    for (int i = 0; i < npix; i++) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
}

// cv::adaptiveThreshold(src, dst, 255, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV, block, C)
// Call site: src/markerdetector.cpp:662. mean = normalised box filter (BORDER_REPLICATE) rounded to u8,
// dst = (src - mean <= -floor(C)) ? 255 : 0.
void adaptive_threshold_mean_inv(const uint8_t* src, int w, int h, int stride, int block, double C, uint8_t* dst) {
    const int r = block / 2;
    const double scale = 1.0 / (double(block) * block);
    const int idelta = (int)std::floor(C);
    std::vector<int> hs((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * stride;
        int* o = hs.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -r; k <= r; k++) acc += s[std::min(std::max(x + k, 0), w - 1)];
            o[x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = -r; k <= r; k++) acc += hs[(size_t)std::min(std::max(y + k, 0), h - 1) * w + x];
            long m = lrint(acc * scale);  // saturate_cast<uchar>(double): round-half-even, then clip
            if (m > 255) m = 255;
            if (m < 0) m = 0;
            dst[(size_t)y * w + x] = ((int)s[x] - (int)m <= -idelta) ? 255 : 0;
        }
    }
}

// cv::threshold(src, dst, thr, 255, THRESH_BINARY_INV) for 8u (src/markerdetector.cpp:653).
void fixed_threshold_inv(const uint8_t* src, int w, int h, int stride, double thr, uint8_t* dst) {
    int ithr = (int)std::floor(thr);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = src[(size_t)y * stride + x] > ithr ? 0 : 255;
}

// cv::findContours(img, contours, RETR_LIST, CHAIN_APPROX_NONE) — Suzuki-Abe border following as coded in
// OpenCV 3.0 (cvStartFindContours / cvFindNextContour / icvFetchContour). Call site: src/markerdetector.cpp:511.
// The 1-px image frame is zeroed first (3.0/3.1 behaviour); scan covers x in [1,w-2], y in [1,h-2].
static void fetch_contour(int8_t* img, int step, int x0, int y0, int is_hole, std::vector<Pt>& out) {
    static const int dxs[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int dys[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int deltas[16];
    for (int i = 0; i < 8; i++) deltas[i] = deltas[i + 8] = dys[i] * step + dxs[i];
    const int8_t nbd = 2;
    int8_t* i0 = img + (size_t)y0 * step + x0;
    int8_t *i1, *i3, *i4 = nullptr;
    int s, s_end;
    Pt pt{x0, y0};
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
        if (*i1 != 0) break;
    } while (s != s_end);
    if (s == s_end) {  // isolated pixel
        *i0 = (int8_t)(nbd | -128);
        out.push_back(pt);
        return;
    }
    i3 = i0;
    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (int8_t)(nbd | -128);  // east neighbour examined and zero
        else if (*i3 == 1)
            *i3 = nbd;
        out.push_back(pt);
        pt.x += dxs[s];
        pt.y += dys[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

void find_contours_list(const uint8_t* bin, int w, int h, std::vector<Contour>& out) {
    out.clear();
    if (w < 3 || h < 3) return;
    std::vector<int8_t> img((size_t)w * h, 0);
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) img[(size_t)y * w + x] = bin[(size_t)y * w + x] ? 1 : 0;
    std::vector<Contour> found;  // discovery order
    for (int y = 1; y < h - 1; y++) {
        int8_t* row = img.data() + (size_t)y * w;
        int prev = 0;
        for (int x = 1; x < w - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) {
                    prev = p;
                    continue;
                }
                is_hole = 1;
            }
            found.emplace_back();
            Contour& c = found.back();
            c.hole = is_hole;
            c.trig_x = x;
            c.trig_y = y;
            fetch_contour(img.data(), w, x - is_hole, y, is_hole, c.pts);
            prev = row[x];  // the scanner re-reads img[x] after a contour was traced
        }
    }
    // RETR_LIST: each new contour is linked at the head of the list -> reverse discovery order
    out.reserve(found.size());
    for (size_t i = found.size(); i-- > 0;) out.push_back(std::move(found[i]));
}

// cv::approxPolyDP(curve, out, eps, closed=true) for int points (approxPolyDP_<int>, OpenCV 3.0 approx.cpp).
// Call site: src/markerdetector.cpp:522.
void approx_poly_dp_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst, int inner_product_rule) {
    dst.clear();
    const int count = (int)src.size();
    if (count == 0) return;
    struct Range { int start, end; };
    std::vector<Range> stack;
    std::vector<Pt> out;
    Range slice{0, 0}, right_slice{0, 0};
    Pt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    int pos = 0;
    bool le_eps = false;
    eps *= eps;
    auto read_pt = [&](Pt& p, int& ps) {
        p = src[ps];
        if (++ps >= count) ps = 0;
    };
    // 1. two (approximately) farthest points
    right_slice.start = 0;
    for (int i = 0; i < 3; i++) {
        double max_dist = 0;
        pos = (pos + right_slice.start) % count;
        read_pt(start_pt, pos);
        for (int j = 1; j < count; j++) {
            read_pt(pt, pos);
            double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            double dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right_slice.start = j;
            }
        }
        le_eps = max_dist <= eps;
    }
    // 2. seed the stack
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else {
        out.push_back(start_pt);
    }
    // 3. recursive splitting
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = src[slice.end];
        pos = slice.start;
        read_pt(start_pt, pos);
        if (pos != slice.end) {
            double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            double max_dist = 0;
            while (pos != slice.end) {
                read_pt(pt, pos);
                double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right_slice.start = (pos + count - 1) % count;
                }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src[slice.start];
        }
        if (le_eps) {
            out.push_back(start_pt);
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
    // 4. clean-up: drop vertices on (almost) straight lines
    int new_count = (int)out.size();
    const int cnt = new_count;
    auto read_dst = [&](Pt& p, int& ps) {
        p = out[ps];
        if (++ps >= cnt) ps = 0;
    };
    pos = cnt - 1;
    read_dst(start_pt, pos);
    int wpos = pos;
    read_dst(pt, pos);
    for (int i = 0; i < cnt && new_count > 2; i++) {
        read_dst(end_pt, pos);
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        double sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 &&
            (!inner_product_rule || sip >= 0)) {
            new_count--;
            out[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            read_dst(pt, pos);
            i++;
            continue;
        }
        out[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    out.resize(new_count);
    dst.swap(out);
}

// cv::isContourConvex for int points (src/markerdetector.cpp:535).
bool is_contour_convex(const std::vector<Pt>& p) {
    int n = (int)p.size();
    if (n < 3) return false;  // OpenCV asserts total >= 0; a <3 polygon never reaches this call with ==4 filter
    Pt prev = p[(n - 2 + n) % n], cur = p[n - 1];
    int dx0 = cur.x - prev.x, dy0 = cur.y - prev.y;
    int orientation = 0;
    for (int i = 0; i < n; i++) {
        prev = cur;
        cur = p[i];
        int dx = cur.x - prev.x, dy = cur.y - prev.y;
        int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
    }
    return true;
}

// Gaussian elimination with partial pivoting on an n x n system (double). OpenCV solves the 8x8 system of
// getPerspectiveTransform with DECOMP_SVD; the solution of a well-conditioned square system is the same to
// rounding, which is all the nearest-neighbour warp can see.
static bool solve_dense(double* A, double* b, int n) {
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) {
            double v = std::fabs(A[r * n + c]);
            if (v > best) best = v, piv = r;
        }
        if (best == 0) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        double inv = 1.0 / A[c * n + c];
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] * inv;
            if (f == 0) continue;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

// cv::getPerspectiveTransform (src/markerdetector.cpp:695): M maps src quad -> dst quad, M[8] = 1.
void perspective_transform(const Pt2f src[4], const Pt2f dst[4], double M[9]) {
    double a[64], b[8];
    std::memset(a, 0, sizeof(a));
    for (int i = 0; i < 4; i++) {
        double sx = src[i].x, sy = src[i].y, dx = dst[i].x, dy = dst[i].y;
        double* r0 = a + i * 8;
        double* r1 = a + (i + 4) * 8;
        r0[0] = r1[3] = sx;
        r0[1] = r1[4] = sy;
        r0[2] = r1[5] = 1;
        r0[6] = -sx * dx;
        r0[7] = -sy * dx;
        r1[6] = -sx * dy;
        r1[7] = -sy * dy;
        b[i] = dx;
        b[i + 4] = dy;
    }
    if (!solve_dense(a, b, 8)) {
        for (int i = 0; i < 8; i++) b[i] = 0;
    }
    for (int i = 0; i < 8; i++) M[i] = b[i];
    M[8] = 1.0;
}

// cv::warpPerspective(src, dst, M, Size(size,size), INTER_NEAREST) with BORDER_CONSTANT(0)
// (src/markerdetector.cpp:696): M is inverted (3x3 adjugate form), dst(x,y) = src(round(X/W), round(Y/W)).
void warp_perspective_nearest(const uint8_t* src, int w, int h, int stride, const double M[9], int size, uint8_t* dst) {
    double iM[9];
    {
        const double* m = M;
        double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                   m[2] * (m[3] * m[7] - m[4] * m[6]);
        d = d != 0 ? 1. / d : 0;
        iM[0] = (m[4] * m[8] - m[5] * m[7]) * d;
        iM[1] = (m[2] * m[7] - m[1] * m[8]) * d;
        iM[2] = (m[1] * m[5] - m[2] * m[4]) * d;
        iM[3] = (m[5] * m[6] - m[3] * m[8]) * d;
        iM[4] = (m[0] * m[8] - m[2] * m[6]) * d;
        iM[5] = (m[2] * m[3] - m[0] * m[5]) * d;
        iM[6] = (m[3] * m[7] - m[4] * m[6]) * d;
        iM[7] = (m[1] * m[6] - m[0] * m[7]) * d;
        iM[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    }
    for (int y = 0; y < size; y++) {
        double X0 = iM[1] * y + iM[2];
        double Y0 = iM[4] * y + iM[5];
        double W0 = iM[7] * y + iM[8];
        for (int x = 0; x < size; x++) {
            double W = W0 + iM[6] * x;
            W = W != 0 ? 1. / W : 0;
            double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + iM[0] * x) * W));
            double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + iM[3] * x) * W));
            long X = lrint(fX), Y = lrint(fY);
            uint8_t v = 0;
            if (X >= 0 && X < w && Y >= 0 && Y < h) v = src[(size_t)Y * stride + X];
            dst[y * size + x] = v;
        }
    }
}

// Otsu threshold value (cv::threshold(..., THRESH_OTSU), getThreshVal_Otsu_8u); src/arucofidmarkers.cpp:446.
int otsu_threshold(const uint8_t* img, int n) {
    int hist[256] = {0};
    for (int i = 0; i < n; i++) hist[img[i]]++;
    double mu = 0, scale = 1. / n;
    for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    return (int)max_val;
}

// cv::getRectSubPix 8u -> 32f (bilinear, replicated border). Used by cornerSubPix (src/markerdetector.cpp:403).
void get_rect_subpix_8u32f(const uint8_t* src, int w, int h, int stride, int pw, int ph, float cx, float cy, float* dst) {
    cx -= (pw - 1) * 0.5f;
    cy -= (ph - 1) * 0.5f;
    int ix = (int)std::floor(cx), iy = (int)std::floor(cy);
    float a = cx - ix, b = cy - iy;
    float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    for (int i = 0; i < ph; i++) {
        int y0 = std::min(std::max(iy + i, 0), h - 1), y1 = std::min(std::max(iy + i + 1, 0), h - 1);
        for (int j = 0; j < pw; j++) {
            int x0 = std::min(std::max(ix + j, 0), w - 1), x1 = std::min(std::max(ix + j + 1, 0), w - 1);
            float s0 = src[(size_t)y0 * stride + x0] * a11 + src[(size_t)y0 * stride + x1] * a12 +
                       src[(size_t)y1 * stride + x0] * a21 + src[(size_t)y1 * stride + x1] * a22;
            dst[i * pw + j] = s0;
        }
    }
}

// cv::getRectSubPix 8u -> 8u (16.16 fixed-point bilinear). Used by SubPixelCorner (src/subpixelcorner.cpp:124-126).
void get_rect_subpix_8u8u(const uint8_t* src, int w, int h, int stride, int pw, int ph, float cx, float cy, uint8_t* dst) {
    cx -= (pw - 1) * 0.5f;
    cy -= (ph - 1) * 0.5f;
    int ix = (int)std::floor(cx), iy = (int)std::floor(cy);
    float a = cx - ix, b = cy - iy;
    auto fix = [](float v) { return (int)lrint((double)(v * (1 << 16))); };
    int a11 = fix((1.f - a) * (1.f - b)), a12 = fix(a * (1.f - b)), a21 = fix((1.f - a) * b), a22 = fix(a * b);
    for (int i = 0; i < ph; i++) {
        int y0 = std::min(std::max(iy + i, 0), h - 1), y1 = std::min(std::max(iy + i + 1, 0), h - 1);
        for (int j = 0; j < pw; j++) {
            int x0 = std::min(std::max(ix + j, 0), w - 1), x1 = std::min(std::max(ix + j + 1, 0), w - 1);
            int s0 = src[(size_t)y0 * stride + x0] * a11 + src[(size_t)y0 * stride + x1] * a12 +
                     src[(size_t)y1 * stride + x0] * a21 + src[(size_t)y1 * stride + x1] * a22;
            dst[i * pw + j] = (uint8_t)((s0 + (1 << 15)) >> 16);
        }
    }
}

}  // namespace orc
