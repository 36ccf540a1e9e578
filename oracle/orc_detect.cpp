// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Restatement of the reference pipeline: aruco::MarkerDetector::detect (src/markerdetector.cpp:302-478),
// detectRectangles (:496-635), warp (:684-697), refineCandidateLines (:931-997 with helpers :83-153),
// FiducidalMarkers::detect (src/arucofidmarkers.cpp:63-137,168-204,438-452), SubPixelCorner::RefineCorner
// (src/subpixelcorner.cpp:7-189), getObjectPoints (src/marker.cpp:91-108), perimeter (src/utils.h:39-46) and
// BoardDetector::detect (src/boarddetector.cpp:90-205).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "orc.h"

namespace orc {

static inline long cv_round(double v) { return lrint(v); }

// cv::norm(Point2f) -> double sqrt of double-accumulated squares
static inline double norm2f(float dx, float dy) { return std::sqrt((double)dx * dx + (double)dy * dy); }

// src/utils.h:39-46 — float accumulator, double norms
static float perimeter(const Pt2f* a, int n) {
    float sum = 0;
    for (int i = 0; i < n; i++) {
        int i2 = (i + 1) % n;
        sum += norm2f(a[i].x - a[i2].x, a[i].y - a[i2].y);
    }
    return sum;
}

// ---------------------------------------------------------------------------------------------
// 5x5 Hamming decode (src/arucofidmarkers.cpp)
// ---------------------------------------------------------------------------------------------
static int count_nonzero(const uint8_t* img, int stride, int x0, int y0, int cell) {
    int n = 0;
    for (int y = 0; y < cell; y++)
        for (int x = 0; x < cell; x++) n += img[(y0 + y) * stride + x0 + x] != 0;
    return n;
}

static int hamm_dist(const uint8_t b[5][5]) {  // :74-98
    static const uint8_t words[4][5] = {{1, 0, 0, 0, 0}, {1, 0, 1, 1, 1}, {0, 1, 0, 0, 1}, {0, 1, 1, 1, 0}};
    int dist = 0;
    for (int y = 0; y < 5; y++) {
        int best = 100000;
        for (int p = 0; p < 4; p++) {
            int s = 0;
            for (int x = 0; x < 5; x++) s += b[y][x] != words[p][x];
            best = std::min(best, s);
        }
        dist += best;
    }
    return dist;
}

// analyzeMarkerImage :100-137 on an already binarised patch. nRotations starts at 0 (the
// reference leaves it uninitialised when rotation 0 wins; 0 is the intent).
int fiducial_decode(const uint8_t* patch, int size, int* nrot) {
    *nrot = 0;
    int sw = size / 7;
    // checkBorders :168-184
    for (int y = 0; y < 7; y++) {
        int inc = (y == 0 || y == 6) ? 1 : 6;
        for (int x = 0; x < 7; x += inc)
            if (count_nonzero(patch, size, x * sw, y * sw, sw) > (sw * sw) / 2) return -1;
    }
    uint8_t rot[4][5][5];
    for (int y = 0; y < 5; y++)  // getMarkerCode :189-204
        for (int x = 0; x < 5; x++) rot[0][y][x] = count_nonzero(patch, size, (x + 1) * sw, (y + 1) * sw, sw) > (sw * sw) / 2;
    int min_dist = hamm_dist(rot[0]);
    for (int r = 1; r < 4; r++) {
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) rot[r][i][j] = rot[r - 1][5 - j - 1][i];  // rotate :63-72
        int d = hamm_dist(rot[r]);
        if (d < min_dist) min_dist = d, *nrot = r;
    }
    if (min_dist != 0) return -1;
    int id = 0;
    for (int y = 0; y < 5; y++) id |= (rot[*nrot][y][1] << 1 | rot[*nrot][y][3]) << 2 * (4 - y);
    return id;
}

// FiducidalMarkers::detect :438-452 — Otsu binarisation in place, then decode.
int fiducial_detect(uint8_t* patch, int size, int* nrot) {
    int t = otsu_threshold(patch, size * size);
    for (int i = 0; i < size * size; i++) patch[i] = patch[i] > t ? 255 : 0;
    return fiducial_decode(patch, size, nrot);
}

// HighlyReliableMarkers::detect (src/highlyreliablemarkers.cpp:332-383): Otsu binarisation, inner n x n cells by majority
// (getMarkerCode, src/arucofidmarkers.cpp:189-204; the border cells are not checked), the code in its four rotations
// (MarkerCode::set :113-142), then the dictionary entry that equals one of the rotations (first rotation that matches,
// BalancedBinaryTree::findId :505-518) or, failing that, the nearest entry within the correction distance
// (Dictionary::distance :262-274 with MarkerCode::distance :160-170: first minimum over entries, then over rotations).
// Returns the entry's position in the dictionary. The reference looks exact matches up through 32-bit ids (2 << bit
// position, :137-138), which are unique for n <= 5 (and overflow for larger n); with unique ids an exact match is the distance-0 case of the
// nearest-entry search with the same first-minimum order, which is what is restated here.
int hrm_detect(uint8_t* patch, int size, const HrmDict& d, int* nrot) {
    *nrot = 0;
    const int n = d.n;
    if (n <= 0 || n * n > 64 || d.codes.empty()) return -1;
    int t = otsu_threshold(patch, size * size);
    for (int i = 0; i < size * size; i++) patch[i] = patch[i] > t ? 255 : 0;
    const int cell = size / (n + 2);
    uint64_t rot[4] = {0, 0, 0, 0};
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            if (!(count_nonzero(patch, size, (x + 1) * cell, (y + 1) * cell, cell) > (cell * cell) / 2)) continue;
            const int ry[4] = {y, x, n - y - 1, n - x - 1}, rx[4] = {x, n - y - 1, n - x - 1, y};
            for (int r = 0; r < 4; r++) rot[r] |= 1ull << (ry[r] * n + rx[r]);
        }
    unsigned best = (unsigned)(n * n), best_marker = 0, best_rot = 0;
    for (size_t i = 0; i < d.codes.size(); i++) {
        unsigned dm = (unsigned)(n * n), rm = 0;
        for (unsigned r = 0; r < 4; r++) {
            unsigned h = (unsigned)__builtin_popcountll(d.codes[i] ^ rot[r]);
            if (h < dm) dm = h, rm = r;
        }
        if (dm < best) best = dm, best_marker = (unsigned)i, best_rot = rm;
    }
    const unsigned correction = (unsigned)(d.rate * (float)((d.tau0 - 1) / 2));   // loadDictionary :318
    if (best == (unsigned)(n * n) || best > correction) return -1;
    *nrot = (int)best_rot;
    return (int)best_marker;
}

// ---------------------------------------------------------------------------------------------
// detectRectangles (src/markerdetector.cpp:496-635)
// ---------------------------------------------------------------------------------------------
void Detector::detect_rectangles(const std::vector<std::vector<uint8_t>>& thr, int w, int h, std::vector<Candidate>& out) {
    out.clear();
    const int minSize = (int)(prm.min_size * std::max(w, h) * 4);
    const int maxSize = (int)(prm.max_size * std::max(w, h) * 4);
    std::vector<Candidate> cands;
    for (size_t t = 0; t < thr.size(); t++) {
        std::vector<Contour> cs;
        find_contours_list(thr[t].data(), w, h, cs);
        std::vector<Pt> approx;
        for (size_t i = 0; i < cs.size(); i++) {
            const size_t n = cs[i].pts.size();
            if (n <= (size_t)minSize || n >= (size_t)maxSize) continue;
            approx_poly_dp_closed(cs[i].pts, double(n) * 0.05, approx, prm.approx_inner_product);
            if (approx.size() != 4) continue;
            if (!is_contour_convex(approx)) continue;
            // :542-552 — intended form: minimum side of the quad must exceed 10 px
            float minDist = 1e10f;
            for (int j = 0; j < 4; j++) {
                float d = (float)norm2f((float)(approx[j].x - approx[(j + 1) % 4].x), (float)(approx[j].y - approx[(j + 1) % 4].y));
                if (d < minDist) minDist = d;
            }
            if (minDist <= 10) continue;
            Candidate c;
            for (int j = 0; j < 4; j++) c.c[j] = c.c0[j] = Pt2f{(float)approx[j].x, (float)approx[j].y};
            c.idx = (int)i;
            c.contour = cs[i].pts;
            cands.push_back(std::move(c));
        }
        if (t == thr.size() / 2) contours.swap(cs);
    }
    // orientation :566-581
    std::vector<char> swapped(cands.size(), 0);
    for (size_t i = 0; i < cands.size(); i++) {
        Pt2f* c = cands[i].c;
        float d1x = c[1].x - c[0].x, d1y = c[1].y - c[0].y, d2x = c[2].x - c[0].x, d2y = c[2].y - c[0].y;
        float o = (d1x * d2y) - (d1y * d2x);
        if (o < 0.0) {
            std::swap(c[1], c[3]);
            swapped[i] = 1;
        }
        for (int j = 0; j < 4; j++) cands[i].c0[j] = c[j];
    }
    // near-duplicate removal :586-613
    std::vector<char> rem(cands.size(), 0);
    for (size_t i = 0; i < cands.size(); i++)
        for (size_t j = i + 1; j < cands.size(); j++) {
            bool near = true;
            for (int c = 0; c < 4 && near; c++) {
                float d = (float)norm2f(cands[i].c[c].x - cands[j].c[c].x, cands[i].c[c].y - cands[j].c[c].y);
                near = d < 6;
            }
            if (!near) continue;
            if (perimeter(cands[i].c, 4) > perimeter(cands[j].c, 4))
                rem[j] = 1;
            else
                rem[i] = 1;
        }
    for (size_t i = 0; i < cands.size(); i++) {
        if (rem[i]) continue;
        out.push_back(cands[i]);
        if (swapped[i]) std::reverse(out.back().contour.begin(), out.back().contour.end());
    }
}

// ---------------------------------------------------------------------------------------------
// LINES corner refinement (src/markerdetector.cpp:931-997)
// ---------------------------------------------------------------------------------------------
// interpolate2Dline :83-130 — least-squares line through points; the reference solves the n x 2 system with a
// float32 SVD, restated here as the (double) normal-equation solution rounded to float.
static void fit_line(const std::vector<Pt2f>& p, float line[3]) {
    float minX = p[0].x, maxX = p[0].x, minY = p[0].y, maxY = p[0].y;
    for (size_t i = 1; i < p.size(); i++) {
        minX = std::min(minX, p[i].x), maxX = std::max(maxX, p[i].x);
        minY = std::min(minY, p[i].y), maxY = std::max(maxY, p[i].y);
    }
    bool yx = (maxX - minX > maxY - minY);  // regress y on x
    double n = (double)p.size(), su = 0, sv = 0, suu = 0, suv = 0;
    // centre on the first point for conditioning
    double u0 = yx ? p[0].x : p[0].y, v0 = yx ? p[0].y : p[0].x;
    for (auto& q : p) {
        double u = (yx ? q.x : q.y) - u0, v = (yx ? q.y : q.x) - v0;
        su += u, sv += v, suu += u * u, suv += u * v;
    }
    double det = n * suu - su * su;
    double a, c;
    if (std::fabs(det) > 0) {
        a = (n * suv - su * sv) / det;
        c = (sv - a * su) / n;
    } else {  // all abscissae equal: minimum-norm solution of the rank-deficient system
        a = 0;
        c = sv / n;
    }
    c = c + v0 - a * u0;
    if (yx) {
        line[0] = (float)a, line[1] = -1.f, line[2] = (float)c;
    } else {
        line[0] = -1.f, line[1] = (float)a, line[2] = (float)c;
    }
}

// getCrossPoint :132-139 — 2x2 solve
static Pt2f cross_point(const float l1[3], const float l2[3]) {
    double a = l1[0], b = l1[1], c = l2[0], d = l2[1], e = -l1[2], f = -l2[2];
    double det = a * d - b * c;
    Pt2f r;
    r.x = (float)((e * d - b * f) / det);
    r.y = (float)((a * f - e * c) / det);
    return r;
}

void refine_lines(Candidate& cand, const float* K, const float* dist, int ndist) {
    const int n = (int)cand.contour.size();
    int ci[4] = {0, 0, 0, 0};
    for (int j = 0; j < n; j++)
        for (int k = 0; k < 4; k++)
            if (cand.contour[j].x == (int)cv_round(cand.c[k].x) && cand.contour[j].y == (int)cv_round(cand.c[k].y)) ci[k] = j;
    bool inverse;
    if ((ci[1] > ci[0]) && (ci[2] > ci[1] || ci[2] < ci[0]))
        inverse = false;
    else if (ci[2] > ci[1] && ci[2] < ci[0])
        inverse = false;
    else
        inverse = true;
    int inc = inverse ? -1 : 1;
    std::vector<Pt2f> c2f(n);
    for (int j = 0; j < n; j++) c2f[j] = Pt2f{(float)cand.contour[j].x, (float)cand.contour[j].y};
    bool undist = K && dist && ndist > 0;
    if (undist) undistort_points(c2f.data(), n, K, dist, ndist, K, c2f.data());
    std::vector<Pt2f> side[4];
    for (int l = 0; l < 4; l++) {
        int j = ci[l];
        int guard = 0;
        while (j != ci[(l + 1) % 4] && guard++ <= 2 * n) {
            side[l].push_back(c2f[j]);
            // :967 — int + int converted to size_t before the modulo
            j = (int)((uint64_t)(int64_t)(j + inc) % (uint64_t)n);
        }
        if (side[l].size() == 1) side[l].push_back(c2f[ci[(l + 1) % 4]]);
    }
    float lines[4][3];
    for (int j = 0; j < 4; j++) fit_line(side[j], lines[j]);
    Pt2f cross[4];
    for (int i = 0; i < 4; i++) cross[i] = cross_point(lines[i], lines[(i + 3) % 4]);
    if (undist) {
        // distortPoints :141-153 — normalise with float K, project with zero extrinsics
        double Kd[9], k[8];
        for (int i = 0; i < 9; i++) Kd[i] = K[i];
        for (int i = 0; i < 8; i++) k[i] = i < ndist ? (double)dist[i] : 0.0;
        for (int i = 0; i < 4; i++) {
            float X = (cross[i].x - K[2]) / K[0], Y = (cross[i].y - K[5]) / K[4];
            double M[3] = {X, Y, 1.0}, r0[3] = {0, 0, 0}, t0[3] = {0, 0, 0}, m[2];
            project_points(M, 1, r0, t0, Kd, k, m, nullptr, nullptr);
            cross[i].x = (float)m[0];
            cross[i].y = (float)m[1];
        }
    }
    for (int j = 0; j < 4; j++) cand.c[j] = cross[j];
}

// ---------------------------------------------------------------------------------------------
// cv::cornerSubPix (SUBPIX branch, src/markerdetector.cpp:402-405)
// ---------------------------------------------------------------------------------------------
void corner_subpix(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n, int win, int max_iter, double eps) {
    const int ww = win * 2 + 1;
    std::vector<float> mask(ww * ww), buf((ww + 2) * (ww + 2));
    for (int i = 0; i < ww; i++) {
        float y = (float)(i - win) / win;
        float vy = std::exp(-y * y);
        for (int j = 0; j < ww; j++) {
            float x = (float)(j - win) / win;
            mask[i * ww + j] = (float)(vy * std::exp(-x * x));
        }
    }
    max_iter = std::min(std::max(max_iter, 1), 100);
    eps = std::max(eps, 0.);
    eps *= eps;
    for (int p = 0; p < n; p++) {
        Pt2f cT = corners[p], cI = cT;
        int iter = 0;
        double err = 0;
        do {
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_subpix_8u32f(gray, w, h, stride, ww + 2, ww + 2, cI.x, cI.y, buf.data());
            const float* sp = buf.data() + (ww + 2) + 1;
            for (int i = 0, k = 0; i < ww; i++, sp += ww + 2) {
                double py = i - win;
                for (int j = 0; j < ww; j++, k++) {
                    double m = mask[k];
                    double tgx = sp[j + 1] - sp[j - 1];
                    double tgy = sp[j + ww + 2] - sp[j - ww - 2];
                    double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    double px = j - win;
                    a += gxx, b += gxy, c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            Pt2f cI2;
            cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
            cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
            err = (cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y);
            cI = cI2;
            if (cI.x < 0 || cI.x >= w || cI.y < 0 || cI.y >= h) break;
        } while (++iter < max_iter && err > eps);
        if (std::fabs(cI.x - cT.x) > win || std::fabs(cI.y - cT.y) > win) cI = cT;
        corners[p] = cI;
    }
}

// ---------------------------------------------------------------------------------------------
// SubPixelCorner::RefineCorner (HARRIS branch, src/subpixelcorner.cpp:70-189), with its quirks:
// exactly one iteration, y update uses A*F only, bounds test compares y against the width.
// ---------------------------------------------------------------------------------------------
void corner_harris_refine(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n) {
    const int win = 15, ap = 3, ps = win + 2 * (ap / 2);  // 17
    float maskX[15], mask[15][15];
    const double coeff = 1. / (win * win);
    for (int i = -win / 2, k = 0; i <= win / 2; i++, k++) maskX[k] = (float)std::exp(-i * i * coeff);
    for (int i = 0; i < win; i++)
        for (int j = 0; j < win; j++) mask[i][j] = maskX[j] * maskX[i];
    uint8_t local[17 * 17];
    float Dx[17][17], Dy[17][17];
    auto refl = [&](int v) { return v < 0 ? -v : (v >= ps ? 2 * ps - 2 - v : v); };  // BORDER_REFLECT_101
    for (int k = 0; k < n; k++) {
        Pt2f est = corners[k], cur;
        if (est.x < 0 || est.y < 0 || est.y > h || est.y > w) continue;
        cur = est;
        get_rect_subpix_8u8u(gray, w, h, stride, ps, ps, cur.x, cur.y, local);
        for (int y = 0; y < ps; y++)
            for (int x = 0; x < ps; x++) {
                auto P = [&](int yy, int xx) { return (int)local[refl(yy) * ps + refl(xx)]; };
                int gx = (P(y - 1, x + 1) + 2 * P(y, x + 1) + P(y + 1, x + 1)) - (P(y - 1, x - 1) + 2 * P(y, x - 1) + P(y + 1, x - 1));
                int gy = (P(y + 1, x - 1) + 2 * P(y + 1, x) + P(y + 1, x + 1)) - (P(y - 1, x - 1) + 2 * P(y - 1, x) + P(y - 1, x + 1));
                Dx[y][x] = (float)gx;
                Dy[y][x] = (float)gy;
            }
        double A = 0, B = 0, C = 0, D = 0, E = 0, F = 0;
        for (int i = ap / 2; i <= win; i++) {
            int ly = i - win / 2 - ap / 2;
            for (int j = ap / 2; j <= win; j++) {
                int lx = j - win / 2 - ap / 2;
                double val = mask[ly + win / 2][lx + win / 2];
                double dxx = Dx[i][j] * Dx[i][j] * val;
                double dyy = Dy[i][j] * Dy[i][j] * val;
                double dxy = Dx[i][j] * Dy[i][j] * val;
                A += dxx, B += dxy, E += dyy;
                C += dxx * lx + dxy * ly;
                F += dxy * lx + dyy * ly;
            }
        }
        double det = A * E - B * B;
        if (std::fabs(det) > DBL_EPSILON * DBL_EPSILON) {
            det = 1.0 / det;
            est.x = (float)(cur.x + ((C * E) - (B * F)) * det);
            est.y = (float)(cur.y + ((A * F) - (C * D)) * det);
        }
        if (std::fabs(corners[k].x - est.x) > win || std::fabs(corners[k].y - est.y) > win) est = corners[k];
        corners[k] = est;
    }
}

// ---------------------------------------------------------------------------------------------
// MarkerDetector::detect (src/markerdetector.cpp:302-478)
// ---------------------------------------------------------------------------------------------
int Detector::detect(const uint8_t* gray, int W, int H, int stride, const float* K, const float* dist, int ndist,
                     float marker_size, int y_perp, std::vector<Marker>& out) {
    out.clear();
    w = W, h = H;

    // thresholds :322-334
    const int nthr = 2 * prm.thres_range + 1;
    std::vector<std::vector<uint8_t>> thr(nthr, std::vector<uint8_t>((size_t)W * H));
    for (int i = 0; i < nthr; i++) {
        double p1 = nthr == 1 ? prm.thres_p1 : prm.thres_p1 - prm.thres_range + prm.thres_range * i;
        if (prm.thres_method == 2) {
            canny_3x3_l1(gray, W, H, stride, 10, 220, thr[i].data());   // :667-676, the parameters are not used
        } else if (prm.thres_method == 0) {
            fixed_threshold_inv(gray, W, H, stride, p1, thr[i].data());
        } else {
            if (p1 < 3)
                p1 = 3;
            else if (((int)p1) % 2 != 1)
                p1 = (int)(p1 + 1);
            adaptive_threshold_mean_inv(gray, W, H, stride, (int)p1, prm.thres_p2, thr[i].data());
        }
    }
    thres = thr[nthr / 2];
    // rectangles :342
    detect_rectangles(thr, W, H, candidates);
    // identify :350-368
    const int ws = prm.warp_size;
    std::vector<uint8_t> patch((size_t)ws * ws);
    for (auto& c : candidates) {
        Pt2f dst[4] = {{0, 0}, {(float)(ws - 1), 0}, {(float)(ws - 1), (float)(ws - 1)}, {0, (float)(ws - 1)}};
        double M[9];
        perspective_transform(c.c, dst, M);
        warp_perspective_nearest(gray, W, H, stride, M, ws, patch.data());
        c.id = hrm.n > 0 ? hrm_detect(patch.data(), ws, hrm, &c.nrot) : fiducial_detect(patch.data(), ws, &c.nrot);
        if (c.id != -1) {
            if (prm.corner_method == 3) refine_lines(c, K, dist, ndist);
            std::rotate(c.c, c.c + 4 - c.nrot, c.c + 4);
        }
    }
    rejected.clear();
    std::vector<Marker> det;
    for (auto& c : candidates) {
        if (c.id != -1) {
            Marker m;
            m.id = c.id;
            for (int k = 0; k < 4; k++) m.c[k] = c.c[k];
            m.ssize = -1;
            m.has_pose = 0;
            for (int k = 0; k < 3; k++) m.rvec[k] = m.tvec[k] = 0;
            det.push_back(m);
        } else {
            rejected.push_back(c);
        }
    }
    // HARRIS / SUBPIX :388-410
    if (!det.empty() && (prm.corner_method == 1 || prm.corner_method == 2)) {
        std::vector<Pt2f> cs;
        for (auto& m : det)
            for (int k = 0; k < 4; k++) cs.push_back(m.c[k]);
        // locked corners (:398-399): search the corner in the surroundings of the estimated location first
        if (prm.use_locked_corners) find_corner_maxima(gray, W, H, stride, cs.data(), (int)cs.size(), (int)prm.thres_p1);
        if (prm.corner_method == 1)
            corner_harris_refine(gray, W, H, stride, cs.data(), (int)cs.size());
        else
            corner_subpix(gray, W, H, stride, cs.data(), (int)cs.size(), (int)prm.thres_p1, 8, 0.005);
        for (size_t i = 0; i < det.size(); i++)
            for (int k = 0; k < 4; k++) det[i].c[k] = cs[i * 4 + k];
    }
    // sort by id :417 (stable: ties keep candidate order), same-id dedupe :421-430
    std::stable_sort(det.begin(), det.end(), [](const Marker& a, const Marker& b) { return a.id < b.id; });
    std::vector<char> rem(det.size(), 0);
    for (int i = 0; i < (int)det.size() - 1; i++) {
        if (det[i].id == det[i + 1].id && !rem[i + 1]) {
            if (perimeter(det[i].c, 4) > perimeter(det[i + 1].c, 4))
                rem[i + 1] = 1;
            else
                rem[i] = 1;
        }
    }
    // border filter :433-444 — Rect(Point(size)*t, Point(size)*(1-t)), contains(Point(cvRound(corner)))
    {
        int x1 = (int)cv_round((float)W * prm.border_dist), y1 = (int)cv_round((float)H * prm.border_dist);
        int x2 = (int)cv_round((float)W * (1.0f - prm.border_dist)), y2 = (int)cv_round((float)H * (1.0f - prm.border_dist));
        int rx = std::min(x1, x2), ry = std::min(y1, y2), rw = std::max(x1, x2) - rx, rh = std::max(y1, y2) - ry;
        for (size_t i = 0; i < det.size(); i++)
            for (int c = 0; c < 4; c++) {
                int px = (int)cv_round(det[i].c[c].x), py = (int)cv_round(det[i].c[c].y);
                if (!(rx <= px && px < rx + rw && ry <= py && py < ry + rh)) {
                    rem[i] = 1;
                    break;
                }
            }
    }
    for (size_t i = 0; i < det.size(); i++)
        if (!rem[i]) out.push_back(det[i]);
    // pose :450-467
    if (K && marker_size > 0) {
        float hs = marker_size / 2.f;  // getObjectPoints src/marker.cpp:91-108
        Pt3f obj[4] = {{-hs, -hs, 0}, {-hs, hs, 0}, {hs, hs, 0}, {hs, -hs, 0}};
        for (auto& m : out) {
            m.has_pose = solve_pnp_iterative(obj, m.c, 4, K, dist, ndist, m.rvec, m.tvec) ? 1 : 0;
            m.ssize = marker_size;
            if (y_perp) rotate_x_axis(m.rvec);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// BoardDetector::detect (src/boarddetector.cpp:90-205)
// ---------------------------------------------------------------------------------------------
float board_detect(const std::vector<Marker>& detected, const BoardConf& bc, const float* K, const float* dist, int ndist,
                   float marker_size, float repj_err_thres, int y_perp, Board& out) {
    out.markers.clear();
    out.has_pose = 0;
    out.prob = 0;
    for (int i = 0; i < 3; i++) out.rvec[i] = out.tvec[i] = 0;
    if (bc.ids.empty()) return -1;
    float ssize = -1;
    auto onorm = [&](int a, int b) {
        double dx = bc.obj[a].x - bc.obj[b].x, dy = bc.obj[a].y - bc.obj[b].y, dz = bc.obj[a].z - bc.obj[b].z;
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    if (bc.info_type == 0 && marker_size > 0)
        ssize = marker_size;
    else if (bc.info_type == 1)
        ssize = (float)onorm(0, 1);
    for (auto& m : detected)
        if (std::find(bc.ids.begin(), bc.ids.end(), m.id) != bc.ids.end()) {
            out.markers.push_back(m);
            out.markers.back().ssize = ssize;
        }
    if (out.markers.empty() || !K) return 0;
    bool enough = (marker_size > 0 && bc.info_type == 0) || bc.info_type == 1;
    if (!enough) return 0;
    double mpp = bc.info_type == 0 ? marker_size / onorm(0, 1) : 1;
    std::vector<Pt3f> obj;
    std::vector<Pt2f> img;
    for (auto& m : out.markers) {
        size_t idx = std::find(bc.ids.begin(), bc.ids.end(), m.id) - bc.ids.begin();
        for (int p = 0; p < 4; p++) {
            img.push_back(m.c[p]);
            const Pt3f& o = bc.obj[idx * 4 + p];
            // Point3f * double -> saturate_cast<float>(component * double)
            obj.push_back(Pt3f{(float)(o.x * mpp), (float)(o.y * mpp), (float)(o.z * mpp)});
        }
    }
    float zeros[4] = {0, 0, 0, 0};
    if (!dist || ndist == 0) dist = zeros, ndist = 4;
    out.has_pose = solve_pnp_iterative(obj.data(), img.data(), (int)obj.size(), K, dist, ndist, out.rvec, out.tvec);
    if (repj_err_thres > 0 && out.has_pose) {
        double Kd[9], k[8];
        for (int i = 0; i < 9; i++) Kd[i] = K[i];
        for (int i = 0; i < 8; i++) k[i] = i < ndist ? (double)dist[i] : 0.0;
        std::vector<Pt3f> obj2;
        std::vector<Pt2f> img2;
        for (size_t i = 0; i < obj.size(); i++) {
            double M[3] = {obj[i].x, obj[i].y, obj[i].z}, m[2];
            project_points(M, 1, out.rvec, out.tvec, Kd, k, m, nullptr, nullptr);
            float rx = (float)m[0], ry = (float)m[1];
            float err = (float)norm2f(rx - img[i].x, ry - img[i].y);
            if (err < repj_err_thres) obj2.push_back(obj[i]), img2.push_back(img[i]);
        }
        out.has_pose = solve_pnp_iterative(obj2.data(), img2.data(), (int)obj2.size(), K, dist, ndist, out.rvec, out.tvec);
    }
    if (y_perp && out.has_pose) rotate_x_axis(out.rvec);
    out.prob = float(out.markers.size()) / float(bc.ids.size());
    return out.prob;
}

}  // namespace orc
