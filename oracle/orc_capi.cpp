// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h). C entry points for ctypes (tests, smoke, bench cpu_baseline).
#include <cstring>

#include "orc.h"

using namespace orc;

extern "C" {

struct orc_marker_t {
    int32_t id;
    float corners[8];
    float ssize;
    int32_t has_pose;
    int32_t pad_;
    double rvec[3];
    double tvec[3];
};

struct orc_params_t {
    int32_t thres_method;
    double thres_p1, thres_p2;
    int32_t thres_range;
    int32_t corner_method;
    float min_size, max_size;
    int32_t warp_size;
    float border_dist;
    int32_t use_locked_corners;
    int32_t approx_inner_product;
};

static void to_c(const Marker& m, orc_marker_t* o) {
    std::memset(o, 0, sizeof(*o));
    o->id = m.id;
    for (int k = 0; k < 4; k++) o->corners[2 * k] = m.c[k].x, o->corners[2 * k + 1] = m.c[k].y;
    o->ssize = m.ssize;
    o->has_pose = m.has_pose;
    for (int k = 0; k < 3; k++) o->rvec[k] = m.rvec[k], o->tvec[k] = m.tvec[k];
}
static Marker from_c(const orc_marker_t& o) {
    Marker m;
    m.id = o.id;
    for (int k = 0; k < 4; k++) m.c[k] = Pt2f{o.corners[2 * k], o.corners[2 * k + 1]};
    m.ssize = o.ssize;
    m.has_pose = o.has_pose;
    for (int k = 0; k < 3; k++) m.rvec[k] = o.rvec[k], m.tvec[k] = o.tvec[k];
    return m;
}

void* orc_create() { return new Detector(); }
// HighlyReliableMarkers::loadDictionary + setMakerDetectorFunction(HighlyReliableMarkers::detect); count = 0 restores the fiducial decoder
void orc_set_hrm(void* h, int n, int count, const uint64_t* codes, int tau0, float rate) {
    HrmDict& d = ((Detector*)h)->hrm;
    d.n = count > 0 ? n : 0, d.tau0 = tau0, d.rate = rate;
    d.codes.assign(codes, codes + (count > 0 ? count : 0));
}
void orc_destroy(void* h) { delete (Detector*)h; }

void orc_get_params(void* h, orc_params_t* p) {
    const Params& q = ((Detector*)h)->prm;
    p->thres_method = q.thres_method, p->thres_p1 = q.thres_p1, p->thres_p2 = q.thres_p2, p->thres_range = q.thres_range;
    p->corner_method = q.corner_method, p->min_size = q.min_size, p->max_size = q.max_size, p->warp_size = q.warp_size;
    p->border_dist = q.border_dist, p->use_locked_corners = q.use_locked_corners;
    p->approx_inner_product = q.approx_inner_product;
}
void orc_set_params(void* h, const orc_params_t* p) {
    Params& q = ((Detector*)h)->prm;
    q.thres_method = p->thres_method, q.thres_p1 = p->thres_p1, q.thres_p2 = p->thres_p2, q.thres_range = p->thres_range;
    q.corner_method = p->corner_method, q.min_size = p->min_size, q.max_size = p->max_size, q.warp_size = p->warp_size;
    q.border_dist = p->border_dist, q.use_locked_corners = p->use_locked_corners;
    q.approx_inner_product = p->approx_inner_product;
}

int orc_detect(void* h, const uint8_t* gray, int w, int hh, int stride, const float* K, const float* dist, int ndist,
               float marker_size, int y_perp, orc_marker_t* out, int cap, int* n_out) {
    std::vector<Marker> ms;
    int rc = ((Detector*)h)->detect(gray, w, hh, stride, K, dist, ndist, marker_size, y_perp, ms);
    *n_out = (int)ms.size();
    for (int i = 0; i < (int)ms.size() && i < cap; i++) to_c(ms[i], out + i);
    return rc;
}

void orc_get_thresholded(void* h, uint8_t* dst) {
    Detector* d = (Detector*)h;
    std::memcpy(dst, d->thres.data(), d->thres.size());
}

// contours of the (middle) threshold image, RETR_LIST order
int orc_num_contours(void* h) { return (int)((Detector*)h)->contours.size(); }
int orc_contour_info(void* h, int i, int* hole, int* trig_x, int* trig_y) {
    const Contour& c = ((Detector*)h)->contours[i];
    *hole = c.hole, *trig_x = c.trig_x, *trig_y = c.trig_y;
    return (int)c.pts.size();
}
void orc_contour_points(void* h, int i, int32_t* xy) {
    const Contour& c = ((Detector*)h)->contours[i];
    for (size_t k = 0; k < c.pts.size(); k++) xy[2 * k] = c.pts[k].x, xy[2 * k + 1] = c.pts[k].y;
}

// candidates after detectRectangles + identify: quad0 = integer corners in candidate order, id, nrot, contour idx/len
int orc_num_candidates(void* h) { return (int)((Detector*)h)->candidates.size(); }
void orc_candidate(void* h, int i, float* quad0, float* quad, int* id, int* nrot, int* idx, int* ncontour) {
    const Candidate& c = ((Detector*)h)->candidates[i];
    for (int k = 0; k < 4; k++) {
        quad0[2 * k] = c.c0[k].x, quad0[2 * k + 1] = c.c0[k].y;
        quad[2 * k] = c.c[k].x, quad[2 * k + 1] = c.c[k].y;
    }
    *id = c.id, *nrot = c.nrot, *idx = c.idx, *ncontour = (int)c.contour.size();
}
void orc_candidate_contour(void* h, int i, int32_t* xy) {
    const Candidate& c = ((Detector*)h)->candidates[i];
    for (size_t k = 0; k < c.contour.size(); k++) xy[2 * k] = c.contour[k].x, xy[2 * k + 1] = c.contour[k].y;
}
int orc_num_rejected(void* h) { return (int)((Detector*)h)->rejected.size(); }
void orc_rejected(void* h, int i, float* quad) {
    const Candidate& c = ((Detector*)h)->rejected[i];
    for (int k = 0; k < 4; k++) quad[2 * k] = c.c[k].x, quad[2 * k + 1] = c.c[k].y;
}

// ---- stage-level entry points
void orc_bgr2gray(const uint8_t* bgr, int npix, uint8_t* gray) { bgr2gray(bgr, npix, gray); }
void orc_adaptive_threshold(const uint8_t* src, int w, int h, int stride, int block, double C, uint8_t* dst) {
    adaptive_threshold_mean_inv(src, w, h, stride, block, C, dst);
}
// standalone findContours: returns an opaque set
void* orc_find_contours(const uint8_t* bin, int w, int h) {
    auto* v = new std::vector<Contour>();
    find_contours_list(bin, w, h, *v);
    return v;
}
int orc_cset_size(void* s) { return (int)((std::vector<Contour>*)s)->size(); }
int orc_cset_info(void* s, int i, int* hole, int* trig_x, int* trig_y) {
    const Contour& c = (*(std::vector<Contour>*)s)[i];
    *hole = c.hole, *trig_x = c.trig_x, *trig_y = c.trig_y;
    return (int)c.pts.size();
}
void orc_cset_points(void* s, int i, int32_t* xy) {
    const Contour& c = (*(std::vector<Contour>*)s)[i];
    for (size_t k = 0; k < c.pts.size(); k++) xy[2 * k] = c.pts[k].x, xy[2 * k + 1] = c.pts[k].y;
}
void orc_cset_free(void* s) { delete (std::vector<Contour>*)s; }

int orc_approx_poly(const int32_t* xy, int n, double eps, int inner_product_rule, int32_t* out_xy, int cap) {
    std::vector<Pt> src(n), dst;
    for (int i = 0; i < n; i++) src[i] = Pt{xy[2 * i], xy[2 * i + 1]};
    approx_poly_dp_closed(src, eps, dst, inner_product_rule);
    for (int i = 0; i < (int)dst.size() && i < cap; i++) out_xy[2 * i] = dst[i].x, out_xy[2 * i + 1] = dst[i].y;
    return (int)dst.size();
}
int orc_is_convex(const int32_t* xy, int n) {
    std::vector<Pt> p(n);
    for (int i = 0; i < n; i++) p[i] = Pt{xy[2 * i], xy[2 * i + 1]};
    return is_contour_convex(p) ? 1 : 0;
}
void orc_warp(const uint8_t* gray, int w, int h, int stride, const float* quad, int size, uint8_t* dst) {
    Pt2f src[4], d[4] = {{0, 0}, {(float)(size - 1), 0}, {(float)(size - 1), (float)(size - 1)}, {0, (float)(size - 1)}};
    for (int k = 0; k < 4; k++) src[k] = Pt2f{quad[2 * k], quad[2 * k + 1]};
    double M[9];
    perspective_transform(src, d, M);
    warp_perspective_nearest(gray, w, h, stride, M, size, dst);
}
int orc_otsu(const uint8_t* img, int n) { return otsu_threshold(img, n); }
int orc_fiducial_detect(const uint8_t* patch, int size, int* nrot) {
    std::vector<uint8_t> p(patch, patch + size * size);
    return fiducial_detect(p.data(), size, nrot);
}
int orc_solve_pnp(const float* obj_xyz, const float* img_xy, int n, const float* K, const float* dist, int ndist,
                  double* rvec, double* tvec) {
    std::vector<Pt3f> o(n);
    std::vector<Pt2f> m(n);
    for (int i = 0; i < n; i++) o[i] = Pt3f{obj_xyz[3 * i], obj_xyz[3 * i + 1], obj_xyz[3 * i + 2]}, m[i] = Pt2f{img_xy[2 * i], img_xy[2 * i + 1]};
    return solve_pnp_iterative(o.data(), m.data(), n, K, dist, ndist, rvec, tvec) ? 1 : 0;
}
void orc_rotate_x_axis(double* rvec) { rotate_x_axis(rvec); }
void orc_corner_subpix(const uint8_t* gray, int w, int h, int stride, float* xy, int n, int win, int max_iter, double eps) {
    corner_subpix(gray, w, h, stride, (Pt2f*)xy, n, win, max_iter, eps);
}
void orc_corner_harris(const uint8_t* gray, int w, int h, int stride, float* xy, int n) {
    corner_harris_refine(gray, w, h, stride, (Pt2f*)xy, n);
}
void orc_undistort(const uint8_t* src, int w, int h, int stride, int cn, const float* K, const float* dist, int ndist, uint8_t* dst) {
    undistort_8u(src, w, h, (size_t)stride, cn, K, dist, ndist, dst);
}
void orc_canny(const uint8_t* src, int w, int h, int stride, int low, int high, uint8_t* dst) { canny_3x3_l1(src, w, h, stride, low, high, dst); }
void orc_find_corner_maxima(const uint8_t* gray, int w, int h, int stride, float* xy, int n, int wsize) {
    find_corner_maxima(gray, w, h, stride, (Pt2f*)xy, n, wsize);
}
int orc_corner_harris_window(const uint8_t* gray, int w, int h, int stride, int x0, int y0, int x1, int y1, float* out) {
    std::vector<float> v;
    corner_harris_window(gray, w, h, stride, x0, y0, x1, y1, v);
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return (int)v.size();
}
// MarkerDetector::refineCandidateLines (src/markerdetector.cpp:931-997) on a caller-supplied contour and corners
void orc_refine_lines(const int32_t* xy, int n, float* corners8, const float* K, const float* dist, int ndist) {
    Candidate c;
    c.contour.resize(n);
    for (int i = 0; i < n; i++) c.contour[i] = Pt{xy[2 * i], xy[2 * i + 1]};
    for (int k = 0; k < 4; k++) c.c[k] = Pt2f{corners8[2 * k], corners8[2 * k + 1]};
    refine_lines(c, K, dist, ndist);
    for (int k = 0; k < 4; k++) corners8[2 * k] = c.c[k].x, corners8[2 * k + 1] = c.c[k].y;
}
float orc_board_detect(const orc_marker_t* ms, int n, const int32_t* ids, const float* obj, int nboard, int info_type,
                       const float* K, const float* dist, int ndist, float marker_size, float repj_thres, int y_perp,
                       orc_marker_t* out_ms, int* n_out, double* rvec, double* tvec, int* has_pose) {
    std::vector<Marker> det(n);
    for (int i = 0; i < n; i++) det[i] = from_c(ms[i]);
    BoardConf bc;
    bc.info_type = info_type;
    bc.ids.assign(ids, ids + nboard);
    bc.obj.resize((size_t)nboard * 4);
    for (int i = 0; i < nboard * 4; i++) bc.obj[i] = Pt3f{obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]};
    Board b;
    float p = board_detect(det, bc, K, dist, ndist, marker_size, repj_thres, y_perp, b);
    *n_out = (int)b.markers.size();
    for (size_t i = 0; i < b.markers.size(); i++) to_c(b.markers[i], out_ms + i);
    for (int k = 0; k < 3; k++) rvec[k] = b.rvec[k], tvec[k] = b.tvec[k];
    *has_pose = b.has_pose;
    return p;
}
}
