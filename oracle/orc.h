// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's marker-detection hot path
// (aruco::MarkerDetector::detect + aruco::BoardDetector::detect, /root/reference/src) and of the
// OpenCV 3.0 primitives that path calls (OpenCV is a third-party dependency of the reference,
// unpinned ">= 2.4.9", goldens date from OpenCV 3.0.x; its sources are NOT under /root/reference,
// so each primitive below restates the published algorithm and is pinned end-to-end on the
// reference's golden vectors: tests/golden/{single,board,chessboard}.json).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code, and only as
// the checker / baseline. The product path (aruco_amd/csrc, include/arucohip.h) never links or calls it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orc {

struct Pt { int x, y; };
struct Pt2f { float x, y; };
struct Pt3f { float x, y, z; };

// One border as cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) returns it.
struct Contour {
    std::vector<Pt> pts;
    int hole;        // 1 = hole border
    int trig_x, trig_y;  // raster position of the scan transition that started it
};

// Mirrors the private members of aruco::MarkerDetector (src/markerdetector.cpp:235-249 defaults).
struct Params {
    int thres_method = 1;       // 0 FIXED_THRES, 1 ADPT_THRES, 2 CANNY
    double thres_p1 = 7, thres_p2 = 7;
    int thres_range = 0;        // _thresParam1_range
    int corner_method = 3;      // 0 NONE, 1 HARRIS, 2 SUBPIX, 3 LINES
    float min_size = 0.04f, max_size = 0.5f;
    int warp_size = 56;
    float border_dist = 0.025f;
    int use_locked_corners = 0; // findCornerMaxima before HARRIS / SUBPIX
    // OpenCV-version knobs (see DESIGN.md "oracle pinning")
    int approx_inner_product = 1;  // approxPolyDP clean-up also requires successive inner product >= 0
};

struct Candidate {
    Pt2f c[4];
    Pt2f c0[4];              // corners as detectRectangles produced them (before refinement / rotation)
    int idx;                 // index in the RETR_LIST contour list
    std::vector<Pt> contour;
    int id = -1;
    int nrot = 0;
};

struct Marker {
    int id;
    Pt2f c[4];
    float ssize;
    int has_pose;
    double rvec[3], tvec[3];
};

struct Board {
    std::vector<Marker> markers;
    int has_pose;
    double rvec[3], tvec[3];
    float prob;
};

struct BoardConf {
    int info_type;  // 0 PIX, 1 METERS, -1 NONE
    std::vector<int> ids;
    std::vector<Pt3f> obj;  // 4 per id
};

// ---- imgproc restatements (orc_imgproc.cpp)
void bgr2gray(const uint8_t* bgr, int npix, uint8_t* gray);
void adaptive_threshold_mean_inv(const uint8_t* src, int w, int h, int stride, int block, double C, uint8_t* dst);
void fixed_threshold_inv(const uint8_t* src, int w, int h, int stride, double thr, uint8_t* dst);
// contours in RETR_LIST order (reverse discovery order)
void find_contours_list(const uint8_t* bin, int w, int h, std::vector<Contour>& out);
void approx_poly_dp_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst, int inner_product_rule);
bool is_contour_convex(const std::vector<Pt>& p);
void perspective_transform(const Pt2f src[4], const Pt2f dst[4], double M[9]);
void warp_perspective_nearest(const uint8_t* src, int w, int h, int stride, const double M[9], int size, uint8_t* dst);
int otsu_threshold(const uint8_t* img, int n);
void get_rect_subpix_8u32f(const uint8_t* src, int w, int h, int stride, int pw, int ph, float cx, float cy, float* dst);
void get_rect_subpix_8u8u(const uint8_t* src, int w, int h, int stride, int pw, int ph, float cx, float cy, uint8_t* dst);

// ---- calib3d restatements (orc_pnp.cpp)
void rodrigues_to_mat(const double r[3], double R[9], double dRdr[27] /*nullable, [9][3]*/);
void rodrigues_to_vec(const double R[9], double r[3]);
// undistortPoints(src, K, dist, R=I, P): P=nullptr -> normalised coords
void undistort_points(const Pt2f* src, int n, const float K[9], const float* dist, int ndist, const float* P, Pt2f* dst);
void undistort_points_d(const double* src_xy, int n, const double K[9], const double k[8], double* dst_xy);
void project_points(const double* obj_xyz, int n, const double r[3], const double t[3], const double K[9],
                    const double k[8], double* img_xy, double* dpdr /*nullable 2n x 3*/, double* dpdt);
bool solve_pnp_iterative(const Pt3f* obj, const Pt2f* img, int n, const float K[9], const float* dist, int ndist,
                         double rvec[3], double tvec[3]);
void rotate_x_axis(double rvec[3]);

// Dictionary of highly reliable markers (aruco::Dictionary, src/highlyreliablemarkers.h:170-190): n x n codes, bit y*n+x of
// codes[i] = cell (y, x) of marker i in rotation 0.
struct HrmDict {
    int n = 0, tau0 = 0;
    float rate = 1.f;                 // correctionDistanceRate of HighlyReliableMarkers::loadDictionary
    std::vector<uint64_t> codes;
};

// ---- detection pipeline (orc_detect.cpp)
struct Detector {
    Params prm;
    HrmDict hrm;                                // hrm.n > 0: markerIdDetectorFunc = HighlyReliableMarkers::detect
    // results retained like the reference's members
    std::vector<uint8_t> thres;                 // getThresholdedImage()
    std::vector<Candidate> rejected;            // getCandidates()
    // stage outputs kept for stage-level parity tests
    std::vector<Contour> contours;              // of the middle threshold image
    std::vector<Candidate> candidates;          // after detectRectangles, with id/nrot after identify
    int w = 0, h = 0;
    int detect(const uint8_t* gray, int w, int h, int stride, const float* K, const float* dist, int ndist,
               float marker_size, int y_perp, std::vector<Marker>& out);
    void detect_rectangles(const std::vector<std::vector<uint8_t>>& thr, int w, int h, std::vector<Candidate>& out);
};
int fiducial_decode(const uint8_t* patch, int size, int* nrot);  // patch is Otsu-binarised in place by the caller
int fiducial_detect(uint8_t* patch, int size, int* nrot);
int hrm_detect(uint8_t* patch, int size, const HrmDict& d, int* nrot);
void refine_lines(Candidate& cand, const float* K, const float* dist, int ndist);
void corner_subpix(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n, int win, int max_iter, double eps);
void corner_harris_refine(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n);
// orc_extra.cpp
void canny_3x3_l1(const uint8_t* src, int w, int h, int stride, int low, int high, uint8_t* dst);
void undistort_8u(const uint8_t* src, int w, int h, size_t stride, int cn, const float K[9], const float* dist, int ndist, uint8_t* dst);
void find_corner_maxima(const uint8_t* gray, int w, int h, int stride, Pt2f* corners, int n, int wsize);
void corner_harris_window(const uint8_t* gray, int w, int h, int stride, int x0, int y0, int x1, int y1, std::vector<float>& harr);
float board_detect(const std::vector<Marker>& detected, const BoardConf& bc, const float* K, const float* dist,
                   int ndist, float marker_size, float repj_err_thres, int y_perp, Board& out);

}  // namespace orc
