// Header-only C++ shim: the reference's classes for the detection path, implemented on the arucohip C ABI.
//
//   aruco::MarkerDetector   /root/reference/src/markerdetector.h:42-310
//   aruco::Marker           /root/reference/src/marker.h:43-140
//   aruco::CameraParameters /root/reference/src/cameraparameters.h:36-127 (data, resize, readFromXMLFile, GL / Ogre projection)
//   aruco::BoardConfiguration, aruco::Board  /root/reference/src/board.h:54-137 (data, readFromFile, GL / Ogre pose)
//   aruco::BoardDetector    /root/reference/src/boarddetector.h:40-148
//   aruco::Dictionary, aruco::MarkerCode, aruco::HighlyReliableMarkers  /root/reference/src/highlyreliablemarkers.h:50-260
//                           (dictionary files, loadDictionary, the decoder token for setMakerDetectorFunction)
//
// Same member names, argument meaning and failure behaviour (CV_Assert -> cv::Exception) as the reference, so a caller
// of the reference compiles against this header and links libarucohip.so instead of libaruco + OpenCV imgproc/calib3d.
// When <opencv2/core.hpp> is on the include path the real cv:: types are used; otherwise a minimal stand-in with the
// same spelling (cv::Mat view, Point2f, Point3f, Size, Mat_<T>, Exception) keeps the header self-contained — that is
// the configuration tested in this repository (OpenCV is not installed in the build image).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <cstdlib>
#include <iterator>
#include <fstream>
#include <type_traits>
#include <vector>

#include "arucohip.h"

#if defined(ARUCOHIP_USE_OPENCV) || (defined(__has_include) && __has_include(<opencv2/core.hpp>) && !defined(ARUCOHIP_NO_OPENCV))
#include <opencv2/core.hpp>
#define ARUCOHIP_HAVE_OPENCV 1
#else
#define ARUCOHIP_HAVE_OPENCV 0
namespace cv {
enum { CV_8UC1_ = 0, CV_8UC3_ = 16, CV_32FC1_ = 5, CV_64FC1_ = 6 };
#ifndef CV_8UC1
#define CV_8UC1 cv::CV_8UC1_
#define CV_8UC3 cv::CV_8UC3_
#define CV_32FC1 cv::CV_32FC1_
#define CV_64FC1 cv::CV_64FC1_
#endif
struct Point2f {
    float x, y;
    Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {}
};
struct Point3f {
    float x, y, z;
    Point3f(float x_ = 0, float y_ = 0, float z_ = 0) : x(x_), y(y_), z(z_) {}
};
struct Point {   // cv::Point = Point_<int>: contour points (MarkerCandidate::contour)
    int x, y;
    Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {}
};
struct Size {
    int width, height;
    Size(int w = 0, int h = 0) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};
inline std::ostream& operator<<(std::ostream& s, const Point2f& p) { return s << "[" << p.x << ", " << p.y << "]"; }
class Exception : public std::runtime_error {
public:
    int code;
    Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
// Dense 2-D array: owns its storage or views caller memory (like a cv::Mat header over external data).
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), data((unsigned char*)ext), type_(type) {
        step = step_ ? step_ : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        type_ = type, rows = r, cols = c;
        step = (size_t)c * elemSize();
        store_ = std::make_shared<std::vector<unsigned char> >((size_t)r * step, (unsigned char)0);
        data = store_->data();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    size_t elemSize() const { return type_ == CV_8UC1_ ? 1 : type_ == CV_8UC3_ ? 3 : type_ == CV_32FC1_ ? 4 : 8; }
    Size size() const { return Size(cols, rows); }
    template <class T> T& at(int r, int c) { return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }
protected:
    int type_ = CV_8UC1_;
    std::shared_ptr<std::vector<unsigned char> > store_;
};
template <class T> struct MatDepth_;
template <> struct MatDepth_<float> { enum { value = CV_32FC1_ }; };
template <> struct MatDepth_<double> { enum { value = CV_64FC1_ }; };
template <> struct MatDepth_<unsigned char> { enum { value = CV_8UC1_ }; };
template <class T> class Mat_ : public Mat {
public:
    Mat_() { type_ = MatDepth_<T>::value; }
    Mat_(int r, int c) : Mat(r, c, MatDepth_<T>::value) {}
    T& operator()(int r, int c) { return at<T>(r, c); }
    const T& operator()(int r, int c) const { return at<T>(r, c); }
    T& operator()(int i) { return ((T*)data)[i]; }
    const T& operator()(int i) const { return ((const T*)data)[i]; }
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}  // namespace cv
#endif

namespace aruco {

inline void arucohip_throw_(int rc, const char* what, arucohip_handle* h) {
    if (rc == ARUCOHIP_OK) return;
    std::string msg = std::string(what) + ": " + (h ? arucohip_last_error_string(h) : "") + " (arucohip status " + std::to_string(rc) + ")";
    throw cv::Exception(rc == ARUCOHIP_E_INVALID ? -215 /* StsAssert */ : -2, msg
#if ARUCOHIP_HAVE_OPENCV
                        , "arucohip", __FILE__, __LINE__
#endif
    );
}

// ---- small conversions between the caller's matrices and the ABI's plain arrays
inline bool mat_to_K_(const cv::Mat& m, float K[9]) {
    if (m.empty()) return false;
    if (m.rows != 3 || m.cols != 3) arucohip_throw_(ARUCOHIP_E_INVALID, "camera matrix must be 3x3", nullptr);
    for (int i = 0; i < 9; i++)
        K[i] = m.type() == CV_64FC1 ? (float)m.at<double>(i / 3, i % 3) : m.at<float>(i / 3, i % 3);
    return true;
}
inline int mat_to_dist_(const cv::Mat& m, float d[8]) {
    if (m.empty()) return 0;
    int n = (int)std::min<size_t>(m.total(), 8);
    for (int i = 0; i < n; i++) {
        int r = m.cols == 1 ? i : 0, c = m.cols == 1 ? 0 : i;
        d[i] = m.type() == CV_64FC1 ? (float)m.at<double>(r, c) : m.at<float>(r, c);
    }
    return n;
}

// ---- minimal reader for the OpenCV FileStorage YAML files the reference ships and writes (camera intrinsics
// src/cameraparameters.cpp:187-222, board configurations src/serialization.cpp:94-120): top-level scalars, !!opencv-matrix
// blocks and the flow-style marker list. Not a general YAML parser.
namespace yml_ {
inline std::string slurp(const std::string& path) {
    std::ifstream f(path.c_str());
    if (!f) arucohip_throw_(ARUCOHIP_E_INVALID, ("cannot open " + path).c_str(), nullptr);
    std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return all;
}
// position right after "key:" where key starts a line (top level); npos if absent
inline size_t top_key(const std::string& t, const std::string& key, size_t from = 0) {
    for (size_t p = t.find(key + ":", from); p != std::string::npos; p = t.find(key + ":", p + 1))
        if (p == 0 || t[p - 1] == '\n') return p + key.size() + 1;
    return std::string::npos;
}
inline bool scalar(const std::string& t, const std::string& key, double* v) {
    size_t p = top_key(t, key);
    if (p == std::string::npos) return false;
    *v = std::strtod(t.c_str() + p, nullptr);
    return true;
}
// numbers of the next [...] group(s) starting at p: reads until `count` numbers were found or the text ends
inline std::vector<double> numbers(const std::string& t, size_t p, size_t count, size_t* endp = nullptr) {
    std::vector<double> out;
    while (p < t.size() && out.size() < count) {
        const char c = t[p];
        if ((c >= '0' && c <= '9') || c == '-' || c == '+' || (c == '.' && p + 1 < t.size() && t[p + 1] >= '0' && t[p + 1] <= '9')) {
            char* e = nullptr;
            out.push_back(std::strtod(t.c_str() + p, &e));
            p = (size_t)(e - t.c_str());
        } else {
            p++;
        }
    }
    if (endp) *endp = p;
    return out;
}
// !!opencv-matrix block: rows, cols and rows*cols values of data: [...]
inline bool matrix(const std::string& t, const std::string& key, int* rows, int* cols, std::vector<double>* data) {
    size_t p = top_key(t, key);
    if (p == std::string::npos) return false;
    size_t r = t.find("rows:", p), c = t.find("cols:", p), d = t.find("data:", p);
    if (r == std::string::npos || c == std::string::npos || d == std::string::npos) return false;
    *rows = (int)std::strtol(t.c_str() + r + 5, nullptr, 10);
    *cols = (int)std::strtol(t.c_str() + c + 5, nullptr, 10);
    if (*rows <= 0 || *cols <= 0) return false;
    *data = numbers(t, d + 5, (size_t)*rows * *cols);
    return data->size() == (size_t)*rows * *cols;
}
}  // namespace yml_

class CameraParameters {
public:
    cv::Mat_<float> CameraMatrix;  // 3x3 (fx 0 cx, 0 fy cy, 0 0 1)
    cv::Mat_<float> Distorsion;    // k1,k2,p1,p2[,k3]
    cv::Size CamSize;
    CameraParameters() : CamSize(-1, -1) {}
    CameraParameters(const float K[9], const float* dist, int ndist, cv::Size size) { setParams(K, dist, ndist, size); }
    void setParams(const float K[9], const float* dist, int ndist, cv::Size size) {
        CameraMatrix = cv::Mat_<float>(3, 3);
        for (int i = 0; i < 9; i++) CameraMatrix(i / 3, i % 3) = K[i];
        Distorsion = cv::Mat_<float>(1, ndist);
        for (int i = 0; i < ndist; i++) Distorsion(0, i) = dist[i];
        CamSize = size;
    }
    bool isValid() const { return !CameraMatrix.empty() && !Distorsion.empty() && CamSize.width != -1 && CamSize.height != -1; }
    // cameraparameters.cpp:187-222: image_width / image_height, camera_matrix -> float, the first 5 distortion coefficients -> float
    void readFromXMLFile(const std::string& filePath) {
        const std::string t = yml_::slurp(filePath);
        double w = -1, h = -1;
        yml_::scalar(t, "image_width", &w), yml_::scalar(t, "image_height", &h);
        int kr = 0, kc = 0, dr = 0, dc = 0;
        std::vector<double> K, D;
        if (!yml_::matrix(t, "camera_matrix", &kr, &kc, &K) || K.size() != 9)
            arucohip_throw_(ARUCOHIP_E_INVALID, ("File :" + filePath + " does not contains valid camera matrix").c_str(), nullptr);
        if (w == -1 || h == 0) arucohip_throw_(ARUCOHIP_E_INVALID, ("File :" + filePath + " does not contains valid camera dimensions").c_str(), nullptr);
        if (!yml_::matrix(t, "distortion_coefficients", &dr, &dc, &D) || D.size() < 4)
            arucohip_throw_(ARUCOHIP_E_INVALID, ("File :" + filePath + " does not contains valid distortion_coefficients").c_str(), nullptr);
        float Kf[9], Df[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 9; i++) Kf[i] = (float)K[i];
        for (size_t i = 0; i < 5 && i < D.size(); i++) Df[i] = (float)D[i];
        setParams(Kf, Df, 5, cv::Size((int)w, (int)h));
    }
    // cameraparameters.cpp:166-179
    void resize(cv::Size size) {
        if (!isValid()) arucohip_throw_(ARUCOHIP_E_INVALID, "invalid camera parameters", nullptr);
        if (size == CamSize) return;
        float AxFactor = float(size.width) / float(CamSize.width);
        float AyFactor = float(size.height) / float(CamSize.height);
        CameraMatrix(0, 0) *= AxFactor;
        CameraMatrix(0, 2) *= AxFactor;
        CameraMatrix(1, 1) *= AyFactor;
        CameraMatrix(1, 2) *= AyFactor;
    }
    // cameraparameters.cpp:226-266 (like the reference it first resizes *this to `size`); orgImgSize is unused there too
    void glGetProjectionMatrix(cv::Size /*orgImgSize*/, cv::Size size, double proj_matrix[16], double gnear, double gfar, bool invert = false) {
        if (!isValid()) arucohip_throw_(ARUCOHIP_E_INVALID, "invalid camera parameters", nullptr);
        float K[9];
        for (int i = 0; i < 9; i++) K[i] = CameraMatrix(i / 3, i % 3);
        arucohip_throw_(arucohip_gl_projection(K, CamSize.width, CamSize.height, size.width, size.height, gnear, gfar, invert, proj_matrix),
                        "glGetProjectionMatrix", nullptr);
        resize(size);   // the reference's call leaves the matrix resized (and CamSize as it was)
    }
    // cameraparameters.cpp:271-295
    void OgreGetProjectionMatrix(cv::Size orgImgSize, cv::Size size, double proj_matrix[16], double gnear, double gfar, bool invert = false) {
        double t[16];
        glGetProjectionMatrix(orgImgSize, size, t, gnear, gfar, invert);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) proj_matrix[r * 4 + c] = (c == 3 ? 1.0 : -1.0) * t[c * 4 + r];
    }
};

// utils.cpp:32-147 behind the C ABI: shared by Marker and Board
inline void arucohip_gl_modelview_(const cv::Mat_<double>& Rvec, const cv::Mat_<double>& Tvec, double modelview_matrix[16]) {
    if (Rvec.empty() || Tvec.empty()) arucohip_throw_(ARUCOHIP_E_INVALID, "extrinsic parameters are not set", nullptr);
    const double r[3] = {Rvec(0), Rvec(1), Rvec(2)}, t[3] = {Tvec(0), Tvec(1), Tvec(2)};
    arucohip_throw_(arucohip_gl_modelview(r, t, modelview_matrix), "glGetModelViewMatrix", nullptr);
}
inline void arucohip_ogre_pose_(const cv::Mat_<double>& Rvec, const cv::Mat_<double>& Tvec, double position[3], double orientation[4]) {
    if (Rvec.empty() || Tvec.empty()) arucohip_throw_(ARUCOHIP_E_INVALID, "extrinsic parameters are not set", nullptr);
    const double r[3] = {Rvec(0), Rvec(1), Rvec(2)}, t[3] = {Tvec(0), Tvec(1), Tvec(2)};
    arucohip_throw_(arucohip_ogre_pose(r, t, position, orientation), "OgreGetPoseParameters", nullptr);
}

// Process-wide handle for calls that have no detector object to hang on (Marker::calculateExtrinsics): created on first
// use on device ARUCOHIP_DEVICE (default 0), shared under a mutex (a handle is not re-entrant), never destroyed — the HIP
// runtime may already be gone when static destructors run.
struct SharedHandle_ {
    std::mutex mu;
    arucohip_handle* h = nullptr;
    static SharedHandle_& get() {
        static SharedHandle_* s = new SharedHandle_();
        return *s;
    }
    int w = 0, hh = 0;
    arucohip_handle* ensure(int width = 64, int height = 64) {   // call with mu held
        if (!h || width > w || height > hh) {
            arucohip_destroy(h);
            h = nullptr;
            w = std::max(std::max(w, width), 32), hh = std::max(std::max(hh, height), 32);
            const char* e = std::getenv("ARUCOHIP_DEVICE");
            arucohip_throw_(arucohip_create(nullptr, e ? std::atoi(e) : 0, w, hh, 1, &h), "arucohip_create", nullptr);
        }
        return h;
    }
};

// cv::undistort(src, dst, cameraMatrix, distCoeffs) on the device: what the reference's GL apps run on
// every frame before detect() (utils/aruco_test_gl.cpp:237-240). 8-bit frames, 1 or 3 channels.
inline void undistort(const cv::Mat& src, cv::Mat& dst, const cv::Mat& cameraMatrix, const cv::Mat& distCoeffs) {
    if (src.type() != CV_8UC1 && src.type() != CV_8UC3) arucohip_throw_(ARUCOHIP_E_INVALID, "undistort: CV_8UC1 or CV_8UC3 frames", nullptr);
    float K[9], d[8];
    if (!mat_to_K_(cameraMatrix, K)) arucohip_throw_(ARUCOHIP_E_INVALID, "undistort: camera matrix is empty", nullptr);
    const int nd = mat_to_dist_(distCoeffs, d);
    const int cn = src.type() == CV_8UC3 ? 3 : 1;
    cv::Mat out(src.rows, src.cols, src.type());
    SharedHandle_& sh = SharedHandle_::get();
    std::lock_guard<std::mutex> lock(sh.mu);
    arucohip_handle* h = sh.ensure(src.cols, src.rows);
    arucohip_throw_(arucohip_undistort(h, src.data, 1, src.cols, src.rows, src.step, (size_t)src.rows * src.step, cn, 0, K, nd ? d : nullptr, nd, out.data, 0),
                    "undistort", h);
    dst = out;
}

class Marker : public std::vector<cv::Point2f> {
public:
    int id;
    float ssize;
    cv::Mat_<double> Rvec, Tvec;
    Marker() : id(-1), ssize(-1) {}
    explicit Marker(const std::vector<cv::Point2f>& corners, int _id = -1) : std::vector<cv::Point2f>(corners), id(_id), ssize(-1) {}
    bool isValid() const { return id != -1 && size() == 4; }
    cv::Point2f getCenter() const {
        cv::Point2f cent(0, 0);
        for (size_t i = 0; i < size(); i++) cent.x += (*this)[i].x, cent.y += (*this)[i].y;
        cent.x /= float(size());
        cent.y /= float(size());
        return cent;
    }
    float getPerimeter() const {  // utils.h:39-46
        float sum = 0;
        for (size_t i = 0; i < size(); i++) {
            size_t j = (i + 1) % size();
            float dx = (*this)[i].x - (*this)[j].x, dy = (*this)[i].y - (*this)[j].y;
            sum += std::sqrt((double)dx * dx + (double)dy * dy);
        }
        return sum;
    }
    float getArea() const {  // marker.cpp:141-151
        assert(size() == 4);
        const Marker& m = *this;
        float a1 = std::fabs((m[1].x - m[0].x) * (m[3].y - m[0].y) - (m[1].y - m[0].y) * (m[3].x - m[0].x));
        float a2 = std::fabs((m[1].x - m[2].x) * (m[3].y - m[2].y) - (m[1].y - m[2].y) * (m[3].x - m[2].x));
        return (a2 + a1) / 2.f;
    }
    // marker.h:77 / marker.cpp:85-90
    void calculateExtrinsics(float markerSize, const CameraParameters& CP, bool setYPerpendicular = true) {
        if (!CP.isValid()) arucohip_throw_(ARUCOHIP_E_INVALID, "!CP.isValid(): invalid camera parameters. It is not possible to calculate extrinsics", nullptr);
        calculateExtrinsics(markerSize, CP.CameraMatrix, CP.Distorsion, setYPerpendicular);
    }
    // marker.h:85 / marker.cpp:112-124: solvePnP of the 4 corners against the marker's own square (device kernel behind
    // arucohip_calculate_extrinsics, on the process-wide handle). MarkerDetector::calculateExtrinsics is the batched form.
    void calculateExtrinsics(float markerSize, cv::Mat CameraMatrix, cv::Mat Distorsion = cv::Mat(), bool setYPerpendicular = true) {
        if (!(markerSize > 0 && isValid())) arucohip_throw_(ARUCOHIP_E_INVALID, "invalid marker. It is not possible to calculate extrinsics", nullptr);
        float K[9], d[8];
        if (!mat_to_K_(CameraMatrix, K)) arucohip_throw_(ARUCOHIP_E_INVALID, "CameraMatrix is empty", nullptr);
        const int nd = mat_to_dist_(Distorsion, d);
        arucohip_marker_t m;
        to_abi(&m);
        SharedHandle_& sh = SharedHandle_::get();
        std::lock_guard<std::mutex> lock(sh.mu);
        arucohip_handle* h = sh.ensure();
        arucohip_throw_(arucohip_calculate_extrinsics(h, &m, 1, K, nd ? d : nullptr, nd, markerSize, setYPerpendicular ? 1 : 0), "Marker::calculateExtrinsics", h);
        Rvec = cv::Mat_<double>(3, 1), Tvec = cv::Mat_<double>(3, 1);
        for (int k = 0; k < 3; k++) Rvec(k) = m.rvec[k], Tvec(k) = m.tvec[k];
        ssize = markerSize;
    }
    void glGetModelViewMatrix(double modelview_matrix[16]) const { arucohip_gl_modelview_(Rvec, Tvec, modelview_matrix); }              // marker.h:90
    void OgreGetPoseParameters(double position[3], double orientation[4]) const { arucohip_ogre_pose_(Rvec, Tvec, position, orientation); }  // marker.h:104
#if ARUCOHIP_HAVE_OPENCV
    // marker.h:69. Drawing is outside the detection path: the member is DECLARED so that the reference's callers compile
    // (utils/aruco_simple.cpp:82); its definition stays the reference's own cv::line / cv::putText code (src/marker.cpp:54-81), which needs
    // nothing but OpenCV imgproc and this class - INTEGRATION.md "drawing".
    void draw(cv::Mat& in, cv::Scalar color, int lineWidth = 1, bool writeId = true) const;
#endif
    friend bool operator<(const Marker& a, const Marker& b) { return a.id < b.id; }
    friend std::ostream& operator<<(std::ostream& str, const Marker& M) {  // marker.h:128-139
        str << M.id << "=";
        for (int i = 0; i < 4 && i < (int)M.size(); i++) str << "(" << M[i].x << "," << M[i].y << ") ";
        if (!M.Tvec.empty() && !M.Rvec.empty()) {
            str << "Txyz=" << M.Tvec(0) << " " << M.Tvec(1) << " " << M.Tvec(2) << " ";
            str << "Rxyz=" << M.Rvec(0) << " " << M.Rvec(1) << " " << M.Rvec(2) << " ";
        }
        return str;
    }
    void to_abi(arucohip_marker_t* o) const {
        std::memset(o, 0, sizeof(*o));
        o->id = id, o->ssize = ssize;
        for (int k = 0; k < 4 && k < (int)size(); k++) o->corners[2 * k] = (*this)[k].x, o->corners[2 * k + 1] = (*this)[k].y;
        if (!Rvec.empty() && !Tvec.empty()) {
            o->has_pose = 1;
            for (int k = 0; k < 3; k++) o->rvec[k] = Rvec(k), o->tvec[k] = Tvec(k);
        }
    }
    static Marker from_abi(const arucohip_marker_t& m) {
        Marker r;
        r.id = m.id, r.ssize = m.ssize;
        for (int k = 0; k < 4; k++) r.push_back(cv::Point2f(m.corners[2 * k], m.corners[2 * k + 1]));
        if (m.has_pose) {
            r.Rvec = cv::Mat_<double>(3, 1), r.Tvec = cv::Mat_<double>(3, 1);
            for (int k = 0; k < 3; k++) r.Rvec(k) = m.rvec[k], r.Tvec(k) = m.tvec[k];
        }
        return r;
    }
};

class BoardConfiguration {
public:
    std::vector<int> ids;
    std::vector<std::vector<cv::Point3f> > objPoints;
    enum MarkerInfoType { NONE = -1, PIX = 0, METERS = 1 };
    int mInfoType;
    BoardConfiguration() : mInfoType(NONE) {}
    // board.cpp:52-56 + src/serialization.cpp:94-120: aruco_bc_nmarkers, aruco_bc_mInfoType, aruco_bc_markers: - { id:.., corners:[ [x,y,z] x4 ] }
    void readFromFile(const std::string& sfile) {
        const std::string t = yml_::slurp(sfile);
        double nm = -1, it = -1;
        if (!yml_::scalar(t, "aruco_bc_nmarkers", &nm)) arucohip_throw_(ARUCOHIP_E_INVALID, "invalid file type", nullptr);
        yml_::scalar(t, "aruco_bc_mInfoType", &it);
        mInfoType = (int)it;
        ids.clear(), objPoints.clear();
        size_t p = yml_::top_key(t, "aruco_bc_markers");
        while (p != std::string::npos) {
            size_t q = t.find("id:", p);
            if (q == std::string::npos) break;
            ids.push_back((int)std::strtol(t.c_str() + q + 3, nullptr, 10));
            size_t c = t.find("corners:", q);
            if (c == std::string::npos) arucohip_throw_(ARUCOHIP_E_INVALID, "BoardConfiguration: marker without corners", nullptr);
            size_t end = c;
            std::vector<double> v = yml_::numbers(t, c + 8, 12, &end);
            if (v.size() != 12) arucohip_throw_(ARUCOHIP_E_INVALID, "BoardConfiguration: a marker needs 4 corners", nullptr);
            std::vector<cv::Point3f> pts(4);
            for (int k = 0; k < 4; k++) pts[k] = cv::Point3f((float)v[3 * k], (float)v[3 * k + 1], (float)v[3 * k + 2]);
            objPoints.push_back(pts);
            p = end;
        }
        if ((int)ids.size() != (int)nm) arucohip_throw_(ARUCOHIP_E_INVALID, "BoardConfiguration: aruco_bc_nmarkers does not match the marker list", nullptr);
    }
    bool isExpressedInMeters() const { return mInfoType == METERS; }
    bool isExpressedInPixels() const { return mInfoType == PIX; }
    const std::vector<cv::Point3f>& getMarkerInfo(int id) const {  // board.cpp:60-66
        for (size_t i = 0; i < ids.size(); i++)
            if (ids[i] == id) return objPoints[i];
        arucohip_throw_(ARUCOHIP_E_INVALID, "BoardConfiguration::getMarkerInfo: marker with the id given is not found", nullptr);
        return objPoints[0];
    }
};

class Board : public std::vector<Marker> {
public:
    BoardConfiguration conf;
    cv::Mat_<double> Rvec, Tvec;
    void glGetModelViewMatrix(double modelview_matrix[16]) const { arucohip_gl_modelview_(Rvec, Tvec, modelview_matrix); }              // board.h:109
    void OgreGetPoseParameters(double position[3], double orientation[4]) const { arucohip_ogre_pose_(Rvec, Tvec, position, orientation); }  // board.h:123
};

// ---- marker-id decoders (markerdetector.h:248 setMakerDetectorFunction). On the accelerated path the decoders run on the
// device; the static functions below are the tokens a caller passes to MarkerDetector::setMakerDetectorFunction.
class FiducidalMarkers {
public:
    // arucofidmarkers.h:84 — the default decoder (5x5 Hamming markers)
    static int detect(const cv::Mat&, int&) {
        arucohip_throw_(ARUCOHIP_E_UNSUPPORTED, "FiducidalMarkers::detect runs on the device; pass it to setMakerDetectorFunction", nullptr);
        return -1;
    }
};

// highlyreliablemarkers.h:50-140 (the part the detection path needs: size and bits)
class MarkerCode {
public:
    explicit MarkerCode(unsigned int n = 0) : _n(n), _bits(n * n, '0') {}
    void fromString(const std::string& s) { _bits = s; }
    std::string toString() const { return _bits; }
    unsigned int n() const { return _n; }
    unsigned int size() const { return _n * _n; }
    bool get(unsigned int pos) const { return _bits[pos] == '1'; }

private:
    unsigned int _n;
    std::string _bits;
};

// highlyreliablemarkers.h:170-190; fromFile reads the reference's dictionary files (src/serialization.cpp:123-150:
// nmarkers, markersize, tau0, marker_<i>: "<n*n bits>")
class Dictionary : public std::vector<MarkerCode> {
public:
    int tau0 = 0;
    bool fromFile(const std::string& filename) {
        std::ifstream f(filename.c_str());
        if (!f) arucohip_throw_(ARUCOHIP_E_INVALID, "Dictionary::fromFile: cannot open file", nullptr);
        clear();
        unsigned int n = 0;
        size_t nmarkers = 0;
        std::vector<std::pair<size_t, std::string> > codes;
        std::string line;
        while (std::getline(f, line)) {
            size_t c = line.find(':');
            if (c == std::string::npos || line[0] == '%') continue;
            std::string key = line.substr(0, c), val = line.substr(c + 1);
            key.erase(0, key.find_first_not_of(" \t"));
            val.erase(0, val.find_first_not_of(" \t\""));
            val.erase(val.find_last_not_of(" \t\"\r") + 1);
            if (key == "nmarkers") nmarkers = (size_t)std::atol(val.c_str());
            else if (key == "markersize") n = (unsigned int)std::atol(val.c_str());
            else if (key == "tau0") tau0 = std::atoi(val.c_str());
            else if (key.compare(0, 7, "marker_") == 0) codes.push_back(std::make_pair((size_t)std::atol(key.c_str() + 7), val));
        }
        if (n == 0 || codes.size() != nmarkers) arucohip_throw_(ARUCOHIP_E_INVALID, "Dictionary::fromFile: malformed dictionary", nullptr);
        resize(nmarkers, MarkerCode(n));
        for (size_t i = 0; i < codes.size(); i++) {
            if (codes[i].first >= nmarkers || codes[i].second.size() != (size_t)n * n)
                arucohip_throw_(ARUCOHIP_E_INVALID, "Dictionary::fromFile: malformed marker", nullptr);
            (*this)[codes[i].first].fromString(codes[i].second);
        }
        return true;
    }
};

// highlyreliablemarkers.h:196-260: static dictionary + decoder token
class HighlyReliableMarkers {
public:
    static bool loadDictionary(Dictionary D, float correctionDistanceRate = 1) {  // highlyreliablemarkers.cpp:311-322
        if (D.size() == 0) return false;
        State_& s = state_();
        s.D = D, s.rate = correctionDistanceRate, s.version++;
        return true;
    }
    static bool loadDictionary(const std::string& filename, float correctionDistance = 1) {
        Dictionary D;
        D.fromFile(filename);
        return loadDictionary(D, correctionDistance);
    }
    static int detect(const cv::Mat&, int&) {
        arucohip_throw_(ARUCOHIP_E_UNSUPPORTED, "HighlyReliableMarkers::detect runs on the device; pass it to setMakerDetectorFunction", nullptr);
        return -1;
    }
    struct State_ {
        Dictionary D;
        float rate = 1;
        int version = 0;
    };
    static State_& state_() {
        static State_ s;
        return s;
    }
};

class MarkerDetector {
public:
    // markerdetector.h:45-62: a candidate to be a marker = the marker (corners, id) + its contour + its position in the contour list.
    // (Private in the reference although refineCandidateLines, a public member, takes one; public here so that a caller can build one.)
    class MarkerCandidate : public Marker {
    public:
        MarkerCandidate() : idx(-1) {}
        MarkerCandidate(const Marker& M) : Marker(M), idx(-1) {}
        std::vector<cv::Point> contour;   // all the points of its contour
        int idx;                          // index position in the global contour list
    };
    enum ThresholdMethods { FIXED_THRES, ADPT_THRES, CANNY };
    enum CornerRefinementMethod { NONE, HARRIS, SUBPIX, LINES };

    explicit MarkerDetector(int device = 0) : h_(nullptr), device_(device), cap_w_(0), cap_h_(0) { arucohip_default_params(&p_); }
    ~MarkerDetector() { arucohip_destroy(h_); }
    MarkerDetector(const MarkerDetector&) = delete;
    MarkerDetector& operator=(const MarkerDetector&) = delete;

    // markerdetector.h:102-103
    void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, cv::Mat camMatrix = cv::Mat(), cv::Mat distCoeff = cv::Mat(),
                float markerSizeMeters = -1, bool setYPerpendicular = false) {
        // markerdetector.cpp:303-310: 8-bit gray frames are used as they are, 3-channel frames are BGR and converted (on the device)
        if (input.type() != CV_8UC1 && input.type() != CV_8UC3)
            arucohip_throw_(ARUCOHIP_E_INVALID, "detect: 8-bit gray (CV_8UC1) or BGR (CV_8UC3) frames", nullptr);
        ensure_(input.cols, input.rows);
        if (hrm_ && hrm_version_ != HighlyReliableMarkers::state_().version) apply_decoder_();
        float K[9], d[8];
        bool hasK = mat_to_K_(camMatrix, K);
        int nd = mat_to_dist_(distCoeff, d);
        std::vector<arucohip_marker_t>& out = out_;   // the detector's own result buffer: a frame loop allocates nothing per call
        if (out.size() < 256) out.resize(256);
        int n = 0, rc;
        for (int attempt = 0;; attempt++) {
            rc = input.type() == CV_8UC3
                     ? arucohip_detect_bgr(h_, input.data, input.cols, input.rows, input.step, hasK ? K : nullptr, nd ? d : nullptr, nd, markerSizeMeters,
                                           setYPerpendicular ? 1 : 0, out.data(), (int)out.size(), &n)
                     : arucohip_detect(h_, input.data, input.cols, input.rows, input.step, hasK ? K : nullptr, nd ? d : nullptr, nd, markerSizeMeters,
                                       setYPerpendicular ? 1 : 0, out.data(), (int)out.size(), &n);
            // a cluttered frame can outgrow the device lists; the reference has no such limit, so the lists are doubled and the
            // frame is detected again instead of failing
            if (rc != ARUCOHIP_E_OVERFLOW || attempt >= 5) break;
            grow_++;
            recreate_();
        }
        arucohip_throw_(rc, "MarkerDetector::detect", h_);
        detectedMarkers.clear();
        for (int i = 0; i < n; i++) detectedMarkers.push_back(Marker::from_abi(out[i]));
        frame_size_ = cv::Size(input.cols, input.rows);
        thres_valid_ = false, cand_valid_ = false;
    }
    // markerdetector.h:116-120
    void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, const CameraParameters& camParams, float markerSizeMeters = -1,
                bool setYPerpendicular = false) {
        detect(input, detectedMarkers, camParams.CameraMatrix, camParams.Distorsion, markerSizeMeters, setYPerpendicular);
    }

#if ARUCOHIP_HAVE_OPENCV
    // With the real OpenCV the reference takes cv::InputArray (markerdetector.h:102): anything that is not already a cv::Mat
    // (std::vector<uchar>, cv::Mat_, cv::UMat ...) comes through getMat(). This branch cannot be compiled in the build container
    // (no OpenCV there): DESIGN.md lists it as untested.
    template <class A, class = typename std::enable_if<!std::is_convertible<const A&, const cv::Mat&>::value>::type>
    void detect(const A& input, std::vector<Marker>& detectedMarkers, cv::Mat camMatrix = cv::Mat(), cv::Mat distCoeff = cv::Mat(),
                float markerSizeMeters = -1, bool setYPerpendicular = false) {
        detect(cv::_InputArray(input).getMat(), detectedMarkers, camMatrix, distCoeff, markerSizeMeters, setYPerpendicular);
    }
#endif

    // markerdetector.h:78
    typedef int (*MarkerdetectorFunc)(const cv::Mat& in, int& nRotations);
    // markerdetector.h:243-245. The library's two decoders select the device kernels; any other function is called on the
    // host with the canonical patch the device warped (contract: markerdetector.h:65-77), the pipeline continues on the device.
    void setMakerDetectorFunction(MarkerdetectorFunc markerdetector_func) {
        // a null function changes nothing: the detector keeps the decoder it had (the check comes before any member is touched)
        if (!markerdetector_func) arucohip_throw_(ARUCOHIP_E_INVALID, "setMakerDetectorFunction: null function", nullptr);
        user_fn_ = nullptr, hrm_ = false;
        if (markerdetector_func == &HighlyReliableMarkers::detect)
            hrm_ = true;
        else if (markerdetector_func != &FiducidalMarkers::detect)
            user_fn_ = markerdetector_func;
        p_.decoder_kind = user_fn_ ? ARUCOHIP_DECODER_USER : hrm_ ? ARUCOHIP_DECODER_HRM : ARUCOHIP_DECODER_FIDUCIAL_5X5;   // part of the parameter set from now on
        hrm_version_ = -1;
        if (h_) apply_decoder_();
    }
    void setThresholdMethod(ThresholdMethods m) { p_.thres_method = (int)m, push_(); }
    ThresholdMethods getThresholdMethod() const { return (ThresholdMethods)p_.thres_method; }
    void setThresholdParams(double param1, double param2) { p_.thres_param1 = param1, p_.thres_param2 = param2, push_(); }
    void setThresholdParamRange(size_t r1 = 0, size_t = 0) { p_.thres_param1_range = (int)r1, recreate_(); }
    void getThresholdParams(double& param1, double& param2) const { param1 = p_.thres_param1, param2 = p_.thres_param2; }
    void enableLockedCornersMethod(bool enable) {  // markerdetector.cpp:291-295
        p_.use_locked_corners = enable;
        if (enable) p_.corner_method = SUBPIX;
        push_();
    }
    const cv::Mat& getThresholdedImage() {  // markerdetector.h:183
        if (!thres_valid_ && h_ && frame_size_.width > 0) {
            thres_ = cv::Mat(frame_size_.height, frame_size_.width, CV_8UC1);
            arucohip_throw_(arucohip_get_thresholded(h_, 0, thres_.data), "getThresholdedImage", h_);
            thres_valid_ = true;
        }
        return thres_;
    }
    void setCornerRefinementMethod(CornerRefinementMethod m) { p_.corner_method = (int)m, push_(); }
    CornerRefinementMethod getCornerRefinementMethod() const { return (CornerRefinementMethod)p_.corner_method; }
    void setMinMaxSize(float min = 0.03f, float max = 0.5f) {  // CV_Assert ranges: markerdetector.cpp:1031-1038
        arucohip_params_t q = p_;
        q.min_size = min, q.max_size = max;
        check_(q);
        p_ = q, push_();
    }
    void getMinMaxSize(float& min, float& max) { min = p_.min_size, max = p_.max_size; }
    void setDesiredSpeed(int val) {  // markerdetector.cpp:265-285
        if (val < 0) val = 0;
        else if (val > 3) val = 2;
        speed_ = val;
        if (val == 0) p_.warp_size = 56, p_.corner_method = SUBPIX;
        else if (val == 1 || val == 2) p_.warp_size = 28, p_.corner_method = NONE;
        push_();
    }
    int getDesiredSpeed() const { return speed_; }
    void setWarpSize(int val) {  // CV_Assert(val >= 10): markerdetector.cpp:1047-1051
        arucohip_params_t q = p_;
        q.warp_size = val;
        check_(q);
        p_ = q, push_();
    }
    int getWarpSize() const { return p_.warp_size; }
    const std::vector<std::vector<cv::Point2f> >& getCandidates() {  // markerdetector.h:266
        if (!cand_valid_ && h_) {
            std::vector<float> q(512 * 8);
            int n = 0;
            arucohip_throw_(arucohip_get_candidates(h_, 0, q.data(), 512, &n), "getCandidates", h_);
            candidates_.assign(n, std::vector<cv::Point2f>(4));
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 4; k++) candidates_[i][k] = cv::Point2f(q[i * 8 + 2 * k], q[i * 8 + 2 * k + 1]);
            cand_valid_ = true;
        }
        return candidates_;
    }
    // stage entry points (markerdetector.h:255-280)
    void thresHold(int method, const cv::Mat& grey, cv::Mat& thresImg, double param1 = -1, double param2 = -1) {
        if (grey.type() != CV_8UC1) arucohip_throw_(ARUCOHIP_E_INVALID, "thresHold: grey.type() == CV_8UC1", nullptr);  // :644
        ensure_(grey.cols, grey.rows);
        thresImg = cv::Mat(grey.rows, grey.cols, CV_8UC1);
        arucohip_throw_(arucohip_threshold(h_, method, grey.data, grey.cols, grey.rows, grey.step, param1, param2, thresImg.data), "thresHold", h_);
    }
    void detectRectangles(const cv::Mat& thresImg, std::vector<std::vector<cv::Point2f> >& candidates) {
        ensure_(thresImg.cols, thresImg.rows);
        std::vector<float> q(512 * 8);
        int n = 0;
        arucohip_throw_(arucohip_detect_rectangles(h_, thresImg.data, thresImg.cols, thresImg.rows, thresImg.step, q.data(), 512, &n), "detectRectangles", h_);
        candidates.assign(n, std::vector<cv::Point2f>(4));
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) candidates[i][k] = cv::Point2f(q[i * 8 + 2 * k], q[i * 8 + 2 * k + 1]);
    }
    void warp(const cv::Mat& in, cv::Mat& out, cv::Size size, std::vector<cv::Point2f> points) {
        if (points.size() != 4) arucohip_throw_(ARUCOHIP_E_INVALID, "warp: points.size() == 4", nullptr);  // :685
        ensure_(in.cols, in.rows);
        float q[8];
        for (int k = 0; k < 4; k++) q[2 * k] = points[k].x, q[2 * k + 1] = points[k].y;
        out = cv::Mat(size.height, size.width, CV_8UC1);
        arucohip_throw_(arucohip_warp(h_, in.data, in.cols, in.rows, in.step, q, size.width, out.data), "warp", h_);
    }
    // markerdetector.h:280 / markerdetector.cpp:931-997: the LINES corner refinement of one candidate, on the device
    // (arucohip_refine_candidate_lines): the four corners become the intersections of the least-squares lines through the contour's sides;
    // with both matrices non-empty the contour is undistorted first and the corners are distorted again.
    void refineCandidateLines(MarkerCandidate& candidate, const cv::Mat& camMatrix, const cv::Mat& distCoeff) {
        if (candidate.size() != 4 || candidate.contour.empty())
            arucohip_throw_(ARUCOHIP_E_INVALID, "refineCandidateLines: a candidate has 4 corners and a contour", nullptr);
        ensure_(std::max(cap_w_, 64), std::max(cap_h_, 64));
        float K[9], d[8], c[8];
        const bool hasK = mat_to_K_(camMatrix, K);
        const int nd = mat_to_dist_(distCoeff, d);
        const bool undist = hasK && nd > 0;   // :958 / :990: both matrices or neither
        std::vector<int32_t> xy(2 * candidate.contour.size());
        for (size_t i = 0; i < candidate.contour.size(); i++) xy[2 * i] = candidate.contour[i].x, xy[2 * i + 1] = candidate.contour[i].y;
        for (int k = 0; k < 4; k++) c[2 * k] = candidate[k].x, c[2 * k + 1] = candidate[k].y;
        arucohip_throw_(arucohip_refine_candidate_lines(h_, xy.data(), (int)candidate.contour.size(), c, undist ? K : nullptr, undist ? d : nullptr, undist ? nd : 0),
                        "refineCandidateLines", h_);
        for (int k = 0; k < 4; k++) candidate[k] = cv::Point2f(c[2 * k], c[2 * k + 1]);
    }
    // Marker::calculateExtrinsics for a whole vector at once (batched device solvePnP)
    void calculateExtrinsics(std::vector<Marker>& markers, float markerSize, cv::Mat camMatrix, cv::Mat distCoeff = cv::Mat(), bool setYPerpendicular = true) {
        if (markers.empty()) return;
        ensure_(std::max(cap_w_, 64), std::max(cap_h_, 64));
        float K[9], d[8];
        if (!mat_to_K_(camMatrix, K)) arucohip_throw_(ARUCOHIP_E_INVALID, "CameraMatrix is empty", nullptr);
        int nd = mat_to_dist_(distCoeff, d);
        std::vector<arucohip_marker_t> m(markers.size());
        for (size_t i = 0; i < markers.size(); i++) markers[i].to_abi(&m[i]);
        arucohip_throw_(arucohip_calculate_extrinsics(h_, m.data(), (int)m.size(), K, nd ? d : nullptr, nd, markerSize, setYPerpendicular), "calculateExtrinsics", h_);
        for (size_t i = 0; i < markers.size(); i++) markers[i] = Marker::from_abi(m[i]);
    }
    arucohip_handle* handle(int w = 640, int h = 480) {
        ensure_(std::max(w, cap_w_), std::max(h, cap_h_));
        return h_;
    }

private:
    void check_(const arucohip_params_t& q) {
        arucohip_handle* tmp = h_;
        if (!tmp) ensure_(640, 480), tmp = h_;
        arucohip_params_t saved;
        arucohip_get_params(tmp, &saved);
        int rc = arucohip_set_params(tmp, &q);
        if (rc != ARUCOHIP_OK) {
            arucohip_set_params(tmp, &saved);
            arucohip_throw_(rc, "MarkerDetector parameter", tmp);
        }
    }
    void push_() {
        if (h_) arucohip_throw_(arucohip_set_params(h_, &p_), "MarkerDetector parameter", h_);
    }
    void recreate_() {
        if (!h_) return;
        int w = cap_w_, hh = cap_h_;
        arucohip_destroy(h_);
        h_ = nullptr, cap_w_ = cap_h_ = 0;
        ensure_(w, hh);
    }
    void ensure_(int w, int hh) {
        if (h_ && w <= cap_w_ && hh <= cap_h_) return;   // device arrays are sized per dimension
        w = std::max(std::max(w, cap_w_), 32), hh = std::max(std::max(hh, cap_h_), 32);   // a handle is at least 32 x 32; frames may be smaller
        arucohip_destroy(h_);
        h_ = nullptr;
        arucohip_limits_t lim;
        arucohip_default_limits(&lim, w, hh, 1);
        lim.max_thres_planes = std::max(1, 2 * p_.thres_param1_range + 1);
        // the reference has no list limits: after a device list overflow detect() retries with larger lists (grow_)
        for (int g = 0; g < grow_; g++) {
            lim.triggers_per_frame = std::min(lim.triggers_per_frame * 2, 1 << 22);
            lim.contours_per_frame = std::min(lim.contours_per_frame * 2, 1 << 18);
            lim.points_per_frame = std::min(lim.points_per_frame * 2, 1 << 24);
            lim.long_walks_per_plane = std::min(lim.long_walks_per_plane * 2, 1 << 16);
        }
        lim.candidates_per_frame = 512, lim.markers_per_frame = 256;
        arucohip_throw_(arucohip_create_ex(&p_, device_, &lim, &h_), "arucohip_create", nullptr);
        cap_w_ = w, cap_h_ = hh;
        hrm_version_ = -1;
        if (hrm_ || user_fn_) apply_decoder_();
    }
    // HighlyReliableMarkers' static dictionary -> the handle (arucohip_set_dictionary), or back to the fiducial decoder
    void apply_decoder_() {
        if (hrm_) {
            const HighlyReliableMarkers::State_& s = HighlyReliableMarkers::state_();
            if (s.D.empty()) arucohip_throw_(ARUCOHIP_E_INVALID, "HighlyReliableMarkers: no dictionary loaded", nullptr);
            std::vector<uint64_t> codes(s.D.size(), 0);
            for (size_t i = 0; i < s.D.size(); i++)
                for (unsigned int b = 0; b < s.D[i].size(); b++)
                    if (s.D[i].get(b)) codes[i] |= 1ull << b;
            arucohip_throw_(arucohip_set_dictionary(h_, (int)s.D[0].n(), (int)codes.size(), codes.data(), s.D.tau0, s.rate), "loadDictionary", h_);
            hrm_version_ = s.version;
        }
        arucohip_throw_(arucohip_set_decoder_callback(h_, user_fn_ ? &user_trampoline_ : nullptr, reinterpret_cast<void*>(user_fn_)), "setMakerDetectorFunction", h_);
        push_();
    }
    // the C ABI's callback -> the reference's function type: a cv::Mat header over the patch, nRotations by reference
    static int user_trampoline_(void* user, uint8_t* patch, int size, int* n_rotations) {
        MarkerdetectorFunc fn = reinterpret_cast<MarkerdetectorFunc>(user);
        cv::Mat in(size, size, CV_8UC1, patch);
        int nrot = *n_rotations;
        const int id = fn(in, nrot);
        *n_rotations = nrot;
        return id;
    }
    MarkerdetectorFunc user_fn_ = nullptr;
    std::vector<arucohip_marker_t> out_;
    arucohip_handle* h_;
    int device_, cap_w_, cap_h_;
    int grow_ = 0;   // doublings of the device list limits after overflows
    arucohip_params_t p_;
    int speed_ = 0;
    bool hrm_ = false;
    int hrm_version_ = -1;
    cv::Size frame_size_;
    cv::Mat thres_;
    bool thres_valid_ = false, cand_valid_ = false;
    std::vector<std::vector<cv::Point2f> > candidates_;
};

// Frames sharded over the GPUs of the node behind one call (arucohip_mgpu_*). No reference counterpart: the
// reference's frame loop (utils/aruco_test.cpp:140-160) calls one MarkerDetector per frame; with this helper the loop hands
// over up to devices x framesPerDevice frames at once, frame f is detected on device f mod G and the markers come back in
// frame order.
class MultiGpuDetector {
public:
    MultiGpuDetector(const std::vector<int>& devices, int maxWidth, int maxHeight, int framesPerDevice, bool gatherOverXgmi = false) : m_(nullptr) {
        arucohip_throw_(arucohip_mgpu_create(nullptr, devices.empty() ? nullptr : devices.data(), devices.empty() ? arucohip_mgpu_device_count() : (int)devices.size(),
                                             maxWidth, maxHeight, framesPerDevice, 128, gatherOverXgmi ? ARUCOHIP_MGPU_GATHER_PEER : ARUCOHIP_MGPU_GATHER_HOST, &m_),
                        "arucohip_mgpu_create", nullptr);
    }
    ~MultiGpuDetector() { arucohip_mgpu_destroy(m_); }
    MultiGpuDetector(const MultiGpuDetector&) = delete;
    MultiGpuDetector& operator=(const MultiGpuDetector&) = delete;
    int devices() const { return arucohip_mgpu_size(m_); }
    // equally sized 8-bit gray frames; same camera arguments as MarkerDetector::detect
    void detect(const std::vector<cv::Mat>& frames, std::vector<std::vector<Marker> >& detectedMarkers, cv::Mat camMatrix = cv::Mat(),
                cv::Mat distCoeff = cv::Mat(), float markerSizeMeters = -1, bool setYPerpendicular = false) {
        detectedMarkers.assign(frames.size(), std::vector<Marker>());
        if (frames.empty()) return;
        const int w = frames[0].cols, h = frames[0].rows;
        std::vector<unsigned char> packed((size_t)frames.size() * w * h);
        for (size_t f = 0; f < frames.size(); f++) {
            if (frames[f].type() != CV_8UC1 || frames[f].cols != w || frames[f].rows != h)
                arucohip_throw_(ARUCOHIP_E_INVALID, "MultiGpuDetector::detect: equally sized CV_8UC1 frames", nullptr);
            for (int r = 0; r < h; r++) std::memcpy(packed.data() + (f * h + r) * (size_t)w, frames[f].data + (size_t)r * frames[f].step, (size_t)w);
        }
        float K[9], d[8];
        const bool hasK = mat_to_K_(camMatrix, K);
        const int nd = mat_to_dist_(distCoeff, d);
        std::vector<arucohip_marker_t> out(frames.size() * 128);
        std::vector<int32_t> n(frames.size(), 0);
        const int rc = arucohip_mgpu_detect_batch(m_, packed.data(), (int)frames.size(), w, h, (size_t)w, (size_t)w * h, hasK ? K : nullptr, nd ? d : nullptr, nd,
                                                  markerSizeMeters, setYPerpendicular ? 1 : 0, out.data(), 128, n.data());
        if (rc != ARUCOHIP_OK) throw cv::Exception(rc == ARUCOHIP_E_INVALID ? -215 : -2, std::string("MultiGpuDetector::detect: ") + arucohip_mgpu_last_error_string(m_)
#if ARUCOHIP_HAVE_OPENCV
                                                   , "arucohip", __FILE__, __LINE__
#endif
        );
        for (size_t f = 0; f < frames.size(); f++)
            for (int i = 0; i < n[f]; i++) detectedMarkers[f].push_back(Marker::from_abi(out[f * 128 + i]));
    }

private:
    arucohip_mgpu* m_;
};

class BoardDetector {
public:
    explicit BoardDetector(bool setYPerpendicular = false) : _setYPerpendicular(setYPerpendicular), _areParamsSet(false), _markerSize(-1), repj_err_thres(-1) {}
    void setParams(const BoardConfiguration& bc, const CameraParameters& cp, float markerSizeMeters = -1) {
        _camParams = cp, _markerSize = markerSizeMeters, _bconf = bc, _areParamsSet = true;
    }
    void setParams(const BoardConfiguration& bc) { _bconf = bc, _areParamsSet = true; }
    // boarddetector.cpp:66-77
    float detect(const cv::Mat& im) {
        _mdetector.detect(im, _vmarkers);
        if (_camParams.isValid())
            return detect(_vmarkers, _bconf, _boardDetected, _camParams.CameraMatrix, _camParams.Distorsion, _markerSize);
        return detect(_vmarkers, _bconf, _boardDetected);
    }
    float detect(const std::vector<Marker>& detectedMarkers, const BoardConfiguration& BConf, Board& Bdetected, const CameraParameters& cp,
                 float markerSizeMeters = -1) {
        return detect(detectedMarkers, BConf, Bdetected, cp.CameraMatrix, cp.Distorsion, markerSizeMeters);
    }
    // boarddetector.cpp:90-205
    float detect(const std::vector<Marker>& detectedMarkers, const BoardConfiguration& BConf, Board& Bdetected, cv::Mat camMatrix = cv::Mat(),
                 cv::Mat distCoeff = cv::Mat(), float markerSizeMeters = -1) {
        float K[9], d[8];
        bool hasK = mat_to_K_(camMatrix, K);
        int nd = mat_to_dist_(distCoeff, d);
        std::vector<arucohip_marker_t> in(std::max<size_t>(detectedMarkers.size(), 1)), out(std::max<size_t>(detectedMarkers.size(), 1));
        for (size_t i = 0; i < detectedMarkers.size(); i++) detectedMarkers[i].to_abi(&in[i]);
        std::vector<int32_t> ids(BConf.ids.begin(), BConf.ids.end());
        std::vector<float> obj;
        for (size_t i = 0; i < BConf.objPoints.size(); i++)
            for (int p = 0; p < 4 && p < (int)BConf.objPoints[i].size(); p++)
                obj.push_back(BConf.objPoints[i][p].x), obj.push_back(BConf.objPoints[i][p].y), obj.push_back(BConf.objPoints[i][p].z);
        arucohip_board_t b;
        float prob = 0;
        arucohip_handle* h = _mdetector.handle();
        int rc = arucohip_board_detect(h, in.data(), (int)detectedMarkers.size(), ids.data(), obj.data(), (int)ids.size(), BConf.mInfoType,
                                       hasK ? K : nullptr, nd ? d : nullptr, nd, markerSizeMeters, repj_err_thres, _setYPerpendicular ? 1 : 0,
                                       out.data(), &b, &prob);
        arucohip_throw_(rc, "BoardDetector::detect", h);
        Bdetected.clear();
        for (int i = 0; i < b.n_markers; i++) Bdetected.push_back(Marker::from_abi(out[i]));
        Bdetected.conf = BConf;
        if (b.has_pose) {
            Bdetected.Rvec = cv::Mat_<double>(3, 1), Bdetected.Tvec = cv::Mat_<double>(3, 1);
            for (int k = 0; k < 3; k++) Bdetected.Rvec(k) = b.rvec[k], Bdetected.Tvec(k) = b.tvec[k];
        }
        return prob;
    }
    static Board detect(const cv::Mat& Image, const BoardConfiguration& bc, const CameraParameters& cp, float markerSizeMeters = -1) {
        BoardDetector BD;
        BD.setParams(bc, cp, markerSizeMeters);
        BD.detect(Image);
        return BD.getDetectedBoard();
    }
    Board& getDetectedBoard() { return _boardDetected; }
    MarkerDetector& getMarkerDetector() { return _mdetector; }
    std::vector<Marker>& getDetectedMarkers() { return _vmarkers; }
    void setYPerpendicular(bool enable) { _setYPerpendicular = enable; }
    bool isYPerpendicular() { return _setYPerpendicular; }   // boarddetector.h:128
    void set_repj_err_thres(float Repj_err_thres) { repj_err_thres = Repj_err_thres; }
    float get_repj_err_thres() const { return repj_err_thres; }

private:
    bool _setYPerpendicular, _areParamsSet;
    BoardConfiguration _bconf;
    Board _boardDetected;
    float _markerSize;
    CameraParameters _camParams;
    MarkerDetector _mdetector;
    std::vector<Marker> _vmarkers;
    float repj_err_thres;
};

}  // namespace aruco

#if !ARUCOHIP_HAVE_OPENCV
namespace cv {
// the reference's apps spell it cv::undistort; without OpenCV the name resolves to the device implementation
inline void undistort(const Mat& src, Mat& dst, const Mat& cameraMatrix, const Mat& distCoeffs) { aruco::undistort(src, dst, cameraMatrix, distCoeffs); }
}  // namespace cv
#endif
