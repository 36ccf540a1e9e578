// Mock of <opencv2/core.hpp> for ONE purpose: to put the ARUCOHIP_HAVE_OPENCV branch of include/aruco_hip_shim.hpp through a
// compiler in a container that has no OpenCV (tests/test_cabi_cpu.py). It is the builder's own and declares only the shapes of the
// OpenCV 3/4 API the shim touches — cv::Mat with a MatStep-like `step`, Mat::zeros returning an expression type, Mat_<T>, Point2f/3f,
// Size, the four-argument-plus-line cv::Exception, _InputArray with getMat(), InputArray / OutputArray as references to proxy classes
// (NOT to cv::Mat, which is what makes the real signature different from the shim's own stand-in), the type macros. It pins nothing
// about OpenCV's behaviour: bodies are the minimum that links.
#ifndef MOCK_OPENCV_CORE_HPP
#define MOCK_OPENCV_CORE_HPP
#include <cstddef>
#include <cstring>
#include <exception>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

typedef unsigned char uchar;   // OpenCV's cvdef.h defines it at global scope

namespace cv {
typedef std::string String;
using ::uchar;

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;
template <class T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<float> Point3f;
template <class T> std::ostream& operator<<(std::ostream& s, const Point_<T>& p) { return s << "[" << p.x << ", " << p.y << "]"; }
template <class T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_& o) const { return !(*this == o); }
};
typedef Size_<int> Size;
template <class T> struct Scalar_ {   // cv::Scalar: four values, the colour argument of the drawing calls
    T val[4];
    Scalar_() { val[0] = val[1] = val[2] = val[3] = 0; }
    Scalar_(T v0, T v1 = 0, T v2 = 0, T v3 = 0) { val[0] = v0, val[1] = v1, val[2] = v2, val[3] = v3; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Scalar_<double> Scalar;

class Exception : public std::exception {
public:
    Exception() : code(0), line(0) {}
    Exception(int _code, const String& _err, const String& _func, const String& _file, int _line) : msg(_err), code(_code), err(_err), func(_func), file(_file), line(_line) {}
    virtual ~Exception() throw() {}
    virtual const char* what() const throw() { return msg.c_str(); }
    String msg;
    int code;
    String err, func, file;
    int line;
};

struct MatStep {   // cv::Mat::step is not a size_t
    size_t p0;
    MatStep() : p0(0) {}
    MatStep(size_t s) : p0(s) {}
    operator size_t() const { return p0; }
    MatStep& operator=(size_t s) { p0 = s; return *this; }
};

class Mat;
class MatExpr {   // what Mat::zeros really returns
public:
    int rows, cols, type;
    MatExpr(int r, int c, int t) : rows(r), cols(c), type(t) {}
    operator Mat() const;
};

class Mat {
public:
    int flags, dims, rows, cols;
    uchar* data;
    MatStep step;
    Mat() : flags(0), dims(0), rows(0), cols(0), data(0) {}
    Mat(int r, int c, int type) : flags(0), dims(2), rows(0), cols(0), data(0) { create(r, c, type); }
    Mat(Size sz, int type) : flags(0), dims(2), rows(0), cols(0), data(0) { create(sz.height, sz.width, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0 /* AUTO_STEP */) : flags(type), dims(2), rows(r), cols(c), data((uchar*)ext) {
        step = step_ ? step_ : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        flags = type, dims = 2, rows = r, cols = c;
        step = (size_t)c * elemSize();
        store_ = std::make_shared<std::vector<uchar> >((size_t)r * (size_t)step, (uchar)0);
        data = store_->data();
    }
    void create(Size sz, int type) { create(sz.height, sz.width, type); }
    static MatExpr zeros(int r, int c, int type) { return MatExpr(r, c, type); }
    int type() const { return flags & 0xFFF; }
    int channels() const { return ((flags & 0xFFF) >> 3) + 1; }
    int depth() const { return flags & 7; }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    bool isContinuous() const { return (size_t)step == (size_t)cols * elemSize(); }
    size_t total() const { return (size_t)rows * cols; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_8U ? 1 : depth() == CV_32F ? 4 : 8); }
    Size size() const { return Size(cols, rows); }
    template <class T> T& at(int r, int c) { return *(T*)(data + (size_t)r * (size_t)step + (size_t)c * sizeof(T)); }
    template <class T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * (size_t)step + (size_t)c * sizeof(T)); }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * (size_t)step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * (size_t)step); }
    Mat clone() const {
        Mat m(rows, cols, type());
        for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * (size_t)m.step, data + (size_t)r * (size_t)step, (size_t)cols * elemSize());
        return m;
    }
    void copyTo(Mat& m) const { m = clone(); }
protected:
    std::shared_ptr<std::vector<uchar> > store_;
};
inline MatExpr::operator Mat() const { return Mat(rows, cols, type); }
inline std::ostream& operator<<(std::ostream& s, const Mat& m) {   // OpenCV prints "[a, b; c, d]"
    s << "[";
    for (int r = 0; r < m.rows; r++)
        for (int c = 0; c < m.cols; c++) {
            if (m.depth() == CV_64F) s << m.at<double>(r, c);
            else if (m.depth() == CV_32F) s << m.at<float>(r, c);
            else s << (int)m.at<uchar>(r, c);
            s << (c + 1 < m.cols ? ", " : (r + 1 < m.rows ? ";\n " : ""));
        }
    return s << "]";
}

template <class T> struct DataType;
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<double> { enum { type = CV_64FC1 }; };
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };
template <class T> class Mat_ : public Mat {
public:
    Mat_() { flags = DataType<T>::type; }
    Mat_(int r, int c) : Mat(r, c, DataType<T>::type) {}
    Mat_(const Mat& m) : Mat(m) {}
    T& operator()(int r, int c) { return at<T>(r, c); }
    const T& operator()(int r, int c) const { return at<T>(r, c); }
    T& operator()(int i) { return ((T*)data)[i]; }
    const T& operator()(int i) const { return ((const T*)data)[i]; }
};

class UMat {};

// the proxy classes behind InputArray / OutputArray
class _InputArray {
public:
    _InputArray() : m_(0), v_(0), vn_(0) {}
    _InputArray(const Mat& m) : m_(&m), v_(0), vn_(0) {}
    template <class T> _InputArray(const Mat_<T>& m) : m_(&m), v_(0), vn_(0) {}
    _InputArray(const std::vector<uchar>& v) : m_(0), v_(v.data()), vn_(v.size()) {}
    Mat getMat(int = -1) const { return m_ ? *m_ : Mat(1, (int)vn_, CV_8UC1, (void*)v_); }
    bool empty() const { return m_ ? m_->empty() : vn_ == 0; }
protected:
    const Mat* m_;
    const uchar* v_;
    size_t vn_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m), out_(&m) {}
    void create(int r, int c, int type) const { out_->create(r, c, type); }
    Mat getMat(int = -1) const { return *out_; }
    Mat& getMatRef(int = -1) const { return *out_; }
protected:
    Mat* out_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
#endif
