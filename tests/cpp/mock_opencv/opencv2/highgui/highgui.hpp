// Mock of <opencv2/highgui/highgui.hpp> (builder's own, see ../core.hpp): the GUI / IO calls the reference's apps make around detect()
// (utils/aruco_simple.cpp:51-96, utils/aruco_simple_board.cpp:60-100) as shapes that compile and link. They do nothing: tests/test_cabi_cpu.py
// only proves that the reference's own callers compile, unchanged, against include/aruco_hip_shim.hpp.
#ifndef MOCK_OPENCV_HIGHGUI_HPP
#define MOCK_OPENCV_HIGHGUI_HPP
#include <string>

#include "../core.hpp"

namespace cv {
class VideoCapture {
public:
    VideoCapture() {}
    explicit VideoCapture(const String&) {}
    explicit VideoCapture(int) {}
    bool isOpened() const { return false; }
    bool grab() { return false; }
    bool retrieve(Mat&, int = 0) { return false; }
};
inline Mat imread(const String&, int = 1) { return Mat(); }
inline bool imwrite(const String&, const Mat&) { return false; }
inline void imshow(const String&, const Mat&) {}
inline void namedWindow(const String&, int = 1) {}
inline int waitKey(int = 0) { return -1; }
}  // namespace cv
#endif
