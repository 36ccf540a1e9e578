// Compile-only caller of the shim's ARUCOHIP_HAVE_OPENCV branch against tests/cpp/mock_opencv (tests/test_cabi_cpu.py): the calls a
// reference application makes with real OpenCV types — cv::Mat, cv::Mat_<float>, a std::vector<uchar> through cv::InputArray,
// cv::Mat::zeros expressions as default-like arguments, cv::Exception with OpenCV's five-argument constructor.
#include <opencv2/core.hpp>

#include <cstdio>

#include "aruco_hip_shim.hpp"

#if !ARUCOHIP_HAVE_OPENCV
#error "the OpenCV branch of the shim was not selected"
#endif

static int my_decoder(const cv::Mat& in, int& nRotations) {
    nRotations = 0;
    return in.empty() ? -1 : 7;
}

int main(int argc, char**) {
    if (argc < 2) {
        std::fprintf(stderr, "compile-only check of the shim's OpenCV branch (mock headers)\n");
        return 0;
    }
    try {
        aruco::MarkerDetector det;
        std::vector<aruco::Marker> markers;
        cv::Mat gray(480, 640, CV_8UC1);
        cv::Mat_<float> K(3, 3);
        cv::Mat dist = cv::Mat::zeros(1, 4, CV_32FC1);
        det.detect(gray, markers, K, dist, 0.05f, false);
        std::vector<uchar> raw(640 * 480);
        det.detect(raw, markers);                          // anything an InputArray takes
        aruco::CameraParameters cp;
        det.detect(gray, markers, cp, 0.05f);
        det.setMakerDetectorFunction(&my_decoder);
        det.setThresholdParams(7, 7);
        const cv::Mat& th = det.getThresholdedImage();
        cv::Mat out;
        det.thresHold(aruco::MarkerDetector::ADPT_THRES, gray, out);
        std::vector<std::vector<cv::Point2f> > cands;
        det.detectRectangles(th, cands);
        aruco::BoardDetector bd;
        aruco::BoardConfiguration bc;
        aruco::Board b;
        bd.detect(markers, bc, b, K, dist, 0.05f);
        for (size_t i = 0; i < markers.size(); i++) markers[i].calculateExtrinsics(0.05f, K, dist);
    } catch (const cv::Exception& e) {
        std::fprintf(stderr, "%s (%s:%d)\n", e.what(), e.file.c_str(), e.line);
        return 2;
    }
    return 0;
}
