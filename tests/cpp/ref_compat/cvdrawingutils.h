// Drawing is outside the detection path: a real integration keeps the reference's own src/cvdrawingutils.{h,cpp}, which
// need only OpenCV and the classes the shim provides. This test stand-in declares the same interface (reference src/cvdrawingutils.h:36-47)
// so that the reference's apps compile unchanged; tests/cpp/ref_compat/drawing_stubs.cpp defines it (and Marker::draw) as no-ops for the link.
#pragma once
#include "aruco_hip_shim.hpp"
namespace aruco {
class CvDrawingUtils {
public:
    static void draw3dAxis(cv::Mat& Image, Marker& m, const CameraParameters& CP);
    static void draw3dCube(cv::Mat& Image, Marker& m, const CameraParameters& CP, bool setYperpendicular = false);
    static void draw3dAxis(cv::Mat& Image, Board& m, const CameraParameters& CP);
    static void draw3dCube(cv::Mat& Image, Board& m, const CameraParameters& CP, bool setYperpendicular = false);
};
}  // namespace aruco
