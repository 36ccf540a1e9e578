// No-op definitions of the drawing members the reference's apps call (Marker::draw src/marker.cpp:54-81, CvDrawingUtils
// src/cvdrawingutils.cpp): test infrastructure for the link step of tests/test_cabi_cpu.py::test_reference_apps_compile_against_the_shim.
#include "cvdrawingutils.h"
namespace aruco {
void Marker::draw(cv::Mat&, cv::Scalar, int, bool) const {}
void CvDrawingUtils::draw3dAxis(cv::Mat&, Marker&, const CameraParameters&) {}
void CvDrawingUtils::draw3dCube(cv::Mat&, Marker&, const CameraParameters&, bool) {}
void CvDrawingUtils::draw3dAxis(cv::Mat&, Board&, const CameraParameters&) {}
void CvDrawingUtils::draw3dCube(cv::Mat&, Board&, const CameraParameters&, bool) {}
}  // namespace aruco
