// Host-only use of the shim's OpenGL / Ogre conversions (row f4): prints the matrices for tests/test_gl_conversion.py.
#include <cstdio>

#include "aruco_hip_shim.hpp"

int main() {
    aruco::Marker m;
    m.Rvec = cv::Mat_<double>(3, 1), m.Tvec = cv::Mat_<double>(3, 1);
    m.Rvec(0) = 0.1, m.Rvec(1) = -0.2, m.Rvec(2) = 0.3, m.Tvec(0) = 1, m.Tvec(1) = 2, m.Tvec(2) = 3;
    double mv[16], pos[3], q[4], pr[16], po[16];
    m.glGetModelViewMatrix(mv);
    m.OgreGetPoseParameters(pos, q);
    float K[9] = {600, 0, 320, 0, 610, 240, 0, 0, 1};
    float d[4] = {0, 0, 0, 0};
    aruco::CameraParameters cp(K, d, 4, cv::Size(640, 480));
    cp.glGetProjectionMatrix(cv::Size(640, 480), cv::Size(640, 480), pr, 0.5, 10);
    cp.OgreGetProjectionMatrix(cv::Size(640, 480), cv::Size(640, 480), po, 0.5, 10, true);
    for (int i = 0; i < 16; i++) std::printf("%.17g ", mv[i]);
    std::printf("\n%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", pos[0], pos[1], pos[2], q[0], q[1], q[2], q[3]);
    for (int i = 0; i < 16; i++) std::printf("%.17g ", pr[i]);
    std::printf("\n");
    for (int i = 0; i < 16; i++) std::printf("%.17g ", po[i]);
    std::printf("\n");
    aruco::Board b;   // no pose: the reference asserts, the shim throws
    try {
        b.glGetModelViewMatrix(mv);
        std::printf("nothrow\n");
    } catch (const std::exception&) {
        std::printf("throws\n");
    }
    return 0;
}
