// Row f4 of SURVEY §8 — OpenGL / Ogre conversions of the pose results. Pure host arithmetic (no device work), part of the
// C ABI so that the AR consumers of the reference (utils/aruco_test_gl.cpp, aruco_test_board_gl.cpp) find the whole
// output side behind one library.
//
// Reference: GetGLModelViewMatrix /root/reference/src/utils.cpp:32-69 (Marker::glGetModelViewMatrix src/marker.h:90,
// Board::glGetModelViewMatrix src/board.h:109), GetOgrePoseParameters src/utils.cpp:71-147,
// CameraParameters::glGetProjectionMatrix src/cameraparameters.cpp:226-266 (after CameraParameters::resize :166-179),
// CameraParameters::OgreGetProjectionMatrix :271-295. Golden: testdata/board/expected_gl.yml (tests/golden/board_gl.json).
#include <float.h>
#include <math.h>

#include "../../include/arucohip.h"

namespace {

// cv::Rodrigues, vector -> matrix (double): R = cos(t) I + (1 - cos(t)) r r^T + sin(t) [r]x with r = v / |v|
void rodrigues_host(const double v[3], double R[9]) {
    const double theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double rx = v[0] * itheta, ry = v[1] * itheta, rz = v[2] * itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * rxm[i];
}

}  // namespace

extern "C" {

int arucohip_gl_modelview(const double* rvec, const double* tvec, double* m) {
    if (!rvec || !tvec || !m) return ARUCOHIP_E_INVALID;
    double R[9];
    rodrigues_host(rvec, R);
    // column-major 4x4; the third row (camera z) is negated: OpenGL looks down -z
    for (int col = 0; col < 3; col++) {
        m[0 + col * 4] = R[0 * 3 + col];
        m[1 + col * 4] = R[1 * 3 + col];
        m[2 + col * 4] = -R[2 * 3 + col];
        m[3 + col * 4] = 0.0;
    }
    m[12] = tvec[0], m[13] = tvec[1], m[14] = -tvec[2], m[15] = 1.0;
    return ARUCOHIP_OK;
}

int arucohip_gl_modelview_n(const arucohip_marker_t* markers, int n, double* modelview) {
    if (n < 0 || (n > 0 && (!markers || !modelview))) return ARUCOHIP_E_INVALID;
    for (int i = 0; i < n; i++) {
        if (!markers[i].has_pose) return ARUCOHIP_E_INVALID;   // "extrinsic parameters are not set" (utils.cpp:34-36)
        arucohip_gl_modelview(markers[i].rvec, markers[i].tvec, modelview + (size_t)i * 16);
    }
    return ARUCOHIP_OK;
}

int arucohip_ogre_pose(const double* rvec, const double* tvec, double* position, double* orientation) {
    if (!rvec || !tvec || !position || !orientation) return ARUCOHIP_E_INVALID;
    position[0] = -tvec[0], position[1] = -tvec[1], position[2] = +tvec[2];
    double R[9];
    rodrigues_host(rvec, R);
    // x and y axes of the marker in Ogre's frame, z from their cross product; `ax` holds them as columns
    const double x[3] = {-R[0], -R[3], +R[6]}, y[3] = {-R[1], -R[4], +R[7]};
    const double z[3] = {x[1] * y[2] - x[2] * y[1], -x[0] * y[2] + x[2] * y[0], x[0] * y[1] - x[1] * y[0]};
    const double ax[3][3] = {{x[0], y[0], z[0]}, {x[1], y[1], z[1]}, {x[2], y[2], z[2]}};
    // rotation matrix -> quaternion (w, x, y, z), Shoemake's branches as the reference takes them
    const double trace = ax[0][0] + ax[1][1] + ax[2][2];
    if (trace > 0.0) {
        double root = sqrt(trace + 1.0);
        orientation[0] = 0.5 * root;
        root = 0.5 / root;
        orientation[1] = (ax[2][1] - ax[1][2]) * root;
        orientation[2] = (ax[0][2] - ax[2][0]) * root;
        orientation[3] = (ax[1][0] - ax[0][1]) * root;
    } else {
        int i = 0;
        if (ax[1][1] > ax[0][0]) i = 1;
        if (ax[2][2] > ax[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double root = sqrt(ax[i][i] - ax[j][j] - ax[k][k] + 1.0);
        orientation[1 + i] = 0.5 * root;
        root = 0.5 / root;
        orientation[0] = (ax[k][j] - ax[j][k]) * root;
        orientation[1 + j] = (ax[j][i] + ax[i][j]) * root;
        orientation[1 + k] = (ax[k][i] + ax[i][k]) * root;
    }
    return ARUCOHIP_OK;
}

int arucohip_gl_projection(const float* K, int cam_width, int cam_height, int width, int height, double gnear, double gfar, int invert,
                           double* p) {
    if (!K || !p || cam_width <= 0 || cam_height <= 0 || width <= 0 || height <= 0) return ARUCOHIP_E_INVALID;
    // CameraParameters::resize: the float intrinsics are scaled by float factors
    float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    if (width != cam_width || height != cam_height) {
        const float ax = float(width) / float(cam_width), ay = float(height) / float(cam_height);
        fx *= ax, cx *= ax, fy *= ay, cy *= ay;
    }
    // like the reference, right / bottom use CamSize, which CameraParameters::resize leaves at the size the intrinsics were
    // given for (cameraparameters.cpp:166-179 scales the matrix only); identical when both sizes agree
    const double top = gnear * cy / fy;
    const double left = -gnear * cx / fx;
    const double right = gnear * (cam_width - cx) / fx;
    const double bottom = -gnear * (cam_height - cy) / fy;
    for (int i = 0; i < 16; i++) p[i] = 0.0;
    p[0] = (2.0 * gnear) / (right - left);
    p[5] = (2.0 * gnear) / (top - bottom);
    p[8] = (right + left) / (right - left);
    p[9] = -((top + bottom) / (top - bottom));
    p[10] = -((gfar + gnear) / (gfar - gnear));
    p[11] = -1.0;
    p[14] = -(2.0 * gnear * gfar) / (gfar - gnear);
    if (!invert) p[13] = -p[13], p[1] = -p[1], p[5] = -p[5], p[9] = -p[9];
    return ARUCOHIP_OK;
}

int arucohip_ogre_projection(const float* K, int cam_width, int cam_height, int width, int height, double gnear, double gfar, int invert,
                             double* p) {
    double t[16];
    const int rc = arucohip_gl_projection(K, cam_width, cam_height, width, height, gnear, gfar, invert, t);
    if (rc != ARUCOHIP_OK || !p) return rc != ARUCOHIP_OK ? rc : ARUCOHIP_E_INVALID;
    // transpose, signs flipped except for the last column
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) p[r * 4 + c] = (c == 3 ? 1.0 : -1.0) * t[c * 4 + r];
    return ARUCOHIP_OK;
}

}  // extern "C"
