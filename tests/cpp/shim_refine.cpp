// MarkerDetector::MarkerCandidate + MarkerDetector::refineCandidateLines (reference src/markerdetector.h:45-62, :280; .cpp:931-997) through
// the shim, the way a caller of the reference would use the stage: detect() without corner refinement gives the integer quads, the
// candidate's contour comes from the library's stage inspection calls, refineCandidateLines moves the corners onto the sides' lines.
// Prints "refined <id> x0 y0 ... x3 y3" with the corners in the marker's canonical order (rotated by nRotations like :364-366).
//   shim_refine <image.pgm>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "aruco_hip_shim.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::string magic;
        int w, h, maxv;
        if (!(f >> magic >> w >> h >> maxv) || magic != "P5") return 1;
        f.get();
        cv::Mat gray(h, w, CV_8UC1);
        f.read((char*)gray.data, (std::streamsize)w * h);

        aruco::MarkerDetector det;
        det.setCornerRefinementMethod(aruco::MarkerDetector::NONE);
        std::vector<aruco::Marker> markers;
        det.detect(gray, markers);
        arucohip_handle* hd = det.handle();
        // candidates of the frame in detectRectangles order (integer quads after the orientation swap), ids, rotations
        float quads[256 * 8];
        int32_t ids[256], nrot[256];
        int ncand = 0, ncont = 0;
        if (arucohip_debug_candidates(hd, 0, quads, ids, nrot, 256, &ncand) != ARUCOHIP_OK) return 3;
        if (arucohip_debug_num_contours(hd, 0, &ncont) != ARUCOHIP_OK) return 3;
        std::vector<std::vector<cv::Point> > contours(ncont);
        for (int i = 0; i < ncont; i++) {
            int n = 0;
            arucohip_debug_contour(hd, 0, i, nullptr, nullptr, nullptr, nullptr, 0, &n);
            std::vector<int16_t> xy(2 * (size_t)n);
            if (arucohip_debug_contour(hd, 0, i, nullptr, nullptr, nullptr, xy.data(), n, &n) != ARUCOHIP_OK) return 3;
            for (int k = 0; k < n; k++) contours[i].push_back(cv::Point(xy[2 * k], xy[2 * k + 1]));
        }
        std::cout << std::setprecision(9);
        for (int c = 0; c < ncand; c++) {
            if (ids[c] < 0) continue;
            aruco::MarkerDetector::MarkerCandidate mc;
            mc.id = ids[c];
            for (int k = 0; k < 4; k++) mc.push_back(cv::Point2f(quads[c * 8 + 2 * k], quads[c * 8 + 2 * k + 1]));
            // the candidate's contour: the one that holds its four corners
            for (int i = 0; i < ncont && mc.contour.empty(); i++) {
                int at[4] = {-1, -1, -1, -1};
                for (size_t j = 0; j < contours[i].size(); j++)
                    for (int k = 0; k < 4; k++)
                        if (contours[i][j].x == (int)mc[k].x && contours[i][j].y == (int)mc[k].y) at[k] = (int)j;
                if (at[0] < 0 || at[1] < 0 || at[2] < 0 || at[3] < 0) continue;
                mc.contour = contours[i], mc.idx = i;
                // detectRectangles hands a swapped candidate its contour reversed (:622-625), so that the corners appear in contour order
                const bool forward = ((at[1] > at[0]) && (at[2] > at[1] || at[2] < at[0])) || (at[2] > at[1] && at[2] < at[0]);
                if (!forward) std::reverse(mc.contour.begin(), mc.contour.end());
            }
            if (mc.contour.empty()) return 4;
            det.refineCandidateLines(mc, cv::Mat(), cv::Mat());
            std::rotate(mc.begin(), mc.begin() + 4 - nrot[c], mc.end());
            std::cout << "refined " << mc.id;
            for (int k = 0; k < 4; k++) std::cout << " " << mc[k].x << " " << mc[k].y;
            std::cout << std::endl;
        }
        aruco::BoardDetector bd;
        const bool a = bd.isYPerpendicular();
        bd.setYPerpendicular(true);
        std::cout << "isYPerpendicular " << a << " " << bd.isYPerpendicular() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 2;
    }
}
