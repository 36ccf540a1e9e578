// Headless equivalent of the reference's utils/aruco_simple.cpp (:37-101) and utils/aruco_simple_board.cpp on the
// C++ shim: read a gray PGM, optional intrinsics, detect, print every marker with the reference's operator<<.
//   aruco_simple <image.pgm> [intrinsics.txt] [marker_size] [board.txt]
// intrinsics.txt: "width height" then 9 camera-matrix values then the distortion coefficients.
// board.txt: "info_type n" then per marker "id" + 12 floats (4 corners x,y,z).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "aruco_hip_shim.hpp"

static bool read_pgm(const char* path, cv::Mat& img) {
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int w, h, maxv;
    if (!(f >> magic >> w >> h >> maxv) || magic != "P5") return false;
    f.get();
    img = cv::Mat(h, w, CV_8UC1);
    f.read((char*)img.data, (std::streamsize)w * h);
    return (bool)f;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "Usage: aruco_simple image.pgm [intrinsics.txt] [marker_size] [board.txt]" << std::endl;
        return 1;
    }
    try {
        cv::Mat gray;
        if (!read_pgm(argv[1], gray)) {
            std::cerr << "Could not open input" << std::endl;
            return 1;
        }
        aruco::CameraParameters cam;
        float size = argc > 3 ? (float)atof(argv[3]) : -1.f;
        if (argc > 2 && std::string(argv[2]) != "-") {
            std::ifstream f(argv[2]);
            int w, h;
            float K[9], d[8];
            f >> w >> h;
            for (int i = 0; i < 9; i++) f >> K[i];
            int nd = 0;
            while (nd < 8 && (f >> d[nd])) nd++;
            cam.setParams(K, d, nd, cv::Size(w, h));
            cam.resize(gray.size());
        }
        std::cout << std::setprecision(9);
        if (argc > 4) {
            std::ifstream f(argv[4]);
            aruco::BoardConfiguration bc;
            int n;
            f >> bc.mInfoType >> n;
            for (int i = 0; i < n; i++) {
                int id;
                f >> id;
                bc.ids.push_back(id);
                std::vector<cv::Point3f> pts(4);
                for (auto& p : pts) f >> p.x >> p.y >> p.z;
                bc.objPoints.push_back(pts);
            }
            aruco::BoardDetector bd;
            std::vector<aruco::Marker> markers;
            bd.getMarkerDetector().detect(gray, markers);
            aruco::Board board;
            float prob = bd.detect(markers, bc, board, cam, size);
            for (auto& m : board) std::cout << m << std::endl;
            std::cout << "board prob=" << prob;
            if (!board.Rvec.empty())
                std::cout << " Rvec=" << board.Rvec(0) << " " << board.Rvec(1) << " " << board.Rvec(2) << " Tvec=" << board.Tvec(0) << " "
                          << board.Tvec(1) << " " << board.Tvec(2);
            std::cout << std::endl;
            return 0;
        }
        aruco::MarkerDetector MDetector;
        std::vector<aruco::Marker> Markers;
        if (cam.isValid())
            MDetector.detect(gray, Markers, cam, size);
        else
            MDetector.detect(gray, Markers);
        for (unsigned int i = 0; i < Markers.size(); i++) std::cout << Markers[i] << std::endl;
        std::cout << "candidates=" << MDetector.getCandidates().size() << " thres=" << MDetector.getThresholdedImage().cols << "x"
                  << MDetector.getThresholdedImage().rows << std::endl;
        // parameter validation behaves like the reference's CV_Assert
        try {
            MDetector.setWarpSize(5);
            std::cout << "setWarpSize(5) accepted" << std::endl;
        } catch (cv::Exception& e) {
            std::cout << "setWarpSize(5) rejected" << std::endl;
        }
    } catch (std::exception& ex) {
        std::cout << "Exception :" << ex.what() << std::endl;
        return 2;
    }
    return 0;
}
