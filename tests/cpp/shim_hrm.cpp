// The reference's HRM test (test/core_tests.cpp:310-353) written against the C++ shim: dictionary file, the same setter
// calls, detect with camera parameters; prints the markers with the reference's operator<<.
//   shim_hrm <image.pgm> <dictionary.yml> <intrinsics.txt>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "aruco_hip_shim.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::string magic;
        int w, h, maxv;
        if (!(f >> magic >> w >> h >> maxv) || magic != "P5") return 1;
        f.get();
        cv::Mat gray(h, w, CV_8UC1);
        f.read((char*)gray.data, (std::streamsize)w * h);

        aruco::Dictionary dictionary;
        dictionary.fromFile(argv[2]);
        aruco::HighlyReliableMarkers::loadDictionary(dictionary);

        aruco::CameraParameters cam;
        {
            std::ifstream fi(argv[3]);
            int cw, ch;
            float K[9], d[8];
            fi >> cw >> ch;
            for (int i = 0; i < 9; i++) fi >> K[i];
            int nd = 0;
            while (nd < 8 && (fi >> d[nd])) nd++;
            cam.setParams(K, d, nd, cv::Size(cw, ch));
            cam.resize(gray.size());
        }
        aruco::MarkerDetector MDetector;
        MDetector.enableLockedCornersMethod(false);
        MDetector.setMakerDetectorFunction(aruco::HighlyReliableMarkers::detect);
        MDetector.setThresholdParams(21, 7);
        MDetector.setCornerRefinementMethod(aruco::MarkerDetector::LINES);
        MDetector.setWarpSize((dictionary[0].n() + 2) * 8);
        MDetector.setMinMaxSize(0.005, 0.5);
        std::vector<aruco::Marker> Markers;
        MDetector.detect(gray, Markers, cam, 1.0f);
        std::cout << std::setprecision(9);
        for (auto& m : Markers) std::cout << m << std::endl;
        // back to the default decoder: the HRM markers are not 5x5 Hamming markers
        MDetector.setMakerDetectorFunction(aruco::FiducidalMarkers::detect);
        MDetector.detect(gray, Markers, cam, 1.0f);
        std::cout << "fiducial=" << Markers.size() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 2;
    }
}
