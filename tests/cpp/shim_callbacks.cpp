// Boundary gaps closed in round 2, driven the way a caller of the reference would:
//   * Marker::calculateExtrinsics as a member (reference src/marker.h:77,85): detect with the camera but without a marker
//     size (no pose), then every marker solves its own pose;
//   * MarkerDetector::setMakerDetectorFunction with a function of the caller's own (reference src/markerdetector.h:65-78,
//     :243-245): the 5x5 decoder below is a host implementation written for this test (Otsu, cell votes, rotations,
//     Hamming words) and must give the same markers as the library's device decoder.
//   shim_callbacks <image.pgm> <intrinsics.txt>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "aruco_hip_shim.hpp"

static int g_calls = 0;

// id or -1 of a square 8-bit patch holding a 7x7-cell marker; nRotations as the reference defines it
static int host_5x5_decoder(const cv::Mat& in, int& nRotations) {
    g_calls++;
    const int n = in.rows;
    if (in.cols != n || n < 7) return -1;
    // Otsu's threshold on the 256-bin histogram (maximum between-class variance, first maximum)
    double hist[256] = {0};
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) hist[in.at<unsigned char>(y, x)] += 1;
    const double total = (double)n * n;
    double mu = 0;
    for (int i = 0; i < 256; i++) mu += i * hist[i];
    mu /= total;
    double q1 = 0, mu1 = 0, best = 0;
    int thr = 0;
    for (int i = 0; i < 256; i++) {
        const double p = hist[i] / total;
        mu1 *= q1;
        q1 += p;
        const double q2 = 1.0 - q1;
        if (std::min(q1, q2) < 1.1920929e-7 || std::max(q1, q2) > 1.0 - 1.1920929e-7) continue;
        mu1 = (mu1 + i * p) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > best) best = sigma, thr = i;
    }
    const int sw = n / 7;
    bool cell[7][7];
    for (int cy = 0; cy < 7; cy++)
        for (int cx = 0; cx < 7; cx++) {
            int white = 0;
            for (int y = 0; y < sw; y++)
                for (int x = 0; x < sw; x++) white += in.at<unsigned char>(cy * sw + y, cx * sw + x) > thr;
            cell[cy][cx] = white > (sw * sw) / 2;
        }
    for (int i = 0; i < 7; i++)
        if (cell[0][i] || cell[6][i] || cell[i][0] || cell[i][6]) return -1;   // the frame must be black
    int code[5][5];
    for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++) code[y][x] = cell[y + 1][x + 1];
    static const int words[4][5] = {{1, 0, 0, 0, 0}, {1, 0, 1, 1, 1}, {0, 1, 0, 0, 1}, {0, 1, 1, 1, 0}};
    auto distance = [&](int c[5][5]) {
        int d = 0;
        for (int y = 0; y < 5; y++) {
            int row_best = 100;
            for (int w = 0; w < 4; w++) {
                int s = 0;
                for (int x = 0; x < 5; x++) s += c[y][x] != words[w][x];
                row_best = std::min(row_best, s);
            }
            d += row_best;
        }
        return d;
    };
    int cur[5][5], keep[5][5];
    std::memcpy(cur, code, sizeof(cur)), std::memcpy(keep, code, sizeof(keep));
    int min_dist = distance(cur);
    nRotations = 0;
    for (int r = 1; r < 4; r++) {
        int nxt[5][5];
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) nxt[i][j] = cur[5 - j - 1][i];
        std::memcpy(cur, nxt, sizeof(cur));
        const int d = distance(cur);
        if (d < min_dist) min_dist = d, nRotations = r, std::memcpy(keep, cur, sizeof(keep));
    }
    if (min_dist != 0) return -1;
    int id = 0;
    for (int y = 0; y < 5; y++) {
        id <<= 1;
        id |= keep[y][1];
        id <<= 1;
        id |= keep[y][3];
    }
    return id;
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::string magic;
        int w, h, maxv;
        if (!(f >> magic >> w >> h >> maxv) || magic != "P5") return 1;
        f.get();
        cv::Mat gray(h, w, CV_8UC1);
        f.read((char*)gray.data, (std::streamsize)w * h);
        aruco::CameraParameters cam;
        {
            std::ifstream fi(argv[2]);
            int cw, ch;
            float K[9], d[8];
            fi >> cw >> ch;
            for (int i = 0; i < 9; i++) fi >> K[i];
            int nd = 0;
            while (nd < 8 && (fi >> d[nd])) nd++;
            cam.setParams(K, d, nd, cv::Size(cw, ch));
            cam.resize(gray.size());
        }
        std::cout << std::setprecision(9);
        aruco::MarkerDetector MDetector;
        std::vector<aruco::Marker> Markers;

        // (1) Marker::calculateExtrinsics, both overloads
        MDetector.detect(gray, Markers, cam.CameraMatrix, cam.Distorsion, -1);
        for (size_t i = 0; i < Markers.size(); i++) {
            if (!Markers[i].Rvec.empty()) return 3;   // no marker size: no pose yet
            if (i & 1)
                Markers[i].calculateExtrinsics(1.0f, cam, false);
            else
                Markers[i].calculateExtrinsics(1.0f, cam.CameraMatrix, cam.Distorsion, false);
            std::cout << "extr " << Markers[i] << std::endl;
        }
        try {
            aruco::Marker bad;
            bad.calculateExtrinsics(1.0f, cam, false);
            std::cout << "invalid marker accepted" << std::endl;
        } catch (cv::Exception&) {
            std::cout << "invalid marker rejected" << std::endl;
        }

        // (2) device decoder, then the caller's own decoder, then the device decoder again
        MDetector.detect(gray, Markers, cam, 1.0f);
        for (auto& m : Markers) std::cout << "dev " << m << std::endl;
        MDetector.setMakerDetectorFunction(host_5x5_decoder);
        MDetector.detect(gray, Markers, cam, 1.0f);
        for (auto& m : Markers) std::cout << "usr " << m << std::endl;
        std::cout << "decoder calls=" << g_calls << " candidates=" << MDetector.getCandidates().size() << std::endl;
        const int calls = g_calls;
        MDetector.setMakerDetectorFunction(aruco::FiducidalMarkers::detect);
        MDetector.detect(gray, Markers, cam, 1.0f);
        std::cout << "after reset: calls " << (g_calls - calls) << " markers " << Markers.size() << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 2;
    }
}
