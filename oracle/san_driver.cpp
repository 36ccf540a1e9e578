// ORACLE — TEST INFRASTRUCTURE ONLY. Driver for the sanitizer build of the CPU restatement (-fsanitize=address,undefined on the
// oracle, the thing every parity claim rests on): `make -C oracle SAN=1` links this file with the restatement under ASan + UBSan
// (-fno-sanitize-recover: any finding aborts), tests/test_oracle_sanitized.py feeds it the reference's stills and compares what it prints.
//   san_driver <image.pgm> <case.txt>
// case.txt (written by the test from tests/golden/*.json), one item per line:
//   K <9 floats> | dist <n> <n floats> | size <marker size> | params <thres_p1> <thres_p2> <min_size> <max_size> <warp_size> <corner_method>
//   hrm <n> <tau0> <count> then <count> lines of n*n '0'/'1' | board <info_type> <nboard> then nboard lines: id + 12 floats | repj <thres>
// Output: "marker <id> <8 corner floats> <rvec 3> <tvec 3>" per marker, "board <prob> <has_pose> <rvec 3> <tvec 3>", "thres <checksum>".
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "orc.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::string magic;
    int w, h, maxv;
    if (!(f >> magic >> w >> h >> maxv) || magic != "P5") return 2;
    f.get();
    std::vector<uint8_t> gray((size_t)w * h);
    f.read((char*)gray.data(), (std::streamsize)gray.size());

    orc::Detector det;
    std::vector<float> K, dist;
    float size = -1, repj = -1;
    orc::BoardConf bc;
    bool have_board = false;
    std::ifstream c(argv[2]);
    std::string line;
    while (std::getline(c, line)) {
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "K") {
            K.resize(9);
            for (auto& v : K) ss >> v;
        } else if (key == "dist") {
            int n;
            ss >> n;
            dist.resize(n);
            for (auto& v : dist) ss >> v;
        } else if (key == "size") {
            ss >> size;
        } else if (key == "repj") {
            ss >> repj;
        } else if (key == "params") {
            ss >> det.prm.thres_p1 >> det.prm.thres_p2 >> det.prm.min_size >> det.prm.max_size >> det.prm.warp_size >> det.prm.corner_method;
        } else if (key == "hrm") {
            int n, tau0, count;
            ss >> n >> tau0 >> count;
            det.hrm.n = n, det.hrm.tau0 = tau0, det.hrm.rate = 1.f;
            for (int i = 0; i < count; i++) {
                std::getline(c, line);
                uint64_t code = 0;
                for (int b = 0; b < n * n; b++)
                    if (line[b] == '1') code |= 1ull << b;
                det.hrm.codes.push_back(code);
            }
        } else if (key == "board") {
            int nb;
            ss >> bc.info_type >> nb;
            for (int i = 0; i < nb; i++) {
                std::getline(c, line);
                std::istringstream bs(line);
                int id;
                bs >> id;
                bc.ids.push_back(id);
                for (int p = 0; p < 4; p++) {
                    orc::Pt3f q;
                    bs >> q.x >> q.y >> q.z;
                    bc.obj.push_back(q);
                }
            }
            have_board = true;
        }
    }
    std::vector<orc::Marker> out;
    det.detect(gray.data(), w, h, w, K.empty() ? nullptr : K.data(), dist.empty() ? nullptr : dist.data(), (int)dist.size(), size, 0, out);
    std::printf("markers %zu\n", out.size());
    for (auto& m : out) {
        std::printf("marker %d", m.id);
        for (int k = 0; k < 4; k++) std::printf(" %.9g %.9g", m.c[k].x, m.c[k].y);
        for (int k = 0; k < 3; k++) std::printf(" %.17g", m.rvec[k]);
        for (int k = 0; k < 3; k++) std::printf(" %.17g", m.tvec[k]);
        std::printf("\n");
    }
    if (have_board) {
        orc::Board b;
        const float prob = orc::board_detect(out, bc, K.empty() ? nullptr : K.data(), dist.empty() ? nullptr : dist.data(), (int)dist.size(), size, repj, 0, b);
        std::printf("board %.9g %d %.17g %.17g %.17g %.17g %.17g %.17g\n", prob, b.has_pose, b.rvec[0], b.rvec[1], b.rvec[2], b.tvec[0], b.tvec[1], b.tvec[2]);
    }
    unsigned long sum = 0;
    for (auto v : det.thres) sum = sum * 31 + v;
    std::printf("thres %lu candidates %zu rejected %zu contours %zu\n", sum, det.candidates.size(), det.rejected.size(), det.contours.size());
    return 0;
}
