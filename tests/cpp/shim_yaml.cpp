// Host-only: the shim's readers for the reference's YAML files (row f3): prints what was parsed.
//   shim_yaml <intrinsics.yml> <board.yml>
#include <cstdio>

#include "aruco_hip_shim.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    aruco::CameraParameters cp;
    cp.readFromXMLFile(argv[1]);
    std::printf("%d %d\n", cp.CamSize.width, cp.CamSize.height);
    for (int i = 0; i < 9; i++) std::printf("%.9g ", cp.CameraMatrix(i / 3, i % 3));
    std::printf("\n");
    for (int i = 0; i < 5; i++) std::printf("%.9g ", cp.Distorsion(0, i));
    std::printf("\n");
    aruco::BoardConfiguration bc;
    bc.readFromFile(argv[2]);
    std::printf("%d %zu\n", bc.mInfoType, bc.ids.size());
    for (size_t i = 0; i < bc.ids.size(); i++) {
        std::printf("%d", bc.ids[i]);
        for (int k = 0; k < 4; k++) std::printf(" %.9g %.9g %.9g", bc.objPoints[i][k].x, bc.objPoints[i][k].y, bc.objPoints[i][k].z);
        std::printf("\n");
    }
    try {
        aruco::BoardConfiguration bad;
        bad.readFromFile(argv[1]);   // not a board file
        std::printf("nothrow\n");
    } catch (const std::exception&) {
        std::printf("throws\n");
    }
    // further malformed inputs (argv[3..]): every one must throw, none may crash
    for (int i = 3; i < argc; i++) {
        int thrown = 0;
        try {
            aruco::CameraParameters c2;
            c2.readFromXMLFile(argv[i]);
        } catch (const std::exception&) {
            thrown |= 1;
        }
        try {
            aruco::BoardConfiguration b2;
            b2.readFromFile(argv[i]);
        } catch (const std::exception&) {
            thrown |= 2;
        }
        try {
            aruco::Dictionary d2;
            d2.fromFile(argv[i]);
        } catch (const std::exception&) {
            thrown |= 4;
        }
        std::printf("bad%d %d\n", i - 3, thrown);
    }
    return 0;
}
