// OpenCV 2/3 spelling of <opencv2/core.hpp> (the reference's apps include it: utils/aruco_simple_board.cpp:31)
#include "../core.hpp"
