#pragma once
#include "aruco_hip_shim.hpp"   // aruco::BoardDetector (reference src/boarddetector.h)
