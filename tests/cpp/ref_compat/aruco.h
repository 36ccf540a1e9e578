// What "aruco.h" resolves to when a caller of the reference is built against the MI355X path: the reference's umbrella header
// (/root/reference/src/aruco.h:180-183) pulls markerdetector.h, boarddetector.h and cvdrawingutils.h; here the same class names come from
// the shim. INTEGRATION.md shows this file as the one-line change a maintainer makes.
#pragma once
#include "aruco_hip_shim.hpp"
