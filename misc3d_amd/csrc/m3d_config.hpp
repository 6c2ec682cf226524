// m3d_config.hpp -- the library's tunables in ONE place (include/misc3d_amd.h: m3d_config).
// Read once from the environment (M3D_* variables, listed next to the fields in the public header) when the library
// is first used; m3d_set_config replaces them at run time (tests switch scoring paths with it).
#pragma once
#include "../../include/misc3d_amd.h"

namespace m3d {
// The settings the calling thread works with.  While the thread holds a lane (CtxLock / LaneLock: config_pin) that is the
// SNAPSHOT taken when it took the lane -- m3d_set_config from another thread does not change the switches of a call in
// flight (a fit that started with the histogram bound finishes with it) -- otherwise a copy of the current ones.
const m3d_config& config();
void config_store(const m3d_config& c);
void config_pin();     // nested pins keep the outermost snapshot
void config_unpin();
}  // namespace m3d
