// m3d_config.hpp -- the library's tunables in ONE place (include/misc3d_amd.h: m3d_config).
// Read once from the environment (M3D_* variables, listed next to the fields in the public header) when the library
// is first used; m3d_set_config replaces them at run time (tests switch scoring paths with it).
#pragma once
#include "../../include/misc3d_amd.h"

namespace m3d {
const m3d_config& config();        // current settings
void config_store(const m3d_config& c);
}  // namespace m3d
