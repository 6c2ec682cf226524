// misc3d/logging.h -- error/log convention of the reference (include/misc3d/logging.h:98-206,
// src/logging.cpp:54-106) on top of the C ABI: LogError THROWS std::runtime_error with the
// "[Misc3D Error] " prefix, warnings/info go to stdout, default verbosity is Info.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../misc3d_amd.h"

namespace misc3d {

enum class VerbosityLevel { Error = 0, Warning = 1, Info = 2, Debug = 3 };

inline VerbosityLevel& VerbosityRef() {
    static VerbosityLevel level = VerbosityLevel::Info;  // src/logging.cpp:56
    return level;
}
inline void SetVerbosityLevel(VerbosityLevel level) { VerbosityRef() = level; }
inline VerbosityLevel GetVerbosityLevel() { return VerbosityRef(); }

[[noreturn]] inline void LogError(const std::string& msg) {
    throw std::runtime_error("[Misc3D Error] " + msg);  // src/logging.cpp:64-74
}
inline void LogWarning(const std::string& msg) {
    if (VerbosityRef() >= VerbosityLevel::Warning) std::printf("[Misc3D WARNING] %s\n", msg.c_str());
}
inline void LogInfo(const std::string& msg) {
    if (VerbosityRef() >= VerbosityLevel::Info) std::printf("[Misc3D INFO] %s\n", msg.c_str());
}

// negative C-ABI code -> the exception the reference would have thrown at that point
inline int CheckStatus(int rc) {
    if (rc < 0) LogError(m3d_last_error());
    return rc;
}

}  // namespace misc3d
