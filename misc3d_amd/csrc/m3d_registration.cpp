// m3d_registration.cpp -- registration entry points (placeholder until the kernels land).
#include "m3d_driver.hpp"

using namespace m3d;

extern "C" {

int m3d_kabsch(const double*, const double*, size_t, int, int, double*) {
    return fail(M3D_ERR_INTERNAL, "m3d_kabsch: not implemented yet");
}
int m3d_registration_ransac(const double*, size_t, const double*, size_t, const size_t*, const size_t*,
                            size_t, double, int, double, double, const uint64_t*, int, double*,
                            m3d_reg_stats*) {
    return fail(M3D_ERR_INTERNAL, "m3d_registration_ransac: not implemented yet");
}
int m3d_match_mutual_nn(const double*, size_t, const double*, size_t, int, int, int, int, size_t*,
                        size_t*, size_t*) {
    return fail(M3D_ERR_INTERNAL, "m3d_match_mutual_nn: not implemented yet");
}
}
