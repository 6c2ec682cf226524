// pin_seed.h -- gives the REFERENCE's RandomSampler a fixed seed without touching its sources.
//
// include/misc3d/utils.h:74-77 seeds the sampler with `std::random_device rd; rng_ = std::mt19937(rd());` and the
// RANSAC loop constructs that sampler itself (include/misc3d/common/ransac.h:570: `RandomSampler<size_t>
// sampler(num_points)` -- the class template's Sampler parameter is not what the loop uses), so the only way to run the
// reference's own FitModel / SegmentPlaneIterative on a known stream is to replace the seed source.  This header is
// force-included (`g++ -include pin_seed.h`) in front of every translation unit that pulls in misc3d/utils.h: <random>
// is included FIRST, untouched; afterwards the token `random_device` names a device that returns m3d_pin_seed,
// m3d_pin_seed + 1, ... -- one value per sampler constructed, i.e. per FitModel call, which is the convention of this
// repository's seeded oracle (seed + round in orc_segment_plane_iterative).  Run with OMP_NUM_THREADS=1: the loop is
// then the sequential one the oracle restates.
#ifndef M3D_PIN_SEED_H
#define M3D_PIN_SEED_H
#include <random>
#include <cstdint>
extern "C" {
extern uint64_t m3d_pin_seed;    // defined in pin_reference.cpp
extern uint64_t m3d_pin_calls;
}
namespace std {
struct m3d_pin_random_device {
    using result_type = unsigned int;
    result_type operator()() { return (result_type)(m3d_pin_seed + m3d_pin_calls++); }
};
}  // namespace std
#define random_device m3d_pin_random_device
#endif
