/*
 * misc3d_amd_bench.h -- measurement hooks of libmisc3d_amd.so.  NOT part of the drop-in boundary: nothing here
 * replaces a function of the reference; bench.py / tools/ use it to time single kernels for the `roofline`
 * objects.  Kept apart from include/misc3d_amd.h so that the product header holds only what a reference caller binds.
 */
#ifndef MISC3D_AMD_BENCH_H
#define MISC3D_AMD_BENCH_H

#include "misc3d_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* average duration in ms of the scoring kernel alone
 * (score_k, the dominant kernel) over `reps` launches of `n_hypotheses` hypotheses, timed with HIP
 * events on the library's own stream after one untimed launch.  Not part of the reference. */
int m3d_bench_time_score(m3d_cloud *cloud, int kind, double threshold, const uint32_t *samples,
                         size_t n_hypotheses, int reps, int mode, double *ms_avg,
                         uint64_t *listed_pairs);
/* mode 0: score_list_k (production: counting over the (tile, hypothesis) pairs that survive the box
 * test); mode 1: cull_k (the box tests); mode 2: score_k (dense: every tile x every hypothesis).
 * listed_pairs (may be NULL): number of surviving (tile, hypothesis) pairs, tile = 512 points. */


#ifdef __cplusplus
}
#endif
#endif /* MISC3D_AMD_BENCH_H */
