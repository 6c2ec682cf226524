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


/* fp64 VALU issue rate the device sustains (independent v_mul_f64 / v_add_f64 chains, no FMA, no memory traffic) over
 * about `ms_target` milliseconds of that load, in 1e12 lane-operations per second: the attainable counterpart of the
 * nominal 39.3 (256 CU x 4 SIMD x 16 lanes x 2.4 GHz) -- the chip clocks below 2.4 GHz under sustained fp64 load. */
int m3d_bench_fp64_issue_rate(int device, double ms_target, double *tera_lane_ops_per_s, double *ms_measured);

/* m3d_cloud_create's own wall clock in ms: out[0] = the whole call (always); with m3d_config.kernel_timing set when the
 * cloud was created also the phases, each closed by a stream synchronisation: out[1] = host-to-device copies of the
 * caller's arrays + AoS -> SoA transposes, out[2] = bounding box of the device copy incl. its round trip, out[3] =
 * Hilbert counting sort, out[4] = tile boxes + the tiles' fp32 offsets ("report upload separately"). */
int m3d_bench_cloud_setup_ms(const m3d_cloud *cloud, double out[5]);

/* plane_bound_k (m3d_bound.hip) on its own: ub_out[h] = the histogram upper bound of hypothesis h's inlier count, for EVERY
 * hypothesis of the sample table (n_hypotheses x 3 indices, at most 16 384; nothing is pruned).  What
 * tests/test_gpu_plane_bound.py holds against the exact counts of m3d_cloud_score_range: ub >= count, hypothesis by hypothesis.
 * Builds the cloud's tile frames if it has none yet. */
int m3d_bench_plane_upper_bounds(m3d_cloud *cloud, double threshold, const uint32_t *samples, size_t n_hypotheses,
                                 uint32_t *ub_out);
/* ... for any kind (M3D_PLANE / M3D_SPHERE / M3D_CYLINDER; samples: n_hypotheses x 3 / 4 / 2 indices): spheres and cylinders are
 * bounded through the slab their shell is over one tile (m3d_bound_fp.hpp cyl_pair_ub). */
int m3d_bench_upper_bounds(m3d_cloud *cloud, int kind, double threshold, const uint32_t *samples, size_t n_hypotheses,
                           uint32_t *ub_out);
/* Wall clock of the calling thread's LAST m3d_segment_plane_iterative* call in ms: out[0] the whole call, out[1]
 * m3d_cloud_create (upload, sort, tile boxes), out[2] the round loop, out[3] the final copy of the index lists out of
 * the page-locked staging array (0 when the caller's array is page-locked), out[4] of the round loop: the rounds on more
 * than an eighth of the cloud, out[5] = their number + 1e-4 x all rounds. */
int m3d_bench_last_segment_ms(double out[6]);

/* TEST hook (tests/fp_order_worker.py): the EdgeLength + Distance checkers of the registration path evaluated on the
 * HOST by the very code the kernels compile (m3d_reg_fp.hpp reg_checkers): ps / pd = the 3 sampled source / target
 * points (3 x 3 doubles each), T = 4 x 4 row-major.  Returns 1 = pass, 0 = rejected.  Lets a box without a GPU check
 * that a library built for another M3D_FP_ORDER carries that association into the registration code as well. */
int m3d_bench_reg_checkers(const double *ps, const double *pd, const double *T, double edge_threshold,
                           double distance_threshold);

/* 1: the library was built with -DM3D_EXPERIMENTAL -- round 4's three refuted variants (score_mfma_k, score_screen4_k,
 * compact_write_k<.., ONE>) are compiled in and m3d_config.score_mfma / score_waves4 / compact_one_pass switch them on;
 * 0 (the product build): those switches are ignored and m3d_bench_mfma_probe returns an error. */
int m3d_bench_experimental(void);

/* TEST hook (tests/test_gpu_match_sliced.py): which way the CALLING THREAD's last m3d_match_mutual_nn went -- bit 0: the split-fp16
 * MFMA screen produced the result, bit 1: the matrices went up in slices under the scan (m3d_config.match_pipeline), bit 2: a
 * later slice did not fit the scale chosen from the first ones and the search was redone whole on the resident matrices,
 * bit 3: the fp32 screen, bit 4: fp64 brute force. */
unsigned m3d_bench_match_last_path(void);

/* TEST hook (tests/test_gpu_mfma_screen.py): the MFMA screen of the plane scoring (score_mfma_k) on ONE tile -- 512 points
 * xyz (row-major doubles), their box (centre xyz, half extents xyz: every |x - centre| <= half extent), n_h plane records
 * (a, b, c, d, T, 0, 0, 0) -- through the production kernel's own operand builders and the matrix pipe:
 * out_q[h][i] = (u1, u2) = the pipe's T - S and T + S for hypothesis h and point i, unscaled (inside <=> u1 u2 > 0, decided when
 * |u1 u2| >= out_h[h][0]), out_h[h] = (band h on the product, scale Sigma_h, the bound E_p on either u) (h = NaN: the record
 * is not screened), out_off[i] = the fp32 offsets the pipe was fed.  Sizes: out_q n_h x 512 x 2, out_h n_h x 3, out_off 512 x 3. */
int m3d_bench_mfma_probe(int device, const double *xyz512, const double box[6], double max_abs, const double *records,
                         size_t n_h, double *out_q, double *out_h, float *out_off);

#ifdef __cplusplus
}
#endif
#endif /* MISC3D_AMD_BENCH_H */
