/*
 * misc3d_amd.h -- C ABI of the MI355X (gfx950) implementation of the Misc3D RANSAC hot path.
 *
 * This is the drop-in boundary: every entry point replaces one function of the reference
 * (yuecideng/Misc3D; paths below are relative to its repository root).  Plain pointers and sizes,
 * no C++/torch types.  Inputs are borrowed for the duration of the call; outputs go to
 * caller-allocated buffers; nothing allocated inside crosses the ABI except opaque handles that
 * are released with the matching *_destroy.
 *
 * Return convention (mirrors the reference's three outcomes):
 *     1  reference returned `true`
 *     0  reference returned `false` (soft failure: params are zeroed by the callers exactly as
 *        python/py_common.cpp:21-23,39-41,61-63 do; inliers are still returned)
 *    <0  the reference raises (misc3d::LogError throws std::runtime_error, src/logging.cpp:64-74):
 *        M3D_ERR_* below; m3d_last_error() gives the reference's message text.
 * The library never throws and never falls back to a CPU path: without a usable HIP device every
 * compute entry point returns M3D_ERR_DEVICE.
 *
 * Point clouds are Open3D's layout: contiguous N x 3 float64 (open3d::geometry::PointCloud::points_
 * is std::vector<Eigen::Vector3d>, 24-byte POD elements), so `pcd.points_.data()->data()` can be
 * passed as is.
 */
#ifndef MISC3D_AMD_H
#define MISC3D_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3D_OK 1
#define M3D_FALSE 0
#define M3D_ERR_PROBABILITY (-1) /* "Probability must be > 0 or <= 1.0"          ransac.h:483-485 */
#define M3D_ERR_TOO_FEW_POINTS (-2) /* "Can not fit model due to lack of points" ransac.h:510-513;
                                       "The number of points pair is less than 3." transform_estimation.cpp:17-19,130-133 */
#define M3D_ERR_NO_NORMALS (-3) /* "Fit cylinder requires normals."            py_common.cpp:50-52, ransac.h:356-359 */
#define M3D_ERR_SIZE_MISMATCH (-4) /* "The number of points pair is not equal." transform_estimation.cpp:20-22 */
#define M3D_ERR_INVALID_ARG (-5) /* null pointer / out-of-range index / N >= 2^31 */
#define M3D_ERR_DEVICE (-6) /* no HIP device, HIP runtime error, or out of device memory */
#define M3D_ERR_INTERNAL (-7) /* self-check failed (counts from the scoring kernel and the refine pass disagree) */

enum m3d_model_kind { M3D_PLANE = 0, M3D_SPHERE = 1, M3D_CYLINDER = 2 };

/* Observable state of misc3d::common::RANSAC after FitModel (ransac.h:616-619 logs fitness and
 * count) plus counters of this implementation. */
typedef struct m3d_stats {
    double fitness;              /* best inlier ratio (ransac.h:617) */
    double inlier_rmse;          /* best error/sqrt(n) if it had to be evaluated (ties), else NaN */
    uint64_t count;              /* valid hypotheses evaluated before the adaptive stop ("run {} iterations") */
    uint64_t iterations;         /* loop index at which the reference loop went idle */
    int64_t best_index;          /* hypothesis index of the best model, -1 if none */
    int32_t general_fit_ok;      /* RefineModel's return (ransac.h:548) */
    int32_t ties;                /* equal-fitness comparisons (ransac.h:596) decided during the replay */
    uint64_t hypotheses_scored;  /* hypotheses the GPU scored (>= iterations: speculative chunks) */
    uint64_t exact_rmse_evals;   /* serial-order error sums needed (ties that order-free sums could not decide) */
    double ms_sample;            /* host: std::mt19937 sample table */
    double ms_score;             /* host clock: sampling, minimal fit, scoring and replay, from the first launch to the last replayed chunk (filled with m3d_config.kernel_timing) */
    double ms_refine;            /* device+host: inlier compaction, GeneralFit, copy-out */
    double ms_total;             /* wall clock of the call */
    double ms_score_kernel;      /* device: sum of the scoring-kernel launches alone (HIP events around each; m3d_config.kernel_timing) */
    uint32_t score_launches;     /* number of scoring-kernel launches (chunks) behind ms_score_kernel */
    uint32_t early_pick_redone;  /* 1: RefineModel had been started on the device's own pick of the winner (probability-1
                                  * fits) and the sequential replay chose another hypothesis (rmse tie): it was run again */
    uint64_t pairs_scored;       /* (512-point tile, hypothesis) pairs those launches evaluated after culling and pruning */
    uint64_t pairs_exact;        /* ... of which the fp32 screen (m3d_config.score_fp32_screen) left to the exact fp64 code */
    uint64_t pairs_timed;        /* ... of pairs_scored: evaluated by the launches behind ms_score_kernel / score_launches.  Smaller than
                                    pairs_scored when the leading hypotheses of a fit's first chunk were counted inside cull_lead_k
                                    (one launch for the box tests and the lead pass: one GPU, fp32 paths on), which is not timed */
} m3d_stats;

/* ---- one-shot fits: python/py_common.cpp:11-67 FitPlane / FitSphere / FitCylinder ------------- */
/* seed: NULL = std::random_device (reference behaviour, utils.h:74-77); else std::mt19937(*seed).
 * device: HIP device ordinal.  inliers: capacity n.  params: 4 (plane a,b,c,d; sphere cx,cy,cz,r)
 * or 7 (cylinder px,py,pz,nx,ny,nz,r).  On return 0 the caller zeroes params (py_common.cpp:21-23). */
int m3d_fit_plane(const double *xyz, size_t n, double threshold, size_t max_iteration,
                  double probability, const uint64_t *seed, int device, double params[4],
                  size_t *inliers, size_t *n_inliers, m3d_stats *stats);
int m3d_fit_sphere(const double *xyz, size_t n, double threshold, size_t max_iteration,
                   double probability, const uint64_t *seed, int device, double params[4],
                   size_t *inliers, size_t *n_inliers, m3d_stats *stats);
int m3d_fit_cylinder(const double *xyz, const double *normals, size_t n, double threshold,
                     size_t max_iteration, double probability, const uint64_t *seed, int device,
                     double params[7], size_t *inliers, size_t *n_inliers, m3d_stats *stats);

/* ---- resident cloud: RANSAC::SetPointCloud (ransac.h:469-475) keeps a copy; here the copy lives
 *      in HBM as SoA x[],y[],z[] (+ normals) so repeated fits do not re-upload. ------------------ */
typedef struct m3d_cloud m3d_cloud;
m3d_cloud *m3d_cloud_create(const double *xyz, const double *normals /* may be NULL */, size_t n,
                            int device);
void m3d_cloud_destroy(m3d_cloud *cloud);
/* Device blocks released by the library (destroyed clouds, per-call scratch of the registration / matcher / normals /
 * boundary entry points) are parked on a per-device free list (at most m3d_config.pool_limit_mb, default 4 GiB) and handed out again instead of going
 * through hipFree / hipMalloc on every call; this returns them, and the upload scratch, to the driver. */
void m3d_release_cached(int device);
/* Page-locked host memory for OUTPUT buffers (inlier index lists).  Any host pointer is accepted wherever this
 * header takes an output buffer; one obtained here lets the library start the device-to-host copy of the index
 * list as soon as the list exists, overlapped with the GeneralFit sums, instead of last (a copy into pageable
 * memory occupies the calling thread until it is done).  NULL + m3d_last_error() on failure. */
void *m3d_host_alloc(size_t bytes);
void m3d_host_free(void *p);
/* Points currently in the cloud (shrinks with m3d_cloud_remove_inliers) / points it was created with. */
size_t m3d_cloud_size(const m3d_cloud *cloud);
size_t m3d_cloud_original_size(const m3d_cloud *cloud);
/* RANSAC::SetProbability + SetMaxIteration + FitModel (ransac.h:482-516) on the resident cloud.
 * inliers may be NULL (then only *n_inliers is reported). */
int m3d_cloud_fit(m3d_cloud *cloud, int kind, double threshold, size_t max_iteration,
                  double probability, const uint64_t *seed, double *params, size_t *inliers,
                  size_t *n_inliers, m3d_stats *stats);

/* ---- many short fits in ONE call (round 6).  The reference's fits are called one at a time from Python
 * (python/py_common.cpp:11-78 under its callers' loops); a resident fit of a few dozen hypotheses takes ~80 us on the device, and
 * threads of such calls are serialised by what the binding does per call under the interpreter's lock, not by the library's
 * lanes.  A batch hands the loop over: the jobs of one cloud run in the order given, clouds that live on different lanes of a
 * device (m3d_cloud_create_lane) or on different devices run side by side on threads of the library -- one call, one release of
 * the interpreter's lock.  Every job carries its own rc (m3d_cloud_fit's: 1 / 0 / < 0) and outputs; the call returns M3D_OK, or
 * the first failing job's error code with m3d_last_error() = "job <k>: ...".  inflight: lanes worked at a time (0 = all). */
typedef struct m3d_fit_job {
    m3d_cloud *cloud;
    int32_t kind, has_seed;
    double threshold, probability;
    uint64_t max_iteration, seed;
    size_t *inliers;           /* capacity: the cloud's size; may be NULL (then only n_inliers is reported) */
    double params[8];          /* out */
    uint64_t n_inliers;        /* out */
    int32_t rc, reserved_;     /* out */
    m3d_stats stats;           /* out */
} m3d_fit_job;
int m3d_cloud_fit_batch(m3d_fit_job *jobs, size_t n_jobs, int inflight);
/* m3d_cloud_create on a lane of the caller's choice (0 .. m3d_config.lanes - 1; < 0: the calling thread's own, as m3d_cloud_create):
 * a single-threaded caller that wants its clouds' fits to overlap (m3d_cloud_fit_batch) spreads them over the lanes itself. */
m3d_cloud *m3d_cloud_create_lane(const double *xyz, const double *normals /* may be NULL */, size_t n, int device, int lane);

/* ---- hypothesis-range scoring: the shardable unit (ransac.h:572-590 body for i in [begin,end)).
 * samples: H x m sample indices (m = 3 plane, 4 sphere, 2 cylinder) as the sequential sampler
 * (utils.h:81-97) produces them; m3d_draw_samples reproduces that table from a seed.
 * Outputs (host, length end-begin): valid = MinimalFit's return, counts = inlier_num of
 * EvaluateModel (ransac.h:626-641), models = (end-begin) x 8 doubles (first 4/7 = parameters).
 * Any of counts/valid/models may be NULL. */
int m3d_draw_samples(size_t n_points, int kind, size_t n_hypotheses, uint64_t seed,
                     uint32_t *samples);
/* Estimator::MinimalFit (ransac.h:138-162, 239-294, 354-417) for one sample on the HOST: pts = m x 3
 * doubles (m = 3 / 4 / 2), normals = m x 3 (cylinder only).  model: 8 doubles (first 4 / 7 = parameters),
 * *valid = MinimalFit's return.  Bit-identical to what the device computes for the same sample. */
int m3d_minimal_fit(int kind, const double *pts, const double *normals, double *model, uint8_t *valid);
int m3d_cloud_score_range(m3d_cloud *cloud, int kind, double threshold, const uint32_t *samples,
                          size_t begin, size_t end, uint32_t *counts, uint8_t *valid,
                          double *models);
/* Sequential sampler object = the RandomSampler of ransac.h:570 with an explicit seed; it records
 * every sample it has drawn (m3d_sampler_table(s, H) draws up to H hypotheses and returns the H x m
 * table).  m3d_cloud_score_shard scores ONE rank's share of hypotheses [begin,end) of that single
 * stream: the window is cut into slices of `slice` hypotheses, slice j belongs to rank j % world.
 * Every rank draws the whole window (identical tables everywhere), kernels for this rank's slices are
 * launched as soon as their samples exist, so drawing the other ranks' slices overlaps GPU scoring.
 * counts/valid receive this rank's records in increasing hypothesis order, *n_mine their number.
 * valid == NULL: counts[i] carries MinimalFit's return in bit 31 (counts are < 2^31) -- the 4-byte record the
 * ranks exchange; m3d_replay_chunk accepts the same form (valid == NULL). */
typedef struct m3d_sampler m3d_sampler;
m3d_sampler *m3d_sampler_create(size_t n_points, int kind, uint64_t seed);
void m3d_sampler_destroy(m3d_sampler *s);
size_t m3d_sampler_drawn(const m3d_sampler *s);
const uint32_t *m3d_sampler_table(m3d_sampler *s, size_t n_hypotheses);
int m3d_cloud_score_shard(m3d_cloud *cloud, m3d_sampler *sampler, double threshold, size_t begin,
                          size_t end, size_t slice, uint32_t world, uint32_t rank, uint32_t *counts,
                          uint8_t *valid, size_t *n_mine);
/* Serial-order error sum of EvaluateModel for ONE model (ransac.h:632-640), used to break
 * fitness ties exactly as the reference does.  *count is the inlier number, *error the sum. */
int m3d_cloud_exact_error(m3d_cloud *cloud, int kind, double threshold, const double *model,
                          uint64_t *count, double *error);
/* RefineModel (ransac.h:534-549) for a given pre-refinement model: inlier indices (ascending) and
 * GeneralFit applied in place to params.  Return 1/0 = GeneralFit's return. */
int m3d_cloud_refine(m3d_cloud *cloud, int kind, double threshold, double *params,
                     size_t *inliers, size_t *n_inliers);
/* Same, when the caller already knows the inlier count of `params` from the scoring pass (the record of the best
 * hypothesis, m3d_cloud_score_shard): the GeneralFit sums and the index-list copy are issued without waiting for
 * the compaction's own total, which is only checked at the end (a different total silently takes the ordinary
 * order).  expected_inliers < 0: unknown (= m3d_cloud_refine). */
int m3d_cloud_refine_expect(m3d_cloud *cloud, int kind, double threshold, double *params,
                            int64_t expected_inliers, size_t *inliers, size_t *n_inliers);
/* pcd_copy = pcd_copy->SelectByIndex(inliers, invert=true), src/iterative_plane_segmentation.cpp:33, on
 * the resident cloud: the inliers of `model` (distance < threshold, RefineModel's rule ransac.h:537-543)
 * leave the cloud, the rest keeps its order.  Afterwards every entry point works on the remaining
 * points, and the index lists it returns (m3d_cloud_fit, m3d_cloud_refine) keep referring to the cloud
 * AS CREATED, which is what SegmentPlaneIterative's callers need.  Clouds without normals only. */
int m3d_cloud_remove_inliers(m3d_cloud *cloud, int kind, double threshold, const double *model,
                             size_t *n_removed);
/* Sequential replay of the best-update / adaptive-stop rule (ransac.h:573-575,592-613) over
 * per-hypothesis (valid, count) records in index order; `rmse_cb` is called only for fitness ties.
 * Pure host logic, usable by distributed drivers after gathering counts.  valid == NULL: counts[i] carries
 * the valid flag in bit 31 (m3d_cloud_score_shard's packed form). */
typedef double (*m3d_rmse_fn)(void *user, size_t hypothesis_index);
typedef struct m3d_replay_state {
    double best_fitness, best_rmse;
    int64_t best_index;
    uint64_t best_count; /* inlier_num of the best hypothesis */
    uint64_t count, current_iteration, iterations;
    int32_t best_rmse_known, stopped;
} m3d_replay_state;
void m3d_replay_init(m3d_replay_state *st);
void m3d_replay_chunk(m3d_replay_state *st, size_t n_points, int kind, size_t max_iteration,
                      double probability, size_t begin, size_t end, const uint8_t *valid,
                      const uint32_t *counts, m3d_rmse_fn rmse_cb, void *user);

/* ---- segmentation::SegmentPlaneIterative, src/iterative_plane_segmentation.cpp:8-39 ----------- */
/* planes: 4 x max_clusters; cluster_offsets: max_clusters+1; cluster_indices: capacity n, indices
 * into the ORIGINAL cloud, ascending inside a cluster (SelectByIndex order).  Round r seeds its
 * sampler with *seed + r.  Returns 1; 0 when N < 3 (reference: warning + empty result, :13-17);
 * 2 when a round found no inlier (the reference would spin forever, :29-35). */
int m3d_segment_plane_iterative(const double *xyz, size_t n, double threshold, int max_iteration,
                                double min_ratio, const uint64_t *seed, int device,
                                size_t max_clusters, double *planes, size_t *cluster_offsets,
                                size_t *cluster_indices, size_t *n_clusters);
/* The same call returning the clusters' POINTS as well -- what the reference's function actually returns
 * (`cluster = pcd_copy->SelectByIndex(inliers)`, iterative_plane_segmentation.cpp:32,34: vector<pair<Vector4d, PointCloud>>).
 * cluster_points: capacity n x 3 doubles, may be NULL; receives the xyz of cluster_indices[i] at [3 i, 3 i + 3), gathered
 * on the device from the resident cloud and shipped in one copy (on the host SelectByIndex is a strided walk over the whole
 * input per big cluster: 135 ms of a 166 ms call on a 10 M-point room). */
int m3d_segment_plane_iterative_clouds(const double *xyz, size_t n, double threshold, int max_iteration,
                                       double min_ratio, const uint64_t *seed, int device, size_t max_clusters,
                                       double *planes, size_t *cluster_offsets, size_t *cluster_indices,
                                       double *cluster_points, size_t *n_clusters);

/* ---- registration::LeastSquareSolver::Solve, src/transform_estimation.cpp:49-66 (Eigen::umeyama) */
/* src, dst: n x 3.  T: row-major 4x4. */
int m3d_kabsch(const double *src, const double *dst, size_t n, int scaling, int device,
               double T[16]);

/* ---- registration::RANSACSolver::Solve, src/transform_estimation.cpp:124-164 ------------------ */
typedef struct m3d_reg_stats {
    double fitness, inlier_rmse;
    uint64_t validations;
    int64_t iterations, best_index, est_k;
    double ms_total;
    uint64_t ties;             /* equal-fitness comparisons decided during the replay (hypotheses the validation dropped early
                                * on their partial count or sum never get that far) */
    uint64_t exact_rmse_evals; /* of which needed the serial-order sum of squared distances */
    uint64_t lds_wave_hypotheses;    /* round 6: (256-point source tile, hypothesis) pairs whose record the validation's candidate cache made EXACT -- every
                                      * query answered from registers under a certificate (m3d_config.reg_cache; the field's name is round 2's, the
                                      * slot keeps the layout) */
    uint64_t global_wave_hypotheses; /* round 6: pairs it left as BOUNDS (count from above, sum from below); the neighbour-list walk evaluated those of
                                      * them whose hypothesis the pruning could not drop on the bounds (0 / 0: the cache did not run) */
    uint64_t nn_fp32_screen;         /* 1: the validation's neighbour search ran behind the fp32 screen (m3d_config.reg_fp32_screen and a
                                      * grid the screen admits: cell edge and offsets within fp32's reach) */
    uint64_t nn_screen_fallbacks;    /* queries whose runner-up lay within the rounding bound of the winner: decided by the fp64 walk */
} m3d_reg_stats;
/* corr_src/corr_dst: m index pairs (the std::pair<vector<size_t>,vector<size_t>> of the reference).
 * confidence: Open3D RANSACConvergenceCriteria::confidence_ (the reference always uses the default
 * 0.999, transform_estimation.cpp:160-161).  T: row-major 4x4.  stats may be NULL: the call then returns the pose only (all
 * RANSACSolver::Solve returns) and skips the pass that forms the winner's deterministic inlier_rmse. */
int m3d_registration_ransac(const double *src, size_t n_src, const double *dst, size_t n_dst,
                            const size_t *corr_src, const size_t *corr_dst, size_t m,
                            double threshold, int max_iter, double edge_length_threshold,
                            double confidence, const uint64_t *seed, int device, double T[16],
                            m3d_reg_stats *stats);

/* The same RANSAC as a session, cut into the steps a multi-GPU driver needs (hypotheses sharded, clouds and
 * grid replicated on every GPU).  Every rank creates an identical session (same seed):
 *   m3d_reg_begin_chunk   draws the triples of the next chunk of iterations, 3-point Kabsch + checkers;
 *                         *n_survivors = hypotheses that need validation.  Return 1 = chunk ready,
 *                         0 = the loop is over (call m3d_reg_finish), <0 error.
 *   m3d_reg_validate      (inlier count, sum of squared nearest distances) of survivors [s_begin, s_end) --
 *                         this rank's shard; s_begin must be a multiple of 64.
 *   m3d_reg_replay        sequential best-update / est_k rule over the chunk, given the records of ALL its
 *                         survivors (after the all-gather); identical inputs -> identical state on all ranks.
 * m3d_registration_ransac is exactly this loop with one shard. */
typedef struct m3d_reg m3d_reg;
m3d_reg *m3d_reg_create(const double *src, size_t n_src, const double *dst, size_t n_dst,
                        const size_t *corr_src, const size_t *corr_dst, size_t m, double threshold,
                        int max_iter, double edge_length_threshold, double confidence,
                        const uint64_t *seed, int device);
void m3d_reg_destroy(m3d_reg *reg);
int m3d_reg_begin_chunk(m3d_reg *reg, size_t *n_survivors);
int m3d_reg_validate(m3d_reg *reg, size_t s_begin, size_t s_end, uint32_t *counts, double *sums);
int m3d_reg_replay(m3d_reg *reg, const uint32_t *counts, const double *sums);
int m3d_reg_finish(m3d_reg *reg, double T[16], m3d_reg_stats *stats);

/* ---- misc3d::common::EstimateNormalsFromMap, src/normal_estimation.cpp:180-207 ---- */
/* xyz: h x w x 3 doubles, the organised point map row by row (pc->points_ with shape = (w, h)); a pixel is
 * valid iff its z is not NaN (:90).  k: window half-size ((2k+1)^2 neighbourhood, python default 5).
 * normals: h x w x 3 out; unit eigenvector of the smallest covariance eigenvalue, flipped towards view_point
 * (:165-171); NaN for invalid pixels (the reference leaves those uninitialised).  *ms_device (may be NULL):
 * device time of the three kernels.  The caller checks points == w * h like :187-191. */
int m3d_normals_from_map(const double *xyz, uint32_t w, uint32_t h, uint32_t k, const double view_point[3],
                         int device, double *normals, double *ms_device);

/* ---- misc3d::features::DetectBoundaryPoints, src/boundary_detection.cpp:68-113 -------- */
/* normals: n x 3 or NULL (then estimated from the same neighbourhood, :82-84).  search: 0 = KDTreeSearchParamKNN
 * (max_nn <= 128; radius ignored), 1 = KDTreeSearchParamRadius (radius), 2 = KDTreeSearchParamHybrid (radius,
 * max_nn <= 128) -- the python default is Hybrid(0.01, 30),
 * python/py_features.cpp:19.  angle_threshold in degrees (default 90).  indices: capacity n, written in
 * ascending order (the reference's order depends on thread timing); *k = their number. */
int m3d_detect_boundary_points(const double *xyz, const double *normals, size_t n, int search, double radius,
                               int max_nn, double angle_threshold_deg, int device, size_t *indices, size_t *k);

/* ---- point-to-point ICP refinement of the RANSAC pose ----------------------- */
/* open3d::pipelines::registration::RegistrationICP(source, target, max_correspondence_distance, init,
 * TransformationEstimationPointToPoint(), ICPConvergenceCriteria(relative_fitness, relative_rmse,
 * max_iteration)) as the reference's examples call it right after RANSACSolver::Solve
 * (examples/cpp/transform_estimation.cpp:82-86; defaults 1e-6, 1e-6, 30).  T_init: row-major 4x4 or NULL =
 * identity.  correspondences (n_src entries, may be NULL): target index of every source point, -1 = none
 * (Open3D's correspondence_set_). */
typedef struct m3d_icp_stats {
    double fitness;            /* correspondences / n_src */
    double inlier_rmse;        /* sqrt(sum d^2 / correspondences) */
    uint64_t correspondences;
    int32_t iterations;        /* ICP iterations executed */
    int32_t converged;         /* 1 = both relative criteria met before max_iteration */
    double ms_total;
} m3d_icp_stats;
int m3d_registration_icp(const double *src, size_t n_src, const double *dst, size_t n_dst,
                         double max_correspondence_distance, const double *T_init, int max_iteration,
                         double relative_fitness, double relative_rmse, int device, double T[16],
                         m3d_icp_stats *stats, int64_t *correspondences);

/* open3d::pipelines::registration::GetInformationMatrixFromPointClouds(source, target, max_dist, T): the
 * acceptance test of ReconstructionPipeline::GlobalRegistration, src/pipeline.cpp:818-824.
 * info: 6 x 6 row-major; info[35] == *n_correspondences (may be NULL). */
int m3d_information_matrix(const double *src, size_t n_src, const double *dst, size_t n_dst,
                           double max_correspondence_distance, const double *T, int device, double info[36],
                           uint64_t *n_correspondences);

/* ---- ReconstructionPipeline::GlobalRegistration (Ransac method), src/pipeline.cpp:790-828, and its caller's shape: one
 * std::thread per fragment pair, src/pipeline.cpp:428-439 ----------------------------------------------
 * match (ANNMatcher::Match, :800-802) -> RANSACSolver(1.4 voxel_size).Solve (:806-807) -> pose.isIdentity(1e-8) ? (true,
 * pose, I6) (:814-816) -> GetInformationMatrixFromPointClouds (:818-820) -> info(5,5) / min(Ns, Nt) < 0.3 ? (false, pose,
 * I6) : (true, pose, info) (:821-825).  The whole pair runs on ONE lane of the device; the two clouds are uploaded once and
 * serve the solver and the information matrix.
 * feat_*: one descriptor of `dim` contiguous doubles per point (Feature::data_, Eigen dim x N column-major).
 * max_iter / edge_length_threshold: RANSACSolver's (defaults 100000 / 0.9, transform_estimation.h:121-123); confidence:
 * Open3D RANSACConvergenceCriteria's (0.999).  seed: NULL = std::random_device.
 * Returns M3D_OK = (true, ...), M3D_FALSE = (false, pose, I6), < 0 = the reference throws (fewer than 3 points) / device error.
 * T: 4 x 4 row-major, info: 6 x 6 row-major; both always written (identity on error). */
typedef struct m3d_global_reg_stats {
    uint64_t n_matches;              /* mutual pairs the matcher returned */
    uint64_t n_info_correspondences; /* info(5,5): correspondences within 1.4 voxel_size under the pose (0 when not evaluated) */
    int32_t identity_shortcut;       /* 1: pose.isIdentity(1e-8) ended the call (:814-816) */
    int32_t device, lane;            /* where the pair ran */
    int32_t reserved_;
    double ms_match, ms_ransac, ms_info, ms_total;   /* host clock */
    m3d_reg_stats ransac;
} m3d_global_reg_stats;
int m3d_global_registration(const double *src, size_t n_src, const double *dst, size_t n_dst, const double *feat_src,
                            const double *feat_dst, int dim, double voxel_size, int max_iter, double edge_length_threshold,
                            double confidence, const uint64_t *seed, int device, double T[16], double info[36],
                            m3d_global_reg_stats *stats);
/* BuildPoseGraphForScene's loop over fragment pairs (src/pipeline.cpp:428-439: one std::thread per pair, joined): pair k
 * runs on devices[k % n_dev]; every device works on up to `inflight` (<= 0: m3d_config.lanes) of its pairs at a time, each on a
 * lane of its own, so one pair's uploads and host-side steps (cross-check, RANSAC replay) run under another's kernels.
 * Pairs are independent: no collective, results identical to n_pairs calls of m3d_global_registration with the same seeds.
 * Every pair's rc / T / info / stats are written; the call returns the first failed pair's (negative) code, else M3D_OK. */
typedef struct m3d_fragment_pair {
    const double *src, *dst;           /* n x 3 */
    size_t n_src, n_dst;
    const double *feat_src, *feat_dst; /* n x dim */
    uint64_t seed;
    int32_t has_seed;                  /* 0: std::random_device */
    int32_t rc;                        /* out: M3D_OK accepted / M3D_FALSE rejected / < 0 error */
    double T[16], info[36];            /* out */
    m3d_global_reg_stats stats;        /* out */
} m3d_fragment_pair;
int m3d_global_registration_batch(m3d_fragment_pair *pairs, size_t n_pairs, int dim, double voxel_size, int max_iter,
                                  double edge_length_threshold, double confidence, const int *devices, int n_dev,
                                  int inflight);

/* The same loop as the reference holds its data (preprocessed_fragment_lists_ / fragment_features_ + pairs (s, t) by index,
 * src/pipeline.cpp:415-440): every fragment a device needs is uploaded to it ONCE, by the first pair that asks for it, and
 * stays resident for the call -- a fragment of the all-pairs loop is otherwise uploaded n - 1 times.  Pair k runs on
 * devices[k % n_dev], `inflight` pairs at a time per device.  Results: those of m3d_global_registration on the same arrays
 * and seeds.  Returns the first failed pair's (negative) code, else M3D_OK; every pair carries its own rc. */
typedef struct m3d_fragment_view {
    const double *xyz;  /* n x 3 */
    const double *feat; /* n x dim */
    size_t n;
} m3d_fragment_view;
typedef struct m3d_pair_result {
    int32_t s, t;                /* in: fragment indices (source, target) */
    int32_t has_seed;            /* in: 0 = std::random_device */
    int32_t rc;                  /* out: M3D_OK accepted / M3D_FALSE rejected / < 0 error */
    uint64_t seed;               /* in */
    double T[16], info[36];      /* out */
    m3d_global_reg_stats stats;  /* out */
} m3d_pair_result;
int m3d_register_fragment_pairs(const m3d_fragment_view *frags, size_t n_frags, int dim, m3d_pair_result *pairs,
                                size_t n_pairs, double voxel_size, int max_iter, double edge_length_threshold,
                                double confidence, const int *devices, int n_dev, int inflight);

/* ---- registration::ANNMatcher::Match, src/correspondence_matching.cpp:52-84 ------------------- */
/* feat_*: Eigen MatrixXd dim x N column-major = N descriptors of dim contiguous doubles.
 * method: 0 FLANN, 1 ANNOY (correspondence_matching.h MatchMethod); both run the exact mutual
 * nearest-neighbour search on the GPU.  out_*: capacity n_src.  *k = number of matches. */
int m3d_match_mutual_nn(const double *feat_src, size_t n_src, const double *feat_dst, size_t n_dst,
                        int dim, int method, int n_trees, int device, size_t *out_src,
                        size_t *out_dst, size_t *k);
/* Diagnostics: number of queries of the calling thread's last m3d_match_mutual_nn call whose reduced-precision
 * screen was inconclusive (candidate list overflow / fp32 range) and that were redone by exact brute force. */
uint64_t m3d_match_last_fallbacks(void);

/* ---- multi-GPU: hypotheses sharded, points replicated ----------------------------------
 * The hypothesis loop of ransac.h:571-613 carries one dependency across iterations, the best-model update
 * (:592-613).  Sharded, every rank (one GPU each) holds a replica of the cloud and an identically seeded sampler,
 * scores a contiguous slice of every window of the ONE hypothesis stream, and a single all-gather per window
 * exchanges the 4-byte (MinimalFit's return << 31 | inlier count) records; every rank then replays the same
 * sequence, so best hypothesis, iteration count, inlier list and parameters are identical on every rank and equal
 * to the one-GPU result for any number of ranks.  No point ever crosses the links.
 *
 * m3d_comm is the exchange.  Three ways to get one:
 *   m3d_comm_create_rccl   one process per GPU: rank 0 calls m3d_comm_unique_id, the launcher's own channel
 *                          (torch.distributed store, MPI_Bcast, a file) hands the 128 bytes to every rank, every
 *                          rank calls m3d_comm_create_rccl(id, world, rank, device) -- collective, like
 *                          ncclCommInitRank.  The all-gather then runs as ncclAllGather on the library's stream,
 *                          in place on the device record array (RCCL over xGMI; librccl is bound at run time).
 *   m3d_comm_create_host   the caller supplies the all-gather over host buffers (MPI_Allgather, gloo, tests).
 *   m3d_comm_create_local  `world` communicators for `world` threads of this process, one device each
 *                          (m3d_segment_plane_iterative_multi uses it).
 * A communicator is used by one thread at a time; all ranks must make the same sequence of sharded calls. */
typedef struct m3d_comm m3d_comm;
#define M3D_COMM_ID_BYTES 128
int m3d_comm_unique_id(uint8_t id[M3D_COMM_ID_BYTES]);
m3d_comm *m3d_comm_create_rccl(const uint8_t id[M3D_COMM_ID_BYTES], int world, int rank, int device);
/* recv: world x bytes_per_rank, rank-major.  Return 0 on success. */
typedef int (*m3d_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
m3d_comm *m3d_comm_create_host(int world, int rank, m3d_allgather_fn fn, void *user);
int m3d_comm_create_local(int world, m3d_comm **comms /* world handles out */);
void m3d_comm_destroy(m3d_comm *comm);
int m3d_comm_world(const m3d_comm *comm);
int m3d_comm_rank(const m3d_comm *comm);
uint64_t m3d_comm_collectives(const m3d_comm *comm); /* exchanges made so far */

/* m3d_cloud_fit (RANSAC::FitModel, ransac.h:482-516) with the hypothesis loop sharded over the ranks of `comm`.
 * Every rank passes its own replica of the cloud and the SAME kind / threshold / max_iteration / probability;
 * seed == NULL: rank 0 draws one from std::random_device and shares it.  Outputs are written on every rank.
 * comm == NULL is m3d_cloud_fit.  stats->hypotheses_scored counts this rank's share. */
int m3d_cloud_fit_sharded(m3d_cloud *cloud, m3d_comm *comm, int kind, double threshold, size_t max_iteration,
                          double probability, const uint64_t *seed, double *params, size_t *inliers,
                          size_t *n_inliers, m3d_stats *stats);
/* m3d_segment_plane_iterative with every round's hypothesis loop sharded: each rank uploads the same cloud to its
 * own `device`, removes the same inliers from its replica after every round (no point traffic), and returns the
 * same clusters. */
int m3d_segment_plane_iterative_sharded(const double *xyz, size_t n, double threshold, int max_iteration,
                                        double min_ratio, const uint64_t *seed, int device, m3d_comm *comm,
                                        size_t max_clusters, double *planes, size_t *cluster_offsets,
                                        size_t *cluster_indices, size_t *n_clusters);
/* The same from ONE process that drives n_dev GPUs (the `devices[], n_dev` form): a thread and a
 * replica per device, records exchanged through host memory.  devices must be distinct ordinals.  n_dev == 1 is
 * m3d_segment_plane_iterative on devices[0]. */
int m3d_segment_plane_iterative_multi(const double *xyz, size_t n, double threshold, int max_iteration,
                                      double min_ratio, const uint64_t *seed, const int *devices, int n_dev,
                                      size_t max_clusters, double *planes, size_t *cluster_offsets,
                                      size_t *cluster_indices, size_t *n_clusters);
int m3d_fit_multi(int kind, const double *xyz, const double *normals, size_t n, double threshold,
                  size_t max_iteration, double probability, const uint64_t *seed, const int *devices, int n_dev,
                  double *params, size_t *inliers, size_t *n_inliers, m3d_stats *stats);
/* m3d_registration_ransac with the validation of every chunk's surviving hypotheses sharded (contiguous runs of
 * 64-hypothesis groups per rank, one all-gather of (count, sum d^2) per chunk, identical replay everywhere). */
int m3d_registration_ransac_sharded(const double *src, size_t n_src, const double *dst, size_t n_dst,
                                    const size_t *corr_src, const size_t *corr_dst, size_t m, double threshold,
                                    int max_iter, double edge_length_threshold, double confidence,
                                    const uint64_t *seed, int device, m3d_comm *comm, double T[16],
                                    m3d_reg_stats *stats);

/* ---- tunables ---------------------------------------------------------------------------------------------------
 * Every knob of the library, read ONCE from the environment on first use (variable names in brackets) and
 * replaceable at run time.  None changes a result: they select between code paths that produce identical output
 * (the test-suite runs them against each other) or set launch geometry.  m3d_set_config may be called while other threads
 * compute: a call works with the settings as they were when it took its lane (a snapshot per call), the new ones apply to calls
 * that start afterwards.  Returns M3D_ERR_INVALID_ARG for score_mfma / score_waves4 / compact_one_pass != 0 on a build without
 * those kernels (m3d_bench_experimental() == 0). */
typedef struct m3d_config {
    int32_t dense_scoring;          /* [M3D_DENSE=1]        1: score every (tile, hypothesis) pair (score_k) instead of the culled path */
    int32_t speculative_refine;     /* [M3D_SPEC=0]         default 1: probability-1 fits start RefineModel on the device's own pick */
    int32_t lead_hypotheses;        /* [M3D_LEAD]           default 128 (multiple of 64): hypotheses counted first for the pruning incumbent */
    int32_t score_groups_per_block; /* [M3D_GPB]            default 8: 64-hypothesis groups per scoring workgroup (1..64); the fp32-screen kernels take twice that for
                                       windows of 192 groups and more, and at most 16 */
    int32_t score_min_workgroups;   /* [M3D_SCORE_MIN_WGS]  default 8192: small chunks are cut finer to reach this many workgroups */
    int32_t dense_workgroups;       /* [M3D_SCORE_WGS]      default 8192: workgroup target of the dense kernel */
    int32_t reg_neighbour_lists;    /* [M3D_REG_NL=0]       default 1: per-cell 3x3x3 neighbour lists for the registration validation */
    int32_t reg_prune;              /* [M3D_REG_PRUNE=0]    default 1: bound-and-prune of validations against earlier chunks */
    int32_t match_brute;            /* [M3D_MATCH_BRUTE=1]  1: fp64 brute-force matcher (no screen) */
    int32_t match_fp32_screen;      /* [M3D_MATCH_SCREEN=fp32] 1: fp32 VALU screen instead of the split-fp16 MFMA screen */
    int32_t pool_limit_mb;          /* [M3D_POOL_MB]        default 4096: released device blocks parked per device for re-use (0 = none) */
    int32_t kernel_timing;          /* [M3D_KERNEL_TIMING=1] default 0; 1: HIP events attached to every scoring launch (hipExtLaunchKernel: the
                                       launch's own start / stop times, no barrier packets) fill m3d_stats.ms_score_kernel /
                                       score_launches (bench.py switches it on: ~4 us per C2 fit); ms_score becomes a host clock around the scoring phase */
    int32_t reg_sorted_lists;       /* [M3D_REG_SORTED=0]   default 1: x-sorted neighbour lists, side columns cut off by the x-distance */
    int32_t score_fp32_screen;      /* [M3D_SCORE_SCREEN=0] default 1: planes and spheres are counted by score_screen_k (packed-fp32 screen with a
                                       rounding bound in front of the exact fp64 test: identical counts); 0: fp64 only (score_mask_k) */
    int32_t cull_fp32;              /* [M3D_CULL_FP32=0]    default 1: the box tests of the culled path run in fp32 with outward-rounded margins
                                       (cull_tiles32_k, and cull_hyp32_k for windows of 128 groups and more: conservative, identical results); 0: fp64 box tests
                                       (cull_tiles_k); 2: fp32, always with a lane per tile (cull_tiles32_k: the tests' switch) */
    int32_t reg_fp32_screen;        /* [M3D_REG_SCREEN=0]   default 1: the nearest-neighbour search of the registration validation finds its
                                       candidate in fp32 on 8-byte list entries (16-bit fixed-point coordinates over the cell's 3-cell
                                       block + the entry's position, rounding bound) and evaluates the winner in fp64; a query whose
                                       runner-up is within the bound takes the fp64 walk: identical distances */
    int32_t sorted_tombstones;      /* [M3D_TOMBSTONES=0]   default 1: a segmentation round that removes a sliver of the cloud kills its inliers
                                       in place in the Hilbert-sorted copy (x = NaN: never an inlier; the screen masks the lane) instead
                                       of partitioning the copy; a real compaction follows when an eighth of the copy is dead */
    int32_t score_mfma;             /* [M3D_SCORE_MFMA=1]   EXPERIMENTAL BUILDS ONLY (make DEFS=-DM3D_EXPERIMENTAL; elsewhere the slot is kept for the layout and
                                       must be 0).  default 0; 1: plane hypotheses are counted by score_mfma_k -- the screen's two
                                       one-sided values T - S and T + S on the matrix pipe (split-fp16 operands, rounding bound, exact
                                       fp64 recount of the undecided pairs: identical counts) instead of score_screen_k (packed fp32
                                       VALU); measured slower on C2 as it stands (m3d_score_mfma.hip, STATUS) */
    int32_t score_mfma_groups;      /* [M3D_MFMA_GPB]       default 64: 64-hypothesis groups per workgroup of score_mfma_k (1..64) */
    int32_t score_waves4;           /* [M3D_SCORE_WAVES4=1]  EXPERIMENTAL BUILDS ONLY (must be 0 elsewhere).  default 0; 1: scoring windows of 24 groups and more run score_screen4_k -- four-wave
                                       workgroups over up to score_waves4_groups groups of a tile that share ONE compacted list of the
                                       surviving hypotheses and take its batches of 64 in turn (VERDICT r3 item 2's decomposition: measured
                                       4 % slower than one wave per (tile, 8 groups) on C2, equal on C3 -- profiles/r04_score_waves4.txt) */
    int32_t score_waves4_groups;    /* [M3D_WAVES4_GPB]     default 64 (1..64) */
    int32_t score_phases;           /* [M3D_SCORE_PHASES]   default -1 = 3 for cylinders, 0 for planes and spheres (measured: m3d_cull_kernels.hip,
                                       "PHASED scoring"); 2 / 3 (every kind): a window that has an incumbent is counted on a quarter of the
                                       tiles, re-pruned with what every hypothesis collected (count so far + 512 per tile it can still touch
                                       against the incumbent: exact), (3: counted on a second quarter, re-pruned,) and only the survivors
                                       see the rest; 0: one launch over all tiles */
    int32_t compact_one_pass;       /* [M3D_COMPACT_ONE_PASS=1]  EXPERIMENTAL BUILDS ONLY (must be 0 elsewhere).  default 0; 1: RefineModel's / a removal's ordered compaction of up to 1024 tiles
                                       (2 M points) is ONE launch -- a workgroup publishes its tile's count and waits for the counts of the
                                       tiles below it (compact_write_k, ONE) -- instead of a counting launch and a writing launch.  Same
                                       output, position for position; measured SLOWER (a count crossing the XCDs' L2s costs more than the
                                       launch boundary it replaces: profiles/r04_compact_one_pass.txt) */
    int32_t plane_bound;            /* [M3D_PLANE_BOUND=0]  default 1: plane fits -- and since round 5 cylinder fits: the shell over a tile is a slab up to a
                                       sagitta -- with an incumbent prune with a per-tile HISTOGRAM upper bound of
                                       every (tile, hypothesis) pair's inlier count (tile_frames_k / plane_bound_k, m3d_bound.hip) instead of
                                       512 per touched tile, where that pays (windows of >= 8192 hypotheses on tiles x hypotheses >= 1.5e7, measured: m3d_fit.cpp
                                       bound_pays; 2: whatever the size, and sphere fits too); same results, fewer hypotheses counted point by point.  (Fields are
                                       only ever appended, the offsets of existing fields do not move -- ADVICE r3) */
    int32_t lanes;                  /* [M3D_LANES]          default 4 (1..8): calls a device runs side by side -- every lane has its own streams and
                                       scratch, a call holds one from entry to return, host threads are dealt lanes in the order they arrive
                                       (a single-threaded caller lives on lane 0); 1 = one call at a time per device (rounds 1-4) */
    int32_t wait_spin_us;           /* [M3D_SPIN_US]        default 500: the end of a fit's device work is waited for by polling a completion word in
                                       page-locked memory (the runtime's wait wakes the caller 10-20 us late) for at most this many
                                       microseconds, then by hipStreamSynchronize (a blocked thread, no core burnt: long calls, hosts with
                                       more fitting threads than cores); 0 = the runtime's wait only */
    int32_t prestream;              /* [M3D_PRESTREAM=0]    default 1: MinimalFit + box tests of a fit's chunk k + 1 run on a second stream under the
                                       scoring launches of chunk k; 0: everything on the lane's main stream (same records) */
    int32_t chunk_cap;              /* [M3D_CHUNK_CAP]      default 24576 (1024..262144, multiple of 64): hypotheses per chunk of the culled path */
    int32_t first_chunk;            /* [M3D_FIRST_CHUNK]    default 2048 (0 = off): length of the short first chunk of a fit of several chunks (the
                                       incumbent that prunes the rest) */
    int32_t reg_cells_per_radius;   /* [M3D_REG_K]          default 4 (1..16): cells per search radius of the registration validation's grid for a target of
                                       200 000 points and more; smaller targets get coarser cells (cube root of the size ratio) */
    int32_t match_pipeline;         /* [M3D_MATCH_PIPELINE] default 1: a match of two large host matrices (>= 65 536 rows each) uploads them in slices (two of
                                       the queries, four of the database) and scans block (i, j) while the next slice is on the link (same result:
                                       tests/test_gpu_match_sliced.py) -- when the call is alone on the device (other calls' kernels fill the gap
                                       anyway); 0: both matrices uploaded first; 2: sliced whatever the size and the company (the tests' switch) */
    int32_t reg_cache;              /* [M3D_REG_CACHE]      default 1: a registration RANSAC that validates many hypotheses keeps, per source point, the 32
                                       target points nearest to its position under the incumbent pose in registers and answers a hypothesis'
                                       nearest-neighbour queries from them wherever a certificate holds (every other target point provably
                                       farther); the other (tile, hypothesis) pairs take the neighbour-list walk.  Same counts, sums and pose
                                       (m3d_reg_cache.hip).  0: the walk only; 2: the cache as soon as an incumbent exists (the tests' switch) */
    int32_t device_aliases;         /* [M3D_DEVICE_ALIASES] default 0; N > 0: device ordinals 0 .. max(N, physical devices) - 1 are valid, ordinal d living
                                       on physical device d % physical -- each with its own lanes, streams, scratch, free lists and
                                       resident tables, as a further physical device would have.  For executing the multi-device entry
                                       points (devices[] forms, m3d_fit_multi, m3d_register_fragment_pairs) on a box with fewer GPUs than
                                       the code path wants: it exercises the dealing, the per-device state and the in-process exchange,
                                       NOT peer traffic or RCCL across devices.  m3d_device_count() then returns the logical count (<= 16) */
    int32_t lanes_eager;            /* [M3D_LANES_EAGER]    default 2 (1..lanes): lanes whose compute streams a device's FIRST call creates; the others appear when
                                       a call needs them.  Streams take hardware queues in creation order, and a lane created after another lane's
                                       copy / pre stream shares a queue with it (its calls take turns with that lane's): a process that runs many
                                       calls side by side (m3d_cloud_fit_batch over four lanes, fragment pairs in flight) sets 4 before its first
                                       call; a single-threaded caller keeps 2 (its fits of several chunks are 10-20 % faster that way) */
} m3d_config;
void m3d_get_config(m3d_config *out);
int m3d_set_config(const m3d_config *in);

/* ---- misc -------------------------------------------------------------------------------------- */
const char *m3d_last_error(void); /* thread-local; reference message text for M3D_ERR_* */
int m3d_device_count(void);       /* 0 when no HIP device is usable */
const char *m3d_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MISC3D_AMD_H */
