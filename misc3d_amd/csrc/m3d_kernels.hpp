// m3d_kernels.hpp -- launch interface of the gfx950 kernels (m3d_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace m3d {

constexpr int kModelStride = 8;     // doubles per hypothesis record (64 B: one s_load_dwordx16)
constexpr int kScoreP = 8;          // points per lane held in VGPRs by the scoring kernels
constexpr int kScoreBlock = 256;    // threads per scoring workgroup (4 waves)
constexpr int kScoreTile = kScoreBlock * kScoreP;  // points per scoring workgroup
constexpr int kCompactTile = 2048;  // points per compaction workgroup

// -DM3D_EXPERIMENTAL (make DEFS=-DM3D_EXPERIMENTAL): the three variants round 4 built, measured and refuted are compiled in
// and m3d_config.score_mfma / score_waves4 / compact_one_pass switch them on -- score_mfma_k (m3d_score_mfma.hip: the planes'
// screen on the matrix pipe, profiles/r04_score_mfma.txt), score_screen4_k (four-wave scoring workgroups,
// profiles/r04_score_waves4.txt), compact_write_k<.., ONE> (the ordered compaction as one launch,
// profiles/r04_compact_one_pass.txt).  The product library is built without them: the three switches are then ignored.
#ifdef M3D_EXPERIMENTAL
constexpr bool kExperimentalBuild = true;
#else
constexpr bool kExperimentalBuild = false;
#endif

// SoA view of a resident cloud.  Arrays are padded to a multiple of kScoreTile with NaN so the
// scoring kernels need no bounds checks (a NaN coordinate is never an inlier).
struct CloudView {
    const double* x;
    const double* y;
    const double* z;
    const double* nx;  // may be null
    const double* ny;
    const double* nz;
    uint32_t n;
    uint32_t n_pad;
};

// K0: AoS (n x 3) -> SoA, NaN padding up to n_pad.
void launch_aos_to_soa(const double* aos, double* x, double* y, double* z, uint32_t n,
                       uint32_t n_pad, hipStream_t s);

// K1: minimal fit of `h_count` hypotheses from their sample indices.  Writes per hypothesis
//   score[h*8 ..]  = scoring record (plane: a,b,c,d,T | sphere: cx,cy,cz,s_lo,s_hi |
//                    cylinder: cx,cy,cz,refx,refy,refz,t_lo,t_hi)
//   params[h*8 ..] = reference model parameters (4 or 7 doubles)
//   valid[h]       = MinimalFit's return value
// Records in [h_count, h_pad) are filled with "no inlier" cut-offs.
// Optional extra of launch_minimal_fit on a fit's first chunk: the set-up keep_mask_k would do for the leading hypotheses
// (keep words all ones, their counter replicas and the launch's n_pair pair counters behind them cleared).
// Optional output of launch_minimal_fit: the fp32 records of the box tests (cull_tiles32_k; m3d_fp.hpp), relative to
// the cloud's origin (SortedView::origin / radius), kCull32Stride floats per hypothesis, pairwise interleaved
struct Cull32Out {
    float* out = nullptr;
    double origin[3] = {0.0, 0.0, 0.0};
    double radius = 0.0;
};
// The tombstone pass of a segmentation round (m3d_poison.hpp): the plane `model`'s inliers are killed in place in the sorted
// copy.  Launched on its own (launch_poison_plane_inliers) or as extra workgroups of the NEXT round's minimal_fit_k<0>.
struct PoisonJob {
    double* sx = nullptr;
    const double *sy = nullptr, *sz = nullptr, *boxes = nullptr;
    uint32_t n_tiles = 0;
    float* tile_f32 = nullptr;
    const double* model = nullptr;   // device: (a, b, c, d)
    double thr = 0.0, max_abs = 0.0;
    uint32_t* total = nullptr;       // device: running sum of the points killed
};
struct LeadPrep {
    uint32_t* counts_rep = nullptr;   // n_rep x rep_stride counters, then n_pair pair counters
    unsigned long long* keep = nullptr;
    uint32_t rep_stride = 0, n_rep = 0, n_lead = 0, n_pair = 0;
};
// returns whether a kernel was launched (h_pad == 0: nothing to do -- and nothing a riding PoisonJob could ride in)
bool launch_minimal_fit(int kind, const CloudView& c, const uint32_t* samples, uint32_t h_count,
                        uint32_t h_pad, double thr, double* score, double* params, uint8_t* valid,
                        hipStream_t s,
                        uint32_t* zero_u32 = nullptr /* h_pad - 1 counters cleared by the same launch */,
                        uint32_t* zero_one = nullptr /* EIGHT more words cleared by the same launch */,
                        const LeadPrep* lead = nullptr,
                        double cull_max_abs = __builtin_inf() /* SortedView::max_abs: the plane record's slot 5 receives the
                                                                 cut-off of the box tests (inf: nothing is ever culled) */,
                        const Cull32Out* cull32 = nullptr,
                        const PoisonJob* poison = nullptr /* planes: the previous round's tombstone pass rides in the launch (one
                                                             extra 64-thread workgroup per tile of the sorted copy) */,
                        uint32_t* zero_u32b = nullptr /* a second array of h_pad - 1 counters cleared by the same launch */);

// K2: inlier counting.  partial[tile * h_pad + h] = number of points of scoring tile `tile` whose
// distance to hypothesis h is < thr.  h_pad must be a multiple of 64; the hypotheses are cut into
// (at most) h_splits ranges of whole 64-groups (grid.y).
void launch_score(int kind, const CloudView& c, const double* score, uint32_t h_pad,
                  uint32_t h_splits, uint32_t* partial, hipStream_t s);
// counts[h] = sum over tiles of partial[tile][h].  counts must be zero on entry.
void launch_reduce_partials(const uint32_t* partial, uint32_t n_tiles, uint32_t h_pad,
                            uint32_t* counts, hipStream_t s);

// bounding box of the finite points and their number: out[0..2] = lo, out[3..5] = hi, out[6] = count;
// partial: scratch of kBboxPartialDoubles doubles
constexpr int kBboxPartialDoubles = 1024 * 8;
void launch_bbox(const double* x, const double* y, const double* z, uint32_t n, double* partial, double* out,
                 hipStream_t s);

// K4: ordered compaction with the reference's own distance formula (sqrt and divide per point).
// mode 0: out_idx[k]  = index (or orig[index] when orig != null) of the k-th inlier, ascending
// mode 1: out_dist[k] = distance of the k-th inlier
// mode 2: stable partition of the NON-inliers into (ox,oy,oz,oorig) (segmentation round)
// mode 3: the same for coordinates only (the Z-order sorted copy)
// scratch: one uint32 slot per compaction workgroup (ceil(n / kCompactTile)), ZERO when the buffer is allocated and whenever the
// owner's epoch counter starts over; tag = the launch's epoch (1 .. 2^20 - 1, a different one for every launch on the buffer) in
// bits 12..31 -- what the one-pass form marks a published count with (compact_write_k); tag 0: two passes.
// total[0] receives the inlier count.
// The one-pass form needs every workgroup of the launch resident at once: 5 workgroups fit a CU (92 VGPRs, 21 KB of LDS).
constexpr uint32_t kCompactOnePassMaxTiles = 1024;
constexpr uint32_t kCompactEpochs = 0xFFFFFu;
struct CompactScratch {
    uint32_t* slots = nullptr;
    uint32_t tag = 0;
};
struct PartitionOut {
    double *ox = nullptr, *oy = nullptr, *oz = nullptr;
    uint32_t* oorig = nullptr;
    uint32_t n_pad_cap = 0;
};
// out[3 i ..] = the point idx[i] of the cloud (AoS xyz), i < total
void launch_gather_points(const CloudView& c, const uint64_t* idx, size_t total, double* out, hipStream_t s);
void launch_compact(int kind, const CloudView& c, const double* model, double thr, int mode,
                    const uint32_t* orig, uint64_t* out_idx, double* out_dist, double* ox,
                    double* oy, double* oz, uint32_t* oorig, uint32_t n_pad_out,
                    const CompactScratch& scratch, uint32_t* total, hipStream_t s,
                    double* model_copy = nullptr /* device-visible (pinned host): receives the 8-double model record */,
                    double* moment_partial = nullptr /* mode 0, plane / sphere: scratch of ceil(n / kCompactTile) x 16 doubles ... */,
                    double* moment_out = nullptr /* ... and kFusedMomentDoubles doubles (device-visible host memory): GeneralFit's raw
                                                    moments over the inliers about the model record's provisional centre */,
                    uint64_t* out_idx_host = nullptr /* mode 0: page-locked host copy of the index list, written by the kernel */,
                    uint32_t* total_host = nullptr /* device-visible host word that receives total[0] as well (no copy command) */,
                    const PartitionOut* part = nullptr /* mode 0 with orig != null: the non-inliers' partition rides along */);
// moment_out layout: [0..2] sum s, [3..8] sum s s^T (xx,xy,xz,yy,yz,zz), [9..11] sum s |s|^2 (sphere), [12] inlier count;
// s = p - c0, c0 = model[4..6] (plane: the hypothesis' first sample point) or model[0..2] (sphere: the minimal centre).
constexpr int kFusedMomentDoubles = 16;
// raw moments about c0 -> what the closed forms take: mean[3] and the ten centred moments (xx,xy,xz,yy,yz,zz, sum r q
// (3), sum q with q = |r|^2, r = p - mean) -- the same quantities sum_moments_k produces
void moments_about_mean(const double* moment_out, const double c0[3], double n, double mean[3], double centred[10]);

// serial-order sum of `n[0]` doubles (EvaluateModel's `error += distance`, ransac.h:637)
void launch_serial_sum(const double* v, const uint32_t* n, double* out, hipStream_t s);

// K5/K6: deterministic tree reductions over the inlier list for GeneralFit.
// sums[0..2] = sum of x,y,z over idx (pass 1).  With mean[3] given (pass 2):
//   plane : sums[0..5]  = xx,xy,xz,yy,yz,zz of residuals               (ransac.h:178-188)
//   sphere: sums[0..9]  = xx,xy,xz,yy,yz,zz, x*q, y*q, z*q, q  with q = |residual|^2
// partial: scratch of kSumPartialDoubles doubles.
constexpr int kSumPartialDoubles = 256 * 16;
// both passes, two launches: the per-workgroup moment partials and the coordinate sums go straight to `out_host`
// (device-visible pinned memory, kGeneralFitHostDoubles doubles); general_fit_sums_finish folds them on the host
// (sums14[0..2] = sum x, y, z; sums14[4..13] = the ten centred moments listed above)
constexpr int kGeneralFitHostDoubles = 256 * 16 + 4;
void launch_general_fit_sums(const CloudView& c, const uint64_t* idx, uint32_t n_idx, double* partial_dev,
                             double* out_host, hipStream_t s);
void general_fit_sums_finish(const double* out_host, double* sums14);

// order-free sums of the inlier distances of one or two models (model_b may be null) + their inlier counts, one pass
// (first stage of the tie rule).  out_host (device-visible host memory, 4 doubles) = (count_a, sum_a, count_b, sum_b),
// stored by the launch's last workgroup; partial: kErrorSumScratchDoubles doubles; ticket: a zeroed u32 the kernel resets.
constexpr int kErrorSumScratchDoubles = 1024 * 4;
void launch_error_sum(int kind, const CloudView& c, const double* model_a, const double* model_b, double thr,
                      double* partial, uint32_t* ticket, double* out_host, hipStream_t s);

// exclusive scan of v[0..nb) in place by one workgroup; total[0] = sum
void launch_scan_blocks(uint32_t* v, uint32_t nb, uint32_t* total, hipStream_t s);

// fp64 VALU issue probe: blocks x 256 threads, each 16 * iters independent fp64 mul / add instructions
void launch_fp64_issue_probe(double* out, int blocks, int iters, hipStream_t s);

// iota for the original-index array of a segmentation run
void launch_iota(uint32_t* v, uint32_t n, hipStream_t s);

}  // namespace m3d
