// m3d_reg_kernels.hpp -- launch interface of the registration kernels (m3d_reg_kernels.hip).
#pragma once
#include "m3d_kernels.hpp"

namespace m3d {

constexpr int kRegTStride = 12;  // doubles per transformation record (rows 0..2 of the 4x4)
constexpr int kRegP = 1;         // source points per lane in reg_validate_k
constexpr int kRegTile = 256 * kRegP;
constexpr int kRegMinSplits = 40;    // hypothesis splits per source tile (XCD locality, see reg_validate_k)
constexpr int kRegPruneStride = 8;   // validation phase A runs on every 8th source tile

// Uniform grid over the target cloud.  Cell edge h = 1.001 * threshold / K (K = 4 unless the dense
// cell table would not fit), origin K+1 cells below the bounding-box minimum and K+1 empty cells above
// the maximum, so the (2K+1)^3 neighbourhood of any query cell in [K, n-1-K] stays inside the arrays
// and covers every target point closer than threshold.
constexpr int kWalkB = 2;             // sorted_walk32: list entries per side and trip
constexpr int kWalkPad = 2 * kWalkB;  // sentinels at either end of a list of GridDesc::nl32 (a side's look-ahead reaches at most
                                      // 2 kWalkB - 1 entries past the end)
struct GridDesc {
    double ox, oy, oz, inv_h, r2;
    double h2_in;  // (0.999 h)^2: a neighbour closer than this inside the 3x3x3 block is the nearest
    uint32_t nx, ny, nz;
    int K;
    uint32_t morton_bits;  // != 0: cell ids are Z-order codes of (ix,iy,iz), nx = ny = nz = 2^bits
    // optional neighbour lists (validation grid only): points of the 3x3x3 block of every cell, contiguous
    const uint32_t* nl_start = nullptr;  // ncell + 1
    const double4* nl_pts = nullptr;
    // nl_sorted != 0 (launch_nl_fill with sorted = true): every list is laid out as
    //   [the 9 cells of the query cell's own x-column, ascending x] [the 9 cells one column to the left, DESCENDING x]
    //   [the 9 cells one column to the right, ascending x],   entry 0's w = n_mid + 65536 * n_left
    // so that a query evaluates its own column, and then walks the side columns outwards from the cell only while the
    // x-distance alone is still below the best squared distance found (exact: everything skipped is farther away).
    int nl_sorted = 0;
    const uint32_t* nl_hdr = nullptr;   // nl_sorted: n_mid + 65536 * n_left per cell once more, next to nl_start (the query knows
                                         // all three column bounds after ONE round trip and can fetch from all of them at once)
    // nl_sorted, optional: the screen of the search (sorted_walk32 in m3d_reg_kernels.hip).  nl32: every list once more as
    // 8-byte entries -- x, y, z as 16-bit fixed point over the 3-cell block [-h, 2h) of the list's OWN cell, and the 16-bit
    // position of the fp64 entry in the list -- with the 27 cells merged in ascending x and pads at both ends (its own
    // layout: launch_nl32_offsets); nl_rec[cell] = (first real entry, 5 x 16-bit offsets of the first entry at or right of
    // each quarter boundary of the cell, entries in the top half of w).  The walk finds the nearest candidate and the
    // runner-up in fp32 on the quantised coordinates and evaluates the winner in fp64; a runner-up within the rounding
    // bound sends the query down the fp64 walk.
    const uint2* nl32 = nullptr;
    const uint4* nl_rec = nullptr;
    unsigned long long* nl32_fallbacks = nullptr;   // optional counter: queries the screen handed to the fp64 walk
    // optional (round 6, launch_reg_rings): per cell the Chebyshev distance in cells to the nearest occupied cell (255: none within
    // K + 1).  A query more than K rings out has no target point within the radius: the search returns at once instead of
    // scanning (2K + 1)^3 cells to find that out.
    const uint8_t* ring = nullptr;
    // optional (round 6): the cell-sorted points once more as (x, y, z, 0) records -- ONE 32-byte gather per candidate instead of
    // three 8-byte ones in the list-free search (a call that validates a handful of hypotheses never builds the lists)
    const double4* q4 = nullptr;
};



// The candidate cache of the validation (m3d_reg_cache.hip): per sorted source point, under a reference pose, its position
// xa, its nearest target points in kRegCacheTiers tiers of kRegCacheK (fp32 offsets from xa in pairs, fp64 coordinates) and per
// tier a radius R_t >= 0 that every target point outside tiers 0..t keeps from xa (R < 0: the slot holds no query).  Layouts
// (slot = tier * kRegCacheK + position): cx / cy / cz [tile][slot / 2][256], c64 [tile][slot][256], xa / ya / za [n_pad],
// R [tier][n_pad]; stats[0 / 1]: (tile, hypothesis) pairs whose record the cache made exact / a bound; stats[2]: wave-queries
// that went past tier 0.
// One ring of 32, in registers (96 VGPRs).  Measured on C4's forced run (profiles/r06_reg_cache.txt): 95 ms.  THREE rings of 32
// with the outer two streamed from memory: the pairs left without a certificate fall from 18 % to 6 %, but 182 instead of 152
// VGPRs -- two waves per SIMD instead of three -- and the streamed loads: 126 ms.  TWO rings of 16, both in registers, the outer
// skipped by a wave whose 64 queries are all settled by the inner: 94.4 ms and 9 spilled registers -- a wave rarely has all 64
// settled at 16.  The kernels keep the rings general (slot = ring * kRegCacheK + position, cumulative packing).
constexpr int kRegCacheK = 32;
constexpr int kRegCacheTiers = 1;
struct RegCache {
    float2 *cx = nullptr, *cy = nullptr, *cz = nullptr;
    double4* c64 = nullptr;
    double *xa = nullptr, *ya = nullptr, *za = nullptr;
    float* R = nullptr;
    unsigned long long* stats = nullptr;
    uint8_t* redo = nullptr;   // [n_tiles][s_pad] scratch of a launch_reg_validate call: pairs the cache could not certify
    uint32_t* pairs = nullptr;     // [n_tiles * s_pad] scratch: tile * s_pad + hypothesis of the flagged pairs a resolve launch walks
    uint32_t* n_pairs = nullptr;   // ... and their number
    const uint8_t* ring = nullptr;   // per cell of the target grid: Chebyshev distance in cells to the nearest occupied cell (launch_reg_rings)
};
void launch_reg_cache_build(const CloudView& src_sorted, const double* T_dev, const GridDesc& g, const uint32_t* cell_start,
                            const double* qx, const double* qy, const double* qz, const RegCache& c, hipStream_t s);
void launch_reg_rings(const GridDesc& g, const uint32_t* cell_start, uint8_t* ring /* ncell */, uint8_t* tmp /* 2 ncell */, int rounds,
                      hipStream_t s);
void launch_reg_validate_cached(const CloudView& src, const double* Ts, uint32_t s_pad, uint32_t per_split, uint32_t nsplit,
                                uint32_t slots, const GridDesc& g, const RegCache& c, uint32_t* partial_cnt, double* partial_sum,
                                uint32_t res_mask, uint32_t n_tiles, const uint8_t* keep, uint32_t tiles, uint8_t* redo,
                                hipStream_t s);

void launch_kabsch3_check(const CloudView& src, const CloudView& dst, const uint32_t* corr_src,
                          const uint32_t* corr_dst, const uint32_t* triples, uint32_t h_count,
                          double edge_thr, double dist_thr, double* T12, uint8_t* pass, hipStream_t s);
void launch_gather_T(const double* T12, const uint32_t* list, uint32_t n, uint32_t n_pad, double* out,
                     hipStream_t s);
void launch_grid_build(const CloudView& dst, const GridDesc& g, uint32_t* cell_of_point,
                       uint32_t* cell_start, uint32_t* rank /* scratch, ONE ENTRY PER POINT (dst.n) */,
                       uint32_t* tile_sums, uint32_t* total,
                       double* qx, double* qy, double* qz, hipStream_t s, uint32_t* orig = nullptr, double4* q4 = nullptr /* GridDesc::q4 */);
// the two halves of launch_grid_build; pad_to > 1 pads every group of pad_group consecutive cells to a multiple of pad_to
// slots (the caller pre-fills the arrays: the slots between the runs keep that value); total[0] = length of the layout
void launch_grid_count_scan(const CloudView& dst, const GridDesc& g, uint32_t* cell_of_point, uint32_t* cell_start,
                            uint32_t* rank, uint32_t* tile_sums, uint32_t* total, hipStream_t s, uint32_t pad_to = 0,
                            uint32_t pad_group = 1);
void launch_grid_scatter(const CloudView& dst, const uint32_t* cell_of_point, const uint32_t* cell_start,
                         const uint32_t* rank, double* qx, double* qy, double* qz, hipStream_t s, uint32_t* orig = nullptr,
                         double4* q4 = nullptr);
void launch_signal_host(uint32_t* word /* page-locked, device-visible */, uint32_t value, hipStream_t st);
void launch_fill_nan(double* p, uint32_t n, hipStream_t s);
void launch_nl_count(const GridDesc& g, const uint32_t* cell_start, uint32_t* nl_start, uint32_t* tile_sums,
                     uint32_t* total, hipStream_t s);
// sorted (orig must be null): the points of every grid cell are first ordered by x IN PLACE (qx / qy / qz), then the lists
// are written in the three-column layout of GridDesc::nl_sorted; the caller sets g.nl_sorted = 1.
void launch_nl_fill(const GridDesc& g, const uint32_t* cell_start, const uint32_t* nl_start, double* qx,
                    double* qy, double* qz, double4* nl_pts, hipStream_t s,
                    const uint32_t* orig = nullptr, bool sorted = false, uint32_t* nl_hdr = nullptr /* sorted: ncell words */,
                    uint2* nl32 = nullptr /* sorted, optional: as many entries as nl_pts */, uint4* nl_rec = nullptr /* ncell */,
                    uint32_t* overflow = nullptr /* one word, zero on entry: set when a list is too long for nl_rec */,
                    const uint32_t* nl32_start = nullptr /* launch_nl32_offsets */);
// layout of GridDesc::nl32: nl32_start[0..ncell] (scratch of ncell + 1 words), total[0] = entries incl. sentinels
void launch_nl32_offsets(const uint32_t* nl_start, uint32_t ncell, uint32_t* nl32_start, uint32_t* tile_sums,
                         uint32_t* total, hipStream_t s);
// partial_cnt / partial_sum: rows of s_pad entries, reg_validate_k writes one per 256 source points (the arrays are
// sized for kRegValidateRows per 256).  Returns the rows actually used: what launch_reduce_partials folds.
constexpr int kRegValidateRows = 4;   // rows per 256 source points
uint32_t launch_reg_validate(const CloudView& src, const double* Ts, uint32_t s_pad, const GridDesc& g,
                             const uint32_t* cell_start, const double* qx, const double* qy, const double* qz,
                             uint32_t* partial_cnt, double* partial_sum, double* sums, uint32_t best_cnt,
                             uint32_t n_points, uint8_t* keep, hipStream_t s,
                             double best_sum2 = 0.0 /* order-free sum of squared distances of the hypothesis behind best_cnt */,
                             uint32_t n_hyp = 0 /* real hypotheses among the s_pad records (0: unknown); a handful is spread over more workgroups */,
                             const RegCache* cache = nullptr /* built for a pose near the hypotheses': the pairs it certifies skip the walk (same results) */);
void launch_reg_min_d2(const CloudView& src, const double* T, const GridDesc& g, const uint32_t* cell_start,
                       const double* qx, const double* qy, const double* qz, double* best, hipStream_t s);
void launch_compact_vals(const double* v, uint32_t n, double limit, uint32_t* block_counts, uint32_t* total,
                         double* out, hipStream_t s);
void launch_corr_ratio(const CloudView& src, const CloudView& dst, const uint32_t* corr_src,
                       const uint32_t* corr_dst, uint32_t m, const double* T, double r2, uint32_t* count,
                       hipStream_t s);
void launch_icp_nn(const double* px, const double* py, const double* pz, uint32_t n, const GridDesc& g,
                   const uint32_t* cell_start, const double* qx, const double* qy, const double* qz,
                   const uint32_t* cell_orig, uint32_t* nn, double* d2, hipStream_t s);
void launch_icp_err(const double* d2, uint32_t n, double r2, double* partial, double* out2, hipStream_t s);
void launch_icp_sums(const double* px, const double* py, const double* pz, uint32_t n, const CloudView& dst,
                     const uint32_t* nn, const double* count, double* partial, double* sums, hipStream_t s);
constexpr int kBoundaryMaxNb = 128;   // neighbours kept per point in boundary_k
void launch_boundary(const CloudView& c, const GridDesc& g, const uint32_t* cell_start, const double* qx,
                     const double* qy, const double* qz, const uint32_t* cell_orig, int search, int max_nn,
                     double angle_threshold_deg, uint8_t* flag, uint8_t* overflow, hipStream_t s,
                     const uint32_t* n_sorted /* device: points the grid holds (launch_grid_build's total) */);
void launch_info_sums(uint32_t n, const CloudView& dst, const uint32_t* nn, double* partial, double* sums,
                      hipStream_t s);
void launch_icp_transform(const double* ix, const double* iy, const double* iz, uint32_t n, const double* T_dev,
                          double* ox, double* oy, double* oz, hipStream_t s);
void launch_kabsch_sums(const double* src, const double* dst, uint32_t n, double* partial, double* sums,
                        hipStream_t s);
void launch_nn(const double* q, uint32_t nq, const double* db, uint32_t ndb, int dim, uint32_t splits,
               double* best_d, uint32_t* best_i, uint32_t* nn, hipStream_t s);

// the matcher's cross-check on the device: pairs (i, nn_ab[i]) with nn_ba[nn_ab[i]] == i, in the order of i, to `out` (device-visible:
// page-locked host memory) and their number to *total; block_scratch: mutual_blocks(na) u32 on the device
uint32_t mutual_blocks(uint32_t na);
void launch_mutual_pairs(const uint32_t* nn_ab, const uint32_t* nn_ba, uint32_t na, uint32_t nb, uint32_t* block_scratch, uint32_t* total,
                         uint2* out, hipStream_t s);
// fp32-screened exact nearest neighbour for dim 33 (m3d_match_kernels.hip)
constexpr int kScreenDimP = 36;   // fp32 row stride: 33 values, pad, |row|^2, pad (144 B)
constexpr int kRing = 16;         // candidate ring entries per (query, database slice)
void launch_to_f32_33(const double* f, uint32_t n, float* out32, float* norm2, float* max_norm2, hipStream_t s);
hipError_t launch_nn_screened33(const double* q, const float* q32, const float* qn, uint32_t nq, const double* db,
                                const float* d32, uint32_t ndb, float max_dn2, uint32_t splits, uint2* ring,
                                uint32_t* ring_count, float* part_min, float* evict_min, uint32_t* overflow_list,
                                uint32_t* overflow_count, uint32_t* nn, uint32_t* h_overflow, hipStream_t s);
// MFMA (split-fp16) screen: same rings, 2 * splits slices
constexpr int kMfmaRowHalfs = 112;   // fp16 values per packed row
uint32_t mfma_tiles(uint32_t n);
uint32_t mfma_query_tiles(uint32_t n);
constexpr int kMaxAbsPartials = 2048;   // workgroups of launch_max_abs = doubles it writes
void launch_max_abs(const double* f, size_t count, double* partial /* kMaxAbsPartials doubles */, hipStream_t s);
void launch_max_f32(const float* v, uint32_t n, float* out, hipStream_t s, bool accumulate = false);   // accumulate: out keeps max(out, v[...])
// both layouts of a matrix in one pass: out_a = mfma_tiles(n) tiles in the database (role 0) layout, out_b = mfma_query_tiles(n) tiles
// in the query (role 1) layout, norm2[n]
void launch_pack_f16_both(const double* f, uint32_t n, double scale, void* out_a, void* out_b, float* norm2, hipStream_t s);
void launch_pack_f16(const double* f, uint32_t n, uint32_t n_tiles, double scale, int role, void* out, float* norm2,
                     hipStream_t s);
// both directions (a -> b and b -> a) from ONE pass over the product tiles (m3d_match_scan.hpp, RevOut)
constexpr int kMatchRevCap = 256;    // = kRevCap (m3d_match_scan.hpp): candidates kept per row of b
constexpr int kMatchRevLane = 8;     // = kRevLane: candidates per (slice, query) list of the scan
// The database splits of a scan.  Equal splits leave the chip part-empty for a whole split's length at the end (200 000 queries x 11
// splits = 8 602 workgroups over 768 resident ones: 11.2 rounds take 12) -- so the LAST full-length share of the tiles is cut into
// four splits of 1/2, 1/4, 1/8, 1/8 of it, dispatched last (blockIdx.y-major): the chip runs dry for an eighth of a split instead.
// splits = full + tail; split y begins at tile split_begin(y); the last one ends at (full + (tail ? 1 : 0)) per.
struct SplitPlan {
    uint32_t per = 1;    // tiles of a full-length split
    uint32_t full = 1;   // full-length splits
    uint32_t tail = 0;   // 0 or 4
    __host__ __device__ uint32_t count() const { return full + tail; }
    __host__ __device__ uint32_t begin(uint32_t y) const { return y <= full ? y * per : full * per + per - (per >> (y - full)); }
    __host__ __device__ uint32_t end(uint32_t y) const { return y + 1u >= full + tail ? (full + (tail ? 1u : 0u)) * per : begin(y + 1u); }
};

// One mutual search on the MFMA screen, step by step (m3d_match_kernels.hip; the host's order: m3d_registration.cpp match_mfma).
// Device pointers.  a = queries of the forward search (na x 33 fp64), b = its database.  The forward scan cuts b's tiles into `splits`
// splits of `per` tiles (two slices each: the half-waves of a query see disjoint rows); every per-(slice, query) array is laid out
// [slice][query] PER QUERY SLICE: the arrays of queries q0 .. q0 + nq - 1 start 2 splits q0 entries in, so that a launch over a
// slice of the queries addresses its own block.  max_an2 / max_bn2: device cells holding the largest |row|^2 packed so far.
struct MatchWork {
    const double* a = nullptr;
    const double* b = nullptr;
    uint32_t na = 0, nb = 0;
    void *qB_a = nullptr, *dA_a = nullptr, *qB_b = nullptr, *dA_b = nullptr;   // packed operands (role 1 / role 0 layouts)
    float *an2 = nullptr, *bn2 = nullptr, *max_an2 = nullptr, *max_bn2 = nullptr;
    SplitPlan plan;                 // forward scan: the splits of b's tiles
    uint32_t splits = 1;            // = plan.count()
    uint32_t splits_r = 1;          // reverse warm-up: splits of a's first tiles
    float* premin = nullptr;        // 2 splits x na   forward warm-up minima
    uint2* ring = nullptr;          // 2 splits x na x kRing
    uint32_t* ring_count = nullptr; // 2 splits x na
    float* part_min = nullptr;      // 2 splits x na
    float* evict_min = nullptr;     // 2 splits x na
    float* rev_premin = nullptr;    // 2 splits_r x nb   reverse warm-up minima, [slice][row] per part of b
    float* rthr = nullptr;          // sets x mfma_tiles(nb) x 40: thresholds of the reverse search, one set per query slice
    uint32_t* rperm = nullptr;      // mfma_tiles(nb) x 32: position in the packed database -> row of b (match_order_part)
    double scale = 0.0;             // the call's power-of-two scale (match_order_part packs again)
    uint32_t* rcnt = nullptr;       // nb
    uint2* rcand = nullptr;         // nb x kMatchRevCap
    uint2* rlist = nullptr;         // 2 splits x na x kMatchRevLane
    uint32_t* rlist_cnt = nullptr;  // 2 splits x na
    uint32_t *overflow_list = nullptr, *overflow_count = nullptr, *overflow_list_r = nullptr, *overflow_count_r = nullptr;
    uint32_t *nn_ab = nullptr, *nn_ba = nullptr;
};
// rows row0 .. row0 + rows - 1 of side 0 (a) / 1 (b): both packed layouts, the norms, the side's norm maximum (row0: multiple of 512)
void match_pack(const MatchWork& w, int side, uint32_t row0, uint32_t rows, double scale, hipStream_t s);
// forward warm-up of queries q0 .. q0 + nq - 1 over the first tiles of b (which must be packed)
void match_forward_warm(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s);
// thresholds of rows row0 .. of b for the scans of query slice `set`: set 0 runs the reverse warm-up (the first 1/8 of a must be packed),
// later sets re-derive the thresholds from the same minima under the norm bound of the queries packed by then
void match_reverse_thresholds(const MatchWork& w, uint32_t row0, uint32_t rows, int set, hipStream_t s);
// after set 0's thresholds of a part: its rows ordered by threshold within chunks of 1024 (w.rperm), the part's database layout
// packed again in that order, the run thresholds of the ordered rows
void match_order_part(const MatchWork& w, uint32_t row0, uint32_t rows, hipStream_t s);
// main pass: queries q0 .. q0 + nq - 1 against splits split0 .. split0 + splits - 1 (tiles below tile_end)
void match_scan(const MatchWork& w, uint32_t q0, uint32_t nq, uint32_t split0, uint32_t splits, uint32_t tile_end, int set,
                hipStream_t s);
void match_verify_forward(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s);
void match_reverse_bin(const MatchWork& w, uint32_t q0, uint32_t nq, hipStream_t s);
void match_verify_reverse(const MatchWork& w, hipStream_t s);
// exact brute force for the queries on the overflow lists; scratch: nn_exact_scratch_bytes(max of the two counters) device bytes (the
// queries' rows are then spread over 64 workgroups each), or null (one workgroup per query)
size_t nn_exact_scratch_bytes(uint32_t n_overflow);
hipError_t match_exact_fallbacks(const MatchWork& w, const uint32_t* h_overflow /* [2]: the two counters, on the host */, void* scratch, hipStream_t s);

}  // namespace m3d
