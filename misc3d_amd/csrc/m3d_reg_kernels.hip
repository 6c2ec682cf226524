// m3d_reg_kernels.hip -- gfx950 kernels of the correspondence-registration half of the hot path.
//
// Replaces (third-party code reached from src/transform_estimation.cpp:124-164 and
// src/correspondence_matching.cpp:13-84; algorithms per SURVEY.md 8a rows a17-a20):
//   kabsch3_check_k   Open3D ComputeTransformation (Eigen::umeyama on 3 pairs) + EdgeLength/Distance checkers
//   grid_*_k          Open3D KDTreeFlann(target) -> uniform grid (cell = 1.001 * threshold / K), counting sort
//   reg_validate_k    GetRegistrationResultAndCorrespondences: per surviving hypothesis, #source points whose
//                     nearest target point is at squared distance < threshold^2 and the sum of those distances
//   reg_min_d2_k      the same per point (min squared distance) for the serial-order rmse of ONE hypothesis
//   corr_ratio_k      EvaluateInlierCorrespondenceRatio
//   kabsch_sums*_k    Eigen::umeyama sums for LeastSquareSolver (n correspondences)
//   nn_k              ANNMatcher NearestSearch: exact nearest neighbour in descriptor space
#include "m3d_reg_kernels.hpp"

#include <algorithm>
#include <climits>

#include "m3d_reg_fp.hpp"
#include "m3d_eig3.hpp"

#pragma clang fp contract(off)

namespace m3d {

// ------------------------------------------------------------------------------------------------
// K8  3-point Kabsch + checkers, one thread per hypothesis
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kabsch3_check_k(CloudView src, CloudView dst,
                                                       const uint32_t* __restrict__ corr_src,
                                                       const uint32_t* __restrict__ corr_dst,
                                                       const uint32_t* __restrict__ triples,
                                                       uint32_t h_count, double edge_thr, double dist_thr,
                                                       double* __restrict__ T12,
                                                       uint8_t* __restrict__ pass) {
    const uint32_t h = blockIdx.x * 64u + threadIdx.x;
    if (h >= h_count) return;
    double ps[9], pd[9];
    for (int j = 0; j < 3; ++j) {
        const uint32_t t = triples[3 * (size_t)h + j];
        const uint32_t si = corr_src[t], di = corr_dst[t];
        ps[3 * j] = src.x[si];
        ps[3 * j + 1] = src.y[si];
        ps[3 * j + 2] = src.z[si];
        pd[3 * j] = dst.x[di];
        pd[3 * j + 1] = dst.y[di];
        pd[3 * j + 2] = dst.z[di];
    }
    double T[16];
    umeyama3(ps, pd, T);
    const bool ok = reg_checkers(ps, pd, T, edge_thr, dist_thr);
    for (int k = 0; k < 12; ++k) T12[(size_t)h * kRegTStride + k] = T[k];
    pass[h] = ok ? 1 : 0;
}

void launch_kabsch3_check(const CloudView& src, const CloudView& dst, const uint32_t* corr_src,
                          const uint32_t* corr_dst, const uint32_t* triples, uint32_t h_count,
                          double edge_thr, double dist_thr, double* T12, uint8_t* pass, hipStream_t s) {
    if (!h_count) return;
    kabsch3_check_k<<<(h_count + 63) / 64, 64, 0, s>>>(src, dst, corr_src, corr_dst, triples, h_count,
                                                        edge_thr, dist_thr, T12, pass);
}

// gather the transformations of the surviving hypotheses into a dense array (+1 spare record)
__global__ void gather_T_k(const double* __restrict__ T12, const uint32_t* __restrict__ list,
                           uint32_t n, uint32_t n_pad, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pad * kRegTStride) return;
    const uint32_t s = i / kRegTStride, k = i % kRegTStride;
    out[i] = s < n ? T12[(size_t)list[s] * kRegTStride + k] : u2f(0x7FF8000000000000ull);
}
void launch_gather_T(const double* T12, const uint32_t* list, uint32_t n, uint32_t n_pad, double* out,
                     hipStream_t s) {
    if (!n_pad) return;
    gather_T_k<<<(n_pad * kRegTStride + 255) / 256, 256, 0, s>>>(T12, list, n, n_pad, out);
}

// ------------------------------------------------------------------------------------------------
// uniform grid over the target cloud (counting sort by cell)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cell_of(const GridDesc& g, double x, double y, double z, int lo_pad,
                                        int* ix, int* iy, int* iz) {
    const double fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
    // valid query cells are [lo_pad, n - 1 - lo_pad] (the table carries 2K+1 pad cells per side, so this admits
    // every query within K cells of the bounding box); NaN fails every comparison
    if (!(fx >= (double)lo_pad && fx < (double)(g.nx - lo_pad) && fy >= (double)lo_pad &&
          fy < (double)(g.ny - lo_pad) && fz >= (double)lo_pad && fz < (double)(g.nz - lo_pad)))
        return false;
    *ix = (int)fx;
    *iy = (int)fy;
    *iz = (int)fz;
    return true;
}

// cell_of + the position inside the cell as a fraction of its edge (the scaled coordinate's own fractional part: what
// (int) dropped), for the fp32 offsets of sorted_walk32
__device__ __forceinline__ bool cell_of_frac(const GridDesc& g, double x, double y, double z, int lo_pad, int* ix, int* iy,
                                             int* iz, double* frx, double* fry, double* frz) {
    const double fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
    if (!(fx >= (double)lo_pad && fx < (double)(g.nx - lo_pad) && fy >= (double)lo_pad &&
          fy < (double)(g.ny - lo_pad) && fz >= (double)lo_pad && fz < (double)(g.nz - lo_pad)))
        return false;
    *ix = (int)fx;
    *iy = (int)fy;
    *iz = (int)fz;
    *frx = fx - (double)*ix;   // (exact: fx >= 1 here, and both share their leading bits)
    *fry = fy - (double)*iy;
    *frz = fz - (double)*iz;
    return true;
}

// hist[cell] counts the points of a cell; rank[i] = how many points of its cell had arrived before point i
// (the value the counting add returns): the scatter then needs no second round of atomics.
__global__ void grid_count_k(CloudView dst, GridDesc g, uint32_t* __restrict__ cell_of_point,
                             uint32_t* __restrict__ hist, uint32_t* __restrict__ rank) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= dst.n) return;
    int ix, iy, iz;
    uint32_t cid = 0xFFFFFFFFu;  // points with NaN/inf coordinates are left out of the grid
    if (cell_of(g, dst.x[i], dst.y[i], dst.z[i], 0, &ix, &iy, &iz)) {
        if (g.morton_bits) {  // Z-order cell id: consecutive cells form compact cubes
            auto spread = [](uint32_t v) {  // 10 bits -> every third bit
                v = (v | (v << 16)) & 0x030000FFu;
                v = (v | (v << 8)) & 0x0300F00Fu;
                v = (v | (v << 4)) & 0x030C30C3u;
                v = (v | (v << 2)) & 0x09249249u;
                return v;
            };
            uint32_t X0 = (uint32_t)ix, X1 = (uint32_t)iy, X2 = (uint32_t)iz;
            if (g.morton_bits & 0x100u) {
                // Hilbert curve (Skilling's axes -> transpose): a continuous curve, so runs of consecutive
                // cells have tighter bounding boxes than Z-order runs.  Only XORs and masked exchanges:
                // a bijection on b-bit triples whatever the input, which is all the counting sort needs.
                const uint32_t b = g.morton_bits & 0xFFu;
                const uint32_t M = 1u << (b - 1);
                for (uint32_t Q = M; Q > 1; Q >>= 1) {
                    const uint32_t P = Q - 1;
                    if (X0 & Q) X0 ^= P;  // i = 0: invert
                    if (X1 & Q) {
                        X0 ^= P;
                    } else {
                        const uint32_t t = (X0 ^ X1) & P;
                        X0 ^= t;
                        X1 ^= t;
                    }
                    if (X2 & Q) {
                        X0 ^= P;
                    } else {
                        const uint32_t t = (X0 ^ X2) & P;
                        X0 ^= t;
                        X2 ^= t;
                    }
                }
                X1 ^= X0;  // Gray encode
                X2 ^= X1;
                uint32_t t = 0;
                for (uint32_t Q = M; Q > 1; Q >>= 1)
                    if (X2 & Q) t ^= Q - 1;
                X0 ^= t;
                X1 ^= t;
                X2 ^= t;
                cid = (spread(X0) << 2) | (spread(X1) << 1) | spread(X2);
            } else {
                cid = spread(X0) | (spread(X1) << 1) | (spread(X2) << 2);
            }
        } else {
            cid = ((uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix;
        }
        rank[i] = atomicAdd(&hist[cid], 1u);
    }
    cell_of_point[i] = cid;
}

// exclusive scan, tile of 2048 per workgroup; tile totals go to tile_sums
__global__ __launch_bounds__(256) void tile_scan_k(uint32_t* __restrict__ v, uint32_t n,
                                                    uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t e[8];
    uint32_t s = 0;
    for (int k = 0; k < 8; ++k) {
        e[k] = base + k < n ? v[base + k] : 0u;
        s += e[k];
    }
    // wave inclusive scan of s
    uint32_t incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t run = woff + incl - s;
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) v[base + k] = run;
        run += e[k];
    }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = woff + incl;
}
__global__ void add_tile_offsets_k(uint32_t* __restrict__ v, uint32_t n, const uint32_t* __restrict__ tile_off,
                                   const uint32_t* __restrict__ total) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) v[i] += tile_off[i / 2048u];
    if (i == n) v[n] = total[0];
}

__global__ void grid_scatter_k(CloudView dst, const uint32_t* __restrict__ cell_of_point,
                               const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ rank,
                               double* __restrict__ qx, double* __restrict__ qy, double* __restrict__ qz,
                               uint32_t* __restrict__ orig, double4* __restrict__ q4) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= dst.n) return;
    const uint32_t cid = cell_of_point[i];
    if (cid == 0xFFFFFFFFu) return;
    const uint32_t pos = cell_start[cid] + rank[i];
    const double x = dst.x[i], y = dst.y[i], z = dst.z[i];
    qx[pos] = x;
    qy[pos] = y;
    qz[pos] = z;
    if (orig) orig[pos] = i;
    if (q4) q4[pos] = make_double4(x, y, z, 0.0);
}

__global__ void fill_nan_k(double* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = u2f(0x7FF8000000000000ull);
}
// one thread stores `value` into a page-locked host word: queued behind a stream's work, it is the completion signal a host
// thread spins on (stream_wait_spin, m3d_fit.cpp) -- everything the kernels before it wrote to host memory is visible first
__global__ void signal_host_k(volatile uint32_t* __restrict__ word, uint32_t value) {
    __threadfence_system();
    *word = value;
}
void launch_signal_host(uint32_t* word, uint32_t value, hipStream_t st) { signal_host_k<<<1, 1, 0, st>>>(word, value); }

void launch_fill_nan(double* p, uint32_t n, hipStream_t s) {
    if (n) fill_nan_k<<<(n + 255) / 256, 256, 0, s>>>(p, n);
}

// Every GROUP of `group` consecutive cells (with Hilbert / Z-order cell ids and group = 8^k: an aligned cube of 2^k cells
// per axis) is padded to a multiple of pad_to slots; the padding is booked on the group's last cell.
__global__ void pad_counts_k(uint32_t* __restrict__ hist, uint32_t ncell, uint32_t pad_to, uint32_t group) {
    const uint32_t gidx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c0 = gidx * group;
    if (c0 >= ncell) return;
    const uint32_t c1 = min(ncell, c0 + group);
    uint32_t sum = 0;
    for (uint32_t c = c0; c < c1; ++c) sum += hist[c];
    hist[c1 - 1] += (sum + pad_to - 1) / pad_to * pad_to - sum;
}

// first half of launch_grid_build: cell of every point, rank inside its cell, cell_start = exclusive scan of the
// cell sizes (pad_to > 1: every group of pad_group consecutive cells is padded to a multiple of pad_to slots, so that
// every group's run starts on a pad_to boundary -- the slots in between stay what the caller pre-filled); total[0] =
// length of the sorted layout
void launch_grid_count_scan(const CloudView& dst, const GridDesc& g, uint32_t* cell_of_point, uint32_t* cell_start,
                            uint32_t* rank, uint32_t* tile_sums, uint32_t* total, hipStream_t s, uint32_t pad_to,
                            uint32_t pad_group) {
    const uint32_t ncell = g.nx * g.ny * g.nz;
    (void)hipMemsetAsync(cell_start, 0, sizeof(uint32_t) * ((size_t)ncell + 1), s);
    if (dst.n) grid_count_k<<<(dst.n + 255) / 256, 256, 0, s>>>(dst, g, cell_of_point, cell_start, rank);
    if (pad_to > 1) {
        const uint32_t ngroups = (ncell + pad_group - 1) / pad_group;
        pad_counts_k<<<(ngroups + 255) / 256, 256, 0, s>>>(cell_start, ncell, pad_to, pad_group);
    }
    const uint32_t nt = (ncell + 2047) / 2048;
    tile_scan_k<<<nt, 256, 0, s>>>(cell_start, ncell, tile_sums);
    launch_scan_blocks(tile_sums, nt, total, s);
    add_tile_offsets_k<<<(ncell + 1 + 255) / 256, 256, 0, s>>>(cell_start, ncell, tile_sums, total);
}
void launch_grid_scatter(const CloudView& dst, const uint32_t* cell_of_point, const uint32_t* cell_start,
                         const uint32_t* rank, double* qx, double* qy, double* qz, hipStream_t s, uint32_t* orig, double4* q4) {
    if (dst.n) grid_scatter_k<<<(dst.n + 255) / 256, 256, 0, s>>>(dst, cell_of_point, cell_start, rank, qx, qy, qz, orig, q4);
}
void launch_grid_build(const CloudView& dst, const GridDesc& g, uint32_t* cell_of_point,
                       uint32_t* cell_start /* ncell + 1 */, uint32_t* rank /* one per point: dst.n */,
                       uint32_t* tile_sums, uint32_t* total, double* qx, double* qy, double* qz,
                       hipStream_t s, uint32_t* orig, double4* q4) {
    launch_grid_count_scan(dst, g, cell_of_point, cell_start, rank, tile_sums, total, s, 0, 1);
    launch_grid_scatter(dst, cell_of_point, cell_start, rank, qx, qy, qz, s, orig, q4);
}

// Neighbour lists: for every interior cell the points of its 3x3x3 block, packed (x, y, z, 0) and
// contiguous, in the fixed order (dz, dy) rows ascending, cell_start order inside a row.
__global__ void nl_count_k(GridDesc g, const uint32_t* __restrict__ cell_start, uint32_t ncell,
                           uint32_t* __restrict__ nl_start) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t ix = c % g.nx, iy = (c / g.nx) % g.ny, iz = c / (g.nx * g.ny);
    uint32_t cnt = 0;
    if (ix >= 1 && ix + 1 < g.nx && iy >= 1 && iy + 1 < g.ny && iz >= 1 && iz + 1 < g.nz)
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const uint32_t row = ((iz + dz) * g.ny + (iy + dy)) * g.nx + ix;
                cnt += cell_start[row + 2] - cell_start[row - 1];
            }
    nl_start[c] = cnt;
}
__global__ void nl_fill_k(GridDesc g, const uint32_t* __restrict__ cell_start, uint32_t ncell,
                          const uint32_t* __restrict__ nl_start, const double* __restrict__ qx,
                          const double* __restrict__ qy, const double* __restrict__ qz,
                          double4* __restrict__ nl_pts, const uint32_t* __restrict__ orig) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    uint32_t pos = nl_start[c];
    if (nl_start[c + 1] == pos) return;
    const uint32_t ix = c % g.nx, iy = (c / g.nx) % g.ny, iz = c / (g.nx * g.ny);
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy) {
            const uint32_t row = ((iz + dz) * g.ny + (iy + dy)) * g.nx + ix;
            const uint32_t b = cell_start[row - 1], e = cell_start[row + 2];
            for (uint32_t k = b; k < e; ++k)
                nl_pts[pos++] = make_double4(qx[k], qy[k], qz[k], orig ? (double)orig[k] : 0.0);
        }
}
// points of every cell ordered by x (insertion sort: a cell holds a handful of points)
__global__ void sort_cells_by_x_k(const uint32_t* __restrict__ cell_start, uint32_t ncell, double* __restrict__ qx,
                                  double* __restrict__ qy, double* __restrict__ qz) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t b = cell_start[c], e = cell_start[c + 1];
    for (uint32_t i = b + 1; i < e; ++i) {
        const double x = qx[i], y = qy[i], z = qz[i];
        uint32_t j = i;
        while (j > b && qx[j - 1] > x) {
            qx[j] = qx[j - 1];
            qy[j] = qy[j - 1];
            qz[j] = qz[j - 1];
            --j;
        }
        qx[j] = x;
        qy[j] = y;
        qz[j] = z;
    }
}
// the three-column layout of GridDesc::nl_sorted: per column a 9-way merge of x-sorted cells
__global__ void nl_fill_sorted_k(GridDesc g, const uint32_t* __restrict__ cell_start, uint32_t ncell,
                                 const uint32_t* __restrict__ nl_start, const double* __restrict__ qx,
                                 const double* __restrict__ qy, const double* __restrict__ qz,
                                 double4* __restrict__ nl_pts, uint32_t* __restrict__ nl_hdr,
                                 uint2* __restrict__ nl32, uint4* __restrict__ nl_rec, uint32_t* __restrict__ overflow,
                                 const uint32_t* __restrict__ nl32_start) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t pos0 = nl_start[c];
    if (nl_start[c + 1] == pos0) {
        if (nl_hdr) nl_hdr[c] = 0u;
        if (nl_rec) nl_rec[c] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const uint32_t ix = c % g.nx, iy = (c / g.nx) % g.ny, iz = c / (g.nx * g.ny);
    uint32_t pos = pos0, n_col[3] = {0, 0, 0};
    const int col_dx[3] = {0, -1, 1};
    // (the expressions sorted_walk32 uses for the cell's corner)
    const double hc = 1.0 / g.inv_h;
    const double Ox = g.ox + (double)(int)ix * hc, Oy = g.oy + (double)(int)iy * hc, Oz = g.oz + (double)(int)iz * hc;
    for (int col = 0; col < 3; ++col) {
        uint32_t p[9], e[9];
        for (int k = 0; k < 9; ++k) {
            const int dz = k / 3 - 1, dy = k % 3 - 1;
            const uint32_t cell = ((iz + dz) * g.ny + (iy + dy)) * g.nx + (ix + col_dx[col]);
            p[k] = cell_start[cell];
            e[k] = cell_start[cell + 1];
        }
        const bool desc = col == 1;
        for (;;) {
            int best = -1;
            double bx = 0.0;
            for (int k = 0; k < 9; ++k) {
                if (p[k] >= e[k]) continue;
                const double x = desc ? qx[e[k] - 1] : qx[p[k]];   // descending: take from the cells' upper ends
                if (best < 0 || (desc ? x > bx : x < bx)) {
                    best = k;
                    bx = x;
                }
            }
            if (best < 0) break;
            const uint32_t src = desc ? --e[best] : p[best]++;
            nl_pts[pos++] = make_double4(qx[src], qy[src], qz[src], 0.0);
            n_col[col]++;
        }
    }
    nl_pts[pos0].w = (double)(n_col[0] + 65536u * n_col[1]);   // (a list holds far fewer than 65536 points)
    if (nl_hdr) nl_hdr[c] = n_col[0] + 65536u * n_col[1];
    if (!nl32) return;
    // the screen's copy (sorted_walk32): the three columns merged in ascending fp32 x (left column reversed, own, right --
    // a merge because the columns may overlap by a rounding at the cell faces), every entry 8 BYTES: x, y, z as 16-bit
    // fixed point over the block [-h, 2h) of the list's own cell -- q = rint(v S + 21845), S = 21845 / h, monotone in v:
    // the order is kept -- and the 16-bit position of the fp64 entry in the list; pads at both ends that lie farther from
    // every query of the cell than the search radius (corner of the block), position 0
    const uint32_t n_mid = n_col[0], n_lr = n_col[0] + n_col[1], n_tot = n_lr + n_col[2];
    if (n_tot > 65535u) {
        atomicOr(overflow, 1u);   // (the 16-bit offsets of nl_rec and of the entries: the caller drops the screen)
        return;
    }
    const uint32_t q0 = nl32_start[c] + (uint32_t)kWalkPad;   // first real entry
    for (int k = 0; k < kWalkPad; ++k) {
        nl32[q0 - 1u - (uint32_t)k] = make_uint2(0u, 0u);
        nl32[q0 + n_tot + (uint32_t)k] = make_uint2(0xFFFFFFFFu, 0x0000FFFFu);
    }
    uint32_t il = n_lr, im = 0u, ir = n_lr;   // left: entries il - 1 down to n_mid
    auto off32 = [&](uint32_t rel) {
        const double4 q = nl_pts[pos0 + rel];
        return make_float4((float)(q.x - Ox), (float)(q.y - Oy), (float)(q.z - Oz), __uint_as_float(rel));
    };
    const float S32 = (float)(21845.0 * g.inv_h);
    auto quant = [&](float v) {
        const float f = __builtin_rintf(__builtin_fmaf(v, S32, 21845.0f));
        return (uint32_t)__builtin_fminf(65535.0f, __builtin_fmaxf(0.0f, f));
    };
    float thr[5];
    for (int k = 0; k < 5; ++k) thr[k] = (float)((double)k * hc * 0.25);
    uint32_t offs[5] = {0u, 0u, 0u, 0u, 0u};
    float4 fl, fm, fr;
    if (il > n_mid) fl = off32(il - 1u);
    if (im < n_mid) fm = off32(im);
    if (ir < n_tot) fr = off32(ir);
    for (uint32_t o = 0; o < n_tot; ++o) {
        const bool hl = il > n_mid, hm = im < n_mid, hr = ir < n_tot;
        int pick = hl ? 0 : (hm ? 1 : 2);
        float bx = hl ? fl.x : (hm ? fm.x : fr.x);
        if (pick == 0 && hm && fm.x < bx) {
            pick = 1;
            bx = fm.x;
        }
        if (pick != 2 && hr && fr.x < bx) pick = 2;
        float4 v;
        if (pick == 0) {
            v = fl;
            if (--il > n_mid) fl = off32(il - 1u);
        } else if (pick == 1) {
            v = fm;
            if (++im < n_mid) fm = off32(im);
        } else {
            v = fr;
            if (++ir < n_tot) fr = off32(ir);
        }
        nl32[q0 + o] = make_uint2(quant(v.x) | (quant(v.y) << 16), quant(v.z) | (__float_as_uint(v.w) << 16));
        for (int k = 0; k < 5; ++k) offs[k] += v.x < thr[k] ? 1u : 0u;
    }
    nl_rec[c] = make_uint4(q0, offs[0] | (offs[1] << 16), offs[2] | (offs[3] << 16), offs[4] | (n_tot << 16));
}

// nl32_start[0..ncell]: exclusive scan of (entries + 2 kWalkPad) over the non-empty lists; total[0] = the array's length
__global__ void nl32_count_k(const uint32_t* __restrict__ nl_start, uint32_t ncell, uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t cnt = nl_start[c + 1] - nl_start[c];
    out[c] = cnt ? cnt + 2u * (uint32_t)kWalkPad : 0u;
}
void launch_nl32_offsets(const uint32_t* nl_start, uint32_t ncell, uint32_t* nl32_start, uint32_t* tile_sums,
                         uint32_t* total, hipStream_t s) {
    nl32_count_k<<<(ncell + 255) / 256, 256, 0, s>>>(nl_start, ncell, nl32_start);
    const uint32_t nt = (ncell + 2047) / 2048;
    tile_scan_k<<<nt, 256, 0, s>>>(nl32_start, ncell, tile_sums);
    launch_scan_blocks(tile_sums, nt, total, s);
    add_tile_offsets_k<<<(ncell + 1 + 255) / 256, 256, 0, s>>>(nl32_start, ncell, tile_sums, total);
}

// step 1: counts + exclusive scan into nl_start[0..ncell]; the caller reads nl_start[ncell] (= entries),
// allocates nl_pts and calls step 2.
void launch_nl_count(const GridDesc& g, const uint32_t* cell_start, uint32_t* nl_start, uint32_t* tile_sums,
                     uint32_t* total, hipStream_t s) {
    const uint32_t ncell = g.nx * g.ny * g.nz;
    nl_count_k<<<(ncell + 255) / 256, 256, 0, s>>>(g, cell_start, ncell, nl_start);
    const uint32_t nt = (ncell + 2047) / 2048;
    tile_scan_k<<<nt, 256, 0, s>>>(nl_start, ncell, tile_sums);
    launch_scan_blocks(tile_sums, nt, total, s);
    add_tile_offsets_k<<<(ncell + 1 + 255) / 256, 256, 0, s>>>(nl_start, ncell, tile_sums, total);
}
void launch_nl_fill(const GridDesc& g, const uint32_t* cell_start, const uint32_t* nl_start, double* qx,
                    double* qy, double* qz, double4* nl_pts, hipStream_t s, const uint32_t* orig, bool sorted,
                    uint32_t* nl_hdr, uint2* nl32, uint4* nl_rec, uint32_t* overflow, const uint32_t* nl32_start) {
    const uint32_t ncell = g.nx * g.ny * g.nz;
    if (sorted && !orig) {
        sort_cells_by_x_k<<<(ncell + 255) / 256, 256, 0, s>>>(cell_start, ncell, qx, qy, qz);
        nl_fill_sorted_k<<<(ncell + 255) / 256, 256, 0, s>>>(g, cell_start, ncell, nl_start, qx, qy, qz, nl_pts, nl_hdr, nl32, nl_rec, overflow, nl32_start);
        return;
    }
    nl_fill_k<<<(ncell + 255) / 256, 256, 0, s>>>(g, cell_start, ncell, nl_start, qx, qy, qz, nl_pts, orig);
}

// ------------------------------------------------------------------------------------------------
// K9  validation: nearest target point of every transformed source point, per surviving hypothesis
// ------------------------------------------------------------------------------------------------
// Phase 2 of the search: the (2K+1)^3 block, which covers the radius -- but only the cells that can hold
// a point closer than lim = min(best so far, r^2): a point at squared distance >= best cannot lower the
// minimum, and one at >= r^2 is no correspondence whatever its exact distance (the callers only use
// values < r^2).  Per axis the gap between the query and a cell offset d is a lower bound of the
// coordinate difference to every point of that cell (minus a slack for the rounding of the cell
// assignment, see axis_gap).
__device__ __forceinline__ double axis_gap(int d, double f, double h) {
    // query at fraction f in [0,1) of its cell; cells at offset d span [d, d+1) in the same units.
    // Slack 1e-6 cell edges: cell_of evaluates (x - o) * inv_h with two roundings, relative error < 3 * 2^-53
    // of a value below 2^27, i.e. below 5e-8 edges, for the query and for the stored point alike.
    const double g = d > 0 ? (double)d - f : (d < 0 ? f - (double)(d + 1) : 0.0);
    const double gs = g - 1e-6;
    return gs > 0.0 ? gs * h : 0.0;
}
__device__ __forceinline__ double nearest_phase2(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                                 const double* __restrict__ qx, const double* __restrict__ qy,
                                                 const double* __restrict__ qz, int ix, int iy, int iz, double px,
                                                 double py, double pz, double best) {
    const int K = g.K;
    const double h = 1.0 / g.inv_h;
    const double fx = (px - g.ox) * g.inv_h - (double)ix, fy = (py - g.oy) * g.inv_h - (double)iy,
                 fz = (pz - g.oz) * g.inv_h - (double)iz;
    for (int dz = -K; dz <= K; ++dz) {
        const double gz = axis_gap(dz, fz, h);
        const double gz2 = gz * gz;
        if (!(gz2 < (best < g.r2 ? best : g.r2))) continue;
        for (int dy = -K; dy <= K; ++dy) {
            const double gy = axis_gap(dy, fy, h);
            const double lim = best < g.r2 ? best : g.r2;
            const double rem = lim - (gz2 + gy * gy);
            if (!(rem > 0.0)) continue;
            // x cells that can still matter: the gap grows with |dx|, so they form one interval
            int lo = 0, hi = 0;
            while (lo > -K && axis_gap(lo - 1, fx, h) * axis_gap(lo - 1, fx, h) < rem) --lo;
            while (hi < K && axis_gap(hi + 1, fx, h) * axis_gap(hi + 1, fx, h) < rem) ++hi;
            const bool inner = dz >= -1 && dz <= 1 && dy >= -1 && dy <= 1;   // x in [-1, 1] was phase 1
            const uint32_t row = ((uint32_t)(iz + dz) * g.ny + (uint32_t)(iy + dy)) * g.nx + (uint32_t)ix;
            for (int part = 0; part < 2; ++part) {
                int a, bnd;
                if (inner) {
                    a = part == 0 ? lo : 2;
                    bnd = part == 0 ? -2 : hi;
                } else {
                    if (part == 1) break;
                    a = lo;
                    bnd = hi;
                }
                if (a > bnd) continue;
                const uint32_t b = cell_start[row + a], e = cell_start[row + bnd + 1];
                for (uint32_t c = b; c < e; ++c) {
                    const double ddx = px - qx[c], ddy = py - qy[c], ddz = pz - qz[c];
                    const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (d2 < best) best = d2;
                }
            }
        }
    }
    return best;
}

constexpr int kNlBatch = 2;   // neighbour-list candidates per trip
constexpr int kNlAhead = 4;   // own-column entries per trip of the sorted walk (one batch evaluated, one in flight)

// The walk over a three-column list (GridDesc::nl_sorted) in fp64: the own column in full, in batches of kNlAhead
// entries with the next batch in flight; then the left column from the cell outwards (descending x) and the right one
// (ascending x) only while the x-distance ALONE is below the best squared distance so far -- every entry behind the cut
// is farther away than `best`, so the minimum is the same as over the whole 3x3x3 block.  A batch may run past the end
// of its column into the next one: every entry of the list is a point of the block, evaluating more of them than the
// cut requires leaves the minimum what it is.  (nl_hdr: all three column bounds after ONE round trip.)
__device__ __forceinline__ double sorted_walk64(const GridDesc& g, uint32_t cell, double px, double py, double pz) {
    double best = INFINITY;
    const uint32_t b = g.nl_start[cell], e = g.nl_start[cell + 1], hdr = g.nl_hdr[cell];
    if (b >= e) return best;
    const uint32_t last = e - 1u;
    const uint32_t m_end = b + (hdr & 0xFFFFu), l_end = m_end + (hdr >> 16);
    struct P3 {
        double x, y, z;
    };
    auto ld = [&](uint32_t i) {
        const double4* __restrict__ p = g.nl_pts + min(i, last);
        return P3{p->x, p->y, p->z};
    };
    auto eval = [&](const P3& q) {
        const double ddx = px - q.x, ddy = py - q.y, ddz = pz - q.z;
        const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
        if (d2 < best) best = d2;
    };
    P3 cur[kNlAhead], nxt[kNlAhead];
#pragma unroll
    for (int k = 0; k < kNlAhead; ++k) cur[k] = ld(b + (uint32_t)k);
    P3 lf = ld(m_end), rf = ld(l_end);
    for (uint32_t c = b; c < m_end; c += kNlAhead) {
#pragma unroll
        for (int k = 0; k < kNlAhead; ++k) nxt[k] = ld(c + (uint32_t)(kNlAhead + k));
#pragma unroll
        for (int k = 0; k < kNlAhead; ++k) eval(cur[k]);
#pragma unroll
        for (int k = 0; k < kNlAhead; ++k) cur[k] = nxt[k];
    }
    for (uint32_t c = m_end; c < l_end; ++c) {   // left column, descending x
        const P3 nx = ld(c + 1u);
        const double ddx = px - lf.x;
        if (!(ddx * ddx < best)) break;
        eval(lf);
        lf = nx;
    }
    for (uint32_t c = l_end; c < e; ++c) {       // right column, ascending x
        const P3 nx = ld(c + 1u);
        const double ddx = px - rf.x;
        if (!(ddx * ddx < best)) break;
        eval(rf);
        rf = nx;
    }
    return best;
}
// The SCREEN of the neighbour search (GridDesc::nl32 / nl_rec).  Counters first: with fp64 entries the kernel was
// bound by the L1's tag look-ups (TCP busy 70 % + 14 % tag-conflict stalls; a gather of one list entry costs a look-up
// per distinct cell among the lanes -- ~27 per instruction -- whatever its size), and behind that by the number of
// DEPENDENT round trips of a walk.  So: visit fewer entries, fetch SEVERAL with one load, several per round trip.
//   * nl32 holds every list a second time as 8-BYTE entries: x, y, z as 16-bit fixed point over the 3-cell block
//     [-h, 2h) of the list's own cell -- q = rint(v S + 21845), S = 21845 / h, one unit = 3 h / 65535 -- and the 16-bit
//     position of the fp64 entry in the list (lists of more than 65535 entries drop the screen); all 27 cells MERGED in
//     ascending x (quantisation is monotone: the order by fp32 x is kept), kWalkPad pad entries at either end (the
//     block's corners: farther from every query of the cell than the search radius).  Round 2 had 16-byte entries (fp32
//     offsets + a 32-bit index): two entries per 16-byte load halve the gather instructions -- TCP_TOTAL_CACHE_ACCESSES
//     15.9 G -> 10.3 G on 20 000 iterations of C4, the forced 100 000-iteration call 159 -> 142 ms.
//   * nl_rec[cell] = (start, five 16-bit offsets, entries): offset k = the first entry with x >= k h / 4.  A query
//     starts at the offset nearest to it and walks right (ascending x) and left (descending x) AT THE SAME TIME, two
//     entries (ONE load) per side and trip, the next batch of each side in flight; a side goes on only while the
//     x-distance ALONE of the batch's last entry can still beat the best candidate, and while it has entries left.
//   * Rounding, in units: an entry's q is within 0.55 of (v + h) S (rint: 0.5; the fp32 offset v and the product: < 0.05),
//     the query's (u + h) S within 0.01, their difference rounds by < 0.005: every coordinate difference is within
//     eps = 0.57 of the truth.  s = fma(dz, dz, fma(dy, dy, dx dx)) then differs from the true squared distance (in
//     units^2: the fp64 walk's d2 times S^2) by at most 2 eps (|dx| + |dy| + |dz|) + 3 eps^2 + 3.1 u s
//     <= 1.98 sqrt(s) + 1 + 2^-18 s:  E(s) := 2.1 sqrt(s) + 2 + 2^-18 s (walk_err_bound; tests/cpp/test_nn_screen.cpp replays
//     the walk on the host: worst |s - d2| / E(s) = 0.79 over millions of entries, and a decided query's winner is the
//     fp64 minimum to the last bit).
//   * The walk keeps the smallest s (m1, at entry i1) and the second smallest (m2).  If m2 > m1 + E(m1) + E(m2), entry
//     i1 is strictly nearer in fp64 than every other visited entry, and the result is ITS fp64 distance, evaluated with
//     the fp64 walk's expression on the fp64 entry.  Otherwise (exact ties, duplicates, near ties: 4.5 queries in 10^4 on
//     C4) the query takes the fp64 walk.
//   * A side stops after a batch whose last entry lies on that side of the query with (|dx| - 0.6)^2 >= m1 + E(m1), m1
//     taken BEFORE the batch (larger: later): the list is sorted by the very q the test uses, so the true x-distance of
//     everything behind that entry is at least |dx| - eps, i.e. at least the true distance of entry i1.  For this CUT
//     E(m1) is replaced by the square-root-free m1 (2^-9 + 2^-18) + 567 >= E(m1) (AM-GM): a cut 0.1 % later.  Where the
//     walks start affects their length only: right covers [start, n), left [0, start), every entry is visited at most
//     once; a pad that shares the last batch lies beyond the search radius and can only win when nothing real is within
//     it (its position is 0: the result is then a real entry's distance, larger than the radius like the true minimum).
__device__ __forceinline__ float walk_err_bound(float s) {   // E(s), units^2 (tests/cpp/test_nn_screen.cpp checks it)
    return __builtin_fmaf(s, 0x1p-18f, __builtin_fmaf(2.1f, __builtin_sqrtf(s), 2.0f));
}
__device__ __forceinline__ double sorted_walk32(const GridDesc& g, uint32_t cell, double frx, double fry, double frz,
                                                double px, double py, double pz) {
    uint4 rec = g.nl_rec[cell];
    const uint32_t pos0 = g.nl_start[cell];   // (the list's fp64 entries: independent of rec, the same round trip)
    // (all four words are needed at once: without this the compiler fetches w first, tests it, and fetches the rest in a
    // second, dependent round trip)
    asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z), "+v"(rec.w));
    const int n_tot = (int)(rec.w >> 16);
    if (n_tot == 0) return INFINITY;   // empty list
    const double h = 1.0 / g.inv_h;
    // the query in the entries' units: (offset from the cell's min corner + h) * S, S = 21845 / h
    const float S32 = (float)(21845.0 * g.inv_h);
    const float ux = __builtin_fmaf((float)(frx * h), S32, 21845.0f), uy = __builtin_fmaf((float)(fry * h), S32, 21845.0f),
                uz = __builtin_fmaf((float)(frz * h), S32, 21845.0f);
    const int ke = min(4, max(0, (int)__builtin_fmaf((float)frx, 4.0f, 0.5f)));   // nearest quarter boundary
    const uint32_t offw = ke < 2 ? rec.y : (ke < 4 ? rec.z : rec.w);
    const int start = (int)((ke & 1) ? offw >> 16 : offw & 0xFFFFu);
    const char* __restrict__ base = reinterpret_cast<const char*>(g.nl32 + rec.x + (uint32_t)start);
    typedef uint4 pair_t __attribute__((aligned(8)));   // two entries, ONE load (8-byte aligned: an odd start)
    auto ld2 = [&](int i) { return *reinterpret_cast<const pair_t*>(base + (ptrdiff_t)i * 8); };   // entries i, i + 1
    float m1 = __builtin_inff(), m2 = __builtin_inff();
    uint32_t w1 = 0u;   // the winner's second word (z | position << 16): the position is cut out once, after the walk
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t U_x = {ux, ux}, U_y = {uy, uy}, U_z = {uz, uz};
    // two entries (A first) with packed fp32 arithmetic -- the same IEEE operations per entry as one at a time; returns
    // the dx of entry B
    auto visit2 = [&](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) -> float {
        const f32x2_t X = {(float)(a0 & 0xFFFFu), (float)(b0 & 0xFFFFu)}, Y = {(float)(a0 >> 16), (float)(b0 >> 16)},
                      Z = {(float)(a1 & 0xFFFFu), (float)(b1 & 0xFFFFu)};
        const f32x2_t dx = X - U_x, dy = Y - U_y, dz = Z - U_z;
        const f32x2_t s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        // (the new minimum as a select on the comparison the winner's word needs anyway: fminf would first canonicalise m1)
        const bool ca = s.x < m1;
        w1 = ca ? a1 : w1;
        m2 = __builtin_amdgcn_fmed3f(m1, m2, s.x);   // (m1 <= m2: the median is the new runner-up)
        m1 = ca ? s.x : m1;
        const bool cb = s.y < m1;
        w1 = cb ? b1 : w1;
        m2 = __builtin_amdgcn_fmed3f(m1, m2, s.y);
        m1 = cb ? s.y : m1;
        return dx.y;
    };
    static_assert(kWalkB == 2, "one 16-byte load per side and trip holds the batch");
    // Round 5: of the loop's 33 VALU instructions per side and trip (ISA of round 4) 3 formed an address from a signed index and 2
    // cut the position out of every entry.  Now: running pointers (one 64-bit add per side and trip) and the winner's word kept
    // whole.  Same loads, same values, same order.  (Two trips per iteration on alternating registers, to save the two moves of
    // the prefetched batch as well, made the compiler shuffle seven registers per side instead: not kept.)
    const char* pr = base;        // the batch `right` evaluates next: entries cr, cr + 1
    const char* pl = base - 16;   // ... and `left`: entries -2 - cl, -1 - cl
    auto ldp = [](const char* q) { return *reinterpret_cast<const pair_t*>(q); };
    uint4 rb = ldp(pr), lb = ldp(pl);
    // right has entries left while pr < pr_end, left while pl > pl_end (one 64-bit comparison instead of a counter and its test)
    const char* const pr_end = base + (ptrdiff_t)(n_tot - start) * 8;
    const char* const pl_end = base - 16 - (ptrdiff_t)start * 8;
    bool ar = start < n_tot, al = start > 0;
    while (ar || al) {
        // everything behind a batch's last entry is at least (|dx| - 0.6)^2 away in truth, the winner so far at most
        // m1 + E(m1) (m1 taken before the batch: larger, later).  For the CUT a bound without the square root serves:
        // 2.1 sqrt(s) <= s / 512 + 565 (AM-GM), so E(s) <= s (2^-9 + 2^-18) + 567 -- a cut that comes 0.1 % later
        const float thr = __builtin_fmaf(m1, 1.0f + 0x1p-9f + 0x1p-18f, 567.0f);
        if (ar) {
            pr += 16;
            const uint4 nx = ldp(pr);
            const float t = visit2(rb.x, rb.y, rb.z, rb.w) - 0.6f;
            ar = !(t > 0.0f && !(t * t < thr)) && pr < pr_end;
            rb = nx;
        }
        if (al) {
            pl -= 16;
            const uint4 nx = ldp(pl);
            const float t = -visit2(lb.z, lb.w, lb.x, lb.y) - 0.6f;   // the nearer entry of the batch first
            al = !(t > 0.0f && !(t * t < thr)) && pl > pl_end;
            lb = nx;
        }
    }
    const uint32_t i1 = w1 >> 16;
#ifdef M3D_REG_TRIP_STATS
    if (g.nl32_fallbacks) {   // (diagnostic build: tools only)
        const uint32_t trips = (uint32_t)max((int)((pr - base) / 16), (int)((base - 16 - pl) / 16));
        uint32_t wmax = 0;
        for (uint32_t k = 1; k < 64u; ++k)
            if (__ballot(trips >= k) != 0ull) wmax = k;   // (over the lanes that are here)
        atomicAdd(g.nl32_fallbacks + 8, 1ull);
        atomicAdd(g.nl32_fallbacks + 9, (unsigned long long)trips);
        atomicAdd(g.nl32_fallbacks + 16 + min(trips, 23u), 1ull);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) {   // first active lane
            atomicAdd(g.nl32_fallbacks + 10, 1ull);
            atomicAdd(g.nl32_fallbacks + 11, (unsigned long long)wmax);
        }
    }
#endif
    const float bound = walk_err_bound(m1) + walk_err_bound(m2);   // E(m1) + E(m2)
    const bool decided = m2 == __builtin_inff() ? m1 < __builtin_inff() /* one candidate */ : m2 > m1 + bound * 1.0001f;
    if (!decided) {
        if (g.nl32_fallbacks) atomicAdd(g.nl32_fallbacks, 1ull);
        return sorted_walk64(g, cell, px, py, pz);
    }
    const double4* __restrict__ w = g.nl_pts + pos0 + i1;
    const double ddx = px - w->x, ddy = py - w->y, ddz = pz - w->z;
    return (ddx * ddx + ddy * ddy) + ddz * ddz;
}

// Exact nearest squared distance within the search radius (KDTreeFlann::SearchHybrid(p, r, 1)).
// The grid cell is r/K.  Phase 1 scans the 3x3x3 block around the query's cell (9 contiguous x-rows):
// every point outside that block is at least one cell edge away, so a hit closer than 0.999 h is the
// true nearest neighbour -- the common case for an aligned pose.  Phase 2 (nothing that close) scans
// the (2K+1)^3 block, which covers the whole radius.  `min` is order-free, so the value equals the
// kd-tree's.  Returns +inf when nothing lies in the scanned block.
// SCREEN: the caller has checked that g.nl32 is there (launch_reg_validate picks the instantiation; the others pass false
// and walk in fp64).
template <bool SCREEN = false>
__device__ __forceinline__ double nearest_d2(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                             const double* __restrict__ qx, const double* __restrict__ qy,
                                             const double* __restrict__ qz, double px, double py, double pz) {
    int ix, iy, iz;
    double best = INFINITY;
    if (SCREEN) {
        double frx, fry, frz;
        if (!cell_of_frac(g, px, py, pz, g.K, &ix, &iy, &iz, &frx, &fry, &frz)) {
#ifdef M3D_REG_TRIP_STATS
            if (g.nl32_fallbacks) atomicAdd(g.nl32_fallbacks + 41, 1ull);
#endif
            return best;
        }
        const uint32_t cell = ((uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix;
        best = sorted_walk32(g, cell, frx, fry, frz, px, py, pz);
    } else if (!cell_of(g, px, py, pz, g.K, &ix, &iy, &iz)) {
        return best;
    } else if (g.nl_start && g.nl_sorted) {
        const uint32_t cell = ((uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix;
        best = sorted_walk64(g, cell, px, py, pz);
    } else if (g.nl_start) {
        // the 3x3x3 block of this cell as ONE contiguous list (nl_fill_k): two dependent loads instead of
        // nine row ranges + nine gathers; 4 candidates per trip, the tail repeats the last one (min is idempotent)
        const uint32_t cell = ((uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix;
        const uint32_t b = g.nl_start[cell], e = g.nl_start[cell + 1];
        if (b < e) {
            // software pipeline: the loads of trip t+1 are in flight while trip t is evaluated
            double4 cur[kNlBatch];
#pragma unroll
            for (int k = 0; k < kNlBatch; ++k) cur[k] = g.nl_pts[min(b + (uint32_t)k, e - 1)];
            for (uint32_t c = b; c < e; c += kNlBatch) {
                double4 nxt[kNlBatch];
#pragma unroll
                for (int k = 0; k < kNlBatch; ++k) nxt[k] = g.nl_pts[min(c + (uint32_t)(kNlBatch + k), e - 1)];
#pragma unroll
                for (int k = 0; k < kNlBatch; ++k) {
                    const double ddx = px - cur[k].x, ddy = py - cur[k].y, ddz = pz - cur[k].z;
                    const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (d2 < best) best = d2;
                }
#pragma unroll
                for (int k = 0; k < kNlBatch; ++k) cur[k] = nxt[k];
            }
        }
    } else {
        // (no lists: a call that validates a handful of hypotheses -- the reference's default confidence.  Round 6: the nine rows'
        //  bounds requested together, the candidates as 32-byte records, four in flight per trip: on C4's default-confidence call
        //  reg_min_d2_k 152 -> 87 us, reg_validate_k<false> 350 -> 315 us -- it is the record format that paid, the loads'
        //  order did not: with the source in its original order a wave's 64 queries touch 64 different cells.)
        uint32_t rb[9], re[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const uint32_t row = ((uint32_t)(iz + k / 3 - 1) * g.ny + (uint32_t)(iy + k % 3 - 1)) * g.nx + (uint32_t)ix;
            rb[k] = cell_start[row - 1];
            re[k] = cell_start[row + 2];
        }
        if (g.q4) {   // (x, y, z) of a candidate in ONE 32-byte gather, four candidates in flight per trip (the tail repeats the
                      // row's last one: min is idempotent) -- one candidate per trip was a chain of ~45 dependent round trips per query
#pragma unroll
            for (int k = 0; k < 9; ++k)
                for (uint32_t c = rb[k]; c < re[k]; c += 4u) {
                    double4 q[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) q[j] = g.q4[min(c + j, re[k] - 1u)];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const double ddx = px - q[j].x, ddy = py - q[j].y, ddz = pz - q[j].z;
                        const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                        if (d2 < best) best = d2;
                    }
                }
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                for (uint32_t c = rb[k]; c < re[k]; ++c) {
                    const double ddx = px - qx[c], ddy = py - qy[c], ddz = pz - qz[c];
                    const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (d2 < best) best = d2;
                }
        }
    }
#ifdef M3D_REG_TRIP_STATS
    if (SCREEN && g.nl32_fallbacks) {   // (diagnostic build: where do the queries end?  profiles/r05_reg_query_fates.txt)
        const bool p2 = !(best < g.h2_in || g.K == 1);
        const double fin = p2 ? nearest_phase2(g, cell_start, qx, qy, qz, ix, iy, iz, px, py, pz, best) : best;
        atomicAdd(g.nl32_fallbacks + 40, 1ull);
        if (best == INFINITY) atomicAdd(g.nl32_fallbacks + 42, 1ull);
        if (p2) atomicAdd(g.nl32_fallbacks + 43, 1ull);
        if (p2 && fin < g.r2) atomicAdd(g.nl32_fallbacks + 44, 1ull);
        if (fin < g.r2) atomicAdd(g.nl32_fallbacks + 45, 1ull);
        const unsigned long long here = __ballot(true), m_any = __ballot(fin < g.r2), p_any = __ballot(p2), w_any = __ballot(best < INFINITY);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) {
            atomicAdd(g.nl32_fallbacks + 46, 1ull);
            if (m_any == 0ull) atomicAdd(g.nl32_fallbacks + 47, 1ull);
            if (p_any != 0ull) atomicAdd(g.nl32_fallbacks + 48, 1ull);
            if (w_any == 0ull) atomicAdd(g.nl32_fallbacks + 49, 1ull);
            atomicAdd(g.nl32_fallbacks + 50, (unsigned long long)__popcll(here));
        }
        return fin;
    }
#endif
    if (best < g.h2_in || g.K == 1) return best;
    // (round 6) more than K empty rings around the query's cell: every target point is at least K h = 1.001 r away -- what the
    // scan below would establish cell by cell.  best is then the 3x3x3 block's (empty: +inf), and the callers only use values < r^2.
    if (g.ring && g.ring[((size_t)(uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix] > (uint32_t)g.K) return best;
    return nearest_phase2(g, cell_start, qx, qy, qz, ix, iy, iz, px, py, pz, best);
}

// the tile_local-th tile whose index mod 32 is in res_mask (tiles in ascending order)
__device__ __forceinline__ uint32_t phase_tile(uint32_t tile_local, uint32_t res_mask) {
    const uint32_t k = (uint32_t)__popc(res_mask);
    uint32_t m = res_mask;
    for (uint32_t j = tile_local % k; j > 0; --j) m &= m - 1u;   // drop the j lowest set bits
    return (tile_local / k) * 32u + (uint32_t)(__ffs(m) - 1);
}

// Same decomposition as score_k: source points stay in VGPRs (kRegP rows of 64 per wave), the
// transformations of the surviving hypotheses stream through SGPRs.  Per hypothesis: the number of
// source points whose nearest target point is closer than the threshold (ballot + s_bcnt1) and the
// order-free sum of those squared distances (the rmse numerator; decides fitness ties, see the
// driver).  partial_cnt / partial_sum: [tile][s_pad].
template <bool SCREEN>
__global__ __launch_bounds__(256) void reg_validate_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                       const double* __restrict__ sz, const double* __restrict__ Ts,
                                                       uint32_t s_pad, uint32_t s_per_split, GridDesc g,
                                                       const uint32_t* __restrict__ cell_start,
                                                       const double* __restrict__ qx, const double* __restrict__ qy,
                                                       const double* __restrict__ qz,
                                                       uint32_t* __restrict__ partial_cnt,
                                                       double* __restrict__ partial_sum, uint32_t res_mask,
                                                       uint32_t n_tiles_total, const uint8_t* __restrict__ keep,
                                                       uint32_t n_tiles_launch, uint32_t n_split,
                                                       const uint8_t* __restrict__ redo /* [tile][s_pad] or null: only the flagged pairs (m3d_reg_cache.hip) */) {
    __shared__ uint32_t red[4][64];
    __shared__ double reds[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware block -> (tile, split) map.  Workgroup w runs on XCD w % 8 and every XCD has its own
    // 4 MB L2, so all splits of one tile (same source points, different hypotheses -> same target
    // neighbourhood, because surviving hypotheses are near-identical poses) go to ONE XCD, and an XCD
    // works through its tiles one after the other: the few tiles in flight per XCD keep their
    // neighbour lists L2-resident instead of streaming them from the fabric for every hypothesis.
    const uint32_t xcd = blockIdx.x % 8u, jq = blockIdx.x / 8u;
    const uint32_t tile_local = (jq / n_split) * 8u + xcd, split = jq % n_split;
    if (tile_local >= n_tiles_launch) return;   // whole block (uniform)
    // a launch covers the tiles whose index mod 32 is in res_mask (the pruning phases: launch_reg_validate)
    const uint32_t tile = phase_tile(tile_local, res_mask);
    if (tile >= n_tiles_total) return;
    const size_t base = (size_t)tile * kRegTile + (size_t)wave * (64 * kRegP) + lane;
    double x[kRegP], y[kRegP], z[kRegP];
#pragma unroll
    for (int j = 0; j < kRegP; ++j) {
        x[j] = sx[base + 64 * j];
        y[j] = sy[base + 64 * j];
        z[j] = sz[base + 64 * j];
    }
    // s_per_split < 64 (a call with a handful of hypotheses -- the reference's default confidence validates ~5: one group,
    // which one workgroup per tile would walk serially on a quarter-filled chip): the splits cut the ONE group, split k
    // takes hypotheses [k s_per_split, (k + 1) s_per_split) of it, the last one the padding behind them as well
    const bool sub = s_per_split < 64u;
    const uint32_t s0 = sub ? 0u : split * s_per_split;
    const uint32_t s1 = sub ? min(64u, s_pad) : min(s0 + s_per_split, s_pad);
    const uint32_t ss0 = sub ? split * s_per_split : 0u, ss1 = sub && split + 1u < n_split ? ss0 + s_per_split : 64u;
    for (uint32_t sb = s0; sb < s1; sb += 64) {
        uint32_t acc = 0;
        double acc_sum = 0.0;
        // the 64 keep flags of the block in one load (a byte load + wait per hypothesis stalled the wave for a memory
        // latency each), and the next hypothesis' transformation requested while this one is evaluated
        unsigned long long keep_mask = keep ? __ballot(keep[sb + (uint32_t)lane] != 0) : ~0ull;
        const unsigned long long redo_mask = redo ? __ballot(redo[(size_t)tile * s_pad + sb + (uint32_t)lane] != 0) : ~0ull;
        keep_mask &= redo_mask;
        if (redo && keep_mask == 0ull) continue;   // (wave-uniform, the same in all four waves: no barrier is skipped by one of them alone)
        double tn[12];
        {
            const double* __restrict__ T = Ts + (size_t)(sb + ss0) * kRegTStride;
#pragma unroll
            for (int k = 0; k < 12; ++k) tn[k] = T[k];
        }
        for (uint32_t ss = ss0; ss < ss1; ++ss) {
            double t[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) t[k] = tn[k];
            {   // (the record behind the last one exists: Ts holds s_pad + 1 records)
                const double* __restrict__ T = Ts + (size_t)(sb + ss + 1u) * kRegTStride;
#pragma unroll
                for (int k = 0; k < 12; ++k) tn[k] = T[k];
            }
            uint32_t cnt = 0;
            double sum = 0.0;
            const bool run = (keep_mask >> ss) & 1ull;
            if (t[0] == t[0] && run) {  // padding records are NaN (wave-uniform branch)
#pragma unroll
                for (int j = 0; j < kRegP; ++j) {
                    const double px = ((t[0] * x[j] + t[1] * y[j]) + t[2] * z[j]) + t[3];
                    const double py = ((t[4] * x[j] + t[5] * y[j]) + t[6] * z[j]) + t[7];
                    const double pz = ((t[8] * x[j] + t[9] * y[j]) + t[10] * z[j]) + t[11];
                    const double d2 = nearest_d2<SCREEN>(g, cell_start, qx, qy, qz, px, py, pz);
                    const bool f = d2 < g.r2;
                    cnt += (uint32_t)__popcll(__ballot(f));
                    sum += f ? d2 : 0.0;
                }
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            }
            acc = ((uint32_t)lane == ss) ? cnt : acc;
            acc_sum = ((uint32_t)lane == ss) ? sum : acc_sum;
        }
        red[wave][lane] = acc;
        reds[wave][lane] = acc_sum;
        __syncthreads();
        if (wave == 0 && (uint32_t)lane >= ss0 && (uint32_t)lane < ss1 && ((redo_mask >> lane) & 1ull)) {
            partial_cnt[(size_t)tile * s_pad + sb + lane] =
                (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            partial_sum[(size_t)tile * s_pad + sb + lane] =
                (reds[0][lane] + reds[1][lane]) + (reds[2][lane] + reds[3][lane]);
        }
        __syncthreads();
    }
}


// The walk on a LIST of (tile, hypothesis) pairs -- the pairs the candidate cache has flagged, of the hypotheses still in the race
// (m3d_reg_cache.hip).  reg_validate_k's own arithmetic and record (same per-wave count and sum, same fold over the four waves),
// but one pair per workgroup trip instead of one (tile, 64 hypotheses) block walking its few flagged hypotheses one after the
// other: the flagged pairs are the far poses' -- the dearest walks there are -- and a handful per block.
__global__ __launch_bounds__(256) void redo_pairs_k(const uint8_t* __restrict__ redo, const uint8_t* __restrict__ keep, uint32_t s_pad,
                                                    uint32_t res_mask, uint32_t n_tiles_total, uint32_t n_tiles_launch,
                                                    uint32_t* __restrict__ pairs, uint32_t* __restrict__ n_pairs) {
    const uint32_t tile_local = blockIdx.y;
    if (tile_local >= n_tiles_launch) return;
    const uint32_t tile = phase_tile(tile_local, res_mask);
    if (tile >= n_tiles_total) return;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const bool take = s < s_pad && (!keep || keep[s]) && redo[(size_t)tile * s_pad + s];
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(n_pairs, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    if (take) pairs[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = tile * s_pad + s;
}
template <bool SCREEN>
__global__ __launch_bounds__(256) void reg_validate_pairs_k(const double* __restrict__ sx, const double* __restrict__ sy,
                                                             const double* __restrict__ sz, const double* __restrict__ Ts,
                                                             uint32_t s_pad, GridDesc g, const uint32_t* __restrict__ cell_start,
                                                             const double* __restrict__ qx, const double* __restrict__ qy,
                                                             const double* __restrict__ qz, uint32_t* __restrict__ partial_cnt,
                                                             double* __restrict__ partial_sum, const uint32_t* __restrict__ pairs,
                                                             const uint32_t* __restrict__ n_pairs) {
    __shared__ uint32_t red[4];
    __shared__ double reds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = *n_pairs;
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        const uint32_t e = pairs[p], tile = e / s_pad, s = e % s_pad;
        const size_t base = (size_t)tile * kRegTile + (size_t)wave * 64 + lane;
        const double x = sx[base], y = sy[base], z = sz[base];
        const double* __restrict__ t = Ts + (size_t)s * kRegTStride;
        uint32_t cnt = 0;
        double sum = 0.0;
        if (t[0] == t[0]) {
            const double px = ((t[0] * x + t[1] * y) + t[2] * z) + t[3];
            const double py = ((t[4] * x + t[5] * y) + t[6] * z) + t[7];
            const double pz = ((t[8] * x + t[9] * y) + t[10] * z) + t[11];
            const double d2 = nearest_d2<SCREEN>(g, cell_start, qx, qy, qz, px, py, pz);
            const bool f = d2 < g.r2;
            cnt = (uint32_t)__popcll(__ballot(f));
            sum = f ? d2 : 0.0;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        }
        if (lane == 0) {
            red[wave] = cnt;
            reds[wave] = sum;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            partial_cnt[e] = (red[0] + red[1]) + (red[2] + red[3]);
            partial_sum[e] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
        }
        __syncthreads();
    }
}

// sums[s] = sum over tiles of partial_sum[tile][s] in tile order (deterministic)
// (workgroup = 64 hypotheses x kFoldSlices interleaved slices of the tiles, folded in LDS in a fixed order: a thread per
// hypothesis walking all ~800 tiles alone took 0.2 ms per call)
constexpr int kFoldSlices = 16;
__global__ __launch_bounds__(64 * kFoldSlices) void reduce_sums_k(const double* __restrict__ partial_sum, uint32_t n_tiles,
                                                                  uint32_t s_pad, double* __restrict__ sums) {
    __shared__ double sm[kFoldSlices][64];
    const uint32_t lane = threadIdx.x & 63u, sl = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 64u + lane;   // (s_pad is a multiple of 64)
    double acc = 0.0;
    for (uint32_t t = sl; t < n_tiles; t += kFoldSlices) acc += partial_sum[(size_t)t * s_pad + s];
    sm[sl][lane] = acc;
    __syncthreads();
    if (sl == 0) {
        double a = 0.0;
        for (int k = 0; k < kFoldSlices; ++k) a += sm[k][lane];
        sums[s] = a;
    }
}

// Bound-and-prune inside a chunk, against the best hypothesis of EARLIER chunks (count best_cnt, order-free sum of squared
// distances best_sum2).  After the tiles whose index mod 8 is in done_mask: a = matches so far, q = sum so far, rem = real
// source points on the tiles still to come.  The hypothesis stays in the race when it can still reach MORE than best_cnt
// matches, or exactly best_cnt with a smaller sum -- the replay's order is (fitness, then rmse = sqrt(sum / count)), the
// sum only grows, so q above the incumbent's sum decides a tie against it.  A dropped hypothesis reports its partial
// count, which is below best_cnt, so the replay never selects it.  limit = best_sum2 widened by far more than the
// summation-order tolerance of the replay's comparison.  (On C4 this second rule drops little: the survivors are good
// poses whose sums lie within a factor of two of each other -- the nearest-neighbour distance on a densely sampled
// surface hardly grows with a tangential shift.)
__global__ __launch_bounds__(64 * kFoldSlices) void reg_keep_k(const uint32_t* __restrict__ partial_cnt,
                                                               const double* __restrict__ partial_sum, uint32_t n_tiles,
                                                               uint32_t done_mask, uint32_t s_pad, uint32_t points_rem,
                                                               bool rem_exact, uint32_t best_cnt, double limit,
                                                               uint8_t* __restrict__ keep, int first) {
    __shared__ double smq[kFoldSlices][64];
    __shared__ uint32_t sma[kFoldSlices][64];
    const uint32_t lane = threadIdx.x & 63u, sl = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 64u + lane;   // (s_pad is a multiple of 64)
    const bool live = first || keep[s];
    uint32_t a = 0;
    double q = 0.0;
    if (live)
        for (uint32_t t = sl; t < n_tiles; t += kFoldSlices)
            if ((done_mask >> (t & 31u)) & 1u) {
                a += partial_cnt[(size_t)t * s_pad + s];
                q += partial_sum[(size_t)t * s_pad + s];
            }
    sma[sl][lane] = a;
    smq[sl][lane] = q;
    __syncthreads();
    if (sl == 0 && live) {
        a = 0;
        q = 0.0;
        for (int k = 0; k < kFoldSlices; ++k) {
            a += sma[k][lane];
            q += smq[k][lane];
        }
        const uint64_t bound = (uint64_t)a + points_rem;
        keep[s] = (bound > best_cnt || (bound == best_cnt && (!rem_exact || q <= limit))) ? 1 : 0;
    }
}

// best_cnt / best_sum2: inlier count and order-free sum of squared distances of the best hypothesis of EARLIER chunks
// (best_cnt 0: nothing to prune against).  n_points: real source points.  keep: s_pad bytes of scratch.
// Returns the number of partial rows (what the reduce kernels fold).
uint32_t launch_reg_validate(const CloudView& src, const double* Ts, uint32_t s_pad, const GridDesc& g,
                             const uint32_t* cell_start, const double* qx, const double* qy, const double* qz,
                             uint32_t* partial_cnt, double* partial_sum, double* sums, uint32_t best_cnt,
                             uint32_t n_points, uint8_t* keep, hipStream_t s, double best_sum2, uint32_t n_hyp,
                             const RegCache* cache) {
    if (!s_pad || !src.n_pad) return 0;
    const uint32_t groups = s_pad / 64;
    const uint32_t n_tiles = src.n_pad / kRegTile;
    const uint32_t tile_points = (uint32_t)kRegTile;
    // walk = false: the candidate cache alone (m3d_reg_cache.hip) -- exact records where every query of the pair holds a
    // certificate, an upper bound of the count and a lower bound of the sum (and a flag in cache->redo) where not; walk = true
    // without a cache: reg_validate_k on every pair; with one: on the flagged pairs only (`resolve`)
    auto launch = [&](uint32_t res_mask, const uint8_t* kp, bool resolve) {
        // (upper bound of the tiles of the launch; the kernel drops indices past the last tile)
        const uint32_t tiles = (n_tiles + 31) / 32 * (uint32_t)__builtin_popcount(res_mask);
        // enough blocks to fill the chip several times over (the pruning phases launch an eighth of the tiles: 16 384 blocks
        // instead of 2048 is 4 % on C4), and at least ~40 splits per tile so that the ~160 blocks an XCD holds at a time
        // belong to a handful of tiles (see the block map in the kernel)
        const uint32_t want = std::max<uint32_t>(kRegMinSplits, (16384 + tiles - 1) / tiles);
        const uint32_t splits = std::min(want, groups);
        const uint32_t gps = (groups + splits - 1) / splits;
        uint32_t nsplit = (groups + gps - 1) / gps;
        uint32_t per_split = gps * 64;
        if (groups == 1 && n_hyp && n_hyp <= 32 && !cache) {   // a handful of hypotheses: cut the one group (see the kernel; the cache works on whole groups)
            per_split = std::max<uint32_t>(1, n_hyp / 8);
            nsplit = (n_hyp + per_split - 1) / per_split;
        }
        const uint32_t slots = (tiles + 7) / 8;
        if (cache && !resolve) {
            launch_reg_validate_cached(src, Ts, s_pad, per_split, nsplit, slots, g, *cache, partial_cnt, partial_sum, res_mask,
                                       n_tiles, kp, tiles, cache->redo, s);
            return;
        }
        const uint8_t* redo = cache ? cache->redo : nullptr;
        if (cache && cache->pairs) {   // the flagged pairs as a list, one pair per workgroup trip
            (void)hipMemsetAsync(cache->n_pairs, 0, sizeof(uint32_t), s);
            redo_pairs_k<<<dim3((s_pad + 255) / 256, tiles), 256, 0, s>>>(redo, kp, s_pad, res_mask, n_tiles, tiles, cache->pairs, cache->n_pairs);
            if (g.nl32 && g.nl_rec && g.nl_sorted && g.nl_start && g.nl_hdr)
                reg_validate_pairs_k<true><<<8192, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, g, cell_start, qx, qy, qz, partial_cnt, partial_sum,
                                                               cache->pairs, cache->n_pairs);
            else
                reg_validate_pairs_k<false><<<8192, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, g, cell_start, qx, qy, qz, partial_cnt, partial_sum,
                                                                cache->pairs, cache->n_pairs);
            return;
        }
        if (g.nl32 && g.nl_rec && g.nl_sorted && g.nl_start && g.nl_hdr)
            reg_validate_k<true><<<slots * 8 * nsplit, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, per_split, g, cell_start, qx, qy,
                                                                   qz, partial_cnt, partial_sum, res_mask, n_tiles, kp, tiles, nsplit, redo);
        else
            reg_validate_k<false><<<slots * 8 * nsplit, 256, 0, s>>>(src.x, src.y, src.z, Ts, s_pad, per_split, g, cell_start, qx, qy,
                                                                    qz, partial_cnt, partial_sum, res_mask, n_tiles, kp, tiles, nsplit, redo);
    };
    if (best_cnt == 0 || n_tiles < 16) {
        launch(0xFFFFFFFFu, nullptr, false);
        if (cache) launch(0xFFFFFFFFu, nullptr, true);   // nothing to prune against: every flagged pair is walked
    } else {
        // four phases: an eighth of the tiles, another eighth, a quarter, the remaining half; the hypotheses still in
        // the race are re-assessed in between (reg_keep_k)
        // (residue classes of the tile index mod 32.  Finer first phases -- 1/16, 1/16, 1/8, ... and 1/32, 1/32, 1/16, ... -- were
        // measured with the cache on C4's forced run: 125 / 127 ms against 128 with these four; eight phases of an eighth each:
        // 116 against 112 -- the launches' tails cost what the pairs save.  profiles/r06_reg_cache.txt)
        static const uint32_t kPhase[4] = {0x01010101u, 0x10101010u, 0x44444444u, 0xAAAAAAAAu};
        const int n_phases = 4;
        // real source points on the tiles of a residue class (the real points are the first n_points slots)
        auto points_on = [&](uint32_t mask) {
            uint64_t n = 0;
            for (uint32_t t = 0; t < n_tiles; ++t)
                if ((mask >> (t & 31u)) & 1u) {
                    const uint64_t lo = (uint64_t)t * tile_points;
                    n += (uint64_t)std::min<uint64_t>(tile_points, n_points > lo ? n_points - lo : 0);
                }
            return (uint32_t)std::min<uint64_t>(n, n_points);
        };
        const double limit = best_sum2 * (1.0 + 1e-6);
        uint32_t done = 0;
        const int last = n_phases - 1;
        for (int ph = 0; ph <= last; ++ph) {
            launch(kPhase[ph], ph == 0 ? nullptr : keep, false);
            done |= kPhase[ph];
            const uint32_t rem = points_on(~done);
            auto assess = [&](int first) {
                reg_keep_k<<<s_pad / 64, 64 * kFoldSlices, 0, s>>>(partial_cnt, partial_sum, n_tiles, done, s_pad, rem, true, best_cnt,
                                                               limit, keep, first);
            };
            if (cache) {
                // The records of a flagged pair are BOUNDS (count from above, sum from below): reg_keep_k's rule is as exact on them
                // as on the values themselves -- what it drops can neither beat nor tie the incumbent, and is never walked.  The
                // flagged pairs of the hypotheses still standing are walked now (their bounds may be what keeps them standing)
                // and the hypotheses re-assessed on the values.  (The walk in three steps with an assessment after each -- a far
                // pose that is also a poor one gives itself away on a few tiles -- measured equal: 94.0 against 94.1 ms.)
                assess(ph == 0 ? 1 : 0);
                launch(kPhase[ph], keep, true);
                if (ph < last) assess(0);
            } else if (ph < last) {
                assess(ph == 0 ? 1 : 0);
            }
        }
    }
    reduce_sums_k<<<s_pad / 64, 64 * kFoldSlices, 0, s>>>(partial_sum, n_tiles, s_pad, sums);
    return n_tiles;
}

// per-point nearest squared distance for ONE transformation (device pointer to 12 doubles)
__global__ void reg_min_d2_k(CloudView src, const double* __restrict__ T, GridDesc g,
                             const uint32_t* __restrict__ cell_start, const double* __restrict__ qx,
                             const double* __restrict__ qy, const double* __restrict__ qz,
                             double* __restrict__ best) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= src.n) return;
    double t[12];
    for (int k = 0; k < 12; ++k) t[k] = T[k];
    const double x = src.x[i], y = src.y[i], z = src.z[i];
    const double px = ((t[0] * x + t[1] * y) + t[2] * z) + t[3];
    const double py = ((t[4] * x + t[5] * y) + t[6] * z) + t[7];
    const double pz = ((t[8] * x + t[9] * y) + t[10] * z) + t[11];
    best[i] = nearest_d2(g, cell_start, qx, qy, qz, px, py, pz);
}
void launch_reg_min_d2(const CloudView& src, const double* T, const GridDesc& g, const uint32_t* cell_start,
                       const double* qx, const double* qy, const double* qz, double* best, hipStream_t s) {
    if (src.n) reg_min_d2_k<<<(src.n + 255) / 256, 256, 0, s>>>(src, T, g, cell_start, qx, qy, qz, best);
}

// ordered compaction of the values < limit (3 passes, same structure as compact_*_k)
__global__ __launch_bounds__(256) void vals_count_k(const double* __restrict__ v, uint32_t n, double limit,
                                                     uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cnt = 0;
    for (int r = 0; r < 8; ++r) {
        const uint32_t i = blockIdx.x * 2048u + r * 256u + threadIdx.x;
        const bool f = i < n && v[i] < limit;
        cnt += (uint32_t)__popcll(__ballot(f));
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(256) void vals_write_k(const double* __restrict__ v, uint32_t n, double limit,
                                                     const uint32_t* __restrict__ block_offsets,
                                                     double* __restrict__ out) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t row_base = block_offsets[blockIdx.x];
    for (int r = 0; r < 8; ++r) {
        const uint32_t i = blockIdx.x * 2048u + r * 256u + threadIdx.x;
        const double val = i < n ? v[i] : 0.0;
        const bool f = i < n && val < limit;
        const unsigned long long b = __ballot(f);
        const uint32_t lane_pre = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t rowtot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (f) out[row_base + woff + lane_pre] = val;
        row_base += rowtot;
        __syncthreads();
    }
}
void launch_compact_vals(const double* v, uint32_t n, double limit, uint32_t* block_counts, uint32_t* total,
                         double* out, hipStream_t s) {
    const uint32_t nb = (n + 2047) / 2048;
    if (!nb) {
        (void)hipMemsetAsync(total, 0, sizeof(uint32_t), s);
        return;
    }
    vals_count_k<<<nb, 256, 0, s>>>(v, n, limit, block_counts);
    launch_scan_blocks(block_counts, nb, total, s);
    vals_write_k<<<nb, 256, 0, s>>>(v, n, limit, block_counts, out);
}

// ------------------------------------------------------------------------------------------------
// K10  inlier ratio of the correspondence set
// ------------------------------------------------------------------------------------------------
__global__ void corr_ratio_k(CloudView src, CloudView dst, const uint32_t* __restrict__ corr_src,
                             const uint32_t* __restrict__ corr_dst, uint32_t m, const double* __restrict__ T,
                             double r2, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool f = false;
    if (i < m) {
        double t[12];
        for (int k = 0; k < 12; ++k) t[k] = T[k];
        const uint32_t si = corr_src[i], di = corr_dst[i];
        const double x = src.x[si], y = src.y[si], z = src.z[si];
        const double px = ((t[0] * x + t[1] * y) + t[2] * z) + t[3];
        const double py = ((t[4] * x + t[5] * y) + t[6] * z) + t[7];
        const double pz = ((t[8] * x + t[9] * y) + t[10] * z) + t[11];
        const double ddx = px - dst.x[di], ddy = py - dst.y[di], ddz = pz - dst.z[di];
        f = ((ddx * ddx + ddy * ddy) + ddz * ddz) < r2;
    }
    const uint32_t c = (uint32_t)__popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
void launch_corr_ratio(const CloudView& src, const CloudView& dst, const uint32_t* corr_src,
                       const uint32_t* corr_dst, uint32_t m, const double* T, double r2, uint32_t* count,
                       hipStream_t s) {
    (void)hipMemsetAsync(count, 0, sizeof(uint32_t), s);
    if (m) corr_ratio_k<<<(m + 255) / 256, 256, 0, s>>>(src, dst, corr_src, corr_dst, m, T, r2, count);
}

// ------------------------------------------------------------------------------------------------
// umeyama sums for n correspondences (LeastSquareSolver): AoS inputs, fixed two-level tree
// ------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void tree_reduce_256(double (&acc)[NV], double* sm) {
    for (int k = 0; k < NV; ++k) sm[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < NV; ++k) sm[k * 256 + threadIdx.x] += sm[k * 256 + threadIdx.x + off];
        __syncthreads();
    }
}
// pass 0: sums of src and dst coordinates (6 values); pass 1: with means -> 9 covariance + 3 variance
template <int PASS>
__global__ __launch_bounds__(256) void kabsch_sums_k(const double* __restrict__ src,
                                                      const double* __restrict__ dst, uint32_t n,
                                                      const double* __restrict__ sums0,
                                                      double* __restrict__ partial) {
    constexpr int NV = PASS == 0 ? 6 : 12;
    __shared__ double sm[NV * 256];
    double acc[NV];
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    if (PASS == 1) {
        const double one_over_n = 1.0 / (double)n;
        for (int k = 0; k < 3; ++k) {
            ms[k] = sums0[k] * one_over_n;
            md[k] = sums0[3 + k] * one_over_n;
        }
    }
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 256u * 256u) {
        double s[3], d[3];
        for (int k = 0; k < 3; ++k) {
            s[k] = src[3 * (size_t)i + k];
            d[k] = dst[3 * (size_t)i + k];
        }
        if (PASS == 0) {
            for (int k = 0; k < 3; ++k) {
                acc[k] += s[k];
                acc[3 + k] += d[k];
            }
        } else {
            for (int k = 0; k < 3; ++k) {
                s[k] -= ms[k];
                d[k] -= md[k];
            }
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) acc[3 * r + c] += d[r] * s[c];
                acc[9 + r] += s[r] * s[r];
            }
        }
    }
    tree_reduce_256<NV>(acc, sm);
    if (threadIdx.x < NV) partial[blockIdx.x * 16 + threadIdx.x] = sm[threadIdx.x * 256];
}
template <int NV>
__global__ __launch_bounds__(256) void kabsch_final_k(const double* __restrict__ partial,
                                                       double* __restrict__ sums) {
    __shared__ double sm[NV * 256];
    double acc[NV];
    for (int k = 0; k < NV; ++k) acc[k] = partial[threadIdx.x * 16 + k];
    tree_reduce_256<NV>(acc, sm);
    if (threadIdx.x < NV) sums[threadIdx.x] = sm[threadIdx.x * 256];
}
void launch_kabsch_sums(const double* src, const double* dst, uint32_t n, double* partial,
                        double* sums /* 6 + 12 */, hipStream_t s) {
    kabsch_sums_k<0><<<256, 256, 0, s>>>(src, dst, n, nullptr, partial);
    kabsch_final_k<6><<<1, 256, 0, s>>>(partial, sums);
    kabsch_sums_k<1><<<256, 256, 0, s>>>(src, dst, n, sums, partial);
    kabsch_final_k<12><<<1, 256, 0, s>>>(partial, sums + 6);
}

// ------------------------------------------------------------------------------------------------
// ICP: nearest target point WITH its index, Kabsch sums over the correspondences,
// in-place transformation of the moving cloud
// ------------------------------------------------------------------------------------------------
// (squared distance, original target index) of the nearest target point; lowest index on exact ties (what a
// serial scan in index order finds).  Same two phases as nearest_d2; the neighbour lists carry the original
// index in w, the row scan takes it from cell_orig.
__device__ __forceinline__ void nearest_idx(const GridDesc& g, const uint32_t* __restrict__ cell_start,
                                            const double* __restrict__ qx, const double* __restrict__ qy,
                                            const double* __restrict__ qz, const uint32_t* __restrict__ cell_orig,
                                            double px, double py, double pz, double* d2_out, uint32_t* idx_out) {
    int ix, iy, iz;
    double best = INFINITY;
    uint32_t bo = 0xFFFFFFFFu;
    *d2_out = best;
    *idx_out = bo;
    if (!cell_of(g, px, py, pz, g.K, &ix, &iy, &iz)) return;
    bool need_scan = true;
    if (g.nl_start) {
        const uint32_t cell = ((uint32_t)iz * g.ny + (uint32_t)iy) * g.nx + (uint32_t)ix;
        const uint32_t b = g.nl_start[cell], e = g.nl_start[cell + 1];
        for (uint32_t c = b; c < e; ++c) {
            const double4 q = g.nl_pts[c];
            const double ddx = px - q.x, ddy = py - q.y, ddz = pz - q.z;
            const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
            const uint32_t o = (uint32_t)q.w;
            if (d2 < best || (d2 == best && o < bo)) {
                best = d2;
                bo = o;
            }
        }
        need_scan = !(best < g.h2_in) && g.K > 1;
    }
    if (need_scan) {
        // (2K+1)^3 block, only the cells that can hold a point at distance <= min(best, r^2) (see nearest_phase2)
        const int K = g.K;
        const double h = 1.0 / g.inv_h;
        const double fx = (px - g.ox) * g.inv_h - (double)ix, fy = (py - g.oy) * g.inv_h - (double)iy,
                     fz = (pz - g.oz) * g.inv_h - (double)iz;
        for (int dz = -K; dz <= K; ++dz) {
            const double gz = axis_gap(dz, fz, h);
            const double gz2 = gz * gz;
            if (!(gz2 <= (best < g.r2 ? best : g.r2))) continue;
            for (int dy = -K; dy <= K; ++dy) {
                const double gy = axis_gap(dy, fy, h);
                const double lim = best < g.r2 ? best : g.r2;
                const double rem = lim - (gz2 + gy * gy);
                if (!(rem >= 0.0)) continue;
                int lo = 0, hi = 0;
                while (lo > -K && axis_gap(lo - 1, fx, h) * axis_gap(lo - 1, fx, h) <= rem) --lo;
                while (hi < K && axis_gap(hi + 1, fx, h) * axis_gap(hi + 1, fx, h) <= rem) ++hi;
                const uint32_t row = ((uint32_t)(iz + dz) * g.ny + (uint32_t)(iy + dy)) * g.nx + (uint32_t)ix;
                const uint32_t b = cell_start[row + lo], e = cell_start[row + hi + 1];
                for (uint32_t c = b; c < e; ++c) {
                    const double ddx = px - qx[c], ddy = py - qy[c], ddz = pz - qz[c];
                    const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    const uint32_t o = cell_orig[c];
                    if (d2 < best || (d2 == best && o < bo)) {
                        best = d2;
                        bo = o;
                    }
                }
            }
        }
    }
    *d2_out = best;
    *idx_out = bo;
}

// GetRegistrationResultAndCorrespondences, per point: nn[i] = target index or 0xFFFFFFFF, d2[i] = its squared
// distance (+inf when nothing lies within the radius)
__global__ void icp_nn_k(const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
                         uint32_t n, GridDesc g, const uint32_t* __restrict__ cell_start,
                         const double* __restrict__ qx, const double* __restrict__ qy, const double* __restrict__ qz,
                         const uint32_t* __restrict__ cell_orig, uint32_t* __restrict__ nn, double* __restrict__ d2) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    double best;
    uint32_t bo;
    nearest_idx(g, cell_start, qx, qy, qz, cell_orig, px[i], py[i], pz[i], &best, &bo);
    const bool ok = best < g.r2;
    nn[i] = ok ? bo : 0xFFFFFFFFu;
    d2[i] = ok ? best : INFINITY;
}
void launch_icp_nn(const double* px, const double* py, const double* pz, uint32_t n, const GridDesc& g,
                   const uint32_t* cell_start, const double* qx, const double* qy, const double* qz,
                   const uint32_t* cell_orig, uint32_t* nn, double* d2, hipStream_t s) {
    if (n) icp_nn_k<<<(n + 255) / 256, 256, 0, s>>>(px, py, pz, n, g, cell_start, qx, qy, qz, cell_orig, nn, d2);
}

// GetRegistrationResultAndCorrespondences' error2 and correspondence count in one pass over the per-point
// squared distances: out[0] = sum of d2 < r2 (order-free tree sum: ICP only compares rmse differences with
// 1e-6 and reports rmse at tolerance level; the RANSAC tie rule, which needs the serial order, has its own
// path), out[1] = their number (exact: a sum of ones below 2^53).
__global__ __launch_bounds__(256) void icp_err_k(const double* __restrict__ d2, uint32_t n, double r2,
                                                  double* __restrict__ partial) {
    __shared__ double sm[2 * 256];
    double acc[2] = {0.0, 0.0};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 256u * 256u) {
        const double v = d2[i];
        if (v < r2) {
            acc[0] += v;
            acc[1] += 1.0;
        }
    }
    tree_reduce_256<2>(acc, sm);
    if (threadIdx.x < 2) partial[blockIdx.x * 16 + threadIdx.x] = sm[threadIdx.x * 256];
}
void launch_icp_err(const double* d2, uint32_t n, double r2, double* partial, double* out2, hipStream_t s) {
    icp_err_k<<<256, 256, 0, s>>>(d2, n, r2, partial);
    kabsch_final_k<2><<<1, 256, 0, s>>>(partial, out2);
}

// Eigen::umeyama sums over the correspondence set (moving point i, target point nn[i]): PASS 0 = the six
// coordinate sums, PASS 1 = covariance d s^T (9) and |s|^2 (3) about the means.  Order-free tree sums (the
// n-point Kabsch parity bar is 1e-9).
template <int PASS>
__global__ __launch_bounds__(256) void icp_sums_k(const double* __restrict__ px, const double* __restrict__ py,
                                                   const double* __restrict__ pz, uint32_t n, CloudView dst,
                                                   const uint32_t* __restrict__ nn, const double* __restrict__ count,
                                                   const double* __restrict__ sums0, double* __restrict__ partial) {
    constexpr int NV = PASS == 0 ? 6 : 12;
    __shared__ double sm[NV * 256];
    double acc[NV];
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    if (PASS == 1) {
        const double one_over_n = 1.0 / count[0];
        for (int k = 0; k < 3; ++k) {
            ms[k] = sums0[k] * one_over_n;
            md[k] = sums0[3 + k] * one_over_n;
        }
    }
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 256u * 256u) {
        const uint32_t j = nn[i];
        if (j == 0xFFFFFFFFu) continue;
        double s[3] = {px[i], py[i], pz[i]}, d[3] = {dst.x[j], dst.y[j], dst.z[j]};
        if (PASS == 0) {
            for (int k = 0; k < 3; ++k) {
                acc[k] += s[k];
                acc[3 + k] += d[k];
            }
        } else {
            for (int k = 0; k < 3; ++k) {
                s[k] -= ms[k];
                d[k] -= md[k];
            }
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) acc[3 * r + c] += d[r] * s[c];
                acc[9 + r] += s[r] * s[r];
            }
        }
    }
    tree_reduce_256<NV>(acc, sm);
    if (threadIdx.x < NV) partial[blockIdx.x * 16 + threadIdx.x] = sm[threadIdx.x * 256];
}
// sums: 18 doubles laid out like launch_kabsch_sums (6 coordinate sums, 9 covariance sums, 3 squared sums)
void launch_icp_sums(const double* px, const double* py, const double* pz, uint32_t n, const CloudView& dst,
                     const uint32_t* nn, const double* count, double* partial, double* sums, hipStream_t s) {
    icp_sums_k<0><<<256, 256, 0, s>>>(px, py, pz, n, dst, nn, count, nullptr, partial);
    kabsch_final_k<6><<<1, 256, 0, s>>>(partial, sums);
    icp_sums_k<1><<<256, 256, 0, s>>>(px, py, pz, n, dst, nn, count, sums, partial);
    kabsch_final_k<12><<<1, 256, 0, s>>>(partial, sums + 6);
}

// GetInformationMatrixFromPointClouds: the nine coordinate moments of the matched TARGET points
// (x y z xx yy zz xy xz yz); the 6 x 6 matrix is assembled from them and the count on the host.
__global__ __launch_bounds__(256) void info_sums_k(uint32_t n, CloudView dst, const uint32_t* __restrict__ nn,
                                                    double* __restrict__ partial) {
    constexpr int NV = 9;
    __shared__ double sm[NV * 256];
    double acc[NV];
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 256u * 256u) {
        const uint32_t j = nn[i];
        if (j == 0xFFFFFFFFu) continue;
        const double x = dst.x[j], y = dst.y[j], z = dst.z[j];
        acc[0] += x;
        acc[1] += y;
        acc[2] += z;
        acc[3] += x * x;
        acc[4] += y * y;
        acc[5] += z * z;
        acc[6] += x * y;
        acc[7] += x * z;
        acc[8] += y * z;
    }
    tree_reduce_256<NV>(acc, sm);
    if (threadIdx.x < NV) partial[blockIdx.x * 16 + threadIdx.x] = sm[threadIdx.x * 256];
}
void launch_info_sums(uint32_t n, const CloudView& dst, const uint32_t* nn, double* partial, double* sums,
                      hipStream_t s) {
    info_sums_k<<<256, 256, 0, s>>>(n, dst, nn, partial);
    kabsch_final_k<9><<<1, 256, 0, s>>>(partial, sums);
}

// PointCloud::Transform(update) on the moving cloud, in place (or out of place for the initial pose)
__global__ void icp_transform_k(const double* __restrict__ ix, const double* __restrict__ iy,
                                const double* __restrict__ iz, uint32_t n, const double* __restrict__ T,
                                double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    double t[12];
    for (int k = 0; k < 12; ++k) t[k] = T[k];
    const double x = ix[i], y = iy[i], z = iz[i];
    ox[i] = ((t[0] * x + t[1] * y) + t[2] * z) + t[3];
    oy[i] = ((t[4] * x + t[5] * y) + t[6] * z) + t[7];
    oz[i] = ((t[8] * x + t[9] * y) + t[10] * z) + t[11];
}
void launch_icp_transform(const double* ix, const double* iy, const double* iz, uint32_t n, const double* T_dev,
                          double* ox, double* oy, double* oz, hipStream_t s) {
    if (n) icp_transform_k<<<(n + 255) / 256, 256, 0, s>>>(ix, iy, iz, n, T_dev, ox, oy, oz);
}

// ------------------------------------------------------------------------------------------------
// DetectBoundaryPoints (src/boundary_detection.cpp:21-113), one thread per point
// ------------------------------------------------------------------------------------------------
// Eigen's generic unitOrthogonal on (n, 0) + cross3 ([RECALL], same restatement as the oracle); the boundary
// decision does not depend on the basis (angular gaps are invariant), only its roundings do.
__device__ __forceinline__ void tangent_basis(const double* n, double* u, double* v) {
    const double a[4] = {fabs(n[0]), fabs(n[1]), fabs(n[2]), 0.0};
    int maxi = 0;
    for (int i = 1; i < 4; ++i)
        if (a[i] > a[maxi]) maxi = i;
    int sndi = maxi == 0 ? 1 : 0;
    for (int i = 0; i < 4; ++i)
        if (i != maxi && a[i] > a[sndi]) sndi = i;
    const double src[4] = {n[0], n[1], n[2], 0.0};
    const double invnm = 1.0 / sqrt(src[sndi] * src[sndi] + src[maxi] * src[maxi]);
    double p[4] = {0, 0, 0, 0};
    p[maxi] = -src[sndi] * invnm;
    p[sndi] = src[maxi] * invnm;
    v[0] = p[0];
    v[1] = p[1];
    v[2] = p[2];
    u[0] = n[1] * v[2] - n[2] * v[1];
    u[1] = n[2] * v[0] - n[0] * v[2];
    u[2] = n[0] * v[1] - n[1] * v[0];
}

// Neighbourhood = KDTreeFlann::Search with Radius (all d2 <= r^2), Hybrid (the max_nn nearest with d2 < r^2) or KNN
// (the max_nn nearest), kept sorted by (d2, original index).  Radius / Hybrid: grid cell = 1.001 r, so the 3x3x3
// block covers the radius; KNN: cell from the point density, shells of cells until the k-th distance is certain.
// LDS32: at most 32 neighbours are kept (KNN / Hybrid with max_nn <= 32): the per-thread lists live in LDS ([slot][lane]: no
// bank conflicts between lanes at the same slot) instead of scratch memory -- the sorted insertion moves ~15 entries per
// offer, and that traffic was what the kernel's time went into
template <bool LDS32>
__global__ __launch_bounds__(64) void boundary_k(CloudView c, GridDesc g, const uint32_t* __restrict__ cell_start,
                                                  const double* __restrict__ qx, const double* __restrict__ qy,
                                                  const double* __restrict__ qz, const uint32_t* __restrict__ cell_orig,
                                                  int search, int max_nn, double angle_thr_rad,
                                                  uint8_t* __restrict__ flag, uint8_t* __restrict__ overflow,
                                                  const uint32_t* __restrict__ n_sorted) {
    // thread t takes the t-th point IN GRID ORDER (the points the grid holds: the finite ones; a non-finite point has no
    // neighbours and no flag): the lanes of a wave sit in the same few cells and scan the same rows
    const uint32_t t_sorted = blockIdx.x * 64u + threadIdx.x;
    if (t_sorted >= n_sorted[0]) return;
    const uint32_t i = cell_orig[t_sorted];
    const double px = qx[t_sorted], py = qy[t_sorted], pz = qz[t_sorted];
    int ix, iy, iz;
    if (!cell_of(g, px, py, pz, g.K, &ix, &iy, &iz)) return;
    __shared__ double s_nd[LDS32 ? 32 * 64 : 1];
    __shared__ uint32_t s_ni[LDS32 ? 32 * 64 : 1];
    double nd_l[LDS32 ? 1 : kBoundaryMaxNb];
    uint32_t ni_l[LDS32 ? 1 : kBoundaryMaxNb];
    const int lane_ = (int)threadIdx.x;
    auto ND = [&](int k) -> double& {
        if constexpr (LDS32) return s_nd[k * 64 + lane_];
        else return nd_l[k];
    };
    auto NI = [&](int k) -> uint32_t& {
        if constexpr (LDS32) return s_ni[k * 64 + lane_];
        else return ni_l[k];
    };
    int m = 0;
    const int cap = search == 1 ? kBoundaryMaxNb : max_nn;
    // sorted insertion by (d2, original index); beyond `cap` the farthest entry is replaced (KNN / Hybrid) or
    // the search gives up loudly (Radius has no bound on the neighbourhood)
    auto offer = [&](double d2, uint32_t o) -> bool {
        int pos;
        if (m < cap) {
            pos = m++;
        } else {
            if (search == 1) return false;
            if (!(d2 < ND(m - 1) || (d2 == ND(m - 1) && o < NI(m - 1)))) return true;
            pos = m - 1;
        }
        while (pos > 0 && (d2 < ND(pos - 1) || (d2 == ND(pos - 1) && o < NI(pos - 1)))) {
            ND(pos) = ND(pos - 1);
            NI(pos) = NI(pos - 1);
            --pos;
        }
        ND(pos) = d2;
        NI(pos) = o;
        return true;
    };
    if (search == 0) {
        // KDTreeSearchParamKNN: the max_nn nearest points, no radius.  Shells of cells around the query's cell,
        // ring after ring; every point outside the (2r+1)^3 block is at least r cell edges away, so the search
        // is complete as soon as the k-th distance found is within that bound.
        const double h = 1.0 / g.inv_h;
        const int rmax = (int)max(g.nx, max(g.ny, g.nz));
        for (int r = 0; r <= rmax; ++r) {
            for (int dz = -r; dz <= r; ++dz) {
                const int cz = iz + dz;
                if (cz < 0 || cz >= (int)g.nz) continue;
                for (int dy = -r; dy <= r; ++dy) {
                    const int cy = iy + dy;
                    if (cy < 0 || cy >= (int)g.ny) continue;
                    const bool face = dz == -r || dz == r || dy == -r || dy == r;   // whole x-row is on the shell
                    const uint32_t row = ((uint32_t)cz * g.ny + (uint32_t)cy) * g.nx;
                    const int xl = max(ix - r, 0), xh = min(ix + r, (int)g.nx - 1);
                    // shell = the full row on the four outer faces, otherwise only its two end cells
                    for (int part = 0; part < (face || r == 0 ? 1 : 2); ++part) {
                        int a, b2;
                        if (face || r == 0) {
                            a = xl;
                            b2 = xh;
                        } else {
                            a = b2 = part == 0 ? ix - r : ix + r;
                            if (a < 0 || a >= (int)g.nx) continue;
                        }
                        const uint32_t b = cell_start[row + (uint32_t)a], e = cell_start[row + (uint32_t)b2 + 1];
                        for (uint32_t t = b; t < e; ++t) {
                            const double ddx = px - qx[t], ddy = py - qy[t], ddz = pz - qz[t];
                            offer((ddx * ddx + ddy * ddy) + ddz * ddz, cell_orig[t]);
                        }
                    }
                }
            }
            // strictly inside the bound (minus the cell-assignment slack): an unseen point can then neither be
            // closer nor tie with the k-th neighbour
            const double reach = ((double)r - 1e-6) * h;
            if (m >= cap && reach > 0.0 && ND(m - 1) < reach * reach) break;
        }
    } else {
        const int K = g.K;
        for (int dz = -K; dz <= K; ++dz)
            for (int dy = -K; dy <= K; ++dy) {
                const uint32_t row = ((uint32_t)(iz + dz) * g.ny + (uint32_t)(iy + dy)) * g.nx + (uint32_t)ix;
                const uint32_t b = cell_start[row - K], e = cell_start[row + K + 1];
                for (uint32_t t = b; t < e; ++t) {
                    const double ddx = px - qx[t], ddy = py - qy[t], ddz = pz - qz[t];
                    const double d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (!(search == 1 ? d2 <= g.r2 : d2 < g.r2)) continue;
                    if (!offer(d2, cell_orig[t])) {
                        overflow[0] = 1;
                        return;
                    }
                }
            }
    }
    if (m < 3) return;   // :96-99
    double nrm[3];
    if (c.nx) {
        nrm[0] = c.nx[i];
        nrm[1] = c.ny[i];
        nrm[2] = c.nz[i];
    } else {   // stand-in for Open3D EstimateNormals(param): covariance of the same neighbourhood, J3x3
        double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < m; ++t) {
            const double x = c.x[NI(t)], y = c.y[NI(t)], z = c.z[NI(t)];
            s[0] += x;
            s[1] += y;
            s[2] += z;
            s[3] += x * x;
            s[4] += x * y;
            s[5] += x * z;
            s[6] += y * y;
            s[7] += y * z;
            s[8] += z * z;
        }
        const double inv = 1.0 / (double)m;
        for (int t = 0; t < 9; ++t) s[t] *= inv;
        double Cm[9];
        Cm[0] = s[3] - s[0] * s[0];
        Cm[1] = s[4] - s[0] * s[1];
        Cm[2] = s[5] - s[0] * s[2];
        Cm[4] = s[6] - s[1] * s[1];
        Cm[5] = s[7] - s[1] * s[2];
        Cm[8] = s[8] - s[2] * s[2];
        Cm[3] = Cm[1];
        Cm[6] = Cm[2];
        Cm[7] = Cm[5];
        j3x3_smallest_eigvec(Cm, nrm);
    }
    double u[3], v[3];
    tangent_basis(nrm, u, v);
    // angles of the neighbours in the tangent plane (:33-41), kept sorted as they come
    // (the angles go where the distances were: those are not needed any more)
    int na = 0;
    for (int t = 0; t < m; ++t) {
        const double dx = c.x[NI(t)] - px, dy = c.y[NI(t)] - py, dz = c.z[NI(t)] - pz;
        if (dx == 0.0 && dy == 0.0 && dz == 0.0) continue;
        const double a = atan2((v[0] * dx + v[1] * dy) + v[2] * dz, (u[0] * dx + u[1] * dy) + u[2] * dz);
        int pos = na++;   // na <= t + 1: writing ND(pos) never touches an unread nd[]
        while (pos > 0 && a < ND(pos - 1)) {
            ND(pos) = ND(pos - 1);
            --pos;
        }
        ND(pos) = a;
    }
    if (na == 0) return;
    double max_dif = 0.0;
    for (int t = 0; t + 1 < na; ++t) {
        const double dif = ND(t + 1) - ND(t);
        if (max_dif < dif) max_dif = dif;
    }
    const double wrap = 2 * 3.14159265358979323846 - ND(na - 1) + ND(0);
    if (max_dif < wrap) max_dif = wrap;
    if (max_dif > angle_thr_rad) flag[i] = 1;
}
void launch_boundary(const CloudView& c, const GridDesc& g, const uint32_t* cell_start, const double* qx,
                     const double* qy, const double* qz, const uint32_t* cell_orig, int search, int max_nn,
                     double angle_threshold_deg, uint8_t* flag, uint8_t* overflow, hipStream_t s, const uint32_t* n_sorted) {
    if (!c.n) return;
    const double thr = angle_threshold_deg * 3.14159265358979323846 / 180.0;   // :62
    if (search != 1 && max_nn <= 32)
        boundary_k<true><<<(c.n + 63) / 64, 64, 0, s>>>(c, g, cell_start, qx, qy, qz, cell_orig, search, max_nn, thr, flag,
                                                        overflow, n_sorted);
    else
        boundary_k<false><<<(c.n + 63) / 64, 64, 0, s>>>(c, g, cell_start, qx, qy, qz, cell_orig, search, max_nn, thr, flag,
                                                         overflow, n_sorted);
}

// ------------------------------------------------------------------------------------------------
// K11  exact nearest neighbour in descriptor space (mutual-NN matcher)
// ------------------------------------------------------------------------------------------------
// One query per lane, its DIM coordinates in VGPRs; database descriptors are wave-uniform and come
// in through scalar loads.  Squared L2 accumulated in dimension order, one rounding per product and
// per sum (nanoflann L2_Simple_Adaptor order), strict `<` while scanning ascending indices keeps the
// lowest index among equal distances.  blockIdx.y splits the database; nn_merge_k merges the splits.
template <int DIM>
__global__ __launch_bounds__(256) void nn_k(const double* __restrict__ q, uint32_t nq,
                                             const double* __restrict__ db, uint32_t ndb,
                                             uint32_t db_per_split, double* __restrict__ best_d,
                                             uint32_t* __restrict__ best_i) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double qv[DIM];
    const uint32_t ii = i < nq ? i : (nq - 1);
#pragma unroll
    for (int k = 0; k < DIM; ++k) qv[k] = q[(size_t)ii * DIM + k];
    const uint32_t j0 = blockIdx.y * db_per_split;
    const uint32_t j1 = min(ndb, j0 + db_per_split);
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = j0; j < j1; ++j) {
        const double* __restrict__ d = db + (size_t)j * DIM;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            const double df = qv[k] - d[k];
            acc += df * df;
        }
        if (acc < bd) {
            bd = acc;
            bi = j;
        }
    }
    if (i < nq) {
        best_d[(size_t)blockIdx.y * nq + i] = bd;
        best_i[(size_t)blockIdx.y * nq + i] = bi;
    }
}
// generic dimension: query row staged per lane in LDS (slower; FPFH's 33 uses the template above)
__global__ __launch_bounds__(64) void nn_generic_k(const double* __restrict__ q, uint32_t nq,
                                                    const double* __restrict__ db, uint32_t ndb, int dim,
                                                    uint32_t db_per_split, double* __restrict__ best_d,
                                                    uint32_t* __restrict__ best_i) {
    extern __shared__ double qs[];  // dim x 64, column per lane
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const uint32_t ii = i < nq ? i : (nq - 1);
    for (int k = 0; k < dim; ++k) qs[k * 64 + threadIdx.x] = q[(size_t)ii * dim + k];
    const uint32_t j0 = blockIdx.y * db_per_split;
    const uint32_t j1 = min(ndb, j0 + db_per_split);
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t j = j0; j < j1; ++j) {
        const double* __restrict__ d = db + (size_t)j * dim;
        double acc = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double df = qs[k * 64 + threadIdx.x] - d[k];
            acc += df * df;
        }
        if (acc < bd) {
            bd = acc;
            bi = j;
        }
    }
    if (i < nq) {
        best_d[(size_t)blockIdx.y * nq + i] = bd;
        best_i[(size_t)blockIdx.y * nq + i] = bi;
    }
}
__global__ void nn_merge_k(const double* __restrict__ best_d, const uint32_t* __restrict__ best_i,
                           uint32_t nq, uint32_t splits, uint32_t* __restrict__ nn) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nq) return;
    double bd = INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t s = 0; s < splits; ++s) {  // ascending index ranges: strict < keeps the lowest index
        const double d = best_d[(size_t)s * nq + i];
        if (d < bd) {
            bd = d;
            bi = best_i[(size_t)s * nq + i];
        }
    }
    nn[i] = bi;
}

void launch_nn(const double* q, uint32_t nq, const double* db, uint32_t ndb, int dim, uint32_t splits,
               double* best_d, uint32_t* best_i, uint32_t* nn, hipStream_t s) {
    if (!nq) return;
    const uint32_t per = (ndb + splits - 1) / splits;
    if (dim == 33) {
        nn_k<33><<<dim3((nq + 255) / 256, splits), 256, 0, s>>>(q, nq, db, ndb, per, best_d, best_i);
    } else if (dim == 3) {
        nn_k<3><<<dim3((nq + 255) / 256, splits), 256, 0, s>>>(q, nq, db, ndb, per, best_d, best_i);
    } else {
        nn_generic_k<<<dim3((nq + 63) / 64, splits), 64, sizeof(double) * 64 * dim, s>>>(q, nq, db, ndb, dim, per,
                                                                                      best_d, best_i);
    }
    nn_merge_k<<<(nq + 255) / 256, 256, 0, s>>>(best_d, best_i, nq, splits, nn);
}

}  // namespace m3d
