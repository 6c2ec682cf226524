// m3d_fit.cpp -- the RANSAC loop of the host driver behind the C ABI (include/misc3d_amd.h).
//
// Mirrors the control flow of misc3d::common::RANSAC (include/misc3d/common/ransac.h:455-664) and
// segmentation::SegmentPlaneIterative (src/iterative_plane_segmentation.cpp:8-39), with the data-
// parallel loops moved into the kernels of m3d_kernels.hip:
//
//   reference (sequential, OMP_NUM_THREADS=1 semantics)          here
//   ---------------------------------------------------          ------------------------------------
//   for i < max_iteration:                                       chunks of hypotheses, geometric sizes
//     sample = sampler(m)            utils.h:81-97               host std::mt19937 -> sample table -> HBM
//     MinimalFit(sample)             ransac.h:582                minimal_fit_k   (1 thread / hypothesis)
//     EvaluateModel(all points)      ransac.h:588,626-654        score_k + reduce_partials_k (counts only)
//     critical: best update, k       ransac.h:592-613            replay_chunk() on the host, in index order
//   RefineModel                      ransac.h:534-549            compact_*_k + sum_*_k + closed form
//
// `inlier_rmse` (error / sqrt(n), ransac.h:650) only ever decides between hypotheses of EQUAL
// fitness (ransac.h:595-596) and is never returned, so the serial error sum is evaluated on demand,
// in point order, only when such a tie occurs (exact_error()).
//
// There is no CPU fallback: every entry point needs a HIP device.
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

namespace m3d {

// ------------------------------------------------------------------------------------------------
// sequential replay of ransac.h:573-575, 592-613
// ------------------------------------------------------------------------------------------------
// size_t(double) as gcc/x86-64 evaluates it (cvttsd2si); the C++ conversion is undefined for the
// NaN / negative / huge values the adaptive bound can take (log(0), 0/0).
static uint64_t double_to_size_t_x86(double v) {
    const double two63 = 9223372036854775808.0;
    if (v >= two63) {
        const double w = v - two63;
        if (!(w < two63)) return 0;  // indefinite (0x8000...) xor sign bit
        return static_cast<uint64_t>(static_cast<int64_t>(w)) ^ 0x8000000000000000ull;
    }
    if (!(v > -two63)) return 0x8000000000000000ull;
    return static_cast<uint64_t>(static_cast<int64_t>(v));
}

// tie(i, cnt, &trial_rmse, &trial_rmse_known) decides `inlier_rmse < inlier_rmse_` (ransac.h:596) for a
// trial whose fitness EQUALS the best one; it may fill st->best_rmse when it had to evaluate it.
// valid == nullptr: counts[k] carries MinimalFit's return in bit 31 (the form the scoring kernels ship).
template <class TieFn, class BestFn>
static void replay_range(m3d_replay_state* st, size_t n_points, int kind, size_t max_iteration,
                         double probability, size_t begin, size_t end, const uint8_t* valid,
                         const uint32_t* counts, TieFn tie, BestFn on_best) {
    const bool packed = valid == nullptr;
    const int m = minimal_sample(kind);
    for (size_t i = begin; i < end; ++i) {
        if (st->stopped) return;
        if (st->count > st->current_iteration) {  // ransac.h:573-575
            st->stopped = 1;
            return;
        }
        st->iterations = i + 1;
        const size_t k = i - begin;
        if (!(packed ? (counts[k] >> 31) != 0u : valid[k] != 0)) continue;  // ransac.h:584-586 (no count++)
        const uint32_t cnt = packed ? (counts[k] & 0x7FFFFFFFu) : counts[k];
        if (cnt < st->best_count) {   // fitness = cnt / N is strictly monotone in cnt (N < 2^31): cannot be better or tie
            st->count++;
            continue;
        }
        // EvaluateModel's tail, ransac.h:644-651
        const double fitness = cnt == 0 ? 0.0 : (double)cnt / (double)n_points;
        bool better = fitness > st->best_fitness;
        double trial_rmse = 0.0;
        bool trial_rmse_known = false;
        if (!better && fitness == st->best_fitness) better = tie(i, cnt, &trial_rmse, &trial_rmse_known);
        if (better) {
            st->best_fitness = fitness;
            st->best_rmse = trial_rmse;
            st->best_rmse_known = trial_rmse_known ? 1 : 0;
            st->best_index = (int64_t)i;
            st->best_count = cnt;
            on_best(i);
            if (st->best_fitness < 1.0) {  // ransac.h:601-606
                const double kk =
                    std::log(1 - probability) / std::log(1 - std::pow(st->best_fitness, (double)m));
                const double lim = (double)max_iteration;
                st->current_iteration = double_to_size_t_x86(lim < kk ? lim : kk);
            } else {
                st->current_iteration = 0;  // ransac.h:609
            }
        }
        st->count++;  // ransac.h:612
    }
}

// The same replay when probability == 1 (packed records).  Then only fitness == 1 can stop the loop
// (ransac.h:601-609: the adaptive bound is min(+inf, max_iteration)), so a chunk changes the state only through
// the hypotheses that reach its HIGHEST inlier count: one vectorisable pass finds that count and the number of valid
// records, a second one visits the few records that have it, in index order, with the sequential rule.  Everything
// observable is as replay_range leaves it (best index / count / fitness / rmse bookkeeping, count, iterations); only
// the `ties` statistic no longer includes ties among hypotheses that a later one overtakes anyway.  A chunk in which
// some hypothesis reaches fitness 1 takes the sequential loop.  (10 000 records: 12 -> 2 us; a sharded fit replays
// world x as many on every rank.)
template <class TieFn, class BestFn>
static void replay_range_exhaustive(m3d_replay_state* st, size_t n_points, int kind, size_t max_iteration,
                                    size_t begin, size_t end, const uint32_t* rec, TieFn tie, BestFn on_best) {
    if (st->stopped || end <= begin) return;
    const size_t n = end - begin;
    uint32_t mx = 0, nvalid = 0;
    for (size_t k = 0; k < n; ++k) {
        const uint32_t r = rec[k], v = r >> 31, c = r & 0x7FFFFFFFu;
        nvalid += v;
        const uint32_t cv = v ? c : 0u;
        mx = cv > mx ? cv : mx;
    }
    if (mx >= n_points || st->count > st->current_iteration || st->count + nvalid > st->current_iteration) {
        replay_range(st, n_points, kind, max_iteration, 1.0, begin, end, nullptr, rec, tie, on_best);   // stops inside
        return;
    }
    if (mx > 0 && mx >= st->best_count) {
        const uint32_t want = mx | 0x80000000u;
        for (size_t k = 0; k < n; ++k) {
            if (rec[k] != want) continue;
            const size_t i = begin + k;
            const double fitness = (double)mx / (double)n_points;
            bool better = fitness > st->best_fitness;
            double trial_rmse = 0.0;
            bool trial_rmse_known = false;
            if (!better && fitness == st->best_fitness) better = tie(i, mx, &trial_rmse, &trial_rmse_known);
            if (better) {
                st->best_fitness = fitness;
                st->best_rmse = trial_rmse;
                st->best_rmse_known = trial_rmse_known ? 1 : 0;
                st->best_index = (int64_t)i;
                st->best_count = mx;
                on_best(i);
                // ransac.h:601-606 with probability 1: log(0) / log(1 - w^m) = +inf, min(+inf, max_iteration)
                st->current_iteration = (uint64_t)max_iteration;
            }
        }
    }
    st->count += nvalid;
    st->iterations = end;
}

// ------------------------------------------------------------------------------------------------
// chunk machinery
// ------------------------------------------------------------------------------------------------
struct SampleSource {
    // either a seeded sampler (utils.h:71-97) or a caller-provided table
    Mt19937Mod rng;       // std::mt19937 stream, a block of 624 draws at a time (m3d_mt19937.hpp)
    size_t n_points = 0;
    const uint32_t* table = nullptr;  // H x m, absolute hypothesis index
    int m = 3;
    void seed(uint64_t s) { rng.seed((uint32_t)(s & 0xffffffffull)); }   // std::mt19937(seed): seed mod 2^32
    void fill(size_t begin, size_t end, uint32_t* out) {
        if (table) {
            std::memcpy(out, table + begin * m, sizeof(uint32_t) * (end - begin) * m);
            return;
        }
        // `rng_() % size_` (utils.h:89) with a 32-bit generator output and size < 2^32
        if (rng.d != (uint32_t)n_points) rng.set_modulus((uint32_t)n_points);
        const size_t n = end - begin;
        if (m == 3) rng.fill<3>(out, n);
        else if (m == 2) rng.fill<2>(out, n);
        else rng.fill<4>(out, n);
    }
};

static uint32_t pick_splits(uint32_t n_tiles, uint32_t h_pad) {
    // enough workgroups to fill 256 CUs x 8 resident workgroups a couple of times over
    const uint32_t groups = h_pad / 64;
    const uint32_t target_wgs = (uint32_t)config().dense_workgroups;
    const uint32_t want = std::max<uint32_t>(1, (target_wgs + n_tiles - 1) / n_tiles);
    return std::min(want, groups);
}

// m3d_config.dense_scoring selects the dense scoring kernel (score_k: every tile x every hypothesis) instead of the
// culled path (cull_mask_k + score_mask_k); both produce identical counts (tests/test_gpu_parity.py runs both).
// A cloud created WITHOUT the Hilbert-sorted copy (one-shot fits of few hypotheses: one_shot_fit) can only be scored
// densely: the fit sets this flag of its own thread for its duration.
static thread_local int t_dense_fit = 0;
static bool use_dense_scoring() { return t_dense_fit != 0 || config().dense_scoring != 0; }

// Hypotheses at the head of a probability-1 fit that are counted first, for the incumbent that prunes the rest.
static uint32_t lead_size() { return (uint32_t)config().lead_hypotheses; }

static size_t chunk_cap_for(const CloudView& v, const SortedView& sv, size_t times = 1) {
    // keep the per-chunk scratch below 1 GiB (dense: u32 partial count per (tile, hypothesis);
    // culled: one bit per (tile, hypothesis))
    if (!use_dense_scoring()) {
        // hypotheses per chunk (m3d_config.chunk_cap).  A chunk boundary costs ~35 us (sum_replicas_k, keep_mask_k, their launch
        // boundaries), a chunk's sample table must be drawn while the chunk before it is on the GPU (a sphere's four draws per
        // hypothesis: 1.9 ns against ~7.5 ns of scoring), and the first chunk of a long fit is short (2048).  C3's 50 000
        // hypotheses: 16384 (four chunks) cylinder 0.998 / sphere 0.665 ms, 24576 (three) 0.985 / 0.659, 49152 (two) 0.944 / 0.689 --
        // the sphere's second chunk then waits for its samples.
        // ... and the slot's mask scratch (one bit per (tile, hypothesis), twice: two slots) stays below 1 GiB per slot whatever
        // the cloud's size: 24576 hypotheses reach that at 350 000 tiles (180 M points) -- ADVICE r4
        const size_t by_masks = sv.n_tiles ? std::max<size_t>(((size_t)1 << 30) / ((size_t)sv.n_tiles * 8), 16) * 64 : ~(size_t)0;
        // times = 2: a CYLINDER fit with a forced iteration count (m3d_fit_on).  A window of cylinders costs ~120 us before its first
        // pair is counted (keep_mask_k, plane_bound_k<2> 60 us, three phases with their tails and re-prunings, sum_replicas_k) where a
        // window of planes or spheres costs ~35, and the incumbent of the 2048-hypothesis first chunk already prunes as well as a
        // whole window's would: C3's 50 000 cylinders in one window of 47 952 instead of two of 23 976 evaluate the same 1.93 M pairs,
        // the fit 0.861 -> 0.791 ms (round 6, profiles/r06_c3_chunk_cap.txt; the sphere loses: 0.66 -> 0.73, its second chunk's
        // box tests no longer run under the first one's scoring)
        return std::min<size_t>(std::min<size_t>((size_t)config().chunk_cap * times, 262144), by_masks);
    }
    const uint32_t rows = std::max<uint32_t>(1, v.n_pad / kScoreTile);
    const size_t cap = std::min<size_t>(16384, ((size_t)1 << 28) / rows / 64 * 64);
    return std::max<size_t>(cap, 64);
}

// Where the planes' histogram bound is worth its launch: plane_bound_k costs 12-25 us per window whatever it prunes, and it halves
// a scoring launch -- worth it when that launch is long.  Measured on C2-shaped clouds, resident, ms per fit without / with:
//   points x hypotheses   50 k x 3000 0.103 / 0.135   150 k x 10 000 0.166 / 0.178   150 k x 40 000 0.325 / 0.377   400 k x 10 000 0.176 / 0.175
//   400 k x 40 000 0.413 / 0.386   1 M x 3000 0.207 / 0.213   1 M x 10 000 0.300 / 0.274   1 M x 40 000 0.654 / 0.606
//   4 M x 3000 0.530 / 0.569   4 M x 10 000 0.864 / 0.809   4 M x 40 000 1.876 / 1.671
// i.e. windows of 8192 hypotheses and more on tiles x hypotheses >= 1.5e7.
static bool bound_pays(uint32_t n_tiles, size_t window_hypotheses) {
    return window_hypotheses >= 8192 && (double)n_tiles * (double)window_hypotheses >= 1.5e7;
}

// pre_stream runs the head of a fit's later chunks (issue_chunk, `pre`) beside the main stream: whatever the main stream holds
// when the fit starts -- a removal's compaction of this very cloud, another fit's tail -- must be behind those kernels too.
static bool prestream_enabled() { return config().prestream != 0; }   // (0: everything on the main stream)
static int pre_stream_gate(DeviceCtx* ctx) {
    if (!pre_stream_of(ctx) || !ctx->ev_pre_gate) return M3D_OK;   // (created here, by the lane's first fit of several chunks)
    HIPCHK(hipEventRecord(ctx->ev_pre_gate, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->pre_stream, ctx->ev_pre_gate, 0));
    return M3D_OK;
}

// plane_bound_k's scratch: its tickets -- kBoundTicketWords words at the FRONT of the block, which the kernel leaves zero between
// launches (cleared here when the block is new; at a fixed place: behind a list whose length changes from fit to fit they would
// land on old ids) -- and behind them the survivor list (h_pad + 64 ids)
constexpr size_t kBoundTicketWords = 8192;   // 1 + hypotheses of a window / 64 (M3D_CHUNK_CAP <= 262 144: 4097)
static int reserve_survivor_scratch(DeviceCtx* ctx, uint32_t h_pad) {
    const size_t words = kBoundTicketWords + (size_t)h_pad + 64;
    if ((size_t)h_pad / 64 + 2 > kBoundTicketWords) return fail(M3D_ERR_INTERNAL, "window too large for the bound's tickets");
    if (ctx->surv_list.cap < sizeof(uint32_t) * words) {
        RESERVE(ctx->surv_list, sizeof(uint32_t) * (kBoundTicketWords + 2 * ((size_t)h_pad + 64)));
        HIPCHK(hipMemsetAsync(ctx->surv_list.p, 0, ctx->surv_list.cap, ctx->stream));
    }
    return M3D_OK;
}
static uint32_t* bound_tickets(DeviceCtx* ctx) { return ctx->surv_list.as<uint32_t>(); }
static uint32_t* bound_list(DeviceCtx* ctx) { return ctx->surv_list.as<uint32_t>() + kBoundTicketWords; }

// Sharded fits (comm != null): the chunk is the SAME window of the one hypothesis stream on every rank
// -- sample table, MinimalFit and parameter records for all of it (a thread per hypothesis: microseconds) -- but
// the box tests and the scoring cover only this rank's slice of `sl_pad` hypotheses (+ the window's leading
// hypotheses, for the pruning incumbent; their owner is rank 0).  The slices' records are then all-gathered in place
// in s.counts, so that everything after it (pick_best_k, the replay, the tie rule reading s.params) sees the whole
// chunk exactly as the one-GPU path does.
static int issue_chunk(DeviceCtx* ctx, ChunkSlot& s, const CloudView& v, const SortedView& sv, int kind,
                       double thr, size_t begin, size_t end, SampleSource& src, double* ms_sample,
                       bool prune = false, uint32_t lead = 0, bool new_fit = false /* clears the running best count */,
                       bool device_records = false /* culled path: keep a device copy of the records in s.counts */,
                       m3d_comm* comm = nullptr,
                       bool caller_ships_records = false /* the caller queues pick_best_k behind the chunk and records s.done
                                                            behind THAT (an event between two kernels costs a ~5 us
                                                            gap on this stream; behind pick_best_k nothing follows at once) */,
                       const PickFinal* pick_final = nullptr /* one GPU: the chunk's last kernel takes pick_best_k's decision
                                                                and stores the completion word the host polls (no event) */,
                       bool nothing_to_prune = false /* the fit's ONLY chunk, issued without a lead pass: no incumbent will
                                                        ever exist while it runs */,
                       bool pre = false /* a later chunk of a one-GPU fit: MinimalFit and the box tests go to ctx->pre_stream and
                                           run under the scoring launches of the chunk before (the main stream waits for
                                           s.pre_done in front of the keep masks) */,
                       const double* move_best_from = nullptr /* the fit's best minimal model lives in THIS slot's params, which
                                                                 the chunk is about to overwrite: moved to ctx->best_params
                                                                 first, on the stream MinimalFit will run on */) {
    const int m = minimal_sample(kind);
    const uint32_t count = (uint32_t)(end - begin);
    const bool dense = use_dense_scoring();
    // where the chunk's head runs -- decided ONCE, here, in front of everything that depends on it (the record
    // move used to repeat a part of this condition in the caller)
    pre = pre && !dense && prune && !new_fit && !comm && lead == 0 && !ctx->poison_pending && ctx->pre_stream && s.pre_done;
    hipStream_t st_pre = pre ? ctx->pre_stream : ctx->stream;
    if (move_best_from)   // (before the slot's buffers are touched: a RESERVE below may hand the old block back)
        HIPCHK(hipMemcpyAsync(ctx->best_params.p, move_best_from, sizeof(double) * kModelStride, hipMemcpyDeviceToDevice, st_pre));
    const bool timing = config().kernel_timing != 0;
    if (comm && dense) return fail(M3D_ERR_INVALID_ARG, "sharded fits use the culled scoring path (m3d_config.dense_scoring = 0)");
    const uint32_t world = comm ? (uint32_t)comm->world : 1u, rank = comm ? (uint32_t)comm->rank : 0u;
    const uint32_t sl_pad = round_up((count + world - 1) / world, 64);   // hypotheses per rank
    const uint32_t h_pad = comm ? sl_pad * world : round_up(count, 64);
    const uint32_t n_tiles = v.n_pad / kScoreTile;
    if (comm) device_records = true;
    s.begin = begin;
    s.end = end;
    s.h_pad = h_pad;
    RESERVE(s.samples, sizeof(uint32_t) * (size_t)count * m);
    RESERVE(s.score, sizeof(double) * kModelStride * ((size_t)h_pad + 1));
    RESERVE(s.params, sizeof(double) * kModelStride * ((size_t)h_pad + 1));
    RESERVE(s.valid, (size_t)h_pad + 1);
    Cull32Out c32;
    const bool cull32 = !dense && config().cull_fp32 != 0 && sv.radius < 1e18;
    if (cull32) {
        RESERVE(s.cull32, sizeof(float) * 24 * ((size_t)h_pad / 2 + 1));
        c32.out = s.cull32.as<float>();
        for (int k = 0; k < 3; ++k) c32.origin[k] = sv.origin[k];
        c32.radius = sv.radius;
    }
    if (dense || device_records) RESERVE(s.counts, sizeof(uint32_t) * ((size_t)h_pad + 1));   // (culled path: records go straight to h_counts)
    uint32_t* rec_dev = (!dense && device_records) ? s.counts.as<uint32_t>() : nullptr;
    RESERVE(s.h_samples, sizeof(uint32_t) * (size_t)count * m);
    RESERVE(s.h_counts, sizeof(uint32_t) * ((size_t)h_pad + 4));   // + the launch's pair counters behind the records
    RESERVE(s.h_valid, (size_t)h_pad + 1);
    if (dense) {
        RESERVE(ctx->partial, sizeof(uint32_t) * (size_t)n_tiles * h_pad);
    } else {
        RESERVE(s.masks, sizeof(uint64_t) * (size_t)std::max<uint32_t>(sv.n_tiles, 1) * (h_pad / 64));
        RESERVE(ctx->keep, sizeof(uint64_t) * (size_t)(h_pad / 64));
        RESERVE(ctx->counts_rep, sizeof(uint32_t) * ((size_t)kCountReplicas * h_pad + kPairReplicas));
    }
    const double t0 = now_ms();
    src.fill(begin, end, s.h_samples.as<uint32_t>());
    if (ms_sample) *ms_sample += now_ms() - t0;
    // the sample table is read by minimal_fit_k straight from the slot's page-locked host array (device-visible): 12 bytes
    // per hypothesis over the host link inside the kernel instead of a copy command in front of it
    if (!dense && prune) RESERVE(s.ub, sizeof(uint32_t) * 3 * (size_t)h_pad);   // ub[h_pad], then the phase counters ubp[h_pad], then the cylinders' histogram bound sums
    // (decided here because minimal_fit_k prepares the lead pass of a NEW fit itself: see LeadPrep)
    const bool use_lead = !dense && prune && lead >= 64 && lead % 64 == 0 && lead + 64 <= count && (!comm || sl_pad >= lead + 64);
    const bool own_real_ = !comm || (size_t)rank * sl_pad < count;
    LeadPrep lp;
    const bool lead_prepared = use_lead && new_fit && own_real_ && lead <= h_pad && (uint32_t)kPairReplicas <= h_pad;
    // no lead pass and nothing to prune against (a new fit's only chunk on one GPU, DeviceCtx::no_prune_hint): the same
    // set-up for ALL groups -- keep everything, clear every counter replica -- and no keep_mask_k launch below
    const bool all_prepared = nothing_to_prune && !use_lead && lead == 0 && !dense && prune && new_fit && !comm &&
                              (uint32_t)kPairReplicas <= h_pad;
    if (lead_prepared || all_prepared) {
        lp.counts_rep = ctx->counts_rep.as<uint32_t>();
        lp.keep = ctx->keep.as<unsigned long long>();
        lp.rep_stride = h_pad;
        lp.n_rep = kCountReplicas;
        lp.n_lead = all_prepared ? h_pad : lead;
        lp.n_pair = kPairReplicas;
    }
    const bool fit_launched =
        launch_minimal_fit(kind, v, s.h_samples.as<uint32_t>(), count, h_pad + 1, thr, s.score.as<double>(),
                       s.params.as<double>(), s.valid.as<uint8_t>(), st_pre,
                       (!dense && prune) ? s.ub.as<uint32_t>() : nullptr,    // clears ub[0 .. h_pad) on the way
                       new_fit ? ctx->best_count.as<uint32_t>() : nullptr, (lead_prepared || all_prepared) ? &lp : nullptr, sv.max_abs,
                       cull32 ? &c32 : nullptr, (ctx->poison_pending && kind == M3D_PLANE) ? &ctx->pending_poison : nullptr,
                       (!dense && prune) ? s.ub.as<uint32_t>() + h_pad : nullptr);
    if (fit_launched && ctx->poison_pending && kind == M3D_PLANE) {   // (the previous round's tombstone pass went with it)
        ctx->poison_pending = false;
        if (ctx->poison_expected_at) *ctx->poison_expected_at += ctx->poison_pending_count;
    }
    if (dense)
        HIPCHK(hipMemsetAsync(s.counts.p, 0, sizeof(uint32_t) * (size_t)h_pad, ctx->stream));
    // (culled path: keep_mask_k clears the counter replicas on its way)
    s.lead_groups = 0;
    s.lead_fused = false;
    s.scored = false;
    s.poll_seq = 0;
    s.host_has_records = true;
    s.done_on_copy_stream = false;
    uint32_t* h_pairs = s.h_counts.as<uint32_t>() + h_pad;   // pinned, device-visible
    if (dense) {
        HIPCHK(hipEventRecord(s.k0, ctx->stream));
        launch_score(kind, v, s.score.as<double>(), h_pad, pick_splits(n_tiles, h_pad),
                     ctx->partial.as<uint32_t>(), ctx->stream);
        HIPCHK(hipEventRecord(s.k1, ctx->stream));
        launch_reduce_partials(ctx->partial.as<uint32_t>(), n_tiles, h_pad, s.counts.as<uint32_t>(),
                               ctx->stream);
        s.scored = true;
    } else {
        // prune (fits only): hypotheses that cannot reach the best count of EARLIER hypotheses are masked
        // out (keep_mask_k); ctx->best_count is the device-side running maximum of the fit
        const uint32_t n_groups = h_pad / 64;
        // this rank's groups [g0, g1) of the chunk
        const uint32_t g0 = comm ? rank * (sl_pad / 64) : 0u, g1 = comm ? g0 + sl_pad / 64 : n_groups;
        const bool own_real = (size_t)g0 * 64 < count;   // (a short last window can leave the highest ranks without work)
        uint32_t* ub = prune ? s.ub.as<uint32_t>() : nullptr;
        // phased scoring (launch_score_phased): an incumbent exists, the fp32 box tests run (they count the touched tiles per
        // phase), no tombstones
        uint32_t* ubp = (prune && (use_lead || !new_fit) && c32.out && !sv.has_dead && score_phases_for(kind) != 0) ? ub + h_pad : nullptr;   // (an incumbent exists: this chunk's lead pass, or earlier chunks / windows -- sharded fits included: every rank prunes its slice against the same incumbent)
        auto* masks = s.masks.as<unsigned long long>();
        auto* keep = ctx->keep.as<unsigned long long>();
        uint32_t* pair_rep = ctx->counts_rep.as<uint32_t>() + (size_t)kCountReplicas * h_pad;
        uint32_t* bc = prune ? ctx->best_count.as<uint32_t>() : nullptr;
        // sharded: the host gets the GATHERED records -- unless the communicator has ONE rank: its slice is the window, the
        // gather the identity, and the records go to the host the way the one-GPU path sends them (written by the folding
        // kernels themselves: no copy command, no detour over the copy stream)
        const bool solo = comm && comm->world == 1 && comm->transport == m3d_comm::kRccl;   // (a caller-supplied all-gather is always called)
        uint32_t* rec_host = (comm && !solo) ? nullptr : s.h_counts.as<uint32_t>();
        PickFinal pfin;
        if (pick_final) {
            pfin = *pick_final;
            pfin.params = s.params.as<double>();
        }
        // lead > 0 (a fit's first chunk): the first `lead` hypotheses are counted on their own, and their best
        // count then prunes the rest of the SAME chunk -- what a separate small first chunk did, without its
        // own sample copy, MinimalFit and box-test launches.  The records are complete after the second pass.
        const uint32_t ga = use_lead ? lead / 64 : 0u;
        if (own_real) {
            // fp32 box tests and fp32 screen on: the box tests of the chunk (this rank's slice) and the counting of its leading groups
            // share ONE launch (cull_lead_k; the lead pass runs its own box tests) -- two latency-bound launches less the
            // time of one.  It is not among the timed scoring launches (s.lead_fused: m3d_stats.pairs_timed).
            s.lead_fused = false;
            if (ga && g0 == 0 && c32.out) {   // (sharded fits: the rank that owns the leading groups)
                if (!lead_prepared)   // (a new fit: minimal_fit_k has done it)
                    launch_keep_mask(ub, bc, ga, keep, ctx->stream, ctx->counts_rep.as<uint32_t>(), h_pad, 0);
                s.lead_fused = launch_cull_lead(kind, sv, s.score.as<double>(), c32.out, masks, keep, n_groups, ga,
                                                ctx->counts_rep.as<uint32_t>(), h_pad, pair_rep, ub, g1, ctx->stream, ubp);
            }
            // nothing to prune and every group prepared by minimal_fit_k: the box tests run inside the scoring launch
            bool scored_with_own_tests = false;
            // (Round 6: OFF.  Fusing the box tests into the scoring launch paid in round 3, when cull_tiles32_k was 9 us of dependent
            //  scalar loads; since round 5's staged records the two launches are the faster way round again -- C5's 165 clutter rounds
            //  15.1 against 16.1 ms.  And the other direction is closed as well: ALL of a tile's groups in one scoring workgroup
            //   -- with its own box tests 22.0 ms, behind precomputed masks 21.6 ms: 1528 one-wave workgroups leave
            //  every wave's chain of dependent loads exposed, 12 228 of them hide it.  profiles/r06_c5_tail_rounds.txt)
            constexpr bool kScoreWithOwnTests = false;
            if (kScoreWithOwnTests && all_prepared && c32.out)
                scored_with_own_tests = launch_score_own_tests(kind, sv, s.score.as<double>(), c32.out, masks, keep, n_groups, g1,
                                                               ctx->counts_rep.as<uint32_t>(), h_pad, pair_rep, ctx->stream,
                                                               timing ? s.k0 : nullptr, timing ? s.k1 : nullptr);
            // planes with an incumbent and tile frames: a histogram upper bound per touched (tile, hypothesis) pair replaces
            // "512 per touched tile" in the keep rule (m3d_bound.hip).  The keep kernels below hand plane_bound_k the kept
            // hypotheses as a list (its length: word 6 of best_count); ubsum is the slot's phase-counter array, zeroed by
            // minimal_fit_k and unused when the scoring is not phased.
            // Spheres and cylinders (round 5): the same bound with the shell taken as a slab per tile (cyl_pair_ub); cylinders in front of
            // the phased scoring -- whose phase counters sit where the bound sums go otherwise, so theirs get the block behind them.
            // (Spheres: C3's fit loses a third of its (tile, hypothesis) pairs and gains nothing -- 0.77 ms with and without, the bound's
            // launch costs what it saves; theirs runs when the bound is forced, m3d_config.plane_bound = 2, i.e. in the tests.)
            const bool bound_on = (kind == M3D_CYLINDER || !ubp) && prune && bc && sv.frames && config().plane_bound != 0 &&
                                  (config().plane_bound == 2 || (kind != M3D_SPHERE && bound_pays(sv.n_tiles, comm ? sl_pad : count))) &&
                                  (use_lead || !new_fit) && !scored_with_own_tests && std::max(g0, ga) < g1;
            uint32_t* const ubsum = ubp ? ub + 2 * (size_t)h_pad : ub + h_pad;
            if (bound_on) {
                const int src = reserve_survivor_scratch(ctx, h_pad);
                if (src != M3D_OK) return src;
                if (ubp) HIPCHK(hipMemsetAsync(ubsum, 0, sizeof(uint32_t) * h_pad, ctx->stream));   // (the block behind ub is cleared by minimal_fit_k)
            }
            uint32_t* const surv_count = bound_on ? bc + 6 : nullptr;
            uint32_t* const surv = bound_on ? bound_list(ctx) : nullptr;
            // cylinders in front of the bound: the box tests also leave a word per hypothesis and 64 tiles (cull_hyp32_k), which
            // plane_bound_k reads instead of repeating the tests
            unsigned long long* touched = nullptr;
            bool touched_written = false;
            if (bound_on && kind == M3D_CYLINDER && c32.out && !s.lead_fused && !scored_with_own_tests && !(ga && g0 >= ga) && g1 - g0 >= 128u) {
                RESERVE(s.touched, sizeof(uint64_t) * (size_t)((sv.n_tiles + 63u) / 64u) * h_pad);
                touched = s.touched.as<unsigned long long>();
            }
            if (!s.lead_fused && !scored_with_own_tests) {
                if (ga && g0 >= ga)   // (rank > 0: the lead is somebody else's slice)
                    launch_cull_mask(kind, sv, s.score.as<double>(), s.valid.as<uint8_t>(), count, n_groups, masks, ub,
                                     ctx->stream, /*ub_is_zero=*/true, 0, ga, c32.out);
                launch_cull_mask(kind, sv, s.score.as<double>(), s.valid.as<uint8_t>(), count, n_groups, masks, ub, st_pre,
                                 /*ub_is_zero=*/true, g0, g1, c32.out, ubp, touched, h_pad, &touched_written);
            }
            if (pre) {   // everything below needs the records, the masks and ub: the main stream picks up here
                HIPCHK(hipEventRecord(s.pre_done, ctx->pre_stream));
                HIPCHK(hipStreamWaitEvent(ctx->stream, s.pre_done, 0));
            }
            uint32_t g_lo = g0;
            if (ga) {
                s.lead_groups = ga;
                g_lo = std::max(g0, ga);
                if (!s.lead_fused) {
                    if (!lead_prepared)   // (a new fit: minimal_fit_k has done it)
                        launch_keep_mask(ub, bc, ga, keep, ctx->stream, ctx->counts_rep.as<uint32_t>(), h_pad, 0);
                    launch_score_mask(kind, sv, s.score.as<double>(), masks, keep, n_groups, ctx->counts_rep.as<uint32_t>(),
                                      h_pad, pair_rep, ctx->stream, 0, ga, timing ? s.k2 : nullptr, timing ? s.k3 : nullptr);
                }
                // fold of the lead's counters + keep masks of the rest of this rank's groups: one launch
                launch_lead_fold_keep(ctx->counts_rep.as<uint32_t>(), h_pad, lead, s.valid.as<uint8_t>(), count,
                                      g0 == 0 ? rec_host : nullptr, bc, ub, keep, g1 - g_lo, ctx->stream,
                                      g0 == 0 ? rec_dev : nullptr, g_lo, pick_final ? pick_final->key : nullptr,
                                      pick_final ? pick_final->key2 : nullptr, surv_count, surv);
            } else if (!all_prepared) {   // (all_prepared: minimal_fit_k has done it)
                launch_keep_mask(ub, bc, g1 - g0, keep, ctx->stream, ctx->counts_rep.as<uint32_t>(), h_pad, g0, surv_count, surv);
            }
            if (bound_on)
                launch_plane_bound(kind, sv, s.score.as<double>(), masks, keep, n_groups, g_lo, g1, ubsum, bc, surv_count, surv,
                                   bound_tickets(ctx), c32.out, ctx->stream, config().plane_bound == 2,
                                   touched_written && g_lo == g0 ? touched : nullptr, h_pad);
            bool phased = false;
            if (!scored_with_own_tests && ubp)
                phased = launch_score_phased(kind, sv, s.score.as<double>(), masks, keep, n_groups, ctx->counts_rep.as<uint32_t>(), h_pad,
                                             pair_rep, ub, ubp, bc, ctx->stream, g_lo, g1, timing ? s.k0 : nullptr, timing ? s.k1 : nullptr);
            if (!scored_with_own_tests && !phased)
                launch_score_mask(kind, sv, s.score.as<double>(), masks, keep, n_groups, ctx->counts_rep.as<uint32_t>(), h_pad,
                                  pair_rep, ctx->stream, g_lo, g1, timing ? s.k0 : nullptr, timing ? s.k1 : nullptr, bound_on);
            // one launch: fold the counter replicas, tag MinimalFit's return into bit 31, update the incumbent
            // ... and (one GPU) write the records straight into the slot's pinned host array (device-visible): no copy
            // command behind the kernel (a 39 KB D2H copy started ~20 us after the kernel that fed it)
            launch_sum_replicas(ctx->counts_rep.as<uint32_t>(), h_pad, g1 * 64u, rec_host, pair_rep, h_pairs,
                                s.valid.as<uint8_t>(), count, bc, ctx->stream, g_lo * 64u, rec_dev, pick_final ? &pfin : nullptr,
                                phased ? keep : nullptr);
            if (pick_final) s.poll_seq = pick_final->seq;
            s.scored = timing;
        } else {
            HIPCHK(hipMemsetAsync(rec_dev + (size_t)g0 * 64, 0, sizeof(uint32_t) * sl_pad, ctx->stream));
            h_pairs[0] = h_pairs[1] = h_pairs[2] = 0;
        }
        if (solo) {
            comm->collectives++;   // (the window's exchange, degenerate)
            s.host_has_records = true;
        } else if (comm) {
            // the one exchange of the window: every rank's slice of records, in place (RCCL: ncclAllGather on this
            // stream, nothing on the host; host transports wait for the stream and leave the records in h_counts)
            int host_has_all = 0;
            const int rc = comm->allgather_u32_device(rec_dev, sl_pad, ctx->stream, s.h_counts.as<uint32_t>(), &host_has_all);
            if (rc != M3D_OK) return rc;
            s.host_has_records = host_has_all != 0;
            if (!host_has_all && caller_ships_records) {
                // the gathered window (world x slice records: hundreds of KB at 8 ranks) goes to the host through the copy
                // engine, on the copy stream, while pick_best_k and the early compaction run on the main stream; s.done
                // then sits on the copy stream.  (pick_best_k's one workgroup would need 40 us for 80 000 records.)
                HIPCHK(hipEventRecord(ctx->ev_compact, ctx->stream));
                HIPCHK(hipStreamWaitEvent(copy_stream_of(ctx), ctx->ev_compact, 0));
                HIPCHK(hipMemcpyAsync(s.h_counts.p, rec_dev, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost,
                                      copy_stream_of(ctx)));
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(s.done, copy_stream_of(ctx)));
                s.host_has_records = true;
                s.done_on_copy_stream = true;
                return M3D_OK;
            }
            if (!host_has_all && !caller_ships_records) {
                HIPCHK(hipMemcpyAsync(s.h_counts.p, rec_dev, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost,
                                      ctx->stream));
                s.host_has_records = true;
            }
        }
    }
    // counts of the chunk + (culled path) the number of (tile, hypothesis) pairs the launch evaluated
    if (dense)
        HIPCHK(hipMemcpyAsync(s.h_counts.p, s.counts.p, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost,
                              ctx->stream));
    if (dense) HIPCHK(hipMemcpyAsync(s.h_valid.p, s.valid.p, (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    if (!caller_ships_records && !s.poll_seq) HIPCHK(hipEventRecord(s.done, ctx->stream));
    return M3D_OK;
}

// Completion of a chunk issued with a PickFinal: the last kernel of the chunk stores `seq` into the pinned
// BestPickHost behind everything else it (and the kernels before it) wrote to host memory.  A spin on that word
// replaces an event record between two kernels (a ~5 us bubble on the stream) and the event wait.
// The spin is bounded in time: a chunk on one million points is done in a few hundred microseconds, but one on ten million
// can take tens of milliseconds, and a thread that spins that long starves the host work it shares a core (or a cgroup CPU
// quota) with -- after m3d_config.wait_spin_us the loop sleeps between looks.
static int wait_pick_seq(DeviceCtx* ctx, uint32_t seq) {
    const volatile uint32_t* p = &ctx->h_pick.as<BestPickHost>()->seq;
    const int spin_us = std::max(config().wait_spin_us, 30);
    const auto t0 = std::chrono::steady_clock::now();
    bool sleeping = false;   // past m3d_config.wait_spin_us the thread sleeps between looks (a long chunk: no core burnt)
    for (uint32_t spins = 1;; ++spins) {
        if ((int32_t)(*p - seq) >= 0) break;
        if (sleeping) {
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 0x3Fu) == 0) {   // a fault on the device would leave the word unwritten
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q == hipSuccess) {
                    if ((int32_t)(*p - seq) >= 0) break;
                    return fail(M3D_ERR_INTERNAL, "the scoring chunk finished without its completion word");
                }
                if (q != hipErrorNotReady) return fail(M3D_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
            }
            continue;
        }
        if ((spins & 0xFFu) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us)
            sleeping = true;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return M3D_OK;
}

// The end of a fit's device work, waited for by polling a word the stream's last (one-thread) kernel stores into page-locked
// memory instead of hipStreamSynchronize: the runtime's wait wakes the caller 10-20 us after the stream has drained.  Bounded
// in TIME (m3d_config.wait_spin_us, default 500 us; 0: the runtime's wait only): a wait that lasts longer -- a 10 M-point call,
// a device another lane keeps busy -- becomes a blocked thread instead of a core at 100 % (VERDICT r4 / ADVICE r4).
int stream_wait_spin(DeviceCtx* ctx) {
    const int spin_us = config().wait_spin_us;
    if (spin_us <= 0) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return M3D_OK;
    }
    if (!ctx->h_sync.p) {
        RESERVE(ctx->h_sync, 64);
        std::memset(ctx->h_sync.p, 0, 64);
    }
    if (++ctx->sync_seq == 0) ++ctx->sync_seq;   // (never 0: the word's initial value)
    const uint32_t seq = ctx->sync_seq;
    launch_signal_host(ctx->h_sync.as<uint32_t>(), seq, ctx->stream);
    HIPCHK(hipGetLastError());
    const volatile uint32_t* p = ctx->h_sync.as<uint32_t>();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; ++spins) {
        if (*p == seq) break;
        if ((spins & 0xFFu) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (*p != seq) return fail(M3D_ERR_INTERNAL, "the stream drained without its completion word");
            break;
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return M3D_OK;
}

// After the slot's `done` event: the culled path ships (valid << 31 | count) in one array (written by
// sum_replicas_k straight into the pinned h_counts); split it into the h_valid / h_counts views the replay and
// the callers read.
static void unpack_slot(ChunkSlot& s) {
    if (use_dense_scoring()) return;
    uint32_t* c = s.h_counts.as<uint32_t>();
    uint8_t* v = s.h_valid.as<uint8_t>();
    const size_t n = s.end - s.begin;
    for (size_t i = 0; i < n; ++i) {
        v[i] = (uint8_t)(c[i] >> 31);
        c[i] &= 0x7FFFFFFFu;
    }
}

// Scratch of launch_compact (m3d_kernels.hpp, CompactScratch): one slot per compaction workgroup, zero when the buffer is
// (re)allocated and when the epoch counter starts over; every launch gets the next epoch.
// ------------------------------------------------------------------------------------------------
// RANSAC::FitModelParallel on a resident view.  Leaves the best minimal model in
// device memory at ctx->last_best_dev; RefineModel's first kernel forwards it to ctx->h_best (pinned host).
// ------------------------------------------------------------------------------------------------
struct RansacOut {
    m3d_replay_state st;
    uint64_t hypotheses_scored = 0;
    uint64_t exact_rmse_evals = 0;
    int32_t ties = 0;
    double ms_sample = 0, ms_score = 0;
    double ms_score_kernel = 0;   // sum over the chunks' scoring-kernel launches (HIP events k0..k1)
    uint32_t score_launches = 0;
    uint64_t pairs_scored = 0;   // (tile, hypothesis) pairs the scoring launches evaluated (culled path)
    uint64_t pairs_exact = 0;    // ... of which the fp32 screen could not decide (recounted in fp64)
    uint64_t pairs_timed = 0;    // ... of pairs_scored: evaluated by the launches behind ms_score_kernel (a lead pass inside cull_lead_k is not)
    int internal_error = 0;
    int spec_hits = 0, spec_misses = 0;   // early compaction on the device's pick kept / redone
};

static int run_ransac(DeviceCtx* ctx, const CloudView& v, const SortedView& sv, int kind, double thr,
                      size_t max_iter, double prob, uint64_t seed, RansacOut* out, size_t iterations_hint = 0,
                      const uint32_t* orig_dev = nullptr /* index map of a shrunk cloud (RefineModel's compaction) */,
                      m3d_comm* comm = nullptr /* hypotheses sharded over its ranks (issue_chunk) */,
                      uint64_t* idx_host = nullptr /* the caller's page-locked index list (early compaction writes it) */) {
    m3d_replay_init(&out->st);
    RESERVE(ctx->best_params, sizeof(double) * kModelStride);
    RESERVE(ctx->h_small, 256);
    // Probability-1 fits: nothing but fitness == 1 stops the loop, so the winner is (up to rmse ties) the hypothesis
    // with the most inliers, lowest index first.  The device picks it itself after every chunk (pick_best_k) and,
    // behind the LAST chunk, RefineModel's compaction is queued on that pick at once -- while the host is still
    // waking up and replaying the records.  The replay stays the authority: cloud_fit_locked keeps the early
    // compaction only if it names the same hypothesis.  m3d_config.speculative_refine = 0 switches the prediction off.
    bool hinted_chunk = false;   // the adaptive-stop fit got ONE chunk sized from the caller's iteration hint (below)
    const bool spec_enabled = config().speculative_refine != 0;
    const bool spec = spec_enabled && prob >= 1.0 && !use_dense_scoring() && max_iter > 0;
    // Adaptive stop (probability < 1) with a caller's hint that the loop will run to max_iter (a segmentation round on
    // clutter: fitness ~0.005, the bound never bites): the hinted chunk covers the whole loop, so the device's pick behind
    // it IS the replay's winner unless the loop stops early after all or an rmse tie goes to a later hypothesis -- the
    // compaction (and the round's partition) is queued on the pick at once, and the host's ~15 us of waking up, replaying
    // and launching disappear from the round.  A miss redoes RefineModel, exactly as on the probability-1 path.
    ctx->spec_compaction = false;
    // one GPU, culled path: every chunk ends with the completion word of sum_replicas_k's tail (and its pick, which only
    // probability-1 fits act on) -- the host polls pinned memory instead of waiting for an event
    const bool poll_done = !comm && !use_dense_scoring() && max_iter > 0;
    if (spec || poll_done) {
        RESERVE(ctx->pick, sizeof(BestPick));
        if (!ctx->h_pick.p) {
            RESERVE(ctx->h_pick, 128);
            std::memset(ctx->h_pick.p, 0, 128);   // (BestPickHost::seq starts at 0; wait_pick_seq's values never are)
        }
        RESERVE(ctx->h_best, sizeof(double) * 2 * kModelStride);
    }
    SampleSource src;
    src.seed(seed);
    src.n_points = v.n;
    src.m = minimal_sample(kind);

    // sharded: the cap holds per rank, and the geometric start gives every rank a first slice of 128
    const size_t n_ranks = comm ? (size_t)comm->world : 1;
    const size_t chunk_cap = chunk_cap_for(v, sv, kind == M3D_CYLINDER && prob >= 1.0 ? 2 : 1) * n_ranks;
    // prob < 1: the adaptive bound usually stops the loop after O(100) hypotheses -> start small;
    // prob == 1: only fitness == 1 can stop it -> as few, equal chunks as the scratch cap allows
    // (one chunk up to chunk_cap hypotheses; more chunks are pipelined two deep)
    // Either way an incumbent exists early -- a small first chunk (prob < 1) or the first hypotheses (lead_size()) of
    // the first chunk counted in a pass of their own (prob == 1): its inlier count lets everything after it
    // skip the hypotheses that cannot reach it (bound-and-prune).
    size_t chunk = 128 * n_ranks;
    size_t growth = 2;
    size_t after_first = 0;  // prob == 1: size of the chunks after the first one
    uint32_t lead = 0;   // prob == 1: leading hypotheses of the FIRST chunk counted on their own (issue_chunk)
    if (prob >= 1.0 && max_iter > 1024) {
        const size_t n_chunks = (max_iter + chunk_cap - 1) / chunk_cap;
        chunk = after_first = ((max_iter + n_chunks - 1) / n_chunks + 63) / 64 * 64;
        lead = lead_size();   // (C2 without it: 22 % of the (tile, hypothesis) pairs evaluated instead of 7.7 %, the step 0.44 ms instead of 0.28)
        // Several chunks: a SHORT first one.  What prunes chunk k is the best count of chunks 0 .. k - 1, and the first chunk has
        // only its 128 leading hypotheses to prune against -- on C3's cylinders (50 000 hypotheses, four chunks) its scoring
        // launches took 353 us where each later chunk took 169.  A first chunk of 2048 puts a good incumbent in front of 96 %
        // of the hypotheses instead of 75 %, with the same number of chunks when the rest still fits them.
        const size_t first_small = (size_t)config().first_chunk;
        if (n_chunks >= 2 && first_small >= 2 * (size_t)lead && first_small * n_ranks < chunk) {
            const size_t first = first_small * n_ranks;
            const size_t rest = max_iter - first;
            const size_t n_rest = (rest + chunk_cap - 1) / chunk_cap;
            chunk = first;
            after_first = ((rest + n_rest - 1) / n_rest + 63) / 64 * 64;
        }
    } else if (prob >= 1.0) {
        chunk = std::max<size_t>(max_iter, 64);
        growth = 1;
    } else if ((hinted_chunk = iterations_hint > 2 * (size_t)lead_size())) {
        // adaptive stop, but the caller knows how many iterations a similar fit just took (segmentation rounds):
        // ONE chunk of that size whose first hypotheses are counted on their own -- the same device work as a
        // small first chunk with the second one queued behind it, minus a sample upload, a MinimalFit and a
        // box-test launch.  Should the bound not be reached, the loop below issues what is left of it.
        chunk = (iterations_hint + iterations_hint / 16 + 16 + 63) / 64 * 64;
        // (a segmentation round in the clutter: the best plane holds well under a percent of the cloud while every
        // hypothesis' bound -- the population of the tiles its slab meets -- is tens of times that: nothing is ever pruned,
        // and the lead pass that exists to prune costs a launch and a fold.  The caller says so: DeviceCtx::no_prune_hint.)
        lead = ctx->no_prune_hint ? 0u : lead_size();
        iterations_hint = 0;   // (consumed: no second chunk queued on the first pass)
    }
    chunk = std::min(std::min(chunk, chunk_cap), std::max<size_t>((max_iter + 63) / 64 * 64, 64));
    RESERVE(ctx->best_count, 32);   // cleared by the first chunk's minimal_fit_k (eight words: best count, ticket, key, key2, the survivor list's length at word 6)

    const bool timing_events = config().kernel_timing != 0;   // (m3d_stats.ms_score / ms_score_kernel)
    const double t_score0 = now_ms();   // (m3d_stats.ms_score: a host clock -- an event pair here cost a fit ~8 us)
    int rc = M3D_OK;
    int cur = 0;
    size_t next_begin = 0;
    double best_approx = 0, pending_approx = 0;  // tree error sums of the current best / the trial
    bool best_approx_known = false, pending_valid = false;
    // device address of the current best model: inside the params of the slot whose chunk produced it, until
    // that slot is about to be re-issued -- only then is it copied to ctx->best_params (a fit of one or two
    // chunks never needs the copy; RefineModel reads the record where it lies)
    const double* best_dev = ctx->best_params.as<double>();
    int best_slot = -1;
    // in_flight: hypotheses already issued whose records have not been replayed yet
    const bool prestream_on = prestream_enabled();
    auto issue_next = [&](int slot_id, size_t in_flight, size_t forced = 0) -> int {
        // a later chunk of a probability-1 fit on one GPU: its MinimalFit and box tests run on pre_stream, under the scoring
        // launches of the chunk before (issue_chunk, `pre`)
        const bool pre = prestream_on && prob >= 1.0 && !comm && next_begin > 0 && !use_dense_scoring() && !ctx->poison_pending;
        // the slot holding the best model is recycled: issue_chunk moves the record out first, on the stream that is about
        // to overwrite the slot (MinimalFit of the new chunk)
        const double* move_best_from = slot_id == best_slot ? best_dev : nullptr;
        size_t want = chunk;
        if (forced) {
            want = forced;
        } else if (prob < 1.0 && out->st.best_index >= 0) {
            // a best model exists: the adaptive bound (ransac.h:605-611) can only shrink from here, so ONE chunk
            // that covers what is left of it (+6 % for invalid minimal fits, which do not count) ends the loop.
            // Scoring a few hundred pruned hypotheses too many is cheaper than another chunk's launches.
            const uint64_t done = out->st.count + in_flight;
            const uint64_t left = out->st.current_iteration > done ? out->st.current_iteration - done : 0;
            want = (size_t)((left + left / 16 + 16 + 63) / 64 * 64);
        }
        want = std::min(std::max<size_t>(want, 64), chunk_cap);
        const size_t b = next_begin, e = std::min(max_iter, b + want);
        // one GPU: the chunk's last kernel (sum_replicas_k) picks the device's best itself and stores the completion word
        PickFinal pf;
        const bool fused_pick = poll_done;
        if (fused_pick) {
            pf.key = reinterpret_cast<unsigned long long*>(ctx->best_count.as<uint32_t>() + 2);
            pf.key2 = reinterpret_cast<unsigned long long*>(ctx->best_count.as<uint32_t>() + 4);
            pf.ticket = ctx->best_count.as<uint32_t>() + 1;
            pf.index_base = (unsigned long long)b;
            pf.first_chunk = b == 0 ? 1 : 0;
            pf.pick = ctx->pick.as<BestPick>();
            pf.pick_host = ctx->h_pick.as<BestPickHost>();
            if (++ctx->pick_seq == 0) ++ctx->pick_seq;
            pf.seq = ctx->pick_seq;
            // (pf.params: issue_chunk fills in the slot's parameter array once it has reserved it)
        }
        int r = issue_chunk(ctx, ctx->slot[slot_id], v, sv, kind, thr, b, e, src, &out->ms_sample, true,
                            b == 0 ? lead : 0, b == 0, spec, comm, /*caller_ships_records=*/spec && !fused_pick,
                            fused_pick ? &pf : nullptr, /*nothing_to_prune=*/b == 0 && e == max_iter && lead == 0 && !comm, pre,
                            move_best_from);
        if (move_best_from && r == M3D_OK) {
            best_dev = ctx->best_params.as<double>();
            best_slot = -1;
        }
        const bool spec_adaptive = spec_enabled && prob < 1.0 && hinted_chunk && b == 0 && !comm && !use_dense_scoring();
        if (r == M3D_OK && fused_pick && (spec || spec_adaptive) && e == max_iter) {   // last chunk: RefineModel's first stage on the prediction, now
            // (a segmentation round: the partition of the rest rides along, decided without the inlier count: -2)
            const PartitionOut* part = spec_adaptive && ctx->partition_hook && orig_dev ? (*ctx->partition_hook)(-2) : nullptr;
            r = issue_refine_compaction(ctx, v, orig_dev, kind, thr, ctx->pick.as<BestPick>()->params,
                                        h_best_at(ctx), h_total_at(ctx), /*fused=*/true,
                                        idx_host, part);
            ctx->spec_compaction = r == M3D_OK;
            // ... and so does the removal of the same inliers from the sorted copy (-3): by the time the host has replayed
            // the round and learnt the inlier count, the device is already through both
            // (RefineModel waits for ev_compact: recorded here, in front of the removal, not when refine() gets to run)
            // (a round that may finish its RefineModel later -- DeviceCtx::defer_refine -- waits for no event at all)
            if (r == M3D_OK && part && ctx->defer_refine) {
                (void)(*ctx->partition_hook)(-3);
            } else if (r == M3D_OK && part && hipEventRecord(ctx->ev_compact, ctx->stream) == hipSuccess) {
                ctx->ev_compact_early = true;
                (void)(*ctx->partition_hook)(-3);
            }
        }
        if (r == M3D_OK && spec && !fused_pick) {
            ChunkSlot& sl = ctx->slot[slot_id];
            // (sharded: the records are the gathered ones; the other ranks' counts raise this rank's incumbent too.  Should the
            // records not be on their way to the host yet, the kernel passes them on to the pinned array the replay reads.)
            const bool ship = comm && !sl.host_has_records;
            launch_pick_best(sl.counts.as<uint32_t>(), (uint32_t)(e - b), (unsigned long long)b, sl.params.as<double>(),
                             b == 0, ctx->pick.as<BestPick>(), ctx->h_pick.as<BestPickHost>(), ctx->stream,
                             ship ? sl.h_counts.as<uint32_t>() : nullptr, comm ? ctx->best_count.as<uint32_t>() : nullptr);
            if (ship) sl.host_has_records = true;
            if (!sl.done_on_copy_stream) HIPCHK(hipEventRecord(sl.done, ctx->stream));
            if (e == max_iter) {   // last chunk: RefineModel's first stage on the prediction, now
                r = issue_refine_compaction(ctx, v, orig_dev, kind, thr, ctx->pick.as<BestPick>()->params,
                                            h_best_at(ctx), h_total_at(ctx), /*fused=*/true,
                                            idx_host);
                ctx->spec_compaction = r == M3D_OK;
            }
        }
        if (r == M3D_OK) {
            next_begin = e;
            out->hypotheses_scored += (e - b + n_ranks - 1) / n_ranks;   // this rank's share
            chunk = after_first ? std::min(after_first, chunk_cap) : std::min(chunk * growth, chunk_cap);
        }
        return r;
    };
    if (max_iter > 0) {
        if (prestream_on && prob >= 1.0 && !comm && max_iter > chunk && !use_dense_scoring()) {   // (a fit of several chunks)
            rc = pre_stream_gate(ctx);
            if (rc != M3D_OK) return rc;
        }
        rc = issue_next(0, 0);
        if (rc != M3D_OK) return rc;
        bool first_pass = true;
        for (;;) {
            ChunkSlot& s = ctx->slot[cur];
            bool issued = false;
            if (next_begin < max_iter) {
                // speculate only when the adaptive bound cannot be reached inside the current chunk
                const bool safe = prob >= 1.0 || (out->st.best_index >= 0 &&
                                                  out->st.current_iteration > out->st.count + (s.end - s.begin));
                if (safe) {
                    rc = issue_next(cur ^ 1, s.end - s.begin);
                    if (rc != M3D_OK) break;
                    issued = true;
                } else if (first_pass && prob < 1.0 && iterations_hint > s.end) {
                    // the caller knows how many iterations a similar fit just took (segmentation rounds): queue
                    // that many behind the first chunk without waiting for its counts (they prune on the device)
                    const size_t left = iterations_hint - s.end;
                    rc = issue_next(cur ^ 1, s.end - s.begin, (left + left / 16 + 16 + 63) / 64 * 64);
                    if (rc != M3D_OK) break;
                    issued = true;
                }
            }
            first_pass = false;
            if (s.poll_seq) {
                const int wrc = wait_pick_seq(ctx, s.poll_seq);
                if (wrc != M3D_OK) return wrc;
            } else {
                HIPCHK(hipEventSynchronize(s.done));
            }
            if (use_dense_scoring()) unpack_slot(s);   // (culled path: the replay reads the packed records as shipped)
            {
                float kms = 0;
                if (s.scored && hipEventElapsedTime(&kms, s.k0, s.k1) == hipSuccess) {
                    out->ms_score_kernel += kms;
                    out->score_launches++;
                }
                if (s.scored && s.lead_groups && !s.lead_fused && hipEventElapsedTime(&kms, s.k2, s.k3) == hipSuccess) {
                    out->ms_score_kernel += kms;
                    out->score_launches++;
                }
                if (!use_dense_scoring()) {
                    out->pairs_scored += s.h_counts.as<uint32_t>()[s.h_pad];
                    out->pairs_exact += s.h_counts.as<uint32_t>()[s.h_pad + 1];
                    out->pairs_timed += s.h_counts.as<uint32_t>()[s.h_pad] - (s.lead_fused ? s.h_counts.as<uint32_t>()[s.h_pad + 2] : 0u);
                }
            }
            int cb_rc = M3D_OK;
            // exact EvaluateModel rmse (serial-order error sum), ransac.h:632-650
            auto exact_rmse = [&](const double* model, uint32_t expect, bool check) -> double {
                uint64_t c = 0;
                double err = 0;
                const int r = exact_error(ctx, v, kind, thr, model, &c, &err);
                if (r != M3D_OK) cb_rc = r;
                out->exact_rmse_evals++;
                if (check && c != expect) out->internal_error = 1;
                return c == 0 ? 1e+10 : err / std::sqrt((double)c);
            };
            // Tie rule `inlier_rmse < inlier_rmse_` (ransac.h:596) for equal inlier counts n.  Both rmse
            // are error/sqrt(n) with the same n, hence monotone in the error sums.  Stage 1: order-free
            // tree sums A_t, A_b; any summation order of n non-negative terms is within n*u*sum of the
            // exact value, so the serial sums differ from A by at most 2*n*u*A: if the A's are further
            // apart than the two margins (x2 safety) the comparison of the serial sums is decided.
            // Stage 2 (rare: equal or near-equal hypotheses): the serial sums themselves.
            auto tie = [&](size_t i, uint32_t cnt, double* trial_rmse, bool* trial_known) -> bool {
                const double* model_t = s.params.as<double>() + (i - s.begin) * kModelStride;
                *trial_known = false;
                pending_valid = false;
                out->ties++;
                if (cnt == 0) {  // rmse = 1e10 (ransac.h:646) against the initial 0: never better
                    *trial_rmse = 1e+10;
                    *trial_known = true;
                    return *trial_rmse < out->st.best_rmse;
                }
                uint64_t c = 0;
                double a_t = 0;
                if (!best_approx_known) {   // both sums behind one wait
                    uint64_t cb = 0;
                    const int r = approx_error_pair(ctx, v, kind, thr, model_t, best_dev, &c, &a_t, &cb, &best_approx);
                    if (r != M3D_OK) cb_rc = r;
                    best_approx_known = true;
                } else {
                    const int r = approx_error(ctx, v, kind, thr, model_t, &c, &a_t);
                    if (r != M3D_OK) cb_rc = r;
                }
                if (c != cnt) out->internal_error = 1;
                const double nu4 = 4.0 * (double)cnt * 1.1102230246251565e-16;
                const double m_t = nu4 * a_t, m_b = nu4 * best_approx;
                if (a_t + m_t < best_approx - m_b) {
                    pending_valid = true;
                    pending_approx = a_t;
                    return true;
                }
                if (a_t - m_t > best_approx + m_b) return false;
                *trial_rmse = exact_rmse(model_t, cnt, true);
                *trial_known = true;
                if (!out->st.best_rmse_known) {
                    out->st.best_rmse = exact_rmse(best_dev, 0, false);
                    out->st.best_rmse_known = 1;
                }
                if (*trial_rmse < out->st.best_rmse) {
                    pending_valid = true;
                    pending_approx = a_t;
                    return true;
                }
                return false;
            };
            auto on_best = [&](size_t i) {
                // the model stays in the chunk's slot (issue_next moves it out before the slot is recycled)
                best_dev = s.params.as<double>() + (i - s.begin) * kModelStride;
                best_slot = cur;
                best_approx_known = pending_valid;
                best_approx = pending_approx;
                pending_valid = false;
            };
            if (use_dense_scoring())
                replay_range(&out->st, v.n, kind, max_iter, prob, s.begin, s.end, s.h_valid.as<uint8_t>(),
                             s.h_counts.as<uint32_t>(), tie, on_best);
            else if (prob >= 1.0)
                replay_range_exhaustive(&out->st, v.n, kind, max_iter, s.begin, s.end, s.h_counts.as<uint32_t>(), tie, on_best);
            else
                replay_range(&out->st, v.n, kind, max_iter, prob, s.begin, s.end, nullptr, s.h_counts.as<uint32_t>(), tie,
                             on_best);
            if (cb_rc != M3D_OK) {
                rc = cb_rc;
                break;
            }
            if (out->st.stopped) break;
            if (!issued) {
                if (next_begin >= max_iter) break;
                rc = issue_next(cur ^ 1, 0);
                if (rc != M3D_OK) break;
            }
            cur ^= 1;
        }
    }
    if (out->st.best_index < 0) {   // no valid hypothesis at all: the "best model" is all zeros
        HIPCHK(hipMemsetAsync(ctx->best_params.p, 0, sizeof(double) * kModelStride, ctx->stream));
        best_dev = ctx->best_params.as<double>();
    }
    ctx->last_best_dev = best_dev;
    if (timing_events) out->ms_score = now_ms() - t_score0;
    // The best minimal model travels to the host (pinned ctx->h_best) with RefineModel: its first kernel stores the
    // record there and RefineModel's own wait delivers it (no wait and no copy command here).
    RESERVE(ctx->h_best, sizeof(double) * 2 * kModelStride);
    if (rc != M3D_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    if (out->internal_error)
        return fail(M3D_ERR_INTERNAL, "scoring kernel and exact evaluation disagree on an inlier count");
    return M3D_OK;
}

int validate_fit_args(int kind, size_t n, bool has_normals, double prob) {
    // order of the reference's checks: FitCylinder normals (py_common.cpp:50-52), SetProbability
    // (ransac.h:482-487), FitModel's point count (ransac.h:509-513)
    if (kind == M3D_CYLINDER && !has_normals)
        return fail(M3D_ERR_NO_NORMALS, "Fit cylinder requires normals.");
    if (prob <= 0 || prob > 1 || prob != prob)
        return fail(M3D_ERR_PROBABILITY, "Probability must be > 0 or <= 1.0");
    if (n < (size_t)minimal_sample(kind))
        return fail(M3D_ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points");
    return M3D_OK;
}

uint64_t resolve_seed(const uint64_t* seed) {
    if (seed) return *seed;
    std::random_device rd;  // utils.h:75
    return rd();
}

// GeneralFit of a round whose RefineModel was left running (DeviceCtx::deferred): the stream has passed its kernels
int finalize_deferred_refine(DeviceCtx* ctx) {
    DeviceCtx::DeferredRefine& d = ctx->deferred;
    if (!d.pending) return M3D_OK;
    d.pending = false;
    const double* rec = ctx->h_best.as<double>() + (size_t)d.slot * kModelStride;
    uint32_t ni_chk;
    std::memcpy(&ni_chk, ctx->h_pick.as<uint8_t>() + 64 + 8 * d.slot, 4);
    if (ni_chk != d.ni) return fail(M3D_ERR_INTERNAL, "refine pass and scoring kernel disagree on the inlier count");
    double model[4] = {rec[0], rec[1], rec[2], rec[3]};   // the best minimal model, refined in place when GeneralFit succeeds
    const double c0[3] = {rec[4], rec[5], rec[6]};
    double mean[3], sums[14], out[4];
    moments_about_mean(ctx->h_moments.as<double>() + (size_t)d.slot * kFusedMomentDoubles, c0, (double)d.ni, mean, sums + 4);
    if (plane_from_moments(mean, sums + 4, out)) std::memcpy(model, out, sizeof(out));
    std::memcpy(d.params_out, model, sizeof(model));
    return M3D_OK;
}

// Tile frames + histograms for plane_bound_k (m3d_bound.hip): a property of the resident cloud's sorted copy like its tile
// boxes, but only plane fits with an incumbent to prune against use them -- built by the first such fit that is long enough to
// pay for it (one launch, ~0.05 ms on 1 M points), kept for the cloud's lifetime.  The working cloud of a segmentation
// (re-partitioned between rounds) has none.
static int ensure_plane_frames(m3d_cloud* c, int kind, size_t n_hypotheses) {
    // 1: where it pays (bound_pays, below) -- and from 65 536 hypotheses on for a cloud that lives for one call (tile_frames_k
    // takes ~0.09 ms per million points, the bound saves ~0.02 ms per 10 000 hypotheses on them: m3d_fit_plane 0.95 -> 1.04 ms
    // otherwise); 2: always (tests)
    const int mode = config().plane_bound;
    if ((kind == M3D_SPHERE && mode != 2) || c->frames_ready || c->frames_failed || mode == 0 || c->work.active || c->n_tiles == 0 ||
        (mode == 1 && (!bound_pays(c->n_tiles, std::min<size_t>(n_hypotheses, chunk_cap_for(c->view(), c->sorted()))) ||
                       (c->one_shot && n_hypotheses < 65536u))))
        return M3D_OK;
    DeviceCtx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    if (!c->frames.reserve(sizeof(double) * kFrameStride * (size_t)c->n_tiles) ||
        !c->frame_cum.reserve(sizeof(uint16_t) * kCumStride * (size_t)c->n_tiles)) {
        // the bound is an optimisation: without it the keep rule prices a hypothesis at 512 points per touched tile and the
        // fit returns the same result -- a cloud that cannot have its frames (memory pressure) fits without, and is not
        // asked again
        c->frames.release();
        c->frame_cum.release();
        c->frames_failed = true;
        return M3D_OK;
    }
    launch_tile_frames(c->sorted(), c->frames.as<double>(), c->frame_cum.as<uint16_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    c->frames_ready = true;
    return M3D_OK;
}

int cloud_fit_locked(m3d_cloud* c, int kind, double thr, size_t max_iter, double prob,
                            uint64_t seed, double* params, size_t* inliers, size_t* n_inliers,
                            m3d_stats* stats, const std::function<int(int64_t)>* before_refine_wait,
                            size_t* iterations_hint /* in: iterations of a similar fit, out: of this one */,
                            m3d_comm* comm) {
    DeviceCtx* ctx = c->ctx;
    const double t0 = now_ms();
    HIPCHK(hipSetDevice(ctx->device));
    struct DenseGuard {
        int saved;
        explicit DenseGuard(bool on) : saved(t_dense_fit) { if (on) t_dense_fit = 1; }
        ~DenseGuard() { t_dense_fit = saved; }
    } dense_guard(c->n_tiles == 0 && !c->work.active);   // no sorted copy: dense scoring (m3d_cloud_create_impl)
    {
        const int frc = ensure_plane_frames(c, kind, max_iter);
        if (frc != M3D_OK) return frc;
    }
    const CloudView v = c->view();
    const CloudView gather = c->base_view();   // index lists / GeneralFit refer to the cloud as created
    const uint32_t* orig = c->orig();
    RansacOut ro;
    uint64_t* idx_host = inliers && !ctx->idx_out_override &&
                                 is_library_pinned(inliers, sizeof(uint64_t) * (size_t)std::max<uint32_t>(v.n, 1))
                             ? reinterpret_cast<uint64_t*>(inliers) : nullptr;
    ctx->compaction_idx_host = nullptr;
    ctx->compaction_fused = false;
    ctx->spec_hit = false;
    ctx->ev_compact_early = false;
    if (ctx->defer_refine) ctx->refine_slot ^= 1;   // (the previous fit's words may still be waiting for finalize_deferred_refine)
    int rc = run_ransac(ctx, v, c->sorted(), kind, thr, max_iter, prob, seed, &ro, iterations_hint ? *iterations_hint : 0,
                        orig, comm, idx_host);
    if (rc != M3D_OK) return rc;
    // (this fit's records have arrived: the stream is past the previous fit's RefineModel)
    rc = finalize_deferred_refine(ctx);
    if (rc != M3D_OK) return rc;
    if (iterations_hint) *iterations_hint = (size_t)ro.st.iterations;
    const double t1 = now_ms();
    double model[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};   // filled from ctx->h_best by refine's wait
    size_t ni = 0;
    int gf_ok = 1;
    const int64_t expected = ro.st.best_index >= 0 ? (int64_t)ro.st.best_count : -1;
    bool refined = false, deferred_now = false;
    if (ctx->spec_compaction) {
        // the compaction is already running on the device's pick (its record reached pinned memory with the chunk's
        // completion word): RefineModel is finished on it if the replay named the same hypothesis -- it does unless an rmse
        // tie went the other way or, on the adaptive path, the loop stopped early after all.  Otherwise the speculative
        // launches are left to drain (their outputs are rewritten below, later in stream order) and RefineModel runs on
        // the replay's model with the caller's hooks untouched.
        const BestPickHost* ph = ctx->h_pick.as<BestPickHost>();
        // (a pick marked `tie` shipped nothing: its model record was poisoned -- sum_replicas_k, "COUNT TIES")
        const bool hit = !ph->tie && (ro.st.best_index >= 0 ? (ph->have && ph->index == (unsigned long long)ro.st.best_index) : !ph->have);
        ro.spec_hits = hit ? 1 : 0;
        ro.spec_misses = hit ? 0 : 1;
        ctx->spec_hit = hit;
        const bool defer = hit && ctx->defer_refine && !comm && kind == M3D_PLANE && expected >= 3 && ctx->compaction_fused &&
                           inliers && ctx->compaction_idx_host == reinterpret_cast<uint64_t*>(inliers) && before_refine_wait &&
                           (uint64_t)expected <= v.n;
        if (defer) {
            // nothing of RefineModel is needed to go on: the count is the scoring pass's, the list is on its way to the
            // caller's pinned array, the refined plane is only reported
            const int hr = (*before_refine_wait)(expected);
            before_refine_wait = nullptr;
            if (hr != M3D_OK) return hr;
            ctx->deferred.pending = true;
            ctx->deferred.slot = refine_slot(ctx);
            ctx->deferred.ni = (uint32_t)expected;
            ctx->deferred.params_out = params;
            ni = (size_t)expected;
            refined = deferred_now = true;
        } else if (hit) {
            rc = refine(ctx, v, gather, orig, kind, thr, ctx->pick.as<BestPick>()->params, model, inliers, &ni, &gf_ok,
                        expected, before_refine_wait, h_best_at(ctx), h_total_at(ctx), /*fused=*/true);
            if (rc != M3D_OK) return rc;
            refined = true;
        }
    }
    if (ctx->spec_compaction && comm) {
        // sharded fits: pick_best_k's record is only known to have landed once the stream has been waited for (its
        // chunk's event may sit on the copy stream, in front of it) -- the decision above is re-taken after that wait
        const BestPickHost* ph = ctx->h_pick.as<BestPickHost>();
        if (!refined) {   // (judged a miss on a record that may not have arrived yet)
            rc = refine(ctx, v, gather, orig, kind, thr, ctx->pick.as<BestPick>()->params, model, inliers, &ni, &gf_ok,
                        expected, before_refine_wait, h_best_at(ctx), h_total_at(ctx), /*fused=*/true);
            if (rc != M3D_OK) return rc;
        }
        before_refine_wait = nullptr;   // (a speculative RefineModel has run its hook by now, in either branch)
        refined = ro.st.best_index >= 0 ? (ph->have && ph->index == (unsigned long long)ro.st.best_index) : !ph->have;
        ro.spec_hits = refined ? 1 : 0;
        ro.spec_misses = refined ? 0 : 1;
    }
    if (!refined) {
        rc = refine(ctx, v, gather, orig, kind, thr, ctx->last_best_dev, model, inliers, &ni,
                    &gf_ok, expected, before_refine_wait, h_best_at(ctx), nullptr,
                    /*fused=*/true);
        if (rc != M3D_OK) return rc;
    }
    if (ro.st.best_index >= 0 && ni != ro.st.best_count)
        return fail(M3D_ERR_INTERNAL, "refine pass and scoring kernel disagree on the inlier count");
    if (n_inliers) *n_inliers = ni;
    if (!deferred_now) std::memcpy(params, model, sizeof(double) * num_params(kind));   // (deferred: finalize_deferred_refine writes them)
    const double t2 = now_ms();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->fitness = ro.st.best_fitness;
        stats->inlier_rmse = ro.st.best_rmse_known ? ro.st.best_rmse : std::numeric_limits<double>::quiet_NaN();
        stats->count = ro.st.count;
        stats->iterations = ro.st.iterations;
        stats->best_index = ro.st.best_index;
        stats->general_fit_ok = gf_ok;
        stats->hypotheses_scored = ro.hypotheses_scored;
        stats->exact_rmse_evals = ro.exact_rmse_evals;
        stats->ties = ro.ties;
        stats->ms_sample = ro.ms_sample;
        stats->ms_score = ro.ms_score;
        stats->ms_score_kernel = ro.ms_score_kernel;
        stats->score_launches = ro.score_launches;
        stats->pairs_scored = ro.pairs_scored;
        stats->pairs_exact = ro.pairs_exact;
        stats->pairs_timed = ro.pairs_timed;
        stats->early_pick_redone = (uint32_t)ro.spec_misses;   // 1: the device's early pick lost an rmse tie, RefineModel was redone
        stats->ms_refine = t2 - t1;
        stats->ms_total = t2 - t0;
    }
    return gf_ok ? M3D_OK : M3D_FALSE;
}
}  // namespace m3d

using namespace m3d;

extern "C" {

void m3d_replay_init(m3d_replay_state* st) {
    std::memset(st, 0, sizeof(*st));
    st->best_fitness = 0;  // Clear(), ransac.h:519-522
    st->best_rmse = 0;
    st->best_rmse_known = 1;
    st->best_index = -1;
    st->current_iteration = std::numeric_limits<uint64_t>::max();  // ransac.h:569
}

void m3d_replay_chunk(m3d_replay_state* st, size_t n_points, int kind, size_t max_iteration,
                      double probability, size_t begin, size_t end, const uint8_t* valid,
                      const uint32_t* counts, m3d_rmse_fn rmse_cb, void* user) {
    auto tie = [&](size_t i, uint32_t cnt, double* trial_rmse, bool* trial_known) -> bool {
        *trial_rmse = cnt == 0 ? 1e+10 : (rmse_cb ? rmse_cb(user, i) : 0.0);
        *trial_known = true;
        if (!st->best_rmse_known) {
            st->best_rmse = rmse_cb ? rmse_cb(user, (size_t)st->best_index) : 0.0;
            st->best_rmse_known = 1;
        }
        return *trial_rmse < st->best_rmse;
    };
    replay_range(st, n_points, kind, max_iteration, probability, begin, end, valid, counts, tie, [](size_t) {});
}

int m3d_cloud_fit(m3d_cloud* c, int kind, double threshold, size_t max_iteration, double probability,
                  const uint64_t* seed, double* params, size_t* inliers, size_t* n_inliers,
                  m3d_stats* stats) {
    if (!c || !params || kind < 0 || kind > 2) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int vr = validate_fit_args(kind, c->n, c->has_normals, probability);
    if (vr != M3D_OK) return vr;
    CtxLock lock(c->ctx);
    return cloud_fit_locked(c, kind, threshold, max_iteration, probability, resolve_seed(seed), params,
                            inliers, n_inliers, stats);
}

// python/py_common.cpp:11-78's callers' loops as ONE call: see the header.  Jobs are grouped by the lane their cloud lives on;
// a worker thread per group (at most `inflight` at a time) holds the lane for its whole group.
int m3d_cloud_fit_batch(m3d_fit_job* jobs, size_t n_jobs, int inflight) {
    if (!jobs && n_jobs) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    std::vector<DeviceCtx*> lanes;
    std::vector<std::vector<size_t>> groups;
    for (size_t k = 0; k < n_jobs; ++k) {
        m3d_fit_job& j = jobs[k];
        j.rc = M3D_ERR_INTERNAL;
        j.n_inliers = 0;
        std::memset(j.params, 0, sizeof(j.params));
        std::memset(&j.stats, 0, sizeof(j.stats));
        if (!j.cloud || j.kind < 0 || j.kind > 2) {
            j.rc = M3D_ERR_INVALID_ARG;
            return fail(M3D_ERR_INVALID_ARG, "job " + std::to_string(k) + ": invalid argument");
        }
    }
    for (size_t k = 0; k < n_jobs; ++k) {
        size_t g = 0;
        while (g < lanes.size() && lanes[g] != jobs[k].cloud->ctx) ++g;
        if (g == lanes.size()) {
            lanes.push_back(jobs[k].cloud->ctx);
            groups.emplace_back();
        }
        groups[g].push_back(k);
    }
    std::vector<std::string> errs(n_jobs);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups.size()) break;
            try {
                CtxLock lock(lanes[g]);
                for (size_t k : groups[g]) {
                    m3d_fit_job& j = jobs[k];
                    int rc = validate_fit_args(j.kind, j.cloud->n, j.cloud->has_normals, j.probability);
                    size_t ni = 0;
                    if (rc == M3D_OK) {
                        const uint64_t sd = j.seed;
                        rc = cloud_fit_locked(j.cloud, j.kind, j.threshold, (size_t)j.max_iteration, j.probability,
                                              resolve_seed(j.has_seed ? &sd : nullptr), j.params, j.inliers, &ni, &j.stats);
                    }
                    j.n_inliers = ni;
                    j.rc = rc;
                    if (rc < 0) errs[k] = m3d_last_error();
                }
            } catch (...) {   // (extern "C": nothing may pass; the group's remaining jobs keep M3D_ERR_INTERNAL)
                for (size_t k : groups[g])
                    if (jobs[k].rc == M3D_ERR_INTERNAL && errs[k].empty()) errs[k] = "an exception in the batch worker (out of memory?)";
            }
        }
    };
    const size_t want = std::min<size_t>(groups.size(), inflight > 0 ? (size_t)inflight : groups.size());
    std::vector<std::thread> th;
    try {
        for (size_t t = 1; t < want; ++t) th.emplace_back(worker);
    } catch (...) {   // (no more threads: the calling thread drains the queue)
    }
    worker();
    for (auto& t : th) t.join();
    for (size_t k = 0; k < n_jobs; ++k)
        if (jobs[k].rc < 0) {
            set_error("job " + std::to_string(k) + ": " + errs[k]);
            return jobs[k].rc;
        }
    return M3D_OK;
}

}  // extern "C"
namespace m3d {
// std::random_device seeds differ per rank: rank 0's is the fit's (one tiny exchange, only when no seed was given)
int agree_seed(m3d_comm* comm, const uint64_t* seed, hipStream_t st, uint64_t* out) {
    const uint64_t mine = resolve_seed(seed);
    *out = mine;
    if (!comm || comm->world == 1 || seed) return M3D_OK;
    std::vector<uint64_t> all((size_t)comm->world);
    const int rc = comm->allgather_host(&mine, all.data(), sizeof(uint64_t), st);
    if (rc == M3D_OK) *out = all[0];
    return rc;
}

}  // namespace m3d
extern "C" {

int m3d_cloud_fit_sharded(m3d_cloud* c, m3d_comm* comm, int kind, double threshold, size_t max_iteration,
                          double probability, const uint64_t* seed, double* params, size_t* inliers,
                          size_t* n_inliers, m3d_stats* stats) {
    if (!comm) return m3d_cloud_fit(c, kind, threshold, max_iteration, probability, seed, params, inliers, n_inliers, stats);
    if (!c || !params || kind < 0 || kind > 2) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int vr = validate_fit_args(kind, c->n, c->has_normals, probability);
    if (vr != M3D_OK) return vr;
    if (comm->transport == m3d_comm::kRccl && comm->device != c->ctx->logical)
        return fail(M3D_ERR_INVALID_ARG, "the cloud and the RCCL communicator live on different devices");
    CtxLock lock(c->ctx);
    HIPCHK(hipSetDevice(c->ctx->device));
    uint64_t sd = 0;
    const int rs = agree_seed(comm, seed, c->ctx->stream, &sd);
    if (rs != M3D_OK) return rs;
    return cloud_fit_locked(c, kind, threshold, max_iteration, probability, sd, params, inliers, n_inliers, stats,
                            nullptr, nullptr, comm);
}

static int one_shot_fit(int kind, const double* xyz, const double* normals, size_t n, double thr,
                        size_t max_iter, double prob, const uint64_t* seed, int device, double* params,
                        size_t* inliers, size_t* n_inliers, m3d_stats* stats) {
    if (!params || (!xyz && n)) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int vr = validate_fit_args(kind, n, normals != nullptr, prob);
    if (vr != M3D_OK) return vr;
    // SetPointCloud, ransac.h:469-475.  The Hilbert-sorted copy and its tile boxes (0.2 ms per million points) pay for
    // themselves from about a thousand hypotheses on: a call that cannot run more -- the README's fit_plane(pcd, 0.01,
    // 100), the default 1000 with the adaptive stop (a few dozen iterations on a cloud with a dominant plane) -- gets a
    // cloud without them and the dense scoring kernel (identical results: tests/test_gpu_parity.py runs both paths).
    // 1 M points, defaults: 0.84 -> 0.67 ms.  With probability 1 every one of the max_iter hypotheses is a full fp64 pass over the
    // cloud (4.9 T pairs/s): beyond ~2e9 pairs -- 2 M points x 1000 -- the sort (0.2 ms per million points) is the cheaper
    // way even for a thousand hypotheses (the cut-over was on max_iter alone).
    const bool few = max_iter <= 1024 && (prob < 1.0 || (double)n * (double)max_iter <= 2.0e9);
    m3d_cloud* c = m3d_cloud_create_impl(xyz, normals, n, device, few ? 0 : 1);
    if (!c) return M3D_ERR_DEVICE;
    c->one_shot = true;
    const int rc = m3d_cloud_fit(c, kind, thr, max_iter, prob, seed, params, inliers, n_inliers, stats);
    m3d_cloud_destroy(c);
    return rc;
}

int m3d_fit_plane(const double* xyz, size_t n, double threshold, size_t max_iteration,
                  double probability, const uint64_t* seed, int device, double params[4],
                  size_t* inliers, size_t* n_inliers, m3d_stats* stats) {
    return one_shot_fit(M3D_PLANE, xyz, nullptr, n, threshold, max_iteration, probability, seed, device,
                        params, inliers, n_inliers, stats);
}
int m3d_fit_sphere(const double* xyz, size_t n, double threshold, size_t max_iteration,
                   double probability, const uint64_t* seed, int device, double params[4],
                   size_t* inliers, size_t* n_inliers, m3d_stats* stats) {
    return one_shot_fit(M3D_SPHERE, xyz, nullptr, n, threshold, max_iteration, probability, seed, device,
                        params, inliers, n_inliers, stats);
}
int m3d_fit_cylinder(const double* xyz, const double* normals, size_t n, double threshold,
                     size_t max_iteration, double probability, const uint64_t* seed, int device,
                     double params[7], size_t* inliers, size_t* n_inliers, m3d_stats* stats) {
    return one_shot_fit(M3D_CYLINDER, xyz, normals, n, threshold, max_iteration, probability, seed,
                        device, params, inliers, n_inliers, stats);
}

int m3d_draw_samples(size_t n_points, int kind, size_t n_hypotheses, uint64_t seed, uint32_t* samples) {
    if (kind < 0 || kind > 2 || !samples) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (n_points < (size_t)minimal_sample(kind))
        return fail(M3D_ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points");
    SampleSource src;
    src.seed(seed);
    src.n_points = n_points;
    src.m = minimal_sample(kind);
    src.fill(0, n_hypotheses, samples);
    return M3D_OK;
}

// MinimalFit on the host for ONE sample (ransac.h:576-582): the same m3d_fp.hpp code the device runs, compiled
// for the host with the same flags (no contraction), so the model is bit-identical to minimal_fit_k's.
int m3d_minimal_fit(int kind, const double* pts, const double* normals, double* model, uint8_t* valid) {
    if (kind < 0 || kind > 2 || !pts || !model || !valid || (kind == M3D_CYLINDER && !normals))
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    double par[kModelStride] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok;
    if (kind == M3D_PLANE)
        ok = plane_minimal_fit(pts, pts + 3, pts + 6, par);
    else if (kind == M3D_SPHERE)
        ok = sphere_minimal_fit(pts, par);
    else
        ok = cylinder_minimal_fit(pts, normals, par);
    if (!ok) std::memset(par, 0, sizeof(par));
    std::memcpy(model, par, sizeof(par));
    *valid = ok ? 1 : 0;
    return M3D_OK;
}

int m3d_cloud_score_range(m3d_cloud* c, int kind, double threshold, const uint32_t* samples,
                          size_t begin, size_t end, uint32_t* counts, uint8_t* valid, double* models) {
    if (!c || kind < 0 || kind > 2 || !samples || end < begin)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (kind == M3D_CYLINDER && !c->has_normals)
        return fail(M3D_ERR_NO_NORMALS, "Fit cylinder requires normals.");
    if (c->n < (size_t)minimal_sample(kind))
        return fail(M3D_ERR_TOO_FEW_POINTS, "Can not fit model due to lack of points");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    const CloudView v = c->view();
    const int m = minimal_sample(kind);
    for (size_t h = begin; h < end; ++h)
        for (int k = 0; k < m; ++k)
            if (samples[h * m + k] >= v.n) return fail(M3D_ERR_INVALID_ARG, "sample index out of range");
    SampleSource src;
    src.table = samples;
    src.m = m;
    const SortedView sv = c->sorted();
    const size_t chunk_cap = chunk_cap_for(v, sv);
    ChunkSlot& s = ctx->slot[0];
    for (size_t b = begin; b < end; b += chunk_cap) {
        const size_t e = std::min(end, b + chunk_cap);
        const int rc = issue_chunk(ctx, s, v, sv, kind, threshold, b, e, src, nullptr);
        if (rc != M3D_OK) return rc;
        HIPCHK(hipEventSynchronize(s.done));
        unpack_slot(s);
        if (counts) std::memcpy(counts + (b - begin), s.h_counts.p, sizeof(uint32_t) * (e - b));
        if (valid) std::memcpy(valid + (b - begin), s.h_valid.p, e - b);
        if (models)
            HIPCHK(hipMemcpy(models + (b - begin) * kModelStride, s.params.p,
                             sizeof(double) * kModelStride * (e - b), hipMemcpyDeviceToHost));
    }
    return M3D_OK;
}

// ---- sequential sampler object + sharded scoring (multi-GPU) -----------------------------------------
struct m3d_sampler {
    SampleSource src;
    int kind = 0;
    std::vector<uint32_t> table;  // every sample drawn so far, H x m
    size_t drawn = 0;
    uint32_t incumbent = 0;       // best inlier count this rank has seen for the stream so far (bound-and-prune)
    void draw_until(size_t h_end) {
        if (h_end <= drawn) return;
        table.resize(h_end * (size_t)src.m);
        src.fill(drawn, h_end, table.data() + drawn * (size_t)src.m);
        drawn = h_end;
    }
};

m3d_sampler* m3d_sampler_create(size_t n_points, int kind, uint64_t seed) {
    if (kind < 0 || kind > 2 || n_points < (size_t)minimal_sample(kind) || n_points >= ((size_t)1 << 31)) {
        set_error("m3d_sampler_create: invalid argument");
        return nullptr;
    }
    m3d_sampler* s = new m3d_sampler();
    s->kind = kind;
    s->src.seed(seed);
    s->src.n_points = n_points;
    s->src.m = minimal_sample(kind);
    return s;
}
void m3d_sampler_destroy(m3d_sampler* s) { delete s; }
size_t m3d_sampler_drawn(const m3d_sampler* s) { return s ? s->drawn : 0; }
const uint32_t* m3d_sampler_table(m3d_sampler* s, size_t n_hypotheses) {
    if (!s) return nullptr;
    s->draw_until(n_hypotheses);
    return s->table.data();
}

int m3d_cloud_score_shard(m3d_cloud* c, m3d_sampler* sampler, double threshold, size_t begin, size_t end,
                          size_t slice, uint32_t world, uint32_t rank, uint32_t* counts, uint8_t* valid,
                          size_t* n_mine) {
    if (!c || !sampler || !counts || !n_mine || end < begin || slice == 0 || world == 0 || rank >= world)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    const int kind = sampler->kind;
    if (kind == M3D_CYLINDER && !c->has_normals)
        return fail(M3D_ERR_NO_NORMALS, "Fit cylinder requires normals.");
    if (sampler->src.n_points != c->n) return fail(M3D_ERR_INVALID_ARG, "sampler and cloud sizes differ");
    if (sampler->drawn > begin) return fail(M3D_ERR_INVALID_ARG, "sampler is already past `begin`");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    {
        const int frc = ensure_plane_frames(c, kind, end - begin);
        if (frc != M3D_OK) return frc;
    }
    const CloudView v = c->view();
    const SortedView sv = c->sorted();
    const size_t chunk_cap = chunk_cap_for(v, sv);
    SampleSource tsrc;  // table-backed view of the sampler
    tsrc.m = sampler->src.m;
    size_t out = 0;
    struct Pending {
        bool active = false, discard = false;
        size_t out_pos = 0, n = 0;
    } pend[2];
    int cur = 0;
    auto collect = [&](int k) -> int {
        if (!pend[k].active) return M3D_OK;
        ChunkSlot& s = ctx->slot[k];
        HIPCHK(hipEventSynchronize(s.done));
        if (!pend[k].discard) {
            if (valid) {
                unpack_slot(s);
                std::memcpy(counts + pend[k].out_pos, s.h_counts.p, sizeof(uint32_t) * pend[k].n);
                std::memcpy(valid + pend[k].out_pos, s.h_valid.p, pend[k].n);
            } else if (!use_dense_scoring()) {   // records as shipped: valid << 31 | count
                std::memcpy(counts + pend[k].out_pos, s.h_counts.p, sizeof(uint32_t) * pend[k].n);
            } else {
                const uint32_t* hc = s.h_counts.as<uint32_t>();
                const uint8_t* hv = s.h_valid.as<uint8_t>();
                for (size_t i = 0; i < pend[k].n; ++i)
                    counts[pend[k].out_pos + i] = hc[i] | (hv[i] ? 0x80000000u : 0u);
            }
        }
        pend[k].active = pend[k].discard = false;
        return M3D_OK;
    };
    size_t j = 0;
    bool first_piece = true;
    // Slices of several pieces (what a low world size leaves a rank): a SHORT first own piece -- its best count is what prunes
    // the pieces behind it (run_ransac's short first chunk) -- and MinimalFit + box tests of every later piece on pre_stream,
    // under the scoring launches of the piece before (issue_chunk, `pre`).
    const bool prestream_on = prestream_enabled();
    constexpr size_t kFirstPiece = 2048;
    bool first_own = true;
    if (prestream_on && !use_dense_scoring()) {
        const int grc = pre_stream_gate(ctx);
        if (grc != M3D_OK) return grc;
    }
    // the pruning incumbent belongs to the FIT, i.e. to the sampler whose stream is being scored (another fit may
    // have used this device between two windows): it travels with the sampler
    RESERVE(ctx->best_count, 32);
    RESERVE(ctx->h_inc, 64);
    if (begin == 0) sampler->incumbent = 0;  // new fit
    *ctx->h_inc.as<uint32_t>() = sampler->incumbent;
    HIPCHK(hipMemcpyAsync(ctx->best_count.p, ctx->h_inc.p, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->best_count.as<uint32_t>() + 6, 0, sizeof(uint32_t), ctx->stream));   // (the survivor list's length: plane_bound_k)
    // A rank whose first slice starts late in the window does not wait for the host to walk the stream up to it
    // before the GPU gets work: the window's first hypotheses (lead_size()) (another rank's, lower in the sequence than
    // anything this rank owns -- exactly what bound-and-prune may use) are scored at once for their best count
    // only; their records are dropped (the owner reports them).  The rank's own slice then goes in one pruned
    // launch.
    {
        const size_t first_own = begin + (size_t)rank * slice;
        if (rank > 0 && first_own < end) {
            const size_t w = std::min<size_t>(std::min<size_t>(lead_size(), chunk_cap), first_own - begin);
            sampler->draw_until(begin + w);
            tsrc.table = sampler->table.data();
            const int rc = issue_chunk(ctx, ctx->slot[cur], v, sv, kind, threshold, begin, begin + w, tsrc, nullptr, true);
            if (rc != M3D_OK) return rc;
            pend[cur].active = pend[cur].discard = true;
            pend[cur].n = w;
            cur ^= 1;
            first_piece = false;
        }
    }
    for (size_t b = begin; b < end; b += slice, ++j) {
        const size_t e = std::min(end, b + slice);
        // every rank draws the whole stream (the host draws the other ranks' slices while this rank's
        // previous slice is being scored on the GPU); its own slices are drawn piece by piece, so the first
        // (small) piece is on the GPU before the rest of the slice has been drawn
        if (j % world != rank) {
            sampler->draw_until(e);
            continue;
        }
        // bound-and-prune against LOWER-index hypotheses only, which is what the sequential replay allows
        for (size_t bb = b; bb < e;) {
            // the first launch of the call counts its first lead_size() hypotheses on their own (issue_chunk's `lead`)
            const uint32_t lead = first_piece ? lead_size() : 0u;
            first_piece = false;
            const bool short_first = first_own && e - bb > chunk_cap && kFirstPiece >= 2 * (size_t)lead_size();
            const size_t ee = std::min(e, bb + (short_first ? kFirstPiece : chunk_cap));
            const bool pre = prestream_on && !first_own && lead == 0;
            first_own = false;
            sampler->draw_until(ee);
            int rc = collect(cur);
            if (rc != M3D_OK) return rc;
            tsrc.table = sampler->table.data();
            rc = issue_chunk(ctx, ctx->slot[cur], v, sv, kind, threshold, bb, ee, tsrc, nullptr, true, lead, false, false, nullptr, false,
                             nullptr, false, pre);
            if (rc != M3D_OK) return rc;
            pend[cur].active = true;
            pend[cur].out_pos = out;
            pend[cur].n = ee - bb;
            out += ee - bb;
            cur ^= 1;
            bb = ee;
        }
    }
    int rc = collect(0);
    if (rc != M3D_OK) return rc;
    rc = collect(1);
    if (rc != M3D_OK) return rc;
    HIPCHK(hipMemcpyAsync(ctx->h_inc.p, ctx->best_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    sampler->incumbent = *ctx->h_inc.as<uint32_t>();
    *n_mine = out;
    return M3D_OK;
}

int m3d_bench_time_score(m3d_cloud* c, int kind, double threshold, const uint32_t* samples,
                         size_t n_hypotheses, int reps, int mode, double* ms_avg, uint64_t* listed_pairs) {
    if (!c || kind < 0 || kind > 2 || !samples || !ms_avg || reps < 1 || n_hypotheses == 0 ||
        n_hypotheses > 16384 || mode < 0 || mode > 2)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (kind == M3D_CYLINDER && !c->has_normals)
        return fail(M3D_ERR_NO_NORMALS, "Fit cylinder requires normals.");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    const CloudView v = c->view();
    const SortedView sv = c->sorted();
    SampleSource src;
    src.table = samples;
    src.m = minimal_sample(kind);
    ChunkSlot& s = ctx->slot[0];
    int rc = issue_chunk(ctx, s, v, sv, kind, threshold, 0, n_hypotheses, src, nullptr);  // warm-up + records
    if (rc != M3D_OK) return rc;
    const uint32_t count = (uint32_t)n_hypotheses;
    const uint32_t n_tiles = v.n_pad / kScoreTile;
    // make sure the buffers of the timed mode exist whatever path issue_chunk took
    RESERVE(ctx->partial, sizeof(uint32_t) * (size_t)n_tiles * s.h_pad);
    const uint32_t n_groups = s.h_pad / 64;
    RESERVE(ctx->masks, sizeof(uint64_t) * (size_t)std::max<uint32_t>(sv.n_tiles, 1) * n_groups);
    RESERVE(ctx->keep, sizeof(uint64_t) * (size_t)n_groups);
    RESERVE(ctx->counts_rep, sizeof(uint32_t) * ((size_t)kCountReplicas * s.h_pad + kPairReplicas));
    RESERVE(ctx->small, 256);
    auto* masks = ctx->masks.as<unsigned long long>();
    auto* keep = ctx->keep.as<unsigned long long>();
    const float* c32 = (!use_dense_scoring() && config().cull_fp32 != 0 && sv.radius < 1e18) ? s.cull32.as<float>() : nullptr;
    launch_cull_mask(kind, sv, s.score.as<double>(), s.valid.as<uint8_t>(), count, n_groups, masks, nullptr, ctx->stream, false, 0,
                     0xFFFFFFFFu, c32);
    launch_keep_mask(nullptr, nullptr, n_groups, keep, ctx->stream);
    if (listed_pairs) {
        launch_count_bits(masks, keep, sv.n_tiles, n_groups, ctx->small.as<unsigned long long>(), ctx->stream);
        unsigned long long tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, ctx->small.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *listed_pairs = tot;  // (tile, hypothesis) pairs that survive the box test
    }
    const uint32_t splits = pick_splits(n_tiles, s.h_pad);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        if (mode == 0)
            launch_score_mask(kind, sv, s.score.as<double>(), masks, keep, n_groups, ctx->counts_rep.as<uint32_t>(),
                              s.h_pad, ctx->counts_rep.as<uint32_t>() + (size_t)kCountReplicas * s.h_pad, ctx->stream);
        else if (mode == 1)
            launch_cull_mask(kind, sv, s.score.as<double>(), s.valid.as<uint8_t>(), count, n_groups, masks, nullptr,
                             ctx->stream, false, 0, 0xFFFFFFFFu, c32);
        else
            launch_score(kind, v, s.score.as<double>(), s.h_pad, splits, ctx->partial.as<uint32_t>(), ctx->stream);
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_avg = (double)ms / reps;
    return M3D_OK;
}

// m3d_bench_plane_upper_bounds: plane_bound_k on its own -- the histogram upper bound of EVERY hypothesis of a sample table
// (nothing pruned, every hypothesis on the list), for the test that holds it against the exact counts
int m3d_bench_upper_bounds(m3d_cloud* c, int kind, double threshold, const uint32_t* samples, size_t n_hypotheses, uint32_t* ub_out);
int m3d_bench_plane_upper_bounds(m3d_cloud* c, double threshold, const uint32_t* samples, size_t n_hypotheses, uint32_t* ub_out) {
    return m3d_bench_upper_bounds(c, M3D_PLANE, threshold, samples, n_hypotheses, ub_out);
}
// ... for any kind (spheres and cylinders: cyl_pair_ub)
int m3d_bench_upper_bounds(m3d_cloud* c, int kind, double threshold, const uint32_t* samples, size_t n_hypotheses, uint32_t* ub_out) {
    if (!c || !samples || !ub_out || n_hypotheses == 0 || n_hypotheses > 16384 || kind < M3D_PLANE || kind > M3D_CYLINDER)
        return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    if (kind == M3D_CYLINDER && !c->has_normals) return fail(M3D_ERR_INVALID_ARG, "cylinder hypotheses need normals");
    DeviceCtx* ctx = c->ctx;
    CtxLock lock(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    if (c->work.active || c->n_tiles == 0) return fail(M3D_ERR_INVALID_ARG, "the cloud has no sorted copy of its own");
    if (!c->frames_ready) {
        if (!c->frames.reserve(sizeof(double) * kFrameStride * (size_t)c->n_tiles) ||
            !c->frame_cum.reserve(sizeof(uint16_t) * kCumStride * (size_t)c->n_tiles))
            return fail(M3D_ERR_DEVICE, "out of device memory (tile frames)");
        launch_tile_frames(c->sorted(), c->frames.as<double>(), c->frame_cum.as<uint16_t>(), ctx->stream);
        c->frames_ready = true;
    }
    const CloudView v = c->view();
    const SortedView sv = c->sorted();
    SampleSource src;
    src.table = samples;
    src.m = (uint32_t)minimal_sample(kind);
    ChunkSlot& s = ctx->slot[0];
    int rc = issue_chunk(ctx, s, v, sv, kind, threshold, 0, n_hypotheses, src, nullptr);   // records (+ fp32 box-test records)
    if (rc != M3D_OK) return rc;
    const uint32_t n_groups = s.h_pad / 64;
    RESERVE(ctx->masks, sizeof(uint64_t) * (size_t)std::max<uint32_t>(sv.n_tiles, 1) * n_groups);
    RESERVE(ctx->keep, sizeof(uint64_t) * (size_t)n_groups);
    RESERVE(ctx->small, 256);
    RESERVE(s.ub, sizeof(uint32_t) * 3 * (size_t)s.h_pad);
    rc = reserve_survivor_scratch(ctx, s.h_pad);
    if (rc != M3D_OK) return rc;
    auto* masks = ctx->masks.as<unsigned long long>();
    auto* keep = ctx->keep.as<unsigned long long>();
    uint32_t* ubsum = s.ub.as<uint32_t>() + s.h_pad;
    uint32_t* ctl = ctx->small.as<uint32_t>();   // [0]: an incumbent of 0 (the keep rule stays inert); [1]: the list's length
    uint32_t* surv = bound_list(ctx);
    const float* c32 = (!use_dense_scoring() && config().cull_fp32 != 0 && sv.radius < 1e18) ? s.cull32.as<float>() : nullptr;
    HIPCHK(hipMemsetAsync(ubsum, 0, sizeof(uint32_t) * (size_t)s.h_pad, ctx->stream));
    HIPCHK(hipMemsetAsync(ctl, 0, 8, ctx->stream));
    launch_cull_mask(kind, sv, s.score.as<double>(), s.valid.as<uint8_t>(), (uint32_t)n_hypotheses, n_groups, masks, nullptr,
                     ctx->stream, false, 0, 0xFFFFFFFFu, c32);
    launch_keep_mask(nullptr, nullptr, n_groups, keep, ctx->stream, nullptr, 0, 0, ctl + 1, surv);
    launch_plane_bound(kind, sv, s.score.as<double>(), masks, keep, n_groups, 0, n_groups, ubsum, ctl, ctl + 1, surv,
                       bound_tickets(ctx), c32, ctx->stream, /*always=*/true);
    HIPCHK(hipMemcpyAsync(ub_out, ubsum, sizeof(uint32_t) * n_hypotheses, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return M3D_OK;
}

// m3d_bench_last_segment_ms: where the calling thread's last segmentation call spent its wall clock

}  // extern "C"
