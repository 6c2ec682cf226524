// m3d_device.cpp -- the host driver's ground floor: thread-local errors, grow-only device / pinned buffers with per-lane free
// lists, device contexts = LANES (m3d_driver.hpp), resident clouds (SetPointCloud, include/misc3d/common/ransac.h:469-475: upload,
// transpose, bounding box, Hilbert sort, tile boxes) and page-locked host blocks.  There is no CPU fallback: every entry point
// needs a HIP device.
#include "m3d_driver_internal.hpp"

#pragma clang fp contract(off)

namespace m3d {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

namespace {
constexpr int kPoolDevices = 16;
constexpr int kPools = kPoolDevices * kMaxLanes;   // one free list per (device, lane)
static size_t pool_limit() { return (size_t)config().pool_limit_mb << 20; }   // bytes parked per list
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void*> blocks[kPools];
    size_t bytes[kPools] = {};
};
DevPool& dev_pool() {
    static DevPool* p = new DevPool();   // never destroyed (buffers may be released from finalisers at exit)
    return *p;
}
thread_local int t_lane = 0;   // the lane the calling thread holds (CtxLock / LaneLock); 0 outside of any
thread_local int t_dev = -1;   // ... and its LOGICAL device (free lists are per logical device: two aliases of one physical device
                               // have streams of their own, and a list is ordered by one stream only); -1 outside of any: hipGetDevice
inline int pool_index(int dev, int lane) { return dev * kMaxLanes + lane; }
}  // namespace

bool DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return true;
    release();
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    int d = t_dev;
    if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
    const int ln = (t_lane >= 0 && t_lane < kMaxLanes) ? t_lane : 0;
    if (d >= 0 && d < kPoolDevices) {
        DevPool& pool = dev_pool();
        const int pi = pool_index(d, ln);
        std::lock_guard<std::mutex> lock(pool.mu);
        auto it = pool.blocks[pi].lower_bound(want);
        if (it != pool.blocks[pi].end() && it->first <= 2 * want + ((size_t)1 << 20)) {
            p = it->second;
            cap = it->first;
            dev = d;
            lane = ln;
            pool.bytes[pi] -= cap;
            pool.blocks[pi].erase(it);
            return true;
        }
    }
    if (hipMalloc(&p, want) != hipSuccess) {
        dev_pool_trim(d);   // give the parked blocks back and try once more
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc failed (" + std::to_string(want) + " bytes)");
            return false;
        }
    }
    cap = want;
    dev = d;
    lane = ln;
    return true;
}
void DevBuf::release() {
    if (p) {
        bool parked = false;
        if (dev >= 0 && dev < kPoolDevices && lane >= 0 && lane < kMaxLanes) {
            DevPool& pool = dev_pool();
            const int pi = pool_index(dev, lane);
            std::lock_guard<std::mutex> lock(pool.mu);
            if (pool.bytes[pi] + cap <= pool_limit()) {
                pool.blocks[pi].emplace(cap, p);
                pool.bytes[pi] += cap;
                parked = true;
            }
        }
        if (!parked) (void)hipFree(p);
    }
    p = nullptr;
    cap = 0;
    dev = -1;
    lane = 0;
}
// Frees every parked block of the device.  hipFree waits for the device, so a block another lane parked a moment ago
// (its kernels possibly still running: a lane's list is ordered by that lane's stream only) is idle when it goes.
void dev_pool_trim(int device) {
    if (device < 0 || device >= kPoolDevices) return;
    std::vector<void*> drop;
    {
        DevPool& pool = dev_pool();
        std::lock_guard<std::mutex> lock(pool.mu);
        for (int ln = 0; ln < kMaxLanes; ++ln) {
            const int pi = pool_index(device, ln);
            for (auto& kv : pool.blocks[pi]) drop.push_back(kv.second);
            pool.blocks[pi].clear();
            pool.bytes[pi] = 0;
        }
    }
    for (void* q : drop) (void)hipFree(q);
}
bool PinBuf::reserve(size_t bytes) {
    if (bytes <= cap) return true;
    release();
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        p = nullptr;
        set_error("hipHostMalloc failed (" + std::to_string(want) + " bytes)");
        return false;
    }
    cap = want;
    return true;
}
void PinBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

// ------------------------------------------------------------------------------------------------
// device context
// ------------------------------------------------------------------------------------------------
static std::mutex g_ctx_mu;
static std::map<int, DeviceCtx*> g_ctx;   // key: device * kMaxLanes + lane

int lane_count() { return std::min(std::max((int)config().lanes, 1), kMaxLanes); }

// Logical devices.  m3d_config.device_aliases = N > 0 makes ordinals 0 .. max(N, physical) - 1 valid, ordinal d living on
// physical device d % physical: what a one-GPU box needs to EXECUTE the code that deals work to several devices
// (run_on_devices, m3d_global_registration_batch, m3d_register_fragment_pairs) -- every logical device has its own lanes,
// streams, scratch, free lists and resident-fragment table, exactly as a second physical device would.  It proves the
// dealing, the per-device tables and the in-process exchange; it does NOT exercise peer traffic or RCCL across devices.
static int physical_count() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}
int logical_device_count() {
    const int phys = physical_count();
    if (phys <= 0) return 0;
    return std::max(phys, std::min((int)config().device_aliases, kPoolDevices));
}
int physical_device(int logical) {
    const int phys = physical_count();
    if (phys <= 0) {
        set_error("no HIP device available (misc3d_amd has no CPU fallback)");
        return -1;
    }
    if (logical < 0 || logical >= logical_device_count()) {
        set_error("invalid HIP device ordinal " + std::to_string(logical));
        return -1;
    }
    return logical % phys;
}

static DeviceCtx* create_lane_locked(int device, int lane);
DeviceCtx* get_lane(int device, int lane) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (lane < 0 || lane >= kMaxLanes) lane = 0;
    auto it = g_ctx.find(device * kMaxLanes + lane);
    if (it != g_ctx.end()) return it->second;
    // A device's first call creates lanes 0 AND 1 (see the stream layout below): the second lane's compute stream gets its hardware
    // queue before lane 0's copy stream can take it.
    DeviceCtx* mine = nullptr;
    // (m3d_config.lanes_eager, default 2: how many lanes' compute streams a device's first call creates.  Streams take hardware queues
    //  in the order they are created: a lane born AFTER another lane's copy / pre stream shares a queue and its calls take turns with
    //  that lane's -- a process that will run many calls side by side asks for all its lanes up front (4: batches of fits over four
    //  lanes 48 k fits/s instead of 15 k after a sliced match has run, four fragment pairs in flight 964 instead of 813 pairs/s;
    //  the price: single-threaded fits of several chunks 10-20 % slower -- C3 0.84 -> 0.94 / 0.64 -> 0.77 ms: profiles/r06_lanes_eager.txt))
    const int eager = std::min(std::max((int)config().lanes_eager, 1), kMaxLanes);
    for (int ln = 0; ln <= std::max(lane, std::min(eager, lane_count()) - 1); ++ln) {
        if (g_ctx.count(device * kMaxLanes + ln)) continue;
        DeviceCtx* c = create_lane_locked(device, ln);
        if (ln == lane) mine = c;
        if (!c) break;
    }
    if (mine) return mine;
    it = g_ctx.find(device * kMaxLanes + lane);
    if (it != g_ctx.end()) return it->second;
    return create_lane_locked(device, lane);
}
static DeviceCtx* create_lane_locked(int device, int lane) {
    const int phys = physical_device(device);
    if (phys < 0) return nullptr;
    if (hipSetDevice(phys) != hipSuccess) {
        set_error("hipSetDevice failed");
        return nullptr;
    }
    DeviceCtx* c = new DeviceCtx();
    c->device = phys;
    c->logical = device;
    c->lane = lane;
    // The lanes' streams and the runtime's hardware queues (tools/ubench/stream_queues.hip prints the map): a process gets FOUR queues
    // per stream priority; a new stream takes a free one and, once all four are taken, shares one -- two streams on one queue run
    // their kernels one after the other.  A lane has a compute stream, and a copy stream (normal priority) and a pre-stream (high
    // priority: a pool of its own) that the first call that needs them creates (copy_stream_of / pre_stream_of).  What was measured
    // (profiles/r05_stream_layouts.txt; single-threaded C3 / C5 fits, the loop over fragment pairs after a sliced match, threads of
    // plain fits): the three streams created with the lane -- lanes 0 and 1 took turns (two threads 1.24x of one); secondaries on
    // demand, lanes on demand -- fine unless lane 0's copy stream exists before lane 1 does (two pairs in flight 350 pairs/s instead
    // of 500); every lane's compute stream created by a device's first call -- the multi-threaded cases fine, single-threaded fits
    // of several chunks lost 15-20 % (C3 fit_sphere 0.64 -> 0.79 ms: the copy stream then shares a queue with an idle lane's
    // stream); a copy stream at high or low priority -- threads of fits 1.1-1.3x of one.  Kept: lanes 0 and 1 created by a device's
    // first call (get_lane), the others on demand, both secondaries on demand at the priorities above.
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_compact, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_pre_gate, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
        ok = hipEventCreateWithFlags(&c->slot[k].pre_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->slot[k].done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreate(&c->slot[k].k0) == hipSuccess && hipEventCreate(&c->slot[k].k1) == hipSuccess &&
             hipEventCreate(&c->slot[k].k2) == hipSuccess && hipEventCreate(&c->slot[k].k3) == hipSuccess;
    if (!ok) {
        set_error("failed to create HIP stream/events");
        delete c;
        return nullptr;
    }
    g_ctx[device * kMaxLanes + lane] = c;
    return c;
}
DeviceCtx* get_ctx(int device) { return get_lane(device, 0); }
// the lane's secondary streams, created on first use (the caller holds the lane); nullptr + last error on failure
hipStream_t copy_stream_of(DeviceCtx* ctx) {
    if (!ctx->copy_stream) {
        if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
            ctx->copy_stream = nullptr;
            set_error("failed to create the lane's copy stream");
        }
    }
    return ctx->copy_stream;
}
hipStream_t pre_stream_of(DeviceCtx* ctx) {
    if (!ctx->pre_stream) {   // (high priority: its short latency-bound kernels get in between the scoring workgroups)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&ctx->pre_stream, hipStreamNonBlocking, hi) != hipSuccess) {
            ctx->pre_stream = nullptr;
            set_error("failed to create the lane's pre stream");
        }
    }
    return ctx->pre_stream;
}
static DeviceCtx* find_lane(int device, int lane) {   // an existing lane, or nullptr
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    auto it = g_ctx.find(device * kMaxLanes + lane);
    return it == g_ctx.end() ? nullptr : it->second;
}

// calls holding a lane right now, all devices (a hint for calls that would rather spread over several streams when alone)
static std::atomic<int> g_lanes_held{0};
int lanes_held() { return g_lanes_held.load(std::memory_order_relaxed); }

CtxLock::CtxLock(DeviceCtx* c) : ctx_(c), prev_lane_(t_lane), prev_dev_(t_dev) {
    ctx_->mu.lock();
    g_lanes_held.fetch_add(1, std::memory_order_relaxed);
    t_lane = ctx_->lane;
    t_dev = ctx_->logical;
    config_pin();
}
CtxLock::~CtxLock() {
    config_unpin();
    t_lane = prev_lane_;
    t_dev = prev_dev_;
    g_lanes_held.fetch_sub(1, std::memory_order_relaxed);
    ctx_->mu.unlock();
}
// event k of the lane's spare events (no timing), created on first use; nullptr + last error on failure
hipEvent_t aux_event_of(DeviceCtx* ctx, size_t k) {
    while (ctx->aux_events.size() <= k) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            set_error("failed to create an event");
            return nullptr;
        }
        ctx->aux_events.push_back(e);
    }
    return ctx->aux_events[k];
}

namespace {
std::atomic<int> g_thread_arrivals{0};
thread_local int t_home_lane = -1;   // dealt on the thread's first LaneLock
}  // namespace

LaneLock::LaneLock(int device, int prefer) : prev_lane_(t_lane), prev_dev_(t_dev) {
    const int lanes = lane_count();
    if (prefer < 0 && t_home_lane < 0) t_home_lane = g_thread_arrivals.fetch_add(1, std::memory_order_relaxed);
    const int home = (prefer >= 0 ? prefer : t_home_lane) % lanes;
    DeviceCtx* mine = get_lane(device, home);
    if (!mine) return;
    if (mine->mu.try_lock()) {
        ctx = mine;
    } else {
        for (int ln = 0; ln < lanes && !ctx; ++ln) {
            if (ln == home) continue;
            DeviceCtx* c = get_lane(device, ln);
            if (c && c->mu.try_lock()) ctx = c;
        }
        if (!ctx) {
            mine->mu.lock();
            ctx = mine;
        }
    }
    g_lanes_held.fetch_add(1, std::memory_order_relaxed);
    t_lane = ctx->lane;
    t_dev = ctx->logical;
    config_pin();
}
LaneLock::~LaneLock() {
    if (!ctx) return;
    config_unpin();
    t_lane = prev_lane_;
    t_dev = prev_dev_;
    g_lanes_held.fetch_sub(1, std::memory_order_relaxed);
    ctx->mu.unlock();
}

const std::string& last_error_string() { return g_last_error; }

// m3d_host_alloc registry: is [p, p + bytes) inside a page-locked block handed out by this library?
// (never destroyed: m3d_host_free may still be called from a host language's finalisers at process exit)
struct PinnedRegistry {
    std::mutex mu;
    std::vector<std::pair<const char*, size_t>> blocks;
};
static PinnedRegistry& pinned_registry() {
    static PinnedRegistry* r = new PinnedRegistry();
    return *r;
}
bool is_library_pinned(const void* p, size_t bytes) {
    PinnedRegistry& r = pinned_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    const char* q = static_cast<const char*>(p);
    for (const auto& b : r.blocks)
        if (q >= b.first && q + bytes <= b.first + b.second) return true;
    return false;
}
// every device buffer a cloud can own, in one place
template <class F>
static void for_each_buffer(m3d_cloud* c, F f) {
    f(c->x); f(c->y); f(c->z); f(c->nx); f(c->ny); f(c->nz);
    f(c->sx); f(c->sy); f(c->sz); f(c->boxes); f(c->tile_f32); f(c->frames); f(c->frame_cum);
    for (int k = 0; k < 2; ++k) {
        f(c->work.bx[k]); f(c->work.by[k]); f(c->work.bz[k]); f(c->work.bo[k]);
        f(c->work.sbx[k]); f(c->work.sby[k]); f(c->work.sbz[k]);
    }
    f(c->work.sboxes); f(c->work.stile_f32);
}
void release_buffers(m3d_cloud* c) {
    for_each_buffer(c, [](DevBuf& b) { b.release(); });
}
}  // namespace m3d

m3d::CloudView m3d_cloud::base_view() const {
    m3d::CloudView v;
    v.x = x.as<double>();
    v.y = y.as<double>();
    v.z = z.as<double>();
    v.nx = has_normals ? nx.as<double>() : nullptr;
    v.ny = has_normals ? ny.as<double>() : nullptr;
    v.nz = has_normals ? nz.as<double>() : nullptr;
    v.n = work.active ? n0 : n;
    v.n_pad = work.active ? n_pad0 : n_pad;
    return v;
}

m3d::CloudView m3d_cloud::view() const { return work.active && !work.cur_is_v0 ? work.cur : base_view(); }

m3d::SortedView m3d_cloud::sorted() const {
    if (work.active && !work.cur_is_v0) return work.scur;
    m3d::SortedView s;
    s.x = sx.as<double>();
    s.y = sy.as<double>();
    s.z = sz.as<double>();
    s.boxes = boxes.as<double>();
    s.tile_f32 = tile_f32.as<float>();
    if (frames_ready) {
        s.frames = frames.as<double>();
        s.frame_cum = frame_cum.as<uint16_t>();
    }
    s.n_tiles = n_tiles;
    s.max_abs = max_abs;
    for (int k = 0; k < 3; ++k) s.origin[k] = origin[k];
    s.radius = radius;
    return s;
}

using namespace m3d;

extern "C" {

const char* m3d_last_error(void) { return g_last_error.c_str(); }
const char* m3d_version(void) { return "misc3d_amd 0.1 (gfx950)"; }
int m3d_device_count(void) { return logical_device_count(); }

m3d_cloud* m3d_cloud_create(const double* xyz, const double* normals, size_t n, int device) {
    return m3d_cloud_create_impl(xyz, normals, n, device, /*with_sorted_copy=*/1);
}
// with_sorted_copy == 0: no Hilbert-sorted copy and no tile boxes -- what the registration, ICP and boundary entry points
// need of a resident cloud is its SoA arrays and its bounding box (they sort by their own grids); the fits need the copy
m3d_cloud* m3d_cloud_create_impl(const double* xyz, const double* normals, size_t n, int device, int with_sorted_copy) {
    LaneLock lane(device);
    if (!lane.ctx) return nullptr;
    return m3d_cloud_create_on(lane.ctx, xyz, normals, n, with_sorted_copy);
}
m3d_cloud* m3d_cloud_create_lane(const double* xyz, const double* normals, size_t n, int device, int lane) {
    if (lane < 0) return m3d_cloud_create(xyz, normals, n, device);
    if (lane >= lane_count()) {
        set_error("m3d_cloud_create_lane: lane " + std::to_string(lane) + " of " + std::to_string(lane_count()) + " (m3d_config.lanes)");
        return nullptr;
    }
    DeviceCtx* ctx = get_lane(device, lane);   // THIS lane, waited for if busy (a LaneLock would move on to a free one)
    if (!ctx) return nullptr;
    CtxLock lock(ctx);
    return m3d_cloud_create_on(ctx, xyz, normals, n, 1);
}

}  // extern "C"

using namespace m3d;
m3d_cloud* m3d_cloud_create_on(m3d::DeviceCtx* ctx, const double* xyz, const double* normals, size_t n, int with_sorted_copy) {
    if (!xyz && n > 0) {
        set_error("xyz is null");
        return nullptr;
    }
    if (n >= ((size_t)1 << 31)) {
        set_error("point clouds of 2^31 points or more are not supported");
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        set_error("hipSetDevice failed");
        return nullptr;
    }
    m3d_cloud* c = new m3d_cloud();
    c->ctx = ctx;
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    const bool phase_timing = config().kernel_timing != 0;
    auto mark = [&](int slot) {   // (phase clocks drain the stream: only on request)
        if (!phase_timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        c->setup_ms[slot] += std::chrono::duration<double, std::milli>(t - t_last).count();
        t_last = t;
    };
    c->n = (uint32_t)n;
    c->n_pad = std::max<uint32_t>(round_up((uint32_t)n, kScoreTile), kScoreTile);
    c->has_normals = normals != nullptr;
    const size_t bytes = sizeof(double) * (size_t)c->n_pad;
    DevBuf& stage = ctx->cc_stage;
    bool ok = c->x.reserve(bytes) && c->y.reserve(bytes) && c->z.reserve(bytes) &&
              stage.reserve(sizeof(double) * 3 * std::max<size_t>(n, 1));
    if (ok && c->has_normals) ok = c->nx.reserve(bytes) && c->ny.reserve(bytes) && c->nz.reserve(bytes);
    if (ok && n)
        ok = hipMemcpyAsync(stage.p, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream) ==
             hipSuccess;
    if (ok)
        launch_aos_to_soa(stage.as<double>(), c->x.as<double>(), c->y.as<double>(), c->z.as<double>(),
                          c->n, c->n_pad, ctx->stream);
    if (ok && c->has_normals) {
        // the staging buffer is reused: stream order keeps the first transpose ahead of this copy
        if (n)
            ok = hipMemcpyAsync(stage.p, normals, sizeof(double) * 3 * n, hipMemcpyHostToDevice,
                                ctx->stream) == hipSuccess;
        if (ok)
            launch_aos_to_soa(stage.as<double>(), c->nx.as<double>(), c->ny.as<double>(),
                              c->nz.as<double>(), c->n, c->n_pad, ctx->stream);
    }
    mark(1);
    // Hilbert-sorted copy + tile boxes for the culled scoring path.  The bounding box of the finite points comes
    // from the device copy (a host pass over the caller's 10 M-point array took 9 ms, as long as the rest of
    // the upload and sort together)
    DevBuf &t_cell = ctx->cc_cell, &t_start = ctx->cc_start, &t_fill = ctx->cc_fill, &t_sums = ctx->cc_sums,
           &t_total = ctx->cc_total, &t_bbox = ctx->cc_bbox;
    if (ok) {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t n_finite = 0;
        if (n) {
            double bb[7];
            ok = t_bbox.reserve(sizeof(double) * (kBboxPartialDoubles + 8));
            if (ok) {
                launch_bbox(c->x.as<double>(), c->y.as<double>(), c->z.as<double>(), c->n, t_bbox.as<double>(),
                            t_bbox.as<double>() + kBboxPartialDoubles, ctx->stream);
                ok = hipMemcpyAsync(bb, t_bbox.as<double>() + kBboxPartialDoubles, sizeof(bb), hipMemcpyDeviceToHost,
                                    ctx->stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->stream) == hipSuccess;
            }
            if (ok) {
                for (int k = 0; k < 3; ++k) {
                    lo[k] = bb[k];
                    hi[k] = bb[3 + k];
                }
                n_finite = (uint32_t)bb[6];
                if (n_finite) {
                    for (int k = 0; k < 6; ++k) c->bb[k] = bb[k];
                    c->bb_known = true;
                    double m = 0.0;
                    for (int k = 0; k < 3; ++k) m = std::max(m, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
                    c->max_abs = m;   // (stays +inf when something is off: the box tests then keep every tile)
                    double r = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        c->origin[k] = 0.5 * lo[k] + 0.5 * hi[k];
                        r = std::max(r, std::max(hi[k] - c->origin[k], c->origin[k] - lo[k]));
                    }
                    if (std::isfinite(r) && std::isfinite(c->origin[0]) && std::isfinite(c->origin[1]) && std::isfinite(c->origin[2]))
                        c->radius = r * (1.0 + 1e-12);   // (stays +inf otherwise: fp64 box tests)
                }
            }
        }
        mark(2);
        const uint32_t cap = std::max<uint32_t>(round_up((uint32_t)n, kTilePoints), kTilePoints);
        c->n_tiles = with_sorted_copy ? cap / kTilePoints : 0;
        c->n_sorted = n_finite;
        if (ok && !with_sorted_copy) n_finite = 0;   // (skips the sort below; c->n_sorted keeps the count)
        ok = ok && (!with_sorted_copy ||
                    (c->sx.reserve(sizeof(double) * cap) && c->sy.reserve(sizeof(double) * cap) &&
                     c->sz.reserve(sizeof(double) * cap) && c->boxes.reserve(sizeof(double) * kBoxStride * c->n_tiles) &&
                     c->tile_f32.reserve(sizeof(float) * kTileF32Floats * (size_t)c->n_tiles)));
        if (ok && with_sorted_copy) {
            launch_fill_nan(c->sx.as<double>(), cap, ctx->stream);
            launch_fill_nan(c->sy.as<double>(), cap, ctx->stream);
            launch_fill_nan(c->sz.as<double>(), cap, ctx->stream);
        }
        if (ok && n_finite) {
            double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
            if (!std::isfinite(ext)) ext = 0.0;  // absurdly large clouds: one cell, culling degenerates gracefully
            // about 8 points per cell, at most 2^8 cells per axis
            uint32_t bits = 1;
            while (bits < 8 && ((uint64_t)1 << (3 * bits)) * 8 < n_finite) ++bits;
            GridDesc gs;
            gs.K = 0;
            gs.morton_bits = bits | 0x100u;   // Hilbert order (plain Z-order, bits alone, measured slower: DESIGN.md)
            gs.nx = gs.ny = gs.nz = 1u << bits;
            gs.ox = lo[0];
            gs.oy = lo[1];
            gs.oz = lo[2];
            gs.inv_h = ext > 0.0 ? (double)(1u << bits) / (ext * (1.0 + 1e-9)) : 0.0;
            gs.r2 = gs.h2_in = 0.0;
            const uint32_t ncell = 1u << (3 * bits);
            ok = t_cell.reserve(sizeof(uint32_t) * n) && t_start.reserve(sizeof(uint32_t) * ((size_t)ncell + 1)) &&
                 t_fill.reserve(sizeof(uint32_t) * std::max<size_t>(n, 1)) &&   // rank of every point in its cell
                 t_sums.reserve(sizeof(uint32_t) * ((size_t)(ncell + 2047) / 2048 + 1)) && t_total.reserve(16);
            if (ok)
                launch_grid_build(c->view(), gs, t_cell.as<uint32_t>(), t_start.as<uint32_t>(), t_fill.as<uint32_t>(),
                                  t_sums.as<uint32_t>(), t_total.as<uint32_t>(), c->sx.as<double>(),
                                  c->sy.as<double>(), c->sz.as<double>(), ctx->stream);
        }
        mark(3);
        if (ok && with_sorted_copy) launch_tile_boxes(c->sorted(), c->boxes.as<double>(), ctx->stream);
    }
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    mark(4);
    c->setup_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (!ok) {
        if (g_last_error.empty()) set_error("cloud upload failed");
        release_buffers(c);
        delete c;
        return nullptr;
    }
    return c;
}
void m3d_cloud_destroy_on(m3d_cloud* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    release_buffers(c);   // (to the lane's free list: the next m3d_cloud_create takes them from there)
    delete c;
}
extern "C" {


void* m3d_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocPortable) != hipSuccess || !p) {
        set_error("hipHostMalloc failed (" + std::to_string(bytes) + " bytes)");
        return nullptr;
    }
    PinnedRegistry& r = pinned_registry();
    std::lock_guard<std::mutex> lock(r.mu);
    r.blocks.emplace_back(static_cast<const char*>(p), std::max<size_t>(bytes, 1));
    return p;
}
void m3d_host_free(void* p) {
    if (!p) return;
    {
        PinnedRegistry& r = pinned_registry();
        std::lock_guard<std::mutex> lock(r.mu);
        for (size_t i = 0; i < r.blocks.size(); ++i)
            if (r.blocks[i].first == p) {
                r.blocks.erase(r.blocks.begin() + (long)i);
                break;
            }
    }
    (void)hipHostFree(p);
}

void m3d_cloud_destroy(m3d_cloud* c) {
    if (!c) return;
    CtxLock lock(c->ctx);
    m3d_cloud_destroy_on(c);
}

void m3d_release_cached(int device) {
    if (!get_ctx(device)) return;
    for (int ln = 0; ln < kMaxLanes; ++ln) {   // every lane the device has (none is created here)
        DeviceCtx* ctx = find_lane(device, ln);
        if (!ctx) continue;
        CtxLock lock(ctx);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->cc_stage.release(); ctx->cc_cell.release(); ctx->cc_start.release(); ctx->cc_fill.release();
        ctx->cc_sums.release(); ctx->cc_total.release(); ctx->cc_bbox.release();
        if (ctx->seg_staging) m3d_host_free(ctx->seg_staging);
        ctx->seg_staging = nullptr;
        ctx->seg_staging_cap = 0;
    }
    dev_pool_trim(device);
}

size_t m3d_cloud_size(const m3d_cloud* c) { return c ? c->n : 0; }

size_t m3d_cloud_original_size(const m3d_cloud* c) { return c ? (c->work.active ? c->n0 : c->n) : 0; }

}  // extern "C"
