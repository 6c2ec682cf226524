// m3d_comm.cpp -- transports of the hypothesis-sharded entry points (see m3d_comm.hpp).
#include "m3d_comm.hpp"

#include <dlfcn.h>
// types and enums only: the entry points are bound with dlsym (no link-time dependency).  A ROCm installation without
// the RCCL development headers still builds the library: the handful of declarations this file needs are ABI-stable
// (nccl.h: 128-byte unique id, ncclSuccess == 0, ncclUint8 == 1, ncclUint32 == 3)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;
}
#endif

#include <cstdlib>
#include <cstring>
#include <string>

namespace m3d {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// librccl is bound once, on first use.  The library's own HIP runtime is the one it was linked against
// (libamdhip64.so.7 of the ROCm installation); a host process may ALSO carry a second, privately bundled RCCL +
// HIP pair (PyTorch wheels do) whose streams and pointers are not interchangeable with ours, so the ROCm
// installation's librccl is asked for by path first and bound with RTLD_LOCAL | RTLD_DEEPBIND: its symbols never
// mix with a bundled copy's.  M3D_RCCL_PATH overrides the search.
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* e = std::getenv("M3D_RCCL_PATH")) names.emplace_back(e);
        if (const char* r = std::getenv("ROCM_PATH")) names.emplace_back(std::string(r) + "/lib/librccl.so.1");
        names.emplace_back("/opt/rocm/lib/librccl.so.1");
        names.emplace_back("librccl.so.1");
        names.emplace_back("librccl.so");
        for (const auto& n : names) {
            api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (api.handle) break;
            if (const char* de = dlerror()) api.error = de;
        }
        if (!api.handle) {
            api.error = "librccl not found (" + api.error + ")";
            return;
        }
        auto sym = [&](const char* s) -> void* {
            void* p = dlsym(api.handle, s);
            if (!p) api.error = std::string("librccl lacks ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
            dlclose(api.handle);
            api.handle = nullptr;
            return;
        }
        // the streams and device pointers this library hands to ncclAllGather belong to the HIP runtime IT is linked
        // against; a librccl whose own HIP dependency resolved to another copy of libamdhip64 (a host process can hold
        // two: PyTorch wheels bundle theirs) would take them for foreign handles.  Compare the two runtimes' load
        // addresses and refuse the binding when they differ -- the callers then fall back to the host transport.
        if (void* theirs = dlsym(api.handle, "hipMalloc")) {
            Dl_info di_theirs{}, di_ours{};
            void* ours = reinterpret_cast<void*>(static_cast<hipError_t (*)(void**, size_t)>(&hipMalloc));
            if (dladdr(theirs, &di_theirs) && dladdr(ours, &di_ours) && di_theirs.dli_fbase != di_ours.dli_fbase) {
                api.error = std::string("librccl is bound to another HIP runtime (") + (di_theirs.dli_fname ? di_theirs.dli_fname : "?") +
                            ") than this library (" + (di_ours.dli_fname ? di_ours.dli_fname : "?") + ")";
                dlclose(api.handle);
                api.handle = nullptr;
            }
        }
    });
    return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
    RcclApi& a = rccl();
    return fail(M3D_ERR_DEVICE, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}

}  // namespace
}  // namespace m3d

using namespace m3d;

int m3d_comm::allgather_host(const void* send, void* recv, size_t bytes, hipStream_t st) {
    collectives++;
    if (bytes == 0) return M3D_OK;
    if (transport == kHost) {
        const int rc = host_fn(host_user, send, recv, bytes);
        return rc == 0 ? M3D_OK : fail(M3D_ERR_DEVICE, "the caller's all-gather reported failure " + std::to_string(rc));
    }
    if (transport == kLocal) {
        LocalGroup& g = *local;
        std::unique_lock<std::mutex> lock(g.mu);
        if (g.aborted) return fail(M3D_ERR_DEVICE, "another rank of the local group failed");
        const uint64_t gen = g.generation;
        // two buffers, by generation parity: a rank that is already back for exchange gen + 1 writes into the other
        // buffer while laggards still copy exchange gen out (nobody can be two exchanges ahead: gen + 1 cannot
        // complete before every rank has arrived for it)
        std::vector<uint8_t>& buf = g.buf[gen & 1];
        if (g.arrived == 0) {
            g.bytes_per_rank = bytes;
            buf.resize((size_t)world * bytes);
        }
        if (g.bytes_per_rank != bytes) return fail(M3D_ERR_INTERNAL, "ranks disagree on the size of an exchange");
        std::memcpy(buf.data() + (size_t)rank * bytes, send, bytes);
        if (++g.arrived == world) {
            g.arrived = 0;
            g.generation++;
            g.cv.notify_all();
        } else {
            g.cv.wait(lock, [&] { return g.generation != gen || g.aborted; });
            if (g.generation == gen) return fail(M3D_ERR_DEVICE, "another rank of the local group failed");
        }
        std::memcpy(recv, buf.data(), (size_t)world * bytes);
        return M3D_OK;
    }
    // RCCL: host -> device -> all-gather -> host
    RcclApi& a = rccl();
    if (!a.handle || !nccl) return fail(M3D_ERR_DEVICE, "RCCL communicator is not usable: " + a.error);
    if (!stage.reserve((size_t)(world + 1) * bytes) || !h_stage.reserve((size_t)(world + 1) * bytes)) return M3D_ERR_DEVICE;
    uint8_t* d = stage.as<uint8_t>();
    uint8_t* h = h_stage.as<uint8_t>();
    std::memcpy(h, send, bytes);
    if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return fail(M3D_ERR_DEVICE, "hipMemcpyAsync failed");
    const ncclResult_t r = a.AllGather(d, d + bytes, bytes, ncclUint8, static_cast<ncclComm_t>(nccl), st);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    if (hipMemcpyAsync(h + bytes, d + bytes, (size_t)world * bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(M3D_ERR_DEVICE, "copy after ncclAllGather failed");
    std::memcpy(recv, h + bytes, (size_t)world * bytes);
    return M3D_OK;
}

int m3d_comm::allgather_u32_device(uint32_t* buf, size_t count, hipStream_t st, uint32_t* host_scratch,
                                   int* host_has_all) {
    *host_has_all = 0;
    if (transport == kRccl) {
        collectives++;
        RcclApi& a = rccl();
        if (!a.handle || !nccl) return fail(M3D_ERR_DEVICE, "RCCL communicator is not usable: " + a.error);
        // in place: this rank's slice already sits at buf + rank * count
        const ncclResult_t r = a.AllGather(buf + (size_t)rank * count, buf, count, ncclUint32, static_cast<ncclComm_t>(nccl), st);
        if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
        return M3D_OK;
    }
    // host transports: own slice down, exchange, everything up again (the kernels that follow read the gathered array)
    uint32_t* mine = host_scratch + (size_t)rank * count;
    if (hipMemcpyAsync(mine, buf + (size_t)rank * count, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(M3D_ERR_DEVICE, "record download failed");
    std::vector<uint32_t> own(mine, mine + count);   // (recv may alias send otherwise)
    const int rc = allgather_host(own.data(), host_scratch, sizeof(uint32_t) * count, st);
    if (rc != M3D_OK) return rc;
    if (hipMemcpyAsync(buf, host_scratch, sizeof(uint32_t) * count * (size_t)world, hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(M3D_ERR_DEVICE, "record upload failed");
    *host_has_all = 1;
    return M3D_OK;
}

extern "C" {

int m3d_comm_unique_id(uint8_t id[M3D_COMM_ID_BYTES]) {
    static_assert(M3D_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as ncclUniqueId");
    if (!id) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    RcclApi& a = rccl();
    if (!a.handle) return fail(M3D_ERR_DEVICE, a.error);
    ncclUniqueId u;
    const ncclResult_t r = a.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return M3D_OK;
}

m3d_comm* m3d_comm_create_rccl(const uint8_t id[M3D_COMM_ID_BYTES], int world, int rank, int device) {
    if (!id || world < 1 || rank < 0 || rank >= world) {
        set_error("m3d_comm_create_rccl: invalid argument");
        return nullptr;
    }
    RcclApi& a = rccl();
    if (!a.handle) {
        set_error(a.error);
        return nullptr;
    }
    DeviceCtx* ctx = get_ctx(device);   // (checks the ordinal, creates the library stream)
    if (!ctx) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) {
        set_error("hipSetDevice failed");
        return nullptr;
    }
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = a.CommInitRank(&c, world, u, rank);
    if (r != ncclSuccess) {
        (void)rccl_fail("ncclCommInitRank", r);
        return nullptr;
    }
    m3d_comm* q = new m3d_comm();
    q->transport = m3d_comm::kRccl;
    q->world = world;
    q->rank = rank;
    q->device = device;
    q->nccl = c;
    return q;
}

m3d_comm* m3d_comm_create_host(int world, int rank, m3d_allgather_fn fn, void* user) {
    if (world < 1 || rank < 0 || rank >= world || !fn) {
        set_error("m3d_comm_create_host: invalid argument");
        return nullptr;
    }
    m3d_comm* q = new m3d_comm();
    q->transport = m3d_comm::kHost;
    q->world = world;
    q->rank = rank;
    q->host_fn = fn;
    q->host_user = user;
    return q;
}

int m3d_comm_create_local(int world, m3d_comm** comms) {
    if (world < 1 || !comms) return fail(M3D_ERR_INVALID_ARG, "invalid argument");
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    for (int r = 0; r < world; ++r) {
        m3d_comm* q = new m3d_comm();
        q->transport = m3d_comm::kLocal;
        q->world = world;
        q->rank = r;
        q->local = g;
        comms[r] = q;
    }
    return M3D_OK;
}

void m3d_comm_destroy(m3d_comm* q) {
    if (!q) return;
    if (q->transport == m3d_comm::kRccl && q->nccl) {
        RcclApi& a = rccl();
        if (q->device >= 0) {
            const int phys = physical_device(q->device);   // (q->device: the ordinal the caller used)
            if (phys >= 0) (void)hipSetDevice(phys);
        }
        if (a.handle) (void)a.CommDestroy(static_cast<ncclComm_t>(q->nccl));
    }
    q->stage.release();
    q->h_stage.release();
    delete q;
}

int m3d_comm_world(const m3d_comm* q) { return q ? q->world : 1; }
int m3d_comm_rank(const m3d_comm* q) { return q ? q->rank : 0; }
uint64_t m3d_comm_collectives(const m3d_comm* q) { return q ? q->collectives : 0; }

}  // extern "C"
