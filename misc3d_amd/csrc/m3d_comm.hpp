// m3d_comm.hpp -- the communicator of the hypothesis-sharded entry points.
//
// The reference's hypothesis loop (include/misc3d/common/ransac.h:571-613) has ONE cross-iteration dependency,
// the best-model update.  Sharded over GPUs that becomes ONE exchange per window of hypotheses: every rank
// contributes the 4-byte (valid << 31 | inlier count) records of its slice and receives everybody's, then
// every rank replays the same sequence.  Three transports behind one interface:
//   RCCL    one process per GPU (torchrun / mpirun ...): ncclAllGather on the library's own stream, in place on the
//           device record array -- no host bounce between the scoring kernels and the replay's copy.  librccl is
//           bound at run time (dlopen), so the library loads on hosts without it.
//   HOST    a caller-supplied all-gather over host buffers (MPI, gloo, a test harness).
//   LOCAL   several devices driven by threads of ONE process (the `devices[]` form of the C++ API): a
//           rendezvous through host memory.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/misc3d_amd.h"
#include "m3d_driver.hpp"

namespace m3d {

// rendezvous of the LOCAL transport: n ranks = n threads of one process
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<uint8_t> buf[2];   // world x bytes_per_rank of the exchange in flight (by generation parity)
    size_t bytes_per_rank = 0;
    bool aborted = false;          // a rank failed outside an exchange: the others must not wait for it
    int first_failed = -1;         // ... and which one gave up first (its error is the call's; the others' is "another rank failed")
    void abort(int rank = -1) {
        std::lock_guard<std::mutex> lock(mu);
        if (!aborted) first_failed = rank;
        aborted = true;
        cv.notify_all();
    }
};

}  // namespace m3d

struct m3d_comm {
    enum Transport { kRccl = 0, kHost = 1, kLocal = 2 };
    Transport transport = kHost;
    int world = 1, rank = 0;
    int device = -1;               // RCCL: the device the communicator was created on
    void* nccl = nullptr;          // ncclComm_t
    m3d_allgather_fn host_fn = nullptr;
    void* host_user = nullptr;
    std::shared_ptr<m3d::LocalGroup> local;
    uint64_t collectives = 0;      // exchanges so far (statistics)
    m3d::DevBuf stage;             // RCCL: device staging of host-side exchanges
    m3d::PinBuf h_stage;

    // all-gather of `bytes` bytes per rank between HOST buffers (recv: world x bytes, rank-major).
    // RCCL stages through the device on `st` and waits for it.
    int allgather_host(const void* send, void* recv, size_t bytes, hipStream_t st);
    // in-place all-gather of 32-bit records on the DEVICE: buf holds world x count words, this rank's slice at
    // [rank * count, (rank + 1) * count).  RCCL: enqueued on `st`, no host wait.  HOST / LOCAL: `host_scratch`
    // (pinned, world x count words) receives the gathered records, the stream is waited for, the gathered array is
    // uploaded back into buf; *host_has_all = 1 tells the caller host_scratch already holds everything.
    int allgather_u32_device(uint32_t* buf, size_t count, hipStream_t st, uint32_t* host_scratch, int* host_has_all);
};
