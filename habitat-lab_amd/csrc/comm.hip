// comm.hip -- device-side collectives of the DD-PPO exchange: an RCCL communicator owned by the library, driven from inside the
// engine's forward / backward with no interpreter in between.
//
// Replaces what DistributedDataParallel's reducer does for the reference (habitat_baselines/rl/ddppo/algo/ddppo.py:128-140: buckets
// all-reduced from autograd hooks while backward is still running) and the two all_reduce calls of
// rl/ddppo/policy/running_mean_and_var.py:38-49.  Round 2-3 issued these through two ctypes callbacks into Python
// (rl/ddppo/ddppo.py `_tail_ready` / `_avg`): 2-4 interpreter round trips inside every backward and two inside every training
// forward, on each of the 8 ranks.  Here:
//   * hab_comm_create wraps ncclCommInitRank (the unique id travels through torch.distributed's store on the Python side, once);
//   * the communicator owns a HIP stream: when the engine reports a finished tail of the gradient arena it records an event on the
//     compute stream, makes the communicator stream wait for it and enqueues ncclAllReduce of that range there -- the exchange runs
//     beside the backward of the earlier layers (xGMI is point-to-point, ring all-reduce of 34-52 MB: ~0.4 ms, hidden);
//   * hab_policy_grad_sync enqueues the head of the arena and makes the compute stream wait for the communicator stream;
//   * the RunningMeanAndVar moments (9 + 8 floats) are all-reduced in order on the compute stream itself.
// librccl is bound with dlopen (the copy PyTorch already loaded when there is one): libhabitat_amd.so has no link-time dependency on
// it and loads on hosts without RCCL; hab_comm_available() says whether the symbols were found.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "engine.h"

namespace {

typedef struct { char internal[128]; } NcclUniqueId;  // rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void* NcclComm;
constexpr int kNcclFloat32 = 7, kNcclSum = 0;        // rccl.h: ncclDataType_t / ncclRedOp_t

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl load_rccl() {
    Rccl r;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process (PyTorch's), if any
        if (r.h) break;
    }
    if (!r.h)
        for (const char* n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
    if (!r.h) return r;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.h, "ncclAllReduce"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    return r;
}
const Rccl& rccl() {
    static const Rccl r = load_rccl();
    return r;
}

}  // namespace

struct hab_comm {
    NcclComm comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;        // the exchange runs here, beside the compute stream
    hipEvent_t ev_ready = nullptr;       // compute -> communicator: the range is final
    hipEvent_t ev_done = nullptr;        // communicator -> compute: every enqueued exchange has finished
};

extern "C" int hab_comm_available() { return rccl().ok ? 1 : 0; }

extern "C" int hab_comm_unique_id(uint8_t* out128) {
    if (!out128) return HAB_ERR_ARG;
    if (!rccl().ok) return HAB_ERR_UNSUPPORTED;
    NcclUniqueId id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != 0) return 1000 + rc;
    memcpy(out128, id.internal, 128);
    return HAB_OK;
}

extern "C" void hab_comm_destroy(hab_comm* c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int hab_comm_create(const uint8_t* id128, int world, int rank, hab_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return HAB_ERR_ARG;
    if (!rccl().ok) return HAB_ERR_UNSUPPORTED;
    hab_comm* c = new hab_comm();
    c->world = world; c->rank = rank;
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    int rc = rccl().CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) { c->comm = nullptr; hab_comm_destroy(c); return 1000 + rc; }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { hab_comm_destroy(c); return (int)e; }
    *out = c;
    return HAB_OK;
}

extern "C" int hab_comm_world_size(const hab_comm* c) { return c ? c->world : HAB_ERR_ARG; }

// in-place sum over the ranks, in order on `stream`
extern "C" int hab_comm_allreduce_sum(hab_comm* c, float* buf, int64_t count, hipStream_t stream) {
    if (!c || !c->comm || !buf || count <= 0) return HAB_ERR_ARG;
    const int rc = rccl().AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, stream);
    return rc == 0 ? HAB_OK : 1000 + rc;
}

// ---- engine side -------------------------------------------------------------------------------------------------------------
// [first, first + count) of `buf` is final on `compute`: exchange it on the communicator's stream
int comm_exchange_async(hab_comm* c, float* buf, int64_t first, int64_t count, hipStream_t compute) {
    if (count <= 0) return HAB_OK;
    hipError_t e = hipEventRecord(c->ev_ready, compute);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_ready, 0);
    if (e != hipSuccess) return (int)e;
    return hab_comm_allreduce_sum(c, buf + first, count, c->stream);
}
// everything exchanged so far is visible to what `compute` runs next
int comm_join(hab_comm* c, hipStream_t compute) {
    hipError_t e = hipEventRecord(c->ev_done, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(compute, c->ev_done, 0);
    return e == hipSuccess ? HAB_OK : (int)e;
}

extern "C" int hab_policy_set_comm(hab_policy* e, hab_comm* c) {
    if (!e) return HAB_ERR_ARG;
    e->comm = c;
    e->comm_first = -1;
    e->comm_err = HAB_OK;
    // cleared: back to one rank until hab_policy_set_allreduce installs the callback form with its own world size (the RunningMeanAndVar
    // divisor must not keep the old communicator's)
    e->world_size = c ? c->world : 1;
    return HAB_OK;
}

// After hab_policy_backward: exchange whatever part of the gradient arena the backward did not report as a finished tail (the head: the
// early convolution layers; or everything, for a backward without reports) and make `stream` wait for all of it.  The sums over ranks
// are in the arena afterwards; the 1 / world_size is folded into the fused clip + Adam step (hab_clip_adam_step grad_scale).
extern "C" int hab_policy_grad_sync(hab_policy* e, hipStream_t stream) {
    if (!e || !e->G) return HAB_ERR_ARG;
    if (!e->comm) return HAB_ERR_UNSUPPORTED;
    if (e->comm_err != HAB_OK) { const int rc = e->comm_err; e->comm_err = HAB_OK; e->comm_first = -1; return rc; }
    const int64_t head = e->comm_first < 0 ? e->param_floats : e->comm_first;
    e->comm_first = -1;
    HAB_TRY(comm_exchange_async(e->comm, e->G, 0, head, stream));
    return comm_join(e->comm, stream);
}
