// K18: the one exchange step of the path -- a SUM all-reduce of gradient ranges over RCCL (xGMI), issued on a side HIP stream
// so that it overlaps with the backward that is still running on the compute stream.  Reference: DistributedDataParallel's
// gradient reduction, engine/processor.py:100-105 (find_unused_parameters=True); SURVEY.md 8(b) / 8(e).
//
// These four entry points are what a caller that binds the C ABI stage by stage uses; the Python
// host path (signal_amd/parallel/reducer.py) reaches the same RCCL through torch.distributed's 'nccl' backend instead, because
// the process group is the caller's (train.py creates it) and its store does the rendezvous.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already carries an RCCL (PyTorch-ROCm bundles one with
// the same soname) keeps using THAT copy -- linking a second one into this library would give the process two sets of the
// same symbols.  A single-GPU user never loads it.
#include <dlfcn.h>

#include <mutex>

#include "sig_kernels.h"

namespace {
typedef struct { char internal[128]; } UniqueId;                     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                                   // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kSum = 0;                                 // ncclFloat32, ncclSum (rccl.h)

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init = nullptr;
    AllReduceFn allreduce = nullptr;
    CommDestroyFn destroy = nullptr;
    GetErrorStringFn errstr = nullptr;
};
static char g_rccl_err[256] = "symbols missing";     // why the library could not be bound (captured ONCE: dlerror() clears itself)
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;                        // reached from the caller's and from autograd's thread
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            const char* e = dlerror();
            if (e) snprintf(g_rccl_err, sizeof(g_rccl_err), "%s", e);
        }
        if (r.h) {
            r.get_id = (GetUniqueIdFn)dlsym(r.h, "ncclGetUniqueId");
            r.init = (CommInitRankFn)dlsym(r.h, "ncclCommInitRank");
            r.allreduce = (AllReduceFn)dlsym(r.h, "ncclAllReduce");
            r.destroy = (CommDestroyFn)dlsym(r.h, "ncclCommDestroy");
            r.errstr = (GetErrorStringFn)dlsym(r.h, "ncclGetErrorString");
        }
    });
    return (r.h && r.get_id && r.init && r.allreduce && r.destroy) ? &r : nullptr;
}
}  // namespace

struct SigComm {
    Comm comm = nullptr;
    hipStream_t side = nullptr;       // the collective's own stream
    hipEvent_t ready = nullptr;       // compute -> side: the gradient range is final
    hipEvent_t done = nullptr;        // side -> compute: every all-reduce issued so far has finished
    int rank = 0, world = 1;
};

#define SIG_RCCL_CHECK(call, what)                                                                      \
    do {                                                                                                \
        const int rc_ = (call);                                                                         \
        if (rc_ != 0) {                                                                                 \
            sig_set_error("%s: RCCL error %d (%s)", what, rc_, r->errstr ? r->errstr(rc_) : "?");      \
            return 3;                                                                                   \
        }                                                                                               \
    } while (0)
#define SIG_HIP_CHECK(call, what)                                                   \
    do {                                                                            \
        const hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                                     \
            sig_set_error("%s: %s", what, hipGetErrorString(e_));                   \
            return 2;                                                               \
        }                                                                           \
    } while (0)

int sig_comm_unique_id_impl(void* id128) {
    SIG_CHECK_ARG(id128, "comm_unique_id: null pointer");
    Rccl* r = rccl();
    SIG_CHECK_ARG(r, "comm_unique_id: librccl.so.1 not loadable (%s)", g_rccl_err);
    SIG_RCCL_CHECK(r->get_id((UniqueId*)id128), "comm_unique_id");
    return 0;
}

int sig_comm_init_impl(SigComm** out, int rank, int world, const void* id128) {
    SIG_CHECK_ARG(out && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
    Rccl* r = rccl();
    SIG_CHECK_ARG(r, "comm_init: librccl.so.1 not loadable (%s)", g_rccl_err);
    SigComm* c = new SigComm();
    c->rank = rank; c->world = world;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    int rc = r->init(&c->comm, world, id, rank);
    if (rc != 0) {
        sig_set_error("comm_init (ncclCommInitRank): RCCL error %d (%s)", rc, r->errstr ? r->errstr(rc) : "?");
        c->comm = nullptr;
        (void)sig_comm_destroy_impl(c);
        return 3;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        sig_set_error("comm_init: side stream / events: %s", hipGetErrorString(hipGetLastError()));
        (void)sig_comm_destroy_impl(c);      // releases whatever was created
        return 2;
    }
    *out = c;
    return 0;
}

// buf[0..count) <- sum over ranks, in place, as soon as everything enqueued on `compute_stream` so far has finished.
int sig_comm_allreduce_async_impl(SigComm* c, float* buf, size_t count, hipStream_t compute_stream) {
    SIG_CHECK_ARG(c && c->comm && buf && count > 0, "comm_allreduce_async: bad arguments");
    Rccl* r = rccl();
    SIG_CHECK_ARG(r, "comm_allreduce_async: RCCL not loaded");
    SIG_HIP_CHECK(hipEventRecord(c->ready, compute_stream), "comm_allreduce_async: record");
    SIG_HIP_CHECK(hipStreamWaitEvent(c->side, c->ready, 0), "comm_allreduce_async: wait");
    SIG_RCCL_CHECK(r->allreduce(buf, buf, count, kFloat32, kSum, c->comm, c->side), "comm_allreduce_async (ncclAllReduce)");
    return 0;
}

// `stream` waits (on the device, no host sync) for every all-reduce issued so far.
int sig_comm_wait_impl(SigComm* c, hipStream_t stream) {
    SIG_CHECK_ARG(c && c->comm, "comm_wait: bad arguments");
    SIG_HIP_CHECK(hipEventRecord(c->done, c->side), "comm_wait: record");
    SIG_HIP_CHECK(hipStreamWaitEvent(stream, c->done, 0), "comm_wait: wait");
    return 0;
}

int sig_comm_destroy_impl(SigComm* c) {
    if (!c) return 0;
    Rccl* r = rccl();
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (r && c->comm) (void)r->destroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}
