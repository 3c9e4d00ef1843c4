// gcpx_comm_*: the collectives of the data-parallel paths behind the C boundary: the gradient all-reduce of
// the training step (the reference: nn.DataParallel's reduce, /root/reference/gcp/prediction/training/gcp_builder.py:71-78) and the
// all-gather of the CEM candidates' costs (gcp/planning/run.py:108-120 spreads trajectories over GPUs; here one population is sharded).
// One communicator per process (one process per GPU), RCCL over xGMI, enqueued on the caller's stream; nothing here synchronises.
// The Python host (video-gcp_amd/dist.py) reaches the same RCCL through torch.distributed; these exports are what an operator-level
// binder without torch calls.  RCCL is resolved at run time (dlopen): the library has no link-time dependency on it, and a process that
// already holds a copy (torch ships one) shares it.
#include "common.h"

#include <cstring>
#include <string>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// Resolved once, thread-safely (function-local static initialised by a lambda).  A copy of RCCL the process already holds — torch's
// bundled one, under whatever path or soname — is found first through the global symbol scope (dlsym(RTLD_DEFAULT)) and through
// RTLD_NOLOAD, so that two copies never serve one process; only then is a new one loaded.
struct RcclLoad {
    Rccl r;
    bool ok = false;
    std::string why;
};

const RcclLoad& rccl_load() {
    static const RcclLoad load = []() {
        RcclLoad l;
        Rccl& r = l.r;
        auto bind = [&](void* h) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
            r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
            return r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
        };
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce") && bind(RTLD_DEFAULT)) {           // already in the process (e.g. torch's copy)
            l.ok = true;
            return l;
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const int flags : {RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD, RTLD_NOW | RTLD_GLOBAL}) {
            for (const char* name : names) {
                r.handle = dlopen(name, flags);
                if (r.handle) break;
            }
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();                   // (NULL when the last failure was already reported)
            l.why = e ? e : "unknown dlopen error";
            return l;
        }
        l.ok = bind(r.handle);
        if (!l.ok) l.why = "missing symbol";
        return l;
    }();
    return load;
}

Rccl* rccl() {
    const RcclLoad& l = rccl_load();
    if (!l.ok) gcpx_set_error("gcpx_comm: librccl.so not found or incomplete (%s)", l.why.c_str());
    return l.ok ? const_cast<Rccl*>(&l.r) : nullptr;
}

int fail(Rccl* r, const char* what, ncclResult_t st) {
    gcpx_set_error("gcpx_comm: %s: %s", what, r->GetErrorString ? r->GetErrorString(st) : "RCCL error");
    return GCPX_ERR_COMM;
}

}  // namespace

extern "C" int gcpx_comm_unique_id(void* id_out) {
    GCPX_CHECK_ARG(id_out, "null pointer");
    Rccl* r = rccl();
    if (!r) return GCPX_ERR_COMM;
    ncclUniqueId id;
    const ncclResult_t st = r->GetUniqueId(&id);
    if (st != ncclSuccess) return fail(r, "ncclGetUniqueId", st);
    memcpy(id_out, &id, sizeof(id));
    return GCPX_OK;
}

extern "C" int gcpx_comm_init(void** comm, int32_t rank, int32_t world, const void* id) {
    GCPX_CHECK_ARG(comm && id && world > 0 && rank >= 0 && rank < world, "bad arguments");
    static_assert(sizeof(ncclUniqueId) == GCPX_COMM_ID_BYTES, "GCPX_COMM_ID_BYTES");
    Rccl* r = rccl();
    if (!r) return GCPX_ERR_COMM;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    const ncclResult_t st = r->CommInitRank(&c, world, uid, rank);
    if (st != ncclSuccess) return fail(r, "ncclCommInitRank", st);
    *comm = c;
    return GCPX_OK;
}

extern "C" int gcpx_comm_allreduce(void* comm, float* buf, int64_t n, void* stream) {
    GCPX_CHECK_ARG(comm && buf && n >= 0, "bad arguments");
    Rccl* r = rccl();
    if (!r) return GCPX_ERR_COMM;
    if (n == 0) return GCPX_OK;
    const ncclResult_t st = r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream));
    return st == ncclSuccess ? GCPX_OK : fail(r, "ncclAllReduce", st);
}

extern "C" int gcpx_comm_allgather(void* comm, const float* send, float* recv, int64_t n, void* stream) {
    GCPX_CHECK_ARG(comm && send && recv && n >= 0, "bad arguments");
    Rccl* r = rccl();
    if (!r) return GCPX_ERR_COMM;
    if (n == 0) return GCPX_OK;
    const ncclResult_t st = r->AllGather(send, recv, (size_t)n, ncclFloat32, static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream));
    return st == ncclSuccess ? GCPX_OK : fail(r, "ncclAllGather", st);
}

extern "C" int gcpx_comm_destroy(void* comm) {
    GCPX_CHECK_ARG(comm, "null communicator");
    Rccl* r = rccl();
    if (!r) return GCPX_ERR_COMM;
    const ncclResult_t st = r->CommDestroy(static_cast<ncclComm_t>(comm));
    return st == ncclSuccess ? GCPX_OK : fail(r, "ncclCommDestroy", st);
}
