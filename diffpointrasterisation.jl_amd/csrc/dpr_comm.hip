// dpr_comm_* / dpr_raster_pullback_sharded_*: the multi-GPU exchange of the batched pullback
// behind the C ABI, so that a non-Python host (the Julia extension) can shard poses over the
// GPUs of a node: one process (or task) per GPU, RCCL over xGMI.
//
// The reference's in-process analogue is the threaded batched pullback: pose chunks with private
// ds_dpoints / ds_dpoint_weight slabs and a final sum (/root/reference/src/raster_pullback.jl:112-147).
// Here a chunk is a rank: every per-pose output is disjoint across ranks, only the point
// gradients sum over poses -- ONE all-reduce(sum).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library a PyTorch-ROCm process has
// already loaded, or the system one): libdpr.so itself has no link-time dependency on it, and
// single-GPU users never load it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include <rccl/rccl.h>

#include "../../include/dpr.h"
#include "dpr_tiled.h"

struct dpr_comm {
    ncclComm_t comm;
    int world, rank;
};

namespace dpr {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
    bool ok;
    char why[256];  // why the binding failed (dlopen's message, captured once)
};

static const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a{};
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            const char* e = dlerror();  // one call: it clears the message
            snprintf(a.why, sizeof(a.why), "%s", e ? e : "dlopen failed");
            return a;
        }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GroupStart &&
               a.GroupEnd && a.GetErrorString;
        if (!a.ok) snprintf(a.why, sizeof(a.why), "a required nccl* symbol is missing");
        return a;
    }();
    return api;
}

static int need_rccl() {
    if (!rccl().ok)
        return fail(DPR_ERR_HIP, "RCCL (librccl.so.1) could not be loaded: %s", rccl().why);
    return DPR_OK;
}

#define DPR_NCCL(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess)                                                            \
            return dpr::fail(DPR_ERR_HIP, "%s failed: %s", #expr,                            \
                             dpr::rccl().GetErrorString(r_));                              \
    } while (0)


// The exchange step of dpr_raster_pullback_sharded_*: ONE all-reduce(sum) when the caller fused
// [ds_dpoints | ds_dpoint_weight] into one buffer, else a group of two.
//
// A rank whose LOCAL pullback failed (rc_local != 0: e.g. a workspace sized for another rank's
// B_local) must not simply return: its peers are already inside -- or on their way into -- the
// collective and would block forever.  Defined behaviour: when the rank's two gradient buffers
// are usable (non-NULL, P > 0) it fills them with NaN (all-ones bytes), JOINS the all-reduce --
// every rank then sees NaN point gradients, loud instead of silently short of one rank's poses --
// and returns its own error code afterwards.  Only when the buffers themselves are unusable
// (NULL) can the rank not take part; the error text then says that the communicator has to be
// torn down (dpr_comm_destroy on every rank) because the peers are blocked.
template <typename T>
static int sharded_exchange(dpr_comm_t* comm, hipStream_t st, int rc_local, int n_in, int64_t P,
                            T* ds_dpoints, T* ds_dpoint_weight, ncclDataType_t dt) {
    if (P <= 0) return rc_local;  // nothing to exchange (every rank sees the same P)
    std::string local_msg;
    if (rc_local != DPR_OK) {
        local_msg = dpr_last_error();
        if (!ds_dpoints || !ds_dpoint_weight)
            return fail(rc_local,
                        "%s -- this rank cannot join the all-reduce (gradient buffers are NULL): "
                        "the other ranks block in it; destroy the communicator on every rank",
                        local_msg.c_str());
        // The NaN fill is sized by n_in: an argument error may be ABOUT n_in, so it is validated
        // here before it sizes a write (the library accepts 1 <= n_in <= 3).  A rank with an
        // unusable n_in cannot know how large its peers' buffers are: it cannot join.
        if (n_in < 1 || n_in > 3)
            return fail(rc_local,
                        "%s -- this rank cannot join the all-reduce (n_in = %d does not size the "
                        "gradient buffer): the other ranks block in it; destroy the communicator "
                        "on every rank", local_msg.c_str(), n_in);
        const hipError_t e1 = hipMemsetAsync(ds_dpoints, 0xff, sizeof(T) * (size_t)P * n_in, st);
        const hipError_t e2 = hipMemsetAsync(ds_dpoint_weight, 0xff, sizeof(T) * (size_t)P, st);
        if (e1 != hipSuccess || e2 != hipSuccess)  // joins all the same (peers must not block); said below
            local_msg += std::string(" [the NaN fill of the gradient buffers failed too: ") +
                         hipGetErrorString(e1 != hipSuccess ? e1 : e2) + "]";
    }
    ncclResult_t r = ncclSuccess;
    if (ds_dpoint_weight == ds_dpoints + (size_t)P * n_in) {
        r = rccl().AllReduce(ds_dpoints, ds_dpoints, (size_t)P * (n_in + 1), dt, ncclSum,
                             comm->comm, st);
    } else {
        r = rccl().GroupStart();
        if (r == ncclSuccess) {
            // an open group is always closed, whatever the calls inside return
            const ncclResult_t r1 = rccl().AllReduce(ds_dpoints, ds_dpoints, (size_t)P * n_in, dt,
                                                     ncclSum, comm->comm, st);
            const ncclResult_t r2 = rccl().AllReduce(ds_dpoint_weight, ds_dpoint_weight, (size_t)P,
                                                     dt, ncclSum, comm->comm, st);
            const ncclResult_t r3 = rccl().GroupEnd();
            r = r1 != ncclSuccess ? r1 : (r2 != ncclSuccess ? r2 : r3);
        }
    }
    if (rc_local != DPR_OK)
        return fail(rc_local, "%s (this rank joined the all-reduce with NaN gradients)",
                    local_msg.c_str());
    if (r != ncclSuccess)
        return fail(DPR_ERR_HIP, "all-reduce of the point gradients failed: %s",
                    rccl().GetErrorString(r));
    return DPR_OK;
}

}  // namespace dpr


extern "C" {

int dpr_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < DPR_COMM_ID_BYTES)
        return dpr::fail(DPR_ERR_INVALID_ARG, "dpr_comm_unique_id: need a %d-byte buffer",
                         DPR_COMM_ID_BYTES);
    if (int rc = dpr::need_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == DPR_COMM_ID_BYTES, "id size");
    DPR_NCCL(dpr::rccl().GetUniqueId((ncclUniqueId*)id_out));
    return DPR_OK;
}

int dpr_comm_init(dpr_comm_t** comm_out, int world, int rank, const void* id) {
    if (!comm_out || !id || world < 1 || rank < 0 || rank >= world)
        return dpr::fail(DPR_ERR_INVALID_ARG, "dpr_comm_init: bad arguments (world %d, rank %d)",
                         world, rank);
    if (int rc = dpr::need_rccl()) return rc;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    DPR_NCCL(dpr::rccl().CommInitRank(&c, world, uid, rank));  // binds the current HIP device
    *comm_out = new dpr_comm{c, world, rank};
    return DPR_OK;
}

int dpr_comm_destroy(dpr_comm_t* comm) {
    if (!comm) return DPR_OK;
    if (int rc = dpr::need_rccl()) return rc;
    ncclResult_t r = dpr::rccl().CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess)
        return dpr::fail(DPR_ERR_HIP, "ncclCommDestroy failed: %s", dpr::rccl().GetErrorString(r));
    return DPR_OK;
}

int dpr_comm_world(const dpr_comm_t* comm) { return comm ? comm->world : 0; }
int dpr_comm_rank(const dpr_comm_t* comm) { return comm ? comm->rank : -1; }

void dpr_shard_range(int64_t batch, int rank, int world, int64_t* lo, int64_t* hi) {
    // contiguous pose blocks whose sizes differ by at most one (ChunkSplitters.chunks,
    // src/raster_pullback.jl:117)
    if (world < 1 || rank < 0 || rank >= world || batch < 0) {  // no valid block: an empty range
        if (lo) *lo = 0;
        if (hi) *hi = 0;
        return;
    }
    const int64_t base = batch / world, rem = batch % world;
    const int64_t l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

#define DPR_DEFINE_SHARDED(SUF, T, NCCLT)                                                         \
    int dpr_raster_pullback_sharded_##SUF(                                                        \
        dpr_comm_t* comm, void* stream, int n_in, int n_out, const int64_t* grid, int64_t P,      \
        int64_t B_local, const T* ds_dout_local, const T* points, const T* rotation_local,        \
        const T* translation_local, const T* out_weight_local, const T* point_weight,             \
        T* ds_dpoints, T* ds_drotation_local, T* ds_dtranslation_local, T* ds_dbackground_local,  \
        T* ds_dout_weight_local, T* ds_dpoint_weight, void* workspace, size_t workspace_bytes) {  \
        if (!comm) return dpr::fail(DPR_ERR_INVALID_ARG, "dpr_raster_pullback_sharded: comm is NULL"); \
        if (int rc = dpr::need_rccl()) return rc;                                                 \
        const int rc_local = dpr_raster_pullback_##SUF(                                           \
            stream, n_in, n_out, grid, P, B_local, ds_dout_local, points, rotation_local,         \
            translation_local, out_weight_local, point_weight, ds_dpoints, ds_drotation_local,    \
            ds_dtranslation_local, ds_dbackground_local, ds_dout_weight_local, ds_dpoint_weight,  \
            workspace, workspace_bytes);                                                          \
        return dpr::sharded_exchange<T>(comm, (hipStream_t)stream, rc_local, n_in, P, ds_dpoints, \
                                        ds_dpoint_weight, NCCLT);                                 \
    }
DPR_DEFINE_SHARDED(f32, float, ncclFloat)
DPR_DEFINE_SHARDED(f64, double, ncclDouble)
#undef DPR_DEFINE_SHARDED

}  // extern "C"
