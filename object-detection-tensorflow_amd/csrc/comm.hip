// Collectives of the C-ABI (SURVEY.md 8b's proposed export list: odtk_comm_{init,allreduce,destroy}; 8e): the gradient sum of the data-parallel
// step for binders that are NOT PyTorch (the package's own dist.py can use it too: collective='odtk').  A thin layer over RCCL -- ring / tree
// all-reduce over the xGMI links is RCCL's job; nothing here re-implements it.  RCCL is bound at FIRST USE with dlopen, so libodtk.so keeps loading
// (and the single-device path keeps working) where no RCCL is installed; inside a PyTorch process the dlopen resolves to the RCCL torch already
// loaded (same soname).  The reference has no counterpart: it is single-device (testSSD300.py:14).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only: no RCCL symbol is linked

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/odtk.h"
#include "common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";          // dlopen's message for librccl.so.1, captured once (dlerror() is consumed by the call that reads it)
};

Rccl g_rccl;
std::once_flag g_once;

void bind_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
        if (n == names[0]) {
            const char* e = dlerror();
            snprintf(g_rccl.why, sizeof g_rccl.why, "%s", e ? e : "dlopen failed");
        }
    }
    if (!g_rccl.lib) return;
    snprintf(g_rccl.why, sizeof g_rccl.why, "symbols missing");
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.broadcast = (decltype(g_rccl.broadcast))dlsym(g_rccl.lib, "ncclBroadcast");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.error_string = (decltype(g_rccl.error_string))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.broadcast && g_rccl.comm_destroy && g_rccl.error_string;
}

int need_rccl() {
    std::call_once(g_once, bind_rccl);
    if (!g_rccl.ok) {
        odtk::set_error("odtk_comm: RCCL not available (dlopen librccl.so.1: %s)", g_rccl.why);
        return ODTK_ERR_HIP;
    }
    return ODTK_OK;
}

#define ODTK_CHECK_RCCL(expr)                                                                                  \
    do {                                                                                                       \
        const ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) {                                                                               \
            odtk::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.error_string(r_));           \
            return ODTK_ERR_HIP;                                                                               \
        }                                                                                                      \
    } while (0)

}  // namespace

struct odtk_comm {
    ncclComm_t comm;
    int rank, world, device;
};

extern "C" int odtk_comm_unique_id(void* id128) {
    ODTK_REQUIRE(id128, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == ODTK_COMM_ID_BYTES, "ODTK_COMM_ID_BYTES");
    if (int e = need_rccl()) return e;
    ODTK_CHECK_RCCL(g_rccl.get_unique_id((ncclUniqueId*)id128));
    return ODTK_OK;
}

extern "C" int odtk_comm_init(const void* id128, int rank, int world, odtk_comm** out) {
    ODTK_REQUIRE(id128 && out, "comm_init: null pointer");
    ODTK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    if (int e = need_rccl()) return e;
    int dev = -1;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ODTK_CHECK_RCCL(g_rccl.comm_init_rank(&c, world, id, rank));
    *out = new odtk_comm{c, rank, world, dev};
    return ODTK_OK;
}

extern "C" int odtk_comm_info(const odtk_comm* comm, int* rank, int* world) {
    ODTK_REQUIRE(comm, "comm_info: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return ODTK_OK;
}

// the communicator belongs to the device that was current in odtk_comm_init; a collective enqueued with another device current would launch on the wrong GPU
static int on_comm_device(const odtk_comm* comm, const char* what) {
    int dev = -1;
    ODTK_CHECK_HIP(hipGetDevice(&dev));
    ODTK_REQUIRE(dev == comm->device, "%s: current device %d, the communicator was created on device %d", what, dev, comm->device);
    return ODTK_OK;
}

static int rccl_dtype(int dtype, ncclDataType_t* t) {
    if (dtype == ODTK_F32) { *t = ncclFloat32; return ODTK_OK; }
    if (dtype == ODTK_BF16) { *t = ncclBfloat16; return ODTK_OK; }
    odtk::set_error("odtk_comm: dtype %d (ODTK_F32 or ODTK_BF16)", dtype);
    return ODTK_ERR_ARG;
}

extern "C" int odtk_comm_allreduce(odtk_comm* comm, const void* send, void* recv, long long count, int dtype, void* stream) {
    ODTK_REQUIRE(comm && comm->comm, "comm_allreduce: null communicator");
    ODTK_REQUIRE(count >= 0 && (count == 0 || (send && recv)), "comm_allreduce: bad buffer / count %lld", count);
    ncclDataType_t t;
    if (int e = rccl_dtype(dtype, &t)) return e;
    if (count == 0) return ODTK_OK;
    if (int e = on_comm_device(comm, "comm_allreduce")) return e;
    ODTK_CHECK_RCCL(g_rccl.all_reduce(send, recv, (size_t)count, t, ncclSum, comm->comm, (hipStream_t)stream));
    return ODTK_OK;
}

extern "C" int odtk_comm_broadcast(odtk_comm* comm, void* buf, long long count, int dtype, int root, void* stream) {
    ODTK_REQUIRE(comm && comm->comm, "comm_broadcast: null communicator");
    ODTK_REQUIRE(count >= 0 && (count == 0 || buf) && root >= 0 && root < comm->world, "comm_broadcast: bad buffer / count %lld / root %d", count, root);
    ncclDataType_t t;
    if (int e = rccl_dtype(dtype, &t)) return e;
    if (count == 0) return ODTK_OK;
    if (int e = on_comm_device(comm, "comm_broadcast")) return e;
    ODTK_CHECK_RCCL(g_rccl.broadcast(buf, buf, (size_t)count, t, root, comm->comm, (hipStream_t)stream));
    return ODTK_OK;
}

extern "C" int odtk_comm_destroy(odtk_comm* comm) {
    if (!comm) return ODTK_OK;
    ncclResult_t r = ncclSuccess;
    if (comm->comm) r = g_rccl.comm_destroy(comm->comm);
    delete comm;                                  // the handle is gone whatever RCCL answered
    if (r != ncclSuccess) {
        odtk::set_error("odtk_comm_destroy: ncclCommDestroy -> %s", g_rccl.error_string(r));
        return ODTK_ERR_HIP;
    }
    return ODTK_OK;
}
