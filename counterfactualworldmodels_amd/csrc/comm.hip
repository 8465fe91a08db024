// Collectives around the sharded counterfactual-sampling loop (BASELINE configs[3]): RCCL called directly, one
// communicator per process (= per GPU), everything enqueued on the caller's HIP stream.
//
// The path itself has no collective (prompts are independent forwards); around it there are exactly two:
//   cwm_broadcast   rank 0 -> all: one packed buffer {header | frame pair | prompt table | rectangularised masks}
//   cwm_allgatherv  every rank's block of predictions -> all ranks, blocks may differ by a row (grouped ncclBroadcast,
//                   no padding copies); cwm_allgather is the equal-block form (ncclAllGather)
// plus cwm_allreduce_sum_f32 for the sample-sharded flow statistics (f-4: motion-map sum).
//
// RCCL is bound at run time (dlopen) so that libcwm_hip.so loads on machines without it and so that the process uses ONE RCCL:
// the copy PyTorch already mapped if there is one (cwm_comm_load(path) names it explicitly; the default tries the loaded
// sonames first, then the ROCm install).  Signatures below restate rccl.h (NCCL 2.2x API, stable across 2.26 / 2.27).
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include "../../include/cwm_hip.h"
#include "common.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
enum { kNcclUint8 = 1, kNcclFloat32 = 7, kNcclSum = 0 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;

template <typename F>
bool bind(F& fn, const char* name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    return fn != nullptr;
}

int load_rccl(const char* path) {
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        CWM_REQUIRE(h, "cwm_comm_load: dlopen(%s) failed: %s", path, dlerror());
    } else {
        const char* loaded[] = {"librccl.so.1", "librccl.so"};  // a copy this process already mapped (PyTorch's), by soname
        for (const char* n : loaded)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        const char* fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : fresh)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        CWM_REQUIRE(h, "cwm_comm: RCCL (librccl.so.1) not found: %s", dlerror());
    }
    g_rccl.handle = h;
    const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                    bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.Broadcast, "ncclBroadcast") &&
                    bind(g_rccl.AllGather, "ncclAllGather") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                    bind(g_rccl.GroupStart, "ncclGroupStart") && bind(g_rccl.GroupEnd, "ncclGroupEnd") &&
                    bind(g_rccl.GetErrorString, "ncclGetErrorString") && bind(g_rccl.GetVersion, "ncclGetVersion");
    if (!ok) {
        g_rccl = Rccl();
        cwm_set_error("cwm_comm: the RCCL library lacks a required symbol");
        return CWM_ERR_INVALID;
    }
    return 0;
}

#define CWM_RCCL_CHECK(expr)                                                                                  \
    do {                                                                                                      \
        const int _r = (expr);                                                                                \
        if (_r != 0) {                                                                                        \
            cwm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));      \
            return CWM_ERR_HIP;                                                                               \
        }                                                                                                     \
    } while (0)

}  // namespace

struct cwm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

extern "C" int cwm_comm_load(const char* rccl_path) { return load_rccl(rccl_path); }

extern "C" int cwm_comm_version(void) {
    int v = 0;
    if (load_rccl(nullptr) != 0 || g_rccl.GetVersion(&v) != 0) return -1;
    return v;
}

extern "C" int cwm_comm_unique_id(uint8_t* id_out) {
    CWM_REQUIRE(id_out, "cwm_comm_unique_id: null argument");
    if (int rc = load_rccl(nullptr)) return rc;
    ncclUniqueId id;
    CWM_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, CWM_COMM_ID_BYTES);
    return CWM_OK;
}

extern "C" int cwm_comm_init(int rank, int nranks, const uint8_t* id_in, cwm_comm** out) {
    CWM_REQUIRE(id_in && out && nranks >= 1 && rank >= 0 && rank < nranks, "cwm_comm_init: bad argument (rank %d of %d)", rank, nranks);
    if (int rc = load_rccl(nullptr)) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id_in, CWM_COMM_ID_BYTES);
    cwm_comm* c = new cwm_comm();
    c->rank = rank;
    c->nranks = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        cwm_set_error("cwm_comm_init: no current HIP device");
        return CWM_ERR_HIP;
    }
    const int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        delete c;
        cwm_set_error("cwm_comm_init: ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
        return CWM_ERR_HIP;
    }
    *out = c;
    return CWM_OK;
}

extern "C" void cwm_comm_destroy(cwm_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int cwm_comm_rank(const cwm_comm* c) { return c ? c->rank : -1; }
extern "C" int cwm_comm_size(const cwm_comm* c) { return c ? c->nranks : -1; }

extern "C" int cwm_broadcast(cwm_comm* c, void* buf_dev, size_t bytes, int root, void* stream) {
    CWM_REQUIRE(c && buf_dev && root >= 0 && root < c->nranks, "cwm_broadcast: bad argument");
    if (int rc = cwm_require_device(c->device, "cwm_broadcast")) return rc;
    if (bytes == 0) return CWM_OK;
    CWM_RCCL_CHECK(g_rccl.Broadcast(buf_dev, buf_dev, bytes, kNcclUint8, root, c->comm, (hipStream_t)stream));
    return CWM_OK;
}

extern "C" int cwm_allgather(cwm_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    CWM_REQUIRE(c && send_dev && recv_dev, "cwm_allgather: bad argument");
    if (int rc = cwm_require_device(c->device, "cwm_allgather")) return rc;
    if (bytes_per_rank == 0) return CWM_OK;
    CWM_RCCL_CHECK(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, c->comm, (hipStream_t)stream));
    return CWM_OK;
}

extern "C" int cwm_allgatherv(cwm_comm* c, const void* send_dev, void* recv_dev, const size_t* offsets, const size_t* counts, void* stream) {
    CWM_REQUIRE(c && recv_dev && offsets && counts, "cwm_allgatherv: bad argument");
    if (int rc = cwm_require_device(c->device, "cwm_allgatherv")) return rc;
    CWM_REQUIRE(send_dev || counts[c->rank] == 0, "cwm_allgatherv: null send buffer for a non-empty block");
    // one fused group of per-root broadcasts: rank r's block lands at recv + offsets[r] on every rank (its own block too)
    CWM_RCCL_CHECK(g_rccl.GroupStart());
    for (int r = 0; r < c->nranks; ++r) {
        if (counts[r] == 0) continue;
        char* dst = (char*)recv_dev + offsets[r];
        const void* src = (r == c->rank) ? send_dev : (const void*)dst;
        const int rc = g_rccl.Broadcast(src, dst, counts[r], kNcclUint8, r, c->comm, (hipStream_t)stream);
        if (rc != 0) {
            (void)g_rccl.GroupEnd();
            cwm_set_error("cwm_allgatherv: ncclBroadcast(root %d) failed: %s", r, g_rccl.GetErrorString(rc));
            return CWM_ERR_HIP;
        }
    }
    CWM_RCCL_CHECK(g_rccl.GroupEnd());
    return CWM_OK;
}

extern "C" int cwm_allreduce_sum_f32(cwm_comm* c, float* buf_dev, size_t count, void* stream) {
    CWM_REQUIRE(c && buf_dev, "cwm_allreduce_sum_f32: bad argument");
    if (int rc = cwm_require_device(c->device, "cwm_allreduce_sum_f32")) return rc;
    if (count == 0) return CWM_OK;
    CWM_RCCL_CHECK(g_rccl.AllReduce(buf_dev, buf_dev, count, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream));
    return CWM_OK;
}
