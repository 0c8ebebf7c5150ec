// The one collective of the design: an all-gather of the solved first controls u0 of every rank's
// problem shard, issued once per MPC step.  (B/G)*nu elements per rank -- a single small, latency-bound message over
// xGMI; there is no collective on the callback path and the reference has no distributed code at all.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first nempc_comm_* call): a process that never shards
// (the B=1 drop-in path, the single-GPU bench) does not load it, and a process that already holds RCCL through
// torch.distributed shares that copy (same SONAME).  The communicator is created by the caller's ranks from one
// ncclUniqueId that rank 0 obtains with nempc_comm_unique_id and hands to the others over whatever channel the host
// side already has (torch.distributed's store / broadcast in pyneuralempc_amd/parallel.py, MPI, a file).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "nempc_internal.h"

namespace nempc {
namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

RcclApi g_api;

RcclApi* rccl() {
    RcclApi& api = g_api;
    static std::once_flag once;
    std::call_once(once, [&api] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            const char* e = dlerror();
            api.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [&](const char* s) { return dlsym(api.lib, s); };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
            api.why = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return api.lib ? &api : nullptr;
}

int rccl_missing(const char* who) {
    set_error(std::string(who) + ": " + g_api.why);
    return NEMPC_EUNSUPPORTED;
}

int rccl_fail(RcclApi* a, ncclResult_t r, const char* what) {
    set_error(std::string(what) + ": " + a->GetErrorString(r));
    return NEMPC_EHIP;
}

// u0 of every problem of the shard -> the rank's slot of the gather buffer; rows past B (ragged shards padded to the
// largest) are zero-filled.  Lanes run over the flat (row, control) index: contiguous stores, nu-strided loads.
template <typename T>
__global__ void pack_u0_kernel(const T* __restrict__ Z, const T* __restrict__ u0, T* __restrict__ slot, int B, int rows,
                               int nu, int n, int u_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nu) return;
    const int b = i / nu, c = i - b * nu;
    T v = T(0);
    if (b < B) v = Z ? Z[(size_t)b * n + u_off + c] : u0[i];
    slot[i] = v;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = -1;
};

void comm_free(Handle& h) {
    Comm* c = static_cast<Comm*>(h.comm);
    if (!c) return;
    if (c->comm) {
        RcclApi* a = rccl();
        if (a) (void)a->CommDestroy(c->comm);
    }
    delete c;
    h.comm = nullptr;
}

}  // namespace nempc

using namespace nempc;

extern "C" {

int nempc_comm_unique_id(void* id_out) {
    if (!id_out) { set_error("nempc_comm_unique_id: null argument"); return NEMPC_EINVAL; }
    RcclApi* a = rccl();
    if (!a) return rccl_missing("nempc_comm_unique_id");
    static_assert(sizeof(ncclUniqueId) == NEMPC_COMM_ID_BYTES, "NEMPC_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(a, r, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return NEMPC_OK;
}

int nempc_comm_init(nempc_handle hh, int32_t nranks, int32_t rank, const void* id) {
    if (!hh || !id) { set_error("nempc_comm_init: null argument"); return NEMPC_EINVAL; }
    if (nranks < 1 || rank < 0 || rank >= nranks) { set_error("nempc_comm_init: rank outside [0, nranks)"); return NEMPC_EINVAL; }
    Handle& h = *reinterpret_cast<Handle*>(hh);
    RcclApi* a = rccl();
    if (!a) return rccl_missing("nempc_comm_init");
    comm_free(h);
    int prev = -1;
    NEMPC_HIP(hipGetDevice(&prev));
    if (prev != h.cfg.device) NEMPC_HIP(hipSetDevice(h.cfg.device));
    Comm* c = new Comm();
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = a->CommInitRank(&c->comm, nranks, uid, rank);
    if (prev != h.cfg.device) (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail(a, r, "ncclCommInitRank");
    }
    c->nranks = nranks;
    c->rank = rank;
    h.comm = c;
    return NEMPC_OK;
}

int nempc_comm_destroy(nempc_handle hh) {
    if (!hh) { set_error("nempc_comm_destroy: null handle"); return NEMPC_EINVAL; }
    comm_free(*reinterpret_cast<Handle*>(hh));
    return NEMPC_OK;
}

int nempc_allgather_u0(nempc_handle hh, int32_t B, int32_t rows_per_rank, const void* Z, const void* u0, void* gathered,
                       void* stream) {
    if (!hh) { set_error("nempc_allgather_u0: null handle"); return NEMPC_EINVAL; }
    Handle& h = *reinterpret_cast<Handle*>(hh);
    Comm* c = static_cast<Comm*>(h.comm);
    if (!c) { set_error("nempc_allgather_u0: call nempc_comm_init first"); return NEMPC_ESTATE; }
    if (B < 0 || rows_per_rank < 1 || B > rows_per_rank) {
        set_error("nempc_allgather_u0: need 0 <= B <= rows_per_rank, rows_per_rank >= 1");
        return NEMPC_EINVAL;
    }
    if ((!Z) == (!u0) && B > 0) { set_error("nempc_allgather_u0: pass exactly one of Z (B,n) and u0 (B,nu)"); return NEMPC_EINVAL; }
    if (!gathered) { set_error("nempc_allgather_u0: gathered is null"); return NEMPC_EINVAL; }
    RcclApi* a = rccl();
    if (!a) return rccl_missing("nempc_allgather_u0");
    int prev = -1;
    NEMPC_HIP(hipGetDevice(&prev));
    if (prev != h.cfg.device) NEMPC_HIP(hipSetDevice(h.cfg.device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nu = h.cfg.nu, cnt = rows_per_rank * nu;
    char* slot = static_cast<char*>(gathered) + (size_t)c->rank * cnt * h.esz;   // in-place all-gather: send = own slot
    const int thr = 256, blocks = (cnt + thr - 1) / thr;
    if (h.cfg.dtype == NEMPC_F64)
        pack_u0_kernel<double><<<blocks, thr, 0, s>>>(static_cast<const double*>(Z), static_cast<const double*>(u0),
                                                       reinterpret_cast<double*>(slot), B, rows_per_rank, nu, h.n,
                                                       h.cfg.H * h.cfg.nx);
    else
        pack_u0_kernel<float><<<blocks, thr, 0, s>>>(static_cast<const float*>(Z), static_cast<const float*>(u0),
                                                      reinterpret_cast<float*>(slot), B, rows_per_rank, nu, h.n,
                                                      h.cfg.H * h.cfg.nx);
    hipError_t le = hipGetLastError();
    ncclResult_t r = ncclSuccess;
    if (le == hipSuccess)
        r = a->AllGather(slot, gathered, (size_t)cnt, h.cfg.dtype == NEMPC_F64 ? ncclDouble : ncclFloat, c->comm, s);
    if (prev != h.cfg.device) (void)hipSetDevice(prev);
    if (le != hipSuccess) return hip_fail(le, "pack_u0_kernel launch");
    if (r != ncclSuccess) return rccl_fail(a, r, "ncclAllGather");
    return NEMPC_OK;
}

int nempc_comm_size(nempc_handle hh, int32_t* nranks, int32_t* rank) {
    if (!hh) { set_error("nempc_comm_size: null handle"); return NEMPC_EINVAL; }
    Comm* c = static_cast<Comm*>(reinterpret_cast<Handle*>(hh)->comm);
    if (nranks) *nranks = c ? c->nranks : 0;
    if (rank) *rank = c ? c->rank : -1;
    return NEMPC_OK;
}

}  // extern "C"
