// nst_comm.cpp - the collective of the sharded closure (BASELINE config 4) behind the C ABI: one RCCL
// communicator per rank (one rank per GPU), an in-place all-reduce(sum, fp32) on the job's own HIP stream.
//
// The reference has no counterpart (single process, single device: neural_style_transfer.py:236-245).  What travels
// per closure is ONE buffer: the 3*H0*W0 pixel-gradient floats followed by the 4*levels+1 loss scalars
// (nst_opt_shard_levels_comm packs them), so a closure costs one collective on the xGMI links (two in stripe mode).
//
// librccl is resolved at run time (dlopen), not at link time: libnst_hip.so then loads on hosts without RCCL, and in
// a process that already carries a librccl (PyTorch ships one with the same SONAME) the same copy is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/nst_hip.h"

extern "C" int nst_internal_fail(nst_ctx* ctx, int code, const char* msg);

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[NST_COMM_ID_BYTES]; };
using Comm = void*;
struct Api {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};

Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { a.why = std::string("librccl not found: ") + dlerror(); return; }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
        if (!a.ok) a.why = "librccl lacks an expected symbol";
    });
    return a;
}

int rccl_fail(const char* what, int rc) {
    Api& a = api();
    return nst_internal_fail(nullptr, NST_E_HIP, (std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "?")).c_str());
}

}  // namespace

struct nst_comm {
    Comm comm = nullptr;
    int device = 0, rank = 0, world = 1;
    long calls = 0;
    double bytes = 0;
};

extern "C" {

int nst_comm_unique_id(void* id) {
    if (!id) return nst_internal_fail(nullptr, NST_E_ARG, "null argument");
    Api& a = api();
    if (!a.ok) return nst_internal_fail(nullptr, NST_E_STATE, a.why.c_str());
    UniqueId u;
    const int rc = a.GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, NST_COMM_ID_BYTES);
    return NST_OK;
}

int nst_comm_create(int device, int rank, int world, const void* id, nst_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return nst_internal_fail(nullptr, NST_E_ARG, "bad communicator arguments");
    Api& a = api();
    if (!a.ok) return nst_internal_fail(nullptr, NST_E_STATE, a.why.c_str());
    if (hipSetDevice(device) != hipSuccess) return nst_internal_fail(nullptr, NST_E_HIP, "hipSetDevice failed");
    nst_comm* c = new (std::nothrow) nst_comm();
    if (!c) return nst_internal_fail(nullptr, NST_E_NOMEM, "out of host memory");
    c->device = device; c->rank = rank; c->world = world;
    UniqueId u;
    std::memcpy(u.internal, id, NST_COMM_ID_BYTES);
    const int rc = a.CommInitRank(&c->comm, world, u, rank);
    if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return NST_OK;
}

void nst_comm_destroy(nst_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)api().CommDestroy(c->comm);
    delete c;
}

int nst_comm_info(const nst_comm* c, int* rank, int* world, long* calls, double* bytes) {
    if (!c) return nst_internal_fail(nullptr, NST_E_ARG, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (calls) *calls = c->calls;
    if (bytes) *bytes = c->bytes;
    return NST_OK;
}

int nst_comm_allreduce_sum(nst_comm* c, float* buf, size_t n, void* stream) {
    if (!c || !buf) return nst_internal_fail(nullptr, NST_E_ARG, "null argument");
    if (hipSetDevice(c->device) != hipSuccess) return nst_internal_fail(nullptr, NST_E_HIP, "hipSetDevice failed");
    const int rc = api().AllReduce(buf, buf, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    c->calls += 1;
    c->bytes += (double)n * 4.0;
    return NST_OK;
}

}  // extern "C"
