// dp_comm.h — the RCCL binding of libcae_hip.so's data-parallel path (include/cae_hip.h, cae_dp_*).
//
// The reference trains on one device (conv_ae_model.py:294-297, 312-313); data parallelism is this build's addition
// (SURVEY.md §8e).  RCCL is bound at run time with dlopen/dlsym instead of at link time: a Python host has already
// loaded torch's librccl.so (built against torch's own libamdhip64), and a process must not hold two HIP runtimes
//  — so the copy that is already resident is looked up first (RTLD_NOLOAD), and only a host without
// one gets /opt/rocm's.  Only the six entry points used here are declared; their signatures and the enum values are
// RCCL's public ABI (rccl.h: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstddef>

namespace cae {

struct RcclApi {
    using comm_t = void*;
    struct UniqueId {
        char internal[128];
    };
    static constexpr int kFloat32 = 7, kFloat64 = 8, kSum = 0;

    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* where = "";

    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllReduce && Broadcast && GetErrorString; }

    // nullptr on success, else what went wrong
    const char* load() {
        if (ok()) return nullptr;
        void* h = nullptr;
        const char* resident[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : resident)
            if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) where = "already resident (the host's copy)";
        for (const char* n : resident)
            if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) where = "loaded from the library path";
        if (!h) return "librccl.so is neither resident in this process nor on the library path";
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(h, "ncclBroadcast"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        return ok() ? nullptr : "librccl.so lacks one of ncclGetUniqueId/CommInitRank/CommDestroy/AllReduce/Broadcast/GetErrorString";
    }
};

inline RcclApi& rccl() {
    static RcclApi api;
    return api;
}

}  // namespace cae
