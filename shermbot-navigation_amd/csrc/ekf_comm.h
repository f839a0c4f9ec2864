// ekf_comm.h -- the one exchange step of the Monte-Carlo batch: every rank's batch-statistics vector
// (2 len + 6 doubles, a few KB) gathered over RCCL / xGMI and added in RANK ORDER on the device, so the total is
// bit-reproducible and independent of ring order.  There is no collective anywhere in the data path; this runs once
// per reporting interval.  Included by nuslam_hip.hip (it needs the batch handle's stream and buffers).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU host program never loads it, and a missing
// library surfaces as NUSLAM_E_COMM from nuslam_comm_unique_id / nuslam_comm_create -- there is no fallback path.
#pragma once
#include <dlfcn.h>

namespace nuslam_comm_detail {

// the subset of rccl.h this file needs (NCCL 2.x ABI: 128-byte unique id, ncclDouble = 8)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int Result;
constexpr int kDouble = 8;

struct Api {
    void* so = nullptr;
    Result (*GetUniqueId)(UniqueId*) = nullptr;
    Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(Result) = nullptr;
    std::string err;
};

inline Api& api()
{
    static Api a;
    if (a.so || !a.err.empty()) return a;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) {
        a.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.so) break;
    }
    if (!a.so) { a.err = std::string("dlopen(librccl): ") + dlerror(); return a; }
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.so, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.so, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.so, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.so, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.so, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
        a.err = "librccl: a required symbol is missing";
        a.so = nullptr;
    }
    return a;
}

// total[i] = sum over ranks r = 0, 1, ... (in that order) of rows[r * len + i]
__global__ __launch_bounds__(256) void k_rank_ordered_sum(const double* __restrict__ rows, int world, int len,
                                                          double* __restrict__ total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    double a = 0.0;
    for (int r = 0; r < world; ++r) a = a + rows[(size_t)r * len + i];
    total[i] = a;
}

} // namespace nuslam_comm_detail

struct nuslam_comm {
    nuslam_comm_detail::Comm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    double* gathered = nullptr;    // [world][cap]
    double* total = nullptr;       // [cap]
    int cap = 0;
};
