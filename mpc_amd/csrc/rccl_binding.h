// rccl_binding.h — the ten RCCL entry points comm.cpp binds with dlsym, as function pointers, checked AT COMPILE TIME against
// the declarations of <rccl/rccl.h> (a dlsym'ed symbol carries no type: a prototype written by hand that drifts from the
// library's — an enum that changes width, a by-value ncclUniqueId that changes size — would only show with real ranks).
// Also compiled by tests/test_rccl_binding.py on CPU.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <type_traits>

namespace gc_rccl {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";
};

#define GC_RCCL_SAME(member, fn) \
    static_assert(std::is_same<decltype(Rccl::member), decltype(&fn)>::value, "comm.cpp binds " #fn " with another prototype than <rccl/rccl.h> declares")
GC_RCCL_SAME(GetVersion, ncclGetVersion);
GC_RCCL_SAME(GetUniqueId, ncclGetUniqueId);
GC_RCCL_SAME(CommInitRank, ncclCommInitRank);
GC_RCCL_SAME(CommInitAll, ncclCommInitAll);
GC_RCCL_SAME(CommDestroy, ncclCommDestroy);
GC_RCCL_SAME(AllGather, ncclAllGather);
GC_RCCL_SAME(AllReduce, ncclAllReduce);
GC_RCCL_SAME(GroupStart, ncclGroupStart);
GC_RCCL_SAME(GroupEnd, ncclGroupEnd);
GC_RCCL_SAME(GetErrorString, ncclGetErrorString);
#undef GC_RCCL_SAME
// what crosses the C ABI and the wire between the ranks (include/gcengine.h: GC_COMM_ID_BYTES) and what comm.cpp passes
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is the 128 bytes gc_comm_get_unique_id hands out");
static_assert(std::is_trivially_copyable<ncclUniqueId>::value, "ncclUniqueId travels as plain bytes");
static_assert((int)ncclSuccess == 0, "ncclSuccess");
static_assert((int)ncclUint8 == 1 && (int)ncclFloat64 == 8, "the data types of the gather (bytes) and of the barrier / max (doubles)");
static_assert((int)ncclMax == 2 && (int)ncclSum == 0, "the reductions of gc_comm_allreduce_max / the barrier");

}  // namespace gc_rccl
