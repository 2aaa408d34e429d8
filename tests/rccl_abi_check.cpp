// Compile-time check (tests/test_abi.py: g++ -fsyntax-only): the hand-written slice of rccl.h that csrc/svr_rccl.cpp binds by dlsym
// (csrc/svr_rccl_abi.h) against the RCCL header of this image.  Nothing here runs; a mismatch is a compile error naming the entry point.
#include <rccl/rccl.h>

#include <type_traits>

#include "../fetalreconstruction_amd/csrc/svr_rccl_abi.h"

namespace abi = svr_rccl_abi;

// enum values and sizes
static_assert((int)::ncclSuccess == (int)abi::ncclSuccess, "ncclSuccess");
static_assert((int)::ncclInt32 == (int)abi::ncclInt32 && (int)::ncclFloat32 == (int)abi::ncclFloat32 && (int)::ncclFloat64 == (int)abi::ncclFloat64, "ncclDataType_t values");
static_assert((int)::ncclSum == (int)abi::ncclSum && (int)::ncclProd == (int)abi::ncclProd && (int)::ncclMax == (int)abi::ncclMax && (int)::ncclMin == (int)abi::ncclMin, "ncclRedOp_t values");
static_assert(sizeof(::ncclUniqueId) == sizeof(abi::ncclUniqueId) && alignof(::ncclUniqueId) == alignof(abi::ncclUniqueId) && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId");
static_assert(std::is_trivially_copyable<::ncclUniqueId>::value && std::is_standard_layout<::ncclUniqueId>::value, "ncclUniqueId is passed by value");
static_assert(sizeof(::ncclResult_t) == sizeof(int) && sizeof(::ncclDataType_t) == sizeof(int) && sizeof(::ncclRedOp_t) == sizeof(int), "RCCL's enums are passed as int");
static_assert(sizeof(::ncclComm_t) == sizeof(abi::ncclComm_t) && std::is_pointer<::ncclComm_t>::value, "ncclComm_t is an opaque pointer");

// prototypes: RCCL's own, with its enum types mapped to int and its handles to the hand-written ones, must be the hand-written ones
template <class T> struct map_arg { typedef typename std::conditional<std::is_enum<T>::value, int, T>::type type; };
template <> struct map_arg<::ncclComm_t> { typedef abi::ncclComm_t type; };
template <> struct map_arg<::ncclComm_t *> { typedef abi::ncclComm_t *type; };
template <> struct map_arg<::ncclUniqueId> { typedef abi::ncclUniqueId type; };
template <> struct map_arg<::ncclUniqueId *> { typedef abi::ncclUniqueId *type; };
template <class F> struct map_fn;
template <class R, class... A> struct map_fn<R (*)(A...)> { typedef typename map_arg<R>::type (*type)(typename map_arg<A>::type...); };
#define SAME(fn, hand) static_assert(std::is_same<map_fn<decltype(&::fn)>::type, abi::hand>::value, #fn " is not what csrc/svr_rccl_abi.h declares")
SAME(ncclGetUniqueId, GetUniqueId_fn);
SAME(ncclCommInitRank, CommInitRank_fn);
SAME(ncclCommDestroy, CommDestroy_fn);
SAME(ncclCommCount, CommCount_fn);
SAME(ncclAllReduce, AllReduce_fn);
SAME(ncclAllGather, AllGather_fn);
SAME(ncclReduceScatter, ReduceScatter_fn);
SAME(ncclGroupStart, GroupStart_fn);
SAME(ncclGroupEnd, GroupEnd_fn);
SAME(ncclGetErrorString, GetErrorString_fn);

int main() { return 0; }
