// svr_rccl_abi.h -- the slice of <rccl/rccl.h> that csrc/svr_rccl.cpp binds by dlsym: handles, enum values and the prototypes of the ten
// entry points, written out by hand so that libsvr_hip.so builds and loads without RCCL's headers or library (a single-GPU user never
// opens librccl).  tests/rccl_abi_check.cpp includes this file NEXT TO the real rccl.h and static_asserts that the two agree -- enum
// values, sizeof(ncclUniqueId), and every prototype after mapping RCCL's enum types to int (tests/test_abi.py compiles it).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

namespace svr_rccl_abi {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 };

// (result, data type and reduction operator are C enums in rccl.h: int-sized, passed as int)
typedef int (*GetUniqueId_fn)(ncclUniqueId *);
typedef int (*CommInitRank_fn)(ncclComm_t *, int, ncclUniqueId, int);
typedef int (*CommDestroy_fn)(ncclComm_t);
typedef int (*CommCount_fn)(ncclComm_t, int *);
typedef int (*AllReduce_fn)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*AllGather_fn)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef int (*ReduceScatter_fn)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*GroupStart_fn)();
typedef int (*GroupEnd_fn)();
typedef const char *(*GetErrorString_fn)(int);

}  // namespace svr_rccl_abi
