/* Fake <rccl/rccl.h> for the CPU emulator build (TEST INFRASTRUCTURE): single-rank only. */
#ifndef PMG_FAKE_RCCL_H
#define PMG_FAKE_RCCL_H
#include <cstring>
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct emu_comm_s* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandled = 1 };
enum ncclDataType_t { ncclFloat = 7 };
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* u) { memset(u, 0, sizeof(*u)); return ncclSuccess; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId, int) { *c = (ncclComm_t)1; return n == 1 ? ncclSuccess : ncclUnhandled; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
static inline const char* ncclGetErrorString(ncclResult_t) { return "emulated rccl: only 1 rank"; }
static inline ncclResult_t ncclAllGather(const void* s, void* d, size_t n, ncclDataType_t, ncclComm_t, void*) { memcpy(d, s, n * 4); return ncclSuccess; }
#endif
