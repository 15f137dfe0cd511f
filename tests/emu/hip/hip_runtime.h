/* Fake <hip/hip_runtime.h> for the CPU emulator build (TEST INFRASTRUCTURE):
 * "device" memory is host memory, kernels run on fibers (hip_emu.h). */
#ifndef PMG_FAKE_HIP_RUNTIME_H
#define PMG_FAKE_HIP_RUNTIME_H
#include "../hip_emu.h"
#include <chrono>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct emu_stream_s* hipStream_t;
struct emu_event_s { std::chrono::steady_clock::time_point t; };
typedef emu_event_s* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

static inline const char* hipGetErrorString(hipError_t) { return "emulated hip error"; }
/* ranks of a multi-process test each pick "their" device; PMG_EMU_DEVICES=1 plays a launcher that shows every rank one GPU */
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("PMG_EMU_DEVICES"); *n = e ? atoi(e) : 8; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; } /* an MI355X */
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostMalloc(void** p, size_t n) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event_s(); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } /* the emulator runs everything in order */
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((int)(grid).x, (int)(block).x, [&]() { kernel(__VA_ARGS__); })
#endif
