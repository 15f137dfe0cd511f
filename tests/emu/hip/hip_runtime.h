/* Fake <hip/hip_runtime.h> for the CPU emulator build (TEST INFRASTRUCTURE):
 * "device" memory is host memory, kernels run on fibers (hip_emu.h). */
#ifndef PMG_FAKE_HIP_RUNTIME_H
#define PMG_FAKE_HIP_RUNTIME_H
#include "../hip_emu.h"
#include <chrono>
#include <deque>
#include <functional>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
/* Streams.  The emulator runs every launch in order at the call -- except on a NON-BLOCKING stream when PMG_EMU_LAZY_COMM=1: work
 * enqueued there (the product's communication stream: the overlapped all-gather) is DEFERRED until somebody depends on it --
 * a wait on one of its events, a synchronise -- i.e. it runs as LATE as the recorded dependencies allow.  A missing dependency
 * (the next step overwriting the rows an all-gather has yet to read) then shows up as wrong data instead of hiding behind the
 * emulator's in-order execution (tests/test_distributed_gloo.py). */
struct emu_event_s;
struct emu_stream_s {
    bool lazy = false;
    std::deque<std::function<void()>> q;
    void flush() { while (!q.empty()) { auto f = std::move(q.front()); q.pop_front(); f(); } }
    void flush_until(emu_event_s* e);
};
typedef emu_stream_s* hipStream_t;
struct emu_event_s { std::chrono::steady_clock::time_point t; emu_stream_s* pending = nullptr; };
inline void emu_stream_s::flush_until(emu_event_s* e) { while (e->pending == this && !q.empty()) { auto f = std::move(q.front()); q.pop_front(); f(); } }
typedef emu_event_s* hipEvent_t;
/* enqueue on a stream: now, or deferred on a lazy one */
template <class F> static inline void emu_enqueue(hipStream_t s, F&& f) { if (s && s->lazy) s->q.emplace_back(std::forward<F>(f)); else f(); }
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
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags)
{
    const char* lz = getenv("PMG_EMU_LAZY_COMM");
    *s = new emu_stream_s();
    (*s)->lazy = (flags & hipStreamNonBlocking) && lz && atoi(lz) != 0;
    return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t s) { if (s) { s->flush(); delete s; } return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { if (s) s->flush(); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event_s(); return hipSuccess; }
/* an eager stream runs in order at the call: a wait on an event of a lazy stream runs that stream up to the event NOW; a lazy
 * stream that waits (for an event of an eager stream: complete since its record; of a lazy one: run it that far) defers the wait */
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    if (s && s->lazy) { s->q.emplace_back([e]() { if (e->pending) e->pending->flush_until(e); }); return hipSuccess; }
    if (e->pending) e->pending->flush_until(e);
    return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) { if (e->pending) e->pending->flush_until(e); delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (s && s->lazy) { if (e->pending && e->pending != s) e->pending->flush_until(e); e->pending = s; s->q.emplace_back([e]() { e->t = std::chrono::steady_clock::now(); e->pending = nullptr; }); return hipSuccess; }
    if (e->pending) e->pending->flush_until(e);
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) { if (e->pending) e->pending->flush_until(e); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    if (a->pending) a->pending->flush_until(a);
    if (b->pending) b->pending->flush_until(b);
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((int)(grid).x, (int)(block).x, [&]() { kernel(__VA_ARGS__); })
#endif
