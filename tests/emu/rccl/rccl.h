/* Fake <rccl/rccl.h> for the CPU emulator build (TEST INFRASTRUCTURE).  Ranks are separate host processes
 * (torch.distributed.run / pytest-spawned): the "communicator" is a POSIX shared-memory segment named after the
 * unique id, all-gather = copy own shard in, barrier, copy every shard out, barrier.  Enough to run the product's
 * multi-rank control flow (pmg_comm_init / pmg_allgather_packed, bench.py --gpus N) without GPUs. */
#ifndef PMG_FAKE_RCCL_H
#define PMG_FAKE_RCCL_H
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/time.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
typedef struct { char internal[128]; } ncclUniqueId;
struct emu_comm_s {
    char name[128];
    unsigned char* base;
    size_t bytes;
    int n, rank;
    long seq;
};
typedef struct emu_comm_s* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandled = 1 };
enum ncclDataType_t { ncclFloat = 7 };
static constexpr size_t EMU_SLOT_BYTES = 8u << 20; /* per-rank shard limit of the emulated all-gather */
static constexpr size_t EMU_HDR_BYTES = 4096;
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* u)
{
    memset(u, 0, sizeof(*u));
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    snprintf(u->internal, sizeof(u->internal), "/pmg_emu_%d_%ld_%ld", (int)getpid(), (long)tv.tv_sec, (long)tv.tv_usec);
    return ncclSuccess;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId id, int rank)
{
    if (n < 1 || rank < 0 || rank >= n) return ncclUnhandled;
    emu_comm_s* m = new emu_comm_s();
    snprintf(m->name, sizeof(m->name), "%s", id.internal[0] ? id.internal : "/pmg_emu_default");
    m->n = n; m->rank = rank; m->seq = 0;
    m->bytes = EMU_HDR_BYTES + (size_t)n * EMU_SLOT_BYTES;
    int fd = shm_open(m->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)m->bytes) != 0) { delete m; return ncclUnhandled; }
    m->base = (unsigned char*)mmap(nullptr, m->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m->base == (unsigned char*)MAP_FAILED) { delete m; return ncclUnhandled; }
    *c = m;
    return ncclSuccess;
}
static inline void emu_barrier(emu_comm_s* m)
{
    auto* ctr = reinterpret_cast<std::atomic<long>*>(m->base); /* zero-initialised by ftruncate */
    m->seq++;
    ctr->fetch_add(1);
    while (ctr->load() < m->seq * m->n) sched_yield();
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t m)
{
    if (!m) return ncclSuccess;
    munmap(m->base, m->bytes);
    if (m->rank == 0) shm_unlink(m->name);
    delete m;
    return ncclSuccess;
}
static inline const char* ncclGetErrorString(ncclResult_t) { return "emulated rccl failure"; }
static inline void emu_allgather_now(const void* s, void* d, size_t b, ncclComm_t m)
{
    memcpy(m->base + EMU_HDR_BYTES + (size_t)m->rank * EMU_SLOT_BYTES, s, b);
    emu_barrier(m);
    for (int r = 0; r < m->n; r++) memcpy((unsigned char*)d + (size_t)r * b, m->base + EMU_HDR_BYTES + (size_t)r * EMU_SLOT_BYTES, b);
    emu_barrier(m);
}
/* stream-ordered like the real one: on a lazy stream of the fake HIP runtime (hip/hip_runtime.h) the copy is deferred with
 * the rest of that stream's work -- every rank defers and flushes at the same points of the same program */
static inline ncclResult_t ncclAllGather(const void* s, void* d, size_t count, ncclDataType_t, ncclComm_t m, hipStream_t stream)
{
    size_t b = count * 4;
    if (b > EMU_SLOT_BYTES) return ncclUnhandled;
    emu_enqueue(stream, [=]() { emu_allgather_now(s, d, b, m); });
    return ncclSuccess;
}
#endif
