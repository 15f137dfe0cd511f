/* hip_emu.cpp -- fiber scheduler of the SPMD emulator (TEST INFRASTRUCTURE, see hip_emu.h). */
#include "hip_emu.h"

#include <cstdio>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
float emu_xf[16 * 64 * 16];

namespace {
constexpr size_t STACK_BYTES = 512 * 1024;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    int wait_gen = -1;     /* >= 0: parked at the workgroup barrier of that generation (the scheduler skips it) */
};
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;
int g_bar_gen = 0, g_bar_count = 0, g_live = 0;   /* workgroup barrier: generation, arrivals, live threads */

void trampoline()
{
    (*g_body)();
    g_fibers[g_cur].done = true;
    g_live--;
    if (g_live > 0 && g_bar_count >= g_live) { g_bar_count = 0; g_bar_gen++; }   /* the rest may be waiting for this one */
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}
}  // namespace

/* __syncthreads(): every LIVE thread of the workgroup has to arrive.  Wavefronts of one workgroup may run different code
 * between two workgroup barriers (the helper wavefront of the two-wave step kernel), with different numbers of
 * cross-lane yields in between: an arrived fiber keeps yielding until the last one is in. */
void emu_block_barrier()
{
    const int gen = g_bar_gen;
    if (++g_bar_count >= g_live) { g_bar_count = 0; g_bar_gen++; }
    const int me = g_cur;
    g_fibers[me].wait_gen = gen;
    while (g_bar_gen == gen) emu_barrier();
    g_fibers[me].wait_gen = -1;
}
void emu_barrier()
{
    /* yield to the scheduler; it resumes this fiber after every other live
     * fiber of the block has run up to its own next barrier */
    int me = g_cur;
    swapcontext(&g_fibers[me].ctx, &g_sched);
}

void emu::launch(int grid, int block, const std::function<void()>& body)
{
    g_body = &body;
    gridDim = {(unsigned)grid, 1, 1};
    blockDim = {(unsigned)block, 1, 1};
    if ((int)g_fibers.size() < block) g_fibers.resize(block);
    for (int i = 0; i < block; i++)
        if (!g_fibers[i].stack) g_fibers[i].stack = (char*)malloc(STACK_BYTES);
    for (int b = 0; b < grid; b++) {
        blockIdx = {(unsigned)b, 0, 0};
        for (int i = 0; i < block; i++) {
            Fiber& f = g_fibers[i];
            f.done = false;
            f.wait_gen = -1;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK_BYTES;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        g_live = block; g_bar_count = 0;
        bool alive = true;
        while (alive) {
            alive = false;
            for (int i = 0; i < block; i++) {
                if (g_fibers[i].done) continue;
                if (g_fibers[i].wait_gen >= 0 && g_fibers[i].wait_gen == g_bar_gen) { alive = true; continue; }   /* parked */
                g_cur = i;
                threadIdx = {(unsigned)i, 0, 0};
                swapcontext(&g_sched, &g_fibers[i].ctx);
                if (!g_fibers[i].done) alive = true;
            }
        }
    }
    g_cur = -1;
}
