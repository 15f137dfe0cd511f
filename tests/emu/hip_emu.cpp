/* hip_emu.cpp -- fiber scheduler of the SPMD emulator (TEST INFRASTRUCTURE, see hip_emu.h). */
#include "hip_emu.h"

#include <cstdio>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
float emu_xf[16 * 64 * 16];
long long pmge_face_clip_calls = 0;
extern "C" long long pmge_face_clip_count() { return pmge_face_clip_calls; }
long long pmge_cyl_contact_calls = 0, pmge_cyl_redo_calls = 0, pmge_cyl_spec_taken_n = 0;
extern "C" long long pmge_cyl_spec_taken() { return pmge_cyl_spec_taken_n; }
extern "C" long long pmge_cyl_contact_count() { return pmge_cyl_contact_calls; }
extern "C" long long pmge_cyl_redo_count() { return pmge_cyl_redo_calls; }

/* Context switch.  glibc's swapcontext() saves and restores the signal mask with two system calls per switch, and the
 * emulator switches at every cross-lane primitive of every lane; on x86-64 the fibers switch with a dozen instructions
 * instead (callee-saved registers + stack pointer: System V ABI), ucontext elsewhere. */
#if defined(__x86_64__) && !defined(PMG_EMU_UCONTEXT)
#define PMG_EMU_FAST_SWITCH 1
extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch, .-emu_ctx_switch
)");
#endif

namespace {
constexpr size_t STACK_BYTES = 512 * 1024;
struct Fiber {
#ifdef PMG_EMU_FAST_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool done = false;
    int wait_gen = -1;     /* >= 0: parked at the workgroup barrier of that generation (the scheduler skips it) */
};
#ifdef PMG_EMU_FAST_SWITCH
void* g_sched_sp = nullptr;
#else
ucontext_t g_sched;
#endif
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
#ifdef PMG_EMU_FAST_SWITCH
    emu_ctx_switch(&g_fibers[g_cur].sp, g_sched_sp);
    __builtin_trap();                                       /* a finished fiber is never resumed */
#else
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
#endif
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
#ifdef PMG_EMU_FAST_SWITCH
    emu_ctx_switch(&g_fibers[me].sp, g_sched_sp);
#else
    swapcontext(&g_fibers[me].ctx, &g_sched);
#endif
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
#ifdef PMG_EMU_FAST_SWITCH
            /* a fresh stack that "returns" into trampoline(): [6 callee-saved registers][&trampoline][0]; after the
             * switch's pops and ret the stack pointer is 8 below a 16-byte boundary, as behind a call */
            void** top = (void**)(((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15);
            *--top = nullptr;
            *--top = (void*)trampoline;
            for (int r = 0; r < 6; r++) *--top = nullptr;
            f.sp = top;
#else
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK_BYTES;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
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
#ifdef PMG_EMU_FAST_SWITCH
                emu_ctx_switch(&g_sched_sp, g_fibers[i].sp);
#else
                swapcontext(&g_sched, &g_fibers[i].ctx);
#endif
                if (!g_fibers[i].done) alive = true;
            }
        }
    }
    g_cur = -1;
}
