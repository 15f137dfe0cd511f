/*
 * hip_emu.h -- TEST INFRASTRUCTURE: a fiber-based SPMD emulator that lets the
 * product's HIP device code (pybullet_multigoal_gym_amd/csrc/pmg_device.h)
 * be compiled with g++ and executed lane-by-lane on the CPU, so that the
 * kernel LOGIC can be parity-checked against the oracle in the CPU-only test
 * tier.  It is never part of the product: the shipped library is built by
 * hipcc for gfx950 and refuses to run without a GPU.
 *
 * One workgroup runs at a time; each of its threads is a ucontext fiber.
 * __syncthreads() and every cross-lane primitive are implemented with a
 * "yield until everybody arrived" barrier, so the semantics are those of a
 * lock-step wavefront as long as cross-lane ops are called convergently
 * (which the GPU requires anyway).
 */
#ifndef PMG_HIP_EMU_H
#define PMG_HIP_EMU_H

#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define PMG_EMULATE 1

struct emu_dim3 {
    unsigned x, y, z;
};
extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
#define __constant__ static const

void emu_barrier();         /* wave-level: yield until every fiber has reached its next cross-lane point */
void emu_block_barrier();   /* workgroup-level: every live thread arrives */
#define __syncthreads() emu_block_barrier()

/* exchange buffer for cross-lane primitives */
extern float emu_xf[16 * 64 * 16]; /* [wave][slot][lane] */
extern long long pmge_face_clip_calls;   /* calls of box_face_clip (pmg_contact_body.inc) since load */
extern long long pmge_cyl_contact_calls, pmge_cyl_redo_calls, pmge_cyl_spec_taken_n;   /* cylinder pairs found in contact by the float pass / repeated in double (cyl_redo64) */

namespace emu {
void launch(int grid, int block, const std::function<void()>& body);
}

#endif
