/*
 * pmg_wave.h -- wavefront (64-lane) cross-lane primitives for gfx950 / CDNA4.
 *
 * One environment is simulated by one wavefront.  Lane l < 9 owns movable
 * link / DoF l of the Kuka (iiwa_joint_1..7, finger1, finger2); all the small
 * per-body reductions of the articulated-body maths are done with
 *   - v_readlane_b32 broadcasts (bcast*): lane -> SGPR -> every lane,
 *   - DPP row shifts inside the first 16-lane row (chain prefix / suffix sums),
 *   - DPP butterfly + 4 readlanes for full-wave sums and maxima,
 * never through memory.  (tests/emu/pmg_wave.h is the CPU stand-in the
 * test-only emulator uses; this file is the only one that ships.)
 */
#ifndef PMG_WAVE_H
#define PMG_WAVE_H

#include <hip/hip_runtime.h>

/* non-temporal (streaming) global accesses of the HBM-bound reward kernels: data touched exactly once */
namespace nt {
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4(const float4* p) { v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void store4(float4 r, float4* p) { v4f v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, (v4f*)p); }
__device__ __forceinline__ float load(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void store(unsigned int v, unsigned int* p) { __builtin_nontemporal_store(v, p); }
}  // namespace nt

#ifndef PMG_DOT6_SPLIT
#define PMG_DOT6_SPLIT 0
#endif
namespace wv {
/* a pointer KNOWN to point into LDS: with the assumption the compiler's address-space inference turns every access
 * through it into ds_read / ds_write (an out-of-line function's pointer parameters are generic otherwise: flat_load) */
template <class T>
__device__ __forceinline__ T* as_lds(T* p)
{
#if defined(__HIP_DEVICE_COMPILE__)      /* (the builtin exists in the device pass only) */
    __builtin_assume(__builtin_amdgcn_is_shared((const void*)p));
#endif
    return p;
}
__device__ __forceinline__ long long cycles() { return (long long)__builtin_readcyclecounter(); }   /* shader clock (s_memtime) */
/* issue priority of this wavefront among the wavefronts of its SIMD (s_setprio takes an immediate) */
__device__ __forceinline__ void set_priority(int level)
{
    if (level == 1) __builtin_amdgcn_s_setprio(1);
    else if (level == 2) __builtin_amdgcn_s_setprio(2);
    else if (level >= 3) __builtin_amdgcn_s_setprio(3);
}

constexpr int LANES = 64; /* lanes that cooperate on one env */
/* the lane id as a value the optimiser cannot trace back to threadIdx: per-lane LDS ADDRESSES derived from it are
 * computed where they are used (one v_mad) instead of being hoisted out of the substep loop, spilled to scratch memory
 * and reloaded -- a memory round trip in front of the LDS access -- every substep */
__device__ __forceinline__ int lane_local()
{
    int l = (int)threadIdx.x & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(l));
#endif
    return l;
}
/* (two-wave workgroups: the helper wavefront's lanes.)  PMG_OPAQUE_LANE: every call yields a value the optimiser cannot
 * trace back to threadIdx -- lane masks (l == k) and per-lane LDS addresses are then re-derived where they are used (one
 * or two VALU instructions) instead of being hoisted out of the 100-substep loop as loop invariants, kept live across
 * it, spilled (SGPR pairs to VGPR lanes, addresses to SCRATCH) and fetched back in front of every use */
/* Measured (round 4): VGPR spills 16 -> 0, spilled SGPRs 305 -> 124 (reach), 335 -> 181 (one object), 178 -> 116 (blocks) --
 * and reach 3.83 -> 3.77 M, block_stack-4 0.610 -> 0.596 M, chest_push 0.406 -> 0.395 M: in the tight loops re-deriving a
 * mask costs more issue slots than fetching a spilled SGPR pair back.  Off; lane_local() below is the targeted form. */
#ifndef PMG_OPAQUE_LANE
#define PMG_OPAQUE_LANE 0
#endif
__device__ __forceinline__ int lane()
{
    int l = (int)threadIdx.x & 63;
#if PMG_OPAQUE_LANE && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(l));
#endif
    return l;
}

/* wave-level LDS ordering.  A workgroup is ONE wavefront and the LDS unit executes a wavefront's DS instructions in
 * program order, so a later read sees an earlier write of any lane: all that is needed is to stop the COMPILER from
 * moving memory operations across this point -- a wavefront-scope fence plus the (code-less) wave barrier.  No
 * s_barrier: the callers sit in control flow that diverges by row in the packed layout, where a workgroup barrier
 * would be undefined. */
/* Two-wavefront workgroups count exactly two workgroup barriers per substep on either wavefront (helper_wave_loop): a
 * build that turns the wave-level fences into __syncthreads, or adds the profile build's barriers to wavefront 0 only,
 * would pair them up wrongly.  Such builds must switch the two-wave kernels off (tools/prof_k.hip instantiates the
 * one-wavefront kernels only and says so). */
#if defined(PMG_LDS_SYNC_BARRIER) && !defined(PMG_NO_TWO_WAVE_KERNELS)
#error "PMG_LDS_SYNC_BARRIER is incompatible with the two-wavefront workgroups: also define PMG_NO_TWO_WAVE_KERNELS (and PMG_LIST_TWO_WAVES=0)"
#endif
__device__ __forceinline__ void lds_sync()
{
#ifdef PMG_LDS_SYNC_BARRIER
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

/* value of lane `src` (wave-uniform) in every lane */
__device__ __forceinline__ float bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
/* broadcast of a compile-time lane; the one on the serial chain of the Gauss-Seidel visits */
template <int SRC>
__device__ __forceinline__ float bcast_c(float v) { return bcast(v, SRC); }

template <int N>
__device__ __forceinline__ void bcastn(const float* v, int src, float* out)
{
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = bcast(v[k], src);
}
/* acc1 += c1 * v[lane SRC], acc2 += c2 * v[lane SRC]: the tail of a Gauss-Seidel visit (one v_readlane, two fmas) */
template <int SRC>
__device__ __forceinline__ void fma2_bcast_c(float v, float c1, float& acc1, float c2, float& acc2)
{
    const float b = bcast_c<SRC>(v);
    acc1 = fmaf(c1, b, acc1);
    acc2 = fmaf(c2, b, acc2);
}
/* HALF-ROW broadcasts: lane SRC8 (0..7) of every 8-lane half row to its own half row -- two bank-masked DPP
 * row_newbcast operations (banks 0-1 take lane SRC8 of the 16-lane row, banks 2-3 lane 8 + SRC8).  A block of the
 * multi-block layouts owns one half row (pmg_contact_body.inc: LaneDof). */
template <int SRC8>
__device__ __forceinline__ float half_bcast_c(float v)
{
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + SRC8, 0xF, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x158 + SRC8, 0xF, 0xC, false);
    return __int_as_float(r);
}
/* acc += c * v[lane SRC8 of my half row] */
template <int SRC8>
__device__ __forceinline__ void half_fma_bcast_c(float v, float c, float& acc)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\t"
                 "v_fmac_f32_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc"
                 : "+v"(acc)
                 : "v"(v), "v"(c), "n"(SRC8), "n"(8 + SRC8));
}
/* does p hold in any lane of the env?  (wave-uniform here) */
__device__ __forceinline__ bool any_lane(bool p) { return __ballot(p) != 0ull; }

/* ROW-0 broadcasts.  The robot's nine DoFs live in lanes 0..8, i.e. inside the first 16-lane DPP row, and most of the
 * robot maths is consumed there only.  For those call sites lane SRC is handed out by ONE DPP row_newbcast instead of
 * v_readlane -> SGPR -> operand (measured: ~25 cycles per dependent readlane + use, ~12 for the DPP form, which can also
 * ride on the consuming v_fmac).  Lanes 16..63 receive lane SRC of THEIR row: garbage by contract -- callers that need
 * the value wave-wide (FK hand-over to the pair lanes, tip frame, row-space contact lanes) keep bcast(). */
#ifndef PMG_NO_R0_DPP
template <int SRC>
__device__ __forceinline__ float bcast_r0_c(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + SRC, 0xF, 0xF, true)); }
__device__ __forceinline__ float bcast_r0(float v, int src)
{
    if (__builtin_constant_p(src)) {
        switch (src & 15) {
        case 0: return bcast_r0_c<0>(v);   case 1: return bcast_r0_c<1>(v);   case 2: return bcast_r0_c<2>(v);   case 3: return bcast_r0_c<3>(v);
        case 4: return bcast_r0_c<4>(v);   case 5: return bcast_r0_c<5>(v);   case 6: return bcast_r0_c<6>(v);   case 7: return bcast_r0_c<7>(v);
        case 8: return bcast_r0_c<8>(v);   case 9: return bcast_r0_c<9>(v);   case 10: return bcast_r0_c<10>(v); case 11: return bcast_r0_c<11>(v);
        case 12: return bcast_r0_c<12>(v); case 13: return bcast_r0_c<13>(v); case 14: return bcast_r0_c<14>(v); default: return bcast_r0_c<15>(v);
        }
    }
    return bcast(v, src);
}
template <int SRC>
__device__ __forceinline__ void fma2_bcast_r0_c(float v, float c1, float& acc1, float c2, float& acc2)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc1), "+v"(acc2)
                 : "v"(v), "v"(c1), "v"(c2), "n"(SRC));
}
#else
template <int SRC>
__device__ __forceinline__ float bcast_r0_c(float v) { return bcast_c<SRC>(v); }
__device__ __forceinline__ float bcast_r0(float v, int src) { return bcast(v, src); }
template <int SRC>
__device__ __forceinline__ void fma2_bcast_r0_c(float v, float c1, float& acc1, float c2, float& acc2) { fma2_bcast_c<SRC>(v, c1, acc1, c2, acc2); }
#endif
template <int N>
__device__ __forceinline__ void bcastn_r0(const float* v, int src, float* out)
{
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = bcast_r0(v[k], src);
}

/* Row-local fused broadcast arithmetic of the mass-matrix code (CRBA entries, Gauss-Jordan): the lane-SRC operand rides on
 * the DPP port of the multiply-add itself (row_newbcast) -- the compiler leaves most such broadcasts as a v_mov_b32_dpp in
 * front of the fma (475 of them in the reach kernel).  One s_nop 1 per block: the 2 wait states a DPP read needs after a
 * VALU write of its source, which the hazard recognizer cannot see through the asm (tools/check_dpp_hazards.py checks the
 * shipped binary). */
/* sum_a v[a](lane SRC of my row) * c[a], a < 6, summed in index order */
template <int SRC>
__device__ __forceinline__ float dot6_bcast_r0_c(const float* v, const float* c)
{
#if !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC) && PMG_DOT6_SPLIT
    float acc, a1;   /* two interleaved chains */
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %2, %8 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %3, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %5, %11 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %6, %12 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %7, %13 row_newbcast:%14 row_mask:0xf bank_mask:0xf"
        : "=&v"(acc), "=&v"(a1)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "n"(SRC));
    return acc + a1;
#elif !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC)
    float acc;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %1, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %5, %11 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %6, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
        : "=&v"(acc)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "n"(SRC));
    return acc;
#else
    float acc = bcast_r0_c<SRC>(v[0]) * c[0];
#pragma unroll
    for (int a = 1; a < 6; a++) acc = fmaf(bcast_r0_c<SRC>(v[a]), c[a], acc);
    return acc;
#endif
}
/* a[j] += f * a[j](lane P of my row) for the eight j != P of a nine-entry row, j = P+1 first (the next pivot's column is the
 * oldest write when the next block reads it) */
template <int P>
__device__ __forceinline__ void gj9_eliminate_r0_c(float* a, float f)
{
#if !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC)
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(a[(P + 1) % 9]), "+v"(a[(P + 2) % 9]), "+v"(a[(P + 3) % 9]), "+v"(a[(P + 4) % 9]), "+v"(a[(P + 5) % 9]), "+v"(a[(P + 6) % 9]),
          "+v"(a[(P + 7) % 9]), "+v"(a[(P + 8) % 9])
        : "v"(f), "n"(P));
#else
#pragma unroll
    for (int k = 1; k < 9; k++) { const int j = (P + k) % 9; a[j] = fmaf(bcast_r0_c<P>(a[j]), f, a[j]); }
#endif
}
/* One Gauss-Jordan pivot of a 6 x 6 system [A | e], lane a < 6 = row a (the IK's damped least squares): every row loses
 * f x row_P in its five other columns and its right-hand side, f as in gj9_eliminate_r0_c */
template <int P>
__device__ __forceinline__ void gj6_eliminate_r0_c(float* a, float& e, float f)
{
#if !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC)
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(a[(P + 1) % 6]), "+v"(a[(P + 2) % 6]), "+v"(a[(P + 3) % 6]), "+v"(a[(P + 4) % 6]), "+v"(a[(P + 5) % 6]), "+v"(e)
        : "v"(f), "n"(P));
#else
#pragma unroll
    for (int k = 1; k < 6; k++) { const int j = (P + k) % 6; a[j] = fmaf(bcast_r0_c<P>(a[j]), f, a[j]); }
    e = fmaf(bcast_r0_c<P>(e), f, e);
#endif
}
/* sum_a c[a] * v(lane a of my row), a < 6 */
__device__ __forceinline__ float dot6_lanes_r0(const float* c, float v)
{
#if !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC)
    float acc;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf"
        : "=&v"(acc)
        : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
    return acc;
#else
    float acc = bcast_r0_c<0>(v) * c[0];
    acc = fmaf(bcast_r0_c<1>(v), c[1], acc); acc = fmaf(bcast_r0_c<2>(v), c[2], acc); acc = fmaf(bcast_r0_c<3>(v), c[3], acc);
    acc = fmaf(bcast_r0_c<4>(v), c[4], acc); acc = fmaf(bcast_r0_c<5>(v), c[5], acc);
    return acc;
#endif
}
/* lanes 8..11 of every row: y <- y(lane l - 2) + x; the other lanes keep y (the finger-2 step of chain_prefix) */
__device__ __forceinline__ float add_shr2_bank2(float y, float x)
{
#if !defined(PMG_NO_R0_DPP) && !defined(PMG_NO_DPP_FMAC)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0x4"
        : "+v"(y)
        : "v"(x));
    return y;
#else
    const float y2 = row_shr<2>(y, 0.f);
    const int rl = (int)threadIdx.x & 15;
    return (rl >= 8 && rl < 12) ? y2 + x : y;
#endif
}
/* sqrt by the hardware instruction alone (1 ulp, no rescaling of denormal arguments: they flush to zero -- used for the
 * speeds of the link damping terms, which enter as 1 + |v|) */
__device__ __forceinline__ float fsqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
/* 1 / x by the hardware reciprocal (1 ulp; the compiler's 2.5-ulp division is a frexp / rcp / ldexp sequence of six) */
__device__ __forceinline__ float rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f / x;
#endif
}

template <int CTRL>
__device__ __forceinline__ float dpp(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
/* lane i <- lane i-N inside its 16-lane row, `fill` where there is no source */
template <int N>
__device__ __forceinline__ float row_shr(float v, float fill) { return dpp<0x110 + N>(fill, v); }
/* lane i <- lane i+N inside its 16-lane row */
template <int N>
__device__ __forceinline__ float row_shl(float v, float fill) { return dpp<0x100 + N>(fill, v); }

/* sum over each 16-lane row, result in every lane of the row */
__device__ __forceinline__ float row_sum(float v)
{
    v += dpp<0xB1>(0.f, v);  /* quad_perm [1,0,3,2] */
    v += dpp<0x4E>(0.f, v);  /* quad_perm [2,3,0,1] */
    v += dpp<0x141>(0.f, v); /* row_half_mirror */
    v += dpp<0x140>(0.f, v); /* row_mirror */
    return v;
}
/* sum over each 8-lane half row (lanes 8h .. 8h+7), result in every lane of the half row */
__device__ __forceinline__ float half_sum(float v)
{
    v += dpp<0xB1>(0.f, v);  /* quad_perm [1,0,3,2] */
    v += dpp<0x4E>(0.f, v);  /* quad_perm [2,3,0,1] */
    v += dpp<0x141>(0.f, v); /* row_half_mirror */
    return v;
}
__device__ __forceinline__ float row_max(float v)
{
    v = fmaxf(v, dpp<0xB1>(v, v));
    v = fmaxf(v, dpp<0x4E>(v, v));
    v = fmaxf(v, dpp<0x141>(v, v));
    v = fmaxf(v, dpp<0x140>(v, v));
    return v;
}
/* sum over lanes 0..15 only (robot DoFs live there), uniform result */
__device__ __forceinline__ float sum_row0(float v) { return bcast(row_sum(v), 0); }
__device__ __forceinline__ float max_row0(float v) { return bcast(row_max(v), 0); }
/* sum over the first NR 16-lane rows, uniform result */
template <int NR>
__device__ __forceinline__ float sum_rows(float v)
{
    v = row_sum(v);
    float s = bcast(v, 0);
#pragma unroll
    for (int r = 1; r < NR; r++) s += bcast(v, 16 * r);
    return s;
}
/* sum over all 64 lanes through the wave-level DPP broadcasts (row_bcast:15 hands lane 15 of rows 0 / 2 to rows 1 / 3,
 * row_bcast:31 lane 31 to rows 2 / 3; both exist on gfx950, tools/probe_dpp.hip): row butterfly, two DPP adds, ONE
 * v_readlane of lane 63 -- instead of four v_readlane and three dependent adds (sum_all).  Uniform result. */
__device__ __forceinline__ float sum_wave(float v)
{
    v = row_sum(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return bcast(v, 63);
}
/* sum over all 64 lanes, uniform result */
__device__ __forceinline__ float sum_all(float v)
{
    v = row_sum(v);
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
__device__ __forceinline__ float max_all(float v)
{
    v = row_max(v);
    return fmaxf(fmaxf(bcast(v, 0), bcast(v, 16)), fmaxf(bcast(v, 32), bcast(v, 48)));
}
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
/* lanes 0..15 of the env(s) of this wavefront where p holds, wave-uniform: here one env = the whole wave */
__device__ __forceinline__ unsigned any_row_mask(bool p) { return (unsigned)(__ballot(p) & 0xFFFFull); }
/* v holds the same value in every lane: test v > 0 on the scalar unit (one v_readfirstlane) */
__device__ __forceinline__ bool uniform_positive(float v) { return __builtin_amdgcn_readfirstlane(__float_as_int(v)) > 0; }
/* optimisation barrier on a per-lane index: everything loaded through it is re-loaded */
__device__ __forceinline__ void opaque(int& i) { asm volatile("" : "+v"(i)); }
/* lane LANE takes a, every other lane keeps b.  The predicate is a compile-time CONSTANT lane mask handed to v_cndmask in
 * an SGPR pair (one or two s_mov) -- written as `l == LANE ? a : b` the compare is hoisted out of the substep loop as a
 * loop invariant, one SGPR pair per distinct lane, and with 24 contact rows + 9 DoFs those pairs are spilled to VGPR lanes
 * and fetched back with two v_readlane in front of every visit (ISA census of the reach kernel: 240 per contact substep) */
/* Measured (round 4): reach 4.07 -> 3.96 M, pick_and_place 2.04 -> 2.01 M with the constant masks -- the s_mov_b64 + the wait
 * state between an SALU write and the VALU read of the pair cost more than the two v_readlane they replace.  Off. */
#ifndef PMG_CONST_LANE_MASKS
#define PMG_CONST_LANE_MASKS 0
#endif
__device__ __forceinline__ float sel_lane(float a, float b, int l, int lane_k)   /* lane_k: a constant once the caller's loop is unrolled */
{
#if PMG_CONST_LANE_MASKS && defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(1ull << lane_k));
    return r;
#else
    return l == lane_k ? a : b;
#endif
}
/* Software pipelining by data dependence: an address offset (always 0) that the compiler must take to depend on `done`.
 * LDS reads addressed through it cannot be issued before `done` has been computed -- the one fence the iterative-ilp
 * machine scheduler respects (it hoists the reads of a fully unrolled loop across sched_barrier and across asm memory
 * clobbers alike: 72 reads in flight, 110 spilled registers). */
__device__ __forceinline__ void chain(int& off, float& done) { asm volatile("" : "+v"(off), "+v"(done)); }

}  // namespace wv

/* Row-packed variant: FOUR environments per wavefront, one per 16-lane DPP row (the contact-free reach path).
 * Same API as wv with every lane index taken inside the caller's row: the DPP shifts / butterflies of wv are
 * row-local already; broadcasts go through the LDS crossbar (ds_bpermute, no LDS memory) because a
 * v_readlane would hand all four rows the value of ONE of them.  Control flow that is uniform per env
 * (solver early exits, IK iteration counts) simply diverges by row. */
namespace wr {

constexpr int LANES = 16; /* lanes that cooperate on one env */
__device__ __forceinline__ int lane_local()
{
    int l = (int)threadIdx.x & 15;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(l));
#endif
    return l;
}
__device__ __forceinline__ int lane()
{
    int l = (int)threadIdx.x & 15;
#if PMG_OPAQUE_LANE && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(l));
#endif
    return l;
}
__device__ __forceinline__ int row() { return ((int)threadIdx.x >> 4) & 3; }
template <class T>
__device__ __forceinline__ T* as_lds(T* p) { return wv::as_lds(p); }   /* (& 3: the second wavefront of a two-wave workgroup) */
__device__ __forceinline__ void lds_sync() { wv::lds_sync(); }

/* lane SRC (compile time) of the caller's row in every lane of the row: ONE DPP move (row_newbcast, gfx90a+), which the
 * compiler may fold into the consuming VALU instruction -- no LDS crossbar round trip */
template <int SRC>
__device__ __forceinline__ int bcast_ci(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + SRC, 0xF, 0xF, true); }
__device__ __forceinline__ int bcast_i(int v, int src)
{
#ifndef PMG_NO_NEWBCAST
    /* the robot maths calls this from fully unrolled loops: src is a constant by the time the check is lowered */
    if (__builtin_constant_p(src)) {
        switch (src & 15) {
        case 0: return bcast_ci<0>(v);   case 1: return bcast_ci<1>(v);   case 2: return bcast_ci<2>(v);   case 3: return bcast_ci<3>(v);
        case 4: return bcast_ci<4>(v);   case 5: return bcast_ci<5>(v);   case 6: return bcast_ci<6>(v);   case 7: return bcast_ci<7>(v);
        case 8: return bcast_ci<8>(v);   case 9: return bcast_ci<9>(v);   case 10: return bcast_ci<10>(v); case 11: return bcast_ci<11>(v);
        case 12: return bcast_ci<12>(v); case 13: return bcast_ci<13>(v); case 14: return bcast_ci<14>(v); default: return bcast_ci<15>(v);
        }
    }
#endif
    return __builtin_amdgcn_ds_bpermute((((int)threadIdx.x & 48) | src) << 2, v);
}
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(bcast_i(__float_as_int(v), src)); }
template <int N>
__device__ __forceinline__ void bcastn(const float* v, int src, float* out)
{
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = bcast(v[k], src);
}
/* broadcast of a COMPILE-TIME row lane, for the one value that sits on the serial chain of every Gauss-Seidel visit */
template <int SRC>
__device__ __forceinline__ float bcast_c(float v)
{
#ifndef PMG_NO_NEWBCAST
    return __int_as_float(bcast_ci<SRC>(__float_as_int(v)));
#else
    /* a quad broadcast, then the source quad rotated into the other quads with bank-masked DPP moves -- 3 dependent
     * VALU operations instead of an LDS crossbar round trip.  Quad 3 is not written. */
    constexpr int d = SRC & 3, Q = SRC >> 2;
    const int q0 = __builtin_amdgcn_update_dpp(0, __float_as_int(v), d | (d << 2) | (d << 4) | (d << 6), 0xF, 0xF, false);
    int r = q0;
    if (((Q + 1) & 3) != 3) r = __builtin_amdgcn_update_dpp(r, q0, 0x124, 0xF, 1 << ((Q + 1) & 3), false);
    if (((Q + 2) & 3) != 3) r = __builtin_amdgcn_update_dpp(r, q0, 0x128, 0xF, 1 << ((Q + 2) & 3), false);
    if (((Q + 3) & 3) != 3) r = __builtin_amdgcn_update_dpp(r, q0, 0x12C, 0xF, 1 << ((Q + 3) & 3), false);
    return __int_as_float(r);
#endif
}
/* acc1 += c1 * v[row lane SRC], acc2 += c2 * v[row lane SRC]: the broadcast rides on the DPP operand of the two
 * multiply-adds (v_fmac_f32_dpp; the compiler does not fold row_newbcast itself).  The s_nop covers the two wait states
 * a DPP read needs after the VALU write of v, which the compiler cannot see through the asm. */
template <int SRC>
__device__ __forceinline__ void fma2_bcast_c(float v, float c1, float& acc1, float c2, float& acc2)
{
#if !defined(PMG_NO_NEWBCAST) && !defined(PMG_NO_DPP_FMAC)
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc1), "+v"(acc2)
                 : "v"(v), "v"(c1), "v"(c2), "n"(SRC));
#else
    const float b = bcast_c<SRC>(v);
    acc1 = fmaf(c1, b, acc1);
    acc2 = fmaf(c2, b, acc2);
#endif
}
/* does p hold in any lane of the caller's env (= row)?  Per lane, diverges by row */
__device__ __forceinline__ bool any_lane(bool p) { return ((__ballot(p) >> ((int)threadIdx.x & 48)) & 0xFFFFull) != 0ull; }
/* (the multi-block layouts are never packed; these keep the shared sources compiling) */
template <int SRC8>
__device__ __forceinline__ float half_bcast_c(float v) { return wv::half_bcast_c<SRC8>(v); }
template <int SRC8>
__device__ __forceinline__ void half_fma_bcast_c(float v, float c, float& acc) { wv::half_fma_bcast_c<SRC8>(v, c, acc); }
/* "row 0" of an env IS its row here */
template <int SRC>
__device__ __forceinline__ float bcast_r0_c(float v) { return bcast_c<SRC>(v); }
__device__ __forceinline__ float bcast_r0(float v, int src) { return bcast(v, src); }
template <int N>
__device__ __forceinline__ void bcastn_r0(const float* v, int src, float* out) { bcastn<N>(v, src, out); }
template <int SRC>
__device__ __forceinline__ void fma2_bcast_r0_c(float v, float c1, float& acc1, float c2, float& acc2) { fma2_bcast_c<SRC>(v, c1, acc1, c2, acc2); }
template <int SRC>
__device__ __forceinline__ float dot6_bcast_r0_c(const float* v, const float* c) { return wv::dot6_bcast_r0_c<SRC>(v, c); }   /* (row_newbcast is row-local) */
template <int P>
__device__ __forceinline__ void gj9_eliminate_r0_c(float* a, float f) { wv::gj9_eliminate_r0_c<P>(a, f); }
__device__ __forceinline__ float rcp(float x) { return wv::rcp(x); }
template <int P>
__device__ __forceinline__ void gj6_eliminate_r0_c(float* a, float& e, float f) { wv::gj6_eliminate_r0_c<P>(a, e, f); }
__device__ __forceinline__ float dot6_lanes_r0(const float* c, float v) { return wv::dot6_lanes_r0(c, v); }
__device__ __forceinline__ float fsqrt(float x) { return wv::fsqrt(x); }
__device__ __forceinline__ float add_shr2_bank2(float y, float x) { return wv::add_shr2_bank2(y, x); }
template <int N>
__device__ __forceinline__ float row_shr(float v, float fill) { return wv::row_shr<N>(v, fill); }
template <int N>
__device__ __forceinline__ float row_shl(float v, float fill) { return wv::row_shl<N>(v, fill); }
__device__ __forceinline__ float row_sum(float v) { return wv::row_sum(v); }
__device__ __forceinline__ float half_sum(float v) { return wv::half_sum(v); }
__device__ __forceinline__ float row_max(float v) { return wv::row_max(v); }
__device__ __forceinline__ float sum_row0(float v) { return wv::row_sum(v); } /* "row 0" = the caller's own row */
__device__ __forceinline__ float max_row0(float v) { return wv::row_max(v); }
/* an env owns exactly one row here: "all lanes of the env" = the row */
template <int NR>
__device__ __forceinline__ float sum_rows(float v) { return wv::row_sum(v); }
__device__ __forceinline__ float sum_all(float v) { return wv::row_sum(v); }
__device__ __forceinline__ float sum_wave(float v) { return wv::row_sum(v); }
__device__ __forceinline__ float max_all(float v) { return wv::row_max(v); }
/* v is uniform over the ROW: the test stays per lane and control flow diverges by row */
__device__ __forceinline__ bool uniform_positive(float v) { return v > 0.f; }
/* predicate mask of the caller's row (bit i = row lane i) */
__device__ __forceinline__ unsigned long long ballot(bool p) { return (__ballot(p) >> ((int)threadIdx.x & 48)) & 0xFFFFull; }
/* row lanes where p holds in ANY of the four envs of the wavefront: wave-uniform (a scalar), unlike ballot() */
__device__ __forceinline__ unsigned any_row_mask(bool p)
{
    const unsigned long long b = __ballot(p);
    return (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xFFFFull);
}
__device__ __forceinline__ void opaque(int& i) { asm volatile("" : "+v"(i)); }
__device__ __forceinline__ void chain(int& off, float& done) { wv::chain(off, done); }
__device__ __forceinline__ float sel_lane(float a, float b, int l, int lane_k) { return l == lane_k ? a : b; }   /* (lane_k of EACH 16-lane row) */

}  // namespace wr
#endif
