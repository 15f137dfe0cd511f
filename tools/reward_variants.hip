// GPU box micro-benchmark: variants of the G = 3 HER-relabel reward kernel (pmg_k_reward3), 64 Mi goal pairs resident in HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/reward_variants.hip -o /tmp/rv && /tmp/rv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4* p) { v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(float4 r, float4* p) { v4f v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, (v4f*)p); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void quad(const float4& a0, const float4& a1, const float4& a2, const float4& d0, const float4& d1, const float4& d2,
                                     float thr, float4& r, unsigned& flags)
{
    float e[12] = {a0.x - d0.x, a0.y - d0.y, a0.z - d0.z, a0.w - d0.w, a1.x - d1.x, a1.y - d1.y,
                   a1.z - d1.z, a1.w - d1.w, a2.x - d2.x, a2.y - d2.y, a2.z - d2.z, a2.w - d2.w};
    float rr[4];
    flags = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float d = sqrtf(e[3 * i] * e[3 * i] + e[3 * i + 1] * e[3 * i + 1] + e[3 * i + 2] * e[3 * i + 2]);
        bool na = d > thr;
        rr[i] = na ? -1.f : -0.f;
        flags |= (na ? 0u : 1u) << (8 * i);
    }
    r = make_float4(rr[0], rr[1], rr[2], rr[3]);
}
// A: the shipped kernel (grid-stride, 256 threads, <= 4096 blocks)
__global__ void __launch_bounds__(256) kA(const float4* ag, const float4* dg, long long quads, float thr, float4* reward, unsigned* ok)
{
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
        float4 r; unsigned f;
        quad(ag[3 * q], ag[3 * q + 1], ag[3 * q + 2], dg[3 * q], dg[3 * q + 1], dg[3 * q + 2], thr, r, f);
        reward[q] = r; ok[q] = f;
    }
}
// B: the block's 3 x 256 float4 per array read as three fully coalesced float4 sweeps (lane = consecutive 16 B), regrouped through LDS
template <int NT>
__global__ void __launch_bounds__(256) kB(const float4* ag, const float4* dg, long long quads, float thr, float4* reward, unsigned* ok)
{
    __shared__ float4 sa[3 * 256], sd[3 * 256];
    const int t = threadIdx.x;
    for (long long base = (long long)blockIdx.x * 256; base < quads; base += (long long)gridDim.x * 256) {
        const long long n = quads - base < 256 ? quads - base : 256;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            long long w = k * 256 + t;
            if (w < 3 * n) {
                sa[w] = NT ? ntload(&ag[3 * base + w]) : ag[3 * base + w];
                sd[w] = NT ? ntload(&dg[3 * base + w]) : dg[3 * base + w];
            }
        }
        __syncthreads();
        if (t < n) {
            float4 r; unsigned f;
            quad(sa[3 * t], sa[3 * t + 1], sa[3 * t + 2], sd[3 * t], sd[3 * t + 1], sd[3 * t + 2], thr, r, f);
            if (NT) { ntstore(r, &reward[base + t]); __builtin_nontemporal_store(f, &ok[base + t]); }
            else { reward[base + t] = r; ok[base + t] = f; }
        }
        __syncthreads();
    }
}
// C: as A with two quads per thread in flight and non-temporal stores
template <int NT>
__global__ void __launch_bounds__(256) kC(const float4* ag, const float4* dg, long long quads, float thr, float4* reward, unsigned* ok)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += 2 * stride) {
        const long long q2 = q + stride;
        const bool two = q2 < quads;
        float4 a0 = ag[3 * q], a1 = ag[3 * q + 1], a2 = ag[3 * q + 2], d0 = dg[3 * q], d1 = dg[3 * q + 1], d2 = dg[3 * q + 2];
        float4 b0 = a0, b1 = a1, b2 = a2, c0 = d0, c1 = d1, c2 = d2;
        if (two) { b0 = ag[3 * q2]; b1 = ag[3 * q2 + 1]; b2 = ag[3 * q2 + 2]; c0 = dg[3 * q2]; c1 = dg[3 * q2 + 1]; c2 = dg[3 * q2 + 2]; }
        float4 r; unsigned f;
        quad(a0, a1, a2, d0, d1, d2, thr, r, f);
        if (NT) { ntstore(r, &reward[q]); __builtin_nontemporal_store(f, &ok[q]); } else { reward[q] = r; ok[q] = f; }
        if (two) {
            quad(b0, b1, b2, c0, c1, c2, thr, r, f);
            if (NT) { ntstore(r, &reward[q2]); __builtin_nontemporal_store(f, &ok[q2]); } else { reward[q2] = r; ok[q2] = f; }
        }
    }
}
int main()
{
    const long long B = 64ll << 20, quads = B / 4;
    float4 *ag, *dg, *rw; unsigned* ok;
    CK(hipMalloc(&ag, B * 12)); CK(hipMalloc(&dg, B * 12)); CK(hipMalloc(&rw, B * 4)); CK(hipMalloc(&ok, B));
    CK(hipMemset(ag, 0, B * 12)); CK(hipMemset(dg, 1, B * 12));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const double bytes = (double)B * (24 + 5);
    auto run = [&](const char* name, auto launch) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 12; rep++) {
            (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-44s best %.3f ms = %.2f TB/s, mean %.3f ms = %.2f TB/s\n", name, best, bytes / best / 1e9, sum / 10, bytes / (sum / 10) / 1e9);
    };
    for (int grid : {2048, 4096, 8192, 16384, 65536})
        run(("A grid-stride, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, ag, dg, quads, 0.05f, rw, ok); });
    for (int grid : {2048, 4096, 8192, 65536}) {
        run(("B LDS-regrouped, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(kB<0>, dim3(grid), dim3(256), 0, 0, ag, dg, quads, 0.05f, rw, ok); });
        run(("B LDS-regrouped nt, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(kB<1>, dim3(grid), dim3(256), 0, 0, ag, dg, quads, 0.05f, rw, ok); });
    }
    for (int grid : {2048, 4096, 8192}) {
        run(("C two quads in flight, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(kC<0>, dim3(grid), dim3(256), 0, 0, ag, dg, quads, 0.05f, rw, ok); });
        run(("C two quads in flight, nt stores, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(kC<1>, dim3(grid), dim3(256), 0, 0, ag, dg, quads, 0.05f, rw, ok); });
    }
    return 0;
}
