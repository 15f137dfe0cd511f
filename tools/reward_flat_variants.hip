// GPU box micro-benchmark: variants of the multi-block HER-relabel reward kernel (pmg_k_reward_flat), 16 Mi goal pairs of
// width G resident in HBM, 8 G + 5 bytes per item.
//   hipcc --offload-arch=gfx950 -O3 tools/reward_flat_variants.hip -o /tmp/rfv && /tmp/rfv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4* p) { v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// F0: the kernel as shipped in round 3
template <int VEC>
__global__ void __launch_bounds__(256) kF0(const float* __restrict__ ag, const float* __restrict__ dg, long long B, int G, float thr, int binary,
                                           float* __restrict__ reward, unsigned char* __restrict__ ok)
{
    __shared__ float part[256 * 20];
    const int t = (int)threadIdx.x;
    const int wpi = G / VEC;
    for (long long base = (long long)blockIdx.x * 256; base < B; base += (long long)gridDim.x * 256) {
        const long long items = B - base < 256 ? B - base : 256;
        const long long words = items * wpi;
        const float* a = ag + base * G;
        const float* d = dg + base * G;
        for (long long w = t; w < words; w += 256) {
            float s;
            if (VEC == 4) {
                float4 x = ((const float4*)a)[w], y = ((const float4*)d)[w];
                float e0 = x.x - y.x, e1 = x.y - y.y, e2 = x.z - y.z, e3 = x.w - y.w;
                s = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            } else {
                float e = a[w] - d[w];
                s = e * e;
            }
            part[w] = s;
        }
        __syncthreads();
        if (t < items) {
            float s = 0.f;
            for (int k = 0; k < wpi; k++) s += part[t * wpi + k];
            float dist = sqrtf(s);
            bool na = dist > thr;
            if (reward) reward[base + t] = binary ? (na ? -1.f : -0.f) : -dist;
            if (ok) ok[base + t] = na ? 0 : 1;
        }
        __syncthreads();
    }
}

// F1: every load of the workgroup's span is a float4 of the FLAT span (256 G floats = 64 G float4 whatever G is), all of a
// thread's loads in flight before the first use; SUM4: G % 4 == 0, a float4 lies inside one item and leaves one partial
// sum, else the four squares go to LDS one by one.  Flags of four items leave as one dword.
// MODE bit 0: non-temporal loads, bit 1: flags packed into dwords, bit 2: rewards as float4 via LDS
template <int SUM4, int MODE, int IPW>
__global__ void __launch_bounds__(256) kF1(const float* __restrict__ ag, const float* __restrict__ dg, long long B, int G, float thr, int binary,
                                           float* __restrict__ reward, unsigned char* __restrict__ ok)
{
    constexpr int R = IPW / 256;                 /* items per thread */
    constexpr int MAXQ = SUM4 ? 5 * R : 5 * R;   /* float4 per thread per array: G <= 20 */
    __shared__ float part[(SUM4 ? 5 : 20) * IPW];
    const int t = (int)threadIdx.x;
    const int q4 = (G * IPW / 4);                /* float4 per array of a full workgroup */
    for (long long base = (long long)blockIdx.x * IPW; base + IPW <= B; base += (long long)gridDim.x * IPW) {
        const float4* a = (const float4*)(ag + base * G);
        const float4* d = (const float4*)(dg + base * G);
        float4 x[MAXQ], y[MAXQ];
#pragma unroll
        for (int k = 0; k < MAXQ; k++) {
            const int w = t + 256 * k;
            if (w < q4) {
                if (MODE & 1) { x[k] = ntload(a + w); y[k] = ntload(d + w); }
                else { x[k] = a[w]; y[k] = d[w]; }
            }
        }
#pragma unroll
        for (int k = 0; k < MAXQ; k++) {
            const int w = t + 256 * k;
            if (w < q4) {
                const float e0 = x[k].x - y[k].x, e1 = x[k].y - y[k].y, e2 = x[k].z - y[k].z, e3 = x[k].w - y[k].w;
                if (SUM4) part[w] = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                else ((float4*)part)[w] = make_float4(e0 * e0, e1 * e1, e2 * e2, e3 * e3);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int it = t + 256 * r;
            float s = 0.f;
            const int wpi = SUM4 ? G / 4 : G;
            for (int k = 0; k < wpi; k++) s += part[it * wpi + k];
            const float dist = sqrtf(s);
            const bool na = dist > thr;
            if (reward) reward[base + it] = binary ? (na ? -1.f : -0.f) : -dist;
            if (ok) {
                if (MODE & 2) {
                    unsigned f = na ? 0u : 1u;
                    f |= (unsigned)__shfl_down((int)f, 1) << 8;
                    f |= (unsigned)__shfl_down((int)f, 2) << 16;
                    if ((t & 3) == 0) ((unsigned*)(ok + base))[it >> 2] = f;
                } else ok[base + it] = na ? 0 : 1;
            }
        }
        __syncthreads();
    }
}

int main()
{
    const long long B = 16ll << 20;
    float *ag, *dg, *rw; unsigned char* ok;
    CK(hipMalloc(&ag, B * 4 * 20)); CK(hipMalloc(&dg, B * 4 * 20)); CK(hipMalloc(&rw, B * 4)); CK(hipMalloc(&ok, B));
    CK(hipMemset(ag, 0, B * 4 * 20)); CK(hipMemset(dg, 1, B * 4 * 20));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int G : {12, 7, 9, 13, 16, 19}) {
        const double bytes = (double)B * (8 * G + 5);
        auto run = [&](const std::string& name, auto launch) {
            float best = 1e9, sum = 0;
            for (int rep = 0; rep < 12; rep++) {
                (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("G %2d %-52s best %.3f ms = %.2f TB/s, mean %.2f TB/s\n", G, name.c_str(), best, bytes / best / 1e9, bytes / (sum / 10) / 1e9);
        };
        for (int grid : {8192, 65536}) {
            if (G % 4 == 0) run("F0<4> shipped, grid " + std::to_string(grid), [&] { hipLaunchKernelGGL((kF0<4>), dim3(grid), dim3(256), 0, 0, ag, dg, B, G, 0.05f, 1, rw, ok); });
            else run("F0<1> shipped, grid " + std::to_string(grid), [&] { hipLaunchKernelGGL((kF0<1>), dim3(grid), dim3(256), 0, 0, ag, dg, B, G, 0.05f, 1, rw, ok); });
        }
#define RUN1(S, M, I, grid) run(std::string("F1 sum4=" #S " mode=" #M " ipw=" #I ", grid ") + std::to_string(grid), [&] { hipLaunchKernelGGL((kF1<S, M, I>), dim3(grid), dim3(256), 0, 0, ag, dg, B, G, 0.05f, 1, rw, ok); })
        for (int grid : {4096, 8192, 16384, 65536}) {
            if (G % 4 == 0) { RUN1(1, 0, 256, grid); RUN1(1, 1, 256, grid); RUN1(1, 2, 256, grid); RUN1(1, 3, 256, grid); RUN1(1, 2, 512, grid); RUN1(1, 3, 512, grid); }
            RUN1(0, 0, 256, grid); RUN1(0, 1, 256, grid); RUN1(0, 2, 256, grid); RUN1(0, 3, 256, grid);
        }
    }
    return 0;
}
