// hardware probe (not shipped): where does the dispatcher put 1024 one-wave workgroups, and what does co-residency on a
// SIMD cost a latency-bound wave?  Every workgroup runs the same dependent chain and records its HW_ID / XCC_ID, start and
// duration; the host prints the SIMD load histogram and the duration by number of co-resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

template <int OCC>
__global__ void __launch_bounds__(64, OCC) k(unsigned* ids, long long* times, float* out, float a, int iters)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = __builtin_readcyclecounter();
    float x = a + threadIdx.x * 1e-3f, y = 0.999f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 64; r++) x = fmaf(x, y, a);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) { ids[2 * blockIdx.x] = hw; ids[2 * blockIdx.x + 1] = xcc; times[2 * blockIdx.x] = t0; times[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int OCC>
void run(int grid, int iters)
{
    unsigned* ids; long long* times; float* out;
    hipMalloc(&ids, grid * 8); hipMalloc(&times, grid * 16); hipMalloc(&out, grid * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OCC><<<grid, 64>>>(ids, times, out, 0.5f, 16);
    hipEventRecord(a);
    k<OCC><<<grid, 64>>>(ids, times, out, 0.5f, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2 * grid); std::vector<long long> t(2 * grid);
    hipMemcpy(h.data(), ids, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(t.data(), times, grid * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    long long tmin = t[0];
    for (int i = 0; i < grid; i++) tmin = std::min(tmin, t[2 * i]);
    std::vector<unsigned> key(grid);
    for (int i = 0; i < grid; i++) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        key[i] = (cukey << 2) | simd;
        per_simd[key[i]]++; per_cu[cukey]++;
    }
    int hist[9] = {0}, cuhist[33] = {0};
    for (auto& p : per_simd) hist[std::min(p.second, 8)]++;
    for (auto& p : per_cu) cuhist[std::min(p.second, 32)]++;
    double dsum[9] = {0}; int dcnt[9] = {0}; long long dmax = 0, late = 0;
    for (int i = 0; i < grid; i++) { int n = std::min(per_simd[key[i]], 8); dsum[n] += t[2 * i + 1]; dcnt[n]++; dmax = std::max(dmax, t[2 * i + 1]); late = std::max(late, t[2 * i] - tmin); }
    printf("occ<=%d grid %5d: kernel %.3f ms | SIMDs used %zu, CUs used %zu | SIMDs by #waves:", OCC, grid, ms, per_simd.size(), per_cu.size());
    for (int n = 1; n <= 8; n++) if (hist[n]) printf(" %dx:%d", n, hist[n]);
    printf(" | CUs by #waves:");
    for (int n = 1; n <= 32; n++) if (cuhist[n]) printf(" %d:%d", n, cuhist[n]);
    printf(" | mean wave cycles by co-residency:");
    for (int n = 1; n <= 8; n++) if (dcnt[n]) printf(" %dx:%.0f", n, dsum[n] / dcnt[n]);
    printf(" | max %lld, latest start +%lld cycles\n", dmax, late);
    hipFree(ids); hipFree(times); hipFree(out);
}

int main()
{
    for (int grid : {256, 512, 1024, 1100, 2048, 4096}) { run<2>(grid, 4000); run<1>(grid, 4000); }
    return 0;
}
