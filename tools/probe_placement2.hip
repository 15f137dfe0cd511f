// hardware probe (not shipped): where do the wavefronts of TWO-wave workgroups land?  grid x 128 threads, LDS bytes per
// workgroup as given; every wavefront runs the same dependent chain and records HW_ID / XCC_ID and its duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(128, 2) k(unsigned* ids, long long* times, float* out, float a, int iters, int second_works)
{
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int w = 2 * blockIdx.x + (threadIdx.x >> 6);
    long long t0 = __builtin_readcyclecounter();
    float x = a + threadIdx.x * 1e-3f, y = 0.999f;
    if ((threadIdx.x >> 6) == 0 || second_works) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 64; r++) x = fmaf(x, y, a);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (a < 0.f) lds[threadIdx.x] = x;
    out[blockIdx.x * 128 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) { ids[2 * w] = hw; ids[2 * w + 1] = xcc; times[2 * w] = t0; times[2 * w + 1] = t1 - t0; }
}

void run(int grid, int iters, int lds_bytes, int second_works)
{
    const int nw = 2 * grid;
    unsigned* ids; long long* times; float* out;
    hipMalloc(&ids, nw * 8); hipMalloc(&times, nw * 16); hipMalloc(&out, grid * 512);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    k<<<grid, 128, lds_bytes>>>(ids, times, out, 0.5f, 16, second_works);
    hipEventRecord(a);
    k<<<grid, 128, lds_bytes>>>(ids, times, out, 0.5f, iters, second_works);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2 * nw); std::vector<long long> t(2 * nw);
    hipMemcpy(h.data(), ids, nw * 8, hipMemcpyDeviceToHost); hipMemcpy(t.data(), times, nw * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    std::vector<unsigned> key(nw);
    int same_simd = 0;
    for (int i = 0; i < nw; i++) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        key[i] = (cukey << 2) | simd;
        per_simd[key[i]]++; per_cu[cukey]++;
        if ((i & 1) && key[i] == key[i - 1]) same_simd++;
    }
    int hist[9] = {0}, cuhist[33] = {0};
    for (auto& p : per_simd) hist[std::min(p.second, 8)]++;
    for (auto& p : per_cu) cuhist[std::min(p.second, 32)]++;
    double dsum[9] = {0}; int dcnt[9] = {0}; long long dmax = 0;
    for (int i = 0; i < nw; i += (second_works ? 1 : 2)) { int n = std::min(per_simd[key[i]], 8); dsum[n] += t[2 * i + 1]; dcnt[n]++; dmax = std::max(dmax, t[2 * i + 1]); }
    printf("grid %5d x 128, LDS %6d B, wave 1 %s: kernel %.3f ms | SIMDs used %zu, CUs %zu | SIMDs by #waves:", grid, lds_bytes, second_works ? "works" : "idles", ms, per_simd.size(), per_cu.size());
    for (int n = 1; n <= 8; n++) if (hist[n]) printf(" %dx:%d", n, hist[n]);
    printf(" | CUs by #waves:");
    for (int n = 1; n <= 32; n++) if (cuhist[n]) printf(" %d:%d", n, cuhist[n]);
    printf(" | both waves of a workgroup on one SIMD: %d | mean working-wave cycles by co-residency:", same_simd);
    for (int n = 1; n <= 8; n++) if (dcnt[n]) printf(" %dx:%.0f", n, dsum[n] / dcnt[n]);
    printf(" | max %lld\n", dmax);
    hipFree(ids); hipFree(times); hipFree(out);
}

int main()
{
    for (int lds : {1024, 32000, 40960})
        for (int grid : {512, 1024})
            for (int sw : {0, 1}) run(grid, 4000, lds, sw);
    return 0;
}
