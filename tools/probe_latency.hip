// hardware probe (not shipped): dependent-issue latency of the instruction patterns on the substep's serial chains, one wave
// per SIMD -- the regime of the 4096-env headline.  Prints shader cycles (s_memtime) per operation and the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP 256
template <int KIND>
__global__ void __launch_bounds__(64, 1) k(float* out, long long* cyc, float a, float b)
{
    float x = a + (float)threadIdx.x * 1e-3f, y = b, z = a * 0.5f, w = b * 0.25f;
    int xi = (int)threadIdx.x;
    long long w0 = wall_clock64();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) x = fmaf(x, y, b);                                   // dependent fma chain
            if (KIND == 1) { x = fmaf(x, y, b); z = fmaf(z, y, b); }            // two independent chains
            if (KIND == 2) { x = fmaf(x, y, b); z = fmaf(z, y, b); w = fmaf(w, y, b); a = fmaf(a, y, b); }   // four
            if (KIND == 3) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + 3, 0xF, 0xF, true)) * y + b;   // dpp newbcast + fma
            if (KIND == 4) x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 3)) * y + b;                               // readlane + fma
            if (KIND == 5) x = __int_as_float(__builtin_amdgcn_ds_bpermute(12, __float_as_int(x))) * y + b;                             // bpermute + fma
            if (KIND == 6) { float n = __builtin_amdgcn_fmed3f(x, -b, b); float d = n - z; z = n; x = fmaf(y, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x150 + 3, 0xF, 0xF, true)), x); }   // one fast-sweep visit
            if (KIND == 7) { float n = __builtin_amdgcn_fmed3f(x, -b, b); float d = n - z; z = n; x = fmaf(y, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 3)), x); }   // visit, readlane
            if (KIND == 8) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xF, 0xF, true)) + x;           // row_shr:1 + add (scan step)
            if (KIND == 9) x = sqrtf(fabsf(x)) + b;                                                                                    // sqrt (2.5 ulp build)
            if (KIND == 10) x = b / (x + 2.f);                                                                                         // division
            if (KIND == 11) { float s, c; __sincosf(x, &s, &c); x = s + c * y; }                                                       // fast sincos
            if (KIND == 12) { float s, c; sincosf(x, &s, &c); x = s + c * y; }                                                         // libm sincos
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x + z + w + a + (float)xi;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int KIND>
void run(const char* name, int grid, float* d, long long* c)
{
    k<KIND><<<grid, 64>>>(d, c, 0.37f, 0.911f);
    k<KIND><<<grid, 64>>>(d, c, 0.37f, 0.911f);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double ops = 16.0 * REP;
    printf("%-34s grid %5d: %7.1f cycles/iter  %7.2f ns/iter  (clock %.0f MHz)\n", name, grid, h[0] / ops, h[1] * 10.0 / ops, h[0] / (h[1] * 0.01));
}

int main()
{
    float* d; long long* c;
    hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&c, 16);
    for (int grid : {1024, 2048, 4096}) {
        run<0>("fma chain", grid, d, c);
        run<1>("2 independent fma chains", grid, d, c);
        run<2>("4 independent fma chains", grid, d, c);
        run<3>("dpp row_newbcast + fma", grid, d, c);
        run<4>("v_readlane + fma", grid, d, c);
        run<5>("ds_bpermute + fma", grid, d, c);
        run<6>("sweep visit (med3,sub,dpp,fma)", grid, d, c);
        run<7>("sweep visit (med3,sub,readlane,fma)", grid, d, c);
        run<8>("dpp row_shr + add", grid, d, c);
        run<9>("sqrt + add", grid, d, c);
        run<10>("add + div", grid, d, c);
        run<11>("__sincosf + fma", grid, d, c);
        run<12>("sincosf + fma", grid, d, c);
    }
    return 0;
}
