// debug harness (not shipped): shader cycles of one box_box_fast call by the path it takes, one lane per pair, `np` pairs in lockstep.
//   case 0: 3 cm cube flat on a big box (fast path: incident face inside the reference face, 4 points)
//   case 1: finger-sized box half over the edge of a cube's top face (general clipping path through box_box_resume)
//   case 2: two cubes edge to edge, rotated (edge-edge: 1 point)
//   case 3: cube tilted on a cube, corner region (general path, fewer points)
//   case 4: case 0 tilted by 2 mrad (the big box becomes the reference face: the no-clip fast path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "pmg_kernels.h"

__global__ void __launch_bounds__(64, 2) k(int cas, int np, long long* cyc, float* outp, int* nout)
{
    __shared__ float A[64][12], B[64][12], hA[64][3], hB[64][3], out[64][4 * pmg::CP], W[64][pmg::BOX_WORK];
    const int l = threadIdx.x;
    float ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0}, ha[3], hb[3], ya = 0.f, ta = 0.f;
    if (cas == 0) { ha[0] = ha[1] = ha[2] = 0.015f; hb[0] = 0.5f; hb[1] = 0.5f; hb[2] = 0.1f; ca[2] = 0.1149f; ya = 0.3f + 0.01f * l; }
    else if (cas == 1) { ha[0] = 0.0125f; ha[1] = 0.005f; ha[2] = 0.04f; hb[0] = hb[1] = hb[2] = 0.015f; ca[0] = 0.014f; ca[1] = 0.003f; ca[2] = 0.0545f; ya = 0.2f + 0.01f * l; }
    else if (cas == 2) { ha[0] = ha[1] = ha[2] = 0.015f; hb[0] = hb[1] = hb[2] = 0.015f; ca[0] = 0.0208f; ca[1] = 0.0208f; ca[2] = 0.002f; ya = 0.785f; ta = 0.0f; }
    else if (cas == 4) { ha[0] = ha[1] = ha[2] = 0.015f; hb[0] = 0.5f; hb[1] = 0.5f; hb[2] = 0.1f; ca[2] = 0.1149f; ya = 0.3f + 0.01f * l; ta = 0.002f; }   // as case 0, tilted by 2 mrad: the big box is the reference, fast path
    else { ha[0] = ha[1] = ha[2] = 0.015f; hb[0] = hb[1] = hb[2] = 0.015f; ca[0] = 0.012f; ca[1] = 0.011f; ca[2] = 0.0305f; ya = 0.4f + 0.01f * l; ta = 0.08f; }
    {   // A: yaw ya then tilt ta about x; B: identity
        float cy = cosf(ya), sy = sinf(ya), ct = cosf(ta), st = sinf(ta);
        float R[9] = {cy, -sy * ct, sy * st, sy, cy * ct, -cy * st, 0.f, st, ct};
        for (int a = 0; a < 3; a++) { A[l][a] = ca[a]; B[l][a] = cb[a]; hA[l][a] = ha[a]; hB[l][a] = hb[a]; }
        for (int a = 0; a < 9; a++) { A[l][3 + a] = R[a]; B[l][3 + a] = (a % 4 == 0) ? 1.f : 0.f; }
    }
    __syncthreads();
    int n = 0;
    long long t0 = wv::cycles();
    for (int it = 0; it < 50; it++) {
        if (l < np) n = pmg::box_box_fast(A[l], A[l] + 3, hA[l], B[l], B[l] + 3, hB[l], pmg::CONTACT_MARGIN, out[l], W[l]);
        if (l < np) A[l][0] += 1e-7f * n;                  // (keep the calls apart)
        wv::lds_sync();
    }
    long long t1 = wv::cycles();
    if (l == 0) { cyc[0] = (t1 - t0) / 50; nout[0] = n; }
    if (l < np) outp[l] = out[l][9];
}
int main()
{
    long long* c; float* o; int* n;
    (void)hipMalloc(&c, 8); (void)hipMalloc(&o, 256); (void)hipMalloc(&n, 4);
    for (int cas = 0; cas < 5; cas++)
        for (int np : {0, 1, 4}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cas, np, c, o, n);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cas, np, c, o, n);
            (void)hipDeviceSynchronize();
            long long hc; int hn; float ho;
            (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hn, n, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&ho, o, 4, hipMemcpyDeviceToHost);
            printf("case %d, %d pair lane(s): %6lld cycles per call, %d points (dist %.5f)\n", cas, np, hc, hn, ho);
        }
    return 0;
}
