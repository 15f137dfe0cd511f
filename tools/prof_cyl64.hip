// debug harness (not shipped): shader cycles of the cylinder narrowphase, float pass (cyl_box<float>) against the double repeat
// (cyl_redo64: poses re-derived in double + cyl_box<double>), one lone lane.
//   case 0: the slide puck flat on the table (tilt 1e-6)           -- block state row -> pose, table static
//   case 1: the gripper base's cap rim on the edge of a chest wall -- double forward kinematics + closest-feature pass
//   case 2: fk64_link alone (gripper base)
#define PMG_CYL_REDO64 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "pmg_kernels.h"

__global__ void __launch_bounds__(64, 2) k(int cas, long long* cyc, float* outp, int* nout)
{
    __shared__ float q9[16], blk[16], door[4], out[4 * pmg::CP + 8], W[pmg::BOX_WORK], opsA[12], opsB[12], hb[3], bc[3], kc[24];
    const int l = threadIdx.x;
    if (l == 0) {
        const float q[9] = {0.3f, -0.6f, 0.1f, 1.7f, 0.05f, -0.8f, 0.2f, 0.03f, 0.03f};
        for (int i = 0; i < 9; i++) q9[i] = q[i];
        blk[0] = -0.5f; blk[1] = 0.02f; blk[2] = 0.17f; blk[3] = 5e-7f; blk[4] = -3e-7f; blk[5] = 0.2f; blk[6] = 0.979796f;
        door[0] = 0.01f;
        bc[0] = -0.7f; bc[1] = 0.f; bc[2] = 0.08f; hb[0] = 0.5f; hb[1] = 0.45f; hb[2] = 0.08f;
        for (int i = 0; i < 3; i++) { kc[i] = bc[i]; kc[3 + i] = hb[i]; kc[18 + i] = 0.015f; }
    }
    __syncthreads();
    int n = 0;
    long long t0 = wv::cycles(), t1 = t0, t2 = t0;
    double p[3], R[9];
    for (int it = 0; it < 20; it++) {
        if (l == 0) {
            if (cas == 0) {
                float Rc[9], I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                pmg::quat_to_R(blk + 3, Rc);
                for (int i = 0; i < 9; i++) { opsA[i] = Rc[i]; opsB[i] = I3[i]; }
                n = pmg::cyl_box(blk, opsA, 0.03f, 0.01f, bc, opsB, hb[0], hb[1], hb[2], pmg::CONTACT_MARGIN, out, W);
            }
        }
        wv::lds_sync();
    }
    t1 = wv::cycles();
    for (int it = 0; it < 20; it++) {
        if (l == 0) {
            if (cas == 0) n = pmg::cyl_redo64<-1>(0, pmg::BODY_STATIC, -1, q9, blk, door, kc, 0.03f, 0.01f, out, W);
            if (cas == 2) pmg::fk64_link(q9, pmg::BODY_GBASE, p, R);
            if (cas == 1) n = pmg::cyl_redo64<1>(pmg::BODY_GBASE, pmg::BODY_STATIC, 1, q9, blk, door, kc, 0.05f, 0.02f, out, W);
        }
        if (l == 0 && cas == 2) q9[0] += (float)(1e-9 * p[0] * R[4]);
        wv::lds_sync();
    }
    t2 = wv::cycles();
    if (l == 0) { cyc[0] = (t1 - t0) / 20; cyc[1] = (t2 - t1) / 20; nout[0] = n; outp[0] = out[9]; }
}
int main()
{
    long long* c; float* o; int* n;
    (void)hipMalloc(&c, 16); (void)hipMalloc(&o, 256); (void)hipMalloc(&n, 4);
    for (int cas = 0; cas < 3; cas++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cas, c, o, n);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cas, c, o, n);
        (void)hipDeviceSynchronize();
        long long hc[2]; int hn; float ho;
        (void)hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(&hn, n, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&ho, o, 4, hipMemcpyDeviceToHost);
        printf("case %d: float pass %6lld cycles, double path %6lld cycles per call, %d points (dist %.7f)\n", cas, hc[0], hc[1], hn, ho);
    }
    return 0;
}
