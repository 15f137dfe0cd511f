// hardware probe: DPP row_newbcast:n must hand lane n of every 16-lane row to all lanes of that row (wr::bcast relies on it)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void k(int* o) { int v = (int)threadIdx.x * 3 + 7; o[N * 64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x150 + N, 0xF, 0xF, false); }
int main()
{
    int* d; hipMalloc(&d, 16 * 64 * 4); hipMemset(d, 0, 16 * 64 * 4);
    k<0><<<1, 64>>>(d); k<5><<<1, 64>>>(d); k<8><<<1, 64>>>(d); k<15><<<1, 64>>>(d);
    int h[16 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int n : {0, 5, 8, 15}) for (int l = 0; l < 64; l++) { int want = ((l & 48) | n) * 3 + 7; if (h[n * 64 + l] != want) { bad++; if (bad < 8) printf("row_newbcast:%d lane %d got %d want %d\n", n, l, h[n * 64 + l], want); } }
    printf(bad ? "ROW_NEWBCAST_MISMATCH %d\n" : "ROW_NEWBCAST_OK\n", bad);
    return bad != 0;
}
