// Probe: which source lane does each DPP control deliver?  (diagnostic; prints lane maps of one 16-lane row)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__device__ int probe(int lane) {
    return __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xf, BANK, false);
}
__global__ void k(int *out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = probe<0x104, 0xf>(l);   // row_shl:4
    out[1 * 64 + l] = probe<0x114, 0xf>(l);   // row_shr:4
    out[2 * 64 + l] = probe<0x124, 0xf>(l);   // row_ror:4
    out[3 * 64 + l] = probe<0x128, 0xf>(l);   // row_ror:8
    out[4 * 64 + l] = probe<0x104, 0x5>(l);   // row_shl:4, banks 0,2
    out[5 * 64 + l] = probe<0x114, 0xA>(l);   // row_shr:4, banks 1,3
    int t = __builtin_amdgcn_update_dpp(-1, l, 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, l, 0x114, 0xf, 0xA, false);
    out[6 * 64 + l] = t;                      // intended: lane ^ 4
    out[7 * 64 + l] = probe<0x141, 0xf>(l);   // row_half_mirror
}
int main() {
    int *d, h[8 * 64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[8] = {"row_shl:4", "row_shr:4", "row_ror:4", "row_ror:8", "shl4 banks 0,2", "shr4 banks 1,3", "combined (xor 4?)", "half_mirror"};
    for (int r = 0; r < 8; r++) { printf("%-18s", nm[r]); for (int l = 0; l < 16; l++) printf(" %3d", h[r * 64 + l]); printf("\n"); }
    return 0;
}
