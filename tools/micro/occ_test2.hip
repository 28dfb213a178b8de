// Map of which (threads per workgroup, VGPRs per lane) let two workgroups with ~80 KB of LDS each share a CU (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
template <int NT, int WPE, int VREG>
__global__ __launch_bounds__(NT, WPE) void spin(unsigned long long *out, int cycles) {
    if (VREG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (VREG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    extern __shared__ double lds[];
    const unsigned long long w0 = wall_clock64();
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    double acc = lds[(threadIdx.x + 1) % NT];
    while (clock64() - t0 < cycles) acc = acc * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + 1] = w0;
        out[blockIdx.x * 4 + 2] = wall_clock64();
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) + (unsigned long long)(acc > 1e300);
    }
}
template <typename K>
static void run(const char *name, K kern, int nt, size_t ldsb) {
    const int B = 512;
    unsigned long long *d; hipMalloc(&d, B * 32);
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), ldsb, 0, d, 2000000);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(B * 4); hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
    std::map<long long, std::vector<int>> g;
    for (int i = 0; i < B; i++) { const unsigned long long hw = h[4 * i]; const long long key = (long long)(h[4 * i + 3] & 0xF) * 100000 + (long long)((hw >> 8) & 0xF) + 100 * ((hw >> 12) & 1) + 1000 * ((hw >> 13) & 7); g[key].push_back(i); }
    int pairs = 0, ov = 0;
    for (auto &kv : g) for (size_t a = 0; a < kv.second.size(); a++) for (size_t c = a + 1; c < kv.second.size(); c++) {
        pairs++; const long long lo = std::max(h[4 * kv.second[a] + 1], h[4 * kv.second[c] + 1]), hi = std::min(h[4 * kv.second[a] + 2], h[4 * kv.second[c] + 2]); if (hi > lo) ov++; }
    printf("%-26s lds %6zu B: CUs %zu, pairs on a CU %d, overlapping %d (%s)\n", name, ldsb, g.size(), pairs, ov, hipGetErrorString(e));
    hipFree(d);
}
int main() {
    const size_t l = 79696;
    run("256 thr, 256 vgprs", spin<256, 2, 256>, 256, l);
    run("256 thr, 168 vgprs", spin<256, 3, 168>, 256, l);
    run("320 thr, 168 vgprs", spin<320, 3, 168>, 320, l);
    run("320 thr, 128 vgprs", spin<320, 4, 128>, 320, l);
    run("384 thr, 168 vgprs", spin<384, 3, 168>, 384, l);
    run("384 thr, 128 vgprs", spin<384, 4, 128>, 384, l);
    run("448 thr, 128 vgprs", spin<448, 4, 128>, 448, l);
    run("512 thr, 128 vgprs", spin<512, 4, 128>, 512, l);
    return 0;
}
