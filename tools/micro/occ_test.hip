// Which resource keeps two 384-thread workgroups from sharing a CU?  Launches 512 workgroups of a spin kernel with a given dynamic LDS size,
// VGPR budget (launch bounds) and scratch use; counts the workgroup pairs that ran on the same CU at overlapping times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
template <int WPE, bool SCRATCH, int VREG = 0>
__global__ __launch_bounds__(384, WPE) void spin(unsigned long long *out, int cycles, int idx_rt) {
    if (VREG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VREG == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    extern __shared__ double lds[];
    const unsigned long long w0 = wall_clock64();
    volatile double priv[SCRATCH ? 64 : 1];
    if (SCRATCH) { for (int i = 0; i < 64; i++) priv[i] = i; }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    double acc = lds[(threadIdx.x + 1) % 384];
    while (clock64() - t0 < cycles) acc = acc * 1.0000001 + 1e-9;
    if (SCRATCH) acc += priv[idx_rt & 63];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + 1] = w0;
        out[blockIdx.x * 4 + 2] = wall_clock64();
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) + (unsigned long long)(acc > 1e300);
    }
}
template <typename K>
static void run(const char *name, K kern, size_t ldsb) {
    const int B = 512;
    unsigned long long *d; hipMalloc(&d, B * 32);
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 384, ldsb);
    hipLaunchKernelGGL(kern, dim3(B), dim3(384), ldsb, 0, d, 2000000, 3);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(B * 4); hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
    std::map<long long, std::vector<int>> g;
    for (int i = 0; i < B; i++) { const unsigned long long hw = h[4 * i]; const long long key = (long long)(h[4 * i + 3] & 0xF) * 100000 + (long long)((hw >> 8) & 0xF) + 100 * ((hw >> 12) & 1) + 1000 * ((hw >> 13) & 7); g[key].push_back(i); }
    int pairs = 0, ov = 0;
    for (auto &kv : g) for (size_t a = 0; a < kv.second.size(); a++) for (size_t c = a + 1; c < kv.second.size(); c++) {
        pairs++; const long long lo = std::max(h[4 * kv.second[a] + 1], h[4 * kv.second[c] + 1]), hi = std::min(h[4 * kv.second[a] + 2], h[4 * kv.second[c] + 2]); if (hi > lo) ov++; }
    printf("%-28s lds %6zu B: occupancy API %d, CUs %zu, pairs on a CU %d, overlapping %d (%s)\n", name, ldsb, nb, g.size(), pairs, ov, hipGetErrorString(e));
    hipFree(d);
}
int main() {
    for (size_t l : {32768, 65536, 73728, 77824, 79696, 81408}) run("wpe3 no scratch", spin<3, false>, l);
    for (size_t l : {32768, 65536, 79696}) run("wpe3 scratch", spin<3, true>, l);
    for (size_t l : {65536, 79696}) run("wpe4 no scratch", spin<4, false>, l);
    run("wpe3 168 vgprs", spin<3, false, 168>, 79696);
    run("wpe3 168 vgprs scratch", spin<3, true, 168>, 79696);
    run("wpe3 160 vgprs", spin<3, false, 160>, 79696);
    run("wpe3 128 vgprs", spin<3, false, 128>, 79696);
    return 0;
}
