// Micro-benchmark: FP64 vector FMA peak of the device (v_fma_f64), the `peak` the bench's roofline is priced against.
// SURVEY.md 8(d) quotes 78.6 TFLOP/s from the datasheet (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz) and asks for a
// measurement.  Every wave runs 16 independent FMA chains (no memory traffic inside the loop); sweeps waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fp64_peak.hip -o tools/micro/fp64_peak.bin && tools/micro/fp64_peak.bin
// prints one JSON line {"tflops": best, ...} (copied to profiles/r02_fp64_peak.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_fma(int iters, double seed, double *sink) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-9 + i;
    const double m = 1.0 - 1e-9, c = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_fma(a[i], m, c);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    if (s == 12345.678) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;     // never true: keeps the chains alive
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double *sink; hipMalloc(&sink, sizeof(double) * 1024 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0; int best_w = 0;
    const int iters = 20000;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"sweep\": [", p.name, cus, p.clockRate / 1000);
    bool first = true;
    for (int wg_per_cu : {1, 2, 4, 8}) {                 // 256-thread workgroups: 1..8 waves per SIMD
        const int grid = cus * wg_per_cu;
        hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, 100, 1.0, sink);      // warm-up
        hipDeviceSynchronize();
        double tf_best = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, iters, 1.0, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 16 * 8 * (double)iters * 256.0 * grid;
            const double tf = flops / (ms * 1e-3) / 1e12;
            if (tf > tf_best) tf_best = tf;
        }
        printf("%s{\"waves_per_simd\": %d, \"tflops\": %.2f}", first ? "" : ", ", wg_per_cu, tf_best);
        first = false;
        if (tf_best > best) { best = tf_best; best_w = wg_per_cu; }
    }
    printf("], \"tflops\": %.2f, \"best_waves_per_simd\": %d, \"spec_tflops\": 78.6, \"kernel\": \"16 independent v_fma_f64 chains per lane, no memory traffic\"}\n", best, best_w);
    return 0;
}
