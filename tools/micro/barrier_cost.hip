// Micro-benchmark: cost of a workgroup barrier (+ an LDS write -> barrier -> read hand-off) as a function of the
// workgroup size on gfx950.  One workgroup per CU.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int iters, unsigned long long *out, double *sink) {
    __shared__ double buf[1024];
    const int tid = threadIdx.x;
    double acc = tid;
    buf[tid] = acc;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { __syncthreads(); }
        if (MODE == 1) { buf[tid] = acc; __syncthreads(); acc += buf[(tid + 64) % blockDim.x]; }
        if (MODE == 2) { buf[tid] = acc; __syncthreads(); acc += buf[(tid + 64) % blockDim.x]; acc = acc * 1.0000001 + 0.5; acc = acc * 1.0000001 + 0.5; acc = acc * 1.0000001 + 0.5; acc = acc * 1.0000001 + 0.5; }
    }
    const unsigned long long t1 = clock64();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + tid] = acc;
}
int main() {
    unsigned long long *d; double *s; hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 1024 * 8);
    const int iters = 2000;
    for (int nt : {64, 128, 256, 320, 512, 768, 1024}) {
        for (int mode = 0; mode < 3; mode++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(nt), 0, 0, iters, d, s);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(nt), 0, 0, iters, d, s);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(nt), 0, 0, iters, d, s);
            hipDeviceSynchronize();
            unsigned long long h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < 256; i++) m += h[i]; m /= 256.0;
            printf("threads %4d mode %d : %.1f cycles / iteration\n", nt, mode, m / iters);
        }
    }
    return 0;
}
