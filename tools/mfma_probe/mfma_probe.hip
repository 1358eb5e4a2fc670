// Tuning aid (not part of the product): what the f64 matrix pipe of this part sustains when nothing but MFMAs is issued --
// every wave runs NIT x 8 independent v_mfma_f64_16x16x4_f64 from registers -- and the shader clock it runs at meanwhile
// (clock64() = shader cycles, wall_clock64() = constant 100 MHz).  Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) probe(int nit, double *out, long long *clk) {
    d4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < nit; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));      // (the builtin form made the
        // compiler shuttle the accumulators between VGPRs and AGPRs every iteration: 128 moves per 8 MFMAs)
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0.0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char **argv) {
    const int nit = argc > 1 ? atoi(argv[1]) : 20000, wgs_per_cu = argc > 2 ? atoi(argv[2]) : 2;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int nblk = p.multiProcessorCount * wgs_per_cu;
    double *out; long long *clk;
    hipMalloc(&out, sizeof(double) * nblk * 256); hipMalloc(&clk, sizeof(long long) * 2 * nblk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, nit, out, clk); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double flops = 2048.0 * 8 * nit * 4.0 * nblk;      // 4 waves per workgroup
        printf("CUs %d, %d workgroups of 4 waves, %d x 8 MFMAs per wave: %.3f ms = %.1f TFLOP/s; block 0: %lld shader cycles in %lld ticks of 100 MHz = %.0f MHz; %.1f cycles per MFMA per wave\n",
               p.multiProcessorCount, nblk, nit, ms, flops / ms / 1e9, h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / (8.0 * nit));
    }
    return 0;
}
