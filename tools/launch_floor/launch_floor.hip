// What a single-QP solve cannot go below on this platform: one kernel launch that raises a host-mapped completion word,
// the host spinning on it (the protocol of rsqp_solve for LDS-scale handles).  hipcc --offload-arch=gfx950 -O3 launch_floor.hip
//   empty      : the kernel only writes the word
//   mapped_in  : + 5 dependent-free loads from host-mapped memory (the vectors of the QP) and a 200-B result written back
//   chain<k>   : + a loop of k dependent f64 FMAs by one wave (a loop trip -- compare, branch, FMA -- measured 15 ns: not a statement about clocks)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void k_empty(volatile int *flag, int v) {
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void k_mapped(const double *in, double *out, volatile int *flag, int v, int chain) {
    double a = in[threadIdx.x] + in[64 + threadIdx.x] + in[128 + threadIdx.x] + in[192 + threadIdx.x] + in[256 + threadIdx.x];
    for (int i = 0; i < chain; i++) a = fma(a, 1.0000001, 1e-9);
    if (threadIdx.x < 25) out[threadIdx.x] = a;
    __threadfence_system();
    if (threadIdx.x == 0) *flag = v;
}

int main() {
    void *h = nullptr;
    hipHostMalloc(&h, 1 << 16, hipHostMallocMapped);
    std::memset(h, 0, 1 << 16);
    void *d = nullptr;
    hipHostGetDevicePointer(&d, h, 0);
    volatile int *hflag = reinterpret_cast<volatile int *>(static_cast<char *>(h) + 8192);
    int *dflag = reinterpret_cast<int *>(static_cast<char *>(d) + 8192);
    double *din = static_cast<double *>(d), *dout = static_cast<double *>(d) + 512;
    const int reps = 2000;
    for (int variant = 0; variant < 5; variant++) {
        const int chain = variant <= 1 ? 0 : (variant == 2 ? 1000 : (variant == 3 ? 4000 : 16000));
        int v = 0;
        double best = 1e30, sum = 0.0;
        for (int r = 0; r < reps + 100; r++) {
            ++v;
            const auto t0 = std::chrono::steady_clock::now();
            if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, nullptr, dflag, v);
            else hipLaunchKernelGGL(k_mapped, dim3(1), dim3(64), 0, nullptr, din, dout, dflag, v, chain);
            while (*hflag != v) { }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (r >= 100) { sum += us; if (us < best) best = us; }
        }
        std::printf("variant %d (chain %5d): mean %.2f us, min %.2f us per launch + spin\n", variant, chain, sum / reps, best);
    }
    return 0;
}
