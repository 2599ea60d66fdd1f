// How many scalar-unit instructions per cycle can one CU of an MI355X retire?  (the peak bench.py's `fractions.scalar` divides by)
// Every wave runs a loop of independent s_add_u32 on four SGPRs; 1, 2, 4, 8 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o scalar_peak tools/microbench/scalar_peak.hip && ./scalar_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_salu(int iters, unsigned long long *clk, int *sink) {
    unsigned a = blockIdx.x, b = blockIdx.x + 1, c = blockIdx.x + 2, d = blockIdx.x + 3;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 5\n\ts_add_u32 %3, %3, 7" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
    if (a + b + c + d == 0x12345678u) sink[0] = 1;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    volatile int itersV = 20000;
    const int iters = itersV;
    int wallKHz = 0;
    hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
    unsigned long long *clk; int *sink;
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 64);
    hipMalloc(&sink, 4);
    printf("%s: %d CUs, wall clock %d kHz\n", p.name, cus, wallKHz);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;          // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(256), 0, 0, 10, clk, sink);   // warm-up
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(256), 0, 0, iters, clk, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e)); }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
        cyc /= blocks; wall /= blocks;
        const double insts = (double)iters * 64.0 * 4.0 * wps;             // scalar instructions per CU (4 waves per block, 64 per iteration)
        const double mhz = cyc / (wall / (wallKHz * 1e3)) / 1e6;           // clock64 ticks per second while the waves ran
        printf("%d wave(s)/SIMD: kernel %.3f ms, shader clock %.0f MHz: %.3f scalar instructions per CU per ns = %.2f per cycle\n",
               wps, ms, mhz, insts / (ms * 1e6), insts / (ms * 1e-3 * mhz * 1e6));
    }
    return 0;
}
