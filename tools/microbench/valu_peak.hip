// How many wave64 VALU instructions per cycle does one SIMD of an MI355X issue?  (the peak bench.py's `fractions.valu` divides by:
// a wave64 fp32 op over 2 cycles on a SIMD-32, MI355X_MICROARCH.md:54,473)  Independent v_add_f32 / v_mul_f32 in every wave.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/microbench/valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_valu(int iters, float *sink) {
    float a = threadIdx.x, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f, e = a + 4.0f, f = a + 5.0f, g = a + 6.0f, h = a + 7.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            asm volatile("v_add_f32 %0, 1.0, %0\n\tv_mul_f32 %1, 1.0, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_mul_f32 %3, 1.0, %3\n\t"
                         "v_add_f32 %4, 1.0, %4\n\tv_mul_f32 %5, 1.0, %5\n\tv_add_f32 %6, 1.0, %6\n\tv_mul_f32 %7, 1.0, %7"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
    if (a + b + c + d + e + f + g + h == 12345.678f) sink[0] = 1.0f;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    volatile int itersV = 20000;
    const int iters = itersV;
    float *sink;
    (void)hipMalloc(&sink, 4);
    printf("%d CUs\n", cus);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;   // 256 threads = one wave per SIMD; wps blocks per CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, 10, sink);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, iters, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 64.0 * wps;   // wave64 VALU instructions per SIMD
        printf("%d wave(s)/SIMD: kernel %.3f ms: %.3f wave64 VALU instructions per SIMD per ns = %.2f per cycle at 2.4 GHz (%.1f T lane-op/s on the chip)\n",
               wps, ms, insts / (ms * 1e6), insts / (ms * 1e-3 * 2.4e9), insts * 64.0 * 4.0 * cus / (ms * 1e-3) / 1e12);
    }
    return 0;
}
