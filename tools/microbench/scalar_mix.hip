// Issue rate of the scalar-side instruction MIX a traversal kernel executes: bench.py's scalar fraction
// summed SALU + branch + scalar-memory instructions against a peak measured with independent s_add_u32 only.  On GCN-lineage
// CUs those are separate issue classes; this measures, per CU and cycle, what each class sustains alone and what mixes of them
// sustain together, with 1 / 2 / 4 / 6 / 8 waves per SIMD on every CU:
//     salu      independent s_add_u32                               (tools/microbench/scalar_peak.hip's loop)
//     br_nt     s_cbranch_scc1 that is never taken
//     br_t      s_branch to the next instruction (always taken)
//     smem      s_load_dwordx8 / x4 from one cached line, eight in flight between waits
//     mix_pk    per 25 instructions 20 salu + 2 br_nt + 2 br_t + 1 smem  (k_packet's 80 % / 16 % / 4 %, profiles/pmc_traversal.json)
//     mix_ln    per 25 instructions 20 salu + 5 branches                 (k_intersect<*,0> on C3: 80 % / 20 % / 0)
// If the classes shared one issue port, a mix would retire  1 / (f_s/p_s + f_b/p_b + f_m/p_m)  instructions per cycle (additive
// model); if they issue side by side it retires more, up to  min_i p_i / f_i.  bench.py reads the table this prints
// (profiles/r03/scalar_mix_peak.txt) and divides a launch's counters by the model the measurement supports.
//   hipcc --offload-arch=gfx950 -O3 -o scalar_mix tools/microbench/scalar_mix.hip && ./scalar_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));

#define SALU4 "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 5\n\ts_add_u32 %3, %3, 7\n\t"
#define BR_NT "s_cbranch_scc1 1f\n\t"          /* scc is 0 here: falls through; label 1 is the end of the block */
#define BR_T(n) "s_branch " #n "f\n\t" #n ":\n\t"

// kind 0 salu, 1 br_nt, 2 br_t, 3 smem, 4 mix_pk, 5 mix_ln; every block is 25 scalar-unit instructions (s_waitcnt / s_cmp excluded
// from the count where present: one s_cmp per block keeps scc = 0, it is counted as the block's first SALU instruction)
template <int KIND>
__global__ void k_mix(int iters, const int *__restrict__ line, unsigned long long *clk, int *sink) {
    unsigned a = blockIdx.x, b = blockIdx.x + 1, c = blockIdx.x + 2, d = blockIdx.x + 3;
    v8i q = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (KIND == 0)
                asm volatile(SALU4 SALU4 SALU4 SALU4 SALU4 SALU4 "s_add_u32 %0, %0, 1\n\t" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
            else if (KIND == 1)
                asm volatile("s_cmp_lg_u32 0, 0\n\t" BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT BR_NT
                             BR_NT BR_NT "1:\n\t" : "+s"(a) : : "scc");
            else if (KIND == 2)
                asm volatile(BR_T(10) BR_T(11) BR_T(12) BR_T(13) BR_T(14) BR_T(15) BR_T(16) BR_T(17) BR_T(18) BR_T(19) BR_T(20) BR_T(21) BR_T(22) BR_T(23) BR_T(24)
                             BR_T(25) BR_T(26) BR_T(27) BR_T(28) BR_T(29) BR_T(30) BR_T(31) BR_T(32) BR_T(33) BR_T(34) : "+s"(a) : :);
            else if (KIND == 3)
                asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_load_dwordx8 %0, %1, 0x20\n\t"
                             "s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=&s"(q) : "s"(line) : "memory");
            else if (KIND == 4)   // the load is requested first and waited for last, as k_packet's two register sets do
                asm volatile("s_load_dwordx8 %4, %5, 0x0\n\ts_cmp_lg_u32 0, 0\n\t" SALU4 BR_NT SALU4 BR_T(40) SALU4 "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\t"
                             BR_NT SALU4 BR_T(41) "1:\n\ts_waitcnt lgkmcnt(0)\n\t" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "=&s"(q) : "s"(line) : "scc", "memory");
            else
                asm volatile("s_cmp_lg_u32 0, 0\n\t" SALU4 BR_NT SALU4 BR_T(50) SALU4 BR_NT "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\t" BR_T(51) SALU4 BR_NT
                             "1:\n\t" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
    if (a + b + c + d + (unsigned)q[0] == 0x12345678u) sink[0] = 1;
}

template <int KIND>
static double run(const char *name, int cus, int wps, int iters, int wallKHz, const int *line, unsigned long long *clk, int *sink) {
    const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix<KIND>, dim3(blocks), dim3(256), 0, 0, 10, line, clk, sink);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<KIND>, dim3(blocks), dim3(256), 0, 0, iters, line, clk, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e)); }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    cyc /= blocks; wall /= blocks;
    const double insts = (double)iters * 100.0 * 4.0 * wps;   // scalar-unit instructions per CU: 4 blocks of 25 per iteration, 4 waves per workgroup
    const double mhz = cyc / (wall / (wallKHz * 1e3)) / 1e6;
    const double perCycle = insts / (ms * 1e-3 * mhz * 1e6);
    printf("%-7s %d wave(s)/SIMD: kernel %8.3f ms, shader clock %4.0f MHz: %.3f instructions per CU per cycle\n", name, wps, ms, mhz, perCycle);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return perCycle;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    volatile int itersV = 4000;
    const int iters = itersV;
    int wallKHz = 0;
    hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
    unsigned long long *clk; int *sink, *line;
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus * 64);
    hipMalloc(&sink, 4);
    hipMalloc(&line, 256);
    hipMemset(line, 0, 256);
    printf("%s: %d CUs, wall clock %d kHz; every block of the loops is 25 scalar-unit instructions\n", p.name, cus, wallKHz);
    const int W[5] = {1, 2, 4, 6, 8};
    double r[6][5];
    for (int wi = 0; wi < 5; wi++) {
        const int w = W[wi];
        r[0][wi] = run<0>("salu", cus, w, iters, wallKHz, line, clk, sink);
        r[1][wi] = run<1>("br_nt", cus, w, iters, wallKHz, line, clk, sink);
        r[2][wi] = run<2>("br_t", cus, w, iters, wallKHz, line, clk, sink);
        r[3][wi] = run<3>("smem", cus, w, iters, wallKHz, line, clk, sink);
        r[4][wi] = run<4>("mix_pk", cus, w, iters, wallKHz, line, clk, sink);
        r[5][wi] = run<5>("mix_ln", cus, w, iters, wallKHz, line, clk, sink);
    }
    printf("\nmodel check (instructions per CU per cycle):\n");
    for (int wi = 0; wi < 5; wi++) {
        const double ps = r[0][wi], pb = 0.5 * (r[1][wi] + r[2][wi]) > 0 ? 2.0 / (1.0 / r[1][wi] + 1.0 / r[2][wi]) : 0, pm = r[3][wi];
        // mix_pk: 20 salu (the s_cmp included), 2 + 2 branches, 1 smem; mix_ln: 20 salu, 3 + 2 branches
        const double addPk = 25.0 / (20.0 / ps + 2.0 / r[1][wi] + 2.0 / r[2][wi] + 1.0 / pm), addLn = 25.0 / (20.0 / ps + 3.0 / r[1][wi] + 2.0 / r[2][wi]);
        const double sidePk = 25.0 / (20.0 / ps), sideLn = 25.0 / (20.0 / ps);
        printf("%d wave(s)/SIMD: salu %.3f branch(harmonic) %.3f smem %.3f | mix_pk measured %.3f, one shared port would give %.3f, side-by-side issue %.3f | mix_ln measured %.3f, shared %.3f, side by side %.3f\n",
               W[wi], ps, pb, pm, r[4][wi], addPk, sidePk, r[5][wi], addLn, sideLn);
    }
    return 0;
}
