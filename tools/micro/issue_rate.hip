// Micro-benchmark (diagnostic, not part of the library): issue rate of vector and scalar instructions on one
// gfx950 SIMD as a function of resident wavefronts.  Build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 32 independent v_fma_f32
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 1) {  // 32 s_add_u32 (4 chains)
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        } else if (MODE == 2) {  // 16 v_fma + 16 s_add interleaved
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %5, %5, 1\n"
                             "v_fma_f32 %2, %2, %2, %2\n s_add_u32 %6, %6, 1\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %7, %7, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        } else if (MODE == 3) {  // 32 independent v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 4) {  // 32 v_cndmask (VCC read)
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");
        } else if (MODE == 6) {  // 32 v_cndmask on 8 independent registers
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.5f) : "vcc");
        } else if (MODE == 7) {  // 32 v_fma_f32 in 4 chains (dependence distance 4)
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (MODE == 8) {  // 32 v_fma_f32 in ONE chain (fully dependent)
#pragma unroll
            for (int r = 0; r < 32; r++) asm volatile("v_fma_f32 %0, %0, %0, %0\n" : "+v"(a0));
        } else if (MODE == 9) {  // 32 v_max3_f32, 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                             "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 10) {  // 32 v_cndmask_b32_e64 with an SGPR-pair mask, 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                             "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.5f) : "s20", "s21");
        } else if (MODE == 11) {  // v_mul_f32_e32 (VOP2), 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                             "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 12) {  // v_min_f32_e32
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                             "v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 13) {  // v_mov_b32
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 14) {  // v_cmp_gt_f32_e32 (writes vcc) x32
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n"
                             ::"v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if (MODE == 15) {  // (v_cmp_gt_f32 vcc ; v_cndmask_b32 e32) x16
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %0, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %5, %5, %1, vcc\n"
                             "v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %6, %6, %2, vcc\n v_cmp_gt_f32 vcc, %3, %0\n v_cndmask_b32 %7, %7, %3, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"vcc");
        } else if (MODE == 16) {  // s_and_saveexec_b64 + s_or_b64 exec (mask juggling) x16 pairs
#pragma unroll
            for (int r = 0; r < 16; r++)
                asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n" ::: "s20", "s21", "scc");
        } else if (MODE == 17) {  // not-taken s_cbranch_execz + s_nop... x32
#pragma unroll
            for (int r = 0; r < 32; r++) asm volatile("s_cbranch_execz 0\n" :::);
        } else if (MODE == 18) {  // v_cmp_gt_f32_e64 to an SGPR pair x32
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %1, %2\n v_cmp_gt_f32 s[24:25], %2, %3\n v_cmp_gt_f32 s[26:27], %3, %0\n"
                             ::"v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 19) {  // 32 v_fma_mix_f32 (fp16 operand converted in the instruction), 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_fma_mix_f32 %0, %8, %0, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %8, %2, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %3, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %4, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %8, %6, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %7, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x3c003c00));
        } else if (MODE == 20) {  // 32 v_add_u32, 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(3));
        } else if (MODE == 21) {  // 32 v_and_b32 / v_lshlrev_b32 alternating, 8 independent
#pragma unroll
            for (int r = 0; r < 4; r++)
                asm volatile("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_and_b32 %2, %2, %8\n v_lshlrev_b32 %3, 1, %3\n"
                             "v_and_b32 %4, %4, %8\n v_lshlrev_b32 %5, 1, %5\n v_and_b32 %6, %6, %8\n v_lshlrev_b32 %7, 1, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x7fffffff));
        } else if (MODE == 5) {  // 16 x (v_cmp + s_and_saveexec-like scalar mask op)
#pragma unroll
            for (int r = 0; r < 16; r++)
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n" ::"v"(a0), "v"(a1) : "vcc", "scc");
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
}

template <int MODE>
int run(const char* name, int instrPerIter) {
    float* out; long long* cyc; long long h;
    CHECK(hipMalloc(&out, 256 * 2048 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int threads = wavesPerSimd * 4 * 64;  // one block per CU
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, 10, cyc);  // warm
        const int blocks = threads > 1024 ? 256 * (threads / 1024) : 256;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, iters, cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        // clock64 = s_memtime ticks (100 MHz realtime or shader clock depending on build): report wall-derived cycles at 2.4 GHz too
        const double instrPerSimd = (double)iters * instrPerIter * wavesPerSimd;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per instr per SIMD (= %.2f cycles at 2.4 GHz), clock64 delta %lld\n", name,
               wavesPerSimd, ms, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4, h);
    }
    hipFree(out); hipFree(cyc);
    return 0;
}
#include <cstdlib>
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int which = argc > 1 ? atoi(argv[1]) : 0;
    if (which == 0) {
        if (run<0>("v_fma_f32 x32 (8 chains)", 32)) return 1;
        if (run<7>("v_fma_f32 x32 (4 chains)", 32)) return 1;
        if (run<8>("v_fma_f32 x32 (1 chain)", 32)) return 1;
        if (run<3>("v_pk_fma_f32 x32", 32)) return 1;
        if (run<9>("v_max3_f32 x32", 32)) return 1;
        if (run<6>("v_cndmask vcc x32 (8 indep)", 32)) return 1;
        if (run<10>("v_cndmask sgpr x32 (8 indep)", 32)) return 1;
        if (run<1>("s_add_u32 x32", 32)) return 1;
        if (run<2>("v_fma + s_add x16 each", 32)) return 1;
        if (run<5>("v_cmp + s_and x16 each", 32)) return 1;
    } else if (which == 1) {
        if (run<11>("v_mul_f32_e32 x32", 32)) return 1;
        if (run<12>("v_min_f32_e32 x32", 32)) return 1;
        if (run<13>("v_mov_b32 x32", 32)) return 1;
        if (run<14>("v_cmp_gt_f32 vcc x32", 32)) return 1;
        if (run<15>("v_cmp vcc + v_cndmask x16", 32)) return 1;
        if (run<18>("v_cmp_gt_f32 sgpr x32", 32)) return 1;
        if (run<16>("saveexec + or exec x16", 32)) return 1;
        if (run<17>("s_cbranch_execz (not taken) x32", 32)) return 1;
    } else if (which == 2) {
        if (run<19>("v_fma_mix_f32 x32", 32)) return 1;
        if (run<20>("v_add_u32 x32", 32)) return 1;
        if (run<21>("v_and_b32 / v_lshlrev_b32 x32", 32)) return 1;
    }
    return 0;
}
