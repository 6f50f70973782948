// Issue rates of the vector instructions the traversal and gather kernels are made of (gfx950): cycles per wave64
// instruction with W waves per SIMD, measured with s_memtime around long unrolled runs of independent instructions.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    f2 p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed}, p4 = {seed + 4, seed}, p5 = {seed + 5, seed}, p6 = {seed + 6, seed}, p7 = {seed + 7, seed};
    const float k = seed * 0.5f;
    const double kd = seed * 0.5;
    const f2 kp = {k, k};
    int sl = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kp));) }
        if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 3) { REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 4) { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 5) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kp));) }
        if (OP == 6) { REP16(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 7) { REP16(asm volatile("v_max_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_min_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 8) { REP16(asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 5\n v_readlane_b32 %8, %2, 7\n v_readlane_b32 %8, %3, 9\n v_readlane_b32 %8, %4, 11\n v_readlane_b32 %8, %5, 13\n v_readlane_b32 %8, %6, 15\n v_readlane_b32 %8, %7, 17" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(sl));) }
        if (OP == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kp));) }
        if (OP == 10) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cmp_lt_f64 vcc, %1, %8\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n v_cmp_lt_f64 vcc, %4, %8\n v_cmp_lt_f64 vcc, %5, %8\n v_cmp_lt_f64 vcc, %6, %8\n v_cmp_lt_f64 vcc, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd) : "vcc");) }
        if (OP == 11) { REP16(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 12) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");) }
        if (OP == 13) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s10","s11");) }
        if (OP == 14) { REP16(asm volatile("v_max3_f32 %0, %0, %8, %8\n v_max3_f32 %1, %1, %8, %8\n v_max3_f32 %2, %2, %8, %8\n v_max3_f32 %3, %3, %8, %8\n v_max3_f32 %4, %4, %8, %8\n v_max3_f32 %5, %5, %8, %8\n v_max3_f32 %6, %6, %8, %8\n v_max3_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 15) { REP16(asm volatile("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %1, %1, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 16) { REP16(asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 17) { REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 18) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 8, %8\n v_lshl_add_u32 %1, %1, 8, %8\n v_lshl_add_u32 %2, %2, 8, %8\n v_lshl_add_u32 %3, %3, 8, %8\n v_lshl_add_u32 %4, %4, 8, %8\n v_lshl_add_u32 %5, %5, 8, %8\n v_lshl_add_u32 %6, %6, 8, %8\n v_lshl_add_u32 %7, %7, 8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 19) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");) }
        if (OP == 20) { REP16(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %8\n v_cmp_lt_f32_e64 s[10:11], %1, %8\n v_cmp_lt_f32_e64 s[10:11], %2, %8\n v_cmp_lt_f32_e64 s[10:11], %3, %8\n v_cmp_lt_f32_e64 s[10:11], %4, %8\n v_cmp_lt_f32_e64 s[10:11], %5, %8\n v_cmp_lt_f32_e64 s[10:11], %6, %8\n v_cmp_lt_f32_e64 s[10:11], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s10","s11");) }
        if (OP == 21) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 22) { REP16(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));) }
        if (OP == 23) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 24) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 25) { REP16(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(k));) }
        if (OP == 26) { REP16(asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %8\n v_cvt_f32_f64 %2, %8\n v_cvt_f32_f64 %3, %8\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %8\n v_cvt_f32_f64 %6, %8\n v_cvt_f32_f64 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(kd));) }
        if (OP == 27) { REP16(asm volatile("v_sub_f32 %0, s10, %0\n v_sub_f32 %1, s10, %1\n v_sub_f32 %2, s10, %2\n v_sub_f32 %3, s10, %3\n v_sub_f32 %4, s10, %4\n v_sub_f32 %5, s10, %5\n v_sub_f32 %6, s10, %6\n v_sub_f32 %7, s10, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s10");) }
        if (OP == 28) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 29) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");) }
        if (OP == 30) { REP16(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %4, %8\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %5, %8\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %6, %8\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], %7, %8\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s10","s11");) }
        if (OP == 31) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 32) { REP16(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 33) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 34) { REP16(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 35) { REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
        if (OP == 36) { REP16(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + (float)sl;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, float* d_out, unsigned long long* d_cyc)
{
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;     // one block per CU: 4 SIMDs x W waves
        const int blocks = 256, iters = 200;
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters, 1.5f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters, 1.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(blocks * threads / 64);
        hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : c) sum += (double)v;
        const double per_wave = sum / c.size();
        const double n_inst = (double)iters * 16 * 8;
        // s_memtime ticks at a constant 100 MHz-derived rate on some parts and at shader clock on others: report raw ticks too
        printf("%-14s W=%d  ticks/inst per wave %.3f   per SIMD (W waves share it) %.3f\n", name, waves_per_simd, per_wave / n_inst, per_wave / n_inst / waves_per_simd);
    }
}

int main()
{
    float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, 256 * 1024 * 4);
    hipMalloc(&d_cyc, 256 * 16 * 8);
    run<0>("v_add_f32", d_out, d_cyc);
    run<4>("v_mul_f32", d_out, d_cyc);
    run<1>("v_pk_add_f32", d_out, d_cyc);
    run<5>("v_pk_mul_f32", d_out, d_cyc);
    run<6>("v_minmax_f32", d_out, d_cyc);
    run<9>("v_pk_fma_f32", d_out, d_cyc);
    run<2>("v_add_f64", d_out, d_cyc);
    run<3>("v_mul_f64", d_out, d_cyc);
    run<11>("v_fma_f64", d_out, d_cyc);
    run<7>("v_minmax_f64", d_out, d_cyc);
    run<10>("v_cmp_lt_f64", d_out, d_cyc);
    run<12>("v_cndmask_b32", d_out, d_cyc);
    run<8>("v_readlane_b32", d_out, d_cyc);
    run<13>("v_cndmask_e64", d_out, d_cyc);
    run<28>("v_cndmask_vcc", d_out, d_cyc);
    run<14>("v_max3_f32", d_out, d_cyc);
    run<15>("v_med3_f32", d_out, d_cyc);
    run<16>("v_cvt_i32_f32", d_out, d_cyc);
    run<17>("v_mov_b32", d_out, d_cyc);
    run<18>("v_lshl_add_u32", d_out, d_cyc);
    run<19>("v_cmp_f32_vcc", d_out, d_cyc);
    run<20>("v_cmp_f32_sgpr", d_out, d_cyc);
    run<21>("v_rcp_f64", d_out, d_cyc);
    run<22>("v_sqrt_f64", d_out, d_cyc);
    run<23>("v_rcp_f32", d_out, d_cyc);
    run<24>("v_exp_f32", d_out, d_cyc);
    run<25>("v_cvt_f64_f32", d_out, d_cyc);
    run<26>("v_cvt_f32_f64", d_out, d_cyc);
    run<27>("v_sub_f32_sgpr", d_out, d_cyc);
    run<29>("cmp+cndmask vcc (2 inst)", d_out, d_cyc);
    run<30>("cmp+cndmask sgpr (2 inst)", d_out, d_cyc);
    run<31>("v_fma_f32", d_out, d_cyc);
    run<32>("v_and_b32", d_out, d_cyc);
    run<33>("v_add_u32", d_out, d_cyc);
    run<34>("v_max_f32", d_out, d_cyc);
    run<35>("v_sub_f32", d_out, d_cyc);
    run<36>("v_mul_f32 swapped", d_out, d_cyc);
    return 0;
}
