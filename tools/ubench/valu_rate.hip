// Issue rate of a few VALU instructions with one wave per SIMD (256 threads per workgroup, one
// workgroup per CU): time per wave-instruction over a long run of independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k64(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1 + blockIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 1) { REP8(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 2) { REP8(asm volatile("v_max_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_min_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 3) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 4) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 5) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cmp_lt_f64 vcc, %1, %8\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n v_cmp_lt_f64 vcc, %4, %8\n v_cmp_lt_f64 vcc, %5, %8\n v_cmp_lt_f64 vcc, %6, %8\n v_cmp_lt_f64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (KIND == 6) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %8\n v_cmp_lt_u64 vcc, %1, %8\n v_cmp_lt_u64 vcc, %2, %8\n v_cmp_lt_u64 vcc, %3, %8\n v_cmp_lt_u64 vcc, %4, %8\n v_cmp_lt_u64 vcc, %5, %8\n v_cmp_lt_u64 vcc, %6, %8\n v_cmp_lt_u64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (KIND == 7) { REP8(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 8) { REP8(asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
__global__ __launch_bounds__(256) void k32(double* out, long long* cyc, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = 1 + blockIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 1) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (KIND == 2) { REP8(asm volatile("v_max_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 3) { REP8(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 4) { REP8(asm volatile("v_pk_max_f16 %0, %0, %8\n v_pk_min_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_min_f16 %3, %3, %8\n v_pk_max_f16 %4, %4, %8\n v_pk_min_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_min_f16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 6) { REP8(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n v_accvgpr_write_b32 a4, %4\n v_accvgpr_write_b32 a5, %5\n v_accvgpr_write_b32 a6, %6\n v_accvgpr_write_b32 a7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (KIND == 7) { REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, double* out, long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double n = 64.0 * iters;
    printf("%-16s %8.2f counter ticks / instr   kernel %.3f ms -> %.2f ns / instr\n", name, h[0] / n, ms, ms * 1e6 / n);
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(double));
    hipMalloc(&cyc, 256 * sizeof(long long));
    run("v_fma_f64", k64<0>, out, cyc);
    run("v_min_f64", k64<1>, out, cyc);
    run("v_minmax_f64", k64<2>, out, cyc);
    run("v_add_f64", k64<3>, out, cyc);
    run("v_mul_f64", k64<4>, out, cyc);
    run("v_cmp_lt_f64", k64<5>, out, cyc);
    run("v_cmp_lt_u64", k64<6>, out, cyc);
    run("v_mov_b64", k64<7>, out, cyc);
    run("v_lshlrev_b64", k64<8>, out, cyc);
    run("v_xor_b32", k32<0>, out, cyc);
    run("v_cndmask_b32", k32<1>, out, cyc);
    run("v_minmax_u32", k32<2>, out, cyc);
    run("v_minmax_f32", k32<3>, out, cyc);
    run("v_pk_minmax_f16", k32<4>, out, cyc);
    run("v_fma_f32", k32<5>, out, cyc);
    run("v_accvgpr_write", k32<6>, out, cyc);
    run("v_mov_dpp", k32<7>, out, cyc);
    return 0;
}
