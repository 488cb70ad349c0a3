// Issue cost of fp64 vector instructions on gfx950, one wave per SIMD: independent chains (throughput) and one dependent
// chain (latency), v_add_f64 / v_mul_f64 / v_fma_f64 / v_cvt_f64_f32 / v_add_f32, cycles per instruction (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP, int CHAINS>
__global__ void k(double *out, long long *cyc, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)a[i];
    const double m = 1.0000001;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < 100; r++) {
#pragma unroll
        for (int u = 0; u < REP; u++) {
            const int c = u % CHAINS;
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(m));
            if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
            if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[7]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP, int CHAINS>
void run(const char *name, int waves_per_block) {
    double *out; long long *cyc, h;
    hipMalloc(&out, 8 * 1024 * 64); hipMalloc(&cyc, 8);
    k<OP, CHAINS><<<1, 64 * waves_per_block>>>(out, cyc, 1.0);
    k<OP, CHAINS><<<1, 64 * waves_per_block>>>(out, cyc, 1.0);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s chains %d waves/CU %d : %.2f cycles per instruction (memtime ticks)\n", name, CHAINS, waves_per_block, (double)h / (100.0 * REP));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<4, 8>("v_add_f32", 1); run<4, 1>("v_add_f32", 1);
    run<0, 8>("v_add_f64", 1); run<0, 1>("v_add_f64", 1); run<0, 8>("v_add_f64", 8);
    run<1, 8>("v_mul_f64", 1); run<1, 1>("v_mul_f64", 1);
    run<2, 8>("v_fma_f64", 1); run<2, 1>("v_fma_f64", 1);
    run<3, 8>("v_cvt_f64_f32", 1);
    return 0;
}
