// Micro-benchmark of the fused stage-A scan wave's 64-float step AS THE KERNEL HAS IT (four lanes of a quad share a row, exec-masked
// phases, DPP broadcast of the carry; stage_a_fused.hip), one wave per workgroup, in three forms of the sixteen dependent adds of a phase:
//   V0  one asm statement per add           (the hazard recogniser puts an s_nop 0 behind every inline-asm VALU write)
//   V1  one asm statement per phase         (sixteen v_add_f32 back to back)
//   V2  plain C
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/experiments/ubench_scan2.hip -o /tmp/ubench_scan2 && /tmp/ubench_scan2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define EH_BC(val, k) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(val), (k) * 0x55, 0xf, 0xf, false))
template <int V>
__global__ __launch_bounds__(64) void k_scan(float *out, long long *cyc, int w, int reps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, WP = 764, PAD = 4;
    for (int i = lane; i < 16 * WP + 256; i += 64) sm[i] = 1.f + (float)(i & 7) * 0.125f;
    __syncthreads();
    const int n16 = w >> 4, nss = (n16 + 3) >> 2;
    const int srow = lane >> 2, sq = lane & 3;
    float *row = sm + srow * WP + PAD;
    float acc = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        acc = 0.f;
        float4 cur[4], nxt[4];
        auto load = [&](float4 (&v)[4], int ss) __attribute__((always_inline)) {
            const float *p = row + (4 * ss + sq) * 16;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const float4 *>(p + 4 * i);
        };
        auto step = [&](float4 (&v)[4], float4 (&nx)[4], int ss) __attribute__((always_inline)) {
            load(nx, ss + 1);
#define EH_ADDC(DST, PREV) asm volatile("v_add_f32 %0, %1, %0" : "+v"(DST) : "v"(PREV));
#define EH_ADD4(I, CARRY) EH_ADDC(v[I].x, CARRY) EH_ADDC(v[I].y, v[I].x) EH_ADDC(v[I].z, v[I].y) EH_ADDC(v[I].w, v[I].z)
#define EH_CHAIN16_ASM(CARRY)                                                                                                     \
    asm volatile("v_add_f32 %0, %16, %0\n v_add_f32 %1, %0, %1\n v_add_f32 %2, %1, %2\n v_add_f32 %3, %2, %3\n"                   \
                 "v_add_f32 %4, %3, %4\n v_add_f32 %5, %4, %5\n v_add_f32 %6, %5, %6\n v_add_f32 %7, %6, %7\n"                    \
                 "v_add_f32 %8, %7, %8\n v_add_f32 %9, %8, %9\n v_add_f32 %10, %9, %10\n v_add_f32 %11, %10, %11\n"               \
                 "v_add_f32 %12, %11, %12\n v_add_f32 %13, %12, %13\n v_add_f32 %14, %13, %14\n v_add_f32 %15, %14, %15"          \
                 : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z), "+v"(v[1].w), \
                   "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w)  \
                 : "v"(CARRY));
#define EH_PHASE(P)                                                                                          \
    {                                                                                                        \
        const int c = 4 * ss + (P);                                                                          \
        float tot = acc;                                                                                     \
        if (c < n16) {                                                                                       \
            if (sq == (P)) {                                                                                 \
                if (V == 0) { EH_ADD4(0, acc) EH_ADD4(1, v[0].w) EH_ADD4(2, v[1].w) EH_ADD4(3, v[2].w) }     \
                else if (V == 1) { EH_CHAIN16_ASM(acc) }                                                     \
                else {                                                                                       \
                    float a = acc;                                                                           \
                    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                          \
                        v[i].x = a = a + v[i].x; v[i].y = a = a + v[i].y; v[i].z = a = a + v[i].z; v[i].w = a = a + v[i].w; \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
            tot = v[3].w;                                                                                    \
        }                                                                                                    \
        acc = EH_BC(tot, P);                                                                                 \
    }
            EH_PHASE(0) EH_PHASE(1) EH_PHASE(2) EH_PHASE(3)
            const int c = 4 * ss + sq;
            float *q = row + c * 16;
            if (c < n16) {
#pragma unroll
                for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(q + 4 * i) = v[i];
            }
        };
        load(cur, 0);
        int ss = 0;
        for (; ss + 1 < nss; ss += 2) {
            step(cur, nxt, ss);
            step(nxt, cur, ss + 1);
        }
        if (ss < nss) step(cur, nxt, ss);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V>
static void run(const char *name, float *out, long long *cyc) {
    const int reps = 64, w = 752;
    long long h = 0;
    float o[64];
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_scan<V>, dim3(1), dim3(64), 16 * 764 * 4 + 2048, 0, out, cyc, w, reps);
        hipEventRecord(e1, 0);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(o, out, 256, hipMemcpyDeviceToHost);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %9lld ticks of s_memtime = %.2f per element; kernel %.1f us = %.2f ns per element; lane0 total %.4f\n", name, h,
               (double)h / (reps * (double)w), ms * 1e3, ms * 1e6 / (reps * (double)w), o[3]);
    }
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run<0>("V0 one asm per add", out, cyc);
    run<1>("V1 one asm per phase", out, cyc);
    run<2>("V2 plain C", out, cyc);
    return 0;
}
