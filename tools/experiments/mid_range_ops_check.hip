// mid_range_ops_check.hip — the division / square-root sequences of ctx.h (div_mid, inv_mid, MidDivisor, div2_mid, div_rn with rcp_nr,
// sqrt_ge1, div2_mid_f32, sqrtf_mid) against the compiler's own a / b, 1 / b, sqrt(x), sqrtf(x): bit for bit, on the GPU, over random and
// adversarial operands in the range the kernels use them in.  Not part of the product or the test-suite; run once per
// change of those helpers, result in profiles/r06_mid_range_ops_check.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Irebvo_amd/csrc -Irebvo_amd/host/include \
//         tools/experiments/mid_range_ops_check.hip -o /tmp/mid_range_ops_check && /tmp/mid_range_ops_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "ctx.h"

using namespace edgehip;

__device__ __forceinline__ uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a double with a random sign, an exponent in [-E, E] and one of four kinds of significand: random, just above 1, just below 2, short
__device__ double rnd_f64(uint64_t &s, int E, bool positive) {
    const uint64_t r = splitmix(s), k = splitmix(s);
    const int e = (int)(k % (uint64_t)(2 * E + 1)) - E;
    uint64_t m = r & 0xFFFFFFFFFFFFFull;
    switch ((k >> 40) & 7) {
        case 0: m &= 0xFFull; break;                        // 1 + a few ulps
        case 1: m |= 0xFFFFFFFFFFF00ull; break;             // 2 - a few ulps
        case 2: m &= 0xFFFF000000000ull; break;             // a short significand
        case 3: m = 0; break;                               // a power of two
        default: break;
    }
    const uint64_t bits = ((positive ? 0ull : (k >> 63)) << 63) | ((uint64_t)(e + 1023) << 52) | m;
    return __longlong_as_double((long long)bits);
}
__device__ float rnd_f32(uint64_t &s, int E, bool positive) {
    const uint64_t r = splitmix(s), k = splitmix(s);
    const int e = (int)(k % (uint64_t)(2 * E + 1)) - E;
    uint32_t m = (uint32_t)r & 0x7FFFFFu;
    switch ((k >> 40) & 7) {
        case 0: m &= 0xFu; break;
        case 1: m |= 0x7FFFF0u; break;
        case 2: m &= 0x7F0000u; break;
        case 3: m = 0; break;
        default: break;
    }
    const uint32_t bits = ((positive ? 0u : (uint32_t)(k >> 63)) << 31) | ((uint32_t)(e + 127) << 23) | m;
    return __uint_as_float(bits);
}
__device__ __forceinline__ bool same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b); }
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

enum { DIV_MID, INV_MID, MIDDIVISOR, DIV2_MID, DIV_RN, SQRT_GE1, DIV2_F32, SQRTF_MID, SPECIALS, NCHK };

__global__ void k_check(uint64_t seed, int iters, int E64, int E32, unsigned long long *bad, double *first_bad) {
    uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0xD1342543DE82EF95ull);
    unsigned long long nb[NCHK] = {0};
    auto report = [&](int which, double x, double y) {
        if (nb[which]++ == 0 && atomicAdd(&bad[NCHK + which], 1ull) == 0) { first_bad[2 * which] = x; first_bad[2 * which + 1] = y; }
    };
    for (int it = 0; it < iters; it++) {
        const double a = rnd_f64(s, E64, false), b = rnd_f64(s, E64, false), c = rnd_f64(s, E64, false);
        if (!same(div_mid(a, b), a / b)) report(DIV_MID, a, b);
        if (!same(inv_mid(b), 1.0 / b)) report(INV_MID, b, 0);
        { const MidDivisor d(b); if (!same(d(a), a / b) || !same(d(c), c / b)) report(MIDDIVISOR, a, b); }
        { double q0, q1; div2_mid(a, c, b, q0, q1); if (!same(q0, a / b) || !same(q1, c / b)) report(DIV2_MID, a, b); }
        if (!same(div_rn(a, b, rcp_for_div_rn(b)), a / b)) report(DIV_RN, a, b);
        { const double x = rnd_f64(s, E64, true); const double x1 = x < 1.0 ? 1.0 / x : x; if (!same(sqrt_ge1(x1), sqrt(x1))) report(SQRT_GE1, x1, 0); }
        { const float n0 = rnd_f32(s, E32, false), n1 = rnd_f32(s, E32, false), d = rnd_f32(s, E32, false); float q0, q1; div2_mid_f32(n0, n1, d, q0, q1);
          if (!same(q0, n0 / d) || !same(q1, n1 / d)) report(DIV2_F32, (double)n0, (double)d); }
        { const float x = rnd_f32(s, E32, true); if (!same(sqrtf_mid(x), sqrtf(x))) report(SQRTF_MID, (double)x, 0); }
    }
    // zeros, infinities, NaNs: what v_div_fixup (and the square roots' own arithmetic) must hand through
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double inf = __longlong_as_double(0x7FF0000000000000ll), nan = __longlong_as_double(0x7FF8000000000000ll);
        const double v[8] = {0.0, -0.0, 1.0, -3.5, inf, -inf, nan, 7.25};
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) {
                if (!same(div_mid(v[i], v[j]), v[i] / v[j])) report(SPECIALS, v[i], v[j]);
                if (!same(inv_mid(v[j]), 1.0 / v[j])) report(SPECIALS, 1.0, v[j]);
                double q0, q1; div2_mid(v[i], v[(i + 3) & 7], v[j], q0, q1);
                if (!same(q0, v[i] / v[j]) || !same(q1, v[(i + 3) & 7] / v[j])) report(SPECIALS, v[i], v[j]);
                float f0, f1; div2_mid_f32((float)v[i], (float)v[(i + 3) & 7], (float)v[j], f0, f1);
                if (!same(f0, (float)v[i] / (float)v[j]) || !same(f1, (float)v[(i + 3) & 7] / (float)v[j])) report(SPECIALS, v[i], v[j]);
            }
        if (!same(sqrt_ge1(inf), sqrt(inf)) || !same(sqrt_ge1(nan), sqrt(nan)) || !same(sqrt_ge1(1.0), 1.0)) report(SPECIALS, inf, 0);
        const float finf = __uint_as_float(0x7F800000u);
        if (!same(sqrtf_mid(0.f), 0.f) || !same(sqrtf_mid(finf), finf) || !same(sqrtf_mid(1.f), 1.f)) report(SPECIALS, 0, 0);
    }
    for (int i = 0; i < NCHK; i++)
        if (nb[i]) atomicAdd(&bad[i], nb[i]);
}

int main(int argc, char **argv) {
    const int blocks = 256 * 8, threads = 256, iters = argc > 1 ? atoi(argv[1]) : 8192;
    unsigned long long *bad; double *fb;
    (void)hipMalloc(&bad, 2 * NCHK * sizeof(unsigned long long)); (void)hipMalloc(&fb, 2 * NCHK * sizeof(double));
    const char *names[NCHK] = {"div_mid(a, b) vs a / b", "inv_mid(b) vs 1 / b", "MidDivisor(b)(a) vs a / b", "div2_mid vs a / b, c / b", "div_rn(a, b, rcp_nr(b)) vs a / b",
                               "sqrt_ge1(x) vs sqrt(x), x >= 1", "div2_mid_f32 vs n0 / d, n1 / d", "sqrtf_mid(x) vs sqrtf(x)", "zeros / infinities / NaNs through all of them"};
    // exponent ranges: what the kernels feed them is within 2^+-40 (doubles) / 2^+-20 (floats); checked far beyond that, and once at the range's ends
    const int ranges[3][2] = {{60, 30}, {300, 60}, {1000, 120}};
    for (int r = 0; r < 3; r++) {
        (void)hipMemset(bad, 0, 2 * NCHK * sizeof(unsigned long long)); (void)hipMemset(fb, 0, 2 * NCHK * sizeof(double));
        hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + r, iters, ranges[r][0], ranges[r][1], bad, fb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        unsigned long long h[2 * NCHK]; double hf[2 * NCHK];
        (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(hf, fb, sizeof(hf), hipMemcpyDeviceToHost);
        const double n = (double)blocks * threads * iters;
        printf("exponents within 2^+-%d (double) / 2^+-%d (float), %.3g operand sets per check:\n", ranges[r][0], ranges[r][1], n);
        for (int i = 0; i < NCHK; i++) {
            printf("  %-48s %llu differing", names[i], h[i]);
            if (h[i]) printf("   (first: %.17g, %.17g)", hf[2 * i], hf[2 * i + 1]);
            printf("\n");
        }
    }
    return 0;
}
