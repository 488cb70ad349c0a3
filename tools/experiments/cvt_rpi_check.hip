// Exhaustive check of v_cvt_rpi_i32_f32 (round to nearest, ties towards +inf) against round() half-away-from-zero
// as the field kernels need it: which float inputs give a different integer?  build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ int ref_round(float v) {
    const float a = fabsf(v);
    float t = floorf(a + 0.5f);
    t = a < 0.5f ? 0.f : t;
    return (int)copysignf(t, v);
}
__device__ __forceinline__ int rpi(float v) {
    int r;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__global__ void k(unsigned long long *cnt, unsigned *ex, float lim) {
    const unsigned long long base = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    for (int j = 0; j < 256; j++) {
        const unsigned bits = (unsigned)(base + j);
        const float v = __uint_as_float(bits);
        if (!(fabsf(v) < lim)) continue;
        const int a = ref_round(v), b = rpi(v);
        // only results that can index a pixel matter: both negative = both rejected
        if (a != b && !(a < 0 && b < 0)) {
            const unsigned long long i = atomicAdd(cnt, 1ull);
            if (i < 16) { ex[2 * i] = bits; ex[2 * i + 1] = (unsigned)b; }
        }
    }
}
int main() {
    unsigned long long *cnt; unsigned *ex;
    hipMalloc(&cnt, 8); hipMalloc(&ex, 128); hipMemset(cnt, 0, 8); hipMemset(ex, 0, 128);
    k<<<(1u << 24) / 256 * 1, 256>>>(cnt, ex, 4096.f);   // 2^24 threads * 256 = 2^32 bit patterns
    unsigned long long h; unsigned he[32];
    hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 128, hipMemcpyDeviceToHost);
    printf("mismatches that matter: %llu\n", h);
    for (int i = 0; i < 16 && i < (int)h; i++) { float v; memcpy(&v, &he[2 * i], 4); printf("  v=%.9g (0x%08x) rpi=%d round=%d\n", v, he[2 * i], (int)he[2 * i + 1], (int)roundf(v)); }
    return 0;
}
