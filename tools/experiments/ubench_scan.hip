// Micro-benchmark behind the design of the fused stage-A scan wave: cycles of (a) a dependent v_add_f32 chain, (b) ds_read_b128 /
// ds_write_b128 with 16 or 64 active lanes, (c) the 16-float scan step as the kernel has it, one wave per workgroup.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/experiments/ubench_scan.hip -o /tmp/ubench_scan && /tmp/ubench_scan
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_addchain(float *out, long long *cyc, float x) {
    float acc = x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 256; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) acc = acc + x;
        asm volatile("" : "+v"(acc));
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int LANES>
__global__ void k_lds(float *out, long long *cyc) {
    __shared__ float4 s[64 * 8 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 8 + 64; i += 64) s[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    long long t0 = __builtin_readcyclecounter();
    if (lane < LANES) {
#pragma unroll 1
        for (int i = 0; i < 256; i++) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = s[lane * 8 + j + (i & 1)];
#pragma unroll
            for (int j = 0; j < 4; j++) { acc.x += v[j].x; }
#pragma unroll
            for (int j = 0; j < 4; j++) s[lane * 8 + 4 + j] = v[j];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc.x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int LANES>
__global__ void k_scanstep(float *out, long long *cyc) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x, WP = 764;
    for (int i = lane; i < 16 * WP + 64; i += 64) sm[i] = 1.f;
    __syncthreads();
    float acc = 0.f;
    long long t0 = __builtin_readcyclecounter();
    if (lane < LANES) {
        float *row = sm + (lane & 15) * WP + 4;
        float4 cur[4], nxt[4];
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = *reinterpret_cast<float4 *>(row + 4 * i);
#pragma unroll 1
        for (int c = 0; c < 46; c += 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) nxt[i] = *reinterpret_cast<float4 *>(row + (c + 1) * 16 + 4 * i);
#pragma unroll
            for (int i = 0; i < 4; i++) { cur[i].x = acc = acc + cur[i].x; cur[i].y = acc = acc + cur[i].y; cur[i].z = acc = acc + cur[i].z; cur[i].w = acc = acc + cur[i].w; }
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(row + c * 16 + 4 * i) = cur[i];
#pragma unroll
            for (int i = 0; i < 4; i++) cur[i] = *reinterpret_cast<float4 *>(row + (c + 2) * 16 + 4 * i);
#pragma unroll
            for (int i = 0; i < 4; i++) { nxt[i].x = acc = acc + nxt[i].x; nxt[i].y = acc = acc + nxt[i].y; nxt[i].z = acc = acc + nxt[i].z; nxt[i].w = acc = acc + nxt[i].w; }
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(row + (c + 1) * 16 + 4 * i) = nxt[i];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// the same scan with the LDS reads issued DIST 16-float steps ahead of their use (NB = DIST + 1 register buffers, statically rotated)
template <int NB>
__global__ void k_scanpipe(float *out, long long *cyc) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x, WP = 764;
    for (int i = lane; i < 16 * WP + 256; i += 64) sm[i] = 1.f;
    __syncthreads();
    float acc = 0.f;
    long long t0 = __builtin_readcyclecounter();
    if (lane < 16) {
        float *row = sm + lane * WP + 4;
        float4 buf[NB][4];
#pragma unroll
        for (int b = 0; b < NB - 1; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) buf[b][i] = *reinterpret_cast<float4 *>(row + b * 16 + 4 * i);
#pragma unroll 1
        for (int c = 0; c + NB <= 46 + NB; c += NB) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int nb = (b + NB - 1) % NB;
#pragma unroll
                for (int i = 0; i < 4; i++) buf[nb][i] = *reinterpret_cast<float4 *>(row + (c + b + NB - 1) * 16 + 4 * i);
#pragma unroll
                for (int i = 0; i < 4; i++) { buf[b][i].x = acc = acc + buf[b][i].x; buf[b][i].y = acc = acc + buf[b][i].y; buf[b][i].z = acc = acc + buf[b][i].z; buf[b][i].w = acc = acc + buf[b][i].w; }
#pragma unroll
                for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(row + (c + b) * 16 + 4 * i) = buf[b][i];
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_addchain, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("dependent v_add_f32: %.2f cycles each (4096 adds, %lld cycles)\n", h / 4096.0, h);
        hipLaunchKernelGGL(k_lds<16>, dim3(1), dim3(64), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("4x ds_read_b128 + 4x ds_write_b128, 16 lanes: %.1f cycles per group of 8\n", h / 256.0);
        hipLaunchKernelGGL(k_lds<64>, dim3(1), dim3(64), 0, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("4x ds_read_b128 + 4x ds_write_b128, 64 lanes: %.1f cycles per group of 8\n", h / 256.0);
        hipLaunchKernelGGL(k_scanstep<16>, dim3(1), dim3(64), 16 * 764 * 4 + 1024, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("scan of 736 floats per lane, 16 lanes: %lld cycles = %.1f per element\n", h, h / 736.0);
        hipLaunchKernelGGL(k_scanstep<64>, dim3(1), dim3(64), 16 * 764 * 4 + 1024, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("scan of 736 floats per lane, 64 lanes: %lld cycles = %.1f per element\n", h, h / 736.0);
        hipLaunchKernelGGL(k_scanpipe<2>, dim3(1), dim3(64), 16 * 764 * 4 + 2048, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("pipelined scan, reads 1 step ahead: %lld cycles = %.1f per element\n", h, h / 768.0);
        hipLaunchKernelGGL(k_scanpipe<3>, dim3(1), dim3(64), 16 * 764 * 4 + 2048, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("pipelined scan, reads 2 steps ahead: %lld cycles = %.1f per element\n", h, h / 768.0);
        hipLaunchKernelGGL(k_scanpipe<4>, dim3(1), dim3(64), 16 * 764 * 4 + 2048, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("pipelined scan, reads 3 steps ahead: %lld cycles = %.1f per element\n", h, h / 768.0);
        hipLaunchKernelGGL(k_scanpipe<6>, dim3(1), dim3(64), 16 * 764 * 4 + 2048, 0, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("pipelined scan, reads 5 steps ahead: %lld cycles = %.1f per element\n", h, h / 768.0);
    }
    return 0;
}
