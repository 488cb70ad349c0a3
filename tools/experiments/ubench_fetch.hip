// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this path, with known byte counts
// and a working set far past the 256 MB Infinity Cache: wide and narrow coalesced streams, random 16-byte
// and 2-byte reads at 64-byte granules (the two gathers of k_try_velrot), a 4-byte store stream.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_fetch.hip -o tools/experiments/bin/ubench_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -- tools/experiments/bin/ubench_fetch      (and a second pass with WRITE_SIZE)
// The program prints, per kernel, the bytes a perfect memory system would move at 32 / 64 / 128-byte granularity;
// tools/experiments/fetch_calibration.py divides them by the counters.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <unordered_set>
constexpr size_t BYTES = 2ull << 30;            // 2 GiB buffer
constexpr uint32_t NG = 1u << 24;               // gathers per kernel (16.7 M)
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_stream16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream4(const uint32_t *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream2(const uint16_t *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_gather16(const uint8_t *p, uint32_t ngran, uint32_t *sink) {   // 16 B at the start of a random 64-B granule
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)(hash32(i) % ngran) * 64);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}
__global__ void k_gather2(const uint8_t *p, uint32_t ngran, uint32_t *sink) {    // 2 B inside a random 64-B granule
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t hsh = hash32(i ^ 0x9e3779b9u);
    const uint16_t v = *reinterpret_cast<const uint16_t *>(p + (size_t)(hsh % ngran) * 64 + ((hsh >> 27) & 30));
    if (v == 0x1234u) sink[0] = v;
}
__global__ void k_store4(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
static void distinct(const char *name, uint32_t salt, bool two) {
    const uint32_t ngran = (uint32_t)(BYTES / 64);
    std::vector<uint8_t> s32(BYTES / 32 / 8 + 1, 0), s64(BYTES / 64 / 8 + 1, 0), s128(BYTES / 128 / 8 + 1, 0);
    size_t d32 = 0, d64 = 0, d128 = 0;
    auto mark = [](std::vector<uint8_t> &b, size_t i, size_t &cnt) { if (!(b[i >> 3] & (1 << (i & 7)))) { b[i >> 3] |= 1 << (i & 7); cnt++; } };
    for (uint32_t i = 0; i < NG; i++) {
        const uint32_t hsh = hash32(i ^ salt);
        const size_t off = (size_t)(hsh % ngran) * 64 + (two ? ((hsh >> 27) & 30) : 0);
        mark(s32, off / 32, d32); mark(s64, off / 64, d64); mark(s128, off / 128, d128);
    }
    printf("EXPECT %s requested=%zu granule32=%zu granule64=%zu granule128=%zu reads=%zu\n", name, (size_t)NG * (two ? 2 : 16), d32 * 32, d64 * 64, d128 * 128, (size_t)NG);
}
int main() {
    uint8_t *buf; uint32_t *sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const dim3 g(256 * 32), b(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream16, g, b, 0, 0, (const uint4 *)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_stream4, g, b, 0, 0, (const uint32_t *)buf, BYTES / 4, sink);
        hipLaunchKernelGGL(k_stream2, g, b, 0, 0, (const uint16_t *)buf, BYTES / 2, sink);
        hipLaunchKernelGGL(k_gather16, dim3(NG / 256), b, 0, 0, buf, (uint32_t)(BYTES / 64), sink);
        hipLaunchKernelGGL(k_gather2, dim3(NG / 256), b, 0, 0, buf, (uint32_t)(BYTES / 64), sink);
        hipLaunchKernelGGL(k_store4, g, b, 0, 0, (uint32_t *)buf, BYTES / 4);
        hipDeviceSynchronize();
    }
    printf("EXPECT k_stream16 requested=%zu granule32=%zu granule64=%zu granule128=%zu\n", BYTES, BYTES, BYTES, BYTES);
    printf("EXPECT k_stream4 requested=%zu granule32=%zu granule64=%zu granule128=%zu\n", BYTES, BYTES, BYTES, BYTES);
    printf("EXPECT k_stream2 requested=%zu granule32=%zu granule64=%zu granule128=%zu\n", BYTES, BYTES, BYTES, BYTES);
    distinct("k_gather16", 0u, false);
    distinct("k_gather2", 0x9e3779b9u, true);
    printf("EXPECT k_store4 requested=%zu granule32=%zu granule64=%zu granule128=%zu\n", BYTES, BYTES, BYTES, BYTES);
    return 0;
}
