// stage_a.hip — scale space + KeyLine extraction on the GPU, bit-exact with the reference CPU path.
//
// Replaces (reference file:line)
//   Image<float>::ConvertRGB2BW            include/VideoLib/image.h:197-203
//   iimage::load / iimage::average         src/mtracklib/iimage.cpp:53-71, 86-128
//   iigauss::smooth                        src/mtracklib/iigauss.cpp:91-101
//   sspace::build/build_dog/calc_gradient  src/mtracklib/sspace.cpp:52-85
//   edge_finder::build_mask/join_edges/detect/reEstimateThresh   src/mtracklib/edge_finder.cpp:67-405
//
// Why it is laid out this way.  The reference smooths with iterated box filters on a *float32* integral
// image whose entries exceed 2^24, so every prefix-sum add rounds and the result depends on the
// sequential left-to-right / top-to-bottom order.  Bit-exact edge masks therefore need the same serial
// order inside each row and each column; the parallelism is across rows (row pass), across columns
// (column pass), across the two filters and across the nseq batched sequences.  Row passes go through an
// LDS transpose tile so that global traffic stays coalesced while one lane walks one row.
//
// Launch sequence per frame (all batched over blockIdx.z = sequence):
//   k_rgb_rowscan      RGB24 -> grey -> exact integer row prefix (row sums < 2^24, so order-free)
//   k_colscan          serial column prefix, one thread per column            (x nlevels+1)
//   k_avg_rowscan      box average from 4 integral taps fused with the next serial row prefix (x nlevels)
//   k_detect           last box average of both filters + DoG + gradient + build_mask tests + plane fit;
//                      candidates are staged per (band, wave) strip in raster order
//   k_strip_scan       exclusive scan of strip counts, kl_max truncation, P-controller state update
//   k_emit             raster-order KeyLine SoA + img_mask_kl ids, n_m extremes
//   k_join_histo       join_edges (n_id, atomicMax p_id) + reEstimateThresh histogram
//   k_retune           reEstimateThresh tail (incl. its off-by-one accumulation quirk)
//
// Compile with -ffp-contract=off: the reference is built without FMA contraction.

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "ctx.h"
#include "stage_a_dev.h"

namespace edgehip {

// ---------------------------------------------------------------------------------------------------
// k_rgb_rowscan: one wave per image row.  b+g+r <= 765 and a row has <= 2^14 pixels, so the row prefix
// stays below 2^24 and float addition of these integers is exact: any order gives the reference's bits
// (iimage.cpp:56-61).  Each lane owns CH consecutive pixels; wave-level exclusive scan of lane totals.
// ---------------------------------------------------------------------------------------------------
// (undist_row6 / undist_mix / undist_rgb: stage_a_dev.h — the one-kernel stage A resamples with them too)
template <int CH, bool UNDIST>
__global__ __launch_bounds__(256) void k_rgb_rowscan(const uint8_t *__restrict__ rgb, const int32_t *__restrict__ fidx,
                                                     float *__restrict__ dst, int w, int h, size_t n,
                                                     const int32_t *__restrict__ und_base, const uint4 *__restrict__ und_iw) {
    const int seq = blockIdx.z;
    rgb += (size_t)(fidx ? fidx[seq] : seq) * n * 3;   // this sequence's frame: its slice of the slot, or a frame of a bound pool
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + wave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per-wave staging of one RGB row (w*3 bytes, dword padded)
    const int row_bytes = w * 3;
    const int row_dw = (row_bytes + 3) >> 2;
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem) + (size_t)wave * row_dw;
    if (y < h) {
        if (UNDIST) {
            // undistortion fused into the load: lane x resamples pixel (x, y) from the distorted frame
            const uint8_t *frame = rgb;
            unsigned char *sbw = reinterpret_cast<unsigned char *>(stage);
            for (int x = lane; x < w; x += 64) {
                const size_t pix = (size_t)y * w + x;
                const uchar3 c = undist_rgb(frame, und_base[pix], und_iw[pix], w, (int)n);
                sbw[x * 3] = c.x; sbw[x * 3 + 1] = c.y; sbw[x * 3 + 2] = c.z;
            }
        } else {
            // dword loads from the row's first byte: aligned when w % 4 == 0, any byte address otherwise (gfx9 global loads take
            // it); the last dword of a row whose length is not a multiple of 4 reads up to 3 bytes of the next row / of the 16
            // bytes of slack every frame store has behind it — staged, never used
            const uint8_t *src = rgb + (size_t)y * row_bytes;
            for (int i = lane; i < row_dw; i += 64) { uint32_t v; __builtin_memcpy(&v, src + 4 * (size_t)i, 4); stage[i] = v; }
        }
    }
    __syncthreads();
    if (y >= h) return;
    const unsigned char *sb = reinterpret_cast<const unsigned char *>(stage);
    const int x0 = lane * CH;
    int v[CH];
    int run = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int x = x0 + i;
        int s = 0;
        if (x < w) s = (int)sb[x * 3] + (int)sb[x * 3 + 1] + (int)sb[x * 3 + 2];
        run += s;
        v[i] = run;
    }
    // exclusive scan of lane totals across the wave
    int tot = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(tot, off, 64);
        if (lane >= off) tot += t;
    }
    const int base = tot - run;
    float *out = dst + (size_t)seq * n + (size_t)y * w + x0;
    if ((w & 3) == 0) {
#pragma unroll
        for (int i = 0; i < CH; i += 4) {
            if (x0 + i < w) {  // w % 4 == 0: whole float4 or nothing (rows and x0 are 16-byte aligned)
                float4 o;
                o.x = (float)(v[i] + base);
                o.y = (float)(v[i + 1] + base);
                o.z = (float)(v[i + 2] + base);
                o.w = (float)(v[i + 3] + base);
                *reinterpret_cast<float4 *>(out + i) = o;
            }
        }
    } else {   // any other width: rows start at any float, the last group of a row may be partial
#pragma unroll
        for (int i = 0; i < CH; i++)
            if (x0 + i < w) out[i] = (float)(v[i] + base);
    }
}

// ---------------------------------------------------------------------------------------------------
// k_colscan: img(x,y) += img(x,y-1) for y = 1..h-1, serial per column (iimage.cpp:63-67).  One thread per
// column; rows are fetched UNR at a time so the loads (independent of the add chain) overlap it.
// blockIdx.y selects the plane.
// ---------------------------------------------------------------------------------------------------
struct PlanePtrs {
    float *p[2];
};

template <int UNR>
__global__ __launch_bounds__(64) void k_colscan(PlanePtrs planes, int w, int h, size_t n) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    float *img = planes.p[blockIdx.y] + (size_t)blockIdx.z * n + x;
    float run = img[0];
    int y = 1;
    float cur[UNR], nxt[UNR];
    if (y + UNR <= h) {
#pragma unroll
        for (int i = 0; i < UNR; i++) cur[i] = img[(size_t)(y + i) * w];
    }
    for (; y + UNR <= h; y += UNR) {
        const bool more = (y + 2 * UNR <= h);
        if (more) {
#pragma unroll
            for (int i = 0; i < UNR; i++) nxt[i] = img[(size_t)(y + UNR + i) * w];
        }
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            run = cur[i] + run;  // img(x,y) += img(x,y-1): operand order irrelevant for IEEE add
            img[(size_t)(y + i) * w] = run;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < UNR; i++) cur[i] = nxt[i];
        }
    }
    for (; y < h; y++) {
        run = img[(size_t)y * w] + run;
        img[(size_t)y * w] = run;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_colscan_chain: the same serial column prefix for a few planes (a live camera), where k_colscan's one wave per
// 64 columns pays h / UNR dependent load round trips.  NW waves per 64 columns: every wave loads its band of rows
// at once (all h loads of a column in flight together), then the bands take turns down the column — wave k adds
// its rows onto the carry wave k-1 left in LDS — and every wave stores its band.  The chain of h dependent float
// adds per column is the reference's, so are its bits.
// ---------------------------------------------------------------------------------------------------
template <int NW, int RPW>
__global__ __launch_bounds__(NW * 64) void k_colscan_chain(PlanePtrs planes, int w, int h, size_t n) {
    __shared__ float carry[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const bool on = x < w;
    float *img = planes.p[blockIdx.y] + (size_t)blockIdx.z * n + (on ? x : 0);
    const int rpw = (h + NW - 1) / NW;
    const int y0 = wv * rpw;
    const int cnt = min(h - y0, rpw);   // rows of this wave (<= 0: none)
    float v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++) v[i] = (on && i < cnt) ? img[(size_t)(y0 + i) * w] : 0.f;
    for (int k = 0; k < NW; k++) {
        if (wv == k && cnt > 0) {
            float run = k ? carry[lane] : 0.f;
#pragma unroll
            for (int i = 0; i < RPW; i++) {
                if (i < cnt) {
                    run = (k == 0 && i == 0) ? v[0] : v[i] + run;   // img(x,0) stays; img(x,y) += img(x,y-1)
                    v[i] = run;
                }
            }
            carry[lane] = run;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPW; i++)
        if (on && i < cnt) img[(size_t)(y0 + i) * w] = v[i];
}

// ---------------------------------------------------------------------------------------------------
// Box average from an integral image: iimage::average's nine border regions collapse to one formula
// with zero-substituted taps (x - 0 and x + 0 are exact), EXCEPT that the bottom band (y >= h-d2)
// subtracts the upper tap before the left tap (iimage.cpp:118-126 vs :105-113).  The multiplier is
// div(x,y) = (float)(1.0/count); in the interior the reference uses a = (float)(1.0/(d*d)), which is the
// same float, so one LUT indexed by count serves both.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_avg(const float *__restrict__ ii, int x, int y, int w, int h, int d, int d2,
                                         const float *__restrict__ lut) {
    int xr = x + d2, yb = y + d2;
    const int xl = x - d2 - 1, yt = y - d2 - 1;
    int cx = d, cy = d;
    if (xl < 0) cx = x + d2 + 1;
    if (xr > w - 1) { xr = w - 1; cx = w - x + d2; }
    if (yt < 0) cy = y + d2 + 1;
    const bool bottom = yb > h - 1;
    if (bottom) { yb = h - 1; cy = h - y + d2; }
    // unconditional (clamped) loads + selects: no divergent branches, all four taps in flight together
    const int xlc = xl < 0 ? 0 : xl, ytc = yt < 0 ? 0 : yt;
    const float A = ii[(size_t)yb * w + xr];
    float B = ii[(size_t)yb * w + xlc];
    float C = ii[(size_t)ytc * w + xr];
    float D = ii[(size_t)ytc * w + xlc];
    const float m = lut[cx * cy];
    B = xl >= 0 ? B : 0.f;
    C = yt >= 0 ? C : 0.f;
    D = (xl >= 0 && yt >= 0) ? D : 0.f;
    const float t = bottom ? ((A - C) - B) + D : ((A - B) - C) + D;
    return t * m;
}

// Same arithmetic, split so that callers can put the tap loads of several pixels in flight before any use.
struct BoxTaps {
    float A, B, C, D, m;
    bool bottom;
};
__device__ __forceinline__ BoxTaps box_taps(const float *__restrict__ ii, int x, int y, int w, int h, int d, int d2,
                                            const float *__restrict__ lut, float a_int) {
    int xr = x + d2, yb = y + d2;
    const int xl = x - d2 - 1, yt = y - d2 - 1;
    int cx = d, cy = d;
    if (xl < 0) cx = x + d2 + 1;
    if (xr > w - 1) { xr = w - 1; cx = w - x + d2; }
    if (yt < 0) cy = y + d2 + 1;
    BoxTaps t;
    t.bottom = yb > h - 1;
    if (t.bottom) { yb = h - 1; cy = h - y + d2; }
    const int xlc = xl < 0 ? 0 : xl, ytc = yt < 0 ? 0 : yt;
    t.A = ii[(size_t)yb * w + xr];
    t.B = ii[(size_t)yb * w + xlc];
    t.C = ii[(size_t)ytc * w + xr];
    t.D = ii[(size_t)ytc * w + xlc];
    t.m = a_int;                                  // interior: a = (float)(1.0/(d*d)) == lut[d*d]
    if (cx * cy != d * d) t.m = lut[cx * cy];     // border pixels only (rare, mostly wave-uniform)
    t.B = xl >= 0 ? t.B : 0.f;
    t.C = yt >= 0 ? t.C : 0.f;
    t.D = (xl >= 0 && yt >= 0) ? t.D : 0.f;
    return t;
}
__device__ __forceinline__ float box_combine(const BoxTaps &t) {
    const float s = t.bottom ? ((t.A - t.C) - t.B) + t.D : ((t.A - t.B) - t.C) + t.D;
    return s * t.m;
}

// ---------------------------------------------------------------------------------------------------
// k_avg_rowscan: dst(x,y) = sum_{x'<=x} avg(src)(x',y), the box average fused with the serial row prefix
// of the next iimage::load (iigauss.cpp:95-98).  A block owns RB rows; it walks the row in TC-column
// tiles: all threads compute the averages of the tile into LDS (coalesced taps), the first wave walks
// one row per lane through LDS (conflict-free: row stride TC+1), all threads store the tile coalesced.
// blockIdx.y selects (src, dst, d).
// ---------------------------------------------------------------------------------------------------
struct AvgJob {
    const float *src[2];
    float *dst[2];
    int d[2];
};

template <int NT, int RB, int TC>
__global__ __launch_bounds__(NT) void k_avg_rowscan(AvgJob job, const float *__restrict__ lut, int w, int h, size_t n) {
    __shared__ float tile[RB][TC + 1];
    const int seq = blockIdx.z;
    const float *src = job.src[blockIdx.y] + (size_t)seq * n;
    float *dst = job.dst[blockIdx.y] + (size_t)seq * n;
    const int d = job.d[blockIdx.y], d2 = d / 2;
    const int y0 = blockIdx.x * RB;
    const int tid = threadIdx.x;
    float run = 0.f;  // running row sum of lane's row (first wave only)
    for (int x0 = 0; x0 < w; x0 += TC) {
        for (int idx = tid; idx < RB * TC; idx += NT) {
            const int r = idx / TC, c = idx - r * TC;
            const int x = x0 + c, y = y0 + r;
            float v = 0.f;
            if (x < w && y < h) v = box_avg(src, x, y, w, h, d, d2, lut);
            tile[r][c] = v;
        }
        __syncthreads();
        if (tid < RB) {
            constexpr int CH = TC < 64 ? TC : 64;   // registers hold 64 values of the row at a time
#pragma unroll 1
            for (int cc = 0; cc < TC; cc += CH) {
                float vals[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) vals[c] = tile[tid][cc + c];
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    run = (x0 + cc + c == 0) ? vals[c] : run + vals[c];  // img(0,y)=l(0,y); img(x,y)=img(x-1,y)+l(x,y)
                    vals[c] = run;
                }
#pragma unroll
                for (int c = 0; c < CH; c++) tile[tid][cc + c] = vals[c];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < RB * TC; idx += NT) {
            const int r = idx / TC, c = idx - r * TC;
            const int x = x0 + c, y = y0 + r;
            if (x < w && y < h) dst[(size_t)y * w + x] = tile[r][c];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_avg_rowscan_wide: k_avg_rowscan for a few planes (a live camera).  A block owns RB whole rows: the taps of all
// its pixels are in flight at once (PT pixels per thread, registers), the averages go to LDS, RB lanes walk one row
// each (the reference's chain of w dependent adds), everything is stored coalesced — one round trip to memory per
// block instead of one per 256-column tile.  Same operations in the same order.
// ---------------------------------------------------------------------------------------------------
template <int NT, int RB, int TC>
__global__ __launch_bounds__(NT) void k_avg_rowscan_wide(AvgJob job, const float *__restrict__ lut, int w, int h, size_t n) {
    __shared__ float tile[RB][TC + 1];
    constexpr int PT = RB * TC / NT;
    const int seq = blockIdx.z;
    const float *src = job.src[blockIdx.y] + (size_t)seq * n;
    float *dst = job.dst[blockIdx.y] + (size_t)seq * n;
    const int d = job.d[blockIdx.y], d2 = d / 2;
    const float a_int = lut[d * d];
    const int y0 = blockIdx.x * RB;
    const int tid = threadIdx.x;
    BoxTaps t[PT];
#pragma unroll
    for (int j = 0; j < PT; j++) {
        const int idx = tid + j * NT, r = idx / TC, c = idx - r * TC;
        const int x = min(c, w - 1), y = min(y0 + r, h - 1);
        t[j] = box_taps(src, x, y, w, h, d, d2, lut, a_int);
    }
#pragma unroll
    for (int j = 0; j < PT; j++) {
        const int idx = tid + j * NT, r = idx / TC, c = idx - r * TC;
        tile[r][c] = box_combine(t[j]);
    }
    __syncthreads();
    if (tid < RB) {
        static_assert(TC % 64 == 0, "whole 64-column chunks");
        float run = 0.f;
#pragma unroll 1
        for (int cc = 0; cc < w; cc += 64) {   // columns >= w of the last chunk hold averages of clamped pixels: added after the row's last real column, never stored
            float vals[64];
#pragma unroll
            for (int c = 0; c < 64; c++) vals[c] = tile[tid][cc + c];
#pragma unroll
            for (int c = 0; c < 64; c++) {
                run = (cc + c == 0) ? vals[c] : run + vals[c];  // img(0,y)=l(0,y); img(x,y)=img(x-1,y)+l(x,y)
                vals[c] = run;
            }
#pragma unroll
            for (int c = 0; c < 64; c++) tile[tid][cc + c] = vals[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT; j++) {
        const int idx = tid + j * NT, r = idx / TC, c = idx - r * TC;
        if (c < w && y0 + r < h) dst[(size_t)(y0 + r) * w + c] = tile[r][c];
    }
}

// ---------------------------------------------------------------------------------------------------
// k_level: one level of the box chain for a whole plane in ONE pass — source values (grey of the RGB frame,
// or the box average of the previous integral image), serial row prefix, serial column prefix — written once.
// Replaces k_rgb_rowscan / k_avg_rowscan + k_colscan when enough planes are in flight to fill the GPU
// (one workgroup per (sequence, plane); HBM traffic per level drops from 16N to 8N bytes per plane).
//
// The float32 rounding order is the reference's: along a row img(x,y) = img(x-1,y) + l(x,y) strictly left to
// right, then down a column img(x,y) += img(x,y-1) strictly top to bottom (iimage.cpp:56-67).  A block walks
// the plane in batches of LV_RB rows through two LDS buffers:
//     waves 1..12   thread <-> column: produce the batch's source values (coalesced), later add the scanned
//                   batch onto the running column sums held in registers and store the rows (coalesced)
//     wave 0        lane <-> row: serial left-to-right prefix of the batch's rows in LDS (float4 steps; the row
//                   stride WP has WP/4 odd, so the 16 lanes hit 16 disjoint bank quads)
// and the phases are software-pipelined: scan(k+1) runs while the column threads finish batch k and produce
// batch k+2 — one barrier per batch.
// ---------------------------------------------------------------------------------------------------
constexpr int LV_RB = 16;                 // rows per batch
constexpr int LV_NT = 832;                // 13 waves
constexpr int LV_NC = LV_NT - 64;         // column-owner threads
constexpr int LV_MAXCOL = 2;              // max columns per owner thread (w <= 1536); template parameter MC picks 1 or 2

__host__ __device__ inline int level_row_stride(int w) {
    int wp = (w + 3) & ~3;
    if (((wp >> 2) & 1) == 0) wp += 4;
    return wp;
}

struct LevelJob {
    const float *src[2];
    float *dst[2];
    int d[2];
    float a[2];
};

template <int SRC, int MC>   // SRC 0: box average of job.src; 1: grey of the RGB24 frame; 2: grey of the undistorted frame
__global__ __launch_bounds__(LV_NT) void k_level(LevelJob job, const uint8_t *__restrict__ rgb, const int32_t *__restrict__ fidx,
                                                 const int32_t *__restrict__ und_base, const uint4 *__restrict__ und_iw,
                                                 const float *__restrict__ lut, int w, int h, size_t n, int ablate) {
    extern __shared__ __attribute__((aligned(16))) float s_T[];   // [2][LV_RB][WP]
    const int WP = level_row_stride(w);
    const int seq = blockIdx.z, jb = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float *src = SRC == 0 ? job.src[jb] + (size_t)seq * n : nullptr;
    float *dst = job.dst[jb] + (size_t)seq * n;
    const uint8_t *frame = SRC != 0 ? rgb + (size_t)(fidx ? fidx[seq] : seq) * n * 3 : nullptr;
    const int d = job.d[jb], d2 = d / 2;
    const bool reuse = d == 3 || d == 5;   // vertical tap reuse (SRC 0)
    const int nb = (h + LV_RB - 1) / LV_RB;
    const int ct = tid - 64;              // column-owner index (waves 1..12)
    float *s_lut = s_T + (size_t)2 * LV_RB * WP + 32;   // [kDivLutMax] reciprocal-count table (after the scan's tail pad)
    if (SRC == 0 && tid < kDivLutMax) s_lut[tid] = lut[tid];
    float run[MC];                        // running column sums of the owned columns
#pragma unroll
    for (int j = 0; j < MC; j++) run[j] = 0.f;

    // Source values of a batch are produced in two steps so that the global-load latency of batch k+1 hides
    // behind the column work and the barrier of batch k: issue(k) starts the loads into registers, commit(k)
    // (one iteration later) turns them into the batch's LDS rows.
    //   SRC 0: 4 integral-image taps per pixel, iimage::average with the column geometry hoisted out of the
    //          row loop (which taps exist and the box width along x depend on the column only; the clipped rows
    //          and the box height on the wave-uniform row).
    //   SRC 1: the 3 bytes of the RGB pixel.     SRC 2: undistorted on the fly at commit time (no prefetch).
    float tp[MC][LV_RB][4];
    auto issue = [&](int k) {
        if ((ablate & 2) || SRC == 2) return;
        const int y0 = k * LV_RB;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            const int x = ct + j * LV_NC;
            if (x >= w) break;
            if constexpr (SRC == 0) {
                const int xl = x - d2 - 1;
                const int xlc = xl >= 0 ? xl : 0;
                const int xr = x + d2 > w - 1 ? w - 1 : x + d2;
                // The top taps of row r (integral row y-d2-1) are the bottom taps of row r-d (integral row y-d+d2): with
                // the usual box widths (3 and 5) only the first 5 rows of a batch load their top taps, the others reuse
                // registers in commit() — 42 loads per column and batch instead of 64.
#pragma unroll
                for (int r = 0; r < LV_RB; r++) {
                    int y = y0 + r;
                    y = y < h ? y : h - 1;                           // rows past the image: harmless duplicates
                    const int yb = y + d2 > h - 1 ? h - 1 : y + d2;
                    const float *rowb = src + (size_t)yb * w;
                    tp[j][r][0] = rowb[xr];
                    tp[j][r][1] = rowb[xlc];
                    if (r < 5 || !reuse) {
                        const int yt = y - d2 - 1;
                        const float *rowt = src + (size_t)(yt < 0 ? 0 : yt) * w;
                        tp[j][r][2] = rowt[xr];
                        tp[j][r][3] = rowt[xlc];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < LV_RB; r++) {
                    int y = y0 + r;
                    y = y < h ? y : h - 1;
                    // the pixel's 3 bytes sit inside the 8 bytes that start at its dword-aligned address: one
                    // 64-bit load per pixel instead of three byte loads (the rgb allocation has 16 B of slack)
                    const size_t byte0 = ((size_t)y * w + x) * 3;
                    const uint2 q = *reinterpret_cast<const uint2 *>(frame + (byte0 & ~(size_t)3));
                    tp[j][r][0] = __uint_as_float(q.x);
                    tp[j][r][1] = __uint_as_float(q.y);
                }
            }
        }
    };
    auto commit = [&](int k) {
        if (ablate & 2) return;
        float *T = s_T + (size_t)(k & 1) * LV_RB * WP;
        const int y0 = k * LV_RB;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            const int x = ct + j * LV_NC;
            if (x >= w) break;
            if (SRC == 0) {
                const bool hasL = x - d2 - 1 >= 0;
                int cx = d;
                if (!hasL) cx = x + d2 + 1;
                if (x + d2 > w - 1) cx = w - x + d2;
#pragma unroll
                for (int r = 0; r < LV_RB; r++) {
                    int y = y0 + r;
                    y = y < h ? y : h - 1;
                    int cy = d;
                    const int yt = y - d2 - 1;
                    if (yt < 0) cy = y + d2 + 1;
                    const bool bot = y + d2 > h - 1;
                    if (bot) cy = h - y + d2;
                    const float A = tp[j][r][0];
                    float Bv = tp[j][r][1], C = tp[j][r][2], D = tp[j][r][3];
                    // top taps: the bottom taps of row r-d when that row is in this batch (all static register indices;
                    // for a row past the image, whose result is discarded, the two would differ)
                    const int r3 = r >= 3 ? r - 3 : 0, r5 = r >= 5 ? r - 5 : 0;
                    const float c3 = tp[j][r3][0], d3 = tp[j][r3][1], c5 = tp[j][r5][0], d5 = tp[j][r5][1];
                    const bool u3 = r >= 3 && d == 3, u5 = r >= 5 && d == 5;
                    C = u3 ? c3 : u5 ? c5 : C;
                    D = u3 ? d3 : u5 ? d5 : D;
                    if (!hasL) { Bv = 0.f; D = 0.f; }
                    if (yt < 0) { C = 0.f; D = 0.f; }
                    // div(x,y) = (float)(1.0/count) from the LDS copy of the table: a global load here would sit
                    // behind the row stores of finish() in the vmcnt queue.  Interior: lut[d*d] == a.
                    const float m = s_lut[cx * cy];
                    const float sum = bot ? ((A - C) - Bv) + D : ((A - Bv) - C) + D;
                    T[r * WP + x] = sum * m;
                }
            } else if (SRC == 1) {
#pragma unroll
                for (int r = 0; r < LV_RB; r++) {   // b+g+r (image.h:197-203): integers, exact in float
                    int y = y0 + r;
                    y = y < h ? y : h - 1;
                    const unsigned sh = (unsigned)((((size_t)y * w + x) * 3) & 3) * 8;
                    const unsigned long long q8 = ((unsigned long long)__float_as_uint(tp[j][r][1]) << 32) | __float_as_uint(tp[j][r][0]);
                    const unsigned p = (unsigned)(q8 >> sh);
                    T[r * WP + x] = (float)((int)(p & 0xFF) + (int)((p >> 8) & 0xFF) + (int)((p >> 16) & 0xFF));
                }
            } else {
#pragma unroll 4
                for (int r = 0; r < LV_RB; r++) {
                    const int y = y0 + r;
                    float v = 0.f;
                    if (y < h) {
                        const size_t pix = (size_t)y * w + x;
                        const uchar3 c = undist_rgb(frame, und_base[pix], und_iw[pix], w, (int)n);
                        v = (float)((int)c.x + (int)c.y + (int)c.z);
                    }
                    T[r * WP + x] = v;
                }
            }
        }
    };
    // serial row prefix of batch k (wave 0, one lane per row)
    auto scan = [&](int k) {
        if (lane >= LV_RB || k * LV_RB + lane >= h || (ablate & 1)) return;
        float *row = s_T + (size_t)(k & 1) * LV_RB * WP + (size_t)lane * WP;
        // LV_SC floats per step; the next step's LDS reads are issued (unconditionally: the row pad / the next row /
        // the tail pad of the allocation absorb the overrun) before this step's add chain so their latency hides
        // behind it, and the loop body stays branch-free.
        constexpr int LV_SC = 16;
        const int nfull = w / LV_SC;
        float acc = 0.f;
        int c = 0;
        if (nfull > 0) {
            float4 cur[LV_SC / 4], nxt[LV_SC / 4];
#pragma unroll
            for (int i = 0; i < LV_SC / 4; i++) cur[i] = *reinterpret_cast<float4 *>(row + 4 * i);
            for (int ch = 0; ch < nfull; ch++, c += LV_SC) {
#pragma unroll
                for (int i = 0; i < LV_SC / 4; i++) nxt[i] = *reinterpret_cast<float4 *>(row + c + LV_SC + 4 * i);
                // img(0,y) = l(0,y); img(x,y) = img(x-1,y) + l(x,y)
                cur[0].x = acc = ch == 0 ? cur[0].x : acc + cur[0].x;
                cur[0].y = acc = acc + cur[0].y;
                cur[0].z = acc = acc + cur[0].z;
                cur[0].w = acc = acc + cur[0].w;
#pragma unroll
                for (int i = 1; i < LV_SC / 4; i++) {
                    cur[i].x = acc = acc + cur[i].x;
                    cur[i].y = acc = acc + cur[i].y;
                    cur[i].z = acc = acc + cur[i].z;
                    cur[i].w = acc = acc + cur[i].w;
                }
#pragma unroll
                for (int i = 0; i < LV_SC / 4; i++) *reinterpret_cast<float4 *>(row + c + 4 * i) = cur[i];
#pragma unroll
                for (int i = 0; i < LV_SC / 4; i++) cur[i] = nxt[i];
            }
        }
        for (; c < w; c += 4) {   // w % 16 != 0: up to three more float4 steps
            float4 v = *reinterpret_cast<float4 *>(row + c);
            v.x = acc = c == 0 ? v.x : acc + v.x;
            v.y = acc = acc + v.y;
            v.z = acc = acc + v.z;
            v.w = acc = acc + v.w;
            *reinterpret_cast<float4 *>(row + c) = v;
        }
    };
    // column prefix of batch k onto the running sums + store
    auto finish = [&](int k) {
        if (ablate & 4) return;
        const float *T = s_T + (size_t)(k & 1) * LV_RB * WP;
        const int y0 = k * LV_RB;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            const int x = ct + j * LV_NC;
            if (x >= w) break;
            float v[LV_RB];
#pragma unroll
            for (int r = 0; r < LV_RB; r++) v[r] = T[r * WP + x];
            float acc = run[j];
#pragma unroll
            for (int r = 0; r < LV_RB; r++) {
                acc = (y0 + r == 0) ? v[r] : v[r] + acc;    // img(x,y) += img(x,y-1)
                v[r] = acc;
            }
            run[j] = acc;
#pragma unroll
            for (int r = 0; r < LV_RB; r++)
                if (y0 + r < h) dst[(size_t)(y0 + r) * w + x] = v[r];
        }
    };

    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);   // the serial scan is the critical path: let it win issue arbitration
    } else {
        issue(0);
        commit(0);
        if (nb > 1) issue(1);
    }
    lds_barrier();
    if (wave == 0) scan(0);
    else if (nb > 1) {
        commit(1);
        if (nb > 2) issue(2);
    }
    lds_barrier();
    for (int k = 0; k < nb; k++) {
        if (wave == 0) {
            if (k + 1 < nb) scan(k + 1);
        } else {
            finish(k);                               // reads buffer k&1 ...
            if (k + 2 < nb) commit(k + 2);           // ... which batch k+2 then overwrites (same thread, same columns)
            if (k + 3 < nb) issue(k + 3);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_detect: one block per band of kBandRows image rows (rows 2..h-3 are scanned, edge_finder.cpp:105).
// Phase 1 builds img0 (=G(sigma0)) and DoG for the band plus a 2-row halo in LDS from the last integral
// images of both filters (last iimage::average of iigauss::smooth + sspace::build_dog).
// Phase 2: wave v owns the v-th quarter of the band's raster order and walks it in 64-pixel chunks; a
// lane tests one pixel (edge_finder.cpp:117-159) and survivors are appended, in raster order, to the
// (band, wave) strip of the staging area using the wave ballot.
// ---------------------------------------------------------------------------------------------------
struct CandStage {  // SoA staging of candidates: [B][nstrips][strip_cap]
    int32_t *p_inx;
    float2 *m;   // (theta0, theta1) as float
    float2 *s;   // (xs, ys)
};

struct DetectArgs {
    const float *iic0, *iic1;  // last integral image of filter0 / filter1, [B][N]
    int d0, d1;                // last box widths
    float a0, a1;              // (float)(1.0/(d*d)) of the last boxes
    const float *lut;
    const double *pinv;        // [3][25]
    float *planes;             // optional debug planes [5][B][N]
    int32_t *mask;             // [B][N] of the slot
    SeqA *seq;
    CandStage st;
    int32_t *strip_cnt;        // [B][nstrips]
    int w, h, nseq, strip_cap;
    size_t n;
    // thresholds (edge_finder::detect arguments)
    double gain, tmax, tmin;
    int kl_ref;
    float dog_thresh_f;        // (float)DetectorDoGThresh
    double pn_thresh;          // (double)((float)((2 win_s + 1)^2) * (float)DetectorPosNegThresh)
    int ablate;                // debug: bit0 skip phase 1 math, bit1 skip 2a, bit2 skip 2b
    int band_rows;             // image rows per band (edgehip_ctx::band_rows: as many as the LDS planes of this width allow)
};

// WS = DetectorPlaneFitSize (win_s of build_mask, edge_finder.cpp:67-100): the DoG window of the sign balance and of the plane
// fit is (2 WS + 1)^2, rows / columns closer than WS to the border are not scanned (:110-111).
template <int WS>
__global__ __launch_bounds__(kDetWaves * 64) void k_detect(DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WIN = 2 * WS + 1, NW2 = WIN * WIN;
    const int w = a.w, h = a.h;
    const int BR = a.band_rows;
    const int HR = BR + 2 * WS;        // rows held in LDS
    float *s_img0 = reinterpret_cast<float *>(smem);   // [HR][w]
    float *s_dog = s_img0 + (size_t)HR * w;            // [HR][w]
    __shared__ double s_pinv[3 * NW2];
    __shared__ float s_lut[kDivLutMax];
    const int seq = blockIdx.z, band = blockIdx.x;
    const int tid = threadIdx.x;
    const int y0 = WS + band * BR;        // first output row of the band
    const int yb0 = y0 - WS;              // first LDS row
    const size_t so = (size_t)seq * a.n;
    const float *iic0 = a.iic0 + so, *iic1 = a.iic1 + so;
    if (tid < 3 * NW2) s_pinv[tid] = a.pinv[tid];
    if (tid < kDivLutMax) s_lut[tid] = a.lut[tid];
    __syncthreads();

    // ---- phase 1: img0 = G(sigma0), DoG = G(sigma1) - G(sigma0) for the band + halo -> LDS -------------------
    // thread <-> column: which taps exist and the box width along x are column properties, hoisted out of the
    // row loop; clipped rows and the box height are wave-uniform.  The tap loads of 4 rows x 2 filters (32 per
    // thread) are all in flight before the first is used; no global load sits inside a branch.
    for (int x = tid; x < w; x += kDetWaves * 64) {
        int xlc[2], xr[2], cx[2], dd[2], dh[2];
        bool hasL[2];
        const float *ii[2] = {iic0, iic1};
        dd[0] = a.d0; dd[1] = a.d1;
#pragma unroll
        for (int f = 0; f < 2; f++) {
            dh[f] = dd[f] / 2;
            const int xl = x - dh[f] - 1;
            hasL[f] = xl >= 0;
            xlc[f] = hasL[f] ? xl : 0;
            xr[f] = x + dh[f];
            cx[f] = dd[f];
            if (!hasL[f]) cx[f] = x + dh[f] + 1;
            if (xr[f] > w - 1) { xr[f] = w - 1; cx[f] = w - x + dh[f]; }
        }
        // bands whose rows all have the full box height (all but the first and last of an image) take the short form of
        // the average: no clipped taps, no bottom-row operand order, div(x,y) = the column's table entry read once
        const int dhm = dh[0] > dh[1] ? dh[0] : dh[1];
        const bool interior = yb0 - dhm - 1 >= 0 && yb0 + HR - 1 + dhm <= h - 1;   // block-uniform
        float lut_in[2];
#pragma unroll
        for (int f = 0; f < 2; f++) lut_in[f] = s_lut[cx[f] * dd[f]];
#pragma nounroll
        for (int r0 = 0; r0 < HR; r0 += 4) {
            float tp[4][2][4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int y = yb0 + r0 + i;
                y = y < h ? y : h - 1;
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    const int yb = y + dh[f] > h - 1 ? h - 1 : y + dh[f];
                    const int yt = y - dh[f] - 1;
                    const float *rowb = ii[f] + (size_t)yb * w;
                    const float *rowt = ii[f] + (size_t)(yt < 0 ? 0 : yt) * w;
                    tp[i][f][0] = rowb[xr[f]];
                    tp[i][f][1] = rowb[xlc[f]];
                    tp[i][f][2] = rowt[xr[f]];
                    tp[i][f][3] = rowt[xlc[f]];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + i, y = yb0 + r;
                if (r >= HR) continue;     // HR is a multiple of 4 for the default band (12 + 4), not for every band height / window
                float g[2];
                if (interior) {
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        const float Bv = hasL[f] ? tp[i][f][1] : 0.f, D = hasL[f] ? tp[i][f][3] : 0.f;
                        g[f] = (((tp[i][f][0] - Bv) - tp[i][f][2]) + D) * lut_in[f];   // iimage.cpp:119-126
                    }
                    s_img0[r * w + x] = g[0];
                    s_dog[r * w + x] = g[1] - g[0];                                    // sspace.cpp:66
                    if (a.planes && ((y >= y0 && y < y0 + BR) || (band == 0 && y < WS) ||
                                     (y >= WS + (int)gridDim.x * BR))) {
                        float *pl = a.planes + so;
                        const size_t pstride = (size_t)a.nseq * a.n;
                        pl[0 * pstride + (size_t)y * w + x] = g[0];
                        pl[1 * pstride + (size_t)y * w + x] = g[1];
                        pl[2 * pstride + (size_t)y * w + x] = g[1] - g[0];
                    }
                    continue;
                }
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    const int yc = y < h ? y : h - 1;
                    int cy = dd[f];
                    const int yt = yc - dh[f] - 1;
                    if (yt < 0) cy = yc + dh[f] + 1;
                    const bool bot = yc + dh[f] > h - 1;
                    if (bot) cy = h - yc + dh[f];
                    const float A = tp[i][f][0];
                    float Bv = tp[i][f][1], C = tp[i][f][2], D = tp[i][f][3];
                    if (!hasL[f]) { Bv = 0.f; D = 0.f; }
                    if (yt < 0) { C = 0.f; D = 0.f; }
                    const float sum = bot ? ((A - C) - Bv) + D : ((A - Bv) - C) + D;   // iimage.cpp:105-126
                    g[f] = sum * s_lut[cx[f] * cy];                                    // div(x,y); interior == a
                }
                const bool inimg = y < h;
                s_img0[r * w + x] = inimg ? g[0] : 0.f;
                s_dog[r * w + x] = inimg ? g[1] - g[0] : 0.f;                          // sspace.cpp:66
                // every image row is written to the debug planes by exactly one band
                if (a.planes && inimg && ((y >= y0 && y < y0 + BR) || (band == 0 && y < WS) ||
                                          (y >= WS + (int)gridDim.x * BR))) {
                    float *pl = a.planes + so;
                    const size_t pstride = (size_t)a.nseq * a.n;
                    pl[0 * pstride + (size_t)y * w + x] = g[0];
                    pl[1 * pstride + (size_t)y * w + x] = g[1];
                    pl[2 * pstride + (size_t)y * w + x] = g[1] - g[0];
                }
            }
        }
    }
    __syncthreads();

    SeqA *sq = a.seq + seq;
    const double tresh = update_thresh(sq->tresh, sq->l_kl_num, a.kl_ref, a.gain, a.tmax, a.tmin);
    const float grad_thresh = (float)tresh;                      // build_mask takes float grad_thesh
    const float gt1 = grad_thresh * 765;                         // grad_thesh*max_img_value (int 765 -> float)
    const float thr_g = gt1 * gt1;                               // util::square(...)
    const float gt2 = gt1 * a.dog_thresh_f;
    const float thr_d = gt2 * gt2;

    // ---- phase 2: tests, raster order inside the wave's quarter of the band ----
    const int wave = tid >> 6, lane = tid & 63;
    const int npx = BR * w;
    const int nchunk = (npx + 63) >> 6;
    const int cpw = (nchunk + kDetWaves - 1) / kDetWaves;  // chunks per wave
    const int strip = band * kDetWaves + wave;
    const int nstrips = gridDim.x * kDetWaves;
    const size_t sbase = ((size_t)seq * nstrips + strip) * a.strip_cap;
    int32_t *mask = a.mask + so;
    // 2a: cheap gradient gate on every pixel of the strip; survivors are compacted (raster order kept)
    //     into a per-wave LDS list so that the expensive window tests run with full lanes.
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_dog + (size_t)HR * w) + (size_t)wave * cpw * 64;
    int nlist = 0;
    if (!(a.ablate & 2)) {
        // A band is a contiguous run of image rows and its LDS planes have the image's row pitch: pixel q of the band (in
        // raster order) is image pixel y0*w + q and LDS element WS*w + q — no (row, column) arithmetic per pixel.  Only
        // the column is tracked, for the WS border columns on either side.
        int q = wave * cpw * 64 + lane;
        int x = q % w;                                           // one division per thread; then advance by 64 pixels per chunk
        const int q_end = min(npx, (h - WS - y0) * w);           // rows y >= h-WS are not scanned (edge_finder.cpp:110)
        int32_t *mrow = mask + (size_t)y0 * w;
        const float *img0c = s_img0 + WS * w;
        for (int ci = 0; ci < cpw; ci++, q += 64, x += 64) {
            while (x >= w) x -= w;
            bool pass = false;
            if (q < q_end) {
                // default: no KeyLine (edge_finder.cpp:109); border columns are never KeyLines either
                mrow[q] = -1;
                if (x >= WS && x < w - WS) {
                    const float *c0 = img0c + q;
                    const float dx = c0[1] - c0[-1];   // sspace.cpp:80
                    const float dy = c0[w] - c0[-w];   // sspace.cpp:81
                    if (a.planes) {
                        float *pl = a.planes + so;
                        const size_t pstride = (size_t)a.nseq * a.n;
                        pl[3 * pstride + (size_t)y0 * w + q] = dx;
                        pl[4 * pstride + (size_t)y0 * w + q] = dy;
                    }
                    const float n2g = dx * dx + dy * dy;
                    pass = !(n2g < thr_g);
                }
            }
            const unsigned long long bal = __ballot(pass);
            if (pass) s_list[nlist + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)q;
            nlist += __popcll(bal);
        }
    }
    // 2b: DoG sign balance, plane fit, sub-pixel zero crossing, DoG-gradient gate (edge_finder.cpp:125-159)
    // PInv(0,k) depends on the window column only, PInv(1,k) on the window row only, PInv(2,k) is constant (the
    // window is symmetric; checked on the host when the table is built): 11 coefficients live in registers
    // instead of 75 LDS reads per candidate.  The accumulation order over k is unchanged.
    // (WS = 2, what every shipped configuration uses; other windows take the matrix as it is.)
    double pc0[5], pc1[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { pc0[j] = WS == 2 ? s_pinv[j] : 0.0; pc1[j] = WS == 2 ? s_pinv[25 + 5 * j] : 0.0; }
    const double pc2 = WS == 2 ? s_pinv[50] : 0.0;
    // 2b': the DoG sign-balance test (:125-137) needs no arithmetic: run it first and compact the list once more
    // (in place: the write index never passes the read index), so the fp64 plane fit below runs with full lanes on
    // the pixels that can still become KeyLines.
    if (!(a.ablate & 4)) {
        int nkeep = 0;
        for (int base = 0; base < nlist; base += 64) {
            const int li = base + lane;
            bool keep = false;
            uint16_t qv = 0;
            if (li < nlist) {
                qv = s_list[li];
                const float *dg = s_dog + WS * w + (int)qv;   // band pixel q = LDS element WS*w + q
                int npos = 0;   // pn = (#positive) - (#not positive) = 2*npos - (2 WS + 1)^2
#pragma unroll
                for (int i = -WS; i <= WS; i++)
#pragma unroll
                    for (int j = -WS; j <= WS; j++) npos += (dg[i * w + j] > 0) ? 1 : 0;
                const int pn = 2 * npos - NW2;
                const int apn = pn < 0 ? -pn : pn;
                keep = !((double)apn > a.pn_thresh);
            }
            const unsigned long long bal = __ballot(keep);   // every lane has read its entry before any lane writes
            if (keep) s_list[nkeep + __popcll(bal & ((1ull << lane) - 1ull))] = qv;
            nkeep += __popcll(bal);
        }
        nlist = nkeep;
    }
    int count = 0;  // wave-uniform running count of the strip
    for (int base = 0; base < ((a.ablate & 4) ? 0 : nlist); base += 64) {
        const int li = base + lane;
        bool cand = false;
        float mx = 0.f, my = 0.f, xs = 0.f, ys = 0.f;
        int pix = 0;
        if (li < nlist) {
            const int q = s_list[li];
            pix = y0 * w + q;                                // = (y0 + r) * w + x
            const float *dg = s_dog + WS * w + q;
            double t0 = 0, t1 = 0, t2 = 0;
            if constexpr (WS == 2) {
#pragma unroll
                for (int i = -WS; i <= WS; i++) {
#pragma unroll
                    for (int j = -WS; j <= WS; j++) {
                        const double yv = (double)dg[i * w + j];
                        t0 += pc0[j + 2] * yv;         // TooN dot product: result += a[i]*b[i], k = 0..24 in order
                        t1 += pc1[i + 2] * yv;
                        t2 += pc2 * yv;
                    }
                }
            } else {                                   // theta = PInv * Y (edge_finder.cpp:144), row by row in the same order; window rows
#pragma nounroll                                       // as a loop: 49 values x 3 rows unrolled is more than the register file holds
                for (int i = -WS; i <= WS; i++) {
                    const int k0 = (i + WS) * WIN;
#pragma unroll
                    for (int j = -WS; j <= WS; j++) {
                        const double yv = (double)dg[i * w + j];
                        t0 += s_pinv[k0 + j + WS] * yv;
                        t1 += s_pinv[NW2 + k0 + j + WS] * yv;
                        t2 += s_pinv[2 * NW2 + k0 + j + WS] * yv;
                    }
                }
            }
            {
                const double den = t0 * t0 + t1 * t1;
                xs = (float)(-t0 * t2 / den);
                ys = (float)(-t1 * t2 / den);
                if (!(fabsf(xs) > 0.5f || fabsf(ys) > 0.5f)) {
                    mx = (float)t0;
                    my = (float)t1;
                    const float n2m = mx * mx + my * my;
                    if (!(n2m < thr_d)) cand = true;
                }
            }
        }
        const unsigned long long bal = __ballot(cand);
        if (cand) {
            const int rank = count + __popcll(bal & ((1ull << lane) - 1ull));
            a.st.p_inx[sbase + rank] = pix;
            a.st.m[sbase + rank] = make_float2(mx, my);
            a.st.s[sbase + rank] = make_float2(xs, ys);
        }
        count += __popcll(bal);
    }
    if (lane == 0) a.strip_cnt[(size_t)seq * nstrips + strip] = count;
}

// ---------------------------------------------------------------------------------------------------
// k_strip_scan: one block per sequence.  Exclusive scan of the strip counts = raster-order KeyLine ids;
// kn = min(total, kl_max) (edge_finder.cpp:203-209); stores the P-controller state (detect(), :355-364).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_strip_scan(const int32_t *__restrict__ strip_cnt, int32_t *__restrict__ strip_off,
                                                    SeqA *seqs, int32_t *__restrict__ kn_out, double *__restrict__ tresh_out,
                                                    int nstrips, int kl_max,
                                                    int kl_ref, double gain, double tmax, double tmin) {
    __shared__ int s_part[256];
    const int seq = blockIdx.x, tid = threadIdx.x;
    const int32_t *cnt = strip_cnt + (size_t)seq * nstrips;
    int32_t *off = strip_off + (size_t)seq * (nstrips + 1);
    const int per = (nstrips + 255) / 256;
    int loc = 0;
    for (int i = 0; i < per; i++) {
        const int s = tid * per + i;
        if (s < nstrips) loc += cnt[s];
    }
    s_part[tid] = loc;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    int run = s_part[tid] - loc;
    for (int i = 0; i < per; i++) {
        const int s = tid * per + i;
        if (s < nstrips) {
            off[s] = run;
            run += cnt[s];
        }
    }
    if (tid == 255) {
        const int total = s_part[255];
        off[nstrips] = total;
        SeqA *sq = seqs + seq;
        const int kn = total < kl_max ? total : kl_max;
        const double t = update_thresh(sq->tresh, sq->l_kl_num, kl_ref, gain, tmax, tmin);
        sq->tresh = t;
        sq->tresh_used = t;
        tresh_out[seq] = t;
        sq->l_kl_num = kn;
        sq->kn_new = kn;
        kn_out[seq] = kn;
        sq->nm_max = 0.f;                       // n_m > 0: integer atomics on the float bits order correctly
        sq->nm_min = __int_as_float(0x7f800000);
    }
}


// ---------------------------------------------------------------------------------------------------
// k_emit: thread i < kn builds KeyLine i (edge_finder.cpp:166-200) from its staged candidate and writes
// the mask id.  Also reduces max/min of n_m for reEstimateThresh (:376-382) and clears the histogram.
// ---------------------------------------------------------------------------------------------------
struct EmitArgs {
    CandStage st;
    const int32_t *strip_off;  // [B][nstrips+1]
    KlSoA *kl;                 // [B] of the slot
    int32_t *mask;             // [B][N]
    SeqA *seq;
    int32_t *histo;            // [B][256]
    int nstrips, strip_cap, w;
    size_t n;
    float ppx, ppy;
};

__global__ __launch_bounds__(256) void k_emit(EmitArgs a) {
    const int seq = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    SeqA *sq = a.seq + seq;
    const int kn = sq->kn_new;
    if (blockIdx.x == 0) a.histo[(size_t)seq * 256 + threadIdx.x] = 0;
    float nm = 0.f;
    const bool valid = i < kn;
    if (valid) {
        const int32_t *off = a.strip_off + (size_t)seq * (a.nstrips + 1);
        int lo = 0, hi = a.nstrips;  // largest s with off[s] <= i (empty strips share an offset: skip them)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= i) lo = mid; else hi = mid;
        }
        const size_t src = ((size_t)seq * a.nstrips + lo) * a.strip_cap + (i - off[lo]);
        const int p = a.st.p_inx[src];
        const float2 m = a.st.m[src];
        const float2 s = a.st.s[src];
        const KlSoA &k = a.kl[seq];
        const int y = p / a.w, x = p - y * a.w;
        const float n2m = m.x * m.x + m.y * m.y;             // util::norm2(mn.x,mn.y)
        nm = sqrtf(n2m);                                     // kl.n_m = sqrt(n2_m)
        const float2 u = make_float2(m.x / nm, m.y / nm);
        const float2 cp = make_float2((float)x + s.x, (float)y + s.y);    // {x+xs, y+ys}
        const float2 pm = make_float2(cp.x - a.ppx, cp.y - a.ppy);        // cam_model::Img2Hom
        k.p_inx[i] = p;
        k.m_m[i] = m;
        k.n_m[i] = nm;
        k.u_m[i] = u;
        k.c_p[i] = cp;
        k.p_m[i] = pm;
        k.p_m_0[i] = pm;
        k.rho[i] = 1.0;          // RhoInit
        k.s_rho[i] = 20.0;       // RHO_MAX
        k.rho0[i] = 1.0;
        k.s_rho0[i] = 20.0;
        k.rho_nr[i] = 1.0;
        k.s_rho_nr[i] = 20.0;
        k.m_num[i] = 0;
        k.n_id[i] = -1;
        k.p_id[i] = -1;
        k.m_id[i] = -1;
        if (k.stereo_m_id) { k.stereo_m_id[i] = -1; k.stereo_rho[i] = 1.0; k.stereo_s_rho[i] = 20.0; }   // edge_finder.cpp:192-194
        k.m_id_f[i] = -1;
        k.m_id_kf[i] = -1;
        k.m_m0[i] = make_float2(0.f, 0.f);
        k.n_m0[i] = 0.0;
        MatchRec rec;
        rec.c_px = cp.x; rec.c_py = cp.y; rec.u_mx = u.x; rec.u_my = u.y;
        rec.m_mx = m.x; rec.m_my = m.y; rec.n_m = nm; rec.pad = 0.f;
        k.rec[i] = rec;
        k.grec[i] = make_float4(cp.x, cp.y, m.x, m.y);
        a.mask[(size_t)seq * a.n + p] = i;
    }
    __shared__ float s_mx[4], s_mn[4];
    float mx = valid ? nm : 0.f, mn = valid ? nm : __int_as_float(0x7f800000);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_mn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x * 256 < kn) {
        mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        atomicMax(reinterpret_cast<int *>(&sq->nm_max), __float_as_int(mx));
        atomicMin(reinterpret_cast<int *>(&sq->nm_min), __float_as_int(mn));
    }
}

// ---------------------------------------------------------------------------------------------------
// k_join_histo: join_edges + NextPoint (edge_finder.cpp:221-320) and the histogram of reEstimateThresh
// (:384-398).  kl[ikl2].p_id = ikl is last-writer-wins in KeyLine order  ==  atomicMax.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int x86_cvttss2si(float f) {
    // (int)float as x86-64 does it: NaN / out of range -> 0x80000000
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return (int)0x80000000;
    return (int)f;
}

// DEFAULTS: also give KeyLine i the constant fields of a freshly detected KeyLine (edge_finder.cpp:176-196) — after the fused
// stage-A kernel, which writes only the computed fields and p_id (the atomicMax below needs p_id = -1 before any thread
// runs): a full-wave streaming store here instead of a partial-wave one there.
// DERIVE (with DEFAULTS): the fused kernel's fit wave left only what the plane fit produced — p_inx and {xs, ys, m_m} in
// the grec slot — and every field that follows from them (edge_finder.cpp:166-200: n_m, u_m, c_p, p_m, p_m_0, the gather
// records) is computed here, at full lanes and streaming, instead of by one wave of the detector at a third of its lanes.
// fwd_fills (with DEFAULTS): the whole-frame driver's FordwardMatch will write the ten fields it forwards for EVERY new KeyLine
// (k_fwd_apply's fill mode: the forwarded values or these defaults), so they are not written here — 68 of the 180 bytes.
#ifndef EDGEHIP_JOIN_NT
#define EDGEHIP_JOIN_NT 1
#endif
template <bool DEFAULTS, bool DERIVE = false>
__global__ __launch_bounds__(256) void k_join_histo(KlSoA *kls, const int32_t *__restrict__ masks, SeqA *seqs,
                                                    int32_t *histo, int w, size_t n, int nbins, float ppx = 0.f, float ppy = 0.f,
                                                    int fwd_fills = 0) {
    const int seq = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    SeqA *sq = seqs + seq;
    const int kn = sq->kn_new;
    __shared__ int s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    if (i < kn) {
        const KlSoA &k = kls[seq];
        const int32_t *mask = masks + (size_t)seq * n;
        float2 cp, m, u_d, pm_d;
        float nm_i;
        if (DERIVE) {
            const float4 raw = k.grec[i];           // xs, ys, m_m.x, m_m.y
            const int p = k.p_inx[i];
            const int py = p / w, px = p - py * w;
            m = make_float2(raw.z, raw.w);
            const float n2m = m.x * m.x + m.y * m.y;
            nm_i = sqrtf(n2m);
            u_d = make_float2(m.x / nm_i, m.y / nm_i);
            cp = make_float2((float)px + raw.x, (float)py + raw.y);
            pm_d = make_float2(cp.x - ppx, cp.y - ppy);      // cam_model::Img2Hom
            // (stored below, after the mask probes: a load issued behind stores waits for them — vmcnt counts both)
        } else {
            cp = k.c_p[i];
            m = k.m_m[i];
            nm_i = k.n_m[i];
        }
        const int x = (int)((double)cp.x + 0.5);  // util::round2int_positive: float + 0.5 (double)
        const int y = (int)((double)cp.y + 0.5);
        const float tx = -m.y, ty = m.x;
        int sx, sy;
        if (ty > 0) { sy = 1; sx = (tx > 0) ? 1 : -1; }
        else        { sy = -1; sx = (tx < 0) ? -1 : 1; }
        int j = mask[(size_t)y * w + (x + sx)];
        if (j < 0) j = mask[(size_t)(y + sy) * w + x];
        if (j < 0) j = mask[(size_t)(y + sy) * w + (x + sx)];
        if (DERIVE) {
#if EDGEHIP_JOIN_NT   // streaming stores for the 1.7 GB this kernel writes per 1024 frames: 507 -> 465 us (same box, profiles/r06_nontemporal_stores_ab.txt)
            typedef float f2v __attribute__((ext_vector_type(2)));
            typedef float f4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f2v{m.x, m.y}, reinterpret_cast<f2v *>(k.m_m + i));
            __builtin_nontemporal_store(nm_i, k.n_m + i);
            __builtin_nontemporal_store(f2v{u_d.x, u_d.y}, reinterpret_cast<f2v *>(k.u_m + i));
            __builtin_nontemporal_store(f2v{cp.x, cp.y}, reinterpret_cast<f2v *>(k.c_p + i));
            __builtin_nontemporal_store(f2v{pm_d.x, pm_d.y}, reinterpret_cast<f2v *>(k.p_m + i));
            if (!fwd_fills) k.p_m_0[i] = pm_d;
            f4v *rp = reinterpret_cast<f4v *>(k.rec + i);
            __builtin_nontemporal_store(f4v{cp.x, cp.y, u_d.x, u_d.y}, rp);
            __builtin_nontemporal_store(f4v{m.x, m.y, nm_i, 0.f}, rp + 1);
            __builtin_nontemporal_store(f4v{cp.x, cp.y, m.x, m.y}, reinterpret_cast<f4v *>(k.grec + i));
#else
            k.m_m[i] = m;
            k.n_m[i] = nm_i;
            k.u_m[i] = u_d;
            k.c_p[i] = cp;
            k.p_m[i] = pm_d;
            if (!fwd_fills) k.p_m_0[i] = pm_d;
            MatchRec rec;
            rec.c_px = cp.x; rec.c_py = cp.y; rec.u_mx = u_d.x; rec.u_my = u_d.y;
            rec.m_mx = m.x; rec.m_my = m.y; rec.n_m = nm_i; rec.pad = 0.f;
            k.rec[i] = rec;
            k.grec[i] = make_float4(cp.x, cp.y, m.x, m.y);
#endif
        }
        if (DEFAULTS) {
            if (!fwd_fills) {
                k.rho[i] = 1.0;          // RhoInit
                k.s_rho[i] = 20.0;       // RHO_MAX
                k.rho_nr[i] = 1.0;
                k.s_rho_nr[i] = 20.0;
                k.m_num[i] = 0;
                k.m_id[i] = -1;
                k.m_id_kf[i] = -1;
                k.m_m0[i] = make_float2(0.f, 0.f);
                k.n_m0[i] = 0.0;
            }
#if EDGEHIP_JOIN_NT
            __builtin_nontemporal_store(1.0, k.rho0 + i);
            __builtin_nontemporal_store(20.0, k.s_rho0 + i);
            if (k.stereo_m_id) { k.stereo_m_id[i] = -1; k.stereo_rho[i] = 1.0; k.stereo_s_rho[i] = 20.0; }
            __builtin_nontemporal_store(-1, k.m_id_f + i);
            __builtin_nontemporal_store(j, k.n_id + i);
#else
            k.rho0[i] = 1.0;
            k.s_rho0[i] = 20.0;
            if (k.stereo_m_id) { k.stereo_m_id[i] = -1; k.stereo_rho[i] = 1.0; k.stereo_s_rho[i] = 20.0; }
            k.m_id_f[i] = -1;
            k.n_id[i] = j;           // -1 without a neighbour
#endif
        } else if (j >= 0) k.n_id[i] = j;
        if (j >= 0) atomicMax(&k.p_id[j], i);
        // histogram position, edge_finder.cpp:392
        const float mxd = sq->nm_max, mnd = sq->nm_min;
        int b = x86_cvttss2si((float)nbins * (mxd - nm_i) / (mxd - mnd));
        b = b > nbins - 1 ? nbins - 1 : b;
        b = b < 0 ? 0 : b;
        atomicAdd(&s_h[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < nbins && s_h[threadIdx.x] != 0 && blockIdx.x * 256 < kn)
        atomicAdd(&histo[(size_t)seq * 256 + threadIdx.x], s_h[threadIdx.x]);
}

// k_retune: tail of reEstimateThresh (edge_finder.cpp:400-403).  The reference loop
//     for(int a=0; i<n && a<knum; i++, a+=histo[i]);
// accumulates histo[i] AFTER incrementing i, i.e. bin 0 is never counted (and histo[n] is read past the
// end on the last step, where it no longer matters).
__global__ __launch_bounds__(256) void k_retune(SeqA *seqs, const int32_t *__restrict__ histo, float *__restrict__ retuned_out,
                                                int nseq, int knum, int nbins) {
    const int seq = blockIdx.x;
    __shared__ int s_h[256];
    s_h[threadIdx.x] = histo[(size_t)seq * 256 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    SeqA *sq = seqs + seq;
    int i = 0;
    for (int acc = 0; i < nbins && acc < knum;) {
        i++;
        if (i < nbins) acc += s_h[i];
    }
    const float mxd = sq->nm_max, mnd = sq->nm_min;
    float r = mxd - (float)i * (mxd - mnd) / (float)nbins;
    if (sq->kn_new <= 0) r = 0.f;
    sq->retuned = r;
    retuned_out[seq] = r;
}

// image_undistort::undistort<true> (image_undistort.h:105-122) fused with ConvertRGB2BW (image.h:197-203): b+g+r of the
// resampled pixel as a 16-bit plane — the input of the fused stage-A kernel when UseUndistort is set (the four bilinear taps
// are data-dependent gathers, which that kernel's thread <-> column-pair layout cannot prefetch; here they are plain loads
// of a streaming kernel: 36 N map bytes + the frame in, 2 N out).
// SG sequences per thread: the map entry of a pixel (20 bytes, the same for every sequence of the batch) is read once for all of
// them — one sequence per thread had the kernel bound by 20 B of map per 6-byte pixel out of L2.  All 2 * SG tap loads in flight.
template <int SG>
__global__ __launch_bounds__(256) void k_undistort_grey(const uint8_t *__restrict__ rgb, const int32_t *__restrict__ fidx,
                                                        uint16_t *__restrict__ out, int w, int n, const int32_t *__restrict__ und_base,
                                                        const uint4 *__restrict__ und_iw, int nseq) {
    const int seq0 = blockIdx.z * SG;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= n) return;
    const int32_t base = und_base[pix];
    const uint4 iw = und_iw[pix];
    uint64_t t[SG], u[SG];
#pragma unroll
    for (int s = 0; s < SG; s++) {
        const int seq = min(seq0 + s, nseq - 1);
        const uint8_t *frame = rgb + (size_t)(fidx ? fidx[seq] : seq) * (size_t)n * 3;
        t[s] = undist_row6(frame, base, n);
        u[s] = undist_row6(frame, base + w, n);
    }
#pragma unroll
    for (int s = 0; s < SG; s++) {
        if (seq0 + s >= nseq) break;
        const uchar3 c = undist_mix(t[s], u[s], iw);
        out[(size_t)(seq0 + s) * n + pix] = (uint16_t)((int)c.x + (int)c.y + (int)c.z);
    }
}

__global__ void k_undistort_frame(const uint8_t *__restrict__ frame, uint8_t *__restrict__ out, int w, int n,
                                  const int32_t *__restrict__ und_base, const uint4 *__restrict__ und_iw) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= n) return;
    const uchar3 c = undist_rgb(frame, und_base[pix], und_iw[pix], w, (int)n);
    out[(size_t)pix * 3] = c.x; out[(size_t)pix * 3 + 1] = c.y; out[(size_t)pix * 3 + 2] = c.z;
}

// 8-bit mono frames -> the slot's RGB24 storage (r = g = b = v, what DataSetCam makes of a mono image, datasetcam.cpp:
// 109-171) for the paths that read RGB24: the multi-kernel stage A of small batches, the undistorting load, the stereo
// pair slot.  N bytes in, 3 N out; the one-kernel stage A of large batches reads the 8-bit frames directly (SRC_GREY8).
__global__ __launch_bounds__(256) void k_expand_grey8(const uint8_t *__restrict__ g8, const int32_t *__restrict__ fidx,
                                                      uint8_t *__restrict__ rgb, int n) {
    const int seq = blockIdx.z;
    const uint8_t *src = g8 + (size_t)(fidx ? fidx[seq] : seq) * (size_t)n;
    uint8_t *dst = rgb + (size_t)seq * (size_t)n * 3;
    const int q = blockIdx.x * 256 + threadIdx.x;          // four pixels per thread: 4 B in, 12 B out
    if (q * 4 >= n) return;
    if ((n & 3) != 0) {     // frames of n % 4 != 0 pixels start off a dword boundary and end inside one: byte by byte
        for (int i = q * 4; i < min(n, q * 4 + 4); i++) {
            const uint8_t g = src[i];
            dst[(size_t)i * 3] = g; dst[(size_t)i * 3 + 1] = g; dst[(size_t)i * 3 + 2] = g;
        }
        return;
    }
    const uint32_t v = *reinterpret_cast<const uint32_t *>(src + (size_t)q * 4);
    const uint32_t a = v & 0xFFu, b = (v >> 8) & 0xFFu, c2 = (v >> 16) & 0xFFu, d = v >> 24;
    uint3 o;
    o.x = a | (a << 8) | (a << 16) | (b << 24);
    o.y = b | (b << 8) | (c2 << 16) | (c2 << 24);
    o.z = c2 | (d << 8) | (d << 16) | (d << 24);
    *reinterpret_cast<uint3 *>(dst + (size_t)q * 12) = o;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// out_dev: 3 N bytes at the start of the stage-A scratch (c->ii, 16 N bytes per sequence); a mono slot's frame is expanded to
// RGB24 behind it first (r = g = b = v, what the undistorting load of stage A sees of it)
int undistort_frame_enqueue(edgehip_ctx *c, int seq, int slot, uint8_t *out_dev) {
    const DevicePlan &pl = c->plan;
    const edgehip_ctx::SlotSrc &ss = c->slot_src[slot];
    const uint8_t *src;
    if (ss.grey8) {
        const uint8_t *g8 = ss.base ? ss.base + (size_t)ss.host_idx[seq] * pl.n : c->grey8 + ((size_t)slot * pl.nseq + seq) * pl.n;
        uint8_t *rgb = out_dev + (((size_t)pl.n * 3 + 255) & ~(size_t)255);
        hipLaunchKernelGGL(k_expand_grey8, dim3((unsigned)((pl.n / 4 + 255) / 256), 1, 1), dim3(256), 0, c->stream_a, g8, (const int32_t *)nullptr, rgb, pl.n);
        EH_LAUNCH_CHECK();
        src = rgb;
    } else {
        src = ss.base ? ss.base + (size_t)ss.host_idx[seq] * pl.n * 3 : rgbof(c, slot) + (size_t)seq * pl.n * 3;
    }
    hipLaunchKernelGGL(k_undistort_frame, dim3((pl.n + 255) / 256), dim3(256), 0, c->stream_a, src, out_dev, pl.w, pl.n, c->und_base, c->und_iw);
    EH_LAUNCH_CHECK();
    return 0;
}

static int rowscan_ch(int w) { return 4 * ((w + 255) / 256); }

int stage_a_enqueue(edgehip_ctx *c, int slot, bool fwd_fills, bool defer_retune) {
    c->fwd_fill[slot] = false;
    const DevicePlan &pl = c->plan;
    const int w = pl.w, h = pl.h, B = pl.nseq;
    const size_t n = pl.n;
    float *ii[4];
    for (int i = 0; i < 4; i++) ii[i] = c->ii + (size_t)i * B * n;
    hipStream_t st = c->stream_a;
    // where the slot's frames are: its own storage (sequence-major) or frames of a bound device pool (edgehip_bind_rgb_indexed)
    const uint8_t *rgb_base = c->slot_src[slot].base ? c->slot_src[slot].base : rgbof(c, slot);
    const int32_t *rgb_idx = c->slot_src[slot].base ? c->slot_src[slot].idx_row : nullptr;

    // Batches that fill the GPU with one workgroup per sequence: the whole of stage A up to the KeyLine records in one
    // kernel (stage_a_fused.hip); level_mode 3 forces it, 1 / 2 keep the multi-kernel path (A/B measurements, tests).
    const int fused_min = c->fused_min_batch > 0 ? c->fused_min_batch
                                                 : fused_min_batch_for(c, c->und_base != nullptr && !c->fused_undist, c->slot_src[slot].grey8 && !c->und_base);
    const bool use_fused = fused_supported(c) && (c->level_mode == 3 || (c->level_mode == 0 && B >= fused_min));
    // 8-bit mono frames: the one-kernel stage A reads them as they are; every other path gets the RGB24 expansion first
    const uint8_t *grey8 = nullptr;
    if (c->slot_src[slot].grey8) {
        const uint8_t *g8 = c->slot_src[slot].base ? c->slot_src[slot].base : c->grey8 + (size_t)slot * B * n;
        if (use_fused && !c->und_base) {
            grey8 = g8;
        } else {
            ProfScope ps(c, PROF_A_ROWSCAN, st);
            hipLaunchKernelGGL(k_expand_grey8, dim3((unsigned)((n / 4 + 255) / 256), 1, B), dim3(256), 0, st, g8, rgb_idx, rgbof(c, slot), (int)n);
            EH_LAUNCH_CHECK();
            rgb_base = rgbof(c, slot);
            rgb_idx = nullptr;
        }
    }
    if (use_fused) {
        const uint16_t *grey16 = nullptr;
        const bool und_in_load = c->und_base && c->fused_undist;
        if (c->und_base && !und_in_load) {   // UseUndistort: resample + grey first (the integral-image scratch is free on this path)
            ProfScope ps(c, PROF_A_ROWSCAN, st);
            uint16_t *g16 = reinterpret_cast<uint16_t *>(c->ii);
            hipLaunchKernelGGL(k_undistort_grey<8>, dim3((unsigned)((n + 255) / 256), 1, (B + 7) / 8), dim3(256), 0, st, rgb_base, rgb_idx, g16, w,
                               (int)n, c->und_base, c->und_iw, B);
            EH_LAUNCH_CHECK();
            grey16 = g16;
        }
        if (int e = stage_a_fused_enqueue(c, slot, rgb_base, rgb_idx, grey16, grey8, und_in_load)) return e;
        ProfScope ps(c, PROF_A_JOIN, st);
        hipLaunchKernelGGL((k_join_histo<true, true>), dim3((pl.cap + 255) / 256, 1, B), dim3(256), 0, st, kldev(c, slot),
                           maskof(c, slot), c->seqa, c->histo, w, n, c->p.qcut_nbins, c->slot_cam[slot].ppx, c->slot_cam[slot].ppy,
                           fwd_fills ? 1 : 0);
        c->fwd_fill[slot] = fwd_fills;   // FordwardMatch into this slot must run in fill mode
        EH_LAUNCH_CHECK();
        if (!defer_retune)
            hipLaunchKernelGGL(k_retune, dim3(B), dim3(256), 0, st, c->seqa, c->histo,
                               c->retuned_slot + (size_t)slot * B, B, c->p.track_points,
                               c->p.qcut_nbins);
        EH_LAUNCH_CHECK();
        return 0;
    }

    float *cur[2] = {ii[0], ii[0]};
    const int planes_in_flight = B * 2;
    // (k_level's row scan walks float4 steps: widths that are not a multiple of 4 take the multi-pass kernels)
    const bool use_level = (w & 3) == 0 && w <= LV_NC * LV_MAXCOL && (c->level_mode == 2 || (c->level_mode == 0 && planes_in_flight >= 192));
    if (use_level) {
        // one pass per level and plane (k_level): grey -> integral #1, then the box levels
        const size_t sm = ((size_t)2 * LV_RB * level_row_stride(w) + 32 + kDivLutMax) * sizeof(float);   // + scan tail pad + LUT copy
#ifdef EDGEHIP_EXPERIMENTS   // make EXPERIMENTS=1: phase ablation for timing experiments (wrong results by design)
        const int lv_ablate = getenv("EDGEHIP_LEVEL_ABLATE") ? atoi(getenv("EDGEHIP_LEVEL_ABLATE")) : 0;
#else
        const int lv_ablate = 0;
#endif
        if (sm > 64 * 1024) {   // more than the default dynamic-LDS limit: opt in once per kernel and context (= per device)
            bool &done = c->lds_optin_level;
            if (!done) {
                const void *fns[6] = {(const void *)&k_level<0, 1>, (const void *)&k_level<1, 1>, (const void *)&k_level<2, 1>,
                                      (const void *)&k_level<0, 2>, (const void *)&k_level<1, 2>, (const void *)&k_level<2, 2>};
                for (const void *f : fns) EH_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                done = true;
            }
        }
        {
            ProfScope ps(c, PROF_A_LEVEL, st);
            LevelJob job = {};
            job.dst[0] = ii[0]; job.d[0] = 1; job.a[0] = 1.f;
#define EH_LEVEL(SRCV, GRID)                                                                                                   \
    do {                                                                                                                      \
        if (w <= LV_NC) hipLaunchKernelGGL((k_level<SRCV, 1>), GRID, dim3(LV_NT), sm, st, job, rgb_base, rgb_idx, c->und_base,    \
                                           c->und_iw, c->div_lut, w, h, n, lv_ablate);                                        \
        else hipLaunchKernelGGL((k_level<SRCV, 2>), GRID, dim3(LV_NT), sm, st, job, rgb_base, rgb_idx, c->und_base, c->und_iw,   \
                                c->div_lut, w, h, n, lv_ablate);                                                              \
    } while (0)
            if (c->und_base) EH_LEVEL(2, dim3(1, 1, B));
            else EH_LEVEL(1, dim3(1, 1, B));
            EH_LAUNCH_CHECK();
        }
        int next_free = 1;
        bool shared = true;
        for (int lvl = 0; lvl < kMaxBoxes - 1; lvl++) {
            const int d0 = pl.box[0][lvl], d1 = pl.box[1][lvl];
            LevelJob job = {};
            int njobs;
            if (shared && d0 == d1) {
                float *dst = ii[next_free++];
                job.src[0] = cur[0]; job.dst[0] = dst; job.d[0] = d0; job.a[0] = pl.box_a[0][lvl];
                njobs = 1;
                cur[0] = cur[1] = dst;
            } else {
                shared = false;
                float *dst0 = nullptr, *dst1 = nullptr;
                for (int i = 0; i < 4 && !dst1; i++) {
                    if (ii[i] == cur[0] || ii[i] == cur[1]) continue;
                    if (!dst0) dst0 = ii[i]; else dst1 = ii[i];
                }
                job.src[0] = cur[0]; job.dst[0] = dst0; job.d[0] = d0; job.a[0] = pl.box_a[0][lvl];
                job.src[1] = cur[1]; job.dst[1] = dst1; job.d[1] = d1; job.a[1] = pl.box_a[1][lvl];
                njobs = 2;
                cur[0] = dst0;
                cur[1] = dst1;
            }
            ProfScope ps(c, PROF_A_LEVEL, st);
            EH_LEVEL(0, dim3(njobs, 1, B));
            EH_LAUNCH_CHECK();
        }
    } else {
        // 1. grey + exact row prefix, then serial column prefix: iimage::load of the input (shared by both
        //    filters: filter0.smooth(data) and filter1.smooth(data) start from the same integral image).
        {
            ProfScope ps(c, PROF_A_ROWSCAN, st);
            const int ch = rowscan_ch(w);
            const size_t sm = (size_t)4 * ((w * 3 + 3) / 4) * 4;
            dim3 g((h + 3) / 4, 1, B);
            const bool und = c->und_base != nullptr;
    #define EH_ROWSCAN(CHV)                                                                                              \
        case CHV:                                                                                                        \
            if (und) hipLaunchKernelGGL((k_rgb_rowscan<CHV, true>), g, dim3(256), sm, st, rgb_base, rgb_idx, ii[0], w, h, n, \
                                        c->und_base, c->und_iw);                                                         \
            else hipLaunchKernelGGL((k_rgb_rowscan<CHV, false>), g, dim3(256), sm, st, rgb_base, rgb_idx, ii[0], w, h, n,   \
                                    c->und_base, c->und_iw);                                                             \
            break;
            switch (ch) {
                EH_ROWSCAN(4) EH_ROWSCAN(8) EH_ROWSCAN(12) EH_ROWSCAN(16) EH_ROWSCAN(20) EH_ROWSCAN(24) EH_ROWSCAN(28)
                default:
                    if (und) hipLaunchKernelGGL((k_rgb_rowscan<32, true>), g, dim3(256), sm, st, rgb_base, rgb_idx, ii[0], w, h, n,
                                                c->und_base, c->und_iw);
                    else hipLaunchKernelGGL((k_rgb_rowscan<32, false>), g, dim3(256), sm, st, rgb_base, rgb_idx, ii[0], w, h, n,
                                            c->und_base, c->und_iw);
                    break;
            }
    #undef EH_ROWSCAN
            EH_LAUNCH_CHECK();
        }
        auto colscan = [&](float *a, float *b) -> int {
            ProfScope ps(c, PROF_A_COLSCAN, st);
            PlanePtrs pp;
            pp.p[0] = a;
            pp.p[1] = b ? b : a;
            const size_t nblk = (size_t)((w + 63) / 64) * (b ? 2 : 1) * B;
            if (nblk <= 512 && h <= 16 * 32)
                hipLaunchKernelGGL((k_colscan_chain<16, 32>), dim3((w + 63) / 64, b ? 2 : 1, B), dim3(1024), 0, st, pp, w, h, n);
            else if (nblk <= 512 && h <= 16 * 64)
                hipLaunchKernelGGL((k_colscan_chain<16, 64>), dim3((w + 63) / 64, b ? 2 : 1, B), dim3(1024), 0, st, pp, w, h, n);
            else
                hipLaunchKernelGGL(k_colscan<32>, dim3((w + 63) / 64, b ? 2 : 1, B), dim3(64), 0, st, pp, w, h, n);
            EH_LAUNCH_CHECK();
            return 0;
        };
        if (int e = colscan(ii[0], nullptr)) return e;

        // 2. the remaining box passes.  Levels that have identical box-width prefixes in both filters are
        //    computed once (EuRoC: {3,3,5} / {3,5,5} share the first level).
        //    cur[f] = integral image that filter f averages next.
        int next_free = 1;
        bool shared = true;
        for (int lvl = 0; lvl < kMaxBoxes - 1; lvl++) {
            const int d0 = pl.box[0][lvl], d1 = pl.box[1][lvl];
            AvgJob job;
            int njobs;
            if (shared && d0 == d1) {
                float *dst = ii[next_free++];
                job.src[0] = cur[0]; job.dst[0] = dst; job.d[0] = d0;
                job.src[1] = cur[0]; job.dst[1] = dst; job.d[1] = d0;
                njobs = 1;
                cur[0] = cur[1] = dst;
            } else {
                shared = false;
                // two destinations; reuse planes that are no longer read
                float *dst0 = nullptr, *dst1 = nullptr;
                for (int i = 0; i < 4 && !dst1; i++) {
                    if (ii[i] == cur[0] || ii[i] == cur[1]) continue;
                    if (!dst0) dst0 = ii[i]; else dst1 = ii[i];
                }
                job.src[0] = cur[0]; job.dst[0] = dst0; job.d[0] = d0;
                job.src[1] = cur[1]; job.dst[1] = dst1; job.d[1] = d1;
                njobs = 2;
                cur[0] = dst0;
                cur[1] = dst1;
            }
            {
                ProfScope ps(c, PROF_A_AVGROW, st);
                // A few sequences (a live camera): 64-row blocks leave most CUs idle and every block pays the tap-load latency
                // of w/64 column tiles one after the other — 16-row blocks over 256-column tiles: 4x the blocks, a quarter
                // of the tile round trips (100 -> ~20 us per launch for one 752x480 frame).  Same operations, same order.
                // ... and for one or two sequences 4-row blocks: 240 blocks for one 752x480 frame instead of 60, a quarter of the
                // averages per tile and block (the row prefix itself is a fixed chain of w dependent adds per row either way).
                if ((size_t)B * njobs * ((h + 15) / 16) < 128 && w <= 768)
                    hipLaunchKernelGGL((k_avg_rowscan_wide<256, 4, 768>), dim3((h + 3) / 4, njobs, B), dim3(256), 0, st, job,
                                       c->div_lut, w, h, n);
                else if ((size_t)B * njobs * ((h + 15) / 16) < 128)
                    hipLaunchKernelGGL((k_avg_rowscan<256, 4, 256>), dim3((h + 3) / 4, njobs, B), dim3(256), 0, st, job,
                                       c->div_lut, w, h, n);
                else if ((size_t)B * njobs * ((h + 63) / 64) < 256)
                    hipLaunchKernelGGL((k_avg_rowscan<256, 16, 256>), dim3((h + 15) / 16, njobs, B), dim3(256), 0, st, job,
                                       c->div_lut, w, h, n);
                else
                    hipLaunchKernelGGL((k_avg_rowscan<256, 64, 64>), dim3((h + 63) / 64, njobs, B), dim3(256), 0, st, job,
                                       c->div_lut, w, h, n);
                EH_LAUNCH_CHECK();
            }
            if (int e = colscan(cur[0], njobs == 2 ? cur[1] : nullptr)) return e;
        }

    }

    // 3. last average + DoG + gradient + detection
    const int nbands = c->nbands, nstrips = nbands * kDetWaves;
    CandStage cs;
    {
        char *base = (char *)c->band_stage;
        const size_t cnt = (size_t)B * nstrips * c->band_cap;
        cs.p_inx = (int32_t *)base;
        cs.m = (float2 *)(base + cnt * 4);
        cs.s = (float2 *)(base + cnt * 12);
    }
    {
        ProfScope ps(c, PROF_A_DETECT, st);
        DetectArgs a;
        a.iic0 = cur[0]; a.iic1 = cur[1];
        a.d0 = pl.box[0][kMaxBoxes - 1]; a.d1 = pl.box[1][kMaxBoxes - 1];
        a.a0 = pl.box_a[0][kMaxBoxes - 1]; a.a1 = pl.box_a[1][kMaxBoxes - 1];
        a.lut = c->div_lut; a.pinv = c->pinv;
        a.planes = c->planes;
        a.mask = maskof(c, slot);
        a.seq = c->seqa;
        a.st = cs;
        a.strip_cnt = c->band_cnt;
        a.w = w; a.h = h; a.nseq = B; a.strip_cap = c->band_cap; a.n = n;
        a.gain = c->p.auto_gain; a.tmax = c->p.max_thresh; a.tmin = c->p.min_thresh;
        a.kl_ref = c->p.reference_points;
        a.dog_thresh_f = (float)c->p.dog_thresh;
        const int ws = c->p.plane_fit_size;
        a.pn_thresh = (double)(((float)((2.0 * ws + 1.0) * (2.0 * ws + 1.0))) * (float)c->p.pos_neg_thresh);
#ifdef EDGEHIP_EXPERIMENTS
        a.ablate = getenv("EDGEHIP_ABLATE") ? atoi(getenv("EDGEHIP_ABLATE")) : 0;
#else
        a.ablate = 0;
#endif
        a.band_rows = c->band_rows;
        const size_t sm = detect_lds_bytes(w, c->band_rows, ws);
        void (*fn)(DetectArgs) = ws == 1 ? k_detect<1> : ws == 3 ? k_detect<3> : k_detect<2>;
        if (sm > 64 * 1024) {
            bool &done = c->lds_optin_detect;
            if (!done) {
                // (the limit is 160 KB minus the kernel's static LDS: the pseudo inverse and the reciprocal table, 1.2-2.2 KB)
                hipFuncAttributes fa;
                EH_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(fn)));
                const int dyn_max = 160 * 1024 - (int)fa.sharedSizeBytes;   // the same for every context of the process: the attribute belongs to the kernel
                const hipError_t ae = (size_t)dyn_max < sm ? hipErrorInvalidValue
                                                           : hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max);
                if (ae != hipSuccess) {
                    (void)hipGetLastError();
                    set_error("stage A: the detector's LDS planes (" + std::to_string(sm) + " B for a " + std::to_string(w) + "-column band of " +
                              std::to_string(c->band_rows) + " rows, window " + std::to_string(ws) + ") were refused: " + hipGetErrorString(ae));
                    return EDGEHIP_ERR_DEVICE;
                }
                done = true;
            }
        }
        hipLaunchKernelGGL(fn, dim3(nbands, 1, B), dim3(kDetWaves * 64), sm, st, a);
        EH_LAUNCH_CHECK();
    }
    {
        ProfScope ps(c, PROF_A_COMPACT, st);
        int kl_max = c->p.max_points;
        if (kl_max > pl.cap) kl_max = pl.cap;
        hipLaunchKernelGGL(k_strip_scan, dim3(B), dim3(256), 0, st, c->band_cnt, c->band_off, c->seqa,
                           c->kn_slot + (size_t)slot * B, c->tresh_slot + (size_t)slot * B, nstrips, kl_max,
                           c->p.reference_points, c->p.auto_gain, c->p.max_thresh, c->p.min_thresh);
        EH_LAUNCH_CHECK();
        EmitArgs e;
        e.st = cs; e.strip_off = c->band_off; e.kl = kldev(c, slot); e.mask = maskof(c, slot); e.seq = c->seqa;
        e.histo = c->histo; e.nstrips = nstrips; e.strip_cap = c->band_cap; e.w = w; e.n = n;
        e.ppx = c->slot_cam[slot].ppx; e.ppy = c->slot_cam[slot].ppy;   // the slot's camera (a stereo pair slot may differ)
        hipLaunchKernelGGL(k_emit, dim3((pl.cap + 255) / 256, 1, B), dim3(256), 0, st, e);
        EH_LAUNCH_CHECK();
        c->grec_ok[slot] = true;   // freshly detected KeyLines: u_m = m_m / |m_m| holds for all of them
        c->rec_stale[slot] = false;
        c->rot_pending[slot] = false;
    }
    {
        ProfScope ps(c, PROF_A_JOIN, st);
        hipLaunchKernelGGL(k_join_histo<false>, dim3((pl.cap + 255) / 256, 1, B), dim3(256), 0, st, kldev(c, slot),
                           maskof(c, slot), c->seqa, c->histo, w, n, c->p.qcut_nbins);
        EH_LAUNCH_CHECK();
        if (!defer_retune)
            hipLaunchKernelGGL(k_retune, dim3(B), dim3(256), 0, st, c->seqa, c->histo,
                               c->retuned_slot + (size_t)slot * B, B, c->p.track_points,
                               c->p.qcut_nbins);
        EH_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace edgehip
