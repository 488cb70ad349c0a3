// stage_a_dev.h — device helpers and launch records shared by stage_a.hip and stage_a_fused.hip.
#pragma once
#include "ctx.h"

namespace edgehip {

// UpdateThresh, edge_finder.cpp:330-335
__device__ __forceinline__ double update_thresh(double tresh, int l_kl_num, int kl_ref, double gain, double tmax,
                                                double tmin) {
    if (gain > 0) {
        tresh -= gain * (double)(kl_ref - l_kl_num);
        tresh = tresh > tmax ? tmax : (tresh < tmin ? tmin : tresh);
    }
    return tresh;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every
// outstanding global load AND store of the wave; the stage-A kernels exchange data between threads through LDS alone,
// and letting the global prefetches / row stores stay in flight across the barrier is the whole point.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS rows of the fused kernel: kFusedPad zero floats left of column 0 (the taps x-r-1 < 0 of iimage::average read 0),
// row stride with (stride / 4) odd so that the scan wave's lanes (one row each, float4 steps) spread over the banks.
constexpr int kFusedPad = 4;   // also the right pad: the scan wave copies the last value of the row there (taps right of column w-1)
__host__ __device__ constexpr inline int fused_row_stride(int w) {
    const int wp = (w + 2 * kFusedPad + 3) & ~3;
    return ((wp >> 2) & 1) == 0 ? wp + 4 : wp;
}

struct FusedArgs {
    const uint8_t *rgb;        // frames of the slot (sequence-major) or a bound pool
    const int32_t *fidx;       // [B] frame index inside the pool, or null
    const uint16_t *grey16;    // [B][N] b+g+r of the undistorted frame (SRC_GREY16 instantiations), or null
    const int32_t *und_base;   // [N] first tap of every pixel / [N] its four 16.16 weights (SRC_UNDIST instantiations), or null
    const uint4 *und_iw;
    const uint8_t *grey8;      // 8-bit mono frames, 1 B per pixel (SRC_GREY8 instantiations: frame = grey8 + (fidx ? fidx[seq] : seq) * n), or null
    const float *lut;          // [kDivLutMax] (float)(1.0/count)
    float *planes;             // optional debug planes [5][B][N]
    int32_t *mask;             // [B][N] of the slot
    SeqA *seq;
    KlSoA *kl;                 // [B] of the slot
    int32_t *histo;            // [B][256]
    int32_t *kn_out;           // [B] edge_finder::kn of the slot
    double *tresh_out;         // [B] threshold the slot was detected with
    int w, h, nseq;
    size_t n;
    double gain, tmax, tmin;   // edge_finder::detect arguments
    int kl_ref, kl_max;
    float dog_thresh_f;        // (float)DetectorDoGThresh
    double pn_thresh;          // (double)(25.0f * (float)DetectorPosNegThresh)
    const double *pinv;           // [3][25] plane-fit pseudo inverse (device): row 0 depends on the window column only, row 1 on the
                                  // window row, row 2 is constant (checked at create) — read with scalar loads where it is used,
                                  // not carried in 22 scalar registers through the whole kernel
    float ppx, ppy;
    int ablate;                // timing experiments (EXPERIMENTS builds only)
};

// LDS of k_detect<WS> for a band of `band_rows` image rows: img0 and DoG of the band + WS halo rows either side, and the
// per-wave candidate lists (one 16-bit entry per band pixel)
inline size_t detect_lds_bytes(int w, int band_rows, int ws) {
    const int npx = band_rows * w, cpw = (((npx + 63) >> 6) + kDetWaves - 1) / kDetWaves;
    return (size_t)2 * (band_rows + 2 * ws) * w * sizeof(float) + (size_t)kDetWaves * cpw * 64 * sizeof(uint16_t);
}
// the tallest band (12, 8, 4 or 2 rows) whose planes fit the LDS the kernel may opt into; 0 if not even two rows do
inline int detect_band_rows(int w, int ws) {
    for (int br : {kBandRows, 8, 4, 2})
        if (detect_lds_bytes(w, br, ws) <= (size_t)156 * 1024) return br;   // + up to 2.2 KB of static LDS (pseudo inverse, reciprocal table) <= 160 KB
    return 0;
}

// image_undistort::biInterp for RGB24 (include/VideoLib/image_undistort.h:66-79): integer 16.16 weights, >>16,
// truncation to 8 bits.  Taps sit at base, base+1, base+w, base+w+1; an invalid tap has weight 0.
// The two pixels of a tap row are six consecutive bytes: one unaligned 8-byte load per row (gfx9 global loads take any byte
// address) instead of six byte loads behind a branch on the weight.  Unconditional: a tap with weight 0 may lie outside the frame,
// so the load is kept inside the frame's 3n bytes and its bytes are shifted to where the taps expect them (the bytes that
// fall off belong to pixels outside the frame, whose weight is 0); the products are exact integers either way.
__device__ __forceinline__ uint64_t undist_row6(const uint8_t *__restrict__ frame, int pi, int n) {
    pi = pi < -2 ? -2 : (pi > n ? n : pi);              // further out both pixels are outside
    const int want = pi * 3, last = n * 3 - 8;
    const int at = want < 0 ? 0 : (want < last ? want : last);
    uint64_t v;
    __builtin_memcpy(&v, frame + at, 8);
    const int d = want - at;                            // -6 ... 8 bytes: the load was moved to stay inside the frame
    return d >= 0 ? (d < 8 ? v >> (8 * d) : 0) : v << (8 * -d);   // e.g. pi = -1: pixel 0 is the row's SECOND tap
}
__device__ __forceinline__ uchar3 undist_mix(const uint64_t t, const uint64_t u, const uint4 iw) {
    const int w0 = (int)iw.x, w1 = (int)iw.y, w2 = (int)iw.z, w3 = (int)iw.w;
    const int r = w0 * (int)(t & 0xFF) + w1 * (int)((t >> 24) & 0xFF) + w2 * (int)(u & 0xFF) + w3 * (int)((u >> 24) & 0xFF);
    const int g = w0 * (int)((t >> 8) & 0xFF) + w1 * (int)((t >> 32) & 0xFF) + w2 * (int)((u >> 8) & 0xFF) + w3 * (int)((u >> 32) & 0xFF);
    const int b = w0 * (int)((t >> 16) & 0xFF) + w1 * (int)((t >> 40) & 0xFF) + w2 * (int)((u >> 16) & 0xFF) + w3 * (int)((u >> 40) & 0xFF);
    return make_uchar3((unsigned char)(r >> 16), (unsigned char)(g >> 16), (unsigned char)(b >> 16));
}
__device__ __forceinline__ uchar3 undist_rgb(const uint8_t *__restrict__ frame, int32_t base, uint4 iw, int w, int n) {
    return undist_mix(undist_row6(frame, base, n), undist_row6(frame, base + w, n), iw);
}

bool fused_supported(const edgehip_ctx *c);
int fused_min_batch_for(const edgehip_ctx *c, bool grey16, bool grey8);   // the dispatch rule: sequences per launch from which the one-kernel stage A runs
// what the fused kernel's first load reads: the RGB24 frame (ConvertRGB2BW fused: b+g+r), the 16-bit grey plane of
// k_undistort_grey, or an 8-bit mono frame (b+g+r of r = g = b = v: 3 v, the same integer ConvertRGB2BW computes from the
// RGB24 expansion DataSetCam makes of a mono image, image.h:197-203)
// SRC_UNDIST: the RGB24 frame resampled through the distortion map inside the load (image_undistort::undistort<true>, 4 taps with 16.16
// weights, then b+g+r) — EDGEHIP_FUSED_UNDIST=1; the default for UseUndistort stays k_undistort_grey + SRC_GREY16 (measured faster)
enum FusedSrc { SRC_RGB24 = 0, SRC_GREY16 = 1, SRC_GREY8 = 2, SRC_UNDIST = 3 };
int stage_a_fused_enqueue(edgehip_ctx *c, int slot, const uint8_t *rgb_base, const int32_t *rgb_idx, const uint16_t *grey16,
                          const uint8_t *grey8 = nullptr, bool undist_in_load = false);

}  // namespace edgehip
