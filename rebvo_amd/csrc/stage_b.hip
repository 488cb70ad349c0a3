// stage_b.hip — the tracker: uncertainty quantile, auxiliary distance field, TryVelRot evaluation and the
// Levenberg-Marquardt driver of Minimizer_RV, all device resident.
//
// Replaces (reference file:line)
//   edge_tracker::EstimateQuantile          src/mtracklib/edge_tracker.cpp:1148-1186
//   global_tracker::build_field             src/mtracklib/global_tracker.cpp:61-105
//   KltoI3PMatrix / Ne10::ProyI3Pto3PMatrix  global_tracker.cpp:553-570, include/UtilLib/ne10wrapper.h:414-424
//   global_tracker::TryVelRot<double,...>   global_tracker.cpp:289-543 (+ Calc_f_J2 :228-271, Test_f_k .h:90-104)
//   global_tracker::Minimizer_RV<double>    global_tracker.cpp:580-819
//
// Design.  The evaluation is ONE fused kernel (thread per old KeyLine): SE(3) transform, projection, field
// lookup (random 4-byte read), gather of the matched KeyLine's 32-byte record, residual, Jacobian row and
// the 21+6+1 sums of J^T J, J^T f, f^T f reduced with wavefront shuffles, LDS across the 4 waves, and one
// 28-double partial per block (fixed order => run-to-run deterministic; no float atomics).
// Between evaluations a one-wave kernel per sequence (k_lm_step) finishes the reduction and runs the LM
// logic (6x6 Jacobi-SVD / LDL^T solves, gain ratio, accept/reject, buffer swaps) so the host never waits
// inside a frame.
//
// Sequential quirks of the reference restated as parallel rules:
//  * DResidualNew[ikl] = fi where `fi` is only refreshed by a successful match (global_tracker.cpp:344,
//    391, 406): an unmatched KeyLine inherits the residual of the last matched KeyLine before it.  Inside
//    a block this is a "last valid" scan (ballot + shuffle, then LDS across waves); across blocks the
//    value is written as a marker NaN and resolved by the next reader from per-block carries that
//    k_lm_step prefixes.
//  * build_field keeps, per pixel, the smallest |t| and among equals the LAST KeyLine: one atomicMin on
//    (dist << 16 | 0xFFFF - ikl) in an LDS tile; what reaches HBM is the winner's index alone (2 bytes: the evaluation
//    never reads the distance), 8x4-pixel tiles so that the KeyLines of an edge share cache lines.
//  * The matched KeyLine is gathered as 16 bytes (c_p, m_m) with u_m recomputed when the new slot's KeyLines are
//    unrotated (GREC), else as the 32-byte MatchRec.
//  * kl.m_id_f reflects the last EVALUATED state, accepted or not (:354, :265).

#include <math.h>
#include <string.h>

#include <vector>

#include <type_traits>

#include "ctx.h"
#include "wave_reduce.h"

namespace edgehip {

// ---------------------------------------------------------------------------------------------------
// EstimateQuantile
// ---------------------------------------------------------------------------------------------------
// NT = 256 for whole batches, 1024 for a few sequences (a quarter of the dependent load round trips per block).  The histogram is
// integer, so neither the thread count nor the lane-parallel search for the first bin past the quantile changes the result.
// t_in != null (whole-frame driver): one thread of the block also does SecondThread's frame begin for its sequence (ctx.h::frame_begin;
// nothing here reads what it writes) — one dependent launch fewer per frame.
// rt.seqa != null (whole-frame driver, mono): the block's last wave also finishes the detector's reEstimateThresh for the NEW edge map
// (stage_a.hip::k_retune, whose launch stage A then leaves out): the same integer walk over its histogram, all bins at once.
struct RetuneArgs {
    SeqA *seqa;                 // [B] or null
    const int32_t *histo;       // [B][256] modulus histogram of the new edge map (k_join_histo)
    float *retuned_out;         // [B]
    int knum, nbins;
};
template <int NT>
__global__ __launch_bounds__(NT) void k_quantile(const KlSoA *kls, const int32_t *__restrict__ kns, SeqDev *seqs,
                                                 double smin, double smax, double pct, int nbins, const double *__restrict__ t_in, double fps,
                                                 RetuneArgs rt) {
    __shared__ int s_h[256];
    const int seq = blockIdx.x, tid = threadIdx.x;
    if (t_in && tid == NT - 1) frame_begin(seqs + seq, t_in[seq], fps);
    if (rt.seqa && tid >= NT - 64) {
        // k_retune's loop: i = 0, acc = 0; while (i < n && acc < knum) { i++; if (i < n) acc += h[i]; }  ->  the first i >= 1 whose
        // sum h[1..i] reaches knum, else n (0 when knum <= 0); bin 0 is never counted (edge_finder.cpp:400-403)
        const int l = tid - (NT - 64);
        int h4[4], inc4[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = l * 4 + j;
            h4[j] = (b >= 1 && b < rt.nbins) ? rt.histo[(size_t)seq * 256 + b] : 0;
            tot += h4[j];
            inc4[j] = tot;
        }
        int inc = tot;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (l >= o) inc += up;
        }
        const int base = inc - tot;
        int first = rt.nbins;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int b = l * 4 + j;
            if (b >= 1 && b < rt.nbins && base + inc4[j] >= rt.knum) first = b;
        }
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        if (rt.knum <= 0) first = 0;
        if (l == 0) {
            SeqA *sa = rt.seqa + seq;
            const float mxd = sa->nm_max, mnd = sa->nm_min;
            float r = mxd - (float)first * (mxd - mnd) / (float)rt.nbins;
            if (sa->kn_new <= 0) r = 0.f;
            sa->retuned = r;
            rt.retuned_out[seq] = r;
        }
    }
    if (tid < 256) s_h[tid] = 0;
    __syncthreads();
    const int kn = kns[seq];
    const double *s_rho = kls[seq].s_rho;
    for (int i0 = tid; i0 < kn; i0 += 4 * NT) {
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = i0 + j * NT < kn ? s_rho[i0 + j * NT] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j * NT >= kn) break;
            int b = x86_cvttsd2si((double)nbins * (v[j] - smin) / (smax - smin));
            b = b > nbins - 1 ? nbins - 1 : b;
            b = b < 0 ? 0 : b;
            atomicAdd(&s_h[b], 1);
        }
    }
    __syncthreads();
    if (tid < 64) {
        // first bin i whose exclusive prefix a_i = sum_{k<i} h[k] exceeds pct * kn (the loop of edge_tracker.cpp:1170-1181, all bins at once)
        int h[4], ex[4];
        int tot = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { h[j] = s_h[tid * 4 + j]; ex[j] = tot; tot += h[j]; }
        int inc = tot;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (tid >= o) inc += up;
        }
        const int base = inc - tot;
        int first = 1 << 30;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int i = tid * 4 + j;
            if (i < nbins && (double)(base + ex[j]) > pct * (double)kn) first = i;
        }
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        if (tid == 0) seqs[seq].pub.s_rho_q = first < nbins ? (double)first * (smax - smin) / (double)nbins + smin : 1e3;
    }
}

#ifdef EDGEHIP_EXPERIMENTS   // reference-shaped scatter with global atomics: A/B measurements only (EDGEHIP_FIELD_MODE=1)
// ---------------------------------------------------------------------------------------------------
// build_field: thread per (KeyLine, t) pair, t in [-r, r)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_field_scatter(const KlSoA *kls, const int32_t *__restrict__ kns,
                                                       const float *__restrict__ retuned, uint32_t *__restrict__ field,
                                                       int w, int h, size_t fstride, int ftx, int radius, float min_mod_arg) {
    const int seq = blockIdx.z;
    const int kn = kns[seq];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int span = 2 * radius;
    const int ikl = idx / span;
    if (ikl >= kn) return;
    const int t = idx - ikl * span - radius;
    const KlSoA &k = kls[seq];
    const float min_mod = min_mod_arg < 0.f ? retuned[seq] : min_mod_arg;
    const MatchRec r = k.rec[ikl];
    if (min_mod > 0 && r.n_m < min_mod) return;
    const float fx = r.u_mx * (float)t + r.c_px;
    const float fy = r.u_my * (float)t + r.c_py;
    const int xi = round_half_away_i(fx), yi = round_half_away_i(fy);  // Image::GetIndexRC uses round()
    if (xi >= w || yi >= h || xi < 0 || yi < 0) return;
    const uint32_t at = (uint32_t)(t < 0 ? -t : t);
    atomicMin(&field[(size_t)seq * fstride + field_index(xi, yi, ftx)], (at << 16) | (uint32_t)(0xFFFF - ikl));
}
#endif   // EDGEHIP_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------
// build_field, tiled: one block per FT x FT pixel tile.  The per-pixel result of the sequential scatter is
// order independent (min over dist<<16 | 0xFFFF-ikl), so instead of ~kn*2r global atomics the block
//   1. walks img_mask_kl over the tile grown by the radius (the only KeyLines whose segment can reach it),
//      FG rows at a time, compacting the KeyLine ids it finds into an LDS list,
//   2. lets each thread take KeyLines off the list and rasterise only the t-range that can fall in the tile
//      (same float expression as the reference, :78), min-reducing into an LDS copy of the tile,
//   3. stores the tile coalesced (which also replaces the field clear).
// ---------------------------------------------------------------------------------------------------
constexpr int FT = 64;    // tile edge
constexpr int FG = 12;    // mask rows scanned per list fill

__global__ __launch_bounds__(256) void k_field_tiles(const KlSoA *kls, const int32_t *__restrict__ masks,
                                                     const float *__restrict__ retuned, uint32_t *__restrict__ field,
                                                     int w, int h, size_t fstride, int ftx, int radius, float min_mod_arg) {
    __shared__ uint32_t s_tile[FT * FT];
    __shared__ int s_list[4096];
    __shared__ int s_cnt;
    const int seq = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int tx0 = blockIdx.x * FT, ty0 = blockIdx.y * FT;
    const KlSoA &k = kls[seq];
    const int32_t *mask = masks + (size_t)seq * ((size_t)w * h);
    const float min_mod = min_mod_arg < 0.f ? retuned[seq] : min_mod_arg;
    for (int i = tid; i < FT * FT; i += 256) s_tile[i] = 0xFFFFFFFFu;
    if (tid == 0) s_cnt = 0;
    // region of KeyLine centres that can reach the tile: c_p = pixel + (xs,ys), |xs|,|ys| <= 0.5, |u*t| <= r
    const int rx0 = max(tx0 - radius - 1, 0), rx1 = min(tx0 + FT + radius + 1, w);
    const int ry0 = max(ty0 - radius - 1, 0), ry1 = min(ty0 + FT + radius + 1, h);
    const int rw = rx1 - rx0;
    const int rows_per_fill = max(1, min(FG, 4096 / rw));
    __syncthreads();
    for (int yg = ry0; yg < ry1; yg += rows_per_fill) {
        const int ye = min(yg + rows_per_fill, ry1);
        const int npx = (ye - yg) * rw;
        for (int base = 0; base < npx; base += 256) {
            const int idx = base + tid;
            int id = -1;
            if (idx < npx) {
                const int yy = yg + idx / rw, xx = rx0 + idx % rw;
                id = mask[(size_t)yy * w + xx];
            }
            const unsigned long long bal = __ballot(id >= 0);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(&s_cnt, __popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (id >= 0) s_list[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = id;
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int li = tid; li < cnt; li += 256) {
            const int ikl = s_list[li];
            const MatchRec r = k.rec[ikl];
            if (min_mod > 0 && r.n_m < min_mod) continue;
            // conservative t-range whose samples can land inside the tile (exact test per sample below)
            float tlo = (float)(-radius), thi = (float)(radius - 1);
            const float bx0 = (float)tx0 - 1.f - r.c_px, bx1 = (float)(tx0 + FT) - r.c_px;
            const float by0 = (float)ty0 - 1.f - r.c_py, by1 = (float)(ty0 + FT) - r.c_py;
            if (fabsf(r.u_mx) > 1e-6f) {
                const float a = bx0 / r.u_mx, b = bx1 / r.u_mx;
                tlo = fmaxf(tlo, fminf(a, b) - 1.f);
                thi = fminf(thi, fmaxf(a, b) + 1.f);
            } else if (bx0 > 0.f || bx1 < 0.f) continue;
            if (fabsf(r.u_my) > 1e-6f) {
                const float a = by0 / r.u_my, b = by1 / r.u_my;
                tlo = fmaxf(tlo, fminf(a, b) - 1.f);
                thi = fminf(thi, fmaxf(a, b) + 1.f);
            } else if (by0 > 0.f || by1 < 0.f) continue;
            const int t0 = (int)floorf(tlo), t1 = (int)ceilf(thi);
            const uint32_t idk = (uint32_t)(0xFFFF - ikl);
            for (int t = max(t0, -radius); t <= min(t1, radius - 1); t++) {
                const float fx = r.u_mx * (float)t + r.c_px;
                const float fy = r.u_my * (float)t + r.c_py;
                const int xi = round_half_away_i(fx), yi = round_half_away_i(fy);  // Image::GetIndexRC uses round()
                if (xi >= w || yi >= h || xi < 0 || yi < 0) continue;
                const int lx = xi - tx0, ly = yi - ty0;
                if ((unsigned)lx >= (unsigned)FT || (unsigned)ly >= (unsigned)FT) continue;
                const uint32_t at = (uint32_t)(t < 0 ? -t : t);
                atomicMin(&s_tile[ly * FT + lx], (at << 16) | idk);
            }
        }
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
    }
    // store in the 4x4-tiled layout: 16 consecutive threads write one 64-B tile, a tile row of the block is 1 KB
    // contiguous (FT and the block origin are multiples of 4)
    uint32_t *out = field + (size_t)seq * fstride;
    for (int i = tid; i < FT * FT; i += 256) {
        const int st = i >> 4, in = i & 15;
        const int lx = ((st % (FT / 4)) << 2) | (in & 3), ly = ((st / (FT / 4)) << 2) | (in >> 2);
        const int x = tx0 + lx, y = ty0 + ly;
        if (x < w && y < h) out[field_index(x, y, ftx)] = s_tile[ly * FT + lx];
    }
}

// ---------------------------------------------------------------------------------------------------
// build_field, binned: (1) k_field_bin — thread per KeyLine: which FT x FT tiles can its +-radius segment
// touch?  Ids are appended to per-tile bins; the 256 KeyLines of a block are raster neighbours and hit the
// same few tiles, so ranks are taken from LDS counters and only one global atomic per (block, tile) reserves
// the slots.  (2) k_field_raster — block per tile: rasterise the binned KeyLines' in-tile t-ranges into an
// LDS tile with atomicMin, then store the tile coalesced (no separate clear pass).  Bin order is
// irrelevant: the per-pixel result is a min.
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxTiles = 256;

__device__ __forceinline__ bool tile_trange(const MatchRec &r, int tx0, int ty0, int radius, int &t0, int &t1) {
    // t-range whose samples can round into the tile [tx0,tx0+FT) x [ty0,ty0+FT): pixel x = round(u t + c) is inside iff
    // u t + c lies in [tx0 - 0.5, tx0 + FT - 0.5]; conservative by 0.01 px (the sample is evaluated in float: < 1e-4 px off)
    // and by a slack on t that covers the rounding of the bounds (c and the reciprocal: relative 1e-7, amplified by 1/u).
    // Every sample kept here is still tested against the tile exactly; a sample dropped here must be outside.  Both users —
    // the binning and the rasteriser — go through this one function.  (v_rcp_f32 + products instead of IEEE divisions.)
    float tlo = (float)(-radius), thi = (float)(radius - 1);
    const float bx0 = (float)tx0 - 0.51f - r.c_px, bx1 = (float)(tx0 + FT) - 0.49f - r.c_px;
    const float by0 = (float)ty0 - 0.51f - r.c_py, by1 = (float)(ty0 + FT) - 0.49f - r.c_py;
    if (fabsf(r.u_mx) > 1e-6f) {
        const float iu = __builtin_amdgcn_rcpf(r.u_mx);
        const float a = bx0 * iu, b = bx1 * iu, sl = 0.25f + 1e-4f * fabsf(iu);
        tlo = fmaxf(tlo, fminf(a, b) - sl);
        thi = fminf(thi, fmaxf(a, b) + sl);
    } else if (bx0 > 0.f || bx1 < 0.f) return false;
    if (fabsf(r.u_my) > 1e-6f) {
        const float iu = __builtin_amdgcn_rcpf(r.u_my);
        const float a = by0 * iu, b = by1 * iu, sl = 0.25f + 1e-4f * fabsf(iu);
        tlo = fmaxf(tlo, fminf(a, b) - sl);
        thi = fminf(thi, fmaxf(a, b) + sl);
    } else if (by0 > 0.f || by1 < 0.f) return false;
    t0 = max((int)ceilf(tlo), -radius);
    t1 = min((int)floorf(thi), radius - 1);
    return t0 <= t1;
}

__global__ __launch_bounds__(256) void k_field_bin(const KlSoA *kls, const int32_t *__restrict__ kns,
                                                   const float *__restrict__ retuned, int32_t *__restrict__ bin_cnt,
                                                   int32_t *__restrict__ bins, int w, int h, int radius, float min_mod_arg,
                                                   int ntx, int nty, int bin_cap, unsigned long long *__restrict__ fwd_key,
                                                   int32_t *__restrict__ fwd_win) {
    __shared__ int s_cnt[kMaxTiles], s_base[kMaxTiles];
    const int seq = blockIdx.z, tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const int kn = kns[seq];
    if (blockIdx.x * 256 >= kn) return;

    const int ntiles = ntx * nty;
    for (int t = tid; t < ntiles; t += 256) s_cnt[t] = 0;
    __syncthreads();
    bool active = i < kn;
    MatchRec r;
    int txa = 0, txb = -1, tya = 0, tyb = -1;
    if (active) {
        r = kls[seq].rec[i];
        const float min_mod = min_mod_arg < 0.f ? retuned[seq] : min_mod_arg;
        if (min_mod > 0 && r.n_m < min_mod) active = false;
    }
    if (active) {
        // bounding box of the pixels the samples can round into: +-1 px beyond the segment's ends (a sample at
        // x = 383.5 lands on pixel 384, i.e. in the NEXT tile column, however flat the segment is; tile_trange then
        // decides exactly)
        const float rr = (float)radius + 1.5f;
        const float xa = r.c_px - fabsf(r.u_mx) * rr - 1.f, xb = r.c_px + fabsf(r.u_mx) * rr + 1.f;
        const float ya = r.c_py - fabsf(r.u_my) * rr - 1.f, yb = r.c_py + fabsf(r.u_my) * rr + 1.f;
        txa = max((int)floorf(xa) / FT, 0); txb = min((int)floorf(xb) / FT, ntx - 1);
        tya = max((int)floorf(ya) / FT, 0); tyb = min((int)floorf(yb) / FT, nty - 1);
        if (xb < 0.f || yb < 0.f) txb = -1;
    }
    // pass A: count per tile (LDS).  Which tiles of the bounding box the segment really touches is remembered in a bit mask
    // for pass B (up to 32 tiles: always at the shipped search ranges, where a segment spans at most 3 x 3)
    const int bw = txb - txa + 1;
    const bool cached = bw > 0 && (tyb - tya + 1) * bw <= 32;
    uint32_t hitmask = 0;
    for (int ty = tya; ty <= tyb; ty++)
        for (int tx = txa; tx <= txb; tx++) {
            int t0, t1;
            if (tile_trange(r, tx * FT, ty * FT, radius, t0, t1)) {
                atomicAdd(&s_cnt[ty * ntx + tx], 1);
                if (cached) hitmask |= 1u << ((ty - tya) * bw + (tx - txa));
            }
        }
    __syncthreads();
    for (int t = tid; t < ntiles; t += 256) {
        const int c = s_cnt[t];
        s_base[t] = c ? atomicAdd(&bin_cnt[(size_t)seq * kMaxTiles + t], c) : 0;
        s_cnt[t] = 0;
    }
    __syncthreads();
    // pass B: write ids
    for (int ty = tya; ty <= tyb; ty++)
        for (int tx = txa; tx <= txb; tx++) {
            int t0, t1;
            if (cached ? ((hitmask >> ((ty - tya) * bw + (tx - txa))) & 1u) != 0u : tile_trange(r, tx * FT, ty * FT, radius, t0, t1)) {
                const int t = ty * ntx + tx;
                const int pos = s_base[t] + atomicAdd(&s_cnt[t], 1);
                if (pos < bin_cap) bins[((size_t)seq * ntiles + t) * bin_cap + pos] = i;
            }
        }
    // whole-frame driver: this kernel visits every KeyLine of the NEW edge map right before the minimisation, so it also
    // resets FordwardMatch's arbitration entries of that KeyLine (12 B in a stream) instead of two memsets over [B][CAP] —
    // at the end: a load issued behind a store waits for it
    if (fwd_key && i < kn) { fwd_key[(size_t)seq * bin_cap + i] = 0ull; fwd_win[(size_t)seq * bin_cap + i] = -1; }
}

#ifndef EDGEHIP_NT_RASTER
#define EDGEHIP_NT_RASTER 1   // the 16-bit plane as streaming stores: B.build_field 1162 -> 1154 us, the evaluations that gather from it unchanged
#endif
__global__ __launch_bounds__(256) void k_field_raster(const KlSoA *kls, int32_t *__restrict__ bin_cnt,
                                                      const int32_t *__restrict__ bins, uint32_t *__restrict__ field,
                                                      uint16_t *__restrict__ field16, size_t f16stride, int f16tx, int keep32,
                                                      int w, int h, size_t fstride, int ftx, int radius, int ntx, int bin_cap) {
    // Row stride FT + 2 words, not FT (round 6): the 32-bit LDS atomics of a wave are served in two groups of 32 lanes on 32 banks, and
    // neighbouring lanes rasterise neighbouring KeyLines of one edge at the same t — samples one pixel apart along the edge.  With a
    // 64-word row a vertical edge puts all 32 lanes of a group on ONE bank (32-way); at 66 words the bank moves by 2 per row, by 1 per
    // column and by 3 / 1 per diagonal step: at most 2-way for any edge direction (SQ_LDS_BANK_CONFLICT of this kernel: a third of its
    // LDS cycles before).  65 would make vertical edges free and anti-diagonal ones 32-way.
#ifndef EDGEHIP_RASTER_PAD
#define EDGEHIP_RASTER_PAD 2
#endif
    constexpr int TS = FT + EDGEHIP_RASTER_PAD;
    __shared__ uint32_t s_tile[FT * TS];
    const int ntiles = ntx * gridDim.y;
    const int seq = blockIdx.z, tid = threadIdx.x;
    const int tx0 = blockIdx.x * FT, ty0 = blockIdx.y * FT;
    const int tile = blockIdx.y * ntx + blockIdx.x;
    for (int i = tid; i < FT * TS; i += 256) s_tile[i] = 0xFFFFFFFFu;
    __syncthreads();
    const int cnt = min(bin_cnt[(size_t)seq * kMaxTiles + tile], bin_cap);
    const int32_t *list = bins + ((size_t)seq * ntiles + tile) * bin_cap;
    const KlSoA &k = kls[seq];
    // in-tile AND in-image in two unsigned compares: lx >= 0 implies xi >= tx0 >= 0, and the tile is clipped to
    // the image once per block
    const unsigned ex = (unsigned)min(FT, w - tx0), ey = (unsigned)min(FT, h - ty0);
    // FSPLIT threads share one KeyLine's in-tile t-range: a bin holds ~170 KeyLines with ranges of 1..2r samples, so
    // one thread per KeyLine leaves lanes idle and the rest waiting for the longest range; every further split pays the
    // per-part set-up again.
#ifndef EDGEHIP_FSPLIT
#define EDGEHIP_FSPLIT 2   // measured at r = 40 after the cheaper tile range: 1: 1296, 2: 1260, 3: 1292, 4: 1310, 8: 1538, 16: 2054 us per 1024 frames
#endif
#ifndef EDGEHIP_RASTER_ABL
#define EDGEHIP_RASTER_ABL 0   // timing experiments only, wrong fields by design (tools/experiments/CALLS.md: r04_r): 1 plain store for
#endif                         // the atomic, 2 no LDS access, 3 no samples, 4 = 3 and no output stage, 5 = 3 and no record gather, 6 = 5 and no bin read
#ifndef EDGEHIP_RASTER_UNROLL
#define EDGEHIP_RASTER_UNROLL 2   // 1198 -> 1156 us per 1024 frames (same-box A/B, tools/experiments/CALLS.md: r04_g)
#endif
    // (A split chosen per tile so that the last round of 256 threads is as full as possible — 1..4 parts, block-uniform —
    // measured slower, 1198 -> 1283 us: the run-time divisor costs every item more than the fuller rounds save.  So did dealing the
    // samples of 256 KeyLines to the threads in equal shares (ranges parked in LDS, a scan of their lengths, one binary search per
    // thread, then a flat walk over samples and KeyLines): 1158 -> 1434 us — the ablations say why: the sample loop already runs at
    // ~17 lane-cycles per sample for ~16 vector instructions, i.e. with nearly full lanes; the flat walk adds instructions to every
    // sample and saves waiting that is not there.  Where the kernel's ~970 us go (r = 40, 15.4 k KeyLines, profiles/r04_r_raster_ablations.txt):
    // samples 526, the 16-bit plane's store 160 (0.74 GB: the HBM rate), record gathers 104, bin reads 57, tile clear / ranges / ramp ~120.)
    constexpr int FSPLIT = EDGEHIP_FSPLIT;
    // The hardware rounding differs from round() in a way that matters only for a coordinate of exactly -0.5 (pixel 0
    // instead of -1, ctx.h): that can only be accepted by a tile that starts at column / row 0, so only the tiles on
    // the left / top image border pay for the fix-up.  (Block-uniform choice of one of four loop bodies.)
#ifndef EDGEHIP_RASTER_LEAN
#define EDGEHIP_RASTER_LEAN 1
#endif
    const unsigned ex4 = ex << 2;
    const int neg4tx0 = -4 * tx0;
    auto raster = [&](auto fix_x, auto fix_y, auto narrow) {
        for (int wi = tid; wi < cnt * FSPLIT; wi += 256) {
            const int li = wi / FSPLIT, part = wi - li * FSPLIT;
#if EDGEHIP_RASTER_ABL == 6     // no bin read
            const int ikl = li * 7;
#else
            const int ikl = list[li];
#endif
#if EDGEHIP_RASTER_ABL >= 5     // no record gather
            MatchRec r;
            r.c_px = (float)(tx0 + (ikl & 63)); r.c_py = (float)(ty0 + ((ikl >> 6) & 63)); r.u_mx = 0.6f; r.u_my = 0.8f;
#else
            // (the record's first half — c_p, u_m: all the rasteriser reads — as a global load: a FLAT one makes the wait behind it a wait for the tile's LDS atomics too)
            MatchRec r;
            { const float4 q = ldg(reinterpret_cast<const float4 *>(k.rec), 2 * (size_t)ikl); r.c_px = q.x; r.c_py = q.y; r.u_mx = q.z; r.u_my = q.w; }
#endif
            int t0, t1;
            if (!tile_trange(r, tx0, ty0, radius, t0, t1)) continue;
            {
                const int chunk = (t1 - t0 + FSPLIT) / FSPLIT;   // ceil(len / FSPLIT)
                t0 += part * chunk;
                t1 = min(t1, t0 + chunk - 1);
            }
            const uint32_t idk = (uint32_t)(0xFFFF - ikl);
#if EDGEHIP_RASTER_ABL == 2
            uint32_t acc_abl = 0;
#endif
#if EDGEHIP_RASTER_ABL >= 3      // the set-up only, no samples
            if (t0 == 12345) s_tile[0] = idk;
            continue;
#endif
            // t runs as a float (|t| <= 255: every value and the increment are exact), so the reference's (float)t costs
            // nothing and |t| is an operand modifier of the one conversion back
            const float t1f = (float)t1;
#ifndef EDGEHIP_RASTER_PK
#define EDGEHIP_RASTER_PK 0   // 1: x and y of a sample as one packed pair (v_pk_mul_f32 + v_pk_add_f32, each component rounded like the scalar op): two
                              // instructions fewer per sample on paper, 6 % SLOWER measured (B.build_field 1155 -> 1230 us per 1024 frames, same box,
                              // profiles/r06_raster_pad_and_packed_ab.txt): the packed forms issue at half rate and want their operands in register pairs
#endif
#if EDGEHIP_RASTER_PK
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f u2 = {r.u_mx, r.u_my}, c2 = {r.c_px, r.c_py};
#endif
            auto sample = [&](const float tf) __attribute__((always_inline)) {
#if EDGEHIP_RASTER_PK
                const v2f t2 = {tf, tf};
                const v2f f2 = u2 * t2 + c2;             // global_tracker.cpp:78, the same two float expressions (no contraction: -ffp-contract=off)
                const float fx = f2.x, fy = f2.y;
#else
                const float fx = r.u_mx * tf + r.c_px;   // global_tracker.cpp:78, same float expression
                const float fy = r.u_my * tf + r.c_py;
#endif
                // Image::GetIndexRC uses round()
#if EDGEHIP_RASTER_LEAN && EDGEHIP_RASTER_ABL == 0
                // Two vector instructions fewer per in-tile sample (of ~16; the loop runs the vector ALU at 1.0): the column comes out as a BYTE offset
                // in one shift-add — (round(fx) << 2) - 4 tx0, compared against 4 ex: the same test, |lx| is far below 2^29 — and |t|, an exact small
                // integer in a float, goes into byte 2 of the stored word with v_cvt_pk_u8_f32 (one instruction for the conversion, the shift and the OR;
                // |t| <= radius <= 255 is the host's condition for this form: `narrow`).
                const int rx = decltype(fix_x)::value ? round_half_away_i(fx) : round_ties_up_i(fx);
                const int ly = (decltype(fix_y)::value ? round_half_away_i(fy) : round_ties_up_i(fy)) - ty0;
                const unsigned lx4 = (unsigned)((rx << 2) + neg4tx0);
                if (lx4 >= ex4 || (unsigned)ly >= ey) return;
                const uint32_t val = decltype(narrow)::value ? __builtin_amdgcn_cvt_pk_u8_f32(fabsf(tf), 2u, idk) : (((uint32_t)fabsf(tf) << 16) | idk);
                atomicMin(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_tile) + (__umul24((unsigned)ly, (unsigned)(TS * 4)) + lx4)), val);
#else
                const int lx = (decltype(fix_x)::value ? round_half_away_i(fx) : round_ties_up_i(fx)) - tx0;
                const int ly = (decltype(fix_y)::value ? round_half_away_i(fy) : round_ties_up_i(fy)) - ty0;
                if ((unsigned)lx >= ex || (unsigned)ly >= ey) return;
                const uint32_t at = (uint32_t)fabsf(tf);
#if EDGEHIP_RASTER_ABL == 1      // a plain store instead of the atomic
                s_tile[ly * TS + lx] = (at << 16) | idk;
#elif EDGEHIP_RASTER_ABL == 2    // no LDS access at all
                acc_abl += (at << 16) | idk | (uint32_t)(ly * TS + lx);
#else
                atomicMin(&s_tile[__umul24((unsigned)ly, (unsigned)TS) + (unsigned)lx], (at << 16) | idk);   // (a 24-bit product: the 32-bit one runs at a quarter of the rate, once per sample)
#endif
#endif
            };
#if EDGEHIP_RASTER_UNROLL == 2
            // two samples per trip: the loop's own bookkeeping (exec-mask save / restore, compare, branch: as many scalar
            // instructions as the sample's arithmetic) is paid once per pair
            float tf = (float)t0;
            for (; tf + 1.f <= t1f; tf += 2.f) { sample(tf); sample(tf + 1.f); }
            if (tf <= t1f) sample(tf);
#else
            for (float tf = (float)t0; tf <= t1f; tf += 1.f) sample(tf);
#endif
#if EDGEHIP_RASTER_ABL == 2
            if (acc_abl == 0x12345u) s_tile[1] = acc_abl;
#endif
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (radius <= 255) {
        if (tx0 == 0) { if (ty0 == 0) raster(T{}, T{}, T{}); else raster(T{}, F{}, T{}); }
        else { if (ty0 == 0) raster(F{}, T{}, T{}); else raster(F{}, F{}, T{}); }
    } else {
        if (tx0 == 0) { if (ty0 == 0) raster(T{}, T{}, F{}); else raster(T{}, F{}, F{}); }
        else { if (ty0 == 0) raster(F{}, T{}, F{}); else raster(F{}, F{}, F{}); }
    }
    __syncthreads();
    if (tid == 0) bin_cnt[(size_t)seq * kMaxTiles + tile] = 0;   // every tile's count is consumed by exactly this block: ready for the next k_field_bin (no memset launch)
#if EDGEHIP_RASTER_ABL == 4
    if (s_tile[tid] != 0x12345u) return;
#endif
    // store in the 4x4-tiled layout: 16 consecutive threads write one 64-B tile, a tile row of the block is 1 KB
    // contiguous (FT and the block origin are multiples of 4)
    // (the {dist, ikl} form is kept only for edgehip_download_field: params.debug_planes)
    if (keep32) {
        uint32_t *out = field + (size_t)seq * fstride;
        for (int i = tid; i < FT * FT; i += 256) {
            const int st = i >> 4, in = i & 15;
            const int lx = ((st % (FT / 4)) << 2) | (in & 3), ly = ((st / (FT / 4)) << 2) | (in >> 2);
            const int x = tx0 + lx, y = ty0 + ly;
            if (x < w && y < h) out[field_index(x, y, ftx)] = s_tile[ly * TS + lx];
        }
    }
    // What the tracker gathers: the KeyLine-index plane, ikl + 1 (0 = empty) in 8x4-pixel tiles of 64 B; a thread
    // stores two pixels (w is a multiple of 4: a pair is inside or outside the image together), 16 consecutive threads
    // one tile
    uint16_t *o16 = field16 + (size_t)seq * f16stride;
    for (int i = tid; i < FT * FT / 2; i += 256) {
        const int st = i >> 4, in = i & 15;
        const int lx = ((st % (FT / 8)) << 3) | ((in & 3) << 1), ly = ((st / (FT / 8)) << 2) | (in >> 2);
        const int x = tx0 + lx, y = ty0 + ly;
        if (x < w && y < h) {
            const uint32_t v0 = s_tile[ly * TS + lx], v1 = s_tile[ly * TS + lx + 1];
            // low half = 0xFFFF - ikl  ->  ikl + 1 = 0x10000 - low half (mod 2^16); empty (all ones) -> 0
            const uint32_t a0 = v0 == 0xFFFFFFFFu ? 0u : ((0x10000u - (v0 & 0xFFFFu)) & 0xFFFFu);
            const uint32_t a1 = v1 == 0xFFFFFFFFu ? 0u : ((0x10000u - (v1 & 0xFFFFu)) & 0xFFFFu);
#if EDGEHIP_NT_RASTER
            st_stream(reinterpret_cast<uint32_t *>(o16 + field16_index(x, y, f16tx)), a0 | (a1 << 16));
#else
            *reinterpret_cast<uint32_t *>(o16 + field16_index(x, y, f16tx)) = a0 | (a1 << 16);
#endif
        }
    }
}

// u32 field (dist<<16 | 0xFFFF-ikl, 4x4 tiles) -> u16 KeyLine-index plane (ikl+1, 8x4 tiles): thread per pixel
__global__ __launch_bounds__(256) void k_field_to16(const uint32_t *__restrict__ field, uint16_t *__restrict__ f16, int w, int h,
                                                    size_t fstride, int ftx, size_t f16stride, int f16tx) {
    const int seq = blockIdx.z;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // element of the u16 plane
    if (e >= f16stride) return;
    const size_t tile = e >> 5; const int in = (int)(e & 31);
    const int x = (int)(tile % (size_t)f16tx) * 8 + (in & 7), y = (int)(tile / (size_t)f16tx) * 4 + (in >> 3);
    uint16_t v = 0;
    if (x < w && y < h) {
        const uint32_t f = field[(size_t)seq * fstride + field_index(x, y, ftx)];
        if (f != 0xFFFFFFFFu) v = (uint16_t)(0xFFFF - (int)(f & 0xFFFFu) + 1);
    }
    f16[(size_t)seq * f16stride + e] = v;
}

// ---------------------------------------------------------------------------------------------------
// KltoI3PMatrix + ProyI3Pto3PMatrix: P0 = (x*z/zf, y*z/zf, z), z = 1/rho  (once per frame pair)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tvr_prepare(const KlSoA *kls, const int32_t *__restrict__ kns, double *__restrict__ P0,
                                                     double *__restrict__ resid0, double *__restrict__ carry0, SeqDev *seqs,
                                                     int cap, int nblk, double zfm) {
    const int seq = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int kn = kns[seq];
    if (i == 0) seqs[seq].kn_old = kn;
    if ((i % kTvrBlock) == 0 && i / kTvrBlock < nblk) carry0[(size_t)seq * nblk + i / kTvrBlock] = 0.0;
    if (i >= kn) return;
    // (P0 itself is not stored any more: k_try_velrot / k_try_vel rebuild it from p_m and rho, 16 B instead of 24 B per read)
    resid0[(size_t)seq * cap + i] = 0.0;  // "Init residuals", global_tracker.cpp:625
}

// ---------------------------------------------------------------------------------------------------
// SO3 exponential as TooN::SO3<>::exp + rodrigues_so3_exp (TooN so3.h:203-285)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void so3_exp_inl(const double (&w)[3], double (&R)[9]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double theta = sqrt(theta_sq);
    double A, B;
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
    } else if (theta_sq < 1e-6) {
        B = 0.5 - 0.25 * one_6th * theta_sq;
        A = 1.0 - theta_sq * one_6th * (1.0 - one_20th * theta_sq);
    } else {
        const double inv_theta = 1.0 / theta;
        A = sin(theta) * inv_theta;
        B = (1 - cos(theta)) * (inv_theta * inv_theta);
    }
    const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R[0] = 1.0 - B * (wy2 + wz2);
    R[4] = 1.0 - B * (wx2 + wz2);
    R[8] = 1.0 - B * (wx2 + wy2);
    double a = A * w[2], b = B * (w[0] * w[1]);
    R[1] = b - a; R[3] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R[2] = b + a; R[6] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R[5] = b - a; R[7] = b + a;
}
// (out of line for the callers that are not on anybody's critical path: its arrays then live in scratch memory)
__device__ __noinline__ void so3_exp(const double w[3], double R[9]) {
    const double wv[3] = {w[0], w[1], w[2]};
    double Rr[9];
    so3_exp_inl(wv, Rr);
    for (int i = 0; i < 9; i++) R[i] = Rr[i];
}

// Per-evaluation constants of TryVelRot (global_tracker.cpp:309-341): R0 = exp(W), RM = 2x2 block of
// exp((0,0,W_z)), Vt = V.
__device__ __noinline__ void tvr_setup(SeqDev *sq, const double *X) {
    double R0[9], Rz[9];
    so3_exp(X + 3, R0);
    const double wz[3] = {0.0, 0.0, X[5]};
    so3_exp(wz, Rz);
    for (int i = 0; i < 9; i++) sq->Rt[i] = R0[i];  // row-major R0(i,j)
    sq->RM[0] = Rz[0]; sq->RM[1] = Rz[1]; sq->RM[2] = Rz[3]; sq->RM[3] = Rz[4];
    for (int i = 0; i < 3; i++) sq->Vt[i] = X[i];
}


// The same on two lanes of a wave: lane 0 takes exp(W) (and Vt), lane 1 exp((0,0,W_z)) — the two exponentials are the longest
// stretch of the LM step (sin and cos in double precision, 3.6 of its 9 us for a single camera), and they do not depend on
// each other.  X in LDS / global, visible to both lanes.
// `zero`: the transform of the zero-init chain (zRt / zVt / zRM) instead of the running one; the pair of lanes is (l0, l0 + 1).
__device__ __forceinline__ void tvr_setup2(SeqDev *sq, const double *X, const int lane, const bool zero = false, const int l0 = 0) {
    if (lane >= l0 && lane < l0 + 2) {
        const bool first = lane == l0;
        const double wv[3] = {first ? X[3] : 0.0, first ? X[4] : 0.0, X[5]};
        double R[9];
        so3_exp_inl(wv, R);   // inlined: R stays in registers (through a call it is scratch memory, a round trip each way)
        double *Rt = zero ? sq->zRt : sq->Rt, *Vt = zero ? sq->zVt : sq->Vt, *RM = zero ? sq->zRM : sq->RM;
        if (first) {
            for (int i = 0; i < 9; i++) Rt[i] = R[i];
            for (int i = 0; i < 3; i++) Vt[i] = X[i];
        } else {
            RM[0] = R[0]; RM[1] = R[1]; RM[2] = R[3]; RM[3] = R[4];
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// TryVelRot
// ---------------------------------------------------------------------------------------------------
struct TvrArgs {
    const KlSoA *kl_old;       // [B]
    const KlSoA *kl_new;       // [B]  (field KeyLines)
    const int32_t *kn_old;     // [B]
    const uint16_t *field16;   // [B][f16stride] KeyLine-index plane of the field (ctx.h field16_index)
    const double *P0;          // [B][3][CAP]
    double *resid;             // [kResidBufs][B][CAP]
    double *resid_carry;       // [kResidBufs][B][nblk]  resolved carry-in per block
    double *block_last;        // [B][nblk] last valid fi of each block (marker NaN if none) of THIS call
    double *partials;          // [B][nblk][kNumSums]
    double *block_last_z;      // the same two for the zero-init chain of a two-chain evaluation (k_try_velrot2)
    double *partials_z;
    SeqDev *seq;
    const uint32_t *framecount;  // [B] of the new slot
    int w, h, cap, nblk, nseq;
    size_t f16stride;          // index-plane elements per sequence
    int f16tx;                 // 8x4-pixel tiles per tile row
    double zfm, max_r, match_thresh, k_huber, inv_k_huber;
    double inv_zfm;            // 1 / zfm as IEEE division gives it (the host's): the kernels formed it per KeyLine and evaluation, a full fp64
                               // division sequence (~22 of ~600 vector instructions of an evaluation that is vector-ALU-bound)
    float ppx, ppy;
    uint32_t match_num_thresh;
    int write_mid;             // store kl.m_id_f (only the last evaluation of a minimisation needs to)
    int use_grec;   // host-side choice of the gather record (edgehip_ctx::grec_ok of the new slot)
    unsigned long long *fwd_key;   // [B][CAP] or null: with write_mid, also post FordwardMatch's key of the matched new KeyLine
                                   // (edge_tracker.cpp:413: the old KeyLine with the larger rho wins) — saves k_fwd_key's pass
    // KF instantiation (kfvo::TryVelRot): per-sequence scale ratio, the thresholds of Calc_f_J_Complete
    const edgehip_kf_request *kf;   // [B]
    double kf_match_mod, kf_match_cang, kf_rho_tol;
};

// 1 / sqrt(x) for x >= 1: hardware estimate + two Newton steps (relative error of a few 1e-16; x < 1e300)
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return x > 1e300 ? 0.0 : y;   // 1 / sqrt(inf) = 0 as the reference's x / sqrt(inf) gives; NaN stays NaN
}

__device__ __forceinline__ bool is_carry(double v) { return __double_as_longlong(v) == (long long)resid_carry_bits(); }

// KF = kfvo::TryVelRot (src/mtracklib/kfvo.cpp:1389-1668), the key-frame flavour of the same evaluation: the KeyLines of the
// current frame (kl_old here) against the field of a key frame (kl_new).  It differs from global_tracker::TryVelRot in four
// places: the match-count gate ignores FrameCount (:1441), the gradient is not z-rotated (:1476-1478 are commented out),
// the match test is Calc_f_J_Complete (global_tracker.cpp:116-165: angle, modulus ratio and inverse-depth consistency
// against the matched KeyLine, with the transformed inverse depth divided by the scale ratio Kr), and the gate radius is
// the field's own (gt.getMaxSRadius()).
#ifndef EDGEHIP_TVR_MFMA
#define EDGEHIP_TVR_MFMA 0   // 1: the 28 sums as a Gram matrix on the f64 matrix core (measured 4 % slower on gfx950, see tvr_body)
#endif
// Weight and uncertainty scaling of the residual and the Jacobian row.
//   1: the reference's own sequence of roundings (global_tracker.cpp:370-372, 399-404, 452-463): weight = k / |r| by division, applied
//      to the residual and the gradient BEFORE the Jacobian row is formed, q_rho = sqrt(s_rho qvel s_rho qvel + 1) from the weighted
//      gradient, seven quotients by q_rho (div_rn below: all seven share one reciprocal and still round like IEEE division).
//   0: one scale factor w / q_rho = rsqrt((s_rho qvel)^2 + 1 / w^2) and seven products: the same real number, an ulp or two away
//      in each value, ~10 % less time in the evaluation.  Measured side by side (profiles/r04_n_rounding_order.txt): the ulp is enough to
//      decide WHICH knife-edge frames tip a free-running sequence away from the reference (TUM leg: 2 of 33 sequences leave with 0, none
//      with 1; default batch: one transient departure more with 1), and inside tolerance 1 stays 1e5 times closer on the TUM leg.
#ifndef EDGEHIP_TVR_REF_ORDER
#define EDGEHIP_TVR_REF_ORDER 1
#endif
// (div_rn, div_mid / inv_mid / rcp_nr / MidDivisor, sqrt_ge1, sqrtf_mid, div2_mid_f32 — the division and square-root sequences without their exponent
// scaling — are in ctx.h: the one-kernel stage A's plane fit and the matcher use them too, and tools/experiments/mid_range_ops_check.hip compares each of
// them with the compiler's own operation bit for bit)
#ifndef EDGEHIP_TVR_ABL
#define EDGEHIP_TVR_ABL 0   // timing experiments only (tools/experiments/exp_tvr_ablate.sh): 1 no cross-lane reduction, 2 no div/sqrt,
#endif                      // 4 no matched-KeyLine gather, 8 no field gather, 16 no residual stream
template <bool REWEIGHT, bool PROCJF, bool GREC, bool KF = false>
__device__ __forceinline__ void tvr_body(const TvrArgs &a, const int seq, const int blk, const int tid) {
    constexpr int ABL = KF ? 0 : EDGEHIP_TVR_ABL;
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * kTvrBlock >= kn) return;  // whole block beyond the list (block-uniform)
    const KlSoA &ko = a.kl_old[seq];
    const int res_in = sq->res_cur, res_out = sq->lm_phase == 0 ? sq->res_t : sq->res_new;
    const double *rin = a.resid + ((size_t)res_in * a.nseq + seq) * a.cap;
    double *rout = a.resid + ((size_t)res_out * a.nseq + seq) * a.cap;
    const double carry_in_prev = a.resid_carry[((size_t)res_in * a.nseq + seq) * a.nblk + blk];
    const double marker = __longlong_as_double((long long)resid_carry_bits());

    // A block owns kTvrBlock consecutive KeyLines and walks them in kTvrPasses passes of kTvrThreads (pass p, thread
    // t -> KeyLine blk*kTvrBlock + p*kTvrThreads + t: coalesced, and KeyLine order = (pass, thread) order).  The 28
    // products of every pass accumulate in registers; the cross-lane reduction runs once per block instead of once
    // per KeyLine.
    constexpr int NW = kTvrThreads / 64;   // waves per block
    constexpr bool GRAM_MFMA = PROCJF && kTvrPasses == 1 && EDGEHIP_TVR_MFMA && !(ABL & 1);
    __shared__ double s_rows[GRAM_MFMA ? NW : 1][GRAM_MFMA ? 64 : 1][8];
    __shared__ double s_wlast[kTvrPasses][NW];
    __shared__ int s_whas[kTvrPasses][NW];
    double sums[kNumSums];
#pragma unroll
    for (int i = 0; i < kNumSums; i++) sums[i] = 0;

#pragma unroll 1
    for (int pass = 0; pass < kTvrPasses; pass++) {
        const int base = blk * kTvrBlock + pass * kTvrThreads;
        if (base >= kn) break;               // block-uniform
        const int ikl = base + tid;
        double J[6] = {0, 0, 0, 0, 0, 0};
        double fm, dfx, dfy;   // (set where the evaluation begins: `if (!skip)`)
        // What only an evaluated KeyLine (status != 0) reads further down is left without a default: a default is a register move at every
        // level of the nest below (the compiler materialised 44 v_mov_b64 for them, 8 % of the evaluation's vector instructions), and the
        // Jacobian section now runs for evaluated KeyLines only (a skipped one contributed exact zeros: J = 0, fm = 0 stand for it).
        double ptx, pty, ptz, pix, piy, rho_p, s_rho;
#if EDGEHIP_TVR_REF_ORDER
        double wgt_ref;      // the Huber weight (REWEIGHT)
#else
        double inv_w2 = 1;   // 1 / weight^2 (REWEIGHT)
#endif
        int mid_f = -1;
        // status: 0 = skipped (no residual written), 1 = out of image (max_r), 2 = evaluated & matched (own fi),
        //         3 = evaluated, unmatched (inherits the previous valid fi)
        int status = 0;
        double fi = 0;        // (every lane hands it to the shuffles below: it keeps its default)
        double rho_own;       // the KeyLine's rho, kept for the key post of the last evaluation (no load behind the residual store); read with mid_f >= 0 only
        if (ikl < kn) {
            // Everything the KeyLine streams in is requested here, before the first use: the skip test, the projection, the
            // in-image test and the two gathers are a chain of dependent memory round trips, and with the loads inside the
            // branches each level of the chain paid its own.  (The ~10 % of KeyLines the tests drop load 36 B in vain.)
            s_rho = ldg(ko.s_rho, ikl);
            const int32_t mnum = ldg(ko.m_num, ikl);
            const float2 pm0 = ldg(ko.p_m, ikl);
            const double rho0 = ldg(ko.rho, ikl);
            rho_own = rho0;
            const float2 klm = ldg(ko.m_m, ikl);
            const float knm = ldg(ko.n_m, ikl);
            double rprev = 0;
            if (REWEIGHT && !(ABL & 16)) rprev = rin[ikl];
            const uint32_t fc = KF ? 0xFFFFFFFFu : a.framecount[seq];
            const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
            const bool skip = s_rho > sq->s_rho_min_eval || (uint32_t)mnum < mthr;  // int vs uint compare
            if (!skip) {
                // KltoI3PMatrix + ProyI3Pto3PMatrix (global_tracker.cpp:553-570, ne10wrapper.h:414-424): P0 = (x z / zf, y z / zf, z),
                // z = 1 / rho — from p_m and rho (16 B) instead of a stored P0 (24 B): one fp64 division for a third less traffic
                const double sz = inv_mid(rho0);
                const double pz_zf0 = a.inv_zfm * sz;
                const double sx = pz_zf0 * (double)pm0.x, sy = pz_zf0 * (double)pm0.y;
                const double *R = sq->Rt, *V = sq->Vt;
                // Ne10::SE3on3PMatrix: dst = R(i,0)*x; dst += R(i,1)*y; dst += R(i,2)*z; dst = V + dst
                ptx = R[0] * sx; ptx += R[1] * sy; ptx += R[2] * sz; ptx = V[0] + ptx;
                pty = R[3] * sx; pty += R[4] * sy; pty += R[5] * sz; pty = V[1] + pty;
                ptz = R[6] * sx; ptz += R[7] * sy; ptz += R[8] * sz; ptz = V[2] + ptz;
                // Ne10::ProyP3toI3PMatrix
                rho_p = inv_mid(ptz);
                const double pz_zf = a.zfm * rho_p;
                pix = pz_zf * ptx;
                piy = pz_zf * pty;
                const double px = pix + (double)a.ppx, py = piy + (double)a.ppy;  // cam_model::Hom2Img
                const int x = cvt_trunc_sat_i32(px + 0.5), y = cvt_trunc_sat_i32(py + 0.5);   // (only the in-image test below looks at an out-of-range value)
                // Huber weight k / |r| of the previous iteration's residual (global_tracker.cpp:370-372).  It multiplies the
                // residual and the gradient, the uncertainty scaling q_rho = sqrt((s_rho w qvel)^2 + 1) divides them again
                // (:452-463): together w / q_rho = 1 / sqrt((s_rho qvel)^2 + 1 / w^2), what EDGEHIP_TVR_REF_ORDER 0 computes.
                dfx = 0; dfy = 0;
#if EDGEHIP_TVR_REF_ORDER
                wgt_ref = 1;
#endif
                if (REWEIGHT) {
                    if (is_carry(rprev)) rprev = carry_in_prev;
#if EDGEHIP_TVR_REF_ORDER
                    if (fabs(rprev) > a.k_huber) wgt_ref = div_mid(a.k_huber, fabs(rprev));
#else
                    if (fabs(rprev) > a.k_huber && !(ABL & 2)) { const double rk = fabs(rprev) * a.inv_k_huber; inv_w2 = rk * rk; }
#endif
                }
                fm = a.max_r;
                if (x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1) {
                    status = 1;
                } else {
                    status = 3;
                    // temporarily z-rotated gradient, stored back into a float Point2DF (:386-388)
                    const float rmx = KF ? klm.x : (float)(sq->RM[0] * (double)klm.x + sq->RM[1] * (double)klm.y);
                    const float rmy = KF ? klm.y : (float)(sq->RM[2] * (double)klm.x + sq->RM[3] * (double)klm.y);
                    const uint32_t f = (ABL & 8) ? (uint32_t)(ikl + 1) : a.field16[(size_t)seq * a.f16stride + field16_index(x, y, a.f16tx)];
                    if (KF) {
                        if (f != 0u) {
                            const int ikf = (int)f - 1;
                            const KlSoA &kf = a.kl_new[seq];
                            const MatchRec fr = kf.rec[ikf];
                            const double f_rho = kf.rho[ikf], f_srho = kf.s_rho[ikf];
                            // Calc_f_J_Complete (global_tracker.cpp:138-147): float products and quotients, compared in double
                            const double cang = (double)((klm.x * fr.m_mx + klm.y * fr.m_my) / knm);
                            const double rho_t = rho_p / a.kf[seq].Kr;          // PtIm[2*pnum+ikl]/Kr (kfvo.cpp:1481)
                            const bool reject = cang < a.kf_match_cang || (double)fabsf(knm / fr.n_m - 1) > a.kf_match_mod ||
                                                fabs(rho_t - f_rho) > a.kf_rho_tol * (f_srho + s_rho * rho_t / rho0);
                            if (!reject) {
                                const double dx = px - (double)fr.c_px, dy = py - (double)fr.c_py;
                                fi = dx * (double)fr.u_mx + dy * (double)fr.u_my;
                                dfx = (double)fr.u_mx;
                                dfy = (double)fr.u_my;
                                fm = fi;
                                mid_f = ikf;
                                status = 2;
                            }
                        }
                    } else {
                        // (an empty field entry reads record 0 and fails the test below by `f != 0`: one level of the nest less — each level costs
                        // the moves of everything it may leave unchanged — and record 0's line is the same for every such lane)
                        const int ikf = f != 0u ? (int)f - 1 : 0;
                        // The matched KeyLine's c_p, m_m, u_m.  GREC: a 16-byte record (four records share the 64 bytes
                        // a random gather moves, instead of two) and u_m recomputed with the detector's own float
                        // expressions (k_emit; edge_finder.cpp:166-200), valid for KeyLines nothing has rotated since.
                        float f_cpx, f_cpy, f_mx, f_my, f_ux, f_uy;
                        if (ABL & 4) {
                            f_cpx = (float)px; f_cpy = (float)py; f_mx = klm.x; f_my = klm.y; f_ux = 1.f; f_uy = 0.f;
                        } else if (GREC) {
                            const float4 g = ldg(a.kl_new[seq].grec, ikf);
                            f_cpx = g.x; f_cpy = g.y; f_mx = g.z; f_my = g.w;
                        } else {
                            const MatchRec fr = a.kl_new[seq].rec[ikf];
                            f_cpx = fr.c_px; f_cpy = fr.c_py; f_mx = fr.m_mx; f_my = fr.m_my; f_ux = fr.u_mx; f_uy = fr.u_my;
                        }
                        // Test_f_k (float arithmetic inside, compared in double)
                        const double p_n2 = (double)(knm * knm);
                        const double p_esc = (double)(rmx * f_mx + rmy * f_my);
                        if ((f != 0u) & !(fabs(p_esc - p_n2) > a.match_thresh * p_n2)) {   // (`&`: one level, the record's load is not conditional)
                            if (GREC) {
                                const float n2m = f_mx * f_mx + f_my * f_my;
                                const float nm = sqrtf_mid(n2m);
                                div2_mid_f32(f_mx, f_my, nm, f_ux, f_uy);
                            }
                            const double dx = px - (double)f_cpx, dy = py - (double)f_cpy;
                            fi = dx * (double)f_ux + dy * (double)f_uy;
                            dfx = (double)f_ux;
                            dfy = (double)f_uy;
                            fm = fi;
                            mid_f = ikf;
                            status = 2;
                        }
                    }
                }
            }
        }

        // ---- DResidualNew: "last valid fi" propagation (KeyLine order = pass, wave, lane) ----
        {
            const unsigned long long vmask = __ballot(status == 2);
            const unsigned long long below = vmask & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll(below) : 0;
            const double inh = __shfl(fi, src, 64);
            const int top = vmask ? 63 - __clzll(vmask) : 0;
            const double wl = __shfl(fi, top, 64);
            if (lane == 0) {
                s_whas[pass][wave] = vmask != 0;
                s_wlast[pass][wave] = wl;
            }
            __syncthreads();   // entries of earlier passes were published by earlier barriers
            if (status == 3) {
                double v = marker;   // no valid KeyLine before this one inside the block: resolved from the block carries
                bool have = false;
                if (below) { v = inh; have = true; }
                for (int pp = pass; pp >= 0 && !have; pp--)
                    for (int pw = (pp == pass ? wave - 1 : NW - 1); pw >= 0 && !have; pw--)
                        if (s_whas[pp][pw]) { v = s_wlast[pp][pw]; have = true; }
                rout[ikl] = v;
            } else if (status == 2) {
                rout[ikl] = fi;
            } else if (status == 1) {
                rout[ikl] = a.max_r;
            } else if (ikl < kn) {
                // a KeyLine the gates skip (the same ones in every evaluation of a minimisation): the reference leaves its entry
                // alone and nothing ever reads it; it is written anyway so that the wave stores whole lines (with ~13 % of the
                // lanes masked most lines would be written in part, which the memory system turns into read-modify-write)
                rout[ikl] = 0.0;
            }
        }
        if (a.write_mid && ikl < kn) {
            stg(ko.m_id_f, ikl, mid_f);   // (a global store: the FLAT one also counts on the LDS counter, ctx.h)
            if (!KF && a.fwd_key && mid_f >= 0) atomicMax(&a.fwd_key[(size_t)seq * a.cap + mid_f], ord_bits(rho_own));
        }

        // ---- Jacobian row, uncertainty scaling (global_tracker.cpp:419-463) ----
        double fs = 0;   // the scaled residual: the row's seventh value (zero, like J, for a KeyLine that was not evaluated)
        if (status != 0) {
#if EDGEHIP_TVR_REF_ORDER
            if (REWEIGHT) { fm *= wgt_ref; dfx *= wgt_ref; dfy *= wgt_ref; }
#endif
            if (PROCJF) {
                double t0 = a.zfm * rho_p;
                J[0] = t0 * dfx;
                J[1] = t0 * dfy;
                t0 = rho_p * pix;
                J[2] = t0 * dfx;
                t0 = rho_p * piy;
                J[2] += t0 * dfy;
                J[3] = J[1] * ptz; J[3] += J[2] * pty;
                J[4] = J[0] * ptz; J[4] += J[2] * ptx;
                t0 = J[0] * pty;
                J[5] = -1 * t0; J[5] += J[1] * ptx;
            }
            const double qvel = (a.zfm * dfx * sq->Vt[0] + a.zfm * dfy * sq->Vt[1] + (pix * dfx + piy * dfy) * sq->Vt[2]);
            // the reference divides the seven values by q_rho one by one (EDGEHIP_TVR_REF_ORDER above; with 0 qvel is the unweighted one)
#if EDGEHIP_TVR_REF_ORDER
            const double q_rho = REWEIGHT ? sqrt_ge1(s_rho * qvel * s_rho * qvel + 1) : s_rho;
            const double r_q = rcp_for_div_rn(q_rho);
            if (PROCJF) {
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] = div_rn(J[j], q_rho, r_q);
            }
            fs = div_rn(fm, q_rho, r_q);
#else
            double inv_q;
            if (REWEIGHT) {
                const double sq = s_rho * qvel;
                inv_q = (ABL & 2) ? sq + inv_w2 : rsqrt_f64(sq * sq + inv_w2);
            } else {
                inv_q = 1.0 / s_rho;
            }
            if (PROCJF) {
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] *= inv_q;
            }
            fs = fm * inv_q;
#endif
        }
        if (GRAM_MFMA) {
            // the KeyLine's row (J0..J5, fm, 0) for the Gram matrix below; rows of skipped / absent KeyLines are zero
            double2 *row = reinterpret_cast<double2 *>(&s_rows[wave][lane][0]);
            row[0] = make_double2(J[0], J[1]);
            row[1] = make_double2(J[2], J[3]);
            row[2] = make_double2(J[4], J[5]);
            row[3] = make_double2(fs, 0.0);
        } else {
            int ns = 0;
            if (PROCJF) {
#pragma unroll
                // (one KeyLine per thread: the product IS the lane's term — `0.0 + product` would cost an add per sum, 28 of the evaluation's ~600
                // vector instructions, to turn a -0.0 product into +0.0, which the reference's PairWiseVAdd over the bare products does not do either)
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = i; j < 6; j++) { const double pr = J[i] * J[j]; sums[ns] = kTvrPasses == 1 ? pr : sums[ns] + pr; ns++; }
#pragma unroll
                for (int i = 0; i < 6; i++) { const double pr = J[i] * fs; sums[ns] = kTvrPasses == 1 ? pr : sums[ns] + pr; ns++; }
            }
            { const double pr = fs * fs; const int at = PROCJF ? ns : kNumSums - 1; sums[at] = kTvrPasses == 1 ? pr : sums[at] + pr; }
        }
    }

    // last valid fi of the whole block (marker if none), for the carries of the following blocks
    if (tid == 0) {
        double bl = marker;
        for (int pp = kTvrPasses - 1; pp >= 0 && is_carry(bl); pp--) {
            if (blk * kTvrBlock + pp * kTvrThreads >= kn) continue;
            for (int pw = NW - 1; pw >= 0; pw--)
                if (s_whas[pp][pw]) { bl = s_wlast[pp][pw]; break; }
        }
        a.block_last[(size_t)seq * a.nblk + blk] = bl;
    }

    // ---- block reduction: transposed (halving) wave reduction, LDS across waves, one partial per block ----
    __shared__ double s_red[NW][32];
    if (GRAM_MFMA) {
        // J^T J, J^T f and f^T f of the wave's 64 KeyLines are the Gram matrix A^T A of its 64 x 7 rows: 16 issues of
        // v_mfma_f64_16x16x4_f64 (k = 4 KeyLines each; A and B operand are the same register: lane (m, kk) holds element m of
        // KeyLine 4s + kk, zero for m >= 7) instead of 27 products per lane and a 28-value cross-lane reduction (~300 of the
        // kernel's ~1000 instructions).  Deterministic (fixed k order), sums within rounding of the butterfly's, all parity
        // tests pass — and 4 % SLOWER (3300 -> 3440 us per 12 evaluations, same box): gfx950 runs f64 MFMA at the vector
        // rate (32 flop/cycle/SIMD), a 16x16x4 issue occupies the matrix pipe for 64 cycles, and only 7 x 7 of its 16 x 16
        // outputs are wanted: 1024 pipe cycles per wave, more than the vector instructions it replaces.  Off by default.
        typedef double v4d __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int m = (lane & 15) < 7 ? (lane & 15) : 7, kk = lane >> 4;
        const double *src = &s_rows[wave][kk][m];
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 16; st++) {
            const double v = src[st * 4 * 8];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
        }
        // D(i, j): column j = lane & 15, row i = (lane >> 4) + 4 * reg   (f64 16x16x4 accumulator layout)
        const int j = lane & 15;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = (lane >> 4) + 4 * r;
            if (j < 7 && i <= j) {
                const int idx = j < 6 ? i * 6 - (i * (i - 1)) / 2 + (j - i) : (i < 6 ? 21 + i : 27);
                s_red[wave][idx] = acc[r];
            }
        }
    } else if (PROCJF && (ABL & 1)) {
        if (lane < kNumSums) s_red[wave][lane] = sums[lane % 4];
    } else if (PROCJF) {
        const int idx = wave_reduce28(sums, lane);
        if ((lane & 1) == 0) s_red[wave][idx] = sums[0];
    } else {
        double v = sums[kNumSums - 1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_red[wave][kNumSums - 1] = v;
    }
    __syncthreads();
    if (PROCJF ? tid < kNumSums : tid == kNumSums - 1) {
        double v = s_red[0][tid];
#pragma unroll
        for (int wv = 1; wv < NW; wv++) v += s_red[wv][tid];   // fixed order: deterministic
        a.partials[((size_t)seq * a.nblk + blk) * kNumSums + tid] = v;
    }
}

template <bool REWEIGHT, bool PROCJF, bool GREC>
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot(TvrArgs a) {
    tvr_body<REWEIGHT, PROCJF, GREC>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}
// kfvo::TryVelRot<double, true, true, false>: the only instantiation Minimizer_RV_KF uses (kfvo.cpp:1744, 1772)
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot_kf(TvrArgs a) {
    tvr_body<true, true, false, true>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// global_tracker::TryVelRot<float, ReWeight, ProcJF, false> — the reference's OTHER instantiation of the tracker
// (global_tracker.cpp:824: Minimizer_RV<float>, selected by USE_NE10 at rebvo_second_t.cpp:339-343; NEON only does float).  Round 6,
// a second configuration next to the fp64 one (edgehip_set_tracker_precision(ctx, 32)); the headline stays fp64.
// Everything the reference declares as T is a float here and rounds where the reference's float code rounds: P0 = ((1/zf) z) x with
// z = 1 / (float)rho (KltoI3PMatrix<float>, ProyI3Pto3PMatrix<float>: ne10wrapper.h:414-424), the SE(3) transform and the
// projection as float multiply / add chains (SE3on3PMatrix, ProyP3toI3PMatrix), Hom2Img, the z-rotated gradient, Test_f_k<float>,
// the residual and its gradient (Calc_f_J2<float>), the Huber weight, the Jacobian row (MulVect / MlAcVect chains); what the
// reference keeps in double stays double: the uncertainty gate, round2int_positive's + 0.5, q_rho = sqrt(s_rho qvel s_rho qvel + 1)
// and the seven quotients by it (float / double, rounded back to float).  The 28 products are float and are summed in float in the
// fixed halving tree of wave_reduce.h (the reference: PairWiseVAdd<float>) — float sums of ~14 k terms agree to ~1e-6 relative, not
// bit for bit, which is what the float tolerance of the tests states.  The residual memory keeps its fp64 buffers (float values,
// exactly representable): the marker / carry logic of the "last valid fi" propagation is shared with the fp64 kernels.
// ---------------------------------------------------------------------------------------------------
template <bool REWEIGHT, bool PROCJF, bool GREC>
__device__ __forceinline__ void tvr_body_f32(const TvrArgs &a, const int seq, const int blk, const int tid) {
    static_assert(kTvrPasses == 1, "one KeyLine per thread");
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * kTvrBlock >= kn) return;
    const KlSoA &ko = a.kl_old[seq];
    const int res_in = sq->res_cur, res_out = sq->lm_phase == 0 ? sq->res_t : sq->res_new;
    const double *rin = a.resid + ((size_t)res_in * a.nseq + seq) * a.cap;
    double *rout = a.resid + ((size_t)res_out * a.nseq + seq) * a.cap;
    const double carry_in_prev = a.resid_carry[((size_t)res_in * a.nseq + seq) * a.nblk + blk];
    const double marker = __longlong_as_double((long long)resid_carry_bits());
    constexpr int NW = kTvrThreads / 64;
    __shared__ float s_wlast[NW];
    __shared__ int s_whas[NW];
    float sums[kNumSums];
#pragma unroll
    for (int i = 0; i < kNumSums; i++) sums[i] = 0.f;

    const int ikl = blk * kTvrBlock + tid;
    float J[6] = {0, 0, 0, 0, 0, 0};
    // (no defaults for what only an evaluated KeyLine — status != 0 — reads in the Jacobian section: see tvr_body)
    float fm, dfx, dfy;
    float ptx, pty, ptz, pix, piy, rho_p;
    double s_rho;
    float wgt;
    int mid_f = -1, status = 0;
    float fi = 0;
    double rho_own;
    const float zf = (float)a.zfm, max_r = (float)a.max_r, k_huber = (float)a.k_huber, simil_t = (float)a.match_thresh;
    float Vt0, Vt1, Vt2;
    if (ikl < kn) {
        s_rho = ldg(ko.s_rho, ikl);
        const int32_t mnum = ldg(ko.m_num, ikl);
        const float2 pm0 = ldg(ko.p_m, ikl);
        const double rho0 = ldg(ko.rho, ikl);
        rho_own = rho0;
        const float2 klm = ldg(ko.m_m, ikl);
        const float knm = ldg(ko.n_m, ikl);
        double rprev_d = 0;
        if (REWEIGHT) rprev_d = rin[ikl];
        const uint32_t fc = a.framecount[seq];
        const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
        const bool skip = s_rho > sq->s_rho_min_eval || (uint32_t)mnum < mthr;
        Vt0 = (float)sq->Vt[0]; Vt1 = (float)sq->Vt[1]; Vt2 = (float)sq->Vt[2];
        if (!skip) {
            const float sz = 1.f / (float)rho0;
            const float pz_zf0 = (1.f / zf) * sz;
            const float sx = pz_zf0 * pm0.x, sy = pz_zf0 * pm0.y;
            const double *Rd = sq->Rt;
            const float R0 = (float)Rd[0], R1 = (float)Rd[1], R2 = (float)Rd[2], R3 = (float)Rd[3], R4 = (float)Rd[4], R5 = (float)Rd[5],
                        R6 = (float)Rd[6], R7 = (float)Rd[7], R8 = (float)Rd[8];
            ptx = R0 * sx; ptx += R1 * sy; ptx += R2 * sz; ptx = Vt0 + ptx;
            pty = R3 * sx; pty += R4 * sy; pty += R5 * sz; pty = Vt1 + pty;
            ptz = R6 * sx; ptz += R7 * sy; ptz += R8 * sz; ptz = Vt2 + ptz;
            rho_p = 1.f / ptz;
            const float pz_zf = zf * rho_p;
            pix = pz_zf * ptx;
            piy = pz_zf * pty;
            const float px = pix + a.ppx, py = piy + a.ppy;
            const int x = cvt_trunc_sat_i32((double)px + 0.5), y = cvt_trunc_sat_i32((double)py + 0.5);   // (read by the in-image test only)
            dfx = 0; dfy = 0; wgt = 1;
            if (REWEIGHT) {
                if (is_carry(rprev_d)) rprev_d = carry_in_prev;
                const float rprev = (float)rprev_d;
                if (fabsf(rprev) > k_huber) wgt = k_huber / fabsf(rprev);
            }
            fm = max_r;
            if (x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1) {
                status = 1;
            } else {
                status = 3;
                const float rmx = (float)sq->RM[0] * klm.x + (float)sq->RM[1] * klm.y;   // Matrix<2,2,float> RM (global_tracker.cpp:318, 386-388)
                const float rmy = (float)sq->RM[2] * klm.x + (float)sq->RM[3] * klm.y;
                const uint32_t f = a.field16[(size_t)seq * a.f16stride + field16_index(x, y, a.f16tx)];
                {
                    const int ikf = f != 0u ? (int)f - 1 : 0;   // (an empty entry reads record 0 and fails by `f != 0` below: tvr_body)
                    float f_cpx, f_cpy, f_mx, f_my, f_ux, f_uy;
                    if (GREC) {
                        const float4 g = ldg(a.kl_new[seq].grec, ikf);
                        f_cpx = g.x; f_cpy = g.y; f_mx = g.z; f_my = g.w;
                    } else {
                        const MatchRec fr = a.kl_new[seq].rec[ikf];
                        f_cpx = fr.c_px; f_cpy = fr.c_py; f_mx = fr.m_mx; f_my = fr.m_my; f_ux = fr.u_mx; f_uy = fr.u_my;
                    }
                    const float p_n2 = knm * knm;                      // Test_f_k<float> (global_tracker.h:90-104)
                    const float p_esc = rmx * f_mx + rmy * f_my;
                    if ((f != 0u) & !(fabsf(p_esc - p_n2) > simil_t * p_n2)) {
                        if (GREC) {
                            const float n2m = f_mx * f_mx + f_my * f_my;
                            const float nm = sqrtf_mid(n2m);
                            div2_mid_f32(f_mx, f_my, nm, f_ux, f_uy);
                        }
                        const float dx = px - f_cpx, dy = py - f_cpy;    // Calc_f_J2<float> (global_tracker.cpp:228-271)
                        fi = dx * f_ux + dy * f_uy;
                        dfx = f_ux;
                        dfy = f_uy;
                        fm = fi;
                        mid_f = ikf;
                        status = 2;
                    }
                }
            }
        }
    }
    // ---- DResidualNew: "last valid fi" propagation (KeyLine order = wave, lane), as in tvr_body ----
    {
        const unsigned long long vmask = __ballot(status == 2);
        const unsigned long long below = vmask & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __clzll(below) : 0;
        const float inh = __shfl(fi, src, 64);
        const int top = vmask ? 63 - __clzll(vmask) : 0;
        const float wl = __shfl(fi, top, 64);
        if (lane == 0) {
            s_whas[wave] = vmask != 0;
            s_wlast[wave] = wl;
        }
        __syncthreads();
        if (status == 3) {
            double v = marker;
            bool have = false;
            if (below) { v = (double)inh; have = true; }
            for (int pw = wave - 1; pw >= 0 && !have; pw--)
                if (s_whas[pw]) { v = (double)s_wlast[pw]; have = true; }
            rout[ikl] = v;
        } else if (status == 2) {
            rout[ikl] = (double)fi;
        } else if (status == 1) {
            rout[ikl] = (double)max_r;
        } else if (ikl < kn) {
            rout[ikl] = 0.0;
        }
    }
    if (a.write_mid && ikl < kn) {
        stg(ko.m_id_f, ikl, mid_f);
        if (a.fwd_key && mid_f >= 0) atomicMax(&a.fwd_key[(size_t)seq * a.cap + mid_f], ord_bits(rho_own));
    }
    // ---- Jacobian row, uncertainty scaling (global_tracker.cpp:399-463 with T = float) ----
    float fs = 0.f;   // the scaled residual (zero, like J, for a KeyLine that was not evaluated)
    if (status != 0) {
        if (REWEIGHT) { fm *= wgt; dfx *= wgt; dfy *= wgt; }
        if (PROCJF) {
            float t0 = zf * rho_p;
            J[0] = t0 * dfx;
            J[1] = t0 * dfy;
            t0 = rho_p * pix;
            J[2] = t0 * dfx;
            t0 = rho_p * piy;
            J[2] += t0 * dfy;
            J[3] = J[1] * ptz; J[3] += J[2] * pty;
            J[4] = J[0] * ptz; J[4] += J[2] * ptx;
            t0 = J[0] * pty;
            J[5] = -1.f * t0; J[5] += J[1] * ptx;
        }
        // qvel: a float expression (zfm, the derivatives, PtIm and Vt are floats) assigned to a double (:452-453); q_rho in double
        const double qvel = (double)(zf * dfx * Vt0 + zf * dfy * Vt1 + (pix * dfx + piy * dfy) * Vt2);
        const double q_rho = REWEIGHT ? sqrt_ge1(s_rho * qvel * s_rho * qvel + 1) : s_rho;
        const double r_q = rcp_for_div_rn(q_rho);
        if (PROCJF) {
#pragma unroll
            for (int j = 0; j < 6; j++) J[j] = (float)div_rn((double)J[j], q_rho, r_q);   // float /= double: the quotient in double, rounded to float
        }
        fs = (float)div_rn((double)fm, q_rho, r_q);
    }
    {
        int ns = 0;
        if (PROCJF) {
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) sums[ns++] = J[i] * J[j];
#pragma unroll
            for (int i = 0; i < 6; i++) sums[ns++] = J[i] * fs;
        }
        sums[PROCJF ? ns : kNumSums - 1] = fs * fs;
    }
    if (tid == 0) {
        double bl = marker;
        for (int pw = NW - 1; pw >= 0; pw--)
            if (s_whas[pw]) { bl = (double)s_wlast[pw]; break; }
        a.block_last[(size_t)seq * a.nblk + blk] = bl;
    }
    __shared__ float s_red[NW][32];
    if (PROCJF) {
        const int idx = wave_reduce28_f32(sums, lane);
        if ((lane & 1) == 0) s_red[wave][idx] = sums[0];
    } else {
        float v = sums[kNumSums - 1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_red[wave][kNumSums - 1] = v;
    }
    __syncthreads();
    if (PROCJF ? tid < kNumSums : tid == kNumSums - 1) {
        float v = s_red[0][tid];
#pragma unroll
        for (int wv = 1; wv < NW; wv++) v += s_red[wv][tid];
        a.partials[((size_t)seq * a.nblk + blk) * kNumSums + tid] = (double)v;   // the step kernel adds the blocks' sums (in double) and rounds to float
    }
}
template <bool REWEIGHT, bool PROCJF, bool GREC>
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot_f32(TvrArgs a) {
    tvr_body_f32<REWEIGHT, PROCJF, GREC>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// Two evaluations in one launch: TrackerInitType = 2 runs `init_iter + 1` un-reweighted evaluations from X = 0 and the same
// number from the prior (Vel, W0) and keeps the better end point (global_tracker.cpp:649-692, 698-738, 740-749).  The two
// chains are independent: neither reads a residual buffer (ReWeight = false), they write different ones (Rest /
// ResidualNew), and `FrameCount`, the uncertainty gate and the KeyLines are the same for both.  So evaluation i of the
// zero-init chain and evaluation i of the prior-init chain go out as ONE launch: the KeyLine's streams (s_rho, m_num, p_m,
// rho, m_m, n_m: 40 of the evaluation's 84 bytes) are loaded once, P0 is rebuilt once, and the two transforms walk the two
// dependent gathers side by side (both field reads in flight together, then both record reads) — 12 dependent launches
// become 9.  Per chain the arithmetic is tvr_body<false, PROCJF, GREC>'s, expression for expression, and each chain keeps
// its own partial sums / last-residual rows in the same block order: results identical bit for bit to the launch chain
// (tests/test_pipeline_gpu.py::test_two_chain_evaluation_is_bit_identical_to_the_launch_chain).
// Chain 0 = zero-init (transform zRt / zVt / zRM, residuals -> res_t, sums -> partials_z), chain 1 = prior-init (Rt / Vt /
// RM, residuals -> res_new, sums -> partials).
// ---------------------------------------------------------------------------------------------------
#ifndef EDGEHIP_TVR2_PARK
#define EDGEHIP_TVR2_PARK 1   // the second chain's Jacobian row waits in LDS while the first chain's products are reduced
#endif
template <bool PROCJF, bool GREC>
__device__ __forceinline__ void tvr2_body(const TvrArgs &a, const int seq, const int blk, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * kTvrBlock >= kn) return;  // whole block beyond the list (block-uniform)
    static_assert(kTvrPasses == 1, "one KeyLine per thread");
    const KlSoA &ko = a.kl_old[seq];
    double *rout0 = a.resid + ((size_t)sq->res_t * a.nseq + seq) * a.cap;
    double *rout1 = a.resid + ((size_t)sq->res_new * a.nseq + seq) * a.cap;
    const double marker = __longlong_as_double((long long)resid_carry_bits());
    constexpr int NW = kTvrThreads / 64;
    __shared__ double s_wlast[2][NW];
    __shared__ int s_whas[2][NW];

    const int ikl = blk * kTvrBlock + tid;
    // per chain: what tvr_body keeps per KeyLine
    // (no defaults for what only an evaluated KeyLine — status != 0 — reads in the Jacobian section: see tvr_body)
    double fm[2], dfx[2], dfy[2], fi[2] = {0, 0};
    double ptx[2], pty[2], ptz[2], pix[2], piy[2], rho_p[2];
    int status[2] = {0, 0};   // 0 skipped, 1 out of image, 2 matched, 3 evaluated and unmatched (tvr_body)
    double s_rho = 1;
    if (ikl < kn) {
        s_rho = ldg(ko.s_rho, ikl);
        const int32_t mnum = ldg(ko.m_num, ikl);
        const float2 pm0 = ldg(ko.p_m, ikl);
        const double rho0 = ldg(ko.rho, ikl);
        const float2 klm = ldg(ko.m_m, ikl);
        const float knm = ldg(ko.n_m, ikl);
        const uint32_t fc = a.framecount[seq];
        const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
        const bool skip = s_rho > sq->s_rho_min_eval || (uint32_t)mnum < mthr;  // int vs uint compare
        if (!skip) {
            const double sz = inv_mid(rho0);
            const double pz_zf0 = a.inv_zfm * sz;
            const double sx = pz_zf0 * (double)pm0.x, sy = pz_zf0 * (double)pm0.y;
            double px[2], py[2];
            bool inimg[2];
            size_t fidx[2];
            float rmx[2], rmy[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const double *R = c ? sq->Rt : sq->zRt, *V = c ? sq->Vt : sq->zVt, *RM = c ? sq->RM : sq->zRM;
                ptx[c] = R[0] * sx; ptx[c] += R[1] * sy; ptx[c] += R[2] * sz; ptx[c] = V[0] + ptx[c];
                pty[c] = R[3] * sx; pty[c] += R[4] * sy; pty[c] += R[5] * sz; pty[c] = V[1] + pty[c];
                ptz[c] = R[6] * sx; ptz[c] += R[7] * sy; ptz[c] += R[8] * sz; ptz[c] = V[2] + ptz[c];
                rho_p[c] = inv_mid(ptz[c]);
                const double pz_zf = a.zfm * rho_p[c];
                pix[c] = pz_zf * ptx[c];
                piy[c] = pz_zf * pty[c];
                px[c] = pix[c] + (double)a.ppx; py[c] = piy[c] + (double)a.ppy;
                const int x = x86_cvttsd2si(px[c] + 0.5), y = x86_cvttsd2si(py[c] + 0.5);   // (cvt_trunc_sat_i32, as tvr_body has it, measured 1 % slower here: four asm statements)
                inimg[c] = !(x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1);
                fidx[c] = inimg[c] ? field16_index(x, y, a.f16tx) : (size_t)0;
                rmx[c] = (float)(RM[0] * (double)klm.x + RM[1] * (double)klm.y);
                rmy[c] = (float)(RM[2] * (double)klm.x + RM[3] * (double)klm.y);
                fm[c] = a.max_r;
                dfx[c] = 0; dfy[c] = 0;
                status[c] = inimg[c] ? 3 : 1;
            }
            // the two gathers of the two chains, level by level: both field reads are in flight together, then both records
            // (unconditional loads: an out-of-image projection reads pixel 0, an empty pixel record 0 — results unused)
            const uint16_t *fld = a.field16 + (size_t)seq * a.f16stride;
            const uint32_t f0 = fld[fidx[0]], f1 = fld[fidx[1]];
            const bool hit0 = inimg[0] && f0 != 0u, hit1 = inimg[1] && f1 != 0u;
            const int ikf0 = hit0 ? (int)f0 - 1 : 0, ikf1 = hit1 ? (int)f1 - 1 : 0;
            float f_cpx[2], f_cpy[2], f_mx[2], f_my[2], f_ux[2] = {0, 0}, f_uy[2] = {0, 0};
            if (GREC) {
                const float4 g0 = ldg(a.kl_new[seq].grec, ikf0), g1 = ldg(a.kl_new[seq].grec, ikf1);
                f_cpx[0] = g0.x; f_cpy[0] = g0.y; f_mx[0] = g0.z; f_my[0] = g0.w;
                f_cpx[1] = g1.x; f_cpy[1] = g1.y; f_mx[1] = g1.z; f_my[1] = g1.w;
            } else {
                const MatchRec r0 = a.kl_new[seq].rec[ikf0], r1 = a.kl_new[seq].rec[ikf1];
                f_cpx[0] = r0.c_px; f_cpy[0] = r0.c_py; f_mx[0] = r0.m_mx; f_my[0] = r0.m_my; f_ux[0] = r0.u_mx; f_uy[0] = r0.u_my;
                f_cpx[1] = r1.c_px; f_cpy[1] = r1.c_py; f_mx[1] = r1.m_mx; f_my[1] = r1.m_my; f_ux[1] = r1.u_mx; f_uy[1] = r1.u_my;
            }
            const double p_n2 = (double)(knm * knm);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                {
                    // Test_f_k (float arithmetic inside, compared in double); `&`: one level of the nest for the hit and the test (tvr_body)
                    const double p_esc = (double)(rmx[c] * f_mx[c] + rmy[c] * f_my[c]);
                    if ((c ? hit1 : hit0) & !(fabs(p_esc - p_n2) > a.match_thresh * p_n2)) {
                        if (GREC) {
                            const float n2m = f_mx[c] * f_mx[c] + f_my[c] * f_my[c];
                            const float nm = sqrtf_mid(n2m);
                            div2_mid_f32(f_mx[c], f_my[c], nm, f_ux[c], f_uy[c]);
                        }
                        const double dx = px[c] - (double)f_cpx[c], dy = py[c] - (double)f_cpy[c];
                        fi[c] = dx * (double)f_ux[c] + dy * (double)f_uy[c];
                        dfx[c] = (double)f_ux[c];
                        dfy[c] = (double)f_uy[c];
                        fm[c] = fi[c];
                        status[c] = 2;
                    }
                }
            }
        }
    }

    // ---- DResidualNew of both chains: "last valid fi" propagation (KeyLine order = wave, lane), one barrier for the two ----
    unsigned long long below[2];
    double inh[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const unsigned long long vmask = __ballot(status[c] == 2);
        below[c] = vmask & ((1ull << lane) - 1ull);
        const int src = below[c] ? 63 - __clzll(below[c]) : 0;
        inh[c] = __shfl(fi[c], src, 64);
        const int top = vmask ? 63 - __clzll(vmask) : 0;
        const double wl = __shfl(fi[c], top, 64);
        if (lane == 0) {
            s_whas[c][wave] = vmask != 0;
            s_wlast[c][wave] = wl;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
        double *rout = c ? rout1 : rout0;
        if (status[c] == 3) {
            double v = marker;   // no valid KeyLine before this one inside the block: resolved from the block carries
            bool have = false;
            if (below[c]) { v = inh[c]; have = true; }
            for (int pw = wave - 1; pw >= 0 && !have; pw--)
                if (s_whas[c][pw]) { v = s_wlast[c][pw]; have = true; }
            rout[ikl] = v;
        } else if (status[c] == 2) {
            rout[ikl] = fi[c];
        } else if (status[c] == 1) {
            rout[ikl] = a.max_r;
        } else if (ikl < kn) {
            rout[ikl] = 0.0;   // a KeyLine the gates skip: whole-line stores (tvr_body)
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            double bl = marker;
            for (int pw = NW - 1; pw >= 0; pw--)
                if (s_whas[c][pw]) { bl = s_wlast[c][pw]; break; }
            (c ? a.block_last : a.block_last_z)[(size_t)seq * a.nblk + blk] = bl;
        }
    }

    // ---- Jacobian rows, uncertainty scaling, the 28 sums: chain by chain ----
    // The transposed reduction holds 28 products (56 registers) of one chain; what the other chain needs for its own row would
    // have to stay alive beside them (87 registers, 5 waves per SIMD instead of 8).  So both rows are formed first, chain 1's seven
    // values (J0..J5, f) wait in LDS ([7][256]: conflict-free) while chain 0's products are reduced, and come back for their turn.
    __shared__ double s_red[2][NW][32];
#if EDGEHIP_TVR2_PARK
    __shared__ double s_park[PROCJF ? 7 : 1][kTvrThreads];
#endif
    const double inv_q = rcp_for_div_rn(s_rho);
    double Jc[2][6], fmc[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        double *J = Jc[c];
#pragma unroll
        for (int j = 0; j < 6; j++) J[j] = 0;
        fmc[c] = 0;
        if (status[c] != 0) {   // (a skipped KeyLine's row is exact zeros)
            fmc[c] = fm[c];
            if (PROCJF) {
                double t0 = a.zfm * rho_p[c];
                J[0] = t0 * dfx[c];
                J[1] = t0 * dfy[c];
                t0 = rho_p[c] * pix[c];
                J[2] = t0 * dfx[c];
                t0 = rho_p[c] * piy[c];
                J[2] += t0 * dfy[c];
                J[3] = J[1] * ptz[c]; J[3] += J[2] * pty[c];
                J[4] = J[0] * ptz[c]; J[4] += J[2] * ptx[c];
                t0 = J[0] * pty[c];
                J[5] = -1 * t0; J[5] += J[1] * ptx[c];
#if EDGEHIP_TVR_REF_ORDER
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] = div_rn(J[j], s_rho, inv_q);   // the reference's seven quotients (see div_rn)
#else
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] *= inv_q;
#endif
            }
#if EDGEHIP_TVR_REF_ORDER
            fmc[c] = div_rn(fmc[c], s_rho, inv_q);
#else
            fmc[c] *= inv_q;
#endif
        }
    }
#if EDGEHIP_TVR2_PARK
    if (PROCJF) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_park[j][tid] = Jc[1][j];
        s_park[6][tid] = fmc[1];
    }
#endif
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (PROCJF) {
            double J[6], f;
#if EDGEHIP_TVR2_PARK
            if (c == 1) {   // own entries: no barrier needed
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] = s_park[j][tid];
                f = s_park[6][tid];
            } else
#endif
            {
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] = Jc[c][j];
                f = fmc[c];
            }
            double sums[kNumSums];
            int ns = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) sums[ns++] = J[i] * J[j];   // (the bare products, as tvr_body's)
#pragma unroll
            for (int i = 0; i < 6; i++) sums[ns++] = J[i] * f;
            sums[ns] = f * f;
            const int idx = wave_reduce28(sums, lane);
            if ((lane & 1) == 0) s_red[c][wave][idx] = sums[0];
        } else {
            double v = fmc[c] * fmc[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) s_red[c][wave][kNumSums - 1] = v;
        }
    }
    __syncthreads();
    // wave 0 finishes chain 1, wave 1 chain 0 (fixed order over the block's waves: deterministic)
    if (wave < 2) {
        const int c = 1 - wave;
        if (PROCJF ? lane < kNumSums : lane == kNumSums - 1) {
            double v = s_red[c][0][lane];
#pragma unroll
            for (int wv = 1; wv < NW; wv++) v += s_red[c][wv][lane];
            (c ? a.partials : a.partials_z)[((size_t)seq * a.nblk + blk) * kNumSums + lane] = v;
        }
    }
}

// tvr2_body for the float tracker: evaluation i of both initialisation chains of Minimizer_RV<float> in one launch — the KeyLine's
// streams loaded once, the two transforms walked side by side through the two gathers — per chain the arithmetic of tvr_body_f32<false, ...>,
// expression for expression (un-reweighted: q_rho = s_rho, a double; the seven quotients are float / double rounded back to float).
template <bool PROCJF, bool GREC>
__device__ __forceinline__ void tvr2_body_f32(const TvrArgs &a, const int seq, const int blk, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * kTvrBlock >= kn) return;
    static_assert(kTvrPasses == 1, "one KeyLine per thread");
    const KlSoA &ko = a.kl_old[seq];
    double *rout0 = a.resid + ((size_t)sq->res_t * a.nseq + seq) * a.cap;
    double *rout1 = a.resid + ((size_t)sq->res_new * a.nseq + seq) * a.cap;
    const double marker = __longlong_as_double((long long)resid_carry_bits());
    constexpr int NW = kTvrThreads / 64;
    __shared__ float s_wlast[2][NW];
    __shared__ int s_whas[2][NW];
    const float zf = (float)a.zfm, max_r = (float)a.max_r, simil_t = (float)a.match_thresh;

    const int ikl = blk * kTvrBlock + tid;
    // (tvr_body's "no defaults / status != 0" form measured 2 % SLOWER here, 774 -> 791 us per step, profiles/r06_tvr_mid_range_quotients_ab.txt: kept as it was)
    float fm[2] = {0, 0}, dfx[2] = {0, 0}, dfy[2] = {0, 0}, fi[2] = {0, 0};
    float ptx[2] = {0, 0}, pty[2] = {0, 0}, ptz[2] = {1, 1}, pix[2] = {0, 0}, piy[2] = {0, 0}, rho_p[2] = {1, 1};
    int status[2] = {0, 0};
    double s_rho = 1;
    if (ikl < kn) {
        s_rho = ldg(ko.s_rho, ikl);
        const int32_t mnum = ldg(ko.m_num, ikl);
        const float2 pm0 = ldg(ko.p_m, ikl);
        const double rho0 = ldg(ko.rho, ikl);
        const float2 klm = ldg(ko.m_m, ikl);
        const float knm = ldg(ko.n_m, ikl);
        const uint32_t fc = a.framecount[seq];
        const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
        const bool skip = s_rho > sq->s_rho_min_eval || (uint32_t)mnum < mthr;
        if (!skip) {
            const float sz = 1.f / (float)rho0;
            const float pz_zf0 = (1.f / zf) * sz;
            const float sx = pz_zf0 * pm0.x, sy = pz_zf0 * pm0.y;
            float px[2], py[2];
            bool inimg[2];
            size_t fidx[2];
            float rmx[2], rmy[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const double *R = c ? sq->Rt : sq->zRt, *V = c ? sq->Vt : sq->zVt, *RM = c ? sq->RM : sq->zRM;
                ptx[c] = (float)R[0] * sx; ptx[c] += (float)R[1] * sy; ptx[c] += (float)R[2] * sz; ptx[c] = (float)V[0] + ptx[c];
                pty[c] = (float)R[3] * sx; pty[c] += (float)R[4] * sy; pty[c] += (float)R[5] * sz; pty[c] = (float)V[1] + pty[c];
                ptz[c] = (float)R[6] * sx; ptz[c] += (float)R[7] * sy; ptz[c] += (float)R[8] * sz; ptz[c] = (float)V[2] + ptz[c];
                rho_p[c] = 1.f / ptz[c];
                const float pz_zf = zf * rho_p[c];
                pix[c] = pz_zf * ptx[c];
                piy[c] = pz_zf * pty[c];
                px[c] = pix[c] + a.ppx; py[c] = piy[c] + a.ppy;
                const int x = x86_cvttsd2si((double)px[c] + 0.5), y = x86_cvttsd2si((double)py[c] + 0.5);
                inimg[c] = !(x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1);
                fidx[c] = inimg[c] ? field16_index(x, y, a.f16tx) : (size_t)0;
                rmx[c] = (float)RM[0] * klm.x + (float)RM[1] * klm.y;
                rmy[c] = (float)RM[2] * klm.x + (float)RM[3] * klm.y;
                fm[c] = max_r;
                status[c] = inimg[c] ? 3 : 1;
            }
            const uint16_t *fld = a.field16 + (size_t)seq * a.f16stride;
            const uint32_t f0 = fld[fidx[0]], f1 = fld[fidx[1]];
            const bool hit0 = inimg[0] && f0 != 0u, hit1 = inimg[1] && f1 != 0u;
            const int ikf0 = hit0 ? (int)f0 - 1 : 0, ikf1 = hit1 ? (int)f1 - 1 : 0;
            float f_cpx[2], f_cpy[2], f_mx[2], f_my[2], f_ux[2] = {0, 0}, f_uy[2] = {0, 0};
            if (GREC) {
                const float4 g0 = ldg(a.kl_new[seq].grec, ikf0), g1 = ldg(a.kl_new[seq].grec, ikf1);
                f_cpx[0] = g0.x; f_cpy[0] = g0.y; f_mx[0] = g0.z; f_my[0] = g0.w;
                f_cpx[1] = g1.x; f_cpy[1] = g1.y; f_mx[1] = g1.z; f_my[1] = g1.w;
            } else {
                const MatchRec r0 = a.kl_new[seq].rec[ikf0], r1 = a.kl_new[seq].rec[ikf1];
                f_cpx[0] = r0.c_px; f_cpy[0] = r0.c_py; f_mx[0] = r0.m_mx; f_my[0] = r0.m_my; f_ux[0] = r0.u_mx; f_uy[0] = r0.u_my;
                f_cpx[1] = r1.c_px; f_cpy[1] = r1.c_py; f_mx[1] = r1.m_mx; f_my[1] = r1.m_my; f_ux[1] = r1.u_mx; f_uy[1] = r1.u_my;
            }
            const float p_n2 = knm * knm;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (c ? hit1 : hit0) {
                    const float p_esc = rmx[c] * f_mx[c] + rmy[c] * f_my[c];
                    if (!(fabsf(p_esc - p_n2) > simil_t * p_n2)) {
                        if (GREC) {
                            const float n2m = f_mx[c] * f_mx[c] + f_my[c] * f_my[c];
                            const float nm = sqrtf_mid(n2m);
                            div2_mid_f32(f_mx[c], f_my[c], nm, f_ux[c], f_uy[c]);
                        }
                        const float dx = px[c] - f_cpx[c], dy = py[c] - f_cpy[c];
                        fi[c] = dx * f_ux[c] + dy * f_uy[c];
                        dfx[c] = f_ux[c];
                        dfy[c] = f_uy[c];
                        fm[c] = fi[c];
                        status[c] = 2;
                    }
                }
            }
        }
    }
    unsigned long long below[2];
    float inh[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const unsigned long long vmask = __ballot(status[c] == 2);
        below[c] = vmask & ((1ull << lane) - 1ull);
        const int src = below[c] ? 63 - __clzll(below[c]) : 0;
        inh[c] = __shfl(fi[c], src, 64);
        const int top = vmask ? 63 - __clzll(vmask) : 0;
        const float wl = __shfl(fi[c], top, 64);
        if (lane == 0) {
            s_whas[c][wave] = vmask != 0;
            s_wlast[c][wave] = wl;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
        double *rout = c ? rout1 : rout0;
        if (status[c] == 3) {
            double v = marker;
            bool have = false;
            if (below[c]) { v = (double)inh[c]; have = true; }
            for (int pw = wave - 1; pw >= 0 && !have; pw--)
                if (s_whas[c][pw]) { v = (double)s_wlast[c][pw]; have = true; }
            rout[ikl] = v;
        } else if (status[c] == 2) {
            rout[ikl] = (double)fi[c];
        } else if (status[c] == 1) {
            rout[ikl] = (double)max_r;
        } else if (ikl < kn) {
            rout[ikl] = 0.0;
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            double bl = marker;
            for (int pw = NW - 1; pw >= 0; pw--)
                if (s_whas[c][pw]) { bl = (double)s_wlast[c][pw]; break; }
            (c ? a.block_last : a.block_last_z)[(size_t)seq * a.nblk + blk] = bl;
        }
    }
    __shared__ float s_red[2][NW][32];
    const double inv_q = 1.0 / s_rho;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float J[6] = {0, 0, 0, 0, 0, 0};
        float f = fm[c];
        if (ikl < kn) {
            if (PROCJF) {
                float t0 = zf * rho_p[c];
                J[0] = t0 * dfx[c];
                J[1] = t0 * dfy[c];
                t0 = rho_p[c] * pix[c];
                J[2] = t0 * dfx[c];
                t0 = rho_p[c] * piy[c];
                J[2] += t0 * dfy[c];
                J[3] = J[1] * ptz[c]; J[3] += J[2] * pty[c];
                J[4] = J[0] * ptz[c]; J[4] += J[2] * ptx[c];
                t0 = J[0] * pty[c];
                J[5] = -1.f * t0; J[5] += J[1] * ptx[c];
#pragma unroll
                for (int j = 0; j < 6; j++) J[j] = (float)div_rn((double)J[j], s_rho, inv_q);
            }
            f = (float)div_rn((double)f, s_rho, inv_q);
        }
        if (PROCJF) {
            float sums[kNumSums];
            int ns = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) sums[ns++] = J[i] * J[j];
#pragma unroll
            for (int i = 0; i < 6; i++) sums[ns++] = J[i] * f;
            sums[ns] = f * f;
            const int idx = wave_reduce28_f32(sums, lane);
            if ((lane & 1) == 0) s_red[c][wave][idx] = sums[0];
        } else {
            float v = f * f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) s_red[c][wave][kNumSums - 1] = v;
        }
    }
    __syncthreads();
    if (wave < 2) {
        const int c = 1 - wave;
        if (PROCJF ? lane < kNumSums : lane == kNumSums - 1) {
            float v = s_red[c][0][lane];
#pragma unroll
            for (int wv = 1; wv < NW; wv++) v += s_red[c][wv][lane];
            (c ? a.partials : a.partials_z)[((size_t)seq * a.nblk + blk) * kNumSums + lane] = (double)v;
        }
    }
}
template <bool PROCJF, bool GREC>
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot2_f32(TvrArgs a) {
    tvr2_body_f32<PROCJF, GREC>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}

#ifdef EDGEHIP_EXPERIMENTS   // two KeyLines per thread in the reweighted evaluation: measured no faster (EDGEHIP_TVR_RW2)
// ---------------------------------------------------------------------------------------------------
// The reweighted evaluation with TWO KeyLines per thread (whole batches): thread t of block b owns KeyLines 512 b + t and
// 512 b + 256 + t and walks them level by level like tvr2_body walks its two chains — both KeyLines' streams, then both field
// reads, then both records in flight together — so that a thread has two independent chains of dependent gathers outstanding
// instead of one, and the 28 products of both accumulate in registers before ONE transposed wave reduction (the reduction is
// a quarter of the one-KeyLine kernel's instructions).  The residual buffers, their "last valid residual" markers and the
// per-256-KeyLine carries are exactly the one-KeyLine kernel's (each half is a block of its own for that purpose: the LM step
// and the next evaluation see no difference); the partial sums are one row per 512 KeyLines (the odd row is zero), i.e. the
// 28 sums are added in another — fixed — order: same tolerance against the reference, not the same bits as tvr_body.
// Measured (round 4): no gain at 1024 sequences — the kernel is not short of loads in flight at 8 waves per SIMD, and at 84
// registers it runs 5 — so it is an option (EDGEHIP_TVR_RW2 = smallest launch, in evaluation blocks, that takes it), not the default.
// ---------------------------------------------------------------------------------------------------
template <bool GREC>
__device__ __forceinline__ void tvr_rw2_body(const TvrArgs &a, const int seq, const int blk, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * 2 * kTvrBlock >= kn) return;  // whole block beyond the list (block-uniform)
    static_assert(kTvrPasses == 1 && kTvrThreads == kTvrBlock, "one KeyLine per thread and carry granule");
    const KlSoA &ko = a.kl_old[seq];
    const int res_in = sq->res_cur, res_out = sq->lm_phase == 0 ? sq->res_t : sq->res_new;
    const double *rin = a.resid + ((size_t)res_in * a.nseq + seq) * a.cap;
    double *rout = a.resid + ((size_t)res_out * a.nseq + seq) * a.cap;
    const double *carry_row = a.resid_carry + ((size_t)res_in * a.nseq + seq) * a.nblk;
    const double marker = __longlong_as_double((long long)resid_carry_bits());
    constexpr int NW = kTvrThreads / 64;
    __shared__ double s_wlast[2][NW];
    __shared__ int s_whas[2][NW];

    int ikl[2];
    double fm[2] = {0, 0}, dfx[2] = {0, 0}, dfy[2] = {0, 0}, fi[2] = {0, 0};
    double ptx[2] = {0, 0}, pty[2] = {0, 0}, ptz[2] = {1, 1}, pix[2] = {0, 0}, piy[2] = {0, 0}, rho_p[2] = {1, 1};
    double s_rho[2] = {1, 1}, inv_w2[2] = {1, 1}, rho_own[2] = {0, 0};
    int status[2] = {0, 0}, mid_f[2] = {-1, -1};
    // ---- level 0: both KeyLines' streams ----
    int32_t mnum[2] = {0, 0};
    float2 pm0[2], klm[2];
    float knm[2] = {1.f, 1.f};
    double rprev[2] = {0, 0}, cin[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 2; c++) {
        ikl[c] = (2 * blk + c) * kTvrBlock + tid;
        pm0[c] = make_float2(0.f, 0.f); klm[c] = make_float2(0.f, 0.f);
        if (ikl[c] < kn) {
            s_rho[c] = ldg(ko.s_rho, ikl[c]);
            mnum[c] = ldg(ko.m_num, ikl[c]);
            pm0[c] = ldg(ko.p_m, ikl[c]);
            rho_own[c] = ldg(ko.rho, ikl[c]);
            klm[c] = ldg(ko.m_m, ikl[c]);
            knm[c] = ldg(ko.n_m, ikl[c]);
            rprev[c] = rin[ikl[c]];
        }
        cin[c] = 2 * blk + c < a.nblk ? carry_row[2 * blk + c] : 0.0;
    }
    const uint32_t fc = a.framecount[seq];
    const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
    // ---- level 1: both projections, both field reads ----
    double px[2] = {0, 0}, py[2] = {0, 0};
    bool live[2], inimg[2];
    size_t fidx[2];
    float rmx[2], rmy[2];
    const double *R = sq->Rt, *V = sq->Vt, *RM = sq->RM;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        live[c] = ikl[c] < kn && !(s_rho[c] > sq->s_rho_min_eval || (uint32_t)mnum[c] < mthr);
        inimg[c] = false;
        fidx[c] = 0;
        rmx[c] = rmy[c] = 0.f;
        if (live[c]) {
            const double sz = 1 / rho_own[c];
            const double pz_zf0 = a.inv_zfm * sz;
            const double sx = pz_zf0 * (double)pm0[c].x, sy = pz_zf0 * (double)pm0[c].y;
            ptx[c] = R[0] * sx; ptx[c] += R[1] * sy; ptx[c] += R[2] * sz; ptx[c] = V[0] + ptx[c];
            pty[c] = R[3] * sx; pty[c] += R[4] * sy; pty[c] += R[5] * sz; pty[c] = V[1] + pty[c];
            ptz[c] = R[6] * sx; ptz[c] += R[7] * sy; ptz[c] += R[8] * sz; ptz[c] = V[2] + ptz[c];
            rho_p[c] = 1 / ptz[c];
            const double pz_zf = a.zfm * rho_p[c];
            pix[c] = pz_zf * ptx[c];
            piy[c] = pz_zf * pty[c];
            px[c] = pix[c] + (double)a.ppx; py[c] = piy[c] + (double)a.ppy;
            const int x = x86_cvttsd2si(px[c] + 0.5), y = x86_cvttsd2si(py[c] + 0.5);
            double rp = rprev[c];
            if (is_carry(rp)) rp = cin[c];
            if (fabs(rp) > a.k_huber) {
#if EDGEHIP_TVR_REF_ORDER
                inv_w2[c] = a.k_huber / fabs(rp);   // the weight itself here
#else
                const double rk = fabs(rp) * a.inv_k_huber;
                inv_w2[c] = rk * rk;
#endif
            }
            inimg[c] = !(x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1);
            fidx[c] = inimg[c] ? field16_index(x, y, a.f16tx) : (size_t)0;
            rmx[c] = (float)(RM[0] * (double)klm[c].x + RM[1] * (double)klm[c].y);
            rmy[c] = (float)(RM[2] * (double)klm[c].x + RM[3] * (double)klm[c].y);
            fm[c] = a.max_r;
            status[c] = inimg[c] ? 3 : 1;
        }
    }
    const uint16_t *fld = a.field16 + (size_t)seq * a.f16stride;
    const uint32_t f0 = fld[fidx[0]], f1 = fld[fidx[1]];
    // ---- level 2: both records ----
    const bool hit0 = inimg[0] && f0 != 0u, hit1 = inimg[1] && f1 != 0u;
    const int ikf0 = hit0 ? (int)f0 - 1 : 0, ikf1 = hit1 ? (int)f1 - 1 : 0;
    float f_cpx[2], f_cpy[2], f_mx[2], f_my[2], f_ux[2] = {0, 0}, f_uy[2] = {0, 0};
    if (GREC) {
        const float4 g0 = ldg(a.kl_new[seq].grec, ikf0), g1 = ldg(a.kl_new[seq].grec, ikf1);
        f_cpx[0] = g0.x; f_cpy[0] = g0.y; f_mx[0] = g0.z; f_my[0] = g0.w;
        f_cpx[1] = g1.x; f_cpy[1] = g1.y; f_mx[1] = g1.z; f_my[1] = g1.w;
    } else {
        const MatchRec r0 = a.kl_new[seq].rec[ikf0], r1 = a.kl_new[seq].rec[ikf1];
        f_cpx[0] = r0.c_px; f_cpy[0] = r0.c_py; f_mx[0] = r0.m_mx; f_my[0] = r0.m_my; f_ux[0] = r0.u_mx; f_uy[0] = r0.u_my;
        f_cpx[1] = r1.c_px; f_cpy[1] = r1.c_py; f_mx[1] = r1.m_mx; f_my[1] = r1.m_my; f_ux[1] = r1.u_mx; f_uy[1] = r1.u_my;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c ? hit1 : hit0) {
            const double p_n2 = (double)(knm[c] * knm[c]);
            const double p_esc = (double)(rmx[c] * f_mx[c] + rmy[c] * f_my[c]);
            if (!(fabs(p_esc - p_n2) > a.match_thresh * p_n2)) {
                if (GREC) {
                    const float n2m = f_mx[c] * f_mx[c] + f_my[c] * f_my[c];
                    const float nm = sqrtf_mid(n2m);
                    f_ux[c] = f_mx[c] / nm; f_uy[c] = f_my[c] / nm;
                }
                const double dx = px[c] - (double)f_cpx[c], dy = py[c] - (double)f_cpy[c];
                fi[c] = dx * (double)f_ux[c] + dy * (double)f_uy[c];
                dfx[c] = (double)f_ux[c];
                dfy[c] = (double)f_uy[c];
                fm[c] = fi[c];
                mid_f[c] = c ? ikf1 : ikf0;
                status[c] = 2;
            }
        }
    }
    // ---- DResidualNew: each half is a block of its own for the "last valid fi" propagation (tvr_body's rule) ----
    unsigned long long below[2];
    double inh[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const unsigned long long vmask = __ballot(status[c] == 2);
        below[c] = vmask & ((1ull << lane) - 1ull);
        const int src = below[c] ? 63 - __clzll(below[c]) : 0;
        inh[c] = __shfl(fi[c], src, 64);
        const int top = vmask ? 63 - __clzll(vmask) : 0;
        const double wl = __shfl(fi[c], top, 64);
        if (lane == 0) {
            s_whas[c][wave] = vmask != 0;
            s_wlast[c][wave] = wl;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (status[c] == 3) {
            double v = marker;
            bool have = false;
            if (below[c]) { v = inh[c]; have = true; }
            for (int pw = wave - 1; pw >= 0 && !have; pw--)
                if (s_whas[c][pw]) { v = s_wlast[c][pw]; have = true; }
            rout[ikl[c]] = v;
        } else if (status[c] == 2) {
            rout[ikl[c]] = fi[c];
        } else if (status[c] == 1) {
            rout[ikl[c]] = a.max_r;
        } else if (ikl[c] < kn) {
            rout[ikl[c]] = 0.0;
        }
        if (a.write_mid && ikl[c] < kn) {
            ko.m_id_f[ikl[c]] = mid_f[c];
            if (a.fwd_key && mid_f[c] >= 0) atomicMax(&a.fwd_key[(size_t)seq * a.cap + mid_f[c]], ord_bits(rho_own[c]));
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if ((2 * blk + c) * kTvrBlock >= kn || 2 * blk + c >= a.nblk) continue;
            double bl = marker;
            for (int pw = NW - 1; pw >= 0; pw--)
                if (s_whas[c][pw]) { bl = s_wlast[c][pw]; break; }
            a.block_last[(size_t)seq * a.nblk + 2 * blk + c] = bl;
        }
    }
    // ---- Jacobian rows, weights, the 28 sums of both KeyLines, one reduction ----
    double sums[kNumSums];
#pragma unroll
    for (int i = 0; i < kNumSums; i++) sums[i] = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        double J[6] = {0, 0, 0, 0, 0, 0};
        double fmc = fm[c];
        if (ikl[c] < kn) {
#if EDGEHIP_TVR_REF_ORDER
            fmc *= inv_w2[c]; dfx[c] *= inv_w2[c]; dfy[c] *= inv_w2[c];
#endif
            double t0 = a.zfm * rho_p[c];
            J[0] = t0 * dfx[c];
            J[1] = t0 * dfy[c];
            t0 = rho_p[c] * pix[c];
            J[2] = t0 * dfx[c];
            t0 = rho_p[c] * piy[c];
            J[2] += t0 * dfy[c];
            J[3] = J[1] * ptz[c]; J[3] += J[2] * pty[c];
            J[4] = J[0] * ptz[c]; J[4] += J[2] * ptx[c];
            t0 = J[0] * pty[c];
            J[5] = -1 * t0; J[5] += J[1] * ptx[c];
            const double qvel = (a.zfm * dfx[c] * V[0] + a.zfm * dfy[c] * V[1] + (pix[c] * dfx[c] + piy[c] * dfy[c]) * V[2]);
#if EDGEHIP_TVR_REF_ORDER
            const double q_rho = sqrt(s_rho[c] * qvel * s_rho[c] * qvel + 1), r_q = 1.0 / q_rho;
#pragma unroll
            for (int j = 0; j < 6; j++) J[j] = div_rn(J[j], q_rho, r_q);
            fmc = div_rn(fmc, q_rho, r_q);
#else
            const double sq_ = s_rho[c] * qvel;
            const double inv_q = rsqrt_f64(sq_ * sq_ + inv_w2[c]);
#pragma unroll
            for (int j = 0; j < 6; j++) J[j] *= inv_q;
            fmc *= inv_q;
#endif
        }
        int ns = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) sums[ns++] += J[i] * J[j];
#pragma unroll
        for (int i = 0; i < 6; i++) sums[ns++] += J[i] * fmc;
        sums[ns] += fmc * fmc;
    }
    __shared__ double s_red[NW][32];
    const int idx = wave_reduce28(sums, lane);
    if ((lane & 1) == 0) s_red[wave][idx] = sums[0];
    __syncthreads();
    if (tid < kNumSums) {
        double v = s_red[0][tid];
#pragma unroll
        for (int wv = 1; wv < NW; wv++) v += s_red[wv][tid];
        a.partials[((size_t)seq * a.nblk + 2 * blk) * kNumSums + tid] = v;
        if (2 * blk + 1 < a.nblk) a.partials[((size_t)seq * a.nblk + 2 * blk + 1) * kNumSums + tid] = 0.0;   // the LM step adds one row per 256 KeyLines
    }
}

template <bool GREC>
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot_rw2(TvrArgs a) {
    tvr_rw2_body<GREC>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}
#endif   // EDGEHIP_EXPERIMENTS

#ifndef EDGEHIP_TVR2_WAVES
#define EDGEHIP_TVR2_WAVES 0   // > 0: occupancy the two-chain evaluation is compiled for (waves per SIMD), A/B experiments
#endif
template <bool PROCJF, bool GREC>
__global__ __launch_bounds__(kTvrThreads)
#if EDGEHIP_TVR2_WAVES > 0
__attribute__((amdgpu_waves_per_eu(EDGEHIP_TVR2_WAVES, EDGEHIP_TVR2_WAVES)))
#endif
void k_try_velrot2(TvrArgs a) {
    tvr2_body<PROCJF, GREC>(a, blockIdx.z, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// 6x6 linear algebra for the LM step (single lane)
// ---------------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition by cyclic Jacobi; A = V diag(e) V^T.  Stands in for LAPACK dgesvd_ behind
// TooN::SVD<> (for a symmetric matrix singular values = |e|, U = V*sign(e)).
__device__ __noinline__ void jacobi_eig6(const double Ain[36], double V[36], double e[6], double *A /*[36] scratch*/) {
#pragma nounroll
    for (int i = 0; i < 36; i++) { A[i] = Ain[i]; V[i] = 0; }
#pragma nounroll
    for (int i = 0; i < 6; i++) V[i * 7] = 1;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, diag = 0;
#pragma nounroll
        for (int p = 0; p < 6; p++) {
            diag += A[p * 7] * A[p * 7];
#pragma nounroll
            for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q];
        }
        if (!(off > 1e-32 * diag) || !(off > 0)) break;  // off-diagonal energy below fp64 roundoff of the diagonal
#pragma nounroll
        for (int p = 0; p < 5; p++)
#pragma nounroll
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0) continue;
                const double theta = (A[q * 7] - A[p * 7]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma nounroll
                for (int k = 0; k < 6; k++) {
                    const double akp = A[k * 6 + p], akq = A[k * 6 + q];
                    A[k * 6 + p] = c * akp - s * akq;
                    A[k * 6 + q] = s * akp + c * akq;
                }
#pragma nounroll
                for (int k = 0; k < 6; k++) {
                    const double apk = A[p * 6 + k], aqk = A[q * 6 + k];
                    A[p * 6 + k] = c * apk - s * aqk;
                    A[q * 6 + k] = s * apk + c * aqk;
                }
#pragma nounroll
                for (int k = 0; k < 6; k++) {
                    const double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
                    V[k * 6 + p] = c * vkp - s * vkq;
                    V[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
#pragma nounroll
    for (int i = 0; i < 6; i++) e[i] = A[i * 7];
}

// h = SVD(A).backsub(b) with TooN's conditioning (SVD.h:176-196, 264-272; condition_no = 1e9)
__device__ __noinline__ void svd_backsub6(const double A[36], const double b[6], double h[6], double *V, double *Awork,
                                          double *e, double *y) {
    jacobi_eig6(A, V, e, Awork);
    double smax = 0;
#pragma nounroll
    for (int i = 0; i < 6; i++) smax = fmax(smax, fabs(e[i]));
#pragma nounroll
    for (int i = 0; i < 6; i++) {
        double d = 0;
#pragma nounroll
        for (int k = 0; k < 6; k++) d += V[k * 6 + i] * b[k];
        const double inv = (fabs(e[i]) * 1e9 <= smax) ? 0.0 : 1.0 / e[i];
        y[i] = d * inv;
    }
#pragma nounroll
    for (int k = 0; k < 6; k++) {
        double d = 0;
#pragma nounroll
        for (int i = 0; i < 6; i++) d += V[k * 6 + i] * y[i];
        h[k] = d;
    }
}

// TooN::Cholesky<6> (LDL^T, Cholesky.h:88-125) and its vector backsub (:131-160)
__device__ __noinline__ void chol6(const double A[36], double L[36]) {
#pragma nounroll
    for (int i = 0; i < 36; i++) L[i] = A[i];
#pragma nounroll
    for (int col = 0; col < 6; col++) {
        double inv_diag = 1;
#pragma nounroll
        for (int row = col; row < 6; row++) {
            double val = L[row * 6 + col];
#pragma nounroll
            for (int col2 = 0; col2 < col; col2++) val -= L[col2 * 6 + col] * L[row * 6 + col2];
            if (row == col) {
                L[row * 6 + col] = val;
                if (val == 0) return;  // rank deficient: TooN stops here
                inv_diag = 1 / val;
            } else {
                L[col * 6 + row] = val;
                L[row * 6 + col] = val * inv_diag;
            }
        }
    }
}
__device__ __noinline__ void chol6_backsub(const double L[36], const double v[6], double r[6], double *y) {
#pragma nounroll
    for (int i = 0; i < 6; i++) {
        double val = v[i];
#pragma nounroll
        for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
        y[i] = val;
    }
#pragma nounroll
    for (int i = 0; i < 6; i++) y[i] /= L[i * 7];
#pragma nounroll
    for (int i = 5; i >= 0; i--) {
        double val = y[i];
#pragma nounroll
        for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
        r[i] = val;
    }
}
// matrix backsub of the identity: get_inverse() (Cholesky.h:165-200); note y[i] *= (1/d) here, not y[i] /= d
__device__ __noinline__ void chol6_inverse(const double L[36], double Inv[36], double *y, double *r) {
#pragma nounroll
    for (int c = 0; c < 6; c++) {
#pragma nounroll
        for (int i = 0; i < 6; i++) {
            double val = (i == c) ? 1.0 : 0.0;
#pragma nounroll
            for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
            y[i] = val;
        }
#pragma nounroll
        for (int i = 0; i < 6; i++) y[i] *= (1 / L[i * 7]);
#pragma nounroll
        for (int i = 5; i >= 0; i--) {
            double val = y[i];
#pragma nounroll
            for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
            r[i] = val;
        }
#pragma nounroll
        for (int i = 0; i < 6; i++) Inv[i * 6 + c] = r[i];
    }
}

// Register-resident versions (arrays indexed only by unrolled constants) of chol6 / backsub / inverse above.
__device__ __forceinline__ void chol6_r(const double (&A)[36], double (&L)[36]) {
#pragma unroll
    for (int i = 0; i < 36; i++) L[i] = A[i];
    bool alive = true;   // TooN returns at the first zero pivot, leaving the rest untouched
#pragma unroll
    for (int col = 0; col < 6; col++) {
        double inv_diag = 1;
#pragma unroll
        for (int row = col; row < 6; row++) {
            double val = L[row * 6 + col];
#pragma unroll
            for (int col2 = 0; col2 < col; col2++) val -= L[col2 * 6 + col] * L[row * 6 + col2];
            if (row == col) {
                if (alive) L[row * 6 + col] = val;
                if (val == 0) alive = false;
                inv_diag = 1 / val;
            } else if (alive) {
                L[col * 6 + row] = val;
                L[row * 6 + col] = val * inv_diag;
            }
        }
    }
}
__device__ __forceinline__ void chol6_backsub_r(const double (&L)[36], const double (&vv)[6], double (&r)[6]) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double val = vv[i];
#pragma unroll
        for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
        y[i] = val;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] /= L[i * 7];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double val = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
        r[i] = val;
    }
}
__device__ __forceinline__ void chol6_inverse_r(const double (&L)[36], double (&Inv)[36]) {
    double invd[6];
#pragma unroll
    for (int i = 0; i < 6; i++) invd[i] = 1 / L[i * 7];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double y[6], r[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double val = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
            y[i] = val;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) y[i] *= invd[i];   // y[i] *= (1/d), Cholesky.h:165-200
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            double val = y[i];
#pragma unroll
            for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
            r[i] = val;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) Inv[i * 6 + c] = r[i];
    }
}

// The 6x6 solve of one LM step: h = TooN::Cholesky<6>(ApI).backsub(nb) (global_tracker.cpp:767-768) or, in the init phase,
// h = TooN::SVD<>(ApI).backsub(nb) (:660-661, 711-712).  Shared by lm_body and edgehip_lm_solve (the parity test of the rule below).
__device__ __forceinline__ void lm_solve6(const double (&ApI)[36], const double (&nb)[6], const bool svd_rule, double (&hh)[6],
                                          double (*s_m)[36], double (*s_v)[6]) {
    double L[36];
    chol6_r(ApI, L);
    bool use_svd = false;
    if (svd_rule) {
        // TooN::SVD<>::backsub zeroes singular values s_i with s_i * 1e9 <= s_max (SVD.h:264-272); when none is, the
        // pseudo-inverse IS the inverse and LDL^T gives the same h.  A bound that needs no decomposition: for SPD A with LDL^T
        // pivots d_i and trace T, s_max <= T and det A = prod d_i, so s_min >= prod d_i / T^5 and
        // cond(A) <= prod (T / d_i).  ApI = JtJ + u I with u ~ 1e-3 max(JtJ) has condition 10..1e4 and passes; anything the bound
        // cannot clear (or that is not positive definite) takes the Jacobi path, which applies TooN's rule value by value.
        // (Round 2 compared the pivot ratio with 1e7: unsound — pivots 8e9..4e3 at condition 1.7e9 — found by
        // tests/test_knife_edge_gpu.py::test_lm_solve_against_toon_either_side_of_the_svd_cutoff.)
        double dmin = L[0], tr = ApI[0];
#pragma unroll
        for (int i = 1; i < 6; i++) { dmin = fmin(dmin, L[i * 7]); tr += ApI[i * 7]; }
        double bound = 1;
#pragma unroll
        for (int i = 0; i < 6; i++) bound *= tr / L[i * 7];
        use_svd = !(dmin > 0) || !(bound < 0.99e9);
    }
    if (use_svd) {
        double *A_l = s_m[0], *b_l = s_v[0], *h_l = s_v[2] + 0;
        for (int i = 0; i < 36; i++) A_l[i] = ApI[i];
        for (int i = 0; i < 6; i++) b_l[i] = nb[i];
        svd_backsub6(A_l, b_l, h_l, s_m[2], s_m[3], s_v[1], s_m[1]);
#pragma unroll
        for (int i = 0; i < 6; i++) hh[i] = h_l[i];
    } else {
        chol6_backsub_r(L, nb, hh);
    }
}

// edgehip_lm_solve: n independent systems through lm_solve6, one wave each (lane 0 does the algebra, like k_lm_step)
__global__ __launch_bounds__(64) void k_lm_solve(const double *__restrict__ A, const double *__restrict__ b, double *__restrict__ h,
                                                  int svd_rule) {
    __shared__ double s_m[4][36], s_v[3][6];
    if (threadIdx.x != 0) return;
    double ApI[36], nb[6], hh[6];
    for (int i = 0; i < 36; i++) ApI[i] = A[(size_t)blockIdx.x * 36 + i];
    for (int i = 0; i < 6; i++) nb[i] = b[(size_t)blockIdx.x * 6 + i];
    lm_solve6(ApI, nb, svd_rule != 0, hh, s_m, s_v);
    for (int i = 0; i < 6; i++) h[(size_t)blockIdx.x * 6 + i] = hh[i];
}

// ---------------------------------------------------------------------------------------------------
// k_lm_step: everything Minimizer_RV does between two TryVelRot evaluations (global_tracker.cpp:631-816).
// The host knows the (static) call sequence and passes it as a bit mask of operations.
// ---------------------------------------------------------------------------------------------------
enum LmOps : unsigned {
    LM_REDUCE_CUR = 1u << 0,   // partials -> F, JtJ, JtF
    LM_REDUCE_NEW = 1u << 1,   // partials -> Fnew, JtJnew, JtFnew
    LM_NOJAC = 1u << 2,        // the evaluation had ProcJF=false (score only)
    LM_INIT = 1u << 3,         // F0 = F; u = tau*max(JtJ)
    LM_RESET_V = 1u << 4,      // v = 2
    LM_GAIN_RATIO = 1u << 5,   // gain = (F-Fnew)/(0.5*h*(u*h-JtF)); accept / reject
    LM_GAIN_DIFF = 1u << 6,    // gain = F-Fnew; accept / reject
    LM_SWAP_ON_ACCEPT = 1u << 7,
    LM_SAVE_T = 1u << 8,       // stash zero-init result, restart from the prior (Vel, W0)
    LM_PICK = 1u << 9,         // keep the better of the two initialisations, swap residual buffers
    LM_SOLVE_SVD = 1u << 10,
    LM_SOLVE_CHOL = 1u << 11,
    LM_SETUP_X = 1u << 12,     // next evaluation at X
    LM_SETUP_XNEW = 1u << 13,  // next evaluation at Xnew
    LM_FINISH = 1u << 14,
    LM_BEGIN = 1u << 15,       // start of a minimisation (init state, X from init_type)
    LM_PHASE_A = 1u << 16,     // next evaluation writes Rest (zero-init trial of init_type 2)
    LM_PHASE_BC = 1u << 17,    // next evaluation writes ResidualNew
    LM_BEGIN_KF = 1u << 18,    // kfvo::Minimizer_RV_KF: X and the uncertainty gate come with the request (kfvo.cpp:1737-1738)
    LM_FINISH_KF = 1u << 19,   // ... and the result goes to the caller, not into the sequence state (:1808-1821)
    // Two-chain initialisation (k_try_velrot2 / k_lm_step2): both chains of TrackerInitType = 2 advance in the same launches
    LM_BEGIN2 = 1u << 20,      // with LM_BEGIN: X = 0 for the zero-init chain, X = (Vel, W0) for the running one, both transforms set up
    LM_PICK2 = 1u << 21,       // k_lm_step2 only: keep the better of the two chains (global_tracker.cpp:740-752), swap residual buffers
};

struct LmArgs {
    SeqDev *seq;
    const double *partials;   // [B][nblk][kNumSums]
    const double *block_last; // [B][nblk]
    const double *partials_z = nullptr, *block_last_z = nullptr;   // the zero-init chain's (k_lm_step2)
    double *resid_carry;      // [kResidBufs][B][nblk]
    uint32_t *framecount;     // [B] of the new slot
    int nblk, nseq;
    unsigned ops;
    int init_type;
    const edgehip_kf_request *kf_in;   // [B] (LM_BEGIN_KF)
    edgehip_kf_result *kf_out;         // [B] (LM_FINISH_KF)
    const int32_t *kn_src = nullptr;   // [B] or null: KeyLine count of the tracked edge map, when the step runs in the launch that would
                                       // otherwise have left it in the state (k_tvr_prepare_begin)
    int f32 = 0;                       // Minimizer_RV<float>: what the reference declares as T — JtJ, JtF, ApI, h, X, Xnew (global_tracker.cpp:601-603) —
                                       // is rounded to float where the reference's assignments round it; the solves, u, v, gain and F stay double
};
__device__ __forceinline__ double lm_rt(const double x, const int f32) { return f32 ? (double)(float)x : x; }

template <bool WAVE_ONLY>
__device__ __forceinline__ void lm_sync() {
    if (WAVE_ONLY) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // LDS traffic of one wave is in order; keep the compiler from moving it
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// One wave per sequence.  WAVE_ONLY: the caller is a single wave of a larger workgroup (k_minimizer_persistent), so the
// points where the lanes hand data to each other through LDS are wave-level fences instead of workgroup barriers.
// DUAL (k_lm_step2, 128 threads): the step of both chains of a two-chain initialisation at once — wave 0 runs the step on the
// running (prior-init) chain's locals, wave 1 the same operations on the zero-init chain's (SeqDev::z*), each reducing its own
// partial sums and resolving the carries of its own residual buffer; LM_PICK2 follows on wave 0 when both are done.
template <bool WAVE_ONLY, bool DUAL = false>
__device__ __forceinline__ void lm_body(const LmArgs &a, const int seq, const int tid) {
    static_assert(!(WAVE_ONLY && DUAL), "the two chains meet at workgroup barriers");
    constexpr int NT = DUAL ? 128 : 64;
    const int lane = tid & 63;
    const int ch = DUAL ? tid >> 6 : 0;      // 1 = zero-init chain
    const unsigned ops = a.ops;
    // The sequence state is staged in LDS for the whole step: the serial LM logic touches it hundreds of
    // times and a global round trip per access was the dominant cost of this kernel.
    static_assert(sizeof(SeqDev) % 8 == 0, "SeqDev is copied as 64-bit words");
    constexpr int kWords = sizeof(SeqDev) / 8;
    __shared__ unsigned long long s_state[kWords];
    unsigned long long *gstate = reinterpret_cast<unsigned long long *>(a.seq + seq);
    const double *a_partials = ch ? a.partials_z : a.partials;
    const double *a_block_last = ch ? a.block_last_z : a.block_last;
    // Everything the step reads from global memory is requested here, in one go: the sequence state, the evaluation's
    // per-block partial sums and the blocks' last residuals.  The step is one wave per sequence and nothing but latency —
    // it used to pay the state's round trip, then (the block count comes out of the state) four more for the partials in
    // batches of eight, then one for the residuals.  How many blocks count is decided after the state has arrived; blocks
    // beyond it hold stale sums and are loaded in vain (a.nblk <= 64 blocks: 32 per lane; larger contexts keep the loop).
    constexpr int kPre = 32;
    const bool prefetch = a.nblk <= 2 * kPre && (ops & (LM_REDUCE_CUR | LM_REDUCE_NEW));
    const int pv_v = lane & 31, pv_g = lane >> 5;
    const bool pv_on = pv_v < kNumSums && (!(ops & LM_NOJAC) || pv_v == kNumSums - 1);
    double pv[kPre];
    double bl_pre = 0;
    if (prefetch) {
        const double *pp = a_partials + (size_t)seq * a.nblk * kNumSums;
#pragma unroll
        for (int j = 0; j < kPre; j++) {
            const int b = pv_g + 2 * j;
            pv[j] = (pv_on && b < a.nblk) ? pp[(size_t)b * kNumSums + pv_v] : 0.0;
        }
        if (lane < a.nblk) bl_pre = a_block_last[(size_t)seq * a.nblk + lane];
    }
    for (int i = tid; i < kWords; i += NT) s_state[i] = gstate[i];
    lm_sync<WAVE_ONLY>();
    SeqDev *sq = reinterpret_cast<SeqDev *>(s_state);
    if (a.kn_src) {
        if (tid == 0) sq->kn_old = a.kn_src[seq];
        lm_sync<WAVE_ONLY>();
    }
    const int kn = sq->kn_old;
    if (ops & LM_BEGIN) {
        if (tid == 0) {
            sq->eff_steps = 0;
            sq->v = 2;
            sq->res_cur = 0; sq->res_new = 1; sq->res_t = 2;
            sq->pub.minimizer_evals = 0;
            if (ops & LM_BEGIN2) {
                // both chains at once: the zero-init trial (global_tracker.cpp:650) in the z* locals, the prior-init trial
                // (:698-699) in the running ones — what LM_SAVE_T would set up after the first chain
                for (int i = 0; i < 6; i++) sq->zX[i] = 0;
                sq->z_eff_steps = 0;
                sq->zv = 2;
                for (int i = 0; i < 3; i++) { sq->X[i] = lm_rt(sq->pub.V[i], a.f32); sq->X[3 + i] = lm_rt(sq->pub.W[i], a.f32); }
            } else if (a.init_type == 1) {
                for (int i = 0; i < 3; i++) { sq->X[i] = lm_rt(sq->pub.V[i], a.f32); sq->X[3 + i] = lm_rt(sq->pub.W[i], a.f32); }
            } else {
                for (int i = 0; i < 6; i++) sq->X[i] = 0;
            }
            sq->s_rho_min_eval = sq->pub.s_rho_q;
        }
        lm_sync<WAVE_ONLY>();
    }
    if (ops & LM_BEGIN_KF) {
        if (lane == 0) {
            sq->eff_steps = 0;
            sq->v = 2;
            sq->res_cur = 0; sq->res_new = 1; sq->res_t = 2;
            sq->pub.minimizer_evals = 0;
            for (int i = 0; i < 6; i++) sq->X[i] = a.kf_in[seq].X0[i];
            sq->s_rho_min_eval = a.kf_in[seq].max_s_rho;
        }
        lm_sync<WAVE_ONLY>();
    }
    if (kn <= 0 && (ops & LM_FINISH_KF) && lane == 0) {   // Minimizer_RV_KF returns 0 on an empty list (kfvo.cpp:1700-1701)
        edgehip_kf_result &o = a.kf_out[seq];
        for (int i = 0; i < 6; i++) o.X[i] = a.kf_in[seq].X0[i];
        for (int i = 0; i < 36; i++) o.RRV[i] = 0;
        o.score_ratio = 0; o.F = 0; o.F0 = 0; o.evals = 0; o.mnum = 0;
    }
    if (kn <= 0) {  // Minimizer_RV returns immediately on an empty list (global_tracker.cpp:597-598)
        for (int i = tid; i < kWords; i += NT) gstate[i] = s_state[i];
        return;
    }
    const int nblk_used = (kn + kTvrBlock - 1) / kTvrBlock;

    // ---- finish the reduction of the evaluation that just ran (fixed order: deterministic) ----
    // 2 lane groups x 32 value slots: group g sums blocks g, g+2, ... (independent, unrolled loads), then
    // slot v adds the two group sums.
    constexpr int NCH = DUAL ? 2 : 1;
    __shared__ double s_part_[NCH][2][32];
    __shared__ double s_m_[NCH][4][36], s_v_[NCH][3][6];  // 6x6 work matrices / vectors of the solves (LDS, not scratch)
    __shared__ double s_sum_[NCH][kNumSums];
    __shared__ double s_bl_[NCH][256];
    double (*s_part)[32] = s_part_[ch];
    double (*s_m)[36] = s_m_[ch];
    double (*s_v)[6] = s_v_[ch];
    double *s_sum = s_sum_[ch], *s_bl = s_bl_[ch];
    // the chain's locals (LDS copy of the state)
    double *cX = ch ? sq->zX : sq->X, *cXnew = ch ? sq->zXnew : sq->Xnew, *ch_h = ch ? sq->zh : sq->h;
    double *cJtJ = ch ? sq->zJtJ : sq->JtJ, *cJtF = ch ? sq->zJtF : sq->JtF;
    double *cJtJnew = ch ? sq->zJtJnew : sq->JtJnew, *cJtFnew = ch ? sq->zJtFnew : sq->JtFnew;
    double *cF = ch ? &sq->zF : &sq->F, *cFnew = ch ? &sq->zFnew : &sq->Fnew, *cF0 = ch ? &sq->zF0 : &sq->F0;
    double *cu = ch ? &sq->zu : &sq->u, *cv = ch ? &sq->zv : &sq->v, *cgain = ch ? &sq->zgain : &sq->gain;
    int32_t *c_eff = ch ? &sq->z_eff_steps : &sq->eff_steps;
    if (ops & (LM_REDUCE_CUR | LM_REDUCE_NEW)) {
        const double *pp = a_partials + (size_t)seq * a.nblk * kNumSums;
        const int v = lane & 31, g = lane >> 5;
        double acc = 0;
        if (prefetch) {
#pragma unroll
            for (int j = 0; j < kPre; j++)
                if (g + 2 * j < nblk_used) acc += pv[j];   // same blocks, same order as the loop below
        } else if (v < kNumSums && (!(ops & LM_NOJAC) || v == kNumSums - 1)) {
#pragma unroll 8
            for (int b = g; b < nblk_used; b += 2) acc += pp[(size_t)b * kNumSums + v];
        }
        s_part[g][v] = acc;
        const int res_out = DUAL ? (ch ? sq->res_t : sq->res_new) : (sq->lm_phase == 0 ? sq->res_t : sq->res_new);
        double *cr = a.resid_carry + ((size_t)res_out * a.nseq + seq) * a.nblk;
        const double *bl = a_block_last + (size_t)seq * a.nblk;
        if (prefetch) {
            // resolve the carries across the lanes: block b starts from the last valid residual of the blocks before it
            // (fi = 0 at the top): the newest valid lane below b, found in the ballot
            const bool valid = lane < nblk_used && !is_carry(bl_pre);
            const unsigned long long vm = __ballot(valid);
            const unsigned long long below = vm & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll(below) : 0;
            const double inh = __shfl(bl_pre, src, 64);
            if (lane < nblk_used) cr[lane] = below ? inh : 0.0;
        } else {
            // per-block last residuals of the buffer that was just written -> LDS
            for (int b = lane; b < nblk_used; b += 64) s_bl[b & 255] = bl[b];
        }
        lm_sync<WAVE_ONLY>();
        if (lane < kNumSums) {
            double t = 0;
            t = s_part[0][lane] + s_part[1][lane];
            s_sum[lane] = t;
        }
        if (!prefetch && lane == 32) {  // resolve the carries: prefix "last valid" over the blocks (T fi=0 at the top)
            double run = 0;
            for (int b = 0; b < nblk_used; b++) {
                cr[b] = run;
                const double v2 = nblk_used <= 256 ? s_bl[b] : bl[b];
                if (!is_carry(v2)) run = v2;
            }
        }
    }
    lm_sync<WAVE_ONLY>();
    if (lane == 0) {
    // The serial LM logic runs on register copies of the hot state (fully unrolled 6x6 algebra): with the state
    // left in LDS every one of its ~1000 dependent accesses paid an LDS round trip (~25 us per step).
    double JtJ[36], JtF[6], X[6], Xn[6], hh[6];
#pragma unroll
    for (int i = 0; i < 36; i++) JtJ[i] = cJtJ[i];
#pragma unroll
    for (int i = 0; i < 6; i++) { JtF[i] = cJtF[i]; X[i] = cX[i]; Xn[i] = cXnew[i]; hh[i] = ch_h[i]; }
    double F = *cF, Fnew = *cFnew, F0 = *cF0, u = *cu, v = *cv;
    int eff_steps = *c_eff, res_cur = sq->res_cur, res_new = sq->res_new, res_t = sq->res_t;

    if (ops & (LM_REDUCE_CUR | LM_REDUCE_NEW)) {
        double Jn[36], Fn6[6];
        if (!(ops & LM_NOJAC)) {
            {
                int ns = 0;
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = i; j < 6; j++) Jn[i * 6 + j] = lm_rt(s_sum[ns++], a.f32);
#pragma unroll
                for (int i = 0; i < 6; i++) Fn6[i] = lm_rt(s_sum[ns++], a.f32);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {  // sign fix-ups, global_tracker.cpp:484-490
                Fn6[i + 2] = -Fn6[i + 2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    Jn[(i + 0) * 6 + j + 2] = -Jn[(i + 0) * 6 + j + 2];
                    Jn[(i + 2) * 6 + j + 4] = -Jn[(i + 2) * 6 + j + 4];
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i + 1; j < 6; j++) Jn[j * 6 + i] = Jn[i * 6 + j];
            if (ops & LM_REDUCE_CUR) {
#pragma unroll
                for (int i = 0; i < 36; i++) JtJ[i] = Jn[i];
#pragma unroll
                for (int i = 0; i < 6; i++) JtF[i] = Fn6[i];
            } else {
#pragma unroll
                for (int i = 0; i < 36; i++) cJtJnew[i] = Jn[i];
#pragma unroll
                for (int i = 0; i < 6; i++) cJtFnew[i] = Fn6[i];
            }
        }
        if (ops & LM_REDUCE_CUR) F = lm_rt(s_sum[kNumSums - 1], a.f32); else Fnew = lm_rt(s_sum[kNumSums - 1], a.f32);
        if (!DUAL) sq->pub.minimizer_evals++;
        else if (ch == 0) sq->pub.minimizer_evals += 2;   // the launch evaluated both chains
    }
    const double tau = 1e-3;
    if (ops & LM_INIT) {
        F0 = F;
        double mx = JtJ[0];
#pragma unroll
        for (int i = 1; i < 36; i++) mx = JtJ[i] > mx ? JtJ[i] : mx;  // TooN::max_element(JtJ).first
        u = tau * mx;
    }
    if (ops & LM_RESET_V) v = 2;
    if (ops & (LM_GAIN_RATIO | LM_GAIN_DIFF)) {
        double gain;
        if (ops & LM_GAIN_DIFF) gain = F - Fnew;
        else {
            double den = 0;  // (0.5*h) * (u*h - JtF)
#pragma unroll
            for (int i = 0; i < 6; i++) den += (0.5 * hh[i]) * (u * hh[i] - JtF[i]);
            gain = (F - Fnew) / den;
        }
        *cgain = gain;
        if (gain > 0) {
            F = Fnew;
#pragma unroll
            for (int i = 0; i < 6; i++) { X[i] = Xn[i]; JtF[i] = cJtFnew[i]; }
#pragma unroll
            for (int i = 0; i < 36; i++) JtJ[i] = cJtJnew[i];
            const double g = 2 * gain - 1;
            const double m = 1 - (g * g * g);
            u *= (0.33 > m ? 0.33 : m);   // std::max(0.33, ...)
            v = 2;
            eff_steps++;
            if (ops & LM_SWAP_ON_ACCEPT) { const int t = res_new; res_new = res_cur; res_cur = t; }
        } else {
            u *= v;
            v *= 2;
        }
    }
    if (ops & LM_SAVE_T) {
#pragma unroll
        for (int i = 0; i < 6; i++) sq->Xt[i] = X[i];
        sq->Ft = F; sq->F0t = F0; sq->ut = u; sq->vt = v; sq->eff_steps_t = eff_steps;
        eff_steps = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) { X[i] = lm_rt(sq->pub.V[i], a.f32); X[3 + i] = lm_rt(sq->pub.W[i], a.f32); }
    }
    if (ops & LM_PICK) {
        if (F > sq->Ft) {
#pragma unroll
            for (int i = 0; i < 6; i++) X[i] = sq->Xt[i];
            F = sq->Ft; F0 = sq->F0t; u = sq->ut; v = sq->vt; eff_steps = sq->eff_steps_t;
            const int t = res_new; res_new = res_t; res_t = t;   // ResidualNew = Rest
        }
        const int t = res_new; res_new = res_cur; res_cur = t;       // std::swap
    }
    if (ops & (LM_SOLVE_SVD | LM_SOLVE_CHOL)) {
        double ApI[36], nb[6];
#pragma unroll
        for (int i = 0; i < 36; i++) ApI[i] = JtJ[i];
#pragma unroll
        for (int i = 0; i < 6; i++) { ApI[i * 7] = lm_rt(JtJ[i * 7] + 1.0 * u, a.f32); nb[i] = -JtF[i]; }
        lm_solve6(ApI, nb, (ops & LM_SOLVE_SVD) != 0, hh, s_m, s_v);
#pragma unroll
        for (int i = 0; i < 6; i++) { hh[i] = lm_rt(hh[i], a.f32); Xn[i] = lm_rt(X[i] + hh[i], a.f32); }
    }
    if (ops & LM_PHASE_A) sq->lm_phase = 0;
    if (ops & LM_PHASE_BC) sq->lm_phase = 1;
    // (the transform of the next evaluation is set up behind this block, on two lanes: tvr_setup2)
    if (ops & LM_FINISH) {
        double L[36], Inv[36];
        chol6_r(JtJ, L);
        chol6_inverse_r(L, Inv);
#pragma unroll
        for (int i = 0; i < 3; i++) { sq->pub.V[i] = X[i]; sq->pub.W[i] = X[3 + i]; }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                sq->pub.P_V[i * 3 + j] = Inv[i * 6 + j];
                sq->pub.P_W[i * 3 + j] = Inv[(i + 3) * 6 + j + 3];
            }
        if (eff_steps > 0) {
            double nh = 0, nx = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) { nh += hh[i] * hh[i]; nx += X[i] * X[i]; }
            sq->pub.rel_error = sqrt(nh) / (sqrt(nx) + 1e-30);
            sq->pub.rel_error_score = F / F0;
        } else {
            sq->pub.rel_error = 1e20;
            sq->pub.rel_error_score = 1e20;
        }
        sq->pub.score = F;
        a.framecount[seq]++;
    }
    if (ops & LM_FINISH_KF) {
        double L[36], Inv[36];
        chol6_r(JtJ, L);
        chol6_inverse_r(L, Inv);          // RRV = Cholesky<6>(JtJ).get_inverse()
        edgehip_kf_result &o = a.kf_out[seq];
#pragma unroll
        for (int i = 0; i < 6; i++) o.X[i] = X[i];
#pragma unroll
        for (int i = 0; i < 36; i++) o.RRV[i] = Inv[i];
        o.score_ratio = F / F0; o.F = F; o.F0 = F0;
        o.evals = sq->pub.minimizer_evals;
    }
    // registers -> state
#pragma unroll
    for (int i = 0; i < 36; i++) cJtJ[i] = JtJ[i];
#pragma unroll
    for (int i = 0; i < 6; i++) { cJtF[i] = JtF[i]; cX[i] = X[i]; cXnew[i] = Xn[i]; ch_h[i] = hh[i]; }
    *cF = F; *cFnew = Fnew; *cF0 = F0; *cu = u; *cv = v;
    *c_eff = eff_steps;
    if (ch == 0) { sq->res_cur = res_cur; sq->res_new = res_new; sq->res_t = res_t; }
    }  // lane 0
    lm_sync<WAVE_ONLY>();
    if (DUAL && (ops & LM_PICK2)) {
        // "Check for the lowerst score, and use it" (global_tracker.cpp:740-752): F of the prior-init chain against the zero-init
        // chain's; the loser's residual buffer is dropped (ResidualNew = Rest), then std::swap(ResidualNew, Residual)
        if (tid == 0) {
            if (sq->F > sq->zF) {
                for (int i = 0; i < 6; i++) sq->X[i] = sq->zX[i];
                sq->F = sq->zF; sq->F0 = sq->zF0; sq->u = sq->zu; sq->v = sq->zv; sq->eff_steps = sq->z_eff_steps;
                const int t = sq->res_new; sq->res_new = sq->res_t; sq->res_t = t;
            }
            const int t = sq->res_new; sq->res_new = sq->res_cur; sq->res_cur = t;
        }
        lm_sync<WAVE_ONLY>();
    }
    if (ops & (LM_SETUP_X | LM_SETUP_XNEW)) {
        tvr_setup2(sq, (ops & LM_SETUP_XNEW) ? cXnew : cX, lane, ch != 0);
        if (!DUAL && (ops & LM_BEGIN2)) tvr_setup2(sq, sq->zX, lane, true, 2);   // the zero-init chain's first transform, lanes 2 and 3
        lm_sync<WAVE_ONLY>();
    }
    for (int i = tid; i < kWords; i += NT) gstate[i] = s_state[i];
}

__global__ __launch_bounds__(64) void k_lm_step(LmArgs a) { lm_body<false>(a, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(128) void k_lm_step2(LmArgs a) { lm_body<false, true>(a, blockIdx.x, threadIdx.x); }

// k_tvr_prepare and the step that opens a minimisation (LM_BEGIN: initial X, the transform of the first evaluation — no
// evaluation precedes it, so it reads nothing the preparation writes) in one launch: the first wave of a sequence's first
// block runs the step.  One dependent launch fewer per frame pair.
__global__ __launch_bounds__(256) void k_tvr_prepare_begin(const int32_t *__restrict__ kns, double *__restrict__ resid0,
                                                           double *__restrict__ carry0, int cap, int nblk, LmArgs l) {
    const int seq = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int kn = kns[seq];
    if ((i % kTvrBlock) == 0 && i / kTvrBlock < nblk) carry0[(size_t)seq * nblk + i / kTvrBlock] = 0.0;
    if (i < kn) resid0[(size_t)seq * cap + i] = 0.0;  // "Init residuals", global_tracker.cpp:625
    if (blockIdx.x == 0 && threadIdx.x < 64) lm_body<true>(l, seq, threadIdx.x);   // l.kn_src = kns: the state's kn_old
}

#ifdef EDGEHIP_EXPERIMENTS   // evaluation + LM step in one launch for small batches: measured no faster (EDGEHIP_PERSIST_LM)
// ---------------------------------------------------------------------------------------------------
// k_try_velrot_lm: an evaluation and the LM step that follows it in ONE launch, for small batches.  A single camera (or
// one sequence per GPU, BASELINE config 5) is bound by the chain of dependent launches of the minimiser (evaluate -> LM
// step -> evaluate ..., ~25 of them, ~8-10 us from one dependent kernel to the next whatever it does; HIP graphs do not
// remove that).  The block that finishes a sequence's evaluation last (a counter per sequence, the usual fence + atomic
// hand-over) runs the step in its first wave: half the launches.  Same code (tvr_body / lm_body), same schedule, same
// reduction order as the separate kernels, so the results are identical bit for bit.  Only for small batches: the LM
// step needs ~230 vector registers, which would cut the occupancy the bandwidth-bound evaluation lives on.
// (A fully persistent minimiser — all blocks resident, meeting at a device-memory barrier between evaluations — was
// measured slower than the launch chain: agent-scope fences between the XCDs' L2s at every barrier, 386 vs 356 us.)
// ---------------------------------------------------------------------------------------------------
template <bool REWEIGHT, bool PROCJF, bool GREC>
__global__ __launch_bounds__(kTvrThreads) void k_try_velrot_lm(TvrArgs a, LmArgs l, unsigned *cnt) {
    const int seq = blockIdx.z, tid = threadIdx.x;
    tvr_body<REWEIGHT, PROCJF, GREC>(a, seq, blockIdx.x, tid);
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        __threadfence();                                           // this block's partials are visible before its ticket is
        s_last = atomicAdd(&cnt[seq], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    __threadfence();                                               // see the partials of the blocks that came before
    lm_body<true>(l, seq, tid);
    if (tid == 0) cnt[seq] = 0;                                    // for the next launch (stream order)
}
#endif   // EDGEHIP_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------
// IMU-branch tracker: global_tracker::TryVel<double> + Calc_f_J (global_tracker.cpp:830-934, 178-219) and
// Minimizer_V<double> (:1037-1093) — translation only (the rotation was applied to the KeyLines beforehand),
// weights 1/s_rho, one residual buffer updated in place (Residuals[ikl] = |fi|).
// The reference accumulates the 9 sums + score sequentially in double; here: wave butterfly + fixed-order
// partials (deterministic; agrees to fp64 rounding, not bit for bit — tolerance stated in the test).
// ---------------------------------------------------------------------------------------------------
constexpr int kTvNum = 10;   // JtJ(0,0) (1,1) (2,2) (0,1) (0,2) (1,2), JtF[0..2], score

template <bool USE_NEW>
__global__ __launch_bounds__(256) void k_try_vel(TvrArgs a) {
    const int seq = blockIdx.z, blk = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    SeqDev *sq = a.seq + seq;
    const int kn = a.kn_old[seq];
    if (blk * 256 >= kn) return;
    const int ikl = blk * 256 + tid;
    const KlSoA &ko = a.kl_old[seq];
    double *res = a.resid + (size_t)seq * a.cap;            // residual buffer 0, updated in place
    const double marker = __longlong_as_double((long long)resid_carry_bits());
    const double *V = USE_NEW ? sq->mv_Vnew : sq->mv_V;
    const double v0 = V[0], v1 = V[1], v2 = V[2];
    double s[kTvNum];
#pragma unroll
    for (int i = 0; i < kTvNum; i++) s[i] = 0;
    // status: 0 skipped / no residual written, 2 matched (own fi), 3 evaluated but unmatched (inherits fi)
    int status = 0, mid_f = -1;
    double fi = 0;
    bool resolved = false;   // the entry held a marker: if this evaluation leaves it alone, the resolved value goes back
    double rkeep = 0, rho_own = 0, res_old = 0;
    if (ikl < kn) {
        // everything the KeyLine streams in is requested before the first use (as in k_try_velrot): with the loads inside the
        // branches the skip test, the projection and the two gathers each paid their own memory round trip
        const float nm = ldg(ko.n_m, ikl);
        const double s_rho = ldg(ko.s_rho, ikl);
        const int32_t mnum = ldg(ko.m_num, ikl);
        const double res_in = res[ikl];
        res_old = res_in;
        const float2 pm_in = ldg(ko.p_m, ikl);
        const double rho_in = ldg(ko.rho, ikl);
        const float2 klm_in = ldg(ko.m_m, ikl);
        const uint32_t fc = a.framecount[seq];
        const uint32_t mthr = a.match_num_thresh < fc ? a.match_num_thresh : fc;
        const float min_mod = sq->mv_min_mod;
        const bool skip = (min_mod > 0 && nm < min_mod) || s_rho > sq->mv_s_rho_min || (uint32_t)mnum < mthr;
        if (!skip) {
            double weight = 1;
            double rprev = res_in;
            // a marker of the evaluation before: the carry k_lmv_step resolved for this block (the pass that used to rewrite the
            // buffer between two evaluations, k_tv_resolve, now runs after the last one only)
            if (is_carry(rprev)) { rprev = a.resid_carry[(size_t)seq * a.nblk + blk]; resolved = true; }
            rkeep = rprev;
            if (rprev > a.k_huber) weight = a.k_huber / rprev;
            const float2 pm = pm_in;
            rho_own = rho_in;
            const double z_p = 1.0 / rho_own + v2;
            bool done = false;
            double f = 0, rho_p = 0, pjx = 0, pjy = 0;
            int x = 0, y = 0;
            double pix = 0, piy = 0;
            if (z_p <= 0) {
                f = (1 / s_rho) * a.max_r * weight;
                done = true;
            } else {
                rho_p = 1.0 / z_p;
                pjx = rho_p * (v0 * a.zfm - v2 * (double)pm.x) + (double)pm.x;
                pjy = rho_p * (v1 * a.zfm - v2 * (double)pm.y) + (double)pm.y;
                pix = pjx + (double)a.ppx; piy = pjy + (double)a.ppy;       // cam_model::Hom2Img
                x = x86_cvttsd2si(pix + 0.5); y = x86_cvttsd2si(piy + 0.5);
                if (x < 1 || y < 1 || x >= a.w - 1 || y >= a.h - 1) {
                    f = (1 / s_rho) * a.max_r * weight;
                    done = true;
                }
            }
            if (done) {
                s[9] = f * f;
            } else {
                status = 3;
                double dfx = 0, dfy = 0;
                f = a.max_r / s_rho;                                       // Calc_f_J: no KeyLine / no similarity
                const uint32_t fv = a.field16[(size_t)seq * a.f16stride + field16_index(x, y, a.f16tx)];
                if (fv != 0u) {
                    const int ikf = (int)fv - 1;
                    // the matched KeyLine's c_p, m_m, u_m: the 16-byte record with u_m = m_m / |m_m| recomputed by the detector's own
                    // float expressions when nothing has rotated the new edge map since detection (a.use_grec), as in k_try_velrot
                    MatchRec fr;
                    if (a.use_grec) {
                        const float4 g = ldg(a.kl_new[seq].grec, ikf);
                        fr.c_px = g.x; fr.c_py = g.y; fr.m_mx = g.z; fr.m_my = g.w;
                        const float n2m = g.z * g.z + g.w * g.w;
                        const float nmf = sqrtf(n2m);
                        fr.u_mx = g.z / nmf; fr.u_my = g.w / nmf;
                    } else {
                        fr = a.kl_new[seq].rec[ikf];
                    }
                    const float2 klm = klm_in;
                    const double p_n2 = (double)(nm * nm);                 // Test_f_k
                    const double p_esc = (double)(klm.x * fr.m_mx + klm.y * fr.m_my);
                    if (!(fabs(p_esc - p_n2) > a.match_thresh * p_n2)) {
                        const double dx = pix - (double)fr.c_px, dy = piy - (double)fr.c_py;
                        fi = dx * (double)fr.u_mx + dy * (double)fr.u_my;
                        dfx = (double)fr.u_mx / s_rho;
                        dfy = (double)fr.u_my / s_rho;
                        f = fi / s_rho;
                        mid_f = ikf;
                        status = 2;
                    }
                }
                f *= weight;
                const double jx = rho_p * a.zfm * dfx * weight;
                const double jy = rho_p * a.zfm * dfy * weight;
                const double jz = -rho_p * (pjx * dfx + pjy * dfy) * weight;
                s[0] = jx * jx; s[1] = jy * jy; s[2] = jz * jz; s[3] = jx * jy; s[4] = jx * jz; s[5] = jy * jz;
                s[6] = jx * f; s[7] = jy * f; s[8] = jz * f; s[9] = f * f;
            }
        }
    }
    // Residuals[ikl] = fabs(fi) with fi only refreshed by a match: "last valid" propagation as in k_try_velrot
    __shared__ double s_wlast[4];
    __shared__ int s_whas[4];
    {
        const double afi = fabs(fi);
        const unsigned long long vmask = __ballot(status == 2);
        const unsigned long long below = vmask & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __clzll(below) : 0;
        const double inh = __shfl(afi, src, 64);
        const int top = vmask ? 63 - __clzll(vmask) : 0;
        const double wl = __shfl(afi, top, 64);
        if (lane == 0) { s_whas[wave] = vmask != 0; s_wlast[wave] = wl; }
        __syncthreads();
        if (status == 3) {
            double v = marker;
            bool have = false;
            if (below) { v = inh; have = true; }
            for (int pw = wave - 1; pw >= 0 && !have; pw--)
                if (s_whas[pw]) { v = s_wlast[pw]; have = true; }
            res[ikl] = v;
        } else if (status == 2) {
            res[ikl] = afi;
        } else if (ikl < kn) {
            res[ikl] = resolved ? rkeep : res_old;   // unchanged entries are rewritten too: whole-line stores (a partial line is a read-modify-write)
        }
        if (tid == 0) {
            double bl = marker;
            for (int pw = 3; pw >= 0; pw--)
                if (s_whas[pw]) { bl = s_wlast[pw]; break; }
            a.block_last[(size_t)seq * a.nblk + blk] = bl;
        }
    }
    if (ikl < kn) ko.m_id_f[ikl] = mid_f;                      // kl.m_id_f = -1 / match, every evaluation
    // the minimisation's last evaluation (the one whose m_id_f FordwardMatch will read) also posts FordwardMatch's arbitration key
    // of the matched new KeyLine, as k_try_velrot's does: no k_fwd_key pass
    if (a.fwd_key && mid_f >= 0) atomicMax(&a.fwd_key[(size_t)seq * a.cap + mid_f], ord_bits(rho_own));
    __shared__ double s_red[4][16];
    {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = i < kTvNum ? s[i] : 0.0;
        const int idx = wave_reduce16(v, lane);   // the xor butterfly's pairs and order, a sixth of its shuffles
        if ((lane & 3) == 0) s_red[wave][idx] = v[0];
    }
    __syncthreads();
    if (tid < kTvNum)
        a.partials[((size_t)seq * a.nblk + blk) * kNumSums + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// markers left by k_try_vel -> |fi| of the last matched KeyLine of the preceding blocks (carry computed by k_lmv_step)
__global__ __launch_bounds__(256) void k_tv_resolve(double *__restrict__ resid, const double *__restrict__ carry,
                                                    const int32_t *__restrict__ kns, int cap, int nblk) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kns[seq]) return;
    double *r = resid + (size_t)seq * cap;
    if (is_carry(r[i])) r[i] = carry[(size_t)seq * nblk + blockIdx.x];
}

// util::Matrix3x3Inv (include/UtilLib/toon_util.h:32-41): adjugate / TooN::determinant, the latter by Gaussian
// elimination with partial pivoting (TooN/determinant.h:91-146)
__device__ inline double det3_ge(const double Ain[9]) {
    double A[9];
    for (int i = 0; i < 9; i++) A[i] = Ain[i];
    double det = 1;
    for (int i = 0; i < 3; i++) {
        int argmax = i;
        double maxval = fabs(A[i * 3 + i]);
        for (int ii = i + 1; ii < 3; ii++) {
            const double v = fabs(A[ii * 3 + i]);
            if (v > maxval) { maxval = v; argmax = ii; }
        }
        const double pivot = A[argmax * 3 + i];
        if (argmax != i) {
            det *= -1;
            for (int j = i; j < 3; j++) { const double t = A[i * 3 + j]; A[i * 3 + j] = A[argmax * 3 + j]; A[argmax * 3 + j] = t; }
        }
        det *= A[i * 3 + i];
        if (det == 0) return 0;
        for (int u = i + 1; u < 3; u++) {
            const double factor = A[u * 3 + i] / pivot;
            for (int j = i + 1; j < 3; j++) A[u * 3 + j] = A[u * 3 + j] - factor * A[i * 3 + j];
        }
    }
    return det;
}
__device__ inline void mat3_inv(const double A[9], double B[9]) {
    B[0] = A[8] * A[4] - A[7] * A[5]; B[1] = -(A[8] * A[1] - A[7] * A[2]); B[2] = A[5] * A[1] - A[4] * A[2];
    B[3] = -(A[8] * A[3] - A[6] * A[5]); B[4] = A[8] * A[0] - A[6] * A[2]; B[5] = -(A[5] * A[0] - A[3] * A[2]);
    B[6] = A[7] * A[3] - A[6] * A[4]; B[7] = -(A[7] * A[0] - A[6] * A[1]); B[8] = A[4] * A[0] - A[3] * A[1];
    const double det = det3_ge(A);
    for (int i = 0; i < 9; i++) B[i] = B[i] / det;
}

enum LmvOps : unsigned { LMV_BEGIN = 1, LMV_REDUCE_CUR = 2, LMV_REDUCE_NEW = 4, LMV_GAIN = 8, LMV_SOLVE = 16, LMV_FINISH = 32 };

__global__ __launch_bounds__(64) void k_lmv_step(SeqDev *seqs, const double *__restrict__ partials, const double *__restrict__ block_last,
                                                 double *__restrict__ carry, const int32_t *__restrict__ kn_old, int nblk, unsigned ops) {
    const int seq = blockIdx.x, lane = threadIdx.x;
    SeqDev *sq = seqs + seq;
    const int kn = kn_old[seq];
    const int nblk_used = (kn + 255) / 256;
    __shared__ double s_sum[kTvNum];
    // One wave per sequence and nothing but latency: everything the step reads is requested before anything is used or stored —
    // the blocks' partial sums (all of a value's up to 64 loads in flight, then summed in block order as before), their last
    // residuals, the state — and the state is written back once at the end.  (Value by value and field by field the step paid
    // ~50 dependent memory round trips: 16 us for a single camera.)
    constexpr int kPre = 64;
    const bool red = (ops & (LMV_REDUCE_CUR | LMV_REDUCE_NEW)) != 0;
    const bool pre = nblk_used <= kPre;
    double pv[kPre];
    double bl = 0;
    if (red && pre) {
        if (lane < kTvNum) {
#pragma unroll
            for (int b = 0; b < kPre; b++) pv[b] = b < nblk_used ? partials[((size_t)seq * nblk + b) * kNumSums + lane] : 0.0;
        }
        if (lane < nblk_used) bl = block_last[(size_t)seq * nblk + lane];
    }
    double V[3], Vn[3], h[3], JtJ[9], JtF[3], JtJn[9], JtFn[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { V[i] = sq->mv_V[i]; Vn[i] = sq->mv_Vnew[i]; h[i] = sq->mv_h[i]; JtF[i] = sq->mv_JtF[i]; JtFn[i] = sq->mv_JtFnew[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { JtJ[i] = sq->mv_JtJ[i]; JtJn[i] = sq->mv_JtJnew[i]; }
    double F = sq->mv_F, Fnew = sq->mv_Fnew, u = sq->mv_u, v = sq->mv_v;
    if (red) {
        if (lane < kTvNum) {
            double acc = 0;
            if (pre) {
#pragma unroll
                for (int b = 0; b < kPre; b++) if (b < nblk_used) acc += pv[b];
            } else {
                for (int b = 0; b < nblk_used; b++) acc += partials[((size_t)seq * nblk + b) * kNumSums + lane];
            }
            s_sum[lane] = acc;
        }
        // carries: |fi| of the last matched KeyLine before each block (0 at the top: double fi=0)
        if (pre) {
            const bool valid = lane < nblk_used && !is_carry(bl);
            const unsigned long long vm = __ballot(valid);
            const unsigned long long below = vm & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll(below) : 0;
            const double inh = __shfl(bl, src, 64);
            if (lane < nblk_used) carry[(size_t)seq * nblk + lane] = below ? inh : 0.0;
        } else if (lane == 32) {
            double run = 0;
            for (int b = 0; b < nblk_used; b++) {
                carry[(size_t)seq * nblk + b] = run;
                const double v2 = block_last[(size_t)seq * nblk + b];
                if (!is_carry(v2)) run = v2;
            }
        }
    }
    __syncthreads();
    if (lane != 0) return;
    const double tau = 1e-3;
    if (ops & LMV_BEGIN) v = 2;
    if (red) {
        double *J = (ops & LMV_REDUCE_CUR) ? JtJ : JtJn;
        double *Fv = (ops & LMV_REDUCE_CUR) ? JtF : JtFn;
        J[0] = s_sum[0]; J[4] = s_sum[1]; J[8] = s_sum[2];
        J[1] = J[3] = s_sum[3]; J[2] = J[6] = s_sum[4]; J[5] = J[7] = s_sum[5];
        Fv[0] = s_sum[6]; Fv[1] = s_sum[7]; Fv[2] = s_sum[8];
        if (ops & LMV_REDUCE_CUR) {
            F = s_sum[9];
            double mx = JtJ[0];
#pragma unroll
            for (int i = 1; i < 9; i++) mx = JtJ[i] > mx ? JtJ[i] : mx;   // TooN::max_element(JtJ).first
            u = tau * mx;
        } else {
            Fnew = s_sum[9];
        }
    }
    if (ops & LMV_GAIN) {
        double den = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) den += (0.5 * h[i]) * (u * h[i] - JtF[i]);
        const double gain = (F - Fnew) / den;
        if (gain > 0) {
            F = Fnew;
#pragma unroll
            for (int i = 0; i < 3; i++) { V[i] = Vn[i]; JtF[i] = JtFn[i]; }
#pragma unroll
            for (int i = 0; i < 9; i++) JtJ[i] = JtJn[i];
            const double g = 2 * gain - 1;
            const double m = 1 - (g * g * g);
            u *= (0.33 > m ? 0.33 : m);
            v = 2;
        } else {
            u *= v;
            v *= 2;
        }
    }
    if (ops & LMV_SOLVE) {
        double ApI[9], Inv[9];
#pragma unroll
        for (int i = 0; i < 9; i++) ApI[i] = JtJ[i] + ((i % 4 == 0) ? 1.0 * u : 0.0);
        mat3_inv(ApI, Inv);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double d = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) d += Inv[i * 3 + k] * (-JtF[k]);
            h[i] = d;
            Vn[i] = V[i] + d;
        }
    }
    double RVel[9];
    if (ops & LMV_FINISH) mat3_inv(JtJ, RVel);
    // ---- stores ----
#pragma unroll
    for (int i = 0; i < 3; i++) { sq->mv_V[i] = V[i]; sq->mv_Vnew[i] = Vn[i]; sq->mv_h[i] = h[i]; sq->mv_JtF[i] = JtF[i]; sq->mv_JtFnew[i] = JtFn[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { sq->mv_JtJ[i] = JtJ[i]; sq->mv_JtJnew[i] = JtJn[i]; }
    sq->mv_F = F; sq->mv_Fnew = Fnew; sq->mv_u = u; sq->mv_v = v;
    if (ops & LMV_FINISH) {
#pragma unroll
        for (int i = 0; i < 9; i++) sq->mv_RVel[i] = RVel[i];
    }
}

// standalone evaluation helper: X given by the host -> setup
__global__ void k_tvr_setup_from_host(SeqDev *seqs, const double *__restrict__ X, const double *__restrict__ smin, int nseq,
                                      int res_in, int res_out) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    double x[6];
    for (int i = 0; i < 6; i++) x[i] = X[seq * 6 + i];
    tvr_setup(sq, x);
    sq->s_rho_min_eval = smin[seq];
    sq->res_cur = res_in < 0 ? 0 : res_in;
    sq->res_new = res_out;
    sq->lm_phase = 1;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int quantile_enqueue(edgehip_ctx *c, int slot, double smin, double smax, double pct, int nbins, bool frame_begins, int retune_slot) {
    RetuneArgs rt = {};
    if (retune_slot >= 0) {
        rt.seqa = c->seqa; rt.histo = c->histo; rt.retuned_out = c->retuned_slot + (size_t)retune_slot * c->plan.nseq;
        rt.knum = c->p.track_points; rt.nbins = c->p.qcut_nbins;
    }
    ProfScope ps(c, PROF_B_QUANTILE);
    if (nbins < 1 || nbins > 256) { set_error("quantile: 1 <= nbins <= 256"); return EDGEHIP_ERR_ARG; }
    if (c->plan.nseq <= 64)
        hipLaunchKernelGGL(k_quantile<1024>, dim3(c->plan.nseq), dim3(1024), 0, c->stream, kldev(c, slot),
                           c->kn_slot + (size_t)slot * c->plan.nseq, c->seq, smin, smax, pct, nbins, frame_begins ? c->t_src : (const double *)nullptr, c->p.config_fps, rt);
    else
        hipLaunchKernelGGL(k_quantile<256>, dim3(c->plan.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                           c->kn_slot + (size_t)slot * c->plan.nseq, c->seq, smin, smax, pct, nbins, frame_begins ? c->t_src : (const double *)nullptr, c->p.config_fps, rt);
    EH_LAUNCH_CHECK();
    return 0;
}

int build_field_enqueue(edgehip_ctx *c, int slot, int radius, float min_mod, bool clear_fwd) {
    c->fwd_cleared = false;
    ProfScope ps(c, PROF_B_FIELD);
    const DevicePlan &pl = c->plan;
    if (radius < 1 || radius > 255) { set_error("build_field: 1 <= radius <= 255"); return EDGEHIP_ERR_ARG; }
    c->field_radius = radius;
    const int ntx = (pl.w + FT - 1) / FT, nty = (pl.h + FT - 1) / FT;
    if (c->field_mode == 0 && ntx * nty <= kMaxTiles) {
        hipLaunchKernelGGL(k_field_bin, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                           c->kn_slot + (size_t)slot * pl.nseq, c->retuned_slot + (size_t)slot * pl.nseq, c->bin_cnt, c->bins,
                           pl.w, pl.h, radius, min_mod, ntx, nty, pl.cap, clear_fwd ? c->fwd_key : nullptr, clear_fwd ? c->fwd_win : nullptr);
        c->fwd_cleared = clear_fwd;
        hipLaunchKernelGGL(k_field_raster, dim3(ntx, nty, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot), c->bin_cnt,
                           c->bins, c->field, c->field16, pl.f16stride, pl.f16tx, c->p.debug_planes ? 1 : 0, pl.w, pl.h,
                           pl.fstride, pl.ftx, radius, ntx, pl.cap);
        c->field32_valid = c->p.debug_planes != 0;
        EH_LAUNCH_CHECK();
        return 0;
    } else if (c->field_mode == 2 || c->field_mode == 0) {  // mask-scan tiles (any image size)
        hipLaunchKernelGGL(k_field_tiles, dim3((pl.w + FT - 1) / FT, (pl.h + FT - 1) / FT, pl.nseq), dim3(256), 0, c->stream,
                           kldev(c, slot), maskof(c, slot), c->retuned_slot + (size_t)slot * pl.nseq, c->field, pl.w, pl.h,
                           pl.fstride, pl.ftx, radius, min_mod);
    } else {
#ifdef EDGEHIP_EXPERIMENTS   // reference-shaped scatter with global atomics (EDGEHIP_FIELD_MODE=1)
        EH_CHECK(hipMemsetAsync(c->field, 0xFF, sizeof(uint32_t) * pl.nseq * pl.fstride, c->stream));
        const long long threads = (long long)pl.cap * 2 * radius;
        hipLaunchKernelGGL(k_field_scatter, dim3((unsigned)((threads + 255) / 256), 1, pl.nseq), dim3(256), 0, c->stream,
                           kldev(c, slot), c->kn_slot + (size_t)slot * pl.nseq, c->retuned_slot + (size_t)slot * pl.nseq,
                           c->field, pl.w, pl.h, pl.fstride, pl.ftx, radius, min_mod);
#else
        set_error("build_field: unknown field mode");
        return EDGEHIP_ERR_STATE;
#endif
    }
    EH_LAUNCH_CHECK();
    // the other two builders produce the {dist, ikl} field; the tracker's index plane is derived from it
    hipLaunchKernelGGL(k_field_to16, dim3((unsigned)((pl.f16stride + 255) / 256), 1, pl.nseq), dim3(256), 0, c->stream, c->field,
                       c->field16, pl.w, pl.h, pl.fstride, pl.ftx, pl.f16stride, pl.f16tx);
    c->field32_valid = true;
    EH_LAUNCH_CHECK();
    return 0;
}

int tvr_prepare_enqueue(edgehip_ctx *c, int slot_old, unsigned begin_ops) {
    ProfScope ps(c, PROF_B_PREP);
    const DevicePlan &pl = c->plan;
    if (begin_ops) {
        LmArgs l;
        l.seq = c->seq; l.partials = c->partials; l.block_last = c->block_last; l.resid_carry = c->resid_carry;
        l.framecount = c->framecount + (size_t)c->fc_index * pl.nseq;
        l.nblk = c->nblk_tvr; l.nseq = pl.nseq; l.ops = begin_ops; l.init_type = c->p.tracker_init_type;
        l.kf_in = c->kf_req_dev; l.kf_out = c->kf_res_dev;
        l.kn_src = c->kn_slot + (size_t)slot_old * pl.nseq;
        l.f32 = c->tracker_f32 ? 1 : 0;
        hipLaunchKernelGGL(k_tvr_prepare_begin, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, l.kn_src, c->resid,
                           c->resid_carry, pl.cap, c->nblk_tvr, l);
        EH_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_tvr_prepare, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot_old),
                       c->kn_slot + (size_t)slot_old * pl.nseq, c->P0, c->resid, c->resid_carry, c->seq, pl.cap,
                       c->nblk_tvr, pl.zfm);
    EH_LAUNCH_CHECK();
    return 0;
}

static TvrArgs make_tvr_args(edgehip_ctx *c, int slot_new, int slot_old, double match_thresh, double k_huber,
                             uint32_t match_num_thresh, int write_mid) {
    const DevicePlan &pl = c->plan;
    TvrArgs a;
    a.kl_old = kldev(c, slot_old); a.kl_new = kldev(c, slot_new);
    a.kn_old = c->kn_slot + (size_t)slot_old * pl.nseq;
    a.field16 = c->field16; a.f16stride = pl.f16stride; a.f16tx = pl.f16tx; a.P0 = c->P0; a.resid = c->resid; a.resid_carry = c->resid_carry;
    a.block_last = c->block_last; a.partials = c->partials; a.seq = c->seq;
    a.block_last_z = c->block_last + (size_t)pl.nseq * c->nblk_tvr;
    a.partials_z = c->partials + (size_t)pl.nseq * c->nblk_tvr * kNumSums;
    a.framecount = c->framecount + (size_t)c->fc_index * pl.nseq;
    a.w = pl.w; a.h = pl.h; a.cap = pl.cap; a.nblk = c->nblk_tvr; a.nseq = pl.nseq;
    a.zfm = pl.zfm; a.inv_zfm = 1.0 / pl.zfm; a.max_r = (double)c->field_radius; a.match_thresh = match_thresh; a.k_huber = k_huber; a.inv_k_huber = 1.0 / k_huber;
    a.ppx = pl.ppx; a.ppy = pl.ppy; a.match_num_thresh = match_num_thresh; a.write_mid = write_mid;
    a.use_grec = c->grec_ok[slot_new] && !c->no_grec;
    a.kf = nullptr; a.kf_match_mod = a.kf_match_cang = a.kf_rho_tol = 0;
    a.fwd_key = (write_mid && c->fwd_key_in_tvr) ? c->fwd_key : nullptr;
    return a;
}

static int launch_tvr(edgehip_ctx *c, const TvrArgs &a, bool reweight, bool procjf) {
    ProfScope ps(c, PROF_B_TRYVELROT);
    dim3 g(c->nblk_tvr, 1, c->plan.nseq), b(kTvrThreads);
    // whole batches: the reweighted evaluation with two KeyLines per thread (tvr_rw2_body); a few sequences keep one KeyLine per
    // thread (half the blocks would leave most CUs idle, and a block's own latency is what a single camera waits for)
#ifdef EDGEHIP_EXPERIMENTS
    if (reweight && procjf && c->tvr_rw2 && (size_t)c->nblk_tvr * c->plan.nseq >= (size_t)c->tvr_rw2) {
        dim3 g2((c->nblk_tvr + 1) / 2, 1, c->plan.nseq);
        if (a.use_grec) hipLaunchKernelGGL((k_try_velrot_rw2<true>), g2, b, 0, c->stream, a);
        else hipLaunchKernelGGL((k_try_velrot_rw2<false>), g2, b, 0, c->stream, a);
        EH_LAUNCH_CHECK();
        return 0;
    }
#endif
#ifdef EDGEHIP_EXPERIMENTS
    // occupancy experiment (tools/experiments): unused dynamic LDS per block caps the resident blocks per CU
    static const size_t dyn_lds = getenv("EDGEHIP_TVR_LDS") ? (size_t)atoi(getenv("EDGEHIP_TVR_LDS")) : 0;
#else
    constexpr size_t dyn_lds = 0;
#endif
    if (c->tracker_f32) {   // Minimizer_RV<float> (edgehip_set_tracker_precision)
#define EH_TVF(RW, JF)                                                                                                   \
    do {                                                                                                                 \
        if (a.use_grec) hipLaunchKernelGGL((k_try_velrot_f32<RW, JF, true>), g, b, 0, c->stream, a);                      \
        else hipLaunchKernelGGL((k_try_velrot_f32<RW, JF, false>), g, b, 0, c->stream, a);                                \
    } while (0)
        if (reweight && procjf) EH_TVF(true, true);
        else if (reweight) EH_TVF(true, false);
        else if (procjf) EH_TVF(false, true);
        else EH_TVF(false, false);
#undef EH_TVF
        EH_LAUNCH_CHECK();
        return 0;
    }
    if (a.use_grec) {
        if (reweight && procjf) hipLaunchKernelGGL((k_try_velrot<true, true, true>), g, b, dyn_lds, c->stream, a);
        else if (reweight) hipLaunchKernelGGL((k_try_velrot<true, false, true>), g, b, dyn_lds, c->stream, a);
        else if (procjf) hipLaunchKernelGGL((k_try_velrot<false, true, true>), g, b, dyn_lds, c->stream, a);
        else hipLaunchKernelGGL((k_try_velrot<false, false, true>), g, b, dyn_lds, c->stream, a);
    } else {
        if (reweight && procjf) hipLaunchKernelGGL((k_try_velrot<true, true, false>), g, b, dyn_lds, c->stream, a);
        else if (reweight) hipLaunchKernelGGL((k_try_velrot<true, false, false>), g, b, dyn_lds, c->stream, a);
        else if (procjf) hipLaunchKernelGGL((k_try_velrot<false, true, false>), g, b, dyn_lds, c->stream, a);
        else hipLaunchKernelGGL((k_try_velrot<false, false, false>), g, b, dyn_lds, c->stream, a);
    }
    EH_LAUNCH_CHECK();
    return 0;
}

static int launch_lm(edgehip_ctx *c, int slot_new, unsigned ops, bool two_chains = false) {
    ProfScope ps(c, PROF_B_LMSTEP);
    LmArgs a;
    a.seq = c->seq; a.partials = c->partials; a.block_last = c->block_last; a.resid_carry = c->resid_carry;
    a.block_last_z = c->block_last + (size_t)c->plan.nseq * c->nblk_tvr;
    a.partials_z = c->partials + (size_t)c->plan.nseq * c->nblk_tvr * kNumSums;
    a.framecount = c->framecount + (size_t)c->fc_index * c->plan.nseq;
    a.nblk = c->nblk_tvr; a.nseq = c->plan.nseq; a.ops = ops; a.init_type = c->p.tracker_init_type;
    a.kf_in = c->kf_req_dev; a.kf_out = c->kf_res_dev;
    a.f32 = c->tracker_f32 ? 1 : 0;
    if (two_chains) hipLaunchKernelGGL(k_lm_step2, dim3(c->plan.nseq), dim3(128), 0, c->stream, a);
    else hipLaunchKernelGGL(k_lm_step, dim3(c->plan.nseq), dim3(64), 0, c->stream, a);
    EH_LAUNCH_CHECK();
    return 0;
}

// one launch for evaluation i of both initialisation chains (tvr2_body)
static int launch_tvr2(edgehip_ctx *c, const TvrArgs &a, bool procjf) {
    ProfScope ps(c, PROF_B_TRYVELROT2);
    dim3 g(c->nblk_tvr, 1, c->plan.nseq), b(kTvrThreads);
    if (c->tracker_f32) {
        if (a.use_grec) {
            if (procjf) hipLaunchKernelGGL((k_try_velrot2_f32<true, true>), g, b, 0, c->stream, a);
            else hipLaunchKernelGGL((k_try_velrot2_f32<false, true>), g, b, 0, c->stream, a);
        } else {
            if (procjf) hipLaunchKernelGGL((k_try_velrot2_f32<true, false>), g, b, 0, c->stream, a);
            else hipLaunchKernelGGL((k_try_velrot2_f32<false, false>), g, b, 0, c->stream, a);
        }
        EH_LAUNCH_CHECK();
        return 0;
    }
    if (a.use_grec) {
        if (procjf) hipLaunchKernelGGL((k_try_velrot2<true, true>), g, b, 0, c->stream, a);
        else hipLaunchKernelGGL((k_try_velrot2<false, true>), g, b, 0, c->stream, a);
    } else {
        if (procjf) hipLaunchKernelGGL((k_try_velrot2<true, false>), g, b, 0, c->stream, a);
        else hipLaunchKernelGGL((k_try_velrot2<false, false>), g, b, 0, c->stream, a);
    }
    EH_LAUNCH_CHECK();
    return 0;
}

// m_id_f >= 0 over a KeyLine list (kfvo::OptimizePosGT's count after the minimisation, kfvo.cpp:76-82)
__global__ __launch_bounds__(256) void k_count_forward(const KlSoA *kls, const int32_t *__restrict__ kns, edgehip_kf_result *out) {
    __shared__ int s_n[4];
    const int seq = blockIdx.x, kn = kns[seq];
    int n = 0;
    for (int i = threadIdx.x; i < kn; i += 256) n += kls[seq].m_id_f[i] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) out[seq].mnum = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}

// kfvo::Minimizer_RV_KF<double,false> (kfvo.cpp:1679-1825): the schedule is the reweighted Levenberg-Marquardt loop of
// Minimizer_RV without an initialisation phase — every evaluation with ReWeight and ProcJF, Cholesky solves throughout.
int minimizer_kf_enqueue(edgehip_ctx *c, int slot_kf, int slot_cur, double match_mod, double match_ang, double rho_tol, int iter_max,
                         double reweight_distance, uint32_t match_num_thresh) {
    int e;
    c->fc_index = slot_kf;   // not read by the KF evaluation, not written by LM_FINISH_KF
#define EH_TRY(x) if ((e = (x)) != 0) return e
    EH_TRY(rec_refresh_enqueue(c, slot_kf));   // a key frame has been the old slot of a later frame: its gradients were turned
    EH_TRY(tvr_prepare_enqueue(c, slot_cur));
    auto eval = [&](bool last) -> int {
        TvrArgs a = make_tvr_args(c, slot_kf, slot_cur, 0.0, reweight_distance, match_num_thresh, last ? 1 : 0);
        a.kf = c->kf_req_dev; a.kf_match_mod = match_mod; a.kf_match_cang = cos(match_ang); a.kf_rho_tol = rho_tol;
        a.use_grec = 0;
        ProfScope ps(c, PROF_B_TRYVELROT);
        hipLaunchKernelGGL(k_try_velrot_kf, dim3(c->nblk_tvr, 1, c->plan.nseq), dim3(kTvrThreads), 0, c->stream, a);
        EH_LAUNCH_CHECK();
        return 0;
    };
    const int M = iter_max;
    EH_TRY(launch_lm(c, slot_kf, LM_BEGIN_KF | LM_SETUP_X | LM_PHASE_BC));
    EH_TRY(eval(M <= 0));
    {
        unsigned ops = LM_REDUCE_CUR | LM_INIT | LM_RESET_V;
        if (M > 0) ops |= LM_SOLVE_CHOL | LM_SETUP_XNEW; else ops |= LM_FINISH_KF;
        EH_TRY(launch_lm(c, slot_kf, ops));
    }
    for (int it = 0; it < M; it++) {
        EH_TRY(eval(it == M - 1));
        unsigned ops = LM_REDUCE_NEW | LM_GAIN_RATIO | LM_SWAP_ON_ACCEPT;
        if (it < M - 1) ops |= LM_SOLVE_CHOL | LM_SETUP_XNEW; else ops |= LM_FINISH_KF;
        EH_TRY(launch_lm(c, slot_kf, ops));
    }
    hipLaunchKernelGGL(k_count_forward, dim3(c->plan.nseq), dim3(256), 0, c->stream, kldev(c, slot_cur),
                       c->kn_slot + (size_t)slot_cur * c->plan.nseq, c->kf_res_dev);
    EH_LAUNCH_CHECK();
#undef EH_TRY
    return 0;
}

// Minimizer_RV<double,false>: the static evaluation/step schedule of global_tracker.cpp:631-816
int minimizer_enqueue(edgehip_ctx *c, int slot_new, int slot_old, int fc_index) {
    const edgehip_params &p = c->p;
    c->fc_index = fc_index;
    int e;
    if ((e = rec_refresh_enqueue(c, slot_new))) return e;   // (never on the frame path: the new slot's KeyLines were detected this frame)
    // A few sequences: the step that opens the minimisation rides on the preparation's launch (one dependent launch fewer).  Whole
    // batches: the step's ~230 registers cost the preparation more than the launch saves (measured at 1024 sequences: +90 us against -26).
    // Small batches: an evaluation is held back until the LM step that follows it is known, and both go out as one launch.
    const bool fuse = c->persist_lm_max > 0 && c->plan.nseq <= c->persist_lm_max;
    // TrackerInitType = 2: the zero-init and the prior-init chain advance in the same launches (k_try_velrot2 / k_lm_step2)
    const bool two_chains = p.tracker_init_type >= 2 && c->dual_init && !fuse;
    const unsigned begin_ops = two_chains ? (LM_BEGIN | LM_BEGIN2 | LM_SETUP_X | LM_PHASE_BC)
                                          : (LM_BEGIN | LM_SETUP_X | (p.tracker_init_type >= 2 ? LM_PHASE_A : LM_PHASE_BC));
    const bool begin_rides = c->plan.nseq <= 64;
    if ((e = tvr_prepare_enqueue(c, slot_old, begin_rides ? begin_ops : 0u))) return e;
    if (!begin_rides && (e = edgehip::launch_lm(c, slot_new, begin_ops))) return e;
    if (c->fwd_key_in_tvr && !c->fwd_cleared) EH_CHECK(hipMemsetAsync(c->fwd_key, 0, sizeof(unsigned long long) * (size_t)c->plan.nseq * c->plan.cap, c->stream));
    const int I = p.tracker_init_iter_num, M = p.tracker_iter_num;
    const int total_evals = (p.tracker_init_type >= 2 ? 2 * (1 + (I > 0 ? I : 0)) : 0) + 1 + (M > 0 ? M : 0);
    int evals = 0;
    struct { bool on, rw, jf; int write_mid; } held = {false, false, false, 0};
    auto eval = [&](bool rw, bool jf) -> int {
        evals++;
        if (fuse) {
            held = {true, rw, jf, evals == total_evals};
            return 0;
        }
        TvrArgs a = make_tvr_args(c, slot_new, slot_old, p.tracker_match_thresh, p.reweight_distance, p.match_num_thresh,
                                  evals == total_evals);
        return launch_tvr(c, a, rw, jf);
    };
    auto launch_lm = [&](edgehip_ctx *cc, int sn, unsigned ops) -> int {   // shadows the free function
        if (!held.on) return edgehip::launch_lm(cc, sn, ops);
#ifndef EDGEHIP_EXPERIMENTS
        set_error("minimizer: the fused evaluation + step launch is an EXPERIMENTS build option");
        return EDGEHIP_ERR_STATE;
#else
        held.on = false;
        ProfScope ps(c, PROF_B_MINIMIZER);
        TvrArgs a = make_tvr_args(c, slot_new, slot_old, p.tracker_match_thresh, p.reweight_distance, p.match_num_thresh, held.write_mid);
        LmArgs l;
        l.seq = c->seq; l.partials = c->partials; l.block_last = c->block_last; l.resid_carry = c->resid_carry;
        l.framecount = c->framecount + (size_t)c->fc_index * c->plan.nseq;
        l.nblk = c->nblk_tvr; l.nseq = c->plan.nseq; l.ops = ops; l.init_type = c->p.tracker_init_type;
        l.kf_in = nullptr; l.kf_out = nullptr;
        dim3 g(c->nblk_tvr, 1, c->plan.nseq), b(kTvrThreads);
#define EH_TVLM(RW, JF)                                                                                                     \
    do {                                                                                                                   \
        if (a.use_grec) hipLaunchKernelGGL((k_try_velrot_lm<RW, JF, true>), g, b, 0, c->stream, a, l, c->sync_cnt);        \
        else hipLaunchKernelGGL((k_try_velrot_lm<RW, JF, false>), g, b, 0, c->stream, a, l, c->sync_cnt);                  \
    } while (0)
        if (held.rw && held.jf) EH_TVLM(true, true);
        else if (held.rw) EH_TVLM(true, false);
        else if (held.jf) EH_TVLM(false, true);
        else EH_TVLM(false, false);
#undef EH_TVLM
        EH_LAUNCH_CHECK();
        return 0;
#endif
    };
#define EH_TRY(x) if ((e = (x)) != 0) return e
    if (two_chains) {
        // evaluation i of both chains per launch; the step kernel runs the same operations on both chains' locals and, behind
        // the last one, picks the better end point (the tails LM_SAVE_T / LM_PICK of the launch chain below)
        auto eval2 = [&](bool jf) -> int {
            evals += 2;
            TvrArgs a = make_tvr_args(c, slot_new, slot_old, p.tracker_match_thresh, p.reweight_distance, p.match_num_thresh, 0);
            return launch_tvr2(c, a, jf);
        };
        EH_TRY(eval2(true));
        unsigned ops = LM_REDUCE_CUR | LM_INIT | LM_RESET_V;
        ops |= I > 0 ? (LM_SOLVE_SVD | LM_SETUP_XNEW) : (LM_PICK2 | LM_SETUP_X);
        EH_TRY(edgehip::launch_lm(c, slot_new, ops, true));
        for (int i = 0; i < I; i++) {
            const bool last = (i == I - 1);
            EH_TRY(eval2(!last));
            ops = LM_REDUCE_NEW | (last ? (LM_NOJAC | LM_GAIN_DIFF) : LM_GAIN_RATIO);
            ops |= !last ? (LM_SOLVE_SVD | LM_SETUP_XNEW) : (LM_PICK2 | LM_SETUP_X);
            EH_TRY(edgehip::launch_lm(c, slot_new, ops, true));
        }
    } else if (p.tracker_init_type >= 2) {
        for (int trial = 0; trial < 2; trial++) {
            const unsigned phase = trial == 0 ? LM_PHASE_A : LM_PHASE_BC;
            (void)phase;   // trial 0: LM_BEGIN | LM_SETUP_X | LM_PHASE_A went out with tvr_prepare_enqueue
            EH_TRY(eval(false, true));
            unsigned ops = LM_REDUCE_CUR | LM_INIT | (trial == 1 ? LM_RESET_V : 0);
            if (I > 0) ops |= LM_SOLVE_SVD | LM_SETUP_XNEW;
            unsigned tail = trial == 0 ? (LM_SAVE_T | LM_SETUP_X | LM_PHASE_BC) : (LM_PICK | LM_SETUP_X);
            if (I <= 0) ops |= tail;
            EH_TRY(launch_lm(c, slot_new, ops));
            for (int i = 0; i < I; i++) {
                const bool last = (i == I - 1);
                EH_TRY(eval(false, !last));
                ops = LM_REDUCE_NEW | (last ? (LM_NOJAC | LM_GAIN_DIFF) : LM_GAIN_RATIO);
                if (!last) ops |= LM_SOLVE_SVD | LM_SETUP_XNEW;
                else ops |= tail;
                EH_TRY(launch_lm(c, slot_new, ops));
            }
        }
    }
    // reweighted Levenberg-Marquardt
    EH_TRY(eval(true, true));
    {
        unsigned ops = LM_REDUCE_CUR | LM_INIT | LM_RESET_V;
        if (M > 0) ops |= LM_SOLVE_CHOL | LM_SETUP_XNEW; else ops |= LM_FINISH;
        EH_TRY(launch_lm(c, slot_new, ops));
    }
    for (int it = 0; it < M; it++) {
        EH_TRY(eval(true, true));
        unsigned ops = LM_REDUCE_NEW | LM_GAIN_RATIO | LM_SWAP_ON_ACCEPT;
        if (it < M - 1) ops |= LM_SOLVE_CHOL | LM_SETUP_XNEW; else ops |= LM_FINISH;
        EH_TRY(launch_lm(c, slot_new, ops));
    }
#undef EH_TRY
    return 0;
}

// Minimizer_V<double>: evaluate, then iter_max x (solve, evaluate at Vnew, gain test)
int minimizer_v_enqueue(edgehip_ctx *c, int slot_new, int slot_old, int fc_index, int iter_max, double match_thresh,
                        uint32_t match_num_thresh, double reweight_distance) {
    const DevicePlan &pl = c->plan;
    c->fc_index = fc_index;
    const int nblk256 = (pl.cap + 255) / 256;
    if (nblk256 > c->nblk_tvr * (kTvrBlock / 256)) { set_error("minimizer_v: block tables too small"); return EDGEHIP_ERR_STATE; }
    // per-256-KeyLine tables live in the buffers sized for TryVelRot (nblk_tvr >= nblk256 because kTvrBlock == 256)
    static_assert(kTvrBlock == 256, "k_try_vel shares the per-block tables of k_try_velrot");
    if (int e = rec_refresh_enqueue(c, slot_new)) return e;
    EH_CHECK(hipMemsetAsync(c->resid, 0, sizeof(double) * pl.nseq * pl.cap, c->stream));   // residuals[i] = 0
    TvrArgs a = make_tvr_args(c, slot_new, slot_old, match_thresh, reweight_distance, match_num_thresh, 1);
    a.fwd_key = nullptr;
    // the last evaluation posts FordwardMatch's keys when k_field_bin left the arbitration arrays cleared (whole-frame driver)
    c->fwd_keys_posted = c->fwd_cleared;
    const int32_t *kn_old = c->kn_slot + (size_t)slot_old * pl.nseq;
    auto eval = [&](bool use_new, bool last) {
        dim3 g(nblk256, 1, pl.nseq), b(256);
        TvrArgs al = a;
        if (last && c->fwd_keys_posted) al.fwd_key = c->fwd_key;
        if (use_new) hipLaunchKernelGGL((k_try_vel<true>), g, b, 0, c->stream, al);
        else hipLaunchKernelGGL((k_try_vel<false>), g, b, 0, c->stream, al);
    };
    auto step = [&](unsigned ops) {
        hipLaunchKernelGGL(k_lmv_step, dim3(pl.nseq), dim3(64), 0, c->stream, c->seq, c->partials, c->block_last, c->resid_carry,
                           kn_old, c->nblk_tvr, ops);
        if ((ops & (LMV_REDUCE_CUR | LMV_REDUCE_NEW)) && (ops & LMV_FINISH))   // between evaluations k_try_vel resolves the markers it meets
            hipLaunchKernelGGL(k_tv_resolve, dim3(nblk256, 1, pl.nseq), dim3(256), 0, c->stream, c->resid, c->resid_carry, kn_old,
                               pl.cap, c->nblk_tvr);
    };
    eval(false, iter_max <= 0);
    step(LMV_BEGIN | LMV_REDUCE_CUR | (iter_max > 0 ? LMV_SOLVE : LMV_FINISH));
    for (int it = 0; it < iter_max; it++) {
        eval(true, it == iter_max - 1);
        step(LMV_REDUCE_NEW | LMV_GAIN | (it < iter_max - 1 ? LMV_SOLVE : LMV_FINISH));
    }
    EH_LAUNCH_CHECK();
    return 0;
}

}  // namespace edgehip

using namespace edgehip;

extern "C" {

int edgehip_quantile(edgehip_ctx *c, int slot, double a, double b, double pct, int n) {
    EH_ENTER(c);
    if (!c || slot < 0 || slot >= c->plan.nslots) return EDGEHIP_ERR_ARG;
    if (int e = rot_materialize_enqueue(c, slot)) return e;   // s_rho of a slot the whole-frame driver rotated out of place
    return quantile_enqueue(c, slot, a, b, pct, n);
}
int edgehip_build_field(edgehip_ctx *c, int slot, int r, float m) {
    EH_ENTER(c);
    if (!c || slot < 0 || slot >= c->plan.nslots) return EDGEHIP_ERR_ARG;
    return build_field_enqueue(c, slot, r, m);
}

int edgehip_lm_solve(edgehip_ctx *c, const double *A, const double *b, int n, int svd_rule, double *h) {
    EH_ENTER(c);
    if (!A || !b || !h || n <= 0) {
        set_error("lm_solve: bad argument");
        return EDGEHIP_ERR_ARG;
    }
    double *d = nullptr;
    EH_CHECK(hipMalloc(&d, sizeof(double) * 48 * (size_t)n));
    hipError_t e = hipMemcpyAsync(d, A, sizeof(double) * 36 * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 36 * (size_t)n, b, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_lm_solve, dim3(n), dim3(64), 0, c->stream, d, d + 36 * (size_t)n, d + 42 * (size_t)n, svd_rule);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d + 42 * (size_t)n, sizeof(double) * 6 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    EH_CHECK(e);
    return 0;
}

int edgehip_try_velrot(edgehip_ctx *c, int slot_new, int slot_old, const double *X, int reweight, int procjf,
                       double match_thresh, const double *s_rho_min, uint32_t match_num_thresh, double k_huber,
                       int resid_in, int resid_out, double *out) {
    EH_ENTER(c);
    if (!c || !X || !s_rho_min || !out || slot_new < 0 || slot_old < 0 || slot_new >= c->plan.nslots ||
        slot_old >= c->plan.nslots || resid_in >= kResidBufs || resid_out < 0 || resid_out >= kResidBufs) {
        set_error("try_velrot: bad argument");
        return EDGEHIP_ERR_ARG;
    }
    const int B = c->plan.nseq;
    int e;
    c->fc_index = slot_new;
    if ((e = rot_materialize_enqueue(c, slot_old))) return e;
    if ((e = rec_refresh_enqueue(c, slot_new))) return e;
    // P0 is rebuilt every call (the old slot may have been edited through upload_keylines)
    if (resid_in < 0) {
        if ((e = tvr_prepare_enqueue(c, slot_old))) return e;  // also zeroes buffer 0
        if (resid_out == 0) { set_error("try_velrot: resid_out 0 is the zero buffer when resid_in < 0"); return EDGEHIP_ERR_ARG; }
    } else {
        // keep residual buffers; only refresh P0/kn_old: prepare writes resid0, so save/restore is avoided by
        // launching prepare with a scratch destination when buffer 0 is live
        const DevicePlan &pl = c->plan;
        hipLaunchKernelGGL(k_tvr_prepare, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot_old),
                           c->kn_slot + (size_t)slot_old * pl.nseq, c->P0, c->rs_tmp /*scratch*/, c->block_last /*scratch*/,
                           c->seq, pl.cap, c->nblk_tvr, pl.zfm);
        EH_LAUNCH_CHECK();
    }
    double *dX = c->pinned_out, *dS = c->pinned_out + (size_t)B * 6;
    memcpy(dX, X, sizeof(double) * 6 * B);
    memcpy(dS, s_rho_min, sizeof(double) * B);
    double *devX = c->rs_tmp + (size_t)B * c->plan.cap;  // second half of the scratch
    EH_CHECK(hipMemcpyAsync(devX, dX, sizeof(double) * 7 * B, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_tvr_setup_from_host, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, devX, devX + (size_t)B * 6,
                       B, resid_in, resid_out);
    EH_LAUNCH_CHECK();
    TvrArgs a = make_tvr_args(c, slot_new, slot_old, match_thresh, k_huber, match_num_thresh, 1);
    if ((e = launch_tvr(c, a, reweight != 0, procjf != 0))) return e;
    if ((e = launch_lm(c, slot_new, LM_REDUCE_CUR | (procjf ? 0 : LM_NOJAC)))) return e;
    EH_CHECK(hipMemcpyAsync(c->pinned_seq, c->seq, sizeof(SeqDev) * B, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < B; s++) {
        memcpy(out + (size_t)s * 43, c->pinned_seq[s].JtJ, sizeof(double) * 36);
        memcpy(out + (size_t)s * 43 + 36, c->pinned_seq[s].JtF, sizeof(double) * 6);
        out[(size_t)s * 43 + 42] = c->pinned_seq[s].F;
    }
    return 0;
}

__global__ void k_resolve_resid(const double *__restrict__ resid, const double *__restrict__ carry, double *__restrict__ out,
                                int cap, int nblk) {
    const int seq = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    double v = resid[(size_t)seq * cap + i];
    if (is_carry(v)) v = carry[(size_t)seq * nblk + i / kTvrBlock];
    out[(size_t)seq * cap + i] = v;
}

int edgehip_download_resid(edgehip_ctx *c, int which, double *resid) {
    EH_ENTER(c);
    if (!c || !resid || which < 0 || which >= kResidBufs) return EDGEHIP_ERR_ARG;
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_resolve_resid, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream,
                       c->resid + (size_t)which * pl.nseq * pl.cap, c->resid_carry + (size_t)which * pl.nseq * c->nblk_tvr,
                       c->rs_tmp, pl.cap, c->nblk_tvr);
    EH_LAUNCH_CHECK();
    EH_CHECK(hipMemcpyAsync(resid, c->rs_tmp, sizeof(double) * pl.nseq * pl.cap, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_minimizer_v(edgehip_ctx *c, int slot_new, int slot_old, double *V, const double *s_rho_min, float min_mod,
                        double match_thresh, int iter_max, uint32_t match_num_thresh, double reweight_distance, double *RVel,
                        double *F) {
    EH_ENTER(c);
    if (!c || !V || !s_rho_min || slot_new < 0 || slot_old < 0 || slot_new >= c->plan.nslots || slot_old >= c->plan.nslots ||
        iter_max < 0) {
        set_error("minimizer_v: bad argument");
        return EDGEHIP_ERR_ARG;
    }
    if (int e = rot_materialize_enqueue(c, slot_old)) return e;
    const int B = c->plan.nseq;
    EH_CHECK(hipMemcpyAsync(c->pinned_seq, c->seq, sizeof(SeqDev) * B, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < B; s++) {
        SeqDev &q = c->pinned_seq[s];
        for (int i = 0; i < 3; i++) q.mv_V[i] = V[s * 3 + i];
        q.mv_s_rho_min = s_rho_min[s];
        q.mv_min_mod = min_mod < 0.f ? 0.f : min_mod;
    }
    EH_CHECK(hipMemcpyAsync(c->seq, c->pinned_seq, sizeof(SeqDev) * B, hipMemcpyHostToDevice, c->stream));
    if (min_mod < 0.f) {   // old_buf.ef->getThresh(): each sequence's retuned threshold of the old slot
        std::vector<float> rt(B);
        EH_CHECK(hipMemcpyAsync(rt.data(), c->retuned_slot + (size_t)slot_old * B, sizeof(float) * B, hipMemcpyDeviceToHost, c->stream));
        EH_CHECK(hipStreamSynchronize(c->stream));
        for (int s = 0; s < B; s++) c->pinned_seq[s].mv_min_mod = rt[s];
        EH_CHECK(hipMemcpyAsync(c->seq, c->pinned_seq, sizeof(SeqDev) * B, hipMemcpyHostToDevice, c->stream));
    }
    if (int e = minimizer_v_enqueue(c, slot_new, slot_old, slot_new, iter_max, match_thresh, match_num_thresh, reweight_distance)) return e;
    EH_CHECK(hipMemcpyAsync(c->pinned_seq, c->seq, sizeof(SeqDev) * B, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < B; s++) {
        const SeqDev &q = c->pinned_seq[s];
        for (int i = 0; i < 3; i++) V[s * 3 + i] = q.mv_V[i];
        if (RVel) for (int i = 0; i < 9; i++) RVel[s * 9 + i] = q.mv_RVel[i];
        if (F) F[s] = q.mv_F;
    }
    return 0;
}

int edgehip_minimizer_rv_kf(edgehip_ctx *c, int slot_kf, int slot_cur, const edgehip_kf_request *req, double match_mod, double match_ang,
                            double rho_tol, int iter_max, double reweight_distance, uint32_t match_num_thresh, edgehip_kf_result *res) {
    EH_ENTER(c);
    if (!c || !req || !res || slot_kf < 0 || slot_cur < 0 || slot_kf >= c->plan.nslots || slot_cur >= c->plan.nslots || slot_kf == slot_cur)
        return EDGEHIP_ERR_ARG;
    const size_t B = c->plan.nseq;
    if (!c->kf_req_dev) {
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(edgehip_kf_request) * B) != hipSuccess) { (void)hipGetLastError(); set_error("kf request alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->kf_req_dev = (edgehip_kf_request *)q;
        if (hipMalloc(&q, sizeof(edgehip_kf_result) * B) != hipSuccess) { (void)hipGetLastError(); set_error("kf result alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->kf_res_dev = (edgehip_kf_result *)q;
    }
    if (int e = rot_materialize_enqueue(c, slot_cur)) return e;
    if (int e = sync_all(c)) return e;   // the request comes from pageable memory and the field is about to be rebuilt
    EH_CHECK(hipMemcpy(c->kf_req_dev, req, sizeof(edgehip_kf_request) * B, hipMemcpyHostToDevice));
    // the key frame's global_tracker: the field of ITS KeyLines, as built when the frame was current (rebvo_second_t.cpp:177;
    // keyframe.cpp:31 copies it).  The context has one field, so it is rebuilt here; the next frame rebuilds its own.
    if (int e = build_field_enqueue(c, slot_kf, c->p.search_range, -1.f)) return e;
    if (int e = minimizer_kf_enqueue(c, slot_kf, slot_cur, match_mod, match_ang, rho_tol, iter_max, reweight_distance, match_num_thresh)) return e;
    EH_CHECK(hipStreamSynchronize(c->stream));
    EH_CHECK(hipMemcpy(res, c->kf_res_dev, sizeof(edgehip_kf_result) * B, hipMemcpyDeviceToHost));
    return 0;
}

int edgehip_minimizer_rv(edgehip_ctx *c, int slot_new, int slot_old) {
    EH_ENTER(c);
    if (!c || slot_new < 0 || slot_old < 0 || slot_new >= c->plan.nslots || slot_old >= c->plan.nslots) return EDGEHIP_ERR_ARG;
    if (int e = rot_materialize_enqueue(c, slot_old)) return e;
    return minimizer_enqueue(c, slot_new, slot_old, slot_new);
}

}  // extern "C"
