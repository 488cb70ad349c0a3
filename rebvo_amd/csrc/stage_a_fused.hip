// stage_a_fused.hip — the whole of stage A up to the KeyLine records in ONE kernel: one workgroup walks one frame top to
// bottom and nothing but the RGB frame is read from HBM, nothing but img_mask_kl and the KeyLines is written.
//
// Replaces, for batches that fill the GPU (one workgroup per sequence): k_level x3, k_detect, k_strip_scan, k_emit, i.e.
//   Image<float>::ConvertRGB2BW            include/VideoLib/image.h:197-203
//   iimage::load / iimage::average         src/mtracklib/iimage.cpp:53-71, 86-128     (x3 per filter)
//   iigauss::smooth                        src/mtracklib/iigauss.cpp:91-101
//   sspace::build / build_dog / calc_gradient   src/mtracklib/sspace.cpp:52-85
//   edge_finder::build_mask + the P-controller of detect()   src/mtracklib/edge_finder.cpp:67-214, 330-365
// The multi-kernel path (stage_a.hip) moves 31 N bytes through HBM for the three integral-image levels, another ~11 N for
// the detector's taps and writes the mask twice; this one moves 3 N (N for 8-bit mono input) + 4 N + 24 bytes per KeyLine.
//
// How the box chain stays on chip.  A box average needs four taps of the integral image: bottom row y+r and top row
// y-r-1, columns x+r and x-r-1.  The integral image itself is never materialised:
//   * the serial left-to-right row prefix of a row (iimage.cpp:56-61, order-defining in float32) is done in LDS by the
//     scan wave, one lane per row, as in k_level;
//   * the serial top-to-bottom column prefix (iimage.cpp:63-67) is a running sum; the thread that owns column x keeps the
//     running sums of ITS TWO TAP COLUMNS x+r and x-r-1 (every column sum is therefore computed by two threads, with the
//     same operands in the same order, hence the same bits) — no exchange of integral values between threads;
//   * the top taps of output row y are the bottom taps of output row y-d: a register ring per tap column with static
//     positions (the tick loop is unrolled over two ticks = 2 RB rows and the ring lengths divide 2 RB, so nothing is moved).
// A thread owns two ADJACENT columns and computes them with packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32: two
// IEEE operations per lane and instruction, each rounded exactly like the scalar one), tap pairs come out of LDS with one
// ds_read2_b32.  So a level costs, per pixel pair: two LDS reads, two packed adds, the packed four-tap combine.  The levels are
// chained through LDS row buffers that hold RB rows each: produced in tick t, row-scanned in tick t+1, consumed in
// tick t+2 (two buffer sets, alternating).  Level l's rows trail its input by r rows, so img0 = G(sigma0) trails the
// input by r1+r2a+r3a rows and img1 by r1+r2b+r3b; the kernel is instantiated for the box widths of the shipped
// configurations ({3,3,5} / {3,5,5}: Sigma0 1.7818, KSigma 1.2599), everything else takes the multi-kernel path.
//
// A workgroup = ceil(w / 128) column waves + the scan wave + the fit wave.  A tick (RB image rows) of the column waves:
//   phase 1   the img_mask_kl rows of the rows tested two ticks ago (the fit wave left their KeyLine ids in LDS); grey
//             values of the prefetched RGB (or 8-bit mono / 16-bit grey) rows; the five box averages of the tick from the
//             scanned rows of buffer set t&1; DoG rows into the LDS ring
//   -- barrier (LDS only) --
//   phase 2   store the produced rows into buffer set t&1 (all its readers are past the barrier); gradient gate and sign
//             balance of build_mask on the rows whose 5x5 DoG window is complete; the survivors go, as (row, column) codes,
//             into the (row, wave) segment lists of the tick
//   -- barrier --
// while the scan wave row-scans buffer set (t+1)&1, half before and half after the middle barrier, and the fit wave works
// through the previous tick's survivors: fp64 plane fit (edge_finder.cpp:139-159) 64 candidates at a time in raster order,
// sub-pixel and DoG-gradient tests, KeyLine ids as a running count — the id IS the raster rank (edge_finder.cpp:166-200) and
// one wave sees every candidate of the frame in order — and the fit's results (p_inx, {xs, ys, m_m}, p_id) to HBM;
// k_join_histo<true, true> derives the other KeyLine fields.  No staging pass, no second detector kernel.
//
// Compile with -ffp-contract=off (the reference is built without FMA contraction).

#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "ctx.h"
#include "stage_a_dev.h"

namespace edgehip {

typedef float v2f __attribute__((ext_vector_type(2)));   // the two adjacent columns of a thread: packed fp32 math

// Timing experiments (-DEDGEHIP_FUSED_TSTAMP=k builds only; tools/experiments/exp_fused_where.sh, exp_fused_roles.sh): kn_out returns a time instead of
// kn.  k = 1..6: an interval of the workgroup's life in 10 ns units (1 set-up, 2 ticks 0-9, 3 ticks 10-109, 4 ticks 110-end, 5 after the loop, 6 all).
// k = RX: a role's cycles per tick, mean over ticks 10..109 — R: 1 / 4 / 5 = column waves 0 / 2 / 5, 2 = the scan wave, 3 = the fit wave; X: 1 work before
// the middle barrier, 2 wait there, 3 work before the end barrier, 4 wait there.  EH_TS_BAR stands where the roles' loops call lds_barrier().
#ifdef EDGEHIP_FUSED_TSTAMP
#define EH_TS_DECL long long tsw[4] = {0, 0, 0, 0}, ts_mark = 0;
#define EH_TS_BEGIN(t) if ((t) >= 10 && (t) < 110) ts_mark = clock64();
#define EH_TS_BAR(i, t)                                                   \
    {                                                                     \
        const long long ts_x = clock64();                                 \
        lds_barrier();                                                    \
        const long long ts_y = clock64();                                 \
        if ((t) >= 10 && (t) < 110) { tsw[i] += ts_x - ts_mark; tsw[(i) + 1] += ts_y - ts_x; } \
        ts_mark = ts_y;                                                   \
    }
#define EH_TS_OUT(role) if (lane == 0 && EDGEHIP_FUSED_TSTAMP / 10 == (role)) a.kn_out[seq] = (int)(tsw[EDGEHIP_FUSED_TSTAMP % 10 - 1] / 100);
#else
#define EH_TS_DECL
#define EH_TS_BEGIN(t)
#define EH_TS_BAR(i, t) lds_barrier();
#define EH_TS_OUT(role)
#endif

// Running column sums of a level's tap columns x+r (a) and x-r-1 (b), for the last L integral rows, as a ring with static
// positions: row q of the unrolled tick pair lives at position q % L.
template <int L>
struct TapRing {
    v2f a[L], b[L];
};

__device__ __forceinline__ v2f ld2(const float *p) { v2f v; v.x = p[0]; v.y = p[1]; return v; }   // -> ds_read2_b32

// One input row arrives for a level (ring position P): vr / vl are the row-prefixed input values at the tap columns.
// Returns the box average of output row (input row - r) — iimage::average (iimage.cpp:86-128) on an integral image that is
// never stored.  Steady form: image row whose box is clipped neither above nor below; m = div(x,y) of the two columns.
template <int D, int L>
__device__ __forceinline__ v2f ring_row_steady(TapRing<L> &H, int P, v2f vr, v2f vl, v2f m) {
    const int P1 = (P + L - 1) % L, PD = (P + L - D) % L;
    H.a[P] = H.a[P1] + vr;                                       // img(x,y) += img(x,y-1), iimage.cpp:63-67
    H.b[P] = H.b[P1] + vl;
    return (((H.a[P] - H.b[P]) - H.a[PD]) + H.b[PD]) * m;        // ((A-B)-C)+D, iimage.cpp:119-126
}
// General form; every condition is wave-uniform.  Rows above the image leave the sums at 0 (their taps are the zeros x - 0
// needs); the r virtual rows below it leave the bottom taps on row h-1 and take the bottom band's operand order
// ((A-C)-B)+D (iimage.cpp:105-113); div(x,y) = 1/count comes from the table.
template <int D, int L>
__device__ __forceinline__ v2f ring_row_general(TapRing<L> &H, int P, v2f vr, v2f vl, int cx0, int cx1, const float *s_lut,
                                                int yin, int h) {
    constexpr int R = D / 2;
    const int P1 = (P + L - 1) % L, PD = (P + L - D) % L;
    v2f out = {0.f, 0.f};
    if (yin >= 0 && yin < h) {
        H.a[P] = H.a[P1] + vr;
        H.b[P] = H.b[P1] + vl;
        const int cy = yin < D ? yin + 1 : D;
        v2f m; m.x = s_lut[cx0 * cy]; m.y = s_lut[cx1 * cy];
        out = (((H.a[P] - H.b[P]) - H.a[PD]) + H.b[PD]) * m;
    } else {
        H.a[P] = H.a[P1];
        H.b[P] = H.b[P1];
        if (yin >= h && yin < h + R) {
            const int cy = h - yin + D - 1;
            v2f m; m.x = s_lut[cx0 * cy]; m.y = s_lut[cx1 * cy];
            out = (((H.a[P] - H.a[PD]) - H.b[P]) + H.b[PD]) * m;
        }
    }
    return out;
}

// build_mask's plane fit on the 5x5 DoG window (edge_finder.cpp:139-159) from the LDS ring: ro[k] = element offset of
// window row k, x = column.  Same operation order as k_detect (TooN dot product, k = 0..24).
#ifndef EDGEHIP_NT_EMIT
#define EDGEHIP_NT_EMIT 1   // the fit wave's KeyLine stores as streaming stores: A.fused 2202 -> 2186 us, k_join_histo unchanged
#endif
#ifndef EDGEHIP_FIT_SHARED_PRODUCT
#define EDGEHIP_FIT_SHARED_PRODUCT 1
#endif
struct FitOut { bool cand; float mx, my, xs, ys; };
// The 11 coefficients of the pseudo inverse that are not zero / not repeated (row 0 depends on the window column only, row 1
// on the window row, row 2 is constant: checked at create), loaded ONCE by the wave that fits — a load inside the fit would
// sit in the chunk loop behind an s_waitcnt vmcnt(0), i.e. behind every KeyLine store still in flight.
struct FitCoef { double pc0[5], pc1[5], pc2; };
__device__ __forceinline__ FitCoef load_fit_coef(const double *__restrict__ pinv) {
    FitCoef c;
#pragma unroll
    for (int j = 0; j < 5; j++) { c.pc0[j] = pinv[j]; c.pc1[j] = pinv[25 + 5 * j]; }
    c.pc2 = pinv[50];
    return c;
}
// The window is loaded (fit_load) and evaluated (fit_eval) separately so that the fit wave can request the next chunk's 25
// values before it evaluates the current one: it is a single wave, nobody else hides its LDS round trips.
__device__ __forceinline__ void fit_load(const float *s_dog, const int ro[5], float (&v)[25]) {   // ro[k]: element offset of window row k at column x - 2
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const float *rp = s_dog + ro[i];        // one address per window row, the five columns as immediate offsets
#pragma unroll
        for (int j = 0; j < 5; j++) v[i * 5 + j] = rp[j];
    }
}
// between(i) runs after window row i's terms: the fit wave puts the steps of the NEXT chunks' list search there, so that the LDS round trip of a
// step passes under a row's arithmetic instead of in front of the whole fit.
template <class Between>
__device__ __forceinline__ FitOut fit_eval(const float (&v)[25], const FitCoef &fc, float thr_d, Between between) {
    double t0 = 0, t1 = 0, t2 = 0;
    const double (&pc0)[5] = fc.pc0;
    const double (&pc1)[5] = fc.pc1;
    const double pc2 = fc.pc2;
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        // The middle column of row 0 and the middle row of row 1 are exactly zero (fused_supported checks it): a product with
        // them is +-0, and adding that changes nothing — a partial sum is never -0 (it starts at +0, and x + (-x) = +0) — so those
        // ten terms are skipped; the sums keep their bits.  Same operation order as k_detect (TooN dot product, k = 0..24).
        for (int j = 0; j < 5; j++) {
            const double yv = (double)v[i * 5 + j];
#if EDGEHIP_FIT_SHARED_PRODUCT
            // The coefficients of the symmetric 5x5 window are {-2u, -u, 0, u, 2u} along a row of PInv's row 0 (and down a column of
            // its row 1) and 2u everywhere in row 2, u = fl(1/50) (fused_supported checks exactly that): a scaling by two is exact, so
            // the three rounded products of a window value are -+q, -+2q and 2q with ONE rounded product q = u * yv, and a sum plus
            // the exact 2q rounds once — as an fma of (q, 2, sum) — to what the reference's sum plus its rounded product rounds to.
            const double q = pc0[3] * yv;
            if (j == 0) t0 = __builtin_fma(q, -2.0, t0);
            if (j == 1) t0 = t0 - q;
            if (j == 3) t0 = t0 + q;
            if (j == 4) t0 = __builtin_fma(q, 2.0, t0);
            if (i == 0) t1 = __builtin_fma(q, -2.0, t1);
            if (i == 1) t1 = t1 - q;
            if (i == 3) t1 = t1 + q;
            if (i == 4) t1 = __builtin_fma(q, 2.0, t1);
            t2 = __builtin_fma(q, 2.0, t2);
            (void)pc1; (void)pc2;
#else
            if (j != 2) t0 += pc0[j] * yv;
            if (i != 2) t1 += pc1[i] * yv;
            t2 += pc2 * yv;
#endif
        }
        // (empty statements that "use" the three sums: the row's arithmetic has to stand in front of the first, the next row's behind the second —
        // left alone, the compiler lines up all five steps of the search and then the whole fit)
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
        between(i);
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
    }
    FitOut o;
    o.cand = false;
    const double den = t0 * t0 + t1 * t1;
    {   // xs, ys = -t0 t2 / den, -t1 t2 / den (edge_finder.cpp:150-153): the two divisions as their own sequence with the divisor's part shared
        // (div2_mid, ctx.h: the same fma chain as the compiler's a / b for operands in the middle of the exponent range — sums of at most 25 float
        // products and their squares —, 0 / 0 -> NaN through v_div_fixup as before): 13 fp64 instructions fewer per fit on the wave that sets the tick
        double qx, qy;
        div2_mid(-t0 * t2, -t1 * t2, den, qx, qy);
        o.xs = (float)qx;
        o.ys = (float)qy;
    }
    o.mx = (float)t0;
    o.my = (float)t1;
    if (!(fabsf(o.xs) > 0.5f || fabsf(o.ys) > 0.5f)) {
        const float n2m = o.mx * o.mx + o.my * o.my;
        if (!(n2m < thr_d)) o.cand = true;
    }
    return o;
}

// W = image width known at compile time (LDS offsets become instruction immediates), 0 = taken from the arguments.
// DBG = also write the img0 / img1 / DoG / dx / dy planes (debug_planes contexts: tests).
// SRC = what the first load reads (FusedSrc, stage_a_dev.h): the RGB24 frame, a plane of 16-bit grey values b+g+r (a.grey16:
// the frame after k_undistort_grey, image_undistort fused with ConvertRGB2BW), or an 8-bit mono frame (a.grey8: 3 v).
template <int W, bool DBG, int SRC, int RB, int D1, int D2A, int D2B, int D3A, int D3B>
__global__ __launch_bounds__(512) void k_stage_a_fused(FusedArgs a) {
    constexpr bool GREY16 = SRC == SRC_GREY16, GREY8 = SRC == SRC_GREY8, UND = SRC == SRC_UNDIST;
    constexpr int R1 = D1 / 2, R2A = D2A / 2, R2B = D2B / 2, R3A = D3A / 2, R3B = D3B / 2;
    constexpr int LB = R1 + R2B + R3B;                  // rows img1 trails the input by
    static_assert(R1 + R2A + R3A + 1 == LB, "img0 must lead img1 by exactly one row (it is held for one step)");
    constexpr int RING = 2 * RB + 4;                    // DoG rows in LDS: RB being written + RB + 4 being read
    constexpr int PAD = kFusedPad;
    // ring lengths: the smallest divisor of 2 RB that holds D + 1 rows
    constexpr int L1 = D1 + 1 <= RB ? RB : 2 * RB, L2A = D2A + 1 <= RB ? RB : 2 * RB, L2B = D2B + 1 <= RB ? RB : 2 * RB;
    constexpr int L3A = D3A + 1 <= RB ? RB : 2 * RB, L3B = D3B + 1 <= RB ? RB : 2 * RB;
    static_assert(D1 + 1 <= L1 && D2A + 1 <= L2A && D2B + 1 <= L2B && D3A + 1 <= L3A && D3B + 1 <= L3B, "box wider than two ticks");
    static_assert(R1 <= 3 && R2A <= 3 && R2B <= 3 && R3A <= 3 && R3B <= 3, "taps must stay inside the row pads");
#ifdef EDGEHIP_FUSED_ABL    // tools/experiments/exp_fused_ablate.sh: phase ablation for timing experiments (wrong results by design)
    constexpr int ABL = EDGEHIP_FUSED_ABL;
#else
    constexpr int ABL = 0;
#endif

    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef EDGEHIP_FUSED_TSTAMP
    long long ts_start = wall_clock64(), ts_pro = 0, ts_a = 0, ts_b = 0, ts_loop = 0;
#endif
    const int w = W ? W : a.w, h = a.h;
    const int WP = fused_row_stride(w);
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction; telling the compiler so moves everything derived from it (list / result pointers, clipping
    // flags, column offsets of the wave) from vector to scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = W ? (W + 127) / 128 : (int)(blockDim.x >> 6) - 2;   // column waves (128 columns each); + the scan wave + the fit wave
    const int seq = blockIdx.x;
    const size_t so = (size_t)seq * a.n;

    float *s_set = smem;                                            // [2][4][RB][WP]
    float *s_dog = s_set + (size_t)2 * 4 * RB * WP;                 // [RING][WP]
    float *s_lut = s_dog + (size_t)RING * WP + 32;                  // [kDivLutMax]   (+32: read overrun of the last scanned row)
    // [RB][NW + 2][2] {img0, DoG sign bits} at the first / last column (pair) of every wave, with an all-zero entry either side
    // of the image (waves -1 and NW: what the gate reads left of column 0 and right of column w-1), so that the reader needs no
    // "is there a wave next to me" test
    uint2 *s_edge2 = reinterpret_cast<uint2 *>(s_lut + kDivLutMax);
    float *s_red = reinterpret_cast<float *>(s_edge2 + (size_t)RB * (NW + 2) * 2);   // [4] n_m extremes, the frame's candidate count (end of frame)
    // candidates of a tick's tested rows, published by the column waves for the fit wave, double-buffered by tick parity: per
    // (row, wave) segment the number of survivors of the two dense gates and their codes (row << 10 | x) in column order
    int *s_ccnt = reinterpret_cast<int *>(s_red + 4);                                   // [2][RB*NW]
    uint16_t *s_clist = reinterpret_cast<uint16_t *>(s_ccnt + 2 * RB * NW);             // [2][RB*NW][128]
    uint16_t *s_res = s_clist + (size_t)2 * NW * RB * 128;          // [2 (tick parity)][RB][NW*128] id + 1 of the KeyLine at a tested pixel, 0 = none
    uint16_t *s_dummy = s_res + (size_t)2 * NW * RB * 128;          // [4] where the lanes without a candidate "publish" (a store needs no branch then)
    // ---- set-up -------------------------------------------------------------------------------------------------------
    for (int i = tid; i < 2 * 4 * RB * PAD; i += blockDim.x) s_set[(size_t)(i / PAD) * WP + (i % PAD)] = 0.f;   // left pads: taps left of column 0
    for (int i = tid; i < kDivLutMax - 2; i += blockDim.x) s_lut[i] = a.lut[i];   // (indices reach 15 * 15; the last two entries carry the threshold, below)
    for (int i = tid; i < 2 * NW * RB * 128; i += blockDim.x) s_res[i] = 0;
    for (int i = tid; i < 2 * RB * NW; i += blockDim.x) s_ccnt[i] = 0;
    for (int i = tid; i < RB * (NW + 2) * 2; i += blockDim.x) s_edge2[i] = make_uint2(0u, 0u);
    for (int i = tid; i < 256; i += blockDim.x) a.histo[(size_t)seq * 256 + i] = 0;   // reEstimateThresh's histogram (k_join_histo fills it)
    SeqA *sq = a.seq + seq;
    // wave-uniform floats are computed by the vector ALU and would sit in (scarce) vector registers: readfirstlane moves them
    // to scalar registers
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    float thr_g, thr_d;
    {
        const double tresh = update_thresh(sq->tresh, sq->l_kl_num, a.kl_ref, a.gain, a.tmax, a.tmin);
        // kept in two unused entries of the reciprocal table for the end of the frame: the controller's constants then need
        // no scalar registers through the frame loop
        if (tid == 0) { s_lut[kDivLutMax - 2] = __int_as_float(__double2loint(tresh)); s_lut[kDivLutMax - 1] = __int_as_float(__double2hiint(tresh)); }
        const float grad_thresh = (float)tresh;                     // build_mask takes float grad_thesh
        const float gt1 = grad_thresh * 765;                        // grad_thesh*max_img_value
        const float gt2 = gt1 * a.dog_thresh_f;
        thr_g = uni(gt1 * gt1);
        thr_d = uni(gt2 * gt2);
    }
    __syncthreads();
#ifdef EDGEHIP_FUSED_TSTAMP
    ts_pro = wall_clock64();
#endif

    // number of ticks: the tests of tick t cover rows (t-6)*RB - LB - 2 + [0, RB), their KeyLines are emitted in tick t+1
    const int t_last = 6 + (h - 1 + LB + 2) / RB;                   // tick that tests row h-1
    // + 1: the fit wave works on its rows, + 2: the column waves write their img_mask_kl rows; the tick loop is unrolled by two
    const int nticks = (t_last + 3 + 1) & ~1;

    // Who shares a SIMD with whom (waves go to the four SIMDs of a CU round-robin).  The scan wave (a dependent add chain at
    // raised priority) and the fit wave (one wave's worth of serial fp64) are the two long poles of a tick next to the column
    // waves; put on the same SIMD (scan = wave 3, fit = wave 7) they pay for each other's issue slots, so the scan wave is
    // wave 2: SIMD 2 = scan + a column wave, SIMD 3 = a column wave + fit, SIMDs 0 / 1 two column waves each (-2 %, same box).
#ifndef EDGEHIP_FUSED_SCANW
#define EDGEHIP_FUSED_SCANW 2
#endif
#ifndef EDGEHIP_FUSED_FITPRIO
#define EDGEHIP_FUSED_FITPRIO 2
#endif
#ifndef EDGEHIP_FUSED_SCANPRIO
#define EDGEHIP_FUSED_SCANPRIO 3
#endif
    const int scan_wave = NW >= EDGEHIP_FUSED_SCANW ? EDGEHIP_FUSED_SCANW : NW;
    if (wave == scan_wave) {
        // ---- the scan wave: the serial left-to-right prefix of iimage::load (iimage.cpp:56-61) for the 4 RB rows of the buffer
        // set (plane-major, RB rows per plane), in place.  The adds of a row are one dependent chain (~7 cycles each,
        // tools/experiments/ubench_scan.hip) and an LDS instruction costs the wave ~16 cycles of issue whatever the number of
        // active lanes, so all 64 lanes carry data: the four lanes of a quad share a row and every lane owns one 16-float chunk
        // of the row's current 64-float step (four ds_read_b128 per step and wave).  The chunks are chained in time, not in
        // space: the quad's lanes take turns (exec mask), lane p runs its 16 plain adds from the carry lane p-1 left, then its
        // total is broadcast to the quad (one DPP move) as the carry of lane p+1.  Per element: one add on the chain and nothing
        // else (the earlier form — every lane running the whole chain and selecting its four sums — spent 28 vector
        // instructions per 16 floats, 13-15 cycles per element; this one is bound by the chain).  After the last chunk the
        // row's total goes into the right pad (taps right of w-1).
        __builtin_amdgcn_s_setprio(EDGEHIP_FUSED_SCANPRIO);
        static_assert(4 * RB <= 16, "one quad per row");
        const int n16 = w >> 4;                 // full 16-float chunks
        const int rem4 = (w & 15) >> 2;         // float4s of the last, partial chunk (w % 4 == 0)
        const int nch = n16 + (rem4 ? 1 : 0);   // chunks per row
        const int nss = (nch + 3) >> 2;         // 64-float steps
        const int half_ss = nss >> 1;
        const int srow = lane >> 2, sq = lane & 3;
        const bool on = srow < 4 * RB && !(ABL & 1);
#define EH_BC(val, k) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(val), (k) * 0x55, 0xf, 0xf, true))
        EH_TS_DECL
        for (int t = 0; t < nticks; t++) {
            EH_TS_BEGIN(t)
            float *row = s_set + ((size_t)((t + 1) & 1) * 4 * RB + (srow < 4 * RB ? srow : 0)) * WP + PAD;
            float acc = 0.f;
            float4 cur[4], nxt[4];
            // chunk c of the row lives at row + 16 c; reads past the row's end fetch pad / the next row and are never used
            auto load = [&](float4 (&v)[4], int ss) __attribute__((always_inline)) {
                const float *p = row + (4 * ss + sq) * 16;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const float4 *>(p + 4 * i);
            };
            // one 64-float step on `v` (read one step earlier); the next step's reads are issued first
            auto step = [&](float4 (&v)[4], float4 (&nx)[4], int ss) __attribute__((always_inline)) {
                load(nx, ss + 1);
// the chain runs through the elements' own registers (x' = carry + x, y' = x' + y, ...): one instruction per element
// (plain C: as one asm statement per add — the round-2 form, to pin the in-register chain — every add drew an s_nop 0 behind it, the hazard
// recogniser's assumption about an opaque VALU write; the compiler emits the same sixteen back-to-back v_add_f32 by itself: 15.8 -> 11.5 cycles
// per element in tools/experiments/ubench_scan2.hip, A.fused 2750 -> 2580 us per 1024 frames)
#define EH_ADDC(DST, PREV) DST = (PREV) + DST;
#define EH_ADD4(I, CARRY) EH_ADDC(v[I].x, CARRY) EH_ADDC(v[I].y, v[I].x) EH_ADDC(v[I].z, v[I].y) EH_ADDC(v[I].w, v[I].z)
#define EH_PHASE(P)                                                                                  \
    {                                                                                                \
        const int c = 4 * ss + (P);               /* wave-uniform */                                 \
        float tot = acc;                                                                             \
        if (c < n16) {                                                                               \
            if (sq == (P)) { EH_ADD4(0, acc) EH_ADD4(1, v[0].w) EH_ADD4(2, v[1].w) EH_ADD4(3, v[2].w) } \
            tot = v[3].w;                         /* lane P's: the other lanes' is never looked at */ \
        } else if (c == n16 && rem4) {                                                               \
            if (sq == (P)) {                                                                         \
                if (0 < rem4) { EH_ADD4(0, acc) }                                                    \
                if (1 < rem4) { EH_ADD4(1, v[0].w) }                                                 \
                if (2 < rem4) { EH_ADD4(2, v[1].w) }                                                 \
            }                                                                                        \
            tot = rem4 == 1 ? v[0].w : (rem4 == 2 ? v[1].w : v[2].w);                                \
        }                                                                                            \
        acc = EH_BC(tot, P);                      /* the running total so far: lane P's, for the whole quad */ \
    }
                EH_PHASE(0) EH_PHASE(1) EH_PHASE(2) EH_PHASE(3)
#undef EH_PHASE
#undef EH_ADD4
#undef EH_ADDC
                // store the chunk (all of it, or the float4s the row still has)
                const int c = 4 * ss + sq;
                float *q = row + c * 16;
                if (c < n16) {
#pragma unroll
                    for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(q + 4 * i) = v[i];
                } else if (c == n16) {
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        if (i < rem4) *reinterpret_cast<float4 *>(q + 4 * i) = v[i];
                }
            };
            auto steps64 = [&](int s0, int s1) __attribute__((always_inline)) {   // ends with the next step's data in `cur`
                if (!on) return;
                int ss = s0;
                for (; ss + 1 < s1; ss += 2) {
                    step(cur, nxt, ss);
                    step(nxt, cur, ss + 1);
                }
                if (ss < s1) {
                    step(cur, nxt, ss);
#pragma unroll
                    for (int i = 0; i < 4; i++) cur[i] = nxt[i];
                }
            };
            if (on) load(cur, 0);
            steps64(0, half_ss);
            EH_TS_BAR(0, t)
            steps64(half_ss, nss);
            if (on && sq == 3) *reinterpret_cast<float4 *>(row + w) = make_float4(acc, acc, acc, acc);
            EH_TS_BAR(2, t)
        }
        EH_TS_OUT(2)
#undef EH_BC
    } else if (wave == NW + 1) {
        // ---- the fit wave: the sparse part of build_mask (edge_finder.cpp:139-209), one tick behind the column waves ----------
        // Of the ~360 k pixels of a frame ~25 k pass the gradient gate and the sign balance and ~13 k become KeyLines.  Inside the
        // column waves that work ran at a third of the lanes (each wave compacting, fitting and emitting its own 128 columns) and
        // dragged the id bookkeeping of six waves with it; here ONE wave sees all candidates of the tick's RB rows in raster
        // order: list from the published bits, fp64 plane fits 64 at a time, ids as a running count (the KeyLine id IS the raster
        // rank, edge_finder.cpp:166-200 — no cross-wave scan), KeyLine records and the tested rows of img_mask_kl.  Nothing of
        // this is on the column waves' critical path any more; the wave shares its SIMD with the scan wave, whose dependent add
        // chain leaves the issue slots free.
        int total = 0;                              // KeyLine candidates of the frame so far
        float nm_mx = 0.f, nm_mn = __int_as_float(0x7f800000);
        // the three arrays the wave stores to, and the fit's coefficients: fetched once, before the tick loop (inside it every
        // load would wait for the KeyLine stores in flight: vmcnt counts loads and stores alike)
        const KlSoA &klr = a.kl[seq];
        // (as global-memory pointers: loaded from the KlSoA record they are generic to the compiler, and a FLAT store also counts on the LDS
        // counter — every wait for an LDS answer behind it waits for the store's address check as well)
        typedef int32_t __attribute__((address_space(1))) gi32;
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef f4v __attribute__((address_space(1))) gf4;
        gi32 *const k_pinx = (gi32 *)klr.p_inx;
        gi32 *const k_pid = (gi32 *)klr.p_id;
        gf4 *const k_grec = (gf4 *)klr.grec;
        const FitCoef fc = load_fit_coef(a.pinv);
        const int WRES = NW * 128;                  // row stride of s_res
        __builtin_amdgcn_s_setprio(EDGEHIP_FUSED_FITPRIO);   // behind the scan wave's chain, ahead of the column waves
        auto below = [&](unsigned long long m) __attribute__((always_inline)) {   // set bits of a wave mask below this lane
            return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        };
        EH_TS_DECL
        for (int t = 0; t < nticks; t++) {
            EH_TS_BEGIN(t)
            // the rows tested in tick t-1: ytest0 + [0, RB); their 5x5 DoG windows sit in ring slots rq - 4 .. rq + RB - 1 with
            // rq = slot of tick t-1's first DoG row, untouched by the rows tick t writes (slots rq + RB ..)
            const int ytest0 = (t - 7) * RB - LB - 2;
            const bool live = t >= 1 && ytest0 + RB - 1 >= 0 && ytest0 < h && !(ABL & 8);
            int rq = ((t - 1) * RB) % RING;
            rq += rq < 0 ? RING : 0;
            // The candidates wait in per-(row, wave) segments of 128 slots (written by the column waves, raster order = segment
            // order).  Lane s holds segment s's count; an inclusive scan over the lanes gives the list position where each
            // segment ends, and list entry li lives at slot li + (128 s - start of s) of its segment s.
            int ncand = 0, seg_end = 0x7fffffff, seg_shift = 0;
            if (live) {
                const int nseg = RB * NW;
                const int cnt = lane < nseg ? s_ccnt[((t - 1) & 1) * nseg + lane] : 0;
                int inc = cnt;
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);    // row_shr:1, zeros shifted in
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);    // row_shr:2
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);    // row_shr:4
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);    // row_shr:8  -> inclusive scan inside each row of 16
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
                ncand = __builtin_amdgcn_readlane(inc, nseg - 1);
                seg_end = lane < nseg ? inc : 0x7fffffff;        // lanes beyond the last segment never match
                seg_shift = lane * 128 - (inc - cnt);
            }
            if (ABL & 64) ncand = 0;                // (timing experiments: 64 no fits / emission, 128 no emission, 256 no plane fit)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int nchunks = (ncand + 63) >> 6;
            // Software pipeline over the chunks of 64 candidates: while chunk c is evaluated, the 25 window values of chunk
            // c+1 and the list entries of chunk c+2 are on their way (LDS returns a wave's requests in order).
            float wv_cur[25], wv_nxt[25];
            int code_cur = 0, code_nxt = 0, code_nn = 0;
            const uint16_t *clist = s_clist + (size_t)((t - 1) & 1) * RB * NW * 128;
            // The segment of list entry li = the first lane whose segment ends beyond li: a binary search over the (non-decreasing) ends
            // held in lanes 0..31 (RB * NW <= 32), five lane reads, then the entry itself — six LDS round trips in a row.  The tick's first
            // two chunks are searched side by side (two probes per round trip); every later chunk's search is spread over the plane fit
            // two chunks ahead of it (one_chunk).
            auto code_of2 = [&](int ca, int cb, int &code_a, int &code_b) __attribute__((always_inline)) {
                const int la = ca * 64 + lane, lb = cb * 64 + lane;
                int loa = 0, lob = 0;
#pragma unroll
                for (int step = 16; step > 0; step >>= 1) {
                    const int ea = __shfl(seg_end, loa + step - 1, 64), eb = __shfl(seg_end, lob + step - 1, 64);
                    loa += ea <= la ? step : 0;
                    lob += eb <= lb ? step : 0;
                }
                const int sha = __shfl(seg_shift, loa, 64), shb = __shfl(seg_shift, lob, 64);
                code_a = la < ncand ? (int)clist[la + sha] : 2;     // (padding lanes: row 0, column 2, an address that exists)
                code_b = lb < ncand ? (int)clist[lb + shb] : 2;
            };
            auto request = [&](int code, float (&v)[25]) __attribute__((always_inline)) {
                const int i = code >> 10, x = code & 1023;
                int ro[5];
                int s0 = rq + i - 4;                    // slot of y_i - 2
                s0 += s0 < 0 ? RING : 0;
                // window row k sits in slot (s0 + k) mod RING: one 24-bit product for the first row (a 32-bit integer multiply runs at a quarter of
                // the rate), the others a constant further on, less the ring's length from the row on that wraps
                const int base = (int)__umul24((unsigned)s0, (unsigned)WP) + (PAD - 2) + x, kw = RING - s0;
                ro[0] = base;
#pragma unroll
                for (int k = 1; k < 5; k++) ro[k] = (k >= kw ? base - RING * WP : base) + k * WP;
                fit_load(s_dog, ro, v);
            };
            if (nchunks > 0) {
                code_of2(0, 1, code_cur, code_nxt);
                request(code_cur, wv_cur);
            }
#ifndef EDGEHIP_FIT_UNROLL
#define EDGEHIP_FIT_UNROLL 1   // 2785 -> 2733 us per 1024 frames (same-box A/B, tools/experiments/CALLS.md: r04_g)
#endif
            // one chunk: evaluate the window in `wc` (requested one chunk earlier) while the next chunk's window lands in `wn`
            auto one_chunk = [&](int c, float (&wc)[25], float (&wn)[25]) __attribute__((always_inline)) {
                    // the list search of chunk c + 2, one step per window row of this chunk's fit: a step's LDS round trip passes under a
                    // row's arithmetic (2440 -> 2350 us per 1024 frames against the search in one piece in front of the fit, same box,
                    // profiles/r06_fused_scan_fit_directed_ab.txt).  Its first probe goes out ahead of the next chunk's 25 window reads:
                    // LDS answers in order, a probe behind them would wait for all of them.
                    const int li2 = (c + 2) * 64 + lane;
                    int s_lo = 0, s_e = __shfl(seg_end, 15, 64), s_sh = 0;
                    if (c + 1 < nchunks) request(code_nxt, wn);
                    const bool on = c * 64 + lane < ncand;
                    const int code = code_cur;
                    const int i = code >> 10, x = code & 1023;
                    FitOut f;
                    if (ABL & 256) { f.cand = (code & 1) != 0; f.mx = 3.f; f.my = 4.f; f.xs = 0.f; f.ys = 0.f; }
                    else f = fit_eval(wc, fc, thr_d, [&](int i) __attribute__((always_inline)) {
                        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would gather the five steps in front of the arithmetic again)
                        if (i == 0) { s_lo += s_e <= li2 ? 16 : 0; s_e = __shfl(seg_end, s_lo + 7, 64); }
                        if (i == 1) { s_lo += s_e <= li2 ? 8 : 0; s_e = __shfl(seg_end, s_lo + 3, 64); }
                        if (i == 2) { s_lo += s_e <= li2 ? 4 : 0; s_e = __shfl(seg_end, s_lo + 1, 64); }
                        if (i == 3) { s_lo += s_e <= li2 ? 2 : 0; s_e = __shfl(seg_end, s_lo, 64); }
                        if (i == 4) { s_lo += s_e <= li2 ? 1 : 0; s_sh = __shfl(seg_shift, s_lo, 64); }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if (ABL & 256) {   // (the search alone, for the timing experiment without the plane fit)
                        s_lo = 0;
                        for (int step = 16; step > 0; step >>= 1) s_lo += __shfl(seg_end, s_lo + step - 1, 64) <= li2 ? step : 0;
                        s_sh = __shfl(seg_shift, s_lo, 64);
                    }
                    code_nn = li2 < ncand ? (int)clist[li2 + s_sh] : 2;
                    const bool fin = on && f.cand;
                    const unsigned long long bal = __ballot(fin);
                    const int id = total + below(bal);
                    if (fin && id < a.kl_max && !(ABL & 128)) {
                        // KeyLine `id` (edge_finder.cpp:166-200): what the fit produced; k_join_histo<true, true> derives the rest
                        const int y = ytest0 + i;
#if EDGEHIP_NT_EMIT
                        __builtin_nontemporal_store(y * w + x, k_pinx + id);
                        __builtin_nontemporal_store(f4v{f.xs, f.ys, f.mx, f.my}, k_grec + id);
                        __builtin_nontemporal_store(-1, k_pid + id);
#else
                        k_pinx[id] = y * w + x;
                        k_grec[id] = f4v{f.xs, f.ys, f.mx, f.my};
                        k_pid[id] = -1;        // join_edges' atomicMax needs it before any thread of k_join_histo runs
#endif
                        const float n2m = f.mx * f.mx + f.my * f.my;   // n_m = sqrtf(n2m) is monotonic in n2m: extremes of n2m here,
                        nm_mx = fmaxf(nm_mx, n2m);                     // one square root at the end of the frame
                        nm_mn = fminf(nm_mn, n2m);
                        s_res[((t & 1) * RB + i) * WRES + x] = (uint16_t)(id + 1);
                    }
                    total += __popcll(bal);
                    __builtin_amdgcn_sched_barrier(0);
                    code_cur = code_nxt;
                    code_nxt = code_nn;
            };
            static_assert(EDGEHIP_FIT_UNROLL == 1, "the copying form (wv_cur = wv_nxt after every chunk: 25 moves) is gone");
            // the window buffers swap roles from chunk to chunk: two chunks per trip, no copy of the 25 values in between (the
            // buffer a run of chunks starts with is whichever the chunks before it left the requested window in)
            bool in_cur = true;
            auto chunks = [&](int c0, int c1) __attribute__((always_inline)) {
                int c = c0;
                if (c < c1 && !in_cur) { one_chunk(c, wv_nxt, wv_cur); in_cur = true; c++; }
                for (; c + 1 < c1; c += 2) { one_chunk(c, wv_cur, wv_nxt); one_chunk(c + 1, wv_nxt, wv_cur); }
                if (c < c1) { one_chunk(c, wv_cur, wv_nxt); in_cur = false; }
            };
#ifndef EDGEHIP_FIT_SPLIT
#define EDGEHIP_FIT_SPLIT -1   // the first half of the tick also holds the tick's set-up (segment scan, the first two searches): one chunk fewer there, 2301 -> 2273 us per 1024 frames
#endif
            const int half = (nchunks + EDGEHIP_FIT_SPLIT > 0 ? nchunks + EDGEHIP_FIT_SPLIT : 0) >> 1;
            chunks(0, half);
            EH_TS_BAR(0, t)
            chunks(half, nchunks);
            EH_TS_BAR(2, t)
        }
        EH_TS_OUT(3)
        // end of frame: reEstimateThresh's extremes (edge_finder.cpp:376-382) and the candidate count
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nm_mx = fmaxf(nm_mx, __shfl_xor(nm_mx, o, 64));
            nm_mn = fminf(nm_mn, __shfl_xor(nm_mn, o, 64));
        }
        if (lane == 0) { s_red[0] = sqrtf(nm_mx); s_red[1] = sqrtf(nm_mn); s_red[2] = __int_as_float(total); }
    } else {
    // ---- column waves ---------------------------------------------------------------------------------------------------
    const int wv = wave < scan_wave ? wave : wave - 1;   // 0..NW-1
    const int x0 = 2 * (wv * 64 + lane);        // owned columns x0, x0 + 1
    const bool act = x0 < w;                    // w % 4 == 0: both or neither
    const int xr0 = act ? x0 : w - 2;           // address column of inactive threads
    // ... and where their LDS stores go: the first two floats of the row's right pad.  In the row sets the scan wave writes the
    // row's total there AFTER the column waves have stored the row (one tick later), in the DoG ring nobody reads the pad — so the
    // stores of a tick need no "is this lane inside the image" mask
    const int xs0 = act ? x0 : w;
    auto CX = [&](int x, int r) { const int rr = x + r < w - 1 ? x + r : w - 1; const int l = x - r - 1; return rr - (l > -1 ? l : -1); };   // box width along x
    // div(x,y) of rows with the full box height: (float)(1.0/(d*d)), except in the few columns at the left and right image border
    // whose box is clipped in x.  A per-thread constant of (box radius, box width): read from the table once, here (the five levels
    // have two distinct (r, d) pairs with the shipped box widths — four registers; reading it per row cost the two waves that own
    // border columns 40 LDS reads a tick, and every tick waits for its slowest wave).
    auto mrow_of = [&](int r, int d) __attribute__((always_inline)) {
        v2f m;
        m.x = s_lut[CX(xr0, r) * d];
        m.y = s_lut[CX(xr0 + 1, r) * d];
        return m;
    };
    const v2f m1 = mrow_of(R1, D1), m2a = mrow_of(R2A, D2A), m2b = mrow_of(R2B, D2B), m3a = mrow_of(R3A, D3A), m3b = mrow_of(R3B, D3B);
    TapRing<L1> H1;
    TapRing<L2A> H2A;
    TapRing<L2B> H2B;
    TapRing<L3A> H3A;
    TapRing<L3B> H3B;
    const v2f zero2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < L1; k++) H1.a[k] = H1.b[k] = zero2;
#pragma unroll
    for (int k = 0; k < L2A; k++) H2A.a[k] = H2A.b[k] = zero2;
#pragma unroll
    for (int k = 0; k < L2B; k++) H2B.a[k] = H2B.b[k] = zero2;
#pragma unroll
    for (int k = 0; k < L3A; k++) H3A.a[k] = H3A.b[k] = zero2;
#pragma unroll
    for (int k = 0; k < L3B; k++) H3B.a[k] = H3B.b[k] = zero2;
    v2f iv[RB + 2];                             // img0 of the last RB + 2 rows: iv[k] = img0(DoG row of step k - 1)
#pragma unroll
    for (int k = 0; k < RB + 2; k++) iv[k] = zero2;
    uint32_t gbits0 = 0, gbits1 = 0;            // gradient-gate results of the last rows, newest in bit 0, per owned column
    // DoG sign balance (edge_finder.cpp:125-137), incrementally: per owned column the number of positive DoG values in the
    // 5-wide strip of each of the last five rows (3 bits each, newest lowest) and their sum = positives in the 5x5 window
    // of the row two above the newest; bbits = the test's outcome, newest in bit 0
    uint32_t hh0 = 0, hh1 = 0, ws0 = 0, ws1 = 0, bbits0 = 0, bbits1 = 0;
    // pn = 2 npos - 25 passes unless (double)|pn| > pn_thresh: the admissible npos range
    int np_lo = 26, np_hi = -1;
    for (int np = 0; np <= 25; np++) {
        const int pn = 2 * np - 25, apn = pn < 0 ? -pn : pn;
        if (!((double)apn > a.pn_thresh)) { np_lo = np < np_lo ? np : np_lo; np_hi = np; }
    }
    np_lo = __builtin_amdgcn_readfirstlane(np_lo);
    np_hi = __builtin_amdgcn_readfirstlane(np_hi);
    const uint8_t *frame = GREY16 ? reinterpret_cast<const uint8_t *>(a.grey16 + (size_t)seq * a.n)
                           : GREY8 ? a.grey8 + (size_t)(a.fidx ? a.fidx[seq] : seq) * a.n
                                   : a.rgb + (size_t)(a.fidx ? a.fidx[seq] : seq) * a.n * 3;

    int rq0 = 0;                                // ring slot of this tick's first DoG row
    int32_t *mask = a.mask + so;
    float *pl = DBG && a.planes ? a.planes + so : nullptr;
    const size_t pstride = (size_t)a.nseq * a.n;

    // (Round 4: a second copy of the tick for the ~110 of 137 ticks whose rows are all interior image rows — the ~70 wave-uniform
    // row tests and the general form of the box average folded away at compile time — measured SLOWER, 2610-2674 -> 2676-2738 us
    // per 1024 frames, profiles/r04_k_fused_branch_free_ab.txt: scalar compares and uniform branches ride along with the other wave's
    // vector instructions for free, a second 16 KB loop body does not.  What did pay is below: no divergent region around the
    // LDS stores and the neighbour reads of the two gates.)
    EH_TS_DECL
    auto tick = [&](const int t, auto tt_tag) __attribute__((always_inline)) {
        constexpr int TT = decltype(tt_tag)::value;     // t & 1: the buffer set, and the half of the long tap rings this tick writes
        EH_TS_BEGIN(t)
        constexpr int set = TT;
        // ================= phase 1a: img_mask_kl rows of the rows tested in tick t-2 ====================================
        // (the fit wave put the ids of their KeyLines into s_res during tick t-1), written once: KeyLine id or -1
        // (edge_finder.cpp:109, 198, 203-209)
        {
            const int ym0 = (t - 8) * RB - LB - 2;
            if (ym0 + RB - 1 >= 0 && ym0 < h && !(ABL & 8)) {
                uint32_t *rp = reinterpret_cast<uint32_t *>(s_res + (size_t)(((t - 1) & 1) * RB) * (NW * 128) + wv * 128) + lane;
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int y = ym0 + i;
                    const uint32_t r2 = rp[i * (NW * 64)];
                    rp[i * (NW * 64)] = 0;
                    if (y >= 0 && y < h && act)
                        // (a plain store: as a streaming store the kernel gains 0.5 % and k_join_histo, whose probes read these rows next, loses as much)
                        *reinterpret_cast<int2 *>(mask + (uint32_t)(y * w + x0)) = make_int2((int)(r2 & 0xFFFFu) - 1, (int)(r2 >> 16) - 1);
                }
            }
        }
        // ================= phase 1b: the box chain on the scanned rows of buffer set `set` ==================================
        v2f l1[RB], l2a[RB], l2b[RB];
        uint32_t ppack = 0;                             // DoG > 0 of the tick's rows at the owned columns, 2 bits per row
        // one per-thread address (column xr0 of the LDS array) + non-negative constants: the offsets fold into the DS
        // instructions' immediates instead of costing an address register each
        const float *lb = smem + xr0;
        const float *P0 = lb + ((set * 4 + 0) * RB) * WP;
        const float *P1 = lb + ((set * 4 + 1) * RB) * WP;
        const float *P2A = lb + ((set * 4 + 2) * RB) * WP;
        const float *P2B = lb + ((set * 4 + 3) * RB) * WP;
        const int y1in0 = (t - 2) * RB;                 // level 1 input row (image row) of slot 0
        const int y2in0 = (t - 4) * RB - R1;            // level 2 input row (level-1 row) of slot 0
        const int y3ain0 = (t - 6) * RB - R1 - R2A;     // level 3 input rows
        const int y3bin0 = (t - 6) * RB - R1 - R2B;
        const int ydog0 = (t - 6) * RB - LB;            // DoG / img1 row of slot 0 (img0 row is one below)
        // steady ticks: every row of every level is an image row with its full box height
        const bool steady = y3bin0 >= D3B && y3ain0 >= D3A && y1in0 + RB - 1 <= h - 1;
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 2) { l1[j] = l2a[j] = l2b[j] = zero2; continue; }
            constexpr int dummy = 0; (void)dummy;
            const int q = TT * RB + j;                  // row of the unrolled tick pair: ring positions
            const int slot = rq0 + j >= RING ? rq0 + j - RING : rq0 + j;
            // the row's ten tap pairs (unconditional: rows that do not exist read garbage that is never used)
            const v2f t3a_r = ld2(P2A + j * WP + (PAD + R3A)), t3a_l = ld2(P2A + j * WP + (PAD - R3A - 1));
            const v2f t3b_r = ld2(P2B + j * WP + (PAD + R3B)), t3b_l = ld2(P2B + j * WP + (PAD - R3B - 1));
            const v2f t2a_r = ld2(P1 + j * WP + (PAD + R2A)), t2a_l = ld2(P1 + j * WP + (PAD - R2A - 1));
            const v2f t2b_r = ld2(P1 + j * WP + (PAD + R2B)), t2b_l = ld2(P1 + j * WP + (PAD - R2B - 1));
            const v2f t1_r = ld2(P0 + j * WP + (PAD + R1)), t1_l = ld2(P0 + j * WP + (PAD - R1 - 1));
            v2f i0n, i1;
            if (steady) {
                i0n = ring_row_steady<D3A, L3A>(H3A, q % L3A, t3a_r, t3a_l, m3a);   // img0 = G(sigma0) of row ydog+1
                i1 = ring_row_steady<D3B, L3B>(H3B, q % L3B, t3b_r, t3b_l, m3b);    // img1 = G(sigma1) of row ydog
                l2a[j] = ring_row_steady<D2A, L2A>(H2A, q % L2A, t2a_r, t2a_l, m2a);   // the two filters part ways at level 2
                l2b[j] = ring_row_steady<D2B, L2B>(H2B, q % L2B, t2b_r, t2b_l, m2b);
                l1[j] = ring_row_steady<D1, L1>(H1, q % L1, t1_r, t1_l, m1);          // level 1 is shared by both filters
            } else {
                i0n = ring_row_general<D3A, L3A>(H3A, q % L3A, t3a_r, t3a_l, CX(xr0, R3A), CX(xr0 + 1, R3A), s_lut, y3ain0 + j, h);
                i1 = ring_row_general<D3B, L3B>(H3B, q % L3B, t3b_r, t3b_l, CX(xr0, R3B), CX(xr0 + 1, R3B), s_lut, y3bin0 + j, h);
                l2a[j] = ring_row_general<D2A, L2A>(H2A, q % L2A, t2a_r, t2a_l, CX(xr0, R2A), CX(xr0 + 1, R2A), s_lut, y2in0 + j, h);
                l2b[j] = ring_row_general<D2B, L2B>(H2B, q % L2B, t2b_r, t2b_l, CX(xr0, R2B), CX(xr0 + 1, R2B), s_lut, y2in0 + j, h);
                l1[j] = ring_row_general<D1, L1>(H1, q % L1, t1_r, t1_l, CX(xr0, R1), CX(xr0 + 1, R1), s_lut, y1in0 + j, h);
            }
            {
                const int yd = ydog0 + j;                   // DoG row; iv[j+1] = img0 of that row
                const v2f dg = i1 - iv[j + 1];              // sspace.cpp:66
                *reinterpret_cast<v2f *>(smem + xs0 + (2 * 4 * RB * WP + PAD) + slot * WP) = dg;
                ppack |= ((act && dg.x > 0 ? 1u : 0u) | (act && dg.y > 0 ? 2u : 0u)) << (2 * j);
                iv[j + 2] = i0n;
                if (DBG && pl && act) {
                    if (yd >= 0 && yd < h) {
                        *reinterpret_cast<v2f *>(pl + 1 * pstride + (size_t)yd * w + x0) = i1;
                        *reinterpret_cast<v2f *>(pl + 2 * pstride + (size_t)yd * w + x0) = dg;
                    }
                    if (yd + 1 >= 0 && yd + 1 < h) *reinterpret_cast<v2f *>(pl + 0 * pstride + (size_t)(yd + 1) * w + x0) = i0n;
                }
            }
        }
        // img0 at the first / last column of every wave, rows of steps 0..RB-1 (iv[1..RB]): the gate's x-neighbours
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int j = 0; j < RB; j++)
                s_edge2[((size_t)j * (NW + 2) + wv + 1) * 2 + (lane ? 1 : 0)] =
                    make_uint2(__float_as_uint(lane ? iv[j + 1].y : iv[j + 1].x), (ppack >> (2 * j)) & 3u);
        }
        EH_TS_BAR(0, t)
        // ================= phase 2 ==============================================================================================
        // RGB rows of batch t: the loads fly under the stores and tests below and are used at the end of the phase
        uint2 pre[RB];                          // the 8 bytes that hold the two pixels' 6
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 16) { pre[j] = make_uint2(0, 0); continue; }
            int y = t * RB + j;
            y = y < h ? y : h - 1;
            if (UND) {
                // the first taps of the two pixels (the map is the same for every sequence: L2 hits); their weights and the taps
                // themselves follow at the end of the phase — a dependent round trip nothing hides (16 + 16 registers per row)
                pre[j] = *reinterpret_cast<const uint2 *>(a.und_base + (y * w + xr0));
            } else if (GREY16) {
                pre[j] = make_uint2(*reinterpret_cast<const uint32_t *>(frame + (uint32_t)(y * w + xr0) * 2u), 0u);   // two 16-bit values
            } else if (GREY8) {
                pre[j] = make_uint2(*reinterpret_cast<const uint16_t *>(frame + (uint32_t)(y * w + xr0)), 0u);        // two 8-bit values
            } else {
                const uint32_t byte0 = (uint32_t)(y * w + xr0) * 3u;     // < 2^31: the frame is w*h*3 bytes
                pre[j] = *reinterpret_cast<const uint2 *>(frame + (byte0 & ~3u));
            }
        }
        {
            float *Q1 = smem + xs0 + (((set * 4 + 1) * RB) * WP + PAD);
            float *Q2A = smem + xs0 + (((set * 4 + 2) * RB) * WP + PAD);
            float *Q2B = smem + xs0 + (((set * 4 + 3) * RB) * WP + PAD);
#pragma unroll
            for (int j = 0; j < RB; j++) {
                *reinterpret_cast<v2f *>(Q1 + j * WP) = l1[j];
                *reinterpret_cast<v2f *>(Q2A + j * WP) = l2a[j];
                *reinterpret_cast<v2f *>(Q2B + j * WP) = l2b[j];
            }
        }
        // gradient gate of the rows of steps 0..RB-1 (edge_finder.cpp:117-119, sspace.cpp:80-81)
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 32) continue;
            const int y = ydog0 + j;
            const v2f cv = iv[j + 1];                       // img0 of row y at columns x0, x0+1
            // neighbours in x: lane l-1 / l+1 of the wave (DPP wave shifts), the adjacent wave's edge lane through LDS
            float lft = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cv.y), 0x138, 0xf, 0xf, true));   // wave_shr:1 -> column x0 - 1
            float rgt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cv.x), 0x130, 0xf, 0xf, true));   // wave_shl:1 -> column x0 + 2
            const uint32_t pj = (ppack >> (2 * j)) & 3u;    // DoG > 0 at (x0, x0+1) in row y
            uint32_t pL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pj, 0x138, 0xf, 0xf, true);
            uint32_t pR = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pj, 0x130, 0xf, 0xf, true);
            {   // every lane reads the two records (wave-uniform addresses: LDS broadcasts), lanes 0 / 63 keep them: no branch
                const uint2 eL = s_edge2[((size_t)j * (NW + 2) + wv) * 2 + 1];        // last column of the wave to the left (zeros left of the image)
                const uint2 eR = s_edge2[((size_t)j * (NW + 2) + wv + 2) * 2 + 0];    // first column of the wave to the right
                lft = lane == 0 ? __uint_as_float(eL.x) : lft;
                pL = lane == 0 ? eL.y : pL;
                rgt = lane == 63 ? __uint_as_float(eR.x) : rgt;
                pR = lane == 63 ? eR.y : pR;
            }
            {   // positives among columns x-2..x+2 of this row, for x = x0 (pL, pj, low bit of pR) and x0+1 (high bit of pL, pj, pR)
                const uint32_t he = __popc((pL & 3u) | (pj << 2) | ((pR & 1u) << 4));
                const uint32_t ho = __popc(((pL >> 1) & 1u) | (pj << 1) | ((pR & 3u) << 3));
                ws0 = ws0 + he - ((hh0 >> 12) & 7u);        // window sum over the last five rows = npos of row y - 2
                ws1 = ws1 + ho - ((hh1 >> 12) & 7u);
                hh0 = ((hh0 << 3) | he) & 0x7FFFu;
                hh1 = ((hh1 << 3) | ho) & 0x7FFFu;
                bbits0 = (bbits0 << 1) | (((int)ws0 >= np_lo && (int)ws0 <= np_hi) ? 1u : 0u);
                bbits1 = (bbits1 << 1) | (((int)ws1 >= np_lo && (int)ws1 <= np_hi) ? 1u : 0u);
            }
            const bool yok = y >= 2 && y < h - 2;
            const bool v0 = yok && x0 >= 2 && x0 < w - 2, v1 = yok && x0 + 1 >= 2 && x0 + 1 < w - 2;
            const float dx0 = cv.y - lft, dx1 = rgt - cv.x;  // sspace.cpp:80
            const v2f dy = iv[j + 2] - iv[j];                // sspace.cpp:81
            if (DBG && pl) {
                if (v0) { pl[3 * pstride + (size_t)y * w + x0] = dx0; pl[4 * pstride + (size_t)y * w + x0] = dy.x; }
                if (v1) { pl[3 * pstride + (size_t)y * w + x0 + 1] = dx1; pl[4 * pstride + (size_t)y * w + x0 + 1] = dy.y; }
            }
            const float n0 = dx0 * dx0 + dy.x * dy.x, n1 = dx1 * dx1 + dy.y * dy.y;
            gbits0 = (gbits0 << 1) | ((v0 && !(n0 < thr_g)) ? 1u : 0u);
            gbits1 = (gbits1 << 1) | ((v1 && !(n1 < thr_g)) ? 1u : 0u);
        }
        iv[0] = iv[RB];
        iv[1] = iv[RB + 1];
        // build_mask's gates on rows ydog0 - 2 + i, i = 0..RB-1 (their DoG windows are complete with this tick's rows): the
        // survivors of the gradient gate and the sign balance go to the fit wave as one bit per pixel — the sparse rest of
        // build_mask (plane fit, sub-pixel test, ids, KeyLines, mask rows) runs there at full lanes, one tick behind
        if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < RB; i++) {
                // gradient gate of row y_i (bit 2 + RB-1-i) and sign balance of its window (complete with the newest row: bit RB-1-i)
                const bool p0 = (gbits0 >> (2 + RB - 1 - i)) & (bbits0 >> (RB - 1 - i)) & 1u, p1 = (gbits1 >> (2 + RB - 1 - i)) & (bbits1 >> (RB - 1 - i)) & 1u;
                const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
                const int sgm = i * NW + wv;
                uint16_t *seg = s_clist + ((size_t)set * RB * NW + sgm) * 128;
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                                (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                const int code = (i << 10) | (wv * 128 + 2 * lane);
                // (stores without a branch: a lane with nothing to publish writes the dummy slot, every lane the segment's count)
                uint16_t *d0 = p0 ? seg + pos : s_dummy;
                uint16_t *d1 = p1 ? seg + pos + (p0 ? 1 : 0) : s_dummy + 1;
                *d0 = (uint16_t)code;
                *d1 = (uint16_t)(code | 1);
                s_ccnt[set * RB * NW + sgm] = __popcll(b0) + __popcll(b1);
            }
        }
        if (UND) {   // image_undistort::undistort<true> (image_undistort.h:105-122) + ConvertRGB2BW for the thread's 2 x RB pixels
            float *Q0 = smem + xs0 + (((set * 4 + 0) * RB) * WP + PAD);
            uint4 iw[RB][2];
            uint64_t tt[RB][2], tu[RB][2];
#pragma unroll
            for (int j = 0; j < RB; j++) {
                int y = t * RB + j;
                y = y < h ? y : h - 1;
                const uint4 *wp = a.und_iw + (y * w + xr0);
                iw[j][0] = wp[0];
                iw[j][1] = wp[1];
                tt[j][0] = undist_row6(frame, (int)pre[j].x, (int)a.n);
                tu[j][0] = undist_row6(frame, (int)pre[j].x + w, (int)a.n);
                tt[j][1] = undist_row6(frame, (int)pre[j].y, (int)a.n);
                tu[j][1] = undist_row6(frame, (int)pre[j].y + w, (int)a.n);
            }
#pragma unroll
            for (int j = 0; j < RB; j++) {
                const uchar3 c0 = undist_mix(tt[j][0], tu[j][0], iw[j][0]), c1 = undist_mix(tt[j][1], tu[j][1], iw[j][1]);
                v2f g;
                g.x = (float)((int)c0.x + (int)c0.y + (int)c0.z);
                g.y = (float)((int)c1.x + (int)c1.y + (int)c1.z);
                *reinterpret_cast<v2f *>(Q0 + j * WP) = g;
            }
        } else {   // grey of batch t -> plane 0 of the buffer set (b+g+r, image.h:197-203: integers, exact in float)
            float *Q0 = smem + xs0 + (((set * 4 + 0) * RB) * WP + PAD);
#pragma unroll
            for (int j = 0; j < RB; j++) {
                int y = t * RB + j;
                y = y < h ? y : h - 1;
                v2f g;
                if (GREY16) {
                    g.x = (float)(int)(pre[j].x & 0xFFFFu);
                    g.y = (float)(int)(pre[j].x >> 16);
                } else if (GREY8) {
                    g.x = (float)(3 * (int)(pre[j].x & 0xFFu));     // b + g + r of a mono pixel
                    g.y = (float)(3 * (int)((pre[j].x >> 8) & 0xFFu));
                } else {
                    const unsigned sh = (((uint32_t)(y * w + x0) * 3u) & 3u) * 8u;   // 0 or 16
                    const unsigned long long q8 = (((unsigned long long)pre[j].y << 32) | pre[j].x) >> sh;
                    const unsigned lo = (unsigned)q8, hi = (unsigned)(q8 >> 24);
                    g.x = (float)((int)(lo & 0xFF) + (int)((lo >> 8) & 0xFF) + (int)((lo >> 16) & 0xFF));
                    g.y = (float)((int)(hi & 0xFF) + (int)((hi >> 8) & 0xFF) + (int)((hi >> 16) & 0xFF));
                }
                *reinterpret_cast<v2f *>(Q0 + j * WP) = g;
            }
        }
        rq0 += RB;
        rq0 -= rq0 >= RING ? RING : 0;
        EH_TS_BAR(2, t)
    };
    for (int t = 0; t < nticks; t += 2) {
#ifdef EDGEHIP_FUSED_TSTAMP
        if (t == 10) ts_a = wall_clock64();
        if (t == 110) ts_b = wall_clock64();
#endif
        tick(t, std::integral_constant<int, 0>{});
        tick(t + 1, std::integral_constant<int, 1>{});
    }
#ifdef EDGEHIP_FUSED_TSTAMP
    ts_loop = wall_clock64();
    if (wv == 0) { EH_TS_OUT(1) }
    if (wv == 2) { EH_TS_OUT(4) }
    if (wv == 5) { EH_TS_OUT(5) }
#endif

    }   // column waves
    __syncthreads();
    if (tid == 0) {   // kn and the P-controller state (edge_finder.cpp:355-364); reEstimateThresh's extremes (:376-382)
        const float nm_mx = s_red[0], nm_mn = s_red[1];
        const int total = __float_as_int(s_red[2]);
        const int kn = total < a.kl_max ? total : a.kl_max;
        const double tresh = __hiloint2double(__float_as_int(s_lut[kDivLutMax - 1]), __float_as_int(s_lut[kDivLutMax - 2]));   // update_thresh, computed at the top
        sq->tresh = tresh;
        sq->tresh_used = tresh;
        a.tresh_out[seq] = tresh;
        sq->l_kl_num = kn;
        sq->kn_new = kn;
#if !defined(EDGEHIP_FUSED_TSTAMP) || EDGEHIP_FUSED_TSTAMP < 10
        a.kn_out[seq] = kn;
#endif
        sq->nm_max = nm_mx;
        sq->nm_min = nm_mn;
#ifdef EDGEHIP_FUSED_TSTAMP
        {
            const long long ts_end = wall_clock64();
            const int sel = EDGEHIP_FUSED_TSTAMP;
            const long long v = sel == 1 ? ts_pro - ts_start : sel == 2 ? ts_a - ts_pro : sel == 3 ? ts_b - ts_a : sel == 4 ? ts_loop - ts_b
                              : sel == 5 ? ts_end - ts_loop : ts_end - ts_start;
            if (sel < 10) a.kn_out[seq] = (int)v;
        }
#endif
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
constexpr int kFusedRB = 4;

static int fused_col_waves(int w) { return (w + 127) / 128; }

size_t fused_lds_bytes(int w) {
    const int nw = fused_col_waves(w), WP = fused_row_stride(w), RB = kFusedRB;
    const size_t fl = (size_t)2 * 4 * RB * WP + (size_t)(2 * RB + 4) * WP + 32 + kDivLutMax + (size_t)RB * (nw + 2) * 4 + 4 + (size_t)2 * RB * nw;
    return fl * 4 + (size_t)4 * nw * RB * 128 * 2 + 8;
}

bool fused_supported(const edgehip_ctx *c) {
    const DevicePlan &pl = c->plan;
    if (pl.box[0][0] != 3 || pl.box[0][1] != 3 || pl.box[0][2] != 5) return false;
    if (pl.box[1][0] != 3 || pl.box[1][1] != 5 || pl.box[1][2] != 5) return false;
    if (c->p.plane_fit_size != 2) return false;
    if ((pl.w & 3) != 0) return false;                       // column pairs, float4 scan steps
    if (c->pinv_host[2] != 0.0 || c->pinv_host[25 + 10] != 0.0) return false;   // plane_fit5 skips these terms
    {   // ... and forms the three products of a window value from one (fit_eval): PInv = {-2u, -u, 0, u, 2u} x / y, 2u
        const double *pv = c->pinv_host, u = pv[3];
        const double want[5] = {-2 * u, -u, 0.0, u, 2 * u};
        for (int j = 0; j < 5; j++)
            if (pv[j] != want[j] || pv[25 + 5 * j] != want[j]) return false;
        if (pv[50] != 2 * u || !(u > 0)) return false;
    }
    const int nw = fused_col_waves(pl.w);
    if (nw + 2 > 8) return false;                           // column waves + scan wave + fit wave; 256 VGPRs per thread need <= 8 waves per workgroup
    if (pl.w > 1023) return false;                          // candidate codes carry x in 10 bits
    if (kFusedRB * nw > 32) return false;                   // the fit wave searches the segment ends in 32 lanes
    if (pl.cap > 65534) return false;                       // ids travel through LDS as uint16
    return fused_lds_bytes(pl.w) <= 160 * 1024;
}

// From how many sequences per launch on the one-kernel stage A is the better choice (the dispatch rule of stage_a.hip; EDGEHIP_FUSED_MIN_BATCH overrides it with
// one number).  A workgroup's life does not depend on the batch, the multi-kernel path's time does, and below one sequence per CU the workgroups leave CUs to
// the previous frame's tracking.  Measured per instantiation (profiles/r06_fused_threshold.txt): the compile-time widths 752 / 640 from 32 sequences on (64
// sequences 62 -> 90 k frames/s); 320 (the reference's default GlobalConfig) from 64 (stage A alone: 32 sequences 221 us multi-kernel against 315, 64: 307 / 324,
// 128: 427 / 351, 256: 699 / 377); the run-time-width instantiations, several times slower per pixel, from 192 (376 x 240, 48 sequences: 68.9 k frames/s
// multi-kernel, 37.7 k one-kernel; 160: equal).  Mirrors the selection in stage_a_fused_enqueue.
int fused_min_batch_for(const edgehip_ctx *c, bool grey16, bool grey8) {
    const int w = c->plan.w;
    if (c->planes) return 192;
    if (grey16) return w == 640 ? 32 : 192;
    if (grey8) return w == 752 ? 32 : (w == 320 ? 64 : 192);
    return (w == 752 || w == 640) ? 32 : (w == 320 ? 64 : 192);
}

int stage_a_fused_enqueue(edgehip_ctx *c, int slot, const uint8_t *rgb_base, const int32_t *rgb_idx, const uint16_t *grey16,
                          const uint8_t *grey8, bool undist_in_load) {
    const DevicePlan &pl = c->plan;
    const int B = pl.nseq;
    const int nw = fused_col_waves(pl.w);
    FusedArgs a;
    a.rgb = rgb_base; a.fidx = rgb_idx;
    a.lut = c->div_lut;
    a.planes = c->planes;
    a.mask = maskof(c, slot);
    a.seq = c->seqa;
    a.kl = kldev(c, slot);
    a.histo = c->histo;
    a.kn_out = c->kn_slot + (size_t)slot * B;
    a.tresh_out = c->tresh_slot + (size_t)slot * B;
    a.w = pl.w; a.h = pl.h; a.nseq = B; a.n = pl.n;
    a.gain = c->p.auto_gain; a.tmax = c->p.max_thresh; a.tmin = c->p.min_thresh;
    a.kl_ref = c->p.reference_points;
    a.kl_max = c->p.max_points < pl.cap ? c->p.max_points : pl.cap;
    a.dog_thresh_f = (float)c->p.dog_thresh;
    const int ws = c->p.plane_fit_size;
    a.pn_thresh = (double)(((float)((2.0 * ws + 1.0) * (2.0 * ws + 1.0))) * (float)c->p.pos_neg_thresh);
    a.pinv = c->pinv;
    a.ppx = c->slot_cam[slot].ppx; a.ppy = c->slot_cam[slot].ppy;
#ifdef EDGEHIP_EXPERIMENTS
    a.ablate = getenv("EDGEHIP_FUSED_ABLATE") ? atoi(getenv("EDGEHIP_FUSED_ABLATE")) : 0;
#else
    a.ablate = 0;
#endif
    size_t sm = fused_lds_bytes(pl.w);
#ifdef EDGEHIP_EXPERIMENTS
    // occupancy experiment (tools/experiments/exp_fused_occupancy.sh): at narrow widths two workgroups fit a CU's LDS; unused dynamic
    // LDS on top forces one per CU again, so the same kernel can be timed at one and at two resident workgroups per CU
    static const size_t lds_pad = getenv("EDGEHIP_FUSED_LDS_PAD") ? (size_t)atoi(getenv("EDGEHIP_FUSED_LDS_PAD")) : 0;
    if (sm + lds_pad <= 160 * 1024) sm += lds_pad;
#endif
    a.grey16 = grey16;
    a.grey8 = grey8;
    a.und_base = undist_in_load ? c->und_base : nullptr;
    a.und_iw = undist_in_load ? c->und_iw : nullptr;
    // the shipped image sizes get their own instantiation (EuRoC 752 from RGB or mono, TUM 640 from the undistorted grey
    // plane, the default GlobalConfig's 320 from RGB or mono), any other width — and contexts with debug planes — the generic ones
#define EH_FUSED(WW, DBG, SRC) k_stage_a_fused<WW, DBG, SRC, kFusedRB, 3, 3, 5, 5, 5>
    void (*fn)(FusedArgs);
#ifdef EDGEHIP_EXPERIMENTS   // the undistortion inside this kernel's load: measured 14 % slower than the pre-pass (EDGEHIP_FUSED_UNDIST)
    if (undist_in_load) {
        fn = EH_FUSED(0, false, SRC_UNDIST);
        if (c->planes) fn = EH_FUSED(0, true, SRC_UNDIST);
        else if (pl.w == 640) fn = EH_FUSED(640, false, SRC_UNDIST);
    } else
#else
    if (undist_in_load) { set_error("stage A: the undistortion inside the one-kernel load is an EXPERIMENTS build option"); return EDGEHIP_ERR_STATE; }
#endif
    if (grey16) {
        fn = EH_FUSED(0, false, SRC_GREY16);
        if (c->planes) fn = EH_FUSED(0, true, SRC_GREY16);
        else if (pl.w == 640) fn = EH_FUSED(640, false, SRC_GREY16);
    } else if (grey8) {
        fn = EH_FUSED(0, false, SRC_GREY8);
        if (c->planes) fn = EH_FUSED(0, true, SRC_GREY8);
        else if (pl.w == 752) fn = EH_FUSED(752, false, SRC_GREY8);
        else if (pl.w == 320) fn = EH_FUSED(320, false, SRC_GREY8);
    } else {
        fn = EH_FUSED(0, false, SRC_RGB24);
        if (c->planes) fn = EH_FUSED(0, true, SRC_RGB24);
        else if (pl.w == 752) fn = EH_FUSED(752, false, SRC_RGB24);
        else if (pl.w == 640) fn = EH_FUSED(640, false, SRC_RGB24);
        else if (pl.w == 320) fn = EH_FUSED(320, false, SRC_RGB24);   // the reference's default GlobalConfig (320 x 240: a live camera)
#ifdef EDGEHIP_EXPERIMENTS   // the occupancy experiment's widths with compile-time LDS offsets (the generic instantiation is several times slower)
        else if (pl.w == 256) fn = EH_FUSED(256, false, SRC_RGB24);
        else if (pl.w == 384) fn = EH_FUSED(384, false, SRC_RGB24);
#endif
    }
    if (!c->lds_optin_fused) {
        const void *fns[] = {
#ifdef EDGEHIP_EXPERIMENTS
                               (const void *)EH_FUSED(0, false, SRC_UNDIST), (const void *)EH_FUSED(0, true, SRC_UNDIST),
                               (const void *)EH_FUSED(640, false, SRC_UNDIST),
#endif
#ifdef EDGEHIP_EXPERIMENTS
                               (const void *)EH_FUSED(256, false, SRC_RGB24), (const void *)EH_FUSED(384, false, SRC_RGB24),
#endif
                               (const void *)EH_FUSED(0, false, SRC_RGB24), (const void *)EH_FUSED(0, true, SRC_RGB24),
                               (const void *)EH_FUSED(752, false, SRC_RGB24), (const void *)EH_FUSED(640, false, SRC_RGB24),
                               (const void *)EH_FUSED(320, false, SRC_RGB24), (const void *)EH_FUSED(320, false, SRC_GREY8),
                               (const void *)EH_FUSED(0, false, SRC_GREY16), (const void *)EH_FUSED(0, true, SRC_GREY16),
                               (const void *)EH_FUSED(640, false, SRC_GREY16), (const void *)EH_FUSED(0, false, SRC_GREY8),
                               (const void *)EH_FUSED(0, true, SRC_GREY8), (const void *)EH_FUSED(752, false, SRC_GREY8)};
        for (const void *f : fns) EH_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->lds_optin_fused = true;
    }
#undef EH_FUSED
    {
        ProfScope ps(c, PROF_A_FUSED, c->stream_a);
        hipLaunchKernelGGL(fn, dim3(B), dim3((nw + 2) * 64), sm, c->stream_a, a);
        EH_LAUNCH_CHECK();
    }
    c->grec_ok[slot] = true;   // freshly detected KeyLines: u_m = m_m / |m_m| holds for all of them
    c->rec_stale[slot] = false;
    c->rot_pending[slot] = false;
    return 0;
}

}  // namespace edgehip
