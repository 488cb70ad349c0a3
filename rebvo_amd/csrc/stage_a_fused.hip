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
// the detector's taps and writes the mask twice; this one moves 3 N + 4 N + the KeyLine records (SURVEY 8d's stage-A bytes).
//
// How the box chain stays on chip.  A box average needs four taps of the integral image: bottom row y+r and top row
// y-r-1, columns x+r and x-r-1.  The integral image itself is never materialised:
//   * the serial left-to-right row prefix of a row (iimage.cpp:56-61, order-defining in float32) is done in LDS by the
//     scan wave, one lane per row, as in k_level;
//   * the serial top-to-bottom column prefix (iimage.cpp:63-67) is a running sum; the thread that owns column x keeps the
//     running sums of ITS TWO TAP COLUMNS x+r and x-r-1 (every column sum is therefore computed by two threads, with the
//     same operands in the same order, hence the same bits) — no exchange of integral values between threads;
//   * the top taps of output row y are the bottom taps of output row y-d: a register history of d entries per tap column.
// So a level costs, per pixel: two LDS reads of row-prefixed values, two adds, the four-tap combine.  The levels are
// chained through LDS row buffers that hold RB rows each: produced in tick t, row-scanned in tick t+1, consumed in
// tick t+2 (two buffer sets, alternating).  Level l's rows trail its input by r rows, so img0 = G(sigma0) trails the
// input by r1+r2a+r3a rows and img1 by r1+r2b+r3b; the kernel is instantiated for the box widths of the shipped
// configurations ({3,3,5} / {3,5,5}: Sigma0 1.7818, KSigma 1.2599), everything else takes the multi-kernel path.
//
// A tick (RB image rows) of the column waves:
//   phase 1   emit the KeyLines found in the previous tick (ids need every wave's counts: published before the barrier)
//             and their img_mask_kl rows; grey values of the prefetched RGB rows; the five box averages of the tick
//             from the scanned rows of buffer set t&1; DoG rows into the LDS ring
//   -- barrier (LDS only) --
//   phase 2   store the produced rows into buffer set t&1 (all its readers are past the barrier); gradient gate;
//             build_mask's tests on the rows whose 5x5 DoG window is complete; publish per-segment counts
//   -- barrier --
// while the scan wave row-scans buffer set (t+1)&1, half before and half after the middle barrier.
// KeyLine ids are raster-order ranks (edge_finder.cpp:166-200): the workgroup sees the rows in order, so an id is the
// running total + an exclusive scan over the tick's (row, column group) segments — no staging, no second kernel.
//
// Compile with -ffp-contract=off (the reference is built without FMA contraction).

#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "ctx.h"
#include "stage_a_dev.h"

namespace edgehip {

// Running column sums of a level's two tap columns x+r and x-r-1 (age 0) and their values of the last D integral rows.
template <int D>
struct TapHist {
    float a[D + 1], b[D + 1];
};

// One row arrives for a level: yin = its index as a row of the level's input, vr / vl = the row-prefixed input values at the
// tap columns x+r / x-r-1 (prow = the LDS row).  Returns the box average of output row yin - r (0 for rows that do not
// exist) — iimage::average (iimage.cpp:86-128) on an integral image that is never stored:
//   image rows (0 <= yin < h) add onto the running column sums (iimage.cpp:63-67: img(x,y) += img(x,y-1));
//   the r virtual rows below the image leave them on row h-1 and take the bottom band's operand order ((A-C)-B)+D;
//   top taps above the image are the zeros the history starts with (x - 0 is exact); taps left of the image read the
//   zero pad of the LDS row.
// Every condition is wave-uniform (a scalar branch); the branches also keep the instruction scheduler from pulling the
// tap loads of all rows to the front, which costs more registers than there are.
template <int D>
__device__ __forceinline__ float level_row(TapHist<D> &H, const float *prow, int xr, int xl, int cx, bool xclip, float mu,
                                           const float *s_lut, int yin, int h) {
    constexpr int R = D / 2;
    float out = 0.f;
    if (yin >= 0 && yin < h + R) {
#pragma unroll
        for (int k = D; k >= 1; k--) { H.a[k] = H.a[k - 1]; H.b[k] = H.b[k - 1]; }
        if (yin < h) {
            H.a[0] = H.a[0] + prow[xr];
            H.b[0] = H.b[0] + prow[xl];
            float m = mu;
            if (yin < D) m = s_lut[cx * (yin + 1)];          // box clipped by the top border: div(x,y) = 1/count
            else if (xclip) m = s_lut[cx * D];
            out = (((H.a[0] - H.b[0]) - H.a[D]) + H.b[D]) * m;
        } else {
            const float m = s_lut[cx * (h - yin + D - 1)];
            out = (((H.a[0] - H.a[D]) - H.b[0]) + H.b[D]) * m;
        }
    }
    return out;
}

// build_mask's plane fit on the 5x5 DoG window (edge_finder.cpp:139-159), from the LDS ring.  rs[k] = ring row of window
// row k.  Same operation order as k_detect (TooN dot product, k = 0..24).
struct FitOut { bool cand; float mx, my, xs, ys; };
__device__ __forceinline__ FitOut plane_fit5(const float *const rs[5], int x, const FusedArgs &a, float thr_d) {
    double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double yv = (double)rs[i][x + j - 2];
            t0 += a.pc0[j] * yv;
            t1 += a.pc1[i] * yv;
            t2 += a.pc2 * yv;
        }
    }
    FitOut o;
    o.cand = false;
    const double den = t0 * t0 + t1 * t1;
    o.xs = (float)(-t0 * t2 / den);
    o.ys = (float)(-t1 * t2 / den);
    o.mx = (float)t0;
    o.my = (float)t1;
    if (!(fabsf(o.xs) > 0.5f || fabsf(o.ys) > 0.5f)) {
        const float n2m = o.mx * o.mx + o.my * o.my;
        if (!(n2m < thr_d)) o.cand = true;
    }
    return o;
}

template <int RB, int MC, int D1, int D2A, int D2B, int D3A, int D3B>
__global__ __launch_bounds__(512) void k_stage_a_fused(FusedArgs a) {
    constexpr int R1 = D1 / 2, R2A = D2A / 2, R2B = D2B / 2, R3A = D3A / 2, R3B = D3B / 2;
    constexpr int LB = R1 + R2B + R3B;                  // rows img1 trails the input by
    static_assert(R1 + R2A + R3A + 1 == LB, "img0 must lead img1 by exactly one row (it is held for one step)");
    constexpr int RING = 2 * RB + 4;                    // DoG rows in LDS: RB being written + RB + 4 being read
    constexpr int S = RB * MC;                          // (row, column group) segments a wave tests per tick
    constexpr int PAD = kFusedPad;
#ifdef EDGEHIP_EXPERIMENTS   // make EXPERIMENTS=1: phase ablation for timing experiments (wrong results by design)
    const int ABL = a.ablate;
#else
    constexpr int ABL = 0;
#endif

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = a.w, h = a.h;
    const int WP = fused_row_stride(w);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NW = (blockDim.x >> 6) - 1;               // column waves
    const int NC = NW * 64;
    const int G = MC * NW;                              // column groups of 64 per row
    const int seq = blockIdx.x;
    const size_t so = (size_t)seq * a.n;

    float *s_set = smem;                                            // [2][4][RB][WP]
    float *s_dog = s_set + (size_t)2 * 4 * RB * WP;                 // [RING][WP]
    float *s_lut = s_dog + (size_t)RING * WP + 32;                  // [kDivLutMax]   (+32: read overrun of the last scanned row)
    float *s_edge = s_lut + kDivLutMax;                             // [RB][G][2] img0 at the first / last lane of every group
    int *s_cnt = reinterpret_cast<int *>(s_edge + (size_t)RB * G * 2);   // [RB*G] final candidates per segment, raster order
    float *s_red = reinterpret_cast<float *>(s_cnt + RB * G);       // [2][NW] n_m extremes, then the frame's candidate count
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_red + 2 * NW + 2);     // [NW][S*64] per-wave candidate lists
    uint16_t *s_res = s_list + (size_t)NW * S * 64;                 // [NW][S*64] id + 1 of the KeyLine at a tested pixel, 0 = none

    // ---- set-up -------------------------------------------------------------------------------------------------------
    for (int i = tid; i < 2 * 4 * RB * PAD; i += blockDim.x) s_set[(size_t)(i / PAD) * WP + (i % PAD)] = 0.f;   // left pads: taps left of column 0
    for (int i = tid; i < kDivLutMax; i += blockDim.x) s_lut[i] = a.lut[i];
    for (int i = tid; i < NW * S * 64; i += blockDim.x) s_res[i] = 0;
    for (int i = tid; i < RB * G; i += blockDim.x) s_cnt[i] = 0;
    for (int i = tid; i < 256; i += blockDim.x) a.histo[(size_t)seq * 256 + i] = 0;   // reEstimateThresh's histogram (k_join_histo fills it)
    SeqA *sq = a.seq + seq;
    const double tresh = update_thresh(sq->tresh, sq->l_kl_num, a.kl_ref, a.gain, a.tmax, a.tmin);
    const float grad_thresh = (float)tresh;                         // build_mask takes float grad_thesh
    const float gt1 = grad_thresh * 765;                            // grad_thesh*max_img_value
    const float thr_g = gt1 * gt1;
    const float gt2 = gt1 * a.dog_thresh_f;
    const float thr_d = gt2 * gt2;
    __syncthreads();

    // number of ticks: the tests of tick t cover rows (t-6)*RB - LB - 2 + [0, RB), their KeyLines are emitted in tick t+1
    const int t_last = 6 + (h - 1 + LB + 2) / RB;                   // tick that tests row h-1
    const int nticks = t_last + 2;

    if (wave == 0) {
        // ---- the scan wave: lane l row-scans row l of the buffer set (plane-major, RB rows per plane) -------------------
        __builtin_amdgcn_s_setprio(3);
        const int n16 = w >> 4;                 // 16-float steps
        const int half16 = n16 >> 1;
        for (int t = 0; t < nticks; t++) {
            float *row = s_set + ((size_t)((t + 1) & 1) * 4 * RB + lane) * WP + PAD;
            float acc = 0.f;
            const bool on = lane < 4 * RB;
            auto steps16 = [&](int c0, int c1) {
                if (!on || (ABL & 1)) return;
                for (int c = c0; c < c1; c++) {
                    float4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<float4 *>(row + c * 16 + 4 * i);
#pragma unroll
                    for (int i = 0; i < 4; i++) {   // img(x,y) = img(x-1,y) + l(x,y), iimage.cpp:58-60
                        v[i].x = acc = acc + v[i].x;
                        v[i].y = acc = acc + v[i].y;
                        v[i].z = acc = acc + v[i].z;
                        v[i].w = acc = acc + v[i].w;
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(row + c * 16 + 4 * i) = v[i];
                }
            };
            steps16(0, half16);
            lds_barrier();
            steps16(half16, n16);
            if (on && !(ABL & 1)) {
                for (int c = n16 * 16; c < w; c += 4) {   // w % 16 != 0: up to three float4 steps
                    float4 v = *reinterpret_cast<float4 *>(row + c);
                    v.x = acc = acc + v.x;
                    v.y = acc = acc + v.y;
                    v.z = acc = acc + v.z;
                    v.w = acc = acc + v.w;
                    *reinterpret_cast<float4 *>(row + c) = v;
                }
            }
            lds_barrier();
        }
    } else {
    // ---- column waves ---------------------------------------------------------------------------------------------------
    const int wv = wave - 1;                    // 0..NW-1
    const int ct = tid - 64;
    int xc[MC];                                 // owned columns
    bool act[MC];
#pragma unroll
    for (int m = 0; m < MC; m++) { xc[m] = ct + m * NC; act[m] = xc[m] < w; }
    auto XR = [&](int x, int r) { const int v = x + r; return v < w - 1 ? v : w - 1; };
    auto XL = [&](int x, int r) { const int v = x - r - 1; return v < w - 1 ? v : w - 1; };      // >= -PAD: the zero pad
    auto CX = [&](int x, int r) { const int l = x - r - 1; return XR(x, r) - (l > -1 ? l : -1); };   // box width along x
    // div(x,y) of rows with the full box height: (float)(1.0/(d*d)) except in the few columns whose box is clipped in x
    const float mu1 = s_lut[D1 * D1], mu2a = s_lut[D2A * D2A], mu2b = s_lut[D2B * D2B], mu3a = s_lut[D3A * D3A], mu3b = s_lut[D3B * D3B];
    constexpr int RMAX = R3B > R2B ? (R3B > R1 ? R3B : R1) : (R2B > R1 ? R2B : R1);
    bool xclip[MC];                             // some box of this column is clipped by the left or right image border
    TapHist<D1> H1[MC];
    TapHist<D2A> H2A[MC];
    TapHist<D2B> H2B[MC];
    TapHist<D3A> H3A[MC];
    TapHist<D3B> H3B[MC];
    float iv[MC][RB + 2];                       // img0 of the last RB + 2 rows (own column): iv[k] = img0(row of step k-2 + 1)
    uint32_t gbits[MC];                         // gradient-gate results of the last rows, newest in bit 0
#pragma unroll
    for (int m = 0; m < MC; m++) {
        xclip[m] = xc[m] - RMAX - 1 < 0 || xc[m] + RMAX > w - 1;
#pragma unroll
        for (int k = 0; k <= D1; k++) H1[m].a[k] = H1[m].b[k] = 0.f;
#pragma unroll
        for (int k = 0; k <= D2A; k++) H2A[m].a[k] = H2A[m].b[k] = 0.f;
#pragma unroll
        for (int k = 0; k <= D2B; k++) H2B[m].a[k] = H2B[m].b[k] = 0.f;
#pragma unroll
        for (int k = 0; k <= D3A; k++) H3A[m].a[k] = H3A[m].b[k] = 0.f;
#pragma unroll
        for (int k = 0; k <= D3B; k++) H3B[m].a[k] = H3B[m].b[k] = 0.f;
#pragma unroll
        for (int k = 0; k < RB + 2; k++) iv[m][k] = 0.f;
        gbits[m] = 0;
    }
    const uint8_t *frame = a.rgb + (size_t)(a.fidx ? a.fidx[seq] : seq) * a.n * 3;
    uint16_t *my_list = s_list + (size_t)wv * S * 64;
    uint16_t *my_res = s_res + (size_t)wv * S * 64;
    int nfinal = 0;                             // final candidates of the previous tick in my_list
    int total = 0;                              // KeyLine candidates of the frame so far (workgroup-uniform)
    int rq0 = 0;                                // ring slot of this tick's first DoG row
    int rq_prev = 0;
    float nm_mx = 0.f, nm_mn = __int_as_float(0x7f800000);
    int32_t *mask = a.mask + so;
    const KlSoA &kl = a.kl[seq];
    float *pl = a.planes ? a.planes + so : nullptr;
    const size_t pstride = (size_t)a.nseq * a.n;

    // ring rows of the 5x5 window of test row i (rows y_i-2 .. y_i+2), for the tick whose first DoG row sits in slot rq
    auto window_rows = [&](int rq, int i, const float *rs[5]) {
        int s0 = rq + i - 4;                    // slot of y_i - 2 = (last DoG row of the tick) - (RB - 1 - i) - 4
        s0 += s0 < 0 ? RING : 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int sk = s0 + k;
            sk -= sk >= RING ? RING : 0;
            rs[k] = s_dog + (size_t)sk * WP + PAD;
        }
    };

    for (int t = 0; t < nticks; t++) {
        const int set = t & 1;
        // ================= phase 1a: KeyLines of the rows tested in tick t-1, and those rows of img_mask_kl ==============
        {
            const int ytest0 = (t - 7) * RB - LB - 2;           // first row tested in tick t-1
            if (ytest0 + RB - 1 >= 0 && ytest0 < h && !(ABL & 8)) {
                // exclusive scan of the segment counts in raster order (row-major, then column group)
                const int nseg = RB * G;                        // <= 64 (checked on the host)
                const int c = lane < nseg ? s_cnt[lane] : 0;
                int incl = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - c;
                const int tick_total = __shfl(incl, 63, 64);
                for (int base = 0; base < nfinal; base += 64) {
                    const int e = base + lane;
                    const bool on = e < nfinal;
                    const int code = on ? my_list[e] : 0;
                    const int s = code >> 6, i = s / MC, m = s - i * MC, ln = code & 63;
                    const int ridx = i * G + m * NW + wv;
                    const int off = __shfl(excl, ridx, 64);
                    // rank inside the segment: the list is segment-ordered, so e minus the entries of the wave's earlier segments
                    int before = 0;
#pragma unroll
                    for (int s2 = 0; s2 < S; s2++) {
                        const int i2 = s2 / MC, m2 = s2 - i2 * MC;
                        const int c2 = __shfl(c, i2 * G + m2 * NW + wv, 64);
                        before += s2 < s ? c2 : 0;
                    }
                    const int id = total + off + (e - before);
                    if (on && id < a.kl_max) {
                        const float *rs[5];
                        window_rows(rq_prev, i, rs);
                        const int x = ln + m * NC + wv * 64, y = ytest0 + i;
                        const FitOut f = plane_fit5(rs, x, a, thr_d);
                        // KeyLine `id` (edge_finder.cpp:166-200)
                        const int p = y * w + x;
                        const float n2m = f.mx * f.mx + f.my * f.my;
                        const float nm = sqrtf(n2m);
                        const float2 mm = make_float2(f.mx, f.my);
                        const float2 u = make_float2(f.mx / nm, f.my / nm);
                        const float2 cp = make_float2((float)x + f.xs, (float)y + f.ys);
                        const float2 pm = make_float2(cp.x - a.ppx, cp.y - a.ppy);      // cam_model::Img2Hom
                        kl.p_inx[id] = p;
                        kl.m_m[id] = mm;
                        kl.n_m[id] = nm;
                        kl.u_m[id] = u;
                        kl.c_p[id] = cp;
                        kl.p_m[id] = pm;
                        kl.p_m_0[id] = pm;
                        kl.rho[id] = 1.0;        // RhoInit
                        kl.s_rho[id] = 20.0;     // RHO_MAX
                        kl.rho0[id] = 1.0;
                        kl.s_rho0[id] = 20.0;
                        kl.rho_nr[id] = 1.0;
                        kl.s_rho_nr[id] = 20.0;
                        kl.m_num[id] = 0;
                        kl.n_id[id] = -1;
                        kl.p_id[id] = -1;
                        kl.m_id[id] = -1;
                        if (kl.stereo_m_id) { kl.stereo_m_id[id] = -1; kl.stereo_rho[id] = 1.0; kl.stereo_s_rho[id] = 20.0; }
                        kl.m_id_f[id] = -1;
                        kl.m_id_kf[id] = -1;
                        kl.m_m0[id] = make_float2(0.f, 0.f);
                        kl.n_m0[id] = 0.0;
                        MatchRec rec;
                        rec.c_px = cp.x; rec.c_py = cp.y; rec.u_mx = u.x; rec.u_my = u.y;
                        rec.m_mx = mm.x; rec.m_my = mm.y; rec.n_m = nm; rec.pad = 0.f;
                        kl.rec[id] = rec;
                        kl.grec[id] = make_float4(cp.x, cp.y, mm.x, mm.y);
                        nm_mx = fmaxf(nm_mx, nm);
                        nm_mn = fminf(nm_mn, nm);
                        my_res[s * 64 + ln] = (uint16_t)(id + 1);
                    }
                }
                total += tick_total;
                // the tested rows of img_mask_kl, written once: KeyLine id or -1 (edge_finder.cpp:109, 198, 203-209)
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const int y = ytest0 + i;
#pragma unroll
                    for (int m = 0; m < MC; m++) {
                        const int r = my_res[(i * MC + m) * 64 + lane];
                        my_res[(i * MC + m) * 64 + lane] = 0;
                        if (y >= 0 && y < h && act[m]) mask[(size_t)y * w + xc[m]] = r - 1;
                    }
                }
            }
        }
        // ================= phase 1b: the box chain on the scanned rows of buffer set `set` ==================================
        float l1[MC][RB], l2a[MC][RB], l2b[MC][RB];
        const float *P0 = s_set + ((size_t)(set * 4 + 0) * RB) * WP + PAD;
        const float *P1 = s_set + ((size_t)(set * 4 + 1) * RB) * WP + PAD;
        const float *P2A = s_set + ((size_t)(set * 4 + 2) * RB) * WP + PAD;
        const float *P2B = s_set + ((size_t)(set * 4 + 3) * RB) * WP + PAD;
        const int y1in0 = (t - 2) * RB;                 // level 1 input row (image row) of slot 0
        const int y2in0 = (t - 4) * RB - R1;            // level 2 input row (level-1 row) of slot 0
        const int y3ain0 = (t - 6) * RB - R1 - R2A;     // level 3 input rows
        const int y3bin0 = (t - 6) * RB - R1 - R2B;
        const int ydog0 = (t - 6) * RB - LB;            // DoG / img1 row of slot 0 (img0 row is one below)
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 2) { for (int m = 0; m < MC; m++) l1[m][j] = l2a[m][j] = l2b[m][j] = 0.f; continue; }
            const int slot = rq0 + j >= RING ? rq0 + j - RING : rq0 + j;
            float *dogrow = s_dog + (size_t)slot * WP + PAD;
#pragma unroll
            for (int m = 0; m < MC; m++) {
                const int x = act[m] ? xc[m] : w - 1;
                // ---- level 3: img0 = G(sigma0) of row ydog+1, img1 = G(sigma1) of row ydog ----
                const float i0n = level_row<D3A>(H3A[m], P2A + j * WP, XR(x, R3A), XL(x, R3A), CX(x, R3A), xclip[m], mu3a, s_lut, y3ain0 + j, h);
                const float i1 = level_row<D3B>(H3B[m], P2B + j * WP, XR(x, R3B), XL(x, R3B), CX(x, R3B), xclip[m], mu3b, s_lut, y3bin0 + j, h);
                {
                    const int yd = ydog0 + j;                   // DoG row; iv[j+1] = img0 of that row
                    const float dg = i1 - iv[m][j + 1];         // sspace.cpp:66
                    if (act[m]) dogrow[x] = dg;
                    iv[m][j + 2] = i0n;
                    if (pl && act[m]) {
                        if (yd >= 0 && yd < h) {
                            pl[1 * pstride + (size_t)yd * w + x] = i1;
                            pl[2 * pstride + (size_t)yd * w + x] = dg;
                        }
                        if (yd + 1 >= 0 && yd + 1 < h) pl[0 * pstride + (size_t)(yd + 1) * w + x] = i0n;
                    }
                }
                // ---- level 2: the two filters part ways (same input, box widths D2A / D2B) ----
                l2a[m][j] = level_row<D2A>(H2A[m], P1 + j * WP, XR(x, R2A), XL(x, R2A), CX(x, R2A), xclip[m], mu2a, s_lut, y2in0 + j, h);
                l2b[m][j] = level_row<D2B>(H2B[m], P1 + j * WP, XR(x, R2B), XL(x, R2B), CX(x, R2B), xclip[m], mu2b, s_lut, y2in0 + j, h);
                // ---- level 1 (shared by both filters) ----
                l1[m][j] = level_row<D1>(H1[m], P0 + j * WP, XR(x, R1), XL(x, R1), CX(x, R1), xclip[m], mu1, s_lut, y1in0 + j, h);
            }
        }
        // img0 at the first / last lane of every column group, rows of steps 0..RB-1 (iv[1..RB]): the gate's x-neighbours
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int m = 0; m < MC; m++)
#pragma unroll
                for (int j = 0; j < RB; j++) s_edge[((size_t)j * G + m * NW + wv) * 2 + (lane ? 1 : 0)] = iv[m][j + 1];
        }
        lds_barrier();
        // ================= phase 2 ==============================================================================================
        // RGB rows of batch t: the loads fly under the stores and tests below and are used at the end of the phase
        uint2 pre[MC][RB];                      // the 8 bytes that hold the pixel's 3
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 16) { for (int m = 0; m < MC; m++) pre[m][j] = make_uint2(0, 0); continue; }
            int y = t * RB + j;
            y = y < h ? y : h - 1;
#pragma unroll
            for (int m = 0; m < MC; m++) {
                const int x = act[m] ? xc[m] : w - 1;
                const size_t byte0 = ((size_t)y * w + x) * 3;
                pre[m][j] = *reinterpret_cast<const uint2 *>(frame + (byte0 & ~(size_t)3));
            }
        }
        {
            float *Q1 = s_set + ((size_t)(set * 4 + 1) * RB) * WP + PAD;
            float *Q2A = s_set + ((size_t)(set * 4 + 2) * RB) * WP + PAD;
            float *Q2B = s_set + ((size_t)(set * 4 + 3) * RB) * WP + PAD;
#pragma unroll
            for (int m = 0; m < MC; m++) {
                if (!act[m]) continue;
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    Q1[j * WP + xc[m]] = l1[m][j];
                    Q2A[j * WP + xc[m]] = l2a[m][j];
                    Q2B[j * WP + xc[m]] = l2b[m][j];
                }
            }
        }
        // gradient gate of the rows of steps 0..RB-1 (edge_finder.cpp:117-119, sspace.cpp:80-81)
#pragma unroll
        for (int j = 0; j < RB; j++) {
            if (ABL & 32) continue;
            const int y = ydog0 + j;
#pragma unroll
            for (int m = 0; m < MC; m++) {
                const float cv = iv[m][j + 1];
                float rgt = __shfl_down(cv, 1, 64), lft = __shfl_up(cv, 1, 64);
                const int g = m * NW + wv;
                if (lane == 63 && g + 1 < G) rgt = s_edge[((size_t)j * G + g + 1) * 2 + 0];
                if (lane == 0 && g > 0) lft = s_edge[((size_t)j * G + g - 1) * 2 + 1];
                const int x = xc[m];
                const bool valid = y >= 2 && y < h - 2 && x >= 2 && x < w - 2;
                const float dx = rgt - lft;                      // sspace.cpp:80
                const float dy = iv[m][j + 2] - iv[m][j];        // sspace.cpp:81
                if (pl && valid) {
                    pl[3 * pstride + (size_t)y * w + x] = dx;
                    pl[4 * pstride + (size_t)y * w + x] = dy;
                }
                const float n2g = dx * dx + dy * dy;
                const bool pass = valid && !(n2g < thr_g);
                gbits[m] = (gbits[m] << 1) | (pass ? 1u : 0u);
            }
        }
#pragma unroll
        for (int m = 0; m < MC; m++) { iv[m][0] = iv[m][RB]; iv[m][1] = iv[m][RB + 1]; }
        // build_mask's window tests on rows ydog0 + RB - 1 - 2 - (RB - 1 - i), i = 0..RB-1 (their DoG windows are complete)
        if (!(ABL & 4)) {
            int nlist = 0;
#pragma unroll
            for (int i = 0; i < RB; i++)
#pragma unroll
                for (int m = 0; m < MC; m++) {
                    const bool pass = (gbits[m] >> (2 + RB - 1 - i)) & 1u;
                    const unsigned long long bal = __ballot(pass);
                    if (pass) my_list[nlist + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(((i * MC + m) << 6) | lane);
                    nlist += __popcll(bal);
                }
            // DoG sign balance (edge_finder.cpp:125-137): compact in place
            int nkeep = 0;
            for (int base = 0; base < nlist; base += 64) {
                const int li = base + lane;
                bool keep = false;
                int code = 0;
                if (li < nlist) {
                    code = my_list[li];
                    const int s = code >> 6, i = s / MC, m = s - i * MC;
                    const int x = (code & 63) + m * NC + wv * 64;
                    const float *rs[5];
                    window_rows(rq0, i, rs);
                    int npos = 0;
#pragma unroll
                    for (int r = 0; r < 5; r++)
#pragma unroll
                        for (int q = -2; q <= 2; q++) npos += (rs[r][x + q] > 0) ? 1 : 0;
                    const int pn = 2 * npos - 25;
                    const int apn = pn < 0 ? -pn : pn;
                    keep = !((double)apn > a.pn_thresh);
                }
                const unsigned long long bal = __ballot(keep);
                if (keep) my_list[nkeep + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)code;
                nkeep += __popcll(bal);
            }
            // plane fit, sub-pixel position, DoG-gradient gate (:139-159): keep the finals, count them per segment
            int nfin = 0;
            int cnt[S];
#pragma unroll
            for (int s = 0; s < S; s++) cnt[s] = 0;
            for (int base = 0; base < nkeep; base += 64) {
                const int li = base + lane;
                bool cand = false;
                int code = 0;
                if (li < nkeep) {
                    code = my_list[li];
                    const int s = code >> 6, i = s / MC, m = s - i * MC;
                    const int x = (code & 63) + m * NC + wv * 64;
                    const float *rs[5];
                    window_rows(rq0, i, rs);
                    cand = plane_fit5(rs, x, a, thr_d).cand;
                }
                const unsigned long long bal = __ballot(cand);
                if (cand) my_list[nfin + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)code;
                nfin += __popcll(bal);
#pragma unroll
                for (int s = 0; s < S; s++) cnt[s] += __popcll(__ballot(cand && (code >> 6) == s));
            }
            nfinal = nfin;
            if (lane == 0) {
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int i = s / MC, m = s - i * MC;
                    s_cnt[i * G + m * NW + wv] = cnt[s];
                }
            }
        }
        {   // grey of batch t -> plane 0 of the buffer set (b+g+r, image.h:197-203: integers, exact in float)
            float *Q0 = s_set + ((size_t)(set * 4 + 0) * RB) * WP + PAD;
#pragma unroll
            for (int j = 0; j < RB; j++) {
                int y = t * RB + j;
                y = y < h ? y : h - 1;
#pragma unroll
                for (int m = 0; m < MC; m++) {
                    const int x = act[m] ? xc[m] : w - 1;
                    const unsigned sh = (unsigned)((((size_t)y * w + x) * 3) & 3) * 8;
                    const unsigned long long q8 = ((unsigned long long)pre[m][j].y << 32) | pre[m][j].x;
                    const unsigned p = (unsigned)(q8 >> sh);
                    if (act[m]) Q0[j * WP + x] = (float)((int)(p & 0xFF) + (int)((p >> 8) & 0xFF) + (int)((p >> 16) & 0xFF));
                }
            }
        }
        rq_prev = rq0;
        rq0 += RB;
        rq0 -= rq0 >= RING ? RING : 0;
        lds_barrier();
    }

    // ---- end of frame: kn, the P-controller state (edge_finder.cpp:355-364), reEstimateThresh's extremes (:376-382) ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nm_mx = fmaxf(nm_mx, __shfl_xor(nm_mx, o, 64));
        nm_mn = fminf(nm_mn, __shfl_xor(nm_mn, o, 64));
    }
    if (lane == 0) { s_red[wv] = nm_mx; s_red[NW + wv] = nm_mn; }
    if (tid == 64) s_red[2 * NW] = __int_as_float(total);
    }   // column waves
    __syncthreads();
    if (tid == 64) {
        float nm_mx = s_red[0], nm_mn = s_red[NW];
        for (int i = 1; i < NW; i++) { nm_mx = fmaxf(nm_mx, s_red[i]); nm_mn = fminf(nm_mn, s_red[NW + i]); }
        const int total = __float_as_int(s_red[2 * NW]);
        const int kn = total < a.kl_max ? total : a.kl_max;
        sq->tresh = tresh;
        sq->tresh_used = tresh;
        a.tresh_out[seq] = tresh;
        sq->l_kl_num = kn;
        sq->kn_new = kn;
        a.kn_out[seq] = kn;
        sq->nm_max = nm_mx;
        sq->nm_min = nm_mn;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
constexpr int kFusedRB = 4, kFusedMC = 2;

size_t fused_lds_bytes(int w, int nw) {
    const int WP = fused_row_stride(w), RB = kFusedRB, MC = kFusedMC, G = MC * nw, S = RB * MC;
    size_t fl = (size_t)2 * 4 * RB * WP + (size_t)(2 * RB + 4) * WP + 32 + kDivLutMax + (size_t)RB * G * 2 + RB * G + 2 * nw + 2;
    return fl * 4 + (size_t)2 * nw * S * 64 * 2;
}

bool fused_supported(const edgehip_ctx *c) {
    const DevicePlan &pl = c->plan;
    if (pl.box[0][0] != 3 || pl.box[0][1] != 3 || pl.box[0][2] != 5) return false;
    if (pl.box[1][0] != 3 || pl.box[1][1] != 5 || pl.box[1][2] != 5) return false;
    if (c->und_base) return false;                          // the undistorting source keeps the multi-kernel path
    if (c->p.plane_fit_size != 2) return false;
    const int nw = (pl.w + 64 * kFusedMC - 1) / (64 * kFusedMC);
    if (nw + 1 > 8) return false;                           // 256 VGPRs per thread need <= 8 waves per workgroup
    if (kFusedRB * kFusedMC * nw > 64) return false;        // one lane per segment in the id scan
    if (pl.cap > 65534) return false;                       // ids travel through LDS as uint16
    return fused_lds_bytes(pl.w, nw) <= 160 * 1024;
}

int stage_a_fused_enqueue(edgehip_ctx *c, int slot, const uint8_t *rgb_base, const int32_t *rgb_idx) {
    const DevicePlan &pl = c->plan;
    const int B = pl.nseq;
    const int nw = (pl.w + 64 * kFusedMC - 1) / (64 * kFusedMC);
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
    for (int j = 0; j < 5; j++) { a.pc0[j] = c->pinv_host[j]; a.pc1[j] = c->pinv_host[25 + 5 * j]; }
    a.pc2 = c->pinv_host[50];
    a.ppx = c->slot_cam[slot].ppx; a.ppy = c->slot_cam[slot].ppy;
#ifdef EDGEHIP_EXPERIMENTS
    a.ablate = getenv("EDGEHIP_FUSED_ABLATE") ? atoi(getenv("EDGEHIP_FUSED_ABLATE")) : 0;
#else
    a.ablate = 0;
#endif
    const size_t sm = fused_lds_bytes(pl.w, nw);
    auto fn = k_stage_a_fused<kFusedRB, kFusedMC, 3, 3, 5, 5, 5>;
    if (!c->lds_optin_fused) {
        EH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->lds_optin_fused = true;
    }
    {
        ProfScope ps(c, PROF_A_FUSED, c->stream_a);
        hipLaunchKernelGGL(fn, dim3(B), dim3((nw + 1) * 64), sm, c->stream_a, a);
        EH_LAUNCH_CHECK();
    }
    c->grec_ok[slot] = true;   // freshly detected KeyLines: u_m = m_m / |m_m| holds for all of them
    return 0;
}

}  // namespace edgehip
