// stage_c.hip — matching and mapping: forward match, KeyLine rotation, directed (epipolar) matching,
// depth regularisation, the per-KeyLine inverse-depth EKF, rescaling, and the per-frame driver that strings
// stage A/B/C together without host synchronisation.
//
// Replaces (reference file:line)
//   edge_tracker::FordwardMatch              src/mtracklib/edge_tracker.cpp:380-436
//   edge_tracker::rotate_keylines            edge_tracker.cpp:42-76
//   edge_tracker::directed_matching/search_match   edge_tracker.cpp:302-374, 158-295
//   edge_tracker::Regularize_1_iter          edge_tracker.cpp:87-148
//   edge_tracker::UpdateInverseDepthKalman(ARLU)   edge_tracker.cpp:695-724, 954-1055
//   edge_tracker::EstimateReScalingOpt       edge_tracker.cpp:1104-1140
//   REBVO::SecondThread glue (ImuMode==0)    src/rebvo/rebvo_second_t.cpp:145-178, 341-422, 453-487, 550-606
//
// Parallel restatements of sequential rules:
//  * FordwardMatch: several old KeyLines may forward-match the same new KeyLine; sequentially the winner
//    is the one with the largest rho, the LAST one among equals (:413).  Two atomicMax passes: first on
//    the order-preserving bit pattern of rho, then on the KeyLine index among those that hold the maximum.
//  * directed_matching / Regularize / EKF are independent per KeyLine (Regularize reads the PRE-update
//    neighbour values: its results go through a scratch array before the EKF consumes them).
//  * Whole frames (ImuMode 0, no stereo pair) match in one pass: the arbitration rides on rotate_keylines' pass over the old
//    KeyLines, which writes the turned values next to the slot's own arrays (k_rotate<OUT, WIN>), and FordwardMatch's copy
//    happens inside the directed-matching kernel (k_directed_fused), which writes every new KeyLine's ten matching fields
//    once — the directed match's, else the forward match's, else a fresh KeyLine's.  The three-kernel form below
//    (k_fwd_win, k_fwd_apply, k_rotate in place, k_directed) serves the stage-level entry points, the IMU branch and stereo.

#include <math.h>
#include <string.h>

#include <vector>

#include "ctx.h"
#include "wave_reduce.h"

namespace edgehip {

constexpr double kRhoMax = 20.0, kRhoMin = 1e-3, kRhoInit = 1.0;  // edge_finder.h:38-40

// ---------------------------------------------------------------------------------------------------
// FordwardMatch
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_key(const KlSoA *kl_old, const int32_t *__restrict__ kn_old,
                                                 const int32_t *__restrict__ kn_new, unsigned long long *__restrict__ key,
                                                 int cap) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kn_old[seq]) return;
    const int f = kl_old[seq].m_id_f[i];
    if (f < 0 || f >= kn_new[seq]) return;
    atomicMax(&key[(size_t)seq * cap + f], ord_bits(kl_old[seq].rho[i]));
}
__device__ inline void so3_exp_c(const double w[3], double R[9]);
__device__ inline void rot_from_state(SeqDev *sq, double *__restrict__ Rbuf_seq);
__device__ inline void glue_after_tracking(SeqDev *sq);
// tail_seqs != null (whole-frame driver, ImuMode 0): the first thread of a sequence's first block also does the two pieces of
// per-sequence scalar work that follow the minimiser — R0 = exp(W) for rotate_keylines and the NaN check of
// rebvo_second_t.cpp:387-397 — which nothing in this kernel or in k_fwd_apply reads: two dependent launches fewer per frame.
__global__ __launch_bounds__(256) void k_fwd_win(const KlSoA *kl_old, const int32_t *__restrict__ kn_old,
                                                 const int32_t *__restrict__ kn_new, const unsigned long long *__restrict__ key,
                                                 int32_t *__restrict__ win, int cap, SeqDev *tail_seqs, double *__restrict__ tail_Rbuf) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (tail_seqs && i == 0) {
        rot_from_state(tail_seqs + seq, tail_Rbuf + (size_t)seq * 9);
        glue_after_tracking(tail_seqs + seq);
    }
    if (i >= kn_old[seq]) return;
    const int f = kl_old[seq].m_id_f[i];
    if (f < 0 || f >= kn_new[seq]) return;
    if (key[(size_t)seq * cap + f] == ord_bits(kl_old[seq].rho[i])) atomicMax(&win[(size_t)seq * cap + f], i);
}
// fill: the detector of the new edge map left these ten fields unwritten (k_join_histo's fwd_fills): every new KeyLine gets
// them here, the forwarded values or the defaults of a fresh KeyLine (edge_finder.cpp:176-196) — whole-wave stores of whole
// lines.  (Masked stores, 85 % of the lanes, leave three lines in four partially written, which the memory system turns
// into read-modify-write.)
template <bool FILL>
__global__ __launch_bounds__(256) void k_fwd_apply(const KlSoA *kl_old, const KlSoA *kl_new, const int32_t *__restrict__ kn_new,
                                                   const int32_t *__restrict__ win, SeqDev *seqs, int cap) {
    const int seq = blockIdx.z, f = blockIdx.x * 256 + threadIdx.x;
    int hit = 0;
    if (f < kn_new[seq]) {
        const int i = win[(size_t)seq * cap + f];
        const KlSoA &o = kl_old[seq], &n = kl_new[seq];
        if (i >= 0 || FILL) {
            // every gather first, then every store: a load issued behind a store waits for it (vmcnt counts loads and stores
            // in issue order), so field-by-field copies pay one memory round trip per field
            double rho = kRhoInit, s_rho = kRhoMax, rho_nr = kRhoInit, s_rho_nr = kRhoMax;
            int32_t m_num = -1, m_id_kf = -1;
            float2 pm = make_float2(0.f, 0.f), mm = make_float2(0.f, 0.f);
            float nm = 0.f;
            if (i >= 0) {
                rho = o.rho[i]; s_rho = o.s_rho[i]; rho_nr = o.rho_nr[i]; s_rho_nr = o.s_rho_nr[i];
                m_num = o.m_num[i]; m_id_kf = o.m_id_kf[i];
                pm = o.p_m[i]; mm = o.m_m[i];
                nm = o.n_m[i];
            } else {
                pm = n.p_m[f];            // p_m_0 of a fresh KeyLine is its own p_m
            }
            n.rho[f] = rho;
            n.s_rho[f] = s_rho;
            n.rho_nr[f] = rho_nr;
            n.s_rho_nr[f] = s_rho_nr;
            n.m_num[f] = m_num + 1;       // 0 for a fresh KeyLine
            n.m_id[f] = i;                // -1 for a fresh KeyLine
            n.p_m_0[f] = pm;
            n.m_m0[f] = mm;
            n.n_m0[f] = (double)nm;
            n.m_id_kf[f] = m_id_kf;
            hit = i >= 0;
        }
    }
    const int cnt = __popcll(__ballot(hit));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&seqs[seq].pub.klm_fwd, cnt);
}

// ---------------------------------------------------------------------------------------------------
// rotate_keylines.  R given per sequence (row-major); when from_state != 0 each sequence uses
// R0 = exp(W) from its state and stores the back-rotation R = R0^T (rebvo_second_t.cpp:360-361).
// ---------------------------------------------------------------------------------------------------
__device__ inline void so3_exp_c(const double w[3], double R[9]);

// The turned values of a (slot, sequence) next to its own arrays (edgehip_ctx::rot_*; matching in one pass)
struct RotOut {
    float2 *p_m, *m_m;   // null: rotate in place
    double *rho, *s_rho;
};
// WIN (matching in one pass): this pass over the old KeyLines is also FordwardMatch's arbitration (k_fwd_win) and the per-sequence
// scalar work behind the minimiser that used to ride on that kernel — R0 = exp(W) and the NaN check of rebvo_second_t.cpp:387-397.
// Every block forms R0 for itself from the sequence's W (nothing in this kernel writes W); the sequence's first block also stores it
// (state.R, Rbuf) and runs the check.
struct WinArgs {
    const int32_t *kn_new;
    const unsigned long long *key;   // [B][CAP] arbitration keys the minimiser's last evaluation posted
    int32_t *win;                    // [B][CAP]
    SeqDev *seqs;
    double *Rbuf;
};
template <bool OUT, bool WIN>
__global__ __launch_bounds__(256) void k_rotate(const KlSoA *kls, const int32_t *__restrict__ kns, const double *__restrict__ Rin,
                                                double zf, RotOut out, int cap, WinArgs wa) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    __shared__ double s_R[9];
    if (WIN) {
        if (blockIdx.x * 256 >= kns[seq] && blockIdx.x != 0) return;   // block-uniform
        if (threadIdx.x == 0) {
            SeqDev *sq = wa.seqs + seq;
            double R0[9];
            so3_exp_c(sq->pub.W, R0);
            for (int q = 0; q < 9; q++) s_R[q] = R0[q];
            if (blockIdx.x == 0) {   // rot_from_state + glue_after_tracking
                for (int q = 0; q < 9; q++) wa.Rbuf[(size_t)seq * 9 + q] = R0[q];
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) sq->pub.R[a * 3 + b] = R0[b * 3 + a];
                glue_after_tracking(sq);
            }
        }
        __syncthreads();
    }
    if (i >= kns[seq]) return;
    double Rw[9];
    if (WIN) {   // block-uniform values: back into scalar registers
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const double v = s_R[q];
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
            Rw[q] = __hiloint2double((int)hi, (int)lo);
        }
    }
    const double *R = WIN ? Rw : Rin + (size_t)seq * 9;
    const KlSoA &k = kls[seq];
    const float2 pm = k.p_m[i];
    const float2 m = k.m_m[i];                       // all loads before the first store (a load behind a store waits for it)
    const double rho = k.rho[i], s_rho = k.s_rho[i];
    // (the gather record's copy of m_m is NOT kept up to date here — 20 B read and 32 B written per KeyLine for a field whose only
    // reader on the frame path, search_match, finds the turned gradient in m_m itself; edgehip_ctx::rec_stale, k_rec_refresh)
    const double v0 = (double)pm.x / zf, v1 = (double)pm.y / zf, v2 = 1;
    double q0 = 0, q1 = 0, q2 = 0;  // TooN matrix*vector: row dot products accumulated from 0
    q0 += R[0] * v0; q0 += R[1] * v1; q0 += R[2] * v2;
    q1 += R[3] * v0; q1 += R[4] * v1; q1 += R[5] * v2;
    q2 += R[6] * v0; q2 += R[7] * v1; q2 += R[8] * v2;
    const double m0 = (double)m.x, m1 = (double)m.y;
    double r0 = 0, r1 = 0;
    r0 += R[0] * m0; r0 += R[1] * m1; r0 += R[2] * 0.0;
    r1 += R[3] * m0; r1 += R[4] * m1; r1 += R[5] * 0.0;
    const float2 mr = make_float2((float)r0, (float)r1);
    if (WIN) {   // k_fwd_win's rule: among the old KeyLines that point at new KeyLine f the largest rho wins, ties the largest index
        const int f = k.m_id_f[i];
        if (f >= 0 && f < wa.kn_new[seq] && wa.key[(size_t)seq * cap + f] == ord_bits(rho)) atomicMax(&wa.win[(size_t)seq * cap + f], i);
    }
    if (OUT) {
        // every KeyLine's four values, turned or (q2 == 0: the reference leaves them) as they are: whole lines
        const size_t o = (size_t)seq * cap + i;
        const bool turn = fabs(q2) > 0;
#if EDGEHIP_NT_ROT
        st_stream(out.p_m + o, turn ? make_float2((float)(q0 / q2 * zf), (float)(q1 / q2 * zf)) : pm);
        st_stream(out.rho + o, turn ? rho / q2 : rho);
        st_stream(out.s_rho + o, turn ? s_rho / q2 : s_rho);
        st_stream(out.m_m + o, mr);
#else
        out.p_m[o] = turn ? make_float2((float)(q0 / q2 * zf), (float)(q1 / q2 * zf)) : pm;
        out.rho[o] = turn ? rho / q2 : rho;
        out.s_rho[o] = turn ? s_rho / q2 : s_rho;
        out.m_m[o] = mr;
#endif
        return;
    }
    if (fabs(q2) > 0) {
        k.p_m[i] = make_float2((float)(q0 / q2 * zf), (float)(q1 / q2 * zf));
        k.rho[i] = rho / q2;
        k.s_rho[i] = s_rho / q2;
    }
    k.m_m[i] = mr;
}

// rot_pending: the turned values into the slot's own arrays (what rotate_keylines in place would have left)
__global__ __launch_bounds__(256) void k_rot_materialize(const KlSoA *kls, const int32_t *__restrict__ kns, RotOut src, int cap) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kns[seq]) return;
    const KlSoA &k = kls[seq];
    const size_t o = (size_t)seq * cap + i;
    const float2 pm = src.p_m[o], mm = src.m_m[o];
    const double rho = src.rho[o], s_rho = src.s_rho[o];
    k.p_m[i] = pm; k.m_m[i] = mm; k.rho[i] = rho; k.s_rho[i] = s_rho;
}
static RotOut rot_of(edgehip_ctx *c, int slot) {
    const size_t off = (size_t)slot * c->plan.nseq * c->plan.cap;
    RotOut r;
    r.p_m = c->rot_pm + off; r.m_m = c->rot_mm + off; r.rho = c->rot_rho + off; r.s_rho = c->rot_srho + off;
    return r;
}
int rot_materialize_enqueue(edgehip_ctx *c, int slot) {
    if (slot < 0 || slot >= c->plan.nslots || !c->rot_pending[slot]) return 0;
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_rot_materialize, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, rot_of(c, slot), pl.cap);
    EH_LAUNCH_CHECK();
    c->rot_pending[slot] = false;
    return 0;
}

// KlSoA::rec after rotate_keylines: the record's m_m from the KeyLine's (whole records: full sectors)
__global__ __launch_bounds__(256) void k_rec_refresh(const KlSoA *kls, const int32_t *__restrict__ kns) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kns[seq]) return;
    const KlSoA &k = kls[seq];
    const float2 cp = k.c_p[i], um = k.u_m[i], m = k.m_m[i];
    const float nm = k.n_m[i];
    MatchRec rec;
    rec.c_px = cp.x; rec.c_py = cp.y; rec.u_mx = um.x; rec.u_my = um.y;
    rec.m_mx = m.x; rec.m_my = m.y; rec.n_m = nm; rec.pad = 0.f;
    k.rec[i] = rec;
}
int rec_refresh_enqueue(edgehip_ctx *c, int slot) {
    if (int e = rot_materialize_enqueue(c, slot)) return e;
    if (!c->rec_stale[slot]) return 0;
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_rec_refresh, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq);
    EH_LAUNCH_CHECK();
    c->rec_stale[slot] = false;
    return 0;
}

#ifdef EDGEHIP_EXPERIMENTS   // FordwardMatch + rotate_keylines in one scattering pass: EDGEHIP_FWD_MODE=2, measured slower than the one-pass matching
// FordwardMatch's copy (edge_tracker.cpp:396-432) and rotate_keylines (:42-76) in ONE pass over the OLD KeyLines.  The
// reference copies first (old values), then rotates the old list in place; a thread that owns old KeyLine i does both for
// its own KeyLine: if it is the winner of its target (win[f] == i, decided by k_fwd_win before this launch) it scatters its
// fields to the new KeyLine f, then it rotates itself — nobody else reads old[i] in this kernel, so there is no hazard, and
// every old field is read once instead of by a gather pass (k_fwd_apply: ten 64-byte granules per matched KeyLine) plus a
// streaming pass (k_rotate).  Same arithmetic, same order, same bits as the two kernels it replaces.
__global__ __launch_bounds__(256) void k_fwd_apply_rotate(const KlSoA *kl_old, const KlSoA *kl_new, const int32_t *__restrict__ kn_old,
                                                          const int32_t *__restrict__ kn_new, const int32_t *__restrict__ win,
                                                          SeqDev *seqs, const double *__restrict__ Rin, double zf, int cap) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    int hit = 0;
    if (i < kn_old[seq]) {
        const KlSoA &o = kl_old[seq];
        const int f = o.m_id_f[i];
        const float2 pm = o.p_m[i];
        const float2 m = o.m_m[i];
        const double rho = o.rho[i], s_rho = o.s_rho[i];
        if (f >= 0 && f < kn_new[seq] && win[(size_t)seq * cap + f] == i) {
            const KlSoA &n = kl_new[seq];
            const double rho_nr = o.rho_nr[i], s_rho_nr = o.s_rho_nr[i];
            const int32_t m_num = o.m_num[i], m_id_kf = o.m_id_kf[i];
            const float nm = o.n_m[i];
            n.rho[f] = rho;
            n.s_rho[f] = s_rho;
            n.rho_nr[f] = rho_nr;
            n.s_rho_nr[f] = s_rho_nr;
            n.m_num[f] = m_num + 1;
            n.m_id[f] = i;
            n.p_m_0[f] = pm;
            n.m_m0[f] = m;
            n.n_m0[f] = (double)nm;
            n.m_id_kf[f] = m_id_kf;
            hit = 1;
        }
        const double *R = Rin + (size_t)seq * 9;
        const double v0 = (double)pm.x / zf, v1 = (double)pm.y / zf, v2 = 1;
        double q0 = 0, q1 = 0, q2 = 0;  // TooN matrix*vector: row dot products accumulated from 0
        q0 += R[0] * v0; q0 += R[1] * v1; q0 += R[2] * v2;
        q1 += R[3] * v0; q1 += R[4] * v1; q1 += R[5] * v2;
        q2 += R[6] * v0; q2 += R[7] * v1; q2 += R[8] * v2;
        if (fabs(q2) > 0) {
            o.p_m[i] = make_float2((float)(q0 / q2 * zf), (float)(q1 / q2 * zf));
            o.rho[i] = rho / q2;
            o.s_rho[i] = s_rho / q2;
        }
        const double m0 = (double)m.x, m1 = (double)m.y;
        double r0 = 0, r1 = 0;
        r0 += R[0] * m0; r0 += R[1] * m1; r0 += R[2] * 0.0;
        r1 += R[3] * m0; r1 += R[4] * m1; r1 += R[5] * 0.0;
        const float2 mr = make_float2((float)r0, (float)r1);
        o.m_m[i] = mr;   // (the record's copy: edgehip_ctx::rec_stale)
    }
    const int cnt = __popcll(__ballot(hit));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&seqs[seq].pub.klm_fwd, cnt);
}
#endif   // EDGEHIP_EXPERIMENTS

// TooN SO3 exp / ln (so3.h:203-285, 288-334), device copies used by the frame glue
__device__ inline void so3_exp_c(const double w[3], double R[9]) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double theta = sqrt(theta_sq);
    double A, B;
    if (theta_sq < 1e-8) { A = 1.0 - one_6th * theta_sq; B = 0.5; }
    else if (theta_sq < 1e-6) { B = 0.5 - 0.25 * one_6th * theta_sq; A = 1.0 - theta_sq * one_6th * (1.0 - one_20th * theta_sq); }
    else { const double it = 1.0 / theta; A = sin(theta) * it; B = (1 - cos(theta)) * (it * it); }
    const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R[0] = 1.0 - B * (wy2 + wz2); R[4] = 1.0 - B * (wx2 + wz2); R[8] = 1.0 - B * (wx2 + wy2);
    double a = A * w[2], b = B * (w[0] * w[1]);
    R[1] = b - a; R[3] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R[2] = b + a; R[6] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R[5] = b - a; R[7] = b + a;
}

__device__ inline void so3_ln(const double M[9], double r[3]) {
    const double cos_angle = (M[0] + M[4] + M[8] - 1.0) * 0.5;
    r[0] = (M[7] - M[5]) / 2;
    r[1] = (M[2] - M[6]) / 2;
    r[2] = (M[3] - M[1]) / 2;
    const double sin_angle_abs = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (cos_angle > M_SQRT1_2) {
        if (sin_angle_abs > 0) {
            const double s = asin(sin_angle_abs) / sin_angle_abs;
            r[0] *= s; r[1] *= s; r[2] *= s;
        }
    } else if (cos_angle > -M_SQRT1_2) {
        const double s = acos(cos_angle) / sin_angle_abs;
        r[0] *= s; r[1] *= s; r[2] *= s;
    } else {
        const double angle = M_PI - asin(sin_angle_abs);
        const double d0 = M[0] - cos_angle, d1 = M[4] - cos_angle, d2 = M[8] - cos_angle;
        double r2[3];
        if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) { r2[0] = d0; r2[1] = (M[3] + M[1]) / 2; r2[2] = (M[2] + M[6]) / 2; }
        else if (d1 * d1 > d2 * d2) { r2[0] = (M[3] + M[1]) / 2; r2[1] = d1; r2[2] = (M[7] + M[5]) / 2; }
        else { r2[0] = (M[2] + M[6]) / 2; r2[1] = (M[7] + M[5]) / 2; r2[2] = d2; }
        if (r2[0] * r[0] + r2[1] * r[1] + r2[2] * r[2] < 0) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
        const double nn = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
        for (int i = 0; i < 3; i++) r[i] = angle * (r2[i] / nn);
    }
}

// R0 = exp(W); Rbuf[seq] = R0 (for k_rotate); state.R = R0^T * I^T ... i.e. R.T() = R0 * R.T() with R = I
__device__ inline void rot_from_state(SeqDev *sq, double *__restrict__ Rbuf_seq) {
    double R0[9];
    so3_exp_c(sq->pub.W, R0);
    for (int i = 0; i < 9; i++) Rbuf_seq[i] = R0[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) sq->pub.R[i * 3 + j] = R0[j * 3 + i];
}
__global__ void k_rot_from_state(SeqDev *seqs, double *__restrict__ Rbuf, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    rot_from_state(seqs + seq, Rbuf + (size_t)seq * 9);
}

// ---------------------------------------------------------------------------------------------------
// directed_matching + search_match: thread per NEW KeyLine, walking the OLD mask along the epipolar line
// ---------------------------------------------------------------------------------------------------
struct DirArgs {
    const KlSoA *kl_new, *kl_old;
    const int32_t *kn_new;
    const int32_t *mask_old;  // [B][N]
    SeqDev *seq;
    int w, h;
    size_t n;
    double zfm, min_thr_mod, cang_min_edge, max_radius, loc_unc;
    float ppx, ppy;
    int stereo_mode;          // REBVO/StereoAvaiable: clone rho0/s_rho0 instead of rho/s_rho (edge_tracker.cpp:343-351)
    // matching in one pass (FUSED instantiations; edgehip_ctx::fuse_match)
    const int32_t *win;       // [B][CAP] FordwardMatch's winner (old KeyLine) of every new KeyLine, -1 none
    RotOut rot;               // the old slot's turned p_m / m_m / rho / s_rho (its own arrays still hold the unturned ones)
    int cap;
};

// FUSED: FordwardMatch's copy (edge_tracker.cpp:380-436) and directed_matching in one visit of the new KeyLine.  The reference copies the
// forwarded fields first, turns the old KeyLines, and then lets directed_matching overwrite the same ten fields wherever it finds a
// match (~90 % of the KeyLines).  Here the search prior (rho, s_rho of the forward match) and, where the search fails, the other
// eight fields are read from the old KeyLine that won the forward arbitration — its arrays are still unturned, k_rotate<OUT> put the
// turned values the candidates are tested with into `rot` — and every new KeyLine's ten fields are written once: the match's, the
// forward match's, or (FILL: the detector left them to this kernel) a fresh KeyLine's.  Same values, same bits.  A sequence whose
// tracker returned NaN (skip_match) gets the forward copy alone, as the reference's FordwardMatch has already happened by then.
#ifndef EDGEHIP_NT_ROT
#define EDGEHIP_NT_ROT 1   // C.rotate 218 -> 210 us, the walk that gathers from these arrays unchanged
#endif
#ifndef EDGEHIP_NT_EKF
#define EDGEHIP_NT_EKF 1   // C.regularize_ekf 382 -> 367 us and C.rescale, which reads these arrays next, 202 -> 170 us
#endif
#ifndef EDGEHIP_NT_DIRECTED
#define EDGEHIP_NT_DIRECTED 1   // 1.1 GB per 1024 frames written once, read by the next kernels long after the caches have turned over: 1055 -> 1018 us
#endif
template <bool FUSED, bool FILL>
__device__ __forceinline__ void directed_body(const DirArgs &a) {
    const int seq = blockIdx.z, ik = blockIdx.x * (int)blockDim.x + threadIdx.x;
    SeqDev *sq = a.seq + seq;
    const bool searching = !sq->skip_match;   // block-uniform
    if (!FUSED && !searching) return;
    __shared__ double s_v[3], s_rv[9], s_br[9];
    if (threadIdx.x == 0 && searching) {
        // Vel = BackRot*Vel; RVel = BackRot*RVel*BackRot.T()  (edge_tracker.cpp:323-324)
        const double *BR = sq->pub.R, *V = sq->pub.V, *P = sq->pub.P_V;
        for (int i = 0; i < 9; i++) s_br[i] = BR[i];
        for (int i = 0; i < 3; i++) {
            double d = 0;
            for (int j = 0; j < 3; j++) d += BR[i * 3 + j] * V[j];
            s_v[i] = d;
        }
        double T[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double d = 0;
                for (int k = 0; k < 3; k++) d += BR[i * 3 + k] * P[k * 3 + j];
                T[i * 3 + j] = d;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double d = 0;
                for (int k = 0; k < 3; k++) d += T[i * 3 + k] * BR[j * 3 + k];
                s_rv[i * 3 + j] = d;
            }
    }
    __syncthreads();
    int matched = 0, kfm = 0, fwd = 0;
    if (ik < a.kn_new[seq]) {
        const KlSoA &kn = a.kl_new[seq], &ko = a.kl_old[seq];
        const int32_t *mask = a.mask_old + (size_t)seq * a.n;
        const size_t ro = (size_t)seq * a.cap;   // FUSED: this sequence's turned values
        const float2 kpm = ldg(kn.p_m, ik);
        const float2 kmm = ldg(kn.m_m, ik);
        const float knm = ldg(kn.n_m, ik);
        int iw = -1;
        double krho, ksrho;
        if (FUSED) {
            iw = a.win[ro + ik];
            fwd = iw >= 0;
            if (iw >= 0) { krho = ldg(ko.rho, iw); ksrho = ldg(ko.s_rho, iw); }                 // what FordwardMatch copies: the unturned values
            else if (FILL) { krho = kRhoInit; ksrho = kRhoMax; }                      // a fresh KeyLine (edge_finder.cpp:178-196)
            else { krho = ldg(kn.rho, ik); ksrho = ldg(kn.s_rho, ik); }                         // ... which its detector wrote itself
        } else {
            krho = ldg(kn.rho, ik); ksrho = ldg(kn.s_rho, ik);
        }
        int found = -1;
        if (searching) {
        const double zf = a.zfm;
        double p3[3];
        for (int i = 0; i < 3; i++) {
            double d = 0;
            d += s_br[i * 3 + 0] * (double)kpm.x;
            d += s_br[i * 3 + 1] * (double)kpm.y;
            d += s_br[i * 3 + 2] * zf;
            p3[i] = d;
        }
        // (three quotients by p3[2] — a depth in units of the focal length, ~ zf — and two by norm_t below: the divisor's reciprocal once each,
        // MidDivisor in ctx.h: the division sequence itself without its exponent scaling, the same bits)
        const MidDivisor by_z(p3[2]);
        const float pmx = (float)by_z(p3[0] * zf);
        const float pmy = (float)by_z(p3[1] * zf);
        const double k_rho = by_z(krho * zf);
        const float pi0x = pmx + a.ppx, pi0y = pmy + a.ppy;
        double t_x = -(s_v[0] * zf - s_v[2] * (double)pmx);
        double t_y = -(s_v[1] * zf - s_v[2] * (double)pmy);
        double norm_t = sqrt(t_x * t_x + t_y * t_y);
        const double drdv[3] = {zf, zf, (double)(-pmx - pmy)};
        double row[3];
        for (int j = 0; j < 3; j++) {
            double d = 0;
            for (int i = 0; i < 3; i++) d += drdv[i] * s_rv[i * 3 + j];
            row[j] = d;
        }
        double sigma2_t = 0;
        for (int j = 0; j < 3; j++) sigma2_t += row[j] * drdv[j];

        double dq_min, dq_max, dq_rho;
        int t_steps;
        if (norm_t > 1e-6) {
            const MidDivisor by_n(norm_t);
            t_x = by_n(t_x);
            t_y = by_n(t_y);
            dq_rho = norm_t * k_rho;
            dq_min = fmax(0.0, norm_t * (k_rho - ksrho)) - a.loc_unc;
            dq_max = fmin(a.max_radius, norm_t * (k_rho + ksrho)) + a.loc_unc;
            if (dq_rho > dq_max) {
                dq_rho = (dq_max + dq_min) / 2;
                t_steps = x86_cvttsd2si(dq_rho + 0.5);       // util::round2int_positive on x86-64 (see ctx.h)
            } else {
                t_steps = x86_cvttsd2si(fmax(dq_max - dq_rho, dq_rho - dq_min) + 0.5);
            }
        } else {
            t_x = (double)kmm.x;
            t_y = (double)kmm.y;
            norm_t = (double)knm;
            const MidDivisor by_n(norm_t);
            t_x = by_n(t_x);
            t_y = by_n(t_y);
            norm_t = 1;
            dq_min = -a.max_radius - a.loc_unc;
            dq_max = a.max_radius + a.loc_unc;
            dq_rho = 0;
            t_steps = (int)dq_max;
        }
        const double norm_m = (double)knm;
        // search_match walks t_i = 0,1,2,... probing first tn = dq_rho - t_i, then tp = dq_rho + 1 + t_i, and stops at
        // the first candidate that passes the tests (edge_tracker.cpp:239-292).  The probe POSITIONS do not depend
        // on earlier probes, so the mask reads of DM_CH steps (2*DM_CH gathers) are issued together and only then
        // examined in the reference's order: the dependent-latency chain shrinks DM_CH-fold, the result is identical.
        constexpr int DM_CH = 2;   // measured in the one-pass form, us per 1024 frames (same box): 1 step per trip 1105, 2: 1067, 3: 1163, 4: 1388
        // A step probes the old mask only if its position lies inside the image, i.e. |t| <= Tmax (the direction is a unit
        // vector).  With a sane velocity estimate every step qualifies.  With a diverged one norm_t * rho reaches 1e8 and
        // beyond and the reference walks through all of it: tn comes down from dq_rho to dq_min one pixel at a time, millions
        // of steps that fall outside the image, before (and after) the few hundred that can see the mask — the kernel once
        // took 50 s for one frame.  Only the steps that can see the mask are executed here: at most two runs of step
        // indices (one where tn is within Tmax of the image, one where tp is), each entered with tn / tp computed directly.
        // That is the value the reference's repeated -= 1 / += 1 arrives at: moving towards zero in unit steps is exact in
        // fp64 (the grid only gets finer), and a counter that moves away from zero is beyond Tmax for good.
        const double Tmax = (double)(a.w + a.h) + fabs((double)pi0x) + fabs((double)pi0y) + 4.0;
        int seg0[2] = {0, 0}, seg1[2] = {t_steps, 0};   // [seg0[k], seg1[k]) step-index runs; default: everything
        int nseg = 1;
        if (t_steps > 256 && fabs(dq_rho) < 1e15 && Tmax < 1e15) {
            // steps at which tn = dq_rho - i (>= dq_min) resp. tp = dq_rho + 1 + i (<= dq_max) is within Tmax of zero
            const double n0 = fmax(0.0, floor(dq_rho - Tmax) - 2.0), n1 = fmin((double)t_steps, ceil(fmin(dq_rho + Tmax, dq_rho - dq_min)) + 2.0);
            const double p0 = fmax(0.0, floor(-Tmax - dq_rho - 1.0) - 2.0), p1 = fmin((double)t_steps, ceil(fmin(Tmax - dq_rho - 1.0, dq_max - dq_rho - 1.0)) + 2.0);
            const bool hn = n1 > n0, hp = p1 > p0;
            nseg = 0;
            if (hn && hp && !(n1 < p0 || p1 < n0)) {   // overlapping: one run
                seg0[0] = (int)fmin(n0, p0); seg1[0] = (int)fmax(n1, p1); nseg = 1;
            } else {
                if (hn) { seg0[nseg] = (int)n0; seg1[nseg] = (int)n1; nseg++; }
                if (hp) { seg0[nseg] = (int)p0; seg1[nseg] = (int)p1; nseg++; }
                if (nseg == 2 && seg0[1] < seg0[0]) {
                    const int s0 = seg0[0], s1 = seg1[0];
                    seg0[0] = seg0[1]; seg1[0] = seg1[1]; seg0[1] = s0; seg1[1] = s1;
                }
            }
        }
        for (int sg = 0; sg < nseg && found < 0; sg++) {
        // counters as the reference's repeated -= 1 / += 1 leave them at step seg0[sg]
        double tn = dq_rho - (double)seg0[sg], tp = (dq_rho + 1) + (double)seg0[sg];
        const int t_end = seg1[sg];
        for (int t0i = seg0[sg]; t0i < t_end && found < 0; t0i += DM_CH) {
            double tv[DM_CH][2];
            int jm[DM_CH][2];
#pragma unroll
            for (int c = 0; c < DM_CH; c++) {
                tv[c][0] = tn; tv[c][1] = tp;
                tn -= 1; tp += 1;
#pragma unroll
                for (int dir = 0; dir < 2; dir++) {
                    const double t = tv[c][dir];
                    jm[c][dir] = -1;
                    if (t0i + c >= t_end) continue;
                    if (dir ? t > dq_max : t < dq_min) continue;
                    const float fx = (float)(t_x * t + (double)pi0x), fy = (float)(t_y * t + (double)pi0y);
                    const int xi = round_half_away_i(fx), yi = round_half_away_i(fy);
                    if (xi >= a.w || yi >= a.h || xi < 0 || yi < 0) continue;
                    jm[c][dir] = mask[__umul24((unsigned)yi, (unsigned)a.w) + (unsigned)xi];   // (a 24-bit product: the 64-bit one is two quarter-rate multiplies per probe)
                }
            }
#pragma unroll
            for (int c = 0; c < DM_CH; c++) {
#pragma unroll
                for (int dir = 0; dir < 2; dir++) {
                    const int j = jm[c][dir];
                    if (j < 0 || found >= 0) continue;
                    const double t = tv[c][dir];
                    // the old KeyLine's (turned) gradient from m_m / n_m themselves: neighbouring threads test neighbouring old KeyLines,
                    // so the 8- and 4-byte gathers share their 64-byte lines at least as well as the 32-byte records did, and
                    // rotate_keylines no longer has to rewrite a record per KeyLine
                    const float2 omm = FUSED ? ldg(a.rot.m_m, ro + j) : ldg(ko.m_m, j);
                    const double norm_m0 = (double)ldg(ko.n_m, j);
                    const double cang = div_mid((double)(omm.x * kmm.x + omm.y * kmm.y), norm_m0 * norm_m);
                    if (cang < a.cang_min_edge || fabs(norm_m0 / norm_m - 1) > a.min_thr_mod) continue;
                    const double s_rho = FUSED ? ldg(a.rot.s_rho, ro + j) : ldg(ko.s_rho, j), rho = FUSED ? ldg(a.rot.rho, ro + j) : ldg(ko.rho, j);
                    const double v_rho_dr = (a.loc_unc * a.loc_unc + s_rho * s_rho * norm_t * norm_t + sigma2_t * rho * rho);
                    const double dd = t - norm_t * rho;
                    if (dd * dd > v_rho_dr) continue;
                    found = j;
                }
            }
        }
        }
        }   // searching
        if (FUSED) {
            // the ten fields of this KeyLine, written once: every gather first, then every store
            const int src = found >= 0 ? found : iw;
            if (src >= 0 || FILL) {
                double rho = kRhoInit, s_rho = kRhoMax, rho_nr = kRhoInit, s_rho_nr = kRhoMax;
                int32_t m_num = -1, m_id_kf = -1;
                float2 pm, mm = make_float2(0.f, 0.f);
                float nm = 0.f;
                if (src < 0) {
                    pm = ldg(kn.p_m, ik);                             // p_m_0 of a fresh KeyLine is its own p_m (re-read: not held across the walk)
                } else {
                    const bool turned = found >= 0;              // a directed match clones the turned old KeyLine (edge_tracker.cpp:343-366)
                    rho = turned ? ldg(a.rot.rho, ro + src) : ldg(ko.rho, src);
                    s_rho = turned ? ldg(a.rot.s_rho, ro + src) : ldg(ko.s_rho, src);
                    pm = turned ? ldg(a.rot.p_m, ro + src) : ldg(ko.p_m, src);
                    mm = turned ? ldg(a.rot.m_m, ro + src) : ldg(ko.m_m, src);
                    rho_nr = ldg(ko.rho_nr, src); s_rho_nr = ldg(ko.s_rho_nr, src);
                    m_num = ldg(ko.m_num, src); m_id_kf = ldg(ko.m_id_kf, src);
                    nm = ldg(ko.n_m, src);
                }
#if EDGEHIP_NT_DIRECTED
                stg_stream(kn.rho, ik, rho);
                stg_stream(kn.s_rho, ik, s_rho);
                stg_stream(kn.rho_nr, ik, rho_nr);
                stg_stream(kn.s_rho_nr, ik, s_rho_nr);
                stg_stream(kn.m_num, ik, m_num + 1);
                stg_stream(kn.m_id, ik, src);
                stg_stream(kn.p_m_0, ik, pm);
                stg_stream(kn.m_m0, ik, mm);
                stg_stream(kn.n_m0, ik, (double)nm);
                stg_stream(kn.m_id_kf, ik, m_id_kf);
#else
                stg(kn.rho, ik, rho);
                stg(kn.s_rho, ik, s_rho);
                stg(kn.rho_nr, ik, rho_nr);
                stg(kn.s_rho_nr, ik, s_rho_nr);
                stg(kn.m_num, ik, m_num + 1);
                stg(kn.m_id, ik, src);
                stg(kn.p_m_0, ik, pm);
                stg(kn.m_m0, ik, mm);
                stg(kn.n_m0, ik, (double)nm);
                stg(kn.m_id_kf, ik, m_id_kf);
#endif
                matched = found >= 0;
                kfm = matched && m_id_kf >= 0;
            }
        } else if (found >= 0) {
            const int j = found;
            // gathers first, stores after (a load behind a store waits for the store)
            const double c_rho = a.stereo_mode ? ldg(ko.rho0, j) : ldg(ko.rho, j), c_srho = a.stereo_mode ? ldg(ko.s_rho0, j) : ldg(ko.s_rho, j);
            double c_rho_nr = 0, c_srho_nr = 0;
            if (!a.stereo_mode) { c_rho_nr = ldg(ko.rho_nr, j); c_srho_nr = ldg(ko.s_rho_nr, j); }
            const int32_t c_mnum = ldg(ko.m_num, j), mk = ldg(ko.m_id_kf, j);
            const float2 c_pm = ldg(ko.p_m, j), c_mm = ldg(ko.m_m, j);
            const float c_nm = ldg(ko.n_m, j);
            stg(kn.rho, ik, c_rho);
            stg(kn.s_rho, ik, c_srho);
            if (!a.stereo_mode) {
                stg(kn.rho_nr, ik, c_rho_nr);
                stg(kn.s_rho_nr, ik, c_srho_nr);
            }
            stg(kn.m_id, ik, j);
            stg(kn.m_num, ik, c_mnum + 1);
            stg(kn.p_m_0, ik, c_pm);
            stg(kn.m_m0, ik, c_mm);
            stg(kn.n_m0, ik, (double)c_nm);
            stg(kn.m_id_kf, ik, mk);
            matched = 1;
            kfm = mk >= 0;
        }
    }
    const int c1 = __popcll(__ballot(matched)), c2 = __popcll(__ballot(kfm)), c3 = FUSED ? __popcll(__ballot(fwd)) : 0;
    if ((threadIdx.x & 63) == 0) {
        if (c1) atomicAdd(&sq->pub.klm_num, c1);
        if (c2) atomicAdd(&sq->pub.kf_matchs, c2);
        if (c3) atomicAdd(&sq->pub.klm_fwd, c3);
    }
}
__global__ __launch_bounds__(256) void k_directed(DirArgs a) { directed_body<false, false>(a); }
// The one-pass form holds two registers more than fit 7 waves per SIMD (74); compiled for 7 its small per-thread arrays move to LDS
// (2 KB per block) and the walk gains what an eighth more waves in flight hide: 1148 -> 1066 us per 1024 frames, same box
// (profiles/r04_z_matching_in_one_pass_ab.txt).
template <bool FILL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_directed_fused(DirArgs a) { directed_body<true, FILL>(a); }


// ---------------------------------------------------------------------------------------------------
// Regularize_1_iter -> scratch (r, s), then the EKF consumes the scratch
// ---------------------------------------------------------------------------------------------------
__device__ inline void glue_after_matching(SeqDev *sq);
// match_threshold >= 0 (whole-frame driver): the check that follows directed_matching (rebvo_second_t.cpp:412-422, too few matches ->
// no mapping this frame) is made here instead of in a launch of its own.  Every thread evaluates the condition from fields nothing
// in this kernel writes; the sequence's first thread applies its consequences to the state.
__global__ __launch_bounds__(256) void k_regularize(const KlSoA *kls, const int32_t *__restrict__ kns, double *__restrict__ rs,
                                                    SeqDev *seqs, int cap, double thresh, int enabled, int match_threshold) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    SeqDev *sq = seqs + seq;
    bool skip = sq->skip_map != 0;
    if (match_threshold >= 0 && !sq->skip_match && sq->pub.klm_num < match_threshold) {
        if (i == 0) glue_after_matching(sq);
        skip = true;
    }
    if (skip) return;
    if (i >= kns[seq]) return;
    const KlSoA &K = kls[seq];
    double r = ldg(K.rho, i), s = ldg(K.s_rho, i);
    const int ni = ldg(K.n_id, i), pi = ldg(K.p_id, i);
    if (enabled && ni >= 0 && pi >= 0) {
        const double nrho = ldg(K.rho, ni), prho = ldg(K.rho, pi), nsr = ldg(K.s_rho, ni), psr = ldg(K.s_rho, pi);
        const double d = nrho - prho;
        if (!(d * d > nsr * nsr + psr * psr)) {
            const float2 nm = ldg(K.m_m, ni), pm = ldg(K.m_m, pi);
            // all-float expression converted to double afterwards (edge_tracker.cpp:119)
            double alpha = (double)((nm.x * pm.x + nm.y * pm.y) / (ldg(K.n_m, ni) * ldg(K.n_m, pi)));
            if (!(alpha - thresh < 0)) {
                alpha = (alpha - thresh) / (1 - thresh);
                alpha /= fabs(nrho - prho) / (nsr + psr) + 1;
                const double wr = 1 / (s * s), wrn = alpha / (nsr * nsr), wrp = alpha / (psr * psr);
                const double r2 = (r * wr + nrho * wrn + prho * wrp) / (wr + wrn + wrp);
                const double s2 = (s * wr + nsr * wrn + psr * wrp) / (wr + wrn + wrp);
                r = r2;
                s = s2;
            }
        }
    }
    rs[(size_t)seq * 2 * cap + i] = r;     // (plain stores: the EKF reads them back at once; as streaming stores nothing moved)
    rs[(size_t)seq * 2 * cap + cap + i] = s;
}

__global__ __launch_bounds__(256) void k_ekf(const KlSoA *kls, const int32_t *__restrict__ kns, const double *__restrict__ rs,
                                             const SeqDev *seqs, int cap, double zf, double q_abs, double loc_unc, int do_ekf) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    const SeqDev *sq = seqs + seq;
    if (sq->skip_map) return;
    if (i >= kns[seq]) return;
    const KlSoA &K = kls[seq];
    double rho = rs[(size_t)seq * 2 * cap + i], s_rho = rs[(size_t)seq * 2 * cap + cap + i];
    if (do_ekf && ldg(K.m_id, i) >= 0) {
        // UpdateInverseDepthKalmanARLU, edge_tracker.cpp:954-1055
        const double v0 = sq->pub.V[0], v1 = sq->pub.V[1], v2 = sq->pub.V[2];
        const double s_rho0 = s_rho;
        const float2 q = ldg(K.p_m, i), q0 = ldg(K.p_m_0, i), mm0 = ldg(K.m_m0, i);
        const double qx = q.x, qy = q.y, q0x = q0.x, q0y = q0.y;
        double v_rho = s_rho * s_rho;
        const double nm0 = ldg(K.n_m0, i);
        const double u_x = (double)mm0.x / nm0, u_y = (double)mm0.y / nm0;
        const double Y = u_x * (qx - q0x) + u_y * (qy - q0y);
        const double H = u_x * (v0 * zf - v2 * q0x) + u_y * (v1 * zf - v2 * q0y);
        const double rho_p = 1 / (1.0 / rho + v2);
        const double rho0 = rho_p;
        double F = 1 / (1 + rho * v2);
        F = F * F;
        const double p_p = F * v_rho * F + q_abs * q_abs;
        const double e = Y - H * rho_p;
        const double S = H * p_p * H + loc_unc * loc_unc;
        const double Kg = p_p * H * (1 / S);
        rho = rho_p + (Kg * e);
        v_rho = (1 - Kg * H) * p_p;
        s_rho = sqrt(v_rho);
        if (rho < kRhoMin) {
            s_rho += kRhoMin - rho;
            rho = kRhoMin;
        } else if (rho > kRhoMax) {
            rho = kRhoMax;
        } else if (isnan(rho) || isnan(s_rho) || isinf(rho) || isinf(s_rho)) {
            rho = kRhoInit;
            s_rho = kRhoMax;
        } else if (s_rho < 0) {
            rho = kRhoInit;
            s_rho = kRhoMax;
        }
#if EDGEHIP_NT_EKF
        stg_stream(K.rho0, i, rho0);
        stg_stream(K.s_rho0, i, s_rho0);
    }
    stg_stream(K.rho, i, rho);
    stg_stream(K.s_rho, i, s_rho);
#else
        stg(K.rho0, i, rho0);
        stg(K.s_rho0, i, s_rho0);
    }
    stg(K.rho, i, rho);
    stg(K.s_rho, i, s_rho);
#endif
}

// ---------------------------------------------------------------------------------------------------
// EstimateReScalingOpt: 5 dependent weighted sums; one block per sequence
// ---------------------------------------------------------------------------------------------------
// 1 / x for a positive normal x: v_rcp_f64 + two Newton steps (relative error of a few 1e-16)
__device__ __forceinline__ double recip_f64(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}

// what the frame's last per-sequence step needs (frame_end below: pose integration + nav record)
struct FrameEndArgs {
    edgehip_nav *nav;            // [B] (null: no frame end in this launch)
    const int32_t *kn_new;
    const double *tresh_new;
    const float *retuned_new;
    edgehip_nav *nav_log;
    int nav_log_len, nseq, have_pair;
};
__device__ inline void frame_end(SeqDev *sq, const int seq, const FrameEndArgs &fe);

// The sums are those of 1024 "virtual" threads (KeyLine i belongs to virtual thread i % 1024, its terms are added in increasing i,
// the 16 wave sums in wave order), whatever the block: NT = 1024 real threads with PER KeyLines each in registers over the five
// passes, or NT = 512 threads that carry two virtual threads each (the form in use: PER = 12, 256 VGPRs per thread hold 12288
// KeyLines, so no pass waits for loads).  Same additions in the same order, same bits.
// fe.nav != null (whole-frame driver, mono, ImuMode 0): the block's first thread goes on to the frame's last step, pose integration
// and the nav record, which reads the Kp it has just written — the frame ends with this launch.
// LJ: the next LJ KeyLines of every virtual thread (KeyLines PER*1024 ... (PER+LJ)*1024-1) keep their four constants in LDS
// (dynamic, LJ * 32 KB) instead of streaming from memory in each of the five passes: with one block per CU and nothing else in
// flight (a few sequences) every streamed pass is a memory round trip per KeyLine of the tail.
template <int NT, int PER, int LJ>
__global__ __launch_bounds__(NT) void k_rescale(const KlSoA *kls, const int32_t *__restrict__ kns, SeqDev *seqs,
                                                double s_rho_min, unsigned match_num_min, int re_escale, FrameEndArgs fe) {
    constexpr int VT = 1024 / NT;
    extern __shared__ double s_tail[];   // [4][LJ][1024]: r2, r02, s2, s02
    const int seq = blockIdx.x, tid = threadIdx.x;
    SeqDev *sq = seqs + seq;
    const int kn = kns[seq];
    bool run = !sq->skip_map;   // block-uniform
    if (run && kn <= 0) { if (tid == 0) sq->pub.Kp = 1; run = false; }
    if (run) {
    const KlSoA &K = kls[seq];
    __shared__ double s_a[16], s_b[16];
    __shared__ double s_kp;
    // per-KeyLine constants of the five passes are read once: rho^2, rho0^2, s_rho^2, s_rho0^2
    double r2[VT][PER], r02[VT][PER], s2[VT][PER], s02[VT][PER];
    // (six KeyLines' five fields in flight together, unconditionally: a load inside a branch cannot be hoisted over the
    // one before it, and this kernel has nothing but its own loads to wait for when the batch is small)
    constexpr int LB = 6;
    static_assert(PER % LB == 0, "whole load batches");
#pragma unroll
    for (int v = 0; v < VT; v++) {
#pragma unroll
        for (int j0 = 0; j0 < PER; j0 += LB) {
            double l_sr0[LB], l_sr[LB], l_rho[LB], l_rho0[LB];
            int32_t l_mn[LB];
#pragma unroll
            for (int q = 0; q < LB; q++) {
                const int i = min(tid + v * NT + (j0 + q) * 1024, kn - 1);
                l_sr0[q] = ldg(K.s_rho0, i); l_sr[q] = ldg(K.s_rho, i); l_mn[q] = ldg(K.m_num, i); l_rho[q] = ldg(K.rho, i); l_rho0[q] = ldg(K.rho0, i);
            }
#pragma unroll
            for (int q = 0; q < LB; q++) {
                const int j = j0 + q;
                const bool use = tid + v * NT + j * 1024 < kn &&
                                 !((unsigned)l_mn[q] < match_num_min || l_sr0[q] <= 0 || l_sr[q] > s_rho_min);
                r2[v][j] = use ? l_rho[q] * l_rho[q] : 0.0;
                r02[v][j] = use ? l_rho0[q] * l_rho0[q] : 0.0;
                s2[v][j] = use ? l_sr[q] * l_sr[q] : 1.0;
                s02[v][j] = use ? l_sr0[q] * l_sr0[q] : 0.0;
            }
        }
    }
    // the tail's constants into LDS (defaults for KeyLines that do not count, as in the registers)
    const int njt = LJ ? min(LJ, max(0, (kn - PER * 1024 + 1023) / 1024)) : 0;   // block-uniform
    if (LJ) {
#pragma unroll
        for (int v = 0; v < VT; v++) {
            double l_sr0[LJ ? LJ : 1], l_sr[LJ ? LJ : 1], l_rho[LJ ? LJ : 1], l_rho0[LJ ? LJ : 1];
            int32_t l_mn[LJ ? LJ : 1];
#pragma unroll
            for (int q = 0; q < LJ; q++) {
                const int i = min(tid + v * NT + (PER + q) * 1024, kn - 1);
                l_sr0[q] = ldg(K.s_rho0, i); l_sr[q] = ldg(K.s_rho, i); l_mn[q] = ldg(K.m_num, i); l_rho[q] = ldg(K.rho, i); l_rho0[q] = ldg(K.rho0, i);
            }
#pragma unroll
            for (int q = 0; q < LJ; q++) {
                const bool use = tid + v * NT + (PER + q) * 1024 < kn &&
                                 !((unsigned)l_mn[q] < match_num_min || l_sr0[q] <= 0 || l_sr[q] > s_rho_min);
                const int at = q * 1024 + tid + v * NT;
                s_tail[at] = use ? l_rho[q] * l_rho[q] : 0.0;
                s_tail[LJ * 1024 + at] = use ? l_rho0[q] * l_rho0[q] : 0.0;
                s_tail[2 * LJ * 1024 + at] = use ? l_sr[q] * l_sr[q] : 1.0;
                s_tail[3 * LJ * 1024 + at] = use ? l_sr0[q] * l_sr0[q] : 0.0;
            }
        }
        // (each thread reads back only what it wrote: no barrier needed)
    }
    double Kp = 1, RKp = sq->pub.P_Kp;
    for (int iter = 0; iter < 5; iter++) {
        const double kp2 = Kp * Kp;
#pragma unroll
        for (int v = 0; v < VT; v++) {
            double a = 0, b = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) {
                // one reciprocal for the two quotients: the hardware estimate + two Newton steps (an ulp or two per term off the
                // reference's rho^2 / den and rho0^2 / den, next to a summation order that already differs from its sequential
                // one: Kp agrees to 1e-10, tests/test_stage_c_gpu.py).  The five passes are all this kernel does with a CU's
                // fp64 pipe, and the IEEE division is 3x the instructions.
                const double inv = recip_f64(s2[v][j] + kp2 * s02[v][j]);
                a += r2[v][j] * inv;
                b += r02[v][j] * inv;
            }
            for (int q = 0; q < njt; q++) {
                const int at = q * 1024 + tid + v * NT;
                const double inv = recip_f64(s_tail[2 * LJ * 1024 + at] + kp2 * s_tail[3 * LJ * 1024 + at]);
                a += s_tail[at] * inv;
                b += s_tail[LJ * 1024 + at] * inv;
            }
            for (int i = tid + v * NT + (PER + LJ) * 1024; i < kn; i += 1024) {
                const double sr0 = ldg(K.s_rho0, i), sr = ldg(K.s_rho, i);
                if ((unsigned)ldg(K.m_num, i) < match_num_min || sr0 <= 0 || sr > s_rho_min) continue;
                const double s2t = sr * sr, s02t = sr0 * sr0;   // the same expressions as the register path: the result does not depend on PER
                const double inv = recip_f64(s2t + kp2 * s02t);
                const double rho = ldg(K.rho, i), rho0 = ldg(K.rho0, i);
                a += (rho * rho) * inv;
                b += (rho0 * rho0) * inv;
            }
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if ((tid & 63) == 0) { s_a[(tid + v * NT) >> 6] = a; s_b[(tid + v * NT) >> 6] = b; }
        }
        __syncthreads();
        if (tid == 0) {
            double ta = 0, tb = 0;
            for (int k = 0; k < 16; k++) { ta += s_a[k]; tb += s_b[k]; }
            s_kp = tb > 0 ? sqrt(ta / tb) : 1;
            s_a[0] = 1 / tb;
        }
        __syncthreads();
        Kp = s_kp;
        RKp = s_a[0];
        __syncthreads();
    }
    if (re_escale) {
        // loads of a batch before its stores (a load behind a store waits for it)
        constexpr int SB = 8;
        for (int i0 = tid; i0 < kn; i0 += SB * NT) {
            double a_[SB], b_[SB];
#pragma unroll
            for (int q = 0; q < SB; q++) {
                const int i = min(i0 + q * NT, kn - 1);
                a_[q] = ldg(K.rho, i); b_[q] = ldg(K.s_rho, i);
            }
#pragma unroll
            for (int q = 0; q < SB; q++) {
                const int i = i0 + q * NT;
                if (i < kn) { stg(K.rho, i, a_[q] / Kp); stg(K.s_rho, i, b_[q] / Kp); }
            }
        }
    }
    if (tid == 0) { sq->pub.Kp = Kp; sq->pub.P_Kp = RKp; }
    }  // run
    if (fe.nav && tid == 0) frame_end(sq, seq, fe);
}

// ---------------------------------------------------------------------------------------------------
// per-frame glue of SecondThread (one thread per sequence)
// ---------------------------------------------------------------------------------------------------
__device__ inline void ident_scaled(double *M, double s) {
    for (int i = 0; i < 9; i++) M[i] = 0;
    M[0] = M[4] = M[8] = s;
}
__device__ inline bool any_nan3(const double *v) { return isnan(v[0]) || isnan(v[1]) || isnan(v[2]); }

// what follows Minimizer_RV + FordwardMatch + rotate_keylines (rebvo_second_t.cpp:387-397): the tracker's own estimate is kept
// for the nav record, a NaN estimate switches matching and mapping off for the frame
__device__ inline void glue_after_tracking(SeqDev *sq) {
    edgehip_seq_state &p = sq->pub;
    for (int i = 0; i < 3; i++) { sq->V_track[i] = p.V[i]; sq->W_track[i] = p.W[i]; }
    for (int i = 0; i < 9; i++) { sq->PV_track[i] = p.P_V[i]; sq->PW_track[i] = p.P_W[i]; }
    if (any_nan3(p.V) || any_nan3(p.W)) {
        ident_scaled(p.P_V, 1e50);
        p.V[0] = p.V[1] = p.V[2] = 0;
        p.Kp = 1;
        p.P_Kp = 1e50;
        p.estimation_ok = 0;
        sq->skip_match = 1;
        sq->skip_map = 1;
    }
}

// too few matches after directed_matching (rebvo_second_t.cpp:412-422)
__device__ inline void glue_after_matching(SeqDev *sq) {
    edgehip_seq_state &p = sq->pub;
    ident_scaled(p.P_V, 1e50);
    p.V[0] = p.V[1] = p.V[2] = 0;
    p.Kp = 1;
    p.P_Kp = 10;
    p.estimation_ok = 0;
    sq->skip_map = 1;
}

// pose integration + nav record (rebvo_second_t.cpp:550-606); one thread per sequence
__device__ inline void frame_end(SeqDev *sq, const int seq, const FrameEndArgs &fe) {
    edgehip_seq_state &p = sq->pub;
    const int have_pair = fe.have_pair;
    edgehip_nav &o = fe.nav[seq];
    // Two phases, each of them loads first, arithmetic in registers, stores last: one thread, nothing to hide a memory round
    // trip behind, and a load issued behind a store waits for it (the field-by-field version paid eight round trips).  Two
    // phases rather than one so that the live values fit the 128 registers of a 1024-thread block.
    {   // ---- pose integration: Pose = Pose*R; Pos += -Pose*V*K; P_V /= dt^2 ----
        double Pose[9], R[9], V[3], Pos[3], P_V[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { Pose[i] = p.Pose[i]; R[i] = p.R[i]; P_V[i] = p.P_V[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) { V[i] = p.V[i]; Pos[i] = p.Pos[i]; }
        const double K = p.K, dt = p.dt;
        if (have_pair) {
            double P2[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    double d = 0;
#pragma unroll
                    for (int k = 0; k < 3; k++) d += Pose[i * 3 + k] * R[k * 3 + j];
                    P2[i * 3 + j] = d;
                }
#pragma unroll
            for (int i = 0; i < 9; i++) Pose[i] = P2[i];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                double d = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) d += (-Pose[i * 3 + k]) * V[k];
                Pos[i] += d * K;
            }
#pragma unroll
            for (int i = 0; i < 9; i++) P_V[i] /= dt * dt;
        }
        double RotLie[3], PoseLie[3];
        so3_ln(R, RotLie);
        so3_ln(Pose, PoseLie);
        if (have_pair) {
#pragma unroll
            for (int i = 0; i < 9; i++) { p.Pose[i] = Pose[i]; p.P_V[i] = P_V[i]; }
#pragma unroll
            for (int i = 0; i < 3; i++) p.Pos[i] = Pos[i];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) { o.Rot[i] = R[i]; o.Pose[i] = Pose[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) { o.RotLie[i] = RotLie[i]; o.PoseLie[i] = PoseLie[i]; o.Vel[i] = -V[i] * K / dt; o.Pos[i] = Pos[i]; }
        o.dt = dt;
    }
    int frame;
    {   // ---- the rest of the nav record ----
        double PVt[9], PWt[9], Vt[3], Wt[3];
#pragma unroll
        for (int i = 0; i < 9; i++) { PVt[i] = sq->PV_track[i]; PWt[i] = sq->PW_track[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) { Vt[i] = sq->V_track[i]; Wt[i] = sq->W_track[i]; }
        const double t_cur = sq->t_cur;
        const double Kp = p.Kp, P_Kp = p.P_Kp, s_rho_q = p.s_rho_q, score = p.score, rel_error = p.rel_error, rel_error_score = p.rel_error_score;
        const int klm_fwd = p.klm_fwd, klm_num = p.klm_num, kf_matchs = p.kf_matchs, est_ok = p.estimation_ok, evals = p.minimizer_evals;
        frame = p.frame;
        const double tresh = fe.tresh_new[seq];
        const float retuned = fe.retuned_new[seq];
        const int kn = fe.kn_new[seq];
        o.t = t_cur;
#pragma unroll
        for (int i = 0; i < 3; i++) { o.V[i] = Vt[i]; o.W[i] = Wt[i]; }
#pragma unroll
        for (int i = 0; i < 9; i++) { o.P_V[i] = PVt[i]; o.P_W[i] = PWt[i]; }
        o.Kp = Kp; o.RKp = P_Kp; o.s_rho_q = s_rho_q; o.tresh = tresh;
        o.score = score; o.rel_error = rel_error; o.rel_error_score = rel_error_score;
        o.retuned_thresh = retuned;
        o.kn = kn; o.klm_fwd = klm_fwd; o.klm_num = klm_num; o.kf_matchs = kf_matchs;
        o.estimation_ok = have_pair ? est_ok : 0;
        o.frame = frame; o.minimizer_evals = evals;
        p.t_prev = t_cur;
        p.frame = frame + 1;
    }
    if (fe.nav_log_len > 0) {   // the log's copy of the record, word by word
        static_assert(sizeof(edgehip_nav) % 8 == 0, "copied as 64-bit words");
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&o);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(fe.nav_log + (size_t)(frame % fe.nav_log_len) * fe.nseq + seq);
        constexpr int W8 = sizeof(edgehip_nav) / 8, CH = 17;
#pragma unroll 1
        for (int c0 = 0; c0 < W8; c0 += CH) {
            unsigned long long wbuf[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) wbuf[i] = src[min(c0 + i, W8 - 1)];
#pragma unroll
            for (int i = 0; i < CH; i++)
                if (c0 + i < W8) dst[c0 + i] = wbuf[i];
        }
    }
}

// mode 0: frame begin (:145-168); 1: after Minimizer+FordwardMatch+rotate (:387-397); 2: after
// directed_matching (:412-422); 3: pose integration + nav record (:550-606)
__global__ void k_frame_glue(SeqDev *seqs, const double *__restrict__ t_in, edgehip_nav *__restrict__ nav,
                             const int32_t *__restrict__ kn_new, const double *__restrict__ tresh_new,
                             const float *__restrict__ retuned_new, int nseq, int mode, double fps, int match_threshold,
                             int have_pair, edgehip_nav *__restrict__ nav_log, int nav_log_len, const int32_t *__restrict__ stereo_cnt,
                             int32_t *__restrict__ stereo_log) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    edgehip_seq_state &p = sq->pub;
    if (mode == 0) {
        frame_begin(sq, t_in[seq], fps);
    } else if (mode == 1) {
        glue_after_tracking(sq);
    } else if (mode == 2) {
        if (!sq->skip_match && p.klm_num < match_threshold) glue_after_matching(sq);
    } else {
        FrameEndArgs fe;
        fe.nav = nav; fe.kn_new = kn_new; fe.tresh_new = tresh_new; fe.retuned_new = retuned_new;
        fe.nav_log = nav_log; fe.nav_log_len = nav_log_len; fe.nseq = nseq; fe.have_pair = have_pair;
        // (with a stereo rig and a log: the frame's stereo_match_num beside its record, before frame_end moves p.frame on)
        if (stereo_log && nav_log_len > 0) stereo_log[(size_t)(p.frame % nav_log_len) * nseq + seq] = have_pair ? stereo_cnt[seq] : 0;
        frame_end(sq, seq, fe);
    }
}

// ---------------------------------------------------------------------------------------------------
// edge_tracker::ExtRotVel (edge_tracker.cpp:1207-1296): one row of the linear 6-DoF system per forward-matched
// KeyLine, normalised by its expected error and a Huber-like weight; 21 + 6 sums of Phi^T Phi, Phi^T Y (+ the row
// count) reduced with wave shuffles to one partial per block.  The 6x6 SVD solve stays with the caller.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ext_rotvel(const KlSoA *kls, const int32_t *__restrict__ kns, const double *__restrict__ vel,
                                                    const SeqDev *__restrict__ seqs, double *__restrict__ partials, int nblk, double zf,
                                                    double loc_unc, double hub) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the velocity Minimizer_V found: from the caller (stage entry point) or straight from the sequence state (frame driver)
    const double *vp = seqs ? seqs[seq].mv_V : vel + seq * 3;
    const double v0 = vp[0], v1 = vp[1], v2 = vp[2];
    double row[6] = {0, 0, 0, 0, 0, 0}, y = 0, used = 0;
    if (i < kns[seq] && kls[seq].m_id[i] >= 0) {
        const KlSoA &K = kls[seq];
        const float2 u = K.u_m[i], pm = K.p_m[i], pm0 = K.p_m_0[i];
        const double rho_t = 1 / (1 / K.rho[i] + v2);
        const float qt_x = (float)((double)pm0.x + rho_t * (v0 * zf - v2 * (double)pm0.x));
        const float qt_y = (float)((double)pm0.y + rho_t * (v1 * zf - v2 * (double)pm0.y));
        const double s_rho = K.s_rho[i];
        const float q_x = pm.x, q_y = pm.y;
        row[0] = (double)u.x * rho_t * zf;
        row[1] = (double)u.y * rho_t * zf;
        row[2] = (double)u.x * (-rho_t * (double)q_x) + (double)u.y * (-rho_t * (double)q_y);
        row[3] = (double)(-u.x * q_x * q_y) / zf - (double)u.y * (zf + (double)(q_y * q_y) / zf);
        row[4] = (double)(+u.y * q_x * q_y) / zf + (double)u.x * (zf + (double)(q_x * q_x) / zf);
        row[5] = (double)(-u.x * q_y + u.y * q_x);
        y = (double)(u.x * (pm.x - qt_x) + u.y * (pm.y - qt_y));
        const float dqvel = (float)((double)u.x * (v0 * zf - v2 * (double)pm0.x) + (double)u.y * (v1 * zf - v2 * (double)pm0.y));
        const float s_y = (float)sqrt(s_rho * s_rho * (double)dqvel * (double)dqvel + loc_unc * loc_unc);
        double weigth = 1;
        if (fabs(y) > hub) weigth = fabs(y) / hub;
        const double den = (double)s_y * weigth;
#pragma unroll
        for (int k = 0; k < 6; k++) row[k] /= den;
        y /= den;
        used = 1;
    }
    double sums[kNumSums];
    {
        int ns = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) sums[ns++] = row[a] * row[b];
#pragma unroll
        for (int a = 0; a < 6; a++) sums[ns++] = row[a] * y;
        sums[ns++] = used;
    }
    __shared__ double s_red[4][32];
    {
        const int idx = wave_reduce28(sums, lane);   // the xor butterfly's pairs and order, a sixth of its shuffles
        if ((lane & 1) == 0) s_red[wave][idx] = sums[0];
    }
    __syncthreads();
    if (threadIdx.x < kNumSums)
        partials[((size_t)seq * nblk + blockIdx.x) * kNumSums + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

// REBVO::Reset() as executed by SecondThread after a frame (rebvo_second_t.cpp:609-620)
__global__ __launch_bounds__(256) void k_depth_reset(const KlSoA *kls, const int32_t *__restrict__ kns, SeqDev *seqs, int only_seq,
                                                      int pose_too) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (only_seq >= 0 && seq != only_seq) return;
    if (i == 0) {
        edgehip_seq_state &p = seqs[seq].pub;
        if (pose_too) {   // ImuMode > 0: the pose belongs to the IMU stream (imu_pose_reset_enqueue)
            ident_scaled(p.Pose, 1);
            for (int k = 0; k < 3; k++) p.Pos[k] = 0;
        }
        for (int k = 0; k < 3; k++) { p.V[k] = 0; p.W[k] = 0; }
    }
    if (i >= kns[seq]) return;
    kls[seq].rho[i] = kRhoInit;
    kls[seq].s_rho[i] = kRhoMax;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// keys_posted: the minimiser's last evaluation already left the arbitration keys in fwd_key (TvrArgs::fwd_key)
int forward_match_enqueue(edgehip_ctx *c, int slot_old, int slot_new, bool keys_posted, bool frame_tail) {
    ProfScope ps(c, PROF_C_FORWARD);
    const DevicePlan &pl = c->plan;
    const size_t B = pl.nseq;
    const bool cleared = c->fwd_cleared;   // k_field_bin reset both arrays' entries of the new edge map (whole-frame driver)
    c->fwd_cleared = false;
    if (!keys_posted && !cleared) EH_CHECK(hipMemsetAsync(c->fwd_key, 0, sizeof(unsigned long long) * B * pl.cap, c->stream));
    if (!cleared) EH_CHECK(hipMemsetAsync(c->fwd_win, 0xFF, sizeof(int32_t) * B * pl.cap, c->stream));
    dim3 g((pl.cap + 255) / 256, 1, pl.nseq), b(256);
    const int32_t *kno = c->kn_slot + slot_old * B, *knn = c->kn_slot + slot_new * B;
    if (!keys_posted) hipLaunchKernelGGL(k_fwd_key, g, b, 0, c->stream, kldev(c, slot_old), kno, knn, c->fwd_key, pl.cap);
    hipLaunchKernelGGL(k_fwd_win, g, b, 0, c->stream, kldev(c, slot_old), kno, knn, c->fwd_key, c->fwd_win, pl.cap,
                       frame_tail ? c->seq : (SeqDev *)nullptr, frame_tail ? c->rot_buf : (double *)nullptr);
    if (c->fwd_fill[slot_new]) hipLaunchKernelGGL(k_fwd_apply<true>, g, b, 0, c->stream, kldev(c, slot_old), kldev(c, slot_new), knn, c->fwd_win, c->seq, pl.cap);
    else hipLaunchKernelGGL(k_fwd_apply<false>, g, b, 0, c->stream, kldev(c, slot_old), kldev(c, slot_new), knn, c->fwd_win, c->seq, pl.cap);
    c->fwd_fill[slot_new] = false;
    EH_LAUNCH_CHECK();
    return 0;
}

#ifdef EDGEHIP_EXPERIMENTS
// Whole-frame driver, ImuMode == 0 (rebvo_second_t.cpp:354-369): FordwardMatch + R0 = exp(W) + rotate_keylines(R0).  The
// arbitration keys were posted by the minimiser's last evaluation (TvrArgs::fwd_key; cleared in minimizer_enqueue).
int forward_rotate_enqueue(edgehip_ctx *c, int slot_old, int slot_new) {
    c->grec_ok[slot_old] = false;   // m_m turns, u_m does not
    c->rec_stale[slot_old] = true;
    const DevicePlan &pl = c->plan;
    const size_t B = pl.nseq;
    dim3 g((pl.cap + 255) / 256, 1, pl.nseq), b(256);
    const int32_t *kno = c->kn_slot + slot_old * B, *knn = c->kn_slot + slot_new * B;
    {
        ProfScope ps(c, PROF_C_FORWARD);
        if (!c->fwd_cleared) EH_CHECK(hipMemsetAsync(c->fwd_win, 0xFF, sizeof(int32_t) * B * pl.cap, c->stream));
        c->fwd_cleared = false;
        hipLaunchKernelGGL(k_fwd_win, g, b, 0, c->stream, kldev(c, slot_old), kno, knn, c->fwd_key, c->fwd_win, pl.cap,
                           (SeqDev *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL(k_rot_from_state, dim3((pl.nseq + 63) / 64), dim3(64), 0, c->stream, c->seq, c->rot_buf, pl.nseq);
        EH_LAUNCH_CHECK();
    }
    ProfScope ps(c, PROF_C_ROTATE);
    hipLaunchKernelGGL(k_fwd_apply_rotate, g, b, 0, c->stream, kldev(c, slot_old), kldev(c, slot_new), kno, knn, c->fwd_win, c->seq,
                       c->rot_buf, pl.zfm, pl.cap);
    EH_LAUNCH_CHECK();
    return 0;
}
#endif   // EDGEHIP_EXPERIMENTS

// Does the whole-frame driver (ImuMode 0) match this frame in one pass — rotate_keylines out of place, FordwardMatch's copy inside
// k_directed_fused?  ONE definition for frame_enqueue and for the graph replay's bookkeeping of rot_pending: the flag must say what
// the captured launches did.
static inline bool frame_matches_in_one_pass(const edgehip_ctx *c, int slot_pair) {
    return c->fuse_match && c->fwd_mode == 0 && slot_pair < 0 && !c->p.stereo_available;
}

// Matching in one pass, the old KeyLines' side: FordwardMatch's arbitration (keys posted by the minimiser's last evaluation), R0 = exp(W),
// the NaN check, and rotate_keylines(R0) out of place — one launch (k_rotate<OUT, WIN>)
static int forward_rotate_one_pass_enqueue(edgehip_ctx *c, int slot_old, int slot_new) {
    if (int e = rot_materialize_enqueue(c, slot_old)) return e;
    const DevicePlan &pl = c->plan;
    const size_t B = pl.nseq;
    const bool cleared = c->fwd_cleared;   // k_field_bin reset the arbitration entries of the new edge map
    c->fwd_cleared = false;
    if (!cleared) EH_CHECK(hipMemsetAsync(c->fwd_win, 0xFF, sizeof(int32_t) * B * pl.cap, c->stream));
    c->grec_ok[slot_old] = false;
    c->rec_stale[slot_old] = true;
    ProfScope ps(c, PROF_C_ROTATE);
    WinArgs wa;
    wa.kn_new = c->kn_slot + slot_new * B; wa.key = c->fwd_key; wa.win = c->fwd_win; wa.seqs = c->seq; wa.Rbuf = c->rot_buf;
    hipLaunchKernelGGL((k_rotate<true, true>), dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot_old),
                       c->kn_slot + slot_old * B, (const double *)nullptr, pl.zfm, rot_of(c, slot_old), pl.cap, wa);
    EH_LAUNCH_CHECK();
    c->rot_pending[slot_old] = true;
    return 0;
}

// R_in_buf: rot_buf already holds the rotations (k_fwd_win's frame tail)
int rotate_enqueue(edgehip_ctx *c, int slot, const double *R_host, bool R_in_buf) {
    if (int e = rot_materialize_enqueue(c, slot)) return e;   // (a second rotation of a slot whose first one is still pending)
    c->grec_ok[slot] = false;   // m_m turns, u_m does not (edge_tracker.cpp:42-76): u_m can no longer be recomputed from m_m
    c->rec_stale[slot] = true;
    ProfScope ps(c, PROF_C_ROTATE);
    const DevicePlan &pl = c->plan;
    double *Rbuf = c->rot_buf;
    if (R_host) {
        EH_CHECK(hipStreamSynchronize(c->stream));  // pinned_out is reused
        memcpy(c->pinned_out, R_host, sizeof(double) * 9 * pl.nseq);
        EH_CHECK(hipMemcpyAsync(Rbuf, c->pinned_out, sizeof(double) * 9 * pl.nseq, hipMemcpyHostToDevice, c->stream));
    } else if (!R_in_buf) {
        hipLaunchKernelGGL(k_rot_from_state, dim3((pl.nseq + 63) / 64), dim3(64), 0, c->stream, c->seq, Rbuf, pl.nseq);
    }
    hipLaunchKernelGGL((k_rotate<false, false>), dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, Rbuf, pl.zfm, RotOut{nullptr, nullptr, nullptr, nullptr}, pl.cap, WinArgs{});
    EH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Stereo depth (REBVO/StereoAvaiable): directed_matching_stereo / search_match_stereo / getDepthFromStereo
// (edge_tracker.cpp:453-668) and fuseStereoDepth (:670-688).  Thread per KeyLine of the main camera; the walk over the
// pair's edge mask is sequential per KeyLine because the "exactly one candidate" rule depends on the order.
// ---------------------------------------------------------------------------------------------------
struct StereoArgs {
    const KlSoA *kl, *kl_pair;
    const int32_t *kn;
    const int32_t *mask_pair;   // [B][N]
    int32_t *nmatch;            // [B]
    int w, h;
    size_t n;
    double t[3], R[9];
    double zfm0, zfm1;          // focal lengths of the main / pair camera
    float pp1x, pp1y;           // principal point of the pair camera
    double min_thr_mod, cang_min_edge, max_radius, loc_unc, loc_unc_model;
    const SeqDev *seqs;         // whole-frame driver: sequences whose mapping step is skipped this frame (else null)
};

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (a.seqs && a.seqs[seq].skip_map) return;
    int matched = 0;
    if (i < a.kn[seq]) {
        const KlSoA &k = a.kl[seq], &kp = a.kl_pair[seq];
        const int32_t *mask = a.mask_pair + (size_t)seq * a.n;
        const float2 pm = k.p_m[i], mm = k.m_m[i];
        const float nm = k.n_m[i];
        const double rho = k.rho[i], s_rho = k.s_rho[i];
        const double min_rho = fmax(rho - s_rho, 1e-3), max_rho = fmin(rho + s_rho, 20.0);   // RHO_MIN, RHO_MAX
        double q1[2][3];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double r = e ? max_rho : min_rho;
            // cam0.unprojectHomCordVec (cam_model.h:163-169), p1 = R p0 + t, cam1.projectHomCordVec (:146-152)
            const double p0[3] = {(double)pm.x / r / a.zfm0, (double)pm.y / r / a.zfm0, 1.0 / r};
            double p1[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double d = 0;
                d += a.R[c * 3 + 0] * p0[0]; d += a.R[c * 3 + 1] * p0[1]; d += a.R[c * 3 + 2] * p0[2];
                p1[c] = d + a.t[c];
            }
            q1[e][0] = p1[0] / p1[2] * a.zfm1; q1[e][1] = p1[1] / p1[2] * a.zfm1; q1[e][2] = 1 / p1[2];
        }
        const double dqx = q1[1][0] - q1[0][0], dqy = q1[1][1] - q1[0][1];
        const double pi0x = q1[0][0] + (double)a.pp1x, pi0y = q1[0][1] + (double)a.pp1y;   // Hom2Img<Point2D<double>>
        double norm_t = sqrt(dqx * dqx + dqy * dqy);
        double t_x, t_y, dq_min, dq_max;
        if (norm_t > 1e-6) {
            t_x = dqx / norm_t;
            t_y = dqy / norm_t;
            dq_min = -a.loc_unc;
            dq_max = fmin(a.max_radius, norm_t + a.loc_unc);
        } else {   // no displacement: search across the edge
            t_x = (double)mm.x;
            t_y = (double)mm.y;
            norm_t = (double)nm;
            t_x /= norm_t;
            t_y /= norm_t;
            norm_t = 1;
            dq_min = -a.max_radius / 2 - a.loc_unc;
            dq_max = a.max_radius / 2 + a.loc_unc;
        }
        const double norm_m = (double)nm;
        int match = -1;
        bool ambiguous = false;
        float2 pm_match = make_float2(0.f, 0.f);
        for (int t = (int)dq_min; (double)t < dq_max; t++) {
            const float fx = (float)(t_x * (double)(float)t + pi0x), fy = (float)(t_y * (double)(float)t + pi0y);
            const int xi = round_half_away_i(fx), yi = round_half_away_i(fy);
            if (xi >= a.w || yi >= a.h || xi < 0 || yi < 0) continue;
            const int j = mask[(size_t)yi * a.w + xi];
            if (j < 0) continue;
            const MatchRec r = kp.rec[j];
            const double norm_m0 = (double)r.n_m;
            const double cang = (double)(r.m_mx * mm.x + r.m_my * mm.y) / (norm_m0 * norm_m);
            if (cang < a.cang_min_edge || fabs(norm_m0 / norm_m - 1) > a.min_thr_mod) continue;
            const float2 pj = kp.p_m[j];
            if (match >= 0) {   // a second candidate: only acceptable next to the first one, else no match at all
                const float dx = pj.x - pm_match.x, dy = pj.y - pm_match.y;
                if ((double)(dx * dx + dy * dy) > a.loc_unc * a.loc_unc) { ambiguous = true; break; }
            }
            match = j;
            pm_match = pj;
        }
        if (ambiguous) {
            match = -1;            // search_match_stereo returns before touching stereo_rho / stereo_s_rho
        } else if (match >= 0) {   // getDepthFromStereo
            const double qh0[3] = {(double)pm.x / a.zfm0, (double)pm.y / a.zfm0, 1.0};
            double qh1[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double d = 0;
                d += a.R[c * 3 + 0] * qh0[0]; d += a.R[c * 3 + 1] * qh0[1]; d += a.R[c * 3 + 2] * qh0[2];
                qh1[c] = d;
            }
            const float2 u = kp.u_m[match];
            const double qx = (double)pm_match.x, qy = (double)pm_match.y, ux = (double)u.x, uy = (double)u.y, zf1 = a.zfm1;
            const double div = ux * (zf1 * a.t[0] - qx * a.t[2]) + uy * (zf1 * a.t[1] - qy * a.t[2]);
            const double mul = (double)(-u.x) * (zf1 * qh1[0] - qx * qh1[2]) - uy * (zf1 * qh1[1] - qy * qh1[2]);
            double srho = mul / div;
            const double den = qh1[2] + a.t[2] * srho;
            const double df = ux * zf1 * (a.t[0] * den - a.t[2] * (qh1[0] + a.t[0] * srho)) / (den * den) +
                              uy * zf1 * (a.t[1] * den - a.t[2] * (qh1[1] + a.t[1] * srho)) / (den * den);
            double I_rho = (df / a.loc_unc_model) * (df / a.loc_unc_model);
            if (srho != srho || df != df) { srho = 1; I_rho = 1e-10; }
            double ss = 1 / sqrt(I_rho);
            if (srho < 0) { ss = 1e3; srho = 1.0; match = -1; }   // RhoInit
            k.stereo_rho[i] = srho;
            k.stereo_s_rho[i] = ss;
        }
        k.stereo_m_id[i] = match;
        matched = match >= 0;
    }
    const int c1 = __popcll(__ballot(matched));
    if ((threadIdx.x & 63) == 0 && c1) atomicAdd(&a.nmatch[seq], c1);
}

__global__ __launch_bounds__(256) void k_fuse_stereo(const KlSoA *kls, const int32_t *__restrict__ kns, SeqDev *seqs) {
    const int seq = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (seqs) {   // whole-frame driver (rebvo_second_t.cpp:483-486): fuse, and Kp = 1 instead of the re-scaling estimate
        if (seqs[seq].skip_map) return;
        if (i == 0) seqs[seq].pub.Kp = 1;
    }
    if (i >= kns[seq]) return;
    const KlSoA &k = kls[seq];
    const double r0 = k.rho[i], s0 = k.s_rho[i];
    k.rho0[i] = r0;
    k.s_rho0[i] = s0;
    if (k.stereo_m_id[i] < 0) return;
    const double sr = k.stereo_rho[i], ss = k.stereo_s_rho[i];
    const double s = sqrt(1.0 / (1.0 / (s0 * s0) + 1.0 / (ss * ss)));
    k.s_rho[i] = s;
    k.rho[i] = (r0 / (s0 * s0) + sr / (ss * ss)) * (s * s);
}

// fused: matching in one pass — FordwardMatch's arbitration has run (fwd_win), rotate_keylines has written rot_of(slot_old)
int directed_enqueue(edgehip_ctx *c, int slot_new, int slot_old, bool fused) {
    ProfScope ps(c, PROF_C_DIRECTED);
    const DevicePlan &pl = c->plan;
    if (!fused) { if (int e = rot_materialize_enqueue(c, slot_old)) return e; }
    DirArgs a;
    a.kl_new = kldev(c, slot_new); a.kl_old = kldev(c, slot_old);
    a.kn_new = c->kn_slot + (size_t)slot_new * pl.nseq;
    a.mask_old = maskof(c, slot_old);
    a.seq = c->seq; a.w = pl.w; a.h = pl.h; a.n = pl.n; a.zfm = pl.zfm;
    a.min_thr_mod = c->p.match_thresh_module;
    a.cang_min_edge = cos(c->p.match_thresh_angle * M_PI / 180.0);
    a.max_radius = (double)c->p.search_range;
    a.loc_unc = c->p.loc_unc_match;
    a.ppx = pl.ppx; a.ppy = pl.ppy;
    a.stereo_mode = c->p.stereo_available != 0;
    a.win = c->fwd_win; a.cap = pl.cap;
    a.rot = fused ? rot_of(c, slot_old) : RotOut{nullptr, nullptr, nullptr, nullptr};
    // Block size of the walk (EDGEHIP_DIRECTED_BLOCK, a build option for A/Bs): a wave is as long as its longest lane's walk and a block
    // holds its wave slots until its last wave is done
#ifndef EDGEHIP_DIRECTED_BLOCK
#define EDGEHIP_DIRECTED_BLOCK 128
#endif
    constexpr int kDirBlock = EDGEHIP_DIRECTED_BLOCK;
    const dim3 g((pl.cap + kDirBlock - 1) / kDirBlock, 1, pl.nseq), b(kDirBlock);
    if (fused) {
        if (c->fwd_fill[slot_new]) hipLaunchKernelGGL(k_directed_fused<true>, g, b, 0, c->stream, a);
        else hipLaunchKernelGGL(k_directed_fused<false>, g, b, 0, c->stream, a);
        c->fwd_fill[slot_new] = false;
    } else {
        hipLaunchKernelGGL(k_directed, g, b, 0, c->stream, a);
    }
    EH_LAUNCH_CHECK();
    return 0;
}

int regekf_enqueue(edgehip_ctx *c, int slot, int do_reg, int do_ekf, bool frame_glue) {
    ProfScope ps(c, PROF_C_REGEKF);
    const DevicePlan &pl = c->plan;
    dim3 g((pl.cap + 255) / 256, 1, pl.nseq), b(256);
    const int32_t *kn = c->kn_slot + (size_t)slot * pl.nseq;
    hipLaunchKernelGGL(k_regularize, g, b, 0, c->stream, kldev(c, slot), kn, c->rs_tmp, c->seq, pl.cap,
                       c->p.regularize_thresh, do_reg, frame_glue ? c->p.global_match_threshold : -1);
    hipLaunchKernelGGL(k_ekf, g, b, 0, c->stream, kldev(c, slot), kn, c->rs_tmp, c->seq, pl.cap, pl.zfm, c->p.reshape_q_abs,
                       c->p.loc_unc, do_ekf);
    EH_LAUNCH_CHECK();
    return 0;
}

int rescale_enqueue(edgehip_ctx *c, int slot, bool frame_ends) {
    ProfScope ps(c, PROF_C_RESCALE);
    const DevicePlan &pl = c->plan;
    FrameEndArgs fe = {};
    if (frame_ends) {
        fe.nav = c->nav_dev;
        fe.kn_new = c->kn_slot + (size_t)slot * pl.nseq;
        // the detector's threshold state behind the frame's stage A: with a stereo rig the pair image is detected after the frame's
        // own, through the same controller (rebvo_first_t.cpp:275-290), so the state is the one the pair slot was detected with
        const int slot_t = c->rig.enabled && c->rig.slot_pair >= 0 ? c->rig.slot_pair : slot;
        fe.tresh_new = c->tresh_slot + (size_t)slot_t * pl.nseq;
        fe.retuned_new = c->retuned_slot + (size_t)slot * pl.nseq;
        fe.nav_log = c->nav_log; fe.nav_log_len = c->nav_log_len; fe.nseq = pl.nseq; fe.have_pair = 1;
    }
    // 512 threads with 12 + 12 KeyLines each in registers and the next 4096 in 128 KB of LDS, whatever the batch: for whole batches
    // the 1024-thread form (6 KeyLines per thread in registers, two blocks per CU, the rest streamed out of L2 five times) took 265 us
    // per 1024 sequences against 201.
    if (!c->lds_optin_rescale) {   // 128 KB of dynamic LDS: opt in once per context (= per device)
        EH_CHECK(hipFuncSetAttribute((const void *)&k_rescale<512, 12, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * 1024 * (int)sizeof(double)));
        c->lds_optin_rescale = true;
    }
    hipLaunchKernelGGL((k_rescale<512, 12, 4>), dim3(pl.nseq), dim3(512), 4 * 4 * 1024 * sizeof(double), c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, c->seq, kRhoMax, 1u, c->p.do_rescaling > 0, fe);
    EH_LAUNCH_CHECK();
    return 0;
}

// rotate_keylines with the rotations a device kernel left in rot_buf (the IMU branch of the frame driver)
static int rotate_buf_enqueue(edgehip_ctx *c, int slot) {
    if (int e = rot_materialize_enqueue(c, slot)) return e;
    c->grec_ok[slot] = false;
    c->rec_stale[slot] = true;
    ProfScope ps(c, PROF_C_ROTATE);
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL((k_rotate<false, false>), dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, c->rot_buf, pl.zfm, RotOut{nullptr, nullptr, nullptr, nullptr}, pl.cap, WinArgs{});
    EH_LAUNCH_CHECK();
    return 0;
}
// ExtRotVel's sums with the velocity taken from the sequence state; the solve is k_imu_mid's
static int ext_rotvel_enqueue(edgehip_ctx *c, int slot) {
    const DevicePlan &pl = c->plan;
    const int B = pl.nseq, nblk = (pl.cap + 255) / 256;
    if (nblk > c->nblk_tvr) { set_error("ext_rot_vel: partial table too small"); return EDGEHIP_ERR_STATE; }
    hipLaunchKernelGGL(k_ext_rotvel, dim3(nblk, 1, B), dim3(256), 0, c->stream, kldev(c, slot), c->kn_slot + (size_t)slot * B,
                       (const double *)nullptr, c->seq, c->partials, c->nblk_tvr, pl.zfm, c->p.loc_unc, c->p.reweight_distance);
    EH_LAUNCH_CHECK();
    return 0;
}

static int glue(edgehip_ctx *c, int mode, int slot_new, int have_pair) {
    const DevicePlan &pl = c->plan;
    const int slot_t = c->rig.enabled && c->rig.slot_pair >= 0 ? c->rig.slot_pair : slot_new;   // see rescale_enqueue
    hipLaunchKernelGGL(k_frame_glue, dim3((pl.nseq + 63) / 64), dim3(64), 0, c->stream, c->seq, c->t_src, c->nav_dev,
                       c->kn_slot + (size_t)slot_new * pl.nseq, c->tresh_slot + (size_t)slot_t * pl.nseq,
                       c->retuned_slot + (size_t)slot_new * pl.nseq, pl.nseq, mode, c->p.config_fps, c->p.global_match_threshold,
                       have_pair, c->nav_log, c->nav_log_len, c->stereo_cnt, c->rig.enabled ? c->stereo_log : nullptr);
    EH_LAUNCH_CHECK();
    return 0;
}

}  // namespace edgehip

using namespace edgehip;

extern "C" {

static int chk2(edgehip_ctx *c, int a, int b) {
    if (!c || a < 0 || b < 0 || a >= c->plan.nslots || b >= c->plan.nslots) { set_error("slot out of range"); return EDGEHIP_ERR_ARG; }
    // a slot the whole-frame driver rotated out of place: the stage-level calls work on the slot's own arrays
    if (int e = rot_materialize_enqueue(c, a)) return e;
    return rot_materialize_enqueue(c, b);
}

int edgehip_forward_match(edgehip_ctx *c, int slot_old, int slot_new) {
    EH_ENTER(c);
    if (int e = chk2(c, slot_old, slot_new)) return e;
    return forward_match_enqueue(c, slot_old, slot_new);
}
int edgehip_rotate_keylines(edgehip_ctx *c, int slot, const double *R) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    return rotate_enqueue(c, slot, R);
}
int edgehip_directed_matching(edgehip_ctx *c, int slot_new, int slot_old) {
    EH_ENTER(c);
    if (int e = chk2(c, slot_new, slot_old)) return e;
    return directed_enqueue(c, slot_new, slot_old);
}
int edgehip_regularize_ekf(edgehip_ctx *c, int slot, int do_reg, int do_ekf) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    return regekf_enqueue(c, slot, do_reg, do_ekf);
}
int edgehip_rescale(edgehip_ctx *c, int slot) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    return rescale_enqueue(c, slot);
}

int edgehip_set_slot_camera(edgehip_ctx *c, int slot, double ppx, double ppy, double zfx, double zfy) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    drop_frame_graphs(c);
    // REBVOParameters / cam_model keep these as float (cam_model.h:51-57)
    c->slot_cam[slot].ppx = (float)ppx;
    c->slot_cam[slot].ppy = (float)ppy;
    c->slot_cam[slot].zfm = (double)(((float)zfx + (float)zfy) / 2);
    return 0;
}

static int stereo_enqueue(edgehip_ctx *c, int slot, int slot_pair, const double *t, const double *R, double min_thr_mod,
                          double min_thr_ang, double max_radius, double loc_unc, double loc_unc_model, bool frame_driver) {
    const DevicePlan &pl = c->plan;
    if (int e = rec_refresh_enqueue(c, slot_pair)) return e;   // (a pair slot somebody rotated through the stage-level API)
    StereoArgs a;
    a.kl = kldev(c, slot); a.kl_pair = kldev(c, slot_pair);
    a.kn = c->kn_slot + (size_t)slot * pl.nseq;
    a.mask_pair = maskof(c, slot_pair);
    a.nmatch = c->stereo_cnt;
    a.w = pl.w; a.h = pl.h; a.n = pl.n;
    memcpy(a.t, t, sizeof a.t); memcpy(a.R, R, sizeof a.R);
    a.zfm0 = c->slot_cam[slot].zfm; a.zfm1 = c->slot_cam[slot_pair].zfm;
    a.pp1x = c->slot_cam[slot_pair].ppx; a.pp1y = c->slot_cam[slot_pair].ppy;
    a.min_thr_mod = min_thr_mod; a.cang_min_edge = cos(min_thr_ang * M_PI / 180.0); a.max_radius = max_radius;
    a.loc_unc = loc_unc; a.loc_unc_model = loc_unc_model;
    a.seqs = frame_driver ? c->seq : nullptr;
    EH_CHECK(hipMemsetAsync(c->stereo_cnt, 0, sizeof(int32_t) * pl.nseq, c->stream));
    hipLaunchKernelGGL(k_stereo_match, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, a);
    EH_LAUNCH_CHECK();
    return 0;
}
static int fuse_stereo_enqueue(edgehip_ctx *c, int slot, bool frame_driver) {
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_fuse_stereo, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, frame_driver ? c->seq : nullptr);
    EH_LAUNCH_CHECK();
    return 0;
}

int edgehip_directed_matching_stereo(edgehip_ctx *c, int slot, int slot_pair, const double *t, const double *R, double min_thr_mod,
                                     double min_thr_ang, double max_radius, double loc_unc, double q_abs, double q_rel,
                                     double loc_unc_model, int32_t *nmatch) {
    EH_ENTER(c);
    (void)q_abs; (void)q_rel;
    if (int e = chk2(c, slot, slot_pair)) return e;
    if (!t || !R) return EDGEHIP_ERR_ARG;
    if (c->imu_enabled) { set_error("set_stereo_rig: the device IMU branch does not run the stereo rig (edgehip_imu_enable was called)"); return EDGEHIP_ERR_STATE; }
    if (!c->p.stereo_available) { set_error("directed_matching_stereo: context created without stereo_available"); return EDGEHIP_ERR_STATE; }
    if (int e = stereo_enqueue(c, slot, slot_pair, t, R, min_thr_mod, min_thr_ang, max_radius, loc_unc, loc_unc_model, false)) return e;
    if (nmatch) EH_CHECK(hipMemcpyAsync(nmatch, c->stereo_cnt, sizeof(int32_t) * c->plan.nseq, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_fuse_stereo_depth(edgehip_ctx *c, int slot) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    if (!c->p.stereo_available) { set_error("fuse_stereo_depth: context created without stereo_available"); return EDGEHIP_ERR_STATE; }
    return fuse_stereo_enqueue(c, slot, false);
}

int edgehip_set_stereo_rig(edgehip_ctx *c, int slot_pair, const double *t, const double *R, double max_radius) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    drop_frame_graphs(c);
    if (slot_pair < 0) {   // switch the rig off: the whole ring is available again
        c->rig.enabled = false;
        c->ring_slots = c->plan.nslots;
        return 0;
    }
    if (!t || !R) return EDGEHIP_ERR_ARG;
    if (!c->p.stereo_available) { set_error("set_stereo_rig: context created without stereo_available"); return EDGEHIP_ERR_STATE; }
    if (slot_pair != c->plan.nslots - 1 || c->plan.nslots < 3) {
        set_error("set_stereo_rig: the pair slot is the last of at least three slots");
        return EDGEHIP_ERR_ARG;
    }
    if (c->frames_seen != 0) { set_error("set_stereo_rig: set the rig before the first frame (or after edgehip_reset)"); return EDGEHIP_ERR_STATE; }
    c->rig.enabled = true;
    c->rig.slot_pair = slot_pair;
    memcpy(c->rig.t, t, sizeof c->rig.t);
    memcpy(c->rig.R, R, sizeof c->rig.R);
    c->rig.max_radius = max_radius;
    c->ring_slots = c->plan.nslots - 1;
    return 0;
}

int edgehip_get_stereo_matches(edgehip_ctx *c, int32_t *nmatch) {
    EH_ENTER(c);
    if (!c || !nmatch) return EDGEHIP_ERR_ARG;
    if (!c->stereo_cnt) { set_error("get_stereo_matches: context created without stereo_available"); return EDGEHIP_ERR_STATE; }
    EH_CHECK(hipMemcpyAsync(nmatch, c->stereo_cnt, sizeof(int32_t) * c->plan.nseq, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// Host-side 6x6 symmetric eigen-decomposition (cyclic Jacobi), standing in for LAPACK dgesvd_ behind TooN::SVD<>
static void jacobi_eig6_host(const double Ain[36], double V[36], double e[6]) {
    double A[36];
    for (int i = 0; i < 36; i++) { A[i] = Ain[i]; V[i] = 0; }
    for (int i = 0; i < 6; i++) V[i * 7] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < 6; p++) {
            diag += A[p * 7] * A[p * 7];
            for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q];
        }
        if (!(off > 1e-34 * diag) || !(off > 0)) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0) continue;
                const double theta = (A[q * 7] - A[p * 7]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < 6; k++) { const double a = A[k * 6 + p], b = A[k * 6 + q]; A[k * 6 + p] = cs * a - sn * b; A[k * 6 + q] = sn * a + cs * b; }
                for (int k = 0; k < 6; k++) { const double a = A[p * 6 + k], b = A[q * 6 + k]; A[p * 6 + k] = cs * a - sn * b; A[q * 6 + k] = sn * a + cs * b; }
                for (int k = 0; k < 6; k++) { const double a = V[k * 6 + p], b = V[k * 6 + q]; V[k * 6 + p] = cs * a - sn * b; V[k * 6 + q] = sn * a + cs * b; }
            }
    }
    for (int i = 0; i < 6; i++) e[i] = A[i * 7];
}

int edgehip_ext_rot_vel(edgehip_ctx *c, int slot, const double *vel, double loc_unc, double hub_reweight, double *X, double *Wx,
                        double *Rx, int32_t *ok) {
    EH_ENTER(c);
    if (int e = chk2(c, slot, slot)) return e;
    if (!vel || !X) return EDGEHIP_ERR_ARG;
    const DevicePlan &pl = c->plan;
    const int B = pl.nseq, nblk = (pl.cap + 255) / 256;
    if (nblk > c->nblk_tvr) { set_error("ext_rot_vel: partial table too small"); return EDGEHIP_ERR_STATE; }
    EH_CHECK(hipStreamSynchronize(c->stream));
    memcpy(c->pinned_out, vel, sizeof(double) * 3 * B);
    double *dvel = c->rot_buf;                                      // [B][9] scratch, 3 used
    EH_CHECK(hipMemcpyAsync(dvel, c->pinned_out, sizeof(double) * 3 * B, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemsetAsync(c->partials, 0, sizeof(double) * (size_t)B * c->nblk_tvr * kNumSums, c->stream));
    hipLaunchKernelGGL(k_ext_rotvel, dim3(nblk, 1, B), dim3(256), 0, c->stream, kldev(c, slot), c->kn_slot + (size_t)slot * B, dvel,
                       (const SeqDev *)nullptr, c->partials, c->nblk_tvr, pl.zfm, loc_unc, hub_reweight);
    EH_LAUNCH_CHECK();
    std::vector<double> part((size_t)B * c->nblk_tvr * kNumSums);
    EH_CHECK(hipMemcpyAsync(part.data(), c->partials, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < B; s++) {
        double sum[kNumSums] = {0};
        for (int b = 0; b < nblk; b++)
            for (int k = 0; k < kNumSums; k++) sum[k] += part[((size_t)s * c->nblk_tvr + b) * kNumSums + k];
        double JtJ[36], JtF[6];
        int ns = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { JtJ[a * 6 + b] = sum[ns]; JtJ[b * 6 + a] = sum[ns]; ns++; }
        for (int a = 0; a < 6; a++) JtF[a] = sum[ns++];
        // X = SVD(JtJ).backsub(JtF), Rx = SVD(JtJ).get_pinv() with TooN's conditioning (SVD.h:176-207, 1e9)
        double V[36], e[6], inv[6], smax = 0;
        jacobi_eig6_host(JtJ, V, e);
        for (int i = 0; i < 6; i++) smax = fmax(smax, fabs(e[i]));
        for (int i = 0; i < 6; i++) inv[i] = (fabs(e[i]) * 1e9 <= smax) ? 0.0 : 1.0 / e[i];
        bool good = true;
        for (int r = 0; r < 6; r++) {
            double x = 0;
            for (int cc = 0; cc < 6; cc++) {
                double p = 0;
                for (int i = 0; i < 6; i++) p += V[r * 6 + i] * inv[i] * V[cc * 6 + i];
                if (Rx) Rx[(size_t)s * 36 + r * 6 + cc] = p;
                if (p != p) good = false;
                x += p * JtF[cc];
            }
            X[(size_t)s * 6 + r] = x;
            if (x != x) good = false;
        }
        if (Wx) memcpy(Wx + (size_t)s * 36, JtJ, sizeof JtJ);
        if (ok) ok[s] = good ? 1 : 0;
    }
    return 0;
}

int edgehip_depth_reset(edgehip_ctx *c, int seq) {
    EH_ENTER(c);
    if (!c || seq >= c->plan.nseq) return EDGEHIP_ERR_ARG;
    if (c->frame_slot < 0) return 0;  // nothing detected yet: the initial state already is the reset state
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_depth_reset, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream,
                       kldev(c, c->frame_slot), c->kn_slot + (size_t)c->frame_slot * pl.nseq, c->seq, seq, c->imu_enabled ? 0 : 1);
    EH_LAUNCH_CHECK();
    if (c->imu_enabled) return imu_pose_reset_enqueue(c, seq);
    return 0;
}

int edgehip_depth_reset_slot(edgehip_ctx *c, int seq, int slot) {
    EH_ENTER(c);
    if (!c || seq >= c->plan.nseq || slot < 0 || slot >= c->plan.nslots) return EDGEHIP_ERR_ARG;
    if (int e = rot_materialize_enqueue(c, slot)) return e;
    const DevicePlan &pl = c->plan;
    hipLaunchKernelGGL(k_depth_reset, dim3((pl.cap + 255) / 256, 1, pl.nseq), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * pl.nseq, c->seq, seq, c->imu_enabled ? 0 : 1);
    EH_LAUNCH_CHECK();
    if (c->imu_enabled) return imu_pose_reset_enqueue(c, seq);
    return 0;
}

int edgehip_next_slot(edgehip_ctx *c) { return c ? (c->frame_slot + 1) % c->ring_slots : -1; }
int edgehip_cur_slot(edgehip_ctx *c) { return c ? c->frame_slot : -1; }

// Everything edgehip_process_frame enqueues for one frame (both streams); also what gets captured into a graph.
static int frame_enqueue(edgehip_ctx *c, int sn, int so, int sp, int have_pair, const double *tp) {
    int e;
    c->t_src = tp;   // page-locked ring entry, read in place by the frame-begin glue (no copy on the critical path)
#define EH_TRY(x) if ((e = (x)) != 0) return e
    // Stage A of this frame runs on its own stream: it only has to wait for the B/C work that still reads the slot it
    // overwrites (two frames back), so it overlaps the tracking/mapping of the previous frame — what the reference's
    // first and second thread do (rebvo_first_t.cpp:134, rebvo_second_t.cpp:102).
    // With overlap off (the default: per-kernel timings stay attributable) stage A simply follows everything before it.
    if (!c->overlap) {
        EH_TRY(order_a_after_bc(c));
    } else if (c->use_valid[sn]) {
        EH_CHECK(hipStreamWaitEvent(c->stream_a, c->ev_use[sn], 0));
    }
    if (sp >= 0 && c->overlap && c->use_valid[sp]) EH_CHECK(hipStreamWaitEvent(c->stream_a, c->ev_use[sp], 0));
    // (with a frame pair FordwardMatch follows below in every branch: the detector may leave the forwarded fields to it;
    // mode 2 keeps its scattering pass, which cannot fill)
    bool frame_ended = false;
    const bool begin_in_quantile = have_pair != 0;                  // k_quantile, the first kernel of stage B, does it per sequence
    // ... and finishes the detector's reEstimateThresh — unless something may overwrite the histogram and the detector's extremes before
    // k_quantile has read them: the pair image's stage A, or (with overlap) the next frame's, which runs beside this frame's stage B
    const bool retune_in_quantile = begin_in_quantile && sp < 0 && !c->overlap;
    EH_TRY(stage_a_enqueue(c, sn, have_pair && c->fwd_mode != 2, retune_in_quantile));
    if (sp >= 0) EH_TRY(stage_a_enqueue(c, sp));   // the pair image, after the main one as in rebvo_first_t.cpp:259-290
    if (sp >= 0) {   // the pair slot's frame storage is free from here on: the next pair frame's copy may run under the rest of this frame
        c->rig_a_valid = false;
        if (!c->capturing) {
            EH_CHECK(hipEventRecord(c->ev_a[sp], c->stream_a));
            c->rig_a_valid = true;
        }
    }
    if (c->stream_a != c->stream) {   // (one stream: already in order, and an event record is a packet the device has to work through)
        EH_CHECK(hipEventRecord(c->ev_a[sn], c->stream_a));
        EH_CHECK(hipStreamWaitEvent(c->stream, c->ev_a[sn], 0));
    }
    if (!begin_in_quantile) {
        ProfScope ps(c, PROF_C_POSE);
        EH_TRY(glue(c, 0, sn, have_pair));
    }
    if (c->imu_enabled) {
        // ---- ImuMode > 0 (rebvo_second_t.cpp:182-336, 387-493, 519-606): everything on the device, stage_imu.hip has the filters ----
        EH_TRY(imu_begin_enqueue(c));
        if (have_pair) {
            EH_TRY(quantile_enqueue(c, so, kRhoMin, kRhoMax, c->p.qcut_quantile, c->p.qcut_nbins, true, retune_in_quantile ? sn : -1));  // :145-168, :172
            EH_TRY(build_field_enqueue(c, sn, c->p.search_range, -1.f, true));                        // :177 (also resets FordwardMatch's arbitration arrays)
            { ProfScope ps(c, PROF_IMU_FILTERS); EH_TRY(imu_pre_enqueue(c, so)); }                         // :183-213
            EH_TRY(rotate_buf_enqueue(c, so));                                                       // :215 gyro pre-rotation
            {
                ProfScope ps(c, PROF_B_MINIMIZER_V);
                EH_TRY(minimizer_v_enqueue(c, sn, so, c->frames_seen % kRefRing, c->p.tracker_iter_num, c->p.tracker_match_thresh,
                                           c->p.match_num_thresh, c->p.reweight_distance));          // :223
            }
            EH_TRY(forward_match_enqueue(c, so, sn, c->fwd_keys_posted));                            // :230
            c->fwd_keys_posted = false;
            { ProfScope ps(c, PROF_C_EXTROTVEL); EH_TRY(ext_rotvel_enqueue(c, sn)); }                 // :237
            { ProfScope ps(c, PROF_IMU_FILTERS); EH_TRY(imu_mid_enqueue(c)); }                             // :237-272, :387-397
            EH_TRY(rotate_buf_enqueue(c, so));                                                       // :319
            EH_TRY(directed_enqueue(c, sn, so));                                                     // :410
            EH_TRY(regekf_enqueue(c, sn, 1, 1, true));                                               // :412-422 (in k_regularize), :453, :460
            EH_TRY(rescale_enqueue(c, sn));                                                          // :487
        }
        EH_TRY(imu_post_enqueue(c, sn, have_pair));                                                  // :280-312, :519-606
    } else if (have_pair) {
        EH_TRY(quantile_enqueue(c, so, kRhoMin, kRhoMax, c->p.qcut_quantile, c->p.qcut_nbins, true, retune_in_quantile ? sn : -1));  // rebvo_second_t.cpp:145-168, :172
        EH_TRY(build_field_enqueue(c, sn, c->p.search_range, -1.f, c->fwd_mode != 1));            // :177
        c->fwd_key_in_tvr = c->fwd_mode != 1;
        e = minimizer_enqueue(c, sn, so, c->frames_seen % kRefRing);                              // :346
        c->fwd_key_in_tvr = false;
        if (e) return e;
        // matching in one pass (ctx.h: fuse_match): the forward copy waits for k_directed, which needs the unturned old KeyLines for it
        const bool one_pass = frame_matches_in_one_pass(c, sp);
#ifdef EDGEHIP_EXPERIMENTS
        if (c->fwd_mode == 2) {
            EH_TRY(forward_rotate_enqueue(c, so, sn));                                           // :354-369
        } else
#endif
        if (one_pass) {
            EH_TRY(forward_rotate_one_pass_enqueue(c, so, sn));                                  // :354 (arbitration), :360-369, :387-397
        } else {
            EH_TRY(forward_match_enqueue(c, so, sn, c->fwd_mode != 1, true));                    // :354 (+ exp(W) and :387-397 in k_fwd_win's tail)
            EH_TRY(rotate_enqueue(c, so, nullptr, true));                                        // :360-369
        }
        if (c->fwd_mode == 2) { ProfScope ps(c, PROF_C_POSE); EH_TRY(glue(c, 1, sn, have_pair)); }   // :387-397
        EH_TRY(directed_enqueue(c, sn, so, one_pass));                                           // :410
        EH_TRY(regekf_enqueue(c, sn, 1, 1, true));                                               // :412-422 (in k_regularize), :453, :460
        if (sp >= 0) {                                                                           // :465-486
            EH_TRY(stereo_enqueue(c, sn, sp, c->rig.t, c->rig.R, c->p.match_thresh_module, c->p.match_thresh_angle, c->rig.max_radius,
                                  c->p.loc_unc_match, c->p.loc_unc, true));
            EH_TRY(fuse_stereo_enqueue(c, sn, true));
        } else {
            EH_TRY(rescale_enqueue(c, sn, true));                                                // :487 and, in its tail, :550-606
            frame_ended = true;
        }
    }
    if (!c->imu_enabled && !frame_ended) {
        ProfScope ps(c, PROF_C_POSE);
        EH_TRY(glue(c, 3, sn, have_pair));                                                       // :550-606
    }
#undef EH_TRY
    // B/C of this frame were the last readers of both slots (only a stage-A stream of its own has to be told)
    if (c->stream_a != c->stream) {
        EH_CHECK(hipEventRecord(c->ev_use[sn], c->stream));
        c->use_valid[sn] = true;
        if (so >= 0) {
            EH_CHECK(hipEventRecord(c->ev_use[so], c->stream));
            c->use_valid[so] = true;
        }
        if (sp >= 0) {
            EH_CHECK(hipEventRecord(c->ev_use[sp], c->stream));
            c->use_valid[sp] = true;
        }
    }
    return 0;
}

int edgehip_process_frame(edgehip_ctx *c, const double *t) {
    EH_ENTER(c);
    if (!c || !t) return EDGEHIP_ERR_ARG;
    const DevicePlan &pl = c->plan;
    // ImuMode > 0: the reference grabs the IMU data of the interval for every frame (rebvo_first_t.cpp:232-246); a frame without
    // edgehip_set_imu would silently integrate the previous interval's record again
    if (c->imu_enabled && !c->imu_pending) { set_error("process_frame: edgehip_set_imu was not called for this frame"); return EDGEHIP_ERR_STATE; }
    const int sn = (c->frame_slot + 1) % c->ring_slots, so = c->frame_slot;
    const int sp = c->rig.enabled ? c->rig.slot_pair : -1;   // stereo pair slot (its frame was uploaded by the caller)
    const int have_pair = c->frames_seen >= 1;
    // time stamps travel through a small ring of pinned slots so that back-to-back frames need no sync
    double *tp = c->pinned_t + (size_t)(c->frames_seen % 8) * pl.nseq;
    if (int e = wait_pinned_ring(c)) return e;
    memcpy(tp, t, sizeof(double) * pl.nseq);
    const bool profiling = c->prof && c->prof->on;
    const bool graph_path = c->use_graph && !c->overlap && !profiling && c->frames_seen >= 2;
    // frames uploaded on the upload stream (edgehip_upload_rgb_pinned): stage A waits for the copy into its slot
    if (int e = wait_upload(c, sn, graph_path ? c->stream : c->stream_a)) return e;
    if (sp >= 0) { if (int e = wait_upload(c, sp, graph_path ? c->stream : c->stream_a)) return e; }
    // the first frames run eagerly (one-time kernel attributes, no frame pair yet); then every (slot, FrameCount row,
    // pinned time-stamp slot) combination — period lcm(ring, 8) — is captured once and replayed
    if (graph_path) {
        const int key = sn + 8 * (c->frames_seen % 8) + 64 * (sp + 1) + 512 * c->slot_src[sn].idx_ring + 4096 * (sp >= 0 ? c->slot_src[sp].idx_ring : 0);   // every pointer a node bakes in
        if (int e = order_bc_after_a(c)) return e;   // the caller's uploads (stage-A stream) precede the graph
        auto it = c->frame_graphs.find(key);
        if (it == c->frame_graphs.end()) {
            hipGraph_t graph = nullptr;
            EH_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            c->capturing = true;
            const int rc = frame_enqueue(c, sn, so, sp, have_pair, tp);
            c->capturing = false;
            const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
            if (rc != 0 || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                if (rc != 0) return rc;
                return hip_fail(ce != hipSuccess ? ce : hipErrorUnknown, "hipStreamEndCapture(frame graph)", __FILE__, __LINE__);
            }
            hipGraphExec_t exec = nullptr;
            const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) return hip_fail(ie, "hipGraphInstantiate(frame graph)", __FILE__, __LINE__);
            it = c->frame_graphs.emplace(key, exec).first;
        }
        EH_CHECK(hipGraphLaunch(it->second, c->stream));
        c->rig_a_valid = false;   // (a replayed graph records no event behind the pair's stage A: pair uploads wait for the frame's end)
        // stage A ran inside the graph, on this stream: the caller's next uploads (stage-A stream) may overwrite a
        // frame slot only after the graphs that read it (a caller that never synchronises is several frames ahead)
        if (int e = order_a_after_bc(c)) return e;
        // host-side bookkeeping the replayed enqueue code would have done (stage A: fresh KeyLines; rotate_keylines
        // of the old slot)
        c->grec_ok[sn] = true; c->rec_stale[sn] = false; c->rot_pending[sn] = false;
        if (sp >= 0) { c->grec_ok[sp] = true; c->rec_stale[sp] = false; c->rot_pending[sp] = false; }
        if (have_pair && so >= 0) {
            c->grec_ok[so] = false; c->rec_stale[so] = true;
            c->rot_pending[so] = !c->imu_enabled && frame_matches_in_one_pass(c, sp);   // what the captured frame_enqueue decided
        }
    } else {
        if (int e = frame_enqueue(c, sn, so, sp, have_pair, tp)) return e;
    }
    // the pinned ring entries of this frame (time stamps, bound frame indices) are free once it has run
    EH_CHECK(hipEventRecord(c->ev_ring[c->frames_seen % 8], c->stream));
    c->ring_valid[c->frames_seen % 8] = true;
    c->slot_ring[sn] = c->frames_seen % 8;
    c->a_api_valid[sn] = false;
    if (sp >= 0) { c->slot_ring[sp] = c->frames_seen % 8; c->a_api_valid[sp] = false; }
    c->frame_slot = sn;
    c->frames_seen++;
    c->imu_pending = false;
    if (c->nav_log) {   // edgehip_read_nav_log (possibly another thread) orders its copies after this frame's record
        std::lock_guard<std::mutex> g(c->log_mu);
        EH_CHECK(hipEventRecord(c->ev_log, c->stream_imu ? c->stream_imu : c->stream));
        EH_CHECK(hipEventRecord(c->ev_log_ring[(c->frames_seen - 1) % 8], c->stream_imu ? c->stream_imu : c->stream));
        if (c->frames_logged.load() == 0) c->log_first = c->frames_seen - 1;
        c->log_last = c->frames_seen - 1;
        c->frames_logged++;
    }
    return 0;
}

int edgehip_read_nav(edgehip_ctx *c, edgehip_nav *nav) {
    EH_ENTER(c);
    if (!c || !nav) return EDGEHIP_ERR_ARG;
    if (c->stream_imu) EH_CHECK(hipStreamSynchronize(c->stream_imu));   // ImuMode > 0: the records are written on the IMU stream
    EH_CHECK(hipMemcpyAsync(c->pinned_nav, c->nav_dev, sizeof(edgehip_nav) * c->plan.nseq, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    memcpy(nav, c->pinned_nav, sizeof(edgehip_nav) * c->plan.nseq);
    return 0;
}

}  // extern "C"
