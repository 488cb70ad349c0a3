// edgeport_b.cpp — CPU restatement, stage B: uncertainty quantile, auxiliary distance field, TryVelRot and the
// Levenberg-Marquardt driver Minimizer_RV.  TEST INFRASTRUCTURE ONLY (see edgeport_a.cpp for the rules).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "edgeport.h"
#include "port_math.h"

namespace port {

// parity diagnostics: the decompositions the minimiser asked for (port_svd_trace, oracle_abi.h)
static OrcSvdRec *g_svd_buf = nullptr;
static int g_svd_cap = 0, g_svd_n = 0;
void svd_trace_record(const double A[36], const double e[6]) {
    if (!g_svd_buf || g_svd_n >= g_svd_cap) return;
    OrcSvdRec *r = &g_svd_buf[g_svd_n++];
    r->rows = r->cols = 6;
    for (int i = 0; i < 36; i++) r->A[i] = A[i];
    for (int i = 0; i < 6; i++) r->s[i] = e[i];
}
void svd_trace_set(OrcSvdRec *buf, int cap) { g_svd_buf = buf; g_svd_cap = cap; g_svd_n = 0; }
int svd_trace_count() { return g_svd_n; }

// ---- edge_tracker::EstimateQuantile (src/mtracklib/edge_tracker.cpp:1148-1186) ----------------------------------
double estimate_quantile(const Slot &s, double s_rho_min, double s_rho_max, double percentile, int n) {
    std::vector<int> histo(n, 0);
    for (int ikl = 0; ikl < s.kn; ikl++) {
        int i = n * (s.kl[ikl].s_rho - s_rho_min) / (s_rho_max - s_rho_min);
        i = i > n - 1 ? n - 1 : i;
        i = i < 0 ? 0 : i;
        histo[i]++;
    }
    double s_rho = 1e3;
    for (int i = 0, a = 0; i < n; i++) {
        if (a > percentile * s.kn) {
            s_rho = (double)i * (s_rho_max - s_rho_min) / (double)n + s_rho_min;
            break;
        }
        a += histo[i];
    }
    return s_rho;
}

// ---- global_tracker::build_field (src/mtracklib/global_tracker.cpp:61-105) ---------------------------------------
// Sequential scatter: a pixel keeps the KeyLine with the smallest |t|; among equal distances the LAST one written.
void build_field(Ctx &c, int slot, int radius, float min_mod) {
    Slot &s = c.slots[slot];
    const int w = c.p.w, h = c.p.h;
    s.max_r = radius;
    s.field_slot = slot;
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) s.field[2 * i + 1] = -1;            // only ikl is reset (:66-68)
    for (int ikl = 0; ikl < s.kn; ikl++) {
        const OrcKeyLine &kl = s.kl[ikl];
        if (min_mod > 0 && kl.n_m < min_mod) continue;
        for (int t = -radius; t < radius; t += 1) {
            const float fx = kl.u_m[0] * (float)t + kl.c_p[0], fy = kl.u_m[1] * (float)t + kl.c_p[1];
            const int xi = (int)std::round(fx), yi = (int)std::round(fy);   // Image::GetIndexRC (image.h:121-126)
            if (xi >= w || yi >= h || xi < 0 || yi < 0) continue;
            const size_t inx = (size_t)yi * w + xi;
            const int at = std::abs(t);
            if (s.field[2 * inx + 1] >= 0 && at > s.field[2 * inx]) continue;
            s.field[2 * inx] = at;
            s.field[2 * inx + 1] = ikl;
        }
    }
}

// ---- Ne10 C fallbacks used by TryVelRot (include/UtilLib/ne10wrapper.h:224-445) ------------------------------------
// PairWiseVAdd: halve the vector by adding its two halves until fewer than 4 elements remain, collecting the odd
// element of each step, then add what is left in order (:334-361).  The summation tree is part of the result.
static double pairwise_vadd(const double *src, int pnum, std::vector<double> &b0, std::vector<double> &b1) {
    b0.resize(pnum >> 1);
    b1.resize(pnum >> 1);
    const double *p_in = src;
    double *p_out = b0.data();
    double odd = 0;
    bool swap = true;
    while (pnum > 3) {
        odd += (pnum & 0x01) ? p_in[pnum - 1] : 0;
        pnum >>= 1;
        for (int k = 0; k < pnum; k++) p_out[k] = p_in[k] + p_in[pnum + k];
        if (swap) { p_in = b0.data(); p_out = b1.data(); }
        else      { p_in = b1.data(); p_out = b0.data(); }
        swap = !swap;
    }
    for (int i = 0; i < pnum; i++) odd += p_in[i];
    return odd;
}
static double dot_product(const double *a, const double *b, int pnum, std::vector<double> &prod, std::vector<double> &b0,
                          std::vector<double> &b1) {
    prod.resize(pnum);
    for (int k = 0; k < pnum; k++) prod[k] = a[k] * b[k];
    return pairwise_vadd(prod.data(), pnum, b0, b1);
}

// ---- KltoI3PMatrix + Ne10::ProyI3Pto3PMatrix (global_tracker.cpp:553-570; ne10wrapper.h:414-424) --------------------
void kl_to_p0(const Ctx &c, const Slot &klist, int pnum, std::vector<double> &P0m) {
    std::vector<double> P0Im((size_t)pnum * 3);
    int ikl = 0;
    for (; ikl < klist.kn; ikl++) {
        P0Im[ikl] = klist.kl[ikl].p_m[0];
        P0Im[pnum + ikl] = klist.kl[ikl].p_m[1];
        P0Im[2 * pnum + ikl] = klist.kl[ikl].rho;
    }
    for (; ikl < pnum; ikl++) { P0Im[ikl] = 0; P0Im[pnum + ikl] = 0; P0Im[2 * pnum + ikl] = 1; }
    P0m.resize((size_t)pnum * 3);
    for (int i = 0; i < pnum; i++) P0m[2 * pnum + i] = 1 / P0Im[2 * pnum + i];
    for (int i = 0; i < pnum; i++) {
        const double pz_zf = (1 / c.zfm) * P0m[2 * pnum + i];
        P0m[i] = pz_zf * P0Im[i];
        P0m[pnum + i] = pz_zf * P0Im[pnum + i];
    }
}

// ---- global_tracker::TryVelRot<double,ReWeight,ProcJF,false> (global_tracker.cpp:289-543) + Calc_f_J2 (:228-271)
// + Test_f_k (include/mtracklib/global_tracker.h:90-104).  `gt` = the slot that owns the field (new edge map),
// `klist` = the old edge map whose KeyLines are transformed.
double try_velrot(Ctx &c, Slot &gt, Slot &klist, bool ReWeight, bool ProcJF, double JtJ[36], double JtF[6], const double VelRot[6],
                  const double *P0m, int pnum, double match_thresh, double s_rho_min, unsigned MatchNumThresh, double k_huber,
                  const double *DResidual, double *DResidualNew) {
    const int w = c.p.w, h = c.p.h;
    const Slot &fl = c.slots[gt.field_slot];                            // klist_f
    double R0[9], RMz[9];
    so3_exp(VelRot + 3, R0);
    const double wz[3] = {0, 0, VelRot[5]};
    so3_exp(wz, RMz);
    const double RM[4] = {RMz[0], RMz[1], RMz[3], RMz[4]};
    const double max_r = gt.max_r;
    std::vector<double> Ptm((size_t)pnum * 3), PtIm((size_t)pnum * 3), dfx(pnum), dfy(pnum), fm(pnum);
    const double Vt[3] = {VelRot[0], VelRot[1], VelRot[2]};
    // Ne10::SE3on3PMatrix (MulMat3Vect then AddCVect, ne10wrapper.h:364-404) with Rt(j,i) = R0(i,j)
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < pnum; k++) {
            double d = R0[r * 3 + 0] * P0m[k];
            d += R0[r * 3 + 1] * P0m[pnum + k];
            d += R0[r * 3 + 2] * P0m[2 * pnum + k];
            Ptm[(size_t)r * pnum + k] = Vt[r] + d;
        }
    // Ne10::ProyP3toI3PMatrix (:430-445)
    for (int k = 0; k < pnum; k++) {
        PtIm[2 * pnum + k] = 1 / Ptm[2 * pnum + k];
        const double pz_zf = c.zfm * PtIm[2 * pnum + k];
        PtIm[k] = pz_zf * Ptm[k];
        PtIm[pnum + k] = pz_zf * Ptm[pnum + k];
    }
    double fi = 0;   // NOT reset per KeyLine: an unmatched KeyLine stores the residual of the last matched one (:344, 391, 406)
    const unsigned mthr = std::min(MatchNumThresh, gt.FrameCount);
    for (int ikl = 0; ikl < klist.kn; ikl++) {
        OrcKeyLine &kl = klist.kl[ikl];
        kl.m_id_f = -1;
        if (kl.s_rho > s_rho_min || (unsigned)kl.m_num < mthr) {       // :356 (int vs uint comparison)
            fm[ikl] = 0; dfx[ikl] = 0; dfy[ikl] = 0;
            continue;
        }
        const double px = PtIm[ikl] + c.ppx, py = PtIm[pnum + ikl] + c.ppy;     // cam_model::Hom2Img
        const int x = (int)(px + 0.5), y = (int)(py + 0.5);                     // util::round2int_positive
        double weigth = 1;
        if (ReWeight && std::fabs(DResidual[ikl]) > k_huber) weigth = k_huber / std::fabs(DResidual[ikl]);
        if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) {
            fm[ikl] = max_r;
            if (ReWeight) fm[ikl] *= weigth;
            dfx[ikl] = 0; dfy[ikl] = 0;
            DResidualNew[ikl] = max_r;
            continue;
        }
        // gradient temporarily rotated about z, stored back into the float Point2DF (:386-388)
        const float klmx = kl.m_m[0], klmy = kl.m_m[1];
        const float rmx = RM[0] * klmx + RM[1] * klmy;
        const float rmy = RM[2] * klmx + RM[3] * klmy;
        // Calc_f_J2
        const size_t f_inx = (size_t)y * w + x;
        const int fikl = gt.field[2 * f_inx + 1];
        double f = max_r, gx = 0, gy = 0;
        if (fikl >= 0) {
            const OrcKeyLine &f_kl = fl.kl[fikl];
            const double p_n2 = (kl.n_m * kl.n_m);                                // Test_f_k: float products, compared in double
            const double p_esc = rmx * f_kl.m_m[0] + rmy * f_kl.m_m[1];
            if (!(std::fabs(p_esc - p_n2) > match_thresh * p_n2)) {
                const double dx = px - f_kl.c_p[0], dy = py - f_kl.c_p[1];
                fi = (dx * f_kl.u_m[0] + dy * f_kl.u_m[1]);
                gx = f_kl.u_m[0];
                gy = f_kl.u_m[1];
                kl.m_id_f = fikl;
                f = fi;
            }
        }
        fm[ikl] = f; dfx[ikl] = gx; dfy[ikl] = gy;
        if (ReWeight) { fm[ikl] *= weigth; dfx[ikl] *= weigth; dfy[ikl] *= weigth; }
        DResidualNew[ikl] = fi;
    }
    std::vector<double> prod, b0, b1;
    if (ProcJF) {
        std::vector<double> Jm((size_t)pnum * 6), tmp(pnum);
        for (int k = 0; k < pnum; k++) {                                          // :419-449, same products in the same order
            double t0 = c.zfm * PtIm[2 * pnum + k];
            Jm[k] = t0 * dfx[k];
            Jm[pnum + k] = t0 * dfy[k];
            t0 = PtIm[2 * pnum + k] * PtIm[k];
            Jm[2 * pnum + k] = t0 * dfx[k];
            t0 = PtIm[2 * pnum + k] * PtIm[pnum + k];
            Jm[2 * pnum + k] += t0 * dfy[k];
            Jm[3 * pnum + k] = Jm[pnum + k] * Ptm[2 * pnum + k];
            Jm[3 * pnum + k] += Jm[2 * pnum + k] * Ptm[pnum + k];
            Jm[4 * pnum + k] = Jm[k] * Ptm[2 * pnum + k];
            Jm[4 * pnum + k] += Jm[2 * pnum + k] * Ptm[k];
            t0 = Jm[k] * Ptm[pnum + k];
            Jm[5 * pnum + k] = -1 * t0;
            Jm[5 * pnum + k] += Jm[pnum + k] * Ptm[k];
        }
        for (int ikl = 0; ikl < klist.kn; ikl++) {                                // :452-463
            const double qvel = (c.zfm * dfx[ikl] * Vt[0] + c.zfm * dfy[ikl] * Vt[1] +
                                 (PtIm[ikl] * dfx[ikl] + PtIm[pnum + ikl] * dfy[ikl]) * Vt[2]);
            double q_rho = std::sqrt(klist.kl[ikl].s_rho * qvel * klist.kl[ikl].s_rho * qvel + 1);
            if (!ReWeight) q_rho = klist.kl[ikl].s_rho;
            for (int j = 0; j < 6; j++) Jm[(size_t)pnum * j + ikl] /= q_rho;
            fm[ikl] /= q_rho;
        }
        for (int ikl = klist.kn; ikl < pnum; ikl++) {
            for (int i = 0; i < 6; i++) Jm[(size_t)pnum * i + ikl] = 0;
            fm[ikl] = 0;
        }
        for (int i = 0; i < 6; i++) {
            for (int j = i; j < 6; j++) JtJ[i * 6 + j] = dot_product(&Jm[(size_t)pnum * i], &Jm[(size_t)pnum * j], pnum, prod, b0, b1);
            JtF[i] = dot_product(&Jm[(size_t)pnum * i], fm.data(), pnum, prod, b0, b1);
        }
        for (int i = 0; i < 2; i++) {                                             // sign fix-ups :484-490
            JtF[i + 2] = -JtF[i + 2];
            for (int j = 0; j < 2; j++) {
                JtJ[(i + 0) * 6 + j + 2] = -JtJ[(i + 0) * 6 + j + 2];
                JtJ[(i + 2) * 6 + j + 4] = -JtJ[(i + 2) * 6 + j + 4];
            }
        }
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) JtJ[j * 6 + i] = JtJ[i * 6 + j];
    } else {
        for (int ikl = 0; ikl < klist.kn; ikl++) {
            const double qvel = (c.zfm * dfx[ikl] * Vt[0] + c.zfm * dfy[ikl] * Vt[1] +
                                 (PtIm[ikl] * dfx[ikl] + PtIm[pnum + ikl] * dfy[ikl]) * Vt[2]);
            double q_rho = std::sqrt(klist.kl[ikl].s_rho * qvel * klist.kl[ikl].s_rho * qvel + 1);
            if (!ReWeight) q_rho = klist.kl[ikl].s_rho;
            fm[ikl] /= q_rho;
        }
        for (int ikl = klist.kn; ikl < pnum; ikl++) fm[ikl] = 0;
    }
    return dot_product(fm.data(), fm.data(), pnum, prod, b0, b1);
}

// ---- global_tracker::Minimizer_RV<double,false> (global_tracker.cpp:580-819) -------------------------------------------
double minimizer_rv(Ctx &c, Slot &gt, Slot &klist, double Vel[3], double W0[3], double RVel[9], double RW0[9], double match_thresh,
                    int iter_max, int init_type, double reweigth_distance, double &rel_error, double &rel_error_score,
                    double max_s_rho, unsigned MatchNumThresh, double init_iter, double W_X[36]) {
    if (klist.kn <= 0) return 0;
    double JtJ[36], ApI[36], JtJnew[36], JtF[6], JtFnew[6], h[6] = {0, 0, 0, 0, 0, 0}, Xnew[6], X[6], Xt[6], L[36], nb[6];
    const int pnum = (klist.kn + 0x3) & (~0x3);
    std::vector<double> P0m;
    kl_to_p0(c, klist, pnum, P0m);
    std::vector<double> Res0(pnum, 0.0), Res1(pnum, 0.0), Rest(pnum, 0.0);
    double *Residual = Res0.data(), *ResidualNew = Res1.data();
    double F = 0, Fnew, F0 = 0;
    double v = 2, tau = 1e-3, u = 0, gain;
    int eff_steps = 0;
    const double k_hubber = reweigth_distance;
    auto eval = [&](bool rw, bool jf, double *JJ, double *JF, const double *Xs, double *rout) {
        return try_velrot(c, gt, klist, rw, jf, JJ, JF, Xs, P0m.data(), pnum, match_thresh, max_s_rho, MatchNumThresh, k_hubber,
                          Residual, rout);
    };
    auto max36 = [](const double *M) { double m = M[0]; for (int i = 1; i < 36; i++) m = M[i] > m ? M[i] : m; return m; };
    auto lm_update = [&](bool swap_res) {
        if (gain > 0) {
            F = Fnew;
            for (int i = 0; i < 6; i++) { X[i] = Xnew[i]; JtF[i] = JtFnew[i]; }
            for (int i = 0; i < 36; i++) JtJ[i] = JtJnew[i];
            const double g = 2 * gain - 1;
            u *= std::max(0.33, 1 - (g * g * g));
            v = 2;
            eff_steps++;
            if (swap_res) std::swap(ResidualNew, Residual);
        } else {
            u *= v;
            v *= 2;
        }
    };
    auto gain_ratio = [&]() {                                                      // (F-Fnew)/(0.5*h*(u*h-JtF))
        double den = 0;
        for (int i = 0; i < 6; i++) den += (0.5 * h[i]) * (u * h[i] - JtF[i]);
        return (F - Fnew) / den;
    };
    auto init_trial = [&](double *rout) {                                          // one arm of init_type 2 (:647-731)
        F = eval(false, true, JtJ, JtF, X, rout);
        F0 = F;
        u = tau * max36(JtJ);
        for (int i = 0; i < init_iter; i++) {
            for (int k = 0; k < 36; k++) ApI[k] = JtJ[k];
            for (int k = 0; k < 6; k++) { ApI[k * 7] = JtJ[k * 7] + 1.0 * u; nb[k] = -JtF[k]; }
            svd6_backsub(ApI, nb, h);
            for (int k = 0; k < 6; k++) Xnew[k] = X[k] + h[k];
            if (i == init_iter - 1) {
                Fnew = eval(false, false, JtJnew, JtFnew, Xnew, rout);
                gain = (F - Fnew);
            } else {
                Fnew = eval(false, true, JtJnew, JtFnew, Xnew, rout);
                gain = gain_ratio();
            }
            lm_update(false);
        }
    };
    switch (init_type) {
        case 0:
            for (int i = 0; i < 6; i++) X[i] = 0;
            break;
        case 1:
            for (int i = 0; i < 3; i++) { X[i] = Vel[i]; X[3 + i] = W0[i]; }
            break;
        case 2:
        default: {
            for (int i = 0; i < 6; i++) X[i] = 0;
            init_trial(Rest.data());
            for (int i = 0; i < 6; i++) Xt[i] = X[i];
            const double Ft = F, F0t = F0, ut = u, vt = v;
            const int eff_steps_t = eff_steps;
            eff_steps = 0;
            for (int i = 0; i < 3; i++) { X[i] = Vel[i]; X[3 + i] = W0[i]; }
            v = 2;
            // note: the reference sets v=2 after the first evaluation of this arm; nothing reads v in between
            init_trial(ResidualNew);
            if (F > Ft) {
                for (int i = 0; i < 6; i++) X[i] = Xt[i];
                F = Ft; F0 = F0t; u = ut; v = vt; eff_steps = eff_steps_t;
                ResidualNew = Rest.data();
            }
            std::swap(ResidualNew, Residual);
            break;
        }
    }
    F0 = F = eval(true, true, JtJ, JtF, X, ResidualNew);                          // :756
    u = tau * max36(JtJ);
    v = 2;
    for (int lm_iter = 0; lm_iter < iter_max; lm_iter++) {
        for (int k = 0; k < 36; k++) ApI[k] = JtJ[k];
        for (int k = 0; k < 6; k++) { ApI[k * 7] = JtJ[k * 7] + 1.0 * u; nb[k] = -JtF[k]; }
        chol6(ApI, L);
        chol6_backsub(L, nb, h);
        for (int k = 0; k < 6; k++) Xnew[k] = X[k] + h[k];
        Fnew = eval(true, true, JtJnew, JtFnew, Xnew, ResidualNew);
        gain = gain_ratio();
        lm_update(true);
    }
    double Inv[36];
    chol6(JtJ, L);
    chol6_inverse(L, Inv);
    for (int i = 0; i < 3; i++) { Vel[i] = X[i]; W0[i] = X[3 + i]; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { RVel[i * 3 + j] = Inv[i * 6 + j]; RW0[i * 3 + j] = Inv[(i + 3) * 6 + j + 3]; }
    for (int i = 0; i < 36; i++) W_X[i] = JtJ[i];
    if (eff_steps > 0) {
        double nh = 0, nx = 0;
        for (int i = 0; i < 6; i++) { nh += h[i] * h[i]; nx += X[i] * X[i]; }
        rel_error = std::sqrt(nh) / (std::sqrt(nx) + 1e-30);
        rel_error_score = F / F0;
    } else {
        rel_error = 1e20;
        rel_error_score = 1e20;
    }
    gt.FrameCount++;
    return F;
}

// ---- util::Matrix3x3Inv (include/UtilLib/toon_util.h:32-41) with TooN::determinant = Gaussian elimination with
// partial pivoting (TooN/determinant.h:91-146; TOON_DETERMINANT_LAPACK is not set in the vendored config) ---------
static double det3_ge(const double Ain[9]) {
    double A[9];
    for (int i = 0; i < 9; i++) A[i] = Ain[i];
    double det = 1;
    for (int i = 0; i < 3; i++) {
        int argmax = i;
        double maxval = std::fabs(A[i * 3 + i]);
        for (int ii = i + 1; ii < 3; ii++) {
            const double v = std::fabs(A[ii * 3 + i]);
            if (v > maxval) { maxval = v; argmax = ii; }
        }
        const double pivot = A[argmax * 3 + i];
        if (argmax != i) {
            det *= -1;
            for (int j = i; j < 3; j++) std::swap(A[i * 3 + j], A[argmax * 3 + j]);
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
static void mat3_inv(const double A[9], double B[9]) {
    B[0] = A[8] * A[4] - A[7] * A[5]; B[1] = -(A[8] * A[1] - A[7] * A[2]); B[2] = A[5] * A[1] - A[4] * A[2];
    B[3] = -(A[8] * A[3] - A[6] * A[5]); B[4] = A[8] * A[0] - A[6] * A[2]; B[5] = -(A[5] * A[0] - A[3] * A[2]);
    B[6] = A[7] * A[3] - A[6] * A[4]; B[7] = -(A[7] * A[0] - A[6] * A[1]); B[8] = A[4] * A[0] - A[3] * A[1];
    const double det = det3_ge(A);
    for (int i = 0; i < 9; i++) B[i] = B[i] / det;
}

// ---- global_tracker::TryVel<double> + Calc_f_J (global_tracker.cpp:830-934, 178-219) -----------------------------------
static double try_vel(Ctx &c, Slot &gt, Slot &klist, double JtJ[9], double JtF[3], const double Vel[3], double match_thresh,
                      double s_rho_min, unsigned MatchNumThresh, double *Residuals, double reweigth_distance, float min_mod) {
    const int w = c.p.w, h = c.p.h;
    const Slot &fl = c.slots[gt.field_slot];
    double score = 0, f;
    for (int i = 0; i < 9; i++) JtJ[i] = 0;
    for (int i = 0; i < 3; i++) JtF[i] = 0;
    double fi = 0;
    const double max_r = gt.max_r;
    const unsigned mthr = std::min(MatchNumThresh, gt.FrameCount);
    for (int ikl = 0; ikl < klist.kn; ikl++) {
        OrcKeyLine &kl = klist.kl[ikl];
        kl.m_id_f = -1;
        if (min_mod > 0 && kl.n_m < min_mod) continue;
        if (kl.s_rho > s_rho_min || (unsigned)kl.m_num < mthr) continue;
        double weight = 1;
        if (Residuals[ikl] > reweigth_distance) weight = reweigth_distance / Residuals[ikl];
        const double z_p = 1.0 / kl.rho + Vel[2];
        if (z_p <= 0) {
            f = (1 / (kl.s_rho)) * max_r * weight;
            score += f * f;
            continue;
        }
        const double rho_p = 1.0 / z_p;
        const double pjx = rho_p * (Vel[0] * c.zfm - Vel[2] * kl.p_m[0]) + kl.p_m[0];
        const double pjy = rho_p * (Vel[1] * c.zfm - Vel[2] * kl.p_m[1]) + kl.p_m[1];
        const double pix = pjx + c.ppx, piy = pjy + c.ppy;                      // cam_model::Hom2Img
        const int x = (int)(pix + 0.5), y = (int)(piy + 0.5);
        if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) {
            f = (1 / (kl.s_rho)) * max_r * weight;
            score += f * f;
            continue;
        }
        double df_dx = 0, df_dy = 0;
        f = max_r / kl.s_rho;                                                    // Calc_f_J
        const int fikl = gt.field[2 * ((size_t)y * w + x) + 1];
        if (fikl >= 0) {
            const OrcKeyLine &f_kl = fl.kl[fikl];
            const double p_n2 = (kl.n_m * kl.n_m);
            const double p_esc = kl.m_m[0] * f_kl.m_m[0] + kl.m_m[1] * f_kl.m_m[1];
            if (!(std::fabs(p_esc - p_n2) > match_thresh * p_n2)) {
                const double dx = pix - f_kl.c_p[0], dy = piy - f_kl.c_p[1];
                fi = (dx * f_kl.u_m[0] + dy * f_kl.u_m[1]);
                df_dx = f_kl.u_m[0] / kl.s_rho;
                df_dy = f_kl.u_m[1] / kl.s_rho;
                kl.m_id_f = fikl;
                f = fi / kl.s_rho;
            }
        }
        f *= weight;
        score += f * f;
        const double jx = rho_p * c.zfm * df_dx * weight;
        const double jy = rho_p * c.zfm * df_dy * weight;
        const double jz = -rho_p * (pjx * df_dx + pjy * df_dy) * weight;
        JtJ[0] += jx * jx; JtJ[4] += jy * jy; JtJ[8] += jz * jz;
        JtJ[1] += jx * jy; JtJ[2] += jx * jz; JtJ[5] += jy * jz;
        JtF[0] += jx * f; JtF[1] += jy * f; JtF[2] += jz * f;
        Residuals[ikl] = std::fabs(fi);
    }
    JtJ[3] = JtJ[1]; JtJ[6] = JtJ[2]; JtJ[7] = JtJ[5];
    return score;
}

// ---- global_tracker::Minimizer_V<double> (global_tracker.cpp:1037-1093) -------------------------------------------------
double minimizer_v(Ctx &c, Slot &gt, Slot &klist, double Vel[3], double RVel[9], double match_thresh, int iter_max,
                   double s_rho_min, unsigned MatchNumThresh, double reweigth_distance, float min_mod) {
    double JtJ[9], ApI[9], JtJnew[9], JtF[3], JtFnew[3], h[3], Vnew[3], Inv[9];
    std::vector<double> residuals(std::max(klist.kn, 1), 0.0);
    double F = try_vel(c, gt, klist, JtJ, JtF, Vel, match_thresh, s_rho_min, MatchNumThresh, residuals.data(), reweigth_distance, min_mod), Fnew;
    double v = 2, tau = 1e-3;
    double mx = JtJ[0];
    for (int i = 1; i < 9; i++) mx = JtJ[i] > mx ? JtJ[i] : mx;
    double u = tau * mx, gain;
    for (int lm_iter = 0; lm_iter < iter_max; lm_iter++) {
        for (int i = 0; i < 9; i++) ApI[i] = JtJ[i] + ((i % 4 == 0) ? 1.0 * u : 0.0);
        mat3_inv(ApI, Inv);
        for (int i = 0; i < 3; i++) {
            double d = 0;
            for (int k = 0; k < 3; k++) d += Inv[i * 3 + k] * (-JtF[k]);
            h[i] = d;
            Vnew[i] = Vel[i] + h[i];
        }
        Fnew = try_vel(c, gt, klist, JtJnew, JtFnew, Vnew, match_thresh, s_rho_min, MatchNumThresh, residuals.data(), reweigth_distance, min_mod);
        double den = 0;
        for (int i = 0; i < 3; i++) den += (0.5 * h[i]) * (u * h[i] - JtF[i]);
        gain = (F - Fnew) / den;
        if (gain > 0) {
            F = Fnew;
            for (int i = 0; i < 3; i++) { Vel[i] = Vnew[i]; JtF[i] = JtFnew[i]; }
            for (int i = 0; i < 9; i++) JtJ[i] = JtJnew[i];
            const double g = 2 * gain - 1;
            u *= std::max(0.33, 1 - (g * g * g));
            v = 2;
        } else {
            u *= v;
            v *= 2;
        }
    }
    mat3_inv(JtJ, RVel);
    return F;
}

}  // namespace port
