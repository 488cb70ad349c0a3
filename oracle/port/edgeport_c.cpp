// edgeport_c.cpp — CPU restatement, stage C (matching + mapping) and the C ABI (port_* of ../oracle_abi.h)
// including the per-frame sequencing of FirstThr + SecondThread for ImuMode==0.
// TEST INFRASTRUCTURE ONLY (see edgeport_a.cpp for the rules).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "edgeport.h"
#include "port_math.h"

namespace port {

// ---- edge_tracker::FordwardMatch (src/mtracklib/edge_tracker.cpp:380-436) ---------------------------------------
int forward_match(Slot &from, Slot &et) {
    double nmatch = 0;
    for (int ikl = 0; ikl < from.kn; ikl++) {
        const OrcKeyLine &k = from.kl[ikl];
        const int ikl_f = k.m_id_f;
        if (ikl_f < 0) continue;
        if (ikl_f >= et.kn) continue;                                   // "this should never happen"
        OrcKeyLine &e = et.kl[ikl_f];
        if (e.m_id >= 0 && e.rho > k.rho) continue;                     // double match: keep the KeyLine in front
        e.rho = k.rho; e.s_rho = k.s_rho;
        e.rho_nr = k.rho_nr; e.s_rho_nr = k.s_rho_nr;
        e.m_num = k.m_num + 1;
        e.m_id = ikl;
        e.p_m_0[0] = k.p_m[0]; e.p_m_0[1] = k.p_m[1];
        e.m_m0[0] = k.m_m[0]; e.m_m0[1] = k.m_m[1];
        e.n_m0 = k.n_m;
        e.m_id_kf = k.m_id_kf;
        nmatch++;
    }
    et.nmatch = (int)nmatch;
    return (int)nmatch;
}

// ---- edge_tracker::rotate_keylines (edge_tracker.cpp:42-76) -------------------------------------------------------
void rotate_keylines(const Ctx &c, Slot &s, const double RotF[9]) {
    const double zf = c.zfm;
    for (int i = 0; i < s.kn; i++) {
        OrcKeyLine &k = s.kl[i];
        const double v[3] = {k.p_m[0] / zf, k.p_m[1] / zf, 1};
        double q[3];
        mat3_vec(RotF, v, q);
        if (std::fabs(q[2]) > 0) {
            k.p_m[0] = q[0] / q[2] * zf;
            k.p_m[1] = q[1] / q[2] * zf;
            k.rho /= q[2];
            k.s_rho = k.s_rho / q[2];
        }
        const double m[3] = {k.m_m[0], k.m_m[1], 0};
        mat3_vec(RotF, m, q);
        k.m_m[0] = q[0];
        k.m_m[1] = q[1];
    }
}

// ---- edge_tracker::search_match (edge_tracker.cpp:158-295): walk the epipolar segment on et0's mask ----------------
static int search_match(const Ctx &c, const Slot &et0, const OrcKeyLine &k, const double Vel[3], const double RVel[9],
                        const double BackRot[9], double min_thr_mod, double min_thr_ang, double max_radius, double loc_uncertainty) {
    const int w = c.p.w, h = c.p.h;
    const double cang_min_edge = std::cos(min_thr_ang * M_PI / 180.0);
    double dq_min = 0, dq_max = 0, t_x = 0, t_y = 0, dq_rho = 0;
    int t_steps = 0;
    const double pin[3] = {k.p_m[0], k.p_m[1], c.zfm};
    double p_m3[3];
    mat3_vec(BackRot, pin, p_m3);
    const float pmx = p_m3[0] * c.zfm / p_m3[2], pmy = p_m3[1] * c.zfm / p_m3[2];
    const double k_rho = k.rho * c.zfm / p_m3[2];
    const float pi0x = pmx + c.ppx, pi0y = pmy + c.ppy;                 // cam_model::Hom2Img on Point2DF
    t_x = -(Vel[0] * c.zfm - Vel[2] * pmx);
    t_y = -(Vel[1] * c.zfm - Vel[2] * pmy);
    double norm_t = std::sqrt(t_x * t_x + t_y * t_y);
    const double DrDv[3] = {c.zfm, c.zfm, -pmx - pmy};
    double row[3];
    for (int j = 0; j < 3; j++) {
        double d = 0;
        for (int i = 0; i < 3; i++) d += DrDv[i] * RVel[i * 3 + j];
        row[j] = d;
    }
    double sigma2_t = 0;
    for (int j = 0; j < 3; j++) sigma2_t += row[j] * DrDv[j];
    if (norm_t > 1e-6) {
        t_x /= norm_t;
        t_y /= norm_t;
        dq_rho = norm_t * k_rho;
        dq_min = std::max(0.0, norm_t * (k_rho - k.s_rho)) - loc_uncertainty;
        dq_max = std::min(max_radius, norm_t * (k_rho + k.s_rho)) + loc_uncertainty;
        if (dq_rho > dq_max) {
            dq_rho = (dq_max + dq_min) / 2;
            t_steps = (int)(dq_rho + 0.5);
        } else {
            t_steps = (int)(std::max(dq_max - dq_rho, dq_rho - dq_min) + 0.5);
        }
    } else {
        t_x = k.m_m[0];
        t_y = k.m_m[1];
        norm_t = k.n_m;
        t_x /= norm_t;
        t_y /= norm_t;
        norm_t = 1;
        dq_min = -max_radius - loc_uncertainty;
        dq_max = max_radius + loc_uncertainty;
        dq_rho = 0;
        t_steps = dq_max;
    }
    const double norm_m = k.n_m;
    double tn = dq_rho, tp = dq_rho + 1;
    for (int t_i = 0; t_i < t_steps; t_i++, tp += 1, tn -= 1) {
        for (int i_inx = 0; i_inx < 2; i_inx++) {
            double t;
            if (i_inx) { t = tp; if (t > dq_max) continue; }
            else       { t = tn; if (t < dq_min) continue; }
            const float fx = t_x * t + pi0x, fy = t_y * t + pi0y;      // GetIndexRC(const float, const float)
            const int xi = (int)std::round(fx), yi = (int)std::round(fy);
            if (xi >= w || yi >= h || xi < 0 || yi < 0) continue;
            const int j = et0.mask[(size_t)yi * w + xi];
            if (j < 0) continue;
            const OrcKeyLine &o = et0.kl[j];
            const double norm_m0 = o.n_m;
            const double cang = (o.m_m[0] * k.m_m[0] + o.m_m[1] * k.m_m[1]) / (norm_m0 * norm_m);
            if (cang < cang_min_edge || std::fabs(norm_m0 / norm_m - 1) > min_thr_mod) continue;
            const double v_rho_dr = (loc_uncertainty * loc_uncertainty + o.s_rho * o.s_rho * norm_t * norm_t + sigma2_t * o.rho * o.rho);
            const double dd = t - norm_t * o.rho;
            if (dd * dd > v_rho_dr) continue;
            return j;
        }
    }
    return -1;
}

// ---- edge_tracker::directed_matching (edge_tracker.cpp:302-374), stereo_mode = clear = false ---------------------------
int directed_matching(const Ctx &c, Slot &s, const double Vel_in[3], const double RVel_in[9], const double BackRot[9], Slot &et0,
                      int &kf_matchs, double min_thr_mod, double min_thr_ang, double max_radius, double loc_uncertainty) {
    s.nmatch = 0;
    kf_matchs = 0;
    double Vel[3], T[9], RVel[9], BRt[9];
    mat3_vec(BackRot, Vel_in, Vel);                                      // Vel = BackRot*Vel
    mat3_mul(BackRot, RVel_in, T);                                       // RVel = BackRot*RVel*BackRot.T()
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) BRt[i * 3 + j] = BackRot[j * 3 + i];
    mat3_mul(T, BRt, RVel);
    for (int i_kn = 0; i_kn < s.kn; i_kn++) {
        OrcKeyLine &k = s.kl[i_kn];
        const int i_mch = search_match(c, et0, k, Vel, RVel, BackRot, min_thr_mod, min_thr_ang, max_radius, loc_uncertainty);
        if (i_mch < 0) continue;
        const OrcKeyLine &o = et0.kl[i_mch];
        k.rho = o.rho; k.s_rho = o.s_rho; k.rho_nr = o.rho_nr; k.s_rho_nr = o.s_rho_nr;
        k.m_id = i_mch;
        k.m_num = o.m_num + 1;
        k.p_m_0[0] = o.p_m[0]; k.p_m_0[1] = o.p_m[1];
        k.m_m0[0] = o.m_m[0]; k.m_m0[1] = o.m_m[1];
        k.n_m0 = o.n_m;
        k.m_id_kf = o.m_id_kf;
        if (k.m_id_kf >= 0) kf_matchs++;
        s.nmatch++;
    }
    return s.nmatch;
}

// ---- edge_tracker::Regularize_1_iter (edge_tracker.cpp:87-148): compute all, then write ----------------------------------
int regularize_1_iter(Slot &s, double thresh) {
    int r_num = 0;
    std::vector<double> r(s.kn), sr(s.kn);
    std::vector<char> set(s.kn, 0);
    for (int ikl = 0; ikl < s.kn; ikl++) {
        if (s.kl[ikl].n_id < 0 || s.kl[ikl].p_id < 0) continue;
        const OrcKeyLine &k = s.kl[ikl], &kn = s.kl[k.n_id], &kp = s.kl[k.p_id];
        const double d = kn.rho - kp.rho;
        if (d * d > kn.s_rho * kn.s_rho + kp.s_rho * kp.s_rho) continue;       // util::square vs util::norm2
        double alpha = (kn.m_m[0] * kp.m_m[0] + kn.m_m[1] * kp.m_m[1]) / (kn.n_m * kp.n_m);   // float expression (:119)
        if (alpha - thresh < 0) continue;
        alpha = (alpha - thresh) / (1 - thresh);
        alpha /= std::fabs(kn.rho - kp.rho) / (kn.s_rho + kp.s_rho) + 1;
        const double wr = 1 / (k.s_rho * k.s_rho), wrn = alpha / (kn.s_rho * kn.s_rho), wrp = alpha / (kp.s_rho * kp.s_rho);
        r[ikl] = (k.rho * wr + kn.rho * wrn + kp.rho * wrp) / (wr + wrn + wrp);
        sr[ikl] = (k.s_rho * wr + kn.s_rho * wrn + kp.s_rho * wrp) / (wr + wrn + wrp);
        set[ikl] = 1;
        r_num++;
    }
    for (int ikl = 0; ikl < s.kn; ikl++)
        if (set[ikl]) { s.kl[ikl].rho = r[ikl]; s.kl[ikl].s_rho = sr[ikl]; }
    return r_num;
}

// ---- edge_tracker::UpdateInverseDepthKalman -> UpdateInverseDepthKalmanARLU (edge_tracker.cpp:695-724, 954-1055) --------
void update_inverse_depth_kalman(const Ctx &c, Slot &s, const double vel[3], double ReshapeQAbsolute, double LocationUncertainty) {
    const double zf = c.zfm;
    for (int i = 0; i < s.kn; i++) {
        OrcKeyLine &kli = s.kl[i];
        if (kli.m_id < 0) continue;
        double &rho = kli.rho, &s_rho = kli.s_rho;
        kli.s_rho0 = s_rho;
        const double qx = kli.p_m[0], qy = kli.p_m[1], q0x = kli.p_m_0[0], q0y = kli.p_m_0[1];
        double v_rho = s_rho * s_rho;
        const double u_x = kli.m_m0[0] / kli.n_m0, u_y = kli.m_m0[1] / kli.n_m0;
        const double Y = u_x * (qx - q0x) + u_y * (qy - q0y);
        const double H = u_x * (vel[0] * zf - vel[2] * q0x) + u_y * (vel[1] * zf - vel[2] * q0y);
        const double rho_p = 1 / (1.0 / rho + vel[2]);
        kli.rho0 = rho_p;
        double F = 1 / (1 + rho * vel[2]);
        F = F * F;
        const double p_p = F * v_rho * F + ReshapeQAbsolute * ReshapeQAbsolute;
        const double e = Y - H * rho_p;
        const double S = H * p_p * H + LocationUncertainty * LocationUncertainty;
        const double K = p_p * H * (1 / S);
        rho = rho_p + (K * e);
        v_rho = (1 - K * H) * p_p;
        s_rho = std::sqrt(v_rho);
        if (rho < kRhoMin) {
            s_rho += kRhoMin - rho;
            rho = kRhoMin;
        } else if (rho > kRhoMax) {
            rho = kRhoMax;
        } else if (std::isnan(rho) || std::isnan(s_rho) || std::isinf(rho) || std::isinf(s_rho)) {
            rho = kRhoInit;
            s_rho = kRhoMax;
        } else if (s_rho < 0) {
            rho = kRhoInit;
            s_rho = kRhoMax;
        }
    }
}

// ---- edge_tracker::ExtRotVel (edge_tracker.cpp:1207-1296) ------------------------------------------------------------------
bool ext_rot_vel(const Ctx &c, Slot &s, const double vel[3], double Wx[36], double Rx[36], double X[6], double LocUncert, double HubReweigth) {
    const double zf = c.zfm;
    double JtJ[36], JtF[6];
    for (int i = 0; i < 36; i++) JtJ[i] = 0;
    for (int i = 0; i < 6; i++) JtF[i] = 0;
    for (int i = 0; i < s.kn; i++) {
        const OrcKeyLine &k = s.kl[i];
        if (k.m_id < 0) continue;
        const float u_x = k.u_m[0], u_y = k.u_m[1];
        const double rho_t = 1 / (1 / k.rho + vel[2]);
        const float qt_x = k.p_m_0[0] + rho_t * (vel[0] * zf - vel[2] * k.p_m_0[0]);
        const float qt_y = k.p_m_0[1] + rho_t * (vel[1] * zf - vel[2] * k.p_m_0[1]);
        const float q_x = k.p_m[0], q_y = k.p_m[1];
        double phi[6];
        phi[0] = u_x * rho_t * zf;
        phi[1] = u_y * rho_t * zf;
        phi[2] = u_x * (-rho_t * q_x) + u_y * (-rho_t * q_y);
        phi[3] = -u_x * q_x * q_y / zf - u_y * (zf + q_y * q_y / zf);
        phi[4] = +u_y * q_x * q_y / zf + u_x * (zf + q_x * q_x / zf);
        phi[5] = -u_x * q_y + u_y * q_x;
        double y = u_x * (k.p_m[0] - qt_x) + u_y * (k.p_m[1] - qt_y);
        const float dqvel = u_x * (vel[0] * zf - vel[2] * k.p_m_0[0]) + u_y * (vel[1] * zf - vel[2] * k.p_m_0[1]);
        const float s_y = std::sqrt(k.s_rho * k.s_rho * dqvel * dqvel + LocUncert * LocUncert);
        double weigth = 1;
        if (std::fabs(y) > HubReweigth) weigth = std::fabs(y) / HubReweigth;
        for (int a = 0; a < 6; a++) phi[a] /= s_y * weigth;
        y /= s_y * weigth;
        for (int a = 0; a < 6; a++) {                      // Phi.T()*Phi and Phi.T()*Y, row by row
            for (int b = 0; b < 6; b++) JtJ[a * 6 + b] += phi[a] * phi[b];
            JtF[a] += phi[a] * y;
        }
    }
    svd6_solve_pinv(JtJ, JtF, X, Rx);
    for (int i = 0; i < 36; i++) Wx[i] = JtJ[i];
    for (int i = 0; i < 36; i++) if (std::isnan(Rx[i])) return false;
    for (int i = 0; i < 6; i++) if (std::isnan(X[i])) return false;
    return true;
}

// ---- edge_tracker::EstimateReScalingOpt (edge_tracker.cpp:1104-1140) ----------------------------------------------------
double estimate_rescaling_opt(Slot &s, double &RKp, double s_rho_min, unsigned MatchNumMin, bool re_escale) {
    if (s.kn <= 0) return 1;
    double Kp = 1;
    for (int iter = 0; iter < 5; iter++) {
        double rTr = 0, rTr0 = 0;
        for (int ikl = 0; ikl < s.kn; ikl++) {
            const OrcKeyLine &k = s.kl[ikl];
            if ((unsigned)k.m_num < MatchNumMin || k.s_rho0 <= 0 || k.s_rho > s_rho_min) continue;
            rTr += k.rho * k.rho / (k.s_rho * k.s_rho + Kp * Kp * k.s_rho0 * k.s_rho0);
            rTr0 += k.rho0 * k.rho0 / (k.s_rho * k.s_rho + Kp * Kp * k.s_rho0 * k.s_rho0);
        }
        Kp = rTr0 > 0 ? std::sqrt(rTr / rTr0) : 1;
        RKp = 1 / rTr0;
    }
    if (re_escale)
        for (int ikl = 0; ikl < s.kn; ikl++) { s.kl[ikl].rho /= Kp; s.kl[ikl].s_rho /= Kp; }
    return Kp;
}

static void reset_seq(Ctx *c) {
    c->tresh = c->p.detector_thresh;                 // rebvo_first_t.cpp:94
    c->l_kl_num = 0;
    c->frame = 0;
    c->t_prev = 0;
    c->Kp = 1; c->K = 1; c->P_Kp = 5e-6;             // rebvo_second_t.cpp:54, 65
    for (int i = 0; i < 3; i++) { c->V[i] = 0; c->W[i] = 0; c->Pos[i] = 0; }
    for (int i = 0; i < 9; i++) c->Pose[i] = (i % 4 == 0) ? 1 : 0;
    for (Slot &s : c->slots) s.FrameCount = 0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace port

using namespace port;

extern "C" {

void *port_create(const OrcParams *p, int nslots) {
    Ctx *c = new Ctx;
    c->p = *p;
    c->ppx = (float)p->ppx; c->ppy = (float)p->ppy; c->zfx = (float)p->zfx; c->zfy = (float)p->zfy;
    c->zfm = (double)((c->zfx + c->zfy) / 2);
    const int w = p->w, h = p->h;
    const size_t n = (size_t)w * h;
    c->filter[0].sigma_r = kovesi_boxes(p->sigma0, kBoxes, c->filter[0].box_d);                       // sspace.cpp:45
    c->filter[1].sigma_r = kovesi_boxes(c->filter[0].sigma_r * p->ksigma, kBoxes, c->filter[1].box_d);
    for (int f = 0; f < 2; f++)
        for (int i = 0; i < kBoxes; i++) build_average(c->filter[f].box_d[i], w, h, c->filter[f].div[i]);
    c->integral.assign(n, 0.f);
    plane_fit_pinv(p->plane_fit_size, c->pinv);
    if (p->use_undistort) build_undistort_map(*c);
    c->slots.resize(nslots);
    const int cap = std::min(p->max_points, 50000);                      // KEYLINE_MAX (edge_finder.h:43)
    for (Slot &s : c->slots) {
        s.imgc.assign(n * 3, 0);
        s.bw.assign(n, 0.f); s.img0.assign(n, 0.f); s.img1.assign(n, 0.f); s.dog.assign(n, 0.f); s.dx.assign(n, 0.f); s.dy.assign(n, 0.f);
        s.mask.assign(n, -1);
        s.kl.resize(std::max(cap, 1));
        s.field.assign(n * 2, 0);
    }
    reset_seq(c);
    return c;
}
void port_destroy(void *ctx) { delete (Ctx *)ctx; }
void port_reset_sequence(void *ctx) { reset_seq((Ctx *)ctx); }
void port_get_seq_state(void *ctx, OrcSeqState *o) {
    Ctx *c = (Ctx *)ctx;
    o->tresh = c->tresh; o->t_prev = c->t_prev; o->Kp = c->Kp; o->K = c->K; o->P_Kp = c->P_Kp;
    for (int i = 0; i < 3; i++) { o->V[i] = c->V[i]; o->W[i] = c->W[i]; o->Pos[i] = c->Pos[i]; }
    for (int i = 0; i < 9; i++) o->Pose[i] = c->Pose[i];
    o->l_kl_num = c->l_kl_num; o->frame = c->frame;
}
void port_svd_trace(OrcSvdRec *buf, int cap) { svd_trace_set(buf, cap); }
int port_svd_trace_count(void) { return svd_trace_count(); }
void port_depth_reset(void *ctx) {   // rebvo_second_t.cpp:609-620: depth reset of the newest edge map, pose and velocity reset
    Ctx *c = (Ctx *)ctx;
    if (c->frame == 0) return;
    Slot &nb = c->slots[(c->frame + (int)c->slots.size() - 1) % (int)c->slots.size()];
    for (int i = 0; i < nb.kn; i++) { nb.kl[i].rho = kRhoInit; nb.kl[i].s_rho = kRhoMax; }
    for (int i = 0; i < 9; i++) c->Pose[i] = (i % 4 == 0) ? 1 : 0;
    for (int i = 0; i < 3; i++) { c->Pos[i] = 0; c->W[i] = 0; c->V[i] = 0; }
}
int port_cur_slot(void *ctx) {
    Ctx *c = (Ctx *)ctx;
    return (c->frame + (int)c->slots.size() - 1) % (int)c->slots.size();
}
int port_stage_a(void *ctx, int slot, const uint8_t *rgb24, double *tresh_io, int *l_kl_num_io) {
    return stage_a(*(Ctx *)ctx, slot, rgb24, tresh_io, l_kl_num_io);
}
const uint8_t *port_imgc(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].imgc.data(); }
int port_undistort_map(void *ctx, int32_t *inx, int32_t *iw) {
    Ctx *c = (Ctx *)ctx;
    if (c->umap.empty()) return -1;
    for (size_t i = 0; i < c->umap.size(); i++)
        for (int k = 0; k < 4; k++) {
            inx[i * 4 + k] = k < c->umap[i].num ? c->umap[i].inx[k] : -1;
            iw[i * 4 + k] = k < c->umap[i].num ? c->umap[i].iw[k] : 0;
        }
    return 0;
}
const float *port_plane(void *ctx, int slot, int which) {
    Slot &s = ((Ctx *)ctx)->slots[slot];
    switch (which) {
        case 0: return s.img0.data();
        case 1: return s.img1.data();
        case 2: return s.dog.data();
        case 3: return s.dx.data();
        case 4: return s.dy.data();
        default: return s.bw.data();
    }
}
const int32_t *port_mask(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].mask.data(); }
int port_kn(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].kn; }
OrcKeyLine *port_keylines(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].kl.data(); }
float port_retuned(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].retuned; }
void port_set_keylines(void *ctx, int slot, const OrcKeyLine *kl, int kn, const int32_t *mask, float retuned) {
    Ctx *c = (Ctx *)ctx;
    Slot &s = c->slots[slot];
    memcpy(s.kl.data(), kl, sizeof(OrcKeyLine) * kn);
    s.kn = kn;
    if (mask) memcpy(s.mask.data(), mask, sizeof(int32_t) * c->p.w * c->p.h);
    s.retuned = retuned;
}
unsigned port_get_framecount(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].FrameCount; }
void port_set_framecount(void *ctx, int slot, unsigned fc) { ((Ctx *)ctx)->slots[slot].FrameCount = fc; }

double port_quantile(void *ctx, int slot, double smin, double smax, double pct, int n) {
    return estimate_quantile(((Ctx *)ctx)->slots[slot], smin, smax, pct, n);
}
void port_build_field(void *ctx, int slot, int radius, float min_mod) { build_field(*(Ctx *)ctx, slot, radius, min_mod); }
const int32_t *port_field(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].field.data(); }

double port_try_velrot(void *ctx, int slot_new, int slot_old, const double X[6], int reweight, int procjf, double match_thresh,
                       double s_rho_min, unsigned match_num_thresh, double k_huber, const double *resid_in, double *resid_out,
                       double JtJ[36], double JtF[6]) {
    Ctx *c = (Ctx *)ctx;
    Slot &klist = c->slots[slot_old];
    const int kn = klist.kn, pnum = (kn + 0x3) & (~0x3);
    std::vector<double> P0m, rin(pnum, 0.0), rout(pnum, 0.0);
    kl_to_p0(*c, klist, pnum, P0m);
    if (resid_in) memcpy(rin.data(), resid_in, sizeof(double) * kn);
    if (resid_out) memcpy(rout.data(), resid_out, sizeof(double) * kn);
    for (int i = 0; i < 36; i++) JtJ[i] = 0;
    for (int i = 0; i < 6; i++) JtF[i] = 0;
    const double F = try_velrot(*c, c->slots[slot_new], klist, reweight != 0, procjf != 0, JtJ, JtF, X, P0m.data(), pnum, match_thresh,
                                s_rho_min, match_num_thresh, k_huber, rin.data(), rout.data());
    if (resid_out) memcpy(resid_out, rout.data(), sizeof(double) * kn);
    return F;
}
double port_minimizer_rv(void *ctx, int slot_new, int slot_old, double V[3], double W[3], double RVel[9], double RW0[9],
                         double match_thresh, int iter_max, int init_type, double reweight_distance, double *rel_error,
                         double *rel_error_score, double max_s_rho, unsigned match_num_thresh, double init_iter, double W_X[36]) {
    Ctx *c = (Ctx *)ctx;
    return minimizer_rv(*c, c->slots[slot_new], c->slots[slot_old], V, W, RVel, RW0, match_thresh, iter_max, init_type,
                        reweight_distance, *rel_error, *rel_error_score, max_s_rho, match_num_thresh, init_iter, W_X);
}
double port_minimizer_v(void *ctx, int slot_new, int slot_old, double V[3], double RVel[9], double match_thresh, int iter_max,
                        double s_rho_min, unsigned match_num_thresh, double reweight_distance, float min_mod) {
    Ctx *c = (Ctx *)ctx;
    return minimizer_v(*c, c->slots[slot_new], c->slots[slot_old], V, RVel, match_thresh, iter_max, s_rho_min, match_num_thresh,
                       reweight_distance, min_mod);
}
int port_ext_rot_vel(void *ctx, int slot, const double vel[3], double loc_unc, double hub_reweight, double X[6], double Wx[36], double Rx[36]) {
    Ctx *c = (Ctx *)ctx;
    return ext_rot_vel(*c, c->slots[slot], vel, Wx, Rx, X, loc_unc, hub_reweight);
}
int port_forward_match(void *ctx, int slot_old, int slot_new) {
    Ctx *c = (Ctx *)ctx;
    return forward_match(c->slots[slot_old], c->slots[slot_new]);
}
void port_rotate_keylines(void *ctx, int slot, const double R[9]) { rotate_keylines(*(Ctx *)ctx, ((Ctx *)ctx)->slots[slot], R); }
int port_directed_matching(void *ctx, int slot_new, int slot_old, const double V[3], const double RVel[9], const double BackRot[9],
                           int *kf_matchs, double min_thr_mod, double min_thr_ang, double max_radius, double loc_unc) {
    Ctx *c = (Ctx *)ctx;
    return directed_matching(*c, c->slots[slot_new], V, RVel, BackRot, c->slots[slot_old], *kf_matchs, min_thr_mod, min_thr_ang,
                             max_radius, loc_unc);
}
int port_regularize(void *ctx, int slot, double thresh) { return regularize_1_iter(((Ctx *)ctx)->slots[slot], thresh); }
void port_ekf(void *ctx, int slot, const double V[3], const double RVel[9], const double RW0[9], double q_abs, double q_rel,
              double loc_unc) {
    (void)RVel; (void)RW0; (void)q_rel;   // accepted but unused by the ARLU variant, as in the reference
    update_inverse_depth_kalman(*(Ctx *)ctx, ((Ctx *)ctx)->slots[slot], V, q_abs, loc_unc);
}
double port_rescale(void *ctx, int slot, double *RKp, double s_rho_min, unsigned match_num_min, int re_escale) {
    return estimate_rescaling_opt(((Ctx *)ctx)->slots[slot], *RKp, s_rho_min, match_num_min, re_escale != 0);
}

// One frame through FirstThr (rebvo_first_t.cpp:229-272) + SecondThread, ImuMode==0 (rebvo_second_t.cpp:128-629)
int port_process_frame(void *ctx, const uint8_t *rgb24, double t, OrcNav *nav) {
    Ctx *c = (Ctx *)ctx;
    const OrcParams &p = c->p;
    const int ns = (int)c->slots.size();
    const int sn = c->frame % ns, so = (c->frame + ns - 1) % ns;
    memset(nav, 0, sizeof(*nav));
    double t0 = now();
    stage_a(*c, sn, rgb24, &c->tresh, &c->l_kl_num);
    nav->dtp0 = now() - t0;
    Slot &nb = c->slots[sn];
    nav->frame = c->frame;
    nav->t = t;
    nav->kn = nb.kn;
    nav->tresh = c->tresh;
    nav->retuned_thresh = nb.retuned;
    if (c->frame == 0) {                                                 // dummy processing of the first frame (:108-121)
        c->frame++;
        c->t_prev = t;
        return 0;
    }
    Slot &ob = c->slots[so];
    double t1 = now();
    bool EstimationOk = true;
    double dt_frame = t - c->t_prev;                                     // :145-147
    if (dt_frame < 0.001) dt_frame = 1 / p.config_fps;
    int klm_num = 0, num_kf_back_m = 0;
    double P_V[9] = {1e50, 0, 0, 0, 1e50, 0, 0, 0, 1e50}, P_W[9] = {1e50, 0, 0, 0, 1e50, 0, 0, 0, 1e50}, R[9];   // :166-168
    double *V = c->V, *W = c->W;
    double error_vel = 0, error_score = 0, W_X[36];
    const double s_rho_q = estimate_quantile(ob, kRhoMin, kRhoMax, p.qcut_quantile, p.qcut_nbins);      // :172
    build_field(*c, sn, p.search_range, nb.retuned);                                                   // :177
    nav->score = minimizer_rv(*c, nb, ob, V, W, P_V, P_W, p.tracker_match_thresh, p.tracker_iter_num, p.tracker_init_type,
                              p.reweight_distance, error_vel, error_score, s_rho_q, p.match_num_thresh,
                              p.tracker_init_iter_num, W_X);                                             // :346
    nav->klm_fwd = forward_match(ob, nb);                                                              // :354
    double R0[9];
    so3_exp(W, R0);                                                                                    // :360
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = R0[j * 3 + i];                                        // R.T() = R0*I  (:361)
    rotate_keylines(*c, ob, R0);                                                                       // :369
    for (int i = 0; i < 3; i++) { nav->V[i] = V[i]; nav->W[i] = W[i]; }
    for (int i = 0; i < 9; i++) { nav->P_V[i] = P_V[i]; nav->P_W[i] = P_W[i]; }
    auto set_ident = [](double *M, double s) { for (int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? s : 0; };
    if (std::isnan(V[0]) || std::isnan(V[1]) || std::isnan(V[2]) || std::isnan(W[0]) || std::isnan(W[1]) || std::isnan(W[2])) {   // :387-397
        set_ident(P_V, 1e50);
        V[0] = V[1] = V[2] = 0;
        c->Kp = 1;
        c->P_Kp = 1e50;
        EstimationOk = false;
    } else {
        klm_num = directed_matching(*c, nb, V, P_V, R, ob, num_kf_back_m, p.match_thresh_module, p.match_thresh_angle,
                                    p.search_range, p.loc_unc_match);                                    // :410
        if (klm_num < p.global_match_threshold) {                                                      // :412-422
            set_ident(P_V, 1e50);
            V[0] = V[1] = V[2] = 0;
            c->Kp = 1;
            c->P_Kp = 10;
            EstimationOk = false;
        } else {
            regularize_1_iter(nb, p.regularize_thresh);                                                // :453
            update_inverse_depth_kalman(*c, nb, V, p.reshape_q_abs, p.loc_unc);                          // :460
            c->Kp = estimate_rescaling_opt(nb, c->P_Kp, kRhoMax, 1, p.do_rescaling > 0);                 // :487
        }
    }
    double P2[9];
    mat3_mul(c->Pose, R, P2);                                                                          // Pose = Pose*R (:550)
    for (int i = 0; i < 9; i++) c->Pose[i] = P2[i];
    for (int i = 0; i < 3; i++) {                                                                      // Pos += -Pose*V*K (:551)
        double d = 0;
        for (int k = 0; k < 3; k++) d += (-c->Pose[i * 3 + k]) * V[k];
        c->Pos[i] += d * c->K;
    }
    nav->dtp1 = now() - t1;
    nav->dt = dt_frame;
    nav->Kp = c->Kp;
    nav->RKp = c->P_Kp;
    nav->s_rho_q = s_rho_q;
    nav->rel_error = error_vel;
    nav->rel_error_score = error_score;
    for (int i = 0; i < 9; i++) { nav->Rot[i] = R[i]; nav->Pose[i] = c->Pose[i]; }
    so3_ln(R, nav->RotLie);
    so3_ln(c->Pose, nav->PoseLie);
    for (int i = 0; i < 3; i++) { nav->Vel[i] = -V[i] * c->K / dt_frame; nav->Pos[i] = c->Pos[i]; }
    nav->klm_num = klm_num;
    nav->kf_matchs = num_kf_back_m;
    nav->estimation_ok = EstimationOk;
    c->frame++;
    c->t_prev = t;
    return 1;
}

}  // extern "C"
