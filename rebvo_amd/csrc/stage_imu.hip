// stage_imu.hip — the IMU branch of SecondThread (ImuMode > 0) for whole batches, on the device.
//
// Replaces, per frame pair and sequence (reference file:line):
//   gyro bias start-up                                   src/rebvo/rebvo_second_t.cpp:183-203
//   gyro pre-rotation of the old KeyLines                :208-215   (k_imu_pre writes the rotation, k_rotate applies it)
//   edge_tracker::ExtRotVel's 6x6 solve                  src/mtracklib/edge_tracker.cpp:1283-1296 (sums: k_ext_rotvel)
//   edge_tracker::BiasCorrect                            src/mtracklib/edge_tracker.cpp:1308-1343
//   roto-translation / covariance bookkeeping            rebvo_second_t.cpp:240-272
//   ScaleEstimator::EstAcelLsq4 / MeanAcel4 / estKaGMEKBias   src/mtracklib/scaleestimator.cpp:38-318   (:280-312)
//   gravity-aligned pose and the NavData record          rebvo_second_t.cpp:519-606
// The per-KeyLine work in between (Minimizer_V, FordwardMatch, ExtRotVel's rows, rotate, matching, EKF, rescaling) is the
// kernels of stage_b.hip / stage_c.hip; edgehip_process_frame strings everything together without a host synchronisation
// (stage_c.hip::frame_enqueue).
//
// The filters are 3..11-dimensional dense algebra with data-dependent control flow and no parallelism worth a wave: one
// THREAD per sequence runs the very code the host library runs for a single live camera (rebvo/imu_filters.h over
// rebvo/linalg.h, both host+device), so the two paths cannot drift apart.  Slow per thread (the scale filter is 20
// Gauss-Newton steps on an 11-row problem, ~1 ms on one lane) but the batch dimension fills the lanes, and nothing in
// the tracker or mapper waits for it: it only feeds the pose and the record of the frame.

#include <math.h>
#include <string.h>

#include "ctx.h"
#include "rebvo/imu_filters.h"

namespace edgehip {

namespace la = rebvo::la;
using la::Mat;
using la::Vec;

// SecondThread's IMU-branch locals that persist from frame to frame (REBVO::ImuTrack of the host library), per sequence.
struct ImuTrackDev {
    int32_t n_frame, init, n_giro_init, est_ok;
    double K, QKp, Rg, dt_frame;
    Vec<3> Vg, Bg, Av, As, dVv, dWv, dVgv, dWgv, Vgv, Wgv, dVgva, dWgva, Vgva;
    Vec<3> g_est, u_est, b_est, Posgv, Posgva, giro_init, g_init;
    Mat<3, 3> P_Vg, RGiro, RGBias, W_Bg, Qrot, Qg, Qbias, Rs, Rv, Rgva, R, R0;
    Vec<7> X;
    Mat<7, 7> P;
    Vec<6> Xgv;        // fused roto-translation and its information: handed from k_imu_mid to k_imu_post
    Mat<6, 6> W_Xgv;
    rebvo::ScaleEstimator se;
    edgehip_imu_integrated imud;   // integrated IMU data of the running frame
};

__device__ inline Vec<3> v3(const double *p) { Vec<3> r; r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; return r; }
__device__ inline Mat<3, 3> m3(const double *p) { Mat<3, 3> r; for (int i = 0; i < 9; i++) r.a[i] = p[i]; return r; }
__device__ inline void put(double *d, const Vec<3> &v) { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
__device__ inline void put(double *d, const Mat<3, 3> &m) { for (int i = 0; i < 9; i++) d[i] = m.a[i]; }

// rebvo_second_t.cpp:68-84 (the state a REBVO object starts its IMU branch with)
__global__ void k_imu_init(ImuTrackDev *tracks, edgehip_imu_params ip, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    ImuTrackDev &s = tracks[seq];
    s.n_frame = 0; s.init = 0; s.n_giro_init = 0; s.est_ok = 1;
    s.K = 1; s.QKp = 0; s.dt_frame = 0;
    const Vec<3> z = Vec<3>::zeros();
    s.Vg = s.Bg = s.Av = s.As = s.dVv = s.dWv = s.dVgv = s.dWgv = s.Vgv = s.Wgv = s.dVgva = s.dWgva = s.Vgva = z;
    s.g_est = s.b_est = s.Posgv = s.Posgva = s.giro_init = s.g_init = z;
    s.u_est = z; s.u_est[0] = 1;
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    s.P_Vg = Mat<3, 3>::identity(1e50);
    s.RGiro = I3; s.RGBias = I3; s.Qrot = I3; s.Rv = I3; s.Rgva = I3; s.R = I3; s.R0 = I3;
    s.W_Bg = la::inv3(s.RGBias * 100.0);
    s.Qg = I3 * ip.g_uncert * ip.g_uncert;
    s.Rg = ip.g_module_uncer * ip.g_module_uncer;
    s.Rs = I3 * ip.acel_meas_std * ip.acel_meas_std;
    s.Qbias = I3 * ip.vbias_std * ip.vbias_std;
    s.X = Vec<7>::zeros();
    s.X[0] = M_PI / 4;
    s.X[2] = ip.g_module;
    s.P = Mat<7, 7>::zeros();
    s.P(0, 0) = ip.scale_std_init * ip.scale_std_init;
    s.P(1, 1) = s.P(2, 2) = s.P(3, 3) = 100;
    s.P(4, 4) = s.P(5, 5) = s.P(6, 6) = ip.vbias_std * ip.vbias_std * 1e1;
    s.Xgv = Vec<6>::zeros();
    s.W_Xgv = Mat<6, 6>::zeros();
    s.se = rebvo::ScaleEstimator();
}

// After the frame-begin glue and EstimateQuantile: gyro bias start-up, R = SO3(Bg)-corrected inter-frame rotation, the
// rotation k_rotate applies to the old KeyLines (R^T), the inputs of Minimizer_V.
__global__ void k_imu_pre(SeqDev *seqs, ImuTrackDev *tracks, const edgehip_imu_integrated *__restrict__ imu_in, double *__restrict__ rot_buf,
                          const float *__restrict__ retuned_old, edgehip_imu_params ip, int tracker_init_type, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    ImuTrackDev &s = tracks[seq];
    s.imud = imu_in[seq];
    const Vec<3> giro = v3(s.imud.giro), cacel = v3(s.imud.cacel);
    s.dt_frame = sq->pub.dt;
    s.est_ok = 1;
    if (!s.init && s.n_frame > 0) {                                            // :183-203
        if (ip.init_bias > 0) {
            s.giro_init = s.giro_init + giro * s.imud.dt;
            s.g_init = s.g_init - cacel;
            if (++s.n_giro_init > ip.init_bias_frame_num) {
                s.Bg = s.giro_init / (double)s.n_giro_init;
                s.init = 1;
                s.W_Bg = la::inv3(s.RGBias * 1e2);
                la::set_slice(s.X, 1, s.g_init / (double)s.n_giro_init);
            }
        } else {
            s.init = 1;
            s.Bg = v3(ip.bias_init_guess) * s.imud.dt;
        }
    }
    Mat<3, 3> R = m3(s.imud.Rot);                                               // :208
    R = la::transpose(la::so3_exp(s.Bg) * la::transpose(R));                    // R.T() = SO3(Bg) * R.T()
    s.R = R;
    put(rot_buf + (size_t)seq * 9, la::transpose(R));                           // forward pre-rotation of the old KeyLines
    if (tracker_init_type == 0) s.Vg = Vec<3>::zeros();
    put(sq->mv_V, s.Vg);
    sq->mv_s_rho_min = sq->pub.s_rho_q;
    sq->mv_min_mod = retuned_old[seq];                                          // old_buf.ef->getThresh()
}

// After Minimizer_V, FordwardMatch and the ExtRotVel sums: the 6x6 solve, BiasCorrect, the fused roto-translation and its
// covariances (rebvo_second_t.cpp:237-272), the second rotation of the old KeyLines, and what the mapper reads from the
// sequence state (V, P_V, P_W, R); the NaN restart of :387-397.
__global__ void k_imu_mid(SeqDev *seqs, ImuTrackDev *tracks, const double *__restrict__ partials, double *__restrict__ rot_buf,
                          edgehip_imu_params ip, int nblk, int nblk_stride, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    edgehip_seq_state &p = sq->pub;
    ImuTrackDev &s = tracks[seq];
    const double dt = s.dt_frame;
    s.Vg = v3(sq->mv_V);                                                        // Minimizer_V's result
    s.P_Vg = m3(sq->mv_RVel);
    // ExtRotVel: Phi^T Phi, Phi^T Y (blocks summed in order), X = SVD(JtJ).backsub(JtF), Rx = get_pinv (edge_tracker.cpp:1283-1296)
    Mat<6, 6> W_Xv;
    Vec<6> JtF;
    {
        double sum[kNumSums];
        for (int k = 0; k < kNumSums; k++) sum[k] = 0;
        const double *pp = partials + (size_t)seq * nblk_stride * kNumSums;
        for (int b = 0; b < nblk; b++)
            for (int k = 0; k < kNumSums; k++) sum[k] += pp[(size_t)b * kNumSums + k];
        int ns = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { W_Xv(a, b) = sum[ns]; W_Xv(b, a) = sum[ns]; ns++; }
        for (int a = 0; a < 6; a++) JtF[a] = sum[ns++];
    }
    const la::SymSVD<6> svd(W_Xv);
    const Mat<6, 6> R_Xv = svd.pinv();
    Vec<6> Xv = R_Xv * JtF;
    bool ok = !(la::has_nan(Xv) || la::has_nan(R_Xv));
    s.dVv = la::slice<3>(Xv, 0);
    s.dWv = la::slice<3>(Xv, 3);
    Vec<6> Xgv = Xv;
    Mat<6, 6> W_Xgv = W_Xv;
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    s.RGBias = I3 * ip.giro_bias_std * ip.giro_bias_std * dt * dt;              // :247-254
    s.RGiro = I3 * ip.giro_meas_std * ip.giro_meas_std * dt * dt;
    Vec<3> dgbias = Vec<3>::zeros();
    rebvo::imufilter::BiasCorrect(Xgv, W_Xgv, dgbias, s.W_Bg, s.RGiro, s.RGBias);
    s.Bg = s.Bg + dgbias;
    s.dVgv = la::slice<3>(Xgv, 0);
    s.dWgv = la::slice<3>(Xgv, 3);
    Mat<3, 3> R = s.R;
    s.Rgva = R;
    const Mat<3, 3> R0 = la::so3_exp(s.dWgv);                                   // forward rotation
    R = la::transpose(R0 * la::transpose(R));
    s.R = R;
    s.R0 = R0;
    s.Vgv = R0 * s.Vg + s.dVgv;
    s.Wgv = la::so3_ln(R);
    const Mat<6, 6> R_Xgv = la::Cholesky<6>(W_Xgv).inverse();
    Mat<3, 3> P_V = la::block<3, 3>(R_Xgv, 0, 0), P_W = la::block<3, 3>(R_Xgv, 3, 3);
    s.Xgv = Xgv;
    s.W_Xgv = W_Xgv;
    s.Rv = P_V / (dt * dt * dt * dt);                                           // :284-286
    s.Qrot = P_W;
    s.QKp = p.P_Kp;
    s.est_ok = ok ? 1 : 0;
    put(rot_buf + (size_t)seq * 9, R0);                                         // :319 forward-rotate the old KeyLines
    // what the mapper kernels read; W stays zero in this branch
    Vec<3> V = s.Vgv;
    put(sq->V_track, V);
    sq->W_track[0] = sq->W_track[1] = sq->W_track[2] = 0;
    for (int i = 0; i < 9; i++) { sq->PV_track[i] = P_V.a[i]; sq->PW_track[i] = P_W.a[i]; }
    p.estimation_ok = ok ? 1 : 0;
    if (la::has_nan(V)) {                                                       // :387-397
        P_V = Mat<3, 3>::identity(1e50);
        V = Vec<3>::zeros();
        p.Kp = 1;
        p.P_Kp = 1e50;
        p.estimation_ok = 0;
        sq->skip_match = 1;
        sq->skip_map = 1;
    }
    put(p.V, V);
    p.W[0] = p.W[1] = p.W[2] = 0;
    put(p.P_V, P_V);
    put(p.P_W, P_W);
    put(p.R, R);
}

// After the mapper: the accelerometer / scale filter (:280-312 — it reads what the tracker left, nothing of the mapper
// except P_Kp of the frame before), the gravity-aligned pose (:519-544) and the record of the frame (:550-606).
__global__ void k_imu_post(SeqDev *seqs, ImuTrackDev *tracks, edgehip_nav *__restrict__ nav, edgehip_nav_imu *__restrict__ nav_imu,
                           const int32_t *__restrict__ kn_new, const double *__restrict__ tresh_new, const float *__restrict__ retuned_new,
                           edgehip_imu_params ip, int have_pair, edgehip_nav *__restrict__ nav_log, int nav_log_len, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    edgehip_seq_state &p = sq->pub;
    ImuTrackDev &s = tracks[seq];
    edgehip_nav &o = nav[seq];
    edgehip_nav_imu &oi = nav_imu[seq];
    memset(&oi, 0, sizeof oi);
    oi.kn = kn_new[seq];
    if (have_pair) {
        const double dt = s.dt_frame;
        const Mat<3, 3> R = s.R;
        s.se.EstAcelLsq4((-s.Vgv) / dt, s.Av, R, dt);                           // :280
        s.se.MeanAcel4(v3(s.imud.cacel), s.As, R);
        Vec<6> Xgva = s.Xgv;
        if (s.n_frame > 4 + ip.init_bias_frame_num) {                           // :291-312
            s.K = rebvo::ScaleEstimator::estKaGMEKBias(s.As, s.Av, 1, R, s.X, s.P, s.Qg, s.Qrot, s.Qbias, s.QKp, s.Rg, s.Rs, s.Rv,
                                                       s.g_est, s.b_est, s.W_Xgv, Xgva, ip.g_module);
            s.dVgva = la::slice<3>(Xgva, 0);
            s.dWgva = la::slice<3>(Xgva, 3);
            const Mat<3, 3> R0gva = la::so3_exp(s.dWgva);
            s.Rgva = la::transpose(R0gva * la::transpose(s.Rgva));
            s.Vgva = R0gva * s.Vg + s.dVgva;
        } else {
            s.dVgva = s.dVgv;
            s.dWgva = s.dWgv;
            s.Rgva = R;
            s.Vgva = s.Vgv;
        }
        Mat<3, 3> Pose = m3(p.Pose);
        Vec<3> Pos = v3(p.Pos);
        if (s.n_frame > 4 + ip.init_bias_frame_num) {                           // :521-541
            s.u_est = la::transpose(s.Rgva) * s.u_est;
            s.u_est = s.u_est - s.g_est * (la::dot(s.u_est, s.g_est) / la::dot(s.g_est, s.g_est));
            s.u_est = s.u_est / sqrt(la::dot(s.u_est, s.u_est));                // TooN::normalize
            Vec<3> ey = Vec<3>::zeros(), ex = Vec<3>::zeros();
            ey[1] = 1; ex[0] = 1;
            const Mat<3, 3> PoseP1 = la::so3_from_to(s.g_est, ey);
            const Mat<3, 3> PoseP2 = la::so3_from_to(PoseP1 * s.u_est, ex);
            Pose = PoseP2 * PoseP1;
            Pos = Pos + (-Pose) * s.Vgva * s.K;
            s.Posgva = Pos;
            s.Posgv = s.Posgv + (-Pose) * s.Vgv * s.K;
        }
        put(p.Pose, Pose);
        put(p.Pos, Pos);
        p.K = s.K;
        const Vec<3> V = v3(p.V);
        // the record (:550-606)
        oi.dt = dt; oi.K = s.K; oi.Kp = p.Kp; oi.RKp = p.P_Kp; oi.s_rho_q = p.s_rho_q; oi.scale = s.K;
        put(oi.Rot, R);
        put(oi.RotLie, la::so3_ln(R));
        put(oi.RotGiro, la::so3_ln(s.Rgva) / dt);
        put(oi.Vel, ((-V) * s.K) / dt);
        put(oi.Pose, Pose);
        put(oi.PoseLie, la::so3_ln(Pose));
        put(oi.Pos, Pos);
        put(oi.g, s.g_est);
        put(oi.Vg, s.Vg); put(oi.Bg, s.Bg); put(oi.dVv, s.dVv); put(oi.dWv, s.dWv); put(oi.Vgv, s.Vgv); put(oi.Vgva, s.Vgva);
        put(oi.Av, s.Av); put(oi.As, s.As);
        for (int i = 0; i < 7; i++) oi.X[i] = s.X[i];
        put(oi.b_est, s.b_est); put(oi.u_est, s.u_est);
        oi.klm_num = p.klm_num;
        oi.estimation_ok = p.estimation_ok && s.est_ok;
        oi.init = s.init;
        s.n_frame++;
    }
    // the common record, as k_frame_glue (mode 3) fills it
    o.t = sq->t_cur; o.dt = p.dt;
    for (int i = 0; i < 3; i++) { o.V[i] = sq->V_track[i]; o.W[i] = sq->W_track[i]; }
    for (int i = 0; i < 9; i++) { o.P_V[i] = sq->PV_track[i]; o.P_W[i] = sq->PW_track[i]; o.Rot[i] = p.R[i]; o.Pose[i] = p.Pose[i]; }
    for (int i = 0; i < 3; i++) { o.RotLie[i] = oi.RotLie[i]; o.PoseLie[i] = oi.PoseLie[i]; o.Vel[i] = oi.Vel[i]; o.Pos[i] = p.Pos[i]; }
    o.Kp = p.Kp; o.RKp = p.P_Kp; o.s_rho_q = p.s_rho_q; o.tresh = tresh_new[seq];
    o.score = p.score; o.rel_error = p.rel_error; o.rel_error_score = p.rel_error_score;
    o.retuned_thresh = retuned_new[seq];
    o.kn = kn_new[seq]; o.klm_fwd = p.klm_fwd; o.klm_num = p.klm_num; o.kf_matchs = p.kf_matchs;
    o.estimation_ok = have_pair ? oi.estimation_ok : 0;
    o.frame = p.frame; o.minimizer_evals = p.minimizer_evals;
    if (nav_log_len > 0) nav_log[(size_t)(p.frame % nav_log_len) * nseq + seq] = o;
    p.t_prev = sq->t_cur;
    p.frame++;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int imu_pre_enqueue(edgehip_ctx *c, int slot_old) {
    const int B = c->plan.nseq;
    hipLaunchKernelGGL(k_imu_pre, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, c->imu_in_dev,
                       c->rot_buf, c->retuned_slot + (size_t)slot_old * B, c->imu_params, c->p.tracker_init_type, B);
    EH_LAUNCH_CHECK();
    return 0;
}
int imu_mid_enqueue(edgehip_ctx *c) {
    const int B = c->plan.nseq, nblk = (c->plan.cap + 255) / 256;
    hipLaunchKernelGGL(k_imu_mid, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, c->partials,
                       c->rot_buf, c->imu_params, nblk, c->nblk_tvr, B);
    EH_LAUNCH_CHECK();
    return 0;
}
int imu_post_enqueue(edgehip_ctx *c, int slot_new, int have_pair) {
    const int B = c->plan.nseq;
    hipLaunchKernelGGL(k_imu_post, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, c->nav_dev,
                       c->nav_imu_dev, c->kn_slot + (size_t)slot_new * B, c->tresh_slot + (size_t)slot_new * B,
                       c->retuned_slot + (size_t)slot_new * B, c->imu_params, have_pair, c->nav_log, c->nav_log_len, B);
    EH_LAUNCH_CHECK();
    return 0;
}

}  // namespace edgehip

using namespace edgehip;

extern "C" {

int edgehip_imu_enable(edgehip_ctx *c, const edgehip_imu_params *imu) {
    EH_ENTER(c);
    if (!imu) return EDGEHIP_ERR_ARG;
    if (c->frames_seen != 0) { set_error("edgehip_imu_enable: call before the first frame"); return EDGEHIP_ERR_STATE; }
    if (c->rig.enabled) { set_error("edgehip_imu_enable: the device IMU branch does not run the stereo rig (use the stage entry points)"); return EDGEHIP_ERR_STATE; }
    const size_t B = c->plan.nseq;
    if (!c->imu_track) {
        void *q;
        if (hipMalloc(&q, sizeof(ImuTrackDev) * B) != hipSuccess) { (void)hipGetLastError(); set_error("imu state alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->imu_track = q;
        if (hipMalloc(&q, sizeof(edgehip_imu_integrated) * B) != hipSuccess) { (void)hipGetLastError(); set_error("imu input alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->imu_in_dev = (edgehip_imu_integrated *)q;
        if (hipMalloc(&q, sizeof(edgehip_nav_imu) * B) != hipSuccess) { (void)hipGetLastError(); set_error("imu nav alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->nav_imu_dev = (edgehip_nav_imu *)q;
        EH_CHECK(hipMemsetAsync(c->nav_imu_dev, 0, sizeof(edgehip_nav_imu) * B, c->stream));
        EH_CHECK(hipHostMalloc(&q, sizeof(edgehip_imu_integrated) * B * 8, hipHostMallocDefault));
        c->pinned_imu = (edgehip_imu_integrated *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(edgehip_nav_imu) * B, hipHostMallocDefault));
        c->pinned_nav_imu = (edgehip_nav_imu *)q;
    }
    c->imu_params = *imu;
    c->imu_enabled = true;
    c->imu_pending = false;
    drop_frame_graphs(c);
    hipLaunchKernelGGL(k_imu_init, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream, (ImuTrackDev *)c->imu_track, c->imu_params, (int)B);
    EH_LAUNCH_CHECK();
    return 0;
}

int edgehip_set_imu(edgehip_ctx *c, const edgehip_imu_integrated *per_seq) {
    EH_ENTER(c);
    if (!per_seq) return EDGEHIP_ERR_ARG;
    if (!c->imu_enabled) { set_error("edgehip_set_imu: edgehip_imu_enable was not called"); return EDGEHIP_ERR_STATE; }
    const size_t B = c->plan.nseq;
    if (int e = wait_pinned_ring(c)) return e;          // the entry of the frame eight back is free again
    edgehip_imu_integrated *pi = c->pinned_imu + (size_t)(c->frames_seen % 8) * B;
    memcpy(pi, per_seq, sizeof(edgehip_imu_integrated) * B);
    EH_CHECK(hipMemcpyAsync(c->imu_in_dev, pi, sizeof(edgehip_imu_integrated) * B, hipMemcpyHostToDevice, c->stream));
    c->imu_pending = true;
    return 0;
}

int edgehip_read_nav_imu(edgehip_ctx *c, edgehip_nav_imu *out) {
    EH_ENTER(c);
    if (!out) return EDGEHIP_ERR_ARG;
    if (!c->imu_enabled) { set_error("edgehip_read_nav_imu: edgehip_imu_enable was not called"); return EDGEHIP_ERR_STATE; }
    if (int e = sync_all(c)) return e;
    const size_t B = c->plan.nseq;
    EH_CHECK(hipMemcpyAsync(c->pinned_nav_imu, c->nav_imu_dev, sizeof(edgehip_nav_imu) * B, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pinned_nav_imu, sizeof(edgehip_nav_imu) * B);
    return 0;
}

}  // extern "C"
