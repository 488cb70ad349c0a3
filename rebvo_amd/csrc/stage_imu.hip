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
// rebvo/linalg.h, both host+device), so the two paths cannot drift apart.  The batch dimension fills the lanes; the kernels
// are bounded to 64 threads so that a lane gets the full register file, and the file is compiled with a high unroll
// threshold (Makefile) so that the small matrices index statically and stay in registers.  Measured at 1024 sequences:
// the scale filter 20 ms with the default 128-VGPR budget (12 KB of scratch per lane), 6.9 ms bounded to 64 threads, 2.5 ms
// unrolled, 0.7-0.9 ms after round 3 (normal equations without their structural zeros, eigen-solve in registers with hardware
// reciprocals and a warm start: rebvo/imu_filters.h, rebvo/linalg.h) — and nothing in the tracker or mapper waits for it (see
// ImuSnap below).

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ctx.h"
#include "rebvo/imu_filters.h"

namespace edgehip {

namespace la = rebvo::la;
using la::Mat;
using la::Vec;

// SecondThread's IMU-branch locals that persist from frame to frame (REBVO::ImuTrack of the host library), per sequence,
// in two halves that never share a writer.  The tracker half lives on the context's main stream.  The scale filter and the
// pose (20 Gauss-Newton steps on an 11-row problem per frame: ~10 ms on one lane, whatever the batch) feed nothing back
// into the tracker or the mapper, so they live on a stream of their own (k_imu_filter, then k_imu_record) and run under the rest of the frame
// and the start of the next; what they need of a frame travels in an ImuSnap (two of them per sequence, alternating), ordered by events.
struct ImuTrackDev {            // main stream: k_imu_pre / k_imu_mid
    int32_t n_frame, init, n_giro_init, est_ok;
    double dt_frame;
    Vec<3> Vg, Bg, giro_init, g_init;
    Mat<3, 3> P_Vg, RGiro, RGBias, W_Bg, R;
    edgehip_imu_integrated imud;   // integrated IMU data of the running frame
};
struct ImuSnap {                // written on the main stream (k_imu_pre, k_imu_mid, k_imu_snap), read by k_imu_filter / k_imu_record
    int32_t have_pair, est_ok, init, x_grav_set;
    double dt, QKp;
    Vec<3> x_grav;              // gyro start-up finished in this frame: the gravity estimate X[1..3] starts from here (:196)
    Vec<3> Vg, Bg, dVv, dWv, dVgv, dWgv, Vgv, cacel;
    Mat<3, 3> R, R_pre, Rv, Qrot;   // frame rotation after / before the visual correction R0
    Vec<6> Xgv;                 // fused roto-translation and its information
    Mat<6, 6> W_Xgv;
    // the sequence state as the mapper left it
    double V[3], Kp, P_Kp, s_rho_q, t_cur, V_track[3], W_track[3], PV_track[9], PW_track[9], score, rel_error, rel_error_score;
    int32_t klm_num, klm_fwd, kf_matchs, estimation_ok, frame, minimizer_evals;
    // and of the new edge map (its slot is recycled before k_imu_record is guaranteed to have run)
    int32_t kn, pad;
    double tresh;
    float retuned, padf;
};
struct ImuFilterDev {           // side stream: k_imu_filter (k_imu_record reads it)
    int32_t n_frame, pad;
    double K, Rg;
    Vec<3> Av, As, g_est, u_est, b_est, Posgv, Posgva, dVgva, dWgva, Vgva, Pos;
    Mat<3, 3> Qg, Qbias, Rs, Rgva, Pose;
    Vec<7> X;
    Mat<7, 7> P;
    rebvo::ScaleEstimator se;
};

__device__ inline Vec<3> v3(const double *p) { Vec<3> r; r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; return r; }
__device__ inline Mat<3, 3> m3(const double *p) { Mat<3, 3> r; for (int i = 0; i < 9; i++) r.a[i] = p[i]; return r; }
__device__ inline void put(double *d, const Vec<3> &v) { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
__device__ inline void put(double *d, const Mat<3, 3> &m) { for (int i = 0; i < 9; i++) d[i] = m.a[i]; }

// rebvo_second_t.cpp:68-84 (the state a REBVO object starts its IMU branch with)
__global__ __launch_bounds__(64) void k_imu_init(ImuTrackDev *tracks, ImuFilterDev *filters, ImuSnap *snaps, edgehip_imu_params ip, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    const Vec<3> z = Vec<3>::zeros();
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    ImuTrackDev &s = tracks[seq];
    s.n_frame = 0; s.init = 0; s.n_giro_init = 0; s.est_ok = 1; s.dt_frame = 0;
    s.Vg = s.Bg = s.giro_init = s.g_init = z;
    s.P_Vg = Mat<3, 3>::identity(1e50);
    s.RGiro = I3; s.RGBias = I3; s.R = I3;
    s.W_Bg = la::inv3(s.RGBias * 100.0);
    ImuFilterDev &f = filters[seq];
    f.n_frame = 0; f.pad = 0; f.K = 1;
    f.Av = f.As = f.g_est = f.b_est = f.Posgv = f.Posgva = f.dVgva = f.dWgva = f.Vgva = f.Pos = z;
    f.u_est = z; f.u_est[0] = 1;
    f.Rgva = I3; f.Pose = I3;
    f.Qg = I3 * ip.g_uncert * ip.g_uncert;
    f.Rg = ip.g_module_uncer * ip.g_module_uncer;
    f.Rs = I3 * ip.acel_meas_std * ip.acel_meas_std;
    f.Qbias = I3 * ip.vbias_std * ip.vbias_std;
    f.X = Vec<7>::zeros();
    f.X[0] = M_PI / 4;
    f.X[2] = ip.g_module;
    f.P = Mat<7, 7>::zeros();
    f.P(0, 0) = ip.scale_std_init * ip.scale_std_init;
    f.P(1, 1) = f.P(2, 2) = f.P(3, 3) = 100;
    f.P(4, 4) = f.P(5, 5) = f.P(6, 6) = ip.vbias_std * ip.vbias_std * 1e1;
    f.se = rebvo::ScaleEstimator();
    for (int b = 0; b < 2; b++) memset(&snaps[(size_t)b * nseq + seq], 0, sizeof(ImuSnap));
}

// After the frame-begin glue and EstimateQuantile: gyro bias start-up, R = SO3(Bg)-corrected inter-frame rotation, the
// rotation k_rotate applies to the old KeyLines (R^T), the inputs of Minimizer_V.
__global__ __launch_bounds__(64) void k_imu_pre(SeqDev *seqs, ImuTrackDev *tracks, ImuSnap *snaps, const edgehip_imu_integrated *__restrict__ imu_in,
                          double *__restrict__ rot_buf, const float *__restrict__ retuned_old, edgehip_imu_params ip, int tracker_init_type,
                          int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    ImuTrackDev &s = tracks[seq];
    ImuSnap &sn = snaps[seq];
    sn.x_grav_set = 0;
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
                sn.x_grav_set = 1;                                              // istate.X.slice<1,3>() = g_init / n: the filter's state
                sn.x_grav = s.g_init / (double)s.n_giro_init;                   // is k_imu_filter's, so the value travels with the frame
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
__global__ __launch_bounds__(64) void k_imu_mid(SeqDev *seqs, ImuTrackDev *tracks, ImuSnap *snaps, const double *__restrict__ partials,
                          double *__restrict__ rot_buf, edgehip_imu_params ip, int nblk, int nblk_stride, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    edgehip_seq_state &p = sq->pub;
    ImuTrackDev &s = tracks[seq];
    ImuSnap &sn = snaps[seq];
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
    sn.dVv = la::slice<3>(Xv, 0);
    sn.dWv = la::slice<3>(Xv, 3);
    Vec<6> Xgv = Xv;
    Mat<6, 6> W_Xgv = W_Xv;
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    s.RGBias = I3 * ip.giro_bias_std * ip.giro_bias_std * dt * dt;              // :247-254
    s.RGiro = I3 * ip.giro_meas_std * ip.giro_meas_std * dt * dt;
    Vec<3> dgbias = Vec<3>::zeros();
    rebvo::imufilter::BiasCorrect(Xgv, W_Xgv, dgbias, s.W_Bg, s.RGiro, s.RGBias);
    s.Bg = s.Bg + dgbias;
    const Vec<3> dVgv = la::slice<3>(Xgv, 0), dWgv = la::slice<3>(Xgv, 3);
    Mat<3, 3> R = s.R;
    sn.R_pre = R;                                                               // Rgva = R (:257)
    const Mat<3, 3> R0 = la::so3_exp(dWgv);                                     // forward rotation
    R = la::transpose(R0 * la::transpose(R));
    s.R = R;
    const Vec<3> Vgv = R0 * s.Vg + dVgv;
    const Mat<6, 6> R_Xgv = la::Cholesky<6>(W_Xgv).inverse();
    Mat<3, 3> P_V = la::block<3, 3>(R_Xgv, 0, 0), P_W = la::block<3, 3>(R_Xgv, 3, 3);
    sn.dt = dt; sn.R = R; sn.Vg = s.Vg; sn.Bg = s.Bg; sn.dVgv = dVgv; sn.dWgv = dWgv; sn.Vgv = Vgv; sn.cacel = v3(s.imud.cacel);
    sn.Xgv = Xgv;
    sn.W_Xgv = W_Xgv;
    sn.Rv = P_V / (dt * dt * dt * dt);                                          // :284-286
    sn.Qrot = P_W;
    sn.QKp = p.P_Kp;
    sn.est_ok = ok ? 1 : 0;
    sn.init = s.init;
    s.est_ok = ok ? 1 : 0;
    s.n_frame++;
    put(rot_buf + (size_t)seq * 9, R0);                                         // :319 forward-rotate the old KeyLines
    // what the mapper kernels read; W stays zero in this branch
    Vec<3> V = Vgv;
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

// End of the frame on the main stream: what k_imu_record needs of the sequence state goes into the frame's ImuSnap (the next
// frame's kernels overwrite the state while k_imu_record may still be waiting), and the frame counter / time stamp move on (:585-606).
__global__ __launch_bounds__(64) void k_imu_snap(SeqDev *seqs, ImuTrackDev *tracks, ImuSnap *snaps, const int32_t *__restrict__ kn_new,
                           const double *__restrict__ tresh_new, const float *__restrict__ retuned_new, int have_pair, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    SeqDev *sq = seqs + seq;
    edgehip_seq_state &p = sq->pub;
    ImuSnap &sn = snaps[seq];
    sn.have_pair = have_pair;
    if (!have_pair) { sn.x_grav_set = 0; sn.init = tracks[seq].init; sn.est_ok = 0; sn.R = m3(p.R); sn.dt = p.dt; }
    sn.kn = kn_new[seq]; sn.tresh = tresh_new[seq]; sn.retuned = retuned_new[seq];
    for (int i = 0; i < 3; i++) { sn.V[i] = p.V[i]; sn.V_track[i] = sq->V_track[i]; sn.W_track[i] = sq->W_track[i]; }
    for (int i = 0; i < 9; i++) { sn.PV_track[i] = sq->PV_track[i]; sn.PW_track[i] = sq->PW_track[i]; }
    sn.Kp = p.Kp; sn.P_Kp = p.P_Kp; sn.s_rho_q = p.s_rho_q; sn.t_cur = sq->t_cur;
    sn.score = p.score; sn.rel_error = p.rel_error; sn.rel_error_score = p.rel_error_score;
    sn.klm_num = p.klm_num; sn.klm_fwd = p.klm_fwd; sn.kf_matchs = p.kf_matchs; sn.estimation_ok = p.estimation_ok;
    sn.frame = p.frame; sn.minimizer_evals = p.minimizer_evals;
    p.t_prev = sq->t_cur;
    p.frame++;
}

// On the IMU stream.  k_imu_filter — the accelerometer / scale filter (:280-312) and the gravity-aligned pose (:519-544) — reads
// what the tracker and k_imu_mid left in the snapshot and nothing of the mapper (QKp comes from P_Kp of the frame before), so it
// starts as soon as k_imu_mid is done and runs under the same frame's matching and mapping kernels (short blocks: they flow around it;
// under the next frame's one-workgroup-per-sequence stage A its sixteen waves held sixteen CUs back and cost that kernel a sixth of
// its time).  k_imu_record follows the end-of-frame snapshot and writes the frame's records (:550-606).
__global__ __launch_bounds__(64) void k_imu_filter(SeqDev *seqs, ImuFilterDev *filters, const ImuSnap *__restrict__ snaps, edgehip_imu_params ip, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    edgehip_seq_state &p = seqs[seq].pub;      // Pose, Pos, K: written here only (edgehip_get_state reads them)
    ImuFilterDev &s = filters[seq];
    const ImuSnap &sn = snaps[seq];
    const double dt = sn.dt;
    const Mat<3, 3> R = sn.R;
    if (sn.x_grav_set) la::set_slice(s.X, 1, sn.x_grav);
    s.se.EstAcelLsq4((-sn.Vgv) / dt, s.Av, R, dt);                          // :280
    s.se.MeanAcel4(sn.cacel, s.As, R);
    Vec<6> Xgva = sn.Xgv;
    s.Rgva = sn.R_pre;
    if (s.n_frame > 4 + ip.init_bias_frame_num) {                           // :291-312
        s.K = rebvo::ScaleEstimator::estKaGMEKBias(s.As, s.Av, 1, R, s.X, s.P, s.Qg, sn.Qrot, s.Qbias, sn.QKp, s.Rg, s.Rs, sn.Rv,
                                                   s.g_est, s.b_est, sn.W_Xgv, Xgva, ip.g_module);
        s.dVgva = la::slice<3>(Xgva, 0);
        s.dWgva = la::slice<3>(Xgva, 3);
        const Mat<3, 3> R0gva = la::so3_exp(s.dWgva);
        s.Rgva = la::transpose(R0gva * la::transpose(s.Rgva));
        s.Vgva = R0gva * sn.Vg + s.dVgva;
    } else {
        s.dVgva = sn.dVgv;
        s.dWgva = sn.dWgv;
        s.Rgva = R;
        s.Vgva = sn.Vgv;
    }
    if (s.n_frame > 4 + ip.init_bias_frame_num) {                           // :521-541
        s.u_est = la::transpose(s.Rgva) * s.u_est;
        s.u_est = s.u_est - s.g_est * (la::dot(s.u_est, s.g_est) / la::dot(s.g_est, s.g_est));
        s.u_est = s.u_est / sqrt(la::dot(s.u_est, s.u_est));                // TooN::normalize
        Vec<3> ey = Vec<3>::zeros(), ex = Vec<3>::zeros();
        ey[1] = 1; ex[0] = 1;
        const Mat<3, 3> PoseP1 = la::so3_from_to(s.g_est, ey);
        const Mat<3, 3> PoseP2 = la::so3_from_to(PoseP1 * s.u_est, ex);
        s.Pose = PoseP2 * PoseP1;
        s.Pos = s.Pos + (-s.Pose) * s.Vgva * s.K;
        s.Posgva = s.Pos;
        s.Posgv = s.Posgv + (-s.Pose) * sn.Vgv * s.K;
    }
    put(p.Pose, s.Pose);
    put(p.Pos, s.Pos);
    p.K = s.K;
    s.n_frame++;
}

__global__ __launch_bounds__(64) void k_imu_record(const ImuFilterDev *__restrict__ filters, const ImuSnap *__restrict__ snaps, edgehip_nav *__restrict__ nav,
                           edgehip_nav_imu *__restrict__ nav_imu, edgehip_nav *__restrict__ nav_log, edgehip_nav_imu *__restrict__ nav_imu_log, int nav_log_len, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq) return;
    const ImuFilterDev &s = filters[seq];
    const ImuSnap &sn = snaps[seq];
    edgehip_nav &o = nav[seq];
    edgehip_nav_imu &oi = nav_imu[seq];
    memset(&oi, 0, sizeof oi);
    oi.kn = sn.kn;
    const int have_pair = sn.have_pair;
    if (have_pair) {
        const double dt = sn.dt;
        const Mat<3, 3> R = sn.R;
        const Vec<3> V = v3(sn.V);
        // the record (:550-606)
        oi.dt = dt; oi.K = s.K; oi.Kp = sn.Kp; oi.RKp = sn.P_Kp; oi.s_rho_q = sn.s_rho_q; oi.scale = s.K;
        put(oi.Rot, R);
        put(oi.RotLie, la::so3_ln(R));
        put(oi.RotGiro, la::so3_ln(s.Rgva) / dt);
        put(oi.Vel, ((-V) * s.K) / dt);
        put(oi.Pose, s.Pose);
        put(oi.PoseLie, la::so3_ln(s.Pose));
        put(oi.Pos, s.Pos);
        put(oi.g, s.g_est);
        put(oi.Vg, sn.Vg); put(oi.Bg, sn.Bg); put(oi.dVv, sn.dVv); put(oi.dWv, sn.dWv); put(oi.Vgv, sn.Vgv); put(oi.Vgva, s.Vgva);
        put(oi.Av, s.Av); put(oi.As, s.As);
        for (int i = 0; i < 7; i++) oi.X[i] = s.X[i];
        put(oi.b_est, s.b_est); put(oi.u_est, s.u_est);
        oi.klm_num = sn.klm_num;
        oi.estimation_ok = sn.estimation_ok && sn.est_ok;
        oi.init = sn.init;
    }
    // the common record, as k_frame_glue (mode 3) fills it
    o.t = sn.t_cur; o.dt = sn.dt;
    for (int i = 0; i < 3; i++) { o.V[i] = sn.V_track[i]; o.W[i] = sn.W_track[i]; }
    for (int i = 0; i < 9; i++) { o.P_V[i] = sn.PV_track[i]; o.P_W[i] = sn.PW_track[i]; o.Rot[i] = sn.R.a[i]; o.Pose[i] = s.Pose.a[i]; }
    for (int i = 0; i < 3; i++) { o.RotLie[i] = oi.RotLie[i]; o.PoseLie[i] = oi.PoseLie[i]; o.Vel[i] = oi.Vel[i]; o.Pos[i] = s.Pos[i]; }
    o.Kp = sn.Kp; o.RKp = sn.P_Kp; o.s_rho_q = sn.s_rho_q; o.tresh = sn.tresh;
    o.score = sn.score; o.rel_error = sn.rel_error; o.rel_error_score = sn.rel_error_score;
    o.retuned_thresh = sn.retuned;
    o.kn = sn.kn; o.klm_fwd = sn.klm_fwd; o.klm_num = sn.klm_num; o.kf_matchs = sn.kf_matchs;
    o.estimation_ok = have_pair ? oi.estimation_ok : 0;
    o.frame = sn.frame; o.minimizer_evals = sn.minimizer_evals;
    if (nav_log_len > 0) nav_log[(size_t)(sn.frame % nav_log_len) * nseq + seq] = o;
    if (nav_log_len > 0 && nav_imu_log) nav_imu_log[(size_t)(sn.frame % nav_log_len) * nseq + seq] = oi;
}

// REBVO::Reset() (rebvo_second_t.cpp:609-620): pose and position start over; they are this stream's
__global__ void k_imu_pose_reset(SeqDev *seqs, ImuFilterDev *filters, int only_seq, int nseq) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= nseq || (only_seq >= 0 && seq != only_seq)) return;
    ImuFilterDev &s = filters[seq];
    s.Pose = Mat<3, 3>::identity();
    s.Pos = Vec<3>::zeros();
    put(seqs[seq].pub.Pose, s.Pose);
    put(seqs[seq].pub.Pos, s.Pos);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// edgehip_reset: the IMU branch starts over with the sequences (both streams are idle when this is called)
int imu_reset_enqueue(edgehip_ctx *c) {
    const int B = c->plan.nseq;
    hipLaunchKernelGGL(k_imu_init, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream, (ImuTrackDev *)c->imu_track,
                       (ImuFilterDev *)c->imu_filter, (ImuSnap *)c->imu_snap, c->imu_params, B);
    EH_LAUNCH_CHECK();
    c->imu_post_valid[0] = c->imu_post_valid[1] = false;
    c->imu_pending = false;
    return 0;
}
int imu_pose_reset_enqueue(edgehip_ctx *c, int seq) {
    const int B = c->plan.nseq;
    hipLaunchKernelGGL(k_imu_pose_reset, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream_imu, c->seq, (ImuFilterDev *)c->imu_filter, seq, B);
    EH_LAUNCH_CHECK();
    return 0;
}
static ImuSnap *snap_of(edgehip_ctx *c) { return (ImuSnap *)c->imu_snap + (size_t)(c->frames_seen & 1) * c->plan.nseq; }

// start of a frame's IMU work on the main stream: its snapshot slot was last read by k_imu_filter / k_imu_record two frames ago
int imu_begin_enqueue(edgehip_ctx *c) {
    const int b = c->frames_seen & 1;
    if (c->imu_post_valid[b]) EH_CHECK(hipStreamWaitEvent(c->stream, c->ev_imu_post[b], 0));
    return 0;
}
int imu_pre_enqueue(edgehip_ctx *c, int slot_old) {
    const int B = c->plan.nseq;
    hipLaunchKernelGGL(k_imu_pre, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, snap_of(c),
                       c->imu_in_dev, c->rot_buf, c->retuned_slot + (size_t)slot_old * B, c->imu_params, c->p.tracker_init_type, B);
    EH_LAUNCH_CHECK();
    return 0;
}
int imu_mid_enqueue(edgehip_ctx *c) {
    const int B = c->plan.nseq, nblk = (c->plan.cap + 255) / 256;
    hipLaunchKernelGGL(k_imu_mid, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, snap_of(c),
                       c->partials, c->rot_buf, c->imu_params, nblk, c->nblk_tvr, B);
    EH_LAUNCH_CHECK();
    // the scale filter + pose of this frame: on the IMU stream from here on, under the frame's matching and mapping
    const int b = c->frames_seen & 1;
    EH_CHECK(hipEventRecord(c->ev_imu_mid[b], c->stream));
    EH_CHECK(hipStreamWaitEvent(c->stream_imu, c->ev_imu_mid[b], 0));
    {
        ProfScope ps(c, PROF_IMU_SCALE_POSE, c->stream_imu);
        static const int bs = []() { const char *e = getenv("EDGEHIP_IMU_BLOCK"); const int v = e ? atoi(e) : 64; return v > 0 && v <= 64 ? v : 64; }();
        hipLaunchKernelGGL(k_imu_filter, dim3((B + bs - 1) / bs), dim3(bs), 0, c->stream_imu, c->seq, (ImuFilterDev *)c->imu_filter, snap_of(c),
                           c->imu_params, B);
        EH_LAUNCH_CHECK();
    }
    return 0;
}
// end of the frame: snapshot on the main stream, then the frame's records on the IMU stream, behind the scale filter + pose that
// imu_mid_enqueue put there (the nav records are complete once that stream is — every reader synchronises both)
int imu_post_enqueue(edgehip_ctx *c, int slot_new, int have_pair) {
    const int B = c->plan.nseq, b = c->frames_seen & 1;
    hipLaunchKernelGGL(k_imu_snap, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->seq, (ImuTrackDev *)c->imu_track, snap_of(c),
                       c->kn_slot + (size_t)slot_new * B, c->tresh_slot + (size_t)slot_new * B, c->retuned_slot + (size_t)slot_new * B, have_pair, B);
    EH_LAUNCH_CHECK();
    EH_CHECK(hipEventRecord(c->ev_imu_snap[b], c->stream));
    EH_CHECK(hipStreamWaitEvent(c->stream_imu, c->ev_imu_snap[b], 0));
    hipLaunchKernelGGL(k_imu_record, dim3((B + 63) / 64), dim3(64), 0, c->stream_imu, (const ImuFilterDev *)c->imu_filter, snap_of(c),
                       c->nav_dev, c->nav_imu_dev, c->nav_log, c->nav_imu_log, c->nav_log_len, B);
    EH_LAUNCH_CHECK();
    EH_CHECK(hipEventRecord(c->ev_imu_post[b], c->stream_imu));
    c->imu_post_valid[b] = true;
    return 0;
}

}  // namespace edgehip

using namespace edgehip;

extern "C" {

// The scale filter is ONE long kernel per frame (~0.5 ms, a thread per sequence) that is meant to run UNDER the next frames' kernels.  HIP
// maps its streams onto a handful of hardware queues (four by default); a context has up to six streams (frames, stage A, uploads, the
// nav log, the KeyLine export, this one), and whichever stream lands on the filter's queue waits behind it — behind a batch group
// (upload stream + log stream in use) the next frame's copy / stage A did: a step of eight sequences took frame + filter = 1.1 ms instead
// of max(frame, filter) = 0.58.  A stream of another PRIORITY gets a hardware queue of its own kind: the filter runs at the lowest
// priority (it has a whole frame of slack), everything else stays where it was.
static bool imu_stream_create(hipStream_t *st) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
        hipStreamCreateWithPriority(st, hipStreamNonBlocking, least) == hipSuccess)
        return true;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
}

int edgehip_imu_enable(edgehip_ctx *c, const edgehip_imu_params *imu) {
    EH_ENTER(c);
    if (!imu) return EDGEHIP_ERR_ARG;
    if (c->frames_seen != 0) { set_error("edgehip_imu_enable: call before the first frame"); return EDGEHIP_ERR_STATE; }
    if (c->rig.enabled) { set_error("edgehip_imu_enable: the device IMU branch does not run the stereo rig (use the stage entry points)"); return EDGEHIP_ERR_STATE; }
    const size_t B = c->plan.nseq;
    if (!c->imu_pinned_ok) {
        // All or nothing: a call that fails half-way leaves the context without IMU buffers, so that a second call starts over
        // instead of enabling the branch with null pointers.
        void *track = nullptr, *filter = nullptr, *snap = nullptr, *in_dev = nullptr, *nav_dev = nullptr, *pin_in = nullptr, *pin_nav = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool ok = hipMalloc(&track, sizeof(ImuTrackDev) * B) == hipSuccess && hipMalloc(&filter, sizeof(ImuFilterDev) * B) == hipSuccess &&
                  hipMalloc(&snap, sizeof(ImuSnap) * B * 2) == hipSuccess && hipMalloc(&in_dev, sizeof(edgehip_imu_integrated) * B) == hipSuccess &&
                  hipMalloc(&nav_dev, sizeof(edgehip_nav_imu) * B) == hipSuccess &&
                  hipHostMalloc(&pin_in, sizeof(edgehip_imu_integrated) * B * 8, hipHostMallocDefault) == hipSuccess &&
                  hipHostMalloc(&pin_nav, sizeof(edgehip_nav_imu) * B, hipHostMallocDefault) == hipSuccess &&
                  imu_stream_create(&st);
        for (int i = 0; ok && i < 6; i++) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (ok) ok = hipMemsetAsync(nav_dev, 0, sizeof(edgehip_nav_imu) * B, c->stream) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (int i = 0; i < 6; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (st) (void)hipStreamDestroy(st);
            if (pin_nav) (void)hipHostFree(pin_nav);
            if (pin_in) (void)hipHostFree(pin_in);
            for (void *q : {nav_dev, in_dev, snap, filter, track}) if (q) (void)hipFree(q);
            set_error("edgehip_imu_enable: allocating the IMU branch's state failed");
            return EDGEHIP_ERR_MEMORY;
        }
        c->imu_track = track; c->imu_filter = filter; c->imu_snap = snap;
        c->imu_in_dev = (edgehip_imu_integrated *)in_dev; c->nav_imu_dev = (edgehip_nav_imu *)nav_dev;
        c->pinned_imu = (edgehip_imu_integrated *)pin_in; c->pinned_nav_imu = (edgehip_nav_imu *)pin_nav;
        c->stream_imu = st;
        for (int i = 0; i < 2; i++) { c->ev_imu_snap[i] = ev[i]; c->ev_imu_post[i] = ev[2 + i]; c->ev_imu_mid[i] = ev[4 + i]; c->imu_post_valid[i] = false; }
        c->imu_pinned_ok = true;
    }
    if (c->nav_log_len > 0 && !c->nav_imu_log) {   // a log set before the branch was switched on: its IMU half (edgehip_read_nav_imu_log)
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(edgehip_nav_imu) * (size_t)c->nav_log_len * B) != hipSuccess) { (void)hipGetLastError(); set_error("edgehip_imu_enable: nav log alloc failed"); return EDGEHIP_ERR_MEMORY; }
        EH_CHECK(hipMemsetAsync(q, 0, sizeof(edgehip_nav_imu) * (size_t)c->nav_log_len * B, c->stream));
        c->nav_imu_log = (edgehip_nav_imu *)q;
    }
    c->imu_params = *imu;
    c->imu_enabled = true;
    c->imu_pending = false;
    c->imu_post_valid[0] = c->imu_post_valid[1] = false;
    drop_frame_graphs(c);
    c->use_graph = false;   // the IMU stream runs past the end of a frame: that is not a capturable fork/join
    hipLaunchKernelGGL(k_imu_init, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream, (ImuTrackDev *)c->imu_track,
                       (ImuFilterDev *)c->imu_filter, (ImuSnap *)c->imu_snap, c->imu_params, (int)B);
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
