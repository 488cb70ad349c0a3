// rebvo_imu.cpp — the IMU branch of the tracking thread: what REBVO::SecondThread does per frame pair when
// ImuMode > 0 (src/rebvo/rebvo_second_t.cpp:54-94 set-up, :128-336 tracker + filters, :387-493 mapper, :519-606 pose and
// NavData), with the per-KeyLine work on the GPU through the stage entry points of include/edgehip.h and the 3..11
// dimensional filters on the host (rebvo/imu.h).
//
// Not rebuilt: the pose-graph log (cf->poses.addFrameMeas, :325-336) and key frames — neither feeds back into the
// estimate.  W (the 3-vector the non-IMU branch estimates) stays zero in this branch, as in the reference.

#include <cmath>
#include <cstring>
#include <iostream>

#include "edgehip.h"
#include "rebvo/rebvo.h"

namespace rebvo {

using la::Mat;
using la::Vec;

namespace {
constexpr double RHO_MAX = 20, RHO_MIN = 1e-3;   // include/mtracklib/edge_tracker.h:37-38

inline Vec<3> lav(const Vector3 &v) { Vec<3> r; for (int i = 0; i < 3; i++) r[i] = v[i]; return r; }
inline Vector3 v3(const Vec<3> &v) { Vector3 r = Zeros3(); for (int i = 0; i < 3; i++) r[i] = v[i]; return r; }
inline Matrix3x3 m3(const Mat<3, 3> &m) { Matrix3x3 r = Identity3(); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = m(i, j); return r; }
}  // namespace

// SecondThread's locals that persist from frame to frame in the IMU branch, in la:: types
struct REBVO::ImuTrack {
    int n_frame = 0;          // frame pairs processed (SecondThread's n_frame)
    double t_prev = 0;
    double Kp = 1, K = 1, P_Kp = 5e-6;
    Vec<3> V = Vec<3>::zeros(), W = Vec<3>::zeros(), Pos = Vec<3>::zeros();
    Mat<3, 3> Pose = Mat<3, 3>::identity(), Rgva = Mat<3, 3>::identity();
    // IMUState in la:: types
    Vec<3> Vg = Vec<3>::zeros(), Bg = Vec<3>::zeros(), Av = Vec<3>::zeros(), As = Vec<3>::zeros();
    Vec<3> dVv, dWv, dVgv, dWgv, Vgv, Wgv, dVgva, dWgva, Vgva;
    Mat<3, 3> P_Vg = Mat<3, 3>::identity(1e50), RGiro = Mat<3, 3>::identity(), RGBias = Mat<3, 3>::identity();
    Mat<3, 3> W_Bg, Qrot = Mat<3, 3>::identity(), Qg, Qbias, Rs, Rv = Mat<3, 3>::identity();
    Vec<7> X;
    Mat<7, 7> P;
    double QKp = 0, Rg = 0;
    Vec<3> g_est = Vec<3>::zeros(), u_est, b_est = Vec<3>::zeros(), Posgv = Vec<3>::zeros(), Posgva = Vec<3>::zeros();
    bool init = false;
    int n_giro_init = 0;
    Vec<3> giro_init = Vec<3>::zeros(), g_init = Vec<3>::zeros();
    ScaleEstimator se;
    ImuTrack() {
        dVv = dWv = dVgv = dWgv = Vgv = Wgv = dVgva = dWgva = Vgva = Vec<3>::zeros();
    }
};

// rebvo_second_t.cpp:68-84
void REBVO::imuTrackInit() {
    imuTrackFree();
    imutrack = new ImuTrack;
    ImuTrack &s = *imutrack;
    const REBVOParameters &p = params;
    s.W_Bg = la::inv3(s.RGBias * 100.0);
    s.Qg = Mat<3, 3>::identity() * p.g_uncert * p.g_uncert;
    s.Rg = p.g_module_uncer * p.g_module_uncer;
    s.Rs = Mat<3, 3>::identity() * p.AcelMeasStdDev * p.AcelMeasStdDev;
    s.Qbias = Mat<3, 3>::identity() * p.VBiasStdDev * p.VBiasStdDev;
    s.X = Vec<7>::zeros();
    s.X[0] = M_PI / 4;
    s.X[2] = p.g_module;
    s.P = Mat<7, 7>::zeros();
    s.P(0, 0) = p.ScaleStdDevInit * p.ScaleStdDevInit;
    s.P(1, 1) = s.P(2, 2) = s.P(3, 3) = 100;
    s.P(4, 4) = s.P(5, 5) = s.P(6, 6) = p.VBiasStdDev * p.VBiasStdDev * 1e1;
    s.u_est = Vec<3>::zeros();
    s.u_est[0] = 1;
}

void REBVO::imuTrackFree() {
    delete imutrack;
    imutrack = nullptr;
}

bool REBVO::setCamImuSE3(const Matrix3x3 &RCam2IMU, const Vector3 &TCam2IMU) {
    if (!imu) return false;
    Mat<3, 3> R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = RCam2IMU(i, j);
    return imu->LoadCamImuSE3(R, lav(TCam2IMU));
}
Matrix3x3 REBVO::getCam2ImuRot() { return imu ? m3(imu->RDataSetCam2IMU) : Identity3(); }
Vector3 REBVO::getCam2ImuPos() { return imu ? v3(imu->TDataSetCam2IMU) : Zeros3(); }

// rebvo_second_t.cpp:609-620 in this branch: the newest edge map's depth on the device, the trajectory on the host
void REBVO::resetImuTrack(int slot_new) {
    edgehip_depth_reset_slot(hip, -1, slot_new);
    if (!imutrack) return;
    imutrack->Pose = Mat<3, 3>::identity();
    imutrack->Pos = imutrack->W = imutrack->V = Vec<3>::zeros();
}

#define EH(x) do { if ((rc = (x)) != 0) return rc; } while (0)

int REBVO::trackFrameImu(int sn, int so, bool have_pair, double t, PipeBuffer &new_buf) {
    const REBVOParameters &p = params;
    ImuTrack &s = *imutrack;
    int rc = 0;
    EH(edgehip_stage_a(hip, sn));   // FirstThr: scale space, KeyLines, auto threshold (rebvo_first_t.cpp:259-272)
    if (p.StereoAvaiable) EH(edgehip_stage_a(hip, 3));   // the pair image (slot behind the ring), :275-290
    new_buf.stereo_match_num = 0;
    int32_t kn = 0;
    edgehip_seq_state st;
    if (!have_pair) {               // "dummy processing of the first frame" (rebvo_second_t.cpp:108-121)
        EH(edgehip_get_state(hip, 0, &st));
        new_buf.ef->reTunedThresh = st.retuned_thresh;
        new_buf.ef->nmatch = 0;
        new_buf.nav = NavData();
        new_buf.dt = 0; new_buf.K = 1; new_buf.Kp = 1; new_buf.RKp = 0; new_buf.s_rho_p = 0;
        new_buf.EstimationOK = false;
        s.t_prev = t;
        return 0;
    }
    const IntegratedImuData &imud = new_buf.imu;
    bool EstimationOk = true;
    double dt_frame = t - s.t_prev;                                            // :145-148
    if (dt_frame < 0.001) dt_frame = 1 / p.config_fps;
    int klm_num = 0;
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    Mat<3, 3> P_V = I3 * 1e50, P_W = I3 * 1e50, R = I3;                        // :166-168
    Vec<3> &V = s.V;

    EH(edgehip_quantile(hip, so, RHO_MIN, RHO_MAX, p.QCutOffQuantile, (int)p.QCutOffNumBins));   // :172
    EH(edgehip_get_state(hip, 0, &st));
    const double s_rho_q = st.s_rho_q;
    new_buf.ef->reTunedThresh = st.retuned_thresh;
    EH(edgehip_build_field(hip, sn, (int)p.SearchRange, -1.f));               // :177

    // ---- gyro bias start-up (:183-203) ----
    if (!s.init && s.n_frame > 0) {
        if (p.InitBias) {
            s.giro_init = s.giro_init + imud.giro * imud.dt;
            s.g_init = s.g_init - imud.cacel;
            if (++s.n_giro_init > p.InitBiasFrameNum) {
                s.Bg = s.giro_init / (double)s.n_giro_init;
                s.init = true;
                s.W_Bg = la::inv3(s.RGBias * 1e2);
                la::set_slice(s.X, 1, s.g_init / (double)s.n_giro_init);
            }
        } else {
            s.init = true;
            s.Bg = lav(p.BiasInitGuess) * imud.dt;
        }
    }

    // ---- gyro pre-rotation, translation-only minimisation, forward match, linear roto-translation (:208-237) ----
    R = imud.Rot;
    R = la::transpose(la::so3_exp(s.Bg) * la::transpose(R));                  // R.T() = SO3(Bg) * R.T()
    {
        const Mat<3, 3> Rt = la::transpose(R);
        EH(edgehip_rotate_keylines(hip, so, Rt.a));                            // forward pre-rotation of the old KeyLines
    }
    if (p.TrackerInitType == 0) s.Vg = Vec<3>::zeros();
    {
        double F = 0;
        EH(edgehip_minimizer_v(hip, sn, so, s.Vg.v, &s_rho_q, -1.f, p.TrackerMatchThresh, p.TrackerIterNum, p.MatchNumThresh,
                               p.ReweigthDistance, s.P_Vg.a, &F));
    }
    EH(edgehip_forward_match(hip, so, sn));
    Vec<6> Xv;
    Mat<6, 6> W_Xv, R_Xv;
    {
        int32_t ok = 0;
        EH(edgehip_ext_rot_vel(hip, sn, s.Vg.v, p.LocationUncertainty, p.ReweigthDistance, Xv.v, W_Xv.a, R_Xv.a, &ok));
        EstimationOk &= ok != 0;
    }
    s.dVv = la::slice<3>(Xv, 0);
    s.dWv = la::slice<3>(Xv, 3);
    Vec<6> Xgv = Xv;
    Mat<6, 6> W_Xgv = W_Xv;

    // ---- gyro prior (:247-272) ----
    s.RGBias = I3 * p.GiroBiasStdDev * p.GiroBiasStdDev * dt_frame * dt_frame;
    s.RGiro = I3 * p.GiroMeasStdDev * p.GiroMeasStdDev * dt_frame * dt_frame;
    Vec<3> dgbias = Vec<3>::zeros();
    imufilter::BiasCorrect(Xgv, W_Xgv, dgbias, s.W_Bg, s.RGiro, s.RGBias);
    s.Bg = s.Bg + dgbias;
    s.dVgv = la::slice<3>(Xgv, 0);
    s.dWgv = la::slice<3>(Xgv, 3);
    s.Rgva = R;                                                               // previous matrix
    const Mat<3, 3> R0 = la::so3_exp(s.dWgv);                                 // forward rotation
    R = la::transpose(R0 * la::transpose(R));                                 // R is a backward rotation
    s.Vgv = R0 * s.Vg + s.dVgv;
    V = s.Vgv;
    s.Wgv = la::so3_ln(R);
    const Mat<6, 6> R_Xgv = la::Cholesky<6>(W_Xgv).inverse();
    P_V = la::block<3, 3>(R_Xgv, 0, 0);
    P_W = la::block<3, 3>(R_Xgv, 3, 3);

    // ---- accelerometer / scale filter (:280-312) ----
    s.se.EstAcelLsq4((-s.Vgv) / dt_frame, s.Av, R, dt_frame);
    s.se.MeanAcel4(imud.cacel, s.As, R);
    Vec<6> Xgva = Xgv;
    s.Rv = P_V / (dt_frame * dt_frame * dt_frame * dt_frame);
    s.Qrot = P_W;
    s.QKp = s.P_Kp;
    if (s.n_frame > 4 + p.InitBiasFrameNum) {
        s.K = ScaleEstimator::estKaGMEKBias(s.As, s.Av, 1, R, s.X, s.P, s.Qg, s.Qrot, s.Qbias, s.QKp, s.Rg, s.Rs, s.Rv, s.g_est,
                                            s.b_est, W_Xgv, Xgva, p.g_module);
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
    EH(edgehip_rotate_keylines(hip, so, R0.a));                               // forward-rotate the old KeyLines (:319)

    // ---- mapper (:387-493): the device stages read V, P_V, R, P_W, P_Kp from the sequence state ----
    if (la::has_nan(V) || la::has_nan(s.W)) {
        P_V = I3 * 1e50;
        V = Vec<3>::zeros();
        s.Kp = 1;
        s.P_Kp = 1e50;
        EstimationOk = false;
        EH(edgehip_get_kn(hip, sn, &kn));
        std::printf("\nCamara Frontal: error in the estimation, not many KeyLines (%d)?\n", kn);
    } else {
        EH(edgehip_get_state(hip, 0, &st));
        std::memcpy(st.V, V.v, sizeof st.V);
        std::memcpy(st.W, s.W.v, sizeof st.W);
        std::memcpy(st.P_V, P_V.a, sizeof st.P_V);
        std::memcpy(st.P_W, P_W.a, sizeof st.P_W);
        std::memcpy(st.R, R.a, sizeof st.R);
        st.Kp = s.Kp;
        st.P_Kp = s.P_Kp;
        st.klm_num = 0;
        st.kf_matchs = 0;
        EH(edgehip_set_state(hip, 0, &st));
        EH(edgehip_directed_matching(hip, sn, so));                           // :410
        EH(edgehip_get_state(hip, 0, &st));
        klm_num = st.klm_num;
        if (klm_num < p.MatchThreshold) {                                     // :412-422
            P_V = I3 * 1e50;
            V = Vec<3>::zeros();
            s.Kp = 1;
            s.P_Kp = 10;
            EstimationOk = false;
            EH(edgehip_get_kn(hip, sn, &kn));
            std::printf("\nCamara Frontal: restarting the estimation, match threshold low (%d,%d)?\n", kn, klm_num);
        } else {
            EH(edgehip_regularize_ekf(hip, sn, 1, 1));                        // :453, :460
            if (p.StereoAvaiable) {                                           // :465-486
                int32_t nm = 0;
                EH(edgehip_directed_matching_stereo(hip, sn, 3, kTCam2Pair, kRCam2Pair, p.MatchThreshModule, p.MatchThreshAngle, 100,
                                                    p.LocationUncertaintyMatch, p.ReshapeQAbsolute, p.ReshapeQRelative,
                                                    p.LocationUncertainty, &nm));
                new_buf.stereo_match_num = nm;
                EH(edgehip_fuse_stereo_depth(hip, sn));
                s.Kp = 1;
            } else {
                EH(edgehip_rescale(hip, sn));                                 // :487
                EH(edgehip_get_state(hip, 0, &st));
                s.Kp = st.Kp;
                s.P_Kp = st.P_Kp;
            }
        }
    }

    // ---- pose (:519-544): with the IMU, gravity fixes two axes and u_est carries the heading ----
    if (s.n_frame > 4 + p.InitBiasFrameNum) {
        s.u_est = la::transpose(s.Rgva) * s.u_est;
        s.u_est = s.u_est - s.g_est * (la::dot(s.u_est, s.g_est) / la::dot(s.g_est, s.g_est));
        s.u_est = s.u_est / std::sqrt(la::dot(s.u_est, s.u_est));             // TooN::normalize
        Vec<3> ey = Vec<3>::zeros(), ex = Vec<3>::zeros();
        ey[1] = 1; ex[0] = 1;
        const Mat<3, 3> PoseP1 = la::so3_from_to(s.g_est, ey);
        const Mat<3, 3> PoseP2 = la::so3_from_to(PoseP1 * s.u_est, ex);
        s.Pose = PoseP2 * PoseP1;
        s.Pos = s.Pos + (-s.Pose) * s.Vgva * s.K;
        s.Posgva = s.Pos;
        s.Posgv = s.Posgv + (-s.Pose) * s.Vgv * s.K;
    }

    // ---- hand-over (:550-583) ----
    new_buf.dt = dt_frame;
    new_buf.K = s.K;
    new_buf.Kp = s.Kp;
    new_buf.RKp = s.P_Kp;
    NavData &nav = new_buf.nav;
    nav.dt = dt_frame;
    nav.t = t;
    nav.Rot = m3(R);
    nav.RotLie = v3(la::so3_ln(R));
    nav.RotGiro = v3(la::so3_ln(s.Rgva) / dt_frame);
    nav.Vel = v3(((-V) * s.K) / dt_frame);
    nav.Pose = m3(s.Pose);
    nav.PoseLie = v3(la::so3_ln(s.Pose));
    nav.Pos = v3(s.Pos);
    nav.g = v3(s.g_est);
    nav.scale = s.K;
    new_buf.s_rho_p = s_rho_q;
    new_buf.EstimationOK = EstimationOk;
    new_buf.ef->nmatch = klm_num;
    IMUState &is = new_buf.imustate;
    is.Vg = v3(s.Vg); is.dVv = v3(s.dVv); is.dWv = v3(s.dWv); is.dVgv = v3(s.dVgv); is.dWgv = v3(s.dWgv);
    is.Vgv = v3(s.Vgv); is.Wgv = v3(s.Wgv); is.dVgva = v3(s.dVgva); is.dWgva = v3(s.dWgva); is.Vgva = v3(s.Vgva);
    is.P_Vg = m3(s.P_Vg); is.RGiro = m3(s.RGiro); is.RGBias = m3(s.RGBias); is.Bg = v3(s.Bg); is.W_Bg = m3(s.W_Bg);
    is.Av = v3(s.Av); is.As = v3(s.As); is.X = s.X; is.P = s.P; is.Qrot = m3(s.Qrot); is.Qg = m3(s.Qg); is.Qbias = m3(s.Qbias);
    is.QKp = s.QKp; is.Rg = s.Rg; is.Rs = m3(s.Rs); is.Rv = m3(s.Rv); is.g_est = v3(s.g_est); is.u_est = v3(s.u_est);
    is.b_est = v3(s.b_est); is.Posgv = v3(s.Posgv); is.Posgva = v3(s.Posgva); is.init = s.init;
    s.t_prev = t;
    s.n_frame++;
    return 0;
}

}  // namespace rebvo
