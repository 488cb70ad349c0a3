/* oracle_abi.h — C ABI shared by the two CPU oracles (TEST INFRASTRUCTURE ONLY).
 *
 *   ref_*  : oracle/_ref/libreforacle.so — the reference's own mtracklib classes, compiled in place from
 *            /root/reference and driven by oracle/ref_harness.cpp.
 *   port_* : oracle/libedgeport.so — our plain C++ restatement of the same algorithm (oracle/port/).
 *
 * Both export the same entry points with the prefix swapped, so tests can run either behind one Python
 * wrapper (oracle/oracle.py).  Nothing under rebvo_amd/ may include, link or load anything from oracle/.
 */
#ifndef ORACLE_ABI_H
#define ORACLE_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Byte-for-byte mirror of rebvo::KeyLine (reference include/mtracklib/edge_finder.h:45-91), 168 B. */
typedef struct OrcKeyLine {
    int32_t p_inx;
    float m_m[2];
    float u_m[2];
    float n_m;
    float score;
    float c_p[2];
    double rho, s_rho, rho_nr, s_rho_nr, rho0, s_rho0;
    float p_m[2];
    float p_m_0[2];
    int32_t m_id, m_id_f, m_id_kf, m_num;
    float m_m0[2];
    double n_m0;
    int32_t p_id, n_id, net_id;
    int32_t stereo_m_id;
    double stereo_rho, stereo_s_rho;
} OrcKeyLine;

/* Subset of rebvo::REBVOParameters (reference include/rebvo/rebvo.h:64-235) that reaches the hot path. */
typedef struct OrcParams {
    int32_t w, h;
    double ppx, ppy, zfx, zfy;
    double kc[5];
    double sigma0, ksigma;
    int32_t plane_fit_size;
    double pos_neg_thresh, dog_thresh;
    int32_t max_points, reference_points, track_points;
    double detector_thresh, auto_gain, max_thresh, min_thresh;
    int32_t search_range, qcut_nbins;
    double qcut_quantile;
    int32_t tracker_iter_num, tracker_init_type, tracker_init_iter_num;
    double tracker_match_thresh, match_thresh_module, match_thresh_angle;
    uint32_t match_num_thresh;
    int32_t do_rescaling;
    double reweight_distance, regularize_thresh;
    double loc_unc_match, reshape_q_abs, reshape_q_rel, loc_unc;
    int32_t global_match_threshold;
    int32_t use_undistort;   /* Camera/UseUndistort: image_undistort::undistort<true> before RGB->grey */
    double config_fps;
} OrcParams;

/* Per-frame record: what SecondThread leaves in PipeBuffer/NavData (rebvo_second_t.cpp:550-606). */
typedef struct OrcNav {
    double t, dt;
    double V[3], W[3];          /* tracker output of this frame (state carried to the next) */
    double P_V[9], P_W[9];      /* RVel, RW0 as returned by Minimizer_RV (before the /dt^2) */
    double Rot[9], RotLie[3], Vel[3], Pose[9], PoseLie[3], Pos[3];
    double Kp, RKp, s_rho_q;
    double tresh;               /* detector threshold after UpdateThresh */
    double score, rel_error, rel_error_score;
    double dtp0, dtp1;          /* CPU seconds spent in stage A / stage B+C */
    float retuned_thresh;
    int32_t kn, klm_fwd, klm_num, kf_matchs, estimation_ok, frame;
    int32_t pad0;
} OrcNav;

/* One 6x6 (or smaller) decomposition the minimiser asked for, in call order (parity diagnostics: the systems of
 * Minimizer_RV's init phase, global_tracker.cpp:659-661, 710-712, and their singular values as the back end returned them). */
typedef struct OrcSvdRec {
    int32_t rows, cols;
    double A[36];   /* row-major, leading dimension 6 */
    double s[6];    /* ref: singular values, descending; port: eigenvalues in Jacobi order (symmetric input) */
} OrcSvdRec;

/* The FirstThr / SecondThread locals that persist from frame to frame (rebvo_first_t.cpp:92-94, rebvo_second_t.cpp:54-66):
 * what a teacher-forced replay hands to the path under test before every frame. */
typedef struct OrcSeqState {
    double tresh, t_prev, Kp, K, P_Kp;
    double V[3], W[3], Pos[3], Pose[9];
    int32_t l_kl_num, frame;
} OrcSeqState;

#define ORC_DECLARE(P)                                                                                   \
    void *P##_create(const OrcParams *p, int nslots);                                                    \
    void P##_destroy(void *ctx);                                                                         \
    /* stage A: ConvertRGB2BW + sspace::build + edge_finder::detect + reEstimateThresh */               \
    int P##_stage_a(void *ctx, int slot, const uint8_t *rgb24, double *tresh_io, int *l_kl_num_io);      \
    const float *P##_plane(void *ctx, int slot, int which); /* 0 img0 1 img1 2 dog 3 dx 4 dy 5 bw */    \
    const uint8_t *P##_imgc(void *ctx, int slot);           /* RGB24 frame stage A consumed */          \
    int P##_undistort_map(void *ctx, int32_t *inx, int32_t *iw); /* [n*4] each; -1 if not enabled */   \
    const int32_t *P##_mask(void *ctx, int slot);                                                        \
    int P##_kn(void *ctx, int slot);                                                                     \
    OrcKeyLine *P##_keylines(void *ctx, int slot);                                                       \
    float P##_retuned(void *ctx, int slot);                                                              \
    void P##_set_keylines(void *ctx, int slot, const OrcKeyLine *kl, int kn, const int32_t *mask,        \
                          float retuned);                                                                \
    unsigned P##_get_framecount(void *ctx, int slot);                                                    \
    void P##_set_framecount(void *ctx, int slot, unsigned fc);                                           \
    /* stage B */                                                                                        \
    double P##_quantile(void *ctx, int slot, double smin, double smax, double pct, int n);               \
    void P##_build_field(void *ctx, int slot, int radius, float min_mod);                                \
    const int32_t *P##_field(void *ctx, int slot); /* {dist, ikl} per pixel */                           \
    double P##_try_velrot(void *ctx, int slot_new, int slot_old, const double X[6], int reweight,        \
                          int procjf, double match_thresh, double s_rho_min, unsigned match_num_thresh,  \
                          double k_huber, const double *resid_in, double *resid_out, double JtJ[36],     \
                          double JtF[6]);                                                                \
    double P##_minimizer_rv(void *ctx, int slot_new, int slot_old, double V[3], double W[3],             \
                            double RVel[9], double RW0[9], double match_thresh, int iter_max,            \
                            int init_type, double reweight_distance, double *rel_error,                  \
                            double *rel_error_score, double max_s_rho, unsigned match_num_thresh,        \
                            double init_iter, double W_X[36]);                                           \
    /* global_tracker::Minimizer_V<double> (IMU branch): V in/out, RVel out; returns the score F */     \
    double P##_minimizer_v(void *ctx, int slot_new, int slot_old, double V[3], double RVel[9],           \
                           double match_thresh, int iter_max, double s_rho_min,                          \
                           unsigned match_num_thresh, double reweight_distance, float min_mod);          \
    /* edge_tracker::ExtRotVel (IMU branch): returns its bool; X[6], Wx[36], Rx[36] */                   \
    int P##_ext_rot_vel(void *ctx, int slot, const double vel[3], double loc_unc, double hub_reweight,   \
                        double X[6], double Wx[36], double Rx[36]);                                      \
    /* stage C */                                                                                        \
    int P##_forward_match(void *ctx, int slot_old, int slot_new);                                        \
    void P##_rotate_keylines(void *ctx, int slot, const double R[9]);                                    \
    int P##_directed_matching(void *ctx, int slot_new, int slot_old, const double V[3],                  \
                              const double RVel[9], const double BackRot[9], int *kf_matchs,             \
                              double min_thr_mod, double min_thr_ang, double max_radius,                 \
                              double loc_unc);                                                           \
    int P##_regularize(void *ctx, int slot, double thresh);                                              \
    void P##_ekf(void *ctx, int slot, const double V[3], const double RVel[9], const double RW0[9],      \
                 double q_abs, double q_rel, double loc_unc);                                            \
    double P##_rescale(void *ctx, int slot, double *RKp, double s_rho_min, unsigned match_num_min,       \
                       int re_escale);                                                                   \
    /* whole frame, non-IMU branch of FirstThr + SecondThread; state lives in ctx */                     \
    int P##_process_frame(void *ctx, const uint8_t *rgb24, double t, OrcNav *nav);                       \
    int P##_cur_slot(void *ctx);                                                                         \
    /* REBVO::Reset() as SecondThread runs it after a frame (rebvo_second_t.cpp:609-620) */              \
    void P##_depth_reset(void *ctx);                                                                     \
    void P##_reset_sequence(void *ctx);                                                                  \
    void P##_get_seq_state(void *ctx, OrcSeqState *out);                                                 \
    /* record the next `cap` decompositions into buf (NULL stops); process-wide, not thread-safe */     \
    void P##_svd_trace(OrcSvdRec *buf, int cap);                                                         \
    int P##_svd_trace_count(void);

ORC_DECLARE(ref)
ORC_DECLARE(port)

/* (_ref only) the two 6x6 solves of Minimizer_RV exactly as it spells them: h = TooN::SVD<>(A).backsub(b)
 * (global_tracker.cpp:660-661; condition_no = 1e9, TooN/SVD.h:37,179) and h = TooN::Cholesky<6>(A).backsub(b) (:767-768). */
void ref_svd_backsub(const double A[36], const double b[6], double h[6]);
void ref_chol_backsub(const double A[36], const double b[6], double h[6]);
/* (_ref only) which dgesvd_ serves TooN::SVD<>: 0 = LAPACK (MKL), 1 = the harness's one-sided Jacobi.  Returns the old one. */
int ref_svd_backend(int which);

/* ---- key-frame tracker: reference only.  kfvo::Minimizer_RV_KF<double,false> (kfvo.cpp:1679-1825),
 * called directly with the arguments kfvo::OptimizePosGT passes (kfvo.cpp:74): gt = slot_kf's global_tracker (its field as
 * built for that frame), klist = slot_cur's edge_tracker.  X in/out ([translation, rotation]); returns F/F0. ---- */
double ref_minimizer_rv_kf(void *ctx, int slot_kf, int slot_cur, double X[6], double Kr, double match_mod, double match_ang,
                           double rho_tol, int iter_max, double reweight_distance, double max_s_rho, unsigned match_num_thresh,
                           double RRV[36], int *mnum);

/* ---- stereo depth (REBVO/StereoAvaiable): reference only ---- */
void ref_set_slot_cam(void *ctx, int slot, double ppx, double ppy, double zfx, double zfy);   /* pair camera intrinsics */
void ref_set_tracker_f32(void *ctx, int on);   /* reference oracle only: Minimizer_RV<float> (global_tracker.cpp:824, USE_NE10) instead of <double>,
                                                  in ref_minimizer_rv and in the whole-frame drivers */
void ref_set_stereo_mode(void *ctx, int on);   /* the stereo_mode argument ref_directed_matching passes on */
int ref_directed_matching_stereo(void *ctx, int slot, int slot_pair, const double t[3], const double R[9], double min_thr_mod,
                                 double min_thr_ang, double max_radius, double loc_unc, double q_abs, double q_rel,
                                 double loc_unc_model);
void ref_fuse_stereo_depth(void *ctx, int slot);
/* whole frame with StereoAvaiable: pair camera + rig, then ref_process_frame_stereo per frame pair of images;
 * OrcNav::pad0 carries stereo_match_num */
void ref_enable_stereo(void *ctx, double ppx, double ppy, double zfx, double zfy, const double t[3], const double R[9],
                       double max_radius);
int ref_process_frame_stereo(void *ctx, const uint8_t *rgb24, const uint8_t *rgb24_pair, double t, OrcNav *nav);
/* (_ref only) replay n frames out of a frame pool and time them: threads 1 = stage A + B/C back to back, 2 = the reference's
   FirstThr / SecondThread overlap; done_s[k] = seconds after the start at which frame k was finished */
int ref_run_sequence(void *ctx, const uint8_t *pool, unsigned long long frame_bytes, const int *idx, int n, double t0, double dt,
                     int threads, double *done_s, OrcNav *navs);

/* ---- visualizer wire format: copy_net_keyline / copy_net_keyline_nextid, 15-byte records ---- */
int ref_copy_net_keyline(void *ctx, int slot, int slot_pair /* -1: none */, void *out, int kl_size, double k_prof);
int ref_copy_net_keyline_nextid(void *ctx, int slot, void *out, int kl_size);

/* ---- IMU branch: reference only.  The restatement under test is the host library
 * (rebvo_amd/host/src/imu.cpp, rebvo_imu.cpp); these entry points run the reference's own ImuGrabber, BiasCorrect and
 * ScaleEstimator, and ref_process_frame_imu restates the ImuMode > 0 sequencing of rebvo_second_t.cpp over them.
 * NOTE: ScaleEstimator::EstAcelLsq4 / MeanAcel4 keep their histories in function-local statics — one per process;
 * tests that need a clean history run the oracle in a fresh process. */
typedef struct OrcImuIntegrated {   /* rebvo::IntegratedImuData */
    int32_t n, pad;
    double dt, Rot[9], giro[3], acel[3], comp[3], dgiro[3], cacel[3];
} OrcImuIntegrated;

typedef struct OrcImuParams {       /* the IMU members of REBVOParameters (include/rebvo/rebvo.h:150-173) */
    double giro_meas_std, giro_bias_std;
    int32_t init_bias, init_bias_frame_num;
    double bias_init_guess[3];
    double acel_meas_std, g_module, g_module_uncer, g_uncert, vbias_std;
    double scale_std_mult, scale_std_max, scale_std_init;
} OrcImuParams;

typedef struct OrcNavImu {          /* NavData + the IMUState members worth comparing */
    double Rot[9], RotLie[3], RotGiro[3], Vel[3], Pose[9], PoseLie[3], Pos[3], g[3], scale;
    double dt, K, Kp, RKp, s_rho_q;
    double Vg[3], Bg[3], dVv[3], dWv[3], Vgv[3], Vgva[3], Av[3], As[3], X[7], b_est[3], u_est[3];
    int32_t kn, klm_num, estimation_ok, init;
} OrcNavImu;

void ref_imu_bias_correct(double *X, double *Wx, double *Gb, double *Wb, const double *Rg, const double *Rb);
void ref_est_acel_lsq4(const double *vel, double *acel, const double *R, double dt);
void ref_mean_acel4(const double *s_acel, double *acel, const double *R);
double ref_est_ka_gmek_bias(const double *s_acel, const double *f_acel, double kP, const double *Rot, double *X, double *P,
                            const double *Qg, const double *Qrot, const double *Qbias, double QKp, double Rg,
                            const double *Rs, const double *Rf, double *g_est, double *b_est, const double *Wvw,
                            double *Xvw, double g_gravit);
void *ref_imu_grabber_new(int list_size, double tsamp);
void *ref_imu_grabber_load(const char *csv_file, double time_scale);
void ref_imu_grabber_free(void *g);
int ref_imu_grabber_set_se3(void *g, const double *R, const double *T);
int ref_imu_grabber_load_se3(void *g, const char *se3_file);
int ref_imu_grabber_push(void *g, double tstamp, const double *giro, const double *acel);
void ref_imu_grabber_grab(void *g, double tstart, double tend, OrcImuIntegrated *out);
double ref_imu_grabber_tsample(void *g);
/* whole frame, ImuMode > 0 branch; imu = the inter-frame data FirstThr grabbed for this frame */
void ref_imu_setup(void *ctx, const OrcImuParams *ip);
int ref_process_frame_imu(void *ctx, const uint8_t *rgb24, double t, const OrcImuIntegrated *imu, OrcNavImu *nav);

#ifdef __cplusplus
}
#endif
#endif
