/* edgehip.h — C ABI of libedgehip.so: REBVO's per-frame edge pipeline as hand-written HIP for gfx950.
 *
 * This is the drop-in boundary.  Every entry point replaces one call that the reference's
 * REBVO::FirstThr / REBVO::SecondThread make into mtracklib; the file:line each one replaces is cited at
 * its declaration (paths relative to the reference tree).  Plain pointers and sizes only — no C++ or
 * torch types — so the same library binds from C++ (rebvo_amd/host), ctypes (rebvo_amd/edgehip.py) or
 * any other FFI.
 *
 * Model.  One context = `nseq` independent image sequences that advance in lock-step on ONE GPU (the
 * sequence index is the batch dimension of every kernel launch), each with a ring of `nslots` frame
 * slots — the device-side counterpart of the reference's PipeBuffer ring (src/rebvo/rebvo.cpp:297-312).
 * All KeyLine lists, masks, auxiliary fields and the tracker/mapper state live in HBM as
 * structure-of-arrays; the 168-byte AoS `KeyLine` is materialised only by edgehip_download_keylines()
 * for callback consumers and parity tests.
 *
 * Calls enqueue work on the context's HIP stream and return immediately unless documented as
 * synchronising.  Every function returns 0 on success or a negative edgehip_status; the message of the
 * last failure on the calling thread is available from edgehip_last_error().  Nothing here ever falls
 * back to a CPU implementation: if no gfx950 device is usable, edgehip_create() fails.
 */
#ifndef EDGEHIP_H
#define EDGEHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDGEHIP_ABI_VERSION 2
#define EDGEHIP_KEYLINE_MAX 50000 /* KEYLINE_MAX, include/mtracklib/edge_finder.h:43 */

typedef enum edgehip_status {
    EDGEHIP_OK = 0,
    EDGEHIP_ERR_ARG = -1,     /* bad argument (slot/sequence out of range, null pointer, size mismatch) */
    EDGEHIP_ERR_DEVICE = -2,  /* no usable gfx950 device / HIP runtime error */
    EDGEHIP_ERR_MEMORY = -3,  /* hipMalloc / hipHostMalloc failed */
    EDGEHIP_ERR_STATE = -4    /* call sequence error (e.g. track before two frames were detected) */
} edgehip_status;

typedef struct edgehip_ctx edgehip_ctx;

/* The REBVOParameters fields (include/rebvo/rebvo.h:64-235) that reach the hot path.  Same meaning and
 * units as the reference's config keys (app/rebvorun/GlobalConfig_EuRoC). */
typedef struct edgehip_params {
    int32_t w, h;                 /* ImageWidth, ImageHeight */
    double ppx, ppy, zfx, zfy;    /* PPx, PPy, ZfX, ZfY (stored as float like REBVOParameters does) */
    double kc[5];                 /* KcR2 KcR4 KcR6 KcP1 KcP2 */
    double sigma0, ksigma;        /* Sigma0, KSigma */
    int32_t plane_fit_size;       /* DetectorPlaneFitSize: 1, 2 or 3 = a 3x3, 5x5 or 7x7 plane-fit window (edge_finder.cpp:110-137) */
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
    int32_t debug_planes;         /* !=0: also store img0/img1/dog/dx/dy planes (parity tests) */
    double config_fps;
    /* UseUndistort: resample every input frame through the radial-tangential model `kc` before RGB->grey,
     * i.e. image_undistort::undistort<true> of rebvo_first_t.cpp:231 (include/VideoLib/image_undistort.h:
     * 66-79, 105-122; map as built by src/VideoLib/image_undistort.cpp:29-95), fused into the first
     * stage-A kernel: the bilinear map (4 taps, 16.16 integer weights) is built once at create time. */
    int32_t use_undistort;
    /* REBVO/StereoAvaiable: allocate the stereo KeyLine fields (stereo_m_id, stereo_rho, stereo_s_rho), and run
     * directed_matching in its stereo mode (the match copies rho0/s_rho0 instead of rho/s_rho and leaves rho_nr alone,
     * edge_tracker.cpp:343-351).  The pair image's edge map lives in a ring slot of its own (see
     * edgehip_directed_matching_stereo). */
    int32_t stereo_available;
} edgehip_params;

/* Byte-for-byte the reference's rebvo::KeyLine (include/mtracklib/edge_finder.h:45-91), 168 B. */
typedef struct edgehip_keyline {
    int32_t p_inx;
    float m_m[2], u_m[2], n_m, score, c_p[2];
    double rho, s_rho, rho_nr, s_rho_nr, rho0, s_rho0;
    float p_m[2], p_m_0[2];
    int32_t m_id, m_id_f, m_id_kf, m_num;
    float m_m0[2];
    double n_m0;
    int32_t p_id, n_id, net_id, stereo_m_id;
    double stereo_rho, stereo_s_rho;
} edgehip_keyline;

/* Per-sequence tracker/mapper state: the locals of FirstThr (rebvo_first_t.cpp:92-94) and SecondThread
 * (rebvo_second_t.cpp:54-66) that persist from frame to frame, kept in HBM so that a whole frame can be
 * enqueued without a host round trip. */
typedef struct edgehip_seq_state {
    double tresh;               /* detector threshold (P-controller state) */
    double V[3], W[3];          /* velocity / rotation estimate carried to the next frame */
    double P_V[9], P_W[9];      /* RVel, RW0 of the last Minimizer_RV */
    double R[9];                /* back-rotation of the last frame pair */
    double Pose[9], Pos[3];     /* integrated pose */
    double Kp, P_Kp, K;
    double s_rho_q;             /* EstimateQuantile result of the last frame pair */
    double score, rel_error, rel_error_score;
    double t_prev, dt;
    float retuned_thresh;       /* edge_finder::reTunedThresh of the newest slot */
    int32_t l_kl_num;           /* KeyLines on the last edge map (P-controller input) */
    int32_t frame;              /* frames detected so far */
    int32_t klm_fwd;            /* KeyLines of the new edge map that received a forward match.  (edge_tracker::FordwardMatch returns the
                                   number of assignments it made, the ones it later overwrote on a double match included; SecondThread
                                   overwrites that value with directed_matching's before anything reads it, rebvo_second_t.cpp:354 / 410.) */
    int32_t klm_num, kf_matchs; /* directed_matching: matched KeyLines, key-frame matches among them */
    int32_t estimation_ok;
    int32_t minimizer_evals;    /* TryVelRot evaluations spent on the last frame pair */
} edgehip_seq_state;

/* What SecondThread leaves in PipeBuffer/NavData per frame (rebvo_second_t.cpp:550-606). */
typedef struct edgehip_nav {
    double t, dt;
    double V[3], W[3], P_V[9], P_W[9];
    double Rot[9], RotLie[3], Vel[3], Pose[9], PoseLie[3], Pos[3];
    double Kp, RKp, s_rho_q, tresh, score, rel_error, rel_error_score;   /* tresh: FirstThr's threshold state behind the frame's detection(s) —
                                                                           with a stereo rig, behind the pair image's (rebvo_first_t.cpp:266-290) */
    float retuned_thresh;
    int32_t kn, klm_fwd, klm_num, kf_matchs, estimation_ok, frame, minimizer_evals;
} edgehip_nav;

/* ---- IMU branch on the device (ImuMode > 0 for whole batches) -------------------------------------------------------
 * The &IMU parameters REBVO::SecondThread uses (include/rebvo/rebvo.h:172-199; GlobalConfig_EuRoC:107-140). */
typedef struct edgehip_imu_params {
    double giro_meas_std, giro_bias_std;       /* GiroMeasStdDev, GiroBiasStdDev */
    int32_t init_bias, init_bias_frame_num;    /* InitBias, InitBiasFrameNum */
    double bias_init_guess[3];                 /* BiasHintX/Y/Z */
    double acel_meas_std, g_module, g_module_uncer, g_uncert, vbias_std;   /* AcelMeasStdDev, g_module, g_module_uncer, g_uncert, VBiasStdDev */
    double scale_std_mult, scale_std_max, scale_std_init;                  /* ScaleStdDevMult, ScaleStdDevMax, ScaleStdDevInit */
} edgehip_imu_params;
/* rebvo::IntegratedImuData (include/UtilLib/imugrabber.h:57-69): what ImuGrabber::GrabAndIntegrate hands the tracker
 * for the interval between two frames. */
typedef struct edgehip_imu_integrated {
    int32_t n, pad;
    double dt, Rot[9], giro[3], acel[3], comp[3], dgiro[3], cacel[3];
} edgehip_imu_integrated;
/* What the IMU branch adds to the per-frame record: NavData's IMU fields (rebvo.h:292-308) and the IMUState members the
 * reference logs (rebvo_second_t.cpp:550-606). */
typedef struct edgehip_nav_imu {
    double Rot[9], RotLie[3], RotGiro[3], Vel[3], Pose[9], PoseLie[3], Pos[3], g[3];
    double scale, dt, K, Kp, RKp, s_rho_q;
    double Vg[3], Bg[3], dVv[3], dWv[3], Vgv[3], Vgva[3], Av[3], As[3], X[7], b_est[3], u_est[3];
    int32_t kn, klm_num, estimation_ok, init;
} edgehip_nav_imu;

/* ---- lifetime -------------------------------------------------------------------------------------- */
/* Replaces the per-slot `new sspace / new edge_tracker / new global_tracker` of REBVO::construct
 * (src/rebvo/rebvo.cpp:297-312).  `device` is the HIP device ordinal. */
int edgehip_create(const edgehip_params *params, int nseq, int nslots, int device, edgehip_ctx **out);
int edgehip_destroy(edgehip_ctx *ctx);
const char *edgehip_last_error(void);
int edgehip_abi_version(void);
/* 1 if the library was built with `make EXPERIMENTS=1`: it then also holds the alternative kernels that measured slower than the
 * defaults and reads their EDGEHIP_* switches (timing experiments; the tests of those paths need such a build). */
int edgehip_experiments(void);
/* Block the caller until everything enqueued so far has finished. */
int edgehip_sync(edgehip_ctx *ctx);
/* The hipStream_t the tracker / mapper kernels (and, for whole batches, everything) are launched on, as an opaque pointer (for event
 * timing by the caller).  For batches of fewer sequences than the device has CUs the detection of a frame (stage A) runs on a second
 * stream of the context, beside the tracking and mapping of the frame before, unless EDGEHIP_OVERLAP=0 is set when the context is
 * created; edgehip_sync and the read-back calls wait for both. */
void *edgehip_stream(edgehip_ctx *ctx);
/* Device box-filter widths chosen for (sigma0, ksigma), as iigauss::iigauss does
 * (src/mtracklib/iigauss.cpp:43-81): out[0..2] filter0, out[3..5] filter1. */
int edgehip_box_widths(edgehip_ctx *ctx, int out[6]);

/* ---- frame input ----------------------------------------------------------------------------------- */
/* Copy RGB24 frames (host memory, [count][h][w][3]) into slot `slot` of sequences [seq_first,
 * seq_first+count): the `(*pbuf.imgc) = data` of rebvo_first_t.cpp:250.  Asynchronous through a pinned
 * staging buffer; the source may be reused when the call returns. */
int edgehip_upload_rgb(edgehip_ctx *ctx, int slot, const uint8_t *rgb24, int seq_first, int count);
/* Same, from device memory ([nseq][h][w][3], all sequences), device-to-device on the context stream. */
int edgehip_upload_rgb_device(edgehip_ctx *ctx, int slot, const void *rgb24_dev);
/* Host frames without the staging copy: the source is page-locked memory from edgehip_alloc_pinned (count frames for
 * sequences seq_first .. seq_first+count-1) and is read by an asynchronous copy — leave it untouched until the next
 * call that synchronises (edgehip_sync, edgehip_read_nav, ...) or write the next frames into a second buffer.
 * The copy is enqueued on a dedicated upload stream: it runs under stage A and stages B/C of the frames before, waits
 * by itself for the frame that last used the slot, and stage A of the slot waits for it. */
int edgehip_alloc_pinned(size_t bytes, void **out);
int edgehip_free_pinned(void *p);
int edgehip_upload_rgb_pinned(edgehip_ctx *ctx, int slot, const uint8_t *rgb24_pinned, int seq_first, int count);
/* Block the caller until the page-locked sources of every *_pinned upload enqueued so far have been read (the copies on the upload
 * stream are done) — the point at which a camera ring may hand the buffers back to the application
 * (cam_pipe.ReleaseBuffer, src/VideoLib/customcam.cpp:70-76).  The frames' processing is NOT waited for. */
int edgehip_upload_sync(edgehip_ctx *ctx);
/* The same for the *_pinned uploads into ONE slot: copies into other slots enqueued behind them keep running.  A camera ring whose
 * frame k+1 is already going up behind frame k hands frame k's buffers back with this (rebvo_amd/host/src/batch_group.cpp). */
int edgehip_upload_wait(edgehip_ctx *ctx, int slot);

/* Bench/replay helper: frame pool resident in HBM ([pool_frames][h][w][3]); sequence s takes frame
 * idx[s] (host array, nseq entries).  One gather kernel on the context stream. */
int edgehip_upload_rgb_indexed(edgehip_ctx *ctx, int slot, const void *pool_dev, int pool_frames, const int32_t *idx);
/* The same selection without the copy: stage A of `slot` reads sequence s's frame directly at
 * pool_dev + idx[s] * w*h*3 (what ConvertRGB2BW does with the camera buffer, rebvo_first_t.cpp:259).  The pool must
 * stay valid and unchanged until that stage A has run, and must extend at least 16 bytes past its last frame (pixels
 * are fetched as aligned 8-byte words).  Any edgehip_upload_rgb* call on the slot returns it to its own storage. */
int edgehip_bind_rgb_indexed(edgehip_ctx *ctx, int slot, const void *pool_dev, int pool_frames, const int32_t *idx);
/* 8-bit mono frames, 1 byte per pixel: what a mono camera or the EuRoC data set delivers.  The reference's DataSetCam expands
 * such an image to RGB24 with r = g = b (src/VideoLib/datasetcam.cpp:109-171) because Image<float>::ConvertRGB2BW
 * (include/VideoLib/image.h:197-203) wants RGB24; b + g + r is then 3 v, and that is what the first load of stage A computes
 * from the 8-bit frame directly — bit-identical results at a third of the bytes over PCIe and out of HBM.  Same three forms
 * as the RGB24 uploads: pageable host memory (staged), page-locked host memory (asynchronous, on the upload stream), frames of
 * a device-resident pool read in place.  A slot holds whichever format was uploaded or bound to it last. */
int edgehip_upload_grey8(edgehip_ctx *ctx, int slot, const uint8_t *grey8, int seq_first, int count);
int edgehip_upload_grey8_pinned(edgehip_ctx *ctx, int slot, const uint8_t *grey8_pinned, int seq_first, int count);
int edgehip_bind_grey8_indexed(edgehip_ctx *ctx, int slot, const void *pool_dev, int pool_frames, const int32_t *idx);

/* ---- stage A: scale space + KeyLine extraction ----------------------------------------------------- */
/* Image<float>::ConvertRGB2BW + sspace::build + edge_finder::detect + reEstimateThresh for every
 * sequence's slot `slot` (rebvo_first_t.cpp:259-272; sspace.cpp:52-85; edge_finder.cpp:67-405).
 * Reads and updates seq_state.tresh / l_kl_num exactly as detect()'s UpdateThresh does. */
int edgehip_stage_a(edgehip_ctx *ctx, int slot);
/* Number of KeyLines per sequence in `slot` (edge_finder::KNum).  Synchronises.  kn_out[nseq]. */
int edgehip_get_kn(edgehip_ctx *ctx, int slot, int32_t *kn_out);

/* ---- stage B: tracker ------------------------------------------------------------------------------ */
/* edge_tracker::EstimateQuantile(RHO_MIN,RHO_MAX,pct,nbins) on slot (rebvo_second_t.cpp:172;
 * edge_tracker.cpp:1148-1186).  Result in seq_state.s_rho_q. */
int edgehip_quantile(edgehip_ctx *ctx, int slot, double s_rho_min, double s_rho_max, double pct, int nbins);
/* global_tracker::build_field (rebvo_second_t.cpp:177; global_tracker.cpp:61-105).  min_mod < 0 takes
 * each sequence's retuned threshold of that slot (what the reference passes: new_buf.ef->getThresh()). */
int edgehip_build_field(edgehip_ctx *ctx, int slot, int radius, float min_mod);
/* One global_tracker::TryVelRot<double,ReWeight,ProcJF,false> evaluation (global_tracker.cpp:289-543) of
 * the old slot's KeyLines against the new slot's field, at state X[nseq][6].  Residual buffers are
 * device-resident and named by index 0..2 (Res0/Res1/Rest of Minimizer_RV, :611-612); resid_in < 0 means
 * all-zero.  out[nseq][43] = JtJ(36, row-major, sign-fixed and symmetrised) | JtF(6) | score.
 * Parity: every per-KeyLine value (projection, match, residual, Huber weight k/|r|, q_rho, the seven quotients by q_rho) is formed
 * by the reference's own sequence of IEEE operations (:363-463; ne10wrapper.h:414-424), so the residual memory is bit-identical;
 * the 28 sums are added in another (fixed) pair-wise tree than ne10wrapper.h:333-362 and agree to ~4e-16 relative.
 * Synchronises. */
int edgehip_try_velrot(edgehip_ctx *ctx, int slot_new, int slot_old, const double *X, int reweight, int procjf,
                       double match_thresh, const double *s_rho_min, uint32_t match_num_thresh, double k_huber,
                       int resid_in, int resid_out, double *out);
/* Copy a residual buffer to the host (resid[nseq][cap] with cap = max_points).  Synchronises. */
int edgehip_download_resid(edgehip_ctx *ctx, int which, double *resid);
/* global_tracker::Minimizer_RV<double,false> (rebvo_second_t.cpp:346; global_tracker.cpp:580-819): the
 * whole Levenberg-Marquardt loop runs on the device (evaluate kernels + a one-wave solve kernel between
 * them, no host round trip).  Reads seq_state.V/W/s_rho_q, writes V, W, P_V, P_W, score, rel_error*. */
int edgehip_minimizer_rv(edgehip_ctx *ctx, int slot_new, int slot_old);
/* Which instantiation of the tracker edgehip_minimizer_rv / edgehip_process_frame run: 64 (default) = global_tracker::Minimizer_RV<double>,
 * what the reference runs on x86 (rebvo_second_t.cpp:346); 32 = Minimizer_RV<float> with TryVelRot<float, ...> (global_tracker.cpp:824), what
 * the reference runs when it is built with USE_NE10 (rebvo_second_t.cpp:339-343: NEON only does float).  Everything the reference declares
 * as T (P0, the transformed points, residuals, gradients, Jacobian rows, the 28 sums, JtJ / JtF / h / X) is then computed and rounded in
 * float; the uncertainty gate, q_rho, the 6x6 solves and the Levenberg-Marquardt scalars stay double, as there.  Results follow the
 * reference's float instantiation to float accuracy (the 28 float sums are added in another tree than PairWiseVAdd<float>):
 * tests/test_tracker_f32_gpu.py states the tolerance.  The depth filter, the matcher and the detector are not affected.  ImuMode 0 only. */
int edgehip_set_tracker_precision(edgehip_ctx *ctx, int bits);
/* The 6x6 solve between two evaluations, n independent systems (A [n][36] row-major, b [n][6], h [n][6], host pointers):
 * svd_rule = 0: h = TooN::Cholesky<6>(A).backsub(b) (global_tracker.cpp:767-768); svd_rule = 1: h = TooN::SVD<>(A).backsub(b) with
 * its condition_no = 1e9 cut-off (global_tracker.cpp:660-661, 711-712; TooN/SVD.h:37, 179).  Exposed so that the parity tests can
 * feed the device ill-conditioned systems on either side of the cut-off; edgehip_minimizer_rv runs the same device function.
 * Synchronises. */
int edgehip_lm_solve(edgehip_ctx *ctx, const double *A, const double *b, int n, int svd_rule, double *h);

/* kfvo::Minimizer_RV_KF<double,false> with kfvo::TryVelRot<double,true,true,false> and global_tracker::Calc_f_J_Complete
 * (src/mtracklib/kfvo.cpp:1679-1825, 1389-1668; global_tracker.cpp:116-165) — the key-frame tracker of SURVEY section 8 row
 * f4: the KeyLines of slot_cur (the current frame, `klist`) against the field of slot_kf's KeyLines (the key frame's
 * global_tracker), relative pose X = [translation, rotation] starting from X0, scale ratio Kr.  Reached in the reference
 * through kfvo::OptimizePosGT (kfvo.cpp:58-113), which has no caller upstream; built to the letter of the scope row and
 * checked against the reference's own function.  The field of slot_kf is (re)built inside the call; KeyLine m_id_f of
 * slot_cur is left as the last evaluation wrote it, mnum counts its non-negative entries (kfvo.cpp:76-82).  req/res are
 * host arrays of nseq entries.  Synchronises. */
typedef struct edgehip_kf_request {
    double X0[6];        /* BRelPos, BRelRotW (kfvo.cpp:63-69) */
    double Kr;           /* K / mkf.K */
    double max_s_rho;    /* s_rho_q */
} edgehip_kf_request;
typedef struct edgehip_kf_result {
    double X[6];
    double RRV[36];      /* Cholesky<6>(JtJ).get_inverse() */
    double score_ratio;  /* F / F0, the function's return value */
    double F, F0;
    int32_t evals, mnum;
} edgehip_kf_result;
int edgehip_minimizer_rv_kf(edgehip_ctx *ctx, int slot_kf, int slot_cur, const edgehip_kf_request *req, double match_mod,
                            double match_ang, double rho_tol, int iter_max, double reweight_distance, uint32_t match_num_thresh,
                            edgehip_kf_result *res);

/* global_tracker::Minimizer_V<double> (IMU branch, rebvo_second_t.cpp:223; global_tracker.cpp:1037-1093 with
 * TryVel :830-934 and Calc_f_J :178-219): translation-only Levenberg-Marquardt of the old slot's KeyLines (already
 * rotated by the gyro estimate) against the new slot's field.  V[nseq][3] in/out, s_rho_min[nseq], min_mod < 0
 * takes each sequence's retuned threshold of the OLD slot (old_buf.ef->getThresh()); RVel[nseq][9] and F[nseq]
 * may be NULL.  FrameCount is read, not incremented (as in the reference).  Synchronises. */
int edgehip_minimizer_v(edgehip_ctx *ctx, int slot_new, int slot_old, double *V, const double *s_rho_min, float min_mod,
                        double match_thresh, int iter_max, uint32_t match_num_thresh, double reweight_distance,
                        double *RVel, double *F);

/* ---- stage C: matching + mapping ------------------------------------------------------------------- */
/* edge_tracker::FordwardMatch (rebvo_second_t.cpp:354; edge_tracker.cpp:380-436). */
int edgehip_forward_match(edgehip_ctx *ctx, int slot_old, int slot_new);
/* edge_tracker::rotate_keylines (rebvo_second_t.cpp:369; edge_tracker.cpp:42-76).  R == NULL rotates
 * each sequence by exp(seq_state.W) and stores the back-rotation in seq_state.R (:360-361). */
int edgehip_rotate_keylines(edgehip_ctx *ctx, int slot, const double *R /* [nseq][9] or NULL */);
/* edge_tracker::directed_matching (rebvo_second_t.cpp:410; edge_tracker.cpp:302-374, 158-295) with
 * V, P_V, R taken from seq_state; the match count lands in seq_state.klm_num / kf_matchs. */
int edgehip_directed_matching(edgehip_ctx *ctx, int slot_new, int slot_old);
/* Regularize_1_iter + UpdateInverseDepthKalman fused (rebvo_second_t.cpp:453, 460;
 * edge_tracker.cpp:87-148, 695-724, 954-1055).  do_regularize/do_ekf select either half (tests). */
int edgehip_regularize_ekf(edgehip_ctx *ctx, int slot, int do_regularize, int do_ekf);
/* edge_tracker::ExtRotVel (IMU branch, rebvo_second_t.cpp:237; edge_tracker.cpp:1207-1296): linear 6-DoF
 * roto-translation increment from the forward matches of `slot` (the new edge map after edgehip_forward_match),
 * given the translation estimate vel[nseq][3].  X[nseq][6]; Wx[nseq][36] = Phi^T Phi and Rx[nseq][36] = its
 * pseudo inverse may be NULL; ok[nseq] (may be NULL) is the function's return value (false on a NaN result).
 * The per-KeyLine rows and the 27 sums run on the device, the 6x6 SVD solve on the host.  Synchronises. */
int edgehip_ext_rot_vel(edgehip_ctx *ctx, int slot, const double *vel, double loc_unc, double hub_reweight, double *X,
                        double *Wx, double *Rx, int32_t *ok);
/* EstimateReScalingOpt (rebvo_second_t.cpp:487; edge_tracker.cpp:1104-1140) -> seq_state.Kp, P_Kp.
 * Parity: the five dependent weighted sums are block reductions in a fixed order (the reference adds KeyLine by KeyLine) and
 * the per-KeyLine divisions are reciprocal + Newton steps (an ulp or two each), so Kp — and every rho that DoReScaling
 * multiplies by it — follows the reference to about 1e-10 relative, not bit for bit (tests/test_stage_c_gpu.py: 1e-10). */
int edgehip_rescale(edgehip_ctx *ctx, int slot);

/* ---- stereo depth (REBVO/StereoAvaiable, experimental upstream) ------------- */
/* Intrinsics of the camera whose frames go into `slot` when they differ from the context's (the pair camera of a
 * stereo rig, cam_stereo of rebvo.cpp:202-216): stage A of that slot forms p_m with this principal point, and the
 * stereo search projects with this focal length.  zfm is taken as (float)((zfx + zfy) / 2), like cam_model. */
int edgehip_set_slot_camera(edgehip_ctx *ctx, int slot, double ppx, double ppy, double zfx, double zfy);
/* edge_tracker::directed_matching_stereo (rebvo_second_t.cpp:471; edge_tracker.cpp:580-618 with search_match_stereo
 * :453-571 and getDepthFromStereo :622-668): every KeyLine of `slot` walks, in the pair slot's edge mask, the segment
 * its depth interval [rho - s_rho, rho + s_rho] projects to through p1 = R p0 + t, keeps a match only when it is
 * unique (or all candidates lie within loc_unc of each other) and triangulates stereo_rho / stereo_s_rho from it.
 * t[3], R[9] are shared by all sequences; nmatch[nseq] (may be NULL) returns the function's result per sequence.
 * q_abs / q_rel are accepted and unused, as in the reference.  Needs params.stereo_available.  Synchronises. */
int edgehip_directed_matching_stereo(edgehip_ctx *ctx, int slot, int slot_pair, const double *t, const double *R,
                                     double min_thr_mod, double min_thr_ang, double max_radius, double loc_unc,
                                     double q_abs, double q_rel, double loc_unc_model, int32_t *nmatch);
/* edge_tracker::fuseStereoDepth (rebvo_second_t.cpp:484; edge_tracker.cpp:670-688): rho0/s_rho0 = rho/s_rho, then the
 * information-weighted mean with the stereo depth where a stereo match exists. */
int edgehip_fuse_stereo_depth(edgehip_ctx *ctx, int slot);
/* Stereo inside edgehip_process_frame (what SecondThread does with StereoAvaiable, rebvo_second_t.cpp:410, 465-486):
 * the last ring slot becomes the pair slot (the frame ring then cycles through the others); the caller uploads the pair
 * image into it before every edgehip_process_frame, which then also runs stage A on it (after the main image, sharing
 * the detector threshold state like rebvo_first_t.cpp:283-289) and, after the depth EKF, directed_matching_stereo with
 * (t, R, max_radius) and the context's matching thresholds, fuseStereoDepth, and Kp = 1 in place of
 * EstimateReScalingOpt.  Set edgehip_set_slot_camera for the pair slot first when the cameras differ.
 * slot_pair < 0 switches the rig off.  edgehip_get_stereo_matches: stereo_match_num of the last frame.  Synchronises. */
int edgehip_set_stereo_rig(edgehip_ctx *ctx, int slot_pair, const double *t, const double *R, double max_radius);
int edgehip_get_stereo_matches(edgehip_ctx *ctx, int32_t *nmatch);

/* ---- whole frame ------------------------------------------------------------------------------------ */
/* Everything FirstThr + SecondThread (ImuMode==0) do for one new frame of every sequence, on the
 * context's current ring slot, enqueued back to back without host synchronisation: stage A on the new
 * slot, then (from the second frame on) quantile, build_field, Minimizer_RV, FordwardMatch, rotate,
 * directed_matching, Regularize, EKF, rescale and the pose integration of rebvo_second_t.cpp:550-551.
 * The frame must have been uploaded into edgehip_next_slot() first.  t[nseq] = frame time stamps.
 * Uploads and stage A are enqueued on a second HIP stream; with EDGEHIP_OVERLAP=1 in the environment at
 * edgehip_create() time, stage A of this frame only waits for the B/C work that still reads the slot it overwrites
 * and so runs under the tracking/mapping of the previous frame (the reference's T0 || T1 pipelining).
 * With EDGEHIP_GRAPH=1 the launches of a frame are captured into HIP graphs (one per ring-slot / FrameCount-row
 * combination) the first time they occur and replayed afterwards. */
int edgehip_process_frame(edgehip_ctx *ctx, const double *t);
int edgehip_next_slot(edgehip_ctx *ctx);
int edgehip_cur_slot(edgehip_ctx *ctx);
/* Per-sequence record of the last processed frame.  Synchronises.  nav[nseq]. */
int edgehip_read_nav(edgehip_ctx *ctx, edgehip_nav *nav);
/* ImuMode > 0 for every sequence of the context (rebvo_second_t.cpp:54-94 set-up, :182-336 tracker + filters, :519-544
 * pose): from the next frame on edgehip_process_frame takes the IMU branch — gyro pre-rotation, Minimizer_V, FordwardMatch,
 * ExtRotVel, BiasCorrect, rotate, the mapper, the scale filter and the gravity-aligned pose — entirely on the device, no
 * host synchronisation, one thread per sequence for the 3..11-dimensional filters.  Call before the first frame. */
int edgehip_imu_enable(edgehip_ctx *ctx, const edgehip_imu_params *imu);
/* The integrated IMU data of the interval that ends with the NEXT edgehip_process_frame, one record per sequence
 * (ImuGrabber::GrabAndIntegrate stays with the caller: it is I/O).  Asynchronous; the records are copied before return. */
int edgehip_set_imu(edgehip_ctx *ctx, const edgehip_imu_integrated *per_seq);
/* The IMU part of the newest frame's record, one per sequence (synchronises like edgehip_read_nav). */
int edgehip_read_nav_imu(edgehip_ctx *ctx, edgehip_nav_imu *out);
/* Keep the last `len` per-frame records of every sequence in HBM (ring indexed by frame number) so that a
 * replay can run many frames without reading back; edgehip_read_nav_log copies records of frames
 * [first, first+count) as out[count][nseq]; the frames must have been enqueued (EDGEHIP_ERR_STATE before the first one).
 * Threading: a context is driven by ONE thread at a time — with this exception: edgehip_read_nav_log may be called from a
 * second thread while the first keeps enqueueing frames (the nav gather of a multi-GPU replay does).  It waits, on a stream
 * of its own, for the newest frame it is asked for — not for frames enqueued behind it, so a caller may keep frames in flight
 * and read the records one or two frames late — and never touches the context's frame streams (which may be capturing a
 * frame graph).  edgehip_set_nav_log itself belongs to the driving thread, with no reader active. */
int edgehip_set_nav_log(edgehip_ctx *ctx, int len);
int edgehip_read_nav_log(edgehip_ctx *ctx, int first, int count, edgehip_nav *out);
/* The same records into DEVICE memory of the context's GPU (out_dev[count][nseq] edgehip_nav, e.g. a tensor the caller's RCCL
 * communicator sends from): the multi-GPU nav gather of SURVEY.md section 8(e) takes its payload from HBM to the wire without a
 * host bounce (rebvo_amd/shard.py NavMover).  Same waiting, threading and range rules; the copy is complete on return. */
int edgehip_read_nav_log_device(edgehip_ctx *ctx, int first, int count, void *out_dev);
/* ImuMode > 0: the IMU part of the logged records (what edgehip_read_nav_imu returns for the newest frame), frames [first, first+count)
 * as out[count][nseq] — same ring, same waiting and threading rules as edgehip_read_nav_log.  A caller that keeps frames in flight
 * (a batch group of ImuMode = 1 / 2 objects, rebvo_amd/host/src/batch_group.cpp) reads both halves of frame k while k+1 and k+2 run. */
int edgehip_read_nav_imu_log(edgehip_ctx *ctx, int first, int count, edgehip_nav_imu *out);
/* With a stereo rig (edgehip_set_stereo_rig): stereo_match_num (rebvo_second_t.cpp:471-477, what edgehip_get_stereo_matches returns for the
 * newest frame) of the logged frames [first, first+count) as out[count][nseq] — same ring, same waiting and threading rules as
 * edgehip_read_nav_log; 0 for a sequence's first frame.  A batch group of StereoAvaiable objects reads it with frames in flight. */
int edgehip_read_stereo_matches_log(edgehip_ctx *ctx, int first, int count, int32_t *out);
/* Restart every sequence from scratch: state as after edgehip_create (thresholds, priors, pose, frame
 * counters) and an empty ring.  Not something the reference does at run time (it would re-construct REBVO). */
int edgehip_reset(edgehip_ctx *ctx);
/* REBVO::Reset() as SecondThread executes it after a frame (rebvo_second_t.cpp:609-620): depth reset of the
 * newest edge map (rho = RhoInit, s_rho = RHO_MAX for every KeyLine), Pose = I, Pos = V = W = 0.  Everything
 * else (detector threshold, K, frame counters) carries on.  seq < 0 applies it to all sequences. */
int edgehip_depth_reset(edgehip_ctx *ctx, int seq);
/* The same on an explicit ring slot, for callers that drive the stages one by one (the host's IMU branch) instead of
 * through edgehip_process_frame. */
int edgehip_depth_reset_slot(edgehip_ctx *ctx, int seq, int slot);

/* ---- state / data exchange (callback consumers, parity tests) ---------------------------------------- */
int edgehip_get_state(edgehip_ctx *ctx, int seq, edgehip_seq_state *out);       /* synchronises */
int edgehip_set_state(edgehip_ctx *ctx, int seq, const edgehip_seq_state *in);
int edgehip_get_framecount(edgehip_ctx *ctx, int seq, int slot, uint32_t *fc);  /* global_tracker::FrameCount */
int edgehip_set_framecount(edgehip_ctx *ctx, int seq, int slot, uint32_t fc);
/* AoS KeyLine list + img_mask_kl of one sequence/slot, as the output callback sees them
 * (PipeBuffer::ef, include/rebvo/rebvo.h:312-351).  kl has room for max_points entries; mask (h*w int32)
 * may be NULL.  Synchronises.  Returns kn through *kn_out.
 * The OLD slot of a frame edgehip_process_frame() has run holds what old_buf.ef holds after rotate_keylines (rebvo_second_t.cpp:369):
 * the frame driver keeps the turned p_m / m_m / rho / s_rho next to the slot's arrays for its own matching kernel and brings them in
 * when anybody else — this call, any stage-level entry point — asks for the slot. */
int edgehip_download_keylines(edgehip_ctx *ctx, int seq, int slot, edgehip_keyline *kl, int32_t *mask,
                              int32_t *kn_out);
/* The same lists for several sequences of one slot in ONE packing kernel and one copy per list (the output callbacks of a batch of cameras,
 * rebvo_amd/host/src/batch_group.cpp): seqs[n] sequence indices, kl[n] destinations of max_points entries each, kn_out[n].  Record for record what
 * edgehip_download_keylines returns (no mask).  Synchronises. */
int edgehip_download_keylines_batch(edgehip_ctx *ctx, int slot, int n, const int32_t *seqs, edgehip_keyline *const *kl, int32_t *kn_out);
/* Output callbacks at full pipeline depth (round 6; the consumer: setOutputCallback, include/rebvo/rebvo.h:595-609, called by
 * REBVO::ThirdThread, src/rebvo/rebvo_third_t.cpp:174 — every real user of the surface has one, ros/src/rebvo_ros/src/rebvo_nodelet.cpp:146-242).
 * A callback gets frame k-1's edge map as frame k's tracking left it: the OLD slot of the frame processed last.
 *   edgehip_export_keylines  right after edgehip_process_frame(k): packs that slot's KeyLines of sequences seqs[n] as AoS records into a
 *                            device-side staging ring, in-stream behind the frame (no synchronisation; the ring slot itself is free for
 *                            the frame after next).  Returns a ticket; at most four may be outstanding.
 *   edgehip_export_fetch     once the caller knows the lists' lengths (kn[j] = edgehip_nav::kn of frame k-1 for seqs[j]): enqueues the copies
 *                            of exactly kn[j] records into dst[j] on a stream of their own (page-locked destinations — edgehip_register_host —
 *                            are written by DMA under the frames that follow; pageable ones work, slower).  Does not block.
 *   edgehip_export_wait      blocks until the ticket's copies have landed and releases the ticket (never fetched: just releases it).
 * Record for record what edgehip_download_keylines_batch returns for the same slot at the same point. */
int edgehip_export_keylines(edgehip_ctx *ctx, int n, const int32_t *seqs, int *ticket_out);
int edgehip_export_fetch(edgehip_ctx *ctx, int ticket, const int32_t *kn, edgehip_keyline *const *dst);
int edgehip_export_wait(edgehip_ctx *ctx, int ticket);
/* Page-lock host memory the caller owns, in place (hipHostRegister), so that edgehip_download_keylines_batch copies straight into
 * it: a destination inside a registered range skips the library's staging buffer and the host memcpy behind it (2.4 MB per list of
 * 14 k KeyLines).  What a batch group does with the KeyLine arrays of the members that have an output callback
 * (PipeBuffer::ef of setOutputCallback, include/rebvo/rebvo.h:595-609).  Unregister before the memory is freed. */
int edgehip_register_host(void *p, size_t bytes);
int edgehip_unregister_host(void *p);
/* Inject a KeyLine list (+ mask) into a slot: stage-isolated parity tests. */
int edgehip_upload_keylines(edgehip_ctx *ctx, int seq, int slot, const edgehip_keyline *kl, int32_t kn,
                            const int32_t *mask, float retuned_thresh);
/* Planes of the scale space (debug_planes must be set): which = 0 img0, 1 img1, 2 dog, 3 dx, 4 dy.
 * out[h*w] float.  Synchronises. */
int edgehip_download_plane(edgehip_ctx *ctx, int seq, int which, float *out);
/* Auxiliary field of the tracker in the reference's {dist, ikl} form (global_tracker.h:33-36); out[h*w*2].
 * The tracker only ever reads ikl, so the device keeps a 2-byte KeyLine-index plane; dist is stored as well (and
 * returned here) when params.debug_planes is set, otherwise it is reported as -1 (0 where the pixel is empty). */
int edgehip_download_field(edgehip_ctx *ctx, int seq, int32_t *out);
/* The undistortion map edgehip_create() builds for `params` (host-only, needs no device), in the reference's
 * undistMapPoint form (image_undistort.h:41-47): inx[h*w*4] valid taps first, -1 beyond `num`; iw[h*w*4]. */
int edgehip_build_undistort_map(const edgehip_params *params, int32_t *inx, int32_t *iw);
/* Debug: the undistorted RGB24 frame that stage A consumed for sequence `seq` in `slot`, recomputed on the
 * device with the same integer arithmetic (what PipeBuffer::imgc holds after rebvo_first_t.cpp:231). */
int edgehip_download_undistorted(edgehip_ctx *ctx, int seq, int slot, uint8_t *rgb24);

/* ---- measurement ------------------------------------------------------------------------------------- */
/* Names of the kernel groups timed by the built-in HIP-event profiler, and their accumulated device time.
 * edgehip_profile_enable(ctx, 1) brackets every launch group with events on the context stream (adds host
 * overhead: use for attribution, not for throughput).  ms[n], calls[n] with n = edgehip_profile_count(). */
int edgehip_profile_enable(edgehip_ctx *ctx, int on);
/* Restrict the profiler to the groups whose bit is set (bit i = group i); default: all. */
int edgehip_profile_select(edgehip_ctx *ctx, uint64_t mask);
int edgehip_profile_count(void);
const char *edgehip_profile_name(int i);
int edgehip_profile_read(edgehip_ctx *ctx, double *ms, int64_t *calls); /* synchronises, then resets */

#ifdef __cplusplus
}
#endif
#endif /* EDGEHIP_H */
