// ref_harness.cpp — C-ABI driver over the REFERENCE's own mtracklib classes.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; every algorithmic line it executes lives in
// /root/reference (compiled in place by oracle/Makefile, never copied).  It exists so that the parity
// tests can ask "what does the reference compute for this input?" stage by stage, and so that bench.py
// can time the reference CPU path (`cpu_baseline.kind = "reference"`).
//
// The per-frame sequencing in ref_process_frame() restates, single-threaded, what the reference does in
//   REBVO::FirstThr      src/rebvo/rebvo_first_t.cpp:259-272   (stage A)
//   REBVO::SecondThread  src/rebvo/rebvo_second_t.cpp:128-629  (stage B/C, ImuMode==0 branch)
// including the 8-slot PipeBuffer ring (one sspace/edge_tracker/global_tracker per slot,
// src/rebvo/rebvo.cpp:297-312) because global_tracker::FrameCount is per slot.

#include <cstring>
#include <cmath>
#include <chrono>
#include <vector>
#include <atomic>
#include <thread>
#include <algorithm>

#include "mtracklib/sspace.h"
#include "mtracklib/edge_finder.h"
#include "mtracklib/edge_tracker.h"
#include "mtracklib/global_tracker.h"
#include "VideoLib/image_undistort.h"
#include "mtracklib/scaleestimator.h"
#include "UtilLib/imugrabber.h"
#include "CommLib/net_keypoint.h"
// TryVelRot<> is only defined in the .cpp; include it so the harness can instantiate it directly.
#include "src/mtracklib/global_tracker.cpp"
// kfvo::Minimizer_RV_KF<> / kfvo::TryVelRot<> likewise live in kfvo.cpp.  Its header chain (kfvo.h -> keyframe.h ->
// visualizer/depth_filler.h) does not compile with this image's g++ 11 (depth_filler.h:141: std::max(float, double));
// the key-frame tracker uses nothing of depth_filler — class keyframe only holds a shared_ptr to it and names its
// bound_modes enum in one declaration — so the header's include guard is taken and a stand-in with that enum declared.
#define DEPTH_FILLER_H
namespace rebvo { class depth_filler { public: enum bound_modes { BOUND_NONE }; }; }
#include "src/mtracklib/kfvo.cpp"

#include <TooN/so3.h>
#include "oracle_abi.h"

using namespace rebvo;
using namespace TooN;

static_assert(sizeof(OrcKeyLine) == sizeof(KeyLine), "OrcKeyLine must mirror rebvo::KeyLine");
static_assert(offsetof(OrcKeyLine, rho) == offsetof(KeyLine, rho), "layout");
static_assert(offsetof(OrcKeyLine, p_m) == offsetof(KeyLine, p_m), "layout");
static_assert(offsetof(OrcKeyLine, m_id) == offsetof(KeyLine, m_id), "layout");
static_assert(offsetof(OrcKeyLine, n_m0) == offsetof(KeyLine, n_m0), "layout");
static_assert(offsetof(OrcKeyLine, p_id) == offsetof(KeyLine, p_id), "layout");
static_assert(offsetof(OrcKeyLine, stereo_rho) == offsetof(KeyLine, stereo_rho), "layout");
static_assert(sizeof(gt_field_data) == 8, "field layout");

// ---- dgesvd_ as TooN::SVD<> calls it (TooN/SVD.h:121-163) ---------------------------------------------------------------
// The harness owns the symbol: it records every decomposition the reference asks for (ref_svd_trace: the parity work needs
// to see the 6x6 systems of Minimizer_RV's init phase and their singular values) and forwards to LAPACK (MKL's `dgesvd`), or
// — REF_HARNESS_DGESVD_SHIM / ref_svd_backend(1) — to a one-sided Jacobi decomposition of its own.  Two back ends exist
// because LAPACK is an un-vendored dependency of the reference with no version pinned: what the reference computes at a
// rounding-level knife edge depends on it (MKL even picks its code path by CPU model).
extern "C" void dgesvd_(const char *jobu, const char *jobvt, int *m, int *n, double *a, int *lda, double *s,
                        double *u, int *ldu, double *vt, int *ldvt, double *work, int *lwork, int *info);
#ifndef REF_HARNESS_DGESVD_SHIM
extern "C" void dgesvd(const char *jobu, const char *jobvt, int *m, int *n, double *a, int *lda, double *s,
                       double *u, int *ldu, double *vt, int *ldvt, double *work, int *lwork, int *info);
static int g_svd_backend = 0;
#else
static int g_svd_backend = 1;
#endif
static OrcSvdRec *g_svd_buf = nullptr;
static int g_svd_cap = 0, g_svd_n = 0;

// One-sided (Hestenes) Jacobi for the call shape TooN uses on a square or vertical row-major matrix M (LAPACK sees M^T,
// jobu = 'S', jobvt = 'O'): on return `a` holds U (row-major), `u` holds V^T (row-major), s descending.
static void jacobi_gesvd(int cols, int rows, double *a, int lda, double *s, double *vt_out, int ldv) {
    std::vector<double> W(rows * cols), V(cols * cols, 0.0);
    for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) W[r * cols + c] = a[r * lda + c];
    for (int c = 0; c < cols; c++) V[c * cols + c] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < cols - 1; p++)
            for (int q = p + 1; q < cols; q++) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < rows; r++) { al += W[r * cols + p] * W[r * cols + p]; be += W[r * cols + q] * W[r * cols + q]; ga += W[r * cols + p] * W[r * cols + q]; }
                if (ga == 0 || std::fabs(ga) <= 1e-17 * std::sqrt(al * be)) continue;
                rotated = true;
                const double zeta = (be - al) / (2 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double c = 1 / std::sqrt(1 + t * t), sn = c * t;
                for (int r = 0; r < rows; r++) { const double x = W[r * cols + p], y = W[r * cols + q]; W[r * cols + p] = c * x - sn * y; W[r * cols + q] = sn * x + c * y; }
                for (int r = 0; r < cols; r++) { const double x = V[r * cols + p], y = V[r * cols + q]; V[r * cols + p] = c * x - sn * y; V[r * cols + q] = sn * x + c * y; }
            }
        if (!rotated) break;
    }
    std::vector<double> nrm(cols);
    std::vector<int> ord(cols);
    for (int c = 0; c < cols; c++) { double d = 0; for (int r = 0; r < rows; r++) d += W[r * cols + c] * W[r * cols + c]; nrm[c] = std::sqrt(d); ord[c] = c; }
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return nrm[x] > nrm[y]; });
    for (int j = 0; j < cols; j++) {
        const int c = ord[j];
        s[j] = nrm[c];
        for (int r = 0; r < rows; r++) a[r * lda + j] = nrm[c] > 0 ? W[r * cols + c] / nrm[c] : 0.0;
        for (int r = 0; r < cols; r++) vt_out[j * ldv + r] = V[r * cols + c];
    }
}

extern "C" void dgesvd_(const char *jobu, const char *jobvt, int *m, int *n, double *a, int *lda, double *s,
                        double *u, int *ldu, double *vt, int *ldvt, double *work, int *lwork, int *info) {
    const bool query = *lwork == -1;
    // the shape every SVD<> on this path has: rows >= cols (TooN's "vertical" case), LAPACK-side jobu 'S', jobvt 'O'
    const bool tall = *jobu == 'S' && *jobvt == 'O' && *n >= *m;
    OrcSvdRec *rec = nullptr;
    if (!query && g_svd_buf && g_svd_n < g_svd_cap && *m <= 6 && *n <= 6) {
        rec = &g_svd_buf[g_svd_n++];
        std::memset(rec, 0, sizeof *rec);
        rec->rows = *n; rec->cols = *m;
        for (int r = 0; r < *n; r++) for (int c = 0; c < *m; c++) rec->A[r * 6 + c] = a[r * *lda + c];
    }
    if (g_svd_backend == 1 && tall) {
        *info = 0;
        if (query) work[0] = 1; else jacobi_gesvd(*m, *n, a, *lda, s, u, *ldu);
    } else {
#ifndef REF_HARNESS_DGESVD_SHIM
        dgesvd(jobu, jobvt, m, n, a, lda, s, u, ldu, vt, ldvt, work, lwork, info);
#else
        *info = -1;   // a shape the built-in decomposition does not cover
#endif
    }
    if (rec) for (int i = 0; i < (*m < *n ? *m : *n); i++) rec->s[i] = s[i];
}
extern "C" void ref_svd_trace(OrcSvdRec *buf, int cap) { g_svd_buf = buf; g_svd_cap = cap; g_svd_n = 0; }
extern "C" int ref_svd_trace_count(void) { return g_svd_n; }
extern "C" int ref_svd_backend(int which) { const int old = g_svd_backend;
#ifndef REF_HARNESS_DGESVD_SHIM
    if (which == 0 || which == 1) g_svd_backend = which;
#endif
    return old; }

namespace {

// The members of rebvo::IMUState (include/rebvo/rebvo.h:239-290) the IMU branch uses.  Declared here because that
// header pulls in the visualizer / video / network stack, which does not build in this image.
struct IMUState {
    Vector<3> Vg = Zeros, dVv = Zeros, dWv = Zeros, dVgv = Zeros, dWgv = Zeros, Vgv = Zeros, Wgv = Zeros;
    Vector<3> dVgva = Zeros, dWgva = Zeros, Vgva = Zeros;
    Matrix<3, 3> P_Vg = Identity * 1e50, RGiro = Identity, RGBias = Identity;
    Vector<3> Bg = Zeros;
    Matrix<3, 3> W_Bg = Identity;
    Vector<3> Av = Zeros, As = Zeros;
    Vector<7> X;
    Matrix<7, 7> P;
    Matrix<3, 3> Qrot, Qg, Qbias;
    double QKp, Rg;
    Matrix<3, 3> Rs, Rv;
    Vector<3> g_est, u_est, b_est;
    Matrix<6, 6> Wvw;
    Vector<6> Xvw;
    Vector<3> Posgv = Zeros, Posgva = Zeros;
    bool init = false;
};

struct Slot {
    sspace *ss;
    edge_tracker *ef;
    global_tracker *gt;
    Image<float> *img;
    Image<RGB24Pixel> *imgc;
};

struct Ctx {
    OrcParams p;
    cam_model cam;
    std::vector<Slot> slots;
    // FirstThr state
    double tresh;
    int l_kl_num;
    // SecondThread state
    int frame;
    double t_prev;
    double Kp, K, P_Kp;
    Vector<3> V, W, Pos;
    Matrix<3, 3> Pose;
    std::vector<float> bw;
    image_undistort *undist;          // rebvo_first_t.cpp:125 (only when use_undistort)
    Image<RGB24Pixel> *img_dist;      // the distorted input frame (rebvo_first_t.cpp:112)
    // IMU branch of SecondThread (rebvo_second_t.cpp:54-94)
    OrcImuParams ip;
    IMUState istate;
    int n_frame, n_giro_init;
    Vector<3> giro_init, g_init;
    Matrix<3, 3> Rgva;
    bool stereo_mode;                 // REBVO/StereoAvaiable as directed_matching sees it
    bool tracker_f32;                 // Minimizer_RV<float> in place of Minimizer_RV<double>: what USE_NE10 selects (rebvo_second_t.cpp:339-343)
    int ring;                         // slots of the frame ring (a stereo pair slot, if any, sits behind them)
    bool rig;                         // whole-frame stereo (ref_enable_stereo)
    Vector<3> rig_t;
    Matrix<3, 3> rig_R;
    double rig_radius;
    const uint8_t *pair_rgb;          // pair image of the frame being processed
};

void reset_seq(Ctx *c) {
    c->tresh = c->p.detector_thresh;
    c->l_kl_num = 0;
    c->frame = 0;
    c->t_prev = 0;
    c->Kp = 1;
    c->K = 1;
    c->P_Kp = 5e-6;
    c->V = Zeros;
    c->W = Zeros;
    c->Pos = Zeros;
    c->Pose = Identity;
    for (Slot &s : c->slots) s.gt->FrameCount = 0;
}

inline Matrix<3, 3> m3(const double *r) {
    Matrix<3, 3> M;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M(i, j) = r[i * 3 + j];
    return M;
}
inline void put3(double *r, const Matrix<3, 3> &M) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 3 + j] = M(i, j);
}
inline Vector<3> v3(const double *v) { return makeVector(v[0], v[1], v[2]); }
inline void putv(double *r, const Vector<3> &v) {
    for (int i = 0; i < 3; i++) r[i] = v[i];
}
inline double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void *ref_create(const OrcParams *p, int nslots) {
    Ctx *c = new Ctx;
    c->p = *p;
    cam_model::rad_tan_distortion kc = {p->kc[0], p->kc[1], p->kc[2], p->kc[3], p->kc[4]};
    // REBVOParameters keeps pp/zf as float (rebvo.h:108-111); cam_model as in rebvo.cpp:231
    Size2D sz = {(u_int)p->w, (u_int)p->h};
    c->cam = cam_model({(float)p->ppx, (float)p->ppy}, {(float)p->zfx, (float)p->zfy}, kc, sz);
    c->slots.resize(nslots);
    for (Slot &s : c->slots) {  // rebvo.cpp:297-312
        s.ss = new sspace(p->sigma0, p->ksigma, c->cam.sz, 3);
        s.ef = new edge_tracker(c->cam, 255 * 3);
        s.gt = new global_tracker(s.ef->GetCam());
        s.img = new Image<float>(c->cam.sz);
        s.imgc = new Image<RGB24Pixel>(c->cam.sz);
        // the reference leaves these heap planes uninitialised; zero them so dumps are deterministic
        memset(s.ss->ImgDx().Data(), 0, sizeof(float) * p->w * p->h);
        memset(s.ss->ImgDy().Data(), 0, sizeof(float) * p->w * p->h);
        memset(s.gt->field.Data(), 0, sizeof(gt_field_data) * p->w * p->h);
    }
    c->undist = nullptr;
    c->img_dist = nullptr;
    c->stereo_mode = false;
    c->tracker_f32 = false;
    c->ring = nslots;
    c->rig = false;
    c->pair_rgb = nullptr;
    if (p->use_undistort) {
        c->undist = new image_undistort(c->cam);
        c->img_dist = new Image<RGB24Pixel>(c->cam.sz);
    }
    reset_seq(c);
    return c;
}

void ref_destroy(void *ctx) {
    Ctx *c = (Ctx *)ctx;
    for (Slot &s : c->slots) {
        delete s.gt;
        delete s.ef;
        delete s.ss;
        delete s.img;
        delete s.imgc;
    }
    delete c->undist;
    delete c->img_dist;
    delete c;
}

void ref_reset_sequence(void *ctx) { reset_seq((Ctx *)ctx); }
void ref_depth_reset(void *ctx) {   // the `if(cf->system_reset)` block of rebvo_second_t.cpp:609-620, on the newest slot
    Ctx *c = (Ctx *)ctx;
    if (c->frame == 0) return;
    Slot &nb = c->slots[(c->frame + c->ring - 1) % c->ring];
    for (auto &kl : (*nb.ef)) {
        kl.rho = RhoInit;
        kl.s_rho = RHO_MAX;
    }
    c->Pose = Identity;
    c->Pos = Zeros;
    c->W = Zeros;
    c->V = Zeros;
}
int ref_cur_slot(void *ctx) {
    Ctx *c = (Ctx *)ctx;
    return (c->frame + c->ring - 1) % c->ring;
}

int ref_stage_a(void *ctx, int slot, const uint8_t *rgb24, double *tresh_io, int *l_kl_num_io) {
    Ctx *c = (Ctx *)ctx;
    Slot &s = c->slots[slot];
    const OrcParams &p = c->p;
    if (c->undist) {
        memcpy(c->img_dist->Data(), rgb24, (size_t)p.w * p.h * 3);   // rebvo_first_t.cpp:218
        c->undist->undistort<true>(*s.imgc, *c->img_dist);           // :231
    } else {
        memcpy(s.imgc->Data(), rgb24, (size_t)p.w * p.h * 3);        // :250
    }
    Image<float>::ConvertRGB2BW(*s.img, *s.imgc);                  // :259
    s.ss->build(*s.img);                                           // :263
    s.ef->detect(s.ss, p.plane_fit_size, p.pos_neg_thresh, p.dog_thresh, p.max_points, *tresh_io,
                 *l_kl_num_io, p.reference_points, p.auto_gain, p.max_thresh, p.min_thresh);  // :266
    s.ef->reEstimateThresh(p.track_points, p.qcut_nbins);          // :272
    return s.ef->KNum();
}

const uint8_t *ref_imgc(void *ctx, int slot) {  // PipeBuffer::imgc (undistorted when use_undistort)
    return (const uint8_t *)((Ctx *)ctx)->slots[slot].imgc->Data();
}

// the undistortion map in the reference's own form: inx[n*4] (-1 beyond num), iw[n*4]
int ref_undistort_map(void *ctx, int32_t *inx, int32_t *iw) {
    Ctx *c = (Ctx *)ctx;
    if (!c->undist) return -1;
    const int n = c->p.w * c->p.h;
    for (int i = 0; i < n; i++) {
        const auto &u = c->undist->umap[i];
        for (int k = 0; k < 4; k++) {
            inx[i * 4 + k] = k < u.num ? u.inx[k] : -1;
            iw[i * 4 + k] = k < u.num ? u.iw[k] : 0;
        }
    }
    return 0;
}

const float *ref_plane(void *ctx, int slot, int which) {
    Slot &s = ((Ctx *)ctx)->slots[slot];
    switch (which) {
        case 0: return s.ss->Img(0).Data();
        case 1: return s.ss->Img(1).Data();
        case 2: return s.ss->ImgDOG().Data();
        case 3: return s.ss->ImgDx().Data();
        case 4: return s.ss->ImgDy().Data();
        default: return s.img->Data();
    }
}
const int32_t *ref_mask(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].ef->img_mask_kl.Data(); }
int ref_kn(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].ef->KNum(); }
OrcKeyLine *ref_keylines(void *ctx, int slot) { return (OrcKeyLine *)((Ctx *)ctx)->slots[slot].ef->kl; }
float ref_retuned(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].ef->reTunedThresh; }

void ref_set_keylines(void *ctx, int slot, const OrcKeyLine *kl, int kn, const int32_t *mask, float retuned) {
    Ctx *c = (Ctx *)ctx;
    edge_tracker *ef = c->slots[slot].ef;
    memcpy(ef->kl, kl, sizeof(KeyLine) * kn);
    ef->kn = kn;
    if (mask) memcpy(ef->img_mask_kl.Data(), mask, sizeof(int) * c->p.w * c->p.h);
    ef->reTunedThresh = retuned;
}
unsigned ref_get_framecount(void *ctx, int slot) { return ((Ctx *)ctx)->slots[slot].gt->FrameCount; }
void ref_set_framecount(void *ctx, int slot, unsigned fc) { ((Ctx *)ctx)->slots[slot].gt->FrameCount = fc; }

double ref_quantile(void *ctx, int slot, double smin, double smax, double pct, int n) {
    return ((Ctx *)ctx)->slots[slot].ef->EstimateQuantile(smin, smax, pct, n);
}
void ref_build_field(void *ctx, int slot, int radius, float min_mod) {
    Slot &s = ((Ctx *)ctx)->slots[slot];
    s.gt->build_field(*s.ef, radius, min_mod);
}
const int32_t *ref_field(void *ctx, int slot) { return (const int32_t *)((Ctx *)ctx)->slots[slot].gt->field.Data(); }

// One evaluation of global_tracker::TryVelRot<double,...> (global_tracker.cpp:289-543) at state X.
double ref_try_velrot(void *ctx, int slot_new, int slot_old, const double X[6], int reweight, int procjf,
                      double match_thresh, double s_rho_min, unsigned match_num_thresh, double k_huber,
                      const double *resid_in, double *resid_out, double JtJo[36], double JtFo[6]) {
    Ctx *c = (Ctx *)ctx;
    global_tracker *gt = c->slots[slot_new].gt;
    edge_tracker &klist = *c->slots[slot_old].ef;
    int kn = klist.KNum();
    int pnum = (kn + 0x3) & (~0x3);                                  // global_tracker.cpp:608
    std::vector<double> P0Im(pnum * 3), P0m(pnum * 3), rin(pnum, 0.0), rout(pnum, 0.0);
    KltoI3PMatrix<double>(klist, pnum, P0Im.data());                 // :615
    Ne10::ProyI3Pto3PMatrix<double>(P0m.data(), P0Im.data(), gt->cam_mod.zfm, pnum);  // :617
    if (resid_in) memcpy(rin.data(), resid_in, sizeof(double) * kn);
    if (resid_out) memcpy(rout.data(), resid_out, sizeof(double) * kn);
    Matrix<6, 6, double> JtJ = Zeros;
    Vector<6, double> JtF = Zeros, Xv;
    for (int i = 0; i < 6; i++) Xv[i] = X[i];
    Vector<3> V0 = Zeros, W0 = Zeros;
    Matrix<3, 3> P0 = Identity;
    double F;
    if (reweight) {
        if (procjf)
            F = gt->TryVelRot<double, true, true, false>(JtJ, JtF, Xv, V0, P0, W0, P0, klist, P0m.data(), pnum,
                                                         match_thresh, s_rho_min, match_num_thresh, k_huber,
                                                         rin.data(), rout.data());
        else
            F = gt->TryVelRot<double, true, false, false>(JtJ, JtF, Xv, V0, P0, W0, P0, klist, P0m.data(), pnum,
                                                          match_thresh, s_rho_min, match_num_thresh, k_huber,
                                                          rin.data(), rout.data());
    } else {
        if (procjf)
            F = gt->TryVelRot<double, false, true, false>(JtJ, JtF, Xv, V0, P0, W0, P0, klist, P0m.data(), pnum,
                                                          match_thresh, s_rho_min, match_num_thresh, k_huber,
                                                          rin.data(), rout.data());
        else
            F = gt->TryVelRot<double, false, false, false>(JtJ, JtF, Xv, V0, P0, W0, P0, klist, P0m.data(), pnum,
                                                           match_thresh, s_rho_min, match_num_thresh, k_huber,
                                                           rin.data(), rout.data());
    }
    if (resid_out) memcpy(resid_out, rout.data(), sizeof(double) * kn);
    for (int i = 0; i < 6; i++) {
        JtFo[i] = JtF[i];
        for (int j = 0; j < 6; j++) JtJo[i * 6 + j] = JtJ(i, j);
    }
    return F;
}

double ref_minimizer_rv(void *ctx, int slot_new, int slot_old, double V[3], double W[3], double RVel[9],
                        double RW0[9], double match_thresh, int iter_max, int init_type,
                        double reweight_distance, double *rel_error, double *rel_error_score, double max_s_rho,
                        unsigned match_num_thresh, double init_iter, double W_Xo[36]) {
    Ctx *c = (Ctx *)ctx;
    Vector<3> Vv = v3(V), Wv = v3(W);
    Matrix<3, 3> RV = m3(RVel), RW = m3(RW0);
    Matrix<6, 6, double> W_X = Zeros;
    double F;
    if (c->tracker_f32) {   // the reference's float instantiation (global_tracker.cpp:824), its own code
        Matrix<6, 6, float> W_Xf = Zeros;
        F = c->slots[slot_new].gt->Minimizer_RV<float>(Vv, Wv, RV, RW, *c->slots[slot_old].ef, match_thresh, iter_max, init_type,
                                                       reweight_distance, *rel_error, *rel_error_score, max_s_rho, match_num_thresh,
                                                       init_iter, W_Xf);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) W_X(i, j) = W_Xf(i, j);
    } else
    F = c->slots[slot_new].gt->Minimizer_RV<double>(Vv, Wv, RV, RW, *c->slots[slot_old].ef, match_thresh,
                                                           iter_max, init_type, reweight_distance, *rel_error,
                                                           *rel_error_score, max_s_rho, match_num_thresh,
                                                           init_iter, W_X);
    putv(V, Vv);
    putv(W, Wv);
    put3(RVel, RV);
    put3(RW0, RW);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) W_Xo[i * 6 + j] = W_X(i, j);
    return F;
}

double ref_minimizer_rv_kf(void *ctx, int slot_kf, int slot_cur, double X[6], double Kr, double match_mod, double match_ang,
                           double rho_tol, int iter_max, double reweight_distance, double max_s_rho, unsigned match_num_thresh,
                           double RRV[36], int *mnum) {
    Ctx *c = (Ctx *)ctx;
    edge_tracker &et = *c->slots[slot_cur].ef;
    Vector<3> Vel = makeVector(X[0], X[1], X[2]), W0 = makeVector(X[3], X[4], X[5]);
    Matrix<3, 3> RVel = Identity * 1e50, RW0 = Identity * 1e50;       // kfvo.cpp:66-67
    for (KeyLine &kl : et) kl.m_id_f = -1;                             // kfvo.cpp:71-72
    Vector<6, double> Xv;
    Matrix<6> R;
    const double r = kfvo::Minimizer_RV_KF<double>(Vel, W0, RVel, RW0, *c->slots[slot_kf].gt, et, c->cam, Kr, match_mod, match_ang, rho_tol,
                                                   iter_max, reweight_distance, max_s_rho, match_num_thresh, Xv, R);   // kfvo.cpp:74
    int n = 0;
    for (KeyLine &kl : et) n += kl.m_id_f >= 0;                        // kfvo.cpp:76-82
    *mnum = n;
    for (int i = 0; i < 6; i++) {
        X[i] = i < 3 ? Vel[i] : W0[i - 3];
        for (int j = 0; j < 6; j++) RRV[i * 6 + j] = R(i, j);
    }
    return r;
}

double ref_minimizer_v(void *ctx, int slot_new, int slot_old, double V[3], double RVel[9], double match_thresh,
                       int iter_max, double s_rho_min, unsigned match_num_thresh, double reweight_distance, float min_mod) {
    Ctx *c = (Ctx *)ctx;
    Vector<3> Vv = v3(V);
    Matrix<3, 3> RV = Identity;
    double F = c->slots[slot_new].gt->Minimizer_V<double>(Vv, RV, *c->slots[slot_old].ef, match_thresh, iter_max, s_rho_min,
                                                          match_num_thresh, reweight_distance, min_mod);   // rebvo_second_t.cpp:223
    putv(V, Vv);
    put3(RVel, RV);
    return F;
}

int ref_ext_rot_vel(void *ctx, int slot, const double vel[3], double loc_unc, double hub_reweight, double X[6], double Wx[36],
                    double Rx[36]) {
    Ctx *c = (Ctx *)ctx;
    Matrix<6, 6> W, R;
    Vector<6> Xv;
    const bool ok = c->slots[slot].ef->ExtRotVel(v3(vel), W, R, Xv, loc_unc, hub_reweight);   // rebvo_second_t.cpp:237
    for (int i = 0; i < 6; i++) {
        X[i] = Xv[i];
        for (int j = 0; j < 6; j++) { Wx[i * 6 + j] = W(i, j); Rx[i * 6 + j] = R(i, j); }
    }
    return ok;
}

int ref_forward_match(void *ctx, int slot_old, int slot_new) {
    Ctx *c = (Ctx *)ctx;
    return c->slots[slot_old].ef->FordwardMatch(c->slots[slot_new].ef);
}
void ref_rotate_keylines(void *ctx, int slot, const double R[9]) {
    ((Ctx *)ctx)->slots[slot].ef->rotate_keylines(m3(R));
}
int ref_directed_matching(void *ctx, int slot_new, int slot_old, const double V[3], const double RVel[9],
                          const double BackRot[9], int *kf_matchs, double min_thr_mod, double min_thr_ang,
                          double max_radius, double loc_unc) {
    Ctx *c = (Ctx *)ctx;
    return c->slots[slot_new].ef->directed_matching(v3(V), m3(RVel), m3(BackRot), c->slots[slot_old].ef, *kf_matchs,
                                                    min_thr_mod, min_thr_ang, max_radius, loc_unc, c->stereo_mode);
}
// ---- stereo (REBVO/StereoAvaiable; reference only) ----
// the pair camera's own cam_model: re-creates the slot's objects the way rebvo.cpp:303-311 builds ss_pair / ef_pair
void ref_set_slot_cam(void *ctx, int slot, double ppx, double ppy, double zfx, double zfy) {
    Ctx *c = (Ctx *)ctx;
    Slot &s = c->slots[slot];
    const OrcParams &p = c->p;
    cam_model::rad_tan_distortion kc = {p.kc[0], p.kc[1], p.kc[2], p.kc[3], p.kc[4]};
    Size2D sz = {(u_int)p.w, (u_int)p.h};
    cam_model cam({(float)ppx, (float)ppy}, {(float)zfx, (float)zfy}, kc, sz);
    delete s.gt; delete s.ef; delete s.ss;
    s.ss = new sspace(p.sigma0, p.ksigma, cam.sz, 3);
    s.ef = new edge_tracker(cam, 255 * 3);
    s.gt = new global_tracker(s.ef->GetCam());
    memset(s.ss->ImgDx().Data(), 0, sizeof(float) * p.w * p.h);
    memset(s.ss->ImgDy().Data(), 0, sizeof(float) * p.w * p.h);
    memset(s.gt->field.Data(), 0, sizeof(gt_field_data) * p.w * p.h);
}
void ref_set_stereo_mode(void *ctx, int on) { ((Ctx *)ctx)->stereo_mode = on != 0; }
void ref_set_tracker_f32(void *ctx, int on) { ((Ctx *)ctx)->tracker_f32 = on != 0; }
int ref_directed_matching_stereo(void *ctx, int slot, int slot_pair, const double t[3], const double R[9], double min_thr_mod,
                                 double min_thr_ang, double max_radius, double loc_unc, double q_abs, double q_rel,
                                 double loc_unc_model) {
    Ctx *c = (Ctx *)ctx;
    Vector<3> tv = v3(t);
    Matrix<3, 3> Rm = m3(R);
    return c->slots[slot].ef->directed_matching_stereo(tv, Rm, c->slots[slot_pair].ef, min_thr_mod, min_thr_ang, max_radius, loc_unc,
                                                       q_abs, q_rel, loc_unc_model);
}
void ref_fuse_stereo_depth(void *ctx, int slot) { ((Ctx *)ctx)->slots[slot].ef->fuseStereoDepth(); }
// whole-frame stereo: one more slot behind the ring for the pair image's edge map, built with the pair camera
void ref_enable_stereo(void *ctx, double ppx, double ppy, double zfx, double zfy, const double t[3], const double R[9],
                       double max_radius) {
    Ctx *c = (Ctx *)ctx;
    if (!c->rig) {
        Slot s = {nullptr, nullptr, nullptr, new Image<float>(c->cam.sz), new Image<RGB24Pixel>(c->cam.sz)};
        c->slots.push_back(s);
    }
    ref_set_slot_cam(ctx, c->ring, ppx, ppy, zfx, zfy);
    c->rig = true;
    c->stereo_mode = true;
    c->rig_t = v3(t);
    c->rig_R = m3(R);
    c->rig_radius = max_radius;
}

int ref_regularize(void *ctx, int slot, double thresh) {
    return ((Ctx *)ctx)->slots[slot].ef->Regularize_1_iter(thresh);
}
void ref_ekf(void *ctx, int slot, const double V[3], const double RVel[9], const double RW0[9], double q_abs,
             double q_rel, double loc_unc) {
    ((Ctx *)ctx)->slots[slot].ef->UpdateInverseDepthKalman(v3(V), m3(RVel), m3(RW0), q_abs, q_rel, loc_unc);
}
double ref_rescale(void *ctx, int slot, double *RKp, double s_rho_min, unsigned match_num_min, int re_escale) {
    return ((Ctx *)ctx)->slots[slot].ef->EstimateReScalingOpt(*RKp, s_rho_min, match_num_min, re_escale != 0);
}

// One frame through FirstThr + SecondThread (ImuMode==0).  Returns 1 when stage B/C ran (frame>=1).
// FirstThr's part of a frame (rebvo_first_t.cpp:259-290): stage A of frame number `frame` into its ring slot.  What
// SecondThread reports of it is parked with the slot (in the reference both threads see the same PipeBuffer).
struct ARec { int kn; double tresh; float retuned; double dtp0; };
static ARec g_unused_arec;
static void frame_a(Ctx *c, int frame, const uint8_t *rgb24, ARec *rec) {
    const int sn = frame % c->ring;
    const double t0 = now();
    ref_stage_a(c, sn, rgb24, &c->tresh, &c->l_kl_num);
    if (c->rig && c->pair_rgb) ref_stage_a(c, c->ring, c->pair_rgb, &c->tresh, &c->l_kl_num);   // rebvo_first_t.cpp:275-290
    rec->dtp0 = now() - t0;
    rec->kn = c->slots[sn].ef->KNum();
    rec->tresh = c->tresh;
    rec->retuned = c->slots[sn].ef->getThresh();
}
static int frame_bc(Ctx *c, const ARec &rec, double t, OrcNav *nav);

int ref_process_frame(void *ctx, const uint8_t *rgb24, double t, OrcNav *nav) {
    Ctx *c = (Ctx *)ctx;
    ARec rec;
    frame_a(c, c->frame, rgb24, &rec);
    return frame_bc(c, rec, t, nav);
}

// SecondThread's part (rebvo_second_t.cpp:128-629, ImuMode == 0): frame c->frame against frame c->frame - 1.
static int frame_bc(Ctx *c, const ARec &rec, double t, OrcNav *nav) {
    const OrcParams &p = c->p;
    const int ns = c->ring;
    const int sn = c->frame % ns, so = (c->frame + ns - 1) % ns;
    memset(nav, 0, sizeof(*nav));
    nav->dtp0 = rec.dtp0;
    Slot &nb = c->slots[sn];
    nav->frame = c->frame;
    nav->t = t;
    nav->kn = rec.kn;
    nav->tresh = rec.tresh;
    nav->retuned_thresh = rec.retuned;

    if (c->frame == 0) {  // "dummy processing of the first frame" rebvo_second_t.cpp:108-121
        c->frame++;
        c->t_prev = t;
        return 0;
    }
    Slot &ob = c->slots[so];
    double t1 = now();
    bool EstimationOk = true;
    double dt_frame = t - c->t_prev;                                 // :145
    if (dt_frame < 0.001) dt_frame = 1 / p.config_fps;
    int klm_num = 0, num_kf_back_m = 0;
    Matrix<3, 3> P_V = Identity * 1e50, P_W = Identity * 1e50, R = Identity;  // :166-168
    Vector<3> &V = c->V, &W = c->W;
    double error_vel = 0, error_score = 0;

    double s_rho_q = ob.ef->EstimateQuantile(RHO_MIN, RHO_MAX, p.qcut_quantile, p.qcut_nbins);  // :172
    nb.gt->build_field(*nb.ef, p.search_range, nb.ef->getThresh());                            // :177
    TooN::Matrix<6, 6, double> W_X;
    if (c->tracker_f32) {
        TooN::Matrix<6, 6, float> W_Xf;
        nav->score = nb.gt->Minimizer_RV<float>(V, W, P_V, P_W, *ob.ef, p.tracker_match_thresh, p.tracker_iter_num,
                                                p.tracker_init_type, p.reweight_distance, error_vel, error_score,
                                                s_rho_q, p.match_num_thresh, p.tracker_init_iter_num, W_Xf);  // :343 (USE_NE10)
    } else
    nav->score = nb.gt->Minimizer_RV<double>(V, W, P_V, P_W, *ob.ef, p.tracker_match_thresh, p.tracker_iter_num,
                                             p.tracker_init_type, p.reweight_distance, error_vel, error_score,
                                             s_rho_q, p.match_num_thresh, p.tracker_init_iter_num, W_X);  // :346
    nav->klm_fwd = ob.ef->FordwardMatch(nb.ef);                      // :354
    SO3<> R0(W);                                                     // :360
    R.T() = R0.get_matrix() * R.T();                                 // :361
    ob.ef->rotate_keylines(R0.get_matrix());                         // :369
    putv(nav->V, V);
    putv(nav->W, W);
    put3(nav->P_V, P_V);
    put3(nav->P_W, P_W);

    if (util::isNaN(V) || util::isNaN(W)) {                          // :387-397
        P_V = Identity * 1e50;
        V = Zeros;
        c->Kp = 1;
        c->P_Kp = 1e50;
        EstimationOk = false;
    } else {
        klm_num = nb.ef->directed_matching(V, P_V, R, ob.ef, num_kf_back_m, p.match_thresh_module,
                                           p.match_thresh_angle, p.search_range, p.loc_unc_match, c->stereo_mode);  // :410
        if (klm_num < p.global_match_threshold) {                    // :412-422
            P_V = Identity * 1e50;
            V = Zeros;
            c->Kp = 1;
            c->P_Kp = 10;
            EstimationOk = false;
        } else {
            nb.ef->Regularize_1_iter(p.regularize_thresh);           // :453
            nb.ef->UpdateInverseDepthKalman(V, P_V, P_W, p.reshape_q_abs, p.reshape_q_rel, p.loc_unc);  // :460
            if (c->rig) {                                                                               // :465-486
                nav->pad0 = nb.ef->directed_matching_stereo(c->rig_t, c->rig_R, c->slots[c->ring].ef, p.match_thresh_module,
                                                            p.match_thresh_angle, c->rig_radius, p.loc_unc_match, p.reshape_q_abs,
                                                            p.reshape_q_rel, p.loc_unc);               // stereo_match_num
                nb.ef->fuseStereoDepth();
                c->Kp = 1;
            } else {
                c->Kp = nb.ef->EstimateReScalingOpt(c->P_Kp, RHO_MAX, 1, p.do_rescaling > 0);         // :487
            }
        }
    }
    c->Pose = c->Pose * R;                                           // :550
    c->Pos += -c->Pose * V * c->K;                                   // :551
    nav->dtp1 = now() - t1;

    nav->dt = dt_frame;
    nav->Kp = c->Kp;
    nav->RKp = c->P_Kp;
    nav->s_rho_q = s_rho_q;
    nav->rel_error = error_vel;
    nav->rel_error_score = error_score;
    put3(nav->Rot, R);
    putv(nav->RotLie, SO3<>(R).ln());
    putv(nav->Vel, -V * c->K / dt_frame);
    put3(nav->Pose, c->Pose);
    putv(nav->PoseLie, SO3<>(c->Pose).ln());
    putv(nav->Pos, c->Pos);
    nav->klm_num = klm_num;
    nav->kf_matchs = num_kf_back_m;
    nav->estimation_ok = EstimationOk;
    // nav->V/W above are the tracker outputs; the state carried forward may have been reset (V=0)
    c->frame++;
    c->t_prev = t;
    return 1;
}

// Replay n frames (frame k = pool + idx[k] * frame_bytes, time stamp t0 + k * dt) and time them like the reference runs
// them: threads == 1: stage A and stages B/C back to back on the calling thread; threads == 2: the reference's own
// threading — FirstThr (stage A of frame k+1) next to SecondThread (B/C of frame k), rebvo_first_t.cpp:134 /
// rebvo_second_t.cpp:102, coupled through the frame ring like Pipeline<PipeBuffer> (a slot is handed on when its
// predecessor stage released it).  done_s[k] = seconds since the start at which frame k left stage B/C.
int ref_run_sequence(void *ctx, const uint8_t *pool, unsigned long long frame_bytes, const int *idx, int n, double t0, double dt,
                     int threads, double *done_s, OrcNav *navs) {
    Ctx *c = (Ctx *)ctx;
    std::vector<ARec> recs(n);
    OrcNav scratch;
    const double start = now();
    if (threads <= 1) {
        for (int k = 0; k < n; k++) {
            frame_a(c, c->frame, pool + (size_t)idx[k] * frame_bytes, &recs[k]);
            frame_bc(c, recs[k], t0 + k * dt, navs ? &navs[k] : &scratch);
            done_s[k] = now() - start;
        }
        return n;
    }
    const int first = c->frame;
    std::atomic<int> a_done(0), bc_done(0);
    std::thread first_thr([&]() {
        for (int k = 0; k < n; k++) {
            // the slot of frame k still serves as the "old" frame of k - ring + 1: wait for SecondThread
            while (k - bc_done.load(std::memory_order_acquire) > c->ring - 2) std::this_thread::yield();
            frame_a(c, first + k, pool + (size_t)idx[k] * frame_bytes, &recs[k]);
            a_done.store(k + 1, std::memory_order_release);
        }
    });
    for (int k = 0; k < n; k++) {
        while (a_done.load(std::memory_order_acquire) <= k) std::this_thread::yield();
        frame_bc(c, recs[k], t0 + k * dt, navs ? &navs[k] : &scratch);
        bc_done.store(k + 1, std::memory_order_release);
        done_s[k] = now() - start;
    }
    first_thr.join();
    return n;
}

}  // extern "C"

// the visualizer wire format (src/CommLib/net_keypoint.cpp:29-108); 15-byte records
extern "C" void ref_get_seq_state(void *ctx, OrcSeqState *o) {
    Ctx *c = (Ctx *)ctx;
    o->tresh = c->tresh; o->t_prev = c->t_prev; o->Kp = c->Kp; o->K = c->K; o->P_Kp = c->P_Kp;
    putv(o->V, c->V); putv(o->W, c->W); putv(o->Pos, c->Pos); put3(o->Pose, c->Pose);
    o->l_kl_num = c->l_kl_num; o->frame = c->frame;
}
extern "C" void ref_svd_backsub(const double A[36], const double b[6], double h[6]) {
    Matrix<6, 6> M; Vector<6> v;
    for (int i = 0; i < 6; i++) { v[i] = b[i]; for (int j = 0; j < 6; j++) M(i, j) = A[i * 6 + j]; }
    SVD<> svd(M);
    Vector<6> r = svd.backsub(v);
    for (int i = 0; i < 6; i++) h[i] = r[i];
}
extern "C" void ref_chol_backsub(const double A[36], const double b[6], double h[6]) {
    Matrix<6, 6> M; Vector<6> v;
    for (int i = 0; i < 6; i++) { v[i] = b[i]; for (int j = 0; j < 6; j++) M(i, j) = A[i * 6 + j]; }
    Cholesky<6> ch(M);
    Vector<6> r = ch.backsub(v);
    for (int i = 0; i < 6; i++) h[i] = r[i];
}
extern "C" int ref_copy_net_keyline(void *ctx, int slot, int slot_pair, void *out, int kl_size, double k_prof) {
    Ctx *c = (Ctx *)ctx;
    return copy_net_keyline(*c->slots[slot].ef, slot_pair >= 0 ? c->slots[slot_pair].ef : nullptr, (net_keyline *)out, kl_size, k_prof);
}
extern "C" int ref_copy_net_keyline_nextid(void *ctx, int slot, void *out, int kl_size) {
    return copy_net_keyline_nextid(*((Ctx *)ctx)->slots[slot].ef, (net_keyline *)out, kl_size);
}

extern "C" int ref_process_frame_stereo(void *ctx, const uint8_t *rgb24, const uint8_t *rgb24_pair, double t, OrcNav *nav) {
    Ctx *c = (Ctx *)ctx;
    c->pair_rgb = rgb24_pair;
    const int r = ref_process_frame(ctx, rgb24, t, nav);
    c->pair_rgb = nullptr;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// IMU branch: the reference's filters and grabber behind flat entry points, and the ImuMode > 0 frame sequence
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <int N> Vector<N> vN(const double *p) { Vector<N> v; for (int i = 0; i < N; i++) v[i] = p[i]; return v; }
template <int N> Matrix<N, N> mN(const double *p) { Matrix<N, N> m; for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) m(i, j) = p[i * N + j]; return m; }
template <int N> void putN(double *p, const Vector<N> &v) { for (int i = 0; i < N; i++) p[i] = v[i]; }
template <int N> void putM(double *p, const Matrix<N, N> &m) { for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) p[i * N + j] = m(i, j); }
void put_imu(OrcImuIntegrated *o, const IntegratedImuData &d) {
    o->n = d.n; o->pad = 0; o->dt = d.dt;
    putM<3>(o->Rot, d.Rot); putN<3>(o->giro, d.giro); putN<3>(o->acel, d.acel); putN<3>(o->comp, d.comp);
    putN<3>(o->dgiro, d.dgiro); putN<3>(o->cacel, d.cacel);
}
}  // namespace

extern "C" {

void ref_imu_bias_correct(double *X, double *Wx, double *Gb, double *Wb, const double *Rg, const double *Rb) {
    Vector<6> x = vN<6>(X);
    Matrix<6, 6> wx = mN<6>(Wx);
    Vector<3> gb = vN<3>(Gb);
    Matrix<3, 3> wb = mN<3>(Wb);
    edge_tracker::BiasCorrect(x, wx, gb, wb, mN<3>(Rg), mN<3>(Rb));
    putN<6>(X, x); putM<6>(Wx, wx); putN<3>(Gb, gb); putM<3>(Wb, wb);
}
void ref_est_acel_lsq4(const double *vel, double *acel, const double *R, double dt) {
    Vector<3> a = vN<3>(acel);
    ScaleEstimator::EstAcelLsq4(vN<3>(vel), a, mN<3>(R), dt);
    putN<3>(acel, a);
}
void ref_mean_acel4(const double *s_acel, double *acel, const double *R) {
    Vector<3> a = vN<3>(acel);
    ScaleEstimator::MeanAcel4(vN<3>(s_acel), a, mN<3>(R));
    putN<3>(acel, a);
}
double ref_est_ka_gmek_bias(const double *s_acel, const double *f_acel, double kP, const double *Rot, double *X, double *P,
                            const double *Qg, const double *Qrot, const double *Qbias, double QKp, double Rg,
                            const double *Rs, const double *Rf, double *g_est, double *b_est, const double *Wvw,
                            double *Xvw, double g_gravit) {
    Vector<7> x = vN<7>(X);
    Matrix<7, 7> p = mN<7>(P);
    Vector<3> g = vN<3>(g_est), b = vN<3>(b_est);
    Vector<6> xvw = vN<6>(Xvw);
    const double k = ScaleEstimator::estKaGMEKBias(vN<3>(s_acel), vN<3>(f_acel), kP, mN<3>(Rot), x, p, mN<3>(Qg), mN<3>(Qrot),
                                                   mN<3>(Qbias), QKp, Rg, mN<3>(Rs), mN<3>(Rf), g, b, mN<6>(Wvw), xvw, g_gravit);
    putN<7>(X, x); putM<7>(P, p); putN<3>(g_est, g); putN<3>(b_est, b); putN<6>(Xvw, xvw);
    return k;
}

void *ref_imu_grabber_new(int list_size, double tsamp) { return new ImuGrabber(list_size, tsamp); }
void *ref_imu_grabber_load(const char *csv_file, double time_scale) {
    bool error = false;
    std::vector<ImuData> d = ImuGrabber::LoadDataSet(csv_file, false, time_scale, error);
    if (error) return nullptr;
    return new ImuGrabber(d);
}
void ref_imu_grabber_free(void *g) { delete (ImuGrabber *)g; }
int ref_imu_grabber_set_se3(void *g, const double *R, const double *T) { return ((ImuGrabber *)g)->LoadCamImuSE3(mN<3>(R), vN<3>(T)) ? 1 : 0; }
int ref_imu_grabber_load_se3(void *g, const char *se3_file) {
    try { return ((ImuGrabber *)g)->LoadCamImuSE3(se3_file) ? 1 : 0; } catch (...) { return 0; }
}
int ref_imu_grabber_push(void *g, double tstamp, const double *giro, const double *acel) {
    try { return ((ImuGrabber *)g)->PushData(ImuData(tstamp, vN<3>(giro), vN<3>(acel))) ? 1 : 0; } catch (const std::overflow_error &) { return -1; }
}
void ref_imu_grabber_grab(void *g, double tstart, double tend, OrcImuIntegrated *out) {
    put_imu(out, ((ImuGrabber *)g)->GrabAndIntegrate(tstart, tend));
}
double ref_imu_grabber_tsample(void *g) { return ((ImuGrabber *)g)->tsample; }

// rebvo_second_t.cpp:66-84: IMUState set-up (members the reference leaves to the heap are zeroed here)
void ref_imu_setup(void *ctx, const OrcImuParams *ip) {
    Ctx *c = (Ctx *)ctx;
    c->ip = *ip;
    c->istate = IMUState();
    IMUState &is = c->istate;
    is.g_est = Zeros; is.b_est = Zeros; is.Qrot = Identity; is.Rv = Identity; is.QKp = 0; is.Wvw = Zeros; is.Xvw = Zeros;
    is.W_Bg = util::Matrix3x3Inv(is.RGBias * 100);
    is.Qg = Identity * ip->g_uncert * ip->g_uncert;
    is.Rg = ip->g_module_uncer * ip->g_module_uncer;
    is.Rs = Identity * ip->acel_meas_std * ip->acel_meas_std;
    is.Qbias = Identity * ip->vbias_std * ip->vbias_std;
    is.X = makeVector(M_PI / 4, 0, ip->g_module, 0, 0, 0, 0);
    is.P = makeVector(ip->scale_std_init * ip->scale_std_init, 100, 100, 100, ip->vbias_std * ip->vbias_std * 1e1,
                      ip->vbias_std * ip->vbias_std * 1e1, ip->vbias_std * ip->vbias_std * 1e1).as_diagonal();
    is.u_est = makeVector(1, 0, 0);
    c->n_frame = 0;
    c->n_giro_init = 0;
    c->giro_init = Zeros;
    c->g_init = Zeros;
    c->Rgva = Identity;
}

// One frame through FirstThr + SecondThread with ImuMode > 0 (rebvo_second_t.cpp:128-606).  The pose-graph log and
// key frames are left out (they do not feed back).  Returns 1 when stage B/C ran.
int ref_process_frame_imu(void *ctx, const uint8_t *rgb24, double t, const OrcImuIntegrated *imu_in, OrcNavImu *nav) {
    Ctx *c = (Ctx *)ctx;
    const OrcParams &p = c->p;
    const OrcImuParams &ip = c->ip;
    IMUState &istate = c->istate;
    const int ns = c->ring;
    const int sn = c->frame % ns, so = (c->frame + ns - 1) % ns;
    memset(nav, 0, sizeof(*nav));
    ref_stage_a(ctx, sn, rgb24, &c->tresh, &c->l_kl_num);
    Slot &nb = c->slots[sn];
    nav->kn = nb.ef->KNum();
    if (c->frame == 0) {
        c->frame++;
        c->t_prev = t;
        return 0;
    }
    Slot &ob = c->slots[so];
    IntegratedImuData imu;
    imu.n = imu_in->n; imu.dt = imu_in->dt; imu.Rot = mN<3>(imu_in->Rot); imu.giro = vN<3>(imu_in->giro);
    imu.acel = vN<3>(imu_in->acel); imu.comp = vN<3>(imu_in->comp); imu.dgiro = vN<3>(imu_in->dgiro); imu.cacel = vN<3>(imu_in->cacel);

    bool EstimationOk = true;
    double dt_frame = t - c->t_prev;
    if (dt_frame < 0.001) dt_frame = 1 / p.config_fps;
    int klm_num = 0, num_kf_back_m = 0;
    Matrix<3, 3> P_V = Identity * 1e50, P_W = Identity * 1e50, R = Identity;
    Vector<3> &V = c->V, &W = c->W;
    Matrix<3, 3> &Rgva = c->Rgva;
    double &K = c->K, &Kp = c->Kp, &P_Kp = c->P_Kp;
    const int n_frame = c->n_frame;

    double s_rho_q = ob.ef->EstimateQuantile(RHO_MIN, RHO_MAX, p.qcut_quantile, p.qcut_nbins);   // :172
    nb.gt->build_field(*nb.ef, p.search_range, nb.ef->getThresh());                             // :177
    if (!istate.init && n_frame > 0) {                                                          // :183-203
        if (ip.init_bias > 0) {
            c->giro_init += imu.giro * imu.dt;
            c->g_init -= imu.cacel;
            if (++c->n_giro_init > ip.init_bias_frame_num) {
                istate.Bg = c->giro_init / c->n_giro_init;
                istate.init = true;
                istate.W_Bg = util::Matrix3x3Inv(istate.RGBias * 1e2);
                istate.X.slice<1, 3>() = c->g_init / c->n_giro_init;
            }
        } else {
            istate.init = true;
            istate.Bg = vN<3>(ip.bias_init_guess) * imu.dt;
        }
    }
    R = imu.Rot;                                                                                // :208
    R.T() = TooN::SO3<>(istate.Bg) * R.T();
    ob.ef->rotate_keylines(R.T());
    if (p.tracker_init_type == 0) istate.Vg = Zeros;
    nb.gt->Minimizer_V<double>(istate.Vg, istate.P_Vg, *ob.ef, p.tracker_match_thresh, p.tracker_iter_num, s_rho_q,
                               p.match_num_thresh, p.reweight_distance, ob.ef->getThresh());   // :223
    ob.ef->FordwardMatch(nb.ef);                                                                // :230
    Matrix<6, 6> R_Xv, R_Xgv, W_Xv, W_Xgv;
    Vector<6> Xv, Xgv, Xgva;
    EstimationOk &= nb.ef->ExtRotVel(istate.Vg, W_Xv, R_Xv, Xv, p.loc_unc, p.reweight_distance);   // :237
    istate.dVv = Xv.slice<0, 3>();
    istate.dWv = Xv.slice<3, 3>();
    Xgv = Xv;
    W_Xgv = W_Xv;
    istate.RGBias = Identity * ip.giro_bias_std * ip.giro_bias_std * dt_frame * dt_frame;
    istate.RGiro = Identity * ip.giro_meas_std * ip.giro_meas_std * dt_frame * dt_frame;
    Vector<3> dgbias = Zeros;
    edge_tracker::BiasCorrect(Xgv, W_Xgv, dgbias, istate.W_Bg, istate.RGiro, istate.RGBias);     // :254
    istate.Bg += dgbias;
    istate.dVgv = Xgv.slice<0, 3>();
    istate.dWgv = Xgv.slice<3, 3>();
    Rgva = R;
    SO3<> R0(istate.dWgv);
    R.T() = R0.get_matrix() * R.T();
    istate.Vgv = R0 * istate.Vg + istate.dVgv;
    V = istate.Vgv;
    istate.Wgv = SO3<>(R).ln();
    R_Xgv = Cholesky<6>(W_Xgv).get_inverse();
    P_V = R_Xgv.slice<0, 0, 3, 3>();
    P_W = R_Xgv.slice<3, 3, 3, 3>();
    ScaleEstimator::EstAcelLsq4(-istate.Vgv / dt_frame, istate.Av, R, dt_frame);                // :280
    ScaleEstimator::MeanAcel4(imu.cacel, istate.As, R);
    Xgva = Xgv;
    istate.Rv = (P_V / (dt_frame * dt_frame * dt_frame * dt_frame));
    istate.Qrot = P_W;
    istate.QKp = P_Kp;
    if (n_frame > 4 + ip.init_bias_frame_num) {                                                 // :291-312
        K = ScaleEstimator::estKaGMEKBias(istate.As, istate.Av, 1, R, istate.X, istate.P, istate.Qg, istate.Qrot, istate.Qbias,
                                          istate.QKp, istate.Rg, istate.Rs, istate.Rv, istate.g_est, istate.b_est, W_Xgv, Xgva,
                                          ip.g_module);
        istate.dVgva = Xgva.slice<0, 3>();
        istate.dWgva = Xgva.slice<3, 3>();
        SO3<> R0gva(istate.dWgva);
        Rgva.T() = R0gva.get_matrix() * Rgva.T();
        istate.Vgva = R0gva * istate.Vg + istate.dVgva;
    } else {
        istate.dVgva = istate.dVgv;
        istate.dWgva = istate.dWgv;
        Rgva = R;
        istate.Vgva = istate.Vgv;
    }
    ob.ef->rotate_keylines(R0.get_matrix());                                                    // :319

    if (util::isNaN(V) || util::isNaN(W)) {                                                     // :387-397
        P_V = Identity * 1e50;
        V = Zeros;
        Kp = 1;
        P_Kp = 1e50;
        EstimationOk = false;
    } else {
        klm_num = nb.ef->directed_matching(V, P_V, R, ob.ef, num_kf_back_m, p.match_thresh_module, p.match_thresh_angle,
                                           p.search_range, p.loc_unc_match, false);            // :410
        if (klm_num < p.global_match_threshold) {
            P_V = Identity * 1e50;
            V = Zeros;
            Kp = 1;
            P_Kp = 10;
            EstimationOk = false;
        } else {
            nb.ef->Regularize_1_iter(p.regularize_thresh);
            nb.ef->UpdateInverseDepthKalman(V, P_V, P_W, p.reshape_q_abs, p.reshape_q_rel, p.loc_unc);
            Kp = nb.ef->EstimateReScalingOpt(P_Kp, RHO_MAX, 1, p.do_rescaling > 0);
        }
    }
    if (n_frame > 4 + ip.init_bias_frame_num) {                                                 // :521-541
        istate.u_est = Rgva.T() * istate.u_est;
        istate.u_est = istate.u_est - (istate.u_est * istate.g_est) / (istate.g_est * istate.g_est) * istate.g_est;
        TooN::normalize(istate.u_est);
        Matrix<3> PoseP1 = TooN::SO3<>(istate.g_est, makeVector(0, 1, 0)).get_matrix();
        Matrix<3> PoseP2 = TooN::SO3<>(PoseP1 * istate.u_est, makeVector(1, 0, 0)).get_matrix();
        c->Pose = PoseP2 * PoseP1;
        c->Pos += -c->Pose * istate.Vgva * K;
        istate.Posgva = c->Pos;
        istate.Posgv += -c->Pose * istate.Vgv * K;
    }
    nav->dt = dt_frame; nav->K = K; nav->Kp = Kp; nav->RKp = P_Kp; nav->s_rho_q = s_rho_q;
    put3(nav->Rot, R);
    putv(nav->RotLie, SO3<>(R).ln());
    putv(nav->RotGiro, SO3<>(Rgva).ln() / dt_frame);
    putv(nav->Vel, -V * K / dt_frame);
    put3(nav->Pose, c->Pose);
    putv(nav->PoseLie, SO3<>(c->Pose).ln());
    putv(nav->Pos, c->Pos);
    putv(nav->g, istate.g_est);
    nav->scale = K;
    putv(nav->Vg, istate.Vg); putv(nav->Bg, istate.Bg); putv(nav->dVv, istate.dVv); putv(nav->dWv, istate.dWv);
    putv(nav->Vgv, istate.Vgv); putv(nav->Vgva, istate.Vgva); putv(nav->Av, istate.Av); putv(nav->As, istate.As);
    putN<7>(nav->X, istate.X); putv(nav->b_est, istate.b_est); putv(nav->u_est, istate.u_est);
    nav->klm_num = klm_num;
    nav->estimation_ok = EstimationOk;
    nav->init = istate.init;
    c->frame++;
    c->n_frame++;
    c->t_prev = t;
    return 1;
}

}  // extern "C"
