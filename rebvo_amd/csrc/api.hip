// api.hip — C-ABI entry points of libedgehip.so (include/edgehip.h): context lifetime, HBM layout,
// frame/KeyLine/state exchange and the built-in HIP-event profiler.  Kernels live in stage_*.hip.

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "ctx.h"
#include "stage_a_dev.h"

namespace edgehip {

static thread_local std::string g_err;

// The pinned rings (time stamps, frame indices) have 8 entries and the copies out of them are asynchronous: a caller
// that enqueues frames without synchronising may be more than 8 frames ahead of the device, so wait for the frame that
// last used the entry before overwriting it.
int wait_pinned_ring(edgehip_ctx *c) {
    const int r = c->frames_seen % 8;
    if (c->ring_valid[r]) EH_CHECK(hipEventSynchronize(c->ev_ring[r]));
    return 0;
}

// Before the page-locked index row of a binding is rewritten: stage A reads the row in place, so if `pi` is the row the slot is
// bound through right now, whoever may still be reading it has to finish first — a stage A enqueued by the stage-level entry
// point (bind; edgehip_stage_a; bind again with no synchronisation in between: frames_seen has not moved, it is the same
// row), or, for a binding kept across frames (a pair slot bound once), the frame that last ran in this slot, whose ring entry
// comes round again eight frames later.  The steady replay (a new binding per frame) never waits here: its previous binding
// of the slot lives in another row.
static int wait_idx_row(edgehip_ctx *c, int slot, const int32_t *pi) {
    if (c->slot_src[slot].idx_row != pi) return 0;
    if (c->a_api_valid[slot]) EH_CHECK(hipEventSynchronize(c->ev_a[slot]));
    const int r = c->slot_ring[slot];
    if (r >= 0 && c->ring_valid[r]) EH_CHECK(hipEventSynchronize(c->ev_ring[r]));
    return 0;
}

int wait_upload(edgehip_ctx *c, int slot, hipStream_t st) {
    if (slot >= 0 && slot < 4 && c->up_valid[slot]) {
        EH_CHECK(hipStreamWaitEvent(st, c->ev_up[slot], 0));
        c->up_valid[slot] = false;
    }
    return 0;
}

void drop_frame_graphs(edgehip_ctx *c) {
    if (c->frame_graphs.empty()) return;
    // a graph launched for the frame before may still be executing (a caller with look-ahead uploads re-binds a slot right behind
    // edgehip_process_frame): an exec is destroyed only once the stream it was launched on has drained
    (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->frame_graphs) (void)hipGraphExecDestroy(kv.second);
    c->frame_graphs.clear();
}
int order_a_after_bc(edgehip_ctx *c) {
    if (c->stream_a == c->stream) return 0;   // one stream under two names (no overlap): already in order
    EH_CHECK(hipEventRecord(c->ev_tmp, c->stream));
    EH_CHECK(hipStreamWaitEvent(c->stream_a, c->ev_tmp, 0));
    return 0;
}
int order_bc_after_a(edgehip_ctx *c) {
    if (c->stream_a == c->stream) return 0;
    EH_CHECK(hipEventRecord(c->ev_tmp, c->stream_a));
    EH_CHECK(hipStreamWaitEvent(c->stream, c->ev_tmp, 0));
    return 0;
}
int sync_all(edgehip_ctx *c) {
    if (c->stream_imu) EH_CHECK(hipStreamSynchronize(c->stream_imu));   // it waits for main-stream events: no work is left behind it
    EH_CHECK(hipStreamSynchronize(c->stream_up));
    EH_CHECK(hipStreamSynchronize(c->stream_a));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
void set_error(const std::string &m) { g_err = m; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    g_err = buf;
    (void)hipGetLastError();   // reported: take it out of the thread's last-error slot (it would stay there, ROCm 7)
    return EDGEHIP_ERR_DEVICE;
}
void enter_ctx(edgehip_ctx *c) {
    static thread_local int cur_dev = -1;      // device this thread last made current through us
    int dev = -1;
    if (cur_dev != c->device || hipGetDevice(&dev) != hipSuccess || dev != c->device) {
        (void)hipSetDevice(c->device);
        cur_dev = c->device;
    }
    (void)hipGetLastError();
}

// Alternatives that were built, measured slower than (or equal to) what runs by default, and kept for A/B measurements — the
// global-atomic field scatter (EDGEHIP_FIELD_MODE=1), the two other arrangements of FordwardMatch / rotate_keylines
// (EDGEHIP_FWD_MODE=1,2), two KeyLines per thread in the reweighted evaluation (EDGEHIP_TVR_RW2), evaluation + LM step in one launch
// (EDGEHIP_PERSIST_LM), the undistortion inside the one-kernel stage A's load (EDGEHIP_FUSED_UNDIST), disjoint CU sets for the two
// streams (EDGEHIP_A_CUS) — are compiled only by `make EXPERIMENTS=1`; the default library neither contains them nor reads their
// switches (edgehip_experiments() says which build this is).
#ifdef EDGEHIP_EXPERIMENTS
#define EH_EXP_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define EH_EXP_ENV(name, dflt) (dflt)
#endif

static const char *kProfNames[PROF_COUNT] = {
    "A.rgb_rowscan", "A.colscan", "A.avg_rowscan", "A.detect", "A.compact", "A.join_retune",
    "B.quantile", "B.build_field", "B.tvr_prepare", "B.try_velrot", "B.lm_step",
    "C.forward_match", "C.rotate", "C.directed_matching", "C.regularize_ekf", "C.rescale", "C.pose",
    "A.level", "B.minimizer", "A.fused", "IMU.bias_rototranslation", "IMU.scale_filter_pose", "B.minimizer_v", "C.ext_rot_vel", "B.try_velrot2"};

ProfScope::ProfScope(edgehip_ctx *ctx, int pid, hipStream_t stream) : c(ctx), id(pid), st(stream ? stream : ctx->stream) {
    Profiler *p = c->prof;
    if (!p || !p->on || !((p->mask >> pid) & 1ull)) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!p->pool.empty()) { e = p->pool.back(); p->pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    c->prof->pending.push_back({a, b, id});
}

// Box widths exactly as iigauss::iigauss picks them (src/mtracklib/iigauss.cpp:51-71).
static double kovesi_boxes(double sigma, int box_num, int *box_d) {
    double wideal = sqrt(12 * sigma * sigma / box_num + 1);
    int wl = (int)wideal;
    int tmp = wl / 2;
    if (tmp * 2 == wl) wl--;
    int m = (int)round((3 * box_num + 4 * box_num * wl + box_num * wl * wl - 12 * sigma * sigma) / (4 + 4 * wl));
    int i;
    for (i = 0; i < m; i++) box_d[i] = wl;
    for (; i < box_num; i++) box_d[i] = wl + 2;
    return sqrt((m * wl * wl + (box_num - m) * (wl + 2.0) * (wl + 2.0) - box_num) / 12.0);
}

// Plane-fit pseudo inverse, PInv = Matrix3x3Inv(Phi^T Phi) * Phi^T (edge_finder.cpp:83-100,
// toon_util.h:32-41), evaluated with the same operation order in double.
static void plane_fit_pinv(int win_s, double *pinv /*[3][nn]*/) {
    const int nn = (2 * win_s + 1) * (2 * win_s + 1);
    std::vector<double> Phi(nn * 3);
    for (int i = -win_s, k = 0; i <= win_s; i++)
        for (int j = -win_s; j <= win_s; j++, k++) {
            Phi[k * 3 + 0] = j;
            Phi[k * 3 + 1] = i;
            Phi[k * 3 + 2] = 1;
        }
    double A[3][3];
    for (int r = 0; r < 3; r++)
        for (int cidx = 0; cidx < 3; cidx++) {
            double s = 0;
            for (int k = 0; k < nn; k++) s += Phi[k * 3 + r] * Phi[k * 3 + cidx];
            A[r][cidx] = s;
        }
    double Bm[3][3];
    Bm[0][0] = A[2][2] * A[1][1] - A[2][1] * A[1][2];
    Bm[0][1] = -(A[2][2] * A[0][1] - A[2][1] * A[0][2]);
    Bm[0][2] = A[1][2] * A[0][1] - A[1][1] * A[0][2];
    Bm[1][0] = -(A[2][2] * A[1][0] - A[2][0] * A[1][2]);
    Bm[1][1] = A[2][2] * A[0][0] - A[2][0] * A[0][2];
    Bm[1][2] = -(A[1][2] * A[0][0] - A[1][0] * A[0][2]);
    Bm[2][0] = A[2][1] * A[1][0] - A[2][0] * A[1][1];
    Bm[2][1] = -(A[2][1] * A[0][0] - A[2][0] * A[0][1]);
    Bm[2][2] = A[1][1] * A[0][0] - A[1][0] * A[0][1];
    // Phi^T Phi is diagonal for a symmetric window, so Gaussian elimination returns the plain product
    const double det = A[0][0] * A[1][1] * A[2][2];
    for (int r = 0; r < 3; r++)
        for (int cidx = 0; cidx < 3; cidx++) Bm[r][cidx] = Bm[r][cidx] / det;
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < nn; k++) {
            double s = 0;
            for (int j = 0; j < 3; j++) s += Bm[r][j] * Phi[k * 3 + j];
            pinv[r * nn + k] = s;
        }
}

// Undistortion map exactly as image_undistort::image_undistort builds it (src/VideoLib/image_undistort.cpp:
// 29-95) with cam_model::Img2Hom / distortHom2Hom / Hom2Img (include/UtilLib/cam_model.h:76-98): float
// Point2D arithmetic, double inside distortHom2Hom, weights normalised in float, iw = (int)(w * 65536.f).
// Output per pixel: base = index of tap p00 = floor(id) (kept even when that tap is invalid: taps are at
// base, base+1, base+w, base+w+1) and the four integer weights in tap order, 0 for an invalid tap.  The
// reference compacts valid taps to the front; the integer sum is order independent.
void build_undistort_map(const edgehip_params &p, std::vector<int32_t> &base, std::vector<uint32_t> &iw,
                         std::vector<int32_t> *ref_inx, std::vector<int32_t> *ref_iw) {
    const int w = p.w, h = p.h;
    const float ppx = (float)p.ppx, ppy = (float)p.ppy, zfx = (float)p.zfx, zfy = (float)p.zfy;
    const double zfm = (double)((zfx + zfy) / 2);
    const double Kc2 = p.kc[0], Kc4 = p.kc[1], Kc6 = p.kc[2], P1 = p.kc[3], P2 = p.kc[4];
    base.assign((size_t)w * h, 0);
    iw.assign((size_t)w * h * 4, 0);
    if (ref_inx) ref_inx->assign((size_t)w * h * 4, -1);
    if (ref_iw) ref_iw->assign((size_t)w * h * 4, 0);
    auto valid = [&](float fx, float fy) {
        // Image::isInxValid(const uint&, const uint&): the float arguments convert to unsigned the x86-64 way
        // (cvttss2si to 64 bit, low 32 bits kept): negatives become huge and fail the range test.
        const uint32_t ux = (uint32_t)(int64_t)fx, uy = (uint32_t)(int64_t)fy;
        return ux < (uint32_t)w && uy < (uint32_t)h;
    };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float qx = (float)x - ppx, qy = (float)y - ppy;            // Img2Hom
            {
                const double xp = qx / zfm, yp = qy / zfm;             // distortHom2Hom
                const double r2 = xp * xp + yp * yp;
                const double xpp = xp * (1 + r2 * (Kc2 + r2 * (Kc4 + r2 * Kc6))) + 2 * P1 * xp * yp + P2 * (r2 + 2 * xp * xp);
                const double ypp = yp * (1 + r2 * (Kc2 + r2 * (Kc4 + r2 * Kc6))) + P1 * (r2 + 2 * yp * yp) + 2 * P2 * xp * yp;
                qx = (float)(xpp * zfx);
                qy = (float)(ypp * zfy);
            }
            const float idx = qx + ppx, idy = qy + ppy;                // Hom2Img
            const float fx0 = floorf(idx), fy0 = floorf(idy);
            const float p00x = fx0, p00y = fy0, p11x = fx0 + 1, p11y = fy0 + 1;
            const float tx[4] = {p00x, p11x, p00x, p11x}, ty[4] = {p00y, p00y, p11y, p11y};
            float wgt[4];
            wgt[0] = (p11x - idx) * (p11y - idy);
            wgt[1] = (idx - p00x) * (p11y - idy);
            wgt[2] = (p11x - idx) * (idy - p00y);
            wgt[3] = (idx - p00x) * (idy - p00y);
            bool ok[4];
            float sum_w = 0;
            for (int i = 0; i < 4; i++) {
                ok[i] = valid(tx[i], ty[i]);
                if (ok[i]) sum_w += wgt[i];
            }
            const size_t pix = (size_t)y * w + x;
            // floor coordinates that are far outside still give a well-defined (unused) base
            const long long b = (long long)fy0 * w + (long long)fx0;
            base[pix] = (int32_t)std::max<long long>(std::min<long long>(b, 0x3fffffff), -0x3fffffff);
            int k = 0;
            for (int i = 0; i < 4; i++) {
                if (!ok[i]) continue;
                const float wn = wgt[i] / sum_w;
                const int iwv = (int)(wn * 65536.0f);
                iw[pix * 4 + i] = (uint32_t)iwv;
                if (ref_inx) (*ref_inx)[pix * 4 + k] = (int)roundf(ty[i]) * w + (int)roundf(tx[i]);
                if (ref_iw) (*ref_iw)[pix * 4 + k] = iwv;
                k++;
            }
        }
}

template <typename T>
static int dmalloc(edgehip_ctx *c, T **p, size_t count, std::vector<void *> &track, int fill = -2) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    if (hipMalloc(&q, bytes) != hipSuccess) {
        (void)hipGetLastError();   // the failure is reported here; do not leave it in the thread's last-error slot
        set_error("hipMalloc failed (" + std::to_string(bytes) + " bytes)");
        return EDGEHIP_ERR_MEMORY;
    }
    track.push_back(q);
    if (fill != -2) EH_CHECK(hipMemsetAsync(q, fill, bytes, c->stream));
    *p = (T *)q;
    return 0;
}

struct CtxAllocs {
    std::vector<void *> dev;
    std::vector<void *> host;
};
static std::vector<std::pair<edgehip_ctx *, CtxAllocs *>> g_allocs;
static std::mutex g_allocs_mu;   // contexts are created and destroyed from any thread (one rebvo::REBVO per GPU)

static void init_state_a(const edgehip_params &p, SeqA *s) {
    memset(s, 0, sizeof(*s));
    s->tresh = p.detector_thresh;           // rebvo_first_t.cpp:94
    s->band_trunc = -1;
}
static void init_state(const edgehip_params &p, SeqDev *s) {
    memset(s, 0, sizeof(*s));
    s->pub.tresh = p.detector_thresh;       // rebvo_first_t.cpp:94
    s->pub.Kp = 1; s->pub.K = 1; s->pub.P_Kp = 5e-6;  // rebvo_second_t.cpp:54, 65
    for (int i = 0; i < 3; i++) {
        s->pub.Pose[i * 4] = 1;
        s->pub.R[i * 4] = 1;
        s->pub.P_V[i * 4] = 1e50;
        s->pub.P_W[i * 4] = 1e-10;
    }
    s->band_trunc = -1;
}

// ---- AoS <-> SoA KeyLine exchange ---------------------------------------------------------------------
template <typename T>
static int d2h(edgehip_ctx *c, std::vector<T> &dst, const T *src, size_t count) {
    dst.resize(count);
    if (count == 0) return 0;
    EH_CHECK(hipMemcpyAsync(dst.data(), src, sizeof(T) * count, hipMemcpyDeviceToHost, c->stream));
    return 0;
}
template <typename T>
static int h2d(edgehip_ctx *c, T *dst, const std::vector<T> &src) {
    if (src.empty()) return 0;
    EH_CHECK(hipMemcpyAsync(dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, c->stream));
    return 0;
}


__global__ void k_gather_frames(const uint4 *__restrict__ pool, const int32_t *__restrict__ idx, uint4 *__restrict__ dst,
                                size_t frame_vec) {
    const int seq = blockIdx.y;
    const uint4 *src = pool + (size_t)idx[seq] * frame_vec;
    uint4 *d = dst + (size_t)seq * frame_vec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < frame_vec; i += (size_t)gridDim.x * blockDim.x) d[i] = src[i];
}

}  // namespace edgehip

using namespace edgehip;

extern "C" {

int edgehip_abi_version(void) { return EDGEHIP_ABI_VERSION; }
const char *edgehip_last_error(void) { return g_err.c_str(); }

static int create_body(edgehip_ctx *c, CtxAllocs *al, const edgehip_params &p, int nseq, int nslots, int device);

int edgehip_create(const edgehip_params *params, int nseq, int nslots, int device, edgehip_ctx **out) {
    if (!params || !out || nseq < 1 || nslots < 2) { set_error("edgehip_create: bad argument"); return EDGEHIP_ERR_ARG; }
    const edgehip_params &p = *params;
    // Any image size the reference's own containers take (iimage / sspace / edge_finder are sized from Size2D, iimage.cpp:53-128),
    // up to 2048 columns (the row-prefix kernel keeps at most 32 pixels of a row per lane) and 2^31 bytes per frame.  The
    // one-kernel stage A and the one-pass level kernel serve widths that are a multiple of 4 (up to 788 / 1536 columns); every
    // other width takes the multi-kernel path.
    if (p.w < 16 || p.h < 16 || p.w > 2048 || (size_t)p.w * p.h * 3 >= ((size_t)1 << 31)) {
        set_error("edgehip_create: image width must be in [16,2048], height >= 16, w*h*3 < 2^31");
        return EDGEHIP_ERR_ARG;
    }
    // DetectorPlaneFitSize (win_s of edge_finder::build_mask): the 3x3, 5x5 and 7x7 windows
    if (p.plane_fit_size < 1 || p.plane_fit_size > 3) { set_error("edgehip_create: DetectorPlaneFitSize must be 1, 2 or 3"); return EDGEHIP_ERR_ARG; }
    if (2 * p.plane_fit_size >= p.h || 2 * p.plane_fit_size >= p.w || detect_band_rows(p.w, p.plane_fit_size) == 0) {
        set_error("edgehip_create: image too small / too wide for this DetectorPlaneFitSize");
        return EDGEHIP_ERR_ARG;
    }
    if (p.max_points < 1 || p.qcut_nbins < 1 || p.qcut_nbins > 256 || p.search_range < 1 || p.search_range > 255) {
        set_error("edgehip_create: max_points>=1, 1<=QCutOffNumBins<=256, 1<=SearchRange<=255 required");
        return EDGEHIP_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) {
        set_error("edgehip_create: no usable HIP device (libedgehip has no CPU fallback)");
        return EDGEHIP_ERR_DEVICE;
    }
    EH_CHECK(hipSetDevice(device));
    {   // the kernels are written for gfx950: k_stage_a_fused and k_rescale opt into 128-160 KB of LDS per workgroup and have no
        // smaller form — a device without it is refused here instead of failing at the first frame
        // (stacks that report only the default, non-opt-in limit under MaxSharedMemoryPerBlock: the opt-in figure counts too — the
        // kernels obtain their LDS through hipFuncAttributeMaxDynamicSharedMemorySize)
        int lds_max = 0, lds_optin = 0;
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
        if (hipDeviceGetAttribute(&lds_optin, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess) { (void)hipGetLastError(); lds_optin = 0; }
        if (lds_optin > lds_max) lds_max = lds_optin;
        if (lds_max > 0 && lds_max < 160 * 1024) {
            set_error("edgehip_create: device " + std::to_string(device) + " offers " + std::to_string(lds_max) +
                      " B of LDS per workgroup; libedgehip is built for MI355X / gfx950 (160 KB)");
            return EDGEHIP_ERR_DEVICE;
        }
    }
    edgehip_ctx *c = new edgehip_ctx();
    CtxAllocs *al = new CtxAllocs();
    {
        std::lock_guard<std::mutex> lk(g_allocs_mu);
        g_allocs.push_back({c, al});
    }
    // everything below may fail half-way (out of device memory with a large batch, ...): the partial context is torn
    // down again instead of being leaked
    const int rc = create_body(c, al, p, nseq, nslots, device);
    if (rc != 0) {
        const std::string why = g_err;
        (void)edgehip_destroy(c);
        (void)hipGetLastError();   // whatever made create fail has been reported through rc / edgehip_last_error
        g_err = why;
        return rc;
    }
    *out = c;
    return 0;
}

static int create_body(edgehip_ctx *c, CtxAllocs *al, const edgehip_params &p, int nseq, int nslots, int device) {
    c->p = p;
    c->device = device;
    c->frame_slot = -1;
    c->frames_seen = 0;
    c->ring_slots = nslots;
    // EDGEHIP_GRAPH=1: replay whole-frame HIP graphs.  Off by default: measured 0.793 -> 0.776 ms per frame for a single
    // sequence (tools/experiments/exp_graph.py) — the frame is a chain of ~75 dependent kernels and the time between
    // dependent kernels on the device is the same inside a graph.
    c->use_graph = getenv("EDGEHIP_GRAPH") ? atoi(getenv("EDGEHIP_GRAPH")) != 0 : false;
    c->prof = new Profiler();
    // Stage A gets a stream of its own only when it may overlap stages B/C (EDGEHIP_OVERLAP=1).  Otherwise it is the main
    // stream under another name: a hop between two streams costs a single camera 12-16 us of idle device, twice per frame.
    // EDGEHIP_A_CUS=k (with overlap): the two streams get disjoint CU sets, k of a device's CUs for stage A (bits of the CU
    // mask in numbering order: the driver deals them round-robin over the XCDs) and the rest for the tracker / mapper.
    // Default: on for the batches the multi-kernel stage A serves (a live camera, a handful of sequences): their kernels leave most
    // of the device idle, so the next frame's detection runs beside this frame's tracking and mapping — what the reference's first
    // and second thread do — 0.36 -> 0.32 ms per frame for one sequence, 0.47 -> 0.39 for eight.  Off for whole batches, whose
    // stage A fills every CU by itself (+-0, and per-kernel times stop being attributable), and under frame graphs (one stream).
    // From 32 sequences per launch on stage A is the one-kernel form (one workgroup per sequence, ~0.55 ms whatever the batch) running beside the previous
    // frame's tracking and mapping on the CUs it leaves free: 64 sequences 62.1 -> 89.9 k frames/s, 128: 85.4 -> 107.3 k, 32: 52.3 -> 56.5 k, 24: 45.2 -> 42.7 k
    // (tools/experiments/exp_fused_threshold.sh, profiles/r06_fused_threshold.txt).  It was 192 until round 6: set when the kernel took 0.68 ms.  The number
    // depends on the instantiation (fused_min_batch_for, stage_a_fused.hip: 32 at widths 752 / 640, 64 at 320, 192 for the run-time-width ones).  0 = that rule;
    // EDGEHIP_FUSED_MIN_BATCH sets one number for every width.
    const int fused_min_env = getenv("EDGEHIP_FUSED_MIN_BATCH") ? atoi(getenv("EDGEHIP_FUSED_MIN_BATCH")) : 0;
    int ncu_dev = 0;
    (void)hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, device);
    // (the one-kernel stage A is one workgroup per sequence: below one sequence per CU it leaves CUs idle too — 192 sequences: 86.8 -> 96.0 k frames/s)
    const int ovl_below = fused_min_env > ncu_dev ? fused_min_env : ncu_dev;
    // ... and above it whenever the one-kernel stage A's last round of workgroups is a partial one (one workgroup per sequence and CU: 288 sequences are
    // a full round and one of 32 — the CUs it leaves idle take the previous frame's tracking): 288 sequences 95.8 -> 112.2 k frames/s, 384: 107.5 -> 119.0 k,
    // 640: 114.2 -> 123.7 k (tools/experiments/exp_batch_size.sh, profiles/r06_batch_size_and_overlap.txt).  Whole multiples of the CU count stay on one stream:
    // +1..3 % there, and the kernels' own durations stop being attributable (A.fused 2.14 -> 3.54 ms per launch at 1024 under overlap).
    const bool partial_round = ncu_dev > 0 && nseq > ncu_dev && (nseq % ncu_dev) != 0;
    const bool ovl = getenv("EDGEHIP_OVERLAP") ? atoi(getenv("EDGEHIP_OVERLAP")) != 0 : ((nseq < ovl_below || partial_round) && !c->use_graph);
    c->overlap = ovl ? 1 : 0;
    const int a_cus = ovl ? EH_EXP_ENV("EDGEHIP_A_CUS", 0) : 0;
    if (a_cus > 0) {
        hipDeviceProp_t prop;
        EH_CHECK(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount;
        if (a_cus >= ncu) { set_error("EDGEHIP_A_CUS: must leave CUs for the tracker"); return EDGEHIP_ERR_ARG; }
        const int words = (ncu + 31) / 32;
        std::vector<uint32_t> ma(words, 0u), mb(words, 0u);
        for (int i = 0; i < ncu; i++) (i < a_cus ? ma : mb)[i / 32] |= 1u << (i % 32);
        EH_CHECK(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)words, mb.data()));
        EH_CHECK(hipExtStreamCreateWithCUMask(&c->stream_a, (uint32_t)words, ma.data()));
    } else {
        EH_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (ovl) {
            EH_CHECK(hipStreamCreateWithFlags(&c->stream_a, hipStreamNonBlocking));
        } else {
            c->stream_a = c->stream;
        }
    }
    EH_CHECK(hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) { EH_CHECK(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming)); c->up_valid[i] = false; c->slot_ring[i] = -1; c->a_api_valid[i] = false; c->rec_stale[i] = false; c->grec_ok[i] = false; }
    for (int i = 0; i < 4; i++) {
        EH_CHECK(hipEventCreateWithFlags(&c->ev_a[i], hipEventDisableTiming));
        EH_CHECK(hipEventCreateWithFlags(&c->ev_use[i], hipEventDisableTiming));
        c->use_valid[i] = false;
    }
    EH_CHECK(hipEventCreateWithFlags(&c->ev_tmp, hipEventDisableTiming));
    EH_CHECK(hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    EH_CHECK(hipEventCreateWithFlags(&c->ev_stage8, hipEventDisableTiming));
    for (int i = 0; i < 8; i++) { EH_CHECK(hipEventCreateWithFlags(&c->ev_ring[i], hipEventDisableTiming)); c->ring_valid[i] = false; }
    if (nslots > 4) { set_error("edgehip_create: at most 4 frame slots"); return EDGEHIP_ERR_ARG; }

    DevicePlan &pl = c->plan;
    pl.w = p.w; pl.h = p.h; pl.n = p.w * p.h; pl.nseq = nseq; pl.nslots = nslots;
    pl.ftx = (p.w + 3) / 4;
    pl.fstride = (size_t)pl.ftx * (size_t)((p.h + 3) / 4) * 16;
    pl.f16tx = (p.w + 7) / 8;
    pl.f16stride = (size_t)pl.f16tx * (size_t)((p.h + 3) / 4) * 32;
    pl.cap = std::min(p.max_points, EDGEHIP_KEYLINE_MAX);  // build_mask clamps kl_max to kl_size
    const double sr0 = kovesi_boxes(p.sigma0, kMaxBoxes, pl.box[0]);       // sspace.cpp:45
    kovesi_boxes(sr0 * p.ksigma, kMaxBoxes, pl.box[1]);
    for (int f = 0; f < 2; f++)
        for (int i = 0; i < kMaxBoxes; i++) {
            if (pl.box[f][i] < 1 || pl.box[f][i] * pl.box[f][i] >= kDivLutMax) {
                set_error("edgehip_create: box width out of range for these sigmas");
                return EDGEHIP_ERR_ARG;
            }
            pl.box_a[f][i] = (float)(1.0 / (pl.box[f][i] * pl.box[f][i]));
        }
    pl.ppx = (float)p.ppx; pl.ppy = (float)p.ppy; pl.zfx = (float)p.zfx; pl.zfy = (float)p.zfy;
    pl.zfm = (double)((pl.zfx + pl.zfy) / 2);  // cam_model.h:57: float sum, float /2, then double

    const size_t B = nseq, S = nslots, N = pl.n, CAP = pl.cap;
    int e;
#define EH_TRY(x) if ((e = (x)) != 0) return e
    EH_TRY(dmalloc(c, &c->rgb, S * B * N * 3 + 16, al->dev, 0));   // + slack: k_level reads pixels as aligned 8-byte words
    EH_TRY(dmalloc(c, &c->ii, 4 * B * N, al->dev, 0));
    c->planes = nullptr;
    if (p.debug_planes) EH_TRY(dmalloc(c, &c->planes, 5 * B * N, al->dev, 0));
    EH_TRY(dmalloc(c, &c->mask, S * B * N, al->dev, 0xFF));       // img_mask_kl.Reset(-1)
    EH_TRY(dmalloc(c, &c->field, B * pl.fstride, al->dev, 0xFF));
    EH_TRY(dmalloc(c, &c->field16, B * pl.f16stride, al->dev, 0));
    c->field32_valid = false;
    c->und_base = nullptr;
    c->und_iw = nullptr;
    if (p.use_undistort) {
        EH_TRY(dmalloc(c, &c->und_base, N, al->dev));
        EH_TRY(dmalloc(c, &c->und_iw, N, al->dev));
        std::vector<int32_t> base;
        std::vector<uint32_t> iw;
        build_undistort_map(p, base, iw, nullptr, nullptr);
        EH_CHECK(hipMemcpy(c->und_base, base.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
        EH_CHECK(hipMemcpy(c->und_iw, iw.data(), sizeof(uint32_t) * 4 * N, hipMemcpyHostToDevice));
    }
    EH_TRY(dmalloc(c, &c->div_lut, kDivLutMax, al->dev));
    EH_TRY(dmalloc(c, &c->pinv, 3 * 49, al->dev, 0));
    EH_TRY(dmalloc(c, &c->seq, B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->seqa, B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->tresh_slot, S * B, al->dev, 0));
    c->fc_rows = std::max<int>(nslots, kRefRing);
    c->fc_index = 0;
    EH_TRY(dmalloc(c, &c->framecount, (size_t)c->fc_rows * B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->kn_slot, S * B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->retuned_slot, S * B, al->dev, 0));
    c->band_rows = detect_band_rows(p.w, p.plane_fit_size);
    c->nbands = (p.h - 2 * p.plane_fit_size + c->band_rows - 1) / c->band_rows;
    {
        const int npx = c->band_rows * p.w, nchunk = (npx + 63) / 64, cpw = (nchunk + kDetWaves - 1) / kDetWaves;
        c->band_cap = cpw * 64;
    }
    const size_t nstrips = (size_t)c->nbands * kDetWaves;
    EH_TRY(dmalloc(c, &c->band_cnt, B * nstrips, al->dev, 0));
    EH_TRY(dmalloc(c, &c->band_off, B * (nstrips + 1), al->dev, 0));
    {
        char *bs;
        EH_TRY(dmalloc(c, &bs, B * nstrips * c->band_cap * 20, al->dev));
        c->band_stage = bs;
    }
    EH_TRY(dmalloc(c, &c->histo, B * 256, al->dev, 0));
    c->nblk_tvr = (int)((CAP + kTvrBlock - 1) / kTvrBlock);
    EH_TRY(dmalloc(c, &c->P0, B * 3 * CAP, al->dev, 0));
    EH_TRY(dmalloc(c, &c->resid, (size_t)kResidBufs * B * CAP, al->dev, 0));
    EH_TRY(dmalloc(c, &c->resid_carry, (size_t)kResidBufs * B * c->nblk_tvr, al->dev, 0));
    EH_TRY(dmalloc(c, &c->partials, 2 * B * c->nblk_tvr * kNumSums, al->dev, 0));   // [2]: running chain, zero-init chain (k_try_velrot2)
    EH_TRY(dmalloc(c, &c->block_last, 2 * B * c->nblk_tvr, al->dev, 0));
    EH_TRY(dmalloc(c, &c->bin_cnt, B * 256, al->dev, 0));
    {
        const size_t ntiles = (size_t)((p.w + 63) / 64) * ((p.h + 63) / 64);
        EH_TRY(dmalloc(c, &c->bins, B * ntiles * CAP, al->dev));
    }
    c->field_radius = p.search_range;
    c->field_mode = EH_EXP_ENV("EDGEHIP_FIELD_MODE", 0);
    c->no_grec = getenv("EDGEHIP_NO_GREC") && atoi(getenv("EDGEHIP_NO_GREC")) != 0;
    c->level_mode = getenv("EDGEHIP_LEVEL_MODE") ? atoi(getenv("EDGEHIP_LEVEL_MODE")) : 0;
    c->fwd_mode = EH_EXP_ENV("EDGEHIP_FWD_MODE", 0);
    c->fused_min_batch = fused_min_env;
    // measured at 1024 sequences (tools/experiments/CALLS.md: r04_j, same box, three rounds): the step 10.53 / 10.50 / 10.43 ms without, 10.48 / 10.39 / 10.39 ms with —
    // inside the run-to-run spread, while the group's own HIP-event time went UP (2.62 -> 2.73 ms: 84 registers, 5 waves per SIMD).  Off by default.
    c->fused_undist = EH_EXP_ENV("EDGEHIP_FUSED_UNDIST", 0) != 0;
    c->tvr_rw2 = EH_EXP_ENV("EDGEHIP_TVR_RW2", 0);
    c->dual_init = getenv("EDGEHIP_DUAL_INIT") ? atoi(getenv("EDGEHIP_DUAL_INIT")) : 1;
    c->persist_lm_max = EH_EXP_ENV("EDGEHIP_PERSIST_LM", 0);   // measured: no gain (tools/experiments/exp_single_latency.py), so off
    EH_TRY(dmalloc(c, &c->sync_cnt, B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->fwd_key, B * CAP, al->dev, 0));
    EH_TRY(dmalloc(c, &c->fwd_win, B * CAP, al->dev, 0xFF));
    EH_TRY(dmalloc(c, &c->rs_tmp, B * 2 * CAP, al->dev, 0));
    c->fuse_match = !(getenv("EDGEHIP_FUSE_MATCH") && atoi(getenv("EDGEHIP_FUSE_MATCH")) == 0);
    if (c->fuse_match) {
        EH_TRY(dmalloc(c, &c->rot_pm, S * B * CAP, al->dev, 0)); EH_TRY(dmalloc(c, &c->rot_mm, S * B * CAP, al->dev, 0));
        EH_TRY(dmalloc(c, &c->rot_rho, S * B * CAP, al->dev, 0)); EH_TRY(dmalloc(c, &c->rot_srho, S * B * CAP, al->dev, 0));
    }
    EH_TRY(dmalloc(c, &c->rot_buf, B * 9, al->dev, 0));
    EH_TRY(dmalloc(c, &c->nav_dev, B, al->dev, 0));
    EH_TRY(dmalloc(c, &c->idx_dev, B, al->dev, 0));
    c->slot_src.assign(S, edgehip_ctx::SlotSrc());
    c->stereo_cnt = nullptr;
    if (p.stereo_available) EH_TRY(dmalloc(c, &c->stereo_cnt, B, al->dev, 0));
    c->nav_log = nullptr;
    c->nav_log_len = 0;

    // KeyLine SoA arena
    {
        const size_t al256 = 256;
        auto up = [&](size_t x) { return (x + al256 - 1) / al256 * al256; };
        const bool stereo = p.stereo_available != 0;
        const size_t per = up(CAP * 4) * 8 /* p_inx, n_m, m_id, m_id_f, m_id_kf, m_num, p_id, n_id */ +
                           up(CAP * 8) * 6 /* float2 */ + up(CAP * 8) * 7 /* double */ + up(CAP * 32) + up(CAP * 16) /* grec */ +
                           (stereo ? up(CAP * 4) + 2 * up(CAP * 8) : 0);
        char *arena;
        EH_TRY(dmalloc(c, &arena, per * S * B, al->dev, 0));
        c->kl_arena = arena;
        c->kl.resize(S * B);
        for (size_t sb = 0; sb < S * B; sb++) {
            char *q = arena + sb * per;
            KlSoA &k = c->kl[sb];
            auto take = [&](size_t bytes) { char *r = q; q += up(bytes); return r; };
            k.p_inx = (int32_t *)take(CAP * 4);
            k.m_m = (float2 *)take(CAP * 8); k.u_m = (float2 *)take(CAP * 8); k.c_p = (float2 *)take(CAP * 8);
            k.p_m = (float2 *)take(CAP * 8); k.p_m_0 = (float2 *)take(CAP * 8); k.m_m0 = (float2 *)take(CAP * 8);
            k.n_m = (float *)take(CAP * 4);
            k.rho = (double *)take(CAP * 8); k.s_rho = (double *)take(CAP * 8); k.rho_nr = (double *)take(CAP * 8);
            k.s_rho_nr = (double *)take(CAP * 8); k.rho0 = (double *)take(CAP * 8); k.s_rho0 = (double *)take(CAP * 8);
            k.n_m0 = (double *)take(CAP * 8);
            k.m_id = (int32_t *)take(CAP * 4); k.m_id_f = (int32_t *)take(CAP * 4); k.m_id_kf = (int32_t *)take(CAP * 4);
            k.m_num = (int32_t *)take(CAP * 4); k.p_id = (int32_t *)take(CAP * 4); k.n_id = (int32_t *)take(CAP * 4);
            k.rec = (MatchRec *)take(CAP * 32);
            k.grec = (float4 *)take(CAP * 16);
            k.stereo_m_id = nullptr; k.stereo_rho = nullptr; k.stereo_s_rho = nullptr;
            if (stereo) {
                k.stereo_m_id = (int32_t *)take(CAP * 4);
                k.stereo_rho = (double *)take(CAP * 8); k.stereo_s_rho = (double *)take(CAP * 8);
            }
        }
        c->slot_cam.assign(S, edgehip_ctx::SlotCam{pl.ppx, pl.ppy, pl.zfm});
        EH_TRY(dmalloc(c, &c->kl_dev, S * B, al->dev));
        EH_CHECK(hipMemcpyAsync(c->kl_dev, c->kl.data(), sizeof(KlSoA) * S * B, hipMemcpyHostToDevice, c->stream));
    }
    // constant tables
    {
        std::vector<float> lut(kDivLutMax, 0.f);
        for (int k = 1; k < kDivLutMax; k++) lut[k] = (float)(1.0 / (double)(float)k);  // iimage.cpp:176-178
        EH_CHECK(hipMemcpyAsync(c->div_lut, lut.data(), sizeof(float) * kDivLutMax, hipMemcpyHostToDevice, c->stream));
        double pinv[3 * 49] = {0};
        plane_fit_pinv(p.plane_fit_size, pinv);
        // The 5x5 window (what every shipped configuration uses): k_detect<2> and the one-kernel stage A keep PInv as 5 + 5 + 1
        // coefficients — row 0 varies with the window column only, row 1 with the window row only, row 2 is constant
        // (symmetric window).  Verify instead of assuming.  Other windows are applied as the 3 x n matrix they are.
        if (p.plane_fit_size == 2)
            for (int i = 0; i < 5; i++)
                for (int j = 0; j < 5; j++) {
                    const int k = i * 5 + j;
                    if (!(pinv[k] == pinv[j] && pinv[25 + k] == pinv[25 + 5 * i] && pinv[50 + k] == pinv[50])) {
                        set_error("edgehip_create: plane-fit pseudo inverse lost its separable structure");
                        return EDGEHIP_ERR_STATE;
                    }
                }
        memcpy(c->pinv_host, pinv, sizeof c->pinv_host);
        EH_CHECK(hipMemcpyAsync(c->pinv, pinv, sizeof pinv, hipMemcpyHostToDevice, c->stream));
        EH_CHECK(hipStreamSynchronize(c->stream));  // lut/pinv are stack temporaries
    }
    // pinned staging
    c->pinned_rgb_bytes = B * N * 3;
    {
        void *q;
        EH_CHECK(hipHostMalloc(&q, c->pinned_rgb_bytes, hipHostMallocDefault)); al->host.push_back(q); c->pinned_rgb = (uint8_t *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(SeqDev) * B, hipHostMallocDefault)); al->host.push_back(q); c->pinned_seq = (SeqDev *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(SeqA) * B, hipHostMallocDefault)); al->host.push_back(q); c->pinned_seqa = (SeqA *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(double) * B * 64, hipHostMallocDefault)); al->host.push_back(q); c->pinned_out = (double *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(double) * B * 8, hipHostMallocDefault)); al->host.push_back(q); c->pinned_t = (double *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(int32_t) * B * 8 * 4, hipHostMallocDefault)); al->host.push_back(q); c->pinned_idx = (int32_t *)q;
        EH_CHECK(hipHostMalloc(&q, sizeof(edgehip_nav) * B, hipHostMallocDefault)); al->host.push_back(q); c->pinned_nav = (edgehip_nav *)q;
    }
    for (size_t i = 0; i < B; i++) { init_state(p, &c->pinned_seq[i]); init_state_a(p, &c->pinned_seqa[i]); }
    EH_CHECK(hipMemcpyAsync(c->seq, c->pinned_seq, sizeof(SeqDev) * B, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemcpyAsync(c->seqa, c->pinned_seqa, sizeof(SeqA) * B, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
#undef EH_TRY
    return 0;
}

int edgehip_destroy(edgehip_ctx *c) {
    if (c) drop_frame_graphs(c);
    if (!c) return EDGEHIP_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream_up);
    (void)hipStreamSynchronize(c->stream_a);
    (void)hipStreamSynchronize(c->stream);
    CtxAllocs *mine = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_allocs_mu);
        for (size_t i = 0; i < g_allocs.size(); i++) {
            if (g_allocs[i].first != c) continue;
            mine = g_allocs[i].second;
            g_allocs.erase(g_allocs.begin() + i);
            break;
        }
    }
    if (mine) {
        for (void *q : mine->dev) (void)hipFree(q);
        for (void *q : mine->host) (void)hipHostFree(q);
        delete mine;
    }
    if (c->prof) {
        for (auto &r : c->prof->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : c->prof->pool) (void)hipEventDestroy(e);
        delete c->prof;
    }
    if (c->aos_dev) { (void)hipFree(c->aos_dev); (void)hipHostFree(c->aos_host); (void)hipFree(c->aos_req_dev); (void)hipHostFree(c->aos_req_host); }
    if (c->kl_export) {
        auto *x = c->kl_export;
        if (x->stream) { (void)hipStreamSynchronize(x->stream); (void)hipStreamDestroy(x->stream); }
        for (hipEvent_t e : x->ev_pack) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : x->ev_done) if (e) (void)hipEventDestroy(e);
        if (x->dev) (void)hipFree(x->dev);
        if (x->req) (void)hipHostFree(x->req);
        for (edgehip_keyline *h : x->host) if (h) (void)hipHostFree(h);
        delete x;
        c->kl_export = nullptr;
    }
    if (c->stream_log) {
        (void)hipStreamSynchronize(c->stream_log); (void)hipStreamDestroy(c->stream_log); (void)hipEventDestroy(c->ev_log);
        for (hipEvent_t e : c->ev_log_ring) if (e) (void)hipEventDestroy(e);
    }
    if (c->grey8) (void)hipFree(c->grey8);
    if (c->pinned_grey8) (void)hipHostFree(c->pinned_grey8);
    if (c->nav_log) (void)hipFree(c->nav_log);
    if (c->nav_imu_log) (void)hipFree(c->nav_imu_log);
    if (c->stereo_log) (void)hipFree(c->stereo_log);
    if (c->stream_imu) { (void)hipStreamSynchronize(c->stream_imu); (void)hipStreamDestroy(c->stream_imu); for (int i = 0; i < 2; i++) { (void)hipEventDestroy(c->ev_imu_snap[i]); (void)hipEventDestroy(c->ev_imu_post[i]); (void)hipEventDestroy(c->ev_imu_mid[i]); } }
    if (c->kf_req_dev) (void)hipFree(c->kf_req_dev);
    if (c->kf_res_dev) (void)hipFree(c->kf_res_dev);
    if (c->imu_track) (void)hipFree(c->imu_track);
    if (c->imu_filter) (void)hipFree(c->imu_filter);
    if (c->imu_snap) (void)hipFree(c->imu_snap);
    if (c->imu_in_dev) (void)hipFree(c->imu_in_dev);
    if (c->nav_imu_dev) (void)hipFree(c->nav_imu_dev);
    if (c->pinned_imu) (void)hipHostFree(c->pinned_imu);
    if (c->pinned_nav_imu) (void)hipHostFree(c->pinned_nav_imu);
    for (int i = 0; i < 4; i++) { (void)hipEventDestroy(c->ev_a[i]); (void)hipEventDestroy(c->ev_use[i]); }
    (void)hipEventDestroy(c->ev_tmp);
    (void)hipEventDestroy(c->ev_stage);
    (void)hipEventDestroy(c->ev_stage8);
    for (int i = 0; i < 8; i++) (void)hipEventDestroy(c->ev_ring[i]);
    for (int i = 0; i < 4; i++) (void)hipEventDestroy(c->ev_up[i]);
    (void)hipStreamDestroy(c->stream_up);
    if (c->stream_a != c->stream) (void)hipStreamDestroy(c->stream_a);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int edgehip_reset(edgehip_ctx *c) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    const size_t B = c->plan.nseq, S = c->plan.nslots;
    if (int e = sync_all(c)) return e;
    for (size_t i = 0; i < B; i++) { init_state(c->p, &c->pinned_seq[i]); init_state_a(c->p, &c->pinned_seqa[i]); }
    for (int i = 0; i < 4; i++) { c->use_valid[i] = false; c->rot_pending[i] = false; c->rec_stale[i] = false; }   // (every slot is empty from here on)
    EH_CHECK(hipMemcpyAsync(c->seq, c->pinned_seq, sizeof(SeqDev) * B, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemcpyAsync(c->seqa, c->pinned_seqa, sizeof(SeqA) * B, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemsetAsync(c->framecount, 0, sizeof(uint32_t) * c->fc_rows * B, c->stream));
    EH_CHECK(hipMemsetAsync(c->kn_slot, 0, sizeof(int32_t) * S * B, c->stream));
    if (c->imu_enabled) { if (int e = imu_reset_enqueue(c)) return e; }
    EH_CHECK(hipStreamSynchronize(c->stream));
    c->frame_slot = -1;
    c->frames_seen = 0;
    {   // frame numbers start over: so does the log
        std::lock_guard<std::mutex> g(c->log_mu);
        c->frames_logged = 0;
        c->log_first = 0; c->log_last = -1;
    }
    return 0;
}

int edgehip_sync(edgehip_ctx *c) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    return sync_all(c);
}
void *edgehip_stream(edgehip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int edgehip_box_widths(edgehip_ctx *c, int out[6]) {
    EH_ENTER(c);
    if (!c || !out) return EDGEHIP_ERR_ARG;
    for (int f = 0; f < 2; f++)
        for (int i = 0; i < 3; i++) out[f * 3 + i] = c->plan.box[f][i];
    return 0;
}

static int check_slot(edgehip_ctx *c, int slot) {
    if (!c || slot < 0 || slot >= c->plan.nslots) { set_error("slot out of range"); return EDGEHIP_ERR_ARG; }
    return 0;
}
static int check_seq(edgehip_ctx *c, int seq) {
    if (!c || seq < 0 || seq >= c->plan.nseq) { set_error("sequence out of range"); return EDGEHIP_ERR_ARG; }
    return 0;
}

// the slot reads its own storage again (after edgehip_bind_rgb_indexed)
// the slot's frames come from its own storage again, as RGB24 (grey8 = false) or 8-bit mono (grey8 = true)
static void unbind_rgb(edgehip_ctx *c, int slot, bool grey8 = false) {
    edgehip_ctx::SlotSrc &ss = c->slot_src[slot];
    if (ss.base || ss.grey8 != grey8) drop_frame_graphs(c);   // the frame source and its format are kernel arguments / choices
    ss.base = nullptr;
    ss.host_idx.clear();
    ss.grey8 = grey8;
}
static int ensure_grey8(edgehip_ctx *c) {
    if (c->grey8) return 0;
    void *q = nullptr;
    const size_t bytes = (size_t)c->plan.nslots * c->plan.nseq * c->plan.n;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("grey8 frame storage alloc failed"); return EDGEHIP_ERR_MEMORY; }
    c->grey8 = (uint8_t *)q;
    return 0;
}

int edgehip_upload_rgb(edgehip_ctx *c, int slot, const uint8_t *rgb24, int seq_first, int count) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    unbind_rgb(c, slot);
    if (int e = wait_upload(c, slot, c->stream_a)) return e;
    if (!rgb24 || seq_first < 0 || count < 1 || seq_first + count > c->plan.nseq) { set_error("upload_rgb: bad range"); return EDGEHIP_ERR_ARG; }
    const size_t fb = (size_t)c->plan.n * 3;
    // the pinned buffer is reused: wait for the previous copy out of it (an event, not the stream: with stage A on the main
    // stream that would wait for the whole frame before)
    if (c->stage_busy) EH_CHECK(hipEventSynchronize(c->ev_stage));
    memcpy(c->pinned_rgb + fb * seq_first, rgb24, fb * count);
    EH_CHECK(hipMemcpyAsync(rgbof(c, slot) + fb * seq_first, c->pinned_rgb + fb * seq_first, fb * count,
                            hipMemcpyHostToDevice, c->stream_a));
    EH_CHECK(hipEventRecord(c->ev_stage, c->stream_a));
    c->stage_busy = true;
    return 0;
}

// Page-locked host memory for frames that go straight to the device (no staging copy)
int edgehip_alloc_pinned(size_t bytes, void **out) {
    if (!out || bytes == 0) return EDGEHIP_ERR_ARG;
    EH_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return 0;
}
int edgehip_free_pinned(void *p) {
    if (p) EH_CHECK(hipHostFree(p));
    return 0;
}
// Host memory the caller owns, page-locked in place (hipHostRegister): edgehip_download_keylines_batch copies straight into a
// destination that lies in a registered range instead of through its own staging buffer and a host memcpy.
static std::mutex g_reg_mu;
static std::vector<std::pair<const char *, size_t>> g_reg;
int edgehip_register_host(void *p, size_t bytes) {
    if (!p || bytes == 0) { set_error("register_host: null range"); return EDGEHIP_ERR_ARG; }
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        set_error("register_host: hipHostRegister refused the range");
        return EDGEHIP_ERR_MEMORY;
    }
    std::lock_guard<std::mutex> g(g_reg_mu);
    g_reg.emplace_back(static_cast<const char *>(p), bytes);
    return 0;
}
int edgehip_unregister_host(void *p) {
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto it = std::find_if(g_reg.begin(), g_reg.end(), [&](const std::pair<const char *, size_t> &r) { return r.first == p; });
        if (it == g_reg.end()) { set_error("unregister_host: not a registered range"); return EDGEHIP_ERR_ARG; }
        g_reg.erase(it);
    }
    EH_CHECK(hipHostUnregister(p));
    return 0;
}
static bool host_registered(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    const char *q = static_cast<const char *>(p);
    for (const auto &r : g_reg)
        if (q >= r.first && q + bytes <= r.first + r.second) return true;
    return false;
}
// the page-locked sources of every *_pinned upload so far have been read; the frames themselves may still be in flight
int edgehip_upload_sync(edgehip_ctx *c) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    if (c->stream_up) EH_CHECK(hipStreamSynchronize(c->stream_up));
    return 0;
}
// the page-locked sources of the copies into ONE slot have been read (copies into other slots issued behind them may still run)
int edgehip_upload_wait(edgehip_ctx *c, int slot) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    EH_CHECK(hipEventSynchronize(c->ev_up[slot]));   // (an event that was never recorded counts as complete)
    return 0;
}
int edgehip_upload_rgb_pinned(edgehip_ctx *c, int slot, const uint8_t *rgb24_pinned, int seq_first, int count) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    unbind_rgb(c, slot);
    if (!rgb24_pinned || seq_first < 0 || count < 1 || seq_first + count > c->plan.nseq) { set_error("upload_rgb_pinned: bad range"); return EDGEHIP_ERR_ARG; }
    const size_t fb = (size_t)c->plan.n * 3;
    // On the upload stream: the copy of frame k+1 runs under stage A AND stages B/C of frame k.  It may start once the
    // last frame that was processed in this slot is done (nothing reads a slot's RGB after its own stage A).
    if (c->rig.enabled && slot == c->rig.slot_pair && c->rig_a_valid) EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_a[slot], 0));   // the pair slot: free behind its own stage A
    else if (c->slot_ring[slot] >= 0) EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_ring[c->slot_ring[slot]], 0));
    if (c->a_api_valid[slot]) { EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_a[slot], 0)); c->a_api_valid[slot] = false; }
    EH_CHECK(hipMemcpyAsync(rgbof(c, slot) + fb * seq_first, rgb24_pinned, fb * count, hipMemcpyHostToDevice, c->stream_up));
    EH_CHECK(hipEventRecord(c->ev_up[slot], c->stream_up));
    c->up_valid[slot] = true;
    return 0;
}

// ---- 8-bit mono ingest (EuRoC is mono: datasetcam.cpp:109-171 expands it to RGB24 only because the CPU path wants that) ----
int edgehip_upload_grey8(edgehip_ctx *c, int slot, const uint8_t *grey8, int seq_first, int count) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (!grey8 || seq_first < 0 || count < 1 || seq_first + count > c->plan.nseq) { set_error("upload_grey8: bad range"); return EDGEHIP_ERR_ARG; }
    if (int e = ensure_grey8(c)) return e;
    const size_t fb = c->plan.n;
    if (!c->pinned_grey8) {   // everything that can fail comes before the slot changes its format: a failed call leaves the slot as it was
        void *q = nullptr;
        EH_CHECK(hipHostMalloc(&q, fb * c->plan.nseq, hipHostMallocDefault));
        c->pinned_grey8 = (uint8_t *)q;
    }
    if (int e = wait_upload(c, slot, c->stream_a)) return e;
    unbind_rgb(c, slot, true);
    if (c->stage8_busy) EH_CHECK(hipEventSynchronize(c->ev_stage8));   // the staging buffer is reused: wait for the previous copy out of it
    memcpy(c->pinned_grey8 + fb * seq_first, grey8, fb * count);
    EH_CHECK(hipMemcpyAsync(c->grey8 + ((size_t)slot * c->plan.nseq + seq_first) * fb, c->pinned_grey8 + fb * seq_first, fb * count,
                            hipMemcpyHostToDevice, c->stream_a));
    EH_CHECK(hipEventRecord(c->ev_stage8, c->stream_a));
    c->stage8_busy = true;
    return 0;
}
int edgehip_upload_grey8_pinned(edgehip_ctx *c, int slot, const uint8_t *grey8_pinned, int seq_first, int count) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (!grey8_pinned || seq_first < 0 || count < 1 || seq_first + count > c->plan.nseq) { set_error("upload_grey8_pinned: bad range"); return EDGEHIP_ERR_ARG; }
    if (int e = ensure_grey8(c)) return e;
    unbind_rgb(c, slot, true);
    const size_t fb = c->plan.n;
    // on the upload stream, like edgehip_upload_rgb_pinned: the copy of frame k+1 runs under the whole of frame k
    if (c->rig.enabled && slot == c->rig.slot_pair && c->rig_a_valid) EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_a[slot], 0));   // the pair slot: free behind its own stage A
    else if (c->slot_ring[slot] >= 0) EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_ring[c->slot_ring[slot]], 0));
    if (c->a_api_valid[slot]) { EH_CHECK(hipStreamWaitEvent(c->stream_up, c->ev_a[slot], 0)); c->a_api_valid[slot] = false; }
    EH_CHECK(hipMemcpyAsync(c->grey8 + ((size_t)slot * c->plan.nseq + seq_first) * fb, grey8_pinned, fb * count, hipMemcpyHostToDevice, c->stream_up));
    EH_CHECK(hipEventRecord(c->ev_up[slot], c->stream_up));
    c->up_valid[slot] = true;
    return 0;
}
int edgehip_bind_grey8_indexed(edgehip_ctx *c, int slot, const void *pool_dev, int pool_frames, const int32_t *idx) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (!pool_dev || !idx || pool_frames < 1) return EDGEHIP_ERR_ARG;
    const int B = c->plan.nseq;
    int32_t *pi = c->pinned_idx + ((size_t)(c->frames_seen % 8) * 4 + slot) * B;
    if (int e = wait_pinned_ring(c)) return e;
    for (int s = 0; s < B; s++)    // validate first: a rejected call leaves the current binding's row alone
        if (idx[s] < 0 || idx[s] >= pool_frames) { set_error("bind_grey8_indexed: index out of range"); return EDGEHIP_ERR_ARG; }
    if (int e = wait_idx_row(c, slot, pi)) return e;
    for (int s = 0; s < B; s++) pi[s] = idx[s];
    // stage A reads the row in place (page-locked, device-visible): no copy on the frame's critical path.  The row belongs to
    // this ring entry until the frame that uses it has run (wait_pinned_ring).
    edgehip_ctx::SlotSrc &ss = c->slot_src[slot];
    ss.idx_row = pi;
    ss.idx_ring = c->frames_seen % 8;
    if (ss.base != (const uint8_t *)pool_dev || !ss.grey8) drop_frame_graphs(c);
    ss.base = (const uint8_t *)pool_dev;
    ss.grey8 = true;
    ss.host_idx.assign(idx, idx + B);
    return 0;
}

int edgehip_upload_rgb_device(edgehip_ctx *c, int slot, const void *rgb24_dev) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    unbind_rgb(c, slot);
    if (int e = wait_upload(c, slot, c->stream_a)) return e;
    if (!rgb24_dev) return EDGEHIP_ERR_ARG;
    EH_CHECK(hipMemcpyAsync(rgbof(c, slot), rgb24_dev, (size_t)c->plan.nseq * c->plan.n * 3, hipMemcpyDeviceToDevice, c->stream_a));
    return 0;
}

int edgehip_upload_rgb_indexed(edgehip_ctx *c, int slot, const void *pool_dev, int pool_frames, const int32_t *idx) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    unbind_rgb(c, slot);
    if (int e = wait_upload(c, slot, c->stream_a)) return e;
    if (!pool_dev || !idx || pool_frames < 1) return EDGEHIP_ERR_ARG;
    const int B = c->plan.nseq;
    int32_t *pi = c->pinned_idx + ((size_t)(c->frames_seen % 8) * 4 + slot) * B;   // one row per (ring entry, slot)
    if (int e = wait_pinned_ring(c)) return e;
    for (int s = 0; s < B; s++) {
        if (idx[s] < 0 || idx[s] >= pool_frames) { set_error("upload_rgb_indexed: index out of range"); return EDGEHIP_ERR_ARG; }
        pi[s] = idx[s];
    }
    const size_t fbytes = (size_t)c->plan.n * 3;
    if (fbytes % 16 != 0 || ((uintptr_t)pool_dev & 15) != 0) {
        // frames that are not whole 16-byte vectors (or a pool that does not start on one): one device copy per sequence
        for (int s = 0; s < B; s++)
            EH_CHECK(hipMemcpyAsync(rgbof(c, slot) + fbytes * s, (const uint8_t *)pool_dev + fbytes * idx[s], fbytes, hipMemcpyDeviceToDevice, c->stream_a));
        return 0;
    }
    EH_CHECK(hipMemcpyAsync(c->idx_dev, pi, sizeof(int32_t) * B, hipMemcpyHostToDevice, c->stream_a));
    hipLaunchKernelGGL(k_gather_frames, dim3(64, B), dim3(256), 0, c->stream_a, (const uint4 *)pool_dev, c->idx_dev,
                       (uint4 *)rgbof(c, slot), fbytes / 16);
    EH_LAUNCH_CHECK();
    return 0;
}

int edgehip_bind_rgb_indexed(edgehip_ctx *c, int slot, const void *pool_dev, int pool_frames, const int32_t *idx) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (!pool_dev || !idx || pool_frames < 1) return EDGEHIP_ERR_ARG;
    const int B = c->plan.nseq;
    // One pinned row per (ring entry, slot): several slots may be bound between two process_frame calls (main slot and
    // stereo pair slot, a prefetch), and each copy out of the ring is asynchronous.
    int32_t *pi = c->pinned_idx + ((size_t)(c->frames_seen % 8) * 4 + slot) * B;
    if (int e = wait_pinned_ring(c)) return e;
    for (int s = 0; s < B; s++)    // validate first: a rejected call leaves the current binding's row alone
        if (idx[s] < 0 || idx[s] >= pool_frames) { set_error("bind_rgb_indexed: index out of range"); return EDGEHIP_ERR_ARG; }
    if (int e = wait_idx_row(c, slot, pi)) return e;
    for (int s = 0; s < B; s++) pi[s] = idx[s];
    // stage A reads the row in place (page-locked, device-visible): no copy on the frame's critical path.  The row belongs to
    // this ring entry until the frame that uses it has run (wait_pinned_ring).
    edgehip_ctx::SlotSrc &ss = c->slot_src[slot];
    ss.idx_row = pi;
    ss.idx_ring = c->frames_seen % 8;
    if (ss.base != (const uint8_t *)pool_dev || ss.grey8) drop_frame_graphs(c);   // the frame source is a kernel argument
    ss.base = (const uint8_t *)pool_dev;
    ss.grey8 = false;
    ss.host_idx.assign(idx, idx + B);
    return 0;
}

int edgehip_set_tracker_precision(edgehip_ctx *c, int bits) {
    EH_ENTER(c);
    if (!c || (bits != 32 && bits != 64)) { set_error("set_tracker_precision: 32 or 64"); return EDGEHIP_ERR_ARG; }
    if (bits == 32 && (c->imu_enabled || c->rig.enabled)) { set_error("set_tracker_precision: the float tracker is Minimizer_RV<float> (ImuMode 0, no stereo rig)"); return EDGEHIP_ERR_STATE; }
    EH_CHECK(hipStreamSynchronize(c->stream));
    drop_frame_graphs(c);   // a captured frame holds the kernels of the other precision
    c->tracker_f32 = bits == 32;
    return 0;
}

int edgehip_set_nav_log(edgehip_ctx *c, int len) {
    EH_ENTER(c);
    if (!c || len < 0) return EDGEHIP_ERR_ARG;
    EH_CHECK(hipStreamSynchronize(c->stream));
    drop_frame_graphs(c);   // the per-frame record kernel takes the log pointer as an argument
    if (c->stream_log) EH_CHECK(hipStreamSynchronize(c->stream_log));
    if (c->nav_log) { (void)hipFree(c->nav_log); c->nav_log = nullptr; }
    if (c->nav_imu_log) { (void)hipFree(c->nav_imu_log); c->nav_imu_log = nullptr; }
    if (c->stereo_log) { (void)hipFree(c->stereo_log); c->stereo_log = nullptr; }
    c->nav_log_len = 0;
    c->frames_logged = 0;
    c->log_first = 0; c->log_last = -1;
    if (len > 0 && !c->stream_log) {
        EH_CHECK(hipStreamCreateWithFlags(&c->stream_log, hipStreamNonBlocking));
        EH_CHECK(hipEventCreateWithFlags(&c->ev_log, hipEventDisableTiming));
        for (hipEvent_t &e : c->ev_log_ring) EH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (len > 0) {
        void *q;
        if (hipMalloc(&q, sizeof(edgehip_nav) * (size_t)len * c->plan.nseq) != hipSuccess) {
            (void)hipGetLastError();
            set_error("nav log alloc failed");
            return EDGEHIP_ERR_MEMORY;
        }
        EH_CHECK(hipMemsetAsync(q, 0, sizeof(edgehip_nav) * (size_t)len * c->plan.nseq, c->stream));
        EH_CHECK(hipStreamSynchronize(c->stream));
        c->nav_log = (edgehip_nav *)q;
        if (c->imu_enabled) {   // ImuMode > 0: the IMU half of the records in a ring of the same length
            if (hipMalloc(&q, sizeof(edgehip_nav_imu) * (size_t)len * c->plan.nseq) != hipSuccess) {
                (void)hipGetLastError();
                set_error("nav log alloc failed");
                return EDGEHIP_ERR_MEMORY;
            }
            EH_CHECK(hipMemsetAsync(q, 0, sizeof(edgehip_nav_imu) * (size_t)len * c->plan.nseq, c->stream));
            EH_CHECK(hipStreamSynchronize(c->stream));
            c->nav_imu_log = (edgehip_nav_imu *)q;
        }
        if (c->stereo_cnt) {   // StereoAvaiable: stereo_match_num of every logged frame (edgehip_read_stereo_matches_log)
            if (hipMalloc(&q, sizeof(int32_t) * (size_t)len * c->plan.nseq) != hipSuccess) {
                (void)hipGetLastError();
                set_error("nav log alloc failed");
                return EDGEHIP_ERR_MEMORY;
            }
            EH_CHECK(hipMemsetAsync(q, 0, sizeof(int32_t) * (size_t)len * c->plan.nseq, c->stream));
            EH_CHECK(hipStreamSynchronize(c->stream));
            c->stereo_log = (int32_t *)q;
        }
        c->nav_log_len = len;
    }
    return 0;
}

// Shared body of edgehip_read_nav_log (host destination) and edgehip_read_nav_log_device (device destination).
enum LogPart { LOG_NAV = 0, LOG_IMU = 1, LOG_STEREO = 2 };
static int read_nav_log_impl(edgehip_ctx *c, int first, int count, void *out_v, hipMemcpyKind kind, LogPart part = LOG_NAV) {
    EH_ENTER(c);
    if (part == LOG_IMU && c && !c->nav_imu_log) { set_error("read_nav_imu_log: needs edgehip_imu_enable and edgehip_set_nav_log"); return EDGEHIP_ERR_STATE; }
    if (part == LOG_STEREO && c && !c->stereo_log) { set_error("read_stereo_matches_log: needs a context with stereo_available and edgehip_set_nav_log"); return EDGEHIP_ERR_STATE; }
    const size_t rec = part == LOG_IMU ? sizeof(edgehip_nav_imu) : part == LOG_STEREO ? sizeof(int32_t) : sizeof(edgehip_nav);
    char *out = static_cast<char *>(out_v);
    const char *log = !c ? nullptr : part == LOG_IMU ? reinterpret_cast<const char *>(c->nav_imu_log) : part == LOG_STEREO ? reinterpret_cast<const char *>(c->stereo_log)
                                                                                                   : reinterpret_cast<const char *>(c->nav_log);
    if (!c || !out || !c->nav_log || first < 0 || count < 1 || count > c->nav_log_len) { set_error("read_nav_log: bad range or log disabled"); return EDGEHIP_ERR_ARG; }
    // Records of frames that were never enqueued do not exist; `first` counts frames since edgehip_reset / the first frame
    // (edgehip_nav::frame), the counter frames since edgehip_set_nav_log — equal when the log is set before the first frame.
    const size_t B = c->plan.nseq;
    // [first, first + count) must be frames whose records exist: enqueued since the log was set, and not yet overwritten by a
    // frame one ring length later.  Anything else used to come back as stale or zero records with rc 0.
    auto in_ring = [&](long long last) { return (long long)first + count - 1 <= last && (long long)first > last - c->nav_log_len; };
    {
        std::lock_guard<std::mutex> g(c->log_mu);
        if (c->frames_logged.load() < 1) { set_error("read_nav_log: no frame has been enqueued since the log was set"); return EDGEHIP_ERR_STATE; }
        if ((long long)first < c->log_first || !in_ring(c->log_last)) {
            set_error("read_nav_log: frames " + std::to_string(first) + ".." + std::to_string(first + count - 1) + " are not in the log (it holds " +
                      std::to_string(std::max(c->log_first, c->log_last - c->nav_log_len + 1)) + ".." + std::to_string(c->log_last) + ")");
            return EDGEHIP_ERR_STATE;
        }
        // the newest frame asked for, on whichever stream wrote its record — not everything enqueued since: a caller that keeps
        // frames in flight reads the record of frame k while k + 1 and k + 2 run.  (The ring's events are re-recorded eight frames
        // later: a frame older than that is covered by the newest event.)
        const long long want = (long long)first + count - 1;
        hipEvent_t ev = c->log_last - want < 8 ? c->ev_log_ring[want % 8] : c->ev_log;
        EH_CHECK(hipStreamWaitEvent(c->stream_log, ev, 0));
    }
    for (int k = 0; k < count; k++) {
        const int slot = (first + k) % c->nav_log_len;
        EH_CHECK(hipMemcpyAsync(out + (size_t)k * B * rec, log + (size_t)slot * B * rec, rec * B, kind, c->stream_log));
    }
    EH_CHECK(hipStreamSynchronize(c->stream_log));
    {   // the thread that enqueues frames may have gone on meanwhile: a frame one ring length ahead writes the entries just copied
        std::lock_guard<std::mutex> g(c->log_mu);
        if (!in_ring(c->log_last)) { set_error("read_nav_log: the ring was overwritten during the read (frames enqueued more than its length ahead)"); return EDGEHIP_ERR_STATE; }
    }
    return 0;
}

int edgehip_read_nav_log(edgehip_ctx *c, int first, int count, edgehip_nav *out) { return read_nav_log_impl(c, first, count, out, hipMemcpyDeviceToHost); }

int edgehip_read_nav_log_device(edgehip_ctx *c, int first, int count, void *out_dev) {
    return read_nav_log_impl(c, first, count, out_dev, hipMemcpyDeviceToDevice);
}
int edgehip_read_nav_imu_log(edgehip_ctx *c, int first, int count, edgehip_nav_imu *out) {
    return read_nav_log_impl(c, first, count, out, hipMemcpyDeviceToHost, LOG_IMU);
}
int edgehip_read_stereo_matches_log(edgehip_ctx *c, int first, int count, int32_t *out) {
    return read_nav_log_impl(c, first, count, out, hipMemcpyDeviceToHost, LOG_STEREO);
}

int edgehip_stage_a(edgehip_ctx *c, int slot) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (int e = order_a_after_bc(c)) return e;
    if (int e = wait_upload(c, slot, c->stream_a)) return e;
    if (int e = stage_a_enqueue(c, slot)) return e;
    EH_CHECK(hipEventRecord(c->ev_a[slot], c->stream_a));
    c->a_api_valid[slot] = true;
    return order_bc_after_a(c);
}

static int fetch_states(edgehip_ctx *c) {
    if (int e = sync_all(c)) return e;
    EH_CHECK(hipMemcpyAsync(c->pinned_seq, c->seq, sizeof(SeqDev) * c->plan.nseq, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipMemcpyAsync(c->pinned_seqa, c->seqa, sizeof(SeqA) * c->plan.nseq, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < c->plan.nseq; s++) {   // the public record shows the detector state too
        c->pinned_seq[s].pub.tresh = c->pinned_seqa[s].tresh;
        c->pinned_seq[s].pub.l_kl_num = c->pinned_seqa[s].l_kl_num;
        c->pinned_seq[s].pub.retuned_thresh = c->pinned_seqa[s].retuned;
    }
    return 0;
}

int edgehip_get_kn(edgehip_ctx *c, int slot, int32_t *kn_out) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (!kn_out) return EDGEHIP_ERR_ARG;
    if (int e = sync_all(c)) return e;
    EH_CHECK(hipMemcpyAsync(c->pinned_out, c->kn_slot + (size_t)slot * c->plan.nseq, sizeof(int32_t) * c->plan.nseq,
                            hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    memcpy(kn_out, c->pinned_out, sizeof(int32_t) * c->plan.nseq);
    return 0;
}

int edgehip_get_state(edgehip_ctx *c, int seq, edgehip_seq_state *out) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (!out) return EDGEHIP_ERR_ARG;
    if (int e = fetch_states(c)) return e;
    *out = c->pinned_seq[seq].pub;
    return 0;
}

int edgehip_set_state(edgehip_ctx *c, int seq, const edgehip_seq_state *in) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (!in) return EDGEHIP_ERR_ARG;
    if (int e = fetch_states(c)) return e;
    c->pinned_seq[seq].pub = *in;
    c->pinned_seqa[seq].tresh = in->tresh;
    c->pinned_seqa[seq].l_kl_num = in->l_kl_num;
    c->pinned_seqa[seq].retuned = in->retuned_thresh;
    EH_CHECK(hipMemcpyAsync(c->seq + seq, c->pinned_seq + seq, sizeof(SeqDev), hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemcpyAsync(c->seqa + seq, c->pinned_seqa + seq, sizeof(SeqA), hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_get_framecount(edgehip_ctx *c, int seq, int slot, uint32_t *fc) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (int e = check_slot(c, slot)) return e;
    EH_CHECK(hipMemcpyAsync(fc, c->framecount + (size_t)slot * c->plan.nseq + seq, 4, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}
int edgehip_set_framecount(edgehip_ctx *c, int seq, int slot, uint32_t fc) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (int e = check_slot(c, slot)) return e;
    EH_CHECK(hipMemcpyAsync(c->framecount + (size_t)slot * c->plan.nseq + seq, &fc, 4, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_download_keylines(edgehip_ctx *c, int seq, int slot, edgehip_keyline *kl, int32_t *mask, int32_t *kn_out) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (int e = check_slot(c, slot)) return e;
    if (!kl || !kn_out) return EDGEHIP_ERR_ARG;
    if (int e = rot_materialize_enqueue(c, slot)) return e;   // a slot the whole-frame driver rotated out of place (ctx.h: fuse_match)
    if (int e = sync_all(c)) return e;
    int32_t kn = 0;
    EH_CHECK(hipMemcpyAsync(&kn, c->kn_slot + (size_t)slot * c->plan.nseq + seq, 4, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    const KlSoA &k = klof(c, slot, seq);
    std::vector<int32_t> p_inx, m_id, m_id_f, m_id_kf, m_num, p_id, n_id;
    std::vector<float2> m_m, u_m, c_p, p_m, p_m_0, m_m0;
    std::vector<float> n_m;
    std::vector<double> rho, s_rho, rho_nr, s_rho_nr, rho0, s_rho0, n_m0;
    int e = 0;
    e |= d2h(c, p_inx, k.p_inx, kn); e |= d2h(c, m_id, k.m_id, kn); e |= d2h(c, m_id_f, k.m_id_f, kn);
    e |= d2h(c, m_id_kf, k.m_id_kf, kn); e |= d2h(c, m_num, k.m_num, kn); e |= d2h(c, p_id, k.p_id, kn);
    e |= d2h(c, n_id, k.n_id, kn);
    e |= d2h(c, m_m, k.m_m, kn); e |= d2h(c, u_m, k.u_m, kn); e |= d2h(c, c_p, k.c_p, kn);
    e |= d2h(c, p_m, k.p_m, kn); e |= d2h(c, p_m_0, k.p_m_0, kn); e |= d2h(c, m_m0, k.m_m0, kn);
    e |= d2h(c, n_m, k.n_m, kn);
    e |= d2h(c, rho, k.rho, kn); e |= d2h(c, s_rho, k.s_rho, kn); e |= d2h(c, rho_nr, k.rho_nr, kn);
    e |= d2h(c, s_rho_nr, k.s_rho_nr, kn); e |= d2h(c, rho0, k.rho0, kn); e |= d2h(c, s_rho0, k.s_rho0, kn);
    e |= d2h(c, n_m0, k.n_m0, kn);
    std::vector<int32_t> st_id;
    std::vector<double> st_rho, st_srho;
    if (k.stereo_m_id) { e |= d2h(c, st_id, k.stereo_m_id, kn); e |= d2h(c, st_rho, k.stereo_rho, kn); e |= d2h(c, st_srho, k.stereo_s_rho, kn); }
    if (e) return EDGEHIP_ERR_DEVICE;
    if (mask) EH_CHECK(hipMemcpyAsync(mask, maskof(c, slot) + (size_t)seq * c->plan.n, sizeof(int32_t) * c->plan.n, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < kn; i++) {
        edgehip_keyline &o = kl[i];
        memset(&o, 0, sizeof o);
        o.p_inx = p_inx[i];
        o.m_m[0] = m_m[i].x; o.m_m[1] = m_m[i].y; o.u_m[0] = u_m[i].x; o.u_m[1] = u_m[i].y;
        o.n_m = n_m[i]; o.score = 0.f;
        o.c_p[0] = c_p[i].x; o.c_p[1] = c_p[i].y;
        o.rho = rho[i]; o.s_rho = s_rho[i]; o.rho_nr = rho_nr[i]; o.s_rho_nr = s_rho_nr[i];
        o.rho0 = rho0[i]; o.s_rho0 = s_rho0[i];
        o.p_m[0] = p_m[i].x; o.p_m[1] = p_m[i].y; o.p_m_0[0] = p_m_0[i].x; o.p_m_0[1] = p_m_0[i].y;
        o.m_id = m_id[i]; o.m_id_f = m_id_f[i]; o.m_id_kf = m_id_kf[i]; o.m_num = m_num[i];
        o.m_m0[0] = m_m0[i].x; o.m_m0[1] = m_m0[i].y; o.n_m0 = n_m0[i];
        o.p_id = p_id[i]; o.n_id = n_id[i];
        o.net_id = -1; o.stereo_m_id = -1; o.stereo_rho = 1.0; o.stereo_s_rho = 20.0;  // edge_finder.cpp:183-194
        if (k.stereo_m_id) { o.stereo_m_id = st_id[i]; o.stereo_rho = st_rho[i]; o.stereo_s_rho = st_srho[i]; }
    }
    *kn_out = kn;
    return 0;
}

// One KeyLine of one requested sequence as the reference's 168-byte record (edge_finder.h:45-91): the SoA fields, and the constants
// edgehip_download_keylines gives the fields the device does not keep (score, net_id, the stereo defaults of edge_finder.cpp:183-194)
__global__ __launch_bounds__(256) void k_pack_keylines(const KlSoA *kls, const int32_t *__restrict__ kns, const int32_t *__restrict__ req,
                                                       int32_t *__restrict__ kn_out, edgehip_keyline *__restrict__ out, int cap) {
    const int j = blockIdx.y, seq = req[j];
    const int kn = kns[seq];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && kn_out) kn_out[j] = kn;
    if (i >= kn) return;
    const KlSoA &k = kls[seq];
    edgehip_keyline o = {};   // (the staging buffers are also zeroed when allocated: the record's padding never carries stale device memory)
    o.p_inx = k.p_inx[i];
    const float2 m_m = k.m_m[i], u_m = k.u_m[i], c_p = k.c_p[i], p_m = k.p_m[i], p_m_0 = k.p_m_0[i], m_m0 = k.m_m0[i];
    o.m_m[0] = m_m.x; o.m_m[1] = m_m.y; o.u_m[0] = u_m.x; o.u_m[1] = u_m.y;
    o.n_m = k.n_m[i]; o.score = 0.f;
    o.c_p[0] = c_p.x; o.c_p[1] = c_p.y;
    o.rho = k.rho[i]; o.s_rho = k.s_rho[i]; o.rho_nr = k.rho_nr[i]; o.s_rho_nr = k.s_rho_nr[i]; o.rho0 = k.rho0[i]; o.s_rho0 = k.s_rho0[i];
    o.p_m[0] = p_m.x; o.p_m[1] = p_m.y; o.p_m_0[0] = p_m_0.x; o.p_m_0[1] = p_m_0.y;
    o.m_id = k.m_id[i]; o.m_id_f = k.m_id_f[i]; o.m_id_kf = k.m_id_kf[i]; o.m_num = k.m_num[i];
    o.m_m0[0] = m_m0.x; o.m_m0[1] = m_m0.y; o.n_m0 = k.n_m0[i];
    o.p_id = k.p_id[i]; o.n_id = k.n_id[i];
    o.net_id = -1; o.stereo_m_id = -1; o.stereo_rho = 1.0; o.stereo_s_rho = 20.0;
    if (k.stereo_m_id) { o.stereo_m_id = k.stereo_m_id[i]; o.stereo_rho = k.stereo_rho[i]; o.stereo_s_rho = k.stereo_s_rho[i]; }
    out[(size_t)j * cap + i] = o;
}

int edgehip_download_keylines_batch(edgehip_ctx *c, int slot, int n, const int32_t *seqs, edgehip_keyline *const *kl, int32_t *kn_out) {
    EH_ENTER(c);
    if (int e = check_slot(c, slot)) return e;
    if (n < 1 || !seqs || !kl || !kn_out) { set_error("download_keylines_batch: bad argument"); return EDGEHIP_ERR_ARG; }
    for (int j = 0; j < n; j++) {
        if (int e = check_seq(c, seqs[j])) return e;
        if (!kl[j]) { set_error("download_keylines_batch: null destination"); return EDGEHIP_ERR_ARG; }
    }
    const size_t cap = (size_t)c->plan.cap;
    if (n > c->aos_requests) {   // staging for n requests (grown, never shrunk)
        if (c->aos_dev) { (void)hipFree(c->aos_dev); (void)hipHostFree(c->aos_host); (void)hipFree(c->aos_req_dev); (void)hipHostFree(c->aos_req_host); }
        c->aos_dev = nullptr; c->aos_host = nullptr; c->aos_req_dev = nullptr; c->aos_req_host = nullptr; c->aos_requests = 0;
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(edgehip_keyline) * cap * n) != hipSuccess) { (void)hipGetLastError(); set_error("download_keylines_batch: staging alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->aos_dev = (edgehip_keyline *)q;
        EH_CHECK(hipMemsetAsync(c->aos_dev, 0, sizeof(edgehip_keyline) * cap * n, c->stream));
        if (hipHostMalloc(&q, sizeof(edgehip_keyline) * cap * n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("download_keylines_batch: pinned alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->aos_host = (edgehip_keyline *)q;
        if (hipMalloc(&q, sizeof(int32_t) * 2 * n) != hipSuccess) { (void)hipGetLastError(); set_error("download_keylines_batch: staging alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->aos_req_dev = (int32_t *)q;
        if (hipHostMalloc(&q, sizeof(int32_t) * 2 * n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("download_keylines_batch: pinned alloc failed"); return EDGEHIP_ERR_MEMORY; }
        c->aos_req_host = (int32_t *)q;
        c->aos_requests = n;
    }
    if (int e = rot_materialize_enqueue(c, slot)) return e;   // a slot the whole-frame driver rotated out of place (ctx.h: fuse_match)
    if (int e = sync_all(c)) return e;                        // both frame streams: the slot's producers
    for (int j = 0; j < n; j++) c->aos_req_host[j] = seqs[j];
    EH_CHECK(hipMemcpyAsync(c->aos_req_dev, c->aos_req_host, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pack_keylines, dim3((unsigned)((cap + 255) / 256), (unsigned)n), dim3(256), 0, c->stream, kldev(c, slot),
                       c->kn_slot + (size_t)slot * c->plan.nseq, c->aos_req_dev, c->aos_req_dev + n, c->aos_dev, (int)cap);
    EH_LAUNCH_CHECK();
    EH_CHECK(hipMemcpyAsync(c->aos_req_host + n, c->aos_req_dev + n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    // only the records that exist cross the link — straight into a destination the caller has page-locked (edgehip_register_host),
    // through the staging buffer and a host copy otherwise
    std::vector<char> direct(n, 0);
    for (int j = 0; j < n; j++) {
        const int32_t kn = c->aos_req_host[n + j];
        if (kn <= 0) continue;
        direct[j] = host_registered(kl[j], sizeof(edgehip_keyline) * kn) ? 1 : 0;
        edgehip_keyline *dst = direct[j] ? kl[j] : c->aos_host + (size_t)j * cap;
        EH_CHECK(hipMemcpyAsync(dst, c->aos_dev + (size_t)j * cap, sizeof(edgehip_keyline) * kn, hipMemcpyDeviceToHost, c->stream));
    }
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int j = 0; j < n; j++) {
        const int32_t kn = c->aos_req_host[n + j];
        if (kn > 0 && !direct[j]) memcpy(kl[j], c->aos_host + (size_t)j * cap, sizeof(edgehip_keyline) * kn);
        kn_out[j] = kn;
    }
    return 0;
}

// ---- output callbacks at full pipeline depth (round 6) ----------------------------------------------------------------------------
// What a callback receives of frame k-1 is its edge map as frame k's tracking left it (rebvo_second_t.cpp:622-623; rebvo_third_t.cpp:174):
// the OLD slot of the frame processed last.  edgehip_download_keylines_batch reads it behind a synchronisation of both frame streams,
// and the frame after next detects into that slot — so a caller that wanted KeyLines could keep only one frame in flight.  Here the
// lists are packed in-stream, right behind the frame, into a staging ring of their own; the slot is free again as far as callbacks go,
// the copies to the host run on a stream of their own under the frames that follow, and nothing synchronises the frame streams.
int edgehip_export_keylines(edgehip_ctx *c, int n, const int32_t *seqs, int *ticket_out) {
    EH_ENTER(c);
    if (!c || n < 1 || !seqs || !ticket_out) { set_error("export_keylines: bad argument"); return EDGEHIP_ERR_ARG; }
    for (int j = 0; j < n; j++)
        if (int e = check_seq(c, seqs[j])) return e;
    if (c->frames_seen < 2 || c->frame_slot < 0) { set_error("export_keylines: needs two processed frames (the old slot of a frame pair)"); return EDGEHIP_ERR_STATE; }
    if (!c->kl_export) c->kl_export = new edgehip_ctx::KlExport;
    auto *x = c->kl_export;
    constexpr int R = edgehip_ctx::KlExport::R;
    const size_t cap = (size_t)c->plan.cap;
    if (!x->stream) {
        // Which hardware queue the copies land on decides what waits behind them (HIP maps a context's streams onto four queues per priority
        // class; stage_imu.hip imu_stream_create).  Measured with eight cameras that take every frame's KeyLines: ImuMode 0 — default
        // priority 16.8-17.2 k frames/s, highest 5.6 k (the copies' event waits stall the frame streams); ImuMode 2, where the frame streams
        // are idle half the step under the scale filter — default 5.5-6.1 k, highest 11.3 k.  So: a queue class of its own beside the IMU branch.
        int least = 0, greatest = 0;
        if (!(c->imu_enabled && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
              hipStreamCreateWithPriority(&x->stream, hipStreamNonBlocking, greatest) == hipSuccess)) {
            (void)hipGetLastError();
            EH_CHECK(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
        }
        for (int i = 0; i < R; i++) {
            EH_CHECK(hipEventCreateWithFlags(&x->ev_pack[i], hipEventDisableTiming));
            EH_CHECK(hipEventCreateWithFlags(&x->ev_done[i], hipEventDisableTiming));
        }
    }
    if (n > x->n_cap) {   // staging for n lists per ticket (grown, never shrunk): only with no ticket outstanding
        for (auto &t : x->t)
            if (t.id >= 0) { set_error("export_keylines: more lists than before while tickets are outstanding"); return EDGEHIP_ERR_STATE; }
        EH_CHECK(hipStreamSynchronize(c->stream));
        EH_CHECK(hipStreamSynchronize(x->stream));
        if (x->dev) { (void)hipFree(x->dev); x->dev = nullptr; }
        if (x->req) { (void)hipHostFree(x->req); x->req = nullptr; }
        for (edgehip_keyline *&h : x->host) if (h) { (void)hipHostFree(h); h = nullptr; }
        x->n_cap = 0;
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(edgehip_keyline) * cap * n * R) != hipSuccess) { (void)hipGetLastError(); set_error("export_keylines: staging alloc failed"); return EDGEHIP_ERR_MEMORY; }
        x->dev = (edgehip_keyline *)q;
        EH_CHECK(hipMemsetAsync(x->dev, 0, sizeof(edgehip_keyline) * cap * n * R, c->stream));   // on the stream the packing kernel follows on (hipMemset
                                                                                                   // runs on the null stream, which a non-blocking stream does not wait for)
        if (hipHostMalloc(&q, sizeof(int32_t) * n * R, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("export_keylines: pinned alloc failed"); return EDGEHIP_ERR_MEMORY; }
        x->req = (int32_t *)q;
        x->n_cap = n;
    }
    const int e = (int)(x->next % R);
    if (x->t[e].id >= 0) { set_error("export_keylines: four tickets outstanding (edgehip_export_wait releases one)"); return EDGEHIP_ERR_STATE; }
    const int so = (c->frame_slot - 1 + c->ring_slots) % c->ring_slots;
    if (int er = rot_materialize_enqueue(c, so)) return er;   // a slot the whole-frame driver rotated out of place (ctx.h: fuse_match)
    int32_t *req = x->req + (size_t)e * x->n_cap;
    for (int j = 0; j < n; j++) req[j] = seqs[j];
    hipLaunchKernelGGL(k_pack_keylines, dim3((unsigned)((cap + 255) / 256), (unsigned)n), dim3(256), 0, c->stream, kldev(c, so),
                       c->kn_slot + (size_t)so * c->plan.nseq, req, (int32_t *)nullptr, x->dev + (size_t)e * x->n_cap * cap, (int)cap);
    EH_LAUNCH_CHECK();
    EH_CHECK(hipEventRecord(x->ev_pack[e], c->stream));
    if (c->stream_a != c->stream) {   // the frame after next detects into this slot on the stage-A stream: not before the lists are out
        EH_CHECK(hipEventRecord(c->ev_use[so], c->stream));
        c->use_valid[so] = true;
    }
    x->t[e].id = x->next;
    x->t[e].n = n;
    x->t[e].fetched = false;
    *ticket_out = (int)(x->next & 0x7fffffff);
    x->next++;
    return 0;
}

static edgehip_ctx::KlExport::Ticket *export_ticket(edgehip_ctx *c, int ticket, int &e) {
    auto *x = c ? c->kl_export : nullptr;
    if (!x) return nullptr;
    for (e = 0; e < edgehip_ctx::KlExport::R; e++)
        if (x->t[e].id >= 0 && (int)(x->t[e].id & 0x7fffffff) == ticket) return &x->t[e];
    return nullptr;
}

int edgehip_export_fetch(edgehip_ctx *c, int ticket, const int32_t *kn, edgehip_keyline *const *dst) {
    EH_ENTER(c);
    int e = 0;
    auto *t = export_ticket(c, ticket, e);
    if (!t || !kn || !dst) { set_error("export_fetch: unknown ticket or null argument"); return EDGEHIP_ERR_ARG; }
    if (t->fetched) { set_error("export_fetch: ticket already fetched"); return EDGEHIP_ERR_STATE; }
    auto *x = c->kl_export;
    const size_t cap = (size_t)c->plan.cap;
    for (int j = 0; j < t->n; j++)
        if (kn[j] < 0 || (size_t)kn[j] > cap || (kn[j] > 0 && !dst[j])) { set_error("export_fetch: KeyLine count beyond the capacity, or null destination"); return EDGEHIP_ERR_ARG; }
    EH_CHECK(hipStreamWaitEvent(x->stream, x->ev_pack[e], 0));
    t->staged_dst.assign(t->n, nullptr);
    t->staged_kn.assign(t->n, 0);
    for (int j = 0; j < t->n; j++) {
        if (kn[j] <= 0) continue;
        edgehip_keyline *to = dst[j];
        if (!host_registered(dst[j], sizeof(edgehip_keyline) * kn[j])) {
            // a pageable destination: through a page-locked staging list of the ticket, and a host copy in edgehip_export_wait
            if (!x->host[e]) {
                void *q = nullptr;
                if (hipHostMalloc(&q, sizeof(edgehip_keyline) * cap * x->n_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("export_fetch: pinned alloc failed"); return EDGEHIP_ERR_MEMORY; }
                x->host[e] = (edgehip_keyline *)q;
            }
            to = x->host[e] + (size_t)j * cap;
            t->staged_dst[j] = dst[j];
            t->staged_kn[j] = kn[j];
        }
        EH_CHECK(hipMemcpyAsync(to, x->dev + ((size_t)e * x->n_cap + j) * cap, sizeof(edgehip_keyline) * kn[j], hipMemcpyDeviceToHost, x->stream));
    }
    EH_CHECK(hipEventRecord(x->ev_done[e], x->stream));
    t->fetched = true;
    return 0;
}

int edgehip_export_wait(edgehip_ctx *c, int ticket) {
    EH_ENTER(c);
    int e = 0;
    auto *t = export_ticket(c, ticket, e);
    if (!t) { set_error("export_wait: unknown ticket"); return EDGEHIP_ERR_ARG; }
    if (t->fetched) {
        EH_CHECK(hipEventSynchronize(c->kl_export->ev_done[e]));
        for (int j = 0; j < t->n; j++)
            if (t->staged_dst[j]) memcpy(t->staged_dst[j], c->kl_export->host[e] + (size_t)j * c->plan.cap, sizeof(edgehip_keyline) * t->staged_kn[j]);
    }
    t->id = -1;          // (a ticket that was never fetched is simply dropped: its staging entry is free again)
    t->fetched = false;
    return 0;
}

int edgehip_upload_keylines(edgehip_ctx *c, int seq, int slot, const edgehip_keyline *kl, int32_t kn, const int32_t *mask, float retuned) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (int e = check_slot(c, slot)) return e;
    if (int e = rot_materialize_enqueue(c, slot)) return e;   // the other sequences of the slot keep their (turned) KeyLines
    if (!kl || kn < 0 || kn > c->plan.cap) { set_error("upload_keylines: kn exceeds capacity"); return EDGEHIP_ERR_ARG; }
    if (int e = sync_all(c)) return e;
    const KlSoA &k = klof(c, slot, seq);
    std::vector<int32_t> p_inx(kn), m_id(kn), m_id_f(kn), m_id_kf(kn), m_num(kn), p_id(kn), n_id(kn);
    std::vector<float2> m_m(kn), u_m(kn), c_p(kn), p_m(kn), p_m_0(kn), m_m0(kn);
    std::vector<float> n_m(kn);
    std::vector<double> rho(kn), s_rho(kn), rho_nr(kn), s_rho_nr(kn), rho0(kn), s_rho0(kn), n_m0(kn);
    std::vector<MatchRec> rec(kn);
    std::vector<float4> grec(kn);
    bool consistent = true;   // u_m == m_m / sqrt(m_m . m_m) in float arithmetic, for every KeyLine
    for (int i = 0; i < kn; i++) {
        const edgehip_keyline &o = kl[i];
        p_inx[i] = o.p_inx;
        m_m[i] = make_float2(o.m_m[0], o.m_m[1]); u_m[i] = make_float2(o.u_m[0], o.u_m[1]);
        n_m[i] = o.n_m; c_p[i] = make_float2(o.c_p[0], o.c_p[1]);
        rho[i] = o.rho; s_rho[i] = o.s_rho; rho_nr[i] = o.rho_nr; s_rho_nr[i] = o.s_rho_nr; rho0[i] = o.rho0; s_rho0[i] = o.s_rho0;
        p_m[i] = make_float2(o.p_m[0], o.p_m[1]); p_m_0[i] = make_float2(o.p_m_0[0], o.p_m_0[1]);
        m_id[i] = o.m_id; m_id_f[i] = o.m_id_f; m_id_kf[i] = o.m_id_kf; m_num[i] = o.m_num;
        m_m0[i] = make_float2(o.m_m0[0], o.m_m0[1]); n_m0[i] = o.n_m0; p_id[i] = o.p_id; n_id[i] = o.n_id;
        MatchRec r; r.c_px = o.c_p[0]; r.c_py = o.c_p[1]; r.u_mx = o.u_m[0]; r.u_my = o.u_m[1];
        r.m_mx = o.m_m[0]; r.m_my = o.m_m[1]; r.n_m = o.n_m; r.pad = 0.f;
        rec[i] = r;
        grec[i] = make_float4(o.c_p[0], o.c_p[1], o.m_m[0], o.m_m[1]);
        {
            volatile float n2 = o.m_m[0] * o.m_m[0];   // volatile: no contraction, no excess precision
            volatile float t2 = o.m_m[1] * o.m_m[1];
            n2 = n2 + t2;
            const float nn = sqrtf(n2);
            volatile float ux = o.m_m[0] / nn, uy = o.m_m[1] / nn;
            if (!(ux == o.u_m[0] && uy == o.u_m[1])) consistent = false;
        }
    }
    int e = 0;
    e |= h2d(c, k.p_inx, p_inx); e |= h2d(c, k.m_id, m_id); e |= h2d(c, k.m_id_f, m_id_f); e |= h2d(c, k.m_id_kf, m_id_kf);
    e |= h2d(c, k.m_num, m_num); e |= h2d(c, k.p_id, p_id); e |= h2d(c, k.n_id, n_id);
    e |= h2d(c, k.m_m, m_m); e |= h2d(c, k.u_m, u_m); e |= h2d(c, k.c_p, c_p); e |= h2d(c, k.p_m, p_m);
    e |= h2d(c, k.p_m_0, p_m_0); e |= h2d(c, k.m_m0, m_m0); e |= h2d(c, k.n_m, n_m);
    e |= h2d(c, k.rho, rho); e |= h2d(c, k.s_rho, s_rho); e |= h2d(c, k.rho_nr, rho_nr); e |= h2d(c, k.s_rho_nr, s_rho_nr);
    e |= h2d(c, k.rho0, rho0); e |= h2d(c, k.s_rho0, s_rho0); e |= h2d(c, k.n_m0, n_m0); e |= h2d(c, k.rec, rec); e |= h2d(c, k.grec, grec);
    std::vector<int32_t> st_id;          // function scope: the asynchronous copies out of them end at the sync below
    std::vector<double> st_rho, st_srho;
    if (k.stereo_m_id) {
        st_id.resize(kn); st_rho.resize(kn); st_srho.resize(kn);
        for (int i = 0; i < kn; i++) { st_id[i] = kl[i].stereo_m_id; st_rho[i] = kl[i].stereo_rho; st_srho[i] = kl[i].stereo_s_rho; }
        e |= h2d(c, k.stereo_m_id, st_id); e |= h2d(c, k.stereo_rho, st_rho); e |= h2d(c, k.stereo_s_rho, st_srho);
    }
    if (e) return EDGEHIP_ERR_DEVICE;
    // the 16-byte gather records stand for the slot only if they do for every sequence in it
    {
        const bool was = c->grec_ok[slot];
        c->grec_ok[slot] = consistent && (c->plan.nseq == 1 || c->grec_ok[slot]);
        if (was != c->grec_ok[slot]) drop_frame_graphs(c);   // captured frames chose their TryVelRot variant by this flag
    }
    EH_CHECK(hipMemcpyAsync(c->kn_slot + (size_t)slot * c->plan.nseq + seq, &kn, 4, hipMemcpyHostToDevice, c->stream));
    if (mask) EH_CHECK(hipMemcpyAsync(maskof(c, slot) + (size_t)seq * c->plan.n, mask, sizeof(int32_t) * c->plan.n, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipMemcpyAsync(c->retuned_slot + (size_t)slot * c->plan.nseq + seq, &retuned, 4, hipMemcpyHostToDevice, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_download_plane(edgehip_ctx *c, int seq, int which, float *out) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (!c->planes) { set_error("download_plane: context was created without debug_planes"); return EDGEHIP_ERR_STATE; }
    if (which < 0 || which > 4 || !out) return EDGEHIP_ERR_ARG;
    if (int e = sync_all(c)) return e;
    const size_t n = c->plan.n;
    EH_CHECK(hipMemcpyAsync(out, c->planes + ((size_t)which * c->plan.nseq + seq) * n, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int edgehip_download_field(edgehip_ctx *c, int seq, int32_t *out) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (!out) return EDGEHIP_ERR_ARG;
    const size_t n = c->plan.n;
    const size_t fs = c->plan.fstride;
    if (!c->field32_valid) {
        // the product path keeps only the KeyLine-index plane the tracker gathers: ikl from it, dist reported as -1
        const size_t fs16 = c->plan.f16stride;
        std::vector<uint16_t> f16(fs16);
        EH_CHECK(hipMemcpyAsync(f16.data(), c->field16 + (size_t)seq * fs16, 2 * fs16, hipMemcpyDeviceToHost, c->stream));
        EH_CHECK(hipStreamSynchronize(c->stream));
        for (int y = 0; y < c->plan.h; y++)
            for (int x = 0; x < c->plan.w; x++) {
                const uint16_t v = f16[edgehip::field16_index(x, y, c->plan.f16tx)];
                const size_t i = (size_t)y * c->plan.w + x;
                out[2 * i] = v ? -1 : 0;
                out[2 * i + 1] = v ? (int32_t)v - 1 : -1;
            }
        return 0;
    }
    std::vector<uint32_t> f(fs);
    EH_CHECK(hipMemcpyAsync(f.data(), c->field + (size_t)seq * fs, 4 * fs, hipMemcpyDeviceToHost, c->stream));
    EH_CHECK(hipStreamSynchronize(c->stream));
    for (int y = 0; y < c->plan.h; y++)
        for (int x = 0; x < c->plan.w; x++) {   // tiled device layout -> row-major (dist, ikl) pairs
            const uint32_t v = f[edgehip::field_index(x, y, c->plan.ftx)];
            const size_t i = (size_t)y * c->plan.w + x;
            if (v == 0xFFFFFFFFu) { out[2 * i] = 0; out[2 * i + 1] = -1; }
            else { out[2 * i] = (int32_t)(v >> 16); out[2 * i + 1] = (int32_t)(0xFFFFu - (v & 0xFFFFu)); }
        }
    (void)n;
    return 0;
}

int edgehip_build_undistort_map(const edgehip_params *params, int32_t *inx, int32_t *iw) {
    if (!params || !inx || !iw || params->w < 1 || params->h < 1) return EDGEHIP_ERR_ARG;
    std::vector<int32_t> base, rinx, riw;
    std::vector<uint32_t> w4;
    build_undistort_map(*params, base, w4, &rinx, &riw);
    memcpy(inx, rinx.data(), sizeof(int32_t) * rinx.size());
    memcpy(iw, riw.data(), sizeof(int32_t) * riw.size());
    return 0;
}

int edgehip_download_undistorted(edgehip_ctx *c, int seq, int slot, uint8_t *rgb24) {
    EH_ENTER(c);
    if (int e = check_seq(c, seq)) return e;
    if (int e = check_slot(c, slot)) return e;
    if (!rgb24) return EDGEHIP_ERR_ARG;
    if (!c->p.use_undistort) { set_error("download_undistorted: context was created without use_undistort"); return EDGEHIP_ERR_STATE; }
    uint8_t *tmp = reinterpret_cast<uint8_t *>(c->ii);  // stage-A scratch: free between frames
    if (int e = sync_all(c)) return e;
    if (int e = undistort_frame_enqueue(c, seq, slot, tmp)) return e;
    EH_CHECK(hipMemcpyAsync(rgb24, tmp, (size_t)c->plan.n * 3, hipMemcpyDeviceToHost, c->stream_a));
    EH_CHECK(hipStreamSynchronize(c->stream_a));
    return 0;
}

// ---- profiler -------------------------------------------------------------------------------------------
int edgehip_profile_enable(edgehip_ctx *c, int on) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    c->prof->on = on != 0;
    return 0;
}
int edgehip_profile_select(edgehip_ctx *c, uint64_t mask) {
    EH_ENTER(c);
    if (!c) return EDGEHIP_ERR_ARG;
    c->prof->mask = mask;
    return 0;
}
int edgehip_profile_count(void) { return PROF_COUNT; }
int edgehip_experiments(void) {
#ifdef EDGEHIP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
const char *edgehip_profile_name(int i) { return (i >= 0 && i < PROF_COUNT) ? kProfNames[i] : ""; }
int edgehip_profile_read(edgehip_ctx *c, double *ms, int64_t *calls) {
    EH_ENTER(c);
    if (!c || !ms || !calls) return EDGEHIP_ERR_ARG;
    if (int e = sync_all(c)) return e;
    Profiler *p = c->prof;
    for (auto &r : p->pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { p->ms[r.id] += t; p->calls[r.id]++; }
        p->pool.push_back(r.a);
        p->pool.push_back(r.b);
    }
    p->pending.clear();
    for (int i = 0; i < PROF_COUNT; i++) { ms[i] = p->ms[i]; calls[i] = p->calls[i]; p->ms[i] = 0; p->calls[i] = 0; }
    return 0;
}

}  // extern "C"
