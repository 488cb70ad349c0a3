// ctx.h — internal layout of an edgehip context (device-resident SoA) and small host helpers.
//
// HBM layout (per context; B = nseq, S = nslots, N = w*h, CAP = KeyLine capacity = max_points):
//   rgb      [S][B][N*3] u8      input frames (one per ring slot, like PipeBuffer::imgc)
//   ii       [4][B][N]  f32      integral-image scratch of the box-filter chain (stage A only)
//   planes   [5][B][N]  f32      img0,img1,dog,dx,dy — only when params.debug_planes
//   mask     [S][B][N]  i32      img_mask_kl
//   field16  [B][FS16]  u16      tracker auxiliary image as the tracker reads it: KeyLine index + 1, 0 = empty;
//                               8x4-pixel tiles (field16_index), FS16 = ceil(w/8)*ceil(h/4)*32
//   field    [B][FS]    u32      the same with distances, packed (dist<<16 | 0xFFFF-ikl), 0xFFFFFFFF = empty, 4x4-pixel
//                               tiles (field_index), FS = ceil(w/4)*ceil(h/4)*16 — written only for download_field
//                               (params.debug_planes) and by the alternative builders
//   KeyLines [S][B][CAP] per field (structure of arrays, see KlSoA)
//   stage buffers for the raster-order compaction, LM scratch, residual buffers, per-sequence state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>
#include <mutex>
#include <atomic>

#include "edgehip.h"

namespace edgehip {

constexpr int kMaxBoxes = 3;        // bf_num of sspace (rebvo.cpp:299 passes 3)
constexpr int kBandRows = 12;       // rows per detect block (+4 halo rows in LDS)
constexpr int kDetWaves = 16;       // waves per detect block; a (band, wave) strip is the raster-order compaction granule
constexpr int kDivLutMax = 256;     // reciprocal-count LUT entries (box width up to 15)
constexpr int kTvrThreads = 256;    // threads per TryVelRot block (more KeyLines per block — bigger blocks or several passes — measured slower: the kernel lives on occupancy to hide its two dependent gathers)
constexpr int kTvrPasses = 1;       // KeyLines per thread
constexpr int kTvrBlock = kTvrThreads * kTvrPasses;   // KeyLines per block = granule of the residual carries
constexpr int kNumSums = 28;        // 21 JtJ + 6 JtF + score
constexpr int kRefRing = 8;         // CBUFSIZE of the reference (frame ring length that FrameCount semantics follow)
constexpr int kResidBufs = 3;       // Res0, Res1, Rest (global_tracker.cpp:611)

// "inherit the last valid residual of the previous blocks" marker in residual buffers (see stage_b.hip)
__host__ __device__ inline uint64_t resid_carry_bits() { return 0x7FF8C0DEC0DEC0DEull; }

// One record per KeyLine gathered at random by TryVelRot / search_match: 32 B, one 2x16-B load.
struct __attribute__((aligned(16))) MatchRec {
    float c_px, c_py, u_mx, u_my, m_mx, m_my, n_m, pad;
};

// Structure-of-arrays KeyLine storage of one (slot, sequence): every pointer addresses CAP elements.
struct KlSoA {
    int32_t *p_inx;
    float2 *m_m, *u_m, *c_p, *p_m, *p_m_0, *m_m0;
    float *n_m;
    double *rho, *s_rho, *rho_nr, *s_rho_nr, *rho0, *s_rho0, *n_m0;
    int32_t *m_id, *m_id_f, *m_id_kf, *m_num, *p_id, *n_id;
    MatchRec *rec;  // (c_p, u_m, m_m, n_m) packed for gathers; kept in sync with the fields above
    // (c_p, m_m): the 16 bytes TryVelRot needs of the KeyLine it matched, when u_m == m_m / sqrt(m_m . m_m) holds bit for
    // bit (every freshly detected KeyLine: edge_finder.cpp:166-200) and can be recomputed instead of fetched.  Valid
    // only while edgehip_ctx::grec_ok[slot] (cleared by rotate_keylines, which turns m_m but not u_m).
    float4 *grec;
    // stereo fields (null unless params.stereo_available)
    int32_t *stereo_m_id;
    double *stereo_rho, *stereo_s_rho;
};

// Device-side per-sequence state (superset of edgehip_seq_state).
struct SeqDev {
    edgehip_seq_state pub;
    // stage A scratch
    int32_t kn_new;            // KeyLines of the slot just detected
    int32_t band_trunc;        // band index where kl_max truncation happened (-1: none)
    float nm_max, nm_min;      // reEstimateThresh extremes
    double tresh_used;         // threshold used by the running detect
    // minimiser scratch (global_tracker::Minimizer_RV locals)
    double X[6], Xnew[6], Xt[6], h[6];
    double JtJ[36], JtF[6], JtJnew[36], JtFnew[6];
    double F, Fnew, F0, Ft, F0t, u, v, ut, vt, gain;
    int32_t eff_steps, eff_steps_t, res_cur, res_new, res_t, lm_phase;
    int32_t kn_old, pad1;
    double s_rho_min_eval;     // s_rho threshold of the running evaluation
    double Rt[9], Vt[3], RM[4];  // rotation / translation / z-rotation of the running evaluation
    int32_t nmatch_tmp, kf_tmp;
    // The zero-init chain of Minimizer_RV's double initialisation (global_tracker.cpp:649-692) while it runs beside the
    // prior-init chain (:698-738) in the same launches (k_try_velrot2 / k_lm_step2): the two chains share nothing but their
    // inputs, so each evaluation launch serves both and this is the second chain's copy of the locals above.
    double zX[6], zXnew[6], zh[6];
    double zJtJ[36], zJtF[6], zJtJnew[36], zJtFnew[6];
    double zF, zFnew, zF0, zu, zv, zgain;
    double zRt[9], zVt[3], zRM[4];
    int32_t z_eff_steps, z_pad;
    // 3-DoF minimiser scratch (global_tracker::Minimizer_V locals)
    double mv_V[3], mv_Vnew[3], mv_h[3], mv_JtJ[9], mv_JtF[3], mv_JtJnew[9], mv_JtFnew[3], mv_RVel[9];
    double mv_F, mv_Fnew, mv_u, mv_v, mv_s_rho_min;
    float mv_min_mod; int32_t mv_pad;
    // per-frame glue (SecondThread locals)
    int32_t skip_match, skip_map;   // NaN estimate / too few matches: later stages are no-ops
    double t_cur;
    double V_track[3], W_track[3], PV_track[9], PW_track[9];  // tracker output before any reset
};

// Detector state of one sequence: the locals of FirstThr (rebvo_first_t.cpp:92-94) plus stage-A scratch.  Kept apart
// from SeqDev because stage A of frame k+1 runs on its own stream while stages B/C of frame k still use SeqDev.
struct SeqA {
    double tresh;              // detector threshold (P-controller state)
    double tresh_used;         // threshold used by the running detect
    int32_t l_kl_num;          // KeyLines on the last edge map (P-controller input)
    int32_t kn_new;            // KeyLines of the slot being detected
    int32_t band_trunc, pad0;
    float nm_max, nm_min;      // reEstimateThresh extremes
    float retuned, pad1;       // edge_finder::reTunedThresh of the newest slot
};

struct DevicePlan {  // everything a kernel needs that is constant for the context
    int w, h, n, cap, nseq, nslots;
    int ftx;                        // field layout: 4x4-pixel tiles per tile row = ceil(w/4)
    size_t fstride;                 // field elements per sequence = ftx * ceil(h/4) * 16
    int f16tx;                      // index plane: 8x4-pixel tiles per tile row = ceil(w/8)
    size_t f16stride;               // index-plane elements per sequence = f16tx * ceil(h/4) * 32
    int box[2][kMaxBoxes];          // box widths of filter0 / filter1
    float box_a[2][kMaxBoxes];      // (float)(1.0/(d*d))
    float ppx, ppy, zfx, zfy;       // cam_model keeps these as float (cam_model.h:51-52)
    double zfm;                     // (double)((zfx+zfy)/2) computed in float (cam_model.h:57)
};

// (int)double as the reference's x86-64 build converts it (cvttsd2si): out-of-range values and NaN give INT_MIN, where the
// GPU's v_cvt_i32_f64 saturates to INT_MAX / 0.  Matters wherever the reference turns an unbounded double into a loop count
// or an index: search_match's t_steps (edge_tracker.cpp:211-213) becomes negative there — no iterations — and 2^31
// iterations with a saturating conversion.
__device__ __forceinline__ int x86_cvttsd2si(double f) {
    if (!(f > -2147483649.0 && f < 2147483648.0)) return (int)0x80000000;
    return (int)f;
}

// The same conversion where its result only feeds an in-image test (1 <= x < w - 1): the bare v_cvt_i32_f64, which saturates (INT_MAX /
// INT_MIN, 0 for a NaN) where cvttsd2si answers INT_MIN — every one of those values fails the test exactly as INT_MIN does, so the
// range check in front of the conversion (two fp64 compares and a select per coordinate) buys nothing there.
__device__ __forceinline__ int cvt_trunc_sat_i32(double f) {
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// fp64 quotients whose operands sit in the middle of the exponent range (the evaluations' 1 / rho, 1 / z, k / |r|, 1 / q_rho in stage_b.hip; the plane
// fit's sub-pixel offsets in stage_a_fused.hip): the compiler's division is v_div_scale x 2, v_rcp_f64, two
// Newton steps, quotient, remainder, v_div_fmas, v_div_fixup — 11 instructions and the vcc traffic of the scaling.  The two v_div_scale and
// v_div_fmas only move operands whose exponents sit near the ends of the range (|exponent| beyond ~ 900) back to the middle; for every other
// pair they are the identity, and what remains is the same chain of fma's on the same values: the same bits.  No operand here comes near
// those ends (inverse depths, camera-frame depths, residuals in pixels, q_rho >= 1), so the scaling is dropped; v_div_fixup stays for the
// zeros, infinities and NaNs (a KeyLine exactly on the camera plane still gets its IEEE infinity).
#ifndef EDGEHIP_FASTDIV
#define EDGEHIP_FASTDIV 1
#endif
// 1 / b as the division sequence has it before its last correction (v_rcp_f64 and two Newton steps): what div_rn takes as `rb`
__device__ __forceinline__ double rcp_nr(const double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double div_mid(const double a, const double b) {   // a / b, operands in the middle of the exponent range
#if EDGEHIP_FASTDIV
    const double r = rcp_nr(b);
    const double q = a * r;
    const double e = __builtin_fma(-b, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(e, r, q), b, a);
#else
    return a / b;
#endif
}
// two quotients by the same divisor: its reciprocal and Newton steps once (the plane fit's xs, ys = -t0 t2 / den, -t1 t2 / den)
__device__ __forceinline__ void div2_mid(const double a0, const double a1, const double b, double &q0, double &q1) {
#if EDGEHIP_FASTDIV
    const double r = rcp_nr(b);
    double q = a0 * r;
    q0 = __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-b, q, a0), r, q), b, a0);
    q = a1 * r;
    q1 = __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-b, q, a1), r, q), b, a1);
#else
    q0 = a0 / b; q1 = a1 / b;
#endif
}
// several quotients by one divisor: MidDivisor d(b); d(a0), d(a1), ... — each is a / b to the bit (operands in the middle of the range)
struct MidDivisor {
    double b, r;
    __device__ __forceinline__ explicit MidDivisor(const double b_) : b(b_), r(EDGEHIP_FASTDIV ? rcp_nr(b_) : 0.0) {}
    __device__ __forceinline__ double operator()(const double a) const {
#if EDGEHIP_FASTDIV
        const double q = a * r;
        return __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-b, q, a), r, q), b, a);
#else
        return a / b;
#endif
    }
};
__device__ __forceinline__ double inv_mid(const double b) {   // 1 / b, likewise (the quotient estimate 1 * r is r itself)
#if EDGEHIP_FASTDIV
    const double r = rcp_nr(b);
    const double e = __builtin_fma(-b, r, 1.0);
    return __builtin_amdgcn_div_fixup(__builtin_fma(e, r, r), b, 1.0);
#else
    return 1.0 / b;
#endif
}

// a / b rounded as IEEE division rounds it, given rb = RN(1 / b): the closing step of the division sequence the compiler itself
// emits (quotient estimate, exact remainder by fma, correction by fma), without its scaling (no operand here is near the exponent
// range's ends).  Against a / b on 4e8 random and adversarial pairs (significands near 1, near 2, short): no difference.
__device__ __forceinline__ double div_rn(const double a, const double b, const double rb) {
    const double q = a * rb;
    const double e = __builtin_fma(-q, b, a);
    return __builtin_fma(e, rb, q);
}
// sqrt(x) for x >= 1 (q_rho^2 = (s_rho qvel)^2 + 1): the compiler's own sequence (v_rsq_f64, one coupled Newton step on the root and its
// half reciprocal, two corrections from the exact remainder) without what it wraps around it for arguments below 2^-767 (a compare, two
// selects, two v_ldexp_f64 by 0 here) — same values through the same fma's; +inf is the one special value left above 1.
__device__ __forceinline__ double sqrt_ge1(const double x) {
#if EDGEHIP_FASTDIV
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_isinf(x) ? x : g;
#else
    return sqrt(x);
#endif
}
// sqrtf(x) for a float in the middle of the range (|m_m|^2 of a KeyLine that passed the detector's threshold): the compiler's sequence — v_sqrt_f32
// (1 ulp), then the neighbour below / above if the exact remainder says so — without the scaling it wraps around it for x < 2^-96 and without the
// class test behind it (zero and +inf come out of the sequence as themselves: the remainders are NaN or -0 there and select nothing).
__device__ __forceinline__ float sqrtf_mid(const float x) {
#if EDGEHIP_FASTDIV
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.f ? s_dn : s;
    r = r_up > 0.f ? s_up : r;
    return r;
#else
    return sqrtf(x);
#endif
}
// The matched KeyLine's unit gradient u_m = m_m / |m_m| (float, as the detector computes it): two float quotients by the same divisor.
// The compiler's float division is the double one's shape (v_div_scale x 2, v_rcp_f32, one Newton step, quotient, two corrections,
// v_div_fmas, v_div_fixup); the divisor's part — the reciprocal and its Newton step — is shared here and the scaling dropped as above
// (|m_m| is a gradient modulus that passed the detector's threshold: the middle of the range).
__device__ __forceinline__ void div2_mid_f32(const float n0, const float n1, const float d, float &q0, float &q1) {
#if EDGEHIP_FASTDIV
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
    float q = n0 * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n0), r, q);
    q0 = __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-d, q, n0), r, q), d, n0);
    q = n1 * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n1), r, q);
    q1 = __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-d, q, n1), r, q), d, n1);
#else
    q0 = n0 / d; q1 = n1 / d;
#endif
}
__device__ __forceinline__ double rcp_for_div_rn(const double b) {   // div_rn's `rb`: the sequence's own reciprocal is enough (and is what a / b uses)
#if EDGEHIP_FASTDIV
    return rcp_nr(b);
#else
    return 1.0 / b;
#endif
}

// round() as Image::GetIndexRC uses it (half away from zero) for pixel coordinates, in 3 instructions.
// v_cvt_rpi_i32_f32 converts with "round to nearest, ties towards +infinity", evaluated exactly (not as a float
// addition of 0.5): for v >= 0 that IS half-away-from-zero; for v < 0 the two differ only on exact ties, where both
// results are negative (an out-of-image coordinate either way) except v == -0.5 -> 0 instead of -1, fixed up below.
// tools/experiments/cvt_rpi_check.hip compares it with round() over all 2^32 float bit patterns with |v| < 4096 on the
// GPU: -0.5 is the only input whose result could select a different pixel.
__device__ __forceinline__ int round_ties_up_i(float v) {   // the bare conversion: differs from round() only as described above
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int round_half_away_i(float v) { return v == -0.5f ? -1 : round_ties_up_i(v); }

// The tracker field is stored in 4x4-pixel tiles of 64 B (tile-row-major, pixels row-major inside a tile): the
// TryVelRot gather touches one pixel per KeyLine, and the KeyLines of one edge sit on consecutive rows, so with a
// row-major plane every gather pulls its own 64-B line while here up to four rows of an edge share one.
__host__ __device__ __forceinline__ size_t field_index(int x, int y, int ftx) {
    return (((size_t)(y >> 2) * (size_t)ftx + (size_t)(x >> 2)) << 4) | (size_t)(((y & 3) << 2) | (x & 3));
}

// The KeyLine-index plane of the field alone, 2 bytes per pixel (0 = empty, ikl + 1 otherwise) in 8x4-pixel tiles of 64 B:
// what TryVelRot / TryVel gather (they never use the distance).
__host__ __device__ __forceinline__ size_t field16_index(int x, int y, int f16tx) {
    return (((size_t)(y >> 2) * (size_t)f16tx + (size_t)(x >> 3)) << 5) | (size_t)(((y & 3) << 3) | (x & 7));
}

// Streaming ("nontemporal") stores for arrays a kernel writes once, front to back, and that exceed the caches many times over before anyone reads them.
// Build switches EDGEHIP_NT_* choose per kernel (A/Bs in profiles/r06_nontemporal_stores_ab.txt).
template <class T> __device__ __forceinline__ void st_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(float2 *p, float2 v) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(f2v{v.x, v.y}, reinterpret_cast<f2v *>(p));
}
__device__ __forceinline__ void st_stream(int2 *p, int2 v) {
    typedef int i2v __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(i2v{v.x, v.y}, reinterpret_cast<i2v *>(p));
}

// Loads through pointers that come out of a KlSoA record (or any other device-memory table of pointers) are GENERIC to the compiler: it
// emits FLAT loads, which take the address-space check and count on the LDS counter as well as on the vector-memory one (a wait for an LDS
// or scalar-load answer behind them waits for them too).  Everything a KlSoA points to is global memory: ldg() says so.
// EDGEHIP_GLOBAL_LD=0 keeps the generic form (A/B).
#ifndef EDGEHIP_GLOBAL_LD
#define EDGEHIP_GLOBAL_LD 1
#endif
#ifdef __HIPCC__
template <class T> __device__ __forceinline__ T ldg(const T *p, size_t i) {
#if EDGEHIP_GLOBAL_LD
    typedef const T __attribute__((address_space(1))) G;
    return ((G *)p)[i];
#else
    return p[i];
#endif
}
__device__ __forceinline__ float2 ldg(const float2 *p, size_t i) {
#if EDGEHIP_GLOBAL_LD
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef const f2v __attribute__((address_space(1))) G;
    const f2v v = ((G *)p)[i];
    return make_float2(v.x, v.y);
#else
    return p[i];
#endif
}
__device__ __forceinline__ float4 ldg(const float4 *p, size_t i) {
#if EDGEHIP_GLOBAL_LD
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const f4v __attribute__((address_space(1))) G;
    const f4v v = ((G *)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return p[i];
#endif
}
#endif

#ifdef __HIPCC__
// ... and the stores (plain and streaming) through such pointers
template <class T> __device__ __forceinline__ void stg(T *p, size_t i, T v) {
#if EDGEHIP_GLOBAL_LD
    typedef T __attribute__((address_space(1))) G;
    ((G *)p)[i] = v;
#else
    p[i] = v;
#endif
}
__device__ __forceinline__ void stg(float2 *p, size_t i, float2 v) {
#if EDGEHIP_GLOBAL_LD
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef f2v __attribute__((address_space(1))) G;
    ((G *)p)[i] = f2v{v.x, v.y};
#else
    p[i] = v;
#endif
}
template <class T> __device__ __forceinline__ void stg_stream(T *p, size_t i, T v) {
#if EDGEHIP_GLOBAL_LD
    typedef T __attribute__((address_space(1))) G;
    __builtin_nontemporal_store(v, (G *)p + i);
#else
    st_stream(p + i, v);
#endif
}
__device__ __forceinline__ void stg_stream(float2 *p, size_t i, float2 v) {
#if EDGEHIP_GLOBAL_LD
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef f2v __attribute__((address_space(1))) G;
    __builtin_nontemporal_store(f2v{v.x, v.y}, (G *)p + i);
#else
    st_stream(p + i, v);
#endif
}
#endif

struct Profiler;

}  // namespace edgehip

struct edgehip_ctx {
    edgehip_params p;
    edgehip::DevicePlan plan;
    int device;
    hipStream_t stream;    // stages B and C, state exchange
    hipStream_t stream_a;  // uploads and stage A: frame k+1 is detected while frame k is still being tracked
    hipEvent_t ev_a[4];    // [slot] stage A of the frame in this slot has finished
    hipEvent_t ev_use[4];  // [slot] the last B/C work that read this slot has finished
    hipEvent_t ev_tmp;     // ordering of the stage-level entry points
    hipEvent_t ev_stage, ev_stage8;   // last copy out of the pageable-upload staging buffers (RGB24, 8-bit)
    bool stage_busy = false, stage8_busy = false;
    hipStream_t stream_up; // uploads of page-locked frames: they overlap stage A as well as B/C of the frames before
    hipEvent_t ev_up[4];   // [slot] the last upload into this slot on stream_up has finished
    bool up_valid[4];
    int slot_ring[4];      // [slot] ev_ring entry of the last frame processed in this slot (-1: none)
    bool no_grec;          // EDGEHIP_NO_GREC=1: always gather the 32-byte record (A/B measurements)
    bool grec_ok[4];       // [slot] KlSoA::grec describes the slot's KeyLines (all sequences)
    bool rec_stale[4];     // [slot] rotate_keylines has turned m_m since KlSoA::rec was written: rec.m_m is behind (rec_refresh_enqueue brings
                           // it up to date for the rare reader — a rotated slot used as the tracker's field side, a key frame, a stereo pair)
    bool rig_a_valid = false;   // ev_a[pair slot] was recorded behind the pair image's stage A by the whole-frame driver (not while capturing): the NEXT pair
                                // frame's copy waits for that, not for the end of the frame (nothing reads a slot's frame storage after its own stage A)
    bool capturing = false;     // edgehip_process_frame is capturing a frame graph (events recorded now are graph edges, not events)
    bool a_api_valid[4];   // [slot] ev_a was recorded by the stage-level edgehip_stage_a (an upload must wait for it)
    hipEvent_t ev_ring[8]; // [frame % 8] the frame that used this entry of the pinned time-stamp / frame-index rings is done
    bool ring_valid[8];
    bool use_valid[4];
    int overlap;           // 1: stage A of frame k+1 may run under stages B/C of frame k (EDGEHIP_OVERLAP=1); 0: one after the other
    bool lds_optin_level = false, lds_optin_detect = false, lds_optin_fused = false;   // > 64 KB dynamic LDS opted in for this context's device
    bool lds_optin_rescale = false;   // k_rescale<512, 12, 4>: 128 KB of dynamic LDS
    int frame_slot;        // ring position of the newest slot (-1 before the first frame)
    int frames_seen;
    // device buffers
    uint8_t *rgb;          // [S][B][N*3]
    float *ii;             // [4][B][N]
    float *planes;         // [5][B][N] or null
    int32_t *mask;         // [S][B][N]
    uint32_t *field;       // [B][plan.fstride], tiled (field_index): {dist, ikl}, for download_field / the other builders
    uint16_t *field16;     // [B][plan.f16stride], KeyLine-index plane (field16_index)
    bool field32_valid;    // the {dist, ikl} field was written by the last build_field (debug_planes, or the other builders)
    int32_t *und_base;     // [N] undistortion map: pixel index of the p00 tap (may lie outside the image), or null
    uint4 *und_iw;         // [N] 16.16 integer weights of taps p00,p01,p10,p11 (0 = tap not valid)
    float *div_lut;        // [kDivLutMax] (float)(1.0/count)
    double *pinv;          // [3*25] plane-fit pseudo inverse
    void *kl_arena;        // one allocation carved into KlSoA arrays
    std::vector<edgehip::KlSoA> kl;   // [S*B] host copies of the carved pointers
    edgehip::KlSoA *kl_dev;           // [S*B] same, on device
    edgehip::SeqDev *seq;             // [B]
    edgehip::SeqA *seqa;              // [B] detector state (stage-A stream)
    double *tresh_slot;               // [S][B] detector threshold each slot was detected with
    // global_tracker::FrameCount lives in the reference's PipeBuffer slot objects, of which there are CBUFSIZE=8
    // (include/rebvo/rebvo.h:51, src/rebvo/rebvo.cpp:297-312): the counter a frame sees depends on that ring
    // length, not on ours.  [max(S,8)][B]; process_frame indexes it by (frame number % 8), the stage-level
    // entry points by the slot they are given.
    uint32_t *framecount;
    int fc_rows;                      // max(S, kRefRing)
    int fc_index;                     // row used by the running minimisation
    int32_t *kn_slot;                 // [S][B] edge_finder::kn per slot
    float *retuned_slot;              // [S][B] edge_finder::reTunedThresh per slot
    // stage A compaction staging
    int32_t *band_cnt;     // [B][nbands]
    int32_t *band_off;     // [B][nbands]
    void *band_stage;      // [B][nbands][band_cap] candidate records
    int nbands, band_cap;
    int band_rows;         // image rows per k_detect band (kBandRows, fewer for images too wide for 12-row planes in LDS)
    int32_t *histo;        // [B][256] scratch histograms
    // stage B scratch
    double *P0;            // [B][3][CAP]
    double *resid;         // [kResidBufs][B][CAP]
    double *resid_carry;   // [kResidBufs][B][nblk_tvr] last valid residual per block
    double *partials;      // [B][nblk_tvr][kNumSums]
    double *block_last;    // [B][nblk_tvr] last valid residual of each block of the running evaluation
    // Whole-frame HIP graphs (edgehip_process_frame): a frame is ~75 launches whose arguments repeat with the ring slot
    // and the FrameCount row, i.e. with period lcm(ring, 8); each variant is captured the first time it comes up and
    // replayed afterwards.  Pays when the batch is small and the frame is launch-bound (a single live camera).
    bool use_graph;
    std::map<int, hipGraphExec_t> frame_graphs;
    struct StereoRig {                // edgehip_set_stereo_rig: the pair camera of edgehip_process_frame
        bool enabled = false;
        int slot_pair = -1;
        double t[3], R[9], max_radius;
    } rig;
    int ring_slots;                   // slots the frame ring cycles through (nslots, or nslots - 1 with a stereo rig)
    // base == nullptr: the slot's own storage.  grey8: the frames are 8-bit mono, 1 B per pixel (edgehip_upload_grey8* /
    // edgehip_bind_grey8_indexed) — in the slot's grey8 buffer or, bound, in a pool of grey8 frames.
    struct SlotSrc { const uint8_t *base = nullptr; std::vector<int32_t> host_idx; bool grey8 = false;
                     const int32_t *idx_row = nullptr; int idx_ring = 0; };   // idx_row: the page-locked row of the binding (stage A reads it in place)
    uint8_t *grey8 = nullptr;        // [S][B][N] 8-bit frames of the slots (allocated by the first grey8 upload)
    uint8_t *pinned_grey8 = nullptr; // [B][N] staging of edgehip_upload_grey8 (pageable input)
    std::vector<SlotSrc> slot_src;   // per ring slot: where stage A reads its frames from
    struct SlotCam { float ppx, ppy; double zfm; };
    std::vector<SlotCam> slot_cam;   // per ring slot: principal point stage A uses, focal length of that camera (stereo pair slot)
    int field_radius;      // radius of the last build_field (global_tracker::max_r)
    int field_mode;        // 0 = binned tiles (default), 1 = global-atomic scatter, 2 = mask-scan tiles (A/B)
    int level_mode;        // stage A: 0 = auto (fused kernel from fused_min_batch sequences on, else one-pass k_level when >= 192 planes in flight), 1 = multi-pass, 2 = k_level, 3 = fused
    int fused_min_batch;   // EDGEHIP_FUSED_MIN_BATCH
    double pinv_host[3 * 49];  // plane-fit pseudo inverse, host copy: 3 x (2 ws + 1)^2 values, ws <= 3 (the fused stage A reads the 5x5 one)
    int32_t *bin_cnt;      // [B][256] KeyLines binned per field tile
    int32_t *bins;         // [B][256][CAP] KeyLine ids per field tile (allocated for the tiles in use)
    int nblk_tvr;
    bool fused_undist;     // EDGEHIP_FUSED_UNDIST=1: the one-kernel stage A resamples through the distortion map inside its load (default: k_undistort_grey first)
    int tvr_rw2;           // EDGEHIP_TVR_RW2: from this many evaluation blocks per launch on, the reweighted evaluation takes two KeyLines per thread (0 = never)
    bool tracker_f32 = false;   // edgehip_set_tracker_precision(ctx, 32): Minimizer_RV<float> / TryVelRot<float> (k_try_velrot_f32), the reference's USE_NE10 instantiation
    int dual_init;         // EDGEHIP_DUAL_INIT (default 1): the two initialisation chains of TrackerInitType = 2 share their launches (stage_b.hip tvr2_body)
    int persist_lm_max;    // batches up to this many sequences fuse every TryVelRot evaluation with the LM step after it (EDGEHIP_PERSIST_LM, 0 = never)
    unsigned *sync_cnt;    // [B] per-sequence block tickets of k_try_velrot_lm (0 between launches)
    unsigned long long *fwd_key;  // [B][CAP] forward-match arbitration keys
    int32_t *fwd_win;      // [B][CAP]
    int fwd_mode = 0;              // EDGEHIP_FWD_MODE: 0 keys by the minimiser + k_fwd_win / k_fwd_apply / k_rotate, 1 the round-2 chain (own key pass),
                                   // 2 keys by the minimiser + k_fwd_win + k_fwd_apply_rotate (one scattering pass over the old KeyLines)
    // Matching in one pass (whole-frame driver, ImuMode 0, no stereo pair; EDGEHIP_FUSE_MATCH=0 keeps the three-kernel form): rotate_keylines
    // writes the turned p_m / m_m / rho / s_rho of the old slot into rot_* instead of in place, and k_directed — which visits every new
    // KeyLine anyway — takes the search prior and, where it finds no match, FordwardMatch's ten fields from the UNTURNED old KeyLine
    // that won the forward arbitration: k_fwd_apply's pass (ten gathers and ten stores per KeyLine, nine tenths of them overwritten by
    // directed_matching a kernel later) is gone.  rot_pending[slot]: the slot's own arrays still hold the unturned values; whoever else
    // wants the slot (any stage-level entry point, a download) gets them turned first (rot_materialize_enqueue).
    bool fuse_match = true;
    float2 *rot_pm = nullptr, *rot_mm = nullptr;     // [S][B][CAP]
    double *rot_rho = nullptr, *rot_srho = nullptr;
    bool rot_pending[4] = {false, false, false, false};
    bool fwd_fill[4] = {false, false, false, false};   // [slot] its detector left the forwarded KeyLine fields to FordwardMatch (fill mode)
    bool fwd_cleared = false;      // fwd_key / fwd_win of the new edge map were reset by k_field_bin (whole-frame driver)
    bool fwd_keys_posted = false;   // minimizer_v_enqueue: its last evaluation posted FordwardMatch's keys
    bool fwd_key_in_tvr = false;   // whole-frame driver: the minimiser's last evaluation also posts FordwardMatch's arbitration keys
    double *rs_tmp;        // [B][2][CAP] regularised (rho, s_rho) ping-pong
    double *rot_buf;       // [B][9] rotation applied by rotate_keylines
    const double *t_src = nullptr;   // page-locked time stamps of the frame being enqueued (read in place by k_frame_glue mode 0)
    // edgehip_download_keylines_batch: AoS staging on the device ([requests][CAP] KeyLine records), request lists, page-locked mirror
    edgehip_keyline *aos_dev = nullptr;
    edgehip_keyline *aos_host = nullptr;   // page-locked, same shape
    int32_t *aos_req_dev = nullptr, *aos_req_host = nullptr;   // [2][requests]: sequence ids | KeyLine counts (host side page-locked)
    int aos_requests = 0;
    // edgehip_export_keylines / _fetch / _wait: the AoS KeyLine lists of output callbacks without a host synchronisation — packed
    // in-stream behind the frame that finishes with the slot into a staging ring on the device, copied out on a stream of their own
    struct KlExport {
        static constexpr int R = 4;                 // tickets in flight (a group keeps at most three: two steps in flight + the one being delivered)
        hipStream_t stream = nullptr;               // the copies to the host (never the log's stream: edgehip_read_nav_log synchronises that one)
        edgehip_keyline *dev = nullptr;             // [R][n_cap][CAP]
        int32_t *req = nullptr;                     // page-locked [R][n_cap]: sequence ids, read in place by the packing kernel
        int n_cap = 0;
        hipEvent_t ev_pack[R] = {}, ev_done[R] = {};
        edgehip_keyline *host[R] = {};              // page-locked [n_cap][CAP] per ticket, allocated when a destination is not page-locked itself
        struct Ticket { long long id = -1; int n = 0; bool fetched = false; std::vector<edgehip_keyline *> staged_dst; std::vector<int32_t> staged_kn; } t[R];
        long long next = 0;
    } *kl_export = nullptr;
    edgehip_nav *nav_dev;  // [B] per-frame record
    edgehip_nav *nav_log;  // [nav_log_len][B] ring of per-frame records (optional)
    edgehip_nav_imu *nav_imu_log = nullptr;   // the IMU part of the same records, same ring (allocated when both the log and the IMU branch are on)
    int nav_log_len;
    int32_t *stereo_log = nullptr;   // [nav_log_len][B] stereo_match_num of the logged frames (a context with stereo_available and a log)
    // The log is read by a thread of its own while another enqueues frames (shard.NavMover): the read-out has its own stream,
    // ordered after the frames it covers by ev_log (re-recorded behind every frame on the stream that writes the records);
    // it never touches c->stream, which may be capturing a frame graph.  log_mu orders the record / wait pair on the event.
    hipStream_t stream_log = nullptr;
    hipEvent_t ev_log = nullptr;
    hipEvent_t ev_log_ring[8] = {};            // [frame % 8] behind that frame's record: a reader of frames up to f waits for f, not for everything enqueued since
    std::mutex log_mu;
    std::atomic<long long> frames_logged{0};   // frames enqueued since the log was set
    long long log_first = 0, log_last = -1;    // frame numbers (frames_seen at enqueue) of the first / newest logged frame (under log_mu)
    int32_t *idx_dev;      // [B] frame-pool indices of upload_rgb_indexed
    int32_t *stereo_cnt;   // [B] stereo match counters (params.stereo_available)
    // host staging
    uint8_t *pinned_rgb;   // [B][N*3]
    size_t pinned_rgb_bytes;
    edgehip::SeqDev *pinned_seq;  // [B]
    edgehip::SeqA *pinned_seqa;   // [B]
    double *pinned_out;    // misc readback
    double *pinned_t;      // [8][B] time stamps, one row per frame ring entry
    int32_t *pinned_idx;   // [8][4][B] frame-pool indices, one row per (frame ring entry, slot)
    edgehip_nav *pinned_nav;  // [B]
    edgehip::Profiler *prof;
    // the IMU branch of the frame driver (edgehip_imu_enable; stage_imu.hip)
    bool imu_enabled = false, imu_pending = false;
    bool imu_pinned_ok = false;   // every buffer / stream / event of the IMU branch exists (edgehip_imu_enable is all-or-nothing)
    edgehip_imu_params imu_params;
    edgehip_kf_request *kf_req_dev = nullptr;        // [B] edgehip_minimizer_rv_kf (allocated on first use)
    edgehip_kf_result *kf_res_dev = nullptr;
    void *imu_track = nullptr;                       // [B] ImuTrackDev (main stream)
    void *imu_filter = nullptr;                      // [B] ImuFilterDev (IMU stream)
    void *imu_snap = nullptr;                        // [2][B] ImuSnap: a frame's hand-over from the main to the IMU stream
    hipStream_t stream_imu = nullptr;                // scale filter + pose + nav record of frame k, under frame k+1
    hipEvent_t ev_imu_snap[2], ev_imu_post[2];       // [frame & 1] snapshot written (main) / consumed (IMU stream)
    hipEvent_t ev_imu_mid[2];                        // [frame & 1] k_imu_mid done: the scale filter may start
    bool imu_post_valid[2] = {false, false};
    edgehip_imu_integrated *imu_in_dev = nullptr;    // [B] integrated IMU data of the frame being enqueued
    edgehip_nav_imu *nav_imu_dev = nullptr;          // [B]
    edgehip_imu_integrated *pinned_imu = nullptr;    // [8][B] ring, like the time stamps
    edgehip_nav_imu *pinned_nav_imu = nullptr;       // [B]
};

namespace edgehip {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define EH_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return ::edgehip::hip_fail(_e, #expr, __FILE__, __LINE__);    \
    } while (0)

#define EH_LAUNCH_CHECK() EH_CHECK(hipGetLastError())

// First statement of every entry point that takes a context.  (i) The calls below go to the context's device whatever
// device the calling thread had current (one rebvo::REBVO per GPU in one process).  (ii) HIP keeps the last error of a
// thread until somebody reads it (ROCm 7: later successful calls do not clear it), and EH_LAUNCH_CHECK reads that slot
// after a kernel launch: an error some earlier call left behind — a failed hipMalloc of an edgehip_create that ran out of
// memory, or another library's — must not be reported by the next launch, so the slot is emptied on the way in.
void enter_ctx(edgehip_ctx *c);
#define EH_ENTER(c)                                                                         \
    do {                                                                                    \
        if (!(c)) { ::edgehip::set_error("null context"); return EDGEHIP_ERR_ARG; }         \
        ::edgehip::enter_ctx(c);                                                            \
    } while (0)

// monotone map double -> u64, never 0: FordwardMatch's "larger rho wins" as an integer atomicMax
__host__ __device__ inline unsigned long long ord_bits(double v) {
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

inline KlSoA &klof(edgehip_ctx *c, int slot, int seq) { return c->kl[(size_t)slot * c->plan.nseq + seq]; }
inline KlSoA *kldev(edgehip_ctx *c, int slot) { return c->kl_dev + (size_t)slot * c->plan.nseq; }
inline int32_t *maskof(edgehip_ctx *c, int slot) { return c->mask + (size_t)slot * c->plan.nseq * c->plan.n; }
inline uint8_t *rgbof(edgehip_ctx *c, int slot) { return c->rgb + (size_t)slot * c->plan.nseq * c->plan.n * 3; }

// ---- profiler: HIP events around kernel groups on the context stream ---------------------------------
enum ProfId {
    PROF_A_ROWSCAN = 0, PROF_A_COLSCAN, PROF_A_AVGROW, PROF_A_DETECT, PROF_A_COMPACT, PROF_A_JOIN,
    PROF_B_QUANTILE, PROF_B_FIELD, PROF_B_PREP, PROF_B_TRYVELROT, PROF_B_LMSTEP,
    PROF_C_FORWARD, PROF_C_ROTATE, PROF_C_DIRECTED, PROF_C_REGEKF, PROF_C_RESCALE, PROF_C_POSE,
    PROF_A_LEVEL, PROF_B_MINIMIZER, PROF_A_FUSED, PROF_IMU_FILTERS, PROF_IMU_SCALE_POSE, PROF_B_MINIMIZER_V, PROF_C_EXTROTVEL,
    PROF_B_TRYVELROT2,   // the two-chain evaluation (k_try_velrot2): its own bytes per KeyLine (40 + 2 x 44)
    PROF_COUNT
};
struct Profiler {
    bool on = false;
    uint64_t mask = ~0ull;
    struct Rec { hipEvent_t a, b; int id; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double ms[PROF_COUNT] = {0};
    int64_t calls[PROF_COUNT] = {0};
};
struct ProfScope {  // RAII bracket; no-op unless profiling is enabled
    edgehip_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr; hipStream_t st = nullptr;
    ProfScope(edgehip_ctx *ctx, int pid, hipStream_t stream = nullptr);
    ~ProfScope();
};

// stage entry points shared between translation units (all enqueue on c->stream)
int stage_a_enqueue(edgehip_ctx *c, int slot, bool fwd_fills = false, bool defer_retune = false);   // defer_retune: k_quantile of the same frame finishes reEstimateThresh
// ordering between the two streams for entry points that are not edgehip_process_frame: everything enqueued so far on
// one stream is finished before anything enqueued afterwards on the other starts
int wait_upload(edgehip_ctx *c, int slot, hipStream_t st);   // make `st` wait for a pending stream_up upload into the slot
int wait_pinned_ring(edgehip_ctx *c);     // before writing entry frames_seen % 8 of the pinned time-stamp / frame-index rings
void drop_frame_graphs(edgehip_ctx *c);   // after anything that changes what a captured frame would enqueue
int order_a_after_bc(edgehip_ctx *c);
int order_bc_after_a(edgehip_ctx *c);
int sync_all(edgehip_ctx *c);
int undistort_frame_enqueue(edgehip_ctx *c, int seq, int slot, uint8_t *out_dev);
// SecondThread's frame begin (rebvo_second_t.cpp:145-168): dt, the tracker's priors, the per-frame counters and flags
__device__ inline void frame_begin(SeqDev *sq, double t, double fps) {
    edgehip_seq_state &p = sq->pub;
    double dt = t - p.t_prev;
    if (dt < 0.001) dt = 1 / fps;
    p.dt = dt;
    sq->t_cur = t;
    for (int i = 0; i < 9; i++) {
        const double d = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        p.P_V[i] = d * 1e50; p.P_W[i] = d * 1e50; p.R[i] = d;
    }
    p.klm_fwd = 0; p.klm_num = 0; p.kf_matchs = 0;
    p.estimation_ok = 1;
    sq->skip_match = 0;
    sq->skip_map = 0;
    p.minimizer_evals = 0;
}

int quantile_enqueue(edgehip_ctx *c, int slot, double smin, double smax, double pct, int nbins, bool frame_begins = false, int retune_slot = -1);
int build_field_enqueue(edgehip_ctx *c, int slot, int radius, float min_mod, bool clear_fwd = false);
int tvr_prepare_enqueue(edgehip_ctx *c, int slot_old, unsigned begin_ops = 0);   // begin_ops: LM ops of the step that opens a minimisation, run in the same launch
int minimizer_enqueue(edgehip_ctx *c, int slot_new, int slot_old, int fc_index);
int minimizer_v_enqueue(edgehip_ctx *c, int slot_new, int slot_old, int fc_index, int iter_max, double match_thresh,
                         uint32_t match_num_thresh, double reweight_distance);
int forward_match_enqueue(edgehip_ctx *c, int slot_old, int slot_new, bool keys_posted = false, bool frame_tail = false);
int forward_rotate_enqueue(edgehip_ctx *c, int slot_old, int slot_new);   // FordwardMatch (keys already posted by the minimiser) + rotate_keylines(exp(W))
int rotate_enqueue(edgehip_ctx *c, int slot, const double *R_host, bool R_in_buf = false);
int rot_materialize_enqueue(edgehip_ctx *c, int slot);   // rot_pending[slot]: the turned values into the slot's own arrays
int rec_refresh_enqueue(edgehip_ctx *c, int slot);   // KlSoA::rec's copy of m_m, if rotate_keylines has turned m_m since (edgehip_ctx::rec_stale)
int directed_enqueue(edgehip_ctx *c, int slot_new, int slot_old, bool fused = false);
int regekf_enqueue(edgehip_ctx *c, int slot, int do_reg, int do_ekf, bool frame_glue = false);
int rescale_enqueue(edgehip_ctx *c, int slot, bool frame_ends = false);
int pose_enqueue(edgehip_ctx *c, int slot_new, const double *t_host);
int imu_begin_enqueue(edgehip_ctx *c);                // stage_imu.hip
int imu_reset_enqueue(edgehip_ctx *c);
int imu_pose_reset_enqueue(edgehip_ctx *c, int seq);   // REBVO::Reset()'s pose part, on the IMU stream
int imu_pre_enqueue(edgehip_ctx *c, int slot_old);
int imu_mid_enqueue(edgehip_ctx *c);
int imu_post_enqueue(edgehip_ctx *c, int slot_new, int have_pair);

}  // namespace edgehip
