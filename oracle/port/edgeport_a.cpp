// edgeport_a.cpp — CPU restatement of REBVO's per-frame edge pipeline, stage A (scale space + KeyLine
// extraction + optional undistortion).  TEST INFRASTRUCTURE ONLY: the checker behind tests/, smoke() and
// bench.py's cpu_baseline; nothing under rebvo_amd/ includes, links or loads it.
//
// Plain sequential C++ written from the reference's algorithm; every function names the reference lines it
// follows (paths relative to the reference tree).  Build flags (-O2 -ffp-contract=off, no -march=native,
// oracle/Makefile) make float32 results the canonical x86-64 SSE2 ones, which is what "bit-exact" means here.
// Pinned against the reference itself: tests/test_golden_cpu.py runs this port on the fixtures produced by the
// reference's own code (tests/golden/*.npz) and against oracle/_ref when that is built.
#include "edgeport.h"

#include <cmath>
#include <cstring>

namespace port {

// ---- iigauss::iigauss: box widths after Kovesi (src/mtracklib/iigauss.cpp:43-81) ---------------------------
double kovesi_boxes(double sigma, int box_num, int *box_d) {
    const double wideal = std::sqrt(12 * sigma * sigma / box_num + 1);
    int wl = (int)wideal;
    const int tmp = wl / 2;
    if (tmp * 2 == wl) wl--;
    const int m = (int)std::round((3 * box_num + 4 * box_num * wl + box_num * wl * wl - 12 * sigma * sigma) / (4 + 4 * wl));
    int i;
    for (i = 0; i < m; i++) box_d[i] = wl;
    for (; i < box_num; i++) box_d[i] = wl + 2;
    return std::sqrt((m * wl * wl + (box_num - m) * (wl + 2.0) * (wl + 2.0) - box_num) / 12.0);  // sigma_r
}

// ---- iimage::build_average: reciprocal pixel count of the (border-clipped) box (iimage.cpp:134-179) ----------
void build_average(int d, int w, int h, std::vector<float> &div) {
    div.resize((size_t)w * h);
    const int d2 = d / 2, a = d * d;
    auto D = [&](int x, int y) -> float & { return div[(size_t)y * w + x]; };
    int x, y;
    for (y = 0; y < d2 + 1; y++) {
        for (x = 0; x < d2 + 1; x++) D(x, y) = (x + d2 + 1) * (y + d2 + 1);
        for (; x < w - d2; x++) D(x, y) = d * (y + d2 + 1);
        for (; x < w; x++) D(x, y) = (w - x + d2) * (y + d2 + 1);
    }
    for (; y < h - d2; y++) {
        for (x = 0; x < d2 + 1; x++) D(x, y) = (x + d2 + 1) * d;
        for (; x < w - d2; x++) D(x, y) = a;
        for (; x < w; x++) D(x, y) = (w - x + d2) * d;
    }
    for (; y < h; y++) {
        for (x = 0; x < d2 + 1; x++) D(x, y) = (h - y + d2) * (x + d2 + 1);
        for (; x < w - d2; x++) D(x, y) = (h - y + d2) * d;
        for (; x < w; x++) D(x, y) = (h - y + d2) * (w - x + d2);
    }
    for (size_t i = 0; i < div.size(); i++) div[i] = 1.0 / div[i];   // double division, stored as float (:176-178)
}

// ---- iimage::load: float32 integral image, row prefix then column prefix (iimage.cpp:53-71) -----------------
// The sums exceed 2^24, so every add rounds: the sequential order below IS the specification.
static void iimage_load(const float *l, float *img, int w, int h) {
    for (int y = 0; y < h; y++) {
        img[(size_t)y * w] = l[(size_t)y * w];
        for (int x = 1; x < w; x++) img[(size_t)y * w + x] = img[(size_t)y * w + x - 1] + l[(size_t)y * w + x];
    }
    for (int x = 0; x < w; x++)
        for (int y = 1; y < h; y++) img[(size_t)y * w + x] += img[(size_t)(y - 1) * w + x];
}

// ---- iimage::average: box mean from 4 integral taps, nine border regions (iimage.cpp:86-128) ----------------
// Note the operand order: rows above the bottom band compute ((A-B)-C)+D with B the left tap, the bottom band
// subtracts the upper tap first; the interior multiplies by a=(float)(1.0/(d*d)), the borders by div(x,y).
static void iimage_average(float *buf, const float *img, int d, const float *div, int w, int h) {
    const int d2 = d / 2;
    const float a = 1.0 / (d * d);
    auto I = [&](int x, int y) -> float { return img[(size_t)y * w + x]; };
    auto B = [&](int x, int y) -> float & { return buf[(size_t)y * w + x]; };
    auto Dv = [&](int x, int y) -> float { return div[(size_t)y * w + x]; };
    int x, y;
    for (y = 0; y < d2 + 1; y++) {
        for (x = 0; x < d2 + 1; x++) B(x, y) = I(x + d2, y + d2) * Dv(x, y);
        for (; x < w - d2; x++) B(x, y) = (I(x + d2, y + d2) - I(x - d2 - 1, y + d2)) * Dv(x, y);
        for (; x < w; x++) B(x, y) = (I(w - 1, y + d2) - I(x - d2 - 1, y + d2)) * Dv(x, y);
    }
    for (; y < h - d2; y++) {
        for (x = 0; x < d2 + 1; x++) B(x, y) = (I(x + d2, y + d2) - I(x + d2, y - d2 - 1)) * Dv(x, y);
        for (; x < w - d2; x++)
            B(x, y) = (I(x + d2, y + d2) - I(x - d2 - 1, y + d2) - I(x + d2, y - d2 - 1) + I(x - d2 - 1, y - d2 - 1)) * a;
        for (; x < w; x++)
            B(x, y) = (I(w - 1, y + d2) - I(x - d2 - 1, y + d2) - I(w - 1, y - d2 - 1) + I(x - d2 - 1, y - d2 - 1)) * Dv(x, y);
    }
    for (; y < h; y++) {
        for (x = 0; x < d2 + 1; x++) B(x, y) = (I(x + d2, h - 1) - I(x + d2, y - d2 - 1)) * Dv(x, y);
        for (; x < w - d2; x++)
            B(x, y) = (I(x + d2, h - 1) - I(x + d2, y - d2 - 1) - I(x - d2 - 1, h - 1) + I(x - d2 - 1, y - d2 - 1)) * Dv(x, y);
        for (; x < w; x++)
            B(x, y) = (I(w - 1, h - 1) - I(w - 1, y - d2 - 1) - I(x - d2 - 1, h - 1) + I(x - d2 - 1, y - d2 - 1)) * Dv(x, y);
    }
}

// ---- iigauss::smooth (iigauss.cpp:91-101) --------------------------------------------------------------------
static void smooth(const Filter &f, const float *in, float *out, float *integral, int w, int h) {
    iimage_load(in, integral, w, h);
    for (int i = 0; i < kBoxes - 1; i++) {
        iimage_average(out, integral, f.box_d[i], f.div[i].data(), w, h);
        iimage_load(out, integral, w, h);
    }
    iimage_average(out, integral, f.box_d[kBoxes - 1], f.div[kBoxes - 1].data(), w, h);
}

// ---- sspace::build / build_dog / calc_gradient (src/mtracklib/sspace.cpp:52-85) --------------------------------
void sspace_build(Ctx &c, Slot &s) {
    const int w = c.p.w, h = c.p.h;
    smooth(c.filter[0], s.bw.data(), s.img0.data(), c.integral.data(), w, h);
    smooth(c.filter[1], s.bw.data(), s.img1.data(), c.integral.data(), w, h);
    const size_t n = (size_t)w * h;
    for (size_t k = 0; k < n; k++) s.dog[k] = s.img1[k] - s.img0[k];
    for (int y = 1; y < h - 1; y++)                       // interior only; the border of dx/dy is never written
        for (int x = 1; x < w - 1; x++) {
            s.dx[(size_t)y * w + x] = s.img0[(size_t)y * w + x + 1] - s.img0[(size_t)y * w + x - 1];
            s.dy[(size_t)y * w + x] = s.img0[(size_t)(y + 1) * w + x] - s.img0[(size_t)(y - 1) * w + x];
        }
}

// ---- plane-fit pseudo inverse: PInv = Matrix3x3Inv(Phi^T Phi) * Phi^T (edge_finder.cpp:83-100,
// include/UtilLib/toon_util.h:32-41; determinant of the diagonal Phi^T Phi = product of the diagonal) --------
void plane_fit_pinv(int win_s, std::vector<double> &pinv) {
    const int nn = (2 * win_s + 1) * (2 * win_s + 1);
    std::vector<double> Phi((size_t)nn * 3);
    for (int i = -win_s, k = 0; i <= win_s; i++)
        for (int j = -win_s; j <= win_s; j++, k++) {
            Phi[k * 3 + 0] = j;
            Phi[k * 3 + 1] = i;
            Phi[k * 3 + 2] = 1;
        }
    double A[3][3];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) {
            double s = 0;
            for (int k = 0; k < nn; k++) s += Phi[k * 3 + r] * Phi[k * 3 + q];
            A[r][q] = s;
        }
    double B[3][3];
    B[0][0] = A[2][2] * A[1][1] - A[2][1] * A[1][2]; B[0][1] = -(A[2][2] * A[0][1] - A[2][1] * A[0][2]); B[0][2] = A[1][2] * A[0][1] - A[1][1] * A[0][2];
    B[1][0] = -(A[2][2] * A[1][0] - A[2][0] * A[1][2]); B[1][1] = A[2][2] * A[0][0] - A[2][0] * A[0][2]; B[1][2] = -(A[1][2] * A[0][0] - A[1][0] * A[0][2]);
    B[2][0] = A[2][1] * A[1][0] - A[2][0] * A[1][1]; B[2][1] = -(A[2][1] * A[0][0] - A[2][0] * A[0][1]); B[2][2] = A[1][1] * A[0][0] - A[1][0] * A[0][1];
    const double det = A[0][0] * A[1][1] * A[2][2];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) B[r][q] = B[r][q] / det;
    pinv.assign((size_t)3 * nn, 0.0);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < nn; k++) {
            double s = 0;
            for (int j = 0; j < 3; j++) s += B[r][j] * Phi[k * 3 + j];
            pinv[(size_t)r * nn + k] = s;
        }
}

// ---- edge_finder::build_mask (src/mtracklib/edge_finder.cpp:67-214) ---------------------------------------------
static void build_mask(Ctx &c, Slot &s, int kl_max, int win_s, float per_hist, float grad_thesh, float dog_thesh) {
    const int w = c.p.w, h = c.p.h, nn = (2 * win_s + 1) * (2 * win_s + 1);
    const int max_img_value = 255 * 3;                                  // rebvo.cpp:300
    if (kl_max > (int)s.kl.size()) kl_max = (int)s.kl.size();
    s.kn = 0;
    std::vector<double> Y(nn);
    const double *PInv = c.pinv.data();
    for (int y = win_s; y < h - win_s; y++) {
        for (int x = win_s; x < w - win_s; x++) {
            int img_inx = y * w + x;
            s.mask[img_inx] = -1;
            const float gx = s.dx[img_inx], gy = s.dy[img_inx];
            const float n2gI = gx * gx + gy * gy;                       // util::norm2
            const float gt = grad_thesh * max_img_value;
            if (n2gI < gt * gt) continue;                               // :117-119
            int pn = 0;
            for (int i = -win_s, k = 0; i <= win_s; i++)
                for (int j = -win_s; j <= win_s; j++, k++) {
                    const float v = s.dog[(size_t)(y + i) * w + x + j];
                    Y[k] = v;
                    if (v > 0) pn++; else pn--;
                }
            if (std::fabs((double)pn) > ((float)((2.0 * win_s + 1.0) * (2.0 * win_s + 1.0))) * per_hist) continue;   // :137
            double theta[3];
            for (int r = 0; r < 3; r++) {                               // theta = PInv * Y (sums from 0, in order)
                double d = 0;
                for (int k = 0; k < nn; k++) d += PInv[(size_t)r * nn + k] * Y[k];
                theta[r] = d;
            }
            const float xs = -theta[0] * theta[2] / (theta[0] * theta[0] + theta[1] * theta[1]);
            const float ys = -theta[1] * theta[2] / (theta[0] * theta[0] + theta[1] * theta[1]);
            if (std::fabs(xs) > 0.5 || std::fabs(ys) > 0.5) continue;   // :149
            const float mx = (float)theta[0], my = (float)theta[1];
            const float n2_m = mx * mx + my * my;
            const float gd = grad_thesh * max_img_value * dog_thesh;
            if (n2_m < gd * gd) continue;                               // :157-159
            OrcKeyLine &k = s.kl[s.kn];
            std::memset(&k, 0, sizeof k);
            k.p_inx = img_inx;
            k.m_m[0] = mx; k.m_m[1] = my;
            k.n_m = std::sqrt(n2_m);
            k.u_m[0] = k.m_m[0] / k.n_m; k.u_m[1] = k.m_m[1] / k.n_m;
            k.c_p[0] = x + xs; k.c_p[1] = y + ys;
            k.p_m[0] = k.c_p[0] - c.ppx; k.p_m[1] = k.c_p[1] - c.ppy;  // cam_model::Img2Hom
            k.p_m_0[0] = k.p_m[0]; k.p_m_0[1] = k.p_m[1];
            k.rho = kRhoInit; k.s_rho = kRhoMax; k.rho0 = kRhoInit; k.s_rho0 = kRhoMax; k.rho_nr = kRhoInit; k.s_rho_nr = kRhoMax;
            k.m_num = 0;
            k.n_id = -1; k.p_id = -1; k.net_id = -1; k.m_id = -1; k.m_id_f = -1; k.m_id_kf = -1;
            k.stereo_m_id = -1; k.stereo_rho = kRhoInit; k.stereo_s_rho = kRhoMax;
            s.mask[img_inx] = s.kn;
            if (++s.kn >= kl_max) {                                     // :203-209
                for (++img_inx; img_inx < w * h; img_inx++) s.mask[img_inx] = -1;
                return;
            }
        }
    }
}

// ---- NextPoint + edge_finder::join_edges (edge_finder.cpp:221-320) ---------------------------------------------
static int next_point(int x, int y, const float m[2], const int32_t *mask, int w) {
    const float tx = -m[1], ty = m[0];
    auto M = [&](int xx, int yy) { return mask[(size_t)yy * w + xx]; };
    int k;
    if (ty > 0) {
        if (tx > 0) { if ((k = M(x + 1, y)) >= 0) return k; if ((k = M(x, y + 1)) >= 0) return k; if ((k = M(x + 1, y + 1)) >= 0) return k; }
        else        { if ((k = M(x - 1, y)) >= 0) return k; if ((k = M(x, y + 1)) >= 0) return k; if ((k = M(x - 1, y + 1)) >= 0) return k; }
    } else {
        if (tx < 0) { if ((k = M(x - 1, y)) >= 0) return k; if ((k = M(x, y - 1)) >= 0) return k; if ((k = M(x - 1, y - 1)) >= 0) return k; }
        else        { if ((k = M(x + 1, y)) >= 0) return k; if ((k = M(x, y - 1)) >= 0) return k; if ((k = M(x + 1, y - 1)) >= 0) return k; }
    }
    return -1;
}
static void join_edges(Ctx &c, Slot &s) {
    for (int ikl = 0; ikl < s.kn; ikl++) {
        const int x = (int)(s.kl[ikl].c_p[0] + 0.5);                    // util::round2int_positive (util.h:41-44)
        const int y = (int)(s.kl[ikl].c_p[1] + 0.5);
        const int ikl2 = next_point(x, y, s.kl[ikl].m_m, s.mask.data(), c.p.w);
        if (ikl2 < 0) continue;
        s.kl[ikl2].p_id = ikl;                                          // sequential: the last writer wins
        s.kl[ikl].n_id = ikl2;
    }
}

// ---- edge_finder::detect + UpdateThresh (edge_finder.cpp:330-365) -----------------------------------------------
void detect(Ctx &c, Slot &s, double &tresh, int &l_kl_num) {
    const OrcParams &p = c.p;
    if (p.auto_gain > 0) {
        tresh -= p.auto_gain * (double)(p.reference_points - l_kl_num);
        tresh = tresh > p.max_thresh ? p.max_thresh : (tresh < p.min_thresh ? p.min_thresh : tresh);   // util::Constrain
    }
    build_mask(c, s, p.max_points, p.plane_fit_size, (float)p.pos_neg_thresh, (float)tresh, (float)p.dog_thresh);
    join_edges(c, s);
    l_kl_num = s.kn;
}

// ---- edge_finder::reEstimateThresh (edge_finder.cpp:373-405) ------------------------------------------------------
// Reproduces the loop `for(int a=0;i<n && a<knum;i++,a+=histo[i]);` literally: histo[i] is added AFTER i++, so
// bin 0 is never counted and the last step reads histo[n] (one past the end: whatever lies there no longer
// influences i).  The reference reads kl[0].n_m even when kn == 0; here an empty list returns 0.
float re_estimate_thresh(Slot &s, int knum, int n) {
    if (s.kn <= 0) return s.retuned = 0.f;
    float max_dog = s.kl[0].n_m, min_dog = s.kl[0].n_m;
    for (int ikl = 1; ikl < s.kn; ikl++) {
        if (s.kl[ikl].n_m > max_dog) max_dog = s.kl[ikl].n_m;
        if (s.kl[ikl].n_m < min_dog) min_dog = s.kl[ikl].n_m;
    }
    std::vector<int> histo(n + 1, 0);
    for (int ikl = 0; ikl < s.kn; ikl++) {
        int i = n * (max_dog - s.kl[ikl].n_m) / (max_dog - min_dog);   // float expression -> int (cvttss2si)
        i = i > n - 1 ? n - 1 : i;
        i = i < 0 ? 0 : i;
        histo[i]++;
    }
    int i = 0;
    for (int a = 0; i < n && a < knum; i++, a += (i < n ? histo[i] : 0));
    return s.retuned = max_dog - (float)i * (max_dog - min_dog) / (float)n;
}

// ---- image_undistort (include/VideoLib/image_undistort.h:38-122, src/VideoLib/image_undistort.cpp:29-95) ----------
void build_undistort_map(Ctx &c) {
    const int w = c.p.w, h = c.p.h;
    c.umap.assign((size_t)w * h, UndistPoint());
    const double Kc2 = c.p.kc[0], Kc4 = c.p.kc[1], Kc6 = c.p.kc[2], P1 = c.p.kc[3], P2 = c.p.kc[4];
    auto valid = [&](float fx, float fy) {   // Image::isInxValid(const uint&, const uint&) fed with floats
        const uint32_t ux = (uint32_t)(int64_t)fx, uy = (uint32_t)(int64_t)fy;
        return ux < (uint32_t)w && uy < (uint32_t)h;
    };
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            UndistPoint &u = c.umap[(size_t)y * w + x];
            float qx = (float)x - c.ppx, qy = (float)y - c.ppy;                        // cam_model::Img2Hom
            {                                                                          // cam_model::distortHom2Hom (cam_model.h:76-88)
                const double xp = qx / c.zfm, yp = qy / c.zfm;
                const double r2 = xp * xp + yp * yp;
                const double xpp = xp * (1 + r2 * (Kc2 + r2 * (Kc4 + r2 * Kc6))) + 2 * P1 * xp * yp + P2 * (r2 + 2 * xp * xp);
                const double ypp = yp * (1 + r2 * (Kc2 + r2 * (Kc4 + r2 * Kc6))) + P1 * (r2 + 2 * yp * yp) + 2 * P2 * xp * yp;
                qx = xpp * c.zfx;
                qy = ypp * c.zfy;
            }
            const float idx = qx + c.ppx, idy = qy + c.ppy;                            // Hom2Img
            const float p00x = std::floor(idx), p00y = std::floor(idy), p11x = std::floor(idx) + 1, p11y = std::floor(idy) + 1;
            const float tx[4] = {p00x, p11x, p00x, p11x}, ty[4] = {p00y, p00y, p11y, p11y};
            const float wt[4] = {(p11x - idx) * (p11y - idy), (idx - p00x) * (p11y - idy), (p11x - idx) * (idy - p00y),
                                 (idx - p00x) * (idy - p00y)};
            u.num = 0;
            for (int i = 0; i < 4; i++)
                if (valid(tx[i], ty[i])) {
                    u.w[u.num] = wt[i];
                    u.inx[u.num] = (int)std::round(ty[i]) * w + (int)std::round(tx[i]);   // GetIndexRC
                    u.num++;
                }
            if (u.num > 0) {
                float sum_w = 0;
                for (int i = 0; i < u.num; i++) sum_w += u.w[i];
                for (int i = 0; i < u.num; i++) {
                    u.w[i] /= sum_w;
                    u.iw[i] = u.w[i] * 65536.0f;                                          // i_mult = 1<<16
                }
            }
        }
}
void undistort_rgb(const Ctx &c, const uint8_t *in, uint8_t *out) {       // undistort<true> + biInterp (RGB24)
    const size_t n = (size_t)c.p.w * c.p.h;
    for (size_t inx = 0; inx < n; inx++) {
        const UndistPoint &u = c.umap[inx];
        int r = 0, g = 0, b = 0;
        for (int i = 0; i < u.num; i++) {
            r += u.iw[i] * in[(size_t)u.inx[i] * 3 + 0];
            g += u.iw[i] * in[(size_t)u.inx[i] * 3 + 1];
            b += u.iw[i] * in[(size_t)u.inx[i] * 3 + 2];
        }
        out[inx * 3 + 0] = (uint8_t)(r >> 16);
        out[inx * 3 + 1] = (uint8_t)(g >> 16);
        out[inx * 3 + 2] = (uint8_t)(b >> 16);
    }
}

// ---- stage A of one frame: rebvo_first_t.cpp:229-272 ------------------------------------------------------------------
int stage_a(Ctx &c, int slot, const uint8_t *rgb24, double *tresh_io, int *l_kl_num_io) {
    Slot &s = c.slots[slot];
    const size_t n = (size_t)c.p.w * c.p.h;
    if (c.p.use_undistort) undistort_rgb(c, rgb24, s.imgc.data());     // :231
    else std::memcpy(s.imgc.data(), rgb24, n * 3);                     // :250
    for (size_t i = 0; i < n; i++)                                     // Image<float>::ConvertRGB2BW (image.h:197-203)
        s.bw[i] = (int)s.imgc[i * 3] + (int)s.imgc[i * 3 + 1] + (int)s.imgc[i * 3 + 2];
    sspace_build(c, s);                                                // :263
    detect(c, s, *tresh_io, *l_kl_num_io);                             // :266
    re_estimate_thresh(s, c.p.track_points, c.p.qcut_nbins);           // :272
    return s.kn;
}

}  // namespace port
