// imu_filters.h — the scalar filters of the IMU branch, callable from the host library and from device code alike:
//   edge_tracker::BiasCorrect      src/mtracklib/edge_tracker.cpp:1308-1343
//   ScaleEstimator::EstAcelLsq4    src/mtracklib/scaleestimator.cpp:38-92
//   ScaleEstimator::MeanAcel4      src/mtracklib/scaleestimator.cpp:94-109
//   ScaleEstimator::estKaGMEKBias  src/mtracklib/scaleestimator.cpp:117-318 (+ Minimizer<>::GaussNewton,
//                                  include/UtilLib/minimizer.h:84-114)
// Same names and argument meaning as the reference (see rebvo/imu.h for the two deliberate differences).  The host
// library drives them for one live camera (rebvo_imu.cpp); libedgehip runs them on the GPU, one thread per sequence,
// for batches (csrc/stage_imu.hip).
#ifndef REBVO_AMD_HOST_IMU_FILTERS_H
#define REBVO_AMD_HOST_IMU_FILTERS_H

#include "rebvo/linalg.h"

namespace rebvo {

using la::Mat;
using la::Vec;

class ScaleEstimator {
    // histories (function-local statics in the reference)
    la::Vec<3> V, V0, V1, V2, V3;
    double T[5];
    double Dt[4];
    la::Vec<3> A, A0, A1, A2;

public:
    REBVO_HD ScaleEstimator() {
        V = V0 = V1 = V2 = V3 = A = A0 = A1 = A2 = la::Vec<3>::zeros();
        for (int i = 0; i < 5; i++) T[i] = 0;
        for (int i = 0; i < 4; i++) Dt[i] = 0;
    }
    // least-squares slope of the last five (rotated) visual velocities: the visual acceleration
    REBVO_HD inline void EstAcelLsq4(const la::Vec<3> &vel, la::Vec<3> &acel, const la::Mat<3, 3> &R, const double &dt);
    // mean of the last four (rotated) accelerometer readings
    REBVO_HD inline void MeanAcel4(const la::Vec<3> &s_acel, la::Vec<3> &acel, const la::Mat<3, 3> &R);
    // Bayesian scale / gravity / visual-bias filter: linear prior, 20 Gauss-Newton steps on the 11-row problem
    REBVO_HD static inline double estKaGMEKBias(const la::Vec<3> &s_acel, const la::Vec<3> &f_acel, double kP, la::Mat<3, 3> Rot,
                                                la::Vec<7> &X, la::Mat<7, 7> &P, const la::Mat<3, 3> &Qg, const la::Mat<3, 3> &Qrot,
                                                const la::Mat<3, 3> &Qbias, const double &QKp, const double &Rg, const la::Mat<3, 3> &Rs,
                                                const la::Mat<3, 3> &Rf, la::Vec<3> &g_est, la::Vec<3> &b_est, const la::Mat<6, 6> &Wvw,
                                                la::Vec<6> &Xvw, double g_gravit);
};

// ------------------------------------------------------------------------------------------------------------
// BiasCorrect (edge_tracker.cpp:1308-1343)
// ------------------------------------------------------------------------------------------------------------
namespace imufilter {
REBVO_HD inline void BiasCorrect(Vec<6> &X, Mat<6, 6> &Wx, Vec<3> &Gb, Mat<3, 3> &Wb, const Mat<3, 3> &Rg, const Mat<3, 3> &Rb) {
    const Mat<3, 3> I3 = Mat<3, 3>::identity();
    const Mat<3, 3> Wg = la::inv3(Rg);                 // gyro measurement information
    Wb = la::inv3(la::inv3(Wb) + Rb);                  // bias uncertainty update
    Mat<6, 6> Wxb = Wx;
    const Mat<3, 3> iWgWb = la::inv3(Wg + Wb);
    la::set_block(Wxb, 3, 3, la::block<3, 3>(Wxb, 3, 3) + Wg * (I3 - iWgWb * Wg));
    Vec<6> X1 = Wx * X;
    la::set_slice(X1, 3, la::slice<3>(X1, 3) + ((Wg * iWgWb) * Wb) * Gb);
    X = la::Cholesky<6>(Wxb).inverse() * X1;
    Gb = iWgWb * (Wg * la::slice<3>(X, 3) + Wb * Gb);
    Wb = Wg + Wb;
    la::set_block(Wx, 3, 3, la::block<3, 3>(Wx, 3, 3) + Wg);
}
}  // namespace imufilter

// ------------------------------------------------------------------------------------------------------------
// ScaleEstimator
// ------------------------------------------------------------------------------------------------------------
// scaleestimator.cpp:38-92
REBVO_HD inline void ScaleEstimator::EstAcelLsq4(const Vec<3> &vel, Vec<3> &acel, const Mat<3, 3> &R, const double &dt) {
    const Mat<3, 3> Rt = la::transpose(R);
    V3 = Rt * V2;
    V2 = Rt * V1;
    V1 = Rt * V0;
    V0 = Rt * V;
    V = vel;
    for (int i = 0; i < 3; i++) Dt[i] = Dt[i + 1];
    Dt[3] = dt;
    T[0] = 0;
    double mt = 0;
    for (int i = 0; i < 4; i++) {
        T[i + 1] = T[i] + Dt[i];
        mt += T[i + 1];
    }
    mt /= 5;
    double num = 0, den = 0, vm;
    for (int i = 0; i < 5; i++) den += (T[i] - mt) * (T[i] - mt);
    for (int i = 0; i < 3; i++) {
        // :74 adds V[3] — one element past the end of the static 3-vector V — where V3[i] was meant.  The mean only
        // shifts all five samples by the same amount and sum(T[k] - mt) == 0, so any finite value there changes the
        // slope by rounding only; the stray element is taken as 0 (what the oracle build of the reference reads: the
        // zero-initialised .bss that follows V), which reproduces the reference bit for bit.
        vm = (V[i] + V0[i] + V1[i] + V2[i] + 0.0) / 5.0;
        num = (V[i] - vm) * (T[4] - mt);
        num += (V0[i] - vm) * (T[3] - mt);
        num += (V1[i] - vm) * (T[2] - mt);
        num += (V2[i] - vm) * (T[1] - mt);
        num += (V3[i] - vm) * (T[0] - mt);
        if (den > 0) acel[i] = num / den;
    }
}

// scaleestimator.cpp:94-109
REBVO_HD inline void ScaleEstimator::MeanAcel4(const Vec<3> &s_acel, Vec<3> &acel, const Mat<3, 3> &R) {
    const Mat<3, 3> Rt = la::transpose(R);
    A2 = Rt * A1;
    A1 = Rt * A0;
    A0 = Rt * A;
    A = s_acel;
    acel = (((A + A0) + A1) + A2) / 4.0;
}

namespace imufilter_detail {

struct KaGMEKBiasParams {   // FunParams_KaGMEKBias (scaleestimator.cpp:115-124)
    Vec<3> a_v, a_s;
    double G;
    Vec<7> x_p;
    Mat<3, 3> Rv, Rs;
    double Rg;
    Mat<7, 7> Pp;
    Mat<7, 7> W7;   // Cholesky<7>(Pp).get_inverse(): the same for all 21 evaluations of a frame, computed once (problem_KaGMEKBias)
};

// Problem_KaGMEKBias (scaleestimator.cpp:126-199): normal equations of the 11-row residual whose weight depends on
// the scale angle a = x[0] (hence the dW/da terms).  This is the reference's function as it is written, dense 11x11
// algebra and all; the filters run problem_KaGMEKBias below, which leaves out the products with structural zeros
// (tests/test_imu_cpu.py holds the two equal).
REBVO_HD inline void problem_KaGMEKBias_dense(Mat<7, 7> &JtJ, Vec<7> &JtF, const Vec<7> &x, const KaGMEKBiasParams &p) {
    const double a = x[0];
    const Vec<3> g = la::slice<3>(x, 1), b = la::slice<3>(x, 4);
    const Vec<3> &a_s = p.a_s, &a_v = p.a_v;

    Vec<11> F = Vec<11>::zeros();
    la::set_slice(F, 0, (a_s + g) * std::cos(a) - a_v * std::sin(a));
    F[3] = la::dot(g, g) - p.G * p.G;
    F[4] = x[0] - p.x_p[0];   // scale angle prior, wrapped to (-pi, pi]
    if (F[4] > M_PI) F[4] -= 2 * M_PI;
    else if (F[4] < -M_PI) F[4] += 2 * M_PI;
    const Mat<3, 3> Rb = la::so3_exp(b);
    la::set_slice(F, 5, Rb * g - la::slice<3>(p.x_p, 1));   // gravity prior through the bias rotation
    la::set_slice(F, 8, b - la::slice<3>(p.x_p, 4));        // bias prior

    Vec<11> dFda = Vec<11>::zeros();
    la::set_slice(dFda, 0, -(a_s + g) * std::sin(a) - a_v * std::cos(a));
    dFda[4] = 1;

    const Vec<3> Rg = Rb * g;
    Mat<3, 3> Gx;
    Gx(0, 0) = 0; Gx(0, 1) = Rg[2]; Gx(0, 2) = -Rg[1];
    Gx(1, 0) = -Rg[2]; Gx(1, 1) = 0; Gx(1, 2) = Rg[0];
    Gx(2, 0) = Rg[1]; Gx(2, 1) = -Rg[0]; Gx(2, 2) = 0;

    Mat<11, 6> dFdx1 = Mat<11, 6>::zeros();
    la::set_block(dFdx1, 0, 0, Mat<3, 3>::identity() * std::cos(a));
    for (int j = 0; j < 3; j++) dFdx1(3, j) = 2 * g[j];
    la::set_block(dFdx1, 5, 0, Rb);
    la::set_block(dFdx1, 5, 3, Gx);
    la::set_block(dFdx1, 8, 3, Mat<3, 3>::identity());

    const Mat<3, 3> Pz = (std::sin(a) * std::sin(a)) * p.Rv + (std::cos(a) * std::cos(a)) * p.Rs;
    Mat<11, 11> P = Mat<11, 11>::zeros();
    la::set_block(P, 0, 0, Pz);
    P(3, 3) = p.Rg;
    la::set_block(P, 4, 4, p.Pp);
    Mat<11, 11> W = Mat<11, 11>::zeros();
    la::set_block(W, 0, 0, la::Cholesky<3>(Pz).inverse());
    W(3, 3) = 1 / p.Rg;
    la::set_block(W, 4, 4, la::Cholesky<7>(p.Pp).inverse());
    Mat<11, 11> dPda = Mat<11, 11>::zeros();
    la::set_block(dPda, 0, 0, ((2 * std::sin(a)) * std::cos(a)) * (p.Rv - p.Rs));
    const Mat<11, 11> dWda = ((-W) * dPda) * W;

    const Mat<6, 11> Jt = la::transpose(dFdx1);
    JtJ(0, 0) = la::dot((((0.25 * F) * dWda) * P) * dWda, F) + la::dot(dFda * dWda, F) + la::dot(dFda * W, dFda);
    const Vec<6> col = ((0.5 * Jt) * dWda) * F + (Jt * W) * dFda;
    for (int i = 0; i < 6; i++) { JtJ(1 + i, 0) = col[i]; JtJ(0, 1 + i) = col[i]; }
    la::set_block(JtJ, 1, 1, (Jt * W) * dFdx1);
    JtF[0] = la::dot((0.5 * F) * dWda, F) + la::dot(dFda * W, F);
    la::set_slice(JtF, 1, (Jt * W) * F);
}

// The same normal equations without the multiplications by structural zeros.  P, W, dP/da and dW/da are block diagonal
// (3 + 1 + 7), dP/da and dW/da live in the 3x3 block only, dF/da has four non-zero entries and dF/dx1 is made of identity,
// rotation and cross-product blocks: the reference's 11x11x11 products (5.5 k multiply-adds per evaluation, 21 evaluations
// per frame) come down to ~600.  Every sum keeps the reference's order (k ascending from 0) over its non-zero terms, and
// x + (+-0) = x, so the results are the dense function's bit for bit (an exactly-zero sum may differ in the sign of its
// zero).  W7 = Cholesky<7>(Pp).get_inverse() does not depend on x and comes in with the parameters.
REBVO_HD inline void problem_KaGMEKBias(Mat<7, 7> &JtJ, Vec<7> &JtF, const Vec<7> &x, const KaGMEKBiasParams &p) {
    const double a = x[0];
    const double ca = std::cos(a), sa = std::sin(a);
    const Vec<3> g = la::slice<3>(x, 1), b = la::slice<3>(x, 4);
    const Vec<3> &a_s = p.a_s, &a_v = p.a_v;
    const Mat<7, 7> &W7 = p.W7;

    double F[11];
    {
        const Vec<3> fa = (a_s + g) * ca - a_v * sa;
        for (int i = 0; i < 3; i++) F[i] = fa[i];
    }
    F[3] = la::dot(g, g) - p.G * p.G;
    F[4] = x[0] - p.x_p[0];
    if (F[4] > M_PI) F[4] -= 2 * M_PI;
    else if (F[4] < -M_PI) F[4] += 2 * M_PI;
    const Mat<3, 3> Rb = la::so3_exp(b);
    const Vec<3> Rg = Rb * g;
    for (int i = 0; i < 3; i++) { F[5 + i] = Rg[i] - p.x_p[1 + i]; F[8 + i] = b[i] - p.x_p[4 + i]; }
    const Vec<3> da = -(a_s + g) * sa - a_v * ca;          // dF/da, rows 0..2 (row 4 is 1, the rest 0)
    Mat<3, 3> Gx;
    Gx(0, 0) = 0; Gx(0, 1) = Rg[2]; Gx(0, 2) = -Rg[1];
    Gx(1, 0) = -Rg[2]; Gx(1, 1) = 0; Gx(1, 2) = Rg[0];
    Gx(2, 0) = Rg[1]; Gx(2, 1) = -Rg[0]; Gx(2, 2) = 0;

    const Mat<3, 3> Pz = (sa * sa) * p.Rv + (ca * ca) * p.Rs;
    const Mat<3, 3> W3 = la::Cholesky<3>(Pz).inverse();
    const Mat<3, 3> dP3 = ((2 * sa) * ca) * (p.Rv - p.Rs);
    const Mat<3, 3> D3 = ((-W3) * dP3) * W3;                // the non-zero block of dW/da = ((-W) dP/da) W
    const double iRg = 1 / p.Rg;

    // M2 = Jt W (6 x 11), Jt = (dF/dx1)^T
    double M2[6][11];
    for (int r = 0; r < 3; r++) {
        for (int j = 0; j < 3; j++) M2[r][j] = ca * W3(r, j);
        M2[r][3] = (2 * g[r]) * iRg;
        for (int jj = 0; jj < 7; jj++) {
            double t = 0;
            for (int ii = 1; ii <= 3; ii++) t += Rb(ii - 1, r) * W7(ii, jj);
            M2[r][4 + jj] = t;
        }
    }
    for (int rr = 0; rr < 3; rr++) {
        for (int j = 0; j < 4; j++) M2[3 + rr][j] = 0;
        for (int jj = 0; jj < 7; jj++) {
            double t = 0;
            for (int ii = 1; ii <= 3; ii++) t += Gx(ii - 1, rr) * W7(ii, jj);
            M2[3 + rr][4 + jj] = t + W7(4 + rr, jj);
        }
    }

    // JtJ(0,0) = (((F/4) dW) P) dW . F + (dFda dW) . F + (dFda W) . dFda
    double v1[3], v2[3], v3[3], u[3], w[3];
    for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += (0.25 * F[i]) * D3(i, j); v1[j] = t; }
    for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += v1[i] * Pz(i, j); v2[j] = t; }
    for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += v2[i] * D3(i, j); v3[j] = t; }
    for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += da[i] * D3(i, j); u[j] = t; }
    for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += da[i] * W3(i, j); w[j] = t; }
    double t1 = 0, t2 = 0, t3 = 0;
    for (int j = 0; j < 3; j++) { t1 += v3[j] * F[j]; t2 += u[j] * F[j]; t3 += w[j] * da[j]; }
    t3 += W7(0, 0);
    JtJ(0, 0) = t1 + t2 + t3;

    // first row / column: ((Jt/2) dW) F + (Jt W) dFda
    for (int r = 0; r < 6; r++) {
        double c1 = 0;
        if (r < 3) for (int j = 0; j < 3; j++) c1 += ((0.5 * ca) * D3(r, j)) * F[j];
        double c2 = 0;
        for (int j = 0; j < 3; j++) c2 += M2[r][j] * da[j];
        c2 += M2[r][4];
        const double c = c1 + c2;
        JtJ(1 + r, 0) = c;
        JtJ(0, 1 + r) = c;
    }
    // (Jt W) dF/dx1
    for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 3; c++) {
            double t = M2[r][c] * ca;
            t += M2[r][3] * (2 * g[c]);
            for (int k = 0; k < 3; k++) t += M2[r][5 + k] * Rb(k, c);
            JtJ(1 + r, 1 + c) = t;
        }
        for (int cc = 0; cc < 3; cc++) {
            double t = 0;
            for (int k = 0; k < 3; k++) t += M2[r][5 + k] * Gx(k, cc);
            JtJ(1 + r, 4 + cc) = t + M2[r][8 + cc];
        }
    }
    // JtF
    {
        double q[3], s1 = 0, s2 = 0;
        for (int j = 0; j < 3; j++) { double t = 0; for (int i = 0; i < 3; i++) t += (0.5 * F[i]) * D3(i, j); q[j] = t; }
        for (int j = 0; j < 3; j++) { s1 += q[j] * F[j]; s2 += w[j] * F[j]; }
        for (int jj = 0; jj < 7; jj++) s2 += W7(0, jj) * F[4 + jj];
        JtF[0] = s1 + s2;
    }
    for (int r = 0; r < 6; r++) {
        double t = 0;
        for (int j = 0; j < 11; j++) t += M2[r][j] * F[j];
        JtF[1 + r] = t;
    }
}

REBVO_HD inline double saturate(double t, double limit) { return t > limit ? limit : (t < -limit ? -limit : t); }

// FunT_KaGMEKBias (scaleestimator.cpp:201-204): wrap the angle, clamp the bias to +-0.02
REBVO_HD inline Vec<7> funT_KaGMEKBias(const Vec<7> &x) {
    Vec<7> r;
    r[0] = std::atan2(std::sin(x[0]), std::cos(x[0]));
    r[1] = x[1]; r[2] = x[2]; r[3] = x[3];
    for (int i = 4; i < 7; i++) r[i] = saturate(x[i], 5e-1 / 25);
    return r;
}

}  // namespace imufilter_detail

// scaleestimator.cpp:208-318
REBVO_HD inline double ScaleEstimator::estKaGMEKBias(const Vec<3> &s_acel, const Vec<3> &f_acel, double kP, Mat<3, 3> Rot, Vec<7> &X, Mat<7, 7> &P,
                                     const Mat<3, 3> &Qg, const Mat<3, 3> &Qrot, const Mat<3, 3> &Qbias, const double &QKp,
                                     const double &Rg, const Mat<3, 3> &Rs, const Mat<3, 3> &Rf, Vec<3> &g_est, Vec<3> &b_est,
                                     const Mat<6, 6> &Wvw, Vec<6> &Xvw, double g_gravit) {
    // linear prior
    Mat<7, 7> F = Mat<7, 7>::zeros();
    F(0, 0) = kP;
    la::set_block(F, 1, 1, la::transpose(Rot));
    la::set_block(F, 4, 4, Mat<3, 3>::identity());
    const Vec<3> Gtmp = la::slice<3>(X, 1);
    Mat<3, 3> GProd;
    GProd(0, 0) = 0; GProd(0, 1) = Gtmp[2]; GProd(0, 2) = -Gtmp[1];
    GProd(1, 0) = -Gtmp[2]; GProd(1, 1) = 0; GProd(1, 2) = Gtmp[0];
    GProd(2, 0) = Gtmp[1]; GProd(2, 1) = -Gtmp[0]; GProd(2, 2) = 0;
    Mat<7, 7> Q = Mat<7, 7>::zeros();
    { const double tn = std::tan(X[0]); Q(0, 0) = QKp / (1 + tn * tn); }
    la::set_block(Q, 1, 1, (la::transpose(GProd) * Qrot) * GProd + Qg);
    la::set_block(Q, 4, 4, Qbias);
    X = F * X;
    const Mat<7, 7> Pp = (F * P) * la::transpose(F) + Q;

    // non-linear posterior: Minimizer<7,11,...>::GaussNewton with a_tol = r_tol = 0 runs all 20 iterations
    imufilter_detail::KaGMEKBiasParams params;
    params.a_s = s_acel;
    params.a_v = f_acel;
    params.Rs = Rs;
    params.Rv = Rf;
    params.Pp = Pp;
    params.W7 = la::Cholesky<7>(Pp).inverse();
    params.Rg = Rg;
    params.G = g_gravit;
    params.x_p = X;
    Mat<7, 7> JtJ;
    Vec<7> JtF;
    // (the eigen-solve of every iteration but the first starts from the eigenvectors of the one before: la::SymSVD)
    Mat<7, 7> Vprev;
    for (int it = 0; it < 20; it++) {
        imufilter_detail::problem_KaGMEKBias(JtJ, JtF, X, params);
        const la::SymSVD<7> svd(JtJ, 1e9, it ? &Vprev : nullptr);
        const Vec<7> h = svd.backsub(-JtF);
        Vprev = svd.V;
        X = X + h;
        X = imufilter_detail::funT_KaGMEKBias(X);
    }
    imufilter_detail::problem_KaGMEKBias(JtJ, JtF, X, params);
    P = la::Cholesky<7>(JtJ).inverse();
    double k = std::tan(X[0]);
    if (k < 0 || std::isnan(k) || std::isinf(k)) k = 0;
    g_est = la::slice<3>(X, 1);
    b_est = la::slice<3>(X, 4);

    // correct the visual measurement with the bias estimate
    const Mat<3, 3> WVBias = la::block<3, 3>(JtJ, 4, 4);
    Mat<6, 6> Wb = Mat<6, 6>::zeros();
    la::set_block(Wb, 3, 3, WVBias);
    const Vec<3> wc = la::slice<3>(Xvw, 3) - b_est;
    Vec<6> WXc = Vec<6>::zeros();
    la::set_slice(WXc, 3, WVBias * wc);
    Xvw = la::Cholesky<6>(Wb + Wvw).inverse() * (Wvw * Xvw + WXc);
    return k;
}


}  // namespace rebvo
#endif
