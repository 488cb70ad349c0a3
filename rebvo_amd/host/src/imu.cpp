// imu.cpp — ImuGrabber, BiasCorrect and the ScaleEstimator filters (see rebvo/imu.h for the reference lines).
// Host-only scalar code: nothing here touches the GPU.  Expression order follows the reference's TooN expressions
// (left-to-right products with temporaries) so that results agree to rounding.
#include "rebvo/imu.h"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

namespace rebvo {

using la::Mat;
using la::Vec;

// ------------------------------------------------------------------------------------------------------------
// ImuGrabber
// ------------------------------------------------------------------------------------------------------------
// imugrabber.cpp:34-40: both indexers start at 0 over list_size+1 slots, then --read_inx
ImuGrabber::ImuGrabber(int list_size, double tsamp)
    : imu(list_size + 1), write_inx(0), read_inx(0), size(list_size + 1), tsample(tsamp) {
    read_inx = prev(read_inx);
}

// imugrabber.cpp:46-66
ImuGrabber::ImuGrabber(const std::vector<ImuData> &data_set_data)
    : imu(data_set_data.size() + 1), write_inx(0), read_inx(0), size((int)data_set_data.size() + 1), tsample(0) {
    read_inx = prev(read_inx);
    for (int i = 0; i < size - 1; i++) {
        imu[i] = data_set_data[i];
        write_inx = next(write_inx);
    }
    if (data_set_data.size() > 1) {   // sample time = mean time-stamp spacing
        tsample = (data_set_data[data_set_data.size() - 1].tstamp - data_set_data[0].tstamp) / (data_set_data.size() - 1);
        std::cout << "\nImuGraber: tsample:" << tsample << "\n";
    } else {
        tsample = 0;
        std::cout << "\nImuGraber: coud not set tsample\n";
    }
}

static std::string shrink_ws(const std::string &s) {   // Configurator::ShrinkWS (configurator.cpp:33-48)
    size_t p1 = 0;
    while (p1 < s.size() && (s[p1] == ' ' || s[p1] == 0x09)) p1++;
    if (p1 == s.size()) return std::string();
    size_t p2 = s.size() - 1;
    while (p2 > p1 && (s[p2] == ' ' || s[p2] == 0x09)) p2--;
    return s.substr(p1, p2 - p1 + 1);
}

// imugrabber.cpp:80-131
std::vector<ImuData> ImuGrabber::LoadDataSet(const char *data_file, bool comp_data, double time_scale, bool &error) {
    std::ifstream ifile(data_file);
    error = false;
    if (!ifile.is_open()) {
        std::cout << "\nImuGrabber: Failed to open file " << data_file << "\n";
        error = true;
        return std::vector<ImuData>();
    }
    std::vector<ImuData> vector_data;
    int lines = 0;
    while (!ifile.eof()) {
        std::string line;
        std::getline(ifile, line);
        line = shrink_ws(line);
        if (line.size() == 0) continue;
        if (line[0] == '#') continue;
        ImuData d;
        if (comp_data)
            std::sscanf(line.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", &d.tstamp, &d.giro[0], &d.giro[1], &d.giro[2], &d.acel[0],
                        &d.acel[1], &d.acel[2], &d.comp[0], &d.comp[1], &d.comp[2]);
        else
            std::sscanf(line.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf", &d.tstamp, &d.giro[0], &d.giro[1], &d.giro[2], &d.acel[0],
                        &d.acel[1], &d.acel[2]);
        d.tstamp *= time_scale;
        lines++;
        vector_data.push_back(d);
    }
    std::cout << "\nImugrabber: Loaded " << lines << " datums\n";
    return vector_data;
}

// imugrabber.cpp:133-158
bool ImuGrabber::LoadCamImuSE3(const char *se3_file) {
    std::ifstream file(se3_file);
    if (!file.is_open()) {
        std::cout << "LoadCamImuSE3: could not open SE3 file" << se3_file << " \n";
        return false;
    }
    std::string num;
    try {
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                std::getline(file, num, ',');
                RDataSetCam2IMU(i, j) = std::stod(num);
            }
            std::getline(file, num, ',');
            TDataSetCam2IMU[i] = std::stod(num);
        }
    } catch (const std::exception &) {   // the reference lets std::stod throw out of the constructor chain
        std::cout << "LoadCamImuSE3: malformed SE3 file " << se3_file << "\n";
        return false;
    }
    return true;
}

bool ImuGrabber::LoadCamImuSE3(const Mat<3, 3> &RCam2IMU, const Vec<3> &TCam2IMU) {
    RDataSetCam2IMU = RCam2IMU;
    TDataSetCam2IMU = TCam2IMU;
    return true;
}

// imugrabber.cpp:175-212: [begin, end) of the samples with tstart < t <= tend (end is exclusive); an empty range
// when the buffer does not yet reach past tend
std::pair<int, int> ImuGrabber::SeachByTimeStamp(double tstart, double tend) {
    std::lock_guard<std::mutex> locker(rw_mut);
    int inx = read_inx;
    do {
        inx = next(inx);
        if (inx == write_inx) return std::pair<int, int>(inx, inx);
    } while (imu[inx].tstamp <= tstart);
    const int begin = inx;
    do {
        inx = next(inx);
        if (inx == write_inx) return std::pair<int, int>(begin, begin);
    } while (imu[inx].tstamp < tend);
    if (imu[inx].tstamp - tend < 1e-12) inx = next(inx);   // a sample exactly at tend is included
    return std::pair<int, int>(begin, inx);
}

// imugrabber.cpp:219-252
IntegratedImuData ImuGrabber::GrabAndIntegrate(double tstart, double tend) {
    const std::pair<int, int> range = SeachByTimeStamp(tstart, tend);
    const Mat<3, 3> Rt = la::transpose(RDataSetCam2IMU);
    IntegratedImuData d;
    for (int inx = range.first; inx != range.second; inx = next(inx)) {
        d.giro = d.giro + Rt * imu[inx].giro;
        d.acel = d.acel + Rt * imu[inx].acel;
        d.comp = d.comp + Rt * imu[inx].comp;
        d.Rot = d.Rot * la::so3_exp((Rt * imu[inx].giro) * tsample);   // integrate the rotation on SO(3)
        d.n++;
    }
    d.dt = d.n * tsample;
    if (d.n > 1) {
        d.giro = d.giro / (double)d.n;
        d.acel = d.acel / (double)d.n;
        d.comp = d.comp / (double)d.n;
        d.dgiro = Rt * (imu[prev(range.second)].giro - imu[range.first].giro);   // angular acceleration, finite difference
        d.dgiro = d.dgiro / d.dt;
    }
    d.cacel = d.acel + la::cross(d.dgiro, -(Rt * TDataSetCam2IMU));   // tangential acceleration of the lever arm
    {
        std::lock_guard<std::mutex> locker(rw_mut);
        read_inx = prev(range.second);   // release the samples read
    }
    return d;
}

// imugrabber.cpp:258-269
bool ImuGrabber::PushData(const ImuData &data) {
    std::lock_guard<std::mutex> locker(rw_mut);
    if (write_inx == read_inx) throw std::overflow_error("ImuGrabber circular buffer full");
    imu[write_inx] = data;
    write_inx = next(write_inx);
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// BiasCorrect (edge_tracker.cpp:1308-1343)
// ------------------------------------------------------------------------------------------------------------
void imufilter::BiasCorrect(Vec<6> &X, Mat<6, 6> &Wx, Vec<3> &Gb, Mat<3, 3> &Wb, const Mat<3, 3> &Rg, const Mat<3, 3> &Rb) {
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

// ------------------------------------------------------------------------------------------------------------
// ScaleEstimator
// ------------------------------------------------------------------------------------------------------------
// scaleestimator.cpp:38-92
void ScaleEstimator::EstAcelLsq4(const Vec<3> &vel, Vec<3> &acel, const Mat<3, 3> &R, const double &dt) {
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
void ScaleEstimator::MeanAcel4(const Vec<3> &s_acel, Vec<3> &acel, const Mat<3, 3> &R) {
    const Mat<3, 3> Rt = la::transpose(R);
    A2 = Rt * A1;
    A1 = Rt * A0;
    A0 = Rt * A;
    A = s_acel;
    acel = (((A + A0) + A1) + A2) / 4.0;
}

namespace {

struct KaGMEKBiasParams {   // FunParams_KaGMEKBias (scaleestimator.cpp:115-124)
    Vec<3> a_v, a_s;
    double G;
    Vec<7> x_p;
    Mat<3, 3> Rv, Rs;
    double Rg;
    Mat<7, 7> Pp;
};

// Problem_KaGMEKBias (scaleestimator.cpp:126-199): normal equations of the 11-row residual whose weight depends on
// the scale angle a = x[0] (hence the dW/da terms)
void problem_KaGMEKBias(Mat<7, 7> &JtJ, Vec<7> &JtF, const Vec<7> &x, const KaGMEKBiasParams &p) {
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

inline double saturate(double t, double limit) { return t > limit ? limit : (t < -limit ? -limit : t); }

// FunT_KaGMEKBias (scaleestimator.cpp:201-204): wrap the angle, clamp the bias to +-0.02
Vec<7> funT_KaGMEKBias(const Vec<7> &x) {
    Vec<7> r;
    r[0] = std::atan2(std::sin(x[0]), std::cos(x[0]));
    r[1] = x[1]; r[2] = x[2]; r[3] = x[3];
    for (int i = 4; i < 7; i++) r[i] = saturate(x[i], 5e-1 / 25);
    return r;
}

}  // namespace

// scaleestimator.cpp:208-318
double ScaleEstimator::estKaGMEKBias(const Vec<3> &s_acel, const Vec<3> &f_acel, double kP, Mat<3, 3> Rot, Vec<7> &X, Mat<7, 7> &P,
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
    KaGMEKBiasParams params;
    params.a_s = s_acel;
    params.a_v = f_acel;
    params.Rs = Rs;
    params.Rv = Rf;
    params.Pp = Pp;
    params.Rg = Rg;
    params.G = g_gravit;
    params.x_p = X;
    Mat<7, 7> JtJ;
    Vec<7> JtF;
    for (int it = 0; it < 20; it++) {
        problem_KaGMEKBias(JtJ, JtF, X, params);
        const Vec<7> h = la::SymSVD<7>(JtJ).backsub(-JtF);
        X = X + h;
        X = funT_KaGMEKBias(X);
    }
    problem_KaGMEKBias(JtJ, JtF, X, params);
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
