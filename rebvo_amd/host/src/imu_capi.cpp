// imu_capi.cpp — the flat C view declared in rebvo/imu_c.h.
#include <cstring>

#include "rebvo/imu.h"
#include "rebvo/imu_c.h"

using namespace rebvo;
using la::Mat;
using la::Vec;

namespace {
template <int N> Vec<N> vin(const double *p) { Vec<N> v; std::memcpy(v.v, p, sizeof v.v); return v; }
template <int R, int C> Mat<R, C> min_(const double *p) { Mat<R, C> m; std::memcpy(m.a, p, sizeof m.a); return m; }
template <int N> void vout(double *p, const Vec<N> &v) { std::memcpy(p, v.v, sizeof v.v); }
template <int R, int C> void mout(double *p, const Mat<R, C> &m) { std::memcpy(p, m.a, sizeof m.a); }
}  // namespace

extern "C" {

void rebvo_imu_bias_correct(double *X, double *Wx, double *Gb, double *Wb, const double *Rg, const double *Rb) {
    Vec<6> x = vin<6>(X);
    Mat<6, 6> wx = min_<6, 6>(Wx);
    Vec<3> gb = vin<3>(Gb);
    Mat<3, 3> wb = min_<3, 3>(Wb);
    imufilter::BiasCorrect(x, wx, gb, wb, min_<3, 3>(Rg), min_<3, 3>(Rb));
    vout(X, x); mout(Wx, wx); vout(Gb, gb); mout(Wb, wb);
}

void *rebvo_scale_estimator_new(void) { return new ScaleEstimator; }
void rebvo_scale_estimator_free(void *se) { delete (ScaleEstimator *)se; }
void rebvo_est_acel_lsq4(void *se, const double *vel, double *acel, const double *R, double dt) {
    Vec<3> a = vin<3>(acel);
    ((ScaleEstimator *)se)->EstAcelLsq4(vin<3>(vel), a, min_<3, 3>(R), dt);
    vout(acel, a);
}
void rebvo_mean_acel4(void *se, const double *s_acel, double *acel, const double *R) {
    Vec<3> a = vin<3>(acel);
    ((ScaleEstimator *)se)->MeanAcel4(vin<3>(s_acel), a, min_<3, 3>(R));
    vout(acel, a);
}
double rebvo_est_ka_gmek_bias(const double *s_acel, const double *f_acel, double kP, const double *Rot, double *X, double *P,
                              const double *Qg, const double *Qrot, const double *Qbias, double QKp, double Rg, const double *Rs,
                              const double *Rf, double *g_est, double *b_est, const double *Wvw, double *Xvw, double g_gravit) {
    Vec<7> x = vin<7>(X);
    Mat<7, 7> p = min_<7, 7>(P);
    Vec<3> g = vin<3>(g_est), b = vin<3>(b_est);
    Vec<6> xvw = vin<6>(Xvw);
    const double k = ScaleEstimator::estKaGMEKBias(vin<3>(s_acel), vin<3>(f_acel), kP, min_<3, 3>(Rot), x, p, min_<3, 3>(Qg),
                                                   min_<3, 3>(Qrot), min_<3, 3>(Qbias), QKp, Rg, min_<3, 3>(Rs), min_<3, 3>(Rf), g, b,
                                                   min_<6, 6>(Wvw), xvw, g_gravit);
    vout(X, x); mout(P, p); vout(g_est, g); vout(b_est, b); vout(Xvw, xvw);
    return k;
}

// problem_KaGMEKBias both ways (the reference's dense algebra / the structural zeros left out): JtJ (49), JtF (7) of each
void rebvo_problem_ka_gmek_bias(const double *x, const double *a_v, const double *a_s, double G, const double *x_p, const double *Rv,
                                const double *Rs, double Rg, const double *Pp, double *JtJ_dense, double *JtF_dense, double *JtJ_sparse,
                                double *JtF_sparse) {
    rebvo::imufilter_detail::KaGMEKBiasParams p;
    p.a_v = vin<3>(a_v); p.a_s = vin<3>(a_s); p.G = G; p.x_p = vin<7>(x_p); p.Rv = min_<3, 3>(Rv); p.Rs = min_<3, 3>(Rs); p.Rg = Rg;
    p.Pp = min_<7, 7>(Pp);
    p.W7 = rebvo::la::Cholesky<7>(p.Pp).inverse();
    Mat<7, 7> J;
    Vec<7> F;
    rebvo::imufilter_detail::problem_KaGMEKBias_dense(J, F, vin<7>(x), p);
    mout(JtJ_dense, J); vout(JtF_dense, F);
    rebvo::imufilter_detail::problem_KaGMEKBias(J, F, vin<7>(x), p);
    mout(JtJ_sparse, J); vout(JtF_sparse, F);
}

void *rebvo_imu_grabber_new(int list_size, double tsamp) { return new ImuGrabber(list_size, tsamp); }
void *rebvo_imu_grabber_load(const char *csv_file, double time_scale) {
    bool error = false;
    std::vector<ImuData> d = ImuGrabber::LoadDataSet(csv_file, false, time_scale, error);
    if (error) return nullptr;
    return new ImuGrabber(d);
}
void rebvo_imu_grabber_free(void *g) { delete (ImuGrabber *)g; }
int rebvo_imu_grabber_set_se3(void *g, const double *R, const double *T) {
    return ((ImuGrabber *)g)->LoadCamImuSE3(min_<3, 3>(R), vin<3>(T)) ? 1 : 0;
}
int rebvo_imu_grabber_load_se3(void *g, const char *se3_file) { return ((ImuGrabber *)g)->LoadCamImuSE3(se3_file) ? 1 : 0; }
int rebvo_imu_grabber_push(void *g, double tstamp, const double *giro, const double *acel) {
    try {
        return ((ImuGrabber *)g)->PushData(ImuData(tstamp, vin<3>(giro), vin<3>(acel))) ? 1 : 0;
    } catch (const std::overflow_error &) {
        return -1;
    }
}
void rebvo_imu_grabber_grab(void *g, double tstart, double tend, rebvo_imu_integrated *out) {
    const IntegratedImuData d = ((ImuGrabber *)g)->GrabAndIntegrate(tstart, tend);
    out->n = d.n; out->pad = 0; out->dt = d.dt;
    mout(out->Rot, d.Rot); vout(out->giro, d.giro); vout(out->acel, d.acel); vout(out->comp, d.comp);
    vout(out->dgiro, d.dgiro); vout(out->cacel, d.cacel);
}
double rebvo_imu_grabber_tsample(void *g) { return ((ImuGrabber *)g)->SampleTime(); }

}  // extern "C"
