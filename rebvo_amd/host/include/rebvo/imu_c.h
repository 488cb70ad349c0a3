/* imu_c.h — flat C view of the host-side IMU pieces (rebvo/imu.h) for callers that are not C++: the parity tests load
 * librebvohost.so through ctypes and compare these against the reference's own code.  Matrices are row-major. */
#ifndef REBVO_AMD_HOST_IMU_C_H
#define REBVO_AMD_HOST_IMU_C_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rebvo_imu_integrated {   /* rebvo::IntegratedImuData */
    int n, pad;
    double dt, Rot[9], giro[3], acel[3], comp[3], dgiro[3], cacel[3];
} rebvo_imu_integrated;

/* edge_tracker::BiasCorrect: X[6], Wx[36], Gb[3], Wb[9] in/out */
void rebvo_imu_bias_correct(double *X, double *Wx, double *Gb, double *Wb, const double *Rg, const double *Rb);

/* ScaleEstimator with its sample histories */
void *rebvo_scale_estimator_new(void);
void rebvo_scale_estimator_free(void *se);
void rebvo_est_acel_lsq4(void *se, const double *vel, double *acel /* in/out */, const double *R, double dt);
void rebvo_mean_acel4(void *se, const double *s_acel, double *acel, const double *R);
/* test hook: problem_KaGMEKBias as the reference writes it (dense) and as the filters run it (structural zeros left out) */
void rebvo_problem_ka_gmek_bias(const double *x /*[7]*/, const double *a_v, const double *a_s, double G, const double *x_p /*[7]*/,
                                const double *Rv, const double *Rs, double Rg, const double *Pp /*[49]*/, double *JtJ_dense /*[49]*/,
                                double *JtF_dense /*[7]*/, double *JtJ_sparse, double *JtF_sparse);
double rebvo_est_ka_gmek_bias(const double *s_acel, const double *f_acel, double kP, const double *Rot, double *X /*[7] io*/,
                              double *P /*[49] io*/, const double *Qg, const double *Qrot, const double *Qbias, double QKp,
                              double Rg, const double *Rs, const double *Rf, double *g_est, double *b_est, const double *Wvw,
                              double *Xvw /*[6] io*/, double g_gravit);

/* ImuGrabber */
void *rebvo_imu_grabber_new(int list_size, double tsamp);
void *rebvo_imu_grabber_load(const char *csv_file, double time_scale);   /* NULL when the file cannot be read */
void rebvo_imu_grabber_free(void *g);
int rebvo_imu_grabber_set_se3(void *g, const double *RCam2IMU, const double *TCam2IMU);
int rebvo_imu_grabber_load_se3(void *g, const char *se3_file);
int rebvo_imu_grabber_push(void *g, double tstamp, const double *giro, const double *acel);   /* 1, or -1 = buffer full */
void rebvo_imu_grabber_grab(void *g, double tstart, double tend, rebvo_imu_integrated *out);
double rebvo_imu_grabber_tsample(void *g);

#ifdef __cplusplus
}
#endif
#endif
