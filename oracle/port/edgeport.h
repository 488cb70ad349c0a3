// edgeport.h — shared declarations of the CPU restatement (TEST INFRASTRUCTURE ONLY; see ../oracle_abi.h).
#pragma once
#include <cstdint>
#include <vector>

#include "oracle_abi.h"

namespace port {

constexpr int kBoxes = 3;                                    // bf_num passed by rebvo.cpp:299
constexpr double kRhoMax = 20.0, kRhoMin = 1e-3, kRhoInit = 1.0;   // include/mtracklib/edge_finder.h:38-40

struct Filter {              // iigauss: box widths + one reciprocal-count image per box
    int box_d[kBoxes];
    double sigma_r;
    std::vector<float> div[kBoxes];
};

struct UndistPoint {         // image_undistort::undistMapPoint (include/VideoLib/image_undistort.h:41-47)
    int num = 0;
    int inx[4] = {0, 0, 0, 0};
    float w[4] = {0, 0, 0, 0};
    int iw[4] = {0, 0, 0, 0};
};

struct Slot {                // one PipeBuffer slot: sspace + edge_tracker + global_tracker (rebvo.cpp:297-312)
    std::vector<uint8_t> imgc;
    std::vector<float> bw, img0, img1, dog, dx, dy;
    std::vector<int32_t> mask;                // img_mask_kl
    std::vector<OrcKeyLine> kl;
    int kn = 0;
    float retuned = 0;                        // reTunedThresh
    int nmatch = 0;
    // global_tracker
    std::vector<int32_t> field;               // {dist, ikl} per pixel (gt_field_data, global_tracker.h:33-36)
    int field_slot = -1;                      // klist_f: the slot whose KeyLines the field indexes
    int max_r = 0;
    unsigned FrameCount = 0;
};

struct Ctx {
    OrcParams p;
    float ppx, ppy, zfx, zfy;                 // cam_model keeps pp / zf as float (cam_model.h:51-52)
    double zfm;                               // (zf.x+zf.y)/2 evaluated in float, stored as double (:57)
    Filter filter[2];
    std::vector<float> integral;              // iimage::img_data (scratch shared by both filters)
    std::vector<double> pinv;                 // 3 x 25 plane-fit pseudo inverse
    std::vector<UndistPoint> umap;
    std::vector<Slot> slots;
    // FirstThr / SecondThread locals that persist from frame to frame
    double tresh;
    int l_kl_num;
    int frame;
    double t_prev, Kp, K, P_Kp;
    double V[3], W[3], Pos[3], Pose[9];
};

// stage A (edgeport_a.cpp)
double kovesi_boxes(double sigma, int box_num, int *box_d);
void build_average(int d, int w, int h, std::vector<float> &div);
void plane_fit_pinv(int win_s, std::vector<double> &pinv);
void build_undistort_map(Ctx &c);
int stage_a(Ctx &c, int slot, const uint8_t *rgb24, double *tresh_io, int *l_kl_num_io);

// stage B (edgeport_b.cpp)
double estimate_quantile(const Slot &s, double s_rho_min, double s_rho_max, double percentile, int n);
void build_field(Ctx &c, int slot, int radius, float min_mod);
double try_velrot(Ctx &c, Slot &gt, Slot &klist, bool ReWeight, bool ProcJF, double JtJ[36], double JtF[6], const double VelRot[6],
                  const double *P0m, int pnum, double match_thresh, double s_rho_min, unsigned MatchNumThresh, double k_huber,
                  const double *DResidual, double *DResidualNew);
void kl_to_p0(const Ctx &c, const Slot &klist, int pnum, std::vector<double> &P0m);
double minimizer_rv(Ctx &c, Slot &gt, Slot &klist, double Vel[3], double W0[3], double RVel[9], double RW0[9], double match_thresh,
                    int iter_max, int init_type, double reweigth_distance, double &rel_error, double &rel_error_score,
                    double max_s_rho, unsigned MatchNumThresh, double init_iter, double W_X[36]);

double minimizer_v(Ctx &c, Slot &gt, Slot &klist, double Vel[3], double RVel[9], double match_thresh, int iter_max,
                   double s_rho_min, unsigned MatchNumThresh, double reweigth_distance, float min_mod);

// stage C (edgeport_c.cpp)
int forward_match(Slot &from, Slot &et);
void rotate_keylines(const Ctx &c, Slot &s, const double RotF[9]);
int directed_matching(const Ctx &c, Slot &s, const double Vel[3], const double RVel[9], const double BackRot[9], Slot &et0,
                      int &kf_matchs, double min_thr_mod, double min_thr_ang, double max_radius, double loc_uncertainty);
int regularize_1_iter(Slot &s, double thresh);
void update_inverse_depth_kalman(const Ctx &c, Slot &s, const double vel[3], double ReshapeQAbsolute, double LocationUncertainty);
bool ext_rot_vel(const Ctx &c, Slot &s, const double vel[3], double Wx[36], double Rx[36], double X[6], double LocUncert, double HubReweigth);
double estimate_rescaling_opt(Slot &s, double &RKp, double s_rho_min, unsigned MatchNumMin, bool re_escale);

}  // namespace port
