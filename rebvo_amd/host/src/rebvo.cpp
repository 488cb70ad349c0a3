// rebvo.cpp — host runtime behind the rebvo::REBVO surface: config parsing, the custom-camera ring, one
// tracking thread that drives libedgehip (what FirstThr + SecondThread do in the reference,
// src/rebvo/rebvo_first_t.cpp:87-337, src/rebvo/rebvo_second_t.cpp:43-636) and the output thread
// (src/rebvo/rebvo_third_t.cpp:48-410: .m log, TUM trajectory, user callback).
//
// Hand-off order mirrors the reference's 4-player ring: a frame's PipeBuffer reaches the output thread only
// after the NEXT frame has been tracked against it (the reference releases `old_buf` to player 3,
// rebvo_second_t.cpp:622-623), so the KeyLines a callback sees are the previous edge map after
// rotate_keylines/FordwardMatch touched it, and the last frame of a run is never delivered.

#include "rebvo/rebvo.h"

#include <pthread.h>
#include <sched.h>

#include <chrono>
#include <cstdio>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>

#include "edgehip.h"
#include "rebvo/datasetcam.h"
#include "rebvo_internal.h"

namespace rebvo {

// ---- configuration file: "&Section", "name = value", "//" comments (src/UtilLib/configurator.cpp:82-160) --
namespace {

std::string shrink_ws(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

class Configurator {
    std::map<std::string, std::map<std::string, std::string>> sections;

public:
    bool ParseConfigFile(const char *filename) {
        std::ifstream file(filename);
        if (!file.is_open()) {
            std::cout << "\nConfigurator: cannot open the configuration file " << filename << "\n";
            return false;
        }
        std::string s, current;
        int line = 0;
        while (std::getline(file, s)) {
            line++;
            const size_t c = s.find("//");
            if (c != std::string::npos) s.resize(c);
            s = shrink_ws(s);
            if (s.empty()) continue;
            if (s[0] == '&') { current = shrink_ws(s.substr(1)); sections[current]; continue; }
            const size_t pos = s.find('=');
            if (pos == 0) { std::cout << "Configurator: syntax error line " << line << ", empty name\n"; return false; }
            if (pos == std::string::npos) { std::cout << "Configurator: syntax error line " << line << ", use name = value\n"; return false; }
            const std::string name = shrink_ws(s.substr(0, pos));
            // first definition wins, like the reference's linear search over the parameter list
            sections[current].emplace(name, s.substr(pos + 1));
        }
        return true;
    }
    bool raw(const char *sec, const char *name, std::string &out, bool complain) const {
        auto s = sections.find(sec);
        if (s == sections.end()) {
            if (complain) std::cout << "\nConfigurator: error, section " << sec << " not found in the configuration file\n";
            return false;
        }
        auto p = s->second.find(name);
        if (p == s->second.end()) {
            if (complain) std::cout << "\nConfigurator: error, parameter " << name << " not found in section " << sec << "\n";
            return false;
        }
        out = p->second;
        return true;
    }
    template <typename T>
    bool get(const char *sec, const char *name, T &param, bool complain = true) const {
        std::string v;
        if (!raw(sec, name, v, complain)) return false;
        std::istringstream iss(v);
        iss >> param;   // trailing characters ("100;") are ignored, as with the reference's stream extraction
        return !iss.fail();
    }
    bool get(const char *sec, const char *name, std::string &param, bool complain = true) const {
        std::string v;
        if (!raw(sec, name, v, complain)) return false;
        param = shrink_ws(v);
        return true;
    }
};

// util::LieRot2Quaternion (include/UtilLib/toon_util.h:63-72)
void lie2quat(const Vector3 &W, double q[4]) {
    const double angle = std::sqrt(W[0] * W[0] + W[1] * W[1] + W[2] * W[2]);
    for (int i = 0; i < 3; i++) q[i] = angle > 0 ? W[i] / angle * std::sin(angle / 2) : 0.0;
    q[3] = std::cos(angle / 2);
}

}  // namespace

namespace detail {
double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool set_affinity(int cpu) {
    if (cpu < 0 || cpu >= CPU_SETSIZE) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    return pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &set) == 0;
}

void fill_hip_params(const REBVOParameters &p, edgehip_params &h) {
    std::memset(&h, 0, sizeof h);
    h.w = (int)p.ImageSize.w; h.h = (int)p.ImageSize.h;
    h.ppx = p.pp_x; h.ppy = p.pp_y; h.zfx = p.z_f_x; h.zfy = p.z_f_y;
    h.kc[0] = p.kc.Kc2; h.kc[1] = p.kc.Kc4; h.kc[2] = p.kc.Kc6; h.kc[3] = p.kc.P1; h.kc[4] = p.kc.P2;
    h.sigma0 = p.Sigma0; h.ksigma = p.KSigma;
    h.plane_fit_size = p.DetectorPlaneFitSize;
    h.pos_neg_thresh = p.DetectorPosNegThresh; h.dog_thresh = p.DetectorDoGThresh;
    h.max_points = p.MaxPoints; h.reference_points = p.ReferencePoints; h.track_points = p.TrackPoints;
    h.detector_thresh = p.DetectorThresh; h.auto_gain = p.DetectorAutoGain;
    h.max_thresh = p.DetectorMaxThresh; h.min_thresh = p.DetectorMinThresh;
    h.search_range = (int)p.SearchRange; h.qcut_nbins = (int)p.QCutOffNumBins; h.qcut_quantile = p.QCutOffQuantile;
    h.tracker_iter_num = p.TrackerIterNum; h.tracker_init_type = p.TrackerInitType;
    h.tracker_init_iter_num = p.TrackerInitIterNum;
    h.tracker_match_thresh = p.TrackerMatchThresh; h.match_thresh_module = p.MatchThreshModule;
    h.match_thresh_angle = p.MatchThreshAngle; h.match_num_thresh = p.MatchNumThresh;
    h.do_rescaling = p.DoReScaling > 0 ? 1 : 0;
    h.reweight_distance = p.ReweigthDistance; h.regularize_thresh = p.RegularizeThresh;
    h.loc_unc_match = p.LocationUncertaintyMatch; h.reshape_q_abs = p.ReshapeQAbsolute;
    h.reshape_q_rel = p.ReshapeQRelative; h.loc_unc = p.LocationUncertainty;
    h.global_match_threshold = p.MatchThreshold;
    h.config_fps = p.config_fps;
    h.use_undistort = p.useUndistort ? 1 : 0;
}

void fill_nav(const edgehip_nav &n, NavData &nav) {
    nav.t = n.t; nav.dt = n.dt; nav.scale = 1;
    for (int i = 0; i < 3; i++) {
        nav.RotLie[i] = n.RotLie[i]; nav.Vel[i] = n.Vel[i]; nav.PoseLie[i] = n.PoseLie[i]; nav.Pos[i] = n.Pos[i];
        nav.RotGiro[i] = 0; nav.g[i] = 0;
        for (int j = 0; j < 3; j++) { nav.Rot(i, j) = n.Rot[i * 3 + j]; nav.Pose(i, j) = n.Pose[i * 3 + j]; }
    }
}
void fill_nav_imu(const edgehip_nav_imu &n, PipeBuffer &pb) {
    NavData &nav = pb.nav;
    nav.dt = n.dt; nav.scale = n.scale;
    for (int i = 0; i < 3; i++) {
        nav.RotLie[i] = n.RotLie[i]; nav.RotGiro[i] = n.RotGiro[i]; nav.Vel[i] = n.Vel[i]; nav.PoseLie[i] = n.PoseLie[i]; nav.Pos[i] = n.Pos[i];
        nav.g[i] = n.g[i];
        for (int j = 0; j < 3; j++) { nav.Rot(i, j) = n.Rot[i * 3 + j]; nav.Pose(i, j) = n.Pose[i * 3 + j]; }
    }
    pb.dt = n.dt;
    pb.K = n.K; pb.Kp = n.Kp; pb.RKp = n.RKp;
    pb.s_rho_p = n.s_rho_q;
    pb.EstimationOK = n.estimation_ok != 0;   // (the match count, edge_tracker::nmatch, is the caller's to set: it is a friend)
    IMUState &is = pb.imustate;
    for (int i = 0; i < 3; i++) {
        is.Vg[i] = n.Vg[i]; is.Bg[i] = n.Bg[i]; is.dVv[i] = n.dVv[i]; is.dWv[i] = n.dWv[i]; is.Vgv[i] = n.Vgv[i]; is.Vgva[i] = n.Vgva[i];
        is.Av[i] = n.Av[i]; is.As[i] = n.As[i]; is.b_est[i] = n.b_est[i]; is.u_est[i] = n.u_est[i]; is.g_est[i] = n.g[i];
    }
    for (int i = 0; i < 7; i++) is.X[i] = n.X[i];
    is.init = n.init != 0;
}
}  // namespace detail
using detail::fill_hip_params;
using detail::fill_nav;
using detail::now_s;

const double REBVO::kRCam2Pair[9] = {0.999997256477450, 0.002312067192420, 0.000376008102351,
                                     -0.002317135723285, 0.999898048506528, 0.014089835846697,
                                     -0.000343393120589, -0.014090668452670, 0.999900662638179};
const double REBVO::kTCam2Pair[3] = {-0.110073808127139, 0.000399121547014, -0.000853702503351};

// ---- construction -------------------------------------------------------------------------------------------
REBVO::REBVO(const char *configFile)
    : quit(true), pipe(CBUFSIZE, 3), system_reset(false), cam_pipe(CCAMBUFSIZE, 2), cam_pipe_stereo(CCAMBUFSIZE, 2), outputFunc(nullptr) {
    Configurator config;
    if (!(InitOK = config.ParseConfigFile(configFile))) return;
    REBVOParameters &p = params;
    // keys that reach this path are mandatory, exactly as in REBVO::REBVO (src/rebvo/rebvo.cpp:57-193)
    InitOK &= config.get("REBVO", "CameraType", p.CameraType);
    InitOK &= config.get("Camera", "ImageWidth", p.ImageSize.w);
    InitOK &= config.get("Camera", "ImageHeight", p.ImageSize.h);
    InitOK &= config.get("Camera", "ZfX", p.z_f_x);
    InitOK &= config.get("Camera", "ZfY", p.z_f_y);
    InitOK &= config.get("Camera", "PPx", p.pp_x);
    InitOK &= config.get("Camera", "PPy", p.pp_y);
    InitOK &= config.get("Camera", "KcR2", p.kc.Kc2);
    InitOK &= config.get("Camera", "KcR4", p.kc.Kc4);
    InitOK &= config.get("Camera", "KcR6", p.kc.Kc6);
    InitOK &= config.get("Camera", "KcP1", p.kc.P1);
    InitOK &= config.get("Camera", "KcP2", p.kc.P2);
    InitOK &= config.get("Camera", "UseUndistort", p.useUndistort);
    InitOK &= config.get("Camera", "FPS", p.config_fps);
    if (!config.get("Camera", "SoftFPS", p.soft_fps, false)) p.soft_fps = p.config_fps;
    InitOK &= config.get("Detector", "Sigma0", p.Sigma0);
    InitOK &= config.get("Detector", "KSigma", p.KSigma);
    InitOK &= config.get("Detector", "ReferencePoints", p.ReferencePoints);
    InitOK &= config.get("Detector", "MaxPoints", p.MaxPoints);
    InitOK &= config.get("Detector", "TrackPoints", p.TrackPoints);
    InitOK &= config.get("Detector", "DetectorThresh", p.DetectorThresh);
    InitOK &= config.get("Detector", "DetectorAutoGain", p.DetectorAutoGain);
    InitOK &= config.get("Detector", "DetectorMaxThresh", p.DetectorMaxThresh);
    InitOK &= config.get("Detector", "DetectorMinThresh", p.DetectorMinThresh);
    InitOK &= config.get("Detector", "DetectorPlaneFitSize", p.DetectorPlaneFitSize);
    InitOK &= config.get("Detector", "DetectorPosNegThresh", p.DetectorPosNegThresh);
    InitOK &= config.get("Detector", "DetectorDoGThresh", p.DetectorDoGThresh);
    InitOK &= config.get("TrackMaper", "SearchRange", p.SearchRange);
    InitOK &= config.get("TrackMaper", "QCutOffNumBins", p.QCutOffNumBins);
    InitOK &= config.get("TrackMaper", "QCutOffQuantile", p.QCutOffQuantile);
    InitOK &= config.get("TrackMaper", "TrackerIterNum", p.TrackerIterNum);
    InitOK &= config.get("TrackMaper", "TrackerInitIterNum", p.TrackerInitIterNum);
    InitOK &= config.get("TrackMaper", "TrackerInitType", p.TrackerInitType);
    InitOK &= config.get("TrackMaper", "TrackerMatchThresh", p.TrackerMatchThresh);
    InitOK &= config.get("TrackMaper", "MatchThreshModule", p.MatchThreshModule);
    InitOK &= config.get("TrackMaper", "MatchThreshAngle", p.MatchThreshAngle);
    InitOK &= config.get("TrackMaper", "MatchNumThresh", p.MatchNumThresh);
    InitOK &= config.get("TrackMaper", "RegularizeThresh", p.RegularizeThresh);
    InitOK &= config.get("TrackMaper", "ReweigthDistance", p.ReweigthDistance);
    InitOK &= config.get("TrackMaper", "ReshapeQAbsolute", p.ReshapeQAbsolute);
    InitOK &= config.get("TrackMaper", "ReshapeQRelative", p.ReshapeQRelative);
    InitOK &= config.get("TrackMaper", "LocationUncertainty", p.LocationUncertainty);
    InitOK &= config.get("TrackMaper", "LocationUncertaintyMatch", p.LocationUncertaintyMatch);
    InitOK &= config.get("TrackMaper", "DoReScaling", p.DoReScaling);
    InitOK &= config.get("TrackMaper", "GlobalMatchThreshold", p.MatchThreshold);
    InitOK &= config.get("REBVO", "SaveLog", p.SaveLog);
    InitOK &= config.get("REBVO", "LogFile", p.LogFile);
    InitOK &= config.get("REBVO", "TrayFile", p.TrayFile);
    InitOK &= config.get("IMU", "ImuMode", p.ImuMode);
    if (p.CameraType == 2) {   // DataSetCam keys (src/rebvo/rebvo.cpp:68-70)
        InitOK &= config.get("DataSetCamera", "DataSetDir", p.DataSetDir);
        InitOK &= config.get("DataSetCamera", "DataSetFile", p.DataSetFile);
        InitOK &= config.get("DataSetCamera", "TimeScale", p.CamTimeScale);
    }
    if (p.ImuMode > 0) {   // src/rebvo/rebvo.cpp:160-183
        InitOK &= config.get("IMU", "GiroMeasStdDev", p.GiroMeasStdDev);
        InitOK &= config.get("IMU", "GiroBiasStdDev", p.GiroBiasStdDev);
        InitOK &= config.get("IMU", "InitBias", p.InitBias);
        InitOK &= config.get("IMU", "InitBiasFrameNum", p.InitBiasFrameNum);
        InitOK &= config.get("IMU", "BiasHintX", p.BiasInitGuess[0]);
        InitOK &= config.get("IMU", "BiasHintY", p.BiasInitGuess[1]);
        InitOK &= config.get("IMU", "BiasHintZ", p.BiasInitGuess[2]);
        InitOK &= config.get("IMU", "g_module", p.g_module);
        InitOK &= config.get("IMU", "AcelMeasStdDev", p.AcelMeasStdDev);
        InitOK &= config.get("IMU", "g_module_uncer", p.g_module_uncer);
        InitOK &= config.get("IMU", "g_uncert", p.g_uncert);
        InitOK &= config.get("IMU", "VBiasStdDev", p.VBiasStdDev);
        InitOK &= config.get("IMU", "ScaleStdDevMult", p.ScaleStdDevMult);
        InitOK &= config.get("IMU", "ScaleStdDevMax", p.ScaleStdDevMax);
        InitOK &= config.get("IMU", "ScaleStdDevInit", p.ScaleStdDevInit);
        p.UseCamIMUSE3File = config.get("IMU", "CamImuSE3File", p.SE3File, false);
        InitOK &= config.get("IMU", "TimeDesinc", p.TimeDesinc);
    }
    if (p.ImuMode == 2) {
        InitOK &= config.get("IMU", "ImuFile", p.ImuFile);
        InitOK &= config.get("IMU", "TimeScale", p.ImuTimeScale);
    } else if (p.ImuMode == 1) {
        InitOK &= config.get("IMU", "SampleTime", p.SampleTime);
        InitOK &= config.get("IMU", "CircBufferSize", p.CircBufferSize);
        p.ImuTimeScale = 1;
    }
    // accepted and ignored (subsystems that do not exist on this path)
    config.get("Camera", "Rotate180", p.rotatedCam, false);
    config.get("REBVO", "VideoNetEnabled", p.VideoNetEnabled, false);
    config.get("REBVO", "TrackKeyFrames", p.TrackKeyFrames, false);
    if (config.get("REBVO", "StereoAvaiable", p.StereoAvaiable, false) && p.StereoAvaiable) {   // src/rebvo/rebvo.cpp:195-216
        InitOK &= config.get("DataSetCamera", "DataSetDirStereo", p.DataSetDirStereo);
        InitOK &= config.get("DataSetCamera", "DataSetFileStereo", p.DataSetFileStereo);
        InitOK &= config.get("Stereo", "ZfX", p.z_f_x_stereo);
        InitOK &= config.get("Stereo", "ZfY", p.z_f_y_stereo);
        InitOK &= config.get("Stereo", "PPx", p.pp_x_stereo);
        InitOK &= config.get("Stereo", "PPy", p.pp_y_stereo);
        InitOK &= config.get("Stereo", "KcR2", p.kc_stereo.Kc2);
        InitOK &= config.get("Stereo", "KcR4", p.kc_stereo.Kc4);
        InitOK &= config.get("Stereo", "KcR6", p.kc_stereo.Kc6);
        InitOK &= config.get("Stereo", "KcP1", p.kc_stereo.P1);
        InitOK &= config.get("Stereo", "KcP2", p.kc_stereo.P2);
    } else {
        p.StereoAvaiable = false;
    }
    // &ProcesorConfig (src/rebvo/rebvo.cpp:101-104): honoured when present.  CamaraT1 is the thread that takes frames and tracks them
    // (FirstThr and SecondThread of the reference are one thread here: the device does their work), CamaraT3 the output thread;
    // CamaraT2 is read and unused.
    config.get("ProcesorConfig", "SetAffinity", p.cpuSetAffinity, false);
    config.get("ProcesorConfig", "CamaraT1", p.cpu0, false);
    config.get("ProcesorConfig", "CamaraT2", p.cpu1, false);
    config.get("ProcesorConfig", "CamaraT3", p.cpu2, false);
    config.get("GPU", "Device", p.GpuDevice, false);
    config.get("GPU", "BatchGroup", p.GpuBatchGroup, false);
    config.get("GPU", "BatchSize", p.GpuBatchSize, false);
    {
        int mono = 1;
        if (config.get("GPU", "MonoUpload", mono, false)) p.GpuMonoUpload = mono != 0;
        int tprec = 64;
        if (config.get("GPU", "TrackerPrecision", tprec, false)) p.GpuTrackerPrecision = tprec;
    }
    construct();
}

REBVO::REBVO(const REBVOParameters &parameters)
    : params(parameters), quit(true), pipe(CBUFSIZE, 3), system_reset(false), cam_pipe(CCAMBUFSIZE, 2), cam_pipe_stereo(CCAMBUFSIZE, 2),
      outputFunc(nullptr) {
    construct();
}

void REBVO::construct() {
    if (!InitOK) return;
    cam = cam_model({params.pp_x, params.pp_y}, {params.z_f_x, params.z_f_y}, params.kc, params.ImageSize);
    if (params.ImageSize.w == 0 || params.ImageSize.h == 0) { InitOK = false; return; }
    switch (params.ImuMode) {   // IMU grabber (src/rebvo/rebvo.cpp:248-281)
    case 1:   // the application pushes samples (pushIMU)
        imu = new ImuGrabber(params.CircBufferSize, params.SampleTime);
        if (params.UseCamIMUSE3File && !imu->LoadCamImuSE3(params.SE3File.data())) {
            std::cout << "REBVO: Failed to load cam-imu transformation \n";
            InitOK = false;
            return;
        }
        break;
    case 2: {   // whole data set from a csv file
        bool error = false;
        imu = new ImuGrabber(ImuGrabber::LoadDataSet(params.ImuFile.data(), false, params.ImuTimeScale, error));
        if (error) {
            std::cout << "REBVO: Failed to initialize the imu grabber\n";
            InitOK = false;
            return;
        }
        if (params.UseCamIMUSE3File && !imu->LoadCamImuSE3(params.SE3File.data())) {
            std::cout << "Failed to load cam-imu transformation \n";
            InitOK = false;
            return;
        }
        break;
    }
    default: break;
    }
    // rebvo.cpp:284-285.  (An object that runs on the group engine gets page-locked views of its group's ring in their place when
    // it attaches: Init(), batch_group.cpp.)
    // (... so it gets no heap images here: 4 x 1 MB per object that would be thrown away at Init() — a thousand cameras, 4 GB of page faults)
    if (!useGroupEngine())
        for (unsigned i = 0; i < cam_pipe.Size(); i++) cam_pipe[i].img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);
    cam_stereo = cam;
    if (params.StereoAvaiable) {
        cam_stereo = cam_model({params.pp_x_stereo, params.pp_y_stereo}, {params.z_f_x_stereo, params.z_f_y_stereo}, params.kc_stereo,
                               params.ImageSize);
        if (!useGroupEngine())   // (a group member's pair ring: views of the group's second page-locked ring, like the main one)
            for (unsigned i = 0; i < cam_pipe_stereo.Size(); i++)
                cam_pipe_stereo[i].img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);
    }
    const bool lazy_views = useGroupEngine();        // (what only a callback or a snapshot looks at is allocated when one appears)
    for (PipeBuffer &pbuf : pipe) {                  // rebvo.cpp:297-312 (host views only; the rest lives in HBM)
        pbuf.ef = new edge_tracker(cam, params.MaxPoints > 0 ? params.MaxPoints : 1, lazy_views);
        pbuf.img = lazy_views ? nullptr : new Image<float>(params.ImageSize);
        pbuf.imgc = lazy_views ? nullptr : new Image<RGB24Pixel>(params.ImageSize);
        pbuf.imgc_pair = params.StereoAvaiable ? new Image<RGB24Pixel>(params.ImageSize) : nullptr;
        pbuf.ss = nullptr;
        pbuf.gt = nullptr;
    }
}

REBVO::~REBVO() {
    if (!quit) CleanUp();
    delete imu;
    if (InitOK)
        for (PipeBuffer &pbuf : pipe) {
            delete pbuf.ef;
            delete pbuf.img;
            delete pbuf.imgc;
            delete pbuf.imgc_pair;
        }
}

bool REBVO::Init() {
    if (!InitOK) return false;
    if (!quit) return true;
    if (params.CameraType != 3 && params.CameraType != 2) {
        last_error = "REBVO(hip): only CameraType=3 (custom camera) and CameraType=2 (dataset camera) are available";
        std::cout << last_error << "\n";
        return false;
    }
    if (params.CameraType == 2) {   // REBVO::initCamera, src/rebvo/rebvo.cpp:256-258
        dscam = new DataSetCam(params.DataSetDir.data(), params.DataSetFile.data(), params.ImageSize, params.CamTimeScale);
        if (dscam->Error()) {
            last_error = "REBVO: Failed to initialize the main camera (dataset list " + params.DataSetFile + ")";
            std::cout << last_error << "\n";
            delete dscam;
            dscam = nullptr;
            return false;
        }
    }
    if (params.ImuMode < 0 || params.ImuMode > 2) {
        last_error = "REBVO(hip): ImuMode must be 0, 1 (pushIMU) or 2 (IMU data set file)";
        std::cout << last_error << "\n";
        return false;
    }
    if (params.CameraType == 2 && params.StereoAvaiable) {   // REBVO::initPairCamera, src/rebvo/rebvo_first_t.cpp:64-76
        dscam_pair = new DataSetCam(params.DataSetDirStereo.data(), params.DataSetFileStereo.data(), params.ImageSize, params.CamTimeScale);
        if (dscam_pair->Error()) {
            last_error = "REBVO: Failed to initialize the stereo camera (dataset list " + params.DataSetFileStereo + ")";
            std::cout << last_error << "\n";
            delete dscam_pair;
            dscam_pair = nullptr;
            delete dscam;
            dscam = nullptr;
            return false;
        }
    }
    if (group) {   // Init() again without CleanUp() (e.g. after a device error closed the seat): let go of the old seat first
        quit = true;
        groupDetach();
    }
    if (useGroupEngine()) {   // batch_group.cpp: the (possibly shared) context and its tracker thread
        if (groupAttach()) return true;
        delete dscam;         // a refused member (parameter mismatch, full group, edgehip_create failure) keeps nothing
        dscam = nullptr;
        delete dscam_pair;
        dscam_pair = nullptr;
        return false;
    }
    if (!params.GpuBatchGroup.empty()) {
        last_error = "REBVO(hip): &GPU BatchGroup does not take a stereo pair together with ImuMode > 0 (the device-side IMU branch does not run the stereo rig)";
        std::cout << last_error << "\n";
        delete dscam;
        dscam = nullptr;
        delete dscam_pair;
        dscam_pair = nullptr;
        return false;
    }
    if (params.ImuMode > 0) imuTrackInit();
    edgehip_params hp;
    fill_hip_params(params, hp);
    hp.stereo_available = params.StereoAvaiable ? 1 : 0;
    // ring of 3 frame slots; with a stereo pair one more slot, behind the ring, holds the pair image's edge map
    int rc = edgehip_create(&hp, 1, params.StereoAvaiable ? 4 : 3, params.GpuDevice, &hip);
    if (rc == 0 && params.GpuTrackerPrecision != 64) rc = edgehip_set_tracker_precision(hip, params.GpuTrackerPrecision);   // (refused with ImuMode > 0 / a stereo rig)
    if (rc == 0 && params.StereoAvaiable) {
        // search radius 100 (rebvo_second_t.cpp:473); with the IMU branch the host drives the stereo stages itself
        rc = edgehip_set_slot_camera(hip, 3, params.pp_x_stereo, params.pp_y_stereo, params.z_f_x_stereo, params.z_f_y_stereo);
        if (rc == 0 && params.ImuMode == 0) rc = edgehip_set_stereo_rig(hip, 3, kTCam2Pair, kRCam2Pair, 100.0);
    }
    if (rc != 0) {   // no CPU fallback: fail loudly
        last_error = std::string("REBVO(hip): edgehip_create failed: ") + edgehip_last_error();
        std::cout << last_error << "\n";
        if (hip) edgehip_destroy(hip);
        hip = nullptr;
        return false;
    }
    quit = false;
    Thr0 = std::thread(TrackThread, this);
    return true;
}

void REBVO::ensureHostViews(PipeBuffer &pb, bool keylines) {
    if (!pb.img) pb.img = new Image<float>(params.ImageSize);
    if (!pb.imgc) pb.imgc = new Image<RGB24Pixel>(params.ImageSize);
    if (keylines) pb.ef->ensureKeyLines();
}

bool REBVO::CleanUp() {
    quit = true;
    if (group) groupDetach();
    if (Thr0.joinable()) Thr0.join();
    if (hip) { edgehip_destroy(hip); hip = nullptr; }
    imuTrackFree();
    if (dscam) { delete dscam; dscam = nullptr; }
    if (dscam_pair) { delete dscam_pair; dscam_pair = nullptr; }
    return true;
}

// ---- tracking thread ----------------------------------------------------------------------------------------
void REBVO::TrackThread(REBVO *cf) {
    std::thread Thr2(ThirdThread, cf);
    if (cf->params.cpuSetAffinity && !detail::set_affinity(cf->params.cpu0)) {   // rebvo_first_t.cpp:136-141
        std::cout << "REBVO: Cannot set cpu affinity on the first thread";
        cf->quit = true;
    }
    const size_t frame_bytes = (size_t)cf->params.ImageSize.w * cf->params.ImageSize.h * 3;
    static_assert(sizeof(KeyLine) == sizeof(edgehip_keyline), "KeyLine mirrors edgehip_keyline");
    double t0 = 0;
    const double min_frame_dt = 1.0 / cf->params.soft_fps - 0.5 / cf->params.config_fps;   // rebvo_first_t.cpp:146
    PipeBuffer *old_buf = nullptr;   // previous frame: held (player 1) until the next frame has been tracked
    int p_num = 0;
    int n_frames = 0;   // frames detected so far
    bool failed = false;
    while (!cf->quit && !failed) {
        PipeBuffer &new_buf = cf->pipe.RequestBuffer(0);
        if (cf->frame_by_frame) {   // rebvo_first_t.cpp:154-159
            while (!cf->frame_by_frame_advance && !cf->quit) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            cf->frame_by_frame_advance = false;
            std::cout << "Advancing frame...\n";
        }
        // ---- grab: custom camera ring (src/VideoLib/customcam.cpp:56-68: 1 ms time-out, retry) or dataset list ----
        customCam::CustomCamPipeBuffer *cbuf = nullptr;
        const RGB24Pixel *data = nullptr;
        double t = 0;
        while (true) {
            if (cf->dscam) {
                double ts = 0;
                data = cf->dscam->GrabBuffer(ts, false);
                if (!data) break;                  // end of the list / unreadable image: camera error -> quit
                t = ts;
                p_num = (int)cf->dscam->PakNum();
                if (t - t0 < min_frame_dt) { cf->dscam->ReleaseBuffer(); data = nullptr; if (cf->quit) break; continue; }
                break;
            }
            while ((cbuf = cf->cam_pipe.RequestBufferTimeoutable(1, 0.001)) == nullptr)
                if (cf->quit) break;
            if (!cbuf) break;
            t = cbuf->timestamp;
            p_num++;
            if (t - t0 < min_frame_dt) { cf->cam_pipe.ReleaseBuffer(1); cbuf = nullptr; continue; }   // soft-FPS drop (t0 starts at 0), :89, :172-177
            data = cbuf->img->Data();
            break;
        }
        // ---- the stereo pair: one frame per accepted main frame, no drop logic (rebvo_first_t.cpp:183-199) ----
        customCam::CustomCamPipeBuffer *cbuf_pair = nullptr;
        const RGB24Pixel *data_pair = nullptr;
        if (data && cf->params.StereoAvaiable) {
            double t_stereo = 0;
            if (cf->dscam_pair) {
                data_pair = cf->dscam_pair->GrabBuffer(t_stereo, false);
            } else {
                while ((cbuf_pair = cf->cam_pipe_stereo.RequestBufferTimeoutable(1, 0.001)) == nullptr)
                    if (cf->quit) break;
                if (cbuf_pair) { data_pair = cbuf_pair->img->Data(); t_stereo = cbuf_pair->timestamp; }
            }
            if (!data_pair) {
                std::cout << "bye bye cruel world on stereo\n";
                if (cbuf) cf->cam_pipe.ReleaseBuffer(1);
                else cf->dscam->ReleaseBuffer();
                data = nullptr;
            } else if (std::fabs(t - t_stereo) > 0.5 / cf->params.config_fps) {
                std::cout << "REBVO Warning: cameras are unsync: " << t - t_stereo << "\n";
            }
        }
        if (!data) {   // quitting or camera error: pass the flag down the ring (rebvo_first_t.cpp:165-170)
            new_buf.quit = true;
            cf->pipe.ReleaseBuffer(0);
            break;
        }
        const double tp0 = now_s();
        const bool imu_mode = cf->imu != nullptr;
        // ring of 3 device slots; without the IMU branch the library's own whole-frame driver advances it
        const int slot = imu_mode ? n_frames % 3 : edgehip_next_slot(cf->hip);
        // a mono data set (EuRoC) goes to the device as the 8-bit plane the file holds; the RGB24 expansion stays on the host for
        // the callback's PipeBuffer::imgc
        const uint8_t *grey = (!cbuf && cf->dscam) ? cf->dscam->GreyBuffer() : nullptr;
        int rc = grey ? edgehip_upload_grey8(cf->hip, slot, grey, 0, 1)
                      : edgehip_upload_rgb(cf->hip, slot, reinterpret_cast<const uint8_t *>(data), 0, 1);
        std::memcpy(new_buf.imgc->Data(), data, frame_bytes);
        new_buf.imgc_valid = true;
        if (cbuf) cf->cam_pipe.ReleaseBuffer(1);
        else cf->dscam->ReleaseBuffer();
        if (data_pair) {   // the pair image goes to the slot behind the ring
            const uint8_t *grey_pair = (!cbuf_pair && cf->dscam_pair) ? cf->dscam_pair->GreyBuffer() : nullptr;
            if (rc == 0) rc = grey_pair ? edgehip_upload_grey8(cf->hip, 3, grey_pair, 0, 1)
                                        : edgehip_upload_rgb(cf->hip, 3, reinterpret_cast<const uint8_t *>(data_pair), 0, 1);
            std::memcpy(new_buf.imgc_pair->Data(), data_pair, frame_bytes);
            if (cbuf_pair) cf->cam_pipe_stereo.ReleaseBuffer(1);
            else cf->dscam_pair->ReleaseBuffer();
        }
        if (imu_mode) {   // inter-frame IMU data, waiting for the samples to arrive (rebvo_first_t.cpp:294-304)
            while (!cf->quit) {
                new_buf.imu = cf->imu->GrabAndIntegrate(t0 + cf->params.TimeDesinc, t + cf->params.TimeDesinc);
                if (new_buf.imu.n > 0) break;
                std::this_thread::sleep_for(std::chrono::duration<double>(cf->params.SampleTime));
            }
        }
        t0 = t;
        new_buf.t = t;
        new_buf.p_id = p_num - 1;
        new_buf.quit = false;
        new_buf.dtp0 = 0;
        if (imu_mode) {
            // ---- IMU branch: GPU stages driven one by one, host filters in between ----
            if (rc == 0) rc = cf->trackFrameImu(slot, (slot + 2) % 3, old_buf != nullptr, t, new_buf);
        } else {
            // ---- the whole frame on the GPU: stage A + (from the second frame on) tracking and mapping ----
            if (rc == 0) rc = edgehip_process_frame(cf->hip, &t);
            edgehip_nav n;
            if (rc == 0) rc = edgehip_read_nav(cf->hip, &n);
            if (rc == 0) {
                new_buf.dt = n.dt;
                new_buf.K = 1; new_buf.Kp = n.Kp; new_buf.RKp = n.RKp;
                new_buf.s_rho_p = n.s_rho_q;
                new_buf.EstimationOK = n.estimation_ok != 0;
                new_buf.ef->nmatch = n.klm_num;
                new_buf.ef->reTunedThresh = n.retuned_thresh;
                new_buf.ef->kn = n.kn;   // edge_finder::KNum() of this frame (the .m log reports it even without a callback)
                if (old_buf) fill_nav(n, new_buf.nav);
                else new_buf.nav = NavData();   // first frame: "dummy processing", no estimate (rebvo_second_t.cpp:108-121)
                new_buf.stereo_match_num = 0;
                if (cf->params.StereoAvaiable && old_buf && n.estimation_ok) {
                    int32_t nm = 0;
                    if (edgehip_get_stereo_matches(cf->hip, &nm) == 0) new_buf.stereo_match_num = nm;
                }
            }
        }
        if (rc != 0) {
            std::cout << "REBVO(hip): " << edgehip_last_error() << "\n";
            failed = true;
            new_buf.quit = true;
            cf->pipe.ReleaseBuffer(0);
            break;
        }
        n_frames++;
        new_buf.dtp1 = now_s() - tp0;
        if (old_buf) cf->pushNav(new_buf.nav);
        // ---- hand the PREVIOUS frame to the output thread, with its edge map as the tracker left it ----
        if (old_buf) {
            const bool want = cf->haveCallBack();
            if (want) {
                const int so = (slot + 2) % 3;   // ring of 3: the slot before `slot`
                int32_t kn = 0;
                const int32_t seq0 = 0;
                edgehip_keyline *dst = reinterpret_cast<edgehip_keyline *>(old_buf->ef->kl.data());
                rc = edgehip_download_keylines_batch(cf->hip, so, 1, &seq0, &dst, &kn);   // (one packing kernel + one page-locked copy)
                if (rc != 0) {   // a failed download is a device error like any other: report it and stop (no silent empty map)
                    std::cout << "\nREBVO: edgehip_download_keylines failed: " << edgehip_last_error() << "\n";
                    failed = true;
                    old_buf->ef->kn = 0;
                } else {
                    old_buf->ef->kn = kn;
                }
                // (PipeBuffer::img for the callback: ConvertRGB2BW runs on the output thread, beside the next frame's tracking)
            } else {
                // nobody reads the KeyLine payload: skip the download, but keep edge_finder::KNum() as it was set when the frame was
                // detected — the .m log reports it (rebvo_third_t.cpp:280)
            }
            cf->pipe.ReleaseBuffer(1);
        }
        if (cf->system_reset) {   // rebvo_second_t.cpp:609-620
            if (imu_mode) cf->resetImuTrack(slot);
            else edgehip_depth_reset(cf->hip, -1);
            cf->system_reset = false;
        }
        // new_buf becomes old_buf: player 1 takes the slot player 0 releases now
        cf->pipe.ReleaseBuffer(0);
        old_buf = &cf->pipe.RequestBuffer(1);
    }
    // shutdown: the frame still held as old_buf is not delivered (as in the reference); pass the quit flag on
    if (old_buf) {
        old_buf->quit = true;
        cf->pipe.ReleaseBuffer(1);
    } else {
        PipeBuffer &b = cf->pipe.RequestBuffer(1);
        b.quit = true;
        cf->pipe.ReleaseBuffer(1);
    }
    cf->quit = true;
    Thr2.join();
}

// ---- output thread (src/rebvo/rebvo_third_t.cpp:172-347) -------------------------------------------------------
void REBVO::ThirdThread(REBVO *cf) {
    if (cf->params.cpuSetAffinity && !detail::set_affinity(cf->params.cpu2)) {   // rebvo_third_t.cpp:54-59
        std::cout << "REBVO: Cannot set cpu affinity on the third thread";
        cf->quit = true;
    }
    std::ofstream a_log, t_log;
    if (cf->params.SaveLog) {
        a_log.open(cf->params.LogFile.c_str());
        t_log.open(cf->params.TrayFile.c_str());
        if (!a_log.is_open() || !t_log.is_open()) {
            std::cout << "REBVO: Cannot open log files\n";
            cf->quit = true;
        }
    }
    int a_log_inx = 0;
    double t_proc_last = 0;
    while (true) {
        PipeBuffer pbuf = cf->pipe.RequestBuffer(2);   // struct copy; pointers alias the ring slot (:174)
        if (pbuf.quit) { cf->pipe.ReleaseBuffer(2); break; }
        const double ts = now_s();
        if (cf->params.SaveLog) {
            a_log_inx++;
            const NavData &nv = pbuf.nav;
            a_log << "Kp_cv(" << a_log_inx << ",:)=" << pbuf.Kp << ";\n";
            a_log << "RKp_cv(" << a_log_inx << ",:)=" << pbuf.RKp << ";\n";
            a_log << "Rot_cv(" << a_log_inx << ",:,:)=[" << nv.Rot(0, 0) << "," << nv.Rot(0, 1) << "," << nv.Rot(0, 2) << ";"
                  << nv.Rot(1, 0) << "," << nv.Rot(1, 1) << "," << nv.Rot(1, 2) << ";" << nv.Rot(2, 0) << "," << nv.Rot(2, 1) << ","
                  << nv.Rot(2, 2) << "];\n";
            a_log << "Vel_cv(" << a_log_inx << ",:)=[" << nv.Vel[0] << "," << nv.Vel[1] << "," << nv.Vel[2] << "];\n";
            a_log << "RotGiro_cv(" << a_log_inx << ",:)=[" << nv.RotGiro[0] << "," << nv.RotGiro[1] << "," << nv.RotGiro[2] << "];\n";
            a_log << "t_cv(" << a_log_inx << ",:)=" << pbuf.t << ";\n";
            a_log << "dt_cv(" << a_log_inx << ",:)=" << pbuf.dt << ";\n";
            a_log << "i_cv(" << a_log_inx << ",:)=" << pbuf.p_id << ";\n";
            a_log << "Pose_cv(" << a_log_inx << ",:,:)=[" << nv.Pose(0, 0) << "," << nv.Pose(0, 1) << "," << nv.Pose(0, 2) << ";"
                  << nv.Pose(1, 0) << "," << nv.Pose(1, 1) << "," << nv.Pose(1, 2) << ";" << nv.Pose(2, 0) << "," << nv.Pose(2, 1)
                  << "," << nv.Pose(2, 2) << "];\n";
            a_log << "Pos_cv(" << a_log_inx << ",:)=[" << nv.Pos[0] << "," << nv.Pos[1] << "," << nv.Pos[2] << "];\n";
            a_log << "K_cv(" << a_log_inx << ",:)=" << pbuf.K << ";\n";
            a_log << "KLN_cv(" << a_log_inx << ",:)=" << pbuf.ef->KNum() << ";\n";
            // the IMU / stereo block of the reference's log (rebvo_third_t.cpp:279-300), same keys in the same order
            auto v3log = [&](const char *key, const double *v) {
                a_log << key << "_cv(" << a_log_inx << ",:)=[" << v[0] << "," << v[1] << "," << v[2] << "];\n";
            };
            const IMUState &is = pbuf.imustate;
            v3log("Giro", &pbuf.imu.giro[0]);
            v3log("Acel", &pbuf.imu.acel[0]);
            v3log("CAcel", &pbuf.imu.cacel[0]);
            v3log("DGiro", &pbuf.imu.dgiro[0]);
            v3log("GBias", &is.Bg[0]);
            v3log("dWv", &is.dWv[0]);
            v3log("dWgv", &is.dWgv[0]);
            v3log("g", &is.g_est[0]);
            v3log("VBias", &is.b_est[0]);
            v3log("Av", &is.Av[0]);
            v3log("As", &is.As[0]);
            v3log("Posgv", &is.Posgv[0]);
            a_log << "SMM_cv(" << a_log_inx << ",:)=" << pbuf.stereo_match_num << ";\n";
            a_log << "TProc0_cv(" << a_log_inx << ",:)=" << pbuf.dtp0 << ";\n";
            a_log << "TProc1_cv(" << a_log_inx << ",:)=" << pbuf.dtp1 << ";\n";
            a_log << "TProc2_cv(" << a_log_inx << ",:)=" << t_proc_last << ";\n";
            // trajectory in TUM format: t pos quat (:311)
            double q[4];
            lie2quat(nv.PoseLie, q);
            t_log << std::scientific << std::setprecision(18) << pbuf.t / cf->params.ImuTimeScale << " " << nv.Pos[0] << " "
                  << nv.Pos[1] << " " << nv.Pos[2] << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
        }
        if (cf->haveCallBack() && pbuf.imgc_valid) {   // Image<float>::ConvertRGB2BW (include/VideoLib/image.h:197-203): the grey image a consumer may look at
            const RGB24Pixel *c = pbuf.imgc->Data();
            float *bw = pbuf.img->Data();
            const uint npx = pbuf.img->bSize();
            for (uint i = 0; i < npx; i++) bw[i] = (float)(c[i].pix.r + c[i].pix.g + c[i].pix.b);
        }
        cf->callCallBack(pbuf);   // :329, under call_mutex
        t_proc_last = now_s() - ts;
        if (cf->saveImg && pbuf.imgc_valid) {   // rebvo_third_t.cpp:335-343 (SavePPM, video_io.cpp:230-244); a frame launched before the request
                                                // carries no image (group engine): the next one that does is saved
            cf->saveImg = false;
            char name[128];
            snprintf(name, sizeof(name), "Snap%d.ppm", cf->snap_n);
            std::cout << "\nCamara Frontal: Tomando foto" << cf->snap_n++ << "\n";
            if (FILE *fout = fopen(name, "w")) {
                fprintf(fout, "P6\n%d %d 255\n", (int)cf->params.ImageSize.w, (int)cf->params.ImageSize.h);
                fwrite(pbuf.imgc->Data(), (size_t)cf->params.ImageSize.w * cf->params.ImageSize.h * sizeof(RGB24Pixel), 1, fout);
                fclose(fout);
            } else {
                perror("Video Out: Cannot open image!");
            }
        }
        cf->pipe.ReleaseBuffer(2);
    }
    if (a_log.is_open()) a_log.close();
    if (t_log.is_open()) t_log.close();
}

}  // namespace rebvo
