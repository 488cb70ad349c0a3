// rebvo.h — host-side mirror of the reference's library surface (include/rebvo/rebvo.h:64-640) on top of
// libedgehip.so.  An application written against the reference (app/rebvorun/main_custom_cam_example.cpp,
// ros/src/rebvo_ros/src/rebvo_nodelet.cpp) keeps its calls:
//
//     rebvo::REBVO cf("GlobalConfig");  cf.Init();
//     cf.setOutputCallback(&callback);
//     cf.requestCustomCamBuffer(ptr, tstamp);  (*ptr).copyFrom(data);  cf.releaseCustomCamBuffer();
//     NavData n = cf.getNav();  ...  cf.CleanUp();
//
// Same names, argument meaning, ownership and error behaviour (bool returns, InitOK on a bad config, false on
// a request time-out); the per-frame edge pipeline behind it runs on the GPU through include/edgehip.h.
//
// Differences, all forced by what is out of scope here:
//  * CameraType 3 (custom camera: the application feeds frames) and 2 (DataSetCam: EuRoC data.csv / TUM rgb.txt
//    image lists, PNG/PGM/PPM decoded without libgd, rebvo/datasetcam.h) are available; V4L and SimCam are
//    device I/O that this repository does not rebuild.
//  * ImuMode 1 (samples pushed with pushIMU) and 2 (IMU csv data set) run the IMU branch of SecondThread
//    (rebvo_second_t.cpp:182-336, 519-544): gyro pre-rotation, Minimizer_V and ExtRotVel on the GPU, BiasCorrect and
//    the ScaleEstimator filters on the host (rebvo/imu.h).  The pose-graph log (cf->poses) and key frames do not exist.
//  * StereoAvaiable runs the stereo depth steps of SecondThread (pair image through stage A with the &Stereo
//    intrinsics, stereo-mode directed matching, directed_matching_stereo with the rig hard-coded in
//    rebvo_second_t.cpp:466-470, fuseStereoDepth, Kp = 1) on the GPU; PipeBuffer::imgc_pair is filled, ::ef_pair /
//    ::ss_pair stay null, ::stereo_match_num is set.
//  * PipeBuffer::ss and ::gt are null (scale space and auxiliary field stay in HBM); PipeBuffer::ef is a
//    host view with the edge_finder members consumers use: KNum(), operator[], begin()/end(), GetCam(),
//    getThresh(), NumMatches().
//  * Vectors/matrices are minimal POD types (Vector3, Matrix3x3: operator[] / operator()(i,j)) with the
//    layout of TooN::Vector<3> / TooN::Matrix<3,3>; compile with -DREBVO_HAVE_TOON to get the TooN types.
//  * Only the config keys that reach this path are mandatory; keys of subsystems that do not exist here
//    (UDP, encoders, SimuCamera, ProcesorConfig ...) are accepted and ignored.  Optional section:
//        &GPU  Device=0  BatchGroup=<name>  BatchSize=<N>
//  * Batch groups (CameraType 3 or 2; all members alike: ImuMode 0, or ImuMode 1 / 2 with the IMU branch batched on the device, or — ImuMode 0 —
//    StereoAvaiable with a pair frame per main frame through requestStereoCustomCamBuffer).  N rebvo::REBVO objects whose configs name the same &GPU BatchGroup share ONE
//    edgehip context of N sequences: the group's tracker thread takes the newest frame of every member's camera ring and runs them
//    through one edgehip_process_frame (every kernel launch carries the N cameras), each object keeps its own ring, callback, log and
//    getNav().  The members advance in lock-step — one frame of every running member per step, like a synchronised camera rig or a
//    multi-sequence replay (BASELINE configs[4]); the group starts once BatchSize members have called Init() and a member that calls
//    CleanUp() leaves it; a DataSetCam (CameraType 2) may be a member too — N data sets replayed as one batch on one device, each
//    through its own object — and leaves when its list ends.  Camera buffers are page-locked and go to the device by asynchronous copies that run under the frames
//    before (the reference's T0 || T1, rebvo_first_t.cpp:134 / rebvo_second_t.cpp:102); the host waits for a frame's record only
//    when the next one is already enqueued (or no further frame is waiting), and KeyLines come back as AoS only for a member with a
//    callback.  An object without a BatchGroup is a group of one: the same engine.
#ifndef REBVO_AMD_HOST_REBVO_H
#define REBVO_AMD_HOST_REBVO_H

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rebvo/imu.h"
#include "rebvo/pipeline.h"

#ifdef REBVO_HAVE_TOON
#include <TooN/TooN.h>
#endif

struct edgehip_ctx;

namespace rebvo {

class DataSetCam;

typedef unsigned int uint;

// ---- small value types (reference: include/VideoLib/video_io.h:45-68) ------------------------------------
union RGB24Pixel {
    struct { uint8_t r, g, b; } pix;
    uint8_t dat[3];
};
struct Size2D { uint w, h; };
struct Point2DF { float x, y; };
struct Point2DI { int x, y; };

#ifdef REBVO_HAVE_TOON
typedef TooN::Vector<3> Vector3;
typedef TooN::Matrix<3, 3> Matrix3x3;
inline Vector3 Zeros3() { return TooN::Zeros; }
inline Matrix3x3 Identity3() { return TooN::Identity; }
#else
struct Vector3 {
    double v[3] = {0, 0, 0};
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
};
struct Matrix3x3 {  // row-major, like TooN::Matrix<3,3>
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double &operator()(int r, int c) { return m[r * 3 + c]; }
    const double &operator()(int r, int c) const { return m[r * 3 + c]; }
    double *operator[](int r) { return m + r * 3; }
    const double *operator[](int r) const { return m + r * 3; }
};
inline Vector3 Zeros3() { return Vector3(); }
inline Matrix3x3 Identity3() { return Matrix3x3(); }
#endif
inline Matrix3x3 scaled_identity3(double s) {
    Matrix3x3 m = Identity3();
    for (int i = 0; i < 3; i++) m(i, i) = s;
    return m;
}

// ---- Image<T> (reference: include/VideoLib/image.h:42-217; the members applications use) ---------------
template <typename DataType>
class Image {
    DataType *data = nullptr;
    Size2D size = {0, 0};
    uint bsize = 0;
    bool data_owned = false;

public:
    Image() {}
    explicit Image(const Size2D &i_size) : data(new DataType[(size_t)i_size.w * i_size.h]), size(i_size), bsize(i_size.w * i_size.h), data_owned(true) {}
    Image(DataType *i_data, const Size2D &i_size) : data(i_data), size(i_size), bsize(i_size.w * i_size.h), data_owned(false) {}
    Image(const Image &img) : data(new DataType[img.bsize]), size(img.size), bsize(img.bsize), data_owned(true) {
        std::memcpy(data, img.data, sizeof(DataType) * bsize);
    }
    Image &operator=(const Image &img) {
        if (img.size.w != size.w || img.size.h != size.h) throw std::length_error("Image: size mismatch");
        std::memcpy(data, img.data, sizeof(DataType) * bsize);
        return *this;
    }
    ~Image() { if (data_owned) delete[] data; }
    DataType *Data() { return data; }
    const DataType *Data() const { return data; }
    DataType &operator[](const uint inx) { return data[inx]; }
    const DataType &operator[](const uint inx) const { return data[inx]; }
    DataType &operator()(const uint x, const uint y) { return data[(size_t)y * size.w + x]; }
    uint GetIndex(const uint x, const uint y) const { return y * size.w + x; }
    bool isInxValid(const uint &x, const uint &y) const { return x < size.w && y < size.h; }
    const Size2D &Size() const { return size; }
    const uint &bSize() const { return bsize; }
    Image &operator=(DataType *img) { std::memcpy(data, img, sizeof(DataType) * bsize); return *this; }
    void copyTo(DataType *img) const { std::memcpy(img, data, sizeof(DataType) * bsize); }
    void copyFrom(const DataType *img) { std::memcpy(data, img, sizeof(DataType) * bsize); }
    void Reset(DataType d) { for (uint i = 0; i < bsize; i++) data[i] = d; }
};

// ---- camera model (reference: include/UtilLib/cam_model.h:33-179; projection helpers consumers call) -----
class cam_model {
public:
    struct rad_tan_distortion { double Kc2 = 0, Kc4 = 0, Kc6 = 0, P1 = 0, P2 = 0; };
    Point2DF pp = {0, 0};
    Point2DF zf = {1, 1};
    double zfm = 1;
    rad_tan_distortion Kc;
    Size2D sz = {0, 0};
    cam_model() {}
    cam_model(Point2DF prin_point, Point2DF focal_dist, rad_tan_distortion DistKc, Size2D ImageSize)
        : pp(prin_point), zf(focal_dist), zfm((focal_dist.x + focal_dist.y) / 2), Kc(DistKc), sz(ImageSize) {}
    template <typename P> P Hom2Img(const P &ph) const { return {ph.x + pp.x, ph.y + pp.y}; }
    template <typename P> P Img2Hom(const P &pi) const { return {pi.x - pp.x, pi.y - pp.y}; }
    // p = (x_hom, y_hom, rho) -> 3D point, as cam_model::unprojectHomCordVec
    Vector3 unprojectHomCordVec(double x, double y, double rho) const {
        Vector3 r;
        r[0] = x / rho / zfm; r[1] = y / rho / zfm; r[2] = 1 / rho;
        return r;
    }
};

// ---- KeyLine: byte-for-byte the reference's struct (include/mtracklib/edge_finder.h:45-91), 168 bytes -----
struct KeyLine {
    int p_inx;
    Point2DF m_m;
    Point2DF u_m;
    float n_m;
    float score;
    Point2DF c_p;
    double rho, s_rho, rho_nr, s_rho_nr, rho0, s_rho0;
    Point2DF p_m;
    Point2DF p_m_0;
    int m_id, m_id_f, m_id_kf;
    uint m_num;
    Point2DF m_m0;
    double n_m0;
    int p_id, n_id, net_id;
    int stereo_m_id;
    double stereo_rho, stereo_s_rho;
};
static_assert(sizeof(KeyLine) == 168, "KeyLine must keep the reference layout");

class sspace;          // stay on the device: PipeBuffer::ss / ::gt are null
class global_tracker;

// Host view of one frame's edge map: what the output callback iterates.
class edge_tracker {
    cam_model cam_mod;
    std::vector<KeyLine> kl;
    int kn = 0;
    int nmatch = 0;
    float reTunedThresh = 0;
    int kl_cap = 0;
    friend class REBVO;

public:
    edge_tracker(const cam_model &cam, int kl_num_max) : cam_mod(cam), kl(kl_num_max) {}
    // (mirror only) an object on the group engine gets its 2.7 MB KeyLine array when somebody wants KeyLines — the first output
    // callback — not at construction: a thousand cameras without callbacks are a thousand objects without 8 x 5 MB of host views each
    edge_tracker(const cam_model &cam, int kl_num_max, bool lazy) : cam_mod(cam), kl(lazy ? 0 : kl_num_max), kl_cap(kl_num_max) {}
    void ensureKeyLines() { if (kl.size() < (size_t)kl_cap) kl.resize((size_t)kl_cap); }
    cam_model &GetCam() { return cam_mod; }
    int KNum() const { return kn; }
    int NumMatches() const { return nmatch; }
    KeyLine &operator[](uint inx) { return kl[inx]; }
    float getThresh() const { return reTunedThresh; }
    typedef KeyLine *iterator;
    typedef const KeyLine *const_iterator;
    iterator begin() { return kl.data(); }
    iterator end() { return kl.data() + kn; }
};

// ---- REBVOParameters (reference: include/rebvo/rebvo.h:64-235, same member names) ----------------------
struct REBVOParameters {
    int CameraType = 3;
    std::string VideoNetHost; int VideoNetPort = 0; bool VideoNetEnabled = false; bool BlockingUDP = false;
    int VideoSave = 0; std::string VideoSaveFile; int VideoSaveBuffersize = 0;
    int encoder_type = 0; std::string encoder_dev; uint EdgeMapDelay = 0;
    bool SaveLog = false; std::string LogFile; std::string TrayFile;
    bool TrackKeyFrames = false; double KFSavePercent = 0; bool StereoAvaiable = false;
    std::string DataSetFile, DataSetDir, DataSetFileStereo, DataSetDirStereo; double CamTimeScale = 1;
    Size2D ImageSize = {0, 0};
    float z_f_x = 0, z_f_y = 0, pp_y = 0, pp_x = 0;
    cam_model::rad_tan_distortion kc;
    float z_f_x_stereo = 0, z_f_y_stereo = 0, pp_y_stereo = 0, pp_x_stereo = 0;   // &Stereo section (StereoAvaiable)
    cam_model::rad_tan_distortion kc_stereo;
    double config_fps = 30, soft_fps = 30;
    bool useUndistort = false, rotatedCam = false;
    std::string CameraDevice;
    std::string SimFile; double sim_save_nframes = 0; int simu_time_on = 0, simu_time_step = 0;
    double simu_time_sweep = 0, simu_time_start = 0;
    int ImuMode = 0; std::string ImuFile; bool UseCamIMUSE3File = false; std::string SE3File; double ImuTimeScale = 1;
    // IMU branch (reference include/rebvo/rebvo.h:150-173)
    double GiroMeasStdDev = 1.6968e-04, GiroBiasStdDev = 1.9393e-05;
    bool InitBias = false; int InitBiasFrameNum = 10;
    Vector3 BiasInitGuess = Zeros3();
    double AcelMeasStdDev = 2e-3, g_module = 9.8, g_module_uncer = 0.2e3, g_uncert = 2e-3, VBiasStdDev = 1e-7;
    double ScaleStdDevMult = 1e-2, ScaleStdDevMax = 1e-4, ScaleStdDevInit = 1.2e-3;
    double SampleTime = 0.00125; int CircBufferSize = 1000; double TimeDesinc = 0;
    int cpuSetAffinity = 0, cpu0 = 0, cpu1 = 0, cpu2 = 0;
    // Detector
    double Sigma0 = 1.7818, KSigma = 1.2599;
    int DetectorPlaneFitSize = 2; double DetectorPosNegThresh = 0.4, DetectorDoGThresh = 0.095259868922420;
    int ReferencePoints = 12000, TrackPoints = 12000, MaxPoints = 16000;
    double DetectorThresh = 0.01, DetectorAutoGain = 5e-7, DetectorMaxThresh = 0.5, DetectorMinThresh = 0.005;
    // Tracker-Mapper
    int MatchThreshold = 500;
    double SearchRange = 40, QCutOffNumBins = 100, QCutOffQuantile = 0.9;
    int TrackerIterNum = 5, TrackerInitIterNum = 2, TrackerInitType = 2;
    double TrackerMatchThresh = 0.5, LocationUncertaintyMatch = 2, MatchThreshModule = 1, MatchThreshAngle = 45;
    double ReweigthDistance = 2;
    uint MatchNumThresh = 0;
    double RegularizeThresh = 0.5, ReshapeQAbsolute = 1e-4, ReshapeQRelative = 1.6968e-04, LocationUncertainty = 1;
    double DoReScaling = 0;
    // extension: HIP device ordinal (optional config section &GPU, key Device)
    int GpuDevice = 0;
    // extension: objects with the same non-empty GpuBatchGroup share one device context of GpuBatchSize sequences (&GPU BatchGroup / BatchSize)
    std::string GpuBatchGroup;
    int GpuBatchSize = 0;
    // extension: frames whose pixels all have R = G = B (a mono camera's, tripled to fit the RGB24 surface) cross PCIe as their 8-bit
    // plane when every frame of a step is such a frame (&GPU MonoUpload, default 1; src/mono_pack.cpp, batch_group.cpp)
    bool GpuMonoUpload = true;
    // &GPU TrackerPrecision (64, or 32): which instantiation of the tracker the device runs — global_tracker::Minimizer_RV<double> (the x86
    // reference, rebvo_second_t.cpp:346) or Minimizer_RV<float> (what a USE_NE10 build of the reference runs, :339-343) — the run-time form of
    // the reference's compile-time switch (edgehip_set_tracker_precision).  ImuMode 0 only; members of a batch group must agree.
    int GpuTrackerPrecision = 64;
};

// Filter state SecondThread keeps in the IMU branch (reference include/rebvo/rebvo.h:239-290, same member names).
// Unlike the reference every member starts initialised (the reference leaves g_est, b_est, X, P ... to the heap).
struct IMUState {
    Vector3 Vg = Zeros3();                              // translation from Minimizer_V (gyro-rotated frame)
    Vector3 dVv = Zeros3(), dWv = Zeros3();             // visual increment (ExtRotVel)
    Vector3 dVgv = Zeros3(), dWgv = Zeros3();           // after the gyro prior (BiasCorrect)
    Vector3 Vgv = Zeros3(), Wgv = Zeros3();
    Vector3 dVgva = Zeros3(), dWgva = Zeros3(), Vgva = Zeros3();   // after the accelerometer filter
    Matrix3x3 P_Vg = scaled_identity3(1e50);
    Matrix3x3 RGiro = Identity3(), RGBias = Identity3();
    Vector3 Bg = Zeros3();                              // gyro bias
    Matrix3x3 W_Bg = Identity3();                       // its information
    Vector3 Av = Zeros3(), As = Zeros3();               // visual / accelerometer acceleration
    la::Vec<7> X = la::Vec<7>::zeros();                 // (scale angle, g, visual bias)
    la::Mat<7, 7> P = la::Mat<7, 7>::zeros();
    Matrix3x3 Qrot = Identity3(), Qg = Identity3(), Qbias = Identity3();
    double QKp = 0, Rg = 0;
    Matrix3x3 Rs = Identity3(), Rv = Identity3();
    Vector3 g_est = Zeros3(), u_est = Zeros3(), b_est = Zeros3();
    la::Mat<6, 6> Wvw = la::Mat<6, 6>::zeros();
    la::Vec<6> Xvw = la::Vec<6>::zeros();
    Vector3 Posgv = Zeros3(), Posgva = Zeros3();
    bool init = false;
};

struct NavData {
    double t = 0, dt = 0, scale = 1;
    Matrix3x3 Rot = Identity3();
    Vector3 RotLie = Zeros3(), RotGiro = Zeros3(), Vel = Zeros3(), g = Zeros3();
    Matrix3x3 Pose = Identity3();
    Vector3 PoseLie = Zeros3(), Pos = Zeros3();
};

struct PipeBuffer {
    sspace *ss = nullptr;
    global_tracker *gt = nullptr;
    edge_tracker *ef = nullptr;
    Image<RGB24Pixel> *imgc = nullptr;
    Image<float> *img = nullptr;
    sspace *ss_pair = nullptr;
    edge_tracker *ef_pair = nullptr;
    Image<RGB24Pixel> *imgc_pair = nullptr;
    Image<float> *img_pair = nullptr;
    double t = 0, dt = 0, s_rho_p = 0;
    NavData nav;
    IMUState imustate;
    double dtp0 = 0, dtp1 = 0;
    double K = 1, Kp = 1, RKp = 0;
    int p_id = 0;
    bool EstimationOK = false;
    bool quit = false;
    int stereo_match_num = 0;
    IntegratedImuData imu;
    bool imgc_valid = true;   // (mirror only) imgc holds THIS frame: the group engine copies the frame for the output thread only when a
                              // callback or a snapshot request is pending at launch time — per frame, not per object
};

namespace customCam {
struct CustomCamPipeBuffer {
    std::shared_ptr<Image<RGB24Pixel>> img;
    double timestamp = 0;
};
}  // namespace customCam

constexpr int CBUFSIZE = 0x08;
constexpr int CCAMBUFSIZE = 0x04;

class REBVO {
    REBVOParameters params;
    std::thread Thr0;
    bool InitOK = true;
    std::mutex nav_mutex;
    NavData nav;
    std::atomic_bool quit;
    Pipeline<PipeBuffer> pipe;
    std::atomic_bool system_reset;
    std::atomic_bool saveImg{false};
    int snap_n = 0;
    std::atomic_bool frame_by_frame{false}, frame_by_frame_advance{false};
    Pipeline<customCam::CustomCamPipeBuffer> cam_pipe;
    Pipeline<customCam::CustomCamPipeBuffer> cam_pipe_stereo;   // pair camera (StereoAvaiable, CameraType 3)
    cam_model cam;
    cam_model cam_stereo;
    std::mutex call_mutex;
    std::function<bool(PipeBuffer &)> outputFunc;
    edgehip_ctx *hip = nullptr;
    DataSetCam *dscam = nullptr;   // CameraType == 2
    DataSetCam *dscam_pair = nullptr;   // its stereo pair (DataSetDirStereo / DataSetFileStereo)
    ImuGrabber *imu = nullptr;     // ImuMode > 0
    struct ImuTrack;               // SecondThread's IMU-branch locals (rebvo_imu.cpp)
    ImuTrack *imutrack = nullptr;
    std::string last_error;
    class BatchGroup;              // batch_group.cpp: the shared-context engine behind CameraType 3 / ImuMode 0 / mono objects
    friend class BatchGroup;
    BatchGroup *group = nullptr;
    int group_seat = -1;
    customCam::CustomCamPipeBuffer *cam_cur = nullptr;   // the buffer the application holds between request and releaseCustomCamBuffer
    customCam::CustomCamPipeBuffer *cam_cur_pair = nullptr;   // ... and between requestStereoCustomCamBuffer and releaseStereoCustomCamBuffer
    void groupFrameWritten(customCam::CustomCamPipeBuffer *b, bool pair = false);
    bool cam_pinned = false;       // the camera ring's images are page-locked views of the group's ring (batch_group.cpp), not heap images
    bool groupAttach();            // Init() of such an object
    void groupDetach();            // CleanUp()
    // the custom camera always; a DataSetCam when its config names a BatchGroup (a multi-sequence replay on one device): its images
    // then reach the group through the object's own camera ring, put there by a feeder thread
    bool useGroupEngine() const {
        // ImuMode 1 / 2 (round 6): members of a NAMED group run the device-side IMU branch for the whole batch (edgehip_imu_enable /
        // edgehip_set_imu; the group thread grabs every member's inter-frame IMU data); an object alone keeps the host-side filters.
        // StereoAvaiable: members of a NAMED group (ImuMode 0) share a context with a pair slot and the rig inside
        // edgehip_process_frame, their pair frames cross in the group's second page-locked ring; an object alone keeps its own thread.
        const bool named = !params.GpuBatchGroup.empty();
        if (!(params.CameraType == 3 || (params.CameraType == 2 && named))) return false;
        if (params.StereoAvaiable) return named && params.ImuMode == 0;
        return params.ImuMode == 0 || named;
    }
    std::thread feeder;            // CameraType 2 in a batch group: DataSetCam -> camera ring
    static void FeedThread(REBVO *cf);

    // the stereo rig SecondThread hard-codes (rebvo_second_t.cpp:466-470; EuRoC cam0 -> cam1)
    static const double kRCam2Pair[9], kTCam2Pair[3];

    bool callCallBack(PipeBuffer &pbuf) {
        std::lock_guard<std::mutex> locker(call_mutex);
        if (outputFunc) return outputFunc(pbuf);
        return true;
    }
    bool haveCallBack() {
        std::lock_guard<std::mutex> locker(call_mutex);
        return (bool)outputFunc;
    }
    void pushNav(const NavData &navdat) {
        std::lock_guard<std::mutex> locker(nav_mutex);
        nav = navdat;
    }
    void construct();
    static void TrackThread(REBVO *cf);   // FirstThr + SecondThread of the reference: one GPU frame per loop
    // IMU branch of one frame: stage A, then (from the second frame on) the tracker / filters / mapper sequence of
    // rebvo_second_t.cpp:128-606 with ImuMode > 0.  Returns 0 or an edgehip error code.
    int trackFrameImu(int slot_new, int slot_old, bool have_pair, double t, PipeBuffer &new_buf);
    void imuTrackInit();
    void imuTrackFree();
    void resetImuTrack(int slot_new);   // REBVO::Reset() in the IMU branch: depth reset of the newest map, pose to identity
    static void ThirdThread(REBVO *cf);   // output: log, trajectory, callback

public:
    REBVO(const char *configFile);
    REBVO(const REBVOParameters &parameters);
    ~REBVO();
    bool Init();
    bool CleanUp();

    void StartSimSave() {}
    // the next delivered frame's image is written as Snap<n>.ppm into the working directory (rebvo.h:459, rebvo_third_t.cpp:335-343);
    // with the group engine the image is kept for the output thread only while a callback is registered or a snapshot is pending
    void TakeSnapshot() { saveImg = true; }
    // (mirror only) the host views of a PipeBuffer that exist only for an output callback / a snapshot — the KeyLine array, the grey
    // image, the colour image — allocated on first use by an object on the group engine (batch_group.cpp)
    void ensureHostViews(PipeBuffer &pb, bool keylines);
    void Reset() { system_reset = true; }
    bool Running() { return !quit; }
    void startKeyFrames() {}
    void endKeyFrames() {}
    bool toggleKeyFrames() { return false; }
    // frame-by-frame mode: no new frame is taken from the camera until advanceFrameByFrame() (rebvo.h:481-488, rebvo_first_t.cpp:154-159)
    bool toggleFrameByFrame() { return frame_by_frame = !frame_by_frame; }
    bool advanceFrameByFrame() { return frame_by_frame_advance = true; }

    NavData getNav() {
        std::lock_guard<std::mutex> locker(nav_mutex);
        NavData navdat = nav;
        return navdat;
    }
    const REBVOParameters &getParams() { return params; }
    // Cam-IMU transformation, Pimu = RCam2IMU * Pcam + TCam2IMU (rebvo.h:519-526); false without an ImuGrabber
    bool setCamImuSE3(const Matrix3x3 &RCam2IMU, const Vector3 &TCam2IMU);
    // Push one IMU sample (ImuMode 1; rebvo.h:534-539).  ImuGrabber::PushData throws std::overflow_error when the
    // circular buffer is full, as in the reference.
    bool pushIMU(const ImuData &data) {
        if (imu) return imu->PushData(data);
        return false;
    }

    bool requestCustomCamBuffer(std::shared_ptr<Image<RGB24Pixel>> &ptr, double time_stamp, double timeout_secs = 0) {
        customCam::CustomCamPipeBuffer *ccpb = cam_pipe.RequestBufferTimeoutable(0, timeout_secs);
        if (ccpb == nullptr) return false;
        if (!(*ccpb).img) (*ccpb).img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);   // (an object on the group engine outside Init()..CleanUp(): no ring yet)
        ptr = (*ccpb).img;
        (*ccpb).timestamp = time_stamp;
        cam_cur = ccpb;
        return true;
    }
    void releaseCustomCamBuffer() {
        if (group && cam_cur) groupFrameWritten(cam_cur);   // (a mono frame's 8-bit plane, on this thread: src/mono_pack.cpp)
        cam_cur = nullptr;
        cam_pipe.ReleaseBuffer(0);
    }
    // the pair camera's ring (reference rebvo.h:570-586); one pair frame is consumed per accepted main frame
    bool requestStereoCustomCamBuffer(std::shared_ptr<Image<RGB24Pixel>> &ptr, double time_stamp, double timeout_secs = 0) {
        customCam::CustomCamPipeBuffer *ccpb = cam_pipe_stereo.RequestBufferTimeoutable(0, timeout_secs);
        if (ccpb == nullptr) return false;
        if (!(*ccpb).img) (*ccpb).img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);   // (group engine outside Init()..CleanUp())
        ptr = (*ccpb).img;
        (*ccpb).timestamp = time_stamp;
        cam_cur_pair = ccpb;
        return true;
    }
    void releaseStereoCustomCamBuffer() {
        if (group && cam_cur_pair) groupFrameWritten(cam_cur_pair, true);   // (a mono pair frame's 8-bit plane, like releaseCustomCamBuffer)
        cam_cur_pair = nullptr;
        cam_pipe_stereo.ReleaseBuffer(0);
    }

    template <typename T>
    void setOutputCallback(bool (T::*method)(PipeBuffer &), T *obj) {
        std::lock_guard<std::mutex> locker(call_mutex);
        outputFunc = std::bind(method, obj, std::placeholders::_1);
    }
    void setOutputCallback(bool (*func)(PipeBuffer &)) {
        std::lock_guard<std::mutex> locker(call_mutex);
        if (func) outputFunc = std::bind(func, std::placeholders::_1);
        else outputFunc = nullptr;
    }
    bool isInitOk() const { return InitOK; }
    const std::string &lastError() const { return last_error; }
    Matrix3x3 getCam2ImuRot();
    Vector3 getCam2ImuPos();
};

}  // namespace rebvo
#endif
