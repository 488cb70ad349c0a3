// custom_cam_replay — the reference's custom-camera example (app/rebvorun/main_custom_cam_example.cpp:46-118)
// against the HIP-backed library: frames come from a raw file instead of DataSetCam.
//
//   custom_cam_replay <GlobalConfig> <frames.rgb24> <n_frames> <t0> <dt> [dump.txt [imu.csv]]
//
// frames.rgb24 = n_frames x ImageHeight x ImageWidth x 3 bytes.  Every frame goes through
// requestCustomCamBuffer / releaseCustomCamBuffer; the output callback (third thread) appends one line per
// delivered frame to dump.txt:  p_id t kn nmatch EstimationOK Pos[3] PoseLie[3] Vel[3] sum(rho) sum(s_rho) and the
// columns dataset_replay adds (RotLie, RotGiro, g, scale, K, Kp, RKp, IMU state, dt).
// With ImuMode=1 in the config, imu.csv ("t,gx,gy,gz,ax,ay,az", seconds) is fed through pushIMU the way an
// application would: every sample up to a little past a frame's time stamp is pushed before that frame is submitted.
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <vector>

#include "rebvo/rebvo.h"

using namespace rebvo;

static std::ofstream g_dump;
static int g_calls = 0;

static bool callback(PipeBuffer &p) {
    g_calls++;
    if (!g_dump.is_open()) return true;
    double sr = 0, ss = 0;
    for (KeyLine &kl : *p.ef) { sr += kl.rho; ss += kl.s_rho; }
    g_dump << std::setprecision(17) << p.p_id << " " << p.t << " " << p.ef->KNum() << " " << p.ef->NumMatches() << " "
           << (int)p.EstimationOK;
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.Pos[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.PoseLie[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.Vel[i];
    g_dump << " " << sr << " " << ss;
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.RotLie[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.RotGiro[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.nav.g[i];
    g_dump << " " << p.nav.scale << " " << p.K << " " << p.Kp << " " << p.RKp;
    for (int i = 0; i < 3; i++) g_dump << " " << p.imustate.Vg[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.imustate.Bg[i];
    for (int i = 0; i < 7; i++) g_dump << " " << p.imustate.X[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.imustate.b_est[i];
    for (int i = 0; i < 3; i++) g_dump << " " << p.imustate.u_est[i];
    g_dump << " " << p.dt << "\n";
    return true;
}

int main(int argn, char **argv) {
    if (argn < 6) {
        std::cout << "usage: custom_cam_replay <GlobalConfig> <frames.rgb24> <n_frames> <t0> <dt> [dump.txt [imu.csv]]\n";
        return 2;
    }
    REBVO cf(argv[1]);
    if (!cf.isInitOk()) { std::cout << "config error\n"; return 3; }
    if (argn > 6) g_dump.open(argv[6]);
    cf.setOutputCallback(&callback);
    if (!cf.Init()) return 4;

    const Size2D sz = cf.getParams().ImageSize;
    const size_t fb = (size_t)sz.w * sz.h * 3;
    const int n = atoi(argv[3]);
    const double t0 = atof(argv[4]), dt = atof(argv[5]);
    std::ifstream in(argv[2], std::ios::binary);
    if (!in.is_open()) { std::cout << "cannot open " << argv[2] << "\n"; cf.CleanUp(); return 5; }
    std::vector<RGB24Pixel> frame((size_t)sz.w * sz.h);
    std::vector<ImuData> imu_samples;
    if (argn > 7) {
        bool error = false;
        imu_samples = ImuGrabber::LoadDataSet(argv[7], false, 1.0, error);
        if (error) { cf.CleanUp(); return 5; }
    }
    size_t imu_next = 0;
    const double imu_ahead = 2.5 * cf.getParams().SampleTime;   // the grabber needs a sample at or past the frame time
    for (int k = 0; k < n && cf.Running(); k++) {
        in.read(reinterpret_cast<char *>(frame.data()), fb);
        if ((size_t)in.gcount() != fb) break;
        while (imu_next < imu_samples.size() && imu_samples[imu_next].tstamp <= t0 + dt * k + imu_ahead) {
            try {
                cf.pushIMU(imu_samples[imu_next]);
            } catch (const std::overflow_error &e) {
                std::cout << "pushIMU: " << e.what() << "\n";
                cf.CleanUp();
                return 7;
            }
            imu_next++;
        }
        std::shared_ptr<Image<RGB24Pixel>> ptr;
        while (!cf.requestCustomCamBuffer(ptr, t0 + dt * k, 0.1))
            if (!cf.Running()) break;
        if (!ptr) break;
        (*ptr).copyFrom(frame.data());
        cf.releaseCustomCamBuffer();
    }
    // let the pipeline drain: every submitted frame has been consumed once a full ring of requests succeeds
    for (int i = 0; i < CCAMBUFSIZE && cf.Running(); i++) {
        std::shared_ptr<Image<RGB24Pixel>> ptr;
        while (!cf.requestCustomCamBuffer(ptr, -1e9, 0.1))   // stamped in the past: dropped by the soft-FPS gate
            if (!cf.Running()) break;
        cf.releaseCustomCamBuffer();
    }
    NavData nav = cf.getNav();
    cf.CleanUp();
    std::cout << "frames delivered to the callback: " << g_calls << "  final Pos = " << nav.Pos[0] << " " << nav.Pos[1] << " "
              << nav.Pos[2] << "\n";
    return 0;
}
