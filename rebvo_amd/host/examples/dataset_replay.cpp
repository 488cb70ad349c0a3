// dataset_replay — what `rebvorun <GlobalConfig>` does for CameraType=2 (app/rebvorun/main.cpp:58-140), headless:
// the library reads the image list itself (DataSetCam), tracks every frame on the GPU and calls back.
//
//   dataset_replay <GlobalConfig> [dump.txt]       (DataSetDir / DataSetFile / TimeScale come from the config)
//   dataset_replay --decode <image> <out.rgb24>    (image reader check: writes w*h*3 raw bytes, prints "w h")
#include <chrono>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <thread>

#include "rebvo/datasetcam.h"
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
    // columns 16..: RotLie, RotGiro, g, scale, K, Kp, RKp, then the IMU-branch state (zero with ImuMode = 0)
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
    if (argn >= 4 && std::string(argv[1]) == "--decode") {
        std::vector<RGB24Pixel> px;
        unsigned w, h;
        std::string err;
        if (!LoadImageRGB24(argv[2], px, w, h, err)) { std::cout << err << "\n"; return 6; }
        std::ofstream(argv[3], std::ios::binary).write(reinterpret_cast<const char *>(px.data()), (std::streamsize)px.size() * 3);
        std::cout << w << " " << h << "\n";
        return 0;
    }
    if (argn < 2) { std::cout << "usage: dataset_replay <GlobalConfig> [dump.txt]\n"; return 2; }
    REBVO cf(argv[1]);
    if (!cf.isInitOk()) { std::cout << "config error\n"; return 3; }
    if (argn > 2) g_dump.open(argv[2]);
    cf.setOutputCallback(&callback);
    if (!cf.Init()) return 4;
    while (cf.Running()) std::this_thread::sleep_for(std::chrono::milliseconds(5));   // ends with the image list
    NavData nav = cf.getNav();
    cf.CleanUp();
    std::cout << "frames delivered to the callback: " << g_calls << "  final Pos = " << nav.Pos[0] << " " << nav.Pos[1] << " "
              << nav.Pos[2] << "\n";
    return 0;
}
