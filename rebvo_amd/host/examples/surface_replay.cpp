// surface_replay — N rebvo::REBVO objects fed through the plugin surface only (requestCustomCamBuffer / releaseCustomCamBuffer,
// setOutputCallback, getNav: include/rebvo/rebvo.h:548-609 of the reference), optionally as ONE batch group (&GPU BatchGroup:
// the objects share a device context of N sequences, rebvo_amd/host/src/batch_group.cpp).  Used by tests/test_batch_group_gpu.py
// (per-object results against the reference) and by bench.py's `host_surface` leg (frames per second through the surface).
//
//   surface_replay <GlobalConfig> <frames.rgb24> <pool_frames> <objects> <frames_per_object> <t0> <dt>
//                  [--group NAME] [--callback] [--dump PREFIX] [--threads T] [--warmup W] [--leave I:F] [--step-mode] [--snapshot-at F] [--dup I:F] [--tint I:F]
//
// frames.rgb24 = pool_frames x ImageHeight x ImageWidth x 3 bytes.  Object i's frame k is pool frame tri(k + i): the triangle wave
// over the pool bench.py uses (forward then backward: continuous motion), every object at its own phase.  Frame k of every object
// carries the time stamp t0 + k dt.  --group puts all objects into one batch group of that name; without it every object has its
// own context.  --callback registers an output callback per object (KeyLines come back as AoS); --dump PREFIX makes it write
// PREFIX.<i>.txt, one line per delivered frame in custom_cam_replay's columns.  T producer threads feed the objects (object i
// belongs to thread i % T), each with one copyFrom() per frame like the reference's example.  The last line on stdout is JSON:
//   {"objects": N, "frames_per_object": K, "timed_frames": ..., "seconds": ..., "fps": ..., "callbacks": ..., "group": ...}
// --leave I:F: object I calls CleanUp() after its frame F-1 (a camera that goes away; the others carry on).
// --step-mode: object 0 runs frame by frame (toggleFrameByFrame; a helper thread calls advanceFrameByFrame() every millisecond).
// --dup I:F: object I submits one more frame in front of its frame F, stamped like frame F-1: the soft-FPS gate drops it (rebvo_first_t.cpp:172-177)
//   and the object's camera ring runs one entry ahead of the others' from then on (the group then copies its frames separately).
// --tint I:F: the first byte of object I's frame F is flipped (^ 0x80): a coloured pixel in an otherwise mono frame — that step crosses PCIe as RGB24.
// --stagger: object i's frame k carries the stamp t0 + dt (k + i) (object i enters the common time line i frames late: with one IMU file
// for all objects — ImuMode=2 — every object's images then agree with the IMU samples of its own stamps).
// --stereo PAIRS.rgb24: every object has a stereo pair (the config's StereoAvaiable=1 and &Stereo section): pool_frames pair images, pair
//   frame j belongs to pool frame j; each object's pair frame goes in through requestStereoCustomCamBuffer / releaseStereoCustomCamBuffer
//   right before its main frame, with the same stamp.  The dump's last column is PipeBuffer::stereo_match_num.
// --snapshot-at F: object 0's TakeSnapshot() is called before its frame F is submitted (Snap0.ppm in the working directory).
// timed over the frames after the first W of every object (default 0), from the submission of frame W to the moment every
// object's getNav() shows its last frame.
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rebvo/rebvo.h"

using namespace rebvo;

static int tri(long k, int n) {
    if (n < 2) return 0;
    const int p = 2 * (n - 1);
    const int r = (int)(k % p);
    return r < n ? r : p - r;
}

struct Sink {
    std::ofstream dump;
    std::atomic<int> calls{0};
    bool cb(PipeBuffer &p) {
        calls++;
        if (!dump.is_open()) {
            volatile int kn = p.ef->KNum();   // a consumer that looks at the edge map
            (void)kn;
            return true;
        }
        double sr = 0, ss = 0;
        for (KeyLine &kl : *p.ef) { sr += kl.rho; ss += kl.s_rho; }
        dump << std::setprecision(17) << p.p_id << " " << p.t << " " << p.ef->KNum() << " " << p.ef->NumMatches() << " " << (int)p.EstimationOK;
        for (int i = 0; i < 3; i++) dump << " " << p.nav.Pos[i];
        for (int i = 0; i < 3; i++) dump << " " << p.nav.PoseLie[i];
        for (int i = 0; i < 3; i++) dump << " " << p.nav.Vel[i];
        dump << " " << sr << " " << ss;
        // columns 16..48 as dataset_replay writes them: RotLie, RotGiro, g, scale, K, Kp, RKp, then the IMU-branch state (zero with ImuMode = 0), dt
        for (int i = 0; i < 3; i++) dump << " " << p.nav.RotLie[i];
        for (int i = 0; i < 3; i++) dump << " " << p.nav.RotGiro[i];
        for (int i = 0; i < 3; i++) dump << " " << p.nav.g[i];
        dump << " " << p.nav.scale << " " << p.K << " " << p.Kp << " " << p.RKp;
        for (int i = 0; i < 3; i++) dump << " " << p.imustate.Vg[i];
        for (int i = 0; i < 3; i++) dump << " " << p.imustate.Bg[i];
        for (int i = 0; i < 7; i++) dump << " " << p.imustate.X[i];
        for (int i = 0; i < 3; i++) dump << " " << p.imustate.b_est[i];
        for (int i = 0; i < 3; i++) dump << " " << p.imustate.u_est[i];
        dump << " " << p.dt << " " << p.stereo_match_num << "\n";
        return true;
    }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argn, char **argv) {
    if (argn < 8) {
        std::cout << "usage: surface_replay <GlobalConfig> <frames.rgb24> <pool_frames> <objects> <frames_per_object> <t0> <dt> "
                     "[--group NAME] [--callback] [--dump PREFIX] [--threads T] [--warmup W]\n";
        return 2;
    }
    const int pool_frames = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]);
    const double t0 = atof(argv[6]), dt = atof(argv[7]);
    std::string group, dump_prefix, pair_file;
    bool want_cb = false;
    int T = 1, W = 0, leave_obj = -1, leave_at = 0, snapshot_at = -1, dup_obj = -1, dup_at = 0, tint_obj = -1, tint_at = 0;
    bool step_mode = false, stagger = false;
    for (int a = 8; a < argn; a++) {
        const std::string s = argv[a];
        if (s == "--group" && a + 1 < argn) group = argv[++a];
        else if (s == "--callback") want_cb = true;
        else if (s == "--dump" && a + 1 < argn) { dump_prefix = argv[++a]; want_cb = true; }
        else if (s == "--threads" && a + 1 < argn) T = atoi(argv[++a]);
        else if (s == "--warmup" && a + 1 < argn) W = atoi(argv[++a]);
        else if (s == "--step-mode") step_mode = true;
        else if (s == "--stagger") stagger = true;
        else if (s == "--stereo" && a + 1 < argn) pair_file = argv[++a];
        else if (s == "--snapshot-at" && a + 1 < argn) snapshot_at = atoi(argv[++a]);
        else if (s == "--tint" && a + 1 < argn) { if (std::sscanf(argv[++a], "%d:%d", &tint_obj, &tint_at) != 2) return 2; }
        else if (s == "--dup" && a + 1 < argn) { if (std::sscanf(argv[++a], "%d:%d", &dup_obj, &dup_at) != 2) return 2; }
        else if (s == "--leave" && a + 1 < argn) { if (std::sscanf(argv[++a], "%d:%d", &leave_obj, &leave_at) != 2) return 2; }
        else { std::cout << "unknown argument " << s << "\n"; return 2; }
    }
    if (pool_frames < 1 || N < 1 || K < 2 || T < 1 || W < 0 || W >= K - 1) { std::cout << "bad counts\n"; return 2; }
    T = std::min(T, N);

    REBVO proto(argv[1]);
    if (!proto.isInitOk()) { std::cout << "config error\n"; return 3; }
    REBVOParameters prm = proto.getParams();
    if (!group.empty()) { prm.GpuBatchGroup = group; prm.GpuBatchSize = N; }
    const Size2D sz = prm.ImageSize;
    const size_t fb = (size_t)sz.w * sz.h * 3;
    std::vector<uint8_t> pool(fb * pool_frames);
    {
        std::ifstream in(argv[2], std::ios::binary);
        if (!in.is_open()) { std::cout << "cannot open " << argv[2] << "\n"; return 5; }
        in.read(reinterpret_cast<char *>(pool.data()), (std::streamsize)pool.size());
        if ((size_t)in.gcount() != pool.size()) { std::cout << "short read of " << argv[2] << "\n"; return 5; }
    }

    std::vector<uint8_t> pair_pool;
    if (!pair_file.empty()) {
        if (!prm.StereoAvaiable) { std::cout << "--stereo needs StereoAvaiable=1 in the config\n"; return 3; }
        pair_pool.resize(fb * pool_frames);
        std::ifstream in(pair_file, std::ios::binary);
        in.read(reinterpret_cast<char *>(pair_pool.data()), (std::streamsize)pair_pool.size());
        if (!in.is_open() || (size_t)in.gcount() != pair_pool.size()) { std::cout << "cannot read " << pair_file << "\n"; return 5; }
    }

    const double t_phase0 = now_s();
    auto phase = [&](const char *what) { if (getenv("SURFACE_REPLAY_PHASES")) std::fprintf(stderr, "surface_replay: %-28s at %8.3f s\n", what, now_s() - t_phase0); };
    std::vector<std::unique_ptr<REBVO>> obj;
    std::vector<std::unique_ptr<Sink>> sink;
    for (int i = 0; i < N; i++) {
        obj.emplace_back(new REBVO(prm));
        sink.emplace_back(new Sink);
        if (!obj[i]->isInitOk()) { std::cout << "object " << i << ": bad parameters\n"; return 3; }
        if (!dump_prefix.empty()) sink[i]->dump.open(dump_prefix + "." + std::to_string(i) + ".txt");
        if (want_cb) obj[i]->setOutputCallback(&Sink::cb, sink[i].get());
    }
    phase("objects constructed");
    for (int i = 0; i < N; i++)
        if (!obj[i]->Init()) { std::cout << "object " << i << ": Init failed: " << obj[i]->lastError() << "\n"; return 4; }
    phase("Init() of every object done");

    if (step_mode && !obj[0]->toggleFrameByFrame()) { std::cout << "toggleFrameByFrame did not switch on\n"; return 7; }
    std::atomic<bool> bad{false};
    std::atomic<int> at_warm{0};
    double t_start = 0;
    std::atomic<bool> go{false};
    // where the application's threads spend a frame: waiting for a free camera buffer, writing the frame (copyFrom), handing it over
    // (releaseCustomCamBuffer: the library's mono test + 8-bit plane).  Summed over the timed frames of all threads.
    std::vector<std::array<double, 3>> ptime(T, std::array<double, 3>{0, 0, 0});
    auto producer = [&](int tid) {
        for (int k = 0; k < K && !bad; k++) {
            if (k == W) {   // all producers line up behind the warm-up frames: the clock starts when the first timed frame is submitted
                if (++at_warm == T) { t_start = now_s(); go = true; }
                while (!go && !bad) std::this_thread::yield();
            }
            for (int i = tid; i < N && !bad; i += T) {
                if (i == leave_obj && k >= leave_at) {
                    if (k == leave_at) obj[i]->CleanUp();
                    continue;
                }
                if (i == dup_obj && k == dup_at && k > 0) {   // a frame the soft-FPS gate drops: same stamp as the one before
                    std::shared_ptr<Image<RGB24Pixel>> dp;
                    while (!obj[i]->requestCustomCamBuffer(dp, t0 + dt * (k - 1), 0.1))
                        if (!obj[i]->Running()) { bad = true; break; }
                    if (bad) break;
                    (*dp).copyFrom(reinterpret_cast<const RGB24Pixel *>(pool.data()));
                    obj[i]->releaseCustomCamBuffer();
                }
                if (i == 0 && k == snapshot_at) obj[0]->TakeSnapshot();
                if (!pair_pool.empty()) {   // the pair camera's frame of this instant
                    std::shared_ptr<Image<RGB24Pixel>> pp;
                    while (!obj[i]->requestStereoCustomCamBuffer(pp, stagger ? t0 + dt * (k + i) : t0 + dt * k, 0.1))
                        if (!obj[i]->Running()) { bad = true; break; }
                    if (bad) break;
                    (*pp).copyFrom(reinterpret_cast<const RGB24Pixel *>(pair_pool.data() + fb * tri((long)k + i, pool_frames)));
                    obj[i]->releaseStereoCustomCamBuffer();
                }
                std::shared_ptr<Image<RGB24Pixel>> ptr;
                const double tq0 = now_s();
                while (!obj[i]->requestCustomCamBuffer(ptr, stagger ? t0 + dt * (k + i) : t0 + dt * k, 0.1))
                    if (!obj[i]->Running()) { bad = true; break; }
                if (bad) break;
                const double tq1 = now_s();
                (*ptr).copyFrom(reinterpret_cast<const RGB24Pixel *>(pool.data() + fb * tri((long)k + i, pool_frames)));
                if (i == tint_obj && k == tint_at) *reinterpret_cast<uint8_t *>((*ptr).Data()) ^= 0x80;
                const double tq2 = now_s();
                obj[i]->releaseCustomCamBuffer();
                if (k >= W) { const double tq3 = now_s(); ptime[tid][0] += tq1 - tq0; ptime[tid][1] += tq2 - tq1; ptime[tid][2] += tq3 - tq2; }
            }
        }
    };
    // step mode: somebody presses "advance" every millisecond (the flag is a level, not a counter: rebvo.h:485-488)
    std::atomic<bool> stepping{step_mode};
    std::thread stepper([&] { while (stepping) { obj[0]->advanceFrameByFrame(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); } });
    std::vector<std::thread> thr;
    for (int t = 0; t < T; t++) thr.emplace_back(producer, t);
    for (auto &t : thr) t.join();
    phase("producers done");
    // every object's record of its last frame
    const double deadline = now_s() + 30;
    for (int i = 0; i < N && !bad; i++) {
        const double t_last = stagger ? t0 + dt * (K - 1 + i) : t0 + dt * (K - 1);
        while (i != leave_obj && obj[i]->getNav().t < t_last - 1e-9 * (1 + std::fabs(t_last))) {
            if (!obj[i]->Running() || now_s() > deadline) { bad = true; break; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    const double seconds = now_s() - t_start;
    stepping = false;
    stepper.join();
    phase("last records seen");
    std::vector<NavData> navs;
    for (int i = 0; i < N; i++) navs.push_back(obj[i]->getNav());
    for (int i = 0; i < N; i++) obj[i]->CleanUp();
    phase("CleanUp() of every object done");
    int calls = 0;
    for (int i = 0; i < N; i++) calls += sink[i]->calls;
    if (bad) { std::cout << "an object stopped before its last frame\n"; return 6; }
    for (int i = 0; i < N; i++)
        std::cout << "object " << i << " final Pos = " << std::setprecision(17) << navs[i].Pos[0] << " " << navs[i].Pos[1] << " " << navs[i].Pos[2] << "\n";
    const long timed = (long)N * (K - W);
    double pw = 0, pc = 0, pr = 0;
    for (auto &a : ptime) { pw += a[0]; pc += a[1]; pr += a[2]; }
    std::printf("{\"objects\": %d, \"frames_per_object\": %d, \"timed_frames\": %ld, \"seconds\": %.6f, \"fps\": %.1f, \"callbacks\": %d, "
                "\"group\": %s, \"producer_threads\": %d, \"ms_per_step\": %.4f, \"producer_us_per_frame\": {\"wait_for_buffer\": %.1f, \"copyFrom\": %.1f, "
                "\"releaseCustomCamBuffer\": %.1f}, \"producer_busy_share\": %.3f}\n",
                N, K, timed, seconds, timed / seconds, calls, group.empty() ? "null" : ("\"" + group + "\"").c_str(), T, seconds / (K - W) * 1e3,
                pw / timed * 1e6, pc / timed * 1e6, pr / timed * 1e6, (pc + pr) / (seconds * T));
    return 0;
}
