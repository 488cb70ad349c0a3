// multi_device_replay — BASELINE configs[4] on the host side, in C++: N data sets replayed at once, one rebvo::REBVO object
// per sequence, sequence i on HIP device i % <devices> (`&GPU Device=` of the host mirror, rebvo/rebvo.h).  The sequences share
// nothing (the path does not shard inside a sequence), so there is no exchange between the objects;
// what the reference would need N processes of rebvorun for (app/rebvorun/main.cpp:58-140, one camera each) is N objects of
// the same library here, each with its own threads, its own device context and its own callback.
//
//   multi_device_replay [--devices D] [--group] [--dump PREFIX] <GlobalConfig_0> [<GlobalConfig_1> ...]
//
// Every config names its own data set (CameraType=2: DataSetDir / DataSetFile / TimeScale).  --devices D overrides the number of
// HIP devices the sequences are dealt over (default: all visible ones); a `&GPU Device=` in a config is replaced by the dealt
// ordinal.  --group: the sequences dealt to one device form one batch group there (&GPU BatchGroup, rebvo/rebvo.h: one shared context, one
// launch set per step for all of them, lock-step; a sequence leaves the group when its list ends) instead of a context each.
// Prints one line per sequence and the node aggregate: frames delivered to the callbacks / wall time from the first
// Init() to the last sequence's end.  --dump PREFIX writes PREFIX<i>.txt: frame id, time stamp, KeyLines, matches, EstimationOK,
// Pos, PoseLie, Vel per delivered frame (the first 14 columns of dataset_replay's dump).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rebvo/rebvo.h"

using namespace rebvo;

namespace {

struct Sequence {
    int id = 0, device = 0;
    std::unique_ptr<REBVO> vo;
    std::ofstream dump;
    int frames = 0, ok_frames = 0;
    double t_last_s = 0;    // wall time of the last delivered frame, from the common start
    std::chrono::steady_clock::time_point t0;

    bool callback(PipeBuffer &p) {   // runs on this object's output thread
        frames++;
        ok_frames += p.EstimationOK ? 1 : 0;
        t_last_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dump.is_open()) {
            dump << std::setprecision(17) << p.p_id << " " << p.t << " " << p.ef->KNum() << " " << p.ef->NumMatches() << " "
                 << (int)p.EstimationOK;
            for (int i = 0; i < 3; i++) dump << " " << p.nav.Pos[i];
            for (int i = 0; i < 3; i++) dump << " " << p.nav.PoseLie[i];
            for (int i = 0; i < 3; i++) dump << " " << p.nav.Vel[i];
            dump << "\n";
        }
        return true;
    }
};

}  // namespace

int main(int argn, char **argv) {
    int devices = -1;
    bool grouped = false;
    std::string dump_prefix;
    std::vector<std::string> configs;
    for (int i = 1; i < argn; i++) {
        const std::string a = argv[i];
        if (a == "--devices" && i + 1 < argn) devices = std::atoi(argv[++i]);
        else if (a == "--dump" && i + 1 < argn) dump_prefix = argv[++i];
        else if (a == "--group") grouped = true;
        else configs.push_back(a);
    }
    if (configs.empty()) {
        std::cout << "usage: multi_device_replay [--devices D] [--group] [--dump PREFIX] <GlobalConfig_0> [<GlobalConfig_1> ...]\n";
        return 2;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        std::cout << "multi_device_replay: no HIP device (there is no CPU path)\n";
        return 5;
    }
    if (devices < 1 || devices > visible) devices = visible;

    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::unique_ptr<Sequence>> seqs;
    for (size_t i = 0; i < configs.size(); i++) {
        REBVO parsed(configs[i].c_str());                 // the config parser of the library, as REBVO::REBVO(const char *) runs it
        if (!parsed.isInitOk()) {
            std::cout << "sequence " << i << ": config error in " << configs[i] << "\n";
            return 3;
        }
        REBVOParameters p = parsed.getParams();
        p.GpuDevice = (int)(i % (size_t)devices);          // sequence id -> device id
        if (grouped) {                                      // ... and the sequences of a device into one batch group
            p.GpuBatchGroup = "device" + std::to_string(p.GpuDevice);
            p.GpuBatchSize = (int)((configs.size() - (size_t)p.GpuDevice + (size_t)devices - 1) / (size_t)devices);
        }
        auto s = std::make_unique<Sequence>();
        s->id = (int)i;
        s->device = p.GpuDevice;
        s->t0 = t0;
        s->vo = std::make_unique<REBVO>(p);
        if (!s->vo->isInitOk()) {
            std::cout << "sequence " << i << ": parameters rejected\n";
            return 3;
        }
        if (!dump_prefix.empty()) s->dump.open(dump_prefix + std::to_string(i) + ".txt");
        s->vo->setOutputCallback(&Sequence::callback, s.get());
        seqs.push_back(std::move(s));
    }
    for (auto &s : seqs)
        if (!s->vo->Init()) {
            std::cout << "sequence " << s->id << ": Init() failed on device " << s->device << "\n";
            for (auto &o : seqs) o->vo->CleanUp();
            return 4;
        }
    for (bool any = true; any;) {                          // every sequence ends with its image list
        any = false;
        for (auto &s : seqs) any |= s->vo->Running();
        if (any) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    int total = 0;
    double t_end = 0;
    for (auto &s : seqs) {
        NavData nav = s->vo->getNav();
        s->vo->CleanUp();
        total += s->frames;
        t_end = s->t_last_s > t_end ? s->t_last_s : t_end;
        std::cout << "sequence " << s->id << " device " << s->device << ": frames delivered " << s->frames << " (EstimationOK " << s->ok_frames
                  << "), " << std::setprecision(6) << (s->t_last_s > 0 ? s->frames / s->t_last_s : 0.0) << " frames/s, final Pos = " << std::setprecision(17)
                  << nav.Pos[0] << " " << nav.Pos[1] << " " << nav.Pos[2] << "\n";
    }
    std::cout << "node aggregate: " << seqs.size() << " sequences on " << devices << " device(s), " << total << " frames in " << std::setprecision(6)
              << t_end << " s = " << (t_end > 0 ? total / t_end : 0.0) << " frames/s (includes context creation and image decoding)\n";
    return 0;
}
