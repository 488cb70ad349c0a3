// batch_group.cpp — one device context shared by N rebvo::REBVO objects (CameraType 3 or a DataSetCam; ImuMode 0, or — all members
// alike — 1 / 2 with the IMU branch of SecondThread batched on the device, round 6; mono, or — all members alike, ImuMode 0 — a stereo
// rig each: one pair frame per accepted main frame (rebvo_first_t.cpp:183-199) through a second page-locked ring into the context's pair
// slot, the stereo steps of SecondThread (rebvo_second_t.cpp:465-486) inside the group's edgehip_process_frame), and the pipelined
// frame loop behind them.  The plugin surface stays per object (requestCustomCamBuffer / releaseCustomCamBuffer,
// setOutputCallback, getNav: include/rebvo/rebvo.h:548-609 of the reference); what FirstThr and SecondThread do per frame
// (src/rebvo/rebvo_first_t.cpp:87-337, src/rebvo/rebvo_second_t.cpp:43-636) happens once per STEP for every member at once:
//
//   gather    the newest frame of every running member's camera ring (customcam.cpp:56-68: 1 ms time-outs; the soft-FPS drop
//             of rebvo_first_t.cpp:146,172-177 per member)
//   enqueue   the members' frames go out of ONE page-locked ring the group owns ([ring entry][member][frame]: every member's camera
//             buffers are views of it, so the application's copyFrom() writes where the DMA reads, and members that are at the
//             same ring entry — all of them, unless somebody dropped a frame — go up in a single asynchronous copy on the upload
//             stream, under the kernels of the frames before), then one edgehip_process_frame for all members
//   look ahead: if every member's next frame is waiting already, it goes up at once, behind this step's copy (the link stays busy
//             while this thread turns to the results of earlier steps)
//   release   this step's camera buffers, once the copies have read them (edgehip_upload_wait: the frames themselves are still
//             running, and so may the next step's copies)
//   complete  the PREVIOUS step: its per-sequence records out of the device's nav log (edgehip_read_nav_log waits for that frame
//             only), NavData / PipeBuffer of every member, and the hand-off of the frame before to the member's output thread —
//             with its KeyLines as AoS only if that member has a callback
//
// so the host works on earlier steps' results while the device runs the newest, and frame k+1 crosses PCIe under frame k: the
// reference's T0 || T1.  Up to two steps stay in flight when nobody has a callback (a single camera's frame is ~40 dependent
// launches: the host needs as long to enqueue one as the device to run it, and one step of slack is not enough to keep the device
// fed) — with callbacks too (round 6): the edge map a callback receives lives in a ring slot the step after next overwrites, so its
// AoS KeyLine lists are packed on the device right behind the step that finishes with the slot (edgehip_export_keylines: in-stream,
// a staging ring of its own), copied out on a stream of their own into the members' page-locked PipeBuffer arrays while the next
// steps run (edgehip_export_fetch, issued one completion ahead, as soon as the list lengths are known), and only waited for when the
// frame is handed to the member's output thread (edgehip_export_wait).  When no further frame is waiting, everything in flight is
// completed at once — a lone camera at 20 Hz sees its record as soon as the frame is done, not a frame later.
//
// Hand-off order is the reference's (rebvo_second_t.cpp:622-623): frame j reaches a member's callback after frame j+1 has been
// tracked against it, carrying its own record and its edge map as the tracker left it; the last frame is never delivered.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "edgehip.h"
#include "rebvo/datasetcam.h"
#include "rebvo/rebvo.h"
#include "rebvo_internal.h"

extern "C" int rebvo_pack_mono(const uint8_t *rgb, size_t npix, uint8_t *grey);   // mono_pack.cpp

namespace rebvo {

class REBVO::BatchGroup {
public:
    struct Seat {
        REBVO *cf = nullptr;
        bool running = false;      // attached, tracked by the group thread
        bool closed = true;        // the group thread no longer touches cf
        std::thread out_thread;    // the member's ThirdThread
        customCam::CustomCamPipeBuffer *cbuf = nullptr;   // frame gathered for the step being assembled
        customCam::CustomCamPipeBuffer *chold = nullptr;  // frame of the step launched last: held until its copy has read it
        double t_frame = 0, t0 = 0;
        int p_num = 0;
        long frames = 0;           // frames of this member enqueued so far
        PipeBuffer *buf_of[4] = {nullptr, nullptr, nullptr, nullptr};   // [step & 3] PipeBuffer of a step in flight (released by player 0,
                                                                        // not yet requested by player 1): later steps are enqueued first
        int ring_idx = -1;         // ring entry of the gathered frame (page-locked ring), -1: a heap image
        uint8_t mono_of[CCAMBUFSIZE] = {};   // per ring entry: the frame in it is mono and its 8-bit plane is in grey_ring (written by the
                                             // application's thread before it releases the entry, read by the group thread after it took it)
        bool mono = false;         // ... of the gathered frame
        // With an output callback (or a pending snapshot) PipeBuffer::imgc must hold the frame.  The application's thread copies it
        // into side_img[ring entry] when it releases the camera buffer, and the group thread SWAPS that image into the step's
        // PipeBuffer (1 MB per member and step that the group thread used to copy itself).
        Image<RGB24Pixel> *side_img[CCAMBUFSIZE] = {};
        uint8_t side_ok[CCAMBUFSIZE] = {};
        bool kl_pinned = false;    // the KeyLine arrays of this member's PipeBuffers are page-locked (edgehip_register_host): AoS lists
                                   // for its callback land in them without a staging copy
        bool step_granted = false; // frame-by-frame mode: the application's "advance" has been taken for the frame about to be grabbed
        bool leaving = false;      // cf->quit seen: the seat closes once the step in flight (which may carry its last frame) is done
        // StereoAvaiable: the pair frame that goes with the gathered main frame, and the ones of the steps in flight ([step & 3]).  A pair
        // frame's copy into the ONE pair slot waits for the frame before to finish, so "the copy has read it" is known without another
        // wait only when the step's record is read (complete()): the buffer is held until then — three of the ring's four at most
        customCam::CustomCamPipeBuffer *cbuf_pair = nullptr, *pair_of[4] = {nullptr, nullptr, nullptr, nullptr};
        int pair_idx = -1;         // entry of the pair ring the gathered pair frame sits in (-1: a heap image)
        uint8_t pair_mono_of[CCAMBUFSIZE] = {};
        bool pair_mono = false;
        IntegratedImuData imu_data;   // ImuMode > 0: the integrated IMU data of the interval that ends with the gathered frame (rebvo_first_t.cpp:294-304)
        bool out_inline = false;   // nobody listens to this member (no callback, no log file, no snapshot pending): it has no output thread,
                                   // the group's thread passes its frames through the third player's position itself
        bool have_prev = false;    // a completed frame waits (as player 1's next buffer) for its successor before it is delivered
        int slot_prev = -1;        // ring slot of that frame
    };

    std::string name;
    int cap = 0, device = 0;
    edgehip_params hp;
    int tracker_bits = 64;         // &GPU TrackerPrecision of the member that created the context
    bool imu_mode = false;         // every member has ImuMode 1 or 2: the context runs the device-side IMU branch (edgehip_imu_enable)
    edgehip_imu_params ip;
    std::vector<edgehip_imu_integrated> imu_in;   // [cap] what edgehip_set_imu takes for the step being launched
    std::vector<edgehip_nav_imu> navs_imu;        // [cap] the IMU half of a completed step's records
    bool stereo = false;           // every member has a stereo pair (ImuMode 0): a fourth slot behind the ring holds the pair images' edge maps
    float stereo_cam[4] = {0, 0, 0, 0};   // &Stereo PPx, PPy, ZfX, ZfY (the same for every member: edgehip_set_slot_camera is per context)
    std::vector<int32_t> stereo_nm;       // [cap] stereo_match_num of a completed step
    static constexpr int kPairSlot = 3;
    edgehip_ctx *hip = nullptr;
    std::vector<Seat> seats;
    int attached = 0;
    std::thread thr;
    std::mutex mut;
    std::condition_variable cv;
    bool started = false, failed = false;
    std::string error;
    // The page-locked rings live as long as anybody can reach them: the group holds one reference, every camera-buffer view handed
    // out through requestCustomCamBuffer another (the application's shared_ptr may outlive CleanUp(), as a heap image's would).
    struct RingOwner {
        uint8_t *ring = nullptr, *grey = nullptr, *pair = nullptr, *pair_grey = nullptr;
        ~RingOwner() {
            for (uint8_t *p : {ring, grey, pair, pair_grey})
                if (p) edgehip_free_pinned(p);
        }
    };
    std::shared_ptr<RingOwner> ring_owner;
    uint8_t *ring = nullptr;       // page-locked [CCAMBUFSIZE][cap][frame]: the members' camera buffers (null: heap images, staged uploads)
    size_t frame_bytes = 0;
    uint8_t *grey_ring = nullptr;  // page-locked [CCAMBUFSIZE][cap][w * h]: 8-bit planes of the mono frames among them (null: MonoUpload off)
    size_t grey_bytes = 0;
    uint8_t *pair_ring = nullptr, *pair_grey_ring = nullptr;   // the same two rings for the members' pair cameras (StereoAvaiable)
    long mono_steps = 0;           // steps that went up as 8-bit planes
    long newest = -1;              // newest step enqueued
    // where the group thread's time goes (REBVO_GROUP_TIMING=1 prints it when the thread ends)
    struct Timing { double gather = 0, upload = 0, process = 0, buffers = 0, ahead = 0, held = 0, records = 0, handoff = 0; long steps = 0, ahead_hits = 0; } tm;
    std::vector<std::array<double, 5>> tlog;   // REBVO_GROUP_TIMING=2: when things happened on the group thread, step by step

    // ---- registry of named groups ----
    static std::mutex &regMutex() { static std::mutex m; return m; }
    static std::map<std::string, BatchGroup *> &registry() { static std::map<std::string, BatchGroup *> r; return r; }

    static constexpr int kNavLog = 8;

    // AoS KeyLine lists for the members' callbacks, one export per step (edgehip_export_keylines right behind the step's
    // edgehip_process_frame): which seats it covers, and whether its copies have been enqueued yet
    struct Export { long step = -1; int ticket = 0; std::vector<int32_t> seats; bool fetched = false; };
    Export exp_of[4];              // [step & 3]
    int cb_depth = 2;              // steps in flight when somebody has a callback (REBVO_GROUP_CB_DEPTH=1: the round-5 behaviour, A/B)

    void threadMain();
    int exportFetch(Export &ex, const std::vector<int32_t> &kn, const std::vector<edgehip_keyline *> &dst);
    void dropExports();
    bool gather(bool block, bool &any_running, bool &any_leaving);
    int upload(std::vector<double> &ts, int &slot);
    int launch(long step, const std::vector<double> &ts);
    int releaseHeld(int slot);
    int complete(long step, int slot, std::vector<edgehip_nav> &navs);
    uint8_t *ringImage(int entry, int seat) { return ring + ((size_t)entry * cap + seat) * frame_bytes; }
    uint8_t *greyImage(int entry, int seat) { return grey_ring + ((size_t)entry * cap + seat) * grey_bytes; }
    int ringEntryOf(const uint8_t *p) const { return ring && p >= ring && p < ring + frame_bytes * CCAMBUFSIZE * cap ? (int)((size_t)(p - ring) / (frame_bytes * cap)) : -1; }
    uint8_t *pairImage(int entry, int seat) { return pair_ring + ((size_t)entry * cap + seat) * frame_bytes; }
    uint8_t *pairGrey(int entry, int seat) { return pair_grey_ring + ((size_t)entry * cap + seat) * grey_bytes; }
    int pairEntryOf(const uint8_t *p) const {
        return pair_ring && p >= pair_ring && p < pair_ring + frame_bytes * CCAMBUFSIZE * cap ? (int)((size_t)(p - pair_ring) / (frame_bytes * cap)) : -1;
    }
    int uploadRuns(int slot, bool pair);
    void closeSeat(Seat &st);
    void pinKeyLines(Seat &st, bool pin);
};

// ---- attach / detach (application threads) -----------------------------------------------------------------------------
bool REBVO::groupAttach() {
    edgehip_params hp;
    detail::fill_hip_params(params, hp);
    hp.stereo_available = params.StereoAvaiable ? 1 : 0;
    const bool stereo = params.StereoAvaiable;
    const float stereo_cam[4] = {params.pp_x_stereo, params.pp_y_stereo, params.z_f_x_stereo, params.z_f_y_stereo};
    const bool named = !params.GpuBatchGroup.empty();
    const int want = named ? params.GpuBatchSize : 1;
    const bool imu_mode = params.ImuMode > 0;
    edgehip_imu_params ip;
    std::memset(&ip, 0, sizeof ip);
    if (imu_mode) {   // the &IMU keys SecondThread uses (include/rebvo/rebvo.h:172-199 of the reference), as edgehip.h names them
        ip.giro_meas_std = params.GiroMeasStdDev; ip.giro_bias_std = params.GiroBiasStdDev;
        ip.init_bias = params.InitBias ? 1 : 0; ip.init_bias_frame_num = params.InitBiasFrameNum;
        for (int i = 0; i < 3; i++) ip.bias_init_guess[i] = params.BiasInitGuess[i];
        ip.acel_meas_std = params.AcelMeasStdDev; ip.g_module = params.g_module; ip.g_module_uncer = params.g_module_uncer;
        ip.g_uncert = params.g_uncert; ip.vbias_std = params.VBiasStdDev;
        ip.scale_std_mult = params.ScaleStdDevMult; ip.scale_std_max = params.ScaleStdDevMax; ip.scale_std_init = params.ScaleStdDevInit;
    }
    auto fail = [&](const std::string &msg) {
        last_error = msg;
        std::cout << last_error << "\n";
        return false;
    };
    if (want < 1) return fail("REBVO(hip): &GPU BatchGroup needs BatchSize >= 1 (the number of objects that share the context)");
    std::unique_lock<std::mutex> reg(BatchGroup::regMutex());
    BatchGroup *g = nullptr;
    if (named) {
        auto it = BatchGroup::registry().find(params.GpuBatchGroup);
        if (it != BatchGroup::registry().end()) g = it->second;
    }
    if (!g) {
        g = new BatchGroup;
        g->name = params.GpuBatchGroup;
        g->cap = want;
        g->device = params.GpuDevice;
        g->hp = hp;
        g->seats.resize(want);
        // ring of 3 frame slots per sequence, `want` sequences.  No CPU fallback: fail loudly.
        g->imu_mode = imu_mode;
        g->ip = ip;
        g->imu_in.resize(want);
        g->navs_imu.resize(want);
        g->stereo = stereo;
        std::memcpy(g->stereo_cam, stereo_cam, sizeof stereo_cam);
        g->stereo_nm.assign(want, 0);
        for (edgehip_imu_integrated &r : g->imu_in) {   // (a seat nobody sits on: a record that integrates to nothing)
            std::memset(&r, 0, sizeof r);
            r.n = 1; r.dt = 1.0 / params.config_fps;
            r.Rot[0] = r.Rot[4] = r.Rot[8] = 1;
        }
        const bool dbg = getenv("REBVO_GROUP_TIMING") && atoi(getenv("REBVO_GROUP_TIMING")) >= 3;
        const double tc0 = detail::now_s();
        // with stereo pairs one more slot, behind the ring, holds the pair images' edge maps (as REBVO::Init does for an object alone)
        int rc = edgehip_create(&hp, want, stereo ? 4 : 3, params.GpuDevice, &g->hip);
        if (dbg) std::fprintf(stderr, "REBVO(hip) group '%s': edgehip_create(%d sequences) took %.2f s\n", g->name.c_str(), want, detail::now_s() - tc0);
        g->tracker_bits = params.GpuTrackerPrecision;
        if (rc == 0 && params.GpuTrackerPrecision != 64) rc = edgehip_set_tracker_precision(g->hip, params.GpuTrackerPrecision);
        if (rc == 0 && imu_mode) rc = edgehip_imu_enable(g->hip, &ip);
        if (rc == 0 && stereo) {   // the pair camera's intrinsics for stage A of the pair slot; rig and search radius of rebvo_second_t.cpp:466-473
            rc = edgehip_set_slot_camera(g->hip, BatchGroup::kPairSlot, stereo_cam[0], stereo_cam[1], stereo_cam[2], stereo_cam[3]);
            if (rc == 0) rc = edgehip_set_stereo_rig(g->hip, BatchGroup::kPairSlot, kTCam2Pair, kRCam2Pair, 100.0);
        }
        if (rc == 0) rc = edgehip_set_nav_log(g->hip, BatchGroup::kNavLog);
        g->frame_bytes = (size_t)params.ImageSize.w * params.ImageSize.h * sizeof(RGB24Pixel);
        void *ringp = nullptr;
        if (rc == 0 && edgehip_alloc_pinned(g->frame_bytes * CCAMBUFSIZE * want, &ringp) == 0) g->ring = static_cast<uint8_t *>(ringp);
        g->grey_bytes = (size_t)params.ImageSize.w * params.ImageSize.h;
        void *greyp = nullptr;
        if (rc == 0 && g->ring && params.GpuMonoUpload && edgehip_alloc_pinned(g->grey_bytes * CCAMBUFSIZE * want, &greyp) == 0) g->grey_ring = static_cast<uint8_t *>(greyp);
        if (rc == 0 && stereo && g->ring) {   // the pair cameras' rings (without them: heap images, staged uploads)
            void *pp = nullptr, *pg = nullptr;
            if (edgehip_alloc_pinned(g->frame_bytes * CCAMBUFSIZE * want, &pp) == 0) g->pair_ring = static_cast<uint8_t *>(pp);
            if (g->pair_ring && g->grey_ring && edgehip_alloc_pinned(g->grey_bytes * CCAMBUFSIZE * want, &pg) == 0) g->pair_grey_ring = static_cast<uint8_t *>(pg);
        }
        g->ring_owner = std::make_shared<BatchGroup::RingOwner>();
        g->ring_owner->ring = g->ring;
        g->ring_owner->grey = g->grey_ring;
        g->ring_owner->pair = g->pair_ring;
        g->ring_owner->pair_grey = g->pair_grey_ring;
        if (dbg) std::fprintf(stderr, "REBVO(hip) group '%s': page-locked rings %s / %s after %.2f s\n", g->name.c_str(), g->ring ? "ok" : "NONE", g->grey_ring ? "ok" : "none",
                              detail::now_s() - tc0);
        if (rc != 0) {
            const std::string msg = std::string("REBVO(hip): edgehip_create failed: ") + edgehip_last_error();
            g->ring_owner.reset();
            if (g->hip) edgehip_destroy(g->hip);
            delete g;
            return fail(msg);
        }
        if (named) BatchGroup::registry()[g->name] = g;
    } else {
        if (g->cap != want || g->device != params.GpuDevice || std::memcmp(&g->hp, &hp, sizeof hp) != 0)
            return fail("REBVO(hip): BatchGroup '" + g->name + "': every member needs the same BatchSize, Device, camera and detector / tracker parameters");
        if (g->tracker_bits != params.GpuTrackerPrecision)
            return fail("REBVO(hip): BatchGroup '" + g->name + "': every member needs the same &GPU TrackerPrecision");
        if (g->imu_mode != imu_mode || (imu_mode && std::memcmp(&g->ip, &ip, sizeof ip) != 0))
            return fail("REBVO(hip): BatchGroup '" + g->name + "': every member needs the same ImuMode (0, or 1 / 2) and the same &IMU filter parameters");
        if (g->stereo != stereo || (stereo && std::memcmp(g->stereo_cam, stereo_cam, sizeof stereo_cam) != 0))
            return fail("REBVO(hip): BatchGroup '" + g->name + "': every member needs the same StereoAvaiable and the same &Stereo intrinsics");
    }
    std::unique_lock<std::mutex> lk(g->mut);
    if (g->started && g->attached >= g->cap) {
        lk.unlock();
        return fail("REBVO(hip): BatchGroup '" + g->name + "' is full (a group is made of the objects that Init() before it starts; nobody joins later)");
    }
    int seat = -1;
    for (int i = 0; i < g->cap; i++)
        if (!g->seats[i].cf) { seat = i; break; }
    if (seat < 0 || g->started) {
        lk.unlock();
        return fail("REBVO(hip): BatchGroup '" + g->name + "' has no free seat");
    }
    BatchGroup::Seat &st = g->seats[seat];
    st = BatchGroup::Seat();
    st.cf = this;
    st.running = true;
    st.closed = false;
    group = g;
    group_seat = seat;
    if (g->ring) {   // the camera ring's images become views of the group's page-locked ring: entry j of this member = ring[j][seat]
        std::shared_ptr<BatchGroup::RingOwner> owner = g->ring_owner;
        for (unsigned j = 0; j < cam_pipe.Size(); j++)
            cam_pipe[j].img = std::shared_ptr<Image<RGB24Pixel>>(new Image<RGB24Pixel>(reinterpret_cast<RGB24Pixel *>(g->ringImage((int)j, seat)), params.ImageSize),
                                                                 [owner](Image<RGB24Pixel> *p) { delete p; });   // the view keeps the ring alive
        cam_pinned = true;
    } else {   // no page-locked ring (the allocation failed): heap images, staged uploads
        for (unsigned j = 0; j < cam_pipe.Size(); j++)
            if (!cam_pipe[j].img) cam_pipe[j].img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);
    }
    if (stereo && g->pair_ring) {
        std::shared_ptr<BatchGroup::RingOwner> owner = g->ring_owner;
        for (unsigned j = 0; j < cam_pipe_stereo.Size(); j++)
            cam_pipe_stereo[j].img = std::shared_ptr<Image<RGB24Pixel>>(new Image<RGB24Pixel>(reinterpret_cast<RGB24Pixel *>(g->pairImage((int)j, seat)), params.ImageSize),
                                                                        [owner](Image<RGB24Pixel> *p) { delete p; });
    } else if (stereo) {
        for (unsigned j = 0; j < cam_pipe_stereo.Size(); j++)
            if (!cam_pipe_stereo[j].img) cam_pipe_stereo[j].img = std::make_shared<Image<RGB24Pixel>>(params.ImageSize);
    }
    quit = false;
    // The output thread exists to call the callback, write the log and save snapshots (rebvo_third_t.cpp:174-343).  A member that has
    // none of these gets no thread (a thousand cameras are not a thousand threads woken per step); the group's thread starts one the
    // moment a callback or a snapshot request shows up (launch()).
    st.out_inline = !haveCallBack() && !params.SaveLog && !saveImg && !params.cpuSetAffinity;
    if (!st.out_inline) st.out_thread = std::thread(ThirdThread, this);
    if (dscam) feeder = std::thread(FeedThread, this);
    g->attached++;
    if (g->attached == g->cap) {   // the group is complete: its tracker thread starts
        g->started = true;
        g->thr = std::thread([g] { g->threadMain(); });
    }
    return true;
}

// A DataSetCam as a group member: the list's images go through the object's own camera ring, as an application's would
// (DataSetCam::GrabBuffer decodes; one copy into the page-locked ring).  At the end of the list — or at an image that does not load:
// the camera's error state — the frames still in the ring are let through (a full round of requests stamped in the past, which the
// soft-FPS gate drops, succeeds only when everything before has been taken), then the object quits, as REBVO does on a camera
// error (rebvo_first_t.cpp:165-170): the group lets it go and carries on.
void REBVO::FeedThread(REBVO *cf) {
    auto slot = [&](double stamp) -> customCam::CustomCamPipeBuffer * {
        customCam::CustomCamPipeBuffer *b = nullptr;
        while (!cf->quit && (b = cf->cam_pipe.RequestBufferTimeoutable(0, 0.01)) == nullptr) {}
        if (b) b->timestamp = stamp;
        return b;
    };
    while (!cf->quit) {
        double ts = 0;
        const RGB24Pixel *data = cf->dscam->GrabBuffer(ts, false);
        if (!data) break;
        customCam::CustomCamPipeBuffer *b = slot(ts);
        if (!b) return;
        b->img->copyFrom(data);
        cf->groupFrameWritten(b);
        cf->cam_pipe.ReleaseBuffer(0);
        cf->dscam->ReleaseBuffer();
        if (cf->dscam_pair) {   // the pair list, image by image beside the main one (REBVO::initPairCamera, rebvo_first_t.cpp:64-76, 183-199)
            double tp = 0;
            const RGB24Pixel *dp = cf->dscam_pair->GrabBuffer(tp, false);
            if (!dp) { std::cout << "bye bye cruel world on stereo\n"; break; }
            customCam::CustomCamPipeBuffer *pb = nullptr;
            while (!cf->quit && (pb = cf->cam_pipe_stereo.RequestBufferTimeoutable(0, 0.01)) == nullptr) {}
            if (!pb) return;
            pb->timestamp = tp;
            pb->img->copyFrom(dp);
            cf->groupFrameWritten(pb, true);
            cf->cam_pipe_stereo.ReleaseBuffer(0);
            cf->dscam_pair->ReleaseBuffer();
        }
    }
    for (int i = 0; i < CCAMBUFSIZE && !cf->quit; i++)
        if (slot(-1e300)) cf->cam_pipe.ReleaseBuffer(0);
    cf->quit = true;
}

void REBVO::groupDetach() {
    BatchGroup *g = group;
    if (!g) return;
    if (feeder.joinable()) feeder.join();   // (quit is up: CleanUp() set it)
    BatchGroup::Seat &st = g->seats[group_seat];
    bool last = false;
    {
        std::unique_lock<std::mutex> lk(g->mut);
        if (!g->started) {
            // the group never became complete: nobody is tracking this member; pass the quit flag down its ring ourselves
            PipeBuffer &b = pipe.RequestBuffer(0);
            b.quit = true;
            pipe.ReleaseBuffer(0);
            PipeBuffer &b1 = pipe.RequestBuffer(1);
            b1.quit = true;
            pipe.ReleaseBuffer(1);
            st.running = false;
            st.closed = true;
        } else {
            g->cv.wait(lk, [&] { return st.closed; });   // the group thread saw cf->quit and let go of this member
        }
    }
    if (st.out_thread.joinable()) st.out_thread.join();
    for (Image<RGB24Pixel> *&im : st.side_img) { delete im; im = nullptr; }
    {
        std::unique_lock<std::mutex> reg(BatchGroup::regMutex());
        std::unique_lock<std::mutex> lk(g->mut);
        st.cf = nullptr;
        g->attached--;
        last = g->attached == 0;
        if (last && !g->name.empty()) BatchGroup::registry().erase(g->name);
    }
    group = nullptr;
    group_seat = -1;
    if (cam_pinned) {   // the ring goes with the group (with the last view of it): this object has no camera buffers until it is Init()ed again
        for (unsigned j = 0; j < cam_pipe.Size(); j++) cam_pipe[j].img.reset();
        if (params.StereoAvaiable)
            for (unsigned j = 0; j < cam_pipe_stereo.Size(); j++) cam_pipe_stereo[j].img.reset();
        cam_pinned = false;
    }
    if (last) {   // the last member out stops the thread and frees the context
        if (g->thr.joinable()) g->thr.join();
        if (g->hip) edgehip_destroy(g->hip);
        g->ring_owner.reset();   // freed with the last view of it (normally: here)
        delete g;
    }
}

// The application (or the feeder thread) has written a frame into one of this member's camera buffers and is about to release it:
// if the frame is a mono camera's, its 8-bit plane goes into the group's second page-locked ring (src/mono_pack.cpp).
void REBVO::groupFrameWritten(customCam::CustomCamPipeBuffer *b, bool pair) {
    BatchGroup *g = group;
    if (!g || !b || !b->img) return;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(b->img->Data());
    if (pair) {   // the pair camera's frame: its 8-bit plane only (PipeBuffer::imgc_pair is copied by the group's thread, for listeners)
        const int pe = g->pairEntryOf(p);
        if (pe >= 0 && g->pair_grey_ring) g->seats[group_seat].pair_mono_of[pe] = (uint8_t)rebvo_pack_mono(p, g->grey_bytes, g->pairGrey(pe, group_seat));
        return;
    }
    const int entry = g->ringEntryOf(p);
    if (entry < 0) return;
    BatchGroup::Seat &st = g->seats[group_seat];
    if (g->grey_ring) st.mono_of[entry] = (uint8_t)rebvo_pack_mono(p, g->grey_bytes, g->greyImage(entry, group_seat));
    st.side_ok[entry] = 0;
    if (haveCallBack() || saveImg) {   // the frame for PipeBuffer::imgc, copied here instead of on the group's thread
        if (!st.side_img[entry]) st.side_img[entry] = new Image<RGB24Pixel>(params.ImageSize);
        std::memcpy(st.side_img[entry]->Data(), p, g->frame_bytes);
        st.side_ok[entry] = 1;
    }
}

// ---- the group's tracker thread -------------------------------------------------------------------------------------------
// Page-lock (or release) the KeyLine arrays of a member's PipeBuffers.  A range the driver refuses stays pageable: the lists then
// come through the library's staging buffer as before.
void REBVO::BatchGroup::pinKeyLines(Seat &st, bool pin) {
    REBVO *cf = st.cf;
    for (unsigned j = 0; j < cf->pipe.Size(); j++) {
        if (pin) cf->pipe[j].ef->ensureKeyLines();
        std::vector<KeyLine> &kl = cf->pipe[j].ef->kl;
        if (kl.empty()) continue;
        if (pin) (void)edgehip_register_host(kl.data(), kl.size() * sizeof(KeyLine));
        else (void)edgehip_unregister_host(kl.data());
    }
    st.kl_pinned = pin;
}

void REBVO::BatchGroup::closeSeat(Seat &st) {
    // shutdown of one member: the frame still held for it is not delivered (as in the reference); pass the quit flag on
    REBVO *cf = st.cf;
    if (st.kl_pinned) pinKeyLines(st, false);
    if (st.chold) { cf->cam_pipe.ReleaseBufferAt(1, st.chold); st.chold = nullptr; }
    if (st.cbuf) { cf->cam_pipe.ReleaseBufferAt(1, st.cbuf); st.cbuf = nullptr; }
    for (customCam::CustomCamPipeBuffer *&pb : st.pair_of)   // oldest first (the ring hands buffers out and takes them back in order)
        if (pb) { cf->cam_pipe_stereo.ReleaseBufferAt(1, pb); pb = nullptr; }
    if (st.cbuf_pair) { cf->cam_pipe_stereo.ReleaseBufferAt(1, st.cbuf_pair); st.cbuf_pair = nullptr; }
    if (st.frames == 0) {   // nothing ever went through player 0: open the ring for player 1
        PipeBuffer &b = cf->pipe.RequestBuffer(0);
        b.quit = true;
        cf->pipe.ReleaseBuffer(0);
    }
    PipeBuffer &b1 = cf->pipe.RequestBuffer(1);   // the newest frame (completed or still in flight), or the flag carrier above
    b1.quit = true;
    cf->pipe.ReleaseBuffer(1);
    cf->quit = true;
    {
        std::lock_guard<std::mutex> lk(mut);
        st.running = false;
        st.closed = true;
    }
    cv.notify_all();
}

// One frame of every running member, or false.  block: wait (in 1 ms slices, watching the quit flags) until they are all there;
// otherwise a single pass.  A member whose quit flag is up is marked `leaving` and the call returns at once: the caller finishes
// the step in flight — that member's last submitted frame may be in it, and a frame that was taken from the ring is tracked to
// the end, as in the reference, where CleanUp() joins a thread that is never interrupted inside a frame — and closes the seat.
bool REBVO::BatchGroup::gather(bool block, bool &any_running, bool &any_leaving) {
    while (true) {
        bool all = true, waited = false;   // a blocking pass waits (1 ms at most) on the FIRST member whose frame is missing and only looks at the
                                           // others: a pass never takes longer than that, however many members there are (the quit flags
                                           // are read at the head of every pass — a thousand 1 ms waits in a row made CleanUp() of a
                                           // 1024-member group take a second per member)
        any_running = false;
        any_leaving = false;
        for (Seat &st : seats) {
            if (!st.running) continue;
            REBVO *cf = st.cf;
            if (cf->quit) { st.leaving = true; any_leaving = true; continue; }
            any_running = true;
            const double min_frame_dt = 1.0 / cf->params.soft_fps - 0.5 / cf->params.config_fps;   // rebvo_first_t.cpp:146
            if (!st.cbuf && cf->frame_by_frame && !st.step_granted) {   // frame-by-frame mode (rebvo_first_t.cpp:154-159): no new frame until the application says so
                if (!cf->frame_by_frame_advance) {
                    all = false;
                    if (block && !waited) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); waited = true; }
                    continue;
                }
                cf->frame_by_frame_advance = false;
                st.step_granted = true;   // one "advance" = one frame, however long the frame takes to arrive
                std::cout << "Advancing frame...\n";
            }
            while (!st.cbuf) {
                customCam::CustomCamPipeBuffer *cb = cf->cam_pipe.RequestBufferTimeoutable(1, block && !waited ? 0.001 : 0.0);
                if (!cb) { waited = true; break; }
                st.p_num++;
                if (cb->timestamp - st.t0 < min_frame_dt) { cf->cam_pipe.ReleaseBuffer(1); continue; }   // soft-FPS drop, :172-177
                st.cbuf = cb;
                st.step_granted = false;
                st.t_frame = cb->timestamp;
                if (imu_mode) {   // inter-frame IMU data, waiting for the samples to arrive (rebvo_first_t.cpp:294-304)
                    while (!cf->quit) {
                        st.imu_data = cf->imu->GrabAndIntegrate(st.t0 + cf->params.TimeDesinc, st.t_frame + cf->params.TimeDesinc);
                        if (st.imu_data.n > 0) break;
                        std::this_thread::sleep_for(std::chrono::duration<double>(cf->params.SampleTime));
                    }
                }
                st.ring_idx = -1;
                st.ring_idx = ringEntryOf(reinterpret_cast<const uint8_t *>(cb->img->Data()));
                st.mono = grey_ring && st.ring_idx >= 0 && st.mono_of[st.ring_idx] != 0;
            }
            if (stereo && st.cbuf && !st.cbuf_pair) {   // one pair frame per accepted main frame, no drop logic (rebvo_first_t.cpp:183-199)
                customCam::CustomCamPipeBuffer *pb = cf->cam_pipe_stereo.RequestBufferTimeoutable(1, block && !waited ? 0.001 : 0.0);
                if (!pb) {
                    waited = true;
                } else {
                    st.cbuf_pair = pb;
                    if (std::fabs(st.t_frame - pb->timestamp) > 0.5 / cf->params.config_fps)
                        std::cout << "REBVO Warning: cameras are unsync: " << st.t_frame - pb->timestamp << "\n";
                    st.pair_idx = pairEntryOf(reinterpret_cast<const uint8_t *>(pb->img->Data()));
                    st.pair_mono = pair_grey_ring && st.pair_idx >= 0 && st.pair_mono_of[st.pair_idx] != 0;
                }
            }
            if (!st.cbuf || (stereo && !st.cbuf_pair)) all = false;
        }
        if (any_leaving || !any_running) return false;
        if (all) return true;
        if (!block) return false;
    }
}

// The gathered frames of one step go up, into the slot the next edgehip_process_frame will take (and, with stereo pairs, their pair
// frames into the slot behind the ring).
int REBVO::BatchGroup::upload(std::vector<double> &ts, int &slot) {
    slot = edgehip_next_slot(hip);
    for (int i = 0; i < cap; i++)
        if (seats[i].running) ts[i] = seats[i].t_frame;
    int rc = uploadRuns(slot, false);
    // the pair frames: the copy into the pair slot waits by itself (on the upload stream) for the frame that used the slot last —
    // the step before — so it runs behind that step, not under it; the main frames' copies of the step after queue up behind it
    if (rc == 0 && stereo) rc = uploadRuns(kPairSlot, true);
    return rc;
}

int REBVO::BatchGroup::uploadRuns(int slot, bool pair) {
    int rc = 0;
    uint8_t *const grey_r = pair ? pair_grey_ring : grey_ring;
    auto idx_of = [&](const Seat &st) { return pair ? st.pair_idx : st.ring_idx; };
    // a step whose frames are ALL mono goes up as 8-bit planes (a third of the bytes; the slot's format is one per step)
    bool all_mono = grey_r != nullptr;
    for (const Seat &st : seats) all_mono = all_mono && (!st.running || (pair ? st.pair_mono : st.mono));
    if (all_mono && !pair) mono_steps++;
    // runs of neighbouring members whose frames sit in the same entry of the page-locked ring are contiguous memory: one copy each
    // (in lock-step without drops: one copy for the whole group); a heap image goes through the library's staging buffer
    for (int i = 0; i < cap && rc == 0;) {
        Seat &st = seats[i];
        if (!st.running) { i++; continue; }   // a member that left: its sequence keeps running on whatever the slot holds, nobody reads it
        const int e = idx_of(st);
        if (e < 0) {
            rc = edgehip_upload_rgb(hip, slot, reinterpret_cast<const uint8_t *>((pair ? st.cbuf_pair : st.cbuf)->img->Data()), i, 1);
            i++;
            continue;
        }
        int n = 1;
        while (i + n < cap && seats[i + n].running && idx_of(seats[i + n]) == e) n++;
        rc = all_mono ? edgehip_upload_grey8_pinned(hip, slot, pair ? pairGrey(e, i) : greyImage(e, i), i, n)
                      : edgehip_upload_rgb_pinned(hip, slot, pair ? pairImage(e, i) : ringImage(e, i), i, n);
        i += n;
    }
    return rc;
}

// One edgehip_process_frame for the step whose frames upload() sent, and the members' PipeBuffers of it.
int REBVO::BatchGroup::launch(long step, const std::vector<double> &ts) {
    const double tp0 = detail::now_s();
    int rc = 0;
    if (imu_mode) {   // every member's integrated IMU data of the interval that ends with this step's frame (edgehip_set_imu)
        for (int i = 0; i < cap; i++) {
            const Seat &st = seats[i];
            if (!st.running) continue;
            const IntegratedImuData &d = st.imu_data;
            edgehip_imu_integrated &o = imu_in[i];
            o.n = d.n; o.pad = 0; o.dt = d.dt;
            std::memcpy(o.Rot, d.Rot.a, sizeof o.Rot);
            std::memcpy(o.giro, d.giro.v, sizeof o.giro); std::memcpy(o.acel, d.acel.v, sizeof o.acel); std::memcpy(o.comp, d.comp.v, sizeof o.comp);
            std::memcpy(o.dgiro, d.dgiro.v, sizeof o.dgiro); std::memcpy(o.cacel, d.cacel.v, sizeof o.cacel);
        }
        rc = edgehip_set_imu(hip, imu_in.data());
        if (rc != 0) return rc;
    }
    rc = edgehip_process_frame(hip, ts.data());
    if (rc != 0) return rc;
    const double tp1 = detail::now_s();
    tm.process += tp1 - tp0;
    // the members' PipeBuffers of this step (player 0), and — while the copies run — the image for a callback's PipeBuffer::imgc
    for (int i = 0; i < cap; i++) {
        Seat &st = seats[i];
        if (!st.running) continue;
        REBVO *cf = st.cf;
        PipeBuffer &nb = cf->pipe.RequestBuffer(0);
        nb.t = st.t_frame;
        nb.p_id = st.p_num - 1;
        nb.quit = false;
        nb.dtp0 = 0;
        nb.dtp1 = tp0;   // start of the step; complete() turns it into the step's duration
        if (imu_mode) nb.imu = st.imu_data;
        nb.imgc_valid = false;
        const bool listened = cf->haveCallBack() || cf->saveImg;
        if (st.out_inline && listened) {   // somebody listens now: from here on this member has its output thread
            st.out_inline = false;
            st.out_thread = std::thread(ThirdThread, cf);
        }
        if (listened) {
            cf->ensureHostViews(nb, false);
            nb.imgc_valid = true;   // the output thread converts / saves only a frame that was really kept (a request that arrives after the
                                    // launch is honoured by the first later frame launched with it pending)
            if (st.ring_idx >= 0 && st.side_ok[st.ring_idx]) {   // the application's thread made the copy: take it
                std::swap(nb.imgc, st.side_img[st.ring_idx]);
                st.side_ok[st.ring_idx] = 0;
            } else {
                std::memcpy(nb.imgc->Data(), st.cbuf->img->Data(), (size_t)cf->params.ImageSize.w * cf->params.ImageSize.h * 3);
            }
            if (stereo && nb.imgc_pair && st.cbuf_pair)   // rebvo_first_t.cpp:259 ((*pbuf.imgc_pair) = data_pair), for listeners only
                std::memcpy(nb.imgc_pair->Data(), st.cbuf_pair->img->Data(), frame_bytes);
        }
        cf->pipe.ReleaseBuffer(0);
        st.buf_of[step & 3] = &nb;
        st.t0 = st.t_frame;
        if (cf->system_reset) {   // rebvo_second_t.cpp:609-620: behind this frame, before the next
            rc = edgehip_depth_reset(hip, i);
            cf->system_reset = false;
            if (rc != 0) return rc;
        }
        st.chold = st.cbuf;   // the copy may still be reading it: releaseHeld()
        st.cbuf = nullptr;
        st.pair_of[step & 3] = st.cbuf_pair;
        st.cbuf_pair = nullptr;
        st.frames++;
    }
    newest = step;
    // the callbacks' KeyLine lists of the frame BEFORE this one (the old slot, as this step's tracking leaves it): packed in-stream now
    Export &ex = exp_of[step & 3];
    ex.step = -1;
    if (step >= 1 && cb_depth > 1) {
        ex.seats.clear();
        for (int i = 0; i < cap; i++) {
            Seat &st = seats[i];
            if (!st.running || !st.cf->haveCallBack() || st.frames < 2) continue;   // (frames counts this step's frame already)
            if (!st.kl_pinned) pinKeyLines(st, true);
            ex.seats.push_back(i);
        }
        if (!ex.seats.empty()) {
            rc = edgehip_export_keylines(hip, (int)ex.seats.size(), ex.seats.data(), &ex.ticket);
            if (rc != 0) return rc;
            ex.step = step;
            ex.fetched = false;
        }
    }
    tm.buffers += detail::now_s() - tp1;
    tm.steps++;
    return 0;
}

int REBVO::BatchGroup::exportFetch(Export &ex, const std::vector<int32_t> &kn, const std::vector<edgehip_keyline *> &dst) {
    const int rc = edgehip_export_fetch(hip, ex.ticket, kn.data(), dst.data());
    if (rc == 0) ex.fetched = true;
    return rc;
}

// Every export still outstanding is waited for (or dropped): before KeyLine arrays are unpinned, a copy must not be on its way into them.
void REBVO::BatchGroup::dropExports() {
    for (Export &ex : exp_of)
        if (ex.step >= 0) { (void)edgehip_export_wait(hip, ex.ticket); ex.step = -1; }
}

// Hand the camera buffers of the step launched last back as soon as the copies into its slot have read them (the frames are still
// being processed, and the next step's copies may already run behind).
int REBVO::BatchGroup::releaseHeld(int slot) {
    const double t0 = detail::now_s();
    struct Acc { double &a; double t; ~Acc() { a += detail::now_s() - t; } } acc{tm.held, t0};
    if (ring) {
        const int rc = edgehip_upload_wait(hip, slot);
        if (rc != 0) return rc;
    }
    for (Seat &st : seats) {
        if (!st.chold) continue;
        st.cf->cam_pipe.ReleaseBufferAt(1, st.chold);
        st.chold = nullptr;
    }
    return 0;
}

int REBVO::BatchGroup::complete(long step, int slot, std::vector<edgehip_nav> &navs) {
    const double tr0 = detail::now_s();
    int rc = edgehip_read_nav_log(hip, (int)step, 1, navs.data());   // waits for this frame, not for the ones enqueued behind it
    if (rc == 0 && imu_mode) rc = edgehip_read_nav_imu_log(hip, (int)step, 1, navs_imu.data());
    if (rc == 0 && stereo) rc = edgehip_read_stereo_matches_log(hip, (int)step, 1, stereo_nm.data());
    if (rc != 0) return rc;
    const double now = detail::now_s();
    tm.records += now - tr0;
    struct Acc { double &a; double t; ~Acc() { a += detail::now_s() - t; } } acc{tm.handoff, now};
    // the members' records; who gets the frame before delivered with its KeyLines
    std::vector<int32_t> cb_seq;
    std::vector<edgehip_keyline *> cb_dst;
    std::vector<PipeBuffer *> cb_buf, deliver(cap, nullptr), mine(cap, nullptr);
    int slot_before = -1;
    Export &ex = exp_of[step & 3];
    const bool have_ex = ex.step == step;
    auto in_export = [&](int seat) { return have_ex && std::find(ex.seats.begin(), ex.seats.end(), seat) != ex.seats.end(); };
    for (int i = 0; i < cap; i++) {
        Seat &st = seats[i];
        if (!st.running || !st.buf_of[step & 3]) continue;
        REBVO *cf = st.cf;
        PipeBuffer &nb = *st.buf_of[step & 3];
        st.buf_of[step & 3] = nullptr;
        mine[i] = &nb;
        const edgehip_nav &n = navs[i];
        const bool first = !st.have_prev;   // this member's first frame: "dummy processing" (rebvo_second_t.cpp:108-121)
        nb.dt = n.dt;
        nb.K = 1; nb.Kp = n.Kp; nb.RKp = n.RKp;
        nb.s_rho_p = n.s_rho_q;
        nb.EstimationOK = n.estimation_ok != 0;
        nb.ef->nmatch = n.klm_num;
        nb.ef->reTunedThresh = n.retuned_thresh;
        nb.ef->kn = n.kn;
        if (!first) detail::fill_nav(n, nb.nav);
        else nb.nav = NavData();
        if (imu_mode && !first) {   // the IMU branch's hand-over (rebvo_second_t.cpp:550-606): gravity-aligned pose, metric velocity, filter state
            const edgehip_nav_imu &ni = navs_imu[i];
            detail::fill_nav_imu(ni, nb);
            nb.ef->nmatch = ni.klm_num;
        }
        nb.stereo_match_num = stereo && !first && n.estimation_ok ? stereo_nm[i] : 0;   // rebvo_second_t.cpp:471-477
        if (st.pair_of[step & 3]) {   // this step's pair frame: its copy ran before the step's stage A, the step is done
            cf->cam_pipe_stereo.ReleaseBufferAt(1, st.pair_of[step & 3]);
            st.pair_of[step & 3] = nullptr;
        }
        nb.dtp1 = now - nb.dtp1;
        if (!first) cf->pushNav(nb.nav);
        if (st.have_prev) {   // the frame before goes to the output thread, with its edge map as this frame's tracking left it
            PipeBuffer &ob = cf->pipe.RequestBuffer(1);
            deliver[i] = &ob;
            if (cf->haveCallBack() && cb_depth > 1) {
                // the lists were packed behind this step (launch()); a callback registered after that gets this one delivery without
                // KeyLines — the ring slot may have been detected into again — and the next with them
                if (!in_export(i)) ob.ef->kn = 0;
            } else if (cf->haveCallBack() && newest - step >= 2) {
                ob.ef->kn = 0;
            } else if (cf->haveCallBack()) {
                if (!st.kl_pinned) pinKeyLines(st, true);
                cb_seq.push_back(i);
                cb_dst.push_back(reinterpret_cast<edgehip_keyline *>(ob.ef->kl.data()));
                cb_buf.push_back(&ob);
                slot_before = st.slot_prev;   // (lock-step: the same slot for every member)
            }
        }
        st.have_prev = true;
        st.slot_prev = slot;
    }
    if (!cb_seq.empty()) {   // (REBVO_GROUP_CB_DEPTH=1) AoS KeyLines of every member with a callback: one packing kernel, one copy per list, synchronising
        std::vector<int32_t> kn(cb_seq.size(), 0);
        rc = edgehip_download_keylines_batch(hip, slot_before, (int)cb_seq.size(), cb_seq.data(), cb_dst.data(), kn.data());
        if (rc != 0) std::cout << "\nREBVO: edgehip_download_keylines_batch failed: " << edgehip_last_error() << "\n";
        for (size_t j = 0; j < cb_buf.size(); j++) cb_buf[j]->ef->kn = rc == 0 ? kn[j] : 0;
    }
    if (have_ex) {
        // this step's export: the lists of the frames about to be delivered.  Normally its copies were enqueued one completion ago
        // (below) and have landed under the steps since; a step completed right behind its launch enqueues them now.
        if (!ex.fetched) {
            std::vector<int32_t> kn(ex.seats.size(), 0);
            std::vector<edgehip_keyline *> dst(ex.seats.size(), nullptr);
            for (size_t j = 0; j < ex.seats.size(); j++)
                if (PipeBuffer *ob = deliver[ex.seats[j]]) { kn[j] = ob->ef->kn; dst[j] = reinterpret_cast<edgehip_keyline *>(ob->ef->kl.data()); }
            rc = exportFetch(ex, kn, dst);
        }
        const int rw = edgehip_export_wait(hip, ex.ticket);
        ex.step = -1;
        if (rc == 0) rc = rw;
        if (rc != 0) {
            std::cout << "\nREBVO: KeyLine export failed: " << edgehip_last_error() << "\n";
            for (int seat : ex.seats) if (deliver[seat]) deliver[seat]->ef->kn = 0;
        }
    }
    {   // the NEXT step's export (already packed if that step has been launched): its lists belong to the frames whose records were just
        // read — the lengths are known now, the destinations are these frames' PipeBuffers — so its copies go out at once and run under
        // the steps in flight
        Export &nx = exp_of[(step + 1) & 3];
        if (rc == 0 && nx.step == step + 1 && !nx.fetched) {
            std::vector<int32_t> kn(nx.seats.size(), 0);
            std::vector<edgehip_keyline *> dst(nx.seats.size(), nullptr);
            for (size_t j = 0; j < nx.seats.size(); j++)
                if (PipeBuffer *nb = mine[nx.seats[j]]) { kn[j] = nb->ef->kn; dst[j] = reinterpret_cast<edgehip_keyline *>(nb->ef->kl.data()); }
            rc = exportFetch(nx, kn, dst);
        }
    }
    for (int i = 0; i < cap; i++) {   // (PipeBuffer::img, the grey image a callback may look at, is formed by the member's output thread)
        if (!deliver[i]) continue;
        REBVO *cf = seats[i].cf;
        cf->pipe.ReleaseBuffer(1);
        if (seats[i].out_inline) {    // nobody listens: the frame passes the third player's position right here
            (void)cf->pipe.RequestBuffer(2);
            cf->pipe.ReleaseBuffer(2);
        }
    }
    return rc;
}

void REBVO::BatchGroup::threadMain() {
    static_assert(sizeof(KeyLine) == sizeof(edgehip_keyline), "KeyLine mirrors edgehip_keyline");
    std::vector<double> ts(cap, 0.0);
    std::vector<edgehip_nav> navs(cap);
    // &ProcesorConfig: the group's thread is every member's "first thread"; the first member's CamaraT1 places it (rebvo_first_t.cpp:136-141)
    if (seats[0].cf && seats[0].cf->params.cpuSetAffinity && !detail::set_affinity(seats[0].cf->params.cpu0)) {
        std::cout << "REBVO: Cannot set cpu affinity on the first thread";
        for (Seat &st : seats)
            if (st.running) st.cf->quit = true;
    }
    if (const char *e = getenv("REBVO_GROUP_CB_DEPTH")) cb_depth = std::max(1, std::min(2, atoi(e)));
    int base_depth = imu_mode ? 3 : 2;
    if (const char *e = getenv("REBVO_GROUP_DEPTH")) base_depth = std::max(1, std::min(3, atoi(e)));   // (A/B measurements)
    long step = 0;                       // frames of the context enqueued so far
    struct InFlight { long step; int slot; };
    std::vector<InFlight> pending;       // steps enqueued and not yet completed, oldest first (at most 2)
    int rc = 0;
    auto complete_oldest = [&]() {
        const InFlight f = pending.front();
        pending.erase(pending.begin());
        return complete(f.step, f.slot, navs);
    };
    const bool look_ahead = !getenv("REBVO_GROUP_LOOKAHEAD") || atoi(getenv("REBVO_GROUP_LOOKAHEAD")) != 0;   // (0: A/B measurements)
    bool have_next = false;              // the next step's frames are gathered and on their way up (slot_next)
    int slot_next = -1;
    while (rc == 0) {
        if (!have_next) {
            bool any = false, leaving = false;
            // with steps in flight a single pass decides: either the next frames are all waiting (enqueue them under what runs), or
            // the records of what is in flight are read now
            const double tg0 = detail::now_s();
            bool ready = gather(pending.empty(), any, leaving);
            tm.gather += detail::now_s() - tg0;
            if (!ready) {
                while (rc == 0 && !pending.empty()) rc = complete_oldest();
                if (rc != 0) break;
            }
            if (leaving) {   // (nothing of theirs is in flight any more)
                dropExports();
                for (Seat &st : seats)
                    if (st.running && st.leaving) closeSeat(st);
                continue;
            }
            if (!any) break;
            if (!ready) continue;
            const double tu0 = detail::now_s();
            rc = upload(ts, slot_next);
            tm.upload += detail::now_s() - tu0;
            if (rc != 0) break;
        }
        have_next = false;
        bool callbacks = false;
        for (Seat &st : seats) callbacks |= st.running && st.cf->haveCallBack();
        // steps in flight.  ImuMode > 0: a frame's record exists only behind the scale filter, which runs on a stream of its own for ~0.5 ms per
        // step (one thread per sequence) under the NEXT frames — two steps of slack left the device idle between filters (1.09 ms per step for
        // eight members against 0.58 through the C-ABI); three cover it
        const size_t depth = callbacks && cb_depth < 2 ? 1 : (size_t)base_depth;
        const int slot = slot_next;
        std::array<double, 5> tl{};
        tl[0] = detail::now_s();
        rc = launch(step, ts);
        if (rc != 0) break;
        tl[1] = detail::now_s();
        pending.push_back({step, slot});
        step++;
        {   // look ahead: if every member's next frame is waiting already, its copy goes behind this step's on the upload stream
            // — before this thread turns to the records of the steps in flight, so the link does not idle while the host works
            bool any = false, leaving = false;
            const double ta0 = detail::now_s();
            if (look_ahead && gather(false, any, leaving)) {
                rc = upload(ts, slot_next);
                if (rc != 0) break;
                have_next = true;
                tm.ahead_hits++;
            }
            tm.ahead += detail::now_s() - ta0;
        }
        tl[2] = detail::now_s();
        rc = releaseHeld(slot);
        tl[3] = detail::now_s();
        while (rc == 0 && pending.size() > depth) rc = complete_oldest();
        tl[4] = detail::now_s();
        if (tlog.size() < 64) tlog.push_back(tl);
        if (getenv("REBVO_GROUP_TIMING") && atoi(getenv("REBVO_GROUP_TIMING")) >= 3)
            std::fprintf(stderr, "REBVO(hip) group '%s': step %ld launched %.3f ms, look-ahead %.3f, copy done %.3f, completed %.3f (since launch start)\n", name.c_str(), step - 1,
                         (tl[1] - tl[0]) * 1e3, (tl[2] - tl[0]) * 1e3, (tl[3] - tl[0]) * 1e3, (tl[4] - tl[0]) * 1e3);
    }
    while (rc == 0 && !pending.empty()) rc = complete_oldest();
    if (rc != 0) {
        std::cout << "REBVO(hip): " << edgehip_last_error() << "\n";
        std::lock_guard<std::mutex> lk(mut);
        failed = true;
        error = edgehip_last_error();
    }
    if (rc != 0) (void)edgehip_sync(hip);
    dropExports();
    for (Seat &st : seats)
        if (st.running) closeSeat(st);
    if (getenv("REBVO_GROUP_TIMING") && tm.steps > 0) {
        const double k = 1e6 / tm.steps;
        std::printf("REBVO(hip) group '%s': %ld steps (%ld as 8-bit planes), look-ahead copies %ld; us per step on the group thread: gather %.0f, upload calls %.0f, "
                    "process_frame %.0f, PipeBuffers %.0f, look-ahead %.0f, wait for the copy + release %.0f, wait for the records %.0f, hand-off %.0f\n",
                    name.c_str(), tm.steps, mono_steps, tm.ahead_hits, tm.gather * k, tm.upload * k, tm.process * k, tm.buffers * k, tm.ahead * k, tm.held * k,
                    tm.records * k, tm.handoff * k);
        if (atoi(getenv("REBVO_GROUP_TIMING")) > 1)
            for (size_t j = tlog.size() > 12 ? tlog.size() - 12 : 0; j < tlog.size(); j++)
                std::printf("  step %2zu: launch %8.0f..%8.0f  look-ahead copy issued %8.0f  this step's copy done %8.0f  records of step-2 read %8.0f  (us)\n", j,
                            (tlog[j][0] - tlog[0][0]) * 1e6, (tlog[j][1] - tlog[0][0]) * 1e6, (tlog[j][2] - tlog[0][0]) * 1e6, (tlog[j][3] - tlog[0][0]) * 1e6,
                            (tlog[j][4] - tlog[0][0]) * 1e6);
    }
}

}  // namespace rebvo

// ---- flat C hook for the tests (ctypes): who may sit together -------------------------------------------------------------------
// Builds objects from one GlobalConfig and tries to seat them in one group: same parameters and BatchSize -> accepted; another
// tracker parameter, another BatchSize, or one object too many -> Init() returns false with the reason in lastError(), and the
// group that is left incomplete (or running without frames) shuts down cleanly.  Returns 0 when every step went as it must,
// otherwise the number of the step that did not.
extern "C" int rebvo_group_selftest(const char *config_file) {
    using namespace rebvo;
    REBVO proto(config_file);
    if (!proto.isInitOk()) return 1;
    REBVOParameters p = proto.getParams();
    p.CameraType = 3; p.ImuMode = 0; p.StereoAvaiable = false;
    p.GpuBatchGroup = "selftest";
    p.GpuBatchSize = 2;
    REBVO a(p);
    if (!a.Init()) return 2;                                   // first member: the context exists, the group waits for its partner
    REBVOParameters q = p;
    q.TrackerIterNum += 1;
    REBVO b(q);
    if (b.Init() || b.lastError().find("same") == std::string::npos) return 3;        // another tracker parameter
    q = p;
    q.GpuBatchSize = 3;
    REBVO c(q);
    if (c.Init() || c.lastError().find("same") == std::string::npos) return 4;        // another BatchSize
    q = p;
    q.ImuMode = 1;
    REBVO d(q);
    if (d.Init() || d.lastError().find("BatchGroup") == std::string::npos) return 5;  // a group member must be CameraType 3 / ImuMode 0 / mono
    q = p;
    q.StereoAvaiable = true;                                   // a stereo member does not sit with mono members (nor one with other &Stereo intrinsics
    q.pp_x_stereo = p.pp_x; q.pp_y_stereo = p.pp_y;            // with stereo members: the pair slot's camera is one per context)
    q.z_f_x_stereo = p.z_f_x; q.z_f_y_stereo = p.z_f_y;
    REBVO s(q);
    if (s.Init() || s.lastError().find("same") == std::string::npos) return 12;
    REBVO e(p);
    if (!e.Init()) return 6;                                   // the partner: the group is complete and starts
    REBVO f(p);
    if (f.Init() || f.lastError().find("selftest") == std::string::npos) return 7;   // one too many: nobody joins a running group
    if (!a.Running() || !e.Running()) return 8;
    a.CleanUp();                                               // leaves; the group carries on with e alone
    if (!e.Running()) return 9;
    e.CleanUp();
    REBVO g(p), h(p);                                          // the name is free again once the last member has gone
    if (!g.Init() || !h.Init()) return 10;
    g.CleanUp();
    h.CleanUp();
    p.GpuBatchSize = 0;
    REBVO z(p);
    if (z.Init()) return 11;                                   // a named group needs BatchSize >= 1
    return 0;
}

// ---- flat C hook for the CPU tests: the ring of players, with one player holding two buffers ---------------------------------------
// Player 0 writes 0, 1, 2, ... into a Pipeline<long> of four entries; player 1 takes the next entry BEFORE it lets go of the one
// before (ReleaseBufferAt: what the group thread does with the camera rings while a copy still reads the older frame) and checks
// that the values arrive in order and that an entry it still holds is never written.  0 = as it must be.
extern "C" int rebvo_pipeline_selftest(int count) {
    using namespace rebvo;
    Pipeline<long> pipe(4, 2);
    for (unsigned j = 0; j < pipe.Size(); j++) pipe[j] = -1;
    std::thread producer([&] {
        for (long v = 0; v < count; v++) {
            long &b = pipe.RequestBuffer(0);
            b = v;
            pipe.ReleaseBuffer(0);
        }
    });
    int bad = 0;
    long *held = nullptr;
    long held_value = -1;
    for (long v = 0; v < count && !bad; v++) {
        long *b = pipe.RequestBufferTimeoutable(1, 5.0);
        if (!b) { bad = 1; break; }                                // the producer starved although an entry was free
        if (*b != v) bad = 2;                                      // out of order
        if (held && *held != held_value) bad = 3;                  // written while held
        if (held) pipe.ReleaseBufferAt(1, held);
        held = b;
        held_value = v;
    }
    if (held) pipe.ReleaseBufferAt(1, held);
    if (bad == 1) {   // let the producer run out: hand everything back as it comes
        for (;;) { long *b = pipe.RequestBufferTimeoutable(1, 0.2); if (!b) break; pipe.ReleaseBuffer(1); }
    }
    producer.join();
    return bad;
}
