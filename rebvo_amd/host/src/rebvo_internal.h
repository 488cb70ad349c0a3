// rebvo_internal.h — helpers shared by the translation units of librebvohost (not installed)
#ifndef REBVO_AMD_HOST_INTERNAL_H
#define REBVO_AMD_HOST_INTERNAL_H
#include "edgehip.h"
#include "rebvo/rebvo.h"

namespace rebvo {
namespace detail {
double now_s();
// REBVOParameters -> the fields of edgehip_params that reach the hot path (include/rebvo/rebvo.h:64-235)
void fill_hip_params(const REBVOParameters &p, edgehip_params &h);
// edgehip_nav -> NavData, ImuMode 0 (rebvo_second_t.cpp:550-606)
void fill_nav(const edgehip_nav &n, NavData &nav);
// edgehip_nav_imu -> the IMU branch's hand-over into a PipeBuffer (rebvo_second_t.cpp:550-606): NavData (gravity-aligned pose, metric
// velocity, gyro rotation, g, scale), K / Kp / RKp / EstimationOK / match count, and the IMUState members the device keeps per frame
void fill_nav_imu(const edgehip_nav_imu &n, PipeBuffer &pb);
// REBVO::setAffinity (include/rebvo/rebvo.h:424-430): the calling thread onto one CPU; false when the kernel refuses (or cpu is no CPU)
bool set_affinity(int cpu);
}  // namespace detail
}  // namespace rebvo
#endif
