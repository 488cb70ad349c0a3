// imu.h — host side of the IMU branch: the inter-frame IMU integrator and the
// scalar filters SecondThread runs between the GPU stages when ImuMode > 0.
//
//   ImuGrabber                     include/UtilLib/imugrabber.h, src/UtilLib/imugrabber.cpp:34-266
//   edge_tracker::BiasCorrect      src/mtracklib/edge_tracker.cpp:1308-1343
//   ScaleEstimator::EstAcelLsq4    src/mtracklib/scaleestimator.cpp:38-92
//   ScaleEstimator::MeanAcel4      src/mtracklib/scaleestimator.cpp:94-109
//   ScaleEstimator::estKaGMEKBias  src/mtracklib/scaleestimator.cpp:117-318 (+ Minimizer<>::GaussNewton,
//                                  include/UtilLib/minimizer.h:84-114)
//
// Same names and argument meaning as the reference.  Two deliberate differences:
//  * the reference keeps the sample history of EstAcelLsq4 / MeanAcel4 in function-local statics (one history per
//    process); here it is per ScaleEstimator object, so two REBVO instances do not share it;
//  * EstAcelLsq4 adds `V[3]` of a 3-vector when it forms the mean (scaleestimator.cpp:74), one element past the end
//    of a static.  The mean cancels out of the least-squares slope, so the value only matters through rounding; here
//    it is 0, which is what the reference reads when built for the oracle (tests/test_imu_cpu.py pins it).
#ifndef REBVO_AMD_HOST_IMU_H
#define REBVO_AMD_HOST_IMU_H

#include <mutex>
#include <stdexcept>
#include <utility>
#include <vector>

#include "rebvo/linalg.h"
#include "rebvo/imu_filters.h"   // BiasCorrect, ScaleEstimator (host + device)

namespace rebvo {

// Inter-frame IMU sample (include/UtilLib/imugrabber.h:38-53)
struct ImuData {
    double tstamp = 0;
    la::Vec<3> giro = la::Vec<3>::zeros();   // gyroscope, 3 axis
    la::Vec<3> acel = la::Vec<3>::zeros();   // accelerometer
    la::Vec<3> comp = la::Vec<3>::zeros();   // compass
    ImuData() {}
    ImuData(double t, const la::Vec<3> &g, const la::Vec<3> &a, const la::Vec<3> &c = la::Vec<3>::zeros())
        : tstamp(t), giro(g), acel(a), comp(c) {}
};

// What GrabAndIntegrate hands to the tracker (imugrabber.h:57-69)
struct IntegratedImuData {
    int n = 0;
    double dt = 0;
    la::Mat<3, 3> Rot = la::Mat<3, 3>::identity();   // inter-frame rotation
    la::Vec<3> giro = la::Vec<3>::zeros();           // mean gyro
    la::Vec<3> acel = la::Vec<3>::zeros();           // mean accelerometer
    la::Vec<3> comp = la::Vec<3>::zeros();
    la::Vec<3> dgiro = la::Vec<3>::zeros();          // mean angular acceleration
    la::Vec<3> cacel = la::Vec<3>::zeros();          // accelerometer compensated for the lever arm
};

class ImuGrabber {
    std::vector<ImuData> imu;   // circular buffer, one slot more than the capacity
    int write_inx, read_inx;    // util::CircListIndexer values (index of the first free / last read slot)
    const int size;
    double tsample;
    std::mutex rw_mut;
    int next(int i) const { return (i + 1) % size; }
    int prev(int i) const { return (i - 1 + size) % size; }

public:
    la::Mat<3, 3> RDataSetCam2IMU = la::Mat<3, 3>::identity();   // Pimu = RCam2Imu * Pcam + TCam2Imu
    la::Vec<3> TDataSetCam2IMU = la::Vec<3>::zeros();

    ImuGrabber(int list_size, double tsamp);                     // empty ring for pushIMU (ImuMode 1)
    explicit ImuGrabber(const std::vector<ImuData> &data_set_data);   // whole data set (ImuMode 2)

    // csv: tstamp,giro_x,giro_y,giro_z,acel_x,acel_y,acel_z[,comp_x,comp_y,comp_z]; '#' comments
    static std::vector<ImuData> LoadDataSet(const char *data_file, bool comp_data, double time_scale, bool &error);
    bool LoadCamImuSE3(const char *se3_file);   // 3 rows of "r0,r1,r2,t," (comma after every number)
    bool LoadCamImuSE3(const la::Mat<3, 3> &RCam2IMU, const la::Vec<3> &TCam2IMU);
    bool PushData(const ImuData &data);          // throws std::overflow_error when the ring is full
    std::pair<int, int> SeachByTimeStamp(double tstart, double tend);
    IntegratedImuData GrabAndIntegrate(double tstart, double tend);
    double SampleTime() const { return tsample; }
};

}  // namespace rebvo
#endif
