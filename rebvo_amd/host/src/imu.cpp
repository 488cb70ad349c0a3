// imu.cpp — ImuGrabber, BiasCorrect and the ScaleEstimator filters (see rebvo/imu.h for the reference lines).
// Host-only scalar code: nothing here touches the GPU.  Expression order follows the reference's TooN expressions
// (left-to-right products with temporaries) so that results agree to rounding.
#include "rebvo/imu.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

namespace rebvo {

using la::Mat;
using la::Vec;

// ------------------------------------------------------------------------------------------------------------
// ImuGrabber
// ------------------------------------------------------------------------------------------------------------
// imugrabber.cpp:34-40: both indexers start at 0 over list_size+1 slots, then --read_inx
ImuGrabber::ImuGrabber(int list_size, double tsamp)
    : imu(list_size + 1), write_inx(0), read_inx(0), size(list_size + 1), tsample(tsamp) {
    read_inx = prev(read_inx);
}

// imugrabber.cpp:46-66
ImuGrabber::ImuGrabber(const std::vector<ImuData> &data_set_data)
    : imu(data_set_data.size() + 1), write_inx(0), read_inx(0), size((int)data_set_data.size() + 1), tsample(0) {
    read_inx = prev(read_inx);
    for (int i = 0; i < size - 1; i++) {
        imu[i] = data_set_data[i];
        write_inx = next(write_inx);
    }
    if (data_set_data.size() > 1) {   // sample time = mean time-stamp spacing
        tsample = (data_set_data[data_set_data.size() - 1].tstamp - data_set_data[0].tstamp) / (data_set_data.size() - 1);
        std::cout << "\nImuGraber: tsample:" << tsample << "\n";
    } else {
        tsample = 0;
        std::cout << "\nImuGraber: coud not set tsample\n";
    }
}

namespace {

// Blanks and tabs off both ends of a line (what Configurator::ShrinkWS does to it, configurator.cpp:33-48).
std::string trimmed(const std::string &s) {
    const size_t first = s.find_first_not_of(" \t");
    return first == std::string::npos ? std::string() : s.substr(first, s.find_last_not_of(" \t") - first + 1);
}

// The leading comma-separated numbers of a text line into out[0..n): as many as are there.  This is what the reference's
// sscanf("%lf,%lf,...") leaves in its arguments (imugrabber.cpp:108-116) — a field that is missing or unreadable, and every field
// behind it, keeps the value it had; blanks in front of a number are skipped, blanks in front of a comma end the record.
int leading_numbers(const std::string &line, double *out, int n) {
    const char *at = line.c_str();
    int got = 0;
    for (; got < n; got++) {
        char *behind = nullptr;
        const double v = std::strtod(at, &behind);
        if (behind == at) break;
        out[got] = v;
        at = behind;
        if (*at != ',') { got++; break; }
        at++;
    }
    return got;
}

}  // namespace

// The IMU data set of ImuMode 2: one sample per line, "t,gx,gy,gz,ax,ay,az[,cx,cy,cz]", '#' lines and empty lines skipped, the time
// stamp scaled to seconds (imugrabber.cpp:80-131; EuRoC's imu0/data.csv with TimeScale 1e-9).  Same messages as the reference.
std::vector<ImuData> ImuGrabber::LoadDataSet(const char *data_file, bool comp_data, double time_scale, bool &error) {
    std::vector<ImuData> samples;
    std::ifstream in(data_file);
    error = !in.is_open();
    if (error) {
        std::cout << "\nImuGrabber: Failed to open file " << data_file << "\n";
        return samples;
    }
    for (std::string raw; std::getline(in, raw);) {
        const std::string line = trimmed(raw);
        if (line.empty() || line.front() == '#') continue;
        ImuData d;
        double f[10] = {d.tstamp, d.giro[0], d.giro[1], d.giro[2], d.acel[0], d.acel[1], d.acel[2], d.comp[0], d.comp[1], d.comp[2]};
        leading_numbers(line, f, comp_data ? 10 : 7);
        d.tstamp = f[0] * time_scale;
        for (int k = 0; k < 3; k++) { d.giro[k] = f[1 + k]; d.acel[k] = f[4 + k]; d.comp[k] = f[7 + k]; }
        samples.push_back(d);
    }
    std::cout << "\nImugrabber: Loaded " << samples.size() << " datums\n";
    return samples;
}

// Cam-IMU transformation from a text file of twelve comma-separated numbers, row by row [R | t] (imugrabber.cpp:133-158: every
// number is what std::stod makes of the text up to the next comma).  false, with a message, on a missing or malformed file.
bool ImuGrabber::LoadCamImuSE3(const char *se3_file) {
    std::ifstream in(se3_file);
    if (!in.is_open()) {
        std::cout << "LoadCamImuSE3: could not open SE3 file" << se3_file << " \n";
        return false;
    }
    double rt[12];
    for (double &v : rt) {
        std::string field;
        std::getline(in, field, ',');
        char *behind = nullptr;
        v = std::strtod(field.c_str(), &behind);
        if (behind == field.c_str()) {   // (the reference lets std::stod throw out of the constructor chain)
            std::cout << "LoadCamImuSE3: malformed SE3 file " << se3_file << "\n";
            return false;
        }
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) RDataSetCam2IMU(r, c) = rt[4 * r + c];
        TDataSetCam2IMU[r] = rt[4 * r + 3];
    }
    return true;
}

bool ImuGrabber::LoadCamImuSE3(const Mat<3, 3> &RCam2IMU, const Vec<3> &TCam2IMU) {
    RDataSetCam2IMU = RCam2IMU;
    TDataSetCam2IMU = TCam2IMU;
    return true;
}

// imugrabber.cpp:175-212: [begin, end) of the samples with tstart < t <= tend (end is exclusive); an empty range
// when the buffer does not yet reach past tend
std::pair<int, int> ImuGrabber::SeachByTimeStamp(double tstart, double tend) {
    std::lock_guard<std::mutex> locker(rw_mut);
    int inx = read_inx;
    do {
        inx = next(inx);
        if (inx == write_inx) return std::pair<int, int>(inx, inx);
    } while (imu[inx].tstamp <= tstart);
    const int begin = inx;
    do {
        inx = next(inx);
        if (inx == write_inx) return std::pair<int, int>(begin, begin);
    } while (imu[inx].tstamp < tend);
    if (imu[inx].tstamp - tend < 1e-12) inx = next(inx);   // a sample exactly at tend is included
    return std::pair<int, int>(begin, inx);
}

// imugrabber.cpp:219-252
IntegratedImuData ImuGrabber::GrabAndIntegrate(double tstart, double tend) {
    const std::pair<int, int> range = SeachByTimeStamp(tstart, tend);
    const Mat<3, 3> Rt = la::transpose(RDataSetCam2IMU);
    IntegratedImuData d;
    for (int inx = range.first; inx != range.second; inx = next(inx)) {
        d.giro = d.giro + Rt * imu[inx].giro;
        d.acel = d.acel + Rt * imu[inx].acel;
        d.comp = d.comp + Rt * imu[inx].comp;
        d.Rot = d.Rot * la::so3_exp((Rt * imu[inx].giro) * tsample);   // integrate the rotation on SO(3)
        d.n++;
    }
    d.dt = d.n * tsample;
    if (d.n > 1) {
        d.giro = d.giro / (double)d.n;
        d.acel = d.acel / (double)d.n;
        d.comp = d.comp / (double)d.n;
        d.dgiro = Rt * (imu[prev(range.second)].giro - imu[range.first].giro);   // angular acceleration, finite difference
        d.dgiro = d.dgiro / d.dt;
    }
    d.cacel = d.acel + la::cross(d.dgiro, -(Rt * TDataSetCam2IMU));   // tangential acceleration of the lever arm
    {
        std::lock_guard<std::mutex> locker(rw_mut);
        read_inx = prev(range.second);   // release the samples read
    }
    return d;
}

// imugrabber.cpp:258-269
bool ImuGrabber::PushData(const ImuData &data) {
    std::lock_guard<std::mutex> locker(rw_mut);
    if (write_inx == read_inx) throw std::overflow_error("ImuGrabber circular buffer full");
    imu[write_inx] = data;
    write_inx = next(write_inx);
    return true;
}

}  // namespace rebvo
