// imu.cpp — ImuGrabber, BiasCorrect and the ScaleEstimator filters (see rebvo/imu.h for the reference lines).
// Host-only scalar code: nothing here touches the GPU.  Expression order follows the reference's TooN expressions
// (left-to-right products with temporaries) so that results agree to rounding.
#include "rebvo/imu.h"

#include <cstdio>
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

static std::string shrink_ws(const std::string &s) {   // Configurator::ShrinkWS (configurator.cpp:33-48)
    size_t p1 = 0;
    while (p1 < s.size() && (s[p1] == ' ' || s[p1] == 0x09)) p1++;
    if (p1 == s.size()) return std::string();
    size_t p2 = s.size() - 1;
    while (p2 > p1 && (s[p2] == ' ' || s[p2] == 0x09)) p2--;
    return s.substr(p1, p2 - p1 + 1);
}

// imugrabber.cpp:80-131
std::vector<ImuData> ImuGrabber::LoadDataSet(const char *data_file, bool comp_data, double time_scale, bool &error) {
    std::ifstream ifile(data_file);
    error = false;
    if (!ifile.is_open()) {
        std::cout << "\nImuGrabber: Failed to open file " << data_file << "\n";
        error = true;
        return std::vector<ImuData>();
    }
    std::vector<ImuData> vector_data;
    int lines = 0;
    while (!ifile.eof()) {
        std::string line;
        std::getline(ifile, line);
        line = shrink_ws(line);
        if (line.size() == 0) continue;
        if (line[0] == '#') continue;
        ImuData d;
        if (comp_data)
            std::sscanf(line.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", &d.tstamp, &d.giro[0], &d.giro[1], &d.giro[2], &d.acel[0],
                        &d.acel[1], &d.acel[2], &d.comp[0], &d.comp[1], &d.comp[2]);
        else
            std::sscanf(line.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf", &d.tstamp, &d.giro[0], &d.giro[1], &d.giro[2], &d.acel[0],
                        &d.acel[1], &d.acel[2]);
        d.tstamp *= time_scale;
        lines++;
        vector_data.push_back(d);
    }
    std::cout << "\nImugrabber: Loaded " << lines << " datums\n";
    return vector_data;
}

// imugrabber.cpp:133-158
bool ImuGrabber::LoadCamImuSE3(const char *se3_file) {
    std::ifstream file(se3_file);
    if (!file.is_open()) {
        std::cout << "LoadCamImuSE3: could not open SE3 file" << se3_file << " \n";
        return false;
    }
    std::string num;
    try {
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                std::getline(file, num, ',');
                RDataSetCam2IMU(i, j) = std::stod(num);
            }
            std::getline(file, num, ',');
            TDataSetCam2IMU[i] = std::stod(num);
        }
    } catch (const std::exception &) {   // the reference lets std::stod throw out of the constructor chain
        std::cout << "LoadCamImuSE3: malformed SE3 file " << se3_file << "\n";
        return false;
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
