// datasetcam.cpp — see datasetcam.h.  List parsing and frame hand-off follow src/VideoLib/datasetcam.cpp:32-220.
#include "rebvo/datasetcam.h"

#include <cstdlib>
#include <fstream>
#include <iostream>

namespace rebvo {

namespace {
// blanks, tabs, CR, LF off both ends (Configurator::ShrinkWS + ShrinkNV)
std::string strip(const std::string &s) {
    const char *ws = " \t\r\n\v\f";
    const size_t first = s.find_first_not_of(ws);
    return first == std::string::npos ? std::string() : s.substr(first, s.find_last_not_of(ws) - first + 1);
}
}  // namespace

// The list: "<time stamp>[,| ]<file name>" per line, '#' lines and empty lines skipped (EuRoC data.csv, TUM rgb.txt;
// src/VideoLib/datasetcam.cpp:51-85).  A line without a number in front, or without a name behind it, is a syntax error that
// leaves the camera in its error state, with the reference's messages.
DataSetCam::DataSetCam(const char *DataSetDir, const char *DataSetFile, Size2D frame_size, double time_scale, const char *)
    : frame(frame_size) {
    std::ifstream list(DataSetFile);
    if (!list.is_open()) {
        std::cout << "\nDataSetCamera: Failed to open file " << DataSetFile << "\n";
        return;
    }
    const std::string dir(DataSetDir);
    for (std::string raw; std::getline(list, raw);) {
        const std::string line = strip(raw);
        if (line.empty() || line.front() == '#') continue;
        char *behind = nullptr;
        const double t = std::strtod(line.c_str(), &behind);
        const size_t used = (size_t)(behind - line.c_str());
        if (used == 0 || used == line.size()) {
            std::cout << "\nDataSetCamera: EDataFile sintax error line " << listed.size() << "String:" << line << "\n";
            return;
        }
        listed.push_back({t * time_scale, dir + strip(line.substr(used + (line[used] == ',' ? 1 : 0)))});
    }
    std::cout << "\nLoaded " << listed.size() << " File names\n";
    error = false;
}

// One image of the list into `frame` (PNG / JPEG / PGM / PPM by content: png_reader.cpp, jpeg_reader.cpp); -1 with a message when it
// does not load or has another size than the configured one (datasetcam.cpp:109-171 reports the same two conditions).
int DataSetCam::LoadImage(const std::string &i_name) {
    std::vector<RGB24Pixel> px;
    unsigned w = 0, h = 0;
    std::string why;
    bool mono = false;
    const bool ok = LoadImageRGB24(i_name, px, w, h, why, &mono);
    if (!ok) std::cout << "\nDataSetCam: Image " << i_name << " " << why << "\n";
    else if (w != frame.Size().w || h != frame.Size().h)
        std::cout << "\nDataSetCam: Error the image size (" << w << "," << h << ") doesn't match the configures size (" << frame.Size().w << ","
                  << frame.Size().h << ")\n";
    else {
        frame.copyFrom(px.data());
        grey.clear();
        if (mono) {
            grey.resize(px.size());
            for (size_t i = 0; i < px.size(); i++) grey[i] = px[i].pix.r;
        }
        return 0;
    }
    return -1;
}

bool DataSetCam::ensureLoaded() {
    if (loaded) return true;
    if (error) return false;
    if (cursor >= listed.size()) std::cout << "\nDataSetCamera: End of file list after " << cursor << " Images\n";
    // (an image that does not load: the reference would retry the same file forever; here the camera fails)
    if (cursor >= listed.size() || LoadImage(listed[cursor].path) < 0) {
        error = true;
        return false;
    }
    stamp = listed[cursor++].stamp;
    return loaded = true;
}

int DataSetCam::WaitFrame(bool) { return ensureLoaded() ? 0 : -1; }

// the two ways a frame leaves the camera (datasetcam.cpp:173-220): copied out, or as a pointer to the camera's own buffer
int DataSetCam::GrabFrame(RGB24Pixel *data, double &tstamp, bool drop_frames) {
    double t = 0;
    const RGB24Pixel *src = GrabBuffer(t, drop_frames);
    if (!src) return -1;
    frame.copyTo(data);
    tstamp = t;
    return 0;
}

RGB24Pixel *DataSetCam::GrabBuffer(double &tstamp, bool) {
    if (!ensureLoaded()) return nullptr;
    loaded = false;
    grabbed++;
    tstamp = stamp;
    return frame.Data();
}

}  // namespace rebvo
