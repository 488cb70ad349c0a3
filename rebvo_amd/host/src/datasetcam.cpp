// datasetcam.cpp — see datasetcam.h.  List parsing and frame hand-off follow src/VideoLib/datasetcam.cpp:32-220.
#include "rebvo/datasetcam.h"

#include <fstream>
#include <iostream>

namespace rebvo {

static std::string shrink(const std::string &s) {   // Configurator::ShrinkWS + ShrinkNV: blanks, CR, LF at both ends
    size_t a = 0, b = s.size();
    while (a < b && (isspace((unsigned char)s[a]))) a++;
    while (b > a && (isspace((unsigned char)s[b - 1]))) b--;
    return s.substr(a, b - a);
}

DataSetCam::DataSetCam(const char *DataSetDir, const char *DataSetFile, Size2D frame_size, double time_scale, const char *)
    : buffer(frame_size), strDir(DataSetDir) {
    std::ifstream ifile(DataSetFile);
    if (!ifile.is_open()) {
        std::cout << "\nDataSetCamera: Failed to open file " << DataSetFile << "\n";
        error = true;
        return;
    }
    int linea = 0;
    std::string line;
    while (std::getline(ifile, line)) {
        line = shrink(line);
        if (line.empty() || line[0] == '#') continue;
        size_t pos = 0;
        double t = 0;
        try { t = std::stod(line, &pos); } catch (...) { pos = 0; }
        if (pos == 0 || pos == line.size()) {
            std::cout << "\nDataSetCamera: EDataFile sintax error line " << linea << "String:" << line << "\n";
            error = true;
            return;
        }
        img_time.push_back(t * time_scale);
        if (line.at(pos) == ',') pos++;
        img_list.push_back(strDir + shrink(line.substr(pos)));
        linea++;
    }
    std::cout << "\nLoaded " << linea << " File names\n";
    error = false;
    paknum = 0;
}

int DataSetCam::LoadImage(const std::string &i_name) {
    std::vector<RGB24Pixel> px;
    unsigned w = 0, h = 0;
    std::string err;
    bool mono = false;
    if (!LoadImageRGB24(i_name, px, w, h, err, &mono)) {
        std::cout << "\nDataSetCam: Image " << i_name << " " << err << "\n";
        return -1;
    }
    if (w != buffer.Size().w || h != buffer.Size().h) {
        std::cout << "\nDataSetCam: Error the image size (" << w << "," << h << ") doesn't match the configures size ("
                  << buffer.Size().w << "," << buffer.Size().h << ")\n";
        return -1;
    }
    buffer.copyFrom(px.data());
    grey.clear();
    if (mono) {
        grey.resize(px.size());
        for (size_t i = 0; i < px.size(); i++) grey[i] = px[i].pix.r;
    }
    return 0;
}

int DataSetCam::WaitFrame(bool) {
    if (error) return -1;
    if (img_inx >= NumFrames()) {
        std::cout << "\nDataSetCamera: End of file list after " << img_inx << " Images\n";
        error = true;
        return -1;
    }
    if (LoadImage(img_list[img_inx]) < 0) {   // the reference would retry the same file forever; fail the camera instead
        error = true;
        return -1;
    }
    time = img_time[img_inx];
    img_inx++;
    frm_pending = true;
    return 0;
}

int DataSetCam::GrabFrame(RGB24Pixel *data, double &tstamp, bool drop_frames) {
    if (!frm_pending)
        if (WaitFrame(drop_frames) < 0) return -1;
    buffer.copyTo(data);
    tstamp = time;
    frm_pending = false;
    paknum++;
    return 0;
}

RGB24Pixel *DataSetCam::GrabBuffer(double &tstamp, bool drop_frames) {
    if (!frm_pending)
        if (WaitFrame(drop_frames) < 0) return nullptr;
    frm_pending = false;
    tstamp = time;
    paknum++;
    return buffer.Data();
}

}  // namespace rebvo
