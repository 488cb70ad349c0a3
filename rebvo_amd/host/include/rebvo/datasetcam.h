// datasetcam.h — image-list camera for EuRoC / TUM style datasets: same constructor, list-file format and
// GrabBuffer/ReleaseBuffer contract as the reference's DataSetCam (include/VideoLib/datasetcam.h:36-74,
// src/VideoLib/datasetcam.cpp:32-220), without libgd: PNG (what both datasets ship) is decoded with zlib by
// png_reader.cpp, baseline JPEG by jpeg_reader.cpp (libjpeg's integer IDCT / fancy upsampling / colour tables restated: the
// pixels libgd hands the reference); binary PGM/PPM are accepted too.  Baseline and progressive JPEG both decode (jpeg_reader.cpp); arithmetic-coded / 12-bit / CMYK files do not (tools/jpeg_to_png.py).
#ifndef REBVO_AMD_HOST_DATASETCAM_H
#define REBVO_AMD_HOST_DATASETCAM_H

#include <string>
#include <vector>

#include "rebvo/rebvo.h"

namespace rebvo {

// Decode a PNG / baseline JPEG / PGM / PPM file into RGB24 (grey replicated, alpha dropped, 16-bit samples reduced to their high
// byte — what libgd's truecolor conversion yields).  Returns false with a message in `err`.
// `mono` (optional): the file stores one grey channel (PNG colour types 0 / 4, PGM), i.e. r = g = b for every pixel.
bool LoadImageRGB24(const std::string &file, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono = nullptr);

class DataSetCam {
    struct ListedFrame { double stamp; std::string path; };   // one line of the list: time stamp (scaled), DataSetDir + file name
    std::vector<ListedFrame> listed;
    size_t cursor = 0;             // next entry of the list to load
    bool error = true;             // the camera is unusable: list unreadable, end of list reached, an image that does not load
    bool loaded = false;           // `frame` holds an image nobody has grabbed yet
    Image<RGB24Pixel> frame;
    std::vector<uint8_t> grey;     // the frame as 8-bit mono when the file stores one grey channel (EuRoC), else empty
    double stamp = 0;
    unsigned grabbed = 0;          // frames handed out (PakNum)
    bool ensureLoaded();           // a frame is waiting, or the next one of the list has been loaded; false: camera error

public:
    // List file: one "<timestamp>[,| ]<file name>" per line, '#' comments (EuRoC data.csv, TUM rgb.txt);
    // timestamps are multiplied by time_scale (1e-9 for EuRoC nanoseconds), names are prefixed with DataSetDir.
    DataSetCam(const char *DataSetDir, const char *DataSetFile, Size2D frame_size, double time_scale, const char *log_name = nullptr);
    int WaitFrame(bool drop_frames = true);
    int LoadImage(const std::string &i_name);
    int GrabFrame(RGB24Pixel *data, double &tstamp, bool drop_frames = true);
    RGB24Pixel *GrabBuffer(double &tstamp, bool drop_frames = true);
    int ReleaseBuffer() { return 0; }
    // The frame GrabBuffer just returned as 8-bit mono, 1 byte per pixel, or nullptr when the file was a colour image.  The
    // reference expands mono images to RGB24 because its CPU path wants RGB24 (datasetcam.cpp:152-160); the device path takes
    // the 8-bit plane (edgehip_upload_grey8: a third of the bytes, identical results).
    const uint8_t *GreyBuffer() const { return grey.empty() ? nullptr : grey.data(); }
    const bool &Error() { return error; }
    unsigned PakNum() const { return grabbed; }
    size_t NumFrames() const { return listed.size(); }
};

}  // namespace rebvo
#endif
