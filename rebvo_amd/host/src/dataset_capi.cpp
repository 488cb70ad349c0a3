// dataset_capi.cpp — the flat C view declared in rebvo/dataset_c.h.
#include "rebvo/dataset_c.h"

#include "rebvo/datasetcam.h"

using namespace rebvo;

extern "C" {

void *rebvo_dataset_open(const char *dataset_dir, const char *dataset_file, int w, int h, double time_scale) {
    if (!dataset_dir || !dataset_file || w < 1 || h < 1) return nullptr;
    DataSetCam *cam = new DataSetCam(dataset_dir, dataset_file, Size2D{(unsigned)w, (unsigned)h}, time_scale);
    if (cam->Error()) {
        delete cam;
        return nullptr;
    }
    return cam;
}
int rebvo_dataset_frames(void *ds) { return ds ? (int)((DataSetCam *)ds)->NumFrames() : 0; }
int rebvo_dataset_grab(void *ds, unsigned char *rgb24, double *tstamp, int *mono) {
    if (!ds || !rgb24 || !tstamp) return -1;
    DataSetCam *cam = (DataSetCam *)ds;
    if (cam->GrabFrame(reinterpret_cast<RGB24Pixel *>(rgb24), *tstamp) < 0) return -1;
    if (mono) *mono = cam->GreyBuffer() != nullptr;
    return 0;
}
void rebvo_dataset_close(void *ds) { delete (DataSetCam *)ds; }

}  // extern "C"
