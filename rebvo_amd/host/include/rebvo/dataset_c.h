/* dataset_c.h — flat C view of the host library's image-list camera (rebvo/datasetcam.h = the reference's DataSetCam,
 * src/VideoLib/datasetcam.cpp:32-220) for callers that are not C++: bench.py replays a mounted EuRoC / TUM data set through
 * the library's OWN reader (list parser, time stamps, PNG / PGM / PPM decoder), the same code dataset_replay feeds the
 * tracker with. */
#ifndef REBVO_AMD_HOST_DATASET_C_H
#define REBVO_AMD_HOST_DATASET_C_H
#ifdef __cplusplus
extern "C" {
#endif

/* DataSetCam(DataSetDir, DataSetFile, {w, h}, time_scale).  NULL when the list cannot be read (message on stdout, like the
 * reference's camera).  The image size is checked when a frame is grabbed. */
void *rebvo_dataset_open(const char *dataset_dir, const char *dataset_file, int w, int h, double time_scale);
int rebvo_dataset_frames(void *ds);                                   /* entries of the list */
/* The next frame of the list as RGB24 (w*h*3 bytes) with its time stamp; *mono (optional) = the file stored one grey channel.
 * 0 on success, -1 at the end of the list or on a decode / size error. */
int rebvo_dataset_grab(void *ds, unsigned char *rgb24, double *tstamp, int *mono);
void rebvo_dataset_close(void *ds);

#ifdef __cplusplus
}
#endif
#endif
