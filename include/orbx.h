/* orbx.h -- C ABI of the MI355X ORB extractor (drop-in for ORB_SLAM2::ORBextractor).
 *
 * Replaces, for N cameras per call:
 *   ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
 *        reference include/ORBextractor.h:51-52, src/ORBextractor.cc:411-471          -> orbx_create / orbx_tables
 *   void ORBextractor::operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors)
 *        reference include/ORBextractor.h:60-62, src/ORBextractor.cc:1044-1107        -> orbx_extract
 *   GetLevels/GetScaleFactor/GetScaleFactors/GetInverseScaleFactors/GetScaleSigmaSquares/
 *   GetInverseScaleSigmaSquares   reference include/ORBextractor.h:64-84              -> orbx_tables
 * Call sites being replaced: Frame::ExtractORB / ExtractORB_cam2, reference src/Frame.cc:397-419.
 *
 * All compute runs in hand-written HIP kernels on gfx950; there is no CPU fallback (ORB_E_NO_DEVICE).
 * An extractor handle is single-caller (like the reference's non-re-entrant class) and owns one HIP stream
 * and all device memory, sized at create time.  Plain pointers and sizes only.
 */
#ifndef ORBX_H
#define ORBX_H
#include "orb_types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbx_params { /* ctor arguments, reference include/ORBextractor.h:51-52 */
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

/* Host-only ctor arithmetic (reference src/ORBextractor.cc:416-470).  Arrays hold nlevels entries
 * (umax16: 16).  Any output pointer may be NULL. */
int orbx_tables(const orbx_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                int32_t* features_per_level, int32_t* umax16);

/* params: n_cams entries (the reference builds one ORBextractor per camera, cam 2 with nFeatures/2:
 * src/Tracking.cc:144-145).  max_width/max_height bound every image later passed in. */
int orbx_create(const orbx_params* params, int n_cams, int max_width, int max_height, int device,
                orbx_extractor** out);
void orbx_destroy(orbx_extractor* ex);

/* == ORBextractor::operator() for n_cams images at once (host buffers in, host buffers out).
 * gray[c]: 8-bit single channel, height[c] rows of stride[c] bytes.  An empty image (NULL / 0 size) yields
 * n_out[c] = 0 and leaves that camera's outputs untouched (reference src/ORBextractor.cc:1047-1048).
 * kps_out[c] / desc_out[c] have room for cap[c] keypoints / cap[c]*32 bytes; a camera can return up to
 * nfeatures + 2*nlevels keypoints (quadtree overshoot, reference src/ORBextractor.cc:729-732). */
int orbx_extract(orbx_extractor* ex, int n_cams, const uint8_t* const* gray, const int* width, const int* height,
                 const int* stride, orb_keypoint* const* kps_out, uint8_t* const* desc_out, const int* cap,
                 int* n_out);

/* -- resident path (what bench.py times: images already in HBM) --------------------------------------- */
/* copy one camera's image into the handle's HBM pyramid buffer (async on the handle's stream) */
int orbx_upload(orbx_extractor* ex, int cam, const uint8_t* gray, int width, int height, int stride);
/* same, from a DEVICE pointer (e.g. a frame grabber's or torch's buffer).  The image is read by the first kernel of the
 * next orbx_run / orbx_run_async (all cameras in one launch): it must stay valid and unchanged until that run has finished. */
int orbx_upload_device(orbx_extractor* ex, int cam, const uint8_t* d_gray, int width, int height, int stride);
/* run the whole extractor on the resident images of cameras [0, n_cams); results stay in HBM */
int orbx_run(orbx_extractor* ex);
/* Split form of orbx_run for callers that keep enqueueing dependent work on orbx_stream(ex): orbx_run_async returns
 * without a host synchronisation (device-quadtree path; otherwise it behaves like orbx_run), the per-camera counts are in
 * HBM (orbx_device_counts: int[n_cams] of the most recently enqueued run).  Up to TWO runs may be in flight: the second
 * one (the next timestep's images, uploaded after the first run was enqueued) executes behind the first on the same
 * stream while the caller consumes the first one's results.  orbx_finish waits for the OLDEST run in flight and makes
 * orbx_count() valid for it.  It returns
 *   1 (not an error) when a pyramid level was outside the device quadtree's limits and the results were recomputed on
 *     the host path right away: work enqueued against the asynchronous counts must be redone;
 *   2 in the same situation while a newer run is in flight: nothing was recomputed, because the resident images already
 *     belong to the newer run -- upload this run's images again and call orbx_run (which abandons the newer run). */
int orbx_run_async(orbx_extractor* ex);
int orbx_finish(orbx_extractor* ex);
const int* orbx_device_counts(const orbx_extractor* ex);
int orbx_pending(const orbx_extractor* ex); /* asynchronous runs in flight (0..2) */
/* number of keypoints camera `cam` produced in the last run */
int orbx_count(const orbx_extractor* ex, int cam);
/* copy the last run's results of one camera to host buffers */
int orbx_download(orbx_extractor* ex, int cam, orb_keypoint* kps, uint8_t* desc, int cap);
/* device pointers of the last run's results (valid until the next orbx_run) */
const orb_keypoint* orbx_device_keypoints(const orbx_extractor* ex, int cam);
const uint8_t* orbx_device_descriptors(const orbx_extractor* ex, int cam);
/* the handle's hipStream_t */
void* orbx_stream(const orbx_extractor* ex);
/* Everything enqueued on `other_stream` (a hipStream_t) so far happens before whatever this handle enqueues next; no host wait.
 * For a caller with device work in flight on another stream that reads orbx_device_descriptors() / orbx_device_keypoints():
 * called before the handle's next run, which overwrites them. */
int orbx_wait_for_stream(orbx_extractor* ex, void* other_stream);
/* redirect camera `cam`'s result buffers to caller-owned HBM (e.g. an RCCL all-gather send buffer);
 * d_kps holds cap keypoints, d_desc cap*32 bytes.  NULL restores the internal buffers. */
int orbx_bind_output(orbx_extractor* ex, int cam, orb_keypoint* d_kps, uint8_t* d_desc, int cap);

/* Additionally mirror every run's keypoints + descriptors, concatenated over the cameras in camera order, into
 * pinned host memory: pass the DEVICE-visible aliases (hipHostGetDevicePointer) of buffers holding cap_total keypoints /
 * cap_total*32 bytes.  The describe kernel writes them itself (no copy on the stream); they are complete once the
 * handle's stream has been synchronised.  NULL disables. */
int orbx_set_host_mirror(orbx_extractor* ex, orb_keypoint* kps_devptr, uint8_t* desc_devptr, int cap_total);

/* pyramid level `level` of camera `cam` as the last run left it, dense w*h bytes (what the C++ class fills the reference's public
 * ORBextractor::mvImagePyramid from when asked to; the parity tests compare every level through it) */
int orbx_debug_level(orbx_extractor* ex, int cam, int level, uint8_t* out, int cap_bytes, int* w, int* h);
/* (further stage inspection for the parity tests and the bench -- orbx_debug_*: include/orb_debug.h) */
int orbx_set_profiling(orbx_extractor* ex, int on);
int orbx_stage_times_us(const orbx_extractor* ex, float* out6);

#ifdef __cplusplus
}
#endif
#endif
