/* orb_types.h -- POD types shared by the extractor (orbx.h) and matcher (orbm.h) C ABIs.
 *
 * orb_keypoint is byte-for-byte cv::KeyPoint (28 bytes: pt.x, pt.y, size, angle, response, octave, class_id),
 * the type ORBextractor::operator() fills (reference include/ORBextractor.h:60-62) and Frame/KeyFrame copy by
 * value (reference src/Frame.cc:221-239).  Descriptors are N x 32 bytes row-major, bit k of byte i = rBRIEF
 * test 8*i+k (reference src/ORBextractor.cc:123-144) -- the CV_8UC1 N x 32 cv::Mat the reference allocates at
 * src/ORBextractor.cc:1069.
 */
#ifndef ORB_TYPES_H
#define ORB_TYPES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orb_keypoint {
    float x, y;       /* level-0 pixel coordinates (pt *= scale, src/ORBextractor.cc:1096-1103) */
    float size;       /* (int)(31 * scale[octave])          (src/ORBextractor.cc:838,847)       */
    float angle;      /* degrees in [0,360), IC_Angle       (src/ORBextractor.cc:77-104)        */
    float response;   /* FAST-9/16 corner score                                                 */
    int32_t octave;   /* pyramid level                                                          */
    int32_t class_id; /* always -1                                                              */
} orb_keypoint;

/* Pinhole + radial-tangential calibration as the reference holds it (CV_32F mK and mDistCoef, src/Tracking.cc;
 * OtherFiles/multi.yaml:7-16).  k1 == 0 switches undistortion off altogether, as Frame::UndistortKeyPoints does
 * (src/Frame.cc:676-680). */
typedef struct orb_calibration {
    float fx, fy, cx, cy;
    float k1, k2, p1, p2, k3;
} orb_calibration;

/* status codes returned by every entry point (never throws across the ABI) */
enum {
    ORB_OK = 0,
    ORB_E_ARG = -1,      /* bad argument (null pointer, size out of range)          */
    ORB_E_HIP = -2,      /* a HIP runtime call failed; see orb_last_error()        */
    ORB_E_CAPACITY = -3, /* caller-provided output capacity too small              */
    ORB_E_NO_DEVICE = -4,/* no usable gfx950 device: the product has no CPU path   */
    ORB_E_TIMEOUT = -5   /* a multi-GPU exchange waited for another rank longer than MORB_EXCHANGE_TIMEOUT_MS */
};

/* thread-local text of the last error on the calling thread ("" if none) */
const char* orb_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
