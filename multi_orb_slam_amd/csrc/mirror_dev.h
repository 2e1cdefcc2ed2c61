// mirror_dev.h -- copy of a finished frame's per-feature results into a caller's mapped pinned buffers (orbf_step's result
// set): the job description and the device loop, shared by k_mirror_frame (frame.hip) and k_project_side (hamming.hip).
#pragma once
#include <stdint.h>
#include "../../include/orb_types.h"

namespace morb {

struct MirrorJob {
    const uint32_t *kps, *desc, *x, *y, *ur, *depth;   // device (keypoints as 7 dwords, descriptors as 8)
    uint32_t *h_kps, *h_desc, *h_x, *h_y, *h_ur, *h_depth;
    const int* n_dev; int n_host;                      // rows: the device-side total when there is one, capped by n_host
};

#ifdef __HIPCC__
// thread t of `stride` threads
__device__ __forceinline__ void mirror_rows(const MirrorJob& J, int t, int stride) {
    const int n = J.n_dev ? min(*J.n_dev, J.n_host) : J.n_host;
    constexpr int kp_dw = (int)(sizeof(orb_keypoint) / 4);
    for (int i = t; i < n * 8; i += stride) J.h_desc[i] = J.desc[i];
    for (int i = t; i < n * kp_dw; i += stride) J.h_kps[i] = J.kps[i];
    for (int i = t; i < n; i += stride) {
        J.h_x[i] = J.x[i]; J.h_y[i] = J.y[i]; J.h_ur[i] = J.ur[i]; J.h_depth[i] = J.depth[i];
    }
}
#endif

}  // namespace morb
