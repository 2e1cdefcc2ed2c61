// matcher.hip -- the ORB matcher's handle, error plumbing and host-only helpers (include/orbm.h).  The kernels live next
// door: hamming.hip (all-pairs Hamming: top-2, distance matrix, cross-camera top-2), frame.hip (frame assembly and grid),
// search.hip (projection search + first-come resolve); frontend.hip composes them with the extractor into orbf_step
// (include/orbf.h), exchange.hip carries the multi-GPU descriptor exchange.  Shared internal types: matcher_internal.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/orbm.h"
#include "../../include/orb_debug.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "matcher_internal.h"

namespace morb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): this library has no CPU path", hipGetErrorString(e));
        return ORB_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d devices)", device, n); return ORB_E_ARG; }
    hipDeviceProp_t prop;
    MORB_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return ORB_E_NO_DEVICE;
    }
    MORB_HIP(hipSetDevice(device));
    return ORB_OK;
}

}  // namespace morb

extern "C" const char* orb_last_error(void) { return morb::g_err; }


// The single-workgroup kernels (k_resolve, k_frame_build_small) use the opt-in dynamic LDS limit.  The attribute belongs to
// the (kernel, device) pair, so it is raised once per DEVICE, on that device, when the first handle is created there
// (std::call_once: handles are created from several threads).
static int raise_lds_limits(int device) {
    constexpr int MAX_DEV = 64;
    static std::once_flag once[MAX_DEV];
    static int result[MAX_DEV];
    if (device < 0 || device >= MAX_DEV) { morb::set_error("device %d out of range", device); return ORB_E_ARG; }
    std::call_once(once[device], [device] {
        int rc = morb::search_raise_lds_limits();
        if (!rc) rc = morb::frame_raise_lds_limit();
        result[device] = rc;
    });
    return result[device];
}

hipStream_t morb::side_stream(orbm_matcher* m) {
    if (m->side_inline) return m->stream;   // (a front end with a multi-GPU exchange: see orbm_matcher::side_inline)
    if (!m->side_stream) {
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&m->side_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
            (void)hipGetLastError();
            m->side_stream = nullptr;
            morb::set_error("side stream could not be created");
        }
    }
    return m->side_stream;
}

int orbm_create(int device, orbm_matcher** out) {
    MORB_ARG(out != nullptr);
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    if ((rc = raise_lds_limits(device))) return rc;
    orbm_matcher* m = new orbm_matcher();
    m->device = device;
    // Matching is the latency chain a caller waits for while extraction of later timesteps fills the rest of the chip:
    // its streams get the highest priority the device offers.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&m->own_stream, hipStreamNonBlocking, prio_greatest);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete m; return ORB_E_HIP; }
    m->stream = m->own_stream;
    if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_q, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_stage_f, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) {
        morb::set_error("events could not be created");
        orbm_destroy(m);
        return ORB_E_HIP;
    }
    const char* hr = getenv("MORB_HOST_RESOLVE");
    m->host_resolve = hr && atoi(hr) != 0;
    *out = m;
    return ORB_OK;
}

void orbm_destroy(orbm_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    if (m->side_stream) { (void)hipStreamSynchronize(m->side_stream); (void)hipStreamDestroy(m->side_stream); }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->ev_q) (void)hipEventDestroy(m->ev_q);
    m->h_c0.release(); m->h_c1.release(); m->h_c2.release(); m->d_cscratch.release(); m->h_gcnt.release(); m->d_gstart.release();
    m->d_q.release(); m->d_r.release(); m->d_scratch.release(); m->d_queries.release(); m->d_occ.release();
    m->d_i0.release(); m->d_i1.release(); m->d_i2.release(); m->d_choice.release(); m->d_claim.release(); m->d_qmeta.release(); m->d_win2.release();
    m->d_match.release(); m->d_status.release(); m->d_gclaim.release(); m->d_rsync.release(); m->d_mergecnt.release(); m->d_u16.release(); m->d_x0.release(); m->d_x1.release(); m->d_x2.release();
    m->h_i0.release(); m->h_i1.release(); m->h_i2.release(); m->h_match.release(); m->h_u16.release(); m->h_ring.release();
    m->stage_f.release(); m->stage_q.release();
    if (m->ev_stage_f) (void)hipEventDestroy(m->ev_stage_f);
    for (FrameBufs* b : m->pool) { b->release(); delete b; }
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

void* orbm_stream(const orbm_matcher* m) { return m ? (void*)m->stream : nullptr; }

int orbm_debug_last_resolve(const orbm_matcher* m, int* out4) {
    MORB_ARG(m && out4);
    for (int k = 0; k < 4; ++k) out4[k] = m->last_status[k];
    return ORB_OK;
}

int orbm_wait_for_stream(orbm_matcher* m, void* other_stream) {
    MORB_ARG(m != nullptr);
    MORB_HIP(hipSetDevice(m->device));
    // everything enqueued on `other_stream` so far happens before whatever this handle enqueues next (no host wait)
    MORB_HIP(hipEventRecord(m->ev_fork, (hipStream_t)other_stream));
    MORB_HIP(hipStreamWaitEvent(m->stream, m->ev_fork, 0));
    m->foreign_work = true;
    return ORB_OK;
}

int orbm_set_stream(orbm_matcher* m, void* stream) {
    MORB_ARG(m != nullptr);
    MORB_HIP(hipSetDevice(m->device));
    MORB_HIP(hipStreamSynchronize(m->stream));  // nothing of ours may still be in flight on the old stream
    m->stream = stream ? (hipStream_t)stream : m->own_stream;
    return ORB_OK;
}

int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8); memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

int orbm_set_calibration(orbm_matcher* m, const orb_calibration* calib) {
    MORB_ARG(m != nullptr);
    if (calib) m->calib = *calib; else memset(&m->calib, 0, sizeof(m->calib));
    return ORB_OK;
}

int orbm_undistort_points(const orb_calibration* calib, const float* x, const float* y, int n, float* ux, float* uy) {
    MORB_ARG(n >= 0 && (n == 0 || (x && y && ux && uy)));
    for (int i = 0; i < n; ++i) {
        if (!calib || calib->k1 == 0.0f) { ux[i] = x[i]; uy[i] = y[i]; }
        else morb_undistort_point(*calib, x[i], y[i], &ux[i], &uy[i]);
    }
    return ORB_OK;
}

int orbm_image_bounds(const orb_calibration* calib, int cols, int rows, float* out4) {
    MORB_ARG(out4 != nullptr);
    if (!calib || calib->k1 == 0.0f) { out4[0] = 0.f; out4[1] = 0.f; out4[2] = (float)cols; out4[3] = (float)rows; return ORB_OK; }
    const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; ++i) morb_undistort_point(*calib, cx[i], cy[i], &ux[i], &uy[i]);
    out4[0] = std::min(ux[0], ux[2]); out4[2] = std::max(ux[1], ux[3]);  // src/Frame.cc:768-771
    out4[1] = std::min(uy[0], uy[1]); out4[3] = std::max(uy[2], uy[3]);
    return ORB_OK;
}

int orbm_queries_from_motion(const orb_keypoint* kps, const uint8_t* desc, const float* depth, const int32_t* cam_of, int n,
                             float du, float dv, float th, const float* scale_factors, float mbf, orbm_query* out,
                             const float* un_x, const float* un_y) {
    MORB_ARG(n >= 0 && (n == 0 || (kps && desc && depth && cam_of && scale_factors && out)) && ((un_x == nullptr) == (un_y == nullptr)));
    for (int i = 0; i < n; ++i) {
        orbm_query& Q = out[i];
        const orb_keypoint& k = kps[i];
        const float u = (un_x ? un_x[i] : k.x) + du;
        Q.u = u; Q.v = (un_y ? un_y[i] : k.y) + dv;
        Q.radius = scale_factors[k.octave] * th;
        const float inv = depth[i] > 0 ? 1.0f / depth[i] : 0.0f;
        Q.ur = u - mbf * inv;
        Q.min_level = k.octave - 1; Q.max_level = k.octave + 1;
        Q.cam = cam_of[i]; Q.blocks = 1; Q.angle = k.angle;
        memcpy(Q.desc, desc + (size_t)i * 32, 32);
    }
    return ORB_OK;
}

int orbm_count_ratio_accepted(const int32_t* best_dist, const int32_t* second_dist, int n, int th_low, float ratio) {
    // SearchByBoW's acceptance (reference src/ORBmatcher.cc:324-327): best <= TH_LOW and best < ratio * second (float compare)
    if (n < 0 || (n > 0 && (!best_dist || !second_dist))) return ORB_E_ARG;
    int acc = 0;
    for (int i = 0; i < n; ++i)
        acc += (best_dist[i] <= th_low && (float)best_dist[i] < ratio * (float)second_dist[i]) ? 1 : 0;
    return acc;
}

void orbm_three_maxima(const int* histo, int L, int* ind) {
    // Keeps the three fullest bins; an earlier bin wins a tie (strict '>'), 2nd/3rd dropped below 10% of the 1st.
    int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > m1) { m3 = m2; i3 = i2; m2 = m1; i2 = i1; m1 = s; i1 = i; }
        else if (s > m2) { m3 = m2; i3 = i2; m2 = s; i2 = i; }
        else if (s > m3) { m3 = s; i3 = i; }
    }
    if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
    else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
}

extern "C" int morb_debug_phases_matcher(int which, unsigned long long* out64) {   // (experiments: csrc/Makefile PHASES=1)
    return which == 0 ? morb::phases_resolve(out64) : morb::phases_frame_build(out64);
}
