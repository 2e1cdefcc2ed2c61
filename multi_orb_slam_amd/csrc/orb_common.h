// orb_common.h -- shared host-side helpers of the HIP library (error plumbing, small utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/orb_types.h"

// Phase clocks (experiments only, -DMORB_PHASE_CLOCKS): thread 0 of block 0 stamps wall_clock64() (100 MHz) at named
// points of the single-workgroup kernels; read back with morb_debug_phases().  Compiled out of the product build.
#ifdef MORB_PHASE_CLOCKS
#define MORB_PHASE_DECL(name) __device__ unsigned long long name[64]
#define MORB_PHASE(name, i) do { if (threadIdx.x == 0 && blockIdx.x == 0) name[i] = wall_clock64(); } while (0)
#else
#define MORB_PHASE_DECL(name)
#define MORB_PHASE(name, i) do {} while (0)
#endif

#ifdef __HIPCC__
// Kernels that are a handful of workgroups walking a chain of dependent steps (the quadtree, the resolves, the small-rig frame assembly, the
// cell scan) run next to launches that fill every wave slot of the chip; their waves ask for the SIMD's issue slots first (s_setprio 3).
// Measured on one box, three alternating runs each (tools/experiments/prio_ab.sh, profiles/r05/notes_experiments.md): configs[1] +2.1 % inside
// the library (+3.3 % through the binding), configs[4] unchanged.  The same on the WIDE latency kernels (k_project, k_frame_fill, the cell
// scatter / sort, k_rs_reject / k_rs_write, k_top2_merge: thousands of waves) costs configs[4] 1.9 % and gives configs[1] nothing: off.
// Compile time: MORB_WAVE_PRIO=0 nobody, MORB_WAVE_PRIO_WIDE=3 the wide ones as well.
#ifndef MORB_WAVE_PRIO
#define MORB_WAVE_PRIO 3
#endif
#define MORB_LATENCY_KERNEL() do { if (MORB_WAVE_PRIO) __builtin_amdgcn_s_setprio(MORB_WAVE_PRIO); } while (0)
#ifndef MORB_WAVE_PRIO_WIDE
#define MORB_WAVE_PRIO_WIDE 0
#endif
#define MORB_LATENCY_KERNEL_WIDE() do { if (MORB_WAVE_PRIO_WIDE) __builtin_amdgcn_s_setprio(MORB_WAVE_PRIO_WIDE); } while (0)
// Wave64 inclusive prefix sums on the DPP lanes-shift path (row_shr 1/2/4/8, then row_bcast 15 and 31): six VALU-rate
// steps instead of six ds_bpermute round trips per __shfl_up scan.  Every lane of the wave must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int morb_dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }

__device__ __forceinline__ int wave_incl_scan(int v) {
    v += morb_dpp0<0x111, 0xf>(v);
    v += morb_dpp0<0x112, 0xf>(v);
    v += morb_dpp0<0x114, 0xf>(v);
    v += morb_dpp0<0x118, 0xf>(v);
    v += morb_dpp0<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += morb_dpp0<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// wave64 minimum on the same DPP path (six VALU-rate steps; a __shfl_xor butterfly is six dependent LDS-crossbar round
// trips).  Every lane of the wave must be active; the result is returned in every lane.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long morb_dpp0_u64(unsigned long long v) {
    const unsigned int lo = (unsigned int)morb_dpp0<CTRL, ROW_MASK>((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)morb_dpp0<CTRL, ROW_MASK>((int)(unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_incl_scan(unsigned long long v) {
    v += morb_dpp0_u64<0x111, 0xf>(v);
    v += morb_dpp0_u64<0x112, 0xf>(v);
    v += morb_dpp0_u64<0x114, 0xf>(v);
    v += morb_dpp0_u64<0x118, 0xf>(v);
    v += morb_dpp0_u64<0x142, 0xa>(v);
    v += morb_dpp0_u64<0x143, 0xc>(v);
    return v;
}
#endif

// cv::undistortPoints(pts, K, dist, noArray(), K) for one point -- the OpenCV 2.4.x / 3.2 generic path as
// Frame::UndistortKeyPoints uses it (reference src/Frame.cc:692): float parameters promoted to double, five fixed-point
// iterations of the radial-tangential model, re-projection with P = K (the zero entries of K take part in the sums as they
// do in OpenCV), rounded to float.  One operation sequence for host and device (no contraction on either side).
#ifdef __HIPCC__
__host__ __device__
#endif
inline void morb_undistort_point(const orb_calibration& c, float xs, float ys, float* xo, float* yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, k1 = c.k1, k2 = c.k2, p1 = c.p1, p2 = c.p2, k3 = c.k3;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = xs, y = ys;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx;
    const double yy = 0. * x + fy * y + cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww); *yo = (float)(yy * ww);
}

namespace morb {

void set_error(const char* fmt, ...);

#define MORB_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            morb::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ORB_E_HIP;                                                                       \
        }                                                                                           \
    } while (0)

#define MORB_ARG(cond)                                                        \
    do {                                                                      \
        if (!(cond)) {                                                        \
            morb::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return ORB_E_ARG;                                                 \
        }                                                                     \
    } while (0)

// Selects `device` and verifies it is a gfx950 part.  The product has no CPU path: anything else is an error.
int select_device(int device);

template <typename T>
struct DevBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return ORB_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        MORB_HIP(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return ORB_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
struct PinnedBuf {  // grow-only pinned host buffer, mapped into the device address space (dp = device-visible alias)
    T* p = nullptr;
    T* dp = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return ORB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; dp = nullptr; cap = 0;
        MORB_HIP(hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocMapped));
        MORB_HIP(hipHostGetDevicePointer((void**)&dp, p, 0));
        cap = n;
        return ORB_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; }
};

// Host-written, device-read staging: fine-grained DEVICE memory written by the host through the large BAR when the part has
// one (136 KB of queries land in HBM in ~4 us and the kernel reads them locally), otherwise mapped pinned host memory (the
// kernel then fetches over PCIe).  p = host-usable address, dp = device address (the same pointer in the first case; host
// READS of BAR memory work but are slow -- only rare fallback paths read this buffer back).
struct StageBuf {
    uint8_t* p = nullptr;
    uint8_t* dp = nullptr;
    size_t cap = 0;
    bool in_hbm = false;
    int reserve(size_t n) {
        if (n <= cap) return ORB_OK;
        release();
        int dev = 0, large_bar = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { large_bar = 0; (void)hipGetLastError(); }
        const char* off = getenv("MORB_NO_BAR_STAGING");
        if (large_bar && !(off && atoi(off))) {
            void* q = nullptr;
            if (hipExtMallocWithFlags(&q, n, hipDeviceMallocFinegrained) == hipSuccess) { p = dp = (uint8_t*)q; cap = n; in_hbm = true; return ORB_OK; }
            (void)hipGetLastError();
        }
        MORB_HIP(hipHostMalloc((void**)&p, n, hipHostMallocMapped));
        MORB_HIP(hipHostGetDevicePointer((void**)&dp, p, 0));
        cap = n; in_hbm = false;
        return ORB_OK;
    }
    void release() {
        if (p) { if (in_hbm) (void)hipFree(p); else (void)hipHostFree(p); }
        p = nullptr; dp = nullptr; cap = 0; in_hbm = false;
    }
    // after the host has written: make the write-combined stores globally visible before a kernel is launched
    void publish() const { if (in_hbm) __builtin_ia32_sfence(); }
};

}  // namespace morb
