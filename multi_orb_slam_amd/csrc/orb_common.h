// orb_common.h -- shared host-side helpers of the HIP library (error plumbing, small utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

}  // namespace morb
