// runtime.hip -- include/orb_rt.h: thin, checked wrappers over the HIP runtime for non-HIP hosts.
#include "../../include/orb_rt.h"
#include "orb_common.h"

extern "C" {

int orb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orb_device_name(int device, char* out, int cap) {
    MORB_ARG(out && cap > 0);
    hipDeviceProp_t prop;
    MORB_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(out, cap, "%s", prop.gcnArchName);
    return ORB_OK;
}

int orb_set_device(int device) { MORB_HIP(hipSetDevice(device)); return ORB_OK; }
int orb_malloc(void** d_ptr, size_t bytes) { MORB_ARG(d_ptr); MORB_HIP(hipMalloc(d_ptr, bytes)); return ORB_OK; }
int orb_free(void* d_ptr) { MORB_HIP(hipFree(d_ptr)); return ORB_OK; }
int orb_malloc_host(void** h_ptr, size_t bytes) { MORB_ARG(h_ptr); MORB_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault)); return ORB_OK; }
int orb_free_host(void* h_ptr) { MORB_HIP(hipHostFree(h_ptr)); return ORB_OK; }

// "No stream" means synchronous, but NOT the legacy default stream: an operation on the legacy stream implicitly joins every
// blocking stream of the process and is refused outright while another thread captures a launch chain ("operation would make
// the legacy stream depend on a capturing stream").  Each thread keeps a private non-blocking stream per device for these calls.
// (The reference's call pattern uses short-lived threads: the streams go when their thread does.)
namespace {
struct PrivateStreams {
    static constexpr int MAX_DEV = 64;
    hipStream_t st[MAX_DEV] = {};
    ~PrivateStreams() {
        int cur = 0;
        const bool have_cur = hipGetDevice(&cur) == hipSuccess;
        for (int d = 0; d < MAX_DEV; ++d)
            if (st[d] && hipSetDevice(d) == hipSuccess) (void)hipStreamDestroy(st[d]);
        if (have_cur) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    }
};
}  // namespace
static hipStream_t private_stream() {
    static thread_local PrivateStreams P;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PrivateStreams::MAX_DEV) return nullptr;
    if (!P.st[dev] && hipStreamCreateWithFlags(&P.st[dev], hipStreamNonBlocking) != hipSuccess) P.st[dev] = nullptr;
    return P.st[dev];
}

static int copy(void* dst, const void* src, size_t bytes, void* stream, hipMemcpyKind kind) {
    if (bytes == 0) return ORB_OK;
    MORB_ARG(dst && src);
    if (stream) { MORB_HIP(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream)); return ORB_OK; }
    hipStream_t ps = private_stream();
    if (!ps) { morb::set_error("no private stream for a synchronous copy"); return ORB_E_HIP; }
    MORB_HIP(hipMemcpyAsync(dst, src, bytes, kind, ps));
    MORB_HIP(hipStreamSynchronize(ps));
    return ORB_OK;
}
int orb_memcpy_h2d(void* d, const void* h, size_t n, void* s) { return copy(d, h, n, s, hipMemcpyHostToDevice); }
int orb_memcpy_d2h(void* h, const void* d, size_t n, void* s) { return copy(h, d, n, s, hipMemcpyDeviceToHost); }
int orb_memcpy_d2d(void* d, const void* s_, size_t n, void* s) { return copy(d, s_, n, s, hipMemcpyDeviceToDevice); }
int orb_memset(void* d, int v, size_t n, void* s) {
    if (n == 0) return ORB_OK;
    MORB_ARG(d != nullptr);
    if (s) { MORB_HIP(hipMemsetAsync(d, v, n, (hipStream_t)s)); return ORB_OK; }
    hipStream_t ps = private_stream();
    if (!ps) { morb::set_error("no private stream for a synchronous memset"); return ORB_E_HIP; }
    MORB_HIP(hipMemsetAsync(d, v, n, ps));
    MORB_HIP(hipStreamSynchronize(ps));
    return ORB_OK;
}
int orb_stream_sync(void* stream) { MORB_HIP(hipStreamSynchronize((hipStream_t)stream)); return ORB_OK; }
int orb_device_sync(void) { MORB_HIP(hipDeviceSynchronize()); return ORB_OK; }
int orb_event_create(void** ev) { MORB_ARG(ev); hipEvent_t e; MORB_HIP(hipEventCreate(&e)); *ev = (void*)e; return ORB_OK; }
int orb_event_destroy(void* ev) { MORB_HIP(hipEventDestroy((hipEvent_t)ev)); return ORB_OK; }
int orb_event_record(void* ev, void* stream) { MORB_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return ORB_OK; }
int orb_event_elapsed_ms(void* a, void* b, float* ms) {
    MORB_ARG(ms != nullptr);
    MORB_HIP(hipEventSynchronize((hipEvent_t)b));
    MORB_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return ORB_OK;
}

}  // extern "C"
