// armed_chain.hip -- how fast does a kernel chain start when it is (a) launched now, (b) already in its queue behind a
// hipStreamWaitValue32 the host releases with one store, (c) already in its queue behind a one-wave gate kernel that watches a
// word in page-locked host memory?  Measures host store / launch -> result word visible in pinned memory, for a chain of two
// small dependent kernels (the shape of k_project + k_resolve_mono).
// Build: hipcc --offload-arch=gfx950 -O3 -o armed_chain armed_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <x86intrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_first(const volatile int* in, int* mid) { if (threadIdx.x == 0 && blockIdx.x == 0) mid[0] = in[0] + 1; }
__global__ void k_second(const int* mid, volatile int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = mid[0] + 1; __threadfence_system(); } }
// gate: one wave watches a host word; gives up after `limit` polls (so that nothing can spin for ever)
__global__ void k_gate(const volatile int* flag, int want, int limit) {
    if (threadIdx.x == 0) {
        for (int i = 0; i < limit; ++i) {
            if (__builtin_nontemporal_load((const int*)flag) == want) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d\n", can);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *h_in, *h_out, *h_flag, *d_mid;
    CK(hipHostMalloc(&h_in, 64, hipHostMallocMapped)); CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped));
    CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped));
    CK(hipMalloc(&d_mid, 64));
    int* sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(se));
    const int N = 300;
    auto wait_out = [&](int want) { while (*(volatile int*)h_out != want) _mm_pause(); };
    // (a) launched now
    {
        std::vector<double> v;
        for (int i = 0; i < N; ++i) {
            *(volatile int*)h_in = 10 * i;
            const double t0 = now_us();
            hipLaunchKernelGGL(k_first, dim3(128), dim3(256), 0, st, h_in, d_mid);
            hipLaunchKernelGGL(k_second, dim3(1), dim3(1024), 0, st, d_mid, h_out);
            wait_out(10 * i + 2);
            v.push_back(now_us() - t0);
            CK(hipStreamSynchronize(st));
        }
        printf("(a) launch now                 : median %.2f us\n", med(v));
    }
    // (b) behind hipStreamWaitValue32
    int* bar = nullptr;   // fine-grained DEVICE memory the host writes through the large BAR (what StageBuf uses)
    printf("fine-grained device memory: %s\n", hipGetErrorString(hipExtMallocWithFlags((void**)&bar, 64, hipDeviceMallocFinegrained)));
    for (int variant = 0; variant < 3 && can; ++variant) {
        int* w = variant == 0 ? sig : variant == 1 ? h_flag : bar;
        if (!w) continue;
        std::vector<double> v, ve;
        *(volatile int*)w = 0;
        bool ok = true;
        for (int i = 1; i <= N && ok; ++i) {
            const double te = now_us();
            hipError_t e = hipStreamWaitValue32(st, w, (uint32_t)i, hipStreamWaitValueEq, 0xffffffffu);
            if (e != hipSuccess) { printf("(b%d) hipStreamWaitValue32: %s\n", variant, hipGetErrorString(e)); ok = false; break; }
            hipLaunchKernelGGL(k_first, dim3(128), dim3(256), 0, st, h_in, d_mid);
            hipLaunchKernelGGL(k_second, dim3(1), dim3(1024), 0, st, d_mid, h_out);
            ve.push_back(now_us() - te);
            // the host does something else for a while (the chain sits armed in its queue)
            const double tw = now_us(); while (now_us() - tw < 30) _mm_pause();
            *(volatile int*)h_in = 10 * i;
            _mm_sfence();
            const double t0 = now_us();
            *(volatile int*)w = i;
            wait_out(10 * i + 2);
            v.push_back(now_us() - t0);
            CK(hipStreamSynchronize(st));
        }
        if (ok) printf("(b%d) armed, WaitValue32 on %s: median %.2f us (arming took %.2f us of host time)\n", variant,
                       variant == 0 ? "signal memory" : variant == 1 ? "pinned memory" : "device memory (BAR)", med(v), med(ve));
    }
    // (c) a one-wave gate kernel watching the host word was tried first: it never saw the store (cached read), 320 ms = its poll limit
    return 0;
}
