// Is fine-grained device memory writable from the host (large BAR)?  Host writes N bytes, a kernel sums them.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <csignal>
#include <csetjmp>
static sigjmp_buf jb;
static void onsegv(int) { siglongjmp(jb, 1); }
__global__ void k_sum(const unsigned* p, int n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
int main() {
    const int n = 34000;  // 136 KB
    unsigned* d = nullptr; unsigned long long* out = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&d, n * 4, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipHostMalloc((void**)&out, 8, hipHostMallocMapped);
    hipPointerAttribute_t at; hipPointerGetAttributes(&at, d);
    printf("type %d device %d host ptr %p device ptr %p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
    signal(SIGSEGV, onsegv); signal(SIGBUS, onsegv);
    if (sigsetjmp(jb, 1)) { printf("host write FAULTED: no host access to device memory\n"); return 2; }
    unsigned* src = new unsigned[n];
    for (int i = 0; i < n; ++i) src[i] = i;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        memcpy(d, src, n * 4);
        __sync_synchronize();
        auto t1 = std::chrono::steady_clock::now();
        *out = 0;
        k_sum<<<1, 256>>>(d, n, out);
        hipDeviceSynchronize();
        printf("host write %.1f us, kernel sum %llu (expect %llu)\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), *out, (unsigned long long)n * (n - 1) / 2);
    }
    return 0;
}
