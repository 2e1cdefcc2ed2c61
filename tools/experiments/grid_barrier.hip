// Micro-benchmark: cost of a software grid barrier (one 64-bit ticket counter in HBM, device-scope atomics) on gfx950,
// against the cost of a dependent kernel launch on the same stream.  hipcc --offload-arch=gfx950 -O3 grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned long long* ctr, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE);  // agent scope by default for global atomics
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void k_bar(unsigned long long* ctr, unsigned long long base, int nbar, int* sink) {
    int acc = 0;
    for (int k = 0; k < nbar; ++k) {
        acc += k;
        grid_barrier(ctr, base + (unsigned long long)(k + 1) * gridDim.x);
    }
    if (acc == -1) *sink = acc;
}
__global__ void k_empty(int* sink) { if (threadIdx.x == 9999) *sink = 1; }

int main() {
    unsigned long long* ctr; int* sink;
    hipMalloc(&ctr, 8); hipMalloc(&sink, 4); hipMemset(ctr, 0, 8);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long base = 0;
    for (int G : {64, 128, 256, 512}) {
        for (int nbar : {0, 8, 64}) {
            float best = 1e9;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, st, ctr, base, nbar, sink);
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                base += (unsigned long long)nbar * G;
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("G=%d nbar=%d  %.2f us\n", G, nbar, best * 1000);
        }
    }
    for (int n : {1, 8, 64}) {
        float best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0, st);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("chain of %d empty kernels  %.2f us\n", n, best * 1000);
    }
    return 0;
}
