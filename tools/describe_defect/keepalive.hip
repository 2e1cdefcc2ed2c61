// keepalive: a second process that keeps a few compute units of the part busy for N seconds (back-to-back launches of a small ALU
// kernel, WG workgroups of 256 threads, ~200 us each) -- run next to tools/describe_defect/run_rig.py to see whether the k_describe
// defect needs the part to pass through idle phases (clock / power-state transitions) or only the rig's own load pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o keepalive keepalive.hip      run: ./keepalive SECONDS [WG]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_spin(unsigned* sink, int rounds) {
    unsigned v = threadIdx.x * 2654435761u + blockIdx.x;
    for (int r = 0; r < rounds; ++r) v = v * 1664525u + 1013904223u + (v >> 7);
    if (v == 0xdeadbeefu) sink[0] = v;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 30.0;
    const int wg = argc > 2 ? std::atoi(argv[2]) : 16;
    unsigned* sink;
    if (hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 32; ++k) hipLaunchKernelGGL(k_spin, dim3(wg), dim3(256), 0, st, sink, 60000);
        if (hipStreamSynchronize(st) != hipSuccess) return 2;
        n += 32;
    }
    std::printf("keepalive: %ld launches of %d workgroups in %.0f s\n", n, wg, seconds);
    return 0;
}
