// Cost of forking a dependent chain of short kernels over two or three streams and joining it again (plain launches, events
// without timing), against the same kernels in one stream.  Decides whether the extraction chain of an isolated timestep can be
// split by pyramid level.   hipcc --offload-arch=gfx950 -O2 fork_join.hip -o fork_join && ./fork_join
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}

int main(int argc, char** argv) {
    const int unit = argc > 1 ? atoi(argv[1]) : 5;   // microseconds per kernel
    hipStream_t s[3];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t t0, t1, e[8];
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (auto& x : e) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));   // kHz
    const long long us5 = (long long)rate * unit / 1000;
    printf("kernel unit: %d us\n", unit);
    auto spin = [&](hipStream_t st, int n_us5) { hipLaunchKernelGGL(k_spin, dim3(16), dim3(256), 0, st, us5 * n_us5, (int*)nullptr); };
    auto run = [&](const char* name, auto&& body) {
        std::vector<float> ms;
        for (int it = 0; it < 60; ++it) {
            hipDeviceSynchronize();
            hipEventRecord(t0, s[0]);
            body();
            hipEventRecord(t1, s[0]);
            hipEventSynchronize(t1);
            float m = 0; hipEventElapsedTime(&m, t0, t1);
            if (it >= 10) ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-64s median %.1f us  (p10 %.1f, p90 %.1f)\n", name, 1e3 * ms[ms.size() / 2], 1e3 * ms[ms.size() / 10], 1e3 * ms[ms.size() * 9 / 10]);
    };
    run("one stream: 8 x 5 us", [&] { for (int i = 0; i < 8; ++i) spin(s[0], 1); });
    run("one stream: 1 x 40 us", [&] { spin(s[0], 8); });
    run("two streams: 5 | (3x5 || 3x5) | 5   (ideal 25)", [&] {
        spin(s[0], 1);
        hipEventRecord(e[0], s[0]); hipStreamWaitEvent(s[1], e[0], 0);
        for (int i = 0; i < 3; ++i) { spin(s[0], 1); spin(s[1], 1); }
        hipEventRecord(e[1], s[1]); hipStreamWaitEvent(s[0], e[1], 0);
        spin(s[0], 1);
    });
    run("three streams: 5 | (3x5 || 3x5 || 3x5) | 5   (ideal 25)", [&] {
        spin(s[0], 1);
        hipEventRecord(e[0], s[0]); hipStreamWaitEvent(s[1], e[0], 0); hipStreamWaitEvent(s[2], e[0], 0);
        for (int i = 0; i < 3; ++i) { spin(s[0], 1); spin(s[1], 1); spin(s[2], 1); }
        hipEventRecord(e[1], s[1]); hipStreamWaitEvent(s[0], e[1], 0);
        hipEventRecord(e[2], s[2]); hipStreamWaitEvent(s[0], e[2], 0);
        spin(s[0], 1);
    });
    run("three streams, side branches enqueued first", [&] {
        spin(s[0], 1);
        hipEventRecord(e[0], s[0]); hipStreamWaitEvent(s[1], e[0], 0); hipStreamWaitEvent(s[2], e[0], 0);
        for (int i = 0; i < 3; ++i) spin(s[1], 1);
        hipEventRecord(e[1], s[1]);
        for (int i = 0; i < 3; ++i) spin(s[2], 1);
        hipEventRecord(e[2], s[2]);
        for (int i = 0; i < 3; ++i) spin(s[0], 1);
        hipStreamWaitEvent(s[0], e[1], 0); hipStreamWaitEvent(s[0], e[2], 0);
        spin(s[0], 1);
    });
    run("two streams with a mid-chain dependency (side waits main twice)", [&] {
        spin(s[0], 1);
        hipEventRecord(e[0], s[0]); hipStreamWaitEvent(s[1], e[0], 0);
        spin(s[1], 1); spin(s[0], 1);
        hipEventRecord(e[3], s[0]); hipStreamWaitEvent(s[1], e[3], 0);
        spin(s[1], 1); spin(s[0], 1);
        spin(s[1], 1); spin(s[0], 1);
        hipEventRecord(e[1], s[1]); hipStreamWaitEvent(s[0], e[1], 0);
        spin(s[0], 1);
    });
    // a graph with the same fork / join, replayed
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        spin(s[0], 1);
        hipEventRecord(e[0], s[0]); hipStreamWaitEvent(s[1], e[0], 0); hipStreamWaitEvent(s[2], e[0], 0);
        for (int i = 0; i < 3; ++i) { spin(s[0], 1); spin(s[1], 1); spin(s[2], 1); }
        hipEventRecord(e[1], s[1]); hipStreamWaitEvent(s[0], e[1], 0);
        hipEventRecord(e[2], s[2]); hipStreamWaitEvent(s[0], e[2], 0);
        spin(s[0], 1);
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        run("graph replay of the three-stream form", [&] { hipGraphLaunch(ge, s[0]); });
        std::vector<double> host;
        for (int it = 0; it < 50; ++it) {
            hipDeviceSynchronize();
            timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
            hipGraphLaunch(ge, s[0]);
            clock_gettime(CLOCK_MONOTONIC, &b);
            host.push_back((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3);
        }
        std::sort(host.begin(), host.end());
        printf("host cost of that hipGraphLaunch: median %.1f us\n", host[host.size() / 2]);
    }
    hipDeviceSynchronize();
    return 0;
}
