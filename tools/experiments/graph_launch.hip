// Micro-benchmark: host cost and end-to-end time of a 16-kernel dependent chain, stream launches vs one hipGraphLaunch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_small(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int NK = 16, REP = 200;
    for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, 1);
    hipStreamSynchronize(st);
    double host = 0, total = 0;
    for (int r = 0; r < REP; ++r) {
        const double t0 = now_us();
        for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, i);
        const double t1 = now_us();
        hipStreamSynchronize(st);
        const double t2 = now_us();
        host += t1 - t0; total += t2 - t0;
    }
    printf("stream: host enqueue %.1f us, end-to-end %.1f us (%d kernels)\n", host / REP, total / REP, NK);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 10; ++w) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    host = total = 0;
    for (int r = 0; r < REP; ++r) {
        const double t0 = now_us();
        hipGraphLaunch(ge, st);
        const double t1 = now_us();
        hipStreamSynchronize(st);
        const double t2 = now_us();
        host += t1 - t0; total += t2 - t0;
    }
    printf("graph : host enqueue %.1f us, end-to-end %.1f us\n", host / REP, total / REP);
    // graph + 3 kernel-node parameter updates per launch
    hipGraphNode_t nodes[64]; size_t nn = 64; hipGraphGetNodes(g, nodes, &nn);
    host = total = 0;
    for (int r = 0; r < REP; ++r) {
        const double t0 = now_us();
        for (int u = 0; u < 3; ++u) {
            hipKernelNodeParams kp; hipGraphKernelNodeGetParams(nodes[u], &kp);
            int v = r; void* args[2] = {&d, &v}; kp.kernelParams = args;
            hipGraphExecKernelNodeSetParams(ge, nodes[u], &kp);
        }
        hipGraphLaunch(ge, st);
        const double t1 = now_us();
        hipStreamSynchronize(st);
        const double t2 = now_us();
        host += t1 - t0; total += t2 - t0;
    }
    printf("graph + 3 param updates: host %.1f us, end-to-end %.1f us\n", host / REP, total / REP);
    // kernel-only graph with a fork: 1 -> {branch A: 2 kernels | branch B: 9 kernels} -> join -> 2 kernels
    {
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
        hipGraph_t g2; hipGraphExec_t ge2;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, 1);
        hipEventRecord(ef, st); hipStreamWaitEvent(s2, ef, 0);
        for (int i = 0; i < 9; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, d + 8, i);
        hipEventRecord(ej, s2);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, i);
        hipStreamWaitEvent(st, ej, 0);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, d, i);
        hipStreamEndCapture(st, &g2);
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        for (int w = 0; w < 10; ++w) hipGraphLaunch(ge2, st);
        hipStreamSynchronize(st);
        host = total = 0;
        for (int r = 0; r < REP; ++r) {
            const double t0 = now_us();
            hipGraphLaunch(ge2, st);
            const double t1 = now_us();
            hipStreamSynchronize(st);
            const double t2 = now_us();
            host += t1 - t0; total += t2 - t0;
        }
        printf("forked graph (1 + {2 | 9} + 2 kernels): host %.1f us, end-to-end %.1f us (a 12-kernel chain would be ~%.0f)\n", host / REP, total / REP, 12 * 2.6);
    }
    return 0;
}
