// what does hipMemsetAsync launch for 2 GB?  (run under rocprofv3 --kernel-trace; grid / workgroup size in the trace)
#include <hip/hip_runtime.h>
#include <cstdio>
int main() { void* d; hipMalloc(&d, (size_t)2048000000); for (int i = 0; i < 3; ++i) hipMemsetAsync(d, 1, (size_t)2048000000, 0); hipDeviceSynchronize(); return 0; }
