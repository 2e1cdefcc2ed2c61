// Experiment (round 2): the store pattern of a distance-matrix kernel with the MFMA roles swapped -- references resident in
// the B fragments (a lane = one reference column), queries streamed as A: a wave owns 128 adjacent columns (256 B per row),
// the four waves of a workgroup own 512 adjacent columns (1 KB per row), and one store instruction of a wave writes two
// rows x 256 B (lanes 0..31 row a, lanes 32..63 row b, 8 bytes per lane).  A workgroup walks down the rows in steps of 32.
// How fast does that pattern write the 32000 x 32000 uint16 matrix?  (WCOLS = columns per wave: 128 -> dwordx2, 256 -> dwordx4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WCOLS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_store2(uint16_t* out, int n, int rows_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    constexpr int BYTES = WCOLS * 2 / 32;                       // bytes per lane per row
    const int col0 = (blockIdx.x * WAVES + wave) * WCOLS;       // first column of this wave
    if (col0 + WCOLS > n) return;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(n, r_begin + rows_per_block);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;  // the MFMA accumulator's row of register e
            if (row < r_end) {
                char* p = (char*)out + (size_t)row * n * 2 + (size_t)col0 * 2 + c * BYTES;
                if (BYTES == 8) *reinterpret_cast<uint2*>(p) = make_uint2(lane, e);
                else *reinterpret_cast<uint4*>(p) = make_uint4(lane, e, r0, 1);
            }
        }
    }
}

template <int WCOLS, int WAVES>
void run(uint16_t* d, int n, int rpb) {
    dim3 grid((n + WCOLS * WAVES - 1) / (WCOLS * WAVES), (n + rpb - 1) / rpb);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_store2<WCOLS, WAVES>), grid, dim3(64 * WAVES), 0, 0, d, n, rpb);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_store2<WCOLS, WAVES>), grid, dim3(64 * WAVES), 0, 0, d, n, rpb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("wave %3d cols x %d waves (%4d B/row/wg), %5d rows/wg, grid %3d x %3d: %7.1f us  %.2f TB/s\n", WCOLS, WAVES, WCOLS * WAVES * 2, rpb,
           grid.x, grid.y, ms * 100, 2.0 * n * n / (ms * 100) / 1e6);
}

int main() {
    const int n = 32000;
    uint16_t* d; CK(hipMalloc(&d, (size_t)32768 * 32768 * 2));
    CK(hipMemset(d, 0, (size_t)32768 * 32768 * 2));
    for (int rpb : {1024, 2048, 4000, 8000}) {
        run<128, 4>(d, n, rpb); run<256, 4>(d, n, rpb); run<128, 8>(d, n, rpb); run<256, 2>(d, n, rpb);
    }
    return 0;
}
