// Experiment: how fast can 2 GB of a row-major 32000 x 32000 uint16 matrix be written when a 256-thread workgroup emits,
// per step, a block of RW rows x CW bytes (RW * CW = 32 KB), 1 KB per wave-level store instruction?  (Which block
// shape does the distance-matrix kernel have to produce to reach the memset rate?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CW>  // bytes per row per workgroup step
__global__ __launch_bounds__(256) void k_store(uint16_t* out, int n, int steps_per_block) {
    constexpr int RW = 32768 / CW;          // rows per workgroup step
    constexpr int LPR = CW >= 1024 ? 64 : CW / 16;   // lanes per row within one instruction
    constexpr int RPI = 64 / LPR;           // rows per instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col_blocks = (n * 2) / CW;    // per row
    const int rb = blockIdx.y;              // row block
    const uint4 v = make_uint4(lane, wave, rb, 1);
    for (int s = 0; s < steps_per_block; ++s) {
        const int cb = blockIdx.x * steps_per_block + s;
        if (cb >= col_blocks) break;
        // wave handles rows [wave * RW / 4, +RW / 4) of the block, all CW bytes of each
        constexpr int ROWS_PER_WAVE = RW / 4;
        constexpr int INSTR_PER_ROW = CW >= 1024 ? CW / 1024 : 1;
        if (CW >= 1024) {
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; ++r)
#pragma unroll
                for (int k = 0; k < INSTR_PER_ROW; ++k) {
                    const size_t row = (size_t)rb * RW + wave * ROWS_PER_WAVE + r;
                    char* p = (char*)out + row * (size_t)n * 2 + (size_t)cb * CW + k * 1024 + lane * 16;
                    *reinterpret_cast<uint4*>(p) = v;
                }
        } else {
#pragma unroll
            for (int i = 0; i < ROWS_PER_WAVE / RPI; ++i) {
                const size_t row = (size_t)rb * RW + wave * ROWS_PER_WAVE + i * RPI + lane / LPR;
                char* p = (char*)out + row * (size_t)n * 2 + (size_t)cb * CW + (lane % LPR) * 16;
                *reinterpret_cast<uint4*>(p) = v;
            }
        }
    }
}

template <int CW>
void run(uint16_t* d, int n, int spb) {
    constexpr int RW = 32768 / CW;
    const int col_blocks = (n * 2) / CW;
    dim3 grid((col_blocks + spb - 1) / spb, n / RW);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_store<CW>, grid, dim3(256), 0, 0, d, n, spb);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_store<CW>, grid, dim3(256), 0, 0, d, n, spb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("block %4d rows x %5d B, %3d steps/wg, grid %4d x %4d: %7.1f us  %.2f TB/s\n", RW, CW, spb, grid.x, grid.y, ms * 100, 2.0 * n * n / (ms * 100) / 1e6);
}

int main() {
    const int n = 32000;  // 64000-byte rows: divisible by 128, 256, 512 (125), not by 1024 -> use n = 32768 for those
    uint16_t* d; CK(hipMalloc(&d, (size_t)32768 * 32768 * 2));
    CK(hipMemset(d, 0, (size_t)32768 * 32768 * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(d, 0, (size_t)n * n * 2));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("memset: %.1f us %.2f TB/s\n", ms * 100, 2.0 * n * n / (ms * 100) / 1e6);
    for (int spb : {8, 32}) {
        run<128>(d, n, spb); run<256>(d, n, spb); run<512>(d, n, spb);
    }
    const int m = 32768;
    for (int spb : {8, 32}) {
        run<128>(d, m, spb); run<256>(d, m, spb); run<512>(d, m, spb); run<1024>(d, m, spb); run<2048>(d, m, spb); run<4096>(d, m, spb);
    }
    return 0;
}
