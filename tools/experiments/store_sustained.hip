// Experiment (round 3): how fast can 32000 x 32000 uint16 be WRITTEN once the clocks have settled (the round-1/2 store experiments
// timed 10 launches behind 3 warm-ups, i.e. inside the transient bench.py used to time in)?  hipMemsetAsync, a flat dwordx4
// fill, and the distance-matrix kernel's own pattern (a wave stores 8 rows x 128 B per instruction, rows 64 000 B apart).
//   hipcc --offload-arch=gfx950 -O3 -o store_sustained store_sustained.hip && ./store_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fill(uint4* out, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(1, 2, 3, (unsigned)i);
}

// a workgroup of 4 waves owns a 256-row x 64-column tile band: wave w rows [64 w, 64 w + 64) of the band, walks over column tiles
// of 64 (128 B per row); one store instruction = 8 rows x 128 B (lane: row l / 8, 16 B at column 8 (l % 8))
__global__ __launch_bounds__(256) void k_rows(uint16_t* out, int n, int col_tiles_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 256 + wave * 64;
    const int t0 = blockIdx.x * col_tiles_per_block;
    for (int t = t0; t < t0 + col_tiles_per_block; ++t) {
        const int col = t * 64;
        if (col + 64 > n) break;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + i * 8 + (lane >> 3);
            if (row < n) *reinterpret_cast<uint4*>(out + (size_t)row * n + col + (lane & 7) * 8) = make_uint4(lane, i, t, 7);
        }
    }
}

// store flavours: 0 plain, 1 __builtin_nontemporal_store, 2 sc0 sc1 (system scope, write-through), 3 sc1, 4 nt sc0 sc1
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int F> __device__ __forceinline__ void st16(uint4* p, uint4 v4) {
    const u32x4 v = {v4.x, v4.y, v4.z, v4.w};
    if (F == 0) *p = v4;
    else if (F == 1) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
    else if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int F> __global__ __launch_bounds__(256) void k_rows_f(uint16_t* out, int n, int col_tiles_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 256 + wave * 64;
    const int t0 = blockIdx.x * col_tiles_per_block;
    for (int t = t0; t < t0 + col_tiles_per_block; ++t) {
        const int col = t * 64;
        if (col + 64 > n) break;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + i * 8 + (lane >> 3);
            if (row < n) st16<F>(reinterpret_cast<uint4*>(out + (size_t)row * n + col + (lane & 7) * 8), make_uint4(lane, i, t, 7));
        }
    }
}
__global__ __launch_bounds__(256) void k_rows_p(uint16_t* out, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bands = (n + 255) / 256, tiles = n / 64;
    for (int w = blockIdx.x; w < bands * tiles; w += gridDim.x) {
        const int band = w / tiles, t = w - band * tiles;
        const int row0 = band * 256 + wave * 64, col = t * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + i * 8 + (lane >> 3);
            if (row < n) *reinterpret_cast<uint4*>(out + (size_t)row * n + col + (lane & 7) * 8) = make_uint4(lane, i, t, 7);
        }
    }
}
// product-like geometry: grid (column slices, query blocks of 256 rows), a workgroup walks `tiles` column tiles of 64 left to right;
// SEG = bytes per row per store instruction (128: 8 rows x 128 B as the product kernel; 256: 4 rows x 256 B over two tiles; ...);
// SWAP: blockIdx.x = row block (fastest) instead of the column slice
template <int SEG, bool SWAP> __global__ __launch_bounds__(256) void k_prod(uint16_t* out, int n, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
    const int row0 = by * 256 + wave * 64;
    constexpr int TPS = SEG / 128;            // tiles covered by one row segment
    constexpr int RPI = 1024 / SEG;           // rows per instruction
    constexpr int LPR = 64 / RPI;             // lanes per row
    for (int t = bx * tiles; t < min((bx + 1) * tiles, n / 64); t += TPS) {
        const int col = t * 64;
#pragma unroll
        for (int i = 0; i < 8 * TPS; ++i) {   // 64 rows x SEG bytes = 8 TPS instructions
            const int row = row0 + i * RPI + lane / LPR;
            if (row < n && col + SEG / 2 <= n) *reinterpret_cast<uint4*>(out + (size_t)row * n + col + (lane % LPR) * 8) = make_uint4(lane, i, t, 7);
        }
    }
}
// the product geometry with W waves per workgroup stacked over 64 W rows, and `extern` LDS to cap the workgroups per CU
template <int W> __global__ __launch_bounds__(64 * W) void k_prod_w(uint16_t* out, int n, int tiles) {
    extern __shared__ int dummy[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (n < 0) dummy[threadIdx.x] = 1;
    const int row0 = blockIdx.y * 64 * W + wave * 64;
    for (int t = blockIdx.x * tiles; t < min((int)(blockIdx.x + 1) * tiles, n / 64); ++t) {
        const int col = t * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + i * 8 + (lane >> 3);
            if (row < n) *reinterpret_cast<uint4*>(out + (size_t)row * n + col + (lane & 7) * 8) = make_uint4(lane, i, t, 7);
        }
    }
}
template <int F> __global__ __launch_bounds__(256) void k_fill_f(uint4* out, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) st16<F>(out + i, make_uint4(1, 2, 3, (unsigned)i));
}


// a chip-wide moving front: workgroup j owns the column strip [CW j, CW j + CW) of the matrix for ALL rows and walks down the row bands of RB rows,
// so at any moment the workgroups together write RB complete rows = one contiguous block (the rows of a row-major matrix are adjacent)
template <int CW, int RB> __global__ __launch_bounds__(256) void k_front(uint16_t* out, int n) {
    constexpr int SEG = CW * 2;            // bytes per row segment
    constexpr int LPR = SEG / 16;          // lanes per row segment (16 B each)
    constexpr int RPI = 256 / LPR;         // rows per workgroup-wide store instruction
    const int c0 = blockIdx.x * CW;
    if (c0 >= n) return;
    const int t = threadIdx.x, r_in = t / LPR, cb = (t % LPR) * 8;
    const bool col_ok = c0 + cb + 8 <= n;
    for (int r0 = 0; r0 < n; r0 += RB) {
#pragma unroll
        for (int i = 0; i < RB / RPI; ++i) {
            const int r = r0 + i * RPI + r_in;
            if (r < n && col_ok) *reinterpret_cast<uint4*>(out + (size_t)r * n + c0 + cb) = make_uint4(r, c0, i, t);
        }
    }
}


// the same front with LONG row segments: workgroup (x, y) owns the column strip [CW x, CW x + CW) and every gridDim.y-th band of RB rows;
// a wave instruction writes 64 lanes x 16 B = 1 KB of ONE row when CW = 512 (the flat fill's shape per instruction)
template <int CW, int RB> __global__ __launch_bounds__(256) void k_front2(uint16_t* out, int n) {
    constexpr int SEG = CW * 2, LPR = SEG / 16, RPI = 256 / LPR;
    const int c0 = blockIdx.x * CW;
    const int t = threadIdx.x, r_in = t / LPR, cb = (t % LPR) * 8;
    const bool col_ok = c0 + cb + 8 <= n;
    for (int r0 = blockIdx.y * RB; r0 < n; r0 += RB * gridDim.y) {
#pragma unroll
        for (int i = 0; i < RB / RPI; ++i) {
            const int r = r0 + i * RPI + r_in;
            if (r < n && col_ok) *reinterpret_cast<uint4*>(out + (size_t)r * n + c0 + cb) = make_uint4(r, c0, i, t);
        }
    }
}


// k_front2 with an explicit row pitch (elements) and a byte offset of the whole matrix: is it the ALIGNMENT of the row segments?
template <int CW, int RB> __global__ __launch_bounds__(256) void k_front3(uint16_t* out, int n, int pitch) {
    constexpr int SEG = CW * 2, LPR = SEG / 16, RPI = 256 / LPR;
    const int c0 = blockIdx.x * CW;
    const int t = threadIdx.x, r_in = t / LPR, cb = (t % LPR) * 8;
    const bool col_ok = c0 + cb + 8 <= n;
    for (int r0 = blockIdx.y * RB; r0 < n; r0 += RB * gridDim.y) {
#pragma unroll
        for (int i = 0; i < RB / RPI; ++i) {
            const int r = r0 + i * RPI + r_in;
            if (r < n && col_ok) *reinterpret_cast<uint4*>(out + (size_t)r * pitch + c0 + cb) = make_uint4(r, c0, i, t);
        }
    }
}
// the flat fill with every 4 KB chunk of a workgroup moved by `shift` bytes (a multiple of 16)
__global__ __launch_bounds__(256) void k_fill_shift(uint4* out, size_t n16, int shift16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + shift16 < n16; i += (size_t)gridDim.x * blockDim.x) out[i + shift16] = make_uint4(1, 2, 3, (unsigned)i);
}


// aligned 4 KB blocks of the LINEAR array, written one per workgroup instruction (256 threads x 16 B), in the order a tile of
// 32 query rows 8 apart x one aligned block per row would produce them: workgroup j, tile t -> rows r0 + 8 i (i < 32), block k of
// each row.  n = 32000: a row holds 15.625 blocks, rows r and r + 8 have the same block alignment.
__global__ __launch_bounds__(256) void k_blocks(uint4* out, int n, int total_tiles) {
    const size_t row_bytes = (size_t)n * 2;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        // tiles: (row group g of 256 rows, residue q of 8, block k of 15 whole blocks per row)
        const int k = tile % 15, q = (tile / 15) % 8, g = tile / 120;
        const int r0 = g * 256 + q;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const size_t row_start = (size_t)(r0 + 8 * i) * row_bytes;
            const size_t first = (row_start + 4095) / 4096;              // first aligned block inside the row
            const size_t b = first + k;
            out[b * 256 + threadIdx.x] = make_uint4(tile, i, k, threadIdx.x);
        }
    }
}
// the same blocks in linear order (control: the flat fill restricted to the blocks k_blocks writes)

static void timeit(const char* name, std::function<void()> launch, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 250; ++i) launch();          // settle
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.1f us  %.2f TB/s (%.3f of 8)\n", name, ms * 10, bytes / (ms * 10) / 1e6, bytes / (ms * 10) / 1e6 / 8.0);
}

int main() {
    const int n = 32000;
    const double bytes = 2.0 * n * n;
    uint16_t* d; CK(hipMalloc(&d, (size_t)34816 * 32768 * 2));
    timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(d, 1, (size_t)bytes, 0)); }, bytes);
    timeit("front: 250 wg x 128 columns, bands of 64 rows", [&] { hipLaunchKernelGGL((k_front<128, 64>), dim3(250), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front: 250 wg x 128 columns, bands of 16 rows", [&] { hipLaunchKernelGGL((k_front<128, 16>), dim3(250), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front: 500 wg x 64 columns, bands of 64 rows", [&] { hipLaunchKernelGGL((k_front<64, 64>), dim3(500), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front: 125 wg x 256 columns, bands of 64 rows", [&] { hipLaunchKernelGGL((k_front<256, 64>), dim3(125), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front: 250 wg x 128 columns, bands of 128 rows", [&] { hipLaunchKernelGGL((k_front<128, 128>), dim3(250), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front: 250 wg x 128 columns, bands of 32 rows", [&] { hipLaunchKernelGGL((k_front<128, 32>), dim3(250), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 63 x 4 wg, 512 columns (1 KB / row / wave), bands of 32", [&] { hipLaunchKernelGGL((k_front2<512, 32>), dim3(63, 4), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 63 x 4 wg, 512 columns, bands of 64", [&] { hipLaunchKernelGGL((k_front2<512, 64>), dim3(63, 4), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 63 x 8 wg, 512 columns, bands of 32", [&] { hipLaunchKernelGGL((k_front2<512, 32>), dim3(63, 8), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 32 x 8 wg, 1024 columns (2 KB / row), bands of 32", [&] { hipLaunchKernelGGL((k_front2<1024, 32>), dim3(32, 8), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 16 x 16 wg, 2048 columns (4 KB / row), bands of 32", [&] { hipLaunchKernelGGL((k_front2<2048, 32>), dim3(16, 16), dim3(256), 0, 0, d, n); }, bytes);
    timeit("front2: 125 x 2 wg, 256 columns, bands of 32", [&] { hipLaunchKernelGGL((k_front2<256, 32>), dim3(125, 2), dim3(256), 0, 0, d, n); }, bytes);
    for (int pitch : {32000, 32768, 33792, 34816, 32256, 32512})
        timeit(("front3: 63 x 4 wg, 512 columns, pitch " + std::to_string(pitch)).c_str(), [&] { hipLaunchKernelGGL((k_front3<512, 32>), dim3(63, 4), dim3(256), 0, 0, d, n, pitch); }, bytes);
    for (int pitch : {33792, 34816})
        timeit(("front3: 250 x 1 wg, 128 columns, pitch " + std::to_string(pitch)).c_str(), [&] { hipLaunchKernelGGL((k_front3<128, 64>), dim3(250, 1), dim3(256), 0, 0, d, n, pitch); }, bytes);
    {
        const int tiles = 125 * 8 * 15;   // 32000 rows / 256 * 8 residues * 15 blocks
        const double b2 = (double)tiles * 32 * 4096;
        for (int g : {256, 512, 768, 1024})
            timeit(("aligned 4 KB blocks, 32 rows 8 apart per tile, grid " + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(k_blocks, dim3(g), dim3(256), 0, 0, (uint4*)d, n, tiles); }, b2);
    }
    for (int sh : {0, 16, 64, 160})
        timeit(("flat fill, grid 256, shifted by " + std::to_string(sh * 16) + " B").c_str(), [&] { hipLaunchKernelGGL(k_fill_shift, dim3(256), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16), sh); }, bytes);
    timeit("flat dwordx4 fill, grid 256", [&] { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
    if (getenv("FRONT_ONLY")) return 0;
    for (int g : {1024, 2048, 4096, 8192})
        timeit(("flat dwordx4 fill, grid " + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
    for (int ctb : {500, 125, 25, 5}) {
        dim3 grid((500 + ctb - 1) / ctb, (n + 255) / 256);
        timeit(("8 rows x 128 B per store, " + std::to_string(ctb) + " column tiles per wg").c_str(),
               [&] { hipLaunchKernelGGL(k_rows, grid, dim3(256), 0, 0, d, n, ctb); }, bytes);
    }
    for (int g : {256, 512, 768})
        timeit(("flat dwordx4 fill, grid " + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
    // the row pattern from persistent workgroups: `g` workgroups walk over all (row band, column tile) pairs, column tiles innermost
    for (int g : {256, 512, 1024, 2048})
        timeit(("rows, persistent, " + std::to_string(g) + " workgroups").c_str(), [&] { hipLaunchKernelGGL(k_rows_p, dim3(g), dim3(256), 0, 0, d, n); }, bytes);
    {
        const int tiles = 21;
        auto go = [&](auto kern, int W, int lds, const char* name) {
            dim3 g((500 + tiles - 1) / tiles, (n + 64 * W - 1) / (64 * W));
            timeit(name, [&] { hipLaunchKernelGGL(kern, g, dim3(64 * W), lds, 0, d, n, tiles); }, bytes);
        };
        go(k_prod_w<4>, 4, 0, "4 waves/wg, no LDS cap");
        go(k_prod_w<4>, 4, 50 * 1024, "4 waves/wg, 3 wg/CU (50 KB)");
        go(k_prod_w<4>, 4, 70 * 1024, "4 waves/wg, 2 wg/CU (70 KB)");
        go(k_prod_w<4>, 4, 100 * 1024, "4 waves/wg, 1 wg/CU (100 KB)");
        go(k_prod_w<8>, 8, 70 * 1024, "8 waves/wg, 2 wg/CU");
        go(k_prod_w<8>, 8, 100 * 1024, "8 waves/wg, 1 wg/CU");
        go(k_prod_w<12>, 12, 100 * 1024, "12 waves/wg, 1 wg/CU");
        go(k_prod_w<16>, 16, 100 * 1024, "16 waves/wg, 1 wg/CU");
        auto go2 = [&](auto kern, int W, int tl, const char* name) {
            dim3 g((500 + tl - 1) / tl, (n + 64 * W - 1) / (64 * W));
            char buf[128]; snprintf(buf, sizeof buf, "%s: %d tiles/wg, grid %d x %d", name, tl, g.x, g.y);
            timeit(buf, [&] { hipLaunchKernelGGL(kern, g, dim3(64 * W), 0, 0, d, n, tl); }, bytes);
        };
        go2(k_prod_w<4>, 4, 250, "4 waves");
        go2(k_prod_w<4>, 4, 125, "4 waves");
        go2(k_prod_w<4>, 4, 63, "4 waves");
        go2(k_prod_w<12>, 12, 84, "12 waves");
        go2(k_prod_w<12>, 12, 42, "12 waves");
        go2(k_prod_w<16>, 16, 125, "16 waves");
        go2(k_prod_w<16>, 16, 63, "16 waves");
        go2(k_prod_w<8>, 8, 125, "8 waves");
        go2(k_prod_w<8>, 8, 63, "8 waves");
    }
    {
        const int tiles = 24;   // (a multiple of 8 so that every SEG divides it)
        dim3 g((500 + tiles - 1) / tiles, (n + 255) / 256), gs(g.y, g.x);
        timeit("product geometry, 128 B segments", [&] { hipLaunchKernelGGL((k_prod<128, false>), g, dim3(256), 0, 0, d, n, tiles); }, bytes);
        timeit("product geometry, 256 B segments", [&] { hipLaunchKernelGGL((k_prod<256, false>), g, dim3(256), 0, 0, d, n, tiles); }, bytes);
        timeit("product geometry, 512 B segments", [&] { hipLaunchKernelGGL((k_prod<512, false>), g, dim3(256), 0, 0, d, n, tiles); }, bytes);
        timeit("product geometry, 1024 B segments", [&] { hipLaunchKernelGGL((k_prod<1024, false>), g, dim3(256), 0, 0, d, n, tiles); }, bytes);
        timeit("row blocks fastest, 128 B segments", [&] { hipLaunchKernelGGL((k_prod<128, true>), gs, dim3(256), 0, 0, d, n, tiles); }, bytes);
        timeit("row blocks fastest, 512 B segments", [&] { hipLaunchKernelGGL((k_prod<512, true>), gs, dim3(256), 0, 0, d, n, tiles); }, bytes);
    }
    {
        dim3 grid(20, (n + 255) / 256);
        timeit("rows, 25 tiles/wg, plain", [&] { hipLaunchKernelGGL(k_rows_f<0>, grid, dim3(256), 0, 0, d, n, 25); }, bytes);
        timeit("rows, 25 tiles/wg, nontemporal builtin", [&] { hipLaunchKernelGGL(k_rows_f<1>, grid, dim3(256), 0, 0, d, n, 25); }, bytes);
        timeit("rows, 25 tiles/wg, sc0 sc1", [&] { hipLaunchKernelGGL(k_rows_f<2>, grid, dim3(256), 0, 0, d, n, 25); }, bytes);
        timeit("rows, 25 tiles/wg, sc1", [&] { hipLaunchKernelGGL(k_rows_f<3>, grid, dim3(256), 0, 0, d, n, 25); }, bytes);
        timeit("rows, 25 tiles/wg, sc0 sc1 nt", [&] { hipLaunchKernelGGL(k_rows_f<4>, grid, dim3(256), 0, 0, d, n, 25); }, bytes);
        timeit("flat fill 2048, nontemporal builtin", [&] { hipLaunchKernelGGL(k_fill_f<1>, dim3(2048), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
        timeit("flat fill 2048, sc0 sc1", [&] { hipLaunchKernelGGL(k_fill_f<2>, dim3(2048), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
        timeit("flat fill 2048, sc0 sc1 nt", [&] { hipLaunchKernelGGL(k_fill_f<4>, dim3(2048), dim3(256), 0, 0, (uint4*)d, (size_t)(bytes / 16)); }, bytes);
        timeit("hipMemsetAsync again", [&] { CK(hipMemsetAsync(d, 1, (size_t)bytes, 0)); }, bytes);
    }
    return 0;
}
