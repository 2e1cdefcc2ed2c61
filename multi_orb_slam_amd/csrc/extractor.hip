// extractor.hip -- gfx950 kernels + C ABI of the ORB extractor (include/orbx.h).
//
// Per call (all cameras batched in every launch; integer / byte work, no MFMA by design):
//   k_pyramid_tiled K1     every level of every camera in ONE launch (small rigs): a workgroup per level-0 tile computes what hangs
//   k_pyramid_tiled4       below it, level after level in LDS; large rigs: two such launches with four pixels per lane (levels 1-3 below
//   k_resize               level 0, 4.. below level 3); parameter sets outside both: one plain launch per level.  OpenCV's 11-bit
//                          fixed-point bilinear (reference src/ORBextractor.cc:1109-1134), coefficient tables from the host.
//   k_fast_cells    K2+K3  one workgroup per 30-px detection cell (reference src/ORBextractor.cc:790-830): image tile +
//                          3-px halo staged in LDS, threshold-free FAST-9/16 score per pixel (App. A-2), cell-local
//                          3x3 non-max suppression, per-cell threshold choice (iniTh, else minTh) and ordered
//                          (row-major) output into per-cell slots -- all in LDS.
//   k_octree        K4     DistributeOctTree on the device: one workgroup per (camera, level), keys in vector registers.
//   k_compact + host K4'   the exact host quadtree (octree.cpp) for levels beyond the device limits: dense cell-major candidate
//                          list written straight into pinned host memory.
//   k_describe      K5-K7  one wave per keypoint: 45x45 patch (reflect-101 at the level edge) staged in LDS,
//                          intensity-centroid moments + fastAtan2 on the raw patch, 7x7 sigma-2 fixed-point Gaussian
//                          of the 39x39 neighbourhood actually sampled, 256 steered rBRIEF tests, keypoint record.
// The reference's 19-px reflect-101 border of every level is never read by a later stage and is
// not materialised.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <chrono>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/orbx.h"
#include "../../include/orb_debug.h"
#include "octree.h"
#include "orb_common.h"
#include "frame_sink.h"

namespace {

using morb::DevBuf;

constexpr int EDGE_THRESHOLD = 19;   // reference src/ORBextractor.cc:74
constexpr int MIN_BORDER = 16;       // EDGE_THRESHOLD - 3, :772
constexpr int PATCH_SIZE = 31;       // :72
constexpr int HALF_PATCH = 15;       // :73
constexpr int CELL_MAX = 64;         // largest wCell/hCell this build stages in LDS
constexpr int TILE_PITCH = 76;       // LDS pitch of the image tile (CELL_MAX + 6, plus up to 3 bytes of dword alignment, rounded up to 4)
constexpr int SCORE_PITCH = 68;      // LDS pitch of the score map (CELL_MAX + 2 rounded up)
constexpr int MAX_LEVELS = 16;
constexpr int COEF_BITS = 11;        // OpenCV INTER_RESIZE_COEF_BITS

struct LevelInfo {
    int w, h, stride;        // level size, row pitch in bytes (multiple of 64)
    int pyr_off;             // byte offset inside the camera's pyramid buffer
    int n_cols, n_rows, w_cell, h_cell;  // reference :782-788
    int cell_base;           // first global cell id
    int slot_base, slot_cap; // per-cell candidate slots: slot_base + local_cell*slot_cap (u32 units)
    int cand_base;           // dense candidate list of this level (u32 units, in the pinned host buffer)
    int xtab_off, ytab_off;  // resize coefficient tables (valid for level >= 1)
    int xgrp_off;            // the x table once more per group of four destination columns (k_pyramid_tiled4), in groups
    int ini_th, min_th;
    float scale;             // mvScaleFactor[level]
    float patch_size;        // (float)(int)(31 * scale)
    int quota;               // mnFeaturesPerLevel[level]
    int sel_base;            // first slot of this (camera, level) in the device quadtree's output list (quota + 4 slots)
};

struct MirrorArgs {  // optional pinned-host (device-mapped) copy of the results in global, camera-major order
    orb_keypoint* kps; uint8_t* desc;
    int base[64];
};

struct SelListArgs {  // device-quadtree mode of k_describe: the slotted selection and where its bookkeeping goes
    const unsigned short* slot_blk;  // slot -> (camera, level) block; nullptr = host-list mode
    const int* sel_cnt;              // keypoints the quadtree kept per block
    const int* status;               // per-block quadtree status (non-zero: outside the device limits)
    int* n_out;                      // per-camera totals in HBM (downstream kernels size themselves from these) + the OR of `status`
    int* h_n_out;                    // the same + the OR of `status` behind them, in mapped pinned memory
    int n_cams;
};

struct SelKp {  // one keypoint chosen by the quadtree, input of k_describe
    int x, y;        // level ROI coordinates
    int camlevel;    // cam << 8 | level
    int resp_out;    // response << 24 | output index within the camera
};

// the rBRIEF test locations (include/orb_pattern_31.inc) as floats, one 16-byte load per test (rounds 1-3 kept them as signed bytes and
// k_describe converted four per test and lane: 8 instructions)
struct PatternF {
    float v[256][4];
    constexpr PatternF() : v() {
        constexpr signed char src[256][4] = {
#include "../../include/orb_pattern_31.inc"
        };
        for (int i = 0; i < 256; ++i) for (int k = 0; k < 4; ++k) v[i][k] = (float)src[i][k];
    }
};
// umax[v] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3} (ctor :455-470) as 16 nibbles of one constant (entry v
// at bits 4v..4v+3): a register operand instead of a memory load per use
constexpr unsigned long long UMAX_NIBBLES = 0x3689abcddeeeffffull;
// IC_Angle's disc as dword items (k_describe): item i = dword q = i % 9 of patch row r = i / 9 (columns u0 = 4 q - 18 .. u0 + 3 of row
// v = r - 15): the byte mask of the columns inside the disc (|u| <= umax[|v|]) and (u0, v), evaluated at compile time -- the kernel spent
// ~18 vector instructions per item on rebuilding them.
struct IcAngleItems {
    unsigned mask[31 * 9];
    int uv[31 * 9];   // u0 & 0xffff | v << 16
    constexpr IcAngleItems() : mask(), uv() {
        for (int i = 0; i < 31 * 9; ++i) {
            const int r = i / 9, q = i - r * 9, v = r - HALF_PATCH, u0 = 4 * q - 18;
            const int um = (int)((UMAX_NIBBLES >> (4 * (v < 0 ? -v : v))) & 15ull);
            unsigned m = 0;
            for (int b = 0; b < 4; ++b) { const int u = u0 + b; if (u >= -um && u <= um) m |= 0xffu << (8 * b); }
            mask[i] = m;
            uv[i] = (u0 & 0xffff) | (v * 65536);
        }
    }
};
// Both tables are evaluated at compile time on the HOST and live in the extractor's own device memory (orbx_create uploads them,
// k_describe takes the pointer).  (The library is compiled with -fno-slp-vectorize: see the Makefile for what packed f32 arithmetic on
// freshly loaded table registers did in the loopback rig.)
struct DescribeTables { IcAngleItems ic; alignas(16) PatternF pat; };

// block b of a 1-D grid of n -> position in XCD-major order: XCD x = b % 8 owns the contiguous positions
// [x * (n / 8) + min(x, n % 8), ...) and walks them in launch order (see k_fast_cells)
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int x = b & 7, i = b >> 3, per = n >> 3, rem = n & 7;
    return x * per + min(x, rem) + i;
}

// ------------------------------------------------------------------------------------------------ K0
// Level 0 of every camera whose image is already in HBM, in ONE launch (a 2-D copy per camera costs a launch each).
struct IngestArgs { const uint8_t* src[64]; int stride[64]; };

// Level 0 read IN PLACE (round 4, orbx_set_inplace_level0): on large rigs (the resize-chain pyramid) the copy of every camera's
// image into the pyramid buffer was a kernel of its own -- 16.6 MB in and 16.6 MB out per 8 x 1080p step, 10 us alone and 40 us
// next to the other chains -- although level 0 IS the image.  With the caller's promise that a device image stays valid and
// unchanged until the camera's next upload, the three kernels that read level 0 (the first resize, FAST, describe) take it from
// where it lies: a table of {pointer, pitch} per camera in device memory, written by a one-workgroup kernel per run (its
// arguments change every step, so it stays outside a captured chain exactly as k_ingest did); an entry with a null pointer
// means "level 0 is in the pyramid buffer".  Sources must be 4-byte aligned with a pitch that is a multiple of 4 (FAST stages
// its tile as aligned dwords); anything else is copied as before.  A pageable host image the library has staged in HBM itself
// (orbx_upload: the slot of the run that consumes it, rewritten two runs later at the earliest) is read in its staging slot alike.
struct L0Src { const uint8_t* ptr; int stride; int pad; };

__global__ void k_set_l0(IngestArgs A, L0Src* __restrict__ out, int n_cams) {
    const int c = threadIdx.x;
    if (c < n_cams) { L0Src s; s.ptr = A.src[c]; s.stride = A.stride[c]; s.pad = 0; out[c] = s; }
}

__global__ __launch_bounds__(256) void k_ingest(IngestArgs A, const LevelInfo* __restrict__ L, int max_levels,
                                                uint8_t* __restrict__ pyr, size_t cam_pitch) {
    const int cam = blockIdx.z;
    const uint8_t* src = A.src[cam];
    if (!src) return;
    const LevelInfo D = L[cam * max_levels];
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x16 = (blockIdx.x * 64 + threadIdx.x) * 16;
    if (y >= D.h || x16 >= D.w) return;
    const uint8_t* s = src + (size_t)y * A.stride[cam] + x16;
    uint8_t* d = pyr + cam * cam_pitch + D.pyr_off + (size_t)y * D.stride + x16;  // 16-byte aligned: pitch is a multiple of 64
    if (x16 + 16 <= D.w && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    } else {
        for (int k = 0; k < 16 && x16 + k < D.w; ++k) d[k] = s[k];
    }
}

// ------------------------------------------------------------------------------------------------ K1
__global__ __launch_bounds__(256) void k_resize(const LevelInfo* __restrict__ L, int max_levels, int level,
                                                uint8_t* __restrict__ pyr, size_t cam_pitch,
                                                const int2* __restrict__ xtab, const int4* __restrict__ ytab) {
    const int cam = blockIdx.z;
    const LevelInfo D = L[cam * max_levels + level];
    const LevelInfo S = L[cam * max_levels + level - 1];
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= D.h || x4 >= D.w) return;
    const uint8_t* src = pyr + cam * cam_pitch + S.pyr_off;
    uint8_t* dst = pyr + cam * cam_pitch + D.pyr_off;
    const int4 yt = ytab[D.ytab_off + y];  // {row0, row1, beta0, beta1}, rows already clipped to [0, sh-1]
    const uint8_t* s0 = src + (size_t)yt.x * S.stride;
    const uint8_t* s1 = src + (size_t)yt.y * S.stride;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x4 + k;
        if (x < D.w) {
            const int2 xt = xtab[D.xtab_off + x];  // {sx0 | sx1 << 16, alpha0 | alpha1 << 16}
            const int sx0 = xt.x & 0xffff, sx1 = (unsigned)xt.x >> 16;
            const int a0 = (short)(xt.y & 0xffff), a1 = xt.y >> 16;
            const int h0 = s0[sx0] * a0 + s0[sx1] * a1;
            const int h1 = s1[sx0] * a0 + s1[sx1] * a1;
            const int v = ((((yt.z * (h0 >> 4)) >> 16) + ((yt.w * (h1 >> 4)) >> 16) + 2) >> 2);
            out |= (uint32_t)(v & 0xff) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * D.stride + x4) = out;
}

// The whole pyramid of every camera in ONE launch (round 3).  The resize chain is a chain of dependencies only between pixels
// that lie over each other: a workgroup takes one T x T tile of level 0 and computes the part of EVERY level that hangs below
// it, level after level in LDS, storing what it owns.  Ownership: destination column x of level l belongs to the tile that
// owns the column of its left source tap sx0_l(x) on level l - 1 (rows alike with row0); sx0 and row0 are monotone, so on
// every level the tiles own disjoint runs that cover it.  To compute its run a tile needs a little more of the level above
// than it owns there -- the right / lower tap of its last pixel, and what THAT needs one level up: a halo that only grows to
// the right and downwards (~13 pixels on level 0 for 8 levels) and is recomputed by the neighbour that owns it.  Every pixel
// is the same integer function of the same source bytes as in k_resize (tables xt / yt), whoever computes it.
// Spans per (camera, level, tile column / tile row) come from the host: {r0, r1, n1} = owned [r0, r1), needed [r0, n1).
// Seven dependent launches of 3-25 us (5 with k_ingest folded in: A.src) become one; FAST, describe and the rest read the
// levels from HBM as before.
MORB_PHASE_DECL(g_ph_pyr);
#ifdef MORB_PHASE_CLOCKS
#define PPH(i) do { if (threadIdx.x == 0 && kx == 3 && ky == 3 && cam == 0) g_ph_pyr[i] = wall_clock64(); } while (0)
#else
#define PPH(i) do {} while (0)
#endif
struct PyrArgs { const uint8_t* src[64]; int stride[64]; short tx[64], ty[64], w[64], h[64]; };   // src NULL: level 0 is already in place
constexpr int PYR_L0_DW = 8;   // dwords of the level-0 block a thread holds: (64 + 16) / 4 x (64 + 16) / 256 rounded up
constexpr int PYR_HALO = 16;   // what a tile needs of level 0 beyond its own T x T pixels (13 for 8 levels at 1.2; checked by the host)
constexpr int PYR_MAX_LEVELS = 12;

// LDS: [x table entries of all levels (int2)] [y table entries of all levels (int4)] [the regions, level after level]
template <int NT>
__global__ __launch_bounds__(NT) void k_pyramid_tiled(PyrArgs A, const LevelInfo* __restrict__ L, int max_levels,
                                                       uint8_t* __restrict__ pyr, size_t cam_pitch, const int2* __restrict__ xtab,
                                                       const int4* __restrict__ ytab, const int4* __restrict__ spans_x,
                                                       const int4* __restrict__ spans_y, int tx_max, int ty_max, int tab_cap, int T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];
    __shared__ int4 s_sx[PYR_MAX_LEVELS], s_sy[PYR_MAX_LEVELS], s_lv[PYR_MAX_LEVELS];   // spans; {pyr_off, stride, xtab_off, ytab_off}
    __shared__ int s_ox[PYR_MAX_LEVELS + 1], s_oy[PYR_MAX_LEVELS + 1];   // first table entry of a level in tabx / taby
    __shared__ int s_nlev;
    // Tiles are dealt so that every XCD (workgroups reach the eight of them round robin in dispatch order, each has its own L2)
    // works through one contiguous run of them in (camera, tile row, tile column) order: the halo a tile shares with its right
    // and lower neighbour and the 128-byte lines their 32- or 64-byte row pieces share -- on the way in and on the way out --
    // then meet in ONE L2 (round 3 counters at 2 x 640x480: 4.6 MB fetched for 0.6 MB of level 0, 3.9 MB written for 1.9 MB of levels).
    const int dealt = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int per_cam = gridDim.x * gridDim.y;
    const int cam = dealt / per_cam, ky = (dealt - cam * per_cam) / (int)gridDim.x, kx = dealt - cam * per_cam - ky * (int)gridDim.x;
    if (kx >= A.tx[cam] || ky >= A.ty[cam]) return;
    PPH(0);
    const int tid = threadIdx.x;
    const LevelInfo* Lc = L + cam * max_levels;
    uint8_t* base = pyr + cam * cam_pitch;
    int4* taby = reinterpret_cast<int4*>(pyr_lds);                       // tab_cap entries
    int2* tabx = reinterpret_cast<int2*>(pyr_lds + (size_t)tab_cap * 16);  // tab_cap entries
    uint8_t* regions = pyr_lds + (size_t)tab_cap * 24;
    // ---- round trip 1, issued first: what the levels look like for this tile (one thread per level, independent loads: a load
    // per level inside the level loop would put a memory round trip on every level's critical path)
    int4 rt_x = make_int4(0, 0, 0, 0), rt_y = rt_x, rt_v = make_int4(0, -1, 0, 0);
    if (tid < PYR_MAX_LEVELS && tid < max_levels) {
        rt_x = spans_x[(size_t)(cam * max_levels + tid) * tx_max + kx]; rt_y = spans_y[(size_t)(cam * max_levels + tid) * ty_max + ky];
        const LevelInfo Lv = Lc[tid];
        rt_v = make_int4(Lv.pyr_off, Lv.w > 0 ? Lv.stride : -1, Lv.xtab_off, Lv.ytab_off);   // stride -1: no such level
    }
    // ---- level 0 first: the T x T pixels of the tile plus a fixed halo to the right and below, clipped to the image (what the
    // tile needs of level 0 is inside it; nothing here waits for the span tables), from the source image or from level 0 itself;
    // the tile's own pixels go to level 0 when they came from elsewhere.  These loads fly while the two table round trips run.
    const int W0 = A.w[cam], H0 = A.h[cam];
    int px0 = kx * T, py0 = ky * T;
    int pnh = min(T + PYR_HALO, H0 - py0), ppw = T + PYR_HALO;
    uint8_t* cur = regions;
    uint32_t l0v[PYR_L0_DW];
    int l0_ndw = 0; unsigned l0_inv = 0;
    {
        const uint8_t* src = A.src[cam];
        int sstride = A.stride[cam];
        const bool inplace = src == nullptr;
        const int l0_stride = (W0 + 63) & ~63;   // pitch of a pyramid level: align64(width); level 0 sits at offset 0 of the camera's block
        if (inplace) { src = base; sstride = l0_stride; }
        const int nw = min(T + PYR_HALO, W0 - px0), ow = min(T, W0 - px0), oh = min(T, H0 - py0);
        // (the tile's first column is a multiple of T, hence of 4: whole dwords while source rows are 4-aligned; the last
        // partial dword of a row byte by byte, so that nothing is read past the image)
        const bool al = ((reinterpret_cast<uintptr_t>(src) | (unsigned)sstride) & 3u) == 0;
        const int ndw = al ? nw >> 2 : 0, rem = nw - 4 * ndw;
        // (the dwords wait in registers: their LDS stores come behind the first table round trip, so the two overlap)
        l0_ndw = ndw; l0_inv = ndw > 0 ? ((1u << 20) + ndw - 1) / ndw : 0;   // i / ndw for i < 2^20 / ndw
#pragma unroll
        for (int u = 0; u < PYR_L0_DW; ++u) {
            const int i = tid + NT * u;
            l0v[u] = 0;
            if (i < ndw * pnh) {
                const int y = (int)(__umul24((unsigned)i, l0_inv) >> 20), k = i - (int)__umul24(y, ndw);
                l0v[u] = reinterpret_cast<const uint32_t*>(src + (size_t)(py0 + y) * sstride + px0)[k];
            }
        }
        if (rem > 0) {
            const unsigned inv = ((1u << 20) + rem - 1) / rem;
            for (int i = tid; i < rem * pnh; i += NT) {
                const int y = (int)(__umul24((unsigned)i, inv) >> 20), x = 4 * ndw + i - (int)__umul24(y, rem);
                cur[__umul24(y, ppw) + x] = src[(size_t)(py0 + y) * sstride + px0 + x];
            }
        }
        if (!inplace) {   // straight from the source to level 0 (the same bytes once more: L2 hits, independent of the LDS copy)
            const uint8_t* s2 = src;
            uint8_t* dst = base;
            const int odw = al ? ow >> 2 : 0, orem = ow - 4 * odw;
            if (odw > 0) {
                const unsigned inv = ((1u << 20) + odw - 1) / odw;
                for (int i = tid; i < odw * oh; i += NT) {
                    const int y = (int)(__umul24((unsigned)i, inv) >> 20), k = i - (int)__umul24(y, odw);
                    reinterpret_cast<uint32_t*>(dst + (size_t)(py0 + y) * l0_stride + px0)[k] = reinterpret_cast<const uint32_t*>(s2 + (size_t)(py0 + y) * sstride + px0)[k];
                }
            }
            if (orem > 0) {
                const unsigned inv = ((1u << 20) + orem - 1) / orem;
                for (int i = tid; i < orem * oh; i += NT) {
                    const int y = (int)(__umul24((unsigned)i, inv) >> 20), x = 4 * odw + i - (int)__umul24(y, orem);
                    dst[(size_t)(py0 + y) * l0_stride + px0 + x] = s2[(size_t)(py0 + y) * sstride + px0 + x];
                }
            }
        }
    }
    // (round trip 1 lands: the per-level records go to LDS)
    if (tid < PYR_MAX_LEVELS) { s_sx[tid] = rt_x; s_sy[tid] = rt_y; s_lv[tid] = rt_v; }
    __syncthreads();
    if (tid == 0) {
        int n = 1, ox = 0, oy = 0;
        s_ox[0] = s_ox[1] = 0; s_oy[0] = s_oy[1] = 0;
        while (n < PYR_MAX_LEVELS && n < max_levels && s_lv[n].y >= 0) {
            ox += max(s_sx[n].z - s_sx[n].x, 0); oy += max(s_sy[n].z - s_sy[n].x, 0);
            ++n;
            s_ox[n] = ox; s_oy[n] = oy;
        }
        s_nlev = n;
    }
    __syncthreads();
    const int nlev = s_nlev;
    PPH(1);
    // ---- round trip 2: everything else that does not depend on pixels -- the table entries of every level's needed columns
    // and rows (one entry per thread and step, the level found from the prefix counts: all requests in flight together) -- and
    // the needed region of level 0 (from the source image or from level 0 itself)
    {
        const int nx = s_ox[nlev], ny = s_oy[nlev];
        for (int e = tid; e < nx + ny; e += NT) {
            const bool isx = e < nx;
            const int k = isx ? e : e - nx;
            int l = 1;
            while (l + 1 < nlev && k >= (isx ? s_ox[l + 1] : s_oy[l + 1])) ++l;
            if (isx) tabx[k] = xtab[s_lv[l].z + s_sx[l].x + (k - s_ox[l])];
            else taby[k] = ytab[s_lv[l].w + s_sy[l].x + (k - s_oy[l])];
        }
    }
#pragma unroll
    for (int u = 0; u < PYR_L0_DW; ++u) {
        const int i = tid + NT * u;
        if (i < l0_ndw * pnh) { const int y = (int)(__umul24((unsigned)i, l0_inv) >> 20), k = i - (int)__umul24(y, l0_ndw); reinterpret_cast<uint32_t*>(cur + __umul24(y, ppw))[k] = l0v[u]; }
    }
    PPH(2);
    int4 sx, sy;
    __syncthreads();
    PPH(3);
    // ---- levels 1 ..: the needed region of level l from the needed region of level l - 1, both in LDS, over the flattened
    // region (every lane busy whatever the region's shape); two pixels per thread and step, their LDS reads issued together
    // (the regions of consecutive levels are different LDS ranges, which the compiler cannot know: a store behind the first
    // pixel would hold back the loads of the second)
    for (int l = 1; l < nlev; ++l) {
        const int4 lv = s_lv[l];
        sx = s_sx[l]; sy = s_sy[l];
        const int x0 = sx.x, y0 = sy.x, nw = max(sx.z - sx.x, 0), nh = max(sy.z - sy.x, 0), pw = (nw + 3) & ~3;
        const int ow = sx.y - sx.x, oh = sy.y - sy.x;
        uint8_t* nxt = cur + ppw * pnh;
        uint8_t* dst = base + lv.x + (size_t)y0 * lv.y + x0;
        const uint8_t* srcl = cur - py0 * ppw - px0;
        const int4* ty_l = taby + s_oy[l];
        const int2* tx_l = tabx + s_ox[l];
        if (nw > 0) {
            const unsigned inv = (unsigned)sx.w;   // 2^20 / nw, rounded up (host): i / nw for i < 2^20 / nw
            const int npx = nw * nh;
            for (int i0 = tid; i0 < npx; i0 += 2 * NT) {
                int vv[2], yy[2], xx[2];
                int p[2][4], a0[2], a1[2], b0[2], b1[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = min(i0 + NT * u, npx - 1);
                    yy[u] = (int)(__umul24((unsigned)i, inv) >> 20); xx[u] = i - (int)__umul24(yy[u], nw);   // (24-bit multiplies are full rate; 32-bit ones take four passes)
                    const int4 yt = ty_l[yy[u]];   // {row0, row1, beta0, beta1}
                    const int2 xt = tx_l[xx[u]];   // {sx0 | sx1 << 16, alpha0 | alpha1 << 16}
                    const uint8_t* s0 = srcl + __mul24(yt.x, ppw);
                    const uint8_t* s1 = srcl + __mul24(yt.y, ppw);
                    const int c0 = xt.x & 0xffff, c1 = (unsigned)xt.x >> 16;
                    p[u][0] = s0[c0]; p[u][1] = s0[c1]; p[u][2] = s1[c0]; p[u][3] = s1[c1];
                    a0[u] = (short)(xt.y & 0xffff); a1[u] = xt.y >> 16; b0[u] = yt.z; b1[u] = yt.w;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int h0 = __mul24(p[u][0], a0[u]) + __mul24(p[u][1], a1[u]);
                    const int h1 = __mul24(p[u][2], a0[u]) + __mul24(p[u][3], a1[u]);
                    vv[u] = ((((__mul24(b0[u], h0 >> 4)) >> 16) + ((__mul24(b1[u], h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (i0 + NT * u < npx) {
                        nxt[__mul24(yy[u], pw) + xx[u]] = (uint8_t)vv[u];
                        if (yy[u] < oh && xx[u] < ow) dst[__umul24(yy[u], lv.y) + xx[u]] = (uint8_t)vv[u];
                    }
                }
            }
        }
        __syncthreads();
        PPH(3 + l);
        cur = nxt; px0 = x0; py0 = y0; pnh = nh; ppw = pw;
    }
}

// ------------------------------------------------------------------------------------------------ K1, tiled, four pixels per lane
// Round 5: the tile scheme of k_pyramid_tiled (a workgroup takes a tile of ONE level -- the base -- and computes what hangs below it on
// the next levels in LDS, recomputing the halo its neighbours own) with round 3's four-pixels-per-lane arithmetic (four adjacent pixels per lane:
// a source row is three LDS dwords shifted to the first tap, one v_perm + v_dot2 per pixel and row, the x table ready-made per group
// of four columns).  k_pyramid_tiled spends ~70 vector instructions per pixel (one pixel per lane and step, byte reads), this form
// ~25 as the chain does -- so a LARGE rig's pyramid is two launches (levels 1..m below level 0, levels m+1.. below level m) instead of
// one launch per level, at the chain's cost per pixel plus the halo.
//   Ownership along x is by GROUPS of four destination columns: group g of level l belongs to the tile that owns, on level l - 1, the
// column of the group's first tap (rows: the row of the left tap, as k_pyramid_tiled).  First taps are monotone in g, so on every level
// the tiles own disjoint runs of whole groups that cover it; region starts are multiples of four on every level, so a dword of an LDS
// region is a dword of the level.  What a tile needs beyond what it owns grows to the right and downwards only.  Every pixel is the
// same integer function of the same source bytes as in k_resize, whoever computes it.
//   Spans per (camera, level, tile column / row) from the host (TilePlan): x {own0, own1, need1, 2^20 / needed groups} in pixels (all
// multiples of four below the base), y {own0, own1, need1, 0}.
struct Tile4Args {
    const LevelInfo* L; uint8_t* pyr; size_t cam_pitch; const int4* xgrp; const int4* ytab; const int4* sx; const int4* sy; const L0Src* l0;
    int max_levels, base, last, tw, th, halo_x, halo_y, tx_max, ty_max, gcap, rcap;
    short tx[64], ty[64];
};
constexpr int T4_MAX_DW = 16;   // dwords of the base block a thread holds while the tables are on their way: (tw + halo) / 4 x (th + halo) <= 16 x threads (host)

template <int NT>
__global__ __launch_bounds__(NT) void k_pyramid_tiled4(Tile4Args A) {
    extern __shared__ __attribute__((aligned(16))) uint8_t t4_lds[];
    __shared__ int4 s_sx[PYR_MAX_LEVELS], s_sy[PYR_MAX_LEVELS], s_lv[PYR_MAX_LEVELS];   // spans; {pyr_off, stride, xgrp_off, ytab_off} (index: level - base)
    __shared__ int s_og[PYR_MAX_LEVELS + 1], s_or[PYR_MAX_LEVELS + 1];                   // first group / row entry of a level in the LDS tables
    const int dealt = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int per_cam = gridDim.x * gridDim.y;
    const int cam = dealt / per_cam, ky = (dealt - cam * per_cam) / (int)gridDim.x, kx = dealt - cam * per_cam - ky * (int)gridDim.x;
    if (kx >= A.tx[cam] || ky >= A.ty[cam]) return;
    const int tid = threadIdx.x;
    const LevelInfo* Lc = A.L + cam * A.max_levels;
    uint8_t* base_ptr = A.pyr + cam * A.cam_pitch;
    const int nrec = A.last - A.base + 1;   // records of levels base .. last
    int4* gtab = reinterpret_cast<int4*>(t4_lds);                       // 3 x gcap entries
    int4* rtab = gtab + 3 * (size_t)A.gcap;                             // rcap entries
    uint8_t* regions = reinterpret_cast<uint8_t*>(rtab + A.rcap);
    // ---- round trip 1: the span records of this tile's levels (one thread per level)
    int4 rt_x = make_int4(0, 0, 0, 0), rt_y = rt_x, rt_v = rt_x;
    if (tid < nrec) {
        const int l = A.base + tid;
        rt_x = A.sx[(size_t)(cam * A.max_levels + l) * A.tx_max + kx]; rt_y = A.sy[(size_t)(cam * A.max_levels + l) * A.ty_max + ky];
        const LevelInfo Lv = Lc[l];
        rt_v = make_int4(Lv.pyr_off, Lv.stride, Lv.xgrp_off, Lv.ytab_off);
    }
    // ---- the base block: tw x th pixels of the tile plus the halo to the right and below, clipped to the level; whole dwords where the
    // source allows it (always inside the pyramid buffer: rows start on 64-byte boundaries, tile columns are multiples of four)
    const LevelInfo Lb = Lc[A.base];
    const uint8_t* src = base_ptr + Lb.pyr_off;
    int sstride = Lb.stride;
    if (A.base == 0 && A.l0) { const L0Src s0 = A.l0[cam]; if (s0.ptr) { src = s0.ptr; sstride = s0.stride; } }
    int px0 = kx * A.tw, py0 = ky * A.th;
    int ppw = A.tw + A.halo_x, pnh = min(A.th + A.halo_y, Lb.h - py0);
    uint8_t* cur = regions;
    uint32_t held[T4_MAX_DW];
    const int nw = min(A.tw + A.halo_x, Lb.w - px0);
    const bool al = ((reinterpret_cast<uintptr_t>(src) | (unsigned)sstride) & 3u) == 0;
    const int ndw = al ? nw >> 2 : 0, rem = nw - 4 * ndw;
    const unsigned inv_ndw = ndw > 0 ? ((1u << 20) + ndw - 1) / ndw : 0;   // i / ndw for i < 2^20 / ndw
#pragma unroll
    for (int u = 0; u < T4_MAX_DW; ++u) {
        const int i = tid + NT * u;
        held[u] = 0;
        if (i < ndw * pnh) {
            const int y = (int)(__umul24((unsigned)i, inv_ndw) >> 20), k = i - (int)__umul24(y, ndw);
            held[u] = reinterpret_cast<const uint32_t*>(src + (size_t)(py0 + y) * sstride + px0)[k];
        }
    }
    if (rem > 0) {
        const unsigned inv = ((1u << 20) + rem - 1) / rem;
        for (int i = tid; i < rem * pnh; i += NT) {
            const int y = (int)(__umul24((unsigned)i, inv) >> 20), x = 4 * ndw + i - (int)__umul24(y, rem);
            cur[__umul24(y, ppw) + x] = src[(size_t)(py0 + y) * sstride + px0 + x];
        }
    }
    if (tid < nrec) { s_sx[tid] = rt_x; s_sy[tid] = rt_y; s_lv[tid] = rt_v; }
    __syncthreads();
    if (tid == 0) {
        int og = 0, orw = 0;
        s_og[0] = 0; s_or[0] = 0;
        for (int r = 1; r < nrec; ++r) {
            s_og[r] = og; s_or[r] = orw;
            og += max(s_sx[r].z - s_sx[r].x, 0) >> 2; orw += max(s_sy[r].z - s_sy[r].x, 0);
        }
        s_og[nrec] = og; s_or[nrec] = orw;
    }
    __syncthreads();
    // ---- round trip 2: the table entries of every level's needed groups (3 x int4 each) and rows (int4), all requests in flight together
    {
        const int ng3 = 3 * s_og[nrec], nr = s_or[nrec];
        for (int e = tid; e < ng3 + nr; e += NT) {
            const bool isg = e < ng3;
            const int k = isg ? e / 3 : e - ng3;
            int r = 1;
            while (r + 1 < nrec && k >= (isg ? s_og[r + 1] : s_or[r + 1])) ++r;
            if (isg) gtab[e] = A.xgrp[3 * (size_t)(s_lv[r].z + (s_sx[r].x >> 2) + (k - s_og[r])) + (e - 3 * k)];
            else rtab[k] = A.ytab[s_lv[r].w + s_sy[r].x + (k - s_or[r])];
        }
    }
#pragma unroll
    for (int u = 0; u < T4_MAX_DW; ++u) {
        const int i = tid + NT * u;
        if (i < ndw * pnh) { const int y = (int)(__umul24((unsigned)i, inv_ndw) >> 20), k = i - (int)__umul24(y, ndw); reinterpret_cast<uint32_t*>(cur + __umul24(y, ppw))[k] = held[u]; }
    }
    __syncthreads();
    // ---- levels base + 1 .. last: the needed region of a level from the needed region of the level above it, both in LDS
    using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
    for (int r = 1; r < nrec; ++r) {
        const int4 lv = s_lv[r], sx = s_sx[r], sy = s_sy[r];
        const int x0 = sx.x, y0 = sy.x, ngn = max(sx.z - sx.x, 0) >> 2, nh = max(sy.z - sy.x, 0), pw = 4 * ngn;
        const int ngo = (sx.y - sx.x) >> 2, oh = sy.y - sy.x;
        if (ngn == 0 || nh == 0) break;   // (nothing of this level hangs below the tile: nothing of the next ones does either)
        uint8_t* nxt = cur + ppw * pnh;
        uint8_t* dst = base_ptr + lv.x + (size_t)y0 * lv.y + x0;
        const int4* G_l = gtab + 3 * s_og[r];
        const int4* R_l = rtab + s_or[r];
        // A lane keeps ONE group of four columns and walks down its rows (NT / ngn rows per pass; the lanes beyond the last whole row
        // idle): the group's table entry -- 36 bytes -- is read once per level instead of once per four pixels (it was two thirds of
        // the loop's LDS traffic), and the index split per task is gone.
        const unsigned inv = (unsigned)sx.w;   // 2^20 / ngn, rounded up: t / ngn for t * ngn < 2^20
        const int ty = (int)(__umul24((unsigned)tid, inv) >> 20), g = tid - (int)__umul24(ty, ngn);
        const int rpp = (int)(__umul24((unsigned)NT, inv) >> 20);   // rows per pass
        if (ty < rpp) {
            const int4 g0 = G_l[3 * g], g1 = G_l[3 * g + 1];
            const int g2 = G_l[3 * g + 2].x;
            const int c = g0.x - px0, sh = c & 3;
            const uint32_t ps[4] = {(uint32_t)g0.y, (uint32_t)g0.z, (uint32_t)g0.w, (uint32_t)g1.x};
            const uint32_t alp[4] = {(uint32_t)g1.y, (uint32_t)g1.z, (uint32_t)g1.w, (uint32_t)g2};
            const uint32_t* colw = reinterpret_cast<const uint32_t*>(cur) + (c >> 2);
            const bool own_g = g < ngo;
            for (int y = ty; y < nh; y += rpp) {
                const int4 yt = R_l[y];
                uint32_t h[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = (q ? yt.y : yt.x) - py0;
                    const uint32_t* w = colw + (__umul24(row, ppw) >> 2);   // (region pitches are multiples of four)
                    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        h[q][j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, ps[j])), __builtin_bit_cast(u16x2, alp[j]), 0u, false);
                }
                uint32_t out = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = ((((__umul24((uint32_t)yt.z, h[0][j] >> 4)) >> 16) + ((__umul24((uint32_t)yt.w, h[1][j] >> 4)) >> 16) + 2) >> 2) & 0xff;
                    out |= v << (8 * j);
                }
                reinterpret_cast<uint32_t*>(nxt + __umul24(y, pw))[g] = out;
                if (own_g && y < oh) reinterpret_cast<uint32_t*>(dst + __umul24(y, lv.y))[g] = out;   // (columns past w: padding of the 64-byte row pitch, as k_resize)
            }
        }
        __syncthreads();
        cur = nxt; px0 = x0; py0 = y0; pnh = nh; ppw = pw;
    }
}

// ------------------------------------------------------------------------------------------------ K2 + K3
// FAST-9/16 corner score of the pixel at t: max over the 16 arcs of 9 contiguous ring pixels of the minimum
// |centre - ring| with a common sign, minus 1 (== cornerScore<16>, threshold-independent for corners; App. A-2).
__device__ __forceinline__ int fast_score_9_16(const uint8_t* t) {
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * TILE_PITCH];          d[1] = v - t[3 * TILE_PITCH + 1];
    d[2] = v - t[2 * TILE_PITCH + 2];      d[3] = v - t[1 * TILE_PITCH + 3];
    d[4] = v - t[3];                       d[5] = v - t[-1 * TILE_PITCH + 3];
    d[6] = v - t[-2 * TILE_PITCH + 2];     d[7] = v - t[-3 * TILE_PITCH + 1];
    d[8] = v - t[-3 * TILE_PITCH];         d[9] = v - t[-3 * TILE_PITCH - 1];
    d[10] = v - t[-2 * TILE_PITCH - 2];    d[11] = v - t[-1 * TILE_PITCH - 3];
    d[12] = v - t[-3];                     d[13] = v - t[1 * TILE_PITCH - 3];
    d[14] = v - t[2 * TILE_PITCH - 2];     d[15] = v - t[3 * TILE_PITCH - 1];
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
    int sp = -256, sm = 256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);  // min of d[k..k+8]
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);  // max of d[k..k+8]
        sp = max(sp, lo9);
        sm = min(sm, hi9);
    }
    return max(sp, -sm) - 1;
}

// Necessary condition for fast_score_9_16(t) >= min_th, from 4 antipodal ring pairs: every 9-arc of the 16-ring contains
// one pixel of each antipodal pair, so a bright arc (all d >= T, T = min_th + 1) needs d >= T on one side of EVERY pair,
// and likewise a dark arc.  Pixels failing both keep the score 0 they would get anyway (the cv::FAST high-speed test).
__device__ __forceinline__ bool fast_may_be_corner(const uint8_t* t, int T) {
    const int v = t[0];
    const int d0 = v - t[3 * TILE_PITCH], d8 = v - t[-3 * TILE_PITCH];
    const int d4 = v - t[3], d12 = v - t[-3];
    bool bright = (d0 >= T || d8 >= T) && (d4 >= T || d12 >= T);
    bool dark = (d0 <= -T || d8 <= -T) && (d4 <= -T || d12 <= -T);
    if (!(bright || dark)) return false;
    const int d2 = v - t[2 * TILE_PITCH + 2], d10 = v - t[-2 * TILE_PITCH - 2];
    const int d6 = v - t[-2 * TILE_PITCH + 2], d14 = v - t[2 * TILE_PITCH - 2];
    bright = bright && (d2 >= T || d10 >= T) && (d6 >= T || d14 >= T);
    dark = dark && (d2 <= -T || d10 <= -T) && (d6 <= -T || d14 <= -T);
    return bright || dark;
}

// The same test for FOUR horizontally adjacent pixels on packed 16-bit lanes (round 3).  `t32` points at the dword of the
// centre row that holds tile columns [4 gx, 4 gx + 4); the four pixels sit at columns 4 gx + 3 .. 4 gx + 6 (the tile's origin is
// three pixels left of the scored rectangle and dword-aligned in LDS), so every operand is a compile-time byte alignment of two
// neighbouring dwords.  With A = max over the four antipodal pairs of min(r_k, r_k+8) and B = min over the pairs of
// max(r_k, r_k+8), a bright arc needs C - A >= T and a dark arc B - C >= T (T = min_th + 1).
//
// Round 5, second form: NO UNPACKING.  An unsigned 16-bit minimum / maximum orders by the HIGH byte first, so the high byte of any
// min / max tree over 16-bit lanes is that tree over the high bytes, whatever the low bytes hold.  The four bytes starting at a ring
// pixel of p0 therefore ARE the packed operand of the odd pixels p1 and p3 (bytes 1 and 3 = the high bytes of the two lanes, bytes
// 0 and 2 = junk), and the four bytes starting one column earlier that of the even pixels p0 and p2 -- one v_alignbyte per operand
// (none where the start is dword-aligned) where the round-3 form spent an alignbyte and two v_perm.  The junk is kept out of the
// decision by the centre's low byte: with A' = A << 8 | a (a = junk, 0..255) and C0 = C << 8,  sat(C0 - A') = max(0, (C - A) * 256 - a)
// is >= K = (T - 1) * 256 + 1 exactly when C - A >= T; with C1 = C << 8 | 255,  sat(B' - C1) = max(0, (B - C) * 256 + b - 255) is
// >= K exactly when B - C >= T.  A saturating add of 0x8000 - K then puts "passes" into bit 15 of the lane (0 <= min_th <= 127: K
// fits; other thresholds take the round-3 form below).  Returns the pass bits where they fall: p1 -> 15, p3 -> 31, p0 -> 14, p2 -> 30.
using fq_s16x2 = short __attribute__((ext_vector_type(2)));
using fq_u16x2 = unsigned short __attribute__((ext_vector_type(2)));
constexpr int FQ_BIT[8] = {14, 15, 30, 31, 12, 13, 28, 29};   // pixel j of a lane's eight -> its bit (the second four: two to the right)
__device__ __forceinline__ uint32_t fq_scramble(unsigned lin) {   // bit j -> bit FQ_BIT[j]
    return ((lin & 3u) << 14) | (((lin >> 2) & 3u) << 30) | (((lin >> 4) & 3u) << 12) | (((lin >> 6) & 3u) << 28);
}
__device__ __forceinline__ uint32_t fq_tree(const uint32_t (&ring)[8], uint32_t c, uint32_t kadd2) {
    auto U = [](uint32_t w) { return __builtin_bit_cast(fq_u16x2, w); };
    const fq_u16x2 A = __builtin_elementwise_max(
        __builtin_elementwise_max(__builtin_elementwise_min(U(ring[0]), U(ring[1])), __builtin_elementwise_min(U(ring[2]), U(ring[3]))),
        __builtin_elementwise_max(__builtin_elementwise_min(U(ring[4]), U(ring[5])), __builtin_elementwise_min(U(ring[6]), U(ring[7]))));
    const fq_u16x2 B = __builtin_elementwise_min(
        __builtin_elementwise_min(__builtin_elementwise_max(U(ring[0]), U(ring[1])), __builtin_elementwise_max(U(ring[2]), U(ring[3]))),
        __builtin_elementwise_min(__builtin_elementwise_max(U(ring[4]), U(ring[5])), __builtin_elementwise_max(U(ring[6]), U(ring[7]))));
    const fq_u16x2 v = __builtin_elementwise_max(__builtin_elementwise_sub_sat(U(c & 0xff00ff00u), A),
                                                 __builtin_elementwise_sub_sat(B, U(c | 0x00ff00ffu)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(v, U(kadd2)));
}
__device__ __forceinline__ uint32_t fast_may_be_corner_x4(const uint32_t* t32, uint32_t kadd2) {
    constexpr int P = TILE_PITCH / 4;
    auto ab = [](uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); };
    const uint32_t w0 = t32[0], w1 = t32[1], w2 = t32[2];
    const uint32_t u0 = t32[3 * P], u1 = t32[3 * P + 1];            // ring 0: (+3, 0)
    const uint32_t v0 = t32[-3 * P], v1 = t32[-3 * P + 1];          // ring 8: (-3, 0)
    const uint32_t a0 = t32[2 * P], a1 = t32[2 * P + 1], a2 = t32[2 * P + 2];      // ring 2: (+2, +2), ring 14: (+2, -2)
    const uint32_t b0 = t32[-2 * P], b1 = t32[-2 * P + 1], b2 = t32[-2 * P + 2];   // ring 6: (-2, +2), ring 10: (-2, -2)
    // odd pixels (p1, p3): the four bytes from p0's ring pixel on
    const uint32_t odd[8] = {ab(u1, u0, 3), ab(v1, v0, 3),        // 0, 8
                             ab(w2, w1, 2), w0,                   // 4 (0, +3), 12 (0, -3)
                             ab(a2, a1, 1), ab(b1, b0, 1),        // 2, 10
                             ab(b2, b1, 1), ab(a1, a0, 1)};       // 6, 14
    // even pixels (p0, p2): the same, one column earlier (ring 12 would start in the dword before w0: a shift puts the same
    // two bytes into the lanes' high halves)
    const uint32_t even[8] = {ab(u1, u0, 2), ab(v1, v0, 2), ab(w2, w1, 1), w0 << 8, a1, b0, b1, a0};
    const uint32_t e_odd = fq_tree(odd, ab(w1, w0, 3), kadd2), e_even = fq_tree(even, ab(w1, w0, 2), kadd2);
    return (e_odd & 0x80008000u) | ((e_even >> 1) & 0x40004000u);
}
// The round-3 form of the same test (pixels unpacked to 16-bit lanes, signed arithmetic): any threshold.  Same bit positions.
__device__ __forceinline__ uint32_t fast_may_be_corner_x4_wide(const uint32_t* t32, int T) {
    constexpr int P = TILE_PITCH / 4;
    auto ab = [](uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); };
    const uint32_t w0 = t32[0], w1 = t32[1], w2 = t32[2];
    const uint32_t u0 = t32[3 * P], u1 = t32[3 * P + 1];
    const uint32_t v0 = t32[-3 * P], v1 = t32[-3 * P + 1];
    const uint32_t a0 = t32[2 * P], a1 = t32[2 * P + 1], a2 = t32[2 * P + 2];
    const uint32_t b0 = t32[-2 * P], b1 = t32[-2 * P + 1], b2 = t32[-2 * P + 2];
    const uint32_t ring[8] = {ab(u1, u0, 3), ab(v1, v0, 3), ab(w2, w1, 2), w0, ab(a2, a1, 1), ab(b1, b0, 1), ab(b2, b1, 1), ab(a1, a0, 1)};
    const uint32_t c = ab(w1, w0, 3);
    const short Ts = (short)T;
    const fq_s16x2 T2 = {Ts, Ts};
    unsigned pass = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t sel = h ? 0x0c030c02u : 0x0c010c00u;       // bytes 2h, 2h + 1 -> the low bytes of the two 16-bit lanes
        const fq_s16x2 C = __builtin_bit_cast(fq_s16x2, __builtin_amdgcn_perm(0u, c, sel));
        fq_s16x2 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __builtin_bit_cast(fq_s16x2, __builtin_amdgcn_perm(0u, ring[k], sel));
        const fq_s16x2 A = __builtin_elementwise_max(
            __builtin_elementwise_max(__builtin_elementwise_min(r[0], r[1]), __builtin_elementwise_min(r[2], r[3])),
            __builtin_elementwise_max(__builtin_elementwise_min(r[4], r[5]), __builtin_elementwise_min(r[6], r[7])));
        const fq_s16x2 B = __builtin_elementwise_min(
            __builtin_elementwise_min(__builtin_elementwise_max(r[0], r[1]), __builtin_elementwise_max(r[2], r[3])),
            __builtin_elementwise_min(__builtin_elementwise_max(r[4], r[5]), __builtin_elementwise_max(r[6], r[7])));
        const fq_s16x2 m = __builtin_elementwise_max(C - A, B - C) - T2;   // >= 0 where the pixel passes
        const uint32_t mb = __builtin_bit_cast(uint32_t, m);
        pass |= ((mb & 0x8000u) ? 0u : 1u) << (2 * h);
        pass |= ((mb & 0x80000000u) ? 0u : 2u) << (2 * h);
    }
    return fq_scramble(pass);
}

// fast_score_9_16 for TWO pixels at once on packed 16-bit lanes (round 4).  Round 5: on the ring values themselves -- the minimum
// of the differences v - r over an arc is v minus the arc's MAXIMUM, so S+ = v - P with P = min over the arcs of their maximum,
// S- = Q - v with Q = max over the arcs of their minimum -- and with THREE-input extrema: gfx950 has packed three-input minimum /
// maximum only for f16 (v_pk_minimum3_f16 / v_pk_maximum3_f16), and a 16-bit lane holding 0..255 read as f16 is a non-negative
// (denormal) number that orders exactly as the integer does (f16 denormals are preserved: .amdhsa_float_denorm_mode_16_64 3, which
// tests/test_isa_guard.py holds), so the nine-element arc extrema are two levels of three (3 x 3) instead of log steps:
// 2 x (16 + 16 + 8) packed instructions per pair of pixels where the two-input form took 2 x 80.  The selected values are the
// operands' own bit patterns; the last subtractions are integer.  Returns score(t0) | score(t1) << 16.
using fq_h16x2 = _Float16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fq_h16x2 fq_min3(fq_h16x2 a, fq_h16x2 b, fq_h16x2 c) { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); }
__device__ __forceinline__ fq_h16x2 fq_max3(fq_h16x2 a, fq_h16x2 b, fq_h16x2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
__device__ __forceinline__ uint32_t fast_score_9_16_x2(const uint8_t* t0, const uint8_t* t1) {
    auto ring2 = [&](int off) { return (uint32_t)t0[off] | ((uint32_t)t1[off] << 16); };
    const fq_s16x2 v = __builtin_bit_cast(fq_s16x2, ring2(0));
    fq_h16x2 r[16];
    constexpr int R[16] = {3 * TILE_PITCH, 3 * TILE_PITCH + 1, 2 * TILE_PITCH + 2, 1 * TILE_PITCH + 3, 3, -1 * TILE_PITCH + 3,
                           -2 * TILE_PITCH + 2, -3 * TILE_PITCH + 1, -3 * TILE_PITCH, -3 * TILE_PITCH - 1, -2 * TILE_PITCH - 2,
                           -1 * TILE_PITCH - 3, -3, 1 * TILE_PITCH - 3, 2 * TILE_PITCH - 2, 3 * TILE_PITCH - 1};
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = __builtin_bit_cast(fq_h16x2, ring2(R[k]));
    fq_h16x2 lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo3[k] = fq_min3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]); hi3[k] = fq_max3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]); }
    fq_h16x2 amin[16], amax[16];   // extrema of r[k .. k+8]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        amin[k] = fq_min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        amax[k] = fq_max3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    }
    fq_h16x2 P = __builtin_elementwise_minimum(amax[0], amax[1]), Q = __builtin_elementwise_maximum(amin[0], amin[1]);
#pragma unroll
    for (int k = 2; k < 16; k += 2) { P = fq_min3(P, amax[k], amax[k + 1]); Q = fq_max3(Q, amin[k], amin[k + 1]); }
    const fq_s16x2 one = {1, 1};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v - __builtin_bit_cast(fq_s16x2, P), __builtin_bit_cast(fq_s16x2, Q) - v) - one);
}

// 2^20 / d rounded up, for d in 1 .. 127: i / d == (i * inv) >> 20 whenever i * d < 2^20 (the index splits of the cell kernel).
// A table instead of a scalar integer division per workgroup (~40 scalar instructions each, four of them at the head of
// every cell: 185 of the kernel's 400 scalar instructions).
__device__ const unsigned d_inv20[128] = {
#define I20(d) ((d) ? ((1u << 20) + (d) - 1) / ((d) ? (d) : 1) : 0u)
#define I20_8(b) I20(b), I20(b + 1), I20(b + 2), I20(b + 3), I20(b + 4), I20(b + 5), I20(b + 6), I20(b + 7)
    I20_8(0), I20_8(8), I20_8(16), I20_8(24), I20_8(32), I20_8(40), I20_8(48), I20_8(56),
    I20_8(64), I20_8(72), I20_8(80), I20_8(88), I20_8(96), I20_8(104), I20_8(112), I20_8(120)
#undef I20_8
#undef I20
};

// k_fast_cells<NT, GW>: NT threads per cell, GW (4 or 8) adjacent pixels per lane in the quick test.  <128, 8> is the
// throughput form (fewest instructions per cell: rigs whose cells fill the chip several times over); <256, 4> the latency
// form (a cell's phases spread over four waves: 10.2 against 11.7 us for 2 x 640x480, where the launch is one round of cells).
inline size_t fast_cells_lds(int cell_h_max, int cell_px_max) {   // the kernel's carve-up, in bytes
    return (size_t)(cell_h_max + 6) * TILE_PITCH + (size_t)(cell_h_max + 2) * SCORE_PITCH + (size_t)((cell_px_max + 1) & ~1) * 2 +
           (size_t)((cell_px_max + 63) / 64 * 2) * 2 * 4;
}
template <int FAST_NT, int GW>
__global__ __launch_bounds__(FAST_NT) void k_fast_cells(const LevelInfo* __restrict__ L, const int2* __restrict__ cell_map,
                                                    const uint8_t* __restrict__ pyr, size_t cam_pitch, int max_levels,
                                                    int* __restrict__ cell_cnt, uint32_t* __restrict__ cell_items,
                                                    int cell_h_max, int cell_px_max, const L0Src* __restrict__ l0) {
    // LDS sized at launch for the largest cell of the rig (fast_cells_lds: the reference's 30-pixel cells take 7 KB, so that
    // sixteen two-wave cells fit a CU; the 64 x 64 worst case this build supports would take 19 KB):
    // tile rows (pitch TILE_PITCH) | score rows (pitch SCORE_PITCH) | survivor list | the two bitmaps
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
    uint8_t* tile_raw = fast_lds;
    uint8_t* score = tile_raw + (cell_h_max + 6) * TILE_PITCH;
    unsigned short* s_surv = reinterpret_cast<unsigned short*>(score + (cell_h_max + 2) * SCORE_PITCH);  // pixels that pass the quick test
    // bit p of s_max: pixel p (row-major inside the scored rectangle) is a strict local maximum with a score >= minTh;
    // of s_ini: ... and its score reaches iniTh.  s_pref[j]: set bits of the chosen bitmap in front of its 64-bit word j.
    const int bm_words = (cell_px_max + 63) / 64 * 2;
    unsigned int* s_max = reinterpret_cast<unsigned int*>(s_surv + ((cell_px_max + 1) & ~1));
    unsigned int* s_ini = s_max + bm_words;
    __shared__ int s_pref[64];
    __shared__ int s_any, s_nsurv;

    // Workgroups go to the 8 XCDs round robin, each XCD has an L2 of its own, and neighbouring cells share the 128-byte lines
    // of their halos: workgroup b takes cell (b % 8) * (cells / 8) + b / 8, so that an XCD works through ONE contiguous run of
    // cells (a band of rows of a level) instead of every eighth cell of all of them (FETCH_SIZE of this kernel: 4.4 x the
    // pyramid bytes before -- every line was pulled into several L2s).  Affinity only: any placement computes the same.
    const int cell = xcd_contiguous(blockIdx.x, gridDim.x);
    const int2 cm = cell_map[cell];  // {cam * max_levels + level | cell column << 12 | cell row << 22, local cell index}
    const int blk = cm.x & 0xfff, cj = (cm.x >> 12) & 0x3ff, ci = (int)((unsigned)cm.x >> 22);
    const LevelInfo Lv = L[blk];
    const int cam = blk / max_levels;
    // scored rectangle of this cell in level coordinates (App. A-3)
    const int x0 = EDGE_THRESHOLD + cj * Lv.w_cell, y0 = EDGE_THRESHOLD + ci * Lv.h_cell;
    const int cw = min(x0 + Lv.w_cell, Lv.w - EDGE_THRESHOLD) - x0;
    const int ch = min(y0 + Lv.h_cell, Lv.h - EDGE_THRESHOLD) - y0;
    const int tid = threadIdx.x;
    if (cw <= 0 || ch <= 0) {
        if (tid == 0) cell_cnt[cell] = 0;
        return;
    }
    const uint8_t* img = pyr + cam * cam_pitch + Lv.pyr_off;
    int img_stride = Lv.stride;
    if (l0 && blk == cam * max_levels) {   // level 0 may lie in the caller's buffer (k_set_l0)
        const L0Src s0 = l0[cam];
        if (s0.ptr) { img = s0.ptr; img_stride = s0.stride; }
    }
    const int tw = cw + 6, th = ch + 6;
    // p / cw (and i / ndw below) for p < 70 * 70 by multiplication: exact because p * divisor < 2^20 (divisors <= 70)
    const unsigned inv_cw = d_inv20[cw];

    // stage the tile (origin x0-3, y0-3; always inside the level) dword by dword: the rows of a level start on 64-byte
    // boundaries, so the global dwords that cover columns [x0-3, x0-3+tw) are fetched whole (byte loads cost an address
    // computation, a load and an LDS store per BYTE: a ninth of this kernel's instructions) and shifted by the origin's
    // misalignment `sh` on the way in, so that the tile's origin is dword-aligned in LDS (the packed quick test relies on it).
    // Reads stay inside the row: the last dword fetched ends before column w - 5.
    const int sh = (x0 - 3) & 3, ndw = (tw + 3) >> 2;
    const unsigned inv_ndw = d_inv20[ndw];
    {
        const uint8_t* src = img + (size_t)(y0 - 3) * img_stride + (x0 - 3 - sh);
        for (int i = tid; i < ndw * th; i += FAST_NT) {
            const int ty = (int)(__umul24((unsigned)i, inv_ndw) >> 20), k = i - (int)__umul24(ty, ndw);   // (24-bit multiplies: full rate, four times the 32-bit one's)
            const uint32_t* g = reinterpret_cast<const uint32_t*>(src + (size_t)ty * img_stride + 4 * k);
            const uint32_t lo = g[0], hi = sh ? g[1] : 0u;
            reinterpret_cast<uint32_t*>(tile_raw)[ty * (TILE_PITCH / 4) + k] = __builtin_amdgcn_alignbyte(hi, lo, sh);
        }
    }
    const uint8_t* tile = tile_raw;
    for (int i = tid; i < (ch + 2) * (SCORE_PITCH / 4); i += FAST_NT) reinterpret_cast<uint32_t*>(score)[i] = 0;
    static_assert(CELL_MAX * CELL_MAX / 32 <= FAST_NT && (GW == 4 || GW == 8), "one bitmap word per thread; four or eight pixels per lane");
    if (tid < bm_words) { s_max[tid] = 0; s_ini[tid] = 0; }
    if (tid == 0) { s_any = 0; s_nsurv = 0; }
    __syncthreads();

    // pass 1: quick test on every pixel, EIGHT adjacent pixels per lane (two fast_may_be_corner_x4 on neighbouring dwords, which
    // share their loads); survivors compacted (any order) so that pass 2 runs the full score on dense lanes.  What a lane spends
    // around the test -- index split, wave scan, the slot from the block counter -- is per lane, not per pixel: with eight pixels
    // per lane and two waves per cell a 30 x 30 cell costs ~440 vector instructions here instead of ~610 (four pixels, four waves).
    const int lane = tid & 63;
    const int gpr = (cw + GW - 1) / GW, ngroups = gpr * ch;          // groups of GW per row
    const unsigned inv_gpr = d_inv20[gpr];                           // (g / gpr exact: g * gpr < 2^20)
    // (the quick test's constants: the byte form for 0 <= min_th <= 127, the unpacked form otherwise; the pass bits of a lane's
    // pixels sit at FQ_BIT[j], and what the last group of a row has beyond the rectangle is masked there)
    const bool narrow = Lv.min_th >= 0 && Lv.min_th <= 127;
    const uint32_t kadd2 = (uint32_t)(0x8000 - (Lv.min_th * 256 + 1)) * 0x10001u;
    const uint32_t tail_mask = fq_scramble((1u << (cw - GW * (gpr - 1))) - 1u);
    for (int base = 0; base < ngroups; base += FAST_NT) {
        const int g = base + tid;
        uint32_t m8 = 0;
        int p0 = 0;
        if (g < ngroups) {
            const int py = (int)(__umul24((unsigned)g, inv_gpr) >> 20), gx = g - (int)__umul24(py, gpr);
            const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile_raw) + (py + 3) * (TILE_PITCH / 4) + (GW / 4) * gx;
            const int left = cw - GW * gx;                           // pixels of this group inside the scored rectangle
            if (narrow) {
                m8 = fast_may_be_corner_x4(t32, kadd2);
                if (GW == 8 && left > 4) m8 |= fast_may_be_corner_x4(t32 + 1, kadd2) >> 2;   // (the second dword's reads stay inside the staged tile row then)
            } else {
                m8 = fast_may_be_corner_x4_wide(t32, Lv.min_th + 1);
                if (GW == 8 && left > 4) m8 |= fast_may_be_corner_x4_wide(t32 + 1, Lv.min_th + 1) >> 2;
            }
            if (gx == gpr - 1) m8 &= tail_mask;
            p0 = (int)__umul24(py, cw) + GW * gx;
        }
        const int cnt = __popc(m8);
        const int incl = wave_incl_scan(cnt);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int wbase = 0;
        if (lane == 0 && total) wbase = atomicAdd(&s_nsurv, total);
        unsigned short* q = s_surv + (__builtin_amdgcn_readfirstlane(wbase) + incl - cnt);
#pragma unroll
        for (int j = 0; j < GW; ++j)
            if (m8 & (1u << FQ_BIT[j])) *q++ = (unsigned short)(p0 + j);
    }
    __syncthreads();
    // pass 2: the full score, two survivors per lane on packed 16-bit lanes (an odd survivor count scores the last one twice)
    const int nsurv = s_nsurv;
    for (int i = tid; 2 * i < nsurv; i += FAST_NT) {
        const int pa = s_surv[2 * i], pb = s_surv[min(2 * i + 1, nsurv - 1)];
        const int ya = (int)(__umul24((unsigned)pa, inv_cw) >> 20), xa = pa - (int)__umul24(ya, cw);
        const int yb = (int)(__umul24((unsigned)pb, inv_cw) >> 20), xb = pb - (int)__umul24(yb, cw);
        const uint32_t s2 = fast_score_9_16_x2(&tile[(ya + 3) * TILE_PITCH + xa + 3], &tile[(yb + 3) * TILE_PITCH + xb + 3]);
        const int sa = (int)(short)(s2 & 0xffffu), sb = (int)(short)(s2 >> 16);
        // "not a corner at minTh" stores 0, like FAST_t's zeroed score rows
        score[(ya + 1) * SCORE_PITCH + xa + 1] = (uint8_t)(sa >= Lv.min_th ? sa : 0);
        score[(yb + 1) * SCORE_PITCH + xb + 1] = (uint8_t)(sb >= Lv.min_th ? sb : 0);
    }
    __syncthreads();

    // Strict 8-neighbour maxima: only a pixel with a score can be one, so the pass runs over the survivor list (dense lanes,
    // a tenth of the cell) and leaves one bit per maximum in s_max, and in s_ini when the score reaches iniTh (the cell keeps
    // those alone if there is any: reference :809-817 retries with minTh only when the cell is empty at iniTh).
    int any_ini = 0;
    for (int i = tid; i < nsurv; i += FAST_NT) {
        const int p = s_surv[i];
        const int py = (int)(__umul24((unsigned)p, inv_cw) >> 20), px = p - (int)__umul24(py, cw);
        const uint8_t* c = &score[(py + 1) * SCORE_PITCH + px + 1];
        const int s = c[0];
        const int m = max(max(max((int)c[-1], (int)c[1]), max((int)c[-SCORE_PITCH - 1], (int)c[-SCORE_PITCH])),
                          max(max((int)c[-SCORE_PITCH + 1], (int)c[SCORE_PITCH - 1]), max((int)c[SCORE_PITCH], (int)c[SCORE_PITCH + 1])));
        if (s > m) {   // (s > m >= 0: it has a score)
            atomicOr(&s_max[p >> 5], 1u << (p & 31));
            if (s >= Lv.ini_th) { atomicOr(&s_ini[p >> 5], 1u << (p & 31)); any_ini = 1; }
        }
    }
    if (any_ini) s_any = 1;  // benign race: all writers store 1
    __syncthreads();

    // Ordered output: candidates in row-major order, exactly FAST_t's emission order.  Wave 0 counts the chosen bitmap's 64-bit
    // words (lane j: pixels 64j .. 64j+63) and leaves their prefix sums; every maximum then finds its own position -- the
    // word's prefix plus the set bits below its own -- and stores itself (round 3 walked the bits of every word one by one on
    // wave 0: two dependent loops of up to 64 steps on one wave of the four).
    const unsigned int* bm = s_any ? s_ini : s_max;
    if (tid < 64) {
        const unsigned long long w = 2 * tid < bm_words ? (unsigned long long)bm[2 * tid] | ((unsigned long long)bm[2 * tid + 1] << 32) : 0ull;
        const int cnt = __popcll(w);
        const int incl = wave_incl_scan(cnt);
        s_pref[tid] = incl - cnt;
        if (tid == 63) cell_cnt[cell] = incl;
    }
    __syncthreads();
    uint32_t* slot = cell_items + Lv.slot_base + (size_t)cm.y * Lv.slot_cap;
    for (int i = tid; i < nsurv; i += FAST_NT) {
        const int p = s_surv[i];
        const unsigned int wlo = bm[p >> 5];
        if (!((wlo >> (p & 31)) & 1u)) continue;
        const unsigned below = (p & 32) ? (unsigned)__popc(bm[(p >> 5) - 1]) : 0u;   // (low half of the same 64-bit word)
        const int pos = s_pref[p >> 6] + (int)below + __popc(wlo & ((1u << (p & 31)) - 1u));
        const int py = (int)(__umul24((unsigned)p, inv_cw) >> 20), px = p - (int)__umul24(py, cw);
        const int sc = score[(py + 1) * SCORE_PITCH + px + 1];
        const int xr = x0 + px - MIN_BORDER, yr = y0 + py - MIN_BORDER;  // relative to (16,16), :821-826
        if (pos < Lv.slot_cap) slot[pos] = (uint32_t)xr | ((uint32_t)yr << 12) | ((uint32_t)sc << 24);
    }
}

// ------------------------------------------------------------------------------------------------ K3b
__global__ __launch_bounds__(1024) void k_compact(const LevelInfo* __restrict__ L, const int* __restrict__ cell_cnt,
                                                  const uint32_t* __restrict__ cell_items, int* __restrict__ cell_off,
                                                  uint32_t* __restrict__ cand /*pinned host*/,
                                                  int* __restrict__ level_cnt /*pinned host*/,
                                                  uint32_t* __restrict__ cand_dev, int* __restrict__ level_cnt_dev) {
    __shared__ int wsum[16];
    __shared__ int s_total;
    const LevelInfo Lv = L[blockIdx.x];
    const int ncell = Lv.n_cols * Lv.n_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (Lv.w == 0 || ncell == 0) {
        if (tid == 0) { level_cnt[blockIdx.x] = 0; level_cnt_dev[blockIdx.x] = 0; }
        return;
    }
    // exclusive scan of the cell counts, `per` consecutive cells per thread
    const int per = (ncell + 1023) / 1024;
    const int c0 = tid * per, c1 = min(ncell, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += min(cell_cnt[Lv.cell_base + c], Lv.slot_cap);
    const int incl = wave_incl_scan(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; wsum[w] = acc; acc += v; }
        s_total = acc;
    }
    __syncthreads();
    int run = wsum[wave] + incl - mine;
    for (int c = c0; c < c1; ++c) {
        cell_off[Lv.cell_base + c] = run;
        run += min(cell_cnt[Lv.cell_base + c], Lv.slot_cap);
    }
    __syncthreads();
    // copy: one wave per cell
    for (int c = wave; c < ncell; c += 16) {
        const int n = min(cell_cnt[Lv.cell_base + c], Lv.slot_cap);
        const int off = cell_off[Lv.cell_base + c];
        const uint32_t* src = cell_items + Lv.slot_base + (size_t)c * Lv.slot_cap;
        for (int i = lane; i < n; i += 64) { const uint32_t v = src[i]; cand[Lv.cand_base + off + i] = v; cand_dev[Lv.cand_base + off + i] = v; }
    }
    if (tid == 0) { level_cnt[blockIdx.x] = s_total; level_cnt_dev[blockIdx.x] = s_total; }
}


// ------------------------------------------------------------------------------------------------ K4 on the device
// DistributeOctTree (reference src/ORBextractor.cc:540-764) as ONE workgroup per (camera, level), all state in LDS.
//
// The reference keeps a std::list of nodes; a node owns the keypoints inside its box.  What has to be reproduced
// exactly is (i) which nodes get split in which order and (ii) the final list order, because the output is "best
// keypoint of every node, in list order".  Both follow from three facts about the reference's loops:
//   * full pass: every node with > 1 keypoint is split, visiting the list front to back; the children (n1..n4, empty
//     ones dropped) are pushed to the FRONT.  So after a pass the list is  reverse(creation order of all new children)
//     followed by the surviving old nodes in their old order.
//   * careful pass (entered when size + 3*expandable > N): the children created by the previous pass that hold > 1
//     keypoint are sorted by (size, creation order) and split from the back (largest, newest first) until the list
//     reaches N nodes.  Every split grows the list by (#non-empty children - 1), so the stopping point is a prefix sum.
//   * a node's keypoints keep their relative order when they are dealt to its children (stable 4-way partition), which
//     only the final "first maximum response wins" depends on: inside any node the keypoints stay in candidate order.
// Here the list is an array in list order (node id == list position, rebuilt every pass).  Round 3: THE KEYS NEVER MOVE.
// A key is the 32-bit candidate word k_fast_cells wrote (x | y << 12 | response << 24) at its candidate position p, plus
// the 16-bit id of the node that owns it.  Nothing in the algorithm needs a node's keys to be contiguous: a split needs
// the four child COUNTS of the node (an LDS histogram: one 64-bit atomic add of 1 << 16*child per key, one add per wave
// while a wave's keys share a node), the new list needs only per-node records, and "first maximum in the node's key order"
// is the maximum of (response, -p).  So a pass is: node phase (one thread per node: child counts -> one packed 64-bit
// scan -> new node records) and key phase (every key: its new node from its parent's record, and -- same visit -- its
// child inside that new node added to the node's histogram for the NEXT pass).  No per-key prefix sum, no scatter, no
// double-buffered key arrays: 6 bytes of LDS per candidate, so 16 384 candidates (the dense levels of a 1080p image)
// fit the same single-launch kernel that used to hold 4096 (the HBM-backed second launch of rounds 1-2 is gone).
// Creation order inside a pass replaces the reference's heap-address tie-break exactly as the host code and the oracle do
// (SURVEY App. C-1).
constexpr int OCT_NL = 1024;   // live nodes (>= quota + 4)

__device__ __forceinline__ int oct_block_excl_scan(int val, int tid, int* wsum, int* total) {
    // exclusive scan of one int per thread over the 1024-thread block
    const int lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_scan(val);
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[w]; if (w < wave) base += x; tot += x; }
    if (total) *total = tot;
    return base + incl - val;
}

__device__ __forceinline__ unsigned long long oct_pack_box(int ulx, int uly, int brx, int bry) {
    return (unsigned long long)(unsigned)(ulx & 0xffff) | ((unsigned long long)(unsigned)(uly & 0xffff) << 16) |
           ((unsigned long long)(unsigned)(brx & 0xffff) << 32) | ((unsigned long long)(unsigned)(bry & 0xffff) << 48);
}

// child of a key inside a node box: DivideNode's halfX/halfY and its '<' tests (:482-523)
__device__ __forceinline__ int oct_child(int x, int y, unsigned long long bx, int& midx, int& midy) {
    // ceil((float)d / 2) for an integer d >= 0 (boxes never invert; d < 2^24 is exact in float) == (d + 1) >> 1: the float
    // division of the reference costs ~15 instructions per coordinate here, twice per key and pass
    const int ulx = (int)(bx & 0xffff), uly = (int)((bx >> 16) & 0xffff), brx = (int)((bx >> 32) & 0xffff), bry = (int)(bx >> 48);
    midx = ulx + ((brx - ulx + 1) >> 1);
    midy = uly + ((bry - uly + 1) >> 1);
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);
}

// box of child c of box bx (n1..n4 of DivideNode)
__device__ __forceinline__ unsigned long long oct_child_box(unsigned long long bx, int c, int midx, int midy) {
    const int ulx = (int)(bx & 0xffff), uly = (int)((bx >> 16) & 0xffff), brx = (int)((bx >> 32) & 0xffff), bry = (int)(bx >> 48);
    return oct_pack_box((c & 1) ? midx : ulx, (c & 2) ? midy : uly, (c & 1) ? brx : midx, (c & 2) ? bry : midy);
}

__device__ __forceinline__ int oct_nonzero_fields(unsigned long long t) {
    return ((t & 0xffffull) ? 1 : 0) + ((t & 0xffff0000ull) ? 1 : 0) + ((t & 0xffff00000000ull) ? 1 : 0) + ((t >> 48) ? 1 : 0);
}

MORB_PHASE_DECL(g_ph_oct);
#ifdef MORB_PHASE_CLOCKS
__device__ int g_oct_dbg = 0;   // experiments on the instrumented build: 1 = no histogram adds, 2 = no record read either, 4 = no node-id update
#define OCT_DBG(bit) (g_oct_dbg & (bit))
#else
#define OCT_DBG(bit) 0
#endif

constexpr int OCT_KPT = 40;                    // keys per thread
constexpr int OCT_RK = OCT_KPT * 1024 - 1;     // candidates per (camera, level): counts and key positions are 16-bit fields
constexpr int OCT_P = 8;                       // copies of a node's child histogram
constexpr int OCT_MAXCELLS = 8191;             // cells of a level (offsets alias box + pc)

struct OctLdsR {
    union {
        struct { unsigned long long box[2][OCT_NL]; unsigned long long pc[2][OCT_NL]; } n;
        int cell_off[OCT_MAXCELLS + 1];
    } u;
    uint4 rec[OCT_NL];                     // what a key needs to know about its node in this pass (oct_rec_*): ONE 16-byte read per key
    unsigned long long pcp[OCT_NL][OCT_P]; // child histograms of the nodes made in this pass, OCT_P copies each (a lane adds to copy lane % OCT_P)
    unsigned short cnt[2][OCT_NL];
    unsigned short ncrt[2][OCT_NL];
    unsigned short procidx[OCT_NL];
    unsigned short P[OCT_NL];
    unsigned short cnode[OCT_NL];
    unsigned int sortkey[OCT_NL];
    unsigned int best[OCT_NL];             // response << 16 | (65535 - p)
    unsigned long long wsum64[2][16];
    int wsum[16];
    int v[8];
};
static_assert(sizeof(OctLdsR) <= 144 * 1024, "node records of the register-resident quadtree");

// The record a pass's node phase leaves for the key phase (round 6: ~70 vector instructions per key and pass became ~30 -- one
// compute unit walks all keys of a level, so instructions per key ARE the kernel's time on dense levels):
//   node not split in this pass:  w = its new list position
//   node split:  x = midx | midy << 16 (DivideNode's halves), y = the x-halves of its left | right children, z = the y-halves of its upper |
//                lower children (the grandchild test), w = list position of its FIRST non-empty child | OCT_SPLIT | for child c the number of
//                non-empty children before it (2 bits at 16 + 2c: position = first - that) | child c holds more than one key (bit 24 + c)
constexpr unsigned OCT_SPLIT = 0x8000u;
__device__ __forceinline__ uint4 oct_rec_split(unsigned long long bx, unsigned long long cnts, int first_pos) {
    const int ulx = (int)(bx & 0xffff), uly = (int)((bx >> 16) & 0xffff), brx = (int)((bx >> 32) & 0xffff), bry = (int)(bx >> 48);
    const int midx = ulx + ((brx - ulx + 1) >> 1), midy = uly + ((bry - uly + 1) >> 1);
    const int q1x = ulx + ((midx - ulx + 1) >> 1), q3x = midx + ((brx - midx + 1) >> 1);
    const int q1y = uly + ((midy - uly + 1) >> 1), q3y = midy + ((bry - midy + 1) >> 1);
    unsigned w = (unsigned)first_pos | OCT_SPLIT;
    int before = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cc = (int)((cnts >> (16 * c)) & 0xffff);
        w |= (unsigned)before << (16 + 2 * c);
        if (cc > 1) w |= 1u << (24 + c);
        before += cc ? 1 : 0;
    }
    return make_uint4((unsigned)midx | ((unsigned)midy << 16), (unsigned)q1x | ((unsigned)q3x << 16), (unsigned)q1y | ((unsigned)q3y << 16), w);
}
// a key's new node nn and, if that node holds more than one key, its child c2 in there (act)
__device__ __forceinline__ void oct_rec_apply(const uint4 R, int x, int y, int& nn, int& c2, bool& act) {
    nn = (int)(R.w & 0xfffu); c2 = 0; act = false;
    if (R.w & OCT_SPLIT) {
        const int c = (x >= (int)(R.x & 0xffff) ? 1 : 0) | (y >= (int)(R.x >> 16) ? 2 : 0);
        const int qx = (int)((c & 1) ? R.y >> 16 : R.y & 0xffff), qy = (int)((c & 2) ? R.z >> 16 : R.z & 0xffff);
        c2 = (x >= qx ? 1 : 0) | (y >= qy ? 2 : 0);
        nn -= (int)((R.w >> (16 + 2 * c)) & 3u);
        act = ((R.w >> (24 + c)) & 1u) != 0;
    }
}

// One key's contribution to the child histogram of its node: child c of node nn.  The 64 keys of a wave are neighbours in cell order and
// share a few nodes, and same-address LDS atomics take a clock per lane in the ONE LDS unit all sixteen waves share -- so every node has
// OCT_P copies of its histogram and a lane adds to copy lane % OCT_P: at most 64 / OCT_P lanes per address.  (Rounds 3-5 counted a wave's
// keys per node with ballots first: that chain of scalar instructions per key cost 4 us per pass on a level of 10 000 candidates,
// profiles/r06/notes_experiments.md; the copies are summed once per node by the node's own thread.)
__device__ __forceinline__ void oct_add(unsigned long long (*pcp)[OCT_P], int nn, int c, bool act) {
    if (act) atomicAdd(&pcp[nn][threadIdx.x & (OCT_P - 1)], 1ull << (16 * c));
}
__device__ __forceinline__ unsigned long long oct_sum_copies(const unsigned long long (*pcp)[OCT_P], int node) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(pcp[node]);
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < OCT_P / 2; ++i) { const ulonglong2 v = q[i]; t += v.x + v.y; }   // (16-bit fields, totals below 65536: no carries)
    return t;
}
__device__ __forceinline__ void oct_zero_copies(unsigned long long (*pcp)[OCT_P], int node) {
    ulonglong2* q = reinterpret_cast<ulonglong2*>(pcp[node]);
#pragma unroll
    for (int i = 0; i < OCT_P / 2; ++i) q[i] = make_ulonglong2(0ull, 0ull);
}

// (`base` = k * 1024 is kept opaque to the optimiser: anything it could derive per k -- key positions, remaining counts -- would be
// hoisted out of the pass loop into 64 more live registers)
// Keys are walked OCT_G at a time (four; two in the large size classes, whose registers are the keys): one block-uniform test per group, the
// group's bodies in one basic block -- their LDS reads and their global loads are in flight together instead of one round trip per key.
__device__ __forceinline__ int oct_opaque(int v) { asm volatile("" : "+s"(v)); return v; }
#define OCT_FOR_KEYS _Pragma("unroll") for (int g_ = 0; g_ < KPT / OCT_G; ++g_) if (const int gbase_ = oct_opaque(g_ * OCT_G * 1024); gbase_ < n) \
                     _Pragma("unroll") for (int j_ = 0; j_ < OCT_G; ++j_) { const int k = g_ * OCT_G + j_; const int base = gbase_ + j_ * 1024;
#define OCT_KN_GET(k) (((k) & 1) ? (int)(kn2[(k) >> 1] >> 16) : (int)(kn2[(k) >> 1] & 0xffffu))
#define OCT_KN_SET(k, val) do { if ((k) & 1) kn2[(k) >> 1] = (kn2[(k) >> 1] & 0xffffu) | ((uint32_t)(val) << 16); \
                                else kn2[(k) >> 1] = (kn2[(k) >> 1] & 0xffff0000u) | (uint32_t)(val); } while (0)

template <int KPT>
__device__ __forceinline__ void oct_run(OctLdsR& L, const LevelInfo& Lv, const uint32_t* __restrict__ cell_items, SelKp* __restrict__ sel,
                                        int* __restrict__ sel_cnt, int* __restrict__ status, int max_levels, int n, int N, int ncell,
                                        int nIni, float hX, int height) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int OCT_G = KPT <= 16 ? 4 : 2;
    const int* cell_off = L.u.cell_off;
    const unsigned short* coarse = reinterpret_cast<const unsigned short*>(L.sortkey);   // (built by the kernel: cell of every 32nd candidate)
    uint32_t key[KPT];
    uint32_t kn2[KPT / 2];
    unsigned long long rootcnt = 0;
#pragma unroll
    for (int k = 0; k < KPT; ++k) { key[k] = 0; if ((k & 1) == 0) kn2[k >> 1] = 0; }
    OCT_FOR_KEYS
        {
            const int p = base + tid;
            if (p < n) {
                // the cell of candidate p: the cell that holds position p & ~31 (table), then a few cells forward; bisection if that is
                // not enough (runs of empty cells)
                int lo = coarse[p >> 5];
                if (!OCT_DBG(16)) {
#pragma unroll
                for (int sstep = 0; sstep < 4; ++sstep) if (cell_off[lo + 1] <= p) ++lo;
                }
                if (!OCT_DBG(16) && cell_off[lo + 1] <= p) {
                    int hi = ncell;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (cell_off[mid] <= p) lo = mid; else hi = mid;
                    }
                }
                if (OCT_DBG(8)) key[k] = (uint32_t)(lo + p); else
                key[k] = cell_items[Lv.slot_base + (size_t)lo * Lv.slot_cap + (p - cell_off[lo])];   // (nothing below waits for it: all of a thread's loads are in flight together)
            }
        }
    }
    if (nIni > 1) {   // keys per root strip (:544-585: dealt by (int)(x / hX)) -- in a loop of its own, behind ALL the loads
        OCT_FOR_KEYS
            if (base + tid < n) {
                const int r = min(max((int)((float)(key[k] & 0xfff) / hX), 0), nIni - 1);
                rootcnt += 1ull << (16 * r);
            }
        }
    }
    {
        const unsigned long long incl = wave_incl_scan(nIni > 1 ? rootcnt : 0ull);
        if (lane == 63) L.wsum64[0][wave] = incl;
    }
    __syncthreads();   // every key is in its register; the cell offsets are dead (box / pc may be written)
    MORB_PHASE(g_ph_oct, 2);
    int rpos[4] = {0, 0, 0, 0};
    int sz = 0;
    {
        unsigned long long total = 0;
        if (nIni > 1) { for (int w = 0; w < 16; ++w) total += L.wsum64[0][w]; }
        else total = (unsigned long long)n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = (int)((total >> (16 * r)) & 0xffff);
            rpos[r] = sz;
            if (r < nIni && c) {
                if (tid == r) {
                    L.u.n.box[0][sz] = oct_pack_box((int)(hX * (float)r), 0, (int)(hX * (float)(r + 1)), height);
                    L.cnt[0][sz] = (unsigned short)c; L.ncrt[0][sz] = 0xffff; L.u.n.pc[0][sz] = 0;
                    oct_zero_copies(L.pcp, sz);
                }
                ++sz;
            }
        }
    }
    __syncthreads();
    OCT_FOR_KEYS
        {
            const int p = base + tid;
            int nn = 0, c = 0;
            bool act = false;
            if (p < n) {
                const uint32_t v = key[k];
                const int x = (int)(v & 0xfff), y = (int)((v >> 12) & 0xfff);
                int r = 0;
                if (nIni > 1) r = min(max((int)((float)x / hX), 0), nIni - 1);
                nn = rpos[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (r == q) nn = rpos[q];
                OCT_KN_SET(k, nn);
                if (L.cnt[0][nn] > 1) { int mx, my; c = oct_child(x, y, L.u.n.box[0][nn], mx, my); act = true; }
            }
            oct_add(L.pcp, nn, c, act);
        }
    }
    __syncthreads();
    MORB_PHASE(g_ph_oct, 3);
    int a = 0;
    int ph_i = 4;
    bool careful = false, finish = false;
    bool first = true;
    while (!finish) {
        const int b = a ^ 1;
        const int prev_size = sz;
        int np = 0;
        // the histograms the key phase just filled (nodes made in the last pass; the roots the first time): summed by the node's thread.
        // Whoever else reads a node's sum does so behind a barrier (the careful pass's scans), the node phase reads its own.
        if (tid < sz && (first || L.ncrt[a][tid] != 0xffff)) L.u.n.pc[a][tid] = oct_sum_copies(L.pcp, tid);
        first = false;
        if (careful) {
            int isc = 0;
            unsigned int skey = 0;
            if (tid < sz && L.ncrt[a][tid] != 0xffff && L.cnt[a][tid] > 1) {
                isc = 1;
                skey = ((unsigned)L.cnt[a][tid] << 16) | L.ncrt[a][tid];
            }
            int nc = 0;
            const int ci = oct_block_excl_scan(isc, tid, L.wsum, &nc);
            if (isc) { L.sortkey[ci] = skey; L.cnode[ci] = (unsigned short)tid; }
            L.procidx[tid] = 0xffff;
            if (tid == 0) L.v[1] = nc;
            __syncthreads();
            if (nc == 0) break;
            if (tid < nc) {
                const unsigned int mk = L.sortkey[tid];
                int myrank = 0;
                for (int j = 0; j < nc; ++j) myrank += L.sortkey[j] > mk ? 1 : 0;
                const unsigned short node = L.cnode[tid];
                L.P[myrank] = node;
                L.procidx[node] = (unsigned short)myrank;
            }
            __syncthreads();
            int growth = 0;
            if (tid < nc) growth = oct_nonzero_fields(L.u.n.pc[a][L.P[tid]]) - 1;
            int gtot = 0;
            const int gex = oct_block_excl_scan(growth, tid, L.wsum, &gtot);
            if (tid < nc && prev_size + gex + growth >= N && prev_size + gex < N) L.v[1] = tid + 1;
            __syncthreads();
            np = L.v[1];
            if (tid >= np && tid < nc) L.procidx[L.P[tid]] = 0xffff;
            __syncthreads();
        }
        unsigned long long tot = 0, packed = 0;
        int pnode = -1;
        if (careful) { if (tid < np) pnode = L.P[tid]; }
        else if (tid < sz && L.cnt[a][tid] > 1) pnode = tid;
        if (pnode >= 0) {
            tot = L.u.n.pc[a][pnode];
            int nch = 0, nex = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int cc = (int)((tot >> (16 * c)) & 0xffff); nch += cc ? 1 : 0; nex += cc > 1 ? 1 : 0; }
            packed = (unsigned long long)nch | ((unsigned long long)nex << 32) | (1ull << 48);
        }
        const bool survivor = tid < sz && (careful ? L.procidx[tid] == 0xffff : L.cnt[a][tid] <= 1);
        if (survivor) packed |= 1ull << 16;
        const unsigned long long incl = wave_incl_scan(packed);
        unsigned long long* ws = L.wsum64[1];
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const unsigned long long x = ws[w]; if (w < wave) before += x; total += x; }
        before += incl - packed;
        const int M = (int)(total & 0xffff), nexp = (int)((total >> 32) & 0xffff);
        np = (int)(total >> 48);
        const int cb = (int)(before & 0xffff), sr = (int)((before >> 16) & 0xffff);
        if (survivor) {
            const int np_ = M + sr;
            L.rec[tid] = make_uint4(0u, 0u, 0u, (unsigned)np_);
            L.u.n.box[b][np_] = L.u.n.box[a][tid]; L.cnt[b][np_] = L.cnt[a][tid]; L.u.n.pc[b][np_] = L.u.n.pc[a][tid];
            L.ncrt[b][np_] = 0xffff;
        }
        if (pnode >= 0) {
            const unsigned long long bx = L.u.n.box[a][pnode];
            L.rec[pnode] = oct_rec_split(bx, tot, M - 1 - cb);
            int mx, my;
            (void)oct_child(0, 0, bx, mx, my);
            int ci = cb;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = (int)((tot >> (16 * c)) & 0xffff);
                if (cc == 0) continue;
                const int pos = M - 1 - ci;
                L.u.n.box[b][pos] = oct_child_box(bx, c, mx, my);
                L.cnt[b][pos] = (unsigned short)cc; L.ncrt[b][pos] = (unsigned short)ci; L.u.n.pc[b][pos] = 0;
                oct_zero_copies(L.pcp, pos);
                ++ci;
            }
        }
        __syncthreads();
        OCT_FOR_KEYS
            {
                const int p = base + tid;
                int nn = 0, c2 = 0;
                bool act = false;
                if (p < n && !OCT_DBG(2)) {
                    const uint4 R = L.rec[OCT_KN_GET(k)];
                    uint32_t v = key[k];
                    asm volatile("" : "+v"(v));   // (x and y of a key are the same in every pass: not hoisted out of the pass loop either)
                    oct_rec_apply(R, (int)(v & 0xfff), (int)((v >> 12) & 0xfff), nn, c2, act);
                    if (!OCT_DBG(4)) OCT_KN_SET(k, nn);
                }
                if (!OCT_DBG(1)) oct_add(L.pcp, nn, c2, act);
            }
        }
        __syncthreads();
        sz = M + (sz - np);
        a = b;
        MORB_PHASE(g_ph_oct, ph_i); ph_i = min(ph_i + 1, 40);
        if (sz >= N || sz == prev_size) finish = true;
        else if (!careful && sz + nexp * 3 > N) careful = true;
    }
    if (sz > N + 4) { if (tid == 0) { sel_cnt[blk] = 0; status[blk] = 1; } return; }
    // ---- best keypoint per node, first maximum wins (:742-763).  A node's keys sit next to each other in the waves: eight copies of
    // every node's maximum (the boxes and histograms are dead: their LDS takes them), a lane adds to copy lane & 7, so that an atomic
    // instruction meets at most eight lanes per address instead of sixty-four.
    unsigned* best8 = reinterpret_cast<unsigned*>(&L.u);
    for (int i = tid; i < sz * 8; i += 1024) best8[i] = 0;
    __syncthreads();
    OCT_FOR_KEYS
        {
            const int p = base + tid;
            if (p < n) atomicMax(&best8[OCT_KN_GET(k) * 8 + (lane & 7)], ((key[k] >> 24) << 16) | (unsigned)(65535 - p));
        }
    }
    __syncthreads();
    if (tid < sz) {
        const uint4 lo4 = *reinterpret_cast<const uint4*>(best8 + tid * 8), hi4 = *reinterpret_cast<const uint4*>(best8 + tid * 8 + 4);
        L.best[tid] = max(max(max(lo4.x, lo4.y), max(lo4.z, lo4.w)), max(max(hi4.x, hi4.y), max(hi4.z, hi4.w)));
    }
    __syncthreads();
    OCT_FOR_KEYS
        {
            const int p = base + tid;
            if (p < n) {
                const int node = OCT_KN_GET(k);
                const uint32_t v = key[k];
                if (L.best[node] == (((v >> 24) << 16) | (unsigned)(65535 - p))) {   // (response, -p) is unique: one winner per node
                    SelKp K;
                    K.x = (int)(v & 0xfff) + MIN_BORDER; K.y = (int)((v >> 12) & 0xfff) + MIN_BORDER;
                    K.camlevel = ((blk / max_levels) << 8) | (blk % max_levels);
                    K.resp_out = (int)((v & 0xff000000u) | (unsigned)node);
                    sel[Lv.sel_base + node] = K;
                }
            }
        }
    }
    if (tid == 0) { sel_cnt[blk] = sz; status[blk] = 0; }
    MORB_PHASE(g_ph_oct, 62); MORB_PHASE(g_ph_oct, 63);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0 && blk == 0) g_ph_oct[61] = (unsigned long long)ph_i;
#endif
}

__global__ __launch_bounds__(1024) void k_octree(const LevelInfo* __restrict__ Lv_all, const int* __restrict__ cell_cnt,
                                                     const uint32_t* __restrict__ cell_items, SelKp* __restrict__ sel,
                                                     int* __restrict__ sel_cnt, int* __restrict__ status, int max_levels,
                                                     int max_keys) {
    MORB_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) unsigned char oct_raw[];
    OctLdsR& L = *reinterpret_cast<OctLdsR*>(oct_raw);
    const int blk = blockIdx.x, tid = threadIdx.x;
    MORB_PHASE(g_ph_oct, 0);
    const LevelInfo Lv = Lv_all[blk];
    const int N = Lv.quota;
    const int ncell = Lv.w ? Lv.n_cols * Lv.n_rows : 0;
    int* cell_off = L.u.cell_off;
    if (ncell > OCT_MAXCELLS) { if (tid == 0) { sel_cnt[blk] = 0; status[blk] = 1; } return; }
    int n = 0;
    {
        const int per = (ncell + 1023) / 1024;
        const int c0 = min(ncell, tid * per), c1 = min(ncell, c0 + per);
        int mine = 0;
        for (int c = c0; c < c1; ++c) mine += min(cell_cnt[Lv.cell_base + c], Lv.slot_cap);
        const int ex = oct_block_excl_scan(mine, tid, L.wsum, &n);
        int run = ex;
        for (int c = c0; c < c1; ++c) { cell_off[c] = run; run += min(cell_cnt[Lv.cell_base + c], Lv.slot_cap); }
        if (tid == 0) cell_off[ncell] = n;
        __syncthreads();
    }
    if (n == 0) { if (tid == 0) { sel_cnt[blk] = 0; status[blk] = 0; } return; }
    const int width = Lv.w - 2 * MIN_BORDER, height = Lv.h - 2 * MIN_BORDER;
    const int nIni = max(1, (int)roundf((float)width / (float)height));
    if (n > max_keys || n > OCT_RK || N + 4 > OCT_NL || nIni > 4 || width >= 32768 || height >= 32768) {
        if (tid == 0) { sel_cnt[blk] = 0; status[blk] = 1; }
        return;
    }
    const float hX = (float)width / (float)nIni;
    {   // the cell of every 32nd candidate (u16, in the careful pass's sort keys: unused until then)
        unsigned short* coarse = reinterpret_cast<unsigned short*>(L.sortkey);
        for (int c = tid; c < ncell; c += 1024) {
            const int a0 = cell_off[c], b0 = cell_off[c + 1];
            for (int q = (a0 + 31) >> 5; (q << 5) < b0; ++q) coarse[q] = (unsigned short)c;
        }
        __syncthreads();
    }
    MORB_PHASE(g_ph_oct, 1);
    // one instantiation per size class: a level with 3 000 candidates runs loops of four keys per thread, not 64 guards per loop
    if (n <= 4 * 1024) oct_run<4>(L, Lv, cell_items, sel, sel_cnt, status, max_levels, n, N, ncell, nIni, hX, height);
    else if (n <= 16 * 1024) oct_run<16>(L, Lv, cell_items, sel, sel_cnt, status, max_levels, n, N, ncell, nIni, hX, height);
    else if (n <= 32 * 1024) oct_run<32>(L, Lv, cell_items, sel, sel_cnt, status, max_levels, n, N, ncell, nIni, hX, height);
    else oct_run<OCT_KPT>(L, Lv, cell_items, sel, sel_cnt, status, max_levels, n, N, ncell, nIni, hX, height);
}
#undef OCT_FOR_KEYS
#undef OCT_KN_GET
#undef OCT_KN_SET

// ------------------------------------------------------------------------------------------------ K5-K7
__device__ __forceinline__ int reflect101(int p, int n) {
    // |offset| beyond the edge is at most 3 and n >= 39, so one reflection suffices
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

// cv::fastAtan2 (OpenCV 2.4/3.x mathfuncs), float32, no contraction (App. A-5)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Canonical sin/cos: fixed IEEE-double operation sequence, bit-identical on host and device.
__device__ __forceinline__ void det_sincos(float angle_rad, float& cosv, float& sinv) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    const double x = (double)angle_rad;
    const double kf = rint(x * TWO_OVER_PI);
    const double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    const double z = r * r;
    double ps = 1.0 / 355687428096000.0;
    ps = ps * z + (-1.0 / 1307674368000.0);
    ps = ps * z + (1.0 / 6227020800.0);
    ps = ps * z + (-1.0 / 39916800);
    ps = ps * z + (1.0 / 362880);
    ps = ps * z + (-1.0 / 5040);
    ps = ps * z + (1.0 / 120);
    ps = ps * z + (-1.0 / 6);
    const double s = r + r * (z * ps);
    double pc = 1.0 / 20922789888000.0;
    pc = pc * z + (-1.0 / 87178291200.0);
    pc = pc * z + (1.0 / 479001600);
    pc = pc * z + (-1.0 / 3628800);
    pc = pc * z + (1.0 / 40320);
    pc = pc * z + (-1.0 / 720);
    pc = pc * z + (1.0 / 24);
    pc = pc * z + (-1.0 / 2);
    const double c = 1.0 + z * pc;
    const int q = ((int)kf) & 3;
    const double cc = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    const double ss = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cosv = (float)cc;
    sinv = (float)ss;
}

// Each wave owns a private LDS region; a wave's DS operations execute in issue order, so only the compiler has to be
// kept from reordering the LDS stores of one phase past the loads of the next.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int PR = 22;                 // patch half-size: 19 (largest rotated tap + guard) + 3 (blur)
constexpr int PW = 2 * PR + 1;         // 45
constexpr int RAW_PITCH = 48;
constexpr int BR_ = 19;                // blurred half-size
constexpr int BW = 2 * BR_ + 1;        // 39
constexpr int ROW_PITCH = 40;
constexpr int VP_PAIRS = 24;           // row pairs of the horizontal pass: 23 hold the 45 rows, one more is read (never used) by the last segment

MORB_PHASE_DECL(g_ph_desc);
#ifdef MORB_PHASE_CLOCKS
#define DPH(i) do { if (blockIdx.x == 40 && threadIdx.x == 0) g_ph_desc[i] = wall_clock64(); \
                     if ((i) == 0 && (threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 4096) g_desc_wave[0][blockIdx.x * 4 + (threadIdx.x >> 6)] = wall_clock64(); \
                     if ((i) == 7 && (threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 4096) g_desc_wave[1][blockIdx.x * 4 + (threadIdx.x >> 6)] = wall_clock64(); } while (0)
__device__ unsigned long long g_desc_wave[2][4096];
#else
#define DPH(i) do {} while (0)
#endif
#ifdef MORB_DESCRIBE_SELFCHECK
__device__ unsigned long long g_describe_check[8 + 16 * 14 + 16 + 4];
#endif
__global__ __launch_bounds__(256) void k_describe(const LevelInfo* __restrict__ L, int max_levels,
                                                  const uint8_t* __restrict__ pyr, size_t cam_pitch,
                                                  const SelKp* __restrict__ sel, int nsel,
                                                  orb_keypoint* const* __restrict__ kps_out,
                                                  uint8_t* const* __restrict__ desc_out, MirrorArgs mir, SelListArgs sl,
                                                  FrameSink sink, const L0Src* __restrict__ l0, const DescribeTables* __restrict__ tabs) {
    __shared__ alignas(16) uint8_t s_raw[4][PW * RAW_PITCH];
    __shared__ alignas(16) uint32_t s_vp[4][VP_PAIRS * ROW_PITCH];   // horizontal sums as VERTICAL pairs: row 2m | row 2m+1 << 16 per column
#ifdef MORB_DESCRIBE_OWN_BLUR
    __shared__ alignas(16) uint8_t s_blur[4][BW * ROW_PITCH];
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroups go to the 8 XCDs round robin and every XCD has an L2 of its own.  The slots are ordered (camera, level, list
    // position), so workgroup b takes the b-th group of four of XCD (b % 8)'s CONTIGUOUS eighth of them: one L2 then serves
    // whole levels (a whole camera of an 8-camera rig) and a 128-byte line shared by neighbouring 45 x 48-byte patches is
    // fetched into ONE L2, once (round 3 counters: 150 MB fetched per 8 x 1080p step for 53 MB of pyramid -- every level was
    // pulled into all eight).  Affinity only: any placement computes the same.
    const int ki = __builtin_amdgcn_readfirstlane(xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave);
    const unsigned short* slot_blk = sl.slot_blk;
    if (slot_blk && ki == nsel - 1) {
        // bookkeeping of the device-quadtree mode, once per launch: per-camera totals and the fallback flag.  Done by the wave
        // of the launch's last slot (a level's fourth spare slot stays empty, so this wave has no keypoint of its own to
        // delay), one (camera, level) count per lane where the rig fits a wave.
        const int nb = sl.n_cams * max_levels;
        int bad = 0;
        if (nb <= 64) {
            int cnt = 0;
            if (lane < nb) { cnt = sl.sel_cnt[lane]; bad = sl.status[lane]; }
            const int incl = wave_incl_scan(cnt);
            const int last = min(lane * max_levels + max_levels - 1, 63), first = lane * max_levels;   // lane c: camera c's range
            const int hi = __shfl(incl, last), lo = __shfl(incl - cnt, min(first, 63));
            if (lane < sl.n_cams) { sl.n_out[lane] = hi - lo; sl.h_n_out[lane] = hi - lo; }
        } else {
            for (int c = lane; c < sl.n_cams; c += 64) {
                int run = 0;
                for (int l = 0; l < max_levels; ++l) { run += sl.sel_cnt[c * max_levels + l]; bad |= sl.status[c * max_levels + l]; }
                sl.n_out[c] = run; sl.h_n_out[c] = run;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
        if (lane == 0) { sl.h_n_out[sl.n_cams] = bad; sl.n_out[sl.n_cams] = bad; }
    }
    DPH(0);
    if (ki >= nsel) return;
    SelKp K = sel[ki];
    int mirror_base = 0;
    if (slot_blk) {
        // device-quadtree mode: `sel` is the slotted list (quota + 4 slots per (camera, level)); the output position is
        // the level's base inside its camera plus the list position the quadtree stored in the low 24 bits.  Every wave
        // sums the block counts in front of its own block itself (a handful of values): no offsets kernel in between.
        const int blk = slot_blk[ki];
        const int local = ki - L[blk].sel_base;
        if (local >= sl.sel_cnt[blk]) return;
        const int first_of_cam = (blk / max_levels) * max_levels;
        int before_cam = 0, before_blk = 0;
        for (int i = lane; i < blk; i += 64) {
            const int v = sl.sel_cnt[i];
            if (i < first_of_cam) before_cam += v; else before_blk += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before_cam += __shfl_xor(before_cam, o); before_blk += __shfl_xor(before_blk, o); }
        K.resp_out = (int)(((unsigned)K.resp_out & 0xff000000u) | (unsigned)(before_blk + local));
        mirror_base = before_cam;
    }
    const int cam = K.camlevel >> 8, level = K.camlevel & 0xff;
    if (!slot_blk) mirror_base = mir.base[cam];
    const LevelInfo Lv = L[cam * max_levels + level];
    const uint8_t* img = pyr + cam * cam_pitch + Lv.pyr_off;
    int img_stride = Lv.stride;
    if (l0 && level == 0) {   // level 0 may lie in the caller's buffer (k_set_l0)
        const L0Src s0 = l0[cam];
        if (s0.ptr) { img = s0.ptr; img_stride = s0.stride; }
    }
    uint8_t* raw = s_raw[wave];
    uint32_t* vp = s_vp[wave];
    // The blurred window lives where the raw patch was (round 5): the patch is dead once the horizontal pass has turned it into row
    // pairs (the moments read it before that), the vertical pass writes 39 x 40 bytes from `vp` alone, and a wave's LDS accesses keep
    // their order -- 24 KB per workgroup instead of 30: six workgroups per compute unit instead of five.
    static_assert(BW * ROW_PITCH <= PW * RAW_PITCH, "the blurred window fits the raw patch's storage");
#ifdef MORB_DESCRIBE_OWN_BLUR
    uint8_t* blur = s_blur[wave];
#else
    uint8_t* blur = raw;
#endif

    DPH(1);
    // The depth sample of the frame assembly (lane 0, end of the kernel) is a dependent global load: issued here, used there.
    const bool to_sink = slot_blk && sink.x;
    orb_keypoint* const kps_dst = kps_out[cam];   // (pointer loads: also ahead of their use at the end)
    uint8_t* const desc_dst = desc_out[cam];
    float depth_sample = 0.f;
    bool have_depth = false;
    if (lane == 0 && to_sink) {
        const float* depth = sink.cam_depth[cam & 3];
        if (depth) {
            const float kx = level ? (float)K.x * Lv.scale : (float)K.x, ky = level ? (float)K.y * Lv.scale : (float)K.y;
            depth_sample = depth[(size_t)(int)ky * sink.cam_depth_stride[cam & 3] + (int)kx];  // imDepth.at<float>(v,u)
            have_depth = true;
        }
    }
    // 45x45 raw patch, reflect-101 at the level edge (== GaussianBlur's border on the cloned level, :1086-1087), fetched as 12
    // unaligned dwords per row with every load of the wave in flight before the first LDS store (columns 45..47 are padding
    // nothing reads).  Round 4: lane = 12 g + j takes dword j of rows g, g + 5, .., g + 40 -- its nine LDS destinations are the
    // constant offsets lane + 60 k, its nine row addresses one reflection + one multiply-add each (lanes 60..63 idle), where the
    // flat index of rounds 2-3 paid a division by 12 and the address arithmetic per dword.  Rows reflect per load.  A patch that
    // crosses the left or right edge (keypoints sit >= 19 pixels inside, the patch reaches 22: at most three columns) fetches
    // the 48 columns nearest to the edge into scratch and is then put in place, with the reflection, by an LDS-to-LDS byte copy.
    {
        const int xs_want = K.x - PR;
        const int xs = min(max(xs_want, 0), Lv.w - RAW_PITCH);
        const bool shifted = xs != xs_want;
        uint32_t* dst32 = shifted ? vp : reinterpret_cast<uint32_t*>(raw);
        const int g = (lane * 43) >> 9, j = lane - 12 * g;   // lane / 12, lane % 12 (exact for lane < 64)
        const uint8_t* p0 = img + xs + 4 * j;
        uint32_t v[9];
        if (K.y - PR >= 0 && K.y + PR < Lv.h) {   // (wave-uniform: no row of the patch leaves the level -- all but the keypoints of the outermost three rows)
            const uint8_t* pr = p0 + (size_t)(K.y - PR + g) * img_stride;
            const size_t step5 = (size_t)5 * img_stride;
#pragma unroll
            for (int k = 0; k < 9; ++k) {   // (the row addresses by addition: a 64-bit multiply-add per row is four passes of the vector ALU)
                v[k] = 0;
                if (lane < 60) __builtin_memcpy(&v[k], pr, 4);
                pr += step5;
            }
        } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            v[k] = 0;
            if (lane < 60) __builtin_memcpy(&v[k], p0 + (size_t)reflect101(K.y - PR + g + 5 * k, Lv.h) * img_stride, 4);
        }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (lane < 60) dst32[lane + 60 * k] = v[k];   // (row g + 5 k, dword j: position (12 g + j) + 60 k)
        if (shifted) {
            wave_lds_sync();
            const uint8_t* tmp = reinterpret_cast<const uint8_t*>(vp);
            for (int i = lane; i < PW * PW; i += 64) {
                const int ry = i / PW, rx = i - ry * PW;
                raw[ry * RAW_PITCH + rx] = tmp[ry * RAW_PITCH + reflect101(xs_want + rx, Lv.w) - xs];
            }
        }
    }
    wave_lds_sync();
    DPH(2);

    // IC_Angle moments over the circular patch (reference :77-104), exact int32
    // Row v of the disc is the columns |u| <= umax[|v|]; an item is one dword of a row (columns 4 + 4q .. 7 + 4q of the raw
    // patch, u = 4q - 18 + byte): bytes outside the disc are masked off, then sum(I) and sum(byte * I) are one v_dot4 each.
    int m10 = 0, m01 = 0;
    {
        const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(raw);
#pragma unroll
        for (int k = 0; k < (31 * 9 + 63) / 64; ++k) {
            const int i = lane + 64 * k;
            if (i < 31 * 9) {
                const int r = (i * 57) >> 9, q = i - r * 9;      // i / 9 (exact for i < 512)
                const uint32_t mask = tabs->ic.mask[i];
                const int uvp = tabs->ic.uv[i];
                const int u0 = (int)(short)(uvp & 0xffff), v = uvp >> 16;
                const uint32_t d = raw32[(PR - HALF_PATCH + r) * (RAW_PITCH / 4) + 1 + q] & mask;
                const int s1 = (int)__builtin_amdgcn_udot4(d, 0x01010101u, 0u, false), sw = (int)__builtin_amdgcn_udot4(d, 0x03020100u, 0u, false);
                m10 += u0 * s1 + sw;
                m01 += v * s1;
            }
        }
    }
    m10 = __builtin_amdgcn_readlane(wave_incl_scan(m10), 63);
    m01 = __builtin_amdgcn_readlane(wave_incl_scan(m01), 63);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    DPH(3);

    // 7x7 sigma-2 Gaussian, OpenCV 8-bit fixed point: taps [18,34,49,55,49,34,18] on both axes (App. A-4)
    // Horizontal pass (round 4): an item is FOUR output columns of TWO vertically adjacent rows (2m, 2m + 1).  The ten bytes a
    // row's four outputs need are three dwords of the raw row; output j's window is brought into place with two funnel shifts
    // and weighted with two v_dot4_u32_u8 (exact integer sums, <= 257 * 255 < 2^16).  The two rows' sums of a column go into ONE
    // dword -- row 2m in the low half, row 2m + 1 in the high half -- so that the vertical pass can weight a pair with one
    // v_dot2_u32_u16.  23 pairs x 10 column groups = 230 items: four rounds of the wave, one 16-byte LDS store per item.
    {
        const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(raw);
        // (second half of round 4) output j's seven taps are bytes j .. j + 6 of the row's three dwords: instead of moving the window to
        // the weights (two funnel shifts per output), the weights are moved to the window -- ten v_dot4 with constant weight words per
        // four outputs where there were eight behind six shifts.  Same products, same sums.
        constexpr uint32_t K0 = 18u, K1 = 34u, K2 = 49u, K3 = 55u;   // taps 0..3 (= 6..3 mirrored)
        constexpr uint32_t A0 = K0 | (K1 << 8) | (K2 << 16) | (K3 << 24), B0 = K2 | (K1 << 8) | (K0 << 16);                  // j = 0: d0, d1
        constexpr uint32_t A1 = (K0 << 8) | (K1 << 16) | (K2 << 24), B1 = K3 | (K2 << 8) | (K1 << 16) | (K0 << 24);          // j = 1: d0, d1
        constexpr uint32_t A2 = (K0 << 16) | (K1 << 24), B2 = K2 | (K3 << 8) | (K2 << 16) | (K1 << 24), C2 = K0;             // j = 2: d0, d1, d2
        constexpr uint32_t A3 = K0 << 24, B3 = K1 | (K2 << 8) | (K3 << 16) | (K2 << 24), C3 = K1 | (K0 << 8);                // j = 3: d0, d1, d2
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < 23 * (ROW_PITCH / 4)) {
                const int m = (i * 6554) >> 16, q = i - m * (ROW_PITCH / 4);   // i / 10 (exact for i < 16384); columns 4q .. 4q+3 (column 39 is padding)
                const int r0 = 2 * m, r1 = min(2 * m + 1, PW - 1);             // (pair 22's upper row does not exist: row 44 again, never used)
                uint32_t out[4];
                uint32_t hs[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t* row = raw32 + (u ? r1 : r0) * (RAW_PITCH / 4) + q;
                    const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];
                    hs[u][0] = __builtin_amdgcn_udot4(d1, B0, __builtin_amdgcn_udot4(d0, A0, 0u, false), false);
                    hs[u][1] = __builtin_amdgcn_udot4(d1, B1, __builtin_amdgcn_udot4(d0, A1, 0u, false), false);
                    hs[u][2] = __builtin_amdgcn_udot4(d2, C2, __builtin_amdgcn_udot4(d1, B2, __builtin_amdgcn_udot4(d0, A2, 0u, false), false), false);
                    hs[u][3] = __builtin_amdgcn_udot4(d2, C3, __builtin_amdgcn_udot4(d1, B3, __builtin_amdgcn_udot4(d0, A3, 0u, false), false), false);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = hs[0][j] | (hs[1][j] << 16);
                *reinterpret_cast<uint4*>(vp + m * ROW_PITCH + 4 * q) = make_uint4(out[0], out[1], out[2], out[3]);
            }
        }
    }
    wave_lds_sync();
    DPH(4);
    // Vertical pass (round 4): an item is one column and one of three row segments (rows 0..13, 14..27, 28..41 -- every segment
    // starts on an even row, rows 39..41 are computed and not stored): 117 items, two rounds of the wave.  A lane reads the ten
    // row pairs its fourteen outputs need once; an output is four v_dot2_u32_u16 -- (18,34) (49,55) (49,34) (18,0) over the pairs
    // from an even row, (0,18) (34,49) (55,49) (34,18) from an odd one, the rounding constant 32768 riding in the first
    // accumulator -- a shift and the clamp: 6 vector instructions where rounds 2-3 spent 10 (and 38 to unpack the sums).
    {
        using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
        auto dot2 = [](uint32_t pair, uint32_t w, uint32_t acc) {
            return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, w), acc, false);
        };
        constexpr uint32_t E0 = 18u | (34u << 16), E1 = 49u | (55u << 16), E2 = 49u | (34u << 16), E3 = 18u;
        constexpr uint32_t O0 = 18u << 16, O1 = 34u | (49u << 16), O2 = 55u | (49u << 16), O3 = 34u | (18u << 16);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int id = lane + 64 * k;
            if (id < 3 * BW) {
                const int seg = (id * 1681) >> 16, col = id - seg * BW;   // id / 39 (exact for id < 117)
                const uint32_t* src = vp + (7 * seg) * ROW_PITCH + col;   // first pair of the segment: rows 14 seg, 14 seg + 1
                uint32_t P[10];
#pragma unroll
                for (int t = 0; t < 10; ++t) P[t] = src[t * ROW_PITCH];
                uint8_t* dst = blur + (14 * seg) * ROW_PITCH + col;
#pragma unroll
                for (int o = 0; o < 14; ++o) {
                    const int t0 = o >> 1;
                    uint32_t acc;
                    if (o & 1) acc = dot2(P[t0 + 3], O3, dot2(P[t0 + 2], O2, dot2(P[t0 + 1], O1, dot2(P[t0], O0, 32768u))));
                    else acc = dot2(P[t0 + 3], E3, dot2(P[t0 + 2], E2, dot2(P[t0 + 1], E1, dot2(P[t0], E0, 32768u))));
                    const uint32_t val = min(acc, 0xffffffu) >> 16;   // == min(acc >> 16, 255): the clamp first, so that the byte is bits 16..23 of a register (one d16_hi store, no shift)
                    if (o < 11 || seg < 2) dst[o * ROW_PITCH] = (uint8_t)val;   // (rows 39..41 of the last segment do not exist)
                }
            }
        }
    }
    wave_lds_sync();
    DPH(5);

    // steered rBRIEF (reference :108-147): 4 tests per lane, lanes 2j / 2j+1 make byte j
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    det_sincos(angle * factorPI, a, b);
    const uint8_t* centre = &blur[BR_ * ROW_PITCH + BR_];
    int nib = 0;
#ifdef MORB_DESCRIBE_SELFCHECK
    float4 chk_q[4]; int chk_rc[4][4], chk_t[4][2];
#endif
#ifdef MORB_DESCRIBE_FENCE
    float4 qf[4];   // experiment: all four table quads loaded, then a full wait plus idle cycles in front of the first use
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[j] = *reinterpret_cast<const float4*>(tabs->pat.v[4 * lane + j]);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+v"(qf[0].x), "+v"(qf[0].y), "+v"(qf[0].z), "+v"(qf[0].w), "+v"(qf[1].x), "+v"(qf[1].y), "+v"(qf[1].z), "+v"(qf[1].w),
                 "+v"(qf[2].x), "+v"(qf[2].y), "+v"(qf[2].z), "+v"(qf[2].w), "+v"(qf[3].x), "+v"(qf[3].y), "+v"(qf[3].z), "+v"(qf[3].w) :: "memory");
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef MORB_DESCRIBE_FENCE
        const float4 q = qf[j];
#else
        const float4 q = *reinterpret_cast<const float4*>(tabs->pat.v[4 * lane + j]);
#endif
        const float x0 = q.x, y0 = q.y, x1 = q.z, y1 = q.w;
        const int r0 = __float2int_rn(x0 * b + y0 * a), c0 = __float2int_rn(x0 * a - y0 * b);
        const int r1 = __float2int_rn(x1 * b + y1 * a), c1 = __float2int_rn(x1 * a - y1 * b);
        const int t0 = centre[__mul24(r0, ROW_PITCH) + c0], t1 = centre[__mul24(r1, ROW_PITCH) + c1];   // (24-bit multiply-add: one full-rate instruction)
        nib |= (t0 < t1) << j;
#ifdef MORB_DESCRIBE_SELFCHECK
        chk_q[j] = q; chk_rc[j][0] = r0; chk_rc[j][1] = c0; chk_rc[j][2] = r1; chk_rc[j][3] = c1; chk_t[j][0] = t0; chk_t[j][1] = t1;
#endif
    }
#ifdef MORB_DESCRIBE_SELFCHECK
    // Debug build only (csrc/Makefile SELFCHECK=1): every lane does its four tests a second time, step by step, and counts where the
    // two evaluations part -- the table quad re-loaded past the CU's vector cache (class 0: the registers the tests used do not hold
    // what memory holds), the rotation re-done one scalar-float instruction at a time on the registers the tests used (class 1: same
    // operands, different result, or the packed instruction read its operands before the load had written them), the two bytes of the
    // blurred patch re-read (class 2), the test re-decided (class 3).  g_describe_check[0..3] count lanes x tests, [4] the keypoints
    // checked, [8..] hold the first few records.
    {
        auto f_mul = [](float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        auto f_add = [](float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        auto f_sub = [](float x, float y) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        if (lane == 0) atomicAdd(&g_describe_check[4], 1ull);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* src = tabs->pat.v[4 * lane + j];
            float m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float4 q = chk_q[j];
            const bool bad_load = __float_as_uint(m[0]) != __float_as_uint(q.x) || __float_as_uint(m[1]) != __float_as_uint(q.y) ||
                                  __float_as_uint(m[2]) != __float_as_uint(q.z) || __float_as_uint(m[3]) != __float_as_uint(q.w);
            const int r0 = __float2int_rn(f_add(f_mul(q.x, b), f_mul(q.y, a))), c0 = __float2int_rn(f_sub(f_mul(q.x, a), f_mul(q.y, b)));
            const int r1 = __float2int_rn(f_add(f_mul(q.z, b), f_mul(q.w, a))), c1 = __float2int_rn(f_sub(f_mul(q.z, a), f_mul(q.w, b)));
            const bool bad_arith = r0 != chk_rc[j][0] || c0 != chk_rc[j][1] || r1 != chk_rc[j][2] || c1 != chk_rc[j][3];
            const volatile uint8_t* vc = centre;
            const int t0 = vc[chk_rc[j][0] * ROW_PITCH + chk_rc[j][1]], t1 = vc[chk_rc[j][2] * ROW_PITCH + chk_rc[j][3]];
            const bool bad_lds = t0 != chk_t[j][0] || t1 != chk_t[j][1];
            const bool bad_bit = ((nib >> j) & 1) != (int)(chk_t[j][0] < chk_t[j][1]);
            if (bad_load) atomicAdd(&g_describe_check[0], 1ull);
            if (bad_arith) {
                atomicAdd(&g_describe_check[1], 1ull);
                // which of the test's four coordinates (r0 = x0 b + y0 a, c0 = x0 a - y0 b, r1, c1), per table quad j; and the wave quarter
                for (int c4 = 0; c4 < 4; ++c4) if ((c4 == 0 ? r0 : c4 == 1 ? c0 : c4 == 2 ? r1 : c1) != chk_rc[j][c4]) atomicAdd(&g_describe_check[8 + 16 * 14 + 4 * j + c4], 1ull);
                atomicAdd(&g_describe_check[8 + 16 * 14 + 16 + (lane >> 4)], 1ull);
            }
            if (bad_lds) atomicAdd(&g_describe_check[2], 1ull);
            if (bad_bit) atomicAdd(&g_describe_check[3], 1ull);
            if (bad_load || bad_arith || bad_lds || bad_bit) {
                const unsigned long long slot = atomicAdd(&g_describe_check[5], 1ull);
                if (slot < 16) {
                    unsigned long long* rec = g_describe_check + 8 + slot * 14;
                    rec[0] = (unsigned long long)ki << 32 | (unsigned)(lane << 8 | j << 4 | bad_load | bad_arith << 1 | bad_lds << 2 | bad_bit << 3);
                    rec[1] = (unsigned long long)__float_as_uint(q.x) << 32 | __float_as_uint(q.y);
                    rec[2] = (unsigned long long)__float_as_uint(q.z) << 32 | __float_as_uint(q.w);
                    rec[3] = (unsigned long long)__float_as_uint(m[0]) << 32 | __float_as_uint(m[1]);
                    rec[4] = (unsigned long long)__float_as_uint(m[2]) << 32 | __float_as_uint(m[3]);
                    rec[5] = (unsigned long long)__float_as_uint(a) << 32 | __float_as_uint(b);
                    rec[6] = (unsigned long long)(unsigned)chk_rc[j][0] << 32 | (unsigned)chk_rc[j][1];
                    rec[7] = (unsigned long long)(unsigned)chk_rc[j][2] << 32 | (unsigned)chk_rc[j][3];
                    rec[8] = (unsigned long long)(unsigned)r0 << 32 | (unsigned)c0;
                    rec[9] = (unsigned long long)(unsigned)r1 << 32 | (unsigned)c1;
                    rec[10] = (unsigned long long)(unsigned)chk_t[j][0] << 32 | (unsigned)chk_t[j][1];
                    rec[11] = (unsigned long long)(unsigned)t0 << 32 | (unsigned)t1;
                    rec[12] = wall_clock64();
                    rec[13] = (unsigned long long)blockIdx.x << 32 | (unsigned)K.camlevel;
                }
            }
        }
    }
#endif
    const int other = __shfl_xor(nib, 1);
    DPH(6);
    const int out_idx = K.resp_out & 0xffffff;
    // bytes sit in the even lanes (lane 2j = byte j): fold them into 8 dwords held by lanes 0, 8, .., 56
    unsigned int word = (unsigned int)(nib | (other << 4));
    word |= (unsigned int)__shfl_down((int)word, 2) << 8;
    word |= (unsigned int)__shfl_down((int)word, 4) << 16;
    const int g = mirror_base + out_idx;  // global index in camera-major order
    if ((lane & 7) == 0) {
        reinterpret_cast<uint32_t*>(desc_dst)[(size_t)out_idx * 8 + (lane >> 3)] = word;
        if (mir.desc) reinterpret_cast<uint32_t*>(mir.desc)[(size_t)g * 8 + (lane >> 3)] = word;
        if (to_sink) sink.desc[(size_t)g * 8 + (lane >> 3)] = word;
    }
    if (lane == 0) {
        orb_keypoint kp;
        const float fx = (float)K.x, fy = (float)K.y;
        kp.x = level ? fx * Lv.scale : fx;  // pt *= scale only for level != 0 (:1096-1103)
        kp.y = level ? fy * Lv.scale : fy;
        kp.size = Lv.patch_size;
        kp.angle = angle;
        kp.response = (float)((unsigned)K.resp_out >> 24);
        kp.octave = level;
        kp.class_id = -1;
        kps_dst[out_idx] = kp;
        if (mir.kps) mir.kps[g] = kp;
        if (to_sink) {
            // the per-feature half of the frame assembly: `_total` record, ComputeStereoFromRGBD, PosInGrid
            float ux = kp.x, uy = kp.y;  // Frame::UndistortKeyPoints
            if (sink.calib.k1 != 0.0f) morb_undistort_point(sink.calib, kp.x, kp.y, &ux, &uy);
            sink.x[g] = ux; sink.y[g] = uy; sink.oct[g] = level; sink.ang[g] = angle; sink.kps[g] = kp;
            if (sink.h_unx) { sink.h_unx[g] = ux; sink.h_uny[g] = uy; }
            float d = -1.f, u_r = -1.f;
            if (have_depth) {
                const float dv = depth_sample;
                if (dv > 0) { d = dv; u_r = ux - sink.mbf / dv; }  // depth at the distorted pixel, uRight from the undistorted x
            }
            sink.ur[g] = u_r; sink.depth[g] = d;
            if (sink.h_ur) { sink.h_ur[g] = u_r; sink.h_depth[g] = d; }
            const int px = (int)roundf((ux - sink.minX) * sink.invW), py = (int)roundf((uy - sink.minY) * sink.invH);
            sink.cell_of[g] = (px >= 0 && px < 64 && py >= 0 && py < 48) ? (cam * 64 + px) * 48 + py : -1;
        }
    }
    DPH(7);
}

// ------------------------------------------------------------------------------------------------ host tables
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

struct CamTables {
    orbx_params p;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
};

void build_cam_tables(const orbx_params& p, CamTables& T) {
    T.p = p;
    const int n = p.nlevels;
    const double sf = (double)p.scale_factor;  // the reference stores scaleFactor in a double member
    T.scale.assign(n, 1.0f); T.sigma2.assign(n, 1.0f); T.inv_scale.assign(n, 1.0f); T.inv_sigma2.assign(n, 1.0f);
    for (int i = 1; i < n; ++i) {
        T.scale[i] = (float)((double)T.scale[i - 1] * sf);
        T.sigma2[i] = T.scale[i] * T.scale[i];
    }
    for (int i = 0; i < n; ++i) { T.inv_scale[i] = 1.0f / T.scale[i]; T.inv_sigma2[i] = 1.0f / T.sigma2[i]; }
    T.quota.assign(n, 0);
    const float factor = (float)(1.0 / sf);
    float desired = (float)p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)n));
    int sum = 0;
    for (int l = 0; l < n - 1; ++l) {
        T.quota[l] = cv_round(desired);
        sum += T.quota[l];
        desired *= factor;
    }
    T.quota[n - 1] = std::max(p.nfeatures - sum, 0);
}

void compute_umax(int* umax) {
    int v, v0, vmax = cv_floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// OpenCV resize(INTER_LINEAR) coefficient tables for one axis pair (imgwarp.cpp, 8-bit fixed point path)
void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int2>& xt, std::vector<int4>& yt) {
    const int ONE = 1 << COEF_BITS;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    auto sat_short = [](int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; };
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int a0 = sat_short(cv_round((1.f - fx) * ONE)), a1 = sat_short(cv_round(fx * ONE));
        const int sx1 = std::min(sx + 1, sw - 1);
        int2 e;
        e.x = (sx & 0xffff) | (sx1 << 16);
        e.y = (a0 & 0xffff) | (a1 << 16);
        xt.push_back(e);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        int4 e;
        e.x = std::min(std::max(sy, 0), sh - 1);      // each tap's row is clipped, coefficients stay
        e.y = std::min(std::max(sy + 1, 0), sh - 1);
        e.z = sat_short(cv_round((1.f - fy) * ONE));
        e.w = sat_short(cv_round(fy * ONE));
        yt.push_back(e);
    }
}

// k_pyramid_tiled4's view of the x table: per group of four destination columns {c, sel0, sel1, sel2} {sel3, alpha0..2} {alpha3}, c = the
// first tap's source column, sel_j = (left tap - c) | 0x0c << 8 | (right tap - c) << 16 | 0x0c << 24 (v_perm_b32 selector over the
// row's eight bytes from c on: tap | 0 | tap | 0), alpha_j = the column's coefficient pair.  Columns past the level's width repeat
// its last one (they land in the row pitch's padding).
void build_resize_groups(const std::vector<int2>& xt, int xtab_off, int dw, std::vector<int4>& xg) {
    for (int x4 = 0; x4 < dw; x4 += 4) {
        const int c = xt[xtab_off + x4].x & 0xffff;
        uint32_t sel[4], al[4];
        for (int j = 0; j < 4; ++j) {
            const int2 e = xt[xtab_off + std::min(x4 + j, dw - 1)];
            const uint32_t l = (uint32_t)((e.x & 0xffff) - c) & 7u, r = (uint32_t)(((unsigned)e.x >> 16) - c) & 7u;
            sel[j] = l | (0x0cu << 8) | (r << 16) | (0x0cu << 24);
            al[j] = (uint32_t)e.y;
        }
        xg.push_back(make_int4(c, (int)sel[0], (int)sel[1], (int)sel[2]));
        xg.push_back(make_int4((int)sel[3], (int)al[0], (int)al[1], (int)al[2]));
        xg.push_back(make_int4((int)al[3], 0, 0, 0));
    }
}

}  // namespace

// Small persistent worker pool for the host quadtree: the (camera, level) problems of one call are independent.
// Workers spin for a short while after a job (the steady-state gap between frames is ~1 ms) before sleeping.
class TaskPool {
    struct Job {  // one per run(): stale workers can only ever touch their own (finished) job
        std::function<void(int)> fn;
        int n = 0;
        std::atomic<int> next{0}, done{0};
    };
public:
    explicit TaskPool(int n_workers) {
        for (int i = 0; i < n_workers; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    template <typename F>
    void run(int n_tasks, F&& fn) {
        if (workers_.empty() || n_tasks <= 1) { for (int i = 0; i < n_tasks; ++i) fn(i); return; }
        auto job = std::make_shared<Job>();
        job->fn = [&fn](int i) { fn(i); };
        job->n = n_tasks;
        { std::lock_guard<std::mutex> lk(mu_); job_ = job; gen_.fetch_add(1); }
        cv_.notify_all();
        work(*job);
        while (job->done.load(std::memory_order_acquire) < n_tasks) std::this_thread::yield();
        // `fn` may go out of scope now: every task has finished, late workers see next >= n and never call job->fn
    }
private:
    static void work(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1);
            if (i >= j.n) break;
            j.fn(i);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();  // spin ~2 ms for the next job, then block
            while (gen_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return gen_.load() != seen; });
                    break;
                }
            }
            std::shared_ptr<Job> job;
            {
                std::lock_guard<std::mutex> lk(mu_);
                seen = gen_.load();
                if (stop_) return;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<unsigned> gen_{0};
    std::shared_ptr<Job> job_;
    bool stop_ = false;
};

struct OctreeTask {
    int cam, level, n;
    std::vector<int> cx, cy, cr, selected;
};

// ================================================================================================ C ABI
// tile width / height / split level of the large-rig pyramid plan (orbx_debug_pyramid_plan: the tests of the plan's geometry change it
// before they create a handle; the product never does)
static int g_t4_plan[3] = {128, 64, 3};

struct orbx_extractor {
    int device = 0, n_cams = 0, max_w = 0, max_h = 0, max_levels = 0;
    hipStream_t stream = nullptr;
    bool stream_owned = true;            // false: borrowed (orbx_adopt_stream)
    std::vector<CamTables> cams;
    std::vector<int> cur_w, cur_h;   // size of the resident image per camera (0 = none)
    FrameSink sink;                  // orbf_step: merged-frame destination of the describe kernel (x == nullptr: off)
    IngestArgs ingest;               // device-resident sources to copy into level 0 at the start of the next run
    bool ingest_pending = false;
    bool tables_dirty = true;

    // geometry (host copies)
    std::vector<LevelInfo> levels;   // [cam * max_levels + level]
    std::vector<int2> cell_map;
    int cell_h_max = 1, cell_px_max = 1;   // tallest cell / most pixels in a cell over all cameras and levels (k_fast_cells' LDS)
    int total_cells = 0;
    size_t total_slots = 0;
    size_t cam_pitch = 0;            // bytes of pyramid memory per camera
    std::vector<int> out_cap;        // per camera keypoint capacity of the output buffers

    // device memory
    DevBuf<uint8_t> d_pyr;
    DevBuf<LevelInfo> d_levels;
    DevBuf<int2> d_cell_map, d_xtab;
    DevBuf<int4> d_xgrp;              // k_pyramid_tiled4's per-group view of the x table
    DevBuf<uint8_t> d_desc_tabs;      // k_describe's DescribeTables (IC_Angle items + float test locations)
    // the tiled whole-pyramid launch (k_pyramid_tiled): spans per (camera, level, tile column / row), tiles per camera, LDS need
    DevBuf<int4> d_pyr_sx, d_pyr_sy;
    int pyr_tx_max = 0, pyr_ty_max = 0, pyr_lds = 0, pyr_tile = 0, pyr_tab_cap = 0;
    short pyr_tx[64] = {}, pyr_ty[64] = {};
    bool ingest_host = false;         // a pending ingest source lives in host memory (read across PCIe)
    bool tiled_ok = false;            // ... and this geometry fits it (LDS, halo, level count) and is small enough to prefer it
    bool v4_ok = false;               // level steps <= 1.6: the four-pixels-per-lane arithmetic of k_pyramid_tiled4 applies (taps of four neighbours within 8 bytes)
    bool generic_chain = false;       // MORB_PYR_CHAIN=2: neither tile form, one k_resize launch per level
    // large rigs (round 5): the pyramid as TWO tile launches with four pixels per lane (k_pyramid_tiled4): levels 1..m below level 0
    // and levels m+1.. below level m
    struct TilePlan {
        int base = 0, last = 0, tw = 0, th = 0, halo_x = 16, halo_y = 16, tx_max = 0, ty_max = 0, gcap = 0, rcap = 0, lds = 0, threads = 256;
        short tx[64] = {}, ty[64] = {};
        DevBuf<int4> d_sx, d_sy;
        std::vector<int4> sx, sy;     // (host copies while the geometry is being built)
        bool ok = false;
    } tp[2];
    bool tiled4 = false;              // this geometry takes the two tile launches
    DevBuf<int4> d_ytab;
    DevBuf<int> d_cell_cnt, d_cell_off;
    DevBuf<uint32_t> d_cell_items;
    DevBuf<SelKp> d_sel, d_sel_oct;
    DevBuf<uint32_t> d_cand_dev;
    DevBuf<int> d_level_cnt_dev, d_sel_cnt, d_oct_status, d_n_out;
    bool pinned_ingest = false;       // page-locked host images are read by k_ingest directly (orbx_set_pinned_ingest) instead of hipMemcpy2DAsync
    // level 0 in place (orbx_set_inplace_level0): l0_on = the geometry takes it (the large-rig tile launches) and the caller opted in;
    // l0_active = the device table holds pointers of the last run's images; l0_host = what it holds (inspection hook)
    bool l0_optin = false, l0_on = false, l0_active = false;
    DevBuf<L0Src> d_l0;
    std::vector<L0Src> l0_host;
    std::vector<uint8_t> uploaded;    // cameras handed an image (any kind, also an empty one) since the last run
    int oct_max_keys = OCT_RK;        // candidates per (camera, level) the device quadtree takes (MORB_OCT_MAX_KEYS lowers it: tests of the fallback)
    int last_path = 0;                // inspection: 0 device quadtree, 2 host quadtree
    DevBuf<unsigned short> d_slot_blk;
    int total_sel_slots = 0;
    int* h_oct = nullptr;            // pinned, mapped: [0..n_cams) n_out, [n_cams] status
    bool device_octree = true;
    bool cand_valid = false;         // h_cand / h_level_cnt hold the last run's candidates (k_compact ran)
    // (in-flight bookkeeping: see `inflight` below), orbx_finish not yet called
    std::chrono::steady_clock::time_point t_begin_async;
    // MORB_EXTRACT_TIMELINE=1: where the synchronous orbx_extract spends its host time (sums, printed by orbx_destroy)
    // up to two asynchronous runs may be in flight (the second one is the next timestep's, enqueued while the first one's
    // results are being matched): run r uses slot r & 1 of the count mirrors and of the completion events
    struct ChainGraph {  // captured kernel chain of one count slot (see orbx_run_impl)
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        int epoch = -1; orb_keypoint* mirror_kps = nullptr; uint8_t* mirror_desc = nullptr; FrameSink sink; int tail_tag = 0;
        void destroy() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            exec = nullptr; graph = nullptr; epoch = -1;
        }
    };
    static constexpr int CHAIN_WAYS = 4;
    ChainGraph chain[2][CHAIN_WAYS];
    int chain_next[2] = {0, 0};      // replacement cursor per slot
    bool use_graph = true;           // MORB_CHAIN_GRAPH=0 keeps plain launches
    bool defer_done = false;      // orbx_set_defer_done: the caller records ev_done of the next asynchronous runs itself (orbx_record_done)
    bool graph_next_run = true;  // orbx_set_chain_graph: plain launches for the next run (its consumer follows on the same stream)
    orbx_tail_fn tail_fn = nullptr; void* tail_user = nullptr; int tail_tag = 0;  // orbx_set_chain_tail
    int geom_epoch = 0;              // bumped by every rebuild_geometry
    int inflight = 0; unsigned run_seq = 0;
    bool prof_valid[2] = {false, false};  // the stage events were recorded for the run in this slot
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    hipEvent_t ev_foreign = nullptr;   // orbx_wait_for_stream
    int* d_h_oct = nullptr;          // device alias of h_oct
    std::vector<DevBuf<orb_keypoint>> d_kps;
    std::vector<DevBuf<uint8_t>> d_desc;
    std::vector<orb_keypoint*> out_kps;   // active output pointers (internal or bound)
    std::vector<uint8_t*> out_desc;
    std::vector<int> out_cap_active;
    DevBuf<orb_keypoint*> d_out_kps;
    DevBuf<uint8_t*> d_out_desc;
    bool out_ptrs_dirty = true;
    orb_keypoint* mirror_kps = nullptr; uint8_t* mirror_desc = nullptr; int mirror_cap = 0;
    // orbx_extract (host buffers in, host buffers out): pageable images reach HBM through host-written staging (two slots:
    // up to two runs may be in flight) + the ingest kernel, results come back through the handle's own pinned mirror
    morb::StageBuf stage_img[2];
    morb::PinnedBuf<orb_keypoint> own_mirror_kps; morb::PinnedBuf<uint8_t> own_mirror_desc;

    // pinned host (device-visible) buffers
    uint32_t* h_cand = nullptr; size_t h_cand_cap = 0;
    int* h_level_cnt = nullptr;
    SelKp* h_sel = nullptr; size_t h_sel_cap = 0;

    std::vector<int> n_out;          // keypoints per camera of the last run
    std::vector<int> level_cnt_last; // candidates per (cam, level) of the last run
    // host quadtree: one task per (camera, level), run on a small worker pool
    std::vector<OctreeTask> tasks;
    std::unique_ptr<TaskPool> pool;

    bool profiling = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float stage_us[6] = {0, 0, 0, 0, 0, 0};
};

static int rebuild_geometry(orbx_extractor* ex) {
    const int ML = ex->max_levels;
    ++ex->geom_epoch;  // captured launch chains carry the old geometry
    ex->levels.assign((size_t)ex->n_cams * ML, LevelInfo{});
    ex->cell_map.clear();
    ex->cell_h_max = 1; ex->cell_px_max = 1;
    std::vector<int2> xt;
    std::vector<int4> yt, xg;
    int cell_base = 0;
    size_t slot_base = 0, cand_base = 0;
    for (int c = 0; c < ex->n_cams; ++c) {
        const CamTables& T = ex->cams[c];
        const int W = ex->cur_w[c], H = ex->cur_h[c];
        if (W == 0 || H == 0) continue;
        int pyr_off = 0, pw = 0, ph = 0;
        for (int l = 0; l < T.p.nlevels; ++l) {
            LevelInfo& Lv = ex->levels[(size_t)c * ML + l];
            const float s = T.inv_scale[l];
            Lv.w = cv_round((double)((float)W * s));  // reference :1113-1114
            Lv.h = cv_round((double)((float)H * s));
            Lv.stride = align_up(Lv.w, 64);
            Lv.pyr_off = pyr_off;
            pyr_off += align_up(Lv.stride * Lv.h, 256);
            Lv.ini_th = T.p.ini_th_fast; Lv.min_th = T.p.min_th_fast;
            Lv.scale = T.scale[l];
            Lv.patch_size = (float)(int)(PATCH_SIZE * T.scale[l]);  // :838
            // cell grid, :774-788
            const float width = (float)(Lv.w - 2 * MIN_BORDER), height = (float)(Lv.h - 2 * MIN_BORDER);
            const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
            if (nCols < 1 || nRows < 1) {
                morb::set_error("camera %d level %d is %dx%d: too small for the 30-px detection grid", c, l, Lv.w, Lv.h);
                return ORB_E_ARG;
            }
            Lv.n_cols = nCols; Lv.n_rows = nRows;
            Lv.w_cell = (int)std::ceil(width / nCols);
            Lv.h_cell = (int)std::ceil(height / nRows);
            if (Lv.w_cell > CELL_MAX || Lv.h_cell > CELL_MAX || Lv.w > 4096 + 2 * MIN_BORDER || Lv.h > 4096 + 2 * MIN_BORDER) {
                morb::set_error("camera %d level %d (%dx%d): cell %dx%d or level size outside the supported range", c, l,
                                Lv.w, Lv.h, Lv.w_cell, Lv.h_cell);
                return ORB_E_ARG;
            }
            ex->cell_h_max = std::max(ex->cell_h_max, Lv.h_cell); ex->cell_px_max = std::max(ex->cell_px_max, Lv.w_cell * Lv.h_cell);
            Lv.quota = T.quota[l];
            Lv.cell_base = cell_base;
            Lv.slot_cap = ((Lv.w_cell + 1) / 2) * ((Lv.h_cell + 1) / 2);  // strict 8-neighbour maxima cannot be denser
            Lv.slot_base = (int)slot_base;
            Lv.cand_base = (int)cand_base;
            const int ncell = nCols * nRows;
            for (int k = 0; k < ncell; ++k)   // {block | cell column << 12 | cell row << 22, local cell index}
                ex->cell_map.push_back(make_int2((c * ML + l) | ((k % Lv.n_cols) << 12) | ((k / Lv.n_cols) << 22), k));
            cell_base += ncell;
            slot_base += (size_t)ncell * Lv.slot_cap;
            cand_base += (size_t)ncell * Lv.slot_cap;
            if (l > 0) {
                Lv.xtab_off = (int)xt.size(); Lv.ytab_off = (int)yt.size(); Lv.xgrp_off = (int)(xg.size() / 3);
                build_resize_tables(pw, ph, Lv.w, Lv.h, xt, yt);
                build_resize_groups(xt, Lv.xtab_off, Lv.w, xg);
            }
            pw = Lv.w; ph = Lv.h;
        }
        if ((size_t)pyr_off > ex->cam_pitch) { morb::set_error("image larger than the size given at create"); return ORB_E_ARG; }
    }
    // ---- spans of the tiled whole-pyramid launch (k_pyramid_tiled): tile k of a camera owns, on level l, the destination
    // columns whose left source tap lies in what it owns on level l - 1 ([k T, (k + 1) T) on level 0), and needs them plus the
    // taps of everything it needs one level down; rows alike.  Small rigs take 32-pixel tiles (more workgroups: the launch is
    // one round of short workgroups), large ones 64 (less of the halo recomputed).
    std::vector<int4> psx, psy;
    {
        long long px0 = 0;
        for (int c = 0; c < ex->n_cams; ++c) px0 += (long long)ex->cur_w[c] * ex->cur_h[c];
        const int T = px0 <= 1500000 ? 32 : 64;
        ex->pyr_tile = T; ex->pyr_tx_max = 1; ex->pyr_ty_max = 1; ex->pyr_lds = 0; ex->pyr_tab_cap = 0;
        bool halo_ok = true;
        for (int c = 0; c < ex->n_cams; ++c) {
            ex->pyr_tx[c] = (short)((ex->cur_w[c] + T - 1) / T); ex->pyr_ty[c] = (short)((ex->cur_h[c] + T - 1) / T);
            ex->pyr_tx_max = std::max<int>(ex->pyr_tx_max, ex->pyr_tx[c]); ex->pyr_ty_max = std::max<int>(ex->pyr_ty_max, ex->pyr_ty[c]);
        }
        psx.assign((size_t)ex->n_cams * ML * ex->pyr_tx_max, make_int4(0, 0, 0, 0));
        psy.assign((size_t)ex->n_cams * ML * ex->pyr_ty_max, make_int4(0, 0, 0, 0));
        // one axis: dims[l], first / last source index of destination d on level l (l >= 1)
        auto axis = [&](int c, int ntiles, int tmax, bool is_x, std::vector<int4>& out) {
            const int NL = ex->cams[c].p.nlevels;
            auto dim = [&](int l) { const LevelInfo& Lv = ex->levels[(size_t)c * ML + l]; return is_x ? Lv.w : Lv.h; };
            auto s0 = [&](int l, int d) { const LevelInfo& Lv = ex->levels[(size_t)c * ML + l]; return is_x ? (xt[Lv.xtab_off + d].x & 0xffff) : yt[Lv.ytab_off + d].x; };
            auto s1 = [&](int l, int d) { const LevelInfo& Lv = ex->levels[(size_t)c * ML + l]; return is_x ? (int)((unsigned)xt[Lv.xtab_off + d].x >> 16) : yt[Lv.ytab_off + d].y; };
            std::vector<std::vector<int> > b(NL, std::vector<int>(ntiles + 1, 0));
            for (int k = 0; k <= ntiles; ++k) b[0][k] = std::min(k * T, dim(0));
            for (int l = 1; l < NL; ++l) {
                int d = 0;
                for (int k = 0; k <= ntiles; ++k) {   // first destination whose left tap is not in front of the boundary
                    while (d < dim(l) && s0(l, d) < b[l - 1][k]) ++d;
                    b[l][k] = k == ntiles ? dim(l) : d;
                }
            }
            for (int k = 0; k < ntiles; ++k) {
                int n1 = b[NL - 1][k + 1];
                for (int l = NL - 1; l >= 0; --l) {
                    const int nn = n1 - b[l][k];   // .w: 2^20 / needed extent, rounded up (index / extent by multiplication in the kernel)
                    out[((size_t)c * ML + l) * tmax + k] = make_int4(b[l][k], b[l][k + 1], n1, nn > 0 ? (int)(((1u << 20) + nn - 1) / nn) : 0);
                    if (l > 0) n1 = std::max(b[l - 1][k + 1], n1 > b[l][k] ? s1(l, n1 - 1) + 1 : b[l - 1][k]);
                }
            }
        };
        for (int c = 0; c < ex->n_cams; ++c) {
            if (ex->cur_w[c] == 0 || ex->cur_h[c] == 0) { ex->pyr_tx[c] = ex->pyr_ty[c] = 0; continue; }
            axis(c, ex->pyr_tx[c], ex->pyr_tx_max, true, psx);
            axis(c, ex->pyr_ty[c], ex->pyr_ty_max, false, psy);
            for (int ky = 0; ky < ex->pyr_ty[c]; ++ky)
                for (int kx = 0; kx < ex->pyr_tx[c]; ++kx) {
                    int bytes = 0, tx_n = 0, ty_n = 0;
                    for (int l = 0; l < ex->cams[c].p.nlevels; ++l) {
                        const int4 X = psx[((size_t)c * ML + l) * ex->pyr_tx_max + kx], Y = psy[((size_t)c * ML + l) * ex->pyr_ty_max + ky];
                        if (l == 0) {   // the kernel stages a fixed (T + halo)^2 block of level 0: what the tile needs must lie inside it
                            bytes += (T + PYR_HALO) * (T + PYR_HALO);
                            if (X.z - X.x > T + PYR_HALO || Y.z - Y.x > T + PYR_HALO) halo_ok = false;
                            continue;
                        }
                        bytes += ((std::max(X.z - X.x, 0) + 3) & ~3) * std::max(Y.z - Y.x, 0);
                        tx_n += std::max(X.z - X.x, 0); ty_n += std::max(Y.z - Y.x, 0);
                    }
                    ex->pyr_lds = std::max(ex->pyr_lds, bytes);
                    ex->pyr_tab_cap = std::max(ex->pyr_tab_cap, std::max(tx_n, ty_n));
                }
        }
        ex->pyr_tab_cap = (ex->pyr_tab_cap + 3) & ~3;
        ex->pyr_lds += ex->pyr_tab_cap * 24;   // the table entries of a tile's columns (int2) and rows (int4) in front of the regions
        ex->tiled_ok = !(ex->pyr_lds > 60 * 1024 || ML > PYR_MAX_LEVELS || !halo_ok || ex->max_w > 32767 || ex->max_h > 32767);   // (cannot happen for tiles of 64 and scale factors >= 1.05)
        // Large rigs take two tile launches with four pixels per lane instead (k_pyramid_tiled4 below): the one-launch tile kernel is a
        // latency design and costs three times the instructions per pixel.  Their arithmetic needs level steps of at most 1.6 (the taps of
        // four neighbouring pixels within 8 bytes).  MORB_PYR_CHAIN: 1 = the large-rig form at every size, 2 = neither tile form (the
        // generic one-launch-per-level chain, k_resize: what odd parameter sets fall back to), 0 = never the large-rig form; default: the
        // large-rig form above 4 M level-0 pixels over all cameras (8 x 1080p yes, 2 x 1280x720 no).
        ex->v4_ok = true;
        for (int c = 0; c < ex->n_cams && ex->v4_ok; ++c)
            for (int l = 1; l < ex->cams[c].p.nlevels; ++l) {
                const LevelInfo &Ls = ex->levels[(size_t)c * ML + l - 1], &Ld = ex->levels[(size_t)c * ML + l];
                if (Ld.w > 0 && (long long)Ls.w * 10 > (long long)Ld.w * 16) ex->v4_ok = false;
            }
        static const int chain_env = [] { const char* e = getenv("MORB_PYR_CHAIN"); return e ? atoi(e) : -1; }();
        ex->generic_chain = chain_env == 2;
        const bool prefer_large = ex->v4_ok && (chain_env == 1 || (chain_env < 0 && (long long)px0 > 4000000ll));
        if (prefer_large || ex->generic_chain) ex->tiled_ok = false;
    }
    // ---- large rigs: the two tile launches of k_pyramid_tiled4 (TilePlan).  Along x a tile owns whole groups of four destination
    // columns (the group whose first tap lies in what the tile owns one level up), along y rows by their upper tap; what it needs
    // beyond that is the taps of everything it needs one level down (rounded up to whole groups below the base).
    ex->tiled4 = false;
    {
        // tile 128 x 64 below level 0 and below the split level 3, 256 threads (orbx_debug_pyramid_plan: other tiles / another split, for
        // the tests of the plan's geometry)
        const int split_env = g_t4_plan[2], tw_env = g_t4_plan[0], th_env = g_t4_plan[1], tw1_env = 0, th1_env = 0, nt0_env = 256, nt1_env = 256;
        int nl_max = 0;
        for (int c = 0; c < ex->n_cams; ++c) nl_max = std::max(nl_max, ex->cams[c].p.nlevels);
        const bool want = ex->v4_ok && !ex->tiled_ok && !ex->generic_chain && nl_max >= 2 && ML <= PYR_MAX_LEVELS &&
                          tw_env >= 16 && tw_env % 4 == 0 && th_env >= 8 && ex->max_w <= 32767 && ex->max_h <= 32767;
        const int split = std::min(std::max(split_env, 1), nl_max - 1);
        bool all_ok = want;
        for (int pi = 0; pi < 2 && all_ok; ++pi) {
            orbx_extractor::TilePlan& P = ex->tp[pi];
            P.ok = false;
            P.base = pi == 0 ? 0 : split; P.last = pi == 0 ? split : nl_max - 1;
            if (P.last <= P.base) { P.ok = true; P.tx_max = P.ty_max = 0; continue; }   // (nothing left for the second launch)
            P.tw = pi == 1 && tw1_env >= 16 && tw1_env % 4 == 0 ? tw1_env : tw_env; P.th = pi == 1 && th1_env >= 8 ? th1_env : th_env; P.halo_x = 16; P.halo_y = 16;
            P.threads = (pi == 0 ? nt0_env : nt1_env) == 512 ? 512 : 256;
            P.tx_max = P.ty_max = 1; P.gcap = P.rcap = 0; P.lds = 0;
            for (int c = 0; c < ex->n_cams; ++c) {
                const LevelInfo& Lb = ex->levels[(size_t)c * ML + P.base];
                P.tx[c] = (short)(Lb.w > 0 ? (Lb.w + P.tw - 1) / P.tw : 0); P.ty[c] = (short)(Lb.w > 0 ? (Lb.h + P.th - 1) / P.th : 0);
                P.tx_max = std::max<int>(P.tx_max, P.tx[c]); P.ty_max = std::max<int>(P.ty_max, P.ty[c]);
            }
            P.sx.assign((size_t)ex->n_cams * ML * P.tx_max, make_int4(0, 0, 0, 0));
            P.sy.assign((size_t)ex->n_cams * ML * P.ty_max, make_int4(0, 0, 0, 0));
            bool ok = true;
            int need_x = P.tw, need_y = P.th;   // what a tile needs of the base level, at most (the staged block: tile + halo)
            for (int c = 0; c < ex->n_cams && ok; ++c) {
                if (P.tx[c] == 0) continue;
                const int NL = std::min(ex->cams[c].p.nlevels - 1, P.last);   // last level this camera has inside the plan
                auto LV = [&](int l) -> const LevelInfo& { return ex->levels[(size_t)c * ML + l]; };
                // x: boundaries in pixels, multiples of four below the base
                {
                    const int nt = P.tx[c];
                    std::vector<std::vector<int> > b(ML, std::vector<int>(nt + 1, 0));
                    for (int k = 0; k <= nt; ++k) b[P.base][k] = std::min(k * P.tw, LV(P.base).w);
                    for (int l = P.base + 1; l <= NL; ++l) {
                        const int G = (LV(l).w + 3) / 4;
                        int g = 0;
                        for (int k = 0; k <= nt; ++k) {
                            while (g < G && (xt[LV(l).xtab_off + 4 * g].x & 0xffff) < b[l - 1][k]) ++g;
                            b[l][k] = k == nt ? 4 * G : 4 * g;
                        }
                    }
                    for (int k = 0; k < nt; ++k) {
                        int n1 = b[NL][k + 1];
                        for (int l = NL; l >= P.base; --l) {
                            const int ngn = l > P.base ? std::max(n1 - b[l][k], 0) / 4 : 0;
                            P.sx[((size_t)c * ML + l) * P.tx_max + k] = make_int4(b[l][k], b[l][k + 1], n1, ngn > 0 ? (int)(((1u << 20) + ngn - 1) / ngn) : 0);
                            if (l == P.base) { need_x = std::max(need_x, n1 - b[l][k]); break; }
                            // what level l - 1 must hold: its own run, and the right tap of the last column this level needs (padding
                            // columns repeat the level's last one)
                            int up = b[l - 1][k + 1];
                            if (n1 > b[l][k]) up = std::max(up, (int)((unsigned)xt[LV(l).xtab_off + std::min(n1 - 1, LV(l).w - 1)].x >> 16) + 1);
                            n1 = l - 1 > P.base ? std::min((up + 3) & ~3, 4 * ((LV(l - 1).w + 3) / 4)) : up;
                        }
                    }
                }
                // y: rows
                {
                    const int nt = P.ty[c];
                    std::vector<std::vector<int> > b(ML, std::vector<int>(nt + 1, 0));
                    for (int k = 0; k <= nt; ++k) b[P.base][k] = std::min(k * P.th, LV(P.base).h);
                    for (int l = P.base + 1; l <= NL; ++l) {
                        int d = 0;
                        for (int k = 0; k <= nt; ++k) {
                            while (d < LV(l).h && yt[LV(l).ytab_off + d].x < b[l - 1][k]) ++d;
                            b[l][k] = k == nt ? LV(l).h : d;
                        }
                    }
                    for (int k = 0; k < nt; ++k) {
                        int n1 = b[NL][k + 1];
                        for (int l = NL; l >= P.base; --l) {
                            P.sy[((size_t)c * ML + l) * P.ty_max + k] = make_int4(b[l][k], b[l][k + 1], n1, 0);
                            if (l == P.base) { need_y = std::max(need_y, n1 - b[l][k]); break; }
                            int up = b[l - 1][k + 1];
                            if (n1 > b[l][k]) up = std::max(up, yt[LV(l).ytab_off + n1 - 1].y + 1);
                            n1 = up;
                        }
                    }
                }
            }
            P.halo_x = ((need_x - P.tw + 3) & ~3); P.halo_y = need_y - P.th;
            // what a workgroup holds in LDS: the staged base block, the regions of its levels, the table entries of their groups and rows
            for (int c = 0; c < ex->n_cams && ok; ++c) {
                if (P.tx[c] == 0) continue;
                const int NL = std::min(ex->cams[c].p.nlevels - 1, P.last);
                for (int ky = 0; ky < P.ty[c]; ++ky)
                    for (int kx = 0; kx < P.tx[c]; ++kx) {
                        int bytes = (P.tw + P.halo_x) * (P.th + P.halo_y), ng = 0, nr = 0;
                        for (int l = P.base + 1; l <= NL; ++l) {
                            const int4 X = P.sx[((size_t)c * ML + l) * P.tx_max + kx], Y = P.sy[((size_t)c * ML + l) * P.ty_max + ky];
                            const int w = std::max(X.z - X.x, 0), h = std::max(Y.z - Y.x, 0);
                            if ((long long)(w / 4) * h * (w / 4) >= (1 << 20)) ok = false;   // (index split by multiplication: tasks x groups < 2^20)
                            bytes += w * h; ng += w / 4; nr += h;
                        }
                        P.lds = std::max(P.lds, bytes); P.gcap = std::max(P.gcap, ng); P.rcap = std::max(P.rcap, nr);
                    }
            }
            P.gcap = std::max(P.gcap, 1); P.rcap = std::max(P.rcap, 1);
            P.lds += P.gcap * 48 + P.rcap * 16 + 32;
            if (P.lds > 64 * 1024 || (P.tw + P.halo_x) / 4 * (P.th + P.halo_y) > T4_MAX_DW * P.threads || (P.halo_x & 3)) ok = false;
            P.ok = ok;
            all_ok = all_ok && ok;
        }
        ex->tiled4 = all_ok;
    }
    {
        // level 0 is read where the caller's device image lies (orbx_set_inplace_level0) by the large-rig tile launches only: they take
        // the {pointer, pitch} table; the one-launch tile kernel of small rigs and the generic chain read the pyramid buffer
        ex->l0_on = ex->l0_optin && ex->tiled4;
    }
    if (slot_base > (size_t)INT32_MAX) { morb::set_error("candidate slot space exceeds 2^31 entries"); return ORB_E_ARG; }
    ex->total_cells = cell_base;
    ex->total_slots = slot_base;
    // slotted output list of the device quadtree: quota + 4 slots per (camera, level)
    std::vector<unsigned short> slot_blk;
    for (size_t b = 0; b < ex->levels.size(); ++b) {
        LevelInfo& Lv = ex->levels[b];
        Lv.sel_base = (int)slot_blk.size();
        if (Lv.w == 0) continue;
        for (int k = 0; k < Lv.quota + 4; ++k) slot_blk.push_back((unsigned short)b);
    }
    ex->total_sel_slots = (int)slot_blk.size();
    int rc;
    if ((rc = ex->d_levels.reserve(ex->levels.size())) || (rc = ex->d_cell_map.reserve(std::max<size_t>(ex->cell_map.size(), 1))) ||
        (rc = ex->d_xtab.reserve(std::max<size_t>(xt.size(), 1))) || (rc = ex->d_ytab.reserve(std::max<size_t>(yt.size(), 1))) ||
        (rc = ex->d_xgrp.reserve(std::max<size_t>(xg.size(), 1))) ||
        (rc = ex->d_pyr_sx.reserve(std::max<size_t>(psx.size(), 1))) || (rc = ex->d_pyr_sy.reserve(std::max<size_t>(psy.size(), 1))) ||
        (rc = ex->d_cell_cnt.reserve(std::max(cell_base, 1))) || (rc = ex->d_cell_off.reserve(std::max(cell_base, 1))) ||
        (rc = ex->d_cell_items.reserve(std::max<size_t>(slot_base, 1))) || (rc = ex->d_cand_dev.reserve(std::max<size_t>(slot_base, 1))) ||
        (rc = ex->d_level_cnt_dev.reserve(ex->levels.size())) || (rc = ex->d_sel_cnt.reserve(ex->levels.size())) ||
        (rc = ex->d_oct_status.reserve(ex->levels.size())) ||
        (rc = ex->d_n_out.reserve(2 * (ex->n_cams + 1))) ||
        (rc = ex->d_sel_oct.reserve(std::max<size_t>(slot_blk.size(), 1))) || (rc = ex->d_slot_blk.reserve(std::max<size_t>(slot_blk.size(), 1))))
        return rc;
    if (!slot_blk.empty())
        MORB_HIP(hipMemcpyAsync(ex->d_slot_blk.p, slot_blk.data(), slot_blk.size() * sizeof(unsigned short), hipMemcpyHostToDevice, ex->stream));
    // (the level-0 table starts empty with every geometry: level 0 is in the pyramid buffer until a run says otherwise)
    if ((rc = ex->d_l0.reserve(64))) return rc;
    MORB_HIP(hipMemsetAsync(ex->d_l0.p, 0, 64 * sizeof(L0Src), ex->stream));
    ex->l0_host.assign(64, L0Src{nullptr, 0, 0});
    ex->l0_active = false;
    if (slot_base > ex->h_cand_cap) {
        if (ex->h_cand) (void)hipHostFree(ex->h_cand);
        ex->h_cand = nullptr; ex->h_cand_cap = 0;
        MORB_HIP(hipHostMalloc((void**)&ex->h_cand, slot_base * sizeof(uint32_t), hipHostMallocMapped));
        ex->h_cand_cap = slot_base;
    }
    MORB_HIP(hipMemcpyAsync(ex->d_levels.p, ex->levels.data(), ex->levels.size() * sizeof(LevelInfo), hipMemcpyHostToDevice, ex->stream));
    if (!ex->cell_map.empty())
        MORB_HIP(hipMemcpyAsync(ex->d_cell_map.p, ex->cell_map.data(), ex->cell_map.size() * sizeof(int2), hipMemcpyHostToDevice, ex->stream));
    if (!xt.empty()) MORB_HIP(hipMemcpyAsync(ex->d_xtab.p, xt.data(), xt.size() * sizeof(int2), hipMemcpyHostToDevice, ex->stream));
    if (!xg.empty()) MORB_HIP(hipMemcpyAsync(ex->d_xgrp.p, xg.data(), xg.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
    {
        static constexpr DescribeTables k_tabs = DescribeTables();   // (compile-time; the copy below is synchronised with the others)
        int rc_t = ex->d_desc_tabs.reserve(sizeof(DescribeTables));
        if (rc_t) return rc_t;
        MORB_HIP(hipMemcpyAsync(ex->d_desc_tabs.p, &k_tabs, sizeof(DescribeTables), hipMemcpyHostToDevice, ex->stream));
    }
    if (!yt.empty()) MORB_HIP(hipMemcpyAsync(ex->d_ytab.p, yt.data(), yt.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
    if (!psx.empty()) MORB_HIP(hipMemcpyAsync(ex->d_pyr_sx.p, psx.data(), psx.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
    if (!psy.empty()) MORB_HIP(hipMemcpyAsync(ex->d_pyr_sy.p, psy.data(), psy.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
    for (int pi = 0; pi < 2 && ex->tiled4; ++pi) {
        orbx_extractor::TilePlan& P = ex->tp[pi];
        if (P.sx.empty() || P.sy.empty()) continue;
        int rc_p;
        if ((rc_p = P.d_sx.reserve(P.sx.size())) || (rc_p = P.d_sy.reserve(P.sy.size()))) return rc_p;
        MORB_HIP(hipMemcpyAsync(P.d_sx.p, P.sx.data(), P.sx.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
        MORB_HIP(hipMemcpyAsync(P.d_sy.p, P.sy.data(), P.sy.size() * sizeof(int4), hipMemcpyHostToDevice, ex->stream));
    }
    MORB_HIP(hipStreamSynchronize(ex->stream));  // xt / yt are locals
    ex->tables_dirty = false;
    return ORB_OK;
}

static size_t pyramid_bytes(const CamTables& T, int W, int H) {
    size_t off = 0;
    for (int l = 0; l < T.p.nlevels; ++l) {
        const int w = cv_round((double)((float)W * T.inv_scale[l])), h = cv_round((double)((float)H * T.inv_scale[l]));
        off += (size_t)align_up(align_up(w, 64) * h, 256);
    }
    return off;
}

extern "C" {

int orbx_tables(const orbx_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                int32_t* features_per_level, int32_t* umax16) {
    MORB_ARG(p && p->nlevels >= 1 && p->nlevels <= MAX_LEVELS && p->scale_factor > 1.0f && p->nfeatures >= 0);
    CamTables T;
    build_cam_tables(*p, T);
    for (int i = 0; i < p->nlevels; ++i) {
        if (scale) scale[i] = T.scale[i];
        if (inv_scale) inv_scale[i] = T.inv_scale[i];
        if (sigma2) sigma2[i] = T.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = T.inv_sigma2[i];
        if (features_per_level) features_per_level[i] = T.quota[i];
    }
    if (umax16) compute_umax(umax16);
    return ORB_OK;
}

int orbx_create(const orbx_params* params, int n_cams, int max_width, int max_height, int device, orbx_extractor** out) {
    MORB_ARG(params && out && n_cams >= 1 && n_cams <= 64 && max_width >= 64 && max_height >= 64);
    for (int c = 0; c < n_cams; ++c)
        MORB_ARG(params[c].nlevels >= 1 && params[c].nlevels <= MAX_LEVELS && params[c].scale_factor > 1.0f &&
                 params[c].nfeatures >= 1 && params[c].min_th_fast >= 1 && params[c].ini_th_fast >= params[c].min_th_fast &&
                 params[c].ini_th_fast <= 255);
    {   // the kernels hard-code the ctor-derived constants; make sure the ctor arithmetic still yields them
        int um[16];
        compute_umax(um);
        const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
        for (int i = 0; i < 16; ++i) MORB_ARG(um[i] == expect[i]);
    }
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    orbx_extractor* ex = new orbx_extractor();
    ex->device = device; ex->n_cams = n_cams; ex->max_w = max_width; ex->max_h = max_height;
    hipError_t e;
    // MORB_RESERVE_CUS=k (default off): the extraction stream stays off k compute units of every XCD (a CU mask on its queue), which
    // are then free for the matcher's stream -- whose resolve is ONE workgroup that otherwise shares its CU with extraction waves.
    // Measured (round 6, profiles/r06/notes_experiments.md section 10): 4 x 640x480 overlapped 15 200 -> 17 900-18 600 steps/s with k = 4,
    // 2 x 640x480 and 2 x 1280x720 unchanged or slightly worse, 8 x 1080p 9 % worse, every isolated step 3-6 % slower: a tunable.
    if (const char* rs = getenv("MORB_RESERVE_CUS"); rs && atoi(rs) != 0) {
        const int k = std::max(0, atoi(rs));   // (negative: a masked queue with every unit allowed -- what the mask itself costs or buys)
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
        for (int b = 0; b < 8 * k && b < 256; ++b) mask[b >> 5] &= ~(1u << (b & 31));
        e = hipExtStreamCreateWithCUMask(&ex->stream, 8, mask);
    } else
    e = hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete ex; return ORB_E_HIP; }
    ex->cams.resize(n_cams);
    ex->cur_w.assign(n_cams, 0); ex->cur_h.assign(n_cams, 0);
    memset(&ex->ingest, 0, sizeof(ex->ingest));
    memset(&ex->sink, 0, sizeof(ex->sink));
    ex->n_out.assign(n_cams, 0);
    ex->d_kps.resize(n_cams); ex->d_desc.resize(n_cams);
    ex->out_kps.assign(n_cams, nullptr); ex->out_desc.assign(n_cams, nullptr); ex->out_cap_active.assign(n_cams, 0);
    ex->out_cap.assign(n_cams, 0);
    size_t sel_cap = 0;
    for (int c = 0; c < n_cams; ++c) {
        build_cam_tables(params[c], ex->cams[c]);
        ex->max_levels = std::max(ex->max_levels, params[c].nlevels);
        ex->cam_pitch = std::max(ex->cam_pitch, pyramid_bytes(ex->cams[c], max_width, max_height));
        ex->out_cap[c] = params[c].nfeatures + 4 * params[c].nlevels;  // quota + 2 overshoot per level, with margin
        sel_cap += ex->out_cap[c];
    }
    ex->cam_pitch = (ex->cam_pitch + 4095) / 4096 * 4096 + 4096;
#define ORBX_TRY(x) do { int rc_ = (x); if (rc_) { orbx_destroy(ex); return rc_; } } while (0)
#define ORBX_TRY_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { morb::set_error("%s: %s", #x, hipGetErrorString(e_)); orbx_destroy(ex); return ORB_E_HIP; } } while (0)
    ORBX_TRY(ex->d_pyr.reserve(ex->cam_pitch * n_cams));
    for (int c = 0; c < n_cams; ++c) {
        ORBX_TRY(ex->d_kps[c].reserve(ex->out_cap[c]));
        ORBX_TRY(ex->d_desc[c].reserve((size_t)ex->out_cap[c] * 32));
        ex->out_kps[c] = ex->d_kps[c].p; ex->out_desc[c] = ex->d_desc[c].p; ex->out_cap_active[c] = ex->out_cap[c];
    }
    ORBX_TRY(ex->d_out_kps.reserve(n_cams));
    ORBX_TRY(ex->d_out_desc.reserve(n_cams));
    ORBX_TRY(ex->d_sel.reserve(sel_cap));
    ORBX_TRY_HIP(hipHostMalloc((void**)&ex->h_sel, sel_cap * sizeof(SelKp), hipHostMallocDefault));
    ex->h_sel_cap = sel_cap;
    ORBX_TRY_HIP(hipHostMalloc((void**)&ex->h_level_cnt, (size_t)n_cams * MAX_LEVELS * sizeof(int), hipHostMallocMapped));
    ORBX_TRY_HIP(hipHostMalloc((void**)&ex->h_oct, (size_t)2 * (n_cams + 1) * sizeof(int), hipHostMallocMapped));
    ORBX_TRY_HIP(hipHostGetDevicePointer((void**)&ex->d_h_oct, ex->h_oct, 0));
    for (int i = 0; i < 2; ++i) ORBX_TRY_HIP(hipEventCreateWithFlags(&ex->ev_done[i], hipEventDisableTiming | hipEventReleaseToSystem));
    { const char* e = getenv("MORB_HOST_OCTREE"); ex->device_octree = !(e && atoi(e) != 0); }
    { const char* e = getenv("MORB_CHAIN_GRAPH"); ex->use_graph = !(e && atoi(e) == 0); }
    if (const char* e = getenv("MORB_OCT_MAX_KEYS")) ex->oct_max_keys = std::min(OCT_RK, std::max(1, atoi(e)));
    ORBX_TRY_HIP(hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(OctLdsR)));
    for (int i = 0; i < 6; ++i) ORBX_TRY_HIP(hipEventCreate(&ex->ev[i]));
    ex->level_cnt_last.assign((size_t)n_cams * ex->max_levels, 0);
    ex->uploaded.assign(n_cams, 0);
    ex->l0_host.assign(64, L0Src{nullptr, 0, 0});
    {   // host quadtree fallback: four threads (the calling one works too)
        const int hw = (int)std::thread::hardware_concurrency();
        ex->pool.reset(new TaskPool(std::max(1, std::min(4, hw > 1 ? hw : 1)) - 1));
    }
    *out = ex;
    return ORB_OK;
}

void orbx_destroy(orbx_extractor* ex) {
    if (!ex) return;
    (void)hipSetDevice(ex->device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    for (int sl = 0; sl < 2; ++sl) for (int w = 0; w < orbx_extractor::CHAIN_WAYS; ++w) ex->chain[sl][w].destroy();
    ex->d_l0.release(); ex->d_pyr.release(); ex->d_levels.release(); ex->d_cell_map.release(); ex->d_xtab.release(); ex->d_xgrp.release(); ex->d_desc_tabs.release(); ex->d_ytab.release(); ex->d_pyr_sx.release(); ex->d_pyr_sy.release();
    for (auto& P : ex->tp) { P.d_sx.release(); P.d_sy.release(); }
    ex->d_cell_cnt.release(); ex->d_cell_off.release(); ex->d_cell_items.release(); ex->d_sel.release(); ex->d_sel_oct.release();
    ex->d_cand_dev.release(); ex->d_level_cnt_dev.release(); ex->d_sel_cnt.release(); ex->d_oct_status.release();
    ex->d_n_out.release(); ex->d_slot_blk.release();
    for (auto& b : ex->d_kps) b.release();
    for (auto& b : ex->d_desc) b.release();
    ex->d_out_kps.release(); ex->d_out_desc.release();
    ex->stage_img[0].release(); ex->stage_img[1].release(); ex->own_mirror_kps.release(); ex->own_mirror_desc.release();
    if (ex->h_cand) (void)hipHostFree(ex->h_cand);
    if (ex->h_level_cnt) (void)hipHostFree(ex->h_level_cnt);
    if (ex->h_sel) (void)hipHostFree(ex->h_sel);
    if (ex->h_oct) (void)hipHostFree(ex->h_oct);
    for (int i = 0; i < 2; ++i) if (ex->ev_done[i]) (void)hipEventDestroy(ex->ev_done[i]);
    if (ex->ev_foreign) (void)hipEventDestroy(ex->ev_foreign);
    for (int i = 0; i < 6; ++i) if (ex->ev[i]) (void)hipEventDestroy(ex->ev[i]);
    if (ex->stream && ex->stream_owned) (void)hipStreamDestroy(ex->stream);
    delete ex;
}

void* orbx_stream(const orbx_extractor* ex) { return ex ? (void*)ex->stream : nullptr; }

// Everything enqueued on `other_stream` so far happens before whatever this handle enqueues next (no host wait): a caller that has
// device work reading this handle's output buffers in flight on another stream calls this before the next run overwrites them.
int orbx_wait_for_stream(orbx_extractor* ex, void* other_stream) {
    MORB_ARG(ex != nullptr);
    MORB_HIP(hipSetDevice(ex->device));
    if (!ex->ev_foreign) MORB_HIP(hipEventCreateWithFlags(&ex->ev_foreign, hipEventDisableTiming));
    MORB_HIP(hipEventRecord(ex->ev_foreign, (hipStream_t)other_stream));
    MORB_HIP(hipStreamWaitEvent(ex->stream, ex->ev_foreign, 0));
    return ORB_OK;
}

static int upload_common(orbx_extractor* ex, int cam, const uint8_t* src, int width, int height, int stride, hipMemcpyKind kind) {
    MORB_ARG(ex && cam >= 0 && cam < ex->n_cams);
    MORB_HIP(hipSetDevice(ex->device));
    ex->uploaded[cam] = 1;
    if (!src || width <= 0 || height <= 0) {  // empty image: camera produces nothing (reference :1047-1048)
        if (ex->cur_w[cam] != 0) { ex->cur_w[cam] = ex->cur_h[cam] = 0; ex->tables_dirty = true; }
        ex->ingest.src[cam] = nullptr;
        return ORB_OK;
    }
    MORB_ARG(width <= ex->max_w && height <= ex->max_h && stride >= width);
    if (ex->cur_w[cam] != width || ex->cur_h[cam] != height) {
        ex->cur_w[cam] = width; ex->cur_h[cam] = height;
        ex->tables_dirty = true;
    }
    // level 0 lives at offset 0 of the camera's pyramid buffer with pitch align64(width)
    if (kind == hipMemcpyDeviceToDevice) {  // copied by k_ingest at the start of the run, all cameras in one launch
        ex->ingest.src[cam] = src; ex->ingest.stride[cam] = stride; ex->ingest_pending = true;
        return ORB_OK;
    }
    // A host image in pageable memory (a cv::Mat): the runtime would stage it through an internal pinned buffer and a DMA of
    // its own.  Instead the rows are written once into host-visible staging (HBM behind the large BAR where the part has one)
    // and level 0 is filled by the ingest kernel of the run, together with every other camera.  Page-locked images keep the
    // asynchronous DMA (the caller may refill a pageable buffer as soon as this call returns, a pinned one is read later).
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) {
        (void)hipGetLastError();
        // the staging slot is written NOW, not in stream order: with two runs in flight both slots are still being read
        if (ex->inflight >= 2) { morb::set_error("two runs are in flight: orbx_finish the older one before uploading the next images"); return ORB_E_ARG; }
        morb::StageBuf& S = ex->stage_img[ex->run_seq & 1u];   // (the slot of the run that will consume it)
        const size_t per_cam = (size_t)ex->max_w * ex->max_h;
        if (S.reserve(per_cam * ex->n_cams) == ORB_OK) {
            uint8_t* d = S.p + per_cam * cam;
            if (stride == width) memcpy(d, src, (size_t)width * height);
            else for (int y = 0; y < height; ++y) memcpy(d + (size_t)y * width, src + (size_t)y * stride, width);
            S.publish();
            ex->ingest.src[cam] = S.dp + per_cam * cam; ex->ingest.stride[cam] = width; ex->ingest_pending = true;
            if (!S.in_hbm) ex->ingest_host = true;
            return ORB_OK;
        }
    }
    // A page-locked image is device-visible as it is: when the caller asks for it (orbx_set_pinned_ingest: a run somebody
    // is waiting for), the ingest kernel of the run reads it across PCIe, all cameras in ONE launch (in stream order, like the
    // copy it replaces: a pitched hipMemcpy2DAsync per camera costs ~10 us of host time each).  MORB_PINNED_INGEST=0: never.
    if (pinned && ex->pinned_ingest && attr.devicePointer) {
        ex->ingest.src[cam] = static_cast<const uint8_t*>(attr.devicePointer); ex->ingest.stride[cam] = stride; ex->ingest_pending = true;
        ex->ingest_host = true;
        return ORB_OK;
    }
    ex->ingest.src[cam] = nullptr;
    MORB_HIP(hipMemcpy2DAsync(ex->d_pyr.p + (size_t)cam * ex->cam_pitch, align_up(width, 64), src, stride, width, height, kind, ex->stream));
    return ORB_OK;
}

int orbx_upload(orbx_extractor* ex, int cam, const uint8_t* gray, int width, int height, int stride) {
    return upload_common(ex, cam, gray, width, height, stride, hipMemcpyHostToDevice);
}

int orbx_upload_device(orbx_extractor* ex, int cam, const uint8_t* d_gray, int width, int height, int stride) {
    return upload_common(ex, cam, d_gray, width, height, stride, hipMemcpyDeviceToDevice);
}

int orbx_bind_output(orbx_extractor* ex, int cam, orb_keypoint* d_kps, uint8_t* d_desc, int cap) {
    MORB_ARG(ex && cam >= 0 && cam < ex->n_cams);
    if (!d_kps || !d_desc) {
        ex->out_kps[cam] = ex->d_kps[cam].p; ex->out_desc[cam] = ex->d_desc[cam].p; ex->out_cap_active[cam] = ex->out_cap[cam];
    } else {
        MORB_ARG(cap >= 1);
        ex->out_kps[cam] = d_kps; ex->out_desc[cam] = d_desc; ex->out_cap_active[cam] = cap;
    }
    ex->out_ptrs_dirty = true;
    return ORB_OK;
}

int orbx_set_host_mirror(orbx_extractor* ex, orb_keypoint* kps_devptr, uint8_t* desc_devptr, int cap_total) {
    MORB_ARG(ex != nullptr);
    ex->mirror_kps = kps_devptr; ex->mirror_desc = desc_devptr; ex->mirror_cap = (kps_devptr && desc_devptr) ? cap_total : 0;
    if (!ex->mirror_cap) { ex->mirror_kps = nullptr; ex->mirror_desc = nullptr; }
    return ORB_OK;
}

int orbx_debug_last_path(const orbx_extractor* ex) { return ex ? ex->last_path : ORB_E_ARG; }

int orbx_debug_level0_in_place(const orbx_extractor* ex) {
    if (!ex) return ORB_E_ARG;
    int n = 0;
    if (ex->l0_active) for (int c = 0; c < ex->n_cams; ++c) n += ex->l0_host[c].ptr != nullptr;
    return n;
}

// 0 one tile launch (k_pyramid_tiled), 2 the generic chain (one k_resize launch per level), 3 two tile launches with four pixels per
// lane (k_pyramid_tiled4); the geometry of the most recent run
int orbx_debug_pyramid_form(const orbx_extractor* ex) {
    if (!ex) return ORB_E_ARG;
    if (ex->tables_dirty) return -1;
    return ex->tiled_ok ? 0 : (ex->tiled4 ? 3 : 2);
}

// tests of the large-rig plan's geometry: other tiles, another split level (all handles created afterwards; 0 = keep)
int orbx_debug_pyramid_plan(int tile_w, int tile_h, int split) {
    if (tile_w > 0) g_t4_plan[0] = tile_w;
    if (tile_h > 0) g_t4_plan[1] = tile_h;
    if (split > 0) g_t4_plan[2] = split;
    return ORB_OK;
}

int orbx_set_profiling(orbx_extractor* ex, int on) {
    MORB_ARG(ex != nullptr);
    ex->profiling = on != 0;
    return ORB_OK;
}

int orbx_stage_times_us(const orbx_extractor* ex, float* out6) {
    MORB_ARG(ex && out6);
    for (int i = 0; i < 6; ++i) out6[i] = ex->stage_us[i];
    return ORB_OK;
}

static int orbx_run_impl(orbx_extractor* ex, bool allow_async);

int orbx_run(orbx_extractor* ex) {
    MORB_ARG(ex != nullptr);
    int rc = orbx_run_impl(ex, false);
    return rc;
}

// Enqueues the whole extractor without a host synchronisation when the device quadtree is in use (otherwise identical
// to orbx_run).  Counts become known with orbx_finish(); until then they live in HBM (orbx_device_counts).
int orbx_run_async(orbx_extractor* ex) {
    MORB_ARG(ex != nullptr);
    return orbx_run_impl(ex, true);
}

static int finish_device_path(orbx_extractor* ex);

int orbx_finish(orbx_extractor* ex) {
    MORB_ARG(ex != nullptr);
    if (ex->inflight == 0) return ORB_OK;
    MORB_HIP(hipSetDevice(ex->device));
    MORB_HIP(hipEventSynchronize(ex->ev_done[(ex->run_seq - (unsigned)ex->inflight) & 1]));  // the OLDEST run in flight
    return finish_device_path(ex);
}

// orbx_finish for a caller that has PROOF that the oldest run in flight completed (it has seen the results of work that was
// ordered behind the run on the GPU): no wait on the completion event.
int orbx_finish_completed(orbx_extractor* ex) {
    MORB_ARG(ex != nullptr);
    if (ex->inflight == 0) return ORB_OK;
    return finish_device_path(ex);
}

int orbx_adopt_stream(orbx_extractor* ex, void* stream) {
    MORB_ARG(ex && stream && ex->inflight == 0);
    MORB_HIP(hipSetDevice(ex->device));
    if (ex->stream) {
        MORB_HIP(hipStreamSynchronize(ex->stream));
        if (ex->stream_owned) (void)hipStreamDestroy(ex->stream);
    }
    ex->stream = static_cast<hipStream_t>(stream); ex->stream_owned = false;
    return ORB_OK;
}

int orbx_discard(orbx_extractor* ex) {
    MORB_ARG(ex != nullptr);
    if (ex->inflight == 0) return ORB_OK;
    MORB_HIP(hipSetDevice(ex->device));
    MORB_HIP(hipEventSynchronize(ex->ev_done[(ex->run_seq - (unsigned)ex->inflight) & 1]));
    --ex->inflight;
    return ORB_OK;
}

int orbx_set_defer_done(orbx_extractor* ex, int on) {
    MORB_ARG(ex != nullptr);
    ex->defer_done = on != 0;
    return ORB_OK;
}

int orbx_record_done(orbx_extractor* ex) {   // the completion event of the most recently enqueued run, recorded NOW on the stream
    MORB_ARG(ex && ex->inflight > 0);
    MORB_HIP(hipSetDevice(ex->device));
    MORB_HIP(hipEventRecord(ex->ev_done[(ex->run_seq - 1u) & 1u], ex->stream));
    return ORB_OK;
}

// 1: device images handed to orbx_upload_device are read IN PLACE as pyramid level 0 by every run until the camera's next upload
// (large rigs only: the resize-chain pyramid; see k_set_l0).  The caller promises that such an image stays valid and unchanged
// until then -- orbf's own contract for the images of a step.  MORB_L0_INPLACE=0: never.
int orbx_set_inplace_level0(orbx_extractor* ex, int on) {
    MORB_ARG(ex != nullptr);
    if (ex->l0_optin != (on != 0)) { ex->l0_optin = on != 0; ex->tables_dirty = true; }   // (captured chains carry the table pointer: new epoch)
    return ORB_OK;
}

int orbx_set_pinned_ingest(orbx_extractor* ex, int on) {
    MORB_ARG(ex != nullptr);
    ex->pinned_ingest = on != 0;
    return ORB_OK;
}

int orbx_set_chain_graph(orbx_extractor* ex, int on) {
    MORB_ARG(ex != nullptr);
    ex->graph_next_run = on != 0;
    return ORB_OK;
}

int orbx_set_chain_tail(orbx_extractor* ex, orbx_tail_fn fn, void* user, int tag) {
    MORB_ARG(ex != nullptr);
    ex->tail_fn = fn; ex->tail_user = user; ex->tail_tag = tag;
    return ORB_OK;
}

int orbx_set_frame_sink(orbx_extractor* ex, const FrameSink* sink) {
    MORB_ARG(ex != nullptr && (sink == nullptr || ex->n_cams <= 4));
    if (sink) ex->sink = *sink; else memset(&ex->sink, 0, sizeof(ex->sink));
    return ORB_OK;
}

// counts of the most recently enqueued run
// (n_cams totals + the OR of the quadtree's status words behind them: non-zero = this run's results will be redone on the host path)
const int* orbx_device_counts(const orbx_extractor* ex) { return ex ? ex->d_n_out.p + ((ex->run_seq - 1u) & 1u) * (ex->n_cams + 1) : nullptr; }
int orbx_pending(const orbx_extractor* ex) { return ex ? ex->inflight : 0; }
// Status word of the OLDEST run in flight (0: every level stayed inside the device quadtree's limits); only meaningful
// once that run has completed on the device (the caller has seen its completion event).  -1: nothing in flight.
int orbx_peek_status(const orbx_extractor* ex) {
    if (!ex || ex->inflight == 0) return -1;
    const unsigned oldest = (ex->run_seq - (unsigned)ex->inflight) & 1u;
    return ex->h_oct[oldest * (ex->n_cams + 1) + ex->n_cams];
}
void* orbx_done_event(const orbx_extractor* ex) { return ex ? (void*)ex->ev_done[(ex->run_seq - 1u) & 1u] : nullptr; }

// K1 + K2/K3 of a run: the pyramid chain and the per-cell FAST kernel
// fused: the level-0 sources of a pending ingest (k_pyramid_tiled then reads them itself: no k_ingest launch); NULL: level 0 is in place
// the {pointer, pitch} table of level 0 the kernels are handed: the handle's table while the geometry reads level 0 in place
// (part of what a captured chain carries: the flag only changes together with the geometry epoch), else none
static inline const L0Src* l0_table(const orbx_extractor* ex) { return ex->l0_on ? (const L0Src*)ex->d_l0.p : nullptr; }

static int launch_pyramid_fast(orbx_extractor* ex, hipStream_t st, const IngestArgs* fused = nullptr) {
    const int ML = ex->max_levels;
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[0], st));
    if (ex->tiled_ok) {
        PyrArgs A;
        for (int c = 0; c < 64; ++c) {
            A.src[c] = fused && c < ex->n_cams ? fused->src[c] : nullptr; A.stride[c] = fused ? fused->stride[c] : 0;
            A.tx[c] = c < ex->n_cams ? ex->pyr_tx[c] : 0; A.ty[c] = c < ex->n_cams ? ex->pyr_ty[c] : 0;
            A.w[c] = c < ex->n_cams ? (short)ex->cur_w[c] : 0; A.h[c] = c < ex->n_cams ? (short)ex->cur_h[c] : 0;
        }
        hipLaunchKernelGGL(k_pyramid_tiled<256>, dim3(ex->pyr_tx_max, ex->pyr_ty_max, ex->n_cams), dim3(256), (size_t)ex->pyr_lds, st, A,
                           (const LevelInfo*)ex->d_levels.p, ML, ex->d_pyr.p, ex->cam_pitch, (const int2*)ex->d_xtab.p,
                           (const int4*)ex->d_ytab.p, (const int4*)ex->d_pyr_sx.p, (const int4*)ex->d_pyr_sy.p, ex->pyr_tx_max, ex->pyr_ty_max,
                           ex->pyr_tab_cap, ex->pyr_tile);
    } else {
    auto level_dims = [&](int l, int* mw, int* mh) {
        *mw = 0; *mh = 0;
        if (l >= ML) return;
        for (int c = 0; c < ex->n_cams; ++c) {
            const LevelInfo& Lv = ex->levels[(size_t)c * ML + l];
            *mw = std::max(*mw, Lv.w); *mh = std::max(*mh, Lv.h);
        }
    };
    if (ex->tiled4) {
        for (int pi = 0; pi < 2; ++pi) {
            const orbx_extractor::TilePlan& P = ex->tp[pi];
            if (P.last <= P.base || P.tx_max == 0) continue;
            Tile4Args T;
            T.L = (const LevelInfo*)ex->d_levels.p; T.pyr = ex->d_pyr.p; T.cam_pitch = ex->cam_pitch; T.xgrp = (const int4*)ex->d_xgrp.p;
            T.ytab = (const int4*)ex->d_ytab.p; T.sx = (const int4*)P.d_sx.p; T.sy = (const int4*)P.d_sy.p; T.l0 = l0_table(ex);
            T.max_levels = ML; T.base = P.base; T.last = P.last; T.tw = P.tw; T.th = P.th; T.halo_x = P.halo_x; T.halo_y = P.halo_y;
            T.tx_max = P.tx_max; T.ty_max = P.ty_max; T.gcap = P.gcap; T.rcap = P.rcap;
            for (int c = 0; c < 64; ++c) { T.tx[c] = c < ex->n_cams ? P.tx[c] : 0; T.ty[c] = c < ex->n_cams ? P.ty[c] : 0; }
            if (P.threads == 512) hipLaunchKernelGGL(k_pyramid_tiled4<512>, dim3(P.tx_max, P.ty_max, ex->n_cams), dim3(512), (size_t)P.lds, st, T);
            else hipLaunchKernelGGL(k_pyramid_tiled4<256>, dim3(P.tx_max, P.ty_max, ex->n_cams), dim3(256), (size_t)P.lds, st, T);
        }
    } else
    for (int l = 1; l < ML; ++l) {   // the generic chain: one launch per level, any level step (what parameter sets outside both tile forms take)
        int mw = 0, mh = 0;
        level_dims(l, &mw, &mh);
        if (mw == 0) continue;
        dim3 grid((mw + 255) / 256, (mh + 3) / 4, ex->n_cams), block(64, 4, 1);
        hipLaunchKernelGGL(k_resize, grid, block, 0, st, (const LevelInfo*)ex->d_levels.p, ML, l, ex->d_pyr.p, ex->cam_pitch,
                           (const int2*)ex->d_xtab.p, (const int4*)ex->d_ytab.p);
    }
    }
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[1], st));
    // per-cell FAST / NMS / threshold / compaction
    // (throughput form from ~4 rounds of the chip's 2048 resident two-wave cells on; MORB_FAST_FORM=1 / 2 forces the latency / throughput form)
    static const int form_env = [] { const char* e = getenv("MORB_FAST_FORM"); return e ? atoi(e) : 0; }();
    const bool throughput = form_env ? form_env == 2 : ex->total_cells >= 8192;
    if (throughput)
        hipLaunchKernelGGL((k_fast_cells<128, 8>), dim3(ex->total_cells), dim3(128), fast_cells_lds(ex->cell_h_max, ex->cell_px_max), st,
                           (const LevelInfo*)ex->d_levels.p, (const int2*)ex->d_cell_map.p, (const uint8_t*)ex->d_pyr.p, ex->cam_pitch, ML,
                           ex->d_cell_cnt.p, ex->d_cell_items.p, ex->cell_h_max, ex->cell_px_max, l0_table(ex));
    else
        hipLaunchKernelGGL((k_fast_cells<256, 4>), dim3(ex->total_cells), dim3(256), fast_cells_lds(ex->cell_h_max, ex->cell_px_max), st,
                           (const LevelInfo*)ex->d_levels.p, (const int2*)ex->d_cell_map.p, (const uint8_t*)ex->d_pyr.p, ex->cam_pitch, ML,
                           ex->d_cell_cnt.p, ex->d_cell_items.p, ex->cell_h_max, ex->cell_px_max, l0_table(ex));
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[2], st));
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

// K4 on the device + K5-K7 from the slotted list, for the run that uses count slot `slot`
static int launch_tree_describe(orbx_extractor* ex, hipStream_t st, unsigned slot, const FrameSink& sink) {
    const int ML = ex->max_levels;
    int* d_h_oct = ex->d_h_oct + slot * (ex->n_cams + 1);
    MirrorArgs mir;
    mir.kps = nullptr; mir.desc = nullptr;
    if (ex->mirror_kps) { mir.kps = ex->mirror_kps; mir.desc = ex->mirror_desc; }  // cap_total covers every camera's capacity
    for (int c = 0; c < 64; ++c) mir.base[c] = 0;
    hipLaunchKernelGGL(k_octree, dim3(ex->n_cams * ML), dim3(1024), sizeof(OctLdsR), st, (const LevelInfo*)ex->d_levels.p,
                       (const int*)ex->d_cell_cnt.p, (const uint32_t*)ex->d_cell_items.p, ex->d_sel_oct.p, ex->d_sel_cnt.p,
                       ex->d_oct_status.p, ML, ex->oct_max_keys);
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[4], st));
    hipLaunchKernelGGL(k_describe, dim3((ex->total_sel_slots + 3) / 4), dim3(256), 0, st, (const LevelInfo*)ex->d_levels.p, ML,
                       (const uint8_t*)ex->d_pyr.p, ex->cam_pitch, (const SelKp*)ex->d_sel_oct.p, ex->total_sel_slots,
                       (orb_keypoint* const*)ex->d_out_kps.p, (uint8_t* const*)ex->d_out_desc.p, mir,
                       SelListArgs{(const unsigned short*)ex->d_slot_blk.p, (const int*)ex->d_sel_cnt.p,
                                   (const int*)ex->d_oct_status.p, ex->d_n_out.p + slot * (ex->n_cams + 1), d_h_oct, ex->n_cams},
                       sink, l0_table(ex), (const DescribeTables*)ex->d_desc_tabs.p);
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[5], st));
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

static int orbx_run_impl(orbx_extractor* ex, bool allow_async) {
    MORB_HIP(hipSetDevice(ex->device));
    const auto t_begin = std::chrono::steady_clock::now();
    int rc;
    if (ex->tables_dirty && (rc = rebuild_geometry(ex))) return rc;
    if (ex->out_ptrs_dirty) {
        MORB_HIP(hipMemcpyAsync(ex->d_out_kps.p, ex->out_kps.data(), ex->n_cams * sizeof(void*), hipMemcpyHostToDevice, ex->stream));
        MORB_HIP(hipMemcpyAsync(ex->d_out_desc.p, ex->out_desc.data(), ex->n_cams * sizeof(void*), hipMemcpyHostToDevice, ex->stream));
        MORB_HIP(hipStreamSynchronize(ex->stream));
        ex->out_ptrs_dirty = false;
    }
    const int ML = ex->max_levels;
    hipStream_t st = ex->stream;
    if (allow_async && ex->inflight >= 2) { morb::set_error("two runs are already in flight: orbx_finish the older one first"); return ORB_E_ARG; }
    if (!allow_async && ex->inflight > 0) {  // a synchronous run abandons whatever was still in flight
        MORB_HIP(hipStreamSynchronize(st));
        ex->inflight = 0;
    }
    if (ex->inflight == 0) std::fill(ex->n_out.begin(), ex->n_out.end(), 0);
    if (ex->total_cells == 0) {  // every camera empty
        memset(&ex->ingest, 0, sizeof(ex->ingest)); ex->ingest_pending = false; ex->ingest_host = false;
        return ORB_OK;
    }
    uint32_t* d_cand = nullptr; int* d_level_cnt = nullptr;
    MORB_HIP(hipHostGetDevicePointer((void**)&d_cand, ex->h_cand, 0));
    MORB_HIP(hipHostGetDevicePointer((void**)&d_level_cnt, ex->h_level_cnt, 0));

    // Level 0 of the images that are not in place yet (device images, staged pageable images, page-locked images read in place).
    // A run whose launches are issued one by one hands the sources to the tiled pyramid launch, which reads them itself; a
    // run that replays a captured chain (its launches carry no per-run pointers) copies them first with k_ingest.
    const bool dev_tree = ex->device_octree && ex->total_sel_slots > 0;
    const bool will_replay = dev_tree && ex->use_graph && ex->graph_next_run && allow_async && !ex->profiling;
    IngestArgs fused_src;
    const IngestArgs* fused = nullptr;
    bool any_upload = false;
    for (int c = 0; c < ex->n_cams; ++c) any_upload |= ex->uploaded[c] != 0;
    if (ex->ingest_pending || (ex->l0_active && any_upload)) {
        // (sources in host memory are copied exactly once by k_ingest: the tiles' halos would cross PCIe twice)
        if (ex->ingest_pending && ex->tiled_ok && !will_replay && !ex->ingest_host) {
            fused_src = ex->ingest; fused = &fused_src;
        } else {
            // Level 0 in place (k_set_l0 above): every camera that has an image must have been handed a device image for THIS run,
            // 4-byte aligned with a pitch that is a multiple of 4.  Otherwise the images are copied -- and a camera that was read in
            // place so far and got no new image in this run is copied too, from where it lies (valid until its next upload), before
            // the table is emptied: no run ever finds a camera's level 0 in neither place.
            bool inplace = ex->l0_on && ex->ingest_pending && !ex->ingest_host;
            for (int c = 0; c < ex->n_cams && inplace; ++c) {
                if (ex->cur_w[c] == 0) continue;
                const uint8_t* p = ex->ingest.src[c];
                if (!p || (reinterpret_cast<uintptr_t>(p) & 3) || (ex->ingest.stride[c] & 3)) inplace = false;
            }
            if (inplace) {
                IngestArgs T = ex->ingest;
                for (int c = 0; c < 64; ++c) if (c >= ex->n_cams || ex->cur_w[c] == 0) { T.src[c] = nullptr; T.stride[c] = 0; }
                hipLaunchKernelGGL(k_set_l0, dim3(1), dim3(64), 0, st, T, ex->d_l0.p, ex->n_cams);
                for (int c = 0; c < ex->n_cams; ++c) ex->l0_host[c] = L0Src{T.src[c], T.stride[c], 0};
                ex->l0_active = true;
            } else {
                IngestArgs I = ex->ingest;
                if (ex->l0_active) {
                    for (int c = 0; c < ex->n_cams; ++c)
                        if (!ex->uploaded[c] && ex->cur_w[c] > 0 && ex->l0_host[c].ptr) { I.src[c] = ex->l0_host[c].ptr; I.stride[c] = ex->l0_host[c].stride; }
                    MORB_HIP(hipMemsetAsync(ex->d_l0.p, 0, 64 * sizeof(L0Src), st));
                    ex->l0_host.assign(64, L0Src{nullptr, 0, 0});
                    ex->l0_active = false;
                }
                int mw = 0, mh = 0;
                for (int c = 0; c < ex->n_cams; ++c)
                    if (I.src[c]) { mw = std::max(mw, ex->cur_w[c]); mh = std::max(mh, ex->cur_h[c]); }
                if (mw > 0)
                    hipLaunchKernelGGL(k_ingest, dim3((mw + 1023) / 1024, (mh + 3) / 4, ex->n_cams), dim3(64, 4, 1), 0, st, I,
                                       (const LevelInfo*)ex->d_levels.p, ML, ex->d_pyr.p, ex->cam_pitch);
            }
        }
        for (int c = 0; c < ex->n_cams; ++c) ex->ingest.src[c] = nullptr;
        ex->ingest_pending = false; ex->ingest_host = false;
    }
    std::fill(ex->uploaded.begin(), ex->uploaded.end(), 0);
    if (!dev_tree) {
        if ((rc = launch_pyramid_fast(ex, st, fused))) return rc;
        // K3b (only on the host-quadtree path and for the inspection hook): dense cell-major lists, also into pinned host memory
        hipLaunchKernelGGL(k_compact, dim3(ex->n_cams * ML), dim3(1024), 0, st, (const LevelInfo*)ex->d_levels.p,
                           (const int*)ex->d_cell_cnt.p, (const uint32_t*)ex->d_cell_items.p, ex->d_cell_off.p, d_cand, d_level_cnt,
                           ex->d_cand_dev.p, ex->d_level_cnt_dev.p);
        if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[3], st));
        MORB_HIP(hipGetLastError());
    }
    ex->cand_valid = !dev_tree;

    // K4 on the device: quadtree -> K5-K7 straight from the slotted list; ONE sync afterwards.  Falls through to the host
    // quadtree when a level is outside the device limits.
    if (dev_tree) {
        const unsigned slot = ex->run_seq & 1u;
        // The ten launches of the chain are captured once per slot into a kernel-only graph and replayed with one
        // hipGraphLaunch (the host cost of enqueueing them is what bounds overlapped timesteps).  Everything the launches
        // carry is in the key: geometry epoch, result mirrors, frame sink.
        const FrameSink sink = allow_async ? ex->sink : FrameSink{};
        const bool graphable = will_replay;
        bool done = false;
        if (graphable) {
            // a few graphs per slot: a caller rotates through more result sets / frames than there are slots
            int way = -1;
            for (int w = 0; w < orbx_extractor::CHAIN_WAYS; ++w) {
                const orbx_extractor::ChainGraph& C = ex->chain[slot][w];
                if (C.exec && C.epoch == ex->geom_epoch && C.mirror_kps == ex->mirror_kps && C.mirror_desc == ex->mirror_desc &&
                    C.tail_tag == (ex->tail_fn ? ex->tail_tag : 0) && memcmp(&C.sink, &sink, sizeof(FrameSink)) == 0) { way = w; break; }
            }
            const bool hit = way >= 0;
            if (!hit) { way = ex->chain_next[slot]; ex->chain_next[slot] = (way + 1) % orbx_extractor::CHAIN_WAYS; }
            orbx_extractor::ChainGraph& G = ex->chain[slot][way];
            if (hit) {
                MORB_HIP(hipGraphLaunch(G.exec, st));
                done = true;
            } else {
                G.destroy();
                hipGraph_t g = nullptr;
                hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    const int rc1 = launch_pyramid_fast(ex, st);
                    int rc2 = rc1 ? rc1 : launch_tree_describe(ex, st, slot, sink);
                    if (!rc2 && ex->tail_fn) { ++ex->run_seq; rc2 = ex->tail_fn(ex->tail_user, (void*)st); --ex->run_seq; }  // (the tail sees this run as the most recent one)
                    e = hipStreamEndCapture(st, &g);
                    if (!rc2 && e == hipSuccess && g && hipGraphInstantiate(&G.exec, g, nullptr, nullptr, 0) == hipSuccess) {
                        G.graph = g; G.epoch = ex->geom_epoch; G.mirror_kps = ex->mirror_kps; G.mirror_desc = ex->mirror_desc; G.sink = sink;
                        G.tail_tag = ex->tail_fn ? ex->tail_tag : 0;
                        MORB_HIP(hipGraphLaunch(G.exec, st));
                        done = true;
                    } else {
                        if (g) (void)hipGraphDestroy(g);
                        G.exec = nullptr;
                    }
                }
                if (!done) { (void)hipGetLastError(); ex->use_graph = false; }  // plain launches from now on
            }
        }
        if (!done) {
            if ((rc = launch_pyramid_fast(ex, st, fused))) return rc;
            if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[3], st));
            if ((rc = launch_tree_describe(ex, st, slot, sink))) return rc;
            if (ex->tail_fn) { ++ex->run_seq; rc = ex->tail_fn(ex->tail_user, (void*)st); --ex->run_seq; if (rc) return rc; }
        }
        // (a caller that puts the run's consumer on the same stream records the completion event itself, behind that work:
        // an event record with system-scope release standing between two dependent kernels costs ~10 us of queue time each)
        if (!(allow_async && ex->defer_done)) MORB_HIP(hipEventRecord(ex->ev_done[slot], st));
        ex->prof_valid[slot] = ex->profiling && ex->inflight == 0;  // (one set of stage events: not for overlapped runs)
        ++ex->run_seq; ++ex->inflight; ex->t_begin_async = t_begin;
        if (allow_async) return ORB_OK;
        MORB_HIP(hipStreamSynchronize(st));
        if (ex->h_oct[slot * (ex->n_cams + 1) + ex->n_cams] == 0) return finish_device_path(ex);
        ex->inflight = 0;
        std::fill(ex->n_out.begin(), ex->n_out.end(), 0);  // a level exceeded the device limits: redo the selection on the host
        hipLaunchKernelGGL(k_compact, dim3(ex->n_cams * ML), dim3(1024), 0, st, (const LevelInfo*)ex->d_levels.p,
                           (const int*)ex->d_cell_cnt.p, (const uint32_t*)ex->d_cell_items.p, ex->d_cell_off.p, d_cand, d_level_cnt,
                           ex->d_cand_dev.p, ex->d_level_cnt_dev.p);
        MORB_HIP(hipGetLastError());
        MORB_HIP(hipStreamSynchronize(st));
        ex->cand_valid = true;
    } else {
        MORB_HIP(hipStreamSynchronize(st));
    }
    const auto t_host0 = std::chrono::steady_clock::now();

    ex->last_path = 2;
    // K4 (host): quadtree per (camera, level) on the worker pool; output order = level-major, list order inside a level
    int n_tasks = 0;
    for (int c = 0; c < ex->n_cams; ++c) {
        const CamTables& T = ex->cams[c];
        for (int l = 0; l < T.p.nlevels; ++l) {
            const LevelInfo& Lv = ex->levels[(size_t)c * ML + l];
            const int n = Lv.w ? ex->h_level_cnt[c * ML + l] : 0;
            ex->level_cnt_last[(size_t)c * ML + l] = n;
            if (n == 0) continue;
            if ((int)ex->tasks.size() <= n_tasks) ex->tasks.emplace_back();
            OctreeTask& t = ex->tasks[n_tasks++];
            t.cam = c; t.level = l; t.n = n;
        }
    }
    // biggest problems first (level 0 dominates): better balance across workers
    std::sort(ex->tasks.begin(), ex->tasks.begin() + n_tasks, [](const OctreeTask& a, const OctreeTask& b) {
        return a.n != b.n ? a.n > b.n : (a.cam != b.cam ? a.cam < b.cam : a.level < b.level);
    });
    ex->pool->run(n_tasks, [ex, ML](int i) {
        OctreeTask& t = ex->tasks[i];
        const LevelInfo& Lv = ex->levels[(size_t)t.cam * ML + t.level];
        const uint32_t* cand = ex->h_cand + Lv.cand_base;
        t.cx.resize(t.n); t.cy.resize(t.n); t.cr.resize(t.n);
        for (int k = 0; k < t.n; ++k) {
            const uint32_t v = cand[k];
            t.cx[k] = v & 0xfff; t.cy[k] = (v >> 12) & 0xfff; t.cr[k] = v >> 24;
        }
        morb::distribute_octree(t.cx.data(), t.cy.data(), t.cr.data(), t.n, Lv.w - 2 * MIN_BORDER, Lv.h - 2 * MIN_BORDER,
                                ex->cams[t.cam].quota[t.level], t.selected);
    });
    // assemble in (camera, level) order
    std::vector<int> order(n_tasks);
    for (int i = 0; i < n_tasks; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [ex](int a, int b) {
        const OctreeTask &x = ex->tasks[a], &y = ex->tasks[b];
        return x.cam != y.cam ? x.cam < y.cam : x.level < y.level;
    });
    int nsel = 0;
    {
        int cur_cam = -1, out_idx = 0;
        for (int oi = 0; oi < n_tasks; ++oi) {
            const OctreeTask& t = ex->tasks[order[oi]];
            if (t.cam != cur_cam) { if (cur_cam >= 0) ex->n_out[cur_cam] = out_idx; cur_cam = t.cam; out_idx = 0; }
            for (int s : t.selected) {
                if (out_idx >= ex->out_cap_active[t.cam] || (size_t)nsel >= ex->h_sel_cap) {
                    morb::set_error("camera %d produced more keypoints than its output capacity %d", t.cam, ex->out_cap_active[t.cam]);
                    return ORB_E_CAPACITY;
                }
                SelKp& K = ex->h_sel[nsel++];
                K.x = t.cx[s] + MIN_BORDER; K.y = t.cy[s] + MIN_BORDER;  // :843-844
                K.camlevel = (t.cam << 8) | t.level;
                K.resp_out = (int)(((unsigned)t.cr[s] << 24) | (unsigned)out_idx);
                ++out_idx;
            }
        }
        if (cur_cam >= 0) ex->n_out[cur_cam] = out_idx;
    }
    const auto t_host1 = std::chrono::steady_clock::now();
    if (ex->profiling) MORB_HIP(hipEventRecord(ex->ev[4], st));
    MirrorArgs mir;
    mir.kps = nullptr; mir.desc = nullptr;
    if (ex->mirror_kps && nsel <= ex->mirror_cap) {
        mir.kps = ex->mirror_kps; mir.desc = ex->mirror_desc;
        int base = 0;
        for (int c = 0; c < 64; ++c) { mir.base[c] = base; if (c < ex->n_cams) base += ex->n_out[c]; }
    }
    if (nsel > 0) {
        MORB_HIP(hipMemcpyAsync(ex->d_sel.p, ex->h_sel, (size_t)nsel * sizeof(SelKp), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_describe, dim3((nsel + 3) / 4), dim3(256), 0, st, (const LevelInfo*)ex->d_levels.p, ML,
                           (const uint8_t*)ex->d_pyr.p, ex->cam_pitch, (const SelKp*)ex->d_sel.p, nsel,
                           (orb_keypoint* const*)ex->d_out_kps.p, (uint8_t* const*)ex->d_out_desc.p, mir,
                           SelListArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, FrameSink{}, l0_table(ex),
                           (const DescribeTables*)ex->d_desc_tabs.p);
        MORB_HIP(hipGetLastError());
    }
    if (ex->profiling) {
        MORB_HIP(hipEventRecord(ex->ev[5], st));
        MORB_HIP(hipStreamSynchronize(st));
        float ms;
        MORB_HIP(hipEventElapsedTime(&ms, ex->ev[0], ex->ev[1])); ex->stage_us[0] = ms * 1000.f;
        MORB_HIP(hipEventElapsedTime(&ms, ex->ev[1], ex->ev[2])); ex->stage_us[1] = ms * 1000.f;
        MORB_HIP(hipEventElapsedTime(&ms, ex->ev[2], ex->ev[3])); ex->stage_us[2] = ms * 1000.f;
        ex->stage_us[3] = std::chrono::duration<float, std::micro>(t_host1 - t_host0).count();
        MORB_HIP(hipEventElapsedTime(&ms, ex->ev[4], ex->ev[5])); ex->stage_us[4] = ms * 1000.f;
        ex->stage_us[5] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return ORB_OK;
}

// After the stream has drained: adopt the device path's counts, or redo the selection on the host if a level was outside
// the device limits (then the downstream work a caller enqueued on stale counts is invalid: it is told via the return).
static int finish_device_path(orbx_extractor* ex) {
    const unsigned oldest = (ex->run_seq - (unsigned)ex->inflight) & 1u;
    const int* h_oct = ex->h_oct + oldest * (ex->n_cams + 1);
    --ex->inflight;
    if (h_oct[ex->n_cams] == 0) {
        ex->last_path = 0;
        for (int c = 0; c < ex->n_cams; ++c) {
            ex->n_out[c] = h_oct[c];
            if (ex->n_out[c] > ex->out_cap_active[c]) { morb::set_error("a camera produced more keypoints than its output capacity"); return ORB_E_CAPACITY; }
        }
        if (ex->profiling && ex->prof_valid[oldest]) {
            float ms;
            MORB_HIP(hipEventElapsedTime(&ms, ex->ev[0], ex->ev[1])); ex->stage_us[0] = ms * 1000.f;
            MORB_HIP(hipEventElapsedTime(&ms, ex->ev[1], ex->ev[2])); ex->stage_us[1] = ms * 1000.f;
            MORB_HIP(hipEventElapsedTime(&ms, ex->ev[2], ex->ev[3])); ex->stage_us[2] = ms * 1000.f;
            MORB_HIP(hipEventElapsedTime(&ms, ex->ev[3], ex->ev[4])); ex->stage_us[3] = ms * 1000.f;  // device quadtree
            MORB_HIP(hipEventElapsedTime(&ms, ex->ev[4], ex->ev[5])); ex->stage_us[4] = ms * 1000.f;
            ex->stage_us[5] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - ex->t_begin_async).count();
        }
        return ORB_OK;
    }
    // a level exceeded the device limits.  With a newer run in flight the resident images are already being replaced:
    // the caller has to upload this run's images again and run synchronously (2).  Otherwise the synchronous
    // host-quadtree path runs right here on the same resident images (1).
    if (ex->inflight > 0) return 2;
    const bool saved = ex->device_octree;
    ex->device_octree = false;
    int rc = orbx_run_impl(ex, false);
    ex->device_octree = saved;
    return rc ? rc : 1;  // 1 = results valid, but they were produced by the fallback (anything enqueued on the async counts is stale)
}

int orbx_count(const orbx_extractor* ex, int cam) {
    if (!ex || cam < 0 || cam >= ex->n_cams) return ORB_E_ARG;
    return ex->n_out[cam];
}

const orb_keypoint* orbx_device_keypoints(const orbx_extractor* ex, int cam) {
    return (ex && cam >= 0 && cam < ex->n_cams) ? ex->out_kps[cam] : nullptr;
}
const uint8_t* orbx_device_descriptors(const orbx_extractor* ex, int cam) {
    return (ex && cam >= 0 && cam < ex->n_cams) ? ex->out_desc[cam] : nullptr;
}

int orbx_download(orbx_extractor* ex, int cam, orb_keypoint* kps, uint8_t* desc, int cap) {
    MORB_ARG(ex && cam >= 0 && cam < ex->n_cams);
    MORB_HIP(hipSetDevice(ex->device));
    const int n = ex->n_out[cam];
    if (n > cap) { morb::set_error("camera %d has %d keypoints, capacity %d", cam, n, cap); return ORB_E_CAPACITY; }
    if (n > 0) {
        MORB_ARG(kps && desc);
        MORB_HIP(hipMemcpyAsync(kps, ex->out_kps[cam], (size_t)n * sizeof(orb_keypoint), hipMemcpyDeviceToHost, ex->stream));
        MORB_HIP(hipMemcpyAsync(desc, ex->out_desc[cam], (size_t)n * 32, hipMemcpyDeviceToHost, ex->stream));
    }
    MORB_HIP(hipStreamSynchronize(ex->stream));
    return ORB_OK;
}

int orbx_extract(orbx_extractor* ex, int n_cams, const uint8_t* const* gray, const int* width, const int* height,
                 const int* stride, orb_keypoint* const* kps_out, uint8_t* const* desc_out, const int* cap, int* n_out) {
    MORB_ARG(ex && n_cams == ex->n_cams && gray && width && height && stride && kps_out && desc_out && cap && n_out);
    int rc;
    MORB_HIP(hipSetDevice(ex->device));
    for (int c = 0; c < n_cams; ++c)
        if ((rc = orbx_upload(ex, c, gray[c], width[c], height[c], stride[c]))) return rc;
    // results through a pinned mirror the describe kernel writes itself (camera-major, packed): no D2H copies on the stream
    const bool own = ex->mirror_kps == nullptr;
    if (own) {
        int cap_total = 0;
        for (int c = 0; c < n_cams; ++c) cap_total += ex->out_cap[c];
        if ((rc = ex->own_mirror_kps.reserve(cap_total)) || (rc = ex->own_mirror_desc.reserve((size_t)cap_total * 32))) return rc;
        ex->mirror_kps = ex->own_mirror_kps.dp; ex->mirror_desc = ex->own_mirror_desc.dp; ex->mirror_cap = cap_total;
    }
    rc = orbx_run(ex);
    // (the host-quadtree fallback describes from a host-built list with its own mirror bookkeeping: plain downloads there)
    const bool mirrored = own && ex->mirror_cap > 0 && ex->last_path != 2;
    if (own) { ex->mirror_kps = nullptr; ex->mirror_desc = nullptr; ex->mirror_cap = 0; }
    if (rc) return rc;
    int total = 0;
    for (int c = 0; c < n_cams; ++c) total += ex->n_out[c];
    int off = 0;
    for (int c = 0; c < n_cams; ++c) {
        n_out[c] = ex->n_out[c];
        if (n_out[c] == 0) continue;  // outputs untouched for an empty camera
        if (n_out[c] > cap[c]) { morb::set_error("camera %d has %d keypoints, capacity %d", c, n_out[c], cap[c]); return ORB_E_CAPACITY; }
        if (mirrored && total <= (int)ex->own_mirror_kps.cap) {
            MORB_ARG(kps_out[c] && desc_out[c]);
            memcpy(kps_out[c], ex->own_mirror_kps.p + off, (size_t)n_out[c] * sizeof(orb_keypoint));
            memcpy(desc_out[c], ex->own_mirror_desc.p + (size_t)off * 32, (size_t)n_out[c] * 32);
        } else if ((rc = orbx_download(ex, c, kps_out[c], desc_out[c], cap[c]))) return rc;
        off += n_out[c];
    }
    return ORB_OK;
}

int orbx_debug_level(orbx_extractor* ex, int cam, int level, uint8_t* out, int cap_bytes, int* w, int* h) {
    MORB_ARG(ex && cam >= 0 && cam < ex->n_cams && level >= 0 && level < ex->cams[cam].p.nlevels && w && h);
    MORB_ARG(!ex->tables_dirty);
    MORB_HIP(hipSetDevice(ex->device));
    const LevelInfo& Lv = ex->levels[(size_t)cam * ex->max_levels + level];
    *w = Lv.w; *h = Lv.h;
    if (Lv.w == 0) return ORB_OK;
    if (Lv.w * Lv.h > cap_bytes) { morb::set_error("level needs %d bytes", Lv.w * Lv.h); return ORB_E_CAPACITY; }
    const uint8_t* lsrc = ex->d_pyr.p + (size_t)cam * ex->cam_pitch + Lv.pyr_off;
    size_t lpitch = (size_t)Lv.stride;
    if (level == 0 && ex->l0_active && ex->l0_host[cam].ptr) { lsrc = ex->l0_host[cam].ptr; lpitch = (size_t)ex->l0_host[cam].stride; }   // (read in place)
    MORB_HIP(hipMemcpy2DAsync(out, Lv.w, lsrc, lpitch, Lv.w, Lv.h,
                              hipMemcpyDeviceToHost, ex->stream));
    MORB_HIP(hipStreamSynchronize(ex->stream));
    return ORB_OK;
}

int orbx_debug_candidates(orbx_extractor* ex, int cam, int level, orb_keypoint* out, int cap, int* n) {
    MORB_ARG(ex && cam >= 0 && cam < ex->n_cams && level >= 0 && level < ex->cams[cam].p.nlevels && n);
    MORB_ARG(!ex->tables_dirty);
    if (!ex->cand_valid) {  // the device-quadtree path skips the dense lists: build them now from the per-cell slots
        MORB_HIP(hipSetDevice(ex->device));
        uint32_t* d_cand = nullptr; int* d_level_cnt = nullptr;
        MORB_HIP(hipHostGetDevicePointer((void**)&d_cand, ex->h_cand, 0));
        MORB_HIP(hipHostGetDevicePointer((void**)&d_level_cnt, ex->h_level_cnt, 0));
        hipLaunchKernelGGL(k_compact, dim3(ex->n_cams * ex->max_levels), dim3(1024), 0, ex->stream, (const LevelInfo*)ex->d_levels.p,
                           (const int*)ex->d_cell_cnt.p, (const uint32_t*)ex->d_cell_items.p, ex->d_cell_off.p, d_cand, d_level_cnt,
                           ex->d_cand_dev.p, ex->d_level_cnt_dev.p);
        MORB_HIP(hipGetLastError());
        MORB_HIP(hipStreamSynchronize(ex->stream));
        for (size_t b = 0; b < ex->levels.size(); ++b) ex->level_cnt_last[b] = ex->levels[b].w ? ex->h_level_cnt[b] : 0;
        ex->cand_valid = true;
    }
    const LevelInfo& Lv = ex->levels[(size_t)cam * ex->max_levels + level];
    const int cnt = ex->level_cnt_last[(size_t)cam * ex->max_levels + level];
    *n = cnt;
    const uint32_t* cand = ex->h_cand + Lv.cand_base;
    for (int i = 0; i < cnt && i < cap; ++i) {
        const uint32_t v = cand[i];
        out[i].x = (float)(v & 0xfff); out[i].y = (float)((v >> 12) & 0xfff);
        out[i].size = 7.f; out[i].angle = -1.f; out[i].response = (float)(v >> 24); out[i].octave = 0; out[i].class_id = -1;
    }
    return ORB_OK;
}

int orbx_debug_distribute_octree(const orb_keypoint* in, int n, int min_x, int max_x, int min_y, int max_y,
                                 int n_features, orb_keypoint* out, int cap, int* n_out) {
    MORB_ARG(n >= 0 && n_out && (n == 0 || in) && (cap == 0 || out) && max_x > min_x && max_y > min_y);
    std::vector<int> x(n), y(n), r(n), sel;
    for (int i = 0; i < n; ++i) { x[i] = (int)in[i].x; y[i] = (int)in[i].y; r[i] = (int)in[i].response; }
    morb::distribute_octree(x.data(), y.data(), r.data(), n, max_x - min_x, max_y - min_y, n_features, sel);
    *n_out = (int)sel.size();
    for (int i = 0; i < (int)sel.size() && i < cap; ++i) out[i] = in[sel[i]];
    return ORB_OK;
}

}  // extern "C"

#ifdef MORB_DESCRIBE_SELFCHECK
extern "C" int morb_debug_describe_check(unsigned long long* out, int n) {   // debug build only: k_describe's self-check words
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_describe_check), (size_t)std::min(n, 8 + 16 * 14 + 16 + 4) * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef MORB_PHASE_CLOCKS
extern "C" int morb_debug_oct_dbg(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_oct_dbg), &v, sizeof v) == hipSuccess ? 0 : -1; }
extern "C" int morb_debug_phases_extractor(int which, unsigned long long* out64) {
    if (which == 4) return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_pyr), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
    if (which == 2) return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_desc), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
    if (which == 3) return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_desc_wave), 2 * 4096 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_oct), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
