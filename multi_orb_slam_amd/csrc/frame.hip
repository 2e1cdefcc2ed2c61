// frame.hip -- the matcher-side view of a frame: Frame merge, ComputeStereoFromRGBD, AssignFeaturesToGrid on the device
// (reference src/Frame.cc:191-395, :959-986) and the host-array form orbm_frame_create.
//   k_frame_unpack        host-built frames: one staging block -> the frame's arrays
//   k_frame_build_small   one workgroup per frame (<= 8192 features, <= 4 cameras): counts -> LDS histogram -> scan -> scatter -> rank
//   k_frame_fill + k_scan_cells + k_scatter_cells + k_sort_cells   the multi-kernel form for larger frames
//   k_cams_from_counts    camera table finished on the device from the extractor's counts in HBM
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/orbm.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "matcher_internal.h"
#include "mirror_dev.h"

using namespace morb;

namespace {
// ------------------------------------------------------------------------------------------------ host-built frames
// orbm_frame_create: the host packs a frame's arrays back to back (dword granularity) into ONE staging block; this kernel
// scatters them into the frame's own buffers.
struct UnpackPlan { const uint32_t* src; uint32_t* dst[9]; int end[9]; };   // end[k] = first dword behind section k

__global__ __launch_bounds__(256) void k_frame_unpack(UnpackPlan P) {
    const int total = P.end[8];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int k = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) k += (i >= P.end[j]);
        const int base = k ? P.end[k - 1] : 0;
        P.dst[k][i - base] = P.src[i];
    }
}

// Optional pinned-host (device-mapped) destinations: results the host needs are written there by the kernels
// themselves, so no D2H copy kernels sit on the stream.
struct HostMirror { orb_keypoint* kps; uint4* desc; float* ur; float* depth; float* unx; float* uny; orb_calibration calib; };

// One feature of the merged frame: the `_total` record, its stereo coordinate and its grid cell.
__device__ __forceinline__ int frame_fill_one(const CamFeat* __restrict__ cams, int n_cams, int g, float mbf, float minX,
                                              float minY, float invW, float invH, float* __restrict__ x,
                                              float* __restrict__ y, float* __restrict__ ur, float* __restrict__ depth_out,
                                              int* __restrict__ oct, float* __restrict__ ang,
                                              orb_keypoint* __restrict__ kps_g, uint4* __restrict__ desc_g,
                                              const HostMirror& hm) {
    int c = 0;
    while (c + 1 < n_cams && g >= cams[c].base + cams[c].n) ++c;
    const CamFeat C = cams[c];
    const int l = g - C.base;
    const orb_keypoint k = C.kps[l];
    const uint4 d0 = C.desc[2 * l], d1 = C.desc[2 * l + 1];
    float ux = k.x, uy = k.y;  // Frame::UndistortKeyPoints (src/Frame.cc:673-705): a copy when k1 == 0
    if (hm.calib.k1 != 0.0f) morb_undistort_point(hm.calib, k.x, k.y, &ux, &uy);
    x[g] = ux; y[g] = uy; oct[g] = k.octave; ang[g] = k.angle; kps_g[g] = k;
    desc_g[2 * g] = d0; desc_g[2 * g + 1] = d1;
    float d = -1.f, u_r = -1.f;
    if (C.depth) {
        const float dv = C.depth[(size_t)(int)k.y * C.depth_stride + (int)k.x];  // imDepth.at<float>(v,u): float -> int truncation
        if (dv > 0) { d = dv; u_r = ux - mbf / dv; }  // kpU.pt.x - mbf/d (src/Frame.cc:981)
    }
    ur[g] = u_r; depth_out[g] = d;
    if (hm.kps) { hm.kps[g] = k; hm.desc[2 * g] = d0; hm.desc[2 * g + 1] = d1; }
    if (hm.ur) { hm.ur[g] = u_r; hm.depth[g] = d; }
    if (hm.unx) { hm.unx[g] = ux; hm.uny[g] = uy; }
    const int px = (int)roundf((ux - minX) * invW), py = (int)roundf((uy - minY) * invH);
    if (px >= 0 && px < ORBM_GRID_COLS && py >= 0 && py < ORBM_GRID_ROWS) return (c * ORBM_GRID_COLS + px) * ORBM_GRID_ROWS + py;
    return -1;
}

// Frame merge + ComputeStereoFromRGBD + PosInGrid for every feature (reference src/Frame.cc:221-239, :959-986, :632-642)
__global__ __launch_bounds__(256) void k_frame_fill(const CamFeat* __restrict__ cams, int n_cams, int n_total, float mbf,
                                                    float minX, float minY, float invW, float invH,
                                                    float* __restrict__ x, float* __restrict__ y, float* __restrict__ ur,
                                                    float* __restrict__ depth_out, int* __restrict__ oct,
                                                    float* __restrict__ ang, orb_keypoint* __restrict__ kps_g,
                                                    uint4* __restrict__ desc_g, int* __restrict__ cell_of,
                                                    int* __restrict__ cell_cnt, HostMirror hm, const int* __restrict__ n_dev) {
    MORB_LATENCY_KERNEL_WIDE();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (n_dev) n_total = *n_dev;  // counts only known on the device: the launch was sized for the capacity
    if (g >= n_total) return;
    const int cell = frame_fill_one(cams, n_cams, g, mbf, minX, minY, invW, invH, x, y, ur, depth_out, oct, ang, kps_g, desc_g, hm);
    if (cell >= 0) atomicAdd(&cell_cnt[cell], 1);
    cell_of[g] = cell;
}

// The whole frame assembly in ONE workgroup (n_total <= 8192, n_cams <= 4): fill, per-cell counts and cursors in LDS,
// scan, scatter, per-cell sort.  Replaces memset + 4 launches on the small frames of a 2-4 camera rig.
struct CamFeat4 { CamFeat c[4]; };
MORB_PHASE_DECL(g_ph_fb);

// orbm_frame_create_resident: the per-feature fields of a host-built frame arrive in ONE staging block --
// x | y | uright | angle | octave | src (n dwords each; src = cam << 24 | row inside that camera) | cam_start (n_cams + 1) |
// pad to 16 bytes | the descriptor rows of the cameras that are not resident, camera after camera --
// and the descriptors of the resident cameras are read where the extractor left them in HBM.  The grid is built here,
// with the host's arithmetic (round-to-cell insertion of src/Frame.cc:632-642: float subtract, float multiply, roundf).
struct StagedFill {
    const uint32_t* stage;      // NULL: not this mode
    const uint4* d_desc[4];     // resident cameras: device rows; NULL: rows in the staging block from host_row0[c] on
    int host_row0[4];
    int desc_off;               // dword offset of the staged descriptor rows
};

__device__ __forceinline__ int frame_fill_staged(const StagedFill& S, int n, int n_cams, int g, float minX, float minY, float invW,
                                                 float invH, float* __restrict__ x, float* __restrict__ y, float* __restrict__ ur,
                                                 int* __restrict__ oct, float* __restrict__ ang, uint4* __restrict__ desc_g) {
    const uint32_t* w = S.stage;
    const float fx = __uint_as_float(w[g]), fy = __uint_as_float(w[(size_t)n + g]);
    const uint32_t src = w[(size_t)5 * n + g];
    const int cam = (int)(src >> 24), row = (int)(src & 0xffffffu);
    x[g] = fx; y[g] = fy; ur[g] = __uint_as_float(w[(size_t)2 * n + g]);
    ang[g] = __uint_as_float(w[(size_t)3 * n + g]); oct[g] = (int)w[(size_t)4 * n + g];
    if (cam < n_cams) {
        const uint4* rows = S.d_desc[cam] ? S.d_desc[cam] + 2 * (size_t)row
                                          : reinterpret_cast<const uint4*>(w + S.desc_off) + 2 * ((size_t)S.host_row0[cam] + row);
        desc_g[2 * g] = rows[0]; desc_g[2 * g + 1] = rows[1];
    }
    const int px = (int)roundf((fx - minX) * invW), py = (int)roundf((fy - minY) * invH);
    if (cam < n_cams && px >= 0 && px < ORBM_GRID_COLS && py >= 0 && py < ORBM_GRID_ROWS) return (cam * ORBM_GRID_COLS + px) * ORBM_GRID_ROWS + py;
    return -1;
}

__global__ __launch_bounds__(1024) void k_frame_build_small(CamFeat4 cams4, int* __restrict__ cam_start_out, const int* __restrict__ d_counts,
                                                            int* __restrict__ n_total_out, int n_cams, int n_total, float mbf,
                                                            float minX, float minY, float invW, float invH,
                                                            float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ ur, float* __restrict__ depth_out,
                                                            int* __restrict__ oct, float* __restrict__ ang,
                                                            orb_keypoint* __restrict__ kps_g, uint4* __restrict__ desc_g,
                                                            int* __restrict__ cell_start, int* __restrict__ items, HostMirror hm,
                                                            const int* __restrict__ cell_of_in, int desc_rows, StagedFill staged) {
    MORB_LATENCY_KERNEL();
    // cell_of_in != NULL: the per-feature arrays and the cells were already written by the extractor's describe kernel
    // (FrameSink); only the counts, the grid and its item lists are produced here.
    extern __shared__ __attribute__((aligned(16))) int s_cells[];  // [ncell + 1] start | [ncell + 1] cursor | u16 items[8192]
    __shared__ int wsum[16];
    // kernel-argument copy of the per-camera descriptors (no H2D).  With d_counts the real counts come from the device
    // (the extractor has not been synchronised yet): bases and the total are derived here.
    __shared__ CamFeat s_cams[4];
    __shared__ int s_ntotal;
    MORB_PHASE(g_ph_fb, 0);
    if (threadIdx.x == 0) {
        int base = 0;
        for (int c = 0; c < n_cams; ++c) {
            CamFeat cf = cams4.c[c];
            if (d_counts) { cf.n = d_counts[c]; cf.base = base; }
            if (staged.stage) { cf.base = (int)staged.stage[(size_t)6 * n_total + c]; cf.n = (int)staged.stage[(size_t)6 * n_total + c + 1] - cf.base; }
            s_cams[c] = cf;
            cam_start_out[c] = cf.base;
            base = cf.base + cf.n;
        }
        cam_start_out[n_cams] = base;
        s_ntotal = d_counts ? base : n_total;
        if (n_total_out) { n_total_out[0] = s_ntotal; n_total_out[1] = 0; n_total_out[2] = s_ntotal; }  // {features, first query, queries}
        if (desc_g) {  // trailer of the descriptor block: the per-camera counts (what a multi-GPU exchange ships with it)
            int* tail = reinterpret_cast<int*>(desc_g + 2 * (size_t)desc_rows);
            for (int c = 0; c < n_cams; ++c) tail[c] = s_cams[c].n;
            // (the extractor's status word rides behind its counts: a block of a run that will be redone says so -- ORBM_BLOCK_REDO)
            if (d_counts && d_counts[n_cams] != 0) tail[0] |= ORBM_BLOCK_REDO;
        }
    }
    __syncthreads();
    const CamFeat* cams = s_cams;
    n_total = s_ntotal;
    const int ncell = n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    int* s_start = s_cells;
    int* s_cur = s_cells + ncell + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c <= ncell; c += 1024) s_cur[c] = 0;
    __syncthreads();
    MORB_PHASE(g_ph_fb, 1);
    int mycell[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int g = tid + k * 1024;
        mycell[k] = -1;
        if (g < n_total) {
            mycell[k] = cell_of_in ? cell_of_in[g]
                        : staged.stage ? frame_fill_staged(staged, n_total, n_cams, g, minX, minY, invW, invH, x, y, ur, oct, ang, desc_g)
                                   : frame_fill_one(cams, n_cams, g, mbf, minX, minY, invW, invH, x, y, ur, depth_out, oct, ang, kps_g, desc_g, hm);
            if (mycell[k] >= 0) atomicAdd(&s_cur[mycell[k]], 1);
        }
    }
    __syncthreads();
    MORB_PHASE(g_ph_fb, 2);
    // exclusive scan of the counts (in s_cur) -> s_start; s_cur becomes the running insert position
    const int per = (ncell + 1023) / 1024;
    const int c0 = min(ncell, tid * per), c1 = min(ncell, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += s_cur[c];
    const int incl = wave_incl_scan(mine);   // (DPP row shifts: no LDS crossbar round trips)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wave) before += v; total += v; }   // (every thread sums the 16 wave totals itself: no serial pass, no second barrier)
    if (tid == 0) s_start[ncell] = total;
    int run = before + incl - mine;
    for (int c = c0; c < c1; ++c) { const int v = s_cur[c]; s_start[c] = run; s_cur[c] = run; run += v; }
    __syncthreads();
    MORB_PHASE(g_ph_fb, 3);
    // scatter + per-cell sort in LDS (feature indices fit 16 bits here), one coalesced write of the finished item list
    unsigned short* s_items = reinterpret_cast<unsigned short*>(s_cells + 2 * (ncell + 1));
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (mycell[k] >= 0) s_items[atomicAdd(&s_cur[mycell[k]], 1)] = (unsigned short)(tid + k * 1024);
    __syncthreads();
    MORB_PHASE(g_ph_fb, 4);
    for (int c = tid; c <= ncell; c += 1024) cell_start[c] = s_start[c];
    // ascending global index inside every cell: every feature ranks itself among the unsorted items of its cell
    // (independent LDS reads; a per-cell insertion sort is a dependent chain, quadratic in the fullest cell)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (mycell[k] < 0) continue;
        const int g = tid + k * 1024;
        const int sidx = s_start[mycell[k]], e = s_start[mycell[k] + 1];
        int rank = 0;
        for (int i = sidx; i < e; ++i) rank += s_items[i] < g ? 1 : 0;
        items[sidx + rank] = g;
    }
    __syncthreads();
    MORB_PHASE(g_ph_fb, 5);
}

// Large frames with the per-camera counts still on the device: the camera table (count, base), the camera starts, the
// {features, first query, queries} triple and the count trailer of the descriptor block, from the extractor's counts.
__global__ void k_cams_from_counts(CamFeat* __restrict__ cams, int n_cams, const int* __restrict__ d_counts,
                                   int* __restrict__ cam_start, int* __restrict__ range, int* __restrict__ trailer) {
    MORB_LATENCY_KERNEL();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int c = 0; c < n_cams; ++c) {
        const int n = d_counts[c];
        cams[c].n = n; cams[c].base = base;
        cam_start[c] = base; trailer[c] = n;
        base += n;
    }
    if (d_counts[n_cams] != 0) trailer[0] |= ORBM_BLOCK_REDO;
    cam_start[n_cams] = base;
    range[0] = base; range[1] = 0; range[2] = base;
}

// The same with everything the large-rig assembly used to put in front of k_frame_fill in one launch (round 5): the camera table comes in
// the kernel arguments (it was a parameter block copied from pinned memory: two copies), workgroup 0's first thread finishes it from the
// extractor's counts (k_cams_from_counts), and all workgroups clear the per-cell counters (a memset) -- four launches of a chain whose
// small launches cost 5-8 us each inside the loop become one.  Up to 32 cameras (the argument block), else the separate launches.
struct CamFeat32 { CamFeat c[32]; };
__global__ __launch_bounds__(256) void k_frame_head(CamFeat32 H, CamFeat* __restrict__ cams, int n_cams, const int* __restrict__ d_counts,
                                                    int* __restrict__ cam_start, int* __restrict__ range, int* __restrict__ trailer,
                                                    int* __restrict__ cursor, int n_cursor) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_cursor; i += gridDim.x * 256) cursor[i] = 0;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int c = 0; c < n_cams; ++c) {
        const int n = d_counts[c];
        CamFeat e = H.c[c];
        e.n = n; e.base = base;
        cams[c] = e;
        cam_start[c] = base; trailer[c] = n;
        base += n;
    }
    if (d_counts[n_cams] != 0) trailer[0] |= ORBM_BLOCK_REDO;
    cam_start[n_cams] = base;
    range[0] = base; range[1] = 0; range[2] = base;
}

// Round 6: k_frame_head's launch gone too.  Every workgroup of the fill derives the camera table (count, base) from the extractor's
// counts for itself -- one count per lane of its first wave, a wave scan --, workgroup 0 also writes what later kernels and the host
// read (table, camera starts, the {features, first query, queries} triple, the block's count trailer), and nobody counts cells in
// HBM any more: k_grid_cam counts its camera's cells in LDS from the cell indices this kernel leaves (no counters to clear, no
// global atomics).  One launch of the large-rig chain less (12 us of a 552 us isolated 8 x 1080p step, and of every overlapped step).
__global__ __launch_bounds__(256) void k_frame_fill_head(CamFeat32 H, CamFeat* __restrict__ cams_out, int n_cams, const int* __restrict__ d_counts,
                                                         int* __restrict__ cam_start, int* __restrict__ range, int* __restrict__ trailer,
                                                         float mbf, float minX, float minY, float invW, float invH,
                                                         float* __restrict__ x, float* __restrict__ y, float* __restrict__ ur,
                                                         float* __restrict__ depth_out, int* __restrict__ oct,
                                                         float* __restrict__ ang, orb_keypoint* __restrict__ kps_g,
                                                         uint4* __restrict__ desc_g, int* __restrict__ cell_of, HostMirror hm) {
    MORB_LATENCY_KERNEL_WIDE();
    __shared__ CamFeat s_cams[32];
    __shared__ int s_total;
    if (threadIdx.x < 64) {   // (a whole wave: the scan needs every lane)
        const int c = threadIdx.x;
        const int n = c < n_cams ? d_counts[c] : 0;
        const int incl = wave_incl_scan(n);
        if (c < n_cams) {
            CamFeat e = H.c[c];
            e.n = n; e.base = incl - n;
            s_cams[c] = e;
            if (blockIdx.x == 0) {
                cams_out[c] = e; cam_start[c] = incl - n;
                trailer[c] = (c == 0 && d_counts[n_cams] != 0) ? (n | ORBM_BLOCK_REDO) : n;
            }
        }
        if (c == 63) {
            s_total = incl;
            if (blockIdx.x == 0) { cam_start[n_cams] = incl; range[0] = incl; range[1] = 0; range[2] = incl; }
        }
    }
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= s_total) return;
    cell_of[g] = frame_fill_one(s_cams, n_cams, g, mbf, minX, minY, invW, invH, x, y, ur, depth_out, oct, ang, kps_g, desc_g, hm);
}

// exclusive scan of cnt[0..n) into start[0..n], single 1024-thread block; cursor = copy of start
__global__ __launch_bounds__(1024) void k_scan_cells(const int* cnt, int n, int* __restrict__ start, int* cursor) {
    MORB_LATENCY_KERNEL();
    // cnt and cursor may alias (the per-cell counters are turned into insert cursors in place).
    // Every wave owns one contiguous run of cells and walks it 64 cells at a time (coalesced; round 2 gave every THREAD a run
    // of its own: 24 dependent, uncoalesced loads per thread for an 8-camera frame, twice -- 48 us).
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = ((n + 15) / 16 + 63) & ~63;              // cells per wave, a multiple of 64
    const int c0 = min(n, wave * per), c1 = min(n, c0 + per);
    int mine = 0;
    for (int c = c0 + lane; c < c1; c += 64) mine += cnt[c];  // (independent loads: all in flight)
    const int tot = __builtin_amdgcn_readlane(wave_incl_scan(mine), 63);
    if (lane == 0) wsum[wave] = tot;
    __syncthreads();
    int run = 0;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (tid == 1023) start[n] = run + tot;
    for (int c = c0; c < c1; c += 64) {
        const int v = c + lane < c1 ? cnt[c + lane] : 0;
        const int incl = wave_incl_scan(v);
        if (c + lane < c1) { start[c + lane] = run + incl - v; cursor[c + lane] = run + incl - v; }
        run += __builtin_amdgcn_readlane(incl, 63);
    }
}

__global__ __launch_bounds__(256) void k_scatter_cells(const int* __restrict__ cell_of, int n_total, int* __restrict__ cursor,
                                                       int* __restrict__ items, const int* __restrict__ n_dev) {
    MORB_LATENCY_KERNEL_WIDE();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (n_dev) n_total = *n_dev;
    if (g >= n_total) return;
    const int cell = cell_of[g];
    if (cell >= 0) items[atomicAdd(&cursor[cell], 1)] = g;
}

// ascending global index inside every cell (the atomics above scatter in arbitrary order; cells hold a handful of items)
__global__ __launch_bounds__(256) void k_sort_cells(const int* __restrict__ start, int ncell, int* __restrict__ items) {
    MORB_LATENCY_KERNEL_WIDE();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncell) return;
    const int s = start[c], e = start[c + 1];
    for (int i = s + 1; i < e; ++i) {
        const int v = items[i];
        int j = i - 1;
        while (j >= s && items[j] > v) { items[j + 1] = items[j]; --j; }
        items[j + 1] = v;
    }
}

// The grid of a LARGE frame in one launch (round 5): one workgroup per camera turns the per-cell counts k_frame_fill left into the camera's
// part of the CSR -- scan, scatter and the ascending order inside every cell all in LDS -- where k_scan_cells (one workgroup), k_scatter_cells
// (global atomics) and k_sort_cells (insertion sort in HBM) were three dependent launches of 13 + 6-17 + 21-27 us inside the configs[4] loop.
// Cameras are contiguous in both the cell numbering and the feature numbering, so a camera's part starts at the number of in-grid features of
// the cameras in front of it: every workgroup adds those counts up for itself (no workgroup waits for another).  Same contents as the three
// kernels produced: items of a cell in ascending global index.  Cameras of up to GRID_CAM_MAX features; larger ones keep the three launches.
constexpr int GRID_CELLS = ORBM_GRID_COLS * ORBM_GRID_ROWS;   // 3072 cells per camera
constexpr int GRID_CAM_MAX = 16384;
__global__ __launch_bounds__(1024) void k_grid_cam(const int* __restrict__ cell_cnt, const int* __restrict__ cell_of,
                                                   const int* __restrict__ cam_start, int n_cams, int* __restrict__ cell_start,
                                                   int* __restrict__ items) {
    MORB_LATENCY_KERNEL();
    __shared__ int s_start[GRID_CELLS + 1];          // exclusive scan of the camera's counts (local positions)
    __shared__ int s_fill[GRID_CELLS];               // insert cursors
    __shared__ unsigned short s_items[GRID_CAM_MAX]; // feature index inside the camera, cell after cell
    __shared__ int s_w[16], s_before;
    const int cam = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = cam_start[cam], nf = cam_start[cam + 1] - f0;
    const int c3 = 3 * tid;
    static_assert(GRID_CELLS == 3 * 1024, "three cells per thread");
    int before = 0, k0, k1, k2;
    if (cell_cnt) {
        // in-grid features of the cameras in front of this one: the sum of their cells' counts (coalesced, all loads independent)
        for (int i = tid; i < cam * GRID_CELLS; i += 1024) before += cell_cnt[i];
        // this camera's counts: three consecutive cells per thread
        const int* cnt = cell_cnt + cam * GRID_CELLS;
        k0 = cnt[c3]; k1 = cnt[c3 + 1]; k2 = cnt[c3 + 2];   // (3 x 1024 = GRID_CELLS)
    } else {
        // no counts in HBM (k_frame_fill_head): the features in front of this camera that lie inside their grids are counted from
        // their cell indices, this camera's cells in LDS
        for (int i = tid; i < f0; i += 1024) before += cell_of[i] >= 0 ? 1 : 0;
        s_fill[c3] = 0; s_fill[c3 + 1] = 0; s_fill[c3 + 2] = 0;
        __syncthreads();
        for (int i = tid; i < nf; i += 1024) {
            const int cell = cell_of[f0 + i];
            if (cell >= 0) atomicAdd(&s_fill[cell - cam * GRID_CELLS], 1);
        }
        __syncthreads();
        k0 = s_fill[c3]; k1 = s_fill[c3 + 1]; k2 = s_fill[c3 + 2];
    }
    {
        int b = before;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off);
        if (lane == 0) s_w[wave] = b;
    }
    __syncthreads();
    if (tid == 0) { int b = 0; for (int w = 0; w < 16; ++w) b += s_w[w]; s_before = b; }
    const int mine = k0 + k1 + k2;
    const int incl = wave_incl_scan(mine);
    __syncthreads();                                   // (s_w is read above, rewritten below)
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int run = 0;
    for (int w = 0; w < wave; ++w) run += s_w[w];
    const int e0 = run + incl - mine;
    s_start[c3] = e0; s_start[c3 + 1] = e0 + k0; s_start[c3 + 2] = e0 + k0 + k1;
    s_fill[c3] = e0; s_fill[c3 + 1] = e0 + k0; s_fill[c3 + 2] = e0 + k0 + k1;
    if (tid == 1023) s_start[GRID_CELLS] = e0 + mine;
    const int base = s_before;
    int* gs = cell_start + cam * GRID_CELLS;
    gs[c3] = base + e0; gs[c3 + 1] = base + e0 + k0; gs[c3 + 2] = base + e0 + k0 + k1;
    __syncthreads();
    const int n_in = s_start[GRID_CELLS];
    if (cam == n_cams - 1 && tid == 0) cell_start[n_cams * GRID_CELLS] = base + n_in;
    // scatter (LDS atomics: arbitrary order inside a cell), then every cell in ascending order
    for (int i = tid; i < nf; i += 1024) {
        const int cell = cell_of[f0 + i];
        if (cell >= 0) s_items[atomicAdd(&s_fill[cell - cam * GRID_CELLS], 1)] = (unsigned short)i;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int s = s_start[c3 + k], e = s_start[c3 + k + 1];
        for (int i = s + 1; i < e; ++i) {
            const unsigned short v = s_items[i];
            int j = i - 1;
            while (j >= s && s_items[j] > v) { s_items[j + 1] = s_items[j]; --j; }
            s_items[j + 1] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < n_in; i += 1024) items[base + i] = f0 + (int)s_items[i];
}

}  // namespace

int morb::frame_raise_lds_limit() {
    MORB_HIP(hipFuncSetAttribute((const void*)k_frame_build_small, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return ORB_OK;
}

int morb::phases_frame_build(unsigned long long* out64) {
#ifdef MORB_PHASE_CLOCKS
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_fb), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
    (void)out64; return -1;
#endif
}

FrameBufs* morb::take_bufs(orbm_matcher* m) {
    if (!m->pool.empty()) { FrameBufs* b = m->pool.back(); m->pool.pop_back(); return b; }
    return new FrameBufs();
}

int morb::reserve_frame(FrameBufs* b, int n, int n_cams) {
    const size_t nn = (size_t)std::max(n, 1);
    const int ncell = n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    int rc;
    if ((rc = b->d_x.reserve(nn)) || (rc = b->d_y.reserve(nn)) || (rc = b->d_ur.reserve(nn)) || (rc = b->d_depth.reserve(nn)) ||
        (rc = b->d_ang.reserve(nn)) || (rc = b->d_oct.reserve(nn)) || (rc = b->d_desc.reserve(nn * 32 + ORBM_BLOCK_TRAILER)) ||
        (rc = b->d_kps.reserve(nn)) || (rc = b->d_cell_start.reserve(ncell + 1)) || (rc = b->d_items.reserve(nn)) ||
        (rc = b->d_cell_of.reserve(nn)) || (rc = b->d_cursor.reserve(ncell + 1)) || (rc = b->d_cam_start.reserve(n_cams + 1)) ||
        (rc = b->d_ntotal.reserve(4)) ||
        (rc = b->d_cams.reserve(n_cams)))
        return rc;
    return ORB_OK;
}

// host mirrors of a device-built frame (octave / angle / grid), fetched once on demand
int morb::ensure_host_copies(const orbm_frame* f) {
    if (f->host_valid) return ORB_OK;
    orbm_matcher* m = f->owner;
    const int n = f->n_total, ncell = f->n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    f->octave.assign(std::max(n, 1), 0); f->angle.assign(std::max(n, 1), 0.f);
    f->cell_start.assign(ncell + 1, 0); f->items.assign(std::max(n, 1), 0);
    if (n) {
        MORB_HIP(hipMemcpyAsync(f->octave.data(), f->b->d_oct.p, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
        MORB_HIP(hipMemcpyAsync(f->angle.data(), f->b->d_ang.p, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
        MORB_HIP(hipMemcpyAsync(f->items.data(), f->b->d_items.p, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    }
    MORB_HIP(hipMemcpyAsync(f->cell_start.data(), f->b->d_cell_start.p, (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
    f->host_valid = true;
    return ORB_OK;
}

// ------------------------------------------------------------------------------------------------ frames
int orbm_frame_create(orbm_matcher* m, const orbm_frame_desc* f, orbm_frame** out) {
    MORB_ARG(m && f && out);
    MORB_ARG(f->n_total >= 0 && f->n_cams >= 1);
    MORB_ARG(f->max_x > f->min_x && f->max_y > f->min_y);
    const int n = f->n_total;
    MORB_ARG(n == 0 || (f->un_x && f->un_y && f->octave && f->angle && f->uright && f->cam_of && f->local_of && f->desc));
    MORB_HIP(hipSetDevice(m->device));
    orbm_frame* F = new orbm_frame();
    F->owner = m; F->n_total = n; F->n_cams = f->n_cams;
    F->minX = f->min_x; F->minY = f->min_y; F->maxX = f->max_x; F->maxY = f->max_y;
    F->invW = (float)ORBM_GRID_COLS / (f->max_x - f->min_x);  // reference src/Frame.cc:271-272
    F->invH = (float)ORBM_GRID_ROWS / (f->max_y - f->min_y);
    F->octave.assign(f->octave, f->octave + n);
    F->angle.assign(f->angle, f->angle + n);

    // Grid: counting sort by cell keeps ascending global index inside each cell (reference src/Frame.cc:373-393
    // pushes cam 1 then cam 2 in index order).  Insertion cell uses round(), not floor() (src/Frame.cc:634-635).
    const int ncell = f->n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    std::vector<int32_t> cell_of(n);
    F->cell_start.assign(ncell + 1, 0);
    F->cam_start.assign(f->n_cams + 1, 0);
    int last_cam = 0;
    for (int g = 0; g < n; g++) {
        const int px = (int)roundf((f->un_x[g] - F->minX) * F->invW);
        const int py = (int)roundf((f->un_y[g] - F->minY) * F->invH);
        const int cam = f->cam_of[g];
        if (cam >= 0 && cam < f->n_cams) F->cam_start[cam + 1]++;
        if (cam < last_cam || cam >= f->n_cams) F->camera_major = false;   // (interleaved cameras, or a feature of no camera: cam_start[] is counts only)
        else last_cam = cam;
        if (px < 0 || px >= ORBM_GRID_COLS || py < 0 || py >= ORBM_GRID_ROWS || cam < 0 || cam >= f->n_cams) {
            cell_of[g] = -1;
            continue;
        }
        cell_of[g] = (cam * ORBM_GRID_COLS + px) * ORBM_GRID_ROWS + py;
        F->cell_start[cell_of[g] + 1]++;
    }
    for (int c = 0; c < f->n_cams; c++) F->cam_start[c + 1] += F->cam_start[c];
    for (int c = 0; c < ncell; c++) F->cell_start[c + 1] += F->cell_start[c];
    F->items.assign(std::max(n, 1), 0);
    {
        std::vector<int32_t> cursor(F->cell_start.begin(), F->cell_start.end() - 1);
        for (int g = 0; g < n; g++)
            if (cell_of[g] >= 0) F->items[cursor[cell_of[g]]++] = g;
    }
    F->host_valid = true;

    F->b = take_bufs(m);
    int rc = reserve_frame(F->b, n, f->n_cams);
    if (rc) { orbm_frame_destroy(F); return rc; }
    hipStream_t st = m->stream;
    // everything goes through ONE staging block: x, y, uright, angle, octave, items (n dwords each), descriptors re-laid in
    // global-index order so the kernels gather with one index (8n dwords), cell starts, camera starts
    const size_t nn = (size_t)n;
    const size_t words = 6 * nn + 8 * nn + (size_t)(ncell + 1) + (size_t)(f->n_cams + 1);
    if (m->stage_f_busy) { MORB_HIP(hipEventSynchronize(m->ev_stage_f)); m->stage_f_busy = false; }   // (the previous frame's unpack)
    if ((rc = m->stage_f.reserve(words * 4))) { orbm_frame_destroy(F); return rc; }
    uint32_t* w = reinterpret_cast<uint32_t*>(m->stage_f.p);
    UnpackPlan P;
    P.src = reinterpret_cast<const uint32_t*>(m->stage_f.dp);
    size_t pos = 0;
    int sec = 0;
    auto section = [&](const void* src, size_t dwords, void* dst) {
        if (dwords && src) memcpy(w + pos, src, dwords * 4);
        pos += dwords;
        P.dst[sec] = static_cast<uint32_t*>(dst); P.end[sec] = (int)pos; ++sec;
    };
    section(f->un_x, nn, F->b->d_x.p); section(f->un_y, nn, F->b->d_y.p); section(f->uright, nn, F->b->d_ur.p);
    section(f->angle, nn, F->b->d_ang.p); section(f->octave, nn, F->b->d_oct.p); section(F->items.data(), nn, F->b->d_items.p);
    for (int g = 0; g < n; g++) memcpy(w + pos + (size_t)g * 8, f->desc[f->cam_of[g]] + (size_t)f->local_of[g] * 32, 32);
    section(nullptr, 8 * nn, F->b->d_desc.p);
    section(F->cell_start.data(), (size_t)(ncell + 1), F->b->d_cell_start.p);
    section(F->cam_start.data(), (size_t)(f->n_cams + 1), F->b->d_cam_start.p);
    m->stage_f.publish();
    hipLaunchKernelGGL(k_frame_unpack, dim3((unsigned)std::min<size_t>((words + 255) / 256, 512)), dim3(256), 0, st, P);
    MORB_HIP(hipGetLastError());
    MORB_HIP(hipEventRecord(m->ev_stage_f, st));
    m->stage_f_busy = true;   // nothing is waited for here: whatever uses the frame is ordered behind the unpack on the stream
    *out = F;
    return ORB_OK;
}




// orbm_frame_create with the grid built on the device and, for the cameras whose descriptors are still in HBM, without the
// descriptors crossing the bus again.  Frames beyond the single-workgroup build (8192 features, 4 cameras) take
// orbm_frame_create (f->desc must be valid for every camera either way).
int orbm_frame_create_resident(orbm_matcher* m, const orbm_frame_desc* f, const uint8_t* const* d_desc, orbm_frame** out) {
    MORB_ARG(m && f && out);
    MORB_ARG(f->n_total >= 0 && f->n_cams >= 1);
    MORB_ARG(f->max_x > f->min_x && f->max_y > f->min_y);
    const int n = f->n_total, n_cams = f->n_cams;
    const int ncell = n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    const size_t lds_small = (size_t)2 * (ncell + 1) * sizeof(int) + (size_t)8192 * sizeof(unsigned short);
    if (n == 0 || n > 8192 || n_cams > 4 || lds_small > 150 * 1024) return orbm_frame_create(m, f, out);
    MORB_ARG(f->un_x && f->un_y && f->octave && f->angle && f->uright && f->cam_of && f->local_of && f->desc);
    MORB_HIP(hipSetDevice(m->device));
    // per-camera row counts (the highest row a feature names + 1) and the camera starts of the global order
    int rows[4] = {0, 0, 0, 0}, count[4] = {0, 0, 0, 0}, last_cam = 0;
    for (int g = 0; g < n; ++g) {
        const int c = f->cam_of[g], l = f->local_of[g];
        // a feature of no camera, or cameras interleaved in the global order: the device build derives every feature's place from the
        // per-camera counts (it would drop the last features / misplace them) -- such a frame takes the host-built form
        if (c < 0 || c >= n_cams || c < last_cam) return orbm_frame_create(m, f, out);
        last_cam = c;
        MORB_ARG(l >= 0 && l < (1 << 24));
        ++count[c];
        if (l + 1 > rows[c]) rows[c] = l + 1;
    }
    StagedFill S;
    memset(&S, 0, sizeof(S));
    int host_rows = 0;
    for (int c = 0; c < n_cams; ++c) {
        const bool resident = d_desc && d_desc[c] && ((uintptr_t)d_desc[c] & 15) == 0;
        S.d_desc[c] = resident ? reinterpret_cast<const uint4*>(d_desc[c]) : nullptr;
        S.host_row0[c] = host_rows;
        if (!resident) { MORB_ARG(rows[c] == 0 || f->desc[c]); host_rows += rows[c]; }
    }
    int rc;
    orbm_frame* F = nullptr;
    if (m->stage_f_busy) { MORB_HIP(hipEventSynchronize(m->ev_stage_f)); m->stage_f_busy = false; }   // (before the frame exists: nothing to give back on failure)
    if ((rc = frame_shell(m, n, n_cams, f->min_x, f->min_y, f->max_x, f->max_y, false, &F))) return rc;
    F->device_built = false;   // (no keypoint records: orbm_frame_download refuses them, as for orbm_frame_create)
    int base = 0;
    for (int c = 0; c < n_cams; ++c) { F->cam_start[c] = base; base += count[c]; }
    F->cam_start[n_cams] = base;
    const size_t nn = (size_t)n;
    const size_t head = (6 * nn + (size_t)n_cams + 1 + 3) & ~(size_t)3;          // descriptor rows start 16-byte aligned
    const size_t words = head + 8 * (size_t)host_rows;
    if ((rc = m->stage_f.reserve(words * 4))) { orbm_frame_destroy(F); return rc; }
    uint32_t* w = reinterpret_cast<uint32_t*>(m->stage_f.p);
    memcpy(w, f->un_x, nn * 4); memcpy(w + nn, f->un_y, nn * 4); memcpy(w + 2 * nn, f->uright, nn * 4);
    memcpy(w + 3 * nn, f->angle, nn * 4); memcpy(w + 4 * nn, f->octave, nn * 4);
    for (int g = 0; g < n; ++g) {
        const int c = f->cam_of[g];
        w[5 * nn + g] = (c < 0 || c >= n_cams) ? 0xff000000u : ((uint32_t)c << 24) | (uint32_t)f->local_of[g];
    }
    for (int c = 0; c <= n_cams; ++c) w[6 * nn + c] = (uint32_t)F->cam_start[c];
    for (int c = 0; c < n_cams; ++c)
        if (!S.d_desc[c] && rows[c]) memcpy(w + head + 8 * (size_t)S.host_row0[c], f->desc[c], (size_t)rows[c] * 32);
    m->stage_f.publish();
    S.stage = reinterpret_cast<const uint32_t*>(m->stage_f.dp);
    S.desc_off = (int)head;
    CamFeat4 c4;
    memset(&c4, 0, sizeof(c4));
    HostMirror hm{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, orb_calibration{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    hipLaunchKernelGGL(k_frame_build_small, dim3(1), dim3(1024), lds_small, m->stream, c4, F->b->d_cam_start.p, (const int*)nullptr,
                       F->b->d_ntotal.p, n_cams, n, 0.f, F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p,
                       F->b->d_depth.p, F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_start.p,
                       F->b->d_items.p, hm, (const int*)nullptr, F->desc_rows, S);
    if (hipGetLastError() != hipSuccess) { orbm_frame_destroy(F); morb::set_error("k_frame_build_small launch failed"); return ORB_E_HIP; }
    if (hipEventRecord(m->ev_stage_f, m->stream) != hipSuccess) {   // (the staging block would be rewritten under the kernel: wait, give the frame back)
        (void)hipStreamSynchronize(m->stream); orbm_frame_destroy(F);
        morb::set_error("hipEventRecord failed behind the frame build"); return ORB_E_HIP;
    }
    m->stage_f_busy = true;
    *out = F;
    return ORB_OK;
}

int orbm_frame_from_device(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                           float max_x, float max_y, orbm_frame** out) {
    MORB_ARG(out != nullptr);
    *out = nullptr;
    return frame_from_device_impl(m, cams, n_cams, mbf, min_x, min_y, max_x, max_y, nullptr, out);
}



// Frame shell with storage for `n` features, no kernel launched yet.
int morb::frame_shell(orbm_matcher* m, int n, int n_cams, float min_x, float min_y, float max_x, float max_y, bool counts_on_device,
                       orbm_frame** out) {
    orbm_frame* F = new orbm_frame();
    F->owner = m; F->n_total = n; F->n_cams = n_cams; F->device_built = true; F->counts_on_device = counts_on_device;
    F->minX = min_x; F->minY = min_y; F->maxX = max_x; F->maxY = max_y;
    F->invW = (float)ORBM_GRID_COLS / (max_x - min_x);
    F->invH = (float)ORBM_GRID_ROWS / (max_y - min_y);
    F->cam_start.assign(n_cams + 1, 0);
    F->desc_rows = std::max(std::max(n, 1), m->frame_min_rows);
    F->b = take_bufs(m);
    int rc = reserve_frame(F->b, F->desc_rows, n_cams);
    if (rc) { orbm_frame_destroy(F); return rc; }
    *out = F;
    return ORB_OK;
}

// orbf_step: the frame the extractor's describe kernel is about to fill (capacity-sized) and the sink describing it
int morb::frame_prepare_sink(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                              float max_x, float max_y, orbm_frame** out, FrameSink* sink) {
    MORB_ARG(n_cams >= 1 && n_cams <= 4 && max_x > min_x && max_y > min_y);
    int n = 0;
    for (int c = 0; c < n_cams; ++c) n += cams[c].n;
    orbm_frame* F = nullptr;
    int rc = frame_shell(m, n, n_cams, min_x, min_y, max_x, max_y, true, &F);
    if (rc) return rc;
    memset(sink, 0, sizeof(*sink));
    sink->x = F->b->d_x.p; sink->y = F->b->d_y.p; sink->ur = F->b->d_ur.p; sink->depth = F->b->d_depth.p; sink->ang = F->b->d_ang.p;
    sink->oct = F->b->d_oct.p; sink->kps = F->b->d_kps.p; sink->desc = reinterpret_cast<uint32_t*>(F->b->d_desc.p);
    sink->cell_of = F->b->d_cell_of.p;
    sink->h_ur = m->mirror_ur; sink->h_depth = m->mirror_depth; sink->h_unx = m->mirror_unx; sink->h_uny = m->mirror_uny;
    sink->calib = m->calib;
    for (int c = 0; c < n_cams; ++c) { sink->cam_depth[c] = cams[c].d_depth; sink->cam_depth_stride[c] = cams[c].depth_stride; }
    sink->mbf = mbf; sink->minX = F->minX; sink->minY = F->minY; sink->invW = F->invW; sink->invH = F->invH;
    *out = F;
    return ORB_OK;
}

// Copies a finished frame's per-feature results into a caller's mapped pinned buffers (orbf_step's result set).  The
// describe / frame kernels can write those mirrors themselves, but a kernel that stores to host memory does not complete
// before the stores have crossed PCIe (5.6 us for 2000 keypoints + descriptors, measured on k_describe); a step whose
// matching waits behind the extraction on the same stream runs this copy on the side stream instead, next to project +
// resolve.  Row count from the frame's device-side total when it has one.
__global__ __launch_bounds__(256) void k_mirror_frame(MirrorJob J) { mirror_rows(J, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256); }

int morb::frame_mirror_enqueue(orbm_frame* F, void* stream, orb_keypoint* h_kps, uint8_t* h_desc, float* h_unx, float* h_uny,
                               float* h_ur, float* h_depth) {
    if (F && F->n_total <= 0) return ORB_OK;
    MirrorJob J;
    int rc = frame_mirror_job(F, h_kps, h_desc, h_unx, h_uny, h_ur, h_depth, &J);
    if (rc) return rc;
    const int blocks = std::min(64, (F->n_total * 8 + 255) / 256);
    hipLaunchKernelGGL(k_mirror_frame, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

int morb::frame_mirror_job(orbm_frame* F, orb_keypoint* h_kps, uint8_t* h_desc, float* h_unx, float* h_uny, float* h_ur, float* h_depth,
                           MirrorJob* out) {
    MORB_ARG(F && F->b && h_kps && h_desc && h_unx && h_uny && h_ur && h_depth && out);
    static_assert(sizeof(orb_keypoint) % 4 == 0, "keypoints are copied as dwords");
    MirrorJob& J = *out;
    J.kps = reinterpret_cast<const uint32_t*>(F->b->d_kps.p); J.desc = reinterpret_cast<const uint32_t*>(F->b->d_desc.p);
    J.x = reinterpret_cast<const uint32_t*>(F->b->d_x.p); J.y = reinterpret_cast<const uint32_t*>(F->b->d_y.p);
    J.ur = reinterpret_cast<const uint32_t*>(F->b->d_ur.p); J.depth = reinterpret_cast<const uint32_t*>(F->b->d_depth.p);
    J.h_kps = reinterpret_cast<uint32_t*>(h_kps); J.h_desc = reinterpret_cast<uint32_t*>(h_desc);
    J.h_x = reinterpret_cast<uint32_t*>(h_unx); J.h_y = reinterpret_cast<uint32_t*>(h_uny);
    J.h_ur = reinterpret_cast<uint32_t*>(h_ur); J.h_depth = reinterpret_cast<uint32_t*>(h_depth);
    J.n_dev = F->counts_on_device ? F->b->d_ntotal.p : nullptr; J.n_host = F->n_total;
    return ORB_OK;
}

// the sink of an existing (persistent) frame
int morb::frame_sink_of(orbm_matcher* m, orbm_frame* F, const orbm_cam_features* cams, int n_cams, float mbf, FrameSink* sink) {
    MORB_ARG(F && n_cams >= 1 && n_cams <= 4);
    memset(sink, 0, sizeof(*sink));
    sink->x = F->b->d_x.p; sink->y = F->b->d_y.p; sink->ur = F->b->d_ur.p; sink->depth = F->b->d_depth.p; sink->ang = F->b->d_ang.p;
    sink->oct = F->b->d_oct.p; sink->kps = F->b->d_kps.p; sink->desc = reinterpret_cast<uint32_t*>(F->b->d_desc.p);
    sink->cell_of = F->b->d_cell_of.p;
    sink->h_ur = m->mirror_ur; sink->h_depth = m->mirror_depth; sink->h_unx = m->mirror_unx; sink->h_uny = m->mirror_uny;
    sink->calib = m->calib;
    for (int c = 0; c < n_cams; ++c) { sink->cam_depth[c] = cams[c].d_depth; sink->cam_depth_stride[c] = cams[c].depth_stride; }
    sink->mbf = mbf; sink->minX = F->minX; sink->minY = F->minY; sink->invW = F->invW; sink->invH = F->invH;
    return ORB_OK;
}

int morb::frame_from_device_impl(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                                  float max_x, float max_y, const int* d_counts, orbm_frame** out, bool sink_filled) {
    MORB_ARG(m && cams && out && n_cams >= 1 && n_cams <= 64 && max_x > min_x && max_y > min_y);
    MORB_HIP(hipSetDevice(m->device));
    int n = 0;
    for (int c = 0; c < n_cams; ++c) {
        MORB_ARG(cams[c].n >= 0 && (cams[c].n == 0 || (cams[c].d_kps && cams[c].d_desc)));
        MORB_ARG(((uintptr_t)cams[c].d_desc & 15) == 0 && ((uintptr_t)cams[c].d_kps & 3) == 0);
        n += cams[c].n;
    }
    orbm_frame* F = *out;  // non-NULL: a persistent frame of the same capacity is (re)filled
    int rc;
    if (F) {
        MORB_ARG(F->n_total == n && F->n_cams == n_cams && d_counts && F->desc_rows >= n);
    } else {
        MORB_ARG(!sink_filled);
        if ((rc = frame_shell(m, n, n_cams, min_x, min_y, max_x, max_y, d_counts != nullptr, &F))) return rc;
    }
    rc = ORB_OK;
    const size_t slot = 64 * sizeof(CamFeat) + 65 * sizeof(int) + 64 * sizeof(int);
    if (!rc) rc = m->h_ring.reserve(slot * 4);  // ring of 4 parameter blocks: the H2D copies below are asynchronous
    if (rc) { orbm_frame_destroy(F); return rc; }
    uint8_t* hs = m->h_ring.p + (size_t)(m->ring_pos++ & 3) * slot;
    CamFeat* hc = reinterpret_cast<CamFeat*>(hs);
    int* hstart = reinterpret_cast<int*>(hs + 64 * sizeof(CamFeat));
    int base = 0;
    for (int c = 0; c < n_cams; ++c) {
        hc[c].kps = cams[c].d_kps; hc[c].desc = (const uint4*)cams[c].d_desc; hc[c].depth = cams[c].d_depth;
        hc[c].depth_stride = cams[c].depth_stride; hc[c].n = cams[c].n; hc[c].base = base;
        F->cam_start[c] = base;
        base += cams[c].n;
    }
    F->cam_start[n_cams] = base;
    for (int c = 0; c <= n_cams; ++c) hstart[c] = F->cam_start[c];
    const int ncell = n_cams * ORBM_GRID_COLS * ORBM_GRID_ROWS;
    hipStream_t st = m->stream;
    const size_t lds_small = (size_t)2 * (ncell + 1) * sizeof(int) + (size_t)8192 * sizeof(unsigned short);
    const bool small = n > 0 && n <= 8192 && n_cams <= 4 && lds_small <= 150 * 1024;
    MORB_ARG(!sink_filled || small);
    const bool one_head = !small && d_counts && n_cams <= 32;   // (k_frame_head: table, counts and cleared counters in one launch)
    if (!small && !one_head) {
        MORB_HIP(hipMemcpyAsync(F->b->d_cams.p, hc, (size_t)n_cams * sizeof(CamFeat), hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->b->d_cam_start.p, hstart, (size_t)(n_cams + 1) * 4, hipMemcpyHostToDevice, st));
    }
    HostMirror hm{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, m->calib};
    if (m->mirror_kps) { hm.kps = m->mirror_kps; hm.desc = (uint4*)m->mirror_desc; }
    if (m->mirror_ur) { hm.ur = m->mirror_ur; hm.depth = m->mirror_depth; }
    if (m->mirror_unx) { hm.unx = m->mirror_unx; hm.uny = m->mirror_uny; }
    if (small) {
        CamFeat4 c4;
        memset(&c4, 0, sizeof(c4));
        for (int c = 0; c < n_cams; ++c) c4.c[c] = hc[c];
        hipLaunchKernelGGL(k_frame_build_small, dim3(1), dim3(1024), lds_small, st, c4, F->b->d_cam_start.p, d_counts, F->b->d_ntotal.p,
                           n_cams, n, mbf,
                           F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p, F->b->d_depth.p,
                           F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_start.p,
                           F->b->d_items.p, hm, sink_filled ? (const int*)F->b->d_cell_of.p : nullptr, F->desc_rows, StagedFill{});
    } else {
        const int* n_dev = nullptr;
        bool grid_cam = true;   // (every camera's features fit k_grid_cam's LDS list: capacities are upper bounds of the counts)
        for (int c = 0; c < n_cams; ++c) grid_cam = grid_cam && cams[c].n <= GRID_CAM_MAX;
        if (one_head && grid_cam && n) {
            // table + fill in one launch, cells counted by k_grid_cam itself (k_frame_fill_head)
            CamFeat32 H;
            memset(&H, 0, sizeof(H));
            for (int c = 0; c < n_cams; ++c) H.c[c] = hc[c];
            hipLaunchKernelGGL(k_frame_fill_head, dim3((n + 255) / 256), dim3(256), 0, st, H, F->b->d_cams.p, n_cams, d_counts,
                               F->b->d_cam_start.p, F->b->d_ntotal.p, reinterpret_cast<int*>(F->b->d_desc.p + (size_t)F->desc_rows * 32),
                               mbf, F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p, F->b->d_depth.p,
                               F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_of.p, hm);
            hipLaunchKernelGGL(k_grid_cam, dim3(n_cams), dim3(1024), 0, st, (const int*)nullptr, (const int*)F->b->d_cell_of.p,
                               (const int*)F->b->d_cam_start.p, n_cams, F->b->d_cell_start.p, F->b->d_items.p);
            MORB_HIP(hipGetLastError());
            *out = F;
            return ORB_OK;
        }
        if (one_head) {
            CamFeat32 H;
            memset(&H, 0, sizeof(H));
            for (int c = 0; c < n_cams; ++c) H.c[c] = hc[c];
            hipLaunchKernelGGL(k_frame_head, dim3(std::min(64, (ncell + 1 + 255) / 256)), dim3(256), 0, st, H, F->b->d_cams.p, n_cams, d_counts,
                               F->b->d_cam_start.p, F->b->d_ntotal.p, reinterpret_cast<int*>(F->b->d_desc.p + (size_t)F->desc_rows * 32),
                               F->b->d_cursor.p, ncell + 1);
            n_dev = F->b->d_ntotal.p;
        } else if (d_counts) {
            // counts still on the device (cams[c].n are capacities): the camera table is finished by a one-thread kernel
            hipLaunchKernelGGL(k_cams_from_counts, dim3(1), dim3(64), 0, st, F->b->d_cams.p, n_cams, d_counts, F->b->d_cam_start.p,
                               F->b->d_ntotal.p, reinterpret_cast<int*>(F->b->d_desc.p + (size_t)F->desc_rows * 32));
            n_dev = F->b->d_ntotal.p;
        } else {
            // count trailer of the descriptor block (the single-workgroup build writes it itself)
            int* hcnt = reinterpret_cast<int*>(hs + 64 * sizeof(CamFeat) + 65 * sizeof(int));
            for (int c = 0; c < n_cams; ++c) hcnt[c] = cams[c].n;
            MORB_HIP(hipMemcpyAsync(F->b->d_desc.p + (size_t)F->desc_rows * 32, hcnt, (size_t)n_cams * sizeof(int), hipMemcpyHostToDevice, st));
        }
        if (!one_head) MORB_HIP(hipMemsetAsync(F->b->d_cursor.p, 0, (size_t)(ncell + 1) * 4, st));  // used as the per-cell counter first
        if (n) {
            hipLaunchKernelGGL(k_frame_fill, dim3((n + 255) / 256), dim3(256), 0, st, (const CamFeat*)F->b->d_cams.p, n_cams, n, mbf,
                               F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p, F->b->d_depth.p,
                               F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_of.p,
                               F->b->d_cursor.p, hm, n_dev);
        }
        if (grid_cam) {
            hipLaunchKernelGGL(k_grid_cam, dim3(n_cams), dim3(1024), 0, st, (const int*)F->b->d_cursor.p, (const int*)F->b->d_cell_of.p,
                               (const int*)F->b->d_cam_start.p, n_cams, F->b->d_cell_start.p, F->b->d_items.p);
            MORB_HIP(hipGetLastError());
            *out = F;
            return ORB_OK;
        }
        // counts live in d_cursor; scan them into d_cell_start and leave d_cursor = running insert positions
        hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, st, (const int*)F->b->d_cursor.p, ncell, F->b->d_cell_start.p,
                           F->b->d_cursor.p);
        if (n) {
            hipLaunchKernelGGL(k_scatter_cells, dim3((n + 255) / 256), dim3(256), 0, st, (const int*)F->b->d_cell_of.p, n,
                               F->b->d_cursor.p, F->b->d_items.p, n_dev);
            hipLaunchKernelGGL(k_sort_cells, dim3((ncell + 255) / 256), dim3(256), 0, st, (const int*)F->b->d_cell_start.p, ncell,
                               F->b->d_items.p);
        }
    }
    MORB_HIP(hipGetLastError());
    *out = F;
    return ORB_OK;
}

// the caller has synchronised and learnt the real counts
void morb::frame_set_counts(orbm_frame* F, const int* counts) {
    int base = 0;
    for (int c = 0; c < F->n_cams; ++c) { F->cam_start[c] = base; base += counts[c]; }
    F->cam_start[F->n_cams] = base;
    F->n_total = base;
    F->counts_on_device = false;
}


int orbm_frame_count(const orbm_frame* f) { return f ? f->n_total : ORB_E_ARG; }

int orbm_frame_download(orbm_matcher* m, const orbm_frame* f, orb_keypoint* kps, uint8_t* desc, float* uright, float* depth) {
    MORB_ARG(m && f);
    MORB_HIP(hipSetDevice(m->device));
    const size_t n = (size_t)f->n_total;
    if (n) {
        if (kps) {
            MORB_ARG(f->device_built);  // host-built frames were handed keypoint fields, not records
            MORB_HIP(hipMemcpyAsync(kps, f->b->d_kps.p, n * sizeof(orb_keypoint), hipMemcpyDeviceToHost, m->stream));
        }
        if (desc) MORB_HIP(hipMemcpyAsync(desc, f->b->d_desc.p, n * 32, hipMemcpyDeviceToHost, m->stream));
        if (uright) MORB_HIP(hipMemcpyAsync(uright, f->b->d_ur.p, n * 4, hipMemcpyDeviceToHost, m->stream));
        if (depth) {
            MORB_ARG(f->device_built);
            MORB_HIP(hipMemcpyAsync(depth, f->b->d_depth.p, n * 4, hipMemcpyDeviceToHost, m->stream));
        }
    }
    MORB_HIP(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

void orbm_frame_destroy(orbm_frame* f) {
    if (!f) return;
    if (f->b) {
        if (f->owner) {
            // kernels reading these buffers may still be queued: recycle only after the stream drained
            (void)hipSetDevice(f->owner->device);
            (void)hipStreamSynchronize(f->owner->stream);
            f->owner->pool.push_back(f->b);
        } else { f->b->release(); delete f->b; }
    }
    delete f;
}

int orbm_frame_grid(const orbm_frame* f, int32_t* cell_start, int32_t* items) {
    MORB_ARG(f && cell_start);
    int rc = ensure_host_copies(f);
    if (rc) return rc;
    memcpy(cell_start, f->cell_start.data(), f->cell_start.size() * 4);
    if (items && f->cell_start.back() > 0) memcpy(items, f->items.data(), (size_t)f->cell_start.back() * 4);
    return ORB_OK;
}

