// matcher.hip -- gfx950 kernels + C ABI of the ORB matcher (include/orbm.h).
//
// Kernels (all integer / bit work, no MFMA by design -- BASELINE.json north_star):
//   k_hamming_top2    K8/M1  exhaustive top-2 Hamming: one query per lane, references walked with wave-uniform
//                            (scalar-cache) loads, 16 waves per block each scanning 1/16 of the references,
//                            LDS merge.  VALU-bound (18 int ops / pair).  reference src/ORBmatcher.cc:287-321.
//   k_hamming_matrix  M2     full uint16 distance matrix: 8 references per lane held in VGPRs, queries walked with
//                            scalar loads, one 16-byte store per lane per query row.  HBM-write-bound.
//   k_project         K9/M3  projection-gated search: one wave per query walks the 64x48 grid cells of the window in
//                            the reference's visiting order, ballot-compacts the survivors in order and gathers their
//                            descriptors.  reference src/ORBmatcher.cc:3547-3592 + src/Frame.cc:574-629.
// The order-dependent part of SearchByProjection (first-come claims, rotation histogram) is resolved on the host
// from the ordered candidate lists.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <vector>

#include "../../include/orbm.h"
#include "orb_common.h"

namespace morb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): this library has no CPU path", hipGetErrorString(e));
        return ORB_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d devices)", device, n); return ORB_E_ARG; }
    hipDeviceProp_t prop;
    MORB_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return ORB_E_NO_DEVICE;
    }
    MORB_HIP(hipSetDevice(device));
    return ORB_OK;
}

}  // namespace morb

extern "C" const char* orb_last_error(void) { return morb::g_err; }

namespace {

using morb::DevBuf;
using morb::PinnedBuf;

// ------------------------------------------------------------------------------------------------ kernels
constexpr int TOP2_WAVES = 16;

// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32's accumulate operand; hipcc otherwise emits bcnt + add3 trees)
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ unsigned ham256_chain(const uint4& q0, const uint4& q1, const uint4& a, const uint4& b) {
    unsigned d = __popc(q0.x ^ a.x);
    d = bcnt_acc(q0.y ^ a.y, d); d = bcnt_acc(q0.z ^ a.z, d); d = bcnt_acc(q0.w ^ a.w, d);
    d = bcnt_acc(q1.x ^ b.x, d); d = bcnt_acc(q1.y ^ b.y, d); d = bcnt_acc(q1.z ^ b.z, d); d = bcnt_acc(q1.w ^ b.w, d);
    return d;
}

__device__ __forceinline__ int ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// grid.x = ceil(nq/64), grid.y = S reference slices (S == 1: final results; S > 1: partials for k_top2_merge)
__global__ __launch_bounds__(64 * TOP2_WAVES) void k_hamming_top2(const uint4* __restrict__ q, int nq,
                                                                 const uint4* __restrict__ r, int nr,
                                                                 int* __restrict__ best_idx,
                                                                 int* __restrict__ best_dist,
                                                                 int* __restrict__ second_dist) {
    __shared__ int sb[TOP2_WAVES][64], ss[TOP2_WAVES][64], si[TOP2_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.x * 64 + lane;
    const int qc = qi < nq ? qi : nq - 1;
    const uint4 q0 = q[2 * qc], q1 = q[2 * qc + 1];

    const int S = gridDim.y;
    const int slice = (nr + S - 1) / S;
    const int s0 = blockIdx.y * slice, s1 = min(nr, s0 + slice);
    const int chunk = (max(s1 - s0, 0) + TOP2_WAVES - 1) / TOP2_WAVES;
    const int j0 = s0 + wave * chunk, j1 = min(s1, j0 + chunk);

    int b = 256, s = 256, bi = -1;
    // second = 2nd smallest with multiplicity, best index = first minimum (strict '<' chain, ORBmatcher.cc:311-320)
#define TOP2_UPDATE(d, j) do { s = min(s, max(b, (d))); bi = (d) < b ? (j) : bi; b = min(b, (d)); } while (0)
    int j = j0;
    for (; j + 4 <= j1; j += 4) {  // 4 references (128 B through the scalar cache) per trip
        const uint4 a0 = r[2 * j], a1 = r[2 * j + 1], b0 = r[2 * j + 2], b1 = r[2 * j + 3];
        const uint4 c0 = r[2 * j + 4], c1 = r[2 * j + 5], e0 = r[2 * j + 6], e1 = r[2 * j + 7];
        const int d0 = (int)ham256_chain(a0, a1, q0, q1), d1 = (int)ham256_chain(b0, b1, q0, q1);
        const int d2 = (int)ham256_chain(c0, c1, q0, q1), d3 = (int)ham256_chain(e0, e1, q0, q1);
        TOP2_UPDATE(d0, j); TOP2_UPDATE(d1, j + 1); TOP2_UPDATE(d2, j + 2); TOP2_UPDATE(d3, j + 3);
    }
    for (; j < j1; ++j) {
        const uint4 a0 = r[2 * j], a1 = r[2 * j + 1];  // wave-uniform address -> scalar loads
        const int d = (int)ham256_chain(a0, a1, q0, q1);
        TOP2_UPDATE(d, j);
    }
#undef TOP2_UPDATE
    sb[wave][lane] = b; ss[wave][lane] = s; si[wave][lane] = bi;
    __syncthreads();
    if (wave == 0 && qi < nq) {
        int B = 256, Sd = 256, I = -1;
#pragma unroll
        for (int w = 0; w < TOP2_WAVES; ++w) {  // wave order == reference index order: earlier index wins ties
            const int b2 = sb[w][lane], s2 = ss[w][lane], i2 = si[w][lane];
            Sd = min(min(Sd, s2), max(B, b2));
            I = b2 < B ? i2 : I;
            B = min(B, b2);
        }
        const size_t o = (size_t)blockIdx.y * nq + qi;
        best_idx[o] = I; best_dist[o] = B; second_dist[o] = Sd;
    }
}

__global__ void k_top2_merge(const int* __restrict__ p_idx, const int* __restrict__ p_best,
                             const int* __restrict__ p_second, int S, int nq, int* __restrict__ best_idx,
                             int* __restrict__ best_dist, int* __restrict__ second_dist) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int B = 256, Sd = 256, I = -1;
    for (int k = 0; k < S; ++k) {
        const int b2 = p_best[(size_t)k * nq + qi], s2 = p_second[(size_t)k * nq + qi], i2 = p_idx[(size_t)k * nq + qi];
        Sd = min(min(Sd, s2), max(B, b2));
        I = b2 < B ? i2 : I;
        B = min(B, b2);
    }
    best_idx[qi] = I; best_dist[qi] = B; second_dist[qi] = Sd;
}

// One wave = 512 consecutive references (8 per lane, 64 VGPRs), one block = 4 such tiles; grid.y walks the queries in
// chunks.  Queries arrive through the scalar cache (wave-uniform address), two per iteration with the next pair
// prefetched into SGPRs while the current pair is being processed, so the loop body is pure VALU + one 16-byte store
// per lane per query row: 8 v_xor + 8 v_bcnt (accumulating form) per pair.
constexpr int MAT_REFS_PER_LANE = 8;
constexpr int MAT_REFS_PER_WAVE = 64 * MAT_REFS_PER_LANE;

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Makes hipcc treat the eight dwords as used here (it inserts the s_waitcnt for their scalar loads at this point).
__device__ __forceinline__ void touch_sgpr(const uint4& a, const uint4& b) {
    asm volatile("" ::"s"(a.x), "s"(a.y), "s"(a.z), "s"(a.w), "s"(b.x), "s"(b.y), "s"(b.z), "s"(b.w));
}

// FULL: the wave's 512-reference tile lies completely inside [0, nr) and rows are 16-byte aligned: one unconditional
// non-temporal dwordx4 store per lane (no exec-mask branch, so the loop stays one basic block and the scalar prefetch
// below cannot be sunk past it).  Otherwise: guarded 2-byte stores (only the last partial tile / odd nr).
template <bool FULL>
__device__ __forceinline__ void mat_store_row(uint16_t* __restrict__ row, const unsigned (&d)[MAT_REFS_PER_LANE], int r0, int nr) {
    if (FULL) {
        v4u o;
        o.x = d[0] | (d[1] << 16); o.y = d[2] | (d[3] << 16);
        o.z = d[4] | (d[5] << 16); o.w = d[6] | (d[7] << 16);
        *reinterpret_cast<v4u*>(row) = o;  // plain store: measured 3-8 % faster than `nt` for this pattern
    } else {
#pragma unroll
        for (int k = 0; k < MAT_REFS_PER_LANE; ++k)
            if (r0 + k < nr) row[k] = (uint16_t)d[k];
    }
}

// Software pipeline over queries with two SGPR sets (x*, y*): while one query is processed the next one's 32 bytes are
// in flight through the scalar cache.  SMEM returns out of order, so the only wait is lgkmcnt(0): `touch_sgpr` forces
// that wait for the CURRENT set BEFORE the next load is issued; the sched_barriers keep hipcc from moving the load.
template <bool FULL>
__device__ __forceinline__ void mat_rows(const uint4* __restrict__ q, int qa, int qb, const uint4 (&ra)[MAT_REFS_PER_LANE],
                                         const uint4 (&rb)[MAT_REFS_PER_LANE], uint16_t* __restrict__ out, int nr, int r0) {
    uint4 x0 = q[2 * qa], x1 = q[2 * qa + 1], y0, y1;
    unsigned d[MAT_REFS_PER_LANE];
    const int npairs = (qb - qa) >> 1;
    int qi = qa;
    for (int p = 0; p < npairs; ++p, qi += 2) {
        touch_sgpr(x0, x1);
        __builtin_amdgcn_sched_barrier(0);
        y0 = q[2 * (qi + 1)]; y1 = q[2 * (qi + 1) + 1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < MAT_REFS_PER_LANE; ++k) d[k] = ham256_chain(x0, x1, ra[k], rb[k]);
        mat_store_row<FULL>(out + (size_t)qi * nr + r0, d, r0, nr);
        touch_sgpr(y0, y1);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int qn = min(qi + 2, qb - 1);
            x0 = q[2 * qn]; x1 = q[2 * qn + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < MAT_REFS_PER_LANE; ++k) d[k] = ham256_chain(y0, y1, ra[k], rb[k]);
        mat_store_row<FULL>(out + (size_t)(qi + 1) * nr + r0, d, r0, nr);
    }
    if ((qb - qa) & 1) {
#pragma unroll
        for (int k = 0; k < MAT_REFS_PER_LANE; ++k) d[k] = ham256_chain(x0, x1, ra[k], rb[k]);
        mat_store_row<FULL>(out + (size_t)qi * nr + r0, d, r0, nr);
    }
}

constexpr int MAT_WAVES = 8;  // 512 threads: a block writes 8 KB contiguous per query row

// FULL (nr >= 512, nr % 8 == 0, 16-byte aligned rows) is decided on the host: two kernels, so the guarded path's
// registers do not cost the streaming path its 6th wave per SIMD.
template <bool FULL>
__global__ __launch_bounds__(64 * MAT_WAVES, FULL ? 6 : 4) void k_hamming_matrix(const uint4* __restrict__ q, int nq,
                                                        const uint4* __restrict__ r, int nr,
                                                        uint16_t* __restrict__ out, int q_per_block) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * MAT_WAVES + wave;
    if ((size_t)tile * MAT_REFS_PER_WAVE >= (size_t)nr) return;
    // The last, partial tile is shifted back to end exactly at nr (it then recomputes a few columns of its neighbour
    // and stores identical values): every wave keeps the branch-free full-tile path when nr >= 512 and nr % 8 == 0.
    constexpr bool full = FULL;
    const int tile_start = full ? min(tile * MAT_REFS_PER_WAVE, nr - MAT_REFS_PER_WAVE) : tile * MAT_REFS_PER_WAVE;
    const int r0 = tile_start + lane * MAT_REFS_PER_LANE;
    uint4 ra[MAT_REFS_PER_LANE], rb[MAT_REFS_PER_LANE];
#pragma unroll
    for (int k = 0; k < MAT_REFS_PER_LANE; ++k) {
        const int j = min(r0 + k, nr - 1);
        ra[k] = r[2 * j]; rb[k] = r[2 * j + 1];
    }
    const int qa = blockIdx.y * q_per_block, qb = min(nq, qa + q_per_block);
    mat_rows<FULL>(q, qa, qb, ra, rb, out, nr, r0);
}

struct FrameDev {
    int n_total, n_cams;
    const float* un_x; const float* un_y; const float* uright;
    const int* octave;
    const uint4* desc;  // global-index order, 2 x uint4 per feature
    const int* cell_start; const int* items;
    float minX, minY, invW, invH;
};

// One wave per query.  Visits cells ix (outer) / iy (inner) and items in ascending order -- App. A-8 -- so the
// compacted output order is exactly the reference's candidate order (it decides distance ties).
__global__ __launch_bounds__(256) void k_project(FrameDev F, const orbm_query* __restrict__ q, int nq, int cap,
                                                 int gate_right, int with_dist, int* __restrict__ cand_idx,
                                                 uint16_t* __restrict__ cand_dist, int* __restrict__ cand_count) {
    const int lane = threadIdx.x & 63;
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (qi >= nq) return;
    const orbm_query* Q = q + qi;
    const float x = Q->u, y = Q->v, r = Q->radius, ur = Q->ur;
    const int minLevel = Q->min_level, maxLevel = Q->max_level, cam = Q->cam;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(Q->desc);
    const uint4 q0 = make_uint4(qd[0], qd[1], qd[2], qd[3]), q1 = make_uint4(qd[4], qd[5], qd[6], qd[7]);

    int total = 0;
    const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.invW));
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf((x - F.minX + r) * F.invW));
    const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.invH));
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf((y - F.minY + r) * F.invH));
    const bool ok = nMinCellX < ORBM_GRID_COLS && nMaxCellX >= 0 && nMinCellY < ORBM_GRID_ROWS && nMaxCellY >= 0 &&
                    cam >= 0 && cam < F.n_cams;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    if (ok) {
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            for (int iy = nMinCellY; iy <= nMaxCellY; ++iy) {
                const int cell = (cam * ORBM_GRID_COLS + ix) * ORBM_GRID_ROWS + iy;
                const int s = F.cell_start[cell], e = F.cell_start[cell + 1];
                for (int base = s; base < e; base += 64) {
                    const int k = base + lane;
                    const bool valid = k < e;
                    const int g = valid ? F.items[k] : 0;
                    bool pass = valid;
                    if (pass && bCheckLevels) {
                        const int oct = F.octave[g];
                        if (oct < minLevel) pass = false;
                        if (maxLevel >= 0 && oct > maxLevel) pass = false;
                    }
                    if (pass) {
                        const float distx = F.un_x[g] - x, disty = F.un_y[g] - y;
                        pass = fabsf(distx) < r && fabsf(disty) < r;
                    }
                    if (pass && gate_right) {
                        const float urg = F.uright[g];
                        if (urg > 0 && fabsf(ur - urg) > r) pass = false;
                    }
                    const unsigned long long mask = __ballot(pass);
                    const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pass && pos < cap) {
                        const size_t o = (size_t)qi * cap + pos;
                        cand_idx[o] = g;
                        if (with_dist) cand_dist[o] = (uint16_t)ham256(q0, q1, F.desc[2 * g], F.desc[2 * g + 1]);
                    }
                    total += __popcll(mask);
                }
            }
        }
    }
    if (lane == 0) cand_count[qi] = total;
}

// ------------------------------------------------------------------------------------------------ launchers
int launch_top2(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_bi, int32_t* d_bd, int32_t* d_sd,
                void* d_scratch, int S, hipStream_t st) {
    const int qblocks = (nq + 63) / 64;
    if (S <= 1) {
        hipLaunchKernelGGL(k_hamming_top2, dim3(qblocks, 1), dim3(64 * TOP2_WAVES), 0, st, (const uint4*)d_q, nq,
                           (const uint4*)d_r, nr, d_bi, d_bd, d_sd);
    } else {
        int* p = (int*)d_scratch;
        int *p_idx = p, *p_best = p + (size_t)S * nq, *p_second = p + 2 * (size_t)S * nq;
        hipLaunchKernelGGL(k_hamming_top2, dim3(qblocks, S), dim3(64 * TOP2_WAVES), 0, st, (const uint4*)d_q, nq,
                           (const uint4*)d_r, nr, p_idx, p_best, p_second);
        hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, p_idx, p_best, p_second, S, nq, d_bi,
                           d_bd, d_sd);
    }
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

// number of reference slices: enough blocks to give every SIMD of the 256 CUs a wave
int top2_slices(int nq, int nr) {
    const int qblocks = (nq + 63) / 64;
    int S = (128 + qblocks - 1) / qblocks;               // target >= 128 blocks x 16 waves = 2048 waves
    S = std::min(S, std::max(1, nr / (TOP2_WAVES * 16)));  // keep >= 16 references per wave
    return std::max(1, std::min(S, 64));
}

int launch_matrix(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, uint16_t* d_out, hipStream_t st) {
    const int tiles = (nr + MAT_REFS_PER_WAVE - 1) / MAT_REFS_PER_WAVE;
    // queries per block: 256 at all-pairs sizes (long streaming rows); fewer when the grid would otherwise be too
    // small to fill 256 CUs x 4 SIMDs (each wave walks its queries serially, ~0.35 us per query)
    static const int q_per_block_env = [] { const char* e = getenv("MORB_MATRIX_QPB"); return e ? atoi(e) : 0; }();
    int q_per_block = (int)std::min<long long>(256, std::max<long long>(8, ((long long)nq * tiles + 4095) / 4096));
    q_per_block = (q_per_block + 1) & ~1;
    if (q_per_block_env > 0) q_per_block = q_per_block_env;
    dim3 grid((tiles + MAT_WAVES - 1) / MAT_WAVES, (nq + q_per_block - 1) / q_per_block);
    const bool aligned = (nr % 8 == 0) && (((uintptr_t)d_out & 15) == 0) && nr >= MAT_REFS_PER_WAVE;
    if (aligned)
        hipLaunchKernelGGL(k_hamming_matrix<true>, grid, dim3(64 * MAT_WAVES), 0, st, (const uint4*)d_q, nq, (const uint4*)d_r, nr,
                           d_out, q_per_block);
    else
        hipLaunchKernelGGL(k_hamming_matrix<false>, grid, dim3(64 * MAT_WAVES), 0, st, (const uint4*)d_q, nq, (const uint4*)d_r,
                           nr, d_out, q_per_block);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

}  // namespace

// ================================================================================================ C ABI
struct orbm_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_q, d_r, d_scratch, d_queries;
    DevBuf<int32_t> d_i0, d_i1, d_i2;
    DevBuf<uint16_t> d_u16;
    PinnedBuf<int32_t> h_i0, h_i1;
    PinnedBuf<uint16_t> h_u16;
};

struct orbm_frame {
    orbm_matcher* owner = nullptr;
    int n_total = 0, n_cams = 0;
    float minX = 0, minY = 0, maxX = 0, maxY = 0, invW = 0, invH = 0;
    // host copies used by the sequential resolve
    std::vector<int32_t> octave;
    std::vector<float> angle;
    std::vector<int32_t> cell_start, items;
    // device
    DevBuf<float> d_x, d_y, d_ur;
    DevBuf<int32_t> d_oct, d_cell_start, d_items;
    DevBuf<uint8_t> d_desc;
    FrameDev dev() const {
        FrameDev F;
        F.n_total = n_total; F.n_cams = n_cams; F.un_x = d_x.p; F.un_y = d_y.p; F.uright = d_ur.p; F.octave = d_oct.p;
        F.desc = (const uint4*)d_desc.p; F.cell_start = d_cell_start.p; F.items = d_items.p;
        F.minX = minX; F.minY = minY; F.invW = invW; F.invH = invH;
        return F;
    }
};

extern "C" {

int orbm_create(int device, orbm_matcher** out) {
    MORB_ARG(out != nullptr);
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    orbm_matcher* m = new orbm_matcher();
    m->device = device;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete m; return ORB_E_HIP; }
    *out = m;
    return ORB_OK;
}

void orbm_destroy(orbm_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    m->d_q.release(); m->d_r.release(); m->d_scratch.release(); m->d_queries.release();
    m->d_i0.release(); m->d_i1.release(); m->d_i2.release(); m->d_u16.release();
    m->h_i0.release(); m->h_i1.release(); m->h_u16.release();
    (void)hipStreamDestroy(m->stream);
    delete m;
}

void* orbm_stream(const orbm_matcher* m) { return m ? (void*)m->stream : nullptr; }

int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8); memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

void orbm_three_maxima(const int* histo, int L, int* ind) {
    // Keeps the three fullest bins; an earlier bin wins a tie (strict '>'), 2nd/3rd dropped below 10% of the 1st.
    int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > m1) { m3 = m2; i3 = i2; m2 = m1; i2 = i1; m1 = s; i1 = i; }
        else if (s > m2) { m3 = m2; i3 = i2; m2 = s; i2 = i; }
        else if (s > m3) { m3 = s; i3 = i; }
    }
    if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
    else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
    ind[0] = i1; ind[1] = i2; ind[2] = i3;
}

size_t orbm_top2_scratch_bytes(int nq, int nr) {
    if (nq <= 0 || nr <= 0) return 0;
    const int S = top2_slices(nq, nr);
    return S <= 1 ? 0 : (size_t)3 * S * nq * sizeof(int);
}

int orbm_hamming_top2_device(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_best_idx,
                             int32_t* d_best_dist, int32_t* d_second_dist, void* d_scratch, void* stream) {
    MORB_ARG(nq >= 0 && nr >= 0);
    if (nq == 0) return ORB_OK;
    MORB_ARG(d_q && d_best_idx && d_best_dist && d_second_dist && (nr == 0 || d_r));
    MORB_ARG((((uintptr_t)d_q | (uintptr_t)d_r) & 15) == 0);
    int S = top2_slices(nq, std::max(nr, 1));
    if (S > 1 && !d_scratch) S = 1;
    return launch_top2(d_q, nq, d_r, nr, d_best_idx, d_best_dist, d_second_dist, d_scratch, S, (hipStream_t)stream);
}

int orbm_hamming_top2(orbm_matcher* m, const uint8_t* q, int nq, const uint8_t* r, int nr, int32_t* best_idx,
                      int32_t* best_dist, int32_t* second_dist) {
    MORB_ARG(m && nq >= 0 && nr >= 0);
    if (nq == 0) return ORB_OK;
    MORB_ARG(q && best_idx && best_dist && second_dist && (nr == 0 || r));
    MORB_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = m->d_q.reserve((size_t)nq * 32)) || (rc = m->d_r.reserve((size_t)std::max(nr, 1) * 32)) ||
        (rc = m->d_i0.reserve(nq)) || (rc = m->d_i1.reserve(nq)) || (rc = m->d_i2.reserve(nq)) ||
        (rc = m->d_scratch.reserve(std::max<size_t>(orbm_top2_scratch_bytes(nq, nr), 16))))
        return rc;
    MORB_HIP(hipMemcpyAsync(m->d_q.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    if (nr) MORB_HIP(hipMemcpyAsync(m->d_r.p, r, (size_t)nr * 32, hipMemcpyHostToDevice, m->stream));
    rc = orbm_hamming_top2_device(m->d_q.p, nq, m->d_r.p, nr, m->d_i0.p, m->d_i1.p, m->d_i2.p, m->d_scratch.p, m->stream);
    if (rc) return rc;
    MORB_HIP(hipMemcpyAsync(best_idx, m->d_i0.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipMemcpyAsync(best_dist, m->d_i1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipMemcpyAsync(second_dist, m->d_i2.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

int orbm_hamming_matrix_device(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, uint16_t* d_out, void* stream) {
    MORB_ARG(nq >= 0 && nr >= 0);
    if (nq == 0 || nr == 0) return ORB_OK;
    MORB_ARG(d_q && d_r && d_out);
    MORB_ARG((((uintptr_t)d_q | (uintptr_t)d_r) & 15) == 0);
    return launch_matrix(d_q, nq, d_r, nr, d_out, (hipStream_t)stream);
}

int orbm_hamming_matrix(orbm_matcher* m, const uint8_t* q, int nq, const uint8_t* r, int nr, uint16_t* out) {
    MORB_ARG(m && nq >= 0 && nr >= 0);
    if (nq == 0 || nr == 0) return ORB_OK;
    MORB_ARG(q && r && out);
    MORB_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = m->d_q.reserve((size_t)nq * 32)) || (rc = m->d_r.reserve((size_t)nr * 32)) ||
        (rc = m->d_u16.reserve((size_t)nq * nr)))
        return rc;
    MORB_HIP(hipMemcpyAsync(m->d_q.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    MORB_HIP(hipMemcpyAsync(m->d_r.p, r, (size_t)nr * 32, hipMemcpyHostToDevice, m->stream));
    rc = launch_matrix(m->d_q.p, nq, m->d_r.p, nr, m->d_u16.p, m->stream);
    if (rc) return rc;
    MORB_HIP(hipMemcpyAsync(out, m->d_u16.p, (size_t)nq * nr * 2, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
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
    for (int g = 0; g < n; g++) {
        const int px = (int)roundf((f->un_x[g] - F->minX) * F->invW);
        const int py = (int)roundf((f->un_y[g] - F->minY) * F->invH);
        const int cam = f->cam_of[g];
        if (px < 0 || px >= ORBM_GRID_COLS || py < 0 || py >= ORBM_GRID_ROWS || cam < 0 || cam >= f->n_cams) {
            cell_of[g] = -1;
            continue;
        }
        cell_of[g] = (cam * ORBM_GRID_COLS + px) * ORBM_GRID_ROWS + py;
        F->cell_start[cell_of[g] + 1]++;
    }
    for (int c = 0; c < ncell; c++) F->cell_start[c + 1] += F->cell_start[c];
    F->items.assign(std::max(F->cell_start[ncell], 1), 0);
    {
        std::vector<int32_t> cursor(F->cell_start.begin(), F->cell_start.end() - 1);
        for (int g = 0; g < n; g++)
            if (cell_of[g] >= 0) F->items[cursor[cell_of[g]]++] = g;
    }
    // descriptors re-laid in global-index order so the kernel gathers with one index
    std::vector<uint8_t> desc((size_t)std::max(n, 1) * 32);
    for (int g = 0; g < n; g++) memcpy(&desc[(size_t)g * 32], f->desc[f->cam_of[g]] + (size_t)f->local_of[g] * 32, 32);

    int rc;
    const size_t nn = (size_t)std::max(n, 1);
    if ((rc = F->d_x.reserve(nn)) || (rc = F->d_y.reserve(nn)) || (rc = F->d_ur.reserve(nn)) || (rc = F->d_oct.reserve(nn)) ||
        (rc = F->d_desc.reserve(nn * 32)) || (rc = F->d_cell_start.reserve(ncell + 1)) ||
        (rc = F->d_items.reserve(F->items.size()))) {
        orbm_frame_destroy(F);
        return rc;
    }
    hipStream_t st = m->stream;
    if (n) {
        MORB_HIP(hipMemcpyAsync(F->d_x.p, f->un_x, nn * 4, hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->d_y.p, f->un_y, nn * 4, hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->d_ur.p, f->uright, nn * 4, hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->d_oct.p, f->octave, nn * 4, hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->d_desc.p, desc.data(), nn * 32, hipMemcpyHostToDevice, st));
    }
    MORB_HIP(hipMemcpyAsync(F->d_cell_start.p, F->cell_start.data(), (size_t)(ncell + 1) * 4, hipMemcpyHostToDevice, st));
    MORB_HIP(hipMemcpyAsync(F->d_items.p, F->items.data(), F->items.size() * 4, hipMemcpyHostToDevice, st));
    MORB_HIP(hipStreamSynchronize(st));  // `desc` and the caller's arrays may go away after return
    *out = F;
    return ORB_OK;
}

void orbm_frame_destroy(orbm_frame* f) {
    if (!f) return;
    if (f->owner) (void)hipSetDevice(f->owner->device);
    f->d_x.release(); f->d_y.release(); f->d_ur.release(); f->d_oct.release(); f->d_desc.release();
    f->d_cell_start.release(); f->d_items.release();
    delete f;
}

int orbm_frame_grid(const orbm_frame* f, int32_t* cell_start, int32_t* items) {
    MORB_ARG(f && cell_start);
    memcpy(cell_start, f->cell_start.data(), f->cell_start.size() * 4);
    if (items && f->cell_start.back() > 0) memcpy(items, f->items.data(), (size_t)f->cell_start.back() * 4);
    return ORB_OK;
}

static int run_project(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int cap, int gate_right,
                       int with_dist) {
    int rc;
    if ((rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query))) || (rc = m->d_i0.reserve((size_t)nq * cap)) ||
        (rc = m->d_u16.reserve((size_t)nq * cap)) || (rc = m->d_i1.reserve(nq)) || (rc = m->h_i0.reserve((size_t)nq * cap)) ||
        (rc = m->h_u16.reserve((size_t)nq * cap)) || (rc = m->h_i1.reserve(nq)))
        return rc;
    MORB_HIP(hipMemcpyAsync(m->d_queries.p, q, (size_t)nq * sizeof(orbm_query), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_project, dim3((nq + 3) / 4), dim3(256), 0, m->stream, f->dev(),
                       (const orbm_query*)m->d_queries.p, nq, cap, gate_right, with_dist, m->d_i0.p, m->d_u16.p, m->d_i1.p);
    MORB_HIP(hipGetLastError());
    MORB_HIP(hipMemcpyAsync(m->h_i1.p, m->d_i1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipMemcpyAsync(m->h_i0.p, m->d_i0.p, (size_t)nq * cap * 4, hipMemcpyDeviceToHost, m->stream));
    if (with_dist) MORB_HIP(hipMemcpyAsync(m->h_u16.p, m->d_u16.p, (size_t)nq * cap * 2, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

// Runs k_project with a growing per-query capacity until every list fits; results in m->h_i0 / h_u16 / h_i1.
static int project_all(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int gate_right, int with_dist,
                       int* cap_out) {
    int cap = 64;
    for (;;) {
        int rc = run_project(m, f, q, nq, cap, gate_right, with_dist);
        if (rc) return rc;
        int mx = 0;
        for (int i = 0; i < nq; i++) mx = std::max(mx, m->h_i1.p[i]);
        if (mx <= cap) break;
        cap = (mx + 63) & ~63;
    }
    *cap_out = cap;
    return ORB_OK;
}

int orbm_features_in_area(orbm_matcher* m, const orbm_frame* f, int cam, float x, float y, float r, int min_level,
                          int max_level, int32_t* out, int cap, int* n) {
    MORB_ARG(m && f && n && (cap == 0 || out));
    MORB_HIP(hipSetDevice(m->device));
    orbm_query Q;
    memset(&Q, 0, sizeof(Q));
    Q.u = x; Q.v = y; Q.radius = r; Q.min_level = min_level; Q.max_level = max_level; Q.cam = cam;
    int pc = 0;
    int rc = project_all(m, f, &Q, 1, /*gate_right=*/0, /*with_dist=*/0, &pc);
    if (rc) return rc;
    *n = m->h_i1.p[0];
    for (int i = 0; i < *n && i < cap; i++) out[i] = m->h_i0.p[i];
    return ORB_OK;
}

int orbm_project_candidates(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int cap_per_query,
                            int32_t* cand_idx, uint16_t* cand_dist, int32_t* cand_count) {
    MORB_ARG(m && f && nq >= 0 && cap_per_query > 0);
    if (nq == 0) return ORB_OK;
    MORB_ARG(q && cand_idx && cand_dist && cand_count);
    MORB_HIP(hipSetDevice(m->device));
    int rc = run_project(m, f, q, nq, cap_per_query, 1, 1);
    if (rc) return rc;
    bool overflow = false;
    for (int i = 0; i < nq; i++) {
        cand_count[i] = m->h_i1.p[i];
        if (cand_count[i] > cap_per_query) overflow = true;
    }
    memcpy(cand_idx, m->h_i0.p, (size_t)nq * cap_per_query * 4);
    memcpy(cand_dist, m->h_u16.p, (size_t)nq * cap_per_query * 2);
    if (overflow) { morb::set_error("candidate list longer than cap_per_query=%d", cap_per_query); return ORB_E_CAPACITY; }
    return ORB_OK;
}

int orbm_search_by_projection(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                              const uint8_t* occupied, int th_high, int check_orientation, int32_t* match_of_feature,
                              int* nmatches) {
    MORB_ARG(m && cur && nq >= 0 && nmatches && (cur->n_total == 0 || match_of_feature));
    MORB_HIP(hipSetDevice(m->device));
    for (int g = 0; g < cur->n_total; g++) match_of_feature[g] = -1;
    *nmatches = 0;
    if (nq == 0) return ORB_OK;
    MORB_ARG(q != nullptr);
    int cap = 0;
    int rc = project_all(m, cur, q, nq, 1, 1, &cap);
    if (rc) return rc;
    // Sequential first-come resolve in query order (reference src/ORBmatcher.cc:3502-3614): a feature claimed by a
    // query whose MapPoint is observed is invisible to later queries; among the rest the first minimum wins.
    std::vector<int32_t> rot[ORBM_HISTO_LENGTH];
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        const int cnt = m->h_i1.p[i];
        const int32_t* ci = m->h_i0.p + (size_t)i * cap;
        const uint16_t* cd = m->h_u16.p + (size_t)i * cap;
        int best = 256, bidx = -1;
        for (int k = 0; k < cnt; k++) {
            const int g = ci[k];
            const int owner = match_of_feature[g];
            if (owner >= 0 ? q[owner].blocks != 0 : (occupied && occupied[g])) continue;
            if ((int)cd[k] < best) { best = cd[k]; bidx = g; }
        }
        if (best <= th_high && bidx >= 0) {
            match_of_feature[bidx] = i;
            nm++;
            if (check_orientation) {
                float rotv = q[i].angle - cur->angle[bidx];
                if (rotv < 0.0) rotv += 360.0f;
                int bin = (int)roundf(rotv * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rot[bin].push_back(bidx);
            }
        }
    }
    if (check_orientation) {
        int sizes[ORBM_HISTO_LENGTH], ind[3];
        for (int b = 0; b < ORBM_HISTO_LENGTH; b++) sizes[b] = (int)rot[b].size();
        orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; b++)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int g : rot[b]) { match_of_feature[g] = -2; nm--; }
    }
    *nmatches = nm;
    return ORB_OK;
}

int orbm_search_by_projection_points(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                                     const uint8_t* occupied, float nnratio, int th_high, int32_t* match_of_feature,
                                     int* nmatches) {
    MORB_ARG(m && cur && nq >= 0 && nmatches && (cur->n_total == 0 || match_of_feature));
    MORB_HIP(hipSetDevice(m->device));
    for (int g = 0; g < cur->n_total; g++) match_of_feature[g] = -1;
    *nmatches = 0;
    if (nq == 0) return ORB_OK;
    MORB_ARG(q != nullptr);
    std::vector<orbm_query> q0(q, q + nq);
    for (auto& Q : q0) Q.cam = 0;  // camera-1 grid only (reference src/ORBmatcher.cc:88-89, src/Frame.cc:510-563)
    int cap = 0;
    int rc = project_all(m, cur, q0.data(), nq, 1, 1, &cap);
    if (rc) return rc;
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        const int cnt = m->h_i1.p[i];
        const int32_t* ci = m->h_i0.p + (size_t)i * cap;
        const uint16_t* cd = m->h_u16.p + (size_t)i * cap;
        int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1;
        for (int k = 0; k < cnt; k++) {
            const int g = ci[k];
            if (occupied && occupied[g]) continue;
            const int owner = match_of_feature[g];
            if (owner >= 0 && q[owner].blocks) continue;
            const int d = cd[k];
            if (d < best) { best2 = best; best = d; lvl2 = lvl; lvl = cur->octave[g]; bidx = g; }
            else if (d < best2) { lvl2 = cur->octave[g]; best2 = d; }
        }
        if (best <= th_high && bidx >= 0) {
            if (lvl == lvl2 && (float)best > nnratio * (float)best2) continue;
            match_of_feature[bidx] = i;
            nm++;
        }
    }
    *nmatches = nm;
    return ORB_OK;
}

}  // extern "C"
