// matcher.hip -- gfx950 kernels + C ABI of the ORB matcher (include/orbm.h) and of the one-call front end (include/orbf.h).
//
// Kernels (integer / bit work on the vector ALU; the matrix cores are used in exactly two kernels, the all-pairs forms below):
//   k_hamming_top2 / k_cross_top2   M1  exhaustive top-2 Hamming, xor + popcount form: one query per lane, references walked
//                            with wave-uniform (scalar-cache) loads, 16 waves per block each scanning 1/16 of the references,
//                            LDS merge.  VALU-bound.  reference src/ORBmatcher.cc:287-321.
//   k_hamming_top2_mfma      M1  the same results from v_mfma_i32_32x32x32_i8 on the +-1-expanded descriptors
//                            (dot = 256 - 2 * distance, exact); accumulators come out as ready-made sort keys.
//   k_hamming_matrix[_mfma]  M2  full uint16 distance matrix, popcount / matrix-core form.  HBM-write-bound (mfma form).
//   k_project                M3  projection-gated search: one wave per query walks the 64x48 grid cells of the window in
//                            the reference's visiting order, ballot-compacts the survivors in order, gathers their
//                            descriptors and keeps a sorted shortlist.  reference src/ORBmatcher.cc:3547-3592 + src/Frame.cc:574-629.
//   k_resolve / k_rs_*       the order-dependent part of SearchByProjection (first-come claims, rotation histogram,
//                            ComputeThreeMaxima) as a fixed-point iteration ON THE DEVICE; the host replay of the loop
//                            (host_resolve) is the exact fallback (sweep limit, MORB_HOST_RESOLVE=1).
//   k_frame_*                Frame merge, ComputeStereoFromRGBD, AssignFeaturesToGrid on the device (src/Frame.cc:191-395).
//   k_repack_gathered        multi-GPU: the all-gathered export blocks -> one contiguous descriptor list.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <vector>

#include "../../include/orbm.h"
#include <chrono>
#include "orb_common.h"
#include "frame_sink.h"

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

// Running top-2 of one query per lane against the rows [ja, jb) of `r` (wave-uniform addresses -> scalar loads, four rows =
// 128 B per trip); the reported index is j - shift.  second = 2nd smallest with multiplicity, best index = first minimum
// (strict '<' chain, ORBmatcher.cc:311-320).
__device__ __forceinline__ void top2_scan(const uint4* __restrict__ r, int ja, int jb, int shift, const uint4& q0, const uint4& q1,
                                          int& b, int& s, int& bi) {
#define TOP2_UPDATE(d, j) do { s = min(s, max(b, (d))); bi = (d) < b ? (j) : bi; b = min(b, (d)); } while (0)
    int j = ja;
    for (; j + 4 <= jb; j += 4) {
        const uint4 a0 = r[2 * j], a1 = r[2 * j + 1], b0 = r[2 * j + 2], b1 = r[2 * j + 3];
        const uint4 c0 = r[2 * j + 4], c1 = r[2 * j + 5], e0 = r[2 * j + 6], e1 = r[2 * j + 7];
        const int d0 = (int)ham256_chain(a0, a1, q0, q1), d1 = (int)ham256_chain(b0, b1, q0, q1);
        const int d2 = (int)ham256_chain(c0, c1, q0, q1), d3 = (int)ham256_chain(e0, e1, q0, q1);
        TOP2_UPDATE(d0, j - shift); TOP2_UPDATE(d1, j + 1 - shift); TOP2_UPDATE(d2, j + 2 - shift); TOP2_UPDATE(d3, j + 3 - shift);
    }
    for (; j < jb; ++j) {
        const uint4 a0 = r[2 * j], a1 = r[2 * j + 1];
        const int d = (int)ham256_chain(a0, a1, q0, q1);
        TOP2_UPDATE(d, j - shift);
    }
#undef TOP2_UPDATE
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
    top2_scan(r, j0, j1, 0, q0, q1, b, s, bi);
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
                             int* __restrict__ best_dist, int* __restrict__ second_dist, const int* __restrict__ d_range = nullptr) {
    if (d_range) nq = d_range[2];  // partial arrays are laid out with stride nq: the producer used the same device count
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

// ---- the same matrix on the matrix cores --------------------------------------------------------------------------
// With every descriptor bit b mapped to the int8 value 1 - 2b, the dot product of two descriptors is
// (#equal bits) - (#different bits) = 256 - 2 * Hamming: exact in the int32 accumulators of v_mfma_i32_32x32x32_i8, and
// 8 MFMAs (K = 8 x 32) give a 32 x 32 block of distances for ~0.25 SIMD cycles per pair where the xor/popcount chain
// above needs ~1.03.  What is left is the 2 bytes per pair that have to reach HBM.
//
// Workgroup = 4 waves = 256 queries; a wave keeps its 64 queries as B fragments in 64 VGPRs for the whole launch and
// walks the references 64 at a time: the workgroup expands the 64 x 32 bytes of a tile into int8 in fragment order in
// LDS (ds_read_b128 at lane * 16: no bank conflicts; tile t + 1 is fetched and expanded while tile t is multiplied),
// each wave issues 32 MFMAs per tile and transposes its 64 x 64 result through a 4 KB LDS patch (XOR-swizzled 16-byte
// chunks) so that every store instruction writes 8 rows x 128 contiguous bytes.  The order of K inside a fragment is
// irrelevant to a dot product as long as both operands use the same one: fragment (ks, h) = descriptor bits
// [32 ks + 16 h, +16) of row (lane & 31), for A (references) and B (queries) alike.
using mm_i32x4 = __attribute__((ext_vector_type(4))) int;
using mm_i32x16 = __attribute__((ext_vector_type(16))) int;
constexpr int MM_WAVES = 4;
constexpr int MM_Q_PER_BLOCK = 64 * MM_WAVES;
constexpr int MM_R_TILE = 64;

// 16 descriptor bits -> 16 int8: +1 where the bit is clear, -1 where it is set.  (x * 0x00204081) & 0x01010101 spreads
// the four bits of a nibble over four bytes (the shifted copies x, x<<7, x<<14, x<<21 do not overlap for x < 16); the
// bytes 0 / 1 then select 0x01 / 0xff out of a constant with v_perm_b32.
__device__ __forceinline__ mm_i32x4 mm_expand16(uint32_t bits) {
    mm_i32x4 v;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint32_t x = (bits >> (4 * n)) & 15u;
        const uint32_t y = (x * 0x00204081u) & 0x01010101u;
        v[n] = (int)__builtin_amdgcn_perm(0u, 0x0000ff01u, y);
    }
    return v;
}

typedef unsigned short mm_u16x2 __attribute__((ext_vector_type(2)));
// two accumulators (256 - 2 * distance each) -> distance | distance << 16, on the packed 16-bit ALU
__device__ __forceinline__ uint32_t mm_pack2(int d0, int d1) {
    mm_u16x2 p;
    p.x = (unsigned short)d0; p.y = (unsigned short)d1;
    const mm_u16x2 k = {256, 256};
    p = (k - p) >> 1;
    return __builtin_bit_cast(uint32_t, p);
}

__global__ __launch_bounds__(64 * MM_WAVES) void k_hamming_matrix_mfma(const uint32_t* __restrict__ q, int nq,
                                                                      const uint32_t* __restrict__ r, int nr,
                                                                      uint16_t* __restrict__ out, int tiles_per_block) {
    __shared__ mm_i32x4 s_tile[2][2 * 8 * 64];   // [buffer][(reference group, ks, lane)]: 2 x 16 KB
    __shared__ uint4 s_stage[MM_WAVES][32 * 8];  // per wave: 32 query rows x 8 chunks of 8 distances
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * MM_Q_PER_BLOCK + wave * 64;

    // Rows past nq repeat query nq - 1 and are stored onto its row (identical values); the last reference tile is moved
    // back to end at nr (it recomputes columns of its neighbour): no store below is conditional, so the loop body has
    // no exec-mask branches and the wait for a prefetched tile does not have to drain the stores issued after it.
    mm_i32x4 bq[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bq[g][ks] = mm_expand16(h ? (w[ks] >> 16) : (w[ks] & 0xffffu));
    }

    const int n_tiles = (nr + MM_R_TILE - 1) / MM_R_TILE;
    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(n_tiles, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;
    // wave w expands descriptor words 2w, 2w + 1 (ks = 2w, 2w + 1; both halves) of the tile's 64 references, one per lane
    // The prefetch is issued and awaited by hand: hipcc's own accounting drains every outstanding store (vmcnt(0)) when
    // it waits for a load across the loop's back edge, and a wave would then stop once per tile until its previous
    // 8 KB of distances has reached the L2.  gfx950 retires loads and stores in issue order on one counter, so with
    // exactly eight stores issued after the request, vmcnt(8) means "the request has landed".
    auto fetch = [&](int t) {
        const int rr = min(min(t, t_end - 1) * MM_R_TILE, nr - MM_R_TILE) + lane;
        const uint32_t* p = r + (size_t)rr * 8 + wave * 2;
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
        return v;
    };
    auto deposit = [&](int buf, unsigned long long w64) {
        const uint2 w = make_uint2((uint32_t)w64, (uint32_t)(w64 >> 32));
        mm_i32x4* base = &s_tile[buf][(h * 8 + wave * 2) * 64 + c];
        base[0] = mm_expand16(w.x & 0xffffu);
        base[32] = mm_expand16(w.x >> 16);
        base[64] = mm_expand16(w.y & 0xffffu);
        base[96] = mm_expand16(w.y >> 16);
    };
    // this lane's part of the four store instructions of a 32-query group: row (lane >> 3) + 8 i, chunk lane & 7
    const int srow = lane >> 3, sch = lane & 7;
    size_t row_off[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) row_off[g][i] = (size_t)min(q0 + g * 32 + i * 8 + srow, nq - 1) * nr + sch * 8;

    unsigned long long nxt = fetch(t_begin);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) : : "memory");
    deposit(0, nxt);
    nxt = fetch(t_begin + 1);
    __syncthreads();
    auto one_tile = [&](int t, auto first) {
        const int buf = (t - t_begin) & 1;
        mm_i32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][g][e] = 0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const mm_i32x4 a0 = s_tile[buf][ks * 64 + lane], a1 = s_tile[buf][(8 + ks) * 64 + lane];
            acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0][ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[1][ks], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[0][ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[1][ks], acc[1][1], 0, 0, 0);
        }
        // tile t + 1 into the other buffer (its last reader passed the barrier that ended iteration t - 1), tile t + 2
        // requested BEFORE this iteration's stores are issued
        if (decltype(first)::value) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) : : "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(nxt) : : "memory");
        deposit(buf ^ 1, nxt);
        nxt = fetch(t + 2);
        // D[m][n]: lane holds column n = lane & 31 (a query), rows m = (e & 3) + 8 (e >> 2) + 4 h (references)
        const int r0 = min(t * MM_R_TILE, nr - MM_R_TILE);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // 16-byte chunks XOR-swizzled by the row, and the two 8-byte halves of a chunk swapped on every second
                    // group of eight rows: the 32 lanes of a half-wave (same h) then cover all 64 banks once
                    const int chunk = (a * 4 + j) ^ (c & 7);
                    reinterpret_cast<uint2*>(&s_stage[wave][c * 8 + chunk])[h ^ ((c >> 3) & 1)] =
                        make_uint2(mm_pack2(acc[a][g][4 * j + 0], acc[a][g][4 * j + 1]), mm_pack2(acc[a][g][4 * j + 2], acc[a][g][4 * j + 3]));
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = s_stage[wave][(i * 8 + srow) * 8 + (sch ^ srow)];
                if (i & 1) v = make_uint4(v.z, v.w, v.x, v.y);  // rows 8..15, 24..31 hold their halves swapped
                *reinterpret_cast<uint4*>(out + row_off[g][i] + r0) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
    };
    // first tile outside the loop: no stores stand behind its prefetch yet
    one_tile(t_begin, std::true_type{});
    for (int t = t_begin + 1; t < t_end; ++t) one_tile(t, std::false_type{});
}

// ---- exhaustive top-2 on the matrix cores ------------------------------------------------------------------------
// Same tiling as k_hamming_matrix_mfma (a wave keeps 64 queries as B fragments, the workgroup expands 64 references per
// step into LDS), but nothing is stored per pair and the MFMA delivers ready-made sort keys: references are expanded to
// -32 / +32 (bit clear / set), queries to +1 / -1, so a dot product is 64 * distance - 8192, and the accumulators start at
// 8192 + (row of the element within its 32 x 32 block) -- D[m][n] = distance << 6 | m, smaller = better, ties by reference
// order.  Every lane keeps (best, second) of its query column and runs  second = med3(best, second, key); best =
// min(best, key)  over the 16 keys a block gives it: three vector instructions per pair including the accumulator read,
// no branches.  After each block the row bits of `best` are cleared (and the block + row remembered when `best` changed):
// an equal distance in a later block then never replaces it -- the strict '<' chain of ORBmatcher.cc:311-320 (first
// minimum wins, second = 2nd smallest with multiplicity).  grid.x = reference slices (partials for k_top2_merge when
// > 1), grid.y = 256 queries.
__device__ __forceinline__ mm_i32x4 mt_expand16(uint32_t bits) {  // 16 bits -> 16 int8: -32 where clear, +32 where set
    mm_i32x4 v;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint32_t x = (bits >> (4 * n)) & 15u;
        const uint32_t y = (x * 0x00204081u) & 0x01010101u;
        v[n] = (int)__builtin_amdgcn_perm(0u, 0x000020e0u, y);
    }
    return v;
}

__device__ __forceinline__ uint32_t mt_umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

__global__ __launch_bounds__(64 * MM_WAVES) void k_hamming_top2_mfma(const uint32_t* __restrict__ q, int nq,
                                                                    const uint32_t* __restrict__ r, int nr, int slice_len,
                                                                    int* __restrict__ p_idx, int* __restrict__ p_best,
                                                                    int* __restrict__ p_second) {
    __shared__ mm_i32x4 s_tile[2][2 * 8 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * MM_Q_PER_BLOCK + wave * 64;

    mm_i32x4 bq[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bq[g][ks] = mm_expand16(h ? (w[ks] >> 16) : (w[ks] & 0xffffu));
    }
    mm_i32x16 cinit;  // D[m][n]: lane holds column n = lane & 31 (a query), rows m = (e & 3) + 8 (e >> 2) + 4 h (references)
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[e] = 8192 + (e & 3) + 8 * (e >> 2) + 4 * h;

    const int s0 = blockIdx.x * slice_len, s1 = min(nr, s0 + slice_len);  // slice_len is a multiple of 64
    const int n_tiles = (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE;
    auto fetch = [&](int t) {
        const int rr = min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1);  // rows past the end repeat the last one, masked below
        return *reinterpret_cast<const uint2*>(r + (size_t)rr * 8 + wave * 2);
    };
    auto deposit = [&](int buf, uint2 w) {
        mm_i32x4* base = &s_tile[buf][(h * 8 + wave * 2) * 64 + c];
        base[0] = mt_expand16(w.x & 0xffffu);
        base[32] = mt_expand16(w.x >> 16);
        base[64] = mt_expand16(w.y & 0xffffu);
        base[96] = mt_expand16(w.y >> 16);
    };

    constexpr uint32_t KEY_NONE = 256u << 6;
    uint32_t kb[2] = {KEY_NONE, KEY_NONE}, ks2[2] = {KEY_NONE, KEY_NONE};
    int where[2] = {-1, -1};  // (block << 5 | row) of the best key, block = 2 * tile + a
    deposit(0, fetch(0));
    uint2 nxt = fetch(1);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        mm_i32x16 acc[2][2];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const mm_i32x4 a0 = s_tile[buf][ks * 64 + lane], a1 = s_tile[buf][(8 + ks) * 64 + lane];
            acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0][ks], ks ? acc[0][0] : cinit, 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[1][ks], ks ? acc[0][1] : cinit, 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[0][ks], ks ? acc[1][0] : cinit, 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[1][ks], ks ? acc[1][1] : cinit, 0, 0, 0);
        }
        deposit(buf ^ 1, nxt);
        nxt = fetch(t + 2);
        const int valid = s1 - s0 - t * MM_R_TILE;  // references of this tile inside the slice (>= 64 except on the last tile)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t before = kb[g];
                if (valid >= MM_R_TILE) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t key = (uint32_t)acc[a][g][e];
                        ks2[g] = mt_umed3(kb[g], ks2[g], key);
                        kb[g] = min(kb[g], key);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int local = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const uint32_t key = local < valid ? (uint32_t)acc[a][g][e] : KEY_NONE;
                        ks2[g] = mt_umed3(kb[g], ks2[g], key);
                        kb[g] = min(kb[g], key);
                    }
                }
                where[g] = kb[g] != before ? (((2 * t + a) << 5) | (int)(kb[g] & 31u)) : where[g];
                kb[g] &= ~63u;
            }
        __syncthreads();
    }
    // the two half-waves hold disjoint references of the same query: full keys distance << 16 | index decide
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t mine_b = ((kb[g] >> 6) << 16) | (uint32_t)(where[g] & 0xffff), mine_s = (ks2[g] >> 6) << 16 | 0xffffu;
        const uint32_t ob = (uint32_t)__shfl_xor((int)mine_b, 32), os = (uint32_t)__shfl_xor((int)mine_s, 32);
        const uint32_t nb = min(mine_b, ob), ns = min(max(mine_b, ob), min(mine_s, os));
        const int qrow = q0 + g * 32 + c;
        if (h == 0 && qrow < nq) {
            const size_t o = (size_t)blockIdx.x * nq + qrow;
            const int best = (int)(nb >> 16);
            p_best[o] = best;
            p_idx[o] = best < 256 ? s0 + (int)(nb & 0xffffu) : -1;
            p_second[o] = (int)min(ns >> 16, 256u);
        }
    }
}

struct FrameDev {
    int n_total, n_cams;
    const int* n_total_dev;  // non-NULL: the feature count is only known on the device (n_total is then the capacity)
    const float* un_x; const float* un_y; const float* uright;
    const int* octave;
    const uint4* desc;  // global-index order, 2 x uint4 per feature
    const int* cell_start; const int* items;
    float minX, minY, invW, invH;
};

// One wave per query.  The window's grid cells are enumerated ix (outer) / iy (inner) -- the reference's visiting
// order, App. A-8 -- 64 cells at a time, one per lane: every lane fetches its cell's [start, end) in parallel, a wave
// prefix sum turns the counts into ordered item positions, then the items are tested 64 at a time and the survivors
// compacted with a ballot.  The output order is exactly the reference's candidate order (it decides distance ties);
// the dependent-load chain is per 64 cells instead of per cell.
// Output layout: element k of query i at [i*cap + k] (TRANSPOSED == 0) or [k*nq + i] (TRANSPOSED == 1, coalesced for
// the thread-per-query resolve kernel).
// With `topk` != NULL the wave also keeps the RESOLVE_K smallest (distance << 16 | position) keys of its non-occupied
// survivors, sorted, and writes them (+ their feature indices) at topk[k*nq + i] / topk[(K + k)*nq + i]: the shortlist
// the resolve kernel sweeps over.
constexpr int RESOLVE_K = 6;

__global__ __launch_bounds__(256) void k_project(FrameDev F, const orbm_query* __restrict__ q, int nq, int cap,
                                                 int gate_right, int with_dist, int transposed, int* __restrict__ cand_idx,
                                                 uint16_t* __restrict__ cand_dist, int* __restrict__ cand_count,
                                                 const uint8_t* __restrict__ occupied, int* __restrict__ topk, int short_th,
                                                 const float* __restrict__ inv_sigma2 = nullptr, int2* __restrict__ qmeta = nullptr,
                                                 const orbm_window* __restrict__ win2 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (qi >= nq) return;
    const orbm_query* Q = q + qi;
    float x = Q->u, y = Q->v, r = Q->radius;
    const float ur = Q->ur;
    int minLevel = Q->min_level, maxLevel = Q->max_level, cam = Q->cam;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(Q->desc);
    const uint4 q0 = make_uint4(qd[0], qd[1], qd[2], qd[3]), q1 = make_uint4(qd[4], qd[5], qd[6], qd[7]);

    int total = 0;
    int n_elig = 0;                    // survivors that could ever be accepted: not occupied and distance <= short_th
    int sk[RESOLVE_K], sg[RESOLVE_K];  // wave-uniform sorted shortlist
#pragma unroll
    for (int k = 0; k < RESOLVE_K; ++k) { sk[k] = 0x7fffffff; sg[k] = -1; }
    // A query may carry a SECOND window (the two-camera loop search, reference src/ORBmatcher.cc:625-721: the point is projected
    // into both cameras of the keyframe and the best candidate over both windows wins): its candidates simply follow the
    // first window's in the list, i.e. in the reference's visiting order (camera 1's loop runs before camera 2's).
    const int nwin = win2 ? 2 : 1;
    for (int wi = 0; wi < nwin; ++wi) {
    if (wi == 1) {
        const orbm_window* W2 = win2 + qi;
        x = W2->u; y = W2->v; r = W2->radius; cam = W2->cam; minLevel = W2->min_level; maxLevel = W2->max_level;
    }
    const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.invW));
    const int nMaxCellX = min(ORBM_GRID_COLS - 1, (int)ceilf((x - F.minX + r) * F.invW));
    const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.invH));
    const int nMaxCellY = min(ORBM_GRID_ROWS - 1, (int)ceilf((y - F.minY + r) * F.invH));
    const bool ok = nMinCellX < ORBM_GRID_COLS && nMaxCellX >= 0 && nMinCellY < ORBM_GRID_ROWS && nMaxCellY >= 0 &&
                    nMinCellX <= nMaxCellX && nMinCellY <= nMaxCellY && cam >= 0 && cam < F.n_cams;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    if (ok) {
        const int ny = nMaxCellY - nMinCellY + 1, ncells = (nMaxCellX - nMinCellX + 1) * ny;
        for (int cbase = 0; cbase < ncells; cbase += 64) {
            // this lane's cell of the chunk
            const int ci = cbase + lane;
            int cs = 0, cn = 0;
            if (ci < ncells) {
                const int ix = nMinCellX + ci / ny, iy = nMinCellY + ci % ny;
                const int cell = (cam * ORBM_GRID_COLS + ix) * ORBM_GRID_ROWS + iy;
                cs = F.cell_start[cell];
                cn = F.cell_start[cell + 1] - cs;
            }
            const int incl = wave_incl_scan(cn);  // inclusive prefix of the item counts over the lanes (DPP)
            const int items_in_chunk = __builtin_amdgcn_readlane(incl, 63);
            const int excl = incl - cn;
            for (int tbase = 0; tbase < items_in_chunk; tbase += 64) {
                const int t = tbase + lane;  // t-th item of the chunk in (cell, ascending index) order
                const bool valid = t < items_in_chunk;
                // owner lane = first lane whose inclusive prefix exceeds t (binary search over the wave)
                int lo = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int probe = lo + step - 1;
                    const int pv = __shfl(incl, probe);
                    if (pv <= t) lo += step;
                }
                const int oexcl = __shfl(excl, lo), ostart = __shfl(cs, lo);
                const int g = valid ? F.items[ostart + (t - oexcl)] : 0;
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
                if (pass && gate_right == 1) {
                    const float urg = F.uright[g];
                    if (urg > 0 && fabsf(ur - urg) > r) pass = false;   // a NaN `ur` never closes this gate
                }
                if (pass && gate_right == 2) {   // Fuse's reprojection-error gate (src/ORBmatcher.cc:2118-2143)
                    const float kpr = F.uright[g];
                    const float ex = x - F.un_x[g], ey = y - F.un_y[g];
                    if (kpr >= 0) {
                        const float er = ur - kpr;
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * inv_sigma2[F.octave[g]]) > 7.8) pass = false;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * inv_sigma2[F.octave[g]]) > 5.99) pass = false;
                    }
                }
                const unsigned long long mask = __ballot(pass);
                const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                int dist = 0;
                if (pass && with_dist) dist = ham256(q0, q1, F.desc[2 * g], F.desc[2 * g + 1]);
                if (pass && pos < cap) {
                    const size_t o = transposed ? (size_t)pos * nq + qi : (size_t)qi * cap + pos;
                    cand_idx[o] = g;
                    if (with_dist) cand_dist[o] = (uint16_t)dist;
                }
                if (topk) {  // merge this batch's survivors into the sorted shortlist (at most RESOLVE_K extractions)
                    // A frame search accepts only distance <= th_high, so farther candidates can neither win nor matter:
                    // they stay out of the shortlist and out of the "list longer than the shortlist" count (short_th =
                    // th_high there; 256 = keep everything for the top-2 / ratio-test search).
                    const bool elig = pass && !(occupied && occupied[g]) && dist <= short_th;
                    n_elig += __popcll(__ballot(elig));
                    int key = elig ? ((dist << 16) | pos) : 0x7fffffff;
#pragma unroll
                    for (int e = 0; e < RESOLVE_K; ++e) {
                        const int mn = (int)wave_min_u32((unsigned)key);   // keys are non-negative; DPP, no LDS crossbar
                        if (mn >= sk[RESOLVE_K - 1]) break;  // wave-uniform: nothing left that beats the shortlist tail
                        const int mg = __builtin_amdgcn_readlane(g, __ffsll((long long)__ballot(key == mn)) - 1);
                        if (key == mn) key = 0x7fffffff;     // positions are unique, so exactly one lane matches
                        int ck = mn, cg = mg;
#pragma unroll
                        for (int j = 0; j < RESOLVE_K; ++j)
                            if (ck < sk[j]) { const int tk = sk[j], tg = sg[j]; sk[j] = ck; sg[j] = cg; ck = tk; cg = tg; }
                    }
                }
                total += __popcll(mask);
            }
        }
    }
    }  // windows
    if (lane == 0) {
        cand_count[qi] = total;
        // what the resolve needs of a query besides its candidates: it never reads the query records themselves, which may
        // therefore live in pinned host memory (read once, here)
        if (qmeta) qmeta[qi] = make_int2(Q->blocks, __float_as_int(Q->angle));
        if (topk) {
#pragma unroll
            for (int k = 0; k < RESOLVE_K; ++k) {
                topk[(size_t)k * nq + qi] = sk[k];
                topk[(size_t)(RESOLVE_K + k) * nq + qi] = sg[k];
            }
            topk[(size_t)(2 * RESOLVE_K) * nq + qi] = n_elig;
        }
    }
}

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

// ------------------------------------------------------------------------------------------------ device frame build
struct CamFeat {
    const orb_keypoint* kps; const uint4* desc; const float* depth;
    int depth_stride, n, base;
};

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

__global__ __launch_bounds__(1024) void k_frame_build_small(CamFeat4 cams4, int* __restrict__ cam_start_out, const int* __restrict__ d_counts,
                                                            int* __restrict__ n_total_out, int n_cams, int n_total, float mbf,
                                                            float minX, float minY, float invW, float invH,
                                                            float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ ur, float* __restrict__ depth_out,
                                                            int* __restrict__ oct, float* __restrict__ ang,
                                                            orb_keypoint* __restrict__ kps_g, uint4* __restrict__ desc_g,
                                                            int* __restrict__ cell_start, int* __restrict__ items, HostMirror hm,
                                                            const int* __restrict__ cell_of_in, int desc_rows) {
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
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; wsum[w] = acc; acc += v; }
        s_start[ncell] = acc;
    }
    __syncthreads();
    int run = wsum[wave] + incl - mine;
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
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int c = 0; c < n_cams; ++c) {
        const int n = d_counts[c];
        cams[c].n = n; cams[c].base = base;
        cam_start[c] = base; trailer[c] = n;
        base += n;
    }
    cam_start[n_cams] = base;
    range[0] = base; range[1] = 0; range[2] = base;
}

// exclusive scan of cnt[0..n) into start[0..n], single 1024-thread block; cursor = copy of start
__global__ __launch_bounds__(1024) void k_scan_cells(const int* cnt, int n, int* __restrict__ start, int* cursor) {
    // cnt and cursor may alias (the per-cell counters are turned into insert cursors in place)
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int c0 = min(n, tid * per), c1 = min(n, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += cnt[c];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; wsum[w] = acc; acc += v; }
        start[n] = acc;
    }
    __syncthreads();
    int run = wsum[wave] + incl - mine;
    for (int c = c0; c < c1; ++c) { const int v = cnt[c]; start[c] = run; cursor[c] = run; run += v; }
}

__global__ __launch_bounds__(256) void k_scatter_cells(const int* __restrict__ cell_of, int n_total, int* __restrict__ cursor,
                                                       int* __restrict__ items, const int* __restrict__ n_dev) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (n_dev) n_total = *n_dev;
    if (g >= n_total) return;
    const int cell = cell_of[g];
    if (cell >= 0) items[atomicAdd(&cursor[cell], 1)] = g;
}

// ascending global index inside every cell (the atomics above scatter in arbitrary order; cells hold a handful of items)
__global__ __launch_bounds__(256) void k_sort_cells(const int* __restrict__ start, int ncell, int* __restrict__ items) {
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

// ------------------------------------------------------------------------------------------------ device resolve
// The reference resolves claims sequentially in query order (src/ORBmatcher.cc:3502-3614): a feature claimed by a query
// whose MapPoint is observed ("blocks") is invisible to every LATER query.  Query q therefore depends only on queries
// < q, and the sequential answer is the unique fixed point of:  choice[q] = first arg-min over q's ordered candidates
// that are not occupied and not claimed by a blocking query q' < q.  One workgroup iterates that map (Jacobi) until
// nothing changes -- after k sweeps the first k queries are final, in practice a handful of sweeps suffice.
// The claim table lives in LDS (one int per feature); entries carry the sweep number in the high 16 bits, decreasing,
// so atomicMin both selects the newest sweep and the lowest query index and no reset pass is needed.
// Candidates are read in the transposed layout [k*nq + i] (coalesced across the thread-per-query mapping).
// status[0]: 0 ok, 1 not converged within max_it (host falls back), 2 a candidate list exceeded cap (host retries);
// status[1] = nmatches, status[2] = sweeps, status[3] = longest candidate list.
constexpr int RESOLVE_MAX_Q = 65535;
MORB_PHASE_DECL(g_ph_res);

// RESOLVE_K (above): sorted shortlist per query built by k_project; a full rescan happens only when all of it is taken

// Evaluates query i against the current claim table.  `avail(g)` decides visibility.  Candidates are visited in the
// order given; FRAMES: first minimum.  POINTS: best + second (with multiplicity) and their levels.
// LDSQ: the per-query sweep state (shortlist features + distances, blocks flag, current choice) also lives in LDS, so a
// sweep touches no global memory at all; used whenever it fits next to the claim table.
template <bool POINTS, bool LDSQ>
__global__ __launch_bounds__(1024) void k_resolve(FrameDev F, const int2* __restrict__ qmeta /* {blocks, angle bits} */, int nq, int cap,
                                                  const int* __restrict__ cand_idx, const uint16_t* __restrict__ cand_dist,
                                                  const int* __restrict__ cand_count, const uint8_t* __restrict__ occupied,
                                                  const float* __restrict__ f_angle, int th_high, float nnratio,
                                                  int check_ori, int max_it, int* __restrict__ choice,
                                                  const int* __restrict__ topk /* (2*RESOLVE_K+1)*nq ints */,
                                                  int* __restrict__ match_of_feature, int* __restrict__ status, int tagb) {
    // tagb != 0: every result word carries this launch's sequence number in bits 20.. (values are small: a match word is
    // stored as value + 2), so a host that watches the pinned result memory can tell, word by word, what has arrived --
    // words written by different waves reach host memory in no particular order, a single "done" flag proves nothing.
    extern __shared__ __attribute__((aligned(16))) int s_claim[];  // two claim tables, one entry per feature each (capacity F.n_total)
    __shared__ int s_hist[ORBM_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_red, s_nres2[2];  // (s_nres2: rescans per sweep, instrumented build only)
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    MORB_PHASE(g_ph_res, 0);
    // the actual feature count is only needed by the last loops: nothing of the set-up waits for this load
    const int NT = F.n_total_dev ? *F.n_total_dev : F.n_total;
    // shortlist written by k_project: keys (dist << 16 | visiting position) and feature indices, sorted, occupied excluded
    const int* tk_key = topk;                             // [k*nq + i]
    const int* tk_g = topk + RESOLVE_K * nq;      // [k*nq + i]
    // LDS after the two claim tables: candidate counts u16[nq] (padded to 4 bytes); with LDSQ also
    //   choice[nq] | shortlist (distance << 16 | feature, 0xffff = none) [K][nq] | query angle [nq] | feature angle [F.n_total] | flags [nq] (u8)
    int* s_claim2 = s_claim + F.n_total;
    unsigned short* l_cnt = reinterpret_cast<unsigned short*>(s_claim + 2 * F.n_total);   // candidate count of every query
    int* l_choice = s_claim + 2 * F.n_total + (nq + 1) / 2;
    int* l_gd = l_choice + nq;
    float* l_ang = reinterpret_cast<float*>(l_gd + RESOLVE_K * nq);
    float* l_fang = l_ang + nq;
    unsigned char* l_fl = reinterpret_cast<unsigned char*>(l_fang + F.n_total);  // bit0 blocks, bit1 list > K, bits 2.. rotation bin + 1
    if (tid == 0) { s_red = 0; s_nres2[0] = 0; s_nres2[1] = 0; }
    for (int g = tid; g < F.n_total; g += T) {  // capacity-sized: rows past the real count are never referenced
        s_claim[g] = 0x7fffffff; s_claim2[g] = 0x7fffffff;
        if (LDSQ && !POINTS && check_ori) l_fang[g] = f_angle[g];
    }
    int mx = 0;
    for (int i = tid; i < nq; i += T) {  // every load of this pass is independent: one trip to HBM for the whole set-up
        const int cnt_i = cand_count[i];
        mx = max(mx, cnt_i);
        l_cnt[i] = (unsigned short)min(cnt_i, 65535);
        if (LDSQ) {
            l_choice[i] = -1;
            const int2 qm = qmeta[i];
            l_fl[i] = (unsigned char)((qm.x ? 1 : 0) | (topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K ? 2 : 0));
            l_ang[i] = __int_as_float(qm.y);
#pragma unroll
            for (int k = 0; k < RESOLVE_K; ++k) {
                // distance in the high half, feature in the low one; an empty slot (feature -1) reads 0xffff there (the
                // LDS-resident form is only chosen for frames below 65535 features)
                l_gd[k * nq + i] = (tk_key[k * nq + i] & 0xffff0000) | (tk_g[k * nq + i] & 0xffff);
            }
        } else {
            choice[i] = -1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if (lane == 0) atomicMax(&s_red, mx);  // one LDS atomic per wave: same-address atomics of a whole block serialise
    __syncthreads();
    const int maxcount = s_red;
    if (maxcount > cap) {
        if (tid == 0) { status[1] = tagb | 0; status[2] = tagb | 0; status[3] = tagb | maxcount; status[0] = tagb | 2; }
        return;
    }
    MORB_PHASE(g_ph_res, 2);
    constexpr int NEED = POINTS ? 2 : 1;
    int it = 0, changed = 1;
    for (; it < max_it && changed; ++it) {
        // Two claim tables alternate: sweep `it` READS the claims the previous sweep's choices left in `rd` (entries tagged
        // `tag`) and WRITES the claims of its own choices into `wr` (tagged `tag_next`), so a sweep is ONE pass over the
        // queries and one barrier.  Tags decrease, so atomicMin prefers the newer sweep over stale entries of the same
        // table (two sweeps old) and, within a sweep, the lowest query index.
        const int tag = (0x7ffe - it) << 16, tag_next = (0x7ffd - it) << 16;
        const int* rd = (it & 1) ? s_claim2 : s_claim;
        int* wr = (it & 1) ? s_claim : s_claim2;
#ifdef MORB_PHASE_CLOCKS
        int& s_nres = s_nres2[it & 1];
        if (tid == 0) s_nres2[(it + 1) & 1] = 0;
#endif
        int ch = 0;
        // The sweep is bound by the instruction count of its one workgroup (2000 queries on four SIMDs), so the walk is cut
        // in two: entries 0-1 first -- almost every query is decided there -- and entries 2..K-1 only for waves in which
        // some lane is still walking (wave-uniform branch).  A claim hides candidate g from query i when it carries this
        // sweep's read tag and a lower query index, i.e. lies in [tag, tag + i): one subtract and one unsigned compare
        // (older sweeps carry larger tags, 0x7fffffff is larger still).
        constexpr int K0 = 2;
        for (int base = 0; base < nq; base += T) {   // uniform trip count: the cooperative rescans below need whole waves
            const int i = base + tid;
            const bool valid = i < nq;
            int gk[RESOLVE_K], dk[RESOLVE_K], ck[RESOLVE_K];
            int fl = 0, old = -1, nc = -1;
            bool need_rescan = false;
            if (valid) {
                fl = LDSQ ? (int)l_fl[i] : ((qmeta[i].x ? 1 : 0) | (topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K ? 2 : 0));
                old = LDSQ ? l_choice[i] : choice[i];
                int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1, g2 = -1;
                int found = 0, taken = 0;
                bool walking = true;
                auto fetch = [&](int k) {
                    if (LDSQ) {
                        const int v = l_gd[k * nq + i];
                        gk[k] = (v & 0xffff) == 0xffff ? -1 : (v & 0xffff);
                        dk[k] = (int)((unsigned)v >> 16);
                    } else { gk[k] = tk_g[k * nq + i]; dk[k] = tk_key[k * nq + i] >> 16; }
                };
                auto walk = [&](int k) {
                    if (gk[k] < 0) walking = false;   // the shortlist is sorted: empty slots are at the end
                    if (walking) {
                        if ((unsigned)(ck[k] - tag) < (unsigned)i) ++taken;
                        else {
                            if (found == 0) { best = dk[k]; bidx = gk[k]; }
                            else { best2 = dk[k]; g2 = gk[k]; }
                            if (++found >= NEED) walking = false;
                        }
                    }
                };
#pragma unroll
                for (int k = 0; k < K0; ++k) fetch(k);
#pragma unroll
                for (int k = 0; k < K0; ++k) ck[k] = gk[k] >= 0 ? rd[gk[k]] : 0x7fffffff;
#pragma unroll
                for (int k = 0; k < K0; ++k) walk(k);
                if (__ballot(walking)) {
#pragma unroll
                    for (int k = K0; k < RESOLVE_K; ++k) fetch(k);
#pragma unroll
                    for (int k = K0; k < RESOLVE_K; ++k) ck[k] = gk[k] >= 0 ? rd[gk[k]] : 0x7fffffff;
#pragma unroll
                    for (int k = K0; k < RESOLVE_K; ++k) walk(k);
                }
                if (POINTS) { if (bidx >= 0) lvl = F.octave[bidx]; if (g2 >= 0) lvl2 = F.octave[g2]; }
                // the shortlist is exact unless it ran dry while longer lists exist (rare): rescanned right below
                need_rescan = found < NEED && (fl & 2) && taken > 0;
                if (!need_rescan && best <= th_high && bidx >= 0) {
                    nc = bidx;
                    if (POINTS && lvl == lvl2 && (float)best > nnratio * (float)best2) nc = -1;
                }
            }
            // Rescans, one query at a time by the whole wave that owns it, in place: full candidate list of the query, 64
            // candidates per round, keys (distance << 16 | visiting position) -- the smallest available key is the
            // sequential scan's first minimum, the next one its runner-up.  (A separate rescan phase behind a barrier cost
            // one more barrier and ~0.9 us per sweep that had any.)
            unsigned long long todo = __ballot(need_rescan);
#ifdef MORB_PHASE_CLOCKS
            if (todo && lane == 0 && it < 15) atomicAdd(&s_nres, __popcll(todo));
#endif
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int qi = __builtin_amdgcn_readlane(i, src);
                const int full = l_cnt[qi];
                int k1 = 0x7fffffff, k2 = 0x7fffffff, g1 = -1;
                for (int k0 = 0; k0 < full; k0 += 64) {
                    const int k = k0 + lane;
                    int key = 0x7fffffff, g = -1;
                    if (k < full) {
                        g = cand_idx[k * nq + qi];
                        const int d = cand_dist[k * nq + qi];
                        bool avail = !(occupied && occupied[g]);
                        if ((unsigned)(rd[g] - tag) < (unsigned)qi) avail = false;
                        if (avail) key = (d << 16) | k;
                    }
                    const int m1 = (int)wave_min_u32((unsigned)key);   // keys are non-negative: unsigned order == signed order
                    int m2 = 0x7fffffff;
                    if (POINTS) m2 = (int)wave_min_u32((unsigned)(key == m1 ? 0x7fffffff : key));
                    // merge the round's (m1 <= m2) into the running (k1 <= k2); the winner's feature comes along by readlane
                    if (m1 < k1) {
                        k2 = min(k1, m2); k1 = m1;
                        g1 = __builtin_amdgcn_readlane(g, __ffsll((long long)__ballot(key == m1)) - 1);   // positions are unique
                    }
                    else k2 = min(k2, m1);
                }
                int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1;   // (wave-uniform from here on)
                if (k1 != 0x7fffffff) {
                    best = k1 >> 16; bidx = g1;
                    if (POINTS) lvl = F.octave[bidx];
                }
                if (POINTS && k2 != 0x7fffffff) { best2 = k2 >> 16; lvl2 = F.octave[cand_idx[(k2 & 0xffff) * nq + qi]]; }
                int rnc = -1;
                if (best <= th_high && bidx >= 0) {
                    rnc = bidx;
                    if (POINTS && lvl == lvl2 && (float)best > nnratio * (float)best2) rnc = -1;
                }
                if (lane == src) nc = rnc;
            }
            if (valid) {
                if (nc != old) { ch = 1; if (LDSQ) l_choice[i] = nc; else choice[i] = nc; }
                if (nc >= 0 && (fl & 1)) atomicMin(&wr[nc], tag_next | i);  // what the next sweep sees
            }
        }
#ifdef MORB_PHASE_CLOCKS
        __syncthreads();
        if (tid == 0 && it < 15) g_ph_res[40 + it] = (unsigned long long)s_nres;
#endif
        if (it == 0) MORB_PHASE(g_ph_res, 20); else if (it == 5) MORB_PHASE(g_ph_res, 24);
        if (it == 0) MORB_PHASE(g_ph_res, 22); else if (it == 5) MORB_PHASE(g_ph_res, 26);
        changed = __syncthreads_or(ch);
        MORB_PHASE(g_ph_res, min(3 + it, 50));
    }
    if (changed) {  // ran out of sweeps
        if (tid == 0) { status[1] = tagb | 0; status[2] = tagb | it; status[3] = tagb | maxcount; status[0] = tagb | 1; }
        return;
    }
    // owners: the last claimant in query order (claims after a blocking one are impossible, so max index == final owner)
    for (int g = tid; g < NT; g += T) s_claim[g] = -1;
    if (tid < ORBM_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_red = 0;
    __syncthreads();
    MORB_PHASE(g_ph_res, 52);
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int acc = 0;
    for (int i = tid; i < nq; i += T) {
        const int c = LDSQ ? l_choice[i] : choice[i];
        if (c < 0) continue;
        ++acc;
        atomicMax(&s_claim[c], i);
        if (!POINTS && check_ori) {
            float rot = LDSQ ? l_ang[i] - l_fang[c] : __int_as_float(qmeta[i].y) - f_angle[c];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            const bool inr = bin >= 0 && bin < ORBM_HISTO_LENGTH;
            if (LDSQ) l_fl[i] = (unsigned char)((l_fl[i] & 3) | ((inr ? bin + 1 : 0) << 2));
            // most matches of a frame share a rotation bin: up to three bins of the wave (those of its first lanes) are
            // counted with one atomic each, whatever is left (scattered bins: few lanes per address) goes in directly
            unsigned long long todo = __ballot(inr);
            for (int rounds = 0; todo && rounds < 3; ++rounds) {
                const int b0 = __builtin_amdgcn_readlane(bin, __ffsll((long long)todo) - 1);
                const unsigned long long same = __ballot(inr && bin == b0);
                if (inr && bin == b0 && lane == __ffsll((long long)same) - 1) atomicAdd(&s_hist[b0], __popcll(same));
                todo &= ~same;
            }
            if (inr && ((todo >> lane) & 1)) atomicAdd(&s_hist[bin], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) atomicAdd(&s_red, acc);
    __syncthreads();
    MORB_PHASE(g_ph_res, 53);
    if (!POINTS && check_ori) {
        if (tid < 64) {
            // ComputeThreeMaxima (reference src/ORBmatcher.cc:3948-3989).  Its scan with strict '>' keeps the three fullest
            // non-empty bins, the earlier bin first among equals: bin b's place is the number of bins that beat it
            // (fuller, or as full and earlier) -- 30 readlanes on one wave instead of 30 dependent LDS reads on one thread.
            const int sv = tid < ORBM_HISTO_LENGTH ? s_hist[tid] : 0;
            int rank = 0;
#pragma unroll
            for (int j = 0; j < ORBM_HISTO_LENGTH; ++j) {
                const int sj = __builtin_amdgcn_readlane(sv, j);
                rank += (sj > sv || (sj == sv && j < tid)) ? 1 : 0;
            }
            const bool in = tid < ORBM_HISTO_LENGTH && sv > 0;
            const unsigned long long r1 = __ballot(in && rank == 0), r2 = __ballot(in && rank == 1), r3 = __ballot(in && rank == 2);
            int i1 = r1 ? __ffsll((long long)r1) - 1 : -1, i2 = r2 ? __ffsll((long long)r2) - 1 : -1, i3 = r3 ? __ffsll((long long)r3) - 1 : -1;
            const int m1 = i1 >= 0 ? __builtin_amdgcn_readlane(sv, i1) : 0, m2 = i2 >= 0 ? __builtin_amdgcn_readlane(sv, i2) : 0,
                      m3 = i3 >= 0 ? __builtin_amdgcn_readlane(sv, i3) : 0;
            if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
            else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
            if (tid == 0) { s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3; }
        }
        __syncthreads();
        MORB_PHASE(g_ph_res, 54);
        int rej = 0;
        for (int i = tid; i < nq; i += T) {
            const int c = LDSQ ? l_choice[i] : choice[i];
            if (c < 0) continue;
            int bin;
            if (LDSQ) {
                bin = (int)(l_fl[i] >> 2) - 1;  // -1: outside the histogram, never rejected
            } else {
                float rot = __int_as_float(qmeta[i].y) - f_angle[c];
                if (rot < 0.0) rot += 360.0f;
                bin = (int)roundf(rot * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
            }
            if (bin >= 0 && bin < ORBM_HISTO_LENGTH && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                s_claim[c] = -2;  // every writer stores -2; owners were settled before the barrier
                ++rej;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rej += __shfl_xor(rej, o);
        if (lane == 0) atomicSub(&s_red, rej);
        __syncthreads();
    }
    MORB_PHASE(g_ph_res, 60);
    for (int g = tid; g < NT; g += T) match_of_feature[g] = tagb ? (tagb | (s_claim[g] + 2)) : s_claim[g];
    if (tid == 0) { status[1] = tagb | s_red; status[2] = tagb | it; status[3] = tagb | maxcount; status[0] = tagb | 0; }
    MORB_PHASE(g_ph_res, 61);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0) g_ph_res[62] = (unsigned long long)it;
#endif
}

// ---- the same resolve for frames whose claim tables do not fit LDS (beyond ~18 000 features: 8 cameras x 4000), spread
// over the whole chip.  The two claim tables, the choices and the owner table live in HBM (L2-resident); one launch per
// sweep (a grid-wide barrier is exactly what a kernel boundary is), a fixed number of sweeps is enqueued and a sweep
// that finds "nothing changed" in its predecessor's flag does nothing, so no host round trip sits between sweeps.
// state: [0] longest candidate list, [1] matches, [2..4] kept rotation bins, [8..8+RS_MAX_SWEEPS) changed flags,
//        [48..78) rotation histogram.
constexpr int RS_MAX_SWEEPS = 24;
constexpr int RS_STATE_INTS = 80;

__global__ __launch_bounds__(256) void k_rs_init(int n_cap, int nq, int* __restrict__ tab0, int* __restrict__ tab1,
                                                 int* __restrict__ owner, int* __restrict__ choice,
                                                 const int* __restrict__ cand_count, int* __restrict__ state) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_cap) { tab0[i] = 0x7fffffff; tab1[i] = 0x7fffffff; owner[i] = -1; }
    int mx = 0;
    if (i < nq) { choice[i] = -1; mx = cand_count[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(&state[0], mx);
}

template <bool POINTS>
__global__ __launch_bounds__(256) void k_rs_sweep(FrameDev F, const orbm_query* __restrict__ q, int nq, int cap, int it,
                                                  const int* __restrict__ cand_idx, const uint16_t* __restrict__ cand_dist,
                                                  const int* __restrict__ cand_count, const uint8_t* __restrict__ occupied,
                                                  int th_high, float nnratio, int* __restrict__ choice,
                                                  const int* __restrict__ topk, const int* __restrict__ rd,
                                                  int* __restrict__ wr, int* __restrict__ state) {
    if (state[0] > cap) return;                        // a candidate list overflowed: reported by k_rs_write
    if (it > 0 && state[8 + it - 1] == 0) return;      // the previous sweep changed nothing: fixed point reached
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr int NEED = POINTS ? 2 : 1;
    const int tag = (0x7ffe - it) << 16, tag_next = (0x7ffd - it) << 16;
    int ch = 0;
    if (i < nq) {
        const int* tk_key = topk;
        const int* tk_g = topk + RESOLVE_K * nq;
        int sg[RESOLVE_K], sd[RESOLVE_K];
#pragma unroll
        for (int k = 0; k < RESOLVE_K; ++k) { sg[k] = tk_g[k * nq + i]; sd[k] = tk_key[k * nq + i] >> 16; }
        const bool longer = topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K;
        const int old = choice[i], bl = q[i].blocks;
        int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1;
        int found = 0, taken = 0;
#pragma unroll
        for (int k = 0; k < RESOLVE_K; ++k) {
            if (found >= NEED) break;
            const int g = sg[k];
            if (g < 0) break;
            const int cl = rd[g];
            if ((cl >> 16) == (tag >> 16) && (cl & 0xffff) < i) { ++taken; continue; }
            if (found == 0) { best = sd[k]; bidx = g; if (POINTS) lvl = F.octave[g]; }
            else { best2 = sd[k]; lvl2 = F.octave[g]; }
            ++found;
        }
        if (found < NEED && longer && taken > 0) {  // the shortlist ran dry: scan the whole list (rare), 8 loads in flight
            best = 256; best2 = 256; lvl = -1; lvl2 = -1; bidx = -1;
            const int full = cand_count[i];
            for (int k0 = 0; k0 < full; k0 += 8) {
                int cg[8], cdist[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = min(k0 + u, full - 1);
                    cg[u] = cand_idx[k * nq + i];
                    cdist[u] = cand_dist[k * nq + i];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u >= full) continue;
                    const int g = cg[u];
                    if (occupied && occupied[g]) continue;
                    const int cl = rd[g];
                    if ((cl >> 16) == (tag >> 16) && (cl & 0xffff) < i) continue;
                    const int d = cdist[u];
                    if (POINTS) {
                        if (d < best) { best2 = best; best = d; lvl2 = lvl; lvl = F.octave[g]; bidx = g; }
                        else if (d < best2) { lvl2 = F.octave[g]; best2 = d; }
                    } else if (d < best) { best = d; bidx = g; }
                }
            }
        }
        int nc = -1;
        if (best <= th_high && bidx >= 0) {
            nc = bidx;
            if (POINTS && lvl == lvl2 && (float)best > nnratio * (float)best2) nc = -1;
        }
        if (nc != old) { ch = 1; choice[i] = nc; }
        if (nc >= 0 && bl) atomicMin(&wr[nc], tag_next | i);
    }
    if (__syncthreads_or(ch) && threadIdx.x == 0) atomicOr(&state[8 + it], 1);
}

// owners (last claimant in query order) + rotation histogram + match count
__global__ __launch_bounds__(256) void k_rs_owner(const orbm_query* __restrict__ q, int nq, int cap, const int* __restrict__ choice,
                                                  const float* __restrict__ f_angle, int check_ori, int* __restrict__ owner,
                                                  int* __restrict__ state) {
    if (state[0] > cap || state[8 + RS_MAX_SWEEPS - 1] != 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = i < nq ? choice[i] : -1;
    if (c >= 0) atomicMax(&owner[c], i);
    const unsigned long long any = __ballot(c >= 0);
    if (lane == 0 && any) atomicAdd(&state[1], __popcll(any));
    if (check_ori) {
        int bin = -1;
        if (c >= 0) {
            float rot = q[i].angle - f_angle[c];
            if (rot < 0.0) rot += 360.0f;
            bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            if (bin < 0 || bin >= ORBM_HISTO_LENGTH) bin = -1;
        }
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {  // one atomic per distinct bin of the wave
            const int b0 = __shfl(bin, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(bin == b0);
            if (bin == b0 && lane == __ffsll((long long)same) - 1) atomicAdd(&state[48 + b0], __popcll(same));
            todo &= ~same;
        }
    }
}

// ComputeThreeMaxima (every block, redundantly) + rejection of the matches outside the three fullest rotation bins
__global__ __launch_bounds__(256) void k_rs_reject(const orbm_query* __restrict__ q, int nq, int cap, const int* __restrict__ choice,
                                                   const float* __restrict__ f_angle, int* __restrict__ owner,
                                                   int* __restrict__ state) {
    if (state[0] > cap || state[8 + RS_MAX_SWEEPS - 1] != 0) return;
    __shared__ int s_keep[3];
    if (threadIdx.x == 0) {  // reference src/ORBmatcher.cc:3948-3989
        int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            const int sz = state[48 + b];
            if (sz > m1) { m3 = m2; i3 = i2; m2 = m1; i2 = i1; m1 = sz; i1 = b; }
            else if (sz > m2) { m3 = m2; i3 = i2; m2 = sz; i2 = b; }
            else if (sz > m3) { m3 = sz; i3 = b; }
        }
        if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
        else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool rej = false;
    if (i < nq) {
        const int c = choice[i];
        if (c >= 0) {
            float rot = q[i].angle - f_angle[c];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < ORBM_HISTO_LENGTH && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                owner[c] = -2;  // every writer stores -2; the owners were settled by the previous kernel
                rej = true;
            }
        }
    }
    const unsigned long long r = __ballot(rej);
    if (lane == 0 && r) atomicSub(&state[1], __popcll(r));
}

__global__ __launch_bounds__(256) void k_rs_write(int NT_host, const int* __restrict__ n_total_dev, int cap, const int* __restrict__ owner,
                                                  const int* __restrict__ state, int* __restrict__ match_of_feature,
                                                  int* __restrict__ status) {
    const int NT = n_total_dev ? *n_total_dev : NT_host;
    const bool overflow = state[0] > cap, stuck = state[8 + RS_MAX_SWEEPS - 1] != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int sweeps = 0;
        for (int k = 0; k < RS_MAX_SWEEPS; ++k) sweeps += state[8 + k] ? 1 : 0;
        status[0] = overflow ? 2 : (stuck ? 1 : 0);
        status[1] = (overflow || stuck) ? 0 : state[1];
        status[2] = sweeps + 1;
        status[3] = state[0];
    }
    if (overflow || stuck) return;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < NT) match_of_feature[g] = owner[g];
}

// Multi-GPU exchange: `gathered` holds one block per rank (rank order), each = cap_rows descriptor rows (the rank's
// cameras packed back to back) + the count trailer.  The rows in use are copied into one contiguous list in global camera
// order; block (0, 0) also writes the camera starts, the {features, first query, queries} triple of rank `rank`, and a
// copy of all counts into mapped pinned memory.  Every block recomputes the few prefix sums it needs from the trailers.
// A remote trailer is data from another process: every count is clamped to what is left of its rank's cap_rows rows (a
// mismatched or corrupt block can then neither run past its own block nor past the contiguous list), and the number of
// counts that had to be clamped is reported in h_counts[n_cams + 1] (orbm_cross_top2_gathered_collect turns it into an error).
__device__ __forceinline__ int repack_count(const int* __restrict__ tail, int c, int& room, int& bad) {
    const int raw = tail[c];
    const int n = min(max(raw, 0), room);
    bad += (n != raw);
    room -= n;
    return n;
}

__global__ __launch_bounds__(256) void k_repack_gathered(const uint8_t* __restrict__ gathered, int world, size_t block_bytes,
                                                         int cap_rows, int cams_per_rank, int rank, uint4* __restrict__ dst,
                                                         int* __restrict__ cam_start, int* __restrict__ range,
                                                         int* __restrict__ h_counts) {
    const int r = blockIdx.y;
    int goff = 0, n_r = 0, own_off = 0, own_n = 0, total = 0, bad = 0;
    for (int rr = 0; rr < world; ++rr) {
        const int* tail = reinterpret_cast<const int*>(gathered + (size_t)rr * block_bytes + (size_t)cap_rows * 32);
        int nr = 0, room = cap_rows;
        for (int c = 0; c < cams_per_rank; ++c) nr += repack_count(tail, c, room, bad);
        if (rr == r) { goff = total; n_r = nr; }
        if (rr == rank) { own_off = total; own_n = nr; }
        total += nr;
    }
    if (blockIdx.x == 0 && r == 0 && threadIdx.x == 0) {
        int run = 0;
        for (int rr = 0; rr < world; ++rr) {
            const int* tail = reinterpret_cast<const int*>(gathered + (size_t)rr * block_bytes + (size_t)cap_rows * 32);
            int room = cap_rows, ignore = 0;
            for (int c = 0; c < cams_per_rank; ++c) {
                const int n = repack_count(tail, c, room, ignore);
                cam_start[rr * cams_per_rank + c] = run;
                h_counts[rr * cams_per_rank + c] = n;
                run += n;
            }
        }
        cam_start[world * cams_per_rank] = run;
        range[0] = total; range[1] = own_off; range[2] = own_n;
        h_counts[world * cams_per_rank] = own_n;
        h_counts[world * cams_per_rank + 1] = bad;
    }
    const uint4* src = reinterpret_cast<const uint4*>(gathered + (size_t)r * block_bytes);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n_r; i += gridDim.x * 256) dst[2 * (size_t)goff + i] = src[i];
}

// Queries are features [q_off, q_off + nq) (the cameras this process owns); outputs are indexed from 0.
__global__ __launch_bounds__(64 * TOP2_WAVES) void k_cross_top2(const uint4* __restrict__ desc, int n_total,
                                                                const int* __restrict__ cam_start, int n_cams, int q_off,
                                                                int nq, int* __restrict__ best_idx,
                                                                int* __restrict__ best_dist, int* __restrict__ second_dist,
                                                                const int* __restrict__ d_range) {
    // counts only known on the device ({features, first query, queries}): the launch was sized for the capacity
    if (d_range) { n_total = d_range[0]; q_off = d_range[1]; nq = d_range[2]; }
    if (blockIdx.x * 64 >= nq) return;
    __shared__ int sb[TOP2_WAVES][64], ss[TOP2_WAVES][64], si[TOP2_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = blockIdx.x * 64 + lane;
    const int qc = q_off + (qi < nq ? qi : nq - 1);
    const uint4 q0 = desc[2 * qc], q1 = desc[2 * qc + 1];
    int c = 0;
    while (c + 1 < n_cams && qc >= cam_start[c + 1]) ++c;
    const int seg0 = cam_start[c], seg1 = cam_start[c + 1], seglen = seg1 - seg0;

    const int S = gridDim.y;
    const int slice = (n_total + S - 1) / S;
    const int s0 = blockIdx.y * slice, s1 = min(n_total, s0 + slice);
    const int chunk = (max(s1 - s0, 0) + TOP2_WAVES - 1) / TOP2_WAVES;
    const int j0 = s0 + wave * chunk, j1 = min(s1, j0 + chunk);
    int b = 256, s = 256, bi = -1;
    // 64 consecutive queries nearly always belong to one camera: then the own camera's rows are skipped as a range
    // (for a 2-camera rig that is half of all pairs) and the per-pair segment test disappears
    const int seg0u = __builtin_amdgcn_readfirstlane(seg0), seg1u = __builtin_amdgcn_readfirstlane(seg1);
    if (__all(seg0 == seg0u)) {
        top2_scan(desc, j0, min(j1, seg0u), 0, q0, q1, b, s, bi);                 // cameras in front of the own one
        top2_scan(desc, max(j0, seg1u), j1, seg1u - seg0u, q0, q1, b, s, bi);     // cameras behind it: index minus the own count
    } else {
        for (int j = j0; j < j1; ++j) {
            const uint4 a0 = desc[2 * j], a1 = desc[2 * j + 1];
            int d = (int)ham256_chain(a0, a1, q0, q1);
            d = (j >= seg0 && j < seg1) ? 256 : d;               // own camera: distance 256 never registers
            const int jj = j < seg0 ? j : j - seglen;            // index in the concatenation of the other cameras
            s = min(s, max(b, d));
            bi = d < b ? jj : bi;
            b = min(b, d);
        }
    }
    sb[wave][lane] = b; ss[wave][lane] = s; si[wave][lane] = bi;
    __syncthreads();
    if (wave == 0 && qi < nq) {
        int B = 256, Sd = 256, I = -1;
#pragma unroll
        for (int w = 0; w < TOP2_WAVES; ++w) {
            const int b2 = sb[w][lane], s2 = ss[w][lane], i2 = si[w][lane];
            Sd = min(min(Sd, s2), max(B, b2));
            I = b2 < B ? i2 : I;
            B = min(B, b2);
        }
        const size_t o = (size_t)blockIdx.y * nq + qi;
        best_idx[o] = I; best_dist[o] = B; second_dist[o] = Sd;
    }
}

// The same search on the matrix cores: k_hamming_top2_mfma's tiling (a wave keeps 64 queries as B fragments, the workgroup
// expands 64 references per step into LDS, the accumulators come out as sort keys distance << 6 | row) with queries and
// references taken from ONE descriptor list and the rows of a query's OWN camera left out:
//   * a tile that lies inside the own camera of every query of the wave is skipped altogether (no MFMA, no key updates) -- 64
//     consecutive queries nearly always belong to one camera, so 1/n_cams of all pairs costs nothing;
//   * a tile that touches the own segment of some query of the wave takes the masked path (those rows enter as KEY_NONE,
//     exactly like rows past the end of the slice); every other tile takes the unmasked path of the generic kernel;
//   * the reported index is the position in the concatenation of the OTHER cameras (index minus the own count behind it).
// Counts known only on the device come through d_range = {features, first query, queries}; the launch is then sized for the
// capacity, slices beyond the features produce (256, 256, -1) partials and query blocks beyond the queries return at once.
// grid.x = reference slices (partials for k_top2_merge when > 1), grid.y = 256 queries.
__global__ __launch_bounds__(64 * MM_WAVES) void k_cross_top2_mfma(const uint32_t* __restrict__ desc, int n_total,
                                                                  const int* __restrict__ cam_start, int n_cams, int q_off, int nq,
                                                                  int slice_len, int* __restrict__ p_idx, int* __restrict__ p_best,
                                                                  int* __restrict__ p_second, const int* __restrict__ d_range) {
    __shared__ mm_i32x4 s_tile[2][2 * 8 * 64];
    if (d_range) { n_total = d_range[0]; q_off = d_range[1]; nq = d_range[2]; }
    if ((int)blockIdx.y * MM_Q_PER_BLOCK >= nq) return;   // (uniform over the workgroup; implies nq >= 1 and n_total >= 1 below)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * MM_Q_PER_BLOCK + wave * 64;
    const uint32_t* __restrict__ q = desc + (size_t)q_off * 8;
    const uint32_t* __restrict__ r = desc;
    const int nr = n_total;

    mm_i32x4 bq[2][8];
    int seg0[2], seg1[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bq[g][ks] = mm_expand16(h ? (w[ks] >> 16) : (w[ks] & 0xffffu));
        const int qc = q_off + qi;
        int cam = 0;
        while (cam + 1 < n_cams && qc >= cam_start[cam + 1]) ++cam;
        seg0[g] = cam_start[cam]; seg1[g] = cam_start[cam + 1];
    }
    // the union of the wave's own segments: [seg_lo, seg_hi); `one_seg`: every query of the wave has the same own camera
    const int first0 = __builtin_amdgcn_readfirstlane(seg0[0]), first1 = __builtin_amdgcn_readfirstlane(seg1[0]);
    const bool one_seg = __all(seg0[0] == first0 && seg0[1] == first0 && seg1[0] == first1 && seg1[1] == first1);
    const int seg_lo = (int)wave_min_u32((unsigned)min(seg0[0], seg0[1]));
    const int seg_hi = (int)(0x7fffffffu - wave_min_u32(0x7fffffffu - (unsigned)max(seg1[0], seg1[1])));
    mm_i32x16 cinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[e] = 8192 + (e & 3) + 8 * (e >> 2) + 4 * h;

    const int s0 = blockIdx.x * slice_len, s1 = min(nr, s0 + slice_len);  // slice_len is a multiple of 64
    const int n_tiles = s1 > s0 ? (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE : 0;
    auto fetch = [&](int t) {
        const int rr = max(0, min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1));
        return *reinterpret_cast<const uint2*>(r + (size_t)rr * 8 + wave * 2);
    };
    auto deposit = [&](int buf, uint2 w) {
        mm_i32x4* base = &s_tile[buf][(h * 8 + wave * 2) * 64 + c];
        base[0] = mt_expand16(w.x & 0xffffu);
        base[32] = mt_expand16(w.x >> 16);
        base[64] = mt_expand16(w.y & 0xffffu);
        base[96] = mt_expand16(w.y >> 16);
    };

    constexpr uint32_t KEY_NONE = 256u << 6;
    uint32_t kb[2] = {KEY_NONE, KEY_NONE}, ks2[2] = {KEY_NONE, KEY_NONE};
    int where[2] = {-1, -1};  // (block << 5 | row) of the best key, block = 2 * tile + a
    if (n_tiles > 0) deposit(0, fetch(0));
    uint2 nxt = fetch(1);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        const int tj0 = s0 + t * MM_R_TILE;                          // first reference of the tile
        const int valid = s1 - tj0;                                  // references of this tile inside the slice
        const bool touches = tj0 < seg_hi && tj0 + MM_R_TILE > seg_lo;  // wave-uniform
        const bool skip = one_seg && tj0 >= seg_lo && tj0 + min(valid, MM_R_TILE) <= seg_hi;   // the whole tile is the wave's own camera
        if (!skip) {
            mm_i32x16 acc[2][2];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const mm_i32x4 a0 = s_tile[buf][ks * 64 + lane], a1 = s_tile[buf][(8 + ks) * 64 + lane];
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0][ks], ks ? acc[0][0] : cinit, 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[1][ks], ks ? acc[0][1] : cinit, 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[0][ks], ks ? acc[1][0] : cinit, 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[1][ks], ks ? acc[1][1] : cinit, 0, 0, 0);
            }
            deposit(buf ^ 1, nxt);
            nxt = fetch(t + 2);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint32_t before = kb[g];
                    if (valid >= MM_R_TILE && !touches) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const uint32_t key = (uint32_t)acc[a][g][e];
                            ks2[g] = mt_umed3(kb[g], ks2[g], key);
                            kb[g] = min(kb[g], key);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int local = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            const int j = tj0 + local;
                            const bool out = local >= valid || (j >= seg0[g] && j < seg1[g]);   // past the slice, or the query's own camera
                            const uint32_t key = out ? KEY_NONE : (uint32_t)acc[a][g][e];
                            ks2[g] = mt_umed3(kb[g], ks2[g], key);
                            kb[g] = min(kb[g], key);
                        }
                    }
                    where[g] = kb[g] != before ? (((2 * t + a) << 5) | (int)(kb[g] & 31u)) : where[g];
                    kb[g] &= ~63u;
                }
        } else {
            deposit(buf ^ 1, nxt);
            nxt = fetch(t + 2);
        }
        __syncthreads();
    }
    // the two half-waves hold disjoint references of the same query: full keys distance << 16 | index decide
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t mine_b = ((kb[g] >> 6) << 16) | (uint32_t)(where[g] & 0xffff), mine_s = (ks2[g] >> 6) << 16 | 0xffffu;
        const uint32_t ob = (uint32_t)__shfl_xor((int)mine_b, 32), os = (uint32_t)__shfl_xor((int)mine_s, 32);
        const uint32_t nb = min(mine_b, ob), ns = min(max(mine_b, ob), min(mine_s, os));
        const int qrow = q0 + g * 32 + c;
        if (h == 0 && qrow < nq) {
            const size_t o = (size_t)blockIdx.x * nq + qrow;
            const int best = (int)(nb >> 16);
            const int j = s0 + (int)(nb & 0xffffu);
            p_best[o] = best;
            p_idx[o] = best < 256 ? (j < seg0[g] ? j : j - (seg1[g] - seg0[g])) : -1;   // index among the other cameras
            p_second[o] = (int)min(ns >> 16, 256u);
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// Plan of an exhaustive top-2: matrix-core kernel (256 queries x one reference slice per workgroup; slices are multiples
// of 64 references and at most 65536 long: 16-bit indices in the keys) when both sides have at least a tile, else the
// one-query-per-lane kernel.  S = number of reference slices (> 1 needs scratch for the partials).
struct Top2Plan { bool mfma; int S; int slice_len; };
std::atomic<int> g_matrix_cores{-1};  // orbm_use_matrix_cores: -1 = environment default

Top2Plan top2_plan(int nq, int nr, bool have_scratch = true) {
    static const int mfma_env = [] { const char* e = getenv("MORB_TOP2_MFMA"); return e ? atoi(e) : 1; }();
    const int forced = g_matrix_cores.load(std::memory_order_relaxed);
    Top2Plan p{false, 1, nr};
    if ((forced < 0 ? mfma_env : forced) && nq >= 64 && nr >= MM_R_TILE) {
        // Two workgroups are resident per CU (229 registers): a launch runs in ceil(workgroups / slots) rounds of
        // (tiles per slice + ~2) steps; take the slice count that minimises the product (32 000 x 32 000: 8 slices = 1000
        // workgroups = 1.95 rounds instead of 9 slices = 2.2 rounds, i.e. three).
        static const int slots = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return std::max(1, cus) * 2;
        }();
        const int qblocks = (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK;
        const int s_max = have_scratch ? std::min(64, std::max(1, nr / (4 * MM_R_TILE))) : 1;  // at least four tiles per slice
        double best_cost = 1e300;
        for (int S = 1; S <= s_max; ++S) {
            const int len = ((nr + S - 1) / S + MM_R_TILE - 1) / MM_R_TILE * MM_R_TILE;
            if (len > 65536) continue;
            const int s_eff = (nr + len - 1) / len;
            const long long wgs = (long long)s_eff * qblocks;
            const double cost = (double)((wgs + slots - 1) / slots) * (len / MM_R_TILE + 2.0) + 0.05 * s_eff;
            if (cost < best_cost) { best_cost = cost; p.mfma = true; p.slice_len = len; p.S = s_eff; }
        }
        if (p.mfma) return p;
    }
    const int qblocks = (nq + 63) / 64;
    int S = (128 + qblocks - 1) / qblocks;               // target >= 128 blocks x 16 waves = 2048 waves
    S = std::min(S, std::max(1, nr / (TOP2_WAVES * 16)));  // keep >= 16 references per wave
    p.S = have_scratch ? std::max(1, std::min(S, 64)) : 1;
    return p;
}

int launch_top2(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_bi, int32_t* d_bd, int32_t* d_sd,
                void* d_scratch, const Top2Plan& plan, hipStream_t st) {
    const int S = plan.S;
    int* p = (int*)d_scratch;
    int *p_idx = S > 1 ? p : d_bi, *p_best = S > 1 ? p + (size_t)S * nq : d_bd, *p_second = S > 1 ? p + 2 * (size_t)S * nq : d_sd;
    if (plan.mfma)
        hipLaunchKernelGGL(k_hamming_top2_mfma, dim3(S, (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK), dim3(64 * MM_WAVES), 0, st,
                           (const uint32_t*)d_q, nq, (const uint32_t*)d_r, nr, plan.slice_len, p_idx, p_best, p_second);
    else
        hipLaunchKernelGGL(k_hamming_top2, dim3((nq + 63) / 64, S), dim3(64 * TOP2_WAVES), 0, st, (const uint4*)d_q, nq,
                           (const uint4*)d_r, nr, p_idx, p_best, p_second);
    if (S > 1)
        hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, p_idx, p_best, p_second, S, nq, d_bi,
                           d_bd, d_sd);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

// number of reference slices of the one-query-per-lane kernels (k_cross_top2): enough blocks to give every SIMD a wave
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
    // Matrix-core path: rows of 16-byte chunks, enough queries to fill 64-query waves.  MORB_MATRIX_MFMA=0 keeps the VALU kernel.
    static const int mfma_env = [] { const char* e = getenv("MORB_MATRIX_MFMA"); return e ? atoi(e) : 1; }();
    const int forced = g_matrix_cores.load(std::memory_order_relaxed);
    if ((forced < 0 ? mfma_env : forced) && (nr % 8 == 0) && (((uintptr_t)d_out & 15) == 0) && nq >= 64 && nr >= MM_R_TILE && (((uintptr_t)d_r & 7) == 0)) {
        const int n_tiles = (nr + MM_R_TILE - 1) / MM_R_TILE, qblocks = (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK;
        // Tiles per workgroup: three workgroups are resident per CU (168 registers, 48 KB LDS), a launch runs in
        // ceil(workgroups / slots) rounds of (tiles + ~1.5 for the query expansion) steps each; take the count that minimises
        // that product (at 32 000 x 32 000: 21 tiles -> 3000 workgroups = 3.9 rounds, 411 us, where 32 tiles -> 2.6 rounds
        // took 422 and 16 tiles -> 5.2 rounds 431).
        static const int slots = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return std::max(1, cus) * 3;
        }();
        int tpb = 4;
        double best_cost = 1e300;
        for (int c = 4; c <= 32; ++c) {
            const long long wgs = (long long)((n_tiles + c - 1) / c) * qblocks;
            const double cost = (double)((wgs + slots - 1) / slots) * (c + 1.5);
            if (cost < best_cost) { best_cost = cost; tpb = c; }
        }
        static const int tpb_env = [] { const char* e = getenv("MORB_MATRIX_TPB"); return e ? atoi(e) : 0; }();
        if (tpb_env > 0) tpb = tpb_env;
        dim3 g2((n_tiles + tpb - 1) / tpb, qblocks);
        hipLaunchKernelGGL(k_hamming_matrix_mfma, g2, dim3(64 * MM_WAVES), 0, st, (const uint32_t*)d_q, nq, (const uint32_t*)d_r, nr,
                           d_out, tpb);
        MORB_HIP(hipGetLastError());
        return ORB_OK;
    }
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
constexpr size_t ORBM_BLOCK_TRAILER = 256;  // bytes behind the descriptor rows of a frame: int32 per-camera counts

struct FrameBufs {  // device storage of one frame; recycled through the matcher's pool (no hipMalloc per frame)
    DevBuf<float> d_x, d_y, d_ur, d_depth, d_ang;
    DevBuf<int32_t> d_oct, d_cell_start, d_items, d_cell_of, d_cursor, d_cam_start, d_ntotal;
    DevBuf<uint8_t> d_desc;
    DevBuf<orb_keypoint> d_kps;
    DevBuf<CamFeat> d_cams;
    void release() {
        d_x.release(); d_y.release(); d_ur.release(); d_depth.release(); d_ang.release(); d_oct.release();
        d_cell_start.release(); d_items.release(); d_cell_of.release(); d_cursor.release(); d_cam_start.release(); d_ntotal.release();
        d_desc.release(); d_kps.release(); d_cams.release();
    }
};

struct orbm_matcher {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;  // `stream` = the one in use (own or caller's)
    DevBuf<uint8_t> d_q, d_r, d_scratch, d_queries, d_occ;
    DevBuf<int32_t> d_i0, d_i1, d_i2, d_choice, d_claim, d_match, d_status, d_x0, d_x1, d_x2;
    DevBuf<int32_t> d_gclaim;  // claim tables of the resolve when they do not fit LDS (2 x features)
    DevBuf<int2> d_qmeta;      // {blocks, angle} of every query, written by k_project for the resolve
    DevBuf<orbm_window> d_win2; // second windows of a two-camera search
    DevBuf<uint16_t> d_u16;
    PinnedBuf<int32_t> h_i0, h_i1, h_i2, h_match;
    PinnedBuf<int32_t> h_gcnt;            // per-camera counts of a gathered multi-GPU exchange (+ own query count)
    DevBuf<int32_t> d_gstart;             // camera starts + {features, first query, queries} of the gathered list
    int gathered_cams = 0;                // cameras of the last orbm_cross_top2_gathered_enqueue
    PinnedBuf<int32_t> h_c0, h_c1, h_c2;  // cross top-2 results (own buffers: they coexist with a search's h_i0/h_i1)
    DevBuf<uint8_t> d_cscratch;           // cross top-2 slice partials
    hipStream_t side_stream = nullptr;    // orbf_step: cross top-2 runs here, next to project + resolve on `stream`
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_q = nullptr;
    PinnedBuf<uint16_t> h_u16;
    PinnedBuf<uint8_t> h_ring;  // 4 slots of {CamFeat[64], int cam_start[65]} for asynchronous H2D
    // Host-written staging of the host-array entry points (orbm_frame_create; the queries / occupied flags of a search): the
    // host writes the packed arrays once (HBM through the large BAR, or mapped pinned memory), ONE kernel scatters a frame's
    // arrays into its buffers -- instead of a pageable hipMemcpyAsync per array.
    morb::StageBuf stage_f, stage_q;
    hipEvent_t ev_stage_f = nullptr;   // the unpack kernel of the last orbm_frame_create has read stage_f
    bool stage_f_busy = false;
    unsigned ring_pos = 0;
    std::vector<FrameBufs*> pool;  // free list
    // device-visible pinned destinations the next orbm_frame_from_device mirrors its merged arrays into (orbf_step)
    orb_keypoint* mirror_kps = nullptr; uint8_t* mirror_desc = nullptr; float* mirror_ur = nullptr; float* mirror_depth = nullptr;
    float* mirror_unx = nullptr; float* mirror_uny = nullptr;
    orb_calibration calib = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // orbm_set_calibration: undistortion applied by device-built frames (k1 == 0: off)
    int frame_min_rows = 0;  // the next device-built frame gets at least this many descriptor rows (fixed export block size)
    int last_status[4] = {0, 0, 0, 0};  // {status, nmatches, sweeps, longest list} of the last device resolve
    bool host_resolve = false;     // MORB_HOST_RESOLVE=1: always use the host resolve (testing / fallback path)
    int resolve_seq = 0;           // sequence number of the last tagged resolve launch
    bool foreign_work = false;     // something other than a step's own search was put on the stream (orbf_step_end then waits for all of it)
};

struct orbm_frame {
    orbm_matcher* owner = nullptr;
    FrameBufs* b = nullptr;
    int n_total = 0, n_cams = 0;
    float minX = 0, minY = 0, maxX = 0, maxY = 0, invW = 0, invH = 0;
    bool device_built = false;
    bool counts_on_device = false;  // n_total is a capacity until orbf_step has synchronised
    int desc_rows = 0;              // descriptor rows the frame was created for (the count trailer sits behind them)
    // host copies used by the host resolve / orbm_frame_grid; filled at create for host-built frames, lazily otherwise
    mutable std::vector<int32_t> octave, cell_start, items;
    mutable std::vector<float> angle;
    mutable bool host_valid = false;
    std::vector<int32_t> cam_start;  // n_cams + 1
    FrameDev dev() const {
        FrameDev F;
        F.n_total = n_total; F.n_cams = n_cams; F.n_total_dev = counts_on_device ? b->d_ntotal.p : nullptr; F.un_x = b->d_x.p; F.un_y = b->d_y.p; F.uright = b->d_ur.p;
        F.octave = b->d_oct.p; F.desc = (const uint4*)b->d_desc.p; F.cell_start = b->d_cell_start.p; F.items = b->d_items.p;
        F.minX = minX; F.minY = minY; F.invW = invW; F.invH = invH;
        return F;
    }
};

static FrameBufs* take_bufs(orbm_matcher* m) {
    if (!m->pool.empty()) { FrameBufs* b = m->pool.back(); m->pool.pop_back(); return b; }
    return new FrameBufs();
}

static int reserve_frame(FrameBufs* b, int n, int n_cams) {
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
static int ensure_host_copies(const orbm_frame* f) {
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

extern "C" {

// The single-workgroup kernels (k_resolve, k_frame_build_small) use the opt-in dynamic LDS limit.  The attribute belongs to
// the (kernel, device) pair, so it is raised once per DEVICE, on that device, when the first handle is created there
// (std::call_once: handles are created from several threads).
static int raise_lds_limits(int device) {
    constexpr int MAX_DEV = 64;
    static std::once_flag once[MAX_DEV];
    static hipError_t result[MAX_DEV];
    if (device < 0 || device >= MAX_DEV) { morb::set_error("device %d out of range", device); return ORB_E_ARG; }
    std::call_once(once[device], [device] {
        const void* fns[] = {(const void*)k_resolve<true, false>, (const void*)k_resolve<false, false>, (const void*)k_resolve<true, true>,
                             (const void*)k_resolve<false, true>, (const void*)k_frame_build_small};
        hipError_t e = hipSuccess;
        for (const void* fn : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        result[device] = e;
    });
    if (result[device] != hipSuccess) { morb::set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(result[device])); return ORB_E_HIP; }
    return ORB_OK;
}

int orbm_create(int device, orbm_matcher** out) {
    MORB_ARG(out != nullptr);
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    if ((rc = raise_lds_limits(device))) return rc;
    orbm_matcher* m = new orbm_matcher();
    m->device = device;
    // Matching is the latency chain a caller waits for while extraction of later timesteps fills the rest of the chip:
    // its streams get the highest priority the device offers.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&m->own_stream, hipStreamNonBlocking, prio_greatest);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete m; return ORB_E_HIP; }
    m->stream = m->own_stream;
    if (hipStreamCreateWithPriority(&m->side_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_q, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_stage_f, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) {
        morb::set_error("side stream / events could not be created");
        orbm_destroy(m);
        return ORB_E_HIP;
    }
    const char* hr = getenv("MORB_HOST_RESOLVE");
    m->host_resolve = hr && atoi(hr) != 0;
    *out = m;
    return ORB_OK;
}

void orbm_destroy(orbm_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    if (m->side_stream) { (void)hipStreamSynchronize(m->side_stream); (void)hipStreamDestroy(m->side_stream); }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->ev_q) (void)hipEventDestroy(m->ev_q);
    m->h_c0.release(); m->h_c1.release(); m->h_c2.release(); m->d_cscratch.release(); m->h_gcnt.release(); m->d_gstart.release();
    m->d_q.release(); m->d_r.release(); m->d_scratch.release(); m->d_queries.release(); m->d_occ.release();
    m->d_i0.release(); m->d_i1.release(); m->d_i2.release(); m->d_choice.release(); m->d_claim.release(); m->d_qmeta.release(); m->d_win2.release();
    m->d_match.release(); m->d_status.release(); m->d_gclaim.release(); m->d_u16.release(); m->d_x0.release(); m->d_x1.release(); m->d_x2.release();
    m->h_i0.release(); m->h_i1.release(); m->h_i2.release(); m->h_match.release(); m->h_u16.release(); m->h_ring.release();
    m->stage_f.release(); m->stage_q.release();
    if (m->ev_stage_f) (void)hipEventDestroy(m->ev_stage_f);
    for (FrameBufs* b : m->pool) { b->release(); delete b; }
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

void* orbm_stream(const orbm_matcher* m) { return m ? (void*)m->stream : nullptr; }

int orbm_debug_last_resolve(const orbm_matcher* m, int* out4) {
    MORB_ARG(m && out4);
    for (int k = 0; k < 4; ++k) out4[k] = m->last_status[k];
    return ORB_OK;
}

int orbm_wait_for_stream(orbm_matcher* m, void* other_stream) {
    MORB_ARG(m != nullptr);
    MORB_HIP(hipSetDevice(m->device));
    // everything enqueued on `other_stream` so far happens before whatever this handle enqueues next (no host wait)
    MORB_HIP(hipEventRecord(m->ev_fork, (hipStream_t)other_stream));
    MORB_HIP(hipStreamWaitEvent(m->stream, m->ev_fork, 0));
    m->foreign_work = true;
    return ORB_OK;
}

int orbm_set_stream(orbm_matcher* m, void* stream) {
    MORB_ARG(m != nullptr);
    MORB_HIP(hipSetDevice(m->device));
    MORB_HIP(hipStreamSynchronize(m->stream));  // nothing of ours may still be in flight on the old stream
    m->stream = stream ? (hipStream_t)stream : m->own_stream;
    return ORB_OK;
}

int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8); memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

int orbm_set_calibration(orbm_matcher* m, const orb_calibration* calib) {
    MORB_ARG(m != nullptr);
    if (calib) m->calib = *calib; else memset(&m->calib, 0, sizeof(m->calib));
    return ORB_OK;
}

int orbm_undistort_points(const orb_calibration* calib, const float* x, const float* y, int n, float* ux, float* uy) {
    MORB_ARG(n >= 0 && (n == 0 || (x && y && ux && uy)));
    for (int i = 0; i < n; ++i) {
        if (!calib || calib->k1 == 0.0f) { ux[i] = x[i]; uy[i] = y[i]; }
        else morb_undistort_point(*calib, x[i], y[i], &ux[i], &uy[i]);
    }
    return ORB_OK;
}

int orbm_image_bounds(const orb_calibration* calib, int cols, int rows, float* out4) {
    MORB_ARG(out4 != nullptr);
    if (!calib || calib->k1 == 0.0f) { out4[0] = 0.f; out4[1] = 0.f; out4[2] = (float)cols; out4[3] = (float)rows; return ORB_OK; }
    const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; ++i) morb_undistort_point(*calib, cx[i], cy[i], &ux[i], &uy[i]);
    out4[0] = std::min(ux[0], ux[2]); out4[2] = std::max(ux[1], ux[3]);  // src/Frame.cc:768-771
    out4[1] = std::min(uy[0], uy[1]); out4[3] = std::max(uy[2], uy[3]);
    return ORB_OK;
}

int orbm_queries_from_motion(const orb_keypoint* kps, const uint8_t* desc, const float* depth, const int32_t* cam_of, int n,
                             float du, float dv, float th, const float* scale_factors, float mbf, orbm_query* out,
                             const float* un_x, const float* un_y) {
    MORB_ARG(n >= 0 && (n == 0 || (kps && desc && depth && cam_of && scale_factors && out)) && ((un_x == nullptr) == (un_y == nullptr)));
    for (int i = 0; i < n; ++i) {
        orbm_query& Q = out[i];
        const orb_keypoint& k = kps[i];
        const float u = (un_x ? un_x[i] : k.x) + du;
        Q.u = u; Q.v = (un_y ? un_y[i] : k.y) + dv;
        Q.radius = scale_factors[k.octave] * th;
        const float inv = depth[i] > 0 ? 1.0f / depth[i] : 0.0f;
        Q.ur = u - mbf * inv;
        Q.min_level = k.octave - 1; Q.max_level = k.octave + 1;
        Q.cam = cam_of[i]; Q.blocks = 1; Q.angle = k.angle;
        memcpy(Q.desc, desc + (size_t)i * 32, 32);
    }
    return ORB_OK;
}

int orbm_count_ratio_accepted(const int32_t* best_dist, const int32_t* second_dist, int n, int th_low, float ratio) {
    // SearchByBoW's acceptance (reference src/ORBmatcher.cc:324-327): best <= TH_LOW and best < ratio * second (float compare)
    if (n < 0 || (n > 0 && (!best_dist || !second_dist))) return ORB_E_ARG;
    int acc = 0;
    for (int i = 0; i < n; ++i)
        acc += (best_dist[i] <= th_low && (float)best_dist[i] < ratio * (float)second_dist[i]) ? 1 : 0;
    return acc;
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

int orbm_use_matrix_cores(int on) { return g_matrix_cores.exchange(on < 0 ? -1 : (on ? 1 : 0)); }

size_t orbm_top2_scratch_bytes(int nq, int nr) {
    if (nq <= 0 || nr <= 0) return 0;
    const int S = top2_plan(nq, nr).S;
    return S <= 1 ? 0 : (size_t)3 * S * nq * sizeof(int);
}

int orbm_hamming_top2_device(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_best_idx,
                             int32_t* d_best_dist, int32_t* d_second_dist, void* d_scratch, void* stream) {
    MORB_ARG(nq >= 0 && nr >= 0);
    if (nq == 0) return ORB_OK;
    MORB_ARG(d_q && d_best_idx && d_best_dist && d_second_dist && (nr == 0 || d_r));
    MORB_ARG((((uintptr_t)d_q | (uintptr_t)d_r) & 15) == 0);
    const Top2Plan plan = top2_plan(nq, std::max(nr, 1), d_scratch != nullptr);
    return launch_top2(d_q, nq, d_r, nr, d_best_idx, d_best_dist, d_second_dist, d_scratch, plan, (hipStream_t)stream);
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
    F->cam_start.assign(f->n_cams + 1, 0);
    for (int g = 0; g < n; g++) {
        const int px = (int)roundf((f->un_x[g] - F->minX) * F->invW);
        const int py = (int)roundf((f->un_y[g] - F->minY) * F->invH);
        const int cam = f->cam_of[g];
        if (cam >= 0 && cam < f->n_cams) F->cam_start[cam + 1]++;
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

}  // extern "C"

// d_counts != NULL: cams[c].n are CAPACITIES, the real per-camera counts sit in HBM (orbx_device_counts) and are read by
// the build kernel; the frame's n_total stays a capacity until the caller has synchronised and calls frame_set_counts.
// sink_filled: *out is a frame made by frame_prepare_sink whose per-feature arrays the describe kernel has filled.
static int frame_from_device_impl(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                                  float max_x, float max_y, const int* d_counts, orbm_frame** out, bool sink_filled = false);

extern "C" {

int orbm_frame_from_device(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                           float max_x, float max_y, orbm_frame** out) {
    MORB_ARG(out != nullptr);
    *out = nullptr;
    return frame_from_device_impl(m, cams, n_cams, mbf, min_x, min_y, max_x, max_y, nullptr, out);
}

}  // extern "C"

// (the single-workgroup frame kernel's opt-in dynamic LDS limit is raised per device in orbm_create: raise_lds_limits)
static int frame_build_lds_limit() { return ORB_OK; }

// Frame shell with storage for `n` features, no kernel launched yet.
static int frame_shell(orbm_matcher* m, int n, int n_cams, float min_x, float min_y, float max_x, float max_y, bool counts_on_device,
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
static int frame_prepare_sink(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
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

// the sink of an existing (persistent) frame
static int frame_sink_of(orbm_matcher* m, orbm_frame* F, const orbm_cam_features* cams, int n_cams, float mbf, FrameSink* sink) {
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

static int frame_from_device_impl(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
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
    if (!small) {
        MORB_HIP(hipMemcpyAsync(F->b->d_cams.p, hc, (size_t)n_cams * sizeof(CamFeat), hipMemcpyHostToDevice, st));
        MORB_HIP(hipMemcpyAsync(F->b->d_cam_start.p, hstart, (size_t)(n_cams + 1) * 4, hipMemcpyHostToDevice, st));
    }
    HostMirror hm{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, m->calib};
    if (m->mirror_kps) { hm.kps = m->mirror_kps; hm.desc = (uint4*)m->mirror_desc; }
    if (m->mirror_ur) { hm.ur = m->mirror_ur; hm.depth = m->mirror_depth; }
    if (m->mirror_unx) { hm.unx = m->mirror_unx; hm.uny = m->mirror_uny; }
    if (small) {
        if ((rc = frame_build_lds_limit())) { if (!*out) orbm_frame_destroy(F); return rc; }
        CamFeat4 c4;
        memset(&c4, 0, sizeof(c4));
        for (int c = 0; c < n_cams; ++c) c4.c[c] = hc[c];
        hipLaunchKernelGGL(k_frame_build_small, dim3(1), dim3(1024), lds_small, st, c4, F->b->d_cam_start.p, d_counts, F->b->d_ntotal.p,
                           n_cams, n, mbf,
                           F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p, F->b->d_depth.p,
                           F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_start.p,
                           F->b->d_items.p, hm, sink_filled ? (const int*)F->b->d_cell_of.p : nullptr, F->desc_rows);
    } else {
        const int* n_dev = nullptr;
        if (d_counts) {
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
        MORB_HIP(hipMemsetAsync(F->b->d_cursor.p, 0, (size_t)(ncell + 1) * 4, st));  // used as the per-cell counter first
        if (n) {
            hipLaunchKernelGGL(k_frame_fill, dim3((n + 255) / 256), dim3(256), 0, st, (const CamFeat*)F->b->d_cams.p, n_cams, n, mbf,
                               F->minX, F->minY, F->invW, F->invH, F->b->d_x.p, F->b->d_y.p, F->b->d_ur.p, F->b->d_depth.p,
                               F->b->d_oct.p, F->b->d_ang.p, F->b->d_kps.p, (uint4*)F->b->d_desc.p, F->b->d_cell_of.p,
                               F->b->d_cursor.p, hm, n_dev);
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
static void frame_set_counts(orbm_frame* F, const int* counts) {
    int base = 0;
    for (int c = 0; c < F->n_cams; ++c) { F->cam_start[c] = base; base += counts[c]; }
    F->cam_start[F->n_cams] = base;
    F->n_total = base;
    F->counts_on_device = false;
}

extern "C" {

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

// k_project into m->d_i0 (idx) / d_u16 (dist) / d_i1 (count); optionally copied to the pinned host mirrors
static int run_project(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int cap, int gate_right,
                       int with_dist, bool upload_queries, bool to_host, int transposed = 0,
                       const uint8_t* d_occupied = nullptr, int* d_topk = nullptr, int short_th = 256,
                       const float* d_inv_sigma2 = nullptr, const orbm_query* q_device_visible = nullptr, int2* d_qmeta = nullptr,
                       const orbm_window* d_win2 = nullptr) {
    int rc;
    if ((rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query))) || (rc = m->d_i0.reserve((size_t)nq * cap)) ||
        (rc = m->d_u16.reserve((size_t)nq * cap)) || (rc = m->d_i1.reserve(nq)))
        return rc;
    if (upload_queries)
        MORB_HIP(hipMemcpyAsync(m->d_queries.p, q, (size_t)nq * sizeof(orbm_query), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_project, dim3((nq + 3) / 4), dim3(256), 0, m->stream, f->dev(),
                       q_device_visible ? q_device_visible : (const orbm_query*)m->d_queries.p, nq, cap, gate_right, with_dist, transposed,
                       m->d_i0.p, m->d_u16.p, m->d_i1.p, d_occupied, d_topk, short_th, d_inv_sigma2, d_qmeta, d_win2);
    MORB_HIP(hipGetLastError());
    if (to_host) {
        if ((rc = m->h_i0.reserve((size_t)nq * cap)) || (rc = m->h_u16.reserve((size_t)nq * cap)) || (rc = m->h_i1.reserve(nq)))
            return rc;
        MORB_HIP(hipMemcpyAsync(m->h_i1.p, m->d_i1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
        MORB_HIP(hipMemcpyAsync(m->h_i0.p, m->d_i0.p, (size_t)nq * cap * 4, hipMemcpyDeviceToHost, m->stream));
        if (with_dist) MORB_HIP(hipMemcpyAsync(m->h_u16.p, m->d_u16.p, (size_t)nq * cap * 2, hipMemcpyDeviceToHost, m->stream));
        MORB_HIP(hipStreamSynchronize(m->stream));
    }
    return ORB_OK;
}

// Runs k_project with a growing per-query capacity until every list fits; results in m->h_i0 / h_u16 / h_i1.
static int project_all(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int gate_right, int with_dist,
                       int* cap_out, int cap0 = 64, const orbm_window* d_win2 = nullptr) {
    int cap = cap0;
    bool first = true;
    for (;;) {
        int rc = run_project(m, f, q, nq, cap, gate_right, with_dist, first, true, 0, nullptr, nullptr, 256, nullptr, nullptr, nullptr, d_win2);
        if (rc) return rc;
        first = false;
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
    int rc = run_project(m, f, q, nq, cap_per_query, 1, 1, true, true);
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

int orbm_project_best(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, const uint8_t* occupied, int gate,
                      const float* inv_level_sigma2, int n_levels, int32_t* best_idx, int32_t* best_dist) {
    MORB_ARG(m && f && nq >= 0 && gate >= 0 && gate <= 2);
    if (nq == 0) return ORB_OK;
    MORB_ARG(q && best_idx && best_dist);
    MORB_ARG(gate != ORBM_GATE_CHI2 || (inv_level_sigma2 && n_levels > 0 && n_levels <= 64));
    MORB_HIP(hipSetDevice(m->device));
    const int n = f->n_total;
    if (n == 0) { for (int i = 0; i < nq; ++i) { best_idx[i] = -1; best_dist[i] = 256; } return ORB_OK; }
    int rc;
    if ((rc = m->d_claim.reserve((size_t)(2 * RESOLVE_K + 1) * nq)) || (rc = m->d_occ.reserve(std::max(n, 16) + 512)) ||
        (rc = m->h_i2.reserve((size_t)2 * nq)))
        return rc;
    if (occupied) MORB_HIP(hipMemcpyAsync(m->d_occ.p, occupied, (size_t)n, hipMemcpyHostToDevice, m->stream));
    float* d_sig = nullptr;
    if (gate == ORBM_GATE_CHI2) {   // the level table rides behind the occupied bytes
        if (ensure_host_copies(f)) return ORB_E_HIP;
        for (int g = 0; g < n; ++g) MORB_ARG(f->octave[g] >= 0 && f->octave[g] < n_levels);
        d_sig = (float*)(m->d_occ.p + ((std::max(n, 16) + 15) & ~15));
        MORB_HIP(hipMemcpyAsync(d_sig, inv_level_sigma2, (size_t)n_levels * sizeof(float), hipMemcpyHostToDevice, m->stream));
    }
    // the sorted shortlist k_project keeps per query (distance << 16 | visiting position) starts with exactly the candidate
    // the reference's `if (dist < bestDist)` loop ends on: smallest distance, first in visiting order
    if ((rc = run_project(m, f, q, nq, /*cap=*/64, gate, 1, true, false, /*transposed=*/1, occupied ? m->d_occ.p : nullptr, m->d_claim.p, 256, d_sig)))
        return rc;
    MORB_HIP(hipMemcpyAsync(m->h_i2.p, m->d_claim.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipMemcpyAsync(m->h_i2.p + nq, m->d_claim.p + (size_t)RESOLVE_K * nq, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
    for (int i = 0; i < nq; ++i) {
        const int key = m->h_i2.p[i], g = m->h_i2.p[nq + i];
        best_idx[i] = g; best_dist[i] = g >= 0 ? (key >> 16) : 256;
    }
    return ORB_OK;
}

// inspection / bench (roofline M3): the projection kernel alone, in the configuration the frame search launches it in
// (gates on, distances, transposed lists, shortlist extraction), timed with HIP events on the handle's stream
int orbm_debug_time_project(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int th_high, int iters, float* avg_us,
                            long long* n_gated) {
    MORB_ARG(m && f && q && nq > 0 && iters > 0 && avg_us && n_gated && f->n_total > 0);
    MORB_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = m->d_claim.reserve((size_t)(2 * RESOLVE_K + 1) * nq)) || (rc = m->d_qmeta.reserve(nq)) || (rc = m->h_i1.reserve(nq)))
        return rc;
    if ((rc = run_project(m, f, q, nq, 64, 1, 1, true, false, 1, nullptr, m->d_claim.p, th_high, nullptr, nullptr, m->d_qmeta.p))) return rc;
    hipEvent_t e0, e1;
    MORB_HIP(hipEventCreate(&e0)); MORB_HIP(hipEventCreate(&e1));
    MORB_HIP(hipEventRecord(e0, m->stream));
    for (int it = 0; it < iters && !rc; ++it)
        rc = run_project(m, f, q, nq, 64, 1, 1, false, false, 1, nullptr, m->d_claim.p, th_high, nullptr, nullptr, m->d_qmeta.p);
    hipError_t he = hipEventRecord(e1, m->stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    MORB_HIP(he);
    MORB_HIP(hipMemcpy(m->h_i1.p, m->d_i1.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    long long tot = 0;
    for (int i = 0; i < nq; ++i) tot += m->h_i1.p[i];
    *avg_us = ms * 1e3f / (float)iters; *n_gated = tot;
    return ORB_OK;
}

// Sequential resolve on the host from the ordered candidate lists (fallback of the device resolve; same semantics).
static int host_resolve(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq, const uint8_t* occupied,
                        bool points, float nnratio, int th_high, int check_orientation, int cap0, int32_t* match_of_feature,
                        int* nmatches, const orbm_window* d_win2 = nullptr) {
    int rc = ensure_host_copies(cur);
    if (rc) return rc;
    int cap = 0;
    if ((rc = project_all(m, cur, q, nq, 1, 1, &cap, cap0, d_win2))) return rc;
    for (int g = 0; g < cur->n_total; g++) match_of_feature[g] = -1;
    std::vector<int32_t> rot[ORBM_HISTO_LENGTH];
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        const int cnt = m->h_i1.p[i];
        const int32_t* ci = m->h_i0.p + (size_t)i * cap;
        const uint16_t* cd = m->h_u16.p + (size_t)i * cap;
        int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1;
        for (int k = 0; k < cnt; k++) {
            const int g = ci[k];
            const int owner = match_of_feature[g];
            if (owner >= 0 ? q[owner].blocks != 0 : (occupied && occupied[g])) continue;
            const int d = cd[k];
            if (points) {
                if (d < best) { best2 = best; best = d; lvl2 = lvl; lvl = cur->octave[g]; bidx = g; }
                else if (d < best2) { lvl2 = cur->octave[g]; best2 = d; }
            } else if (d < best) { best = d; bidx = g; }
        }
        if (best <= th_high && bidx >= 0) {
            if (points && lvl == lvl2 && (float)best > nnratio * (float)best2) continue;
            match_of_feature[bidx] = i;
            nm++;
            if (!points && check_orientation) {
                float rotv = q[i].angle - cur->angle[bidx];
                if (rotv < 0.0) rotv += 360.0f;
                int bin = (int)roundf(rotv * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < ORBM_HISTO_LENGTH) rot[bin].push_back(bidx);
            }
        }
    }
    if (!points && check_orientation) {
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

// k_project + k_resolve on the device, one D2H of {status, matches}; falls back to host_resolve when the sweep limit is
// hit, retries with a larger capacity when a candidate list overflowed.  Split in two so that a caller can enqueue
// other work on the stream between the launch and the one synchronisation (orbf_step).
struct SearchJob {
    const orbm_frame* cur; const orbm_query* q; int nq; const uint8_t* occupied;
    bool points; float nnratio; int th_high, check_ori;
    int cap; bool device_path;
    bool pollable = false;              // single-workgroup resolve in flight with tagged result words (see k_resolve)
    bool want_tags = false; int seq = 0; // caller wants to watch the results arrive; sequence number of the launch in flight
    const orbm_query* q_dev = nullptr;  // device-visible alias of `q` when it lives in mapped pinned memory: read in place, no H2D
    const uint8_t* occ_dev = nullptr;   // device-visible copy of `occupied` (staged by the caller): no H2D either
    const orbm_window* win2_dev = nullptr;  // second windows of the queries (device memory), or NULL
};

static int search_enqueue(orbm_matcher* m, SearchJob& J, bool queries_already_on_device = false) {
    const int n = J.cur->n_total;
    J.device_path = false; J.pollable = false;
    if (J.nq == 0 || n == 0) return ORB_OK;
    // two claim tables (one int per feature each) + the candidate counts (u16 per query, padded); tables that do not fit LDS go
    // to an HBM workspace (GCL variant of the kernel)
    const size_t lds = (size_t)2 * n * sizeof(int) + (size_t)((J.nq + 1) / 2) * sizeof(int);
    const bool multi = lds > 150 * 1024;  // multi-workgroup resolve with the tables in HBM
    if (m->host_resolve || J.nq > RESOLVE_MAX_Q) return ORB_OK;  // finish() takes the host path
    if (multi) { int rcg = m->d_gclaim.reserve((size_t)2 * n + RS_STATE_INTS); if (rcg) return rcg; }
    // (tables beyond 48 KB use the opt-in dynamic LDS limit, raised per device in orbm_create)
    int rc;
    if ((rc = m->d_choice.reserve(J.nq)) || (rc = m->d_claim.reserve((size_t)(2 * RESOLVE_K + 1) * J.nq)) || (rc = m->d_match.reserve(n)) ||
        (rc = m->d_status.reserve(4)) || (rc = m->h_match.reserve((size_t)n + 4)) || (rc = m->d_occ.reserve(std::max(n, 16))))
        return rc;
    if (J.occupied && !J.occ_dev) MORB_HIP(hipMemcpyAsync(m->d_occ.p, J.occupied, (size_t)n, hipMemcpyHostToDevice, m->stream));
    const uint8_t* d_occ = J.occ_dev ? J.occ_dev : (J.occupied ? m->d_occ.p : nullptr);
    const orbm_frame* cur = J.cur;
    const int nq = J.nq, cap = J.cap, th_high = J.th_high;
    const float nnratio = J.nnratio;
    if ((rc = m->d_qmeta.reserve(nq))) return rc;
    // queries in mapped pinned memory are read in place by k_project (one 68-byte record per wave); only the multi-workgroup
    // resolve, whose kernels read the records themselves, still wants them in HBM
    const orbm_query* q_in_place = (J.q_dev && !multi) ? J.q_dev : nullptr;
    if (J.q_dev && multi) {
        if ((rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query)))) return rc;
        MORB_HIP(hipMemcpyAsync(m->d_queries.p, J.q_dev, (size_t)nq * sizeof(orbm_query), hipMemcpyDefault, m->stream));
    }
    if ((rc = run_project(m, cur, J.q, nq, cap, 1, 1, !queries_already_on_device && !J.q_dev, false, /*transposed=*/1, d_occ, m->d_claim.p,
                          J.points ? 256 : th_high, nullptr, q_in_place, m->d_qmeta.p, J.win2_dev)))
        return rc;
    if (multi) {
        int* tab0 = m->d_gclaim.p; int* tab1 = tab0 + n; int* state = tab1 + n;
        MORB_HIP(hipMemsetAsync(state, 0, RS_STATE_INTS * sizeof(int), m->stream));
        const int nb_all = (std::max(n, nq) + 255) / 256, nb_q = (nq + 255) / 256, nb_f = (n + 255) / 256;
        hipLaunchKernelGGL(k_rs_init, dim3(nb_all), dim3(256), 0, m->stream, n, nq, tab0, tab1, m->d_match.p, m->d_choice.p,
                           (const int*)m->d_i1.p, state);
        for (int it = 0; it < RS_MAX_SWEEPS; ++it) {
            const int* rd = (it & 1) ? tab1 : tab0;
            int* wr = (it & 1) ? tab0 : tab1;
            if (J.points)
                hipLaunchKernelGGL(k_rs_sweep<true>, dim3(nb_q), dim3(256), 0, m->stream, cur->dev(), (const orbm_query*)m->d_queries.p, nq,
                                   cap, it, (const int*)m->d_i0.p, (const uint16_t*)m->d_u16.p, (const int*)m->d_i1.p, d_occ, th_high,
                                   nnratio, m->d_choice.p, (const int*)m->d_claim.p, rd, wr, state);
            else
                hipLaunchKernelGGL(k_rs_sweep<false>, dim3(nb_q), dim3(256), 0, m->stream, cur->dev(), (const orbm_query*)m->d_queries.p, nq,
                                   cap, it, (const int*)m->d_i0.p, (const uint16_t*)m->d_u16.p, (const int*)m->d_i1.p, d_occ, th_high,
                                   nnratio, m->d_choice.p, (const int*)m->d_claim.p, rd, wr, state);
        }
        const int ori = J.points ? 0 : J.check_ori;
        hipLaunchKernelGGL(k_rs_owner, dim3(nb_q), dim3(256), 0, m->stream, (const orbm_query*)m->d_queries.p, nq, cap,
                           (const int*)m->d_choice.p, (const float*)cur->b->d_ang.p, ori, m->d_match.p, state);
        if (ori)
            hipLaunchKernelGGL(k_rs_reject, dim3(nb_q), dim3(256), 0, m->stream, (const orbm_query*)m->d_queries.p, nq, cap,
                               (const int*)m->d_choice.p, (const float*)cur->b->d_ang.p, m->d_match.p, state);
        hipLaunchKernelGGL(k_rs_write, dim3(nb_f), dim3(256), 0, m->stream, n, cur->dev().n_total_dev, cap, (const int*)m->d_match.p,
                           (const int*)state, m->h_match.dp + 4, m->h_match.dp);
        MORB_HIP(hipGetLastError());
        J.device_path = true;
        return ORB_OK;
    }
    // claim table + (when it fits) the per-query sweep state
    const size_t lds_q = lds + (size_t)nq * (sizeof(int) + sizeof(float) + RESOLVE_K * sizeof(int) + 1) + (size_t)n * sizeof(float) + 16;
    const bool ldsq = lds_q <= 150 * 1024 && n < 65535;
    const size_t lds_use = ldsq ? lds_q : lds;
#define MORB_RESOLVE_LAUNCH(PT, LQ)                                                                                      \
    hipLaunchKernelGGL((k_resolve<PT, LQ>), dim3(1), dim3(1024), lds_use, m->stream, cur->dev(), (const int2*)m->d_qmeta.p, \
                       nq, cap, (const int*)m->d_i0.p, (const uint16_t*)m->d_u16.p, (const int*)m->d_i1.p, d_occ,            \
                       (const float*)cur->b->d_ang.p, th_high, nnratio, J.points ? 0 : J.check_ori, 256, m->d_choice.p,      \
                       (const int*)m->d_claim.p, m->h_match.dp + 4, m->h_match.dp, J.seq << 20)
    J.seq = 0;
    if (J.want_tags) { m->resolve_seq = m->resolve_seq % 2047 + 1; J.seq = m->resolve_seq; }   // 1..2047, never 0
    if (ldsq) { if (J.points) MORB_RESOLVE_LAUNCH(true, true); else MORB_RESOLVE_LAUNCH(false, true); }
    else { if (J.points) MORB_RESOLVE_LAUNCH(true, false); else MORB_RESOLVE_LAUNCH(false, false); }
#undef MORB_RESOLVE_LAUNCH
    MORB_HIP(hipGetLastError());  // status + matches are written by the kernel into the mapped pinned buffer
    J.device_path = true;
    J.pollable = J.seq != 0;
    return ORB_OK;
}

// After the stream has been synchronised.  match_of_feature may alias m->h_match.p + 4 (then nothing is copied).
static int search_finish(orbm_matcher* m, SearchJob& J, int32_t* match_of_feature, int* nmatches) {
    const int n = J.cur->n_total;
    *nmatches = 0;
    if (J.nq == 0 || n == 0) { for (int g = 0; g < n; g++) match_of_feature[g] = -1; return ORB_OK; }
    if (!J.device_path) {
        m->last_status[0] = -1; m->last_status[1] = 0; m->last_status[2] = 0; m->last_status[3] = 0;  // (host path)
    }
    if (!J.device_path)
        return host_resolve(m, J.cur, J.q, J.nq, J.occupied, J.points, J.nnratio, J.th_high, J.check_ori, 64, match_of_feature, nmatches, J.win2_dev);
    // Result words of a tagged launch are taken as they arrive (the caller may not have synchronised the stream): wait for
    // the word to carry this launch's sequence number, then strip it.  After ~10 ms without progress the stream is
    // synchronised for good (which also covers a launch that failed).
    bool synced = false;
    auto word = [&](int idx, int bias) -> int {
        volatile int32_t* p = m->h_match.p + idx;
        if (!J.seq) return *p;
        for (int spin = 0;; ++spin) {
            const int w = *p;
            if ((w >> 20) == J.seq) return (w & 0xfffff) - bias;
            if (spin > 100000 && !synced) { (void)hipStreamSynchronize(m->stream); synced = true; spin = 0; }
            else if (spin > 100000) return -3;   // cannot happen after a synchronisation; reported below
            __builtin_ia32_pause();
        }
    };
    for (;;) {
        const int status = word(0, 0);
        m->last_status[0] = status;
        for (int k = 1; k < 4; ++k) m->last_status[k] = word(k, 0);
        if (status == -3) { morb::set_error("resolve results never arrived"); return ORB_E_HIP; }
        if (status == 0) break;
        if (status == 2) {  // a candidate list overflowed: retry with room for the longest one
            J.cap = (m->last_status[3] + 63) & ~63;
            int rc = search_enqueue(m, J, /*queries_already_on_device=*/true);
            if (rc) return rc;
            MORB_HIP(hipStreamSynchronize(m->stream));
            synced = true;
            continue;
        }
        // not converged within the sweep limit: exact host fallback
        return host_resolve(m, J.cur, J.q, J.nq, J.occupied, J.points, J.nnratio, J.th_high, J.check_ori, J.cap, match_of_feature, nmatches, J.win2_dev);
    }
    if (J.seq) {
        for (int g = 0; g < n; ++g) {
            const int v = word(4 + g, 2);
            if (v == -3) { morb::set_error("resolve results never arrived"); return ORB_E_HIP; }
            match_of_feature[g] = v;
        }
    } else if (match_of_feature != m->h_match.p + 4) memcpy(match_of_feature, m->h_match.p + 4, (size_t)n * 4);
    *nmatches = m->last_status[1];
    return ORB_OK;
}

static int search_common(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq, const uint8_t* occupied,
                         bool points, float nnratio, int th_high, int check_orientation, int32_t* match_of_feature,
                         int* nmatches, const orbm_window* second = nullptr) {
    SearchJob J{cur, q, nq, occupied, points, nnratio, th_high, check_orientation, 64, false};
    int rc;
    if (second && nq > 0) {   // second windows (two-camera loop search): a device copy for the projection kernel
        if ((rc = m->d_win2.reserve(nq))) return rc;
        MORB_HIP(hipMemcpyAsync(m->d_win2.p, second, (size_t)nq * sizeof(orbm_window), hipMemcpyHostToDevice, m->stream));
        MORB_HIP(hipStreamSynchronize(m->stream));   // (`second` is the caller's)
        J.win2_dev = m->d_win2.p;
    }
    // The queries and the occupied flags go through host-written staging (HBM behind the large BAR, or mapped pinned memory)
    // and are read in place by the kernels: no pageable hipMemcpyAsync on the call's critical path.  The staging stays
    // untouched until this call has synchronised.
    if (nq > 0 && cur->n_total > 0 && !m->host_resolve && nq <= RESOLVE_MAX_Q) {
        const size_t qbytes = ((size_t)nq * sizeof(orbm_query) + 255) & ~(size_t)255, obytes = occupied ? (size_t)cur->n_total : 0;
        if ((rc = m->stage_q.reserve(qbytes + obytes + 16))) return rc;
        memcpy(m->stage_q.p, q, (size_t)nq * sizeof(orbm_query));
        if (occupied) memcpy(m->stage_q.p + qbytes, occupied, obytes);
        m->stage_q.publish();
        J.q_dev = reinterpret_cast<const orbm_query*>(m->stage_q.dp);
        J.occ_dev = occupied ? m->stage_q.dp + qbytes : nullptr;
    }
    rc = search_enqueue(m, J);
    if (rc) return rc;
    if (J.device_path) MORB_HIP(hipStreamSynchronize(m->stream));
    return search_finish(m, J, match_of_feature, nmatches);
}

int orbm_search_by_projection(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                              const uint8_t* occupied, int th_high, int check_orientation, int32_t* match_of_feature,
                              int* nmatches) {
    MORB_ARG(m && cur && nq >= 0 && nmatches && (cur->n_total == 0 || match_of_feature) && (nq == 0 || q));
    MORB_HIP(hipSetDevice(m->device));
    return search_common(m, cur, q, nq, occupied, false, 0.f, th_high, check_orientation, match_of_feature, nmatches);
}

int orbm_search_by_projection_windows(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, const orbm_window* second, int nq,
                                      const uint8_t* occupied, int th_high, int check_orientation, int32_t* match_of_feature,
                                      int* nmatches) {
    MORB_ARG(m && cur && nq >= 0 && nmatches && (cur->n_total == 0 || match_of_feature) && (nq == 0 || (q && second)));
    MORB_HIP(hipSetDevice(m->device));
    return search_common(m, cur, q, nq, occupied, false, 0.f, th_high, check_orientation, match_of_feature, nmatches, second);
}

int orbm_search_by_projection_points(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                                     const uint8_t* occupied, float nnratio, int th_high, int32_t* match_of_feature,
                                     int* nmatches) {
    MORB_ARG(m && cur && nq >= 0 && nmatches && (cur->n_total == 0 || match_of_feature) && (nq == 0 || q));
    MORB_HIP(hipSetDevice(m->device));
    std::vector<orbm_query> q0(q, q + nq);
    for (auto& Q : q0) Q.cam = 0;  // camera-1 grid only (reference src/ORBmatcher.cc:88-89, src/Frame.cc:510-563)
    return search_common(m, cur, q0.data(), nq, occupied, true, nnratio, th_high, 0, match_of_feature, nmatches);
}

// k_cross_top2 (+ merge) over `n` features in `d_desc` split into cameras by `d_cam_start`; queries [q_off, q_off+nq).
// Results land in the pinned mirrors m->h_c0/h_c1/h_c2 once stream `st` has been synchronised.
// Destination of a cross top-2: three mapped pinned result arrays + the HBM scratch of the slice partials.  The matcher
// owns one (h_c0..2 / d_cscratch); the front end owns one per result set, because it runs the cross matching of steps
// that were announced ahead at the end of their extraction chains.
struct CrossOut {
    PinnedBuf<int32_t> i, b, s;
    DevBuf<uint8_t> scratch;
    int reserve(int nq, int n) {
        const int S = std::max(top2_plan(nq, n).S, top2_slices(nq, n));   // (room for either form of the kernel)
        int rc;
        if ((rc = scratch.reserve(std::max<size_t>((size_t)3 * S * nq * 4, 16))) || (rc = i.reserve(nq)) || (rc = b.reserve(nq)) ||
            (rc = s.reserve(nq)))
            return rc;
        return ORB_OK;
    }
    void release() { i.release(); b.release(); s.release(); scratch.release(); }
};

static int cross_enqueue_to(hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off, int nq,
                            const int* d_n, int* o_idx, int* o_best, int* o_second, void* scratch) {
    if (nq == 0) return ORB_OK;
    const int qblocks = (nq + 63) / 64;
    const Top2Plan plan = top2_plan(nq, n);   // (nq, n may be capacities: the kernels take the counts from d_n then)
    const int S = plan.S;
    if (plan.mfma) {   // matrix-core form (default from one tile of work on; orbm_use_matrix_cores(0) / MORB_TOP2_MFMA=0: popcount form)
        int* p = (int*)scratch;
        int *p_idx = S > 1 ? p : o_idx, *p_best = S > 1 ? p + (size_t)S * nq : o_best, *p_second = S > 1 ? p + 2 * (size_t)S * nq : o_second;
        hipLaunchKernelGGL(k_cross_top2_mfma, dim3(S, (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK), dim3(64 * MM_WAVES), 0, st,
                           (const uint32_t*)d_desc, n, d_cam_start, n_cams, q_off, nq, plan.slice_len, p_idx, p_best, p_second, d_n);
        if (S > 1)
            hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, p_idx, p_best, p_second, S, nq, o_idx, o_best,
                               o_second, d_n);
        MORB_HIP(hipGetLastError());
        return ORB_OK;
    }
    if (S <= 1) {  // final results go straight to the mapped pinned mirrors
        hipLaunchKernelGGL(k_cross_top2, dim3(qblocks, 1), dim3(64 * TOP2_WAVES), 0, st, (const uint4*)d_desc, n, d_cam_start,
                           n_cams, q_off, nq, o_idx, o_best, o_second, d_n);
    } else {
        int* p = (int*)scratch;
        int *p_idx = p, *p_best = p + (size_t)S * nq, *p_second = p + 2 * (size_t)S * nq;
        hipLaunchKernelGGL(k_cross_top2, dim3(qblocks, S), dim3(64 * TOP2_WAVES), 0, st, (const uint4*)d_desc, n, d_cam_start,
                           n_cams, q_off, nq, p_idx, p_best, p_second, d_n);
        hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, p_idx, p_best, p_second, S, nq, o_idx, o_best,
                           o_second, d_n);
    }
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

static int cross_enqueue(orbm_matcher* m, hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off,
                         int nq, const int* d_n = nullptr) {
    if (nq == 0) return ORB_OK;
    const int S = std::max(top2_plan(nq, n).S, top2_slices(nq, n));   // (room for either form: the choice can change at run time)
    int rc;
    if ((rc = m->d_cscratch.reserve(std::max<size_t>((size_t)3 * S * nq * 4, 16))) || (rc = m->h_c0.reserve(nq)) ||
        (rc = m->h_c1.reserve(nq)) || (rc = m->h_c2.reserve(nq)))
        return rc;
    return cross_enqueue_to(st, d_desc, n, d_cam_start, n_cams, q_off, nq, d_n, m->h_c0.dp, m->h_c1.dp, m->h_c2.dp, m->d_cscratch.p);
}

static int cross_launch(orbm_matcher* m, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off, int nq,
                        int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
    if (nq == 0) return ORB_OK;
    int rc = cross_enqueue(m, m->stream, d_desc, n, d_cam_start, n_cams, q_off, nq);
    if (rc) return rc;
    MORB_HIP(hipStreamSynchronize(m->stream));
    memcpy(best_idx, m->h_c0.p, (size_t)nq * 4); memcpy(best_dist, m->h_c1.p, (size_t)nq * 4);
    memcpy(second_dist, m->h_c2.p, (size_t)nq * 4);
    return ORB_OK;
}

int orbm_cross_top2(orbm_matcher* m, const orbm_frame* f, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
    MORB_ARG(m && f);
    MORB_HIP(hipSetDevice(m->device));
    const int n = f->n_total;
    if (n == 0) return ORB_OK;
    MORB_ARG(best_idx && best_dist && second_dist);
    return cross_launch(m, f->b->d_desc.p, n, f->b->d_cam_start.p, f->n_cams, 0, n, best_idx, best_dist, second_dist);
}

int orbm_cross_top2_blocks(orbm_matcher* m, const uint8_t* const* d_desc_blocks, const int* counts, int n_blocks,
                           int first_query_block, int n_query_blocks, int32_t* best_idx, int32_t* best_dist,
                           int32_t* second_dist) {
    MORB_ARG(m && d_desc_blocks && counts && n_blocks >= 1 && n_blocks <= 512 && first_query_block >= 0 &&
             n_query_blocks >= 0 && first_query_block + n_query_blocks <= n_blocks);
    MORB_HIP(hipSetDevice(m->device));
    std::vector<int> start(n_blocks + 1, 0);
    for (int b = 0; b < n_blocks; ++b) { MORB_ARG(counts[b] >= 0); start[b + 1] = start[b] + counts[b]; }
    const int n = start[n_blocks];
    const int q_off = start[first_query_block], nq = start[first_query_block + n_query_blocks] - q_off;
    if (nq == 0) return ORB_OK;
    MORB_ARG(best_idx && best_dist && second_dist);
    int rc;
    if ((rc = m->d_r.reserve((size_t)n * 32)) || (rc = m->d_choice.reserve(n_blocks + 1))) return rc;
    for (int b = 0; b < n_blocks; ++b)
        if (counts[b]) {
            MORB_ARG(d_desc_blocks[b] != nullptr);
            MORB_HIP(hipMemcpyAsync(m->d_r.p + (size_t)start[b] * 32, d_desc_blocks[b], (size_t)counts[b] * 32,
                                    hipMemcpyDeviceToDevice, m->stream));
        }
    MORB_HIP(hipMemcpyAsync(m->d_choice.p, start.data(), (size_t)(n_blocks + 1) * 4, hipMemcpyHostToDevice, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));  // `start` is a local
    return cross_launch(m, m->d_r.p, n, m->d_choice.p, n_blocks, q_off, nq, best_idx, best_dist, second_dist);
}

// enqueue half: repack + top-2 on the handle's SIDE stream (next to whatever the main stream is doing), joined into the
// main stream so that the next synchronisation of the main stream covers it
int orbm_cross_top2_gathered_enqueue(orbm_matcher* m, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows,
                                     int cams_per_rank, int rank, void* after_stream, int wait_after) {
    MORB_ARG(m && d_gathered && world >= 1 && cams_per_rank >= 1 && world * cams_per_rank <= 512 && rank >= 0 && rank < world &&
             cap_rows >= 1 && block_bytes >= (size_t)cap_rows * 32 + (size_t)cams_per_rank * 4 && (block_bytes & 15) == 0);
    MORB_ARG(((uintptr_t)d_gathered & 15) == 0);
    MORB_HIP(hipSetDevice(m->device));
    const int n_cams = world * cams_per_rank;
    const int n_cap = world * cap_rows;  // capacity of the contiguous list
    int rc;
    if ((rc = m->d_r.reserve((size_t)n_cap * 32)) || (rc = m->d_gstart.reserve(n_cams + 1 + 4)) || (rc = m->h_gcnt.reserve(n_cams + 2)))
        return rc;
    hipStream_t sd = m->side_stream;
    if (wait_after) {  // the gathered buffer is produced on another stream (the collective's; NULL = the default stream)
        MORB_HIP(hipEventRecord(m->ev_fork, (hipStream_t)after_stream));
        MORB_HIP(hipStreamWaitEvent(sd, m->ev_fork, 0));
    }
    int* d_cam_start = m->d_gstart.p;
    int* d_range = m->d_gstart.p + n_cams + 1;
    hipLaunchKernelGGL(k_repack_gathered, dim3((2 * cap_rows + 255) / 256, world), dim3(256), 0, sd, d_gathered, world, block_bytes, cap_rows,
                       cams_per_rank, rank, (uint4*)m->d_r.p, d_cam_start, d_range, m->h_gcnt.dp);
    MORB_HIP(hipGetLastError());
    // the launch is sized for the capacity (cap_rows queries against world * cap_rows features); the counts come from HBM
    if ((rc = cross_enqueue(m, sd, m->d_r.p, n_cap, d_cam_start, n_cams, 0, cap_rows, d_range))) return rc;
    MORB_HIP(hipEventRecord(m->ev_join, sd));
    MORB_HIP(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    m->gathered_cams = n_cams;
    m->foreign_work = true;
    return ORB_OK;
}

// collect half, after the main stream has been synchronised (orbf_step_end does)
int orbm_cross_top2_gathered_collect(orbm_matcher* m, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                                     int32_t* counts_out, int* nq_out) {
    MORB_ARG(m && nq_out && m->gathered_cams > 0);
    const int n_cams = m->gathered_cams;
    const int nq = m->h_gcnt.p[n_cams];
    *nq_out = nq;
    if (counts_out) memcpy(counts_out, m->h_gcnt.p, (size_t)n_cams * 4);
    if (m->h_gcnt.p[n_cams + 1] != 0) {
        morb::set_error("gathered export blocks are inconsistent: %d per-camera counts in the trailers were negative or exceeded "
                        "their block's capacity (mismatched cap_rows / cams_per_rank between ranks, or a corrupt block)", m->h_gcnt.p[n_cams + 1]);
        return ORB_E_ARG;
    }
    if (nq && (best_idx || best_dist || second_dist)) {
        MORB_ARG(best_idx && best_dist && second_dist);
        memcpy(best_idx, m->h_c0.p, (size_t)nq * 4); memcpy(best_dist, m->h_c1.p, (size_t)nq * 4);
        memcpy(second_dist, m->h_c2.p, (size_t)nq * 4);
    }
    return ORB_OK;
}

int orbm_cross_top2_gathered_views(orbm_matcher* m, const int32_t** best_idx, const int32_t** best_dist, const int32_t** second_dist) {
    MORB_ARG(m && best_idx && best_dist && second_dist);
    *best_idx = m->h_c0.p; *best_dist = m->h_c1.p; *second_dist = m->h_c2.p;
    return ORB_OK;
}

int orbm_cross_top2_gathered(orbm_matcher* m, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows,
                             int cams_per_rank, int rank, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                             int32_t* counts_out, int* nq_out) {
    MORB_ARG(nq_out != nullptr);
    // (the main stream may have been ordered behind the collective by orbm_wait_for_stream: the side stream inherits that)
    int rc;
    MORB_HIP(hipSetDevice(m ? m->device : 0));
    if (m) { MORB_HIP(hipEventRecord(m->ev_q, m->stream)); MORB_HIP(hipStreamWaitEvent(m->side_stream, m->ev_q, 0)); }
    if ((rc = orbm_cross_top2_gathered_enqueue(m, d_gathered, world, block_bytes, cap_rows, cams_per_rank, rank, nullptr, 0))) return rc;
    MORB_HIP(hipStreamSynchronize(m->stream));
    return orbm_cross_top2_gathered_collect(m, best_idx, best_dist, second_dist, counts_out, nq_out);
}


// ================================================================================================ orbf (include/orbf.h)
}  // extern "C"

// ---- native multi-GPU exchange: RCCL's C API resolved at run time (the copy torch.distributed already loaded, else the
// ROCm one), one communicator per front end, the all-gather issued on the matcher's side stream from inside the step
#include <dlfcn.h>
#include <condition_variable>
#include <map>
namespace {
struct XUniqueId { char internal[128]; };   // ncclUniqueId (rccl.h:43)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(XUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, XUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};
RcclApi& rccl() {
    static RcclApi api;
    if (!api.lib) {
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (api.lib) break;
        }
        if (!api.lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (api.lib) {
            api.GetUniqueId = (int (*)(XUniqueId*))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void**, int, XUniqueId, int))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
            api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
            api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        }
    }
    return api;
}
int rccl_fail(const char* what, int r) {
    RcclApi& R = rccl();
    morb::set_error("%s failed: %s", what, R.GetErrorString ? R.GetErrorString(r) : "RCCL error");
    return ORB_E_HIP;
}

// ---- loopback transport: the same exchange between front ends of ONE process on ONE device (one host thread per "rank").
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so a 1-GPU machine could otherwise never run the world > 1
// path of orbf_step.  The all-gather becomes: every rank announces its block and an event behind the work that produced it,
// all ranks rendezvous on the host, then every rank copies every block into its own receive buffer on ITS stream, behind the
// producers' events.  Same contract as the collective (every rank calls once per step, same block size); everything
// downstream -- k_repack_gathered, the rig-wide top-2, the early / late placement of the exchange inside a step -- is the
// product code unchanged.
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0, members = 0;
    unsigned long generation = 0;
    bool broken = false;
    std::vector<const void*> send;
    std::vector<hipEvent_t> ev;
};
struct LoopComm { LoopGroup* g; int rank; };
std::mutex g_loop_mu;
std::map<int, LoopGroup*> g_loop_groups;

int loop_allgather(LoopComm* C, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st) {
    LoopGroup& G = *C->g;
    {
        std::unique_lock<std::mutex> lk(G.mu);
        if (G.broken) { morb::set_error("loopback exchange: a member has left the group"); return ORB_E_ARG; }
        G.send[C->rank] = sendbuf;
        if (hipEventRecord(G.ev[C->rank], st) != hipSuccess) { morb::set_error("loopback exchange: hipEventRecord failed"); return ORB_E_HIP; }
        const unsigned long gen = G.generation;
        if (++G.arrived == G.world) { G.arrived = 0; ++G.generation; G.cv.notify_all(); }
        else if (!G.cv.wait_for(lk, std::chrono::seconds(20), [&] { return G.generation != gen || G.broken; }) || G.broken) {
            G.broken = true; G.cv.notify_all();
            morb::set_error("loopback exchange: rank %d waited 20 s for the other ranks of its group", C->rank);
            return ORB_E_HIP;
        }
    }
    for (int s = 0; s < G.world; ++s) {
        MORB_HIP(hipStreamWaitEvent(st, G.ev[s], 0));
        MORB_HIP(hipMemcpyAsync((uint8_t*)recvbuf + (size_t)s * bytes, G.send[s], bytes, hipMemcpyDeviceToDevice, st));
    }
    {   // nobody re-records its event / republishes its block before every rank has enqueued this round's copies
        std::unique_lock<std::mutex> lk(G.mu);
        const unsigned long gen = G.generation;
        if (++G.arrived == G.world) { G.arrived = 0; ++G.generation; G.cv.notify_all(); }
        else if (!G.cv.wait_for(lk, std::chrono::seconds(20), [&] { return G.generation != gen || G.broken; }) || G.broken) {
            G.broken = true; G.cv.notify_all();
            morb::set_error("loopback exchange: rank %d waited 20 s for the other ranks of its group", C->rank);
            return ORB_E_HIP;
        }
    }
    return ORB_OK;
}
}  // namespace


#include <chrono>
#include <deque>
#include "../../include/orbf.h"

struct orbf_frontend {
    int device = 0, n_cams = 0, max_w = 0, max_h = 0;
    orbx_extractor* ex = nullptr;        // == exs[0]: the extractor isolated steps run on (orbf_extractor)
    orbx_extractor* exs[2] = {nullptr, nullptr};  // small rigs: consecutive overlapped timesteps alternate between two
    orbm_matcher* mt = nullptr;
    std::vector<const float*> d_depth;
    std::vector<int> depth_stride;
    std::vector<int32_t> counts, cam_cap;
    float mbf = 40.f;
    int th_high = ORBM_TH_HIGH, check_ori = 1;
    orb_calibration calib = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // k1 == 0: no undistortion (orbf_set_calibration)
    int cap_total = 0;
    // pinned host result buffers.  The per-feature results exist twice: the extraction of the NEXT timestep (orbf_prefetch)
    // fills the other set while the caller still reads this step's.
    static constexpr int NSETS = 4;  // this step's (held by the caller) + two timesteps in flight + the one being assigned
    struct ResultSet {
        PinnedBuf<orb_keypoint> kps; PinnedBuf<uint8_t> desc; PinnedBuf<float> ur, depth, unx, uny;
        CrossOut cross;            // cross-camera top-2 of the step, computed at the end of its extraction chain
        bool cross_valid = false;  // ... when the chain was enqueued with cross matching on
    } rs[NSETS];
    int last_flags = 0;  // flags of the most recent step: what announced steps are assumed to want
    int cur = 0;       // set holding the results of the last completed step
    int last_set = 0;  // set most recently handed to an extraction (sets are handed out round robin)
    morb::StageBuf h_queries;   // this step's queries: written by the host, read once by k_project
    PinnedBuf<int32_t> h_match;
    // Small rigs (<= 4 cameras): one persistent frame per result set, filled by the extractor's describe kernel (FrameSink)
    orbm_frame* pframe[NSETS] = {nullptr, nullptr, nullptr, nullptr};
    int pframe_W[NSETS] = {0, 0, 0, 0}, pframe_H[NSETS] = {0, 0, 0, 0};
    // extractions in flight for the NEXT steps (enqueued by earlier orbf_step calls after orbf_prefetch), oldest first
    // fp: content fingerprints of the HOST images taken when their upload was enqueued (see image_fingerprint)
    struct InFlight { std::vector<orbf_image> images; std::vector<uint64_t> fp; int set = 0, W = 0, H = 0, e = 0; };
    std::deque<InFlight> inflight;
    std::deque<std::vector<orbf_image>> announced;  // declared by orbf_prefetch, not enqueued yet (at most 2)
    int last_e = 0;  // extractor most recently handed a timestep
    bool poll_ok = true;     // MORB_POLL=0: orbf_step_end always waits with hipStreamSynchronize
    void* xcomm = nullptr; int xworld = 0, xrank = 0;   // native multi-GPU exchange (orbf_exchange_init)
    bool xloop = false;                                  // ... over the in-process loopback transport (orbf_exchange_init_loopback)
    DevBuf<uint8_t> d_xrecv;                            // the gathered export blocks of all ranks
    bool overlap_ok = true;  // cleared when an overlapped extraction had to be redone on the host path ...
    int clean_steps = 0;     // ... and set again after a few steps that stayed on the device path
    struct Pending {  // a timestep between orbf_step_begin and orbf_step_end
        bool active = false, async_path = false, fr_persistent = false, block_ready = false, cross_from_set = false, forked = false;
        bool x_enqueued = false;   // this step's exchange went out between begin and end
        bool inline_match = false; // the step's own extraction was enqueued by this call: its matching follows on the SAME stream
        int set = 0, e = 0, W = 0, H = 0, nq = 0, flags = 0, n = 0;
        orbm_frame* fr = nullptr;
        SearchJob J{nullptr, nullptr, 0, nullptr, false, 0.f, 0, 0, 64, false};
        std::vector<orbf_image> images;
        std::vector<orbm_cam_features> cams;
        std::chrono::steady_clock::time_point t_impl, t_enqueued;
    } pending;
    hipEvent_t ev_extracted = nullptr;  // extractor stream -> matcher stream on the synchronous path
    hipEvent_t ev_ready[NSETS] = {nullptr, nullptr, nullptr, nullptr};  // extraction + frame grid of the step using that set
    // frame of the last completed step (orbf_export_block); a frame built on the synchronous path is kept until the next step
    orbm_frame* last_frame = nullptr; bool last_frame_owned = false;
    // previous step (for orbf_step_motion)
    int prev_n = 0;
    std::vector<int32_t> prev_cam_of;
    std::vector<float> scale_factors;
    std::chrono::steady_clock::time_point t_entry;
};

static int getenv_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static bool small_rig(const orbf_frontend* f) { return f->n_cams <= 4 && f->cap_total <= 8192 && !f->mt->host_resolve; }

extern "C" {

int orbf_create(const orbx_params* params, int n_cams, int max_width, int max_height, int device, orbf_frontend** out) {
    MORB_ARG(params && out && n_cams >= 1 && n_cams <= 64);
    orbf_frontend* f = new orbf_frontend();
    f->device = device; f->n_cams = n_cams; f->max_w = max_width; f->max_h = max_height;
    int rc = orbx_create(params, n_cams, max_width, max_height, device, &f->exs[0]);
    f->ex = f->exs[0];
    if (!rc) rc = orbm_create(device, &f->mt);
    // two streams: the matcher's own one follows the extractor's through events, so that the next step's extraction can
    // run next to this step's matching
    if (rc) { orbf_destroy(f); return rc; }
    f->d_depth.assign(n_cams, nullptr); f->depth_stride.assign(n_cams, 0); f->counts.assign(n_cams, 0);
    { const char* pe = getenv("MORB_POLL"); f->poll_ok = !(pe && atoi(pe) == 0); }
    f->scale_factors.assign(params[0].nlevels, 1.f);
    if ((rc = orbx_tables(&params[0], f->scale_factors.data(), nullptr, nullptr, nullptr, nullptr, nullptr))) { orbf_destroy(f); return rc; }
    for (int c = 0; c < n_cams; ++c) { f->cam_cap.push_back(params[c].nfeatures + 4 * params[c].nlevels); f->cap_total += f->cam_cap.back(); }
    const size_t cap = (size_t)f->cap_total;
    if (hipEventCreateWithFlags(&f->ev_extracted, hipEventDisableTiming) != hipSuccess) { morb::set_error("hipEventCreate failed"); orbf_destroy(f); return ORB_E_HIP; }
    for (int k = 0; k < orbf_frontend::NSETS; ++k)
        if (hipEventCreateWithFlags(&f->ev_ready[k], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { morb::set_error("hipEventCreate failed"); orbf_destroy(f); return ORB_E_HIP; }
    if (!rc) rc = orbx_create(params, n_cams, max_width, max_height, device, &f->exs[1]);  // overlap partner
    for (int k = 0; k < orbf_frontend::NSETS && !rc; ++k)
        if ((rc = f->rs[k].kps.reserve(cap)) || (rc = f->rs[k].desc.reserve(cap * 32)) || (rc = f->rs[k].ur.reserve(cap)) ||
            (rc = f->rs[k].depth.reserve(cap)) || (rc = f->rs[k].unx.reserve(cap)) || (rc = f->rs[k].uny.reserve(cap))) break;
    if (!rc) rc = f->h_match.reserve(cap);
    if (rc) { orbf_destroy(f); return rc; }
    *out = f;
    return ORB_OK;
}

void orbf_destroy(orbf_frontend* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->xcomm) (void)orbf_exchange_shutdown(f);
    f->d_xrecv.release();
    for (int e = 0; e < 2; ++e) if (f->exs[e]) (void)hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[e]));
    if (f->mt) (void)hipStreamSynchronize(f->mt->stream);
    if (f->last_frame && f->last_frame_owned) orbm_frame_destroy(f->last_frame);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) if (f->pframe[k]) orbm_frame_destroy(f->pframe[k]);  // back to the matcher's pool first
    if (f->mt) orbm_destroy(f->mt);
    for (int e = 0; e < 2; ++e) if (f->exs[e]) orbx_destroy(f->exs[e]);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) { f->rs[k].kps.release(); f->rs[k].desc.release(); f->rs[k].ur.release(); f->rs[k].depth.release(); f->rs[k].unx.release(); f->rs[k].uny.release(); f->rs[k].cross.release(); }
    f->h_queries.release(); f->h_match.release();
    if (f->ev_extracted) (void)hipEventDestroy(f->ev_extracted);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) if (f->ev_ready[k]) (void)hipEventDestroy(f->ev_ready[k]);
    delete f;
}

orbx_extractor* orbf_extractor(orbf_frontend* f) { return f ? f->ex : nullptr; }
orbm_matcher* orbf_matcher(orbf_frontend* f) { return f ? f->mt : nullptr; }

int orbf_set_depth(orbf_frontend* f, int cam, const float* d_depth, int stride_floats) {
    MORB_ARG(f && cam >= 0 && cam < f->n_cams);
    f->d_depth[cam] = d_depth; f->depth_stride[cam] = stride_floats;
    return ORB_OK;
}

static int orbf_drain(orbf_frontend* f);

int orbf_set_calibration(orbf_frontend* f, const orb_calibration* calib) {
    MORB_ARG(f != nullptr);
    int rc = orbf_drain(f);  // (prefetched extractions carry the old calibration in their frame sinks)
    if (rc) return rc;
    f->announced.clear();
    if (calib) f->calib = *calib; else memset(&f->calib, 0, sizeof(f->calib));
    return orbm_set_calibration(f->mt, calib);
}

int orbf_configure(orbf_frontend* f, float mbf, int th_high, int check_orientation) {
    MORB_ARG(f && th_high >= 0 && th_high <= 256);
    f->mbf = mbf; f->th_high = th_high; f->check_ori = check_orientation ? 1 : 0;
    return ORB_OK;
}

static int orbf_step_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                          orbf_result* out, bool queries_in_pinned, const orbf_motion* motion = nullptr);
static int orbf_step_begin_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                                bool queries_in_pinned, int* block_ready, const orbf_motion* motion = nullptr);
static int orbf_step_end_impl(orbf_frontend* f, orbf_result* out);
static int orbf_drain(orbf_frontend* f);

int orbf_step(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, orbf_result* out) {
    MORB_ARG(f && images && out && nq >= 0 && (nq == 0 || queries));
    f->t_entry = std::chrono::steady_clock::now();
    return orbf_step_impl(f, images, queries, nq, flags, out, false);
}

static int orbf_drain(orbf_frontend* f);

int orbf_reset(orbf_frontend* f) {
    MORB_ARG(f != nullptr);
    f->prev_n = 0; f->announced.clear(); f->overlap_ok = true;
    if (f->pending.active) {  // a begun step is abandoned with everything else in flight
        if (f->pending.fr && !f->pending.fr_persistent) { (void)hipStreamSynchronize(f->mt->stream); orbm_frame_destroy(f->pending.fr); }
        f->pending.active = false; f->pending.fr = nullptr;
    }
    return orbf_drain(f);
}

int orbf_export_block(orbf_frontend* f, const uint8_t** d_block, size_t* block_bytes, int* cap_rows) {
    MORB_ARG(f && d_block && block_bytes && cap_rows);
    // between orbf_step_begin and orbf_step_end: the frame of the step in flight; otherwise the last completed step's
    const orbm_frame* F = (f->pending.active && f->pending.fr) ? f->pending.fr : f->last_frame;
    if (!F) { morb::set_error("no step to export"); return ORB_E_ARG; }
    *d_block = F->b->d_desc.p;
    *cap_rows = F->desc_rows;
    *block_bytes = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    return ORB_OK;
}

int orbf_exchange_unique_id(uint8_t* out128) {
    MORB_ARG(out128 != nullptr);
    RcclApi& R = rccl();
    if (!R.ok()) { morb::set_error("librccl is not available"); return ORB_E_HIP; }
    XUniqueId id;
    const int r = R.GetUniqueId(&id);
    if (r) return rccl_fail("ncclGetUniqueId", r);
    memcpy(out128, id.internal, 128);
    return ORB_OK;
}

int orbf_exchange_init(orbf_frontend* f, const uint8_t* uid128, int world, int rank) {
    MORB_ARG(f && uid128 && world >= 1 && rank >= 0 && rank < world && world * f->n_cams <= 512 && !f->xcomm);
    RcclApi& R = rccl();
    if (!R.ok()) { morb::set_error("librccl is not available"); return ORB_E_HIP; }
    MORB_HIP(hipSetDevice(f->device));
    XUniqueId id; memcpy(id.internal, uid128, 128);
    void* comm = nullptr;
    const int r = R.CommInitRank(&comm, world, id, rank);
    if (r) return rccl_fail("ncclCommInitRank", r);
    const size_t block = (size_t)f->cap_total * 32 + ORBM_BLOCK_TRAILER;
    int rc = f->d_xrecv.reserve((size_t)world * block);
    if (rc) { (void)R.CommDestroy(comm); return rc; }
    f->xcomm = comm; f->xworld = world; f->xrank = rank;
    return ORB_OK;
}

int orbf_exchange_active(const orbf_frontend* f) { return f && f->xcomm ? f->xworld : 0; }

int orbf_exchange_init_loopback(orbf_frontend* f, int group, int world, int rank) {
    MORB_ARG(f && world >= 1 && rank >= 0 && rank < world && world * f->n_cams <= 512 && !f->xcomm);
    MORB_HIP(hipSetDevice(f->device));
    const size_t block = (size_t)f->cap_total * 32 + ORBM_BLOCK_TRAILER;
    int rc = f->d_xrecv.reserve((size_t)world * block);
    if (rc) return rc;
    LoopGroup* G = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_loop_mu);
        auto it = g_loop_groups.find(group);
        if (it == g_loop_groups.end()) {
            G = new LoopGroup();
            G->world = world; G->send.assign(world, nullptr); G->ev.assign(world, nullptr);
            for (int r = 0; r < world; ++r)
                if (hipEventCreateWithFlags(&G->ev[r], hipEventDisableTiming) != hipSuccess) {
                    for (hipEvent_t e : G->ev) if (e) (void)hipEventDestroy(e);
                    delete G;
                    morb::set_error("loopback exchange: hipEventCreate failed");
                    return ORB_E_HIP;
                }
            g_loop_groups[group] = G;
        } else {
            G = it->second;
            if (G->world != world || G->members >= world) { morb::set_error("loopback group %d: world size mismatch or group full", group); return ORB_E_ARG; }
        }
        std::lock_guard<std::mutex> lk2(G->mu);
        ++G->members;
    }
    f->xcomm = new LoopComm{G, rank}; f->xworld = world; f->xrank = rank; f->xloop = true;
    return ORB_OK;
}

int orbf_exchange_shutdown(orbf_frontend* f) {
    MORB_ARG(f != nullptr);
    if (!f->xcomm) return ORB_OK;
    MORB_HIP(hipSetDevice(f->device));
    if (f->mt) { (void)hipStreamSynchronize(f->mt->side_stream); (void)hipStreamSynchronize(f->mt->stream); }
    if (f->xloop) {
        LoopComm* C = static_cast<LoopComm*>(f->xcomm);
        LoopGroup* G = C->g;
        bool last = false;
        {
            std::lock_guard<std::mutex> lk(G->mu);
            G->broken = true; G->cv.notify_all();          // (a group that lost a member cannot exchange any more)
            last = --G->members == 0;
        }
        if (last) {
            std::lock_guard<std::mutex> lk(g_loop_mu);
            for (auto it = g_loop_groups.begin(); it != g_loop_groups.end(); ++it) if (it->second == G) { g_loop_groups.erase(it); break; }
            for (hipEvent_t e : G->ev) if (e) (void)hipEventDestroy(e);
            delete G;
        }
        delete C;
    } else {
        (void)rccl().CommDestroy(f->xcomm);
    }
    f->xcomm = nullptr; f->xworld = 0; f->xrank = 0; f->xloop = false;
    return ORB_OK;
}

// all-gather of the frame's export block + the gathered cross-camera top-2, all on the matcher's side stream (joined into its
// main stream): the block must be final (its extraction chain has completed, or the main stream has been synchronised)
static int exchange_enqueue(orbf_frontend* f, const orbm_frame* F) {
    RcclApi& R = rccl();
    orbm_matcher* m = f->mt;
    const size_t block = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    MORB_ARG(F->desc_rows == f->cap_total);
    if (f->xloop) {
        const int rc = loop_allgather(static_cast<LoopComm*>(f->xcomm), F->b->d_desc.p, f->d_xrecv.p, block, m->side_stream);
        if (rc) return rc;
    } else {
        const int r = R.AllGather(F->b->d_desc.p, f->d_xrecv.p, block, /*ncclUint8*/ 1, f->xcomm, m->side_stream);
        if (r) return rccl_fail("ncclAllGather", r);
    }
    return orbm_cross_top2_gathered_enqueue(m, f->d_xrecv.p, f->xworld, block, F->desc_rows, f->n_cams, f->xrank, nullptr, 0);
}

static bool same_images(const std::vector<orbf_image>& a, const orbf_image* b, int n);
static bool same_content(const std::vector<uint64_t>& fp, const orbf_image* b, int n);
static std::vector<uint64_t> image_fingerprints(const orbf_image* images, int n);

int orbf_peek_block(orbf_frontend* f, const orbf_image* images, const uint8_t** d_block, size_t* block_bytes, int* cap_rows) {
    MORB_ARG(f && images && d_block && block_bytes && cap_rows);
    *d_block = nullptr; *block_bytes = 0; *cap_rows = 0;
    if (f->pending.active || f->inflight.empty() || !same_images(f->inflight.front().images, images, f->n_cams) ||
        !same_content(f->inflight.front().fp, images, f->n_cams)) return ORB_OK;
    const orbf_frontend::InFlight& I = f->inflight.front();
    MORB_HIP(hipSetDevice(f->device));
    if (hipEventQuery(f->ev_ready[I.set]) != hipSuccess) { (void)hipGetLastError(); return ORB_OK; }
    if (orbx_peek_status(f->exs[I.e]) != 0) return ORB_OK;
    const orbm_frame* F = f->pframe[I.set];
    if (!F) return ORB_OK;
    *d_block = F->b->d_desc.p; *cap_rows = F->desc_rows; *block_bytes = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    return ORB_OK;
}

int orbf_export_features(orbf_frontend* f, orbf_device_features* out) {
    MORB_ARG(f && out);
    const orbm_frame* F = f->last_frame;
    if (!F || (f->pending.active && f->pending.fr)) { morb::set_error("orbf_export_features: no completed step (call it after orbf_step / orbf_step_end)"); return ORB_E_ARG; }
    memset(out, 0, sizeof(*out));
    out->n_cams = f->n_cams; out->n_total = 0;
    for (int c = 0; c < f->n_cams && c < 8; ++c) { out->counts[c] = f->counts[c]; out->n_total += f->counts[c]; }
    out->d_desc = F->b->d_desc.p; out->d_angle = F->b->d_ang.p; out->d_un_x = F->b->d_x.p; out->d_un_y = F->b->d_y.p;
    out->d_octave = F->b->d_oct.p; out->d_uright = F->b->d_ur.p;
    out->stream = f->mt->stream;
    return ORB_OK;
}

int orbf_prefetch(orbf_frontend* f, const orbf_image* next_images) {
    MORB_ARG(f && next_images);
    // (the step about to be called may itself still be in flight: two timesteps beyond it can be announced)
    if (f->inflight.size() + f->announced.size() >= 3) { morb::set_error("too many future timesteps announced (at most two beyond the next step)"); return ORB_E_ARG; }
    f->announced.emplace_back(next_images, next_images + f->n_cams);
    return ORB_OK;
}

int orbf_step_begin(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, int* block_ready) {
    MORB_ARG(f && images && nq >= 0 && (nq == 0 || queries));
    f->t_entry = std::chrono::steady_clock::now();
    int rc = orbf_step_begin_impl(f, images, queries, nq, flags, false, block_ready);
    if (rc) f->pending.active = false;
    return rc;
}

static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out);

int orbf_step_motion_begin(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, int* block_ready) {
    MORB_ARG(f && images && motion);
    f->t_entry = std::chrono::steady_clock::now();
    int rc = orbf_step_begin_impl(f, images, nullptr, 0, flags, true, block_ready, motion);
    if (rc) f->pending.active = false;
    return rc;
}

int orbf_step_end(orbf_frontend* f, orbf_result* out) {
    MORB_ARG(f && out);
    return orbf_step_end_impl(f, out);
}

// the previous step's features (still in their pinned result set) under the stream's motion -> this step's queries
static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out) {
    const int nq = f->prev_n;
    *nq_out = nq;
    if (!nq) return ORB_OK;
    int rc;
    if ((rc = f->h_queries.reserve((size_t)nq * sizeof(orbm_query)))) return rc;
    const orbf_frontend::ResultSet& R = f->rs[f->cur];
    return orbm_queries_from_motion(R.kps.p, R.desc.p, R.depth.p, f->prev_cam_of.data(), nq, motion->du, motion->dv, motion->th,
                                    f->scale_factors.data(), f->mbf, reinterpret_cast<orbm_query*>(f->h_queries.p),
                                    R.unx.p, R.uny.p);  // (mvKeysUn: equal to the keypoint positions without a calibration)
}

int orbf_step_motion(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, orbf_result* out) {
    MORB_ARG(f && images && motion && out);
    f->t_entry = std::chrono::steady_clock::now();
    return orbf_step_impl(f, images, nullptr, 0, flags, out, true, motion);
}

// An extraction that ran ahead is only valid for the step that consumes it if the images are still the ones that were
// uploaded.  Pointers, sizes and strides say nothing about a caller that refilled the same buffer in between, so host images
// also carry a fingerprint of their content: 32 probes of 64 bytes spread over the rows (2 KB per image, well under a
// microsecond), taken when the upload was enqueued and again when the step arrives.  A mismatch drops what is in flight and the
// step extracts its images again.  Device images cannot be probed from the host: they must stay unchanged, as orbf.h says.
static uint64_t image_fingerprint(const orbf_image& im) {
    if (im.on_device || !im.data || im.width <= 0 || im.height <= 0) return 0;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)im.width << 32) ^ (uint64_t)im.height;
    const int span = std::min(64, im.width);
    for (int k = 0; k < 32; ++k) {
        const int row = (int)(((long long)k * im.height) / 32);
        const int col = im.width > span ? (k * 149) % (im.width - span + 1) : 0;
        const uint8_t* p = im.data + (size_t)row * im.stride + col;
        for (int b = 0; b + 8 <= span; b += 8) { uint64_t v; memcpy(&v, p + b, 8); h = (h ^ v) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    }
    return h | 1;
}

static std::vector<uint64_t> image_fingerprints(const orbf_image* images, int n) {
    std::vector<uint64_t> fp(n);
    for (int c = 0; c < n; ++c) fp[c] = image_fingerprint(images[c]);
    return fp;
}

static bool same_content(const std::vector<uint64_t>& fp, const orbf_image* b, int n) {
    if ((int)fp.size() != n) return false;
    for (int c = 0; c < n; ++c) if (fp[c] != image_fingerprint(b[c])) return false;
    return true;
}

static bool same_images(const std::vector<orbf_image>& a, const orbf_image* b, int n) {
    if ((int)a.size() != n) return false;
    for (int c = 0; c < n; ++c)
        if (a[c].data != b[c].data || a[c].width != b[c].width || a[c].height != b[c].height || a[c].stride != b[c].stride ||
            (a[c].on_device != 0) != (b[c].on_device != 0))
            return false;
    return true;
}


static void fill_cam_capacities(orbf_frontend* f, orbx_extractor* ex, orbm_cam_features* cams) {
    for (int c = 0; c < f->n_cams; ++c) {
        cams[c].d_kps = orbx_device_keypoints(ex, c); cams[c].d_desc = orbx_device_descriptors(ex, c);
        cams[c].n = f->cam_cap[c]; cams[c].d_depth = f->d_depth[c]; cams[c].depth_stride = f->depth_stride[c];
    }
}

// Uploads + the whole extractor `e` for one timestep into result set `set`, nothing synchronised.  Small rigs: the
// describe kernel writes the merged frame pframe[set] through a FrameSink (*went_async = 1 unless the extractor took its
// synchronous host-quadtree path; then the frame was not filled).
static int enqueue_extract(orbf_frontend* f, int e, const orbf_image* images, int set, int* W_out, int* H_out, int* went_async,
                           bool with_cross, bool defer_events = false) {
    orbm_matcher* m = f->mt;
    orbx_extractor* ex = f->exs[e];
    int rc, W = 0, H = 0;
    for (int c = 0; c < f->n_cams; ++c) {
        const orbf_image& im = images[c];
        rc = im.on_device ? orbx_upload_device(ex, c, im.data, im.width, im.height, im.stride)
                          : orbx_upload(ex, c, im.data, im.width, im.height, im.stride);
        if (rc) return rc;
        W = std::max(W, im.width); H = std::max(H, im.height);
    }
    if (W == 0 || H == 0) { W = f->max_w; H = f->max_h; }
    *W_out = W; *H_out = H;
    orbf_frontend::ResultSet& R = f->rs[set];
    R.cross_valid = false;
    if ((rc = orbx_set_host_mirror(ex, R.kps.dp, R.desc.dp, f->cap_total))) return rc;
    *went_async = 0;
    const bool small = small_rig(f);
    std::vector<orbm_cam_features> cams(f->n_cams);
    fill_cam_capacities(f, ex, cams.data());
    float bd[4];
    if ((rc = orbm_image_bounds(&f->calib, W, H, bd))) return rc;  // Frame::ComputeImageBounds
    if (f->pframe[set] && (f->pframe_W[set] != W || f->pframe_H[set] != H || f->pframe[set]->minX != bd[0] ||
                           f->pframe[set]->minY != bd[1] || f->pframe[set]->maxX != bd[2] || f->pframe[set]->maxY != bd[3])) {
        orbm_frame_destroy(f->pframe[set]); f->pframe[set] = nullptr;  // (image size or calibration changed)
    }
    m->mirror_ur = R.ur.dp; m->mirror_depth = R.depth.dp; m->mirror_unx = R.unx.dp; m->mirror_uny = R.uny.dp;
    struct MirrorsOff { orbm_matcher* m; ~MirrorsOff() { m->mirror_ur = nullptr; m->mirror_depth = nullptr; m->mirror_unx = nullptr; m->mirror_uny = nullptr; } } mirrors_off{m};
    if (small) {
        // the describe kernel writes the per-feature half of the frame itself (FrameSink)
        FrameSink sink;
        if (!f->pframe[set]) {
            rc = frame_prepare_sink(m, cams.data(), f->n_cams, f->mbf, bd[0], bd[1], bd[2], bd[3], &f->pframe[set], &sink);
            f->pframe_W[set] = W; f->pframe_H[set] = H;
        } else {
            orbm_frame* F = f->pframe[set];
            F->n_total = f->cap_total; F->counts_on_device = true; F->host_valid = false;
            rc = frame_sink_of(m, F, cams.data(), f->n_cams, f->mbf, &sink);
        }
        if (rc) return rc;
        if ((rc = orbx_set_frame_sink(ex, &sink))) return rc;
    } else {
        // larger rigs: the matcher's own kernels assemble the frame from the extractor's per-camera outputs
        if (!f->pframe[set]) {
            m->frame_min_rows = f->cap_total;
            rc = frame_shell(m, f->cap_total, f->n_cams, bd[0], bd[1], bd[2], bd[3], true, &f->pframe[set]);
            m->frame_min_rows = 0;
            if (rc) return rc;
            f->pframe_W[set] = W; f->pframe_H[set] = H;
        } else {
            orbm_frame* F = f->pframe[set];
            F->n_total = f->cap_total; F->counts_on_device = true; F->host_valid = false;
        }
    }
    // The frame's grid (larger rigs: the whole frame assembly) and the camera-pair top-2 are the tail of the extraction
    // chain: built on the extractor's stream right behind the describe kernel (counts read from HBM), so that a step's
    // matching starts with the search itself.  For small rigs the tail is issued from inside orbx_run_async and is
    // captured into the replayed launch chain (no host launches at all on replay); larger rigs launch it here (their
    // assembly stages a parameter block with a copy, which a replayed chain should not carry).
    struct Tail {
        orbf_frontend* f; orbx_extractor* ex; orbm_cam_features* cams; const float* bd; int set; bool small, with_cross;
        static int run(void* u, void* stream) {
            Tail& T = *static_cast<Tail*>(u);
            orbm_matcher* m = T.f->mt;
            orbf_frontend::ResultSet& R = T.f->rs[T.set];
            orbm_frame* frp = T.f->pframe[T.set];
            hipStream_t keep = m->stream;
            m->stream = (hipStream_t)stream;
            int rc = frame_from_device_impl(m, T.cams, T.f->n_cams, T.f->mbf, T.bd[0], T.bd[1], T.bd[2], T.bd[3], orbx_device_counts(T.ex),
                                            &frp, T.small);
            if (!rc && T.with_cross && T.f->n_cams > 1) {
                const int ncap = frp->n_total;
                rc = cross_enqueue_to(m->stream, frp->b->d_desc.p, ncap, frp->b->d_cam_start.p, T.f->n_cams, 0, ncap, frp->b->d_ntotal.p,
                                      R.cross.i.dp, R.cross.b.dp, R.cross.s.dp, R.cross.scratch.p);
            }
            m->stream = keep;
            return rc;
        }
    } tail{f, ex, cams.data(), bd, set, small, with_cross};
    const bool cross_here = with_cross && f->n_cams > 1;
    if (cross_here && (rc = R.cross.reserve(f->cap_total, f->cap_total))) return rc;  // (storage first: nothing allocates inside a capture)
    if (small) { if ((rc = m->h_ring.reserve((64 * sizeof(CamFeat) + 65 * sizeof(int) + 64 * sizeof(int)) * 4))) return rc; if ((rc = frame_build_lds_limit())) return rc; }
    if (small && (rc = orbx_set_chain_tail(ex, &Tail::run, &tail, 1 + set * 2 + (cross_here ? 1 : 0)))) return rc;
    const int before = orbx_pending(ex);
    rc = orbx_run_async(ex);
    if (small) { (void)orbx_set_frame_sink(ex, nullptr); (void)orbx_set_chain_tail(ex, nullptr, nullptr, 0); }
    if (rc) return rc;
    *went_async = orbx_pending(ex) > before ? 1 : 0;
    if (*went_async) {
        if (!small && (rc = Tail::run(&tail, orbx_stream(ex)))) return rc;
        R.cross_valid = cross_here;
        if (!defer_events) {   // (an inline step records its events behind its matching: step_enqueue)
            hipError_t he = hipEventRecord(f->ev_ready[set], (hipStream_t)orbx_stream(ex));
            if (he != hipSuccess) { morb::set_error("hipEventRecord: %s", hipGetErrorString(he)); return ORB_E_HIP; }
        }
    }
    return ORB_OK;
}

// Everything in flight is waited for and dropped (results of prefetched extractions included).
static int orbf_drain(orbf_frontend* f) {
    MORB_HIP(hipSetDevice(f->device));
    for (int e = 0; e < 2; ++e) if (f->exs[e]) MORB_HIP(hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[e])));
    MORB_HIP(hipStreamSynchronize(f->mt->stream));
    MORB_HIP(hipStreamSynchronize(f->mt->side_stream));
    for (int e = 0; e < 2; ++e)
        while (f->exs[e] && orbx_pending(f->exs[e]) > 0) { int rc = orbx_finish(f->exs[e]); if (rc < 0) return rc; }
    f->inflight.clear();
    return ORB_OK;
}

// The next free result set / extractor for a timestep that is about to be extracted.  Sets go round robin.  Isolated steps
// (nothing in flight) always run on extractor 0; overlapped ones alternate, so that two extraction chains are on the GPU
// at a time and each extractor keeps seeing the same two (count slot, result set) pairs -- its captured launch chains stay valid.
static void next_slot(orbf_frontend* f, int* e, int* set) {
    *set = (f->last_set + 1) % orbf_frontend::NSETS;
    if (*set == f->cur) *set = (*set + 1) % orbf_frontend::NSETS;  // (the caller still reads the last step's results)
    *e = (f->inflight.empty() || !f->exs[1]) ? 0 : (f->last_e ^ 1);
    f->last_set = *set; f->last_e = *e;
}

// A timestep in two halves.  orbf_step_begin enqueues everything (this step's matching, the extraction of the announced
// steps) and returns; orbf_step_end blocks once and collects.  Between the two a caller may enqueue work of its own that
// only needs the step's export block -- the multi-GPU exchange -- when begin reported the block ready.
static int step_enqueue(orbf_frontend* f, orbf_frontend::Pending& P, bool first_attempt);

static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out);

// motion != NULL: the queries are built here from the previous step's features (orbf_step_motion) -- AFTER this step's
// extraction has been enqueued, so that the GPU is already working while the host projects the points.
static int orbf_step_begin_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                                bool queries_in_pinned, int* block_ready, const orbf_motion* motion) {
    orbf_frontend::Pending& P = f->pending;
    MORB_ARG(!P.active);
    P = orbf_frontend::Pending();
    P.t_impl = std::chrono::steady_clock::now();
    MORB_HIP(hipSetDevice(f->device));
    orbm_matcher* m = f->mt;
    int rc, went_async = 0;
    if (f->last_frame && f->last_frame_owned) orbm_frame_destroy(f->last_frame);
    f->last_frame = nullptr; f->last_frame_owned = false;
    if (f->xcomm) flags |= ORBF_SKIP_CROSS;   // the rig-wide matching of the exchange replaces the rank-local one
    P.images.assign(images, images + f->n_cams);
    P.nq = nq; P.flags = flags;

    // ---- this step's extraction: already in flight (orbf_prefetch during an earlier step) or enqueued now
    if (!f->inflight.empty() && same_images(f->inflight.front().images, images, f->n_cams) &&
        same_content(f->inflight.front().fp, images, f->n_cams)) {
        const orbf_frontend::InFlight& I = f->inflight.front();
        P.set = I.set; P.e = I.e; P.W = I.W; P.H = I.H; went_async = 1;
        f->inflight.pop_front();
    } else {
        if (!f->inflight.empty()) {  // prefetched for other images: everything in flight is dropped
            if ((rc = orbf_drain(f))) return rc;
            f->announced.clear();
        }
        if (!f->announced.empty() && same_images(f->announced.front(), images, f->n_cams)) f->announced.pop_front();
        next_slot(f, &P.e, &P.set);
        // Nothing ran ahead for this step (a live rig: the images have only just arrived).  Its matching then goes onto the
        // extractor's own stream, right behind the extraction chain -- a kernel boundary instead of a cross-stream event
        // (measured: ~22 us between the chain's last kernel and the projection kernel on the matcher's stream) -- and the
        // camera-pair top-2 leaves the chain: it forks onto the side stream next to project + resolve instead of standing in
        // front of them.
        P.inline_match = small_rig(f) && !f->xcomm && getenv_int("MORB_INLINE_MATCH", 1) != 0;
        if (P.inline_match) { (void)orbx_set_chain_graph(f->exs[P.e], getenv_int("MORB_INLINE_GRAPH", 0)); (void)orbx_set_defer_done(f->exs[P.e], 1); }
        rc = enqueue_extract(f, P.e, images, P.set, &P.W, &P.H, &went_async, !(flags & ORBF_SKIP_CROSS) && !P.inline_match, P.inline_match);
        if (P.inline_match) { (void)orbx_set_chain_graph(f->exs[P.e], 1); (void)orbx_set_defer_done(f->exs[P.e], 0); }
        if (rc) return rc;
        if (P.inline_match && !went_async) P.inline_match = false;   // (host-quadtree path: everything was synchronous)
    }
    if (motion) {
        if ((rc = queries_from_previous_step(f, motion, &nq))) return rc;
        queries = reinterpret_cast<const orbm_query*>(f->h_queries.p); queries_in_pinned = true;
        P.nq = nq;
    }
    // queries go through pinned (device-mapped) staging and are read from there by the projection kernel
    if (nq) {
        if ((rc = f->h_queries.reserve((size_t)nq * sizeof(orbm_query))) || (rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query))))
            return rc;
        if (!queries_in_pinned) memcpy(f->h_queries.p, queries, (size_t)nq * sizeof(orbm_query));
        f->h_queries.publish();
    }
    P.J = SearchJob{nullptr, reinterpret_cast<const orbm_query*>(f->h_queries.p), nq, nullptr, false, 0.f, f->th_high, f->check_ori, 64, false};
    P.J.q_dev = nq ? reinterpret_cast<const orbm_query*>(f->h_queries.dp) : nullptr;   // no H2D on the step's critical chain
    P.J.want_tags = f->poll_ok;
    if ((rc = f->h_match.reserve(std::max(f->cap_total, 1)))) return rc;
    P.async_path = went_async != 0;
    // The export block of this step is final already when its extraction chain has completed cleanly (the usual case with
    // steps announced ahead): then nothing of this step can be redone and a caller may ship the block right away.
    P.block_ready = false;
    if (P.async_path && !P.inline_match && hipEventQuery(f->ev_ready[P.set]) == hipSuccess) P.block_ready = orbx_peek_status(f->exs[P.e]) == 0;
    else (void)hipGetLastError();
    if ((rc = step_enqueue(f, P, true))) return rc;
    P.active = true;
    if (block_ready) *block_ready = P.block_ready ? 1 : 0;
    return ORB_OK;
}

// Enqueues the matching of the pending step (and, on the first attempt, the extraction of the announced steps).
static int step_enqueue(orbf_frontend* f, orbf_frontend::Pending& P, bool first_attempt) {
    orbm_matcher* m = f->mt;
    hipStream_t st = m->stream;
    orbx_extractor* ex = f->exs[P.e];
    hipStream_t st_e = (hipStream_t)orbx_stream(ex);
    orbf_frontend::ResultSet& R = f->rs[P.set];
    const bool do_cross = !(P.flags & ORBF_SKIP_CROSS);
    int rc;
    std::vector<orbm_cam_features>& cams = P.cams;
    cams.resize(f->n_cams);
    const bool inline_match = P.async_path && P.inline_match;
    struct StreamSwap {   // an inline step issues its matching on the extractor's stream
        orbm_matcher* m; hipStream_t keep; bool on;
        ~StreamSwap() { if (on) m->stream = keep; }
    } swap{m, m->stream, inline_match};
    if (inline_match) { m->stream = st_e; st = st_e; }
    if (P.async_path) {
        // matching follows the extraction chain (which ends with the frame grid): through its event, or simply behind it on
        // the same stream; counts are in HBM
        P.fr = f->pframe[P.set]; P.fr_persistent = true;
        if (!inline_match) MORB_HIP(hipStreamWaitEvent(st, f->ev_ready[P.set], 0));  // extraction + frame grid of this step
        P.n = P.fr->n_total;
    } else {
        rc = orbx_finish(ex);  // synchronises; counts are on the host from here on
        if (rc < 0) return rc;
        // the host-quadtree path returns with its describe kernel still running on the extractor's stream
        MORB_HIP(hipEventRecord(f->ev_extracted, st_e));
        MORB_HIP(hipStreamWaitEvent(st, f->ev_extracted, 0));
        P.n = 0;
        for (int c = 0; c < f->n_cams; ++c) {
            cams[c].d_kps = orbx_device_keypoints(ex, c); cams[c].d_desc = orbx_device_descriptors(ex, c);
            cams[c].n = orbx_count(ex, c);
            cams[c].d_depth = f->d_depth[c]; cams[c].depth_stride = f->depth_stride[c];
            P.n += cams[c].n;
        }
        // the frame-build kernel mirrors the stereo arrays straight into this step's pinned result set (keypoints and
        // descriptors were mirrored by the extractor's describe kernel)
        m->mirror_kps = nullptr; m->mirror_desc = nullptr; m->mirror_ur = R.ur.dp; m->mirror_depth = R.depth.dp;
        m->mirror_unx = R.unx.dp; m->mirror_uny = R.uny.dp;
        m->frame_min_rows = f->cap_total;  // every step's export block has the same size
        float bd[4];
        rc = orbm_image_bounds(&f->calib, P.W, P.H, bd);  // Frame::ComputeImageBounds
        P.fr = nullptr;
        if (!rc) rc = frame_from_device_impl(m, cams.data(), f->n_cams, f->mbf, bd[0], bd[1], bd[2], bd[3], nullptr, &P.fr);
        m->frame_min_rows = 0;
        m->mirror_ur = nullptr; m->mirror_depth = nullptr; m->mirror_unx = nullptr; m->mirror_uny = nullptr;
        if (rc) return rc;
        P.fr_persistent = false;
    }
    orbm_frame* fr = P.fr;
    const int n = P.n;
    P.J.cur = fr; P.J.cap = 64; P.J.device_path = false;
    // fork: the camera-pair top-2 only needs the frame's descriptor block, so it runs on the side stream next to
    // project + resolve (both are a handful of workgroups on a 256-CU part); join before the one host sync
    // (on the asynchronous path the cross top-2 normally rode at the end of the step's extraction chain already)
    P.cross_from_set = do_cross && P.async_path && R.cross_valid;
    const bool forked = do_cross && n > 0 && !P.cross_from_set;
    P.forked = forked;
    if (forked) {  // the fork point is the finished frame; the launches on the side stream come after the search's
        hipError_t fe = hipEventRecord(m->ev_fork, st);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(m->side_stream, m->ev_fork, 0);
        if (fe != hipSuccess) { morb::set_error("stream fork: %s", hipGetErrorString(fe)); if (!P.fr_persistent) orbm_frame_destroy(fr); P.fr = nullptr; return ORB_E_HIP; }
    }
    rc = search_enqueue(m, P.J, /*queries_already_on_device=*/true);
    if (forked) {
        if (!rc) rc = cross_enqueue(m, m->side_stream, fr->b->d_desc.p, n, fr->b->d_cam_start.p, f->n_cams, 0, n,
                                    P.async_path ? fr->b->d_ntotal.p : nullptr);
        // join (also on the error path, so that the side stream never outlives the frame)
        hipError_t je = hipEventRecord(m->ev_join, m->side_stream);
        if (je == hipSuccess) je = hipStreamWaitEvent(st, m->ev_join, 0);
        if (!rc && je != hipSuccess) { morb::set_error("stream join: %s", hipGetErrorString(je)); rc = ORB_E_HIP; }
    }
    if (inline_match && first_attempt) {   // the events the extraction left for us: behind the matching, not in front of it
        const int rd = orbx_record_done(ex);
        hipError_t he = hipEventRecord(f->ev_ready[P.set], st);
        if (!rc && rd) rc = rd;
        if (!rc && he != hipSuccess) { morb::set_error("hipEventRecord: %s", hipGetErrorString(he)); rc = ORB_E_HIP; }
    }
    if (rc) { (void)hipStreamSynchronize(st); if (!P.fr_persistent) orbm_frame_destroy(fr); P.fr = nullptr; return rc; }
    // ---- native exchange: a block that is final already goes out right behind the step's own matching
    if (first_attempt && f->xcomm && P.async_path && P.block_ready && !P.x_enqueued) {
        if ((rc = exchange_enqueue(f, fr))) return rc;
        P.x_enqueued = true;
    }
    // ---- announced timesteps go onto the extractors now: they run while this step is being matched.  At most two
    // are in flight; consecutive ones alternate between the two extractors (an extractor takes its next timestep as
    // a second run behind the one whose results are being matched here).
    while (P.async_path && first_attempt && f->overlap_ok && !f->announced.empty() && f->inflight.size() < 2) {
        const int prev_e = f->inflight.empty() ? P.e : f->inflight.back().e;
        const int e2 = f->exs[1] ? (prev_e ^ 1) : 0;
        if (orbx_pending(f->exs[e2]) >= 2) break;
        int set2 = (f->last_set + 1) % orbf_frontend::NSETS;
        if (set2 == f->cur) set2 = (set2 + 1) % orbf_frontend::NSETS;
        if (set2 == P.set) set2 = (set2 + 1) % orbf_frontend::NSETS;
        int w2 = 0, h2 = 0, async2 = 0;
        rc = enqueue_extract(f, e2, f->announced.front().data(), set2, &w2, &h2, &async2, !(P.flags & ORBF_SKIP_CROSS));
        if (rc) { (void)hipStreamSynchronize(st); return rc; }
        f->last_set = set2; f->last_e = e2;
        if (async2) {
            orbf_frontend::InFlight I;
            I.images = f->announced.front(); I.set = set2; I.W = w2; I.H = h2; I.e = e2;
            I.fp = image_fingerprints(I.images.data(), f->n_cams);   // (the uploads were enqueued just above)
            f->inflight.push_back(std::move(I));
            f->announced.pop_front();
        } else {
            // the extractor ran synchronously (host quadtree): its outputs now belong to that future step, which cannot
            // be kept apart from a later one's -- give up overlapping; the steps extract again when their turn comes
            f->overlap_ok = false;
            f->announced.clear();
        }
    }
    P.t_enqueued = std::chrono::steady_clock::now();
    return ORB_OK;
}

static int orbf_step_end_impl(orbf_frontend* f, orbf_result* out) {
    orbf_frontend::Pending& P = f->pending;
    MORB_ARG(P.active && out);
    P.active = false;
    auto us_between = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<float, std::micro>(b - a).count();
    };
    MORB_HIP(hipSetDevice(f->device));
    orbm_matcher* m = f->mt;
    orbx_extractor* ex = f->exs[P.e];
    struct StreamSwap {   // an inline step's matching lives on the extractor's stream (retries of the search go there too)
        orbm_matcher* m; hipStream_t keep; bool on;
        ~StreamSwap() { if (on) m->stream = keep; }
    } swap{m, m->stream, P.async_path && P.inline_match};
    if (swap.on) m->stream = (hipStream_t)orbx_stream(ex);
    hipStream_t st = m->stream;
    orbf_frontend::ResultSet& R = f->rs[P.set];
    int rc, nmatches = 0;
    auto t_synced = P.t_impl;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1 && (rc = step_enqueue(f, P, false))) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t herr = hipSuccess;
        bool polled = false;
        if (f->poll_ok && P.async_path && P.J.pollable && !P.forked && !m->foreign_work) {
            // the resolve is the last thing on the stream and tags its result words with the launch's sequence number: watch
            // the status word arrive (a few microseconds sooner than the end-of-kernel signal travels through the runtime);
            // search_finish then takes every other word the same way
            volatile int32_t* flag = m->h_match.p;
            for (int spin = 0; spin < 400000; ++spin) {
                if ((*flag >> 20) == P.J.seq) { polled = true; break; }
                __builtin_ia32_pause();
            }
        }
        if (!polled) herr = hipStreamSynchronize(st);
        m->foreign_work = false;
        out->gpu_wait_us = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
        out->host_us[0] = us_between(f->t_entry, P.t_impl); out->host_us[1] = us_between(P.t_impl, P.t_enqueued); out->host_us[2] = out->gpu_wait_us;
        t_synced = std::chrono::steady_clock::now();
        if (herr != hipSuccess) {
            morb::set_error("hipStreamSynchronize: %s", hipGetErrorString(herr));
            if (!P.fr_persistent && P.fr) orbm_frame_destroy(P.fr);
            return ORB_E_HIP;
        }
        if (P.async_path) {
            rc = orbx_finish(ex);  // this step's run (the oldest of its extractor) completed long ago: adopts its counts
            if (rc < 0) return rc;
            if (rc == 1 || rc == 2) {
                // a pyramid level was outside the device quadtree's limits: this step is redone on the synchronous path
                if (rc == 2 || !f->inflight.empty()) {  // ... from its images: later timesteps are in flight, drop them first
                    if ((rc = orbf_drain(f))) return rc;
                    f->announced.clear();
                    f->overlap_ok = false;
                    int w2, h2, a2;
                    if ((rc = enqueue_extract(f, P.e, P.images.data(), P.set, &w2, &h2, &a2, false))) return rc;
                    if (a2) { rc = orbx_finish(ex); if (rc < 0) return rc; }
                }
                P.fr = nullptr; P.fr_persistent = false;
                P.async_path = false;
                f->clean_steps = 0;
                continue;
            }
            for (int c = 0; c < f->n_cams; ++c) f->counts[c] = orbx_count(ex, c);
            frame_set_counts(P.fr, f->counts.data());
            P.n = P.fr->n_total;
            if (!f->overlap_ok && ++f->clean_steps >= 3) f->overlap_ok = true;  // (e.g. the extractor has switched its BIG pass on)
        } else {
            for (int c = 0; c < f->n_cams; ++c) f->counts[c] = P.cams[c].n;
        }
        break;
    }
    if (!P.async_path || !f->overlap_ok) f->announced.clear();  // (hints are only honoured on the asynchronous path)
    const int n = P.n, nq = P.nq;
    const bool do_cross = !(P.flags & ORBF_SKIP_CROSS) && n > 0;
    rc = search_finish(m, P.J, f->h_match.p, &nmatches);
    if (rc) { if (!P.fr_persistent) orbm_frame_destroy(P.fr); return rc; }
    f->last_frame = P.fr; f->last_frame_owned = !P.fr_persistent;  // (returned to the pool when the next step starts)
    f->cur = P.set;
    f->prev_n = n;
    f->prev_cam_of.resize(n);
    for (int c = 0, g = 0; c < f->n_cams; ++c)
        for (int k = 0; k < f->counts[c]; ++k) f->prev_cam_of[g++] = c;
    out->n_queries = nq; out->queries = reinterpret_cast<const orbm_query*>(f->h_queries.p);
    out->n_cams = f->n_cams; out->n_total = n; out->counts = f->counts.data();
    out->kps = R.kps.p; out->desc = R.desc.p; out->uright = R.ur.p; out->depth = R.depth.p;
    out->un_x = R.unx.p; out->un_y = R.uny.p;
    out->nmatches = nmatches; out->match_of_feature = f->h_match.p;
    const bool from_set = P.cross_from_set && P.async_path;  // (a step redone on the synchronous path matched in its own launch)
    out->cross_best_idx = do_cross ? (from_set ? R.cross.i.p : m->h_c0.p) : nullptr;
    out->cross_best_dist = do_cross ? (from_set ? R.cross.b.p : m->h_c1.p) : nullptr;
    out->cross_second_dist = do_cross ? (from_set ? R.cross.s.p : m->h_c2.p) : nullptr;
    out->rig_cams = 0; out->rig_counts = nullptr;
    if (f->xcomm) {
        // every rank issues exactly one all-gather per step: between begin and end when the block was final at begin, here
        // otherwise (the block is final now)
        if (!P.x_enqueued) {
            if ((rc = exchange_enqueue(f, f->last_frame))) return rc;
            MORB_HIP(hipStreamSynchronize(st));
            m->foreign_work = false;
        }
        out->cross_best_idx = m->h_c0.p; out->cross_best_dist = m->h_c1.p; out->cross_second_dist = m->h_c2.p;
        out->rig_cams = m->gathered_cams; out->rig_counts = m->h_gcnt.p;
        if (m->h_gcnt.p[m->gathered_cams + 1] != 0) {   // (k_repack_gathered clamped a remote count: nothing ran out of bounds)
            morb::set_error("multi-GPU exchange: %d per-camera counts of the gathered blocks were out of range (ranks disagree on "
                            "their capacities, or a block is corrupt)", m->h_gcnt.p[m->gathered_cams + 1]);
            return ORB_E_ARG;
        }
    }
    out->host_us[3] = us_between(t_synced, std::chrono::steady_clock::now());
    return ORB_OK;
}

static int orbf_step_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                          orbf_result* out, bool queries_in_pinned, const orbf_motion* motion) {
    int rc = orbf_step_begin_impl(f, images, queries, nq, flags, queries_in_pinned, nullptr, motion);
    if (rc) { f->pending.active = false; return rc; }
    return orbf_step_end_impl(f, out);
}

}  // extern "C"

#ifdef MORB_PHASE_CLOCKS
extern "C" int morb_debug_phases_matcher(int which, unsigned long long* out64) {
    hipError_t e = which == 0 ? hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_res), 64 * sizeof(unsigned long long))
                              : hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_fb), 64 * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : -1;
}
#endif
