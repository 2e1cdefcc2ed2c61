// hamming.hip -- the all-pairs Hamming kernels of the ORB matcher (include/orbm.h) and everything that launches them.
//
//   k_hamming_top2 / k_cross_top2        M1  exhaustive top-2, xor + popcount form: one query per lane, references walked with
//                                        wave-uniform (scalar-cache) loads, 16 waves per block each scanning 1/16 of the
//                                        references, LDS merge.  VALU-bound.  reference src/ORBmatcher.cc:287-321.
//   k_hamming_top2_mfma / k_cross_top2_mfma  the same results from v_mfma_i32_32x32x32_i8 on the +-1-expanded descriptors
//                                        (dot = 256 - 2 * distance, exact); accumulators come out as ready-made sort keys.
//   k_hamming_matrix[_mfma]              M2  full uint16 distance matrix, popcount / matrix-core form (HBM-write-bound).
//   k_repack_gathered                    multi-GPU: the all-gathered export blocks -> one contiguous descriptor list.
// This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (MFMA results straight into VGPRs, csrc/Makefile).
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
#include "hamming_dev.h"
#include "project_dev.h"
#include "mirror_dev.h"

using namespace morb;

namespace {


// ------------------------------------------------------------------------------------------------ kernels
constexpr int TOP2_WAVES = 16;

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
    MORB_LATENCY_KERNEL_WIDE();
    if (d_range) nq = d_range[2];  // partial arrays are laid out with stride nq: the producer used the same device count
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    top2_merge_query(p_idx, p_best, p_second, S, nq, qi, best_idx, best_dist, second_dist);
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

// The matrix instructions take their accumulators in ordinary vector registers.  With a launch bound of 256 threads alone the compiler
// assumes one wave per SIMD may want more than 256 registers, puts the accumulators into the AccVGPR half of the file and moves every result
// across with v_accvgpr_read before the vector ALU can touch it (128-160 moves per tile step in these kernels, 180 registers instead of 148
// for the FP4 top-2, 240 instead of 168 for the distance matrix: two waves per SIMD instead of three).  Promising two waves per SIMD caps the
// budget at 256 registers and the compiler selects the VGPR form.  (Rounds 2-4 got the same code through `-mllvm -amdgpu-mfma-vgpr-form` in
// the Makefile; said here it holds however the file is built, and tests/test_isa_guard.py checks the result.)
#define MORB_MFMA_IN_VGPRS __attribute__((amdgpu_waves_per_eu(2)))

__global__ __launch_bounds__(64 * MM_WAVES) MORB_MFMA_IN_VGPRS void k_hamming_matrix_mfma(const uint32_t* __restrict__ q, int nq,
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

// Pieces shared by the two top-2 kernels.  Round 3: a tile's 64 references are two halves of 32 (a = 0, 1) with an accumulator
// pair each; while the MFMAs of one half run, the wave works through the 2 x 16 sort keys per lane of the half before --
// in rounds 1-2 a wave issued all 32 MFMAs of a tile and only then started on its 128 keys, so the matrix pipe and the vector
// ALU took turns (0.41 of the int8 peak at two waves per SIMD; the two accumulator sets of the pipelined form are live at
// different times, the kernel needs 166 registers instead of 229 = three waves per SIMD, and 32 000 x 32 000 went 252 -> 199 us).
//
// Two forms of the arithmetic (Mt<FP4>), same tiling, same keys, same results:
//   Mt<false>  int8: v_mfma_i32_32x32x32_i8, 8 instructions per 32 x 32 block of 256-bit dot products (above).
//   Mt<true>   FP4 (round 3, the default): gfx950's v_mfma_f32_32x32x64_f8f6f4 takes 64 E2M1 values per lane pair and instruction at
//              twice the int8 rate -- 4 instructions per block -- and half the LDS bytes per reference.  A bit becomes +-4 (reference:
//              set = +4 = 0x6, clear = -4 = 0xE; query: the other way round), a product is -16 where the bits agree and +16 where
//              they differ, the 256-bit dot product is 32 * distance - 4096, and with the accumulators preset to 4096 + row the f32
//              result IS the key 32 * distance + row: an integer below 2^24, so exact, and positive, so its BIT PATTERN orders like
//              the number -- the integer min / med3 chain works on it unchanged.  Which of a lane's 32 K-slots a bit lands in does not
//              matter: both operands are expanded by the same function.  32 000 x 32 000: 196-212 -> 141-145 us in the same run
//              (tools/experiments/top2_fp4.hip), bit-identical at every size tried.
struct Top2Run { uint32_t kb[2], ks2[2]; int where[2]; };   // per query group g: best key, second key, (block << 5 | row) of the best
using mm_i32x8 = __attribute__((ext_vector_type(8))) int;
using mm_f32x16 = __attribute__((ext_vector_type(16))) float;

// 32 descriptor bits -> 32 FP4 values (+-4): two bits select one byte (two nibbles) of a four-byte pool with v_perm_b32.  Output dword j
// takes the 2-bit fields j of the word's four bytes -- which K slot of the instruction a bit lands in is free as long as both operands
// agree, so no bit has to be moved next to its neighbours: one shift, one mask, one permute per dword (11 instructions per word; the
// expansion that kept the bits in order cost 24).
template <bool QUERY>
__device__ __forceinline__ mm_i32x4 f4_expand32(uint32_t bits) {
    constexpr uint32_t POOL = QUERY ? 0xEEE66E66u : 0x666EE6EEu;   // byte f = nibble(bit 1) << 4 | nibble(bit 0), f = the two bits
    mm_i32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (int)__builtin_amdgcn_perm(0u, POOL, (bits >> (2 * j)) & 0x03030303u);
    return v;
}

template <bool FP4> struct Mt;
template <> struct Mt<false> {
    using Acc = mm_i32x16;
    static constexpr int KS = 8;                        // matrix instructions per 32 x 32 block
    static constexpr uint32_t KEY_NONE = 256u << 6;     // key = distance << 6 | row
    static constexpr uint32_t ROW_MASK = 63u;
    static __device__ __forceinline__ void queries(mm_i32x4 (&bq)[KS], const uint32_t (&w)[8], int h) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[ks] = mm_expand16(h ? (w[ks] >> 16) : (w[ks] & 0xffffu));
    }
    static __device__ __forceinline__ void preset(Acc& c, int h) {
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = 8192 + (e & 3) + 8 * (e >> 2) + 4 * h;
    }
    // this thread's part of a tile: reference row `lane` (half h, row c), words 2 wave and 2 wave + 1
    static __device__ __forceinline__ void deposit(mm_i32x4* tile, int h, int wave, int c, uint2 w) {
        mm_i32x4* base = tile + (h * 8 + wave * 2) * 64 + c;
        base[0] = mt_expand16(w.x & 0xffffu);
        base[32] = mt_expand16(w.x >> 16);
        base[64] = mt_expand16(w.y & 0xffffu);
        base[96] = mt_expand16(w.y >> 16);
    }
    static __device__ __forceinline__ Acc mfma(mm_i32x4 a, mm_i32x4 b, Acc c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint32_t key(const Acc& a, int e) { return (uint32_t)a[e]; }
    static __device__ __forceinline__ void close_block(uint32_t& kb, int& where, uint32_t before, int blk) {
        where = kb != before ? ((blk << 5) | (int)(kb & 31u)) : where;
        kb &= ~63u;
    }
    static __device__ __forceinline__ uint32_t distance(uint32_t k) { return k >> 6; }
};
template <> struct Mt<true> {
    using Acc = mm_f32x16;
    static constexpr int KS = 4;
    // Round 4: the accumulators are preset to 16384 + 4096 + row, so that every key 16384 + 32 * distance + row (distance 0 .. 256)
    // lies in ONE binade, [16384, 32768): its bit pattern is 0x46800000 | (32 * distance + row) << 9 -- row in mantissa bits 9..13,
    // distance in bits 14..22 -- and closing a block is integer work on the pattern (one v_and clears the row, a bit-field extract
    // reads it) instead of a conversion to integer, the mask and a conversion back per query group and block.
    static constexpr uint32_t KEY_NONE = 0x46C00000u;   // 16384 + 32 * 256 = 24576.0f: "distance 256"
    static constexpr uint32_t ROW_MASK = 31u << 9;
    static __device__ __forceinline__ void queries(mm_i32x4 (&bq)[KS], const uint32_t (&w)[8], int h) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[ks] = f4_expand32<true>(h ? w[2 * ks + 1] : w[2 * ks]);
    }
    static __device__ __forceinline__ void preset(Acc& c, int h) {
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = (float)(16384 + 4096 + (e & 3) + 8 * (e >> 2) + 4 * h);
    }
    static __device__ __forceinline__ void deposit(mm_i32x4* tile, int h, int wave, int c, uint2 w) {
        mm_i32x4* base = tile + (h * 4 + wave) * 64 + c;   // instruction ks = wave of half h; lanes c / 32 + c take word 2 ks / 2 ks + 1
        base[0] = f4_expand32<false>(w.x);
        base[32] = f4_expand32<false>(w.y);
    }
    static __device__ __forceinline__ Acc mfma(mm_i32x4 a, mm_i32x4 b, Acc c) {
        const mm_i32x8 a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);   // cbsz = blgp = 4: FP4; no block scales
    }
    static __device__ __forceinline__ uint32_t key(const Acc& a, int e) { return __float_as_uint(a[e]); }
    static __device__ __forceinline__ void close_block(uint32_t& kb, int& where, uint32_t before, int blk) {
        where = kb != before ? ((blk << 5) | (int)((kb >> 9) & 31u)) : where;
        kb &= ~(31u << 9);
    }
    static __device__ __forceinline__ uint32_t distance(uint32_t k) { return (k >> 14) & 0x1ffu; }
};

template <bool FP4>
__device__ __forceinline__ void mt_mfma_half(typename Mt<FP4>::Acc (&acc)[2], const mm_i32x4* __restrict__ tile, int a, int lane,
                                             const mm_i32x4 (&bq)[2][Mt<FP4>::KS], const typename Mt<FP4>::Acc& cinit) {
#pragma unroll
    for (int ks = 0; ks < Mt<FP4>::KS; ++ks) {
        const mm_i32x4 af = tile[(a * Mt<FP4>::KS + ks) * 64 + lane];
        acc[0] = Mt<FP4>::mfma(af, bq[0][ks], ks ? acc[0] : cinit);
        acc[1] = Mt<FP4>::mfma(af, bq[1][ks], ks ? acc[1] : cinit);
    }
}

// the 16 keys of one 32 x 32 block per query group; out(g, e) = "this row does not count for the queries of group g"
template <bool FP4, class OutFn>
__device__ __forceinline__ void mt_keys(Top2Run& R, const typename Mt<FP4>::Acc (&acc)[2], int blk, OutFn out) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t before = R.kb[g];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t key = out(g, e) ? Mt<FP4>::KEY_NONE : Mt<FP4>::key(acc[g], e);
            R.ks2[g] = mt_umed3(R.kb[g], R.ks2[g], key);
            R.kb[g] = min(R.kb[g], key);
        }
        Mt<FP4>::close_block(R.kb[g], R.where[g], before, blk);
    }
}
struct MtAll { __device__ __forceinline__ bool operator()(int, int) const { return false; } };

// Round 4: one half-tile step with the key work GATED.  A key changes (best, second) of its lane only if it is below the lane's
// `second` (best <= second always: second' = med3(best, second, key) >= best' = min(best, key), and closing a block only lowers
// best) -- and once a query has seen a few hundred references almost no key is.  So the 16 keys of a block and query group are
// taken four at a time (four consecutive reference rows): their minimum (v_min3 + v_min), one compare against `second`, and the
// eight med3 / min instructions only when some lane of the wave passes it -- a wave-uniform branch, exact by the argument above.
// Over a 4700-reference slice about a third of the quarters still take the long way (256 keys of the wave against thresholds
// that start at "nothing seen"), the rest cost 3 instructions instead of 8.  The branches would keep the compiler from spreading
// the next half's matrix instructions among this vector work (mt_mfma_half + mt_keys were one straight line), so the step is
// written out: one (FP4) or two (int8) matrix instructions in front of every quarter.
#ifndef MORB_TOP2_GATED
#define MORB_TOP2_GATED 1
#endif
template <bool FP4>
__device__ __forceinline__ void mt_step(typename Mt<FP4>::Acc (&nxt)[2], const mm_i32x4* __restrict__ tile, int a, int lane,
                                        const mm_i32x4 (&bq)[2][Mt<FP4>::KS], const typename Mt<FP4>::Acc& cinit, Top2Run& R,
                                        const typename Mt<FP4>::Acc (&acc)[2], int blk) {
#if MORB_TOP2_GATED
    using M = Mt<FP4>;
    uint32_t before = 0;
    // (all reference fragments of the half are requested up front: behind a branch the compiler no longer hoists a read above the
    // vector work in front of it, and a read issued right before its matrix instruction stalls the wave for the LDS round trip)
    mm_i32x4 af[M::KS];
#pragma unroll
    for (int ks = 0; ks < M::KS; ++ks) af[ks] = tile[(a * M::KS + ks) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int g = s >> 2, qd = s & 3;
        // matrix instructions of this slot, in the order (ks, query group) of mt_mfma_half
#pragma unroll
        for (int u = 0; u < 2 * M::KS / 8; ++u) {
            const int idx = s * (2 * M::KS / 8) + u, ks = idx >> 1, gg = idx & 1;
            nxt[gg] = M::mfma(af[ks], bq[gg][ks], ks ? nxt[gg] : cinit);
        }
        if (qd == 0) before = R.kb[g];
        uint32_t k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = M::key(acc[g], 4 * qd + j);
        const uint32_t m = min(min(min(k[0], k[1]), k[2]), k[3]);
        if (__builtin_amdgcn_ballot_w64(m < R.ks2[g]) != 0ull) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R.ks2[g] = mt_umed3(R.kb[g], R.ks2[g], k[j]);
                R.kb[g] = min(R.kb[g], k[j]);
            }
        }
        if (qd == 3) {
            M::close_block(R.kb[g], R.where[g], before, blk);
            // `second` is only ever reported as a distance: without its row bits the gate compares distances, and a key that merely
            // TIES with `second` (it cannot beat `best` then: rows ascend within a block, earlier blocks' rows are cleared) stays out
            R.ks2[g] &= ~M::ROW_MASK;
        }
    }
#else
    mt_mfma_half<FP4>(nxt, tile, a, lane, bq, cinit);
    mt_keys<FP4>(R, acc, blk, MtAll());
#endif
}

template <bool FP4>
__global__ __launch_bounds__(64 * MM_WAVES) MORB_MFMA_IN_VGPRS void k_hamming_top2_mfma(const uint32_t* __restrict__ q, int nq,
                                                                    const uint32_t* __restrict__ r, int nr, int slice_len,
                                                                    int* __restrict__ p_idx, int* __restrict__ p_best,
                                                                    int* __restrict__ p_second) {
    using M = Mt<FP4>;
    __shared__ mm_i32x4 s_tile[2][2 * M::KS * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * MM_Q_PER_BLOCK + wave * 64;

    mm_i32x4 bq[2][M::KS];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        M::queries(bq[g], w, h);
    }
    typename M::Acc cinit;  // D[m][n]: lane holds column n = lane & 31 (a query), rows m = (e & 3) + 8 (e >> 2) + 4 h (references)
    M::preset(cinit, h);

    const int s0 = blockIdx.x * slice_len, s1 = min(nr, s0 + slice_len);  // slice_len is a multiple of 64
    const int n_tiles = (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE;
    auto fetch = [&](int t) {
        const int rr = min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1);  // rows past the end repeat the last one, masked below
        return *reinterpret_cast<const uint2*>(r + (size_t)rr * 8 + wave * 2);
    };
    auto deposit = [&](int buf, uint2 w) { M::deposit(s_tile[buf], h, wave, c, w); };

    Top2Run R{{M::KEY_NONE, M::KEY_NONE}, {M::KEY_NONE, M::KEY_NONE}, {-1, -1}};
    deposit(0, fetch(0));
    uint2 nxt = fetch(1);
    __syncthreads();
    typename M::Acc acc0[2], acc1[2];
    mt_mfma_half<FP4>(acc0, s_tile[0], 0, lane, bq, cinit);
    for (int t = 0; t + 1 < n_tiles; ++t) {   // every tile but the last one is full
        const int buf = t & 1;
        mt_step<FP4>(acc1, s_tile[buf], 1, lane, bq, cinit, R, acc0, 2 * t);   // second half of tile t on the matrix cores | the keys of its first half on the vector ALU
        deposit(buf ^ 1, nxt);
        __syncthreads();
        nxt = fetch(t + 2);
        mt_step<FP4>(acc0, s_tile[buf ^ 1], 0, lane, bq, cinit, R, acc1, 2 * t + 1);   // first half of tile t + 1 | keys of the second half of tile t
    }
    {   // the last tile (its first half is in acc0): rows past the end of the slice do not count
        const int t = n_tiles - 1;
        const int valid = s1 - s0 - t * MM_R_TILE;
        mt_mfma_half<FP4>(acc1, s_tile[t & 1], 1, lane, bq, cinit);
        mt_keys<FP4>(R, acc0, 2 * t, [&](int, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h >= valid; });
        mt_keys<FP4>(R, acc1, 2 * t + 1, [&](int, int e) { return 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= valid; });
    }
    // the two half-waves hold disjoint references of the same query: full keys distance << 16 | index decide
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t mine_b = (M::distance(R.kb[g]) << 16) | (uint32_t)(R.where[g] & 0xffff), mine_s = M::distance(R.ks2[g]) << 16 | 0xffffu;
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

// A remote trailer is data from another process: every count is clamped to what is left of its rank's cap_rows rows (a
// mismatched or corrupt block can then neither run past its own block nor past the contiguous list), and the number of
// counts that had to be clamped is reported in h_counts[n_cams + 1] (orbm_cross_top2_gathered_collect turns it into an error).
// redo_mark: the first count of a block may carry ORBM_BLOCK_REDO ("this block will be shipped again": matcher_internal.h) -- only the
// front end's own exchange ships such blocks; for anybody else's buffer (orbm_cross_top2_gathered) the bit is a count out of range.
__device__ __forceinline__ int repack_count(const int* __restrict__ tail, int c, int& room, int& bad, int redo_mark) {
    int raw = tail[c];
    if (redo_mark && c == 0 && raw >= 0) raw &= ~ORBM_BLOCK_REDO;
    const int n = min(max(raw, 0), room);
    bad += (n != raw);
    room -= n;
    return n;
}

__global__ __launch_bounds__(256) void k_repack_gathered(const uint8_t* __restrict__ gathered, int world, size_t block_bytes,
                                                         int cap_rows, int cams_per_rank, int rank, uint4* __restrict__ dst,
                                                         int* __restrict__ cam_start, int* __restrict__ range,
                                                         int* __restrict__ h_counts, int redo_mark) {
    const int r = blockIdx.y;
    int goff = 0, n_r = 0, own_off = 0, own_n = 0, total = 0, bad = 0;
    for (int rr = 0; rr < world; ++rr) {
        const int* tail = reinterpret_cast<const int*>(gathered + (size_t)rr * block_bytes + (size_t)cap_rows * 32);
        int nr = 0, room = cap_rows;
        for (int c = 0; c < cams_per_rank; ++c) nr += repack_count(tail, c, room, bad, redo_mark);
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
                const int n = repack_count(tail, c, room, ignore, redo_mark);
                cam_start[rr * cams_per_rank + c] = run;
                h_counts[rr * cams_per_rank + c] = n;
                run += n;
            }
        }
        cam_start[world * cams_per_rank] = run;
        range[0] = total; range[1] = own_off; range[2] = own_n;
        h_counts[world * cams_per_rank] = own_n;
        h_counts[world * cams_per_rank + 1] = bad;
        int redo = 0;   // ranks whose block says it will be shipped again (every rank computes the same number from the same blocks)
        for (int rr = 0; rr < world; ++rr) {
            const int first = *reinterpret_cast<const int*>(gathered + (size_t)rr * block_bytes + (size_t)cap_rows * 32);
            redo += redo_mark && first >= 0 && (first & ORBM_BLOCK_REDO) != 0;
        }
        h_counts[world * cams_per_rank + 2] = redo;
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
    if ((int)blockIdx.x * 64 >= nq) return;
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

// The same search on the matrix cores: k_hamming_top2_mfma's pipelined walk with queries and references taken from ONE
// descriptor list and the rows of a query's OWN camera left out.  Round 3: nothing is masked any more --
//   * a workgroup's 256 queries come from ONE camera (query blocks are dealt camera by camera: camera c holds
//     ceil(its queries / 256) blocks, the last one partly filled), so the whole workgroup has the same own segment [seg0, seg1);
//   * the references it walks are the concatenation of the OTHER cameras: position v of that list is row v of the descriptor
//     list in front of the own segment and row v + (seg1 - seg0) behind it.  Only the loads know (every lane fetches its own
//     row); slices, tiles and keys live in the list of the others, which is also what the reported index is defined in.
// So the kernel body IS the generic one (166 registers, three waves per SIMD); rounds 1-2 skipped own-camera tiles and
// masked the rows of boundary tiles per key, which cost a second code path and 230 registers.
// Counts known only on the device come through d_range = {features, first query, queries}; the launch is then sized for the
// capacity (grid.y = cross_query_blocks(capacity, cameras)), slices beyond the references produce (256, 256, -1) partials and
// query blocks beyond the queries return at once.  grid.x = reference slices (partials for k_top2_merge when > 1).
// bx = reference slice, by = query block (camera-major)
template <bool FP4>
__device__ __forceinline__ void cross_top2_mfma_body(mm_i32x4 (&s_tile)[2][2 * Mt<FP4>::KS * 64], const uint32_t* __restrict__ desc, int n_total,
                                                     const int* __restrict__ cam_start, int n_cams, int q_off, int nq, int slice_len,
                                                     int* __restrict__ p_idx, int* __restrict__ p_best, int* __restrict__ p_second,
                                                     const int* __restrict__ d_range, const int bx, const int by) {
    using M = Mt<FP4>;
    if (d_range) { n_total = d_range[0]; q_off = d_range[1]; nq = d_range[2]; }
    // block `by` -> (camera, first query of the block, end of the camera's queries); the queries are the features [q_off, q_off + nq)
    int seg0 = 0, seg1 = 0, qb0 = 0, qb1 = -1;
    for (int cam = 0, before = 0; cam < n_cams; ++cam) {
        const int c0 = cam_start[cam], c1 = cam_start[cam + 1];
        const int lo = max(c0, q_off), hi = min(c1, q_off + nq);
        const int nb = hi > lo ? (hi - lo + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK : 0;
        if (by < before + nb) { seg0 = c0; seg1 = c1; qb0 = lo + (by - before) * MM_Q_PER_BLOCK; qb1 = hi; break; }
        before += nb;
    }
    if (qb1 < 0) return;   // (uniform over the workgroup: a block beyond the queries)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = qb0 + wave * 64;                 // first query of the wave (index in the descriptor list)
    const int own = seg1 - seg0, nr = n_total - own;   // nr references: the other cameras

    mm_i32x4 bq[2][M::KS];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, qb1 - 1);
        const uint4* p = reinterpret_cast<const uint4*>(desc + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        M::queries(bq[g], w, h);
    }
    typename M::Acc cinit;
    M::preset(cinit, h);

    const int s0 = bx * slice_len, s1 = min(nr, s0 + slice_len);  // slice_len is a multiple of 64
    const int n_tiles = s1 > s0 ? (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE : 0;
    auto fetch = [&](int t) {
        const int v = max(0, min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1));   // rows past the end repeat the last one, masked below
        const int rr = v < seg0 ? v : v + own;
        return *reinterpret_cast<const uint2*>(desc + (size_t)min(rr, n_total - 1) * 8 + wave * 2);
    };
    auto deposit = [&](int buf, uint2 w) { M::deposit(s_tile[buf], h, wave, c, w); };
    Top2Run R{{M::KEY_NONE, M::KEY_NONE}, {M::KEY_NONE, M::KEY_NONE}, {-1, -1}};
    if (n_tiles > 0) {
        deposit(0, fetch(0));
        uint2 nxt = fetch(1);
        __syncthreads();
        typename M::Acc acc0[2], acc1[2];
        mt_mfma_half<FP4>(acc0, s_tile[0], 0, lane, bq, cinit);
        for (int t = 0; t + 1 < n_tiles; ++t) {   // every tile but the last one is full
            const int buf = t & 1;
            mt_step<FP4>(acc1, s_tile[buf], 1, lane, bq, cinit, R, acc0, 2 * t);   // second half of tile t on the matrix cores | the keys of its first half on the vector ALU
            deposit(buf ^ 1, nxt);
            __syncthreads();
            nxt = fetch(t + 2);
            mt_step<FP4>(acc0, s_tile[buf ^ 1], 0, lane, bq, cinit, R, acc1, 2 * t + 1);   // first half of tile t + 1 | keys of the second half of tile t
        }
        {   // the last tile (its first half is in acc0): rows past the end of the slice do not count
            const int t = n_tiles - 1;
            const int valid = s1 - s0 - t * MM_R_TILE;
            mt_mfma_half<FP4>(acc1, s_tile[t & 1], 1, lane, bq, cinit);
            mt_keys<FP4>(R, acc0, 2 * t, [&](int, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h >= valid; });
            mt_keys<FP4>(R, acc1, 2 * t + 1, [&](int, int e) { return 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= valid; });
        }
    }
    // the two half-waves hold disjoint references of the same query: full keys distance << 16 | index decide
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t mine_b = (M::distance(R.kb[g]) << 16) | (uint32_t)(R.where[g] & 0xffff), mine_s = M::distance(R.ks2[g]) << 16 | 0xffffu;
        const uint32_t ob = (uint32_t)__shfl_xor((int)mine_b, 32), os = (uint32_t)__shfl_xor((int)mine_s, 32);
        const uint32_t nb = min(mine_b, ob), ns = min(max(mine_b, ob), min(mine_s, os));
        const int qi = q0 + g * 32 + c;
        if (h == 0 && qi < qb1) {
            const size_t o = (size_t)bx * nq + (qi - q_off);
            const int best = (int)(nb >> 16);
            p_best[o] = best;
            p_idx[o] = best < 256 ? s0 + (int)(nb & 0xffffu) : -1;   // position among the other cameras
            p_second[o] = (int)min(ns >> 16, 256u);
        }
    }
}

// query blocks a launch over nq queries from n_cams cameras may need: every camera can end with a partly filled block
__host__ __device__ inline int cross_query_blocks(int nq, int n_cams) { return (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK + n_cams; }

template <bool FP4>
__global__ __launch_bounds__(64 * MM_WAVES) MORB_MFMA_IN_VGPRS void k_cross_top2_mfma(const uint32_t* __restrict__ desc, int n_total,
                                                                  const int* __restrict__ cam_start, int n_cams, int q_off, int nq,
                                                                  int slice_len, int* __restrict__ p_idx, int* __restrict__ p_best,
                                                                  int* __restrict__ p_second, const int* __restrict__ d_range) {
    __shared__ mm_i32x4 s_tile[2][2 * Mt<FP4>::KS * 64];
    cross_top2_mfma_body<FP4>(s_tile, desc, n_total, cam_start, n_cams, q_off, nq, slice_len, p_idx, p_best, p_second, d_range,
                         (int)blockIdx.x, (int)blockIdx.y);
}

// ---- the projection kernel, the camera-pair top-2 and the result mirror of an isolated orbf_step in ONE launch ------------
// Three kinds of workgroup, independent of each other: [0, n_cross) slices x query blocks of the top-2 (the longest: first),
// [n_cross, n_cross + n_project) four queries each of k_project, the rest copies the frame into the pinned result mirrors.
// The slices' partials are merged by k_top2_merge right behind this launch (an in-kernel merge by the last slice to arrive was
// tried: the two device-scope fences it needs write the L2 back and cost ~7 us each).  Everything the top-2 and the mirror
// produce is therefore complete before the resolve kernel starts: a host that watches the resolve's tagged result words
// arrive may take those results as well.
struct SideArgs {
    const uint32_t* desc; int n_total; const int* cam_start; int n_cams; int slice_len; int S;
    int *p_idx, *p_best, *p_second; const int* d_range;
    int n_cross, n_project;
    int with_mirror;
    morb::MirrorJob mirror;
};
template <bool FP4>
__global__ __launch_bounds__(64 * MM_WAVES) MORB_MFMA_IN_VGPRS void k_project_side(morb::ProjectArgs P, SideArgs X) {
    __shared__ mm_i32x4 s_tile[2][2 * Mt<FP4>::KS * 64];
    const int b = blockIdx.x;
    if (b < X.n_cross) {
        const int by = b / X.S, bx = b - by * X.S;
        cross_top2_mfma_body<FP4>(s_tile, X.desc, X.n_total, X.cam_start, X.n_cams, 0, X.n_total, X.slice_len, X.p_idx, X.p_best,
                             X.p_second, X.d_range, bx, by);
    } else if (b < X.n_cross + X.n_project) {
        const int qi = __builtin_amdgcn_readfirstlane((b - X.n_cross) * 4 + (int)(threadIdx.x >> 6));
        if (qi < P.nq) morb::project_wave(P, qi, threadIdx.x & 63);
    } else if (X.with_mirror) {
        const int nb = (int)gridDim.x - X.n_cross - X.n_project;
        morb::mirror_rows(X.mirror, (b - X.n_cross - X.n_project) * 256 + (int)threadIdx.x, nb * 256);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// Plan of an exhaustive top-2: matrix-core kernel (256 queries x one reference slice per workgroup; slices are multiples
// of 64 references and at most 65536 long: 16-bit indices in the keys) when both sides have at least a tile, else the
// one-query-per-lane kernel.  S = number of reference slices (> 1 needs scratch for the partials).
struct Top2Plan { bool mfma; int S; int slice_len; };
// orbm_use_fp4_top2(0) / MORB_TOP2_FP4=0: the matrix-core top-2 kernels run their int8 form (A/B; same results)
std::atomic<int> g_top2_fp4{-1};      // -1 = environment default
static bool top2_fp4() {
    static const bool env_on = [] { const char* e = getenv("MORB_TOP2_FP4"); return !(e && atoi(e) == 0); }();
    const int forced = g_top2_fp4.load(std::memory_order_relaxed);
    return forced < 0 ? env_on : forced != 0;
}
std::atomic<int> g_matrix_cores{-1};  // orbm_use_matrix_cores: -1 = environment default

Top2Plan top2_plan(int nq, int nr, bool have_scratch = true, int wgs_per_cu = 3) {
    const int forced = g_matrix_cores.load(std::memory_order_relaxed);   // (orbm_use_matrix_cores(0): the popcount kernels)
    Top2Plan p{false, 1, nr};
    if (forced != 0 && nq >= 64 && nr >= MM_R_TILE) {
        // wgs_per_cu workgroups are resident per CU (166 registers: three; the 230-register build of the camera-pair kernel:
        // two): a launch runs in ceil(workgroups / slots) rounds of (tiles per slice + ~2) steps; take the slice count that
        // minimises the product.
        static const int n_cus = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return std::max(1, cus);
        }();
        const int slots = n_cus * wgs_per_cu;
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

int top2_slices(int nq, int nr);
// slices any variant of the camera-pair kernels may ask for (scratch sizing: the choice can change at run time)
static int cross_slices_max(int nq, int n) {
    return std::max(std::max(top2_plan(nq, n, true, 2).S, top2_plan(nq, n, true, 3).S), top2_slices(nq, n));
}

int launch_top2(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_bi, int32_t* d_bd, int32_t* d_sd,
                void* d_scratch, const Top2Plan& plan, hipStream_t st) {
    const int S = plan.S;
    int* p = (int*)d_scratch;
    int *p_idx = S > 1 ? p : d_bi, *p_best = S > 1 ? p + (size_t)S * nq : d_bd, *p_second = S > 1 ? p + 2 * (size_t)S * nq : d_sd;
    if (plan.mfma && top2_fp4())
        hipLaunchKernelGGL(k_hamming_top2_mfma<true>, dim3(S, (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK), dim3(64 * MM_WAVES), 0, st,
                           (const uint32_t*)d_q, nq, (const uint32_t*)d_r, nr, plan.slice_len, p_idx, p_best, p_second);
    else if (plan.mfma)
        hipLaunchKernelGGL(k_hamming_top2_mfma<false>, dim3(S, (nq + MM_Q_PER_BLOCK - 1) / MM_Q_PER_BLOCK), dim3(64 * MM_WAVES), 0, st,
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
    int q_per_block = (int)std::min<long long>(256, std::max<long long>(8, ((long long)nq * tiles + 4095) / 4096));
    q_per_block = (q_per_block + 1) & ~1;
    dim3 grid((tiles + MAT_WAVES - 1) / MAT_WAVES, (nq + q_per_block - 1) / q_per_block);
    const bool aligned = (nr % 8 == 0) && (((uintptr_t)d_out & 15) == 0) && nr >= MAT_REFS_PER_WAVE;
    // Matrix-core path: rows of 16-byte chunks, enough queries to fill 64-query waves.  orbm_use_matrix_cores(0) keeps the VALU kernel.
    const int forced = g_matrix_cores.load(std::memory_order_relaxed);
    if (forced != 0 && (nr % 8 == 0) && (((uintptr_t)d_out & 15) == 0) && nq >= 64 && nr >= MM_R_TILE && (((uintptr_t)d_r & 7) == 0)) {
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

int CrossOut::reserve(int nq, int n) {
        const int S = cross_slices_max(nq, n);   // (room for either form of the kernel)
        int rc;
        if ((rc = scratch.reserve(std::max<size_t>((size_t)3 * S * nq * 4, 16))) || (rc = i.reserve(nq)) || (rc = b.reserve(nq)) ||
            (rc = s.reserve(nq)))
            return rc;
        return ORB_OK;
}

int morb::cross_enqueue_to(hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off, int nq,
                            const int* d_n, int* o_idx, int* o_best, int* o_second, void* scratch) {
    if (nq == 0) return ORB_OK;
    const int qblocks = (nq + 63) / 64;
    const Top2Plan plan = top2_plan(nq, n);   // (nq, n may be capacities: the kernels take the counts from d_n then)
    const int S = plan.S;
    if (plan.mfma) {   // matrix-core form (default from one tile of work on; orbm_use_matrix_cores(0) / MORB_TOP2_MFMA=0: popcount form)
        int* p = (int*)scratch;
        int *p_idx = S > 1 ? p : o_idx, *p_best = S > 1 ? p + (size_t)S * nq : o_best, *p_second = S > 1 ? p + 2 * (size_t)S * nq : o_second;
        if (top2_fp4())
            hipLaunchKernelGGL(k_cross_top2_mfma<true>, dim3(S, cross_query_blocks(nq, n_cams)), dim3(64 * MM_WAVES), 0, st,
                               (const uint32_t*)d_desc, n, d_cam_start, n_cams, q_off, nq, plan.slice_len, p_idx, p_best, p_second, d_n);
        else
            hipLaunchKernelGGL(k_cross_top2_mfma<false>, dim3(S, cross_query_blocks(nq, n_cams)), dim3(64 * MM_WAVES), 0, st,
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

int morb::cross_enqueue(orbm_matcher* m, hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off,
                         int nq, const int* d_n) {
    if (nq == 0) return ORB_OK;
    const int S = cross_slices_max(nq, n);   // (room for either form: the choice can change at run time)
    int rc;
    if ((rc = m->d_cscratch.reserve(std::max<size_t>((size_t)3 * S * nq * 4, 16))) || (rc = m->h_c0.reserve(nq)) ||
        (rc = m->h_c1.reserve(nq)) || (rc = m->h_c2.reserve(nq)))
        return rc;
    return cross_enqueue_to(st, d_desc, n, d_cam_start, n_cams, q_off, nq, d_n, m->h_c0.dp, m->h_c1.dp, m->h_c2.dp, m->d_cscratch.p);
}


bool morb::side_fusable(int nq, int n) {
    if (nq <= 0 || n <= 0) return false;
    const Top2Plan plan = top2_plan(nq, n);
    return plan.mfma && plan.S >= 1 && plan.S <= 64;
}

int morb::side_reserve(orbm_matcher* m, int nq, int n) {
    const int S = cross_slices_max(nq, n);
    int rc;
    if ((rc = m->d_cscratch.reserve(std::max<size_t>((size_t)3 * S * nq * 4, 16))) || (rc = m->h_c0.reserve(nq)) ||
        (rc = m->h_c1.reserve(nq)) || (rc = m->h_c2.reserve(nq)))
        return rc;
    return ORB_OK;
}

int morb::launch_project_side(hipStream_t st, const morb::ProjectArgs& P, const SideJob& J, MergeJob* defer_merge) {
    const int nq = J.n;   // every feature of the frame is a query of the camera-pair top-2 (capacity; the count comes from d_range)
    const Top2Plan plan = top2_plan(nq, J.n);
    MORB_ARG(plan.mfma && J.d_range && J.scratch);
    SideArgs X;
    memset(&X, 0, sizeof(X));
    X.desc = (const uint32_t*)J.d_desc; X.n_total = J.n; X.cam_start = J.d_cam_start; X.n_cams = J.n_cams;
    X.slice_len = plan.slice_len; X.S = plan.S;
    int* p = (int*)J.scratch;
    const int S = plan.S;
    X.p_idx = S > 1 ? p : J.o_idx; X.p_best = S > 1 ? p + (size_t)S * nq : J.o_best; X.p_second = S > 1 ? p + 2 * (size_t)S * nq : J.o_second;
    X.d_range = J.d_range;
    X.n_cross = plan.S * cross_query_blocks(nq, J.n_cams);
    X.n_project = (P.nq + 3) / 4;
    X.with_mirror = J.with_mirror ? 1 : 0;
    if (J.with_mirror) X.mirror = J.mirror;
    const int n_mirror = J.with_mirror ? std::min(64, (J.mirror.n_host * 8 + 255) / 256) : 0;
    if (top2_fp4()) hipLaunchKernelGGL(k_project_side<true>, dim3(X.n_cross + X.n_project + n_mirror), dim3(64 * MM_WAVES), 0, st, P, X);
    else hipLaunchKernelGGL(k_project_side<false>, dim3(X.n_cross + X.n_project + n_mirror), dim3(64 * MM_WAVES), 0, st, P, X);
    // the merge of the slice partials: a launch of its own, or -- when the caller's next launch on this stream can carry it (the
    // resolve of an isolated step: one kernel and one kernel boundary less on the step's chain) -- handed back as a job
    if (S > 1 && defer_merge) *defer_merge = MergeJob{X.p_idx, X.p_best, X.p_second, S, nq, J.o_idx, J.o_best, J.o_second, J.d_range, nullptr, 0u};
    else if (S > 1)
        hipLaunchKernelGGL(k_top2_merge, dim3((nq + 255) / 256), dim3(256), 0, st, X.p_idx, X.p_best, X.p_second, S, nq, J.o_idx, J.o_best,
                           J.o_second, J.d_range);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

// (a deferred merge whose carrier did not come after all)
int morb::launch_merge(hipStream_t st, const MergeJob& M) {
    hipLaunchKernelGGL(k_top2_merge, dim3((M.nq + 255) / 256), dim3(256), 0, st, M.p_idx, M.p_best, M.p_second, M.S, M.nq, M.o_idx, M.o_best, M.o_second, M.d_range);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

int orbm_use_matrix_cores(int on) { return g_matrix_cores.exchange(on < 0 ? -1 : (on ? 1 : 0)); }
int orbm_use_fp4_top2(int on) { return g_top2_fp4.exchange(on < 0 ? -1 : (on ? 1 : 0)); }

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
    if ((rc = m->d_r.reserve((size_t)n_cap * 32)) || (rc = m->d_gstart.reserve(n_cams + 1 + 4)) || (rc = m->h_gcnt.reserve(n_cams + 3)))
        return rc;
    hipStream_t sd = morb::side_stream(m);
    if (!sd) return ORB_E_HIP;
    if (wait_after) {  // the gathered buffer is produced on another stream (the collective's; NULL = the default stream)
        MORB_HIP(hipEventRecord(m->ev_fork, (hipStream_t)after_stream));
        MORB_HIP(hipStreamWaitEvent(sd, m->ev_fork, 0));
    }
    int* d_cam_start = m->d_gstart.p;
    int* d_range = m->d_gstart.p + n_cams + 1;
    hipLaunchKernelGGL(k_repack_gathered, dim3((2 * cap_rows + 255) / 256, world), dim3(256), 0, sd, d_gathered, world, block_bytes, cap_rows,
                       cams_per_rank, rank, (uint4*)m->d_r.p, d_cam_start, d_range, m->h_gcnt.dp, 0);
    MORB_HIP(hipGetLastError());
    // the launch is sized for the capacity (cap_rows queries against world * cap_rows features); the counts come from HBM
    if ((rc = cross_enqueue(m, sd, m->d_r.p, n_cap, d_cam_start, n_cams, 0, cap_rows, d_range))) return rc;
    MORB_HIP(hipEventRecord(m->ev_join, sd));
    MORB_HIP(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    m->gathered_cams = n_cams;
    m->foreign_work = true;
    return ORB_OK;
}

// The same two launches on a stream and buffers of the caller's choosing (frontend.hip: a step's exchange runs at the tail of its
// extraction chain, every step in flight with buffers of its own): d_list >= world * cap_rows rows, d_gstart >= n_cams + 5 words,
// h_gcnt_dp (device pointer of mapped pinned memory) n_cams + 3 words -- counts, own queries, clamped counts, blocks marked "redo".
int morb::gathered_enqueue_to(hipStream_t st, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows, int cams_per_rank,
                              int rank, uint8_t* d_list, int* d_gstart, int* h_gcnt_dp, CrossOut& out) {
    const int n_cams = world * cams_per_rank, n_cap = world * cap_rows;
    int* d_range = d_gstart + n_cams + 1;
    hipLaunchKernelGGL(k_repack_gathered, dim3((2 * cap_rows + 255) / 256, world), dim3(256), 0, st, d_gathered, world, block_bytes, cap_rows,
                       cams_per_rank, rank, (uint4*)d_list, d_gstart, d_range, h_gcnt_dp, 1);
    MORB_HIP(hipGetLastError());
    return cross_enqueue_to(st, d_list, n_cap, d_gstart, n_cams, 0, cap_rows, d_range, out.i.dp, out.b.dp, out.s.dp, out.scratch.p);
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
    if (m) { hipStream_t sd = morb::side_stream(m); if (!sd) return ORB_E_HIP; MORB_HIP(hipEventRecord(m->ev_q, m->stream)); MORB_HIP(hipStreamWaitEvent(sd, m->ev_q, 0)); }
    if ((rc = orbm_cross_top2_gathered_enqueue(m, d_gathered, world, block_bytes, cap_rows, cams_per_rank, rank, nullptr, 0))) return rc;
    MORB_HIP(hipStreamSynchronize(m->stream));
    return orbm_cross_top2_gathered_collect(m, best_idx, best_dist, second_dist, counts_out, nq_out);
}


