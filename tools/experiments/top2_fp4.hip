// Experiment (round 3): the exhaustive top-2 with the descriptors' bits as FP4 (E2M1) values on gfx950's block-scaled matrix instruction
// (v_mfma_scale_f32_32x32x64_f8f6f4: K = 64 per instruction at twice the int8 rate) against the int8 form of the product.  A bit is +-4
// (references: set = +4; queries: set = -4), so a product is -16 where the bits agree and +16 where they differ, a dot product over 256 bits
// is 32 * distance - 4096, and with the accumulators preset to 4096 + row the f32 result IS the sort key 32 * distance + row -- exact
// (integers below 2^24), positive, so its bit pattern orders like the number and the integer min / med3 of the int8 form work unchanged.
// The order of the K elements inside an instruction does not matter as long as both operands use the same one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o top2_fp4 top2_fp4.hip && ./top2_fp4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define VALU_PER_GAP 5
using mm_i32x4 = __attribute__((ext_vector_type(4))) int;
using mm_i32x16 = __attribute__((ext_vector_type(16))) int;
constexpr int MM_WAVES = 4;
constexpr int MM_Q_PER_BLOCK = 64 * MM_WAVES;
constexpr int MM_R_TILE = 64;

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
__device__ __forceinline__ mm_i32x4 mt_expand16(uint32_t bits) {
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

template <int MODE>   // 0: no scheduling hints, 1: sched_group_barrier interleave
__global__ __launch_bounds__(64 * MM_WAVES) void k_pipe(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ r, int nr,
                                                        int slice_len, int* __restrict__ p_idx, int* __restrict__ p_best,
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
    mm_i32x16 cinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[e] = 8192 + (e & 3) + 8 * (e >> 2) + 4 * h;
    const int s0 = blockIdx.x * slice_len, s1 = min(nr, s0 + slice_len);
    const int n_tiles = (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE;
    auto fetch = [&](int t) {
        const int rr = min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1);
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
    int where[2] = {-1, -1};
    // half a = rows [32 a, 32 a + 32) of tile t from LDS buffer buf into acc[0..1]
    auto mfma_half = [&](mm_i32x16 (&acc)[2], int buf, int a) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const mm_i32x4 af = s_tile[buf][(a * 8 + ks) * 64 + lane];
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bq[0][ks], ks ? acc[0] : cinit, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bq[1][ks], ks ? acc[1] : cinit, 0, 0, 0);
        }
    };
    auto keys_full = [&](const mm_i32x16 (&acc)[2], int blk) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t before = kb[g];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t key = (uint32_t)acc[g][e];
                ks2[g] = mt_umed3(kb[g], ks2[g], key);
                kb[g] = min(kb[g], key);
            }
            where[g] = kb[g] != before ? ((blk << 5) | (int)(kb[g] & 31u)) : where[g];
            kb[g] &= ~63u;
        }
    };
    auto keys_masked = [&](const mm_i32x16 (&acc)[2], int blk, int a, int valid) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t before = kb[g];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int local = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const uint32_t key = local < valid ? (uint32_t)acc[g][e] : KEY_NONE;
                ks2[g] = mt_umed3(kb[g], ks2[g], key);
                kb[g] = min(kb[g], key);
            }
            where[g] = kb[g] != before ? ((blk << 5) | (int)(kb[g] & 31u)) : where[g];
            kb[g] &= ~63u;
        }
    };
    deposit(0, fetch(0));
    uint2 nxt = fetch(1);
    __syncthreads();
    mm_i32x16 acc0[2], acc1[2];
    mfma_half(acc0, 0, 0);
    for (int t = 0; t + 1 < n_tiles; ++t) {     // tiles 0 .. n_tiles-2 are full
        const int buf = t & 1;
        // P1: second half of tile t on the matrix cores | keys of its first half, next tile into LDS
        mfma_half(acc1, buf, 1);
        keys_full(acc0, 2 * t);
        deposit(buf ^ 1, nxt);
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GAP + 4, 0);
            }
        }
        __syncthreads();
        // P2: first half of tile t + 1 | keys of the second half of tile t
        nxt = fetch(t + 2);
        mfma_half(acc0, buf ^ 1, 0);
        keys_full(acc1, 2 * t + 1);
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GAP, 0);
            }
        }
    }
    {   // last tile (may be partial): its first half is in acc0 already
        const int t = n_tiles - 1, buf = t & 1;
        const int valid = s1 - s0 - t * MM_R_TILE;
        mfma_half(acc1, buf, 1);
        keys_masked(acc0, 2 * t, 0, valid);
        keys_masked(acc1, 2 * t + 1, 1, valid);
    }
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


using mm_i32x8 = __attribute__((ext_vector_type(8))) int;
using mm_f32x16 = __attribute__((ext_vector_type(16))) float;

// 32 descriptor bits -> 32 FP4 values (4 dwords): two bits select one byte of the pool {lo nibble = bit 0, hi nibble = bit 1}
template <bool QUERY>
__device__ __forceinline__ mm_i32x4 f4_expand32(uint32_t bits) {
    // reference: set = +4 (0x6), clear = -4 (0xE); query: the other way round.  Output dword j holds the 2-bit fields j of the word's four
    // bytes (which K slot a bit lands in is free as long as both operands agree): one shift, one mask, one byte permute per dword
    constexpr uint32_t POOL = QUERY ? 0xEEE66E66u : 0x666EE6EEu;   // byte f = nib(bit1) << 4 | nib(bit0) for f = bit1 bit0
    mm_i32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (int)__builtin_amdgcn_perm(0u, POOL, (bits >> (2 * j)) & 0x03030303u);
    return v;
}

template <int SCALE_MODE>
__device__ __forceinline__ mm_f32x16 f4_mfma(mm_i32x4 a, mm_i32x4 b, mm_f32x16 c) {
    const mm_i32x8 a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    if (SCALE_MODE == 0) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 127, 0, 127);
}

template <int SCALE_MODE, bool SKIP = false>
__global__ __launch_bounds__(64 * MM_WAVES) void k_fp4(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ r, int nr,
                                                       int slice_len, int* __restrict__ p_idx, int* __restrict__ p_best,
                                                       int* __restrict__ p_second) {
    __shared__ mm_i32x4 s_tile[2][2 * 4 * 64];   // [buffer][(half * 4 + ks) * 64 + lane]: 8 KB per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * MM_Q_PER_BLOCK + wave * 64;
    mm_i32x4 bq[2][4];   // B fragments: lane (c, h) holds word 2 ks + h of query c of group g
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qi = min(q0 + g * 32 + c, nq - 1);
        const uint4* p = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
        const uint4 lo = p[0], hi = p[1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[g][ks] = f4_expand32<true>(h ? w[2 * ks + 1] : w[2 * ks]);
    }
    mm_f32x16 cinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[e] = (float)(4096 + (e & 3) + 8 * (e >> 2) + 4 * h);
    const int s0 = blockIdx.x * slice_len, s1 = min(nr, s0 + slice_len);
    const int n_tiles = (s1 - s0 + MM_R_TILE - 1) / MM_R_TILE;
    auto fetch = [&](int t) {
        const int rr = min(s0 + min(t, n_tiles - 1) * MM_R_TILE + lane, nr - 1);
        return *reinterpret_cast<const uint2*>(r + (size_t)rr * 8 + wave * 2);
    };
    auto deposit = [&](int buf, uint2 w) {   // this thread: reference row `lane` of the tile (half h, row c), words 2 wave and 2 wave + 1
        mm_i32x4* base = &s_tile[buf][(h * 4 + wave) * 64 + c];
        base[0] = f4_expand32<false>(w.x);
        base[32] = f4_expand32<false>(w.y);
    };
    constexpr uint32_t KEY_NONE = 0x46000000u;   // 8192.0f = 32 * 256
    uint32_t kb[2] = {KEY_NONE, KEY_NONE}, ks2[2] = {KEY_NONE, KEY_NONE};
    int where[2] = {-1, -1};
    auto mfma_half = [&](mm_f32x16 (&acc)[2], int buf, int a) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const mm_i32x4 af = s_tile[buf][(a * 4 + ks) * 64 + lane];
            acc[0] = f4_mfma<SCALE_MODE>(af, bq[0][ks], ks ? acc[0] : cinit);
            acc[1] = f4_mfma<SCALE_MODE>(af, bq[1][ks], ks ? acc[1] : cinit);
        }
    };
    auto keys = [&](const mm_f32x16 (&acc)[2], int blk, int a, int valid) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (SKIP && valid >= 64) {
                // does any key of this block beat some lane's second best?  (min3 tree over the 16 keys, one ballot): if not, nothing changes
                uint32_t m0 = min(min(__float_as_uint(acc[g][0]), __float_as_uint(acc[g][1])), __float_as_uint(acc[g][2]));
                uint32_t m1 = min(min(__float_as_uint(acc[g][3]), __float_as_uint(acc[g][4])), __float_as_uint(acc[g][5]));
                uint32_t m2 = min(min(__float_as_uint(acc[g][6]), __float_as_uint(acc[g][7])), __float_as_uint(acc[g][8]));
                uint32_t m3 = min(min(__float_as_uint(acc[g][9]), __float_as_uint(acc[g][10])), __float_as_uint(acc[g][11]));
                uint32_t m4 = min(min(__float_as_uint(acc[g][12]), __float_as_uint(acc[g][13])), __float_as_uint(acc[g][14]));
                m0 = min(min(m0, m1), m2); m3 = min(min(m3, m4), __float_as_uint(acc[g][15]));
                if (__ballot(min(m0, m3) < ks2[g]) == 0) continue;
            }
            const uint32_t before = kb[g];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int local = a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const uint32_t key = local < valid ? __float_as_uint(acc[g][e]) : KEY_NONE;
                ks2[g] = mt_umed3(kb[g], ks2[g], key);
                kb[g] = min(kb[g], key);
            }
            const int kbi = (int)__uint_as_float(kb[g]);   // 32 * distance + row
            where[g] = kb[g] != before ? ((blk << 5) | (kbi & 31)) : where[g];
            kb[g] = __float_as_uint((float)(kbi & ~31));
        }
    };
    deposit(0, fetch(0));
    uint2 nxt = fetch(1);
    __syncthreads();
    mm_f32x16 acc0[2], acc1[2];
    mfma_half(acc0, 0, 0);
    for (int t = 0; t + 1 < n_tiles; ++t) {
        const int buf = t & 1;
        mfma_half(acc1, buf, 1);
        keys(acc0, 2 * t, 0, 64);
        deposit(buf ^ 1, nxt);
        __syncthreads();
        nxt = fetch(t + 2);
        mfma_half(acc0, buf ^ 1, 0);
        keys(acc1, 2 * t + 1, 1, 64);
    }
    {
        const int t = n_tiles - 1, buf = t & 1;
        const int valid = s1 - s0 - t * MM_R_TILE;
        mfma_half(acc1, buf, 1);
        keys(acc0, 2 * t, 0, valid);
        keys(acc1, 2 * t + 1, 1, valid);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t db = (uint32_t)(int)__uint_as_float(kb[g]) >> 5, ds = (uint32_t)(int)__uint_as_float(ks2[g]) >> 5;
        const uint32_t mine_b = (db << 16) | (uint32_t)(where[g] & 0xffff), mine_s = (ds << 16) | 0xffffu;
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint32_t h32(uint64_t x) { x &= 0xffffffffu; x ^= x >> 16; x = (x * 0x85EBCA6Bu) & 0xffffffffu; x ^= x >> 13; x = (x * 0xC2B2AE35u) & 0xffffffffu; x ^= x >> 16; return (uint32_t)x; }

template <typename K>
static float run(K kern, const char* name, int S, int qb, const uint32_t* dq, int nq, const uint32_t* dr, int nr, int len, int* pi, int* pb, int* ps,
                 int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(S, qb), dim3(256), 0, 0, dq, nq, dr, nr, len, pi, pb, ps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(S, qb), dim3(256), 0, 0, dq, nq, dr, nr, len, pi, pb, ps);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, pairs = (double)nq * nr;
    printf("%-28s %8.1f us  %7.0f int8 TOP/s  (%.3f of 5000, %.3f of the 4404 microbenchmark ceiling)\n", name, us, 512.0 * pairs / (us * 1e-6) / 1e12,
           512.0 * pairs / (us * 1e-6) / 1e12 / 5000.0, 512.0 * pairs / (us * 1e-6) / 1e12 / 4404.0);
    return (float)us;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32000, S = argc > 2 ? atoi(argv[2]) : 8, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int nq = n, nr = n;
    std::vector<uint32_t> hq((size_t)nq * 8), hr((size_t)nr * 8);
    for (size_t i = 0; i < hr.size(); ++i) hr[i] = h32(i * 2654435761u + 17);
    for (int i = 0; i < nq; ++i)
        for (int w = 0; w < 8; ++w) {
            uint32_t v = hr[(size_t)((i * 7919) % nr) * 8 + w];
            if (i & 1) v = h32((uint64_t)i * 8 + w + 0x5bd1e995u);
            else v ^= h32((uint64_t)i * 8 + w + 99) & h32((uint64_t)i * 8 + w + 177) & h32((uint64_t)i * 8 + w + 311);   // ~1/8 of the bits flipped
            hq[(size_t)i * 8 + w] = v;
        }
    uint32_t *dq, *dr; int *pi[2], *pb[2], *ps[2];
    CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dr, hr.size() * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    const int len = ((nr + S - 1) / S + 63) / 64 * 64, s_eff = (nr + len - 1) / len, qb = (nq + 255) / 256;
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&pi[k], (size_t)s_eff * nq * 4)); CK(hipMalloc(&pb[k], (size_t)s_eff * nq * 4)); CK(hipMalloc(&ps[k], (size_t)s_eff * nq * 4)); }
    printf("Q = R = %d, %d slices of %d references, %d workgroups\n", n, s_eff, len, s_eff * qb);
    run(k_pipe<0>, "int8 form (product)", s_eff, qb, dq, nq, dr, nr, len, pi[0], pb[0], ps[0], iters);
    std::vector<int> a((size_t)s_eff * nq * 3), b((size_t)s_eff * nq * 3);
    auto grab = [&](int k, std::vector<int>& v) {
        CK(hipMemcpy(v.data(), pi[k], (size_t)s_eff * nq * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v.data() + (size_t)s_eff * nq, pb[k], (size_t)s_eff * nq * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v.data() + 2 * (size_t)s_eff * nq, ps[k], (size_t)s_eff * nq * 4, hipMemcpyDeviceToHost));
    };
    auto diff = [&] {
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
        if (!bad) { printf("   equal to the int8 form: yes\n"); return; }
        { const size_t N = (size_t)s_eff * nq;
          for (int i = 0; i < 6 && i < nq; ++i) printf("     q%d: int8 (idx %d, best %d, second %d)  fp4 (idx %d, best %d, second %d)\n", i, a[i], a[N + i], a[2 * N + i], b[i], b[N + i], b[2 * N + i]); }
        const size_t N = (size_t)s_eff * nq;
        printf("   equal to the int8 form: NO (%zu of %zu words differ; first at array %zu index %zu: %d vs %d)\n", bad, a.size(), first / N, first % N,
               a[first], b[first]);
    };
    grab(0, a);
    CK(hipMemset(pi[1], 0, (size_t)s_eff * nq * 4));
    run(k_fp4<0>, "fp4 form, scale operands 0", s_eff, qb, dq, nq, dr, nr, len, pi[1], pb[1], ps[1], iters);
    grab(1, b); diff();
    CK(hipMemset(pi[1], 0, (size_t)s_eff * nq * 4));
    run(k_fp4<1>, "fp4 form, scale operands 127", s_eff, qb, dq, nq, dr, nr, len, pi[1], pb[1], ps[1], iters);
    grab(1, b); diff();
    CK(hipMemset(pi[1], 0, (size_t)s_eff * nq * 4));
    run(k_fp4<0, true>, "fp4 form, blocks skipped", s_eff, qb, dq, nq, dr, nr, len, pi[1], pb[1], ps[1], iters);
    grab(1, b); diff();
    run(k_pipe<0>, "int8 form (again)", s_eff, qb, dq, nq, dr, nr, len, pi[0], pb[0], ps[0], iters);
    run(k_fp4<0>, "fp4 form (again)", s_eff, qb, dq, nq, dr, nr, len, pi[1], pb[1], ps[1], iters);
    run(k_fp4<0, true>, "fp4 form, blocks skipped (again)", s_eff, qb, dq, nq, dr, nr, len, pi[1], pb[1], ps[1], iters);
    return 0;
}
