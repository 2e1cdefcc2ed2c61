// hamming_dev.h -- 256-bit Hamming distance device helpers shared by the matcher's translation units.
#pragma once
#include <hip/hip_runtime.h>

// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32's accumulate operand; hipcc otherwise emits bcnt + add3 trees)
static __device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

static __device__ __forceinline__ unsigned ham256_chain(const uint4& q0, const uint4& q1, const uint4& a, const uint4& b) {
    unsigned d = __popc(q0.x ^ a.x);
    d = bcnt_acc(q0.y ^ a.y, d); d = bcnt_acc(q0.z ^ a.z, d); d = bcnt_acc(q0.w ^ a.w, d);
    d = bcnt_acc(q1.x ^ b.x, d); d = bcnt_acc(q1.y ^ b.y, d); d = bcnt_acc(q1.z ^ b.z, d); d = bcnt_acc(q1.w ^ b.w, d);
    return d;
}

static __device__ __forceinline__ int ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}


// Merge of the S slice partials of a top-2 (second = 2nd smallest WITH multiplicity: min(s1, s2, max(b1, b2)); the lower slice wins a tie
// of the best): k_top2_merge's body, also run by extra workgroups of the resolve's launch (MergeJob: search.hip, isolated steps).
struct MergeJob {
    const int* p_idx; const int* p_best; const int* p_second; int S, nq;
    int* o_idx; int* o_best; int* o_second; const int* d_range;
    unsigned* done; unsigned target;   // (carried by the resolve's launch: every merging workgroup adds one to *done behind a system-scope
};                                     //  release; the resolve waits for `target` before its result words say "finished")
static __device__ __forceinline__ void top2_merge_query(const int* __restrict__ p_idx, const int* __restrict__ p_best, const int* __restrict__ p_second,
                                                        int S, int nq, int qi, int* __restrict__ best_idx, int* __restrict__ best_dist,
                                                        int* __restrict__ second_dist) {
    int B = 256, Sd = 256, I = -1;
    for (int k = 0; k < S; ++k) {
        const int b2 = p_best[(size_t)k * nq + qi], s2 = p_second[(size_t)k * nq + qi], i2 = p_idx[(size_t)k * nq + qi];
        Sd = min(min(Sd, s2), max(B, b2));
        I = b2 < B ? i2 : I;
        B = min(B, b2);
    }
    best_idx[qi] = I; best_dist[qi] = B; second_dist[qi] = Sd;
}
