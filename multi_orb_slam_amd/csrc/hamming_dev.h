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

