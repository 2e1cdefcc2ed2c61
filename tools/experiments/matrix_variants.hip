// Experiment harness (not product code): variants of the Hamming distance-matrix kernel, timed with hipEvents.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mv tools/experiments/matrix_variants.hip && /tmp/mv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#ifdef WITH_PROD
#include "../../multi_orb_slam_amd/csrc/matcher.hip"
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xbcnt_acc(unsigned x, unsigned acc) { unsigned r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }
__device__ __forceinline__ unsigned xham(const uint4& q0, const uint4& q1, const uint4& a, const uint4& b) {
    unsigned d = __popc(q0.x ^ a.x);
    d = xbcnt_acc(q0.y ^ a.y, d); d = xbcnt_acc(q0.z ^ a.z, d); d = xbcnt_acc(q0.w ^ a.w, d);
    d = xbcnt_acc(q1.x ^ b.x, d); d = xbcnt_acc(q1.y ^ b.y, d); d = xbcnt_acc(q1.z ^ b.z, d); d = xbcnt_acc(q1.w ^ b.w, d);
    return d;
}
__device__ __forceinline__ void xtouch(const uint4& a, const uint4& b) { asm volatile("" ::"s"(a.x), "s"(a.y), "s"(a.z), "s"(a.w), "s"(b.x), "s"(b.y), "s"(b.z), "s"(b.w)); }

// MODE 0: full; 1: store-only (no compute); 2: compute-only (one store at the end); NT: nontemporal store
template <int MODE, bool NT, int RPL>
__global__ __launch_bounds__(256) void k_mat(const uint4* __restrict__ q, int nq, const uint4* __restrict__ r, int nr, uint16_t* __restrict__ out, int qpb) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if ((tile + 1) * 64 * RPL > nr) return;
    const int r0 = tile * 64 * RPL + lane * RPL;
    uint4 ra[RPL], rb[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) { ra[k] = r[2 * (r0 + k)]; rb[k] = r[2 * (r0 + k) + 1]; }
    const int qa = blockIdx.y * qpb, qb = min(nq, qa + qpb);
    uint4 x0 = q[2 * qa], x1 = q[2 * qa + 1], y0, y1;
    unsigned d[RPL];
    unsigned acc = 0;
    auto row = [&](int qi, const uint4& a0, const uint4& a1) {
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < RPL; ++k) d[k] = a0.x + k;
        } else {
#pragma unroll
            for (int k = 0; k < RPL; ++k) d[k] = xham(a0, a1, ra[k], rb[k]);
        }
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < RPL; ++k) acc += d[k];
            return;
        }
        uint16_t* p = out + (size_t)qi * nr + r0;
        if (RPL == 8) {
            v4u o; o.x = d[0] | (d[1] << 16); o.y = d[2] | (d[3] << 16); o.z = d[4] | (d[5] << 16); o.w = d[6] | (d[7] << 16);
            if (NT) __builtin_nontemporal_store(o, (v4u*)p); else *(v4u*)p = o;
        } else {
            v2u o; o.x = d[0] | (d[1] << 16); o.y = d[2] | (d[3] << 16);
            if (NT) __builtin_nontemporal_store(o, (v2u*)p); else *(v2u*)p = o;
        }
    };
    const int npairs = (qb - qa) >> 1;
    int qi = qa;
    for (int p = 0; p < npairs; ++p, qi += 2) {
        xtouch(x0, x1); __builtin_amdgcn_sched_barrier(0);
        y0 = q[2 * (qi + 1)]; y1 = q[2 * (qi + 1) + 1];
        __builtin_amdgcn_sched_barrier(0);
        row(qi, x0, x1);
        xtouch(y0, y1); __builtin_amdgcn_sched_barrier(0);
        { const int qn = min(qi + 2, qb - 1); x0 = q[2 * qn]; x1 = q[2 * qn + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        row(qi + 1, y0, y1);
    }
    if (MODE == 2) out[(size_t)qa * nr + r0] = (uint16_t)acc;
}

// VALU rate probes: 8 independent chains, OP 0 = v_xor_b32 (SGPR operand), 1 = v_bcnt_u32_b32 accumulate, 2 = v_add_u32
template <int OP>
__global__ __launch_bounds__(256) void k_rate(unsigned* out, unsigned seed, int iters) {
    unsigned a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 8 + k + seed;
    unsigned s = __builtin_amdgcn_readfirstlane(seed * 77);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[k]) : "s"(s));
            if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(a[(k + 1) & 7]));
            if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "s"(s));
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// 16-wave blocks (one per CU): the block covers WAVES*512 consecutive references, i.e. WAVES KB contiguous per row
template <int WAVES, bool NT>
__global__ __launch_bounds__(64 * WAVES) void k_mat_wide(const uint4* __restrict__ q, int nq, const uint4* __restrict__ r, int nr, uint16_t* __restrict__ out, int qpb) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * WAVES + wave;
    if ((tile + 1) * 512 > nr) return;
    const int r0 = tile * 512 + lane * 8;
    uint4 ra[8], rb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ra[k] = r[2 * (r0 + k)]; rb[k] = r[2 * (r0 + k) + 1]; }
    const int qa = blockIdx.y * qpb, qb = min(nq, qa + qpb);
    uint4 x0 = q[2 * qa], x1 = q[2 * qa + 1], y0, y1;
    unsigned d[8];
    auto row = [&](int qi, const uint4& a0, const uint4& a1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = xham(a0, a1, ra[k], rb[k]);
        uint16_t* p = out + (size_t)qi * nr + r0;
        v4u o; o.x = d[0] | (d[1] << 16); o.y = d[2] | (d[3] << 16); o.z = d[4] | (d[5] << 16); o.w = d[6] | (d[7] << 16);
        if (NT) __builtin_nontemporal_store(o, (v4u*)p); else *(v4u*)p = o;
    };
    const int npairs = (qb - qa) >> 1;
    int qi = qa;
    for (int p = 0; p < npairs; ++p, qi += 2) {
        xtouch(x0, x1); __builtin_amdgcn_sched_barrier(0);
        y0 = q[2 * (qi + 1)]; y1 = q[2 * (qi + 1) + 1];
        __builtin_amdgcn_sched_barrier(0);
        row(qi, x0, x1);
        xtouch(y0, y1); __builtin_amdgcn_sched_barrier(0);
        { const int qn = min(qi + 2, qb - 1); x0 = q[2 * qn]; x1 = q[2 * qn + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        row(qi + 1, y0, y1);
    }
}

// pure streaming fill with the same total bytes (ceiling for a store stream from a compute-style grid)
__global__ __launch_bounds__(256) void k_fill(v4u* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    v4u v = {1, 2, 3, 4};
    for (; i < n16; i += stride) __builtin_nontemporal_store(v, out + i);
}

template <typename F> float timeit(F f, int iters = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32768;
    std::vector<uint32_t> h((size_t)n * 8);
    uint32_t s = 12345; for (auto& v : h) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; v = s; }
    uint4 *dq, *dr; uint16_t* dout;
    CK(hipMalloc(&dq, (size_t)n * 32)); CK(hipMalloc(&dr, (size_t)n * 32)); CK(hipMalloc(&dout, (size_t)n * n * 2));
    CK(hipMemcpy(dq, h.data(), (size_t)n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, h.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    const double bytes = 2.0 * n * n;
    auto rep = [&](const char* name, float us) { printf("%-28s %8.1f us  %7.1f GB/s\n", name, us, bytes / us / 1e3); };
    for (int qpb : {64, 128}) {
        printf("-- qpb %d\n", qpb);
        dim3 g8((n / 512 + 3) / 4, (n + qpb - 1) / qpb), g4((n / 256 + 3) / 4, (n + qpb - 1) / qpb);
        rep("full nt rpl8", timeit([&] { hipLaunchKernelGGL((k_mat<0, true, 8>), g8, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("full plain rpl8", timeit([&] { hipLaunchKernelGGL((k_mat<0, false, 8>), g8, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("store-only nt rpl8", timeit([&] { hipLaunchKernelGGL((k_mat<1, true, 8>), g8, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("store-only plain rpl8", timeit([&] { hipLaunchKernelGGL((k_mat<1, false, 8>), g8, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("compute-only rpl8", timeit([&] { hipLaunchKernelGGL((k_mat<2, true, 8>), g8, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("full nt rpl4", timeit([&] { hipLaunchKernelGGL((k_mat<0, true, 4>), g4, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("full plain rpl4", timeit([&] { hipLaunchKernelGGL((k_mat<0, false, 4>), g4, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
        rep("compute-only rpl4", timeit([&] { hipLaunchKernelGGL((k_mat<2, true, 4>), g4, dim3(256), 0, 0, dq, n, dr, n, dout, qpb); }));
    }
    {
        unsigned* dtmp; CK(hipMalloc(&dtmp, 4096 * 256 * 4));
        const int iters = 4096; const double ops = 4096.0 * 256 * 8 * iters;
        float u0 = timeit([&] { hipLaunchKernelGGL(k_rate<0>, dim3(4096), dim3(256), 0, 0, dtmp, 3u, iters); });
        float u1 = timeit([&] { hipLaunchKernelGGL(k_rate<1>, dim3(4096), dim3(256), 0, 0, dtmp, 3u, iters); });
        float u2 = timeit([&] { hipLaunchKernelGGL(k_rate<2>, dim3(4096), dim3(256), 0, 0, dtmp, 3u, iters); });
        printf("rate: v_xor %.1f T lane-ops/s, v_bcnt %.1f T, v_add %.1f T\n", ops / u0 / 1e6, ops / u1 / 1e6, ops / u2 / 1e6);
    }
    for (int qpb : {128, 256, 512, 1024}) {
        dim3 g16((n / 512 + 15) / 16, (n + qpb - 1) / qpb), g8w((n / 512 + 7) / 8, (n + qpb - 1) / qpb);
        char nm[64];
        snprintf(nm, 64, "wide16 plain qpb%d", qpb); rep(nm, timeit([&] { hipLaunchKernelGGL((k_mat_wide<16, false>), g16, dim3(1024), 0, 0, dq, n, dr, n, dout, qpb); }));
        snprintf(nm, 64, "wide16 nt qpb%d", qpb); rep(nm, timeit([&] { hipLaunchKernelGGL((k_mat_wide<16, true>), g16, dim3(1024), 0, 0, dq, n, dr, n, dout, qpb); }));
        snprintf(nm, 64, "wide8 plain qpb%d", qpb); rep(nm, timeit([&] { hipLaunchKernelGGL((k_mat_wide<8, false>), g8w, dim3(512), 0, 0, dq, n, dr, n, dout, qpb); }));
    }
#ifdef WITH_PROD
    rep("PROD lib kernel, null stream", timeit([&] { orbm_hamming_matrix_device((const uint8_t*)dq, n, (const uint8_t*)dr, n, dout, nullptr); }));
    { hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 3; i++) orbm_hamming_matrix_device((const uint8_t*)dq, n, (const uint8_t*)dr, n, dout, st);
      CK(hipStreamSynchronize(st)); CK(hipEventRecord(a, st));
      for (int i = 0; i < 10; i++) orbm_hamming_matrix_device((const uint8_t*)dq, n, (const uint8_t*)dr, n, dout, st);
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
      rep("PROD lib kernel, own stream", ms * 100.f); }
#endif
    {   // isolate stream / overlap effects
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipStream_t st2; CK(hipStreamCreate(&st2));
        dim3 g8w((n / 512 + 7) / 8, (n + 255) / 256);
        auto t_stream = [&](hipStream_t s_, bool sync_each) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_mat_wide<8, false>), g8w, dim3(512), 0, s_, dq, n, dr, n, dout, 256);
            CK(hipStreamSynchronize(s_));
            float tot = 0;
            if (!sync_each) {
                CK(hipEventRecord(a, s_));
                for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_mat_wide<8, false>), g8w, dim3(512), 0, s_, dq, n, dr, n, dout, 256);
                CK(hipEventRecord(b, s_)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&tot, a, b));
            } else {
                for (int i = 0; i < 10; i++) {
                    CK(hipEventRecord(a, s_));
                    hipLaunchKernelGGL((k_mat_wide<8, false>), g8w, dim3(512), 0, s_, dq, n, dr, n, dout, 256);
                    CK(hipEventRecord(b, s_)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms;
                }
            }
            return tot * 100.f;
        };
        rep("wide8 null stream b2b", t_stream(0, false));
        rep("wide8 null stream sync-each", t_stream(0, true));
        rep("wide8 nonblocking b2b", t_stream(st, false));
        rep("wide8 nonblocking sync-each", t_stream(st, true));
        rep("wide8 blocking-stream b2b", t_stream(st2, false));
    }
    for (int blocks : {256, 1024, 2048, 8192})
    { char nm[64]; snprintf(nm, 64, "fill nt %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, (v4u*)dout, (size_t)n * n / 8); })); }
    return 0;
}
