// Probe for the fault of round 4 (profiles/r04/notes_experiments.md): does a packed f32 multiply on a register pair that a vector-memory
// load has just written ever see wrong data when many queues share the part?  Every lane loads four float4 of a table whose contents
// are a function of the index, then (a) checks the loaded values themselves, (b) multiplies them by a wave-uniform (a, b) once with
// v_pk_mul_f32 ... op_sel (what the SLP vectorizer made of k_describe's rotation) and once with v_mul_f32, and compares.
// Four host threads x four streams launch the probe next to an LDS/ALU noise kernel for a few seconds.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_probe pk_f32_probe.hip ; run: ./pk_f32_probe [seconds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float tab_value(int i, int k) { return (float)(((i * 7 + k * 3) % 27) - 13); }

__global__ __launch_bounds__(256) void k_fill(float4* tab, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tab[i] = make_float4(tab_value(i, 0), tab_value(i, 1), tab_value(i, 2), tab_value(i, 3));
}

// errors[0]: loaded value != expected; errors[1]: packed product != scalar product; errors[2..5]: by j; errors[6..9]: by lane / 16
__global__ __launch_bounds__(256) void k_probe(const float4* __restrict__ tab, const float* __restrict__ ab, unsigned* __restrict__ errors,
                                               const unsigned char* __restrict__ img, unsigned* __restrict__ sink) {
    __shared__ unsigned char s_patch[4][2304];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wave;
    // something like the kernel's earlier phases: byte traffic through LDS
    unsigned acc = 0;
    for (int i = lane; i < 2304; i += 64) s_patch[wave][i] = img[(w * 131 + i) & 0xfffff];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 2304; i += 64) acc += s_patch[wave][(i * 37) % 2304];
    const float a = ab[2 * (w & 1023)], b = ab[2 * (w & 1023) + 1];
    const f32x2 abv = {a, b};
    float4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = tab[4 * lane + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * lane + j;
        const f32x2 xy = {q[j].x, q[j].y}, zw = {q[j].z, q[j].w};
        f32x2 p0, p1;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p0) : "v"(xy), "v"(abv));   // (x * b, y * a)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p1) : "v"(zw), "v"(abv));
        float s0, s1, s2, s3;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(q[j].x), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(q[j].y), "v"(a));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s2) : "v"(q[j].z), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s3) : "v"(q[j].w), "v"(a));
        const bool bad_load = q[j].x != tab_value(i, 0) || q[j].y != tab_value(i, 1) || q[j].z != tab_value(i, 2) || q[j].w != tab_value(i, 3);
        const bool bad_mul = __float_as_uint(p0.x) != __float_as_uint(s0) || __float_as_uint(p0.y) != __float_as_uint(s1) ||
                             __float_as_uint(p1.x) != __float_as_uint(s2) || __float_as_uint(p1.y) != __float_as_uint(s3);
        if (bad_load) { atomicAdd(&errors[0], 1u); atomicAdd(&errors[2 + j], 1u); atomicAdd(&errors[6 + (lane >> 4)], 1u); }
        if (bad_mul) atomicAdd(&errors[1], 1u);
        acc += __float_as_uint(p0.x + p1.y);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_noise(unsigned* sink, int rounds) {
    __shared__ unsigned s[4096];
    unsigned v = threadIdx.x * 2654435761u + blockIdx.x;
    for (int r = 0; r < rounds; ++r) {
        s[(threadIdx.x * 17 + r * 31) & 4095] = v;
        __syncthreads();
        v = v * 1664525u + s[(threadIdx.x * 29 + r) & 4095];
        __syncthreads();
    }
    if (v == 0xdeadbeefu) sink[1] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 10.0;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 4, streams_per = argc > 3 ? std::atoi(argv[3]) : 4;
    float4* tab; float* ab; unsigned* errors; unsigned char* img; unsigned* sink;
    CK(hipMalloc(&tab, 256 * sizeof(float4))); CK(hipMalloc(&ab, 2048 * sizeof(float))); CK(hipMalloc(&errors, 64)); CK(hipMalloc(&img, 1 << 20));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(errors, 0, 64)); CK(hipMemset(img, 7, 1 << 20));
    std::vector<float> hab(2048);
    for (int i = 0; i < 2048; ++i) hab[i] = (float)std::sin(0.37 * i + 0.1);
    CK(hipMemcpy(ab, hab.data(), 2048 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, 0, tab, 256);
    CK(hipDeviceSynchronize());
    std::atomic<long> launches{0};
    auto worker = [&](int t) {
        std::vector<hipStream_t> st(streams_per);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const auto t0 = std::chrono::steady_clock::now();
        long n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int k = 0; k < streams_per; ++k) {
                hipLaunchKernelGGL(k_probe, dim3(500), dim3(256), 0, st[k], (const float4*)tab, (const float*)ab, errors, (const unsigned char*)img, sink);
                hipLaunchKernelGGL(k_noise, dim3(200 + 50 * k), dim3(256), 0, st[k], sink, 40 + 10 * t);
                hipLaunchKernelGGL(k_probe, dim3(250), dim3(256), 0, st[k], (const float4*)tab, (const float*)ab, errors, (const unsigned char*)img, sink);
                ++n;
            }
            if ((n & 63) == 0) for (auto& s : st) CK(hipStreamSynchronize(s));
        }
        for (auto& s : st) { CK(hipStreamSynchronize(s)); CK(hipStreamDestroy(s)); }
        launches += 2 * n;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    unsigned h[16];
    CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost));
    std::printf("%d threads x %d streams, %.0f s: %ld probe launches (%.1f M waves x 4 loads): wrong loaded values %u (by j: %u %u %u %u; by lane/16: %u %u %u %u), packed != scalar products %u\n",
                threads, streams_per, seconds, launches.load(), launches.load() * 0.0015, h[0], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[1]);
    return 0;
}
