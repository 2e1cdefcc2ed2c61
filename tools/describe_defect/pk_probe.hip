// Stand-alone probe for the open k_describe defect (DESIGN.md "Known defect", profiles/r05/describe_defect.md).
//
// What the in-kernel self-check of round 5 showed (csrc/Makefile VARIANT=slp_check under tools/describe_defect/run_rig.py): the table
// registers hold what memory holds, the LDS bytes are right, and the wrong result is the LOW half of a
//     v_pk_mul_f32 vD, v[x:y], v[a:b] op_sel:[0,1] op_sel_hi:[1,0]          (D.lo = x * b, D.hi = y * a)
// in lanes 48..63 of a wave -- the product comes out as if x * b were 0 -- and only in the lightly loaded phases of the rig (the last
// steps of a run; 65 % of the runs with every stream on ONE hardware queue, 1 % with the default four, 0.1 % with eight).  This program
// issues exactly that instruction sequence (hard-coded registers, the same neighbours: four 16-byte table loads, two v_cvt_f32_f64
// producing (a, b), s_waitcnt vmcnt(3), the sixteen packed multiplies in the failing build's order) and compares every product with
// what the operands give, under launch patterns from "back to back" to "one launch, then the part idles for milliseconds".
//
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=off -o pk_probe pk_probe.hip     run: ./pk_probe [seconds per cell] [1 = next to a matrix-pipe neighbour kernel, 2 = matrix-pipe waves in the same workgroup, 3 = which packed forms are affected]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

__host__ __device__ inline float tab_value(int i, int k) { return (float)(((i * 7 + k * 3) % 27) - 13); }

__global__ __launch_bounds__(256) void k_fill(float4* tab, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tab[i] = make_float4(tab_value(i, 0), tab_value(i, 1), tab_value(i, 2), tab_value(i, 3));
}

// packed multiply D = (S.lo * b, S.hi * a) [O: op_sel form] or (S.lo * a, S.hi * b) [P: plain]; (a, b) = v[116:117]
#define PKO(d0, d1, s0, s1) "v_pk_mul_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define PKP(d0, d1, s0, s1) "v_pk_mul_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117]\n\t"
// the same products from scalar-float multiplies (control)
#define SCO(d0, d1, s0, s1) "v_mul_f32 v" #d0 ", v" #s0 ", v117\n\tv_mul_f32 v" #d1 ", v" #s1 ", v116\n\t"
#define SCP(d0, d1, s0, s1) "v_mul_f32 v" #d0 ", v" #s0 ", v116\n\tv_mul_f32 v" #d1 ", v" #s1 ", v117\n\t"
// further forms for the scope of the fault (k_mixed, FORM 4..6): packed add, packed fma (addend -0.0 in v[154:155]: x * b + -0 == x * b
// bit for bit), and the multiply with the redirect on the FIRST source (D.lo = S.hi * a, D.hi = S.lo * b)
#define AKO(d0, d1, s0, s1) "v_pk_add_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define AKP(d0, d1, s0, s1) "v_pk_add_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117]\n\t"
#define FKO(d0, d1, s0, s1) "v_pk_fma_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117], v[154:155] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
#define FKP(d0, d1, s0, s1) "v_pk_fma_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117], v[154:155]\n\t"
#define MKO(d0, d1, s0, s1) "v_pk_mul_f32 v[" #d0 ":" #d1 "], v[" #s0 ":" #s1 "], v[116:117] op_sel:[1,0] op_sel_hi:[0,1]\n\t"
#define LOADS \
    "global_load_dwordx4 v[100:103], %[voff], %[base] offset:0\n\t"  \
    "global_load_dwordx4 v[104:107], %[voff], %[base] offset:16\n\t" \
    "global_load_dwordx4 v[108:111], %[voff], %[base] offset:32\n\t" \
    "global_load_dwordx4 v[112:115], %[voff], %[base] offset:48\n\t"
// products: j-th load's registers 100 + 4 j .. ; results 118 + 8 j: xy O, xy P, zw O, zw P -- issued in the failing build's order
#define MULS(O, P) \
    "s_waitcnt vmcnt(3)\n\t" O(118, 119, 100, 101) P(120, 121, 100, 101) P(124, 125, 102, 103) \
    "s_waitcnt vmcnt(2)\n\t" O(130, 131, 106, 107) \
    "s_waitcnt vmcnt(0)\n\t" P(144, 145, 112, 113) "v_add_f32 v150, v118, v119\n\t" O(122, 123, 102, 103) O(126, 127, 104, 105) \
    P(136, 137, 108, 109) O(142, 143, 112, 113) "v_sub_f32 v151, v120, v121\n\tv_sub_f32 v152, v124, v125\n\tv_rndne_f32 v150, v150\n\t" \
    P(128, 129, 104, 105) P(132, 133, 106, 107) O(134, 135, 108, 109) O(138, 139, 110, 111) O(146, 147, 114, 115) \
    "v_rndne_f32 v151, v151\n\t" P(140, 141, 110, 111) P(148, 149, 114, 115)
#define STORES \
    "global_store_dwordx4 %[ooff], v[118:121], %[obase] offset:0\n\t"   "global_store_dwordx4 %[ooff], v[122:125], %[obase] offset:16\n\t" \
    "global_store_dwordx4 %[ooff], v[126:129], %[obase] offset:32\n\t"  "global_store_dwordx4 %[ooff], v[130:133], %[obase] offset:48\n\t" \
    "global_store_dwordx4 %[ooff], v[134:137], %[obase] offset:64\n\t"  "global_store_dwordx4 %[ooff], v[138:141], %[obase] offset:80\n\t" \
    "global_store_dwordx4 %[ooff], v[142:145], %[obase] offset:96\n\t"  "global_store_dwordx4 %[ooff], v[146:149], %[obase] offset:112\n\t" \
    "s_waitcnt vmcnt(0)\n\t"
#define CLOBBERS "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
    "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", \
    "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v154", "v155"

// errors[0..3]: wrong products by form -- op_sel low (x * b), op_sel high (y * a), plain low (x * a), plain high (y * b)
// errors[4..7]: by lane / 16;  errors[8..11]: by table load j;  errors[12]: waves checked;  errors[13]: records;  rec[]: the first records
// FORM 0: the failing kernel's order.  1: s_nop 7 twice between the conversions and the first packed multiply.
//      2: (a, b) arrive by v_mov_b32 (no v_cvt_f32_f64 in front).  3: FORM 0 with every packed multiply as two v_mul_f32 (control).
template <int FORM>
__global__ __launch_bounds__(256) void k_probe(const float4* __restrict__ tab, const double* __restrict__ ang, float* __restrict__ out,
                                               unsigned* __restrict__ errors, unsigned* __restrict__ rec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wave;
    // (a, b) the way the kernel makes them: double arithmetic that ends in a conversion
    double da = ang[2 * (w & 1023)], db = ang[2 * (w & 1023) + 1];
    da = da * 0.99999994 + 1.0e-9 * (double)(w & 7); db = db * 1.00000012 - 1.0e-9 * (double)(w & 3);
    const float fa = (float)da, fb = (float)db;
    const unsigned voff = (unsigned)lane * 64u;
    const unsigned ooff = (unsigned)(blockIdx.x * 256 + threadIdx.x) * 128u;
    if (FORM == 0)
        asm volatile(LOADS "v_cvt_f32_f64 v116, %[da]\n\tv_cvt_f32_f64 v117, %[db]\n\t" MULS(PKO, PKP) STORES
                     : : [voff] "v"(voff), [base] "s"(tab), [da] "v"(da), [db] "v"(db), [fa] "v"(fa), [fb] "v"(fb), [ooff] "v"(ooff), [obase] "s"(out) : CLOBBERS);
    else if (FORM == 1)
        asm volatile(LOADS "v_cvt_f32_f64 v116, %[da]\n\tv_cvt_f32_f64 v117, %[db]\n\ts_nop 7\n\ts_nop 7\n\t" MULS(PKO, PKP) STORES
                     : : [voff] "v"(voff), [base] "s"(tab), [da] "v"(da), [db] "v"(db), [fa] "v"(fa), [fb] "v"(fb), [ooff] "v"(ooff), [obase] "s"(out) : CLOBBERS);
    else if (FORM == 2)
        asm volatile(LOADS "v_mov_b32 v116, %[fa]\n\tv_mov_b32 v117, %[fb]\n\t" MULS(PKO, PKP) STORES
                     : : [voff] "v"(voff), [base] "s"(tab), [da] "v"(da), [db] "v"(db), [fa] "v"(fa), [fb] "v"(fb), [ooff] "v"(ooff), [obase] "s"(out) : CLOBBERS);
    else
        asm volatile(LOADS "v_cvt_f32_f64 v116, %[da]\n\tv_cvt_f32_f64 v117, %[db]\n\t" MULS(SCO, SCP) STORES
                     : : [voff] "v"(voff), [base] "s"(tab), [da] "v"(da), [db] "v"(db), [fa] "v"(fa), [fb] "v"(fb), [ooff] "v"(ooff), [obase] "s"(out) : CLOBBERS);
    const float* mine = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * 32;
    if (lane == 0) atomicAdd(&errors[12], 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * lane + j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // xy, zw
            const float x = tab_value(i, 2 * h), y = tab_value(i, 2 * h + 1);
            const float want[4] = {x * fb, y * fa, x * fa, y * fb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float got = __hip_atomic_load(mine + 8 * j + 4 * h + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__float_as_uint(got) != __float_as_uint(want[k])) {
                    atomicAdd(&errors[k], 1u); atomicAdd(&errors[4 + (lane >> 4)], 1u); atomicAdd(&errors[8 + j], 1u);
                    const unsigned slot = atomicAdd(&errors[13], 1u);
                    if (slot < 32) {
                        unsigned* r = rec + slot * 8;
                        r[0] = (unsigned)w; r[1] = (unsigned)(lane << 16 | j << 8 | h << 4 | k); r[2] = __float_as_uint(got); r[3] = __float_as_uint(want[k]);
                        r[4] = __float_as_uint(x); r[5] = __float_as_uint(y); r[6] = __float_as_uint(fa); r[7] = __float_as_uint(fb);
                    }
                }
            }
        }
    }
}

// A neighbour for the probe: waves that keep the matrix pipe of their SIMD busy (what the rig's exchange adds next to k_describe: the
// rig-wide top-2 runs v_mfma_scale_f32_32x32x64_f8f6f4 -- or v_mfma_i32_32x32x32_i8 with MORB_TOP2_FP4=0 -- on the matcher's stream).
// KIND 0: the FP4 form (cbsz = blgp = 4), 1: int8, 2: no matrix instructions (vector ALU only, the control).
typedef int mm_i32x8 __attribute__((ext_vector_type(8)));
typedef int mm_i32x4 __attribute__((ext_vector_type(4)));
typedef int mm_i32x16 __attribute__((ext_vector_type(16)));
typedef float mm_f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k_neighbour(unsigned* sink, int rounds) {
    const int t = threadIdx.x + blockIdx.x * 256;
    if (KIND == 0) {
        mm_i32x8 a, b; mm_f32x16 c;
        for (int i = 0; i < 8; ++i) { a[i] = 0x22222222 ^ (t * (i + 1)); b[i] = 0x2a2a2a2a ^ (t + i); }
        for (int i = 0; i < 16; ++i) c[i] = (float)i;
        for (int r = 0; r < rounds; ++r) { c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0); a[r & 7] ^= r; }
        float acc = 0; for (int i = 0; i < 16; ++i) acc += c[i];
        if (acc == 12345.678f) sink[2] = 1;
    } else if (KIND == 1) {
        mm_i32x4 a, b; mm_i32x16 c;
        for (int i = 0; i < 4; ++i) { a[i] = 0x01ff01ff ^ (t * (i + 1)); b[i] = 0xff01ff01 ^ (t + i); }
        for (int i = 0; i < 16; ++i) c[i] = i;
        for (int r = 0; r < rounds; ++r) { c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); a[r & 3] ^= r; }
        int acc = 0; for (int i = 0; i < 16; ++i) acc += c[i];
        if (acc == 0x12345678) sink[2] = 1;
    } else {
        unsigned v = t * 2654435761u;
        for (int r = 0; r < rounds * 8; ++r) v = v * 1664525u + 1013904223u + (v >> 7);
        if (v == 0xdeadbeefu) sink[2] = v;
    }
}

// Third part (argv[2] = 2): both kinds of wave in ONE workgroup of 512 threads -- waves 0..3 repeat the packed-multiply sequence, waves
// 4..7 (wave w + 4 shares its SIMD with wave w) issue matrix instructions the whole time, in VGPR form and interleaved with vector work as
// the library's top-2 does.  MKIND 0: v_mfma_f32_32x32x64_f8f6f4 cbsz:4 blgp:4 (FP4), 1: v_mfma_i32_32x32x32_i8, 2: vector ALU only.
typedef float pf_f32x16 __attribute__((ext_vector_type(16)));
typedef int pf_i32x16 __attribute__((ext_vector_type(16)));
typedef int pf_i32x4 __attribute__((ext_vector_type(4)));
template <int FORM, int MKIND>
__global__ __launch_bounds__(512) void k_mixed(const float4* __restrict__ tab, const double* __restrict__ ang, float* __restrict__ out,
                                               unsigned* __restrict__ errors, unsigned* __restrict__ rec, int iters, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        const int t = threadIdx.x + blockIdx.x * 512;
        pf_i32x4 a = {0x22222222 ^ t, 0x2a2a2a2a ^ (t * 3), 0x12121212 ^ (t * 5), 0x1a1a1a1a ^ (t * 7)}, b = {t, t * 9, t * 11, t * 13};
        unsigned v = (unsigned)t * 2654435761u;
        if (MKIND == 0) {
            pf_f32x16 c0, c1;
            for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 1.f; }
            for (int r = 0; r < iters * 40; ++r) {
                asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %2, %3, %0 cbsz:4 blgp:4\n\tv_perm_b32 %4, 0, %4, %5\n\t"
                             "v_mfma_f32_32x32x64_f8f6f4 %1, %3, %2, %1 cbsz:4 blgp:4\n\tv_and_b32 %5, 0x3030303, %4"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(v), "v"(t));
            }
            float acc = 0; for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
            if (acc == 12345.678f) sink[2] = v;
        } else if (MKIND == 1) {
            pf_i32x16 c0, c1;
            for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 1; }
            for (int r = 0; r < iters * 40; ++r) {
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %2, %3, %0\n\tv_perm_b32 %4, 0, %4, %5\n\t"
                             "v_mfma_i32_32x32x32_i8 %1, %3, %2, %1\n\tv_and_b32 %5, 0x3030303, %4"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(v), "v"(t));
            }
            int acc = 0; for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
            if (acc == 0x12345678) sink[2] = v;
        } else {
            for (int r = 0; r < iters * 600; ++r) v = v * 1664525u + 1013904223u + (v >> 7);
            if (v == 0xdeadbeefu) sink[2] = v;
        }
        return;
    }
    const int w = blockIdx.x * 4 + wave;
    double da = ang[2 * (w & 1023)], db = ang[2 * (w & 1023) + 1];
    da = da * 0.99999994 + 1.0e-9 * (double)(w & 7); db = db * 1.00000012 - 1.0e-9 * (double)(w & 3);
    const float fa = (float)da, fb = (float)db;
    const unsigned voff = (unsigned)lane * 64u;
    const unsigned ooff = (unsigned)(blockIdx.x * 256 + (threadIdx.x & 255)) * 128u;
    const float* mine = out + (size_t)(blockIdx.x * 256 + (threadIdx.x & 255)) * 32;
    for (int it = 0; it < iters; ++it) {
#define SEQ(O, P) asm volatile(LOADS "v_cvt_f32_f64 v116, %[da]\n\tv_cvt_f32_f64 v117, %[db]\n\tv_mov_b32 v154, 0x80000000\n\tv_mov_b32 v155, 0x80000000\n\t" MULS(O, P) STORES \
                         : : [voff] "v"(voff), [base] "s"(tab), [da] "v"(da), [db] "v"(db), [fa] "v"(fa), [fb] "v"(fb), [ooff] "v"(ooff), [obase] "s"(out) : CLOBBERS)
        if (FORM == 0) SEQ(PKO, PKP);
        else if (FORM == 4) SEQ(AKO, AKP);
        else if (FORM == 5) SEQ(FKO, FKP);
        else if (FORM == 6) SEQ(MKO, PKP);
        else SEQ(SCO, SCP);
#undef SEQ
        if (lane == 0) atomicAdd(&errors[12], 1u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * lane + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float x = tab_value(i, 2 * h), y = tab_value(i, 2 * h + 1);
                const float want[4] = {FORM == 4 ? x + fb : FORM == 6 ? y * fa : x * fb, FORM == 4 ? y + fa : FORM == 6 ? x * fb : y * fa,
                                       FORM == 4 ? x + fa : x * fa, FORM == 4 ? y + fb : y * fb};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float got = __hip_atomic_load(mine + 8 * j + 4 * h + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__float_as_uint(got) != __float_as_uint(want[k])) {
                        atomicAdd(&errors[k], 1u); atomicAdd(&errors[4 + (lane >> 4)], 1u); atomicAdd(&errors[8 + j], 1u);
                        const unsigned slot = atomicAdd(&errors[13], 1u);
                        if (slot < 32) {
                            unsigned* r = rec + slot * 8;
                            r[0] = (unsigned)w; r[1] = (unsigned)(lane << 16 | j << 8 | h << 4 | k); r[2] = __float_as_uint(got); r[3] = __float_as_uint(want[k]);
                            r[4] = __float_as_uint(x); r[5] = __float_as_uint(y); r[6] = __float_as_uint(fa); r[7] = __float_as_uint(fb);
                        }
                    }
                }
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int FORM>
static void launch(int grid, hipStream_t st, const float4* tab, const double* ang, float* out, unsigned* errors, unsigned* rec) {
    hipLaunchKernelGGL(k_probe<FORM>, dim3(grid), dim3(256), 0, st, tab, ang, out, errors, rec);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 4.0;
    const int GRID = 270;   // k_describe's size at 1000 keypoints: about one workgroup (4 waves, one per SIMD) per CU
    float4* tab; double* ang; float* out; unsigned* errors; unsigned* rec;
    CK(hipMalloc(&tab, 256 * sizeof(float4))); CK(hipMalloc(&ang, 2048 * sizeof(double))); CK(hipMalloc(&errors, 64)); CK(hipMalloc(&rec, 32 * 8 * 4));
    CK(hipMalloc(&out, (size_t)4 * GRID * 256 * 32 * sizeof(float)));
    std::vector<double> hang(2048);
    for (int i = 0; i < 2048; ++i) hang[i] = std::sin(0.37 * i + 0.1);
    CK(hipMemcpy(ang, hang.data(), 2048 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, 0, tab, 256);
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const char* form_name[4] = {"as in the failing build", "s_nop 7 x2 behind the conversions", "(a, b) by v_mov_b32, no conversion", "control: v_mul_f32 pairs"};
    // pattern: idle microseconds between launches (0 = back to back, 64 launches per synchronisation); grid multiple
    const struct { int idle_us; int grid_mul; const char* name; } pat[] = {
        {0, 1, "back to back"}, {0, 4, "back to back, 4 x the workgroups"}, {100, 1, "launch, sync, idle 100 us"}, {1000, 1, "launch, sync, idle 1 ms"},
        {5000, 1, "launch, sync, idle 5 ms"}, {20000, 1, "launch, sync, idle 20 ms"}};
    // Second part (argv[2] = 1): the probe next to a neighbour kernel on a second stream -- launches of both kept in flight together
    // (the neighbour's workgroups share compute units with the probe's), for each kind of neighbour.
    if (argc > 2 && std::atoi(argv[2]) == 1) {
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        const char* nb_name[3] = {"FP4 MFMA neighbour (v_mfma_scale_f32_32x32x64_f8f6f4)", "int8 MFMA neighbour (v_mfma_i32_32x32x32_i8)", "vector-ALU neighbour (control)"};
        for (int form : {0, 3}) {
            for (int nb = 0; nb < 3; ++nb) {
                CK(hipMemset(errors, 0, 64)); CK(hipMemset(rec, 0, 32 * 8 * 4));
                const auto t0 = std::chrono::steady_clock::now();
                long n = 0;
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                    for (int k = 0; k < 8; ++k) {
                        if (nb == 0) hipLaunchKernelGGL(k_neighbour<0>, dim3(512), dim3(256), 0, st2, errors, 400);
                        else if (nb == 1) hipLaunchKernelGGL(k_neighbour<1>, dim3(512), dim3(256), 0, st2, errors, 400);
                        else hipLaunchKernelGGL(k_neighbour<2>, dim3(512), dim3(256), 0, st2, errors, 400);
                        for (int q = 0; q < 4; ++q) { if (form == 0) launch<0>(GRID, st, tab, ang, out, errors, rec); else launch<3>(GRID, st, tab, ang, out, errors, rec); ++n; }
                    }
                    CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
                }
                unsigned h[16], r[32 * 8];
                CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(r, rec, sizeof(r), hipMemcpyDeviceToHost));
                std::printf("form %d (%s) | %s | launches %ld waves %u | wrong products: op_sel lo %u hi %u, plain lo %u hi %u | by lane/16: %u %u %u %u | by load: %u %u %u %u\n",
                            form, form_name[form], nb_name[nb], n, h[12], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
                for (unsigned s2 = 0; s2 < h[13] && s2 < 6; ++s2) {
                    const unsigned* q = r + s2 * 8;
                    float got, want, x, y, a, b;
                    std::memcpy(&got, q + 2, 4); std::memcpy(&want, q + 3, 4); std::memcpy(&x, q + 4, 4); std::memcpy(&y, q + 5, 4); std::memcpy(&a, q + 6, 4); std::memcpy(&b, q + 7, 4);
                    std::printf("    wave %u lane %u load %u half %u product %u: got %.9g want %.9g (x %.0f y %.0f a %.9g b %.9g)\n", q[0], q[1] >> 16, (q[1] >> 8) & 0xff,
                                (q[1] >> 4) & 0xf, q[1] & 0xf, got, want, x, y, a, b);
                }
                std::fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 2 && std::atoi(argv[2]) == 3) {
        // the scope of the fault: which packed-f32 forms go wrong next to int8 MFMA waves on the same SIMDs
        const struct { int form; const char* name; } forms[] = {{0, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (redirect on the second source)"},
            {6, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (redirect on the first source)"}, {4, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]"},
            {5, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]"}, {3, "v_mul_f32 pairs (control)"}};
        for (const auto& F : forms) {
            CK(hipMemset(errors, 0, 64)); CK(hipMemset(rec, 0, 32 * 8 * 4));
            const auto t0 = std::chrono::steady_clock::now();
            long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                for (int k = 0; k < 16; ++k) {
#define LS(Fm) hipLaunchKernelGGL((k_mixed<Fm, 1>), dim3(GRID), dim3(512), 0, st, tab, ang, out, errors, rec, 24, errors)
                    switch (F.form) { case 0: LS(0); break; case 6: LS(6); break; case 4: LS(4); break; case 5: LS(5); break; default: LS(3); break; }
                    ++n;
                }
                CK(hipStreamSynchronize(st));
            }
            unsigned h[16];
            CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost));
            std::printf("%s | int8 MFMA waves on the same SIMDs | sequences %u | wrong results: redirected-form low %u high %u, plain-form low %u high %u | by lane/16: %u %u %u %u\n",
                        F.name, h[12], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            std::fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && std::atoi(argv[2]) == 2) {
        const char* mk_name[3] = {"FP4 MFMA waves on the same SIMDs (v_mfma_f32_32x32x64_f8f6f4 cbsz:4 blgp:4, VGPR form)", "int8 MFMA waves on the same SIMDs (v_mfma_i32_32x32x32_i8)",
                                  "vector-ALU waves on the same SIMDs (control)"};
        for (int form : {0, 3}) {
            for (int mk = 0; mk < 3; ++mk) {
                CK(hipMemset(errors, 0, 64)); CK(hipMemset(rec, 0, 32 * 8 * 4));
                const auto t0 = std::chrono::steady_clock::now();
                long n = 0;
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                    for (int k = 0; k < 16; ++k) {
#define LM(F, M) hipLaunchKernelGGL((k_mixed<F, M>), dim3(GRID), dim3(512), 0, st, tab, ang, out, errors, rec, 24, errors)
                        if (form == 0) { if (mk == 0) LM(0, 0); else if (mk == 1) LM(0, 1); else LM(0, 2); }
                        else { if (mk == 0) LM(3, 0); else if (mk == 1) LM(3, 1); else LM(3, 2); }
                        ++n;
                    }
                    CK(hipStreamSynchronize(st));
                }
                unsigned h[16], r[32 * 8];
                CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(r, rec, sizeof(r), hipMemcpyDeviceToHost));
                std::printf("form %d (%s) | %s | launches %ld sequences %u | wrong products: op_sel lo %u hi %u, plain lo %u hi %u | by lane/16: %u %u %u %u | by load: %u %u %u %u\n",
                            form, form_name[form], mk_name[mk], n, h[12], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
                for (unsigned s2 = 0; s2 < h[13] && s2 < 6; ++s2) {
                    const unsigned* q = r + s2 * 8;
                    float got, want, x, y, a, b;
                    std::memcpy(&got, q + 2, 4); std::memcpy(&want, q + 3, 4); std::memcpy(&x, q + 4, 4); std::memcpy(&y, q + 5, 4); std::memcpy(&a, q + 6, 4); std::memcpy(&b, q + 7, 4);
                    std::printf("    wave %u lane %u load %u half %u product %u: got %.9g want %.9g (x %.0f y %.0f a %.9g b %.9g)\n", q[0], q[1] >> 16, (q[1] >> 8) & 0xff,
                                (q[1] >> 4) & 0xf, q[1] & 0xf, got, want, x, y, a, b);
                }
                std::fflush(stdout);
            }
        }
        return 0;
    }
    for (int form = 0; form < 4; ++form) {
        for (const auto& P : pat) {
            CK(hipMemset(errors, 0, 64)); CK(hipMemset(rec, 0, 32 * 8 * 4));
            const auto t0 = std::chrono::steady_clock::now();
            long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                const int g = GRID * P.grid_mul;
                switch (form) {
                    case 0: launch<0>(g, st, tab, ang, out, errors, rec); break;
                    case 1: launch<1>(g, st, tab, ang, out, errors, rec); break;
                    case 2: launch<2>(g, st, tab, ang, out, errors, rec); break;
                    default: launch<3>(g, st, tab, ang, out, errors, rec); break;
                }
                ++n;
                if (P.idle_us > 0) { CK(hipStreamSynchronize(st)); std::this_thread::sleep_for(std::chrono::microseconds(P.idle_us)); }
                else if ((n & 63) == 0) CK(hipStreamSynchronize(st));
            }
            CK(hipStreamSynchronize(st));
            unsigned h[16], r[32 * 8];
            CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(r, rec, sizeof(r), hipMemcpyDeviceToHost));
            std::printf("form %d (%s) | %s | launches %ld waves %u | wrong products: op_sel lo %u hi %u, plain lo %u hi %u | by lane/16: %u %u %u %u | by load: %u %u %u %u\n",
                        form, form_name[form], P.name, n, h[12], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
            for (unsigned s = 0; s < h[13] && s < 6; ++s) {
                const unsigned* q = r + s * 8;
                float got, want, x, y, a, b;
                std::memcpy(&got, q + 2, 4); std::memcpy(&want, q + 3, 4); std::memcpy(&x, q + 4, 4); std::memcpy(&y, q + 5, 4); std::memcpy(&a, q + 6, 4); std::memcpy(&b, q + 7, 4);
                std::printf("    wave %u lane %u load %u half %u product %u: got %.9g want %.9g (x %.0f y %.0f a %.9g b %.9g)\n", q[0], q[1] >> 16, (q[1] >> 8) & 0xff,
                            (q[1] >> 4) & 0xf, q[1] & 0xf, got, want, x, y, a, b);
            }
            std::fflush(stdout);
        }
    }
    return 0;
}
