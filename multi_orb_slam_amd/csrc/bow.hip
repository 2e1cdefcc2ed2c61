// bow.hip -- vocabulary-tree transform and BoW-gated searches on MI355X (gfx950).  C ABI: include/orbv.h.
//
//   k_bow_transform   one descent per feature, 16 lanes per feature (one child per lane, DBoW2's k <= 20 takes two
//                     passes at most); the tree is renumbered breadth-first at load so that the children of a node are one
//                     contiguous run of 32-byte descriptors (TemplatedVocabulary.h:1219-1260).
//   k_bow_join<MODE, NW>  one (NW = 1) or four (NW = 4, nodes of >= 128 candidates) wavefronts per vocabulary node common to both sides: queries strictly in the reference's order
//                     (the "already matched" state makes them order-dependent inside a node, never across nodes), the
//                     candidates of the node spread over the 64 lanes, top-2 by wave reductions
//                     (src/ORBmatcher.cc:206-388, :996-1165, :1364-1786).
//   k_bow_finish      rotation histogram -> three maxima -> removal -> count, results written to pinned host memory.
//
// Latency-bound integer work (a few 10^4 Hamming distances per call): no LDS tiling to speak of, no matrix cores.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/orbv.h"
#include "orb_common.h"

using morb::DevBuf;
using morb::PinnedBuf;

namespace {

constexpr int HISTO = 30;        // ORBmatcher::HISTO_LENGTH, src/ORBmatcher.cc:39
constexpr int MAX_LEVELS = 32;   // pyramid levels a triangulation search may name
constexpr int JOIN_MAX_NODE = 32768;  // candidates of one node (one LDS byte each)
constexpr int JOIN_LDS_BYTES = 65536; // dynamic LDS of one join workgroup: claimed bytes + staged descriptors

__device__ __forceinline__ int ham256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(256) void k_bow_transform(const uint4* __restrict__ vdesc, const int* __restrict__ first_child,
                                                       const uint32_t* __restrict__ orig, const uint32_t* __restrict__ word,
                                                       const uint4* __restrict__ feat, int n, int nid_level,
                                                       uint32_t* __restrict__ word_out, uint32_t* __restrict__ node_out,
                                                       uint32_t* __restrict__ leaf_out, const int* __restrict__ rank = nullptr,
                                                       const uint8_t* __restrict__ stop = nullptr, int* __restrict__ bin_out = nullptr) {
    const int sub = threadIdx.x & 15;
    const int f = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    if (f >= n) return;
    const uint4 a0 = feat[2 * (size_t)f], a1 = feat[2 * (size_t)f + 1];
    int cur = 0, level = 0, bin = 0;   // bin: 0 = root (no node at nid_level on this path), else 1 + rank of that node by NodeId
    uint32_t nid = 0;
    int fc = first_child[0], fe = first_child[1];
    while (fe > fc) {
        ++level;
        unsigned best = 0xffffffffu;
        for (int c = fc + sub; c < fe; c += 16) {
            const int d = ham256(a0, a1, vdesc[2 * (size_t)c], vdesc[2 * (size_t)c + 1]);
            best = min(best, ((unsigned)d << 22) | (unsigned)(c - fc));   // first child wins a tie (strict '<', :1241)
        }
        best = min(best, (unsigned)__shfl_xor((int)best, 1, 16));
        best = min(best, (unsigned)__shfl_xor((int)best, 2, 16));
        best = min(best, (unsigned)__shfl_xor((int)best, 4, 16));
        best = min(best, (unsigned)__shfl_xor((int)best, 8, 16));
        cur = fc + (int)(best & 0x3fffffu);
        if (level == nid_level) { nid = orig[cur]; if (bin_out) bin = 1 + rank[cur]; }
        fc = first_child[cur]; fe = first_child[cur + 1];
    }
    if (sub == 0) {
        word_out[f] = word[cur]; node_out[f] = nid;
        if (leaf_out) leaf_out[f] = orig[cur];
        if (bin_out) bin_out[f] = stop[cur] ? -1 : bin;     // stopped words (weight 0) enter neither vector (:1156)
    }
}

// ---- FeatureVector on the device: histogram over the nodes of one tree level, compaction of the non-empty ones in
// ascending NodeId, then one wave per node collects its features in ascending index (ballot compaction: stable, no sort)
__global__ void k_fv_count(const int* __restrict__ bin, int n, int* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bin[i] >= 0) atomicAdd(&counts[bin[i]], 1);
}

// one workgroup: counts[nbins] -> node_id / node_start of the non-empty bins, bin_to_node, meta = {n_nodes, largest node, items}
__global__ __launch_bounds__(1024) void k_fv_offsets(const int* __restrict__ counts, int nbins, const uint32_t* __restrict__ orig_sorted,
                                                      uint32_t* __restrict__ node_id, int* __restrict__ node_start, int* __restrict__ bin_to_node,
                                                      int* __restrict__ meta) {
    __shared__ int s_nodes[1024], s_items[1024], s_base[2], s_max;
    const int t = threadIdx.x;
    if (t == 0) { s_base[0] = 0; s_base[1] = 0; s_max = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < nbins; b0 += 1024) {
        const int b = b0 + t;
        const int c = b < nbins ? counts[b] : 0;
        s_nodes[t] = c > 0; s_items[t] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {     // inclusive scans of both columns
            const int a = t >= o ? s_nodes[t - o] : 0, d = t >= o ? s_items[t - o] : 0;
            __syncthreads();
            s_nodes[t] += a; s_items[t] += d;
            __syncthreads();
        }
        if (b < nbins) {
            if (c > 0) {
                const int nd = s_base[0] + s_nodes[t] - 1;
                node_id[nd] = b == 0 ? 0u : orig_sorted[b - 1];
                node_start[nd] = s_base[1] + s_items[t] - c;
                bin_to_node[b] = nd;
                atomicMax(&s_max, c);
            } else bin_to_node[b] = -1;
        }
        __syncthreads();
        if (t == 1023) { s_base[0] += s_nodes[1023]; s_base[1] += s_items[1023]; }
        __syncthreads();
    }
    if (t == 0) { node_start[s_base[0]] = s_base[1]; meta[0] = s_base[0]; meta[1] = s_max; meta[2] = s_base[1]; }
}

__global__ __launch_bounds__(64) void k_fv_fill(const int* __restrict__ bin, int n, const int* __restrict__ bin_to_node,
                                                const int* __restrict__ node_start, uint32_t* __restrict__ items) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nd = bin_to_node[b];
    if (nd < 0) return;
    int pos = node_start[nd];
    for (int f0 = 0; f0 < n; f0 += 64) {
        const int f = f0 + lane;
        const bool mine = f < n && bin[f] == b;
        const unsigned long long m = __ballot(mine);
        if (mine) items[pos + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)f;
        pos += __popcll(m);
    }
}

// camera of a feature from the per-camera starts, flags from the right coordinate (bit0 usable, bit1 stereo: mvuRight >= 0)
struct CamStarts { int n_cams; int start[ORBV_MAX_CAMS + 1]; };
__global__ void k_side_misc(int n, CamStarts C, const float* __restrict__ uright, int32_t* __restrict__ cam_of, uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = 0;
    while (c + 1 < C.n_cams && i >= C.start[c + 1]) ++c;
    cam_of[i] = c;
    flags[i] = (uint8_t)(1 | ((uright && uright[i] >= 0) ? 2 : 0));
}

struct SideDev {
    int n, n_nodes;
    const uint4* desc; const float* angle; const uint8_t* flags; const uint32_t* node_id; const int32_t* node_start;
    const uint32_t* items; const float* x; const float* y; const int32_t* octave; const int32_t* cam_of;
};

struct TriDev {
    float F12[ORBV_MAX_CAMS][9];
    float ex[ORBV_MAX_CAMS], ey[ORBV_MAX_CAMS];
    float scale[MAX_LEVELS], sigma2[MAX_LEVELS];
};

struct JoinWork {
    int32_t* match;    // n_out
    uint8_t* bin_of;   // n_out: histogram bin of an accepted match
    int* hist;         // HISTO bins, then [HISTO] = accepted matches
};

__global__ void k_bow_init(JoinWork W, int n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) W.match[i] = -1;
    if (i <= HISTO) W.hist[i] = 0;
}

__device__ __forceinline__ int rot_bin(float a1, float a2) {
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / HISTO));
    if (bin == HISTO) bin = 0;
    return bin;
}

__device__ __forceinline__ int bcast_i(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ float bcast_f(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }
__device__ __forceinline__ uint4 bcast_u4(const uint4& v, int l) {
    return make_uint4((unsigned)bcast_i((int)v.x, l), (unsigned)bcast_i((int)v.y, l), (unsigned)bcast_i((int)v.z, l), (unsigned)bcast_i((int)v.w, l));
}

// MODE 0: SearchByBoW(KF, F); 1: SearchByBoW(KF, KF); 2: SearchForTriangulation.
// One wavefront per node of A that B also has.  LDS: one "claimed" byte per candidate of the node, then the descriptors of
// its first `lds_cand` candidates as two uint4 planes (conflict-free 16-byte lane stride) with their index and angle; the rest are read
// from HBM/L2 on every pass.  Queries are fetched 64 at a time (one per lane: index, flags, descriptor, angle, position) and
// handed to the whole wave with readlane, so the serial query loop waits on no memory but LDS.
// NW = 1: one wavefront per node (small nodes).  NW = 4 (nodes of a few hundred features, the 8-camera sizes): four waves per
// node.  SearchByBoW (MODE 0 / 1) keeps its queries strictly in order -- a claim hides a candidate from the queries behind it --
// so the four waves split every query's CANDIDATES (stripes of 64), reduce their stripes on their own, meet at one workgroup
// barrier per query to combine the four partial top-2s (every wave computes the same combination: all further decisions are
// workgroup-uniform) and at a second one only when a match was accepted (the claim must be visible before the next scan).
// SearchForTriangulation (MODE 2) has no claims: the waves simply take every fourth query each, no barriers at all.
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_bow_join(SideDev A, SideDev B, TriDev T, int th_low, float nnratio, int check_ori, JoinWork W,
                                                      int claimed_bytes, int lds_cand) {
    extern __shared__ uint4 s_dyn[];
    __shared__ unsigned s_part[2][4][2];   // NW > 1: {best key, second distance} of every wave's stripe, double-buffered by query
    // one wave per workgroup: its LDS operations execute in program order, so a claim written by lane 0 is seen by every lane's
    // next read without a barrier -- volatile keeps the compiler from caching or reordering them.  (A __syncthreads() here
    // would also wait for the match / histogram stores of the accept path to reach L2: ~1 us per accepted match.)
    volatile uint8_t* s_claimed = (volatile uint8_t*)s_dyn;
    uint4* s_lo = s_dyn + claimed_bytes / 16;
    uint4* s_hi = s_lo + lds_cand;
    int* s_idx = (int*)(s_hi + lds_cand);          // feature index and angle of the staged candidates: the accept path
    float* s_ang = (float*)(s_idx + lds_cand);     // of a query touches no global memory but its (unwaited) stores
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int an = blockIdx.x;
    const uint32_t id = A.node_id[an];
    int lo = 0, hi = B.n_nodes;   // FeatureVector::lower_bound
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (B.node_id[mid] < id) lo = mid + 1; else hi = mid;
    }
    if (lo >= B.n_nodes || B.node_id[lo] != id) return;
    const int qa0 = A.node_start[an], qa1 = A.node_start[an + 1];
    const int cb0 = B.node_start[lo], nc = B.node_start[lo + 1] - cb0;
    if (nc <= 0 || qa1 <= qa0) return;
    const int staged = min(nc, lds_cand);
    for (int j = (int)threadIdx.x; j < nc; j += 64 * NW) {
        const int idx2 = (int)B.items[cb0 + j];
        bool usable = true;
        if (MODE != 0 && B.flags) usable = (B.flags[idx2] & 1) != 0;
        s_claimed[j] = usable ? 0 : 1;
        if (j < staged) { s_lo[j] = B.desc[2 * (size_t)idx2]; s_hi[j] = B.desc[2 * (size_t)idx2 + 1]; s_idx[j] = idx2; s_ang[j] = B.angle[idx2]; }
    }
    __syncthreads();
    unsigned parity = 0;   // NW > 1, MODE 0 / 1: which s_part buffer the next query that reaches the barrier uses (alternates
                           // over the whole node: a buffer is rewritten only after a barrier every reader of it has passed)
    for (int kb = qa0; kb < qa1; kb += 64) {
        // this lane's query of the block
        const int kmine = kb + lane;
        const bool have = kmine < qa1;
        const int p_idx1 = have ? (int)A.items[kmine] : 0;
        const int p_fl = have ? (A.flags ? (int)A.flags[p_idx1] : 1) : 0;
        uint4 p_q0 = make_uint4(0, 0, 0, 0), p_q1 = p_q0;
        float p_ang = 0.f, p_x = 0.f, p_y = 0.f; int p_cam = 0;
        if (have && (p_fl & 1)) {
            p_q0 = A.desc[2 * (size_t)p_idx1]; p_q1 = A.desc[2 * (size_t)p_idx1 + 1];
            p_ang = A.angle[p_idx1];
            if (MODE == 2) { p_x = A.x[p_idx1]; p_y = A.y[p_idx1]; p_cam = A.cam_of[p_idx1]; }
        }
        const int kend = min(64, qa1 - kb);
        for (int kq = (MODE == 2 && NW > 1) ? wave : 0; kq < kend; kq += (MODE == 2 && NW > 1) ? NW : 1) {
            const int fl1 = bcast_i(p_fl, kq);
            if (!(fl1 & 1)) continue;                      // wave-uniform
            const int idx1 = bcast_i(p_idx1, kq);
            const uint4 q0 = bcast_u4(p_q0, kq), q1 = bcast_u4(p_q1, kq);
            const float ang1 = bcast_f(p_ang, kq);
            int cam1 = 0; float la = 0, lb = 0, lc = 0, den = 0;
            if (MODE == 2) {
                cam1 = bcast_i(p_cam, kq);
                const float x1 = bcast_f(p_x, kq), y1 = bcast_f(p_y, kq);
                const float* F = T.F12[cam1];              // CheckDistEpipolarLine, src/ORBmatcher.cc:170-178
                la = x1 * F[0] + y1 * F[3] + F[6];
                lb = x1 * F[1] + y1 * F[4] + F[7];
                lc = x1 * F[2] + y1 * F[5] + F[8];
                den = la * la + lb * lb;
            }
            int bd = 256, bj = -1, d2 = 256;               // per-lane top-2 (modes 0/1)
            unsigned key2 = 0x7fffffffu;                   // per-lane best (mode 2): smallest distance, LAST candidate on ties
            // geometric gates of SearchForTriangulation for one candidate that is already close enough (:1562-1600)
            auto tri_ok = [&](int idx2) -> bool {
                if (B.cam_of[idx2] != cam1) return false;
                const int fl2 = B.flags ? B.flags[idx2] : 1;
                const float x2 = B.x[idx2], y2 = B.y[idx2];
                const int oct2 = B.octave[idx2];
                if (!(fl1 & 2) && !(fl2 & 2)) {                                    // both monocular: too close to the epipole
                    const float dex = T.ex[cam1] - x2, dey = T.ey[cam1] - y2;
                    if (dex * dex + dey * dey < 100 * T.scale[oct2]) return false;
                }
                const float num = la * x2 + lb * y2 + lc;
                if (den == 0) return false;
                const float dsqr = num * num / den;
                return (double)dsqr < 3.84 * (double)T.sigma2[oct2];
            };
            // staged candidates: branch-free LDS reads (lanes past the end re-read the last one and are masked), so the passes
            // of one query overlap instead of waiting on each other's LDS round trips
            constexpr int STRIPE = (MODE != 2 && NW > 1) ? NW : 1;   // waves that share one query's candidates
            const int w0 = STRIPE > 1 ? wave * 64 : 0;
#pragma unroll 2
            for (int j0 = w0; j0 < staged; j0 += 64 * STRIPE) {
                const int j = j0 + lane;
                const int jc = min(j, staged - 1);
                const bool ok = j < staged && !s_claimed[jc];
                const uint4 c0 = s_lo[jc], c1 = s_hi[jc];
                const int d = ok ? ham256(q0, q1, c0, c1) : 257;
                if (MODE != 2) {
                    if (d < bd) { d2 = bd; bd = d; bj = j; }
                    else if (d < d2) d2 = d;
                } else if (d <= th_low && tri_ok(s_idx[jc])) {
                    key2 = min(key2, ((unsigned)d << 20) | (unsigned)(0xfffff - j));
                }
            }
            for (int j = staged + w0 + lane; j < nc; j += 64 * STRIPE) {   // nodes larger than the LDS stage: the rest from L2 / HBM
                if (s_claimed[j]) continue;
                const int idx2 = (int)B.items[cb0 + j];
                const int d = ham256(q0, q1, B.desc[2 * (size_t)idx2], B.desc[2 * (size_t)idx2 + 1]);
                if (MODE != 2) {
                    if (d < bd) { d2 = bd; bd = d; bj = j; }
                    else if (d < d2) d2 = d;
                } else if (d <= th_low && tri_ok(idx2)) {
                    key2 = min(key2, ((unsigned)d << 20) | (unsigned)(0xfffff - j));
                }
            }
            if (MODE != 2) {
                unsigned K = wave_min_u32(bj >= 0 ? (((unsigned)bd << 20) | (unsigned)bj) : 0x7fffffffu);
                int second;
                if (STRIPE == 1) {
                    if (K == 0x7fffffffu) continue;            // wave-uniform: nothing closer than 256
                    second = (int)wave_min_u32((unsigned)((((int)(K & 0xfffffu) & 63) == lane) ? d2 : bd));
                } else {
                    // this wave's stripe: its best key and the runner-up inside the stripe (candidate j sits in lane j % 64)
                    const unsigned S = wave_min_u32((unsigned)((K != 0x7fffffffu && ((int)(K & 0xfffffu) & 63) == lane) ? d2 : bd));
                    if (lane == 0) { s_part[parity][wave][0] = K; s_part[parity][wave][1] = S; }
                    __syncthreads();
                    unsigned kk[4], ss[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) { kk[w] = w < NW ? s_part[parity][w][0] : 0x7fffffffu; ss[w] = w < NW ? s_part[parity][w][1] : 256u; }
                    parity ^= 1u;
                    K = min(min(kk[0], kk[1]), min(kk[2], kk[3]));
                    if (K == 0x7fffffffu) continue;            // workgroup-uniform
                    unsigned sec = 256u;                        // the winner's stripe gives its runner-up, every other stripe its best
#pragma unroll
                    for (int w = 0; w < 4; ++w) sec = min(sec, kk[w] == K ? ss[w] : min(kk[w] >> 20, 256u));
                    second = (int)sec;
                }
                const int best = (int)(K >> 20), J = (int)(K & 0xfffffu);
                const bool under = MODE == 0 ? best <= th_low : best < th_low;        // :324 / :1107
                if (under && (float)best < nnratio * (float)second) {
                    if (lane == 0 && (STRIPE == 1 || wave == 0)) {
                        const int idx2 = J < staged ? s_idx[J] : (int)B.items[cb0 + J];
                        const int oi = MODE == 0 ? idx2 : idx1;
                        W.match[oi] = MODE == 0 ? idx1 : idx2;
                        s_claimed[J] = 1;
                        if (check_ori) {
                            const int bin = rot_bin(ang1, J < staged ? s_ang[J] : B.angle[idx2]);
                            W.bin_of[oi] = (uint8_t)bin;
                            atomicAdd(&W.hist[bin], 1);
                        }
                        atomicAdd(&W.hist[HISTO], 1);
                    }
                    if (STRIPE > 1) __syncthreads();   // (workgroup-uniform branch) the claim before anybody's next scan
                }
            } else {
                const unsigned K = wave_min_u32(key2);
                if (K == 0x7fffffffu) continue;
                if (lane == 0) {
                    const int J = 0xfffff - (int)(K & 0xfffffu);
                    const int idx2 = J < staged ? s_idx[J] : (int)B.items[cb0 + J];
                    W.match[idx1] = idx2;
                    if (check_ori) {
                        const int bin = rot_bin(ang1, J < staged ? s_ang[J] : B.angle[idx2]);
                        W.bin_of[idx1] = (uint8_t)bin;
                        atomicAdd(&W.hist[bin], 1);
                    }
                    atomicAdd(&W.hist[HISTO], 1);
                }
            }
        }
    }
}

// Every workgroup works the three maxima out for itself (30 bins from L2) and filters its own 256 results: no single
// workgroup walking the whole array through sixteen dependent trips to memory.  The match count is taken by the host while it
// copies the array out (it touches every word anyway).
__global__ __launch_bounds__(256) void k_bow_finish(JoinWork W, int n_out, int check_ori, int32_t* __restrict__ h_match) {
    __shared__ int s_keep[3];
    __shared__ int s_hist[HISTO];
    if (threadIdx.x < HISTO) s_hist[threadIdx.x] = W.hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        // ComputeThreeMaxima, src/ORBmatcher.cc:3948-3989
        int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < HISTO; i++) {
            const int s = s_hist[i];
            if (s > m1) { m3 = m2; i3 = i2; m2 = m1; i2 = i1; m1 = s; i1 = i; }
            else if (s > m2) { m3 = m2; i3 = i2; m2 = s; i2 = i; }
            else if (s > m3) { m3 = s; i3 = i; }
        }
        if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
        else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    int mt = W.match[i];
    if (check_ori && mt >= 0) {
        const int b = W.bin_of[i];
        if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) mt = -1;
    }
    h_match[i] = mt;
}

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

struct orbv_vocabulary {
    int device = 0, L = 0, k = 0, n_nodes = 0, n_words = 0;
    hipStream_t stream = nullptr;
    DevBuf<uint4> d_desc;            // 2 per node, breadth-first order
    DevBuf<int32_t> d_first_child;   // n_nodes + 1: children of node d are [first_child[d], first_child[d+1])
    DevBuf<uint32_t> d_orig, d_word; // NodeId / WordId of a breadth-first index
    std::vector<double> weight;      // by NodeId
    std::vector<int> lvl_start;      // breadth-first index range of every depth (depth d: [lvl_start[d], lvl_start[d+1]))
    DevBuf<int32_t> d_rank;          // rank of a node among the nodes of its depth, by NodeId
    DevBuf<uint32_t> d_orig_sorted;  // NodeIds of every depth in ascending order (same index space as the breadth-first one)
    DevBuf<uint8_t> d_stop;          // weight <= 0: a stopped word
    DevBuf<uint8_t> d_feat;
    DevBuf<uint32_t> d_out;
    PinnedBuf<uint32_t> h_out;
    PinnedBuf<uint8_t> h_feat;
};

struct orbv_workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    PinnedBuf<uint8_t> h_stage;
    DevBuf<uint8_t> d_stage, d_work;
    PinnedBuf<int32_t> h_match;
};

extern "C" {

int orbv_create(int n_nodes, int L, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                int device, orbv_vocabulary** out) {
    MORB_ARG(out != nullptr && n_nodes >= 2 && L >= 1 && parent && is_leaf && desc && weight);
    for (int i = 1; i < n_nodes; ++i) MORB_ARG(parent[i] >= 0 && parent[i] < n_nodes && parent[i] != i);
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    // children in ascending id (the loader pushes them back in file order, TemplatedVocabulary.h:1386-1393)
    std::vector<int32_t> cstart(n_nodes + 1, 0), clist(n_nodes - 1);
    for (int i = 1; i < n_nodes; ++i) cstart[parent[i] + 1]++;
    for (int i = 0; i < n_nodes; ++i) cstart[i + 1] += cstart[i];
    MORB_ARG(cstart[1] > 0);   // the root has children (the reference indexes children[0] unconditionally, :1236)
    { std::vector<int32_t> fill(cstart.begin(), cstart.end() - 1);
      for (int i = 1; i < n_nodes; ++i) clist[fill[parent[i]]++] = i; }
    std::vector<uint32_t> word_of(n_nodes, 0);   // Node(): word_id(0); set for the flagged nodes in id order (:1409-1416)
    uint32_t words = 0;
    for (int i = 1; i < n_nodes; ++i) if (is_leaf[i]) word_of[i] = words++;
    // breadth-first renumbering from the root: every node's children become one contiguous run
    std::vector<int32_t> order; order.reserve(n_nodes); order.push_back(0);
    std::vector<int32_t> first_child(n_nodes + 1, 0);
    int kmax = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const int id = order[h];
        first_child[h] = (int32_t)order.size();
        for (int c = cstart[id]; c < cstart[id + 1]; ++c) order.push_back(clist[c]);
        kmax = std::max(kmax, cstart[id + 1] - cstart[id]);
    }
    const int reach = (int)order.size();   // nodes not reachable from the root (cycles among themselves) can never be visited
    first_child[reach] = reach;
    orbv_vocabulary* v = new orbv_vocabulary();
    v->device = device; v->L = L; v->k = kmax; v->n_nodes = n_nodes; v->n_words = (int)words;
    v->weight.assign(weight, weight + n_nodes); v->weight[0] = 0.0;
    std::vector<uint8_t> bdesc((size_t)reach * 32);
    std::vector<uint32_t> borig(reach), bword(reach);
    for (int h = 0; h < reach; ++h) {
        const int id = order[h];
        if (id) memcpy(&bdesc[(size_t)h * 32], desc + (size_t)id * 32, 32); else memset(&bdesc[0], 0, 32);
        borig[h] = (uint32_t)id; bword[h] = word_of[id];
    }
    // depth ranges + rank by NodeId inside each depth (the device-side FeatureVector bins)
    std::vector<int> depth(reach, 0);
    for (int h = 0; h < reach; ++h) for (int c = first_child[h]; c < first_child[h + 1]; ++c) depth[c] = depth[h] + 1;
    v->lvl_start.assign(1, 0);
    for (int h = 1; h < reach; ++h) if (depth[h] != depth[h - 1]) v->lvl_start.push_back(h);
    v->lvl_start.push_back(reach);
    std::vector<int32_t> brank(reach, 0); std::vector<uint32_t> bsorted(reach, 0); std::vector<uint8_t> bstop(reach, 0);
    for (size_t d = 0; d + 1 < v->lvl_start.size(); ++d) {
        std::vector<int> idx(v->lvl_start[d + 1] - v->lvl_start[d]);
        std::iota(idx.begin(), idx.end(), v->lvl_start[d]);
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return borig[a] < borig[b]; });
        for (size_t r = 0; r < idx.size(); ++r) { brank[idx[r]] = (int32_t)r; bsorted[v->lvl_start[d] + r] = borig[idx[r]]; }
    }
    for (int h = 0; h < reach; ++h) bstop[h] = v->weight[borig[h]] > 0 ? 0 : 1;
    hipError_t e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete v; return ORB_E_HIP; }
    if ((rc = v->d_rank.reserve(reach)) || (rc = v->d_orig_sorted.reserve(reach)) || (rc = v->d_stop.reserve(reach))) { orbv_destroy(v); return rc; }
    if (hipMemcpyAsync(v->d_rank.p, brank.data(), (size_t)reach * 4, hipMemcpyHostToDevice, v->stream) != hipSuccess ||
        hipMemcpyAsync(v->d_orig_sorted.p, bsorted.data(), (size_t)reach * 4, hipMemcpyHostToDevice, v->stream) != hipSuccess ||
        hipMemcpyAsync(v->d_stop.p, bstop.data(), (size_t)reach, hipMemcpyHostToDevice, v->stream) != hipSuccess) {
        morb::set_error("vocabulary upload failed"); orbv_destroy(v); return ORB_E_HIP;
    }
    if ((rc = v->d_desc.reserve((size_t)reach * 2)) || (rc = v->d_first_child.reserve(reach + 1)) || (rc = v->d_orig.reserve(reach)) ||
        (rc = v->d_word.reserve(reach))) { orbv_destroy(v); return rc; }
    if (hipMemcpyAsync(v->d_desc.p, bdesc.data(), bdesc.size(), hipMemcpyHostToDevice, v->stream) != hipSuccess ||
        hipMemcpyAsync(v->d_first_child.p, first_child.data(), (size_t)(reach + 1) * 4, hipMemcpyHostToDevice, v->stream) != hipSuccess ||
        hipMemcpyAsync(v->d_orig.p, borig.data(), (size_t)reach * 4, hipMemcpyHostToDevice, v->stream) != hipSuccess ||
        hipMemcpyAsync(v->d_word.p, bword.data(), (size_t)reach * 4, hipMemcpyHostToDevice, v->stream) != hipSuccess) {
        morb::set_error("vocabulary upload failed"); orbv_destroy(v); return ORB_E_HIP;
    }
    if (hipStreamSynchronize(v->stream) != hipSuccess) { morb::set_error("vocabulary upload failed"); orbv_destroy(v); return ORB_E_HIP; }  // (the sources are locals)
    *out = v;
    return ORB_OK;
}

int orbv_load_text(const char* path, int device, orbv_vocabulary** out) {
    MORB_ARG(path != nullptr && out != nullptr);
    std::ifstream f(path);
    if (!f.good()) { morb::set_error("cannot open vocabulary file %s", path); return ORB_E_ARG; }
    std::string s;
    std::getline(f, s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    { std::stringstream ss(s); ss >> k >> L >> n1 >> n2; }
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {   // the reference's sanity check (:1360)
        morb::set_error("vocabulary loading failure: %s is not a correct text file", path); return ORB_E_ARG;
    }
    if (n1 != 0 || n2 != 0) { morb::set_error("only L1_NORM scoring with TF_IDF weighting (ORBvoc.txt) is supported, file declares %d %d", n1, n2); return ORB_E_ARG; }
    std::vector<int32_t> parent(1, 0); std::vector<uint8_t> leaf(1, 0), desc(32, 0); std::vector<double> weight(1, 0.0);
    while (std::getline(f, s)) {
        const char* p = s.c_str(); char* q = nullptr;
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (!*p) continue;                                  // empty line (see orbv.h)
        const long pid = strtol(p, &q, 10); p = q;
        const long il = strtol(p, &q, 10); p = q;
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) { d[i] = (uint8_t)strtol(p, &q, 10); p = q; }
        const double w = strtod(p, &q);
        if (q == p) { morb::set_error("vocabulary line %zu is malformed", parent.size()); return ORB_E_ARG; }
        parent.push_back((int32_t)pid); leaf.push_back(il > 0); desc.insert(desc.end(), d, d + 32); weight.push_back(w);
    }
    return orbv_create((int)parent.size(), L, parent.data(), leaf.data(), desc.data(), weight.data(), device, out);
}

void orbv_destroy(orbv_vocabulary* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
    v->d_desc.release(); v->d_first_child.release(); v->d_orig.release(); v->d_word.release(); v->d_feat.release(); v->d_out.release();
    v->d_rank.release(); v->d_orig_sorted.release(); v->d_stop.release();
    v->h_out.release(); v->h_feat.release();
    delete v;
}

int orbv_info(const orbv_vocabulary* v, int* n_nodes, int* n_words, int* k, int* L) {
    MORB_ARG(v != nullptr);
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    if (k) *k = v->k;
    if (L) *L = v->L;
    return ORB_OK;
}

void* orbv_stream(orbv_vocabulary* v) { return v ? (void*)v->stream : nullptr; }

static int transform_enqueue(const orbv_vocabulary* v, const uint8_t* d_features, int n, int levelsup, uint32_t* d_word,
                             uint32_t* d_node, uint32_t* d_leaf, hipStream_t st) {
    if (n == 0) return ORB_OK;
    k_bow_transform<<<(n + 15) / 16, 256, 0, st>>>(v->d_desc.p, v->d_first_child.p, v->d_orig.p, v->d_word.p, (const uint4*)d_features, n,
                                                   v->L - levelsup, d_word, d_node, d_leaf);
    MORB_HIP(hipGetLastError());
    return ORB_OK;
}

int orbv_transform_device(const orbv_vocabulary* v, const uint8_t* d_features, int n, int levelsup, uint32_t* d_word_id,
                          uint32_t* d_node_id, void* stream) {
    MORB_ARG(v != nullptr && n >= 0 && (n == 0 || (d_features && d_word_id && d_node_id)) && ((uintptr_t)d_features & 15) == 0);
    return transform_enqueue(v, d_features, n, levelsup, d_word_id, d_node_id, nullptr, (hipStream_t)stream);
}

// descents of n host features; results in v->h_out: [0,n) word, [n,2n) node, [2n,3n) leaf NodeId
static int transform_host(orbv_vocabulary* v, const uint8_t* features, int n, int levelsup) {
    MORB_HIP(hipSetDevice(v->device));
    int rc;
    if ((rc = v->h_feat.reserve((size_t)n * 32)) || (rc = v->d_feat.reserve((size_t)n * 32)) || (rc = v->d_out.reserve((size_t)n * 3)) ||
        (rc = v->h_out.reserve((size_t)n * 3))) return rc;
    memcpy(v->h_feat.p, features, (size_t)n * 32);
    MORB_HIP(hipMemcpyAsync(v->d_feat.p, v->h_feat.p, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    if ((rc = transform_enqueue(v, v->d_feat.p, n, levelsup, v->d_out.p, v->d_out.p + n, v->d_out.p + 2 * (size_t)n, v->stream))) return rc;
    MORB_HIP(hipMemcpyAsync(v->h_out.p, v->d_out.p, (size_t)n * 12, hipMemcpyDeviceToHost, v->stream));
    MORB_HIP(hipStreamSynchronize(v->stream));
    return ORB_OK;
}

int orbv_transform(orbv_vocabulary* v, const uint8_t* features, int n, int levelsup, uint32_t* word_id, uint32_t* node_id, double* weight) {
    MORB_ARG(v != nullptr && n >= 0 && (n == 0 || (features && word_id && node_id)));
    if (n == 0) return ORB_OK;
    int rc = transform_host(v, features, n, levelsup);
    if (rc) return rc;
    memcpy(word_id, v->h_out.p, (size_t)n * 4);
    memcpy(node_id, v->h_out.p + n, (size_t)n * 4);
    if (weight) for (int i = 0; i < n; ++i) weight[i] = v->weight[v->h_out.p[2 * (size_t)n + i]];
    return ORB_OK;
}

int orbv_bow_vectors(orbv_vocabulary* v, const uint8_t* features, int n, int levelsup, uint32_t* bow_id, double* bow_val,
                     int* n_words, uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes) {
    MORB_ARG(v != nullptr && n >= 0 && n_words && n_fv_nodes && fv_start && (n == 0 || (features && bow_id && bow_val && fv_node && fv_items)));
    *n_words = 0; *n_fv_nodes = 0; fv_start[0] = 0;
    if (n == 0) return ORB_OK;
    int rc = transform_host(v, features, n, levelsup);
    if (rc) return rc;
    const uint32_t* word = v->h_out.p; const uint32_t* node = v->h_out.p + n; const uint32_t* leaf = v->h_out.p + 2 * (size_t)n;
    std::vector<int> keep; keep.reserve(n);
    for (int i = 0; i < n; ++i) if (v->weight[leaf[i]] > 0) keep.push_back(i);   // "not stopped" (:1156)
    // BowVector::addWeight per feature in feature order == per word, the weights of its features added one by one
    std::vector<int> byw(keep);
    std::stable_sort(byw.begin(), byw.end(), [&](int a, int b) { return word[a] < word[b]; });
    int nw = 0;
    for (size_t s = 0; s < byw.size();) {
        size_t e = s;
        double acc = v->weight[leaf[byw[s]]];           // insert(id, w)
        for (e = s + 1; e < byw.size() && word[byw[e]] == word[byw[s]]; ++e) acc += v->weight[leaf[byw[e]]];   // vit->second += w
        bow_id[nw] = word[byw[s]]; bow_val[nw] = acc; ++nw;
        s = e;
    }
    double norm = 0.0;                                  // BowVector::normalize(L1), ascending word id
    for (int i = 0; i < nw; ++i) norm += std::fabs(bow_val[i]);
    if (norm > 0.0) for (int i = 0; i < nw; ++i) bow_val[i] /= norm;
    *n_words = nw;
    // FeatureVector::addFeature: nodes ascending, features of a node in feature order
    std::vector<int> byn(keep);
    std::stable_sort(byn.begin(), byn.end(), [&](int a, int b) { return node[a] < node[b]; });
    int nn = 0, off = 0;
    for (size_t s = 0; s < byn.size();) {
        size_t e = s;
        fv_node[nn] = node[byn[s]]; fv_start[nn] = off;
        for (; e < byn.size() && node[byn[e]] == node[byn[s]]; ++e) fv_items[off++] = (uint32_t)byn[e];
        ++nn; s = e;
    }
    fv_start[nn] = off;
    *n_fv_nodes = nn;
    return ORB_OK;
}

double orbv_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2) {
    // L1Scoring::score over two id-sorted sparse vectors (ScoringObject.cpp:23-68)
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (id1[i] == id2[j]) { score += std::fabs(v1[i] - v2[j]) - std::fabs(v1[i]) - std::fabs(v2[j]); ++i; ++j; }
        else if (id1[i] < id2[j]) i = (int)(std::lower_bound(id1 + i, id1 + n1, id2[j]) - id1);
        else j = (int)(std::lower_bound(id2 + j, id2 + n2, id1[i]) - id2);
    }
    return -score / 2.0;
}

int orbv_workspace_create(int device, orbv_workspace** out) {
    MORB_ARG(out != nullptr);
    int rc = morb::select_device(device);
    if (rc != ORB_OK) return rc;
    orbv_workspace* w = new orbv_workspace();
    w->device = device;
    hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { morb::set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete w; return ORB_E_HIP; }
    *out = w;
    return ORB_OK;
}

void orbv_workspace_destroy(orbv_workspace* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->stream) { (void)hipStreamSynchronize(w->stream); (void)hipStreamDestroy(w->stream); }
    w->h_stage.release(); w->d_stage.release(); w->d_work.release(); w->h_match.release();
    delete w;
}

}  // extern "C"

namespace {

struct Packer {   // lays the arrays of both sides out in one pinned block, mirrored by one device block
    size_t off = 0;
    uint8_t* h = nullptr; uint8_t* d = nullptr;
    size_t reserve_bytes(size_t bytes) { const size_t o = off; off = up16(off + bytes); return o; }
    template <typename T> const T* put(const T* src, size_t count) {
        if (!src) return nullptr;
        const size_t o = reserve_bytes(count * sizeof(T));
        if (h) memcpy(h + o, src, count * sizeof(T));
        return (const T*)(d + o);
    }
};

int side_check(const orbv_side* s, bool tri) {
    MORB_ARG(s != nullptr && s->n >= 0 && s->n_nodes >= 0);
    if (s->n_nodes > 0) {
        MORB_ARG(s->node_id && s->node_start && s->node_start[0] == 0);
        for (int k = 0; k < s->n_nodes; ++k) {
            MORB_ARG(s->node_start[k + 1] >= s->node_start[k]);
            MORB_ARG(k == 0 || s->node_id[k] > s->node_id[k - 1]);
        }
        const int m = s->node_start[s->n_nodes];
        MORB_ARG(m <= s->n && (m == 0 || s->items));
        for (int i = 0; i < m; ++i) MORB_ARG(s->items[i] < (uint32_t)s->n);
    }
    if (s->n > 0) { MORB_ARG(s->desc && s->angle); if (tri) MORB_ARG(s->x && s->y && s->octave && s->cam_of); }
    return ORB_OK;
}

SideDev pack_side(Packer& P, const orbv_side* s, bool tri) {
    SideDev D;
    D.n = s->n; D.n_nodes = s->n_nodes;
    D.desc = (const uint4*)P.put(s->desc, (size_t)s->n * 32);
    D.angle = P.put(s->angle, s->n);
    D.flags = P.put(s->flags, s->n);
    D.node_id = P.put(s->node_id, s->n_nodes);
    D.node_start = P.put(s->node_start, (size_t)s->n_nodes + 1);
    D.items = P.put(s->items, s->n_nodes ? (size_t)s->node_start[s->n_nodes] : 0);
    D.x = tri ? P.put(s->x, s->n) : nullptr; D.y = tri ? P.put(s->y, s->n) : nullptr;
    D.octave = tri ? P.put(s->octave, s->n) : nullptr; D.cam_of = tri ? P.put(s->cam_of, s->n) : nullptr;
    return D;
}

int tri_params(const orbv_triangulation* t, TriDev& T) {
    MORB_ARG(t != nullptr && t->n_cams >= 1 && t->n_cams <= ORBV_MAX_CAMS && t->n_levels >= 1 && t->n_levels <= MAX_LEVELS && t->scale_factors && t->level_sigma2);
    memcpy(T.F12, t->F12, sizeof(T.F12)); memcpy(T.ex, t->ex, sizeof(T.ex)); memcpy(T.ey, t->ey, sizeof(T.ey));
    memcpy(T.scale, t->scale_factors, t->n_levels * sizeof(float)); memcpy(T.sigma2, t->level_sigma2, t->n_levels * sizeof(float));
    return ORB_OK;
}

int max_node_of(const orbv_side* s) {
    int m = 1;
    for (int k = 0; k < s->n_nodes; ++k) m = std::max(m, s->node_start[k + 1] - s->node_start[k]);
    return m;
}

// init + join + finish on the workspace's stream, one synchronisation, results out of pinned memory
int launch_join(orbv_workspace* w, const SideDev& A, const SideDev& B, int max_nc, int mode, const TriDev& T, int th_low, float nnratio,
                int check_ori, int32_t* match, int* nmatches) {
    const int n_out = mode == 0 ? B.n : A.n;
    if (max_nc > JOIN_MAX_NODE) { morb::set_error("a vocabulary node holds %d features (limit %d)", max_nc, JOIN_MAX_NODE); return ORB_E_CAPACITY; }
    int rc;
    const size_t work_bytes = up16((size_t)n_out * 4) + up16((HISTO + 1) * 4) + up16((size_t)n_out);
    if ((rc = w->d_work.reserve(work_bytes)) || (rc = w->h_match.reserve((size_t)n_out + 4))) return rc;
    JoinWork W;
    W.match = (int32_t*)w->d_work.p;
    W.hist = (int*)(w->d_work.p + up16((size_t)n_out * 4));
    W.bin_of = w->d_work.p + up16((size_t)n_out * 4) + up16((HISTO + 1) * 4);
    hipStream_t st = w->stream;
    k_bow_init<<<(std::max(n_out, HISTO + 1) + 255) / 256, 256, 0, st>>>(W, n_out);
    const int claimed_bytes = (max_nc + 63) & ~63;
    int lds_cand = std::max(0, std::min(max_nc, (JOIN_LDS_BYTES - claimed_bytes) / 40));   // candidates staged in LDS (40 B each)
    if (lds_cand < max_nc) lds_cand &= ~63;   // a partial stage ends on a lane-0 boundary: candidate j always belongs to lane j % 64
    const size_t lds = (size_t)claimed_bytes + (size_t)lds_cand * 40;
    // four waves per node from ~128 candidates per node on (the barriers cost more than they save below that)
    const bool wide = max_nc >= 128;
    if (wide) {
        if (mode == 0) k_bow_join<0, 4><<<A.n_nodes, 256, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
        else if (mode == 1) k_bow_join<1, 4><<<A.n_nodes, 256, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
        else k_bow_join<2, 4><<<A.n_nodes, 256, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
    } else {
        if (mode == 0) k_bow_join<0, 1><<<A.n_nodes, 64, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
        else if (mode == 1) k_bow_join<1, 1><<<A.n_nodes, 64, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
        else k_bow_join<2, 1><<<A.n_nodes, 64, lds, st>>>(A, B, T, th_low, nnratio, check_ori, W, claimed_bytes, lds_cand);
    }
    k_bow_finish<<<(n_out + 255) / 256, 256, 0, st>>>(W, n_out, check_ori, w->h_match.dp);
    MORB_HIP(hipGetLastError());
    MORB_HIP(hipStreamSynchronize(st));
    int nm = 0;
    for (int i = 0; i < n_out; ++i) { const int v = w->h_match.p[i]; match[i] = v; nm += v >= 0; }
    *nmatches = nm;   // == accepted - removed by the rotation filter: every accepted match owns one output word
    return ORB_OK;
}

int run_join(orbv_workspace* w, const orbv_side* a, const orbv_side* b, int mode, const orbv_triangulation* t, int th_low,
             float nnratio, int check_ori, int32_t* match, int* nmatches) {
    MORB_ARG(w != nullptr && nmatches != nullptr && mode >= 0 && mode <= 2);
    const bool tri = mode == 2;
    int rc;
    if ((rc = side_check(a, tri)) || (rc = side_check(b, tri))) return rc;
    const int n_out = mode == 0 ? b->n : a->n;
    MORB_ARG(n_out == 0 || match != nullptr);
    TriDev T; memset(&T, 0, sizeof(T));
    if (tri) {
        if ((rc = tri_params(t, T))) return rc;
        for (int i = 0; i < a->n; ++i) MORB_ARG(a->cam_of[i] >= 0 && a->cam_of[i] < t->n_cams);
        for (int i = 0; i < b->n; ++i) MORB_ARG(b->octave[i] >= 0 && b->octave[i] < t->n_levels);
    }
    *nmatches = 0;
    for (int i = 0; i < n_out; ++i) match[i] = -1;
    if (a->n_nodes == 0 || b->n_nodes == 0 || a->n == 0 || b->n == 0) return ORB_OK;
    MORB_HIP(hipSetDevice(w->device));
    Packer size_pass;                                   // first pass: sizes only
    (void)pack_side(size_pass, a, tri); (void)pack_side(size_pass, b, tri);
    if ((rc = w->h_stage.reserve(size_pass.off)) || (rc = w->d_stage.reserve(size_pass.off))) return rc;
    Packer P; P.h = w->h_stage.p; P.d = w->d_stage.p;
    const SideDev A = pack_side(P, a, tri), B = pack_side(P, b, tri);
    MORB_HIP(hipMemcpyAsync(w->d_stage.p, w->h_stage.p, P.off, hipMemcpyHostToDevice, w->stream));
    return launch_join(w, A, B, max_node_of(b), mode, T, th_low, nnratio, check_ori, match, nmatches);
}

}  // namespace

// One frame / keyframe resident in HBM for any number of searches (descriptors, angles, FeatureVector, and the triangulation
// arrays when given): a keyframe is searched against ~20 covisible neighbours by LocalMapping alone (src/LocalMapping.cc).
struct orbv_keyframe {
    int device = 0;
    DevBuf<uint8_t> block;
    SideDev D;
    int max_node = 1, max_cam = 0, max_octave = 0;
    bool tri = false;
    const uint32_t* d_word = nullptr; const uint32_t* d_node = nullptr;   // per-feature descent results (device-built keyframes)
};

namespace {

int run_join_resident(orbv_workspace* w, const orbv_keyframe* a, const uint8_t* flags_a, const orbv_keyframe* b, const uint8_t* flags_b,
                      int mode, const orbv_triangulation* t, int th_low, float nnratio, int check_ori, int32_t* match, int* nmatches) {
    MORB_ARG(w != nullptr && a != nullptr && b != nullptr && nmatches != nullptr && mode >= 0 && mode <= 2);
    MORB_ARG(a->device == w->device && b->device == w->device);
    const int n_out = mode == 0 ? b->D.n : a->D.n;
    MORB_ARG(n_out == 0 || match != nullptr);
    TriDev T; memset(&T, 0, sizeof(T));
    int rc;
    if (mode == 2) {
        MORB_ARG(a->tri && b->tri);
        if ((rc = tri_params(t, T))) return rc;
        MORB_ARG(a->max_cam < t->n_cams && b->max_octave < t->n_levels);
    }
    *nmatches = 0;
    for (int i = 0; i < n_out; ++i) match[i] = -1;
    if (a->D.n_nodes == 0 || b->D.n_nodes == 0 || a->D.n == 0 || b->D.n == 0) return ORB_OK;
    MORB_HIP(hipSetDevice(w->device));
    SideDev A = a->D, B = b->D;
    if (flags_a || flags_b) {       // per-call MapPoint state: two small arrays through the pinned stage
        const size_t na = flags_a ? up16((size_t)A.n) : 0, nb = flags_b ? up16((size_t)B.n) : 0;
        if ((rc = w->h_stage.reserve(na + nb)) || (rc = w->d_stage.reserve(na + nb))) return rc;
        if (flags_a) { memcpy(w->h_stage.p, flags_a, (size_t)A.n); A.flags = w->d_stage.p; }
        if (flags_b) { memcpy(w->h_stage.p + na, flags_b, (size_t)B.n); B.flags = w->d_stage.p + na; }
        MORB_HIP(hipMemcpyAsync(w->d_stage.p, w->h_stage.p, na + nb, hipMemcpyHostToDevice, w->stream));
    }
    return launch_join(w, A, B, b->max_node, mode, T, th_low, nnratio, check_ori, match, nmatches);
}

}  // namespace

extern "C" {

int orbv_search_by_bow(orbv_workspace* w, const orbv_side* a, const orbv_side* b, int mode, int th_low, float nnratio,
                       int check_orientation, int32_t* match, int* nmatches) {
    MORB_ARG(mode == 0 || mode == 1);
    return run_join(w, a, b, mode, nullptr, th_low, nnratio, check_orientation, match, nmatches);
}

int orbv_search_for_triangulation(orbv_workspace* w, const orbv_side* a, const orbv_side* b, const orbv_triangulation* t,
                                  int th_low, int check_orientation, int32_t* match, int* nmatches) {
    return run_join(w, a, b, 2, t, th_low, 0.f, check_orientation, match, nmatches);
}

int orbv_keyframe_create(orbv_workspace* w, const orbv_side* s, orbv_keyframe** out) {
    MORB_ARG(w != nullptr && out != nullptr && s != nullptr);
    const bool tri = s->x != nullptr;
    int rc = side_check(s, tri);
    if (rc) return rc;
    MORB_HIP(hipSetDevice(w->device));
    Packer size_pass;
    (void)pack_side(size_pass, s, tri);
    std::vector<uint8_t> host(std::max<size_t>(size_pass.off, 16));
    orbv_keyframe* k = new orbv_keyframe();
    k->device = w->device; k->tri = tri;
    if ((rc = k->block.reserve(host.size()))) { delete k; return rc; }
    Packer P; P.h = host.data(); P.d = k->block.p;
    k->D = pack_side(P, s, tri);
    k->max_node = max_node_of(s);
    if (tri) for (int i = 0; i < s->n; ++i) {
        if (s->cam_of[i] < 0 || s->octave[i] < 0) { morb::set_error("negative camera / octave"); orbv_keyframe_destroy(k); return ORB_E_ARG; }
        k->max_cam = std::max(k->max_cam, (int)s->cam_of[i]); k->max_octave = std::max(k->max_octave, (int)s->octave[i]);
    }
    if (hipMemcpyAsync(k->block.p, host.data(), P.off, hipMemcpyHostToDevice, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) { morb::set_error("keyframe upload failed"); orbv_keyframe_destroy(k); return ORB_E_HIP; }
    *out = k;
    return ORB_OK;
}

void orbv_keyframe_destroy(orbv_keyframe* k) {
    if (!k) return;
    (void)hipSetDevice(k->device);
    k->block.release();
    delete k;
}

int orbv_keyframe_count(const orbv_keyframe* k) { return k ? k->D.n : 0; }

int orbv_keyframe_from_device(orbv_workspace* w, const orbv_vocabulary* v, const orbv_device_side* s, int levelsup, void* after_stream,
                              orbv_keyframe** out) {
    MORB_ARG(w != nullptr && v != nullptr && s != nullptr && out != nullptr && s->n >= 0 && w->device == v->device);
    MORB_ARG(s->n == 0 || (s->d_desc && s->d_angle && ((uintptr_t)s->d_desc & 15) == 0));
    const bool tri = s->d_x != nullptr;
    if (tri) MORB_ARG(s->d_y && s->d_octave && s->n_cams >= 1 && s->n_cams <= ORBV_MAX_CAMS && s->cam_start[0] == 0 && s->cam_start[s->n_cams] == s->n);
    const int nid_level = v->L - levelsup;
    const int depths = (int)v->lvl_start.size() - 1;
    const int lvl_nodes = (nid_level >= 1 && nid_level < depths) ? v->lvl_start[nid_level + 1] - v->lvl_start[nid_level] : 0;
    const int nbins = 1 + lvl_nodes;
    if (nbins > 4097) { morb::set_error("%d vocabulary nodes at level %d: build this FeatureVector with orbv_bow_vectors", lvl_nodes, nid_level); return ORB_E_CAPACITY; }
    MORB_HIP(hipSetDevice(w->device));
    const int n = s->n;
    orbv_keyframe* k = new orbv_keyframe();
    k->device = w->device; k->tri = tri;
    // block: desc | angle | flags | node_id | node_start | items | word | node | bin | [x | y | octave | cam_of] | counts | bin_to_node | meta
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = up16(off + bytes); return o; };
    const size_t nn = (size_t)std::max(n, 1);
    const size_t o_desc = take(nn * 32), o_ang = take(nn * 4), o_fl = take(nn), o_nid = take((size_t)nbins * 4), o_ns = take((size_t)(nbins + 1) * 4),
                 o_it = take(nn * 4), o_w = take(nn * 4), o_nd = take(nn * 4), o_bin = take(nn * 4);
    const size_t o_x = tri ? take(nn * 4) : 0, o_y = tri ? take(nn * 4) : 0, o_oc = tri ? take(nn * 4) : 0;
    const size_t o_cam = take(nn * 4), o_cnt = take((size_t)nbins * 4), o_b2n = take((size_t)nbins * 4), o_meta = take(16);
    int rc = k->block.reserve(off);
    if (rc) { delete k; return rc; }
    uint8_t* B = k->block.p;
    hipStream_t st = w->stream;
    auto fail = [&](const char* what) { morb::set_error("%s failed: %s", what, hipGetErrorString(hipGetLastError())); orbv_keyframe_destroy(k); return ORB_E_HIP; };
    if (after_stream) {   // the arrays are produced on another stream (the front end's): order ours behind it on the device
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");
        (void)hipEventRecord(ev, (hipStream_t)after_stream); (void)hipStreamWaitEvent(st, ev, 0); (void)hipEventDestroy(ev);
    }
    if (n > 0) {
        if (hipMemcpyAsync(B + o_desc, s->d_desc, (size_t)n * 32, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(B + o_ang, s->d_angle, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("keyframe copy");
        if (tri && (hipMemcpyAsync(B + o_x, s->d_x, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(B + o_y, s->d_y, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(B + o_oc, s->d_octave, (size_t)n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)) return fail("keyframe copy");
    }
    if (hipMemsetAsync(B + o_cnt, 0, (size_t)nbins * 4, st) != hipSuccess) return fail("hipMemsetAsync");
    CamStarts C; memset(&C, 0, sizeof(C));
    C.n_cams = tri ? s->n_cams : 1; if (tri) memcpy(C.start, s->cam_start, sizeof(int) * (s->n_cams + 1)); else C.start[1] = n;
    if (n > 0) {
        k_bow_transform<<<(n + 15) / 16, 256, 0, st>>>(v->d_desc.p, v->d_first_child.p, v->d_orig.p, v->d_word.p, (const uint4*)(B + o_desc), n, nid_level,
                                                       (uint32_t*)(B + o_w), (uint32_t*)(B + o_nd), nullptr, v->d_rank.p,
                                                       v->d_stop.p, (int*)(B + o_bin));
        k_fv_count<<<(n + 255) / 256, 256, 0, st>>>((const int*)(B + o_bin), n, (int*)(B + o_cnt));
        k_side_misc<<<(n + 255) / 256, 256, 0, st>>>(n, C, s->d_uright, (int32_t*)(B + o_cam), B + o_fl);
    }
    k_fv_offsets<<<1, 1024, 0, st>>>((const int*)(B + o_cnt), nbins, v->d_orig_sorted.p + (lvl_nodes ? v->lvl_start[nid_level] : 0), (uint32_t*)(B + o_nid),
                                     (int*)(B + o_ns), (int*)(B + o_b2n), (int*)(B + o_meta));
    if (n > 0) k_fv_fill<<<nbins, 64, 0, st>>>((const int*)(B + o_bin), n, (const int*)(B + o_b2n), (const int*)(B + o_ns), (uint32_t*)(B + o_it));
    int meta[4] = {0, 0, 0, 0};
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(meta, B + o_meta, 12, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return fail("keyframe build");
    SideDev& D = k->D;
    D.n = n; D.n_nodes = meta[0];
    D.desc = (const uint4*)(B + o_desc); D.angle = (const float*)(B + o_ang); D.flags = B + o_fl;
    D.node_id = (const uint32_t*)(B + o_nid); D.node_start = (const int32_t*)(B + o_ns); D.items = (const uint32_t*)(B + o_it);
    D.x = tri ? (const float*)(B + o_x) : nullptr; D.y = tri ? (const float*)(B + o_y) : nullptr;
    D.octave = tri ? (const int32_t*)(B + o_oc) : nullptr; D.cam_of = tri ? (const int32_t*)(B + o_cam) : nullptr;
    k->max_node = std::max(1, meta[1]);
    k->max_cam = tri ? s->n_cams - 1 : 0; k->max_octave = 0;   // octaves stay on the device: the caller vouches for octave < n_levels
    k->d_word = (const uint32_t*)(B + o_w); k->d_node = (const uint32_t*)(B + o_nd);
    *out = k;
    return ORB_OK;
}

int orbv_keyframe_download(orbv_workspace* w, const orbv_keyframe* k, uint32_t* word_id, uint32_t* node_of_feature, uint32_t* fv_node,
                           int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes) {
    MORB_ARG(w != nullptr && k != nullptr && w->device == k->device);
    MORB_HIP(hipSetDevice(w->device));
    const int n = k->D.n, nn = k->D.n_nodes;
    if (word_id || node_of_feature) MORB_ARG(k->d_word != nullptr);   // host-uploaded keyframes carry no descent results
    if (word_id && n) MORB_HIP(hipMemcpyAsync(word_id, k->d_word, (size_t)n * 4, hipMemcpyDeviceToHost, w->stream));
    if (node_of_feature && n) MORB_HIP(hipMemcpyAsync(node_of_feature, k->d_node, (size_t)n * 4, hipMemcpyDeviceToHost, w->stream));
    if (fv_node && nn) MORB_HIP(hipMemcpyAsync(fv_node, k->D.node_id, (size_t)nn * 4, hipMemcpyDeviceToHost, w->stream));
    if (fv_start) MORB_HIP(hipMemcpyAsync(fv_start, k->D.node_start, (size_t)(nn + 1) * 4, hipMemcpyDeviceToHost, w->stream));
    MORB_HIP(hipStreamSynchronize(w->stream));
    if (fv_items && fv_start && fv_start[nn] > 0) {
        MORB_HIP(hipMemcpyAsync(fv_items, k->D.items, (size_t)fv_start[nn] * 4, hipMemcpyDeviceToHost, w->stream));
        MORB_HIP(hipStreamSynchronize(w->stream));
    }
    if (n_fv_nodes) *n_fv_nodes = nn;
    return ORB_OK;
}

int orbv_search_by_bow_resident(orbv_workspace* w, const orbv_keyframe* a, const uint8_t* flags_a, const orbv_keyframe* b,
                                const uint8_t* flags_b, int mode, int th_low, float nnratio, int check_orientation, int32_t* match,
                                int* nmatches) {
    MORB_ARG(mode == 0 || mode == 1);
    return run_join_resident(w, a, flags_a, b, flags_b, mode, nullptr, th_low, nnratio, check_orientation, match, nmatches);
}

int orbv_search_for_triangulation_resident(orbv_workspace* w, const orbv_keyframe* a, const uint8_t* flags_a, const orbv_keyframe* b,
                                           const uint8_t* flags_b, const orbv_triangulation* t, int th_low, int check_orientation,
                                           int32_t* match, int* nmatches) {
    return run_join_resident(w, a, flags_a, b, flags_b, 2, t, th_low, 0.f, check_orientation, match, nmatches);
}

}  // extern "C"
