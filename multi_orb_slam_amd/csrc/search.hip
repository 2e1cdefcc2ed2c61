// search.hip -- the projection-gated searches of the ORB matcher (include/orbm.h).
//   k_project        M3  one wave per query walks the 64x48 grid cells of its window(s) in the reference's visiting order,
//                    ballot-compacts the survivors in order, gathers their descriptors and keeps a sorted shortlist.
//                    reference src/ORBmatcher.cc:3547-3592 + src/Frame.cc:574-629.
//   k_resolve / k_rs_*   the order-dependent part of SearchByProjection (first-come claims, rotation histogram,
//                    ComputeThreeMaxima) as a fixed-point iteration ON THE DEVICE; host_resolve replays the loop on the host
//                    as the exact fallback (sweep limit, MORB_HOST_RESOLVE=1).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/orbm.h"
#include "../../include/orb_debug.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "matcher_internal.h"
#include "hamming_dev.h"
#include "project_dev.h"

using namespace morb;

namespace {
__global__ __launch_bounds__(256) void k_project(ProjectArgs A) {
    MORB_LATENCY_KERNEL_WIDE();
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (qi >= A.nq) return;
    project_wave(A, qi, threadIdx.x & 63);
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
MORB_PHASE_DECL(g_ph_chg);   // (instrumented build) queries whose choice changed, per sweep, accumulated over launches

// RESOLVE_K (above): sorted shortlist per query built by k_project; a full rescan happens only when all of it is taken

// Evaluates query i against the current claim table.  `avail(g)` decides visibility.  Candidates are visited in the
// order given; FRAMES: first minimum.  POINTS: best + second (with multiplicity) and their levels.
// LDSQ: the per-query sweep state (shortlist features + distances, blocks flag, current choice) also lives in LDS, so a
// sweep touches no global memory at all; used whenever it fits next to the claim table.
template <bool POINTS, bool LDSQ>
__global__ __launch_bounds__(1024) void k_resolve(FrameDev F, const int2* __restrict__ qmeta /* {blocks | camera << 1, angle bits} */, int nq, int cap,
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
            l_fl[i] = (unsigned char)((qm.x & 1) | (topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K ? 2 : 0));
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
    mx = (int)(0x7fffffffu - wave_min_u32(0x7fffffffu - (unsigned)mx));   // wave maximum on the DPP path (counts are >= 0)
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
    // The shortlists never change: the entries of a thread's first two queries (four when the sweep state does not fit LDS and
    // would be read from HBM in every sweep) stay in registers over the sweeps; so do their flags and their current choices.
    // (Four for the LDS-resident form as well costs more than it saves: twice the code for batches that do not exist.)
    constexpr int RQ = LDSQ ? 2 : 4;
    int gdr[RQ][RESOLVE_K], flr[RQ], chr[RQ];
    if constexpr (LDSQ) {
        if (!POINTS) {
            // Sweep 0 of the frame search needs no evaluation: without any claim a query takes the head of its shortlist (every
            // entry there is acceptable: not occupied, distance <= th_high).  The choices and their claims -- what sweep 0
            // would have left for sweep 1 to read: table 2, tag 0x7ffd -- are written straight away.
            const int tag_next = 0x7ffd << 16;
            for (int i = tid; i < nq; i += T) {
                const int v = l_gd[i];   // entry 0
                const int g0 = (v & 0xffff) == 0xffff ? -1 : (v & 0xffff);
                const int nc = (g0 >= 0 && (int)((unsigned)v >> 16) <= th_high) ? g0 : -1;
                l_choice[i] = nc;
                if (nc >= 0 && (l_fl[i] & 1)) atomicMin(&s_claim2[nc], tag_next | i);
            }
            __syncthreads();
            MORB_PHASE(g_ph_res, 20); MORB_PHASE(g_ph_res, 3);
            it = 1;
        }
#pragma unroll
        for (int b = 0; b < RQ; ++b) {
            const bool in = b * T + tid < nq;
#pragma unroll
            for (int k = 0; k < RESOLVE_K; ++k) gdr[b][k] = in ? l_gd[k * nq + b * T + tid] : 0xffff;
            flr[b] = in ? (int)l_fl[b * T + tid] : 0;
            chr[b] = in ? l_choice[b * T + tid] : -1;
        }
    } else {
#pragma unroll
        for (int b = 0; b < RQ; ++b) {
            const int i = b * T + tid;
            const bool in = i < nq;
#pragma unroll
            for (int k = 0; k < RESOLVE_K; ++k) gdr[b][k] = in ? ((tk_key[k * nq + i] & 0xffff0000) | (tk_g[k * nq + i] & 0xffff)) : 0xffff;
            flr[b] = in ? ((qmeta[i].x & 1) | (topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K ? 2 : 0)) : 0;
            chr[b] = -1;
        }
        if (!POINTS && nq <= RQ * T) {   // sweep 0 as above, from the registers
            const int tag_next = 0x7ffd << 16;
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                const int i = b * T + tid;
                if (i < nq) {
                    const int v = gdr[b][0];
                    const int g0 = (v & 0xffff) == 0xffff ? -1 : (v & 0xffff);
                    const int nc = (g0 >= 0 && (int)((unsigned)v >> 16) <= th_high) ? g0 : -1;
                    chr[b] = nc;
                    choice[i] = nc;
                    if (nc >= 0 && (flr[b] & 1)) atomicMin(&s_claim2[nc], tag_next | i);
                }
            }
            __syncthreads();
            it = 1;
        }
    }
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
        // one batch of 1024 queries; B < RQ: the batch whose shortlists sit in gdr[B] (uniform trip count over the batches:
        // the cooperative rescans below need whole waves)
        auto batch = [&](const int base, auto BC) {
            constexpr int B = decltype(BC)::value;
            const int i = base + tid;
            const bool valid = i < nq;
            int gk[RESOLVE_K], dk[RESOLVE_K], ck[RESOLVE_K];
            int fl = 0, old = -1, nc = -1;
            bool need_rescan = false;
            if constexpr (!POINTS && B < RQ) {
                // Frame search on register-resident shortlists, without a divergent branch: the first entry that is there and
                // not hidden by a lower blocking query's claim wins.  Entries 0-1 first; 2..K-1 only for waves in which a lane
                // is still looking.  (A lane past the last query holds empty entries and comes out with nc == old == -1.)
                fl = flr[B]; old = chr[B];
                int e[RESOLVE_K];
                bool none[RESOLVE_K], hid[RESOLVE_K];
                auto look = [&](int k) {
                    e[k] = gdr[B][k];
                    none[k] = (e[k] & 0xffff) == 0xffff;
                    const int c = rd[none[k] ? 0 : (e[k] & 0xffff)];
                    hid[k] = !none[k] && (unsigned)(c - tag) < (unsigned)i;   // claimed by a lower blocking query
                };
                look(0); look(1);
                const bool av0 = !none[0] && !hid[0], av1 = !none[1] && !hid[1];
                int pick = av0 ? e[0] : (av1 ? e[1] : -1);
                bool anyh = hid[0] || hid[1];
                const bool open = !av0 && !av1 && !none[1];   // (an empty slot ends the list; slot 0 empty implies slot 1 empty)
                if (__ballot(open)) {
#pragma unroll
                    for (int k = K0; k < RESOLVE_K; ++k) look(k);
                    int p2 = -1;
#pragma unroll
                    for (int k = RESOLVE_K - 1; k >= K0; --k) { p2 = (!none[k] && !hid[k]) ? e[k] : p2; anyh = anyh || (open && hid[k]); }
                    pick = open ? p2 : pick;
                }
                // the shortlist is exact unless it ran dry while longer lists exist (rare): rescanned right below
                need_rescan = pick < 0 && (fl & 2) && anyh;
                if (pick >= 0 && (pick >> 16) <= th_high) nc = pick & 0xffff;
            } else
            if (valid) {
                if constexpr (B < RQ) { fl = flr[B]; old = chr[B]; }
                else {
                    fl = LDSQ ? (int)l_fl[i] : ((qmeta[i].x & 1) | (topk[(2 * RESOLVE_K) * nq + i] > RESOLVE_K ? 2 : 0));
                    old = LDSQ ? l_choice[i] : choice[i];
                }
                int best = 256, best2 = 256, lvl = -1, lvl2 = -1, bidx = -1, g2 = -1;
                int found = 0, taken = 0;
                bool walking = true;
                auto fetch = [&](int k) {
                    if constexpr (B < RQ) {
                        const int v = gdr[B][k];
                        gk[k] = (v & 0xffff) == 0xffff ? -1 : (v & 0xffff);
                        dk[k] = (int)((unsigned)v >> 16);
                    } else if (LDSQ) {
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
#ifdef MORB_PHASE_CLOCKS
                if (nc != old && it < 15) atomicAdd((unsigned long long*)&g_ph_chg[it], 1ull);
#endif
                if (nc != old) { ch = 1; if (LDSQ) l_choice[i] = nc; else choice[i] = nc; }
                if constexpr (B < RQ) chr[B] = nc;
                if (nc >= 0 && (fl & 1)) atomicMin(&wr[nc], tag_next | i);  // what the next sweep sees
            }
        };
        batch(0, std::integral_constant<int, 0>{});
        if (T < nq) batch(T, std::integral_constant<int, 1>{});
        if constexpr (RQ > 2) {
            if (2 * T < nq) batch(2 * T, std::integral_constant<int, 2>{});
            if (3 * T < nq) batch(3 * T, std::integral_constant<int, 3>{});
        }
        for (int base = RQ * T; base < nq; base += T) batch(base, std::integral_constant<int, RQ>{});
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
    acc = __builtin_amdgcn_readlane(wave_incl_scan(acc), 63);   // wave sum on the DPP path
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
        rej = __builtin_amdgcn_readlane(wave_incl_scan(rej), 63);
        if (lane == 0) atomicSub(&s_red, rej);
        __syncthreads();
    }
    MORB_PHASE(g_ph_res, 60);
    for (int g = tid; g < NT; g += T) match_of_feature[g] = tagb ? (tagb | (s_claim[g] + 2)) : s_claim[g];
    // A tagged launch rewrites EVERY word of the frame's capacity, the ones past this frame's count as "no match": a word can
    // then only carry the current sequence number if this launch stored it (the numbers cycle after 2047 launches; a word
    // left alone since its last use -- the count dropped, stayed low for a multiple of 2047 launches and rose again -- would
    // otherwise show the right tag with an old value before this launch's store has crossed PCIe).
    if (tagb) for (int g = NT + tid; g < F.n_total; g += T) match_of_feature[g] = tagb | 1;
    if (tid == 0) { status[1] = tagb | s_red; status[2] = tagb | it; status[3] = tagb | maxcount; status[0] = tagb | 0; }
    MORB_PHASE(g_ph_res, 61);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0) g_ph_res[62] = (unsigned long long)it;
#endif
}

// ---- the frame search's resolve as a MONOTONE iteration (round 3; the default for SearchByProjection(Frame, Frame) whenever the
// sweep state fits LDS).  The Jacobi form above rebuilds every claim in every sweep (two tables, sweep tags) and re-evaluates
// every query: 8-9 sweeps of ~3 us on the benchmark stream.  It does not have to.  Let every query keep a cursor into its ordered
// candidate list and ONE claim table hold, per feature, the lowest index of a blocking query that ever claimed it (atomicMin, never
// reset).  A query moves its cursor on exactly when the entry under it is claimed by a LOWER blocking query, and claims the
// first entry that is not.  Then:
//   * a claim only ever moves to lower indices, so what is hidden from a query stays hidden: cursors only move forward;
//   * the lowest claimant of a feature never has a reason to let go of it (it moves on only if somebody LOWER claims the
//     feature), so at rest every table entry names a query that really picks that feature -- no stale claim hides anything;
//   * at rest every query sits on its first candidate that no lower blocking query picks, which is the fixed point of the
//     reference's sequential loop (src/ORBmatcher.cc:3502-3614), and that fixed point is unique (induction over the query order).
// So a round is: look at the claim on the current pick (one LDS read and a compare for the ~2/3 of the queries it does not
// concern); only a displaced query walks on.  Rounds are counted as before (a claim still travels one link of a dependency
// chain per round) but cost a tenth of a sweep.  Owners, rotation histogram and result words as in k_resolve.  POINTS (top-2 with
// ratio test), two-window queries and states beyond LDS keep k_resolve.
template <int RQ, bool ANG>
__global__ __launch_bounds__(1024) void k_resolve_mono(FrameDev F, const int2* __restrict__ qmeta, int nq, int cap,
                                                       const int* __restrict__ cand_idx, const uint16_t* __restrict__ cand_dist,
                                                       const int* __restrict__ cand_count, const uint8_t* __restrict__ occupied,
                                                       const float* __restrict__ f_angle, int th_high, int check_ori, int max_it,
                                                       const int* __restrict__ topk, int* __restrict__ match_of_feature,
                                                       int* __restrict__ status, int tagb, MergeJob MJ, int worklist) {
    MORB_LATENCY_KERNEL();
    // Workgroups behind the first one (isolated steps only) merge the slice partials of the camera-pair top-2 that rode in the projection's
    // launch: the resolve does not need them, the step does -- one kernel and one kernel boundary less between projection and resolve.
    if (blockIdx.x > 0) {
        const int mq = MJ.d_range ? MJ.d_range[2] : MJ.nq;
        const int qi = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
        if (qi < mq) top2_merge_query(MJ.p_idx, MJ.p_best, MJ.p_second, MJ.S, mq, qi, MJ.o_idx, MJ.o_best, MJ.o_second);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // (system scope: the results live in mapped host memory)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(MJ.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) int s_claim[];  // [0, n): lowest blocking claimant of a feature; [n, 2n): owner (last claimant)
    __shared__ int s_hist[ORBM_HISTO_LENGTH];
    // Every exit of this workgroup publishes tagged result words the host takes as "the launch is over": none of them may be written
    // before the merging workgroups of the same launch have released their results.  Called by all threads; false when
    // the wait gave up -- the status word then says 3 and the host synchronises the stream before it reads anything.
    __shared__ int s_merge_ok;
    auto merge_wait = [&]() -> bool {
        if (!(MJ.S > 1 && MJ.done)) return true;
        if (threadIdx.x == 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(MJ.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - MJ.target) < 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
            s_merge_ok = spins < (1 << 22) ? 1 : 0;
        }
        __syncthreads();
        return s_merge_ok != 0;
    };
    __shared__ int s_keep[3];
    __shared__ int s_red;
    __shared__ int s_flag[3];   // "some wave changed something in round it": [it % 3]
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    MORB_PHASE(g_ph_res, 0);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0) { g_ph_res[1] = 0; g_ph_res[58] = clock64(); }   // (slot 1 zero = this kernel's layout; shader clock next to the 100 MHz one)
#endif
    if (tid < 3) s_flag[tid] = 0;
    const int NT = F.n_total_dev ? *F.n_total_dev : F.n_total;
    const int* tk_g = topk + RESOLVE_K * nq;
    // LDS (mono_lds_bytes on the host): claims, owners (ints); per query: list length, pick, K shortlist features (u16: the frames
    // this kernel takes have < 65535 features; 0xffff = none), flags (u8); with ANG the query / feature angles for the rotation
    // histogram (else they are read from HBM in the tail: what lets 4 x 1000 or 2 x 2000 features in at all)
    int* s_owner = s_claim + F.n_total;
    const int nq2 = (nq + 1) & ~1;
    unsigned short* l_cnt = reinterpret_cast<unsigned short*>(s_owner + F.n_total);
    unsigned short* l_choice = l_cnt + nq2;
    unsigned short* l_gd = l_choice + nq2;
    unsigned char* l_fl = reinterpret_cast<unsigned char*>(l_gd + RESOLVE_K * nq2);  // bit0 blocks, bit1 list > K, bits 2.. rotation bin + 1
    float* l_ang = reinterpret_cast<float*>(l_fl + ((nq + 3) & ~3));
    float* l_fang = l_ang + nq;
    unsigned short* l_wl = reinterpret_cast<unsigned short*>(ANG ? l_fang + F.n_total : l_ang);   // worklist of displaced queries (nq2 entries)
    __shared__ int s_wl_n[2];
    if (tid == 0) s_red = 0;
    if (tid < ORBM_HISTO_LENGTH) s_hist[tid] = 0;
    for (int g = tid; g < F.n_total; g += T) {
        s_claim[g] = 0x7fffffff; s_owner[g] = -1;
        if (ANG && check_ori) l_fang[g] = f_angle[g];
    }
    __syncthreads();   // (the claims of round 0 go into the table right below)
    // per query of this thread (two in registers; more only beyond 2048 queries, re-read from LDS every round): the shortlist
    // (feature indices, 0xffff = none: every entry k_project put there is acceptable -- not occupied, distance <= th_high --, so the
    // distances stay behind), the cursor, the pick under it (a rescanned pick is not on the shortlist)
    constexpr int K = RESOLVE_K;
    int sl[RQ][K], cur[RQ], pos[RQ], flr[RQ];
    int mx = 0;
    // set-up of one query: every load is independent of every other -- one trip to HBM for the whole pass.  Round 0 rides along:
    // without any claim a query takes the head of its shortlist (every entry there is acceptable: not occupied, distance <=
    // th_high) and, if it blocks, claims it
    auto setup = [&](const int i, int (&e)[K], int& c, int& fl) {
        const int cnt_i = cand_count[i];
        mx = max(mx, cnt_i);
        l_cnt[i] = (unsigned short)min(cnt_i, 65535);
        const int2 qm = qmeta[i];
        fl = (qm.x & 1) | (topk[(2 * K) * nq + i] > K ? 2 : 0);
        l_fl[i] = (unsigned char)fl;
        if (ANG) l_ang[i] = __int_as_float(qm.y);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            e[k] = tk_g[k * nq + i] & 0xffff;
            l_gd[k * nq2 + i] = (unsigned short)e[k];
        }
        const bool has = e[0] != 0xffff;
        c = has ? e[0] : -1;
        l_choice[i] = (unsigned short)(has ? e[0] : 0xffff);
        if (has && (fl & 1)) atomicMin(&s_claim[e[0]], i);
    };
#pragma unroll
    for (int b = 0; b < RQ; ++b) {
        const int i = b * T + tid;
        cur[b] = -1; pos[b] = 0; flr[b] = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) sl[b][k] = 0xffff;
        if (i < nq) setup(i, sl[b], cur[b], flr[b]);
    }
    for (int i = RQ * T + tid; i < nq; i += T) { int e[K], c, fl; setup(i, e, c, fl); }
    mx = (int)(0x7fffffffu - wave_min_u32(0x7fffffffu - (unsigned)mx));
    if (lane == 0) atomicMax(&s_red, mx);
    __syncthreads();
    const int maxcount = s_red;
    if (maxcount > cap) {
        (void)merge_wait();
        if (tid == 0) { status[1] = tagb | 0; status[2] = tagb | 0; status[3] = tagb | maxcount; status[0] = tagb | 2; }
        return;
    }
    MORB_PHASE(g_ph_res, 2);
    // one pass over a pair of queries: is the current pick still free of lower blocking claims (one LDS read each, issued
    // together)?  For a displaced query the claims on ALL later shortlist entries are fetched in one batch, the first free one is
    // taken and claimed; a dry shortlist that is not the whole list is rescanned by the whole wave (rare).  Returns "somebody in
    // this wave was displaced".
#ifdef MORB_PHASE_CLOCKS
    int it_dbg = 0;
#endif
    auto pass = [&](const int (&qi)[RQ], const int (&e)[RQ][K], int (&c)[RQ], int (&p)[RQ], const int (&fl)[RQ]) -> bool {
        bool disp[RQ], need_rescan[RQ];
        int cl[RQ];
#pragma unroll
        for (int b = 0; b < RQ; ++b) cl[b] = s_claim[c[b] >= 0 ? (c[b] & 0xffff) : 0];   // (unconditional reads: issued together, no branch)
#pragma unroll
        for (int b = 0; b < RQ; ++b) { disp[b] = c[b] >= 0 && cl[b] < qi[b]; need_rescan[b] = false; }  // (only blocking queries write claims; an own claim equals the index)
        bool any = false;
#pragma unroll
        for (int b = 0; b < RQ; ++b) any |= disp[b];
        const unsigned long long wany = __ballot(any);
        if (!wany) return false;
        // (round 6: a register slot b none of whose 64 queries is displaced is skipped by the WAVE -- after the first pass or two a handful
        //  of lanes are still moving, on one slot, and the wave whose queries depend on everybody else's walks eight passes while the other
        //  fifteen wait at the round's barrier: its pass went from ~2 us to ~0.6, profiles/r06/notes_experiments.md)
        {
            int ck[RQ][K];
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                if (!__ballot(disp[b])) continue;
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const bool want = disp[b] && k > p[b] && (e[b][k] & 0xffff) != 0xffff;
                    const int v = s_claim[want ? (e[b][k] & 0xffff) : 0];   // (the reads of a slot are in flight together.  Measured: asking
                    ck[b][k] = want ? v : -1;                               // for the next entry alone first costs a third trip more often than it saves reads)
                }
            }
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                if (!__ballot(disp[b])) continue;
                if (!disp[b]) continue;
                const int i = qi[b];
#ifdef MORB_PHASE_CLOCKS
                if (it_dbg < 15) atomicAdd((unsigned long long*)&g_ph_chg[it_dbg], 1ull);
#endif
                int nk = K;
#pragma unroll
                for (int k = K - 1; k >= 1; --k) if (ck[b][k] >= i) nk = k;
                int ne = 0xffff;
#pragma unroll
                for (int k = 1; k < K; ++k) if (nk == k) ne = e[b][k];
                if (nk < K) {
                    c[b] = ne; p[b] = nk; l_choice[i] = (unsigned short)ne;
                    if (fl[b] & 1) atomicMin(&s_claim[ne], i);
                } else {
                    c[b] = -1; l_choice[i] = 0xffff;
                    // the shortlist is exact unless it ran dry while longer lists exist: rescanned right below
                    need_rescan[b] = (fl[b] & 2) != 0;
                    p[b] = K;
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RQ; ++b) {
            unsigned long long todo = __ballot(need_rescan[b]);
#ifdef MORB_PHASE_CLOCKS
            if (todo && lane == 0 && it_dbg < 15) atomicAdd((unsigned long long*)&g_ph_chg[16 + it_dbg], (unsigned long long)__popcll(todo));
#endif
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int q = __builtin_amdgcn_readlane(qi[b], src);
                const int full = l_cnt[q];
                int k1 = 0x7fffffff, g1 = -1;
                for (int k0 = 0; k0 < full; k0 += 64) {
                    const int k = k0 + lane;
                    int key = 0x7fffffff, gg = -1;
                    if (k < full) {
                        gg = cand_idx[k * nq + q];
                        const int d = cand_dist[k * nq + q];
                        bool avail = !(occupied && occupied[gg]);
                        if (s_claim[gg] < q) avail = false;
                        if (avail) key = (d << 16) | k;
                    }
                    const int m1 = (int)wave_min_u32((unsigned)key);
                    if (m1 < k1) { k1 = m1; g1 = __builtin_amdgcn_readlane(gg, __ffsll((long long)__ballot(key == m1)) - 1); }
                }
                if (lane == src && k1 != 0x7fffffff && (k1 >> 16) <= th_high) {   // the rescanned pick is the entry under the cursor from now on
                    c[b] = g1; l_choice[q] = (unsigned short)g1;
                    if (fl[b] & 1) atomicMin(&s_claim[g1], q);
                }
            }
        }
        return true;
    };
    // A ROUND = every wave repeats its pass until none of ITS queries is displaced (no barrier in between: what the other waves
    // claim meanwhile is seen as it lands -- the iteration is monotone, so any interleaving ends in the same place), then one
    // barrier; a round in which no wave saw anything displaced is the fixed point.  Dependency chains are followed at the pace
    // of a pass (two LDS trips), not of a barrier: the benchmark stream needs 2-3 rounds where the Jacobi form needs 9 sweeps.
    int it = 1, changed = 1;
    if (worklist) {
        // ---- round 6: the rounds as a WORKLIST.  Above, a wave repeats its pass over its own queries until none of them is displaced,
        // and the wave whose queries depend on everybody else's walks up to eight ~1.7 us passes while fifteen waves wait at the round's
        // barrier.  Here a round is: (A) every thread looks at the claims on its queries' picks (two LDS reads each) and puts the
        // displaced ones on a list; (B) the list is walked by ALL threads, one displaced query per thread: it moves to the first later
        // entry of its shortlist that no lower blocking query claims and claims it.  Same monotone iteration, another schedule (any
        // interleaving ends in the same fixed point); a round with an empty list is the fixed point.  The state of a query is what
        // set-up left in LDS (shortlist, flags, pick); its cursor is the pick's place on the shortlist.
        if (tid < 2) s_wl_n[tid] = 0;
        __syncthreads();
        for (; it < max_it; ++it) {
            int* cnt = &s_wl_n[it & 1];
            for (int i0 = 0; i0 < nq; i0 += RQ * T) {   // (RQ queries per thread and trip: their picks, then the claims on them, in flight together)
                int cc[RQ], cl[RQ];
#pragma unroll
                for (int b = 0; b < RQ; ++b) { const int i = i0 + b * T + tid; cc[b] = i < nq ? (int)l_choice[i] : 0xffff; }
#pragma unroll
                for (int b = 0; b < RQ; ++b) cl[b] = s_claim[cc[b] != 0xffff ? cc[b] : 0];
#pragma unroll
                for (int b = 0; b < RQ; ++b) {
                    const int i = i0 + b * T + tid;
                    const bool disp = cc[b] != 0xffff && cl[b] < i;
                    const unsigned long long m = __ballot(disp);
                    if (m) {
                        int base = 0;
                        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(cnt, __popcll(m));
                        base = __builtin_amdgcn_readlane(base, __ffsll((long long)m) - 1);
                        if (disp) l_wl[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
                    }
                }
            }
            if (tid == 0) s_wl_n[(it + 1) & 1] = 0;
            __syncthreads();
            const int nwl = *cnt;
#ifdef MORB_PHASE_CLOCKS
            if (tid == 0 && it < 15) atomicAdd((unsigned long long*)&g_ph_chg[it], (unsigned long long)nwl);
#endif
            if (nwl == 0) { changed = 0; break; }
            for (int j0 = 0; j0 < nwl; j0 += T) {   // (whole waves: a rescan below is the wave's work)
                const int j = j0 + tid;
                const int i = j < nwl ? (int)l_wl[j] : -1;
                bool need_rescan = false;
                int fl = 0;
                if (i >= 0) {
                    fl = l_fl[i];
                    const int c = l_choice[i];
                    int e[K], p = K;
#pragma unroll
                    for (int k = 0; k < K; ++k) e[k] = l_gd[k * nq2 + i];
#pragma unroll
                    for (int k = K - 1; k >= 0; --k) if (e[k] == c) p = k;   // (a rescanned pick is not on the shortlist: cursor at the end)
                    int ck[K];
#pragma unroll
                    for (int k = 1; k < K; ++k) {
                        const bool want = k > p && e[k] != 0xffff;
                        const int v = s_claim[want ? e[k] : 0];
                        ck[k] = want ? v : -1;
                    }
                    int nk = K;
#pragma unroll
                    for (int k = K - 1; k >= 1; --k) if (ck[k] >= i) nk = k;
                    int ne = 0xffff;
#pragma unroll
                    for (int k = 1; k < K; ++k) if (nk == k) ne = e[k];
                    l_choice[i] = (unsigned short)ne;
                    if (nk < K) { if (fl & 1) atomicMin(&s_claim[ne], i); }
                    else need_rescan = (fl & 2) != 0;   // the shortlist ran dry while the list is longer: rescanned by the wave
                }
                unsigned long long todo = __ballot(need_rescan);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int q = __builtin_amdgcn_readlane(i, src);
                    const int full = l_cnt[q];
                    int k1 = 0x7fffffff, g1 = -1;
                    for (int k0 = 0; k0 < full; k0 += 64) {
                        const int k = k0 + lane;
                        int key = 0x7fffffff, gg = -1;
                        if (k < full) {
                            gg = cand_idx[k * nq + q];
                            const int d = cand_dist[k * nq + q];
                            bool avail = !(occupied && occupied[gg]);
                            if (s_claim[gg] < q) avail = false;
                            if (avail) key = (d << 16) | k;
                        }
                        const int m1 = (int)wave_min_u32((unsigned)key);
                        if (m1 < k1) { k1 = m1; g1 = __builtin_amdgcn_readlane(gg, __ffsll((long long)__ballot(key == m1)) - 1); }
                    }
                    if (lane == src && k1 != 0x7fffffff && (k1 >> 16) <= th_high) {
                        l_choice[q] = (unsigned short)g1;
                        if (fl & 1) atomicMin(&s_claim[g1], q);
                    }
                }
            }
            __syncthreads();
            MORB_PHASE(g_ph_res, min(2 + it, 50));
        }
    } else {
    int qi01[RQ];
#pragma unroll
    for (int b = 0; b < RQ; ++b) qi01[b] = b * T + tid;
    for (; it < max_it && changed; ++it) {
#ifdef MORB_PHASE_CLOCKS
        it_dbg = it;
#endif
        if (tid == 0) s_flag[(it + 1) % 3] = 0;   // (last read two rounds ago)
        bool ch = false;
        for (int guard = 0; guard < 4096; ++guard) {
            bool w = pass(qi01, sl, cur, pos, flr);
            // queries beyond the registers (more than 2048): shortlist and pick come back from LDS; the cursor restarts at the
            // pick's place on the shortlist (a rescanned pick has none: cursor at the end)
            for (int base = RQ * T; base < nq; base += RQ * T) {
                int qx[RQ], e[RQ][K], c[RQ], p[RQ], fl[RQ];
#pragma unroll
                for (int b = 0; b < RQ; ++b) {
                    const int i = base + b * T + tid;
                    qx[b] = i; c[b] = -1; p[b] = K; fl[b] = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) e[b][k] = 0xffff;
                    if (i < nq) {
                        fl[b] = l_fl[i];
                        const int g = l_choice[i] == 0xffff ? -1 : (int)l_choice[i];
#pragma unroll
                        for (int k = 0; k < K; ++k) e[b][k] = l_gd[k * nq2 + i];
                        c[b] = g;
#pragma unroll
                        for (int k = K - 1; k >= 0; --k) if (g >= 0 && e[b][k] == g) p[b] = k;
                    }
                }
                w |= pass(qx, e, c, p, fl);
            }
            if (!w) {
#ifdef MORB_PHASE_CLOCKS
                if (lane == 0 && it_dbg < 15) atomicMax((unsigned long long*)&g_ph_chg[32 + it_dbg], (unsigned long long)guard);
                if (lane == 0 && it_dbg == 1 && blockIdx.x == 0) g_ph_chg[48 + (tid >> 6)] = ((unsigned long long)guard << 32) | (unsigned long long)(wall_clock64() - g_ph_res[0]);   // when this wave left round 1, after how many passes
#endif
                break;
            }
            ch = true;
        }
        if (ch && lane == 0) s_flag[it % 3] = 1;
        __syncthreads();
        changed = s_flag[it % 3];
        MORB_PHASE(g_ph_res, min(2 + it, 50));
    }
    }
    if (changed) {  // ran out of rounds
        (void)merge_wait();
        if (tid == 0) { status[1] = tagb | 0; status[2] = tagb | it; status[3] = tagb | maxcount; status[0] = tagb | 1; }
        return;
    }
    if (tid == 0) s_red = 0;
    __syncthreads();
    MORB_PHASE(g_ph_res, 52);
    // owners: the last claimant in query order (claims after a blocking one are impossible, so max index == final owner)
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int acc = 0;
    for (int i = tid; i < nq; i += T) {
        const int c = l_choice[i] == 0xffff ? -1 : (int)l_choice[i];
        if (c < 0) continue;
        ++acc;
        atomicMax(&s_owner[c], i);
        if (check_ori) {
            float rot = ANG ? l_ang[i] - l_fang[c] : __int_as_float(qmeta[i].y) - f_angle[c];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            const bool inr = bin >= 0 && bin < ORBM_HISTO_LENGTH;
            l_fl[i] = (unsigned char)((l_fl[i] & 3) | ((inr ? bin + 1 : 0) << 2));
            unsigned long long todo = __ballot(inr);
            for (int r = 0; todo && r < 3; ++r) {
                const int b0 = __builtin_amdgcn_readlane(bin, __ffsll((long long)todo) - 1);
                const unsigned long long same = __ballot(inr && bin == b0);
                if (inr && bin == b0 && lane == __ffsll((long long)same) - 1) atomicAdd(&s_hist[b0], __popcll(same));
                todo &= ~same;
            }
            if (inr && ((todo >> lane) & 1)) atomicAdd(&s_hist[bin], 1);
        }
    }
    acc = __builtin_amdgcn_readlane(wave_incl_scan(acc), 63);
    if (lane == 0) atomicAdd(&s_red, acc);
    __syncthreads();
    MORB_PHASE(g_ph_res, 53);
    if (check_ori) {
        if (tid < 64) {   // ComputeThreeMaxima (reference src/ORBmatcher.cc:3948-3989), as in k_resolve
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
            const int c = l_choice[i] == 0xffff ? -1 : (int)l_choice[i];
            if (c < 0) continue;
            const int bin = (int)(l_fl[i] >> 2) - 1;  // -1: outside the histogram, never rejected
            if (bin >= 0 && bin < ORBM_HISTO_LENGTH && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                s_owner[c] = -2;  // every writer stores -2; owners were settled before the barrier
                ++rej;
            }
        }
        rej = __builtin_amdgcn_readlane(wave_incl_scan(rej), 63);
        if (lane == 0) atomicSub(&s_red, rej);
        __syncthreads();
    }
    MORB_PHASE(g_ph_res, 60);
    const bool merged = merge_wait();   // (the merging workgroups of this launch finished long ago; the result words below must not say so before they have)
    for (int g = tid; g < NT; g += T) match_of_feature[g] = tagb ? (tagb | (s_owner[g] + 2)) : s_owner[g];
    if (tagb) for (int g = NT + tid; g < F.n_total; g += T) match_of_feature[g] = tagb | 1;   // (see k_resolve: no stale tag can match)
    if (tid == 0) { status[1] = tagb | s_red; status[2] = tagb | it; status[3] = tagb | maxcount; status[0] = tagb | (merged ? 0 : 3); }
    MORB_PHASE(g_ph_res, 61);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0) { g_ph_res[62] = (unsigned long long)it; g_ph_res[59] = clock64(); }
#endif
}

// ---- the same resolve for frames whose claim tables do not fit LDS (beyond ~18 000 features: 8 cameras x 4000), spread
// over the whole chip.  The two claim tables, the choices and the owner table live in HBM (L2-resident); one launch per
// sweep (a grid-wide barrier is exactly what a kernel boundary is), a fixed number of sweeps is enqueued and a sweep
// that finds "nothing changed" in its predecessor's flag does nothing, so no host round trip sits between sweeps.
// state: [0] longest candidate list, [1] matches, [2..4] kept rotation bins, [5] sweeps enqueued, [6] rounds (per-camera form), [8..8+RS_MAX_SWEEPS) changed flags,
//        [48..78) rotation histogram.
constexpr int RS_MAX_SWEEPS = 24;
constexpr int RS_STATE_INTS = 80;

__global__ __launch_bounds__(256) void k_rs_init(int n_cap, int nq, int* __restrict__ tab0, int* __restrict__ tab1,
                                                 int* __restrict__ owner, int* __restrict__ choice,
                                                 const int* __restrict__ cand_count, int* __restrict__ state, int n_sweeps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) state[5] = n_sweeps;
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
    if (state[0] > cap || state[8 + state[5] - 1] != 0) return;
    // (the histogram and the match count are gathered per workgroup in LDS first: 32 000 queries put ~15 000 atomics on 31 words
    // of HBM otherwise, which the L2 serialises -- 37 us of the 8 x 4000 search)
    __shared__ int s_hist[ORBM_HISTO_LENGTH + 1];
    if (threadIdx.x <= ORBM_HISTO_LENGTH) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = i < nq ? choice[i] : -1;
    if (c >= 0) atomicMax(&owner[c], i);
    const unsigned long long any = __ballot(c >= 0);
    if (lane == 0 && any) atomicAdd(&s_hist[ORBM_HISTO_LENGTH], __popcll(any));
    if (check_ori) {
        int bin = -1;
        if (c >= 0) {
            float rot = q[i].angle - f_angle[c];
            if (rot < 0.0) rot += 360.0f;
            bin = (int)roundf(rot * (1.0f / ORBM_HISTO_LENGTH));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            if (bin < 0 || bin >= ORBM_HISTO_LENGTH) bin = -1;
        }
        if (bin >= 0) atomicAdd(&s_hist[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x < ORBM_HISTO_LENGTH) { const int v = s_hist[threadIdx.x]; if (v) atomicAdd(&state[48 + threadIdx.x], v); }
    if (threadIdx.x == ORBM_HISTO_LENGTH) { const int v = s_hist[ORBM_HISTO_LENGTH]; if (v) atomicAdd(&state[1], v); }
}

// ComputeThreeMaxima (every block, redundantly) + rejection of the matches outside the three fullest rotation bins
__global__ __launch_bounds__(256) void k_rs_reject(const orbm_query* __restrict__ q, int nq, int cap, const int* __restrict__ choice,
                                                   const float* __restrict__ f_angle, int* __restrict__ owner,
                                                   int* __restrict__ state) {
    MORB_LATENCY_KERNEL_WIDE();
    if (state[0] > cap || state[8 + state[5] - 1] != 0) return;
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
    MORB_LATENCY_KERNEL_WIDE();
    const int NT = n_total_dev ? *n_total_dev : NT_host;
    const bool overflow = state[0] > cap, stuck = state[8 + state[5] - 1] != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int sweeps = 0;
        for (int k = 0; k < state[5]; ++k) sweeps += state[8 + k] ? 1 : 0;
        status[0] = overflow ? 2 : (stuck ? 1 : 0);
        status[1] = (overflow || stuck) ? 0 : state[1];
        status[2] = state[6] ? state[6] : sweeps + 1;   // (rounds of the per-camera form, the slowest camera's)
        status[3] = state[0];
    }
    if (overflow || stuck) return;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < NT) match_of_feature[g] = owner[g];
}

// ---- the monotone resolve for frames beyond one workgroup's LDS, ONE WORKGROUP PER CAMERA (round 4).  A query of the frame
// search looks at the grid of exactly one camera (project_dev.h: one window, `cam`), so its candidates are features of that camera
// and a claim never crosses cameras: the reference's sequential loop (src/ORBmatcher.cc:3502-3614) falls apart into one independent
// loop per camera, each over that camera's queries in their global order.  Workgroup c therefore gathers the queries of camera c
// (ordered compaction over the {blocks | camera << 1} words k_project left), keeps that camera's claim and owner tables in LDS
// (8 bytes per feature: 32 KB for 4000 features, against 256 KB for the whole 8-camera frame, which is what sent these frames to
// one launch per sweep before) and runs the monotone iteration of k_resolve_mono on them with every query in registers (four per
// thread: up to 4096 queries per camera).  What needs the WHOLE frame -- the rotation histogram's three maxima and the rejection
// behind them -- stays with k_rs_reject / k_rs_write, fed through the same `state` words as the per-sweep form:
// this kernel leaves choice[] (global feature index or -1 per query), owner[] per feature, state[0] (longest list), state[1]
// (matches), state[5] = 1 and state[8] = "did not finish" (tables or queries beyond this launch's limits, or out of rounds:
// the exact host fallback takes over, search_finish), state[48..78) the histogram.
constexpr int RSC_RQ = 4;
// n != 0: the queries are camera-contiguous (queries built from the previous frame's features; host lists in camera order) and camera
// c's are [start[c], start[c + 1]) -- the workgroup then neither scans the query words for its camera nor gathers indices (8 x 4000
// queries: 11 + 5 us of the kernel's 100)
struct QRanges { int n; int start[65]; };
__global__ __launch_bounds__(1024) void k_resolve_cams(FrameDev F, const int* __restrict__ f_cam_start, const int2* __restrict__ qmeta, int nq,
                                                       int cap, int nf_cap, const int* __restrict__ cand_idx,
                                                       const uint16_t* __restrict__ cand_dist, const int* __restrict__ cand_count,
                                                       const uint8_t* __restrict__ occupied, const float* __restrict__ f_angle, int th_high,
                                                       int check_ori, int max_it, const int* __restrict__ topk, int* __restrict__ state,
                                                       int* __restrict__ match_of_feature, int* __restrict__ status, int tagb, int n_res,
                                                       MergeJob MJ, QRanges QR) {
    MORB_LATENCY_KERNEL();
    if ((int)blockIdx.x >= n_res) {   // workgroups behind the cameras' (isolated steps): the slice merge of the camera-pair top-2, as in k_resolve_mono
        const int mq = MJ.d_range ? MJ.d_range[2] : MJ.nq;
        const int qi_ = ((int)blockIdx.x - n_res) * blockDim.x + threadIdx.x;
        if (qi_ < mq) top2_merge_query(MJ.p_idx, MJ.p_best, MJ.p_second, MJ.S, mq, qi_, MJ.o_idx, MJ.o_best, MJ.o_second);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(MJ.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) int s_claim[];  // [0, nf_cap): lowest blocking claimant (global query index); [nf_cap, 2 nf_cap): owner
    __shared__ int s_hist[ORBM_HISTO_LENGTH];
    __shared__ int s_red, s_cnt, s_code, s_last;
    __shared__ int s_keep[3];
    __shared__ int s_flag[3];
    __shared__ int s_wtot[16];
    constexpr int K = RESOLVE_K, RQ = RSC_RQ, T = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cam = blockIdx.x;
    const int f0 = f_cam_start[cam], nf = f_cam_start[cam + 1] - f0;
    int* s_owner = s_claim + nf_cap;
    unsigned short* l_q = reinterpret_cast<unsigned short*>(s_owner + nf_cap);   // the camera's queries, ascending (global indices < 65536)
    const int* tk_g = topk + K * nq;
    MORB_PHASE(g_ph_res, 0);   // (stamps of camera 0's workgroup: start, counted, gathered, set up, rounds done, owners, end; slot 1 = 2: this layout)
    if (tid < 3) s_flag[tid] = 0;
    if (tid == 0) { s_red = 0; s_cnt = 0; }
    if (tid < ORBM_HISTO_LENGTH) s_hist[tid] = 0;
    for (int g = tid; g < min(nf, nf_cap); g += T) { s_claim[g] = 0x7fffffff; s_owner[g] = -1; }
    // the camera's queries: every WAVE takes a contiguous run of the query list (at most 4096 entries: nq < 65536) and reads it 64
    // entries per load, coalesced, all 32 loads of a trip in flight (the words k_project wrote sit in another XCD's L2: every trip is a trip
    // to memory, ~2 us; a run per THREAD -- the first form of this kernel -- also asked for 64 different lines with every load).  A
    // lane keeps "my entry of group u is this camera's" as bit u of a 64-bit word, so the second pass -- the ordered store of the
    // indices once the waves' totals are known -- needs no memory.
    // Queries that name no camera of the frame have no candidates (project_dev.h): -1 from workgroup 0.
    const int q_first = QR.n ? QR.start[cam] : -1;
    if (QR.n) {
        if (tid == 0) s_cnt = QR.start[cam + 1] - q_first;
        __syncthreads();
    } else {
    const int per_wave = (((nq + 15) / 16) + 63) & ~63, w0 = min(nq, wave * per_wave), w1 = min(nq, w0 + per_wave);
    unsigned long long mybits = 0;
    int mine = 0;
    for (int b0 = 0; b0 < 64; b0 += 32) {   // (one trip for up to 32 768 queries, two beyond)
        if (w0 + 64 * b0 >= w1) break;       // (wave-uniform)
        int qc[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int i = w0 + 64 * (b0 + u) + lane; qc[u] = i < w1 ? qmeta[i].x >> 1 : -0x40000000; }
        unsigned int bits = 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const bool m = qc[u] == cam;
            bits |= (unsigned int)m << u;
            mine += __popcll(__ballot(m));
        }
        mybits |= (unsigned long long)bits << b0;
    }
    if (lane == 0) s_wtot[wave] = mine;
    __syncthreads();
    MORB_PHASE(g_ph_res, 2);
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_wtot[w];
    if (tid == T - 1) s_cnt = off + mine;
    for (int u = 0; u < 64; ++u) {
        if (w0 + 64 * u >= w1) break;    // (wave-uniform)
        const bool m = (mybits >> u) & 1ull;
        const unsigned long long mask = __ballot(m);
        const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
        if (m && pos < RQ * T) l_q[pos] = (unsigned short)(w0 + 64 * u + lane);
        off += __popcll(mask);
    }
    __syncthreads();
    }
    MORB_PHASE(g_ph_res, 3);
    const int nqc = s_cnt;
    int sl[RQ][K], cur[RQ], pos[RQ], flr[RQ], qi[RQ], bin_of[RQ];
    int mx = 0, it = 1, maxcount = 0, acc = 0;
    // 0 done, 1 = did not finish (limits, rounds), 2 = a candidate list overflowed -- the cameras' workgroups MEET below whatever happened
    auto run = [&]() -> int {
    if (nf > nf_cap || nqc > RQ * T) return 1;   // beyond this launch's limits (the host sized them from capacities: cannot happen unless those lied)
#pragma unroll
    for (int b = 0; b < RQ; ++b) {
        const int j = b * T + tid;
        cur[b] = -1; pos[b] = 0; flr[b] = 0; qi[b] = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < K; ++k) sl[b][k] = 0xffff;
        if (j < nqc) {
            // set-up of one query, round 0 riding along as in k_resolve_mono: the head of the shortlist is taken and claimed
            const int i = q_first >= 0 ? q_first + j : (int)l_q[j];
            qi[b] = i;
            const int cnt_i = cand_count[i];
            mx = max(mx, cnt_i);
            flr[b] = (qmeta[i].x & 1) | (topk[(2 * K) * nq + i] > K ? 2 : 0);
#pragma unroll
            for (int k = 0; k < K; ++k) { const int g = tk_g[k * nq + i]; sl[b][k] = g < 0 ? 0xffff : g - f0; }
            const bool has = sl[b][0] != 0xffff;
            cur[b] = has ? sl[b][0] : -1;
            if (has && (flr[b] & 1)) atomicMin(&s_claim[sl[b][0]], i);
        }
    }
    mx = (int)(0x7fffffffu - wave_min_u32(0x7fffffffu - (unsigned)mx));
    if (lane == 0) atomicMax(&s_red, mx);
    __syncthreads();
    maxcount = s_red;
    MORB_PHASE(g_ph_res, 4);
    if (tid == 0 && maxcount > 0) atomicMax(&state[0], maxcount);
    if (maxcount > cap) return 2;   // (reported through the status words, the search is repeated with more room)
    // one pass over this thread's queries: k_resolve_mono's, with local feature indices into the tables and global query indices
    // as the claims' values
    auto pass = [&]() -> bool {
        bool disp[RQ], need_rescan[RQ];
        int cl[RQ];
#pragma unroll
        for (int b = 0; b < RQ; ++b) cl[b] = s_claim[cur[b] >= 0 ? cur[b] : 0];
#pragma unroll
        for (int b = 0; b < RQ; ++b) { disp[b] = cur[b] >= 0 && cl[b] < qi[b]; need_rescan[b] = false; }
        bool any = false;
#pragma unroll
        for (int b = 0; b < RQ; ++b) any |= disp[b];
        if (!__ballot(any)) return false;
        {   // (slots none of whose queries is displaced are skipped by the wave: see k_resolve_mono)
            int ck[RQ][K];
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                if (!__ballot(disp[b])) continue;
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const bool want = disp[b] && k > pos[b] && sl[b][k] != 0xffff;
                    const int v = s_claim[want ? sl[b][k] : 0];
                    ck[b][k] = want ? v : -1;
                }
            }
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                if (!__ballot(disp[b])) continue;
                if (!disp[b]) continue;
                const int i = qi[b];
                int nk = K;
#pragma unroll
                for (int k = K - 1; k >= 1; --k) if (ck[b][k] >= i) nk = k;
                int ne = 0xffff;
#pragma unroll
                for (int k = 1; k < K; ++k) if (nk == k) ne = sl[b][k];
                if (nk < K) {
                    cur[b] = ne; pos[b] = nk;
                    if (flr[b] & 1) atomicMin(&s_claim[ne], i);
                } else {
                    cur[b] = -1; pos[b] = K;
                    need_rescan[b] = (flr[b] & 2) != 0;   // the shortlist ran dry while the list is longer: rescanned by the wave
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RQ; ++b) {
            unsigned long long todo = __ballot(need_rescan[b]);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int q = __builtin_amdgcn_readlane(qi[b], src);
                const int full = cand_count[q];
                int k1 = 0x7fffffff, g1 = -1;
                for (int k0 = 0; k0 < full; k0 += 64) {
                    const int k = k0 + lane;
                    int key = 0x7fffffff, gg = -1;
                    if (k < full) {
                        const int gglob = cand_idx[k * nq + q];
                        const int d = cand_dist[k * nq + q];
                        gg = gglob - f0;
                        bool avail = !(occupied && occupied[gglob]);
                        if (s_claim[gg] < q) avail = false;
                        if (avail) key = (d << 16) | k;
                    }
                    const int m1 = (int)wave_min_u32((unsigned)key);
                    if (m1 < k1) { k1 = m1; g1 = __builtin_amdgcn_readlane(gg, __ffsll((long long)__ballot(key == m1)) - 1); }
                }
                if (lane == src && k1 != 0x7fffffff && (k1 >> 16) <= th_high) {
                    cur[b] = g1;
                    if (flr[b] & 1) atomicMin(&s_claim[g1], q);
                }
            }
        }
        return true;
    };
    int changed = 1;
    for (; it < max_it && changed; ++it) {
        if (tid == 0) s_flag[(it + 1) % 3] = 0;
        bool ch = false;
        for (int guard = 0; guard < 4096; ++guard) {
            if (!pass()) break;
            ch = true;
        }
        if (ch && lane == 0) s_flag[it % 3] = 1;
        __syncthreads();
        changed = s_flag[it % 3];
    }
    if (changed) return 1;  // ran out of rounds
    MORB_PHASE(g_ph_res, 5);
    // owners (the last claimant in query order), this camera's share of the rotation histogram and of the match count
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
#pragma unroll
    for (int b = 0; b < RQ; ++b) {
        const int i = qi[b], c = cur[b];
        bin_of[b] = -1;
        if (c >= 0) { ++acc; atomicMax(&s_owner[c], i); }
        if (check_ori) {
            int bin = -1;
            if (c >= 0) {
                float rot = __int_as_float(qmeta[i].y) - f_angle[c + f0];
                if (rot < 0.0) rot += 360.0f;
                bin = (int)roundf(rot * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin < 0 || bin >= ORBM_HISTO_LENGTH) bin = -1;
            }
            bin_of[b] = bin;
            if (bin >= 0) atomicAdd(&s_hist[bin], 1);
        }
    }
    return 0;
    };
    const int code = run();
    __syncthreads();   // (every wave's owners and histogram adds have landed)
    // ---- the cameras meet: what needs the WHOLE frame -- the rotation histogram's three maxima -- is summed in `state` (device memory:
    // [0] longest list, [1] matches, [6] rounds, [8] did-not-finish, [9] arrived, [10] finished, [48..78) histogram; all zero between launches:
    // the last workgroup leaves it that way).  A workgroup that could not finish still arrives.
    if (code == 0 && tid < ORBM_HISTO_LENGTH) { const int v = s_hist[tid]; if (v) atomicAdd(&state[48 + tid], v); }
    if (tid == 64 && code == 1) atomicOr(&state[8], 1);
    if (tid == 65) atomicMax(&state[6], it);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(&state[9], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&state[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_res && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1 << 22)) atomicOr(&state[8], 1);   // (a camera's workgroup never came: the host's exact pass takes over)
        const int gmax = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_code = gmax > cap ? 2 : (__hip_atomic_load(&state[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0);
    }
    __syncthreads();
    const int all = s_code;
    MORB_PHASE(g_ph_res, 6);
    if (all == 0) {
        int kept = 0;
        if (check_ori) {
            if (tid < 64) {   // ComputeThreeMaxima (reference src/ORBmatcher.cc:3948-3989) over the frame's histogram, by every camera alike
                const int sv = tid < ORBM_HISTO_LENGTH ? __hip_atomic_load(&state[48 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
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
#pragma unroll
            for (int b = 0; b < RQ; ++b) {
                const int c = cur[b], bin = bin_of[b];
                if (c >= 0 && bin >= 0 && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) { s_owner[c] = -2; --acc; }   // (every writer stores -2; owners were settled before the meeting)
            }
        }
        kept = __builtin_amdgcn_readlane(wave_incl_scan(acc), 63);
        if (lane == 0 && kept) atomicAdd(&state[1], kept);
        __syncthreads();
        const int NT = F.n_total_dev ? *F.n_total_dev : F.n_total;
        for (int g = tid; g < nf; g += T) match_of_feature[f0 + g] = tagb ? (tagb | (s_owner[g] + 2)) : s_owner[g];
        if (tagb && cam == 0) for (int g = NT + tid; g < F.n_total; g += T) match_of_feature[g] = tagb | 1;   // (see k_resolve: no stale tag can match)
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&state[10], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n_res - 1;
    __syncthreads();
    if (!s_last) return;
    // the last camera out: the result words (behind the merging workgroups of the same launch, as in k_resolve_mono), then `state` back to zero
    bool merged = true;
    if (MJ.S > 1 && MJ.done) {
        if (tid == 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(MJ.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - MJ.target) < 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
            s_code = spins < (1 << 22) ? 1 : 0;
        }
        __syncthreads();
        merged = s_code != 0;
    }
    if (tid == 0) {
        const int matches = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[1] = tagb | (all ? 0 : matches); status[2] = tagb | __hip_atomic_load(&state[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[3] = tagb | __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[0] = tagb | (all ? all : (merged ? 0 : 3));
    }
    __syncthreads();
    if (tid < RS_STATE_INTS) state[tid] = 0;
    MORB_PHASE(g_ph_res, 7);
#ifdef MORB_PHASE_CLOCKS
    if (tid == 0) { g_ph_res[1] = 2; g_ph_res[62] = (unsigned long long)it; }
#endif
}

// Multi-GPU exchange: `gathered` holds one block per rank (rank order), each = cap_rows descriptor rows (the rank's
// cameras packed back to back) + the count trailer.  The rows in use are copied into one contiguous list in global camera
// order; block (0, 0) also writes the camera starts, the {features, first query, queries} triple of rank `rank`, and a
// copy of all counts into mapped pinned memory.  Every block recomputes the few prefix sums it needs from the trailers.
}  // namespace

int morb::search_raise_lds_limits() {
    const void* fns[] = {(const void*)k_resolve<true, false>, (const void*)k_resolve<false, false>, (const void*)k_resolve<true, true>,
                         (const void*)k_resolve<false, true>, (const void*)k_resolve_mono<2, true>, (const void*)k_resolve_mono<2, false>,
                         (const void*)k_resolve_mono<4, true>, (const void*)k_resolve_mono<4, false>, (const void*)k_resolve_cams};
    for (const void* fn : fns) MORB_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return ORB_OK;
}

int morb::phases_resolve(unsigned long long* out64) {
#ifdef MORB_PHASE_CLOCKS
    if (out64[0] == 0xC4A26Eull)   // (magic in out64[0]: the per-sweep change counters instead of the clocks)
        return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_chg), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ph_res), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
    (void)out64; return -1;
#endif
}

// k_project into m->d_i0 (idx) / d_u16 (dist) / d_i1 (count); optionally copied to the pinned host mirrors
static int run_project(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int cap, int gate_right,
                       int with_dist, bool upload_queries, bool to_host, int transposed = 0,
                       const uint8_t* d_occupied = nullptr, int* d_topk = nullptr, int short_th = 256,
                       const float* d_inv_sigma2 = nullptr, const orbm_query* q_device_visible = nullptr, int2* d_qmeta = nullptr,
                       const orbm_window* d_win2 = nullptr, const SideJob* side = nullptr, const MotionSrc* msrc = nullptr,
                       bool msrc_records = false, MergeJob* defer_merge = nullptr) {
    int rc;
    if ((rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query))) || (rc = m->d_i0.reserve((size_t)nq * cap)) ||
        (rc = m->d_u16.reserve((size_t)nq * cap)) || (rc = m->d_i1.reserve(nq)))
        return rc;
    if (upload_queries)
        MORB_HIP(hipMemcpyAsync(m->d_queries.p, q, (size_t)nq * sizeof(orbm_query), hipMemcpyHostToDevice, m->stream));
    ProjectArgs A;
    A.F = f->dev(); A.q = q_device_visible ? q_device_visible : (const orbm_query*)m->d_queries.p; A.nq = nq; A.cap = cap;
    A.gate_right = gate_right; A.with_dist = with_dist; A.transposed = transposed;
    A.cand_idx = m->d_i0.p; A.cand_dist = m->d_u16.p; A.cand_count = m->d_i1.p;
    A.occupied = d_occupied; A.topk = d_topk; A.short_th = short_th; A.inv_sigma2 = d_inv_sigma2; A.qmeta = d_qmeta; A.win2 = d_win2;
    if (msrc) {   // the kernel builds its queries from the previous frame's arrays (MotionSrc, matcher_internal.h)
        A.from_motion = 1; A.ms = *msrc; A.q = nullptr;
        A.ms.rec_out = msrc_records ? (orbm_query*)m->d_queries.p : nullptr;
    }
    if (side) {   // the caller's side work (camera-pair top-2, result mirror) shares the launch: see k_project_side
        if ((rc = launch_project_side(m->stream, A, *side, defer_merge))) return rc;
    } else {
        hipLaunchKernelGGL(k_project, dim3((nq + 3) / 4), dim3(256), 0, m->stream, A);
    }
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
    MORB_HIP(hipMemcpyAsync(m->h_i1.p, m->d_i1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
    MORB_HIP(hipStreamSynchronize(m->stream));
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

int morb::search_enqueue(orbm_matcher* m, SearchJob& J, bool queries_already_on_device) {
    const int n = J.cur->n_total;
    J.device_path = false; J.pollable = false; J.multi = false;
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
    if (J.q_dev && multi && !J.msrc) {
        if ((rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query)))) return rc;
        MORB_HIP(hipMemcpyAsync(m->d_queries.p, J.q_dev, (size_t)nq * sizeof(orbm_query), hipMemcpyDefault, m->stream));
    }
    MergeJob MJ{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};   // (S > 1 after run_project: a merge waits for its carrier)
    if ((rc = run_project(m, cur, J.q, nq, cap, 1, 1, !queries_already_on_device && !J.q_dev && !J.msrc, false, /*transposed=*/1, d_occ,
                          m->d_claim.p, J.points ? 256 : th_high, nullptr, q_in_place, m->d_qmeta.p, J.win2_dev, J.side, J.msrc, multi,
                          (J.side && !multi && !J.points) ? &MJ : nullptr)))
        return rc;
    J.side = nullptr;   // (a retry of the search with more room per query does not repeat the side work)
    J.seq = 0;
    if (J.want_tags) { m->resolve_seq = m->resolve_seq % 2047 + 1; J.seq = m->resolve_seq; }   // 1..2047, never 0
    // Frame searches (one window per query, no ratio test) resolve PER CAMERA in one launch when every camera's tables fit a workgroup's
    // LDS and its queries a workgroup's registers (k_resolve_cams: a query looks at one camera's grid, so claims never cross cameras;
    // the cameras' workgroups meet once, in the kernel, for the rotation histogram).  Large rigs (tables of the whole frame beyond LDS)
    // always take it.  Smaller rigs only when asked to (MORB_RS_PER_CAMERA = N > 1: from N queries on; default never): measured in round 6
    // (VERDICT r05 #4 asked for it), 2 x 2000 queries take 38 us per camera against 31 in one workgroup and 4 x 1000 31 against 28 -- the
    // rounds do get shorter (11 us against 17), but the scan for the camera's queries (4 us: one trip to memory), the meeting (6 us over
    // L2 between XCDs) and the separate tail cost more than that (profiles/r06/notes_experiments.md).
    // MORB_RS_PER_CAMERA: 0 = never (large rigs take one launch per sweep), 1 = large rigs (default), N > 1 = also smaller rigs from N queries on
    static const int cam_env = [] { const char* e = getenv("MORB_RS_PER_CAMERA"); return e ? atoi(e) : 1; }();
    const int cams_min_q = cam_env > 1 ? cam_env : 0x7fffffff;
    int nf_cap = 0, q_cam_max = J.q_cam_max;
    QRanges QR; QR.n = 0;
    const bool starts_ok = cur->camera_major && (int)cur->cam_start.size() == cur->n_cams + 1 && cur->cam_start[cur->n_cams] == n;
    const bool cams_want = cam_env && !J.points && !J.win2_dev && starts_ok && (multi || (cur->n_cams >= 2 && nq >= cams_min_q));
    if (cams_want) {
        for (int c = 0; c < cur->n_cams; ++c) nf_cap = std::max(nf_cap, cur->cam_start[c + 1] - cur->cam_start[c]);
        if (!q_cam_max && !J.msrc && J.q) {   // (host records: count them -- and see whether they come camera after camera)
            std::vector<int>& per = m->rs_cam_count;
            per.assign((size_t)cur->n_cams, 0);
            bool sorted = cur->n_cams <= 64;
            int last = 0;
            for (int i = 0; i < nq; ++i) {
                const int c = J.q[i].cam;
                if (c >= 0 && c < cur->n_cams) q_cam_max = std::max(q_cam_max, ++per[c]);
                if (c < last || c >= cur->n_cams) sorted = false; else last = c;
            }
            if (!q_cam_max) q_cam_max = 1;
            if (sorted) { QR.n = cur->n_cams; int b = 0; for (int c = 0; c < cur->n_cams; ++c) { QR.start[c] = b; b += per[c]; } QR.start[cur->n_cams] = b; }
        } else if (J.q_cam_start && cur->n_cams <= 64 && J.q_cam_start[cur->n_cams] == nq) {
            QR.n = cur->n_cams;
            for (int c = 0; c <= cur->n_cams; ++c) QR.start[c] = J.q_cam_start[c];
        }
    }
    const size_t lds_cam = (size_t)8 * nf_cap + (size_t)2 * RSC_RQ * 1024;
    const bool cams_fit = cams_want && nf_cap > 0 && q_cam_max > 0 && q_cam_max <= RSC_RQ * 1024 && nq < 65536 && lds_cam <= 150 * 1024;
    auto launch_cams = [&](MergeJob MJc, int merge_blocks) -> int {
        int rcl;
        if (!m->d_rsync.p) {   // the meeting's words: zero between launches (the last workgroup of a launch leaves them so)
            if ((rcl = m->d_rsync.reserve(RS_STATE_INTS))) return rcl;
            MORB_HIP(hipMemsetAsync(m->d_rsync.p, 0, RS_STATE_INTS * sizeof(int), m->stream));
        }
        hipLaunchKernelGGL(k_resolve_cams, dim3(cur->n_cams + merge_blocks), dim3(1024), lds_cam, m->stream, cur->dev(), (const int*)cur->b->d_cam_start.p,
                           (const int2*)m->d_qmeta.p, nq, cap, nf_cap, (const int*)m->d_i0.p, (const uint16_t*)m->d_u16.p,
                           (const int*)m->d_i1.p, d_occ, (const float*)cur->b->d_ang.p, th_high, J.check_ori, 4096,
                           (const int*)m->d_claim.p, m->d_rsync.p, m->h_match.dp + 4, m->h_match.dp, J.seq << 20, cur->n_cams, MJc, QR);
        MORB_HIP(hipGetLastError());
        return ORB_OK;
    };
    if (multi) {
        int* tab0 = m->d_gclaim.p; int* tab1 = tab0 + n; int* state = tab1 + n;
        if (!cams_fit) MORB_HIP(hipMemsetAsync(state, 0, RS_STATE_INTS * sizeof(int), m->stream));
        const int nb_all = (std::max(n, nq) + 255) / 256, nb_q = (nq + 255) / 256, nb_f = (n + 255) / 256;
        // One launch per sweep, enqueued blind: as many as the stream of searches has needed lately plus four (12 at least, 24 at
        // most; 8 x 4000 features converge in 10-11), because a sweep that has nothing to do still costs its launch (~4.7 us each
        // on the critical path of the step).  A search that does not converge in its allotment is finished by the exact host
        // fallback and the next one gets the full 24 again.
        if (cams_fit) {   // one launch, one workgroup per camera (k_resolve_cams): meeting, rejection and the tagged result words inside
            if ((rc = launch_cams(MergeJob{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr}, 0))) return rc;
            J.device_path = true;
            J.pollable = J.seq != 0;
            return ORB_OK;
        }
        J.seq = 0;   // (the per-sweep form's result words carry no tags: the host synchronises the stream)
        const int n_sweeps = std::min(RS_MAX_SWEEPS, std::max(12, m->rs_sweeps_hint));
        hipLaunchKernelGGL(k_rs_init, dim3(nb_all), dim3(256), 0, m->stream, n, nq, tab0, tab1, m->d_match.p, m->d_choice.p,
                           (const int*)m->d_i1.p, state, n_sweeps);
        J.multi = true;
        for (int it = 0; it < n_sweeps; ++it) {
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
    static const int mono_mode = [] { const char* e = getenv("MORB_RESOLVE_MONO"); return e ? atoi(e) : 2; }();   // 0 Jacobi sweeps, 1 monotone per-wave passes, 2 monotone worklist
    const bool mono_env = mono_mode != 0;
    // (the worklist rounds pay from four queries per thread on: 2 x 2000 queries 30.7 -> 24.7 us, 4 x 1000 29.3 -> 27.3; at 2 x 1000 a
    //  round's two barriers cost more than the waves' own passes: 17.3 -> 20.5 us, so small rigs keep those -- profiles/r06/notes_experiments.md)
    const int mono_worklist = mono_mode == 2 && nq > 2048 ? 1 : 0;
    // The frame search's monotone resolve keeps less per query (16-bit features, no distances; mono_lds): it takes frames of up to
    // ~6000 features / queries (4 x 1000, 2 x 2000: where k_resolve's Jacobi form keeps its query state in HBM and costs 53 us),
    // four queries per thread in registers beyond 2048 queries, the rotation angles from HBM when they do not fit next to the rest.
    auto mono_lds = [&](bool ang) {
        const size_t nq2 = ((size_t)nq + 1) & ~(size_t)1;
        return (size_t)n * 8 + nq2 * 4 + (size_t)RESOLVE_K * nq2 * 2 + (((size_t)nq + 3) & ~(size_t)3) + (ang ? ((size_t)nq + (size_t)n) * 4 : 0) + nq2 * 2 + 16;   // (+ the worklist)
    };
    const bool will_mono = !J.points && mono_env && n < 65535 && mono_lds(false) <= 150 * 1024;
    if (MJ.S > 1 && !will_mono && !cams_fit) { if ((rc = launch_merge(m->stream, MJ))) return rc; MJ.S = 0; }   // (no carrier after all: a launch of its own)
    if (cams_fit) {
        const int merge_blocks = MJ.S > 1 ? (MJ.nq + 1023) / 1024 : 0;   // (behind the cameras' workgroups)
        if (merge_blocks) {
            if ((rc = m->d_mergecnt.reserve(4))) return rc;
            if (!m->merge_ready) { MORB_HIP(hipMemsetAsync(m->d_mergecnt.p, 0, 16, m->stream)); m->merge_ready = true; m->merge_target = 0; }
            m->merge_target += (unsigned)merge_blocks;
            MJ.done = reinterpret_cast<unsigned*>(m->d_mergecnt.p); MJ.target = m->merge_target;
        }
        if ((rc = launch_cams(MJ, merge_blocks))) return rc;
        MJ.S = 0;   // (carried)
    } else if (will_mono) {
        const bool ang = mono_lds(true) <= 150 * 1024;
        const size_t ml = mono_lds(ang);
#define MORB_MONO_LAUNCH(RQ_, ANG_)                                                                                                       \
        hipLaunchKernelGGL((k_resolve_mono<RQ_, ANG_>), dim3(1 + merge_blocks), dim3(1024), ml, m->stream, cur->dev(), (const int2*)m->d_qmeta.p, nq, cap, \
                           (const int*)m->d_i0.p, (const uint16_t*)m->d_u16.p, (const int*)m->d_i1.p, d_occ,                                \
                           (const float*)cur->b->d_ang.p, th_high, J.check_ori, 4096, (const int*)m->d_claim.p, m->h_match.dp + 4,          \
                           m->h_match.dp, J.seq << 20, MJ, mono_worklist)
        const int merge_blocks = MJ.S > 1 ? (MJ.nq + 1023) / 1024 : 0;   // (behind workgroup 0, the resolve)
        if (merge_blocks) {
            if ((rc = m->d_mergecnt.reserve(4))) return rc;
            if (!m->merge_ready) { MORB_HIP(hipMemsetAsync(m->d_mergecnt.p, 0, 16, m->stream)); m->merge_ready = true; m->merge_target = 0; }
            m->merge_target += (unsigned)merge_blocks;
            MJ.done = reinterpret_cast<unsigned*>(m->d_mergecnt.p); MJ.target = m->merge_target;
        }
        if (nq <= 2048) { if (ang) MORB_MONO_LAUNCH(2, true); else MORB_MONO_LAUNCH(2, false); }
        else { if (ang) MORB_MONO_LAUNCH(4, true); else MORB_MONO_LAUNCH(4, false); }
        MJ.S = 0;   // (carried)
#undef MORB_MONO_LAUNCH
    } else if (ldsq) { if (J.points) MORB_RESOLVE_LAUNCH(true, true); else MORB_RESOLVE_LAUNCH(false, true); }
    else { if (J.points) MORB_RESOLVE_LAUNCH(true, false); else MORB_RESOLVE_LAUNCH(false, false); }
#undef MORB_RESOLVE_LAUNCH
    MORB_HIP(hipGetLastError());  // status + matches are written by the kernel into the mapped pinned buffer
    J.device_path = true;
    J.pollable = J.seq != 0;
    return ORB_OK;
}

// After the stream has been synchronised.  match_of_feature may alias m->h_match.p + 4 (then nothing is copied).
int morb::search_finish(orbm_matcher* m, SearchJob& J, int32_t* match_of_feature, int* nmatches) {
    const int n = J.cur->n_total;
    *nmatches = 0;
    if (J.nq == 0 || n == 0) { for (int g = 0; g < n; g++) match_of_feature[g] = -1; return ORB_OK; }
    if (!J.device_path) {
        m->last_status[0] = -1; m->last_status[1] = 0; m->last_status[2] = 0; m->last_status[3] = 0;  // (host path)
    }
    auto need_q = [&] { if (J.q_fill) { J.q_fill(J.q_fill_ctx); J.q_fill = nullptr; } };   // (host records of a motion step, on demand)
    if (!J.device_path) {
        need_q();
        return host_resolve(m, J.cur, J.q, J.nq, J.occupied, J.points, J.nnratio, J.th_high, J.check_ori, 64, match_of_feature, nmatches, J.win2_dev);
    }
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
        if (J.multi) m->rs_sweeps_hint = status == 0 ? m->last_status[2] + 4 : RS_MAX_SWEEPS;   // (sweeps the next multi-workgroup resolve enqueues)
        // any other status than "done": nothing else of this launch is read before the stream has drained (the words of a tagged launch
        // say the resolve's workgroup is over, not that the whole launch is -- status 3: its wait for the merging workgroups gave up)
        if (status != 0 && !synced) { MORB_HIP(hipStreamSynchronize(m->stream)); synced = true; }
        if (status == 0 || status == 3) { m->last_status[0] = 0; break; }
        if (status == 2) {  // a candidate list overflowed: retry with room for the longest one
            J.cap = (m->last_status[3] + 63) & ~63;
            int rc = search_enqueue(m, J, /*queries_already_on_device=*/true);
            if (rc) return rc;
            MORB_HIP(hipStreamSynchronize(m->stream));
            synced = true;
            continue;
        }
        // not converged within the sweep limit: exact host fallback
        need_q();
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
    // (synchronised, not polled: watching the tagged result words arrive returns 3 us sooner, but the callers above -- the host classes --
    //  rely on an IDLE stream behind a search: their next frame upload recycles buffers and their extractors order themselves behind this
    //  stream; with the kernel's tail still on it both cost 15-20 us, measured in round 6)
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

