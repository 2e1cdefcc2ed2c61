// project_dev.h -- device body of k_project (one wave per query), shared by search.hip (the kernel proper) and hamming.hip
// (k_project_side: the same waves next to the camera-pair top-2 and the result mirror of an isolated orbf_step).
#pragma once
#include "orb_common.h"
#include "matcher_internal.h"
#include "hamming_dev.h"

namespace morb {

struct ProjectArgs {
    FrameDev F; const orbm_query* q; int nq, cap, gate_right, with_dist, transposed;
    int* cand_idx; uint16_t* cand_dist; int* cand_count;
    const uint8_t* occupied; int* topk; int short_th;
    const float* inv_sigma2; int2* qmeta; const orbm_window* win2;
    int from_motion = 0; MotionSrc ms;   // from_motion != 0: `q` is unused, the queries come from `ms`
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
__device__ __forceinline__ void project_wave(const ProjectArgs& A, const int qi, const int lane) {
    const FrameDev& F = A.F;
    const orbm_query* __restrict__ q = A.q;
    const int nq = A.nq, cap = A.cap, gate_right = A.gate_right, with_dist = A.with_dist, transposed = A.transposed;
    int* __restrict__ cand_idx = A.cand_idx; uint16_t* __restrict__ cand_dist = A.cand_dist; int* __restrict__ cand_count = A.cand_count;
    const uint8_t* __restrict__ occupied = A.occupied; int* __restrict__ topk = A.topk; const int short_th = A.short_th;
    const float* __restrict__ inv_sigma2 = A.inv_sigma2; int2* __restrict__ qmeta = A.qmeta; const orbm_window* __restrict__ win2 = A.win2;

    float x, y, r, ur, q_angle;
    int minLevel, maxLevel, cam, q_cam, q_blocks;
    uint4 q0, q1;
    if (A.from_motion) {
        const MotionSrc& M = A.ms;
        const int oct = M.octave[qi];
        const float dep = M.depth[qi];
        q0 = M.desc[2 * qi]; q1 = M.desc[2 * qi + 1];
        q_angle = M.angle[qi];
        x = M.x[qi] + M.du; y = M.y[qi] + M.dv;
        r = M.scale[oct] * M.th;
        const float inv = dep > 0 ? 1.0f / dep : 0.0f;
        ur = x - M.mbf * inv;
        minLevel = oct - 1; maxLevel = oct + 1;
        cam = 0;
        for (int c = 1; c < M.n_cams; ++c) cam += (M.cam_start[c] <= qi) ? 1 : 0;
        q_blocks = 1;
    } else {
        const orbm_query* Q = q + qi;
        x = Q->u; y = Q->v; r = Q->radius; ur = Q->ur;
        minLevel = Q->min_level; maxLevel = Q->max_level; cam = Q->cam;
        const uint32_t* qd = reinterpret_cast<const uint32_t*>(Q->desc);
        q0 = make_uint4(qd[0], qd[1], qd[2], qd[3]); q1 = make_uint4(qd[4], qd[5], qd[6], qd[7]);
        q_blocks = Q->blocks; q_angle = Q->angle;
    }
    q_cam = cam;

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
                // Everything the gates and the distance need of item g is requested HERE, in one round trip: octave, position,
                // right coordinate and the descriptor.  (Rounds 1-2 loaded each field inside the gate before it -- level,
                // then window, then right coordinate, then the descriptor of the survivors: four dependent trips through a
                // memory system that other streams' kernels keep busy, 20 us per launch next to two extraction chains against
                // 9.5 us alone.  A window holds about twice as many items as pass its gates: the extra bytes are nothing
                // next to the trips.)  Lanes beyond the chunk read item 0 (every frame array has at least one element).
                const int oct = F.octave[g];
                const float gx = F.un_x[g], gy = F.un_y[g], urg = F.uright[g];
                uint4 gd0 = make_uint4(0, 0, 0, 0), gd1 = gd0;
                if (with_dist) { gd0 = F.desc[2 * g]; gd1 = F.desc[2 * g + 1]; }
                bool pass = valid;
                if (bCheckLevels) {
                    if (oct < minLevel) pass = false;
                    if (maxLevel >= 0 && oct > maxLevel) pass = false;
                }
                {
                    const float distx = gx - x, disty = gy - y;
                    pass = pass && fabsf(distx) < r && fabsf(disty) < r;
                }
                if (gate_right == 1) {
                    if (urg > 0 && fabsf(ur - urg) > r) pass = false;   // a NaN `ur` never closes this gate
                }
                if (pass && gate_right == 2) {   // Fuse's reprojection-error gate (src/ORBmatcher.cc:2118-2143)
                    const float kpr = urg;
                    const float ex = x - gx, ey = y - gy;
                    if (kpr >= 0) {
                        const float er = ur - kpr;
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * inv_sigma2[oct]) > 7.8) pass = false;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * inv_sigma2[oct]) > 5.99) pass = false;
                    }
                }
                const unsigned long long mask = __ballot(pass);
                const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                int dist = 0;
                if (pass && with_dist) dist = ham256(q0, q1, gd0, gd1);
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
        if (qmeta) qmeta[qi] = make_int2((q_blocks ? 1 : 0) | (q_cam << 1), __float_as_int(q_angle));   // {blocks | camera << 1, angle}
        if (A.from_motion && A.ms.rec_out) { A.ms.rec_out[qi].blocks = q_blocks; A.ms.rec_out[qi].angle = q_angle; }
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

}  // namespace morb
