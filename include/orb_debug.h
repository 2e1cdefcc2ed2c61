/* orb_debug.h -- inspection, test and bench hooks of libmorb.so.  Nothing a SLAM caller needs: the parity tests read intermediate
 * stages through these, bench.py times single kernels with them, the multi-GPU tests probe the exchange.  Kept out of orbx.h / orbm.h /
 * orbf.h so that those three headers are the product's interface and nothing else. */
#ifndef ORB_DEBUG_H
#define ORB_DEBUG_H
#include "orbx.h"
#include "orbm.h"
#include "orbf.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- extractor: stage inspection for the level-by-level parity tests -------------------------------------------------- */
/* (orbx_debug_level, a pyramid level of the last run, stayed in orbx.h: ORBextractor::mvImagePyramid is filled through it) */
/* candidates handed to the quadtree (x, y relative to (16,16); response = score), cell-major order */
int orbx_debug_candidates(orbx_extractor* ex, int cam, int level, orb_keypoint* out, int cap, int* n);
/* host-only: the library's quadtree (DistributeOctTree, reference src/ORBextractor.cc:540-764) on caller-supplied
 * candidates (x, y relative to (16,16), integral; response); runs without a GPU.  *n_out may exceed cap. */
int orbx_debug_distribute_octree(const orb_keypoint* in, int n, int min_x, int max_x, int min_y, int max_y,
                                 int n_features, orb_keypoint* out, int cap, int* n_out);
/* inspection: which keypoint-distribution path produced the last finished run -- 0 device quadtree, 1 device quadtree
 * including the memory-backed pass for levels beyond 4096 candidates, 2 host quadtree (fallback / MORB_HOST_OCTREE=1) */
int orbx_debug_last_path(const orbx_extractor* ex);
/* inspection: cameras whose pyramid level 0 the most recently enqueued run reads in the caller's device buffer instead of a copy
 * (large rigs driven through orbf_*, which promises the buffers' lifetime; 0 everywhere else) */
int orbx_debug_level0_in_place(const orbx_extractor* ex);
/* inspection: how the geometry of the most recent run builds its pyramid -- 0 one tile launch for all levels (k_pyramid_tiled: small
 * rigs), 2 the generic chain (one k_resize launch per level: parameter sets outside both tile forms, MORB_PYR_CHAIN=2), 3 two tile
 * launches with four pixels per lane (k_pyramid_tiled4: large rigs, MORB_PYR_CHAIN=1); -1 before the first run */
int orbx_debug_pyramid_form(const orbx_extractor* ex);
/* tests of the large-rig plan's geometry: tile width / height and split level (defaults 128, 64, 3; 0 = unchanged) of every handle
 * created afterwards.  Process-wide; the product never calls it. */
int orbx_debug_pyramid_plan(int tile_w, int tile_h, int split);

/* ---- matcher ------------------------------------------------------------------------------------------------------------- */
/* {status, nmatches, sweeps, longest candidate list} of the last device-side resolve (inspection only) */
int orbm_debug_last_resolve(const orbm_matcher* m, int* out4);
/* Inspection / bench (roofline M3): `iters` launches of the projection kernel alone, as the frame search
 * launches it (window + level + right-coordinate gates, distances, shortlist), timed with HIP events on the handle's
 * stream.  *avg_us = average launch duration, *n_gated = candidates that passed the gates, summed over the queries. */
int orbm_debug_time_project(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int th_high, int iters,
                            float* avg_us, long long* n_gated);

/* ---- front end: probes of the multi-GPU exchange ------------------------------------------------------------------------- */
/* With on != 0 every step of a handle with an exchange records (HIP events) when its search and when its exchange (all-gather +
 * repack + rig-wide top-2) had finished on the device; orbf_debug_exchange_us returns the last step's two figures in microseconds
 * from the start of the step's matching (the second one is negative when the exchange -- issued with the step's extraction chain
 * -- was over before the matching began). */
int orbf_debug_exchange_timing(orbf_frontend* f, int on);
int orbf_debug_exchange_us(const orbf_frontend* f, float* out2);
/* steps whose blocks were shipped a second time (a block had gone out before its extraction fell back to the host quadtree) */
long orbf_debug_exchange_redos(const orbf_frontend* f);

#ifdef __cplusplus
}
#endif
#endif
