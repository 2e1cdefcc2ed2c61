// frame_sink.h -- internal seam between the extractor's describe kernel and the matcher's frame (orbf_step only).
// When a sink is set, the describe kernel of the device-quadtree path writes every keypoint straight into the merged
// frame arrays (global index = keypoints of the earlier cameras + index inside its own camera), together with its stereo
// coordinate and grid cell -- the per-feature half of the frame assembly (reference src/Frame.cc:221-239, :959-986,
// :632-642), so that the frame kernel that follows only has to sort cell indices.
#pragma once
#include <stdint.h>
#include "../../include/orb_types.h"

struct orbx_extractor;

struct FrameSink {
    float *x, *y, *ur, *depth, *ang;   // [capacity] un-distorted position, uRight, depth, angle
    int* oct;                          // [capacity] octave
    orb_keypoint* kps;                 // [capacity] the `_total` records
    uint32_t* desc;                    // [capacity][8] descriptors in global-index order
    int* cell_of;                      // [capacity] grid cell (camera-major), -1 outside the grid
    float *h_ur, *h_depth;             // mapped pinned mirrors of ur / depth (may be NULL)
    float *h_unx, *h_uny;              // mapped pinned mirrors of the undistorted position (may be NULL)
    orb_calibration calib;             // undistortion of the keypoint position (k1 == 0: off, src/Frame.cc:676)
    const float* cam_depth[4];         // HBM depth image per camera (NULL: no stereo coordinate)
    int cam_depth_stride[4];
    float mbf, minX, minY, invW, invH;
};

// nullptr switches the sink off.  Used by the next orbx_run_async that takes the device-quadtree path (at most 4 cameras).
extern "C" int orbx_set_frame_sink(orbx_extractor* ex, const FrameSink* sink);

// completion event of the most recently enqueued asynchronous run (valid while it is in flight)
extern "C" void* orbx_done_event(const orbx_extractor* ex);
extern "C" int orbx_peek_status(const orbx_extractor* ex);  // status of the oldest run in flight (valid once it completed)

// Launches a caller wants at the very end of the extractor's launch chain, on its stream (orbf: the frame grid and the
// camera-pair top-2).  They are issued from inside orbx_run_async, so they become part of the captured chain graph; `tag`
// tells chains with different tails apart (the callback is NOT invoked when a captured chain is replayed).  fn == NULL: off.
typedef int (*orbx_tail_fn)(void* user, void* stream);
extern "C" int orbx_set_chain_tail(orbx_extractor* ex, orbx_tail_fn fn, void* user, int tag);
// on = 0: the next asynchronous runs issue their launch chain as plain stream launches instead of replaying the captured
// graph.  A graph replay costs the host one call instead of ten, but work that FOLLOWS it on the same stream starts ~25 us
// after the graph's last kernel (measured); a step whose matching is waiting right behind its own extraction is better off
// with plain launches (the host stays ahead of the 5-25 us kernels anyway).
extern "C" int orbx_set_chain_graph(orbx_extractor* ex, int on);
// on = 1: asynchronous runs do not record their completion event; the caller does (orbx_record_done), after it has enqueued
// the run's consumer on the same stream -- the event then no longer stands between the chain and that consumer.
extern "C" int orbx_set_defer_done(orbx_extractor* ex, int on);
// 1: page-locked host images handed to orbx_upload are read by the run's ingest kernel directly (no copy per camera)
extern "C" int orbx_set_pinned_ingest(orbx_extractor* ex, int on);
// 1: device images are read in place as pyramid level 0 until the camera's next upload (large rigs; extractor.hip: k_set_l0)
extern "C" int orbx_set_inplace_level0(orbx_extractor* ex, int on);
extern "C" int orbx_record_done(orbx_extractor* ex);
// orbx_finish without the wait on the completion event, for a caller that has already seen the results of GPU work ordered
// behind the oldest run in flight (orbf_step_end after it watched the resolve's result words arrive)
extern "C" int orbx_finish_completed(orbx_extractor* ex);
// the oldest run in flight is waited for and FORGOTTEN: no counts adopted, and a run that left the device quadtree's limits is not
// completed on the host path (orbx_finish would run that on the extractor's stream -- which, behind a dropped step of a front end
// with an exchange, may hold a wait for other ranks; the caller extracts those images again when their step comes)
extern "C" int orbx_discard(orbx_extractor* ex);
// the handle gives up its own stream and enqueues on the caller's from now on (nothing in flight; the caller keeps the stream alive
// for as long as the handle lives): the front end's spare extractor works on the matcher's stream -- no fifth hardware queue
extern "C" int orbx_adopt_stream(orbx_extractor* ex, void* stream);
