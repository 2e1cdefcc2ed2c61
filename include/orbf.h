/* orbf.h -- one front-end timestep as ONE C call: N-camera extraction, frame assembly and the tracking-path matcher
 * searches, enqueued back to back on one HIP stream with a single host synchronisation at the end.
 *
 * This is the batched form of what the reference does per frame on its tracking thread:
 *   Frame::Frame (reference src/Frame.cc:148-288): ExtractORB + ExtractORB_cam2, the `_total` merge,
 *     ComputeStereoFromRGBD, AssignFeaturesToGrid                                    -> extraction + device frame build
 *   ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, ...) (src/ORBmatcher.cc:3448) -> projection search
 *   exhaustive top-2 between the cameras (inner loop of src/ORBmatcher.cc:287-321)        -> cross-camera top-2
 * It composes include/orbx.h and include/orbm.h (same results as calling them one by one); it exists because at
 * 640x480 every kernel is a few microseconds and host round trips would otherwise dominate the frame.
 */
#ifndef ORBF_H
#define ORBF_H
#include "orbm.h"
#include "orbx.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbf_frontend orbf_frontend;

typedef struct orbf_image {
    const uint8_t* data; /* 8-bit grey; host pointer, or device pointer when on_device != 0; NULL = empty image */
    int32_t width, height, stride;
    int32_t on_device;
    /* Identity of the CONTENT, supplied by the caller: the frame's timestamp or sequence number (anything that differs between
     * two frames that ever share a buffer); 0 = none.  orbf_prefetch serves an extraction that ran ahead only to a step whose
     * images carry the same (pointer, size, stride, generation): with generations a recycled buffer -- host or device --
     * can never be handed a stale extraction.  Without one (0) a host image falls back to a sampled content fingerprint (a
     * heuristic, see orbf_prefetch) and a device image is identified by its pointer alone. */
    uint64_t generation;
} orbf_image;

enum { ORBF_SKIP_CROSS = 1,          /* flags of orbf_step: no cross-camera top-2 in this step */
       ORBF_NO_QUERY_RECORDS = 2 };  /* orbf_step_motion*: orbf_result::queries stays NULL -- the queries of a motion step only exist
                                      * inside the projection kernel; without this flag the host writes the same records for the
                                      * caller (orbm_queries_from_motion, ~3 ns per byte: 6 us for 2000 queries) while it waits */

typedef struct orbf_result { /* all pointers: pinned host memory owned by the handle, valid until the next step */
    int32_t n_cams, n_total;
    const int32_t* counts;            /* [n_cams] keypoints per camera                                          */
    const orb_keypoint* kps;          /* [n_total] mvKeys_total (global, cam-major index)                       */
    const uint8_t* desc;              /* [n_total][32]                                                          */
    const float* uright;              /* [n_total] mvuRight_total                                               */
    const float* depth;               /* [n_total] mvDepth_total                                                */
    int32_t nmatches;                 /* SearchByProjection return value (0 without queries)                    */
    const int32_t* match_of_feature;  /* [n_total] as orbm_search_by_projection                                 */
    const int32_t* cross_best_idx;    /* [n_total] as orbm_cross_top2 (NULL with ORBF_SKIP_CROSS)               */
    const int32_t* cross_best_dist;
    const int32_t* cross_second_dist;
    float gpu_wait_us;                /* host time spent blocked in the final synchronisation                   */
    int32_t n_queries;                /* queries searched this step                                             */
    const orbm_query* queries;        /* [n_queries] host copy of what was searched (NULL with ORBF_NO_QUERY_RECORDS) */
    const float* un_x;                /* [n_total] undistorted positions (mvKeysUn_total[i].pt): the keypoint positions     */
    const float* un_y;                /*   themselves unless a calibration with k1 != 0 is set                            */
    float host_us[4];                 /* host timeline of the call: query preparation, enqueue of the whole step,
                                         blocked in the final synchronisation, bookkeeping after it            */
    int32_t rig_cams;                 /* native exchange active: cameras of the whole rig (world x n_cams), else 0      */
    const int32_t* rig_counts;        /* [rig_cams] keypoints of every camera of the rig, rank-major; the cross_* arrays
                                         then hold this rank's features matched against ALL cameras of the rig:
                                         cross_best_idx indexes the rank-major concatenation of every rank's features */
} orbf_result;

int orbf_create(const orbx_params* params, int n_cams, int max_width, int max_height, int device, orbf_frontend** out);
/* The same with the look-ahead depth chosen by the caller (1..3; 0 = MORB_AHEAD_DEPTH, default 3): a handle creates one
 * extractor instance, i.e. one stream, per timestep it can extract ahead.  Streams are hardware queues and the part serves four of
 * them side by side.  A multi-GPU exchange (orbf_exchange_init) brings no stream of its own by default: the all-gather and the
 * rig-wide top-2 follow the step's search on the matcher's stream (orbf_exchange_placement() == 1).  The other arrangement
 * (MORB_EXCHANGE_PLACEMENT=side, or =auto from three ranks on; == 2) runs them on a side stream, and the handle then keeps two
 * extractor instances -- 50 % slower in the forced-exchange loop of round 3. */
int orbf_create_depth(const orbx_params* params, int n_cams, int max_width, int max_height, int device, int ahead_depth, orbf_frontend** out);
void orbf_destroy(orbf_frontend* f);
/* HBM-resident depth image (metres, float32) of one camera for ComputeStereoFromRGBD; NULL: uRight = -1 */
int orbf_set_depth(orbf_frontend* f, int cam, const float* d_depth, int stride_floats);
/* Camera calibration of the rig (the reference keeps ONE mK / mDistCoef for all cameras).  With k1 != 0 the frame is
 * assembled as the reference does it: positions undistorted (Frame::UndistortKeyPoints), image bounds from the undistorted
 * corners (ComputeImageBounds), uRight from the undistorted x, depth read at the distorted pixel.  NULL / k1 == 0: off
 * (the default).  Drops whatever orbf_prefetch has in flight. */
int orbf_set_calibration(orbf_frontend* f, const orb_calibration* calib);
/* mbf = Camera.bf; th_high / check_orientation as in ORBmatcher (defaults 40, 100, 1) */
int orbf_configure(orbf_frontend* f, float mbf, int th_high, int check_orientation);
/* Overlap of consecutive timesteps.  Announces the images of a FUTURE step (a FIFO: the step after the next orbf_step /
 * orbf_step_motion call, then the ones after that; at most THREE steps ahead -- MORB_AHEAD_DEPTH=1..3 lowers the limit and
 * the number of extractor instances the handle creates).  A step call enqueues the extraction of the announced images right
 * after it has enqueued its own matching, so they run next to each other on the GPU (matching occupies a handful of the 256
 * CUs; with three steps announced three extraction chains run side by side on three extractor instances -- together with
 * the matcher's stream that is four hardware queues, as many as the part's command processor serves side by side), and the
 * following steps find their features ready or in flight.  Those steps must then be called with
 * exactly the announced images, in order (same pointers, sizes, strides); otherwise everything in flight is dropped and
 * the images are extracted again.  The exact contract: a buffer must stay unchanged from its announcement until the step that
 * consumes it has returned (host images are read by the copy engine while the intervening steps run).  A caller that recycles
 * buffers says so with orbf_image::generation (frame timestamp / sequence number): it is part of an image's identity, so a
 * refilled buffer is a different image and is extracted again.  For HOST images without a generation a sampled content
 * fingerprint (32 probes of 64 bytes each, taken when the upload is enqueued and when the step arrives) catches most refills --
 * a heuristic: a refill that leaves those 2 KB unchanged goes unnoticed, and so does any refill of a DEVICE image without a
 * generation (it is identified by its pointer alone).  Results are bit-identical with and without announcements; rigs of
 * more than 4 cameras ignore them. */
int orbf_prefetch(orbf_frontend* f, const orbf_image* next_images);
/* How many timesteps orbf_prefetch accepts ahead of the step being matched on this handle: the depth it was created with
 * (MORB_AHEAD_DEPTH, default 3), lowered to two only when a multi-GPU exchange was set up on the side stream
 * (orbf_exchange_placement() == 2: its collective then occupies one of the four hardware queues). */
int orbf_ahead_depth(const orbf_frontend* f);
/* Multi-GPU exchange: the HBM block holding the LAST step's merged descriptors -- cap_rows rows of 32 bytes in global
 * (camera-major, packed) order followed by a 256-byte trailer of int32 per-camera counts -- ready to be the send buffer
 * of one all-gather (every rank has the same capacity, hence the same block size).  Valid until the step after the next
 * one starts; orbm_cross_top2_gathered consumes the gathered blocks. */
int orbf_export_block(orbf_frontend* f, const uint8_t** d_block, size_t* block_bytes, int* cap_rows);
/* Before orbf_step_begin: is the export block of the step about to be begun with `images` final already (its extraction
 * ran ahead and completed cleanly)?  Then *d_block is that block (the one orbf_export_block will name after the begin) and a
 * multi-GPU caller may start its all-gather even before it begins the step; otherwise *d_block = NULL.  orbf_step_begin
 * with the same images then reports block_ready = 1. */
int orbf_peek_block(orbf_frontend* f, const orbf_image* images, const uint8_t** d_block, size_t* block_bytes, int* cap_rows);
/* queries: the projected last-frame map points (may be NULL / 0 on the first frame) */
/* HBM-resident per-feature arrays of the last COMPLETED step's merged frame (global feature order, cameras back to back):
 * what a keyframe built on the device starts from (orbv_keyframe_from_device).  Valid until the next step begins. */
typedef struct orbf_device_features {
    int32_t n_total, n_cams;
    int32_t counts[8];          /* features per camera                                                  */
    const uint8_t* d_desc;      /* n_total x 32                                                         */
    const float* d_angle;       /* mvKeys_total[i].angle                                                */
    const float* d_un_x;        /* mvKeysUn_total[i].pt                                                 */
    const float* d_un_y;
    const int32_t* d_octave;
    const float* d_uright;      /* mvuRight_total                                                       */
    void* stream;               /* the stream these arrays were last written on                         */
} orbf_device_features;
int orbf_export_features(orbf_frontend* f, orbf_device_features* out);

/* Native multi-GPU exchange (one process per GPU, the cameras of the rig sharded over the ranks).  Rank 0 draws an id
 * (orbf_exchange_unique_id), the caller ships it to every rank by whatever means it has (torch.distributed broadcast,
 * MPI, a file), and EVERY rank calls orbf_exchange_init with it (a collective: RCCL's ncclCommInitRank).  From then on
 * every step ends with exactly one RCCL all-gather of the step's export block, issued from inside the step on the
 * matcher's side stream -- right behind the step's own matching when the block was final at begin, after it otherwise --
 * followed by the cross-camera top-2 of this rank's features against the whole rig (orbm_cross_top2_gathered); the
 * rank-local cross matching is skipped.  All front ends of a communicator must have the same capacities (orbx_params). */
int orbf_exchange_unique_id(uint8_t* out128);
int orbf_exchange_init(orbf_frontend* f, const uint8_t* uid128, int world, int rank);
int orbf_exchange_active(const orbf_frontend* f);   /* world size, 0 = off */
/* Where this handle's exchange runs, decided per handle when the exchange is set up: 0 no exchange; 3 (default) at the tail of
 * the step's extraction chain, on the extractor's stream -- the all-gather, the repack and the rig-wide top-2 of a step are on the
 * device while the steps before it are still being matched, and the end of the step only waits for their event; 1 behind the
 * step's search on the matcher's own stream (rounds 2-4; MORB_EXCHANGE_PLACEMENT = chain | inline).  With placement 3 an
 * announcement (orbf_prefetch) is BINDING: the block of the announced images is shipped with their extraction, so the step
 * must pass exactly those images (ORB_E_ARG otherwise), and orbf_reset is a decision of all ranks at the same step (it
 * skips the step numbers whose blocks have been shipped).  A block whose extraction fell back to the host quadtree after it
 * had been shipped carries a mark; every rank sees it and all ship that step's final blocks once more at the end of the step. */
int orbf_exchange_placement(const orbf_frontend* f);
/* (probes of the exchange for tests and the bench -- orbf_debug_*: include/orb_debug.h) */
/* The same exchange between `world` front ends of ONE process on ONE device, each driven by its own host thread (RCCL does
 * not admit two ranks on one GPU): every member calls this with the same `group` id and its own rank, then steps as a rank
 * would -- a step's all-gather rendezvouses the members' threads and copies the blocks device-to-device behind the producers'
 * events.  For machines with fewer GPUs than ranks (tests, the 1-GPU CI box); every kernel and every ordering decision of
 * the multi-GPU step is the product code unchanged.  All members must keep stepping in lockstep (one collective per step). */
int orbf_exchange_init_loopback(orbf_frontend* f, int group, int world, int rank);
/* The same exchange between PROCESSES as direct one-hop writes (no RCCL): one process per GPU over xGMI peer access, or several
 * processes per GPU on a machine with fewer GPUs than ranks, which RCCL refuses.  Two collective steps, the caller carries the
 * handles between the processes by whatever means it has (torch.distributed over gloo, MPI, a file):
 *   1. every rank: orbf_exchange_peer_export(f, world, rank, handle)   -- allocates this rank's receive arena (the blocks of all ranks
 *      for the eight steps that can be in flight) and writes orbf_exchange_peer_handle_bytes() bytes (a hipIpcMemHandle_t);
 *   2. every rank, once it holds all handles (rank r's at handles + r * handle_bytes): orbf_exchange_peer_open(f, handles).
 * A step's exchange is then one kernel that copies the export block into every rank's arena and stores the step's number into this
 * rank's arrival word there, and one single-wave kernel that waits for every rank's arrival word -- no collective call, no
 * communicator, no ordering between the exchanges of different steps.  Everything downstream (repack, rig-wide top-2, placement,
 * re-shipment of a block whose extraction fell back) is the code of the RCCL path. */
size_t orbf_exchange_peer_handle_bytes(void);
int orbf_exchange_peer_export(orbf_frontend* f, int world, int rank, uint8_t* handle_out);
int orbf_exchange_peer_open(orbf_frontend* f, const uint8_t* handles);
/* Whatever the transport, the end of a step waits for the other ranks at most MORB_EXCHANGE_TIMEOUT_MS (default 15 000): a rank that
 * died or hangs makes orbf_step / orbf_step_end return ORB_E_TIMEOUT (the peer transport names the ranks that did not deliver; on
 * the RCCL path ncclCommGetAsyncError is asked while waiting -> ORB_E_HIP).  The exchange is unusable afterwards:
 * orbf_exchange_shutdown (RCCL communicators are aborted, not destroyed) or orbf_destroy. */
int orbf_exchange_shutdown(orbf_frontend* f);       /* back to rank-local steps (orbf_destroy does it too) */

int orbf_step(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, orbf_result* out);
/* Synthetic-stream driver: like orbf_step, with the queries built natively from the PREVIOUS step's features moved by a
 * constant image-plane motion (the arithmetic of orbm_queries_from_motion; no queries on the first step or after orbf_reset).
 * The projection kernel builds them itself from the previous step's frame, which is still in HBM: nothing is written or copied
 * for them on the step's critical path.  out->n_queries / out->queries expose what was searched (the same records, written by
 * the host with orbm_queries_from_motion while it waits for the step's results). */
typedef struct orbf_motion { float du, dv, th; } orbf_motion;
int orbf_step_motion(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, orbf_result* out);
/* The three calls of a host that drives a stream, as one (a binding then crosses the ABI once per timestep instead of three
 * times): orbf_prefetch(next_images) when next_images != NULL, orbf_step_motion(images, ...), and *n_cross =
 * orbm_count_ratio_accepted over the step's cross-camera distances (th_low, ratio; -1 when the step has none).  Same results,
 * same errors as the calls it stands for. */
int orbf_step_motion_ahead(orbf_frontend* f, const orbf_image* images, const orbf_image* next_images, const orbf_motion* motion,
                           int flags, int th_low, float ratio, orbf_result* out, int* n_cross);
int orbf_reset(orbf_frontend* f);
/* The synthetic-stream loop in ONE call: for t = t0 .. t0 + steps - 1 announce timestep t + ahead (orbf_prefetch; ahead = 0:
 * nothing is announced, every step is an isolated one; at most min(orbf_ahead_depth() + 1, 3), refused with ORB_E_ARG and a
 * message naming the limit before any step runs otherwise), run orbf_step_motion on ring[(t % ring_len) * n_cams ..] and
 * count the cross-camera matches a SearchByBoW-style acceptance keeps (orbm_count_ratio_accepted(best, second, n, th_low,
 * ratio)).  `ring` holds ring_len timesteps of n_cams images each.  *announced_upto (in/out): the youngest timestep announced so
 * far, so that consecutive calls continue one stream (-1 / t0 - 1 at the start).  This is exactly what a host-language loop
 * over orbf_prefetch + orbf_step_motion does -- a benchmark can time the library without its binding's per-call cost, a test
 * can compare the totals with the step-by-step run. */
typedef struct orbf_stream_stats {
    int64_t features, temporal_matches, cross_accepted;   /* sums over the steps of this call */
    uint64_t digest;                                       /* order-sensitive mix of the three per-step counts */
    double seconds;                                        /* wall time of the loop (host clock, including the last step's wait) */
} orbf_stream_stats;
int orbf_run_stream(orbf_frontend* f, const orbf_image* ring, int ring_len, int t0, int steps, int ahead, int* announced_upto,
                    const orbf_motion* motion, int th_low, float ratio, orbf_stream_stats* out);

/* A step in two halves.  _begin enqueues everything and returns at once; _end blocks (one synchronisation) and fills the
 * result.  *block_ready (may be NULL) = 1 when this step's export block (orbf_export_block, valid right after _begin) is
 * already final -- its extraction had completed cleanly before the call, the normal case with steps announced ahead --
 * so that a multi-GPU caller may enqueue its all-gather and orbm_cross_top2_gathered_enqueue between the two halves and
 * have them run next to the step's matching.  With 0 the block only becomes final in _end (exchange afterwards). */
int orbf_step_begin(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, int* block_ready);
int orbf_step_motion_begin(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, int* block_ready);
int orbf_step_end(orbf_frontend* f, orbf_result* out); /* forget the previous step (next orbf_step_motion searches nothing) */

/* the composed handles, e.g. for orbx_bind_output / orbx_stage_times_us / orbm_cross_top2_blocks */
orbx_extractor* orbf_extractor(orbf_frontend* f);
orbm_matcher* orbf_matcher(orbf_frontend* f);

#ifdef __cplusplus
}
#endif
#endif
