// frontend.hip -- orbf_* (include/orbf.h): one front-end timestep as ONE native call -- N-camera extraction, frame assembly
// on the device, the tracking-path projection search and the cross-camera top-2 -- with consecutive timesteps overlapped
// (orbf_prefetch), isolated timesteps kept on one stream, and the multi-GPU descriptor exchange issued from inside the step.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbm.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "matcher_internal.h"
#include <deque>
#include "../../include/orbf.h"
#include "../../include/orb_debug.h"

using namespace morb;


struct orbf_frontend {
    int device = 0, n_cams = 0, max_w = 0, max_h = 0;
    orbx_extractor* ex = nullptr;        // == exs[0]: the extractor isolated steps run on (orbf_extractor)
    static constexpr int NEX = 3;        // extractor handles = timesteps that can be extracted side by side
    static constexpr int SPARE = NEX, NEX_ALL = NEX + 1;   // + the spare one (see x_spare_until), made when it is first needed
    orbx_extractor* exs[NEX_ALL] = {nullptr, nullptr, nullptr, nullptr};  // consecutive overlapped timesteps go round the handles in use
    int n_ex = 2;                        // handles in use = look-ahead depth (MORB_AHEAD_DEPTH, 1..NEX; orbf_prefetch's limit)
    std::vector<orbx_params> params;     // (instances beyond the first are created when a step first needs them: ensure_extractor)
    orbm_matcher* mt = nullptr;
    std::vector<const float*> d_depth;
    std::vector<int> depth_stride;
    std::vector<int32_t> counts, cam_cap;
    float mbf = 40.f;
    int th_high = ORBM_TH_HIGH, check_ori = 1;
    orb_calibration calib = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // k1 == 0: no undistortion (orbf_set_calibration)
    int cap_total = 0;
    // pinned host result buffers.  The per-feature results exist twice: the extraction of the NEXT timestep (orbf_prefetch)
    // fills the other set while the caller still reads this step's.
    static constexpr int NSETS = NEX + 3;  // this step's (held by the caller) + the timesteps in flight + the one being assigned
    struct ResultSet {
        PinnedBuf<orb_keypoint> kps; PinnedBuf<uint8_t> desc; PinnedBuf<float> ur, depth, unx, uny;
        CrossOut cross;            // cross-camera top-2 of the step, computed at the end of its extraction chain
        bool cross_valid = false;  // ... when the chain was enqueued with cross matching on
    } rs[NSETS];
    int last_flags = 0;  // flags of the most recent step: what announced steps are assumed to want
    int cur = 0;       // set holding the results of the last completed step
    int last_set = 0;  // set most recently handed to an extraction (sets are handed out round robin)
    morb::StageBuf h_queries;   // this step's queries: written by the host, read once by k_project
    // The motion stream (orbf_step_motion) builds its queries inside the projection kernel from the previous frame's arrays in
    // HBM (MotionSrc); the host keeps the same records for orbf_result::queries and the host fallbacks, written while it waits
    std::vector<orbm_query> q_host;
    bool motion_on_device = true;      // MORB_MOTION_ON_DEVICE=0: the host writes the records into the staging buffer first (rounds 1-3)
    PinnedBuf<int32_t> h_match;
    // Small rigs (<= 4 cameras): one persistent frame per result set, filled by the extractor's describe kernel (FrameSink)
    orbm_frame* pframe[NSETS] = {};
    int pframe_W[NSETS] = {}, pframe_H[NSETS] = {};
    // extractions in flight for the NEXT steps (enqueued by earlier orbf_step calls after orbf_prefetch), oldest first
    // fp: content fingerprints of the HOST images taken when their upload was enqueued (see image_fingerprint)
    // done: the run's completion event was seen signalled (asked while an earlier step waited for its results: off the critical path)
    struct InFlight { std::vector<orbf_image> images; std::vector<uint64_t> fp; int set = 0, W = 0, H = 0, e = 0; bool done = false; };
    std::deque<InFlight> inflight;
    std::deque<std::vector<orbf_image>> announced;  // declared by orbf_prefetch, not enqueued yet (at most 2)
    int last_e = 0;  // extractor most recently handed a timestep
    bool poll_ok = true;     // MORB_POLL=0: orbf_step_end always waits with hipStreamSynchronize
    // ---- native multi-GPU exchange (orbf_exchange_init): one all-gather of the step's export block + the rig-wide top-2 per step.
    // A step's exchange needs nothing but the descriptors its extraction produced, so it is issued at the TAIL OF THE STEP'S
    // EXTRACTION CHAIN, on that chain's stream, steps ahead of the step's matching (placement 3; placement 1 = behind the step's
    // search on the matcher's stream, rounds 2-4).  Every step in flight has buffers of its own (XSlot, by step number); the
    // collectives go round NXC communicators by step number -- the same sequence per communicator on every rank whatever stream a
    // rank issues from -- and a block that has to be shipped a second time (its extraction fell back to the host path after it had
    // gone out: ORBM_BLOCK_REDO) goes over a communicator of its own at the end of its step, on every rank.
    static constexpr int NX = 8, NXC = 3;
    struct XSlot {
        DevBuf<uint8_t> recv, list;      // the gathered blocks of all ranks; their rows in use as one list
        DevBuf<int32_t> gstart;          // camera starts + {features, first query, queries} of that list
        PinnedBuf<int32_t> gcnt;         // per-camera counts, own queries, clamped counts, blocks marked "redo"
        CrossOut out;                    // rig-wide top-2 of this rank's features
        hipEvent_t done = nullptr, t_done = nullptr;   // behind the top-2 (t_done: with timing, orbf_debug_exchange_timing)
        long seq = -1;                   // the step this slot's exchange belongs to
        long job = 0;                    // the issuer's job that records `done` (0: the calls were made by the stepping thread)
        std::vector<orbf_image> images; std::vector<uint64_t> fp;   // what the shipped block was extracted from
    } xs[NX];
    void* xcomm = nullptr; int xworld = 0, xrank = 0;   // xcomm == xc[0]: non-null while an exchange is set up
    void* xc[NXC + 1] = {};                              // communicators (RCCL) / loopback groups; the last one ships redone blocks
    int x_placement = 0;                                // 0 no exchange, 1 behind the step's search, 3 at the tail of the step's extraction chain
    long step_seq = 0, x_next = 0;                      // steps begun so far; first step whose exchange has not been issued
    long x_redos = 0;                                   // steps whose blocks were shipped a second time (orbf_debug_exchange_redos)
    // Placement 3 puts a step's exchange -- which ENDS IN A WAIT FOR THE OTHER RANKS -- on the stream of the step's extraction chain.
    // Nothing this rank may still owe the others may queue behind a wait for a LATER step: when steps in flight are dropped after their
    // exchanges were issued (the extraction of an earlier step fell back to the host path), their waits stay on the chains' streams
    // until the other ranks get that far -- and those may be waiting for this rank's re-shipment of the earlier step, whose
    // re-extraction would sit behind exactly these waits (found with rank processes of unequal look-ahead, round 6: every rank sat out
    // the timeout).  So from the drop up to the last step whose exchange is out (x_spare_until) every extraction of this handle runs
    // isolated on a SPARE extractor, whose stream never carries an exchange; by then every dropped wait has been answered.
    // The spare extractor works on the MATCHER's stream (which only ever holds the current step's exchange): a fifth stream would
    // share a hardware queue with one of the chains, possibly behind such a wait.  (Exchange streams of their own would be the general
    // answer: measured, forced exchange on one GPU fell from 23.7 K to 9.7-15.3 K steps/s with one to three more streams next to the
    // three chains and the matcher, an event or a word in HBM as the dependency.)
    long x_spare_until = -1;
    // The ISSUER (placement 3): a host thread of the handle's own that makes the exchange's calls -- the collective (RCCL's enqueue alone
    // is ~20 us of host time), the repack and top-2 launches, the event records.  The stepping thread only queues a job: with the calls
    // on its own path a step's host time exceeded its device time (forced exchange on one GPU, round 5: 56 us per step inside the
    // library against 40 without an exchange).  Rules: every exchange call of the handle goes through the issuer (one thread per
    // communicator at a time, every communicator's sequence in step order); the stepping thread enqueues nothing on a stream for
    // which a job is still queued (x_wait_stream) and waits for a step's job before it waits for the step's event (x_wait_job).
    struct XJob { long id, seq; const uint8_t* send; int rows; hipStream_t st; bool redo; };
    std::thread x_thread;
    std::mutex x_mu;
    std::condition_variable x_cv;
    std::deque<XJob> x_jobs;
    long x_pushed = 0, x_issued = 0;   // jobs queued / jobs whose calls have been made
    int x_rc = 0; std::string x_err;   // first failure of the issuer (sticky: reported by whoever waits next)
    bool x_quit = false, x_async = false;
    long set_xseq[NEX + 3] = {-1, -1, -1, -1, -1, -1};  // per result set: the step whose exchange reads that set's frame as its send buffer
    bool x_timing = false; hipEvent_t ev_x[2] = {nullptr, nullptr}; float x_us[2] = {0.f, 0.f};   // orbf_debug_exchange_timing
    bool xloop = false;                                  // ... over the in-process loopback transport (orbf_exchange_init_loopback)
    morb::PeerComm* xpeer = nullptr;                     // ... as direct writes into the peers' arenas (orbf_exchange_peer_*: processes, IPC)
    bool x_one_comm = false;                             // RCCL: no second communicator could be made -- placement 1 only, which needs one
    bool x_broken = false;                               // a peer did not deliver / RCCL reported an asynchronous error: the exchange is unusable
    long x_timeout_ms = 15000;                           // how long the end of a step waits for the other ranks (MORB_EXCHANGE_TIMEOUT_MS)
    bool overlap_ok = true;  // cleared when an overlapped extraction had to be redone on the host path ...
    int clean_steps = 0;     // ... and set again after a few steps that stayed on the device path
    struct Pending {  // a timestep between orbf_step_begin and orbf_step_end
        bool active = false, async_path = false, fr_persistent = false, block_ready = false, cross_from_set = false, forked = false;
        long seq = 0;              // number of the step (orbf_frontend::step_seq when it began)
        bool seq_taken = false;    // ... taken from step_seq: a begin that fails hands it back (the step can be retried)
        bool inline_match = false; // the step's own extraction was enqueued by this call: its matching follows on the SAME stream
        bool mirror_requested = false, mirror_pending = false;  // the pinned result mirrors are filled by a copy kernel of the step
        int set = 0, e = 0, W = 0, H = 0, nq = 0, flags = 0, n = 0;
        bool ext_done = false;     // the step's extraction + frame grid had completed when the step began (no event wait needed)
        bool use_ms = false; MotionSrc ms; orbf_motion motion;   // queries built by the projection kernel (motion stream)
        bool from_motion = false;  // orbf_step_motion* (either way of building the queries)
        orbm_frame* fr = nullptr;
        SearchJob J{nullptr, nullptr, 0, nullptr, false, 0.f, 0, 0, 64, false};
        std::vector<orbf_image> images;
        std::vector<orbm_cam_features> cams;
        std::chrono::steady_clock::time_point t_impl, t_enqueued;
    } pending;
    hipEvent_t ev_extracted = nullptr;  // extractor stream -> matcher stream on the synchronous path
    hipEvent_t ev_ready[NSETS] = {};  // extraction + frame grid of the step using that set
    // Larger rigs: the camera-pair top-2 FOLLOWS that event on the extractor's stream (ev_cross behind it): an isolated step's search
    // -- which needs the frame, not the top-2 -- then runs NEXT TO the top-2 instead of behind it (8 x 1080p: 129 us of top-2 in front
    // of 60 + 100 us of projection + resolve; the resolve is eight workgroups).  The end of the step waits for ev_cross.  Small rigs keep
    // the top-2 inside the captured chain (or in the projection's launch: SideJob).
    hipEvent_t ev_cross[NSETS] = {};
    bool cross_pending[NSETS] = {};
    // frame of the last completed step (orbf_export_block); a frame built on the synchronous path is kept until the next step
    orbm_frame* last_frame = nullptr; bool last_frame_owned = false;
    // previous step (for orbf_step_motion)
    int prev_n = 0;
    std::vector<int32_t> prev_cam_of;
    std::vector<float> scale_factors;
    std::chrono::steady_clock::time_point t_entry;
};

static int getenv_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static bool small_rig(const orbf_frontend* f) { return f->n_cams <= 4 && f->cap_total <= 8192 && !f->mt->host_resolve; }


int orbf_create(const orbx_params* params, int n_cams, int max_width, int max_height, int device, orbf_frontend** out) {
    return orbf_create_depth(params, n_cams, max_width, max_height, device, 0, out);
}

int orbf_create_depth(const orbx_params* params, int n_cams, int max_width, int max_height, int device, int ahead_depth, orbf_frontend** out) {
    MORB_ARG(params && out && n_cams >= 1 && n_cams <= 64 && ahead_depth >= 0 && ahead_depth <= orbf_frontend::NEX);
    orbf_frontend* f = new orbf_frontend();
    f->device = device; f->n_cams = n_cams; f->max_w = max_width; f->max_h = max_height;
    int rc = orbx_create(params, n_cams, max_width, max_height, device, &f->exs[0]);
    f->ex = f->exs[0];
    // (device images of a step stay valid and unchanged until the step has returned -- this interface's own contract -- and every
    // step hands every camera an image: large rigs read pyramid level 0 in the caller's buffers instead of copying it)
    if (!rc) rc = orbx_set_inplace_level0(f->exs[0], 1);
    if (!rc) rc = orbm_create(device, &f->mt);
    // two streams: the matcher's own one follows the extractor's through events, so that the next step's extraction can
    // run next to this step's matching
    if (rc) { orbf_destroy(f); return rc; }
    f->d_depth.assign(n_cams, nullptr); f->depth_stride.assign(n_cams, 0); f->counts.assign(n_cams, 0);
    { const char* pe = getenv("MORB_POLL"); f->poll_ok = !(pe && atoi(pe) == 0); }
    f->motion_on_device = getenv_int("MORB_MOTION_ON_DEVICE", 1) != 0;
    f->x_timeout_ms = std::max(1L, (long)getenv_int("MORB_EXCHANGE_TIMEOUT_MS", 15000));
    f->scale_factors.assign(params[0].nlevels, 1.f);
    if ((rc = orbx_tables(&params[0], f->scale_factors.data(), nullptr, nullptr, nullptr, nullptr, nullptr))) { orbf_destroy(f); return rc; }
    for (int c = 0; c < n_cams; ++c) { f->cam_cap.push_back(params[c].nfeatures + 4 * params[c].nlevels); f->cap_total += f->cam_cap.back(); }
    const size_t cap = (size_t)f->cap_total;
    if (hipEventCreateWithFlags(&f->ev_extracted, hipEventDisableTiming) != hipSuccess) { morb::set_error("hipEventCreate failed"); orbf_destroy(f); return ORB_E_HIP; }
    for (int k = 0; k < orbf_frontend::NSETS; ++k)
        if (hipEventCreateWithFlags(&f->ev_ready[k], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess ||
            hipEventCreateWithFlags(&f->ev_cross[k], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { morb::set_error("hipEventCreate failed"); orbf_destroy(f); return ORB_E_HIP; }
    // Every stream is a hardware queue and the command processor serves four side by side.  The streams
    // that work next to each other are created HERE, together and first -- extractor 0, the matcher's, the overlap partners --
    // because which queues end up sharing a pipe follows the order of creation (measured: the same four streams created
    // lazily, behind a caller's own streams, ran the loop at 87-195 us per step instead of 50).  A multi-GPU exchange runs on the
    // matcher's stream (exchange_queues), so a handle keeps its depth of three with it.
    { const int d = ahead_depth > 0 ? ahead_depth : getenv_int("MORB_AHEAD_DEPTH", 3); f->n_ex = d < 1 ? 1 : (d > orbf_frontend::NEX ? orbf_frontend::NEX : d); }
    f->params.assign(params, params + n_cams);
    for (int e = 1; e < f->n_ex && !rc; ++e) {   // overlap partners
        rc = orbx_create(params, n_cams, max_width, max_height, device, &f->exs[e]);
        if (!rc) rc = orbx_set_inplace_level0(f->exs[e], 1);
    }
    for (int k = 0; k < orbf_frontend::NSETS && !rc; ++k)
        if ((rc = f->rs[k].kps.reserve(cap)) || (rc = f->rs[k].desc.reserve(cap * 32)) || (rc = f->rs[k].ur.reserve(cap)) ||
            (rc = f->rs[k].depth.reserve(cap)) || (rc = f->rs[k].unx.reserve(cap)) || (rc = f->rs[k].uny.reserve(cap))) break;
    if (!rc) rc = f->h_match.reserve(cap);
    if (rc) { orbf_destroy(f); return rc; }
    *out = f;
    return ORB_OK;
}

static void x_stop(orbf_frontend* f);

void orbf_destroy(orbf_frontend* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->xcomm) (void)orbf_exchange_shutdown(f);
    x_stop(f);   // (whatever the shutdown returned: the issuer thread never outlives the handle)
    if (f->xpeer) { morb::peer_close(f->xpeer); f->xpeer = nullptr; }
    for (int i = 0; i < 2; ++i) if (f->ev_x[i]) (void)hipEventDestroy(f->ev_x[i]);
    for (auto& X : f->xs) {
        X.recv.release(); X.list.release(); X.gstart.release(); X.gcnt.release(); X.out.release();
        if (X.done) (void)hipEventDestroy(X.done);
        if (X.t_done) (void)hipEventDestroy(X.t_done);
    }
    for (int e = 0; e < orbf_frontend::NEX_ALL; ++e) if (f->exs[e]) (void)hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[e]));
    if (f->mt) (void)hipStreamSynchronize(f->mt->stream);
    if (f->last_frame && f->last_frame_owned) orbm_frame_destroy(f->last_frame);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) if (f->pframe[k]) orbm_frame_destroy(f->pframe[k]);  // back to the matcher's pool first
    if (f->exs[orbf_frontend::SPARE]) { orbx_destroy(f->exs[orbf_frontend::SPARE]); f->exs[orbf_frontend::SPARE] = nullptr; }   // (it works on the matcher's stream)
    if (f->mt) orbm_destroy(f->mt);
    for (int e = 0; e < orbf_frontend::NEX_ALL; ++e) if (f->exs[e]) orbx_destroy(f->exs[e]);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) { f->rs[k].kps.release(); f->rs[k].desc.release(); f->rs[k].ur.release(); f->rs[k].depth.release(); f->rs[k].unx.release(); f->rs[k].uny.release(); f->rs[k].cross.release(); }
    f->h_queries.release(); f->h_match.release();
    if (f->ev_extracted) (void)hipEventDestroy(f->ev_extracted);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) if (f->ev_ready[k]) (void)hipEventDestroy(f->ev_ready[k]);
    for (int k = 0; k < orbf_frontend::NSETS; ++k) if (f->ev_cross[k]) (void)hipEventDestroy(f->ev_cross[k]);
    delete f;
}

orbx_extractor* orbf_extractor(orbf_frontend* f) { return f ? f->ex : nullptr; }
orbm_matcher* orbf_matcher(orbf_frontend* f) { return f ? f->mt : nullptr; }

int orbf_set_depth(orbf_frontend* f, int cam, const float* d_depth, int stride_floats) {
    MORB_ARG(f && cam >= 0 && cam < f->n_cams);
    f->d_depth[cam] = d_depth; f->depth_stride[cam] = stride_floats;
    return ORB_OK;
}

static int orbf_drain(orbf_frontend* f);

int orbf_set_calibration(orbf_frontend* f, const orb_calibration* calib) {
    MORB_ARG(f != nullptr);
    int rc = orbf_drain(f);  // (prefetched extractions carry the old calibration in their frame sinks)
    if (rc) return rc;
    f->announced.clear();
    if (calib) f->calib = *calib; else memset(&f->calib, 0, sizeof(f->calib));
    return orbm_set_calibration(f->mt, calib);
}

int orbf_configure(orbf_frontend* f, float mbf, int th_high, int check_orientation) {
    MORB_ARG(f && th_high >= 0 && th_high <= 256);
    f->mbf = mbf; f->th_high = th_high; f->check_ori = check_orientation ? 1 : 0;
    return ORB_OK;
}

static int orbf_step_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                          orbf_result* out, bool queries_in_pinned, const orbf_motion* motion = nullptr);
static int orbf_step_begin_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                                bool queries_in_pinned, int* block_ready, const orbf_motion* motion = nullptr);
static int orbf_step_end_impl(orbf_frontend* f, orbf_result* out);
static int orbf_drain(orbf_frontend* f);

int orbf_step(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, orbf_result* out) {
    MORB_ARG(f && images && out && nq >= 0 && (nq == 0 || queries));
    f->t_entry = std::chrono::steady_clock::now();
    return orbf_step_impl(f, images, queries, nq, flags, out, false);
}

static int orbf_drain(orbf_frontend* f);

int orbf_reset(orbf_frontend* f) {
    MORB_ARG(f != nullptr);
    f->prev_n = 0; f->announced.clear(); f->overlap_ok = true;
    if (f->pending.active) {  // a begun step is abandoned with everything else in flight
        if (f->pending.fr && !f->pending.fr_persistent) { (void)hipStreamSynchronize(f->mt->stream); orbm_frame_destroy(f->pending.fr); }
        f->pending.active = false; f->pending.fr = nullptr;
    }
    const int rc = orbf_drain(f);
    // With an exchange, blocks of steps that had been announced went out with their extraction chains: those step numbers are spent
    // (the peers hold what the abandoned images gave).  The next step takes the first number nothing has been shipped for -- the
    // same on every rank as long as the ranks reset at the same step with the same announcements, which a reset next to an exchange
    // has to be anyway (every rank's collectives go round the communicators by step number).
    if (f->xcomm && f->x_next > f->step_seq) f->step_seq = f->x_next;
    return rc;
}

int orbf_export_block(orbf_frontend* f, const uint8_t** d_block, size_t* block_bytes, int* cap_rows) {
    MORB_ARG(f && d_block && block_bytes && cap_rows);
    // between orbf_step_begin and orbf_step_end: the frame of the step in flight; otherwise the last completed step's
    const orbm_frame* F = (f->pending.active && f->pending.fr) ? f->pending.fr : f->last_frame;
    if (!F) { morb::set_error("no step to export"); return ORB_E_ARG; }
    *d_block = F->b->d_desc.p;
    *cap_rows = F->desc_rows;
    *block_bytes = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    return ORB_OK;
}

int orbf_exchange_unique_id(uint8_t* out128) {
    MORB_ARG(out128 != nullptr);
    return exchange_unique_id(out128);
}

static int orbf_drain(orbf_frontend* f);
// Where a handle's exchange runs -- decided when the exchange is set up, per handle:
//   3  (default) at the tail of the step's extraction chain, on the extractor's stream: the all-gather, the repack and the rig-wide
//      top-2 of step t are on the device while steps t-1, t-2 are still being matched; the end of step t only waits for their event;
//   1  behind the step's search on the matcher's own stream (rounds 2-4; what a step whose extraction ran synchronously gets anyway).
// MORB_EXCHANGE_PLACEMENT = chain | inline chooses.  (Rounds 2-4 also had the collective on the matcher's side stream -- a fork and
// a join per step, one extraction chain fewer; never ahead of `inline` in any measurement, removed in round 5.)
static void x_issuer_main(orbf_frontend* f);
static void x_stop(orbf_frontend* f);
static int exchange_queues(orbf_frontend* f) {
    int placement = 3;
    if (const char* e = getenv("MORB_EXCHANGE_PLACEMENT")) { if (!strcmp(e, "inline")) placement = 1; }
    if (f->x_one_comm) placement = 1;   // (one communicator: a step's one collective at the end of the step, in step order, never a re-shipment)
    f->x_broken = false;
    f->x_placement = placement;
    f->mt->side_inline = true;   // (no side stream next to an exchange: the handle keeps its extraction chains + the matcher = four queues)
    f->x_next = f->step_seq;
    for (auto& X : f->xs) {
        X.seq = -1; X.job = 0;
        if (!X.done) MORB_HIP(hipEventCreateWithFlags(&X.done, hipEventDisableTiming));
    }
    // the issuer thread (placement 1: the stepping thread makes the calls itself)
    f->x_rc = 0; f->x_err.clear(); f->x_pushed = f->x_issued = 0;
    f->x_async = placement == 3;
    if (f->x_async) f->x_thread = std::thread(x_issuer_main, f);
    return ORB_OK;
}

int orbf_exchange_placement(const orbf_frontend* f) { return f ? (f->xcomm ? f->x_placement : 0) : ORB_E_ARG; }

// Test / bench probe: with `on`, every step of a handle with an exchange records when its search and when its exchange had
// finished on the device (HIP events on the matcher's stream; adds two event records to the step).
int orbf_debug_exchange_timing(orbf_frontend* f, int on) {
    MORB_ARG(f != nullptr);
    MORB_HIP(hipSetDevice(f->device));
    if (on) for (int i = 0; i < 2; ++i) if (!f->ev_x[i]) MORB_HIP(hipEventCreate(&f->ev_x[i]));
    if (on) for (auto& X : f->xs) if (!X.t_done) MORB_HIP(hipEventCreate(&X.t_done));
    f->x_timing = on != 0;
    return ORB_OK;
}
long orbf_debug_exchange_redos(const orbf_frontend* f) { return f ? f->x_redos : -1; }
int orbf_debug_exchange_us(const orbf_frontend* f, float* out2) {
    MORB_ARG(f && out2);
    out2[0] = f->x_us[0]; out2[1] = f->x_us[1];
    return ORB_OK;
}

int orbf_exchange_init(orbf_frontend* f, const uint8_t* uid128, int world, int rank) {
    MORB_ARG(f && uid128 && world >= 1 && rank >= 0 && rank < world && world * f->n_cams <= 512 && !f->xcomm);
    MORB_HIP(hipSetDevice(f->device));
    void* comm = nullptr;
    int rc = exchange_comm_init(&comm, world, uid128, rank);
    if (rc) return rc;
    f->xc[0] = comm;
    // Placement 3 needs NXC + 1 INDEPENDENT communicators (steps in flight go round NXC of them, re-shipped blocks take the last).  Every
    // rank tries to make them, in the same order; whether ALL ranks got ALL of them is settled with one all-gather of a flag over the
    // first communicator.  If not, every rank drops its clones and the handle runs placement 1 over the one communicator -- one
    // collective per step, at the end of the step, in step order: nothing there can pair up wrongly.
    int mine = 1;
    for (int k = 1; k <= orbf_frontend::NXC; ++k) {
        (void)exchange_comm_clone(comm, world, rank, &f->xc[k]);   // (collective)
        if (!f->xc[k]) { mine = 0; break; }                        // (the first failure is at the same k on every rank, or detected below)
    }
    int all = mine;
    {
        uint8_t* d = nullptr;
        bool ok = hipMalloc((void**)&d, (size_t)world + 1) == hipSuccess;
        const uint8_t v = (uint8_t)mine;
        ok = ok && hipMemcpy(d, &v, 1, hipMemcpyHostToDevice) == hipSuccess;
        hipStream_t ts = nullptr;   // (a stream of its own, not the null stream)
        ok = ok && hipStreamCreateWithFlags(&ts, hipStreamNonBlocking) == hipSuccess;
        ok = ok && exchange_allgather(comm, d, d + 1, 1, ts) == ORB_OK && hipStreamSynchronize(ts) == hipSuccess;
        if (ts) (void)hipStreamDestroy(ts);
        std::vector<uint8_t> flags(world, 0);
        ok = ok && hipMemcpy(flags.data(), d + 1, world, hipMemcpyDeviceToHost) == hipSuccess;
        if (d) (void)hipFree(d);
        if (!ok) { (void)hipGetLastError(); exchange_comm_destroy(comm); f->xc[0] = nullptr; morb::set_error("multi-GPU exchange: the ranks could not agree on their communicators"); return ORB_E_HIP; }
        for (int r = 0; r < world; ++r) all = all && flags[r];
    }
    f->x_one_comm = !all;
    if (!all) for (int k = 1; k <= orbf_frontend::NXC; ++k) { if (f->xc[k]) exchange_comm_destroy(f->xc[k]); f->xc[k] = nullptr; }
    f->xcomm = comm; f->xworld = world; f->xrank = rank;
    return exchange_queues(f);
}

// ---- the peer transport (exchange.hip): processes that write their blocks straight into each other's arenas
size_t orbf_exchange_peer_handle_bytes(void) { return morb::peer_handle_bytes(); }

int orbf_exchange_peer_export(orbf_frontend* f, int world, int rank, uint8_t* handle_out) {
    MORB_ARG(f && handle_out && world >= 1 && world <= 64 && rank >= 0 && rank < world && world * f->n_cams <= 512 && !f->xcomm && !f->xpeer);
    MORB_HIP(hipSetDevice(f->device));
    const size_t block = (size_t)f->cap_total * 32 + ORBM_BLOCK_TRAILER;
    return morb::peer_export(&f->xpeer, world, rank, block, orbf_frontend::NX, f->x_timeout_ms, handle_out);
}

int orbf_exchange_peer_open(orbf_frontend* f, const uint8_t* handles) {
    MORB_ARG(f && handles && f->xpeer && !f->xcomm);
    MORB_HIP(hipSetDevice(f->device));
    int rc = morb::peer_open(f->xpeer, handles);
    if (rc) { morb::peer_close(f->xpeer); f->xpeer = nullptr; return rc; }
    f->xcomm = f->xpeer; f->xworld = morb::peer_world(f->xpeer); f->xrank = morb::peer_rank(f->xpeer);
    f->x_one_comm = false;
    return exchange_queues(f);
}

int orbf_exchange_active(const orbf_frontend* f) { return f && f->xcomm ? f->xworld : 0; }

int orbf_exchange_init_loopback(orbf_frontend* f, int group, int world, int rank) {
    MORB_ARG(f && world >= 1 && rank >= 0 && rank < world && world * f->n_cams <= 512 && !f->xcomm);
    MORB_HIP(hipSetDevice(f->device));
    int rc;
    for (int k = 0; k <= orbf_frontend::NXC; ++k) {   // one rendezvous group per communicator slot
        LoopComm* C = nullptr;
        if ((rc = loop_join(group * (orbf_frontend::NXC + 1) + k, world, rank, &C))) {
            for (int j = 0; j < k; ++j) { loop_leave(static_cast<LoopComm*>(f->xc[j])); f->xc[j] = nullptr; }
            return rc;
        }
        f->xc[k] = C;
    }
    f->xcomm = f->xc[0]; f->xworld = world; f->xrank = rank; f->xloop = true;
    return exchange_queues(f);
}

int orbf_exchange_shutdown(orbf_frontend* f) {
    MORB_ARG(f != nullptr);
    x_stop(f);   // (needs no device: everything queued is issued first, the thread is joined whatever follows)
    if (!f->xcomm) {
        if (f->xpeer) { (void)hipSetDevice(f->device); morb::peer_close(f->xpeer); f->xpeer = nullptr; }   // (exported, never opened)
        return ORB_OK;
    }
    MORB_HIP(hipSetDevice(f->device));
    if (f->x_broken && !f->xloop && !f->xpeer)   // (a collective that will never complete sits on a stream: abort lets it end)
        for (int k = orbf_frontend::NXC; k >= 0; --k) if (f->xc[k]) { exchange_comm_abort(f->xc[k]); f->xc[k] = nullptr; }
    for (int e = 0; e < orbf_frontend::NEX_ALL; ++e) if (f->exs[e]) (void)hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[e]));   // (exchanges run on the chains' streams)
    if (f->mt) { if (f->mt->side_stream) (void)hipStreamSynchronize(f->mt->side_stream); (void)hipStreamSynchronize(f->mt->stream); }
    if (f->xpeer) { morb::peer_close(f->xpeer); f->xpeer = nullptr; }
    else for (int k = orbf_frontend::NXC; k >= 0; --k) {
        if (!f->xc[k]) continue;
        if (f->xloop) loop_leave(static_cast<LoopComm*>(f->xc[k]));
        else exchange_comm_destroy(f->xc[k]);
        f->xc[k] = nullptr;
    }
    f->xcomm = nullptr; f->xworld = 0; f->xrank = 0; f->xloop = false; f->x_placement = 0; f->x_one_comm = false; f->x_broken = false;
    if (f->mt) f->mt->side_inline = false;
    return ORB_OK;
}

static std::vector<uint64_t> image_fingerprints(const orbf_image* images, int n);

// The calls of one exchange: all-gather of a frame's export block + repack + rig-wide top-2 into the step's slot, on `st` behind whatever
// produced the block there.  Made by the issuer thread (placement 3) or by the stepping thread itself (placement 1).
static int exchange_calls(orbf_frontend* f, const orbf_frontend::XJob& J) {
    orbf_frontend::XSlot& X = f->xs[J.seq % orbf_frontend::NX];
    const size_t block = (size_t)J.rows * 32 + ORBM_BLOCK_TRAILER;
    int rc;
    const uint8_t* recv = X.recv.p;
    if (f->xpeer) {
        const int slot = (int)(J.seq % orbf_frontend::NX);
        rc = morb::peer_allgather(f->xpeer, slot, J.redo ? 1 : 0, (unsigned)(J.seq + 1), J.send, J.st);
        recv = morb::peer_recv(f->xpeer, slot, J.redo ? 1 : 0);
    } else {
        // (one communicator -- placement 1 -- takes every collective, in step order; otherwise steps go round NXC, re-shipments take the last)
        void* comm = f->x_one_comm ? f->xc[0] : f->xc[J.redo ? orbf_frontend::NXC : (int)(J.seq % orbf_frontend::NXC)];
        rc = f->xloop ? loop_allgather(static_cast<LoopComm*>(comm), J.send, X.recv.p, block, J.st)
                      : exchange_allgather(comm, J.send, X.recv.p, block, J.st);
    }
    if (rc) return rc;
    if ((rc = gathered_enqueue_to(J.st, recv, f->xworld, block, J.rows, f->n_cams, f->xrank, X.list.p, X.gstart.p, X.gcnt.dp, X.out))) return rc;
    if (f->x_timing && X.t_done) (void)hipEventRecord(X.t_done, J.st);
    MORB_HIP(hipEventRecord(X.done, J.st));
    return ORB_OK;
}

static void x_issuer_main(orbf_frontend* f) {
    (void)hipSetDevice(f->device);
    std::unique_lock<std::mutex> lk(f->x_mu);
    for (;;) {
        f->x_cv.wait(lk, [&] { return f->x_quit || !f->x_jobs.empty(); });
        if (f->x_jobs.empty()) return;   // (quit with nothing left to do)
        const orbf_frontend::XJob J = f->x_jobs.front();
        lk.unlock();
        const int rc = f->x_rc ? f->x_rc : exchange_calls(f, J);   // (after a failure the remaining jobs are dropped: their waiters get the first error)
        lk.lock();
        if (rc && !f->x_rc) { f->x_rc = rc; f->x_err = orb_last_error(); }
        f->x_jobs.pop_front();
        f->x_issued = J.id;
        f->x_cv.notify_all();
    }
}

static int x_failed(orbf_frontend* f) {   // (x_mu held) the issuer's first failure, in the calling thread's error text
    if (f->x_rc) morb::set_error("%s", f->x_err.c_str());
    return f->x_rc;
}
// the job `id` has been issued (its event recorded); 0: nothing to wait for
static int x_wait_job(orbf_frontend* f, long id) {
    if (!f->x_async || id <= 0) return ORB_OK;
    std::unique_lock<std::mutex> lk(f->x_mu);
    f->x_cv.wait(lk, [&] { return f->x_issued >= id; });
    return x_failed(f);
}
// no job for stream `st` is queued any more: the stepping thread may enqueue on it (st == nullptr: no job at all)
static int x_wait_stream(orbf_frontend* f, hipStream_t st) {
    if (!f->x_async) return ORB_OK;
    std::unique_lock<std::mutex> lk(f->x_mu);
    f->x_cv.wait(lk, [&] { for (const auto& J : f->x_jobs) if (!st || J.st == st) return false; return true; });
    return x_failed(f);
}
static void x_stop(orbf_frontend* f) {
    if (!f->x_thread.joinable()) return;
    { std::lock_guard<std::mutex> lk(f->x_mu); f->x_quit = true; f->x_cv.notify_all(); }   // (the issuer finishes what is queued first)
    f->x_thread.join();
    f->x_quit = false; f->x_async = false;
}

// The end of a step waits for its exchange -- for OTHER RANKS -- and a rank that died or hangs must become an error of this call, not
// a hang of every rank: the event is polled for at most the exchange's timeout (+ a margin: the peer transport's
// own wait kernel gives up after exactly the timeout and then says which ranks were missing), RCCL's asynchronous error state is
// asked while waiting.  After a failure the exchange stays unusable (x_broken): orbf_exchange_shutdown / orbf_destroy clean up.
static int x_wait_done(orbf_frontend* f, orbf_frontend::XSlot& X, long seq, bool redo) {
    if (f->x_broken) { morb::set_error("multi-GPU exchange: unusable after an earlier failure (shut it down and set it up again)"); return ORB_E_TIMEOUT; }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit_ms = (double)f->x_timeout_ms + (f->xpeer ? 5000.0 : 0.0);
    long polls = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(X.done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { f->x_broken = true; morb::set_error("multi-GPU exchange: %s", hipGetErrorString(q)); return ORB_E_HIP; }
        (void)hipGetLastError();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!f->xloop && !f->xpeer && (++polls & 1023) == 0) {
            for (int k = 0; k <= orbf_frontend::NXC; ++k)
                if (f->xc[k] && exchange_comm_async_error(f->xc[k]) != 0) { f->x_broken = true; return ORB_E_HIP; }
        }
        if (ms > limit_ms) {
            f->x_broken = true;
            morb::set_error("multi-GPU exchange: the exchange of step %ld had not completed after %.0f ms -- a rank is dead or far behind "
                            "(MORB_EXCHANGE_TIMEOUT_MS)", seq, ms);
            return ORB_E_TIMEOUT;
        }
        if (ms > 0.2) std::this_thread::sleep_for(std::chrono::microseconds(ms > 5.0 ? 200 : 20)); else __builtin_ia32_pause();
    }
    if (f->xpeer) {
        const unsigned long long miss = morb::peer_missing(f->xpeer, (int)(seq % orbf_frontend::NX), redo ? 1 : 0);
        if (miss) {
            f->x_broken = true;
            std::string who;
            for (int r = 0; r < f->xworld; ++r) if ((miss >> r) & 1) who += (who.empty() ? "" : ", ") + std::to_string(r);
            morb::set_error("multi-GPU exchange: rank(s) %s did not deliver the %sblock of step %ld within %ld ms (MORB_EXCHANGE_TIMEOUT_MS): dead or far behind",
                            who.c_str(), redo ? "re-shipped " : "", seq, f->x_timeout_ms);
            return ORB_E_TIMEOUT;
        }
    }
    return ORB_OK;
}

// The exchange of step `seq`: the slot's buffers, the bookkeeping, and the calls -- queued for the issuer or made here.  redo: the block
// is shipped a second time (see XSlot), over the last communicator.
static int exchange_issue(orbf_frontend* f, long seq, const orbm_frame* F, hipStream_t st, bool redo, const orbf_image* images, int set = -1) {
    orbf_frontend::XSlot& X = f->xs[seq % orbf_frontend::NX];
    const size_t block = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    MORB_ARG(F->desc_rows == f->cap_total);
    const int n_cams = f->xworld * f->n_cams;
    int rc;
    if ((rc = x_wait_job(f, X.job))) return rc;   // (the slot's previous exchange -- eight steps ago -- has long been issued; its buffers may grow now)
    if ((!f->xpeer && (rc = X.recv.reserve((size_t)f->xworld * block))) || (rc = X.list.reserve((size_t)f->xworld * F->desc_rows * 32)) ||
        (rc = X.gstart.reserve(n_cams + 1 + 4)) || (rc = X.gcnt.reserve(n_cams + 3)) || (rc = X.out.reserve(F->desc_rows, f->xworld * F->desc_rows)))
        return rc;
    orbf_frontend::XJob J{0, seq, F->b->d_desc.p, F->desc_rows, st, redo};
    if (f->x_async) {
        std::lock_guard<std::mutex> lk(f->x_mu);
        if ((rc = x_failed(f))) return rc;
        J.id = ++f->x_pushed;
        f->x_jobs.push_back(J);
        f->x_cv.notify_all();
    } else if ((rc = exchange_calls(f, J))) return rc;
    X.job = J.id;
    if (!redo) {
        X.seq = seq;
        if (images) { X.images.assign(images, images + f->n_cams); X.fp = image_fingerprints(images, f->n_cams); }
        else { X.images.clear(); X.fp.clear(); }
        f->x_next = seq + 1;
        if (set >= 0) f->set_xseq[set] = seq;
    }
    return ORB_OK;
}

static bool same_images(const std::vector<orbf_image>& a, const orbf_image* b, int n);
static bool same_content(const std::vector<uint64_t>& fp, const orbf_image* b, int n);
static std::vector<uint64_t> image_fingerprints(const orbf_image* images, int n);

int orbf_peek_block(orbf_frontend* f, const orbf_image* images, const uint8_t** d_block, size_t* block_bytes, int* cap_rows) {
    MORB_ARG(f && images && d_block && block_bytes && cap_rows);
    *d_block = nullptr; *block_bytes = 0; *cap_rows = 0;
    if (f->pending.active || f->inflight.empty() || !same_images(f->inflight.front().images, images, f->n_cams) ||
        !same_content(f->inflight.front().fp, images, f->n_cams)) return ORB_OK;
    const orbf_frontend::InFlight& I = f->inflight.front();
    MORB_HIP(hipSetDevice(f->device));
    if (hipEventQuery(f->ev_ready[I.set]) != hipSuccess) { (void)hipGetLastError(); return ORB_OK; }
    if (orbx_peek_status(f->exs[I.e]) != 0) return ORB_OK;
    const orbm_frame* F = f->pframe[I.set];
    if (!F) return ORB_OK;
    *d_block = F->b->d_desc.p; *cap_rows = F->desc_rows; *block_bytes = (size_t)F->desc_rows * 32 + ORBM_BLOCK_TRAILER;
    return ORB_OK;
}

int orbf_export_features(orbf_frontend* f, orbf_device_features* out) {
    MORB_ARG(f && out);
    const orbm_frame* F = f->last_frame;
    if (!F || (f->pending.active && f->pending.fr)) { morb::set_error("orbf_export_features: no completed step (call it after orbf_step / orbf_step_end)"); return ORB_E_ARG; }
    memset(out, 0, sizeof(*out));
    out->n_cams = f->n_cams; out->n_total = 0;
    for (int c = 0; c < f->n_cams && c < 8; ++c) { out->counts[c] = f->counts[c]; out->n_total += f->counts[c]; }
    out->d_desc = F->b->d_desc.p; out->d_angle = F->b->d_ang.p; out->d_un_x = F->b->d_x.p; out->d_un_y = F->b->d_y.p;
    out->d_octave = F->b->d_oct.p; out->d_uright = F->b->d_ur.p;
    out->stream = f->mt->stream;
    return ORB_OK;
}

int orbf_prefetch(orbf_frontend* f, const orbf_image* next_images) {
    MORB_ARG(f && next_images);
    // (the step about to be called may itself still be in flight: two timesteps beyond it can be announced)
    if ((int)(f->inflight.size() + f->announced.size()) >= f->n_ex + 1) { morb::set_error("too many future timesteps announced (at most %d beyond the next step)", f->n_ex); return ORB_E_ARG; }
    f->announced.emplace_back(next_images, next_images + f->n_cams);
    return ORB_OK;
}

int orbf_ahead_depth(const orbf_frontend* f) { return f ? f->n_ex : ORB_E_ARG; }

// A begin that failed (images that differ from the announced ones, a reserve or an enqueue that failed) leaves no step behind: its
// number goes back, so that the retry -- with the right images -- is the same step on this rank as on the others (a spent
// number made every later step miss its exchange).  What was shipped under the number stays shipped: the retry is checked against it.
static void begin_failed(orbf_frontend* f) {
    if (f->pending.seq_taken && f->step_seq == f->pending.seq + 1) f->step_seq = f->pending.seq;
    f->pending.active = false; f->pending.seq_taken = false;
}

int orbf_step_begin(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags, int* block_ready) {
    MORB_ARG(f && images && nq >= 0 && (nq == 0 || queries));
    f->t_entry = std::chrono::steady_clock::now();
    int rc = orbf_step_begin_impl(f, images, queries, nq, flags, false, block_ready);
    if (rc) begin_failed(f);
    return rc;
}

static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out);

int orbf_step_motion_begin(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, int* block_ready) {
    MORB_ARG(f && images && motion);
    f->t_entry = std::chrono::steady_clock::now();
    int rc = orbf_step_begin_impl(f, images, nullptr, 0, flags, true, block_ready, motion);
    if (rc) begin_failed(f);
    return rc;
}

int orbf_step_end(orbf_frontend* f, orbf_result* out) {
    MORB_ARG(f && out);
    return orbf_step_end_impl(f, out);
}

// the previous step's features (still in their pinned result set) under the stream's motion -> this step's queries
static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out) {
    const int nq = f->prev_n;
    *nq_out = nq;
    if (!nq) return ORB_OK;
    int rc;
    if ((rc = f->h_queries.reserve((size_t)nq * sizeof(orbm_query)))) return rc;
    const orbf_frontend::ResultSet& R = f->rs[f->cur];
    return orbm_queries_from_motion(R.kps.p, R.desc.p, R.depth.p, f->prev_cam_of.data(), nq, motion->du, motion->dv, motion->th,
                                    f->scale_factors.data(), f->mbf, reinterpret_cast<orbm_query*>(f->h_queries.p),
                                    R.unx.p, R.uny.p);  // (mvKeysUn: equal to the keypoint positions without a calibration)
}

int orbf_run_stream(orbf_frontend* f, const orbf_image* ring, int ring_len, int t0, int steps, int ahead, int* announced_upto,
                    const orbf_motion* motion, int th_low, float ratio, orbf_stream_stats* out) {
    MORB_ARG(f && ring && ring_len >= 1 && t0 >= 0 && steps >= 0 && ahead >= 0 && announced_upto && motion && out);
    // (refused before any step has run: orbf_prefetch would refuse the announcement in the middle of the stream otherwise)
    const int ahead_max = std::min(f->n_ex + 1, (int)orbf_frontend::NEX);
    if (ahead > ahead_max) { morb::set_error("orbf_run_stream: ahead = %d, this handle takes at most %d (min(orbf_ahead_depth() + 1, 3))", ahead, ahead_max); return ORB_E_ARG; }
    memset(out, 0, sizeof(*out));
    const auto t_start = std::chrono::steady_clock::now();
    auto images_of = [&](int t) { return ring + (size_t)(t % ring_len) * f->n_cams; };
    orbf_result r;
    for (int t = t0; t < t0 + steps; ++t) {
        // every timestep up to t + ahead is announced exactly once, in order (the step's own images are never announced)
        for (int a = std::max(*announced_upto + 1, t + 1); ahead > 0 && a <= t + ahead; ++a) {
            int rc = orbf_prefetch(f, images_of(a));
            if (rc) return rc;
            *announced_upto = a;
        }
        int rc = orbf_step_motion(f, images_of(t), motion, ORBF_NO_QUERY_RECORDS, &r);   // (the loop never reads orbf_result::queries)
        if (rc) return rc;
        if (*announced_upto < t) *announced_upto = t;
        const int nx = r.cross_best_dist ? orbm_count_ratio_accepted(r.cross_best_dist, r.cross_second_dist, r.n_total, th_low, ratio) : 0;
        if (nx < 0) return nx;
        out->features += r.n_total; out->temporal_matches += r.nmatches; out->cross_accepted += nx;
        uint64_t h = out->digest ^ ((uint64_t)(uint32_t)r.n_total << 40) ^ ((uint64_t)(uint32_t)r.nmatches << 20) ^ (uint64_t)(uint32_t)nx;
        h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        out->digest = h;
    }
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return ORB_OK;
}

int orbf_step_motion(orbf_frontend* f, const orbf_image* images, const orbf_motion* motion, int flags, orbf_result* out) {
    MORB_ARG(f && images && motion && out);
    f->t_entry = std::chrono::steady_clock::now();
    return orbf_step_impl(f, images, nullptr, 0, flags, out, true, motion);
}

int orbf_step_motion_ahead(orbf_frontend* f, const orbf_image* images, const orbf_image* next_images, const orbf_motion* motion,
                           int flags, int th_low, float ratio, orbf_result* out, int* n_cross) {
    MORB_ARG(f && images && motion && out && n_cross);
    int rc;
    if (next_images && (rc = orbf_prefetch(f, next_images))) return rc;
    if ((rc = orbf_step_motion(f, images, motion, flags, out))) return rc;
    *n_cross = out->cross_best_dist ? orbm_count_ratio_accepted(out->cross_best_dist, out->cross_second_dist, out->n_total, th_low, ratio) : -1;
    return *n_cross < -1 ? *n_cross : ORB_OK;
}

// An extraction that ran ahead is only valid for the step that consumes it if the images are still the ones that were
// uploaded.  Pointers, sizes and strides say nothing about a caller that refilled the same buffer in between.  The exact answer
// is the caller's: orbf_image::generation (compared by same_images).  Host images WITHOUT a generation carry a sampled
// fingerprint of their content instead: 32 probes of 64 bytes spread over the rows (2 KB per image, well under a microsecond),
// taken when the upload was enqueued and again when the step arrives -- a heuristic that catches whole-frame refills, not an
// identity.  A mismatch drops what is in flight and the step extracts its images again.  Device images cannot be probed from
// the host: without a generation they are identified by their pointer alone, as orbf.h says.
static uint64_t image_fingerprint(const orbf_image& im) {
    if (im.generation || im.on_device || !im.data || im.width <= 0 || im.height <= 0) return 0;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)im.width << 32) ^ (uint64_t)im.height;
    const int span = std::min(64, im.width);
    for (int k = 0; k < 32; ++k) {
        const int row = (int)(((long long)k * im.height) / 32);
        const int col = im.width > span ? (k * 149) % (im.width - span + 1) : 0;
        const uint8_t* p = im.data + (size_t)row * im.stride + col;
        for (int b = 0; b + 8 <= span; b += 8) { uint64_t v; memcpy(&v, p + b, 8); h = (h ^ v) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    }
    return h | 1;
}

static std::vector<uint64_t> image_fingerprints(const orbf_image* images, int n) {
    std::vector<uint64_t> fp(n);
    for (int c = 0; c < n; ++c) fp[c] = image_fingerprint(images[c]);
    return fp;
}

static bool same_content(const std::vector<uint64_t>& fp, const orbf_image* b, int n) {
    if ((int)fp.size() != n) return false;
    for (int c = 0; c < n; ++c) if (fp[c] != image_fingerprint(b[c])) return false;
    return true;
}

static bool same_images(const std::vector<orbf_image>& a, const orbf_image* b, int n) {
    if ((int)a.size() != n) return false;
    for (int c = 0; c < n; ++c)
        if (a[c].data != b[c].data || a[c].width != b[c].width || a[c].height != b[c].height || a[c].stride != b[c].stride ||
            (a[c].on_device != 0) != (b[c].on_device != 0) || a[c].generation != b[c].generation)
            return false;
    return true;
}


static void fill_cam_capacities(orbf_frontend* f, orbx_extractor* ex, orbm_cam_features* cams) {
    for (int c = 0; c < f->n_cams; ++c) {
        cams[c].d_kps = orbx_device_keypoints(ex, c); cams[c].d_desc = orbx_device_descriptors(ex, c);
        cams[c].n = f->cam_cap[c]; cams[c].d_depth = f->d_depth[c]; cams[c].depth_stride = f->depth_stride[c];
    }
}

// Uploads + the whole extractor `e` for one timestep into result set `set`, nothing synchronised.  Small rigs: the
// describe kernel writes the merged frame pframe[set] through a FrameSink (*went_async = 1 unless the extractor took its
// synchronous host-quadtree path; then the frame was not filled).
static int enqueue_extract(orbf_frontend* f, int e, const orbf_image* images, int set, int* W_out, int* H_out, int* went_async,
                           bool with_cross, bool defer_events = false, bool defer_mirror = false, long seq = -1) {
    orbm_matcher* m = f->mt;
    orbx_extractor* ex = f->exs[e];
    int rc, W = 0, H = 0;
    for (int c = 0; c < f->n_cams; ++c) {
        const orbf_image& im = images[c];
        rc = im.on_device ? orbx_upload_device(ex, c, im.data, im.width, im.height, im.stride)
                          : orbx_upload(ex, c, im.data, im.width, im.height, im.stride);
        if (rc) return rc;
        W = std::max(W, im.width); H = std::max(H, im.height);
    }
    if (W == 0 || H == 0) { W = f->max_w; H = f->max_h; }
    *W_out = W; *H_out = H;
    orbf_frontend::ResultSet& R = f->rs[set];
    R.cross_valid = false;
    if (f->xcomm && (rc = x_wait_stream(f, (hipStream_t)orbx_stream(ex)))) return rc;   // (the issuer is done with this chain's stream: steps ago, normally)
    if (f->xcomm && f->set_xseq[set] >= 0) {
        // an exchange reads this set's frame as its send buffer.  Its step has ended long ago (the end of a step waits for its
        // exchange) -- unless the extraction was dropped and never stepped: then the collective may still be waiting for a peer
        const orbf_frontend::XSlot& X = f->xs[f->set_xseq[set] % orbf_frontend::NX];
        if (X.seq == f->set_xseq[set]) MORB_HIP(hipStreamWaitEvent((hipStream_t)orbx_stream(ex), X.done, 0));
        f->set_xseq[set] = -1;
    }
    // defer_mirror: the kernels of this extraction leave the pinned result mirrors alone; the step copies them on its side
    // stream (frame_mirror_enqueue) instead of making every dependent kernel boundary wait for stores across PCIe
    if ((rc = orbx_set_host_mirror(ex, defer_mirror ? nullptr : R.kps.dp, defer_mirror ? nullptr : R.desc.dp, f->cap_total))) return rc;
    *went_async = 0;
    const bool small = small_rig(f);
    std::vector<orbm_cam_features> cams(f->n_cams);
    fill_cam_capacities(f, ex, cams.data());
    float bd[4];
    if ((rc = orbm_image_bounds(&f->calib, W, H, bd))) return rc;  // Frame::ComputeImageBounds
    if (f->pframe[set] && (f->pframe_W[set] != W || f->pframe_H[set] != H || f->pframe[set]->minX != bd[0] ||
                           f->pframe[set]->minY != bd[1] || f->pframe[set]->maxX != bd[2] || f->pframe[set]->maxY != bd[3])) {
        if (f->xcomm) {   // (an exchange may still read this frame's block: issued? then over?)
            if ((rc = x_wait_stream(f, nullptr))) return rc;
            for (auto& X : f->xs) if (X.done && X.seq >= 0 && !f->x_broken) (void)x_wait_done(f, X, X.seq, false);
        }
        orbm_frame_destroy(f->pframe[set]); f->pframe[set] = nullptr;  // (image size or calibration changed)
    }
    m->mirror_ur = R.ur.dp; m->mirror_depth = R.depth.dp; m->mirror_unx = R.unx.dp; m->mirror_uny = R.uny.dp;
    if (defer_mirror) { m->mirror_ur = nullptr; m->mirror_depth = nullptr; m->mirror_unx = nullptr; m->mirror_uny = nullptr; }
    struct MirrorsOff { orbm_matcher* m; ~MirrorsOff() { m->mirror_ur = nullptr; m->mirror_depth = nullptr; m->mirror_unx = nullptr; m->mirror_uny = nullptr; } } mirrors_off{m};
    if (small) {
        // the describe kernel writes the per-feature half of the frame itself (FrameSink)
        FrameSink sink;
        if (!f->pframe[set]) {
            rc = frame_prepare_sink(m, cams.data(), f->n_cams, f->mbf, bd[0], bd[1], bd[2], bd[3], &f->pframe[set], &sink);
            f->pframe_W[set] = W; f->pframe_H[set] = H;
        } else {
            orbm_frame* F = f->pframe[set];
            F->n_total = f->cap_total; F->counts_on_device = true; F->host_valid = false;
            rc = frame_sink_of(m, F, cams.data(), f->n_cams, f->mbf, &sink);
        }
        if (rc) return rc;
        if ((rc = orbx_set_frame_sink(ex, &sink))) return rc;
    } else {
        // larger rigs: the matcher's own kernels assemble the frame from the extractor's per-camera outputs
        if (!f->pframe[set]) {
            m->frame_min_rows = f->cap_total;
            rc = frame_shell(m, f->cap_total, f->n_cams, bd[0], bd[1], bd[2], bd[3], true, &f->pframe[set]);
            m->frame_min_rows = 0;
            if (rc) return rc;
            f->pframe_W[set] = W; f->pframe_H[set] = H;
        } else {
            orbm_frame* F = f->pframe[set];
            F->n_total = f->cap_total; F->counts_on_device = true; F->host_valid = false;
        }
    }
    // The frame's grid (larger rigs: the whole frame assembly) and the camera-pair top-2 are the tail of the extraction
    // chain: built on the extractor's stream right behind the describe kernel (counts read from HBM), so that a step's
    // matching starts with the search itself.  For small rigs the tail is issued from inside orbx_run_async and is
    // captured into the replayed launch chain (no host launches at all on replay); larger rigs launch it here (their
    // assembly stages a parameter block with a copy, which a replayed chain should not carry).
    struct Tail {
        orbf_frontend* f; orbx_extractor* ex; orbm_cam_features* cams; const float* bd; int set; bool small, with_cross;
        int phases = 3;   // 1 the frame, 2 the camera-pair top-2 (larger rigs issue them apart, with the chain's event in between)
        static int run(void* u, void* stream) {
            Tail& T = *static_cast<Tail*>(u);
            orbm_matcher* m = T.f->mt;
            orbf_frontend::ResultSet& R = T.f->rs[T.set];
            orbm_frame* frp = T.f->pframe[T.set];
            hipStream_t keep = m->stream;
            m->stream = (hipStream_t)stream;
            int rc = ORB_OK;
            if (T.phases & 1)
                rc = frame_from_device_impl(m, T.cams, T.f->n_cams, T.f->mbf, T.bd[0], T.bd[1], T.bd[2], T.bd[3], orbx_device_counts(T.ex),
                                            &frp, T.small);
            if (!rc && (T.phases & 2) && T.with_cross && T.f->n_cams > 1) {
                const int ncap = frp->n_total;
                rc = cross_enqueue_to(m->stream, frp->b->d_desc.p, ncap, frp->b->d_cam_start.p, T.f->n_cams, 0, ncap, frp->b->d_ntotal.p,
                                      R.cross.i.dp, R.cross.b.dp, R.cross.s.dp, R.cross.scratch.p);
            }
            m->stream = keep;
            return rc;
        }
    } tail{f, ex, cams.data(), bd, set, small, with_cross};
    const bool cross_here = with_cross && f->n_cams > 1;
    if (cross_here && (rc = R.cross.reserve(f->cap_total, f->cap_total))) return rc;  // (storage first: nothing allocates inside a capture)
    if (small) { if ((rc = m->h_ring.reserve((64 * sizeof(CamFeat) + 65 * sizeof(int) + 64 * sizeof(int)) * 4))) return rc; }
    if (small && (rc = orbx_set_chain_tail(ex, &Tail::run, &tail, 1 + set * 2 + (cross_here ? 1 : 0)))) return rc;
    const int before = orbx_pending(ex);
    rc = orbx_run_async(ex);
    if (small) { (void)orbx_set_frame_sink(ex, nullptr); (void)orbx_set_chain_tail(ex, nullptr, nullptr, 0); }
    if (rc) return rc;
    *went_async = orbx_pending(ex) > before ? 1 : 0;
    f->cross_pending[set] = false;
    if (*went_async) {
        const bool split = !small && cross_here && !defer_events;   // (the top-2 behind the chain's event: see ev_cross)
        if (!small) { tail.phases = split ? 1 : 3; if ((rc = Tail::run(&tail, orbx_stream(ex)))) return rc; }
        R.cross_valid = cross_here;
        if (!defer_events) {   // (an inline step records its events behind its matching: step_enqueue)
            hipError_t he = hipEventRecord(f->ev_ready[set], (hipStream_t)orbx_stream(ex));
            if (he != hipSuccess) { morb::set_error("hipEventRecord: %s", hipGetErrorString(he)); return ORB_E_HIP; }
        }
        if (split) {
            tail.phases = 2;
            if ((rc = Tail::run(&tail, orbx_stream(ex)))) return rc;
            MORB_HIP(hipEventRecord(f->ev_cross[set], (hipStream_t)orbx_stream(ex)));
            f->cross_pending[set] = true;
        }
        // the step's exchange: behind everything its matching waits for, on this chain's stream (the block is this set's frame).  A
        // block whose quadtree left the device limits says so in its trailer and is shipped again at the end of its step.
        if (f->xcomm && f->x_placement == 3 && seq >= 0 && seq == f->x_next && (rc = exchange_issue(f, seq, f->pframe[set], (hipStream_t)orbx_stream(ex), false, images, set)))
            return rc;
    }
    return ORB_OK;
}

static int ensure_extractor(orbf_frontend* f, int e) {
    if (f->exs[e]) return ORB_OK;
    int rc = orbx_create(f->params.data(), f->n_cams, f->max_w, f->max_h, f->device, &f->exs[e]);
    if (!rc) rc = orbx_set_inplace_level0(f->exs[e], 1);
    // the spare extractor works on the matcher's stream: the one stream of the handle that never holds a wait for a later step, and
    // no fifth hardware queue (a stream beyond four shares a queue with one of the chains -- possibly behind such a wait)
    if (!rc && e == orbf_frontend::SPARE) rc = orbx_adopt_stream(f->exs[e], f->mt->stream);
    return rc;
}

// Everything in flight is waited for and dropped (results of prefetched extractions included).
static int orbf_drain(orbf_frontend* f) {
    MORB_HIP(hipSetDevice(f->device));
    if (f->xcomm) { const int xr = x_wait_stream(f, nullptr); if (xr) return xr; }
    if (f->xcomm && f->x_placement == 3 && !f->inflight.empty()) f->x_spare_until = std::max(f->x_spare_until, f->x_next - 1);
    if (f->xcomm && f->x_placement == 3) {
        // the chains' streams may end in collectives that wait for peers which have not announced that far (and may themselves be
        // waiting for THIS rank, e.g. for a block shipped again): wait for the extractions, not for the exchanges behind them
        for (const auto& I : f->inflight) MORB_HIP(hipEventSynchronize(f->ev_ready[I.set]));
    } else
    for (int e = 0; e < orbf_frontend::NEX_ALL; ++e) if (f->exs[e]) MORB_HIP(hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[e])));
    MORB_HIP(hipStreamSynchronize(f->mt->stream));
    if (f->mt->side_stream) MORB_HIP(hipStreamSynchronize(f->mt->side_stream));
    for (int e = 0; e < orbf_frontend::NEX_ALL; ++e)
        while (f->exs[e] && orbx_pending(f->exs[e]) > 0) { int rc = orbx_discard(f->exs[e]); if (rc < 0) return rc; }   // (dropped: nothing is completed)
    f->inflight.clear();
    return ORB_OK;
}

// The next free result set / extractor for a timestep that is about to be extracted.  Sets go round robin.  Isolated steps
// (nothing in flight) always run on extractor 0; overlapped ones alternate, so that two extraction chains are on the GPU
// at a time and each extractor keeps seeing the same two (count slot, result set) pairs -- its captured launch chains stay valid.
static void next_slot(orbf_frontend* f, int* e, int* set, long seq) {
    *set = (f->last_set + 1) % orbf_frontend::NSETS;
    if (*set == f->cur) *set = (*set + 1) % orbf_frontend::NSETS;  // (the caller still reads the last step's results)
    *e = (f->inflight.empty() || f->n_ex < 2) ? 0 : (f->last_e + 1) % f->n_ex;
    if (f->xcomm && seq <= f->x_spare_until) *e = orbf_frontend::SPARE;   // (nothing is in flight then: see the look-ahead loop)
    f->last_set = *set; f->last_e = *e;
}

// A timestep in two halves.  orbf_step_begin enqueues everything (this step's matching, the extraction of the announced
// steps) and returns; orbf_step_end blocks once and collects.  Between the two a caller may enqueue work of its own that
// only needs the step's export block -- the multi-GPU exchange -- when begin reported the block ready.
static int step_enqueue(orbf_frontend* f, orbf_frontend::Pending& P, bool first_attempt);

static int queries_from_previous_step(orbf_frontend* f, const orbf_motion* motion, int* nq_out);

// motion != NULL: the queries are built here from the previous step's features (orbf_step_motion) -- AFTER this step's
// extraction has been enqueued, so that the GPU is already working while the host projects the points.
static int orbf_step_begin_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                                bool queries_in_pinned, int* block_ready, const orbf_motion* motion) {
    orbf_frontend::Pending& P = f->pending;
    MORB_ARG(!P.active);
    P = orbf_frontend::Pending();
    P.t_impl = std::chrono::steady_clock::now();
    MORB_HIP(hipSetDevice(f->device));
    orbm_matcher* m = f->mt;
    int rc, went_async = 0;
    // (a persistent frame stays intact while its result set is f->cur: the motion stream's projection kernel reads it)
    const orbm_frame* prev_frame = f->last_frame && !f->last_frame_owned ? f->last_frame : nullptr;
    if (f->last_frame && f->last_frame_owned) orbm_frame_destroy(f->last_frame);
    f->last_frame = nullptr; f->last_frame_owned = false;
    if (f->xcomm) flags |= ORBF_SKIP_CROSS;   // the rig-wide matching of the exchange replaces the rank-local one
    P.images.assign(images, images + f->n_cams);
    P.nq = nq; P.flags = flags; P.from_motion = motion != nullptr;
    P.seq = f->step_seq++; P.seq_taken = true;
    if (f->xcomm && f->x_next > P.seq) {
        // this step's block has been shipped already (its extraction was announced ahead): the other ranks hold what THOSE images
        // gave, so with an exchange an announcement is binding
        const orbf_frontend::XSlot& X = f->xs[P.seq % orbf_frontend::NX];
        if (X.seq != P.seq || !same_images(X.images, images, f->n_cams) || !same_content(X.fp, images, f->n_cams)) {
            morb::set_error("multi-GPU exchange: the images of this step differ from the ones announced for it (orbf_prefetch), whose "
                            "descriptors have been shipped to the other ranks already -- with an exchange active announcements are binding");
            return ORB_E_ARG;
        }
    }

    // ---- this step's extraction: already in flight (orbf_prefetch during an earlier step) or enqueued now
    if (!f->inflight.empty() && same_images(f->inflight.front().images, images, f->n_cams) &&
        same_content(f->inflight.front().fp, images, f->n_cams)) {
        const orbf_frontend::InFlight& I = f->inflight.front();
        P.set = I.set; P.e = I.e; P.W = I.W; P.H = I.H; went_async = 1;
        P.ext_done = I.done;
        f->inflight.pop_front();
    } else {
        if (!f->inflight.empty()) {  // prefetched for other images: everything in flight is dropped
            if ((rc = orbf_drain(f))) return rc;
            f->announced.clear();
        }
        if (!f->announced.empty() && same_images(f->announced.front(), images, f->n_cams)) f->announced.pop_front();
        next_slot(f, &P.e, &P.set, P.seq);
        if ((rc = ensure_extractor(f, P.e))) return rc;
        // Nothing ran ahead for this step (a live rig: the images have only just arrived).  Its matching then goes onto the
        // extractor's own stream, right behind the extraction chain -- a kernel boundary instead of a cross-stream event
        // (measured: ~22 us between the chain's last kernel and the projection kernel on the matcher's stream) -- and the
        // camera-pair top-2 leaves the chain: it rides in the projection kernel's launch (SideJob in step_enqueue; on the
        // side stream when that does not apply) instead of standing in front of the search.
        P.inline_match = small_rig(f) && !f->xcomm;
        if (P.inline_match) { (void)orbx_set_chain_graph(f->exs[P.e], 0); (void)orbx_set_defer_done(f->exs[P.e], 1); }
        // (the copy rides with the camera-pair top-2: in the projection kernel's launch, or on the side stream it forks onto)
        P.mirror_requested = P.inline_match && !(flags & ORBF_SKIP_CROSS) && f->n_cams > 1;
        // the step waits for THIS extraction: page-locked host images are read by the ingest kernel directly (one launch for
        // all cameras instead of a pitched copy per camera: -13 us on 2 x 640x480).  Extractions that run ahead keep the copies:
        // they overlap other streams' kernels, a kernel that reads across PCIe would hold CUs for the whole transfer
        // (measured: overlapped steps 7 % slower on configs[1], 25 % on configs[4] with direct reads).
        (void)orbx_set_pinned_ingest(f->exs[P.e], 1);
        rc = enqueue_extract(f, P.e, images, P.set, &P.W, &P.H, &went_async, !(flags & ORBF_SKIP_CROSS) && !P.inline_match, P.inline_match,
                             P.mirror_requested, P.seq);
        (void)orbx_set_pinned_ingest(f->exs[P.e], 0);
        P.mirror_pending = P.mirror_requested;
        if (P.inline_match) { (void)orbx_set_chain_graph(f->exs[P.e], 1); (void)orbx_set_defer_done(f->exs[P.e], 0); }
        if (rc) return rc;
        if (P.inline_match && !went_async) P.inline_match = false;   // (host-quadtree path: everything was synchronous)
    }
    if (motion && f->motion_on_device && prev_frame && f->prev_n > 0 && prev_frame->n_total == f->prev_n &&
        prev_frame->n_cams == f->n_cams && f->scale_factors.size() <= 16) {
        // query i = feature i of the previous frame (still in HBM) moved by the motion: built by the projection kernel itself
        const FrameBufs* B = prev_frame->b;
        nq = f->prev_n; P.nq = nq;
        P.use_ms = true; P.motion = *motion;
        P.ms = MotionSrc{B->d_x.p, B->d_y.p, B->d_depth.p, B->d_ang.p, B->d_oct.p, (const uint4*)B->d_desc.p, B->d_cam_start.p, f->n_cams,
                         {}, motion->du, motion->dv, motion->th, f->mbf, nullptr};
        // (the level scales travel in the kernel arguments: an upload in orbf_create would be one more HIP call next to other threads'
        // graph captures, and a copy on the NULL stream a fifth hardware queue -- the first version measured 9 300 steps/s that way)
        for (size_t l = 0; l < f->scale_factors.size(); ++l) P.ms.scale[l] = f->scale_factors[l];
        f->q_host.resize((size_t)nq);   // (filled by orbf_step_end while it waits: orbf_result::queries, the host fallbacks)
        queries = nullptr;
    } else if (motion) {
        if ((rc = queries_from_previous_step(f, motion, &nq))) return rc;
        queries = reinterpret_cast<const orbm_query*>(f->h_queries.p); queries_in_pinned = true;
        P.nq = nq;
    }
    // queries go through pinned (device-mapped) staging and are read from there by the projection kernel
    if (nq && !P.use_ms) {
        if ((rc = f->h_queries.reserve((size_t)nq * sizeof(orbm_query))) || (rc = m->d_queries.reserve((size_t)nq * sizeof(orbm_query))))
            return rc;
        if (!queries_in_pinned) memcpy(f->h_queries.p, queries, (size_t)nq * sizeof(orbm_query));
        f->h_queries.publish();
    }
    P.J = SearchJob{nullptr, P.use_ms ? f->q_host.data() : reinterpret_cast<const orbm_query*>(f->h_queries.p), nq, nullptr, false, 0.f,
                    f->th_high, f->check_ori, 64, false};
    P.J.q_dev = nq && !P.use_ms ? reinterpret_cast<const orbm_query*>(f->h_queries.dp) : nullptr;   // no H2D on the step's critical chain
    P.J.msrc = P.use_ms ? &P.ms : nullptr;
    if (P.use_ms && (int)prev_frame->cam_start.size() == f->n_cams + 1)   // (query i = feature i of the previous frame: its camera capacities bound the counts)
    {
        for (int c = 0; c < f->n_cams; ++c) P.J.q_cam_max = std::max(P.J.q_cam_max, prev_frame->cam_start[c + 1] - prev_frame->cam_start[c]);
        P.J.q_cam_start = prev_frame->cam_start.data();   // (camera c's queries are its features of the previous frame: contiguous)
    }
    P.J.want_tags = f->poll_ok;
    if ((rc = f->h_match.reserve(std::max(f->cap_total, 1)))) return rc;
    P.async_path = went_async != 0;
    // The export block of this step is final already when its extraction chain has completed cleanly (the usual case with
    // steps announced ahead): then nothing of this step can be redone and a caller may ship the block right away.
    P.block_ready = false;
    if (P.async_path && !P.inline_match && (P.ext_done || hipEventQuery(f->ev_ready[P.set]) == hipSuccess)) {
        P.ext_done = true;   // (the matcher's stream then needs no event wait in front of the search: step_enqueue)
        P.block_ready = orbx_peek_status(f->exs[P.e]) == 0;
    } else { P.ext_done = false; (void)hipGetLastError(); }
    if ((rc = step_enqueue(f, P, true))) return rc;
    P.active = true;
    if (block_ready) *block_ready = P.block_ready ? 1 : 0;
    return ORB_OK;
}

// Enqueues the matching of the pending step (and, on the first attempt, the extraction of the announced steps).
static int step_enqueue(orbf_frontend* f, orbf_frontend::Pending& P, bool first_attempt) {
    orbm_matcher* m = f->mt;
    hipStream_t st = m->stream;
    orbx_extractor* ex = f->exs[P.e];
    hipStream_t st_e = (hipStream_t)orbx_stream(ex);
    orbf_frontend::ResultSet& R = f->rs[P.set];
    const bool do_cross = !(P.flags & ORBF_SKIP_CROSS);
    int rc;
    std::vector<orbm_cam_features>& cams = P.cams;
    cams.resize(f->n_cams);
    const bool inline_match = P.async_path && P.inline_match;
    struct StreamSwap {   // an inline step issues its matching on the extractor's stream
        orbm_matcher* m; hipStream_t keep; bool on;
        ~StreamSwap() { if (on) m->stream = keep; }
    } swap{m, m->stream, inline_match};
    if (inline_match) { m->stream = st_e; st = st_e; }
    if (P.async_path) {
        // matching follows the extraction chain (which ends with the frame grid): through its event, or simply behind it on
        // the same stream; counts are in HBM
        P.fr = f->pframe[P.set]; P.fr_persistent = true;
        if (!inline_match && !P.ext_done) MORB_HIP(hipStreamWaitEvent(st, f->ev_ready[P.set], 0));  // extraction + frame grid of this step
        P.n = P.fr->n_total;
    } else {
        rc = orbx_finish(ex);  // synchronises; counts are on the host from here on
        if (rc < 0) return rc;
        // the host-quadtree path returns with its describe kernel still running on the extractor's stream
        MORB_HIP(hipEventRecord(f->ev_extracted, st_e));
        MORB_HIP(hipStreamWaitEvent(st, f->ev_extracted, 0));
        P.n = 0;
        for (int c = 0; c < f->n_cams; ++c) {
            cams[c].d_kps = orbx_device_keypoints(ex, c); cams[c].d_desc = orbx_device_descriptors(ex, c);
            cams[c].n = orbx_count(ex, c);
            cams[c].d_depth = f->d_depth[c]; cams[c].depth_stride = f->depth_stride[c];
            P.n += cams[c].n;
        }
        // the frame-build kernel mirrors the stereo arrays straight into this step's pinned result set (keypoints and
        // descriptors were mirrored by the extractor's describe kernel)
        m->mirror_kps = nullptr; m->mirror_desc = nullptr; m->mirror_ur = R.ur.dp; m->mirror_depth = R.depth.dp;
        m->mirror_unx = R.unx.dp; m->mirror_uny = R.uny.dp;
        m->frame_min_rows = f->cap_total;  // every step's export block has the same size
        float bd[4];
        rc = orbm_image_bounds(&f->calib, P.W, P.H, bd);  // Frame::ComputeImageBounds
        P.fr = nullptr;
        if (!rc) rc = frame_from_device_impl(m, cams.data(), f->n_cams, f->mbf, bd[0], bd[1], bd[2], bd[3], nullptr, &P.fr);
        m->frame_min_rows = 0;
        m->mirror_ur = nullptr; m->mirror_depth = nullptr; m->mirror_unx = nullptr; m->mirror_uny = nullptr;
        if (rc) return rc;
        P.fr_persistent = false;
    }
    orbm_frame* fr = P.fr;
    const int n = P.n;
    P.J.cur = fr; P.J.cap = 64; P.J.device_path = false;
    // fork: the camera-pair top-2 only needs the frame's descriptor block, so it runs on the side stream next to
    // project + resolve (both are a handful of workgroups on a 256-CU part); join before the one host sync
    // (on the asynchronous path the cross top-2 normally rode at the end of the step's extraction chain already)
    P.cross_from_set = do_cross && P.async_path && R.cross_valid;
    bool forked = do_cross && n > 0 && !P.cross_from_set;
    // An isolated step (its matching sits right behind its own extraction on one stream) takes no fork at all: the camera-pair
    // top-2 and the copy into the pinned result mirrors ride in the projection kernel's launch (SideJob) -- a fork onto the
    // side stream and the join behind it cost ~18 us of queue time per step (measured), a few more workgroups cost none.
    SideJob side;
    const bool fused_side = forked && inline_match && first_attempt && P.nq > 0 && P.nq <= 65535 && !m->host_resolve &&
                            morb::side_fusable(n, n);
    if (fused_side) {
        if ((rc = morb::side_reserve(m, n, n))) return rc;
        memset(&side, 0, sizeof(side));
        side.d_desc = fr->b->d_desc.p; side.n = n; side.d_cam_start = fr->b->d_cam_start.p; side.n_cams = f->n_cams;
        side.d_range = fr->b->d_ntotal.p;
        side.o_idx = m->h_c0.dp; side.o_best = m->h_c1.dp; side.o_second = m->h_c2.dp;
        side.scratch = m->d_cscratch.p;
        side.with_mirror = P.mirror_pending;
        if (P.mirror_pending && (rc = morb::frame_mirror_job(fr, R.kps.dp, R.desc.dp, R.unx.dp, R.uny.dp, R.ur.dp, R.depth.dp, &side.mirror)))
            return rc;
        P.mirror_pending = false;
        P.J.side = &side;
        forked = false;
    }
    P.forked = forked;
    hipStream_t sd = forked ? morb::side_stream(m) : nullptr;
    if (forked && !sd) { if (!P.fr_persistent) orbm_frame_destroy(fr); P.fr = nullptr; return ORB_E_HIP; }
    if (forked) {  // the fork point is the finished frame; the launches on the side stream come after the search's
        hipError_t fe = hipEventRecord(m->ev_fork, st);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(sd, m->ev_fork, 0);
        if (fe != hipSuccess) { morb::set_error("stream fork: %s", hipGetErrorString(fe)); if (!P.fr_persistent) orbm_frame_destroy(fr); P.fr = nullptr; return ORB_E_HIP; }
    }
    const bool x_probe = f->x_timing && f->xcomm && first_attempt;
    if (x_probe) (void)hipEventRecord(f->ev_x[0], st);
    rc = search_enqueue(m, P.J, /*queries_already_on_device=*/true);
    if (x_probe) (void)hipEventRecord(f->ev_x[1], st);   // (behind the resolve, in front of whatever an exchange puts on this stream)
    if (P.mirror_pending && !forked) {   // (no side stream in play: the copy follows the search on its stream)
        if (!rc) rc = frame_mirror_enqueue(fr, st, R.kps.dp, R.desc.dp, R.unx.dp, R.uny.dp, R.ur.dp, R.depth.dp);
        P.mirror_pending = false;
    }
    if (forked) {
        if (!rc) rc = cross_enqueue(m, sd, fr->b->d_desc.p, n, fr->b->d_cam_start.p, f->n_cams, 0, n,
                                    P.async_path ? fr->b->d_ntotal.p : nullptr);
        if (P.mirror_pending) {   // the step's pinned result mirrors: next to project + resolve, behind the camera-pair top-2
            if (!rc) rc = frame_mirror_enqueue(fr, sd, R.kps.dp, R.desc.dp, R.unx.dp, R.uny.dp, R.ur.dp, R.depth.dp);
            P.mirror_pending = false;
        }
        // join (also on the error path, so that the side stream never outlives the frame)
        hipError_t je = hipEventRecord(m->ev_join, sd);
        if (je == hipSuccess) je = hipStreamWaitEvent(st, m->ev_join, 0);
        if (!rc && je != hipSuccess) { morb::set_error("stream join: %s", hipGetErrorString(je)); rc = ORB_E_HIP; }
    }
    if (inline_match && first_attempt) {   // the events the extraction left for us: behind the matching, not in front of it
        const int rd = orbx_record_done(ex);
        hipError_t he = hipEventRecord(f->ev_ready[P.set], st);
        if (!rc && rd) rc = rd;
        if (!rc && he != hipSuccess) { morb::set_error("hipEventRecord: %s", hipGetErrorString(he)); rc = ORB_E_HIP; }
    }
    if (rc) { (void)hipStreamSynchronize(st); if (!P.fr_persistent) orbm_frame_destroy(fr); P.fr = nullptr; return rc; }
    // ---- native exchange: a block that is final already goes out right behind the step's own matching
    if (first_attempt && f->xcomm && P.async_path && P.block_ready && f->x_next == P.seq) {   // (placement 1, or a step whose chain was enqueued before the exchange was set up)
        if ((rc = exchange_issue(f, P.seq, fr, st, false, P.images.data()))) return rc;
        m->foreign_work = true;   // (the end of the step cannot watch the resolve's tags: other work follows them on the stream)
    }
    // ---- announced timesteps go onto the extractors now: they run while this step is being matched.  At most two
    // are in flight; consecutive ones alternate between the two extractors (an extractor takes its next timestep as
    // a second run behind the one whose results are being matched here).
    while (P.async_path && first_attempt && f->overlap_ok && !f->announced.empty() && (int)f->inflight.size() < std::max(f->n_ex, 2)) {
        if (f->xcomm && P.seq + 1 + (long)f->inflight.size() <= f->x_spare_until) break;   // (dropped exchanges are still out: isolated steps on the spare extractor)
        const int prev_e = f->inflight.empty() ? P.e : f->inflight.back().e;
        const int e2 = f->n_ex > 1 ? (prev_e + 1) % f->n_ex : 0;
        if ((rc = ensure_extractor(f, e2))) { (void)hipStreamSynchronize(st); return rc; }
        if (orbx_pending(f->exs[e2]) >= 2) break;
        int set2 = (f->last_set + 1) % orbf_frontend::NSETS;
        if (set2 == f->cur) set2 = (set2 + 1) % orbf_frontend::NSETS;
        if (set2 == P.set) set2 = (set2 + 1) % orbf_frontend::NSETS;
        int w2 = 0, h2 = 0, async2 = 0;
        rc = enqueue_extract(f, e2, f->announced.front().data(), set2, &w2, &h2, &async2, !(P.flags & ORBF_SKIP_CROSS), false, false,
                             P.seq + 1 + (long)f->inflight.size());
        if (rc) { (void)hipStreamSynchronize(st); return rc; }
        f->last_set = set2; f->last_e = e2;
        if (async2) {
            orbf_frontend::InFlight I;
            I.images = f->announced.front(); I.set = set2; I.W = w2; I.H = h2; I.e = e2;
            I.fp = image_fingerprints(I.images.data(), f->n_cams);   // (the uploads were enqueued just above)
            f->inflight.push_back(std::move(I));
            f->announced.pop_front();
        } else {
            // the extractor ran synchronously (host quadtree): its outputs now belong to that future step, which cannot
            // be kept apart from a later one's -- give up overlapping; the steps extract again when their turn comes
            f->overlap_ok = false;
            f->announced.clear();
        }
    }
    P.t_enqueued = std::chrono::steady_clock::now();
    return ORB_OK;
}

static int orbf_step_end_impl(orbf_frontend* f, orbf_result* out) {
    orbf_frontend::Pending& P = f->pending;
    MORB_ARG(P.active && out);
    P.active = false;
    auto us_between = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<float, std::micro>(b - a).count();
    };
    MORB_HIP(hipSetDevice(f->device));
    orbm_matcher* m = f->mt;
    orbx_extractor* ex = f->exs[P.e];
    struct StreamSwap {   // an inline step's matching lives on the extractor's stream (retries of the search go there too)
        orbm_matcher* m; hipStream_t keep; bool on;
        ~StreamSwap() { if (on) m->stream = keep; }
    } swap{m, m->stream, P.async_path && P.inline_match};
    if (swap.on) m->stream = (hipStream_t)orbx_stream(ex);
    hipStream_t st = m->stream;
    orbf_frontend::ResultSet& R = f->rs[P.set];
    int rc, nmatches = 0;
    auto t_synced = P.t_impl;
    // the records of the queries the projection kernel built for itself: written while the GPU works, or (ORBF_NO_QUERY_RECORDS) only if
    // a host fallback of the search asks for them
    struct FillQ {
        static void run(void* ctx) {
            orbf_frontend* f = static_cast<orbf_frontend*>(ctx);
            const orbf_frontend::Pending& P = f->pending;
            const orbf_frontend::ResultSet& Rp = f->rs[f->cur];   // (f->cur / prev_cam_of still describe the previous step here)
            (void)orbm_queries_from_motion(Rp.kps.p, Rp.desc.p, Rp.depth.p, f->prev_cam_of.data(), P.nq, P.motion.du, P.motion.dv, P.motion.th,
                                           f->scale_factors.data(), f->mbf, f->q_host.data(), Rp.unx.p, Rp.uny.p);
        }
    };
    bool have_records = !P.use_ms;
    if (P.use_ms && P.nq > 0) {
        if (P.flags & ORBF_NO_QUERY_RECORDS) { P.J.q_fill = &FillQ::run; P.J.q_fill_ctx = f; }
        else { FillQ::run(f); have_records = true; }
    }
    // has the NEXT step's extraction completed?  Asked now, while this step's matching runs, so that the next orbf_step_begin need not
    if (!f->inflight.empty() && !f->inflight.front().done) {
        if (hipEventQuery(f->ev_ready[f->inflight.front().set]) == hipSuccess) f->inflight.front().done = true;
        else (void)hipGetLastError();
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1 && (rc = step_enqueue(f, P, false))) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t herr = hipSuccess;
        bool polled = false;
        if (f->poll_ok && P.async_path && P.J.pollable && !P.forked && !m->foreign_work) {
            // the resolve is the last thing on the stream and tags its result words with the launch's sequence number: watch
            // the status word arrive (a few microseconds sooner than the end-of-kernel signal travels through the runtime);
            // search_finish then takes every other word the same way
            volatile int32_t* flag = m->h_match.p;
            for (int spin = 0; spin < 400000; ++spin) {
                if ((*flag >> 20) == P.J.seq) { polled = true; break; }
                __builtin_ia32_pause();
            }
        }
        if (!polled) herr = hipStreamSynchronize(st);
        m->foreign_work = false;
        out->gpu_wait_us = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
        out->host_us[0] = us_between(f->t_entry, P.t_impl); out->host_us[1] = us_between(P.t_impl, P.t_enqueued); out->host_us[2] = out->gpu_wait_us;
        t_synced = std::chrono::steady_clock::now();
        if (herr != hipSuccess) {
            morb::set_error("hipStreamSynchronize: %s", hipGetErrorString(herr));
            if (!P.fr_persistent && P.fr) orbm_frame_destroy(P.fr);
            return ORB_E_HIP;
        }
        if (P.async_path) {
            // this step's run (the oldest of its extractor) completed long ago: adopts its counts.  A step whose result words
            // were watched arriving has proof of that (the resolve was ordered behind the run) and skips the event wait.
            rc = polled ? orbx_finish_completed(ex) : orbx_finish(ex);
            if (rc < 0) return rc;
            if (rc == 1 || rc == 2) {
                // a pyramid level was outside the device quadtree's limits: this step is redone on the synchronous path
                if (rc == 2 || !f->inflight.empty()) {  // ... from its images: later timesteps are in flight, drop them first
                    if ((rc = orbf_drain(f))) return rc;
                    f->announced.clear();
                    f->overlap_ok = false;
                    int w2, h2, a2;
                    if (f->xcomm && P.seq <= f->x_spare_until) {   // (the chains' streams may hold waits for the dropped steps)
                        P.e = orbf_frontend::SPARE;
                        if ((rc = ensure_extractor(f, P.e))) return rc;
                        ex = f->exs[P.e];
                    }
                    if ((rc = enqueue_extract(f, P.e, P.images.data(), P.set, &w2, &h2, &a2, false))) return rc;
                    if (a2) { rc = orbx_finish(ex); if (rc < 0) return rc; }
                }
                P.fr = nullptr; P.fr_persistent = false;
                P.async_path = false;
                P.mirror_pending = P.mirror_requested;   // (the redone frame is mirrored again)
                f->clean_steps = 0;
                continue;
            }
            for (int c = 0; c < f->n_cams; ++c) f->counts[c] = orbx_count(ex, c);
            frame_set_counts(P.fr, f->counts.data());
            P.n = P.fr->n_total;
            if (!f->overlap_ok && ++f->clean_steps >= 3) f->overlap_ok = true;  // (e.g. the extractor has switched its BIG pass on)
        } else {
            for (int c = 0; c < f->n_cams; ++c) f->counts[c] = P.cams[c].n;
        }
        break;
    }
    if (!P.async_path || !f->overlap_ok) f->announced.clear();  // (hints are only honoured on the asynchronous path)
    const int n = P.n, nq = P.nq;
    const bool do_cross = !(P.flags & ORBF_SKIP_CROSS) && n > 0;
    rc = search_finish(m, P.J, f->h_match.p, &nmatches);
    if (rc) { if (!P.fr_persistent) orbm_frame_destroy(P.fr); return rc; }
    f->last_frame = P.fr; f->last_frame_owned = !P.fr_persistent;  // (returned to the pool when the next step starts)
    f->cur = P.set;
    f->prev_n = n;
    f->prev_cam_of.resize(n);
    for (int c = 0, g = 0; c < f->n_cams; ++c)
        for (int k = 0; k < f->counts[c]; ++k) f->prev_cam_of[g++] = c;
    out->n_queries = nq;
    if (P.use_ms && (P.flags & ORBF_NO_QUERY_RECORDS)) have_records = P.J.q_fill == nullptr;   // (a host fallback may have written them)
    out->queries = !P.use_ms ? reinterpret_cast<const orbm_query*>(f->h_queries.p) : (have_records && nq > 0 ? f->q_host.data() : nullptr);
    if (P.from_motion && (P.flags & ORBF_NO_QUERY_RECORDS) && !P.use_ms) out->queries = nullptr;   // (the flag means the same on the host-built path)
    out->n_cams = f->n_cams; out->n_total = n; out->counts = f->counts.data();
    out->kps = R.kps.p; out->desc = R.desc.p; out->uright = R.ur.p; out->depth = R.depth.p;
    out->un_x = R.unx.p; out->un_y = R.uny.p;
    out->nmatches = nmatches; out->match_of_feature = f->h_match.p;
    const bool from_set = P.cross_from_set && P.async_path;  // (a step redone on the synchronous path matched in its own launch)
    if (f->cross_pending[P.set]) {   // (larger rigs: the camera-pair top-2 ran next to this step's search, behind an event of its own)
        if (from_set && do_cross) MORB_HIP(hipEventSynchronize(f->ev_cross[P.set]));
        f->cross_pending[P.set] = false;
    }
    out->cross_best_idx = do_cross ? (from_set ? R.cross.i.p : m->h_c0.p) : nullptr;
    out->cross_best_dist = do_cross ? (from_set ? R.cross.b.p : m->h_c1.p) : nullptr;
    out->cross_second_dist = do_cross ? (from_set ? R.cross.s.p : m->h_c2.p) : nullptr;
    out->rig_cams = 0; out->rig_counts = nullptr;
    if (f->xcomm) {
        // Every rank issues exactly one all-gather per step, in step order: with the step's extraction chain (placement 3), behind its
        // search when the block was final at begin (placement 1), or here -- a step whose extraction ran synchronously.
        orbf_frontend::XSlot& X = f->xs[P.seq % orbf_frontend::NX];
        if (f->x_next == P.seq && (rc = exchange_issue(f, P.seq, f->last_frame, st, false, P.images.data()))) return rc;
        MORB_ARG(X.seq == P.seq);
        if ((rc = x_wait_job(f, X.job))) return rc;
        if ((rc = x_wait_done(f, X, P.seq, false))) return rc;
        const int gc = f->xworld * f->n_cams;
        if (X.gcnt.p[gc + 2] != 0) {
            // some rank's block came from an extraction that fell back to the host path afterwards (every rank reads the same marks in
            // the same gathered blocks): all ranks ship the step's final blocks once more
            if ((rc = exchange_issue(f, P.seq, f->last_frame, st, true, nullptr))) return rc;
            ++f->x_redos;
            if ((rc = x_wait_job(f, X.job))) return rc;
            if ((rc = x_wait_done(f, X, P.seq, true))) return rc;
            if (X.gcnt.p[gc + 2] != 0) { morb::set_error("multi-GPU exchange: a block shipped again is still marked unfinished"); return ORB_E_HIP; }
        }
        m->foreign_work = false;
        if (f->x_timing && f->ev_x[0] && X.t_done) {   // (everything has completed: search start -> search done, search start -> exchange done)
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, f->ev_x[0], f->ev_x[1]) == hipSuccess) f->x_us[0] = a * 1000.f; else (void)hipGetLastError();
            if (hipEventElapsedTime(&b, f->ev_x[0], X.t_done) == hipSuccess) f->x_us[1] = b * 1000.f;   // (negative: the exchange was over before the search began)
            else { (void)hipGetLastError(); f->x_us[1] = 0.f; }
        }
        out->cross_best_idx = X.out.i.p; out->cross_best_dist = X.out.b.p; out->cross_second_dist = X.out.s.p;
        out->rig_cams = gc; out->rig_counts = X.gcnt.p;
        if (X.gcnt.p[gc + 1] != 0) {   // (k_repack_gathered clamped a remote count: nothing ran out of bounds)
            morb::set_error("multi-GPU exchange: %d per-camera counts of the gathered blocks were out of range (ranks disagree on "
                            "their capacities, or a block is corrupt)", X.gcnt.p[gc + 1]);
            return ORB_E_ARG;
        }
    }
    if (f->exs[orbf_frontend::SPARE] && P.e != orbf_frontend::SPARE && P.seq > f->x_spare_until && orbx_pending(f->exs[orbf_frontend::SPARE]) == 0) {
        // the spare extractor has done its part: its buffers are given back
        (void)hipStreamSynchronize((hipStream_t)orbx_stream(f->exs[orbf_frontend::SPARE]));
        orbx_destroy(f->exs[orbf_frontend::SPARE]);
        f->exs[orbf_frontend::SPARE] = nullptr;
    }
    out->host_us[3] = us_between(t_synced, std::chrono::steady_clock::now());
    return ORB_OK;
}

static int orbf_step_impl(orbf_frontend* f, const orbf_image* images, const orbm_query* queries, int nq, int flags,
                          orbf_result* out, bool queries_in_pinned, const orbf_motion* motion) {
    int rc = orbf_step_begin_impl(f, images, queries, nq, flags, queries_in_pinned, nullptr, motion);
    if (rc) { begin_failed(f); return rc; }
    return orbf_step_end_impl(f, out);
}


