// matcher_internal.h -- types and internal entry points shared by the translation units of the matcher / front end
// (hamming.hip: all-pairs kernels; frame.hip: frame assembly; search.hip: projection search + resolve; matcher.hip: handle
// + host helpers; exchange.hip: RCCL / loopback transport; frontend.hip: orbf_*).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/orbm.h"
#include "../../include/orbx.h"
#include "orb_common.h"
#include "frame_sink.h"
#include "mirror_dev.h"

using morb::DevBuf;
using morb::PinnedBuf;

struct FrameDev {
    int n_total, n_cams;
    const int* n_total_dev;  // non-NULL: the feature count is only known on the device (n_total is then the capacity)
    const float* un_x; const float* un_y; const float* uright;
    const int* octave;
    const uint4* desc;  // global-index order, 2 x uint4 per feature
    const int* cell_start; const int* items;
    float minX, minY, invW, invH;
};

constexpr int RESOLVE_K = 6;   // sorted shortlist per query built by k_project (search.hip)

struct CamFeat {
    const orb_keypoint* kps; const uint4* desc; const float* depth;
    int depth_stride, n, base;
};

constexpr size_t ORBM_BLOCK_TRAILER = 256;  // bytes behind the descriptor rows of a frame: int32 per-camera counts
// bit 30 of the first count: the block comes from an extraction whose device quadtree left its limits -- the step is being redone
// on the host path and the block will be shipped again (frontend.hip: the ranks of an exchange all see the bit and all take part)
constexpr int ORBM_BLOCK_REDO = 1 << 30;

struct FrameBufs {  // device storage of one frame; recycled through the matcher's pool (no hipMalloc per frame)
    DevBuf<float> d_x, d_y, d_ur, d_depth, d_ang;
    DevBuf<int32_t> d_oct, d_cell_start, d_items, d_cell_of, d_cursor, d_cam_start, d_ntotal;
    DevBuf<uint8_t> d_desc;
    DevBuf<orb_keypoint> d_kps;
    DevBuf<CamFeat> d_cams;
    void release() {
        d_x.release(); d_y.release(); d_ur.release(); d_depth.release(); d_ang.release(); d_oct.release();
        d_cell_start.release(); d_items.release(); d_cell_of.release(); d_cursor.release(); d_cam_start.release(); d_ntotal.release();
        d_desc.release(); d_kps.release(); d_cams.release();
    }
};

struct orbm_matcher {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;  // `stream` = the one in use (own or caller's)
    DevBuf<uint8_t> d_q, d_r, d_scratch, d_queries, d_occ;
    DevBuf<int32_t> d_i0, d_i1, d_i2, d_choice, d_claim, d_match, d_status, d_x0, d_x1, d_x2;
    DevBuf<int32_t> d_gclaim;  // claim tables of the resolve when they do not fit LDS (2 x features)
    DevBuf<int32_t> d_rsync;   // where the per-camera resolve's workgroups meet (k_resolve_cams): zero between launches
    DevBuf<int2> d_qmeta;      // {blocks, angle} of every query, written by k_project for the resolve
    DevBuf<orbm_window> d_win2; // second windows of a two-camera search
    DevBuf<uint16_t> d_u16;
    PinnedBuf<int32_t> h_i0, h_i1, h_i2, h_match;
    PinnedBuf<int32_t> h_gcnt;            // per-camera counts of a gathered multi-GPU exchange (+ own query count)
    DevBuf<int32_t> d_gstart;             // camera starts + {features, first query, queries} of the gathered list
    int gathered_cams = 0;                // cameras of the last orbm_cross_top2_gathered_enqueue
    PinnedBuf<int32_t> h_c0, h_c1, h_c2;  // cross top-2 results (own buffers: they coexist with a search's h_i0/h_i1)
    DevBuf<uint8_t> d_cscratch;           // cross top-2 slice partials
    // Created on first use (morb::side_stream): the cross top-2 next to project + resolve, the multi-GPU exchange.  Every stream is a
    // hardware queue, and the part's command processor keeps four of them busy side by side: a fifth queue shares a pipe with
    // another one, and a kernel then waits behind the DISPATCH of that queue's kernels (measured with three extraction streams +
    // matcher + this one: single kernels of the chains stretched to 40-55 us, 83 us per step instead of 50).
    hipStream_t side_stream = nullptr;
    // A front end that runs a multi-GPU exchange keeps everything on the matcher's own stream: the all-gather, the repack and the
    // rig-wide top-2 follow the step's search there.  Forked onto the side stream they were a fifth queue's worth of trouble next to
    // three extraction chains, and with two chains the fork / join events and the waiting small kernels cost more than the overlap
    // gave: 106-114 us per step against 72-74 us inline with three chains (forced exchange on one GPU, end of round 3).
    bool side_inline = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_q = nullptr;
    PinnedBuf<uint16_t> h_u16;
    PinnedBuf<uint8_t> h_ring;  // 4 slots of {CamFeat[64], int cam_start[65]} for asynchronous H2D
    // Host-written staging of the host-array entry points (orbm_frame_create; the queries / occupied flags of a search): the
    // host writes the packed arrays once (HBM through the large BAR, or mapped pinned memory), ONE kernel scatters a frame's
    // arrays into its buffers -- instead of a pageable hipMemcpyAsync per array.
    morb::StageBuf stage_f, stage_q;
    hipEvent_t ev_stage_f = nullptr;   // the unpack kernel of the last orbm_frame_create has read stage_f
    bool stage_f_busy = false;
    unsigned ring_pos = 0;
    std::vector<FrameBufs*> pool;  // free list
    // device-visible pinned destinations the next orbm_frame_from_device mirrors its merged arrays into (orbf_step)
    orb_keypoint* mirror_kps = nullptr; uint8_t* mirror_desc = nullptr; float* mirror_ur = nullptr; float* mirror_depth = nullptr;
    float* mirror_unx = nullptr; float* mirror_uny = nullptr;
    orb_calibration calib = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // orbm_set_calibration: undistortion applied by device-built frames (k1 == 0: off)
    int frame_min_rows = 0;  // the next device-built frame gets at least this many descriptor rows (fixed export block size)
    int last_status[4] = {0, 0, 0, 0};  // {status, nmatches, sweeps, longest list} of the last device resolve
    std::vector<int> rs_cam_count;     // scratch of search_enqueue: queries per camera (per-camera resolve)
    int rs_sweeps_hint = 24;           // sweeps the multi-workgroup resolve enqueues next time (what the last one needed + 4)
    bool host_resolve = false;     // MORB_HOST_RESOLVE=1: always use the host resolve (testing / fallback path)
    int resolve_seq = 0;           // sequence number of the last tagged resolve launch
    bool foreign_work = false;     // something other than a step's own search was put on the stream (orbf_step_end then waits for all of it)
    DevBuf<int32_t> d_mergecnt;    // running count of the merging workgroups that rode in resolve launches (MergeJob)
    bool merge_ready = false; unsigned merge_target = 0;
};
namespace morb { hipStream_t side_stream(orbm_matcher* m); }   // (lazily created; NULL after a reported failure)


struct orbm_frame {
    orbm_matcher* owner = nullptr;
    FrameBufs* b = nullptr;
    int n_total = 0, n_cams = 0;
    float minX = 0, minY = 0, maxX = 0, maxY = 0, invW = 0, invH = 0;
    bool device_built = false;
    bool counts_on_device = false;  // n_total is a capacity until orbf_step has synchronised
    int desc_rows = 0;              // descriptor rows the frame was created for (the count trailer sits behind them)
    // host copies used by the host resolve / orbm_frame_grid; filled at create for host-built frames, lazily otherwise
    mutable std::vector<int32_t> octave, cell_start, items;
    mutable std::vector<float> angle;
    mutable bool host_valid = false;
    std::vector<int32_t> cam_start;  // n_cams + 1
    // global indices are camera-major (camera c's features are [cam_start[c], cam_start[c + 1])): what every frame built by this
    // library is; a host-built frame with interleaved cam_of[] is not, and the per-camera resolve (k_resolve_cams) must not take it
    bool camera_major = true;
    FrameDev dev() const {
        FrameDev F;
        F.n_total = n_total; F.n_cams = n_cams; F.n_total_dev = counts_on_device ? b->d_ntotal.p : nullptr; F.un_x = b->d_x.p; F.un_y = b->d_y.p; F.uright = b->d_ur.p;
        F.octave = b->d_oct.p; F.desc = (const uint4*)b->d_desc.p; F.cell_start = b->d_cell_start.p; F.items = b->d_items.p;
        F.minX = minX; F.minY = minY; F.invW = invW; F.invH = invH;
        return F;
    }
};

// Queries of the motion stream built where they are used (orbf_step_motion): query i is feature i of the PREVIOUS step's frame,
// still in HBM, moved by (du, dv) -- the arithmetic of orbm_queries_from_motion (matcher.hip), operation for operation, so the
// host never writes a query record on the step's critical path (136 KB through the BAR per 2 x 1000-feature step before).
struct MotionSrc {
    const float* x; const float* y; const float* depth; const float* angle;   // the previous frame's arrays (global feature order)
    const int* octave; const uint4* desc; const int* cam_start; int n_cams;
    float scale[16];       // scale factor per pyramid level, by value (no table in memory: nothing to allocate or copy for it)
    float du, dv, th, mbf;
    orbm_query* rec_out;   // non-NULL: {blocks, angle} of every query into these records (the multi-workgroup resolve reads them)
};

// k_project + k_resolve on the device, one D2H of {status, matches}; falls back to host_resolve when the sweep limit is
// hit, retries with a larger capacity when a candidate list overflowed.  Split in two so that a caller can enqueue
// other work on the stream between the launch and the one synchronisation (orbf_step).
struct SearchJob {
    const orbm_frame* cur; const orbm_query* q; int nq; const uint8_t* occupied;
    bool points; float nnratio; int th_high, check_ori;
    int cap; bool device_path;
    bool pollable = false;              // single-workgroup resolve in flight with tagged result words (see k_resolve)
    bool want_tags = false; int seq = 0; // caller wants to watch the results arrive; sequence number of the launch in flight
    const orbm_query* q_dev = nullptr;  // device-visible alias of `q` when it lives in mapped pinned memory: read in place, no H2D
    const uint8_t* occ_dev = nullptr;   // device-visible copy of `occupied` (staged by the caller): no H2D either
    const orbm_window* win2_dev = nullptr;  // second windows of the queries (device memory), or NULL
    const struct SideJob* side = nullptr;   // work that shares the projection kernel's launch (consumed by search_enqueue)
    bool multi = false;                 // the resolve in flight is the multi-workgroup form (one launch per sweep)
    const MotionSrc* msrc = nullptr;    // queries built by the projection kernel itself (then `q` is only read by the host fallbacks)
    int q_cam_max = 0;                  // upper bound of the queries any one camera has (0 = not known; counted from `q` when that is final)
    const int* q_cam_start = nullptr;   // host, n_cams + 1 entries: the queries are camera-contiguous, camera c's are [q_cam_start[c], q_cam_start[c + 1])
    void (*q_fill)(void*) = nullptr; void* q_fill_ctx = nullptr;   // ... which call this first when `q` has not been written yet
};

// Work of an isolated orbf_step that rides in the projection kernel's launch instead of on a stream of its own (a fork onto a
// side stream and the join behind it cost ~18 us of queue time per step, measured): the camera-pair top-2 over the frame's
// descriptors and the copy of the frame into the pinned result mirrors.  All of it (the slice merge included) is complete
// before the resolve kernel behind it starts.
struct MergeJob;   // (hamming_dev.h)
struct SideJob {
    const uint8_t* d_desc; int n; const int* d_cam_start; int n_cams; const int* d_range;   // d_range = {features, first query, queries} in HBM
    int *o_idx, *o_best, *o_second;   // mapped pinned results
    void* scratch;                    // slice partials (merged by k_top2_merge behind the launch)
    bool with_mirror; morb::MirrorJob mirror;
};

// k_cross_top2 (+ merge) over `n` features in `d_desc` split into cameras by `d_cam_start`; queries [q_off, q_off+nq).
// Results land in the pinned mirrors m->h_c0/h_c1/h_c2 once stream `st` has been synchronised.
// Destination of a cross top-2: three mapped pinned result arrays + the HBM scratch of the slice partials.  The matcher
// owns one (h_c0..2 / d_cscratch); the front end owns one per result set, because it runs the cross matching of steps
// that were announced ahead at the end of their extraction chains.
struct CrossOut {
    PinnedBuf<int32_t> i, b, s;
    DevBuf<uint8_t> scratch;
    int reserve(int nq, int n);   // room for either form of the kernel (hamming.hip)
    void release() { i.release(); b.release(); s.release(); scratch.release(); }
};

namespace morb {
// ---- frame.hip
FrameBufs* take_bufs(orbm_matcher* m);
int reserve_frame(FrameBufs* b, int n, int n_cams);
int ensure_host_copies(const orbm_frame* f);
int frame_raise_lds_limit();   // per device, from orbm_create
int frame_shell(orbm_matcher* m, int n, int n_cams, float min_x, float min_y, float max_x, float max_y, bool counts_on_device,
                orbm_frame** out);
int frame_prepare_sink(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                       float max_x, float max_y, orbm_frame** out, FrameSink* sink);
int frame_sink_of(orbm_matcher* m, orbm_frame* F, const orbm_cam_features* cams, int n_cams, float mbf, FrameSink* sink);
// a finished frame's keypoints, descriptors, undistorted positions and stereo values into mapped pinned buffers, on `stream`
int frame_mirror_enqueue(orbm_frame* F, void* stream, orb_keypoint* h_kps, uint8_t* h_desc, float* h_unx, float* h_uny, float* h_ur,
                         float* h_depth);
int frame_mirror_job(orbm_frame* F, orb_keypoint* h_kps, uint8_t* h_desc, float* h_unx, float* h_uny, float* h_ur, float* h_depth,
                     MirrorJob* out);
// k_project next to a SideJob in one launch (hamming.hip); side_fusable: the sizes take the matrix-core top-2 this needs
struct ProjectArgs;
int launch_project_side(hipStream_t st, const ProjectArgs& P, const SideJob& S, struct ::MergeJob* defer_merge = nullptr);
int launch_merge(hipStream_t st, const struct ::MergeJob& M);
bool side_fusable(int nq, int n);
int side_reserve(orbm_matcher* m, int nq, int n);   // scratch and the pinned result arrays of the matcher's own CrossOut
int frame_from_device_impl(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x, float min_y,
                           float max_x, float max_y, const int* d_counts, orbm_frame** out, bool sink_filled = false);
void frame_set_counts(orbm_frame* F, const int* counts);
int phases_frame_build(unsigned long long* out64);
// ---- search.hip
int search_raise_lds_limits();  // per device, from orbm_create
int search_enqueue(orbm_matcher* m, SearchJob& J, bool queries_already_on_device = false);
int search_finish(orbm_matcher* m, SearchJob& J, int32_t* match_of_feature, int* nmatches);
int phases_resolve(unsigned long long* out64);
// ---- hamming.hip
int cross_enqueue_to(hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off, int nq,
                     const int* d_n, int* o_idx, int* o_best, int* o_second, void* scratch);
int cross_enqueue(orbm_matcher* m, hipStream_t st, const uint8_t* d_desc, int n, const int* d_cam_start, int n_cams, int q_off,
                  int nq, const int* d_n = nullptr);
int gathered_enqueue_to(hipStream_t st, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows, int cams_per_rank, int rank,
                        uint8_t* d_list, int* d_gstart, int* h_gcnt_dp, CrossOut& out);
// ---- exchange.hip
struct LoopComm;
int exchange_rccl_available();
int exchange_unique_id(uint8_t* out128);
int exchange_comm_init(void** comm, int world, const uint8_t* uid128, int rank);
int exchange_comm_clone(void* comm, int world, int rank, void** out);
int exchange_comm_async_error(void* comm);
void exchange_comm_abort(void* comm);
void exchange_comm_destroy(void* comm);
int exchange_allgather(void* comm, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st);
int loop_join(int group, int world, int rank, LoopComm** out);
void loop_leave(LoopComm* c);
int loop_allgather(LoopComm* C, const void* sendbuf, void* recvbuf, size_t bytes, hipStream_t st);
struct PeerComm;
size_t peer_handle_bytes();
int peer_world(const PeerComm* C);
int peer_rank(const PeerComm* C);
int peer_export(PeerComm** out, int world, int rank, size_t block, int nslots, long timeout_ms, uint8_t* handle);
int peer_open(PeerComm* C, const uint8_t* handles);
void peer_close(PeerComm* C);
const uint8_t* peer_recv(const PeerComm* C, int slot, int gen);
int peer_allgather(PeerComm* C, int slot, int gen, unsigned version, const void* send, hipStream_t st);
unsigned long long peer_missing(const PeerComm* C, int slot, int gen);
}  // namespace morb
