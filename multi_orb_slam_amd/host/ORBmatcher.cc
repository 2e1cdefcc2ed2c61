// ORBmatcher.cc -- see ORBmatcher.h.  Host glue: packs flat arrays for include/orbm.h and writes the results back
// into the caller's Frame the way the reference does.
#include "ORBmatcher.h"

#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "../../include/orbm.h"
#include "../../include/orbv.h"
#include "resident.h"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;     // reference src/ORBmatcher.cc:37
const int ORBmatcher::TH_LOW = 50;       // :38
const int ORBmatcher::HISTO_LENGTH = 30; // :39

// ---- error convention.  The reference's searches have no error channel: they return a match count.  A failed device call
// (there is no CPU fallback) returns 0 matches from the search, leaves the caller's match containers as the reference's own
// prologue leaves them, reports once on stderr and keeps the text for ORBmatcher::LastError().  Nothing aborts: the
// classes are called from the tracking, local-mapping and loop-closing threads of a long-running process.
static std::atomic<unsigned long> g_failures{0};
static int fail(const char* what, int rc) {
    g_failures.fetch_add(1, std::memory_order_relaxed);
    std::fprintf(stderr, "ORBmatcher: %s failed (%d): %s -- search reports 0 matches\n", what, rc, orb_last_error());
    return 0;
}
const char* ORBmatcher::LastError() { return orb_last_error(); }
unsigned long ORBmatcher::FailureCount() { return g_failures.load(std::memory_order_relaxed); }

// ---- per-thread device state.  The reference constructs an ORBmatcher on the stack for every use (src/Tracking.cc:1237,
// src/LocalMapping.cc, src/LoopClosing.cc), from three threads.  A matcher handle owns a HIP stream, events and scratch
// buffers: creating one per stack object costs milliseconds.  So the handle (and the BoW workspace, and a small cache of
// uploaded frames) belongs to the THREAD: every ORBmatcher object of a thread shares them, threads never share
// (re-entrant, per-thread stream + scratch).
namespace {
// identity of an uploaded frame: kind (Frame / KeyFrame), the reference's own id (Frame::mnId / KeyFrame::mnId: assigned once
// per constructed frame, copied with it, include/Frame.h, include/KeyFrame.h), the feature count and view, plus `guard`: the
// address of the keypoint array and a hash of a few sampled records -- an object that was refilled under the same id (only
// hand-built test fixtures do that) is a different key.
struct CachedFrame { int kind = -1; unsigned long id = 0; uint64_t guard = 0; int n = -1; bool cam1 = false; orbm_frame* fr = nullptr; unsigned long stamp = 0; };
struct ThreadState {
    orbm_matcher* m = nullptr;
    orbv_workspace* w = nullptr;
    static constexpr int CACHE = 8;
    CachedFrame cache[CACHE];
    unsigned long clock = 0, hits = 0, misses = 0;
    float last_us[3] = {0, 0, 0};   // host query building / frame (lookup + upload) / device search of the last projection search
    std::vector<orbm_query> q;      // query scratch of the per-frame tracking search
    std::vector<MapPoint*> qmp;
    std::vector<int> idx_a, idx_b;  // index-table scratch of flatten() and of the tracking search (never live at the same time)
    ~ThreadState() {
        for (CachedFrame& c : cache) if (c.fr) orbm_frame_destroy(c.fr);
        orbm_destroy(m);
        orbv_workspace_destroy(w);
    }
};
thread_local ThreadState tls;

inline uint64_t mix(uint64_t h, uint64_t v) { h = (h ^ v) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
uint64_t hash_bytes(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; std::memcpy(&v, b + i, 8); h = mix(h, v); }
    uint64_t tail = 0;
    if (i < n) { std::memcpy(&tail, b + i, n - i); h = mix(h, tail ^ ((uint64_t)(n - i) << 56)); }
    return h;
}
using clk = std::chrono::steady_clock;
inline float us_since(clk::time_point a) { return std::chrono::duration<float, std::micro>(clk::now() - a).count(); }

// The per-frame tracking search projects ~2000 points with `R * x + t` on 3x3 / 3x1 CV_32F cv::Mat objects: three
// reference-counted temporaries per point.  apply_rt computes the same three floats without them.  What `R * x + t` IS in
// OpenCV: ONE cv::gemm call (MatOp_GEMM::add folds the addend in), which for a 3x3 * 3x1 product without transposition
// flags takes gemm's small-matrix block (modules/core/src/matmul.cpp, `if( flags == 0 && 2 <= len && len <= 4 && ...`):
// `float t = a[0]*b[0] + a[1]*b[b_step] + a[2]*b[2*b_step]` -- FLOAT products and sums, left to right -- then
// `d = (float)(t*alpha + c*beta)` in double with alpha = beta = 1.  cv_compat.h restates that evaluation (gemm_small_f32 and the
// expression rules around it) and host/test_host `rt` holds apply_rt against the cv::Mat expression bit for bit on 10^6 random
// poses and points.  (Rounds 1-3 summed the three products in double -- the order of gemm's GENERAL path, which this shape
// never reaches)  A build against the real OpenCV keeps the cv::Mat expressions (MORB_VERBATIM_MAT_ALGEBRA is on
// by default there): whoever switches the scalar path on for such a build runs `test_host rt` against the real library first.
#if defined(HAVE_OPENCV) && !defined(MORB_SCALAR_POSE_ALGEBRA) && !defined(MORB_VERBATIM_MAT_ALGEBRA)
#define MORB_VERBATIM_MAT_ALGEBRA 1
#endif
struct Rt { float R[3][3], t[3]; };
inline Rt load_rt(const cv::Mat& R, const cv::Mat& t) {
    Rt P;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P.R[i][j] = R.at<float>(i, j); P.t[i] = t.at<float>(i); }
    return P;
}
inline void apply_rt(const Rt& P, const float* x, float* o) {
    for (int i = 0; i < 3; ++i) {
        const float t = P.R[i][0] * x[0] + P.R[i][1] * x[1] + P.R[i][2] * x[2];   // float, left to right (no contraction: -ffp-contract=off)
        o[i] = (float)((double)t * 1.0 + (double)P.t[i] * 1.0);
    }
}
}  // namespace

// test hook (host/test_host `rt`): R * x + t through apply_rt
void ORBmatcher::DebugApplyRt(const cv::Mat& R, const cv::Mat& t, const float* x, float* out) { apply_rt(load_rt(R, t), x, out); }

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    mRcam21 = cv::Mat(3, 3, CV_32F);
    mtcam21 = cv::Mat(3, 1, CV_32F);
}

ORBmatcher::~ORBmatcher() {}   // (the device state belongs to the thread, not to the object)

void ORBmatcher::LastCallBreakdown(float* us3) { for (int k = 0; k < 3; ++k) us3[k] = tls.last_us[k]; }
void ORBmatcher::FrameCacheStats(unsigned long* hits, unsigned long* misses) { *hits = tls.hits; *misses = tls.misses; }
void ORBmatcher::ResidentStats(unsigned long* served, unsigned long* missed) { resident::stats(served, missed); }

orbv_workspace* ORBmatcher::Bow() {
    if (!tls.w) {
        int rc = orbv_workspace_create(host_device(), &tls.w);
        if (rc) { tls.w = nullptr; fail("orbv_workspace_create", rc); }
    }
    return tls.w;
}

cv::Mat ORBmatcher::SkewSymmetricMatrix(const cv::Mat& v) {  // reference :4012-4017
    cv::Mat m = cv::Mat::zeros(3, 3, CV_32F);
    m.at<float>(0, 1) = -v.at<float>(2); m.at<float>(0, 2) = v.at<float>(1);
    m.at<float>(1, 0) = v.at<float>(2);  m.at<float>(1, 2) = -v.at<float>(0);
    m.at<float>(2, 0) = -v.at<float>(1); m.at<float>(2, 1) = v.at<float>(0);
    return m;
}

orbm_matcher* ORBmatcher::Handle() {
    if (!tls.m) {
        int rc = orbm_create(host_device(), &tls.m);
        if (rc) { tls.m = nullptr; fail("orbm_create", rc); }
    }
    return tls.m;
}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return orbm_descriptor_distance(a.ptr(0), b.ptr(0));
}

// Squared distance of kp2 from the epipolar line of kp1 (l = x1^T F12, float32 as in the reference) against the 95 % chi-square
// bound of one degree of freedom scaled by the octave's sigma^2 (the product is formed in double there, :183).
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF2) {
    float l[3];
    for (int j = 0; j < 3; ++j)
        l[j] = kp1.pt.x * F12.at<float>(0, j) + kp1.pt.y * F12.at<float>(1, j) + F12.at<float>(2, j);
    const float num = l[0] * kp2.pt.x + l[1] * kp2.pt.y + l[2];
    const float den = l[0] * l[0] + l[1] * l[1];
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) {  // reference :151-157
    if (viewCos > 0.998) return 2.5;
    else return 4.0;
}

void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    std::vector<int> sizes(L);
    for (int i = 0; i < L; ++i) sizes[i] = (int)histo[i].size();
    int ind[3];
    orbm_three_maxima(sizes.data(), L, ind);
    ind1 = ind[0]; ind2 = ind[1]; ind3 = ind[2];
}

namespace {

// MORB_DUMP_QUERIES=<file> (tests): appends {count, queries, index of each query's point in the caller's vector}
const char* dump_path() { static const char* p = std::getenv("MORB_DUMP_QUERIES"); return p; }
void dump_queries(const std::vector<orbm_query>& q, const std::vector<int>& src) {
    const char* path = dump_path();
    if (!path) return;
    FILE* f = std::fopen(path, "ab");
    if (!f) return;
    const int n = (int)q.size();
    std::fwrite(&n, 4, 1, f);
    if (n) { std::fwrite(q.data(), sizeof(orbm_query), n, f); std::fwrite(src.data(), 4, n, f); }
    std::fclose(f);
}

// second windows of the last dumped query set: {-count, windows}
void dump_windows(const std::vector<orbm_window>& w) {
    const char* path = dump_path();
    if (!path) return;
    FILE* f = std::fopen(path, "ab");
    if (!f) return;
    const int n = -(int)w.size();
    std::fwrite(&n, 4, 1, f);
    if (!w.empty()) std::fwrite(w.data(), sizeof(orbm_window), w.size(), f);
    std::fclose(f);
}

// One map point seen from one camera of a keyframe: what the reference's projection searches do per point between "the point in
// camera coordinates" and "GetFeaturesInArea" (src/ORBmatcher.cc:2023-2075, :2235-2285, :2650-2700, :2850-2900, :3160-3200 -- five
// copies there).  p3Dc: the point in that camera's coordinates (the caller's pose chain, which differs per search); PO: the vector the
// distance band and -- with view_angle -- the viewing-angle test are taken on (the point from the camera centre in world coordinates,
// or p3Dc itself in the Sim3 searches).  The float expressions and their ORDER are the reference's: the query's bits depend on them.
// Fills u, v, radius, the level window [level - 1, level], cam, blocks and the descriptor; ur stays NaN (*invz_out for callers with a
// stereo gate).  false: the reference's loop `continue`s.
template <class View>
bool point_query(MapPoint* pMP, const cv::Mat& p3Dc, const cv::Mat& PO, View* view, float fx, float fy, float cx, float cy, float th,
                 bool view_angle, int cam, int blocks, orbm_query& Q, float* invz_out = nullptr) {
    if (p3Dc.at<float>(2) < 0.0f) return false;
    const float invz = 1 / p3Dc.at<float>(2);   // (the reference writes 1 / z here and 1.0 / z there: the same float either way)
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    const float u = fx * x + cx;
    const float v = fy * y + cy;
    if (!view->IsInImage(u, v)) return false;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const float dist3D = cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    if (view_angle) {
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) return false;
    }
    const int nPredictedLevel = pMP->PredictScale(dist3D, view);
    Q.u = u; Q.v = v; Q.radius = th * view->mvScaleFactors[nPredictedLevel]; Q.ur = std::nanf("");
    Q.min_level = nPredictedLevel - 1; Q.max_level = nPredictedLevel;
    Q.cam = cam; Q.blocks = blocks; Q.angle = 0;
    const cv::Mat dMP = pMP->GetDescriptor();
    std::memcpy(Q.desc, dMP.ptr(0), 32);
    if (invz_out) *invz_out = invz;
    return true;
}

struct FlatFrame {  // orbm_frame_desc backing store built from a Frame
    std::vector<float> x, y, ang, ur;
    std::vector<int32_t> oct, cam, loc;
    std::vector<const uint8_t*> desc;
    orbm_frame_desc d;
};

// The reference keeps global index -> camera / local index in std::unordered_map<size_t,int> (include/Frame.h:256,261,
// include/KeyFrame.h:243,248) and reads them with one find() per access.  IndexTable reads such a map ONCE, by a single pass
// over its entries in whatever order the container keeps them (so it takes std::map and std::unordered_map alike -- a build
// against the reference's own headers and the stand-in of slam_types.h use the same code), into a flat array indexed by the
// global feature index; a feature without an entry reads -1.
struct IndexTable {
    std::vector<int> own;
    std::vector<int>& v;     // `own`, or scratch the caller keeps from call to call (no allocation per frame)
    template <class Map> IndexTable(const Map& m, size_t n) : v(own) { fill(m, n); }
    template <class Map> IndexTable(const Map& m, size_t n, std::vector<int>& scratch) : v(scratch) { fill(m, n); }
    template <class Map> void fill(const Map& m, size_t n) {
        v.assign(n, -1);
        for (const auto& e : m) if ((size_t)e.first < n) v[(size_t)e.first] = e.second;
    }
    int at(size_t g) const { return g < v.size() ? v[g] : -1; }
};

template <class FrameOrKeyFrame>
bool flatten(const FrameOrKeyFrame& F, bool cam1_only, FlatFrame& ff) {
    const int n = cam1_only ? F.N : F.N_total;
    ff.x.resize(n); ff.y.resize(n); ff.ang.resize(n); ff.ur.resize(n); ff.oct.resize(n); ff.cam.resize(n); ff.loc.resize(n);
    const std::vector<cv::KeyPoint>& kun = cam1_only ? F.mvKeysUn : F.mvKeysUn_total;
    const std::vector<float>& ur = cam1_only ? F.mvuRight : F.mvuRight_total;
    const IndexTable cams(F.keypoint_to_cam, cam1_only ? 0 : n, tls.idx_a), locs(F.cont_idx_to_local_cam_idx, cam1_only ? 0 : n, tls.idx_b);
    for (int g = 0; g < n; ++g) {
        ff.x[g] = kun[g].pt.x; ff.y[g] = kun[g].pt.y; ff.ang[g] = kun[g].angle; ff.oct[g] = kun[g].octave;
        ff.ur[g] = ur[g];
        ff.cam[g] = cam1_only ? 0 : cams.at(g);
        ff.loc[g] = cam1_only ? g : locs.at(g);
        if (ff.cam[g] < 0 || ff.loc[g] < 0) return false;   // a global index without an entry in the frame's maps
    }
    ff.desc.clear();
    if (cam1_only) ff.desc.push_back(F.mDescriptors.ptr(0));
    else for (const cv::Mat& m : F.mDescriptors_total) ff.desc.push_back(m.empty() ? nullptr : m.ptr(0));
    ff.d.n_total = n; ff.d.n_cams = (int)ff.desc.size();
    ff.d.un_x = ff.x.data(); ff.d.un_y = ff.y.data(); ff.d.octave = ff.oct.data(); ff.d.angle = ff.ang.data();
    ff.d.uright = ff.ur.data(); ff.d.cam_of = ff.cam.data(); ff.d.local_of = ff.loc.data(); ff.d.desc = ff.desc.data();
    ff.d.min_x = F.mnMinX; ff.d.min_y = F.mnMinY; ff.d.max_x = F.mnMaxX; ff.d.max_y = F.mnMaxY;
    for (int g = 0; g < n; ++g)
        if (ff.cam[g] >= ff.d.n_cams || !ff.desc[ff.cam[g]]) return false;
    return true;
}

// The matcher-side view of a Frame / KeyFrame (positions, octaves, angles, right coordinates, descriptors, 64x48 grid) in
// HBM.  A Frame's features never change after construction, and the same frame is searched several times (TrackWithMotionModel
// retries with 2*th, TrackLocalMap searches the same frame again, keyframes are fused into and searched for as long as they
// live), so uploaded frames are cached per thread under the frame's IDENTITY (see CachedFrame; rounds 1-2 hashed all ~150 KB
// an upload reads -- 87 us per call -- and trusted a 64-bit hash alone).  Least recently used of 8 entries is dropped.
// A frame that is not cached goes up through orbm_frame_create_resident: per-feature fields in one staging block, the grid
// built on the device, and the descriptor rows of every camera the calling thread has just extracted (resident.h: byte-equal
// to what the extractor handed out) read from where the describe kernel left them instead of being sent again.
// Returns NULL after a reported failure.
template <class FrameOrKeyFrame>
orbm_frame* device_frame(orbm_matcher* m, const FrameOrKeyFrame& F, bool cam1_only) {
    if (!m) return nullptr;
    const int n = cam1_only ? F.N : F.N_total;
    const std::vector<cv::KeyPoint>& kun = cam1_only ? F.mvKeysUn : F.mvKeysUn_total;
    const std::vector<float>& ur = cam1_only ? F.mvuRight : F.mvuRight_total;
    if ((int)kun.size() < n || (int)ur.size() < n) { fail("device_frame (Frame arrays shorter than N)", ORB_E_ARG); return nullptr; }
    const int kind = std::is_same<FrameOrKeyFrame, KeyFrame>::value ? 1 : 0;
    const unsigned long id = (unsigned long)F.mnId;
    // guard: everything an upload reads is either sampled or sized here -- the keypoint array's address, three records with their
    // right coordinates, the image bounds, the sizes of the two index maps, and per camera the descriptor matrix's address, row
    // count, first and last row.  MORB_FRAME_CACHE_FULL_HASH=1 hashes every byte instead (the three-thread class test
    // runs its 600 cached searches under both and expects the same outputs).
    uint64_t guard = mix(0x243F6A8885A308D3ull, (uint64_t)(uintptr_t)kun.data());
    static const bool full_hash = [] { const char* e = std::getenv("MORB_FRAME_CACHE_FULL_HASH"); return e && std::atoi(e) != 0; }();
    {
        const float bounds[4] = {(float)F.mnMinX, (float)F.mnMinY, (float)F.mnMaxX, (float)F.mnMaxY};
        guard = hash_bytes(guard, bounds, sizeof(bounds));
        guard = mix(guard, (uint64_t)F.keypoint_to_cam.size()); guard = mix(guard, (uint64_t)F.cont_idx_to_local_cam_idx.size());
    }
    if (n > 0) {
        const int probe[3] = {0, n / 2, n - 1};
        for (int k = 0; k < 3; ++k) { guard = hash_bytes(guard, &kun[probe[k]], sizeof(cv::KeyPoint)); guard = hash_bytes(guard, &ur[probe[k]], sizeof(float)); }
        auto sample = [&](const cv::Mat& d) {
            guard = mix(guard, (uint64_t)(uintptr_t)d.data); guard = mix(guard, (uint64_t)d.rows);
            if (d.empty()) return;
            if (full_hash) { for (int r = 0; r < d.rows; ++r) guard = hash_bytes(guard, d.ptr(r), 32); return; }
            guard = hash_bytes(guard, d.ptr(0), 32); guard = hash_bytes(guard, d.ptr(d.rows - 1), 32);
        };
        if (cam1_only) sample(F.mDescriptors);
        else for (const cv::Mat& d : F.mDescriptors_total) sample(d);
        if (full_hash) {
            guard = hash_bytes(guard, kun.data(), (size_t)n * sizeof(cv::KeyPoint)); guard = hash_bytes(guard, ur.data(), (size_t)n * sizeof(float));
            for (const auto& e : F.keypoint_to_cam) guard += mix(e.first, (uint64_t)e.second);                        // (order-free)
            for (const auto& e : F.cont_idx_to_local_cam_idx) guard += mix(e.first * 31 + 7, (uint64_t)e.second);
        }
    }
    ThreadState& T = tls;
    ++T.clock;
    CachedFrame* victim = &T.cache[0];
    for (CachedFrame& c : T.cache) {
        if (c.fr && c.kind == kind && c.id == id && c.guard == guard && c.n == n && c.cam1 == cam1_only) { c.stamp = T.clock; ++T.hits; return c.fr; }
        if (!c.fr) { if (victim->fr) victim = &c; }
        else if (victim->fr && c.stamp < victim->stamp) victim = &c;
    }
    ++T.misses;
    // everything that can fail on the caller's data happens before an entry is given up
    static thread_local FlatFrame ff;   // (seven arrays of n: their storage is kept from frame to frame)
    if (!flatten(F, cam1_only, ff)) {
        fail("device_frame (a feature without an entry in keypoint_to_cam / cont_idx_to_local_cam_idx, or without a descriptor row)", ORB_E_ARG);
        return nullptr;
    }
    const uint8_t* dres[4] = {nullptr, nullptr, nullptr, nullptr};
    const void* downer[4] = {nullptr, nullptr, nullptr, nullptr};   // the extractors whose device rows are served
    bool any = false;
    if (ff.d.n_cams <= 4) {
        if (cam1_only) { dres[0] = resident::find(F.mDescriptors.ptr(0), F.mDescriptors.isContinuous() ? F.mDescriptors.rows : 0, &downer[0]); any = dres[0] != nullptr; }
        else for (int c = 0; c < ff.d.n_cams; ++c) {
            const cv::Mat& d = F.mDescriptors_total[c];
            if (d.empty() || !d.isContinuous()) continue;
            dres[c] = resident::find(d.ptr(0), d.rows, &downer[c]);
            any |= dres[c] != nullptr;
        }
    }
    // a full cache gives up its least recently used entry now: its buffers are recycled once the matcher's stream has drained,
    // which it has at this point (the previous search ended with a synchronisation) and would not have right behind the new
    // frame's upload (measured: 13 us of every SearchByProjection went into waiting for the kernel just launched)
    if (victim->fr) { orbm_frame_destroy(victim->fr); victim->fr = nullptr; }
    orbm_frame* fr = nullptr;
    const int rc = orbm_frame_create_resident(m, &ff.d, any ? dres : nullptr, &fr);
    if (rc) { fail("orbm_frame_create_resident", rc); return nullptr; }
    // (the build kernel reads those extractors' rows: the next run of each orders itself behind this stream)
    for (int c = 0; c < 4; ++c) if (dres[c] && downer[c]) resident::note_reader(downer[c], orbm_stream(m));
    victim->kind = kind; victim->id = id; victim->guard = guard; victim->n = n; victim->cam1 = cam1_only; victim->fr = fr; victim->stamp = T.clock;
    return fr;
}

struct FlatSide {  // orbv_side backing store built from a Frame / KeyFrame
    std::vector<uint8_t> desc, flags;
    std::vector<float> ang, x, y;
    std::vector<int32_t> oct, cam, nstart;
    std::vector<uint32_t> nid, items;
    orbv_side s;
};

template <class F>   // Frame or KeyFrame: same member names
void flatten_side(const F& f, const std::vector<cv::KeyPoint>& keys, const DBoW2::FeatureVector& fv, int n, FlatSide& o) {
    o.desc.resize((size_t)n * 32); o.flags.assign(n, 0); o.ang.resize(n); o.x.resize(n); o.y.resize(n); o.oct.resize(n); o.cam.resize(n);
    for (int g = 0; g < n; ++g) {
        const int cam = f.keypoint_to_cam.find(g)->second, loc = f.cont_idx_to_local_cam_idx.find(g)->second;
        std::memcpy(&o.desc[(size_t)g * 32], f.mDescriptors_total[cam].ptr(loc), 32);
        o.ang[g] = keys[g].angle; o.x[g] = keys[g].pt.x; o.y[g] = keys[g].pt.y; o.oct[g] = keys[g].octave; o.cam[g] = cam;
    }
    o.nid.clear(); o.nstart.assign(1, 0); o.items.clear();
    for (const auto& e : fv) {
        o.nid.push_back(e.first);
        o.items.insert(o.items.end(), e.second.begin(), e.second.end());
        o.nstart.push_back((int32_t)o.items.size());
    }
    o.s.n = n; o.s.desc = o.desc.data(); o.s.angle = o.ang.data(); o.s.flags = o.flags.data();
    o.s.n_nodes = (int)o.nid.size(); o.s.node_id = o.nid.data(); o.s.node_start = o.nstart.data(); o.s.items = o.items.data();
    o.s.x = o.x.data(); o.s.y = o.y.data(); o.s.octave = o.oct.data(); o.s.cam_of = o.cam.data();
}

// camera-1 view of a Frame / KeyFrame (mDescriptors, mvKeys / mvKeysUn, mFeatVec_cam1) for the _cam1 overloads
void flatten_side_cam1(const cv::Mat& desc, const std::vector<cv::KeyPoint>& keys, const DBoW2::FeatureVector& fv, int n, FlatSide& o) {
    o.desc.resize((size_t)n * 32); o.flags.assign(n, 0); o.ang.resize(n);
    for (int g = 0; g < n; ++g) { std::memcpy(&o.desc[(size_t)g * 32], desc.ptr(g), 32); o.ang[g] = keys[g].angle; }
    o.nid.clear(); o.nstart.assign(1, 0); o.items.clear();
    for (const auto& e : fv) {
        o.nid.push_back(e.first);
        for (unsigned int idx : e.second) if ((int)idx < n) o.items.push_back(idx);   // `if (realIdx >= N) continue` (:431, :449)
        o.nstart.push_back((int32_t)o.items.size());
    }
    std::memset(&o.s, 0, sizeof(o.s));
    o.s.n = n; o.s.desc = o.desc.data(); o.s.angle = o.ang.data(); o.s.flags = o.flags.data();
    o.s.n_nodes = (int)o.nid.size(); o.s.node_id = o.nid.data(); o.s.node_start = o.nstart.data(); o.s.items = o.items.data();
}

}  // namespace

// reference src/ORBmatcher.cc:390-565: the camera-1 form Tracking::Relocalization calls (src/Tracking.cc:2011)
int ORBmatcher::SearchByBoW_cam1(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches_cam1();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    FlatSide a, b;
    flatten_side_cam1(pKF->mDescriptors, pKF->mvKeysUn, pKF->mFeatVec_cam1, pKF->N, a);
    flatten_side_cam1(F.mDescriptors, F.mvKeys, F.mFeatVec_cam1, F.N, b);
    for (int i = 0; i < pKF->N; ++i) a.flags[i] = (vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad()) ? 1 : 0;
    b.s.flags = nullptr;
    std::vector<int32_t> match(F.N > 0 ? F.N : 1);
    int nmatches = 0;
    const int rc = orbv_search_by_bow(Bow(), &a.s, &b.s, 0, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbv_search_by_bow", rc);
    for (int g = 0; g < F.N; ++g)
        if (match[g] >= 0) vpMapPointMatches[g] = vpMapPointsKF[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:1180-1363: the camera-1 form LoopClosing::ComputeSim3 calls (src/LoopClosing.cc:362)
int ORBmatcher::SearchByBoW_cam1(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches_cam1();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches_cam1();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    FlatSide a, b;
    flatten_side_cam1(pKF1->mDescriptors, pKF1->mvKeysUn, pKF1->mFeatVec_cam1, pKF1->N, a);
    flatten_side_cam1(pKF2->mDescriptors, pKF2->mvKeysUn, pKF2->mFeatVec_cam1, pKF2->N, b);
    for (int i = 0; i < pKF1->N; ++i) a.flags[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;
    for (int i = 0; i < pKF2->N; ++i) b.flags[i] = (vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0;
    std::vector<int32_t> match(pKF1->N > 0 ? pKF1->N : 1);
    int nmatches = 0;
    const int rc = orbv_search_by_bow(Bow(), &a.s, &b.s, 1, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbv_search_by_bow", rc);
    for (int i = 0; i < pKF1->N; ++i)
        if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];
    return nmatches;
}

// reference src/ORBmatcher.cc:3809-3946 (relocalisation).  Candidates come from the camera-1 grid, no right-coordinate
// gate, any MapPoint already in the frame hides its feature, every accepted match does too.
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist) {
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches_cam1();
    std::vector<orbm_query> q;
    std::vector<MapPoint*> qmp;
    std::vector<int> qsrc;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        MapPoint* pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D = cv::norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        orbm_query Q;
        Q.u = u; Q.v = v; Q.radius = radius; Q.ur = std::nanf("");
        Q.min_level = nPredictedLevel - 1; Q.max_level = nPredictedLevel + 1;
        Q.cam = 0; Q.blocks = 1;
        Q.angle = pKF->mvKeysUn[i].angle;
        const cv::Mat dMP = pMP->GetDescriptor();
        std::memcpy(Q.desc, dMP.ptr(0), 32);
        q.push_back(Q); qmp.push_back(pMP); qsrc.push_back((int)i);
    }
    dump_queries(q, qsrc);
    orbm_frame* fr = device_frame(Handle(), CurrentFrame, true);
    if (!fr) return 0;
    std::vector<uint8_t> occupied(CurrentFrame.N > 0 ? CurrentFrame.N : 1, 0);
    for (int g = 0; g < CurrentFrame.N; ++g) occupied[g] = CurrentFrame.mvpMapPoints[g] ? 1 : 0;   // :3881
    int rc;
    std::vector<int32_t> match(occupied.size());
    int nmatches = 0;
    rc = orbm_search_by_projection(Handle(), fr, q.data(), (int)q.size(), occupied.data(), ORBdist, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbm_search_by_projection", rc);
    for (int g = 0; g < CurrentFrame.N; ++g) {
        if (match[g] >= 0) CurrentFrame.mvpMapPoints[g] = qmp[match[g]];
        else if (match[g] == -2) CurrentFrame.mvpMapPoints[g] = NULL;   // :3936
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:753-867 (loop closing, camera 1)
int ORBmatcher::SearchByProjection_cam1(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {
    const float& fx = pKF->fx; const float& fy = pKF->fy; const float& cx = pKF->cx; const float& cy = pKF->cy;
    // Decompose Scw
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    std::vector<orbm_query> q;
    std::vector<MapPoint*> qmp;
    std::vector<int> qsrc;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        orbm_query Q;
        if (!point_query(pMP, p3Dc, p3Dw - Ow, pKF, fx, fy, cx, cy, (float)th, /*view_angle=*/true, 0, 1, Q)) continue;   // (window [level - 1, level], :833-836)
        q.push_back(Q); qmp.push_back(pMP); qsrc.push_back(iMP);
    }
    dump_queries(q, qsrc);
    orbm_frame* fr = device_frame(Handle(), *pKF, true);
    if (!fr) return 0;
    std::vector<uint8_t> occupied(pKF->N > 0 ? pKF->N : 1, 0);
    for (int g = 0; g < pKF->N; ++g) occupied[g] = vpMatched[g] ? 1 : 0;    // :829
    int rc;
    std::vector<int32_t> match(occupied.size());
    int nmatches = 0;
    rc = orbm_search_by_projection(Handle(), fr, q.data(), (int)q.size(), occupied.data(), TH_LOW, 0, match.data(), &nmatches);
    if (rc) return fail("orbm_search_by_projection", rc);
    for (int g = 0; g < pKF->N; ++g)
        if (match[g] >= 0) vpMatched[g] = qmp[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:566-750 (loop closing, both cameras of the keyframe).  Every loop point is projected into camera 1
// and, through the cam2 <- cam1 extrinsics, into camera 2; ONE `dist < bestDist` chain runs over the candidates of both
// windows (camera 1's first), and an accepted match (<= TH_LOW) takes its feature out of the game for the points that follow
// (vpMatched[idx]).  On the device: one query per point with a second window (orbm_search_by_projection_windows).
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<int>& /*vLoopMPCams*/,
                                   std::vector<MapPoint*>& vpMatched, int th, const cv::Mat CalibMatrix) {
    const float& fx = pKF->fx; const float& fy = pKF->fy; const float& cx = pKF->cx; const float& cy = pKF->cy;
    const cv::Mat Rcam12 = CalibMatrix.rowRange(0, 3).colRange(0, 3);
    cv::Mat tcam12(3, 1, CV_32F);
    tcam12.at<float>(0, 0) = CalibMatrix.at<float>(3, 0);
    tcam12.at<float>(1, 0) = CalibMatrix.at<float>(3, 1);
    tcam12.at<float>(2, 0) = CalibMatrix.at<float>(3, 2);
    const cv::Mat Rcam21 = Rcam12.inv();
    const cv::Mat tcam21 = -Rcam21 * tcam12;
    // Decompose Scw
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    std::vector<orbm_query> q;
    std::vector<orbm_window> w2;
    std::vector<MapPoint*> qmp;
    std::vector<int> qsrc;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        orbm_query Q;
        orbm_window W[2];
        W[0].cam = W[1].cam = -1;
        for (int camidx = 0; camidx < 2; ++camidx) {
            cv::Mat p3Dc = Rcw * p3Dw + tcw;
            if (camidx == 1) p3Dc = Rcam21 * p3Dc + tcam21;
            if (p3Dc.at<float>(2) < 0.0) continue;
            const float invz = 1 / p3Dc.at<float>(2);
            const float x = p3Dc.at<float>(0) * invz;
            const float y = p3Dc.at<float>(1) * invz;
            const float u = fx * x + cx;
            const float v = fy * y + cy;
            if (!pKF->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            cv::Mat PO = p3Dw - Ow;
            const float dist = cv::norm(PO);
            if (dist < minDistance || dist > maxDistance) continue;
            cv::Mat Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist) continue;
            int nPredictedLevel = pMP->PredictScale(dist, pKF);
            const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
            W[camidx].u = u; W[camidx].v = v; W[camidx].radius = radius; W[camidx].cam = camidx;
            W[camidx].min_level = nPredictedLevel - 1; W[camidx].max_level = nPredictedLevel;   // :704-707
        }
        if (W[0].cam < 0 && W[1].cam < 0) continue;   // (nothing to search: bestDist stays 256)
        Q.u = W[0].u; Q.v = W[0].v; Q.radius = W[0].radius; Q.ur = std::nanf("");
        Q.min_level = W[0].min_level; Q.max_level = W[0].max_level; Q.cam = W[0].cam;
        if (W[0].cam < 0) { Q.u = Q.v = Q.radius = 0.f; Q.min_level = Q.max_level = -1; }
        Q.blocks = 1; Q.angle = 0;
        const cv::Mat dMP = pMP->GetDescriptor();
        std::memcpy(Q.desc, dMP.ptr(0), 32);
        q.push_back(Q); w2.push_back(W[1]); qmp.push_back(pMP); qsrc.push_back(iMP);
    }
    dump_queries(q, qsrc);
    dump_windows(w2);
    orbm_frame* fr = device_frame(Handle(), *pKF, false);
    if (!fr) return 0;
    const int n = pKF->N_total;
    std::vector<uint8_t> occupied(n > 0 ? n : 1, 0);
    for (int g = 0; g < n; ++g) occupied[g] = vpMatched[g] ? 1 : 0;    // :696
    std::vector<int32_t> match(occupied.size());
    int nmatches = 0;
    const int rc = orbm_search_by_projection_windows(Handle(), fr, q.data(), w2.data(), (int)q.size(), occupied.data(), TH_LOW, 0, match.data(),
                                                     &nmatches);
    if (rc) return fail("orbm_search_by_projection_windows", rc);
    for (int g = 0; g < n; ++g)
        if (match[g] >= 0) vpMatched[g] = qmp[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:868-994 (monocular initialisation; not reached by the RGB-D rig).  The windows around the
// previously matched positions are gathered on the device in the reference's visiting order with their distances
// (orbm_project_candidates: level-0 features of F2's camera-1 grid); the loop that follows mutates per-feature state between
// queries (vMatchedDistance, vnMatches21 with un-matching) and is replayed on the host literally.
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                        int windowSize) {
    int nmatches = 0;
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(F2.mvKeysUn.size(), INT_MAX);
    std::vector<int> vnMatches21(F2.mvKeysUn.size(), -1);
    std::vector<orbm_query> q;
    std::vector<int> qsrc;
    for (size_t i1 = 0, iend1 = F1.mvKeysUn.size(); i1 < iend1; i1++) {
        cv::KeyPoint kp1 = F1.mvKeysUn[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        orbm_query Q;
        Q.u = vbPrevMatched[i1].x; Q.v = vbPrevMatched[i1].y; Q.radius = (float)windowSize; Q.ur = std::nanf("");
        Q.min_level = level1; Q.max_level = level1; Q.cam = 0; Q.blocks = 0; Q.angle = kp1.angle;
        std::memcpy(Q.desc, F1.mDescriptors.ptr((int)i1), 32);
        q.push_back(Q); qsrc.push_back((int)i1);
    }
    dump_queries(q, qsrc);
    const int nq = (int)q.size();
    std::vector<int32_t> cidx, ccnt(nq > 0 ? nq : 1);
    std::vector<uint16_t> cdist;
    int cap = 64;
    if (nq) {
        orbm_frame* fr = device_frame(Handle(), F2, true);
        if (!fr) return 0;
        for (;;) {
            cidx.resize((size_t)nq * cap); cdist.resize((size_t)nq * cap);
            const int rc = orbm_project_candidates(Handle(), fr, q.data(), nq, cap, cidx.data(), cdist.data(), ccnt.data());
            if (rc == ORB_OK) break;
            if (rc != ORB_E_CAPACITY) return fail("orbm_project_candidates", rc);
            int mx = cap;
            for (int i = 0; i < nq; ++i) mx = std::max(mx, (int)ccnt[i]);
            cap = (mx + 63) & ~63;
        }
    }
    for (int k = 0; k < nq; ++k) {
        const size_t i1 = (size_t)qsrc[k];
        if (ccnt[k] == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < ccnt[k]; ++c) {
            const size_t i2 = (size_t)cidx[(size_t)k * cap + c];
            const int dist = cdist[(size_t)k * cap + c];
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = F1.mvKeysUn[i1].angle - F2.mvKeysUn[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    assert(bin >= 0 && bin < HISTO_LENGTH);
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    // Update prev matched
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
}

// reference src/ORBmatcher.cc:3137-3447 (camera 1 only; what LoopClosing::ComputeSim3 calls, src/LoopClosing.cc:444) and
// :2814-3135 (both cameras: a point is searched in the grid of the camera it was observed in, after the cam2 <- cam1
// extrinsics of the 4x3 calibration matrix).  Each keyframe's map points are projected into the other one through the Sim3,
// every point takes the nearest descriptor of its window on its own (orbm_project_best), then the mutual-agreement check.
static int sim3_search(ORBmatcher& self, orbm_matcher* handle, KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12,
                       const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th, const cv::Mat* CalibMatrix) {
    const bool two_cam = CalibMatrix != nullptr;
    const float& fx = pKF1->fx; const float& fy = pKF1->fy; const float& cx = pKF1->cx; const float& cy = pKF1->cy;
    cv::Mat Rcam21, tcam21;
    if (two_cam) {   // :2826-2836
        const cv::Mat Rcam12 = CalibMatrix->rowRange(0, 3).colRange(0, 3);
        cv::Mat tcam12(3, 1, CV_32F);
        tcam12.at<float>(0, 0) = CalibMatrix->at<float>(3, 0);
        tcam12.at<float>(1, 0) = CalibMatrix->at<float>(3, 1);
        tcam12.at<float>(2, 0) = CalibMatrix->at<float>(3, 2);
        Rcam21 = Rcam12.inv();
        tcam21 = -Rcam21 * tcam12;
    }
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint*> vpMapPoints1 = two_cam ? pKF1->GetMapPointMatches() : pKF1->GetMapPointMatches_cam1();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = two_cam ? pKF2->GetMapPointMatches() : pKF2->GetMapPointMatches_cam1();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = two_cam ? pMP->GetIndexInKeyFrame(pKF2) : pMP->GetIndexInKeyFrame_cam1(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    bool failed = false;
    // one direction: map points of `from` into `to`
    auto direction = [&](KeyFrame* from, KeyFrame* to, const std::vector<MapPoint*>& pts, const std::vector<bool>& already, const cv::Mat& Rw,
                         const cv::Mat& tw, const cv::Mat& sR, const cv::Mat& t, std::vector<int>& out) {
        std::vector<orbm_query> q; std::vector<int> src;
        const IndexTable cams(from->keypoint_to_cam, two_cam ? pts.size() : 0);
        for (int i = 0; i < (int)pts.size(); i++) {
            MapPoint* pMP = pts[i];
            if (!pMP || already[i]) continue;
            if (pMP->isBad()) continue;
            const int camIdx = two_cam ? cams.at((size_t)i) : 0;   // the camera the point was observed in (:2893, :2995)
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Da = Rw * p3Dw + tw;
            cv::Mat p3Db = sR * p3Da + t;
            if (camIdx == 1) p3Db = Rcam21 * p3Db + tcam21;
            orbm_query Q;
            if (!point_query(pMP, p3Db, p3Db, to, fx, fy, cx, cy, (float)th, /*view_angle=*/false, camIdx, 0, Q)) continue;
            q.push_back(Q); src.push_back(i);
        }
        dump_queries(q, src);
        if (q.empty() || failed) return;
        orbm_frame* fr = device_frame(handle, *to, /*cam1_only=*/!two_cam);
        if (!fr) { failed = true; return; }
        std::vector<int32_t> bi(q.size()), bd(q.size());
        const int rc = orbm_project_best(handle, fr, q.data(), (int)q.size(), nullptr, ORBM_GATE_NONE, nullptr, 0, bi.data(), bd.data());
        if (rc) { fail("orbm_project_best", rc); failed = true; return; }
        for (size_t k = 0; k < q.size(); ++k)
            if (bi[k] >= 0 && bd[k] <= ORBmatcher::TH_HIGH) out[src[k]] = bi[k];
    };
    direction(pKF1, pKF2, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, vnMatch1);   // :3208-3302 | :2877-2974
    direction(pKF2, pKF1, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, vnMatch2);   // :3306-3403 | :2978-3080
    if (failed) return 0;
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
    }
    return nFound;
}

int ORBmatcher::SearchBySim3_cam1(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                                  const cv::Mat& t12, const float th) {
    return sim3_search(*this, Handle(), pKF1, pKF2, vpMatches12, s12, R12, t12, th, nullptr);
}

int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, const float th, const cv::Mat CalibMatrix) {
    return sim3_search(*this, Handle(), pKF1, pKF2, vpMatches12, s12, R12, t12, th, &CalibMatrix);
}

// reference src/ORBmatcher.cc:1986-2210: every map point projected into both cameras of the keyframe, nearest descriptor
// under the reprojection-error gate (orbm_project_best, ORBM_GATE_CHI2); the merge / replace bookkeeping stays host code
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const cv::Mat CalibMatrix, const float th) {
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float& fx = pKF->fx; const float& fy = pKF->fy; const float& cx = pKF->cx; const float& cy = pKF->cy; const float& bf = pKF->mbf;
    const cv::Mat Rcam12 = CalibMatrix.rowRange(0, 3).colRange(0, 3);
    cv::Mat tcam12(3, 1, CV_32F);
    tcam12.at<float>(0, 0) = CalibMatrix.at<float>(3, 0);
    tcam12.at<float>(1, 0) = CalibMatrix.at<float>(3, 1);
    tcam12.at<float>(2, 0) = CalibMatrix.at<float>(3, 2);
    const cv::Mat Rcam21 = Rcam12.inv();
    const cv::Mat tcam21 = -Rcam21 * tcam12;
    cv::Mat Ow[2] = {pKF->GetCameraCenter(), pKF->GetCameraCenter_cam2()};
    const int nMPs = (int)vpMapPoints.size();
    std::vector<orbm_query> q; std::vector<int> src;
    // IsInKeyFrame(pKF) changes while the reference's loop runs (AddObservation below): it is re-checked in the merge pass
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc;
        for (int cam = 0; cam < 2; cam++) {
            if (cam == 0) p3Dc = Rcw * p3Dw + tcw;
            else p3Dc = Rcam21 * Rcw * p3Dw + Rcam21 * tcw + tcam21;
            orbm_query Q; float invz = 0.f;
            if (!point_query(pMP, p3Dc, p3Dw - Ow[cam], pKF, fx, fy, cx, cy, (float)th, /*view_angle=*/true, cam, 0, Q, &invz)) continue;
            Q.ur = Q.u - bf * invz;
            q.push_back(Q); src.push_back(i);
        }
    }
    dump_queries(q, src);
    std::vector<int32_t> bi(q.size() ? q.size() : 1, -1), bd(q.size() ? q.size() : 1, 256);
    if (!q.empty()) {
        orbm_frame* fr = device_frame(Handle(), *pKF, false);
        if (!fr) return 0;
        int rc;
        rc = orbm_project_best(Handle(), fr, q.data(), (int)q.size(), nullptr, ORBM_GATE_CHI2, pKF->mvInvLevelSigma2.data(),
                               (int)pKF->mvInvLevelSigma2.size(), bi.data(), bd.data());
        if (rc) return fail("orbm_project_best", rc);
    }
    // merge pass in the reference's order: point by point, camera 1 then camera 2 (:2165-2195)
    int nFused = 0;
    size_t k = 0;
    for (int i = 0; i < nMPs; i++) {
        const size_t k0 = k;
        while (k < q.size() && src[k] == i) ++k;
        MapPoint* pMP = vpMapPoints[i];
        if (k0 == k || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;   // :2020-2023 (Replace / AddObservation earlier in this call count)
        for (size_t f = k0; f < k; ++f) {
            if (bi[f] < 0 || bd[f] > TH_LOW) continue;
            MapPoint* pMPinKF = pKF->GetMapPoint(bi[f]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bi[f]);
                pKF->AddMapPoint(pMP, bi[f]);
            }
            nFused++;
        }
    }
    return nFused;
}

// reference src/ORBmatcher.cc:2211-2517: as above through a Sim3 pose, no reprojection-error gate, duplicates reported in
// vpReplacePoint instead of being replaced on the spot (vLoopMPCams is not read there either)
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<int>& vLoopMPCams, float th,
                     std::vector<MapPoint*>& vpReplacePoint, const cv::Mat CalibMatrix) {
    const float& fx = pKF->fx; const float& fy = pKF->fy; const float& cx = pKF->cx; const float& cy = pKF->cy;
    const cv::Mat Rcam12 = CalibMatrix.rowRange(0, 3).colRange(0, 3);
    cv::Mat tcam12(3, 1, CV_32F);
    tcam12.at<float>(0, 0) = CalibMatrix.at<float>(3, 0);
    tcam12.at<float>(1, 0) = CalibMatrix.at<float>(3, 1);
    tcam12.at<float>(2, 0) = CalibMatrix.at<float>(3, 2);
    const cv::Mat Rcam21 = Rcam12.inv();
    const cv::Mat tcam21 = -Rcam21 * tcam12;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    std::vector<orbm_query> q; std::vector<int> src;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        for (int camidx = 0; camidx < 2; ++camidx) {
            cv::Mat p3Dc;
            if (camidx == 1) p3Dc = Rcam21 * Rcw * p3Dw + Rcam21 * tcw + tcam21;
            else p3Dc = Rcw * p3Dw + tcw;
            cv::Mat PO = p3Dw - Ow;
            if (camidx == 1) PO = PO - Rcw.t() * tcam12;
            orbm_query Q;
            if (!point_query(pMP, p3Dc, PO, pKF, fx, fy, cx, cy, (float)th, /*view_angle=*/true, camidx, 0, Q)) continue;
            q.push_back(Q); src.push_back(iMP);
        }
    }
    dump_queries(q, src);
    std::vector<int32_t> bi(q.size() ? q.size() : 1, -1), bd(q.size() ? q.size() : 1, 256);
    if (!q.empty()) {
        orbm_frame* fr = device_frame(Handle(), *pKF, false);
        if (!fr) return 0;
        int rc;
        rc = orbm_project_best(Handle(), fr, q.data(), (int)q.size(), nullptr, ORBM_GATE_NONE, nullptr, 0, bi.data(), bd.data());
        if (rc) return fail("orbm_project_best", rc);
    }
    int nFused = 0;
    for (size_t f = 0; f < q.size(); ++f) {            // point by point, camera 1 then camera 2 (:2484-2506)
        if (bi[f] < 0 || bd[f] > TH_LOW) continue;
        const int iMP = src[f];
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bi[f]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bi[f]);
            pKF->AddMapPoint(pMP, bi[f]);
        }
        nFused++;
    }
    return nFused;
}

// reference src/ORBmatcher.cc:2518-2813: the camera-1 form of the Sim3 fuse (defined in the reference; its only call, at
// src/LoopClosing.cc:842, is commented out there).  The points are projected into camera 1 only and searched in the keyframe's
// camera-1 grid (KeyFrame::GetFeaturesInArea(u, v, r), mvKeysUn, mDescriptors); no reprojection-error gate; the nearest
// descriptor of the window wins on its own (orbm_project_best), duplicates are reported in vpReplacePoint.
int ORBmatcher::Fuse_cam1(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
    const float& fx = pKF->fx; const float& fy = pKF->fy; const float& cx = pKF->cx; const float& cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    std::vector<orbm_query> q; std::vector<int> src;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        orbm_query Q;
        if (!point_query(pMP, p3Dc, p3Dw - Ow, pKF, fx, fy, cx, cy, (float)th, /*view_angle=*/true, 0, 0, Q)) continue;   // (:2660-2663)
        q.push_back(Q); src.push_back(iMP);
    }
    dump_queries(q, src);
    std::vector<int32_t> bi(q.size() ? q.size() : 1, -1), bd(q.size() ? q.size() : 1, 256);
    if (!q.empty()) {
        orbm_frame* fr = device_frame(Handle(), *pKF, true);
        if (!fr) return 0;
        const int rc = orbm_project_best(Handle(), fr, q.data(), (int)q.size(), nullptr, ORBM_GATE_NONE, nullptr, 0, bi.data(), bd.data());
        if (rc) return fail("orbm_project_best", rc);
    }
    int nFused = 0;
    for (size_t f = 0; f < q.size(); ++f) {            // in point order: an added point is what a later duplicate finds (:2700-2713)
        if (bi[f] < 0 || bd[f] > TH_LOW) continue;
        const int iMP = src[f];
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bi[f]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bi[f]);
            pKF->AddMapPoint(pMP, bi[f]);
        }
        nFused++;
    }
    return nFused;
}

// reference src/ORBmatcher.cc:206-388
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N_total, static_cast<MapPoint*>(NULL));
    FlatSide a, b;
    flatten_side(*pKF, pKF->mvKeysUn_total, pKF->mFeatVec, (int)vpMapPointsKF.size(), a);
    flatten_side(F, F.mvKeys_total, F.mFeatVec, F.N_total, b);
    for (size_t i = 0; i < vpMapPointsKF.size(); ++i) a.flags[i] = (vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad()) ? 1 : 0;   // :259-264
    b.s.flags = nullptr;
    std::vector<int32_t> match(F.N_total > 0 ? F.N_total : 1);
    int nmatches = 0;
    const int rc = orbv_search_by_bow(Bow(), &a.s, &b.s, 0, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbv_search_by_bow", rc);
    for (int g = 0; g < F.N_total; ++g)
        if (match[g] >= 0) vpMapPointMatches[g] = vpMapPointsKF[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:996-1165
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    FlatSide a, b;
    flatten_side(*pKF1, pKF1->mvKeysUn_total, pKF1->mFeatVec, (int)vpMapPoints1.size(), a);
    flatten_side(*pKF2, pKF2->mvKeysUn_total, pKF2->mFeatVec, (int)vpMapPoints2.size(), b);
    for (size_t i = 0; i < vpMapPoints1.size(); ++i) a.flags[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;   // :1050-1055
    for (size_t i = 0; i < vpMapPoints2.size(); ++i) b.flags[i] = (vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0;   // :1077-1084
    std::vector<int32_t> match(vpMapPoints1.empty() ? 1 : vpMapPoints1.size());
    int nmatches = 0;
    const int rc = orbv_search_by_bow(Bow(), &a.s, &b.s, 1, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbv_search_by_bow", rc);
    for (size_t i = 0; i < vpMapPoints1.size(); ++i)
        if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];
    return nmatches;
}

// reference src/ORBmatcher.cc:1364-1786.  As there, the F12 argument is not read: one fundamental matrix per camera of
// the rig is recomputed from the keyframe poses (:1375-1423).
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo, std::vector<bool> vbCam) {
    cv::Mat R1w[2] = {pKF1->GetRotation(), pKF1->GetRotation_cam2()}, t1w[2] = {pKF1->GetTranslation(), pKF1->GetTranslation_cam2()};
    cv::Mat R2w[2] = {pKF2->GetRotation(), pKF2->GetRotation_cam2()}, t2w[2] = {pKF2->GetTranslation(), pKF2->GetTranslation_cam2()};
    cv::Mat Cw[2] = {pKF1->GetCameraCenter(), pKF1->GetCameraCenter_cam2()};
    const cv::Mat& K1 = pKF1->mK;
    const cv::Mat& K2 = pKF2->mK;
    orbv_triangulation T;
    std::memset(&T, 0, sizeof(T));
    T.n_cams = 2; T.n_levels = (int)pKF2->mvScaleFactors.size();
    T.scale_factors = pKF2->mvScaleFactors.data(); T.level_sigma2 = pKF2->mvLevelSigma2.data();
    for (int i = 0; i < 2; ++i) {
        const cv::Mat R12 = R1w[i] * R2w[i].t();
        const cv::Mat t12 = -R1w[i] * R2w[i].t() * t2w[i] + t1w[i];
        const cv::Mat t12x = SkewSymmetricMatrix(t12);
        const cv::Mat F = K1.t().inv() * t12x * R12 * K2.inv();
        for (int k = 0; k < 9; ++k) T.F12[i][k] = F.at<float>(k / 3, k % 3);
        const cv::Mat C2 = R2w[i] * Cw[i] + t2w[i];                 // :1441-1452
        const float invz = 1.0f / C2.at<float>(2);
        T.ex[i] = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
        T.ey[i] = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    }
    const int n1 = (int)pKF1->mvKeysUn_total.size(), n2 = (int)pKF2->mvKeysUn_total.size();
    FlatSide a, b;
    flatten_side(*pKF1, pKF1->mvKeysUn_total, pKF1->mFeatVec, n1, a);
    flatten_side(*pKF2, pKF2->mvKeysUn_total, pKF2->mFeatVec, n2, b);
    for (int i = 0; i < n1; ++i) {
        const bool stereo = pKF1->mvuRight_total[i] >= 0;
        const bool usable = !pKF1->GetMapPoint(i) && vbCam[a.cam[i]] && (!bOnlyStereo || stereo);   // :1490-1505
        a.flags[i] = (usable ? 1 : 0) | (stereo ? 2 : 0);
    }
    for (int i = 0; i < n2; ++i) {
        const bool stereo = pKF2->mvuRight_total[i] >= 0;
        const bool usable = !pKF2->GetMapPoint(i) && (!bOnlyStereo || stereo);                        // :1548-1575
        b.flags[i] = (usable ? 1 : 0) | (stereo ? 2 : 0);
    }
    std::vector<int32_t> match(n1 > 0 ? n1 : 1);
    int nmatches = 0;
    const int rc = orbv_search_for_triangulation(Bow(), &a.s, &b.s, &T, TH_LOW, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbv_search_for_triangulation", rc);
    vMatchedPairs.clear();
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; ++i)
        if (match[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[i]));
    return nmatches;
}

// reference src/ORBmatcher.cc:62-149
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    const bool bFactor = th != 1.0;
    std::vector<orbm_query> q;
    std::vector<MapPoint*> qmp;
    q.reserve(vpMapPoints.size());
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;
        if (pMP->isBad()) continue;
        const int& nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);
        if (bFactor) r *= th;
        orbm_query Q;
        Q.u = pMP->mTrackProjX; Q.v = pMP->mTrackProjY;
        Q.radius = r * F.mvScaleFactors[nPredictedLevel];
        Q.ur = pMP->mTrackProjXR;
        Q.min_level = nPredictedLevel - 1; Q.max_level = nPredictedLevel;
        Q.cam = 0;
        Q.blocks = pMP->Observations() > 0 ? 1 : 0;
        Q.angle = 0;
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(Q.desc, d.ptr(0), 32);
        q.push_back(Q); qmp.push_back(pMP);
    }
    orbm_frame* fr = device_frame(Handle(), F, true);
    if (!fr) return 0;
    std::vector<uint8_t> occupied(F.N, 0);
    for (int g = 0; g < F.N; ++g)
        occupied[g] = (F.mvpMapPoints[g] && F.mvpMapPoints[g]->Observations() > 0) ? 1 : 0;  // :107-109
    int rc;
    std::vector<int32_t> match(F.N > 0 ? F.N : 1);
    int nmatches = 0;
    rc = orbm_search_by_projection_points(Handle(), fr, q.data(), (int)q.size(), occupied.data(), mfNNratio, TH_HIGH,
                                          match.data(), &nmatches);
    if (rc) return fail("orbm_search_by_projection_points", rc);
    for (int g = 0; g < F.N; ++g)
        if (match[g] >= 0) F.mvpMapPoints[g] = qmp[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:3448-3641
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                                   cv::Mat CalibMatrix) {
    const clk::time_point t_entry = clk::now();
    // cam2 <- cam1 extrinsics from the 4x3 calibration matrix (:3463-3471)
    cv::Mat Rcam12 = CalibMatrix.rowRange(0, 3).colRange(0, 3);
    cv::Mat tcam12(3, 1, CV_32F);
    tcam12.at<float>(0, 0) = CalibMatrix.at<float>(3, 0);
    tcam12.at<float>(1, 0) = CalibMatrix.at<float>(3, 1);
    tcam12.at<float>(2, 0) = CalibMatrix.at<float>(3, 2);
    mRcam21 = Rcam12.t();
    mtcam21 = -mRcam21 * tcam12;

    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;

    bool bForward[2], bBackward[2];
    bForward[0] = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    bForward[1] = tlc.at<float>(0) > CurrentFrame.mb && !bMono;
    bBackward[0] = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    bBackward[1] = -tlc.at<float>(0) > CurrentFrame.mb && !bMono;

    // The current frame's matcher view is needed by the search only: its upload (one staging write + one unpack kernel,
    // asynchronous) is started first, so that it is in flight while the host projects the points.
    const clk::time_point t_frame = clk::now();
    orbm_frame* fr = device_frame(Handle(), CurrentFrame, false);
    if (!fr) return 0;
    tls.last_us[1] = us_since(t_frame);
    const clk::time_point t_query = clk::now();

    std::vector<orbm_query>& q = tls.q;          // (per-thread scratch: 88 bytes per point, reused from call to call)
    std::vector<MapPoint*>& qmp = tls.qmp;
    q.clear(); qmp.clear();
    q.reserve(LastFrame.N_total); qmp.reserve(LastFrame.N_total);
#ifndef MORB_VERBATIM_MAT_ALGEBRA
    const Rt Pcw = load_rt(Rcw, tcw), Pcam21 = load_rt(mRcam21, mtcam21);
#endif
    const IndexTable cam_of(LastFrame.keypoint_to_cam, LastFrame.N_total > 0 ? LastFrame.N_total : 0, tls.idx_a);   // (device_frame above is done with it)
    for (int i = 0; i < LastFrame.N_total; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        int cam = cam_of.at(i);                   // LastFrame.keypoint_to_cam.find(i)->second (:3490)
        if (cam < 0 || cam > 1) continue;         // no entry: the reference dereferences end() there -- undefined; skipped here
#ifdef MORB_VERBATIM_MAT_ALGEBRA
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        if (cam == 1) x3Dc = mRcam21 * x3Dc + mtcam21;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
#else
        // the same values as the cv::Mat expressions above, element by element (apply_rt), without the three temporaries
        const cv::Mat x3Dw = pMP->GetWorldPos();
        float xw[3] = {x3Dw.at<float>(0), x3Dw.at<float>(1), x3Dw.at<float>(2)}, x3Dc[3];
        apply_rt(Pcw, xw, x3Dc);
        if (cam == 1) { const float t3[3] = {x3Dc[0], x3Dc[1], x3Dc[2]}; apply_rt(Pcam21, t3, x3Dc); }
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
#endif
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        int nLastOctave = LastFrame.mvKeys_total[i].octave;
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        q.emplace_back();
        orbm_query& Q = q.back();
        Q.u = u; Q.v = v; Q.radius = radius;
        Q.ur = u - CurrentFrame.mbf * invzc;
        if (bForward[cam]) { Q.min_level = nLastOctave; Q.max_level = -1; }            // GetFeaturesInArea(cam,u,v,r,nLastOctave)
        else if (bBackward[cam]) { Q.min_level = 0; Q.max_level = nLastOctave; }
        else { Q.min_level = nLastOctave - 1; Q.max_level = nLastOctave + 1; }
        Q.cam = cam;
        Q.blocks = pMP->Observations() > 0 ? 1 : 0;
        Q.angle = LastFrame.mvKeysUn_total[i].angle;
        const cv::Mat dMP = pMP->GetDescriptor();
        std::memcpy(Q.desc, dMP.ptr(0), 32);
        qmp.push_back(pMP);
    }

    tls.last_us[0] = us_since(t_query) + std::chrono::duration<float, std::micro>(t_frame - t_entry).count();
    const clk::time_point t_search = clk::now();
    int rc;
    std::vector<int32_t> match(CurrentFrame.N_total > 0 ? CurrentFrame.N_total : 1);
    int nmatches = 0;
    std::vector<uint8_t> occupied(match.size(), 0);  // :3566-3568 also skips points that were there before the call
    bool any_occupied = false;                       // (none in TrackWithMotionModel: it clears the vector first, Tracking.cc:1254)
    for (int g = 0; g < CurrentFrame.N_total; ++g) {
        occupied[g] = (CurrentFrame.mvpMapPoints[g] && CurrentFrame.mvpMapPoints[g]->Observations() > 0) ? 1 : 0;
        any_occupied |= occupied[g] != 0;
    }
    rc = orbm_search_by_projection(Handle(), fr, q.data(), (int)q.size(), any_occupied ? occupied.data() : nullptr, TH_HIGH,
                                   mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc) return fail("orbm_search_by_projection", rc);
    resident::reader_done(orbm_stream(Handle()));   // (the results are here: the frame build that read the extractors' rows finished long before them)
    tls.last_us[2] = us_since(t_search);
    // The reference starts from whatever CurrentFrame.mvpMapPoints holds (all NULL in TrackWithMotionModel,
    // src/Tracking.cc:1254) and only ever writes accepted matches / NULLs for histogram rejects.
    for (int g = 0; g < CurrentFrame.N_total; ++g) {
        if (match[g] >= 0) CurrentFrame.mvpMapPoints[g] = qmp[match[g]];
        else if (match[g] == -2) CurrentFrame.mvpMapPoints[g] = static_cast<MapPoint*>(NULL);  // histogram reject, :3631
    }
    return nmatches;
}

}  // namespace ORB_SLAM2
