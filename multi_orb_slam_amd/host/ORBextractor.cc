// ORBextractor.cc -- see ORBextractor.h.  Host glue only.
#include "ORBextractor.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <thread>
#include "../../include/orbx.h"
#include "resident.h"

namespace ORB_SLAM2 {

static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint must be the 28-byte POD the ABI writes");

// The reference has no error channel here (void operator(), no CPU fallback in this library): a failed device call leaves the
// outputs untouched -- exactly what the reference does for an empty image (src/ORBextractor.cc:1047-1048) -- reports once per
// call on stderr, and keeps the text for ORBextractor::LastError().  The SLAM process is never aborted from here.
static std::atomic<unsigned long> g_failures{0};
static bool fail(const char* what, int rc) {
    g_failures.fetch_add(1, std::memory_order_relaxed);
    std::fprintf(stderr, "ORBextractor: %s failed (%d): %s -- outputs left untouched\n", what, rc, orb_last_error());
    return false;
}
const char* ORBextractor::LastError() { return orb_last_error(); }
unsigned long ORBextractor::FailureCount() { return g_failures.load(std::memory_order_relaxed); }

// ---- resident.h
namespace resident {
namespace {
struct Entry { const void* owner; unsigned long thread; const uint8_t* host_rows; const uint8_t* d_rows; int n; void* reader; };
std::mutex g_mu;
std::vector<Entry> g_entries;
std::atomic<unsigned long> g_served{0}, g_missed{0}, g_tokens{0};
// a thread's token: handed out once, never again (std::thread::id values are reused when a thread has exited -- the reference's
// stereo constructor extracts on short-lived threads)
unsigned long my_token() { static thread_local const unsigned long t = g_tokens.fetch_add(1, std::memory_order_relaxed) + 1; return t; }
}  // namespace
void publish(const void* owner, const uint8_t* host_rows, const uint8_t* d_rows, int n) {
    const unsigned long me = my_token();
    std::lock_guard<std::mutex> lk(g_mu);
    for (Entry& e : g_entries)
        if (e.owner == owner) { e = Entry{owner, me, host_rows, d_rows, n, e.reader}; return; }
    g_entries.push_back(Entry{owner, me, host_rows, d_rows, n, nullptr});
}
void retire(const void* owner) {   // (the rows are about to be overwritten: nobody is served from them any more; a pending reader stays noted)
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_entries.size(); ++i)
        if (g_entries[i].owner == owner) {
            if (g_entries[i].reader) { g_entries[i].d_rows = nullptr; g_entries[i].host_rows = nullptr; g_entries[i].n = 0; }
            else { g_entries[i] = g_entries.back(); g_entries.pop_back(); }
            return;
        }
}
const uint8_t* find(const uint8_t* rows, int n, const void** owner_out) {
    if (!rows || n <= 0) return nullptr;
    const unsigned long me = my_token();
    // this thread's entries of the right size are copied out under the lock and compared outside it: only their owner -- an extractor
    // used on THIS thread -- republishes or retires them, so the rows they point at cannot change under the comparison
    Entry mine[8];
    int k = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (const Entry& e : g_entries) if (e.n == n && e.thread == me && e.d_rows && k < 8) mine[k++] = e;
    }
    for (int i = 0; i < k; ++i)
        if (std::memcmp(rows, mine[i].host_rows, (size_t)n * 32) == 0) {
            g_served.fetch_add(1, std::memory_order_relaxed);
            if (owner_out) *owner_out = mine[i].owner;
            return mine[i].d_rows;
        }
    g_missed.fetch_add(1, std::memory_order_relaxed);
    return nullptr;
}
void note_reader(const void* owner, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Entry& e : g_entries) if (e.owner == owner) { e.reader = stream; return; }
}
void* take_reader(const void* owner) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_entries.size(); ++i)
        if (g_entries[i].owner == owner) {
            void* s = g_entries[i].reader;
            g_entries[i].reader = nullptr;
            if (!g_entries[i].d_rows) { g_entries[i] = g_entries.back(); g_entries.pop_back(); }   // (retired, kept for its reader only)
            return s;
        }
    return nullptr;
}
void reader_done(void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_entries.size();) {
        if (g_entries[i].reader == stream) {
            g_entries[i].reader = nullptr;
            if (!g_entries[i].d_rows) { g_entries[i] = g_entries.back(); g_entries.pop_back(); continue; }   // (retired, kept for its reader only)
        }
        ++i;
    }
}
void stats(unsigned long* served, unsigned long* missed) { *served = g_served.load(); *missed = g_missed.load(); }
}  // namespace resident

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    orbx_params p = {nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST};
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    int rc = orbx_tables(&p, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                         mnFeaturesPerLevel.data(), umax.data());
    if (rc) fail("orbx_tables", rc);   // (bad constructor arguments: the tables stay zero, every later call reports the same)
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { resident::retire(this); (void)resident::take_reader(this); orbx_destroy(handle_); }   // (no entry outlives its owner)

bool ORBextractor::EnsureHandle(int width, int height) {
    if (handle_ && width <= cap_w_ && height <= cap_h_) return true;
    resident::retire(this);
    orbx_destroy(handle_);
    handle_ = nullptr;
    cap_w_ = width > cap_w_ ? width : cap_w_; cap_h_ = height > cap_h_ ? height : cap_h_;
    orbx_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    int rc = orbx_create(&p, 1, cap_w_, cap_h_, host_device(), &handle_);
    if (rc) { handle_ = nullptr; return fail("orbx_create", rc); }
    return true;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    // A failed device call (there is no CPU path) hands the caller an EMPTY frame -- the tracker's own too-few-features
    // handling takes over -- never the previous frame's features under the new timestamp; LastError() / FailureCount() say why.
    if (!EnsureHandle(image.cols, image.rows)) { _keypoints.clear(); _descriptors.release(); return; }
    // one call: the image goes up through host-written staging, the describe kernel mirrors its results into pinned memory,
    // and they are copied from there into scratch of the right capacity (a failed call must not leave the caller with half)
    const int cap = nfeatures + 4 * nlevels;
    scratch_kps_.resize(cap); scratch_desc_.resize((size_t)cap * 32);
    const uint8_t* img_ptr = image.ptr(0);
    const int w = image.cols, h = image.rows, stride = (int)image.step;
    orb_keypoint* kp_ptr = reinterpret_cast<orb_keypoint*>(scratch_kps_.data());
    uint8_t* d_ptr = scratch_desc_.data();
    int n = 0;
    resident::retire(this);   // (the device rows of the previous call are about to be overwritten ...
    if (void* rs = resident::take_reader(this)) (void)orbx_wait_for_stream(handle_, rs);   // ... behind whatever still reads THIS extractor's rows on a matcher's stream)
    int rc = orbx_extract(handle_, 1, &img_ptr, &w, &h, &stride, &kp_ptr, &d_ptr, &cap, &n);
    if (rc) { fail("orbx_extract", rc); _keypoints.clear(); _descriptors.release(); return; }
    // the rows stay where the describe kernel wrote them until this extractor's next call: a search of the frame built from them
    // reads them there (resident.h), held against scratch_desc_ byte for byte
    if (n > 0) resident::publish(this, scratch_desc_.data(), orbx_device_descriptors(handle_, 0), n);
    if (n == 0) {
        _keypoints.clear();
        _descriptors.release();
    } else {
        _keypoints.assign(scratch_kps_.begin(), scratch_kps_.begin() + n);
        _descriptors.create(n, 32, CV_8U);
        cv::Mat out = _descriptors.getMat();
        std::memcpy(out.ptr(0), scratch_desc_.data(), (size_t)n * 32);
    }
    if (materialise_) {
        for (int l = 0; l < nlevels; ++l) {
            int w = 0, h = 0;
            std::vector<unsigned char> buf((size_t)image.cols * image.rows);
            if ((rc = orbx_debug_level(handle_, 0, l, buf.data(), (int)buf.size(), &w, &h))) { fail("orbx_debug_level", rc); return; }
            mvImagePyramid[l].create(h, w, CV_8UC1);
            std::memcpy(mvImagePyramid[l].ptr(0), buf.data(), (size_t)w * h);
        }
    }
}

void ORBextractor::ExtractBatch(const std::vector<ORBextractor*>& ex, const std::vector<cv::Mat>& images,
                                std::vector<std::vector<cv::KeyPoint> >& keypoints, std::vector<cv::Mat>& descriptors) {
    const int n = (int)ex.size();
    assert((int)images.size() == n);
    keypoints.resize(n); descriptors.resize(n);
    // one shared N-camera handle per calling thread (destroyed when the thread exits)
    struct BatchState {
        orbx_extractor* h = nullptr; std::vector<orbx_params> params; int w = 0, h_px = 0;
        std::vector<std::vector<uint8_t> > ds;   // the last call's descriptors per camera (resident.h holds frames against them)
        char slot[64];                           // &slot[i] = identity of camera i's published rows
        void retire_all() { for (int i = 0; i < 64; ++i) { resident::retire(&slot[i]); (void)resident::take_reader(&slot[i]); } }
        ~BatchState() { retire_all(); orbx_destroy(h); }
    };
    static thread_local BatchState B;
    orbx_extractor*& batch = B.h;
    std::vector<orbx_params>& batch_params = B.params;
    int &bw = B.w, &bh = B.h_px;
    // (a failed call leaves EMPTY outputs for every non-empty input image, as operator() does)
    auto empty_outputs = [&]() { for (int i = 0; i < n; ++i) if (!images[i].empty()) { keypoints[i].clear(); descriptors[i].release(); } };
    std::vector<orbx_params> ps(n);
    int mw = 64, mh = 64;
    for (int i = 0; i < n; ++i) {
        ps[i] = {ex[i]->nfeatures, (float)ex[i]->scaleFactor, ex[i]->nlevels, ex[i]->iniThFAST, ex[i]->minThFAST};
        mw = std::max(mw, images[i].cols); mh = std::max(mh, images[i].rows);
    }
    bool same = batch && (int)batch_params.size() == n && mw <= bw && mh <= bh;
    for (int i = 0; same && i < n; ++i) same = std::memcmp(&ps[i], &batch_params[i], sizeof(orbx_params)) == 0;
    B.retire_all();   // (the device rows of the previous call are about to be overwritten)
    if (!same) {
        orbx_destroy(batch); batch = nullptr;
        bw = std::max(bw, mw); bh = std::max(bh, mh);
        int rc = orbx_create(ps.data(), n, bw, bh, host_device(), &batch);
        if (rc) { batch = nullptr; fail("orbx_create(batch)", rc); empty_outputs(); return; }
        batch_params = ps;
    }
    std::vector<const uint8_t*> img(n);
    std::vector<int> w(n), h(n), st(n), cap(n), cnt(n, 0);
    std::vector<std::vector<cv::KeyPoint> > kp(n);
    std::vector<std::vector<uint8_t> >& ds = B.ds;
    ds.resize(n);
    std::vector<orb_keypoint*> kp_ptr(n);
    std::vector<uint8_t*> d_ptr(n);
    for (int i = 0; i < n; ++i) {
        img[i] = images[i].empty() ? nullptr : images[i].ptr(0);
        w[i] = images[i].cols; h[i] = images[i].rows; st[i] = (int)images[i].step;
        cap[i] = ex[i]->nfeatures + 4 * ex[i]->nlevels;
        kp[i].resize(cap[i]); ds[i].resize((size_t)cap[i] * 32);
        kp_ptr[i] = reinterpret_cast<orb_keypoint*>(kp[i].data()); d_ptr[i] = ds[i].data();
    }
    for (int i = 0; i < n && i < 64; ++i) {   // (behind whatever still reads the previous call's rows of any slot on a matcher's stream)
        resident::retire(&B.slot[i]);
        if (void* rs = resident::take_reader(&B.slot[i])) (void)orbx_wait_for_stream(batch, rs);
    }
    const int rc = orbx_extract(batch, n, img.data(), w.data(), h.data(), st.data(), kp_ptr.data(), d_ptr.data(), cap.data(), cnt.data());
    if (rc) { fail("orbx_extract(batch)", rc); empty_outputs(); return; }
    for (int i = 0; i < n; ++i) {
        if (images[i].empty()) continue;  // untouched outputs, like operator()
        const int k = cnt[i];
        if (k == 0) { keypoints[i].clear(); descriptors[i].release(); continue; }
        kp[i].resize(k);
        keypoints[i].swap(kp[i]);
        descriptors[i].create(k, 32, CV_8U);
        std::memcpy(descriptors[i].ptr(0), ds[i].data(), (size_t)k * 32);
        if (i < 64) resident::publish(&B.slot[i], ds[i].data(), orbx_device_descriptors(batch, i), k);
    }
}

}  // namespace ORB_SLAM2
