// ORBextractor.cc -- see ORBextractor.h.  Host glue only.
#include "ORBextractor.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include "../../include/orbx.h"

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

ORBextractor::~ORBextractor() { orbx_destroy(handle_); }

bool ORBextractor::EnsureHandle(int width, int height) {
    if (handle_ && width <= cap_w_ && height <= cap_h_) return true;
    orbx_destroy(handle_);
    handle_ = nullptr;
    cap_w_ = width > cap_w_ ? width : cap_w_; cap_h_ = height > cap_h_ ? height : cap_h_;
    orbx_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    const char* dev = std::getenv("MORB_DEVICE");
    int rc = orbx_create(&p, 1, cap_w_, cap_h_, dev ? std::atoi(dev) : 0, &handle_);
    if (rc) { handle_ = nullptr; return fail("orbx_create", rc); }
    return true;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    if (!EnsureHandle(image.cols, image.rows)) return;
    int rc = orbx_upload(handle_, 0, image.ptr(0), image.cols, image.rows, (int)image.step);
    if (rc) { fail("orbx_upload", rc); return; }
    if ((rc = orbx_run(handle_))) { fail("orbx_run", rc); return; }
    const int n = orbx_count(handle_, 0);
    if (n == 0) {
        _keypoints.clear();
        _descriptors.release();
    } else {
        // results land in fresh containers first: a failed download must not leave the caller with half of them
        std::vector<cv::KeyPoint> kps(n);
        cv::Mat desc(n, 32, CV_8U);
        rc = orbx_download(handle_, 0, reinterpret_cast<orb_keypoint*>(kps.data()), desc.ptr(0), n);
        if (rc) { fail("orbx_download", rc); return; }
        _keypoints.swap(kps);
        _descriptors.create(n, 32, CV_8U);
        cv::Mat out = _descriptors.getMat();
        std::memcpy(out.ptr(0), desc.ptr(0), (size_t)n * 32);
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
    // one shared N-camera handle, cached on the first extractor of the batch
    static thread_local orbx_extractor* batch = nullptr;
    static thread_local std::vector<orbx_params> batch_params;
    static thread_local int bw = 0, bh = 0;
    std::vector<orbx_params> ps(n);
    int mw = 64, mh = 64;
    for (int i = 0; i < n; ++i) {
        ps[i] = {ex[i]->nfeatures, (float)ex[i]->scaleFactor, ex[i]->nlevels, ex[i]->iniThFAST, ex[i]->minThFAST};
        mw = std::max(mw, images[i].cols); mh = std::max(mh, images[i].rows);
    }
    bool same = batch && (int)batch_params.size() == n && mw <= bw && mh <= bh;
    for (int i = 0; same && i < n; ++i) same = std::memcmp(&ps[i], &batch_params[i], sizeof(orbx_params)) == 0;
    if (!same) {
        orbx_destroy(batch); batch = nullptr;
        bw = std::max(bw, mw); bh = std::max(bh, mh);
        const char* dev = std::getenv("MORB_DEVICE");
        int rc = orbx_create(ps.data(), n, bw, bh, dev ? std::atoi(dev) : 0, &batch);
        if (rc) { batch = nullptr; fail("orbx_create(batch)", rc); return; }
        batch_params = ps;
    }
    int rc;
    for (int i = 0; i < n; ++i)
        if ((rc = orbx_upload(batch, i, images[i].empty() ? nullptr : images[i].ptr(0), images[i].cols, images[i].rows,
                              (int)images[i].step))) {
            fail("orbx_upload", rc); return;
        }
    if ((rc = orbx_run(batch))) { fail("orbx_run", rc); return; }
    for (int i = 0; i < n; ++i) {
        if (images[i].empty()) continue;  // untouched outputs, like operator()
        const int k = orbx_count(batch, i);
        keypoints[i].clear();
        if (k == 0) { descriptors[i].release(); continue; }
        keypoints[i].resize(k);
        descriptors[i].create(k, 32, CV_8U);
        if ((rc = orbx_download(batch, i, reinterpret_cast<orb_keypoint*>(keypoints[i].data()), descriptors[i].ptr(0), k))) {
            fail("orbx_download", rc); keypoints[i].clear(); descriptors[i].release(); return;
        }
    }
}

}  // namespace ORB_SLAM2
