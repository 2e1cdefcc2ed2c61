// ORBextractor.cc -- see ORBextractor.h.  Host glue only.
#include "ORBextractor.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include "../../include/orbx.h"

namespace ORB_SLAM2 {

static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint must be the 28-byte POD the ABI writes");

static void die(const char* what, int rc) {
    std::fprintf(stderr, "ORBextractor: %s failed (%d): %s\n", what, rc, orb_last_error());
    std::abort();  // the reference has no error channel here (void operator()), and there is no CPU fallback
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    orbx_params p = {nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST};
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    int rc = orbx_tables(&p, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                         mnFeaturesPerLevel.data(), umax.data());
    if (rc) die("orbx_tables", rc);
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { orbx_destroy(handle_); }

void ORBextractor::EnsureHandle(int width, int height) {
    if (handle_ && width <= cap_w_ && height <= cap_h_) return;
    orbx_destroy(handle_);
    handle_ = nullptr;
    cap_w_ = width > cap_w_ ? width : cap_w_; cap_h_ = height > cap_h_ ? height : cap_h_;
    orbx_params p = {nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    const char* dev = std::getenv("MORB_DEVICE");
    int rc = orbx_create(&p, 1, cap_w_, cap_h_, dev ? std::atoi(dev) : 0, &handle_);
    if (rc) die("orbx_create", rc);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    const cv::Mat& image = _image;
    assert(image.type() == CV_8UC1);
    EnsureHandle(image.cols, image.rows);
    int rc = orbx_upload(handle_, 0, image.ptr(0), image.cols, image.rows, (int)image.step);
    if (rc) die("orbx_upload", rc);
    if ((rc = orbx_run(handle_))) die("orbx_run", rc);
    const int n = orbx_count(handle_, 0);
    _keypoints.clear();
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        _keypoints.resize(n);
        rc = orbx_download(handle_, 0, reinterpret_cast<orb_keypoint*>(_keypoints.data()), _descriptors.ptr(0), n);
        if (rc) die("orbx_download", rc);
    }
    if (materialise_) {
        for (int l = 0; l < nlevels; ++l) {
            int w = 0, h = 0;
            std::vector<unsigned char> buf((size_t)image.cols * image.rows);
            if ((rc = orbx_debug_level(handle_, 0, l, buf.data(), (int)buf.size(), &w, &h))) die("orbx_debug_level", rc);
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
        if (rc) die("orbx_create(batch)", rc);
        batch_params = ps;
    }
    int rc;
    for (int i = 0; i < n; ++i)
        if ((rc = orbx_upload(batch, i, images[i].empty() ? nullptr : images[i].ptr(0), images[i].cols, images[i].rows,
                              (int)images[i].step)))
            die("orbx_upload", rc);
    if ((rc = orbx_run(batch))) die("orbx_run", rc);
    for (int i = 0; i < n; ++i) {
        if (images[i].empty()) continue;  // untouched outputs, like operator()
        const int k = orbx_count(batch, i);
        keypoints[i].clear();
        if (k == 0) { descriptors[i].release(); continue; }
        keypoints[i].resize(k);
        descriptors[i].create(k, 32, CV_8U);
        if ((rc = orbx_download(batch, i, reinterpret_cast<orb_keypoint*>(keypoints[i].data()), descriptors[i].ptr(0), k)))
            die("orbx_download", rc);
    }
}

}  // namespace ORB_SLAM2
