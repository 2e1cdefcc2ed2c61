// pin_opencv -- pins the CPU oracle (oracle/orb_oracle.cpp) against OpenCV itself and against the reference's own ORBextractor.
// TEST INFRASTRUCTURE.  Cannot run in the build image (no OpenCV: SURVEY section 8c); it is kept compiling here
// (`g++ -fsyntax-only` against declarations, tests/test_oracle_opencv_pin.py) and runs with ONE command on any box that has
// OpenCV 2.4.x / 3.x (the versions the reference supports, CMakeLists.txt:42-45, README.md:19) and a checkout of the reference:
//
//     make -C oracle pin REF=/path/to/Multi_ORB_SLAM && python -m pytest tests/test_oracle_opencv_pin.py
//
// What it does, per OpenCV operator the reference's front end calls (reference call sites in brackets), on every
// image it is given (tests/natural.py frames: 3 photographs x 640x480 / 1280x720 / 1920x1080, written as raw files by the test):
//   resize        cv::resize(INTER_LINEAR) down the 8-level chain           [src/ORBextractor.cc:1122]   vs orc_resize_linear_u8
//   border        cv::copyMakeBorder(19, BORDER_REFLECT_101)               [:1124-1130]                 vs orc_copy_make_border_reflect101
//   fast          cv::FAST(cell, kps, 20 | 7, true) on whole levels        [:810,815]                   vs orc_fast
//   blur          cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101)          [:1087]                      vs orc_gaussian_blur7
//   atan2         cv::fastAtan2 over the moment range of IC_Angle          [:103]                       vs orc_fast_atan2
//   round         cvRound at halves and around them                         [:81,115,119,443,461,1114]   vs orc_cv_round
//   extractor     ORB_SLAM2::ORBextractor::operator() -- the reference's src/ORBextractor.cc compiled from where it lies --
//                 keypoints (as a set per level: the reference breaks quadtree ties by heap address, App. C-1) and descriptors
//                                                                                                        vs orc_extract
// Every line of output is `PIN <operator> <image> ok|FAIL <detail>`; the exit status is the number of failing operators.  DESIGN.md
// section 2 names, per operator, the single oracle function to change when it fails; when all pass, parity is pinned.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#ifndef PIN_NO_REFERENCE
#include "ORBextractor.h"   // the reference's own header, in place (-I$(REF)/include)
#endif

// ---- the oracle's C surface (oracle/orb_oracle.cpp)
struct OrcKeyPoint { float x, y, size, angle, response; int octave, class_id; };
extern "C" {
void orc_level_sizes(int W, int H, float scaleFactor, int nlevels, int* w, int* h);
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void orc_copy_make_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border, int dstride);
int orc_fast(const uint8_t* view, int cols, int rows, int stride, int threshold, OrcKeyPoint* out, int cap);
void orc_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);
float orc_fast_atan2(float y, float x);
int orc_cv_round(double v);
int orc_extract(const uint8_t* img, int W, int H, int stride, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                OrcKeyPoint* kps_out, uint8_t* desc_out, int cap);
}

static int g_fail = 0;
static std::map<std::string, int> g_fail_by_op;

static void report(const char* op, const char* image, bool ok, const std::string& detail) {
    std::printf("PIN %-9s %-24s %s %s\n", op, image, ok ? "ok  " : "FAIL", detail.c_str());
    if (!ok) { ++g_fail; ++g_fail_by_op[op]; }
}

static std::string fmt(const char* f, ...) {
    char b[512];
    va_list ap; va_start(ap, f); std::vsnprintf(b, sizeof b, f, ap); va_end(ap);
    return b;
}

static size_t diff_bytes(const cv::Mat& a, const uint8_t* b, int w, int h, int* fx, int* fy) {
    size_t n = 0;
    for (int y = 0; y < h; ++y) {
        const uint8_t* r = a.ptr<uint8_t>(y);
        for (int x = 0; x < w; ++x)
            if (r[x] != b[(size_t)y * w + x]) { if (!n) { *fx = x; *fy = y; } ++n; }
    }
    return n;
}

// raw image file written by the test: int32 w, int32 h, then h*w bytes
static bool load_raw(const char* path, std::vector<uint8_t>& px, int& w, int& h) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    int hdr[2];
    bool ok = std::fread(hdr, 4, 2, f) == 2;
    if (ok) { w = hdr[0]; h = hdr[1]; px.resize((size_t)w * h); ok = std::fread(px.data(), 1, px.size(), f) == px.size(); }
    std::fclose(f);
    return ok;
}

static void pin_image(const char* name, const std::vector<uint8_t>& px, int W, int H, int nfeatures) {
    const int L = 8;
    int lw[L], lh[L];
    orc_level_sizes(W, H, 1.2f, L, lw, lh);
    cv::Mat level(H, W, CV_8UC1, (void*)px.data(), (size_t)W);
    std::vector<uint8_t> olevel(px);
    size_t bad_resize = 0, bad_border = 0, bad_blur = 0, bad_fast = 0;
    std::string first_resize, first_border, first_blur, first_fast;
    for (int l = 0; l < L; ++l) {
        if (l > 0) {
            // ---- resize: the previous level (the same bytes on both sides) to this level's size
            cv::Mat next;
            cv::resize(level, next, cv::Size(lw[l], lh[l]), 0, 0, cv::INTER_LINEAR);
            std::vector<uint8_t> onext((size_t)lw[l] * lh[l]);
            orc_resize_linear_u8(olevel.data(), lw[l - 1], lh[l - 1], lw[l - 1], onext.data(), lw[l], lh[l], lw[l]);
            int fx = 0, fy = 0;
            const size_t d = diff_bytes(next, onext.data(), lw[l], lh[l], &fx, &fy);
            if (d && first_resize.empty()) first_resize = fmt("level %d: %zu px differ, first at (%d, %d): cv %d, oracle %d", l, d, fx, fy,
                                                              next.at<uint8_t>(fy, fx), onext[(size_t)fy * lw[l] + fx]);
            bad_resize += d;
            // both sides go on from OpenCV's level: a resize difference is reported once, here, and does not cascade into the operators below
            level = next;
            olevel.resize((size_t)lw[l] * lh[l]);
            for (int y = 0; y < lh[l]; ++y) std::memcpy(&olevel[(size_t)y * lw[l]], next.ptr<uint8_t>(y), (size_t)lw[l]);
        }
        const int w = lw[l], h = lh[l];
        // ---- border
        {
            cv::Mat b;
            cv::copyMakeBorder(level, b, 19, 19, 19, 19, cv::BORDER_REFLECT_101);
            std::vector<uint8_t> ob((size_t)(w + 38) * (h + 38));
            orc_copy_make_border_reflect101(olevel.data(), w, h, w, ob.data(), 19, w + 38);
            int fx = 0, fy = 0;
            const size_t d = diff_bytes(b, ob.data(), w + 38, h + 38, &fx, &fy);
            if (d && first_border.empty()) first_border = fmt("level %d: %zu px differ, first at (%d, %d)", l, d, fx, fy);
            bad_border += d;
        }
        // ---- blur
        {
            cv::Mat b;
            cv::GaussianBlur(level, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
            std::vector<uint8_t> ob((size_t)w * h);
            orc_gaussian_blur7(olevel.data(), w, h, ob.data());
            int fx = 0, fy = 0;
            const size_t d = diff_bytes(b, ob.data(), w, h, &fx, &fy);
            if (d && first_blur.empty()) first_blur = fmt("level %d: %zu px differ, first at (%d, %d): cv %d, oracle %d", l, d, fx, fy,
                                                          b.at<uint8_t>(fy, fx), ob[(size_t)fy * w + fx]);
            bad_blur += d;
        }
        // ---- FAST with non-maximum suppression, both thresholds, whole level (the per-cell calls are sub-views of the same routine)
        for (int th : {20, 7}) {
            std::vector<cv::KeyPoint> k;
            cv::FAST(level, k, th, true);
            std::vector<OrcKeyPoint> ok((size_t)w * h / 4 + 16);
            const int on = orc_fast(olevel.data(), w, h, w, th, ok.data(), (int)ok.size());
            size_t d = (size_t)std::abs((int)k.size() - on);
            const size_t m = std::min(k.size(), (size_t)on);
            for (size_t i = 0; i < m; ++i)
                if (k[i].pt.x != ok[i].x || k[i].pt.y != ok[i].y || k[i].response != ok[i].response) {
                    if (first_fast.empty()) first_fast = fmt("level %d th %d: keypoint %zu cv (%g, %g, %g) oracle (%g, %g, %g); counts %zu / %d", l, th, i,
                                                             k[i].pt.x, k[i].pt.y, k[i].response, ok[i].x, ok[i].y, ok[i].response, k.size(), on);
                    ++d;
                }
            if (d && first_fast.empty()) first_fast = fmt("level %d th %d: counts cv %zu, oracle %d", l, th, k.size(), on);
            bad_fast += d;
        }
    }
    report("resize", name, bad_resize == 0, first_resize);
    report("border", name, bad_border == 0, first_border);
    report("blur", name, bad_blur == 0, first_blur);
    report("fast", name, bad_fast == 0, first_fast);
#ifndef PIN_NO_REFERENCE
    // ---- the reference's own extractor against the oracle's (src/ORBextractor.cc:1044-1107)
    {
        ORB_SLAM2::ORBextractor ex(nfeatures, 1.2f, 8, 20, 7);
        cv::Mat img(H, W, CV_8UC1, (void*)px.data(), (size_t)W);
        std::vector<cv::KeyPoint> k;
        cv::Mat d;
        ex(img, cv::Mat(), k, d);
        const int cap = nfeatures + 3 * 8 + 64;
        std::vector<OrcKeyPoint> ok(cap);
        std::vector<uint8_t> od((size_t)cap * 32);
        const int on = orc_extract(px.data(), W, H, W, nfeatures, 1.2f, 8, 20, 7, ok.data(), od.data(), cap);
        bool same_count = (int)k.size() == on;
        // in order first (holds whenever no quadtree tie was broken by heap address); otherwise as a set keyed by (octave, x, y)
        size_t ordered_bad = 0;
        for (int i = 0; same_count && i < on; ++i) {
            const cv::KeyPoint& a = k[i]; const OrcKeyPoint& b = ok[i];
            if (a.pt.x != b.x || a.pt.y != b.y || a.octave != b.octave || a.angle != b.angle || a.response != b.response || a.size != b.size ||
                std::memcmp(d.ptr<uint8_t>(i), &od[(size_t)i * 32], 32))
                ++ordered_bad;
        }
        size_t set_bad = 0;
        if (!same_count || ordered_bad) {
            std::map<std::vector<float>, int> at;
            for (int i = 0; i < on; ++i) at[{(float)ok[i].octave, ok[i].x, ok[i].y}] = i;
            for (size_t i = 0; i < k.size(); ++i) {
                auto it = at.find({(float)k[i].octave, k[i].pt.x, k[i].pt.y});
                if (it == at.end()) { ++set_bad; continue; }
                const OrcKeyPoint& b = ok[it->second];
                if (k[i].angle != b.angle || k[i].response != b.response || k[i].size != b.size ||
                    std::memcmp(d.ptr<uint8_t>((int)i), &od[(size_t)it->second * 32], 32))
                    ++set_bad;
            }
        }
        const bool okk = same_count && (ordered_bad == 0 || set_bad == 0);
        report("extractor", name, okk, fmt("reference %zu keypoints, oracle %d; %zu differ in order, %zu as a set%s", k.size(), on, ordered_bad, set_bad,
                                           ordered_bad && !set_bad ? " (same set in another order: a quadtree tie the reference breaks by heap address)" : ""));
    }
#else
    (void)nfeatures;
#endif
}

static void pin_scalars() {
    // ---- fastAtan2 over IC_Angle's moment range (|m| < 2^21), dense near the axes and the diagonals, plus the zero cases
    size_t bad = 0; std::string first;
    auto chk = [&](float y, float x) {
        const float a = cv::fastAtan2(y, x), b = orc_fast_atan2(y, x);
        if (std::memcmp(&a, &b, 4)) { if (!bad) first = fmt("fastAtan2(%g, %g): cv %.9g oracle %.9g", y, x, a, b); ++bad; }
    };
    for (int y = -2048; y <= 2048; y += 7)
        for (int x = -2048; x <= 2048; x += 5) chk((float)(y * 1021), (float)(x * 1019));
    for (int i = -300; i <= 300; ++i) { chk((float)i, 0.f); chk(0.f, (float)i); chk((float)i, (float)i); chk((float)i, (float)-i); chk((float)i, (float)(i + 1)); }
    report("atan2", "-", bad == 0, first);
    // ---- cvRound: halves go to even; values the table builders and the level sizes feed it
    bad = 0; first.clear();
    for (int i = -4000; i <= 4000; ++i)
        for (double f : {0.0, 0.25, 0.5 - 1e-12, 0.5, 0.5 + 1e-12, 0.75}) {
            const double v = i + f;
            if (cvRound(v) != orc_cv_round(v)) { if (!bad) first = fmt("cvRound(%.15g): cv %d oracle %d", v, cvRound(v), orc_cv_round(v)); ++bad; }
        }
    report("round", "-", bad == 0, first);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: pin_opencv <nfeatures>:<image.raw> ...\n"); return 64; }
    std::printf("PIN opencv %s\n", CV_VERSION);
    pin_scalars();
    for (int i = 1; i < argc; ++i) {
        const char* colon = std::strchr(argv[i], ':');
        if (!colon) { std::fprintf(stderr, "argument %s: expected <nfeatures>:<path>\n", argv[i]); return 64; }
        const int nf = std::atoi(argv[i]);
        std::vector<uint8_t> px; int w = 0, h = 0;
        if (!load_raw(colon + 1, px, w, h)) { std::fprintf(stderr, "cannot read %s\n", colon + 1); return 65; }
        const char* base = std::strrchr(colon + 1, '/');
        pin_image(base ? base + 1 : colon + 1, px, w, h, nf);
    }
    for (auto& kv : g_fail_by_op) std::printf("PIN SUMMARY %s: %d image(s) differ\n", kv.first.c_str(), kv.second);
    std::printf("PIN SUMMARY %s\n", g_fail ? "parity NOT pinned" : "every operator and the reference extractor agree with the oracle: parity pinned");
    return (int)g_fail_by_op.size();
}
