// orb_oracle.cpp -- CPU restatement of the Multi_ORB_SLAM ORB front end (ORBextractor + ORBmatcher).
//
// THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
// `cpu_baseline` leg of bench.py may load it, and only as the checker / reported CPU baseline.  The
// product path (multi_orb_slam_amd/csrc/*.hip behind include/*.h) never calls into it.
//
// PARITY STATUS: **parity unpinned at the OpenCV boundary.**  The reference (C++, /root/reference) cannot be
// compiled here: src/ORBextractor.cc and src/ORBmatcher.cc need OpenCV 2.4/3.x headers + libraries, which are
// absent from this image and un-vendored (CMakeLists.txt:42-45), and the reference ships no tests, fixtures or
// golden vectors.  The OpenCV operators on the path (cv::resize INTER_LINEAR 8-bit, cv::FAST 9/16 + NMS,
// cv::GaussianBlur 7x7 sigma 2 on 8-bit, cv::fastAtan2, cvRound) are restated below from the published
// OpenCV 2.4.x / 3.2 generic C++ algorithms (modules/imgproc/src/imgwarp.cpp, smooth.cpp, filter.cpp,
// modules/features2d/src/fast.cpp + fast_score.cpp, modules/core/src/mathfuncs.cpp) -- see SURVEY.md App. A.
// Everything that is the reference's OWN code (tables, cell loop, quadtree, orientation, rBRIEF, Hamming,
// projection searches, histogram) is restated literally with file:line citations and pinned by known-answer
// tests in tests/ (SURVEY.md section 8c).
//
// Canonical choices where the reference itself is not reproducible:
//   C-1  quadtree tie-break by heap address  -> (size, creation sequence), newest first among equal sizes.
//   C-2  FP contraction                      -> none (compile with -ffp-contract=off).
//   C-3  libm cosf/sinf                      -> det_sincos(): a fixed double-precision operation sequence
//        (fdlibm-style 2-term pi/2 reduction + Taylor/Horner), identical on CPU and GPU by construction;
//        oracle/check_sincos.cpp compares it exhaustively with this image's glibc cosf/sinf.
//
// Build: see oracle/Makefile (g++ -O3 -march=native -ffp-contract=off -shared -fPIC).

#include <stdexcept>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// A-0 rounding primitives (OpenCV cvRound = round-half-to-even via cvtsd2si; cvFloor; cvCeil)
// ------------------------------------------------------------------------------------------------
inline int cv_round(double v) { return (int)std::nearbyint(v); }  // FE_TONEAREST is never changed here
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

struct KeyPoint {  // cv::KeyPoint POD layout, 28 bytes
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

const int PATCH_SIZE = 31;       // ORBextractor.cc:72
const int HALF_PATCH_SIZE = 15;  // ORBextractor.cc:73
const int EDGE_THRESHOLD = 19;   // ORBextractor.cc:74

struct Quad { signed char x0, y0, x1, y1; };
const Quad kPattern[256] = {
#include "../include/orb_pattern_31.inc"
};

// ------------------------------------------------------------------------------------------------
// a1  ORBextractor::ORBextractor  (ORBextractor.cc:411-471)
// ------------------------------------------------------------------------------------------------
struct Params {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;  // the member is declared double (ORBextractor.h:97) and initialised from a float
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[HALF_PATCH_SIZE + 1];
};

void init_params(Params& P, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    P.nfeatures = nfeatures; P.nlevels = nlevels; P.iniTh = iniTh; P.minTh = minTh;
    P.scaleFactor = (double)scaleFactor;
    P.scale.assign(nlevels, 1.0f); P.sigma2.assign(nlevels, 1.0f);
    for (int i = 1; i < nlevels; i++) {                       // :416-424
        P.scale[i] = (float)((double)P.scale[i - 1] * P.scaleFactor);
        P.sigma2[i] = P.scale[i] * P.scale[i];
    }
    P.inv_scale.resize(nlevels); P.inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {                       // :426-432
        P.inv_scale[i] = 1.0f / P.scale[i];
        P.inv_sigma2[i] = 1.0f / P.sigma2[i];
    }
    P.quota.assign(nlevels, 0);                               // :436-447
    float factor = (float)(1.0 / P.scaleFactor);              // 1.0f / (double) -> double -> float
    float nDesired = (float)nfeatures * (1 - factor) /
                     (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        P.quota[l] = cv_round(nDesired);
        sum += P.quota[l];
        nDesired *= factor;
    }
    P.quota[nlevels - 1] = std::max(nfeatures - sum, 0);

    // :455-470 umax
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (P.umax[v0] == P.umax[v0 + 1]) ++v0;
        P.umax[v] = v0;
        ++v0;
    }
}

// ------------------------------------------------------------------------------------------------
// a3  ComputePyramid (ORBextractor.cc:1109-1134) -- cv::resize INTER_LINEAR 8UC1 [OCV, App. A-1]
// ------------------------------------------------------------------------------------------------
struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // dense, stride == w  (the 19-px reflect-101 border of the reference's
                              // padded buffers is never read by any later stage: SURVEY App. A-1)
    const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
    uint8_t* row(int y) { return px.data() + (size_t)y * w; }
};

void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                      int dstride) {
    const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short(cv_round((1.f - fx) * COEF_SCALE));
        alpha[2 * dx + 1] = sat_short(cv_round(fx * COEF_SCALE));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;  // rows are clipped per tap below, coefficients are NOT altered (resizeGeneric_Invoker)
        beta[2 * dy] = sat_short(cv_round((1.f - fy) * COEF_SCALE));
        beta[2 * dy + 1] = sat_short(cv_round(fy * COEF_SCALE));
    }
    std::vector<int> H0(dw), H1(dw);
    auto hrow = [&](int sy, std::vector<int>& H) {
        sy = std::min(std::max(sy, 0), sh - 1);
        const uint8_t* S = src + (size_t)sy * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int s1 = S[std::min(sx + 1, sw - 1)];  // alpha1 == 0 whenever sx+1 is clamped
            H[dx] = S[sx] * alpha[2 * dx] + s1 * alpha[2 * dx + 1];
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        hrow(yofs[dy], H0);
        hrow(yofs[dy] + 1, H1);
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (H0[x] >> 4)) >> 16) + ((b1 * (H1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

inline int reflect101(int p, int n) {  // cv::borderInterpolate(BORDER_REFLECT_101)
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

void copy_make_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border,
                                 int dstride) {  // :1124-1131 (result unused downstream; kept for a3 coverage)
    for (int y = -border; y < h + border; y++) {
        const uint8_t* S = src + (size_t)reflect101(y, h) * sstride;
        uint8_t* D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
    }
}

void level_size(const Params& P, int W, int H, int level, int& w, int& h) {  // :1113-1114
    float s = P.inv_scale[level];
    w = cv_round((double)((float)W * s));
    h = cv_round((double)((float)H * s));
}

void compute_pyramid(const Params& P, const uint8_t* img, int W, int H, int stride, std::vector<Image>& pyr) {
    pyr.resize(P.nlevels);
    for (int l = 0; l < P.nlevels; l++) {
        int w, h;
        level_size(P, W, H, l, w, h);
        pyr[l].w = w; pyr[l].h = h; pyr[l].px.resize((size_t)w * h);
        if (l == 0)
            for (int y = 0; y < h; y++) std::memcpy(pyr[0].row(y), img + (size_t)y * stride, w);
        else
            resize_linear_u8(pyr[l - 1].px.data(), pyr[l - 1].w, pyr[l - 1].h, pyr[l - 1].w,
                             pyr[l].px.data(), w, h, w);
    }
}

// ------------------------------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmaxSuppression=true), type 9_16  [OCV, App. A-2 / A-3]
// ------------------------------------------------------------------------------------------------
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16> (fast_score.cpp), scalar path.
int corner_score16(const uint8_t* p, int stride, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = p[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - p[kRing[k & 15][1] * stride + kRing[k & 15][0]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]); a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]); a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]); b = std::max(b, (int)d[k + 4]); b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]); b = std::max(b, (int)d[k + 7]); b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// Segment test: >= 9 contiguous ring pixels all brighter than v+t or all darker than v-t (FAST_t<16>: count > K).
bool is_corner16(const uint8_t* p, int stride, int t) {
    int v = p[0];
    int cb = 0, cd = 0;
    for (int k = 0; k < 25; k++) {
        int x = p[kRing[k & 15][1] * stride + kRing[k & 15][0]];
        if (x > v + t) { if (++cb > 8) return true; } else cb = 0;
        if (x < v - t) { if (++cd > 8) return true; } else cd = 0;
    }
    return false;
}

// view = cols x rows window of an image; emits keypoints (x, y relative to the view, response = score)
// row-major, exactly the order of FAST_t's rolling 3-row buffer.
void fast_9_16_nms(const uint8_t* view, int cols, int rows, int stride, int threshold,
                   std::vector<KeyPoint>& out) {
    out.clear();
    if (cols < 7 || rows < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> score((size_t)cols * rows, 0);  // 0 outside the scored rectangle / for non-corners
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            const uint8_t* p = view + (size_t)i * stride + j;
            if (is_corner16(p, stride, threshold)) score[(size_t)i * cols + j] = (uint8_t)corner_score16(p, stride, threshold);
        }
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            int s = score[(size_t)i * cols + j];
            if (!s) continue;  // a corner's score is >= threshold; threshold 0 corners with score 0 never pass '>' either
            const uint8_t* pp = &score[(size_t)(i - 1) * cols + j];
            const uint8_t* pc = &score[(size_t)i * cols + j];
            const uint8_t* pn = &score[(size_t)(i + 1) * cols + j];
            if (s > pc[-1] && s > pc[1] && s > pp[-1] && s > pp[0] && s > pp[1] && s > pn[-1] && s > pn[0] &&
                s > pn[1]) {
                KeyPoint kp{(float)j, (float)i, 7.f, -1.f, (float)s, 0, -1};
                out.push_back(kp);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// a5  ExtractorNode::DivideNode (:482-538) and ORBextractor::DistributeOctTree (:540-764)
// ------------------------------------------------------------------------------------------------
struct Pt2i { int x, y; };
struct Node {
    std::vector<KeyPoint> keys;
    Pt2i UL, UR, BL, BR;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;  // creation sequence: canonical stand-in for the heap address (App. C-1)
};

void divide_node(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil((float)(n.UR.x - n.UL.x) / 2);
    const int halfY = (int)std::ceil((float)(n.BR.y - n.UL.y) / 2);
    n1.UL = n.UL; n1.UR = {n.UL.x + halfX, n.UL.y}; n1.BL = {n.UL.x, n.UL.y + halfY}; n1.BR = {n.UL.x + halfX, n.UL.y + halfY};
    n2.UL = n1.UR; n2.UR = n.UR; n2.BL = n1.BR; n2.BR = {n.UR.x, n.UL.y + halfY};
    n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = n.BL; n3.BR = {n1.BR.x, n.BL.y};
    n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = n.BR;
    for (const KeyPoint& kp : n.keys) {
        if (kp.x < (float)n1.UR.x) {
            if (kp.y < (float)n1.BR.y) n1.keys.push_back(kp); else n3.keys.push_back(kp);
        } else if (kp.y < (float)n1.BR.y) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint>& in, int minX, int maxX, int minY,
                                        int maxY, int N) {
    typedef std::pair<int, Node*> SP;
    auto sp_less = [](const SP& a, const SP& b) {  // std::pair '<' with the pointer replaced by seq (C-1)
        return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq;
    };
    long seq = 0;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    // The reference divides by nIni and indexes vpIniNodes[kp.pt.x / hX] (:544-565): for a level that is more than twice as high as
    // it is wide nIni is 0 and the reference's behaviour is undefined.  There is nothing to restate: the oracle refuses such a
    // level (ORC_E_UNDEFINED from extract), it does not guess.
    if (nIni < 1) throw std::domain_error("DistributeOctTree: round(width / height) == 0 is undefined in the reference");
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.UL = {(int)(hX * (float)i), 0};
        ni.UR = {(int)(hX * (float)(i + 1)), 0};
        ni.BL = {ni.UL.x, maxY - minY};
        ni.BR = {ni.UR.x, maxY - minY};
        ni.seq = seq++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (const KeyPoint& kp : in) ini[(int)(kp.x / hX)]->keys.push_back(kp);
    auto lit = nodes.begin();
    while (lit != nodes.end()) {
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = nodes.erase(lit);
        else ++lit;
    }
    bool finish = false;
    std::vector<SP> sizeAndNode;
    auto push_child = [&](Node& c, int* nToExpand) {
        if (c.keys.empty()) return;
        c.seq = seq++;
        nodes.push_front(c);
        if (c.keys.size() > 1) {
            if (nToExpand) ++*nToExpand;
            sizeAndNode.push_back(std::make_pair((int)c.keys.size(), &nodes.front()));
            nodes.front().lit = nodes.begin();
        }
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        lit = nodes.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (lit != nodes.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            push_child(n1, &nToExpand); push_child(n2, &nToExpand);
            push_child(n3, &nToExpand); push_child(n4, &nToExpand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SP> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end(), sp_less);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    nodes.erase(prev[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    std::vector<KeyPoint> result;
    for (Node& n : nodes) {
        const KeyPoint* best = &n.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > maxResponse) { best = &n.keys[k]; maxResponse = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

// ------------------------------------------------------------------------------------------------
// a6  IC_Angle (:77-104) + cv::fastAtan2 [OCV, App. A-5]
// ------------------------------------------------------------------------------------------------
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void ic_moments(const Image& im, int cx, int cy, const int* umax, int& m01, int& m10) {
    m01 = 0; m10 = 0;
    const uint8_t* center = im.row(cy) + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m10 += u * center[u];
    int step = im.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m10 += u * (val_plus + val_minus);
        }
        m01 += v * v_sum;
    }
}

float ic_angle(const Image& im, float px, float py, const int* umax) {
    int m01, m10;
    ic_moments(im, cv_round(px), cv_round(py), umax, m01, m10);
    return fast_atan2((float)m01, (float)m10);
}

// ------------------------------------------------------------------------------------------------
// a7  GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on an 8-bit clone of the level [OCV 2.4/3.2, App. A-4]
// ------------------------------------------------------------------------------------------------
void gaussian_kernel_7_s2(int k[7]) {
    // getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256) (createSeparableLinearFilter, bits = 8)
    float cf[7];
    double sigma = 2.0, scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_round((double)cf[i] * 256.0);
    }
}

void gaussian_blur7(const Image& src, Image& dst) {
    int k[7];
    gaussian_kernel_7_s2(k);  // = {18,34,49,55,49,34,18}
    const int w = src.w, h = src.h;
    dst.w = w; dst.h = h; dst.px.resize((size_t)w * h);
    std::vector<int> R((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * S[reflect101(x + i, w)];
            R[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * R[(size_t)reflect101(y + i, h) * w + x];
            dst.px[(size_t)y * w + x] = sat_u8((s + 32768) >> 16);
        }
}

// ------------------------------------------------------------------------------------------------
// a8  computeOrbDescriptor (:108-147), with the canonical sin/cos of App. C-3
// ------------------------------------------------------------------------------------------------
void det_sincos(float angle_rad, float& cosv, float& sinv) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
    const double S1 = -1.0 / 6, S2 = 1.0 / 120, S3 = -1.0 / 5040, S4 = 1.0 / 362880, S5 = -1.0 / 39916800,
                 S6 = 1.0 / 6227020800.0, S7 = -1.0 / 1307674368000.0, S8 = 1.0 / 355687428096000.0;
    const double C1 = -1.0 / 2, C2 = 1.0 / 24, C3 = -1.0 / 720, C4 = 1.0 / 40320, C5 = -1.0 / 3628800,
                 C6 = 1.0 / 479001600, C7 = -1.0 / 87178291200.0, C8 = 1.0 / 20922789888000.0;
    double x = (double)angle_rad;
    double kf = std::nearbyint(x * TWO_OVER_PI);
    double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    double z = r * r;
    double ps = S8; ps = ps * z + S7; ps = ps * z + S6; ps = ps * z + S5; ps = ps * z + S4;
    ps = ps * z + S3; ps = ps * z + S2; ps = ps * z + S1;
    double s = r + r * (z * ps);
    double pc = C8; pc = pc * z + C7; pc = pc * z + C6; pc = pc * z + C5; pc = pc * z + C4;
    pc = pc * z + C3; pc = pc * z + C2; pc = pc * z + C1;
    double c = 1.0 + z * pc;
    int q = ((int)kf) & 3;
    double cc = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    double ss = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cosv = (float)cc;
    sinv = (float)ss;
}

void orb_descriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = kpt.angle * factorPI;
    float a, b;
    det_sincos(angle, a, b);  // reference: a = (float)cos(angle), b = (float)sin(angle)  (:113)
    const uint8_t* center = img.row(cv_round(kpt.y)) + cv_round(kpt.x);
    const int step = img.w;
    auto tap = [&](int px, int py) -> int {
        float fx = (float)px, fy = (float)py;
        int r = cv_round((double)(fx * b + fy * a));
        int c = cv_round((double)(fx * a - fy * b));
        return center[r * step + c];
    };
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const Quad& q = kPattern[8 * i + k];
            int t0 = tap(q.x0, q.y0), t1 = tap(q.x1, q.y1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

// ------------------------------------------------------------------------------------------------
// a4  ComputeKeyPointsOctTree (:766-854): literal per-cell cv::FAST calls
// ------------------------------------------------------------------------------------------------
void cell_candidates(const Params& P, const Image& im, std::vector<KeyPoint>& toDistribute) {
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    toDistribute.clear();
    std::vector<KeyPoint> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
            const uint8_t* view = im.row(y0) + x0;
            fast_9_16_nms(view, x1 - x0, y1 - y0, im.w, P.iniTh, cell);
            if (cell.empty()) fast_9_16_nms(view, x1 - x0, y1 - y0, im.w, P.minTh, cell);
            for (KeyPoint& kp : cell) {
                kp.x += j * wCell;
                kp.y += i * hCell;
                toDistribute.push_back(kp);
            }
        }
    }
}

void level_keypoints(const Params& P, const Image& im, int level, std::vector<KeyPoint>& kps) {
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
    std::vector<KeyPoint> cand;
    cell_candidates(P, im, cand);
    kps = distribute_octree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, P.quota[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * P.scale[level]);
    for (KeyPoint& kp : kps) {
        kp.x += minBorderX; kp.y += minBorderY;
        kp.octave = level;
        kp.size = (float)scaledPatchSize;
    }
    for (KeyPoint& kp : kps) kp.angle = ic_angle(im, kp.x, kp.y, P.umax);  // computeOrientation :473-480
}

// ------------------------------------------------------------------------------------------------
// a2  ORBextractor::operator() (:1044-1107)
// ------------------------------------------------------------------------------------------------
constexpr int ORC_E_UNDEFINED = INT32_MIN;   // the reference's own behaviour is undefined for this input (see distribute_octree)
int extract(const Params& P, const uint8_t* img, int W, int H, int stride, KeyPoint* kps_out, uint8_t* desc_out,
            int cap) {
    if (!img || W <= 0 || H <= 0) return 0;
    std::vector<Image> pyr;
    compute_pyramid(P, img, W, H, stride, pyr);
    std::vector<std::vector<KeyPoint>> all(P.nlevels);
    try {
        for (int l = 0; l < P.nlevels; l++) level_keypoints(P, pyr[l], l, all[l]);
    } catch (const std::domain_error&) {
        return ORC_E_UNDEFINED;
    }
    int n = 0;
    for (int l = 0; l < P.nlevels; l++) n += (int)all[l].size();
    if (n > cap) return -n;
    int offset = 0;
    for (int l = 0; l < P.nlevels; l++) {
        std::vector<KeyPoint>& kps = all[l];
        if (kps.empty()) continue;
        Image blurred;
        gaussian_blur7(pyr[l], blurred);
        for (size_t i = 0; i < kps.size(); i++) orb_descriptor(kps[i], blurred, desc_out + (size_t)(offset + i) * 32);
        if (l != 0) {
            float scale = P.scale[l];
            for (KeyPoint& kp : kps) { kp.x *= scale; kp.y *= scale; }
        }
        std::memcpy(kps_out + offset, kps.data(), kps.size() * sizeof(KeyPoint));
        offset += (int)kps.size();
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
// a9  ORBmatcher::DescriptorDistance (ORBmatcher.cc:3994-4010)
// ------------------------------------------------------------------------------------------------
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        int32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        unsigned int v = (unsigned int)(pa ^ pb);
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// ------------------------------------------------------------------------------------------------
// a14 Frame grid (Frame.cc:348-395, :632-642) and GetFeaturesInArea(cam, ...) (:574-629)  [App. A-8]
// ------------------------------------------------------------------------------------------------
const int GRID_COLS = 64, GRID_ROWS = 48;  // Frame.h:37-38

struct FrameView {  // flat view of the Frame members the matcher reads
    int n_total;              // N_total
    const float* un_x;        // mvKeysUn_total[g].pt.x
    const float* un_y;
    const int* octave;        // mvKeysUn_total[g].octave
    const float* angle;       // mvKeysUn_total[g].angle
    const float* uright;      // mvuRight_total[g]
    const int* cam_of;        // keypoint_to_cam[g]
    const int* local_of;      // cont_idx_to_local_cam_idx[g]
    const uint8_t* const* desc;  // mDescriptors_total[cam] (N_cam x 32)
    float minX, minY, maxX, maxY, invW, invH;
    int n_cams;
    std::vector<std::vector<int>> grid;  // [cam][ix*GRID_ROWS+iy] -> ascending global indices
};

void build_grid(FrameView& F) {
    F.grid.assign((size_t)F.n_cams * GRID_COLS * GRID_ROWS, {});
    for (int g = 0; g < F.n_total; g++) {
        int posX = (int)std::round((F.un_x[g] - F.minX) * F.invW);
        int posY = (int)std::round((F.un_y[g] - F.minY) * F.invH);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        F.grid[((size_t)F.cam_of[g] * GRID_COLS + posX) * GRID_ROWS + posY].push_back(g);
    }
}

void features_in_area(const FrameView& F, int cam, float x, float y, float r, int minLevel, int maxLevel,
                      std::vector<int>& out) {
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - F.minX - r) * F.invW));
    if (nMinCellX >= GRID_COLS) return;
    const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - F.minX + r) * F.invW));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - F.minY - r) * F.invH));
    if (nMinCellY >= GRID_ROWS) return;
    const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - F.minY + r) * F.invH));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int>& cell = F.grid[((size_t)cam * GRID_COLS + ix) * GRID_ROWS + iy];
            for (int g : cell) {
                if (bCheckLevels) {
                    if (F.octave[g] < minLevel) continue;
                    if (maxLevel >= 0 && F.octave[g] > maxLevel) continue;
                }
                const float distx = F.un_x[g] - x, disty = F.un_y[g] - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(g);
            }
        }
}

// a13 ComputeThreeMaxima (ORBmatcher.cc:3948-3989)
void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

// =================================================================================================
// flat C surface (ctypes)
// =================================================================================================
extern "C" {

struct orc_query {  // one projected last-frame map point (App. A-10: projection stays on the host)
    float u, v, radius, ur;   // ur = u - mbf*invzc (ORBmatcher.cc:3573)
    int min_level, max_level; // arguments handed to GetFeaturesInArea (:3547-3552)
    int cam;
    int blocks;               // 1 if the query's MapPoint has Observations()>0 (a claim by it hides the feature)
    float angle;              // LastFrame.mvKeysUn_total[i].angle (:3604)
    uint8_t desc[32];         // pMP->GetDescriptor()
};

struct orc_frame {
    int n_total, n_cams;
    const float* un_x; const float* un_y; const int* octave; const float* angle; const float* uright;
    const int* cam_of; const int* local_of;
    const uint8_t* const* desc;
    float minX, minY, maxX, maxY;
};

int orc_keypoint_size() { return (int)sizeof(KeyPoint); }

void orc_tables(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, float* scale, float* inv_scale,
                float* sigma2, float* inv_sigma2, int* quota, int* umax16) {
    Params P; init_params(P, nfeatures, scaleFactor, nlevels, iniTh, minTh);
    for (int i = 0; i < nlevels; i++) {
        scale[i] = P.scale[i]; inv_scale[i] = P.inv_scale[i]; sigma2[i] = P.sigma2[i];
        inv_sigma2[i] = P.inv_sigma2[i]; quota[i] = P.quota[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = P.umax[i];
}

void orc_level_sizes(int W, int H, float scaleFactor, int nlevels, int* w, int* h) {
    Params P; init_params(P, 1000, scaleFactor, nlevels, 20, 7);
    for (int l = 0; l < nlevels; l++) level_size(P, W, H, l, w[l], h[l]);
}

void orc_pattern(signed char* out1024) { std::memcpy(out1024, kPattern, 1024); }

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}

void orc_copy_make_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border,
                                     int dstride) {
    copy_make_border_reflect101(src, w, h, sstride, dst, border, dstride);
}

// Pyramid: levels written back-to-back, dense, into out (caller sizes it with orc_level_sizes).
void orc_pyramid(const uint8_t* img, int W, int H, int stride, float scaleFactor, int nlevels, uint8_t* out) {
    Params P; init_params(P, 1000, scaleFactor, nlevels, 20, 7);
    std::vector<Image> pyr; compute_pyramid(P, img, W, H, stride, pyr);
    size_t off = 0;
    for (auto& L : pyr) { std::memcpy(out + off, L.px.data(), L.px.size()); off += L.px.size(); }
}

int orc_corner_score(const uint8_t* img, int stride, int x, int y, int threshold) {
    return corner_score16(img + (size_t)y * stride + x, stride, threshold);
}
int orc_is_corner(const uint8_t* img, int stride, int x, int y, int threshold) {
    return is_corner16(img + (size_t)y * stride + x, stride, threshold) ? 1 : 0;
}

int orc_fast(const uint8_t* view, int cols, int rows, int stride, int threshold, KeyPoint* out, int cap) {
    std::vector<KeyPoint> v; fast_9_16_nms(view, cols, rows, stride, threshold, v);
    int n = (int)std::min<size_t>(v.size(), (size_t)cap);
    if (n > 0) std::memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return (int)v.size();
}

// Candidates handed to the quadtree for one level image (coordinates relative to (16,16)), :790-830.
int orc_cell_candidates(const uint8_t* img, int w, int h, int iniTh, int minTh, KeyPoint* out, int cap) {
    Params P; init_params(P, 1000, 1.2f, 8, iniTh, minTh);
    Image im; im.w = w; im.h = h; im.px.assign(img, img + (size_t)w * h);
    std::vector<KeyPoint> v; cell_candidates(P, im, v);
    int n = (int)std::min<size_t>(v.size(), (size_t)cap);
    if (n > 0) std::memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return (int)v.size();
}

int orc_distribute_octree(const KeyPoint* in, int n_in, int minX, int maxX, int minY, int maxY, int N, KeyPoint* out,
                          int cap) {
    std::vector<KeyPoint> v(in, in + n_in);
    std::vector<KeyPoint> r;
    try { r = distribute_octree(v, minX, maxX, minY, maxY, N); } catch (const std::domain_error&) { return INT32_MIN; }   // (ORC_E_UNDEFINED)
    int n = (int)std::min<size_t>(r.size(), (size_t)cap);
    if (n > 0) std::memcpy(out, r.data(), (size_t)n * sizeof(KeyPoint));
    return (int)r.size();
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_cv_round(double v) { return cv_round(v); }

float orc_ic_angle(const uint8_t* img, int w, int h, int x, int y, int* m01, int* m10) {
    Params P; init_params(P, 1000, 1.2f, 8, 20, 7);
    Image im; im.w = w; im.h = h; im.px.assign(img, img + (size_t)w * h);
    int a, b; ic_moments(im, x, y, P.umax, a, b);
    if (m01) *m01 = a;
    if (m10) *m10 = b;
    return fast_atan2((float)a, (float)b);
}

void orc_gaussian_kernel(int* k7) { gaussian_kernel_7_s2(k7); }

void orc_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
    Image s, d; s.w = w; s.h = h; s.px.assign(src, src + (size_t)w * h);
    gaussian_blur7(s, d);
    std::memcpy(dst, d.px.data(), d.px.size());
}

void orc_det_sincos(float angle_rad, float* c, float* s) { det_sincos(angle_rad, *c, *s); }

// descriptor of one keypoint (level coordinates, integral) on an ALREADY BLURRED level image
void orc_orb_descriptor(const uint8_t* blurred, int w, int h, float x, float y, float angle_deg, uint8_t* desc32) {
    Image im; im.w = w; im.h = h; im.px.assign(blurred, blurred + (size_t)w * h);
    KeyPoint kp{x, y, 31.f, angle_deg, 0.f, 0, -1};
    orb_descriptor(kp, im, desc32);
}

// Full ORBextractor::operator().  Returns N (>=0), or -N if cap is too small.
int orc_extract(const uint8_t* img, int W, int H, int stride, int nfeatures, float scaleFactor, int nlevels, int iniTh,
                int minTh, KeyPoint* kps_out, uint8_t* desc_out, int cap) {
    Params P; init_params(P, nfeatures, scaleFactor, nlevels, iniTh, minTh);
    return extract(P, img, W, H, stride, kps_out, desc_out, cap);
}

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) as Frame::UndistortKeyPoints calls it (src/Frame.cc:692):
// the OpenCV 2.4.x / 3.2 generic path (modules/imgproc/src/undistort.cpp, cvUndistortPoints) restated -- K and the
// distortion vector are CV_32F in the reference (src/Tracking.cc) and are promoted to double, the point is promoted to
// double, normalised, run through FIVE fixed-point iterations of the radial-tangential model (k4..k6 and the thin-prism
// terms are zero for the reference's 4/5-element vector, so the numerator of icdist is exactly 1), re-projected with
// P = K (R = I, so ww == 1) and rounded to float.  calib = {fx, fy, cx, cy, k1, k2, p1, p2, k3} as floats.
static void undistort_point(const float* calib, float xs, float ys, float* xo, float* yo) {
    const double fx = calib[0], fy = calib[1], cx = calib[2], cy = calib[3];
    const double k1 = calib[4], k2 = calib[5], p1 = calib[6], p2 = calib[7], k3 = calib[8];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = xs, y = ys;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx;   // RR = K * I: the zero entries take part in the sum as they do in OpenCV
    const double yy = 0. * x + fy * y + cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww); *yo = (float)(yy * ww);
}

// Frame::UndistortKeyPoints (src/Frame.cc:673-705): a plain copy when k1 == 0.
void orc_undistort_points(const float* calib, const float* x, const float* y, int n, float* ux, float* uy) {
    for (int i = 0; i < n; i++) {
        if (calib == nullptr || calib[4] == 0.0f) { ux[i] = x[i]; uy[i] = y[i]; }
        else undistort_point(calib, x[i], y[i], &ux[i], &uy[i]);
    }
}

// Frame::ComputeImageBounds (src/Frame.cc:743-778): out = {minX, minY, maxX, maxY}.
void orc_image_bounds(const float* calib, int cols, int rows, float* out4) {
    if (calib == nullptr || calib[4] == 0.0f) { out4[0] = 0.f; out4[1] = 0.f; out4[2] = (float)cols; out4[3] = (float)rows; return; }
    const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) undistort_point(calib, cx[i], cy[i], &ux[i], &uy[i]);
    out4[0] = std::min(ux[0], ux[2]); out4[2] = std::max(ux[1], ux[3]);
    out4[1] = std::min(uy[0], uy[1]); out4[3] = std::max(uy[2], uy[3]);
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:959-986): depth lookup at (int)v,(int)u of the DISTORTED keypoint, virtual
// right coordinate from the UNDISTORTED x (un_x == NULL: k1 == 0, undistorted == distorted, src/Frame.cc:676-680).
void orc_stereo_from_depth(const KeyPoint* kps, int n, const float* depth, int stride, float mbf, float* uright,
                           float* depth_out, const float* un_x) {
    for (int i = 0; i < n; i++) {
        uright[i] = -1; depth_out[i] = -1;
        const float v = kps[i].y, u = kps[i].x;
        const float d = depth[(size_t)(int)v * stride + (int)u];
        if (d > 0) { depth_out[i] = d; uright[i] = (un_x ? un_x[i] : kps[i].x) - mbf / d; }
    }
}

// a12 brute-force top-2 (ORBmatcher.cc:287-321 inner loop over an unrestricted candidate set, App. A-9):
// best_idx = -1 / best = second = 256 when nothing is closer than 256.
void orc_bf_top2(const uint8_t* q, int nq, const uint8_t* r, int nr, int* best_idx, int* best_dist, int* second_dist) {
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nr; j++) {
            const int d = descriptor_distance(q + (size_t)i * 32, r + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) { b2 = d; }
        }
        best_idx[i] = bi; best_dist[i] = b1; second_dist[i] = b2;
    }
}

void orc_hamming_matrix(const uint8_t* q, int nq, const uint8_t* r, int nr, uint16_t* out) {
    for (int i = 0; i < nq; i++)
        for (int j = 0; j < nr; j++)
            out[(size_t)i * nr + j] = (uint16_t)descriptor_distance(q + (size_t)i * 32, r + (size_t)j * 32);
}

void orc_three_maxima(const int* sizes, int L, int* ind) {
    int a = -1, b = -1, c = -1; three_maxima(sizes, L, a, b, c);
    ind[0] = a; ind[1] = b; ind[2] = c;
}

static void make_view(const orc_frame* f, FrameView& F) {
    F.n_total = f->n_total; F.n_cams = f->n_cams; F.un_x = f->un_x; F.un_y = f->un_y; F.octave = f->octave;
    F.angle = f->angle; F.uright = f->uright; F.cam_of = f->cam_of; F.local_of = f->local_of; F.desc = f->desc;
    F.minX = f->minX; F.minY = f->minY; F.maxX = f->maxX; F.maxY = f->maxY;
    F.invW = (float)GRID_COLS / (F.maxX - F.minX);  // Frame.cc:271-272
    F.invH = (float)GRID_ROWS / (F.maxY - F.minY);
    build_grid(F);
}

// Grid as CSR (cell = (cam*64 + ix)*48 + iy), for checking the product's grid builder.
void orc_grid_csr(const orc_frame* f, int* cell_start /*n_cams*3072+1*/, int* items /*n_total*/) {
    FrameView F; make_view(f, F);
    int off = 0;
    for (size_t c = 0; c < F.grid.size(); c++) {
        cell_start[c] = off;
        for (int g : F.grid[c]) items[off++] = g;
    }
    cell_start[F.grid.size()] = off;
}

int orc_features_in_area(const orc_frame* f, int cam, float x, float y, float r, int minLevel, int maxLevel, int* out,
                         int cap) {
    FrameView F; make_view(f, F);
    std::vector<int> v; features_in_area(F, cam, x, y, r, minLevel, maxLevel, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

// a10  ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono, Calib)  (ORBmatcher.cc:3448-3641),
// from the projected queries on (lines :3502-3552 happen on the host, App. A-10).
//   match_of_feature[g]  : index of the query whose MapPoint ends up in CurrentFrame.mvpMapPoints[g], or -1
//   returns nmatches exactly as the reference counts it (overwritten claims stay counted, :3597-3598).
//   occupied[g] != 0 where CurrentFrame.mvpMapPoints[g] already holds an observed point before the call (may be NULL;
//   TrackWithMotionModel clears the vector first, src/Tracking.cc:1254).
//   match_of_feature[g]: >= 0 query index, -1 untouched, -2 set to NULL by the rotation-histogram filter (:3631).
int orc_search_by_projection_frames(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied,
                                    int th_high, int check_ori, int* match_of_feature) {
    FrameView F; make_view(cur, F);
    const int HISTO_LENGTH = 30;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int g = 0; g < F.n_total; g++) match_of_feature[g] = -1;
    int nmatches = 0;
    std::vector<int> cand;
    for (int i = 0; i < nq; i++) {
        const orc_query& Q = q[i];
        features_in_area(F, Q.cam, Q.u, Q.v, Q.radius, Q.min_level, Q.max_level, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (occupied && occupied[i2] && match_of_feature[i2] < 0) continue;          // pre-existing observed point
            if (match_of_feature[i2] >= 0 && q[match_of_feature[i2]].blocks) continue;  // :3566-3568
            if (F.uright[i2] > 0) {                                                      // :3571-3577
                const float er = std::fabs(Q.ur - F.uright[i2]);
                if (er > Q.radius) continue;
            }
            const uint8_t* d = F.desc[Q.cam] + (size_t)F.local_of[i2] * 32;              // :3580-3582
            const int dist = descriptor_distance(Q.desc, d);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= th_high) {
            match_of_feature[bestIdx2] = i;
            nmatches++;
            if (check_ori) {
                float rot = Q.angle - F.angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int g : rotHist[i]) { match_of_feature[g] = -2; nmatches--; }
    }
    return nmatches;
}

// f4  SearchByProjection(KeyFrame*, Scw, vpPoints, vLoopMPCams, vpMatched, th, Calib), ORBmatcher.cc:566-750: the two-camera
// loop search from the projected windows on.  Point i: first window = q[i] (cam < 0: the point failed the gates of :636-672
// for camera 1), second window = w2[i]; GetFeaturesInArea is called WITHOUT level arguments (:680), the level gate
// [nPredictedLevel-1, nPredictedLevel] (:704-707) is applied inside the candidate loop; bestDist / bestIdxs are shared by the
// two cameras (:624-625 sit outside the `for camidx` loop); `vpMatched[idx]` (:696) hides both the features that were matched
// before the call (occupied) and the ones matched by earlier points of this call.
struct orc_window { float u, v, radius; int cam, min_level, max_level; };

int orc_search_by_projection_loop2(const orc_frame* cur, const orc_query* q, const orc_window* w2, int nq, const uint8_t* occupied,
                                   int th_low, int* match_of_feature) {
    FrameView F; make_view(cur, F);
    for (int g = 0; g < F.n_total; g++) match_of_feature[g] = -1;
    int nmatches = 0;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nq; iMP++) {
        int bestDist = 256;
        int bestIdxs = -1;
        for (int camidx = 0; camidx < 2; ++camidx) {
            float u, v, radius; int cam, lo, hi;
            if (camidx == 0) { u = q[iMP].u; v = q[iMP].v; radius = q[iMP].radius; cam = q[iMP].cam; lo = q[iMP].min_level; hi = q[iMP].max_level; }
            else { u = w2[iMP].u; v = w2[iMP].v; radius = w2[iMP].radius; cam = w2[iMP].cam; lo = w2[iMP].min_level; hi = w2[iMP].max_level; }
            if (cam < 0) continue;                                            // one of the `continue`s of :636-672
            features_in_area(F, cam, u, v, radius, -1, -1, vIndices);         // :680
            if (vIndices.empty()) continue;
            for (int idx : vIndices) {
                if ((occupied && occupied[idx]) || match_of_feature[idx] >= 0) continue;   // if(vpMatched[idx]) :696
                const int kpLevel = F.octave[idx];
                if (kpLevel < lo || kpLevel > hi) continue;                                // :704-707
                const uint8_t* dKF = F.desc[cam] + (size_t)F.local_of[idx] * 32;           // :710-711
                const int dist = descriptor_distance(q[iMP].desc, dKF);
                if (dist < bestDist) { bestDist = dist; bestIdxs = idx; }
            }
        }
        if (bestDist <= th_low) { match_of_feature[bestIdxs] = iMP; nmatches++; }          // :732-736
    }
    return nmatches;
}

// f4  SearchForInitialization, ORBmatcher.cc:868-994, from the level-0 keypoints of F1 on (one query per kept keypoint, in
// order: window centre = vbPrevMatched, radius = windowSize, levels level1..level1, descriptor and angle of the keypoint).
// match12[k] = index in F2 matched to query k or -1 (vnMatches12 restricted to the kept keypoints); returns nmatches.
int orc_search_for_initialization(const orc_frame* f2, const orc_query* q, int nq, float nnratio, int check_ori, int th_low,
                                  int* match12) {
    FrameView F; make_view(f2, F);
    const int HISTO_LENGTH = 30;
    int nmatches = 0;
    for (int k = 0; k < nq; k++) match12[k] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(F.n_total, INT_MAX);
    std::vector<int> vnMatches21(F.n_total, -1);
    std::vector<int> vIndices2;
    for (int i1 = 0; i1 < nq; i1++) {
        const orc_query& Q = q[i1];
        features_in_area(F, Q.cam, Q.u, Q.v, Q.radius, Q.min_level, Q.max_level, vIndices2);   // :887
        if (vIndices2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const uint8_t* d2 = F.desc[Q.cam] + (size_t)F.local_of[i2] * 32;
            const int dist = descriptor_distance(Q.desc, d2);
            if (vMatchedDistance[i2] <= dist) continue;                                       // :906-907
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= th_low) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { match12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                match12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    float rot = Q.angle - F.angle[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (match12[idx1] >= 0) { match12[idx1] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// f4  the inner loop shared by SearchBySim3 (ORBmatcher.cc:2814-3135) and Fuse (:1986-2509): nearest candidate of every
// projected point on its own (no claims).  gate 0: none; 1: the tracking right-coordinate window (:3571-3577);
// 2: Fuse's reprojection-error gate (:2118-2143) with q.ur = the projected right coordinate.
void orc_project_best(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied, int gate, const float* inv_sigma2,
                      int* best_idx, int* best_dist) {
    FrameView F; make_view(cur, F);
    std::vector<int> cand;
    for (int i = 0; i < nq; i++) {
        const orc_query& Q = q[i];
        features_in_area(F, Q.cam, Q.u, Q.v, Q.radius, Q.min_level, Q.max_level, cand);
        int bestDist = 256, bestIdx = -1;
        for (int idx : cand) {
            if (occupied && occupied[idx]) continue;
            if (gate == 1 && F.uright[idx] > 0) {
                const float er = std::fabs(Q.ur - F.uright[idx]);
                if (er > Q.radius) continue;
            }
            if (gate == 2) {
                const float kpx = F.un_x[idx], kpy = F.un_y[idx], kpr = F.uright[idx];
                const int kpLevel = F.octave[idx];
                const float ex = Q.u - kpx, ey = Q.v - kpy;
                if (kpr >= 0) {
                    const float er = Q.ur - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const uint8_t* d = F.desc[Q.cam] + (size_t)F.local_of[idx] * 32;
            const int dist = descriptor_distance(Q.desc, d);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
}

// a11  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)  (ORBmatcher.cc:62-149).
// Query fields: u,v = mTrackProjX/Y; radius = r*mvScaleFactors[level] (:82-89); ur = mTrackProjXR;
// min_level/max_level = level-1, level; cam = 0 (camera-1 grid only, Frame.cc:510-563); blocks as above.
// `occupied[g]` = 1 where F.mvpMapPoints[g] already holds a point with Observations()>0 before the call.
int orc_search_by_projection_points(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied,
                                    float nnratio, int th_high, int* match_of_feature) {
    FrameView F; make_view(cur, F);
    for (int g = 0; g < F.n_total; g++) match_of_feature[g] = -1;
    int nmatches = 0;
    std::vector<int> cand;
    for (int i = 0; i < nq; i++) {
        const orc_query& Q = q[i];
        features_in_area(F, 0, Q.u, Q.v, Q.radius, Q.min_level, Q.max_level, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
            if (occupied && occupied[idx]) continue;                                       // :107-109
            if (match_of_feature[idx] >= 0 && q[match_of_feature[idx]].blocks) continue;
            if (F.uright[idx] > 0) {                                                       // :111-116
                const float er = std::fabs(Q.ur - F.uright[idx]);
                if (er > Q.radius) continue;
            }
            const uint8_t* d = F.desc[0] + (size_t)idx * 32;                               // :118 (F.mDescriptors = cam 1)
            const int dist = descriptor_distance(Q.desc, d);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.octave[idx]; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.octave[idx]; bestDist2 = dist;
            }
        }
        if (bestDist <= th_high) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match_of_feature[bestIdx] = i;
            nmatches++;
        }
    }
    return nmatches;
}

}  // extern "C"
