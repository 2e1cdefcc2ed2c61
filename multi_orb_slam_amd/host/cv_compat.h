// cv_compat.h -- the handful of OpenCV types the ORB front end's public signatures mention, for builds without OpenCV
// (this image has none).  Define HAVE_OPENCV to use the real headers instead; the wrappers only rely on members that
// exist in both (cv::InputArray / cv::OutputArray are proxy classes here as they are there: getMat(), create(), release()).  cv::KeyPoint is byte-compatible with orb_keypoint (28 bytes), which the wrappers static_assert.
#pragma once
#ifdef HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point { int x = 0, y = 0; };

struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};

// Dense 2-D matrix of uchar or float with shared storage and row/column views (enough for descriptors, 3x3 / 4x4
// pose algebra and 8-bit images).
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    unsigned char* data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), type_(type) {
        step = step_ ? step_ : (size_t)c * elemSize(); data = (unsigned char*)ext;
    }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, m.step * r); return m; }
    static Mat eye(int r, int c, int type) { Mat m = zeros(r, c, type); for (int i = 0; i < r && i < c; ++i) m.at<float>(i, i) = 1.f; return m; }
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && owner_) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
        const size_t bytes = step * (size_t)(r > 0 ? r : 1);
        if (bytes <= sizeof(SmallBlock)) {
            // poses, points and descriptors (<= 64 bytes) are what the matcher's projection loops create by the thousand: one
            // pooled block holds the reference count and the payload
            std::shared_ptr<SmallBlock> b = std::allocate_shared<SmallBlock>(PoolAlloc<SmallBlock>());
            owner_ = std::shared_ptr<unsigned char>(b, b->d);
        } else {
            owner_.reset(new unsigned char[bytes], std::default_delete<unsigned char[]>());
        }
        data = owner_.get();
    }
    void release() { owner_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * r); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * r); }
    unsigned char* ptr(int r = 0) { return data + step * r; }
    const unsigned char* ptr(int r = 0) const { return data + step * r; }
    template <typename T> T& at(int r, int c = 0) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c = 0) const { return ptr<T>(r)[c]; }
    // single-index access on column vectors, like cv::Mat::at<T>(i)
    Mat row(int r) const { return view(r, r + 1, 0, cols); }
    Mat col(int c) const { return view(0, rows, c, c + 1); }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
        return m;
    }
    // 3x3 CV_32F inverse the way cv::invert(DECOMP_LU) special-cases it: determinant and cofactors in double, one rounding
    // to float per element (modules/core/src/lapack.cpp); a singular matrix gives zeros.
    Mat inv() const {
        assert(type_ == CV_32F && rows == 3 && cols == 3);
        const Mat& S = *this;
        auto m = [&](int r, int c) { return (double)S.at<float>(r, c); };
        double d = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
                   m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
        Mat D = zeros(3, 3, CV_32F);
        if (d != 0.) {
            d = 1. / d;
            D.at<float>(0, 0) = (float)((m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * d);
            D.at<float>(0, 1) = (float)((m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * d);
            D.at<float>(0, 2) = (float)((m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * d);
            D.at<float>(1, 0) = (float)((m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * d);
            D.at<float>(1, 1) = (float)((m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * d);
            D.at<float>(1, 2) = (float)((m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * d);
            D.at<float>(2, 0) = (float)((m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * d);
            D.at<float>(2, 1) = (float)((m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * d);
            D.at<float>(2, 2) = (float)((m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * d);
        }
        return D;
    }
    // a.dot(b): products summed in double (cv::Mat::dot for CV_32F)
    double dot(const Mat& b) const {
        assert(type_ == CV_32F && rows * cols == b.rows * b.cols);
        double s = 0;
        for (int k = 0; k < rows * cols; ++k) s += (double)at<float>(k / cols, k % cols) * (double)b.at<float>(k / b.cols, k % b.cols);
        return s;
    }
    Mat t() const {
        assert(type_ == CV_32F);
        Mat m(cols, rows, CV_32F);
        for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }
private:
    struct SmallBlock { alignas(16) unsigned char d[64]; };
    // thread-local free list of fixed-size blocks (shared_ptr control block + SmallBlock); a block freed by another thread
    // simply joins that thread's list
    template <class T> struct PoolAlloc {
        using value_type = T;
        PoolAlloc() {}
        template <class U> PoolAlloc(const PoolAlloc<U>&) {}
        static constexpr size_t BLOCK = 160;
        struct Node { Node* next; };
        struct FreeList {   // the thread's pooled blocks go back to the heap when the thread exits
            Node* h = nullptr;
            ~FreeList() { while (h) { Node* p = h; h = p->next; ::operator delete(p); } }
        };
        static Node*& head() { static thread_local FreeList l; return l.h; }
        T* allocate(size_t n) {
            if (n == 1 && sizeof(T) <= BLOCK) {
                Node*& h = head();
                if (h) { Node* p = h; h = p->next; return reinterpret_cast<T*>(p); }
                return static_cast<T*>(::operator new(BLOCK));
            }
            return static_cast<T*>(::operator new(n * sizeof(T)));
        }
        void deallocate(T* p, size_t n) {
            if (n == 1 && sizeof(T) <= BLOCK) { Node* q = reinterpret_cast<Node*>(p); q->next = head(); head() = q; return; }
            ::operator delete(p);
        }
        template <class U> bool operator==(const PoolAlloc<U>&) const { return true; }
        template <class U> bool operator!=(const PoolAlloc<U>&) const { return false; }
    };
    Mat view(int r0, int r1, int c0, int c1) const {
        Mat m; m.rows = r1 - r0; m.cols = c1 - c0; m.type_ = type_; m.step = step; m.owner_ = owner_;
        m.data = data + step * r0 + (size_t)c0 * elemSize();
        return m;
    }
    int type_ = CV_8U;
    std::shared_ptr<unsigned char> owner_;
};

// float matrix product / sum with double accumulation then rounding to float (what cv::gemm does for CV_32F)
inline Mat operator*(const Mat& a, const Mat& b) {
    assert(a.cols == b.rows);
    Mat m(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < b.cols; ++j) {
            double s = 0;
            for (int k = 0; k < a.cols; ++k) s += (double)a.at<float>(i, k) * (double)b.at<float>(k, j);
            m.at<float>(i, j) = (float)s;
        }
    return m;
}
inline Mat operator+(const Mat& a, const Mat& b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j);
    return m;
}
inline Mat operator-(const Mat& a) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = -a.at<float>(i, j);
    return m;
}

inline Mat operator-(const Mat& a, const Mat& b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
    return m;
}
// scaling by a scalar: cv::MatExpr evaluates alpha*M (and M/s as (1/s)*M) through convertTo, whose float->float path works in
// float with alpha rounded to float
inline Mat operator*(double alpha, const Mat& a) {
    Mat m(a.rows, a.cols, CV_32F);
    const float al = (float)alpha;
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) * al;
    return m;
}
inline Mat operator/(const Mat& a, double s) { return (1.0 / s) * a; }
// L2 norm: squares summed in double, square root in double (cv::norm, NORM_L2, CV_32F)
inline double norm(const Mat& a) {
    double s = 0;
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) s += (double)a.at<float>(i, j) * (double)a.at<float>(i, j);
    return std::sqrt(s);
}

// Proxy argument types with the members of OpenCV's own cv::_InputArray / cv::_OutputArray that the wrappers use (getMat,
// empty, create, release): the same wrapper source compiles against these stand-ins and against the real headers.
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }   // a header sharing the caller's pixels, as in OpenCV
    bool empty() const { return !m_ || m_->empty(); }
private:
    const Mat* m_ = nullptr;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int rows, int cols, int type) const { m_->create(rows, cols, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
