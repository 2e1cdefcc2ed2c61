// cv_compat.h -- the handful of OpenCV types the ORB front end's public signatures mention, for builds without OpenCV
// (this image has none).  Define HAVE_OPENCV to use the real headers instead; the wrappers only rely on members that
// exist in both (cv::InputArray / cv::OutputArray are proxy classes here as they are there: getMat(), create(), release()).  cv::KeyPoint is byte-compatible with orb_keypoint (28 bytes), which the wrappers static_assert.
#pragma once
#ifdef HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };
typedef Point Point2i;
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
template <typename T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
    Vec(T a, T b, T c) { static_assert(N == 3, "three-element form"); val[0] = a; val[1] = b; val[2] = c; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<float, 3> Vec3f;
template <typename T> struct DataType;
template <> struct DataType<unsigned char> { enum { type = CV_8U }; };
template <> struct DataType<float> { enum { type = CV_32F }; };
class MatExpr;

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
    void copyTo(Mat& m) const {   // (re)allocates m unless it already has this size and type, like cv::Mat::copyTo
        if (empty()) { m.release(); return; }
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memmove(m.ptr(r), ptr(r), (size_t)cols * elemSize());
    }
    // M.inv() and M.t() are expressions, as in OpenCV (MatExpr below): what they evaluate to depends on what they are
    // multiplied with (inv() * B is cv::solve, t() * B a gemm flag).
    inline MatExpr inv() const;
    inline MatExpr t() const;
    // 3x3 CV_32F inverse the way cv::invert(DECOMP_LU) special-cases it: determinant and cofactors in double, one rounding
    // to float per element (modules/core/src/lapack.cpp, cv::invert, "else if( n == 3 )" branch); a singular matrix gives zeros.
    Mat inverted() const {
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
    Mat transposed() const {
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

template <typename T> class Mat_ : public Mat {
public:
    Mat_() {}
    Mat_(int r, int c) : Mat(r, c, DataType<T>::type) {}
};

// ---- cv::gemm for CV_32F, restated from OpenCV 2.4.x / 3.2 modules/core/src/matmul.cpp (the versions the reference's
// CMakeLists.txt:33-38 accepts).  gemm has TWO evaluation orders and picks by shape:
//   * gemm_small_f32: the block in cv::gemm headed `if( flags == 0 && 2 <= len && len <= 4 && (len == d_size.width ||
//     len == d_size.height) )`, CV_32F arm: every element is `float t = a0*b0 + a1*b1 [+ a2*b2 [+ a3*b3]]` -- products and sums in
//     FLOAT, left to right -- then `d = (float)(t*alpha + c*beta)` with alpha, beta double and c = 0.0f when there is no C.
//     This is the path of every R*x+t, R*R and T*T of the matcher (3x3 * 3x1, 3x3 * 3x3, 4x4 * 4x4, no transposition flag).
//   * gemm_general_f32: everything else goes to GEMMSingleMul<float,double> (same file): products and sums in DOUBLE from 0.0
//     (four interleaved partial sums over k for A*Bt when len >= 4, one running sum otherwise), `s*alpha`, `+ double(c)*beta`,
//     ONE rounding to float.  Transposition flags (A.t()*B arrives here as GEMM_1_T) and lengths outside 2..4 take this path.
// Neither is contracted (no FMA: distribution builds of OpenCV target baseline x86-64; this file is compiled with
// -ffp-contract=off).  host/test_host `gemm` holds both against known answers where the two orders differ.
enum { GEMM_1_T = 1, GEMM_2_T = 2, GEMM_3_T = 4 };

inline bool gemm_takes_small_path(int len, int d_rows, int d_cols, int flags) {
    return flags == 0 && 2 <= len && len <= 4 && (len == d_cols || len == d_rows);
}
inline float gemm_small_elem(const float* a, size_t a_stride, const float* b, size_t b_stride, int len, double alpha, float c, double beta) {
    float t = a[0] * b[0] + a[a_stride] * b[b_stride];
    if (len > 2) t = t + a[2 * a_stride] * b[2 * b_stride];
    if (len > 3) t = t + a[3 * a_stride] * b[3 * b_stride];
    return (float)(t * alpha + c * beta);
}
inline void gemm_small_f32(const Mat& A, const Mat& B, double alpha, const Mat* C, double beta, Mat& D) {
    const int len = A.cols;
    const size_t bs = B.step / 4;
    for (int i = 0; i < D.rows; ++i)
        for (int j = 0; j < D.cols; ++j)
            D.at<float>(i, j) = gemm_small_elem(A.ptr<float>(i), 1, B.ptr<float>(0) + j, bs, len, alpha, C ? C->at<float>(i, j) : 0.0f, beta);
}
inline void gemm_general_f32(const Mat& A, const Mat& B, double alpha, const Mat* C, double beta, Mat& D, int flags) {
    const bool at = (flags & GEMM_1_T) != 0, bt = (flags & GEMM_2_T) != 0, ct = (flags & GEMM_3_T) != 0;
    const int n = at ? A.rows : A.cols;
    auto a_at = [&](int i, int k) { return (double)(at ? A.at<float>(k, i) : A.at<float>(i, k)); };
    auto b_at = [&](int k, int j) { return (double)(bt ? B.at<float>(j, k) : B.at<float>(k, j)); };
    auto c_at = [&](int i, int j) { return (double)(ct ? C->at<float>(j, i) : C->at<float>(i, j)); };
    for (int i = 0; i < D.rows; ++i)
        for (int j = 0; j < D.cols; ++j) {
            double s;
            if (n == 1) {                       // "external product" branch: (a*alpha)*b
                s = (a_at(i, 0) * alpha) * b_at(0, j);
            } else if (bt) {                    // A*Bt branch: four partial sums over k (CV_ENABLE_UNROLLED), then (s0+s1+s2+s3)*alpha
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                int k = 0;
                for (; k <= n - 4; k += 4) {
                    s0 += a_at(i, k) * b_at(k, j); s1 += a_at(i, k + 1) * b_at(k + 1, j);
                    s2 += a_at(i, k + 2) * b_at(k + 2, j); s3 += a_at(i, k + 3) * b_at(k + 3, j);
                }
                for (; k < n; ++k) s0 += a_at(i, k) * b_at(k, j);
                s = (s0 + s1 + s2 + s3) * alpha;
            } else {                            // row-times-columns branch: one running sum, then s*alpha
                double s0 = 0;
                for (int k = 0; k < n; ++k) s0 += a_at(i, k) * b_at(k, j);
                s = s0 * alpha;
            }
            D.at<float>(i, j) = C ? (float)(s + c_at(i, j) * beta) : (float)s;
        }
}
inline Mat gemm(const Mat& A, const Mat& B, double alpha, const Mat& Cin, double beta, int flags = 0) {
    assert(A.type() == CV_32F && B.type() == CV_32F);
    const int d_rows = (flags & GEMM_1_T) ? A.cols : A.rows, d_cols = (flags & GEMM_2_T) ? B.rows : B.cols;
    const int len = (flags & GEMM_2_T) ? B.cols : B.rows;
    assert(((flags & GEMM_1_T) ? A.rows : A.cols) == len && d_cols <= 400);
    const Mat* C = (beta != 0 && !Cin.empty()) ? &Cin : nullptr;
    Mat D(d_rows, d_cols, CV_32F);      // always a fresh block: an aliased destination (x = R*x + t) reads its operands first, as cv::gemm does
    if (gemm_takes_small_path(len, d_rows, d_cols, flags)) gemm_small_f32(A, B, alpha, C, beta, D);
    else gemm_general_f32(A, B, alpha, C, beta, D, flags);
    return D;
}

// ---- cv::solve(A, B, DECOMP_LU) for CV_32F with more than one right-hand column: Gaussian elimination with partial pivoting in
// float on copies of A and B (modules/core/src/lapack.cpp: LUImpl, eps = FLT_EPSILON*10 as hal::LU32f of 3.x passes; 2.4.x compares
// the pivot with FLT_EPSILON -- the two differ only for matrices that are singular to working precision).  A singular system gives zeros.
// The one-column closed forms of cv::solve (n <= 3, b.cols == 1) are not restated: nothing on the path uses them.
inline Mat solve_lu(const Mat& Ain, const Mat& Bin) {
    assert(Ain.type() == CV_32F && Ain.rows == Ain.cols && Bin.rows == Ain.rows && !(Ain.rows <= 3 && Bin.cols == 1));
    Mat Am = Ain.clone(), Bm = Bin.clone();
    const int m = Am.rows, n = Bm.cols;
    float* A = Am.ptr<float>(0); float* b = Bm.ptr<float>(0);
    const size_t astep = Am.step / 4, bstep = Bm.step / 4;
    const float eps = FLT_EPSILON * 10;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (std::abs(A[j * astep + i]) > std::abs(A[k * astep + i])) k = j;
        if (std::abs(A[k * astep + i]) < eps) return Mat::zeros(m, n, CV_32F);
        if (k != i) {
            for (int j = i; j < m; j++) std::swap(A[i * astep + j], A[k * astep + j]);
            for (int j = 0; j < n; j++) std::swap(b[i * bstep + j], b[k * bstep + j]);
        }
        const float d = -1 / A[i * astep + i];
        for (int j = i + 1; j < m; j++) {
            const float alpha = A[j * astep + i] * d;
            for (k = i + 1; k < m; k++) A[j * astep + k] += alpha * A[i * astep + k];
            for (k = 0; k < n; k++) b[j * bstep + k] += alpha * b[i * bstep + k];
        }
        A[i * astep + i] = -d;
    }
    for (int i = m - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            float s = b[i * bstep + j];
            for (int k = i + 1; k < m; k++) s -= A[i * astep + k] * b[k * bstep + j];
            b[i * bstep + j] = s * A[i * astep + i];
        }
    return Bm;
}

// element-wise float ops as cv::add / cv::subtract / Mat::convertTo(scale) / cv::addWeighted evaluate them for CV_32F
inline Mat ew_add(const Mat& a, const Mat& b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j);
    return m;
}
inline Mat ew_sub(const Mat& a, const Mat& b) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
    return m;
}
// alpha*M: +1 is cv::add(M, Scalar(0)), -1 is cv::subtract(Scalar(0), M), anything else convertTo(alpha), whose float -> float
// kernel computes src*scale + shift in float with scale = (float)alpha, shift = 0.0f (modules/core/src/convert.cpp, cvtScale_)
inline Mat ew_scale(const Mat& a, double alpha) {
    Mat m(a.rows, a.cols, CV_32F);
    const float al = (float)alpha;
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < a.cols; ++j) {
            const float x = a.at<float>(i, j);
            m.at<float>(i, j) = alpha == 1 ? x + 0.0f : alpha == -1 ? 0.0f - x : x * al + 0.0f;
        }
    return m;
}
// alpha*A + beta*B for weights other than +-1: cv::addWeighted, float weights, `a*alpha + b*beta + 0`
inline Mat ew_weighted(const Mat& a, double alpha, const Mat& b, double beta) {
    Mat m(a.rows, a.cols, CV_32F);
    const float al = (float)alpha, be = (float)beta;
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) m.at<float>(i, j) = a.at<float>(i, j) * al + b.at<float>(i, j) * be + 0.0f;
    return m;
}

// ---- cv::MatExpr, as far as the pose algebra of ORBmatcher.cc goes.  OpenCV does not evaluate `R*x + t` as a product followed
// by a sum: operator* builds a lazy expression and operator+ folds the addend into the SAME gemm call
// (modules/core/src/matop.cpp: MatOp_GEMM::add / ::subtract, MatOp::matmul, MatOp_T::multiply, MatOp_Invert::matmul,
// MatOp_AddEx::assign, operator-(const MatExpr&)).  The rules restated here:
//   A*B                 -> GEMM(A, B, 1)                      A.t()*B -> GEMM(A, B, GEMM_1_T)       A*B.t() -> GEMM(.., GEMM_2_T)
//   (s*A)*B, (-A)*B     -> GEMM(A, B, alpha = s)              (A*B)*C -> GEMM(eval(A*B), C)
//   A*B + C, C + A*B    -> the same GEMM with C, beta = 1     C - A*B -> GEMM(A, B, -alpha, C, 1)   A*B - C -> beta = -1
//   -(A.t())            -> evaluates the transpose, then alpha = -1 on the evaluated matrix (MatOp::subtract(Scalar, expr))
//   A.inv()*B           -> cv::solve(A, B, DECOMP_LU); every other use of A.inv() evaluates cv::invert(A)
//   s*A, A/s            -> scaled (ew_scale when evaluated on its own)      s*A.t() -> transposed with alpha = s
class MatExpr {
public:
    enum Kind { IDENT, SCALED, TRANSP, GEMM, INVERT, SOLVE, ADDEX };
    Kind kind = IDENT;
    Mat a, b, c;
    double alpha = 1, beta = 0;
    int flags = 0;
    MatExpr() {}
    explicit MatExpr(const Mat& m) : a(m) {}
    static MatExpr make(Kind k, const Mat& a, const Mat& b = Mat(), double alpha = 1, const Mat& c = Mat(), double beta = 0, int flags = 0) {
        MatExpr e; e.kind = k; e.a = a; e.b = b; e.c = c; e.alpha = alpha; e.beta = beta; e.flags = flags; return e;
    }
    Mat eval() const {
        switch (kind) {
        case IDENT: return a;
        case SCALED: return ew_scale(a, alpha);
        case TRANSP: { Mat m = a.transposed(); return alpha != 1 ? ew_scale_cvt(m, alpha) : m; }
        case GEMM: return gemm(a, b, alpha, c, beta, flags);
        case INVERT: return a.inverted();
        case SOLVE: return solve_lu(a, b);
        case ADDEX:
            if (alpha == 1 && beta == 1) return ew_add(a, b);
            if (alpha == 1 && beta == -1) return ew_sub(a, b);
            return ew_weighted(a, alpha, b, beta);
        }
        return Mat();
    }
    operator Mat() const { return eval(); }
    // the members the reference's statements use on expressions
    MatExpr t() const {   // MatOp_T::transpose undoes a transposition; anything else is evaluated first
        if (kind == TRANSP) return alpha == 1 ? MatExpr(a) : make(SCALED, a, Mat(), alpha);
        return make(TRANSP, eval());
    }
    MatExpr inv() const { return make(INVERT, eval()); }
    bool is_matprod() const { return kind == GEMM && (c.empty() || beta == 0); }
private:
    static Mat ew_scale_cvt(const Mat& m, double alpha) {   // convertTo(m, type, alpha): always the scale kernel
        Mat r(m.rows, m.cols, CV_32F);
        const float al = (float)alpha;
        for (int i = 0; i < m.rows; ++i) for (int j = 0; j < m.cols; ++j) r.at<float>(i, j) = m.at<float>(i, j) * al + 0.0f;
        return r;
    }
};
inline MatExpr Mat::inv() const { assert(type_ == CV_32F); return MatExpr::make(MatExpr::INVERT, *this); }
inline MatExpr Mat::t() const { assert(type_ == CV_32F); return MatExpr::make(MatExpr::TRANSP, *this); }

// MatOp::matmul with the special case of MatOp_Invert::matmul in front
inline MatExpr expr_matmul(const MatExpr& e1, const MatExpr& e2) {
    if (e1.kind == MatExpr::INVERT && e2.kind == MatExpr::IDENT) return MatExpr::make(MatExpr::SOLVE, e1.a, e2.a);
    double scale = 1; int flags = 0; Mat m1, m2;
    if (e1.kind == MatExpr::TRANSP) { flags |= GEMM_1_T; scale *= e1.alpha; m1 = e1.a; }
    else if (e1.kind == MatExpr::SCALED) { scale *= e1.alpha; m1 = e1.a; }
    else m1 = e1.eval();
    if (e2.kind == MatExpr::TRANSP) { flags |= GEMM_2_T; scale *= e2.alpha; m2 = e2.a; }
    else if (e2.kind == MatExpr::SCALED) { scale *= e2.alpha; m2 = e2.a; }
    else m2 = e2.eval();
    return MatExpr::make(MatExpr::GEMM, m1, m2, scale, Mat(), 0, flags);
}
// MatOp_GEMM::add / ::subtract (sign = +1 / -1 on e2) with MatOp::add / ::subtract behind them
inline MatExpr expr_addsub(const MatExpr& e1, const MatExpr& e2, double sign) {
    auto foldable = [](const MatExpr& e) { return e.kind == MatExpr::IDENT || e.kind == MatExpr::SCALED || e.kind == MatExpr::TRANSP; };
    auto weight = [](const MatExpr& e) { return e.kind == MatExpr::IDENT ? 1.0 : e.alpha; };
    if (e1.is_matprod() && foldable(e2))
        return MatExpr::make(MatExpr::GEMM, e1.a, e1.b, e1.alpha, e2.a, sign * weight(e2), (e1.flags & ~GEMM_3_T) | (e2.kind == MatExpr::TRANSP ? GEMM_3_T : 0));
    if (e2.is_matprod() && foldable(e1))
        return MatExpr::make(MatExpr::GEMM, e2.a, e2.b, sign * e2.alpha, e1.a, weight(e1), (e2.flags & ~GEMM_3_T) | (e1.kind == MatExpr::TRANSP ? GEMM_3_T : 0));
    double al = 1, be = 1; Mat m1, m2;
    if (e1.kind == MatExpr::SCALED) { m1 = e1.a; al = e1.alpha; } else m1 = e1.eval();
    if (e2.kind == MatExpr::SCALED) { m2 = e2.a; be = e2.alpha; } else m2 = e2.eval();
    return MatExpr::make(MatExpr::ADDEX, m1, m2, al, Mat(), sign * be);
}
// operator-(const MatExpr&): MatOp_AddEx::subtract(Scalar, e) flips the weight of a scaled matrix; every other kind is
// evaluated and negated as alpha = -1 (MatOp::subtract(const Scalar&, const MatExpr&, MatExpr&))
inline MatExpr expr_neg(const MatExpr& e) {
    if (e.kind == MatExpr::SCALED) return MatExpr::make(MatExpr::SCALED, e.a, Mat(), -e.alpha);
    return MatExpr::make(MatExpr::SCALED, e.eval(), Mat(), -1);
}
// s * expr: MatOp_T::multiply / MatOp_AddEx::multiply / MatOp_GEMM::multiply scale the weights; the rest is evaluated first
inline MatExpr expr_scale(const MatExpr& e, double s) {
    if (e.kind == MatExpr::SCALED || e.kind == MatExpr::TRANSP) { MatExpr r = e; r.alpha *= s; return r; }
    if (e.kind == MatExpr::GEMM) { MatExpr r = e; r.alpha *= s; r.beta *= s; return r; }
    return MatExpr::make(MatExpr::SCALED, e.eval(), Mat(), s);
}

inline MatExpr operator*(const Mat& a, const Mat& b) { return expr_matmul(MatExpr(a), MatExpr(b)); }
inline MatExpr operator*(const MatExpr& a, const Mat& b) { return expr_matmul(a, MatExpr(b)); }
inline MatExpr operator*(const Mat& a, const MatExpr& b) { return expr_matmul(MatExpr(a), b); }
inline MatExpr operator*(const MatExpr& a, const MatExpr& b) { return expr_matmul(a, b); }
inline MatExpr operator+(const Mat& a, const Mat& b) { return expr_addsub(MatExpr(a), MatExpr(b), 1); }
inline MatExpr operator+(const MatExpr& a, const Mat& b) { return expr_addsub(a, MatExpr(b), 1); }
inline MatExpr operator+(const Mat& a, const MatExpr& b) { return expr_addsub(MatExpr(a), b, 1); }
inline MatExpr operator+(const MatExpr& a, const MatExpr& b) { return expr_addsub(a, b, 1); }
inline MatExpr operator-(const Mat& a, const Mat& b) { return expr_addsub(MatExpr(a), MatExpr(b), -1); }
inline MatExpr operator-(const MatExpr& a, const Mat& b) { return expr_addsub(a, MatExpr(b), -1); }
inline MatExpr operator-(const Mat& a, const MatExpr& b) { return expr_addsub(MatExpr(a), b, -1); }
inline MatExpr operator-(const MatExpr& a, const MatExpr& b) { return expr_addsub(a, b, -1); }
inline MatExpr operator-(const Mat& a) { return MatExpr::make(MatExpr::SCALED, a, Mat(), -1); }
inline MatExpr operator-(const MatExpr& a) { return expr_neg(a); }
inline MatExpr operator*(double s, const Mat& a) { return MatExpr::make(MatExpr::SCALED, a, Mat(), s); }
inline MatExpr operator*(const Mat& a, double s) { return MatExpr::make(MatExpr::SCALED, a, Mat(), s); }
inline MatExpr operator*(double s, const MatExpr& a) { return expr_scale(a, s); }
inline MatExpr operator*(const MatExpr& a, double s) { return expr_scale(a, s); }
inline MatExpr operator/(const Mat& a, double s) { return MatExpr::make(MatExpr::SCALED, a, Mat(), 1. / s); }
inline MatExpr operator/(const MatExpr& a, double s) { return expr_scale(a, 1. / s); }
// L2 norm: squares summed in double, square root in double (cv::norm, NORM_L2, CV_32F: normL2_<float, double>)
inline double norm(const Mat& a) {
    double s = 0;
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) s += (double)a.at<float>(i, j) * (double)a.at<float>(i, j);
    return std::sqrt(s);
}

// cv::FileStorage / cv::FileNode: DECLARATIONS only, so that Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h (whose YAML save / load
// members mention them) parses in a build without OpenCV.  Nothing in this repository defines or calls them -- ORBVocabulary
// loads the text format (loadFromTextFile) -- and a program that does call them fails at link time, not silently.
class FileNode {
public:
    FileNode operator[](const std::string& name) const;
    FileNode operator[](const char* name) const;
    FileNode operator[](int i) const;
    size_t size() const;
    bool empty() const;
    operator int() const;
    operator float() const;
    operator double() const;
    operator std::string() const;
};
class FileStorage {
public:
    enum { READ = 0, WRITE = 1, APPEND = 2 };
    FileStorage();
    FileStorage(const std::string& filename, int flags);
    ~FileStorage();
    bool isOpened() const;
    void release();
    FileNode operator[](const std::string& name) const;
    FileNode operator[](const char* name) const;
};
template <typename T> FileStorage& operator<<(FileStorage& fs, const T& value);
FileStorage& operator<<(FileStorage& fs, const char* value);

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
