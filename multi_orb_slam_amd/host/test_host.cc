// test_host.cc -- C++ driver for the reference-signature host classes (ORB_SLAM2::ORBextractor / ORBmatcher).
// tests/test_gpu_host_cpp.py writes the inputs, runs this program on the GPU box and compares its outputs with the
// oracle.  Binary I/O only: little-endian int32 / float32 / uint8 arrays.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;

static std::vector<unsigned char> slurp(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b(n);
    if (n && std::fread(b.data(), 1, n, f) != (size_t)n) { std::perror("read"); std::exit(2); }
    std::fclose(f);
    return b;
}

struct Reader {
    const unsigned char* p;
    template <typename T> T get() { T v; std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    template <typename T> std::vector<T> arr(size_t n) { std::vector<T> v(n); std::memcpy(v.data(), p, n * sizeof(T)); p += n * sizeof(T); return v; }
};

static void put(FILE* f, const void* p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) { std::perror("write"); std::exit(2); } }

// extract <image.bin> <w> <h> <nfeatures> <out.bin> [w2 h2 image2.bin nfeatures2]
static int run_extract(int argc, char** argv) {
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), nf = std::atoi(argv[5]);
    std::vector<unsigned char> img = slurp(argv[2]);
    cv::Mat im(h, w, CV_8UC1, img.data());
    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps; cv::Mat desc;
    ex(im, cv::Mat(), kps, desc);
    // twice on the same object: the class is reusable and deterministic
    std::vector<cv::KeyPoint> kps2; cv::Mat desc2;
    ex(im, cv::Mat(), kps2, desc2);
    if (kps.size() != kps2.size() || std::memcmp(kps.data(), kps2.data(), kps.size() * sizeof(cv::KeyPoint)) != 0) return 3;
    // empty image: outputs untouched
    std::vector<cv::KeyPoint> untouched(7); cv::Mat udesc(7, 32, CV_8U);
    ex(cv::Mat(), cv::Mat(), untouched, udesc);
    if (untouched.size() != 7 || udesc.rows != 7) return 4;
    // getters (reference include/ORBextractor.h:64-84)
    if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8 || ex.GetInverseScaleSigmaSquares().size() != 8) return 5;
    FILE* f = std::fopen(argv[6], "wb");
    int n = (int)kps.size();
    put(f, &n, 4); put(f, kps.data(), (size_t)n * sizeof(cv::KeyPoint));
    if (n) put(f, desc.ptr(0), (size_t)n * 32);
    std::vector<float> sf = ex.GetScaleFactors();
    put(f, sf.data(), 8 * 4);
    std::fclose(f);
    return 0;
}

// batch <out.bin> <w> <h> <image0.bin> <nf0> <image1.bin> <nf1>
static int run_batch(int argc, char** argv) {
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    std::vector<unsigned char> i0 = slurp(argv[5]), i1 = slurp(argv[7]);
    ORBextractor e0(std::atoi(argv[6]), 1.2f, 8, 20, 7), e1(std::atoi(argv[8]), 1.2f, 8, 20, 7);
    std::vector<cv::Mat> images = {cv::Mat(h, w, CV_8UC1, i0.data()), cv::Mat(h, w, CV_8UC1, i1.data())};
    std::vector<std::vector<cv::KeyPoint> > kps; std::vector<cv::Mat> desc;
    ORBextractor::ExtractBatch({&e0, &e1}, images, kps, desc);
    FILE* f = std::fopen(argv[2], "wb");
    for (int c = 0; c < 2; ++c) {
        int n = (int)kps[c].size();
        put(f, &n, 4); put(f, kps[c].data(), (size_t)n * sizeof(cv::KeyPoint));
        if (n) put(f, desc[c].ptr(0), (size_t)n * 32);
    }
    std::fclose(f);
    return 0;
}

static void read_frame(Reader& R, Frame& F) {
    F.N = R.get<int>(); F.N_cam2 = R.get<int>(); F.N_total = F.N + F.N_cam2;
    const int n = F.N_total;
    std::vector<float> x = R.arr<float>(n), y = R.arr<float>(n), ang = R.arr<float>(n), ur = R.arr<float>(n);
    std::vector<int> oct = R.arr<int>(n);
    F.mvKeys_total.resize(n); F.mvKeysUn_total.resize(n); F.mvuRight_total = ur;
    for (int g = 0; g < n; ++g) {
        cv::KeyPoint k; k.pt.x = x[g]; k.pt.y = y[g]; k.angle = ang[g]; k.octave = oct[g];
        F.mvKeys_total[g] = k; F.mvKeysUn_total[g] = k;
        F.keypoint_to_cam[g] = g < F.N ? 0 : 1;
        F.cont_idx_to_local_cam_idx[g] = g < F.N ? g : g - F.N;
    }
    F.mvKeysUn.assign(F.mvKeysUn_total.begin(), F.mvKeysUn_total.begin() + F.N);
    F.mvuRight.assign(ur.begin(), ur.begin() + F.N);
    F.mDescriptors_total.resize(2);
    for (int c = 0; c < 2; ++c) {
        const int nc = c == 0 ? F.N : F.N_cam2;
        F.mDescriptors_total[c].create(nc > 0 ? nc : 1, 32, CV_8U);
        std::vector<unsigned char> d = R.arr<unsigned char>((size_t)nc * 32);
        if (nc) std::memcpy(F.mDescriptors_total[c].ptr(0), d.data(), d.size());
    }
    F.mDescriptors = F.mDescriptors_total[0];
    F.mvpMapPoints.assign(n, nullptr);
    F.mvbOutlier.assign(n, false);
    F.mvScaleFactors = R.arr<float>(8);
    F.mTcw = cv::Mat::eye(4, 4, CV_32F);
    std::vector<float> T = R.arr<float>(16);
    for (int i = 0; i < 16; ++i) F.mTcw.at<float>(i / 4, i % 4) = T[i];
    F.fx = R.get<float>(); F.fy = R.get<float>(); F.cx = R.get<float>(); F.cy = R.get<float>();
    F.mbf = R.get<float>(); F.mb = F.mbf / F.fx;
    F.mnMinX = R.get<float>(); F.mnMinY = R.get<float>(); F.mnMaxX = R.get<float>(); F.mnMaxY = R.get<float>();
}

// match <case.bin> <out.bin>: [current frame][last frame][nmp x {world xyz, desc32, obs, has, outlier}][calib 4x3]
//                             [local points: count x {projX, projY, projXR, level, viewCos, desc32, inview, bad}]
static int do_match(const std::vector<unsigned char>& buf, FILE* f) {
    Reader R{buf.data()};
    Frame Cur, Last;
    read_frame(R, Cur); read_frame(R, Last);
    std::vector<MapPoint> pool(Last.N_total);
    for (int i = 0; i < Last.N_total; ++i) {
        MapPoint& mp = pool[i];
        std::vector<float> xyz = R.arr<float>(3);
        mp.mWorldPos = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; ++k) mp.mWorldPos.at<float>(k) = xyz[k];
        mp.mDescriptor = cv::Mat(1, 32, CV_8U);
        std::vector<unsigned char> d = R.arr<unsigned char>(32);
        std::memcpy(mp.mDescriptor.ptr(0), d.data(), 32);
        mp.nObs = R.get<int>();
        const int has = R.get<int>(), outlier = R.get<int>();
        Last.mvpMapPoints[i] = has ? &mp : nullptr;
        Last.mvbOutlier[i] = outlier != 0;
    }
    cv::Mat calib(4, 3, CV_32F);
    std::vector<float> cm = R.arr<float>(12);
    for (int i = 0; i < 12; ++i) calib.at<float>(i / 3, i % 3) = cm[i];
    const float th = R.get<float>();
    const int check_ori = R.get<int>();

    ORBmatcher matcher(0.9f, check_ori != 0);
    const int n1 = matcher.SearchByProjection(Cur, Last, th, false, calib);
    put(f, &n1, 4);
    for (int g = 0; g < Cur.N_total; ++g) {
        int idx = Cur.mvpMapPoints[g] ? (int)(Cur.mvpMapPoints[g] - pool.data()) : -1;
        put(f, &idx, 4);
    }
    // second overload on a fresh current frame: SearchByProjection(F, vpMapPoints, th)
    const int nlocal = R.get<int>();
    std::vector<MapPoint> local(nlocal);
    std::vector<MapPoint*> vp(nlocal);
    for (int i = 0; i < nlocal; ++i) {
        MapPoint& mp = local[i];
        mp.mTrackProjX = R.get<float>(); mp.mTrackProjY = R.get<float>(); mp.mTrackProjXR = R.get<float>();
        mp.mnTrackScaleLevel = R.get<int>(); mp.mTrackViewCos = R.get<float>();
        mp.mDescriptor = cv::Mat(1, 32, CV_8U);
        std::vector<unsigned char> d = R.arr<unsigned char>(32);
        std::memcpy(mp.mDescriptor.ptr(0), d.data(), 32);
        mp.mbTrackInView = R.get<int>() != 0; mp.mbBad = R.get<int>() != 0; mp.nObs = 2;
        vp[i] = &mp;
    }
    const float th2 = R.get<float>();
    Cur.mvpMapPoints.assign(Cur.N_total, nullptr);
    ORBmatcher m2(0.8f);
    const int n2 = m2.SearchByProjection(Cur, vp, th2);
    put(f, &n2, 4);
    for (int g = 0; g < Cur.N; ++g) {
        int idx = Cur.mvpMapPoints[g] ? (int)(Cur.mvpMapPoints[g] - local.data()) : -1;
        put(f, &idx, 4);
    }
    // static DescriptorDistance
    const int dd = ORBmatcher::DescriptorDistance(Cur.mDescriptors.row(0), Cur.mDescriptors.row(1));
    put(f, &dd, 4);
    return 0;
}
static int run_file(int (*fn)(const std::vector<unsigned char>&, FILE*), char** argv) {
    std::vector<unsigned char> buf = slurp(argv[2]);
    FILE* f = std::fopen(argv[3], "wb");
    if (!f) { std::perror(argv[3]); return 2; }
    const int rc = fn(buf, f);
    std::fclose(f);
    return rc;
}

// One keyframe / frame of a bow case: N, N_cam2, x y angle uright (f32 each), octave (i32), descriptors cam 0 then cam 1,
// has-MapPoint and bad flags (u8 each), Tcw and Tcw_cam2 (16 f32 each).
struct BowEntity {
    int N = 0, N2 = 0, n = 0;
    std::vector<cv::KeyPoint> keys;
    std::vector<float> ur;
    std::vector<cv::Mat> desc;
    std::vector<unsigned char> has, bad;
    cv::Mat Tcw, Tcw2;
    std::vector<MapPoint> pool;
    std::vector<MapPoint*> mps;
    std::unordered_map<size_t, int> to_cam, to_local;
};

static void read_entity(Reader& R, BowEntity& E) {
    E.N = R.get<int>(); E.N2 = R.get<int>(); E.n = E.N + E.N2;
    const int n = E.n;
    std::vector<float> x = R.arr<float>(n), y = R.arr<float>(n), ang = R.arr<float>(n);
    E.ur = R.arr<float>(n);
    std::vector<int> oct = R.arr<int>(n);
    E.keys.resize(n);
    for (int g = 0; g < n; ++g) {
        E.keys[g].pt.x = x[g]; E.keys[g].pt.y = y[g]; E.keys[g].angle = ang[g]; E.keys[g].octave = oct[g];
        E.to_cam[g] = g < E.N ? 0 : 1; E.to_local[g] = g < E.N ? g : g - E.N;
    }
    E.desc.resize(2);
    for (int c = 0; c < 2; ++c) {
        const int nc = c == 0 ? E.N : E.N2;
        E.desc[c].create(nc > 0 ? nc : 1, 32, CV_8U);
        std::vector<unsigned char> d = R.arr<unsigned char>((size_t)nc * 32);
        if (nc) std::memcpy(E.desc[c].ptr(0), d.data(), d.size());
    }
    E.has = R.arr<unsigned char>(n); E.bad = R.arr<unsigned char>(n);
    E.pool.resize(n); E.mps.assign(n, nullptr);
    for (int g = 0; g < n; ++g) { E.pool[g].mbBad = E.bad[g] != 0; if (E.has[g]) E.mps[g] = &E.pool[g]; }
    E.Tcw = cv::Mat(4, 4, CV_32F); E.Tcw2 = cv::Mat(4, 4, CV_32F);
    std::vector<float> T = R.arr<float>(16), T2 = R.arr<float>(16);
    for (int i = 0; i < 16; ++i) { E.Tcw.at<float>(i / 4, i % 4) = T[i]; E.Tcw2.at<float>(i / 4, i % 4) = T2[i]; }
}

static std::vector<cv::Mat> rows_of(const BowEntity& E) {  // Converter::toDescriptorVector(mDescriptors_total)
    std::vector<cv::Mat> v;
    for (int g = 0; g < E.n; ++g) v.push_back(E.desc[g < E.N ? 0 : 1].row(g < E.N ? g : g - E.N));
    return v;
}

// bow <case.bin> <out.bin>: [vocabulary arrays][levelsup][KF1][KF2][F][fx fy cx cy][scale 8][sigma2 8][nnratio][check_ori]
//                           [bOnlyStereo][vbCam0][vbCam1]
static int do_bow(const std::vector<unsigned char>& buf, FILE* f) {
    Reader R{buf.data()};
    const int n_nodes = R.get<int>(), L = R.get<int>();
    std::vector<int> parent = R.arr<int>(n_nodes);
    std::vector<unsigned char> leaf = R.arr<unsigned char>(n_nodes), vdesc = R.arr<unsigned char>((size_t)n_nodes * 32);
    std::vector<double> weight = R.arr<double>(n_nodes);
    const int levelsup = R.get<int>();
    ORBVocabulary voc;
    if (!voc.empty()) return 3;
    if (!voc.create(n_nodes, L, parent.data(), leaf.data(), vdesc.data(), weight.data())) return 4;
    BowEntity E1, E2, EF;
    read_entity(R, E1); read_entity(R, E2); read_entity(R, EF);
    const float fx = R.get<float>(), fy = R.get<float>(), cx = R.get<float>(), cy = R.get<float>();
    std::vector<float> scale = R.arr<float>(8), sigma2 = R.arr<float>(8);
    const float nnratio = R.get<float>();
    const int check_ori = R.get<int>(), only_stereo = R.get<int>(), cam0 = R.get<int>(), cam1 = R.get<int>();

    KeyFrame K1, K2; Frame F;
    BowEntity* Es[2] = {&E1, &E2}; KeyFrame* Ks[2] = {&K1, &K2};
    for (int k = 0; k < 2; ++k) {
        KeyFrame& K = *Ks[k]; BowEntity& E = *Es[k];
        K.mvKeysUn_total = E.keys; K.mvuRight_total = E.ur; K.mDescriptors_total = E.desc; K.keypoint_to_cam = E.to_cam;
        K.cont_idx_to_local_cam_idx = E.to_local; K.mvpMapPoints = E.mps; K.mvScaleFactors = scale; K.mvLevelSigma2 = sigma2;
        K.mK = cv::Mat::eye(3, 3, CV_32F);
        K.mK.at<float>(0, 0) = fx; K.mK.at<float>(1, 1) = fy; K.mK.at<float>(0, 2) = cx; K.mK.at<float>(1, 2) = cy;
        K.fx = fx; K.fy = fy; K.cx = cx; K.cy = cy; K.Tcw = E.Tcw; K.Tcw_cam2 = E.Tcw2;
        voc.transform(rows_of(E), K.mBowVec, K.mFeatVec, levelsup);     // KeyFrame::ComputeBoW
    }
    F.N = EF.N; F.N_cam2 = EF.N2; F.N_total = EF.n; F.mvKeys_total = EF.keys; F.mvKeysUn_total = EF.keys; F.mDescriptors_total = EF.desc;
    F.keypoint_to_cam = EF.to_cam; F.cont_idx_to_local_cam_idx = EF.to_local;
    voc.transform(rows_of(EF), F.mBowVec, F.mFeatVec, levelsup);         // Frame::ComputeBoW, src/Frame.cc:649-659

    int nw = (int)K1.mBowVec.size(); put(f, &nw, 4);
    for (const auto& e : K1.mBowVec) { put(f, &e.first, 4); put(f, &e.second, 8); }
    int nn = (int)K1.mFeatVec.size(); put(f, &nn, 4);
    for (const auto& e : K1.mFeatVec) { int c = (int)e.second.size(); put(f, &e.first, 4); put(f, &c, 4); put(f, e.second.data(), (size_t)c * 4); }
    const double s12 = voc.score(K1.mBowVec, K2.mBowVec), s11 = voc.score(K1.mBowVec, K1.mBowVec);
    put(f, &s12, 8); put(f, &s11, 8);
    unsigned int words = voc.size(); put(f, &words, 4);

    ORBmatcher m1(nnratio, check_ori != 0);
    std::vector<MapPoint*> vF;
    const int na = m1.SearchByBoW(&K1, F, vF);
    put(f, &na, 4);
    for (int g = 0; g < F.N_total; ++g) { int idx = vF[g] ? (int)(vF[g] - E1.pool.data()) : -1; put(f, &idx, 4); }
    std::vector<MapPoint*> v12;
    const int nb = m1.SearchByBoW(&K1, &K2, v12);
    put(f, &nb, 4);
    for (int g = 0; g < E1.n; ++g) { int idx = v12[g] ? (int)(v12[g] - E2.pool.data()) : -1; put(f, &idx, 4); }
    std::vector<std::pair<size_t, size_t> > pairs;
    std::vector<bool> vbCam = {cam0 != 0, cam1 != 0};
    const int nc = m1.SearchForTriangulation(&K1, &K2, cv::Mat(), pairs, only_stereo != 0, vbCam);
    int np = (int)pairs.size();
    put(f, &nc, 4); put(f, &np, 4);
    for (const auto& pr : pairs) { int a = (int)pr.first, b = (int)pr.second; put(f, &a, 4); put(f, &b, 4); }
    // the camera-1 forms the reference's threads call: camera-1 vocabulary vectors, camera-1 descriptors / keypoints
    auto cam1_rows = [](const BowEntity& E) { std::vector<cv::Mat> v; for (int g = 0; g < E.N; ++g) v.push_back(E.desc[0].row(g)); return v; };
    for (int k = 0; k < 2; ++k) {
        KeyFrame& K = *Ks[k]; BowEntity& E = *Es[k];
        K.N = E.N; K.N_cam2 = E.N2; K.N_total = E.n; K.mDescriptors = E.desc[0];
        K.mvKeysUn.assign(E.keys.begin(), E.keys.begin() + E.N);
        voc.transform(cam1_rows(E), K.mBowVec_cam1, K.mFeatVec_cam1, levelsup);
    }
    F.mDescriptors = EF.desc[0]; F.mvKeys.assign(EF.keys.begin(), EF.keys.begin() + EF.N);
    voc.transform(cam1_rows(EF), F.mBowVec_cam1, F.mFeatVec_cam1, levelsup);
    std::vector<MapPoint*> vF1, v121;
    const int nd = m1.SearchByBoW_cam1(&K1, F, vF1);
    put(f, &nd, 4);
    for (int g = 0; g < F.N; ++g) { int idx = vF1[g] ? (int)(vF1[g] - E1.pool.data()) : -1; put(f, &idx, 4); }
    const int ne = m1.SearchByBoW_cam1(&K1, &K2, v121);
    put(f, &ne, 4);
    for (int g = 0; g < E1.N; ++g) { int idx = v121[g] ? (int)(v121[g] - E2.pool.data()) : -1; put(f, &idx, 4); }
    return 0;
}

static void to_keyframe(const Frame& F, KeyFrame& K) {
    K.N = F.N; K.N_cam2 = F.N_cam2; K.N_total = F.N_total;
    K.mvKeysUn_total = F.mvKeysUn_total; K.mvKeysUn = F.mvKeysUn; K.mvuRight_total = F.mvuRight_total; K.mvuRight = F.mvuRight;
    K.mDescriptors_total = F.mDescriptors_total; K.mDescriptors = F.mDescriptors;
    K.keypoint_to_cam = F.keypoint_to_cam; K.cont_idx_to_local_cam_idx = F.cont_idx_to_local_cam_idx;
    K.mvScaleFactors = F.mvScaleFactors;
    K.mvLevelSigma2.clear(); K.mvInvLevelSigma2.clear();
    for (float s : F.mvScaleFactors) { K.mvLevelSigma2.push_back(s * s); K.mvInvLevelSigma2.push_back(1.0f / (s * s)); }
    K.fx = F.fx; K.fy = F.fy; K.cx = F.cx; K.cy = F.cy; K.mbf = F.mbf;
    K.mnMinX = F.mnMinX; K.mnMinY = F.mnMinY; K.mnMaxX = F.mnMaxX; K.mnMaxY = F.mnMaxY;
    K.Tcw = F.mTcw.clone(); K.Tcw_cam2 = F.mTcw.clone();
    K.mfLogScaleFactor = F.mfLogScaleFactor; K.mnScaleLevels = F.mnScaleLevels;
    K.mvpMapPoints.assign(F.N_total, nullptr);
}

// f4 <case.bin> <out.bin>: the remaining projection searches through the reference signatures.
// [Cur][KA][KB][Tcw_cam2 of KB 16f][M points: xyz 3f, desc 32, normal 3f, mind f, maxd f, bad i, nobs i]
// [ids of Cur (N_total i), KA (N_total i), KB (N_total i)][nfound, ids][nloop, ids][vpMatched init KA.N i][Scw 16f][th_loop i]
// [s12 f][R12 9f][t12 3f][vpMatches12 init KA.N i][th_sim3 f][nfuse, ids][Calib 12f][th_fuse f][th_reloc f][ORBdist i][check_ori i]
static int do_f4(const std::vector<unsigned char>& buf, FILE* f) {
    Reader R{buf.data()};
    Frame Cur, FA, FB;
    read_frame(R, Cur); read_frame(R, FA); read_frame(R, FB);
    for (Frame* F : {&Cur, &FA, &FB}) { F->mfLogScaleFactor = std::log(1.2f); F->mnScaleLevels = 8; }
    KeyFrame KA, KB;
    to_keyframe(FA, KA); to_keyframe(FB, KB);
    { std::vector<float> T = R.arr<float>(16); for (int i = 0; i < 16; ++i) KB.Tcw_cam2.at<float>(i / 4, i % 4) = T[i]; }
    const int M = R.get<int>();
    std::vector<MapPoint> pool(M);
    for (int i = 0; i < M; ++i) {
        MapPoint& mp = pool[i];
        std::vector<float> xyz = R.arr<float>(3);
        mp.mWorldPos = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; ++k) mp.mWorldPos.at<float>(k) = xyz[k];
        mp.mDescriptor = cv::Mat(1, 32, CV_8U);
        std::vector<unsigned char> d = R.arr<unsigned char>(32); std::memcpy(mp.mDescriptor.ptr(0), d.data(), 32);
        std::vector<float> nv = R.arr<float>(3);
        mp.mNormalVector = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; ++k) mp.mNormalVector.at<float>(k) = nv[k];
        mp.mfMinDistance = R.get<float>(); mp.mfMaxDistance = R.get<float>();
        mp.mbBad = R.get<int>() != 0; mp.nObs = R.get<int>();
    }
    auto ids_to = [&](std::vector<MapPoint*>& v, int n, KeyFrame* owner) {
        std::vector<int> ids = R.arr<int>(n);
        v.assign(n, nullptr);
        for (int g = 0; g < n; ++g) if (ids[g] >= 0) { v[g] = &pool[ids[g]]; if (owner) pool[ids[g]].mObservations[owner] = g; }
    };
    auto id_of = [&](MapPoint* p) { return p ? (int)(p - pool.data()) : -1; };
    ids_to(Cur.mvpMapPoints, Cur.N_total, nullptr); ids_to(KA.mvpMapPoints, KA.N_total, &KA); ids_to(KB.mvpMapPoints, KB.N_total, &KB);
    KeyFrame KA1 = KA;   // the keyframe as it is before any fuse touches it: Fuse_cam1 runs on this copy at the end
    std::set<MapPoint*> found;
    { int n = R.get<int>(); std::vector<int> ids = R.arr<int>(n); for (int i : ids) found.insert(&pool[i]); }
    std::vector<MapPoint*> loop_pts;
    { int n = R.get<int>(); std::vector<int> ids = R.arr<int>(n); for (int i : ids) loop_pts.push_back(&pool[i]); }
    std::vector<MapPoint*> vpMatched; ids_to(vpMatched, KA.N, nullptr);
    cv::Mat Scw(4, 4, CV_32F);
    { std::vector<float> T = R.arr<float>(16); for (int i = 0; i < 16; ++i) Scw.at<float>(i / 4, i % 4) = T[i]; }
    const int th_loop = R.get<int>();
    const float s12 = R.get<float>();
    cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
    { std::vector<float> v = R.arr<float>(9); for (int i = 0; i < 9; ++i) R12.at<float>(i / 3, i % 3) = v[i]; }
    { std::vector<float> v = R.arr<float>(3); for (int i = 0; i < 3; ++i) t12.at<float>(i) = v[i]; }
    std::vector<MapPoint*> vpMatches12; ids_to(vpMatches12, KA.N, nullptr);
    const float th_sim3 = R.get<float>();
    std::vector<MapPoint*> fuse_pts;
    { int n = R.get<int>(); std::vector<int> ids = R.arr<int>(n); for (int i : ids) fuse_pts.push_back(i >= 0 ? &pool[i] : nullptr); }
    cv::Mat calib(4, 3, CV_32F);
    { std::vector<float> cm = R.arr<float>(12); for (int i = 0; i < 12; ++i) calib.at<float>(i / 3, i % 3) = cm[i]; }
    const float th_fuse = R.get<float>(), th_reloc = R.get<float>();
    const int ORBdist = R.get<int>(), check_ori = R.get<int>();
    // inputs of the two-camera overloads and of SearchForInitialization
    std::vector<MapPoint*> vpMatchedFull; ids_to(vpMatchedFull, KA.N_total, nullptr);
    std::vector<MapPoint*> vpMatches12Full; ids_to(vpMatches12Full, KA.N_total, nullptr);
    std::vector<float> prev_x = R.arr<float>(FA.N), prev_y = R.arr<float>(FA.N);
    const int window_size = R.get<int>();

    ORBmatcher m(0.9f, check_ori != 0);
    const int n1 = m.SearchByProjection(Cur, &KA, found, th_reloc, ORBdist);
    put(f, &n1, 4);
    for (int g = 0; g < Cur.N_total; ++g) { int id = id_of(Cur.mvpMapPoints[g]); put(f, &id, 4); }
    const int n2 = m.SearchByProjection_cam1(&KA, Scw, loop_pts, vpMatched, th_loop);
    put(f, &n2, 4);
    for (int g = 0; g < KA.N; ++g) { int id = id_of(vpMatched[g]); put(f, &id, 4); }
    const int n3 = m.SearchBySim3_cam1(&KA, &KB, vpMatches12, s12, R12, t12, th_sim3);
    put(f, &n3, 4);
    for (int g = 0; g < KA.N; ++g) { int id = id_of(vpMatches12[g]); put(f, &id, 4); }
    // the overloads over both cameras of the rig (reference src/ORBmatcher.cc:566-750, :2814-3135) and SearchForInitialization
    std::vector<int> loop_cams(loop_pts.size(), 0);
    const int n6 = m.SearchByProjection(&KA, Scw, loop_pts, loop_cams, vpMatchedFull, th_loop, calib);
    put(f, &n6, 4);
    for (int g = 0; g < KA.N_total; ++g) { int id = id_of(vpMatchedFull[g]); put(f, &id, 4); }
    const int n7 = m.SearchBySim3(&KA, &KB, vpMatches12Full, s12, R12, t12, th_sim3, calib);
    put(f, &n7, 4);
    for (int g = 0; g < KA.N_total; ++g) { int id = id_of(vpMatches12Full[g]); put(f, &id, 4); }
    std::vector<cv::Point2f> vbPrev(FA.N);
    for (int i = 0; i < FA.N; ++i) vbPrev[i] = cv::Point2f(prev_x[i], prev_y[i]);
    std::vector<int> vnMatches12;
    const int n8 = m.SearchForInitialization(FA, FB, vbPrev, vnMatches12, window_size);
    put(f, &n8, 4);
    put(f, vnMatches12.data(), vnMatches12.size() * 4);
    for (int i = 0; i < FA.N; ++i) { put(f, &vbPrev[i].x, 4); put(f, &vbPrev[i].y, 4); }
    const int n4 = m.Fuse(&KB, fuse_pts, calib, th_fuse);
    put(f, &n4, 4);
    for (int g = 0; g < KB.N_total; ++g) { int id = id_of(KB.mvpMapPoints[g]); put(f, &id, 4); }
    for (int i = 0; i < M; ++i) { int rep = id_of(pool[i].mpReplaced), bad = pool[i].mbBad ? 1 : 0; put(f, &rep, 4); put(f, &bad, 4); }
    // Fuse through the Sim3 pose of the loop: the loop points into KA (both cameras)
    std::vector<MapPoint*> vpReplace(loop_pts.size(), nullptr);
    std::vector<int> cams(loop_pts.size(), 0);
    const int n5 = m.Fuse(&KA, Scw, loop_pts, cams, th_fuse + 1.0f, vpReplace, calib);
    put(f, &n5, 4);
    for (int g = 0; g < KA.N_total; ++g) { int id = id_of(KA.mvpMapPoints[g]); put(f, &id, 4); }
    for (size_t i = 0; i < vpReplace.size(); ++i) { int id = id_of(vpReplace[i]); put(f, &id, 4); }
    // the camera-1 form of the same fuse (reference :2518-2813)
    std::vector<MapPoint*> vpReplace1(loop_pts.size(), nullptr);
    const int n9 = m.Fuse_cam1(&KA1, Scw, loop_pts, th_fuse + 1.0f, vpReplace1);
    put(f, &n9, 4);
    for (int g = 0; g < KA1.N_total; ++g) { int id = id_of(KA1.mvpMapPoints[g]); put(f, &id, 4); }
    for (size_t i = 0; i < vpReplace1.size(); ++i) { int id = id_of(vpReplace1[i]); put(f, &id, 4); }
    return 0;
}

// dropin <stream.bin> <out.bin> <batch 0|1>
// The reference's per-frame call pattern on its tracking thread, timed: Frame::Frame runs the two extractors back to back
// (reference src/Frame.cc:182,185), merges their outputs (:191-239) and fills mvuRight / mvDepth from the depth image
// (:959-986); TrackWithMotionModel then constructs an ORBmatcher on the stack and searches the last frame's points in the
// new frame (reference src/Tracking.cc:1254-1267).  batch = 1 replaces the two operator() calls by ONE ExtractBatch call (the
// integration INTEGRATION.md recommends).  The Frame assembly in between is the reference's own host code and is timed
// separately (it is not part of the accelerated path).
// stream.bin: int32 {w, h, nf0, nf1, T, iters, warmup}; float32 {fx, fy, cx, cy, mbf, th, du, dv}; T x 2 images (w*h u8);
//             2 depth images (w*h f32).
// out.bin: float32 medians {extract0, extract1, frame_host, search, class_calls_total, loop_total, search: host projection,
//          search: frame hash + upload, search: device} in us + {frame-cache hits, misses, failed calls}, then the per-step
//          class-call times (iters f32), then for the last step: both cameras' keypoints + descriptors, nmatches, per current
//          feature the index of the last-frame feature whose point it received (-1 none), then the last frame: N_total,
//          world points (3 f32 each), has-point flags, octaves, angles, descriptors, cameras.
#include <algorithm>
#include <chrono>
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

static int run_dropin(int argc, char** argv) {
    std::vector<unsigned char> buf = slurp(argv[2]);
    Reader R{buf.data()};
    const int w = R.get<int>(), h = R.get<int>(), nf0 = R.get<int>(), nf1 = R.get<int>(), T = R.get<int>(), iters = R.get<int>(),
              warmup = R.get<int>();
    const float fx = R.get<float>(), fy = R.get<float>(), cx = R.get<float>(), cy = R.get<float>(), mbf = R.get<float>(),
                th = R.get<float>(), du = R.get<float>(), dv = R.get<float>();
    const bool batch = std::atoi(argv[4]) != 0;
    std::vector<std::vector<unsigned char> > img(2 * T);
    for (int i = 0; i < 2 * T; ++i) img[i] = R.arr<unsigned char>((size_t)w * h);
    std::vector<float> depth[2] = {R.arr<float>((size_t)w * h), R.arr<float>((size_t)w * h)};
    ORBextractor ex0(nf0, 1.2f, 8, 20, 7), ex1(nf1, 1.2f, 8, 20, 7);   // Tracking.cc:144-145
    const std::vector<float> sf = ex0.GetScaleFactors();
    cv::Mat calib = cv::Mat::zeros(4, 3, CV_32F);
    for (int i = 0; i < 3; ++i) calib.at<float>(i, i) = 1.f;   // cam 2 = cam 1 (identity extrinsics, zero baseline between the rigs' cameras)

    struct Step { Frame F; std::vector<MapPoint> pool; std::vector<cv::KeyPoint> k[2]; cv::Mat d[2]; std::vector<float> world; };
    std::unique_ptr<Step> last, cur;
    std::vector<double> t_e0, t_e1, t_fr, t_s, t_calls, t_loop, t_brk[3];
    std::vector<int> match_src;
    int nmatches = 0;
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int it = 0; it < warmup + iters; ++it) {
        const int t = it % T;
        cur.reset(new Step());
        Step& S = *cur;
        cv::Mat im0(h, w, CV_8UC1, img[2 * t].data()), im1(h, w, CV_8UC1, img[2 * t + 1].data());
        const clk::time_point a0 = clk::now();
        clk::time_point a1;
        if (batch) {
            std::vector<std::vector<cv::KeyPoint> > kk; std::vector<cv::Mat> dd;
            ORBextractor::ExtractBatch({&ex0, &ex1}, {im0, im1}, kk, dd);
            S.k[0].swap(kk[0]); S.k[1].swap(kk[1]); S.d[0] = dd[0]; S.d[1] = dd[1];
            a1 = clk::now();
        } else {
            ex0(im0, cv::Mat(), S.k[0], S.d[0]);
            a1 = clk::now();
            ex1(im1, cv::Mat(), S.k[1], S.d[1]);
        }
        const clk::time_point a2 = clk::now();
        // ---- the reference's host-side Frame assembly (merge, maps, ComputeStereoFromRGBD; k1 == 0: no undistortion)
        Frame& F = S.F;
        F.N = (int)S.k[0].size(); F.N_cam2 = (int)S.k[1].size(); F.N_total = F.N + F.N_cam2;
        F.mvKeys_total.reserve(F.N_total);
        for (int c = 0; c < 2; ++c) F.mvKeys_total.insert(F.mvKeys_total.end(), S.k[c].begin(), S.k[c].end());
        F.mvKeysUn_total = F.mvKeys_total;
        F.mvKeysUn.assign(F.mvKeys_total.begin(), F.mvKeys_total.begin() + F.N);
        F.mvuRight_total.assign(F.N_total, -1.f); F.mvDepth_total.assign(F.N_total, -1.f);
        for (int g = 0; g < F.N_total; ++g) {
            const int c = g < F.N ? 0 : 1;
            F.keypoint_to_cam[g] = c; F.cont_idx_to_local_cam_idx[g] = c == 0 ? g : g - F.N;
            const cv::KeyPoint& kp = F.mvKeys_total[g];
            const float d = depth[c][(size_t)(int)kp.pt.y * w + (int)kp.pt.x];
            if (d > 0) { F.mvDepth_total[g] = d; F.mvuRight_total[g] = F.mvKeysUn_total[g].pt.x - mbf / d; }
        }
        F.mvuRight.assign(F.mvuRight_total.begin(), F.mvuRight_total.begin() + F.N);
        F.mDescriptors_total = {S.d[0], S.d[1]}; F.mDescriptors = S.d[0];
        F.mvpMapPoints.assign(F.N_total, nullptr); F.mvbOutlier.assign(F.N_total, false);
        F.mvScaleFactors = sf;
        F.mTcw = cv::Mat::eye(4, 4, CV_32F);
        F.fx = fx; F.fy = fy; F.cx = cx; F.cy = cy; F.mbf = mbf; F.mb = mbf / fx;
        F.mnMinX = 0; F.mnMinY = 0; F.mnMaxX = (float)w; F.mnMaxY = (float)h;
        // every feature carries a map point (observed once) that the stream's motion moves by (du, dv) pixels
        S.pool.resize(F.N_total); S.world.resize((size_t)3 * F.N_total);
        for (int g = 0; g < F.N_total; ++g) {
            const cv::KeyPoint& kp = F.mvKeysUn_total[g];
            const float z = F.mvDepth_total[g] > 0 ? F.mvDepth_total[g] : 2.f;
            MapPoint& mp = S.pool[g];
            mp.mWorldPos = cv::Mat(3, 1, CV_32F);
            mp.mWorldPos.at<float>(0) = (kp.pt.x + du - cx) / fx * z; mp.mWorldPos.at<float>(1) = (kp.pt.y + dv - cy) / fy * z;
            mp.mWorldPos.at<float>(2) = z;
            for (int k = 0; k < 3; ++k) S.world[(size_t)3 * g + k] = mp.mWorldPos.at<float>(k);
            mp.mDescriptor = S.d[g < F.N ? 0 : 1].row(g < F.N ? g : g - F.N);
            mp.nObs = 1;
        }
        const clk::time_point a3 = clk::now();
        double search_us = 0;
        if (last) {
            Frame& L = last->F;
            for (int g = 0; g < L.N_total; ++g) L.mvpMapPoints[g] = &last->pool[g];
            std::fill(F.mvpMapPoints.begin(), F.mvpMapPoints.end(), static_cast<MapPoint*>(NULL));   // Tracking.cc:1254
            const clk::time_point b0 = clk::now();
            ORBmatcher matcher(0.9f, true);                                                           // Tracking.cc:1237
            nmatches = matcher.SearchByProjection(F, L, th, false, calib);                            // Tracking.cc:1267
            search_us = us(b0, clk::now());
            float b3[3]; ORBmatcher::LastCallBreakdown(b3);
            if (it >= warmup) for (int k = 0; k < 3; ++k) t_brk[k].push_back(b3[k]);
            if (it == warmup + iters - 1) {
                match_src.assign(F.N_total, -1);
                for (int g = 0; g < F.N_total; ++g)
                    if (F.mvpMapPoints[g]) match_src[g] = (int)(F.mvpMapPoints[g] - last->pool.data());
            }
        }
        const clk::time_point a4 = clk::now();
        if (it >= warmup) {
            t_e0.push_back(us(a0, a1)); t_e1.push_back(us(a1, a2)); t_fr.push_back(us(a2, a3)); t_s.push_back(search_us);
            t_calls.push_back(us(a0, a2) + search_us); t_loop.push_back(us(a0, a4));
        }
        if (it + 1 < warmup + iters) last = std::move(cur);
    }
    FILE* f = std::fopen(argv[3], "wb");
    unsigned long hits = 0, misses = 0;
    ORBmatcher::FrameCacheStats(&hits, &misses);
    unsigned long served = 0, unserved = 0;
    ORBmatcher::ResidentStats(&served, &unserved);
    const float meds[14] = {(float)med(t_e0), (float)med(t_e1), (float)med(t_fr), (float)med(t_s), (float)med(t_calls), (float)med(t_loop),
                            (float)med(t_brk[0]), (float)med(t_brk[1]), (float)med(t_brk[2]), (float)hits, (float)misses,
                            (float)(ORBmatcher::FailureCount() + ORBextractor::FailureCount()), (float)served, (float)unserved};
    put(f, meds, sizeof(meds));
    for (double v : t_calls) { const float x = (float)v; put(f, &x, 4); }
    for (int c = 0; c < 2; ++c) {
        const int n = (int)cur->k[c].size();
        put(f, &n, 4); put(f, cur->k[c].data(), (size_t)n * sizeof(cv::KeyPoint));
        if (n) put(f, cur->d[c].ptr(0), (size_t)n * 32);
    }
    put(f, &nmatches, 4);
    put(f, match_src.data(), match_src.size() * 4);
    const Frame& L = last->F;
    put(f, &L.N_total, 4); put(f, &L.N, 4);
    put(f, last->world.data(), last->world.size() * 4);
    for (int g = 0; g < L.N_total; ++g) { const int o = L.mvKeys_total[g].octave; put(f, &o, 4); }
    for (int g = 0; g < L.N_total; ++g) { const float a = L.mvKeysUn_total[g].angle; put(f, &a, 4); }
    for (int c = 0; c < 2; ++c) { const int n = c == 0 ? L.N : L.N_cam2; if (n) put(f, last->d[c].ptr(0), (size_t)n * 32); }
    std::fclose(f);
    return 0;
}

// threads <match_case.bin> <bow_case.bin> <f4_case.bin> <iters>
// The reference calls ORBmatcher from three threads at once: Tracking (SearchByProjection, src/Tracking.cc:1267,1764),
// LocalMapping (SearchForTriangulation + Fuse, src/LocalMapping.cc:361,741) and LoopClosing (SearchByBoW, SearchBySim3,
// SearchByProjection, Fuse; src/LoopClosing.cc:362-536,841).  Here: the `match` case on one thread, the `bow` case (both
// SearchByBoW forms, SearchForTriangulation) on a second, the `f4` case (relocalisation / loop searches, both SearchBySim3
// forms, both Fuse overloads) on a third, each repeated `iters` times CONCURRENTLY on freshly parsed frames (so every
// iteration uploads its frames again), every iteration's output compared byte for byte with the same case run alone.
#include <atomic>
#include <thread>
static std::vector<unsigned char> run_mem(int (*fn)(const std::vector<unsigned char>&, FILE*), const std::vector<unsigned char>& buf, int* rc) {
    char* p = nullptr; size_t n = 0;
    FILE* f = open_memstream(&p, &n);
    *rc = fn(buf, f);
    std::fclose(f);
    std::vector<unsigned char> out(p, p + n);
    std::free(p);
    return out;
}

static int run_threads(int argc, char** argv) {
    typedef int (*Fn)(const std::vector<unsigned char>&, FILE*);
    const Fn fns[3] = {do_match, do_bow, do_f4};
    std::vector<unsigned char> cases[3] = {slurp(argv[2]), slurp(argv[3]), slurp(argv[4])};
    const int iters = std::atoi(argv[5]);
    std::vector<unsigned char> serial[3];
    for (int k = 0; k < 3; ++k) {
        int rc = 0;
        serial[k] = run_mem(fns[k], cases[k], &rc);
        if (rc || serial[k].empty()) { std::fprintf(stderr, "threads: serial run of case %d failed (%d)\n", k, rc); return 5; }
    }
    std::atomic<int> mismatches{0}, errors{0}, started{0};
    std::vector<std::thread> th;
    for (int k = 0; k < 3; ++k)
        th.emplace_back([&, k]() {
            ++started;
            while (started.load() < 3) std::this_thread::yield();      // all three loops begin together
            for (int it = 0; it < iters; ++it) {
                int rc = 0;
                std::vector<unsigned char> out = run_mem(fns[k], cases[k], &rc);
                if (rc) ++errors;
                if (out != serial[k]) ++mismatches;
            }
        });
    for (std::thread& t : th) t.join();
    const unsigned long fails = ORBmatcher::FailureCount() + ORBextractor::FailureCount();
    std::printf("threads: 3 x %d concurrent iterations, %d mismatches, %d errors, %lu failed device calls\n", iters, mismatches.load(),
                errors.load(), fails);
    return (mismatches.load() || errors.load() || fails) ? 6 : 0;
}

// rt <n>: R * x + t on CV_32F cv::Mat objects (what the reference's projection loops write) against the scalar routine the
// per-frame tracking search uses (ORBmatcher::DebugApplyRt), bit for bit, on n random poses x points: rotations with entries in
// [-1, 1], translations and points over eight decades of magnitude, plus chained application (cam 2 behind cam 1).  No GPU.
static int run_rt(int argc, char** argv) {
    const long n = std::atol(argv[2]);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    auto unit = [&]() { return (float)((double)(rnd() >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    long bad = 0;
    cv::Mat Rm(3, 3, CV_32F), tm(3, 1, CV_32F), xm(3, 1, CV_32F), R2(3, 3, CV_32F), t2(3, 1, CV_32F);
    for (long i = 0; i < n; ++i) {
        const float mag_t = std::pow(10.f, (float)(rnd() % 8) - 3.f), mag_x = std::pow(10.f, (float)(rnd() % 8) - 3.f);
        for (int k = 0; k < 9; ++k) { Rm.at<float>(k / 3, k % 3) = unit(); R2.at<float>(k / 3, k % 3) = unit(); }
        for (int k = 0; k < 3; ++k) { tm.at<float>(k) = unit() * mag_t; t2.at<float>(k) = unit() * mag_t; xm.at<float>(k) = unit() * mag_x; }
        cv::Mat y = Rm * xm + tm;
        cv::Mat z = R2 * y + t2;
        const float x[3] = {xm.at<float>(0), xm.at<float>(1), xm.at<float>(2)};
        float ys[3], zs[3];
        ORBmatcher::DebugApplyRt(Rm, tm, x, ys);
        ORBmatcher::DebugApplyRt(R2, t2, ys, zs);
        for (int k = 0; k < 3; ++k) {
            if (std::memcmp(&ys[k], &y.at<float>(k), 4) != 0 || std::memcmp(&zs[k], &z.at<float>(k), 4) != 0) ++bad;
        }
    }
    // CheckDistEpipolarLine (protected in the reference, src/ORBmatcher.cc:167-184): with F12 chosen so that the epipolar line of
    // (u, v) is x = u, a point 2 px off the line passes at sigma^2 = 1.44 (4 < 3.84 * 1.44) and fails at sigma^2 = 1; a zero line fails
    struct Probe : ORBmatcher { Probe() : ORBmatcher(0.6f, true) {} using ORBmatcher::CheckDistEpipolarLine; } probe;
    cv::Mat F12 = cv::Mat::zeros(3, 3, CV_32F);
    F12.at<float>(2, 0) = 1.f; F12.at<float>(0, 2) = -1.f;      // l = (1, 0, -u)
    KeyFrame kf;
    kf.mvLevelSigma2 = {1.0f, 1.44f};
    cv::KeyPoint k1, k2;
    k1.pt.x = 5.f; k1.pt.y = 9.f; k2.pt.x = 7.f; k2.pt.y = -3.f;
    k2.octave = 1; if (!probe.CheckDistEpipolarLine(k1, k2, F12, &kf)) ++bad;
    k2.octave = 0; if (probe.CheckDistEpipolarLine(k1, k2, F12, &kf)) ++bad;
    if (probe.CheckDistEpipolarLine(k1, k2, cv::Mat::zeros(3, 3, CV_32F), &kf)) ++bad;
    std::printf("rt: %ld poses x points, %ld differing floats\n", n, bad);
    return bad ? 7 : 0;
}


// gemm: known answers for the two evaluation orders of cv::gemm that cv_compat.h restates (small-matrix float block / general
// double path), for the folding of `A*B + C` into one call, for alpha = -1, for cv::solve's LU and cv::invert's 3x3 closed form.
// Every expected value is worked out by hand in the comments.  No GPU.
static int run_gemm_kat() {
    int bad = 0;
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto expect = [&](const char* what, float got, uint32_t want) {
        if (bits(got) != want) { std::printf("gemm KAT %s: got %08x, expected %08x\n", what, bits(got), want); ++bad; }
    };
    const float e = std::ldexp(1.f, -24);   // half an ulp of 1.0f
    // (1) row (1, e, e) . (1, 1, 1).  Float order: 1 + e = 1 (tie, even), + e = 1 -> 3f800000.
    //     Double order: 1 + 2e = 1 + 2^-23 exactly -> 3f800001.
    cv::Mat A = cv::Mat::zeros(3, 3, CV_32F), x(3, 1, CV_32F);
    A.at<float>(0, 0) = 1; A.at<float>(0, 1) = e; A.at<float>(0, 2) = e;
    for (int k = 0; k < 3; ++k) x.at<float>(k) = 1;
    cv::Mat y = A * x;                                   // 3x3 * 3x1, no flags: small block
    expect("small 3x3*3x1", y.at<float>(0), 0x3f800000u);
    cv::Mat At = A.t();                                  // evaluated transpose; At.t() * x is GEMM_1_T: general path
    cv::Mat yg = At.t() * x;
    expect("general (A_T) 3x3*3x1", yg.at<float>(0), 0x3f800001u);
    expect("gemm() small", cv::gemm(A, x, 1, cv::Mat(), 0, 0).at<float>(0), 0x3f800000u);
    expect("gemm() general", cv::gemm(At, x, 1, cv::Mat(), 0, cv::GEMM_1_T).at<float>(0), 0x3f800001u);
    // a 1x3 * 3x1 product has len 3 but neither side of D is 3: general path -> 3f800001
    expect("1x3*3x1 is general", ((cv::Mat)(A.row(0) * x)).at<float>(0), 0x3f800001u);
    // 3x3 * 3x3 (row form of the small block): column of ones -> 3f800000
    cv::Mat ones3(3, 3, CV_32F); for (int k = 0; k < 9; ++k) ones3.at<float>(k / 3, k % 3) = 1;
    expect("small 3x3*3x3", ((cv::Mat)(A * ones3)).at<float>(0, 2), 0x3f800000u);
    // (2) len 4: row (1, e, e, e) . ones.  Float: 1.  Double: 1 + 3e = 1 + 1.5 ulp -> tie between 1+ulp and 1+2ulp -> even: 3f800002.
    cv::Mat T = cv::Mat::zeros(4, 4, CV_32F), ones4(4, 4, CV_32F);
    T.at<float>(0, 0) = 1; T.at<float>(0, 1) = e; T.at<float>(0, 2) = e; T.at<float>(0, 3) = e;
    for (int k = 0; k < 16; ++k) ones4.at<float>(k / 4, k % 4) = 1;
    expect("small 4x4*4x4", ((cv::Mat)(T * ones4)).at<float>(0, 1), 0x3f800000u);
    expect("general 4x4*4x4t", ((cv::Mat)(T * ones4.t())).at<float>(0, 1), 0x3f800002u);
    // (3) folding: row of zeros times (-1,-1,-1): t = -0.0f.  Alone: (float)(-0.0*1 + 0.0f*0) = +0.0 -> 00000000.
    //     `Z*m + c` with c = -0.0f is ONE call: (float)(-0.0 + -0.0) = -0.0 -> 80000000 (product-then-sum would give +0.0 + -0.0 = +0.0).
    cv::Mat Z = cv::Mat::zeros(3, 3, CV_32F), m(3, 1, CV_32F), c(3, 1, CV_32F);
    for (int k = 0; k < 3; ++k) { m.at<float>(k) = -1; c.at<float>(k) = -0.0f; }
    expect("product alone", ((cv::Mat)(Z * m)).at<float>(0), 0x00000000u);
    expect("A*B + C folded", ((cv::Mat)(Z * m + c)).at<float>(0), 0x80000000u);
    expect("C + A*B folded", ((cv::Mat)(c + Z * m)).at<float>(0), 0x80000000u);
    // (4) -A*B is alpha = -1 on the same call: t = 1 -> -1 (bf800000); t = +0.0: (+0.0 * -1) + 0 = +0.0 (00000000, not 80000000)
    cv::Mat I = cv::Mat::eye(3, 3, CV_32F);
    expect("-I*x", ((cv::Mat)(-I * x)).at<float>(1), 0xbf800000u);
    cv::Mat zero3 = cv::Mat::zeros(3, 1, CV_32F);
    expect("-I*0", ((cv::Mat)(-I * zero3)).at<float>(1), 0x00000000u);
    // C - A*B: alpha = -1, beta = 1: (2,2,2) - I*(1,1,1) = 1
    cv::Mat two(3, 1, CV_32F); for (int k = 0; k < 3; ++k) two.at<float>(k) = 2;
    expect("C - A*B", ((cv::Mat)(two - I * x)).at<float>(2), 0x3f800000u);
    // -R.t() * t: the transpose is evaluated, alpha = -1, no flag -> small block (float order): row (1,e,e) again -> -1 exactly
    expect("-A.t().t()*x small", ((cv::Mat)(-At.t() * x)).at<float>(0), 0xbf800000u);
    // s*A.t() and A/s: float scale kernel x*(float)s + 0
    expect("0.5*A.t()", ((cv::Mat)(0.5 * A.t())).at<float>(0, 0), 0x3f000000u);
    expect("A/4", ((cv::Mat)(A / 4.0)).at<float>(0, 0), 0x3e800000u);
    // (5) cv::solve LU with a row swap: A = [[0,1,0],[2,0,0],[0,0,4]], B = I -> X = [[0,.5,0],[1,0,0],[0,0,.25]] (all exact)
    cv::Mat S = cv::Mat::zeros(3, 3, CV_32F);
    S.at<float>(0, 1) = 1; S.at<float>(1, 0) = 2; S.at<float>(2, 2) = 4;
    cv::Mat X = S.inv() * I;                             // inv() * Mat -> solve
    expect("solve (0,1)", X.at<float>(0, 1), 0x3f000000u);
    expect("solve (1,0)", X.at<float>(1, 0), 0x3f800000u);
    expect("solve (2,2)", X.at<float>(2, 2), 0x3e800000u);
    expect("solve (0,0)", X.at<float>(0, 0), 0x00000000u);
    // singular -> zeros
    expect("solve singular", ((cv::Mat)(Z.inv() * I)).at<float>(1, 1), 0x00000000u);
    // (6) cv::invert 3x3 closed form: diag(2,4,8) -> diag(.5,.25,.125)
    cv::Mat Dg = cv::Mat::zeros(3, 3, CV_32F);
    Dg.at<float>(0, 0) = 2; Dg.at<float>(1, 1) = 4; Dg.at<float>(2, 2) = 8;
    cv::Mat Di = Dg.inv();
    expect("inv (0,0)", Di.at<float>(0, 0), 0x3f000000u);
    expect("inv (1,1)", Di.at<float>(1, 1), 0x3e800000u);
    expect("inv (2,2)", Di.at<float>(2, 2), 0x3e000000u);
    // I * D.inv(): the inverse on the right is evaluated (no solve)
    expect("I*inv", ((cv::Mat)(I * Dg.inv())).at<float>(2, 2), 0x3e000000u);
    // (7) an aliased destination reads its operands first: x = P*x + t with a permutation P
    cv::Mat P = cv::Mat::zeros(3, 3, CV_32F), v(3, 1, CV_32F);
    P.at<float>(0, 1) = 1; P.at<float>(1, 2) = 1; P.at<float>(2, 0) = 1;
    v.at<float>(0) = 1; v.at<float>(1) = 2; v.at<float>(2) = 3;
    v = P * v + zero3;
    expect("aliased (0)", v.at<float>(0), 0x40000000u);
    expect("aliased (1)", v.at<float>(1), 0x40400000u);
    expect("aliased (2)", v.at<float>(2), 0x3f800000u);
    std::printf("gemm: %d known answers wrong\n", bad);
    return bad ? 8 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: test_host extract|batch|match|bow ...\n"); return 1; }
    const std::string mode = argv[1];
    if (mode == "extract" && argc >= 7) return run_extract(argc, argv);
    if (mode == "batch" && argc >= 9) return run_batch(argc, argv);
    if (mode == "match" && argc >= 4) return run_file(do_match, argv);
    if (mode == "bow" && argc >= 4) return run_file(do_bow, argv);
    if (mode == "f4" && argc >= 4) return run_file(do_f4, argv);
    if (mode == "threads" && argc >= 6) return run_threads(argc, argv);
    if (mode == "rt" && argc >= 3) return run_rt(argc, argv);
    if (mode == "gemm") return run_gemm_kat();
    if (mode == "dropin" && argc >= 5) return run_dropin(argc, argv);
    std::fprintf(stderr, "bad arguments\n");
    return 1;
}
