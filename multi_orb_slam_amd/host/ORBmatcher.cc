// ORBmatcher.cc -- see ORBmatcher.h.  Host glue: packs flat arrays for include/orbm.h and writes the results back
// into the caller's Frame the way the reference does.
#include "ORBmatcher.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../include/orbm.h"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;     // reference src/ORBmatcher.cc:37
const int ORBmatcher::TH_LOW = 50;       // :38
const int ORBmatcher::HISTO_LENGTH = 30; // :39

static void die(const char* what, int rc) {
    std::fprintf(stderr, "ORBmatcher: %s failed (%d): %s\n", what, rc, orb_last_error());
    std::abort();
}

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    mRcam21 = cv::Mat(3, 3, CV_32F);
    mtcam21 = cv::Mat(3, 1, CV_32F);
}

ORBmatcher::~ORBmatcher() { orbm_destroy(handle_); }

orbm_matcher* ORBmatcher::Handle() {
    if (!handle_) {
        const char* dev = std::getenv("MORB_DEVICE");
        int rc = orbm_create(dev ? std::atoi(dev) : 0, &handle_);
        if (rc) die("orbm_create", rc);
    }
    return handle_;
}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return orbm_descriptor_distance(a.ptr(0), b.ptr(0));
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

struct FlatFrame {  // orbm_frame_desc backing store built from a Frame
    std::vector<float> x, y, ang, ur;
    std::vector<int32_t> oct, cam, loc;
    std::vector<const uint8_t*> desc;
    orbm_frame_desc d;
};

void flatten(const Frame& F, bool cam1_only, FlatFrame& ff) {
    const int n = cam1_only ? F.N : F.N_total;
    ff.x.resize(n); ff.y.resize(n); ff.ang.resize(n); ff.ur.resize(n); ff.oct.resize(n); ff.cam.resize(n); ff.loc.resize(n);
    const std::vector<cv::KeyPoint>& kun = cam1_only ? F.mvKeysUn : F.mvKeysUn_total;
    const std::vector<float>& ur = cam1_only ? F.mvuRight : F.mvuRight_total;
    for (int g = 0; g < n; ++g) {
        ff.x[g] = kun[g].pt.x; ff.y[g] = kun[g].pt.y; ff.ang[g] = kun[g].angle; ff.oct[g] = kun[g].octave;
        ff.ur[g] = ur[g];
        ff.cam[g] = cam1_only ? 0 : F.keypoint_to_cam.find(g)->second;
        ff.loc[g] = cam1_only ? g : F.cont_idx_to_local_cam_idx.find(g)->second;
    }
    ff.desc.clear();
    if (cam1_only) ff.desc.push_back(F.mDescriptors.ptr(0));
    else for (const cv::Mat& m : F.mDescriptors_total) ff.desc.push_back(m.empty() ? nullptr : m.ptr(0));
    ff.d.n_total = n; ff.d.n_cams = (int)ff.desc.size();
    ff.d.un_x = ff.x.data(); ff.d.un_y = ff.y.data(); ff.d.octave = ff.oct.data(); ff.d.angle = ff.ang.data();
    ff.d.uright = ff.ur.data(); ff.d.cam_of = ff.cam.data(); ff.d.local_of = ff.loc.data(); ff.d.desc = ff.desc.data();
    ff.d.min_x = F.mnMinX; ff.d.min_y = F.mnMinY; ff.d.max_x = F.mnMaxX; ff.d.max_y = F.mnMaxY;
}

}  // namespace

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
    FlatFrame ff;
    flatten(F, /*cam1_only=*/true, ff);
    std::vector<uint8_t> occupied(F.N, 0);
    for (int g = 0; g < F.N; ++g)
        occupied[g] = (F.mvpMapPoints[g] && F.mvpMapPoints[g]->Observations() > 0) ? 1 : 0;  // :107-109
    orbm_frame* fr = nullptr;
    int rc = orbm_frame_create(Handle(), &ff.d, &fr);
    if (rc) die("orbm_frame_create", rc);
    std::vector<int32_t> match(F.N > 0 ? F.N : 1);
    int nmatches = 0;
    rc = orbm_search_by_projection_points(Handle(), fr, q.data(), (int)q.size(), occupied.data(), mfNNratio, TH_HIGH,
                                          match.data(), &nmatches);
    orbm_frame_destroy(fr);
    if (rc) die("orbm_search_by_projection_points", rc);
    for (int g = 0; g < F.N; ++g)
        if (match[g] >= 0) F.mvpMapPoints[g] = qmp[match[g]];
    return nmatches;
}

// reference src/ORBmatcher.cc:3448-3641
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                                   cv::Mat CalibMatrix) {
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

    std::vector<orbm_query> q;
    std::vector<MapPoint*> qmp;
    q.reserve(LastFrame.N_total);
    for (int i = 0; i < LastFrame.N_total; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        int cam = LastFrame.keypoint_to_cam.find(i)->second;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        if (cam == 1) x3Dc = mRcam21 * x3Dc + mtcam21;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        int nLastOctave = LastFrame.mvKeys_total[i].octave;
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        orbm_query Q;
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
        q.push_back(Q); qmp.push_back(pMP);
    }

    FlatFrame ff;
    flatten(CurrentFrame, /*cam1_only=*/false, ff);
    orbm_frame* fr = nullptr;
    int rc = orbm_frame_create(Handle(), &ff.d, &fr);
    if (rc) die("orbm_frame_create", rc);
    std::vector<int32_t> match(CurrentFrame.N_total > 0 ? CurrentFrame.N_total : 1);
    int nmatches = 0;
    std::vector<uint8_t> occupied(match.size(), 0);  // :3566-3568 also skips points that were there before the call
    for (int g = 0; g < CurrentFrame.N_total; ++g)
        occupied[g] = (CurrentFrame.mvpMapPoints[g] && CurrentFrame.mvpMapPoints[g]->Observations() > 0) ? 1 : 0;
    rc = orbm_search_by_projection(Handle(), fr, q.data(), (int)q.size(), occupied.data(), TH_HIGH,
                                   mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    orbm_frame_destroy(fr);
    if (rc) die("orbm_search_by_projection", rc);
    // The reference starts from whatever CurrentFrame.mvpMapPoints holds (all NULL in TrackWithMotionModel,
    // src/Tracking.cc:1254) and only ever writes accepted matches / NULLs for histogram rejects.
    for (int g = 0; g < CurrentFrame.N_total; ++g) {
        if (match[g] >= 0) CurrentFrame.mvpMapPoints[g] = qmp[match[g]];
        else if (match[g] == -2) CurrentFrame.mvpMapPoints[g] = static_cast<MapPoint*>(NULL);  // histogram reject, :3631
    }
    return nmatches;
}

}  // namespace ORB_SLAM2
