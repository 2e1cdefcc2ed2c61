// slam_types.h -- minimal stand-ins for the reference's Frame (include/Frame.h:155-261), KeyFrame (include/KeyFrame.h:218-248)
// and MapPoint (include/MapPoint.h) carrying the members ORBmatcher's searches read or write, under the reference's names
// and with the reference's container types for everything the wrapper iterates or indexes (std::unordered_map<size_t,int>
// for the two index maps, std::vector<cv::Mat> for the per-camera descriptors, DBoW2's map-derived vectors).  What is NOT the
// reference's: the image bounds and the ids are per-object here and static members there (same spelling at the point of use),
// nNextId is an atomic accessor, MapPoint keeps its observations in a std::map<KeyFrame*,size_t> without the mutexes.
// A real integration compiles the wrapper against the reference's own headers instead: define MORB_USE_REFERENCE_TYPES and put
// the reference's include/ (and its root, for Thirdparty/DBoW2) on the include path; tests/test_reference_headers.py holds
// that build (-fsyntax-only, reference headers used in place) to zero errors on every CPU run.
#pragma once
#ifndef MORB_USE_REFERENCE_TYPES
#include <atomic>
#include <cmath>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>
#include "cv_compat.h"
#include "ORBVocabulary.h"

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;

class MapPoint {
public:
    cv::Mat GetWorldPos() { return mWorldPos; }
    cv::Mat GetDescriptor() { return mDescriptor; }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    // tracking scratch written by Frame::isInFrustum (src/Frame.cc:443-499)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1.f;

    // members the remaining projection searches read (src/MapPoint.cc:559-617)
    cv::Mat GetNormal() { return mNormalVector; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    template <class FrameOrKeyFrame> int PredictScale(const float& currentDist, FrameOrKeyFrame* pF) {
        const float ratio = mfMaxDistance / currentDist;
        int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
        return nScale;
    }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { auto it = mObservations.find(pKF); return it != mObservations.end() ? (int)it->second : -1; }
    int GetIndexInKeyFrame_cam1(KeyFrame* pKF) { auto it = mObservations.find(pKF); return it != mObservations.end() ? (int)it->second : -1; }
    void AddObservation(KeyFrame* pKF, size_t idx) { if (mObservations.count(pKF)) return; mObservations[pKF] = idx; nObs++; }
    void Replace(MapPoint* pMP) { if (pMP == this) return; mpReplaced = pMP; mbBad = true; }

    cv::Mat mWorldPos;    // 3x1 CV_32F
    cv::Mat mDescriptor;  // 1x32 CV_8U
    cv::Mat mNormalVector;  // 3x1 CV_32F
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::map<KeyFrame*, size_t> mObservations;
    MapPoint* mpReplaced = nullptr;
    int nObs = 1;
    bool mbBad = false;
};

class Frame {
public:
    // identity (include/Frame.h: `static long unsigned int nNextId; long unsigned int mnId;`, assigned in the constructors,
    // src/Frame.cc:257; copies keep it)
    // (atomic here: the stand-in's test driver constructs frames on several threads; the reference's tracking thread is alone)
    static std::atomic<long unsigned int>& NextId() { static std::atomic<long unsigned int> n{0}; return n; }
    long unsigned int mnId = NextId()++;
    // multi-camera "_total" view (src/Frame.cc:191-239): global index g, cam-major
    int N = 0, N_cam2 = 0, N_total = 0;
    std::vector<cv::KeyPoint> mvKeys_total, mvKeysUn_total, mvKeysUn;
    std::vector<float> mvuRight_total, mvuRight, mvDepth_total;
    std::vector<cv::Mat> mDescriptors_total;  // per camera, N_c x 32
    cv::Mat mDescriptors;                     // camera 1
    std::unordered_map<size_t, int> keypoint_to_cam, cont_idx_to_local_cam_idx;   // include/Frame.h:256,261 / include/KeyFrame.h:243,248
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors;
    cv::Mat mTcw;  // 4x4 CV_32F
    float fx = 0, fy = 0, cx = 0, cy = 0, mb = 0, mbf = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;  // static members in the reference
    DBoW2::BowVector mBowVec;       // all cameras (include/Frame.h:185-187)
    DBoW2::FeatureVector mFeatVec;
    DBoW2::BowVector mBowVec_cam1;  // camera 1 only (include/Frame.h:186,188)
    DBoW2::FeatureVector mFeatVec_cam1;
    std::vector<cv::KeyPoint> mvKeys;   // camera 1, distorted
    float mfLogScaleFactor = 0; int mnScaleLevels = 0;
};

// KeyFrame members the BoW-gated searches read (include/KeyFrame.h:53-59, :104-113, :218-219 and the Frame copies of
// src/KeyFrame.cc:31-80).  Poses are world -> camera, one per camera of the rig.
class KeyFrame {
public:
    // identity (include/KeyFrame.h: `static long unsigned int nNextId; long unsigned int mnId;`)
    static std::atomic<long unsigned int>& NextId() { static std::atomic<long unsigned int> n{0}; return n; }
    long unsigned int mnId = NextId()++;
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    cv::Mat GetDescriptor(const int& cam, const size_t& idx) const { return mDescriptors_total[cam].row((int)idx); }
    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
    cv::Mat GetRotation_cam2() { return Tcw_cam2.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation_cam2() { return Tcw_cam2.rowRange(0, 3).col(3).clone(); }
    cv::Mat GetCameraCenter() { return -(GetRotation().t() * GetTranslation()); }
    cv::Mat GetCameraCenter_cam2() { return -(GetRotation_cam2().t() * GetTranslation_cam2()); }

    std::vector<cv::KeyPoint> mvKeysUn_total;
    std::vector<float> mvuRight_total;
    std::vector<cv::Mat> mDescriptors_total;
    std::unordered_map<size_t, int> keypoint_to_cam, cont_idx_to_local_cam_idx;   // include/Frame.h:256,261 / include/KeyFrame.h:243,248
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<float> mvScaleFactors, mvLevelSigma2;
    DBoW2::BowVector mBowVec, mBowVec_cam1;
    DBoW2::FeatureVector mFeatVec, mFeatVec_cam1;
    cv::Mat mK;                      // 3x3 CV_32F
    float fx = 0, fy = 0, cx = 0, cy = 0;
    cv::Mat Tcw, Tcw_cam2;           // 4x4 CV_32F

    // members the remaining projection searches read
    std::vector<MapPoint*> GetMapPointMatches_cam1() { return std::vector<MapPoint*>(mvpMapPoints.begin(), mvpMapPoints.begin() + N); }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
    bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }
    int N = 0, N_cam2 = 0, N_total = 0;
    std::vector<cv::KeyPoint> mvKeysUn;   // camera 1
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;                 // camera 1
    std::vector<float> mvInvLevelSigma2;
    float mbf = 0, mfLogScaleFactor = 0; int mnScaleLevels = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
};

}  // namespace ORB_SLAM2
#else
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#endif
