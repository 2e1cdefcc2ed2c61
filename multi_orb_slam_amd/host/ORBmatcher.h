// ORBmatcher.h -- drop-in for the hot subset of the reference's include/ORBmatcher.h:37-137: constructor defaults,
// static DescriptorDistance, the two tracking SearchByProjection overloads, the two SearchByBoW overloads,
// SearchForTriangulation, the public constants and the public mRcam21 / mtcam21 scratch.  Candidate gathering and Hamming distances run in libmorb.so's HIP kernels
// (include/orbm.h); the 3-D projection of map points and the order-dependent accept/overwrite/histogram logic stay on
// the host exactly where the reference has them; the BoW-gated searches run whole on the device (include/orbv.h).  The
// remaining projection searches (Fuse, SearchBySim3, loop/relocalisation SearchByProjection: SURVEY section 8 f4) are not
// part of this round.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <vector>
#include "cv_compat.h"
#include "slam_types.h"

struct orbm_matcher;
struct orbv_workspace;

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    ~ORBmatcher();
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // Computes the Hamming distance between two ORB descriptors
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches
    // Used to track the local map (Tracking)
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                           cv::Mat CalibMatrix);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame.
    // Brute force constrained to ORB that belong to the same vocabulary node (at a certain level)
    // Used in Relocalisation and Loop Detection
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo, std::vector<bool> vbCam);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    cv::Mat mRcam21;
    cv::Mat mtcam21;

protected:
    float RadiusByViewingCos(const float& viewCos);
    cv::Mat SkewSymmetricMatrix(const cv::Mat& v);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;

private:
    orbm_matcher* handle_ = nullptr;  // created on first GPU use; one per ORBmatcher (stack object, as in the reference)
    orbm_matcher* Handle();
    orbv_workspace* bow_ = nullptr;   // stream + scratch of the BoW-gated searches, created on first use
    orbv_workspace* Bow();
};

}  // namespace ORB_SLAM2

#endif
