// ORBmatcher.h -- drop-in for the hot subset of the reference's include/ORBmatcher.h:37-137: constructor defaults,
// static DescriptorDistance, the two tracking SearchByProjection overloads, the two SearchByBoW overloads,
// SearchForTriangulation, the relocalisation / loop-closing SearchByProjection overloads, SearchBySim3_cam1, both Fuse overloads,
// the public constants and the public mRcam21 / mtcam21 scratch.  Candidate gathering and Hamming distances run in libmorb.so's HIP kernels
// (include/orbm.h); the 3-D projection of map points and the order-dependent accept/overwrite/histogram logic stay on
// the host exactly where the reference has them; the BoW-gated searches run whole on the device (include/orbv.h).  Of
// the remaining overloads are here as well: the two-camera loop SearchByProjection, the two-camera SearchBySim3
// and SearchForInitialization (the reference's threads call the _cam1 forms; same device primitives, different host loops).
// MORB_DUMP_QUERIES=<file>: every projection search appends the queries it built (int32 count + orbm_query records), so a
// test can hold the device result against the oracle on exactly those queries.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
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
    // Declared by the reference (include/ORBmatcher.h:53,55) and defined nowhere in it (single-camera forms the multi-camera
    // fork left behind: a program that called them would not link there either).  Declared here for source compatibility,
    // likewise undefined: there is no reference behaviour to reproduce.
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection_cam1(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking)
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing)
    int SearchByProjection_cam1(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);

    // The same over both cameras of the keyframe: each point is projected into camera 1 and camera 2, the best candidate over
    // both windows wins (reference :566-750)
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<int>& vLoopMPCams,
                           std::vector<MapPoint*>& vpMatched, int th, const cv::Mat CalibMatrix);

    // Matching for the Map Initialization (only used in the monocular case)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    // In the stereo and RGB-D case, s12=1
    int SearchBySim3_cam1(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                          const cv::Mat& t12, const float th);
    // Both cameras of the rig: every point is searched in the grid of the camera it was observed in (reference :2814-3135)
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th, const cv::Mat CalibMatrix);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints.
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const cv::Mat CalibMatrix, const float th = 3.0);

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<int>& vLoopMPCams, float th,
             std::vector<MapPoint*>& vpReplacePoint, const cv::Mat CalibMatrix);
    // The camera-1 form (reference include/ORBmatcher.h:110, src/ORBmatcher.cc:2518-2813)
    int Fuse_cam1(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame.
    // Brute force constrained to ORB that belong to the same vocabulary node (at a certain level)
    // Used in Relocalisation and Loop Detection
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchByBoW_cam1(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW_cam1(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo, std::vector<bool> vbCam);
    // (include/ORBmatcher.h:90-91: the five-argument form is declared and never defined in the reference; same here)
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo);

    // public in the reference (include/ORBmatcher.h:116-117); ComputeF12 is declared there and defined nowhere (LocalMapping has
    // its own, src/LocalMapping.cc) -- declared, undefined
    cv::Mat SkewSymmetricMatrix(const cv::Mat& v);
    cv::Mat ComputeF12(KeyFrame*& pKF1, KeyFrame*& pKF2);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    cv::Mat mRcam21;
    cv::Mat mtcam21;

protected:
    // (reference include/ORBmatcher.h:129 / src/ORBmatcher.cc:167-184; the searches run the same test on the device)
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;

public:
    // MI355X additions.  There is no CPU fallback and the reference's signatures have no error channel: a failed device call
    // makes the search return 0 matches (the caller's containers are left as the reference's own prologue leaves them), is
    // reported once on stderr and kept here.  Thread-local text of the last failure on the calling thread / process-wide count.
    static const char* LastError();
    static unsigned long FailureCount();
    // inspection: host microseconds of the calling thread's last SearchByProjection(Frame&, const Frame&, ...) call spent
    // {projecting the points (the reference's cv::Mat algebra), hashing + uploading the frame, searching on the device},
    // and the hit / miss counts of the calling thread's cache of uploaded frames
    static void LastCallBreakdown(float* us3);
    static void FrameCacheStats(unsigned long* hits, unsigned long* misses);
    // process-wide: descriptor matrices of uploaded frames that were found still in HBM where ORBextractor left them (byte-equal,
    // same thread: host/resident.h) / that were sent from the host
    static void ResidentStats(unsigned long* served, unsigned long* missed);
    // test hook: R * x + t (3x3, 3x1 CV_32F) through the scalar routine the per-frame tracking search uses instead of three
    // cv::Mat temporaries per point; host/test_host `rt` compares it with the cv::Mat expression bit for bit
    static void DebugApplyRt(const cv::Mat& R, const cv::Mat& t, const float* x, float* out);

private:
    // The device state (matcher handle with its stream and scratch, BoW workspace, cache of uploaded frames) belongs to the
    // calling THREAD, created on its first GPU use and shared by every ORBmatcher object of that thread: the reference
    // constructs an ORBmatcher on the stack for each use, from three threads.
    orbm_matcher* Handle();
    orbv_workspace* Bow();
};

}  // namespace ORB_SLAM2

#endif
