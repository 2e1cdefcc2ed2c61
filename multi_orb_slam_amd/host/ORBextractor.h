// ORBextractor.h -- drop-in for the reference's include/ORBextractor.h:45-112: same namespace, class name, constructor
// and operator() signature, same getters, same public mvImagePyramid member.  All compute runs in libmorb.so's HIP
// kernels through include/orbx.h; this class only owns the handle and converts containers.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>
#include "cv_compat.h"

struct orbx_extractor;

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored (as in the reference).
    // Empty image: returns leaving the outputs untouched; zero keypoints: descriptors.release().
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Nobody reads this in the reference (only commented-out stereo code); it stays empty unless
    // MaterialisePyramid(true) asks for a host copy of the levels after every call.
    std::vector<cv::Mat> mvImagePyramid;
    void MaterialisePyramid(bool on) { materialise_ = on; }

    // MI355X addition: several extractors (one per camera, as the reference creates them in Tracking.cc:144-145)
    // run as ONE batched GPU call.  images[i] feeds extractors[i]; outputs as for operator().
    static void ExtractBatch(const std::vector<ORBextractor*>& extractors, const std::vector<cv::Mat>& images,
                             std::vector<std::vector<cv::KeyPoint> >& keypoints, std::vector<cv::Mat>& descriptors);

    // MI355X addition: there is no CPU fallback and the reference's signatures have no error channel, so a failed device
    // call leaves the outputs untouched (as an empty image does) and is reported here.  Thread-local text of the last
    // failure on the calling thread ("" if none) and the process-wide number of failed calls.
    static const char* LastError();
    static unsigned long FailureCount();

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

private:
    bool EnsureHandle(int width, int height);
    std::vector<cv::KeyPoint> scratch_kps_;      // results of a call land here first (capacity nfeatures + 4 * nlevels)
    std::vector<unsigned char> scratch_desc_;
    orbx_extractor* handle_ = nullptr;
    int cap_w_ = 0, cap_h_ = 0;
    bool materialise_ = false;
};

}  // namespace ORB_SLAM2

#endif
