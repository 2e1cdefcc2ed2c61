// ORBVocabulary.h -- drop-in for the part of the reference's include/ORBVocabulary.h (= DBoW2::TemplatedVocabulary<FORB>,
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) the SLAM threads call: loadFromTextFile (:1339), transform of a descriptor set
// into BowVector + FeatureVector (:1127-1180; callers src/Frame.cc:649-659, KeyFrame::ComputeBoW), score (:1185-1190, L1),
// size/empty.  The tree lives in HBM; descents run in libmorb.so's k_bow_transform (include/orbv.h).  BowVector and
// FeatureVector keep DBoW2's container types (std::map), so KeyFrameDatabase / ORBmatcher read them unchanged.
#ifndef ORBVOCABULARY_H
#define ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>
#include "cv_compat.h"

struct orbv_vocabulary;

#ifdef MORB_USE_REFERENCE_TYPES
// inside the reference's tree the containers are DBoW2's own (Frame.h / KeyFrame.h include the same two headers)
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
// The reference's own headers name std::vector / std::list unqualified (include/Frame.h:104,240, include/KeyFrame.h:214,
// include/Map.h:62, include/KeyFrameDatabase.h:68) and get the using-directive from Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36
// through the include/ORBVocabulary.h this file takes the place of.  A replacement that did not pass it on would break them.
#include <list>
using namespace std;
#else
namespace DBoW2 {
typedef unsigned int WordId;   // BowVector.h:22
typedef double WordValue;      // BowVector.h:25
typedef unsigned int NodeId;   // BowVector.h:28
class BowVector : public std::map<WordId, WordValue> {};                          // BowVector.h:56
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};     // FeatureVector.h:20
}  // namespace DBoW2
#endif

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    ORBVocabulary() {}
    ~ORBVocabulary();
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;

    bool loadFromTextFile(const std::string& filename);
    // tree from arrays (tests; a text file of the stock k=10, L=6 vocabulary is 145 MB)
    bool create(int n_nodes, int L, const int* parent, const unsigned char* is_leaf, const unsigned char* desc, const double* weight);
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const;
    unsigned int size() const;   // number of words
    bool empty() const { return size() == 0; }

private:
    orbv_vocabulary* handle_ = nullptr;
};

}  // namespace ORB_SLAM2
#endif
