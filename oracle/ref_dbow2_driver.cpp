// ref_dbow2_driver.cpp -- C entry points around the REFERENCE's own BowVector / FeatureVector classes.
//
// TEST INFRASTRUCTURE.  This file is ours; what it is linked with is not: `make -C oracle ref` compiles
// /root/reference/Thirdparty/DBoW2/DBoW2/BowVector.cpp and FeatureVector.cpp where they lie (they need nothing but the
// STL) into oracle/_ref/libdbow2_ref.so.  It is the one part of the reference that builds in this image -- everything
// else on the path needs OpenCV -- and it pins the BowVector / FeatureVector half of
// oracle/bow_oracle.cpp's orc_bow_vectors (accumulation order of the weights, L1 normalisation, grouping of the feature
// indices) against the reference's compiled code: tests/test_oracle_ref_dbow2.py, fixtures in tests/golden/.
// The loop below is TemplatedVocabulary::transform(features, v, fv, levelsup) for TF_IDF + L1
// (TemplatedVocabulary.h:1127-1180) from the point where the per-feature descent has produced (word id, weight, node id);
// the descent itself is a template over cv::Mat descriptors and cannot be built here.
#include <cstdint>
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" int ref_bow_build(const uint32_t* word_id, const double* weight, const uint32_t* node_id, int n, uint32_t* bow_id,
                             double* bow_val, uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes) {
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; ++i) {
        if (weight[i] > 0) {  // not stopped
            v.addWeight(word_id[i], weight[i]);
            fv.addFeature(node_id[i], (unsigned)i);
        }
    }
    v.normalize(DBoW2::L1);  // L1Scoring::mustNormalize -> true, L1
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_id[k] = it->first; bow_val[k] = it->second; }
    int off = 0, m = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++m) {
        fv_node[m] = it->first; fv_start[m] = off;
        for (size_t j = 0; j < it->second.size(); ++j) fv_items[off++] = it->second[j];
    }
    fv_start[m] = off;
    *n_fv_nodes = m;
    return k;
}

// BowVector::addIfNotExist (BowVector.cpp:50-58) and normalize(L2), for completeness of the pin
extern "C" int ref_bow_add_if_not_exist(const uint32_t* word_id, const double* weight, int n, int l2, uint32_t* bow_id, double* bow_val) {
    DBoW2::BowVector v;
    for (int i = 0; i < n; ++i) v.addIfNotExist(word_id[i], weight[i]);
    v.normalize(l2 ? DBoW2::L2 : DBoW2::L1);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_id[k] = it->first; bow_val[k] = it->second; }
    return k;
}
