// ORBVocabulary.cc -- see ORBVocabulary.h.
#include "ORBVocabulary.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/orbv.h"

#include "resident.h"

namespace ORB_SLAM2 {

static int device() { return host_device(); }

ORBVocabulary::~ORBVocabulary() { orbv_destroy(handle_); }

bool ORBVocabulary::loadFromTextFile(const std::string& filename) {
    orbv_destroy(handle_); handle_ = nullptr;
    const int rc = orbv_load_text(filename.c_str(), device(), &handle_);
    if (rc) std::fprintf(stderr, "Vocabulary loading failure: %s\n", orb_last_error());   // the reference prints and returns false (:1362)
    return rc == 0;
}

bool ORBVocabulary::create(int n_nodes, int L, const int* parent, const unsigned char* is_leaf, const unsigned char* desc, const double* weight) {
    orbv_destroy(handle_); handle_ = nullptr;
    return orbv_create(n_nodes, L, parent, is_leaf, desc, weight, device(), &handle_) == 0;
}

unsigned int ORBVocabulary::size() const {
    int words = 0;
    if (handle_) orbv_info(handle_, nullptr, &words, nullptr, nullptr);
    return (unsigned int)words;
}

void ORBVocabulary::transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    v.clear(); fv.clear();
    if (empty()) return;                                 // :1137
    const int n = (int)features.size();
    if (n == 0) return;
    std::vector<unsigned char> flat((size_t)n * 32);
    for (int i = 0; i < n; ++i) std::memcpy(&flat[(size_t)i * 32], features[i].ptr(0), 32);
    std::vector<uint32_t> bid(n), fnode(n), fitems(n);
    std::vector<double> bval(n);
    std::vector<int32_t> fstart(n + 1);
    int nw = 0, nn = 0;
    const int rc = orbv_bow_vectors(handle_, flat.data(), n, levelsup, bid.data(), bval.data(), &nw, fnode.data(), fstart.data(), fitems.data(), &nn);
    if (rc) { std::fprintf(stderr, "ORBVocabulary::transform failed (%d): %s\n", rc, orb_last_error()); std::abort(); }
    for (int i = 0; i < nw; ++i) v.insert(v.end(), std::make_pair(bid[i], bval[i]));
    for (int k = 0; k < nn; ++k) {
        DBoW2::FeatureVector::iterator it = fv.insert(fv.end(), std::make_pair(fnode[k], std::vector<unsigned int>()));
        it->second.assign(fitems.begin() + fstart[k], fitems.begin() + fstart[k + 1]);
    }
}

double ORBVocabulary::score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
    std::vector<uint32_t> ia, ib; std::vector<double> va, vb;
    for (const auto& e : a) { ia.push_back(e.first); va.push_back(e.second); }
    for (const auto& e : b) { ib.push_back(e.first); vb.push_back(e.second); }
    return orbv_score_l1(ia.data(), va.data(), (int)ia.size(), ib.data(), vb.data(), (int)ib.size());
}

}  // namespace ORB_SLAM2
