// bow_oracle.cpp -- CPU restatement of the vocabulary-tree transform (DBoW2, vendored in the reference) and of the
// BoW-gated searches of ORBmatcher.
//
// THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as orb_oracle.cpp: only tests/, smoke() and the
// cpu_baseline leg of bench.py may load it).  PARITY STATUS: **parity unpinned** for the descent and the three searches --
// the reference ships no vocabulary file (Vocabulary/ORBvoc.txt.tar.gz is a missing blob), no tests and no fixtures, and
// those parts need OpenCV to build; **pinned against the reference's compiled code** for the BowVector / FeatureVector
// half of orc_bow_vectors (weight accumulation order, L1 normalisation, grouping of feature indices):
// oracle/_ref/libdbow2_ref.so = the reference's BowVector.cpp + FeatureVector.cpp built where they lie (make ref),
// tests/test_oracle_ref_dbow2.py, fixtures tests/golden/dbow2_ref_vectors.npz.  What is restated here is the
// reference's OWN code, literally, with std::map containers as the reference has them:
//   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1180 (transform of a feature set, TF_IDF + L1 as ORBvoc declares),
//   :1219-1260 (descent of one feature), :1339-1425 (text loader: node ids in file order, children in ascending id order),
//   BowVector.cpp:34-84, FeatureVector.cpp:31-45, ScoringObject.cpp:23-68 (L1 score),
//   src/ORBmatcher.cc:206-388 (SearchByBoW KF-Frame), :996-1165 (SearchByBoW KF-KF), :1364-1786 (SearchForTriangulation),
//   :167-184 (CheckDistEpipolarLine).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

extern "C" int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);  // ORBmatcher.cc:3994-4010 == FORB.cpp:77-96
extern "C" void orc_three_maxima(const int* sizes, int L, int* ind);

namespace {

struct VNode {  // TemplatedVocabulary::Node (:297-330)
    std::vector<unsigned> children;
    unsigned parent = 0, word_id = 0;
    double weight = 0;
    uint8_t desc[32];
    bool isLeaf() const { return children.empty(); }
};

struct Vocab {
    int L = 0;
    std::vector<VNode> nodes;
};

typedef std::map<unsigned, double> BowVector;                     // BowVector.h
typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;  // FeatureVector.h

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)  (:1219-1260)
void transform_one(const Vocab& V, const uint8_t* feature, unsigned& word_id, double& weight, unsigned* nid, int levelsup) {
    const int nid_level = V.L - levelsup;
    if (nid_level <= 0 && nid != nullptr) *nid = 0;  // root
    unsigned final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const std::vector<unsigned>& nodes = V.nodes[final_id].children;
        final_id = nodes[0];
        double best_d = orc_descriptor_distance(feature, V.nodes[final_id].desc);
        for (size_t k = 1; k < nodes.size(); ++k) {
            const unsigned id = nodes[k];
            const double d = orc_descriptor_distance(feature, V.nodes[id].desc);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid != nullptr && current_level == nid_level) *nid = final_id;
    } while (!V.nodes[final_id].isLeaf());
    word_id = V.nodes[final_id].word_id;
    weight = V.nodes[final_id].weight;
}

struct Side {  // one frame / keyframe as the BoW searches read it
    int n; const uint8_t* desc; const float* angle; const uint8_t* flags;
    int n_nodes; const uint32_t* node_id; const int32_t* node_start; const uint32_t* items;
    const float* x; const float* y; const int32_t* octave; const int32_t* cam_of;
};

FeatureVector to_map(const Side& s) {
    FeatureVector fv;
    for (int k = 0; k < s.n_nodes; ++k)
        fv[s.node_id[k]] = std::vector<unsigned>(s.items + s.node_start[k], s.items + s.node_start[k + 1]);
    return fv;
}

}  // namespace

extern "C" {

struct orc_bow_side {
    int n; const uint8_t* desc; const float* angle; const uint8_t* flags;
    int n_nodes; const uint32_t* node_id; const int32_t* node_start; const uint32_t* items;
    const float* x; const float* y; const int32_t* octave; const int32_t* cam_of;
};

struct orc_triangulation {
    int n_cams, n_levels;
    const float* F12;           // n_cams x 9, row-major (F12s[cam], ORBmatcher.cc:1395-1399)
    const float* ex; const float* ey;  // epipole of KF1's camera c in KF2's camera c (:1407-1416)
    const float* scale_factors; const float* level_sigma2;  // pKF2->mvScaleFactors / mvLevelSigma2
};

// Tree from flat arrays, node ids as the text loader assigns them (:1377-1421): node 0 = root; parent[i] for i >= 1;
// children in ascending id order; word ids in order of appearance of the is_leaf flags.
void* orc_vocab_create(int n_nodes, int L, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight) {
    Vocab* V = new Vocab();
    V->L = L;
    V->nodes.resize(n_nodes);
    unsigned words = 0;
    for (int i = 1; i < n_nodes; ++i) {
        VNode& nd = V->nodes[i];
        nd.parent = (unsigned)parent[i];
        V->nodes[parent[i]].children.push_back((unsigned)i);
        std::memcpy(nd.desc, desc + (size_t)i * 32, 32);
        nd.weight = weight[i];
        if (is_leaf[i]) nd.word_id = words++;
    }
    return V;
}
void orc_vocab_destroy(void* v) { delete (Vocab*)v; }

void orc_bow_transform(const void* v, const uint8_t* features, int n, int levelsup, uint32_t* word_id, uint32_t* node_id, double* weight) {
    const Vocab& V = *(const Vocab*)v;
    for (int i = 0; i < n; ++i) {
        unsigned w = 0, nid = 0; double wt = 0;
        transform_one(V, features + (size_t)i * 32, w, wt, &nid, levelsup);
        word_id[i] = w; node_id[i] = nid; weight[i] = wt;
    }
}

// transform(features, BowVector, FeatureVector, levelsup) for TF_IDF weighting + L1 norm (:1127-1180; ORBvoc.txt declares
// scoring 0 = L1_NORM, weighting 0 = TF_IDF), flattened: bow sorted by word id, feature vector as CSR sorted by node id.
// Returns the number of BoW entries; *n_fv_nodes the FeatureVector entries.  Output arrays hold up to n entries.
int orc_bow_vectors(const void* v, const uint8_t* features, int n, int levelsup, uint32_t* bow_id, double* bow_val,
                    uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes) {
    const Vocab& V = *(const Vocab*)v;
    BowVector bow; FeatureVector fv;
    for (int i = 0; i < n; ++i) {
        unsigned id = 0, nid = 0; double w = 0;
        transform_one(V, features + (size_t)i * 32, id, w, &nid, levelsup);
        if (w > 0) {                                   // not stopped
            BowVector::iterator vit = bow.lower_bound(id);  // BowVector::addWeight
            if (vit != bow.end() && !(bow.key_comp()(id, vit->first))) vit->second += w;
            else bow.insert(vit, BowVector::value_type(id, w));
            fv[nid].push_back((unsigned)i);            // FeatureVector::addFeature
        }
    }
    double norm = 0.0;                                 // BowVector::normalize(L1)
    for (auto& e : bow) norm += std::fabs(e.second);
    if (norm > 0.0) for (auto& e : bow) e.second /= norm;
    int k = 0;
    for (auto& e : bow) { bow_id[k] = e.first; bow_val[k] = e.second; ++k; }
    int off = 0, m = 0;
    for (auto& e : fv) {
        fv_node[m] = e.first; fv_start[m] = off;
        for (unsigned f : e.second) fv_items[off++] = f;
        ++m;
    }
    fv_start[m] = off;
    *n_fv_nodes = m;
    return k;
}

// L1Scoring::score (ScoringObject.cpp:23-68)
double orc_bow_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2) {
    BowVector a, b;
    for (int i = 0; i < n1; ++i) a[id1[i]] = v1[i];
    for (int i = 0; i < n2; ++i) b[id2[i]] = v2[i];
    BowVector::const_iterator i1 = a.begin(), i2 = b.begin();
    double score = 0;
    while (i1 != a.end() && i2 != b.end()) {
        const double vi = i1->second, wi = i2->second;
        if (i1->first == i2->first) { score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi); ++i1; ++i2; }
        else if (i1->first < i2->first) i1 = a.lower_bound(i2->first);
        else i2 = b.lower_bound(i1->first);
    }
    return -score / 2.0;
}

// SearchByBoW.  mode 0: (KeyFrame* pKF, Frame& F, vpMapPointMatches)  ORBmatcher.cc:206-388
//                       a = pKF (flags bit0: the feature has a good MapPoint), b = F; match[] has b.n entries:
//                       index of the KF feature whose MapPoint ends up in vpMapPointMatches[idxF], else -1.
//               mode 1: (KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12)  :996-1165
//                       flags bit0 on both sides: good MapPoint; match[] has a.n entries: idx2 or -1.
// Returns nmatches.
int orc_search_by_bow(const orc_bow_side* a_, const orc_bow_side* b_, int mode, int th_low, float nnratio, int check_ori, int32_t* match) {
    Side A, B; std::memcpy(&A, a_, sizeof(Side)); std::memcpy(&B, b_, sizeof(Side));
    const FeatureVector fa = to_map(A), fb = to_map(B);
    const int HISTO_LENGTH = 30;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    const int n_out = mode == 0 ? B.n : A.n;
    for (int i = 0; i < n_out; ++i) match[i] = -1;
    std::vector<bool> matched2(B.n, false);
    int nmatches = 0;
    FeatureVector::const_iterator ait = fa.begin(), bit = fb.begin();
    while (ait != fa.end() && bit != fb.end()) {
        if (ait->first == bit->first) {
            const std::vector<unsigned>& ia = ait->second; const std::vector<unsigned>& ib = bit->second;
            for (size_t k1 = 0; k1 < ia.size(); ++k1) {
                const unsigned idx1 = ia[k1];
                if (!(A.flags[idx1] & 1)) continue;                       // !pMP || pMP->isBad()
                const uint8_t* d1 = A.desc + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t k2 = 0; k2 < ib.size(); ++k2) {
                    const unsigned idx2 = ib[k2];
                    if (mode == 0) { if (match[idx2] >= 0) continue; }   // vpMapPointMatches[realIdxF]  :283
                    else { if (matched2[idx2] || !(B.flags[idx2] & 1)) continue; }  // :1077-1081
                    const int dist = orc_descriptor_distance(d1, B.desc + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                const bool under = mode == 0 ? bestDist1 <= th_low : bestDist1 < th_low;   // :324 vs :1107
                if (under && (float)bestDist1 < nnratio * (float)bestDist2) {
                    if (mode == 0) match[bestIdx2] = (int)idx1; else { match[idx1] = bestIdx2; matched2[bestIdx2] = true; }
                    if (check_ori) {
                        float rot = A.angle[idx1] - B.angle[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(mode == 0 ? bestIdx2 : (int)idx1);
                    }
                    nmatches++;
                }
            }
            ++ait; ++bit;
        } else if (ait->first < bit->first) ait = fa.lower_bound(bit->first);
        else bit = fb.lower_bound(ait->first);
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], ind[3];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, ind);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int j : rotHist[i]) { match[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// CheckDistEpipolarLine (ORBmatcher.cc:167-184)
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, int octave2, const float* F12, const float* sigma2) {
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2[octave2];
}

// SearchForTriangulation (ORBmatcher.cc:1364-1786) from the fundamental matrices on (their construction from the poses,
// :1375-1399, is the caller's).  flags bit0: usable (no MapPoint yet, camera enabled in vbCam, stereo if bOnlyStereo);
// bit1: stereo (mvuRight_total >= 0).  match[] has a.n entries (idx2 or -1); the reference's vMatchedPairs is the list of
// (i, match[i]) for match[i] >= 0 in ascending i.  Returns nmatches.
int orc_search_for_triangulation(const orc_bow_side* a_, const orc_bow_side* b_, const orc_triangulation* T, int th_low,
                                 int check_ori, int32_t* match) {
    Side A, B; std::memcpy(&A, a_, sizeof(Side)); std::memcpy(&B, b_, sizeof(Side));
    const FeatureVector fa = to_map(A), fb = to_map(B);
    const int HISTO_LENGTH = 30;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < A.n; ++i) match[i] = -1;
    int nmatches = 0;
    FeatureVector::const_iterator ait = fa.begin(), bit = fb.begin();
    while (ait != fa.end() && bit != fb.end()) {
        if (ait->first == bit->first) {
            for (unsigned idx1 : ait->second) {
                if (!(A.flags[idx1] & 1)) continue;
                const int cam1 = A.cam_of[idx1];
                const bool stereo1 = (A.flags[idx1] & 2) != 0;
                const uint8_t* d1 = A.desc + (size_t)idx1 * 32;
                int bestDist = th_low, bestIdx2 = -1;
                for (unsigned idx2 : bit->second) {
                    if (!(B.flags[idx2] & 1)) continue;          // vbMatched2 is never set in this overload (:1430, no write)
                    if (B.cam_of[idx2] != cam1) continue;        // :1562
                    const bool stereo2 = (B.flags[idx2] & 2) != 0;
                    const int dist = orc_descriptor_distance(d1, B.desc + (size_t)idx2 * 32);
                    if (dist > th_low || dist > bestDist) continue;
                    if (!stereo1 && !stereo2) {                  // :1582-1595
                        const float dex = T->ex[cam1] - B.x[idx2], dey = T->ey[cam1] - B.y[idx2];
                        if (dex * dex + dey * dey < 100 * T->scale_factors[B.octave[idx2]]) continue;
                    }
                    if (check_dist_epipolar_line(A.x[idx1], A.y[idx1], B.x[idx2], B.y[idx2], B.octave[idx2], T->F12 + 9 * cam1, T->level_sigma2)) {
                        bestIdx2 = (int)idx2; bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    match[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) {
                        float rot = A.angle[idx1] - B.angle[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            ++ait; ++bit;
        } else if (ait->first < bit->first) ait = fa.lower_bound(bit->first);
        else bit = fb.lower_bound(ait->first);
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], ind[3];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int)rotHist[i].size();
        orc_three_maxima(sizes, HISTO_LENGTH, ind);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (int j : rotHist[i]) { match[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // extern "C"
