/* orbv.h -- C ABI of the vocabulary-tree transform and the BoW-gated searches.
 *
 * Replaces:
 *   ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>)
 *     loadFromTextFile          reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1339-1425  -> orbv_load_text
 *     transform(feature, ...)   reference TemplatedVocabulary.h:1219-1260 (one descent: at every level the FIRST child
 *                               with the smallest Hamming distance, FORB.cpp:77-96)               -> orbv_transform[_device]
 *     transform(features, BowVector&, FeatureVector&, levelsup)
 *                               reference TemplatedVocabulary.h:1127-1180, BowVector.cpp:34-84,
 *                               FeatureVector.cpp:31-45 (TF_IDF weighting, L1 norm: what ORBvoc.txt declares; callers
 *                               src/Frame.cc:649-659, src/KeyFrame.cc ComputeBoW)                 -> orbv_bow_vectors
 *     score(BowVector, BowVector)  reference ScoringObject.cpp:23-68 (L1)                         -> orbv_score_l1
 *   int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)
 *                               reference include/ORBmatcher.h:64, src/ORBmatcher.cc:206-388      -> orbv_search_by_bow, mode 0
 *   int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)
 *                               reference include/ORBmatcher.h:65, src/ORBmatcher.cc:996-1165     -> orbv_search_by_bow, mode 1
 *   int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool, vector<bool>)
 *                               reference include/ORBmatcher.h:85-87, src/ORBmatcher.cc:1364-1786 -> orbv_search_for_triangulation
 *
 * No KeyFrame* / MapPoint* / std::map crosses the ABI: a DBoW2::FeatureVector travels as CSR arrays (node ids ascending, as
 * the std::map iterates; feature indices per node in push_back order), MapPoint validity as one flag byte per feature.
 * The vocabulary file itself is not part of the reference checkout (Vocabulary/ORBvoc.txt.tar.gz is a missing blob):
 * orbv_load_text reads the text format the reference's loader reads, orbv_create takes the same tree from arrays.
 *
 * A vocabulary handle and a search workspace each own one HIP stream and scratch buffers: the (large, read-only) tree may be
 * shared by threads only through orbv_transform_device; everything else is one handle per calling thread.
 */
#ifndef ORBV_H
#define ORBV_H
#include "orb_types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbv_vocabulary orbv_vocabulary;
typedef struct orbv_workspace orbv_workspace; /* stream + scratch of the BoW-gated searches (an ORBmatcher owns one) */

/* Tree from arrays, node ids as the reference's loader assigns them (TemplatedVocabulary.h:1377-1421): node 0 is the
 * root (its parent / descriptor / weight entries are ignored), parent[i] < i is not required but parent[i] must exist;
 * children are visited in ascending node id; word ids count the nodes flagged is_leaf in id order; a descent ends at the
 * first node without children.  desc: n_nodes x 32 bytes; weight: n_nodes doubles (idf of the words). */
int orbv_create(int n_nodes, int L, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                int device, orbv_vocabulary** out);
/* "k L scoring weighting" header line, then one line per node: parent is_leaf 32 descriptor bytes weight.  Only
 * scoring 0 (L1_NORM) with weighting 0 (TF_IDF) -- the ORBvoc.txt configuration -- is accepted.  A trailing empty line is
 * ignored (the reference's `while(!f.eof())` turns it into a node with an uninitialised descriptor). */
int orbv_load_text(const char* path, int device, orbv_vocabulary** out);
void orbv_destroy(orbv_vocabulary* v);
int orbv_info(const orbv_vocabulary* v, int* n_nodes, int* n_words, int* k, int* L);
void* orbv_stream(orbv_vocabulary* v);

/* One descent per feature.  features: n x 32 bytes on the host; word_id / node_id: n entries each (node_id = the node
 * `levelsup` levels above the leaves, 0 when L - levelsup <= 0).  weight may be NULL. */
int orbv_transform(orbv_vocabulary* v, const uint8_t* features, int n, int levelsup, uint32_t* word_id, uint32_t* node_id,
                   double* weight);
/* Enqueue-only form for resident descriptors (e.g. the block orbf_export_block hands out): device pointers, caller's stream. */
int orbv_transform_device(const orbv_vocabulary* v, const uint8_t* d_features, int n, int levelsup, uint32_t* d_word_id,
                          uint32_t* d_node_id, void* stream);

/* BowVector + FeatureVector of one frame: descents on the device, the two std::map constructions restated on flat
 * arrays on the host (double sums in feature order, L1 normalisation in ascending word order -- bit-identical doubles).
 * Outputs hold up to n entries (fv_start: n + 1).  *n_words / *n_fv_nodes receive the entry counts. */
int orbv_bow_vectors(orbv_vocabulary* v, const uint8_t* features, int n, int levelsup, uint32_t* bow_id, double* bow_val,
                     int* n_words, uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes);
double orbv_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2);

int orbv_workspace_create(int device, orbv_workspace** out);
void orbv_workspace_destroy(orbv_workspace* w);

/* One frame / keyframe as the BoW searches read it. */
typedef struct orbv_side {
    int n;                     /* features (N_total)                                                                  */
    const uint8_t* desc;       /* n x 32, global feature order (mDescriptors_total[cam].row(local))                  */
    const float* angle;        /* n: mvKeysUn_total[i].angle (mvKeys_total for a Frame: undistortion keeps the angle) */
    const uint8_t* flags;      /* n, or NULL = every feature usable and not stereo.  bit0 usable, bit1 stereo        */
    int n_nodes;               /* FeatureVector entries                                                               */
    const uint32_t* node_id;   /* n_nodes, strictly ascending                                                         */
    const int32_t* node_start; /* n_nodes + 1                                                                         */
    const uint32_t* items;     /* node_start[n_nodes] feature indices                                                 */
    const float* x;            /* triangulation only: mvKeysUn_total[i].pt                                            */
    const float* y;
    const int32_t* octave;     /* triangulation only                                                                  */
    const int32_t* cam_of;     /* triangulation only: keypoint_to_cam                                                 */
} orbv_side;

/* mode 0, SearchByBoW(pKF, F, vpMapPointMatches): a = pKF with flags bit0 = "has a MapPoint that is not bad", b = F;
 *         match[] has b->n entries: index of the keyframe feature whose MapPoint lands in vpMapPointMatches[i], or -1;
 *         accepted when best <= th_low and (float)best < nnratio * (float)second.
 * mode 1, SearchByBoW(pKF1, pKF2, vpMatches12): flags bit0 on both sides as above; match[] has a->n entries: feature of
 *         pKF2 whose MapPoint lands in vpMatches12[i], or -1; accepted when best < th_low (strict) and the ratio test.
 * Rotation-histogram filter when check_orientation != 0.  *nmatches = the reference's return value. */
int orbv_search_by_bow(orbv_workspace* w, const orbv_side* a, const orbv_side* b, int mode, int th_low, float nnratio,
                       int check_orientation, int32_t* match, int* nmatches);

enum { ORBV_MAX_CAMS = 8 };
typedef struct orbv_triangulation {
    int n_cams, n_levels;
    float F12[ORBV_MAX_CAMS][9];                /* F12s[cam], row-major (reference src/ORBmatcher.cc:1395-1399)        */
    float ex[ORBV_MAX_CAMS], ey[ORBV_MAX_CAMS]; /* epipole of pKF1's camera c in pKF2's camera c (:1407-1416)          */
    const float* scale_factors;                 /* pKF2->mvScaleFactors, n_levels                                      */
    const float* level_sigma2;                  /* pKF2->mvLevelSigma2, n_levels                                       */
} orbv_triangulation;

/* SearchForTriangulation from the fundamental matrices on.  flags bit0: the feature takes part (no MapPoint yet, its camera
 * enabled in vbCam, stereo when bOnlyStereo); bit1: mvuRight_total >= 0.  match[] has a->n entries (feature of pKF2 or -1):
 * vMatchedPairs is {(i, match[i]) : match[i] >= 0} in ascending i.  *nmatches = the reference's return value. */
int orbv_search_for_triangulation(orbv_workspace* w, const orbv_side* a, const orbv_side* b, const orbv_triangulation* t,
                                  int th_low, int check_orientation, int32_t* match, int* nmatches);

/* Resident form: a frame / keyframe is uploaded once (descriptors, angles, FeatureVector and -- when s->x is given -- the
 * triangulation arrays; the caller's arrays are free again on return) and searched any number of times.  The MapPoint state
 * changes between searches, so every search takes the two flag arrays of the moment (a->n / b->n bytes, meaning as above;
 * NULL = the flags uploaded with the keyframe, which may themselves be NULL = all usable). */
typedef struct orbv_keyframe orbv_keyframe;
int orbv_keyframe_create(orbv_workspace* w, const orbv_side* s, orbv_keyframe** out);
void orbv_keyframe_destroy(orbv_keyframe* k);
int orbv_keyframe_count(const orbv_keyframe* k);
/* A keyframe that never leaves HBM: descriptors, angles (and positions / octaves / right coordinates for triangulation)
 * are copied device-to-device -- e.g. from the frame orbf_export_features hands out --, the descents run on them and the
 * FeatureVector is built on the device (the nodes `levelsup` above the leaves must number <= 4096: 100 for the stock
 * k = 10, L = 6 vocabulary at levelsup = 4).  Flags start as bit0 = 1, bit1 = (uright >= 0).  after_stream: the stream the
 * source arrays are produced on (NULL: they are complete).  One synchronisation (the node count comes back to the host). */
typedef struct orbv_device_side {
    int n;
    const uint8_t* d_desc;     /* n x 32, 16-byte aligned                       */
    const float* d_angle;      /* n                                             */
    const float* d_x;          /* n, or NULL: no triangulation arrays           */
    const float* d_y;
    const int32_t* d_octave;
    const float* d_uright;     /* n, or NULL (= no feature is stereo)           */
    int n_cams;                /* cameras are contiguous index ranges:          */
    int cam_start[9];          /* camera c = [cam_start[c], cam_start[c+1])     */
} orbv_device_side;
int orbv_keyframe_from_device(orbv_workspace* w, const orbv_vocabulary* v, const orbv_device_side* s, int levelsup, void* after_stream,
                              orbv_keyframe** out);
/* Per-feature word / node ids of a device-built keyframe (for the BowVector, which stays a host std::map) and its
 * FeatureVector as CSR; any output pointer may be NULL.  fv_items needs fv_start. */
int orbv_keyframe_download(orbv_workspace* w, const orbv_keyframe* k, uint32_t* word_id, uint32_t* node_of_feature, uint32_t* fv_node,
                           int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes);

int orbv_search_by_bow_resident(orbv_workspace* w, const orbv_keyframe* a, const uint8_t* flags_a, const orbv_keyframe* b,
                                const uint8_t* flags_b, int mode, int th_low, float nnratio, int check_orientation, int32_t* match,
                                int* nmatches);
int orbv_search_for_triangulation_resident(orbv_workspace* w, const orbv_keyframe* a, const uint8_t* flags_a, const orbv_keyframe* b,
                                           const uint8_t* flags_b, const orbv_triangulation* t, int th_low, int check_orientation,
                                           int32_t* match, int* nmatches);

#ifdef __cplusplus
}
#endif
#endif
