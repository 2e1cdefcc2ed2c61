/* orbm.h -- C ABI of the MI355X ORB matcher (drop-in for the hot part of ORB_SLAM2::ORBmatcher).
 *
 * Replaces:
 *   static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)
 *        reference include/ORBmatcher.h:44, src/ORBmatcher.cc:3994-4010                -> orbm_descriptor_distance
 *   the exhaustive top-2 Hamming loops inside SearchByBoW / SearchForTriangulation
 *        reference src/ORBmatcher.cc:287-321, :1069-1104, :1533-1594                  -> orbm_hamming_top2[_device]
 *   (new, for cross-camera all-pairs work)                                            -> orbm_hamming_matrix[_device]
 *   int ORBmatcher::SearchByProjection(Frame&, const Frame&, float th, bool bMono, cv::Mat Calib)
 *        reference include/ORBmatcher.h:54-55, src/ORBmatcher.cc:3448-3641            -> orbm_search_by_projection
 *   int ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, float th)
 *        reference include/ORBmatcher.h:48, src/ORBmatcher.cc:62-149                  -> orbm_search_by_projection_points
 *   Frame::AssignFeaturesToGrid / GetFeaturesInArea(cam, ...)
 *        reference src/Frame.cc:348-395, :574-629                                     -> orbm_frame_create / orbm_features_in_area
 *   ORBmatcher::ComputeThreeMaxima   reference src/ORBmatcher.cc:3948-3989             -> orbm_three_maxima
 *
 * No Frame* / MapPoint* crosses the ABI: the C++ wrapper (multi_orb_slam_amd/host/ORBmatcher.h) packs flat
 * arrays.  The 3-D projection of map points stays on the host (it is cv::Mat float algebra in the reference,
 * src/ORBmatcher.cc:3513-3528); queries arrive already projected.
 *
 * A matcher handle owns one HIP stream and scratch; use one handle per thread (the reference constructs an
 * ORBmatcher on the stack per use and calls it from three threads).  No global mutable state.
 */
#ifndef ORBM_H
#define ORBM_H
#include "orb_types.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { ORBM_TH_HIGH = 100, ORBM_TH_LOW = 50, ORBM_HISTO_LENGTH = 30 }; /* reference src/ORBmatcher.cc:37-39 */
enum { ORBM_GRID_COLS = 64, ORBM_GRID_ROWS = 48 };                     /* reference include/Frame.h:37-38   */

typedef struct orbm_matcher orbm_matcher;
typedef struct orbm_frame orbm_frame;

int orbm_create(int device, orbm_matcher** out);
void orbm_destroy(orbm_matcher* m);
void* orbm_stream(const orbm_matcher* m);
/* Make the matcher issue all its work on a caller-owned hipStream_t (e.g. orbx_stream(ex), so frame building and
 * matching are ordered after the extractor's kernels without events); NULL restores the matcher's own stream. */
/* Orders this handle's stream behind everything enqueued so far on another HIP stream of the same device (e.g. the
 * stream a collective library ran an all-gather on) without blocking the host. */
int orbm_wait_for_stream(orbm_matcher* m, void* other_stream);
int orbm_set_stream(orbm_matcher* m, void* stream);

/* host helper, identical result to the reference's SWAR popcount; rows need 1-byte alignment only */
int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orbm_three_maxima(const int* bin_sizes, int L, int* ind3);
/* host: how many top-2 results pass SearchByBoW's acceptance (reference src/ORBmatcher.cc:324-327): best <= th_low and
 * (float)best < ratio * (float)second.  Returns the count (>= 0) or ORB_E_ARG. */
int orbm_count_ratio_accepted(const int32_t* best_dist, const int32_t* second_dist, int n, int th_low, float ratio);

/* Exhaustive top-2 per query over all nr references, strict '<' updates in reference order:
 * best_idx = lowest index attaining the minimum, second_dist = 2nd smallest WITH multiplicity,
 * (-1, 256, 256) when nothing is closer than 256.  Host pointers. */
int orbm_hamming_top2(orbm_matcher* m, const uint8_t* q, int nq, const uint8_t* r, int nr, int32_t* best_idx,
                      int32_t* best_dist, int32_t* second_dist);
/* Same on device pointers, asynchronous on `stream` (a hipStream_t; NULL = default stream).
 * d_scratch: at least orbm_top2_scratch_bytes(nq, nr) bytes of HBM (may be NULL when that returns 0). */
size_t orbm_top2_scratch_bytes(int nq, int nr);
int orbm_hamming_top2_device(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, int32_t* d_best_idx,
                             int32_t* d_best_dist, int32_t* d_second_dist, void* d_scratch, void* stream);

/* Full nq x nr distance matrix, uint16 row-major (the HBM-write-bound mode). */
int orbm_hamming_matrix(orbm_matcher* m, const uint8_t* q, int nq, const uint8_t* r, int nr, uint16_t* out);
int orbm_hamming_matrix_device(const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, uint16_t* d_out,
                               void* stream);

/* The two all-pairs entry points above compute their distances either with xor + popcount on the vector ALU or, from one
 * tile of work on (>= 64 queries, >= 64 references), as an int8 dot product of the +-1-expanded descriptors on the matrix
 * cores (256 - 2 * distance, exact): same results bit for bit, the second form about 1.4x / 2x faster at 32 000 x 32 000.
 * on = 1 / 0 selects the matrix-core / popcount kernels for the whole process, -1 restores the default (matrix cores unless
 * MORB_MATRIX_MFMA=0 / MORB_TOP2_MFMA=0 are set).  Returns the previous setting.  orbm_top2_scratch_bytes follows it. */
int orbm_use_matrix_cores(int on);
/* The matrix-core form of the top-2 searches (orbm_hamming_top2*, the camera-pair searches of orbm_cross_top2* / orbf_step) has two
 * arithmetic forms of its own: FP4 (the default: descriptor bits as +-4 in E2M1 on gfx950's v_mfma_f32_32x32x64_f8f6f4, twice the
 * int8 rate, the f32 result is the exact integer sort key) and int8 (v_mfma_i32_32x32x32_i8).  Same results bit for bit.
 * on = 1 / 0 selects FP4 / int8 for the whole process, -1 restores the default (FP4 unless MORB_TOP2_FP4=0).  Returns the
 * previous setting. */
int orbm_use_fp4_top2(int on);

/* -- projection-gated search ---------------------------------------------------------------------------- */
typedef struct orbm_frame_desc { /* flat view of the Frame members the matcher reads (src/Frame.cc:191-288) */
    int32_t n_total, n_cams;     /* N_total; global index g: cam-major concatenation (cam 2: g = N + i)      */
    const float* un_x;           /* mvKeysUn_total[g].pt.x                                                   */
    const float* un_y;
    const int32_t* octave;       /* mvKeysUn_total[g].octave                                                 */
    const float* angle;          /* mvKeysUn_total[g].angle                                                  */
    const float* uright;         /* mvuRight_total[g]                                                        */
    const int32_t* cam_of;       /* keypoint_to_cam[g]                                                       */
    const int32_t* local_of;     /* cont_idx_to_local_cam_idx[g]                                             */
    const uint8_t* const* desc;  /* mDescriptors_total[cam], N_cam x 32                                      */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY                                      */
} orbm_frame_desc;

typedef struct orbm_query { /* one projected map point */
    float u, v;             /* projection in the current frame                                               */
    float radius;           /* th * mvScaleFactors[octave]                      (src/ORBmatcher.cc:3543)     */
    float ur;               /* u - mbf*invzc  |  mTrackProjXR                   (:3573 | :113); NaN = no right-   */
                            /* coordinate gate (relocalisation :3809-3946 and loop :753-867 overloads have none) */
    int32_t min_level, max_level; /* as handed to GetFeaturesInArea             (:3547-3552 | :89)           */
    int32_t cam;
    int32_t blocks;         /* 1 if the MapPoint has Observations()>0: its claim hides the feature (:3566)   */
    float angle;            /* LastFrame.mvKeysUn_total[i].angle                (:3604)                      */
    uint8_t desc[32];       /* pMP->GetDescriptor()                                                          */
} orbm_query;

/* Host-only convenience for synthetic streams and tests: the last frame's features become projected map points under
 * a constant image-plane motion (du, dv): u = x + du, v = y + dv, radius = th * scale_factors[octave],
 * ur = u - mbf / depth (u where depth <= 0), levels octave-1 .. octave+1, blocks = 1, angle/desc copied.  A SLAM
 * caller computes the same fields from its 3-D map points instead (reference src/ORBmatcher.cc:3502-3552). */
int orbm_queries_from_motion(const orb_keypoint* kps, const uint8_t* desc, const float* depth, const int32_t* cam_of, int n,
                             float du, float dv, float th, const float* scale_factors, float mbf, orbm_query* out,
                             const float* un_x, const float* un_y); /* undistorted positions, or NULL, NULL: kps[i].x/y */

/* Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:673-778) = cv::undistortPoints(pts, K, dist,
 * noArray(), K) of OpenCV 2.4.x / 3.2, host side (the same operation sequence the kernels run).  calib == NULL or
 * k1 == 0: plain copy / (0, 0, cols, rows), as in the reference.  out4 = {minX, minY, maxX, maxY}. */
int orbm_undistort_points(const orb_calibration* calib, const float* x, const float* y, int n, float* ux, float* uy);
int orbm_image_bounds(const orb_calibration* calib, int cols, int rows, float* out4);
/* Calibration applied by the frames this handle assembles ON THE DEVICE from now on (orbm_frame_from_device): positions
 * are undistorted before grid assignment and uRight, the depth image is still read at the distorted pixel
 * (src/Frame.cc:968-981).  The caller passes matching bounds (orbm_image_bounds).  NULL switches it off. */
int orbm_set_calibration(orbm_matcher* m, const orb_calibration* calib);

/* Builds the 64x48 per-camera grid (round-to-cell insertion, ascending global indices) and uploads the frame. */
int orbm_frame_create(orbm_matcher* m, const orbm_frame_desc* f, orbm_frame** out);

/* orbm_frame_create for a frame whose descriptors are (partly) still in HBM: for every camera c with d_desc[c] != NULL the rows
 * N_c x 32 are read from that DEVICE pointer (16-byte aligned; e.g. orbx_device_descriptors of the extraction that produced
 * the frame -- the caller guarantees they hold what f->desc[c] holds and stay unchanged until the frame's first search has
 * returned); the other cameras' rows are taken from f->desc[c] as usual.  The 64x48 grid is built on the device (same
 * round-to-cell arithmetic), nothing but x, y, uright, angle, octave and one index word per feature crosses the bus.
 * d_desc == NULL: every camera from the host.  f->desc must be valid for every camera in any case: frames beyond 8192
 * features or 4 cameras, and MORB_RESIDENT_FRAMES=0, take orbm_frame_create. */
int orbm_frame_create_resident(orbm_matcher* m, const orbm_frame_desc* f, const uint8_t* const* d_desc, orbm_frame** out);

/* Frame assembly ON THE DEVICE from HBM-resident extractor outputs -- the merge of reference src/Frame.cc:191-239,
 * ComputeStereoFromRGBD (:959-986: depth lookup at (int)kp.pt.y,(int)kp.pt.x, uRight = x - mbf/d, -1 where d <= 0) and
 * AssignFeaturesToGrid (:348-395) -- with no host round trip.  Undistortion is the identity (k1 == 0, :676-680).
 * Everything is enqueued on the matcher's stream; the inputs must stay valid until that work has run. */
typedef struct orbm_cam_features {
    const orb_keypoint* d_kps; /* device, n keypoints (e.g. orbx_device_keypoints)                               */
    const uint8_t* d_desc;     /* device, n x 32                                                                 */
    int32_t n;
    const float* d_depth;      /* device depth image in metres (imDepth after convertTo), or NULL: uRight = -1   */
    int32_t depth_stride;      /* floats per row                                                                 */
} orbm_cam_features;
int orbm_frame_from_device(orbm_matcher* m, const orbm_cam_features* cams, int n_cams, float mbf, float min_x,
                           float min_y, float max_x, float max_y, orbm_frame** out);
/* Host copies of a frame's merged arrays (any pointer may be NULL): keypoints and descriptors in global (cam-major)
 * order, mvuRight_total, mvDepth_total.  Synchronises the matcher's stream. */
int orbm_frame_download(orbm_matcher* m, const orbm_frame* f, orb_keypoint* kps, uint8_t* desc, float* uright,
                        float* depth);
int orbm_frame_count(const orbm_frame* f);

/* Cross-camera exhaustive top-2 in one launch: every feature g of the frame against all features of the OTHER
 * cameras (reference analogue: the unrestricted inner loop of src/ORBmatcher.cc:287-321).  best_idx indexes the
 * concatenation of the other cameras' features in camera order; outputs have n_total entries (host pointers). */
int orbm_cross_top2(orbm_matcher* m, const orbm_frame* f, int32_t* best_idx, int32_t* best_dist,
                    int32_t* second_dist);
/* Same over a list of HBM-resident descriptor blocks, one per camera of the whole rig in global camera order (e.g. the
 * slices of an RCCL all-gather receive buffer): the cameras [first_query_block, +n_query_blocks) are the queries (the
 * ones this process owns), every other block is a reference.  Outputs hold sum(counts[query blocks]) entries. */
int orbm_cross_top2_blocks(orbm_matcher* m, const uint8_t* const* d_desc_blocks, const int* counts, int n_blocks,
                           int first_query_block, int n_query_blocks, int32_t* best_idx, int32_t* best_dist,
                           int32_t* second_dist);

/* Multi-GPU form of orbm_cross_top2_blocks: `d_gathered` is the result of ONE all-gather of every rank's export block
 * (orbf_export_block: cap_rows descriptor rows, the rank's cameras packed back to back, followed by a trailer of int32
 * per-camera counts), `world` blocks of `block_bytes` in rank order.  Queries = the features of rank `rank`, candidates =
 * every other camera of the rig; indices as in orbm_cross_top2 (position in the concatenation of the other cameras, global
 * camera order).  Nothing about the counts has to be known on the host beforehand: counts_out[world * cams_per_rank]
 * (may be NULL) and *nq_out are filled from the gathered trailers; the result arrays need room for cap_rows entries. */
int orbm_cross_top2_gathered(orbm_matcher* m, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows,
                             int cams_per_rank, int rank, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                             int32_t* counts_out, int* nq_out);
/* The same in two halves, for a caller that overlaps the exchange with an orbf_step in flight (orbf_step_begin reported
 * the export block ready): _enqueue puts repack + top-2 on this handle's side stream -- behind `after_stream`, the stream
 * the all-gather was enqueued on, when wait_after != 0 (NULL then means the default stream) -- and joins it into the
 * handle's main stream; after that stream has been synchronised (orbf_step_end does) _collect copies the results out. */
int orbm_cross_top2_gathered_enqueue(orbm_matcher* m, const uint8_t* d_gathered, int world, size_t block_bytes, int cap_rows,
                                     int cams_per_rank, int rank, void* after_stream, int wait_after);
/* (collect: the three result pointers may all be NULL when the caller reads the results in place through _views: pinned
 * host arrays of *nq_out entries, valid until the next cross-camera search of this matcher) */
int orbm_cross_top2_gathered_views(orbm_matcher* m, const int32_t** best_idx, const int32_t** best_dist, const int32_t** second_dist);
int orbm_cross_top2_gathered_collect(orbm_matcher* m, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                                     int32_t* counts_out, int* nq_out);
void orbm_frame_destroy(orbm_frame* f);
/* grid as CSR: cell = (cam*64 + ix)*48 + iy; cell_start has n_cams*3072+1 entries */
int orbm_frame_grid(const orbm_frame* f, int32_t* cell_start, int32_t* items);
/* GetFeaturesInArea(cam, x, y, r, minLevel, maxLevel) on the GPU; returns count in *n (may exceed cap) */
int orbm_features_in_area(orbm_matcher* m, const orbm_frame* f, int cam, float x, float y, float r,
                          int min_level, int max_level, int32_t* out, int cap, int* n);

/* Ordered candidate lists: for query i, cand_count[i] candidates in the reference's visiting order
 * (ix, iy, ascending index) that pass the window/level/right-coordinate gates, with their distances.
 * Lists are stored at [i*cap_per_query ...]; a count above cap_per_query => ORB_E_CAPACITY. */
int orbm_project_candidates(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, int cap_per_query,
                            int32_t* cand_idx, uint16_t* cand_dist, int32_t* cand_count);

/* (inspection and timing hooks -- orbm_debug_*: include/orb_debug.h) */

/* SearchByProjection(CurrentFrame, LastFrame, th, bMono, Calib) from the projected queries on.
 * occupied[g] != 0 where CurrentFrame.mvpMapPoints[g] already holds an observed point before the call (may be NULL:
 * TrackWithMotionModel clears the vector first, reference src/Tracking.cc:1254).
 * match_of_feature[g]: >= 0 = index of the query whose MapPoint ends in CurrentFrame.mvpMapPoints[g]; -1 = untouched;
 * -2 = set to NULL by the rotation-histogram filter (reference src/ORBmatcher.cc:3631).
 * *nmatches = the reference's return value. */
int orbm_search_by_projection(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                              const uint8_t* occupied, int th_high, int check_orientation,
                              int32_t* match_of_feature, int* nmatches);

/* The two-camera loop-closing search, SearchByProjection(KeyFrame*, Scw, vpPoints, vLoopMPCams, vpMatched, th, Calib)
 * (reference src/ORBmatcher.cc:566-750): every point is projected into BOTH cameras of the keyframe; the candidates of the
 * camera-2 window follow those of the camera-1 window (the reference's `for camidx` loop), ONE strict `<` chain runs over both,
 * and an accepted match (<= th_high = TH_LOW there) hides its feature from the later points (vpMatched[idx]).  q[i] holds
 * the first window (q[i].cam < 0: the point is not visible in that camera), second[i] the other one (cam < 0: none); the
 * descriptor, `blocks` and `angle` come from q[i].  Everything else as orbm_search_by_projection. */
typedef struct orbm_window {
    float u, v, radius;
    int32_t cam;                  /* < 0: no window */
    int32_t min_level, max_level; /* level gate of this window, as in orbm_query */
} orbm_window;
int orbm_search_by_projection_windows(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, const orbm_window* second,
                                      int nq, const uint8_t* occupied, int th_high, int check_orientation,
                                      int32_t* match_of_feature, int* nmatches);

/* The inner loop the remaining projection searches share: every projected point scans its window
 * -- same cell walk and level gate as above -- and reports the FIRST candidate in visiting order with the smallest
 * distance (`if(dist<bestDist)`), independently of every other point: no claims between queries.
 *   SearchBySim3 (reference src/ORBmatcher.cc:2814-3135, each direction): gate NONE, caller accepts <= TH_HIGH
 *   Fuse x2 (:1986-2509): gate CHI2 = the reprojection-error test of :2118-2143 (7.8 with a right coordinate, 5.99
 *     without; q.ur = projected right coordinate), levels nPredictedLevel-1 .. nPredictedLevel, caller accepts <= TH_LOW
 *     and then merges / replaces map points on the host as the reference does
 * occupied[g] != 0 hides feature g (may be NULL).  best_idx[i] = -1 / best_dist[i] = 256 when the window holds nothing. */
enum { ORBM_GATE_NONE = 0, ORBM_GATE_RIGHT = 1, ORBM_GATE_CHI2 = 2 };
int orbm_project_best(orbm_matcher* m, const orbm_frame* f, const orbm_query* q, int nq, const uint8_t* occupied, int gate,
                      const float* inv_level_sigma2, int n_levels, int32_t* best_idx, int32_t* best_dist);

/* SearchByProjection(F, vpMapPoints, th): camera-1 grid only, top-2 with level bookkeeping and nnratio.
 * occupied[g] != 0 where F.mvpMapPoints[g] already holds an observed point (may be NULL). */
int orbm_search_by_projection_points(orbm_matcher* m, const orbm_frame* cur, const orbm_query* q, int nq,
                                     const uint8_t* occupied, float nnratio, int th_high,
                                     int32_t* match_of_feature, int* nmatches);

#ifdef __cplusplus
}
#endif
#endif
