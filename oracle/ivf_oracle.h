/*
 * ivf_oracle.h -- CPU ORACLE for the IV-SLAM visual front end (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference's CPU algorithm for the hot path
 * (ORBextractor -> stereo matcher -> Hamming matcher).  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: "PARITY UNPINNED at the OpenCV boundary".  The reference's arithmetic for
 * FAST / resize / GaussianBlur / fastAtan2 / retainBest lives in un-vendored, version-unpinned
 * OpenCV (ORB/CMakeLists.txt:37-46) which is absent from this image, and the reference ships no
 * tests or golden vectors for this path.  Those primitives are therefore restated from OpenCV 4.x's
 * published plain-C++ algorithms (see DESIGN.md "Frozen primitive semantics").  What IS pinned:
 *   - std::nth_element / std::partition order  -> checked against this image's libstdc++ (oracle/stl_pin.cpp)
 *   - cosf / sinf                              -> checked exhaustively against this image's glibc 2.35
 *   - the introspection FCN                    -> golden vectors from the reference's own Python model
 *
 * ORB/ = /root/reference/introspective_ORB_SLAM/
 */
#ifndef IVF_ORACLE_H
#define IVF_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* cv::KeyPoint subset used by the path (ORB/src/ORBextractor.cc:1155-1156, 1286-1292) */
typedef struct {
    float x, y;      /* pt */
    float size;      /* 31 * scale[octave] truncated to int */
    float angle;     /* degrees [0,360) */
    float response;  /* FAST score (x quality factor when introspection is on) */
    int32_t octave;
} orc_keypoint;

typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int enable_introspection;
} orc_params;

typedef struct orc_extractor orc_extractor;

/* ---- frozen primitives (Appendix A of SURVEY.md) ---- */
int   orc_cv_round_f(float v);
int   orc_cv_round_d(double v);
const char* orc_build_id(void);   /* sha256[:16] of the checker's sources at build time (oracle/Makefile) */
/* OpenCV-version switches (process-wide; 0,0,0 = OpenCV >= 3.4.2 / 4.x, the default): see ivf_oracle.c */
void  orc_set_opencv_variant(int blur, int retain, int atan);
float orc_fast_atan2(float y, float x);
float orc_cosf(float x);   /* restated glibc >= 2.28 cosf (ARM optimized-routines algorithm) */
float orc_logf(float x);   /* restated glibc >= 2.27 logf (A-12): MapPoint::PredictScale */
float orc_sinf(float x);
/* FAST-9/16 corner score map of a (sub-)image: out[y*cols+x] = score if corner at `threshold` else 0,
 * zero outside rows [3,rows-3) x cols [3,cols-3). */
void  orc_fast_score_map(const uint8_t* img, int stride, int cols, int rows, int threshold, uint8_t* out);
/* cv::FAST(img, kps, threshold, nonmaxSuppression=true): returns number of keypoints written (<= cap). */
int   orc_fast_detect(const uint8_t* img, int stride, int cols, int rows, int threshold,
                      orc_keypoint* out, int cap);
void  orc_resize_linear_8u(const uint8_t* src, int sstride, int sw, int sh,
                           uint8_t* dst, int dstride, int dw, int dh);
/* KAT hook: destination indices whose bilinear (offset, coefficient) triple differs between OpenCV's 1. / ((double)d / s) and (double)s / d */
int   orc_resize_coef_mismatches(int ssize, int dsize);
void  orc_gauss7_8u(const uint8_t* src, int sstride, int w, int h, uint8_t* dst, int dstride);
/* libstdc++ std::nth_element(first, first+nth, first+n, response-greater) restated */
void  orc_nth_element_resp(orc_keypoint* v, int n, int nth);
/* cv::KeyPointsFilter::retainBest followed by the reference's resize(n_points); returns new size */
int   orc_retain_best(orc_keypoint* v, int n, int n_points);
int   orc_hamming256(const uint8_t* a, const uint8_t* b);
const int8_t* orc_bit_pattern_31(void);   /* 1024 int8: 256 x (x0,y0,x1,y1) */

/* ---- extractor (ORB/src/ORBextractor.cc:411-476, 880-1357) ---- */
orc_extractor* orc_extractor_create(const orc_params* p);
void  orc_extractor_destroy(orc_extractor* e);
int   orc_extractor_levels(const orc_extractor* e);
void  orc_extractor_tables(const orc_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                           int* features_per_level, int* umax16);
/* operator(): returns 0 on success; *n_out = number of keypoints (<= cap else error -2). */
int   orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride,
                  const uint8_t* cost, int cost_stride,
                  orc_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* mvImagePyramid[level] (un-padded view, contiguous) after the last orc_extract */
int   orc_pyramid_level(const orc_extractor* e, int level, const uint8_t** data, int* w, int* h);
int   orc_quality_level(const orc_extractor* e, int level, const uint8_t** data, int* w, int* h);
/* per-level keypoint count of the last call (debug / stage parity) */
int   orc_level_count(const orc_extractor* e, int level);

/* ---- Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932) ---- */
int   orc_stereo_match(const orc_extractor* eL, const orc_extractor* eR,
                       const orc_keypoint* kpL, int nL, const uint8_t* descL,
                       const orc_keypoint* kpR, int nR, const uint8_t* descR,
                       float bf, float b, float* u_right, float* depth);

/* ---- Frame grid (ORB/src/Frame.cc:415-430, 615-680) + ORBmatcher::SearchByProjection(cur,last)
 *      (ORB/src/ORBmatcher.cc:1372-1518) on flat, already-projected queries ---- */
typedef struct { float min_x, min_y, max_x, max_y; } orc_bounds;
int   orc_search_by_projection(const orc_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                               const orc_bounds* bounds,
                               int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                               const int32_t* q_min_level, const int32_t* q_max_level,
                               const float* q_angle, const uint8_t* q_desc,
                               const uint8_t* q_valid, const uint8_t* q_blocks,
                               int check_orientation, int32_t* cur_assign, int* nmatches);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORB/src/ORBmatcher.cc:45-135), flat queries */
int   orc_search_map_points(const orc_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                            const orc_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                            const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                            const uint8_t* q_valid, const uint8_t* q_blocks, float nn_ratio,
                            int32_t* cur_assign, int* nmatches);
/* Frame::GetFeaturesInArea on a freshly built grid: returns count, indices in reference order */
int   orc_features_in_area(const orc_keypoint* kps, int n, const orc_bounds* bounds,
                           float x, float y, float r, int min_level, int max_level, int32_t* out, int cap);
/* ORBmatcher::SearchForInitialization (ORB/src/ORBmatcher.cc:410-519): prev_xy [n1][2] in/out, matches12 [n1] out */
int   orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                    const orc_keypoint* k2, const uint8_t* d2, int n2, const orc_bounds* bounds2,
                                    float* prev_xy, int window_size, float nn_ratio, int check_orientation,
                                    int32_t* matches12, int* nmatches);
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORB/src/ORBmatcher.cc:296-404), flat queries */
int   orc_search_keyframe_points(const orc_keypoint* kf_kps, const uint8_t* kf_desc, int n_kf, const orc_bounds* bounds,
                                 int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                 const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches);
/* ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) matching core (ORB/src/ORBmatcher.cc:893-955), flat queries */
int   orc_fuse_candidates(const orc_keypoint* kf_kps, const uint8_t* kf_desc, const float* kf_uright, int n_kf,
                          const orc_bounds* bounds, const float* inv_level_sigma2,
                          int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                          const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid,
                          int32_t* best_idx, int32_t* best_dist);
/* ORBmatcher::SearchBySim3 (ORB/src/ORBmatcher.cc:1145-1254) on the two sets of projected map points */
int   orc_search_by_sim3(const orc_keypoint* k1, const uint8_t* d1, int n1, const orc_bounds* b1,
                         const orc_keypoint* k2, const uint8_t* d2, int n2, const orc_bounds* b2,
                         const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                         const uint8_t* q12_desc, const uint8_t* q12_valid,
                         const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                         const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORB/src/ORBmatcher.cc:165-294); feature vectors in CSR */
int   orc_search_by_bow(const orc_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_mp, int n_kf,
                        const int32_t* kf_node, const int32_t* kf_start, const int32_t* kf_idx, int kf_nodes,
                        const orc_keypoint* f_kps, const uint8_t* f_desc, int n_f,
                        const int32_t* f_node, const int32_t* f_start, const int32_t* f_idx, int f_nodes,
                        float nn_ratio, int check_orientation, int32_t* f_match, int* nmatches);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORB/src/ORBmatcher.cc:528-661) */
int   orc_search_by_bow_keyframes(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, int n1,
                                  const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                  const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, int n2,
                                  const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                  float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches);
/* ORBmatcher::SearchForTriangulation (ORB/src/ORBmatcher.cc:663-829) + CheckDistEpipolarLine (:146-163) */
int   orc_search_for_triangulation(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const uint8_t* stereo1, int n1,
                                   const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                   const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const uint8_t* stereo2, int n2,
                                   const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                   const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                   int only_stereo, int check_orientation, int32_t* matches12, int* nmatches);
/* ORBmatcher::SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (ORB/src/ORBmatcher.cc:1520-1652) */
int   orc_search_by_projection_reloc(const orc_keypoint* cur_kps, const uint8_t* cur_desc, int n_cur, const orc_bounds* bounds,
                                     int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                     const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                     int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches);
/* DBoW2 TemplatedVocabulary<FORB>::transform per descriptor (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259) */
int   orc_bow_transform(int n_nodes, const int32_t* child_start, const int32_t* child, const uint8_t* node_desc,
                        const int32_t* node_word, const double* node_weight, int depth_L,
                        const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight);
/* MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312): index of the least-median descriptor */
int   orc_distinctive_descriptor(const uint8_t* desc, int n, int* best_idx, int* best_median);
/* ORBmatcher::UpdateQualityScores(Frame&) (ORB/src/ORBmatcher.cc:1108-1121) */
void  orc_update_quality_scores(const int32_t* assign, int n, float* kp_quality, float* mp_quality);
/* ORBmatcher::ComputeThreeMaxima (ORB/src/ORBmatcher.cc:1654-1695) on bin sizes */
void  orc_three_maxima(const int* histo_sizes, int L, int* ind1, int* ind2, int* ind3);

/* cv::initUndistortRectifyMap, CV_32FC1 maps (stereo_kitti.cc:285-343 call sites; OpenCV calib3d, un-vendored: the
 * plain C++ path restated -- SURVEY Appendix A style frozen semantics).  K, R (NULL = identity), P = 3x3
 * row-major doubles; dist = k1,k2,p1,p2[,k3[,k4,k5,k6[,s1..s4]]] (n_dist 0/4/5/8/12).  Returns 0, -1 bad argument. */
int   orc_init_undistort_rectify_map(const double* K, const double* dist, int n_dist, const double* R, const double* P,
                                     int w, int h, float* map1, float* map2);
/* cv::remap(src, dst, map1, map2, INTER_LINEAR) with CV_32FC1 maps, 8-bit, cn channels, BORDER_CONSTANT 0
 * (stereo_kitti.cc:462-468, :519-521): 5-bit sub-pixel positions, 15-bit weight table, (sum + 2^14) >> 15 (A-10). */
void  orc_remap_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn, const float* map1, const float* map2,
                            int w, int h, uint8_t* dst, int dstride);
/* the 1024 x 4 fixed-point bilinear table exactly as OpenCV's initInterTab2D leaves it (A-10) */
void  orc_remap_weight_table(int16_t* tab4096);
/* cv::cvtColor 8UC3 -> 8UC1 (Tracking.cc:272-295): rgb 0 = CV_BGR2GRAY, 1 = CV_RGB2GRAY; cv3 0 = OpenCV 4.x's 15-bit coefficients, 1 = <= 3.x's 14-bit (A-13) */
void  orc_gray_from_color(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int rgb, int cv3);

#ifdef __cplusplus
}
#endif
#endif
