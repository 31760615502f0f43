/*
 * ivfront.h -- flat C-ABI of the MI355X-native IV-SLAM visual front end (libivfront.so).
 *
 * The reference (ut-amrl/IV_SLAM) has no plugin/FFI layer for this path: its boundary is two C++
 * classes, ORB_SLAM2::ORBextractor (ORB/include/ORBextractor.h:51-126) and ORB_SLAM2::ORBmatcher
 * (ORB/include/ORBmatcher.h:37-108), plus Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932)
 * which reads ORBextractor::mvImagePyramid directly.  This header is the C-ABI a maintainer binds
 * BEHIND those class surfaces (see include/ivfront_orbslam.hpp and INTEGRATION.md); every entry
 * point names the reference interface it replaces.  ORB/ = introspective_ORB_SLAM/.
 *
 * Conventions: plain pointers and sizes only; caller-allocated outputs; every function returns an
 * int status (IVF_OK or a negative IVF_E_*), never throws; ivf_last_error() gives a thread-local
 * message.  One handle = one non-re-entrant instance (like one ORBextractor); different handles
 * may be used concurrently from different threads (ORB/src/Frame.cc:116-124 runs L/R on 2 threads).
 * All compute runs in hand-written HIP kernels on gfx950; there is NO CPU fallback: without a
 * usable GPU every compute entry point fails with IVF_E_NO_DEVICE.
 */
#ifndef IVFRONT_H
#define IVFRONT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVF_OK             0
#define IVF_E_INVALID     -1   /* bad argument */
#define IVF_E_CAPACITY    -2   /* caller buffer too small */
#define IVF_E_GEOMETRY    -3   /* cell grid leaves the image where the reference would throw / read OOB (with introspection: only without a cost map) */
#define IVF_E_NO_DEVICE   -4   /* no usable HIP device / HIP runtime error */
#define IVF_E_STATE       -5   /* call order violated (e.g. stereo match before extract) */

#define IVF_MAX_LEVELS 16

/* cv::KeyPoint subset produced by ORBextractor::operator() (ORB/src/ORBextractor.cc:1155-1156,1286-1294) */
typedef struct ivf_keypoint {
    float x, y;        /* pt, level-0 coordinates (level coords * mvScaleFactor[octave]) */
    float size;        /* (int)(31 * mvScaleFactor[octave]) */
    float angle;       /* degrees, [0,360) */
    float response;    /* FAST score, times the quality factor when introspection is active */
    int32_t octave;
} ivf_keypoint;

/* ORBextractor constructor arguments (ORB/include/ORBextractor.h:57-58, ORB/src/ORBextractor.cc:411-415) */
typedef struct ivf_extractor_params {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t enable_introspection;   /* bool enableIntrospection */
} ivf_extractor_params;

typedef struct ivf_extractor ivf_extractor;   /* one ORB_SLAM2::ORBextractor */
typedef struct ivf_frontend  ivf_frontend;    /* batched, device-resident stereo front end */

/* ---- library ---- */
int         ivf_version(void);
const char* ivf_last_error(void);
int         ivf_device_count(void);                 /* number of visible HIP devices (0 if none) */
long long   ivf_debug_launch_count(void);           /* measurement aid: kernel launches this process has issued through the library */
/* Build provenance: the first 16 hex digits of sha256 over iv_slam_amd/csrc/{*.hip sorted, ivf_device.h} and include/{* sorted},
 * taken when the library was linked (iv_slam_amd/csrc/Makefile).  iv_slam_amd/_lib.py recomputes it from the sources next to
 * the library and refuses to load a library built from anything else: a stale .so cannot produce a test result or a bench line.
 * r05: the hash also covers the build's variant flags, returned by ivf_build_flags() -- "" for the product, "-DIVF_EXPERIMENT" for the
 * experiment build (make EXPERIMENT=1 -> libivfront_exp.so: the kernel-variant selectors of the experiments in DESIGN.md exist only
 * there), any EXTRA=-D... of a one-off build: a differently compiled library cannot pass as the measured one. */
const char* ivf_build_id(void);
const char* ivf_build_flags(void);
/* Diagnostic: number of per-thread scratch slots (device + pinned host buffers of the per-call entry points) the process has
 * ever created.  Slots are leased by a thread and handed back when it exits, so the count follows the peak number of
 * CONCURRENT caller threads, not the number of threads ever started (ORB/src/Frame.cc:116-124 starts two per frame). */
int         ivf_debug_scratch_slots(void);

/* ---- ORBextractor ---- */
/* ORBextractor::ORBextractor (ORB/src/ORBextractor.cc:411-476); constructed in Tracking (ORB/src/Tracking.cc:174-191) */
int  ivf_extractor_create(const ivf_extractor_params* params, int device_id, ivf_extractor** out);
/* ~ORBextractor / Tracking::Release (ORB/src/Tracking.cc:2617-2626) */
void ivf_extractor_destroy(ivf_extractor* e);
/* GetLevels / GetScaleFactor (ORB/include/ORBextractor.h:69-73) */
int  ivf_extractor_get_levels(const ivf_extractor* e);
float ivf_extractor_get_scale_factor(const ivf_extractor* e);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (ORB/include/ORBextractor.h:75-89); each array has nlevels entries; any pointer may be NULL */
int  ivf_extractor_get_scale_tables(const ivf_extractor* e, float* scale, float* inv_scale,
                                    float* sigma2, float* inv_sigma2);
/* OpenCV-version switches for the three primitives whose arithmetic differs between OpenCV releases (the reference
 * accepts 2.4.3 ... 4.x, ORB/CMakeLists.txt:37-46, and pins none).  0 = OpenCV >= 3.4.2 / 4.x, the default everywhere.
 *   blur        1: cv::GaussianBlur 8U coefficients cvRound(k*256) = [18,34,49,55,49,34,18] (<= 3.4.1; saturating) (:1277)
 *   retain_best 1: KeyPointsFilter::retainBest calls nth_element(begin, begin + n, end) instead of begin + n - 1 (2.4 / 3.x) (:1146,:1164)
 *   atan2       1: cv::fastAtan2 = x*y/(x^2 + 0.28 y^2) in double (<= 2.4.3) instead of the degree-7 polynomial (:104)
 * Takes effect from the next extraction / run on the handle. */
int  ivf_extractor_set_opencv_variant(ivf_extractor* e, int blur, int retain_best, int atan2);
int  ivf_frontend_set_opencv_variant(ivf_frontend* fe, int blur, int retain_best, int atan2);
/* mnFeaturesPerLevel (ORB/src/ORBextractor.cc:438-452) and umax (:458-475), for tests */
int  ivf_extractor_get_feature_tables(const ivf_extractor* e, int32_t* features_per_level, int32_t* umax16);
/* ORBextractor::operator()(image, mask, keypoints, descriptors) (ORB/src/ORBextractor.cc:1224-1296).
 * image: 8-bit grey, host memory.  cost: the `mask` argument = u8 cost map of the same size, or NULL
 * (used only when the handle was created with enable_introspection, as in :1231-1238).
 * kps[cap], desc[cap*32] caller-allocated; *n_out = number written.  Empty image -> IVF_OK, *n_out = 0. */
int  ivf_extract(ivf_extractor* e, const uint8_t* image, int width, int height, int stride,
                 const uint8_t* cost, int cost_stride,
                 ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* mvImagePyramid[level] / mvQualityImagePyramid[level] (ORB/include/ORBextractor.h:91-92): copies the
 * un-padded level of the LAST ivf_extract into dst (rows of dst_stride bytes); dst may be NULL to query size. */
int  ivf_extractor_pyramid_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height);
int  ivf_extractor_quality_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height);
/* The 7x7 sigma-2 blurred copy of mvImagePyramid[level] that the descriptors of the LAST ivf_extract were sampled from
 * (the local `workingMat` of ORB/src/ORBextractor.cc:1276-1277; the reference does not keep it).  Diagnostic / parity
 * tests only; like the reference, levels without keypoints are not blurred (IVF_E_STATE for those). */
int  ivf_extractor_blur_level(const ivf_extractor* e, int level, uint8_t* dst, int dst_stride, int* width, int* height);
/* keypoints per level of the last call (allKeypoints[level].size(), ORB/src/ORBextractor.cc:1253-1254) */
int  ivf_extractor_level_counts(const ivf_extractor* e, int32_t* counts);

/* ---- Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932) ----
 * Uses the pyramids held by the two extractor handles from their last ivf_extract (as the reference
 * reads mpORBextractorLeft/Right->mvImagePyramid, :765,855,867,872).  bf = mbf, b = mb.
 * u_right[nL], depth[nL] receive mvuRight / mvDepth (-1 = no match). */
int  ivf_stereo_match(const ivf_extractor* left, const ivf_extractor* right,
                      const ivf_keypoint* kps_left, int n_left, const uint8_t* desc_left,
                      const ivf_keypoint* kps_right, int n_right, const uint8_t* desc_right,
                      float bf, float b, float* u_right, float* depth);

/* ---- ORBmatcher ---- */
/* ORBmatcher::DescriptorDistance (ORB/src/ORBmatcher.cc:1700-1716); host helper, 32-byte rows */
int  ivf_hamming(const uint8_t* a, const uint8_t* b);
/* dist[i] = DescriptorDistance(desc_a[pairs[2i]], desc_b[pairs[2i+1]]) on the device */
int  ivf_hamming_pairs(const uint8_t* desc_a, int n_a, const uint8_t* desc_b, int n_b,
                       const int32_t* pairs, int n_pairs, int32_t* dist, int device_id);

typedef struct ivf_bounds { float min_x, min_y, max_x, max_y; } ivf_bounds;   /* Frame::mnMinX.. (ORB/src/Frame.cc:724-756) */
/* Frame::GetFeaturesInArea on Frame::mGrid (ORB/src/Frame.cc:415-430, 615-680): host helper that
 * returns indices in the reference's cell-major order; *n_out may exceed cap (IVF_E_CAPACITY). */
int  ivf_features_in_area(const ivf_keypoint* kps, int n, const ivf_bounds* bounds,
                          float x, float y, float r, int min_level, int max_level,
                          int32_t* out, int cap, int* n_out);
/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono) (ORB/src/ORBmatcher.cc:1372-1518)
 * on flat, already-projected queries (one per last-frame map point that passed :1399-1421):
 *   q_u,q_v  projection (:1416-1417); q_ur = u - mbf*invzc (:1453); q_radius = th*mvScaleFactors[oct] (:1427);
 *   q_min_level,q_max_level = GetFeaturesInArea level args chosen at :1429-1434; q_angle = LastFrame.mvKeysUn[i].angle;
 *   q_desc = pMP->GetDescriptor() (32 B each); q_valid (nullable) 0 = skip; q_blocks (nullable, default 1) =
 *   pMP->Observations()>0.  cur_assign[n_cur] in/out: -1 free, -2 occupied by a blocking map point,
 *   >=0 index of the query matched (CurrentFrame.mvpMapPoints).  Window Hamming distances are computed on
 *   the device; the order-dependent greedy assignment (:1447-1472) and rotation histogram (:1475-1511) are
 *   replayed in the reference's order. */
int  ivf_search_by_projection(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                              const ivf_bounds* bounds,
                              int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                              const int32_t* q_min_level, const int32_t* q_max_level,
                              const float* q_angle, const uint8_t* q_desc,
                              const uint8_t* q_valid, const uint8_t* q_blocks,
                              int check_orientation, int32_t* cur_assign, int* nmatches, int device_id);

/* Same search; additionally cur_removed[n_cur] (nullable) = 1 where a keypoint was matched in this call and then dropped by
 * the rotation-consistency filter: the reference sets such an entry of mvpMapPoints to NULL (:1504) even if it held a map
 * point without observations before the call, which cur_assign == -1 alone cannot tell from "never touched". */
int  ivf_search_by_projection_ex(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                                 const ivf_bounds* bounds,
                                 int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                                 const int32_t* q_min_level, const int32_t* q_max_level,
                                 const float* q_angle, const uint8_t* q_desc,
                                 const uint8_t* q_valid, const uint8_t* q_blocks,
                                 int check_orientation, int32_t* cur_assign, uint8_t* cur_removed, int* nmatches, int device_id);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (ORB/src/ORBmatcher.cc:45-135; called
 * from Tracking::SearchLocalPoints, ORB/src/Tracking.cc:2124-2130) on flat queries, one per map point with
 * mbTrackInView && !isBad():  q_u,q_v = mTrackProjX/Y; q_ur = mTrackProjXR; q_radius = r * mvScaleFactors[level] with
 * r = RadiusByViewingCos(mTrackViewCos) (2.5 if viewCos > 0.998 else 4.0), times th when th != 1 (:63-70);
 * q_level = mnTrackScaleLevel (window levels [level-1, level]); nn_ratio = mfNNratio.  Best / second best with the
 * ratio test only when both are in the same octave (:117-121).  cur_assign / q_valid / q_blocks as above. */
int  ivf_search_map_points(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                           const ivf_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                           const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                           const uint8_t* q_valid, const uint8_t* q_blocks, float nn_ratio,
                           int32_t* cur_assign, int* nmatches, int device_id);

/* ORBmatcher::UpdateQualityScores(Frame &F) (ORB/src/ORBmatcher.cc:1108-1121; active only with
 * --ivslam_propagate_keyptqual, ORB/src/ORBmatcher.cc:130,1513,1647).  assign[i] = map-point index of keypoint i or -1
 * (F.mvpMapPoints); kp_quality = F.mvKeyQualScore (in/out); mp_quality = MapPoint quality scores (in/out).  Host-side
 * bookkeeping: sequential by definition (later keypoints see earlier updates). */
int  ivf_update_quality_scores(const int32_t* assign, int n, float* kp_quality, float* mp_quality, int n_map_points);

/* ---- testing hook ----
 * Runs the device's cv::KeyPointsFilter::retainBest core (the wave-cooperative replay of libstdc++ std::nth_element
 * used by the keypoint selection kernels) on caller-supplied non-negative responses: order_out[i] = index of the
 * element that ends at position i after nth_element(begin, begin+n_points-1, end, response-greater).  1 <= n <= 4096. */
int  ivf_test_retain_best(const float* responses, int n, int n_points, int32_t* order_out, int device_id);

/* ---- batched device-resident stereo front end (throughput path; one per GPU) ----
 * Equivalent to, for each pair: left/right ORBextractor::operator() (ORB/src/Frame.cc:115-125),
 * mvKeyQualScore (ORB/src/Frame.cc:130-143) and Frame::ComputeStereoMatches (:758-932), for up to
 * max_pairs pairs per call with all inputs/outputs resident in HBM. */
typedef struct ivf_frontend_config {
    ivf_extractor_params left, right;   /* Tracking builds the right extractor with introspection off (ORB/src/Tracking.cc:182-183) */
    int32_t width, height;              /* fixed image size */
    int32_t max_pairs;                  /* batch capacity */
    float   bf, b;                      /* Frame::mbf, Frame::mb */
    int32_t device_id;
} ivf_frontend_config;

/* Threads and streams (r05 / r06): a handle may be used from any thread, one call at a time; different handles may be used from different threads
 * at once.  But every front end of a process on ONE device runs its batches on the same three internal streams (a per-device pool, created once,
 * never destroyed -- six more streams for a second front end cost the first one a quarter of its throughput on this runtime): two front ends on
 * one device are executed IN TURN, not side by side, and they are coupled through those streams:
 *   - ivf_frontend_sync(A) also waits for what B has enqueued so far;
 *   - work a caller puts on ivf_frontend_batch_stream(A) (a collective, the tracker step) sits in front of B's next batches too -- a slow peer
 *     in A's all-gather stalls an unrelated front end, e.g. the second camera of a rig.  Keep such work short, or run it on a stream of your own
 *     behind ivf_frontend_pack_gather_block_of(..., your_stream) (see IVF_STREAM_OF_BATCH below);
 *   - ivf_frontend_destroy(A) waits for A's own batches only (their completion events); finish what you enqueued behind them on a lent
 *     stream before destroying the handle.
 * Front ends on different devices share nothing. */
int  ivf_frontend_create(const ivf_frontend_config* cfg, ivf_frontend** out);
void ivf_frontend_destroy(ivf_frontend* fe);
/* Enqueue one batch; asynchronous.  The batch is ordered after everything already enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream; it produced the inputs) and runs on one of the front end's three internal
 * streams, so consecutive batches overlap; `hip_stream` itself only waits until the inputs have been ingested
 * (the caller may overwrite them in stream order right after this call).  Results of a run stay valid until two
 * further runs have been enqueued.  Use ivf_frontend_sync / ivf_frontend_fetch / ivf_frontend_pack_gather_block to consume.
 * d_left/d_right: device pointers to n_pairs grey images, image i at base + i*image_stride, rows of row_stride bytes.
 * d_cost: device pointer to n_pairs u8 cost maps (same layout) or NULL. */
int  ivf_frontend_run(ivf_frontend* fe, const uint8_t* d_left, const uint8_t* d_right, const uint8_t* d_cost,
                      size_t image_stride, int row_stride, int n_pairs, void* hip_stream);
/* The same with the grey conversion in front of the extractor fused into the ingest (r06): Tracking::GrabImageStereo converts a 3-channel image with
 * cvtColor(mImGray, mImGray, CV_RGB2GRAY or CV_BGR2GRAY) by mbRGB = Camera.RGB (ORB/src/Tracking.cc:272-295; the right image likewise, :296-311).
 * A side is IVF_COLOR_GRAY (8UC1 as in ivf_frontend_run), IVF_COLOR_BGR (8UC3, bytes B,G,R: what cv::imread gives; CV_BGR2GRAY) or IVF_COLOR_RGB
 * (bytes R,G,B; CV_RGB2GRAY), each side with its own strides (bytes).  cv::cvtColor on 8-bit data is fixed point; the coefficients are those of
 * OpenCV 4.x -- (9798 R + 19235 G + 3735 B + 2^14) >> 15 -- like every other OpenCV primitive of this library; OR IVF_COLOR_CV3 into a code for
 * OpenCV <= 3.x's (4899 R + 9617 G + 1868 B + 2^13) >> 14.  A PCIe-fed pipeline with the introspection FCN ships the left COLOUR image once (the FCN
 * reads the same buffer, ivf_fcn_forward_device) instead of the colour image and a grey copy of it.  d_cost / its strides as in ivf_frontend_run.
 * stereo_kitti.cc:494-495 swaps R and B of the FCN's input IN PLACE; without rectification that is the very buffer GrabImageStereo converts afterwards
 * (SURVEY Appendix D-9): a host that reproduces the reference there hands over the swapped buffer, i.e. the other byte-order code. */
#define IVF_COLOR_GRAY 0
#define IVF_COLOR_BGR  1
#define IVF_COLOR_RGB  2
#define IVF_COLOR_CV3  4
int  ivf_frontend_run_color(ivf_frontend* fe, const uint8_t* d_left, int left_code, size_t left_image_stride, int left_row_stride,
                            const uint8_t* d_right, int right_code, size_t right_image_stride, int right_row_stride,
                            const uint8_t* d_cost, size_t cost_image_stride, int cost_row_stride, int n_pairs, void* hip_stream);
/* Cost maps without a copy (r06).  ivf_frontend_run ingests the cost maps like the images: a copy into the pitched level-0 plane of the batch context
 * (the reference has no such step: extractor and FCN share one cv::Mat, stereo_kitti.cc:508-521 -> Tracking.cc).  A producer on the device -- the FCN --
 * can write there directly: ivf_frontend_cost_plane returns the plane of the context the NEXT ivf_frontend_run / _run_color of this handle will use
 * (pair i's map at d_plane + i * image_stride, rows of row_stride bytes, only the first `width` bytes of a row may be written) and makes hip_stream wait
 * until that context's previous batch is done with it; pass exactly this pointer and these strides as d_cost / cost strides to ivf_frontend_run_color
 * and the ingest of the cost maps is skipped.  ivf_fcn_forward_device_strided writes such a plane. */
int  ivf_frontend_cost_plane(ivf_frontend* fe, uint8_t** d_plane, size_t* image_stride, int* row_stride, void* hip_stream);
/* Block until every ivf_frontend_run on this handle has finished (and, the internal streams being shared per device, whatever other front ends of
 * the process have enqueued on that device so far); reports device-side consistency errors. */
int  ivf_frontend_sync(ivf_frontend* fe);
/* Device-resident results of the last run (valid until the next run).  side 0 = left, 1 = right.
 * d_kps: [max_pairs][cap] ivf_keypoint, d_desc: [max_pairs][cap][32], d_count: [max_pairs] int32, cap = nfeatures;
 * d_uright/d_depth/d_quality: [max_pairs][cap] float (left only).  Any out pointer may be NULL. */
int  ivf_frontend_device_results(const ivf_frontend* fe, int side, const ivf_keypoint** d_kps, const uint8_t** d_desc,
                                 const int32_t** d_count, const float** d_uright, const float** d_depth,
                                 const float** d_quality, int* cap);
/* Copy one pair's results to host (synchronises).  uright/depth/quality are left-only and may be NULL. */
int  ivf_frontend_fetch(ivf_frontend* fe, int pair, int side, ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                        float* uright, float* depth, float* quality);
/* Same for the run `age` runs back (0 = the last one ... 2 = the oldest one still held): results of a run stay valid until
 * two further runs have been enqueued, so a caller that keeps three batches in flight reads each of them this way. */
int  ivf_frontend_fetch_of(ivf_frontend* fe, int age, int pair, int side, ivf_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                           float* uright, float* depth, float* quality);
/* Elapsed milliseconds of the dominant kernel (FAST score + NMS) in the last run, from HIP events
 * recorded on the run's stream around that launch (bench.py's roofline leg); <0 if unavailable. */
float ivf_frontend_last_fast_ms(ivf_frontend* fe);
/* Same measurement summed over the last `last_n` runs (<= 64 are kept; last_n < 1 = all kept):
 * *sum_ms = total milliseconds, *n_out = number of launches summed.  Synchronises on those events. */
int   ivf_frontend_fast_ms_stats(ivf_frontend* fe, int last_n, double* sum_ms, int* n_out);
/* Pack this rank's results of the last run for a descriptor all-gather: writes into d_block (device)
 * n_pairs fixed-size records {int32 n; int32 pad[3]; ivf_keypoint kps[cap]; uint8 desc[cap][32]; float uright[cap];
 * float depth[cap]} (the LEFT frame of each pair: mvKeys, mDescriptors, mvuRight, mvDepth -- what a rank needs to run the
 * tracker's cross-frame search against a frame extracted elsewhere, ivf_tracker_run below) and returns the record size in
 * *record_bytes (= ivf_track_record_bytes(cap)). */
int  ivf_frontend_pack_gather_block(ivf_frontend* fe, uint8_t* d_block, size_t block_bytes, size_t* record_bytes,
                                    void* hip_stream);
/* Same for the run `age` runs back (0 = the last one ... 2 = the oldest one still held).  Note that `hip_stream` waits
 * for that run to finish: pack on a stream of its own (bench.py does), not on the stream that feeds the next batch. */
int  ivf_frontend_pack_gather_block_of(ivf_frontend* fe, int age, uint8_t* d_block, size_t block_bytes, size_t* record_bytes,
                                       void* hip_stream);
/* hip_stream value for the two functions above: pack on the internal stream the batch itself ran on (in order behind it,
 * no cross-stream wait); ivf_frontend_batch_stream returns that stream (a hipStream_t) so that the consumer of the block
 * -- bench.py's all-gather -- can be enqueued behind the pack.
 * Ordering consequence (r04): the three internal streams are also LENT -- the 7x7 blur of the batch in context k runs on the internal
 * stream of context k + 1 (beside its own selection chain; IVF_NO_SIDE_BLUR=1 keeps it at home).  Work a caller puts on
 * batch_stream of batch n (a collective, the tracker step) therefore sits in front of the blur of batch n + 2 on that stream, and the
 * descriptors of batch n + 2 wait for that blur: a slow peer in the collective of batch n delays the extraction of batch n + 2 -- not
 * of batch n + 1.  Keep such work short (bench.py: pack + all-gather + tracker step, ~0.3 ms) or run it on a stream of your own behind
 * ivf_frontend_pack_gather_block_of(..., your_stream). */
#define IVF_STREAM_OF_BATCH ((void*)(intptr_t)-1)
void* ivf_frontend_batch_stream(ivf_frontend* fe, int age);

/* ---- batched, device-resident tracker step: the consumer of the all-gather -----------------------------------------------
 * For every (last, cur) pair of gather records, in ONE launch sequence and without leaving HBM: the matcher part of
 * Tracking::TrackWithMotionModel (ORB/src/Tracking.cc:1303-1330) =
 *   UpdateLastFrame's stereo points (Tracking.cc:1256-1300; Frame::UnprojectStereo, ORB/src/Frame.cc:958-972)
 *   -> ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono = false) (ORB/src/ORBmatcher.cc:1372-1518): projection,
 *      GetFeaturesInArea windows (ORB/src/Frame.cc:615-668), greedy assignment in last-keypoint order with the stereo check,
 *      rotation histogram + ComputeThreeMaxima (:1654-1695)
 *   -> the retry with th_retry when fewer than retry_below matches were found (Tracking.cc:1320-1330).
 * Keypoints are taken as undistorted (rectified stereo: mvKeysUn == mvKeys, Frame.cc:714 is a no-op there). */
typedef struct ivf_track_config {
    int32_t nfeatures;                       /* capacity of a gather record (<= 4096) */
    int32_t nlevels;
    float   scale_factors[IVF_MAX_LEVELS];   /* mvScaleFactors (ivf_extractor_get_scale_tables) */
    float   fx, fy, cx, cy, bf, b;           /* Frame::fx, fy, cx, cy, mbf, mb */
    ivf_bounds bounds;                       /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float   th;                              /* window factor: 7 for stereo, 15 otherwise (Tracking.cc:1313-1317), times the caller's multipliers */
    float   th_retry;                        /* 2 * th in the reference (Tracking.cc:1327) */
    int32_t retry_below;                     /* 20 (Tracking.cc:1320); 0 = never retry */
    int32_t check_orientation;               /* mbCheckOrientation of the matcher (true at Tracking.cc:1305) */
    float   th_depth;                        /* mThDepth: > 0 selects UpdateLastFrame's rule -- every stereo point closer than this and at
                                              * least the 100 closest, as new points WITHOUT observations (localization mode); <= 0: every
                                              * stereo point carries a point and points_block applies */
    int32_t points_block;                    /* default of pMP->Observations() > 0 (:1447-1449) when no flags are passed */
    int32_t max_pairs;                       /* (last, cur) pairs per call */
    int32_t device_id;
} ivf_track_config;
typedef struct ivf_tracker ivf_tracker;
size_t ivf_track_record_bytes(int nfeatures);
int  ivf_tracker_create(const ivf_track_config* cfg, ivf_tracker** out);
void ivf_tracker_destroy(ivf_tracker* t);
/* d_records: n_records gather records, record_bytes apart, 16-byte aligned (a packed block, or the all-gathered buffer);
 * d_pairs [n_pairs][2] int32 on the device = (last record, cur record) indices -- an index outside [0, n_records) gives that
 * pair d_nmatches = -1 and d_assign = -1 instead of a read outside the block;
 * d_poses [n_pairs][2][12] = for every PAIR {LastFrame.mTcw, CurrentFrame.mTcw} row-major 3x4 -- the optimised pose of the last
 * frame and the motion-model prior mVelocity * mLastFrame.mTcw of the current one (Tracking.cc:1311), so a frame that is "cur" in
 * one pair and "last" in the next carries a different pose in each, as in the reference -- or NULL = identity everywhere
 * (zero-motion prior);
 * d_point_flags [n_records][nfeatures] (nullable), per keypoint of a record when it is the LAST frame of a pair: bit 0 = the
 * keypoint carries a map point (0: it carries none, whatever its depth), bit 1 = that point has observations (it blocks the
 * current keypoint it takes, :1447-1449).  With flags th_depth is ignored; without them UpdateLastFrame's rule (th_depth > 0)
 * or "every stereo point" (th_depth <= 0) decides and points_block gives bit 1.
 * d_point_quality / d_key_quality [n_pairs][nfeatures] (both or neither; in/out): ORBmatcher::UpdateQualityScores(CurrentFrame)
 * (ORB/src/ORBmatcher.cc:1108-1121) as SearchByProjection runs it before returning under --ivslam_propagate_keyptqual
 * (:1513-1515), on the device: d_point_quality[p][i] = MapPoint::GetQualityScore() of the point last keypoint i carries,
 * d_key_quality[p][i2] = CurrentFrame.mvKeyQualScore[i2]; q = min of the two, the point's score is rewritten when it moves by
 * more than 0.01, the keypoint's always.  A retried pair is updated twice, like the two calls of Tracking.cc:1319-1328.
 * Outputs: d_assign [n_pairs][nfeatures] = for every keypoint of the current frame the index of the last-frame keypoint whose
 * point it received (CurrentFrame.mvpMapPoints) or -1; d_nmatches [n_pairs] = the return value.  Asynchronous on hip_stream.
 * The handle owns the scratch of ONE run: a run enqueued on a different stream than the previous one waits for it (an event);
 * use one tracker per stream to let runs overlap. */
int  ivf_tracker_run(ivf_tracker* t, const uint8_t* d_records, size_t record_bytes, int n_records, const int32_t* d_pairs, int n_pairs,
                     const float* d_poses, const uint8_t* d_point_flags, float* d_point_quality, float* d_key_quality,
                     int32_t* d_assign, int32_t* d_nmatches, void* hip_stream);

/* One local map point as Tracking::SearchLocalPoints (ORB/src/Tracking.cc:2088-2132) sees it.  80 bytes, 16-byte aligned arrays. */
typedef struct ivf_local_point {
    float   pos[3];                  /* MapPoint::GetWorldPos() */
    float   normal[3];               /* MapPoint::GetNormal() */
    float   min_distance;            /* mfMinDistance: the range test uses 0.8f * it (GetMinDistanceInvariance, MapPoint.cc:378-382) */
    float   max_distance;            /* mfMaxDistance: the range test uses 1.2f * it (:384-388), PredictScale the value itself (:412); > 0 */
    uint8_t desc[32];                /* GetDescriptor() */
    int32_t flags;                   /* bit 0: skip (isBad(), or mnLastFrameSeen == this frame: already matched, Tracking.cc:2114-2115);
                                      * bit 1: Observations() > 0 (a keypoint this point takes is skipped by later points, ORBmatcher.cc:87-89) */
    int32_t pad[3];
} ivf_local_point;
/* Batched Tracking::SearchLocalPoints, from the projection on: for frame slot f = record d_frames[f] with pose d_poses[f]
 * ([n_frames][12], row-major 3x4 Tcw after TrackWithMotionModel's optimisation; NULL = identity) and its local map points
 * d_points[d_point_offsets[f] .. d_point_offsets[f+1]) (at most max_points_per_frame are used): Frame::isInFrustum(pMP,
 * cos_limit = 0.5) (Frame.cc:557-613) incl. MapPoint::PredictScale (MapPoint.cc:407-422), then
 * ORBmatcher(nn_ratio).SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:45-135).  A record index outside [0, n_records)
 * or a decreasing / negative offset gives that frame d_nmatches = -1 and d_assign = -1.
 * d_occupied [n_frames][nfeatures] (nullable): 1 = the keypoint already holds a map point with observations (skipped, :87-89).
 * d_point_quality (indexed like d_points) / d_key_quality [n_frames][nfeatures] (both or neither; in/out): UpdateQualityScores(F)
 * (:128-132, :1108-1121) on the device for the keypoints that receive a point in THIS call; keypoints that held one before went
 * through the same update when they got it, and a second pass over them changes nothing (min(q_mp, q_kp) == q_kp by then).
 * Outputs: d_assign [n_frames][nfeatures] = index (within the frame's point range) of the map point each keypoint received in
 * THIS call or -1; d_nmatches [n_frames] = the return value (assignments made, replaced ones included).  n_frames <= max_pairs.
 * Asynchronous on hip_stream; shares the handle's scratch (and its one-call-at-a-time rule) with ivf_tracker_run. */
int  ivf_tracker_search_local(ivf_tracker* t, const uint8_t* d_records, size_t record_bytes, int n_records, const int32_t* d_frames,
                              int n_frames, const float* d_poses, const ivf_local_point* d_points, const int32_t* d_point_offsets,
                              int max_points_per_frame, const uint8_t* d_occupied, float th, float nn_ratio, float cos_limit,
                              float* d_point_quality, float* d_key_quality, int32_t* d_assign, int32_t* d_nmatches, void* hip_stream);

/* ---- introspection FCN forward (IF/networks/models_light/models_light.py:18-28; called at
 * ORB/Examples/Stereo/stereo_kitti.cc:231-247 (load) and :493-514 (pre-process, forward, u8 truncation)) ----
 * weights_blob: the model's state_dict f32 tensors in state_dict order, num_batches_tracked skipped
 * (tools/export_fcn_weights.py; 2,189,666 floats).  Input = BGR u8 image (what cv::imread returns), in_width x
 * in_height; output = cost map out_width x out_height (the reference sets out_size = image size because the
 * extractor builds the cost pyramid from the mask's own dimensions, ORB/src/ORBextractor.cc:1330-1331). */
typedef struct ivf_fcn ivf_fcn;
int  ivf_fcn_create(const float* weights_blob, size_t n_floats, int in_width, int in_height, int out_width, int out_height,
                    int max_batch, int device_id, ivf_fcn** out);
void ivf_fcn_destroy(ivf_fcn* f);
/* one image, host buffers: bgr rows of `stride` bytes; cost_u8 = (uint8)(cost*255) (truncation, :511), rows of
 * cost_stride bytes; cost_f32 (nullable) = the f32 map [out_height*out_width].  Either output may be NULL. */
int  ivf_fcn_forward(ivf_fcn* f, const uint8_t* bgr, int width, int height, int stride,
                     uint8_t* cost_u8, int cost_stride, float* cost_f32);
/* batch, device buffers: d_bgr = n interleaved BGR images (image i at base + i*image_stride, rows of row_stride
 * bytes); d_cost_u8 [n][out_h][out_w] and/or d_cost_f32 [n][out_h][out_w]; asynchronous on hip_stream. */
int  ivf_fcn_forward_device(ivf_fcn* f, const uint8_t* d_bgr, size_t image_stride, int row_stride, int n,
                            uint8_t* d_cost_u8, float* d_cost_f32, void* hip_stream);
/* the same with the u8 maps written through the caller's strides (r06): map i at d_cost_u8 + i * cost_image_stride, rows of cost_row_stride bytes
 * (>= out_width), e.g. the front end's own cost plane (ivf_frontend_cost_plane above) */
int  ivf_fcn_forward_device_strided(ivf_fcn* f, const uint8_t* d_bgr, size_t image_stride, int row_stride, int n,
                                    uint8_t* d_cost_u8, size_t cost_image_stride, int cost_row_stride, void* hip_stream);
/* Device-side flags of the handle (r05).  The convolutions run as split-f16 MFMA products (x = hi + lo in f16), exact to 22 bits
 * while |x| < 65504.  Weights are pre-scaled per output channel and every hidden tensor is bounded by ReLU6 (mobilenet.py:44-62); the
 * linear-bottleneck outputs are not, so the kernels that store them raise a flag when one reaches 65504 (libtorch's f32 convs,
 * stereo_kitti.cc:508, have no such limit: the reference would simply go on).  ivf_fcn_forward checks the flag itself and returns
 * IVF_E_STATE instead of a cost map; after ivf_fcn_forward_device call ivf_fcn_status(f, hip_stream): it waits for the stream, returns
 * IVF_E_STATE if any forward of this handle since the last check raised the flag, and clears it.  Non-finite weights are refused by
 * ivf_fcn_create (IVF_E_INVALID).  The kernels raise the flag in a word of the DEVICE and the last kernel of a forward moves it into the
 * handle: when two handles run forwards CONCURRENTLY on one device (different streams) handle A's overflow can be collected by handle B's
 * forward -- B then reports IVF_E_STATE and A reports IVF_OK for a wrong cost map.  The flag is never lost for the DEVICE, but it can be lost for
 * the handle that raised it: callers that overlap forwards of several handles on one device must treat IVF_E_STATE from ANY of them as
 * invalidating the forwards of ALL of them since their last checks (bench.py and the reference's one-network-per-process use a single handle). */
int  ivf_fcn_status(ivf_fcn* f, void* hip_stream);

/* ---- next rows of SURVEY section 8(f) ----
 * ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORB/src/ORBmatcher.cc:410-519):
 * kps1, desc1, kps2, desc2 = mvKeysUn and mDescriptors of the two frames, bounds2 = F2's (mnMinX, mnMinY, mnMaxX, mnMaxY);
 * prev_matched_xy [n1][2] in/out (vbPrevMatched), matches12 [n1] out (vnMatches12), *nmatches = return value.
 * nn_ratio / check_orientation = mfNNratio / mbCheckOrientation of the matcher. */
int  ivf_search_for_initialization(const ivf_keypoint* kps1, const uint8_t* desc1, int n1,
                                   const ivf_keypoint* kps2, const uint8_t* desc2, int n2, const ivf_bounds* bounds2,
                                   float* prev_matched_xy, int window_size, float nn_ratio, int check_orientation,
                                   int32_t* matches12, int* nmatches, int device_id);
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORB/src/ORBmatcher.cc:296-404) on projected
 * candidates: the caller runs :318-365 (Sim3 transform, depth / IsInImage / distance-invariance / viewing-angle tests,
 * PredictScale) and passes per surviving point u, v, radius = th * mvScaleFactors[level], level = nPredictedLevel and
 * the map point descriptor.  matched [n_kf] in/out: -1 = vpMatched[idx] is NULL, -2 = occupied on entry, on return
 * >= 0 = index of the query now matched there.  *nmatches = return value. */
int  ivf_search_keyframe_points(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, int n_kf, const ivf_bounds* bounds,
                                int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches, int device_id);
/* ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (ORB/src/ORBmatcher.cc:831-982): the matching core :893-955 on projected
 * map points (u, v, ur = u - bf*invz, radius, level as above); kf_uright = mvuRight, inv_level_sigma2 = mvInvLevelSigma2.
 * best_idx[i] = the keypoint to fuse query i with (-1: none within TH_LOW), best_dist[i] (nullable) its distance.  The
 * Replace / AddObservation bookkeeping (:958-977) stays with the caller, in query order.
 * inv_level_sigma2 == NULL selects the core of Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:983-1106), which
 * has no reprojection gate (kf_uright and q_ur may then be NULL as well). */
int  ivf_fuse_candidates(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, const float* kf_uright, int n_kf,
                         const ivf_bounds* bounds, const float* inv_level_sigma2, int n_levels,
                         int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                         const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid,
                         int32_t* best_idx, int32_t* best_dist, int device_id);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORB/src/ORBmatcher.cc:1145-1254).  The caller
 * projects KF1's map points that are still unmatched into KF2 (:1193-1230) -> per keypoint slot i1 of KF1: q12_valid,
 * u, v, radius = th * mvScaleFactors[level], level = nPredictedLevel, the map point descriptor; and KF2's into KF1
 * (:1273-1310) -> q21_*.  matches12[i1] = keypoint index in KF2 when both directions agree (:1336-1349), else -1;
 * *nfound = return value.  (vpMatches12[i1] = vpMapPoints2[matches12[i1]] is the caller's last step.) */
int  ivf_search_by_sim3(const ivf_keypoint* kps1, const uint8_t* desc1, int n1, const ivf_bounds* bounds1,
                        const ivf_keypoint* kps2, const uint8_t* desc2, int n2, const ivf_bounds* bounds2,
                        const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                        const uint8_t* q12_desc, const uint8_t* q12_valid,
                        const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                        const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound, int device_id);
/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vpMapPointMatches) (ORB/src/ORBmatcher.cc:165-294; used by
 * Tracking::TrackReferenceKeyFrame and Relocalization).  DBoW2::FeatureVector of each side in CSR form: *_node[k] the
 * node ids in std::map (ascending) order, *_idx[*_start[k] .. *_start[k+1]) the feature indices of node k.
 * kf_has_map_point[i] = vpMapPointsKF[i] && !isBad().  kf_kps = pKF->mvKeysUn, f_kps = F.mvKeys (angles only).
 * f_match[iF] = index i of the keyframe keypoint whose map point F's keypoint iF receives (vpMapPointMatches[iF] =
 * vpMapPointsKF[i]) or -1; *nmatches = return value. */
int  ivf_search_by_bow(const ivf_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_map_point, int n_kf,
                       const int32_t* kf_node, const int32_t* kf_start, const int32_t* kf_idx, int kf_nodes,
                       const ivf_keypoint* f_kps, const uint8_t* f_desc, int n_f,
                       const int32_t* f_node, const int32_t* f_start, const int32_t* f_idx, int f_nodes,
                       float nn_ratio, int check_orientation, int32_t* f_match, int* nmatches, int device_id);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORB/src/ORBmatcher.cc:528-661, loop closing):
 * same CSR feature vectors; has_map_point* = vpMapPoints*[i] && !isBad().  matches12[idx1] = idx2 (vpMatches12[idx1] =
 * vpMapPoints2[idx2]) or -1; acceptance is bestDist1 < TH_LOW (strict) here; *nmatches = return value. */
int  ivf_search_by_bow_keyframes(const ivf_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1, int n1,
                                 const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                 const ivf_keypoint* kps2, const uint8_t* desc2, const uint8_t* has_map_point2, int n2,
                                 const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                 float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches, int device_id);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORB/src/ORBmatcher.cc:663-829, local
 * mapping's CreateNewMapPoints) including CheckDistEpipolarLine (:146-163).  has_map_point* = GetMapPoint(i) != NULL,
 * stereo* = mvuRight[i] >= 0; F12 row-major 3x3 f32; (ex, ey) = epipole of camera 1 in image 2 (:670-676);
 * scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2.  matches12[idx1] = idx2 or -1 (vMatchedPairs =
 * the pairs with matches12 >= 0 in ascending idx1, :815-823); *nmatches = return value. */
int  ivf_search_for_triangulation(const ivf_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1, const uint8_t* stereo1, int n1,
                                  const int32_t* node1, const int32_t* start1, const int32_t* idx1, int nodes1,
                                  const ivf_keypoint* kps2, const uint8_t* desc2, const uint8_t* has_map_point2, const uint8_t* stereo2, int n2,
                                  const int32_t* node2, const int32_t* start2, const int32_t* idx2, int nodes2,
                                  const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int n_levels,
                                  int only_stereo, int check_orientation, int32_t* matches12, int* nmatches, int device_id);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)
 * (ORB/src/ORBmatcher.cc:1520-1652, Tracking::Relocalization) on the keyframe's projected map points (caller: :1543-1569):
 * per map point u, v, radius = th * mvScaleFactors[level], level = nPredictedLevel, q_angle = pKF->mvKeysUn[i].angle and
 * its descriptor.  cur_assign [n_cur] in/out: -1 = mvpMapPoints[i2] is NULL, -2 = occupied on entry, >= 0 = query index
 * matched there; orb_dist = ORBdist; *nmatches = return value.  (UpdateQualityScores at :1641: ivf_update_quality_scores.) */
int  ivf_search_by_projection_reloc(const ivf_keypoint* cur_kps, const uint8_t* cur_desc, int n_cur, const ivf_bounds* bounds,
                                    int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                    const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                    int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches, int device_id);
/* ---- DBoW2 vocabulary: Frame::ComputeBoW / KeyFrame::ComputeBoW (ORB/src/Frame.cc:683-694, KeyFrame.cc:66-77) =
 * mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1259).
 * The tree is handed over once as flat arrays (node 0 = root; children of node i = child[child_start[i] ..
 * child_start[i+1]); node_desc [n_nodes][32]; node_word / node_weight = Node::word_id / Node::weight, read at leaves;
 * depth_L = m_L) and stays on the device. */
typedef struct ivf_vocabulary ivf_vocabulary;
int  ivf_vocabulary_create(int n_nodes, const int32_t* child_start, const int32_t* child, const uint8_t* node_desc,
                           const int32_t* node_word, const double* node_weight, int depth_L, int device_id, ivf_vocabulary** out);
void ivf_vocabulary_destroy(ivf_vocabulary* v);
/* per descriptor: word id, weight and the node at level L - levelsup (:1217-1259) */
int  ivf_bow_transform(const ivf_vocabulary* v, const uint8_t* desc, int n, int levelsup,
                       int32_t* word_id, int32_t* node_id, double* weight);
/* mBowVec (TF-IDF, L1-normalised; word ids ascending) and mFeatVec (CSR, node ids ascending: the form ivf_search_by_bow*
 * and ivf_search_for_triangulation take) from those results; *_cap = capacities (n is always enough), *_n = sizes. */
int  ivf_bow_vectors(const int32_t* word_id, const int32_t* node_id, const double* weight, int n,
                     int32_t* bow_word, double* bow_value, int bow_cap, int* bow_n,
                     int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int fv_cap, int* fv_n);
/* MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312): desc = the n observed descriptors (rows of
 * vDescriptors, in mObservations order); *best_index = the row to copy into mDescriptor, *best_median (nullable) its median. */
int  ivf_distinctive_descriptor(const uint8_t* desc, int n, int* best_index, int* best_median, int device_id);

/* ---- device-resident frame -----------------------------------------------------------------------
 * Frame::AssignFeaturesToGrid (ORB/src/Frame.cc:415-430) once per frame on the device: keypoints (mvKeysUn), descriptors
 * and the 64x48 bucket grid (buckets in the reference's ix-major order, keypoints of a bucket in insertion order) stay
 * in HBM; searches against the frame then run GetFeaturesInArea (:615-668) and DescriptorDistance for every query on
 * the device and only replay the order-dependent assignment on the host. */
typedef struct ivf_frame ivf_frame;
int  ivf_frame_create(const ivf_keypoint* kps, const uint8_t* desc, const float* uright, int n, const ivf_bounds* bounds,
                      int device_id, ivf_frame** out);
void ivf_frame_destroy(ivf_frame* f);
/* number of keypoints of the frame (Frame::N) -- NOT the number that fell inside the grid */
int  ivf_frame_count(const ivf_frame* f);
/* the grid as built on the device: cell_start [64*48+1] (cell = ix*48 + iy), cell_index [n] */
int  ivf_frame_grid(const ivf_frame* f, int32_t* cell_start, int32_t* cell_index);
/* ivf_search_by_projection against the resident frame (same query arrays, same results) */
int  ivf_frame_search_by_projection(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                                    const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                                    const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                                    int check_orientation, int32_t* cur_assign, int* nmatches);
/* ivf_search_map_points (SearchByProjection(F, mapPoints), Tracking::SearchLocalPoints) against the resident frame */
int  ivf_frame_search_map_points(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                                 const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid, const uint8_t* q_blocks,
                                 float nn_ratio, int32_t* cur_assign, int* nmatches);

/* A resident frame made straight from a batch of the front end (run `age` back, pair, side 0 = left / 1 = right): keypoints,
 * descriptors and uRight are copied device -> device and the grid is built on the device; only the keypoint count is read
 * back.  The greedy replays' host mirror (angle, octave, uRight: 28 B per keypoint) is fetched by the first search. */
int  ivf_frame_create_from_frontend(ivf_frontend* fe, int age, int pair, int side, const ivf_bounds* bounds, ivf_frame** out);
/* The remaining window searches against a resident frame: same semantics as ivf_search_keyframe_points, ivf_fuse_candidates
 * (the frame's own uRight is the keyframe's mvuRight), ivf_search_by_sim3 (two resident keyframes) and
 * ivf_search_by_projection_reloc, with windows AND Hamming distances computed on the device (k_grid_window). */
int  ivf_frame_search_keyframe_points(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                                      const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches);
int  ivf_frame_fuse_candidates(ivf_frame* f, const float* inv_level_sigma2, int n_levels, int n_q, const float* q_u, const float* q_v,
                               const float* q_ur, const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                               const uint8_t* q_valid, int32_t* best_idx, int32_t* best_dist);
int  ivf_frame_search_by_sim3(ivf_frame* f1, ivf_frame* f2,
                              const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                              const uint8_t* q12_desc, const uint8_t* q12_valid,
                              const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                              const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound);
int  ivf_frame_search_by_projection_reloc(ivf_frame* f, int n_q, const float* q_u, const float* q_v, const float* q_radius,
                                          const int32_t* q_level, const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                          int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches);

/* ---- rectification in front of the extractor -------------------------------------------------
 * cv::initUndistortRectifyMap(K, D, R, P(0:3,0:3), size, CV_32F, map1, map2) as the driver calls it
 * (introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:285-343): host-side, double arithmetic, OpenCV 4.x plain C++
 * path.  K, R (NULL = identity), P: 3x3 row-major; dist = k1,k2,p1,p2[,k3[,k4,k5,k6[,s1..s4]]]
 * (n_dist 0, 4, 5, 8 or 12; tilt terms unsupported); map1/map2: [height][width] f32 (source x / source y). */
int  ivf_init_undistort_rectify_map(const double* K, const double* dist, int n_dist, const double* R, const double* P,
                                    int width, int height, float* map1, float* map2);
/* cv::remap(src, dst, map1, map2, cv::INTER_LINEAR) for 8-bit images, 1 or 3 interleaved channels, BORDER_CONSTANT 0
 * (stereo_kitti.cc:462-468 left/right image, :519-521 predicted cost image).  The maps are converted once
 * to the fixed-point form cv::remap derives per call and stay on the device; (width, height) = destination = map size. */
typedef struct ivf_remap ivf_remap;
int  ivf_remap_create(const float* map1, const float* map2, int width, int height, int src_width, int src_height,
                      int channels, int device_id, ivf_remap** out);
void ivf_remap_destroy(ivf_remap* r);
/* host buffers (drop-in for one cv::remap call; synchronous) */
int  ivf_remap_apply(ivf_remap* r, const uint8_t* src, int src_stride, uint8_t* dst, int dst_stride);
/* device buffers, n_images images image_stride bytes apart, on `hip_stream` (hipStream_t, NULL = default stream); asynchronous */
int  ivf_remap_apply_device(ivf_remap* r, const uint8_t* d_src, int src_stride, size_t src_image_stride, uint8_t* d_dst,
                      int dst_stride, size_t dst_image_stride, int n_images, void* hip_stream);
/* test aid: the fixed-point maps as cv::convertMaps would give them: xy [height][width][2] int16, alpha [height][width] (fy<<5|fx) */
int  ivf_remap_get_fixed_maps(const ivf_remap* r, int16_t* xy, uint16_t* alpha);

/* measurement aid (bench.py): HIP events bracket the network's most expensive launch (fused depthwise 3x3 + 1x1
 * projection 960 -> 160 of block 15, k_fcn_dwpw<5,4>) on the stream each forward runs on.  probe_stats returns the summed
 * duration of the last `last_n` probed forwards (0 = all kept, at most 64) and the batch size of the oldest of them. */
int  ivf_fcn_probe_enable(ivf_fcn* f);
/* r05: two probes -- 0 (default) = block 15 (k_fcn_irbd4<true>: launched twice per forward, blocks 15 and 16), 1 = block 17
 * (k_fcn_irbd4h: the largest single launch); probe_info / probe_stats report the selected one. */
int  ivf_fcn_probe_select(ivf_fcn* f, int which);
int  ivf_fcn_probe_stats(ivf_fcn* f, int last_n, double* sum_ms, int* n_out, int* batch);
/* which kernel the probe bracketed, as dispatched (e.g. "ivffcn::k_fcn_dwpw<5, 4> 960->160"), and its algorithmic HBM bytes
 * per image (hidden tensor read once + residual read + output written).  IVF_E_STATE before the first probed forward. */
int  ivf_fcn_probe_info(const ivf_fcn* f, char* name, int name_cap, double* algorithmic_bytes_per_image);

#ifdef __cplusplus
}
#endif
#endif /* IVFRONT_H */
