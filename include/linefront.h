/* linefront.h -- C ABI of liblinefront.so, the MI355X (gfx950) line front end and pairwise
 * motion solver that drops in behind LineSLAM's Node / GraphManager surface.
 *
 * Every entry point states the reference interface it replaces (paths relative to the
 * yan-lu/LineSLAM tree).  Plain pointers and sizes only; no C++/torch/OpenCV types.  Functions
 * return 0 (LF_OK) or a negative lf_status; nothing calls exit() (the reference does:
 * external/lsd/lsd.cpp:138-142, src/line/lineslam.cpp:272-275).
 *
 * The library has NO CPU compute path: every lf_* compute call runs hand-written HIP kernels and
 * fails with LF_ERR_NO_DEVICE when no gfx950 device is usable.
 *
 * Execution model: a context owns (or is given) ONE HIP stream.  The *_device entry points only enqueue
 * work on it and return (their small host arrays -- node ids, pair lists -- are copied through pinned
 * staging buffers; up to four pair-list calls may be pending per context before a call waits for the
 * oldest one's copy); the getters (lf_frame_get_*, lf_pair_get_*, lf_get_stage_ms) and the host-pointer
 * convenience calls synchronise that stream.  Several contexts on different streams can therefore be
 * driven from one host thread and overlap on the device (bench.py keeps four passes in flight that way).
 * A context is not thread-safe; use one per thread.
 */
#ifndef LINEFRONT_H
#define LINEFRONT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define LF_API
#else
#define LF_API __attribute__((visibility("default")))
#endif

typedef enum lf_status {
  LF_OK = 0,
  LF_ERR_INVALID = -1,      /* bad argument */
  LF_ERR_NO_DEVICE = -2,    /* no usable HIP device / kernel image */
  LF_ERR_HIP = -3,          /* a HIP runtime call failed (see lf_last_error) */
  LF_ERR_CAPACITY = -4,     /* a caller- or context-sized buffer was too small */
  LF_ERR_UNSUPPORTED = -5   /* a parameter value outside what this build supports */
} lf_status;

/* Mirror of the reference's global `SystemParameters sysPara` (src/line/lineslam.h:215-275,
 * filled by SystemParameters::init, src/line/lineslam.cpp:577-640) -- only the members the hot
 * path reads.  lf_params_init() stores the ParameterServer defaults
 * (src/parameter_server.cpp:162-198); lf_params_init_launch() additionally applies the overrides
 * of launch/lineslam.launch (lsd_angle_thres 40, min_matches 10). */
typedef struct lf_params {
  /* LSD (external/lsd/lsd.cpp:2070-2100) */
  double lsd_angle_th;            /* 22.5 */
  double lsd_density_th;          /* 0.7  */
  double lsd_scale;               /* 0.8  (lsd.cpp:2097) */
  double lsd_sigma_scale;         /* 0.6  */
  double lsd_quant;               /* 2.0  */
  double lsd_log_eps;             /* 0.0  */
  int    lsd_n_bins;              /* 1024 */
  double lsd_max_grad;            /* 255  */
  /* 2D / 3D line extraction (src/line/lineslam.cpp:200-357) */
  double line_segment_len_thresh; /* 10 px   line_2d_len_thres */
  double line3d_length_thresh;    /* 0.02 m  line_3d_len_thres_m */
  double ratio_of_collinear_pts;  /* 0.6     collin_pts_ratio */
  int    line_sample_max_num;     /* 100  (at most 103: larger values return LF_ERR_UNSUPPORTED) */
  int    line_sample_min_num;     /* 10  */
  double line_sample_interval;    /* 1   */
  int    line3d_mle_iter_num;     /* 100 */
  double pt2line_mahdist_extractline; /* 1.5 */
  int    ransac_iters_extract_line;   /* 100 */
  int    num_cells_lineseg_range;     /* 10  */
  double ratio_support_pts_on_line;   /* 0.7 */
  double stdev_sample_pt_imgline;     /* 3 px */
  double depth_stdev_coeff_c1;        /* 0.00273  */
  double depth_stdev_coeff_c2;        /* 0.00074  */
  double depth_stdev_coeff_c3;        /* -0.00058 */
  double msld_sample_interval;        /* 1 */
  double depth_scaling;               /* 1.0 (Node::Node passes 1.0, src/node.cpp:214) */
  /* pair solver (src/line/motion.cpp:605-849, src/node.cpp:1494-1694) */
  int    ransac_iters_line_motion;    /* 500 */
  int    adjacent_linematch_window;   /* 3   */
  int    line_match_number_weight;    /* 1   */
  int    min_feature_matches;         /* 20 (launch file: 10) */
  int    min_matches_loopclose;       /* 20  */
  double max_mah_dist_for_inliers;    /* 3   */
  double g2o_line_error_weight;       /* 1   */
  int    g2o_BA_use_kernel;           /* 1   (lineslam.cpp:629) */
  double g2o_BA_kernel_delta;         /* 10  (lineslam.cpp:630) */
  /* the reference draws from the unseeded global rand(); this library uses a counter-based
   * generator keyed by (seed, frame id, line id / pair id, draw index)               */
  uint64_t rng_seed;                  /* 0 */
  /* lines-only RANSAC computeRelativeMotion_Ransac (src/line/motion.cpp:367-526) */
  double pt2line3d_dist_relmotion;    /* 0.05 m  3d_pt2line_dst_relmot_m (parameter_server.cpp:188) */
  double line3d_angle_relmotion;      /* 10 deg  3d_line_angle_relmot_deg (:189) */
  /* the `algorithm` argument of Node::detect3DLines (src/line/lineslam.cpp:200-235) */
  int32_t line_detector;              /* LF_DETECTOR_LSD (0, default) or LF_DETECTOR_EDLINES (1) */
  int32_t reserved_;
} lf_params;
#define LF_DETECTOR_LSD 0
#define LF_DETECTOR_EDLINES 1

/* One 3D line of a frame: the flat, fixed-stride form of the reference's FrameLine /
 * RandomLine3d / RandomPoint3d objects (src/line/lineslam.h:41-151) -- 129 doubles + 2 ints =
 * 1040 bytes.  This is also the unit of the RCCL all-gather of keyframe line maps. */
typedef struct lf_line_record {
  double p[2], q[2];      /* FrameLine::p, q: image end points (pixels)                       */
  double lineEq2d[3];     /* FrameLine::lineEq2d (complineEq2d, lineslam.h:139-150)           */
  double r[2];            /* FrameLine::r: unit gradient direction (getGradient :527-537)     */
  double A[3], B[3];      /* line3d.A, line3d.B: 3D end points after MLE (metres, camera)     */
  double covA[9], covB[9];/* line3d.covA, covB (row-major 3x3)                                */
  double DUa[9], DUb[9];  /* line3d.rndA.DU, rndB.DU = diag(1/sqrt(w)) U^T (lineslam.h:73-77) */
  double Wsa[3], Wsb[3];  /* line3d.rndA.W_sqrt, rndB.W_sqrt                                  */
  double des[72];         /* FrameLine::des: MSLD descriptor (utils.cpp:1544-1610)            */
  int32_t lid;            /* FrameLine::lid: index inside the frame                           */
  int32_t seg;            /* index of the LSD segment (row of lf_lsd_get_segments) it came from */
} lf_line_record;

typedef struct lf_ctx lf_ctx;

/* Capacities of a context.  The reference's containers are unbounded std::vectors; the device buffers are not, so
 * every bound is a context parameter and exceeding one is REPORTED (LF_ERR_CAPACITY from the getter of the frame /
 * pair concerned, lf_pair_result::overflow), never silently truncated.  lf_caps_init() stores the defaults, which
 * are also the compiled maxima of this build (larger values: LF_ERR_UNSUPPORTED from lf_ctx_create_caps); smaller
 * values save device memory. */
typedef struct lf_caps {
  int32_t seg_cap;       /* 4096  LSD segments per frame (= candidates examined by the 3D-line stage)           */
  int32_t line_cap;      /* 512   3D lines (Node::lines) per frame; also the row stride of the all-gather payload */
  int32_t match_cap;     /* 256   line matches per pair handed to the pose solver                                */
  int32_t pt_match_cap;  /* 512   point matches per pair handed to the pose solver                               */
} lf_caps;
LF_API void lf_caps_init(lf_caps *caps);

LF_API void lf_params_init(lf_params *p);
LF_API void lf_params_init_launch(lf_params *p);
LF_API const char *lf_version(void);
LF_API const char *lf_status_str(int status);
LF_API const char *lf_last_error(const lf_ctx *ctx);   /* text of the last failing HIP call */

/* A context owns one HIP stream and all device buffers for batches of up to `max_batch` frames of
 * width x height pixels; it replaces the reference's process-wide globals (`sysPara`, `K`, nfa's
 * static table lsd.cpp:982).  One context per host thread; contexts are independent.
 * `hip_stream` may be NULL (the context creates its own) or an existing hipStream_t to launch on
 * (e.g. torch.cuda.current_stream().cuda_stream). */
LF_API int lf_ctx_create(lf_ctx **out, int device, void *hip_stream, int width, int height,
                         int max_batch, const lf_params *params);
LF_API int lf_ctx_create_caps(lf_ctx **out, int device, void *hip_stream, int width, int height,
                              int max_batch, const lf_params *params, const lf_caps *caps);
LF_API int lf_ctx_get_caps(const lf_ctx *ctx, lf_caps *caps);
LF_API void lf_ctx_destroy(lf_ctx *ctx);
LF_API int lf_ctx_set_params(lf_ctx *ctx, const lf_params *params);
LF_API int lf_ctx_synchronize(lf_ctx *ctx);

/* ---- a1-a8: LSD line-segment detection ------------------------------------------------------
 * Replaces  ntuple_list callLsd(IplImage*)  (src/line/utils.cpp:112-135) -> lsd() ->
 * LineSegmentDetection()  (external/lsd/lsd.cpp:1931-2065; interface external/lsd/lsd.h:210-243)
 * for a batch of frames.
 *
 * lf_lsd_batch_device: `d_gray` is a DEVICE pointer to n_frames 8-bit grey images
 * (frame f, row y at d_gray + f*frame_stride + y*row_stride).  Asynchronous on the context
 * stream; results stay in device memory until fetched. */
LF_API int lf_lsd_batch_device(lf_ctx *ctx, const uint8_t *d_gray, size_t frame_stride,
                               int row_stride, int n_frames);
/* Size of the scaled image in which regions are grown (lsd.cpp:549-550): N = floor(0.8 w) etc. */
LF_API int lf_lsd_dims(const lf_ctx *ctx, int *N, int *M);
/* Segments of frame `frame` of the last batch, rows {x1,y1,x2,y2,width} exactly as the reference's
 * ntuple_list (lsd.cpp:2037-2046).  *n_out = number found (if > cap: LF_ERR_CAPACITY, cap rows
 * written).  Synchronises the stream. */
LF_API int lf_lsd_get_segments(lf_ctx *ctx, int frame, double *segs, int cap, int *n_out);
/* Region label image (the integer line-pixel support, `image_int *region`, lsd.cpp:1989-1990,
 * 2050-2052): M*N values, 0 = no segment, k = k-th segment. */
LF_API int lf_lsd_get_labels(lf_ctx *ctx, int frame, uint16_t *labels);
/* Intermediate products for kernel-level parity tests.  which: 0 scaled image (double M*N),
 * 1 angles (double), 2 modgrad (double), 3 seed list (uint32 addresses y*N+x; *count = length),
 * 4 work counters (8 x uint64: region_grow calls, grow steps, rect_nfa calls, rect pixels,
 * region pixels, seeds, 0, 0). */
LF_API int lf_lsd_get_debug(lf_ctx *ctx, int frame, int which, void *out, size_t out_bytes,
                            int *count);
/* Host-buffer convenience with the signature shape of callLsd: one image in, segments out. */
LF_API int lf_lsd(lf_ctx *ctx, const uint8_t *gray, int row_stride, int width, int height,
                  double *segs, int cap, int *n_out, uint16_t *labels_or_null);

/* ---- EDLines -----------------------------------------------------------------------------
 * Replaces  LS* callEDLines(const cv::Mat& im_uchar, int* numLines)  (src/line/utils.cpp:1829-1853) ->
 * DetectLinesByED(srcImg, width, height, &noLines) of external/EDLines/libEDLines.a for a batch of frames.  The reference has
 * that detector as a BINARY only (x86-64 objects, not stripped); what runs here is the algorithm of its object code, restated
 * function by function from the disassembly (oracle/edlines_oracle.c): cvSmooth 5x5 with the fixed kernel 1 4 6 4 1 / 16,
 * ComputeGradientMapByLSD with threshold 11, anchors with threshold 3 joined strongest first through chain trees
 * (DoDetectEdgesByED), SplitSegment2Lines, JoinCollinearLines(6.0, 1.3), ValidateLineSegments.  On the reference's own example
 * (house.pgm -> LineSegments.txt) all 166 rows of the binary's output are reproduced at the file's 0.01 px resolution; the
 * restatement keeps two more short segments that are borderline in the a-contrario validation.  cvSmooth itself is OpenCV's
 * (absent here: restated from OpenCV 2.4's published algorithm, including the SSE2 column pass's rounding).  Results:
 * lf_lsd_get_segments (rows sx, sy, ex, ey, 0).  lf_detect3d_batch_device runs this detector instead of LSD when
 * lf_params::line_detector == LF_DETECTOR_EDLINES (the `algorithm == "EDLINES"` branch of Node::detect3DLines).  Asynchronous. */
LF_API int lf_edlines_batch_device(lf_ctx *ctx, const uint8_t *d_gray, size_t frame_stride, int row_stride, int n_frames);

/* ---- a9-a18: the rest of the per-frame front end --------------------------------------------
 * Replaces  void Node::detect3DLines(const cv::Mat& gray_uchar, const cv::Mat& depth_float,
 *                double line2d_len_thres, const cv::Mat& K, double ratio_of_collinear_pts,
 *                double line_3d_len_thres_m, double depth_scaling, std::string algorithm)
 * (src/node.h:286-287, src/line/lineslam.cpp:200-357); `algorithm` is lf_params::line_detector, the scalar
 * arguments live in lf_params (line_segment_len_thresh, ratio_of_collinear_pts,
 * line3d_length_thresh, depth_scaling).  The result is Node::lines (only lines with depth, lid =
 * index) as lf_line_record rows.
 *
 * lf_detect3d_batch_device: device-resident batch, asynchronous on the context stream; runs LSD and
 * the 3D-line stage.  `d_depth`: float32 metres, NaN or 0 = no measurement; strides in ELEMENTS.
 * K: 3x3 row-major camera matrix.  frame_ids (host array or NULL = 0..n-1) key the counter-based
 * random generator that replaces rand() in the RANSAC line fit (src/line/utils.h:49-60). */
LF_API int lf_detect3d_batch_device(lf_ctx *ctx, const uint8_t *d_gray, size_t gray_frame_stride,
                                    int gray_row_stride, const float *d_depth,
                                    size_t depth_frame_stride, int depth_row_stride, int n_frames,
                                    const double K[9], const uint64_t *frame_ids);
/* Node::lines of frame `frame` of the last batch (synchronises). */
LF_API int lf_frame_get_lines(lf_ctx *ctx, int frame, lf_line_record *out, int cap, int *n_out);
/* Per-LSD-segment diagnostics for stage-level parity tests: flags[i] = 0 dropped by the 2D length
 * filter, 1 no 3D line, 2 3D line; info[i][32] = A(3) B(3) covA(9) covB(9) numSmp #samples
 * #inliers levmar-iterations levmar-stop RANSAC-A(3). */
LF_API int lf_frame_get_candidates(lf_ctx *ctx, int frame, int32_t *flags, double *info, int cap,
                                   int *n_out);
/* Host-buffer convenience with the argument order of Node::detect3DLines. */
LF_API int lf_detect3d(lf_ctx *ctx, const uint8_t *gray, int gray_row_stride, const float *depth_m,
                       int depth_row_stride, int width, int height, const double K[9],
                       uint64_t frame_id, lf_line_record *out, int cap, int *n_out);

/* ---- a19-a25: line matching + pairwise motion ------------------------------------------------
 * Result of  MatchingResult Node::matchNodePair(const Node* older_node)  (src/node.h:107,
 * src/node.cpp:1494-1615) for one (newer, older) pair, flat form of MatchingResult
 * (src/matching_result.h:23-49) + LoadedEdge3D (src/edge.h:25-33). */
typedef struct lf_pair_result {
  float T[16];            /* final_trafo == ransac_trafo == edge.transform: newer -> older, row-major 4x4 */
  float rmse;             /* MatchingResult::rmse (1e9 when RANSAC could not start, motion.cpp:621-624)   */
  int32_t valid;          /* edge.id1 >= 0  <=>  found_transformation (node.cpp:1606-1607)                */
  int32_t n_matches;      /* all_line_matches.size()                                                      */
  int32_t n_inliers;      /* inlier_line_matches.size()                                                   */
  int32_t id_older;       /* edge.id1 */
  int32_t id_newer;       /* edge.id2 */
  int32_t ransac_best_iter;   /* diagnostics: iteration that produced the best minimal-sample model       */
  int32_t refine_rounds;      /* diagnostics: number of re-score + g2o rounds (motion.cpp:775-839)         */
  int32_t n_point_matches;    /* all_matches.size()    (0 unless the hybrid entry points are used)         */
  int32_t n_point_inliers;    /* inlier_matches.size()                                                     */
  double information_scale;   /* edge.informationMatrix = I6 * (n_pt_inl + n_ln_inl * weight) / rmse^2 (node.cpp:1533) */
  int32_t overflow;           /* 0, or a mask of LF_OVF_*: an input of this pair exceeded a context capacity and was cut (the
                                 getters of the pair then return LF_ERR_CAPACITY; the reference has no such bounds)        */
  int32_t reserved_;
} lf_pair_result;
#define LF_OVF_LINES 1        /* a node of the pair has more than line_cap lines                 */
#define LF_OVF_MATCHES 2      /* more than match_cap line matches                                */
#define LF_OVF_PT_MATCHES 4   /* more than pt_match_cap point matches                            */

/* For each pair i: newer = frame slot query_frames[i], older = train_frames[i] of the LAST
 * lf_detect3d_batch_device batch (node ids = the frame_ids given there).  Runs
 * Node::lineMatching(older, |id diff| <= adjacent_linematch_window) and, with the point-match list
 * empty, getTransform_PtsLines_ransac + getTransformFromHybridMatchesG2O.  Asynchronous.
 * query_frames / train_frames are HOST arrays. */
LF_API int lf_match_pairs_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames,
                                 int n_pairs);
LF_API int lf_pair_get_result(lf_ctx *ctx, int pair, lf_pair_result *out);
/* all_line_matches of pair `pair`: cv::DMatch {queryIdx, trainIdx, distance} (node.cpp:1681-1687). */
LF_API int lf_pair_get_matches(lf_ctx *ctx, int pair, int32_t *query_idx, int32_t *train_idx, double *dist,
                               int cap, int *n_out);
/* inlier_line_matches as indices into the match list. */
LF_API int lf_pair_get_inliers(lf_ctx *ctx, int pair, int32_t *match_idx, int cap, int *n_out);
/* descDiff matrix of pair `pair` (n_query x n_train doubles, 100 = gated out) for parity tests.  The matcher itself
 * never materialises it; this call evaluates it with a separate kernel for the pair list of the last launch. */
LF_API int lf_pair_get_descdiff(lf_ctx *ctx, int pair, double *D, size_t cap_doubles, int *n_query, int *n_train);

/* ---- the three operators of the pair path on their own ------------------------------------
 * unsigned Node::lineMatching(const Node* other, bool adjacentFrame, std::vector<cv::DMatch>* matches) const
 * (src/node.h:288, src/node.cpp:1619-1694) for a batch of pairs, WITHOUT the pose solve: query = frame slot
 * query_frames[i] (`this`), train = train_frames[i] (`other`) of the last batch -- or, if d_ext_recs is not NULL, slot
 * train_frames[i] of that external device-resident map (layout as lf_match_external_device; BASELINE.json config 4: one
 * query against 256 key frames in one launch).  adjacent: HOST array [n_pairs] of the adjacentFrame argument (0 / 1), or
 * NULL = derived from the node ids as Node::matchNodePair does (|id difference| <= adjacent_linematch_window,
 * node.cpp:1505-1507).  Results: lf_pair_get_matches.  Asynchronous. */
LF_API int lf_line_matching_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                   const uint8_t *adjacent, const lf_line_record *d_ext_recs,
                                   const int32_t *d_ext_nlines, const uint64_t *d_ext_ids, int ext_frames,
                                   int ext_line_cap);
/* The same for two nodes in HOST memory -- the exact shape of  unsigned Node::lineMatching(const Node* other, bool
 * adjacentFrame, std::vector<cv::DMatch>* matches) const : `query` = this, `train` = other; adjacent = 0 / 1, or -1 =
 * derived from the node ids.  Uploads both line maps into slots 0 / 1, matches, returns the list (cv::DMatch queryIdx,
 * trainIdx, distance; *n_out = its length, LF_ERR_CAPACITY if > cap).  Synchronous.  Needs max_batch >= 2. */
LF_API int lf_line_matching_node_pair(lf_ctx *ctx, const lf_line_record *query, int n_query, uint64_t id_query,
                                      const lf_line_record *train, int n_train, uint64_t id_train, int adjacent,
                                      int32_t *query_idx, int32_t *train_idx, double *dist, int cap, int *n_out);
/* bool getTransform_PtsLines_ransac(const Node* train, const Node* query, all_point_matches, all_line_matches,
 *      pt_inliers&, ln_inliers&, Eigen::Matrix4f& tf, float& rmse)  (src/line/utils.h:147-153, motion.cpp:605-849) for a
 * batch of pairs with CALLER-SUPPLIED match lists:
 *   lm_query / lm_train  HOST [n_pairs][lm_cap]: cv::DMatch::queryIdx / trainIdx of the line matches (indices into the
 *                        newer / older frame's Node::lines);  n_lm HOST [n_pairs]
 *   d_points .. K        the point side exactly as lf_match_pairs_hybrid_device takes it; d_points == NULL: no point
 *                        matches (pm_* and n_pm are then ignored)
 * A count above the context's match_cap / pt_match_cap is LF_ERR_CAPACITY (nothing is launched).  Results:
 * lf_pair_get_result (valid == the function's return value, T, rmse), lf_pair_get_inliers / _point_inliers (indices
 * into the lists given here).  Asynchronous. */
LF_API int lf_solve_pairs_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                 const int32_t *lm_query, const int32_t *lm_train, const int32_t *n_lm, int lm_cap,
                                 const float *d_points, int pt_cap, const int32_t *pm_query, const int32_t *pm_train,
                                 const int32_t *n_pm, int pm_cap, const double K[9]);
/* void getTransformFromHybridMatchesG2O(const Node* earlier, const Node* newer, pt_matches, ln_matches,
 *      Eigen::Matrix4f& transformation_estimate (in: guess, out: result), int iterations)
 * (src/transformation_estimation.h:21-26, .cpp:218-461) for two nodes in HOST memory: Levenberg-Marquardt over the older
 * camera's pose and one landmark per match, every match given is used.  T: row-major 4x4, newer -> older, in / out.
 * pts_*: [n][4] floats (x, y, z, 1) or NULL with n_pts_* = 0.  Synchronous.  Needs max_batch >= 2. */
LF_API int lf_refine_pair(lf_ctx *ctx, const lf_line_record *newer, int n_newer, const float *pts_newer, int n_pts_newer,
                          const lf_line_record *older, int n_older, const float *pts_older, int n_pts_older,
                          const int32_t *lm_query, const int32_t *lm_train, int n_lm, const int32_t *pm_query,
                          const int32_t *pm_train, int n_pm, const double K[9], float T[16], int iterations);
/* As lf_match_node_pair, for the two operators above with host-resident nodes: uploads both line maps (and point
 * arrays) into slots 0 / 1 and calls lf_solve_pairs_device for the one pair; the result is pair 0. */
LF_API int lf_solve_node_pair(lf_ctx *ctx, const lf_line_record *newer, int n_newer, uint64_t id_newer,
                              const float *pts_newer, int n_pts_newer, const lf_line_record *older, int n_older,
                              uint64_t id_older, const float *pts_older, int n_pts_older, const int32_t *lm_query,
                              const int32_t *lm_train, int n_lm, const int32_t *pm_query, const int32_t *pm_train,
                              int n_pm, const double K[9], lf_pair_result *out);
/* bool Node::getRelativeTransformationTo(const Node* earlier_node, std::vector<cv::DMatch>* initial_matches,
 *      Eigen::Matrix4f& resulting_transformation, float& rmse, std::vector<cv::DMatch>& matches) const
 * (src/node.h:124-128, src/node.cpp:1134-1338): the point-feature RANSAC that matchNodePair runs in builds WITHOUT USE_LINES,
 * for two nodes in HOST memory.  pts_*: feature_locations_3d_ ([n][4] floats, z = NaN without depth); match_*: the
 * members of initial_matches; min_matches / ransac_iterations / max_dist_for_inliers: the ParameterServer options of those
 * names (20 / 200 / 3; launch/lineslam.launch: min_matches 10).  Outputs: T (row-major, newer -> older), rmse, *found = the
 * return value, inlier_idx[0 .. *n_inliers) = indices into the caller's match arrays of `matches`, in the order the
 * reference keeps them (ascending distance).  g2o_refinement_iterations ("g2o_transformation_refinement", 0 by default): the
 * EdgeSE3PointXYZDepth refinement at the end of the reference function is not restated -- values > 0 return
 * LF_ERR_UNSUPPORTED.  rand() -> the library's counter generator (the reference seeds with clock()); ties of equal
 * distance keep the caller's order (std::sort leaves them unspecified).  At most 1024 matches, 4096 points per node.
 * Restated for the DEFAULT values of two more ParameterServer options, which are not parameters here: with
 * allow_features_without_depth = true the reference samples from a depth-filtered, sorted COPY of the matches while it scores
 * the unsorted initial_matches (node.cpp:1151-1170: the inlier order is the caller's), and with segment_to_optimize > 0
 * getTransformFromMatches weights by component [3] instead of z (src/transformation_estimation_euclidean.cpp:24-33); a host
 * that sets either must keep its CPU path.  Non-finite match distances are LF_ERR_INVALID (they have no rank).
 * Synchronous. */
LF_API int lf_relative_transformation_legacy(lf_ctx *ctx, const float *pts_newer, int n_pts_newer, uint64_t id_newer,
                                             const float *pts_older, int n_pts_older, uint64_t id_older,
                                             const int32_t *match_query, const int32_t *match_train, const float *match_dist,
                                             int n_matches, int min_matches, int ransac_iterations, double max_dist_for_inliers,
                                             int g2o_refinement_iterations, float T[16], float *rmse, int32_t *inlier_idx, int cap,
                                             int *n_inliers, int *found);
/* void MLEstimateLine3d(RandomLine3d& line, int maxIter) (src/line/utils.h, utils.cpp:980-1050) for n_lines lines in
 * HOST memory: line i has npts[i] support points pts[pt_offset[i] .. ) (xyz doubles, the RANSAC consensus set
 * `line.pts`, at most 104 each; their covariances are compPt3dCov(pt, K) as at lineslam.cpp:283-285) and the RANSAC
 * end points AB_init[i][6].  out[i] receives A, B, covA, covB, DUa, DUb, Wsa, Wsb (the other members are zeroed);
 * iters[i] (may be NULL) levmar's iteration count.  Synchronous; at most line_cap lines per call. */
LF_API int lf_mle_lines(lf_ctx *ctx, const double *pts, const int32_t *pt_offset, const int32_t *npts, int n_lines,
                        const double *AB_init, const double K[9], lf_line_record *out, int32_t *iters);

/* ---- points + lines (BASELINE.json config 3) ------------------------------------------------
 * As lf_match_pairs_device, but getTransform_PtsLines_ransac (motion.cpp:605-849) also receives point
 * matches, exactly as Node::matchNodePair hands it MatchingResult::all_matches (node.cpp:1519-1530):
 *   d_points   DEVICE [frames of the last batch][pt_cap][4] floats = Node::feature_locations_3d_
 *              (x, y, z, 1 as Node::projectTo3D builds them, node.cpp:952-1018; a NaN z is treated as errorFunction2 does); must stay valid
 *              until the call has completed on the context stream
 *   pm_query / pm_train   HOST [n_pairs][pm_cap]: cv::DMatch::queryIdx / trainIdx into the newer / older
 *              node's point array;  n_pm HOST [n_pairs] (each <= min(pm_cap, 512), else LF_ERR_CAPACITY)
 *   K          camera matrix (K[0] is the focal length of compPt3dCov, transformation_estimation.cpp:245)
 * Keypoint extraction and descriptor matching themselves (Node::featureMatching) stay with the caller. */
LF_API int lf_match_pairs_hybrid_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames,
                                        int n_pairs, const float *d_points, int pt_cap, const int32_t *pm_query,
                                        const int32_t *pm_train, const int32_t *n_pm, int pm_cap, const double K[9]);
/* inlier_matches as indices into the point match list given to lf_match_pairs_hybrid_device. */
LF_API int lf_pair_get_point_inliers(lf_ctx *ctx, int pair, int32_t *match_idx, int cap, int *n_out);
/* Node-level convenience (host memory): lf_match_node_pair with the two nodes' 3D points and their matches. */
LF_API int lf_match_node_pair_hybrid(lf_ctx *ctx, const lf_line_record *newer, int n_newer, uint64_t id_newer,
                                     const float *pts_newer, int n_pts_newer, const lf_line_record *older,
                                     int n_older, uint64_t id_older, const float *pts_older, int n_pts_older,
                                     const int32_t *pm_query, const int32_t *pm_train, int n_pm, const double K[9],
                                     lf_pair_result *out);

/* ---- lines-only RANSAC -------------------------------------------------
 * As lf_match_pairs_device, but the line matches go to computeRelativeMotion_Ransac
 * (src/line/motion.cpp:367-526: 3-line samples with the 5-degree degeneracy test, Euclidean consensus
 * test with pt2line3d_dist_relmotion / line3d_angle_relmotion, optimizeRelmotion = levmar on quaternion + t,
 * repeated while the consensus set grows) instead of getTransform_PtsLines_ransac.  Results:
 * lf_pair_get_result (T = (R|t) rounded to float, n_inliers = size of the returned consensus set, valid <=>
 * that set is not empty; rmse / information_scale are not defined by this solver and are 0),
 * lf_pair_get_inliers (the returned index vector), lf_pair_get_motion (Ro, to in double).       */
LF_API int lf_relmotion_pairs_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames,
                                     int n_pairs);
LF_API int lf_pair_get_motion(lf_ctx *ctx, int pair, double R[9], double t[3]);
/* The free function itself, vector<int> computeRelativeMotion_Ransac(vector<RandomLine3d> a, vector<RandomLine3d> b,
 * cv::Mat& Ro, cv::Mat& to) (src/line/utils.h:132): a[i] <-> b[i] already matched, HOST arrays, n <= 256;
 * x_b = Ro x_a + to.  Ro / to are written only when the returned set is not empty (as the reference).
 * id_a / id_b key the counter-based sample generator. */
LF_API int lf_relmotion_lines(lf_ctx *ctx, const lf_line_record *a, const lf_line_record *b, int n, uint64_t id_a,
                              uint64_t id_b, double R[9], double t[3], int32_t *inliers, int cap, int *n_inliers);

/* ---- raw TUM frames ------------------------------------------------------------------
 * The pixel conversions of OpenNIListener::loadRawData (src/openni_listener.cpp:1233-1246) and Node::Node
 * (src/node.cpp:193) for a batch of decoded frames: d_rgb [n][H][W][3] bytes R,G,B as stored in the PNG files,
 * d_depth16 [n][H][W] 16-bit depth -> the grey image LSD receives (the reference applies CV_RGB2GRAY to a BGR
 * matrix: blue is weighted as red) and depth in metres (value / depth_factor, 0 -> NaN).  The outputs are laid
 * out as lf_detect3d_batch_device expects them (dense, row stride W).  DEVICE pointers.  Asynchronous. */
LF_API int lf_ingest_tum_device(lf_ctx *ctx, const uint8_t *d_rgb, const uint16_t *d_depth16, int n_frames,
                                double depth_factor, uint8_t *d_gray_out, float *d_depth_out);

/* ---- point side feeding the hybrid solver (without the ORB extractor) ----------
 * Node::projectTo3D (src/node.cpp:952-1018): key points (cv::KeyPoint::pt, x then y) + depth image ->
 * feature_locations_3d_ (x, y, Z, 1).  Key points outside the image, NaN, or on a NaN depth are dropped (ordered
 * compaction; d_kept_out receives the indices of the survivors, may be NULL); at most max_keypoints points.
 * All pointers are DEVICE memory; d_depth as in lf_detect3d_batch_device.  Asynchronous. */
LF_API int lf_project_keypoints_device(lf_ctx *ctx, const float *d_depth, size_t depth_frame_stride,
                                       int depth_row_stride, int n_frames, const float *d_kp_xy,
                                       const int32_t *d_nkp, int kp_cap, const double K[9], double depth_scaling,
                                       int max_keypoints, float *d_points_out, int32_t *d_npts_out,
                                       int32_t *d_kept_out);
/* The point front end beside the line front end: Node::Node runs detect3DLines in a second thread while the key points are
 * detected and described (src/node.cpp:208-217, joined at :313-316).  enable != 0: lf_orb_extract_device and
 * lf_project_keypoints_device are enqueued on a second HIP stream owned by the context.  They start after everything that was
 * enqueued on the context's stream BEFORE the call (their inputs) and overlap what is enqueued after it -- so issue them
 * first, then lf_detect3d_batch_device.  They are joined stream-side (no host wait) in front of their first consumer on the
 * context's stream: lf_feature_match_pairs_device, the hybrid solvers, lf_orb_check / lf_orb_get_level, lf_ctx_synchronize --
 * or explicitly by lf_ctx_point_join before the caller enqueues work of its own that reads their outputs. */
LF_API int lf_ctx_point_stream(lf_ctx *ctx, int enable);
LF_API int lf_ctx_point_join(lf_ctx *ctx);

/* The ORB extractor: the ORB branch of Node::Node (src/node.cpp:222-290) for a batch of frames --
 *   detector->detect      = AorbFeatureDetector(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, fast_threshold)
 *                           (src/feature_adjuster.cpp:86-89 with the DetectorAdjuster's start threshold 20, src/features.cpp:75-76;
 *                           src/aorb.cpp:727-940: 8-level pyramid, FAST-9/16 + non-maximum suppression, border filter,
 *                           retainBest, Harris responses, intensity-centroid angle)
 *   removeDepthless + KeyPointsFilter::retainBest(max_keypoints) + resize           (node.cpp:101-125, 257-263)
 *   extractor->compute    = OrbDescriptorExtractor (src/features.cpp:197): GaussianBlur 7x7 sigma 2 per level, rBRIEF-256
 * d_gray / d_depth as lf_detect3d_batch_device (d_depth may be NULL: no depth filter).  Outputs (DEVICE, rows of kp_cap):
 * d_kp_xy [n][kp_cap][2] cv::KeyPoint::pt, d_kp_meta [n][kp_cap][4] (response, angle in degrees, octave, size; may be NULL),
 * d_desc [n][kp_cap][32], d_nkp [n]: exactly what lf_project_keypoints_device and lf_feature_match_pairs_device take.
 * max_keypoints <= 1024.  Asynchronous.  KeyPointsFilter::retainBest leaves its survivors in std::nth_element's order
 * (implementation-defined) and Node cuts that order at max_keypoints; here ties are ordered by detection order. */
LF_API int lf_orb_extract_device(lf_ctx *ctx, const uint8_t *d_gray, size_t gray_frame_stride, int gray_row_stride,
                                 const float *d_depth, size_t depth_frame_stride, int depth_row_stride, int n_frames,
                                 int fast_threshold, int max_keypoints, float *d_kp_xy, float *d_kp_meta, uint8_t *d_desc,
                                 int32_t *d_nkp, int kp_cap);
/* The same extractor behind the reference's threshold adapter (ParameterServer adjuster_max_iterations > 0, features.cpp:86-98):
 * VideoDynamicAdaptedFeatureDetector around DetectorAdjuster("AORB", 20) (src/feature_adjuster.cpp:107-186) -- the FAST threshold
 * is a STATE of the detector: x 0.7 (floor 2) when a detection returns fewer than min_features key points, the detection then
 * repeated at most max_iters times; x 1.3 (ceiling 10000) when it returns more than max_features; the next frame -- and the
 * next call -- starts with what the previous one left.  lf_orb_adjuster_init: the reference's values for `max_keypoints`
 * (min = max_keypoints, max = 1.5 max_keypoints).  reset_state != 0 (or the first call): start from adj->thresh; otherwise from
 * the state the context keeps on the device (lf_orb_adjuster_state reads it back: synchronises).  d_thresholds_out [n] (DEVICE,
 * may be NULL): the threshold of the detection whose key points were returned, per frame.  Frames are taken in order: the
 * thresholds are found by a sequential pass over per-frame score histograms, the detections run batched.  Asynchronous.
 * The grid wrapper (VideoGridAdaptedFeatureDetector, detector_grid_resolution > 1: per-cell pyramids) is not built. */
typedef struct lf_orb_adjuster {
  double thresh, min_thresh, max_thresh, increase_factor, decrease_factor;
  int32_t min_features, max_features, max_iters;
} lf_orb_adjuster;
LF_API void lf_orb_adjuster_init(lf_orb_adjuster *a, int max_keypoints, int max_iters);
LF_API int lf_orb_extract_adjusted_device(lf_ctx *ctx, const uint8_t *d_gray, size_t gray_frame_stride, int gray_row_stride,
                                          const float *d_depth, size_t depth_frame_stride, int depth_row_stride, int n_frames,
                                          const lf_orb_adjuster *adj, int reset_state, int max_keypoints, float *d_kp_xy,
                                          float *d_kp_meta, uint8_t *d_desc, int32_t *d_nkp, int kp_cap, int32_t *d_thresholds_out);
LF_API int lf_orb_adjuster_state(lf_ctx *ctx, double *thresh);
/* LF_ERR_CAPACITY if a frame of the last lf_orb_extract_device call had more corners than the extractor can rank
 * (16384 after non-maximum suppression) or more key points than kp_cap; LF_OK otherwise.  Synchronises. */
LF_API int lf_orb_check(lf_ctx *ctx);
/* Pyramid level / blurred level `level` (0..7) of frame `frame` of the last call, for parity tests: *w x *h bytes. */
LF_API int lf_orb_get_level(lf_ctx *ctx, int frame, int level, int blurred, uint8_t *out, size_t cap_bytes, int *w, int *h);

/* Node::featureMatching, BRUTEFORCE / ORB branch (src/node.cpp:606-641): for every pair (newer = query_frames[i],
 * older = train_frames[i], slots of the last batch -- their node ids key the random distance offset) the two
 * nearest Hamming neighbours of every query descriptor (256-bit ORB, 32 bytes), the ratio test
 * d1/d2 < nn_distance_ratio, the unique-train filter in query order.  d_desc [frames][desc_cap][32], d_ndesc
 * [frames]; outputs [n_pairs][desc_cap] (cv::DMatch queryIdx / trainIdx / distance) + [n_pairs] counts; all DEVICE
 * memory except the two HOST frame lists; desc_cap <= 1024.  Asynchronous. */
LF_API int lf_feature_match_pairs_device(lf_ctx *ctx, const uint8_t *d_desc, const int32_t *d_ndesc, int desc_cap,
                                         const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                         double nn_distance_ratio, int32_t *d_match_q, int32_t *d_match_t,
                                         float *d_match_dist, int32_t *d_nmatch);
/* unsigned int Node::featureMatching(const Node* other, std::vector<cv::DMatch>* matches) (src/node.cpp:568-641, the
 * BRUTEFORCE / ORB branch) for two HOST-resident nodes: desc_* = feature_descriptors_ rows (32 bytes each, at most 1024 per
 * node), newer = this (query), older = other (train).  Outputs cv::DMatch members; *n_out = matches->size();
 * LF_ERR_CAPACITY if cap is too small.  Overwrites the node ids of frame slots 0 and 1.  Synchronises. */
LF_API int lf_feature_match_node_pair(lf_ctx *ctx, const uint8_t *desc_newer, int n_newer, uint64_t id_newer,
                                      const uint8_t *desc_older, int n_older, uint64_t id_older, double nn_distance_ratio,
                                      int32_t *query_idx, int32_t *train_idx, float *dist, int cap, int *n_out);
/* lf_match_pairs_hybrid_device with the point matches already on the DEVICE (e.g. straight from
 * lf_feature_match_pairs_device): rows of pm_stride entries.  A count above pt_match_cap cannot be refused on the host
 * here: the solver then uses the first pt_match_cap matches, sets LF_OVF_PT_MATCHES in lf_pair_result::overflow, and the
 * getters of that pair return LF_ERR_CAPACITY (lf_pair_result::n_point_matches reports the full count). */
LF_API int lf_match_pairs_hybrid_device_pm(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_frames,
                                           int n_pairs, const float *d_points, int pt_cap, const int32_t *d_pm_query,
                                           const int32_t *d_pm_train, const int32_t *d_n_pm, int pm_stride,
                                           const double K[9]);

/* Stage durations (ms) of the last launches, measured with HIP events recorded on the context
 * stream: which = 0 LSD data-parallel kernels, 1 the LSD sweep kernel (k_lsd_sweep), 2 the 3D-line
 * stage, 3 the pair solver.  Synchronises. */
LF_API int lf_get_stage_ms(lf_ctx *ctx, int which, float *ms);

/* ---- multi-GPU: keyframe line maps ----------------------------------------------------------
 * Device pointers of the per-frame line maps of the last batch -- [max_batch][line_cap] records,
 * [max_batch] counts, [max_batch] node ids -- the fixed-stride payload of the RCCL all-gather that
 * gives every rank all keyframe line maps.  The pointers stay owned by the
 * context. */
LF_API int lf_get_device_records(lf_ctx *ctx, lf_line_record **d_recs, int32_t **d_nlines, uint64_t **d_ids,
                                 int *line_cap);
/* As lf_match_pairs_device, but the older (train) node of pair i is slot train_slots[i] of an
 * EXTERNAL device-resident map (ext_frames x ext_line_cap records + counts + node ids), e.g. the
 * all-gathered keyframes of all ranks: loop-closure matching, BASELINE.json configs 4 and 5. */
LF_API int lf_match_external_device(lf_ctx *ctx, const int32_t *query_frames, const int32_t *train_slots,
                                    int n_pairs, const lf_line_record *d_ext_recs,
                                    const int32_t *d_ext_nlines, const uint64_t *d_ext_ids, int ext_frames,
                                    int ext_line_cap);

/* Device memory: bytes this context holds (everything of a batch stays resident: ~18 MB per 640x480 frame of max_batch), and the
 * device's free / total bytes now.  A context whose buffers do not fit returns LF_ERR_CAPACITY from lf_ctx_create* with the
 * buffer's name in lf_last_error -- callers that keep several batches in flight size them with this. */
LF_API int lf_ctx_device_bytes(lf_ctx *ctx, unsigned long long *held, unsigned long long *dev_free, unsigned long long *dev_total);

/* ---- multi-GPU inside the library: ONE RCCL collective per batch of key frames ----------------
 * One process per GPU.  A context joins a communicator of `world_size` ranks: rank 0 obtains a 128-byte id with
 * lf_comm_unique_id and hands it to the other ranks by any side channel (MPI, torch.distributed, a file, ROS), then every
 * rank calls lf_comm_init with it (collective: ncclCommInitRank).  RCCL is loaded at run time (the copy already in the
 * process -- e.g. PyTorch's -- or /opt/rocm/lib/librccl.so.1); without it these calls return LF_ERR_UNSUPPORTED.
 * Several contexts of one process may share a communicator (lf_comm_attach): their collectives are then issued in the
 * same order on every rank, as RCCL requires. */
#define LF_COMM_ID_BYTES 128
LF_API int lf_comm_unique_id(uint8_t id[LF_COMM_ID_BYTES]);
LF_API int lf_comm_init(lf_ctx *ctx, int world_size, int rank, const uint8_t id[LF_COMM_ID_BYTES], int max_keyframes);
LF_API int lf_comm_attach(lf_ctx *ctx, lf_ctx *owner);     /* share owner's communicator (same device, same process) */
LF_API int lf_comm_destroy(lf_ctx *ctx);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank) and how many ncclAllGather calls this context has
 * issued: lets a multi-rank run show that the exchange went through RCCL with the rank count it was launched with. */
LF_API int lf_comm_info(lf_ctx *ctx, int *n_ranks, int *rank, long long *n_allgathers);
/* The exchange of key-frame line maps: the n_kf frame slots kf_slots[] (HOST array) of this context's last batch are
 * packed -- per key frame one header row (line count, node id + id_offset) followed by line_cap lf_line_record rows, all
 * 1040-byte rows -- and gathered from every rank with ONE ncclAllGather on the context stream.  The result is the map
 * lf_match_external_device / lf_line_matching_device take: *d_recs (row stride *ext_line_cap = line_cap + 1 records
 * per key frame), *d_nlines, *d_ids, *n_frames = world_size * n_kf slots, slot = rank * n_kf + k.  The buffers stay owned
 * by the context and are overwritten by its next exchange.  Asynchronous; every rank must pass the same n_kf. */
LF_API int lf_allgather_keyframes(lf_ctx *ctx, const int32_t *kf_slots, int n_kf, uint64_t id_offset,
                                  const lf_line_record **d_recs, const int32_t **d_nlines, const uint64_t **d_ids,
                                  int *n_frames, int *ext_line_cap);

/* MatchingResult Node::matchNodePair(const Node* older_node) (src/node.h:107) for two nodes whose
 * `lines` live in host memory: uploads both line maps into slots 0/1 of the context, runs
 * lineMatching + RANSAC + LM, returns the flat MatchingResult.  Matches / inliers of the pair are
 * then available as pair 0 (lf_pair_get_matches / lf_pair_get_inliers).  Needs max_batch >= 2.
 * Every host-resident entry point (lf_match_node_pair*, lf_line_matching_node_pair, lf_solve_node_pair, lf_feature_match_node_pair,
 * lf_mle_lines, lf_refine_pair) OVERWRITES frame slots 0 and 1 of the context: afterwards lf_frame_get_lines(0 / 1) returns the
 * uploaded maps, while the segment / label getters of those slots still describe the last detect batch -- keep a context for
 * node pairs apart from one whose batch results are still needed.
 * The row stride of a gathered map (lf_allgather_keyframes) is line_cap + 1, its capacity line_cap: a key frame whose owner
 * had more lines than that arrives cut, and every pair against it carries LF_OVF_LINES. */
LF_API int lf_match_node_pair(lf_ctx *ctx, const lf_line_record *newer, int n_newer, uint64_t id_newer,
                              const lf_line_record *older, int n_older, uint64_t id_older,
                              lf_pair_result *out);

/* ---- callers of the pair solver (HOST functions, no device work) ---------------
 * The pose graph as GraphManager sees it when it picks comparison candidates: node ids 0 .. n_nodes-1 in
 * insertion order (vertex id == node id), Node::matchable_ per node (null: all matchable), the edges added so
 * far (undirected), and GraphManager::keyframe_ids_. */
typedef struct lf_graph_view {
  int32_t n_nodes;
  const uint8_t *matchable;
  int32_t n_edges;
  const int32_t *edge_from, *edge_to;
  int32_t n_keyframes;
  const int32_t *keyframe_ids;
} lf_graph_view;
/* QList<int> GraphManager::getPotentialEdgeTargetsWithDijkstra(new_node, sequential_targets, geodesic_targets,
 * sampled_targets, predecessor_id, include_predecessor) (src/graph_manager.cpp:204-320): the older nodes the new
 * node is to be compared with, in the reference's list order (sampled and geodesic picks are pushed to the front,
 * sequential ones and the predecessor to the back).  geodesic_depth is the ParameterServer value of that name;
 * the reference's rand() draws come from lf_rand31(rng_seed, rng_stream, 0, 1, ...).  predecessor_id < 0 means
 * the newest node.  Returns LF_ERR_CAPACITY (with *n_out set) if out_cap is too small. */
LF_API int lf_candidate_targets(const lf_graph_view *graph, int predecessor_id, int sequential_targets,
                                int geodesic_targets, int sampled_targets, int geodesic_depth,
                                int include_predecessor, uint64_t rng_seed, uint64_t rng_stream, int32_t *out_ids,
                                int out_cap, int *n_out);
/* ---- GraphManager::nodeComparisons (src/graph_manager.cpp:419-708) around ONE batched solve, and the edges it hands to
 * GraphManager::addEdgeToG2O (:928-1014).  The pose graph itself (g2o optimizer) stays with the caller; this is the
 * decision logic that chooses the comparisons, judges their results and produces the edge records.
 * lf_compare_params: the ParameterServer options the function reads (src/parameter_server.cpp:91-103,147-148;
 * lf_compare_params_init_launch applies launch/lineslam.launch:15-16,23,34-36). */
typedef struct lf_compare_params {
  double min_translation_meter;   /* 0.0   (launch: 0.01) */
  double min_rotation_degree;     /* 0.0   (launch: 0.1)  */
  double max_translation_meter;   /* 1e10 */
  int32_t max_rotation_degree;    /* 360  */
  int32_t predecessor_candidates; /* 2     (launch: 1) */
  int32_t neighbor_candidates;    /* 2     (launch: 0) */
  int32_t min_sampled_candidates; /* 2     (launch: 0) */
  int32_t geodesic_depth;         /* 3    */
  int32_t keep_all_nodes;         /* 0     (launch: 1) */
  int32_t keep_good_nodes;        /* 0    */
  int32_t min_matches;            /* 20    (launch: 10): "min_matches" of the ParameterServer */
} lf_compare_params;
LF_API void lf_compare_params_init(lf_compare_params *p);
LF_API void lf_compare_params_init_launch(lf_compare_params *p);
/* One call of addEdgeToG2O(edge, n1, n2, largeEdge, set_estimate, motion_estimate), in the order nodeComparisons makes them. */
typedef struct lf_edge {
  int32_t id1, id2;               /* LoadedEdge3D::id1 (older), id2 (the new node)                          */
  double transform[16];           /* LoadedEdge3D::transform, row-major, v2 = v1 * transform                 */
  double information[36];         /* LoadedEdge3D::informationMatrix                                          */
  int32_t large_edge;             /* the largeEdge argument (isBigTrafo)                                      */
  int32_t set_estimate;           /* the set_estimate argument                                                */
  int32_t kind;                   /* 0 visual edge, 1 constant-position edge (:659-682)                       */
  int32_t n_point_inliers;        /* mr.inlier_matches.size(), what ranks the candidates (:568,577,608)       */
  int32_t n_line_inliers;
  int32_t accepted;               /* addEdgeToG2O returned true (an edge to a NEW vertex must be large, :941-946) */
} lf_edge;
typedef struct lf_comparison {
  int32_t added;                  /* the function's return value: a camera-camera edge was added              */
  int32_t n_edges;                /* addEdgeToG2O calls recorded in `edges`                                   */
  int32_t edge_to_keyframe;
  int32_t out_of_bounds;          /* the predecessor transform was too small / too fast: node dropped (:478-492) */
  int32_t best_id1;               /* curr_best_result_.edge.id1 (-1: none)                                    */
  int32_t valid_tf_estimate;      /* 0 if only the constant-position edge holds the node                      */
  int32_t n_candidates;
  int32_t predecessor_matched;
  double pose_new[16];            /* motion_estimate: the new vertex's estimate v1 * transform (identity if none) */
} lf_comparison;
/* The decisions alone (HOST, no device work), given the comparison results: pred = matchNodePair(new, predecessor) of the
 * initial comparison (NULL if it was not run: both minimum-motion parameters <= 0), cand_ids / cand_results = the
 * candidates of the main loop in the order compared.  graph / poses (row-major 4x4 vertex estimates per node) / stamps
 * (seconds per node) describe the graph before the new node; the new node gets id graph->n_nodes. */
LF_API int lf_node_comparisons_decide(const lf_graph_view *graph, const double *poses, const double *stamps, double stamp_new,
                                      const lf_compare_params *cp, const lf_pair_result *pred, const int32_t *cand_ids,
                                      const lf_pair_result *cand_results, int n_cand, int n_features_new, lf_edge *edges,
                                      int edge_cap, lf_comparison *out);
/* GraphManager::nodeComparisons for the frames of the last batch: node i of the graph = frame slot i, the new node = frame
 * slot graph->n_nodes.  Runs the initial comparison with the predecessor (if the minimum-motion parameters ask for it),
 * draws the candidates (lf_candidate_targets; prev_best_id as :533-535, -1: none), solves ALL of them in ONE batched launch
 * (the reference fans them out over a QThreadPool, :555) and applies the decisions above.  n_features_new = the new node's
 * feature_locations_2d_.size() (point features; pass min_matches or more when only lines are used with keep_all_nodes).
 * Synchronous. */
LF_API int lf_node_comparisons(lf_ctx *ctx, const lf_graph_view *graph, const double *poses, const double *stamps,
                               double stamp_new, const lf_compare_params *cp, uint64_t rng_seed, int prev_best_id,
                               int n_features_new, lf_edge *edges, int edge_cap, lf_comparison *out);

/* Node::vel as GraphManager::addNode sets it (src/graph_manager.cpp:764-784): translation difference of the two
 * row-major 4x4 double poses (the node itself and the node five ids before it) over |dt|, cast to float. */
LF_API int lf_instant_velocity(const double *T_new, const double *T_old, double dt, float *vel3);
/* The constant-velocity edge of Node::matchNodePair (src/node.cpp:1584-1599), used when no transformation was
 * found and apply_const_vel is set: T = [I | R_older^T (dt vel_older)] in float, row-major 4x4. */
LF_API int lf_const_velocity_transform(const float *pose_older, const float *vel3, double dt, float *T);

#ifdef __cplusplus
}
#endif
#endif /* LINEFRONT_H */
