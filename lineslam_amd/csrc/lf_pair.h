// lf_pair.h -- internal interface of the pair solver (a19-a25 of SURVEY.md section 8):
// Node::lineMatching (src/node.cpp:1619-1694), getTransform_PtsLines_ransac (src/line/motion.cpp:605-849,
// line matches), getTransformFromHybridMatchesG2O (src/transformation_estimation.cpp:218-461) and the
// matchNodePair bookkeeping (src/node.cpp:1494-1615).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"
#include "lf_pose.h"

#define LF_MAX_MATCHES 256        // line matches per pair handled by the pose kernel (4 per lane)
#define LF_MATCH_LINE_CAP 512     // lines per frame the matcher stages in LDS (the compiled maximum of lf_caps::line_cap)
#define LF_RANSAC_MAX_ITERS 1024  // sample table capacity
#define LF_MOTION_STRIDE 16
#define LF_MAX_PT_MATCHES 512     // point matches per pair handled by the hybrid pose kernel (8 per lane)
// per pair: the matches' compact measurements (48 doubles each) + the RANSAC winner (8 + 8: record, float model) + the blocks of
// the wavefront-per-pair refinement (lf_pose_wave.h: V | W | bl 78, the columns of Vi 36, two landmark sets 12 per match)
#define LF_PAIR_WS_DOUBLES (LF_MAX_MATCHES * (48 + 78 + 36 + 12) + 16)

struct PairConsts {
  lf_params P;
  double cos_angle_thresh;   // cos(30 * 3.14159265 / 180), host libm (node.cpp:1624,1647)
  int line_cap;              // records per frame
  int match_cap;             // rows of the match list per pair
  lf_point_model pm;         // hybrid solver: errorFunction2 constants (misc.cpp:704-711, host libm)
  double focal;              //   K(0,0) for compPt3dCov (transformation_estimation.cpp:245)
  double cos_degeneracy;     // cos(5 * 3.14159265 / 180), host libm (motion.cpp:407,428)
  int pt_match_cap;          // point matches per pair (<= LF_MAX_PT_MATCHES)
  int mode;                  // LF_MODE_SOLVE: RANSAC + refinement;  LF_MODE_REFINE: getTransformFromHybridMatchesG2O alone on
  int refine_iters;          //   every match given, starting from results[pair].T, refine_iters iterations (k_pose_hybrid)
};
enum { LF_MODE_SOLVE = 0, LF_MODE_REFINE = 1 };

struct PairBuffers {
  const lf_line_record *recs;   // [B][line_cap]   query (newer) side: this context's last batch
  const int *nlines;            // [B]
  const uint64_t *frame_ids;    // [B]  node ids (adjacency window, loop-closure threshold, RNG stream)
  const lf_line_record *recs_t; // train (older) side: the same arrays, or an external keyframe map
  const int *nlines_t;          //   (e.g. the result of the RCCL all-gather of other ranks' line maps)
  const uint64_t *frame_ids_t;
  int line_cap_t;
  const int *pair_q, *pair_t;   // [n_pairs] frame slots of the newer (query) and older (train) node
  unsigned *live_idx;           // [n_pairs][line_cap*line_cap] live entries of descDiff (all gates passed, value < 100):
  double *live_val;             //   (query << 16 | train), distance -- the matrix itself is never materialised
  const unsigned char *adjacent;// [n_pairs] Node::lineMatching's adjacentFrame argument: 0 / 1, 255 = from the node ids; may be null
  int *match_q, *match_t;       // [n_pairs][match_cap]
  double *match_d;              // [n_pairs][match_cap]
  int *nmatches;                // [n_pairs] (may exceed match_cap)
  lf_pair_result *results;      // [n_pairs]
  int *inliers;                 // [n_pairs][LF_MAX_MATCHES] indices into the match list
  double *ws;                   // [n_pairs][LF_PAIR_WS_DOUBLES]
  // ---- hybrid solver (points + lines, BASELINE config 3); allocated on first use
  const float *pts, *pts_t;     // [frames][pt_cap][4]  Node::feature_locations_3d_ (x,y,z,1; z NaN = no depth)
  int pt_cap, pt_cap_t;
  const int *pm_q, *pm_t;       // [n_pairs][pm_stride] point matches (queryIdx, trainIdx)
  int pm_stride;
  const int *npm;               // [n_pairs]
  int *pt_inliers;              // [n_pairs][LF_MAX_PT_MATCHES] indices into the point match list
  double *ws_h;                 // [n_pairs][lf_pair_hybrid_ws_doubles()]
  double *motion_d;             // [n_pairs][LF_MOTION_STRIDE] lines-only RANSAC: R (9), t (3), diagnostics
};

enum { LF_SOLVER_LINES = 0, LF_SOLVER_HYBRID = 1, LF_SOLVER_RELMOTION = 2, LF_SOLVER_NONE = 3 };   // what follows k_match
// run_match = false: the match lists (match_q / match_t / nmatches) were supplied by the caller
void lf_pair_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t stream, int solver = LF_SOLVER_LINES,
                    bool run_match = true);
void lf_pair_descdiff_launch(const PairConsts &c, const PairBuffers &b, int pair, double *D, hipStream_t stream);
void lf_pair_relmotion_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t stream);
size_t lf_pair_hybrid_ws_doubles();
void lf_pair_hybrid_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t stream);
