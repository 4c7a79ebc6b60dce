// lf_pair_legacy.h -- internal interface of the legacy point-feature RANSAC (Node::getRelativeTransformationTo,
// src/node.cpp:1134-1338); see lf_pair_legacy.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"
#include "lf_math.h"
#include "lf_linalg.h"
#include "lf_pose.h"

#define LF_LEGACY_CAP 1024        // matches per pair (Node::featureMatching yields at most one per query descriptor: <= 1024)

struct LegacyResult {
  float T[16];                    // resulting_transformation: newer -> older, row-major
  float rmse;
  int found;                      // the return value: matches.size() >= min_inlier_threshold
  int n_inliers;                  // matches.size()
  int valid_iterations, best_iteration, iterations_run;
};
struct LegacyArgs {
  const float *pts_q, *pts_t;     // feature_locations_3d_ of the newer (query) / older (train) node: float4 each
  const int *mq, *mt;             // initial_matches: queryIdx / trainIdx
  const float *md;                //                  distance
  int n;                          // initial_matches->size()
  int min_matches, iterations;    // "min_matches", "ransac_iterations"
  float max_dist_m;               // "max_dist_for_inliers" (float in the reference)
  uint64_t seed, stream;
  lf_point_model pm;
  LegacyResult *out;
  int *out_inliers;               // [n] indices into the caller's match arrays (the inlier DMatches, in the sorted order the reference keeps)
};
void lf_legacy_launch(const LegacyArgs &a, hipStream_t stream);
