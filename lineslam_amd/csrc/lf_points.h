// lf_points.h -- internal interface of the point side: Node::projectTo3D
// (src/node.cpp:952-1018) and Node::featureMatching, BRUTEFORCE / ORB branch (src/node.cpp:606-641).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"

struct PointConsts {
  int W, H;
  double K[9];
  double depth_scaling;   // ParameterServer "depth_scaling_factor" (1.0)
  int max_keyp;           // "max_keypoints" (600)
  int kp_cap;             // key points per frame in the input / output arrays
  int desc_cap;           // descriptors per frame (<= 1024)
  double nn_ratio;        // "nn_distance_ratio" (0.5)
  uint64_t rng_seed;
};
struct PointBuffers {
  const float *depth; size_t depth_frame_stride; int depth_row_stride;   // elements
  const float *kp_xy;     // [frames][kp_cap][2]  cv::KeyPoint::pt
  const int *nkp;         // [frames]
  float *points;          // [frames][kp_cap][4]  feature_locations_3d_
  int *npts;              // [frames]
  int *kept;              // [frames][kp_cap] index of the surviving key points (may be null)
  const uint8_t *desc;    // [frames][desc_cap][32] ORB descriptors
  const int *ndesc;       // [frames]
  const uint64_t *frame_ids;
  const int *pair_q, *pair_t;
  int *fm_q, *fm_t;       // [pairs][desc_cap]
  float *fm_d;            // [pairs][desc_cap]
  int *fm_n;              // [pairs]
};
void lf_points_project_launch(const PointConsts &c, const PointBuffers &b, int n_frames, hipStream_t stream);
void lf_points_match_launch(const PointConsts &c, const PointBuffers &b, int n_pairs, hipStream_t stream);
void lf_points_ingest_launch(const uint8_t *rgb, const uint16_t *depth16, uint8_t *gray, float *depth, size_t npix,
                             double depth_factor, hipStream_t stream);
