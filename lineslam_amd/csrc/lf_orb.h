// lf_orb.h -- internal interface of the ORB extractor: the ORB branch of Node::Node
// (src/node.cpp:222-290) = AorbFeatureDetector (src/aorb.cpp:727-940, src/feature_adjuster.cpp:86-89) + removeDepthless +
// retainBest(max_keypoints) + OrbDescriptorExtractor, for a batch of frames.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"

#define LF_ORB_LEVELS 8
#define LF_ORB_EDGE 31
#define LF_ORB_HALF 15
#ifndef LF_ORB_CAND_CAP
#define LF_ORB_CAND_CAP 16384     // non-maximum-suppressed FAST corners per frame (all levels) the selection stage can sort in LDS
#endif
#define LF_ORB_KP_MAX 1024        // max_keypoints of one call

struct OrbConsts {
  int W, H;
  int lw[LF_ORB_LEVELS], lh[LF_ORB_LEVELS];
  int loff[LF_ORB_LEVELS];        // byte offset of a level inside a frame's pyramid
  int total;                      // bytes of one pyramid
  double scale_x[LF_ORB_LEVELS], scale_y[LF_ORB_LEVELS];   // cv::resize: 1 / (dst / src) of level l from level l-1
  float sf[LF_ORB_LEVELS], inv_sf[LF_ORB_LEVELS];          // getScale(level), 1 / getScale(level)
  int nper[LF_ORB_LEVELS];        // nfeaturesPerLevel
  int umax[LF_ORB_HALF + 2];
  int blur_k[7];                  // 8-bit fixed-point Gaussian taps (host libm exp, as cv::getGaussianKernel)
  int fast_threshold;
  int max_keypoints;
  int kp_cap;                     // row capacity of the output arrays
};

struct OrbBuffers {
  const uint8_t *gray; size_t gray_frame_stride; int gray_row_stride;
  const float *depth; size_t depth_frame_stride; int depth_row_stride;   // may be null: no removeDepthless
  uint8_t *pyr, *blur, *score;    // [B][total]
  unsigned *cand;                 // [B][LF_ORB_CAND_CAP]  score << 24 | level << 19 | y << 10 | x
  int *ncand;                     // [B]
  int *hist;                      // [B][LEVELS][256] FAST scores of the border-filtered corners
  float *sel;                     // [B][LF_ORB_KP_MAX][4]  x_l, y_l (level pixels), level, response of the selected key points
  int *nsel;                      // [B]
  float *kp_xy;                   // [B][kp_cap][2]   out: cv::KeyPoint::pt (level-0 pixels)
  float *kp_meta;                 // [B][kp_cap][4]   out (may be null): response, angle [deg], octave, size
  uint8_t *desc;                  // [B][kp_cap][32]  out
  int *nkp;                       // [B] out; [B + f] = 1 if frame f had more than LF_ORB_CAND_CAP corners (overflow)
  // VideoDynamicAdaptedFeatureDetector (src/feature_adjuster.cpp:107-186): per-frame FAST thresholds chosen on the device
  int *thr_frame;                 // [B] threshold of the detection whose key points are returned (null: c.fast_threshold for all)
  double *adj_state;              // [1] DetectorAdjuster::thresh_, carried from frame to frame and from call to call
};
struct OrbAdjuster { double min_thresh, max_thresh, inc, dec; int min_features, max_features, max_iters, base_threshold; };

void lf_orb_launch(const OrbConsts &c, const OrbBuffers &b, int n_frames, hipStream_t stream);
// the same with the dynamically adapted FAST threshold: b.thr_frame / b.adj_state must be set
void lf_orb_launch_adjusted(const OrbConsts &c, const OrbBuffers &b, const OrbAdjuster &a, int n_frames, hipStream_t stream);
