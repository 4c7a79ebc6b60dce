// lf_front.h -- internal interface of the 3D-line stage (a9-a18 of SURVEY.md section 8):
// everything Node::detect3DLines does after the LSD call (src/line/lineslam.cpp:213-357).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"

#define LF_MAX_SAMPLES 104      // numSmp <= line_sample_max_num (100) -> at most 101 samples; 104 keeps the LDS of k_line3d at 10.4 KB (15 wavefronts per CU)
#define LF_CAND_STRIDE 32       // doubles per candidate in cand_out
// cand_out layout: [0..2] A  [3..5] B  [6..14] covA  [15..23] covB  [24] numSmp  [25] #valid samples
//                  [26] #RANSAC inliers  [27] levmar iterations  [28] levmar stop reason
//                  [29..31] A before MLE (RANSAC end point)

struct FrontConsts {
  int W, H;
  double K[9], Kinv[9];
  lf_params P;
  int cand_cap;      // candidates (LSD segments) examined per frame (= seg_cap: every segment LSD reports is examined)
  int line_cap;      // records per frame
  int seg_cap;       // row capacity of the LSD segment output
  int pts_rows;      // 0: k_mle re-derives a line's supporting points from its segment and inlier mask (the batch path);
                     // 1: lf_mle_lines -- the caller's points, row block i of `pts` belongs to candidate i
};

struct FrontBuffers {
  const uint8_t *gray;  size_t gray_frame_stride;  int gray_row_stride;     // bytes
  const float *depth;   size_t depth_frame_stride; int depth_row_stride;    // elements
  int16_t *gxy;              // [B][H][W][2] Sobel ksize 5 (gx, gy) interleaved: one 32-bit load per pixel (exact integers, |v| <= 24480)
  const double *segs;        // [B][seg_cap][5]   from the LSD stage
  const int *nsegs;          // [B]
  const uint64_t *frame_ids; // [B]  keys of the counter-based generator
  int *cand_flag;            // [B][cand_cap]: 0 short, 1 no depth, 2 3D line
  double *cand_out;          // [B][cand_cap][LF_CAND_STRIDE]
  unsigned long long *cand_mask;       // [B][cand_cap][2] RANSAC inliers of a 3D line as a mask over the candidate's valid depth samples
                             //   (bit i = valid sample i, in sample order): with the segment it IS the list of supporting points
  double *pts;               // [line_cap][LF_MAX_SAMPLES*3] lf_mle_lines only (pts_rows = 1): the caller's supporting points
  lf_line_record *recs;      // [B][line_cap]
  int *nlines;               // [B]  (may exceed line_cap: overflow)
  int *mle_list;             // [B][3][line_cap] line ids for the MLE stage by #support points: <= 16, 17..32, more
  int *mle_cnt;              // [B][3]
};

void lf_front_launch(const FrontConsts &c, const FrontBuffers &b, int n_frames, hipStream_t stream);
// only the MLE kernels, on work lists / support points the caller has placed in the buffers (lf_mle_lines)
void lf_front_launch_mle(const FrontConsts &c, const FrontBuffers &b, int n_frames, hipStream_t stream);
