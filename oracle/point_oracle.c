/* oracle/point_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the point side that feeds the hybrid solver (the part
 * without the ORB extractor):
 *   Node::featureMatching, BRUTEFORCE / ORB branch        src/node.cpp:606-641
 *        (cv::BFMatcher "BruteForce-HammingLUT" knnMatch k = 2, ratio test, unique-train filter, distance offset)
 *   Node::projectTo3D                                     src/node.cpp:952-1018
 * Parity status: "parity unpinned" -- OpenCV is absent.  knnMatch is restated from OpenCV 2.4's batchDistance
 * (modules/core/src/stat.cpp): candidates are visited in train order and inserted into the sorted k-best list
 * BEHIND entries of equal distance, i.e. the lower train index wins ties.  Pinned by known-answer tests against a
 * numpy brute force (tests/test_oracle_points.py).  rand() -> o_rand31(seed, stream, query index).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/linefront.h"
#include "o_linalg.h"   /* the oracle's own counter generator (NOT the product's lf_linalg.h) */

static int o_hamming256(const uint8_t *a, const uint8_t *b) {
  int d = 0, i;
  for (i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

/* returns the number of matches; out_q / out_t / out_d sized >= nq */
int oracle_feature_match(const uint8_t *qdesc, int nq, const uint8_t *tdesc, int nt, double max_dist_ratio_fac,
                         uint64_t seed, uint64_t stream, int *out_q, int *out_t, float *out_d) {
  int i, j, n = 0;
  uint8_t *taken;
  if (nt < 2) return 0;   /* knnMatch would return fewer than k neighbours; the reference reads [1] regardless */
  taken = (uint8_t *)calloc((size_t)nt, 1);
  for (i = 0; i < nq; i++) {
    int b1 = -1, b2 = -1, d1 = 1 << 30, d2 = 1 << 30;
    for (j = 0; j < nt; j++) {
      int d = o_hamming256(qdesc + 32 * (size_t)i, tdesc + 32 * (size_t)j);
      if (d < d1) { d2 = d1; b2 = b1; d1 = d; b1 = j; }
      else if (d < d2) { d2 = d; b2 = j; }
    }
    (void)b2;
    {
      float m1 = (float)d1, m2 = (float)d2;
      float dist_ratio_fac = m1 / m2;
      if (dist_ratio_fac < max_dist_ratio_fac) {   /* float promoted to double, as in the reference */
        if (taken[b1]) continue;
        taken[b1] = 1;
        out_q[n] = i; out_t[n] = b1;
        out_d[n] = (float)(dist_ratio_fac + (float)o_rand31(seed, stream, (uint64_t)i) / (1000.0 * 2147483647.0));
        n++;
      }
    }
  }
  free(taken);
  return n;
}

/* kp: n x 2 floats (KeyPoint::pt).  out_pts: (x,y,Z,1) floats; out_kept: index of the surviving key points.
 * K row-major 3x3 (cam_info->K).  Returns the number of 3D points (<= max_keyp).
 * Deviation: round(p) can reach rows / cols (e.g. y = 479.6) where the reference reads out of bounds; the
 * index is clamped to the last row / column here and on the device. */
int oracle_project_to_3d(const float *kp, int n, const float *depth, int stride_elems, int cols, int rows,
                         const double *K, double depth_scaling, int max_keyp, float *out_pts, int *out_kept) {
  float fx = (float)(1. / K[0]), fy = (float)(1. / K[4]), cx = (float)K[2], cy = (float)K[5];
  int i, m = 0;
  for (i = 0; i < n; i++) {
    float px = kp[2 * i], py = kp[2 * i + 1], Z, x, y;
    int ix, iy;
    if (px >= cols || px < 0 || py >= rows || py < 0 || px != px || py != py) continue;
    iy = (int)round((double)py); ix = (int)round((double)px);
    if (iy > rows - 1) iy = rows - 1;
    if (ix > cols - 1) ix = cols - 1;
    Z = (float)(depth[(size_t)iy * stride_elems + ix] * depth_scaling);
    if (Z != Z) continue;
    x = (px - cx) * Z * fx;
    y = (py - cy) * Z * fy;
    out_pts[4 * m] = x; out_pts[4 * m + 1] = y; out_pts[4 * m + 2] = Z; out_pts[4 * m + 3] = 1.0f;
    out_kept[m] = i;
    m++;
    if (m >= max_keyp) break;
  }
  return m;
}

/* loadRawData pixel conversions (openni_listener.cpp:1233-1246) + Node::Node grey image (node.cpp:193), restated from
 * OpenCV 2.4 (cvtColor fixed point: 4899 / 9617 / 1868, shift 14; convertTo float with the scale cast to float). */
void oracle_ingest_tum(const uint8_t *rgb, const uint16_t *depth16, size_t npix, double depth_factor, uint8_t *gray,
                       float *depth) {
  size_t i;
  float inv = (float)(1.0 / depth_factor);
  for (i = 0; i < npix; i++) {
    unsigned r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    float d = (float)depth16[i];
    gray[i] = (uint8_t)((b * 4899u + g * 9617u + r * 1868u + 8192u) >> 14);
    if (d < 1e-5) d = NAN;
    depth[i] = d * inv;
  }
}
