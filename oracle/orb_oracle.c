/* oracle/orb_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the ORB side of Node::Node for BASELINE.json config 3:
 *   detection    AorbFeatureDetector(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, fastThreshold)   src/feature_adjuster.cpp:86-89
 *                = AORB::operator() -> computeKeyPoints                                            src/aorb.cpp:727-940, 603-688
 *                (pyramid, FAST-9/16 + non-maximum suppression per level, border filter, retainBest, HarrisResponses :57-99,
 *                IC_Angle :103-131, scale back)
 *   prefilter    removeDepthless + KeyPointsFilter::retainBest(max_keypoints) + resize            src/node.cpp:101-125, 257-263
 *   descriptors  OrbDescriptorExtractor::compute = the same operator() with the key points given  src/features.cpp:197, aorb.cpp:855-905
 *                (GaussianBlur 7x7 sigma 2 per level, rBRIEF-256 steered by the key point angle, computeOrbDescriptor :135-183)
 *
 * PARITY UNPINNED: everything underneath is OpenCV 2.4 (absent from this image and from the reference tree, version unpinned,
 * README.md:8) and is restated from its published algorithms: cv::resize INTER_LINEAR for 8-bit (11-bit fixed-point
 * coefficients, imgwarp.cpp), copyMakeBorder BORDER_REFLECT_101, cv::FAST with cornerScore<16> and 3x3 strict non-maximum
 * suppression (fast.cpp / fast_score.cpp), cv::fastAtan2 (the 7th-order polynomial of 2.4 mathfuncs.cpp), cv::GaussianBlur for
 * 8-bit (8-bit fixed-point separable kernel, smooth.cpp / filter.cpp), cvRound = round half to even.  One documented
 * deviation: KeyPointsFilter::retainBest orders its survivors with std::nth_element (implementation-defined), and Node then
 * cuts that order at max_keypoints; here the order is (response descending, detection order ascending), so the SET equals
 * the reference's whenever no two responses tie at the cut.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../lineslam_amd/csrc/lf_orb_pattern.h"
#ifdef ORACLE_LFMATH     /* the `lf` flavour evaluates what the kernels evaluate on the device with lf_math.h (see front_oracle.c) */
#include "../lineslam_amd/csrc/lf_math.h"
#endif

#define ORB_LEVELS 8
#define ORB_EDGE 31
#define ORB_PATCH 31
#define ORB_HALF 15

static int o_cvround(double v) { return (int)nearbyint(v); }
static int o_cvfloor(double v) { return (int)floor(v); }
static int o_reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}

/* getScale (aorb.cpp:555-558) with the float member scaleFactor = 1.2f */
static float o_get_scale(int level) { return (float)pow((double)1.2f, (double)level); }

void oracle_orb_level_size(int w, int h, int level, int *lw, int *lh) {
  float scale = 1 / o_get_scale(level);
  *lw = o_cvround((double)(w * scale));
  *lh = o_cvround((double)(h * scale));
}

/* cv::resize(src, dst, dsize, ., ., INTER_LINEAR) for CV_8UC1 (imgwarp.cpp: 11-bit coefficients, two-pass fixed point) */
void oracle_orb_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
  double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *buf0 = (int *)malloc(sizeof(int) * (size_t)dw), *buf1 = (int *)malloc(sizeof(int) * (size_t)dw);
  short *ia = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
  int dx, dy;
  for (dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = o_cvfloor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ia[2 * dx] = (short)o_cvround((double)((1.f - fx) * 2048));
    ia[2 * dx + 1] = (short)o_cvround((double)(fx * 2048));
  }
  for (dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = o_cvfloor(fy), sy0, sy1;
    short b0, b1;
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    b0 = (short)o_cvround((double)((1.f - fy) * 2048));
    b1 = (short)o_cvround((double)(fy * 2048));
    sy0 = sy; sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
    for (dx = 0; dx < dw; dx++) {
      int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
      buf0[dx] = src[(size_t)sy0 * sw + sx] * ia[2 * dx] + src[(size_t)sy0 * sw + sx1] * ia[2 * dx + 1];
      buf1[dx] = src[(size_t)sy1 * sw + sx] * ia[2 * dx] + src[(size_t)sy1 * sw + sx1] * ia[2 * dx + 1];
    }
    for (dx = 0; dx < dw; dx++)
      dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (buf0[dx] >> 4)) >> 16) + ((b1 * (buf1[dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(buf0); free(buf1); free(ia);
}

/* cornerScore<16> (fast_score.cpp): the largest threshold for which the pixel is still a FAST-9 corner */
static const int o_circle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
static int o_min(int a, int b) { return a < b ? a : b; }
static int o_max(int a, int b) { return a > b ? a : b; }
static int o_corner_score(const uint8_t *img, int w, int x, int y, int threshold) {
  int d[25], k, v = img[(size_t)y * w + x], a0, b0;
  for (k = 0; k < 25; k++) d[k] = v - img[(size_t)(y + o_circle[k % 16][1]) * w + x + o_circle[k % 16][0]];
  a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = o_min(d[k + 1], d[k + 2]);
    a = o_min(a, d[k + 3]);
    if (a <= a0) continue;
    a = o_min(a, d[k + 4]); a = o_min(a, d[k + 5]); a = o_min(a, d[k + 6]); a = o_min(a, d[k + 7]); a = o_min(a, d[k + 8]);
    a0 = o_max(a0, o_min(a, d[k]));
    a0 = o_max(a0, o_min(a, d[k + 9]));
  }
  b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = o_max(d[k + 1], d[k + 2]);
    b = o_max(b, d[k + 3]); b = o_max(b, d[k + 4]); b = o_max(b, d[k + 5]);
    if (b >= b0) continue;
    b = o_max(b, d[k + 6]); b = o_max(b, d[k + 7]); b = o_max(b, d[k + 8]);
    b0 = o_min(b0, o_max(b, d[k]));
    b0 = o_min(b0, o_max(b, d[k + 9]));
  }
  return -b0 - 1;
}
/* the segment test of FAST_t<16> (fast.cpp): nine contiguous circle pixels all darker than v - t or all brighter than v + t */
static int o_is_corner(const uint8_t *img, int w, int x, int y, int threshold) {
  int v = img[(size_t)y * w + x], k, cd = 0, cb = 0;
  for (k = 0; k < 25; k++) {
    int p = img[(size_t)(y + o_circle[k % 16][1]) * w + x + o_circle[k % 16][0]];
    if (p < v - threshold) { if (++cd > 8) return 1; } else cd = 0;
    if (p > v + threshold) { if (++cb > 8) return 1; } else cb = 0;
  }
  return 0;
}
/* score image of one level: cornerScore at FAST corners of rows / columns 3 .. n-4, 0 elsewhere */
void oracle_orb_fast_scores(const uint8_t *img, int w, int h, int threshold, uint8_t *score) {
  int x, y;
  memset(score, 0, (size_t)w * h);
  if (threshold < 0) threshold = 0;
  if (threshold > 255) threshold = 255;
  for (y = 3; y < h - 3; y++)
    for (x = 3; x < w - 3; x++)
      if (o_is_corner(img, w, x, y, threshold)) score[(size_t)y * w + x] = (uint8_t)o_corner_score(img, w, x, y, threshold);
}

/* cv::fastAtan2 (OpenCV 2.4 mathfuncs.cpp), degrees */
static float o_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  float ax = fabsf(x), ay = fabsf(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}
float oracle_orb_fast_atan2(float y, float x) { return o_fast_atan2(y, x); }

typedef struct { float x, y, response, angle; int level, order; } o_kp;

static void o_umax(int *umax) {   /* aorb.cpp:632-647 */
  int v, v0, vmax = o_cvfloor(ORB_HALF * sqrt(2.f) / 2 + 1), vmin = (int)ceil(ORB_HALF * sqrt(2.f) / 2);
  for (v = 0; v <= vmax; ++v) umax[v] = o_cvround(sqrt((double)ORB_HALF * ORB_HALF - v * v));
  for (v = ORB_HALF, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}
static float o_harris(const uint8_t *img, int w, float px, float py) {   /* HarrisResponses, blockSize 7, k 0.04 */
  const int bs = 7, r = bs / 2;
  float scale = (1 << 2) * bs * 255.0f, s4;
  int x0 = o_cvround((double)(px - r)), y0 = o_cvround((double)(py - r)), a = 0, b = 0, c = 0, i, j;
  scale = 1.0f / scale;
  s4 = scale * scale * scale * scale;
  for (i = 0; i < bs; i++)
    for (j = 0; j < bs; j++) {
      const uint8_t *p = img + (size_t)(y0 + i) * w + x0 + j;
      int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
      int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * s4;
}
static float o_ic_angle(const uint8_t *img, int w, float px, float py, const int *umax) {   /* IC_Angle */
  int m01 = 0, m10 = 0, u, v;
  const uint8_t *center = img + (size_t)o_cvround((double)py) * w + o_cvround((double)px);
  for (u = -ORB_HALF; u <= ORB_HALF; ++u) m10 += u * center[u];
  for (v = 1; v <= ORB_HALF; ++v) {
    int vs = 0, d = umax[v];
    for (u = -d; u <= d; ++u) {
      int vp = center[u + v * w], vm = center[u - v * w];
      vs += (vp - vm);
      m10 += u * (vp + vm);
    }
    m01 += v * vs;
  }
  return o_fast_atan2((float)m01, (float)m10);
}

/* (response descending, detection order ascending) */
static int o_cmp_kp(const void *a, const void *b) {
  const o_kp *p = (const o_kp *)a, *q = (const o_kp *)b;
  if (p->response > q->response) return -1;
  if (p->response < q->response) return 1;
  return p->order < q->order ? -1 : (p->order > q->order ? 1 : 0);
}
static int o_cmp_order(const void *a, const void *b) { return ((const o_kp *)a)->order - ((const o_kp *)b)->order; }
/* KeyPointsFilter::retainBest(kps, n): everything whose response is >= the n-th largest response; survivors keep
 * their detection order */
static int o_retain_best(o_kp *k, int n_have, int n_keep) {
  float cut;
  int i, m = 0;
  if (n_keep <= 0) return 0;                 /* (the reference clears the vector for n == 0) */
  if (n_have <= n_keep) return n_have;
  qsort(k, (size_t)n_have, sizeof *k, o_cmp_kp);
  cut = k[n_keep - 1].response;
  for (i = 0; i < n_have; i++) if (k[i].response >= cut) k[m++] = k[i];
  qsort(k, (size_t)m, sizeof *k, o_cmp_order);
  return m;
}

/* 8-bit fixed-point Gaussian 7x7, sigma 2, BORDER_REFLECT_101 (cv::GaussianBlur on CV_8U) */
void oracle_orb_blur_kernel(int *ik) {
  double s2 = -0.5 / (2.0 * 2.0), sum = 0;
  float cf[7];
  int i;
  for (i = 0; i < 7; i++) { double x = i - 3.0, t = exp(s2 * x * x); cf[i] = (float)t; sum += cf[i]; }
  sum = 1. / sum;
  for (i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); ik[i] = o_cvround((double)(cf[i] * 256.f)); }
}
void oracle_orb_blur(const uint8_t *src, int w, int h, uint8_t *dst) {
  int ik[7], x, y, k, *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
  oracle_orb_blur_kernel(ik);
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int s = 0;
      for (k = 0; k < 7; k++) s += ik[k] * src[(size_t)y * w + o_reflect101(x + k - 3, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int s = 0, v;
      for (k = 0; k < 7; k++) s += ik[k] * tmp[(size_t)o_reflect101(y + k - 3, h) * w + x];
      v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  free(tmp);
}

/* computeOrbDescriptor, WTA_K = 2 */
static void o_descriptor(const uint8_t *img, int w, float px, float py, float angle_deg, uint8_t *desc) {
  float angle = angle_deg * (float)(3.1415926535897932384626433832795 / 180.f);
#ifdef ORACLE_LFMATH
  double sd, cd;
  float a, b;
#else
  float a = (float)cos((double)angle), b = (float)sin((double)angle);
#endif
  const uint8_t *center = img + (size_t)o_cvround((double)py) * w + o_cvround((double)px);
  int i, j;
#ifdef ORACLE_LFMATH
  lf_sincos_cr((double)angle, &sd, &cd);
  a = (float)cd; b = (float)sd;
#endif
  for (i = 0; i < 32; i++) {
    int val = 0;
    for (j = 0; j < 8; j++) {
      const signed char *p = LF_ORB_PATTERN + 4 * (8 * i + j);
      int t0 = center[o_cvround((double)(p[0] * b + p[1] * a)) * w + o_cvround((double)(p[0] * a - p[1] * b))];
      int t1 = center[o_cvround((double)(p[2] * b + p[3] * a)) * w + o_cvround((double)(p[2] * a - p[3] * b))];
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

/* Node::Node, ORB branch.  gray [h][w]; depth [h][dstride] floats or NULL (no removeDepthless); outputs at most
 * max_keypoints rows: kp_xy (x, y in level-0 pixels), kp_meta (response, angle [deg], octave, size), desc (32 bytes).
 * levels_out / blurred_out (optional): the eight pyramid levels / their blurred versions, concatenated.            */
/* AorbFeatureDetector::detect (aorb.cpp:727-940) on one frame: pyramid, FAST + non-maximum suppression + border filter, the
 * two retainBest steps per level, Harris responses, angles; key points in level order (coordinates scaled to level 0).
 * Returns their number -- what VideoDynamicAdaptedFeatureDetector counts.  lev / blur [ORB_LEVELS]: malloc'd here when
 * want_images (the caller frees), else freed here.                                                                    */
static int o_orb_detect(const uint8_t *gray, int w, int h, int fast_threshold, int nfeatures, o_kp *all, uint8_t **lev, uint8_t **blur,
                        int *lw, int *lh, int want_images, uint8_t *levels_out, uint8_t *blurred_out) {
  uint8_t *score;
  int nper[ORB_LEVELS], umax[ORB_HALF + 2], l, n_all = 0, order = 0, i;
  size_t off = 0;
  {   /* nfeaturesPerLevel (aorb.cpp:612-624) */
    float factor = (float)(1.0 / (double)1.2f);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)ORB_LEVELS));
    int sum = 0;
    for (l = 0; l < ORB_LEVELS - 1; l++) { nper[l] = o_cvround((double)nd); sum += nper[l]; nd *= factor; }
    nper[ORB_LEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  }
  o_umax(umax);
  for (l = 0; l < ORB_LEVELS; l++) {
    oracle_orb_level_size(w, h, l, &lw[l], &lh[l]);
    lev[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
    blur[l] = want_images ? (uint8_t *)malloc((size_t)lw[l] * lh[l]) : 0;
    if (l == 0) memcpy(lev[0], gray, (size_t)w * h);
    else oracle_orb_resize(lev[l - 1], lw[l - 1], lh[l - 1], lev[l], lw[l], lh[l]);
    if (want_images) oracle_orb_blur(lev[l], lw[l], lh[l], blur[l]);
    if (levels_out) memcpy(levels_out + off, lev[l], (size_t)lw[l] * lh[l]);
    if (blurred_out && want_images) memcpy(blurred_out + off, blur[l], (size_t)lw[l] * lh[l]);
    off += (size_t)lw[l] * lh[l];
  }
  score = (uint8_t *)malloc((size_t)w * h);
  for (l = 0; l < ORB_LEVELS; l++) {
    const int W = lw[l], H = lh[l];
    o_kp *k = all + n_all;
    int n = 0, x, y;
    float sf = o_get_scale(l);
    oracle_orb_fast_scores(lev[l], W, H, fast_threshold, score);
    for (y = 3; y < H - 3; y++)          /* non-maximum suppression, strict, 3x3; scan order = cv::FAST's output order */
      for (x = 3; x < W - 3; x++) {
        int s = score[(size_t)y * W + x];
        if (!s) continue;
        if (s > score[(size_t)y * W + x + 1] && s > score[(size_t)y * W + x - 1] && s > score[(size_t)(y - 1) * W + x - 1] &&
            s > score[(size_t)(y - 1) * W + x] && s > score[(size_t)(y - 1) * W + x + 1] && s > score[(size_t)(y + 1) * W + x - 1] &&
            s > score[(size_t)(y + 1) * W + x] && s > score[(size_t)(y + 1) * W + x + 1]) {
          if (x >= ORB_EDGE && x < W - ORB_EDGE && y >= ORB_EDGE && y < H - ORB_EDGE) {   /* runByImageBorder */
            k[n].x = (float)x; k[n].y = (float)y; k[n].response = (float)s; k[n].level = l; k[n].order = order++; k[n].angle = -1;
            n++;
          }
        }
      }
    n = o_retain_best(k, n, 2 * nper[l]);
    for (i = 0; i < n; i++) k[i].response = o_harris(lev[l], W, k[i].x, k[i].y);
    n = o_retain_best(k, n, nper[l]);
    for (i = 0; i < n; i++) {
      k[i].angle = o_ic_angle(lev[l], W, k[i].x, k[i].y, umax);
      if (l != 0) { k[i].x *= sf; k[i].y *= sf; }      /* keypoint->pt *= scale (aorb.cpp:893-899) */
    }
    n_all += n;
  }
  free(score);
  if (!want_images) for (l = 0; l < ORB_LEVELS; l++) free(lev[l]);
  return n_all;
}

/* VideoDynamicAdaptedFeatureDetector::detectImpl around DetectorAdjuster("AORB", thresh) (src/feature_adjuster.cpp:107-186), frame
 * after frame with ONE detector object, i.e. the threshold state carried along: every detection is really run
 * (AorbFeatureDetector::detect at static_cast<int>(thresh_)) and counted.  thr_out[f] = the threshold of the LAST detection of
 * frame f (whose key points the caller receives); *thresh_io = DetectorAdjuster::thresh_ before / after.               */
void oracle_orb_adjust_thresholds(const uint8_t *gray, int n_frames, int w, int h, int nfeatures, double *thresh_io, double min_thresh,
                                  double max_thresh, double inc, double dec, int min_features, int max_features, int max_iters,
                                  int *thr_out, int *count_out) {
  o_kp *all = (o_kp *)malloc(sizeof(o_kp) * (size_t)w * h / 4 + 64);
  uint8_t *lev[ORB_LEVELS], *blur[ORB_LEVELS];
  int lw[ORB_LEVELS], lh[ORB_LEVELS], f;
  double thresh = *thresh_io;
  for (f = 0; f < n_frames; f++) {
    int iter = max_iters, t, n;
    do {
      t = (int)thresh;
      n = o_orb_detect(gray + (size_t)f * w * h, w, h, t, nfeatures, all, lev, blur, lw, lh, 0, 0, 0);
      if (n < min_features) { thresh *= dec; if (thresh < min_thresh) thresh = min_thresh; }
      else if (n > max_features) { thresh *= inc; if (thresh > max_thresh) thresh = max_thresh; break; }
      else break;
      iter--;
    } while (iter > 0 && thresh > min_thresh && thresh < max_thresh);
    thr_out[f] = t < 0 ? 0 : (t > 255 ? 255 : t);
    if (count_out) count_out[f] = n;
  }
  *thresh_io = thresh;
  free(all);
}

int oracle_orb_extract(const uint8_t *gray, int w, int h, const float *depth, int dstride, int fast_threshold, int nfeatures,
                       int max_keypoints, float *kp_xy, float *kp_meta, uint8_t *desc, uint8_t *levels_out, uint8_t *blurred_out) {
  uint8_t *lev[ORB_LEVELS], *blur[ORB_LEVELS];
  int lw[ORB_LEVELS], lh[ORB_LEVELS], l, n_all, i, n_out;
  o_kp *all = (o_kp *)malloc(sizeof(o_kp) * (size_t)w * h / 4 + 64);
  n_all = o_orb_detect(gray, w, h, fast_threshold, nfeatures, all, lev, blur, lw, lh, 1, levels_out, blurred_out);
  /* Node: removeDepthless, retainBest(max_keypoints) + resize */
  if (depth) {
    int m = 0;
    for (i = 0; i < n_all; i++) {
      float px = all[i].x, py = all[i].y, Z;
      if (px >= w || px < 0 || py >= h || py < 0 || px != px || py != py) continue;
      { int ry = (int)roundf(py), rx = (int)roundf(px); if (ry >= h) ry = h - 1; if (rx >= w) rx = w - 1; Z = depth[(size_t)ry * dstride + rx]; }
      if (Z != Z) continue;
      all[m++] = all[i];
    }
    n_all = m;
  }
  if (n_all > max_keypoints) {
    qsort(all, (size_t)n_all, sizeof *all, o_cmp_kp);
    n_all = max_keypoints;
  }
  /* OrbDescriptorExtractor::compute: border filter on level-0 coordinates, clustering by octave (order inside an octave
   * kept), descriptors on the blurred levels at pt * (1 / scale), coordinates scaled back */
  n_out = 0;
  for (l = 0; l < ORB_LEVELS; l++) {
    float sf = o_get_scale(l), inv = 1 / sf;
    for (i = 0; i < n_all; i++) {
      float px, py;
      if (all[i].level != l) continue;
      px = all[i].x; py = all[i].y;
      if (!(px >= ORB_EDGE && px < w - ORB_EDGE && py >= ORB_EDGE && py < h - ORB_EDGE)) continue;
      if (l != 0) { px *= inv; py *= inv; }
      o_descriptor(blur[l], lw[l], px, py, all[i].angle, desc + 32 * (size_t)n_out);
      if (l != 0) { px *= sf; py *= sf; }
      kp_xy[2 * n_out] = px; kp_xy[2 * n_out + 1] = py;
      kp_meta[4 * n_out] = all[i].response; kp_meta[4 * n_out + 1] = all[i].angle; kp_meta[4 * n_out + 2] = (float)l;
      kp_meta[4 * n_out + 3] = ORB_PATCH * sf;
      n_out++;
    }
  }
  for (l = 0; l < ORB_LEVELS; l++) { free(lev[l]); free(blur[l]); }
  free(all);
  return n_out;
}
