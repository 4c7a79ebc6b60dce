/* oracle/front_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the part of Node::detect3DLines that follows LSD
 * (src/line/lineslam.cpp:213-357): length filter, depth sampling along each 2D segment, per-point
 * covariance, RANSAC 3D line fit, Sobel, line gradient direction, MSLD descriptor, maximum-
 * likelihood 3D line (levmar) and its covariance.  Each function cites the lines it follows.
 *
 * Parity status: the reference's src/line sources cannot be compiled here
 * (OpenCV 2.4, Eigen, Armadillo, PCL absent), and its tests hold no vectors for this stage, so
 * this file is "parity unpinned" against the reference binary: it is a restatement from the
 * source text.  Third-party pieces it has to restate from their published algorithms:
 *   cv::SVD (3x3 symmetric)      -> Jacobi eigen-decomposition, o_linalg.h (the oracle's own statement)
 *   cv::Mat::inv (LU)            -> OpenCV's LU order, o_linalg.h
 *   cv::Sobel ksize 5, cv::LineIterator, cv::clipLine  (OpenCV 2.4 imgproc / core drawing)
 *   dlevmar_dif                  (external/levmar-2.6/lm_core.c:438-846, misc_core.c:137-171) --
 *                                 restated below; cross-checked against the compiled levmar where
 *                                 that is available (tests/test_oracle_front.py)
 *   rand()                       -> o_rand31 counter-based generator (the reference's stream is
 *                                 unseeded and racy under OpenMP, SURVEY.md hard part C)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/linefront.h"
#include "o_linalg.h"   /* the oracle's own Jacobi / LU / counter generator (NOT the product's lf_linalg.h) */

#define O_EPS 1e-10 /* lineslam.h:37 */

typedef struct {   /* RandomPoint3d, lineslam.h:41-82 */
  double pos[3], cov[9], W_sqrt[3], DU[9];
} orpt;

/* RandomPoint3d(pos, cov) ctor, lineslam.h:59-81: SVD of the covariance -> W_sqrt, DU */
static void o_make_rpt(const double pos[3], const double cov[9], orpt *o) {
  double A[9], V[9], w[3];
  int i, j;
  for (i = 0; i < 3; i++) o->pos[i] = pos[i];
  for (i = 0; i < 9; i++) { o->cov[i] = cov[i]; A[i] = cov[i]; }
  o_jacobi(3, A, V, w);
  for (i = 0; i < 3; i++) o->W_sqrt[i] = sqrt(w[i]);
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++) o->DU[3 * i + j] = (1 / o->W_sqrt[i]) * V[3 * j + i]; /* D * U^T */
}

/* depthStdDev + compPt3dCov, utils.cpp:671-722 (time_diff_sec = 0 with MODEL_ASYNCH=0) */
static void o_comp_pt3d_cov(const double pt[3], double f, const lf_params *P, orpt *o) {
  double sig = P->stdev_sample_pt_imgline;
  double c1 = P->depth_stdev_coeff_c1, c2 = P->depth_stdev_coeff_c2 + 0.0 * 0.5, c3 = P->depth_stdev_coeff_c3;
  double sz = c1 * pt[2] * pt[2] + c2 * pt[2] + c3;
  double s2 = sig * sig, sz2 = sz * sz;
  double j00 = pt[2] / f, j02 = pt[0] / pt[2], j11 = pt[2] / f, j12 = pt[1] / pt[2];
  double cov[9];
  /* (J * S) * J^T with J = [j00 0 j02; 0 j11 j12; 0 0 1], S = diag(s2, s2, sz2) */
  cov[0] = (j00 * s2) * j00 + (j02 * sz2) * j02;
  cov[1] = (j02 * sz2) * j12;
  cov[2] = (j02 * sz2);
  cov[3] = (j12 * sz2) * j02;
  cov[4] = (j11 * s2) * j11 + (j12 * sz2) * j12;
  cov[5] = (j12 * sz2);
  cov[6] = sz2 * j02;
  cov[7] = sz2 * j12;
  cov[8] = sz2;
  o_make_rpt(pt, cov, o);
}

/* mah_dist3d_pt_line (fast version), utils.cpp:761-822 */
static double o_mah_dist(const orpt *pt, const double q1[3], const double q2[3]) {
  const double *c = pt->DU;
  double x1 = pt->pos[0], x2 = pt->pos[1], x3 = pt->pos[2];
  double xa = q1[0], ya = q1[1], za = q1[2], xb = q2[0], yb = q2[1], zb = q2[2];
  double a0 = c[0] * (x1 - xa) + c[1] * (x2 - ya) + c[2] * (x3 - za);
  double a1 = c[3] * (x1 - xa) + c[4] * (x2 - ya) + c[5] * (x3 - za);
  double a2 = c[6] * (x1 - xa) + c[7] * (x2 - ya) + c[8] * (x3 - za);
  double b0 = c[0] * (x1 - xb) + c[1] * (x2 - yb) + c[2] * (x3 - zb);
  double b1 = c[3] * (x1 - xb) + c[4] * (x2 - yb) + c[5] * (x3 - zb);
  double b2 = c[6] * (x1 - xb) + c[7] * (x2 - yb) + c[8] * (x3 - zb);
  double t1 = a0 * b1 - a1 * b0, t2 = a0 * b2 - a2 * b0, t3 = a1 * b2 - a2 * b1;
  double t4 = c[0] * (x1 - xa) - c[0] * (x1 - xb) + c[1] * (x2 - ya) - c[1] * (x2 - yb) + c[2] * (x3 - za) - c[2] * (x3 - zb);
  double t5 = c[3] * (x1 - xa) - c[3] * (x1 - xb) + c[4] * (x2 - ya) - c[4] * (x2 - yb) + c[5] * (x3 - za) - c[5] * (x3 - zb);
  double t6 = c[6] * (x1 - xa) - c[6] * (x1 - xb) + c[7] * (x2 - ya) - c[7] * (x2 - yb) + c[8] * (x3 - za) - c[8] * (x3 - zb);
  return sqrt((t1 * t1 + t2 * t2 + t3 * t3) / (t4 * t4 + t5 * t5 + t6 * t6));
}
double oracle_mah_dist(const double pos[3], const double DU[9], const double q1[3], const double q2[3]) {
  orpt p; int i;
  for (i = 0; i < 3; i++) p.pos[i] = pos[i];
  for (i = 0; i < 9; i++) p.DU[i] = DU[i];
  return o_mah_dist(&p, q1, q2);
}

/* verify3dLine (RandomPoint3d version), utils.cpp:570-624 */
static int o_verify3d(const orpt *pts, const int *idx, int n, const double A[3], const double B[3],
                      const lf_params *P) {
  int nCells = P->num_cells_lineseg_range, i, k, i1 = 0, i2 = 0;
  int cells[64];
  double minv = 100, maxv = -100, AB[3], C[3], D[3], mid[3], cd, sum = 0;
  if (nCells > 64) nCells = 64;
  for (i = 0; i < nCells; i++) cells[i] = 0;
  for (k = 0; k < 3; k++) { AB[k] = B[k] - A[k]; mid[k] = (A[k] + B[k]) * 0.5; }
  for (i = 0; i < n; i++) {
    const double *x = pts[idx[i]].pos;
    double d = (x[0] - A[0]) * AB[0] + (x[1] - A[1]) * AB[1] + (x[2] - A[2]) * AB[2];
    if (d < minv) { minv = d; i1 = i; }
    if (d > maxv) { maxv = d; i2 = i; }
  }
  { /* projectPt3d2Ln3d(P, mid, drct): A=mid, B=mid+drct, A + (AB.AP/(AB.AB))*AB  (utils.cpp:495-504) */
    const double *x1 = pts[idx[i1]].pos, *x2 = pts[idx[i2]].pos;
    double Bm[3], ab[3], ap[3], s;
    for (k = 0; k < 3; k++) { Bm[k] = mid[k] + AB[k]; ab[k] = Bm[k] - mid[k]; }
    for (k = 0; k < 3; k++) ap[k] = x1[k] - mid[k];
    s = (ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2]) / (ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    for (k = 0; k < 3; k++) C[k] = mid[k] + s * ab[k];
    for (k = 0; k < 3; k++) ap[k] = x2[k] - mid[k];
    s = (ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2]) / (ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    for (k = 0; k < 3; k++) D[k] = mid[k] + s * ab[k];
  }
  cd = sqrt((D[0] - C[0]) * (D[0] - C[0]) + (D[1] - C[1]) * (D[1] - C[1]) + (D[2] - C[2]) * (D[2] - C[2]));
  if (cd < O_EPS) return 0;
  for (i = 0; i < n; i++) {
    const double *x = pts[idx[i]].pos;
    double lambda = fabs(((x[0] - C[0]) * (D[0] - C[0]) + (x[1] - C[1]) * (D[1] - C[1]) + (x[2] - C[2]) * (D[2] - C[2])) / cd / cd);
    if (lambda >= 1) cells[nCells - 1] += 1;
    else {
      unsigned int c = (unsigned int)floor(lambda * 10); /* hard-coded 10, utils.cpp:608 */
      if (c < 64) cells[c] += 1;
    }
  }
  for (i = 0; i < nCells; i++) if (cells[i] > 0) sum = sum + 1;
  return sum / nCells > P->ratio_support_pts_on_line;
}

/* computeLine3d_svd (index version), utils.cpp:471-493: mean + principal direction.  The n x 3
 * SVD's first right-singular vector == eigenvector of the largest eigenvalue of P^T P.          */
static void o_line3d_svd(const orpt *pts, const int *idx, int n, double mean[3], double drct[3]) {
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V[9], w[3];
  int i, k, l;
  mean[0] = mean[1] = mean[2] = 0;
  for (i = 0; i < n; i++) for (k = 0; k < 3; k++) mean[k] = mean[k] + pts[idx[i]].pos[k];
  for (k = 0; k < 3; k++) mean[k] = mean[k] * (1.0 / n);
  for (i = 0; i < n; i++) {
    double d[3];
    for (k = 0; k < 3; k++) d[k] = pts[idx[i]].pos[k] - mean[k];
    for (k = 0; k < 3; k++) for (l = 0; l < 3; l++) S[3 * k + l] += d[k] * d[l];
  }
  o_jacobi(3, S, V, w);
  for (k = 0; k < 3; k++) drct[k] = V[3 * k + 0];
}

/* random_unique(begin, end, 2), utils.h:49-60 with the counter-based generator */
static void o_random_unique(int *v, int n, int num, uint64_t seed, uint64_t stream, uint64_t *ctr) {
  int b = 0, left = n;
  while (num--) {
    int r = b + (int)(o_rand31(seed, stream, (*ctr)++) % (uint32_t)left);
    int t = v[b]; v[b] = v[r]; v[r] = t;
    ++b; --left;
  }
}

/* extract3dline_mahdist, utils.cpp:343-427.  inl: out, indices of the supporting points. */
static int o_extract3dline(const orpt *pts, int n, const lf_params *P, uint64_t seed, uint64_t stream,
                           double A[3], double B[3], int *inl) {
  int maxIter = P->ransac_iters_extract_line, half = (int)(n * (n - 1) * 0.5);
  double thr = P->pt2line_mahdist_extractline;
  int indexes[128], cur[128], best[128], nbest = 0, iter, i, k, bestA = 0, bestB = 0;
  uint64_t ctr = 0;
  if (half < maxIter) maxIter = half;
  for (i = 0; i < n; i++) indexes[i] = i;
  for (iter = 0; iter < maxIter; iter++) {
    int nc = 0;
    const orpt *a, *b;
    double dn;
    o_random_unique(indexes, n, 2, seed, stream, &ctr);
    a = &pts[indexes[0]]; b = &pts[indexes[1]];
    dn = sqrt((b->pos[0] - a->pos[0]) * (b->pos[0] - a->pos[0]) + (b->pos[1] - a->pos[1]) * (b->pos[1] - a->pos[1]) +
              (b->pos[2] - a->pos[2]) * (b->pos[2] - a->pos[2]));
    if (dn < O_EPS) continue;
    for (i = 0; i < n; i++) if (o_mah_dist(&pts[i], a->pos, b->pos) < thr) cur[nc++] = i;
    if (nc > nbest) {
      if (o_verify3d(pts, cur, nc, a->pos, b->pos, P)) {
        for (i = 0; i < nc; i++) best[i] = cur[i];
        nbest = nc; bestA = indexes[0]; bestB = indexes[1];
      }
    }
    if (nbest > n * 0.9) break;
  }
  if (nbest >= 2) {
    double m[3], d[3], minv = 100, maxv = -100;
    int e1 = 0, e2 = 0;
    for (k = 0; k < 3; k++) { m[k] = (pts[bestA].pos[k] + pts[bestB].pos[k]) * 0.5; d[k] = pts[bestB].pos[k] - pts[bestA].pos[k]; }
    for (;;) {
      double tm[3], td[3], q2[3];
      int nc = 0;
      o_line3d_svd(pts, best, nbest, tm, td);
      for (k = 0; k < 3; k++) q2[k] = tm[k] + td[k];
      for (i = 0; i < n; i++) if (o_mah_dist(&pts[i], tm, q2) < thr) cur[nc++] = i;
      if (nc > nbest) {
        for (i = 0; i < nc; i++) best[i] = cur[i];
        nbest = nc;
        for (k = 0; k < 3; k++) { m[k] = tm[k]; d[k] = td[k]; }
      } else break;
    }
    for (i = 0; i < nbest; i++) {
      const double *x = pts[best[i]].pos;
      double dp = (x[0] - m[0]) * d[0] + (x[1] - m[1]) * d[1] + (x[2] - m[2]) * d[2];
      if (dp < minv) { minv = dp; e1 = i; }
      if (dp > maxv) { maxv = dp; e2 = i; }
    }
    for (k = 0; k < 3; k++) { A[k] = pts[best[e1]].pos[k]; B[k] = pts[best[e2]].pos[k]; }
  } else {
    for (k = 0; k < 3; k++) A[k] = B[k] = 0; /* default-constructed cv::Point3d */
  }
  for (i = 0; i < nbest; i++) inl[i] = best[i];
  return nbest;
}

/* ------------------------------------------------------------------------------ cv::Sobel
 * ksize 5, CV_8U -> CV_64F, BORDER_REFLECT_101 (lineslam.cpp:311-314): separable kernels
 * deriv [-1 -2 0 2 1], smooth [1 4 6 4 1]; all values are exact integers.                    */
static int o_reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
  return i;
}
void oracle_sobel5(const uint8_t *g, int stride, int w, int h, double *gx, double *gy) {
  static const int kd[5] = {-1, -2, 0, 2, 1}, ks[5] = {1, 4, 6, 4, 1};
  int *rx = (int *)malloc(sizeof(int) * (size_t)w * h), *ry = (int *)malloc(sizeof(int) * (size_t)w * h);
  int x, y, k;
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int sd = 0, ss = 0;
      for (k = 0; k < 5; k++) {
        int v = g[(size_t)y * stride + o_reflect101(x + k - 2, w)];
        sd += kd[k] * v; ss += ks[k] * v;
      }
      rx[y * w + x] = sd; ry[y * w + x] = ss;
    }
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int sx = 0, sy = 0;
      for (k = 0; k < 5; k++) {
        int yy = o_reflect101(y + k - 2, h);
        sx += ks[k] * rx[yy * w + x];   /* d/dx: derivative along x, smoothing along y */
        sy += kd[k] * ry[yy * w + x];   /* d/dy */
      }
      gx[y * w + x] = (double)sx; gy[y * w + x] = (double)sy;
    }
  free(rx); free(ry);
}

/* cvRound (SSE2 cvtsd2si: round half to even) used by Point2d -> Point (lineslam.cpp:529) */
static int o_cvround(double v) { return (int)nearbyint(v); }

/* cv::clipLine(Size, Point&, Point&), OpenCV 2.4 core/src/drawing.cpp */
static int o_clipline(int w, int h, int64_t *px1, int64_t *py1, int64_t *px2, int64_t *py2) {
  int64_t x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2, right = w - 1, bottom = h - 1;
  int c1, c2;
  if (w <= 0 || h <= 0) return 0;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (a - y1) * (x2 - x1) / (y2 - y1); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
    if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (a - y2) * (x2 - x1) / (y2 - y1); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) { a = c1 == 1 ? 0 : right; y1 += (a - x1) * (y2 - y1) / (x2 - x1); x1 = a; c1 = 0; }
      if (c2) { a = c2 == 1 ? 0 : right; y2 += (a - x2) * (y2 - y1) / (x2 - x1); x2 = a; c2 = 0; }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
  }
  return (c1 | c2) == 0;
}

/* FrameLine::getGradient, lineslam.cpp:527-537, with cv::LineIterator(img, p, q, 8) (OpenCV 2.4
 * drawing.cpp): Bresenham, count = max(|dx|,|dy|)+1, minor-axis step when err < 0.              */
void oracle_line_gradient(const double *gx, const double *gy, int w, int h, const double p[2],
                          const double q[2], double r[2]) {
  int64_t x1 = o_cvround(p[0]), y1 = o_cvround(p[1]), x2 = o_cvround(q[0]), y2 = o_cvround(q[1]);
  double xs = 0, ys = 0, len;
  int count = 0;
  if ((uint64_t)x1 >= (uint64_t)w || (uint64_t)x2 >= (uint64_t)w || (uint64_t)y1 >= (uint64_t)h ||
      (uint64_t)y2 >= (uint64_t)h) {
    if (!o_clipline(w, h, &x1, &y1, &x2, &y2)) { count = -1; }
  }
  if (count == 0) {
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int sxs = dx < 0 ? -1 : 1, sys = dy < 0 ? -1 : 1;   /* step signs (leftToRight = false) */
    int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    int steep = ady > adx;
    int major = steep ? ady : adx, minor = steep ? adx : ady;
    int err = major - (minor + minor), plusDelta = major + major, minusDelta = -(minor + minor);
    int x = (int)x1, y = (int)y1, i;
    count = major + 1;
    for (i = 0; i < count; i++) {
      int mask;
      xs += gx[y * w + x];
      ys += gy[y * w + x];
      mask = err < 0 ? -1 : 0;
      err += minusDelta + (plusDelta & mask);
      if (steep) { y += sys; if (mask) x += sxs; }
      else { x += sxs; if (mask) y += sys; }
    }
  }
  len = sqrt(xs * xs + ys * ys);
  r[0] = xs / len;
  r[1] = ys / len;
}

/* computeSubPSR, utils.cpp:1510-1542 */
static int o_subpsr(const double *xG, const double *yG, double px, double py, double s, const double g[2],
                    int width, int height, double vs[4]) {
  double tl_x = floor(px - s / 2), tl_y = floor(py - s / 2);
  double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  int x, y;
  if (tl_x < 0 || tl_y < 0 || tl_x + s + 1 > width || tl_y + s + 1 > height) return 0;
  if (!(tl_x == tl_x) || !(tl_y == tl_y)) return 0; /* NaN direction (flat image): the reference indexes out of range */
  for (y = (int)tl_y; y < tl_y + s; ++y)
    for (x = (int)tl_x; x < tl_x + s; ++x) {
      double tmp1 = xG[y * width + x] * g[0] + yG[y * width + x] * g[1];
      double tmp2 = xG[y * width + x] * (-g[1]) + yG[y * width + x] * g[0];
      if (tmp1 >= 0) v1 = v1 + tmp1; else v2 = v2 - tmp1;
      if (tmp2 >= 0) v3 = v3 + tmp2; else v4 = v4 - tmp2;
    }
  vs[0] = v1; vs[1] = v2; vs[2] = v3; vs[3] = v4;
  return 1;
}

/* computeMSLD, utils.cpp:1544-1610.  Returns the number of valid samples. */
int oracle_msld(const double *xG, const double *yG, int width, int height, const double p[2],
                const double q[2], const double r[2], double step, uint64_t seed, uint64_t stream,
                double des[72]) {
  int s = (int)(5 * width / 800.0);
  double len = sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]));
  static const double gauss[9] = {0.24142, 0.30046, 0.35127, 0.38579, 0.39804, 0.38579, 0.35127, 0.30046, 0.24142};
  double *GDM = (double *)malloc(sizeof(double) * 36 * (size_t)(2 * (int)len + 8));
  int n = 0, i, j, cap = 2 * (int)len + 8;
  double nrm;
  for (i = 0; i * step < len && n < cap; ++i) {
    double col[36], t = (i * step / len);
    double ptx = p[0] + (q[0] - p[0]) * t, pty = p[1] + (q[1] - p[1]) * t;
    int fail = 0;
    for (j = -4; j <= 4; ++j) {
      double vs[4];
      if (o_subpsr(xG, yG, ptx + r[0] * (j * s), pty + r[1] * (j * s), (double)s, r, width, height, vs)) {
        col[(j + 4) * 4 + 0] = vs[0]; col[(j + 4) * 4 + 1] = vs[1];
        col[(j + 4) * 4 + 2] = vs[2]; col[(j + 4) * 4 + 3] = vs[3];
      } else { fail = 1; break; }
    }
    if (fail) continue;
    memcpy(GDM + 36 * (size_t)n, col, sizeof col);
    n++;
  }
  if (n == 0) {
    for (i = 0; i < 72; i++) des[i] = (double)o_rand31(seed, stream, 1000 + (uint64_t)i);
    free(GDM);
    return 0;
  }
  for (i = 0; i < 36; ++i) {
    double sum = 0, sum2 = 0, mean, sd;
    for (j = 0; j < n; ++j) {
      double v = GDM[36 * (size_t)j + i] * gauss[i / 4];
      sum += v;
      sum2 += v * v;
    }
    mean = sum / n;
    sd = sqrt(sum2 / n - mean * mean);
    des[i] = mean;
    des[i + 36] = sd;
  }
  /* cv::Mat / scalar == multiply by (1/scalar); cv::norm == sqrt(sum of squares) */
  nrm = 0; for (i = 0; i < 36; i++) nrm += des[i] * des[i];
  nrm = 1.0 / sqrt(nrm); for (i = 0; i < 36; i++) des[i] = des[i] * nrm;
  nrm = 0; for (i = 36; i < 72; i++) nrm += des[i] * des[i];
  nrm = 1.0 / sqrt(nrm); for (i = 36; i < 72; i++) des[i] = des[i] * nrm;
  for (i = 0; i < 72; i++) if (des[i] > 0.4) des[i] = 0.4;
  nrm = 0; for (i = 0; i < 72; i++) nrm += des[i] * des[i];
  nrm = 1.0 / sqrt(nrm); for (i = 0; i < 72; i++) des[i] = des[i] * nrm;
  free(GDM);
  return n;
}

/* --------------------------------------------------------------------------------------------
 * dlevmar_dif restated (external/levmar-2.6/lm_core.c:438-846; forward differences
 * misc_core.c:137-171; constants levmar.h:95-100: LM_INIT_MU 1e-3, LM_DIFF_DELTA 1e-6,
 * EPSILON 1e-12, ONE_THIRD 0.3333333334).  Measurement vector x == 0 as in the reference's calls.
 * Linear solver: AX_EQ_B_LU in LAPACK's dgetf2 / dgetrs order (o_ax_eq_b_lu).  m <= 8.            */
typedef void (*o_lm_func)(const double *p, double *hx, int m, int n, void *adata);

/* AX_EQ_B_LU (external/levmar-2.6/Axb_core.c:738-830; the reference defines HAVE_LAPACK, levmar.h:31, so lm_core.c:706
 * takes this one): A is copied COLUMN-major (Axb_core.c:787-793), factored by LAPACK's dgetrf and solved by dgetrs.
 * LAPACK is a system library of the reference, not part of its tree: what follows transcribes the published reference
 * implementation (netlib LAPACK 3.x + reference BLAS), routine by routine -- for the 6 x 6 / 7 x 7 systems of this
 * path dgetrf takes its unblocked branch (block size 64 > n), i.e. dgetf2:
 *   idamax  first index of the largest |value|
 *   dswap   whole rows, all n columns
 *   dscal   by the reciprocal of the pivot when |pivot| >= sfmin, else element-wise division
 *   dger    A := A - x y^T, column by column, skipping columns with y(j) == 0
 *   dlaswp  row interchanges on b; dtrsm 'L','L','N','U' then dtrsm 'L','U','N','N', both column oriented, skipping
 *           zero b(k), the second dividing by the diagonal.
 * (OpenBLAS / ATLAS / MKL builds of LAPACK order these sums differently)                  */
static int o_dgetf2(int n, double *a /* column-major, lda = n */, int *ipiv) {
  const double sfmin = DBL_MIN; /* dlamch('S') */
  int info = 0, i, j, k;
  for (j = 0; j < n; ++j) {
    int jp = j;
    double dmax = fabs(a[j + j * n]);
    for (i = j + 1; i < n; ++i)                       /* idamax */
      if (fabs(a[i + j * n]) > dmax) { dmax = fabs(a[i + j * n]); jp = i; }
    ipiv[j] = jp;
    if (a[jp + j * n] != 0.0) {
      if (jp != j)                                     /* dswap */
        for (k = 0; k < n; ++k) { double t = a[j + k * n]; a[j + k * n] = a[jp + k * n]; a[jp + k * n] = t; }
      if (j < n - 1) {
        if (fabs(a[j + j * n]) >= sfmin) {             /* dscal */
          double r = 1.0 / a[j + j * n];
          for (i = j + 1; i < n; ++i) a[i + j * n] = r * a[i + j * n];
        } else
          for (i = j + 1; i < n; ++i) a[i + j * n] = a[i + j * n] / a[j + j * n];
      }
    } else if (info == 0)
      info = j + 1;
    if (j < n - 1)                                     /* dger, alpha = -1 */
      for (k = j + 1; k < n; ++k)
        if (a[j + k * n] != 0.0) {
          double temp = -1.0 * a[j + k * n];
          for (i = j + 1; i < n; ++i) a[i + k * n] = a[i + k * n] + a[i + j * n] * temp;
        }
  }
  return info;
}
static void o_dgetrs(int n, const double *a, const int *ipiv, double *b) {
  int i, k;
  for (i = 0; i < n; ++i)                              /* dlaswp, forward */
    if (ipiv[i] != i) { double t = b[i]; b[i] = b[ipiv[i]]; b[ipiv[i]] = t; }
  for (k = 0; k < n; ++k)                              /* dtrsm: L, unit diagonal */
    if (b[k] != 0.0)
      for (i = k + 1; i < n; ++i) b[i] = b[i] - b[k] * a[i + k * n];
  for (k = n - 1; k >= 0; --k)                         /* dtrsm: U, non-unit diagonal */
    if (b[k] != 0.0) {
      b[k] = b[k] / a[k + k * n];
      for (i = 0; i < k; ++i) b[i] = b[i] - b[k] * a[i + k * n];
    }
}
static int o_ax_eq_b_lu(const double *A, const double *B, double *x, int m) {
  double a[64];
  int ipiv[8], i, j;
  for (i = 0; i < m; i++) {
    for (j = 0; j < m; j++) a[i + j * m] = A[i * m + j];
    x[i] = B[i];
  }
  if (o_dgetf2(m, a, ipiv) != 0) return 0;             /* "singular matrix A for dgetrf" */
  o_dgetrs(m, a, ipiv, x);
  return 1;
}
/* dlevmar_L2nrmxmy with x == 0 values (misc_core.c, LEVMAR_L2NRMXMY; lm_core.c:555 and :743 call it, the plain loops
 * there are compiled out): e = 0 - y and ||e||^2 with FOUR running sums over blocks of eight taken from the top of the
 * vector downwards, then the remainder by the fall-through switch, returned as sum0+sum1+sum2+sum3.                  */
static double o_l2nrmxmy(double *e, const double *y, int n) {
  double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0, sum3 = 0.0;
  const int blockn = (n >> 3) << 3;
  int i;
  for (i = blockn - 1; i > 0; i -= 8) {
    e[i] = 0.0 - y[i]; sum0 += e[i] * e[i];
    e[i - 1] = 0.0 - y[i - 1]; sum1 += e[i - 1] * e[i - 1];
    e[i - 2] = 0.0 - y[i - 2]; sum2 += e[i - 2] * e[i - 2];
    e[i - 3] = 0.0 - y[i - 3]; sum3 += e[i - 3] * e[i - 3];
    e[i - 4] = 0.0 - y[i - 4]; sum0 += e[i - 4] * e[i - 4];
    e[i - 5] = 0.0 - y[i - 5]; sum1 += e[i - 5] * e[i - 5];
    e[i - 6] = 0.0 - y[i - 6]; sum2 += e[i - 6] * e[i - 6];
    e[i - 7] = 0.0 - y[i - 7]; sum3 += e[i - 7] * e[i - 7];
  }
  i = blockn;
  if (i < n) {
    switch (n - i) {
      case 7: e[i] = 0.0 - y[i]; sum0 += e[i] * e[i]; ++i; /* fall through */
      case 6: e[i] = 0.0 - y[i]; sum1 += e[i] * e[i]; ++i; /* fall through */
      case 5: e[i] = 0.0 - y[i]; sum2 += e[i] * e[i]; ++i; /* fall through */
      case 4: e[i] = 0.0 - y[i]; sum3 += e[i] * e[i]; ++i; /* fall through */
      case 3: e[i] = 0.0 - y[i]; sum0 += e[i] * e[i]; ++i; /* fall through */
      case 2: e[i] = 0.0 - y[i]; sum1 += e[i] * e[i]; ++i; /* fall through */
      case 1: e[i] = 0.0 - y[i]; sum2 += e[i] * e[i];
    }
  }
  return sum0 + sum1 + sum2 + sum3;
}
/* levmar's OWN LU, used when levmar-2.6 is built without LAPACK (Ax_eq_b_LU_noLapack, external/levmar-2.6/Axb_core.c:1123-1277):
 * Crout's method with implicit row scaling (work[i] = 1 / largest |entry| of row i), pivot = LAST row of the maximal scaled
 * |sum| (`>=`), a zero pivot replaced by LM_REAL_EPSILON (DBL_EPSILON, levmar.h), forward substitution skipping leading zeros
 * of the right-hand side, back substitution with a division.  The reference DEFINES HAVE_LAPACK, so this is not what it runs;
 * it is what lets the restatement of dlevmar_dif be held bit for bit against the reference's levmar compiled from nothing but
 * its own files (oracle/_ref/liblevmar_nolapack_ref.so, tests/test_mle_golden_cpu.py).                                  */
static int o_ax_eq_b_lu_nolapack(const double *A, const double *B, double *x, int m) {
  double a[64], work[8], max, sum, tmp;
  int idx[8], i, j, k, maxi = -1;
  for (i = 0; i < m * m; i++) a[i] = A[i];
  for (i = 0; i < m; i++) x[i] = B[i];
  for (i = 0; i < m; ++i) {
    max = 0.0;
    for (j = 0; j < m; ++j)
      if ((tmp = fabs(a[i * m + j])) > max) max = tmp;
    if (max == 0.0) return 0;
    work[i] = 1.0 / max;
  }
  for (j = 0; j < m; ++j) {
    for (i = 0; i < j; ++i) {
      sum = a[i * m + j];
      for (k = 0; k < i; ++k) sum -= a[i * m + k] * a[k * m + j];
      a[i * m + j] = sum;
    }
    max = 0.0;
    for (i = j; i < m; ++i) {
      sum = a[i * m + j];
      for (k = 0; k < j; ++k) sum -= a[i * m + k] * a[k * m + j];
      a[i * m + j] = sum;
      if ((tmp = work[i] * fabs(sum)) >= max) { max = tmp; maxi = i; }
    }
    if (j != maxi) {
      for (k = 0; k < m; ++k) { tmp = a[maxi * m + k]; a[maxi * m + k] = a[j * m + k]; a[j * m + k] = tmp; }
      work[maxi] = work[j];
    }
    idx[j] = maxi;
    if (a[j * m + j] == 0.0) a[j * m + j] = DBL_EPSILON;
    if (j != m - 1) {
      tmp = 1.0 / (a[j * m + j]);
      for (i = j + 1; i < m; ++i) a[i * m + j] *= tmp;
    }
  }
  for (i = k = 0; i < m; ++i) {
    j = idx[i];
    sum = x[j];
    x[j] = x[i];
    if (k != 0)
      for (j = k - 1; j < i; ++j) sum -= a[i * m + j] * x[j];
    else if (sum != 0.0)
      k = i + 1;
    x[i] = sum;
  }
  for (i = m - 1; i >= 0; --i) {
    sum = x[i];
    for (j = i + 1; j < m; ++j) sum -= a[i * m + j] * x[j];
    x[i] = sum / a[i * m + i];
  }
  return 1;
}
static int o_lu_mode = 0;      /* 0: LAPACK order (what the reference's configuration runs); 1: levmar's in-tree LU */
void oracle_set_lu_mode(int mode) { o_lu_mode = mode; }
/* tests only: replace AX_EQ_B_LU by a caller-supplied solver (tests/test_oracle_front.py plugs in the LAPACK that the
 * compiled levmar was linked with, to separate "the restatement of dlevmar_dif" from "the rounding of one LAPACK build") */
typedef int (*o_lu_fn)(const double *A, const double *B, double *x, int m);
static o_lu_fn o_lu_hook = 0;
void oracle_set_lu_hook(o_lu_fn fn) { o_lu_hook = fn; }
int oracle_levmar_dif(o_lm_func func, double *p, int m, int n, int itmax, const double opts[5],
                      double info[10], void *adata) {
  double *e = (double *)malloc(sizeof(double) * (size_t)n * (4 + m));
  double *hx = e + n, *wrk = hx + n, *wrk2 = wrk + n, *jac = wrk2 + n;
  double jacTe[8], jacTjac[64], Dp[8], diag[8], pDp[8];
  double mu = 0, tmp, p_eL2, jacTe_inf = 0, pDp_eL2, p_L2 = 0, Dp_L2 = DBL_MAX, dF, dL;
  double tau = opts[0], eps1 = opts[1], eps2 = opts[2], eps2_sq = opts[2] * opts[2], eps3 = opts[3], delta = opts[4];
  double init_p_eL2;
  int nu, nu2, stop = 0, nfev, njap = 0, nlss = 0, K = (m >= 10) ? m : 10, updjac = 0, updp = 1, newjac = 0;
  int i, j, k, l, issolved;
  func(p, hx, m, n, adata); nfev = 1;
  p_eL2 = o_l2nrmxmy(e, hx, n);
  init_p_eL2 = p_eL2;
  if (!isfinite(p_eL2)) stop = 7;
  nu = 20;
  for (k = 0; k < itmax && !stop; ++k) {
    if (p_eL2 <= eps3) { stop = 6; break; }
    if ((updp && nu > 16) || updjac == K) {
      for (j = 0; j < m; ++j) { /* forward differences */
        double d = 1E-04 * p[j], t;
        d = fabs(d);
        if (d < delta) d = delta;
        t = p[j]; p[j] += d;
        func(p, wrk, m, n, adata);
        p[j] = t;
        d = 1.0 / d;
        for (i = 0; i < n; ++i) jac[i * m + j] = (wrk[i] - hx[i]) * d;
      }
      ++njap; nfev += m;
      nu = 2; updjac = 0; updp = 0; newjac = 1;
    }
    if (newjac) {
      newjac = 0;
      for (i = m * m; i-- > 0;) jacTjac[i] = 0.0;
      for (i = m; i-- > 0;) jacTe[i] = 0.0;
      for (l = n; l-- > 0;) {
        double *jaclm = jac + l * m;
        for (i = m; i-- > 0;) {
          double *row = jacTjac + i * m, alpha = jaclm[i];
          for (j = i + 1; j-- > 0;) row[j] += jaclm[j] * alpha;
          jacTe[i] += alpha * e[l];
        }
      }
      for (i = m; i-- > 0;) for (j = i + 1; j < m; ++j) jacTjac[i * m + j] = jacTjac[j * m + i];
      for (i = 0, p_L2 = jacTe_inf = 0.0; i < m; ++i) {
        if (jacTe_inf < (tmp = fabs(jacTe[i]))) jacTe_inf = tmp;
        diag[i] = jacTjac[i * m + i];
        p_L2 += p[i] * p[i];
      }
    }
    if (jacTe_inf <= eps1) { Dp_L2 = 0.0; stop = 1; break; }
    if (k == 0) {
      for (i = 0, tmp = DBL_MIN; i < m; ++i) if (diag[i] > tmp) tmp = diag[i];
      mu = tau * tmp;
    }
    for (i = 0; i < m; ++i) jacTjac[i * m + i] += mu;
    issolved = (m <= 8) ? (o_lu_hook ? o_lu_hook(jacTjac, jacTe, Dp, m) : (o_lu_mode ? o_ax_eq_b_lu_nolapack(jacTjac, jacTe, Dp, m) : o_ax_eq_b_lu(jacTjac, jacTe, Dp, m))) : 0; ++nlss;   /* lm_core.c:706 */
    if (issolved) {
      for (i = 0, Dp_L2 = 0.0; i < m; ++i) { pDp[i] = p[i] + (tmp = Dp[i]); Dp_L2 += tmp * tmp; }
      if (Dp_L2 <= eps2_sq * p_L2) { stop = 2; break; }
      if (Dp_L2 >= (p_L2 + eps2) / (1E-12 * 1E-12)) { stop = 4; break; }
      func(pDp, wrk, m, n, adata); ++nfev;
      pDp_eL2 = o_l2nrmxmy(wrk2, wrk, n);
      if (!isfinite(pDp_eL2)) { stop = 7; break; }
      dF = p_eL2 - pDp_eL2;
      if (updp || dF > 0) { /* Broyden rank-one update */
        for (i = 0; i < n; ++i) {
          for (l = 0, tmp = 0.0; l < m; ++l) tmp += jac[i * m + l] * Dp[l];
          tmp = (wrk[i] - hx[i] - tmp) / Dp_L2;
          for (j = 0; j < m; ++j) jac[i * m + j] += tmp * Dp[j];
        }
        ++updjac; newjac = 1;
      }
      for (i = 0, dL = 0.0; i < m; ++i) dL += Dp[i] * (mu * Dp[i] + jacTe[i]);
      if (dL > 0.0 && dF > 0.0) {
        tmp = (2.0 * dF / dL - 1.0);
        tmp = 1.0 - tmp * tmp * tmp;
        mu = mu * ((tmp >= 0.3333333334) ? tmp : 0.3333333334);
        nu = 2;
        for (i = 0; i < m; ++i) p[i] = pDp[i];
        for (i = 0; i < n; ++i) { e[i] = wrk2[i]; hx[i] = wrk[i]; }
        p_eL2 = pDp_eL2;
        updp = 1;
        continue;
      }
    }
    mu *= nu;
    nu2 = nu << 1;
    if (nu2 <= nu) { stop = 5; break; }
    nu = nu2;
    for (i = 0; i < m; ++i) jacTjac[i * m + i] = diag[i];
  }
  if (k >= itmax) stop = 3;
  if (info) {
    /* lm_core.c:815-831: the diagonal restored, info[4] = mu / max J^T J (i,i) */
    for (i = 0, tmp = DBL_MIN; i < m; ++i) if (tmp < diag[i]) tmp = diag[i];
    info[0] = init_p_eL2; info[1] = p_eL2; info[2] = jacTe_inf; info[3] = Dp_L2; info[4] = mu / tmp;
    info[5] = (double)k; info[6] = (double)stop; info[7] = (double)nfev; info[8] = (double)njap; info[9] = (double)nlss;
  }
  free(e);
  return (stop != 4 && stop != 7) ? k : -1;
}

/* costFun_MLEstimateLine3d, utils.cpp:954-978 */
typedef struct { const orpt *pts; int n, idx1, idx2; double ci1[9], ci2[9]; } o_mle_data;
static void o_mle_cost(const double *p, double *error, int m, int n, void *adata) {
  const o_mle_data *d = (const o_mle_data *)adata;
  int i, k;
  (void)m; (void)n;
  for (i = 0; i < d->n; ++i) {
    if (i == d->idx1 || i == d->idx2) {
      const double *ci = (i == d->idx1) ? d->ci1 : d->ci2;
      const double *e = (i == d->idx1) ? p : p + 3;
      double v[3], t[3];
      for (k = 0; k < 3; k++) v[k] = e[k] - d->pts[i].pos[k];
      for (k = 0; k < 3; k++) t[k] = v[0] * ci[0 * 3 + k] + v[1] * ci[1 * 3 + k] + v[2] * ci[2 * 3 + k]; /* v^T * C^-1 */
      error[i] = t[0] * v[0] + t[1] * v[1] + t[2] * v[2];
    } else
      error[i] = o_mah_dist(&d->pts[i], p, p + 3);
  }
}

/* jac_rpt2ln_mahvec_wrt_ln (utils.cpp:1086-1116) in closed form: with a = M(x-A), d = M(B-A),
 * S = a.d, D = d.d and m_j = column j of M, the symbolic entries of the reference are
 *   dv_r/dA_j = M_rj - M_rj S/D - d_r (m_j.a + m_j.d)/D + 2 S (m_j.d) d_r / D^2
 *   dv_r/dB_j = (m_j.a) d_r / D + M_rj S/D - 2 S (m_j.d) d_r / D^2                              */
static void o_jac_line(const orpt *pt, const double l[6], double J[18]) {
  const double *M = pt->DU;
  double a[3], b[3], d[3], S, D;
  int r, j;
  for (r = 0; r < 3; r++) {
    a[r] = M[3 * r] * (pt->pos[0] - l[0]) + M[3 * r + 1] * (pt->pos[1] - l[1]) + M[3 * r + 2] * (pt->pos[2] - l[2]);
    b[r] = M[3 * r] * (pt->pos[0] - l[3]) + M[3 * r + 1] * (pt->pos[1] - l[4]) + M[3 * r + 2] * (pt->pos[2] - l[5]);
    d[r] = a[r] - b[r];
  }
  S = a[0] * d[0] + a[1] * d[1] + a[2] * d[2];
  D = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  for (j = 0; j < 3; j++) {
    double ma = M[j] * a[0] + M[3 + j] * a[1] + M[6 + j] * a[2];
    double md = M[j] * d[0] + M[3 + j] * d[1] + M[6 + j] * d[2];
    for (r = 0; r < 3; r++) {
      double Mrj = M[3 * r + j];
      J[r * 6 + j] = Mrj - Mrj * S / D - d[r] * (ma + md) / D + 2.0 * S * md * d[r] / (D * D);
      J[r * 6 + 3 + j] = ma * d[r] / D + Mrj * S / D - 2.0 * S * md * d[r] / (D * D);
    }
  }
}

/* MLEstimateLine3d + MleLine3dCov, utils.cpp:980-1050, 1138-1159.  pts: the supporting points of
 * the RANSAC line (line.pts).  Outputs A,B (updated), covA, covB.  Returns levmar's iteration
 * count (or -1).                                                                                */
int oracle_mle_line3d(const orpt *pts, int n, int maxIter, double A[3], double B[3], double covA[9],
                      double covB[9], double info[10]) {
  double minv = 100, maxv = -100, para[6], opts[5], H[36], I6[36];
  int e1 = 0, e2 = 0, i, k, l, r, nit;
  o_mle_data data;
  for (i = 0; i < n; ++i) {
    double dp = (pts[i].pos[0] - A[0]) * (A[0] - B[0]) + (pts[i].pos[1] - A[1]) * (A[1] - B[1]) + (pts[i].pos[2] - A[2]) * (A[2] - B[2]);
    if (dp < minv) { minv = dp; e1 = i; }
    if (dp > maxv) { maxv = dp; e2 = i; }
  }
  if (e1 > e2) { int t = e1; e1 = e2; e2 = t; }
  opts[0] = 1E-03; opts[1] = 1E-10; opts[2] = 1E-20; opts[3] = 1E-20; opts[4] = 1E-06;
  data.pts = pts; data.n = n; data.idx1 = e1; data.idx2 = e2;
  o_inv3(pts[e1].cov, data.ci1);
  o_inv3(pts[e2].cov, data.ci2);
  /* paraVec: positions of the points with index idx_end1 then idx_end2 (if equal: 3 params only;
     cannot happen for >= 2 distinct points) */
  for (k = 0; k < 3; k++) { para[k] = pts[e1].pos[k]; para[3 + k] = pts[e2].pos[k]; }
  nit = oracle_levmar_dif(o_mle_cost, para, 6, n, maxIter, opts, info, &data);
  for (k = 0; k < 3; k++) { A[k] = para[k]; B[k] = para[3 + k]; }
  for (i = 0; i < 36; i++) H[i] = 0;
  for (i = 0; i < n; ++i) {
    double J[18];
    for (k = 0; k < 18; k++) J[k] = 0;
    if (i == e1) { for (r = 0; r < 3; r++) for (k = 0; k < 3; k++) J[r * 6 + k] = -pts[i].DU[3 * r + k]; }
    else if (i == e2) { for (r = 0; r < 3; r++) for (k = 0; k < 3; k++) J[r * 6 + 3 + k] = -pts[i].DU[3 * r + k]; }
    else o_jac_line(&pts[i], para, J);
    for (r = 0; r < 3; r++) for (k = 0; k < 6; k++) for (l = 0; l < 6; l++) H[k * 6 + l] += J[r * 6 + k] * J[r * 6 + l];
  }
  for (i = 0; i < 36; i++) I6[i] = (i % 7 == 0) ? 1.0 : 0.0;
  if (!o_lu_solve(6, H, 6, I6)) for (i = 0; i < 36; i++) I6[i] = NAN;
  for (r = 0; r < 3; r++) for (k = 0; k < 3; k++) { covA[3 * r + k] = I6[r * 6 + k]; covB[3 * r + k] = I6[(r + 3) * 6 + 3 + k]; }
  return nit;
}

/* compPt3dCov + the RandomPoint3d ctor for one point (utils.cpp:690-722, lineslam.h:59-81): cov, DU, W_sqrt */
void oracle_pt_cov(const double pt[3], double focal, const lf_params *P, double cov[9], double DU[9], double Ws[3]) {
  orpt o;
  int i;
  o_comp_pt3d_cov(pt, focal, P, &o);
  for (i = 0; i < 9; i++) { cov[i] = o.cov[i]; DU[i] = o.DU[i]; }
  for (i = 0; i < 3; i++) Ws[i] = o.W_sqrt[i];
}

/* MLEstimateLine3d on caller-supplied support points (the free function of src/line/utils.h; the points'
 * covariances come from compPt3dCov as at lineslam.cpp:283-285): pts [n][3], AB = the RANSAC line on entry, the
 * MLE end points on return.  For the golden-vector tests (tests/golden/mle_fixtures.npz).                        */
int oracle_mle_points(const double *pts, int n, double focal, const lf_params *P, double AB[6], double covA[9],
                      double covB[9], double info[10]) {
  orpt *rp = (orpt *)malloc(sizeof(orpt) * (size_t)(n > 0 ? n : 1));
  int i, nit;
  for (i = 0; i < n; i++) o_comp_pt3d_cov(pts + 3 * i, focal, P, &rp[i]);
  nit = oracle_mle_line3d(rp, n, P->line3d_mle_iter_num, AB, AB + 3, covA, covB, info);
  free(rp);
  return nit;
}

/* --------------------------------------------------------------------------------------------
 * Node::detect3DLines after the LSD call (lineslam.cpp:213-357).
 *   segs[nseg][5]        LSD output rows
 *   K[9]                 camera matrix (row-major)
 *   recs                 out: one record per line with depth (`lines`), at most cap
 *   cand_flag[nseg]      out (optional): 0 dropped by the 2D length filter, 1 kept without depth,
 *                        2 kept with depth
 *   cand_info[nseg][8]   out (optional): numSmp, #valid samples, #RANSAC inliers, A(3) before MLE..
 * returns the number of records.                                                               */
int oracle_detect3d(const uint8_t *gray, int gstride, const float *depth, int dstride_elems, int w,
                    int h, const double K[9], const lf_params *P, uint64_t frame_id,
                    const double *segs, int nseg, lf_line_record *recs, int cap, int *cand_flag,
                    double *cand_info) {
  double Kinv[9], *gx, *gy;
  int i, j, k, nrec = 0;
  { /* Eigen::Matrix3d::inverse(): cofactors / determinant */
    double c00 = K[4] * K[8] - K[5] * K[7], c10 = K[2] * K[7] - K[1] * K[8], c20 = K[1] * K[5] - K[2] * K[4];
    double det = c00 * K[0] + c10 * K[3] + c20 * K[6], inv = 1.0 / det;
    Kinv[0] = c00 * inv; Kinv[1] = c10 * inv; Kinv[2] = c20 * inv;
    Kinv[3] = (K[5] * K[6] - K[3] * K[8]) * inv; Kinv[4] = (K[0] * K[8] - K[2] * K[6]) * inv; Kinv[5] = (K[2] * K[3] - K[0] * K[5]) * inv;
    Kinv[6] = (K[3] * K[7] - K[4] * K[6]) * inv; Kinv[7] = (K[1] * K[6] - K[0] * K[7]) * inv; Kinv[8] = (K[0] * K[4] - K[1] * K[3]) * inv;
  }
  gx = (double *)malloc(sizeof(double) * (size_t)w * h);
  gy = (double *)malloc(sizeof(double) * (size_t)w * h);
  oracle_sobel5(gray, gstride, w, h, gx, gy);
  /* Three phases, as the reference: (1) `#pragma omp parallel for` over the LSD segments (lineslam.cpp:246): sampling,
   * point covariances, RANSAC line; (2) the serial compaction with lid = index and getGradient ("must not be
   * parallelized", :332-341); (3) `#pragma omp parallel for` over the kept lines (:344): MSLD + MLE.  The pragmas are
   * active only in the OpenMP flavour of this library (_build/liboracle_omp.so, the reference-shaped CPU baseline of
   * bench.py); results do not depend on the schedule (the random streams are keyed by frame and segment).             */
  {
    typedef struct { int have, ninl; double A[3], B[3]; orpt *sup; } o_seg;
    o_seg *sg = (o_seg *)calloc((size_t)(nseg > 0 ? nseg : 1), sizeof(o_seg));
    int *kept = (int *)malloc(sizeof(int) * (size_t)(nseg > 0 ? nseg : 1));
#pragma omp parallel for schedule(dynamic, 8) private(j, k)
    for (i = 0; i < nseg; i++) {
      const double *s = segs + 5 * (size_t)i;
      double a = s[0], b = s[1], c = s[2], d = s[3], p[2], q[2], len, numSmp, pts3[128][3];
      int np = 0, ninl, inl[128], have = 0;
      orpt rp[128];
      double A[3], B[3];
      if (cand_flag) cand_flag[i] = 0;
      if (cand_info) for (k = 0; k < 8; k++) cand_info[8 * (size_t)i + k] = 0;
      if (!(sqrt((a - c) * (a - c) + (b - d) * (b - d)) > P->line_segment_len_thresh)) continue; /* :218 */
      if (cand_flag) cand_flag[i] = 1;
      p[0] = a; p[1] = b; q[0] = c; q[1] = d;
      len = sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]));
      numSmp = len / P->line_sample_interval;
      if (numSmp < (double)P->line_sample_min_num) numSmp = (double)P->line_sample_min_num;
      if (numSmp > (double)P->line_sample_max_num) numSmp = (double)P->line_sample_max_num;
      for (j = 0; j <= numSmp && np < 128; ++j) { /* :252-288 */
        double ptx = p[0] * (1 - j / numSmp) + q[0] * (j / numSmp);
        double pty = p[1] * (1 - j / numSmp) + q[1] * (j / numSmp);
        int row, col;
        double depval, zval = -1;
        if (ptx < 0 || pty < 0 || ptx >= w || pty >= h) continue;
        if ((floor(ptx) == ptx) && (floor(pty) == pty)) {
          col = (int)(ptx - 1); if (col < 0) col = 0;
          row = (int)(pty - 1); if (row < 0) row = 0;
        } else { col = (int)ptx; row = (int)pty; }
        depval = depth[(size_t)row * dstride_elems + col];
        if (depval < O_EPS || isnan((float)depval)) { } else zval = depval / P->depth_scaling;
        if (zval > 0) {
          double x = Kinv[0] * ptx + Kinv[1] * pty + Kinv[2] * 1.0;
          double y = Kinv[3] * ptx + Kinv[4] * pty + Kinv[5] * 1.0;
          double z = Kinv[6] * ptx + Kinv[7] * pty + Kinv[8] * 1.0;
          x = x / z; y = y / z;
          pts3[np][0] = x * zval; pts3[np][1] = y * zval; pts3[np][2] = zval;
          np++;
        }
      }
      if (cand_info) { cand_info[8 * (size_t)i + 0] = numSmp; cand_info[8 * (size_t)i + 1] = np; }
      { double need = numSmp * P->ratio_of_collinear_pts; if (need < 10.0) need = 10.0; if (np < need) continue; } /* :289 */
      for (j = 0; j < np; j++) o_comp_pt3d_cov(pts3[j], K[0], P, &rp[j]);
      ninl = o_extract3dline(rp, np, P, P->rng_seed, O_STREAM_LINE3D(frame_id, i), A, B, inl);
      if (cand_info) { cand_info[8 * (size_t)i + 2] = ninl; for (k = 0; k < 3; k++) cand_info[8 * (size_t)i + 3 + k] = A[k]; }
      if (ninl / numSmp > P->ratio_of_collinear_pts &&
          sqrt((A[0] - B[0]) * (A[0] - B[0]) + (A[1] - B[1]) * (A[1] - B[1]) + (A[2] - B[2]) * (A[2] - B[2])) > P->line3d_length_thresh)
        have = 1; /* :302-307 */
      if (!have) continue;
      if (cand_flag) cand_flag[i] = 2;
      sg[i].have = 1; sg[i].ninl = ninl;
      for (k = 0; k < 3; k++) { sg[i].A[k] = A[k]; sg[i].B[k] = B[k]; }
      sg[i].sup = (orpt *)malloc(sizeof(orpt) * (size_t)ninl);
      for (j = 0; j < ninl; j++) sg[i].sup[j] = rp[inl[j]];
    }
    for (i = 0; i < nseg; i++) {     /* serial: lid = running index, complineEq2d, getGradient */
      if (!sg[i].have) continue;
      if (nrec < cap) {
        const double *s = segs + 5 * (size_t)i;
        lf_line_record *R = &recs[nrec];
        double p[2] = {s[0], s[1]}, q[2] = {s[2], s[3]}, l0, l1, l2, nn;
        memset(R, 0, sizeof *R);
        R->p[0] = p[0]; R->p[1] = p[1]; R->q[0] = q[0]; R->q[1] = q[1];
        R->lid = nrec; R->seg = i;
        /* complineEq2d, lineslam.h:139-150: (p,1) x (q,1), normalised by sqrt(a^2+b^2) */
        l0 = p[1] * 1.0 - 1.0 * q[1]; l1 = 1.0 * q[0] - p[0] * 1.0; l2 = p[0] * q[1] - p[1] * q[0];
        nn = sqrt(l0 * l0 + l1 * l1);
        R->lineEq2d[0] = l0 / nn; R->lineEq2d[1] = l1 / nn; R->lineEq2d[2] = l2 / nn;
        oracle_line_gradient(gx, gy, w, h, p, q, R->r);
        kept[nrec] = i;
      }
      nrec++;
    }
    {
      const int nk = nrec < cap ? nrec : cap;
      int li;
#pragma omp parallel for schedule(dynamic, 4) private(k)
      for (li = 0; li < nk; li++) {
        const int si = kept[li];
        lf_line_record *R = &recs[li];
        double info[10], A9[9], V[9], wv[3], A[3], B[3];
        int r2, c2;
        oracle_msld(gx, gy, w, h, R->p, R->q, R->r, P->msld_sample_interval, P->rng_seed, O_STREAM_LINE3D(frame_id, si), R->des);
        for (k = 0; k < 3; k++) { A[k] = sg[si].A[k]; B[k] = sg[si].B[k]; }
        oracle_mle_line3d(sg[si].sup, sg[si].ninl, P->line3d_mle_iter_num, A, B, R->covA, R->covB, info);
        for (k = 0; k < 3; k++) { R->A[k] = A[k]; R->B[k] = B[k]; }
        /* rndA / rndB = RandomPoint3d(A, covA) */
        for (k = 0; k < 9; k++) A9[k] = R->covA[k];
        o_jacobi(3, A9, V, wv);
        for (r2 = 0; r2 < 3; r2++) { R->Wsa[r2] = sqrt(wv[r2]); for (c2 = 0; c2 < 3; c2++) R->DUa[3 * r2 + c2] = (1 / R->Wsa[r2]) * V[3 * c2 + r2]; }
        for (k = 0; k < 9; k++) A9[k] = R->covB[k];
        o_jacobi(3, A9, V, wv);
        for (r2 = 0; r2 < 3; r2++) { R->Wsb[r2] = sqrt(wv[r2]); for (c2 = 0; c2 < 3; c2++) R->DUb[3 * r2 + c2] = (1 / R->Wsb[r2]) * V[3 * c2 + r2]; }
      }
    }
    for (i = 0; i < nseg; i++) free(sg[i].sup);
    free(sg); free(kept);
  }
  free(gx); free(gy);
  return nrec;
}

/* threads of the OpenMP flavour (bench.py's reference-shaped CPU baseline); 1 in the other flavours */
#ifdef _OPENMP
#include <omp.h>
int oracle_omp_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int oracle_omp_threads(int n) { (void)n; return 1; }
#endif

/* exported helpers for primitive-level tests */
void oracle_jacobi3(const double *A, double *V, double *w) { double T[9]; int i; for (i = 0; i < 9; i++) T[i] = A[i]; o_jacobi(3, T, V, w); }
void oracle_jacobi4(const double *A, double *V, double *w) { double T[16]; int i; for (i = 0; i < 16; i++) T[i] = A[i]; o_jacobi(4, T, V, w); }
int oracle_solve6(const double *A, const double *b, double *x) { double T[36], B[6]; int i, r; for (i = 0; i < 36; i++) T[i] = A[i]; for (i = 0; i < 6; i++) B[i] = b[i]; r = o_lu_solve(6, T, 1, B); for (i = 0; i < 6; i++) x[i] = B[i]; return r; }
uint32_t oracle_rand31(uint64_t seed, uint64_t stream, uint64_t ctr) { return o_rand31(seed, stream, ctr); }
/* levmar's AX_EQ_B_LU as transcribed above (the product's scalar form is exported by product_hooks.c for the cross-check) */
int oracle_lu_netlib(const double *A, const double *b, double *x, int m) { return o_ax_eq_b_lu(A, b, x, m); }

/* MLEstimateLine3d's levmar problem (costFun_MLEstimateLine3d on the support points, utils.cpp:980-1012) run twice from
 * the same start with the SAME cost function o_mle_cost: by the restatement oracle_levmar_dif, and by a compiled
 * dlevmar_dif handed in as a function pointer (the reference's levmar-2.6 from oracle/_ref, loaded by the test -- nothing of
 * it is linked here).  out_*: parameters [6] and levmar's info [10] of the two runs; returns the two iteration counts. */
typedef int (*o_dlevmar_dif_fn)(void (*)(double *, double *, int, int, void *), double *, double *, int, int, int, double *,
                                double *, double *, double *, void *);
static void o_mle_cost_nc(double *p, double *hx, int m, int n, void *adata) { o_mle_cost(p, hx, m, n, adata); }
void oracle_mle_levmar_pair(const double *pts, int n, double focal, const lf_params *P, const double AB[6],
                            o_dlevmar_dif_fn compiled, double own_p[6], double own_info[10], double ref_p[6],
                            double ref_info[10], int nit[2]) {
  orpt *rp = (orpt *)malloc(sizeof(orpt) * (size_t)(n > 0 ? n : 1));
  double minv = 100, maxv = -100, opts[5], *x0 = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  const double *A = AB, *B = AB + 3;
  int e1 = 0, e2 = 0, i, k;
  o_mle_data data;
  for (i = 0; i < n; i++) o_comp_pt3d_cov(pts + 3 * i, focal, P, &rp[i]);
  for (i = 0; i < n; ++i) {
    double dp = (rp[i].pos[0] - A[0]) * (A[0] - B[0]) + (rp[i].pos[1] - A[1]) * (A[1] - B[1]) + (rp[i].pos[2] - A[2]) * (A[2] - B[2]);
    if (dp < minv) { minv = dp; e1 = i; }
    if (dp > maxv) { maxv = dp; e2 = i; }
  }
  if (e1 > e2) { int t = e1; e1 = e2; e2 = t; }
  opts[0] = 1E-03; opts[1] = 1E-10; opts[2] = 1E-20; opts[3] = 1E-20; opts[4] = 1E-06;
  data.pts = rp; data.n = n; data.idx1 = e1; data.idx2 = e2;
  o_inv3(rp[e1].cov, data.ci1);
  o_inv3(rp[e2].cov, data.ci2);
  for (k = 0; k < 3; k++) { own_p[k] = ref_p[k] = rp[e1].pos[k]; own_p[3 + k] = ref_p[3 + k] = rp[e2].pos[k]; }
  nit[0] = oracle_levmar_dif(o_mle_cost, own_p, 6, n, P->line3d_mle_iter_num, opts, own_info, &data);
  nit[1] = compiled ? compiled(o_mle_cost_nc, ref_p, x0, 6, n, P->line3d_mle_iter_num, opts, ref_info, 0, 0, &data) : -2;
  free(rp); free(x0);
}
