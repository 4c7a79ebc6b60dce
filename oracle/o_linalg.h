/* oracle/o_linalg.h -- TEST INFRASTRUCTURE ONLY.  The oracle's OWN small dense routines, written independently of the
 * product's lineslam_amd/csrc/lf_linalg.h (generic-n loops here, size-specialised macros there) so that the bit-for-bit
 * GPU-vs-oracle tests of the 3D-line stage (a11, a12, a17) and of projectTo3D / featureMatching (f1) compare two
 * implementations end to end.  They follow the SAME published algorithms in the SAME operation order -- which is what makes
 * bit equality a meaningful statement -- and tests/test_oracle_front.py holds each one against numpy:
 *   o_jacobi     cv::SVD of a symmetric 3x3 / 4x4 matrix (lineslam.h:63, motion.cpp:353) as the cyclic Jacobi eigenvalue
 *                iteration: sweeps over (p, q) in row order, rotation t = 1 / (|theta| + sqrt(theta^2 + 1)), at most 30
 *                sweeps, stop when the off-diagonal mass is below 1e-22 of the diagonal's; eigenvalues sorted descending,
 *                ties in index order (as cv::SVD orders singular values)
 *   o_lu_solve   cv::Mat::inv / cv::solve DECOMP_LU (OpenCV 2.4 lapack.cpp LUImpl): partial pivoting with the first largest
 *                |entry|, row operations scaled by the RECIPROCAL pivot, row-oriented back-substitution multiplying by it
 *   o_rand31     the counter-based generator that replaces the reference's unseeded rand(): SplitMix64 finaliser of
 *                (seed, stream, counter), top 31 bits
 * IEEE double, + - * / sqrt only.                                                                                      */
#ifndef O_LINALG_H
#define O_LINALG_H
#include <math.h>
#include <stdint.h>

static uint64_t o_splitmix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}
static uint32_t o_rand31(uint64_t seed, uint64_t stream, uint64_t counter) {
  const uint64_t key = o_splitmix(seed ^ o_splitmix(stream));
  return (uint32_t)(o_splitmix(key + counter * 0xD1B54A32D192ED03ULL) >> 33);
}

/* which consumer draws: the stream ids of the counter generator (frame, segment) / (query node, train node) -- the numbering
 * is part of the interface of the replacement generator, like the seed */
#define O_STREAM_LINE3D(frame, line) ((((uint64_t)(frame)) << 24) ^ (uint64_t)(line) ^ 0x1000000000000000ULL)

/* a: n x n symmetric (row-major, destroyed); v: eigenvectors in columns; w: eigenvalues, descending */
static void o_jacobi(int n, double *a, double *v, double *w) {
  int sweep, p, q, k, i, pass;
  for (p = 0; p < n; p++)
    for (q = 0; q < n; q++) v[p * n + q] = (p == q) ? 1.0 : 0.0;
  for (sweep = 0; sweep < 30; sweep++) {
    double offsum = 0.0, diagsum = 0.0;
    for (p = 0; p < n; p++) {
      diagsum += fabs(a[p * n + p]);
      for (q = p + 1; q < n; q++) offsum += fabs(a[p * n + q]);
    }
    if (offsum == 0.0 || offsum <= 1e-300 || offsum < diagsum * 1e-22) break;
    for (p = 0; p + 1 < n; p++)
      for (q = p + 1; q < n; q++) {
        const double apq = a[p * n + q];
        double app, aqq, theta, t, c, s;
        if (apq == 0.0) continue;
        app = a[p * n + p]; aqq = a[q * n + q];
        theta = (aqq - app) / (2.0 * apq);
        t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        c = 1.0 / sqrt(t * t + 1.0);
        s = t * c;
        a[p * n + p] = app - t * apq;
        a[q * n + q] = aqq + t * apq;
        a[p * n + q] = 0.0;
        a[q * n + p] = 0.0;
        for (k = 0; k < n; k++) {
          double akp, akq, rp, rq;
          if (k == p || k == q) continue;
          akp = a[k * n + p]; akq = a[k * n + q];
          rp = c * akp - s * akq; rq = s * akp + c * akq;
          a[k * n + p] = rp; a[p * n + k] = rp;
          a[k * n + q] = rq; a[q * n + k] = rq;
        }
        for (k = 0; k < n; k++) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (i = 0; i < n; i++) w[i] = a[i * n + i];
  for (pass = 0; pass + 1 < n; pass++)               /* stable exchange sort, descending */
    for (i = 0; i + 1 < n - pass; i++)
      if (w[i] < w[i + 1]) {
        double t = w[i]; w[i] = w[i + 1]; w[i + 1] = t;
        for (k = 0; k < n; k++) { t = v[k * n + i]; v[k * n + i] = v[k * n + i + 1]; v[k * n + i + 1] = t; }
      }
}

/* A: n x n (row-major, destroyed), B: n x m right-hand sides (row-major, replaced by the solution); 0 if singular */
static int o_lu_solve(int n, double *A, int m, double *B) {
  double recip[8];
  int i, j, k;
  for (k = 0; k < n; k++) {
    int best = k;
    double bigv = fabs(A[k * n + k]);
    for (i = k + 1; i < n; i++)
      if (fabs(A[i * n + k]) > bigv) { bigv = fabs(A[i * n + k]); best = i; }
    if (!(bigv > 0.0)) return 0;
    if (best != k) {
      for (j = k; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[best * n + j]; A[best * n + j] = t; }
      for (j = 0; j < m; j++) { double t = B[k * m + j]; B[k * m + j] = B[best * m + j]; B[best * m + j] = t; }
    }
    recip[k] = 1.0 / A[k * n + k];
    for (i = k + 1; i < n; i++) {
      const double f = A[i * n + k] * recip[k];
      if (f != 0.0) {
        for (j = k + 1; j < n; j++) A[i * n + j] -= f * A[k * n + j];
        for (j = 0; j < m; j++) B[i * m + j] -= f * B[k * m + j];
      }
      A[i * n + k] = 0.0;
    }
  }
  for (j = 0; j < m; j++)
    for (i = n - 1; i >= 0; i--) {
      double acc = B[i * m + j];
      for (k = i + 1; k < n; k++) acc -= A[i * n + k] * B[k * m + j];
      B[i * m + j] = acc * recip[i];
    }
  return 1;
}
static int o_inv3(const double *A, double *Ainv) {
  double T[9], E[9];
  int i;
  for (i = 0; i < 9; i++) { T[i] = A[i]; E[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0; }
  if (!o_lu_solve(3, T, 3, E)) return 0;
  for (i = 0; i < 9; i++) Ainv[i] = E[i];
  return 1;
}
/* One-sided (Hestenes) Jacobi SVD of a general 3x3 (row-major): A = U diag(sg) V^T, sg descending.  Stands in for
 * Eigen::JacobiSVD<Matrix3f> inside pcl::TransformationFromCorrespondences (src/line/motion.cpp:540-578).  Column pairs (0,1),
 * (0,2), (1,2) are rotated until a sweep rotates nothing (at most 30 sweeps); a vanishing singular value gets its left vector
 * from the cross product of the others (the reflection fix of the Kabsch step needs a full U). */
static void o_svd3(const double *A, double *U, double *sg, double *V) {
  double W[9];
  int i, j, k, sweep, p, q;
  for (i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0; }
  for (sweep = 0; sweep < 30; sweep++) {
    int any = 0;
    for (p = 0; p < 2; p++)
      for (q = p + 1; q < 3; q++) {
        double pp = 0, qq = 0, pq = 0, zeta, tn, cs, sn;
        for (k = 0; k < 3; k++) { pp += W[3 * k + p] * W[3 * k + p]; qq += W[3 * k + q] * W[3 * k + q]; pq += W[3 * k + p] * W[3 * k + q]; }
        if (pq == 0.0 || pq * pq <= 1e-30 * pp * qq) continue;
        zeta = (qq - pp) / (2.0 * pq);
        tn = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        if (zeta < 0.0) tn = -tn;
        cs = 1.0 / sqrt(1.0 + tn * tn);
        sn = cs * tn;
        any = 1;
        for (k = 0; k < 3; k++) {
          const double wp = W[3 * k + p], wq = W[3 * k + q], vp = V[3 * k + p], vq = V[3 * k + q];
          W[3 * k + p] = cs * wp - sn * wq; W[3 * k + q] = sn * wp + cs * wq;
          V[3 * k + p] = cs * vp - sn * vq; V[3 * k + q] = sn * vp + cs * vq;
        }
      }
    if (!any) break;
  }
  for (j = 0; j < 3; j++) sg[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
  for (i = 0; i < 2; i++)                      /* stable exchange sort of the columns, descending */
    for (j = 0; j + 1 < 3 - i; j++)
      if (sg[j] < sg[j + 1]) {
        double t = sg[j]; sg[j] = sg[j + 1]; sg[j + 1] = t;
        for (k = 0; k < 3; k++) {
          t = W[3 * k + j]; W[3 * k + j] = W[3 * k + j + 1]; W[3 * k + j + 1] = t;
          t = V[3 * k + j]; V[3 * k + j] = V[3 * k + j + 1]; V[3 * k + j + 1] = t;
        }
      }
  for (j = 0; j < 3; j++)
    for (k = 0; k < 3; k++) U[3 * k + j] = (sg[j] > 0.0) ? W[3 * k + j] / sg[j] : 0.0;
  if (!(sg[2] > 1e-12 * sg[0])) {
    if (!(sg[1] > 1e-12 * sg[0])) {            /* rank <= 1: a unit vector orthogonal to u0, from the axis u0 is least aligned with */
      const double ax = fabs(U[0]), ay = fabs(U[3]), az = fabs(U[6]);
      const double e0 = (ax <= ay && ax <= az) ? 1.0 : 0.0, e1 = (e0 == 0.0 && ay <= az) ? 1.0 : 0.0, e2 = (e0 == 0.0 && e1 == 0.0) ? 1.0 : 0.0;
      const double d = e0 * U[0] + e1 * U[3] + e2 * U[6];
      const double v0 = e0 - d * U[0], v1 = e1 - d * U[3], v2 = e2 - d * U[6];
      const double n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
      U[1] = v0 / n; U[4] = v1 / n; U[7] = v2 / n;
    }
    U[2] = U[3] * U[7] - U[6] * U[4];          /* u2 = u0 x u1 */
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}
static double o_det3(const double *M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
#endif
