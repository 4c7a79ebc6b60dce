/* lf_linalg.h -- tiny dense kernels and the counter-based random generator used by the 3D-line
 * and pose stages.  Plain C, host + device, IEEE +,-,*,/,sqrt only (no fma, no libm), so gcc and
 * hipcc (both with -ffp-contract=off) produce identical bits.
 *
 * These stand in for third-party arithmetic that is NOT in the reference tree:
 *   cv::SVD on symmetric 3x3 / 4x4 matrices (src/line/lineslam.h:63, src/line/motion.cpp:353)
 *        -> cyclic Jacobi eigen-decomposition (for a symmetric PSD matrix U = eigenvectors,
 *           W = eigenvalues sorted descending, as cv::SVD orders singular values);
 *   cv::Mat::inv() / LAPACK LU (src/line/motion.cpp:363, levmar AX_EQ_B_LU)
 *        -> Gaussian elimination with partial pivoting;
 *   rand() (src/line/utils.h:49-60, unseeded and shared across OpenMP threads in the reference)
 *        -> lf_rand31(): SplitMix64 of (seed, stream, counter), 31 bits like glibc's RAND_MAX.
 * Validated against numpy.linalg in tests/test_linalg.py.
 */
#ifndef LF_LINALG_H
#define LF_LINALG_H

#include "lf_math.h"

/* ---------------------------------------------------------------- RNG */
LF_HD uint64_t lf_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* stream identifies the consumer (frame, line or pair); counter the draw index */
LF_HD uint32_t lf_rand31(uint64_t seed, uint64_t stream, uint64_t counter) {
  uint64_t k = lf_mix64(seed ^ lf_mix64(stream));
  return (uint32_t)(lf_mix64(k + counter * 0xD1B54A32D192ED03ULL) >> 33);
}
#define LF_STREAM_LINE3D(frame, line) ((((uint64_t)(frame)) << 24) ^ (uint64_t)(line) ^ 0x1000000000000000ULL)
#define LF_STREAM_RELMOTION(fq, ft) ((((uint64_t)(fq)) << 32) ^ (uint64_t)(uint32_t)(ft) ^ 0x3000000000000000ULL)
#define LF_STREAM_PAIR(fq, ft) ((((uint64_t)(fq)) << 32) ^ (uint64_t)(uint32_t)(ft) ^ 0x2000000000000000ULL)

/* ---------------------------------------------------------------- Jacobi eigen-decomposition
 * A: n x n symmetric, row-major, overwritten.  V: columns are eigenvectors.  w: eigenvalues,
 * sorted descending (ties keep index order).  Fixed sweep schedule -> deterministic. */
#define LF_DEFINE_JACOBI(NAME, N)                                                              \
  LF_HD void NAME(double *A, double *V, double *w) {                                           \
    int i, j, p, q, sweep;                                                                     \
    for (i = 0; i < N; i++)                                                                    \
      for (j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;                             \
    for (sweep = 0; sweep < 30; sweep++) {                                                     \
      double off = 0.0, diag = 0.0;                                                            \
      for (p = 0; p < N; p++) {                                                                \
        diag += lf_fabs(A[p * N + p]);                                                         \
        for (q = p + 1; q < N; q++) off += lf_fabs(A[p * N + q]);                              \
      }                                                                                        \
      if (off == 0.0 || off <= 1e-300 || off < diag * 1e-22) break;                            \
      for (p = 0; p < N - 1; p++)                                                              \
        for (q = p + 1; q < N; q++) {                                                          \
          double apq = A[p * N + q];                                                           \
          if (apq == 0.0) continue;                                                            \
          {                                                                                    \
            double app = A[p * N + p], aqq = A[q * N + q];                                     \
            double theta = (aqq - app) / (2.0 * apq);                                          \
            double t = 1.0 / (lf_fabs(theta) + lf_sqrt(theta * theta + 1.0));                  \
            double c, s;                                                                       \
            int k;                                                                             \
            if (theta < 0.0) t = -t;                                                           \
            c = 1.0 / lf_sqrt(t * t + 1.0);                                                    \
            s = t * c;                                                                         \
            A[p * N + p] = app - t * apq;                                                      \
            A[q * N + q] = aqq + t * apq;                                                      \
            A[p * N + q] = 0.0;                                                                \
            A[q * N + p] = 0.0;                                                                \
            for (k = 0; k < N; k++) {                                                          \
              if (k != p && k != q) {                                                          \
                double akp = A[k * N + p], akq = A[k * N + q];                                 \
                double nkp = c * akp - s * akq, nkq = s * akp + c * akq;                       \
                A[k * N + p] = nkp; A[p * N + k] = nkp;                                        \
                A[k * N + q] = nkq; A[q * N + k] = nkq;                                        \
              }                                                                                \
            }                                                                                  \
            for (k = 0; k < N; k++) {                                                          \
              double vkp = V[k * N + p], vkq = V[k * N + q];                                   \
              V[k * N + p] = c * vkp - s * vkq;                                                \
              V[k * N + q] = s * vkp + c * vkq;                                                \
            }                                                                                  \
          }                                                                                    \
        }                                                                                      \
    }                                                                                          \
    for (i = 0; i < N; i++) w[i] = A[i * N + i];                                               \
    for (i = 0; i < N - 1; i++)          /* stable bubble sort, descending; only constant */        \
      for (j = 0; j < N - 1 - i; j++)    /* indices after unrolling (register friendly)   */        \
        if (w[j] < w[j + 1]) {                                                                   \
          double tw = w[j];                                                                      \
          int k;                                                                                 \
          w[j] = w[j + 1]; w[j + 1] = tw;                                                        \
          for (k = 0; k < N; k++) {                                                              \
            double tv = V[k * N + j];                                                            \
            V[k * N + j] = V[k * N + j + 1]; V[k * N + j + 1] = tv;                              \
          }                                                                                      \
        }                                                                                        \
  }
LF_DEFINE_JACOBI(lf_jacobi3, 3)
LF_DEFINE_JACOBI(lf_jacobi4, 4)

/* levmar's LEVMAR_L2NRMXMY (external/levmar-2.6/misc_core.c) for an error vector that is already formed: four running
 * sums over blocks of eight from the top downwards, the remainder by the fall-through switch, sum0+sum1+sum2+sum3. */
LF_HD double lf_l2nrm_sq(const double *e, int n) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const int blockn = (n >> 3) << 3;
  int i;
  for (i = blockn - 1; i > 0; i -= 8) {
    s0 += e[i] * e[i]; s1 += e[i - 1] * e[i - 1]; s2 += e[i - 2] * e[i - 2]; s3 += e[i - 3] * e[i - 3];
    s0 += e[i - 4] * e[i - 4]; s1 += e[i - 5] * e[i - 5]; s2 += e[i - 6] * e[i - 6]; s3 += e[i - 7] * e[i - 7];
  }
  i = blockn;
  if (i < n) {
    switch (n - i) {
      case 7: s0 += e[i] * e[i]; ++i; /* fall through */
      case 6: s1 += e[i] * e[i]; ++i; /* fall through */
      case 5: s2 += e[i] * e[i]; ++i; /* fall through */
      case 4: s3 += e[i] * e[i]; ++i; /* fall through */
      case 3: s0 += e[i] * e[i]; ++i; /* fall through */
      case 2: s1 += e[i] * e[i]; ++i; /* fall through */
      case 1: s2 += e[i] * e[i];
    }
  }
  return s0 + s1 + s2 + s3;
}

/* ---------------------------------------------------------------- linear solve
 * Gaussian elimination with partial pivoting on the n x n row-major matrix A (overwritten) and the
 * n x m right-hand side B (row-major, overwritten with the solution).  Returns 0 if singular. */
#if defined(__HIP_DEVICE_COMPILE__)
#define LF_UNI_I(x) __builtin_amdgcn_readfirstlane(x)   /* the value is the same in every lane: branch, do not select */
#else
#define LF_UNI_I(x) (x)
#endif
#define LF_SAME_I(x) (x)
/* "does any lane of the wavefront need this?" -- a uniform guard around rarely needed, select-heavy blocks (the
 * lanes that do not need it are still protected by their own condition inside); plain condition on the host. */
#if defined(__HIP_DEVICE_COMPILE__)
#define LF_ANY(c) (__builtin_amdgcn_ballot_w64((c)) != 0ull)
#else
#define LF_ANY(c) (c)
#endif
/* UNI = LF_SAME_I: lanes may hold different systems.  UNI = LF_UNI_I (the *_u variants): the caller guarantees
 * that all lanes of the wavefront hold the SAME system, so pivot decisions become scalar branches (same
 * arithmetic, same result; on the host the two variants are identical). */
#define LF_DEFINE_SOLVE(NAME, N, UNI)                                                          \
  LF_HD int NAME(double *A, double *B, int m) {                                                \
    int i, j, k;                                                                               \
    double rp[N];                         /* reciprocals of the pivots (one division each) */  \
    for (k = 0; k < N; k++) {                                                                  \
      int piv = k;                                                                             \
      double big = lf_fabs(A[k * N + k]);                                                      \
      for (i = k + 1; i < N; i++) {                                                            \
        double v = lf_fabs(A[i * N + k]);                                                      \
        if (v > big) { big = v; piv = i; }                                                     \
      }                                                                                        \
      piv = UNI(piv);                                                                          \
      if (UNI((int)!(big > 0.0))) return 0;                                                    \
      if (LF_ANY(piv != k))                                                                    \
        for (i = k + 1; i < N; i++)          /* row swap with constant indices */              \
          if (i == piv) {                                                                      \
            for (j = k; j < N; j++) { double t = A[k * N + j]; A[k * N + j] = A[i * N + j]; A[i * N + j] = t; } /* (columns < k of rows >= k are 0.0) */ \
            for (j = 0; j < m; j++) { double t = B[k * m + j]; B[k * m + j] = B[i * m + j]; B[i * m + j] = t; } \
          }                                                                                    \
      rp[k] = 1.0 / A[k * N + k];                                                              \
      for (i = k + 1; i < N; i++) {                                                            \
        double f = A[i * N + k] * rp[k];                                                       \
        if (UNI((int)(f != 0.0))) {                                                            \
          for (j = k + 1; j < N; j++) A[i * N + j] -= f * A[k * N + j];                        \
          for (j = 0; j < m; j++) B[i * m + j] -= f * B[k * m + j];                            \
        }                                                                                      \
        A[i * N + k] = 0.0;                                                                    \
      }                                                                                        \
    }                                                                                          \
    for (j = 0; j < m; j++)                                                                    \
      for (i = N - 1; i >= 0; i--) {                                                           \
        double s = B[i * m + j];                                                               \
        for (k = i + 1; k < N; k++) s -= A[i * N + k] * B[k * m + j];                          \
        B[i * m + j] = s * rp[i];                                                              \
      }                                                                                        \
    return 1;                                                                                  \
  }
LF_DEFINE_SOLVE(lf_solve3, 3, LF_SAME_I)
LF_DEFINE_SOLVE(lf_solve6, 6, LF_SAME_I)
LF_DEFINE_SOLVE(lf_solve7, 7, LF_SAME_I)
LF_DEFINE_SOLVE(lf_solve6_u, 6, LF_UNI_I)
LF_DEFINE_SOLVE(lf_solve7_u, 7, LF_UNI_I)

/* ---------------------------------------------------------------- LAPACK-order LU solve (levmar's AX_EQ_B_LU)
 * levmar-2.6 solves its augmented normal equations with LAPACK: dgetrf + dgetrs (external/levmar-2.6/Axb_core.c:738-830,
 * called at lm_core.c:706).  LAPACK is not part of the reference tree; this restates the published reference algorithm
 * (netlib LAPACK 3.x; for n = 6, 7 dgetrf is the unblocked dgetf2):
 *   dgetf2: for each column j: pivot = first maximum of |A(j:n, j)| (idamax); swap the rows; scale the sub-column by the
 *           RECIPROCAL of the pivot (by division if |pivot| < sfmin); rank-one update of the trailing block (dger);
 *   dgetrs: the row swaps on b; unit lower solve, column oriented; upper solve, column oriented from the LAST column
 *           upwards with a DIVISION by the diagonal (dtrsm 'L','U','N','N').
 * Unlike LF_DEFINE_SOLVE above (OpenCV's LU: reciprocal pivots, row-oriented back-substitution) the back-substitution
 * subtracts in descending column order and divides.  The `if (x != 0)` shortcuts of dger / dtrsm are not restated: they
 * skip additions of +-0 and can change the sign of an exact zero only.  Row-major A (n x n, overwritten), b (n,
 * overwritten by x).  Returns 0 for an exactly singular matrix (dgetrf info > 0: levmar treats it as "not solved"). */
#define LF_LU_SFMIN 2.2250738585072014e-308   /* dlamch('S') */
#define LF_DEFINE_LU_NETLIB(NAME, N, UNI)                                                      \
  LF_HD int NAME(double *A, double *B) {                                                       \
    int i, j, k;                                                                               \
    for (j = 0; j < N; j++) {                                                                  \
      int jp = j;                                                                              \
      double big = lf_fabs(A[j * N + j]);                                                      \
      for (i = j + 1; i < N; i++) {                                                            \
        double v = lf_fabs(A[i * N + j]);                                                      \
        if (v > big) { big = v; jp = i; }                                                      \
      }                                                                                        \
      jp = UNI(jp);                                                                            \
      if (UNI((int)!(big > 0.0))) return 0;                                                    \
      if (LF_ANY(jp != j))                                                                     \
        for (i = j + 1; i < N; i++)                                                            \
          if (i == jp) {                                                                       \
            for (k = 0; k < N; k++) { double t = A[j * N + k]; A[j * N + k] = A[i * N + k]; A[i * N + k] = t; } \
            { double t = B[j]; B[j] = B[i]; B[i] = t; }                                        \
          }                                                                                    \
      if (UNI((int)(big >= LF_LU_SFMIN))) {                                                    \
        double r = 1.0 / A[j * N + j];                                                         \
        for (i = j + 1; i < N; i++) A[i * N + j] = A[i * N + j] * r;                           \
      } else                                                                                   \
        for (i = j + 1; i < N; i++) A[i * N + j] = A[i * N + j] / A[j * N + j];                \
      for (i = j + 1; i < N; i++) {                                                            \
        for (k = j + 1; k < N; k++) A[i * N + k] -= A[i * N + j] * A[j * N + k];               \
        B[i] -= A[i * N + j] * B[j];                                                           \
      }                                                                                        \
    }                                                                                          \
    for (k = N - 1; k >= 0; k--) {                                                             \
      B[k] = B[k] / A[k * N + k];                                                              \
      for (i = 0; i < k; i++) B[i] -= B[k] * A[i * N + k];                                     \
    }                                                                                          \
    return 1;                                                                                  \
  }
LF_DEFINE_LU_NETLIB(lf_lu6, 6, LF_SAME_I)
LF_DEFINE_LU_NETLIB(lf_lu7, 7, LF_SAME_I)
LF_DEFINE_LU_NETLIB(lf_lu7_u, 7, LF_UNI_I)

/* inverse of a symmetric/any 3x3 (row-major) through lf_solve3; returns 0 if singular */
LF_HD int lf_inv3(const double *A, double *Ainv) {
  double T[9], I[9];
  int i;
  for (i = 0; i < 9; i++) { T[i] = A[i]; I[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  if (!lf_solve3(T, I, 3)) return 0;
  for (i = 0; i < 9; i++) Ainv[i] = I[i];
  return 1;
}


/* ---------------------------------------------------------------- 3x3 singular value decomposition
 * One-sided (Hestenes) Jacobi: A = U diag(sg) V^T, sg descending, for a general 3x3 row-major A.
 * Stands in for Eigen::JacobiSVD<Matrix3f> inside pcl::TransformationFromCorrespondences
 * (src/line/motion.cpp:540-578).  A (numerically) zero singular value gets its left vector from the
 * cross product of the other two, which is what the reflection fix of the Kabsch step needs. */
LF_HD void lf_svd3(const double *A, double *U, double *sg, double *V) {
  double W[9];
  int i, j, k, sweep, p, q;
  for (i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (sweep = 0; sweep < 30; sweep++) {
    int rotated = 0;
    for (p = 0; p < 2; p++)
      for (q = p + 1; q < 3; q++) {
        double al = 0, be = 0, ga = 0;
        for (k = 0; k < 3; k++) { al += W[3 * k + p] * W[3 * k + p]; be += W[3 * k + q] * W[3 * k + q]; ga += W[3 * k + p] * W[3 * k + q]; }
        if (ga == 0.0 || ga * ga <= 1e-30 * al * be) continue;
        {
          double zeta = (be - al) / (2.0 * ga);
          double t = 1.0 / (lf_fabs(zeta) + lf_sqrt(1.0 + zeta * zeta));
          double c, sn;
          if (zeta < 0.0) t = -t;
          c = 1.0 / lf_sqrt(1.0 + t * t);
          sn = c * t;
          rotated = 1;
          for (k = 0; k < 3; k++) {
            double wp = W[3 * k + p], wq = W[3 * k + q], vp = V[3 * k + p], vq = V[3 * k + q];
            W[3 * k + p] = c * wp - sn * wq; W[3 * k + q] = sn * wp + c * wq;
            V[3 * k + p] = c * vp - sn * vq; V[3 * k + q] = sn * vp + c * vq;
          }
        }
      }
    if (!rotated) break;
  }
  for (j = 0; j < 3; j++) sg[j] = lf_sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
  for (i = 0; i < 2; i++)            /* stable bubble sort of the columns, descending */
    for (j = 0; j < 2 - i; j++)
      if (sg[j] < sg[j + 1]) {
        double t = sg[j]; sg[j] = sg[j + 1]; sg[j + 1] = t;
        for (k = 0; k < 3; k++) {
          double tw = W[3 * k + j], tv = V[3 * k + j];
          W[3 * k + j] = W[3 * k + j + 1]; W[3 * k + j + 1] = tw;
          V[3 * k + j] = V[3 * k + j + 1]; V[3 * k + j + 1] = tv;
        }
      }
  for (j = 0; j < 3; j++)
    for (k = 0; k < 3; k++) U[3 * k + j] = (sg[j] > 0.0) ? W[3 * k + j] / sg[j] : 0.0;
  if (!(sg[2] > 1e-12 * sg[0])) {     /* rank <= 2: complete U by the cross product */
    if (!(sg[1] > 1e-12 * sg[0])) {   /* rank <= 1: any unit vector orthogonal to u0 */
      double ax = lf_fabs(U[0]), ay = lf_fabs(U[3]), az = lf_fabs(U[6]);
      double e0 = (ax <= ay && ax <= az) ? 1.0 : 0.0, e1 = (e0 == 0.0 && ay <= az) ? 1.0 : 0.0, e2 = (e0 == 0.0 && e1 == 0.0) ? 1.0 : 0.0;
      double d = e0 * U[0] + e1 * U[3] + e2 * U[6], n;
      double v0 = e0 - d * U[0], v1 = e1 - d * U[3], v2 = e2 - d * U[6];
      n = lf_sqrt(v0 * v0 + v1 * v1 + v2 * v2);
      U[1] = v0 / n; U[4] = v1 / n; U[7] = v2 / n;
    }
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}
LF_HD double lf_det3(const double *M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

#endif /* LF_LINALG_H */
