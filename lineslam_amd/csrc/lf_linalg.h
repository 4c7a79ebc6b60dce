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

/* ---------------------------------------------------------------- linear solve
 * Gaussian elimination with partial pivoting on the n x n row-major matrix A (overwritten) and the
 * n x m right-hand side B (row-major, overwritten with the solution).  Returns 0 if singular. */
#define LF_DEFINE_SOLVE(NAME, N)                                                               \
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
      if (!(big > 0.0)) return 0;                                                              \
      if (piv != k)                                                                            \
        for (i = k + 1; i < N; i++)          /* row swap with constant indices */              \
          if (i == piv) {                                                                      \
            for (j = 0; j < N; j++) { double t = A[k * N + j]; A[k * N + j] = A[i * N + j]; A[i * N + j] = t; } \
            for (j = 0; j < m; j++) { double t = B[k * m + j]; B[k * m + j] = B[i * m + j]; B[i * m + j] = t; } \
          }                                                                                    \
      rp[k] = 1.0 / A[k * N + k];                                                              \
      for (i = k + 1; i < N; i++) {                                                            \
        double f = A[i * N + k] * rp[k];                                                       \
        if (f != 0.0) {                                                                        \
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
LF_DEFINE_SOLVE(lf_solve3, 3)
LF_DEFINE_SOLVE(lf_solve6, 6)
LF_DEFINE_SOLVE(lf_solve7, 7)

/* inverse of a symmetric/any 3x3 (row-major) through lf_solve3; returns 0 if singular */
LF_HD int lf_inv3(const double *A, double *Ainv) {
  double T[9], I[9];
  int i;
  for (i = 0; i < 9; i++) { T[i] = A[i]; I[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  if (!lf_solve3(T, I, 3)) return 0;
  for (i = 0; i < 9; i++) Ainv[i] = I[i];
  return 1;
}

#endif /* LF_LINALG_H */
