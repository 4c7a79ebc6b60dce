/* oracle/product_hooks.c -- TEST INFRASTRUCTURE ONLY.  The ONE place of the oracle library that compiles the product's
 * lineslam_amd/csrc/lf_linalg.h on purpose: it exports the product's own scalar routines so that tests can hold them bit for
 * bit against the oracle's independent statements (o_linalg.h, front_oracle.c) -- tests/test_oracle_front.py.           */
#include <stdint.h>
#include "../lineslam_amd/csrc/lf_linalg.h"

int oracle_lu_product(const double *A, const double *b, double *x, int m) {       /* lf_lu6 / lf_lu7: levmar's AX_EQ_B_LU */
  double T[49], B[7]; int i, r;
  if (m != 6 && m != 7) return -1;
  for (i = 0; i < m * m; i++) T[i] = A[i];
  for (i = 0; i < m; i++) B[i] = b[i];
  r = (m == 6) ? lf_lu6(T, B) : lf_lu7(T, B);
  for (i = 0; i < m; i++) x[i] = B[i];
  return r;
}
void product_jacobi3(const double *A, double *V, double *w) { double T[9]; int i; for (i = 0; i < 9; i++) T[i] = A[i]; lf_jacobi3(T, V, w); }
void product_jacobi4(const double *A, double *V, double *w) { double T[16]; int i; for (i = 0; i < 16; i++) T[i] = A[i]; lf_jacobi4(T, V, w); }
int product_solve6(const double *A, const double *B, int m, double *X) {          /* lf_solve6: cv::Mat::inv's LU, m right-hand sides */
  double T[36], R[36]; int i, r;
  if (m < 1 || m > 6) return -1;
  for (i = 0; i < 36; i++) T[i] = A[i];
  for (i = 0; i < 6 * m; i++) R[i] = B[i];
  r = lf_solve6(T, R, m);
  for (i = 0; i < 6 * m; i++) X[i] = R[i];
  return r;
}
int product_inv3(const double *A, double *Ainv) { return lf_inv3(A, Ainv); }
uint32_t product_rand31(uint64_t seed, uint64_t stream, uint64_t ctr) { return lf_rand31(seed, stream, ctr); }
