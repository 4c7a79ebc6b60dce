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

/* ---- lf_math.h against glibc, measured in ulps of the correctly rounded result (tests/test_lf_math.py).  The yardstick is the
 * x87 long-double libm (64-bit significand: its own error is < 1/2048 ulp of a double).  fn: 0 exp(x) 1 log(x) 2 log10(x) 3 pow(x,y)
 * 4 acos(x) 5 atan2(x = y-argument, y = x-argument) 6 sin(x) 7 cos(x) 8 atan2_cr 9 sin_cr 10 cos_cr.
 * Returns, per element: the lf_math value, glibc's double value, and both errors in ulps.                                    */
#include <math.h>
static double ulps_of(double got, long double want) {
  if (got == (double)want && (isinf(got) || got == 0.0)) return 0.0;
  int e; frexpl(want, &e);                       /* want = m 2^e, 0.5 <= |m| < 1  ->  ulp(double) = 2^(e-53) */
  if (e < -1021) e = -1021;
  return (double)fabsl(((long double)got - want) / ldexpl(1.0L, e - 53));
}
int product_math_ulps(int fn, const double *x, const double *y, int n, double *v_lf, double *v_libm, double *e_lf, double *e_libm) {
  for (int i = 0; i < n; i++) {
    double a = x[i], b = y ? y[i] : 0.0, lf = 0.0, lm = 0.0, s, c; long double w = 0.0L;
    switch (fn) {
      case 0: lf = lf_exp(a); lm = exp(a); w = expl((long double)a); break;
      case 1: lf = lf_log(a); lm = log(a); w = logl((long double)a); break;
      case 2: lf = lf_log10(a); lm = log10(a); w = log10l((long double)a); break;
      case 3: lf = lf_pow(a, b); lm = pow(a, b); w = powl((long double)a, (long double)b); break;
      case 4: lf = lf_acos(a); lm = acos(a); w = acosl((long double)a); break;
      case 5: lf = lf_atan2(a, b); lm = atan2(a, b); w = atan2l((long double)a, (long double)b); break;
      case 6: lf_sincos(a, &s, &c); lf = s; lm = sin(a); w = sinl((long double)a); break;
      case 7: lf_sincos(a, &s, &c); lf = c; lm = cos(a); w = cosl((long double)a); break;
      case 8: lf = lf_atan2_cr(a, b); lm = atan2(a, b); w = atan2l((long double)a, (long double)b); break;
      case 9: lf_sincos_cr(a, &s, &c); lf = s; lm = sin(a); w = sinl((long double)a); break;
      case 10: lf_sincos_cr(a, &s, &c); lf = c; lm = cos(a); w = cosl((long double)a); break;
      default: return -1;
    }
    v_lf[i] = lf; v_libm[i] = lm; e_lf[i] = ulps_of(lf, w); e_libm[i] = ulps_of(lm, w);
  }
  return 0;
}
