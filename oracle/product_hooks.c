/* oracle/product_hooks.c -- TEST INFRASTRUCTURE ONLY.  The ONE place of the oracle library that compiles the product's
 * lineslam_amd/csrc/lf_linalg.h / lf_pose.h / lf_math.h on purpose: it exports the product's own scalar routines so that tests can
 * hold them bit for bit against the oracle's independent statements (o_linalg.h, o_pose.h, front_oracle.c) --
 * tests/test_oracle_front.py, tests/test_oracle_pose_primitives.py.                                                      */
#include <stdint.h>
#include <string.h>
#include "../lineslam_amd/csrc/lf_linalg.h"
#include "../lineslam_amd/csrc/lf_pose.h"

/* ---- lf_math.h's device-side functions on the host (tests/test_oracle_pair.py, test_lf_math.py) */
double oracle_acos(double x) { return lf_acos(x); }
void oracle_sincos_cr(double x, double *s, double *c) { lf_sincos_cr(x, s, c); }
double oracle_atan2_cr(double y, double x) { return lf_atan2_cr(y, x); }
/* sin / cos of theta = atan2_cr(y, x) (+ LF_PI if flip) through lf_sincos_cr_near; returns theta */
double oracle_r2r_angle(double y, double x, int flip, double *s, double *c) {
  double t0, th, theta;
  lf_dd s0, c0;
  th = lf_atan2_cr_sc(y, x, &t0, &s0, &c0);
  theta = flip ? th + LF_PI : th;
  lf_sincos_cr_near(theta, flip, th, t0, s0, c0, s, c);
  return theta;
}

/* ---- the product's lf_pose.h primitives, flat signatures (the oracle's own statements: oracle_prim_* in pair_oracle.c).
 * which: the primitive; in / out: flat double arrays (layouts in tests/test_oracle_pose_primitives.py). */
int product_prim(int which, const double *in, const float *fin, double *out, float *fout) {
  switch (which) {
    case 0: { double R[9], t[3]; int ok = lf_rel_motion_lines(in, in + 18, (int)in[36], R, t); memcpy(out, R, sizeof R); memcpy(out + 9, t, sizeof t); return ok; }
    case 1: out[0] = lf_mah_dist(in, in + 3, in + 12, in + 15); return 1;
    case 2: { double add; int r = lf_line_inlier(fin, in, in + 3, in + 6, in + 9, in + 12, in + 21, in[30], &add); out[0] = add; return r; }
    case 3: lf_line_edge_error(in, in + 9, in + 18, in + 21, in + 24, in + 27, out); return 1;
    case 4: { lf_se3 X, Y; memcpy(&X, in, sizeof X); lf_se3_oplus(&X, in + 12, &Y); memcpy(out, &Y, sizeof Y); return 1; }
    case 5: { lf_se3 X; lf_tf_to_older_pose(fin, &X); memcpy(out, &X, sizeof X); lf_older_pose_to_tf(&X, fout); return 1; }
    case 6: { lf_se3 X; lf_line_meas m; lf_line_blocks B; double Vi[36], T[36], u[6], dl[6]; int ok;
              memcpy(&X, in, sizeof X);
              m.nA = in + 18; m.nB = in + 21; m.nMa = in + 24; m.nMb = in + 33; m.oA = in + 42; m.oB = in + 45; m.oMa = in + 48; m.oMb = in + 57;
              lf_match_blocks(&X, in + 12, &m, in[66], in[67], (int)in[68], &B);
              memcpy(out, &B, sizeof B);                                         /* V W bl Hpp bp: 120 doubles */
              out[120] = lf_match_chi2(&X, in + 12, &m, in[66], in[67], (int)in[68]);
              ok = lf_match_eliminate(&B, in[69], Vi, T, u);
              memcpy(out + 121, Vi, sizeof Vi); memcpy(out + 157, T, sizeof T); memcpy(out + 193, u, sizeof u);
              lf_match_backsub(&B, Vi, in + 70, dl); memcpy(out + 199, dl, sizeof dl);
              return ok; }
    case 7: out[0] = (double)lf_relmotion_inlier(in, in + 9, in + 12, in + 15, in + 18, in + 21, in[24], in[25]);
            out[1] = lf_relmotion_residual(in, in + 9, in + 12, in + 15, in + 26, in + 35, in + 18, in + 21, in + 44, in + 53);
            out[2] = (double)lf_relmotion_degenerate(in + 62, in[80]);
            { double q[4], R2[9]; lf_r2q(in, q); lf_q2r(q, R2); memcpy(out + 3, q, sizeof q); memcpy(out + 7, R2, sizeof R2); }
            return 1;
    case 8: { lf_point_model pm; pm.raster_cov_x = in[0]; pm.raster_cov_y = in[1]; pm.sigma_depth = in[2]; out[0] = lf_error_function2(fin, fin + 4, fin + 8, &pm);
              lf_project_pt_line(in + 3, in + 6, in + 9, out + 1);
              return lf_point_information(fin, in[12], in[13], in[14], in[15], in[16], out + 4); }
    case 9: { lf_tfc t; int i, n = (int)in[0]; lf_tfc_reset(&t); for (i = 0; i < n; i++) lf_tfc_add(&t, fin + 7 * i, fin + 7 * i + 3, fin[7 * i + 6]); lf_tfc_get(&t, fout); return t.n; }
    case 10: { lf_se3 X; lf_point_meas m; lf_point_blocks B; double Vi[9], T[36], u[6], dl[3]; int ok;
               memcpy(&X, in, sizeof X);
               m.mn = in + 15; m.mo = in + 18; m.In = in + 21; m.Io = in + 30;
               lf_ptmatch_blocks(&X, in + 12, &m, in[39], (int)in[40], &B);
               memcpy(out, &B, sizeof B);                                        /* V W bl Hpp bp: 72 doubles */
               out[72] = lf_ptmatch_chi2(&X, in + 12, &m, in[39], (int)in[40]);
               ok = lf_ptmatch_eliminate(&B, in[41], Vi, T, u);
               memcpy(out + 73, Vi, sizeof Vi); memcpy(out + 82, T, sizeof T); memcpy(out + 118, u, sizeof u);
               lf_ptmatch_backsub(&B, Vi, in + 42, dl); memcpy(out + 124, dl, sizeof dl);
               return ok; }
    case 11: { double U[9], sg[3], V[9]; lf_svd3(in, U, sg, V); memcpy(out, U, sizeof U); memcpy(out + 9, sg, sizeof sg); memcpy(out + 12, V, sizeof V); out[21] = lf_det3(in); return 1; }
  }
  return -1;
}

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
