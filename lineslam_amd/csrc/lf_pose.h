/* lf_pose.h -- geometric primitives of the pairwise motion solver, host + device, IEEE-only
 * (+,-,*,/,sqrt; no fma, no libm) so that gcc and hipcc (-ffp-contract=off) agree bit for bit.
 *
 * Reference code restated here:
 *   computeRelativeMotion_svd   src/line/motion.cpp:315-365   (3-line minimal solver, Zhang)
 *   q2r                         src/line/utils.cpp:1677-1694
 *   line inlier test            src/line/motion.cpp:688-699   (float transform, then mah-dist)
 *   EdgeSE3LineEndpts::computeError   src/line/edge_se3_lineendpts.cpp:146-189
 *   getTransformFromHybridMatchesG2O  src/transformation_estimation.cpp:218-461 (line edges only)
 * Third-party arithmetic that is not in the reference tree and is restated from its published
 * algorithm ("parity unpinned"): g2o's VertexSE3 update (translation + compact
 * quaternion, right-multiplied), numeric central-difference Jacobians (delta 1e-9), Huber kernel,
 * OptimizationAlgorithmLevenberg (tau 1e-5, Nielsen-style lambda update, <= 10 retries).  The
 * normal equations have an arrow structure (one 6-dof pose + one 6-dof landmark per line match);
 * they are solved exactly by eliminating the landmark blocks (Schur complement), which is what the
 * sparse Cholesky of g2o does implicitly.
 *
 * The sequential oracle (oracle/pair_oracle.c) and the HIP kernels (lf_pair.hip) both compose these
 * functions; what differs is the driver (sequential loops vs. lanes + ordered accumulation).
 */
#ifndef LF_POSE_H
#define LF_POSE_H

#include "lf_linalg.h"

typedef struct { double R[9]; double t[3]; } lf_se3;   /* x_out = R x_in + t */

LF_HD void lf_cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
LF_HD double lf_norm3(const double *a) { return lf_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* q2r(cv::Mat q), utils.cpp:1677-1694: q = (a,b,c,d), normalised first */
LF_HD void lf_q2r(const double *q, double *R) {
  double a = q[0], b = q[1], c = q[2], d = q[3];
  double nm = lf_sqrt(a * a + b * b + c * c + d * d);
  a = a / nm; b = b / nm; c = c / nm; d = d / nm;
  R[0] = a * a + b * b - c * c - d * d; R[1] = 2 * b * c - 2 * a * d;         R[2] = 2 * b * d + 2 * a * c;
  R[3] = 2 * b * c + 2 * a * d;         R[4] = a * a - b * b + c * c - d * d; R[5] = 2 * c * d - 2 * a * b;
  R[6] = 2 * b * d - 2 * a * c;         R[7] = 2 * c * d + 2 * a * b;         R[8] = a * a - b * b - c * c + d * d;
}

/* computeRelativeMotion_svd (motion.cpp:315-365) for n line pairs (n <= 3 used by RANSAC).
 * la / lb: n x 6 doubles (A,B) of the lines in frame a (query) and frame b (train);  x_b = R x_a + t. */
LF_HD int lf_rel_motion_lines(const double *la, const double *lb, int n, double *R, double *t) {
  double ua[9], ub[9], da[9], db[9], A4[16], V[16], w[4], q[4];
  double uu[9], udr[3], uui[9];
  int i, r, c, k;
  if (n < 2 || n > 3) return 0;
  for (i = 0; i < n; i++) {
    double l[3], m[3], s;
    for (k = 0; k < 3; k++) { l[k] = la[6 * i + 3 + k] - la[6 * i + k]; m[k] = (la[6 * i + k] + la[6 * i + 3 + k]) * 0.5; }
    s = 1 / lf_norm3(l);
    for (k = 0; k < 3; k++) ua[3 * i + k] = l[k] * s;
    lf_cross3(&ua[3 * i], m, &da[3 * i]);
    for (k = 0; k < 3; k++) { l[k] = lb[6 * i + 3 + k] - lb[6 * i + k]; m[k] = (lb[6 * i + k] + lb[6 * i + 3 + k]) * 0.5; }
    s = 1 / lf_norm3(l);
    for (k = 0; k < 3; k++) ub[3 * i + k] = l[k] * s;
    lf_cross3(&ub[3 * i], m, &db[3 * i]);
  }
  for (k = 0; k < 16; k++) A4[k] = 0;
  for (i = 0; i < n; i++) {
    double Ai[16], dm[3], sp[3];
    for (k = 0; k < 16; k++) Ai[k] = 0;
    for (k = 0; k < 3; k++) { dm[k] = ua[3 * i + k] - ub[3 * i + k]; sp[k] = ua[3 * i + k] + ub[3 * i + k]; }
    Ai[1] = dm[0]; Ai[2] = dm[1]; Ai[3] = dm[2];
    Ai[4] = ub[3 * i + 0] - ua[3 * i + 0]; Ai[8] = ub[3 * i + 1] - ua[3 * i + 1]; Ai[12] = ub[3 * i + 2] - ua[3 * i + 2];
    /* vec2SkewMat(sp) into rows 1..3, cols 1..3 */
    Ai[5] = 0;      Ai[6] = -sp[2]; Ai[7] = sp[1];
    Ai[9] = sp[2];  Ai[10] = 0;     Ai[11] = -sp[0];
    Ai[13] = -sp[1]; Ai[14] = sp[0]; Ai[15] = 0;
    for (r = 0; r < 4; r++)
      for (c = 0; c < 4; c++) {
        double s = 0;
        for (k = 0; k < 4; k++) s += Ai[4 * k + r] * Ai[4 * k + c];   /* (Ai^T Ai)[r][c] */
        A4[4 * r + c] = A4[4 * r + c] + s;
      }
  }
  lf_jacobi4(A4, V, w);
  for (k = 0; k < 4; k++) q[k] = V[4 * k + 3];      /* svd.u.col(3): smallest singular value */
  lf_q2r(q, R);
  for (k = 0; k < 9; k++) uu[k] = 0;
  for (k = 0; k < 3; k++) udr[k] = 0;
  for (i = 0; i < n; i++) {
    const double *u = &ub[3 * i];
    double S[9], Rd[3], v[3];
    S[0] = 0; S[1] = -u[2]; S[2] = u[1]; S[3] = u[2]; S[4] = 0; S[5] = -u[0]; S[6] = -u[1]; S[7] = u[0]; S[8] = 0;
    for (r = 0; r < 3; r++)
      for (c = 0; c < 3; c++) {
        double s = 0;
        for (k = 0; k < 3; k++) s += S[3 * r + k] * S[3 * c + k];     /* S S^T */
        uu[3 * r + c] = uu[3 * r + c] + s;
      }
    for (r = 0; r < 3; r++) Rd[r] = R[3 * r] * da[3 * i] + R[3 * r + 1] * da[3 * i + 1] + R[3 * r + 2] * da[3 * i + 2];
    for (r = 0; r < 3; r++) v[r] = db[3 * i + r] - Rd[r];
    for (r = 0; r < 3; r++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += S[3 * k + r] * v[k];               /* S^T v */
      udr[r] = udr[r] + s;
    }
  }
  /* uu.inv(): cv::Mat::inv (DECOMP_LU) returns a ZERO matrix for a singular input (three parallel lines), so the
   * reference goes on with t = 0 and still scores the hypothesis (motion.cpp:363) */
  if (!lf_inv3(uu, uui)) { t[0] = t[1] = t[2] = 0.0; return 1; }
  for (r = 0; r < 3; r++) t[r] = uui[3 * r] * udr[0] + uui[3 * r + 1] * udr[1] + uui[3 * r + 2] * udr[2];
  return 1;
}

/* mah_dist3d_pt_line (utils.cpp:761-822) */
LF_HD double lf_mah_dist(const double *pos, const double *c, const double *q1, const double *q2) {
  double x1 = pos[0], x2 = pos[1], x3 = pos[2];
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
  return lf_sqrt((t1 * t1 + t2 * t2 + t3 * t3) / (t4 * t4 + t5 * t5 + t6 * t6));
}

/* Line inlier test of getTransform_PtsLines_ransac (motion.cpp:688-699): the query end points are
 * transformed with the FLOAT matrix tf (Eigen::Matrix4f * Vector4f; evaluated left to right), then
 * the Mahalanobis distances to the train line's rndA / rndB are taken in double.                  */
LF_HD void lf_tf_point_f(const float *tf, const double *p, double *out) {
  float x = (float)p[0], y = (float)p[1], z = (float)p[2];
  int r;
  for (r = 0; r < 3; r++) out[r] = (double)(((tf[4 * r] * x + tf[4 * r + 1] * y) + tf[4 * r + 2] * z) + tf[4 * r + 3] * 1.0f);
}
LF_HD int lf_line_inlier(const float *tf, const double *qA, const double *qB, const double *tA, const double *tB,
                         const double *tDUa, const double *tDUb, double thr, double *sse_add) {
  double a[3], b[3], da, db;
  lf_tf_point_f(tf, qA, a);
  lf_tf_point_f(tf, qB, b);
  da = lf_mah_dist(tA, tDUa, a, b);
  db = lf_mah_dist(tB, tDUb, a, b);
  if (da < thr && db < thr) { *sse_add = da * da + db * db; return 1; }
  *sse_add = 0.0;
  return 0;
}

/* ---------------------------------------------------------------- LM refinement primitives */
/* EdgeSE3LineEndpts::computeError (edge_se3_lineendpts.cpp:146-189): PA,PB = landmark end points in
 * the camera frame; (Am,Bm) measured end points with whitening matrices Ma, Mb (= D^-1/2 U^T).   */
LF_HD void lf_line_edge_error(const double *Ma, const double *Mb, const double *Am, const double *Bm,
                              const double *PA, const double *PB, double *e) {
  int h, r;
  for (h = 0; h < 2; h++) {
    const double *M = h ? Mb : Ma, *m = h ? Bm : Am;
    double Ap[3], Bp[3], d[3], t;
    for (r = 0; r < 3; r++) {
      Ap[r] = M[3 * r] * (PA[0] - m[0]) + M[3 * r + 1] * (PA[1] - m[1]) + M[3 * r + 2] * (PA[2] - m[2]);
      Bp[r] = M[3 * r] * (PB[0] - m[0]) + M[3 * r + 1] * (PB[1] - m[1]) + M[3 * r + 2] * (PB[2] - m[2]);
      d[r] = Bp[r] - Ap[r];
    }
    t = -(Ap[0] * d[0] + Ap[1] * d[1] + Ap[2] * d[2]) / (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (r = 0; r < 3; r++) e[3 * h + r] = Ap[r] + t * d[r];
  }
}
/* world -> camera (cache->w2n() with identity sensor offset): R^T (p - t) */
LF_HD void lf_se3_inv_apply(const lf_se3 *X, const double *p, double *out) {
  double d0 = p[0] - X->t[0], d1 = p[1] - X->t[1], d2 = p[2] - X->t[2];
  int r;
  for (r = 0; r < 3; r++) out[r] = X->R[r] * d0 + X->R[3 + r] * d1 + X->R[6 + r] * d2;
}
/* g2o VertexSE3::oplusImpl: X <- X * fromVectorMQT(v): translation v[0:3], rotation = compact
 * quaternion v[3:6] (w = sqrt(1 - |q|^2), identity if negative)                                  */
LF_HD void lf_se3_oplus(const lf_se3 *X, const double *v, lf_se3 *out) {
  double w = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
  double dR[9], q[4];
  int r, c, k;
  if (w < 0) { for (k = 0; k < 9; k++) dR[k] = (k % 4 == 0) ? 1.0 : 0.0; }
  else {
    /* Eigen::Quaterniond(w,x,y,z).toRotationMatrix() */
    double tx, ty, tz, twx, twy, twz, txx, txy, txz, tyy, tyz, tzz;
    q[0] = lf_sqrt(w); q[1] = v[3]; q[2] = v[4]; q[3] = v[5];
    tx = 2.0 * q[1]; ty = 2.0 * q[2]; tz = 2.0 * q[3];
    twx = tx * q[0]; twy = ty * q[0]; twz = tz * q[0];
    txx = tx * q[1]; txy = ty * q[1]; txz = tz * q[1];
    tyy = ty * q[2]; tyz = tz * q[2]; tzz = tz * q[3];
    dR[0] = 1.0 - (tyy + tzz); dR[1] = txy - twz; dR[2] = txz + twy;
    dR[3] = txy + twz; dR[4] = 1.0 - (txx + tzz); dR[5] = tyz - twx;
    dR[6] = txz - twy; dR[7] = tyz + twx; dR[8] = 1.0 - (txx + tyy);
  }
  for (r = 0; r < 3; r++) {
    out->t[r] = X->R[3 * r] * v[0] + X->R[3 * r + 1] * v[1] + X->R[3 * r + 2] * v[2] + X->t[r];
    for (c = 0; c < 3; c++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += X->R[3 * r + k] * dR[3 * k + c];
      out->R[3 * r + c] = s;
    }
  }
}
/* Eigen::Quaterniond(Matrix3d) followed by normalisation (g2o::SE3Quat ctor) and back to a matrix:
 * the initial estimate of the older camera (transformation_estimation.cpp:114-118)               */
LF_HD void lf_rot_normalise(double *R) {
  double q[4], t = R[0] + R[4] + R[8];
  if (t > 0) {
    double s = lf_sqrt(t + 1.0);
    q[0] = 0.5 * s; s = 0.5 / s;
    q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
  } else {
    int i = 0, j, k;
    double s;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    j = (i + 1) % 3; k = (j + 1) % 3;
    s = lf_sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[1 + i] = 0.5 * s; s = 0.5 / s;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
    q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
    q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
  lf_q2r(q, R);
}

/* Huber robust kernel (g2o RobustKernelHuber): rho[0] = rho(e2), rho[1] = rho'(e2) */
LF_HD void lf_huber(double e2, double delta, int use, double *rho0, double *rho1) {
  double dsqr = delta * delta;
  if (!use || e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
  else { double sq = lf_sqrt(e2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

/* One line match k of the refinement graph: landmark L (6), newer-camera measurement (world ==
 * newer camera, fixed at identity) and older-camera measurement (pose X).  Outputs the robustified
 * chi2 of its two edges and, if Hout != 0, its blocks of the normal equations built from numeric
 * central-difference Jacobians (delta 1e-9, as g2o's BaseBinaryEdge::linearizeOplus):
 *   V (6x6) landmark block, W (6x6) pose-landmark block (rows pose), bl (6), Hpp (6x6), bp (6).
 * b = -J^T (rho' Omega) e  as in g2o (the system solved is H dx = b).                            */
typedef struct {
  const double *nA, *nB, *nMa, *nMb;   /* newer (query) line: end points, whitening matrices */
  const double *oA, *oB, *oMa, *oMb;   /* older (train) line */
} lf_line_meas;
typedef struct { double V[36], W[36], bl[6], Hpp[36], bp[6]; } lf_line_blocks;

LF_HD void lf_match_errors(const lf_se3 *X, const double *L, const lf_line_meas *m, double *en, double *eo) {
  double PA[3], PB[3];
  lf_line_edge_error(m->nMa, m->nMb, m->nA, m->nB, L, L + 3, en);
  lf_se3_inv_apply(X, L, PA);
  lf_se3_inv_apply(X, L + 3, PB);
  lf_line_edge_error(m->oMa, m->oMb, m->oA, m->oB, PA, PB, eo);
}
LF_HD double lf_match_chi2(const lf_se3 *X, const double *L, const lf_line_meas *m, double wgt, double hdelta, int huber) {
  double en[6], eo[6], c, r0, r1, s;
  int i;
  lf_match_errors(X, L, m, en, eo);
  c = 0; for (i = 0; i < 6; i++) c += en[i] * (wgt * en[i]);
  lf_huber(c, hdelta, huber, &r0, &r1);
  s = r0;
  c = 0; for (i = 0; i < 6; i++) c += eo[i] * (wgt * eo[i]);
  lf_huber(c, hdelta, huber, &r0, &r1);
  return s + r0;
}
/* XP: optional table of the twelve perturbed poses X (+) (+-1e-9 e_d) at [2 d + (minus ? 1 : 0)] -- they do not
 * depend on the match, so a caller with many matches computes them once (lf_perturbed_poses); NULL: computed here. */
LF_HD void lf_perturbed_poses(const lf_se3 *X, lf_se3 *XP) {
  int d, i;
  for (d = 0; d < 6; d++) {
    double v[6];
    for (i = 0; i < 6; i++) v[i] = 0;
    v[d] = 1e-9; lf_se3_oplus(X, v, &XP[2 * d]);
    v[d] = -1e-9; lf_se3_oplus(X, v, &XP[2 * d + 1]);
  }
}
LF_HD void lf_match_blocks_xp(const lf_se3 *X, const lf_se3 *XP, const double *L, const lf_line_meas *m, double wgt, double hdelta,
                              int huber, lf_line_blocks *B) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  double en[6], eo[6], Jn[36], Jo[36], Jp[36];   /* d e_n/dL, d e_o/dL, d e_o/dX  (row-major 6x6) */
  double c, r0, wn, wo;
  int d, i, j, k;
  lf_match_errors(X, L, m, en, eo);
  for (d = 0; d < 6; d++) {
    double Lp[6], ep[6], em[6], ep2[6], em2[6];
    for (i = 0; i < 6; i++) Lp[i] = L[i];
    Lp[d] = L[d] + delta;
    lf_match_errors(X, Lp, m, ep, ep2);
    Lp[d] = L[d] - delta;
    lf_match_errors(X, Lp, m, em, em2);
    for (i = 0; i < 6; i++) { Jn[6 * i + d] = scalar * (ep[i] - em[i]); Jo[6 * i + d] = scalar * (ep2[i] - em2[i]); }
  }
  for (d = 0; d < 6; d++) {
    double v[6], PA[3], PB[3], ep[6], em[6];
    lf_se3 Xp;
    for (i = 0; i < 6; i++) v[i] = 0;
    v[d] = delta;
    if (XP) Xp = XP[2 * d]; else lf_se3_oplus(X, v, &Xp);
    lf_se3_inv_apply(&Xp, L, PA); lf_se3_inv_apply(&Xp, L + 3, PB);
    lf_line_edge_error(m->oMa, m->oMb, m->oA, m->oB, PA, PB, ep);
    v[d] = -delta;
    if (XP) Xp = XP[2 * d + 1]; else lf_se3_oplus(X, v, &Xp);
    lf_se3_inv_apply(&Xp, L, PA); lf_se3_inv_apply(&Xp, L + 3, PB);
    lf_line_edge_error(m->oMa, m->oMb, m->oA, m->oB, PA, PB, em);
    for (i = 0; i < 6; i++) Jp[6 * i + d] = scalar * (ep[i] - em[i]);
  }
  c = 0; for (i = 0; i < 6; i++) c += en[i] * (wgt * en[i]);
  lf_huber(c, hdelta, huber, &r0, &wn);
  c = 0; for (i = 0; i < 6; i++) c += eo[i] * (wgt * eo[i]);
  lf_huber(c, hdelta, huber, &r0, &wo);
  wn = wn * wgt; wo = wo * wgt;
  for (i = 0; i < 6; i++) {
    double sbl_n = 0, sbl_o = 0, sbp = 0;
    for (k = 0; k < 6; k++) { sbl_n += Jn[6 * k + i] * (wn * en[k]); sbl_o += Jo[6 * k + i] * (wo * eo[k]); sbp += Jp[6 * k + i] * (wo * eo[k]); }
    B->bl[i] = -(sbl_n + sbl_o);
    B->bp[i] = -sbp;
    for (j = 0; j < 6; j++) {
      double vn = 0, vo = 0, hw = 0, hp = 0;
      for (k = 0; k < 6; k++) {
        vn += Jn[6 * k + i] * (wn * Jn[6 * k + j]);
        vo += Jo[6 * k + i] * (wo * Jo[6 * k + j]);
        hw += Jp[6 * k + i] * (wo * Jo[6 * k + j]);
        hp += Jp[6 * k + i] * (wo * Jp[6 * k + j]);
      }
      B->V[6 * i + j] = vn + vo;
      B->W[6 * i + j] = hw;
      B->Hpp[6 * i + j] = hp;
    }
  }
}
LF_HD void lf_match_blocks(const lf_se3 *X, const double *L, const lf_line_meas *m, double wgt, double hdelta,
                           int huber, lf_line_blocks *B) {
  lf_match_blocks_xp(X, 0, L, m, wgt, hdelta, huber, B);
}
/* Elimination of one landmark at damping lambda: Vi = (V + lambda I)^-1;  T = W Vi W^T (6x6),
 * u = W Vi bl (6).  Vi is kept for the back-substitution dl = Vi (bl - W^T dp).                  */
LF_HD int lf_match_eliminate(const lf_line_blocks *B, double lambda, double *Vi, double *T, double *u) {
  double A[36], I6[36], WV[36];
  int i, j, k;
  for (i = 0; i < 36; i++) { A[i] = B->V[i]; I6[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (i = 0; i < 6; i++) A[7 * i] += lambda;
  if (!lf_solve6(A, I6, 6)) return 0;
  for (i = 0; i < 36; i++) Vi[i] = I6[i];
  for (i = 0; i < 6; i++)
    for (j = 0; j < 6; j++) { double s = 0; for (k = 0; k < 6; k++) s += B->W[6 * i + k] * Vi[6 * k + j]; WV[6 * i + j] = s; }
  for (i = 0; i < 6; i++) {
    double s = 0;
    for (k = 0; k < 6; k++) s += WV[6 * i + k] * B->bl[k];
    u[i] = s;
    for (j = 0; j < 6; j++) { double s2 = 0; for (k = 0; k < 6; k++) s2 += WV[6 * i + k] * B->W[6 * j + k]; T[6 * i + j] = s2; }
  }
  return 1;
}
LF_HD void lf_match_backsub(const lf_line_blocks *B, const double *Vi, const double *dp, double *dl) {
  double r[6];
  int i, k;
  for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += B->W[6 * k + i] * dp[k]; r[i] = B->bl[i] - s; }
  for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += Vi[6 * i + k] * r[k]; dl[i] = s; }
}

/* float Matrix4f (row-major 16) <-> the pose of the OLDER camera in the newer frame
 * (transformation_estimation.cpp:226-232: vertex 0 is initialised with T^-1; :459 returns
 * estimate().cast<float>().inverse()).                                                          */
LF_HD void lf_tf_to_older_pose(const float *tf, lf_se3 *X) {
  double R[9], t[3];
  int r, c;
  for (r = 0; r < 3; r++) { for (c = 0; c < 3; c++) R[3 * r + c] = (double)tf[4 * r + c]; t[r] = (double)tf[4 * r + 3]; }
  for (r = 0; r < 3; r++) {
    for (c = 0; c < 3; c++) X->R[3 * r + c] = R[3 * c + r];
    X->t[r] = -(R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
  }
  lf_rot_normalise(X->R);
}
LF_HD void lf_older_pose_to_tf(const lf_se3 *X, float *tf) {
  float R[9], t[3];
  int r, c;
  for (r = 0; r < 9; r++) R[r] = (float)X->R[r];
  for (r = 0; r < 3; r++) t[r] = (float)X->t[r];
  for (r = 0; r < 3; r++) {
    for (c = 0; c < 3; c++) tf[4 * r + c] = R[3 * c + r];
    tf[4 * r + 3] = -((R[r] * t[0] + R[3 + r] * t[1]) + R[6 + r] * t[2]);
  }
  tf[12] = 0.0f; tf[13] = 0.0f; tf[14] = 0.0f; tf[15] = 1.0f;
}


/* ================================================================ lines-only RANSAC (a24)
 * computeRelativeMotion_Ransac (src/line/motion.cpp:367-526) + optimizeRelmotion (:98-139).
 * a = lines of the newer (query) node, b = the matched lines of the older (train) node; x_b = R x_a + t. */
/* r2q(cv::Mat R), utils.cpp:1696-1707 */
LF_HD void lf_r2q(const double *R, double *q) {
  double t = R[0] + R[4] + R[8];
  double r = lf_sqrt(1 + t);
  double s = 0.5 / r;
  q[0] = 0.5 * r;
  q[1] = (R[7] - R[5]) * s;
  q[2] = (R[2] - R[6]) * s;
  q[3] = (R[3] - R[1]) * s;
}
/* dist3d_pt_line(cv::Point3d X, A, B), utils.cpp:626-636 (EPS 1e-10, lineslam.h:37) */
LF_HD double lf_dist3d_pt_line(const double *X, const double *A, const double *B) {
  double AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]}, XA[3] = {X[0] - A[0], X[1] - A[1], X[2] - A[2]};
  double nAB = lf_norm3(AB), ax, inv, nv[3], d;
  if (nAB < 1e-10) return -1;
  ax = lf_norm3(XA);
  inv = 1 / nAB;
  nv[0] = (B[0] - A[0]) * inv; nv[1] = (B[1] - A[1]) * inv; nv[2] = (B[2] - A[2]) * inv;
  d = XA[0] * nv[0] + XA[1] * nv[1] + XA[2] * nv[2];
  return lf_sqrt(lf_fabs(ax * ax - d * d));
}
LF_HD void lf_rt_apply(const double *R, const double *t, const double *x, double *out) {       /* R x + t */
  int r;
  for (r = 0; r < 3; r++) out[r] = ((R[3 * r] * x[0] + R[3 * r + 1] * x[1]) + R[3 * r + 2] * x[2]) + t[r];
}
LF_HD void lf_rt_apply_inv(const double *R, const double *t, const double *x, double *out) {   /* R^T (x - t) */
  double d[3] = {x[0] - t[0], x[1] - t[1], x[2] - t[2]};
  int r;
  for (r = 0; r < 3; r++) out[r] = (R[r] * d[0] + R[3 + r] * d[1]) + R[6 + r] * d[2];
}
/* consensus test of motion.cpp:443-455 / 499-510; acos_fn = libm acos (reference flavour) or lf_acos */
#ifndef LF_ACOS
#define LF_ACOS lf_acos
#endif
LF_HD int lf_relmotion_inlier(const double *R, const double *t, const double *aA, const double *aB, const double *bA,
                              const double *bB, double distThresh, double angThresh) {
  double aA_[3], aB_[3], aAB[3] = {aA[0] - aB[0], aA[1] - aB[1], aA[2] - aB[2]}, bAB[3] = {bA[0] - bB[0], bA[1] - bB[1], bA[2] - bB[2]};
  double rab[3], z[3] = {0, 0, 0}, dist, dot, angle;
  lf_rt_apply(R, t, aA, aA_);
  lf_rt_apply(R, t, aB, aB_);
  dist = 0.5 * lf_dist3d_pt_line(aA_, bA, bB) + 0.5 * lf_dist3d_pt_line(aB_, bA, bB);
  lf_rt_apply(R, z, aAB, rab);
  dot = (rab[0] * bAB[0] + rab[1] * bAB[1]) + rab[2] * bAB[2];
  angle = 180 * LF_ACOS(lf_fabs(dot / lf_norm3(aAB) / lf_norm3(bAB))) / 3.14159265;
  return dist < distThresh && angle < angThresh;
}
/* one residual of costFun_optimizeRelmotion (motion.cpp:60-96, OPT_USE_MAHDIST) */
LF_HD double lf_relmotion_residual(const double *R, const double *t, const double *aA, const double *aB,
                                   const double *aDUa, const double *aDUb, const double *bA, const double *bB,
                                   const double *bDUa, const double *bDUb) {
  double Xa[3], Xb[3], Ya[3], Yb[3];
  lf_rt_apply(R, t, aA, Xa);
  lf_rt_apply(R, t, aB, Xb);
  lf_rt_apply_inv(R, t, bA, Ya);
  lf_rt_apply_inv(R, t, bB, Yb);
  return 0.25 * (lf_mah_dist(bA, bDUa, Xa, Xb) + lf_mah_dist(bB, bDUb, Xa, Xb) + lf_mah_dist(aA, aDUa, Ya, Yb) +
                 lf_mah_dist(aB, aDUb, Ya, Yb));
}
/* degeneracy test of a 3-line sample (motion.cpp:424-437): 1 if every pair is parallel within 5 deg */
LF_HD int lf_relmotion_degenerate(const double *la /* 3 x (A,B) */, double cos_thresh) {
  double u[9];
  int i, j;
  for (i = 0; i < 3; i++) {
    double l[3] = {la[6 * i + 3] - la[6 * i], la[6 * i + 4] - la[6 * i + 1], la[6 * i + 5] - la[6 * i + 2]};
    double inv = 1 / lf_norm3(l);
    u[3 * i] = l[0] * inv; u[3 * i + 1] = l[1] * inv; u[3 * i + 2] = l[2] * inv;
  }
  for (i = 0; i < 3; i++)
    for (j = i + 1; j < 3; j++)
      if (lf_fabs((u[3 * i] * u[3 * j] + u[3 * i + 1] * u[3 * j + 1]) + u[3 * i + 2] * u[3 * j + 2]) < cos_thresh) return 0;
  return 1;
}

/* ================================================================ point features (config 3)
 * Point side of getTransform_PtsLines_ransac: the caller supplies Node::feature_locations_3d_
 * (Eigen::Vector4f x,y,z,1; z = NaN without depth, src/node.cpp:952-1018) and the point matches.     */
typedef struct {
  double raster_cov_x, raster_cov_y;   /* (3 tan(58deg/640))^2, (3 tan(45deg/480))^2: misc.cpp:704-711, host libm */
  double sigma_depth;                  /* ParameterServer "sigma_depth" 0.01 (misc2.h:20-35) */
} lf_point_model;

LF_HD double lf_depth_covariance(double depth, double sigma_depth) {   /* misc2.h:20-35 */
  double sd = sigma_depth * depth * depth;
  return sd * sd;
}
/* errorFunction2 (src/misc.cpp:699-786): squared Mahalanobis distance of a point match under tf (query
 * -> train, the float matrix cast to double), with the isotropic-bound shortcut.  x1 = query point,
 * x2 = train point (4 floats each).  DBL_MAX = "certainly not an inlier".                          */
LF_HD double lf_error_function2(const float *x1, const float *x2, const float *tf, const lf_point_model *pm) {
  const double BIG = 1.7976931348623157e308;
  double T[16], mu1[3], mu2[3], m12[3], d[3], R[9], cov1[3], cov2[3], S[9], rhs[3], q;
  int r, c, k;
  if (x1[2] != x1[2] || x2[2] != x2[2]) return BIG;
  for (k = 0; k < 16; k++) T[k] = (double)tf[k];
  for (k = 0; k < 3; k++) { mu1[k] = (double)x1[k]; mu2[k] = (double)x2[k]; }
  for (r = 0; r < 3; r++) m12[r] = ((T[4 * r] * mu1[0] + T[4 * r + 1] * mu1[1]) + T[4 * r + 2] * mu1[2]) + T[4 * r + 3] * (double)x1[3];
  for (k = 0; k < 3; k++) d[k] = m12[k] - mu2[k];
  {
    double dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double s1 = lf_depth_covariance(mu1[2], pm->sigma_depth), s2 = lf_depth_covariance(mu2[2], pm->sigma_depth);
    if (s1 < pm->raster_cov_x) s1 = pm->raster_cov_x;
    if (s2 < pm->raster_cov_x) s2 = pm->raster_cov_x;
    if (dsq > 2.0 * (s1 + s2)) return BIG;
  }
  for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
  cov1[0] = 1 * pm->raster_cov_x * mu1[2]; cov1[1] = 1 * pm->raster_cov_y * mu1[2]; cov1[2] = lf_depth_covariance(mu1[2], pm->sigma_depth);
  cov2[0] = 1 * pm->raster_cov_x * mu2[2]; cov2[1] = 1 * pm->raster_cov_y * mu2[2]; cov2[2] = lf_depth_covariance(mu2[2], pm->sigma_depth);
  /* cov1_in_frame_2 = R^T cov1 R  (as written at misc.cpp:765), then + cov2 */
  for (r = 0; r < 3; r++)
    for (c = 0; c < 3; c++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += (R[3 * k + r] * cov1[k]) * R[3 * k + c];
      S[3 * r + c] = s + ((r == c) ? cov2[r] : 0.0);
    }
  if (d[2] != d[2]) d[2] = 0.0;
  for (k = 0; k < 3; k++) rhs[k] = d[k];
  if (!lf_solve3(S, rhs, 1)) return BIG;          /* Eigen LDLT solve in the reference */
  q = d[0] * rhs[0] + d[1] * rhs[1] + d[2] * rhs[2];
  if (!(q >= 0.0)) return BIG;
  return q;
}

/* projectPt3d2Ln3d_2 (utils.cpp:506-512) */
LF_HD void lf_project_pt_line(const double *P, const double *A, const double *B, double *out) {
  double AB[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, AP[3] = {P[0] - A[0], P[1] - A[1], P[2] - A[2]};
  double s = (AB[0] * AP[0] + AB[1] * AP[1] + AB[2] * AP[2]) / (AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2]);
  out[0] = A[0] + s * AB[0]; out[1] = A[1] + s * AB[1]; out[2] = A[2] + s * AB[2];
}

/* pcl::TransformationFromCorrespondences (PCL 1.7 common/transformation_from_correspondences.hpp), float
 * accumulators as in PCL; the final 3x3 SVD in double (lf_svd3).                                    */
typedef struct { int n; float wsum; float m1[3], m2[3], cov[9]; } lf_tfc;
LF_HD void lf_tfc_reset(lf_tfc *t) { int i; t->n = 0; t->wsum = 0.0f; for (i = 0; i < 3; i++) t->m1[i] = t->m2[i] = 0.0f; for (i = 0; i < 9; i++) t->cov[i] = 0.0f; }
LF_HD void lf_tfc_add(lf_tfc *t, const float *from, const float *to, float w) {
  float alpha, d1[3], d2[3];
  int r, c;
  if (w == 0.0f) return;
  ++t->n;
  t->wsum += w;
  alpha = w / t->wsum;
  for (r = 0; r < 3; r++) { d1[r] = from[r] - t->m1[r]; d2[r] = to[r] - t->m2[r]; }
  for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) t->cov[3 * r + c] = (1.0f - alpha) * (t->cov[3 * r + c] + alpha * (d2[r] * d1[c]));
  for (r = 0; r < 3; r++) { t->m1[r] += alpha * d1[r]; t->m2[r] += alpha * d2[r]; }
}
LF_HD void lf_tfc_get(const lf_tfc *t, float *tf) {
  double C[9], U[9], sg[3], V[9], R[9], s22 = 1.0;
  int r, c, k;
  for (k = 0; k < 9; k++) C[k] = (double)t->cov[k];
  lf_svd3(C, U, sg, V);
  if (lf_det3(U) * lf_det3(V) < 0.0) s22 = -1.0;
  for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) R[3 * r + c] = (U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1]) + s22 * U[3 * r + 2] * V[3 * c + 2];
  for (r = 0; r < 3; r++) {
    float rf[3] = {(float)R[3 * r], (float)R[3 * r + 1], (float)R[3 * r + 2]};
    tf[4 * r] = rf[0]; tf[4 * r + 1] = rf[1]; tf[4 * r + 2] = rf[2];
    tf[4 * r + 3] = t->m2[r] - ((rf[0] * t->m1[0] + rf[1] * t->m1[1]) + rf[2] * t->m1[2]);
  }
  tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
}

/* compPt3dCov, Eigen overload (utils.cpp:724-745), used for the information of point edges
 * (transformation_estimation.cpp:267,283): Matrix3f covariance, cast to double, inverted.          */
LF_HD int lf_point_information(const float *pt, double f, double sig_px, double c1, double c2, double c3, double *info) {
  double x = (double)pt[0], y = (double)pt[1], z = (double)pt[2];
  double sz = c1 * z * z + c2 * z + c3, s2 = sig_px * sig_px, sz2 = sz * sz;
  double j00 = z / f, j02 = x / z, j11 = z / f, j12 = y / z, C[9];
  int k;
  C[0] = (j00 * s2) * j00 + (j02 * sz2) * j02; C[1] = (j02 * sz2) * j12; C[2] = (j02 * sz2);
  C[3] = (j12 * sz2) * j02; C[4] = (j11 * s2) * j11 + (j12 * sz2) * j12; C[5] = (j12 * sz2);
  C[6] = sz2 * j02; C[7] = sz2 * j12; C[8] = sz2;
  for (k = 0; k < 9; k++) C[k] = (double)(float)C[k];        /* Eigen::Matrix3f, then .cast<double>() */
  return lf_inv3(C, info);
}

/* One point match of the refinement graph: landmark p (3), newer measurement mn with information In,
 * older measurement mo with information Io (EdgeSE3PointXYZ::computeError, edge_se3_ptxyz.cpp:84-90).  */
typedef struct { const double *mn, *mo, *In, *Io; } lf_point_meas;
typedef struct { double V[9], W[18], bl[3], Hpp[36], bp[6]; } lf_point_blocks;

LF_HD void lf_ptmatch_errors(const lf_se3 *X, const double *p, const lf_point_meas *m, double *en, double *eo) {
  double q[3];
  int k;
  for (k = 0; k < 3; k++) en[k] = p[k] - m->mn[k];
  lf_se3_inv_apply(X, p, q);
  for (k = 0; k < 3; k++) eo[k] = q[k] - m->mo[k];
}
LF_HD double lf_quad3(const double *e, const double *I) {
  double t0 = I[0] * e[0] + I[1] * e[1] + I[2] * e[2], t1 = I[3] * e[0] + I[4] * e[1] + I[5] * e[2], t2 = I[6] * e[0] + I[7] * e[1] + I[8] * e[2];
  return e[0] * t0 + e[1] * t1 + e[2] * t2;
}
LF_HD double lf_ptmatch_chi2(const lf_se3 *X, const double *p, const lf_point_meas *m, double hdelta, int huber) {
  double en[3], eo[3], r0, r1, s;
  lf_ptmatch_errors(X, p, m, en, eo);
  lf_huber(lf_quad3(en, m->In), hdelta, huber, &r0, &r1);
  s = r0;
  lf_huber(lf_quad3(eo, m->Io), hdelta, huber, &r0, &r1);
  return s + r0;
}
LF_HD void lf_ptmatch_blocks_xp(const lf_se3 *X, const lf_se3 *XP, const double *p, const lf_point_meas *m, double hdelta, int huber,
                                lf_point_blocks *B) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  double en[3], eo[3], Jn[9], Jo[9], Jp[18], wn, wo, r0, On[9], Oo[9], One[3], Ooe[3];
  int d, i, j, k;
  lf_ptmatch_errors(X, p, m, en, eo);
  for (d = 0; d < 3; d++) {
    double pp[3], a[3], b[3], a2[3], b2[3];
    for (i = 0; i < 3; i++) pp[i] = p[i];
    pp[d] = p[d] + delta; lf_ptmatch_errors(X, pp, m, a, a2);
    pp[d] = p[d] - delta; lf_ptmatch_errors(X, pp, m, b, b2);
    for (i = 0; i < 3; i++) { Jn[3 * i + d] = scalar * (a[i] - b[i]); Jo[3 * i + d] = scalar * (a2[i] - b2[i]); }
  }
  for (d = 0; d < 6; d++) {
    double v[6], q[3], a[3], b[3];
    lf_se3 Xp;
    for (i = 0; i < 6; i++) v[i] = 0;
    v[d] = delta; if (XP) Xp = XP[2 * d]; else lf_se3_oplus(X, v, &Xp);
    lf_se3_inv_apply(&Xp, p, q); for (i = 0; i < 3; i++) a[i] = q[i] - m->mo[i];
    v[d] = -delta; if (XP) Xp = XP[2 * d + 1]; else lf_se3_oplus(X, v, &Xp);
    lf_se3_inv_apply(&Xp, p, q); for (i = 0; i < 3; i++) b[i] = q[i] - m->mo[i];
    for (i = 0; i < 3; i++) Jp[6 * i + d] = scalar * (a[i] - b[i]);
  }
  lf_huber(lf_quad3(en, m->In), hdelta, huber, &r0, &wn);
  lf_huber(lf_quad3(eo, m->Io), hdelta, huber, &r0, &wo);
  for (i = 0; i < 9; i++) { On[i] = wn * m->In[i]; Oo[i] = wo * m->Io[i]; }
  for (i = 0; i < 3; i++) {
    One[i] = On[3 * i] * en[0] + On[3 * i + 1] * en[1] + On[3 * i + 2] * en[2];
    Ooe[i] = Oo[3 * i] * eo[0] + Oo[3 * i + 1] * eo[1] + Oo[3 * i + 2] * eo[2];
  }
  for (i = 0; i < 3; i++) {
    double s1 = 0, s2 = 0;
    for (k = 0; k < 3; k++) { s1 += Jn[3 * k + i] * One[k]; s2 += Jo[3 * k + i] * Ooe[k]; }
    B->bl[i] = -(s1 + s2);
  }
  for (i = 0; i < 6; i++) { double s3 = 0; for (k = 0; k < 3; k++) s3 += Jp[6 * k + i] * Ooe[k]; B->bp[i] = -s3; }
  {
    double OJn[9], OJo[9], OJp[18];   /* Omega * J */
    for (i = 0; i < 3; i++) {
      for (j = 0; j < 3; j++) {
        double a = 0, b = 0;
        for (k = 0; k < 3; k++) { a += On[3 * i + k] * Jn[3 * k + j]; b += Oo[3 * i + k] * Jo[3 * k + j]; }
        OJn[3 * i + j] = a; OJo[3 * i + j] = b;
      }
      for (j = 0; j < 6; j++) { double cc = 0; for (k = 0; k < 3; k++) cc += Oo[3 * i + k] * Jp[6 * k + j]; OJp[6 * i + j] = cc; }
    }
    for (i = 0; i < 3; i++)
      for (j = 0; j < 3; j++) {
        double a = 0, b = 0;
        for (k = 0; k < 3; k++) { a += Jn[3 * k + i] * OJn[3 * k + j]; b += Jo[3 * k + i] * OJo[3 * k + j]; }
        B->V[3 * i + j] = a + b;
      }
    for (i = 0; i < 6; i++) {
      for (j = 0; j < 3; j++) { double a = 0; for (k = 0; k < 3; k++) a += Jp[6 * k + i] * OJo[3 * k + j]; B->W[3 * i + j] = a; }
      for (j = 0; j < 6; j++) { double a = 0; for (k = 0; k < 3; k++) a += Jp[6 * k + i] * OJp[6 * k + j]; B->Hpp[6 * i + j] = a; }
    }
  }
}
LF_HD void lf_ptmatch_blocks(const lf_se3 *X, const double *p, const lf_point_meas *m, double hdelta, int huber,
                             lf_point_blocks *B) {
  lf_ptmatch_blocks_xp(X, 0, p, m, hdelta, huber, B);
}
LF_HD int lf_ptmatch_eliminate(const lf_point_blocks *B, double lambda, double *Vi, double *T, double *u) {
  double A[9], WV[18];
  int i, j, k;
  for (i = 0; i < 9; i++) A[i] = B->V[i];
  for (i = 0; i < 3; i++) A[4 * i] += lambda;
  if (!lf_inv3(A, Vi)) return 0;
  for (i = 0; i < 6; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += B->W[3 * i + k] * Vi[3 * k + j]; WV[3 * i + j] = s; }
  for (i = 0; i < 6; i++) {
    double s = 0;
    for (k = 0; k < 3; k++) s += WV[3 * i + k] * B->bl[k];
    u[i] = s;
    for (j = 0; j < 6; j++) { double s2 = 0; for (k = 0; k < 3; k++) s2 += WV[3 * i + k] * B->W[3 * j + k]; T[6 * i + j] = s2; }
  }
  return 1;
}
LF_HD void lf_ptmatch_backsub(const lf_point_blocks *B, const double *Vi, const double *dp, double *dl) {
  double r[3];
  int i, k;
  for (i = 0; i < 3; i++) { double s = 0; for (k = 0; k < 6; k++) s += B->W[3 * k + i] * dp[k]; r[i] = B->bl[i] - s; }
  for (i = 0; i < 3; i++) dl[i] = Vi[3 * i] * r[0] + Vi[3 * i + 1] * r[1] + Vi[3 * i + 2] * r[2];
}

#endif /* LF_POSE_H */
