/* oracle/o_pose.h -- TEST INFRASTRUCTURE ONLY: the ORACLE'S OWN statement of the geometric primitives of the pairwise motion
 * solver.  Until round 5 pair_oracle.c compiled the product's lineslam_amd/csrc/lf_pose.h, so the sequential checker and the HIP
 * kernels shared every formula; this file states them again from the reference's sources (and, for g2o / PCL / OpenCV / Eigen,
 * which are not in the reference tree, from their published algorithms), in the oracle's own shape: small dense helpers
 * (o_mv3, o_dot3 ...) and loops over rows / edges instead of the product's unrolled expressions.  IEEE double, + - * / sqrt only;
 * where a sum has more than two terms its association is the reference expression's (left to right), because the GPU tests hold
 * the kernels to this file BIT FOR BIT.  tests/test_oracle_pose_primitives.py holds every function here against the product's
 * (exported by product_hooks.c) on random inputs, bit for bit -- two statements, one result.  The third voice stays
 * oracle/pose_indep.py (numpy / scipy, different algorithms: SVD, dense solves, full-state LM).
 *
 * Reference code restated:
 *   computeRelativeMotion_svd          src/line/motion.cpp:315-365     q2r / r2q          src/line/utils.cpp:1677-1707
 *   mah_dist3d_pt_line                 src/line/utils.cpp:761-822      dist3d_pt_line     src/line/utils.cpp:626-636
 *   line inlier test                   src/line/motion.cpp:688-699     consensus test     src/line/motion.cpp:443-455
 *   costFun_optimizeRelmotion          src/line/motion.cpp:60-96       degeneracy test    src/line/motion.cpp:424-437
 *   EdgeSE3LineEndpts::computeError    src/line/edge_se3_lineendpts.cpp:146-189
 *   EdgeSE3PointXYZ::computeError      src/line/edge_se3_ptxyz.cpp:84-90
 *   getTransformFromHybridMatchesG2O   src/transformation_estimation.cpp:218-461 (vertex set-up :226-232, result :459)
 *   errorFunction2                     src/misc.cpp:699-786            depth_covariance   src/misc2.h:20-35
 *   projectPt3d2Ln3d_2                 src/line/utils.cpp:506-512      compPt3dCov        src/line/utils.cpp:724-745
 *   getTransform_Lns_Pts_pcl (PCL's TransformationFromCorrespondences)  src/line/motion.cpp:530-579
 */
#ifndef O_POSE_H
#define O_POSE_H
#include <math.h>
#include "o_linalg.h"

#ifdef ORACLE_LFMATH
#include "../lineslam_amd/csrc/lf_math.h"     /* the device-side acos, so that this flavour equals the kernels bit for bit */
#define O_ACOS lf_acos
#else
#define O_ACOS acos                           /* reference flavour: host libm, as motion.cpp:449 */
#endif

typedef struct { double R[9]; double t[3]; } o_se3;          /* x_out = R x_in + t */

/* ---- small dense helpers (every sum left to right) */
static double o_dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double o_len3(const double *a) { return sqrt(o_dot3(a, a)); }
static void o_cross(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void o_sub3(const double *a, const double *b, double *c) { int k; for (k = 0; k < 3; k++) c[k] = a[k] - b[k]; }
/* row r of a row-major 3x3 times a vector */
static double o_row3(const double *M, int r, const double *v) { return M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2]; }
/* column r of a row-major 3x3 times a vector (= row r of the transpose) */
static double o_col3(const double *M, int r, const double *v) { return M[r] * v[0] + M[3 + r] * v[1] + M[6 + r] * v[2]; }

/* q2r, utils.cpp:1677-1694: unit quaternion (a, b, c, d) of the normalised input */
static void o_q2r(const double *q, double *R) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double a = q[0] / n, b = q[1] / n, c = q[2] / n, d = q[3] / n;
  R[0] = a * a + b * b - c * c - d * d; R[1] = 2 * b * c - 2 * a * d;         R[2] = 2 * b * d + 2 * a * c;
  R[3] = 2 * b * c + 2 * a * d;         R[4] = a * a - b * b + c * c - d * d; R[5] = 2 * c * d - 2 * a * b;
  R[6] = 2 * b * d - 2 * a * c;         R[7] = 2 * c * d + 2 * a * b;         R[8] = a * a - b * b - c * c + d * d;
}
/* r2q, utils.cpp:1696-1707 */
static void o_r2q(const double *R, double *q) {
  const double tr = R[0] + R[4] + R[8];
  const double r = sqrt(1 + tr), s = 0.5 / r;
  q[0] = 0.5 * r;
  q[1] = (R[7] - R[5]) * s;
  q[2] = (R[2] - R[6]) * s;
  q[3] = (R[3] - R[1]) * s;
}

/* Pluecker members of a segment (A, B): unit direction u = (B - A) / |B - A| (a reciprocal, then products, as the reference's
 * `l * (1 / norm(l))`), moment d = u x midpoint */
static void o_line_members(const double *AB, double *u, double *d) {
  double l[3], m[3], s;
  int k;
  for (k = 0; k < 3; k++) { l[k] = AB[3 + k] - AB[k]; m[k] = (AB[k] + AB[3 + k]) * 0.5; }
  s = 1 / o_len3(l);
  for (k = 0; k < 3; k++) u[k] = l[k] * s;
  o_cross(u, m, d);
}
/* computeRelativeMotion_svd (motion.cpp:315-365) for n = 2 or 3 line pairs; la / lb: n x (A, B) in frame a / b; x_b = R x_a + t.
 * The reference takes svd.u.col(3) of A = sum A_i^T A_i (symmetric PSD: the eigenvector of the smallest eigenvalue; Jacobi here)
 * and solves the translation from sum [u_b]x [u_b]x^T with cv::Mat::inv (LU; a singular matrix gives a ZERO inverse: t = 0). */
static int o_rel_motion_lines(const double *la, const double *lb, int n, double *R, double *t) {
  double ua[9], ub[9], da[9], db[9], A4[16], V[16], w[4], q[4], uu[9], rhs[3], uui[9];
  int i, r, c, k;
  if (n < 2 || n > 3) return 0;
  for (i = 0; i < n; i++) { o_line_members(la + 6 * i, ua + 3 * i, da + 3 * i); o_line_members(lb + 6 * i, ub + 3 * i, db + 3 * i); }
  for (k = 0; k < 16; k++) A4[k] = 0;
  for (i = 0; i < n; i++) {
    double Ai[16], sp[3];
    for (k = 0; k < 16; k++) Ai[k] = 0;
    for (k = 0; k < 3; k++) {
      Ai[1 + k] = ua[3 * i + k] - ub[3 * i + k];              /* first row: (0, ua - ub) */
      Ai[4 * (k + 1)] = ub[3 * i + k] - ua[3 * i + k];        /* first column: (0, ub - ua) */
      sp[k] = ua[3 * i + k] + ub[3 * i + k];
    }
    Ai[6] = -sp[2]; Ai[7] = sp[1]; Ai[9] = sp[2]; Ai[11] = -sp[0]; Ai[13] = -sp[1]; Ai[14] = sp[0];     /* [ua + ub]x in rows / columns 1..3 */
    for (r = 0; r < 4; r++)
      for (c = 0; c < 4; c++) {
        double s = 0;
        for (k = 0; k < 4; k++) s += Ai[4 * k + r] * Ai[4 * k + c];
        A4[4 * r + c] = A4[4 * r + c] + s;
      }
  }
  o_jacobi(4, A4, V, w);
  for (k = 0; k < 4; k++) q[k] = V[4 * k + 3];
  o_q2r(q, R);
  for (k = 0; k < 9; k++) uu[k] = 0;
  for (k = 0; k < 3; k++) rhs[k] = 0;
  for (i = 0; i < n; i++) {
    const double *u = ub + 3 * i;
    const double S[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
    double Rd[3], v[3];
    for (r = 0; r < 3; r++)
      for (c = 0; c < 3; c++) {
        double s = 0;
        for (k = 0; k < 3; k++) s += S[3 * r + k] * S[3 * c + k];
        uu[3 * r + c] = uu[3 * r + c] + s;
      }
    for (r = 0; r < 3; r++) Rd[r] = o_row3(R, r, da + 3 * i);
    for (r = 0; r < 3; r++) v[r] = db[3 * i + r] - Rd[r];
    for (r = 0; r < 3; r++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += S[3 * k + r] * v[k];
      rhs[r] = rhs[r] + s;
    }
  }
  if (!o_inv3(uu, uui)) { t[0] = t[1] = t[2] = 0.0; return 1; }
  for (r = 0; r < 3; r++) t[r] = o_row3(uui, r, rhs);
  return 1;
}

/* mah_dist3d_pt_line (utils.cpp:761-822): the point `pos` with whitening matrix c = D^-1/2 U^T against the line (q1, q2):
 * |a x b| / |a - b| of the whitened differences, the denominator as the reference writes it (term by term) */
static double o_mah_dist(const double *pos, const double *c, const double *q1, const double *q2) {
  double da[3], db[3], a[3], b[3], cr[3], den[3];
  int r;
  o_sub3(pos, q1, da);
  o_sub3(pos, q2, db);
  for (r = 0; r < 3; r++) { a[r] = o_row3(c, r, da); b[r] = o_row3(c, r, db); }
  cr[0] = a[0] * b[1] - a[1] * b[0];
  cr[1] = a[0] * b[2] - a[2] * b[0];
  cr[2] = a[1] * b[2] - a[2] * b[1];
  for (r = 0; r < 3; r++)
    den[r] = c[3 * r] * da[0] - c[3 * r] * db[0] + c[3 * r + 1] * da[1] - c[3 * r + 1] * db[1] + c[3 * r + 2] * da[2] - c[3 * r + 2] * db[2];
  return sqrt((cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / (den[0] * den[0] + den[1] * den[1] + den[2] * den[2]));
}
/* Eigen::Matrix4f * Vector4f(x, y, z, 1) in float, evaluated left to right (motion.cpp:690-691) */
static void o_tf_point_f(const float *tf, const double *p, double *out) {
  const float x = (float)p[0], y = (float)p[1], z = (float)p[2];
  int r;
  for (r = 0; r < 3; r++) out[r] = (double)(((tf[4 * r] * x + tf[4 * r + 1] * y) + tf[4 * r + 2] * z) + tf[4 * r + 3] * 1.0f);
}
/* the line inlier test of getTransform_PtsLines_ransac (motion.cpp:688-699) */
static int o_line_inlier(const float *tf, const double *qA, const double *qB, const double *tA, const double *tB,
                         const double *tDUa, const double *tDUb, double thr, double *sse_add) {
  double a[3], b[3], da, db;
  o_tf_point_f(tf, qA, a);
  o_tf_point_f(tf, qB, b);
  da = o_mah_dist(tA, tDUa, a, b);
  db = o_mah_dist(tB, tDUb, a, b);
  *sse_add = (da < thr && db < thr) ? da * da + db * db : 0.0;
  return da < thr && db < thr;
}

/* ---------------------------------------------------------------- the refinement graph (g2o restated) */
/* EdgeSE3LineEndpts::computeError (edge_se3_lineendpts.cpp:146-189): for each measured end point m with whitening matrix M,
 * the perpendicular from the whitened m to the whitened landmark line (PA, PB) */
static void o_line_edge_error(const double *Ma, const double *Mb, const double *Am, const double *Bm, const double *PA, const double *PB, double *e) {
  int h, r;
  for (h = 0; h < 2; h++) {
    const double *M = h ? Mb : Ma, *m = h ? Bm : Am;
    double dA[3], dB[3], Ap[3], Bp[3], d[3], t;
    o_sub3(PA, m, dA);
    o_sub3(PB, m, dB);
    for (r = 0; r < 3; r++) { Ap[r] = o_row3(M, r, dA); Bp[r] = o_row3(M, r, dB); d[r] = Bp[r] - Ap[r]; }
    t = -o_dot3(Ap, d) / o_dot3(d, d);
    for (r = 0; r < 3; r++) e[3 * h + r] = Ap[r] + t * d[r];
  }
}
/* world -> camera: R^T (p - t) */
static void o_se3_inv_apply(const o_se3 *X, const double *p, double *out) {
  double d[3];
  int r;
  o_sub3(p, X->t, d);
  for (r = 0; r < 3; r++) out[r] = o_col3(X->R, r, d);
}
/* g2o VertexSE3::oplusImpl: X <- X * (translation v[0:3], rotation from the compact quaternion v[3:6], w = sqrt(1 - |q|^2);
 * identity rotation if that is negative); the rotation matrix as Eigen's Quaterniond::toRotationMatrix */
static void o_se3_oplus(const o_se3 *X, const double *v, o_se3 *out) {
  const double ww = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
  double D[9];
  int r, c, k;
  if (ww < 0) { for (k = 0; k < 9; k++) D[k] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0; }
  else {
    const double qw = sqrt(ww), qx = v[3], qy = v[4], qz = v[5];
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    D[0] = 1.0 - (tyy + tzz); D[1] = txy - twz;         D[2] = txz + twy;
    D[3] = txy + twz;         D[4] = 1.0 - (txx + tzz); D[5] = tyz - twx;
    D[6] = txz - twy;         D[7] = tyz + twx;         D[8] = 1.0 - (txx + tyy);
  }
  for (r = 0; r < 3; r++) {
    out->t[r] = X->R[3 * r] * v[0] + X->R[3 * r + 1] * v[1] + X->R[3 * r + 2] * v[2] + X->t[r];
    for (c = 0; c < 3; c++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += X->R[3 * r + k] * D[3 * k + c];
      out->R[3 * r + c] = s;
    }
  }
}
/* Eigen::Quaterniond(Matrix3d) + normalisation (g2o::SE3Quat's constructor) and back to a matrix */
static void o_rot_normalise(double *R) {
  double q[4];
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q[0] = 0.5 * s; s = 0.5 / s;
    q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
  } else {
    int i = 0, j, k;
    double s;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    j = (i + 1) % 3; k = (j + 1) % 3;
    s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[1 + i] = 0.5 * s; s = 0.5 / s;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
    q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
    q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
  o_q2r(q, R);
}
/* g2o RobustKernelHuber: rho(e2), rho'(e2) */
static void o_huber(double e2, double delta, int use, double *rho0, double *rho1) {
  const double dsqr = delta * delta;
  if (!use || e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; return; }
  { const double sq = sqrt(e2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

typedef struct { const double *nA, *nB, *nMa, *nMb, *oA, *oB, *oMa, *oMb; } o_line_meas;   /* newer (query) / older (train) line */
typedef struct { double V[36], W[36], bl[6], Hpp[36], bp[6]; } o_line_blocks;

/* both edges of one line match: the landmark L against the newer camera (the world frame) and, through X^-1, the older one */
static void o_match_errors(const o_se3 *X, const double *L, const o_line_meas *m, double *en, double *eo) {
  double PA[3], PB[3];
  o_line_edge_error(m->nMa, m->nMb, m->nA, m->nB, L, L + 3, en);
  o_se3_inv_apply(X, L, PA);
  o_se3_inv_apply(X, L + 3, PB);
  o_line_edge_error(m->oMa, m->oMb, m->oA, m->oB, PA, PB, eo);
}
static double o_edge_chi2(const double *e, double wgt) { double c = 0; int i; for (i = 0; i < 6; i++) c += e[i] * (wgt * e[i]); return c; }
static double o_match_chi2(const o_se3 *X, const double *L, const o_line_meas *m, double wgt, double hd, int hub) {
  double en[6], eo[6], rn, ro, w;
  o_match_errors(X, L, m, en, eo);
  o_huber(o_edge_chi2(en, wgt), hd, hub, &rn, &w);
  o_huber(o_edge_chi2(eo, wgt), hd, hub, &ro, &w);
  return rn + ro;
}
/* The blocks of one match of the normal equations from numeric central differences (g2o BaseBinaryEdge::linearizeOplus: delta
 * 1e-9, scalar 1 / 2 delta): V = Jn^T wn Jn + Jo^T wo Jo (landmark), W = Jp^T wo Jo (pose x landmark), Hpp = Jp^T wo Jp,
 * bl = -(Jn^T wn en + Jo^T wo eo), bp = -Jp^T wo eo; w = rho' * the edge weight. */
static void o_match_blocks(const o_se3 *X, const double *L, const o_line_meas *m, double wgt, double hd, int hub, o_line_blocks *B) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  double en[6], eo[6], Jn[36], Jo[36], Jp[36], rho, wn, wo;
  int d, i, j, k;
  o_match_errors(X, L, m, en, eo);
  for (d = 0; d < 6; d++) {                    /* along the landmark */
    double Lq[6], nP[6], nM[6], oP[6], oM[6];
    for (i = 0; i < 6; i++) Lq[i] = L[i];
    Lq[d] = L[d] + delta; o_match_errors(X, Lq, m, nP, oP);
    Lq[d] = L[d] - delta; o_match_errors(X, Lq, m, nM, oM);
    for (i = 0; i < 6; i++) { Jn[6 * i + d] = scalar * (nP[i] - nM[i]); Jo[6 * i + d] = scalar * (oP[i] - oM[i]); }
  }
  for (d = 0; d < 6; d++) {                    /* along the pose */
    double v[6], eP[6], eM[6], PA[3], PB[3];
    o_se3 Xq;
    int sgn;
    for (sgn = 0; sgn < 2; sgn++) {
      for (i = 0; i < 6; i++) v[i] = 0;
      v[d] = sgn ? -delta : delta;
      o_se3_oplus(X, v, &Xq);
      o_se3_inv_apply(&Xq, L, PA);
      o_se3_inv_apply(&Xq, L + 3, PB);
      o_line_edge_error(m->oMa, m->oMb, m->oA, m->oB, PA, PB, sgn ? eM : eP);
    }
    for (i = 0; i < 6; i++) Jp[6 * i + d] = scalar * (eP[i] - eM[i]);
  }
  o_huber(o_edge_chi2(en, wgt), hd, hub, &rho, &wn);
  o_huber(o_edge_chi2(eo, wgt), hd, hub, &rho, &wo);
  wn = wn * wgt; wo = wo * wgt;
  for (i = 0; i < 6; i++) {
    double gn = 0, go = 0, gp = 0;
    for (k = 0; k < 6; k++) { gn += Jn[6 * k + i] * (wn * en[k]); go += Jo[6 * k + i] * (wo * eo[k]); gp += Jp[6 * k + i] * (wo * eo[k]); }
    B->bl[i] = -(gn + go);
    B->bp[i] = -gp;
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
/* Schur elimination of one landmark at damping lambda: Vi = (V + lambda I)^-1 (LU with partial pivoting on six right-hand
 * sides), T = W Vi W^T, u = W Vi bl */
static int o_match_eliminate(const o_line_blocks *B, double lambda, double *Vi, double *T, double *u) {
  double A[36], E[36], WV[36];
  int i, j, k;
  for (i = 0; i < 36; i++) { A[i] = B->V[i]; E[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (i = 0; i < 6; i++) A[7 * i] += lambda;
  if (!o_lu_solve(6, A, 6, E)) return 0;
  for (i = 0; i < 36; i++) Vi[i] = E[i];
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
/* dl = Vi (bl - W^T dp) */
static void o_match_backsub(const o_line_blocks *B, const double *Vi, const double *dp, double *dl) {
  double r[6];
  int i, k;
  for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += B->W[6 * k + i] * dp[k]; r[i] = B->bl[i] - s; }
  for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 6; k++) s += Vi[6 * i + k] * r[k]; dl[i] = s; }
}
/* Matrix4f (row-major 16 floats, newer -> older) <-> the pose of the OLDER camera in the newer frame
 * (transformation_estimation.cpp:226-232: vertex 0 starts at T^-1; :459 returns estimate().cast<float>().inverse()) */
static void o_tf_to_older_pose(const float *tf, o_se3 *X) {
  double R[9], t[3];
  int r, c;
  for (r = 0; r < 3; r++) { for (c = 0; c < 3; c++) R[3 * r + c] = (double)tf[4 * r + c]; t[r] = (double)tf[4 * r + 3]; }
  for (r = 0; r < 3; r++) {
    for (c = 0; c < 3; c++) X->R[3 * r + c] = R[3 * c + r];
    X->t[r] = -o_col3(R, r, t);
  }
  o_rot_normalise(X->R);
}
static void o_older_pose_to_tf(const o_se3 *X, float *tf) {
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

/* ---------------------------------------------------------------- lines-only RANSAC (computeRelativeMotion_Ransac) */
/* dist3d_pt_line(X, A, B), utils.cpp:626-636 (EPS 1e-10, lineslam.h:37) */
static double o_dist3d_pt_line(const double *X, const double *A, const double *B) {
  double AB[3], XA[3], nv[3], nAB, ax, inv, d;
  int k;
  o_sub3(A, B, AB);
  o_sub3(X, A, XA);
  nAB = o_len3(AB);
  if (nAB < 1e-10) return -1;
  ax = o_len3(XA);
  inv = 1 / nAB;
  for (k = 0; k < 3; k++) nv[k] = (B[k] - A[k]) * inv;
  d = o_dot3(XA, nv);
  return sqrt(fabs(ax * ax - d * d));
}
static void o_rt_apply(const double *R, const double *t, const double *x, double *out) {       /* R x + t */
  int r;
  for (r = 0; r < 3; r++) out[r] = ((R[3 * r] * x[0] + R[3 * r + 1] * x[1]) + R[3 * r + 2] * x[2]) + t[r];
}
static void o_rt_apply_inv(const double *R, const double *t, const double *x, double *out) {   /* R^T (x - t) */
  double d[3];
  int r;
  o_sub3(x, t, d);
  for (r = 0; r < 3; r++) out[r] = (R[r] * d[0] + R[3 + r] * d[1]) + R[6 + r] * d[2];
}
/* the consensus test of motion.cpp:443-455 / 499-510 */
static int o_relmotion_inlier(const double *R, const double *t, const double *aA, const double *aB, const double *bA, const double *bB,
                              double distThresh, double angThresh) {
  double pa[3], pb[3], va[3], vb[3], rv[3], zero[3] = {0, 0, 0}, dist, dot, angle;
  o_rt_apply(R, t, aA, pa);
  o_rt_apply(R, t, aB, pb);
  dist = 0.5 * o_dist3d_pt_line(pa, bA, bB) + 0.5 * o_dist3d_pt_line(pb, bA, bB);
  o_sub3(aA, aB, va);
  o_sub3(bA, bB, vb);
  o_rt_apply(R, zero, va, rv);
  dot = (rv[0] * vb[0] + rv[1] * vb[1]) + rv[2] * vb[2];
  angle = 180 * O_ACOS(fabs(dot / o_len3(va) / o_len3(vb))) / 3.14159265;
  return dist < distThresh && angle < angThresh;
}
/* one residual of costFun_optimizeRelmotion (motion.cpp:60-96, OPT_USE_MAHDIST) */
static double o_relmotion_residual(const double *R, const double *t, const double *aA, const double *aB, const double *aDUa, const double *aDUb,
                                   const double *bA, const double *bB, const double *bDUa, const double *bDUb) {
  double Xa[3], Xb[3], Ya[3], Yb[3];
  o_rt_apply(R, t, aA, Xa);
  o_rt_apply(R, t, aB, Xb);
  o_rt_apply_inv(R, t, bA, Ya);
  o_rt_apply_inv(R, t, bB, Yb);
  return 0.25 * (o_mah_dist(bA, bDUa, Xa, Xb) + o_mah_dist(bB, bDUb, Xa, Xb) + o_mah_dist(aA, aDUa, Ya, Yb) + o_mah_dist(aB, aDUb, Ya, Yb));
}
/* the degeneracy test of a 3-line sample (motion.cpp:424-437): 1 if every pair is parallel within the threshold */
static int o_relmotion_degenerate(const double *la, double cos_thresh) {
  double u[9];
  int i, j, k;
  for (i = 0; i < 3; i++) {
    double l[3], inv;
    for (k = 0; k < 3; k++) l[k] = la[6 * i + 3 + k] - la[6 * i + k];
    inv = 1 / o_len3(l);
    for (k = 0; k < 3; k++) u[3 * i + k] = l[k] * inv;
  }
  for (i = 0; i < 3; i++)
    for (j = i + 1; j < 3; j++)
      if (fabs((u[3 * i] * u[3 * j] + u[3 * i + 1] * u[3 * j + 1]) + u[3 * i + 2] * u[3 * j + 2]) < cos_thresh) return 0;
  return 1;
}

/* ---------------------------------------------------------------- point features (config 3) */
typedef struct { double raster_cov_x, raster_cov_y, sigma_depth; } o_point_model;   /* misc.cpp:704-711 (host libm), misc2.h:23 */
static double o_depth_covariance(double depth, double sigma_depth) { const double sd = sigma_depth * depth * depth; return sd * sd; }   /* misc2.h:20-35 */
/* errorFunction2 (misc.cpp:699-786): squared Mahalanobis distance of a point match under tf (query -> train; the float matrix cast
 * to double), with the isotropic-bound shortcut; DBL_MAX = "certainly not an inlier" */
static double o_error_function2(const float *x1, const float *x2, const float *tf, const o_point_model *pm) {
  const double BIG = 1.7976931348623157e308;
  double T[16], mu1[3], mu2[3], m12[3], d[3], R[9], c1[3], c2[3], S[9], rhs[3], q, dsq, s1, s2;
  int r, c, k;
  if (x1[2] != x1[2] || x2[2] != x2[2]) return BIG;
  for (k = 0; k < 16; k++) T[k] = (double)tf[k];
  for (k = 0; k < 3; k++) { mu1[k] = (double)x1[k]; mu2[k] = (double)x2[k]; }
  for (r = 0; r < 3; r++) m12[r] = ((T[4 * r] * mu1[0] + T[4 * r + 1] * mu1[1]) + T[4 * r + 2] * mu1[2]) + T[4 * r + 3] * (double)x1[3];
  o_sub3(m12, mu2, d);
  dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  s1 = o_depth_covariance(mu1[2], pm->sigma_depth); s2 = o_depth_covariance(mu2[2], pm->sigma_depth);
  if (s1 < pm->raster_cov_x) s1 = pm->raster_cov_x;
  if (s2 < pm->raster_cov_x) s2 = pm->raster_cov_x;
  if (dsq > 2.0 * (s1 + s2)) return BIG;
  for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) R[3 * r + c] = T[4 * r + c];
  c1[0] = 1 * pm->raster_cov_x * mu1[2]; c1[1] = 1 * pm->raster_cov_y * mu1[2]; c1[2] = o_depth_covariance(mu1[2], pm->sigma_depth);
  c2[0] = 1 * pm->raster_cov_x * mu2[2]; c2[1] = 1 * pm->raster_cov_y * mu2[2]; c2[2] = o_depth_covariance(mu2[2], pm->sigma_depth);
  for (r = 0; r < 3; r++)                        /* R^T diag(c1) R (as written at misc.cpp:765) + diag(c2) */
    for (c = 0; c < 3; c++) {
      double s = 0;
      for (k = 0; k < 3; k++) s += (R[3 * k + r] * c1[k]) * R[3 * k + c];
      S[3 * r + c] = s + ((r == c) ? c2[r] : 0.0);
    }
  if (d[2] != d[2]) d[2] = 0.0;
  for (k = 0; k < 3; k++) rhs[k] = d[k];
  if (!o_lu_solve(3, S, 1, rhs)) return BIG;      /* Eigen LDLT solve in the reference */
  q = d[0] * rhs[0] + d[1] * rhs[1] + d[2] * rhs[2];
  if (!(q >= 0.0)) return BIG;
  return q;
}
/* projectPt3d2Ln3d_2 (utils.cpp:506-512) */
static void o_project_pt_line(const double *P, const double *A, const double *B, double *out) {
  double AB[3], AP[3], s;
  int k;
  o_sub3(B, A, AB);
  o_sub3(P, A, AP);
  s = o_dot3(AB, AP) / o_dot3(AB, AB);
  for (k = 0; k < 3; k++) out[k] = A[k] + s * AB[k];
}
/* pcl::TransformationFromCorrespondences (PCL 1.7 common/transformation_from_correspondences.hpp): float running means and
 * covariance; the final 3x3 SVD in double (o_svd3) */
typedef struct { int n; float wsum; float m1[3], m2[3], cov[9]; } o_tfc;
static void o_tfc_reset(o_tfc *t) { int i; t->n = 0; t->wsum = 0.0f; for (i = 0; i < 3; i++) t->m1[i] = t->m2[i] = 0.0f; for (i = 0; i < 9; i++) t->cov[i] = 0.0f; }
static void o_tfc_add(o_tfc *t, const float *from, const float *to, float w) {
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
static void o_tfc_get(const o_tfc *t, float *tf) {
  double C[9], U[9], sg[3], V[9], R[9], s22 = 1.0;
  int r, c, k;
  for (k = 0; k < 9; k++) C[k] = (double)t->cov[k];
  o_svd3(C, U, sg, V);
  if (o_det3(U) * o_det3(V) < 0.0) s22 = -1.0;
  for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) R[3 * r + c] = (U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1]) + s22 * U[3 * r + 2] * V[3 * c + 2];
  for (r = 0; r < 3; r++) {
    const float rf[3] = {(float)R[3 * r], (float)R[3 * r + 1], (float)R[3 * r + 2]};
    tf[4 * r] = rf[0]; tf[4 * r + 1] = rf[1]; tf[4 * r + 2] = rf[2];
    tf[4 * r + 3] = t->m2[r] - ((rf[0] * t->m1[0] + rf[1] * t->m1[1]) + rf[2] * t->m1[2]);
  }
  tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
}
/* compPt3dCov, Eigen overload (utils.cpp:724-745): Matrix3f covariance J diag(s^2, s^2, sz^2) J^T, cast to double, inverted:
 * the information of a point edge (transformation_estimation.cpp:267,283) */
static int o_point_information(const float *pt, double f, double sig_px, double c1, double c2, double c3, double *info) {
  const double x = (double)pt[0], y = (double)pt[1], z = (double)pt[2];
  const double sz = c1 * z * z + c2 * z + c3, s2 = sig_px * sig_px, sz2 = sz * sz;
  const double j00 = z / f, j02 = x / z, j11 = z / f, j12 = y / z;
  double C[9];
  int k;
  C[0] = (j00 * s2) * j00 + (j02 * sz2) * j02; C[1] = (j02 * sz2) * j12; C[2] = (j02 * sz2);
  C[3] = (j12 * sz2) * j02; C[4] = (j11 * s2) * j11 + (j12 * sz2) * j12; C[5] = (j12 * sz2);
  C[6] = sz2 * j02; C[7] = sz2 * j12; C[8] = sz2;
  for (k = 0; k < 9; k++) C[k] = (double)(float)C[k];
  return o_inv3(C, info);
}

typedef struct { const double *mn, *mo, *In, *Io; } o_point_meas;     /* newer / older measurement and their information matrices */
typedef struct { double V[9], W[18], bl[3], Hpp[36], bp[6]; } o_point_blocks;
/* EdgeSE3PointXYZ::computeError (edge_se3_ptxyz.cpp:84-90) for both cameras */
static void o_ptmatch_errors(const o_se3 *X, const double *p, const o_point_meas *m, double *en, double *eo) {
  double q[3];
  o_sub3(p, m->mn, en);
  o_se3_inv_apply(X, p, q);
  o_sub3(q, m->mo, eo);
}
static double o_quad3(const double *e, const double *I) { return e[0] * o_row3(I, 0, e) + e[1] * o_row3(I, 1, e) + e[2] * o_row3(I, 2, e); }
static double o_ptmatch_chi2(const o_se3 *X, const double *p, const o_point_meas *m, double hd, int hub) {
  double en[3], eo[3], rn, ro, w;
  o_ptmatch_errors(X, p, m, en, eo);
  o_huber(o_quad3(en, m->In), hd, hub, &rn, &w);
  o_huber(o_quad3(eo, m->Io), hd, hub, &ro, &w);
  return rn + ro;
}
static void o_ptmatch_blocks(const o_se3 *X, const double *p, const o_point_meas *m, double hd, int hub, o_point_blocks *B) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  double en[3], eo[3], Jn[9], Jo[9], Jp[18], wn, wo, rho, On[9], Oo[9], gn[3], go[3], OJn[9], OJo[9], OJp[18];
  int d, i, j, k;
  o_ptmatch_errors(X, p, m, en, eo);
  for (d = 0; d < 3; d++) {
    double pq[3], nP[3], nM[3], oP[3], oM[3];
    for (i = 0; i < 3; i++) pq[i] = p[i];
    pq[d] = p[d] + delta; o_ptmatch_errors(X, pq, m, nP, oP);
    pq[d] = p[d] - delta; o_ptmatch_errors(X, pq, m, nM, oM);
    for (i = 0; i < 3; i++) { Jn[3 * i + d] = scalar * (nP[i] - nM[i]); Jo[3 * i + d] = scalar * (oP[i] - oM[i]); }
  }
  for (d = 0; d < 6; d++) {
    double v[6], q[3], eP[3], eM[3];
    o_se3 Xq;
    int sgn;
    for (sgn = 0; sgn < 2; sgn++) {
      for (i = 0; i < 6; i++) v[i] = 0;
      v[d] = sgn ? -delta : delta;
      o_se3_oplus(X, v, &Xq);
      o_se3_inv_apply(&Xq, p, q);
      o_sub3(q, m->mo, sgn ? eM : eP);
    }
    for (i = 0; i < 3; i++) Jp[6 * i + d] = scalar * (eP[i] - eM[i]);
  }
  o_huber(o_quad3(en, m->In), hd, hub, &rho, &wn);
  o_huber(o_quad3(eo, m->Io), hd, hub, &rho, &wo);
  for (i = 0; i < 9; i++) { On[i] = wn * m->In[i]; Oo[i] = wo * m->Io[i]; }
  for (i = 0; i < 3; i++) { gn[i] = o_row3(On, i, en); go[i] = o_row3(Oo, i, eo); }
  for (i = 0; i < 3; i++) {
    double s1 = 0, s2 = 0;
    for (k = 0; k < 3; k++) { s1 += Jn[3 * k + i] * gn[k]; s2 += Jo[3 * k + i] * go[k]; }
    B->bl[i] = -(s1 + s2);
  }
  for (i = 0; i < 6; i++) { double s3 = 0; for (k = 0; k < 3; k++) s3 += Jp[6 * k + i] * go[k]; B->bp[i] = -s3; }
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
static int o_ptmatch_eliminate(const o_point_blocks *B, double lambda, double *Vi, double *T, double *u) {
  double A[9], WV[18];
  int i, j, k;
  for (i = 0; i < 9; i++) A[i] = B->V[i];
  for (i = 0; i < 3; i++) A[4 * i] += lambda;
  if (!o_inv3(A, Vi)) return 0;
  for (i = 0; i < 6; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += B->W[3 * i + k] * Vi[3 * k + j]; WV[3 * i + j] = s; }
  for (i = 0; i < 6; i++) {
    double s = 0;
    for (k = 0; k < 3; k++) s += WV[3 * i + k] * B->bl[k];
    u[i] = s;
    for (j = 0; j < 6; j++) { double s2 = 0; for (k = 0; k < 3; k++) s2 += WV[3 * i + k] * B->W[3 * j + k]; T[6 * i + j] = s2; }
  }
  return 1;
}
static void o_ptmatch_backsub(const o_point_blocks *B, const double *Vi, const double *dp, double *dl) {
  double r[3];
  int i, k;
  for (i = 0; i < 3; i++) { double s = 0; for (k = 0; k < 6; k++) s += B->W[3 * k + i] * dp[k]; r[i] = B->bl[i] - s; }
  for (i = 0; i < 3; i++) dl[i] = o_row3(Vi, i, r);
}
#endif /* O_POSE_H */
