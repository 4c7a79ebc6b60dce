/* oracle/pair_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the pair solver for line features:
 *   Node::lineMatching                 src/node.cpp:1619-1694  (+ utils.cpp:1250-1273, 1612-1638)
 *   getTransform_PtsLines_ransac       src/line/motion.cpp:605-849   with nPt = 0 (lines-only
 *                                      odometry, BASELINE.json config 2): every minimal sample is
 *                                      three line matches -> getTransform_Line_svd (:581-603)
 *   getTransformFromHybridMatchesG2O   src/transformation_estimation.cpp:218-461 (line edges)
 *   Node::matchNodePair bookkeeping    src/node.cpp:1494-1615 (valid edge, information scale)
 *
 * Parity status: "parity unpinned" -- the reference sources for this stage need OpenCV/Eigen/g2o/PCL
 * (absent) and its tests hold no vectors.  The geometric primitives and the g2o-style LM pieces are
 * the ORACLE'S OWN statements in o_pose.h / o_linalg.h (round 6; until round 5 this file compiled the
 * product's lf_pose.h): the kernels are held to them bit for bit, and tests/test_oracle_pose_primitives.py
 * holds them to the product's scalar functions; THIS file is the sequential composition, pinned by
 * analytic known-answer tests (tests/test_oracle_pair.py: exact rigid motions are recovered) and by the
 * source-independent fixtures of oracle/pose_indep.py (tests/test_pose_golden_cpu.py).
 * rand() -> o_rand31 (see front_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/linefront.h"
#include "o_pose.h"    /* the oracle's own primitives (its acos: host libm, or lf_math.h's under ORACLE_LFMATH) */

#define O_EPS 1e-10
#define O_PI_SHORT 3.14159265   /* lineslam.h:38  #define PI (3.14159265) */

/* pt_to_line_dist2d, utils.cpp:1250-1264 */
static double p_pt_line2d(const double p[2], const double l[3]) {
  double a = l[0], b = l[1], c = l[2];
  return fabs((a * p[0] + b * p[1] + c)) / sqrt(a * a + b * b);
}
/* line_to_line_dist2d, utils.cpp:1265-1273 */
static double p_line_line2d(const lf_line_record *a, const lf_line_record *b) {
  return 0.25 * p_pt_line2d(a->p, b->lineEq2d) + 0.25 * p_pt_line2d(a->q, b->lineEq2d) +
         0.25 * p_pt_line2d(b->p, a->lineEq2d) + 0.25 * p_pt_line2d(b->q, a->lineEq2d);
}
static double p_norm2(const double a[2], const double b[2]) {
  return sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]));
}
/* projectPt2d_to_line2d, utils.cpp:1612-1618 */
static double p_project2d(const double X[2], const double A[2], const double B[2]) {
  double BX[2] = {X[0] - B[0], X[1] - B[1]}, BA[2] = {A[0] - B[0], A[1] - B[1]};
  double n = sqrt(BA[0] * BA[0] + BA[1] * BA[1]);
  return (BX[0] * BA[0] + BX[1] * BA[1]) / n / n;
}
/* lineSegmentOverlap, utils.cpp:1620-1638 */
static double p_overlap(const lf_line_record *a, const lf_line_record *b) {
  if (p_norm2(a->p, a->q) < p_norm2(b->p, b->q)) {
    double lp = p_project2d(a->p, b->p, b->q), lq = p_project2d(a->q, b->p, b->q);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return fabs(lp - lq) * p_norm2(b->p, b->q);
  } else {
    double lp = p_project2d(b->p, a->p, a->q), lq = p_project2d(b->q, a->p, a->q);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return fabs(lp - lq) * p_norm2(a->p, a->q);
  }
}

/* Node::lineMatching, node.cpp:1619-1694.  f1 = this (query), f2 = other (train).  D (n1*n2, may be
 * NULL) receives descDiff.  Returns the number of matches written (queryIdx, trainIdx, distance). */
int oracle_line_matching(const lf_line_record *f1, int n1, const lf_line_record *f2, int n2, int adjacent,
                         int *mq, int *mt, double *md, int cap, double *D_out) {
  double lineDistThresh, lineAngleThresh = 30 * O_PI_SHORT / 180, descDiffThresh, lineOverlapThresh, ratio = 0.7;
  double cosang = cos(lineAngleThresh), *D;
  int i, j, n = 0;
  if (adjacent) { lineDistThresh = 45; descDiffThresh = 0.85; lineOverlapThresh = 0; }
  else { lineDistThresh = 80; descDiffThresh = 0.7; lineOverlapThresh = -1; }
  if (n1 == 0 || n2 == 0) return 0;
  D = (double *)malloc(sizeof(double) * (size_t)n1 * n2);
#pragma omp parallel for private(j)      /* node.cpp:1644 (active in the OpenMP flavour only) */
  for (i = 0; i < n1; ++i)
    for (j = 0; j < n2; ++j) {
      double v = 100;
      if ((f1[i].r[0] * f2[j].r[0] + f1[i].r[1] * f2[j].r[1] > cosang) &&
          (p_line_line2d(&f1[i], &f2[j]) < lineDistThresh) && (p_overlap(&f1[i], &f2[j]) > lineOverlapThresh)) {
        double s = 0;
        int k;
        /* cv::norm(Mat) of the 72 x 1 CV_64F difference: OpenCV 2.4's normL2Sqr_ (modules/core/src/stat.cpp) takes four
         * elements per trip and adds their squares to the running sum as ONE expression, s += v0*v0 + v1*v1 + v2*v2 + v3*v3 */
        for (k = 0; k < 72; k += 4) {
          double v0 = f1[i].des[k] - f2[j].des[k], v1 = f1[i].des[k + 1] - f2[j].des[k + 1];
          double v2 = f1[i].des[k + 2] - f2[j].des[k + 2], v3 = f1[i].des[k + 3] - f2[j].des[k + 3];
          s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
        }
        v = sqrt(s);
      }
      D[(size_t)i * n2 + j] = v;
    }
  for (i = 0; i < n1; ++i) {
    double minVal = D[(size_t)i * n2], minV, rowmin2 = 100, colmin2 = 100;
    int minPos = 0, minP = 0;
    for (j = 1; j < n2; j++) if (D[(size_t)i * n2 + j] < minVal) { minVal = D[(size_t)i * n2 + j]; minPos = j; }   /* minMaxLoc: first minimum */
    if (!(minVal < descDiffThresh)) continue;
    minV = D[minPos];
    for (j = 1; j < n1; j++) if (D[(size_t)j * n2 + minPos] < minV) { minV = D[(size_t)j * n2 + minPos]; minP = j; }
    if (i != minP) continue;
    for (j = 0; j < n2; ++j) { if (j == minPos) continue; if (rowmin2 > D[(size_t)i * n2 + j]) rowmin2 = D[(size_t)i * n2 + j]; }
    for (j = 0; j < n1; ++j) { if (j == minP) continue; if (colmin2 > D[(size_t)j * n2 + minPos]) colmin2 = D[(size_t)j * n2 + minPos]; }
    if (rowmin2 * ratio > minVal && colmin2 * ratio > minVal) {
      if (n < cap) { mq[n] = i; mt[n] = minPos; md[n] = minVal; }   /* haveDepth holds for every record */
      n++;
    }
  }
  if (D_out) memcpy(D_out, D, sizeof(double) * (size_t)n1 * n2);
  free(D);
  return n;
}

/* ---------------------------------------------------------------------------------------------
 * getTransformFromHybridMatchesG2O (transformation_estimation.cpp:218-461): g2o Levenberg on
 * {older camera pose, one 3-d landmark per point match, one 6-d landmark per line match}; newer camera
 * fixed at I.  Vertices/edges are created points first, then lines (:247-292, :322-440), and that is
 * the order of every sum below.                                                                       */
typedef struct { const float *train, *query; int n; const int *mq, *mt; } o_points;   /* xyz1 float4 arrays + matches */

static void p_meas(const lf_line_record *train, const lf_line_record *query, int tq, int tt, o_line_meas *m) {
  m->nA = query[tq].A; m->nB = query[tq].B; m->nMa = query[tq].DUa; m->nMb = query[tq].DUb;
  m->oA = train[tt].A; m->oB = train[tt].B; m->oMa = train[tt].DUa; m->oMb = train[tt].DUb;
}
static void o_point_model_init(o_point_model *pm) {   /* misc.cpp:704-711 + misc2.h:23 */
  const double cam_angle_x = 58.0 / 180.0 * M_PI, cam_angle_y = 45.0 / 180.0 * M_PI;
  double sx = 3 * tan(cam_angle_x / 640), sy = 3 * tan(cam_angle_y / 480);
  pm->raster_cov_x = sx * sx; pm->raster_cov_y = sy * sy; pm->sigma_depth = 0.01;
}

int oracle_refine_hybrid(const lf_line_record *train, const lf_line_record *query, const int *mq, const int *mt,
                         int n, const o_points *pp, int npt, const int *pq, const int *pt, double focal, float *tf,
                         int iterations, const lf_params *P, double *chi_out) {
  o_se3 X, Xn;
  double *L = (double *)malloc(sizeof(double) * 6 * (size_t)(n + 1)), *Ln = (double *)malloc(sizeof(double) * 6 * (size_t)(n + 1));
  o_line_meas *M = (o_line_meas *)malloc(sizeof(o_line_meas) * (size_t)(n + 1));
  o_line_blocks *B = (o_line_blocks *)malloc(sizeof(o_line_blocks) * (size_t)(n + 1));
  double *Vi = (double *)malloc(sizeof(double) * 36 * (size_t)(n + 1));
  /* points */
  double *Pp = (double *)malloc(sizeof(double) * 3 * (size_t)(npt + 1)), *Pn = (double *)malloc(sizeof(double) * 3 * (size_t)(npt + 1));
  double *PM = (double *)malloc(sizeof(double) * 24 * (size_t)(npt + 1));   /* mn(3) mo(3) In(9) Io(9) */
  o_point_meas *Pm = (o_point_meas *)malloc(sizeof(o_point_meas) * (size_t)(npt + 1));
  o_point_blocks *PB = (o_point_blocks *)malloc(sizeof(o_point_blocks) * (size_t)(npt + 1));
  double *PVi = (double *)malloc(sizeof(double) * 9 * (size_t)(npt + 1));
  double lambda = 0, ni = 2, currentChi = 0, wgt = P->g2o_line_error_weight, hd = P->g2o_BA_kernel_delta;
  int hub = P->g2o_BA_use_kernel, it, k, i, done_iters = 0;
  o_tf_to_older_pose(tf, &X);
  for (k = 0; k < npt; k++) {   /* :247-292 */
    const float *qn = pp->query + 4 * (size_t)pq[k], *qo = pp->train + 4 * (size_t)pt[k];
    double *m = PM + 24 * (size_t)k;
    for (i = 0; i < 3; i++) { m[i] = (double)qn[i]; m[3 + i] = (double)qo[i]; Pp[3 * k + i] = (double)qn[i]; }
    o_point_information(qn, focal, P->stdev_sample_pt_imgline, P->depth_stdev_coeff_c1, P->depth_stdev_coeff_c2 + 0.0 * 0.5, P->depth_stdev_coeff_c3, m + 6);
    o_point_information(qo, focal, P->stdev_sample_pt_imgline, P->depth_stdev_coeff_c1, P->depth_stdev_coeff_c2 + 0.0 * 0.5, P->depth_stdev_coeff_c3, m + 15);
    Pm[k].mn = m; Pm[k].mo = m + 3; Pm[k].In = m + 6; Pm[k].Io = m + 15;
  }
  for (k = 0; k < n; k++) {
    p_meas(train, query, mq[k], mt[k], &M[k]);
    for (i = 0; i < 3; i++) { L[6 * k + i] = query[mq[k]].A[i]; L[6 * k + 3 + i] = query[mq[k]].B[i]; }   /* :326-328 */
  }
  for (it = 0; it < iterations && (n + npt) > 0; it++) {
    double Hpp[36], bp[6], rho = 0, tempChi;
    int qmax = 0;
    currentChi = 0;
    for (k = 0; k < npt; k++) currentChi += o_ptmatch_chi2(&X, &Pp[3 * k], &Pm[k], hd, hub);
    for (k = 0; k < n; k++) currentChi += o_match_chi2(&X, &L[6 * k], &M[k], wgt, hd, hub);
    for (i = 0; i < 36; i++) Hpp[i] = 0;
    for (i = 0; i < 6; i++) bp[i] = 0;
    for (k = 0; k < npt; k++) {
      o_ptmatch_blocks(&X, &Pp[3 * k], &Pm[k], hd, hub, &PB[k]);
      for (i = 0; i < 36; i++) Hpp[i] += PB[k].Hpp[i];
      for (i = 0; i < 6; i++) bp[i] += PB[k].bp[i];
    }
    for (k = 0; k < n; k++) {
      o_match_blocks(&X, &L[6 * k], &M[k], wgt, hd, hub, &B[k]);
      for (i = 0; i < 36; i++) Hpp[i] += B[k].Hpp[i];
      for (i = 0; i < 6; i++) bp[i] += B[k].bp[i];
    }
    if (it == 0) {   /* computeLambdaInit: tau * max |diagonal| */
      double mx = 0;
      for (i = 0; i < 6; i++) if (fabs(Hpp[7 * i]) > mx) mx = fabs(Hpp[7 * i]);
      for (k = 0; k < npt; k++) for (i = 0; i < 3; i++) if (fabs(PB[k].V[4 * i]) > mx) mx = fabs(PB[k].V[4 * i]);
      for (k = 0; k < n; k++) for (i = 0; i < 6; i++) if (fabs(B[k].V[7 * i]) > mx) mx = fabs(B[k].V[7 * i]);
      lambda = 1e-5 * mx;
      ni = 2;
    }
    do {
      double S[36], g[6], dp[6], scale = 0;
      int ok2 = 1;
      for (i = 0; i < 36; i++) S[i] = Hpp[i];
      for (i = 0; i < 6; i++) { S[7 * i] += lambda; g[i] = bp[i]; }
      for (k = 0; k < npt && ok2; k++) {
        double T[36], u[6];
        if (!o_ptmatch_eliminate(&PB[k], lambda, &PVi[9 * k], T, u)) { ok2 = 0; break; }
        for (i = 0; i < 36; i++) S[i] -= T[i];
        for (i = 0; i < 6; i++) g[i] -= u[i];
      }
      for (k = 0; k < n && ok2; k++) {
        double T[36], u[6];
        if (!o_match_eliminate(&B[k], lambda, &Vi[36 * k], T, u)) { ok2 = 0; break; }
        for (i = 0; i < 36; i++) S[i] -= T[i];
        for (i = 0; i < 6; i++) g[i] -= u[i];
      }
      if (ok2) { double A[36]; for (i = 0; i < 36; i++) A[i] = S[i]; for (i = 0; i < 6; i++) dp[i] = g[i]; ok2 = o_lu_solve(6, A, 1, dp); }
      tempChi = DBL_MAX;
      if (ok2) {
        o_se3_oplus(&X, dp, &Xn);
        for (i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + bp[i]);
        tempChi = 0;
        for (k = 0; k < npt; k++) {
          double dl[3], sk = 0;
          o_ptmatch_backsub(&PB[k], &PVi[9 * k], dp, dl);
          for (i = 0; i < 3; i++) { Pn[3 * k + i] = Pp[3 * k + i] + dl[i]; sk += dl[i] * (lambda * dl[i] + PB[k].bl[i]); }
          scale += sk;
        }
        for (k = 0; k < n; k++) {
          double dl[6], sk = 0;
          o_match_backsub(&B[k], &Vi[36 * k], dp, dl);
          for (i = 0; i < 6; i++) { Ln[6 * k + i] = L[6 * k + i] + dl[i]; sk += dl[i] * (lambda * dl[i] + B[k].bl[i]); }
          scale += sk;
        }
        for (k = 0; k < npt; k++) tempChi += o_ptmatch_chi2(&Xn, &Pn[3 * k], &Pm[k], hd, hub);
        for (k = 0; k < n; k++) tempChi += o_match_chi2(&Xn, &Ln[6 * k], &M[k], wgt, hd, hub);
      }
      rho = (currentChi - tempChi);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && tempChi <= DBL_MAX && tempChi == tempChi) {
        double t = 2 * rho - 1, alpha = 1. - t * t * t, sf;
        if (alpha > 2. / 3.) alpha = 2. / 3.;
        sf = alpha > 1. / 3. ? alpha : 1. / 3.;
        lambda *= sf;
        ni = 2;
        currentChi = tempChi;
        X = Xn;
        memcpy(L, Ln, sizeof(double) * 6 * (size_t)n);
        memcpy(Pp, Pn, sizeof(double) * 3 * (size_t)npt);
      } else {
        lambda *= ni;
        ni *= 2;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    done_iters = it + 1;
    if (qmax == 10 || rho == 0) break;
  }
  o_older_pose_to_tf(&X, tf);
  if (chi_out) *chi_out = currentChi;
  free(L); free(Ln); free(M); free(B); free(Vi); free(Pp); free(Pn); free(PM); free(Pm); free(PB); free(PVi);
  return done_iters;
}

/* getTransform_Lns_Pts_pcl (motion.cpp:530-579): mixed minimal sample -> weighted Kabsch.  The random
 * point each sampled line is paired with: draw index (1<<20) + 3*iteration + position in the line list. */
static int o_lns_pts_pcl(const lf_line_record *train, const lf_line_record *query, const o_points *pp,
                         const int *spq, const int *spt, int nsp, const int *slq, const int *slt, int nsl,
                         const lf_params *P, uint64_t stream, int iter, float *tf) {
  o_tfc t;
  int i, k;
  if (nsp < 1 || nsp + nsl < 3) return 0;
  o_tfc_reset(&t);
  for (i = 0; i < nsl; ++i) {
    int ptidx = (int)(o_rand31(P->rng_seed, stream, (1ull << 20) + 3ull * (uint64_t)iter + (uint64_t)i) % (uint32_t)nsp);
    const float *tp = pp->train + 4 * (size_t)spt[ptidx], *qp = pp->query + 4 * (size_t)spq[ptidx];
    double tpd[3] = {tp[0], tp[1], tp[2]}, qpd[3] = {qp[0], qp[1], qp[2]}, tprj[3], qprj[3];
    float from[3], to[3], w;
    o_project_pt_line(tpd, train[slt[i]].A, train[slt[i]].B, tprj);
    o_project_pt_line(qpd, query[slq[i]].A, query[slq[i]].B, qprj);
    for (k = 0; k < 3; k++) { from[k] = (float)qprj[k]; to[k] = (float)tprj[k]; }
    if (from[2] != from[2] || to[2] != to[2]) continue;
    w = 1 / (fabsf(to[2]) + fabsf(from[2]));
    o_tfc_add(&t, from, to, w);
  }
  for (i = 0; i < nsp; ++i) {
    const float *from = pp->query + 4 * (size_t)spq[i], *to = pp->train + 4 * (size_t)spt[i];
    float w;
    if (from[2] != from[2] || to[2] != to[2]) continue;
    w = 1 / (fabsf(to[2]) + fabsf(from[2]));
    o_tfc_add(&t, from, to, w);
  }
  if (t.n < 3) return 0;
  o_tfc_get(&t, tf);
  return 1;
}

static int o_score(const lf_line_record *train, const lf_line_record *query, const o_points *pp, int nPt,
                   const int *pq, const int *ptm, const int *mq, const int *mt, int nLn, const float *tf,
                   const o_point_model *pm, double thr, int *pset, int *npin, int *lset, int *nlin,
                   float *sse_f, double *sse_d) {
  int i, np = 0, nl = 0;
  float sf = 0; double sd = 0;
  for (i = 0; i < nPt; ++i) {
    double m = o_error_function2(pp->query + 4 * (size_t)pq[i], pp->train + 4 * (size_t)ptm[i], tf, pm);
    if (m < thr * thr) { pset[np++] = i; sf += m; sd += m; }
  }
  for (i = 0; i < nLn; ++i) {
    double add;
    if (o_line_inlier(tf, query[mq[i]].A, query[mq[i]].B, train[mt[i]].A, train[mt[i]].B, train[mt[i]].DUa,
                       train[mt[i]].DUb, thr, &add)) { lset[nl++] = i; sf += add; sd += add; }
  }
  *npin = np; *nlin = nl; *sse_f = sf; *sse_d = sd;
  return np + nl;
}

/* getTransform_PtsLines_ransac (motion.cpp:605-849).  train = older node, query = newer node.
 *   point matches (pq, ptm)[nPt] into the float4 arrays of `pp`; line matches (mq, mt)[nLn]
 *   tf_out: 4x4 row-major float, query -> train;  pinl / linl: indices into the two match lists        */
int oracle_pose_hybrid_ransac(const lf_line_record *train, const lf_line_record *query, const float *train_pts,
                              const float *query_pts, const int *pq, const int *ptm, int nPt, const int *mq,
                              const int *mt, int nLn, int id_train, int id_query, double focal, const lf_params *P,
                              uint64_t stream, float *tf_out, float *rmse_out, int *pinl, int *n_pinl, int *linl,
                              int *n_linl, int *dbg) {
  int min_inlier = P->min_feature_matches, lw = P->line_match_number_weight, maxIter = P->ransac_iters_line_motion;
  double thr = P->max_mah_dist_for_inliers;
  int nTot = nPt + nLn, *indexes, *bp, *bl, *cp, *cl, *rp, *rl, nbp = 0, nbl = 0, nrp = 0, nrl = 0, iter, i, best_iter = -1, rounds = 0;
  float tf_best[16], sse_best = 1e9f, refined_tf[16];
  double refined_rmse;
  uint64_t ctr = 0;
  int idd = id_train - id_query;
  o_points pp;
  o_point_model pm;
  pp.train = train_pts; pp.query = query_pts;
  o_point_model_init(&pm);
  *n_pinl = 0; *n_linl = 0;
  for (i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (dbg) dbg[0] = dbg[1] = dbg[2] = dbg[3] = 0;
  if (nPt + nLn * lw < min_inlier) { *rmse_out = 1e9f; return 0; }                          /* :621-624 */
  if (min_inlier > 0.7 * (nPt + nLn * lw)) min_inlier = (int)(0.7 * (nPt + nLn * lw));       /* :626-628 */
  if ((idd < 0 ? -idd : idd) > 50) min_inlier = P->min_matches_loopclose;                    /* :631-633 */
  if (nTot < 3) { *rmse_out = 1e9f; return 0; }   /* random_unique(3) needs 3 elements; the reference would read out of range */
  indexes = (int *)malloc(sizeof(int) * (size_t)nTot * 8);
  bp = indexes + nTot; bl = bp + nTot; cp = bl + nTot; cl = cp + nTot; rp = cl + nTot; rl = rp + nTot;
  for (i = 0; i < nTot; i++) indexes[i] = i;
  for (iter = 0; iter < maxIter; iter++) {
    float tf[16], sse_f; double sse_d;
    int ncp, ncl, b = 0, left = nTot, s, spq[3], spt[3], slq[3], slt[3], nsp = 0, nsl = 0, valid;
    for (s = 0; s < 3; s++) {   /* random_unique(indexes, 3) */
      int r = b + (int)(o_rand31(P->rng_seed, stream, ctr++) % (uint32_t)left);
      int tmp = indexes[b]; indexes[b] = indexes[r]; indexes[r] = tmp;
      ++b; --left;
    }
    for (s = 0; s < 3; s++) {
      if (indexes[s] < nPt) { spq[nsp] = pq[indexes[s]]; spt[nsp] = ptm[indexes[s]]; nsp++; }
      else { slq[nsl] = mq[indexes[s] - nPt]; slt[nsl] = mt[indexes[s] - nPt]; nsl++; }
    }
    if (nsl == 3) {   /* getTransform_Line_svd: computeRelativeMotion_svd(query, train) */
      double la[18], lb[18], R[9], t[3];
      int c;
      for (s = 0; s < 3; s++)
        for (c = 0; c < 3; c++) {
          la[6 * s + c] = query[slq[s]].A[c]; la[6 * s + 3 + c] = query[slq[s]].B[c];
          lb[6 * s + c] = train[slt[s]].A[c]; lb[6 * s + 3 + c] = train[slt[s]].B[c];
        }
      valid = o_rel_motion_lines(la, lb, 3, R, t);
      for (i = 0; i < 3; i++) { for (c = 0; c < 3; c++) tf[4 * i + c] = (float)R[3 * i + c]; tf[4 * i + 3] = (float)t[i]; }
      tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
    } else
      valid = o_lns_pts_pcl(train, query, &pp, spq, spt, nsp, slq, slt, nsl, P, stream, iter, tf);
    if (!valid) continue;
    o_score(train, query, &pp, nPt, pq, ptm, mq, mt, nLn, tf, &pm, thr, cp, &ncp, cl, &ncl, &sse_f, &sse_d);
    if (ncp + lw * ncl > nbp + lw * nbl) {
      memcpy(bp, cp, sizeof(int) * (size_t)ncp); memcpy(bl, cl, sizeof(int) * (size_t)ncl);
      nbp = ncp; nbl = ncl; best_iter = iter;
      memcpy(tf_best, tf, sizeof tf);
      sse_best = sse_f;
    }
  }
  if (dbg) { dbg[0] = best_iter; dbg[1] = nbp + nbl; }
  if (nbp + nbl < 3) { free(indexes); *rmse_out = 1e9f; return 0; }                          /* :725-728 */
  {
    int *q1 = (int *)malloc(sizeof(int) * (size_t)(nTot + 1) * 4), *t1 = q1 + nTot + 1, *q2 = t1 + nTot + 1, *t2 = q2 + nTot + 1;
    float sse_f; double tmp_sse;
    memcpy(refined_tf, tf_best, sizeof tf_best);
    for (i = 0; i < nbp; i++) { q1[i] = pq[bp[i]]; t1[i] = ptm[bp[i]]; }
    for (i = 0; i < nbl; i++) { q2[i] = mq[bl[i]]; t2[i] = mt[bl[i]]; }
    oracle_refine_hybrid(train, query, q2, t2, nbl, &pp, nbp, q1, t1, focal, refined_tf, 25, P, NULL);   /* :730 */
    refined_rmse = sqrt(sse_best / (nbp + nbl));                                                         /* :731 */
    for (iter = 0; iter < 20; ++iter) {                                                                   /* :775-839 */
      int ncp, ncl;
      o_score(train, query, &pp, nPt, pq, ptm, mq, mt, nLn, refined_tf, &pm, thr, cp, &ncp, cl, &ncl, &sse_f, &tmp_sse);
      if (ncp + ncl * lw > nrp + nrl * lw) {
        memcpy(rp, cp, sizeof(int) * (size_t)ncp); memcpy(rl, cl, sizeof(int) * (size_t)ncl);
        nrp = ncp; nrl = ncl;
        refined_rmse = sqrt(tmp_sse / (ncp + ncl));
        for (i = 0; i < nrp; i++) { q1[i] = pq[rp[i]]; t1[i] = ptm[rp[i]]; }
        for (i = 0; i < nrl; i++) { q2[i] = mq[rl[i]]; t2[i] = mt[rl[i]]; }
        oracle_refine_hybrid(train, query, q2, t2, nrl, &pp, nrp, q1, t1, focal, refined_tf, 20, P, NULL);
        rounds++;
      } else break;
    }
    free(q1);
  }
  if (dbg) dbg[2] = rounds;
  for (i = 0; i < nrp; i++) pinl[i] = rp[i];
  for (i = 0; i < nrl; i++) linl[i] = rl[i];
  *n_pinl = nrp; *n_linl = nrl;
  *rmse_out = (float)refined_rmse;
  memcpy(tf_out, refined_tf, sizeof refined_tf);
  free(indexes);
  return (nrp + lw * nrl) >= min_inlier;
}

/* lines-only odometry (BASELINE config 2) = the same function with an empty point-match list */
int oracle_pose_lines_ransac(const lf_line_record *train, const lf_line_record *query, const int *mq,
                             const int *mt, int nLn, int id_train, int id_query, const lf_params *P,
                             uint64_t stream, float *tf_out, float *rmse_out, int *inl, int *n_inl, int *dbg) {
  int npi = 0, dummy[1];
  return oracle_pose_hybrid_ransac(train, query, NULL, NULL, NULL, NULL, 0, mq, mt, nLn, id_train, id_query, 525.0, P,
                                   stream, tf_out, rmse_out, dummy, &npi, inl, n_inl, dbg);
}
int oracle_refine_g2o(const lf_line_record *train, const lf_line_record *query, const int *mq, const int *mt,
                      int n, float *tf, int iterations, const lf_params *P, double *chi_out) {
  o_points pp = {NULL, NULL, 0, NULL, NULL};
  return oracle_refine_hybrid(train, query, mq, mt, n, &pp, 0, NULL, NULL, 525.0, tf, iterations, P, chi_out);
}

/* exported primitive wrappers for known-answer tests */
int oracle_rel_motion_lines(const double *la, const double *lb, int n, double *R, double *t) { return o_rel_motion_lines(la, lb, n, R, t); }
double oracle_error_function2(const float *x1, const float *x2, const float *tf) { o_point_model pm; o_point_model_init(&pm); return o_error_function2(x1, x2, tf, &pm); }
void oracle_kabsch(const float *from, const float *to, const float *w, int n, float *tf) { o_tfc t; int i; o_tfc_reset(&t); for (i = 0; i < n; i++) o_tfc_add(&t, from + 3 * i, to + 3 * i, w[i]); o_tfc_get(&t, tf); }


/* ---------------------------------------------------------------------------------------------
 * computeRelativeMotion_Ransac (src/line/motion.cpp:367-526) with optimizeRelmotion (:98-139) and
 * costFun_optimizeRelmotion (:60-96): the lines-only solver (declared in utils.h:132,
 * not called by the reference's ROS pipeline).  a = query (newer) lines, b = train (older) lines of the
 * matches (mq, mt); x_b = R x_a + t.  Returns the size of the consensus set written to inl[]; R/t are
 * only written when that set is not empty (as the reference leaves Ro/to untouched).
 * dbg: [0] winning RANSAC iteration, [1] |maxConSet|, [2] optimise rounds, [3] levmar iterations of the first
 * optimizeRelmotion.                                                                                  */
typedef void (*o_lm_func)(const double *p, double *hx, int m, int n, void *adata);
int oracle_levmar_dif(o_lm_func func, double *p, int m, int n, int itmax, const double opts[5], double info[10], void *adata);
typedef struct { const lf_line_record *a, *b; const int *ia, *ib; } o_relmot_data;
static void o_relmot_cost(const double *p, double *error, int m, int n, void *adata) {
  const o_relmot_data *d = (const o_relmot_data *)adata;
  double R[9];
  int i;
  (void)m;
  o_q2r(p, R);
  for (i = 0; i < n; i++) {
    const lf_line_record *a = &d->a[d->ia[i]], *b = &d->b[d->ib[i]];
    error[i] = o_relmotion_residual(R, p + 4, a->A, a->B, a->DUa, a->DUb, b->A, b->B, b->DUa, b->DUb);
  }
}
static int o_optimize_relmotion(const lf_line_record *a, const lf_line_record *b, const int *ia, const int *ib, int n,
                                double *R, double *t) {
  double opts[5] = {1E-03, 1E-10, 1E-20, 1E-20, 1E-06}, info[10], para[7];
  o_relmot_data d;
  d.a = a; d.b = b; d.ia = ia; d.ib = ib;
  o_r2q(R, para);
  para[4] = t[0]; para[5] = t[1]; para[6] = t[2];
  oracle_levmar_dif(o_relmot_cost, para, 7, n, 50, opts, info, &d);
  o_q2r(para, R);
  t[0] = para[4]; t[1] = para[5]; t[2] = para[6];
  return (int)info[5];
}
static int o_relmot_consensus(const lf_line_record *a, const lf_line_record *b, const int *mq, const int *mt, int n,
                              const double *R, const double *t, const lf_params *P, int *set) {
  int i, c = 0;
  for (i = 0; i < n; ++i)
    if (o_relmotion_inlier(R, t, a[mq[i]].A, a[mq[i]].B, b[mt[i]].A, b[mt[i]].B, P->pt2line3d_dist_relmotion,
                            P->line3d_angle_relmotion)) set[c++] = i;
  return c;
}
int oracle_relmotion_ransac(const lf_line_record *train, const lf_line_record *query, const int *mq, const int *mt,
                            int n, const lf_params *P, uint64_t stream, double *R_out, double *t_out, int *inl,
                            int *dbg) {
  const lf_line_record *a = query, *b = train;
  int maxIters = P->ransac_iters_line_motion, iter = 0, nmax = 0, nprev = 0, i, best_iter = -1, rounds = 0;
  int *indexes, *maxset, *cur, *prev, *ia, *ib;
  double bR[9], bt[3], R[9], t[3], Ro[9], to[3];
  const double cos_deg = cos(5 * O_PI_SHORT / 180);
  uint64_t ctr = 0;
  if (dbg) dbg[0] = -1, dbg[1] = dbg[2] = dbg[3] = 0;
  if (n < 3) return 0;
  indexes = (int *)malloc(sizeof(int) * (size_t)n * 6);
  maxset = indexes + n; cur = maxset + n; prev = cur + n; ia = prev + n; ib = ia + n;
  for (i = 0; i < n; i++) indexes[i] = i;
  while (iter < maxIters) {
    double la[18], lb[18], Rs[9], ts[3];
    int bpos = 0, left = n, s, c, nc;
    iter++;
    for (s = 0; s < 3; s++) {   /* random_unique(indexes, 3) */
      int r = bpos + (int)(o_rand31(P->rng_seed, stream, ctr++) % (uint32_t)left);
      int tmp = indexes[bpos]; indexes[bpos] = indexes[r]; indexes[r] = tmp;
      ++bpos; --left;
    }
    for (s = 0; s < 3; s++)
      for (c = 0; c < 3; c++) {
        la[6 * s + c] = a[mq[indexes[s]]].A[c]; la[6 * s + 3 + c] = a[mq[indexes[s]]].B[c];
        lb[6 * s + c] = b[mt[indexes[s]]].A[c]; lb[6 * s + 3 + c] = b[mt[indexes[s]]].B[c];
      }
    if (o_relmotion_degenerate(la, cos_deg)) continue;
    /* the reference ignores a failed computeRelativeMotion_svd (:440) and would go on with empty matrices;
     * here such a sample is skipped */
    if (!o_rel_motion_lines(la, lb, 3, Rs, ts)) continue;
    nc = o_relmot_consensus(a, b, mq, mt, n, Rs, ts, P, cur);
    if (nc > nmax) {
      memcpy(maxset, cur, sizeof(int) * (size_t)nc); nmax = nc; best_iter = iter - 1;
      memcpy(bR, Rs, sizeof bR); memcpy(bt, ts, sizeof bt);
    }
    if (nmax >= n * 1) break;   /* inlierRatio = 1 */
  }
  if (dbg) { dbg[0] = best_iter; dbg[1] = nmax; }
  if (nmax < 1) { free(indexes); return 0; }
  memcpy(Ro, bR, sizeof Ro); memcpy(to, bt, sizeof to);
  if (nmax < 4) {
    memcpy(R_out, Ro, sizeof Ro); memcpy(t_out, to, sizeof to);
    memcpy(inl, maxset, sizeof(int) * (size_t)nmax);
    free(indexes);
    return nmax;
  }
  for (i = 0; i < nmax; i++) { ia[i] = mq[maxset[i]]; ib[i] = mt[maxset[i]]; }
  i = o_optimize_relmotion(a, b, ia, ib, nmax, Ro, to);
  if (dbg) dbg[3] = i;
  memcpy(R, Ro, sizeof R); memcpy(t, to, sizeof t);
  for (;;) {
    int nc = o_relmot_consensus(a, b, mq, mt, n, R, t, P, cur);
    if (nc <= nprev) break;
    memcpy(prev, cur, sizeof(int) * (size_t)nc); nprev = nc;
    memcpy(Ro, R, sizeof R); memcpy(to, t, sizeof t);
    for (i = 0; i < nprev; i++) { ia[i] = mq[prev[i]]; ib[i] = mt[prev[i]]; }
    o_optimize_relmotion(a, b, ia, ib, nprev, R, t);
    rounds++;
  }
  if (dbg) dbg[2] = rounds;
  memcpy(R_out, Ro, sizeof Ro); memcpy(t_out, to, sizeof to);
  memcpy(inl, prev, sizeof(int) * (size_t)nprev);
  free(indexes);
  return nprev;
}
/* ---- the oracle's own primitives (o_pose.h, o_linalg.h), flat signatures: the twin of product_prim (product_hooks.c); the
 * two are held bit for bit on random inputs by tests/test_oracle_pose_primitives.py */
int oracle_prim(int which, const double *in, const float *fin, double *out, float *fout) {
  switch (which) {
    case 0: { double R[9], t[3]; int ok = o_rel_motion_lines(in, in + 18, (int)in[36], R, t); memcpy(out, R, sizeof R); memcpy(out + 9, t, sizeof t); return ok; }
    case 1: out[0] = o_mah_dist(in, in + 3, in + 12, in + 15); return 1;
    case 2: { double add; int r = o_line_inlier(fin, in, in + 3, in + 6, in + 9, in + 12, in + 21, in[30], &add); out[0] = add; return r; }
    case 3: o_line_edge_error(in, in + 9, in + 18, in + 21, in + 24, in + 27, out); return 1;
    case 4: { o_se3 X, Y; memcpy(&X, in, sizeof X); o_se3_oplus(&X, in + 12, &Y); memcpy(out, &Y, sizeof Y); return 1; }
    case 5: { o_se3 X; o_tf_to_older_pose(fin, &X); memcpy(out, &X, sizeof X); o_older_pose_to_tf(&X, fout); return 1; }
    case 6: { o_se3 X; o_line_meas m; o_line_blocks B; double Vi[36], T[36], u[6], dl[6]; int ok;
              memcpy(&X, in, sizeof X);
              m.nA = in + 18; m.nB = in + 21; m.nMa = in + 24; m.nMb = in + 33; m.oA = in + 42; m.oB = in + 45; m.oMa = in + 48; m.oMb = in + 57;
              o_match_blocks(&X, in + 12, &m, in[66], in[67], (int)in[68], &B);
              memcpy(out, &B, sizeof B);
              out[120] = o_match_chi2(&X, in + 12, &m, in[66], in[67], (int)in[68]);
              ok = o_match_eliminate(&B, in[69], Vi, T, u);
              memcpy(out + 121, Vi, sizeof Vi); memcpy(out + 157, T, sizeof T); memcpy(out + 193, u, sizeof u);
              o_match_backsub(&B, Vi, in + 70, dl); memcpy(out + 199, dl, sizeof dl);
              return ok; }
    case 7: out[0] = (double)o_relmotion_inlier(in, in + 9, in + 12, in + 15, in + 18, in + 21, in[24], in[25]);
            out[1] = o_relmotion_residual(in, in + 9, in + 12, in + 15, in + 26, in + 35, in + 18, in + 21, in + 44, in + 53);
            out[2] = (double)o_relmotion_degenerate(in + 62, in[80]);
            { double q[4], R2[9]; o_r2q(in, q); o_q2r(q, R2); memcpy(out + 3, q, sizeof q); memcpy(out + 7, R2, sizeof R2); }
            return 1;
    case 8: { o_point_model pm; pm.raster_cov_x = in[0]; pm.raster_cov_y = in[1]; pm.sigma_depth = in[2]; out[0] = o_error_function2(fin, fin + 4, fin + 8, &pm);
              o_project_pt_line(in + 3, in + 6, in + 9, out + 1);
              return o_point_information(fin, in[12], in[13], in[14], in[15], in[16], out + 4); }
    case 9: { o_tfc t; int i, n = (int)in[0]; o_tfc_reset(&t); for (i = 0; i < n; i++) o_tfc_add(&t, fin + 7 * i, fin + 7 * i + 3, fin[7 * i + 6]); o_tfc_get(&t, fout); return t.n; }
    case 10: { o_se3 X; o_point_meas m; o_point_blocks B; double Vi[9], T[36], u[6], dl[3]; int ok;
               memcpy(&X, in, sizeof X);
               m.mn = in + 15; m.mo = in + 18; m.In = in + 21; m.Io = in + 30;
               o_ptmatch_blocks(&X, in + 12, &m, in[39], (int)in[40], &B);
               memcpy(out, &B, sizeof B);
               out[72] = o_ptmatch_chi2(&X, in + 12, &m, in[39], (int)in[40]);
               ok = o_ptmatch_eliminate(&B, in[41], Vi, T, u);
               memcpy(out + 73, Vi, sizeof Vi); memcpy(out + 82, T, sizeof T); memcpy(out + 118, u, sizeof u);
               o_ptmatch_backsub(&B, Vi, in + 42, dl); memcpy(out + 124, dl, sizeof dl);
               return ok; }
    case 11: { double U[9], sg[3], V[9]; o_svd3(in, U, sg, V); memcpy(out, U, sizeof U); memcpy(out + 9, sg, sizeof sg); memcpy(out + 12, V, sizeof V); out[21] = o_det3(in); return 1; }
  }
  return -1;
}

/* ================================================================ legacy point-feature RANSAC
 * Node::getRelativeTransformationTo (src/node.cpp:1134-1338) without its g2o step (:1283-1327, off by default), with
 * sample_matches_prefer_by_distance (:1085-1108), getTransformFromMatches (src/transformation_estimation_euclidean.cpp:7-56)
 * and computeInliersAndError (:1021-1080).  Sequential twin of csrc/lf_pair_legacy.hip; rand() -> the counter generator
 * (draw d of iteration n = counter 20002 n + d), std::sort's tie order -> stable.  pts: float4 per feature.
 * Returns found; out_inl = indices into the caller's match arrays, in the kept (ascending distance) order. */
static int o_legacy_score(const float *pq, const float *pt, const int *mq, const int *mt, const int *perm, int n, const float *tf,
                          double thr2, const o_point_model *pm, int *list, double *err_out) {
  int i, cnt = 0;
  double mean = 0.0;
  for (i = 0; i < n; i++) {
    int m = perm[i];
    const float *x1 = pq + 4 * (size_t)mq[m], *x2 = pt + 4 * (size_t)mt[m];
    double e;
    if (x1[2] == 0.0f || x2[2] == 0.0f) continue;
    e = o_error_function2(x1, x2, tf, pm);
    if (e > thr2) continue;
    if (!(e >= 0.0)) continue;
    mean += e;
    list[cnt++] = i;
  }
  *err_out = cnt < 3 ? 1e9 : sqrt(mean / (double)cnt);
  return cnt;
}
static int o_legacy_transform(const float *pq, const float *pt, const int *mq, const int *mt, const int *perm, const int *list,
                              int cnt, float max_dist_m, float *tf) {
  o_tfc t;
  int k, c, have_prev = 0;
  float pf[3] = {0, 0, 0}, pp[3] = {0, 0, 0};
  o_tfc_reset(&t);
  for (k = 0; k < cnt; k++) {
    int m = perm[list[k]];
    const float *from = pq + 4 * (size_t)mq[m], *to = pt + 4 * (size_t)mt[m];
    float w;
    if (from[2] != from[2] || to[2] != to[2]) continue;
    w = 1 / (to[2] + from[2]);
    if (max_dist_m > 0) {
      if (have_prev) {
        float df = ((from[0] - pf[0]) * (from[0] - pf[0]) + (from[1] - pf[1]) * (from[1] - pf[1])) + (from[2] - pf[2]) * (from[2] - pf[2]);
        float dt = ((to[0] - pp[0]) * (to[0] - pp[0]) + (to[1] - pp[1]) * (to[1] - pp[1])) + (to[2] - pp[2]) * (to[2] - pp[2]);
        float d = df - dt;
        if ((d < 0 ? -d : d) > max_dist_m * max_dist_m) return 0;
      }
      for (c = 0; c < 3; c++) { pf[c] = from[c]; pp[c] = to[c]; }
      have_prev = 1;
    }
    o_tfc_add(&t, from, to, w);
  }
  o_tfc_get(&t, tf);
  return 1;
}
int oracle_legacy_ransac(const float *pts_q, const float *pts_t, const int *mq, const int *mt, const float *md, int n,
                         int min_matches, int iterations, double max_dist_for_inliers, uint64_t seed, uint64_t stream, float *T,
                         float *rmse_out, int *out_inl, int *n_inl, int *dbg /* [3] valid iterations, best iteration, iterations run */) {
  o_point_model pm;
  const float max_dist_m = (float)max_dist_for_inliers;
  float rmse = 1e6f;
  int i, it, nbest = 0, valid_iterations = 0, best_iter = -1, real_iterations = 0, enough = 0;
  int *perm, *cur, *ref, *best;
  o_point_model_init(&pm);
  for (i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  *n_inl = 0; *rmse_out = rmse;
  if (dbg) { dbg[0] = 0; dbg[1] = -1; dbg[2] = 0; }
  if (!(n > min_matches)) return 0;
  perm = (int *)malloc(sizeof(int) * (size_t)n * 4); cur = perm + n; ref = cur + n; best = ref + n;
  {
    unsigned min_thr = (unsigned)min_matches;
    const double thr2 = (double)(max_dist_m * max_dist_m);
    if ((double)min_thr > 0.75 * (double)n) min_thr = (unsigned)(0.75 * (double)n);
    for (i = 0; i < n; i++) {
      int j, rank = 0;
      for (j = 0; j < n; j++) rank += (md[j] < md[i] || (md[j] == md[i] && j < i)) ? 1 : 0;
      perm[rank] = i;
    }
    for (it = 0; it < iterations && n >= 4; it++) {
      double refined_error = 1e6, inlier_error = 0.0;
      int nref = 0, ncur, refinements;
      float rtf[16];
      for (i = 0; i < 16; i++) rtf[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      {
        int ids[4], ns = 0, safety = 0, k;
        uint64_t ctr = (uint64_t)it * 20002ull;
        while (ns < 4) {
          int id1 = (int)(o_rand31(seed, stream, ctr) % (uint32_t)n), id2 = (int)(o_rand31(seed, stream, ctr + 1) % (uint32_t)n), pos = 0, dup = 0;
          ctr += 2;
          if (id1 > id2) id1 = id2;
          while (pos < ns && ids[pos] <= id1) { if (ids[pos] == id1) dup = 1; pos++; }
          if (!dup) { for (k = ns; k > pos; k--) ids[k] = ids[k - 1]; ids[pos] = id1; ns++; }
          if (++safety > 10000) break;
        }
        for (k = 0; k < ns; k++) cur[k] = ids[k];
        ncur = ns;
      }
      real_iterations++;
      for (refinements = 1; refinements < 20; refinements++) {
        float tf[16];
        int nan = 0;
        if (!o_legacy_transform(pts_q, pts_t, mq, mt, perm, cur, ncur, max_dist_m, tf)) break;
        for (i = 0; i < 16; i++) nan = nan || (tf[i] != tf[i]);
        if (nan) break;
        ncur = o_legacy_score(pts_q, pts_t, mq, mt, perm, n, tf, thr2, &pm, cur, &inlier_error);
        if ((unsigned)ncur < min_thr || inlier_error > (double)max_dist_m) break;
        if (ncur >= nref && inlier_error <= refined_error) {
          int prev = nref;
          memcpy(rtf, tf, sizeof rtf);
          memcpy(ref, cur, sizeof(int) * (size_t)ncur);
          nref = ncur;
          refined_error = inlier_error;
          if (ncur == prev) break;
        } else break;
      }
      if (nref > 0) {
        valid_iterations++;
        if (refined_error <= (double)rmse && nref >= nbest && (unsigned)nref >= min_thr) {
          rmse = (float)refined_error;
          memcpy(T, rtf, sizeof rtf);
          memcpy(best, ref, sizeof(int) * (size_t)nref);
          nbest = nref;
          best_iter = it;
          if ((double)nref > (double)n * 0.5) it += 10;
          if ((double)nref > (double)n * 0.75) it += 10;
          if ((double)nref > (double)n * 0.8) break;
        }
      }
    }
    if (valid_iterations == 0) {
      float I4[16];
      double inlier_error;
      int nc;
      for (i = 0; i < 16; i++) I4[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      nc = o_legacy_score(pts_q, pts_t, mq, mt, perm, n, I4, thr2, &pm, cur, &inlier_error);
      if ((unsigned)nc > min_thr && inlier_error < (double)max_dist_m) {
        memcpy(T, I4, sizeof I4);
        memcpy(best, cur, sizeof(int) * (size_t)nc);
        nbest = nc;
        rmse = (float)inlier_error;
        valid_iterations++;
      }
    }
    enough = (unsigned)nbest >= min_thr;
  }
  for (i = 0; i < nbest; i++) out_inl[i] = perm[best[i]];
  *n_inl = nbest; *rmse_out = rmse;
  if (dbg) { dbg[0] = valid_iterations; dbg[1] = best_iter; dbg[2] = real_iterations; }
  free(perm);
  return enough;
}
