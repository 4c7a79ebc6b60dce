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
 * the IEEE-only functions of lineslam_amd/csrc/lf_pose.h (shared with the HIP kernels so that both
 * sides can be compared bit for bit); THIS file is the sequential composition and is pinned by
 * analytic known-answer tests (tests/test_oracle_pair.py: exact rigid motions are recovered).
 * rand() -> lf_rand31 (see front_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "../include/linefront.h"
#include "../lineslam_amd/csrc/lf_pose.h"

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
  for (i = 0; i < n1; ++i)
    for (j = 0; j < n2; ++j) {
      double v = 100;
      if ((f1[i].r[0] * f2[j].r[0] + f1[i].r[1] * f2[j].r[1] > cosang) &&
          (p_line_line2d(&f1[i], &f2[j]) < lineDistThresh) && (p_overlap(&f1[i], &f2[j]) > lineOverlapThresh)) {
        double s = 0;
        int k;
        for (k = 0; k < 72; k++) { double d = f1[i].des[k] - f2[j].des[k]; s += d * d; }
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
 * getTransformFromHybridMatchesG2O, line edges only (transformation_estimation.cpp:218-461):
 * g2o Levenberg on {older camera pose, one 6-d landmark per match}; newer camera fixed at I.    */
static void p_meas(const lf_line_record *train, const lf_line_record *query, int tq, int tt, lf_line_meas *m) {
  m->nA = query[tq].A; m->nB = query[tq].B; m->nMa = query[tq].DUa; m->nMb = query[tq].DUb;
  m->oA = train[tt].A; m->oB = train[tt].B; m->oMa = train[tt].DUa; m->oMb = train[tt].DUb;
}
int oracle_refine_g2o(const lf_line_record *train, const lf_line_record *query, const int *mq, const int *mt,
                      int n, float *tf, int iterations, const lf_params *P, double *chi_out) {
  lf_se3 X, Xn;
  double *L = (double *)malloc(sizeof(double) * 6 * (size_t)(n + 1)), *Ln = (double *)malloc(sizeof(double) * 6 * (size_t)(n + 1));
  lf_line_meas *M = (lf_line_meas *)malloc(sizeof(lf_line_meas) * (size_t)(n + 1));
  lf_line_blocks *B = (lf_line_blocks *)malloc(sizeof(lf_line_blocks) * (size_t)(n + 1));
  double *Vi = (double *)malloc(sizeof(double) * 36 * (size_t)(n + 1));
  double lambda = 0, ni = 2, currentChi = 0, wgt = P->g2o_line_error_weight, hd = P->g2o_BA_kernel_delta;
  int hub = P->g2o_BA_use_kernel, it, k, i, done_iters = 0;
  lf_tf_to_older_pose(tf, &X);
  for (k = 0; k < n; k++) {
    p_meas(train, query, mq[k], mt[k], &M[k]);
    for (i = 0; i < 3; i++) { L[6 * k + i] = query[mq[k]].A[i]; L[6 * k + 3 + i] = query[mq[k]].B[i]; }   /* :326-328 */
  }
  for (it = 0; it < iterations && n > 0; it++) {
    double Hpp[36], bp[6], rho = 0, tempChi;
    int qmax = 0;
    currentChi = 0;
    for (k = 0; k < n; k++) currentChi += lf_match_chi2(&X, &L[6 * k], &M[k], wgt, hd, hub);
    for (i = 0; i < 36; i++) Hpp[i] = 0;
    for (i = 0; i < 6; i++) bp[i] = 0;
    for (k = 0; k < n; k++) {
      lf_match_blocks(&X, &L[6 * k], &M[k], wgt, hd, hub, &B[k]);
      for (i = 0; i < 36; i++) Hpp[i] += B[k].Hpp[i];
      for (i = 0; i < 6; i++) bp[i] += B[k].bp[i];
    }
    if (it == 0) {   /* computeLambdaInit: tau * max |diagonal| */
      double mx = 0;
      for (i = 0; i < 6; i++) if (fabs(Hpp[7 * i]) > mx) mx = fabs(Hpp[7 * i]);
      for (k = 0; k < n; k++) for (i = 0; i < 6; i++) if (fabs(B[k].V[7 * i]) > mx) mx = fabs(B[k].V[7 * i]);
      lambda = 1e-5 * mx;
      ni = 2;
    }
    do {
      double S[36], g[6], dp[6], scale = 0;
      int ok2 = 1;
      for (i = 0; i < 36; i++) S[i] = Hpp[i];
      for (i = 0; i < 6; i++) { S[7 * i] += lambda; g[i] = bp[i]; }
      for (k = 0; k < n; k++) {
        double T[36], u[6];
        if (!lf_match_eliminate(&B[k], lambda, &Vi[36 * k], T, u)) { ok2 = 0; break; }
        for (i = 0; i < 36; i++) S[i] -= T[i];
        for (i = 0; i < 6; i++) g[i] -= u[i];
      }
      if (ok2) { double A[36]; for (i = 0; i < 36; i++) A[i] = S[i]; for (i = 0; i < 6; i++) dp[i] = g[i]; ok2 = lf_solve6(A, dp, 1); }
      tempChi = DBL_MAX;
      if (ok2) {
        lf_se3_oplus(&X, dp, &Xn);
        for (i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + bp[i]);
        tempChi = 0;
        for (k = 0; k < n; k++) {
          double dl[6], sk = 0;
          lf_match_backsub(&B[k], &Vi[36 * k], dp, dl);
          for (i = 0; i < 6; i++) { Ln[6 * k + i] = L[6 * k + i] + dl[i]; sk += dl[i] * (lambda * dl[i] + B[k].bl[i]); }
          scale += sk;
        }
        for (k = 0; k < n; k++) tempChi += lf_match_chi2(&Xn, &Ln[6 * k], &M[k], wgt, hd, hub);
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
      } else {
        lambda *= ni;
        ni *= 2;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    done_iters = it + 1;
    if (qmax == 10 || rho == 0) break;
  }
  lf_older_pose_to_tf(&X, tf);
  if (chi_out) *chi_out = currentChi;
  free(L); free(Ln); free(M); free(B); free(Vi);
  return done_iters;
}

/* getTransform_PtsLines_ransac with nPt = 0 (motion.cpp:605-849).
 *   train = older node, query = newer node; (mq, mt)[nLn] = all line matches (queryIdx, trainIdx)
 *   tf_out: 4x4 row-major float, query -> train;  inl[]: indices into the match list
 * returns 1 if enough inliers (the function's bool), 0 otherwise.                                 */
int oracle_pose_lines_ransac(const lf_line_record *train, const lf_line_record *query, const int *mq,
                             const int *mt, int nLn, int id_train, int id_query, const lf_params *P,
                             uint64_t stream, float *tf_out, float *rmse_out, int *inl, int *n_inl,
                             int *dbg /* [4]: best ransac iter, best count, refine rounds, 0 */) {
  int min_inlier = P->min_feature_matches, lw = P->line_match_number_weight, maxIter = P->ransac_iters_line_motion;
  double thr = P->max_mah_dist_for_inliers;
  int *indexes, *best_set, *cur_set, *ref_set, nbest = 0, nref = 0, iter, i, best_iter = -1, rounds = 0;
  float tf_best[16], sse_best = 1e9f, refined_tf[16];
  double refined_rmse;
  uint64_t ctr = 0;
  int idd = id_train - id_query;
  *n_inl = 0;
  for (i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (dbg) dbg[0] = dbg[1] = dbg[2] = dbg[3] = 0;
  if (0 + nLn * lw < min_inlier) { *rmse_out = 1e9f; return 0; }                           /* :621-624 */
  if (min_inlier > 0.7 * (0 + nLn * lw)) min_inlier = (int)(0.7 * (0 + nLn * lw));          /* :626-628 */
  if ((idd < 0 ? -idd : idd) > 50) min_inlier = P->min_matches_loopclose;                   /* :631-633 */
  if (nLn < 3) { *rmse_out = 1e9f; return 0; }   /* random_unique(3) needs 3 elements; the reference would read out of range */
  indexes = (int *)malloc(sizeof(int) * (size_t)nLn * 4);
  best_set = indexes + nLn; cur_set = best_set + nLn; ref_set = cur_set + nLn;
  for (i = 0; i < nLn; i++) indexes[i] = i;
  for (iter = 0; iter < maxIter; iter++) {
    double la[18], lb[18], R[9], t[3];
    float tf[16], sse = 0;
    int nc = 0, b = 0, left = nLn, s;
    for (s = 0; s < 3; s++) {   /* random_unique(indexes, 3) */
      int r = b + (int)(lf_rand31(P->rng_seed, stream, ctr++) % (uint32_t)left);
      int tmp = indexes[b]; indexes[b] = indexes[r]; indexes[r] = tmp;
      ++b; --left;
    }
    for (s = 0; s < 3; s++) {   /* getTransform_Line_svd: computeRelativeMotion_svd(query, train) */
      int k = indexes[s], c;
      for (c = 0; c < 3; c++) {
        la[6 * s + c] = query[mq[k]].A[c]; la[6 * s + 3 + c] = query[mq[k]].B[c];
        lb[6 * s + c] = train[mt[k]].A[c]; lb[6 * s + 3 + c] = train[mt[k]].B[c];
      }
    }
    if (!lf_rel_motion_lines(la, lb, 3, R, t)) continue;
    for (i = 0; i < 3; i++) { int c; for (c = 0; c < 3; c++) tf[4 * i + c] = (float)R[3 * i + c]; tf[4 * i + 3] = (float)t[i]; }
    tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
    for (i = 0; i < nLn; ++i) {
      double add;
      if (lf_line_inlier(tf, query[mq[i]].A, query[mq[i]].B, train[mt[i]].A, train[mt[i]].B, train[mt[i]].DUa,
                         train[mt[i]].DUb, thr, &add)) { cur_set[nc++] = i; sse += add; }
    }
    if (0 + lw * nc > 0 + lw * nbest) {
      memcpy(best_set, cur_set, sizeof(int) * (size_t)nc);
      nbest = nc; best_iter = iter;
      memcpy(tf_best, tf, sizeof tf);
      sse_best = sse;
    }
  }
  if (dbg) { dbg[0] = best_iter; dbg[1] = nbest; }
  if (0 + nbest < 3) { free(indexes); *rmse_out = 1e9f; return 0; }                         /* :725-728 */
  memcpy(refined_tf, tf_best, sizeof tf_best);
  {
    int *bq = (int *)malloc(sizeof(int) * (size_t)nLn * 2), *bt = bq + nLn;
    for (i = 0; i < nbest; i++) { bq[i] = mq[best_set[i]]; bt[i] = mt[best_set[i]]; }
    memcpy(refined_tf, tf_best, sizeof tf_best);
    oracle_refine_g2o(train, query, bq, bt, nbest, refined_tf, 25, P, NULL);               /* :730 */
    refined_rmse = sqrt(sse_best / (0 + nbest));                                            /* :731 */
    for (iter = 0; iter < 20; ++iter) {                                                     /* :775-839 */
      int nc = 0;
      double tmp_sse = 0;
      for (i = 0; i < nLn; ++i) {
        double add;
        if (lf_line_inlier(refined_tf, query[mq[i]].A, query[mq[i]].B, train[mt[i]].A, train[mt[i]].B,
                           train[mt[i]].DUa, train[mt[i]].DUb, thr, &add)) { cur_set[nc++] = i; tmp_sse += add; }
      }
      if (0 + nc * lw > 0 + nref * lw) {
        memcpy(ref_set, cur_set, sizeof(int) * (size_t)nc);
        nref = nc;
        refined_rmse = sqrt(tmp_sse / (0 + nc));
        for (i = 0; i < nref; i++) { bq[i] = mq[ref_set[i]]; bt[i] = mt[ref_set[i]]; }
        oracle_refine_g2o(train, query, bq, bt, nref, refined_tf, 20, P, NULL);
        rounds++;
      } else break;
    }
    free(bq);
  }
  if (dbg) dbg[2] = rounds;
  for (i = 0; i < nref; i++) inl[i] = ref_set[i];
  *n_inl = nref;
  *rmse_out = (float)refined_rmse;
  memcpy(tf_out, refined_tf, sizeof refined_tf);
  free(indexes);
  return (0 + lw * nref) >= min_inlier;
}

/* exported primitive wrappers for known-answer tests */
int oracle_rel_motion_lines(const double *la, const double *lb, int n, double *R, double *t) { return lf_rel_motion_lines(la, lb, n, R, t); }
