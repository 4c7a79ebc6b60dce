// lf_pair_hybrid.hip -- getTransform_PtsLines_ransac (src/line/motion.cpp:605-849) with BOTH point and
// line matches (BASELINE.json config 3), ONE 256-THREAD WORKGROUP PER NODE PAIR.  Launched instead of k_pose when the
// caller supplies 3D points + point matches (Node::feature_locations_3d_ / MatchingResult::all_matches;
// ORB extraction and descriptor matching themselves are SURVEY 8f "next" and stay on the caller's side).
//
//   mixed minimal samples : getTransform_Lns_Pts_pcl (motion.cpp:530-579), weighted Kabsch
//   point scoring         : errorFunction2 (src/misc.cpp:699-786)
//   point edges           : EdgeSE3PointXYZ (src/line/edge_se3_ptxyz.cpp:84-90), information =
//                           inverse compPt3dCov (transformation_estimation.cpp:267,283)
// Same mapping as k_pose: samples generated serially, one hypothesis per thread, arg-max by shuffles + LDS; LM with the
// resident passes of lf_pose_res.h for the LINE landmarks (six lanes per match, state in LDS / registers) and one thread per
// POINT landmark (its 3x3 blocks are small; they live in the pair's HBM workspace); every sum in the oracle's order (points
// first, then lines -- the order in which the reference adds vertices and edges).  One workgroup per CU (160 KB of LDS).
#include "lf_pair.h"
#include "lf_pose.h"
#include <float.h>
#include "lf_pose_wg.h"
#define R_RED_N (LM_RED_N + 8)               // the ordered sums cover the point landmarks, then the line landmarks
#include "lf_pose_res.h"                      // the LINE landmarks of the refinement are resident, as in k_pose
static_assert(RT_N == PT_N, "the resident line passes and the point loops share the workgroup");

#define HP_SLOT (LF_MAX_PT_MATCHES / PT_N)   // point landmarks per thread

struct HRansacShared {                                    // k_ransac_hybrid
  int idx[LF_MAX_PT_MATCHES + LF_MAX_MATCHES];
  unsigned short smp[LF_RANSAC_MAX_ITERS * 3];
  int wcnt[16], wit[16];
};
struct HShared {
  int pset[LF_MAX_PT_MATCHES], lset[LF_MAX_MATCHES];     // current inlier lists
  int pcur[LF_MAX_PT_MATCHES], lcur[LF_MAX_MATCHES];     // scratch lists of the re-scoring loop
  int scnt[HP_SLOT + 1][PW_N];                           // inlier counts per (slot, wavefront) of h_score
  ResShared rs;                                          // line landmarks + the sums / pose state of the refinement (lf_pose_res.h)
};
struct HCtx {
  const double *cm;                    // the line matches' compact measurements [nLn][R_CM] (in the pair's workspace)
  const lf_line_record *train, *query;
  const float *tpts, *qpts;            // float4 per point
  const int *mq, *mt, *pq, *pt;        // line / point matches (indices into records / point arrays)
  double *ws;                          // LM workspace
  lf_params P;
  lf_point_model pm;
  double focal;
};
// workspace layout (doubles)
#define WP_B 0                                          /* point blocks   [512][72]  V9 W18 bl3 Hpp36 bp6 */
#define WP_VI (WP_B + LF_MAX_PT_MATCHES * 72)           /* [512][9]  */
#define WP_TU (WP_VI + LF_MAX_PT_MATCHES * 9)           /* [512][42] */
#define WP_L (WP_TU + LF_MAX_PT_MATCHES * 42)           /* [512][3]  */
#define WP_LN (WP_L + LF_MAX_PT_MATCHES * 3)            /* [512][3]  */
#define WP_M (WP_LN + LF_MAX_PT_MATCHES * 3)            /* [512][24] mn3 mo3 In9 Io9 */
#define WL_B (WP_M + LF_MAX_PT_MATCHES * 24)            /* [256][R_CM] compact measurements of the line matches */
#define W_WIN (WL_B + LF_MAX_MATCHES * R_CM)            /* five ints: the winner of k_ransac_hybrid (iteration, score, sample) */
#define W_TOTAL (W_WIN + 4)

__device__ __forceinline__ void h_pmeas(const double *ws, int i, lf_point_meas *pmm) {
  const double *m = ws + WP_M + 24 * (size_t)i;
  pmm->mn = m; pmm->mo = m + 3; pmm->In = m + 6; pmm->Io = m + 15;
}

// getTransformFromHybridMatchesG2O with point and line edges; sequential twin: oracle_refine_hybrid.
// Point landmarks: one thread each, their small blocks in the pair's HBM workspace.  Line landmarks: the resident passes of
// k_pose (r_blocks / r_eliminate / r_backsub / r_errchi, lf_pose_res.h: six lanes per match, W | bl and both landmark sets in
// LDS, V and Vi rows in registers).  Every ordered sum runs over the points first, then the lines -- the order in which the
// reference adds vertices and edges: the accumulator lanes of the line passes START from the points' partial sums.
__device__ void h_refine(HShared &S, const HCtx &pc, const int *pset, int np, const int *lset, int nl, float *tf, int iterations) {
  ResShared &M = S.rs;
  const double *cm = pc.cm;
  const int tid = threadIdx.x, ntot = np + nl;
  const double wgt = pc.P.g2o_line_error_weight, hd = pc.P.g2o_BA_kernel_delta;
  const int hub = pc.P.g2o_BA_use_kernel;
  double *ws = pc.ws;
  ResRegs R;
  lf_se3 X, Xn;
  double lambda = 0, ni = 2, currentChi = 0;
  int cur = 0;                                 // M.L[cur], M.X[cur]: the lines' current state (the trial step takes the other set)
  lf_tf_to_older_pose(tf, &X);
  if (tid == 0) M.X[0] = X;
  for (int h = 0; h < HP_SLOT; h++) {
    int i = tid + PT_N * h;
    if (i < np) {
      int k = pset[i];
      const float *qn = pc.qpts + 4 * (size_t)pc.pq[k], *qo = pc.tpts + 4 * (size_t)pc.pt[k];
      double *m = ws + WP_M + 24 * (size_t)i;
      for (int c = 0; c < 3; c++) { m[c] = (double)qn[c]; m[3 + c] = (double)qo[c]; ws[WP_L + 3 * i + c] = (double)qn[c]; }
      lf_point_information(qn, pc.focal, pc.P.stdev_sample_pt_imgline, pc.P.depth_stdev_coeff_c1, pc.P.depth_stdev_coeff_c2 + 0.0 * 0.5, pc.P.depth_stdev_coeff_c3, m + 6);
      lf_point_information(qo, pc.focal, pc.P.stdev_sample_pt_imgline, pc.P.depth_stdev_coeff_c1, pc.P.depth_stdev_coeff_c2 + 0.0 * 0.5, pc.P.depth_stdev_coeff_c3, m + 15);
    }
  }
  if (tid < nl) {
    const double *c = cm + (size_t)lset[tid] * R_CM;       // nA | nB: the landmark starts at the newer camera's measurement
    for (int k = 0; k < 6; k++) M.L[0][6 * tid + k] = c[k];
  }
  if (tid < 8) { M.red[0][ntot + tid] = 0.0; M.red[1][ntot + tid] = 0.0; }   // zero padding of the ordered sums
  __syncthreads();
  if (ntot > 0 && iterations > 0) {
    for (int h = 0; h < HP_SLOT; h++) {
      int i = tid + PT_N * h;
      if (i < np) {
        lf_point_meas pmm; h_pmeas(ws, i, &pmm);
        double p[3] = {ws[WP_L + 3 * i], ws[WP_L + 3 * i + 1], ws[WP_L + 3 * i + 2]};
        M.red[0][i] = lf_ptmatch_chi2(&X, p, &pmm, hd, hub);
      }
    }
    r_errchi(cm, lset, nl, &M.X[0], M.L[0], wgt, hd, hub, M.red[0] + np);
    __syncthreads();
    currentChi = p_sum_published(M.red[0], ntot, 0.0);
  }
  for (int it = 0; it < iterations && ntot > 0; it++) {
    double rho = 0, tempChi;
    int qmax = 0;
    double mxl = 0;
    PT(0);
    r_perturbed_poses(M, &M.X[cur]);
    __syncthreads();
    for (int h = 0; h < HP_SLOT; h++) {        // point landmarks: one thread each
      int i = tid + PT_N * h;
      if (i < np) {
        lf_point_meas pmm; h_pmeas(ws, i, &pmm);
        // the landmark's blocks are produced in place: a workspace row has the layout of lf_point_blocks (V 9 | W 18 | bl 3 | Hpp 36 | bp 6);
        // round 4 filled a local struct and copied it (2.5 KB of scratch per lane)
        lf_point_blocks *Bk = reinterpret_cast<lf_point_blocks *>(ws + WP_B + (size_t)i * 72);
        double p[3] = {ws[WP_L + 3 * i], ws[WP_L + 3 * i + 1], ws[WP_L + 3 * i + 2]};
        lf_ptmatch_blocks_xp(&X, M.xp, p, &pmm, hd, hub, Bk);
        for (int k = 0; k < 3; k++) { double a = lf_fabs(Bk->V[4 * k]); if (a > mxl) mxl = a; }
      }
    }
    __syncthreads();                           // the point blocks are in the workspace
    PT(1);
    // Hpp | bp: accumulator lane a walks the landmarks in order, points first; the line passes continue from there
    const double acc0 = (tid < 42) ? p_walk<false>(ws + WP_B + 30 + tid, 72, np, 0.0) : 0.0;
    r_blocks(M, cm, lset, nl, &M.X[cur], M.L[cur], R, wgt, hd, hub, &mxl, acc0);
    PT(2);
    if (it == 0) {
      double mx = r_block_max(M, mxl);         // (barrier inside: hb visible)
#pragma unroll
      for (int i = 0; i < 6; i++) if (lf_fabs(M.hb[7 * i]) > mx) mx = lf_fabs(M.hb[7 * i]);
      lambda = 1e-5 * mx;
      ni = 2;
    } else __syncthreads();
    do {
      double dp[6], scale = 0;
      int bad = 0;
      for (int h = 0; h < HP_SLOT; h++) {
        int i = tid + PT_N * h;
        if (i < np) {
          const lf_point_blocks *Bk = reinterpret_cast<const lf_point_blocks *>(ws + WP_B + (size_t)i * 72);
          if (!lf_ptmatch_eliminate(Bk, lambda, ws + WP_VI + (size_t)i * 9, ws + WP_TU + (size_t)i * 42, ws + WP_TU + (size_t)i * 42 + 36)) bad = 1;
        }
      }
      __syncthreads();                         // the points' T | u rows are in the workspace: subtracted first, then the lines'
      PT(3);
      bad |= r_eliminate(M, nl, lambda, R, ws + WP_TU, np);
      int ok2 = __syncthreads_or(bad) ? 0 : 1; // (barrier: M.sg visible)
      PT(4);
      if (ok2) {
        double A[36];
#pragma unroll
        for (int i = 0; i < 36; i++) A[i] = M.sg[i];
#pragma unroll
        for (int i = 0; i < 6; i++) dp[i] = M.sg[36 + i];
        ok2 = lf_solve6_u(A, dp, 1);   // the pose system is the same in every thread: scalar pivot branches
      }
      PT(5);
      tempChi = DBL_MAX;
      if (ok2) {
        lf_se3_oplus(&X, dp, &Xn);
        if (tid == 0) M.X[cur ^ 1] = Xn;
#pragma unroll
        for (int i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + M.hb[36 + i]);
        for (int h = 0; h < HP_SLOT; h++) {
          int i = tid + PT_N * h;
          if (i < np) {
            const lf_point_blocks *Bk = reinterpret_cast<const lf_point_blocks *>(ws + WP_B + (size_t)i * 72);
            double dl[3], pn[3], s = 0;
            lf_ptmatch_backsub(Bk, ws + WP_VI + (size_t)i * 9, dp, dl);
            for (int k = 0; k < 3; k++) { pn[k] = ws[WP_L + 3 * i + k] + dl[k]; ws[WP_LN + 3 * i + k] = pn[k]; s += dl[k] * (lambda * dl[k] + Bk->bl[k]); }
            lf_point_meas pmm; h_pmeas(ws, i, &pmm);
            M.red[0][i] = s;
            M.red[1][i] = lf_ptmatch_chi2(&Xn, pn, &pmm, hd, hub);
          }
        }
        PT(6);
        r_backsub(M, nl, dp, lambda, M.L[cur], M.L[cur ^ 1], R, M.red[0] + np);
        __syncthreads();                       // the trial line landmarks and the trial pose are complete
        r_errchi(cm, lset, nl, &M.X[cur ^ 1], M.L[cur ^ 1], wgt, hd, hub, M.red[1] + np);
        __syncthreads();
        tempChi = 0.0;
        p_sum2_published(M.red[0], M.red[1], ntot, &scale, &tempChi);
      }
      PT(7);
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
        cur ^= 1;
        for (int k = tid; k < 3 * np; k += PT_N) ws[WP_L + k] = ws[WP_LN + k];
      } else {
        lambda *= ni;
        ni *= 2;
      }
      __syncthreads();
      qmax++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0) break;
  }
  lf_older_pose_to_tf(&X, tf);
  __syncthreads();                             // (the next refinement overwrites M.X / M.L)
}

// inlier scan of ALL point and line matches with tf (motion.cpp:680-699 / 783-812): lists ascending, sums in list
// order (points, then lines); a non-inlier contributes 0.0, which changes neither sum
__device__ void h_score(HShared &S, const HCtx &pc, int nPt, int nLn, const float *tf, double thr, int *pset, int *npin, int *lset,
                        int *nlin, float *sse_f_out, double *sse_d_out) {
  ResShared &M = S.rs;
  const int tid = threadIdx.x, w = tid >> 6, lane = p_lane(), ntot = nPt + nLn;
  bool inp[HP_SLOT], inl = false;
  u64 mp[HP_SLOT], ml;
#pragma unroll
  for (int h = 0; h < HP_SLOT; h++) {
    int i = tid + PT_N * h;
    double add = 0;
    inp[h] = false;
    if (i < nPt) {
      double m = lf_error_function2(pc.qpts + 4 * (size_t)pc.pq[i], pc.tpts + 4 * (size_t)pc.pt[i], tf, &pc.pm);
      if (m < thr * thr) { inp[h] = true; add = m; }
      M.red[0][i] = add;
    }
    mp[h] = __ballot(inp[h]);
    if (lane == 0) S.scnt[h][w] = __popcll(mp[h]);
  }
  {
    double add = 0;
    if (tid < nLn) {
      const lf_line_record *q = &pc.query[pc.mq[tid]], *t = &pc.train[pc.mt[tid]];
      inl = lf_line_inlier(tf, q->A, q->B, t->A, t->B, t->DUa, t->DUb, thr, &add);
      M.red[0][nPt + tid] = add;
    }
    ml = __ballot(inl);
    if (lane == 0) S.scnt[HP_SLOT][w] = __popcll(ml);
  }
  if (tid < 8) M.red[0][ntot + tid] = 0.0;
  __syncthreads();
  int np = 0, nl = 0;
#pragma unroll
  for (int h = 0; h < HP_SLOT; h++) {
    int before = np;
#pragma unroll
    for (int k = 0; k < PW_N; k++) { int c = S.scnt[h][k]; if (k < w) before += c; np += c; }
    if (inp[h]) pset[before + __popcll(mp[h] & p_lt())] = tid + PT_N * h;
  }
  {
    int before = 0;
#pragma unroll
    for (int k = 0; k < PW_N; k++) { int c = S.scnt[HP_SLOT][k]; if (k < w) before += c; nl += c; }
    if (inl) lset[before + __popcll(ml & p_lt())] = tid;
  }
  float sf = 0; double sd = 0;
  const int n8 = (ntot + 7) & ~7;
  for (int l = 0; l < n8; l += 8) {
    double q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = M.red[0][l + k];
#pragma unroll
    for (int k = 0; k < 8; k++) { sf += q[k]; sd += q[k]; }
  }
  *npin = np; *nlin = nl; *sse_f_out = sf; *sse_d_out = sd;
  __syncthreads();
}

// minimal-sample model of RANSAC iteration `it` (uniform or per lane); returns validity
__device__ bool h_model(const HCtx &pc, const int *smp3, int it, int nPt, uint64_t stream, float *tf) {
  int spq[3], spt[3], slq[3], slt[3], nsp = 0, nsl = 0;
  for (int s = 0; s < 3; s++) {
    int k = smp3[s];
    if (k < nPt) { spq[nsp] = pc.pq[k]; spt[nsp] = pc.pt[k]; nsp++; }
    else { slq[nsl] = pc.mq[k - nPt]; slt[nsl] = pc.mt[k - nPt]; nsl++; }
  }
  if (nsl == 3) {   // getTransform_Line_svd (motion.cpp:581-603)
    double la[18], lb[18], R[9], t[3];
    for (int s = 0; s < 3; s++) {
      const lf_line_record *q = &pc.query[slq[s]], *tr = &pc.train[slt[s]];
      for (int c = 0; c < 3; c++) { la[6 * s + c] = q->A[c]; la[6 * s + 3 + c] = q->B[c]; lb[6 * s + c] = tr->A[c]; lb[6 * s + 3 + c] = tr->B[c]; }
    }
    if (!lf_rel_motion_lines(la, lb, 3, R, t)) return false;
    for (int i = 0; i < 3; i++) { for (int c = 0; c < 3; c++) tf[4 * i + c] = (float)R[3 * i + c]; tf[4 * i + 3] = (float)t[i]; }
    tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
    return true;
  }
  // getTransform_Lns_Pts_pcl (motion.cpp:530-579)
  if (nsp < 1 || nsp + nsl < 3) return false;
  lf_tfc tc;
  lf_tfc_reset(&tc);
  for (int i = 0; i < nsl; ++i) {
    int ptidx = (int)(lf_rand31(pc.P.rng_seed, stream, (1ull << 20) + 3ull * (uint64_t)it + (uint64_t)i) % (uint32_t)nsp);
    int sq = (ptidx == 0) ? spq[0] : (ptidx == 1 ? spq[1] : spq[2]);
    int st = (ptidx == 0) ? spt[0] : (ptidx == 1 ? spt[1] : spt[2]);
    const float *tp = pc.tpts + 4 * (size_t)st, *qp = pc.qpts + 4 * (size_t)sq;
    double tpd[3] = {tp[0], tp[1], tp[2]}, qpd[3] = {qp[0], qp[1], qp[2]}, tprj[3], qprj[3];
    float from[3], to[3];
    int lq = (i == 0) ? slq[0] : (i == 1 ? slq[1] : slq[2]), lt = (i == 0) ? slt[0] : (i == 1 ? slt[1] : slt[2]);
    lf_project_pt_line(tpd, pc.train[lt].A, pc.train[lt].B, tprj);
    lf_project_pt_line(qpd, pc.query[lq].A, pc.query[lq].B, qprj);
    for (int k = 0; k < 3; k++) { from[k] = (float)qprj[k]; to[k] = (float)tprj[k]; }
    if (from[2] != from[2] || to[2] != to[2]) continue;
    float w = 1 / (__builtin_fabsf(to[2]) + __builtin_fabsf(from[2]));
    lf_tfc_add(&tc, from, to, w);
  }
  for (int i = 0; i < nsp; ++i) {
    int sq = (i == 0) ? spq[0] : (i == 1 ? spq[1] : spq[2]), st = (i == 0) ? spt[0] : (i == 1 ? spt[1] : spt[2]);
    const float *from = pc.qpts + 4 * (size_t)sq, *to = pc.tpts + 4 * (size_t)st;
    if (from[2] != from[2] || to[2] != to[2]) continue;
    float w = 1 / (__builtin_fabsf(to[2]) + __builtin_fabsf(from[2]));
    lf_tfc_add(&tc, from, to, w);
  }
  if (tc.n < 3) return false;
  lf_tfc_get(&tc, tf);
  return true;
}

// what both kernels of the pair derive from the launch arguments: the context of the pair and the gates of
// getTransform_PtsLines_ransac before the hypothesis loop (motion.cpp:621-633)
struct HGate { int nLn, nPt, n_all, np_all, nTot, ovf, min_inlier, lw, maxIter; bool go; long long id_t, id_q; };
__device__ __forceinline__ HGate h_setup(const PairConsts &c, const PairBuffers &b, int pr, HCtx *pcp) {
  HCtx &pc = *pcp;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  pc.train = b.recs_t + (size_t)ft * b.line_cap_t;
  pc.query = b.recs + (size_t)fq * c.line_cap;
  pc.tpts = b.pts_t + (size_t)ft * b.pt_cap_t * 4;
  pc.qpts = b.pts + (size_t)fq * b.pt_cap * 4;
  pc.mq = b.match_q + (size_t)pr * c.match_cap;
  pc.mt = b.match_t + (size_t)pr * c.match_cap;
  pc.pq = b.pm_q + (size_t)pr * b.pm_stride;
  pc.pt = b.pm_t + (size_t)pr * b.pm_stride;
  pc.ws = b.ws_h + (size_t)pr * W_TOTAL;
  pc.P = c.P;
  pc.pm = c.pm;
  pc.focal = c.focal;
  pc.cm = pc.ws + WL_B;
  HGate g;
  const lf_params &P = c.P;
  g.nLn = b.nmatches[pr]; g.nPt = b.npm[pr];
  g.n_all = g.nLn; g.np_all = g.nPt;
  if (g.nLn > c.match_cap) g.nLn = c.match_cap;
  if (g.nLn > LF_MAX_MATCHES) g.nLn = LF_MAX_MATCHES;
  if (g.nPt > LF_MAX_PT_MATCHES) g.nPt = LF_MAX_PT_MATCHES;
  if (g.nPt > c.pt_match_cap) g.nPt = c.pt_match_cap;
  g.nTot = g.nPt + g.nLn;
  g.ovf = ((g.n_all > c.match_cap || g.n_all > LF_MAX_MATCHES) ? LF_OVF_MATCHES : 0) | ((g.np_all > g.nPt) ? LF_OVF_PT_MATCHES : 0) |
          ((b.nlines[fq] > c.line_cap || b.nlines_t[ft] > (b.line_cap_t < c.line_cap ? b.line_cap_t : c.line_cap)) ? LF_OVF_LINES : 0);
  g.id_t = (long long)b.frame_ids_t[ft]; g.id_q = (long long)b.frame_ids[fq];
  g.min_inlier = P.min_feature_matches; g.lw = P.line_match_number_weight; g.maxIter = P.ransac_iters_line_motion;
  if (g.maxIter > LF_RANSAC_MAX_ITERS) g.maxIter = LF_RANSAC_MAX_ITERS;
  g.go = !(g.nPt + g.nLn * g.lw < g.min_inlier);                                                              // motion.cpp:621-624
  if (g.min_inlier > 0.7 * (g.nPt + g.nLn * g.lw)) g.min_inlier = (int)(0.7 * (g.nPt + g.nLn * g.lw));        // :626-628
  { long long d = g.id_t - g.id_q; if (d < 0) d = -d; if (d > 50) g.min_inlier = P.min_matches_loopclose; }   // :631-633
  if (g.nTot < 3) g.go = false;
  return g;
}

// ------------------------------------------------------------------------------ k_ransac_hybrid
// The hypothesis stage (motion.cpp:635-723) as its own launch, as k_ransac is for k_pose: one hypothesis per thread scored
// against every point and line match, at several wavefronts per SIMD instead of inside the one-workgroup-per-CU refinement
// kernel.  Leaves the winner (iteration, score, its three samples) in the pair's workspace.
#ifndef HR_N
#define HR_N 256
#endif
__global__ void __launch_bounds__(HR_N) k_ransac_hybrid(PairConsts c, PairBuffers b) {
  __shared__ HRansacShared S;
  const int pr = blockIdx.x, tid = threadIdx.x, lane = p_lane();
  HCtx pc;
  const HGate g = h_setup(c, b, pr, &pc);
  int *win = (int *)(pc.ws + W_WIN);
  if (c.mode == LF_MODE_REFINE) return;
  if (!g.go) { if (tid == 0) { win[0] = -1; win[1] = 0; } return; }
  const int nPt = g.nPt, nLn = g.nLn, nTot = g.nTot, maxIter = g.maxIter, lw = g.lw;
  const double thr = c.P.max_mah_dist_for_inliers;
  const uint64_t stream = LF_STREAM_PAIR((uint64_t)g.id_q, (uint64_t)g.id_t);
  for (int i = tid; i < nTot; i += HR_N) S.idx[i] = i;
  __syncthreads();
  if (tid == 0) {   // sample sequence, serial (partial Fisher-Yates state carries over, :635-658)
    uint64_t ctr = 0;
    for (int it = 0; it < maxIter; it++) {
      int bpos = 0, left = nTot;
      for (int s = 0; s < 3; s++) {
        int r = bpos + (int)(lf_rand31(c.P.rng_seed, stream, ctr++) % (uint32_t)left);
        int t = S.idx[bpos]; S.idx[bpos] = S.idx[r]; S.idx[r] = t;
        ++bpos; --left;
      }
      S.smp[3 * it] = (unsigned short)S.idx[0]; S.smp[3 * it + 1] = (unsigned short)S.idx[1]; S.smp[3 * it + 2] = (unsigned short)S.idx[2];
    }
  }
  __syncthreads();
  int my_cnt = -1, my_it = 1 << 30;
  for (int it = tid; it < maxIter; it += HR_N) {   // one hypothesis per thread
    float tf[16];
    const int s3[3] = {S.smp[3 * it], S.smp[3 * it + 1], S.smp[3 * it + 2]};
    if (!h_model(pc, s3, it, nPt, stream, tf)) continue;
    int ncp = 0, ncl = 0;
    for (int i = 0; i < nPt; ++i) {
      double m = lf_error_function2(pc.qpts + 4 * (size_t)pc.pq[i], pc.tpts + 4 * (size_t)pc.pt[i], tf, &pc.pm);
      ncp += (m < thr * thr);
    }
    for (int i = 0; i < nLn; ++i) {
      double add;
      const lf_line_record *q = &pc.query[pc.mq[i]], *tr = &pc.train[pc.mt[i]];
      ncl += lf_line_inlier(tf, q->A, q->B, tr->A, tr->B, tr->DUa, tr->DUb, thr, &add);
    }
    int score = ncp + lw * ncl;
    if (score > my_cnt) { my_cnt = score; my_it = it; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {   // arg-max: score desc, iteration asc -- wavefront, then the wavefronts
    int oc = __shfl_xor(my_cnt, o, 64), oi = __shfl_xor(my_it, o, 64);
    if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
  }
  if (lane == 0) { S.wcnt[tid >> 6] = my_cnt; S.wit[tid >> 6] = my_it; }
  __syncthreads();
  if (tid == 0) {
    my_cnt = S.wcnt[0]; my_it = S.wit[0];
    for (int w = 1; w < HR_N / 64; w++) {
      int oc = S.wcnt[w], oi = S.wit[w];
      if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
    }
    const int best = my_cnt > 0 ? my_it : -1;
    win[0] = best; win[1] = my_cnt;
    for (int s = 0; s < 3; s++) win[2 + s] = best >= 0 ? (int)S.smp[3 * best + s] : 0;
  }
}

__global__ void __launch_bounds__(PT_N) k_pose_hybrid(PairConsts c, PairBuffers b) {
  __shared__ HShared S;
  const int pr = blockIdx.x, tid = threadIdx.x;
  lf_pair_result *res = b.results + pr;
  HCtx pc;
  const HGate g = h_setup(c, b, pr, &pc);
  const lf_params &P = c.P;
  const int nLn = g.nLn, nPt = g.nPt, n_all = g.n_all, np_all = g.np_all, ovf = g.ovf;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  {   // the measurements of the matched lines, 48 contiguous doubles per match (read by every refinement pass)
    double *cmw = pc.ws + WL_B;
    for (int e = tid; e < nLn * 2; e += PT_N) {
      const int k = e >> 1, h = e & 1;
      const lf_line_record *r = h ? &pc.train[pc.mt[k]] : &pc.query[pc.mq[k]];
      double *o = cmw + (size_t)k * R_CM + 24 * h;
      for (int j = 0; j < 3; j++) { o[j] = r->A[j]; o[3 + j] = r->B[j]; }
      for (int j = 0; j < 9; j++) { o[6 + j] = r->DUa[j]; o[15 + j] = r->DUb[j]; }
    }
    __syncthreads();
  }
  if (c.mode == LF_MODE_REFINE) {
    // getTransformFromHybridMatchesG2O on its own (lf_refine_pair): every match given is an edge, the start value is
    // the transform the host stored in the result slot
    float tf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tf[i] = res->T[i];
    for (int i = tid; i < nPt; i += PT_N) S.pset[i] = i;
    for (int i = tid; i < nLn; i += PT_N) S.lset[i] = i;
    __syncthreads();
    h_refine(S, pc, S.pset, nPt, S.lset, nLn, tf, c.refine_iters);
    if (tid == 0) {
      for (int i = 0; i < 16; i++) res->T[i] = tf[i];
      res->rmse = 0.0f; res->valid = 1; res->n_matches = n_all; res->n_inliers = nLn; res->n_point_matches = np_all;
      res->n_point_inliers = nPt; res->id_older = (int)b.frame_ids_t[ft]; res->id_newer = (int)b.frame_ids[fq];
      res->ransac_best_iter = -1; res->refine_rounds = 0; res->information_scale = 0.0; res->overflow = ovf; res->reserved_ = 0;
    }
    return;
  }
#ifdef LF_POSE_PROFILE
  if (blockIdx.x == 7 && tid == 0) g_pprev = __builtin_amdgcn_s_memtime();
#endif
  const long long id_t = g.id_t, id_q = g.id_q;
  const uint64_t stream = LF_STREAM_PAIR((uint64_t)id_q, (uint64_t)id_t);
  float tf_out[16];
#pragma unroll
  for (int i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  float rmse_out = 1e9f;
  int valid = 0, n_pinl = 0, n_linl = 0, best_iter = -1, rounds = 0;
  const int min_inlier = g.min_inlier, lw = g.lw;
  const double thr = P.max_mah_dist_for_inliers;
  if (g.go) {
    const int *win = (const int *)(pc.ws + W_WIN);       // the winner of k_ransac_hybrid
    best_iter = win[0];
    const int s3[3] = {win[2], win[3], win[4]};
    if (best_iter >= 0) {
      float tf_best[16], sse_best = 0;
      double sse_unused;
      h_model(pc, s3, best_iter, nPt, stream, tf_best);
      int nbp, nbl;
      h_score(S, pc, nPt, nLn, tf_best, thr, S.pset, &nbp, S.lset, &nbl, &sse_best, &sse_unused);
      if (nbp + nbl >= 3) {                                                                  // :725-728
        float refined_tf[16];
#pragma unroll
        for (int i = 0; i < 16; i++) refined_tf[i] = tf_best[i];
        h_refine(S, pc, S.pset, nbp, S.lset, nbl, refined_tf, 25);                              // :730
        double refined_rmse = lf_sqrt(sse_best / (nbp + nbl));                               // :731
        int nrp = 0, nrl = 0;
        int *pin = b.pt_inliers + (size_t)pr * LF_MAX_PT_MATCHES, *lin = b.inliers + (size_t)pr * LF_MAX_MATCHES;
        for (int iter = 0; iter < 20; ++iter) {                                              // :775-839
          float tmp_f; double tmp_sse;
          int ncp, ncl;
          __syncthreads();
          h_score(S, pc, nPt, nLn, refined_tf, thr, S.pcur, &ncp, S.lcur, &ncl, &tmp_f, &tmp_sse);
          if (ncp + ncl * lw > nrp + nrl * lw) {
            for (int i = tid; i < ncp; i += PT_N) { S.pset[i] = S.pcur[i]; pin[i] = S.pcur[i]; }
            for (int i = tid; i < ncl; i += PT_N) { S.lset[i] = S.lcur[i]; lin[i] = S.lcur[i]; }
            __syncthreads();
            nrp = ncp; nrl = ncl;
            refined_rmse = lf_sqrt(tmp_sse / (ncp + ncl));
            h_refine(S, pc, S.pset, nrp, S.lset, nrl, refined_tf, 20);
            rounds++;
          } else break;
        }
        n_pinl = nrp; n_linl = nrl;
        rmse_out = (float)refined_rmse;
#pragma unroll
        for (int i = 0; i < 16; i++) tf_out[i] = refined_tf[i];
        valid = ((nrp + lw * nrl) >= min_inlier) ? 1 : 0;
      }
    }
  }
#ifdef LF_POSE_PROFILE
  PT(8);
  if (blockIdx.x == 7 && tid == 0) {
    printf("k_pose_hybrid prof (kticks) np=%d nl=%d: other %.1f | point blocks %.1f walk + line blocks %.1f point elim %.1f line elim %.1f solve %.1f point backsub %.1f line backsub, chi2, sums %.1f | ransac, scoring %.1f\n",
           nPt, nLn, g_pprof[0] / 1e3, g_pprof[1] / 1e3, g_pprof[2] / 1e3, g_pprof[3] / 1e3, g_pprof[4] / 1e3, g_pprof[5] / 1e3, g_pprof[6] / 1e3, g_pprof[7] / 1e3, g_pprof[8] / 1e3);
    for (int i = 0; i < 16; i++) g_pprof[i] = 0;
  }
#endif
  if (tid == 0) {
    for (int i = 0; i < 16; i++) res->T[i] = tf_out[i];
    res->rmse = rmse_out;
    res->valid = valid;
    res->n_matches = n_all;
    res->n_inliers = n_linl;
    res->n_point_matches = np_all;
    res->n_point_inliers = n_pinl;
    res->id_older = valid ? (int)id_t : -1;
    res->id_newer = valid ? (int)id_q : -1;
    res->ransac_best_iter = best_iter;
    res->refine_rounds = rounds;
    float r2 = rmse_out * rmse_out;
    res->information_scale = valid ? (double)((float)(n_pinl + n_linl * lw) / r2) : 0.0;   // node.cpp:1533-1534
    res->overflow = ovf;
    res->reserved_ = 0;
  }
}

size_t lf_pair_hybrid_ws_doubles() { return (size_t)W_TOTAL; }
void lf_pair_hybrid_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t st) {
  if (c.mode != LF_MODE_REFINE) hipLaunchKernelGGL(k_ransac_hybrid, dim3(n_pairs), dim3(HR_N), 0, st, c, b);
  hipLaunchKernelGGL(k_pose_hybrid, dim3(n_pairs), dim3(PT_N), 0, st, c, b);
}
