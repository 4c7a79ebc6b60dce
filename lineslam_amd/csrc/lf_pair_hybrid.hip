// lf_pair_hybrid.hip -- getTransform_PtsLines_ransac (src/line/motion.cpp:605-849) with BOTH point and
// line matches (BASELINE.json config 3), one wavefront per node pair.  Launched instead of k_pose when the
// caller supplies 3D points + point matches (Node::feature_locations_3d_ / MatchingResult::all_matches;
// ORB extraction and descriptor matching themselves are SURVEY 8f "next" and stay on the caller's side).
//
//   mixed minimal samples : getTransform_Lns_Pts_pcl (motion.cpp:530-579), weighted Kabsch
//   point scoring         : errorFunction2 (src/misc.cpp:699-786)
//   point edges           : EdgeSE3PointXYZ (src/line/edge_se3_ptxyz.cpp:84-90), information =
//                           inverse compPt3dCov (transformation_estimation.cpp:267,283)
// Same mapping as k_pose: samples generated serially, one hypothesis per lane, arg-max by shuffles, LM with
// one landmark per lane and per-lane Schur elimination; every sum in the oracle's order (points first, then
// lines -- the order in which the reference adds vertices and edges).
#include "lf_pair.h"
#include "lf_pose.h"
#include <float.h>

typedef unsigned long long u64;
#define HP_SLOT (LF_MAX_PT_MATCHES / 64)
#define HL_SLOT (LF_MAX_MATCHES / 64)

__device__ __forceinline__ int h_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 h_lt() { return (1ull << h_lane()) - 1ull; }
__device__ __forceinline__ double h_rl64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

struct HShared {
  int idx[LF_MAX_PT_MATCHES + LF_MAX_MATCHES];
  unsigned short smp[LF_RANSAC_MAX_ITERS * 3];
  int pset[LF_MAX_PT_MATCHES], lset[LF_MAX_MATCHES];     // current inlier lists
  int pcur[LF_MAX_PT_MATCHES], lcur[LF_MAX_MATCHES];     // scratch lists of the re-scoring loop
  lf_se3 xp[12];                                         // X (+) (+-1e-9 e_d) of the current linearisation (lf_perturbed_poses)
};
struct HCtx {
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
#define WL_B (WP_M + LF_MAX_PT_MATCHES * 24)            /* line blocks    [256][120] */
#define WL_VI (WL_B + LF_MAX_MATCHES * 120)
#define WL_TU (WL_VI + LF_MAX_MATCHES * 36)
#define WL_L (WL_TU + LF_MAX_MATCHES * 42)
#define WL_LN (WL_L + LF_MAX_MATCHES * 6)
#define W_TOTAL (WL_LN + LF_MAX_MATCHES * 6)

__device__ __forceinline__ void h_lmeas(const HCtx &pc, int k, lf_line_meas *m) {
  const lf_line_record *q = &pc.query[pc.mq[k]], *t = &pc.train[pc.mt[k]];
  m->nA = q->A; m->nB = q->B; m->nMa = q->DUa; m->nMb = q->DUb;
  m->oA = t->A; m->oB = t->B; m->oMa = t->DUa; m->oMb = t->DUb;
}
template <int NS>
__device__ __forceinline__ double h_ordered_sum(const double *v, int n, double s) {
#pragma unroll
  for (int h = 0; h < NS; h++) {
    int cnt = n - 64 * h;
    if (cnt > 64) cnt = 64;
    for (int l = 0; l < cnt; l++) s += h_rl64(v[h], l);
  }
  return s;
}


// acc (+/-)= base[k * stride] for k = 0..n-1, strictly in that order; the loads of 8 rows are issued together
// (they do not depend on the running sum), the additions stay sequential.
template <bool SUB>
__device__ __forceinline__ double h_walk(const double *base, size_t stride, int n, double acc) {
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = base[(size_t)(k + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; j++) acc = SUB ? acc - v[j] : acc + v[j];
  }
  for (; k < n; k++) acc = SUB ? acc - base[(size_t)k * stride] : acc + base[(size_t)k * stride];
  return acc;
}

// getTransformFromHybridMatchesG2O with point and line edges; sequential twin: oracle_refine_hybrid.
__device__ void h_refine(HShared &S, const HCtx &pc, const int *pset, int np, const int *lset, int nl, float *tf, int iterations) {
  const int lane = h_lane();
  const double wgt = pc.P.g2o_line_error_weight, hd = pc.P.g2o_BA_kernel_delta;
  const int hub = pc.P.g2o_BA_use_kernel;
  double *ws = pc.ws;
  lf_se3 X, Xn;
  double lambda = 0, ni = 2, currentChi = 0;
  lf_tf_to_older_pose(tf, &X);
  for (int h = 0; h < HP_SLOT; h++) {
    int i = lane + 64 * h;
    if (i < np) {
      int k = pset[i];
      const float *qn = pc.qpts + 4 * (size_t)pc.pq[k], *qo = pc.tpts + 4 * (size_t)pc.pt[k];
      double *m = ws + WP_M + 24 * (size_t)i;
      for (int c = 0; c < 3; c++) { m[c] = (double)qn[c]; m[3 + c] = (double)qo[c]; ws[WP_L + 3 * i + c] = (double)qn[c]; }
      lf_point_information(qn, pc.focal, pc.P.stdev_sample_pt_imgline, pc.P.depth_stdev_coeff_c1, pc.P.depth_stdev_coeff_c2 + 0.0 * 0.5, pc.P.depth_stdev_coeff_c3, m + 6);
      lf_point_information(qo, pc.focal, pc.P.stdev_sample_pt_imgline, pc.P.depth_stdev_coeff_c1, pc.P.depth_stdev_coeff_c2 + 0.0 * 0.5, pc.P.depth_stdev_coeff_c3, m + 15);
    }
  }
  for (int h = 0; h < HL_SLOT; h++) {
    int i = lane + 64 * h;
    if (i < nl) {
      const lf_line_record *q = &pc.query[pc.mq[lset[i]]];
      for (int k = 0; k < 3; k++) { ws[WL_L + 6 * i + k] = q->A[k]; ws[WL_L + 6 * i + 3 + k] = q->B[k]; }
    }
  }
  __syncthreads();
  for (int it = 0; it < iterations && (np + nl) > 0; it++) {
    double Hpp[36], bp[6], rho = 0, tempChi, cvp[HP_SLOT], cvl[HL_SLOT];
    int qmax = 0;
    double mxl = 0;
    if (lane < 12) {   // the perturbed poses of the numeric pose Jacobians do not depend on the landmark
      double v[6];
      for (int k = 0; k < 6; k++) v[k] = (k == (lane >> 1)) ? ((lane & 1) ? -1e-9 : 1e-9) : 0.0;
      lf_se3 Xp;
      lf_se3_oplus(&X, v, &Xp);
      S.xp[lane] = Xp;
    }
    __syncthreads();
    for (int h = 0; h < HP_SLOT; h++) {
      int i = lane + 64 * h;
      cvp[h] = 0;
      if (i < np) {
        const double *m = ws + WP_M + 24 * (size_t)i;
        lf_point_meas pmm; pmm.mn = m; pmm.mo = m + 3; pmm.In = m + 6; pmm.Io = m + 15;
        lf_point_blocks Bk;
        double p[3] = {ws[WP_L + 3 * i], ws[WP_L + 3 * i + 1], ws[WP_L + 3 * i + 2]};
        cvp[h] = lf_ptmatch_chi2(&X, p, &pmm, hd, hub);
        lf_ptmatch_blocks_xp(&X, S.xp, p, &pmm, hd, hub, &Bk);
        double *o = ws + WP_B + (size_t)i * 72;
        for (int k = 0; k < 9; k++) o[k] = Bk.V[k];
        for (int k = 0; k < 18; k++) o[9 + k] = Bk.W[k];
        for (int k = 0; k < 3; k++) { o[27 + k] = Bk.bl[k]; double a = lf_fabs(Bk.V[4 * k]); if (a > mxl) mxl = a; }
        for (int k = 0; k < 36; k++) o[30 + k] = Bk.Hpp[k];
        for (int k = 0; k < 6; k++) o[66 + k] = Bk.bp[k];
      }
    }
    for (int h = 0; h < HL_SLOT; h++) {
      int i = lane + 64 * h;
      cvl[h] = 0;
      if (i < nl) {
        lf_line_meas m;
        lf_line_blocks Bk;
        double L[6];
        h_lmeas(pc, lset[i], &m);
        for (int k = 0; k < 6; k++) L[k] = ws[WL_L + 6 * i + k];
        cvl[h] = lf_match_chi2(&X, L, &m, wgt, hd, hub);
        lf_match_blocks_xp(&X, S.xp, L, &m, wgt, hd, hub, &Bk);
        double *o = ws + WL_B + (size_t)i * 120;
        for (int k = 0; k < 36; k++) { o[k] = Bk.V[k]; o[36 + k] = Bk.W[k]; o[78 + k] = Bk.Hpp[k]; }
        for (int k = 0; k < 6; k++) { o[72 + k] = Bk.bl[k]; o[114 + k] = Bk.bp[k]; double a = lf_fabs(Bk.V[7 * k]); if (a > mxl) mxl = a; }
      }
    }
    currentChi = h_ordered_sum<HP_SLOT>(cvp, np, 0.0);
    currentChi = h_ordered_sum<HL_SLOT>(cvl, nl, currentChi);
    __syncthreads();
    double accH = 0;   // Hpp | bp: accumulator lane a (< 42): points first, then lines
    if (lane < 42) {
      accH = h_walk<false>(ws + WP_B + 30 + lane, 72, np, accH);
      accH = h_walk<false>(ws + WL_B + 78 + lane, 120, nl, accH);
    }
#pragma unroll
    for (int a = 0; a < 36; a++) Hpp[a] = h_rl64(accH, a);
#pragma unroll
    for (int a = 0; a < 6; a++) bp[a] = h_rl64(accH, 36 + a);
    if (it == 0) {
      double mx = mxl;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
#pragma unroll
      for (int i = 0; i < 6; i++) if (lf_fabs(Hpp[7 * i]) > mx) mx = lf_fabs(Hpp[7 * i]);
      lambda = 1e-5 * mx;
      ni = 2;
    }
    do {
      double S[36], g[6], dp[6], scale = 0;
      bool okl = true;
      for (int h = 0; h < HP_SLOT; h++) {
        int i = lane + 64 * h;
        if (i < np) {
          lf_point_blocks Bk;
          const double *o = ws + WP_B + (size_t)i * 72;
          for (int k = 0; k < 9; k++) Bk.V[k] = o[k];
          for (int k = 0; k < 18; k++) Bk.W[k] = o[9 + k];
          for (int k = 0; k < 3; k++) Bk.bl[k] = o[27 + k];
          double Vi[9], T[36], u[6];
          if (!lf_ptmatch_eliminate(&Bk, lambda, Vi, T, u)) okl = false;
          for (int k = 0; k < 9; k++) ws[WP_VI + (size_t)i * 9 + k] = Vi[k];
          for (int k = 0; k < 36; k++) ws[WP_TU + (size_t)i * 42 + k] = T[k];
          for (int k = 0; k < 6; k++) ws[WP_TU + (size_t)i * 42 + 36 + k] = u[k];
        }
      }
      for (int h = 0; h < HL_SLOT; h++) {
        int i = lane + 64 * h;
        if (i < nl) {
          lf_line_blocks Bk;
          const double *o = ws + WL_B + (size_t)i * 120;
          for (int k = 0; k < 36; k++) { Bk.V[k] = o[k]; Bk.W[k] = o[36 + k]; }
          for (int k = 0; k < 6; k++) Bk.bl[k] = o[72 + k];
          double Vi[36], T[36], u[6];
          if (!lf_match_eliminate(&Bk, lambda, Vi, T, u)) okl = false;
          for (int k = 0; k < 36; k++) { ws[WL_VI + (size_t)i * 36 + k] = Vi[k]; ws[WL_TU + (size_t)i * 42 + k] = T[k]; }
          for (int k = 0; k < 6; k++) ws[WL_TU + (size_t)i * 42 + 36 + k] = u[k];
        }
      }
      int ok2 = (__ballot(!okl) == 0) ? 1 : 0;
      __syncthreads();
      {
        double acc = accH;
        if (lane < 36 && lane % 7 == 0) acc = accH + lambda;
        if (lane < 42) {
          acc = h_walk<true>(ws + WP_TU + lane, 42, np, acc);
          acc = h_walk<true>(ws + WL_TU + lane, 42, nl, acc);
        }
#pragma unroll
        for (int a = 0; a < 36; a++) S[a] = h_rl64(acc, a);
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = h_rl64(acc, 36 + a);
      }
      if (ok2) {
        double A[36];
#pragma unroll
        for (int i = 0; i < 36; i++) A[i] = S[i];
#pragma unroll
        for (int i = 0; i < 6; i++) dp[i] = g[i];
        ok2 = lf_solve6_u(A, dp, 1);   // the pose system is the same in every lane: scalar pivot branches
      }
      tempChi = DBL_MAX;
      if (ok2) {
        lf_se3_oplus(&X, dp, &Xn);
#pragma unroll
        for (int i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + bp[i]);
        double skp[HP_SLOT], tcp[HP_SLOT], skl[HL_SLOT], tcl[HL_SLOT];
        for (int h = 0; h < HP_SLOT; h++) {
          int i = lane + 64 * h;
          skp[h] = 0; tcp[h] = 0;
          if (i < np) {
            lf_point_blocks Bk;
            const double *o = ws + WP_B + (size_t)i * 72;
            for (int k = 0; k < 18; k++) Bk.W[k] = o[9 + k];
            for (int k = 0; k < 3; k++) Bk.bl[k] = o[27 + k];
            double Vi[9], dl[3], pn[3], s = 0;
            for (int k = 0; k < 9; k++) Vi[k] = ws[WP_VI + (size_t)i * 9 + k];
            lf_ptmatch_backsub(&Bk, Vi, dp, dl);
            for (int k = 0; k < 3; k++) { pn[k] = ws[WP_L + 3 * i + k] + dl[k]; ws[WP_LN + 3 * i + k] = pn[k]; s += dl[k] * (lambda * dl[k] + Bk.bl[k]); }
            skp[h] = s;
            const double *m = ws + WP_M + 24 * (size_t)i;
            lf_point_meas pmm; pmm.mn = m; pmm.mo = m + 3; pmm.In = m + 6; pmm.Io = m + 15;
            tcp[h] = lf_ptmatch_chi2(&Xn, pn, &pmm, hd, hub);
          }
        }
        for (int h = 0; h < HL_SLOT; h++) {
          int i = lane + 64 * h;
          skl[h] = 0; tcl[h] = 0;
          if (i < nl) {
            lf_line_blocks Bk;
            const double *o = ws + WL_B + (size_t)i * 120;
            for (int k = 0; k < 36; k++) Bk.W[k] = o[36 + k];
            for (int k = 0; k < 6; k++) Bk.bl[k] = o[72 + k];
            double Vi[36], dl[6], Ln[6], s = 0;
            for (int k = 0; k < 36; k++) Vi[k] = ws[WL_VI + (size_t)i * 36 + k];
            lf_match_backsub(&Bk, Vi, dp, dl);
            for (int k = 0; k < 6; k++) { Ln[k] = ws[WL_L + 6 * i + k] + dl[k]; ws[WL_LN + 6 * i + k] = Ln[k]; s += dl[k] * (lambda * dl[k] + Bk.bl[k]); }
            skl[h] = s;
            lf_line_meas m;
            h_lmeas(pc, lset[i], &m);
            tcl[h] = lf_match_chi2(&Xn, Ln, &m, wgt, hd, hub);
          }
        }
        scale = h_ordered_sum<HP_SLOT>(skp, np, scale);
        scale = h_ordered_sum<HL_SLOT>(skl, nl, scale);
        tempChi = h_ordered_sum<HP_SLOT>(tcp, np, 0.0);
        tempChi = h_ordered_sum<HL_SLOT>(tcl, nl, tempChi);
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
        for (int h = 0; h < HP_SLOT; h++) { int i = lane + 64 * h; if (i < np) for (int k = 0; k < 3; k++) ws[WP_L + 3 * i + k] = ws[WP_LN + 3 * i + k]; }
        for (int h = 0; h < HL_SLOT; h++) { int i = lane + 64 * h; if (i < nl) for (int k = 0; k < 6; k++) ws[WL_L + 6 * i + k] = ws[WL_LN + 6 * i + k]; }
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
}

// inlier scan of ALL point and line matches with tf (motion.cpp:680-699 / 783-812)
__device__ void h_score(const HCtx &pc, int nPt, int nLn, const float *tf, double thr, int *pset, int *npin, int *lset,
                        int *nlin, float *sse_f_out, double *sse_d_out) {
  const int lane = h_lane();
  double addp[HP_SLOT], addl[HL_SLOT];
  u64 mp[HP_SLOT], ml[HL_SLOT];
  int np = 0, nl = 0;
#pragma unroll
  for (int h = 0; h < HP_SLOT; h++) {
    int i = lane + 64 * h;
    bool in = false;
    addp[h] = 0;
    if (i < nPt) {
      double m = lf_error_function2(pc.qpts + 4 * (size_t)pc.pq[i], pc.tpts + 4 * (size_t)pc.pt[i], tf, &pc.pm);
      if (m < thr * thr) { in = true; addp[h] = m; }
    }
    mp[h] = __ballot(in);
    if (in) pset[np + __popcll(mp[h] & h_lt())] = i;
    np += __popcll(mp[h]);
  }
#pragma unroll
  for (int h = 0; h < HL_SLOT; h++) {
    int i = lane + 64 * h;
    bool in = false;
    addl[h] = 0;
    if (i < nLn) {
      const lf_line_record *q = &pc.query[pc.mq[i]], *t = &pc.train[pc.mt[i]];
      in = lf_line_inlier(tf, q->A, q->B, t->A, t->B, t->DUa, t->DUb, thr, &addl[h]);
    }
    ml[h] = __ballot(in);
    if (in) lset[nl + __popcll(ml[h] & h_lt())] = i;
    nl += __popcll(ml[h]);
  }
  float sf = 0; double sd = 0;
#pragma unroll
  for (int h = 0; h < HP_SLOT; h++) { u64 m = mp[h]; while (m) { int l = __builtin_ctzll(m); m &= m - 1; double a = h_rl64(addp[h], l); sf += a; sd += a; } }
#pragma unroll
  for (int h = 0; h < HL_SLOT; h++) { u64 m = ml[h]; while (m) { int l = __builtin_ctzll(m); m &= m - 1; double a = h_rl64(addl[h], l); sf += a; sd += a; } }
  *npin = np; *nlin = nl; *sse_f_out = sf; *sse_d_out = sd;
  __syncthreads();
}

// minimal-sample model of RANSAC iteration `it` (uniform or per lane); returns validity
__device__ bool h_model(const HCtx &pc, const unsigned short *smp, int it, int nPt, uint64_t stream, float *tf) {
  int spq[3], spt[3], slq[3], slt[3], nsp = 0, nsl = 0;
  for (int s = 0; s < 3; s++) {
    int k = smp[3 * it + s];
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

__global__ void __launch_bounds__(64) k_pose_hybrid(PairConsts c, PairBuffers b) {
  __shared__ HShared S;
  const int pr = blockIdx.x, lane = h_lane();
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  lf_pair_result *res = b.results + pr;
  HCtx pc;
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
  const lf_params &P = c.P;
  int nLn = b.nmatches[pr], nPt = b.npm[pr];
  const int n_all = nLn, np_all = nPt;
  if (nLn > c.match_cap) nLn = c.match_cap;
  if (nLn > LF_MAX_MATCHES) nLn = LF_MAX_MATCHES;
  if (nPt > LF_MAX_PT_MATCHES) nPt = LF_MAX_PT_MATCHES;
  const int nTot = nPt + nLn;
  const long long id_t = (long long)b.frame_ids_t[ft], id_q = (long long)b.frame_ids[fq];
  const uint64_t stream = LF_STREAM_PAIR((uint64_t)id_q, (uint64_t)id_t);
  float tf_out[16];
#pragma unroll
  for (int i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  float rmse_out = 1e9f;
  int valid = 0, n_pinl = 0, n_linl = 0, best_iter = -1, rounds = 0;
  int min_inlier = P.min_feature_matches, lw = P.line_match_number_weight, maxIter = P.ransac_iters_line_motion;
  if (maxIter > LF_RANSAC_MAX_ITERS) maxIter = LF_RANSAC_MAX_ITERS;
  const double thr = P.max_mah_dist_for_inliers;
  bool go = !(nPt + nLn * lw < min_inlier);                                                  // motion.cpp:621-624
  if (min_inlier > 0.7 * (nPt + nLn * lw)) min_inlier = (int)(0.7 * (nPt + nLn * lw));       // :626-628
  { long long d = id_t - id_q; if (d < 0) d = -d; if (d > 50) min_inlier = P.min_matches_loopclose; }   // :631-633
  if (nTot < 3) go = false;
  if (go) {
    for (int i = lane; i < nTot; i += 64) S.idx[i] = i;
    __syncthreads();
    if (lane == 0) {   // sample sequence, serial (partial Fisher-Yates state carries over, :635-658)
      uint64_t ctr = 0;
      for (int it = 0; it < maxIter; it++) {
        int bpos = 0, left = nTot;
        for (int s = 0; s < 3; s++) {
          int r = bpos + (int)(lf_rand31(P.rng_seed, stream, ctr++) % (uint32_t)left);
          int t = S.idx[bpos]; S.idx[bpos] = S.idx[r]; S.idx[r] = t;
          ++bpos; --left;
        }
        S.smp[3 * it] = (unsigned short)S.idx[0]; S.smp[3 * it + 1] = (unsigned short)S.idx[1]; S.smp[3 * it + 2] = (unsigned short)S.idx[2];
      }
    }
    __syncthreads();
    int my_cnt = -1, my_it = 1 << 30;
    for (int it = lane; it < maxIter; it += 64) {   // one hypothesis per lane
      float tf[16];
      if (!h_model(pc, S.smp, it, nPt, stream, tf)) continue;
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
    for (int o = 32; o > 0; o >>= 1) {
      int oc = __shfl_xor(my_cnt, o, 64), oi = __shfl_xor(my_it, o, 64);
      if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
    }
    best_iter = (my_cnt > 0) ? my_it : -1;
    if (best_iter >= 0) {
      float tf_best[16], sse_best = 0;
      double sse_unused;
      h_model(pc, S.smp, best_iter, nPt, stream, tf_best);
      int nbp, nbl;
      h_score(pc, nPt, nLn, tf_best, thr, S.pset, &nbp, S.lset, &nbl, &sse_best, &sse_unused);
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
          h_score(pc, nPt, nLn, refined_tf, thr, S.pcur, &ncp, S.lcur, &ncl, &tmp_f, &tmp_sse);
          if (ncp + ncl * lw > nrp + nrl * lw) {
            for (int i = lane; i < ncp; i += 64) { S.pset[i] = S.pcur[i]; pin[i] = S.pcur[i]; }
            for (int i = lane; i < ncl; i += 64) { S.lset[i] = S.lcur[i]; lin[i] = S.lcur[i]; }
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
  if (lane == 0) {
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
  }
}

size_t lf_pair_hybrid_ws_doubles() { return (size_t)W_TOTAL; }
void lf_pair_hybrid_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t st) {
  hipLaunchKernelGGL(k_pose_hybrid, dim3(n_pairs), dim3(64), 0, st, c, b);
}
