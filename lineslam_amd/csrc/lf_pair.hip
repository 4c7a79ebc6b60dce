// lf_pair.hip -- pair solver for gfx950: line matching + relative pose for a batch of node pairs.
//
//   k_match   Node::lineMatching (src/node.cpp:1619-1694): all-pairs gated descriptor distances
//             (one 256-thread block per pair; N1 x N2 x 72 fp64, VALU -- no MFMA: N ~ 10^2),
//             mutual nearest neighbour + ratio tests, ordered emission.
//   k_pose    ONE WAVEFRONT PER PAIR:
//             getTransform_PtsLines_ransac (src/line/motion.cpp:605-849) with line matches: the sample
//             sequence is generated serially (partial Fisher-Yates state carries over, :635-658), the
//             500 hypotheses are solved and scored ONE PER LANE, the winner is the arg-max of the
//             inlier count with the lowest iteration on ties (the sequential "strictly greater" rule,
//             :714-720), found with wavefront shuffles;
//             getTransformFromHybridMatchesG2O (src/transformation_estimation.cpp:218-461): LM with
//             one landmark per lane, 6x6 Schur elimination per lane, pose system accumulated in match
//             order by one accumulator lane per matrix entry (bit-identical to the sequential oracle).
#include "lf_pair.h"
#include "lf_pose.h"
#include <float.h>

typedef unsigned long long u64;
#define NSLOT (LF_MAX_MATCHES / 64)

__device__ __forceinline__ int p_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 p_lt() { return (1ull << p_lane()) - 1ull; }
__device__ __forceinline__ double p_rl64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------ k_match
__device__ __forceinline__ double m_pt_line2d(const double *p, const double *l) {   // utils.cpp:1250-1264
  return lf_fabs((l[0] * p[0] + l[1] * p[1] + l[2])) / lf_sqrt(l[0] * l[0] + l[1] * l[1]);
}
__device__ __forceinline__ double m_norm2(const double *a, const double *b) {
  return lf_sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]));
}
__device__ __forceinline__ double m_project2d(const double *X, const double *A, const double *B) {   // :1612-1618
  double BX0 = X[0] - B[0], BX1 = X[1] - B[1], BA0 = A[0] - B[0], BA1 = A[1] - B[1];
  double n = lf_sqrt(BA0 * BA0 + BA1 * BA1);
  return (BX0 * BA0 + BX1 * BA1) / n / n;
}
__device__ double m_overlap(const lf_line_record *a, const lf_line_record *b) {   // utils.cpp:1620-1638
  if (m_norm2(a->p, a->q) < m_norm2(b->p, b->q)) {
    double lp = m_project2d(a->p, b->p, b->q), lq = m_project2d(a->q, b->p, b->q);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return lf_fabs(lp - lq) * m_norm2(b->p, b->q);
  } else {
    double lp = m_project2d(b->p, a->p, a->q), lq = m_project2d(b->q, a->p, a->q);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return lf_fabs(lp - lq) * m_norm2(a->p, a->q);
  }
}

__global__ void __launch_bounds__(256) k_match(PairConsts c, PairBuffers b) {
  __shared__ int s_pos[512];
  __shared__ double s_val[512];
  __shared__ int s_wbase[4];
  const int pr = blockIdx.x, tid = threadIdx.x;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  int n1 = b.nlines[fq], n2 = b.nlines_t[ft];
  if (n1 > c.line_cap) n1 = c.line_cap;
  if (n2 > b.line_cap_t) n2 = b.line_cap_t;
  if (n2 > c.line_cap) n2 = c.line_cap;
  const lf_line_record *f1 = b.recs + (size_t)fq * c.line_cap, *f2 = b.recs_t + (size_t)ft * b.line_cap_t;
  double *D = b.D + (size_t)pr * c.line_cap * c.line_cap;
  long long idd = (long long)b.frame_ids[fq] - (long long)b.frame_ids_t[ft];
  if (idd < 0) idd = -idd;
  const bool adjacent = !(idd > c.P.adjacent_linematch_window);                       // node.cpp:1505-1507
  const double lineDistThresh = adjacent ? 45 : 80, descDiffThresh = adjacent ? 0.85 : 0.7;
  const double lineOverlapThresh = adjacent ? 0 : -1, ratio = 0.7;
  if (n1 == 0 || n2 == 0) { if (tid == 0) b.nmatches[pr] = 0; return; }
  for (int idx = tid; idx < n1 * n2; idx += 256) {
    int i = idx / n2, j = idx - i * n2;
    const lf_line_record *a = &f1[i], *bb = &f2[j];
    double v = 100;
    if ((a->r[0] * bb->r[0] + a->r[1] * bb->r[1] > c.cos_angle_thresh) &&
        (0.25 * m_pt_line2d(a->p, bb->lineEq2d) + 0.25 * m_pt_line2d(a->q, bb->lineEq2d) +
         0.25 * m_pt_line2d(bb->p, a->lineEq2d) + 0.25 * m_pt_line2d(bb->q, a->lineEq2d) < lineDistThresh) &&
        (m_overlap(a, bb) > lineOverlapThresh)) {
      double s = 0;
      for (int k = 0; k < 72; k++) { double d = a->des[k] - bb->des[k]; s += d * d; }
      v = lf_sqrt(s);
    }
    D[(size_t)i * n2 + j] = v;
  }
  __syncthreads();
  for (int i = tid; i < 512; i += 256) s_pos[i] = -1;
  __syncthreads();
  for (int i = tid; i < n1; i += 256) {
    const double *row = D + (size_t)i * n2;
    double minVal = row[0], rowmin2 = 100, colmin2 = 100;
    int minPos = 0, minP = 0;
    for (int j = 1; j < n2; j++) if (row[j] < minVal) { minVal = row[j]; minPos = j; }   // minMaxLoc: first minimum
    if (!(minVal < descDiffThresh)) continue;
    double minV = D[minPos];
    for (int j = 1; j < n1; j++) { double v = D[(size_t)j * n2 + minPos]; if (v < minV) { minV = v; minP = j; } }
    if (i != minP) continue;
    for (int j = 0; j < n2; ++j) { if (j == minPos) continue; if (rowmin2 > row[j]) rowmin2 = row[j]; }
    for (int j = 0; j < n1; ++j) { if (j == minP) continue; double v = D[(size_t)j * n2 + minPos]; if (colmin2 > v) colmin2 = v; }
    if (rowmin2 * ratio > minVal && colmin2 * ratio > minVal) { s_pos[i] = minPos; s_val[i] = minVal; }
  }
  __syncthreads();
  // ordered emission (the reference loops over i ascending, node.cpp:1656)
  int *mq = b.match_q + (size_t)pr * c.match_cap, *mt = b.match_t + (size_t)pr * c.match_cap;
  double *md = b.match_d + (size_t)pr * c.match_cap;
  int base = 0;
  const int wave = tid >> 6, lane = tid & 63;
  for (int i0 = 0; i0 < n1; i0 += 256) {
    int i = i0 + tid;
    bool has = i < n1 && s_pos[i] >= 0;
    u64 m = __ballot(has);
    if (lane == 0) s_wbase[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; w++) off += s_wbase[w];
    int tot = s_wbase[0] + s_wbase[1] + s_wbase[2] + s_wbase[3];
    if (has) {
      int o = base + off + __popcll(m & p_lt());
      if (o < c.match_cap) { mq[o] = i; mt[o] = s_pos[i]; md[o] = s_val[i]; }
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) b.nmatches[pr] = base;
}

// ------------------------------------------------------------------------------ k_pose
struct PoseShared {
  int idx[LF_MAX_MATCHES];
  unsigned char smp[LF_RANSAC_MAX_ITERS * 3];
  int set[LF_MAX_MATCHES];        // current inlier list (indices into the match list)
};
struct PoseCtx {
  const lf_line_record *train, *query;
  const int *mq, *mt;
  double *wsB, *wsVi, *wsTU, *wsL, *wsLn;
  lf_params P;
};

__device__ __forceinline__ void p_meas(const PoseCtx &pc, int k, lf_line_meas *m) {
  const lf_line_record *q = &pc.query[pc.mq[k]], *t = &pc.train[pc.mt[k]];
  m->nA = q->A; m->nB = q->B; m->nMa = q->DUa; m->nMb = q->DUb;
  m->oA = t->A; m->oB = t->B; m->oMa = t->DUa; m->oMb = t->DUb;
}
// sum of per-landmark values in list order; v[h] belongs to list position lane + 64 h
__device__ __forceinline__ double p_ordered_sum(const double *v, int n, double s) {
#pragma unroll
  for (int h = 0; h < NSLOT; h++) {
    int cnt = n - 64 * h;
    if (cnt > 64) cnt = 64;
    for (int l = 0; l < cnt; l++) s += p_rl64(v[h], l);
  }
  return s;
}


// acc (+/-)= base[k * stride] for k = 0..n-1, strictly in that order; the loads of 8 rows are issued together
// (they do not depend on the running sum), the additions stay sequential.
template <bool SUB>
__device__ __forceinline__ double p_walk(const double *base, size_t stride, int n, double acc) {
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

// getTransformFromHybridMatchesG2O (transformation_estimation.cpp:218-461), line edges only; the
// sequential twin is oracle_refine_g2o.  set[0..n) = match indices (LDS).
__device__ void p_refine(const PoseCtx &pc, const int *set, int n, float *tf, int iterations) {
  const int lane = p_lane();
  const double wgt = pc.P.g2o_line_error_weight, hd = pc.P.g2o_BA_kernel_delta;
  const int hub = pc.P.g2o_BA_use_kernel;
  lf_se3 X, Xn;
  double lambda = 0, ni = 2, currentChi = 0;
  lf_tf_to_older_pose(tf, &X);
  for (int h = 0; h < NSLOT; h++) {
    int i = lane + 64 * h;
    if (i < n) {
      const lf_line_record *q = &pc.query[pc.mq[set[i]]];
      for (int k = 0; k < 3; k++) { pc.wsL[6 * i + k] = q->A[k]; pc.wsL[6 * i + 3 + k] = q->B[k]; }
    }
  }
  __syncthreads();
  for (int it = 0; it < iterations && n > 0; it++) {
    double Hpp[36], bp[6], rho = 0, tempChi, cv[NSLOT];
    int qmax = 0;
    double mxl = 0;
    for (int h = 0; h < NSLOT; h++) {
      int i = lane + 64 * h;
      cv[h] = 0;
      if (i < n) {
        lf_line_meas m;
        lf_line_blocks Bk;
        double L[6];
        p_meas(pc, set[i], &m);
        for (int k = 0; k < 6; k++) L[k] = pc.wsL[6 * i + k];
        cv[h] = lf_match_chi2(&X, L, &m, wgt, hd, hub);
        lf_match_blocks(&X, L, &m, wgt, hd, hub, &Bk);
        double *o = pc.wsB + (size_t)i * 120;
        for (int k = 0; k < 36; k++) { o[k] = Bk.V[k]; o[36 + k] = Bk.W[k]; o[78 + k] = Bk.Hpp[k]; }
        for (int k = 0; k < 6; k++) { o[72 + k] = Bk.bl[k]; o[114 + k] = Bk.bp[k]; double a = lf_fabs(Bk.V[7 * k]); if (a > mxl) mxl = a; }
      }
    }
    currentChi = p_ordered_sum(cv, n, 0.0);
    __syncthreads();
    double accH = 0;   // Hpp | bp: accumulator lane a (< 42) walks the matches in order and keeps entry a
    if (lane < 42) accH = p_walk<false>(pc.wsB + 78 + lane, 120, n, accH);
#pragma unroll
    for (int a = 0; a < 36; a++) Hpp[a] = p_rl64(accH, a);
#pragma unroll
    for (int a = 0; a < 6; a++) bp[a] = p_rl64(accH, 36 + a);
    if (it == 0) {   // computeLambdaInit: tau * max |diagonal entry|
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
      for (int h = 0; h < NSLOT; h++) {
        int i = lane + 64 * h;
        if (i < n) {
          lf_line_blocks Bk;
          const double *o = pc.wsB + (size_t)i * 120;
          for (int k = 0; k < 36; k++) { Bk.V[k] = o[k]; Bk.W[k] = o[36 + k]; }
          for (int k = 0; k < 6; k++) Bk.bl[k] = o[72 + k];
          double Vi[36], T[36], u[6];
          if (!lf_match_eliminate(&Bk, lambda, Vi, T, u)) okl = false;
          double *vo = pc.wsVi + (size_t)i * 36, *to = pc.wsTU + (size_t)i * 42;
          for (int k = 0; k < 36; k++) { vo[k] = Vi[k]; to[k] = T[k]; }
          for (int k = 0; k < 6; k++) to[36 + k] = u[k];
        }
      }
      // the oracle stops eliminating at the first failing match; any failure rejects the step
      int ok2 = (__ballot(!okl) == 0) ? 1 : 0;
      __syncthreads();
      {
        double acc = accH;
        if (lane < 36 && lane % 7 == 0) acc = accH + lambda;          // S = Hpp + lambda I ; g = bp
        if (lane < 42) acc = p_walk<true>(pc.wsTU + lane, 42, n, acc);
#pragma unroll
        for (int a = 0; a < 36; a++) S[a] = p_rl64(acc, a);
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = p_rl64(acc, 36 + a);
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
        double sk[NSLOT], tc[NSLOT];
        for (int h = 0; h < NSLOT; h++) {
          int i = lane + 64 * h;
          sk[h] = 0; tc[h] = 0;
          if (i < n) {
            lf_line_blocks Bk;
            const double *o = pc.wsB + (size_t)i * 120;
            for (int k = 0; k < 36; k++) Bk.W[k] = o[36 + k];
            for (int k = 0; k < 6; k++) Bk.bl[k] = o[72 + k];
            double Vi[36], dl[6], Ln[6], s = 0;
            for (int k = 0; k < 36; k++) Vi[k] = pc.wsVi[(size_t)i * 36 + k];
            lf_match_backsub(&Bk, Vi, dp, dl);
            for (int k = 0; k < 6; k++) { Ln[k] = pc.wsL[6 * i + k] + dl[k]; pc.wsLn[6 * i + k] = Ln[k]; s += dl[k] * (lambda * dl[k] + Bk.bl[k]); }
            sk[h] = s;
            lf_line_meas m;
            p_meas(pc, set[i], &m);
            tc[h] = lf_match_chi2(&Xn, Ln, &m, wgt, hd, hub);
          }
        }
        scale = p_ordered_sum(sk, n, scale);
        tempChi = p_ordered_sum(tc, n, 0.0);
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
        for (int h = 0; h < NSLOT; h++) {
          int i = lane + 64 * h;
          if (i < n) for (int k = 0; k < 6; k++) pc.wsL[6 * i + k] = pc.wsLn[6 * i + k];
        }
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

// inlier scan of all matches with tf; returns count, fills set[] (ascending) and the float sse the
// reference accumulates (motion.cpp:688-699 / 795-812)
__device__ int p_score(const PoseCtx &pc, int nLn, const float *tf, double thr, int *set, float *sse_out,
                       double *sse_d_out) {
  const int lane = p_lane();
  double add[NSLOT];
  u64 msk[NSLOT];
  int cnt = 0;
#pragma unroll
  for (int h = 0; h < NSLOT; h++) {
    int i = lane + 64 * h;
    bool in = false;
    add[h] = 0;
    if (i < nLn) {
      const lf_line_record *q = &pc.query[pc.mq[i]], *t = &pc.train[pc.mt[i]];
      in = lf_line_inlier(tf, q->A, q->B, t->A, t->B, t->DUa, t->DUb, thr, &add[h]);
    }
    msk[h] = __ballot(in);
    if (in) set[cnt + __popcll(msk[h] & p_lt())] = i;
    cnt += __popcll(msk[h]);
  }
  float sse = 0;      // `float sse` of the RANSAC loop (motion.cpp:666)
  double sse_d = 0;   // `double tmp_sse` of the re-scoring loop (motion.cpp:778)
#pragma unroll
  for (int h = 0; h < NSLOT; h++) {
    u64 m = msk[h];
    while (m) { int l = __builtin_ctzll(m); m &= m - 1; double a = p_rl64(add[h], l); sse += a; sse_d += a; }
  }
  *sse_out = sse;
  *sse_d_out = sse_d;
  __syncthreads();
  return cnt;
}

__global__ void __launch_bounds__(64) k_pose(PairConsts c, PairBuffers b) {
  __shared__ PoseShared S;
  const int pr = blockIdx.x, lane = p_lane();
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  lf_pair_result *res = b.results + pr;
  PoseCtx pc;
  pc.train = b.recs_t + (size_t)ft * b.line_cap_t;
  pc.query = b.recs + (size_t)fq * c.line_cap;
  pc.mq = b.match_q + (size_t)pr * c.match_cap;
  pc.mt = b.match_t + (size_t)pr * c.match_cap;
  double *ws = b.ws + (size_t)pr * LF_PAIR_WS_DOUBLES;
  pc.wsB = ws; pc.wsVi = ws + LF_MAX_MATCHES * 120; pc.wsTU = pc.wsVi + LF_MAX_MATCHES * 36;
  pc.wsL = pc.wsTU + LF_MAX_MATCHES * 42; pc.wsLn = pc.wsL + LF_MAX_MATCHES * 6;
  pc.P = c.P;
  const lf_params &P = c.P;
  int nLn = b.nmatches[pr];
  const int n_all = nLn;
  if (nLn > c.match_cap) nLn = c.match_cap;
  if (nLn > LF_MAX_MATCHES) nLn = LF_MAX_MATCHES;
  const long long id_t = (long long)b.frame_ids_t[ft], id_q = (long long)b.frame_ids[fq];
  float tf_out[16];
#pragma unroll
  for (int i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  float rmse_out = 1e9f;
  int valid = 0, n_inl = 0, best_iter = -1, rounds = 0;
  int min_inlier = P.min_feature_matches, lw = P.line_match_number_weight, maxIter = P.ransac_iters_line_motion;
  if (maxIter > LF_RANSAC_MAX_ITERS) maxIter = LF_RANSAC_MAX_ITERS;
  const double thr = P.max_mah_dist_for_inliers;
  bool go = !(0 + nLn * lw < min_inlier);                                                   // motion.cpp:621-624
  if (min_inlier > 0.7 * (0 + nLn * lw)) min_inlier = (int)(0.7 * (0 + nLn * lw));          // :626-628
  { long long d = id_t - id_q; if (d < 0) d = -d; if (d > 50) min_inlier = P.min_matches_loopclose; }   // :631-633
  if (nLn < 3) go = false;
  if (go) {
    // ---- sample sequence (serial; partial Fisher-Yates state carries over, :635-658)
    for (int i = lane; i < nLn; i += 64) S.idx[i] = i;
    __syncthreads();
    if (lane == 0) {
      const uint64_t stream = LF_STREAM_PAIR((uint64_t)id_q, (uint64_t)id_t);
      uint64_t ctr = 0;
      for (int it = 0; it < maxIter; it++) {
        int bpos = 0, left = nLn;
        for (int s = 0; s < 3; s++) {
          int r = bpos + (int)(lf_rand31(P.rng_seed, stream, ctr++) % (uint32_t)left);
          int t = S.idx[bpos]; S.idx[bpos] = S.idx[r]; S.idx[r] = t;
          ++bpos; --left;
        }
        S.smp[3 * it] = (unsigned char)S.idx[0]; S.smp[3 * it + 1] = (unsigned char)S.idx[1]; S.smp[3 * it + 2] = (unsigned char)S.idx[2];
      }
    }
    __syncthreads();
    // ---- one hypothesis per lane
    int my_cnt = -1, my_it = 1 << 30;
    for (int it = lane; it < maxIter; it += 64) {
      double la[18], lb[18], R[9], t[3];
      float tf[16];
      for (int s = 0; s < 3; s++) {
        int k = S.smp[3 * it + s];
        const lf_line_record *q = &pc.query[pc.mq[k]], *tr = &pc.train[pc.mt[k]];
        for (int cc = 0; cc < 3; cc++) { la[6 * s + cc] = q->A[cc]; la[6 * s + 3 + cc] = q->B[cc]; lb[6 * s + cc] = tr->A[cc]; lb[6 * s + 3 + cc] = tr->B[cc]; }
      }
      if (!lf_rel_motion_lines(la, lb, 3, R, t)) continue;
      for (int i = 0; i < 3; i++) { for (int cc = 0; cc < 3; cc++) tf[4 * i + cc] = (float)R[3 * i + cc]; tf[4 * i + 3] = (float)t[i]; }
      tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
      int nc = 0;
      for (int i = 0; i < nLn; ++i) {
        double add;
        const lf_line_record *q = &pc.query[pc.mq[i]], *tr = &pc.train[pc.mt[i]];
        nc += lf_line_inlier(tf, q->A, q->B, tr->A, tr->B, tr->DUa, tr->DUb, thr, &add);
      }
      if (nc > my_cnt) { my_cnt = nc; my_it = it; }   // strictly greater: earliest iteration wins inside a lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {   // wavefront arg-max: count desc, iteration asc
      int oc = __shfl_xor(my_cnt, o, 64), oi = __shfl_xor(my_it, o, 64);
      if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
    }
    int nbest = my_cnt > 0 ? my_cnt : 0;
    best_iter = (my_cnt > 0) ? my_it : -1;
    if (0 + nbest >= 3) {                                                                    // :725-728
      // recompute the winning model (uniform) and its inlier list / sse
      double la[18], lb[18], R[9], t[3];
      float tf_best[16], sse_best = 0;
      for (int s = 0; s < 3; s++) {
        int k = S.smp[3 * best_iter + s];
        const lf_line_record *q = &pc.query[pc.mq[k]], *tr = &pc.train[pc.mt[k]];
        for (int cc = 0; cc < 3; cc++) { la[6 * s + cc] = q->A[cc]; la[6 * s + 3 + cc] = q->B[cc]; lb[6 * s + cc] = tr->A[cc]; lb[6 * s + 3 + cc] = tr->B[cc]; }
      }
      lf_rel_motion_lines(la, lb, 3, R, t);
      for (int i = 0; i < 3; i++) { for (int cc = 0; cc < 3; cc++) tf_best[4 * i + cc] = (float)R[3 * i + cc]; tf_best[4 * i + 3] = (float)t[i]; }
      tf_best[12] = tf_best[13] = tf_best[14] = 0.0f; tf_best[15] = 1.0f;
      double sse_unused;
      int nb = p_score(pc, nLn, tf_best, thr, S.set, &sse_best, &sse_unused);
      float refined_tf[16];
#pragma unroll
      for (int i = 0; i < 16; i++) refined_tf[i] = tf_best[i];
      p_refine(pc, S.set, nb, refined_tf, 25);                                               // :730
      double refined_rmse = lf_sqrt(sse_best / (0 + nb));                                    // :731
      int nref = 0;
      for (int iter = 0; iter < 20; ++iter) {                                                // :775-839
        float tmp_sse_f;
        double tmp_sse;
        int *inl = b.inliers + (size_t)pr * LF_MAX_MATCHES;
        // score into a scratch list first (kept only if it improves)
        __syncthreads();
        int ncur = p_score(pc, nLn, refined_tf, thr, S.idx, &tmp_sse_f, &tmp_sse);
        if (0 + ncur * lw > 0 + nref * lw) {
          for (int i = lane; i < ncur; i += 64) { S.set[i] = S.idx[i]; inl[i] = S.idx[i]; }
          __syncthreads();
          nref = ncur;
          refined_rmse = lf_sqrt(tmp_sse / (0 + ncur));
          p_refine(pc, S.set, nref, refined_tf, 20);
          rounds++;
        } else break;
      }
      n_inl = nref;
      rmse_out = (float)refined_rmse;
#pragma unroll
      for (int i = 0; i < 16; i++) tf_out[i] = refined_tf[i];
      valid = ((0 + lw * nref) >= min_inlier) ? 1 : 0;
    }
  }
  if (lane == 0) {
    for (int i = 0; i < 16; i++) res->T[i] = tf_out[i];
    res->rmse = rmse_out;
    res->valid = valid;
    res->n_matches = n_all;
    res->n_inliers = n_inl;
    res->n_point_matches = 0;
    res->n_point_inliers = 0;
    res->id_older = valid ? (int)id_t : -1;        // node.cpp:1606-1607
    res->id_newer = valid ? (int)id_q : -1;
    res->ransac_best_iter = best_iter;
    res->refine_rounds = rounds;
    float r2 = rmse_out * rmse_out;                                   // float arithmetic as node.cpp:1533-1534
    res->information_scale = valid ? (double)((float)(0 + n_inl * lw) / r2) : 0.0;
  }
}

void lf_pair_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t st, int solver) {
  hipLaunchKernelGGL(k_match, dim3(n_pairs), dim3(256), 0, st, c, b);
  if (solver == LF_SOLVER_HYBRID) lf_pair_hybrid_launch(c, b, n_pairs, st);
  else if (solver == LF_SOLVER_RELMOTION) lf_pair_relmotion_launch(c, b, n_pairs, st);
  else hipLaunchKernelGGL(k_pose, dim3(n_pairs), dim3(64), 0, st, c, b);
}
