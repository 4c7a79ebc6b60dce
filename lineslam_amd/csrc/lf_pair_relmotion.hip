// lf_pair_relmotion.hip -- computeRelativeMotion_Ransac (src/line/motion.cpp:367-526): the lines-only solver
// (SURVEY.md 8a row a24), one wavefront per node pair, on the line matches k_match produced.
//   3-line samples with the 5-degree degeneracy test (:424-437) -> computeRelativeMotion_svd (:315-365)
//   consensus: mean end-point distance < pt2line3d_dist_relmotion and direction angle < line3d_angle_relmotion
//   optimizeRelmotion (:98-139): dlevmar_dif (external/levmar-2.6/lm_core.c:438-846) on (quaternion, t), residual
//   = 1/4 of the four Mahalanobis end-point-to-line distances (costFun_optimizeRelmotion :60-96), re-run while
//   the consensus set grows (:481-523).
// Mapping: the sample sequence is generated serially (it carries the shuffle state), every lane then evaluates
// its own hypotheses; the LM runs with one residual row per lane (4 slots), J^T J / J^T e with one accumulator
// per lane walking the rows in levmar's order -- every sum in the order of the sequential code (oracle twin:
// oracle_relmotion_ransac).
#include "lf_pair.h"
#include "lf_pose.h"
#include <float.h>

typedef unsigned long long u64;
#define RM_ROWS LF_MAX_MATCHES
#define RM_M 7

__device__ __forceinline__ int r_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 r_lt() { return (1ull << r_lane()) - 1ull; }
__device__ __forceinline__ double r_rl64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

struct RShared {
  int idx[RM_ROWS];
  unsigned char smp[LF_RANSAC_MAX_ITERS * 3];
  int set[RM_ROWS], cur[RM_ROWS];
  double jac[RM_ROWS * RM_M];
  double hx[RM_ROWS], e[RM_ROWS], wrk[RM_ROWS], wrk2[RM_ROWS];
};
struct RCtx {
  const lf_line_record *a, *b;   // query (newer) / train (older) records
  const int *mq, *mt;
  double distThresh, angThresh;
};

// costFun_optimizeRelmotion over the rows list[0..n)
__device__ void r_cost(const RCtx &pc, const int *list, int n, const double *p, double *out) {
  double R[9];
  lf_q2r(p, R);
  for (int i = r_lane(); i < n; i += 64) {
    const lf_line_record *a = &pc.a[pc.mq[list[i]]], *b = &pc.b[pc.mt[list[i]]];
    out[i] = lf_relmotion_residual(R, p + 4, a->A, a->B, a->DUa, a->DUb, b->A, b->B, b->DUa, b->DUb);
  }
  __syncthreads();
}

// dlevmar_dif for m = 7, x = 0; returns the iteration count
__device__ int r_levmar7(RShared &S, const RCtx &pc, const int *list, int n, double *p, int itmax) {
  const int m = RM_M, lane = r_lane();
  const double tau = 1E-03, eps1 = 1E-10, eps2 = 1E-20, eps2_sq = 1E-20 * 1E-20, eps3 = 1E-20, delta = 1E-06;
  double jacTe[7], jacTjac[49], Dp[7], diag[7], pDp[7];
  double mu = 0, tmp, p_eL2, jacTe_inf = 0, pDp_eL2, p_L2 = 0, Dp_L2 = DBL_MAX, dF, dL;
  int nu, nu2, stop = 0, K = 10, updjac = 0, updp = 1, newjac = 0, k;
  int ai = 0, aj = 0;   // accumulator ownership: lanes 0..27 lower triangle (i,j), lanes 28..34 J^T e
  if (lane < 28) { int a = lane; while ((ai + 1) * (ai + 2) / 2 <= a) ai++; aj = a - ai * (ai + 1) / 2; }
  else if (lane < 35) { ai = lane - 28; aj = -1; }
  r_cost(pc, list, n, p, S.hx);
  for (int i = lane; i < n; i += 64) S.e[i] = 0.0 - S.hx[i];
  __syncthreads();
  p_eL2 = lf_l2nrm_sq(S.e, n);
  if (!(lf_fabs(p_eL2) <= DBL_MAX)) stop = 7;
  nu = 20;
  for (k = 0; k < itmax && !stop; ++k) {
    if (p_eL2 <= eps3) { stop = 6; break; }
    if ((updp && nu > 16) || updjac == K) {
      for (int j = 0; j < m; ++j) {           // forward differences (misc_core.c:137-171)
        double d = 1E-04 * p[j], t;
        d = lf_fabs(d);
        if (d < delta) d = delta;
        t = p[j]; p[j] += d;
        r_cost(pc, list, n, p, S.wrk);
        p[j] = t;
        d = 1.0 / d;
        for (int i = lane; i < n; i += 64) S.jac[i * m + j] = (S.wrk[i] - S.hx[i]) * d;
      }
      __syncthreads();
      nu = 2; updjac = 0; updp = 0; newjac = 1;
    }
    if (newjac) {
      newjac = 0;
      double acc = 0.0;
      if (lane < 35)
        for (int l = n; l-- > 0;) {
          double alpha = S.jac[l * m + ai];
          acc += (aj >= 0) ? S.jac[l * m + aj] * alpha : alpha * S.e[l];
        }
#pragma unroll
      for (int i = 0; i < 7; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) { double v = r_rl64(acc, i * (i + 1) / 2 + j); jacTjac[i * m + j] = v; jacTjac[j * m + i] = v; }
#pragma unroll
      for (int i = 0; i < 7; i++) jacTe[i] = r_rl64(acc, 28 + i);
      p_L2 = jacTe_inf = 0.0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (jacTe_inf < (tmp = lf_fabs(jacTe[i]))) jacTe_inf = tmp;
        diag[i] = jacTjac[i * m + i];
        p_L2 += p[i] * p[i];
      }
    }
    if (jacTe_inf <= eps1) { Dp_L2 = 0.0; stop = 1; break; }
    if (k == 0) {
      tmp = DBL_MIN;
#pragma unroll
      for (int i = 0; i < 7; ++i) if (diag[i] > tmp) tmp = diag[i];
      mu = tau * tmp;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) jacTjac[i * m + i] += mu;
    int issolved;
    {
      double A[49], Bv[7];
#pragma unroll
      for (int i = 0; i < 49; i++) A[i] = jacTjac[i];
#pragma unroll
      for (int i = 0; i < 7; i++) Bv[i] = jacTe[i];
      issolved = lf_lu7_u(A, Bv);   // AX_EQ_B_LU in LAPACK order (lf_linalg.h)
#pragma unroll
      for (int i = 0; i < 7; i++) Dp[i] = Bv[i];
    }
    if (issolved) {
      Dp_L2 = 0.0;
#pragma unroll
      for (int i = 0; i < 7; ++i) { pDp[i] = p[i] + (tmp = Dp[i]); Dp_L2 += tmp * tmp; }
      if (Dp_L2 <= eps2_sq * p_L2) { stop = 2; break; }
      if (Dp_L2 >= (p_L2 + eps2) / (1E-12 * 1E-12)) { stop = 4; break; }
      r_cost(pc, list, n, pDp, S.wrk);
      for (int i = lane; i < n; i += 64) S.wrk2[i] = 0.0 - S.wrk[i];
      __syncthreads();
      pDp_eL2 = lf_l2nrm_sq(S.wrk2, n);
      if (!(lf_fabs(pDp_eL2) <= DBL_MAX)) { stop = 7; break; }
      dF = p_eL2 - pDp_eL2;
      if (updp || dF > 0) {                       // Broyden rank-one update, row-parallel
        for (int i = lane; i < n; i += 64) {
          double t2 = 0.0;
#pragma unroll
          for (int l = 0; l < 7; ++l) t2 += S.jac[i * m + l] * Dp[l];
          t2 = (S.wrk[i] - S.hx[i] - t2) / Dp_L2;
#pragma unroll
          for (int j = 0; j < 7; ++j) S.jac[i * m + j] += t2 * Dp[j];
        }
        __syncthreads();
        ++updjac; newjac = 1;
      }
      dL = 0.0;
#pragma unroll
      for (int i = 0; i < 7; ++i) dL += Dp[i] * (mu * Dp[i] + jacTe[i]);
      if (dL > 0.0 && dF > 0.0) {
        tmp = (2.0 * dF / dL - 1.0);
        tmp = 1.0 - tmp * tmp * tmp;
        mu = mu * ((tmp >= 0.3333333334) ? tmp : 0.3333333334);
        nu = 2;
#pragma unroll
        for (int i = 0; i < 7; ++i) p[i] = pDp[i];
        for (int i = lane; i < n; i += 64) { S.e[i] = S.wrk2[i]; S.hx[i] = S.wrk[i]; }
        __syncthreads();
        p_eL2 = pDp_eL2;
        updp = 1;
        continue;
      }
    }
    mu *= nu;
    nu2 = nu << 1;
    if (nu2 <= nu) { stop = 5; break; }
    nu = nu2;
#pragma unroll
    for (int i = 0; i < 7; ++i) jacTjac[i * m + i] = diag[i];
  }
  return k;
}

// optimizeRelmotion (motion.cpp:98-139)
__device__ int r_optimize(RShared &S, const RCtx &pc, const int *list, int n, double *R, double *t) {
  double para[7];
  lf_r2q(R, para);
  para[4] = t[0]; para[5] = t[1]; para[6] = t[2];
  int it = r_levmar7(S, pc, list, n, para, 50);
  lf_q2r(para, R);
  t[0] = para[4]; t[1] = para[5]; t[2] = para[6];
  return it;
}

// consensus set of (R, t) over all matches, ascending match index (motion.cpp:443-455 / 499-510)
__device__ int r_consensus(const RCtx &pc, int n, const double *R, const double *t, int *set) {
  const int lane = r_lane();
  int c = 0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool in = false;
    if (i < n) {
      const lf_line_record *a = &pc.a[pc.mq[i]], *b = &pc.b[pc.mt[i]];
      in = lf_relmotion_inlier(R, t, a->A, a->B, b->A, b->B, pc.distThresh, pc.angThresh) != 0;
    }
    u64 m = __ballot(in);
    if (in) set[c + __popcll(m & r_lt())] = i;
    c += __popcll(m);
  }
  __syncthreads();
  return c;
}

__device__ __forceinline__ void r_sample_lines(const RCtx &pc, const unsigned char *smp, int it, double *la, double *lb) {
  for (int s = 0; s < 3; s++) {
    int k = smp[3 * it + s];
    const lf_line_record *a = &pc.a[pc.mq[k]], *b = &pc.b[pc.mt[k]];
    for (int c = 0; c < 3; c++) { la[6 * s + c] = a->A[c]; la[6 * s + 3 + c] = a->B[c]; lb[6 * s + c] = b->A[c]; lb[6 * s + 3 + c] = b->B[c]; }
  }
}

__global__ void __launch_bounds__(64) k_relmotion(PairConsts c, PairBuffers b) {
  __shared__ RShared S;
  const int pr = blockIdx.x, lane = r_lane();
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  lf_pair_result *res = b.results + pr;
  const lf_params &P = c.P;
  RCtx pc;
  pc.b = b.recs_t + (size_t)ft * b.line_cap_t;
  pc.a = b.recs + (size_t)fq * c.line_cap;
  pc.mq = b.match_q + (size_t)pr * c.match_cap;
  pc.mt = b.match_t + (size_t)pr * c.match_cap;
  pc.distThresh = P.pt2line3d_dist_relmotion;
  pc.angThresh = P.line3d_angle_relmotion;
  int n = b.nmatches[pr];
  const int n_all = n;
  if (n > c.match_cap) n = c.match_cap;
  if (n > RM_ROWS) n = RM_ROWS;
  const long long id_t = (long long)b.frame_ids_t[ft], id_q = (long long)b.frame_ids[fq];
  int maxIter = P.ransac_iters_line_motion;
  if (maxIter > LF_RANSAC_MAX_ITERS) maxIter = LF_RANSAC_MAX_ITERS;
  double Ro[9], to[3];
#pragma unroll
  for (int i = 0; i < 9; i++) Ro[i] = (i % 4 == 0) ? 1.0 : 0.0;
  to[0] = to[1] = to[2] = 0.0;
  int n_inl = 0, best_iter = -1, rounds = 0, lm_first = 0;
  int *inl = b.inliers + (size_t)pr * LF_MAX_MATCHES;
  if (n >= 3) {                                                                            // :371-375
    for (int i = lane; i < n; i += 64) S.idx[i] = i;
    __syncthreads();
    if (lane == 0) {   // sample sequence (the shuffle state carries over, utils.h:49-60)
      const uint64_t stream = LF_STREAM_RELMOTION((uint64_t)id_q, (uint64_t)id_t);
      uint64_t ctr = 0;
      for (int it = 0; it < maxIter; it++) {
        int bpos = 0, left = n;
        for (int s = 0; s < 3; s++) {
          int r = bpos + (int)(lf_rand31(P.rng_seed, stream, ctr++) % (uint32_t)left);
          int t = S.idx[bpos]; S.idx[bpos] = S.idx[r]; S.idx[r] = t;
          ++bpos; --left;
        }
        S.smp[3 * it] = (unsigned char)S.idx[0]; S.smp[3 * it + 1] = (unsigned char)S.idx[1]; S.smp[3 * it + 2] = (unsigned char)S.idx[2];
      }
    }
    __syncthreads();
    int my_cnt = 0, my_it = 1 << 30;
    for (int it = lane; it < maxIter; it += 64) {   // one hypothesis per lane
      double la[18], lb[18], R[9], t[3];
      r_sample_lines(pc, S.smp, it, la, lb);
      if (lf_relmotion_degenerate(la, c.cos_degeneracy)) continue;
      if (!lf_rel_motion_lines(la, lb, 3, R, t)) continue;
      int nc = 0;
      for (int i = 0; i < n; ++i) {
        const lf_line_record *a = &pc.a[pc.mq[i]], *bb = &pc.b[pc.mt[i]];
        nc += lf_relmotion_inlier(R, t, a->A, a->B, bb->A, bb->B, pc.distThresh, pc.angThresh);
      }
      if (nc > my_cnt) { my_cnt = nc; my_it = it; }   // strictly greater: the earliest iteration wins (:457)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      int oc = __shfl_xor(my_cnt, o, 64), oi = __shfl_xor(my_it, o, 64);
      if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
    }
    if (my_cnt >= 1) {
      best_iter = my_it;
      double la[18], lb[18];
      r_sample_lines(pc, S.smp, best_iter, la, lb);
      lf_rel_motion_lines(la, lb, 3, Ro, to);
      int nmax = r_consensus(pc, n, Ro, to, S.set);
      if (nmax < 4) {                                                                      // :474-475
        n_inl = nmax;
        for (int i = lane; i < nmax; i += 64) inl[i] = S.set[i];
      } else {
        lm_first = r_optimize(S, pc, S.set, nmax, Ro, to);                                  // :483
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = Ro[i];
        t[0] = to[0]; t[1] = to[1]; t[2] = to[2];
        int nprev = 0;
        for (;;) {                                                                         // :487-523
          int nc = r_consensus(pc, n, R, t, S.cur);
          if (nc <= nprev) break;
          for (int i = lane; i < nc; i += 64) { S.set[i] = S.cur[i]; inl[i] = S.cur[i]; }
          __syncthreads();
          nprev = nc;
#pragma unroll
          for (int i = 0; i < 9; i++) Ro[i] = R[i];
          to[0] = t[0]; to[1] = t[1]; to[2] = t[2];
          r_optimize(S, pc, S.set, nprev, R, t);
          rounds++;
        }
        n_inl = nprev;
      }
    }
  }
  if (lane == 0) {
    for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) res->T[4 * r + q] = (float)Ro[3 * r + q]; res->T[4 * r + 3] = (float)to[r]; }
    res->T[12] = res->T[13] = res->T[14] = 0.0f; res->T[15] = 1.0f;
    res->rmse = 0.0f;
    res->valid = n_inl > 0 ? 1 : 0;
    res->n_matches = n_all;
    res->n_inliers = n_inl;
    res->n_point_matches = 0;
    res->n_point_inliers = 0;
    res->id_older = n_inl > 0 ? (int)id_t : -1;
    res->id_newer = n_inl > 0 ? (int)id_q : -1;
    res->ransac_best_iter = best_iter;
    res->refine_rounds = rounds;
    res->information_scale = 0.0;
    res->overflow = ((n_all > c.match_cap || n_all > LF_MAX_MATCHES) ? LF_OVF_MATCHES : 0) |
                    ((b.nlines[b.pair_q[blockIdx.x]] > c.line_cap || b.nlines_t[b.pair_t[blockIdx.x]] > (b.line_cap_t < c.line_cap ? b.line_cap_t : c.line_cap)) ? LF_OVF_LINES : 0);
    res->reserved_ = 0;
    double *o = b.motion_d + (size_t)pr * LF_MOTION_STRIDE;
    for (int i = 0; i < 9; i++) o[i] = Ro[i];
    for (int i = 0; i < 3; i++) o[9 + i] = to[i];
    o[12] = (double)lm_first;   // diagnostics: levmar iterations of the first optimizeRelmotion
  }
}

void lf_pair_relmotion_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t st) {
  hipLaunchKernelGGL(k_relmotion, dim3(n_pairs), dim3(64), 0, st, c, b);
}
