// lf_pose_wg.h -- the workgroup-per-pair building blocks of the pose kernels (k_pose, k_pose_hybrid): ordered sums
// from LDS-published values, ordered walks, and the LM refinement of the LINE landmarks on lane-group tasks (six
// lanes per match).  Included by lf_pair.hip and lf_pair_hybrid.hip after lf_pair.h / lf_pose.h.
#pragma once
typedef unsigned long long u64;
__device__ __forceinline__ int p_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 p_lt() { return (1ull << p_lane()) - 1ull; }
// ONE 256-THREAD WORKGROUP (four wavefronts) PER PAIR.  A match ("landmark" of the refinement graph) belongs to
// thread i; sums over the matches are taken in list order from values published in LDS, so every thread holds
// the same bits the sequential oracle computes.
#define PT_N 256                   // threads per pair (>= LF_MAX_MATCHES)
#define PW_N (PT_N / 64)
#define LM_RED_N (LF_MAX_PT_MATCHES + LF_MAX_MATCHES)   // landmarks of one refinement: point matches, then line matches
struct LmShared {                  // LDS state of the LM refinement (per workgroup)
  double red[2][LM_RED_N + 8];    // per-landmark values of the ordered sums (zero padded to a multiple of 8)
  double hb[42], sg[42];          // Hpp | bp of the current linearisation; S | g of the current damping
  double tile[PW_N][10 * 108];    // per wavefront: the Jacobian columns (or W Vi) of the ten matches of a pass
  double wred[PW_N];
  lf_se3 xp[12];                  // X (+) (+-delta e_d): the perturbed poses of the current linearisation
  int wcnt[PW_N], wit[PW_N];
};
struct PoseCtx {
  const lf_line_record *train, *query;
  const int *mq, *mt;
  double *wsB, *wsVi, *wsTU, *wsL, *wsLn, *wsE;
  lf_params P;
};

__device__ __forceinline__ void p_meas(const PoseCtx &pc, int k, lf_line_meas *m) {
  const lf_line_record *q = &pc.query[pc.mq[k]], *t = &pc.train[pc.mt[k]];
  m->nA = q->A; m->nB = q->B; m->nMa = q->DUa; m->nMb = q->DUb;
  m->oA = t->A; m->oB = t->B; m->oMa = t->DUa; m->oMb = t->DUb;
}
// s + v_0 + v_1 + ... + v_{n-1} strictly in that order, v_i = the value thread i passes (threads >= n pass anything).
// Eight LDS operands are in flight per trip; the padding rows are zero, and x + 0.0 == x.
__device__ __forceinline__ void p_publish(double *red, double v, int n) {
  const int tid = threadIdx.x;
  if (tid < LF_MAX_MATCHES) red[tid] = (tid < n) ? v : 0.0;
  if (tid < 8) red[LF_MAX_MATCHES + tid] = 0.0;   // (callers with n <= LF_MAX_MATCHES and one value per thread)
}
__device__ __forceinline__ double p_sum_published(const double *red, int n, double s) {
  const int n8 = (n + 7) & ~7;
  for (int l = 0; l < n8; l += 8) {
    double q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = red[l + k];
#pragma unroll
    for (int k = 0; k < 8; k++) s += q[k];
  }
  return s;
}
// two ordered sums at once (each in match order, as p_sum_published): the two chains of dependent additions share the issue
// slots, and the next eight terms of both are on their way from LDS while the current eight are added -- at one wavefront per
// SIMD a lone chain pays the LDS latency and the fp64 latency of every term (red[] is zero padded past n8: the read-ahead stays
// inside the array)
__device__ __forceinline__ void p_sum2_published(const double *ra, const double *rb, int n, double *sa_io, double *sb_io) {
  const int n8 = (n + 7) & ~7;
  double sa = *sa_io, sb = *sb_io, qa[8], qb[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { qa[k] = ra[k]; qb[k] = rb[k]; }
  for (int l = 0; l < n8; l += 8) {
    double na[8], nb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { na[k] = ra[l + 8 + k]; nb[k] = rb[l + 8 + k]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { sa += qa[k]; sb += qb[k]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { qa[k] = na[k]; qb[k] = nb[k]; }
  }
  *sa_io = sa; *sb_io = sb;
}
// maximum over the workgroup (order does not matter for a maximum)
__device__ __forceinline__ double p_block_max(LmShared &S, double mx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
  if (p_lane() == 0) S.wred[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = S.wred[0];
#pragma unroll
  for (int w = 1; w < PW_N; w++) { double t = S.wred[w]; mx = t > mx ? t : mx; }
  __syncthreads();
  return mx;
}

// acc (+/-)= base[k * stride] for k = 0..n-1, strictly in that order; the loads of 8 rows are issued together
// (they do not depend on the running sum), the additions stay sequential.
template <bool SUB>
__device__ __forceinline__ double p_walk(const double *base, size_t stride, int n, double acc) {
  int k = 0;
  for (; k + 32 <= n; k += 32) {     // (the rows come from HBM / L2 when the chip is full: keep many loads in flight)
    double v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) v[j] = base[(size_t)(k + j) * stride];
#pragma unroll
    for (int j = 0; j < 32; j++) acc = SUB ? acc - v[j] : acc + v[j];
  }
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

#ifdef LF_POSE_PROFILE   // LF_EXTRA_CFLAGS=-DLF_POSE_PROFILE: s_memtime per phase of pair 7, printed at the end of k_pose
__device__ unsigned long long g_pprof[16], g_pprev;
#define PT(k) do { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 7 && threadIdx.x == 0) { unsigned long long tn = __builtin_amdgcn_s_memtime(); g_pprof[k] += tn - g_pprev; g_pprev = tn; } } while (0)
#else
#define PT(k) do { } while (0)
#endif
// ---- the LM refinement works on TASKS (match i, component d): six neighbouring lanes own one match, ten matches per
// wavefront, forty per pass of the workgroup.  The six lanes read the same workspace rows (one cache line serves
// all of them, and a wavefront's working set stays inside the L1), exchange what they need through a per-wavefront
// LDS tile, and each produces one column / row of the 6x6 blocks.  Every value is produced by the same expression
// as in the sequential lf_match_blocks / lf_match_eliminate / lf_match_backsub (lf_pose.h).
#define PG_N 10                    // matches per wavefront pass
struct PoseTask { int i, d, g; bool act; };
__device__ __forceinline__ PoseTask p_task(int base, int n) {
  PoseTask t;
  const int lane = p_lane();
  t.g = lane / 6; t.d = lane - 6 * t.g;
  if (t.g >= PG_N) { t.g = PG_N - 1; t.d = 0; }
  t.i = base + (int)(threadIdx.x >> 6) * PG_N + t.g;
  t.act = lane < 6 * PG_N && t.i < n;
  return t;
}
__device__ __forceinline__ void p_wave_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// value of component k of the calling lane's match (lanes 6 g .. 6 g + 5 hold components 0 .. 5)
__device__ __forceinline__ double p_sib(double v, const PoseTask &t, int k) { return __shfl(v, 6 * t.g + k, 64); }

// thread i < n: the edge errors of match i at (X, L = Lsrc + 6 i), their robust weights (times the edge weight) into
// slot `slot` of the match's error record (en[6] eo[6] wn wo), and the match's chi2 term (lf_match_chi2) into red[i].
// The record of an accepted trial step IS the base record of the next linearisation (same X, same L, same code).
#define PE_STRIDE 28
__device__ void p_errchi(const PoseCtx &pc, const int *set, int n, const lf_se3 &X, const double *Lsrc, int slot,
                         double wgt, double hd, int hub, double *red) {
  const int i = threadIdx.x;
  if (i < n) {
    lf_line_meas m;
    double L[6], en[6], eo[6], c, r0n, r0o, wn, wo;
    p_meas(pc, set[i], &m);
    for (int k = 0; k < 6; k++) L[k] = Lsrc[6 * i + k];
    lf_match_errors(&X, L, &m, en, eo);
    c = 0; for (int k = 0; k < 6; k++) c += en[k] * (wgt * en[k]);
    lf_huber(c, hd, hub, &r0n, &wn);
    c = 0; for (int k = 0; k < 6; k++) c += eo[k] * (wgt * eo[k]);
    lf_huber(c, hd, hub, &r0o, &wo);
    double *E = pc.wsE + (size_t)i * PE_STRIDE + 14 * slot;
    for (int k = 0; k < 6; k++) { E[k] = en[k]; E[6 + k] = eo[k]; }
    E[12] = wn * wgt; E[13] = wo * wgt;
    red[i] = r0n + r0o;
  }
}

// the twelve perturbed poses X (+) (+-1e-9 e_d) of the numeric pose Jacobians are the same for every match: twelve
// threads compute them once per linearisation (lf_se3_oplus, as lf_match_blocks does per match) into S.xp[2 d + sign]
__device__ __forceinline__ void p_perturbed_poses(LmShared &S, const lf_se3 &X) {
  const int tid = threadIdx.x;
  if (tid < 12) {
    const int d = tid >> 1;
    const double dl = (tid & 1) ? -1e-9 : 1e-9;
    double v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = (k == d) ? dl : 0.0;
    lf_se3 Xp;
    lf_se3_oplus(&X, v, &Xp);
    S.xp[tid] = Xp;
  }
}
// lf_match_blocks: lane (i, d) computes column d of Jn, Jo, Jp (central differences along landmark component d and
// pose component d), publishes them in the tile, and then row d of V, W, Hpp and entry d of bl, bp.
__device__ void p_blocks(LmShared &S, const PoseCtx &pc, const int *set, int n, const lf_se3 &X, int slot, double *mxl_io) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  double mxl = *mxl_io;
  double *tile = S.tile[threadIdx.x >> 6];
  for (int base = 0; base < n; base += PG_N * PW_N) {
    const PoseTask t = p_task(base, n);
    const int i = t.act ? t.i : 0, d = t.d;
    lf_line_meas m;
    double L[6], en[6], eo[6], cn[6], co[6], cp[6];
    p_meas(pc, set[i], &m);
    for (int k = 0; k < 6; k++) L[k] = pc.wsL[6 * i + k];
    const double *E = pc.wsE + (size_t)i * PE_STRIDE + 14 * slot;    // p_errchi: errors and weights at (X, L)
    for (int k = 0; k < 6; k++) { en[k] = E[k]; eo[k] = E[6 + k]; }
    const double wn = E[12], wo = E[13];
    {
      double Lp[6], ep[6], em[6], ep2[6], em2[6];
      for (int k = 0; k < 6; k++) Lp[k] = (k == d) ? L[k] + delta : L[k];
      lf_match_errors(&X, Lp, &m, ep, ep2);
      for (int k = 0; k < 6; k++) Lp[k] = (k == d) ? L[k] - delta : L[k];
      lf_match_errors(&X, Lp, &m, em, em2);
      for (int k = 0; k < 6; k++) { cn[k] = scalar * (ep[k] - em[k]); co[k] = scalar * (ep2[k] - em2[k]); }
    }
    {
      double PA[3], PB[3], ep[6], em[6];
      lf_se3 Xp = S.xp[2 * d];                 // X (+) (+delta e_d), p_perturbed_poses
      lf_se3_inv_apply(&Xp, L, PA); lf_se3_inv_apply(&Xp, L + 3, PB);
      lf_line_edge_error(m.oMa, m.oMb, m.oA, m.oB, PA, PB, ep);
      Xp = S.xp[2 * d + 1];                    // X (+) (-delta e_d)
      lf_se3_inv_apply(&Xp, L, PA); lf_se3_inv_apply(&Xp, L + 3, PB);
      lf_line_edge_error(m.oMa, m.oMb, m.oA, m.oB, PA, PB, em);
      for (int k = 0; k < 6; k++) cp[k] = scalar * (ep[k] - em[k]);
    }
    double *J = tile + t.g * 108;          // Jn[36] Jo[36] Jp[36], row-major 6x6 each
    if (t.act) for (int k = 0; k < 6; k++) { J[6 * k + d] = cn[k]; J[36 + 6 * k + d] = co[k]; J[72 + 6 * k + d] = cp[k]; }
    p_wave_order();
    if (t.act) {
      const int q = d;
      double *o = pc.wsB + (size_t)i * 120;
      double sbl_n = 0, sbl_o = 0, sbp = 0;
      for (int k = 0; k < 6; k++) {
        double wen = wn * en[k], weo = wo * eo[k];
        sbl_n += cn[k] * wen; sbl_o += co[k] * weo; sbp += cp[k] * weo;
      }
      o[72 + q] = -(sbl_n + sbl_o);
      o[114 + q] = -sbp;
      for (int j = 0; j < 6; j++) {
        double vn = 0, vo = 0, hw = 0, hp = 0;
        for (int k = 0; k < 6; k++) {
          double jn = J[6 * k + j], jo = J[36 + 6 * k + j], jp = J[72 + 6 * k + j];
          vn += cn[k] * (wn * jn);
          vo += co[k] * (wo * jo);
          hw += cp[k] * (wo * jo);
          hp += cp[k] * (wo * jp);
        }
        double V = vn + vo;
        o[6 * q + j] = V;
        o[36 + 6 * q + j] = hw;
        o[78 + 6 * q + j] = hp;
        if (j == q) { double a = lf_fabs(V); if (a > mxl) mxl = a; }
      }
    }
    p_wave_order();                         // the tile is free for the next pass
  }
  *mxl_io = mxl;
}

// lf_match_eliminate: lane (i, d) solves (V + lambda I) x = e_d (column d of Vi; the elimination of the matrix is
// the same in the six lanes and the right-hand-side columns are independent), forms column d of W Vi, publishes it,
// and then row d of T = W Vi W^T and entry d of u = W Vi bl.  Returns 1 if some match of this lane is singular.
__device__ int p_eliminate(LmShared &S, const PoseCtx &pc, int n, double lambda) {
  int bad = 0;
  double *tile = S.tile[threadIdx.x >> 6];
  for (int base = 0; base < n; base += PG_N * PW_N) {
    const PoseTask t = p_task(base, n);
    const int i = t.act ? t.i : 0, d = t.d;
    const double *o = pc.wsB + (size_t)i * 120;
    double A[36], x[6], W[36], wvc[6];
#pragma unroll
    for (int k = 0; k < 36; k++) A[k] = o[k];
#pragma unroll
    for (int k = 0; k < 6; k++) { A[7 * k] += lambda; x[k] = (k == d) ? 1.0 : 0.0; }
    const int ok = lf_solve6(A, x, 1);
    if (t.act && !ok) bad = 1;
#pragma unroll
    for (int k = 0; k < 36; k++) W[k] = o[36 + k];
#pragma unroll
    for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 6; k++) s += W[6 * r + k] * x[k]; wvc[r] = s; }
    double *WV = tile + t.g * 108;
    if (t.act) {
      double *vo = pc.wsVi + (size_t)i * 36;
#pragma unroll
      for (int k = 0; k < 6; k++) { vo[6 * k + d] = x[k]; WV[6 * k + d] = wvc[k]; }
    }
    p_wave_order();
    if (t.act) {
      double *to = pc.wsTU + (size_t)i * 42;
      double wv[6], s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) wv[k] = WV[6 * d + k];
#pragma unroll
      for (int k = 0; k < 6; k++) s += wv[k] * o[72 + k];
      to[36 + d] = s;
#pragma unroll
      for (int j = 0; j < 6; j++) { double s2 = 0; for (int k = 0; k < 6; k++) s2 += wv[k] * W[6 * j + k]; to[6 * d + j] = s2; }
    }
    p_wave_order();
  }
  return bad;
}

// lf_match_backsub + the step's chi2: lane (i, a) computes component a of r = bl - W^T dp and of dl = Vi r; the six
// lanes exchange r, the new landmark and the scale terms by lane shuffles.
__device__ void p_backsub(const PoseCtx &pc, int n, const double *dp, double lambda, double *red) {
  for (int base = 0; base < n; base += PG_N * PW_N) {
    const PoseTask t = p_task(base, n);
    const int i = t.act ? t.i : 0, a = t.d;
    const double *o = pc.wsB + (size_t)i * 120, *Vi = pc.wsVi + (size_t)i * 36;
    double tt = 0, rr[6], dl = 0, s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) tt += o[36 + 6 * k + a] * dp[k];
    const double bla = o[72 + a], ra = bla - tt;
#pragma unroll
    for (int k = 0; k < 6; k++) rr[k] = p_sib(ra, t, k);
#pragma unroll
    for (int k = 0; k < 6; k++) dl += Vi[6 * a + k] * rr[k];
    const double La = pc.wsL[6 * i + a] + dl, term = dl * (lambda * dl + bla);
    if (t.act) pc.wsLn[6 * i + a] = La;
#pragma unroll
    for (int k = 0; k < 6; k++) s += p_sib(term, t, k);
    if (t.act && a == 0) red[i] = s;
  }
}
// rows n .. n8-1 of both published columns hold 0.0 (the ordered sums run in trips of eight)
__device__ __forceinline__ void p_pad_published(LmShared &S, int n) {
  const int tid = threadIdx.x;
  if (tid < 8) { S.red[0][n + tid] = 0.0; S.red[1][n + tid] = 0.0; }
}

