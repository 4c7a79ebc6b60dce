// lf_pose_res.h -- the RESIDENT form of the LM refinement of k_pose (lines only): one 256-thread workgroup per pair and
// per CU (one wavefront per SIMD, 512 registers each), the whole state of the refinement on chip.  Per match the normal
// equations need V (36), W (36), bl (6), the landmark and its trial value (12) across the trial steps of a linearisation:
//   LDS        W | bl (42 doubles per match), both landmark sets, both poses, the published chi2 / scale terms, one
//              42-double row per match of the running pass (Jacobian columns, V gather, W Vi, Vi transposition, then the
//              Hpp | bp or T | u row that 42 accumulator lanes add up in match order);
//   registers  row d of V of the lane's match in each of the seven passes (6 lanes per match, ten matches per wavefront,
//              40 per pass), selected by the wave-uniform pass number.  Vi is NOT kept between the elimination and the
//              back-substitution: its column is solved again there (same arithmetic, same bits) -- with V and Vi both
//              resident the compiler spills the hot loops to scratch (6 GB per 1146-pair launch);
//   L2         the measurements of the pair's matches, compacted once per pair to 48 contiguous doubles per match
//              (79 KB for 205 matches; one CU's share of an XCD's L2 holds them); read-only during the refinement.
// Nothing is written to HBM between the first and the last iteration (PMC: 0.24 GB per 1146-pair launch, round 1: 59 GB).
// Every value is produced by the same expression, and every sum over the matches runs in the same order, as in the
// sequential lf_match_blocks_xp / lf_match_eliminate / lf_match_backsub (lf_pose.h) -- the oracle's bits.
// Included by lf_pair.hip after lf_pose_wg.h (lane helpers).
#pragma once
#ifndef RT_N
#define RT_N 256                      // threads per pair
#endif
#define RW_N (RT_N / 64)
#define RP_ROWS (PG_N * RW_N)         // matches per pass of the workgroup
#define RP_N ((LF_MAX_MATCHES + RP_ROWS - 1) / RP_ROWS)   // passes that cover LF_MAX_MATCHES
#define R_ROW 42
#ifndef R_RED_N
#define R_RED_N (LF_MAX_MATCHES + 8)      // k_pose_hybrid: point landmarks in front of the line landmarks (it defines its own)
#endif
#ifndef RQ
#define RQ 1                          // passes a lane runs TOGETHER in the elimination and the back-substitution: RQ independent
#endif                                // 6x6 solves in one instruction stream hide each other's fp64 latency (one wavefront per SIMD)
#ifndef R_COLS
#define R_COLS 1                      // round 6: the landmark systems are eliminated with one matrix column per lane (p_solve6_cols, lf_pose_wg.h)
#endif
#define R_CM 48                       // compact measurement: nA nB nMa nMb oA oB oMa oMb
static_assert(RP_N * RP_ROWS >= LF_MAX_MATCHES, "four passes cover the match list");

struct ResShared {
  double wb[LF_MAX_MATCHES * R_ROW];        // W (row-major 6x6, rows = pose) | bl of the current linearisation
  double L[2][LF_MAX_MATCHES * 6];          // landmarks: current set and the trial step's
  double red[2][R_RED_N];                   // per-landmark terms of the ordered sums (zero padded to a multiple of 8)
  double tile[RQ * RP_ROWS * R_ROW];        // one row per match of the RQ passes running together
  double hb[42], sg[42];                    // Hpp | bp of the linearisation; S | g of the current damping
  double wred[RW_N];
  lf_se3 xp[12];
  lf_se3 X[2];                               // the older camera's pose: current and the trial step's
  int wcnt[RW_N], wit[RW_N];
};
#ifndef RP_VI
#define RP_VI 5                       // passes whose Vi column stays in registers between r_eliminate and r_backsub (the rest solve again)
#endif
struct ResRegs {
  double v[RP_N][6];                  // row d of V of the lane's match, per pass (static indices only)
  double vi[RP_VI > 0 ? RP_VI : 1][6];   // column d of (V + lambda I)^-1 of the last elimination, per pass
};
// (the pass number is wave-uniform: a scalar branch to the copy from / to the pass's fixed registers; the empty asm keeps
// the branches apart -- merged, they become one access with a selected address and the arrays fall back to memory)
template <int NP>
__device__ __forceinline__ void r_get6(const double (&arr)[NP][6], int p, double *out) {
#define R_CASE(q) case q: _Pragma("unroll") for (int k = 0; k < 6; k++) out[k] = arr[q < NP ? q : 0][k]; asm volatile("; pass " #q); break;
  switch (p) { R_CASE(0) R_CASE(1) R_CASE(2) R_CASE(3) R_CASE(4) R_CASE(5) default: R_CASE(6) }
#undef R_CASE
}
template <int NP>
__device__ __forceinline__ void r_put6(double (&arr)[NP][6], int p, const double *in) {
#define R_CASE(q) case q: _Pragma("unroll") for (int k = 0; k < 6; k++) arr[q < NP ? q : 0][k] = in[k]; asm volatile("; pass " #q); break;
  switch (p) { R_CASE(0) R_CASE(1) R_CASE(2) R_CASE(3) R_CASE(4) R_CASE(5) default: R_CASE(6) }
#undef R_CASE
}
static_assert(RP_N <= 7, "r_get6 / r_put6 enumerate seven passes");

__device__ __forceinline__ void r_meas(const double *cm, int k, lf_line_meas *m) {
  const double *c = cm + (size_t)k * R_CM;
  m->nA = c; m->nB = c + 3; m->nMa = c + 6; m->nMb = c + 15;
  m->oA = c + 24; m->oB = c + 27; m->oMa = c + 30; m->oMb = c + 39;
}
// the evaluation loops of r_blocks: unrolled (the selects on the evaluation number fold away); measured faster than rolled
// at one wavefront per SIMD -- the independent evaluations are the instruction-level parallelism that hides the fp64 latency
#define R_EV_PRAGMA _Pragma("unroll")
#if defined(LF_POSE_PROFILE) && LF_POSE_PROFILE == 2      // finer split of r_eliminate (slots 1..4) ...
#define PTE(k) PT(k)
#define PTB(k) do { } while (0)
#elif defined(LF_POSE_PROFILE) && LF_POSE_PROFILE == 3    // ... or of r_blocks
#define PTE(k) do { } while (0)
#define PTB(k) PT(k)
#else
#define PTE(k) do { } while (0)
#define PTB(k) do { } while (0)
#endif
struct ResTask { int i, d, g, row; bool act; };
__device__ __forceinline__ ResTask r_task(int pass, int n) {
  ResTask t;
  const int lane = p_lane();
  t.g = lane / 6; t.d = lane - 6 * t.g;
  if (t.g >= PG_N) { t.g = PG_N - 1; t.d = 0; }
  t.row = (int)(threadIdx.x >> 6) * PG_N + t.g;
  t.i = pass * RP_ROWS + t.row;
  t.act = lane < 6 * PG_N && t.i < n;
  return t;
}
__device__ __forceinline__ double r_sib(double v, const ResTask &t, int k) { return __shfl(v, 6 * t.g + k, 64); }
// acc (+/-)= tile[r][col] for r = 0..cnt-1, strictly in that order (rows of a pass are in match order)
template <bool SUB>
__device__ __forceinline__ double r_stage_walk(const double *tile, int col, int cnt, double acc) {
  int r = 0;
  for (; r + 8 <= cnt; r += 8) {
    double q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = tile[(r + k) * R_ROW + col];
#pragma unroll
    for (int k = 0; k < 8; k++) acc = SUB ? acc - q[k] : acc + q[k];
  }
  for (; r < cnt; r++) acc = SUB ? acc - tile[r * R_ROW + col] : acc + tile[r * R_ROW + col];
  return acc;
}
__device__ __forceinline__ double r_block_max(ResShared &S, double mx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
  if (p_lane() == 0) S.wred[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = S.wred[0];
#pragma unroll
  for (int w = 1; w < RW_N; w++) { double t = S.wred[w]; mx = t > mx ? t : mx; }
  __syncthreads();
  return mx;
}

// lf_solve6 (lf_linalg.h) on Q independent systems at once, one right-hand side each: the SAME operations on every entry, in
// the same order, so each system's result has the bits lf_solve6 gives it; the Q instruction streams share basic blocks
// (the only branch is the wave-uniform "some lane swaps rows"), which is what lets the scheduler interleave them.  A
// singular system clears ok[q] and keeps computing (inf / nan, never used) instead of returning early.
template <int Q>
__device__ __forceinline__ void r_solve6q(double (&A)[Q][36], double (&B)[Q][6], int (&ok)[Q]) {
  double rp[Q][6];
#pragma unroll
  for (int q = 0; q < Q; q++) ok[q] = 1;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int piv[Q];
    bool swaps = false;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      piv[q] = k;
      double big = lf_fabs(A[q][7 * k]);
#pragma unroll
      for (int i = k + 1; i < 6; i++) {
        const double v = lf_fabs(A[q][6 * i + k]);
        if (v > big) { big = v; piv[q] = i; }
      }
      if (!(big > 0.0)) ok[q] = 0;
      swaps = swaps || piv[q] != k;
    }
    if (LF_ANY(swaps)) {
#pragma unroll
      for (int q = 0; q < Q; q++)
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
          const bool sw = i == piv[q];
#pragma unroll
          for (int j = k; j < 6; j++) { const double a = A[q][6 * k + j], b = A[q][6 * i + j]; A[q][6 * k + j] = sw ? b : a; A[q][6 * i + j] = sw ? a : b; }
          { const double a = B[q][k], b = B[q][i]; B[q][k] = sw ? b : a; B[q][i] = sw ? a : b; }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) rp[q][k] = 1.0 / A[q][7 * k];
#pragma unroll
    for (int q = 0; q < Q; q++)
#pragma unroll
      for (int i = k + 1; i < 6; i++) {
        const double f = A[q][6 * i + k] * rp[q][k];
        const bool nz = f != 0.0;
#pragma unroll
        for (int j = k + 1; j < 6; j++) { const double v = A[q][6 * i + j] - f * A[q][6 * k + j]; A[q][6 * i + j] = nz ? v : A[q][6 * i + j]; }
        { const double v = B[q][i] - f * B[q][k]; B[q][i] = nz ? v : B[q][i]; }
      }
  }
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int q = 0; q < Q; q++) {
      double s = B[q][i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) s -= A[q][6 * i + k] * B[q][k];
      B[q][i] = s * rp[q][i];
    }
}

// lf_match_chi2 of match i at (X, L = Lsrc + 6 i) into red[i]: two threads per match, one per edge (newer camera: the
// landmark itself; older camera: X^-1 L), the two robustified terms added in the order of the sequential code.
__device__ __forceinline__ void r_errchi(const double *cm, const int *set, int n, const lf_se3 *X, const double *Lsrc,
                                         double wgt, double hd, int hub, double *red) {
  if (n <= 0) return;                        // (k_pose_hybrid with point landmarks only: set[] holds nothing)
  const int tid = threadIdx.x, h = tid & 1;
  constexpr int RE_N = (LF_MAX_MATCHES + RT_N / 2 - 1) / (RT_N / 2);   // matches per thread pair: both evaluated in ONE instruction
  double r0[RE_N];                                                      // stream (the second is a repeat of match 0 when n <= 128)
  int i0[RE_N];
#pragma unroll
  for (int r = 0; r < RE_N; r++) {
    i0[r] = r * (RT_N / 2) + (tid >> 1);
    const int i = i0[r] < n ? i0[r] : 0;
    const double *c = cm + (size_t)set[i] * R_CM + 24 * h;
    double PA[3], PB[3], e[6], cc = 0, w;
    if (h) { lf_se3_inv_apply(X, Lsrc + 6 * i, PA); lf_se3_inv_apply(X, Lsrc + 6 * i + 3, PB); }
    else {
#pragma unroll
      for (int k = 0; k < 3; k++) { PA[k] = Lsrc[6 * i + k]; PB[k] = Lsrc[6 * i + 3 + k]; }
    }
    lf_line_edge_error(c + 6, c + 15, c, c + 3, PA, PB, e);
#pragma unroll
    for (int k = 0; k < 6; k++) cc += e[k] * (wgt * e[k]);
    lf_huber(cc, hd, hub, &r0[r], &w);
  }
#pragma unroll
  for (int r = 0; r < RE_N; r++) {
    const double other = __shfl_xor(r0[r], 1, 64);
    if (h == 0 && i0[r] < n) red[i0[r]] = r0[r] + other;
  }
}

__device__ __forceinline__ void r_perturbed_poses(ResShared &S, const lf_se3 *X) {
  const int tid = threadIdx.x;
  if (tid < 12) {
    const int d = tid >> 1;
    const double dl = (tid & 1) ? -1e-9 : 1e-9;
    double v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = (k == d) ? dl : 0.0;
    lf_se3 Xp;
    lf_se3_oplus(X, v, &Xp);
    S.xp[tid] = Xp;
  }
}

// lf_match_blocks_xp: lane (i, d) computes column d of Jn, Jo, Jp, exchanges the columns through the match's tile row,
// and produces row d of V (registers), of W and entry d of bl (LDS), row d of Hpp and entry d of bp (tile row; added
// up in match order by the 42 accumulator lanes -> S.hb).  The eight edge-error evaluations of a match (base, +-delta
// along landmark component d for both cameras, +-delta along pose component d) run as two ROLLED loops over one copy
// of the code: the instruction footprint and the live registers stay small (the pass and evaluation numbers are
// wave-uniform, the selects on them are scalar).
__device__ __forceinline__ void r_edge_eval(const double *ms, const lf_se3 *Xe, const double *L, int d, double sd, double *e) {
  double Lp[6], PA[3], PB[3];
#pragma unroll
  for (int k = 0; k < 6; k++) Lp[k] = (k == d && sd != 0.0) ? L[k] + sd : L[k];
  if (Xe) { lf_se3_inv_apply(Xe, Lp, PA); lf_se3_inv_apply(Xe, Lp + 3, PB); }
  else {
#pragma unroll
    for (int k = 0; k < 3; k++) { PA[k] = Lp[k]; PB[k] = Lp[3 + k]; }
  }
  lf_line_edge_error(ms + 6, ms + 15, ms, ms + 3, PA, PB, e);
}
// acc_init: what accumulator lane tid < 42 starts from (k_pose_hybrid: the point landmarks' Hpp | bp, added up first).
__device__ void r_blocks(ResShared &S, const double *cm, const int *set, int n, const lf_se3 *X, const double *Lc,
                         ResRegs &R, double wgt, double hd, int hub, double *mxl_io, double acc_init = 0.0) {
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  const int tid = threadIdx.x;
  double mxl = *mxl_io, acc = acc_init;
  for (int pass = 0; pass * RP_ROWS < n; pass++) {
    const ResTask t = r_task(pass, n);
    const int i = t.act ? t.i : 0, d = t.d;
    const double *c = cm + (size_t)set[i] * R_CM;
    double *row = S.tile + t.row * R_ROW;
    double L[6], vn[6], sbl_n = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) L[k] = Lc[6 * i + k];
    {   // ---- edge to the newer camera (the world frame): the landmark itself against nA nB nMa nMb
      double en[6], cn[6], ep[6], cc = 0, r0, wn;
R_EV_PRAGMA
      for (int ev = 0; ev < 3; ev++) {
        double e[6];
        r_edge_eval(c, (const lf_se3 *)0, L, d, ev == 1 ? delta : ev == 2 ? -delta : 0.0, e);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if (ev == 0) en[k] = e[k];
          else if (ev == 1) ep[k] = e[k];
          else cn[k] = scalar * (ep[k] - e[k]);
        }
      }
      PTB(1);
      for (int k = 0; k < 6; k++) cc += en[k] * (wgt * en[k]);
      lf_huber(cc, hd, hub, &r0, &wn);
      wn = wn * wgt;
      if (t.act) for (int k = 0; k < 6; k++) row[6 * k + d] = cn[k];           // Jn
      p_wave_order();
#pragma unroll
      for (int j = 0; j < 6; j++) { double s = 0; for (int k = 0; k < 6; k++) s += cn[k] * (wn * row[6 * k + j]); vn[j] = s; }
      p_wave_order();
      for (int k = 0; k < 6; k++) { const double wen = wn * en[k]; sbl_n += cn[k] * wen; }
    }
    PTB(2);
    // ---- edge to the older camera (pose X): X^-1 L against oA oB oMa oMb; differences along the landmark and the pose
    double eo[6], co[6], cp[6], wo, vo[6], hw[6], hp[6];
    {
      double ep[6], cc = 0, r0;
R_EV_PRAGMA
      for (int ev = 0; ev < 5; ev++) {
        double e[6];
        const lf_se3 *Xe = ev < 3 ? X : &S.xp[2 * d + (ev - 3)];   // S.xp[2 d], [2 d + 1]: X (+) (+-delta e_d)
        r_edge_eval(c + 24, Xe, L, d, ev == 1 ? delta : ev == 2 ? -delta : 0.0, e);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if (ev == 0) eo[k] = e[k];
          else if (ev == 1 || ev == 3) ep[k] = e[k];
          else if (ev == 2) co[k] = scalar * (ep[k] - e[k]);
          else cp[k] = scalar * (ep[k] - e[k]);
        }
      }
      PTB(3);
      for (int k = 0; k < 6; k++) cc += eo[k] * (wgt * eo[k]);
      lf_huber(cc, hd, hub, &r0, &wo);
      wo = wo * wgt;
    }
    if (t.act) for (int k = 0; k < 6; k++) row[6 * k + d] = co[k];           // Jo
    p_wave_order();
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double s = 0, s2 = 0;
      for (int k = 0; k < 6; k++) { const double jo = row[6 * k + j]; s += co[k] * (wo * jo); s2 += cp[k] * (wo * jo); }
      vo[j] = s; hw[j] = s2;
    }
    p_wave_order();
    if (t.act) for (int k = 0; k < 6; k++) row[6 * k + d] = cp[k];           // Jp
    p_wave_order();
#pragma unroll
    for (int j = 0; j < 6; j++) { double s = 0; for (int k = 0; k < 6; k++) s += cp[k] * (wo * row[6 * k + j]); hp[j] = s; }
    p_wave_order();
    PTB(4);
    double sbl_o = 0, sbp = 0;
    for (int k = 0; k < 6; k++) { const double weo = wo * eo[k]; sbl_o += co[k] * weo; sbp += cp[k] * weo; }
    double *wbrow = S.wb + i * R_ROW;
    double Vr[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Vr[j] = vn[j] + vo[j];
    r_put6(R.v, pass, Vr);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double V = Vr[j];
      if (t.act) {
        wbrow[6 * d + j] = hw[j];
        row[6 * d + j] = hp[j];
        if (j == d) { const double a = lf_fabs(V); if (a > mxl) mxl = a; }
      }
    }
    if (t.act) { wbrow[36 + d] = -(sbl_n + sbl_o); row[36 + d] = -sbp; }
    PTB(8);
    __syncthreads();
    if (tid < 42) { const int left = n - pass * RP_ROWS; acc = r_stage_walk<false>(S.tile, tid, left < RP_ROWS ? left : RP_ROWS, acc); }
    __syncthreads();
  }
  if (tid < 42) S.hb[tid] = acc;
  *mxl_io = mxl;
}

// lf_match_eliminate at damping lambda: the six lanes of a match gather V from their registers through the tile row;
// lane d solves (V + lambda I) x = e_d (column d of Vi, kept in registers for the back-substitution), publishes
// column d of W Vi, then row d of T = W Vi W^T and entry d of u = W Vi bl go to the tile row and are subtracted in
// match order from Hpp + lambda I | bp -> S.sg.  Returns 1 if a match of this lane is singular.
// ptu / np: T | u rows (42 doubles each) of np landmarks that come BEFORE the matches in the sums (k_pose_hybrid: the points).
__device__ int r_eliminate(ResShared &S, int n, double lambda, ResRegs &R, const double *ptu = nullptr, int np = 0) {
  const int tid = threadIdx.x;
  int bad = 0;
  double acc = 0.0;
  if (tid < 42) {
    acc = S.hb[tid];
    if (tid < 36 && tid % 7 == 0) acc = acc + lambda;
    if (np > 0) acc = p_walk<true>(ptu + tid, 42, np, acc);
  }
  for (int pass = 0; pass * RP_ROWS < n; pass += RQ) {
    ResTask t[RQ];
    double *row[RQ];
    const double *wbrow[RQ];
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      t[q] = r_task(pass + q, n);
      row[q] = S.tile + (q * RP_ROWS + t[q].row) * R_ROW;
      wbrow[q] = S.wb + (t[q].act ? t[q].i : 0) * R_ROW;
    }
    const int d = t[0].d;
    double x[RQ][6], wvc[RQ][6], wv[RQ][6];
    int ok[RQ];
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      double Vr[6];
      r_get6(R.v, pass + q, Vr);
      if (t[q].act) {
#pragma unroll
        for (int j = 0; j < 6; j++) row[q][6 * d + j] = Vr[j];
      }
    }
    p_wave_order();
    double W[RQ][R_ROW];      // W | bl of the match, fetched with the V gather: the LDS latency of the products after the solve
#if R_COLS                    //   would otherwise be exposed read by read (one wavefront per SIMD: nothing else to issue)
    double a_[RQ][6];         // round 6: column d of V only -- the system is eliminated with one column per lane (p_solve6_cols)
#pragma unroll
    for (int q = 0; q < RQ; q++) {
#pragma unroll
      for (int k = 0; k < 6; k++) a_[q][k] = row[q][6 * k + d];
#pragma unroll
      for (int k = 0; k < R_ROW; k++) W[q][k] = wbrow[q][k];
    }
    p_wave_order();
#pragma unroll
    for (int q = 0; q < RQ; q++)
#pragma unroll
      for (int k = 0; k < R_ROW; k++) asm volatile("" : "+v"(W[q][k]));   // (in registers from here on, not re-read later)
    PTE(1);
#pragma unroll
    for (int q = 0; q < RQ; q++) {
#pragma unroll
      for (int k = 0; k < 6; k++) { if (k == d) a_[q][k] += lambda; x[q][k] = (k == d) ? 1.0 : 0.0; }
      ok[q] = p_solve6_cols(a_[q], x[q], 6 * t[q].g, d, p_lane() < 6 * PG_N, row[q]);
    }
#else
    double A[RQ][36];
#pragma unroll
    for (int q = 0; q < RQ; q++) {
#pragma unroll
      for (int k = 0; k < 36; k++) A[q][k] = row[q][k];
#pragma unroll
      for (int k = 0; k < R_ROW; k++) W[q][k] = wbrow[q][k];
    }
    p_wave_order();
#pragma unroll
    for (int q = 0; q < RQ; q++)
#pragma unroll
      for (int k = 0; k < R_ROW; k++) asm volatile("" : "+v"(W[q][k]));   // (in registers from here on, not re-read later)
    PTE(1);
#pragma unroll
    for (int q = 0; q < RQ; q++)
#pragma unroll
      for (int k = 0; k < 6; k++) { A[q][7 * k] += lambda; x[q][k] = (k == d) ? 1.0 : 0.0; }
    r_solve6q<RQ>(A, x, ok);
#endif
    PTE(2);
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      if (t[q].act && !ok[q]) bad = 1;
      if (RP_VI > 0 && pass + q < RP_VI) r_put6(R.vi, pass + q, x[q]);
#pragma unroll
      for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 6; k++) s += W[q][6 * r + k] * x[q][k]; wvc[q][r] = s; }
      if (t[q].act) {
#pragma unroll
        for (int k = 0; k < 6; k++) row[q][6 * k + d] = wvc[q][k];
      }
    }
    p_wave_order();
#pragma unroll
    for (int q = 0; q < RQ; q++)
#pragma unroll
      for (int k = 0; k < 6; k++) wv[q][k] = row[q][6 * d + k];
    p_wave_order();
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      double u = 0, T[6];
#pragma unroll
      for (int k = 0; k < 6; k++) u += wv[q][k] * W[q][36 + k];
#pragma unroll
      for (int j = 0; j < 6; j++) { double s2 = 0; for (int k = 0; k < 6; k++) s2 += wv[q][k] * W[q][6 * j + k]; T[j] = s2; }
      if (t[q].act) {
#pragma unroll
        for (int j = 0; j < 6; j++) row[q][6 * d + j] = T[j];
        row[q][36 + d] = u;
      }
    }
    PTE(3);
    __syncthreads();
    if (tid < 42) { const int left = n - pass * RP_ROWS; acc = r_stage_walk<true>(S.tile, tid, left < RQ * RP_ROWS ? left : RQ * RP_ROWS, acc); }
    __syncthreads();
    PTE(4);
  }
  if (tid < 42) S.sg[tid] = acc;
  return bad;
}

// lf_match_backsub + the step's scale term: lane (i, a) computes component a of r = bl - W^T dp, gets row a of Vi from
// the six column holders through the tile row, and writes component a of the trial landmark.
__device__ void r_backsub(ResShared &S, int n, const double *dp, double lambda, const double *Lc, double *Lt, ResRegs &R,
                          double *red) {
  for (int pass = 0; pass * RP_ROWS < n; pass += RQ) {
    ResTask t[RQ];
    double *row[RQ];
    const double *wbrow[RQ];
    int i[RQ];
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      t[q] = r_task(pass + q, n);
      i[q] = t[q].act ? t[q].i : 0;
      row[q] = S.tile + (q * RP_ROWS + t[q].row) * R_ROW;
      wbrow[q] = S.wb + i[q] * R_ROW;
    }
    const int a = t[0].d;
    double rr[RQ][6], bla[RQ];
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      double tt = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) tt += wbrow[q][6 * k + a] * dp[k];
      bla[q] = wbrow[q][36 + a];
      const double ra = bla[q] - tt;
#pragma unroll
      for (int k = 0; k < 6; k++) rr[q][k] = r_sib(ra, t[q], k);
    }
    if (RP_VI > 0 && pass + RQ <= RP_VI) {   // column a of Vi as r_eliminate left it (same lambda: the back-substitution follows it)
#pragma unroll
      for (int q = 0; q < RQ; q++) {
        double xc[6];
        r_get6(R.vi, pass + q, xc);
        if (t[q].act) {
#pragma unroll
          for (int k = 0; k < 6; k++) row[q][6 * k + a] = xc[k];
        }
      }
    } else {   // beyond the kept passes: column a of Vi again, by the arithmetic of r_eliminate (same bits)
      double xc[RQ][6];
      int ok[RQ];
#pragma unroll
      for (int q = 0; q < RQ; q++) {
        double Vr[6];
        r_get6(R.v, pass + q, Vr);
        if (t[q].act) {
#pragma unroll
          for (int j = 0; j < 6; j++) row[q][6 * a + j] = Vr[j];
        }
      }
      p_wave_order();
#if R_COLS
#pragma unroll
      for (int q = 0; q < RQ; q++) {
        double ac[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { ac[k] = row[q][6 * k + a]; xc[q][k] = (k == a) ? 1.0 : 0.0; }
        p_wave_order();
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == a) ac[k] += lambda;
        ok[q] = p_solve6_cols(ac, xc[q], 6 * t[q].g, a, p_lane() < 6 * PG_N, row[q]);
      }
#else
      double A[RQ][36];
#pragma unroll
      for (int q = 0; q < RQ; q++)
#pragma unroll
        for (int k = 0; k < 36; k++) A[q][k] = row[q][k];
      p_wave_order();
#pragma unroll
      for (int q = 0; q < RQ; q++)
#pragma unroll
        for (int k = 0; k < 6; k++) { A[q][7 * k] += lambda; xc[q][k] = (k == a) ? 1.0 : 0.0; }
      r_solve6q<RQ>(A, xc, ok);
#endif
#pragma unroll
      for (int q = 0; q < RQ; q++)
        if (t[q].act) {
#pragma unroll
          for (int k = 0; k < 6; k++) row[q][6 * k + a] = xc[q][k];
        }
    }
    p_wave_order();
    double vi[RQ][6];
#pragma unroll
    for (int q = 0; q < RQ; q++)
#pragma unroll
      for (int k = 0; k < 6; k++) vi[q][k] = row[q][6 * a + k];
    p_wave_order();
#pragma unroll
    for (int q = 0; q < RQ; q++) {
      double dl = 0, s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) dl += vi[q][k] * rr[q][k];
      const double La = Lc[6 * i[q] + a] + dl, term = dl * (lambda * dl + bla[q]);
      if (t[q].act) Lt[6 * i[q] + a] = La;
#pragma unroll
      for (int k = 0; k < 6; k++) s += r_sib(term, t[q], k);
      if (t[q].act && a == 0) red[i[q]] = s;
    }
  }
}
