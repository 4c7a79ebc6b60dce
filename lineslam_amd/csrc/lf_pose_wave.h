// lf_pose_wave.h -- the LM refinement of k_pose (lines only) in a form that SHARES compute units (round 6).
//
// lf_pose_res.h keeps the whole state of a refinement on chip: 512 registers on each of the four SIMDs of a CU and 130 KB
// of LDS per pair, i.e. a k_pose workgroup needs an EMPTY compute unit.  Next to the long-lived wavefronts of the front
// end that never exists, so the pair stage waited for the chip to drain and the passes in flight fell into lock-step
// (profiles/r05_timeline_pipelined_lockstep.txt).  This form fits BESIDE other wavefronts:
//   * WV_W wavefronts per pair (a workgroup of 64 WV_W threads), <= 128 registers each (one of them fits next to two
//     front-end wavefronts on a SIMD) and ~8 KB of LDS per wavefront: one exchange row and one W | bl row per match of the
//     running pass, the per-match terms of the ordered sums, the inlier lists;
//   * the per-match blocks of a linearisation -- V (36, stored by columns), W | bl (42), the columns of (V + lambda I)^-1
//     of the last elimination (36) and both landmark sets (12) -- live in the pair's WORKSPACE (L2 / Infinity Cache;
//     written and read by this one workgroup in whole rows), ten matches per wavefront and pass, six lanes per match;
//   * the 6x6 systems are eliminated with ONE MATRIX COLUMN PER LANE (w_solve6_cols): the pivot column is broadcast inside
//     the six-lane group, every lane updates its own column of A and of the right-hand side -- 12 doubles of state instead
//     of a replicated 36 + 6, a third of the instructions; the pose system S dp = g goes through the same routine;
//   * every phase is a function of its own (not inlined): its registers are allocated for it alone.
// Every value is produced by the same expression, and every sum over the matches runs in the same order, as in the
// sequential lf_match_blocks_xp / lf_match_eliminate / lf_match_backsub (lf_pose.h) and in lf_pose_res.h -- the oracle's
// bits (tests/test_pose_golden_gpu.py, test_fullsize_gpu.py, test_pair_sizes_gpu.py).
// Reference: getTransformFromHybridMatchesG2O, src/transformation_estimation.cpp:218-461 (line edges);
// EdgeSE3LineEndpts, src/line/edge_se3_lineendpts.cpp:146-189.  Included by lf_pair.hip after lf_pose_res.h.
#pragma once
#ifndef WV_W
#define WV_W 4                         // wavefronts per pair
#endif
#define WV_T (64 * WV_W)               // threads per pair
#define WV_G 10                        // matches per wavefront and pass: 6 lanes each, lanes 60..63 idle
#define WV_ROWS (WV_G * WV_W)          // matches per pass of the workgroup
#define WV_WB 42                       // W (36, rows = pose) | bl (6) of one match
#define WV_XROW 42                     // exchange row of one match: Jacobian columns, W Vi, then Hpp | bp or T | u
// the pair's workspace behind the compact measurements and the RANSAC winner (LF_PAIR_WS_DOUBLES, lf_pair.h)
#define WV_OFF_TF (LF_MAX_MATCHES * R_CM + 8)                     // 16 floats: the winning three-line model (k_ransac)
#define WV_OFF_VT (LF_MAX_MATCHES * R_CM + 16)                    // [match][column d][6]: V by columns (lane d solves column d)
#define WV_OFF_WB (WV_OFF_VT + LF_MAX_MATCHES * 36)               // [match][42]: W | bl
#define WV_OFF_VI (WV_OFF_WB + LF_MAX_MATCHES * WV_WB)            // [match][column d][6]: column d of (V + lambda I)^-1
#define WV_OFF_L (WV_OFF_VI + LF_MAX_MATCHES * 36)                // [2][match][6]: landmarks, current set and the trial step's
#define WV_WS_END (WV_OFF_L + 2 * LF_MAX_MATCHES * 6)
static_assert(WV_WS_END <= LF_PAIR_WS_DOUBLES, "the blocks fit the pair's workspace");
static_assert(LF_MAX_MATCHES <= 256, "match indices are kept in bytes");
typedef unsigned char wv_idx;

struct alignas(16) WaveShared {
  double stage[WV_ROWS * WV_WB];             // W | bl rows of the running pass of an elimination; w_blocks: the Jp columns of the pass
  double xch[WV_ROWS * WV_XROW];             // exchange rows
  double red[2][LF_MAX_MATCHES + 8];         // per-match terms of the ordered sums (zero padded to a multiple of 8)
  double hb[42], sg[42];                     // Hpp | bp of the linearisation; S | g of the current damping
  double dp[6];                              // the pose step of the current damping
  double wred[WV_W];
  double lm_lambda, lm_ni, lm_chi;           // the LM state between the phases (every thread writes the same bits)
  double wscale[WV_W];
  lf_se3 xp[12];
  lf_se3 X[2];                               // the older camera's pose: current and the trial step's
  float tf[16];                              // the pair's transform between the refinements (float, as the reference keeps it)
  double sse_d; float sse_f;                 // w_score: the two sums of the accepted squared distances
  int wcnt[WV_W];
  int any_flag;
  wv_idx idx[LF_MAX_MATCHES];                // scratch inlier list of the re-scoring loop
  wv_idx set[LF_MAX_MATCHES];                // current inlier list (indices into the match list)
};

// pointers handed to the (deliberately not inlined) phases arrive as generic addresses: say that they are global memory
typedef __attribute__((address_space(1))) double wv_gd;
#define WV_G_RO(p) ((const double *)w_uni_g((const wv_gd *)(p)))
#define WV_G_RW(p) ((double *)w_uni_g((wv_gd *)(p)))
#define WV_PHASE __device__ __noinline__   // a phase is a function of its own: its registers are allocated for it alone (a kernel
                                           // with every phase inlined needs > 400 registers, the scheduler stretching each phase's
                                           // loads over the others'); a call costs the callee-saved registers once per phase
// the arguments of a phase arrive in vector registers (the calling convention knows nothing about uniformity); they ARE the
// same in every lane: move them to scalar registers, where addresses and loop counters cost no vector register
__device__ __forceinline__ int w_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double w_uni(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <class T> __device__ __forceinline__ T *w_uni(T *p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const wv_gd *w_uni_g(const wv_gd *p) {     // (the integer goes back to a GLOBAL pointer: global_load / global_store, not flat)
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const wv_gd *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ wv_gd *w_uni_g(wv_gd *p) { return (wv_gd *)w_uni_g((const wv_gd *)p); }
// w_order: the workgroup's LDS and workspace writes are visible to all its threads.  w_order_wave: the same inside one
// wavefront (the six lanes of a match always sit in one wavefront).
#if WV_W > 1
#ifdef WV_DIAG_AGENT_FENCE
__device__ __forceinline__ void w_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
#else
__device__ __forceinline__ void w_order() { __syncthreads(); }
#endif
#else
__device__ __forceinline__ void w_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
#endif
__device__ __forceinline__ void w_order_wave() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// "does any thread of the pair say so": an LDS flag between two barriers (every call is preceded by a barrier of its own phase)
struct WaveShared;
__device__ __forceinline__ int w_any(WaveShared &S, int v);

struct WvTask { int i, d, g, base, row; bool act, lane_ok; };
__device__ __forceinline__ WvTask w_task(int pass, int n) {
  WvTask t;
  const int lane = p_lane();
  t.g = lane / 6; t.d = lane - 6 * t.g;
  t.lane_ok = lane < 6 * WV_G;
  if (!t.lane_ok) { t.g = WV_G - 1; t.d = 0; }          // lanes 60..63 shadow lane 54 (same loads, same results, no stores)
  t.base = 6 * t.g;
  t.row = (int)(threadIdx.x >> 6) * WV_G + t.g;
  t.i = pass * WV_ROWS + t.row;
  t.act = t.lane_ok && t.i < n;
  return t;
}

__device__ __forceinline__ int w_any(WaveShared &S, int v) {
#if WV_W > 1
  if (threadIdx.x == 0) S.any_flag = 0;
  w_order();
  if (v) S.any_flag = 1;
  w_order();
  const int r = S.any_flag;
  w_order();                                   // (read by everyone before the next call clears it)
  return r;
#else
  return LF_ANY(v != 0) ? 1 : 0;
#endif
}
__device__ __forceinline__ int w_solve6_cols(double (&a)[6], double (&b)[6], const WvTask &t, double *urow) {
  return p_solve6_cols(a, b, t.base, t.d, t.lane_ok, urow);       // (lf_pose_wg.h)
}

// lf_match_chi2 of match i at (X, L = Lsrc + 6 i) into red[i]: two lanes per match, one per edge (newer camera: the
// landmark itself; older camera: X^-1 L), the two robustified terms added in the order of the sequential code.
// (No barrier inside: the caller orders.)
WV_PHASE void w_errchi(WaveShared &S, const double *cm_, int n, int xi, const double *Lsrc_,
                       double wgt, double hd, int hub, int ri) {
  n = w_uni(n); xi = w_uni(xi); ri = w_uni(ri); hub = w_uni(hub); wgt = w_uni(wgt); hd = w_uni(hd);
  const wv_idx *set = S.set;                    // (the current inlier list: always S.set inside a refinement)
  const double *cm = WV_G_RO(cm_), *Lsrc = WV_G_RO(Lsrc_);
  const lf_se3 *X = &S.X[xi];
  double *red = S.red[ri];
  const int tid = threadIdx.x, h = tid & 1;
  for (int r = 0; r * (WV_T / 2) < n; r++) {
    const int i0 = r * (WV_T / 2) + (tid >> 1);
    const int i = i0 < n ? i0 : 0;
    const double *c = cm + (size_t)set[i] * R_CM + 24 * h;
    double PA[3], PB[3], e[6], cc = 0, w, r0;
    if (h) { lf_se3_inv_apply(X, Lsrc + 6 * i, PA); lf_se3_inv_apply(X, Lsrc + 6 * i + 3, PB); }
    else {
#pragma unroll
      for (int k = 0; k < 3; k++) { PA[k] = Lsrc[6 * i + k]; PB[k] = Lsrc[6 * i + 3 + k]; }
    }
    lf_line_edge_error(c + 6, c + 15, c, c + 3, PA, PB, e);
#pragma unroll
    for (int k = 0; k < 6; k++) cc += e[k] * (wgt * e[k]);
    lf_huber(cc, hd, hub, &r0, &w);
    const double other = __shfl_xor(r0, 1, 64);
    if (h == 0 && i0 < n) red[i0] = r0 + other;
  }
}

__device__ __forceinline__ void w_perturbed_poses(WaveShared &S, const lf_se3 *X) {
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

#ifndef WV_SCHED_FENCE
#define WV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef WV_EV_PRAGMA
#define WV_EV_PRAGMA _Pragma("unroll 1")   // the evaluation loops stay ROLLED: other wavefronts hide the latency, the registers stay few
#endif
// lf_match_blocks_xp: lane (i, d) computes column d of Jn, Jo, Jp, exchanges the columns through the match's exchange
// row, and produces row d of V, of W and entry d of bl (workspace rows of the match), row d of Hpp and entry d of bp
// (exchange row; added up in match order by 42 accumulator lanes -> S.hb).  Same expressions as r_blocks (lf_pose_res.h),
// in TWO phases so that each fits its registers: w_blocks_n = the edge to the newer camera (Jn -> the newer half of V and
// of bl, left in the workspace), w_blocks_o = the edge to the older camera (Jo, Jp -> V, W, bl completed, Hpp | bp summed).
// A column leaves the registers as soon as it exists (exchange row; Jp in the match's row of S.stage, which no elimination
// is using now) and the products read every operand from there.
WV_PHASE void w_blocks_n(WaveShared &S, const double *cm_, int n, const double *Lc_, double *vtg_, double *wbg_,
                         double wgt, double hd, int hub) {
  n = w_uni(n); hub = w_uni(hub); wgt = w_uni(wgt); hd = w_uni(hd);
  const wv_idx *set = S.set;
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  const double *cm = WV_G_RO(cm_), *Lc = WV_G_RO(Lc_);
  double *vtg = WV_G_RW(vtg_), *wbg = WV_G_RW(wbg_);
  for (int pass = 0; pass * WV_ROWS < n; pass++) {
    const WvTask t = w_task(pass, n);
    const int i = t.act ? t.i : 0, d = t.d;
    const double *c = cm + (size_t)set[i] * R_CM;
    double *row = S.xch + t.row * WV_XROW;
    double L[6];
#pragma unroll
    for (int k = 0; k < 6; k++) L[k] = Lc[6 * i + k];
    // ---- edge to the newer camera (the world frame): the landmark itself against nA nB nMa nMb
    double en[6], ep[6], cc = 0, r0, wn;
WV_EV_PRAGMA
    for (int ev = 0; ev < 3; ev++) {
      double e[6];
      r_edge_eval(c, (const lf_se3 *)0, L, d, ev == 1 ? delta : ev == 2 ? -delta : 0.0, e);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        if (ev == 0) en[k] = e[k];
        else if (ev == 1) ep[k] = e[k];
        else if (t.act) row[6 * k + d] = scalar * (ep[k] - e[k]);           // Jn, column d
      }
    }
    for (int k = 0; k < 6; k++) cc += en[k] * (wgt * en[k]);
    lf_huber(cc, hd, hub, &r0, &wn);
    wn = wn * wgt;
    w_order_wave();
    double *vt = vtg + (size_t)i * 36, *wb = wbg + (size_t)i * WV_WB;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += row[6 * k + d] * (wn * row[6 * k + j]);
      if (t.act) vt[6 * j + d] = s;                                           // the newer half of V(d, j)
      WV_SCHED_FENCE();
    }
    double sbl_n = 0;
    for (int k = 0; k < 6; k++) { const double wen = wn * en[k]; sbl_n += row[6 * k + d] * wen; }
    if (t.act) wb[36 + d] = sbl_n;
    w_order_wave();
  }
}
// Returns max |V(d, d)| over the thread's tasks (computeLambdaInit).
WV_PHASE double w_blocks_o(WaveShared &S, const double *cm_, int n, int cur, const double *Lc_,
                           double *vtg_, double *wbg_, double wgt, double hd, int hub) {
  n = w_uni(n); cur = w_uni(cur); hub = w_uni(hub); wgt = w_uni(wgt); hd = w_uni(hd);
  const wv_idx *set = S.set;
  const double delta = 1e-9, scalar = 1.0 / (2 * 1e-9);
  const double *cm = WV_G_RO(cm_), *Lc = WV_G_RO(Lc_);
  double *vtg = WV_G_RW(vtg_), *wbg = WV_G_RW(wbg_);
  const lf_se3 *X = &S.X[cur];
  const int tid = threadIdx.x;
  double mxl = 0.0, acc = 0.0;
  for (int pass = 0; pass * WV_ROWS < n; pass++) {
    const WvTask t = w_task(pass, n);
    const int i = t.act ? t.i : 0, d = t.d;
    const double *c = cm + (size_t)set[i] * R_CM;
    double *row = S.xch + t.row * WV_XROW, *jp = S.stage + t.row * WV_WB;
    double *vt = vtg + (size_t)i * 36, *wb = wbg + (size_t)i * WV_WB;
    // ---- edge to the older camera (pose X): X^-1 L against oA oB oMa oMb; differences along the landmark and the pose
    double eo[6], wo;
    {
      double L[6], ep[6], cc = 0, r0;
#pragma unroll
      for (int k = 0; k < 6; k++) L[k] = Lc[6 * i + k];
WV_EV_PRAGMA
      for (int ev = 0; ev < 5; ev++) {
        double e[6];
        const lf_se3 *Xe = ev < 3 ? X : &S.xp[2 * d + (ev - 3)];   // S.xp[2 d], [2 d + 1]: X (+) (+-delta e_d)
        r_edge_eval(c + 24, Xe, L, d, ev == 1 ? delta : ev == 2 ? -delta : 0.0, e);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if (ev == 0) eo[k] = e[k];
          else if (ev == 1 || ev == 3) ep[k] = e[k];
          else if (ev == 2) { if (t.act) row[6 * k + d] = scalar * (ep[k] - e[k]); }      // Jo, column d
          else { if (t.act) jp[6 * k + d] = scalar * (ep[k] - e[k]); }                    // Jp, column d
        }
      }
      for (int k = 0; k < 6; k++) cc += eo[k] * (wgt * eo[k]);
      lf_huber(cc, hd, hub, &r0, &wo);
      wo = wo * wgt;
    }
    w_order_wave();
    double hp[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double s = 0, s2 = 0, s3 = 0;
      for (int k = 0; k < 6; k++) {
        const double jo = row[6 * k + j];
        s += row[6 * k + d] * (wo * jo); s2 += jp[6 * k + d] * (wo * jo);
        s3 += jp[6 * k + d] * (wo * jp[6 * k + j]);
      }
      const double V = vt[6 * j + d] + s;       // (the newer half, left there by w_blocks_n of this same thread)
      hp[j] = s3;
      if (t.act) {
        vt[6 * j + d] = V;                      // V(d, j), stored by columns
        wb[6 * d + j] = s2;
        if (j == d) { const double a = lf_fabs(V); if (a > mxl) mxl = a; }
      }
      WV_SCHED_FENCE();                         // one j at a time: the scheduler would fetch all 72 operands of the six at once
    }
    double sbl_o = 0, sbp = 0;
    for (int k = 0; k < 6; k++) { const double weo = wo * eo[k]; sbl_o += row[6 * k + d] * weo; sbp += jp[6 * k + d] * weo; }
    w_order_wave();                             // every lane of the match has read Jo: the row takes Hpp | bp now
    if (t.act) {
#pragma unroll
      for (int j = 0; j < 6; j++) row[6 * d + j] = hp[j];
      wb[36 + d] = -(wb[36 + d] + sbl_o); row[36 + d] = -sbp;
    }
    w_order();
    if (tid < 42) { const int left = n - pass * WV_ROWS; acc = r_stage_walk<false>(S.xch, tid, left < WV_ROWS ? left : WV_ROWS, acc); }
    w_order();
  }
  if (tid < 42) S.hb[tid] = acc;
  return mxl;
}

// lf_match_eliminate at damping lambda, ten matches per wavefront and pass: the pass's W | bl rows are staged in LDS
// (whole rows, coalesced); lane (i, d) reads column d of V from the workspace, solves (V + lambda I) x = e_d with the group's
// column-per-lane elimination -> column d of Vi (kept in the workspace for the back-substitution), publishes column d of
// W Vi, then row d of T = W Vi W^T and entry d of u = W Vi bl go to the exchange row and are subtracted in match order from
// Hpp + lambda I | bp -> S.sg.  Returns 1 if a match of this thread is singular.
WV_PHASE int w_eliminate(WaveShared &S, int n, double lambda, const double *vtg_, const double *wbg_, double *vig_) {
  n = w_uni(n); lambda = w_uni(lambda);
  const double *vtg = WV_G_RO(vtg_), *wbg = WV_G_RO(wbg_);
  double *vig = WV_G_RW(vig_);
  const int tid = threadIdx.x;
  int bad = 0;
  double acc = 0.0;
  if (tid < 42) {
    acc = S.hb[tid];
    if (tid < 36 && tid % 7 == 0) acc = acc + lambda;
  }
  for (int pass = 0; pass * WV_ROWS < n; pass++) {
    const WvTask t = w_task(pass, n);
    const int left = n - pass * WV_ROWS, cnt = left < WV_ROWS ? left : WV_ROWS;
    const int i = t.act ? t.i : 0, d = t.d;
    double a[6], x[6];
    {   // column d of V: six contiguous doubles of the workspace
      const double2 *vc = reinterpret_cast<const double2 *>(vtg + (size_t)i * 36 + 6 * d);
      const double2 v0 = vc[0], v1 = vc[1], v2 = vc[2];
      a[0] = v0.x; a[1] = v0.y; a[2] = v1.x; a[3] = v1.y; a[4] = v2.x; a[5] = v2.y;
    }
    {   // the W | bl rows of the pass are contiguous in the workspace: cnt * 42 doubles as 16-byte pieces
      const double2 *src = reinterpret_cast<const double2 *>(wbg + (size_t)pass * WV_ROWS * WV_WB);
      double2 *dst = reinterpret_cast<double2 *>(S.stage);
      for (int e = tid; e < cnt * (WV_WB / 2); e += WV_T) dst[e] = src[e];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) { x[k] = (k == d) ? 1.0 : 0.0; if (k == d) a[k] += lambda; }
    double *row = S.xch + t.row * WV_XROW;
    const int ok = w_solve6_cols(a, x, t, row);
    if (t.act && !ok) bad = 1;
    if (t.act) {
      double2 *vr = reinterpret_cast<double2 *>(vig + (size_t)t.i * 36 + 6 * d);
      vr[0] = make_double2(x[0], x[1]); vr[1] = make_double2(x[2], x[3]); vr[2] = make_double2(x[4], x[5]);
    }
    w_order();                                  // the staged rows are complete
    const double *srow = S.stage + t.row * WV_WB;
    double wvc[6], wv[6];
#pragma unroll
    for (int r = 0; r < 6; r++) { double s = 0; for (int k = 0; k < 6; k++) s += srow[6 * r + k] * x[k]; wvc[r] = s; }
    if (t.act) {
#pragma unroll
      for (int k = 0; k < 6; k++) row[6 * k + d] = wvc[k];
    }
    w_order_wave();
#pragma unroll
    for (int k = 0; k < 6; k++) wv[k] = row[6 * d + k];
    w_order_wave();
    {
      double u = 0, T[6];
#pragma unroll
      for (int k = 0; k < 6; k++) u += wv[k] * srow[36 + k];
#pragma unroll
      for (int j = 0; j < 6; j++) { double s2 = 0; for (int k = 0; k < 6; k++) s2 += wv[k] * srow[6 * j + k]; T[j] = s2; }
      if (t.act) {
#pragma unroll
        for (int j = 0; j < 6; j++) row[6 * d + j] = T[j];
        row[36 + d] = u;
      }
    }
    w_order();
    if (tid < 42) acc = r_stage_walk<true>(S.xch, tid, cnt, acc);
    w_order();
  }
  if (tid < 42) S.sg[tid] = acc;
  return bad;
}

// lf_match_backsub + the step's scale term: lane (i, a) computes component a of r = bl - W^T dp, reads row a of Vi from
// the columns w_eliminate left in the workspace, and writes component a of the trial landmark.
WV_PHASE void w_backsub(WaveShared &S, int n, double lambda, const double *wbg_, const double *vig_, const double *Lc_, double *Lt_) {
  n = w_uni(n); lambda = w_uni(lambda);
  const double *wbg = WV_G_RO(wbg_), *vig = WV_G_RO(vig_), *Lc = WV_G_RO(Lc_);
  double *Lt = WV_G_RW(Lt_), *red = S.red[0];
  double dp[6];
#pragma unroll
  for (int k = 0; k < 6; k++) dp[k] = S.dp[k];
  for (int pass = 0; pass * WV_ROWS < n; pass++) {
    const WvTask t = w_task(pass, n);
    const int i = t.act ? t.i : 0, a = t.d;
    const double *wb = wbg + (size_t)i * WV_WB, *vr = vig + (size_t)i * 36;
    double vi[6], wc[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { wc[k] = wb[6 * k + a]; vi[k] = vr[6 * k + a]; }
    const double bla = wb[36 + a], La0 = Lc[6 * i + a];
    double tt = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) tt += wc[k] * dp[k];
    const double ra = bla - tt;
    double dl = 0, s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) dl += vi[k] * __shfl(ra, t.base + k, 64);
    const double La = La0 + dl, term = dl * (lambda * dl + bla);
    if (t.act) Lt[6 * i + a] = La;
#pragma unroll
    for (int k = 0; k < 6; k++) s += __shfl(term, t.base + k, 64);
    if (t.act && a == 0) red[i] = s;
  }
}

// getTransformFromHybridMatchesG2O (transformation_estimation.cpp:218-461), line edges only; the sequential twin is
// oracle_refine_g2o.  set[0..n) = match indices (LDS); ws = the pair's workspace (its head: the compact measurements);
// the transform comes from and returns to S.tf.
// (inlined into the kernel: what it keeps across the phases' calls -- lambda, chi2, the counters -- is allocated above the phases' registers)
__device__ __forceinline__ void w_refine(WaveShared &S, double *ws, double wgt, double hd, int hub, int n, int iterations) {
  const wv_idx *set = S.set;
  const double *cm = ws;
  float *tf = S.tf;
  const int tid = threadIdx.x;
  double *vtg = ws + WV_OFF_VT, *wbg = ws + WV_OFF_WB, *vig = ws + WV_OFF_VI;
  double *Lg[2] = {ws + WV_OFF_L, ws + WV_OFF_L + LF_MAX_MATCHES * 6};
  // lambda, its growth factor and the current chi2 live in LDS between the phases (S.lm_*: every thread writes the same bits;
  // a write is separated from the other threads' reads of the old value by a barrier) -- kept in registers they would be
  // live across every call and spill
  int cur = 0;                                 // Lg[cur], S.X[cur]: the current state; the other set takes the trial step
  if (tid == 0) { float t0[16]; for (int i = 0; i < 16; i++) t0[i] = tf[i]; lf_se3 X0; lf_tf_to_older_pose(t0, &X0); S.X[0] = X0; }
  for (int i = tid; i < n; i += WV_T) {
    const double *c = cm + (size_t)set[i] * R_CM;    // nA | nB: the landmark starts at the newer camera's measurement
    for (int k = 0; k < 6; k++) Lg[0][6 * i + k] = c[k];
  }
  if (tid < 8) { S.red[0][n + tid] = 0.0; S.red[1][n + tid] = 0.0; }
  w_order();
  {
    double currentChi = 0;
    if (n > 0 && iterations > 0) {
      w_errchi(S, cm, n, 0, Lg[0], wgt, hd, hub, 0);
      w_order();
      currentChi = p_sum_published(S.red[0], n, 0.0);
    }
    S.lm_chi = currentChi; S.lm_lambda = 0; S.lm_ni = 2;
  }
  for (int it = 0; it < iterations && n > 0; it++) {
    int qmax = 0, again, stop;
    w_perturbed_poses(S, &S.X[cur]);
    w_order();
    w_blocks_n(S, cm, n, Lg[cur], vtg, wbg, wgt, hd, hub);
    const double mxl = w_blocks_o(S, cm, n, cur, Lg[cur], vtg, wbg, wgt, hd, hub);
    if (it == 0) {   // computeLambdaInit: tau * max |diagonal entry|
      double mx = mxl;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
#if WV_W > 1
      if (p_lane() == 0) S.wred[tid >> 6] = mx;
      w_order();
      mx = S.wred[0];
#pragma unroll
      for (int w = 1; w < WV_W; w++) { const double t = S.wred[w]; mx = t > mx ? t : mx; }
#else
      w_order();
#endif
#pragma unroll
      for (int i = 0; i < 6; i++) if (lf_fabs(S.hb[7 * i]) > mx) mx = lf_fabs(S.hb[7 * i]);
      S.lm_lambda = 1e-5 * mx;
      S.lm_ni = 2;
    } else w_order();
    do {
      double scale = 0, tempChi = DBL_MAX;
      // the oracle stops eliminating at the first failing match; any failure rejects the step
      int ok2 = w_uni(w_any(S, w_eliminate(S, n, S.lm_lambda, vtg, wbg, vig))) ? 0 : 1;     // (barrier: S.sg visible)
#if WV_W == 1
      w_order();
#endif
      if (ok2) {   // the pose system, one column per lane in every six-lane group alike (lf_solve6_u of the resident form: same operations)
        const WvTask t = w_task(0, WV_ROWS);
        double a[6], dp[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { a[k] = S.sg[6 * k + t.d]; dp[k] = S.sg[36 + k]; }
        ok2 = w_solve6_cols(a, dp, t, S.xch + t.row * WV_XROW);
        ok2 = w_uni(LF_ANY(ok2 == 0) ? 0 : 1);   // (uniform: every group of every wavefront holds the same system)
        if (ok2) {
          const double lambda = S.lm_lambda;
          if (tid == 0) {
            lf_se3 Xn; lf_se3_oplus(&S.X[cur], dp, &Xn); S.X[cur ^ 1] = Xn;
            for (int k = 0; k < 6; k++) S.dp[k] = dp[k];
          }
#pragma unroll
          for (int i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + S.hb[36 + i]);
          S.wscale[tid >> 6] = scale;              // (parked: the same value in every thread)
        }
      }
      if (ok2) {
        w_order();
        w_backsub(S, n, S.lm_lambda, wbg, vig, Lg[cur], Lg[cur ^ 1]);
        w_order();                               // the trial landmarks and the trial pose are complete
        w_errchi(S, cm, n, cur ^ 1, Lg[cur ^ 1], wgt, hd, hub, 1);
        w_order();
        tempChi = 0.0;
        scale = S.wscale[tid >> 6];
        p_sum2_published(S.red[0], S.red[1], n, &scale, &tempChi);
      }
      double lambda = S.lm_lambda, ni = S.lm_ni, currentChi = S.lm_chi;
      double rho = (currentChi - tempChi);
      scale += 1e-3;
      rho /= scale;
      int accepted = 0;
      if (rho > 0 && tempChi <= DBL_MAX && tempChi == tempChi) {
        double t = 2 * rho - 1, alpha = 1. - t * t * t, sf;
        if (alpha > 2. / 3.) alpha = 2. / 3.;
        sf = alpha > 1. / 3. ? alpha : 1. / 3.;
        lambda *= sf;
        ni = 2;
        currentChi = tempChi;
        accepted = 1;
      } else {
        lambda *= ni;
        ni *= 2;
      }
      qmax++;
      again = w_uni((rho < 0 && qmax < 10) ? 1 : 0);
      stop = w_uni((qmax == 10 || rho == 0) ? 1 : 0);
      accepted = w_uni(accepted);
      w_order();                                 // every thread has read the old state
      S.lm_lambda = lambda; S.lm_ni = ni; S.lm_chi = currentChi;
      if (accepted) cur ^= 1;
    } while (again);
    if (stop) break;
  }
  if (tid == 0) { float t1[16]; lf_older_pose_to_tf(&S.X[cur], t1); for (int i = 0; i < 16; i++) tf[i] = t1[i]; }
  w_order();                                   // (the next refinement overwrites S.X)
}

// inlier scan of all matches with S.tf; returns count, fills set[] (ascending) and leaves the two sums the reference
// accumulates (motion.cpp:688-699 / 795-812) in S.sse_f / S.sse_d: 64 WV_W matches per trip.
WV_PHASE int w_score(WaveShared &S, const double *cm_, int nLn, double thr, int to_idx) {
  nLn = w_uni(nLn); thr = w_uni(thr); to_idx = w_uni(to_idx);
  wv_idx *set = to_idx ? S.idx : S.set;
  const double *cm = WV_G_RO(cm_);
  const int tid = threadIdx.x, w = tid >> 6;
  float tf[16];
#pragma unroll
  for (int k = 0; k < 16; k++) tf[k] = S.tf[k];
  int base = 0;
  for (int c0 = 0; c0 < nLn; c0 += WV_T) {      // (the trips cover the list rounded up to WV_T >= rounded up to 8: every term the sums read is written)
    const int i = c0 + tid;
    bool in = false;
    double add = 0;
    if (i < nLn) {
      const double *c = cm + (size_t)i * R_CM;
      in = lf_line_inlier(tf, c, c + 3, c + 24, c + 27, c + 30, c + 39, thr, &add);
    }
    const u64 msk = __ballot(in);
    if (i < LF_MAX_MATCHES + 8) S.red[0][i] = add;   // 0.0 for a non-inlier and beyond the list: adding it changes neither sum
    int wb = 0, tot = __popcll(msk);
#if WV_W > 1
    if (p_lane() == 0) S.wcnt[w] = tot;
    w_order();
    tot = 0;
#pragma unroll
    for (int k = 0; k < WV_W; k++) { const int cc = S.wcnt[k]; if (k < w) wb += cc; tot += cc; }
#endif
    if (in) set[base + wb + __popcll(msk & p_lt())] = (wv_idx)i;
    base += tot;
#if WV_W > 1
    w_order();                                   // (S.wcnt is written again by the next trip)
#endif
  }
  w_order();
  float sse = 0;      // `float sse` of the RANSAC loop (motion.cpp:666)
  double sse_d = 0;   // `double tmp_sse` of the re-scoring loop (motion.cpp:778)
  const int n8 = (nLn + 7) & ~7;
  for (int l = 0; l < n8; l += 8) {
    double q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = S.red[0][l + k];
#pragma unroll
    for (int k = 0; k < 8; k++) { sse += q[k]; sse_d += q[k]; }
  }
  if (tid == 0) { S.sse_f = sse; S.sse_d = sse_d; }
  w_order();
  return base;
}
