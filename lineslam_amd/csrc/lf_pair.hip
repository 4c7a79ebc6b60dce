// lf_pair.hip -- pair solver for gfx950: line matching + relative pose for a batch of node pairs.
//
//   k_match   Node::lineMatching (src/node.cpp:1619-1694): all-pairs gated descriptor distances
//             (one 256-thread block per pair; N1 x N2 x 72 fp64, VALU -- no MFMA: N ~ 10^2),
//             mutual nearest neighbour + ratio tests, ordered emission.
//   k_pose    ONE 256-THREAD WORKGROUP PER PAIR:
//             getTransform_PtsLines_ransac (src/line/motion.cpp:605-849) with line matches: the sample
//             sequence is generated serially (partial Fisher-Yates state carries over, :635-658), the
//             500 hypotheses are solved and scored ONE PER THREAD, the winner is the arg-max of the
//             inlier count with the lowest iteration on ties (the sequential "strictly greater" rule,
//             :714-720), found with wavefront shuffles + LDS;
//             getTransformFromHybridMatchesG2O (src/transformation_estimation.cpp:218-461): LM with six
//             lanes per match (columns / rows of the 6x6 blocks), pose system accumulated in match
//             order by one accumulator thread per matrix entry (bit-identical to the sequential oracle).
#include "lf_pair.h"
#include "lf_pose.h"
#include <float.h>
#include <stdlib.h>
#include "lf_pose_wg.h"
#include "lf_pose_res.h"
#include "lf_pose_wave.h"

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

// The descDiff matrix of the reference (node.cpp:1643-1654) is never materialised: an entry is either 100 (a gate
// failed) or the descriptor distance of a LIVE pair (all three gates passed), and everything lineMatching reads off
// the matrix afterwards -- first minimum of a row / column, second smallest value of a row / column, both started
// at 100 -- is a function of the live entries below 100 alone:
//   * a live value >= 100 can neither be a minimum that passes descDiffThresh (< 1) nor lower a second-best that
//     starts at 100: it is dropped;
//   * a NaN distance (the unguarded sqrt of computeMSLD, utils.cpp:1593) is skipped by every `<` of the scans except
//     where a scan STARTS on it: D[i][0] = NaN kills row i (minMaxLoc keeps NaN), D[0][j] = NaN kills column j
//     (minV stays NaN, minP = 0, and row 0 cannot have its minimum there).
// The 2D members of both line maps are staged once in LDS (SoA), the direction gate runs on all n1 x n2 pairs, the
// pairs that pass are queued and take the expensive gates + the 72-d distance on densely packed lanes; live entries
// go to a small list (L2 resident) and to LDS atomic minima (fp64 bit patterns of non-negative values order like
// unsigned integers).  Two more sweeps over the list give the first arg-min (smallest index among ties, as
// cv::minMaxLoc / the strict `<` of node.cpp:1664-1669) and the second-best values.
// Two forms of the kernel, same arithmetic, same results:
//   light (round 6)  256 threads, 78 registers, 42 KB of LDS: only the direction members r0 r1 are staged, the survivors of the
//                    direction gate read p, q, lineEq2d from the records (L2) -- a workgroup that fits beside the front end's
//                    wavefronts; used for the big batches that run inside a pipelined step (>= LF_MATCH_LIGHT_PAIRS pairs + pose)
//   heavy (rounds 3-5) 512 threads, 172 registers, all nine 2D members staged (103 KB): faster when the launch has the chip to
//                    itself (BASELINE config 4: 256 pairs in 0.38 ms against 0.66), a near-empty-CU requirement in a full chip
#define LF_MATCH_LIGHT_PAIRS 1000
#define LF_D_INF 0x7ff0000000000000ull    // +inf: "no live entry"
template <bool LIGHT> struct MatchCfg {
  static constexpr int N = LIGHT ? 256 : 512;          // threads per pair
  static constexpr int W = N / 64;
  static constexpr int CHUNK = LIGHT ? 1024 : 2048;    // (query, train) pairs per direction-gate round
  static constexpr int D2 = LIGHT ? 2 : 9;             // staged per line: r0 r1 | p0 p1 q0 q1 l0 l1 l2 r0 r1
  static constexpr int R0 = LIGHT ? 0 : 7;
};
template <bool LIGHT> struct MatchSharedT {
  double q2d[MatchCfg<LIGHT>::D2][LF_MATCH_LINE_CAP];   // (p0 p1 q0 q1 l0 l1 l2) r0 r1 of the query lines
  double t2d[MatchCfg<LIGHT>::D2][LF_MATCH_LINE_CAP];   //                            ... of the train lines
  unsigned long long rmin[LF_MATCH_LINE_CAP], cmin[LF_MATCH_LINE_CAP], rmin2[LF_MATCH_LINE_CAP], cmin2[LF_MATCH_LINE_CAP];
  int rarg[LF_MATCH_LINE_CAP], carg[LF_MATCH_LINE_CAP];
  unsigned char rdead[LF_MATCH_LINE_CAP], cdead[LF_MATCH_LINE_CAP];
  int queue[MatchCfg<LIGHT>::CHUNK], qn, nlive, wbase[MatchCfg<LIGHT>::W];
};
__device__ __forceinline__ double m_pl(double px, double py, double l0, double l1, double l2) {   // pt_to_line_dist2d
  return lf_fabs((l0 * px + l1 * py + l2)) / lf_sqrt(l0 * l0 + l1 * l1);
}
__device__ __forceinline__ double m_n2(double ax, double ay, double bx, double by) {
  return lf_sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
}
__device__ __forceinline__ double m_pr(double Xx, double Xy, double Ax, double Ay, double Bx, double By) {   // projectPt2d_to_line2d
  double BX0 = Xx - Bx, BX1 = Xy - By, BA0 = Ax - Bx, BA1 = Ay - By;
  double n = lf_sqrt(BA0 * BA0 + BA1 * BA1);
  return (BX0 * BA0 + BX1 * BA1) / n / n;
}
// lineSegmentOverlap(a, b), utils.cpp:1620-1638, on (p, q) of both lines
__device__ __forceinline__ double m_ov(const double *a, const double *b) {
  if (m_n2(a[0], a[1], a[2], a[3]) < m_n2(b[0], b[1], b[2], b[3])) {
    double lp = m_pr(a[0], a[1], b[0], b[1], b[2], b[3]), lq = m_pr(a[2], a[3], b[0], b[1], b[2], b[3]);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return lf_fabs(lp - lq) * m_n2(b[0], b[1], b[2], b[3]);
  } else {
    double lp = m_pr(b[0], b[1], a[0], a[1], a[2], a[3]), lq = m_pr(b[2], b[3], a[0], a[1], a[2], a[3]);
    if ((lp < 0 && lq < 0) || (lp > 1 && lq > 1)) return -1;
    return lf_fabs(lp - lq) * m_n2(a[0], a[1], a[2], a[3]);
  }
}
__device__ __forceinline__ bool m_adjacent(const PairConsts &c, const PairBuffers &b, int pr, int fq, int ft) {
  if (b.adjacent && b.adjacent[pr] != 255) return b.adjacent[pr] != 0;   // Node::lineMatching's adjacentFrame argument
  long long idd = (long long)b.frame_ids[fq] - (long long)b.frame_ids_t[ft];
  if (idd < 0) idd = -idd;
  return !(idd > c.P.adjacent_linematch_window);                           // as matchNodePair passes it, node.cpp:1505-1507
}

template <bool LIGHT>
__global__ void __launch_bounds__(MatchCfg<LIGHT>::N) k_match(PairConsts c, PairBuffers b) {
  constexpr int MT_N = MatchCfg<LIGHT>::N, MT_W = MatchCfg<LIGHT>::W, MT_CHUNK = MatchCfg<LIGHT>::CHUNK, MT_R0 = MatchCfg<LIGHT>::R0;
  __shared__ MatchSharedT<LIGHT> S;
  const int pr = blockIdx.x, tid = threadIdx.x;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  int n1 = b.nlines[fq], n2 = b.nlines_t[ft];
  if (n1 > c.line_cap) n1 = c.line_cap;
  if (n2 > b.line_cap_t) n2 = b.line_cap_t;
  if (n2 > c.line_cap) n2 = c.line_cap;
  const lf_line_record *f1 = b.recs + (size_t)fq * c.line_cap, *f2 = b.recs_t + (size_t)ft * b.line_cap_t;
  unsigned *live_idx = b.live_idx + (size_t)pr * c.line_cap * c.line_cap;
  double *live_val = b.live_val + (size_t)pr * c.line_cap * c.line_cap;
  const bool adjacent = m_adjacent(c, b, pr, fq, ft);
  const double lineDistThresh = adjacent ? 45 : 80, descDiffThresh = adjacent ? 0.85 : 0.7;   // node.cpp:1623-1635
  const double lineOverlapThresh = adjacent ? 0 : -1, ratio = 0.7;
  if (n1 == 0 || n2 == 0) { if (tid == 0) b.nmatches[pr] = 0; return; }
  for (int l = tid; l < n1; l += MT_N) {
    const lf_line_record *a = &f1[l];
    if constexpr (!LIGHT) {
    S.q2d[0][l] = a->p[0]; S.q2d[1][l] = a->p[1]; S.q2d[2][l] = a->q[0]; S.q2d[3][l] = a->q[1];
    S.q2d[4][l] = a->lineEq2d[0]; S.q2d[5][l] = a->lineEq2d[1]; S.q2d[6][l] = a->lineEq2d[2];
    }
    S.q2d[MT_R0][l] = a->r[0]; S.q2d[MT_R0 + 1][l] = a->r[1];
    S.rmin[l] = LF_D_INF; S.rarg[l] = 0x7fffffff; S.rmin2[l] = 0x4059000000000000ull /* 100.0 */; S.rdead[l] = 0;
  }
  for (int l = tid; l < n2; l += MT_N) {
    const lf_line_record *a = &f2[l];
    if constexpr (!LIGHT) {
    S.t2d[0][l] = a->p[0]; S.t2d[1][l] = a->p[1]; S.t2d[2][l] = a->q[0]; S.t2d[3][l] = a->q[1];
    S.t2d[4][l] = a->lineEq2d[0]; S.t2d[5][l] = a->lineEq2d[1]; S.t2d[6][l] = a->lineEq2d[2];
    }
    S.t2d[MT_R0][l] = a->r[0]; S.t2d[MT_R0 + 1][l] = a->r[1];
    S.cmin[l] = LF_D_INF; S.carg[l] = 0x7fffffff; S.cmin2[l] = 0x4059000000000000ull; S.cdead[l] = 0;
  }
  if (tid == 0) S.nlive = 0;
  __syncthreads();
  for (int base = 0; base < n1 * n2; base += MT_CHUNK) {
    if (tid == 0) S.qn = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MT_CHUNK / MT_N; r++) {
      const int idx = base + r * MT_N + tid;
      if (idx < n1 * n2) {
        const int i = idx / n2, j = idx - i * n2;
        if (S.q2d[MT_R0][i] * S.t2d[MT_R0][j] + S.q2d[MT_R0 + 1][i] * S.t2d[MT_R0 + 1][j] > c.cos_angle_thresh) S.queue[atomicAdd(&S.qn, 1)] = idx;
      }
    }
    __syncthreads();
    const int qn = S.qn;
    for (int k = tid; k < qn; k += MT_N) {
      const int idx = S.queue[k], i = idx / n2, j = idx - i * n2;
      double a[7], t[7];
      if constexpr (LIGHT) {   // p0 p1 q0 q1 l0 l1 l2 of both lines, from their records (L2: the survivors are a fraction of the n1 n2 pairs)
        const lf_line_record *ra = &f1[i], *rt = &f2[j];
        a[0] = ra->p[0]; a[1] = ra->p[1]; a[2] = ra->q[0]; a[3] = ra->q[1]; a[4] = ra->lineEq2d[0]; a[5] = ra->lineEq2d[1]; a[6] = ra->lineEq2d[2];
        t[0] = rt->p[0]; t[1] = rt->p[1]; t[2] = rt->q[0]; t[3] = rt->q[1]; t[4] = rt->lineEq2d[0]; t[5] = rt->lineEq2d[1]; t[6] = rt->lineEq2d[2];
      } else {
#pragma unroll
        for (int e = 0; e < 7; e++) { a[e] = S.q2d[e][i]; t[e] = S.t2d[e][j]; }
      }
      if ((0.25 * m_pl(a[0], a[1], t[4], t[5], t[6]) + 0.25 * m_pl(a[2], a[3], t[4], t[5], t[6]) +
           0.25 * m_pl(t[0], t[1], a[4], a[5], a[6]) + 0.25 * m_pl(t[2], t[3], a[4], a[5], a[6]) < lineDistThresh) &&
          (m_ov(a, t) > lineOverlapThresh)) {
        const double *da = f1[i].des, *db = f2[j].des;
        double s = 0;
        // cv::norm(des_i - des_j) as OpenCV 2.4 sums it (normL2Sqr_, modules/core/src/stat.cpp): four squares per trip,
        // s += ((v0 v0 + v1 v1) + v2 v2) + v3 v3.  Eight loads of each side in flight.
        for (int kk = 0; kk < 72; kk += 8) {
          double x[8], y[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { x[u] = da[kk + u]; y[u] = db[kk + u]; }
#pragma unroll
          for (int u = 0; u < 8; u += 4) {
            const double d0 = x[u] - y[u], d1 = x[u + 1] - y[u + 1], d2 = x[u + 2] - y[u + 2], d3 = x[u + 3] - y[u + 3];
            s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          }
        }
        const double v = lf_sqrt(s);
        if (v != v) { if (j == 0) S.rdead[i] = 1; if (i == 0) S.cdead[j] = 1; }
        else if (v < 100) {
          const int o = atomicAdd(&S.nlive, 1);
          live_idx[o] = ((unsigned)i << 16) | (unsigned)j;
          live_val[o] = v;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
          atomicMin(&S.rmin[i], bits);
          atomicMin(&S.cmin[j], bits);
        }
      }
    }
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  const int nlive = S.nlive;
  for (int k = tid; k < nlive; k += MT_N) {      // first arg-min of every row / column
    const unsigned e = live_idx[k];
    const int i = (int)(e >> 16), j = (int)(e & 0xffffu);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(live_val[k]);
    if (bits == S.rmin[i]) atomicMin(&S.rarg[i], j);
    if (bits == S.cmin[j]) atomicMin(&S.carg[j], i);
  }
  __syncthreads();
  for (int k = tid; k < nlive; k += MT_N) {      // second-best of every row / column (the entry AT the arg-min is excluded)
    const unsigned e = live_idx[k];
    const int i = (int)(e >> 16), j = (int)(e & 0xffffu);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(live_val[k]);
    if (j != S.rarg[i]) atomicMin(&S.rmin2[i], bits);
    if (i != S.carg[j]) atomicMin(&S.cmin2[j], bits);
  }
  __syncthreads();
  // decision per query line (node.cpp:1656-1690) and ordered emission (the reference loops over i ascending)
  int *mq = b.match_q + (size_t)pr * c.match_cap, *mt = b.match_t + (size_t)pr * c.match_cap;
  double *md = b.match_d + (size_t)pr * c.match_cap;
  int base = 0;
  const int wave = tid >> 6, lane = tid & 63;
  for (int i0 = 0; i0 < n1; i0 += MT_N) {
    const int i = i0 + tid;
    bool has = false;
    int minPos = 0;
    double minVal = 0;
    if (i < n1 && !S.rdead[i] && S.rmin[i] != LF_D_INF) {
      minVal = __longlong_as_double((long long)S.rmin[i]);
      minPos = S.rarg[i];
      if (minVal < descDiffThresh && !S.cdead[minPos] && S.carg[minPos] == i) {
        const double rowmin2 = __longlong_as_double((long long)S.rmin2[i]), colmin2 = __longlong_as_double((long long)S.cmin2[minPos]);
        has = rowmin2 * ratio > minVal && colmin2 * ratio > minVal;
      }
    }
    const u64 m = __ballot(has);
    if (lane == 0) S.wbase[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MT_W; w++) { const int cw = S.wbase[w]; if (w < wave) off += cw; tot += cw; }
    if (has) {
      const int o = base + off + __popcll(m & p_lt());
      if (o < c.match_cap) { mq[o] = i; mt[o] = minPos; md[o] = minVal; }
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) b.nmatches[pr] = base;
}

// The dense descDiff matrix itself, for parity tests only (lf_pair_get_descdiff): one entry per thread.
__global__ void __launch_bounds__(256) k_descdiff(PairConsts c, PairBuffers b, int pr, double *D) {
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  int n1 = b.nlines[fq], n2 = b.nlines_t[ft];
  if (n1 > c.line_cap) n1 = c.line_cap;
  if (n2 > b.line_cap_t) n2 = b.line_cap_t;
  if (n2 > c.line_cap) n2 = c.line_cap;
  const lf_line_record *f1 = b.recs + (size_t)fq * c.line_cap, *f2 = b.recs_t + (size_t)ft * b.line_cap_t;
  const bool adjacent = m_adjacent(c, b, pr, fq, ft);
  const double lineDistThresh = adjacent ? 45 : 80, lineOverlapThresh = adjacent ? 0 : -1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n1 * n2) return;
  const int i = idx / n2, j = idx - i * n2;
  const lf_line_record *a = &f1[i], *bb = &f2[j];
  double v = 100;
  if ((a->r[0] * bb->r[0] + a->r[1] * bb->r[1] > c.cos_angle_thresh) &&
      (0.25 * m_pt_line2d(a->p, bb->lineEq2d) + 0.25 * m_pt_line2d(a->q, bb->lineEq2d) +
       0.25 * m_pt_line2d(bb->p, a->lineEq2d) + 0.25 * m_pt_line2d(bb->q, a->lineEq2d) < lineDistThresh) &&
      (m_overlap(a, bb) > lineOverlapThresh)) {
    double s = 0;
    for (int kk = 0; kk < 72; kk += 4) {    // OpenCV's normL2Sqr_ order (four squares per trip)
      const double d0 = a->des[kk] - bb->des[kk], d1 = a->des[kk + 1] - bb->des[kk + 1], d2 = a->des[kk + 2] - bb->des[kk + 2],
                   d3 = a->des[kk + 3] - bb->des[kk + 3];
      s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    v = lf_sqrt(s);
  }
  D[idx] = v;
}
void lf_pair_descdiff_launch(const PairConsts &c, const PairBuffers &b, int pair, double *D, hipStream_t st) {
  const int blocks = (c.line_cap * c.line_cap + 255) / 256;
  hipLaunchKernelGGL(k_descdiff, dim3(blocks), dim3(256), 0, st, c, b, pair, D);
}

// ------------------------------------------------------------------------------ k_pose
struct PoseShared {
  int idx[LF_MAX_MATCHES];        // scratch inlier list of the re-scoring loop
  int set[LF_MAX_MATCHES];        // current inlier list (indices into the match list)
  ResShared rs;
};
// k_ransac -> k_pose: the winner of the hypothesis stage, five ints behind the pair's compact measurements
#define R_WIN_OFF (LF_MAX_MATCHES * R_CM)
static_assert(R_WIN_OFF + 4 <= LF_PAIR_WS_DOUBLES, "the winner record fits the pair's workspace");
struct RansacShared {
  int idx[LF_MAX_MATCHES];
  unsigned char smp[LF_RANSAC_MAX_ITERS * 3];
  int wcnt[16], wit[16];
};

// getTransformFromHybridMatchesG2O (transformation_estimation.cpp:218-461), line edges only; the
// sequential twin is oracle_refine_g2o.  set[0..n) = match indices (LDS); cm = the pair's compact measurements.
__device__ void r_refine(ResShared &S, const double *cm, const lf_params &P, const int *set, int n, float *tf, int iterations) {
  const int tid = threadIdx.x;
  const double wgt = P.g2o_line_error_weight, hd = P.g2o_BA_kernel_delta;
  const int hub = P.g2o_BA_use_kernel;
  ResRegs R;
  double lambda = 0, ni = 2, currentChi = 0;
  int cur = 0;                                 // S.L[cur], S.X[cur]: the current state; the other set takes the trial step
  if (tid == 0) { lf_se3 X0; lf_tf_to_older_pose(tf, &X0); S.X[0] = X0; }
  if (tid < n) {
    const double *c = cm + (size_t)set[tid] * R_CM;    // nA | nB: the landmark starts at the newer camera's measurement
    for (int k = 0; k < 6; k++) S.L[0][6 * tid + k] = c[k];
  }
  if (tid < 8) { S.red[0][n + tid] = 0.0; S.red[1][n + tid] = 0.0; }
  __syncthreads();
  if (n > 0 && iterations > 0) {
    r_errchi(cm, set, n, &S.X[0], S.L[0], wgt, hd, hub, S.red[0]);
    __syncthreads();
    currentChi = p_sum_published(S.red[0], n, 0.0);
  }
  for (int it = 0; it < iterations && n > 0; it++) {
    double rho = 0, tempChi;
    int qmax = 0;
    double mxl = 0;
    PT(0);
    r_perturbed_poses(S, &S.X[cur]);
    __syncthreads();
    r_blocks(S, cm, set, n, &S.X[cur], S.L[cur], R, wgt, hd, hub, &mxl);
#ifdef LF_POSE_PROFILE
    if (blockIdx.x == 7 && threadIdx.x == 0) g_pprof[15]++;
#endif
    PT(5);
    if (it == 0) {   // computeLambdaInit: tau * max |diagonal entry|
      double mx = r_block_max(S, mxl);           // (barrier inside: hb visible)
#pragma unroll
      for (int i = 0; i < 6; i++) if (lf_fabs(S.hb[7 * i]) > mx) mx = lf_fabs(S.hb[7 * i]);
      lambda = 1e-5 * mx;
      ni = 2;
    } else __syncthreads();
    PT(6);
    do {
      double dp[6], scale = 0;
      // the oracle stops eliminating at the first failing match; any failure rejects the step
      int ok2 = __syncthreads_or(r_eliminate(S, n, lambda, R)) ? 0 : 1;   // (barrier: S.sg visible)
#ifdef LF_POSE_PROFILE
      if (blockIdx.x == 7 && threadIdx.x == 0) g_pprof[14]++;
#endif
      PT(7);
      if (ok2) {
        double A[36];
#pragma unroll
        for (int i = 0; i < 36; i++) A[i] = S.sg[i];
#pragma unroll
        for (int i = 0; i < 6; i++) dp[i] = S.sg[36 + i];
        ok2 = lf_solve6_u(A, dp, 1);   // the pose system is the same in every thread: scalar pivot branches
      }
      PT(9);
      tempChi = DBL_MAX;
      if (ok2) {
        if (tid == 0) { lf_se3 Xn; lf_se3_oplus(&S.X[cur], dp, &Xn); S.X[cur ^ 1] = Xn; }
#pragma unroll
        for (int i = 0; i < 6; i++) scale += dp[i] * (lambda * dp[i] + S.hb[36 + i]);
        r_backsub(S, n, dp, lambda, S.L[cur], S.L[cur ^ 1], R, S.red[0]);
        __syncthreads();                       // the trial landmarks are complete
        r_errchi(cm, set, n, &S.X[cur ^ 1], S.L[cur ^ 1], wgt, hd, hub, S.red[1]);
        __syncthreads();
        tempChi = 0.0;
        p_sum2_published(S.red[0], S.red[1], n, &scale, &tempChi);
      }
      PT(10);
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
        cur ^= 1;
      } else {
        lambda *= ni;
        ni *= 2;
      }
      __syncthreads();
      PT(11);
      qmax++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0) break;
  }
  lf_older_pose_to_tf(&S.X[cur], tf);
  __syncthreads();                             // (the next refinement overwrites S.X)
}

// inlier scan of all matches with tf; returns count, fills set[] (ascending) and the float sse the
// reference accumulates (motion.cpp:688-699 / 795-812)
__device__ int r_score(ResShared &S, const double *cm, int nLn, const float *tf, double thr, int *set, float *sse_out,
                       double *sse_d_out) {
  const int tid = threadIdx.x, w = tid >> 6;
  bool in = false;
  double add = 0;
  if (tid < nLn) {
    const double *c = cm + (size_t)tid * R_CM;
    in = lf_line_inlier(tf, c, c + 3, c + 24, c + 27, c + 30, c + 39, thr, &add);
  }
  const u64 msk = __ballot(in);
  if (p_lane() == 0) S.wcnt[w] = __popcll(msk);
  p_publish(S.red[0], add, nLn);       // 0.0 for a non-inlier: adding it changes neither sum
  __syncthreads();
  int base = 0, cnt = 0;
#pragma unroll
  for (int k = 0; k < RW_N; k++) { int c = S.wcnt[k]; if (k < w) base += c; cnt += c; }
  if (in) set[base + __popcll(msk & p_lt())] = tid;
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
  *sse_out = sse;
  *sse_d_out = sse_d;
  __syncthreads();
  return cnt;
}

// the three-line model of RANSAC iteration `it` (sample table in LDS) as the float matrix the reference scores with
__device__ __forceinline__ int r_model(const int *smp3, const double *cm, float *tf) {
  double la[18], lb[18], R[9], t[3];
  for (int s = 0; s < 3; s++) {
    const double *c = cm + (size_t)smp3[s] * R_CM;
    for (int cc = 0; cc < 3; cc++) { la[6 * s + cc] = c[cc]; la[6 * s + 3 + cc] = c[3 + cc]; lb[6 * s + cc] = c[24 + cc]; lb[6 * s + 3 + cc] = c[27 + cc]; }
  }
  if (!lf_rel_motion_lines(la, lb, 3, R, t)) return 0;
  for (int i = 0; i < 3; i++) { for (int cc = 0; cc < 3; cc++) tf[4 * i + cc] = (float)R[3 * i + cc]; tf[4 * i + 3] = (float)t[i]; }
  tf[12] = tf[13] = tf[14] = 0.0f; tf[15] = 1.0f;
  return 1;
}

// the gates of computeRelativeMotion_Ransac before the hypothesis loop (motion.cpp:621-633): shared by k_ransac and k_pose
struct PoseGate { int nLn, n_all, min_inlier, lw, maxIter; bool go; long long id_t, id_q; };
__device__ __forceinline__ PoseGate r_gate(const PairConsts &c, const PairBuffers &b, int pr, int fq, int ft) {
  PoseGate g;
  const lf_params &P = c.P;
  g.nLn = b.nmatches[pr];
  g.n_all = g.nLn;
  if (g.nLn > c.match_cap) g.nLn = c.match_cap;
  if (g.nLn > LF_MAX_MATCHES) g.nLn = LF_MAX_MATCHES;
  g.id_t = (long long)b.frame_ids_t[ft]; g.id_q = (long long)b.frame_ids[fq];
  g.min_inlier = P.min_feature_matches; g.lw = P.line_match_number_weight; g.maxIter = P.ransac_iters_line_motion;
  if (g.maxIter > LF_RANSAC_MAX_ITERS) g.maxIter = LF_RANSAC_MAX_ITERS;
  g.go = !(0 + g.nLn * g.lw < g.min_inlier);                                                          // motion.cpp:621-624
  if (g.min_inlier > 0.7 * (0 + g.nLn * g.lw)) g.min_inlier = (int)(0.7 * (0 + g.nLn * g.lw));        // :626-628
  { long long d = g.id_t - g.id_q; if (d < 0) d = -d; if (d > 50) g.min_inlier = P.min_matches_loopclose; }   // :631-633
  if (g.nLn < 3) g.go = false;
  return g;
}

// ------------------------------------------------------------------------------ k_ransac
// The hypothesis stage of computeRelativeMotion_Ransac (motion.cpp:635-723) as its own launch: one workgroup per pair, one
// three-line hypothesis per thread (500 by the launch file), each scored against every match.  It needs no LDS beyond the
// sample table and ~1/4 of the registers of the refinement, so several of its wavefronts share a SIMD with each other and
// with the front end's kernels, where inside k_pose (one wavefront per SIMD, the CU's whole register file) its dependent
// fp64 chains ran alone.  It also lays down the pair's compact measurements; k_pose picks both up from the pair's workspace.
#ifndef RS_N
#define RS_N 256
#endif
#ifdef RS_WAVES
#define RS_BOUNDS __launch_bounds__(RS_N, RS_WAVES)
#else
#define RS_BOUNDS __launch_bounds__(RS_N)
#endif
__global__ void RS_BOUNDS k_ransac(PairConsts c, PairBuffers b) {
  __shared__ RansacShared S;
  const int pr = blockIdx.x, tid = threadIdx.x, lane = p_lane();
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  const lf_line_record *train = b.recs_t + (size_t)ft * b.line_cap_t, *query = b.recs + (size_t)fq * c.line_cap;
  const int *mq = b.match_q + (size_t)pr * c.match_cap, *mt = b.match_t + (size_t)pr * c.match_cap;
  double *cm = b.ws + (size_t)pr * LF_PAIR_WS_DOUBLES;      // [nLn][R_CM]: the only workspace of the pair
  int *win = (int *)(cm + R_WIN_OFF);
  const lf_params &P = c.P;
  const PoseGate g = r_gate(c, b, pr, fq, ft);
  const int nLn = g.nLn, maxIter = g.maxIter;
  const double thr = P.max_mah_dist_for_inliers;
  if (!g.go) { if (tid == 0) { win[0] = -1; win[1] = 0; } return; }
  // ---- the measurements of the matched lines, 48 contiguous doubles per match (read by every later phase)
  for (int e = tid; e < nLn * 2; e += RS_N) {
    const int k = e >> 1, h = e & 1;
    const lf_line_record *r = h ? &train[mt[k]] : &query[mq[k]];
    double *o = cm + (size_t)k * R_CM + 24 * h;
    for (int j = 0; j < 3; j++) { o[j] = r->A[j]; o[3 + j] = r->B[j]; }
    for (int j = 0; j < 9; j++) { o[6 + j] = r->DUa[j]; o[15 + j] = r->DUb[j]; }
  }
  // ---- sample sequence (serial; partial Fisher-Yates state carries over, :635-658)
  for (int i = tid; i < nLn; i += RS_N) S.idx[i] = i;
  __syncthreads();
  if (tid == 0) {
    const uint64_t stream = LF_STREAM_PAIR((uint64_t)g.id_q, (uint64_t)g.id_t);
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
  __threadfence_block();
  __syncthreads();                              // (also: the workgroup's own cm writes are visible to it)
  // ---- one hypothesis per thread
  int my_cnt = -1, my_it = 1 << 30;
  for (int it = tid; it < maxIter; it += RS_N) {
    float tf[16];
    const int s3[3] = {S.smp[3 * it], S.smp[3 * it + 1], S.smp[3 * it + 2]};
    if (!r_model(s3, cm, tf)) continue;
    int nc = 0;
    for (int i = 0; i < nLn; ++i) {
      double add;
      const double *m = cm + (size_t)i * R_CM;
      nc += lf_line_inlier(tf, m, m + 3, m + 24, m + 27, m + 30, m + 39, thr, &add);
    }
    if (nc > my_cnt) { my_cnt = nc; my_it = it; }   // strictly greater: earliest iteration wins inside a thread
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {   // arg-max: count desc, iteration asc -- wavefront, then the wavefronts
    int oc = __shfl_xor(my_cnt, o, 64), oi = __shfl_xor(my_it, o, 64);
    if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
  }
  if (lane == 0) { S.wcnt[tid >> 6] = my_cnt; S.wit[tid >> 6] = my_it; }
  __syncthreads();
  if (tid == 0) {
    my_cnt = S.wcnt[0]; my_it = S.wit[0];
    for (int w = 1; w < RS_N / 64; w++) {
      int oc = S.wcnt[w], oi = S.wit[w];
      if (oc > my_cnt || (oc == my_cnt && oi < my_it)) { my_cnt = oc; my_it = oi; }
    }
    const int best = my_cnt > 0 ? my_it : -1;
    win[0] = best; win[1] = my_cnt > 0 ? my_cnt : 0;
    for (int s = 0; s < 3; s++) win[2 + s] = best >= 0 ? S.smp[3 * best + s] : 0;
    if (best >= 0) {   // the winning model once more, for the wavefront-per-pair refinement (it re-scores it; r_model needs ~190 registers)
      float tf[16], *wtf = (float *)(cm + WV_OFF_TF);
      const int s3[3] = {S.smp[3 * best], S.smp[3 * best + 1], S.smp[3 * best + 2]};
      r_model(s3, cm, tf);
      for (int i = 0; i < 16; i++) wtf[i] = tf[i];
    }
  }
}

#ifndef LF_POSE_PRIO
#define LF_POSE_PRIO 2      // wave issue priority (s_setprio): latency-bound at one wavefront per SIMD
#endif
#ifdef LF_POSE_WAVES        // register budget for LF_POSE_WAVES wavefronts per SIMD; the LDS block becomes dynamic so that the
#define LF_POSE_ATTR __attribute__((amdgpu_waves_per_eu(LF_POSE_WAVES, LF_POSE_WAVES)))   // compiler does not widen it back
#else
#define LF_POSE_ATTR
#endif
// The refinement stage (motion.cpp:725-839): the winner of k_ransac re-scored, getTransformFromHybridMatchesG2O on its inliers,
// and the re-scoring loop -- resident, one workgroup per pair and per CU (lf_pose_res.h).
__global__ void __launch_bounds__(RT_N) LF_POSE_ATTR k_pose(PairConsts c, PairBuffers b) {
  __builtin_amdgcn_s_setprio(LF_POSE_PRIO);
#ifdef LF_POSE_WAVES
  extern __shared__ double s_pose_dyn[];
  PoseShared &S = *reinterpret_cast<PoseShared *>(s_pose_dyn);
#else
  __shared__ PoseShared S;
#endif
  const int pr = blockIdx.x, tid = threadIdx.x;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  lf_pair_result *res = b.results + pr;
  const double *cm = b.ws + (size_t)pr * LF_PAIR_WS_DOUBLES;      // [nLn][R_CM], laid down by k_ransac
  const int *win = (const int *)(cm + R_WIN_OFF);
  const lf_params &P = c.P;
  const PoseGate g = r_gate(c, b, pr, fq, ft);
  const int nLn = g.nLn, n_all = g.n_all, lw = g.lw, min_inlier = g.min_inlier;
  const long long id_t = g.id_t, id_q = g.id_q;
  float tf_out[16];
#pragma unroll
  for (int i = 0; i < 16; i++) tf_out[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  float rmse_out = 1e9f;
  int valid = 0, n_inl = 0, best_iter = -1, rounds = 0;
  const double thr = P.max_mah_dist_for_inliers;
#ifdef LF_POSE_PROFILE
  if (blockIdx.x == 7 && tid == 0) g_pprev = __builtin_amdgcn_s_memtime();
#endif
  if (g.go) {
    best_iter = win[0];
    const int s3[3] = {win[2], win[3], win[4]};
    const int nbest = win[1];
    if (0 + nbest >= 3) {                                                                    // :725-728
      float tf_best[16], sse_best = 0;
      r_model(s3, cm, tf_best);                // recompute the winning model (uniform) and its inlier list / sse
      float refined_tf[16];
#pragma unroll
      for (int i = 0; i < 16; i++) refined_tf[i] = tf_best[i];
      PT(12);
      double refined_rmse = 0;
      int nref = 0;
      int *inl = b.inliers + (size_t)pr * LF_MAX_MATCHES;
      // round -1: the refinement of the RANSAC winner's inliers (25 iterations, :730-731); rounds 0..19: the re-scoring loop
      // (:775-839).  ONE call site of r_refine: inlined twice the kernel is 22 k instructions, twice the instruction cache
      for (int iter = -1; iter < 20; ++iter) {
        int nset;
        if (iter < 0) {
          double sse_unused;
          nset = r_score(S.rs, cm, nLn, tf_best, thr, S.set, &sse_best, &sse_unused);
          refined_rmse = lf_sqrt(sse_best / (0 + nset));                                     // :731
        } else {
          float tmp_sse_f;
          double tmp_sse;
          __syncthreads();
          const int ncur = r_score(S.rs, cm, nLn, refined_tf, thr, S.idx, &tmp_sse_f, &tmp_sse);   // into a scratch list first (kept only if it improves)
          if (!(0 + ncur * lw > 0 + nref * lw)) break;
          for (int i = tid; i < ncur; i += RT_N) { S.set[i] = S.idx[i]; inl[i] = S.idx[i]; }
          __syncthreads();
          nref = ncur;
          refined_rmse = lf_sqrt(tmp_sse / (0 + ncur));
          rounds++;
          nset = nref;
        }
        r_refine(S.rs, cm, P, S.set, nset, refined_tf, iter < 0 ? 25 : 20);
      }
      n_inl = nref;
      rmse_out = (float)refined_rmse;
#pragma unroll
      for (int i = 0; i < 16; i++) tf_out[i] = refined_tf[i];
      valid = ((0 + lw * nref) >= min_inlier) ? 1 : 0;
    }
  }
#ifdef LF_POSE_PROFILE
  PT(13);
  if (blockIdx.x == 7 && tid == 0) {
    printf("k_pose blocks (kticks): newer evals %.1f vn %.1f older evals %.1f vo hw hp %.1f bl, put %.1f sum+barriers %.1f\n", g_pprof[1] / 1e3, g_pprof[2] / 1e3, g_pprof[3] / 1e3, g_pprof[4] / 1e3, g_pprof[8] / 1e3, g_pprof[5] / 1e3);
    printf("k_pose elim (kticks): gather %.1f solve6 %.1f W Vi, T %.1f ordered sum %.1f tail %.1f | %d linearisations, %d eliminations\n", g_pprof[1] / 1e3, g_pprof[2] / 1e3, g_pprof[3] / 1e3, g_pprof[4] / 1e3, g_pprof[7] / 1e3, (int)g_pprof[15], (int)g_pprof[14]);
    printf("k_pose prof (kticks) n=%d: pre-blocks %.1f blocks %.1f lambda %.1f elim %.1f solve %.1f backsub+chi %.1f accept %.1f | ransac %.1f rescoring/other %.1f\n", nLn,
           g_pprof[0] / 1e3, (g_pprof[5] + g_pprof[8]) / 1e3, g_pprof[6] / 1e3, (g_pprof[1] + g_pprof[2] + g_pprof[3] + g_pprof[4] + g_pprof[7]) / 1e3, g_pprof[9] / 1e3, g_pprof[10] / 1e3, g_pprof[11] / 1e3, g_pprof[12] / 1e3, g_pprof[13] / 1e3);
    for (int i = 0; i < 16; i++) g_pprof[i] = 0;
  }
#endif
  if (tid == 0) {
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
    res->overflow = ((n_all > c.match_cap || n_all > LF_MAX_MATCHES) ? LF_OVF_MATCHES : 0) |
                    ((b.nlines[fq] > c.line_cap || b.nlines_t[ft] > (b.line_cap_t < c.line_cap ? b.line_cap_t : c.line_cap)) ? LF_OVF_LINES : 0);
    res->reserved_ = 0;
  }
}

// ------------------------------------------------------------------------------ k_pose_w (round 6: the default refinement stage)
// The same stage as k_pose with WV_W low-footprint wavefronts per pair (lf_pose_wave.h): <= 128 registers and ~8 KB of LDS each, so that
// its wavefronts are placed beside the front end's instead of waiting for empty compute units.  Results are bit-identical to
// k_pose (which stays selectable: LF_POSE_RES=1 in the environment).
#ifndef LF_POSEW_WAVES
#define LF_POSEW_WAVES 4        // 512 / 4 = 128 registers: the budget of the kernel and of every phase it calls
#endif
__device__ __forceinline__ void w_pair(WaveShared &S, const PairConsts &c, const PairBuffers &b) {
  const int pr = blockIdx.x, tid = threadIdx.x;
  const int fq = w_uni(b.pair_q[pr]), ft = w_uni(b.pair_t[pr]);   // (loaded by every lane alike: scalar from here on)
  lf_pair_result *res = b.results + pr;
  double *ws = b.ws + (size_t)pr * LF_PAIR_WS_DOUBLES;
  const double *cm = ws;                                          // [nLn][R_CM], laid down by k_ransac
  const int *win = (const int *)(cm + R_WIN_OFF);
  const lf_params &P = c.P;
  const PoseGate g = r_gate(c, b, pr, fq, ft);
  const int nLn = w_uni(g.nLn), n_all = w_uni(g.n_all), lw = w_uni(g.lw), min_inlier = w_uni(g.min_inlier);
  const int id_t = w_uni((int)g.id_t), id_q = w_uni((int)g.id_q);     // (the result record keeps them as int)
  const bool go = w_uni((int)g.go) != 0;
  float rmse_out = 1e9f;
  int valid = 0, n_inl = 0, best_iter = -1, rounds = 0;
  bool have_tf = false;                       // (the transform stays in S.tf: nothing of it is live across the phases' calls)
  const double thr = P.max_mah_dist_for_inliers;
  if (go) {
    best_iter = w_uni(win[0]);
    const int nbest = w_uni(win[1]);
    if (0 + nbest >= 3) {                                                                    // :725-728
      float sse_best = 0;
      const float *wtf = (const float *)(cm + WV_OFF_TF);       // the winning model (k_ransac), re-scored for its inlier list / sse
      if (tid < 16) S.tf[tid] = wtf[tid];
      w_order();
      double refined_rmse = 0;
      int nref = 0;
      int *inl = b.inliers + (size_t)pr * LF_MAX_MATCHES;
      // round -1: the refinement of the RANSAC winner's inliers (25 iterations, :730); rounds 0..19: the re-scoring loop (:775-839)
      for (int iter = -1; iter < 20; ++iter) {
        int nset;
        if (iter < 0) {
          nset = w_uni(w_score(S, cm, nLn, thr, 0));
          sse_best = S.sse_f;
          refined_rmse = lf_sqrt(sse_best / (0 + nset));                                       // :731
        } else {
          const int ncur = w_uni(w_score(S, cm, nLn, thr, 1));                                   // into a scratch list first (kept only if it improves)
          const double tmp_sse = S.sse_d;
          if (!(0 + ncur * lw > 0 + nref * lw)) break;
          for (int i = tid; i < ncur; i += WV_T) { S.set[i] = S.idx[i]; inl[i] = S.idx[i]; }
          w_order();
          nref = ncur;
          refined_rmse = lf_sqrt(tmp_sse / (0 + ncur));
          rounds++;
          nset = nref;
        }
        w_refine(S, ws, P.g2o_line_error_weight, P.g2o_BA_kernel_delta, P.g2o_BA_use_kernel, nset, iter < 0 ? 25 : 20);
      }
      n_inl = nref;
      rmse_out = (float)refined_rmse;
      have_tf = true;
      valid = ((0 + lw * nref) >= min_inlier) ? 1 : 0;
    }
  }
  if (tid == 0) {
    for (int i = 0; i < 16; i++) res->T[i] = have_tf ? S.tf[i] : ((i % 5 == 0) ? 1.0f : 0.0f);
    res->rmse = rmse_out;
    res->valid = valid;
    res->n_matches = n_all;
    res->n_inliers = n_inl;
    res->n_point_matches = 0;
    res->n_point_inliers = 0;
    res->id_older = valid ? id_t : -1;             // node.cpp:1606-1607
    res->id_newer = valid ? id_q : -1;
    res->ransac_best_iter = best_iter;
    res->refine_rounds = rounds;
    float r2 = rmse_out * rmse_out;                                   // float arithmetic as node.cpp:1533-1534
    res->information_scale = valid ? (double)((float)(0 + n_inl * lw) / r2) : 0.0;
    res->overflow = ((n_all > c.match_cap || n_all > LF_MAX_MATCHES) ? LF_OVF_MATCHES : 0) |
                    ((b.nlines[fq] > c.line_cap || b.nlines_t[ft] > (b.line_cap_t < c.line_cap ? b.line_cap_t : c.line_cap)) ? LF_OVF_LINES : 0);
    res->reserved_ = 0;
  }
}

// (waves-per-SIMD 4 = the 128-register budget, inherited by every phase; with 16.3 KB of LDS per pair the compiler's occupancy
// estimate is 3 and it says so: the budget is what is wanted, not the fourth wavefront)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
__global__ void __launch_bounds__(WV_T, LF_POSEW_WAVES) k_pose_w(PairConsts c, PairBuffers b) {
  __shared__ WaveShared S;
  w_pair(S, c, b);
}
#pragma clang diagnostic pop

// Which refinement kernel: the resident workgroup-per-pair k_pose (default: the faster one inside the pipelined step, measured)
// or k_pose_w, the form that shares compute units (LF_POSE_WAVE=1; bit-identical, tests/test_pose_wave_gpu.py).  LF_POSE_RES=0
// selects k_pose_w too (A/B scripts of round 6).
static bool pose_resident_selected() {
  static const bool v = [] {
    const char *w = getenv("LF_POSE_WAVE"), *r = getenv("LF_POSE_RES");
    if (w && w[0] == '1') return false;
    if (r && r[0] == '0') return false;
    return true;
  }();
  return v;
}

void lf_pair_launch(const PairConsts &c, const PairBuffers &b, int n_pairs, hipStream_t st, int solver, bool run_match) {
  if (run_match) {
    static const int form = [] { const char *e = getenv("LF_MATCH_FORM"); return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'h' ? 2 : 0)); }();   // tests: force one form
    if (form == 1 || (form == 0 && n_pairs >= LF_MATCH_LIGHT_PAIRS && solver != LF_SOLVER_NONE)) hipLaunchKernelGGL(k_match<true>, dim3(n_pairs), dim3(MatchCfg<true>::N), 0, st, c, b);
    else hipLaunchKernelGGL(k_match<false>, dim3(n_pairs), dim3(MatchCfg<false>::N), 0, st, c, b);
  }
  if (solver == LF_SOLVER_NONE) return;
  if (solver == LF_SOLVER_HYBRID) lf_pair_hybrid_launch(c, b, n_pairs, st);
  else if (solver == LF_SOLVER_RELMOTION) lf_pair_relmotion_launch(c, b, n_pairs, st);
  else {
#ifndef LF_EXP_SKIP_POSE   // (throughput experiments only)
    hipLaunchKernelGGL(k_ransac, dim3(n_pairs), dim3(RS_N), 0, st, c, b);
    if (!pose_resident_selected()) { hipLaunchKernelGGL(k_pose_w, dim3(n_pairs), dim3(WV_T), 0, st, c, b); return; }
#ifdef LF_POSE_WAVES
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void *)k_pose, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoseShared)); attr_set = true; }
    hipLaunchKernelGGL(k_pose, dim3(n_pairs), dim3(RT_N), sizeof(PoseShared), st, c, b);
#else
    hipLaunchKernelGGL(k_pose, dim3(n_pairs), dim3(RT_N), 0, st, c, b);
#endif
#endif
  }
}
