// lf_pair_legacy.hip -- Node::getRelativeTransformationTo (src/node.cpp:1134-1338), the point-feature RANSAC that
// Node::matchNodePair calls in builds WITHOUT USE_LINES (node.cpp:1494-1615): hypotheses from four matches sampled with a
// preference for small descriptor distance (sample_matches_prefer_by_distance, :1085-1108), the weighted closed-form
// rigid transform of getTransformFromMatches (src/transformation_estimation_euclidean.cpp:7-56, PCL's
// TransformationFromCorrespondences), inliers by computeInliersAndError (:1021-1080, errorFunction2), up to 19 refinement
// rounds per hypothesis, the identity as last resort.  The g2o step at its end (:1283-1327, EdgeSE3PointXYZDepth,
// "g2o_transformation_refinement", 0 by default) is NOT restated: the entry point refuses refine iterations > 0.
//
// ONE WAVEFRONT PER NODE PAIR.  The algorithm is a sequential state machine -- the iteration counter jumps by 10 or 20 when
// a hypothesis explains half / three quarters of the matches -- so the control flow is wave-uniform and only the scoring of
// a hypothesis against all matches runs one match per lane (ordered compaction by ballot; the error sum in list order).
// Deviations, both forced (the reference itself is not reproducible here): rand() (seeded with clock(), :1166) ->
// lf_rand31(seed, stream, 20002 * iteration + draw); std::sort's order among equal distances (introsort, unspecified) ->
// stable (ties keep the caller's order).  `abs(delta_f - delta_t)` (:40 of the euclidean file) is taken as the float
// absolute value.  Sequential twin: oracle_legacy_ransac (oracle/pair_oracle.c); independent restatement:
// oracle/pose_indep.py legacy_ransac.
#include "lf_pair_legacy.h"
#include "lf_pose.h"

typedef unsigned long long u64;
__device__ __forceinline__ int g_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ void g_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

struct LegacyShared {
  int perm[LF_LEGACY_CAP];          // sorted position -> index into the caller's match arrays
  int cur[LF_LEGACY_CAP], ref[LF_LEGACY_CAP], best[LF_LEGACY_CAP];   // inlier lists (sorted positions)
  double err[LF_LEGACY_CAP];        // errors of the inliers of the last scoring, in list order
};

// computeInliersAndError (node.cpp:1021-1080) over ALL matches (sorted order); list + count + error
__device__ int g_score(LegacyShared &S, const LegacyArgs &a, const float *tf, double thr2, int *list, double *err_out) {
  const int lane = g_lane();
  int cnt = 0;
  for (int base = 0; base < a.n; base += 64) {
    const int i = base + lane;
    bool keep = false;
    double e = 0.0;
    if (i < a.n) {
      const int m = S.perm[i];
      const float *x1 = a.pts_q + 4 * (size_t)a.mq[m], *x2 = a.pts_t + 4 * (size_t)a.mt[m];
      if (!(x1[2] == 0.0f || x2[2] == 0.0f)) {                 // (does NOT trigger on NaN, :1045)
        e = lf_error_function2(x1, x2, tf, &a.pm);
        keep = !(e > thr2) && (e >= 0.0);
      }
    }
    const u64 mk = __ballot(keep);
    if (keep) { const int at = cnt + __popcll(mk & ((1ull << lane) - 1ull)); list[at] = i; S.err[at] = e; }
    cnt += __popcll(mk);
  }
  g_order();
  double mean = 0.0;
  for (int k = 0; k < cnt; k++) mean += S.err[k];            // in list order (wave-uniform: every lane adds the same terms)
  g_order();
  *err_out = cnt < 3 ? 1e9 : lf_sqrt(mean / (double)cnt);
  return cnt;
}

// getTransformFromMatches (transformation_estimation_euclidean.cpp:7-56); list = sorted positions
__device__ bool g_transform(const LegacyShared &S, const LegacyArgs &a, const int *list, int cnt, float *tf) {
  lf_tfc t;
  lf_tfc_reset(&t);
  bool have_prev = false;
  float pf[3] = {0, 0, 0}, pt[3] = {0, 0, 0};
  for (int k = 0; k < cnt; k++) {
    const int m = S.perm[list[k]];
    const float *from = a.pts_q + 4 * (size_t)a.mq[m], *to = a.pts_t + 4 * (size_t)a.mt[m];
    if (from[2] != from[2] || to[2] != to[2]) continue;
    const float w = 1 / (to[2] + from[2]);
    if (a.max_dist_m > 0) {
      if (have_prev) {
        const float df = ((from[0] - pf[0]) * (from[0] - pf[0]) + (from[1] - pf[1]) * (from[1] - pf[1])) + (from[2] - pf[2]) * (from[2] - pf[2]);
        const float dt = ((to[0] - pt[0]) * (to[0] - pt[0]) + (to[1] - pt[1]) * (to[1] - pt[1])) + (to[2] - pt[2]) * (to[2] - pt[2]);
        const float d = df - dt;
        if ((d < 0 ? -d : d) > a.max_dist_m * a.max_dist_m) return false;
      }
      for (int c = 0; c < 3; c++) { pf[c] = from[c]; pt[c] = to[c]; }
      have_prev = true;
    }
    lf_tfc_add(&t, from, to, w);
  }
  lf_tfc_get(&t, tf);
  return true;
}

__global__ void __launch_bounds__(64) k_legacy_ransac(LegacyArgs a) {
  __shared__ LegacyShared S;
  const int lane = g_lane(), n = a.n;
  LegacyResult *R = a.out;
  float T[16], rmse = 1e6f;
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  int nbest = 0, valid_iterations = 0, best_iter = -1, real_iterations = 0;
  bool enough = false;
  if (n > a.min_matches) {                                                      // :1147-1150
    unsigned min_thr = (unsigned)a.min_matches;
    if ((double)min_thr > 0.75 * (double)n) min_thr = (unsigned)(0.75 * (double)n);   // :1155-1159
    // std::sort(matches_with_depth) by distance (:1193); ties keep the caller's order
    for (int i = lane; i < n; i += 64) {
      const float di = a.md[i];
      int rank = 0;
      for (int j = 0; j < n; j++) { const float dj = a.md[j]; rank += (dj < di || (dj == di && j < i)) ? 1 : 0; }
      S.perm[rank] = i;
    }
    g_order();
    const double thr2 = (double)(a.max_dist_m * a.max_dist_m);                   // max_dist_m*max_dist_m: float product
    for (int it = 0; it < a.iterations && n >= 4; it++) {
      double refined_error = 1e6, inlier_error = 0.0;
      int nref = 0, ncur = 0;
      float rtf[16];
      for (int i = 0; i < 16; i++) rtf[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      {   // sample_matches_prefer_by_distance(4): the smaller of two uniform draws, into a set (ascending), <= 10000 tries
        int ids[4], ns = 0, safety = 0;
        u64 ctr = (u64)it * 20002ull;
        while (ns < 4) {
          int id1 = (int)(lf_rand31(a.seed, a.stream, ctr) % (uint32_t)n), id2 = (int)(lf_rand31(a.seed, a.stream, ctr + 1) % (uint32_t)n);
          ctr += 2;
          if (id1 > id2) id1 = id2;
          int pos = 0;
          bool dup = false;
          while (pos < ns && ids[pos] <= id1) { if (ids[pos] == id1) dup = true; pos++; }
          if (!dup) { for (int k = ns; k > pos; k--) ids[k] = ids[k - 1]; ids[pos] = id1; ns++; }
          if (++safety > 10000) break;
        }
        if (lane == 0) for (int k = 0; k < ns; k++) S.cur[k] = ids[k];
        ncur = ns;
        g_order();
      }
      real_iterations++;
      for (int refinements = 1; refinements < 20; refinements++) {
        float tf[16];
        if (!g_transform(S, a, S.cur, ncur, tf)) break;
        bool nan = false;
        for (int i = 0; i < 16; i++) nan = nan || (tf[i] != tf[i]);
        if (nan) break;
        ncur = g_score(S, a, tf, thr2, S.cur, &inlier_error);
        if ((unsigned)ncur < min_thr || inlier_error > (double)a.max_dist_m) break;
        if (ncur >= nref && inlier_error <= refined_error) {
          const int prev = nref;
          for (int i = 0; i < 16; i++) rtf[i] = tf[i];
          for (int i = lane; i < ncur; i += 64) S.ref[i] = S.cur[i];
          g_order();
          nref = ncur;
          refined_error = inlier_error;
          if (ncur == prev) break;
        } else break;
      }
      if (nref > 0) {
        valid_iterations++;
        if (refined_error <= (double)rmse && nref >= nbest && (unsigned)nref >= min_thr) {
          rmse = (float)refined_error;
          for (int i = 0; i < 16; i++) T[i] = rtf[i];
          for (int i = lane; i < nref; i += 64) S.best[i] = S.ref[i];
          g_order();
          nbest = nref;
          best_iter = it;
          if ((double)nref > (double)n * 0.5) it += 10;
          if ((double)nref > (double)n * 0.75) it += 10;
          if ((double)nref > (double)n * 0.8) break;
        }
      }
    }
    if (valid_iterations == 0) {                                                 // :1253-1275 identity as hypothesis
      float I4[16];
      for (int i = 0; i < 16; i++) I4[i] = (i % 5 == 0) ? 1.0f : 0.0f;
      double inlier_error;
      const int nc = g_score(S, a, I4, thr2, S.cur, &inlier_error);
      if ((unsigned)nc > min_thr && inlier_error < (double)a.max_dist_m) {
        for (int i = 0; i < 16; i++) T[i] = I4[i];
        for (int i = lane; i < nc; i += 64) S.best[i] = S.cur[i];
        g_order();
        nbest = nc;
        rmse = (float)inlier_error;
        valid_iterations++;
      }
    }
    enough = (unsigned)nbest >= min_thr;
  }
  for (int i = lane; i < nbest; i += 64) a.out_inliers[i] = S.perm[S.best[i]];   // indices into the caller's match arrays, in sorted order
  if (lane == 0) {
    for (int i = 0; i < 16; i++) R->T[i] = T[i];
    R->rmse = rmse; R->found = enough ? 1 : 0; R->n_inliers = nbest; R->valid_iterations = valid_iterations;
    R->best_iteration = best_iter; R->iterations_run = real_iterations;
  }
}

void lf_legacy_launch(const LegacyArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_legacy_ransac, dim3(1), dim3(64), 0, st, a); }
