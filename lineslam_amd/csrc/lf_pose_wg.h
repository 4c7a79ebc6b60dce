// lf_pose_wg.h -- the workgroup-per-pair building blocks of the pose kernels (k_pose, k_pose_hybrid): ordered sums
// from LDS-published values, ordered walks over workspace rows, lane helpers.  The LM refinement of the LINE landmarks
// itself is lf_pose_res.h (resident passes, both kernels).  Included by lf_pair.hip and lf_pair_hybrid.hip after
// lf_pair.h / lf_pose.h.
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
// wavefront, forty per pass of the workgroup (lf_pose_res.h).
#define PG_N 10                    // matches per wavefront pass
__device__ __forceinline__ void p_wave_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// lf_solve6 (lf_linalg.h: partial pivoting, reciprocal pivots, row-oriented back-substitution) with one column per lane:
// lane (g, d) holds column d of A in a[] and ONE column of the right-hand side in b[] (its own column of a six-column
// right-hand side, or a copy of the only one).  Step k: column k is fetched from lane k of the group; the pivot search, the
// reciprocal and the multipliers are computed by all six lanes alike; each lane swaps / updates rows k+1.. of its own two
// columns.  Entries the sequential code never reads again (rows > d of column d after step d) are not maintained.  The upper
// triangle is published through `urow` (36 doubles of LDS per group) for the back-substitution.  Each entry goes through
// the operations lf_solve6 applies to it, in its order: same bits.  Returns 0 for a singular system (b is garbage then).
__device__ __forceinline__ int p_solve6_cols(double (&a)[6], double (&b)[6], int base, int d, bool lane_ok, double *urow) {
  double rp[6];
  int ok = 1;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double ck[6];
#pragma unroll
    for (int i = k; i < 6; i++) ck[i] = __shfl(a[i], base + k, 64);
    int piv = k;
    double big = lf_fabs(ck[k]);
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double v = lf_fabs(ck[i]);
      if (v > big) { big = v; piv = i; }
    }
    if (!(big > 0.0)) ok = 0;
    if (LF_ANY(piv != k)) {
#pragma unroll
      for (int i = k + 1; i < 6; i++) {
        const bool sw = i == piv;
        { const double x = a[k], y = a[i]; a[k] = sw ? y : x; a[i] = sw ? x : y; }
        { const double x = b[k], y = b[i]; b[k] = sw ? y : x; b[i] = sw ? x : y; }
        { const double x = ck[k], y = ck[i]; ck[k] = sw ? y : x; ck[i] = sw ? x : y; }
      }
    }
    rp[k] = 1.0 / ck[k];
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double f = ck[i] * rp[k];
      const bool nz = f != 0.0;
      { const double v = a[i] - f * a[k]; a[i] = nz ? v : a[i]; }
      { const double v = b[i] - f * b[k]; b[i] = nz ? v : b[i]; }
    }
  }
  if (lane_ok) {
#pragma unroll
    for (int i = 0; i < 5; i++) urow[6 * i + d] = a[i];      // (rows i < d are the final U entries of column d; the rest is not read)
  }
  p_wave_order();
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = b[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= urow[6 * i + k] * b[k];
    b[i] = s * rp[i];
  }
  p_wave_order();
  return ok;
}

