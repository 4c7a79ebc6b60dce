// lf_front.hip -- 3D-line stage for gfx950: everything Node::detect3DLines does after LSD
// (src/line/lineslam.cpp:213-357), for a batch of frames.
//
//   k_sobel5     cv::Sobel(gray, CV_64F, ksize 5) x2         lineslam.cpp:311-314  (stored int16: the
//                values are exact integers |v| <= 24480, 4x less HBM traffic than the reference's fp64)
//   k_line3d     ONE WAVEFRONT PER 2D SEGMENT:
//                  length filter + depth sampling             lineslam.cpp:213-221, 246-288
//                  compPt3dCov + RandomPoint3d ctor           utils.cpp:671-722, lineslam.h:59-81
//                  extract3dline_mahdist (RANSAC)             utils.cpp:343-427 (+570-624, 471-493)
//                  acceptance                                 lineslam.cpp:302-307
//   k_records    ordered compaction into `lines` (+lid, complineEq2d)              lineslam.cpp:332-341
//   k_mle        ONE WAVEFRONT PER 3D LINE: MLEstimateLine3d (dlevmar_dif) + covariance + rndA/rndB
//                utils.cpp:954-1050, 1086-1159, external/levmar-2.6/lm_core.c:438-846
//   k_describe   ONE WAVEFRONT PER 3D LINE: getGradient (lineslam.cpp:527-537) and computeMSLD
//                (utils.cpp:1510-1610)
//
// Parallel mapping inside a wavefront: sample points / residuals live one or two per lane in LDS;
// every floating-point SUM is accumulated in the reference's index order (one accumulator per lane
// for J^T J, uniform loops elsewhere), so the results are bit-identical to the sequential oracle.
#include "lf_front.h"
#include "lf_linalg.h"
#include <float.h>

typedef unsigned long long u64;
#define F_EPS 1e-10   // lineslam.h:37
#ifndef LF_PIVOT_FAST
#define LF_PIVOT_FAST 1
#endif

#ifndef LF_MLE_PAIR64
#define LF_MLE_PAIR64 0    // 33..64-point lines two per wavefront (two rows per lane): measured slower again in round 3 (19.6 + 11.0 ms vs 20.8 ms)
#endif
#ifndef LF_MLE_SMALL
#define LF_MLE_SMALL 0           // lines with at most this many support points run four to a wavefront; 16 measured slower again in round 2 (3D stage 80.0 vs 77.3 ms: four independent LM state machines diverge), so: none
#endif
__device__ __forceinline__ int f_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 f_lt() { return (1ull << f_lane()) - 1ull; }
__device__ __forceinline__ double f_rl64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void f_sync() { __syncthreads(); }   // blocks are single wavefronts

// ------------------------------------------------------------------------------ Sobel 5x5
__device__ __forceinline__ int f_reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}
#define SOBEL_ROWS 4   // output rows per thread: the horizontal passes of an input row serve all the output rows it touches
__global__ void __launch_bounds__(256) k_sobel5(FrontConsts c, FrontBuffers b) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * SOBEL_ROWS, f = blockIdx.z;
  if (x >= c.W) return;
  const uint8_t *g = b.gray + (size_t)f * b.gray_frame_stride;
  const int kd[5] = {-1, -2, 0, 2, 1}, ks[5] = {1, 4, 6, 4, 1};
  int xi[5];
#pragma unroll
  for (int i = 0; i < 5; i++) xi[i] = f_reflect101(x + i - 2, c.W);
  int rd[SOBEL_ROWS + 4], rs[SOBEL_ROWS + 4];      // horizontal derivative / smoothing of input rows y0-2 .. y0+SOBEL_ROWS+1
#pragma unroll
  for (int j = 0; j < SOBEL_ROWS + 4; j++) {
    const uint8_t *row = g + (size_t)f_reflect101(y0 + j - 2, c.H) * b.gray_row_stride;
    int d = 0, sm = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      int v = row[xi[i]];
      d += kd[i] * v;
      sm += ks[i] * v;
    }
    rd[j] = d; rs[j] = sm;
  }
#pragma unroll
  for (int r = 0; r < SOBEL_ROWS; r++) {
    const int y = y0 + r;
    if (y < c.H) {
      int sx = 0, sy = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) { sx += ks[j] * rd[r + j]; sy += kd[j] * rs[r + j]; }
      size_t o = ((size_t)f * c.H + y) * c.W + x;
      *(uint32_t *)&b.gxy[2 * o] = (uint32_t)(uint16_t)(int16_t)sx | ((uint32_t)(uint16_t)(int16_t)sy << 16);
    }
  }
}

// ------------------------------------------------------------------------------ point helpers
// depthStdDev + compPt3dCov (utils.cpp:671-722), same expression order as the oracle
__device__ __forceinline__ void f_pt_cov(const double *pt, double f, const lf_params &P, double *cov) {
  double sig = P.stdev_sample_pt_imgline;
  double c1 = P.depth_stdev_coeff_c1, c2 = P.depth_stdev_coeff_c2 + 0.0 * 0.5, c3 = P.depth_stdev_coeff_c3;
  double sz = c1 * pt[2] * pt[2] + c2 * pt[2] + c3;
  double s2 = sig * sig, sz2 = sz * sz;
  double j00 = pt[2] / f, j02 = pt[0] / pt[2], j11 = pt[2] / f, j12 = pt[1] / pt[2];
  cov[0] = (j00 * s2) * j00 + (j02 * sz2) * j02;
  cov[1] = (j02 * sz2) * j12;
  cov[2] = (j02 * sz2);
  cov[3] = (j12 * sz2) * j02;
  cov[4] = (j11 * s2) * j11 + (j12 * sz2) * j12;
  cov[5] = (j12 * sz2);
  cov[6] = sz2 * j02;
  cov[7] = sz2 * j12;
  cov[8] = sz2;
}
// RandomPoint3d(pos, cov) ctor (lineslam.h:59-81): W_sqrt and DU = diag(1/W_sqrt) U^T
__device__ __forceinline__ void f_whiten(const double *cov, double *DU, double *Wsq) {
  double A[9], V[9], w[3];
#pragma unroll
  for (int i = 0; i < 9; i++) A[i] = cov[i];
  lf_jacobi3(A, V, w);
#pragma unroll
  for (int i = 0; i < 3; i++) Wsq[i] = lf_sqrt(w[i]);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) DU[3 * i + j] = (1 / Wsq[i]) * V[3 * j + i];
}
// mah_dist3d_pt_line (utils.cpp:761-822)
__device__ __forceinline__ double f_mah(const double *pos, const double *c, const double *q1, const double *q2) {
  double x1 = pos[0], x2 = pos[1], x3 = pos[2];
  double xa = q1[0], ya = q1[1], za = q1[2], xb = q2[0], yb = q2[1], zb = q2[2];
  double a0 = c[0] * (x1 - xa) + c[1] * (x2 - ya) + c[2] * (x3 - za);
  double a1 = c[3] * (x1 - xa) + c[4] * (x2 - ya) + c[5] * (x3 - za);
  double a2 = c[6] * (x1 - xa) + c[7] * (x2 - ya) + c[8] * (x3 - za);
  double b0 = c[0] * (x1 - xb) + c[1] * (x2 - yb) + c[2] * (x3 - zb);
  double b1 = c[3] * (x1 - xb) + c[4] * (x2 - yb) + c[5] * (x3 - zb);
  double b2 = c[6] * (x1 - xb) + c[7] * (x2 - yb) + c[8] * (x3 - zb);
  double t1 = a0 * b1 - a1 * b0, t2 = a0 * b2 - a2 * b0, t3 = a1 * b2 - a2 * b1;
  double t4 = c[0] * (x1 - xa) - c[0] * (x1 - xb) + c[1] * (x2 - ya) - c[1] * (x2 - yb) + c[2] * (x3 - za) - c[2] * (x3 - zb);
  double t5 = c[3] * (x1 - xa) - c[3] * (x1 - xb) + c[4] * (x2 - ya) - c[4] * (x2 - yb) + c[5] * (x3 - za) - c[5] * (x3 - zb);
  double t6 = c[6] * (x1 - xa) - c[6] * (x1 - xb) + c[7] * (x2 - ya) - c[7] * (x2 - yb) + c[8] * (x3 - za) - c[8] * (x3 - zb);
  return lf_sqrt((t1 * t1 + t2 * t2 + t3 * t3) / (t4 * t4 + t5 * t5 + t6 * t6));
}

// first-occurrence arg-min / arg-max over the lanes' candidates (value, index); lower index wins ties
__device__ __forceinline__ void f_argmin(double &v, int &i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
__device__ __forceinline__ void f_argmax(double &v, int &i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
__device__ __forceinline__ u64 f_or64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}

struct L3State {          // LDS of one wavefront
  double pos[LF_MAX_SAMPLES * 3];
  double DU[LF_MAX_SAMPLES * 9];
  int idx[LF_MAX_SAMPLES];
};

// inlier masks of all points against the line (q1,q2); points lane and lane+64
__device__ __forceinline__ void f_inliers(const L3State &S, int n, const double *q1, const double *q2, double thr,
                                          u64 *m0, u64 *m1) {
  int lane = f_lane();
  bool i0 = false, i1 = false;
  if (lane < n) i0 = f_mah(&S.pos[3 * lane], &S.DU[9 * lane], q1, q2) < thr;
  if (lane + 64 < n) i1 = f_mah(&S.pos[3 * (lane + 64)], &S.DU[9 * (lane + 64)], q1, q2) < thr;
  *m0 = __ballot(i0);
  *m1 = __ballot(i1);
}

// verify3dLine (utils.cpp:570-624) on the inlier set (m0,m1)
__device__ bool f_verify3d(const L3State &S, u64 m0, u64 m1, const double *A, const double *B, const lf_params &P) {
  int lane = f_lane();
  int nCells = P.num_cells_lineseg_range;
  if (nCells > 64) nCells = 64;
  double AB[3], mid[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { AB[k] = B[k] - A[k]; mid[k] = (A[k] + B[k]) * 0.5; }
  bool in0 = (m0 >> lane) & 1ull, in1 = (m1 >> lane) & 1ull;
  const double *x0 = &S.pos[3 * lane], *x1 = &S.pos[3 * (lane + 64)];
  double d0 = in0 ? (x0[0] - A[0]) * AB[0] + (x0[1] - A[1]) * AB[1] + (x0[2] - A[2]) * AB[2] : 0.0;
  double d1 = in1 ? (x1[0] - A[0]) * AB[0] + (x1[1] - A[1]) * AB[1] + (x1[2] - A[2]) * AB[2] : 0.0;
  int first = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);   // list position 0
  // arg-min with the reference's initial value 100 (idx 0 if nothing is below it)
  double vmin = 100.0; int imin = 1 << 30;
  if (in0 && d0 < vmin) { vmin = d0; imin = lane; }
  if (in1 && d1 < vmin) { vmin = d1; imin = lane + 64; }
  f_argmin(vmin, imin);
  if (imin == (1 << 30)) imin = first;
  double vmax = -100.0; int imax = 1 << 30;
  if (in0 && d0 > vmax) { vmax = d0; imax = lane; }
  if (in1 && d1 > vmax) { vmax = d1; imax = lane + 64; }
  f_argmax(vmax, imax);
  if (imax == (1 << 30)) imax = first;
  double C[3], D[3];
  {
    double Bm[3], ab[3], ap[3], s;
#pragma unroll
    for (int k = 0; k < 3; k++) { Bm[k] = mid[k] + AB[k]; ab[k] = Bm[k] - mid[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) ap[k] = S.pos[3 * imin + k] - mid[k];
    s = (ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2]) / (ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) C[k] = mid[k] + s * ab[k];
#pragma unroll
    for (int k = 0; k < 3; k++) ap[k] = S.pos[3 * imax + k] - mid[k];
    s = (ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2]) / (ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) D[k] = mid[k] + s * ab[k];
  }
  double cd = lf_sqrt((D[0] - C[0]) * (D[0] - C[0]) + (D[1] - C[1]) * (D[1] - C[1]) + (D[2] - C[2]) * (D[2] - C[2]));
  if (cd < F_EPS) return false;
  u64 cells = 0;
  if (in0) {
    double lambda = lf_fabs(((x0[0] - C[0]) * (D[0] - C[0]) + (x0[1] - C[1]) * (D[1] - C[1]) + (x0[2] - C[2]) * (D[2] - C[2])) / cd / cd);
    if (lambda >= 1) cells |= 1ull << (nCells - 1);
    else { unsigned int cidx = (unsigned int)__builtin_floor(lambda * 10); if (cidx < 64) cells |= 1ull << cidx; }
  }
  if (in1) {
    double lambda = lf_fabs(((x1[0] - C[0]) * (D[0] - C[0]) + (x1[1] - C[1]) * (D[1] - C[1]) + (x1[2] - C[2]) * (D[2] - C[2])) / cd / cd);
    if (lambda >= 1) cells |= 1ull << (nCells - 1);
    else { unsigned int cidx = (unsigned int)__builtin_floor(lambda * 10); if (cidx < 64) cells |= 1ull << cidx; }
  }
  cells = f_or64(cells);
  if (nCells < 64) cells &= (1ull << nCells) - 1ull;
  double sum = (double)__popcll(cells);
  return sum / nCells > P.ratio_support_pts_on_line;
}

// computeLine3d_svd on the index set (m0,m1) (utils.cpp:471-493): sums in ascending index order
__device__ void f_line3d_svd(const L3State &S, u64 m0, u64 m1, int n, double *mean, double *drct) {
  double Sm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V[9], w[3];
  mean[0] = mean[1] = mean[2] = 0;
  for (int half = 0; half < 2; half++) {
    u64 mm = half ? m1 : m0;
    while (mm) {
      int i = __builtin_ctzll(mm) + 64 * half;
      mm &= mm - 1;
#pragma unroll
      for (int k = 0; k < 3; k++) mean[k] = mean[k] + S.pos[3 * i + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) mean[k] = mean[k] * (1.0 / n);
  for (int half = 0; half < 2; half++) {
    u64 mm = half ? m1 : m0;
    while (mm) {
      int i = __builtin_ctzll(mm) + 64 * half;
      mm &= mm - 1;
      double d[3];
#pragma unroll
      for (int k = 0; k < 3; k++) d[k] = S.pos[3 * i + k] - mean[k];
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int l = 0; l < 3; l++) Sm[3 * k + l] += d[k] * d[l];
    }
  }
  lf_jacobi3(Sm, V, w);
#pragma unroll
  for (int k = 0; k < 3; k++) drct[k] = V[3 * k + 0];
}

// jac_rpt2ln_mahvec_wrt_ln (utils.cpp:1086-1116), closed form (see oracle/front_oracle.c o_jac_line)
__device__ __forceinline__ void f_jac_line(const double *pos, const double *M, const double *l, double *J) {
  double a[3], b[3], d[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    a[r] = M[3 * r] * (pos[0] - l[0]) + M[3 * r + 1] * (pos[1] - l[1]) + M[3 * r + 2] * (pos[2] - l[2]);
    b[r] = M[3 * r] * (pos[0] - l[3]) + M[3 * r + 1] * (pos[1] - l[4]) + M[3 * r + 2] * (pos[2] - l[5]);
    d[r] = a[r] - b[r];
  }
  double Sd = a[0] * d[0] + a[1] * d[1] + a[2] * d[2];
  double D = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double ma = M[j] * a[0] + M[3 + j] * a[1] + M[6 + j] * a[2];
    double md = M[j] * d[0] + M[3 + j] * d[1] + M[6 + j] * d[2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double Mrj = M[3 * r + j];
      J[r * 6 + j] = Mrj - Mrj * Sd / D - d[r] * (ma + md) / D + 2.0 * Sd * md * d[r] / (D * D);
      J[r * 6 + 3 + j] = ma * d[r] / D + Mrj * Sd / D - 2.0 * Sd * md * d[r] / (D * D);
    }
  }
}

// Sample j of a 2D segment (pa, pb) - (qc, qd) back-projected with the depth map (lineslam.cpp:252-288): false if the sample
// lies outside the image or has no depth.  Used by k_line3d (all samples of a candidate) and again by k_mle, which re-derives the
// supporting points of a kept line from the segment and the 128-bit inlier mask k_line3d leaves behind -- the same operations
// on the same operands, so the same doubles -- instead of reading them from a per-frame pool of point slots (rounds 3-4:
// 2.5 KB per slot, 2 x line_cap slots per frame, and a frame could run out of them).
__device__ __forceinline__ bool f_sample_point(const FrontConsts &c, const FrontBuffers &b, const float *depth, double pa, double pb,
                                               double qc, double qd, double numSmp, int j, double *X, double *Y, double *Z) {
  if (!((double)j <= numSmp && j < LF_MAX_SAMPLES)) return false;
  double ptx = pa * (1 - j / numSmp) + qc * (j / numSmp);
  double pty = pb * (1 - j / numSmp) + qd * (j / numSmp);
  if (ptx < 0 || pty < 0 || ptx >= c.W || pty >= c.H) return false;
  int row, col;
  if ((__builtin_floor(ptx) == ptx) && (__builtin_floor(pty) == pty)) {
    col = (int)(ptx - 1); if (col < 0) col = 0;
    row = (int)(pty - 1); if (row < 0) row = 0;
  } else { col = (int)ptx; row = (int)pty; }
  float dv = depth[(size_t)row * b.depth_row_stride + col];
  double depval = (double)dv, zval = -1;
  if (depval < F_EPS || dv != dv) { } else zval = depval / c.P.depth_scaling;
  if (!(zval > 0)) return false;
  double x = c.Kinv[0] * ptx + c.Kinv[1] * pty + c.Kinv[2] * 1.0;
  double y = c.Kinv[3] * ptx + c.Kinv[4] * pty + c.Kinv[5] * 1.0;
  double z = c.Kinv[6] * ptx + c.Kinv[7] * pty + c.Kinv[8] * 1.0;
  x = x / z; y = y / z;
  *X = x * zval; *Y = y * zval; *Z = zval;
  return true;
}

__global__ void __launch_bounds__(64) k_line3d(FrontConsts c, FrontBuffers b) {
  __shared__ L3State S;
  const int cand = blockIdx.x, f = blockIdx.y, lane = f_lane();
  const lf_params &P = c.P;
  int *flag = b.cand_flag + (size_t)f * c.cand_cap + cand;
  double *out = b.cand_out + ((size_t)f * c.cand_cap + cand) * LF_CAND_STRIDE;
  int nseg = b.nsegs[f];
  if (nseg > c.seg_cap) nseg = c.seg_cap;
  if (cand >= nseg) { if (lane == 0) *flag = 0; return; }
  const double *sg = b.segs + ((size_t)f * c.seg_cap + cand) * 5;
  double pa = sg[0], pb = sg[1], qc = sg[2], qd = sg[3];
  if (lane < LF_CAND_STRIDE / 2) { out[lane] = 0.0; out[lane + LF_CAND_STRIDE / 2] = 0.0; }
  if (!(lf_sqrt((pa - qc) * (pa - qc) + (pb - qd) * (pb - qd)) > P.line_segment_len_thresh)) {   // lineslam.cpp:218
    if (lane == 0) *flag = 0;
    return;
  }
  double len = lf_sqrt((pa - qc) * (pa - qc) + (pb - qd) * (pb - qd));
  double numSmp = len / P.line_sample_interval;
  if (numSmp < (double)P.line_sample_min_num) numSmp = (double)P.line_sample_min_num;
  if (numSmp > (double)P.line_sample_max_num) numSmp = (double)P.line_sample_max_num;
  const float *depth = b.depth + (size_t)f * b.depth_frame_stride;
  // ---- depth sampling (lineslam.cpp:252-288): sample j = lane and lane+64, order-preserving compaction
  int np = 0;
  for (int h = 0; h < 2; h++) {
    int j = lane + 64 * h;
    double X = 0, Y = 0, Z = 0;
    const bool ok = f_sample_point(c, b, depth, pa, pb, qc, qd, numSmp, j, &X, &Y, &Z);
    u64 mk = __ballot(ok);
    if (ok) {
      int pos = np + __popcll(mk & f_lt());
      S.pos[3 * pos] = X; S.pos[3 * pos + 1] = Y; S.pos[3 * pos + 2] = Z;
    }
    np += __popcll(mk);
  }
  f_sync();
  if (lane == 0) { out[24] = numSmp; out[25] = (double)np; }
  {
    double need = numSmp * P.ratio_of_collinear_pts;
    if (need < 10.0) need = 10.0;
    if (np < need) { if (lane == 0) *flag = 1; return; }                                     // lineslam.cpp:289
  }
  const int n = np;
  for (int i = lane; i < n; i += 64) {
    double cov[9], Wsq[3], DU[9];
    f_pt_cov(&S.pos[3 * i], c.K[0], P, cov);
    f_whiten(cov, DU, Wsq);
#pragma unroll
    for (int k = 0; k < 9; k++) S.DU[9 * i + k] = DU[k];
    S.idx[i] = i;
  }
  f_sync();
  // ---- extract3dline_mahdist (utils.cpp:343-427)
  const double thr = P.pt2line_mahdist_extractline;
  const uint64_t stream = LF_STREAM_LINE3D(b.frame_ids[f], cand);
  int maxIter = P.ransac_iters_extract_line, half = (int)(n * (n - 1) * 0.5);
  if (half < maxIter) maxIter = half;
  u64 best0 = 0, best1 = 0, ctr = 0;
  int nbest = 0, bestA = 0, bestB = 0;
  for (int iter = 0; iter < maxIter; iter++) {
    if (lane == 0) {   // random_unique(indexes, 2): partial Fisher-Yates, state carried over
      int r = 0 + (int)(lf_rand31(P.rng_seed, stream, ctr) % (uint32_t)n);
      int t = S.idx[0]; S.idx[0] = S.idx[r]; S.idx[r] = t;
      r = 1 + (int)(lf_rand31(P.rng_seed, stream, ctr + 1) % (uint32_t)(n - 1));
      t = S.idx[1]; S.idx[1] = S.idx[r]; S.idx[r] = t;
    }
    ctr += 2;
    f_sync();
    int ia = S.idx[0], ib = S.idx[1];
    f_sync();
    double A[3], B[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { A[k] = S.pos[3 * ia + k]; B[k] = S.pos[3 * ib + k]; }
    double dn = lf_sqrt((B[0] - A[0]) * (B[0] - A[0]) + (B[1] - A[1]) * (B[1] - A[1]) + (B[2] - A[2]) * (B[2] - A[2]));
    if (dn < F_EPS) continue;
    u64 m0, m1;
    f_inliers(S, n, A, B, thr, &m0, &m1);
    int nc = __popcll(m0) + __popcll(m1);
    if (nc > nbest) {
      if (f_verify3d(S, m0, m1, A, B, P)) { best0 = m0; best1 = m1; nbest = nc; bestA = ia; bestB = ib; }
    }
    if (nbest > n * 0.9) break;
  }
  double LA[3] = {0, 0, 0}, LB[3] = {0, 0, 0};
  if (nbest >= 2) {
    double mm[3], dd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { mm[k] = (S.pos[3 * bestA + k] + S.pos[3 * bestB + k]) * 0.5; dd[k] = S.pos[3 * bestB + k] - S.pos[3 * bestA + k]; }
    for (;;) {
      double tm[3], td[3], q2[3];
      f_line3d_svd(S, best0, best1, nbest, tm, td);
#pragma unroll
      for (int k = 0; k < 3; k++) q2[k] = tm[k] + td[k];
      u64 m0, m1;
      f_inliers(S, n, tm, q2, thr, &m0, &m1);
      int nc = __popcll(m0) + __popcll(m1);
      if (nc > nbest) {
        best0 = m0; best1 = m1; nbest = nc;
#pragma unroll
        for (int k = 0; k < 3; k++) { mm[k] = tm[k]; dd[k] = td[k]; }
      } else break;
    }
    bool in0 = (best0 >> lane) & 1ull, in1 = (best1 >> lane) & 1ull;
    const double *x0 = &S.pos[3 * lane], *x1 = &S.pos[3 * (lane + 64)];
    double d0 = in0 ? (x0[0] - mm[0]) * dd[0] + (x0[1] - mm[1]) * dd[1] + (x0[2] - mm[2]) * dd[2] : 0.0;
    double d1 = in1 ? (x1[0] - mm[0]) * dd[0] + (x1[1] - mm[1]) * dd[1] + (x1[2] - mm[2]) * dd[2] : 0.0;
    int first = best0 ? __builtin_ctzll(best0) : 64 + __builtin_ctzll(best1);
    double vmin = 100.0; int imin = 1 << 30;
    if (in0 && d0 < vmin) { vmin = d0; imin = lane; }
    if (in1 && d1 < vmin) { vmin = d1; imin = lane + 64; }
    f_argmin(vmin, imin);
    if (imin == (1 << 30)) imin = first;
    double vmax = -100.0; int imax = 1 << 30;
    if (in0 && d0 > vmax) { vmax = d0; imax = lane; }
    if (in1 && d1 > vmax) { vmax = d1; imax = lane + 64; }
    f_argmax(vmax, imax);
    if (imax == (1 << 30)) imax = first;
#pragma unroll
    for (int k = 0; k < 3; k++) { LA[k] = S.pos[3 * imin + k]; LB[k] = S.pos[3 * imax + k]; }
  }
  if (lane == 0) { out[26] = (double)nbest; out[29] = LA[0]; out[30] = LA[1]; out[31] = LA[2]; }
  bool have = (nbest / numSmp > P.ratio_of_collinear_pts) &&
              (lf_sqrt((LA[0] - LB[0]) * (LA[0] - LB[0]) + (LA[1] - LB[1]) * (LA[1] - LB[1]) + (LA[2] - LB[2]) * (LA[2] - LB[2])) > P.line3d_length_thresh);
  if (!have) { if (lane == 0) *flag = 1; return; }                                          // lineslam.cpp:302-307
  // ---- hand the supporting points (line.pts, in list order) to the MLE kernel: as the inlier mask over the candidate's valid
  // samples (bit i of word i / 64 = valid sample i); k_mle re-derives the points from the segment with f_sample_point
  if (lane == 0) { u64 *mk = b.cand_mask + ((size_t)f * c.cand_cap + cand) * 2; mk[0] = best0; mk[1] = best1; }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { out[k] = LA[k]; out[3 + k] = LB[k]; }
    *flag = 2;
  }
}

// ordered compaction (lineslam.cpp:332-341): one wavefront per frame
__global__ void __launch_bounds__(64) k_records(FrontConsts c, FrontBuffers b) {
  const int f = blockIdx.x, lane = f_lane();
  int nseg = b.nsegs[f];
  if (nseg > c.seg_cap) nseg = c.seg_cap;
  if (nseg > c.cand_cap) nseg = c.cand_cap;
  const int *flag = b.cand_flag + (size_t)f * c.cand_cap;
  lf_line_record *recs = b.recs + (size_t)f * c.line_cap;
  int *list0 = b.mle_list + (size_t)f * 3 * c.line_cap, *list1 = list0 + c.line_cap, *list2 = list1 + c.line_cap;
  int base = 0, n0 = 0, n1 = 0, n2 = 0;
  for (int s0 = 0; s0 < nseg; s0 += 64) {
    int s = s0 + lane;
    bool have = s < nseg && flag[s] == 2;
    u64 m = __ballot(have);
    int lid = base + __popcll(m & f_lt());
    {   // MLE work lists by number of RANSAC support points: <= 16, 17..32, more
      bool kept = have && lid < c.line_cap;
      int nsup = kept ? (int)b.cand_out[((size_t)f * c.cand_cap + s) * LF_CAND_STRIDE + 26] : 0;
      // (a G = 16 variant, four lines per wavefront for <= 16 points, was measured slower: list 0 stays empty)
#if LF_MLE_PAIR64   // list 0: 33 .. 64 support points, two lines per wavefront with two rows per lane
      bool small = kept && nsup > 32 && nsup <= 64, mid = kept && nsup <= 32, large = kept && nsup > 64;
#else
      bool small = kept && nsup <= LF_MLE_SMALL, mid = kept && nsup > LF_MLE_SMALL && nsup <= 32, large = kept && nsup > 32;
#endif
      u64 ms = __ballot(small), mm = __ballot(mid), ml = __ballot(large);
      if (small) list0[n0 + __popcll(ms & f_lt())] = lid;
      if (mid) list1[n1 + __popcll(mm & f_lt())] = lid;
      if (large) list2[n2 + __popcll(ml & f_lt())] = lid;
      n0 += __popcll(ms); n1 += __popcll(mm); n2 += __popcll(ml);
    }
    if (have && lid < c.line_cap) {
      lf_line_record *R = &recs[lid];
      const double *sg = b.segs + ((size_t)f * c.seg_cap + s) * 5;
      double p0 = sg[0], p1 = sg[1], q0 = sg[2], q1 = sg[3];
      R->p[0] = p0; R->p[1] = p1; R->q[0] = q0; R->q[1] = q1;
      // complineEq2d (lineslam.h:139-150)
      double l0 = p1 * 1.0 - 1.0 * q1, l1 = 1.0 * q0 - p0 * 1.0, l2 = p0 * q1 - p1 * q0;
      double nn = lf_sqrt(l0 * l0 + l1 * l1);
      R->lineEq2d[0] = l0 / nn; R->lineEq2d[1] = l1 / nn; R->lineEq2d[2] = l2 / nn;
      R->r[0] = 0; R->r[1] = 0;
      R->lid = lid;
      R->seg = s;
    }
    base += __popcll(m);
  }
  if (lane == 0) {
    b.nlines[f] = base;        // (may exceed line_cap: the frame is over capacity, the first line_cap records are complete)
    b.mle_cnt[3 * f] = n0; b.mle_cnt[3 * f + 1] = n1; b.mle_cnt[3 * f + 2] = n2;
  }
}

// ----------------------------------------------------------------------------------------------
// MLEstimateLine3d + MleLine3dCov (utils.cpp:954-1050, 1086-1159): ONE WAVEFRONT PER 3D LINE.
// Supporting points live in LDS (one or two per lane); residuals and Jacobian rows are evaluated
// lane-parallel; J^T J / J^T e use one accumulator per lane walking the rows in levmar's order.
#define MLE_N 104
#define MLE_ROW_DOUBLES 20   // LDS doubles per support point: pos 3, DU 9, Jacobian row 6, e 1, scratch 1
#define MLE_ACC_DOUBLES 88   // + per group: 8 remainder squares of the ordered sums, then the published normal equations: M [6][8]
                             //   (J^T J symmetric in columns 0..5, J^T e in column 6, zeros in column 7) and D [6], the plain diagonal
#define MLE_M(S) ((S).accs)
#define MLE_D(S) ((S).accs + 48)
#define MLE_CI(S) ((S).accs + 56)   // [2][9]: inverse covariances of the two end points (read by the two lanes that own those rows)
// a group's LDS block is ROWS * MLE_ROW_DOUBLES + MLE_ACC_DOUBLES doubles; the accumulator area starts 8 doubles in (behind the
// eight remainder slots of scr), so what it holds -- M 48, D 6 (+2 pad), CI 18 -- must fit MLE_ACC_DOUBLES - 8
static_assert(8 + 56 + 18 <= MLE_ACC_DOUBLES, "scr remainder slots + M + D + CI must fit the group's accumulator area");
static_assert(48 + 6 <= 56, "MLE_D (the plain diagonal) must end before MLE_CI");
static_assert(MLE_ROW_DOUBLES == 3 + 9 + 6 + 1 + 1, "LDS row = pos 3, DU 9, Jacobian row 6, e 1, scr 1 (f_mstate_bind)");
struct MState {   // views into the group's LDS block, sized for the kernel variant's row capacity
  double *pos;    // [rows][3]
  double *DU;     // [rows][9]
  double *jac;    // [rows][6]
  double *e;      // [rows]
  double *scr;    // [rows] operands of row-ordered sums (two-lines-per-wavefront variant)
  double *accs;   // [MLE_ACC_DOUBLES]
};
__device__ __forceinline__ void f_mstate_bind(MState &S, double *lds, int rows) {
  S.pos = lds; S.DU = S.pos + rows * 3; S.jac = S.DU + rows * 9; S.e = S.jac + rows * 6;
  S.scr = S.e + rows; S.accs = S.scr + rows + 8;    // scr: [rows] block rows, [8] remainder rows
}

// ---- lane groups.  A 3D line with n support points is handled by a GROUP of G lanes: G = 64 (one line per
// wavefront, two row slots per lane, n <= 104) or G = 32 (two lines per wavefront, one row per lane, n <= 32: 80 % of
// the lines).  The code also instantiates for G = 16 (two accumulator slots per lane); measured slower, not launched.  Values that are "uniform" for a line are uniform within its group; control flow
// that depends on them simply diverges between the two groups of a wavefront.
template <int G, int ROWS_> struct MleCfgT {
  static constexpr int NG = 64 / G;
  static constexpr int ROWS = ROWS_;
  static constexpr int SLOTS = (ROWS + G - 1) / G;
};
struct MleGroup { int gbase, glane; };   // first lane of my group, my lane inside it
template <int G> __device__ __forceinline__ double g_get(double v, const MleGroup &g, int idx) {   // v of group lane idx
  if constexpr (G == 64) return f_rl64(v, idx);
  else return __shfl(v, g.gbase + idx, 64);
}
template <int G> __device__ __forceinline__ void g_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
template <int G> __device__ __forceinline__ void g_argmin(double &v, int &i) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
template <int G> __device__ __forceinline__ void g_argmax(double &v, int &i) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// ---- 6 x 6 linear systems with ONE COLUMN PER LANE.  Group lane j < 6 holds column j of the matrix in c[0..5], group
// lanes 6 .. 6 + NRHS - 1 one right-hand side each, the other lanes zeros.  Per elimination step the pivot column is
// broadcast to the group (readlane for a whole wavefront, ds_bpermute for half-wavefront groups), the pivot search, the
// reciprocal and the multipliers are then the same in every lane, and each lane applies the row operations to its own
// column: c[i] -= f_i * c[k].  Every entry sees exactly the operations of the sequential elimination (lf_linalg.h), so the
// results are bit-equal -- at 6 doubles of state per lane instead of the 42 (or 54) of a replicated A | B.
//   NETLIB = true : levmar's AX_EQ_B_LU, LAPACK order (lf_lu6: dgetf2 + dgetrs -- the back-substitution walks the columns
//                   from the last one, subtracting x_k U(i,k) from the rows above, and DIVIDES by the diagonal)
//   NETLIB = false: cv::Mat::inv's LU (lf_solve6: row-oriented back-substitution, multiplication by the reciprocal pivot)
// Returns 0 (uniform in the group) for a singular matrix; the solutions replace the right-hand sides.
template <int G, int NRHS, bool NETLIB>
__device__ __forceinline__ int f_lu6_cols(double (&c)[6], const MleGroup &g) {
  int ok = 1;
  double dg[6];                                  // NETLIB: the pivots; else their reciprocals
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double col[6];
#pragma unroll
    for (int i = k; i < 6; i++) col[i] = g_get<G>(c[i], g, k);
    int piv = k;
    double big = lf_fabs(col[k]);
#if LF_PIVOT_FAST
    // the pivot is the FIRST maximum of |col[k..5]|: it is row k unless a later entry is strictly larger -- one running maximum and
    // one compare say so; the search with its selects runs only in the wavefronts where some lane has to swap
    if (k < 5) {
      double mx = lf_fabs(col[k + 1]);
#pragma unroll
      for (int i = k + 2; i < 6; i++) mx = __builtin_fmax(mx, lf_fabs(col[i]));
      if (LF_ANY(mx > big)) {
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
          const double v = lf_fabs(col[i]);
          if (v > big) { big = v; piv = i; }
        }
      }
    }
#else
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double v = lf_fabs(col[i]);
      if (v > big) { big = v; piv = i; }
    }
#endif
    if constexpr (G == 64) piv = __builtin_amdgcn_readfirstlane(piv);
    if (!(big > 0.0)) ok = 0;
    if (LF_ANY(piv != k)) {
#pragma unroll
      for (int i = k + 1; i < 6; i++) {
        const bool sw = (i == piv);
        const double a = c[k], b2 = c[i], ca = col[k], cb = col[i];
        c[k] = sw ? b2 : a; c[i] = sw ? a : b2;
        col[k] = sw ? cb : ca; col[i] = sw ? ca : cb;
      }
    }
    const double pv = col[k];
    if (NETLIB && LF_ANY(!(big >= LF_LU_SFMIN))) {     // dgetf2: a pivot below sfmin divides instead (never on this path's matrices)
      const bool tiny = !(big >= LF_LU_SFMIN);
      const double r = 1.0 / pv;
#pragma unroll
      for (int i = k + 1; i < 6; i++) { const double f = tiny ? col[i] / pv : col[i] * r; c[i] -= f * c[k]; }
      dg[k] = pv;
    } else {
      const double r = 1.0 / pv;
#pragma unroll
      for (int i = k + 1; i < 6; i++) { const double f = col[i] * r; c[i] -= f * c[k]; }
      dg[k] = NETLIB ? pv : r;
    }
  }
  if constexpr (NETLIB) {
    // (column oriented: step k changes the rows above k -- only in the right-hand-side lanes, the matrix lanes must
    // keep U for the steps that follow)
    const bool rhs = g.glane >= 6;
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      double u[6];
#pragma unroll
      for (int i = 0; i < k; i++) u[i] = g_get<G>(c[i], g, k);
      if (rhs) {
        const double x = c[k] / dg[k];
        c[k] = x;
#pragma unroll
        for (int i = 0; i < k; i++) c[i] -= x * u[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      double s = c[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) { const double u = g_get<G>(c[i], g, k); s -= u * c[k]; }
      c[i] = s * dg[i];
    }
  }
  return ok;
}

// costFun_MLEstimateLine3d (utils.cpp:954-978): residuals of the rows this lane owns, in registers
template <int G, int RW>
__device__ __forceinline__ void f_mle_cost(const MState &S, const MleGroup &g, int n, int e1, int e2, const double *ci1,
                                           const double *ci2, const double *p, double *out /* [SLOTS] */) {
#pragma unroll
  for (int h = 0; h < MleCfgT<G, RW>::SLOTS; h++) {
    int i = g.glane + G * h;
    double r = 0.0;
    if (i < n) {
      int pi = i;
      if (i == e1 || i == e2) {
        const double *ci = (i == e1) ? ci1 : ci2;
        const double *e = (i == e1) ? p : p + 3;
        double v[3], t[3];
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = e[k] - S.pos[3 * pi + k];
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = v[0] * ci[0 * 3 + k] + v[1] * ci[1 * 3 + k] + v[2] * ci[2 * 3 + k];
        r = t[0] * v[0] + t[1] * v[1] + t[2] * v[2];
      } else
        r = f_mah(&S.pos[3 * pi], &S.DU[9 * pi], p, p + 3);
    }
    out[h] = r;
  }
}
// levmar's LEVMAR_L2NRMXMY (misc_core.c; called at lm_core.c:555 / :743): the sum of v[i]^2 over the rows with FOUR running
// sums -- blocks of eight from the top of the vector downwards (rows 8b+7-c and 8b+3-c go to sum c), then the remainder
// through the fall-through switch (row blockn + t of a remainder of r rows goes to sum (7 - r + t) mod 4) -- returned as
// sum0+sum1+sum2+sum3.  Group lane l runs chain l mod 4 from the squares published in LDS; every lane then adds the four.
template <int G, int RW>
__device__ __forceinline__ double f_ordered_sumsq(const MState &S, const double *v, const MleGroup &g, int n) {
  // scr: squares of the rows of the complete blocks of eight at their row index (the rows from blockn on stay zero: set once
  // in f_levmar6, never written), the squares of the remainder rows n - r .. n - 1 in the eight slots behind the row area
  // (slots >= r stay zero).  All operands of a chain are loaded before its additions start; blocks of zeros in front of the
  // first real block add +0.0 to a sum that is still +0.0.
  constexpr int ROWS = MleCfgT<G, RW>::ROWS, NB = ROWS / 8, NCH = (NB + 3) / 4;
  const int c = g.glane & 3, nb = n >> 3, r = n & 7, blockn = nb << 3;
#pragma unroll
  for (int h = 0; h < MleCfgT<G, RW>::SLOTS; h++) {
    const int i = g.glane + G * h;
    if (i < n) S.scr[(i < blockn) ? i : ROWS + (i - blockn)] = v[h] * v[h];
  }
  // the remainder slots behind the last remainder row are zeroed on every call (the sum must not depend on a previous call with
  // another n, or on anyone else having used the area); rows blockn .. ROWS-1 of the block area are zero from f_levmar6's clear
  // and are never written: n only selects which of them are overwritten by squares, and every index < blockn IS overwritten
  if (g.glane < 8 && g.glane >= r) S.scr[ROWS + g.glane] = 0.0;
  g_order<G>();
  const int t0 = (c - (7 - r)) & 3;
  const double r0 = S.scr[ROWS + t0], r1 = S.scr[ROWS + t0 + 4];
  double s = 0.0;
#pragma unroll
  for (int cb = NCH - 1; cb >= 0; --cb) {          // four blocks of eight per trip, from the top of the vector downwards
    if (cb * 4 < nb) {
      double q[8];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int b = cb * 4 + 3 - k;
        if (b < NB) { q[2 * k] = S.scr[8 * b + 7 - c]; q[2 * k + 1] = S.scr[8 * b + 3 - c]; }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (cb * 4 + 3 - k < NB) { s += q[2 * k]; s += q[2 * k + 1]; }
    }
  }
  s += r0;
  s += r1;
  g_order<G>();
  const double s0 = g_get<G>(s, g, 0), s1 = g_get<G>(s, g, 1), s2 = g_get<G>(s, g, 2), s3 = g_get<G>(s, g, 3);
  return s0 + s1 + s2 + s3;
}

#ifdef LF_MLE_PROFILE   // LF_EXTRA_CFLAGS=-DLF_MLE_PROFILE=32 (or 64: the lane-group width): s_memtime per LM phase of work item 7 of frame 0, printed
__device__ unsigned long long g_mprof[16];
#define MT(k) do { unsigned long long tn = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 7 && blockIdx.y == 0 && f_lane() == 0 && G == LF_MLE_PROFILE) g_mprof[k] += tn - tprev; tprev = tn; } while (0)
#else
#define MT(k) do { } while (0)
#endif
// dlevmar_dif (external/levmar-2.6/lm_core.c:438-846) for m = 6, x = 0, restated for one wavefront.
// Every lane owns the rows lane and lane+64: hx / wrk / wrk2 stay in registers, sums over the rows are taken
// in row order through readlane; the Jacobian and e live in LDS because J^T J and J^T e use one accumulator per
// lane that walks ALL rows l = n-1 .. 0 exactly as lm_core.c:581-591.  The 6x6 system is identical in all
// lanes (lf_solve6_u: pivot decisions are scalar branches).
template <int G, int RW>
__device__ int f_levmar6(MState &S, const MleGroup &g, int n, int e1, int e2, const double *ci1, const double *ci2,
                         double *p, int itmax, int *stop_out) {
  constexpr int SL = MleCfgT<G, RW>::SLOTS;
  const int m = 6, lane = g.glane;
  const double tau = 1E-03, eps1 = 1E-10, eps2 = 1E-20, eps2_sq = 1E-20 * 1E-20, eps3 = 1E-20, delta = 1E-06;
  double Dp[6];
  double hx[SL], ev[SL], wrk[SL], wrk2[SL];
  double mu = 0, tmp, p_eL2, jacTe_inf = 0, pDp_eL2, p_L2 = 0, Dp_L2 = DBL_MAX, dF, dL;
  int nu, nu2, stop = 0, K = 10, updjac = 0, updp = 1, newjac = 0, k;
  // accumulator ownership: entries 0..20 lower triangle (i,j), 21..26 J^T e; entry a lives in group lane a % G,
  // slot a / G (one slot for G >= 32, two for G = 16)
  constexpr int NACC = (27 + G - 1) / G;
  int ai[NACC], aj[NACC];
#pragma unroll
  for (int q = 0; q < NACC; q++) {
    int a = lane + G * q;
    ai[q] = 0; aj[q] = 0;
    if (a < 21) { while ((ai[q] + 1) * (ai[q] + 2) / 2 <= a) ai[q]++; aj[q] = a - ai[q] * (ai[q] + 1) / 2; }
    else if (a < 27) { ai[q] = a - 21; aj[q] = -1; }
  }
#ifdef LF_MLE_PROFILE
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  // rows n .. n8-1 (n8 = n rounded up to 8) of jac / e / scr hold zeros: the row-ordered sums below then run in
  // full groups of eight (adding +0.0 terms changes nothing)
  const int n8 = (n + 7) & ~7;
#pragma unroll
  for (int h = 0; h < SL; h++) {
    int i = lane + G * h;
    if (i >= n && i < n8) {
#pragma unroll
      for (int j = 0; j < 6; j++) S.jac[i * m + j] = 0.0;
      S.e[i] = 0.0;
    }
  }
  {   // the operand area of the ordered sums starts as zeros (f_ordered_sumsq), column 7 of M stays zero
    constexpr int ROWS = MleCfgT<G, RW>::ROWS;
    for (int i = lane; i < ROWS + 8; i += G) S.scr[i] = 0.0;
    if (lane < 6) MLE_M(S)[lane * 8 + 7] = 0.0;
  }
  f_mle_cost<G, RW>(S, g, n, e1, e2, ci1, ci2, p, hx);
#pragma unroll
  for (int h = 0; h < SL; h++) { ev[h] = 0.0 - hx[h]; int i = lane + G * h; if (i < n) S.e[i] = ev[h]; }
  p_eL2 = f_ordered_sumsq<G, RW>(S, ev, g, n);
  if (!(lf_fabs(p_eL2) <= DBL_MAX)) stop = 7;
  nu = 20;
  for (k = 0; k < itmax && !stop; ++k) {
    if (p_eL2 <= eps3) { stop = 6; break; }
    MT(0);
    if ((updp && nu > 16) || updjac == K) {
      for (int j = 0; j < m; ++j) {           // forward differences (misc_core.c:137-171)
        double d = 1E-04 * p[j], t;
        d = lf_fabs(d);
        if (d < delta) d = delta;
        t = p[j]; p[j] += d;
        f_mle_cost<G, RW>(S, g, n, e1, e2, ci1, ci2, p, wrk);
        p[j] = t;
        d = 1.0 / d;
#pragma unroll
        for (int h = 0; h < SL; h++) { int i = lane + G * h; if (i < n) S.jac[i * m + j] = (wrk[h] - hx[h]) * d; }
      }
      nu = 2; updjac = 0; updp = 0; newjac = 1;
    }
    MT(1);
    if (newjac) {
      newjac = 0;
      g_order<G>();                           // Jacobian rows and e of all lanes visible
      double acc[NACC];
#pragma unroll
      for (int q = 0; q < NACC; q++) {
        acc[q] = 0.0;
        if (lane + G * q < 27) {
          const int ai_ = ai[q], aj_ = aj[q];
          const double *colB = (aj_ >= 0) ? (S.jac + aj_) : S.e;   // second factor: J[l][aj] or e[l]
          const int strideB = (aj_ >= 0) ? m : 1;
          double s = 0.0;
          int l = n8;                           // (the zero rows n .. n8-1 come first and leave s = 0)
          for (; l >= 8; l -= 8) {              // eight rows per trip: loads first, additions in levmar's order
            double av[8], bv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { av[k] = S.jac[(l - 1 - k) * m + ai_]; bv[k] = colB[(l - 1 - k) * strideB]; }
#pragma unroll
            for (int k = 0; k < 8; k++) s += bv[k] * av[k];
          }
          acc[q] = s;
        }
      }
      MT(2);
      {   // the accumulator lanes publish their entries; J^T J stays in LDS (read where the system is set up), the
          // group keeps J^T e and the diagonal in registers
#pragma unroll
        for (int q = 0; q < NACC; q++) {
          const int a = lane + G * q;
          if (a < 21) {                            // J^T J: both halves of the square, the diagonal also into D
            MLE_M(S)[ai[q] * 8 + aj[q]] = acc[q];
            MLE_M(S)[aj[q] * 8 + ai[q]] = acc[q];
            if (ai[q] == aj[q]) MLE_D(S)[ai[q]] = acc[q];
          } else if (a < 27) MLE_M(S)[ai[q] * 8 + 6] = acc[q];      // J^T e: column 6
        }
        g_order<G>();
      }
      p_L2 = jacTe_inf = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (jacTe_inf < (tmp = lf_fabs(MLE_M(S)[i * 8 + 6]))) jacTe_inf = tmp;
        p_L2 += p[i] * p[i];
      }
    }
    MT(3);
    if (jacTe_inf <= eps1) { Dp_L2 = 0.0; stop = 1; break; }
    if (k == 0) {
      tmp = DBL_MIN;
#pragma unroll
      for (int i = 0; i < 6; ++i) { const double dgi = MLE_D(S)[i]; if (dgi > tmp) tmp = dgi; }
      mu = tau * tmp;
    }
    int issolved;
    {   // the augmented normal equations (J^T J + mu I) Dp = J^T e (AX_EQ_B_LU, lm_core.c:706): the diagonal lanes put
        // D + mu I into the published square, then group lane j takes column j of it (j = 6: J^T e, j >= 7: zeros)
      double cl[6];
      if (lane < 6) MLE_M(S)[lane * 9] = MLE_D(S)[lane] + mu;
      g_order<G>();
      const double *colp = MLE_M(S) + ((lane < 7) ? lane : 7);
#pragma unroll
      for (int i = 0; i < 6; i++) cl[i] = colp[i * 8];
      issolved = f_lu6_cols<G, 1, true>(cl, g);
#pragma unroll
      for (int i = 0; i < 6; i++) Dp[i] = g_get<G>(cl[i], g, 6);
    }
    MT(4);
    if (issolved) {
      Dp_L2 = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) { tmp = Dp[i]; Dp_L2 += tmp * tmp; }
      if (Dp_L2 <= eps2_sq * p_L2) { stop = 2; break; }
      if (Dp_L2 >= (p_L2 + eps2) / (1E-12 * 1E-12)) { stop = 4; break; }
      {
        double pDp[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pDp[i] = p[i] + Dp[i];
        f_mle_cost<G, RW>(S, g, n, e1, e2, ci1, ci2, pDp, wrk);
      }
      MT(5);
#pragma unroll
      for (int h = 0; h < SL; h++) wrk2[h] = 0.0 - wrk[h];
      pDp_eL2 = f_ordered_sumsq<G, RW>(S, wrk2, g, n);
      MT(6);
      if (!(lf_fabs(pDp_eL2) <= DBL_MAX)) { stop = 7; break; }
      dF = p_eL2 - pDp_eL2;
      if (updp || dF > 0) {                       // Broyden rank-one update, row-parallel
        g_order<G>();                             // the accumulator lanes are done reading the old rows
#pragma unroll
        for (int h = 0; h < SL; h++) {
          int i = lane + G * h;
          if (i < n) {
            double t2 = 0.0;
#pragma unroll
            for (int l = 0; l < 6; ++l) t2 += S.jac[i * m + l] * Dp[l];
            t2 = (wrk[h] - hx[h] - t2) / Dp_L2;
#pragma unroll
            for (int j = 0; j < 6; ++j) S.jac[i * m + j] += t2 * Dp[j];
          }
        }
        ++updjac; newjac = 1;
      }
      MT(7);
      dL = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) dL += Dp[i] * (mu * Dp[i] + MLE_M(S)[i * 8 + 6]);
      if (dL > 0.0 && dF > 0.0) {
        tmp = (2.0 * dF / dL - 1.0);
        tmp = 1.0 - tmp * tmp * tmp;
        mu = mu * ((tmp >= 0.3333333334) ? tmp : 0.3333333334);
        nu = 2;
#pragma unroll
        for (int i = 0; i < 6; ++i) p[i] = p[i] + Dp[i];          // (== pDp: the same addition)
#pragma unroll
        for (int h = 0; h < SL; h++) { int i = lane + G * h; hx[h] = wrk[h]; if (i < n) S.e[i] = wrk2[h]; }
        p_eL2 = pDp_eL2;
        updp = 1;
        continue;
      }
    }
    mu *= nu;
    nu2 = nu << 1;
    if (nu2 <= nu) { stop = 5; break; }
    nu = nu2;                                   // (the diagonal is set up again from the sums: nothing to restore)
  }
  if (k >= itmax) stop = 3;
  *stop_out = stop;
#ifdef LF_MLE_PROFILE
  MT(8);
  if (blockIdx.x == 7 && blockIdx.y == 0 && f_lane() == 0 && G == LF_MLE_PROFILE) {
    printf("k_mle prof (kcycles) n=%d iters=%d: loop/tail %.1f fdjac %.1f acc %.1f gather %.1f solve %.1f cost %.1f sumsq %.1f broyden %.1f end %.1f\n", n, k,
           g_mprof[0] / 1e3, g_mprof[1] / 1e3, g_mprof[2] / 1e3, g_mprof[3] / 1e3, g_mprof[4] / 1e3, g_mprof[5] / 1e3, g_mprof[6] / 1e3, g_mprof[7] / 1e3, g_mprof[8] / 1e3);
    for (int i = 0; i < 16; i++) g_mprof[i] = 0;
  }
#endif
  return (stop != 4 && stop != 7) ? k : -1;
}
#undef MT


// One 3D line (record lid of frame f) from its RANSAC support points to the refined end points and their covariances:
// the body of k_mle for a lane group of G lanes with RW row slots (RW / G rows per lane).
template <int G, int RW>
__device__ __forceinline__ void f_mle_line(const FrontConsts &c, const FrontBuffers &b, int f, int lid, double *lds, const MleGroup &g) {
  typedef MleCfgT<G, RW> Cfg;
  const int lane = g.glane;
  MState S;
  f_mstate_bind(S, lds, Cfg::ROWS);
  const lf_params &P = c.P;
  lf_line_record *R = b.recs + (size_t)f * c.line_cap + lid;
  const int seg = R->seg;
  double *out = b.cand_out + ((size_t)f * c.cand_cap + seg) * LF_CAND_STRIDE;
  int ns = (int)out[26];
  if (ns > Cfg::ROWS) ns = Cfg::ROWS;
  double LA[3] = {out[0], out[1], out[2]}, LB[3] = {out[3], out[4], out[5]};
  if (c.pts_rows) {            // lf_mle_lines: the caller's supporting points, row block `seg` of b.pts
    const double *pts = b.pts + (size_t)seg * (LF_MAX_SAMPLES * 3);
    for (int i = lane; i < ns; i += G)
#pragma unroll
      for (int k = 0; k < 3; k++) S.pos[3 * i + k] = pts[3 * i + k];
  } else {                     // the inliers of k_line3d's RANSAC, in sample order: valid samples are numbered as they come, the mask picks
    const double *sg = b.segs + ((size_t)f * c.seg_cap + seg) * 5;
    const double pa = sg[0], pb = sg[1], qc = sg[2], qd = sg[3], numSmp = out[24];
    const float *depth = b.depth + (size_t)f * b.depth_frame_stride;
    const u64 *mk = b.cand_mask + ((size_t)f * c.cand_cap + seg) * 2;
    const u64 m0 = mk[0], m1 = mk[1];
    const u64 gmask = (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << g.gbase), glt = (1ull << lane) - 1ull;
    int nvalid = 0, nrow = 0;
    for (int j0 = 0; (double)j0 <= numSmp && j0 < LF_MAX_SAMPLES; j0 += G) {
      double X = 0, Y = 0, Z = 0;
      const bool ok = f_sample_point(c, b, depth, pa, pb, qc, qd, numSmp, j0 + lane, &X, &Y, &Z);
      const u64 mv = (__ballot(ok) & gmask) >> g.gbase;
      const int vi = nvalid + __popcll(mv & glt);
      const bool in = ok && (((vi < 64 ? m0 >> vi : m1 >> (vi - 64)) & 1ull) != 0ull);
      const u64 mi = (__ballot(in) & gmask) >> g.gbase;
      const int row = nrow + __popcll(mi & glt);
      if (in && row < ns) { S.pos[3 * row] = X; S.pos[3 * row + 1] = Y; S.pos[3 * row + 2] = Z; }
      nvalid += __popcll(mv); nrow += __popcll(mi);
    }
  }
  g_order<G>();
  for (int i = lane; i < ns; i += G) {
    double pos[3] = {S.pos[3 * i], S.pos[3 * i + 1], S.pos[3 * i + 2]}, cov[9], DU[9], Wsq[3];
    f_pt_cov(pos, c.K[0], P, cov);
    f_whiten(cov, DU, Wsq);
#pragma unroll
    for (int k = 0; k < 9; k++) S.DU[9 * i + k] = DU[k];
  }
  g_order<G>();
  int e1, e2;
  {   // extremities along the line (utils.cpp:985-999), first occurrence on ties
    double AmB[3] = {LA[0] - LB[0], LA[1] - LB[1], LA[2] - LB[2]};
    double vmin = 100.0, vmax = -100.0;
    int imin = 1 << 30, imax = 1 << 30;
    for (int h = 0; h < Cfg::SLOTS; h++) {
      int i = lane + G * h;
      if (i < ns) {
        const double *x = &S.pos[3 * i];
        double dp = (x[0] - LA[0]) * AmB[0] + (x[1] - LA[1]) * AmB[1] + (x[2] - LA[2]) * AmB[2];
        if (dp < vmin) { vmin = dp; imin = i; }
        if (dp > vmax) { vmax = dp; imax = i; }
      }
    }
    g_argmin<G>(vmin, imin);
    g_argmax<G>(vmax, imax);
    e1 = (imin == (1 << 30)) ? 0 : imin;
    e2 = (imax == (1 << 30)) ? 0 : imax;
    if (e1 > e2) { int t = e1; e1 = e2; e2 = t; }
  }
  double para[6];
  {   // the inverse covariances of the two end points live in LDS: only the two lanes that own those rows read them
    double cov[9], ci[9];
    f_pt_cov(&S.pos[3 * e1], c.K[0], P, cov);
    lf_inv3(cov, ci);
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 9; k++) MLE_CI(S)[k] = ci[k];
    f_pt_cov(&S.pos[3 * e2], c.K[0], P, cov);
    lf_inv3(cov, ci);
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 9; k++) MLE_CI(S)[9 + k] = ci[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { para[k] = S.pos[3 * e1 + k]; para[3 + k] = S.pos[3 * e2 + k]; }
  }
  g_order<G>();
  int stop = 0;
  int nit = f_levmar6<G, RW>(S, g, ns, e1, e2, MLE_CI(S), MLE_CI(S) + 9, para, P.line3d_mle_iter_num, &stop);
  // ---- MleLine3dCov (utils.cpp:1138-1159): H = J^T J in point order, cov = H^-1.  Every lane forms the 3x6 Jacobian
  // of its own rows, the rows are published in LDS (over pos / DU, which are dead by then), and 21 accumulator lanes
  // walk them in the reference's order (point by point, residual row by row); H is symmetric term by term.
  double covA[9], covB[9];
  {
    constexpr int SL = Cfg::SLOTS;
    double J[SL][18];
#pragma unroll
    for (int h = 0; h < SL; h++) {
      const int i = lane + G * h;
#pragma unroll
      for (int k = 0; k < 18; k++) J[h][k] = 0;
      if (i < ns) {
        if (i == e1) {
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) J[h][r * 6 + k] = -S.DU[9 * i + 3 * r + k];
        } else if (i == e2) {
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) J[h][r * 6 + 3 + k] = -S.DU[9 * i + 3 * r + k];
        } else
          f_jac_line(&S.pos[3 * i], &S.DU[9 * i], para, J[h]);
      }
    }
    g_order<G>();
    double *Jl = S.pos;                       // [row][MLE_ROW_DOUBLES], 18 used
#pragma unroll
    for (int h = 0; h < SL; h++) {
      const int i = lane + G * h;
      if (i < ns)
#pragma unroll
        for (int k = 0; k < 18; k++) Jl[i * MLE_ROW_DOUBLES + k] = J[h][k];
    }
    g_order<G>();
    constexpr int NA = (21 + G - 1) / G;      // lower-triangle entries per lane (1 for G >= 32)
#pragma unroll
    for (int q = 0; q < NA; q++) {
      const int a = lane + G * q;
      if (a < 21) {
        int hk = 0;
        while ((hk + 1) * (hk + 2) / 2 <= a) hk++;
        const int hl = a - hk * (hk + 1) / 2;
        double acc = 0.0;
        for (int i = 0; i < ns; ++i) {
          const double *row = Jl + i * MLE_ROW_DOUBLES;
          double a0 = row[hk], b0 = row[hl], a1 = row[6 + hk], b1 = row[6 + hl], a2 = row[12 + hk], b2 = row[12 + hl];
          acc += a0 * b0;
          acc += a1 * b1;
          acc += a2 * b2;
        }
        MLE_M(S)[hk * 8 + hl] = acc;
        MLE_M(S)[hl * 8 + hk] = acc;
      }
    }
    g_order<G>();
  }
  // cov = H^-1 by elimination with the identity as right-hand sides; the record keeps its upper-left and lower-right
  // 3x3 blocks only (covA, covB), and the columns of an inverse are independent: two eliminations with three columns
  // each (the same operations on the columns that are kept) instead of one with six -- a third fewer live registers.
  int inv_ok;
  {   // one column per lane: H in group lanes 0..5, the six columns of the identity in lanes 6..11 (cv::Mat::inv's LU)
    double cl[6];
    const int jc = (lane < 6) ? lane : 5;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double v = MLE_M(S)[i * 8 + jc];
      if (lane >= 6) v = (lane - 6 == i) ? 1.0 : 0.0;
      cl[i] = v;
    }
    inv_ok = f_lu6_cols<G, 6, false>(cl, g);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 3; q++) { covA[3 * r + q] = g_get<G>(cl[r], g, 6 + q); covB[3 * r + q] = g_get<G>(cl[3 + r], g, 9 + q); }
  }
  g_order<G>();
  if (!inv_ok) {
#pragma unroll
    for (int i = 0; i < 9; i++) { covA[i] = lf_from_bits(0x7ff8000000000000ULL); covB[i] = covA[i]; }
  }
  // ---- results: line3d.A/B, covA/covB, rndA/rndB (RandomPoint3d ctor) into the record
  if (lane == 0) {
    double cov[9], DU[9], Wsq[3];
    for (int q = 0; q < 3; q++) { R->A[q] = para[q]; R->B[q] = para[3 + q]; out[29 + q] = LA[q]; out[q] = para[q]; out[3 + q] = para[3 + q]; }
    for (int q = 0; q < 9; q++) cov[q] = covA[q];
    for (int q = 0; q < 9; q++) { R->covA[q] = cov[q]; out[6 + q] = cov[q]; }
    f_whiten(cov, DU, Wsq);
    for (int q = 0; q < 9; q++) R->DUa[q] = DU[q];
    for (int q = 0; q < 3; q++) R->Wsa[q] = Wsq[q];
    for (int q = 0; q < 9; q++) cov[q] = covB[q];
    for (int q = 0; q < 9; q++) { R->covB[q] = cov[q]; out[15 + q] = cov[q]; }
    f_whiten(cov, DU, Wsq);
    for (int q = 0; q < 9; q++) R->DUb[q] = DU[q];
    for (int q = 0; q < 3; q++) R->Wsb[q] = Wsq[q];
    out[27] = (double)nit;
    out[28] = (double)stop;
  }
}

#ifndef LF_MLE_WAVES
#define LF_MLE_WAVES 3           // wavefronts per SIMD the register allocation of k_mle aims at
#endif
template <int G, int RW, int WHICH>
__global__ void __launch_bounds__(64, LF_MLE_WAVES) k_mle(FrontConsts c, FrontBuffers b) {
  typedef MleCfgT<G, RW> Cfg;
  __shared__ double lds_rows[Cfg::NG][Cfg::ROWS * MLE_ROW_DOUBLES + MLE_ACC_DOUBLES];
  const int f = blockIdx.y, wl = f_lane();
  MleGroup g;
  g.gbase = (wl / G) * G; g.glane = wl % G;
  // work lists of k_records: [0] unused, [1] lines with <= 32 support points (two per wavefront), [2] more (one)
  const int which = WHICH;
  const int item = blockIdx.x * Cfg::NG + wl / G;
  if (item >= b.mle_cnt[3 * f + which]) return;
  const int lid = b.mle_list[((size_t)f * 3 + which) * c.line_cap + item];
  if constexpr (G == 64 && RW > 64) {
    // most lines of this list have at most 64 support points: one row per lane, none of the second slot's instructions
    const int seg = b.recs[(size_t)f * c.line_cap + lid].seg;
    const int ns = (int)b.cand_out[((size_t)f * c.cand_cap + seg) * LF_CAND_STRIDE + 26];
    if (ns <= 64) { f_mle_line<64, 64>(c, b, f, lid, lds_rows[0], g); return; }
  }
  f_mle_line<G, RW>(c, b, f, lid, lds_rows[wl / G], g);
}

// ------------------------------------------------------------------------------ getGradient + MSLD
__device__ __forceinline__ int f_cvround(double v) { return (int)__builtin_rint(v); }   // cvRound: half to even
// cv::clipLine (OpenCV 2.4 drawing.cpp)
__device__ bool f_clipline(int w, int h, long long *px1, long long *py1, long long *px2, long long *py2) {
  long long x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2, right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (a - y1) * (x2 - x1) / (y2 - y1); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
    if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (a - y2) * (x2 - x1) / (y2 - y1); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) { a = c1 == 1 ? 0 : right; y1 += (a - x1) * (y2 - y1) / (x2 - x1); x1 = a; c1 = 0; }
      if (c2) { a = c2 == 1 ? 0 : right; y2 += (a - x2) * (y2 - y1) / (x2 - x1); x2 = a; c2 = 0; }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
  }
  return (c1 | c2) == 0;
}

#define MSLD_SPP 7   // samples per pass: 7 x 9 sub-regions on 63 lanes
__device__ const double k_msld_gauss[9] = {0.24142, 0.30046, 0.35127, 0.38579, 0.39804, 0.38579, 0.35127, 0.30046, 0.24142};   // utils.cpp:1560
__global__ void __launch_bounds__(64) k_describe(FrontConsts c, FrontBuffers b) {
  __shared__ double G[MSLD_SPP * 36];
  const int li = blockIdx.x, f = blockIdx.y, lane = f_lane();
  int nl = b.nlines[f];
  if (nl > c.line_cap) nl = c.line_cap;
  if (li >= nl) return;
  lf_line_record *R = b.recs + (size_t)f * c.line_cap + li;
  const uint32_t *gxy = (const uint32_t *)(b.gxy + (size_t)f * c.W * c.H * 2);   // (gx | gy << 16) per pixel
  const int W = c.W, H = c.H;
  const double p0 = R->p[0], p1 = R->p[1], q0 = R->q[0], q1 = R->q[1];
  // ---- FrameLine::getGradient (lineslam.cpp:527-537): cv::LineIterator(img, p, q, 8).  Pixel i of
  // the Bresenham walk in closed form: minor-axis offset k_i = max(0, ceil((2 minor i - major)/(2 major))).
  double r0, r1;
  {
    long long x1 = f_cvround(p0), y1 = f_cvround(p1), x2 = f_cvround(q0), y2 = f_cvround(q1);
    bool ok = true;
    if ((u64)x1 >= (u64)W || (u64)x2 >= (u64)W || (u64)y1 >= (u64)H || (u64)y2 >= (u64)H) ok = f_clipline(W, H, &x1, &y1, &x2, &y2);
    long long sxs = 0, sys_ = 0;
    if (ok) {
      int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
      int sgx = dx < 0 ? -1 : 1, sgy = dy < 0 ? -1 : 1;
      int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
      bool steep = ady > adx;
      int major = steep ? ady : adx, minor = steep ? adx : ady;
      int count = major + 1;
      for (int i = lane; i < count; i += 64) {
        long long num = 2ll * minor * i - major;
        int k = (num > 0 && major > 0) ? (int)((num + 2ll * major - 1) / (2ll * major)) : 0;
        int x = (int)x1 + (steep ? sgx * k : sgx * i);
        int y = (int)y1 + (steep ? sgy * i : sgy * k);
        const uint32_t gv = gxy[y * W + x];
        sxs += (int16_t)(gv & 0xffffu);
        sys_ += (int16_t)(gv >> 16);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sxs += __shfl_xor(sxs, o, 64); sys_ += __shfl_xor(sys_, o, 64); }
    double xs = (double)sxs, ys = (double)sys_;
    double len = lf_sqrt(xs * xs + ys * ys);
    r0 = xs / len;
    r1 = ys / len;
  }
  // ---- computeMSLD (utils.cpp:1544-1610)
  const int s = (int)(5 * W / 800.0);
  const double sd = (double)s, step = c.P.msld_sample_interval;
  const double len = lf_sqrt((p0 - q0) * (p0 - q0) + (p1 - q1) * (p1 - q1));
  // number of sample indices i with i*step < len
  int ntot = 0;
  if (len > 0 && step > 0) {
    ntot = (int)__builtin_ceil(len / step);
    while ((double)ntot * step < len) ntot++;
    while (ntot > 0 && !((double)(ntot - 1) * step < len)) ntot--;
  }
  double sum = 0, sum2 = 0;   // lanes 0..35: running sums of component `lane`
  int nvalid = 0;
  // One lane per (sample, sub-region): MSLD_SPP = 7 samples x 9 sub-regions per pass (lane 63 idles).  A lane keeps the four
  // sums of ITS sub-region only (computeSubPSR, utils.cpp:1510-1542) and writes them straight into the LDS tile of the pass
  // -- no per-lane 36-entry column, no scratch.  A sample counts only if all nine of its sub-regions lie inside the image
  // (the reference breaks out of the sample otherwise): nine-bit groups of one ballot.  Accumulator lane c then adds
  // component c of the valid samples in sample order.
  const int sidx = lane / 9, jj = lane - 9 * sidx, j = jj - 4;
  const double gw = (lane < 36) ? k_msld_gauss[lane >> 2] : 0.0;
  for (int i0 = 0; i0 < ntot; i0 += MSLD_SPP) {
    const int i = i0 + sidx;
    bool ok = lane < 9 * MSLD_SPP && i < ntot;
    double tl_x = 0, tl_y = 0;
    if (ok) {
      double t = (i * step / len);
      double ptx = p0 + (q0 - p0) * t, pty = p1 + (q1 - p1) * t;
      double px = ptx + r0 * (j * s), py = pty + r1 * (j * s);
      tl_x = __builtin_floor(px - sd / 2); tl_y = __builtin_floor(py - sd / 2);
      if (tl_x < 0 || tl_y < 0 || tl_x + sd + 1 > W || tl_y + sd + 1 > H || !(tl_x == tl_x) || !(tl_y == tl_y)) ok = false;
    }
    const u64 mv = __ballot(ok);
    unsigned vs = 0;                                       // valid samples of this pass (uniform)
#pragma unroll
    for (int k = 0; k < MSLD_SPP; k++) vs |= (unsigned)(((mv >> (9 * k)) & 0x1ffull) == 0x1ffull) << k;
    const int cnt = __popc(vs);
    if (cnt == 0) continue;
    if (lane < 9 * MSLD_SPP && ((vs >> sidx) & 1u)) {
      const int slot = __popc(vs & ((1u << sidx) - 1u));
      double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      for (int y = (int)tl_y; y < tl_y + sd; ++y)
        for (int x = (int)tl_x; x < tl_x + sd; ++x) {
          const uint32_t gv = gxy[y * W + x];
          double xg = (double)(int16_t)(gv & 0xffffu), yg = (double)(int16_t)(gv >> 16);
          double tmp1 = xg * r0 + yg * r1;
          double tmp2 = xg * (-r1) + yg * r0;
          if (tmp1 >= 0) v1 = v1 + tmp1; else v2 = v2 - tmp1;
          if (tmp2 >= 0) v3 = v3 + tmp2; else v4 = v4 - tmp2;
        }
      double *g = &G[slot * 36 + jj * 4];
      g[0] = v1; g[1] = v2; g[2] = v3; g[3] = v4;
    }
    f_sync();
    if (lane < 36) {
      for (int q = 0; q < cnt; q++) {
        double v = G[q * 36 + lane] * gw;
        sum += v;
        sum2 += v * v;
      }
    }
    f_sync();
    nvalid += cnt;
  }
  double *des = R->des;
  if (nvalid == 0) {
    uint64_t stream = LF_STREAM_LINE3D(b.frame_ids[f], R->seg);
    for (int i = lane; i < 72; i += 64) des[i] = (double)lf_rand31(c.P.rng_seed, stream, 1000 + (uint64_t)i);
    if (lane == 0) { R->r[0] = r0; R->r[1] = r1; }
    return;
  }
  double mean = sum / nvalid;
  double sdev = lf_sqrt(sum2 / nvalid - mean * mean);
  // normalisations: sequential sums of squares over the 36 / 36 / 72 entries
  double nm = 0, ns = 0;
  for (int i = 0; i < 36; i++) { double v = f_rl64(mean, i); nm += v * v; }
  for (int i = 0; i < 36; i++) { double v = f_rl64(sdev, i); ns += v * v; }
  nm = 1.0 / lf_sqrt(nm);
  ns = 1.0 / lf_sqrt(ns);
  double dm = mean * nm, dsd = sdev * ns;
  if (dm > 0.4) dm = 0.4;
  if (dsd > 0.4) dsd = 0.4;
  double nt = 0;
  for (int i = 0; i < 36; i++) { double v = f_rl64(dm, i); nt += v * v; }
  for (int i = 0; i < 36; i++) { double v = f_rl64(dsd, i); nt += v * v; }
  nt = 1.0 / lf_sqrt(nt);
  if (lane < 36) { des[lane] = dm * nt; des[lane + 36] = dsd * nt; }
  if (lane == 0) { R->r[0] = r0; R->r[1] = r1; }
}

// ----------------------------------------------------------------------------------------------
static void launch_mle_kernels(const FrontConsts &c, const FrontBuffers &b, int B, hipStream_t st) {
#if LF_MLE_PAIR64
  hipLaunchKernelGGL((k_mle<32, 64, 0>), dim3((c.line_cap + 1) / 2, B), dim3(64), 0, st, c, b);   // 33..64 points: 2 lines per wavefront
#else
  if (LF_MLE_SMALL > 0) hipLaunchKernelGGL((k_mle<16, 16, 0>), dim3((c.line_cap + 3) / 4, B), dim3(64), 0, st, c, b);   // 4 lines per wavefront
#endif
  hipLaunchKernelGGL((k_mle<32, 32, 1>), dim3((c.line_cap + 1) / 2, B), dim3(64), 0, st, c, b);   // <= 32 points: 2 lines per wavefront
  hipLaunchKernelGGL((k_mle<64, MLE_N, 2>), dim3(c.line_cap, B), dim3(64), 0, st, c, b);
}
void lf_front_launch_mle(const FrontConsts &c, const FrontBuffers &b, int B, hipStream_t st) { launch_mle_kernels(c, b, B, st); }
void lf_front_launch(const FrontConsts &c, const FrontBuffers &b, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_sobel5, dim3((c.W + 255) / 256, (c.H + SOBEL_ROWS - 1) / SOBEL_ROWS, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_line3d, dim3(c.cand_cap, B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_records, dim3(B), dim3(64), 0, st, c, b);
#ifndef LF_EXP_SKIP_MLE   // (throughput experiments only: what the step costs without this stage)
  launch_mle_kernels(c, b, B, st);
#endif
  hipLaunchKernelGGL(k_describe, dim3(c.line_cap, B), dim3(64), 0, st, c, b);
}
