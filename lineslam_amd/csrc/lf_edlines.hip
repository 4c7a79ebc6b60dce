// lf_edlines.hip -- EDLines for gfx950, batched over frames.  The reference has this detector as a
// binary only (external/EDLines/libEDLines.a); what is implemented is the algorithm of its object code as restated in
// oracle/edlines_oracle.c (function by function from the disassembly; all 166 rows of the reference's example output
// reproduced at its 0.01 px resolution).  The kernels are held bit for bit against that file.
//
//   k_ed_smooth    cvSmooth(CV_GAUSSIAN, 5, 5): taps 1 4 6 4 1 / 16 in 8-bit fixed point, BORDER_REPLICATE, the column pass
//                  rounded to nearest even except in the last W % 4 columns (OpenCV 2.4's SSE2 / scalar split); LDS tile
//   k_ed_gradient  ComputeGradientMapByLSD: 2x2 differences, |gx| + |gy|, threshold 11, direction map
//   k_ed_anchor    anchors (both neighbours across the edge lower by >= 3) marked in the edge map + histogram by gradient
//   k_ed_sort      SortAnchorsByGradValue, one wavefront per frame: counting sort -- scan of the 1021 bins, then a raster
//                  pass that ranks the anchors of every 64-pixel batch inside their bin (strongest first, raster order in
//                  one gradient value)
//   k_ed_link      ONE LANE PER FRAME, frames in flight are the parallel axis: the anchor walk with its explicit stack and
//                  chain tree, the longest path as the edge segment, the leftover branches -- a dependent chain per frame by
//                  nature (every step reads what the previous one marked)
//   k_ed_split_count / k_ed_scan_segments / k_ed_split_join   SplitSegment2Lines + JoinCollinearLines with ONE LANE PER EDGE
//                  SEGMENT (segments are independent; the line list is their concatenation): count, ordered offsets, then the
//                  lines at their slots and the joining inside the segment; fp64 sums in the oracle's order
//   k_ed_validate  ValidateLineSegments with ONE LANE PER LINE (the rectangle pixels are counted as they are enumerated)
//   k_ed_emit      the valid lines in list order -> segment rows (one wavefront per frame)
#include "lf_edlines.h"
#include "lf_math.h"
#include <float.h>

#define ED_GRAD_THRESH 11
#define ED_ANCHOR_THRESH 3
#define ED_VERTICAL 1
#define ED_HORIZONTAL 2
#define ED_ANCHOR 254
#define ED_EDGE 255
#define ED_MIN_PATH 10
#define ED_LINE_ERROR 1.0
#define ED_MAX_DIST 6.0
#define ED_MAX_ERROR 1.3
#define ED_PI 3.14159265358979323846
enum { ED_LEFT = 1, ED_RIGHT = 2, ED_UP = 3, ED_DOWN = 4 };

__device__ __forceinline__ int e_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int e_iabs(int v) { return v < 0 ? -v : v; }

#define ES_TW 64
#define ES_TH 16
__global__ void __launch_bounds__(256) k_ed_smooth(EdConsts c, EdBuffers b) {
  __shared__ uint8_t s_in[ES_TH + 4][ES_TW + 4];
  __shared__ int s_row[ES_TH + 4][ES_TW];
  const int sk[5] = {16, 64, 96, 64, 16};
  const int f = blockIdx.z, W = c.W, H = c.H, tid = threadIdx.x;
  const uint8_t *src = b.gray + (size_t)f * b.gray_frame_stride;
  uint8_t *dst = b.smooth + (size_t)f * W * H;
  const int x0 = blockIdx.x * ES_TW, y0 = blockIdx.y * ES_TH;
  for (int i = tid; i < (ES_TH + 4) * (ES_TW + 4); i += 256) {
    const int ty = i / (ES_TW + 4), tx = i - ty * (ES_TW + 4);
    s_in[ty][tx] = src[(size_t)e_clampi(y0 + ty - 2, 0, H - 1) * b.gray_row_stride + e_clampi(x0 + tx - 2, 0, W - 1)];
  }
  __syncthreads();
  for (int i = tid; i < (ES_TH + 4) * ES_TW; i += 256) {
    const int ty = i / ES_TW, tx = i - ty * ES_TW;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) s += sk[k] * s_in[ty][tx + k];
    s_row[ty][tx] = s;
  }
  __syncthreads();
  for (int i = tid; i < ES_TH * ES_TW; i += 256) {
    const int ty = i / ES_TW, tx = i - ty * ES_TW, x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
      int s = 0, v;
#pragma unroll
      for (int k = 0; k < 5; k++) s += sk[k] * s_row[ty + k][tx];
      if (x < (W & ~3)) { const int n = s >> 16, r = s & 65535; v = r > 32768 ? n + 1 : (r < 32768 ? n : n + (n & 1)); }
      else v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * W + x] = (uint8_t)(v > 255 ? 255 : v);
    }
  }
}

__global__ void __launch_bounds__(256) k_ed_gradient(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, W = c.W, H = c.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const uint8_t *s = b.smooth + (size_t)f * W * H;
  int g = ED_GRAD_THRESH - 1, d = 0;
  if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
    const uint8_t *p = s + i;
    const int com1 = (int)p[W + 1] - p[0], com2 = (int)p[1] - p[W];
    const int gx = e_iabs(com1 + com2), gy = e_iabs(com1 - com2);
    g = gx + gy;
    if (g >= ED_GRAD_THRESH) d = gx >= gy ? ED_VERTICAL : ED_HORIZONTAL;
  }
  b.G[(size_t)f * W * H + i] = (int16_t)g;
  b.D[(size_t)f * W * H + i] = (uint8_t)d;
}

__global__ void __launch_bounds__(256) k_ed_anchor(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, W = c.W, H = c.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const int16_t *G = b.G + (size_t)f * W * H;
  uint8_t e = 0;
  if (y >= 2 && y < H - 2 && x >= 2 && x < W - 2) {
    const int g = G[i];
    if (g >= ED_GRAD_THRESH) {
      bool a;
      if (b.D[(size_t)f * W * H + i] == ED_VERTICAL) a = g - G[i + 1] >= ED_ANCHOR_THRESH && g - G[i - 1] >= ED_ANCHOR_THRESH;
      else a = g - G[i + W] >= ED_ANCHOR_THRESH && g - G[i - W] >= ED_ANCHOR_THRESH;
      if (a) { e = ED_ANCHOR; atomicAdd(&b.hist[(size_t)f * LF_ED_BINS + g], 1); }
    }
  }
  b.E[(size_t)f * W * H + i] = e;
}

// SortAnchorsByGradValue as the walk consumes it (from the last entry of the binary's ascending array down): strongest gradient
// first, raster order inside one gradient value.  base[g] = number of anchors with a larger gradient.
__global__ void __launch_bounds__(64) k_ed_sort(EdConsts c, EdBuffers b) {
  __shared__ int base[LF_ED_BINS];
  const int f = blockIdx.x, lane = threadIdx.x, W = c.W, H = c.H;
  const int *hist = b.hist + (size_t)f * LF_ED_BINS;
  {   // base[g] = anchors with a larger gradient: exclusive suffix sums of the histogram, 64 bins per trip from the top
    int run = 0;
    for (int g0 = LF_ED_BINS - 64; g0 >= 0; g0 -= 64) {
      const int g = g0 + 63 - lane;                 // lane 0 takes the largest value of the group
      const int v = hist[g];
      int inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
      base[g] = run + inc - v;
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) b.nanch[f] = run;
  }
  __syncthreads();
  const uint8_t *E = b.E + (size_t)f * W * H;
  const int16_t *G = b.G + (size_t)f * W * H;
  unsigned *out = b.anchors + (size_t)f * c.anchor_cap;
  // the map reads of the NEXT batch are issued before the current one is ranked (the ranking is one dependent chain of LDS
  // counters; the reads need not wait for it)
  int en = lane < W * H ? (int)E[lane] : 0, gn = lane < W * H ? (int)G[lane] : -1;
  for (int i0 = 0; i0 < W * H; i0 += 64) {
    const int i = i0 + lane;
    const int ec = en, gc = gn;
    { const int in = i + 64; en = in < W * H ? (int)E[in] : 0; gn = in < W * H ? (int)G[in] : -1; }
    const bool a = ec == ED_ANCHOR;
    const int g = a ? gc : -1;
    unsigned long long todo = __ballot(a);
    while (todo) {                                   // one gradient value of the batch at a time, its lanes ranked in lane order
      const int src = __builtin_ctzll(todo);
      const int gsel = __shfl(g, src, 64);
      const unsigned long long m = __ballot(g == gsel);
      if (g == gsel) {
        const int pos = base[gsel] + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < c.anchor_cap) out[pos] = (unsigned)i;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (lane == src) base[gsel] += __popcll(m);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      todo &= ~m;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- linking + lines
struct EdFrame {
  const int16_t *G; const uint8_t *D; uint8_t *E; const uint8_t *img; int img_stride;
  int W, H;
  unsigned *walk, *stack, *segpix; EdChainRec *ch; int *chain_nos, *segtab, *rect;
  EdLine *lines;
  const int *kmin; const double *atan_lut;
  int lut_size, nmax;
};
__device__ __forceinline__ int e_pr(unsigned p) { return (int)(p >> 16); }
__device__ __forceinline__ int e_pc(unsigned p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ unsigned e_mk(int r, int c) { return ((unsigned)r << 16) | (unsigned)c; }

// LongestChain (the binary recurses; a deep tree -- an edge that changes between horizontal and vertical every few pixels --
// would overflow a device stack): the same post-order evaluation through the parent links, the first child's result kept in
// the chain record.  Only the chains reachable from `root` are evaluated and pruned, as in the recursion.
__device__ int e_longest_chain(EdChainRec *ch, int root) {
  if (root == -1 || ch[root].len == 0) return 0;
  int cur = root, st = 0, ret = 0;
  for (;;) {
    if (st == 0) {
      const int c0 = ch[cur].child0;
      if (c0 != -1 && ch[c0].len != 0) { cur = c0; continue; }
      ret = 0; st = 1;
    }
    if (st == 1) {
      st = 2;
      ch[cur].tmp = ret;                          // the length below the first child
      const int c1 = ch[cur].child1;
      if (c1 != -1 && ch[c1].len != 0) { cur = c1; st = 0; continue; }
      ret = 0;
    }
    {
      const int len0 = ch[cur].tmp, len1 = ret;
      int mx;
      if (len0 >= len1) { mx = len0; ch[cur].child1 = -1; }
      else { mx = len1; ch[cur].child0 = -1; }
      ret = ch[cur].len + mx;
    }
    if (cur == root) return ret;
    const int p = ch[cur].parent;
    st = (ch[p].child0 == cur) ? 1 : 2;     // back in the parent: after its first or its second child
    cur = p;
  }
}
__device__ int e_retrieve_chain_nos(const EdChainRec *ch, int root, int *nos, int cap) {
  int count = 0;
  while (root != -1) {
    if (count >= cap) return -1;
    nos[count++] = root;
    if (ch[root].child0 != -1) root = ch[root].child0;
    else root = ch[root].child1;
  }
  return count;
}

// pixel q of segment storage as doubles (x = column, y = row)
__device__ __forceinline__ double e_sx(const unsigned *p, int q) { return (double)e_pc(p[q]); }
__device__ __forceinline__ double e_sy(const unsigned *p, int q) { return (double)e_pr(p[q]); }

__device__ void e_line_fit_err(const unsigned *p, int count, double *pa, double *pb, double *pe, int *pinvert) {
  const double S = count;
  double Sx = 0, Sy = 0, Sxx = 0, Sxy = 0, dx = 0, dy = 0;
  if (count < 2) return;
  for (int i = 0; i < count; i++) { Sx += e_sx(p, i); Sy += e_sy(p, i); }
  const double mx = Sx / count, my = Sy / count;
  for (int i = 0; i < count; i++) { dx += (e_sx(p, i) - mx) * (e_sx(p, i) - mx); dy += (e_sy(p, i) - my) * (e_sy(p, i) - my); }
  const int inv = dx < dy;
  if (inv) { const double d = Sx; Sx = Sy; Sy = d; }
  *pinvert = inv;
  for (int i = 0; i < count; i++) { const double x = inv ? e_sy(p, i) : e_sx(p, i), y = inv ? e_sx(p, i) : e_sy(p, i); Sxx += x * x; Sxy += x * y; }
  const double D = S * Sxx - Sx * Sx;
  const double a = (Sxx * Sy - Sx * Sxy) / D, bb = (S * Sxy - Sx * Sy) / D;
  *pa = a; *pb = bb;
  if (bb == 0.0) {
    double error = 0;
    for (int i = 0; i < count; i++) { const double y = inv ? e_sx(p, i) : e_sy(p, i); error += lf_fabs(a - y); }
    *pe = error / count;
  } else {
    double error = 0;
    for (int i = 0; i < count; i++) {
      const double x = inv ? e_sy(p, i) : e_sx(p, i), y = inv ? e_sx(p, i) : e_sy(p, i);
      const double d = -1.0 / bb, cc = y - d * x;
      const double x2 = (a - cc) / (d - bb), y2 = a + bb * x2;
      error += (x - x2) * (x - x2) + (y - y2) * (y - y2);
    }
    *pe = lf_sqrt(error / count);
  }
}
__device__ void e_line_fit(const unsigned *p, int count, double *pa, double *pb, int invert) {
  const double S = count;
  double Sx = 0, Sy = 0, Sxx = 0, Sxy = 0;
  if (count < 2) return;
  for (int i = 0; i < count; i++) { Sx += e_sx(p, i); Sy += e_sy(p, i); }
  if (invert) { const double d = Sx; Sx = Sy; Sy = d; }
  for (int i = 0; i < count; i++) { const double x = invert ? e_sy(p, i) : e_sx(p, i), y = invert ? e_sx(p, i) : e_sy(p, i); Sxx += x * x; Sxy += x * y; }
  const double D = S * Sxx - Sx * Sx;
  *pa = (Sxx * Sy - Sx * Sxy) / D;
  *pb = (S * Sxy - Sx * Sy) / D;
}
__device__ void e_closest_point(double x1, double y1, double a, double b, int invert, double *xo, double *yo) {
  double x2, y2;
  if (invert == 0) {
    if (b == 0) { x2 = x1; y2 = a; }
    else { const double d = -1.0 / b, c = y1 - d * x1; x2 = (a - c) / (d - b); y2 = a + b * x2; }
  } else {
    if (b == 0) { x2 = a; y2 = y1; }
    else { const double d = -1.0 / b, c = x1 - d * y1; y2 = (a - c) / (d - b); x2 = a + b * y2; }
  }
  *xo = x2; *yo = y2;
}
__device__ double e_min_distance(double x1, double y1, double a, double b, int invert) {
  double x2, y2;
  e_closest_point(x1, y1, a, b, invert, &x2, &y2);
  return lf_sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
}
__device__ void e_update_line_parameters(EdLine *ls) {
  const double dx = ls->ex - ls->sx, dy = ls->ey - ls->sy;
  if (lf_fabs(dx) >= lf_fabs(dy)) {
    ls->invert = 0;
    if (lf_fabs(dy) < 1e-3) { ls->b = 0; ls->a = (ls->sy + ls->ey) / 2; }
    else { ls->b = dy / dx; ls->a = ls->sy - ls->b * ls->sx; }
  } else {
    ls->invert = 1;
    if (lf_fabs(dx) < 1e-3) { ls->b = 0; ls->a = (ls->sx + ls->ex) / 2; }
    else { ls->b = dx / dy; ls->a = ls->sx - ls->b * ls->sy; }
  }
}
__device__ double e_min_distance_between_two_lines(const EdLine *ls1, const EdLine *ls2) {
  double dx = ls1->sx - ls2->sx, dy = ls1->sy - ls2->sy, d = lf_sqrt(dx * dx + dy * dy), mn = d;
  dx = ls1->sx - ls2->ex; dy = ls1->sy - ls2->ey; d = lf_sqrt(dx * dx + dy * dy); if (d < mn) mn = d;
  dx = ls1->ex - ls2->sx; dy = ls1->ey - ls2->sy; d = lf_sqrt(dx * dx + dy * dy); if (d < mn) mn = d;
  dx = ls1->ex - ls2->ex; dy = ls1->ey - ls2->ey; d = lf_sqrt(dx * dx + dy * dy); if (d < mn) mn = d;
  return mn;
}
__device__ int e_try_to_join(EdLine *ls1, EdLine *ls2, double max_dist, double max_err) {
  double dist = e_min_distance_between_two_lines(ls1, ls2), dx, dy, d, mx;
  const EdLine *shorter = ls1, *longer = ls2;
  int which;
  if (dist > max_dist) return 0;
  dx = ls1->sx - ls1->ex; dy = ls1->sy - ls1->ey; const double prevLen = lf_sqrt(dx * dx + dy * dy);
  dx = ls2->sx - ls2->ex; dy = ls2->sy - ls2->ey; const double nextLen = lf_sqrt(dx * dx + dy * dy);
  if (prevLen > nextLen) { shorter = ls2; longer = ls1; }
  dist = e_min_distance(shorter->sx, shorter->sy, longer->a, longer->b, longer->invert);
  dist += e_min_distance((shorter->sx + shorter->ex) / 2.0, (shorter->sy + shorter->ey) / 2.0, longer->a, longer->b, longer->invert);
  dist += e_min_distance(shorter->ex, shorter->ey, longer->a, longer->b, longer->invert);
  dist /= 3.0;
  if (dist > max_err) return 0;
  dx = lf_fabs(ls1->sx - ls2->sx); dy = lf_fabs(ls1->sy - ls2->sy); d = dx + dy; mx = d; which = 1;
  dx = lf_fabs(ls1->sx - ls2->ex); dy = lf_fabs(ls1->sy - ls2->ey); d = dx + dy; if (d > mx) { mx = d; which = 2; }
  dx = lf_fabs(ls1->ex - ls2->sx); dy = lf_fabs(ls1->ey - ls2->sy); d = dx + dy; if (d > mx) { mx = d; which = 3; }
  dx = lf_fabs(ls1->ex - ls2->ex); dy = lf_fabs(ls1->ey - ls2->ey); d = dx + dy; if (d > mx) { mx = d; which = 4; }
  if (which == 1) { ls1->ex = ls2->sx; ls1->ey = ls2->sy; }
  else if (which == 2) { ls1->ex = ls2->ex; ls1->ey = ls2->ey; }
  else if (which == 3) { ls1->sx = ls2->sx; ls1->sy = ls2->sy; }
  else { ls1->sx = ls1->ex; ls1->sy = ls1->ey; ls1->ex = ls2->ex; ls1->ey = ls2->ey; }
  if (ls1->firstPixelIndex + ls1->len + 5 >= ls2->firstPixelIndex) ls1->len += ls2->len;
  else if (ls2->len > ls1->len) { ls1->firstPixelIndex = ls2->firstPixelIndex; ls1->len = ls2->len; }
  e_update_line_parameters(ls1);
  return 1;
}

// ---- validation
__device__ __forceinline__ int e_check_nfa(const EdFrame &F, int n, int k) { return n < F.nmax ? k >= F.kmin[n] : 0; }
__device__ double e_line_angle(const EdLine *ls) {
  double lineAngle;
  if (ls->invert == 0) lineAngle = lf_atan2(ls->b, 1.0);
  else lineAngle = lf_atan2(1.0 / ls->b, 1.0);
  if (lineAngle < 0) lineAngle += ED_PI;
  return lineAngle;
}
__device__ double e_my_atan2(const EdFrame &F, double yy, double xx) {
  double ay = lf_fabs(yy), ax = lf_fabs(xx);
  int invert = 0;
  if (ax < 1e-10) return (ay < 1e-10) ? 0.0 : ED_PI / 2.0;
  if (ay > ax) { const double t = ay; ay = ax; ax = t; invert = 1; }
  const double angle = F.atan_lut[(int)(ay / ax * 1024.0)];
  if ((xx >= 0 && yy >= 0) || (xx < 0 && yy < 0)) return invert ? ED_PI / 2.0 - angle : angle;
  return invert ? angle + ED_PI / 2.0 : ED_PI - angle;
}
__device__ int e_aligned(const EdFrame &F, int r, int c, double lineAngle) {
  const uint8_t *p = F.img + (size_t)r * F.img_stride + c;
  const int w = F.img_stride;
  const double prec = ED_PI / 8.0;
  const int com1 = (int)p[w + 1] - p[-w - 1], com2 = (int)p[-w + 1] - p[w - 1];
  const int gx = com1 + com2 + p[1] - p[-1], gy = com1 - com2 + p[w] - p[-w];
  const double pixelAngle = e_my_atan2(F, (double)gx, (double)-gy), diff = lf_fabs(lineAngle - pixelAngle);
  return diff <= prec || diff >= ED_PI - prec;
}
struct EdRectCount;
__device__ __forceinline__ void e_rect_point(EdRectCount *rc, int x, int y);
__device__ int e_enumerate_rect_points(double sx, double sy, double ex, double ey, EdRectCount *rc, int cap) {
  double vxTmp[4], vyTmp[4], vx[4], vy[4];
  const double x1 = sx, y1 = sy, x2 = ex, y2 = ey, width = 2;
  double dx = x2 - x1, dy = y2 - y1, ys, ye;
  const double vLen = lf_sqrt(dx * dx + dy * dy);
  int offset, x, y, noPoints = 0;
  int maxPoints = 4 * (int)(lf_fabs(sx - ex) + lf_fabs(sy - ey));
  if (maxPoints > cap) maxPoints = cap;
  dx = dx / vLen; dy = dy / vLen;
  vxTmp[0] = x1 - dy * width / 2.0; vyTmp[0] = y1 + dx * width / 2.0;
  vxTmp[1] = x2 - dy * width / 2.0; vyTmp[1] = y2 + dx * width / 2.0;
  vxTmp[2] = x2 + dy * width / 2.0; vyTmp[2] = y2 - dx * width / 2.0;
  vxTmp[3] = x1 + dy * width / 2.0; vyTmp[3] = y1 - dx * width / 2.0;
  if (x1 < x2 && y1 <= y2) offset = 0;
  else if (x1 >= x2 && y1 < y2) offset = 1;
  else if (x1 > x2 && y1 >= y2) offset = 2;
  else offset = 3;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    double sxv = vxTmp[0], syv = vyTmp[0];
#pragma unroll
    for (int q = 1; q < 4; q++) if (((offset + n) & 3) == q) { sxv = vxTmp[q]; syv = vyTmp[q]; }
    vx[n] = sxv; vy[n] = syv;
  }
  x = (int)__builtin_ceil(vx[0]) - 1;
  y = (int)__builtin_ceil(vy[0]);
  ys = ye = -DBL_MAX;
  while (noPoints < maxPoints) {
    y++;
    while (y > ye && x <= vx[2]) {
      x++;
      if (x > vx[2]) break;
      if ((double)x < vx[3]) {
        if (lf_fabs(vx[0] - vx[3]) <= 0.01) {
          if (vy[0] < vy[3]) ys = vy[0];
          else if (vy[0] > vy[3]) ys = vy[3];
          else ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
        } else ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
      } else {
        if (lf_fabs(vx[3] - vx[2]) <= 0.01) {
          if (vy[3] < vy[2]) ys = vy[3];
          else if (vy[3] > vy[2]) ys = vy[2];
          else ys = vy[3] + (x - vx[3]) * (vy[2] - vy[3]) / (vx[2] - vx[3]);
        } else ys = vy[3] + (x - vx[3]) * (vy[2] - vy[3]) / (vx[2] - vx[3]);
      }
      if ((double)x < vx[1]) {
        if (lf_fabs(vx[0] - vx[1]) <= 0.01) {
          if (vy[0] < vy[1]) ye = vy[1];
          else if (vy[0] > vy[1]) ye = vy[0];
          else ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
        } else ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
      } else {
        if (lf_fabs(vx[1] - vx[2]) <= 0.01) {
          if (vy[1] < vy[2]) ye = vy[2];
          else if (vy[1] > vy[2]) ye = vy[1];
          else ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
        } else ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
      }
      y = (int)__builtin_ceil(ys);
    }
    if (x > vx[2]) break;
    e_rect_point(rc, x, y); noPoints++;
  }
  return noPoints;
}
// ValidateLineSegments' rectangle test: the pixels of EnumerateRectPoints are consumed as they are produced (the binary fills
// two arrays first and counts afterwards: same pixels, same order, no per-line scratch -- one lane validates one line)
struct EdRectCount { const EdFrame *F; double lineAngle; int count, aligned; };
__device__ int e_validate_rect(const EdFrame &F, const EdLine *ls) {
  EdRectCount rc;
  rc.F = &F; rc.lineAngle = e_line_angle(ls); rc.count = 0; rc.aligned = 0;
  e_enumerate_rect_points(ls->sx, ls->sy, ls->ex, ls->ey, &rc, (F.W + F.H) * 4);
  return e_check_nfa(F, rc.count, rc.aligned);
}

__device__ __forceinline__ void e_rect_point(EdRectCount *rc, int x, int y) {
  const EdFrame &F = *rc->F;
  if (y <= 0 || y >= F.H - 1 || x <= 0 || x >= F.W - 1) return;
  rc->count++;
  if (e_aligned(F, y, x, rc->lineAngle)) rc->aligned++;
}

// SplitSegment2Lines on the pixels p[0 .. noPixels) of segment segmentNo; line k of the segment goes to out[k] (out == nullptr:
// count only).  Returns the number of lines of the segment.
__device__ int e_split_segment(const unsigned *p, int noPixels, int segmentNo, int min_line_len, EdLine *out) {
  int nlines = 0;
  int firstPixelIndex = 0;
  while (noPixels >= min_line_len) {
    int valid = 0, lastInvert = 0, index, len;
    double lastA = 0, lastB = 0, error = 0;
    while (noPixels >= min_line_len) {
      e_line_fit_err(p, min_line_len, &lastA, &lastB, &error, &lastInvert);
      if (error <= 0.5) { valid = 1; break; }
      noPixels -= 1; p += 1; firstPixelIndex += 1;
    }
    if (!valid) return nlines;
    index = min_line_len;
    len = min_line_len;
    while (index < noPixels) {
      const int startIndex = index;
      int lastGoodIndex = index - 1, goodPixelCount = 0, badPixelCount = 0;
      while (index < noPixels) {
        const double d = e_min_distance(e_sx(p, index), e_sy(p, index), lastA, lastB, lastInvert);
        if (d <= ED_LINE_ERROR) { lastGoodIndex = index; goodPixelCount++; badPixelCount = 0; }
        else { badPixelCount++; if (badPixelCount >= 5) break; }
        index++;
      }
      if (goodPixelCount >= 2) {
        len += lastGoodIndex - startIndex + 1;
        e_line_fit(p, len, &lastA, &lastB, lastInvert);
        index = lastGoodIndex + 1;
      }
      if (goodPixelCount < 2 || index >= noPixels) {
        EdLine l;
        int i0 = 0, i1;
        while (e_min_distance(e_sx(p, i0), e_sy(p, i0), lastA, lastB, lastInvert) > ED_LINE_ERROR) i0++;
        e_closest_point(e_sx(p, i0), e_sy(p, i0), lastA, lastB, lastInvert, &l.sx, &l.sy);
        const int noSkippedPixels = i0;
        i1 = lastGoodIndex;
        while (e_min_distance(e_sx(p, i1), e_sy(p, i1), lastA, lastB, lastInvert) > ED_LINE_ERROR) i1--;
        e_closest_point(e_sx(p, i1), e_sy(p, i1), lastA, lastB, lastInvert, &l.ex, &l.ey);
        l.a = lastA; l.b = lastB; l.invert = lastInvert; l.pad_ = 0; l.pad2_ = 0; l.segmentNo = segmentNo;
        l.firstPixelIndex = firstPixelIndex + noSkippedPixels; l.len = i1 - noSkippedPixels + 1;
        if (out) out[nlines] = l;
        nlines++;
        len = i1 + 1;
        break;
      }
    }
    noPixels -= len; p += len; firstPixelIndex += len;
  }
  return nlines;
}

// append chain cn (forwards from startIndex, or backwards) to the segment under construction, with the clean-up of the binary
__device__ __forceinline__ bool e_near(unsigned a, unsigned b) { return e_iabs(e_pr(a) - e_pr(b)) <= 1 && e_iabs(e_pc(a) - e_pc(b)) <= 1; }

__global__ void __launch_bounds__(64) k_ed_link(EdConsts c, EdBuffers b) {
  // A dependent chain per frame; frames in flight fill the chip.  ALL 64 lanes run the walk with identical (uniform) state --
  // same loads, same decisions, the same value stored to the same address by every lane (one write per instruction) -- so that
  // the inner loop can fetch a 7 x 9 window of E, G and D ahead of the walker with ONE load per map (a cell per lane; the
  // decisions read them with readlane) and walk up to eight pixels on it, instead of four dependent round trips per pixel.
  const int f = blockIdx.x, lane = (int)(threadIdx.x & 63u);
  const int W = c.W, H = c.H;
  const size_t NP = (size_t)W * H;
  EdFrame F;
  F.G = b.G + f * NP; F.D = b.D + f * NP; F.E = b.E + f * NP; F.W = W; F.H = H;
  F.img = b.gray + (size_t)f * b.gray_frame_stride; F.img_stride = b.gray_row_stride;
  F.walk = b.walk + f * NP; F.stack = b.stack + (size_t)f * LF_ED_STACK_CAP * 2; F.segpix = b.segpix + f * NP;
  F.ch = b.chains + (size_t)f * (LF_ED_CHAIN_CAP + 1); F.chain_nos = b.chain_nos + (size_t)f * (W + H) * 8;
  F.segtab = b.segtab + (size_t)f * c.segtab_cap * 2; F.rect = nullptr;
  F.lines = b.lines + (size_t)f * LF_ED_LINE_CAP;
  F.kmin = b.kmin; F.atan_lut = b.atan_lut; F.lut_size = c.lut_size; F.nmax = c.nmax;
  const int16_t *G = F.G; const uint8_t *D = F.D; uint8_t *E = F.E;
  unsigned *pixels = F.walk, *segpix = F.segpix;
  EdChainRec *chains = F.ch;
  int *chainNos = F.chain_nos;
  const unsigned *A = b.anchors + (size_t)f * c.anchor_cap;
  const int nos_cap = (W + H) * 8;
  int noAnchors = b.nanch[f];
  bool overflow = noAnchors > c.anchor_cap;
  if (overflow) noAnchors = c.anchor_cap;
  int nsegments = 0, nsegpix = 0;
  // ---- join the anchors, the one with the greatest gradient first
  // (the anchors are taken 64 at a time: one parallel look at the edge map drops those a walk has already consumed -- most of
  // them -- without a dependent load each; an anchor that is still alive then is looked at again when its turn comes)
  for (int k0 = 0; k0 < noAnchors && !overflow; k0 += 64) {
   const int myidx = (k0 + lane < noAnchors) ? (int)A[k0 + lane] : -1;
   unsigned long long am = __ballot(myidx >= 0 && E[myidx < 0 ? 0 : myidx] == ED_ANCHOR);
   while (am != 0ull && !overflow) {
    const int aL = __builtin_ctzll(am);
    am &= am - 1ull;
    const int idx = __builtin_amdgcn_readlane(myidx, aL), i = idx / W, j = idx - i * W;
    int noChains = 1, len = 0, duplicatePixelCount = 0, top = -1;
    if (E[idx] != ED_ANCHOR) continue;
    chains[0].len = 0; chains[0].parent = -1; chains[0].dir = 0; chains[0].child0 = chains[0].child1 = -1; chains[0].pix = 0;
    if (D[idx] == ED_VERTICAL) {
      ++top; F.stack[2 * top] = e_mk(i, j); F.stack[2 * top + 1] = (0u << 3) | ED_DOWN;
      ++top; F.stack[2 * top] = e_mk(i, j); F.stack[2 * top + 1] = (0u << 3) | ED_UP;
    } else {
      ++top; F.stack[2 * top] = e_mk(i, j); F.stack[2 * top + 1] = (0u << 3) | ED_RIGHT;
      ++top; F.stack[2 * top] = e_mk(i, j); F.stack[2 * top + 1] = (0u << 3) | ED_LEFT;
    }
    while (top >= 0) {
      int r = e_pr(F.stack[2 * top]), cc = e_pc(F.stack[2 * top]), chainLen = 0;
      const int dir = (int)(F.stack[2 * top + 1] & 7u), parent = (int)(F.stack[2 * top + 1] >> 3);
      const bool horizontal = (dir == ED_LEFT || dir == ED_RIGHT);
      const int step = (dir == ED_LEFT || dir == ED_UP) ? -1 : 1;
      const bool child0 = (dir == ED_LEFT || dir == ED_UP);
      bool ended = false;
      top--;
      if (noChains > LF_ED_CHAIN_CAP) { overflow = true; break; }
      if (E[(size_t)r * W + cc] != ED_EDGE) duplicatePixelCount++;
      chains[noChains].dir = dir; chains[noChains].parent = parent; chains[noChains].child0 = chains[noChains].child1 = -1;
      chains[noChains].pix = len;
      pixels[len] = e_mk(r, cc); len++; chainLen++;
      {
        // The walk in (across, along) coordinates: along = the column of a horizontal walk / the row of a vertical one, advancing
        // by `step` per pixel; across = the other coordinate.  A WINDOW of 7 (across, -3 .. +3) x 9 (along, 0 .. 8 steps ahead)
        // pixels of E, G and D is fetched with one load per map (lane (xo + 3) * 9 + k holds cell (xo, k)) and serves the steps
        // until the walker is within two steps of its far edge or more than two pixels off its axis -- up to eight pixels per
        // memory round trip.  The walker's own stores go to its pixel and the two pixels beside it (same along position): every
        // cell a later step consults lies further along, so the fetched values stay current.
        int x = horizontal ? r : cc, a = horizontal ? cc : r;
        int x0 = x, a0 = a, wk = 99, vE = 0, vG = 0, vD = 0;
        const int fxo = lane / 9 - 3, fk = lane - 9 * (lane / 9);
#define WCELL(v, xo, dk) __builtin_amdgcn_readlane(v, (x - x0 + (xo) + 3) * 9 + wk + (dk))
        for (;;) {
          if (wk > 7 || x - x0 > 2 || x0 - x > 2) {          // (re)fetch the window, the walker at (0, 0)
            x0 = x; a0 = a; wk = 0;
            const int px = x0 + fxo, pa = a0 + fk * step;
            const int rr = horizontal ? px : pa, c2 = horizontal ? pa : px;
            vE = 0; vG = 0; vD = 0;
            if (lane < 63 && rr >= 0 && rr < H && c2 >= 0 && c2 < W) {
              const size_t at = (size_t)rr * W + c2;
              vE = (int)E[at]; vG = (int)G[at]; vD = (int)D[at];
            }
          }
          if (WCELL(vD, 0, 0) != (horizontal ? ED_HORIZONTAL : ED_VERTICAL)) break;
          {
            const size_t at = horizontal ? (size_t)x * W + a : (size_t)a * W + x;
            const size_t sd = horizontal ? (size_t)W : (size_t)1;        // one pixel across
            E[at] = ED_EDGE;
            if (WCELL(vE, -1, 0) == ED_ANCHOR) E[at - sd] = 0;
            if (WCELL(vE, 1, 0) == ED_ANCHOR) E[at + sd] = 0;
          }
          int nx = x, enew;
          if ((enew = WCELL(vE, 0, 1)) >= ED_ANCHOR) { }
          else if ((enew = WCELL(vE, step, 1)) >= ED_ANCHOR) nx = x + step;
          else if ((enew = WCELL(vE, -step, 1)) >= ED_ANCHOR) nx = x - step;
          else {
            const int Ag = WCELL(vG, -1, 1), Bg = WCELL(vG, 0, 1), Cg = WCELL(vG, 1, 1);
            if (Ag > Bg) { if (Ag > Cg) nx = x - 1; else nx = x + 1; }
            else if (Cg > Bg) nx = x + 1;
            enew = WCELL(vE, nx - x, 1);
          }
          const int gnew = WCELL(vG, nx - x, 1);
          x = nx; a += step; wk++;
          r = horizontal ? x : a; cc = horizontal ? a : x;
          if (enew == ED_EDGE || gnew < ED_GRAD_THRESH) {
            if (chainLen > 0) {
              chains[noChains].len = chainLen;
              if (child0) chains[parent].child0 = noChains; else chains[parent].child1 = noChains;
              noChains++;
            }
            ended = true;
            break;
          }
          pixels[len] = e_mk(r, cc); len++; chainLen++;
          if (len + 2 >= (int)NP) { overflow = true; break; }
        }
#undef WCELL
      }
      if (overflow) break;
      if (ended) continue;
      if (top + 2 >= LF_ED_STACK_CAP) { overflow = true; break; }
      if (horizontal) {
        ++top; F.stack[2 * top] = e_mk(r, cc); F.stack[2 * top + 1] = ((unsigned)noChains << 3) | ED_DOWN;
        ++top; F.stack[2 * top] = e_mk(r, cc); F.stack[2 * top + 1] = ((unsigned)noChains << 3) | ED_UP;
      } else {
        ++top; F.stack[2 * top] = e_mk(r, cc); F.stack[2 * top + 1] = ((unsigned)noChains << 3) | ED_RIGHT;
        ++top; F.stack[2 * top] = e_mk(r, cc); F.stack[2 * top + 1] = ((unsigned)noChains << 3) | ED_LEFT;
      }
      len--; chainLen--;
      chains[noChains].len = chainLen;
      if (child0) chains[parent].child0 = noChains; else chains[parent].child1 = noChains;
      noChains++;
    }
    if (overflow) break;
    if (len - duplicatePixelCount < ED_MIN_PATH) {
      for (int q = lane; q < len; q += 64) E[(size_t)e_pr(pixels[q]) * W + e_pc(pixels[q])] = 0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (lane-strided stores, read back by every lane as uniform loads: ordered explicitly)
      continue;
    }
    if ((size_t)nsegpix + (size_t)len + 2 >= NP) { overflow = true; break; }
    {
      unsigned *seg = segpix + nsegpix;
      int n = 0, totalLen, count;
      totalLen = e_longest_chain(chains, chains[0].child1);
      if (totalLen > 0) {
        count = e_retrieve_chain_nos(chains, chains[0].child1, chainNos, nos_cap);
        if (count < 0) { overflow = true; break; }
        for (int q = count - 1; q >= 0; q--) {                 // these chains backwards
          const int cn = chainNos[q];
          const unsigned *cp = pixels + chains[cn].pix;
          unsigned fp = cp[chains[cn].len - 1];
          int index = n - 2;
          while (index >= 0) { if (e_near(fp, seg[index])) { n--; index--; } else break; }
          if (chains[cn].len > 1 && n > 0) { fp = cp[chains[cn].len - 2]; if (e_near(fp, seg[n - 1])) chains[cn].len--; }
          { const int cl = chains[cn].len; for (int l = lane; l < cl; l += 64) seg[n + l] = cp[cl - 1 - l]; if (cl > 0) n += cl; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }   // (64 pixels per trip: the lanes share the copy)
          chains[cn].len = 0;
        }
      }
      totalLen = e_longest_chain(chains, chains[0].child0);
      if (totalLen > 1) {
        count = e_retrieve_chain_nos(chains, chains[0].child0, chainNos, nos_cap);
        if (count < 0) { overflow = true; break; }
        chains[chainNos[0]].pix++; chains[chainNos[0]].len--;   // the anchor itself is already there
        for (int q = 0; q < count; q++) {
          const int cn = chainNos[q];
          const unsigned *cp = pixels + chains[cn].pix;
          int index = n - 2, startIndex = 0;
          while (index >= 0) { if (e_near(cp[0], seg[index])) { n--; index--; } else break; }
          if (chains[cn].len > 1 && n > 0) { if (e_near(cp[1], seg[n - 1])) startIndex = 1; }
          { const int cl = chains[cn].len - startIndex; for (int l = lane; l < cl; l += 64) seg[n + l] = cp[startIndex + l]; if (cl > 0) n += cl; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
          chains[cn].len = 0;
        }
      }
      if (n > 1 && e_near(seg[1], seg[n - 1])) { seg++; n--; }   // first pixel of a loop
      if (nsegments >= c.segtab_cap) { overflow = true; break; }
      F.segtab[2 * nsegments] = (int)(seg - segpix); F.segtab[2 * nsegments + 1] = n; nsegments++;
      nsegpix = (int)(seg - segpix) + n;
      for (int q = 2; q < noChains; q++) {       // the rest of the tree: every remaining path of at least ten pixels
        if (chains[q].len < 2) continue;
        totalLen = e_longest_chain(chains, q);
        if (totalLen < 10) continue;
        count = e_retrieve_chain_nos(chains, q, chainNos, nos_cap);
        if (count < 0) { overflow = true; break; }
        seg = segpix + nsegpix; n = 0;
        for (int qq = 0; qq < count; qq++) {
          const int cn = chainNos[qq];
          const unsigned *cp = pixels + chains[cn].pix;
          int index = n - 2, startIndex = 0;
          while (index >= 0) { if (e_near(cp[0], seg[index])) { n--; index--; } else break; }
          if (chains[cn].len > 1 && n > 0) { if (e_near(cp[1], seg[n - 1])) startIndex = 1; }
          { const int cl = chains[cn].len - startIndex; for (int l = lane; l < cl; l += 64) seg[n + l] = cp[startIndex + l]; if (cl > 0) n += cl; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
          chains[cn].len = 0;
        }
        if (nsegments >= c.segtab_cap) { overflow = true; break; }
        F.segtab[2 * nsegments] = nsegpix; F.segtab[2 * nsegments + 1] = n; nsegments++;
        nsegpix += n;
      }
    }
  }
  }
  // ---- the edge segments are complete: the line stage runs on them with one lane per segment / per line (kernels below)
  b.nsegtab[f] = overflow ? -1 : nsegments;
}

__device__ __forceinline__ void e_bind_lines(EdFrame &F, const EdConsts &c, const EdBuffers &b, int f) {
  const size_t NP = (size_t)c.W * c.H;
  F.G = nullptr; F.D = nullptr; F.E = nullptr; F.W = c.W; F.H = c.H;
  F.img = b.gray + (size_t)f * b.gray_frame_stride; F.img_stride = b.gray_row_stride;
  F.walk = nullptr; F.stack = nullptr; F.segpix = b.segpix + f * NP; F.ch = nullptr; F.chain_nos = nullptr;
  F.segtab = b.segtab + (size_t)f * c.segtab_cap * 2; F.rect = nullptr;
  F.lines = b.lines + (size_t)f * LF_ED_LINE_CAP;
  F.kmin = b.kmin; F.atan_lut = b.atan_lut; F.lut_size = c.lut_size; F.nmax = c.nmax;
}
// SplitSegment2Lines, first pass: ONE LANE PER EDGE SEGMENT counts the lines its segment splits into (the segments are
// independent of each other; the binary's line list is their concatenation in segment order)
__global__ void __launch_bounds__(64) k_ed_split_count(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, s = blockIdx.x * 64 + (int)threadIdx.x;
  const int nseg = b.nsegtab[f];
  if (s >= nseg) return;
  const int *segtab = b.segtab + (size_t)f * c.segtab_cap * 2;
  const unsigned *segpix = b.segpix + (size_t)f * c.W * c.H;
  b.seg_nl[((size_t)f * c.segtab_cap + s) * 3] = e_split_segment(segpix + segtab[2 * s], segtab[2 * s + 1], s, c.min_len, nullptr);
}
// ordered offsets: one wavefront per frame scans the per-segment counts (which = 0: lines before the joining -> seg_nl[.][1] =
// first slot of the segment, nslots[f] = all of them, frame over capacity beyond LF_ED_LINE_CAP as the sequential list)
__global__ void __launch_bounds__(64) k_ed_scan_segments(EdConsts c, EdBuffers b) {
  const int f = blockIdx.x, lane = (int)threadIdx.x;
  const int nseg = b.nsegtab[f];
  int *nl = b.seg_nl + (size_t)f * c.segtab_cap * 3;
  int run = 0;
  for (int s0 = 0; s0 < nseg; s0 += 64) {
    const int s = s0 + lane;
    int v = s < nseg ? nl[3 * s] : 0, inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (s < nseg) nl[3 * s + 1] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) b.nslots[f] = nseg < 0 ? -1 : (run > LF_ED_LINE_CAP ? -1 : run);
}
// second pass: the lines of the segment at their slots, then JoinCollinearLines inside the segment (it only ever joins lines
// of one segment: consecutive ones, then the last to the first).  Slots the joining frees are marked dead (len = -1).
__global__ void __launch_bounds__(64) k_ed_split_join(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, s = blockIdx.x * 64 + (int)threadIdx.x;
  const int nseg = b.nsegtab[f];
  if (s >= nseg || b.nslots[f] < 0) return;
  const int *segtab = b.segtab + (size_t)f * c.segtab_cap * 2;
  const unsigned *segpix = b.segpix + (size_t)f * c.W * c.H;
  {
    const int *nl = b.seg_nl + ((size_t)f * c.segtab_cap + s) * 3;
    EdLine *L = b.lines + (size_t)f * LF_ED_LINE_CAP + nl[1];
    const int nlines = e_split_segment(segpix + segtab[2 * s], segtab[2 * s + 1], s, c.min_len, L);
    int last = -1;
    if (nlines > 0) {
      last = 0;
      for (int j = 1; j < nlines; j++) {
        if (!e_try_to_join(&L[last], &L[j], ED_MAX_DIST, ED_MAX_ERROR)) {
          last++;
          if (last != j) L[last] = L[j];
        }
      }
      if (last != 0) { if (e_try_to_join(&L[0], &L[last], ED_MAX_DIST, ED_MAX_ERROR)) last--; }
    }
    for (int j = last + 1; j < nlines; j++) L[j].len = -1;
  }
}
// ValidateLineSegments: ONE LANE PER LINE SLOT
__global__ void __launch_bounds__(64) k_ed_validate(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y;
  const int nslots = b.nslots[f];
  EdFrame F;
  e_bind_lines(F, c, b, f);
  for (int i = blockIdx.x * 64 + (int)threadIdx.x; i < nslots; i += gridDim.x * 64) {
  const EdLine *ls = &F.lines[i];
  int valid = 0;
  if (ls->len >= 80) valid = 1;
  else if (ls->len < 0) valid = 0;                       // joined into an earlier line of its segment
  else if (ls->len <= 25) valid = e_validate_rect(F, ls);
  else {
    const double lineAngle = e_line_angle(ls);
    const unsigned *px = F.segpix + F.segtab[2 * ls->segmentNo] + ls->firstPixelIndex;
    int aligned = 0, count = 0;
    for (int q = 0; q < ls->len; q++) {
      const int r = e_pr(px[q]), cc = e_pc(px[q]);
      if (r <= 0 || r >= c.H - 1 || cc <= 0 || cc >= c.W - 1) continue;
      count++;
      if (e_aligned(F, r, cc, lineAngle)) aligned++;
    }
    valid = e_check_nfa(F, count, aligned);
    if (!valid) valid = e_validate_rect(F, ls);
  }
  b.lvalid[(size_t)f * LF_ED_LINE_CAP + i] = (uint8_t)valid;
  }
}
// the valid lines in list order -> segment rows: one wavefront per frame
__global__ void __launch_bounds__(64) k_ed_emit(EdConsts c, EdBuffers b) {
  const int f = blockIdx.x, lane = (int)threadIdx.x;
  const int nslots = b.nslots[f];
  double *segs = b.segs + (size_t)f * c.seg_cap * 5;
  if (nslots < 0) {
    // an internal capacity was exceeded: the frame is reported as over capacity (nsegs > seg_cap: the getters return
    // LF_ERR_CAPACITY), and its rows become zero-length segments -- the 3D stage treats all seg_cap rows of an over-capacity
    // frame as present, and its length filter drops these instead of reading stale rows
    for (int i = lane; i < c.seg_cap * 5; i += 64) segs[i] = 0.0;
    if (lane == 0) b.nsegs[f] = c.seg_cap + 1;
    return;
  }
  const EdLine *L = b.lines + (size_t)f * LF_ED_LINE_CAP;
  const uint8_t *lv = b.lvalid + (size_t)f * LF_ED_LINE_CAP;
  int nout = 0;
  for (int i0 = 0; i0 < nslots; i0 += 64) {
    const int i = i0 + lane;
    const bool v = i < nslots && lv[i] != 0;
    const unsigned long long m = __ballot(v);
    if (v) {
      const int at = nout + __popcll(m & ((1ull << lane) - 1ull));
      if (at < c.seg_cap) { double *o = segs + 5 * (size_t)at; o[0] = L[i].sx; o[1] = L[i].sy; o[2] = L[i].ex; o[3] = L[i].ey; o[4] = 0.0; }
    }
    nout += __popcll(m);
  }
  if (lane == 0) b.nsegs[f] = nout;
}

void lf_edlines_launch(const EdConsts &c, const EdBuffers &b, int B, hipStream_t st) {
  (void)hipMemsetAsync(b.hist, 0, sizeof(int) * (size_t)B * LF_ED_BINS, st);
  hipLaunchKernelGGL(k_ed_smooth, dim3((c.W + ES_TW - 1) / ES_TW, (c.H + ES_TH - 1) / ES_TH, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_gradient, dim3((c.W * c.H + 255) / 256, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_anchor, dim3((c.W * c.H + 255) / 256, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_sort, dim3(B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_link, dim3(B), dim3(64), 0, st, c, b);
  // one lane per segment slot (blocks past the frame's segment count exit at once; a strided loop over fewer blocks was measured
  // 3-5x slower); the validation strides 16 blocks per frame over its few hundred lines (2.1 -> 1.2 ms)
  const int sblocks = (c.segtab_cap + 63) / 64;
  hipLaunchKernelGGL(k_ed_split_count, dim3(sblocks, B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_scan_segments, dim3(B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_split_join, dim3(sblocks, B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_validate, dim3(16, B), dim3(64), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_emit, dim3(B), dim3(64), 0, st, c, b);
}
