// lf_edlines.hip -- EDLines for gfx950, batched over frames.  The reference has this detector as a
// binary only (external/EDLines/libEDLines.a); what is implemented is the detector of the two papers it comes from (Edge
// Drawing, JVCIR 2012; EDLines, PRL 2011) as stated in oracle/edlines_oracle.c -- parity with the binary is UNPINNED and
// approximate (tests/test_oracle_edlines.py measures it on the one example the reference ships).
//
//   k_ed_smooth    cvSmooth(CV_GAUSSIAN, 5x5, sigma 1) on 8-bit: fixed-point taps, BORDER_REPLICATE, LDS tile + halo
//   k_ed_gradient  Sobel |gx| + |gy|, threshold 36, direction map
//   k_ed_anchor    anchors (local maxima across the edge by >= 8) -> per-frame key list (gradient descending, scan order)
//   k_ed_sort      one 1024-thread workgroup per frame: bitonic sort of the 32-bit keys in LDS (128 KB)
//   k_ed_link      ONE WAVEFRONT PER FRAME, frames in flight are the parallel axis (as the LSD sweep): smart routing from the
//                  anchors in sorted order, least-squares line fitting along each chain in chain order, Helmholtz validation.
//                  A dependent chain per frame by nature (every step reads what the previous one marked); fp64 sums in the
//                  oracle's order.
#include "lf_edlines.h"
#include "lf_math.h"

#define ED_GRAD_THRESH 36
#define ED_ANCHOR_THRESH 8
#define ED_HORIZONTAL 1
#define ED_VERTICAL 2
#define ED_LINE_ERROR 1.0
#define ED_MAX_BAD 5

__device__ __forceinline__ int e_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int e_cvround(double v) { return (int)__builtin_rint(v); }

#define ES_TW 64
#define ES_TH 16
__global__ void __launch_bounds__(256) k_ed_smooth(EdConsts c, EdBuffers b) {
  __shared__ uint8_t s_in[ES_TH + 4][ES_TW + 4];
  __shared__ int s_row[ES_TH + 4][ES_TW];
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
    for (int k = 0; k < 5; k++) s += c.sk[k] * s_in[ty][tx + k];
    s_row[ty][tx] = s;
  }
  __syncthreads();
  for (int i = tid; i < ES_TH * ES_TW; i += 256) {
    const int ty = i / ES_TW, tx = i - ty * ES_TW, x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
      int s = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) s += c.sk[k] * s_row[ty + k][tx];
      const int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

__global__ void __launch_bounds__(256) k_ed_gradient(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, W = c.W, H = c.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const uint8_t *s = b.smooth + (size_t)f * W * H;
  int g = 0, d = 0;
  if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
    const uint8_t *p = s + i;
    const int c1 = (int)p[W + 1] - p[-W - 1], c2 = (int)p[-W + 1] - p[W - 1];
    const int gx = abs(c1 + c2 + 2 * ((int)p[1] - p[-1])), gy = abs(c1 - c2 + 2 * ((int)p[W] - p[-W]));
    const int gg = gx + gy;
    if (gg >= ED_GRAD_THRESH) { g = gg; d = gx >= gy ? ED_VERTICAL : ED_HORIZONTAL; }
  }
  b.G[(size_t)f * W * H + i] = (int16_t)g;
  b.D[(size_t)f * W * H + i] = (uint8_t)d;
  b.E[(size_t)f * W * H + i] = 0;
}

__global__ void __launch_bounds__(256) k_ed_anchor(EdConsts c, EdBuffers b) {
  const int f = blockIdx.y, W = c.W, H = c.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  if (!(y >= 2 && y < H - 2 && x >= 2 && x < W - 2)) return;
  const int16_t *G = b.G + (size_t)f * W * H;
  const int g = G[i];
  if (!g) return;
  bool a;
  if (b.D[(size_t)f * W * H + i] == ED_VERTICAL) a = g - G[i - 1] >= ED_ANCHOR_THRESH && g - G[i + 1] >= ED_ANCHOR_THRESH;
  else a = g - G[i - W] >= ED_ANCHOR_THRESH && g - G[i + W] >= ED_ANCHOR_THRESH;
  if (a) {
    const int at = atomicAdd(&b.nanch[f], 1);
    if (at < LF_ED_ANCHOR_CAP) b.akeys[(size_t)f * LF_ED_ANCHOR_CAP + at] = ((unsigned)(4095 - g) << 19) | (unsigned)i;
  }
}

__global__ void __launch_bounds__(1024) k_ed_sort(EdBuffers b) {
  __shared__ unsigned k[LF_ED_ANCHOR_CAP];
  const int f = blockIdx.x, tid = threadIdx.x;
  int n = b.nanch[f];
  if (n > LF_ED_ANCHOR_CAP) n = LF_ED_ANCHOR_CAP;
  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  unsigned *g = b.akeys + (size_t)f * LF_ED_ANCHOR_CAP;
  for (int i = tid; i < n2; i += 1024) k[i] = i < n ? g[i] : 0xffffffffu;
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < n2 / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned a = k[lo], bb = k[hi];
        if ((a > bb) == up) { k[lo] = bb; k[hi] = a; }
      }
    }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) g[i] = k[i];
}

// ---------------------------------------------------------------------------------------------------- linking + fitting
struct EdView { const int16_t *G; const uint8_t *D; uint8_t *E; int W, H; };
__device__ int e_best3(const EdView &v, int x, int y, int dx, int dy, int *nx, int *ny) {
  int bg = -1;
  const int order[3] = {0, -1, 1};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int o = order[k], cx = dx ? x + dx : x + o, cy = dy ? y + dy : y + o, g = v.G[(size_t)cy * v.W + cx];
    if (g > bg) { bg = g; *nx = cx; *ny = cy; }
  }
  return bg;
}
__device__ int e_walk(const EdView &v, int x, int y, int dir, unsigned *chain, int cap) {
  int n = 0;
  while (x >= 1 && y >= 1 && x < v.W - 1 && y < v.H - 1 && v.G[(size_t)y * v.W + x] > 0 && !v.E[(size_t)y * v.W + x]) {
    int nx = x, ny = y;
    const int d = v.D[(size_t)y * v.W + x];
    v.E[(size_t)y * v.W + x] = 1;
    if (n < cap) chain[n] = ((unsigned)y << 16) | (unsigned)x;
    n++;
    if (d == ED_HORIZONTAL) {
      if (dir > 1) { int ax, ay, bx, by; const int gl = e_best3(v, x, y, -1, 0, &ax, &ay), gr = e_best3(v, x, y, 1, 0, &bx, &by); dir = gr > gl ? 1 : 0; }
      e_best3(v, x, y, dir == 0 ? -1 : 1, 0, &nx, &ny);
    } else {
      if (dir < 2) { int ax, ay, bx, by; const int gu = e_best3(v, x, y, 0, -1, &ax, &ay), gd = e_best3(v, x, y, 0, 1, &bx, &by); dir = gd > gu ? 3 : 2; }
      e_best3(v, x, y, 0, dir == 2 ? -1 : 1, &nx, &ny);
    }
    x = nx; y = ny;
  }
  return n;
}
// pixel i of the chain reverse(walk 1) + walk 2 without the repeated anchor
struct EdChain { const unsigned *c1, *c2; int n1; };
__device__ __forceinline__ void e_px(const EdChain &ch, int i, double *x, double *y) {
  const unsigned e = i < ch.n1 ? ch.c1[ch.n1 - 1 - i] : ch.c2[i - ch.n1 + 1];
  *x = (double)(e & 0xffffu); *y = (double)(e >> 16);
}
__device__ void e_line_fit(const EdChain &ch, int off, int count, double *a, double *b, int *invert, double *err) {
  double Sx = 0, Sy = 0, Sxx = 0, Sxy = 0, dx = 0, dy = 0, e = 0;
  for (int i = 0; i < count; i++) { double x, y; e_px(ch, off + i, &x, &y); Sx += x; Sy += y; }
  const double mx = Sx / count, my = Sy / count;
  for (int i = 0; i < count; i++) { double x, y; e_px(ch, off + i, &x, &y); dx += (x - mx) * (x - mx); dy += (y - my) * (y - my); }
  const int inv = dx < dy;
  if (inv) { const double t = Sx; Sx = Sy; Sy = t; }
  for (int i = 0; i < count; i++) { double x, y; e_px(ch, off + i, &x, &y); const double u = inv ? y : x, v = inv ? x : y; Sxx += u * u; Sxy += u * v; }
  const double D = count * Sxx - Sx * Sx;
  *a = (Sxx * Sy - Sx * Sxy) / D;
  *b = (count * Sxy - Sx * Sy) / D;
  *invert = inv;
  if (err) {
    for (int i = 0; i < count; i++) { double x, y; e_px(ch, off + i, &x, &y); const double u = inv ? y : x, v = inv ? x : y; const double r = (*a + *b * u - v); e += r * r / (1 + *b * *b); }
    *err = lf_sqrt(e / count);
  }
}
__device__ __forceinline__ double e_dist(double px, double py, double a, double b, int invert) {
  const double u = invert ? py : px, v = invert ? px : py;
  return lf_fabs(a + b * u - v) / lf_sqrt(1 + b * b);
}
__device__ __forceinline__ void e_closest(double px, double py, double a, double b, int invert, double *ox, double *oy) {
  const double u = invert ? py : px, v = invert ? px : py;
  const double uu = (u + b * (v - a)) / (1 + b * b), vv = a + b * uu;
  if (invert) { *ox = vv; *oy = uu; } else { *ox = uu; *oy = vv; }
}
__device__ int e_validate(const uint8_t *img, int stride, int w, int h, double sx, double sy, double ex, double ey, const int *kmin, int nmax) {
  const double dx = ex - sx, dy = ey - sy, len = lf_sqrt(dx * dx + dy * dy), tol = 3.14159265358979323846 / 8;
  const int steps = (int)(lf_fabs(dx) > lf_fabs(dy) ? lf_fabs(dx) : lf_fabs(dy));
  int n = 0, k = 0;
  if (len <= 0 || steps < 1) return 0;
  const double la = lf_atan2(dy, dx);
  for (int i = 0; i <= steps; i++) {
    const int x = e_cvround(sx + dx * i / steps), y = e_cvround(sy + dy * i / steps);
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) continue;
    const uint8_t *p = img + (size_t)y * stride + x;
    const int c1 = (int)p[stride + 1] - p[-stride - 1], c2 = (int)p[-stride + 1] - p[stride - 1];
    const int gx = c1 + c2 + ((int)p[1] - p[-1]), gy = c1 - c2 + ((int)p[stride] - p[-stride]);
    n++;
    if (gx == 0 && gy == 0) continue;
    const double ga = lf_atan2((double)gx, (double)-gy);
    double d = lf_fabs(ga - la);
    while (d > 3.14159265358979323846) d = lf_fabs(d - 2 * 3.14159265358979323846);
    if (d > 3.14159265358979323846 / 2) d = 3.14159265358979323846 - d;
    if (d <= tol) k++;
  }
  if (n > nmax) n = nmax;
  return k >= kmin[n];
}

__global__ void __launch_bounds__(64) k_ed_link(EdConsts c, EdBuffers b) {
  const int f = blockIdx.x;
  if ((threadIdx.x & 63u) != 0) return;          // a dependent chain per frame: one lane walks, frames in flight fill the chip
  const int W = c.W, H = c.H;
  EdView v;
  v.G = b.G + (size_t)f * W * H; v.D = b.D + (size_t)f * W * H; v.E = b.E + (size_t)f * W * H; v.W = W; v.H = H;
  const uint8_t *img = b.gray + (size_t)f * b.gray_frame_stride;
  unsigned *c1 = b.chain + (size_t)f * 2 * c.chain_cap, *c2 = c1 + c.chain_cap;
  const unsigned *keys = b.akeys + (size_t)f * LF_ED_ANCHOR_CAP;
  double *segs = b.segs + (size_t)f * c.seg_cap * 5;
  int na = b.nanch[f];
  if (na > LF_ED_ANCHOR_CAP) na = LF_ED_ANCHOR_CAP;
  int nseg = 0;
  for (int ai = 0; ai < na; ai++) {
    const int idx = (int)(keys[ai] & 0x7ffffu), x = idx % W, y = idx / W;
    if (v.E[idx]) continue;
    int n1, n2;
    if (v.D[idx] == ED_HORIZONTAL) { n1 = e_walk(v, x, y, 0, c1, c.chain_cap); v.E[idx] = 0; n2 = e_walk(v, x, y, 1, c2, c.chain_cap); }
    else { n1 = e_walk(v, x, y, 2, c1, c.chain_cap); v.E[idx] = 0; n2 = e_walk(v, x, y, 3, c2, c.chain_cap); }
    if (n1 > c.chain_cap) n1 = c.chain_cap;
    if (n2 > c.chain_cap) n2 = c.chain_cap;
    EdChain ch;
    ch.c1 = c1; ch.c2 = c2; ch.n1 = n1;
    const int n = n1 + (n2 > 0 ? n2 - 1 : 0);
    int off = 0;
    while (n - off >= c.min_len) {
      const int left = n - off;
      int inv = 0, len, index;
      double a = 0, bb = 0, err = 0;
      e_line_fit(ch, off, c.min_len, &a, &bb, &inv, &err);
      if (err > ED_LINE_ERROR) { off++; continue; }
      len = c.min_len; index = c.min_len;
      bool done = false;
      while (!done) {
        const int start = index;
        int last_good = index - 1, good = 0, bad = 0;
        while (index < left) {
          double px, py;
          e_px(ch, off + index, &px, &py);
          if (e_dist(px, py, a, bb, inv) <= ED_LINE_ERROR) { last_good = index; good++; bad = 0; }
          else if (++bad >= ED_MAX_BAD) break;
          index++;
        }
        if (good >= 2) {
          len += last_good - start + 1;
          e_line_fit(ch, off, len, &a, &bb, &inv, nullptr);
          index = last_good + 1;
        }
        if (good < 2 || index >= left) {
          double sx, sy, ex, ey, px, py;
          int i0 = 0, i1 = len - 1;
          for (;;) { e_px(ch, off + i0, &px, &py); if (!(i0 < len - 1 && e_dist(px, py, a, bb, inv) > ED_LINE_ERROR)) break; i0++; }
          e_closest(px, py, a, bb, inv, &sx, &sy);
          for (;;) { e_px(ch, off + i1, &px, &py); if (!(i1 > i0 && e_dist(px, py, a, bb, inv) > ED_LINE_ERROR)) break; i1--; }
          e_closest(px, py, a, bb, inv, &ex, &ey);
          if (e_validate(img, b.gray_row_stride, W, H, sx, sy, ex, ey, b.kmin, c.nmax)) {
            if (nseg < c.seg_cap) { double *o = segs + 5 * (size_t)nseg; o[0] = sx; o[1] = sy; o[2] = ex; o[3] = ey; o[4] = 0.0; }
            nseg++;
          }
          done = true;
        }
      }
      off += len;
    }
  }
  b.nsegs[f] = nseg;
}

void lf_edlines_launch(const EdConsts &c, const EdBuffers &b, int B, hipStream_t st) {
  (void)hipMemsetAsync(b.nanch, 0, sizeof(int) * (size_t)B, st);
  hipLaunchKernelGGL(k_ed_smooth, dim3((c.W + ES_TW - 1) / ES_TW, (c.H + ES_TH - 1) / ES_TH, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_gradient, dim3((c.W * c.H + 255) / 256, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_anchor, dim3((c.W * c.H + 255) / 256, B), dim3(256), 0, st, c, b);
  hipLaunchKernelGGL(k_ed_sort, dim3(B), dim3(1024), 0, st, b);
  hipLaunchKernelGGL(k_ed_link, dim3(B), dim3(64), 0, st, c, b);
}
