// lf_orb.hip -- the ORB extractor for gfx950, batched over frames: the ORB branch of Node::Node
// (src/node.cpp:222-290) = AorbFeatureDetector(10000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, fastThreshold) (src/aorb.cpp:727-940,
// src/feature_adjuster.cpp:86-89), removeDepthless + retainBest(max_keypoints) (node.cpp:101-125, 257-263) and
// OrbDescriptorExtractor::compute (src/features.cpp:197).  Integer / byte work, HBM- and LDS-bound; no MFMA.
//
//   k_orb_level0 / k_orb_resize   pyramid: level l from level l-1 with cv::resize's 11-bit fixed-point bilinear arithmetic
//   k_orb_blur                    cv::GaussianBlur 7x7 sigma 2 on 8-bit (8-bit fixed-point taps), LDS tile + halo, reflect-101
//   k_orb_fast                    FAST-9/16 segment test + cornerScore<16> per pixel -> score image
//   k_orb_nms                     3x3 strict non-maximum suppression + border filter -> per-frame candidate list + score histograms
//   k_orb_select                  ONE 1024-THREAD WORKGROUP PER FRAME: retainBest per level (FAST score, then Harris response),
//                                 removeDepthless, the max_keypoints best by (response desc, detection order asc) -- bitonic sorts of
//                                 64-bit keys in LDS (128 KB) -- clustering by octave, border filter
//   k_orb_describe                one wavefront per key point: IC_Angle (intensity centroid, cv::fastAtan2) and the 256 rBRIEF
//                                 tests steered by the angle (four tests per lane, combined with shuffles)
// Sequential twin: oracle/orb_oracle.c (OpenCV 2.4 itself is absent: parity unpinned, see there for every restated algorithm
// and for the one documented deviation, the order retainBest leaves its survivors in).
#include "lf_orb.h"
#include "lf_math.h"
#include "lf_orb_pattern.h"

typedef unsigned long long u64;
__constant__ signed char c_orb_pattern[1024];

__device__ __forceinline__ int o_cvround(double v) { return (int)__builtin_rint(v); }      // round half to even
__device__ __forceinline__ int o_reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}

__global__ void __launch_bounds__(256) k_orb_level0(OrbConsts c, OrbBuffers b) {
  const int f = blockIdx.y;
  const uint8_t *g = b.gray + (size_t)f * b.gray_frame_stride;
  uint8_t *dst = b.pyr + (size_t)f * c.total;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < c.W * c.H; i += gridDim.x * 256) {
    const int y = i / c.W, x = i - y * c.W;
    dst[i] = g[(size_t)y * b.gray_row_stride + x];
  }
}

// cv::resize INTER_LINEAR, CV_8UC1 (imgwarp.cpp): thread per destination pixel, both passes in registers
__global__ void __launch_bounds__(256) k_orb_resize(OrbConsts c, OrbBuffers b, int l) {
  const int f = blockIdx.y, dw = c.lw[l], dh = c.lh[l], sw = c.lw[l - 1], sh = c.lh[l - 1];
  const uint8_t *src = b.pyr + (size_t)f * c.total + c.loff[l - 1];
  uint8_t *dst = b.pyr + (size_t)f * c.total + c.loff[l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= dw * dh) return;
  const int dy = i / dw, dx = i - dy * dw;
  float fx = (float)((dx + 0.5) * c.scale_x[l] - 0.5);
  int sx = (int)__builtin_floorf(fx);
  fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
  const int a0 = (short)o_cvround((double)((1.f - fx) * 2048)), a1 = (short)o_cvround((double)(fx * 2048));
  float fy = (float)((dy + 0.5) * c.scale_y[l] - 0.5);
  int sy = (int)__builtin_floorf(fy);
  fy -= sy;
  if (sy < 0) { fy = 0; sy = 0; }
  if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
  const int b0 = (short)o_cvround((double)((1.f - fy) * 2048)), b1 = (short)o_cvround((double)(fy * 2048));
  const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1, sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
  const int r0 = src[(size_t)sy * sw + sx] * a0 + src[(size_t)sy * sw + sx1] * a1;
  const int r1 = src[(size_t)sy1 * sw + sx] * a0 + src[(size_t)sy1 * sw + sx1] * a1;
  dst[i] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// cv::GaussianBlur 7x7 on CV_8U: integer taps (sum 257 for sigma 2), row pass exact in int, column pass + (1 << 15) >> 16
#define OB_TW 64
#define OB_TH 16
__global__ void __launch_bounds__(256) k_orb_blur(OrbConsts c, OrbBuffers b, int l) {
  __shared__ uint8_t s_in[OB_TH + 6][OB_TW + 6];
  __shared__ int s_row[OB_TH + 6][OB_TW];
  const int f = blockIdx.z, W = c.lw[l], H = c.lh[l], tid = threadIdx.x;
  const uint8_t *src = b.pyr + (size_t)f * c.total + c.loff[l];
  uint8_t *dst = b.blur + (size_t)f * c.total + c.loff[l];
  const int x0 = blockIdx.x * OB_TW, y0 = blockIdx.y * OB_TH;
  for (int i = tid; i < (OB_TH + 6) * (OB_TW + 6); i += 256) {
    const int ty = i / (OB_TW + 6), tx = i - ty * (OB_TW + 6);
    s_in[ty][tx] = src[(size_t)o_reflect101(y0 + ty - 3, H) * W + o_reflect101(x0 + tx - 3, W)];
  }
  __syncthreads();
  for (int i = tid; i < (OB_TH + 6) * OB_TW; i += 256) {
    const int ty = i / OB_TW, tx = i - ty * OB_TW;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) s += c.blur_k[k] * s_in[ty][tx + k];
    s_row[ty][tx] = s;
  }
  __syncthreads();
  for (int i = tid; i < OB_TH * OB_TW; i += 256) {
    const int ty = i / OB_TW, tx = i - ty * OB_TW, x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
      int s = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) s += c.blur_k[k] * s_row[ty + k][tx];
      int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

// FAST_t<16> (fast.cpp) + cornerScore<16> (fast_score.cpp): thread per pixel of a level
__device__ __forceinline__ int o_imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int o_imax(int a, int b) { return a > b ? a : b; }
__global__ void __launch_bounds__(256) k_orb_fast(OrbConsts c, OrbBuffers b, int l) {
  const int f = blockIdx.y, W = c.lw[l], H = c.lh[l];
  const uint8_t *img = b.pyr + (size_t)f * c.total + c.loff[l];
  uint8_t *score = b.score + (size_t)f * c.total + c.loff[l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  int out = 0;
  if (y >= 3 && y < H - 3 && x >= 3 && x < W - 3) {
    const int cx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int cy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    int thr = c.fast_threshold;
    thr = thr < 0 ? 0 : (thr > 255 ? 255 : thr);
    const int v = img[i];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - (int)img[(size_t)(y + cy[k]) * W + x + cx[k]];
    // segment test: nine contiguous circle pixels darker than v - t (d > t) or brighter than v + t (d < -t)
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { dark |= (unsigned)(d[k] > thr) << k; bright |= (unsigned)(d[k] < -thr) << k; }
    dark |= dark << 16; bright |= bright << 16;
    unsigned rd = dark, rb = bright;
#pragma unroll
    for (int k = 1; k < 9; k++) { rd &= dark >> k; rb &= bright >> k; }
    if (((rd | rb) & 0xffffu) != 0) {
      int a0 = thr;
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        int a = o_imin(d[(k + 1) & 15], d[(k + 2) & 15]);
        a = o_imin(a, d[(k + 3) & 15]);
        if (a <= a0) continue;
        a = o_imin(a, d[(k + 4) & 15]); a = o_imin(a, d[(k + 5) & 15]); a = o_imin(a, d[(k + 6) & 15]);
        a = o_imin(a, d[(k + 7) & 15]); a = o_imin(a, d[(k + 8) & 15]);
        a0 = o_imax(a0, o_imin(a, d[k & 15]));
        a0 = o_imax(a0, o_imin(a, d[(k + 9) & 15]));
      }
      int b0 = -a0;
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        int bb = o_imax(d[(k + 1) & 15], d[(k + 2) & 15]);
        bb = o_imax(bb, d[(k + 3) & 15]); bb = o_imax(bb, d[(k + 4) & 15]); bb = o_imax(bb, d[(k + 5) & 15]);
        if (bb >= b0) continue;
        bb = o_imax(bb, d[(k + 6) & 15]); bb = o_imax(bb, d[(k + 7) & 15]); bb = o_imax(bb, d[(k + 8) & 15]);
        b0 = o_imin(b0, o_imax(bb, d[k & 15]));
        b0 = o_imin(b0, o_imax(bb, d[(k + 9) & 15]));
      }
      out = -b0 - 1;
    }
  }
  score[i] = (uint8_t)out;
}

// 3x3 strict non-maximum suppression (fast.cpp) + KeyPointsFilter::runByImageBorder(edgeThreshold 31): candidates of all levels
// into one per-frame list (unordered: the selection sorts), FAST score histograms per level for retainBest(2 n)
// MODE 0: candidates + histogram at the threshold the score image was made with.  MODE 1: histogram only (the adjuster's
// counting pass at its base threshold).  MODE 2: candidates + histogram of the corners with score >= thr_frame[f] -- a corner
// at threshold t is a pixel whose score (largest threshold it survives, minus one) is >= t, and raising the threshold can
// neither free a suppressed pixel (its stronger neighbour is still a corner) nor suppress a new one, so the non-maximum-
// suppressed corners at t are those of the base threshold with score >= t, exactly.
template <int MODE>
__global__ void __launch_bounds__(256) k_orb_nms(OrbConsts c, OrbBuffers b, int l) {
  const int f = blockIdx.y, W = c.lw[l], H = c.lh[l];
  const uint8_t *sc = b.score + (size_t)f * c.total + c.loff[l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  if (!(x >= LF_ORB_EDGE && x < W - LF_ORB_EDGE && y >= LF_ORB_EDGE && y < H - LF_ORB_EDGE)) return;
  const int s = sc[i];
  if (!s) return;
  if (MODE == 2 && s < b.thr_frame[f]) return;
  if (s > sc[i + 1] && s > sc[i - 1] && s > sc[i - W - 1] && s > sc[i - W] && s > sc[i - W + 1] && s > sc[i + W - 1] && s > sc[i + W] &&
      s > sc[i + W + 1]) {
    if (MODE != 1) {
      const int at = atomicAdd(&b.ncand[f], 1);
      if (at < LF_ORB_CAND_CAP) b.cand[(size_t)f * LF_ORB_CAND_CAP + at] = ((unsigned)s << 24) | ((unsigned)l << 19) | ((unsigned)y << 10) | (unsigned)x;
    }
    atomicAdd(&b.hist[((size_t)f * LF_ORB_LEVELS + l) * 256 + s], 1);
  }
}

// ---- VideoDynamicAdaptedFeatureDetector::detectImpl + DetectorAdjuster (src/feature_adjuster.cpp:107-186): the FAST threshold
// is a STATE of the detector -- multiplied by 0.7 when a detection returns fewer than min_features key points (and the
// detection repeated, at most max_iters times), by 1.3 when it returns more than max_features -- and the next frame starts
// with what the previous one left.  The number of key points AorbFeatureDetector::detect returns at threshold t is
// sum over the levels of min(#corners of the level with score >= t, nfeaturesPerLevel): suffix sums of the score histograms
// (one thread per frame and level), then ONE thread walks the frames in order.
__global__ void __launch_bounds__(64) k_orb_suffix(OrbConsts c, OrbBuffers b, int B) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * LF_ORB_LEVELS) return;
  int *h = b.hist + (size_t)i * 256;
  int acc = 0;
  for (int s = 255; s >= 0; s--) { acc += h[s]; h[s] = acc; }      // h[t] = corners with score >= t
}
__global__ void k_orb_adjust(OrbConsts c, OrbBuffers b, OrbAdjuster a, int B) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double thresh = *b.adj_state;
  for (int f = 0; f < B; f++) {
    const int *h = b.hist + (size_t)f * LF_ORB_LEVELS * 256;
    int iter = a.max_iters, t;
    do {                                                    // detect at least once
      t = (int)thresh;                                      // static_cast<int>(thresh_), feature_adjuster.cpp:86
      int tc = t < 0 ? 0 : (t > 255 ? 255 : t);             // cv::FAST clamps its threshold
      if (tc < a.base_threshold) tc = a.base_threshold;     // (cannot happen: base_threshold <= min_thresh)
      int n = 0;
      for (int l = 0; l < LF_ORB_LEVELS; l++) { const int nl = h[l * 256 + tc]; n += nl < c.nper[l] ? nl : c.nper[l]; }
      if (n < a.min_features) { thresh *= a.dec; if (thresh < a.min_thresh) thresh = a.min_thresh; }            // tooFew
      else if (n > a.max_features) { thresh *= a.inc; if (thresh > a.max_thresh) thresh = a.max_thresh; break; }   // tooMany
      else break;
      iter--;
    } while (iter > 0 && thresh > a.min_thresh && thresh < a.max_thresh);
    b.thr_frame[f] = t < 0 ? 0 : (t > 255 ? 255 : t);
  }
  *b.adj_state = thresh;
}

// ---------------------------------------------------------------------------------------------------- selection
#define OS_N 1024
// float -> unsigned with the same order (negative Harris responses exist: edges)
__device__ __forceinline__ unsigned o_f2ord(float v) { unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float o_ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
// ascending bitonic sort of n2 (power of two) keys in LDS by the whole workgroup
__device__ void o_sort(u64 *k, int n2) {
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < n2 / 2; t += OS_N) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 a = k[lo], bb = k[hi];
        if ((a > bb) == up) { k[lo] = bb; k[hi] = a; }
      }
    }
  __syncthreads();
}
__device__ __forceinline__ int o_pow2(int n) { int p = 2; while (p < n) p <<= 1; return p; }

// HarrisResponses (aorb.cpp:57-99), blockSize 7, k = 0.04
__device__ float o_harris(const uint8_t *img, int w, int x, int y) {
  float scale = (1 << 2) * 7 * 255.0f;
  scale = 1.0f / scale;
  const float s4 = scale * scale * scale * scale;
  int a = 0, bq = 0, cq = 0;
  const uint8_t *p0 = img + (size_t)(y - 3) * w + (x - 3);
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) {
      const uint8_t *p = p0 + (size_t)i * w + j;
      const int Ix = ((int)p[1] - p[-1]) * 2 + ((int)p[-w + 1] - p[-w - 1]) + ((int)p[w + 1] - p[w - 1]);
      const int Iy = ((int)p[w] - p[-w]) * 2 + ((int)p[w - 1] - p[-w - 1]) + ((int)p[w + 1] - p[-w + 1]);
      a += Ix * Ix; bq += Iy * Iy; cq += Ix * Iy;
    }
  return ((float)a * bq - (float)cq * cq - 0.04f * ((float)a + bq) * ((float)a + bq)) * s4;
}

// key layouts (ascending sort):  A: level(3) | ~resp(32) | order(22)   per-level retainBest
//                                B:            ~resp(32) | order(22)   the max_keypoints best
//                                C: level(3) | rank(12)  | order(22)   clustering by octave, rank kept inside
// order = level(3) | y(9) | x(10): the detection order of AORB (level by level, cv::FAST's row-major scan)
__global__ void __launch_bounds__(OS_N) k_orb_select(OrbConsts c, OrbBuffers b) {
  __shared__ u64 keys[LF_ORB_CAND_CAP];
  __shared__ int s_c1[LF_ORB_LEVELS], s_cnt[LF_ORB_LEVELS], s_start[LF_ORB_LEVELS + 1], s_n, s_m;
  __shared__ unsigned s_cut[LF_ORB_LEVELS];
  const int f = blockIdx.x, tid = threadIdx.x;
  const uint8_t *pyr = b.pyr + (size_t)f * c.total;
  int n = b.ncand[f];
  if (tid == 0) b.nkp[gridDim.x + f] = n > LF_ORB_CAND_CAP ? 1 : 0;       // overflow flag (reported by the host entry point)
  if (n > LF_ORB_CAND_CAP) n = LF_ORB_CAND_CAP;
  // retainBest(2 n_l) by FAST score: cut-off score per level from the histogram = the (2 n_l)-th largest score
  if (tid < LF_ORB_LEVELS) {
    const int *h = b.hist + ((size_t)f * LF_ORB_LEVELS + tid) * 256;
    int tot = 0;
    for (int s = 0; s < 256; s++) tot += h[s];
    int c1 = 0;
    const int keep = 2 * c.nper[tid];
    if (keep <= 0) c1 = 256;                               // (retainBest(0) clears the list)
    else if (tot > keep) { int acc = 0; for (int s = 255; s >= 0; s--) { acc += h[s]; if (acc >= keep) { c1 = s; break; } } }
    s_c1[tid] = c1; s_cnt[tid] = 0;
  }
  if (tid == 0) { s_n = 0; s_m = 0; }
  __syncthreads();
  const int n2 = o_pow2(n > 2 ? n : 2);
  for (int i = tid; i < n2; i += OS_N) {
    u64 key = ~0ull;
    if (i < n) {
      const unsigned e = b.cand[(size_t)f * LF_ORB_CAND_CAP + i];
      const int s = (int)(e >> 24), l = (int)((e >> 19) & 7u), y = (int)((e >> 10) & 511u), x = (int)(e & 1023u);
      if (s >= s_c1[l]) {
        const float r = o_harris(pyr + c.loff[l], c.lw[l], x, y);
        key = ((u64)l << 54) | ((u64)(~o_f2ord(r)) << 22) | (u64)(e & 0x3fffffu);
        atomicAdd(&s_cnt[l], 1);
      }
    }
    keys[i] = key;
  }
  __syncthreads();
  bool trunc = false;
  for (int l = 0; l < LF_ORB_LEVELS; l++) trunc = trunc || (s_cnt[l] > c.nper[l]);
  if (trunc) {   // retainBest(n_l) by Harris response (rare with nfeatures = 10000): everything >= the n_l-th largest response
    o_sort(keys, n2);
    if (tid == 0) { int acc = 0; for (int l = 0; l < LF_ORB_LEVELS; l++) { s_start[l] = acc; acc += s_cnt[l]; } s_start[LF_ORB_LEVELS] = acc; }
    __syncthreads();
    if (tid < LF_ORB_LEVELS) {
      const int l = tid;
      s_cut[l] = 0xffffffffu;                               // inverted response: keep keys with inv <= cut
      if (c.nper[l] <= 0) s_cut[l] = 0;                     // (nothing kept; inv == 0 cannot occur for finite responses)
      else if (s_cnt[l] > c.nper[l]) s_cut[l] = (unsigned)((keys[s_start[l] + c.nper[l] - 1] >> 22) & 0xffffffffu);
    }
    __syncthreads();
    for (int i = tid; i < n2; i += OS_N) {
      const u64 k = keys[i];
      if (k != ~0ull) {
        const int l = (int)(k >> 54);
        if ((unsigned)((k >> 22) & 0xffffffffu) > s_cut[l] || c.nper[l] <= 0) keys[i] = ~0ull;
      }
    }
    __syncthreads();
  }
  // removeDepthless (node.cpp:101-125) on the level-0 coordinates, then key B
  for (int i = tid; i < n2; i += OS_N) {
    u64 k = keys[i];
    if (k != ~0ull) {
      const int l = (int)(k >> 54), y = (int)((k >> 10) & 511u), x = (int)(k & 1023u);
      float px = (float)x, py = (float)y;
      if (l != 0) { px *= c.sf[l]; py *= c.sf[l]; }
      bool ok = !(px >= c.W || px < 0 || py >= c.H || py < 0);
      if (ok && b.depth) {
        int ry = (int)__builtin_roundf(py), rx = (int)__builtin_roundf(px);
        if (ry >= c.H) ry = c.H - 1;
        if (rx >= c.W) rx = c.W - 1;
        const float Z = b.depth[(size_t)f * b.depth_frame_stride + (size_t)ry * b.depth_row_stride + rx];
        ok = (Z == Z);
      }
      k = ok ? (k & ((1ull << 54) - 1ull)) : ~0ull;         // drop the level prefix: (inverted response, order)
      keys[i] = k;
    }
  }
  __syncthreads();
  for (int i = tid; i < n2; i += OS_N) if (keys[i] != ~0ull) atomicAdd(&s_n, 1);
  __syncthreads();
  // KeyPointsFilter::retainBest(max_keypoints) + resize ONLY when there are more key points than that (node.cpp:257-263):
  // then the order is by response; otherwise the list keeps AORB's detection order (level by level, row-major)
  const bool keep_order = s_n <= c.max_keypoints;
  if (keep_order)
    for (int i = tid; i < n2; i += OS_N) { const u64 k = keys[i]; if (k != ~0ull) keys[i] = ((k & 0x3fffffull) << 32) | ((k >> 22) & 0xffffffffull); }
  o_sort(keys, n2);
  if (keep_order) {
    for (int i = tid; i < n2; i += OS_N) { const u64 k = keys[i]; if (k != ~0ull) keys[i] = ((k & 0xffffffffull) << 22) | (k >> 32); }
    __syncthreads();
  }
  int m = s_n < c.max_keypoints ? s_n : c.max_keypoints;
  if (m > LF_ORB_KP_MAX) m = LF_ORB_KP_MAX;
  // key C: cluster by octave keeping the rank inside an octave (OrbDescriptorExtractor::compute), border filter on level 0
  const int m2 = o_pow2(m > 2 ? m : 2);
  u64 kc = ~0ull;
  if (tid < m) {
    const u64 k = keys[tid];
    const int l = (int)((k >> 19) & 7u);
    kc = ((u64)l << 44) | ((u64)tid << 32) | (u64)(k & 0xffffffffull);     // low 32 bits: nothing needed but (order); response below
    // keep the response: stash it in a parallel array (reuse the upper half of `keys`)
    keys[LF_ORB_CAND_CAP / 2 + tid] = k;
  }
  __syncthreads();
  if (tid < m2) keys[tid] = kc;
  __syncthreads();
  o_sort(keys, m2);
  if (tid < m) {
    const u64 k = keys[tid];
    const int rank = (int)((k >> 32) & 0xfffu);
    const u64 kb = keys[LF_ORB_CAND_CAP / 2 + rank];
    const int l = (int)((kb >> 19) & 7u), y = (int)((kb >> 10) & 511u), x = (int)(kb & 1023u);
    const float resp = o_ord2f(~(unsigned)((kb >> 22) & 0xffffffffu));
    float px = (float)x, py = (float)y;
    if (l != 0) { px *= c.sf[l]; py *= c.sf[l]; }
    const bool in = px >= LF_ORB_EDGE && px < c.W - LF_ORB_EDGE && py >= LF_ORB_EDGE && py < c.H - LF_ORB_EDGE;   // runByImageBorder
    // ordered compaction over the (<= 1024) threads
    keys[LF_ORB_CAND_CAP / 4 + tid] = in ? 1ull : 0ull;
    __threadfence_block();
    // (prefix below, after the barrier)
    // stash the candidate
    float *o = b.sel + ((size_t)f * LF_ORB_KP_MAX + tid) * 4;
    o[0] = px; o[1] = py; o[2] = (float)l; o[3] = resp;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < m; i++) { const int in = (int)keys[LF_ORB_CAND_CAP / 4 + i]; keys[LF_ORB_CAND_CAP / 4 + i] = in ? (u64)acc : ~0ull; acc += in; }
    s_m = acc;
  }
  __syncthreads();
  float v0 = 0, v1 = 0, v2 = 0, v3 = 0;
  u64 slot = ~0ull;
  if (tid < m) {
    slot = keys[LF_ORB_CAND_CAP / 4 + tid];
    const float *o = b.sel + ((size_t)f * LF_ORB_KP_MAX + tid) * 4;
    v0 = o[0]; v1 = o[1]; v2 = o[2]; v3 = o[3];
  }
  __syncthreads();
  if (tid < m && slot != ~0ull && (int)slot < c.kp_cap) {
    float *o = b.sel + ((size_t)f * LF_ORB_KP_MAX + (int)slot) * 4;
    o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
  }
  if (tid == 0) { b.nsel[f] = s_m < c.kp_cap ? s_m : c.kp_cap; b.nkp[f] = s_m; }
}

// cv::fastAtan2 (OpenCV 2.4 mathfuncs.cpp), degrees
__device__ float o_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  float a, cc, c2;
  if (ax >= ay) {
    cc = ay / (ax + (float)2.2204460492503131e-16);
    c2 = cc * cc;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * cc;
  } else {
    cc = ax / (ay + (float)2.2204460492503131e-16);
    c2 = cc * cc;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * cc;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// one wavefront per selected key point: IC_Angle on the level image, the descriptor on the blurred level
__global__ void __launch_bounds__(64) k_orb_describe(OrbConsts c, OrbBuffers b) {
  const int kp = blockIdx.x, f = blockIdx.y, lane = (int)(threadIdx.x & 63u);
  if (kp >= b.nsel[f]) return;
  const float *s = b.sel + ((size_t)f * LF_ORB_KP_MAX + kp) * 4;
  float px = s[0], py = s[1];
  const int l = (int)s[2];
  const float resp = s[3];
  const int W = c.lw[l];
  // level coordinates as OrbDescriptorExtractor::compute sees them: pt * (1 / scale) of the level-0 float coordinates
  float lx = px, ly = py;
  if (l != 0) { lx *= c.inv_sf[l]; ly *= c.inv_sf[l]; }
  const int cxi = o_cvround((double)lx), cyi = o_cvround((double)ly);
  const uint8_t *center = b.pyr + (size_t)f * c.total + c.loff[l] + (size_t)cyi * W + cxi;
  // IC_Angle (aorb.cpp:103-131): lanes share the rows v = 0 .. 15 (lane = v * 4 + part; each part a quarter of the row span)
  int m01 = 0, m10 = 0;
  {
    const int v = lane >> 2, part = lane & 3;
    if (v <= LF_ORB_HALF) {
      const int d = (v == 0) ? LF_ORB_HALF : c.umax[v];
      const int span = 2 * d + 1, q = (span + 3) / 4, u0 = -d + part * q, u1 = (u0 + q - 1 < d) ? u0 + q - 1 : d;
      int vs = 0;
      for (int u = u0; u <= u1; ++u) {
        if (v == 0) m10 += u * (int)center[u];
        else {
          const int vp = center[u + v * W], vm = center[u - v * W];
          vs += (vp - vm);
          m10 += u * (vp + vm);
        }
      }
      m01 = v * vs;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o, 64); m10 += __shfl_xor(m10, o, 64); }   // (integer sums: order-free)
  const float angle = o_fast_atan2((float)m01, (float)m10);
  // computeOrbDescriptor (aorb.cpp:135-183), WTA_K = 2: lane -> tests 4 lane .. 4 lane + 3
  const float arad = angle * (float)(3.1415926535897932384626433832795 / 180.f);
  double sd, cd;
  lf_sincos_cr((double)arad, &sd, &cd);
  const float a = (float)cd, bq = (float)sd;
  const uint8_t *bc = b.blur + (size_t)f * c.total + c.loff[l] + (size_t)cyi * W + cxi;
  unsigned nib = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const signed char *p = c_orb_pattern + 4 * (4 * lane + j);
    const int t0 = bc[o_cvround((double)(p[0] * bq + p[1] * a)) * W + o_cvround((double)(p[0] * a - p[1] * bq))];
    const int t1 = bc[o_cvround((double)(p[2] * bq + p[3] * a)) * W + o_cvround((double)(p[2] * a - p[3] * bq))];
    nib |= (unsigned)(t0 < t1) << j;
  }
  const unsigned hi = __shfl_down(nib, 1, 64);
  if ((lane & 1) == 0) b.desc[((size_t)f * c.kp_cap + kp) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
  if (lane == 0) {
    float ox = lx, oy = ly;
    if (l != 0) { ox *= c.sf[l]; oy *= c.sf[l]; }             // keypoint->pt *= scale on the way out (aorb.cpp:893-899)
    float *xy = b.kp_xy + ((size_t)f * c.kp_cap + kp) * 2;
    xy[0] = ox; xy[1] = oy;
    if (b.kp_meta) {
      float *mt = b.kp_meta + ((size_t)f * c.kp_cap + kp) * 4;
      mt[0] = resp; mt[1] = angle; mt[2] = (float)l; mt[3] = 31 * c.sf[l];
    }
  }
}

static void orb_pyramid_and_scores(const OrbConsts &c, const OrbBuffers &b, int B, hipStream_t st) {
  static bool pattern_up = false;
  if (!pattern_up) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_orb_pattern), LF_ORB_PATTERN, 1024); pattern_up = true; }
  (void)hipMemsetAsync(b.ncand, 0, sizeof(int) * (size_t)B, st);
  (void)hipMemsetAsync(b.hist, 0, sizeof(int) * (size_t)B * LF_ORB_LEVELS * 256, st);
  hipLaunchKernelGGL(k_orb_level0, dim3(256, B), dim3(256), 0, st, c, b);
  for (int l = 1; l < LF_ORB_LEVELS; l++)
    hipLaunchKernelGGL(k_orb_resize, dim3((c.lw[l] * c.lh[l] + 255) / 256, B), dim3(256), 0, st, c, b, l);
  for (int l = 0; l < LF_ORB_LEVELS; l++) {
    hipLaunchKernelGGL(k_orb_blur, dim3((c.lw[l] + OB_TW - 1) / OB_TW, (c.lh[l] + OB_TH - 1) / OB_TH, B), dim3(256), 0, st, c, b, l);
    hipLaunchKernelGGL(k_orb_fast, dim3((c.lw[l] * c.lh[l] + 255) / 256, B), dim3(256), 0, st, c, b, l);
  }
}
static void orb_select_and_describe(const OrbConsts &c, const OrbBuffers &b, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_orb_select, dim3(B), dim3(OS_N), 0, st, c, b);
  hipLaunchKernelGGL(k_orb_describe, dim3(c.kp_cap < LF_ORB_KP_MAX ? c.kp_cap : LF_ORB_KP_MAX, B), dim3(64), 0, st, c, b);
}
void lf_orb_launch(const OrbConsts &c, const OrbBuffers &b, int B, hipStream_t st) {
  orb_pyramid_and_scores(c, b, B, st);
  for (int l = 0; l < LF_ORB_LEVELS; l++)
    hipLaunchKernelGGL(k_orb_nms<0>, dim3((c.lw[l] * c.lh[l] + 255) / 256, B), dim3(256), 0, st, c, b, l);
  orb_select_and_describe(c, b, B, st);
}
void lf_orb_launch_adjusted(const OrbConsts &c0, const OrbBuffers &b, const OrbAdjuster &a, int B, hipStream_t st) {
  OrbConsts c = c0;
  c.fast_threshold = a.base_threshold;                      // ONE score image, at the lowest threshold the adjuster can reach
  orb_pyramid_and_scores(c, b, B, st);
  for (int l = 0; l < LF_ORB_LEVELS; l++)                    // count: score histograms of the corners at the base threshold
    hipLaunchKernelGGL(k_orb_nms<1>, dim3((c.lw[l] * c.lh[l] + 255) / 256, B), dim3(256), 0, st, c, b, l);
  hipLaunchKernelGGL(k_orb_suffix, dim3((B * LF_ORB_LEVELS + 63) / 64), dim3(64), 0, st, c, b, B);
  hipLaunchKernelGGL(k_orb_adjust, dim3(1), dim3(64), 0, st, c, b, a, B);
  (void)hipMemsetAsync(b.hist, 0, sizeof(int) * (size_t)B * LF_ORB_LEVELS * 256, st);
  for (int l = 0; l < LF_ORB_LEVELS; l++)                    // the corners of every frame at ITS threshold
    hipLaunchKernelGGL(k_orb_nms<2>, dim3((c.lw[l] * c.lh[l] + 255) / 256, B), dim3(256), 0, st, c, b, l);
  orb_select_and_describe(c, b, B, st);
}
