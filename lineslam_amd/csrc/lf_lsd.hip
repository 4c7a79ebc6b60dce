// lf_lsd.hip -- LSD line-segment detection for gfx950 (MI355X), hand-written HIP.
//
// Replaces, bit for bit, the CPU path  callLsd (src/line/utils.cpp:112-135) ->
// lsd() -> LineSegmentDetection (external/lsd/lsd.cpp:1931-2065) for a BATCH of frames.
//
// Kernel map (reference lines each one follows):
//   k_gauss_x / k_gauss_y   gaussian_sampler          lsd.cpp:529-646   data-parallel, HBM-bound
//   k_ll_angle              ll_angle (gradient part)  lsd.cpp:717-770   data-parallel, HBM-bound
//   k_seed_hist/scan/scatter  ll_angle pseudo-ordering lsd.cpp:757-786  stable counting sort that
//                            reproduces the linked-list order (bin desc; x outer, y inner)
//   k_lsd_sweep             the seed loop             lsd.cpp:1996-2053 ONE WAVEFRONT PER FRAME:
//        region_grow :1610-1656, region2rect :1517-1604, get_theta :1474-1512, refine :1853-1921,
//        reduce_region_radius :1775-1841, rect_improve :1662-1768, rect_nfa :1388-1410 (+ the
//        rectangle iterator :1231-1383), nfa :980-1065.
//
// Why one wavefront per frame: the reference is order dependent (one shared `used` mask, seeds
// visited in pseudo-sorted order, reg_angle updated after every accepted pixel).  The sweep keeps
// that order exactly and uses the 64 lanes for everything that is order independent or can be
// speculated and committed in order: 64 seed probes at a time, 64 neighbour tests (7 region
// pixels x 9 neighbours) per region-growing step with first-hit commit, lane-parallel operand
// preparation for the order-dependent fp64 sums (summed serially through readlane in reference
// order), lane-parallel rectangle rasterisation (integer counts), lane-parallel binomial-tail
// terms.  Throughput comes from frames in flight (one per SIMD slot), see DESIGN.md section 4.
//
// All fp64 arithmetic is IEEE without contraction (-ffp-contract=off); the transcendental
// functions evaluated on the device come from lf_math.h.
#include "lf_lsd.h"
#define LF_SINCOS_DD_LANES   // double-double sin/cos: both series at once in two lanes (callers are wavefront-uniform)
#include "lf_math.h"
#include <float.h>

typedef unsigned long long u64;
typedef double lf_d2 __attribute__((ext_vector_type(2)));

// ----------------------------------------------------------------------------------------------
// small wave-level helpers (wave = 64 lanes on gfx950)
// work counters of the sweeps (regions grown, window steps, rectangle evaluations, pixels): for tools/lsd_perf.py and
// tools/lsd_mw_stats.py only -- build with LF_EXTRA_CFLAGS=-DLF_SWEEP_STATS=1; off, their scalar registers and adds are gone
// out-of-line phases of the sweep (build options for the register-pressure experiments of DESIGN.md section 4)
#ifndef LF_NI_IMPROVE
#define LF_NI_IMPROVE
#endif
#ifndef LF_NI_R2R
#define LF_NI_R2R
#endif
#ifndef LF_NI_GROW
#define LF_NI_GROW
#endif
#ifndef LF_SWEEP_STATS
#define LF_SWEEP_STATS 0
#endif
#if LF_SWEEP_STATS
#define LF_STAT(x) x
#else
#define LF_STAT(x) do { } while (0)
#endif
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ double rl64(double v, int l) {   // value of lane l (l wave-uniform)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rl32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}
// compiler-level ordering of this wave's own memory traffic (hardware keeps a wave's vector
// memory operations to one address in program order)
__device__ __forceinline__ void wave_mem_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// ----------------------------------------------------------------------------------------------
// gaussian_sampler (lsd.cpp:529-646).  Taps kx/ky and boundary-folded source indices jx/jy are
// host tables (gaussian_kernel :461-487 uses exp(); the symmetric boundary :597-600 is integer).
// Summation order i = 0..n-1 starting from 0.0, exactly as :590-603 / :622-635.
#define LF_GX_ROWS 8     // image rows per thread of k_gauss_x (the column's taps and source indices stay in registers)
#define LF_GX_TAPS 7     // 1 + 2 ceil(sigma sqrt(2 * 3 ln 10)) for scale 0.8 (sigma = 0.75); other widths take the generic loop
__global__ void __launch_bounds__(256) k_gauss_x(LsdConsts c, LsdBuffers b) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y0 = blockIdx.y * LF_GX_ROWS, f = blockIdx.z;
  if (x >= c.N) return;
  const uint8_t *img = b.gray + (size_t)f * b.gray_frame_stride;
  const double *k = b.kx + (size_t)x * c.ntaps;
  const int *j = b.jx + (size_t)x * c.ntaps;
  double *out = b.aux + ((size_t)f * c.H) * c.N + x;
  if (c.ntaps == LF_GX_TAPS) {
    double kk[LF_GX_TAPS];
    int jj[LF_GX_TAPS];
#pragma unroll
    for (int i = 0; i < LF_GX_TAPS; i++) { kk[i] = k[i]; jj[i] = j[i]; }
#pragma unroll
    for (int r = 0; r < LF_GX_ROWS; r++) {
      const int y = y0 + r;
      if (y < c.H) {
        const uint8_t *row = img + (size_t)y * b.gray_row_stride;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < LF_GX_TAPS; i++) sum += (double)row[jj[i]] * kk[i];   // u8 -> double: utils.cpp:124-129
        out[(size_t)y * c.N] = sum;
      }
    }
    return;
  }
  for (int r = 0; r < LF_GX_ROWS; r++) {
    const int y = y0 + r;
    if (y >= c.H) break;
    const uint8_t *row = img + (size_t)y * b.gray_row_stride;
    double sum = 0.0;
    for (int i = 0; i < c.ntaps; i++) sum += (double)row[j[i]] * k[i];
    out[(size_t)y * c.N] = sum;
  }
}
__global__ void __launch_bounds__(256) k_gauss_y(LsdConsts c, LsdBuffers b) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y, f = blockIdx.z;
  if (x >= c.N) return;
  const double *aux = b.aux + (size_t)f * c.H * c.N;
  const double *k = b.ky + (size_t)y * c.ntaps;
  const int *j = b.jy + (size_t)y * c.ntaps;
  double sum = 0.0;
  for (int i = 0; i < c.ntaps; i++) sum += aux[(size_t)j[i] * c.N + x] * k[i];
  b.scaled[((size_t)f * c.M + y) * c.N + x] = sum;
}

// ll_angle, gradient part (lsd.cpp:717-770).  One thread per pixel of the scaled image.
// 32 x 8 pixel tile per block.  The magnitude bins are stored TRANSPOSED (bins[x][y]): the seed sort walks the image
// column by column (x outer, y inner, lsd.cpp:723-724), so its reads become contiguous; the transpose goes through LDS.
__global__ void __launch_bounds__(256) k_ll_angle(LsdConsts c, LsdBuffers b) {
  __shared__ uint16_t tile[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8, f = blockIdx.z;
  const int x = x0 + tx, y = y0 + ty;
  const size_t NM = (size_t)c.N * c.M;
  uint16_t bin = LF_BIN_NONE;
  if (x < c.N && y < c.M) {
    const double *in = b.scaled + f * NM;
    size_t adr = (size_t)y * c.N + x;
    double ang = LF_NOTDEF, norm = 0.0, ca = 2.0, sa = 0.0;
    if (x < c.N - 1 && y < c.M - 1) {
      double com1 = in[adr + c.N + 1] - in[adr];
      double com2 = in[adr + 1] - in[adr + c.N];
      double gx = com1 + com2;
      double gy = com1 - com2;
      double norm2 = gx * gx + gy * gy;
      norm = lf_sqrt(norm2 / 4.0);
      if (!(norm <= c.rho)) {
        ang = lf_atan2(gx, -gy);
        lf_sincos(ang, &sa, &ca);      // cos/sin of the stored angle, as region_grow evaluates them (lsd.cpp:1652-1653)
        unsigned int i = (unsigned int)(norm * (double)c.n_bins / c.max_grad);
        if (i >= (unsigned int)c.n_bins) i = (unsigned int)c.n_bins - 1;
        bin = (uint16_t)i;
      }
    }
    b.angles[f * NM + adr] = ang;
    b.modgrad[f * NM + adr] = norm;
    *(lf_d2 *)&b.cossin[2 * (f * NM + adr)] = (lf_d2){ca, sa};
  }
  if (c.sweep_lu) {       // (only the opt-in LDS variant of the sweep reads it; not written, not allocated otherwise)
    // NOTDEF bitmap of the frame, rows padded to whole 32-bit words (k_lsd_sweep_lu's initial `unavailable` mask): a wavefront
    // covers two tile rows of 32 pixels -- the two halves of one ballot; pixels outside the image count as unavailable
    const u64 ndm = __ballot(bin == LF_BIN_NONE);
    const int wpr = (c.N + 31) >> 5;
    if ((threadIdx.x & 31) == 0 && y < c.M)
      b.ndbits[((size_t)f * c.M + y) * wpr + blockIdx.x] = (uint32_t)(ndm >> (threadIdx.x & 32));
  }
  tile[ty][tx] = bin;
  __syncthreads();
  const int cx = threadIdx.x >> 3, ry = threadIdx.x & 7;     // 8 consecutive rows of one column per 8 threads
  if (x0 + cx < c.N && y0 + ry < c.M) b.bins[f * NM + (size_t)(x0 + cx) * c.M + (y0 + ry)] = tile[ry][cx];
}

// ----------------------------------------------------------------------------------------------
// Pseudo-ordering of the seeds (lsd.cpp:757-786) as a stable counting sort.  The reference appends
// pixels to per-bin linked lists while scanning x outer / y inner and concatenates the lists from
// the highest non-empty bin down to bin 1.  Here: chunks of LF_SORT_CHUNK_COLS columns; per-chunk
// bin histograms; scan; in-order scatter.
__device__ __forceinline__ int sort_nchunks(const LsdConsts &c) {
  return (c.N - 1 + LF_SORT_CHUNK_COLS - 1) / LF_SORT_CHUNK_COLS;
}
__global__ void __launch_bounds__(256) k_seed_hist(LsdConsts c, LsdBuffers b) {
  extern __shared__ unsigned int hist[];   // n_bins
  int chunk = blockIdx.x, f = blockIdx.y;
  for (int i = threadIdx.x; i < c.n_bins; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const size_t NM = (size_t)c.N * c.M;
  const uint16_t *bins = b.bins + f * NM;
  int x0 = chunk * LF_SORT_CHUNK_COLS;
  int ncol = min(LF_SORT_CHUNK_COLS, c.N - 1 - x0);
  int rows = c.M - 1;
  for (int t = threadIdx.x; t < ncol * rows; t += blockDim.x) {
    int x = x0 + t / rows, y = t % rows;
    unsigned int bn = bins[(size_t)x * c.M + y];
    if (bn != LF_BIN_NONE) atomicAdd(&hist[bn], 1u);
  }
  __syncthreads();
  unsigned int *out = b.cnt + ((size_t)f * sort_nchunks(c) + chunk) * c.n_bins;
  for (int i = threadIdx.x; i < c.n_bins; i += blockDim.x) out[i] = hist[i];
}
// one block of n_bins (<= 1024) threads per frame
__global__ void __launch_bounds__(1024) k_seed_scan(LsdConsts c, LsdBuffers b) {
  __shared__ unsigned int tot[1024];
  __shared__ unsigned int suf[1024];
  __shared__ int top_s;
  int f = blockIdx.x, bin = threadIdx.x;
  int nch = sort_nchunks(c);
  unsigned int *cnt = b.cnt + (size_t)f * nch * c.n_bins;
  unsigned int run = 0;
  if (bin < c.n_bins)
    for (int ch = 0; ch < nch; ch++) {
      unsigned int t = cnt[(size_t)ch * c.n_bins + bin];
      cnt[(size_t)ch * c.n_bins + bin] = run;
      run += t;
    }
  tot[bin] = (bin < c.n_bins) ? run : 0;
  if (bin == 0) top_s = 0;
  __syncthreads();
  if (bin > 0 && bin < c.n_bins && tot[bin] != 0) atomicMax(&top_s, bin);
  __syncthreads();
  int top = top_s;   // highest non-empty bin >= 1, or 0                      (lsd.cpp:777)
  // included bins: top..1 (top > 0)  or  {0} (top == 0)                      (lsd.cpp:778-786)
  bool incl = (top > 0) ? (bin >= 1 && bin <= top) : (bin == 0);
  suf[bin] = incl ? tot[bin] : 0;
  __syncthreads();
  // inclusive suffix sum: suf[bin] = sum_{b' >= bin} included totals
  for (int o = 1; o < 1024; o <<= 1) {
    unsigned int v = (bin + o < 1024) ? suf[bin + o] : 0;
    __syncthreads();
    suf[bin] += v;
    __syncthreads();
  }
  unsigned int base = incl ? (suf[bin] - tot[bin]) : 0xFFFFFFFFu;   // seeds of higher bins come first
  if (bin < c.n_bins)
    for (int ch = 0; ch < nch; ch++) {
      size_t k = (size_t)ch * c.n_bins + bin;
      cnt[k] = incl ? (cnt[k] + base) : 0xFFFFFFFFu;
    }
  if (bin == 0) b.nseeds[f] = (int)suf[top > 0 ? 1 : 0];
}
// one wavefront per (chunk, frame): in-order scatter
__global__ void __launch_bounds__(64) k_seed_scatter(LsdConsts c, LsdBuffers b) {
  extern __shared__ unsigned int ctr[];   // n_bins running positions
  int chunk = blockIdx.x, f = blockIdx.y, lane = lane_id();
  const size_t NM = (size_t)c.N * c.M;
  const unsigned int *cnt = b.cnt + ((size_t)f * sort_nchunks(c) + chunk) * c.n_bins;
  for (int i = lane; i < c.n_bins; i += 64) ctr[i] = cnt[i];
  __syncthreads();
  const uint16_t *bins = b.bins + f * NM;
  uint32_t *seeds = b.seeds + f * NM;
  int x0 = chunk * LF_SORT_CHUNK_COLS;
  int ncol = min(LF_SORT_CHUNK_COLS, c.N - 1 - x0);
  int rows = c.M - 1, total = ncol * rows;
  for (int t0 = 0; t0 < total; t0 += 64) {
    int t = t0 + lane;
    bool v = t < total;
    int x = x0 + (v ? t / rows : 0), y = v ? t % rows : 0;
    unsigned int bn = v ? bins[(size_t)x * c.M + y] : LF_BIN_NONE;
    const bool has = v && bn != LF_BIN_NONE;
    // lanes of the same bin, found with one ballot per bin bit (n_bins <= 1024): a lane's position is its bin's
    // running counter plus its rank among the lanes of that bin in this group of 64 (list order = lane order)
    u64 same = __ballot(has);
#pragma unroll
    for (int bit = 0; bit < 10; bit++) {
      const u64 bm = __ballot((bn >> bit) & 1u);
      same &= ((bn >> bit) & 1u) ? bm : ~bm;
    }
    unsigned int pos = 0xFFFFFFFFu;
    unsigned int base = has ? ctr[bn] : 0xFFFFFFFFu;
    if (has && base != 0xFFFFFFFFu) pos = base + (unsigned int)__popcll(same & lanemask_lt());
    __syncthreads();   // single wave: every lane has read its counter before the bins' first lanes advance them
    if (has && base != 0xFFFFFFFFu && (same & lanemask_lt()) == 0ull) ctr[bn] = base + (unsigned int)__popcll(same);
    __syncthreads();
    if (pos != 0xFFFFFFFFu) seeds[pos] = (uint32_t)(y * c.N + x);
  }
}

// ----------------------------------------------------------------------------------------------
// The sweep.
struct Rect {   // lsd.cpp:1075-1084, plus the precision level (p = p0 / 2^plev) for the log tables
  double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
  int plev;
};
#define LF_TILE_W 8          // LU sweep: side of the (cos, sin) tile staged in LDS around a seed (64 pixels: one 16-byte DMA per lane)
#define LF_TILE_AT 3         // the seed sits at (3, 3) of its tile: offsets -3 .. +4
template <bool MW_, bool LU_ = false>
struct FrameViewT {
  static constexpr bool kMW = MW_;
  static constexpr bool kLU = LU_;                    // `used` lives in LDS as a bitmap, (cos, sin) tiles of the seeds are staged in LDS
  static constexpr int kRing = LU_ ? 512 : 1024;      // most recent region pixels kept in LDS (the growth front reads them back)
  int N, M, lane;
  const double *angles, *modgrad, *lgam, *cossin;
  const double *nfa_tab;  // tabulated nfa() for small n (null: always evaluate)
  const LsdConsts *dc;
  uint8_t *used;          // committed `used` mask of the frame
  uint32_t *ubits;        // LU: the frame's `unavailable` bitmap in LDS -- bit (x & 31) of word y * wpr + (x >> 5) is set when the pixel
  int wpr;                //     is used OR its angle is NOTDEF (region_grow rejects both the same way, lsd.cpp:1639-1641, 807)
  const lf_d2 *tile;      // LU: (cos, sin) of the LF_TILE_W x LF_TILE_W pixels from (tx0, ty0), staged in LDS by lu_stage_tile
  int tx0, ty0;
  uint8_t *tag;           // multi-wave sweep: this wavefront's PRIVATE marks (null: mark `used` directly)
  uint32_t *reg, *tmp;    // region list / scratch, `cap` entries each
  int cap;
  uint32_t *ring;         // LDS ring of the most recent region pixels (per wavefront)
  double *sums;           // LDS, 64 x 4 doubles per wavefront: operands of the order-dependent sums
  int *ring_ok;           // LDS flag: the ring holds the WHOLE current list (the last growth ended with <= LF_RING pixels)
  uint32_t *ever;         // multi-wave sweep: every pixel this region ever accepted (for validation)
  int ever_cap;
  int *n_ever, *overflow; // (wave-uniform values kept in memory visible to the helpers)
#ifdef LF_SWEEP_PROFILE
  u64 *gprof;             // [4] grow: windows, ticks waiting for the gathers, ticks deciding, hits
#endif
};
// `used` as the growing region sees it: committed marks plus its own tentative marks
typedef FrameViewT<false> FrameView;      // sequential sweep: marks go straight to `used`
typedef FrameViewT<true> FrameViewMW;     // multi-wave sweep: private tentative marks
typedef FrameViewT<false, true> FrameViewLU;   // sequential sweep with `used` and the seeds' neighbourhoods in LDS
template <class FV> __device__ __forceinline__ bool fv_is_used(const FV &f, int x, int y) {
  if constexpr (FV::kLU) return ((f.ubits[y * f.wpr + (x >> 5)] >> (x & 31)) & 1u) != 0u;
  else {
    const int p = y * f.N + x;
    unsigned char u = f.used[p];
    if constexpr (FV::kMW) u |= f.tag[p];
    return u != 0;
  }
}
template <class FV> __device__ __forceinline__ void fv_mark(const FV &f, int x, int y) {
  if constexpr (FV::kLU) atomicOr(&f.ubits[y * f.wpr + (x >> 5)], 1u << (x & 31));   // (lanes of one wavefront may share a word)
  else if constexpr (FV::kMW) f.tag[y * f.N + x] = 1;
  else f.used[y * f.N + x] = 1;
}
template <class FV> __device__ __forceinline__ void fv_unmark(const FV &f, int x, int y) {
  if constexpr (FV::kLU) atomicAnd(&f.ubits[y * f.wpr + (x >> 5)], ~(1u << (x & 31)));   // (region pixels are never NOTDEF: the bit was a mark)
  else if constexpr (FV::kMW) f.tag[y * f.N + x] = 0;
  else f.used[y * f.N + x] = 0;
}

// Entry i of the current region list (n entries): the LDS ring holds the most recent FV::kRing entries at [i mod kRing],
// i.e. the WHOLE list whenever n <= kRing -- nearly always -- so the passes over the list do not go to memory.
// (explicit address spaces: with generic pointers hipcc folds "ring or memory" into ONE flat_load of a selected address, which pays
// the memory path's latency for what is nearly always an LDS read)
typedef __attribute__((address_space(3))) uint32_t lf_lds_u32;
typedef __attribute__((address_space(1))) uint32_t lf_glb_u32;
template <class FV> __device__ __forceinline__ uint32_t fv_reg(const FV &f, int i, int n) {
  (void)n;
  if constexpr (!FV::kLU) return *f.ring_ok ? f.ring[i] : f.reg[i];   // (k_lsd_sweep: the branchy form below costs it 34 more spilled registers)
  if (*f.ring_ok) return ((const lf_lds_u32 *)f.ring)[i];      // (wave-uniform branch)
  return ((const lf_glb_u32 *)f.reg)[i];
}

// lsd.cpp:147-165
__device__ __forceinline__ bool d_double_equal(double a, double b) {
  if (a == b) return true;
  double abs_diff = lf_fabs(a - b), aa = lf_fabs(a), bb = lf_fabs(b);
  double abs_max = aa > bb ? aa : bb;
  if (abs_max < DBL_MIN) abs_max = DBL_MIN;
  return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);
}
__device__ __forceinline__ double d_dist(double x1, double y1, double x2, double y2) {   // :170-173
  return lf_sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
}
// lsd.cpp:799-832
__device__ __forceinline__ bool d_isaligned(double a, double theta, double prec) {
  if (a == LF_NOTDEF) return false;
  theta -= a;
  if (theta < 0.0) theta = -theta;
  if (theta > LF_M_3_2_PI) {
    theta -= LF_M_2__PI;
    if (theta < 0.0) theta = -theta;
  }
  return theta < prec;
}
__device__ __forceinline__ double d_angle_diff_signed(double a, double b) {   // :850-857
  a -= b;
  while (a <= -LF_PI) a += LF_M_2__PI;
  while (a > LF_PI) a -= LF_M_2__PI;
  return a;
}
__device__ __forceinline__ double d_angle_diff(double a, double b) {          // :837-845
  a = d_angle_diff_signed(a, b);
  if (a < 0.0) a = -a;
  return a;
}

// region_grow (lsd.cpp:1610-1656).  Slot s = 9*i + nb enumerates the reference's test order
// (region pixel i; neighbour nb = 3*(xx-x+1) + (yy-y+1), xx outer, yy inner).  Each step gathers the
// next 64 slots once and then decides them in order from registers: every slot before the first hit is
// a final "no"; the first hit is committed (used, reg[], sums, reg_angle) and the slots behind it are
// re-tested against the new sums, and so on until the window is exhausted -- exactly the sequential
// semantics, with one memory round trip per 64 slots.
// Alignment test of region_grow without atan2 on the critical path.  The reference decides
//   | atan2(sumdy, sumdx) - a | (wrapped, lsd.cpp:799-832) < prec.
// For 0 < prec < pi/2 that is  cos(angle between (sumdx,sumdy) and (cos a, sin a)) > cos(prec), i.e.
//   dot > 0  and  dot^2 > cos^2(prec) |S|^2     with dot = sumdx cos a + sumdy sin a.
// Both sides are evaluated in fp64 with relative error ~1e-15; whenever dot |dot| lies between (cos^2 prec - 1e-12) |S|^2
// and (cos^2 prec + 1e-12) |S|^2 the lane is AMBIGUOUS and the step falls back to the exact reference arithmetic (atan2 +
// isaligned), so every decision equals the reference's.  cos a / sin a come from k_ll_angle (the same
// lf_sincos values the reference's sums use, lsd.cpp:1652-1653).  prec outside (1e-6, 1.5) -- possible for
// the tolerance tau of refine() -- always takes the exact path (the wrap quirk of isaligned for angle
// differences in (pi, 3pi/2] matters once prec > pi/2).
// GIVEN: the caller hands over the seed's (cos, sin) -- the sweep gathers them for all 64 seeds of its window at once -- and the
// seed's angle is fetched only where it is used (the exact fall-back before the first accepted pixel; a region of the seed alone
// that the caller keeps: *reg_angle_io = -1e300): no dependent load in front of the growth.
template <class FV, bool GIVEN = false>
__device__ LF_NI_GROW int d_region_grow(const FV &f, int sx, int sy, double prec, double k_hi, double k_lo, double *reg_angle_io,
                             u64 *n_steps, double seed_cos = 0.0, double seed_sin = 0.0) {
  uint32_t *ring = f.ring;
  const int N = f.N, M = f.M, lane = f.lane;
  const int seed = sy * N + sx;
  const bool fast = (prec > 1e-6 && prec < 1.5);
  double reg_angle, sumdx, sumdy;
  bool seed_angle = false;                 // the region's angle is still angles[seed], not loaded yet (no pixel accepted)
  if constexpr (GIVEN) { sumdx = seed_cos; sumdy = seed_sin; reg_angle = 0.0; seed_angle = true; }
  else {
    reg_angle = f.angles[seed];
    sumdx = f.cossin[2 * seed]; sumdy = f.cossin[2 * seed + 1];
  }
  double S2 = sumdx * sumdx + sumdy * sumdy;
  bool angle_valid = !GIVEN;     // reg_angle == atan2(sumdy, sumdx) of the current sums (GIVEN: == angles[seed], pending)
  constexpr int LF_RING = FV::kRing;
  if (lane == 0) { uint32_t pk = (uint32_t)sx | ((uint32_t)sy << 16); f.reg[0] = pk; ring[0] = pk; fv_mark(f, sx, sy); }
  wave_mem_order();
  int size = 1, cur = 0;
  bool full = false;
  for (;;) {
    const int total = size * 9;
    if (cur >= total || full) break;
    LF_STAT((*n_steps)++);
    int slot = cur + lane;
    bool act = slot < total;
    int pi = slot / 9, nb = slot - pi * 9;
    uint32_t pk = 0u;
    if (act) pk = (pi + LF_RING >= size) ? ring[pi & (LF_RING - 1)] : f.reg[pi];
    int ox = nb / 3;
    int cx = (int)(pk & 0xffffu) + ox - 1, cy = (int)(pk >> 16) + (nb - ox * 3) - 1;
    bool inb = act && cx >= 0 && cy >= 0 && cx < N && cy < M;
    int ca = inb ? cy * N + cx : 0;
    bool u = fv_is_used(f, inb ? cx : 0, inb ? cy : 0);            // the gathers are issued together
    const lf_d2 csv = *(const lf_d2 *)&f.cossin[2 * ca];
    double cc = csv.x, ss = csv.y;
    bool cand = inb && !u && (cc <= 1.5);   // cos == 2 marks NOTDEF
#ifdef LF_SWEEP_PROFILE
    u64 tp0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    { int probe = __builtin_amdgcn_readfirstlane((int)cand + (int)(cc > 0.0)); asm volatile("" :: "s"(probe)); }
    u64 tp1 = __builtin_amdgcn_s_memtime();
#endif
    // The window (64 slots) is decided entirely from registers: after a hit is accepted, the slots behind
    // it are re-tested against the new sums (their pixel data cannot have changed except for the pixel just
    // accepted, which is masked out) -- one gather latency per window instead of one per accepted pixel.
    // No memory operation inside the decision loops: accepted lanes are collected in `cmask` and write
    // their own used / reg / ring entries when the window is finished.
    const int size0 = size;
    u64 cmask = 0;
    int lastL = -1;
    bool need_exact = !fast;
    if (fast) {
      // candidate lanes still to be decided, as a scalar mask; every lane keeps its own (cc, ss, ca)
      u64 pendm = __ballot(cand);
      // band of the pre-filter: (cos^2 prec -+ 1e-12) |S|^2; the signed square dot |dot| carries "dot > 0" with it (both band
      // edges are positive for prec < 1.5), so one product and two compares decide a lane
      double thr_hi = k_hi * S2, thr_lo = k_lo * S2;
      for (;;) {
        double dot = sumdx * cc + sumdy * ss;
        double lhs = dot * __builtin_fabs(dot);
        u64 ym = __builtin_amdgcn_ballot_w64(lhs > thr_hi) & pendm;      // aligned for certain
        u64 mm = __builtin_amdgcn_ballot_w64(!(lhs < thr_lo)) & pendm;   // aligned or inside the 1e-12 band
        if (mm != ym) { need_exact = true; break; }              // ambiguous lane: exact arithmetic below
        if (ym == 0) break;
        if constexpr (FV::kMW) { if (size >= f.cap) { *f.overflow = 1; full = true; break; } }   // speculative list full: caller re-runs at the frontier
        int L = __builtin_ctzll(ym);
        double cL = rl64(cc, L), sL = rl64(ss, L);
        // lanes up to L are decided; so is every lane that holds the accepted pixel (reached through another parent)
        pendm &= ~((2ull << L) - 1ull) & ~__builtin_amdgcn_ballot_w64(ca == rl32(ca, L));
        cmask |= 1ull << L;
        size++;
        sumdx += cL;
        sumdy += sL;
        asm volatile("" : "+v"(sumdx), "+v"(sumdy));   // keep the running sums in vector registers (no SGPR round trip)
        S2 = sumdx * sumdx + sumdy * sumdy;
        thr_hi = k_hi * S2;
        thr_lo = k_lo * S2;
        angle_valid = false; seed_angle = false;
        lastL = L;
      }
      cand = cand && ((pendm >> lane) & 1ull);   // (for the exact path below: lanes already decided stay decided)
    }
    if (need_exact && !full) {   // the reference's own arithmetic for the rest of the window
      double a_exact = cand ? f.angles[ca] : LF_NOTDEF;
      for (;;) {
        if (!angle_valid) { reg_angle = seed_angle ? f.angles[seed] : lf_atan2(sumdy, sumdx); angle_valid = true; }
        bool ok = cand && lane > lastL && d_isaligned(a_exact, reg_angle, prec);
        u64 mask = __ballot(ok);
        if (mask == 0) break;
        if constexpr (FV::kMW) { if (size >= f.cap) { *f.overflow = 1; full = true; break; } }
        int L = __builtin_ctzll(mask);
        double cL = rl64(cc, L), sL = rl64(ss, L);
        if (ca == rl32(ca, L)) cand = false;
        cmask |= 1ull << L;
        size++;
        sumdx += cL;
        sumdy += sL;
        S2 = sumdx * sumdx + sumdy * sumdy;
        angle_valid = false; seed_angle = false;
        lastL = L;
      }
    }
#ifdef LF_SWEEP_PROFILE
    { u64 tp2 = __builtin_amdgcn_s_memtime(); f.gprof[0]++; f.gprof[1] += tp1 - tp0; f.gprof[2] += tp2 - tp1; f.gprof[3] += (u64)__popcll(cmask); }
#endif
    if ((cmask >> lane) & 1ull) {
      int at = size0 + __popcll(cmask & lanemask_lt());
      uint32_t npk = (uint32_t)cx | ((uint32_t)cy << 16);
      fv_mark(f, cx, cy); f.reg[at] = npk; ring[at & (LF_RING - 1)] = npk;
    }
    wave_mem_order();
    cur += min(64, total - cur);
  }
  if (!angle_valid) reg_angle = seed_angle ? -1e300 : lf_atan2(sumdy, sumdx);   // value after the last accepted pixel (lsd.cpp:1654); GIVEN, seed alone: the caller loads angles[seed] if it keeps the region
  *reg_angle_io = reg_angle;
  if (lane == 0) *f.ring_ok = (size <= LF_RING) ? 1 : 0;   // (a list that shrinks later stays complete: fills are mirrored)
  wave_mem_order();
  return size;
}

// ---- region_grow with the frame's `used` mask and the seed's neighbourhood in LDS (FrameViewLU) -------------------------------
// Same slot enumeration, same window decisions, same arithmetic in the same order as d_region_grow above; what changes is where
// the operands come from:
//   * `used` is a bitmap in LDS (24.5 KB for 512 x 384) that also carries NOTDEF -- region_grow rejects a used pixel
//     (lsd.cpp:1639) and an undefined one (isaligned, :807) alike, and nothing else reads `used` -- so a window first reads its
//     bits (one ds_read) and gathers (cos, sin) for the CANDIDATE lanes only;
//   * (cos, sin) of the 8 x 8 pixels around the seed were staged in LDS by a DMA issued one region EARLIER (lu_stage_tile,
//     global_load_lds: 64 lanes x 16 bytes, no registers): the small regions -- 95 % of all growths, 2.5 windows each -- decide
//     entirely from LDS; a lane whose pixel lies outside the tile gathers it from memory as before;
//   * the seed's own (cos, sin) comes from the tile, and its angle is only fetched in the one case that uses it (the exact
//     fall-back before the first accepted pixel): the dependent load in front of every growth is gone.
// `cossin` is immutable, so a staged tile can never be stale; `used` is read from the authoritative LDS bitmap at use time.
__device__ __forceinline__ void lu_stage_tile(const double *cossin, lf_d2 *slot, int N, int M, int tx0, int ty0, int lane) {
  const int x = tx0 + (lane & (LF_TILE_W - 1)), y = ty0 + (lane >> 3);
  if (x >= 0 && y >= 0 && x < N && y < M) {
    // global_load_lds_dwordx4: LDS destination = M0 (the slot's byte offset in LDS) + lane * 16, i.e. entry ly * 8 + lx of the
    // slot.  Issued as inline assembly ON PURPOSE: with the builtin, hipcc orders every later LDS access and every fence of
    // the wavefront behind the DMA (s_waitcnt vmcnt(0) in front of each ds_read / ds_or), which makes the staging synchronous;
    // here the ONE consumer waits itself (lu_wait_staged) and nothing else touches the slot while the DMA is in flight.
    // An instruction the compiler does not count only makes its own vmcnt(N) waits more conservative (memory operations of a
    // wavefront complete in order), never too short.
    const double *g = cossin + 2 * ((size_t)y * N + x);
    const unsigned lds_off = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)slot);   // low half of the flat address of a __shared__ object = its LDS offset (wave-uniform)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_off) : "memory", "m0");
  }
}
__device__ __forceinline__ void lu_wait_staged() {      // the wavefront's own DMAs have landed (nothing else orders a ds_read behind them)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <class FV>
__device__ LF_NI_GROW int d_region_grow_lu(const FV &f, int sx, int sy, double prec, double k_hi, double k_lo, double *reg_angle_io,
                                           u64 *n_steps) {
  static_assert(FV::kLU && !FV::kMW, "LDS-resident sweep only");
  constexpr int RING = FV::kRing;
  uint32_t *ring = f.ring;
  const int N = f.N, M = f.M, lane = f.lane;
  const int seed = sy * N + sx;
  const bool fast = (prec > 1e-6 && prec < 1.5);
  const int tx0 = f.tx0, ty0 = f.ty0;
  double reg_angle = 0.0;
  bool angle_valid = false, seed_angle = true;     // seed_angle: the region's angle is still angles[seed] (no pixel accepted yet)
  double sumdx, sumdy;
  {
    const lf_d2 sv = f.tile[(sy - ty0) * LF_TILE_W + (sx - tx0)];     // the seed is always inside its own tile
    sumdx = sv.x; sumdy = sv.y;
  }
  double S2 = sumdx * sumdx + sumdy * sumdy;
  // The region's pixel list lives in the LDS ring; the frame's global list f.reg is written only from the window in which the
  // region outgrows the ring (first the ring's content, then every new pixel) -- a region of <= kRing pixels, i.e. nearly
  // every one, never stores to memory while it grows (fv_reg / the label pass read the ring while *f.ring_ok).
  if (lane == 0) { ring[0] = (uint32_t)sx | ((uint32_t)sy << 16); fv_mark(f, sx, sy); }
  wave_mem_order();
  int size = 1, cur = 0;
  for (;;) {
    const int total = size * 9;
    if (cur >= total) break;
    LF_STAT((*n_steps)++);
    int slot = cur + lane;
    bool act = slot < total;
    int pi = slot / 9, nb = slot - pi * 9;
    uint32_t pk = ((const lf_lds_u32 *)ring)[pi & (RING - 1)];      // (inactive lanes read some slot of the ring: harmless)
    if (size > RING) { if (act && pi + RING < size) pk = ((const lf_glb_u32 *)f.reg)[pi]; }   // wave-uniform guard: older entries of a big region
    int ox = nb / 3;
    int cx = (int)(pk & 0xffffu) + ox - 1, cy = (int)(pk >> 16) + (nb - ox * 3) - 1;
    bool inb = act && cx >= 0 && cy >= 0 && cx < N && cy < M;
    int ca = inb ? cy * N + cx : 0;
    // the bitmap word and the tile entry are read together (both addresses depend on (cx, cy) only); a candidate outside the
    // tile gathers its (cos, sin) from memory
    const unsigned lx = (unsigned)(cx - tx0), ly = (unsigned)(cy - ty0);
    const bool intile = inb && lx < (unsigned)LF_TILE_W && ly < (unsigned)LF_TILE_W;
    lf_d2 csv = f.tile[intile ? ly * LF_TILE_W + lx : 0];
    bool cand = inb && !fv_is_used(f, inb ? cx : 0, inb ? cy : 0);
    const bool far = cand && !intile;
    if (__builtin_amdgcn_ballot_w64(far) != 0ull) { if (far) csv = *(const lf_d2 *)&f.cossin[2 * ca]; }
    double cc = cand ? csv.x : 2.0, ss = csv.y;
    cand = cand && (cc <= 1.5);             // (cos == 2 marks NOTDEF: already folded into the bitmap, kept as a guard)
#ifdef LF_SWEEP_PROFILE
    u64 tp0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    { int probe = __builtin_amdgcn_readfirstlane((int)cand + (int)(cc > 0.0)); asm volatile("" :: "s"(probe)); }
    u64 tp1 = __builtin_amdgcn_s_memtime();
#endif
    const int size0 = size;
    u64 cmask = 0;
    int lastL = -1;
    bool need_exact = !fast;
    if (fast) {
      u64 pendm = __ballot(cand);
      double thr_hi = k_hi * S2, thr_lo = k_lo * S2;
      for (;;) {
        double dot = sumdx * cc + sumdy * ss;
        double lhs = dot * __builtin_fabs(dot);
        u64 ym = __builtin_amdgcn_ballot_w64(lhs > thr_hi) & pendm;
        u64 mm = __builtin_amdgcn_ballot_w64(!(lhs < thr_lo)) & pendm;
        if (mm != ym) { need_exact = true; break; }
        if (ym == 0) break;
        int L = __builtin_ctzll(ym);
        double cL = rl64(cc, L), sL = rl64(ss, L);
        pendm &= ~((2ull << L) - 1ull) & ~__builtin_amdgcn_ballot_w64(ca == rl32(ca, L));
        cmask |= 1ull << L;
        size++;
        sumdx += cL;
        sumdy += sL;
        asm volatile("" : "+v"(sumdx), "+v"(sumdy));
        S2 = sumdx * sumdx + sumdy * sumdy;
        thr_hi = k_hi * S2;
        thr_lo = k_lo * S2;
        angle_valid = false; seed_angle = false;
        lastL = L;
      }
      cand = cand && ((pendm >> lane) & 1ull);
    }
    if (need_exact) {   // the reference's own arithmetic for the rest of the window
      double a_exact = cand ? f.angles[ca] : LF_NOTDEF;
      for (;;) {
        if (!angle_valid) { reg_angle = seed_angle ? f.angles[seed] : lf_atan2(sumdy, sumdx); angle_valid = true; }
        bool ok = cand && lane > lastL && d_isaligned(a_exact, reg_angle, prec);
        u64 mask = __ballot(ok);
        if (mask == 0) break;
        int L = __builtin_ctzll(mask);
        double cL = rl64(cc, L), sL = rl64(ss, L);
        if (ca == rl32(ca, L)) cand = false;
        cmask |= 1ull << L;
        size++;
        sumdx += cL;
        sumdy += sL;
        S2 = sumdx * sumdx + sumdy * sumdy;
        angle_valid = false; seed_angle = false;
        lastL = L;
      }
    }
#ifdef LF_SWEEP_PROFILE
    { u64 tp2 = __builtin_amdgcn_s_memtime(); f.gprof[0]++; f.gprof[1] += tp1 - tp0; f.gprof[2] += tp2 - tp1; f.gprof[3] += (u64)__popcll(cmask); }
#endif
    if (size > RING && size0 <= RING) {       // the region outgrows the ring in this window: the list so far goes to memory first
      for (int i = lane; i < size0; i += 64) f.reg[i] = ring[i];
      wave_mem_order();
    }
    if ((cmask >> lane) & 1ull) {
      int at = size0 + __popcll(cmask & lanemask_lt());
      uint32_t npk = (uint32_t)cx | ((uint32_t)cy << 16);
      fv_mark(f, cx, cy); ring[at & (RING - 1)] = npk;
      if (size > RING) f.reg[at] = npk;
    }
    wave_mem_order();
    cur += min(64, total - cur);
  }
  // value after the last accepted pixel (lsd.cpp:1654); a region of the seed alone keeps the seed's stored angle -- its callers
  // drop such a region (size < min_reg_size / size < 2) unless min_reg_size <= 1, so that load is left to them (*reg_angle_io
  // is then -1e300: d_region_angle_of_seed)
  if (!angle_valid) reg_angle = seed_angle ? -1e300 : lf_atan2(sumdy, sumdx);
  *reg_angle_io = reg_angle;
  if (lane == 0) *f.ring_ok = (size <= RING) ? 1 : 0;
  wave_mem_order();
  return size;
}

// Three running sums over the next `cnt` (<= 64) region pixels in list order: sa += a_j, sb += b_j, sc -+= c_j for
// j = 0 .. cnt-1, where lane j holds (a_j, b_j, c_j).  The lanes publish their operands in LDS and every lane adds
// them one by one (broadcast reads, four pixels per trip with the loads issued first) -- 5 instructions per pixel
// instead of 6 readlanes + 3 additions.
template <bool SUBC, class FV>
__device__ __forceinline__ void d_ordered_sum3(const FV &f, double a, double b, double c, int cnt, double &sa, double &sb,
                                               double &sc) {
  double *L = f.sums;
  *(lf_d2 *)&L[4 * f.lane] = (lf_d2){a, b};
  L[4 * f.lane + 2] = c;
  wave_mem_order();
  // lanes >= cnt published zeros (their pixel weight is 0), so the loop runs in full groups of eight: adding +0.0
  // terms changes nothing and there is no tail
  const int cnt8 = (cnt + 7) & ~7;
  for (int j = 0; j < cnt8; j += 8) {
    lf_d2 p[8];
    double c8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { p[k] = *(const lf_d2 *)&L[4 * (j + k)]; c8[k] = L[4 * (j + k) + 2]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { sa += p[k].x; sb += p[k].y; sc = SUBC ? sc - c8[k] : sc + c8[k]; }
  }
  wave_mem_order();
}

// region2rect + get_theta (lsd.cpp:1517-1604, 1474-1512).  The three weighted sums and the three
// inertia sums are accumulated in the reference's pixel order: lanes prepare the 64 next operands,
// a uniform loop adds them one by one.
// Rectangle angle and axis of region2rect (lsd.cpp:1517-1545): correctly rounded atan2 / sin / cos -- the rectangle's end
// pixel lies exactly on its end edge, so this is the one place where LSD's output depends on the last bit of libm
// (DESIGN.md section 3).  Kept OUT OF LINE: its double-double temporaries would otherwise set the register
// allocation of the whole sweep kernel; it runs once per rectangle.
__device__ __noinline__ void d_rect_theta(double Ixx, double Iyy, double Ixy, double lambda, double reg_angle, double prec,
                                          double *theta_out, double *dx_out, double *dy_out) {
  double t0;
  lf_dd s0, c0;
  const bool xx = lf_fabs(Ixx) > lf_fabs(Iyy);
  const double th = lf_atan2_cr_sc(xx ? lambda - Ixx : Ixy, xx ? Ixy : lambda - Iyy, &t0, &s0, &c0);
  double theta = th;
  int flipped = 0;
  if (d_angle_diff(theta, reg_angle) > prec) { theta += LF_PI; flipped = 1; }
  double dy, dx;
  lf_sincos_cr_near(theta, flipped, th, t0, s0, c0, &dy, &dx);   // one double-double sin/cos evaluation serves both
  *theta_out = theta; *dx_out = dx; *dy_out = dy;
}
template <class FV>
__device__ LF_NI_R2R void d_region2rect(const FV &f, int n, double reg_angle, double prec, double p,
                              int plev, Rect *rec) {
  const int N = f.N, lane = f.lane;
  double x = 0.0, y = 0.0, sum = 0.0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool v = i < n;
    uint32_t pk = v ? fv_reg(f, i, n) : 0u;
    int rx = (int)(pk & 0xffffu), ry = (int)(pk >> 16);
    double w = v ? f.modgrad[ry * N + rx] : 0.0;
    double xw = (double)rx * w, yw = (double)ry * w;
    d_ordered_sum3<false>(f, xw, yw, w, min(64, n - base), x, y, sum);
  }
  x /= sum;
  y /= sum;
  double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool v = i < n;
    uint32_t pk = v ? fv_reg(f, i, n) : 0u;
    int rx = (int)(pk & 0xffffu), ry = (int)(pk >> 16);
    double w = v ? f.modgrad[ry * N + rx] : 0.0;
    double ey = (double)ry - y, ex = (double)rx - x;
    double txx = ey * ey * w, tyy = ex * ex * w, txy = ex * ey * w;
    d_ordered_sum3<true>(f, txx, tyy, txy, min(64, n - base), Ixx, Iyy, Ixy);
  }
  double lambda = 0.5 * (Ixx + Iyy - lf_sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
  double theta, dy, dx;
  d_rect_theta(Ixx, Iyy, Ixy, lambda, reg_angle, prec, &theta, &dx, &dy);
  // extents: max/min over the region including 0 (lsd.cpp:1571-1580); order independent
  double l_min = 0.0, l_max = 0.0, w_min = 0.0, w_max = 0.0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    if (i < n) {
      uint32_t pk = fv_reg(f, i, n);
      double ex = (double)(int)(pk & 0xffffu) - x, ey = (double)(int)(pk >> 16) - y;
      double l = ex * dx + ey * dy;
      double w = -ex * dy + ey * dx;
      if (l > l_max) l_max = l;
      if (l < l_min) l_min = l;
      if (w > w_max) w_max = w;
      if (w < w_min) w_min = w;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {   // the four reductions step together: one shuffle round trip per step, not four
    double a = __shfl_xor(l_max, o, 64), b2 = __shfl_xor(l_min, o, 64), c2 = __shfl_xor(w_max, o, 64), d2 = __shfl_xor(w_min, o, 64);
    l_max = a > l_max ? a : l_max;
    l_min = b2 < l_min ? b2 : l_min;
    w_max = c2 > w_max ? c2 : w_max;
    w_min = d2 < w_min ? d2 : w_min;
  }
  rec->x1 = x + l_min * dx;
  rec->y1 = y + l_min * dy;
  rec->x2 = x + l_max * dx;
  rec->y2 = y + l_max * dy;
  rec->width = w_max - w_min;
  rec->x = x;
  rec->y = y;
  rec->theta = theta;
  rec->dx = dx;
  rec->dy = dy;
  rec->prec = prec;
  rec->p = p;
  rec->plev = plev;
  if (rec->width < 1.0) rec->width = 1.0;
}

// nfa (lsd.cpp:980-1065).  log_gamma comes from the host table lgam[i] = log_gamma((double)i);
// log(p), log(1-p), log10(p) from the per-level host tables.  The binomial tail is summed in the
// reference order; the per-term truncation test (pow, log10) is evaluated for a chunk of terms in
// parallel lanes and the first term that satisfies it ends the sum, as the sequential `break`.
template <class FV>
__device__ double d_nfa(const FV &f, int n, int k, double p, int plev, double logNT) {
  const int lane = f.lane;
  const double tolerance = 0.1;
  if (n == 0 || k == 0) return -logNT;
  if (n < LF_NFA_TAB_N && f.nfa_tab) return f.nfa_tab[(size_t)plev * LF_NFA_TAB_TRI + (size_t)(n * (n + 1) / 2 + k)];
  if (n == k) return -logNT - (double)n * f.dc->log10p[plev];
  double p_term = p / (1.0 - p);
  double log1term = f.lgam[n + 1] - f.lgam[k + 1] - f.lgam[n - k + 1] + (double)k * f.dc->logp[plev] +
                    (double)(n - k) * f.dc->log1mp[plev];
  double term = lf_exp(log1term);
  if (d_double_equal(term, 0.0)) {
    if ((double)k > (double)n * p) return -log1term / LF_LN10 - logNT;
    else return -logNT;
  }
  double bin_tail = term;
  int i0 = k + 1, chunk = 8;
  while (i0 <= n) {
    int cnt = min(chunk, n - i0 + 1);
    int i = i0 + lane;
    bool v = lane < cnt;
    double bin_term = v ? (double)(n - i + 1) * (1.0 / (double)i) : 0.0;
    double mult_term = bin_term * p_term;
    double my_term = 0.0, my_tail = 0.0;
    for (int j = 0; j < cnt; j++) {
      term *= rl64(mult_term, j);
      bin_tail += term;
      if (lane == j) { my_term = term; my_tail = bin_tail; }
    }
    bool stop = false;
    if (v && bin_term < 1.0) {
      double err = my_term * ((1.0 - lf_pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      stop = err < tolerance * lf_fabs(-lf_log10(my_tail) - logNT) * my_tail;
    }
    u64 m = __ballot(stop);
    if (m) { bin_tail = rl64(my_tail, __builtin_ctzll(m)); break; }
    i0 += cnt;
    if (chunk < 64) chunk <<= 1;
  }
  return -lf_log10(bin_tail) - logNT;
}

// nfa (lsd.cpp:980-1065) evaluated sequentially by one thread -- the same arithmetic as d_nfa, term by
// term -- for every (level, n, k) with n < LF_NFA_TAB_N: nfa depends on nothing else (p = c.p / 2^level,
// logNT fixed by the image size), and most rectangles rect_improve tests are this small.
__global__ void __launch_bounds__(256) k_nfa_table(LsdConsts c, const LsdConsts *dc, LsdBuffers b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, plev = blockIdx.y;
  if (t >= LF_NFA_TAB_TRI) return;
  int n = (int)((lf_sqrt(8.0 * (double)t + 1.0) - 1.0) / 2.0);
  while ((n + 1) * (n + 2) / 2 <= t) ++n;
  while (n * (n + 1) / 2 > t) --n;
  const int k = t - n * (n + 1) / 2;
  const double tolerance = 0.1, logNT = c.logNT;
  double p = c.p;
  for (int i = 0; i < plev; i++) p /= 2.0;
  double *out = b.nfa_tab + (size_t)plev * LF_NFA_TAB_TRI + t;
  if (n == 0 || k == 0) { *out = -logNT; return; }
  if (n == k) { *out = -logNT - (double)n * dc->log10p[plev]; return; }
  double p_term = p / (1.0 - p);
  double log1term = b.lgam[n + 1] - b.lgam[k + 1] - b.lgam[n - k + 1] + (double)k * dc->logp[plev] +
                    (double)(n - k) * dc->log1mp[plev];
  double term = lf_exp(log1term);
  if (d_double_equal(term, 0.0)) {
    if ((double)k > (double)n * p) *out = -log1term / LF_LN10 - logNT;
    else *out = -logNT;
    return;
  }
  double bin_tail = term;
  for (int i = k + 1; i <= n; i++) {
    double bin_term = (double)(n - i + 1) * (1.0 / (double)i);
    double mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      double err = term * ((1.0 - lf_pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * lf_fabs(-lf_log10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  *out = -lf_log10(bin_tail) - logNT;
}

// lsd.cpp:1183-1215
__device__ __forceinline__ double d_inter_low(double x, double x1, double y1, double x2, double y2) {
  if (d_double_equal(x1, x2) && y1 < y2) return y1;
  if (d_double_equal(x1, x2) && y1 > y2) return y2;
  return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}
__device__ __forceinline__ double d_inter_hi(double x, double x1, double y1, double x2, double y2) {
  if (d_double_equal(x1, x2) && y1 < y2) return y2;
  if (d_double_equal(x1, x2) && y1 > y2) return y1;
  return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}
__device__ __forceinline__ double pick4(int i, double a0, double a1, double a2, double a3) {
  i &= 3;
  return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
}

// rect_nfa (lsd.cpp:1388-1410) with the rectangle iterator (ri_ini :1317-1383, ri_inc :1247-1310,
// ri_end :1231-1240) turned into a rasteriser: column x = ceil(vx[0]).. while x <= vx[2]; inside a
// column y = ceil(ys).. while y <= ye.  Only in-image pixels are counted by the reference, so both
// ranges are clipped to the image before any int conversion.  Lanes take (column, row-phase) pairs;
// the two counters are integers, so their reduction order is irrelevant.
// The two counters are taken with ballots (every pass of the row loop tests one pixel per lane), so they live in
// scalar registers and need no wavefront reduction.  NP = 1: the rectangle's own precision.  NP = 5: the five HALVED
// precisions p/2 .. p/32 of the "finer precision" loops of rect_improve, which look at the same pixels: one pass
// over the rectangle serves all five.
template <int NP, class FV>
__device__ void d_rect_count(const FV &f, const Rect &r, int *pts_out, int *alg_out) {
  const int N = f.N, M = f.M, lane = f.lane;
  double precs[NP];
  if constexpr (NP == 1) precs[0] = r.prec;
  else { double pk = r.p; for (int k = 0; k < NP; k++) { pk /= 2.0; precs[k] = pk * LF_PI; } }   // r.p /= 2; r.prec = r.p * M_PI (lsd.cpp:1677-1678)
  double hw = r.width / 2.0;
  double rx0 = r.x1 - r.dy * hw, ry0 = r.y1 + r.dx * hw;
  double rx1 = r.x2 - r.dy * hw, ry1 = r.y2 + r.dx * hw;
  double rx2 = r.x2 + r.dy * hw, ry2 = r.y2 - r.dx * hw;
  double rx3 = r.x1 + r.dy * hw, ry3 = r.y1 - r.dx * hw;
  int offset;
  if (r.x1 < r.x2 && r.y1 <= r.y2) offset = 0;
  else if (r.x1 >= r.x2 && r.y1 < r.y2) offset = 1;
  else if (r.x1 > r.x2 && r.y1 >= r.y2) offset = 2;
  else offset = 3;
  double vx0 = pick4(offset, rx0, rx1, rx2, rx3), vy0 = pick4(offset, ry0, ry1, ry2, ry3);
  double vx1 = pick4(offset + 1, rx0, rx1, rx2, rx3), vy1 = pick4(offset + 1, ry0, ry1, ry2, ry3);
  double vx2 = pick4(offset + 2, rx0, rx1, rx2, rx3), vy2 = pick4(offset + 2, ry0, ry1, ry2, ry3);
  double vx3 = pick4(offset + 3, rx0, rx1, rx2, rx3), vy3 = pick4(offset + 3, ry0, ry1, ry2, ry3);
  int pts = 0, alg[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) alg[k] = 0;
  double xlo_d = __builtin_ceil(vx0), xhi_d = __builtin_floor(vx2);
  if (xlo_d < 0.0) xlo_d = 0.0;
  if (xhi_d > (double)(N - 1)) xhi_d = (double)(N - 1);
  if (xlo_d <= xhi_d) {
    int xlo = (int)xlo_d, ncols = (int)xhi_d - xlo + 1;
    // lanes per column: 64 / (smallest power of two >= ncols), at least 1
    int cpp = 1;                       // columns per pass
    while (cpp < ncols && cpp < 64) cpp <<= 1;
    int lpc = 64 / cpp;                // lanes (row phases) per column
    int col = lane & (cpp - 1), phase = lane / cpp;
    for (int cb = 0; cb < ncols; cb += cpp) {
      int ci = cb + col;
      int x = xlo + ci, y = 0, yhi = -1;
      if (ci < ncols) {
        double xd = (double)x, ys, ye;
        if (xd < vx3) ys = d_inter_low(xd, vx0, vy0, vx3, vy3);
        else ys = d_inter_low(xd, vx3, vy3, vx2, vy2);
        if (xd < vx1) ye = d_inter_hi(xd, vx0, vy0, vx1, vy1);
        else ye = d_inter_hi(xd, vx1, vy1, vx2, vy2);
        double ylo_d = __builtin_ceil(ys), yhi_d = __builtin_floor(ye);
        if (ylo_d < 0.0) ylo_d = 0.0;
        if (yhi_d > (double)(M - 1)) yhi_d = (double)(M - 1);
        if (ylo_d <= yhi_d) { yhi = (int)yhi_d; y = (int)ylo_d + phase; }
      }
      for (;;) {   // one pixel per lane and trip
        const bool v = y <= yhi;
        const u64 vm = __builtin_amdgcn_ballot_w64(v);
        if (vm == 0) break;
        pts += __popcll(vm);
        double th = 1e300;
        if (v) {
          double a = f.angles[y * N + x];
          if (a != LF_NOTDEF) {   // isaligned (lsd.cpp:799-832): one wrapped difference, NP thresholds
            th = r.theta - a;
            if (th < 0.0) th = -th;
            if (th > LF_M_3_2_PI) { th -= LF_M_2__PI; if (th < 0.0) th = -th; }
          }
        }
#pragma unroll
        for (int k = 0; k < NP; k++) alg[k] += __popcll(__builtin_amdgcn_ballot_w64(th < precs[k]));
        y += lpc;
      }
    }
  }
  *pts_out = pts;
#pragma unroll
  for (int k = 0; k < NP; k++) alg_out[k] = alg[k];
}
template <class FV>
__device__ double d_rect_nfa(const FV &f, const Rect &r, double logNT, u64 *n_px) {
  int pts, alg;
  d_rect_count<1>(f, r, &pts, &alg);
  LF_STAT(*n_px += (u64)pts);
  return d_nfa(f, pts, alg, r.p, r.plev, logNT);
}
// rect_nfa of the five rectangles r with p/2, p/4 .. p/32 (same geometry): one counting pass, the nfa values of small
// rectangles looked up by five lanes at once
template <class FV>
__device__ void d_rect_nfa_finer5(const FV &f, const Rect &r, double logNT, u64 *n_px, double *out) {
  int pts, alg[5];
  d_rect_count<5>(f, r, &pts, alg);
  LF_STAT(*n_px += 5ull * (u64)pts);
  if (pts < LF_NFA_TAB_N && f.nfa_tab && pts > 0) {
    const int k = f.lane < 5 ? f.lane : 4;
    int a = k == 0 ? alg[0] : (k == 1 ? alg[1] : (k == 2 ? alg[2] : (k == 3 ? alg[3] : alg[4])));
    double v = (a == 0) ? -logNT : f.nfa_tab[(size_t)(r.plev + 1 + k) * LF_NFA_TAB_TRI + (size_t)(pts * (pts + 1) / 2 + a)];
#pragma unroll
    for (int q = 0; q < 5; q++) out[q] = rl64(v, q);
  } else {
    double pk = r.p;
    for (int q = 0; q < 5; q++) { pk /= 2.0; out[q] = d_nfa(f, pts, alg[q], pk, r.plev + 1 + q, logNT); }
  }
}

// rect_improve (lsd.cpp:1662-1768)
template <class FV>
__device__ LF_NI_IMPROVE double d_rect_improve(const FV &f, Rect *rec, double logNT, double eps, u64 *n_nfa,
                                 u64 *n_px) {
  Rect r;
  const double delta = 0.5, delta_2 = delta / 2.0;
  double log_nfa, log_nfa_new;
  LF_STAT(++*n_nfa);
  log_nfa = d_rect_nfa(f, *rec, logNT, n_px);
  if (log_nfa > eps) return log_nfa;
  r = *rec;
  {
    double lv[5];
    d_rect_nfa_finer5(f, r, logNT, n_px, lv);
    for (int n = 0; n < 5; n++) {   // finer precisions
      r.p /= 2.0; r.plev++;
      r.prec = r.p * LF_PI;
      LF_STAT(++*n_nfa);
      log_nfa_new = lv[n];
      if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; *rec = r; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec;
  for (int n = 0; n < 5; n++) {   // reduce width
    if ((r.width - delta) >= 0.5) {
      r.width -= delta;
      LF_STAT(++*n_nfa);
      log_nfa_new = d_rect_nfa(f, r, logNT, n_px);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec;
  for (int n = 0; n < 5; n++) {   // reduce one side
    if ((r.width - delta) >= 0.5) {
      r.x1 += -r.dy * delta_2;
      r.y1 += r.dx * delta_2;
      r.x2 += -r.dy * delta_2;
      r.y2 += r.dx * delta_2;
      r.width -= delta;
      LF_STAT(++*n_nfa);
      log_nfa_new = d_rect_nfa(f, r, logNT, n_px);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec;
  for (int n = 0; n < 5; n++) {   // reduce the other side
    if ((r.width - delta) >= 0.5) {
      r.x1 -= -r.dy * delta_2;
      r.y1 -= r.dx * delta_2;
      r.x2 -= -r.dy * delta_2;
      r.y2 -= r.dx * delta_2;
      r.width -= delta;
      LF_STAT(++*n_nfa);
      log_nfa_new = d_rect_nfa(f, r, logNT, n_px);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec;
  {
    double lv[5];
    d_rect_nfa_finer5(f, r, logNT, n_px, lv);
    for (int n = 0; n < 5; n++) {   // even finer precisions
      r.p /= 2.0; r.plev++;
      r.prec = r.p * LF_PI;
      LF_STAT(++*n_nfa);
      log_nfa_new = lv[n];
      if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; *rec = r; }
    }
  }
  return log_nfa;
}

// reduce_region_radius (lsd.cpp:1775-1841).  The reference removes far pixels with
// "reg[i] = reg[last]; --size; --i".  Its result is: every position i < #keep that holds a far
// pixel ("hole", ascending i) receives the kept pixels found at positions >= #keep, taken from the
// END backwards.  That permutation is reproduced with two ballot-ranked passes over the list.
template <class FV>
__device__ bool d_reduce_region_radius(const FV &f, int *reg_size, double reg_angle, double prec,
                                       double p, Rect *rec, double density_th) {
  const int N = f.N, lane = f.lane;
  const int NM = f.cap;   // scratch capacity: holes grow from the bottom, fillers from the top
  int size = *reg_size;
  double density = (double)size / (d_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density >= density_th) return true;
  uint32_t pk0 = fv_reg(f, 0, size);
  double xc = (double)(int)(pk0 & 0xffffu), yc = (double)(int)(pk0 >> 16);
  double rad1 = d_dist(xc, yc, rec->x1, rec->y1);
  double rad2 = d_dist(xc, yc, rec->x2, rec->y2);
  double rad = rad1 > rad2 ? rad1 : rad2;
  while (density < density_th) {
    rad *= 0.75;
    int nkeep = 0;
    for (int base = 0; base < size; base += 64) {
      int i = base + lane;
      bool v = i < size;
      uint32_t pk = v ? fv_reg(f, i, size) : 0u;
      bool keep = v && !(d_dist(xc, yc, (double)(int)(pk & 0xffffu), (double)(int)(pk >> 16)) > rad);
      nkeep += __popcll(__ballot(keep));
    }
    int nh = 0, nf = 0;
    for (int base = 0; base < size; base += 64) {
      int i = base + lane;
      bool v = i < size;
      uint32_t pk = v ? fv_reg(f, i, size) : 0u;
      int rx = (int)(pk & 0xffffu), ry = (int)(pk >> 16);
      bool far = v && (d_dist(xc, yc, (double)rx, (double)ry) > rad);
      if (far) fv_unmark(f, rx, ry);
      bool hole = far && i < nkeep;
      bool fill = v && !far && i >= nkeep;
      u64 mh = __ballot(hole), mf = __ballot(fill);
      if (hole) f.tmp[nh + __popcll(mh & lanemask_lt())] = (uint32_t)i;
      if (fill) f.tmp[NM - 1 - (nf + __popcll(mf & lanemask_lt()))] = pk;
      nh += __popcll(mh);
      nf += __popcll(mf);
    }
    wave_mem_order();
    // hole j (ascending position) <- fillers from the end: filler k (ascending) sits at tmp[NM-1-k]
    for (int base = 0; base < nh; base += 64) {
      int j = base + lane;
      if (j < nh) {
        uint32_t at = f.tmp[j], fl = f.tmp[NM - nf + j];
        if (FV::kLU ? !*f.ring_ok : true) f.reg[at] = fl;      // (LU: the memory list exists only for regions beyond the ring)
        if (*f.ring_ok) f.ring[at] = fl;
      }
    }
    wave_mem_order();
    size = nkeep;
    if (size < 2) { *reg_size = size; return false; }
    d_region2rect(f, size, reg_angle, prec, p, 0, rec);
    density = (double)size / (d_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  }
  *reg_size = size;
  return true;
}

// refine (lsd.cpp:1853-1921)
template <class FV>
__device__ bool d_refine(const FV &f, int *reg_size, double reg_angle, double prec, double p,
                         Rect *rec, double density_th, u64 *n_steps) {
  const int N = f.N, lane = f.lane;
  int size = *reg_size;
  double density = (double)size / (d_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density >= density_th) return true;
  uint32_t pk0 = fv_reg(f, 0, size);
  int sx = (int)(pk0 & 0xffffu), sy = (int)(pk0 >> 16);
  double xc = (double)sx, yc = (double)sy;
  double ang_c = f.angles[sy * N + sx];
  double sum = 0.0, s_sum = 0.0;
  int n = 0;
  for (int base = 0; base < size; base += 64) {
    int i = base + lane;
    bool v = i < size;
    uint32_t pk = v ? fv_reg(f, i, size) : 0u;
    int rx = (int)(pk & 0xffffu), ry = (int)(pk >> 16);
    if (v) fv_unmark(f, rx, ry);
    if constexpr (FV::kMW) { if (v && f.ever) { if (i < f.ever_cap) f.ever[i] = pk; } }   // first growth, kept for validation
    bool q = v && d_dist(xc, yc, (double)rx, (double)ry) < rec->width;
    double ang_d = q ? d_angle_diff_signed(f.angles[ry * N + rx], ang_c) : 0.0;
    u64 m = __ballot(q);
    while (m) {               // reference order: ascending i
      int j = __builtin_ctzll(m);
      m &= m - 1;
      double d = rl64(ang_d, j);
      sum += d;
      s_sum += d * d;
      ++n;
    }
  }
  wave_mem_order();
  double mean_angle = sum / (double)n;
  double tau = 2.0 * lf_sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
  if constexpr (FV::kMW) { if (f.ever) { if (size > f.ever_cap) { *f.overflow = 1; } *f.n_ever = size < f.ever_cap ? size : f.ever_cap; } }
  const double ct = lf_cos(tau), ct2 = ct * ct;      // band of the alignment pre-filter for this tolerance: cos^2 tau -+ 1e-12
  if constexpr (FV::kLU) size = d_region_grow_lu(f, sx, sy, tau, ct2 + 1e-12, ct2 - 1e-12, &reg_angle, n_steps);   // (a region of < 2 pixels is dropped below: its angle is never read)
  else size = d_region_grow(f, sx, sy, tau, ct2 + 1e-12, ct2 - 1e-12, &reg_angle, n_steps);
  *reg_size = size;
  if constexpr (FV::kMW) {
  if (*f.overflow) return false;
  if (f.ever) {   // second growth: append
    int n0 = *f.n_ever;
    if (n0 + size > f.ever_cap) *f.overflow = 1;
    else { for (int i = lane; i < size; i += 64) f.ever[n0 + i] = f.reg[i]; *f.n_ever = n0 + size; }
    wave_mem_order();
    if (*f.overflow) return false;
  }
  }
  if (size < 2) return false;
  d_region2rect(f, size, reg_angle, prec, p, 0, rec);
  density = (double)size / (d_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density < density_th) return d_reduce_region_radius(f, reg_size, reg_angle, prec, p, rec, density_th);
  return true;
}

// LineSegmentDetection main loop (lsd.cpp:1996-2053): one wavefront per frame.
#ifndef LF_SWEEP_PRIO
#define LF_SWEEP_PRIO 3     // wave issue priority (s_setprio): the sweep is one dependent chain per frame -- it goes first, the
#endif                      // VALU-bound kernels of the other passes fill the slots it leaves
#ifndef LF_SWEEP_WAVES
#define LF_SWEEP_WAVES 3
#endif
#define LF_LU_WORDS 6144    // LDS bitmap of k_lsd_sweep_lu in 32-bit words: 512 x 384 pixels (640 x 480 input); larger scaled
                            // images take k_lsd_sweep (global `used`)
// The seed loop for one frame; FV = FrameView (k_lsd_sweep) or FrameViewLU (k_lsd_sweep_lu: tiles = its two LDS tile slots).
template <class FV>
__device__ __forceinline__ void d_sweep_frame(FV &f, const LsdConsts &c, const LsdBuffers &b, int fidx, lf_d2 (*tiles)[64]) {
  const int lane = f.lane;
  const size_t NM = (size_t)c.N * c.M;
  uint16_t *labels = b.labels + fidx * NM;
  const uint32_t *seeds = b.seeds + fidx * NM;
  double *segs = b.segs + (size_t)fidx * c.seg_cap * LF_SEG_STRIDE;
  const int nseeds = b.nseeds[fidx];
  u64 n_grow = 0, n_steps = 0, n_nfa = 0, n_px = 0, n_regpx = 0;
#ifdef LF_SWEEP_PROFILE   // build with LF_EXTRA_CFLAGS=-DLF_SWEEP_PROFILE: s_memtime per phase -> stats[8..15]
  u64 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 gp[4] = {0, 0, 0, 0};
  f.gprof = gp;
  u64 t_prev = __builtin_amdgcn_s_memtime();
  const u64 t_begin = t_prev;
#define PROF(k) do { u64 t_now = __builtin_amdgcn_s_memtime(); cyc[k] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define PROF(k) do { } while (0)
#endif
  int ls_count = 0;
  int s = 0;
  // seeds are taken 64 at a time: the window's addresses are loaded once, only `used` is looked at again after
  // every region (a region may have swallowed later seeds of the window)
  int wbase = -64, wlast = 63;
  uint32_t addr = 0u;
  int sxw = 0, syw = 0;        // this lane's seed of the window
  lf_d2 wcs = (lf_d2){0.0, 0.0};   // plain view: its (cos, sin)
  bool v = false;
  // LU: the two tile slots -- origin of the staged 8 x 8 pixels, DMA possibly still in flight, slot of the last region
  int tox0 = -4096, toy0 = -4096, tox1 = -4096, toy1 = -4096, tcur = 0;
  bool tpend0 = false, tpend1 = false;
  // a slot serves a seed whose 3 x 3 neighbourhood lies inside it (tiles are immutable snapshots: any landed slot stays valid;
  // consecutive seeds of the list are mostly neighbours in the image, so the slot of the last region usually serves the next)
#define LU_COVERS(ox, oy, x, y) ((unsigned)((x) - (ox) - 1) < (unsigned)(LF_TILE_W - 2) && (unsigned)((y) - (oy) - 1) < (unsigned)(LF_TILE_W - 2))
  while (s < nseeds) {
    PROF(6);
    if (wlast >= 63) {            // next window
      wbase += 64; wlast = -1;
      s = wbase;
      if (s >= nseeds) break;
      int idx = s + lane;
      v = idx < nseeds;
      addr = v ? seeds[idx] : 0u;
      if constexpr (FV::kLU) { syw = (int)addr / c.N; sxw = (int)addr - syw * c.N; }   // (the bitmap is addressed by (x, y))
      if constexpr (!FV::kLU) wcs = *(const lf_d2 *)&f.cossin[2 * (size_t)addr];   // (cos, sin) of the window's 64 seeds, one gather
    }
    bool isfree;                                                       // angles != NOTDEF holds for every listed pixel
    if constexpr (FV::kLU) isfree = v && lane > wlast && !fv_is_used(f, sxw, syw);
    else isfree = v && lane > wlast && f.used[addr] == 0;
    u64 m = __ballot(isfree);
    if (m == 0) { wlast = 63; continue; }
    int L = __builtin_ctzll(m);
    wlast = L;
    int sx, sy;
    if constexpr (FV::kLU) { sx = rl32(sxw, L); sy = rl32(syw, L); }
    else { const int sa = rl32((int)addr, L); sx = sa % c.N; sy = sa / c.N; }
    if constexpr (FV::kLU) {
      // the seed's (cos, sin) tile: the slot of the last region if it covers the seed, else the other slot if it does (staged
      // while that region was processed), else staged now around the seed
      const bool c0 = LU_COVERS(tox0, toy0, sx, sy), c1 = LU_COVERS(tox1, toy1, sx, sy);
      int cs;
      if (tcur ? c1 : c0) cs = tcur;
      else if (tcur ? c0 : c1) cs = tcur ^ 1;
      else {
        cs = tcur ^ 1;
        lu_stage_tile(f.cossin, tiles[cs], c.N, c.M, sx - LF_TILE_AT, sy - LF_TILE_AT, lane);
        if (cs) { tox1 = sx - LF_TILE_AT; toy1 = sy - LF_TILE_AT; tpend1 = true; } else { tox0 = sx - LF_TILE_AT; toy0 = sy - LF_TILE_AT; tpend0 = true; }
      }
      if (cs ? tpend1 : tpend0) { lu_wait_staged(); tpend0 = false; tpend1 = false; }
      tcur = cs;
      // stage the tile of the next free seed of the window behind this region's work, unless a slot already serves it
      const u64 m2 = m & (m - 1);
      if (m2) {
        const int L2 = __builtin_ctzll(m2), nx = rl32(sxw, L2), ny = rl32(syw, L2);
        if (!LU_COVERS(tox0, toy0, nx, ny) && !LU_COVERS(tox1, toy1, nx, ny)) {
          lu_stage_tile(f.cossin, tiles[cs ^ 1], c.N, c.M, nx - LF_TILE_AT, ny - LF_TILE_AT, lane);
          if (cs) { tox0 = nx - LF_TILE_AT; toy0 = ny - LF_TILE_AT; tpend0 = true; } else { tox1 = nx - LF_TILE_AT; toy1 = ny - LF_TILE_AT; tpend1 = true; }
        }
      }
      f.tile = tiles[cs]; f.tx0 = cs ? tox1 : tox0; f.ty0 = cs ? toy1 : toy0;
    }
    double reg_angle;
    LF_STAT(++n_grow);
    PROF(0);
    int reg_size;
    if constexpr (FV::kLU) reg_size = d_region_grow_lu(f, sx, sy, c.prec, c.k_hi, c.k_lo, &reg_angle, &n_steps);
    else reg_size = d_region_grow<FV, true>(f, sx, sy, c.prec, c.k_hi, c.k_lo, &reg_angle, &n_steps, rl64(wcs.x, L), rl64(wcs.y, L));
    PROF(1);
    LF_STAT(n_regpx += (u64)reg_size);
    if (reg_size < c.min_reg_size) continue;
    if (reg_angle == -1e300) reg_angle = f.angles[sy * c.N + sx];   // a region of the seed alone that is kept (min_reg_size <= 1)
    Rect rec;
    d_region2rect(f, reg_size, reg_angle, c.prec, c.p, 0, &rec);
    PROF(2);
    bool okr = d_refine(f, &reg_size, reg_angle, c.prec, c.p, &rec, c.density_th, &n_steps);
    PROF(3);
    if (!okr) continue;
    double log_nfa = d_rect_improve(f, &rec, c.logNT, c.eps, &n_nfa, &n_px);
    PROF(4);
    if (log_nfa <= c.eps) continue;
    ++ls_count;
    rec.x1 += 0.5; rec.y1 += 0.5;
    rec.x2 += 0.5; rec.y2 += 0.5;
    if (c.scale != 1.0) {
      rec.x1 /= c.scale; rec.y1 /= c.scale;
      rec.x2 /= c.scale; rec.y2 /= c.scale;
      rec.width /= c.scale;
    }
    if (ls_count <= c.seg_cap && lane == 0) {
      double *o = segs + (size_t)(ls_count - 1) * LF_SEG_STRIDE;
      o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width;
    }
    for (int base = 0; base < reg_size; base += 64) {
      int i = base + lane;
      if (i < reg_size) {
        uint32_t pk = fv_reg(f, i, reg_size);
        labels[(int)(pk >> 16) * c.N + (int)(pk & 0xffffu)] = (uint16_t)ls_count;
      }
    }
    PROF(5);
  }
  if (lane == 0) {
    b.nsegs[fidx] = ls_count;
    if (b.stats) {
      unsigned long long *st = b.stats + (size_t)fidx * LF_STATS_STRIDE;
      st[0] = n_grow; st[1] = n_steps; st[2] = n_nfa; st[3] = n_px; st[4] = n_regpx;
      st[5] = (u64)nseeds; st[6] = 0; st[7] = 0;
#ifdef LF_SWEEP_PROFILE
      for (int k = 0; k < 7; k++) st[8 + k] = cyc[k];   // 0 seed scan, 1 grow, 2 region2rect, 3 refine, 4 rect_improve, 5 output, 6 loop
      st[15] = t_prev - t_begin;
      st[6] = gp[1]; st[7] = gp[2]; st[14] = gp[0];
#endif
    }
  }
}
#undef PROF
#undef LU_COVERS
template <class FV>
__device__ __forceinline__ void d_bind_frame(FV &f, const LsdConsts &c, const LsdConsts *dc, const LsdBuffers &b, int fidx) {
  const size_t NM = (size_t)c.N * c.M;
  f.N = c.N; f.M = c.M; f.lane = lane_id();
  f.angles = b.angles + fidx * NM;
  f.modgrad = b.modgrad + fidx * NM;
  f.cossin = b.cossin + 2 * fidx * NM;
  f.lgam = b.lgam;
  f.nfa_tab = b.nfa_tab;
  f.dc = dc;
  f.used = b.used + fidx * NM;
  f.ubits = nullptr; f.wpr = 0; f.tile = nullptr; f.tx0 = 0; f.ty0 = 0;
  f.tag = nullptr;
  f.reg = b.reg + fidx * NM;
  f.tmp = b.tmp + fidx * NM;
  f.cap = (int)NM;
  f.ever = nullptr; f.ever_cap = 0; f.n_ever = nullptr; f.overflow = nullptr;
}
__global__ void __launch_bounds__(64, LF_SWEEP_WAVES) k_lsd_sweep(LsdConsts c, const LsdConsts *dc, LsdBuffers b) {
  __builtin_amdgcn_s_setprio(LF_SWEEP_PRIO);
  __shared__ uint32_t ring1[FrameView::kRing];
  __shared__ double sums1[64 * 4];
  __shared__ int ring_ok1;
  FrameView f;
  d_bind_frame(f, c, dc, b, blockIdx.x);
  f.sums = sums1; f.ring_ok = &ring_ok1; f.ring = ring1;
  d_sweep_frame(f, c, b, blockIdx.x, nullptr);
}
// The same sweep with the frame's `used` mask (+ NOTDEF) as a bitmap in LDS and the seeds' (cos, sin) neighbourhoods staged in
// LDS one region ahead (d_region_grow_lu): 30.0 KB of LDS per frame, five frames per CU.
__global__ void __launch_bounds__(64, LF_SWEEP_WAVES) k_lsd_sweep_lu(LsdConsts c, const LsdConsts *dc, LsdBuffers b) {
  __builtin_amdgcn_s_setprio(LF_SWEEP_PRIO);
  __shared__ __attribute__((aligned(16))) uint32_t ubits1[LF_LU_WORDS];
  __shared__ __attribute__((aligned(16))) lf_d2 tiles1[2][64];
  __shared__ uint32_t ring1[FrameViewLU::kRing];
  __shared__ double sums1[64 * 4];
  __shared__ int ring_ok1;
  FrameViewLU f;
  const int fidx = blockIdx.x;
  d_bind_frame(f, c, dc, b, fidx);
  f.sums = sums1; f.ring_ok = &ring_ok1; f.ring = ring1;
  f.ubits = ubits1; f.wpr = (c.N + 31) >> 5;
  f.used = nullptr;
  // the bitmap starts as the frame's NOTDEF mask (written by k_ll_angle; pad bits of a row are set)
  const int nw = c.M * f.wpr;
  const uint32_t *nd = b.ndbits + (size_t)fidx * nw;
  for (int i = f.lane; i < nw; i += 64) ubits1[i] = nd[i];
  wave_mem_order();
  d_sweep_frame(f, c, b, fidx, tiles1);
}


// ----------------------------------------------------------------------------------------------
// Multi-wavefront sweep: W wavefronts per frame, SPECULATIVE regions with IN-ORDER commit.
//
// The reference visits the seeds in list order and every region sees the `used` marks of all
// earlier regions.  Here each wavefront takes the next unresolved seed (a TICKET, handed out in list
// order under an LDS lock), grows / refines / validates its region against the COMMITTED `used` mask
// plus its own private marks (f.tag), and then waits until all earlier tickets have committed.  At
// that point the committed mask is exactly the state the sequential algorithm would see, so the
// speculation is valid iff (a) the seed is still unused and (b) no pixel the region ever accepted has
// been marked meanwhile (pixels it saw as used stay used: commits are permanent; pixels it rejected
// as not aligned stay rejected).  A valid region is committed as is; an invalid one is re-run right
// there -- at the frontier nothing can interfere -- unless its seed got covered (then the sequential
// algorithm would have skipped it).  Segment numbers are assigned at commit, i.e. in list order.  The
// result is bit-identical to the one-wavefront sweep; what changes is that W regions are in flight.
struct SweepCtl {
  int lock, scan, next_ticket, frontier, ls_count, done;
};
struct RegionResult { int reg_size; int accepted; Rect rec; };

template <class FV>
__device__ __noinline__ void mw_process(const FV &f, const LsdConsts &c, int sx, int sy, RegionResult *out,
                           u64 *n_steps, u64 *n_nfa, u64 *n_px) {
  double reg_angle;
  out->accepted = 0;
  int reg_size = d_region_grow(f, sx, sy, c.prec, c.k_hi, c.k_lo, &reg_angle, n_steps);
  out->reg_size = reg_size;
  if (f.overflow && *f.overflow) return;
  if (reg_size < c.min_reg_size) return;
  Rect rec;
  d_region2rect(f, reg_size, reg_angle, c.prec, c.p, 0, &rec);
  bool okr = d_refine(f, &reg_size, reg_angle, c.prec, c.p, &rec, c.density_th, n_steps);
  out->reg_size = reg_size;
  if (f.overflow && *f.overflow) return;
  if (!okr) return;
  double log_nfa = d_rect_improve(f, &rec, c.logNT, c.eps, n_nfa, n_px);
  if (log_nfa <= c.eps) return;
  out->accepted = 1;
  out->rec = rec;
}

template <int W>
__global__ void __launch_bounds__(W * 64) k_lsd_sweep_mw(LsdConsts c, const LsdConsts *dc, LsdBuffers b) {
  __shared__ SweepCtl ctl;
  __shared__ uint32_t rings[W][FrameViewMW::kRing];
  __shared__ double sums_w[W][64 * 4];
  __shared__ int ring_ok_w[W];
  __shared__ int s_never[W], s_over[W];
  const int fidx = blockIdx.x, lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  const size_t NM = (size_t)c.N * c.M;
  if (threadIdx.x == 0) { ctl.lock = 0; ctl.scan = 0; ctl.next_ticket = 0; ctl.frontier = 0; ctl.ls_count = 0; ctl.done = 0; }
  __syncthreads();
  FrameViewMW f;
#ifdef LF_SWEEP_PROFILE
  __shared__ u64 gprof_sink[4];
  f.gprof = gprof_sink;
#endif
  f.N = c.N; f.M = c.M; f.lane = lane;
  f.angles = b.angles + fidx * NM;
  f.modgrad = b.modgrad + fidx * NM;
  f.cossin = b.cossin + 2 * fidx * NM;
  f.lgam = b.lgam;
  f.nfa_tab = b.nfa_tab;
  f.dc = dc;
  f.used = b.used + fidx * NM;
  f.tag = b.mw_tag + ((size_t)fidx * W + wave) * NM;
  uint32_t *reg_small = b.mw_lists + (((size_t)fidx * W + wave) * 4) * LF_MW_CAP;
  uint32_t *tmp_small = reg_small + LF_MW_CAP;
  f.ring = rings[wave];
  f.sums = sums_w[wave];
  f.ring_ok = &ring_ok_w[wave];
  f.ever = reg_small + 2 * LF_MW_CAP;
  f.ever_cap = 2 * LF_MW_CAP;
  f.n_ever = &s_never[wave];
  f.overflow = &s_over[wave];
  uint32_t *reg_big = b.reg + fidx * NM, *tmp_big = b.tmp + fidx * NM;
  uint16_t *labels = b.labels + fidx * NM;
  const uint32_t *seeds = b.seeds + fidx * NM;
  double *segs = b.segs + (size_t)fidx * c.seg_cap * LF_SEG_STRIDE;
  const int nseeds = b.nseeds[fidx];
  volatile SweepCtl *vc = &ctl;
  u64 n_steps = 0, n_nfa = 0, n_px = 0, n_regions = 0, n_redo = 0, n_dropped = 0;
  for (;;) {
    // ---- take the next ticket: scan the seed list for the next seed not yet used (committed state)
    if (lane == 0) { while (atomicCAS(&ctl.lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int pos = vc->scan, ticket = -1, sa = 0;
    while (pos < nseeds) {
      int idx = pos + lane;
      bool v = idx < nseeds;
      uint32_t addr = v ? seeds[idx] : 0u;
      bool isfree = v && f.used[addr] == 0;
      u64 m = __ballot(isfree);
      if (m == 0) { pos += 64; continue; }
      int L = __builtin_ctzll(m);
      sa = rl32((int)addr, L);
      pos += L + 1;
      ticket = vc->next_ticket;
      break;
    }
    if (lane == 0) { vc->scan = pos; if (ticket >= 0) vc->next_ticket = ticket + 1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicExch(&ctl.lock, 0);
    if (ticket < 0) break;
    // ---- speculative run on private lists
    const int sx = sa % c.N, sy = sa / c.N;
    RegionResult res;
    f.reg = reg_small; f.tmp = tmp_small; f.cap = LF_MW_CAP;
    if (lane == 0) { s_never[wave] = 0; s_over[wave] = 0; }
    wave_mem_order();
    mw_process(f, c, sx, sy, &res, &n_steps, &n_nfa, &n_px);
    ++n_regions;
    // ---- wait for the frontier
    while (vc->frontier != ticket) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- validate against the committed mask (now == the sequential state before this seed)
    bool valid = (s_over[wave] == 0);
    bool seed_used = f.used[sa] != 0;
    if (valid && !seed_used) {
      const uint32_t *lst = (s_never[wave] > 0) ? f.ever : f.reg;
      const int nl = (s_never[wave] > 0) ? s_never[wave] : res.reg_size;
      bool clash = false;
      for (int base = 0; base < nl; base += 64) {
        int i = base + lane;
        if (i < nl) { uint32_t pk = lst[i]; if (f.used[(int)(pk >> 16) * c.N + (int)(pk & 0xffffu)] != 0) clash = true; }
      }
      if (__ballot(clash) != 0) valid = false;
    }
    if (seed_used || !valid) {
      // drop the tentative marks (everything still tagged is in the current list)
      for (int base = 0; base < res.reg_size; base += 64) {
        int i = base + lane;
        if (i < res.reg_size) { uint32_t pk = f.reg[i]; f.tag[(int)(pk >> 16) * c.N + (int)(pk & 0xffffu)] = 0; }
      }
      wave_mem_order();
      if (seed_used) {      // covered by an earlier region: the reference skips this seed
        ++n_dropped;
        if (lane == 0) vc->frontier = ticket + 1;
        continue;
      }
      ++n_redo;             // re-run at the frontier (always valid), with the frame's full-size lists
      f.reg = reg_big; f.tmp = tmp_big; f.cap = (int)NM;
      uint32_t *ever_save = f.ever;
      f.ever = nullptr;
      if (lane == 0) { s_never[wave] = 0; s_over[wave] = 0; }
      wave_mem_order();
      mw_process(f, c, sx, sy, &res, &n_steps, &n_nfa, &n_px);
      f.ever = ever_save;
    }
    // ---- commit (only the frontier wavefront is ever here)
    int ls = 0;
    if (res.accepted) {
      ls = vc->ls_count + 1;
      if (lane == 0) vc->ls_count = ls;
      Rect rec = res.rec;
      rec.x1 += 0.5; rec.y1 += 0.5;
      rec.x2 += 0.5; rec.y2 += 0.5;
      if (c.scale != 1.0) {
        rec.x1 /= c.scale; rec.y1 /= c.scale;
        rec.x2 /= c.scale; rec.y2 /= c.scale;
        rec.width /= c.scale;
      }
      if (ls <= c.seg_cap && lane == 0) {
        double *o = segs + (size_t)(ls - 1) * LF_SEG_STRIDE;
        o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width;
      }
    }
    for (int base = 0; base < res.reg_size; base += 64) {
      int i = base + lane;
      if (i < res.reg_size) {
        uint32_t pk = f.reg[i];
        int p = (int)(pk >> 16) * c.N + (int)(pk & 0xffffu);
        f.used[p] = 1;
        f.tag[p] = 0;
        if (res.accepted) labels[p] = (uint16_t)ls;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) vc->frontier = ticket + 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) b.nsegs[fidx] = ctl.ls_count;
  if (lane == 0 && b.stats) {
    unsigned long long *st = b.stats + (size_t)fidx * LF_STATS_STRIDE;
    atomicAdd(&st[0], n_regions); atomicAdd(&st[1], n_steps); atomicAdd(&st[2], n_nfa);
    atomicAdd(&st[3], n_redo); atomicAdd(&st[4], n_dropped); atomicAdd(&st[5], n_px);
  }
}

// ----------------------------------------------------------------------------------------------
void lf_lsd_build_tables(const LsdConsts &c, const LsdBuffers &b, hipStream_t st) {
  if (!b.nfa_tab) return;
  hipLaunchKernelGGL(k_nfa_table, dim3((LF_NFA_TAB_TRI + 255) / 256, LF_MAX_PLEVEL), dim3(256), 0, st, c, b.dconsts, b);
}

void lf_lsd_launch(const LsdConsts &c, const LsdBuffers &b, int B, hipStream_t st) {
  const size_t NM = (size_t)c.N * c.M;
  dim3 blk(256);
  if (b.ev_pre) (void)hipEventRecord(b.ev_pre, st);
  hipLaunchKernelGGL(k_gauss_x, dim3((c.N + 255) / 256, (c.H + LF_GX_ROWS - 1) / LF_GX_ROWS, B), blk, 0, st, c, b);
  hipLaunchKernelGGL(k_gauss_y, dim3((c.N + 255) / 256, c.M, B), blk, 0, st, c, b);
  hipLaunchKernelGGL(k_ll_angle, dim3((c.N + 31) / 32, (c.M + 7) / 8, B), blk, 0, st, c, b);
  int nch = (c.N - 1 + LF_SORT_CHUNK_COLS - 1) / LF_SORT_CHUNK_COLS;
  size_t lds = sizeof(unsigned int) * (size_t)c.n_bins;
  hipLaunchKernelGGL(k_seed_hist, dim3(nch, B), blk, lds, st, c, b);
  hipLaunchKernelGGL(k_seed_scan, dim3(B), dim3(1024), 0, st, c, b);
  hipLaunchKernelGGL(k_seed_scatter, dim3(nch, B), dim3(64), lds, st, c, b);
  const bool lu = c.sweep_lu && c.sweep_waves <= 1 && c.M * ((c.N + 31) >> 5) <= LF_LU_WORDS;      // the one-wavefront sweep with `used` in LDS
  if (!lu) (void)hipMemsetAsync(b.used, 0, NM * (size_t)B, st);
  (void)hipMemsetAsync(b.labels, 0, NM * (size_t)B * sizeof(uint16_t), st);
  if (b.ev_sweep0) (void)hipEventRecord(b.ev_sweep0, st);
  if (c.sweep_waves > 1) {
    (void)hipMemsetAsync(b.stats, 0, sizeof(unsigned long long) * LF_STATS_STRIDE * (size_t)B, st);
    if (c.sweep_waves >= 8) hipLaunchKernelGGL(k_lsd_sweep_mw<8>, dim3(B), dim3(8 * 64), 0, st, c, b.dconsts, b);
    else if (c.sweep_waves >= 4) hipLaunchKernelGGL(k_lsd_sweep_mw<4>, dim3(B), dim3(4 * 64), 0, st, c, b.dconsts, b);
    else hipLaunchKernelGGL(k_lsd_sweep_mw<2>, dim3(B), dim3(2 * 64), 0, st, c, b.dconsts, b);
  } else {
#ifndef LF_EXP_SKIP_SWEEP   // (throughput experiments only)
    if (lu) hipLaunchKernelGGL(k_lsd_sweep_lu, dim3(B), dim3(64), 0, st, c, b.dconsts, b);
    else hipLaunchKernelGGL(k_lsd_sweep, dim3(B), dim3(64), 0, st, c, b.dconsts, b);
#endif
  }
  if (b.ev_sweep1) (void)hipEventRecord(b.ev_sweep1, st);
}
