// lf_lsd.h -- internal interface of the gfx950 LSD stage (not part of the C ABI).
//
// Stage a1-a8 of SURVEY.md section 8: callLsd (src/line/utils.cpp:112-135) ->
// LineSegmentDetection (external/lsd/lsd.cpp:1931-2065).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LF_NOTDEF (-1024.0)          // lsd.cpp:91
#define LF_M_3_2_PI 4.71238898038    // lsd.cpp:105 (truncated literal, kept)
#define LF_M_2__PI 6.28318530718     // lsd.cpp:108 (truncated literal, kept)
#define LF_MAX_PLEVEL 11             // p, p/2, ... p/2^10 (rect_improve: two families of 5 halvings)
#define LF_SEG_STRIDE 5              // x1,y1,x2,y2,width
#define LF_BIN_NONE 0xFFFFu
#define LF_SORT_CHUNK_COLS 8         // seed counting sort: columns per chunk
#define LF_MW_CAP 8192               // multi-wave sweep: speculative region list capacity (larger regions re-run at the frontier)
#define LF_MW_MAXW 8                 // wavefronts per frame

// Host-computed constants (host libm, exactly as the reference evaluates them on the CPU).
struct LsdConsts {
  int W, H;            // input image
  int N, M;            // scaled image  (lsd.cpp:549-550)
  int ntaps;           // 1 + 2h        (lsd.cpp:565-566)
  int n_bins;          // 1024
  double max_grad;     // 255
  double rho;          // quant / sin(prec)              (lsd.cpp:1965)
  double prec;         // pi * ang_th / 180              (lsd.cpp:1963)
  double p;            // ang_th / 180                   (lsd.cpp:1964)
  double cos_prec;     // cos(prec), host libm: alignment pre-filter of region_grow
  double k_hi, k_lo;   // cos^2(prec) +- 1e-12: the band of that pre-filter (scalar operands of the decision loop)
  double logNT;        // 5 (log10 N + log10 M) / 2      (lsd.cpp:1983)
  int min_reg_size;    // (int)(-logNT / log10 p)        (lsd.cpp:1984)
  double density_th;   // sysPara.lsd_density_th
  double eps;          // 0
  double scale;        // 0.8
  double logp[LF_MAX_PLEVEL];    // log(p / 2^k)
  double log1mp[LF_MAX_PLEVEL];  // log(1 - p / 2^k)
  double log10p[LF_MAX_PLEVEL];  // log10(p / 2^k)
  int seg_cap;         // rows available per frame in the segment output
  int sweep_waves;     // wavefronts per frame in the seed sweep (1 = sequential kernel)
  int sweep_lu;        // one-wavefront sweep: 1 = k_lsd_sweep_lu (`used` bitmap + seed tiles in LDS, 30 KB per frame), 0 = k_lsd_sweep
};

// Per-batch device pointers (frame f uses offset f * per-frame size).
#define LF_STATS_STRIDE 16
#define LF_NFA_TAB_N 512              // nfa(n, k, level) is tabulated for n < LF_NFA_TAB_N (triangular: n(n+1)/2 + k)
#define LF_NFA_TAB_TRI (LF_NFA_TAB_N * (LF_NFA_TAB_N + 1) / 2)
struct LsdBuffers {
  const uint8_t *gray;   size_t gray_frame_stride;  int gray_row_stride;  // input, bytes
  const double *kx, *ky; // [N][ntaps], [M][ntaps] Gaussian taps
  const int *jx, *jy;    // [N][ntaps], [M][ntaps] source indices after the symmetric boundary
  const double *lgam;    // log_gamma(i), i = 0 .. N*M+1
  const LsdConsts *dconsts; // device copy of the constants (for dynamically indexed tables)
  double *nfa_tab;       // [LF_MAX_PLEVEL][LF_NFA_TAB_TRI] nfa() of small rectangles, filled by k_nfa_table; may be null
  double *aux;           // [B][H][N]
  double *scaled;        // [B][M][N]
  double *angles;        // [B][M][N]
  double *modgrad;       // [B][M][N]
  double *cossin;        // [B][M][N][2] (cos, sin) of the level-line angle, interleaved: one 16-byte gather per pixel (cos = 2 marks NOTDEF)
  uint16_t *bins;        // [B][N][M]  gradient-magnitude bin of every pixel, TRANSPOSED (column-major walk of the seed sort)
  uint32_t *cnt;         // [B][nchunks][n_bins]
  uint32_t *seeds;       // [B][M*N]   pixel address y*N+x in reference list order
  int *nseeds;           // [B]
  uint8_t *used;         // [B][M*N]
  uint32_t *ndbits;      // [B][M][ceil(N/32)] NOTDEF bitmap (bit x & 31 of word x >> 5 of row y), written by k_ll_angle
  uint32_t *reg;         // [B][M*N]   region pixel list, x | y<<16
  uint32_t *tmp;         // [B][M*N]   scratch for reduce_region_radius
  uint8_t *mw_tag;       // [B][W][M*N] private tentative marks of each sweep wavefront
  uint32_t *mw_lists;    // [B][W][4][LF_MW_CAP] private region list, scratch, accepted-ever list (2x)
  uint16_t *labels;      // [B][M*N]   0 = none, k = k-th segment  (the integer pixel support)
  double *segs;          // [B][seg_cap][5]
  int *nsegs;            // [B]  (may exceed seg_cap: overflow)
  unsigned long long *stats; // [B][LF_STATS_STRIDE] optional work counters, may be null
  hipEvent_t ev_pre, ev_sweep0, ev_sweep1;   // optional: recorded before the first kernel / around k_lsd_sweep
};

// (re)computes b.nfa_tab from the constants; call after every change of the LSD parameters
void lf_lsd_build_tables(const LsdConsts &c, const LsdBuffers &b, hipStream_t stream);
void lf_lsd_launch(const LsdConsts &c, const LsdBuffers &b, int n_frames, hipStream_t stream);
