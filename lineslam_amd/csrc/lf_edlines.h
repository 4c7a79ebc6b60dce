// lf_edlines.h -- internal interface of the EDLines detector: what Node::detect3DLines runs for
// algorithm == "EDLINES" (src/line/lineslam.cpp:225-235 -> callEDLines, src/line/utils.cpp:1826-1853 -> the binary-only
// DetectLinesByED of external/EDLines/libEDLines.a).  Paper-level statement, sequential twin oracle/edlines_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"

#define LF_ED_ANCHOR_CAP 32768    // anchors per frame the sort stage holds in LDS (32-bit keys, 128 KB)

struct EdConsts {
  int W, H;
  int sk[5];          // 8-bit fixed-point Gaussian taps (5, sigma 1), host libm
  int min_len;        // minimum line length in pixels
  int nmax;           // length of the kmin table - 1
  int seg_cap;        // rows of the segment output per frame
  int chain_cap;      // pixels per walk
};
struct EdBuffers {
  const uint8_t *gray; size_t gray_frame_stride; int gray_row_stride;
  uint8_t *smooth, *D, *E;      // [B][H*W]
  int16_t *G;                   // [B][H*W]
  unsigned *akeys;              // [B][LF_ED_ANCHOR_CAP]  (4095 - gradient) << 19 | pixel index
  int *nanch;                   // [B]
  unsigned *chain;              // [B][2][chain_cap]  y << 16 | x of the two walks of the current anchor
  const int *kmin;              // [nmax + 1] minimal number of aligned pixels for a meaningful line of n pixels (host table)
  double *segs;                 // [B][seg_cap][5]  rows x1 y1 x2 y2 0 (the layout of the LSD output: the 3D stage reads both)
  int *nsegs;                   // [B]
};
void lf_edlines_launch(const EdConsts &c, const EdBuffers &b, int n_frames, hipStream_t stream);
