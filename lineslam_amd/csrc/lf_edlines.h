// lf_edlines.h -- internal interface of the EDLines detector: what Node::detect3DLines runs for
// algorithm == "EDLINES" (src/line/lineslam.cpp:225-235 -> callEDLines, src/line/utils.cpp:1826-1853 -> the binary-only
// DetectLinesByED of external/EDLines/libEDLines.a).  Restated from the archive's object code; sequential twin
// oracle/edlines_oracle.c (all 166 rows of the reference's example output at its 0.01 px resolution).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/linefront.h"

#define LF_ED_BINS 1024           // gradient values |gx| + |gy| of the 2x2 operator: 0 .. 1020
#define LF_ED_STACK_CAP 16384     // pending walks of one anchor
#define LF_ED_CHAIN_CAP 32767     // chains of one anchor's tree (the binary indexes them with 16-bit integers)
#define LF_ED_LINE_CAP 8192       // line segments of one frame before the validation

struct EdLine { double a, b; int invert, pad_; double sx, sy, ex, ey; int segmentNo, firstPixelIndex, len, pad2_; };   // 72 bytes
struct EdChainRec { int dir, len, parent, child0, child1, pix, tmp, pad_; };   // pix: offset of the chain's first pixel in the walk buffer
struct EdConsts {
  int W, H;
  int min_len;        // max(9, ComputeMinLineLength)
  int lut_size;       // (W + H) / 8: the binary's NFALUT; the table continues to nmax with the same rule evaluated directly
  int nmax;           // entries of kmin
  int seg_cap;        // rows of the segment output per frame
  int anchor_cap;     // anchors per frame
  int segtab_cap;     // pixel chains (edge segments) per frame
};
struct EdBuffers {
  const uint8_t *gray; size_t gray_frame_stride; int gray_row_stride;
  uint8_t *smooth, *D, *E;      // [B][H*W]
  int16_t *G;                   // [B][H*W]
  int *hist;                    // [B][LF_ED_BINS] anchors per gradient value
  unsigned *anchors;            // [B][anchor_cap] pixel indices, strongest gradient first, raster order inside one value
  int *nanch;                   // [B]
  unsigned *walk;               // [B][H*W] pixels (r << 16 | c) of the current anchor's walks
  unsigned *stack;              // [B][LF_ED_STACK_CAP][2]  r << 16 | c, parent << 3 | dir
  EdChainRec *chains;           // [B][LF_ED_CHAIN_CAP + 1]
  int *chain_nos;               // [B][(W + H) * 8]
  unsigned *segpix;             // [B][H*W] pixels of the edge segments, one after the other
  int *segtab;                  // [B][segtab_cap][2] first pixel, number of pixels
  EdLine *lines;                // [B][LF_ED_LINE_CAP] line slots in list order (len = -1: joined into an earlier line)
  int *nsegtab;                 // [B] edge segments of the frame (-1: the walk exceeded a capacity)
  int *seg_nl;                  // [B][segtab_cap][3] lines of the segment before the joining, its first line slot, (spare)
  int *nslots;                  // [B] line slots in use (-1: over capacity)
  uint8_t *lvalid;              // [B][LF_ED_LINE_CAP] ValidateLineSegments per slot
  const int *kmin;              // [nmax] minimal number of aligned pixels for a meaningful line of n pixels (host table)
  const double *atan_lut;       // [1025] atan(i / 1024) (host table, as the binary builds it with libm)
  double *segs;                 // [B][seg_cap][5]  rows x1 y1 x2 y2 0 (the layout of the LSD output: the 3D stage reads both)
  int *nsegs;                   // [B]  (seg_cap + 1 if a capacity above was exceeded: reported by the getters)
};
void lf_edlines_launch(const EdConsts &c, const EdBuffers &b, int n_frames, hipStream_t stream);
