/* oracle/edlines_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU statement of the EDLines detector behind Node::detect3DLines(..., algorithm = "EDLINES")
 * (src/line/lineslam.cpp:225-235 -> callEDLines, src/line/utils.cpp:1826-1853 -> DetectLinesByED of external/EDLines/libEDLines.a).
 *
 * The reference ships EDLines as a BINARY (libEDLines.a: x86-64 objects, not stripped, no source; it calls OpenCV's
 * cvSmooth, absent here, so it cannot be linked either).  This file restates what the OBJECT CODE does, function by function
 * under the symbol names of the archive, with every constant read from its disassembly (objdump -d -r):
 *   DetectLinesByED(uchar*, int, int, EDLines*)        EDLines.o   SmoothImage(sigma 1.0) -> ComputeGradientMapByLSD(thresh 11)
 *                                                                  -> DoDetectEdgesByED(gradient 11, anchor 3) -> min length
 *                                                                  max(9, ComputeMinLineLength) -> SplitSegment2Lines per
 *                                                                  segment -> JoinCollinearLines(6.0, 1.3) -> ValidateLineSegments
 *   SmoothImage                                         ImageSmooth.o   sigma == 1.0: cvSmooth(CV_GAUSSIAN, 5, 5, 0, 0)
 *   ComputeGradientMapByLSD                             GradientOperators.o   2x2 differences, |gx| + |gy|, border = thresh - 1
 *   DoDetectEdgesByED, SortAnchorsByGradValue,          EDInternals.o   anchors (rows / columns 2 .. n-3, both neighbours lower
 *   LongestChain, RetrieveChainNos                                      by >= 3), counting sort, stack walk building a chain
 *                                                                      tree, longest path as the segment, leftovers >= 10 px
 *   ComputeMinLineLength, SplitSegment2Lines,           EDLines.o
 *   JoinCollinearLines, ValidateLineSegments,
 *   ValidateLineSegmentRect, EnumerateRectPoints
 *   LineFit (two forms), ComputeMinDistance,            LineSegment.o
 *   ComputeClosestPoint, TryToJoinTwoLineSegments,
 *   ComputeMinDistanceBetweenTwoLines, UpdateLineParameters
 *   nfa, NFALUT, checkValidationByNFA                   NFA.o
 * The control structure is that of the authors' later public ED_Lib, which the object code matches wherever it was
 * checked (anchor loop, sort, constants 0.5 / 1.0 / 5 bad pixels / 2 good pixels / 80 / 25 / pi/8 / 0.125 / 0.01).
 * OpenCV is absent: cvSmooth is restated from the published algorithm of OpenCV 2.4 (8-bit separable filter, taps
 * 1 4 6 4 1 / 16 as 8-bit fixed point, BORDER_REPLICATE) -- that one step is "parity unpinned" like every OpenCV piece.
 * Anchor to the reference's own output: external/EDLines/house.pgm -> LineSegments.txt (166 rows, two decimals);
 * tests/test_oracle_edlines.py measures how many rows this statement reproduces at the file's 0.01 px resolution.
 * The HIP kernels (lineslam_amd/csrc/lf_edlines.hip) are held bit for bit against THIS file.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_LFMATH     /* the `lf` flavour evaluates what the kernels evaluate on the device with lf_math.h */
#include "../lineslam_amd/csrc/lf_math.h"
#define E_ATAN(x) lf_atan2((x), 1.0)   /* the device evaluates the angle of a line this way */
#define E_ATAN2 lf_atan2
#else
#define E_ATAN atan
#define E_ATAN2 atan2
#endif
/* the NFA look-up table and the minimum line length are HOST tables of the product too (libm, once per image size) */
#define E_LOG10 log10
#define E_LOG log
#define E_EXP exp

/* Precision of the intermediates of the line geometry.  The reference links the 64-bit archive (libEDLines.a, GCC 4.6,
 * SSE2 doubles: every operation rounds to double) -- the default here and what the HIP kernels follow.  The example output
 * the authors ship (LineSegments.txt) was written by their 32-bit test program (EDLinesTest is an i386 ELF; its archive
 * libEDLines-32bit.a, GCC 4.3.4, contains x87 instructions only: 449 in LineSegment.o, no SSE): there, expression
 * intermediates and values returned in st(0) carry the 64-bit mantissa of the x87 registers and round to double only when
 * they are stored (struct fields, reference parameters, arguments pushed on the stack).  ORACLE_ED_X87 restates that:
 * `ereal` locals / return values are long double, everything kept in memory stays double.  Distances of integer pixels to
 * a fitted line land on the thresholds (<= 1.0, error <= 0.5) often enough that the two precisions split two short
 * segments of the example differently (tests/test_oracle_edlines.py).                                                   */
#ifdef ORACLE_ED_X87
typedef long double ereal;
#define E_SQRT sqrtl
#define E_FABS fabsl
#else
typedef double ereal;
#define E_SQRT sqrt
#define E_FABS fabs
#endif

#define ED_GRAD_THRESH 11          /* DetectLinesByED: mov $0xb,%r9d before ComputeGradientMapByLSD / %r8d before DoDetectEdgesByED */
#define ED_ANCHOR_THRESH 3         /* mov $0x3,%r9d */
#define ED_VERTICAL 1              /* dirImg codes (cmpb $0x1 / $0x2) */
#define ED_HORIZONTAL 2
#define ED_ANCHOR 254              /* edgeImg codes (0xfe, 0xff) */
#define ED_EDGE 255
#define ED_MIN_PATH 10             /* cmp $0x9 on len - duplicatePixelCount */
#define ED_LINE_ERROR 1.0
#define ED_MAX_DIST 6.0            /* JoinCollinearLines(lines, 6.0, 1.3) */
#define ED_MAX_ERROR 1.3
#define ED_PI 3.14159265358979323846
enum { ED_LEFT = 1, ED_RIGHT = 2, ED_UP = 3, ED_DOWN = 4 };

static int e_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* SmoothImage with sigma == 1.0: cvSmooth(src, dst, CV_GAUSSIAN, 5, 5, 0, 0) = cv::GaussianBlur(5x5, sigma 0, BORDER_REPLICATE);
 * sigma <= 0 and ksize 5 selects the fixed kernel 1 4 6 4 1 / 16; the 8-bit path runs the row pass in fixed point with
 * 8 fractional bits and rounds once after the column pass (see there) */
void oracle_ed_smooth(const uint8_t *src, int w, int h, uint8_t *dst) {
  static const int ik[5] = {16, 64, 96, 64, 16};
  int x, y, k, *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int s = 0;
      for (k = 0; k < 5; k++) s += ik[k] * src[(size_t)y * w + e_clampi(x + k - 2, 0, w - 1)];
      tmp[(size_t)y * w + x] = s;
    }
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++) {
      int s = 0;
      for (k = 0; k < 5; k++) s += ik[k] * tmp[(size_t)e_clampi(y + k - 2, 0, h - 1) * w + x];
      /* the column pass of OpenCV 2.4 on x86-64 (SymmColumnVec_32s8u, SSE2): four columns at a time in float -- exact
       * here -- converted with cvtps2dq, i.e. rounded to nearest EVEN; the last w % 4 columns go through the scalar
       * fixed-point cast, (sum + 2^15) >> 16 */
      if (x < (w & ~3)) { const int n = s >> 16, r = s & 65535; s = r > 32768 ? n + 1 : (r < 32768 ? n : n + (n & 1)); }
      else s = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)(s > 255 ? 255 : s);
    }
  free(tmp);
}

/* ComputeGradientMapByLSD: first / last row and column = thresh - 1; inside, with A B / C D the 2x2 block whose top-left
 * pixel is (x, y): gx = (D - A) + (B - C), gy = (D - A) - (B - C), G = |gx| + |gy|; the direction is written only where
 * G >= thresh: vertical edge (1) if |gx| >= |gy|, horizontal (2) otherwise.  (D of the other pixels is never consulted.) */
void oracle_ed_gradient(const uint8_t *s, int w, int h, int16_t *G, uint8_t *D) {
  int x, y;
  memset(D, 0, (size_t)w * h);
  for (x = 0; x < w; x++) { G[x] = ED_GRAD_THRESH - 1; G[(size_t)(h - 1) * w + x] = ED_GRAD_THRESH - 1; }
  for (y = 1; y < h - 1; y++) { G[(size_t)y * w] = ED_GRAD_THRESH - 1; G[(size_t)y * w + w - 1] = ED_GRAD_THRESH - 1; }
  for (y = 1; y < h - 1; y++)
    for (x = 1; x < w - 1; x++) {
      const uint8_t *p = s + (size_t)y * w + x;
      int com1 = p[w + 1] - p[0], com2 = p[1] - p[w];
      int gx = abs(com1 + com2), gy = abs(com1 - com2), sum = gx + gy;
      G[(size_t)y * w + x] = (int16_t)sum;
      if (sum >= ED_GRAD_THRESH) D[(size_t)y * w + x] = (gx >= gy) ? ED_VERTICAL : ED_HORIZONTAL;
    }
}

/* the anchor loop of DoDetectEdgesByED (rows 2 .. h-3, columns 2 .. w-3, every pixel: scan interval 1) */
void oracle_ed_anchors(const int16_t *G, const uint8_t *D, int w, int h, uint8_t *E) {
  int x, y;
  memset(E, 0, (size_t)w * h);
  for (y = 2; y < h - 2; y++)
    for (x = 2; x < w - 2; x++) {
      const int g = G[(size_t)y * w + x];
      if (g < ED_GRAD_THRESH) continue;
      if (D[(size_t)y * w + x] == ED_VERTICAL) {
        if (g - G[(size_t)y * w + x + 1] >= ED_ANCHOR_THRESH && g - G[(size_t)y * w + x - 1] >= ED_ANCHOR_THRESH) E[(size_t)y * w + x] = ED_ANCHOR;
      } else {
        if (g - G[(size_t)(y + 1) * w + x] >= ED_ANCHOR_THRESH && g - G[(size_t)(y - 1) * w + x] >= ED_ANCHOR_THRESH) E[(size_t)y * w + x] = ED_ANCHOR;
      }
    }
}

/* ---------------------------------------------------------------- chains */
typedef struct { int r, c; } e_pix;
typedef struct { int r, c, parent, dir; } e_stack;
typedef struct { int dir, len, parent, children[2]; e_pix *pixels; } e_chain;
typedef struct { e_pix *pixels; int n; } e_segment;

static int e_longest_chain(e_chain *ch, int root) {
  int len0 = 0, len1 = 0, mx;
  if (root == -1 || ch[root].len == 0) return 0;
  if (ch[root].children[0] != -1) len0 = e_longest_chain(ch, ch[root].children[0]);
  if (ch[root].children[1] != -1) len1 = e_longest_chain(ch, ch[root].children[1]);
  if (len0 >= len1) { mx = len0; ch[root].children[1] = -1; }
  else { mx = len1; ch[root].children[0] = -1; }
  return ch[root].len + mx;
}
static int e_retrieve_chain_nos(e_chain *ch, int root, int *nos) {
  int count = 0;
  while (root != -1) {
    nos[count++] = root;
    if (ch[root].children[0] != -1) root = ch[root].children[0];
    else root = ch[root].children[1];
  }
  return count;
}

/* ---------------------------------------------------------------- line geometry (LineSegment.o) */
typedef struct { double a, b; int invert; double sx, sy, ex, ey; int segmentNo, firstPixelIndex, len; } e_line;

static void e_line_fit_err(const double *x, const double *y, int count, double *pa, double *pb, double *pe, int *pinvert) {
  ereal S = count, Sx = 0, Sy = 0, Sxx = 0, Sxy = 0, mx, my, dx = 0, dy = 0, D, a, b;
  int i;
  if (count < 2) return;
  for (i = 0; i < count; i++) { Sx += x[i]; Sy += y[i]; }
  mx = Sx / count; my = Sy / count;
  for (i = 0; i < count; i++) { dx += (x[i] - mx) * (x[i] - mx); dy += (y[i] - my) * (y[i] - my); }
  if (dx < dy) { const double *t = x; ereal d = Sx; *pinvert = 1; x = y; y = t; Sx = Sy; Sy = d; }
  else *pinvert = 0;
  for (i = 0; i < count; i++) { Sxx += x[i] * x[i]; Sxy += x[i] * y[i]; }
  D = S * Sxx - Sx * Sx;
  a = (Sxx * Sy - Sx * Sxy) / D;
  b = (S * Sxy - Sx * Sy) / D;
  *pa = (double)a; *pb = (double)b;
  a = *pa; b = *pb;                                  /* (reference parameters: memory) */
  if (b == 0.0) {
    ereal error = 0;
    for (i = 0; i < count; i++) error += E_FABS(a - y[i]);
    *pe = (double)(error / count);
  } else {
    ereal error = 0;
    for (i = 0; i < count; i++) {
      ereal d = -1.0 / b, c = y[i] - d * x[i];
      ereal x2 = (a - c) / (d - b), y2 = a + b * x2;
      error += (x[i] - x2) * (x[i] - x2) + (y[i] - y2) * (y[i] - y2);
    }
    *pe = (double)E_SQRT(error / count);
  }
}
static void e_line_fit(const double *x, const double *y, int count, double *pa, double *pb, int invert) {
  ereal S = count, Sx = 0, Sy = 0, Sxx = 0, Sxy = 0, D;
  int i;
  if (count < 2) return;
  for (i = 0; i < count; i++) { Sx += x[i]; Sy += y[i]; }
  if (invert) { const double *t = x; ereal d = Sx; x = y; y = t; Sx = Sy; Sy = d; }
  for (i = 0; i < count; i++) { Sxx += x[i] * x[i]; Sxy += x[i] * y[i]; }
  D = S * Sxx - Sx * Sx;
  *pa = (double)((Sxx * Sy - Sx * Sxy) / D);
  *pb = (double)((S * Sxy - Sx * Sy) / D);
}
static void e_closest_point(double x1, double y1, double a, double b, int invert, double *xo, double *yo) {
  ereal x2, y2;
  if (invert == 0) {
    if (b == 0) { x2 = x1; y2 = a; }
    else { ereal d = -1.0 / b, c = y1 - d * x1; x2 = (a - c) / (d - b); y2 = a + b * x2; }
  } else {
    if (b == 0) { x2 = a; y2 = y1; }
    else { ereal d = -1.0 / b, c = x1 - d * y1; y2 = (a - c) / (d - b); x2 = a + b * y2; }
  }
  *xo = (double)x2; *yo = (double)y2;
}
static ereal e_min_distance(double x1, double y1, double a, double b, int invert) {
  double x2, y2;
  e_closest_point(x1, y1, a, b, invert, &x2, &y2);
  return E_SQRT(((ereal)x1 - x2) * ((ereal)x1 - x2) + ((ereal)y1 - y2) * ((ereal)y1 - y2));
}
static void e_update_line_parameters(e_line *ls) {
  ereal dx = (ereal)ls->ex - ls->sx, dy = (ereal)ls->ey - ls->sy;
  if (E_FABS(dx) >= E_FABS(dy)) {
    ls->invert = 0;                                  /* y = a + b x */
    if (E_FABS(dy) < 1e-3) { ls->b = 0; ls->a = (double)(((ereal)ls->sy + ls->ey) / 2); }
    else { ls->b = (double)(dy / dx); ls->a = (double)(ls->sy - (ereal)ls->b * ls->sx); }
  } else {
    ls->invert = 1;                                  /* x = a + b y */
    if (E_FABS(dx) < 1e-3) { ls->b = 0; ls->a = (double)(((ereal)ls->sx + ls->ex) / 2); }
    else { ls->b = (double)(dx / dy); ls->a = (double)(ls->sx - (ereal)ls->b * ls->sy); }
  }
}
static ereal e_min_distance_between_two_lines(const e_line *ls1, const e_line *ls2, int *pwhich) {
  ereal dx = (ereal)ls1->sx - ls2->sx, dy = (ereal)ls1->sy - ls2->sy, d = E_SQRT(dx * dx + dy * dy), min = d;
  int which = 0;                                      /* SS */
  dx = (ereal)ls1->sx - ls2->ex; dy = (ereal)ls1->sy - ls2->ey; d = E_SQRT(dx * dx + dy * dy);
  if (d < min) { min = d; which = 1; }                /* SE */
  dx = (ereal)ls1->ex - ls2->sx; dy = (ereal)ls1->ey - ls2->sy; d = E_SQRT(dx * dx + dy * dy);
  if (d < min) { min = d; which = 2; }                /* ES */
  dx = (ereal)ls1->ex - ls2->ex; dy = (ereal)ls1->ey - ls2->ey; d = E_SQRT(dx * dx + dy * dy);
  if (d < min) { min = d; which = 3; }                /* EE */
  if (pwhich) *pwhich = which;
  return min;
}
static int e_try_to_join(e_line *ls1, e_line *ls2, double max_dist, double max_err) {
  int which;
  ereal dist = e_min_distance_between_two_lines(ls1, ls2, &which), dx, dy, prevLen, nextLen, d, mx;
  const e_line *shorter = ls1, *longer = ls2;
  if (dist > max_dist) return 0;
  dx = (ereal)ls1->sx - ls1->ex; dy = (ereal)ls1->sy - ls1->ey; prevLen = E_SQRT(dx * dx + dy * dy);
  dx = (ereal)ls2->sx - ls2->ex; dy = (ereal)ls2->sy - ls2->ey; nextLen = E_SQRT(dx * dx + dy * dy);
  if (prevLen > nextLen) { shorter = ls2; longer = ls1; }
  dist = e_min_distance(shorter->sx, shorter->sy, longer->a, longer->b, longer->invert);
  dist += e_min_distance((double)(((ereal)shorter->sx + shorter->ex) / 2.0), (double)(((ereal)shorter->sy + shorter->ey) / 2.0), longer->a, longer->b, longer->invert);
  dist += e_min_distance(shorter->ex, shorter->ey, longer->a, longer->b, longer->invert);
  dist /= 3.0;
  if (dist > max_err) return 0;
  /* the two end points that are farthest apart (city-block) become the joined line's end points */
  dx = E_FABS((ereal)ls1->sx - ls2->sx); dy = E_FABS((ereal)ls1->sy - ls2->sy); d = dx + dy; mx = d; which = 1;
  dx = E_FABS((ereal)ls1->sx - ls2->ex); dy = E_FABS((ereal)ls1->sy - ls2->ey); d = dx + dy; if (d > mx) { mx = d; which = 2; }
  dx = E_FABS((ereal)ls1->ex - ls2->sx); dy = E_FABS((ereal)ls1->ey - ls2->sy); d = dx + dy; if (d > mx) { mx = d; which = 3; }
  dx = E_FABS((ereal)ls1->ex - ls2->ex); dy = E_FABS((ereal)ls1->ey - ls2->ey); d = dx + dy; if (d > mx) { mx = d; which = 4; }
  if (which == 1) { ls1->ex = ls2->sx; ls1->ey = ls2->sy; }
  else if (which == 2) { ls1->ex = ls2->ex; ls1->ey = ls2->ey; }
  else if (which == 3) { ls1->sx = ls2->sx; ls1->sy = ls2->sy; }
  else { ls1->sx = ls1->ex; ls1->sy = ls1->ey; ls1->ex = ls2->ex; ls1->ey = ls2->ey; }
  if (ls1->firstPixelIndex + ls1->len + 5 >= ls2->firstPixelIndex) ls1->len += ls2->len;
  else if (ls2->len > ls1->len) { ls1->firstPixelIndex = ls2->firstPixelIndex; ls1->len = ls2->len; }
  e_update_line_parameters(ls1);
  return 1;
}

/* ---------------------------------------------------------------- NFA (NFA.o: LSD's nfa with p = 1/8) */
static double e_log_gamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * E_LOG(x + 5.5) - (x + 5.5), b = 0.0;
  int n;
  for (n = 0; n < 7; n++) { a -= E_LOG(x + (double)n); b += q[n] * pow(x, (double)n); }
  return a + E_LOG(b);
}
static double e_log_gamma_windschitl(double x) {
  return 0.918938533204673 + (x - 0.5) * E_LOG(x) - x + 0.5 * x * E_LOG(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
}
static double e_log_gamma(double x) { return x > 15.0 ? e_log_gamma_windschitl(x) : e_log_gamma_lanczos(x); }
static double e_nfa(int n, int k, double p, double logNT) {
  const double tolerance = 0.1;
  double log1term, term, bin_term, mult_term, bin_tail, err, p_term;
  int i;
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * E_LOG10(p);
  p_term = p / (1.0 - p);
  log1term = e_log_gamma((double)n + 1.0) - e_log_gamma((double)k + 1.0) - e_log_gamma((double)(n - k) + 1.0) + (double)k * E_LOG(p) +
             (double)(n - k) * E_LOG(1.0 - p);
  term = E_EXP(log1term);
  if (term == 0.0) {
    if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
    else return -logNT;
  }
  bin_tail = term;
  for (i = k + 1; i <= n; i++) {
    bin_term = (double)(n - i + 1) * (1.0 / (double)i);   /* (the binary multiplies by a tabulated reciprocal, as LSD does) */
    mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * fabs(-E_LOG10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -E_LOG10(bin_tail) - logNT;
}
typedef struct { int *lut, size; double prob, logNT; } e_nfalut;
static void e_nfalut_init(e_nfalut *L, int size, double prob, double logNT) {
  int i, j = 1;
  L->size = size; L->prob = prob; L->logNT = logNT;
  L->lut = (int *)malloc(sizeof(int) * (size_t)(size > 0 ? size : 1));
  L->lut[0] = 1;
  for (i = 1; i < size; i++) {
    double ret;
    L->lut[i] = size + 1;
    ret = e_nfa(i, j, prob, logNT);
    if (ret < 0) {
      while (j < i) { j++; ret = e_nfa(i, j, prob, logNT); if (ret >= 0) break; }
      if (ret < 0) continue;
    }
    L->lut[i] = j;
  }
}
static int e_check_nfa(const e_nfalut *L, int n, int k) {
  if (n >= L->size) return e_nfa(n, k, L->prob, L->logNT) >= 0.0;
  return k >= L->lut[n];
}
void oracle_ed_nfa_table(int w, int h, int nmax, int *kmin) {     /* kmin[n] for n < nmax = (w + h) / 8, as NFALUT */
  e_nfalut L;
  int i;
  e_nfalut_init(&L, nmax, 0.125, 2.0 * (E_LOG10((double)w) + E_LOG10((double)h)));
  for (i = 0; i < nmax; i++) kmin[i] = L.lut[i];
  free(L.lut);
}
double oracle_ed_nfa(int n, int k, int w, int h) { return e_nfa(n, k, 0.125, 2.0 * (E_LOG10((double)w) + E_LOG10((double)h))); }

/* ComputeMinLineLength: Round(2 (log10 w + log10 h) / 0.9031 (= -log10 0.125) * 0.5); DetectLinesByED keeps at least 9 */
static double e_round(double d) { return floor(d + 0.5); }
int oracle_ed_min_line_len(int w, int h) {
  const double logNT = 2.0 * (E_LOG10((double)w) + E_LOG10((double)h));
  int n = (int)e_round(logNT / 0.90308998699194354 * 0.5);
  return n < 9 ? 9 : n;
}

/* ---------------------------------------------------------------- validation (EDLines.o) */
static double e_line_angle(const e_line *ls) {
  double lineAngle;
  if (ls->invert == 0) lineAngle = E_ATAN(ls->b);
  else lineAngle = E_ATAN(1.0 / ls->b);
  if (lineAngle < 0) lineAngle += ED_PI;
  return lineAngle;
}
/* myAtan2 (MyMath.o): the angle of (xx, yy) folded into [0, pi) from a table of atan(i / 1024), i = 0 .. 1024, indexed by
 * the TRUNCATED ratio of the smaller to the larger magnitude (the table is a host table of the product too: libm atan) */
static double e_atan_lut[1025];
static int e_atan_lut_ready = 0;
void oracle_ed_atan_table(double *t) { int i; for (i = 0; i <= 1024; i++) t[i] = atan((double)i * (1.0 / 1024.0)); }
static double e_my_atan2(double yy, double xx) {
  double ay = fabs(yy), ax = fabs(xx), angle;
  int invert = 0;
  if (!e_atan_lut_ready) { oracle_ed_atan_table(e_atan_lut); e_atan_lut_ready = 1; }
  if (ax < 1e-10) return (ay < 1e-10) ? 0.0 : ED_PI / 2.0;
  if (ay > ax) { double t = ay; ay = ax; ax = t; invert = 1; }
  angle = e_atan_lut[(int)(ay / ax * 1024.0)];
  if ((xx >= 0 && yy >= 0) || (xx < 0 && yy < 0)) return invert ? ED_PI / 2.0 - angle : angle;
  return invert ? angle + ED_PI / 2.0 : ED_PI - angle;
}
static int e_aligned(const uint8_t *img, int w, int r, int c, double lineAngle) {
  const uint8_t *p = img + (size_t)r * w + c;
  const double prec = ED_PI / 8.0;
  int com1 = p[w + 1] - p[-w - 1], com2 = p[-w + 1] - p[w - 1];
  int gx = com1 + com2 + p[1] - p[-1], gy = com1 - com2 + p[w] - p[-w];
  double pixelAngle = e_my_atan2((double)gx, (double)-gy), diff = fabs(lineAngle - pixelAngle);
  return diff <= prec || diff >= ED_PI - prec;
}
/* EnumerateRectPoints: the pixels of the rectangle of width 2 around the line (LSD's rectangle iterator) */
static void e_enumerate_rect_points(double sx, double sy, double ex, double ey, int *ptsx, int *ptsy, int *pn) {
  double vxTmp[4], vyTmp[4], vx[4], vy[4];
  double x1 = sx, y1 = sy, x2 = ex, y2 = ey, width = 2, dx = x2 - x1, dy = y2 - y1, vLen = sqrt(dx * dx + dy * dy), ys, ye;
  int n, offset, x, y, noPoints = 0;
  const int maxPoints = 4 * (int)(fabs(sx - ex) + fabs(sy - ey));   /* the binary stops after this many pixels */
  dx = dx / vLen; dy = dy / vLen;
  vxTmp[0] = x1 - dy * width / 2.0; vyTmp[0] = y1 + dx * width / 2.0;
  vxTmp[1] = x2 - dy * width / 2.0; vyTmp[1] = y2 + dx * width / 2.0;
  vxTmp[2] = x2 + dy * width / 2.0; vyTmp[2] = y2 - dx * width / 2.0;
  vxTmp[3] = x1 + dy * width / 2.0; vyTmp[3] = y1 - dx * width / 2.0;
  if (x1 < x2 && y1 <= y2) offset = 0;
  else if (x1 >= x2 && y1 < y2) offset = 1;
  else if (x1 > x2 && y1 >= y2) offset = 2;
  else offset = 3;
  for (n = 0; n < 4; n++) { vx[n] = vxTmp[(offset + n) % 4]; vy[n] = vyTmp[(offset + n) % 4]; }
  x = (int)ceil(vx[0]) - 1;
  y = (int)ceil(vy[0]);
  ys = ye = -DBL_MAX;
  while (noPoints < maxPoints) {
    y++;
    while (y > ye && x <= vx[2]) {
      x++;
      if (x > vx[2]) break;
      if ((double)x < vx[3]) {
        if (fabs(vx[0] - vx[3]) <= 0.01) {
          if (vy[0] < vy[3]) ys = vy[0];
          else if (vy[0] > vy[3]) ys = vy[3];
          else ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
        } else ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
      } else {
        if (fabs(vx[3] - vx[2]) <= 0.01) {
          if (vy[3] < vy[2]) ys = vy[3];
          else if (vy[3] > vy[2]) ys = vy[2];
          else ys = vy[3] + (x - vx[3]) * (vy[2] - vy[3]) / (vx[2] - vx[3]);
        } else ys = vy[3] + (x - vx[3]) * (vy[2] - vy[3]) / (vx[2] - vx[3]);
      }
      if ((double)x < vx[1]) {
        if (fabs(vx[0] - vx[1]) <= 0.01) {
          if (vy[0] < vy[1]) ye = vy[1];
          else if (vy[0] > vy[1]) ye = vy[0];
          else ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
        } else ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
      } else {
        if (fabs(vx[1] - vx[2]) <= 0.01) {
          if (vy[1] < vy[2]) ye = vy[2];
          else if (vy[1] > vy[2]) ye = vy[1];
          else ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
        } else ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
      }
      y = (int)ceil(ys);
    }
    if (x > vx[2]) break;
    ptsx[noPoints] = x; ptsy[noPoints] = y; noPoints++;
  }
  *pn = noPoints;
}
static int e_validate_rect(const uint8_t *img, int w, int h, int *x, int *y, const e_line *ls, const e_nfalut *L) {
  const double lineAngle = e_line_angle(ls);
  int noPoints = 0, count = 0, aligned = 0, i;
  e_enumerate_rect_points(ls->sx, ls->sy, ls->ex, ls->ey, x, y, &noPoints);
  for (i = 0; i < noPoints; i++) {
    const int r = y[i], c = x[i];
    if (r <= 0 || r >= h - 1 || c <= 0 || c >= w - 1) continue;
    count++;
    if (e_aligned(img, w, r, c, lineAngle)) aligned++;
  }
  return e_check_nfa(L, count, aligned);
}

/* ---------------------------------------------------------------- SplitSegment2Lines */
typedef struct { e_line *v; int n, cap; } e_lines;
static void e_lines_push(e_lines *Ls, const e_line *l) {
  if (Ls->n == Ls->cap) { Ls->cap = Ls->cap ? 2 * Ls->cap : 256; Ls->v = (e_line *)realloc(Ls->v, sizeof(e_line) * (size_t)Ls->cap); }
  Ls->v[Ls->n++] = *l;
}
static void e_split_segment(const double *x, const double *y, int noPixels, int segmentNo, int min_line_len, e_lines *Ls) {
  int firstPixelIndex = 0;
  while (noPixels >= min_line_len) {
    int valid = 0, lastInvert = 0, index, len;
    double lastA = 0, lastB = 0, error = 0;
    while (noPixels >= min_line_len) {
      e_line_fit_err(x, y, min_line_len, &lastA, &lastB, &error, &lastInvert);
      if (error <= 0.5) { valid = 1; break; }
      noPixels -= 1; x += 1; y += 1; firstPixelIndex += 1;
    }
    if (!valid) return;
    index = min_line_len;
    len = min_line_len;
    while (index < noPixels) {
      int startIndex = index, lastGoodIndex = index - 1, goodPixelCount = 0, badPixelCount = 0;
      while (index < noPixels) {
        const ereal d = e_min_distance(x[index], y[index], lastA, lastB, lastInvert);
        if (d <= ED_LINE_ERROR) { lastGoodIndex = index; goodPixelCount++; badPixelCount = 0; }
        else { badPixelCount++; if (badPixelCount >= 5) break; }
        index++;
      }
      if (goodPixelCount >= 2) {
        len += lastGoodIndex - startIndex + 1;
        e_line_fit(x, y, len, &lastA, &lastB, lastInvert);
        index = lastGoodIndex + 1;
      }
      if (goodPixelCount < 2 || index >= noPixels) {
        e_line l;
        int i0 = 0, i1, noSkippedPixels;
        while (e_min_distance(x[i0], y[i0], lastA, lastB, lastInvert) > ED_LINE_ERROR) i0++;
        e_closest_point(x[i0], y[i0], lastA, lastB, lastInvert, &l.sx, &l.sy);
        noSkippedPixels = i0;
        i1 = lastGoodIndex;
        while (e_min_distance(x[i1], y[i1], lastA, lastB, lastInvert) > ED_LINE_ERROR) i1--;
        e_closest_point(x[i1], y[i1], lastA, lastB, lastInvert, &l.ex, &l.ey);
        l.a = lastA; l.b = lastB; l.invert = lastInvert; l.segmentNo = segmentNo;
        l.firstPixelIndex = firstPixelIndex + noSkippedPixels; l.len = i1 - noSkippedPixels + 1;
        e_lines_push(Ls, &l);
        len = i1 + 1;
        break;
      }
    }
    noPixels -= len; x += len; y += len; firstPixelIndex += len;
  }
}

/* DetectLinesByED(srcImg, width, height, &noLines): segments rows (sx, sy, ex, ey).  Optional debug outputs: smooth [h][w],
 * G [h][w] int16, D [h][w], E [h][w] (edge map after linking).  Returns the number of segments (rows beyond cap are dropped). */
int oracle_edlines(const uint8_t *img, int w, int h, double *segs, int cap, uint8_t *smooth_out, int16_t *G_out, uint8_t *D_out,
                   uint8_t *E_out) {
  const size_t NP = (size_t)w * h;
  uint8_t *S = (uint8_t *)malloc(NP), *D = (uint8_t *)malloc(NP), *E = (uint8_t *)malloc(NP);
  int16_t *G = (int16_t *)malloc(sizeof(int16_t) * NP);
  e_pix *pixels = (e_pix *)malloc(sizeof(e_pix) * NP), *segpix = (e_pix *)malloc(sizeof(e_pix) * NP);
  e_stack *stack = (e_stack *)malloc(sizeof(e_stack) * NP);
  e_chain *chains = (e_chain *)malloc(sizeof(e_chain) * NP);
  int *chainNos = (int *)malloc(sizeof(int) * (size_t)(w + h) * 8);
  e_segment *segments = (e_segment *)malloc(sizeof(e_segment) * (NP / ED_MIN_PATH + 16));
  int nsegments = 0, nsegpix = 0, *A, noAnchors, k, nout = 0;
  e_lines Ls = {0, 0, 0};
  oracle_ed_smooth(img, w, h, S);
  oracle_ed_gradient(S, w, h, G, D);
  oracle_ed_anchors(G, D, w, h, E);
  {   /* SortAnchorsByGradValue: counting sort, ascending; inside one gradient value the LATER anchor (raster order) gets the lower index */
    const int SIZE = 128 * 256;
    int *C = (int *)calloc((size_t)SIZE, sizeof(int)), i, j;
    for (i = 1; i < h - 1; i++) for (j = 1; j < w - 1; j++) if (E[(size_t)i * w + j] == ED_ANCHOR) C[G[(size_t)i * w + j]]++;
    for (i = 1; i < SIZE; i++) C[i] += C[i - 1];
    noAnchors = C[SIZE - 1];
    A = (int *)malloc(sizeof(int) * (size_t)(noAnchors > 0 ? noAnchors : 1));
    for (i = 1; i < h - 1; i++) for (j = 1; j < w - 1; j++) if (E[(size_t)i * w + j] == ED_ANCHOR) A[--C[G[(size_t)i * w + j]]] = i * w + j;
    free(C);
  }
  /* join the anchors, the one with the greatest gradient first */
  for (k = noAnchors - 1; k >= 0; k--) {
    const int i = A[k] / w, j = A[k] % w;
    int noChains = 1, len = 0, duplicatePixelCount = 0, top = -1;
    if (E[(size_t)i * w + j] != ED_ANCHOR) continue;
    chains[0].len = 0; chains[0].parent = -1; chains[0].dir = 0; chains[0].children[0] = chains[0].children[1] = -1; chains[0].pixels = 0;
    if (D[(size_t)i * w + j] == ED_VERTICAL) {
      stack[++top].r = i; stack[top].c = j; stack[top].dir = ED_DOWN; stack[top].parent = 0;
      stack[++top].r = i; stack[top].c = j; stack[top].dir = ED_UP; stack[top].parent = 0;
    } else {
      stack[++top].r = i; stack[top].c = j; stack[top].dir = ED_RIGHT; stack[top].parent = 0;
      stack[++top].r = i; stack[top].c = j; stack[top].dir = ED_LEFT; stack[top].parent = 0;
    }
    while (top >= 0) {
      int r = stack[top].r, c = stack[top].c, chainLen = 0, ended = 0;
      const int dir = stack[top].dir, parent = stack[top].parent;
      const int horizontal = (dir == ED_LEFT || dir == ED_RIGHT), step = (dir == ED_LEFT || dir == ED_UP) ? -1 : 1;
      const int child = (dir == ED_LEFT || dir == ED_UP) ? 0 : 1;
      top--;
      if (E[(size_t)r * w + c] != ED_EDGE) duplicatePixelCount++;
      chains[noChains].dir = dir; chains[noChains].parent = parent; chains[noChains].children[0] = chains[noChains].children[1] = -1;
      chains[noChains].pixels = &pixels[len];
      pixels[len].r = r; pixels[len].c = c; len++; chainLen++;
      while (D[(size_t)r * w + c] == (horizontal ? ED_HORIZONTAL : ED_VERTICAL)) {
        E[(size_t)r * w + c] = ED_EDGE;
        if (horizontal) {
          /* clean the anchors above and below, then look at the three pixels of the next column */
          if (E[(size_t)(r - 1) * w + c] == ED_ANCHOR) E[(size_t)(r - 1) * w + c] = 0;
          if (E[(size_t)(r + 1) * w + c] == ED_ANCHOR) E[(size_t)(r + 1) * w + c] = 0;
          /* (the diagonal on the side of the walking direction's sign is looked at first: up-left for LEFT, down-right for RIGHT) */
          if (E[(size_t)r * w + c + step] >= ED_ANCHOR) { c += step; }
          else if (E[(size_t)(r + step) * w + c + step] >= ED_ANCHOR) { r += step; c += step; }
          else if (E[(size_t)(r - step) * w + c + step] >= ED_ANCHOR) { r -= step; c += step; }
          else {
            const int Ag = G[(size_t)(r - 1) * w + c + step], Bg = G[(size_t)r * w + c + step], Cg = G[(size_t)(r + 1) * w + c + step];
            if (Ag > Bg) { if (Ag > Cg) r--; else r++; }
            else if (Cg > Bg) r++;
            c += step;
          }
        } else {
          if (E[(size_t)r * w + c - 1] == ED_ANCHOR) E[(size_t)r * w + c - 1] = 0;
          if (E[(size_t)r * w + c + 1] == ED_ANCHOR) E[(size_t)r * w + c + 1] = 0;
          if (E[(size_t)(r + step) * w + c] >= ED_ANCHOR) { r += step; }
          else if (E[(size_t)(r + step) * w + c + step] >= ED_ANCHOR) { r += step; c += step; }
          else if (E[(size_t)(r + step) * w + c - step] >= ED_ANCHOR) { r += step; c -= step; }
          else {
            const int Ag = G[(size_t)(r + step) * w + c - 1], Bg = G[(size_t)(r + step) * w + c], Cg = G[(size_t)(r + step) * w + c + 1];
            if (Ag > Bg) { if (Ag > Cg) c--; else c++; }
            else if (Cg > Bg) c++;
            r += step;
          }
        }
        if (E[(size_t)r * w + c] == ED_EDGE || G[(size_t)r * w + c] < ED_GRAD_THRESH) {
          if (chainLen > 0) { chains[noChains].len = chainLen; chains[parent].children[child] = noChains; noChains++; }
          ended = 1;
          break;
        }
        pixels[len].r = r; pixels[len].c = c; len++; chainLen++;
      }
      if (ended) continue;
      /* the direction of the edge changed: continue in the two perpendicular directions from here */
      if (horizontal) {
        stack[++top].r = r; stack[top].c = c; stack[top].dir = ED_DOWN; stack[top].parent = noChains;
        stack[++top].r = r; stack[top].c = c; stack[top].dir = ED_UP; stack[top].parent = noChains;
      } else {
        stack[++top].r = r; stack[top].c = c; stack[top].dir = ED_RIGHT; stack[top].parent = noChains;
        stack[++top].r = r; stack[top].c = c; stack[top].dir = ED_LEFT; stack[top].parent = noChains;
      }
      len--; chainLen--;
      chains[noChains].len = chainLen; chains[parent].children[child] = noChains; noChains++;
    }
    if (len - duplicatePixelCount < ED_MIN_PATH) {
      int q;
      for (q = 0; q < len; q++) E[(size_t)pixels[q].r * w + pixels[q].c] = 0;
    } else {
      e_pix *seg = segpix + nsegpix;
      int n = 0, totalLen, count, q, l;
      totalLen = e_longest_chain(chains, chains[0].children[1]);
      if (totalLen > 0) {
        count = e_retrieve_chain_nos(chains, chains[0].children[1], chainNos);
        for (q = count - 1; q >= 0; q--) {                 /* these chains backwards */
          const int cn = chainNos[q];
          int fr = chains[cn].pixels[chains[cn].len - 1].r, fc = chains[cn].pixels[chains[cn].len - 1].c, index = n - 2;
          while (index >= 0) {
            if (abs(fr - seg[index].r) <= 1 && abs(fc - seg[index].c) <= 1) { n--; index--; } else break;
          }
          if (chains[cn].len > 1 && n > 0) {
            fr = chains[cn].pixels[chains[cn].len - 2].r; fc = chains[cn].pixels[chains[cn].len - 2].c;
            if (abs(fr - seg[n - 1].r) <= 1 && abs(fc - seg[n - 1].c) <= 1) chains[cn].len--;
          }
          for (l = chains[cn].len - 1; l >= 0; l--) seg[n++] = chains[cn].pixels[l];
          chains[cn].len = 0;
        }
      }
      totalLen = e_longest_chain(chains, chains[0].children[0]);
      if (totalLen > 1) {
        count = e_retrieve_chain_nos(chains, chains[0].children[0], chainNos);
        chains[chainNos[0]].pixels++; chains[chainNos[0]].len--;   /* the anchor itself is already there */
        for (q = 0; q < count; q++) {
          const int cn = chainNos[q];
          int fr = chains[cn].pixels[0].r, fc = chains[cn].pixels[0].c, index = n - 2, startIndex = 0;
          while (index >= 0) {
            if (abs(fr - seg[index].r) <= 1 && abs(fc - seg[index].c) <= 1) { n--; index--; } else break;
          }
          if (chains[cn].len > 1 && n > 0) {
            fr = chains[cn].pixels[1].r; fc = chains[cn].pixels[1].c;
            if (abs(fr - seg[n - 1].r) <= 1 && abs(fc - seg[n - 1].c) <= 1) startIndex = 1;
          }
          for (l = startIndex; l < chains[cn].len; l++) seg[n++] = chains[cn].pixels[l];
          chains[cn].len = 0;
        }
      }
      if (n > 1 && abs(seg[1].r - seg[n - 1].r) <= 1 && abs(seg[1].c - seg[n - 1].c) <= 1) { seg++; n--; }   /* first pixel of a loop */
      segments[nsegments].pixels = seg; segments[nsegments].n = n; nsegments++;
      nsegpix = (int)(seg - segpix) + n;
      /* the rest of the tree: every remaining path of at least ten pixels is a segment of its own */
      for (q = 2; q < noChains; q++) {
        int qq;
        if (chains[q].len < 2) continue;
        totalLen = e_longest_chain(chains, q);
        if (totalLen < 10) continue;
        count = e_retrieve_chain_nos(chains, q, chainNos);
        seg = segpix + nsegpix; n = 0;
        for (qq = 0; qq < count; qq++) {
          const int cn = chainNos[qq];
          int fr = chains[cn].pixels[0].r, fc = chains[cn].pixels[0].c, index = n - 2, startIndex = 0;
          while (index >= 0) {
            if (abs(fr - seg[index].r) <= 1 && abs(fc - seg[index].c) <= 1) { n--; index--; } else break;
          }
          if (chains[cn].len > 1 && n > 0) {
            fr = chains[cn].pixels[1].r; fc = chains[cn].pixels[1].c;
            if (abs(fr - seg[n - 1].r) <= 1 && abs(fc - seg[n - 1].c) <= 1) startIndex = 1;
          }
          for (l = startIndex; l < chains[cn].len; l++) seg[n++] = chains[cn].pixels[l];
          chains[cn].len = 0;
        }
        segments[nsegments].pixels = seg; segments[nsegments].n = n; nsegments++;
        nsegpix += n;
      }
    }
  }
  /* ---- lines */
  {
    const int min_line_len = oracle_ed_min_line_len(w, h);
    double *x = (double *)malloc(sizeof(double) * (size_t)(NP > 16 ? NP : 16)), *y = (double *)malloc(sizeof(double) * (size_t)(NP > 16 ? NP : 16));
    int s, q;
    for (s = 0; s < nsegments; s++) {
      for (q = 0; q < segments[s].n; q++) { x[q] = (double)segments[s].pixels[q].c; y[q] = (double)segments[s].pixels[q].r; }
      e_split_segment(x, y, segments[s].n, s, min_line_len, &Ls);
    }
    free(x); free(y);
  }
  {   /* JoinCollinearLines: neighbours inside a segment, then the segment's first line with its last */
    int lastLineIndex = -1, i = 0;
    while (i < Ls.n) {
      const int segmentNo = Ls.v[i].segmentNo;
      int firstLineIndex, count = 1, j;
      lastLineIndex++;
      if (lastLineIndex != i) Ls.v[lastLineIndex] = Ls.v[i];
      firstLineIndex = lastLineIndex;
      for (j = i + 1; j < Ls.n; j++) {
        if (Ls.v[j].segmentNo != segmentNo) break;
        if (!e_try_to_join(&Ls.v[lastLineIndex], &Ls.v[j], ED_MAX_DIST, ED_MAX_ERROR)) {
          lastLineIndex++;
          if (lastLineIndex != j) Ls.v[lastLineIndex] = Ls.v[j];
        }
        count++;
      }
      if (firstLineIndex != lastLineIndex) {
        if (e_try_to_join(&Ls.v[firstLineIndex], &Ls.v[lastLineIndex], ED_MAX_DIST, ED_MAX_ERROR)) lastLineIndex--;
      }
      i += count;
    }
    Ls.n = lastLineIndex + 1;
  }
  {   /* ValidateLineSegments */
    e_nfalut L;
    int *rx = (int *)malloc(sizeof(int) * (size_t)(w + h) * 8), *ry = rx + (size_t)(w + h) * 4, i;
    e_nfalut_init(&L, (w + h) / 8, 0.125, 2.0 * (E_LOG10((double)w) + E_LOG10((double)h)));
    for (i = 0; i < Ls.n; i++) {
      const e_line *ls = &Ls.v[i];
      int valid;
      if (ls->len >= 80) valid = 1;
      else if (ls->len <= 25) valid = e_validate_rect(img, w, h, rx, ry, ls, &L);
      else {
        const double lineAngle = e_line_angle(ls);
        const e_pix *px = segments[ls->segmentNo].pixels + ls->firstPixelIndex;
        int aligned = 0, count = 0, q;
        for (q = 0; q < ls->len; q++) {
          const int r = px[q].r, c = px[q].c;
          if (r <= 0 || r >= h - 1 || c <= 0 || c >= w - 1) continue;
          count++;
          if (e_aligned(img, w, r, c, lineAngle)) aligned++;
        }
        valid = e_check_nfa(&L, count, aligned);
        if (!valid) valid = e_validate_rect(img, w, h, rx, ry, ls, &L);
      }
      if (valid) {
        if (nout < cap) { segs[4 * nout] = ls->sx; segs[4 * nout + 1] = ls->sy; segs[4 * nout + 2] = ls->ex; segs[4 * nout + 3] = ls->ey; }
        nout++;
      }
    }
    free(rx); free(L.lut);
  }
  if (smooth_out) memcpy(smooth_out, S, NP);
  if (G_out) memcpy(G_out, G, sizeof(int16_t) * NP);
  if (D_out) memcpy(D_out, D, NP);
  if (E_out) memcpy(E_out, E, NP);
  free(S); free(D); free(E); free(G); free(pixels); free(segpix); free(stack); free(chains); free(chainNos); free(segments); free(A); free(Ls.v);
  return nout;
}
