/* oracle/lsd_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the LSD line-segment detector exactly as the
 * reference runs it:  callLsd (src/line/utils.cpp:112-135) -> lsd() ->
 * lsd_scale() -> LineSegmentDetection() (external/lsd/lsd.cpp:1931-2065).
 * Every function cites the reference lines it follows.  The arithmetic (order
 * of operations, comparison directions, truncated literal constants) is kept;
 * the data structures are flat arrays instead of the reference's linked lists
 * and heap objects.
 *
 * Two build flavours (oracle/Makefile):
 *   liboracle_ref.so : transcendental functions from the host libm, like the
 *                      reference.  Pinned bit-for-bit (segments AND integer
 *                      region labels) against the reference's own lsd.c
 *                      compiled into oracle/_ref/liblsd_ref.so
 *                      (tests/test_oracle_lsd.py, tests/golden/).
 *   liboracle_lf.so  : -DORACLE_LFMATH; the functions that the HIP kernels
 *                      evaluate on the device (atan2, sin, cos, exp, log10,
 *                      pow) come from lineslam_amd/csrc/lf_math.h instead, so
 *                      the GPU result can be compared bit-for-bit.  Host-side
 *                      tables (Gaussian taps, log-gamma, log p) use libm in
 *                      both flavours, as the product's host code does.
 *
 * Build: gcc -O2 -ffp-contract=off (IEEE double, no FMA).
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>
#include <math.h>

#ifdef ORACLE_LFMATH
#include "../lineslam_amd/csrc/lf_math.h"
#define D_ATAN2(y, x) lf_atan2((y), (x))
#define D_SIN(x) lf_sin(x)
#define D_COS(x) lf_cos(x)
/* region2rect / get_theta use the correctly rounded atan2 / sin / cos of lf_math.h (the one place where LSD
 * depends on libm's last bit): get_theta keeps the double-double sine / cosine of its Newton start value and
 * region2rect derives sin / cos of the final theta from them (lf_sincos_cr_near), exactly as the kernel does. */
#define D_EXP(x) lf_exp(x)
#define D_LOG10(x) lf_log10(x)
#define D_POW(x, y) lf_pow((x), (y))
#else
#define D_ATAN2(y, x) atan2((y), (x))
#define D_SIN(x) sin(x)
#define D_COS(x) cos(x)

#define D_EXP(x) exp(x)
#define D_LOG10(x) log10(x)
#define D_POW(x, y) pow((x), (y))
#endif

/* literal constants of lsd.cpp:94-110 (note the truncated 3/2 pi and 2 pi) */
#define O_LN10 2.30258509299404568402
#define O_PI 3.14159265358979323846
#define O_NOTDEF (-1024.0)
#define O_3_2_PI 4.71238898038
#define O_2PI 6.28318530718
#define O_RELERR 100.0

typedef struct {
  double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
} orect; /* lsd.cpp:1075-1084 */

typedef struct {
  int X, Y;               /* size of the scaled image */
  const double *angles;   /* level-line angle or NOTDEF */
  const double *modgrad;
  unsigned char *used;
  int *regx, *regy;       /* region pixel list */
  /* statistics (not part of the result) */
  long n_grow, n_aligned_tests, n_rect_nfa, n_rect_pixels, n_accepted_px;
} octx;

/* lsd.cpp:147-165 */
static int o_double_equal(double a, double b) {
  double abs_diff, aa, bb, abs_max;
  if (a == b) return 1;
  abs_diff = fabs(a - b);
  aa = fabs(a);
  bb = fabs(b);
  abs_max = aa > bb ? aa : bb;
  if (abs_max < DBL_MIN) abs_max = DBL_MIN;
  return (abs_diff / abs_max) <= (O_RELERR * DBL_EPSILON);
}

/* lsd.cpp:170-173 */
static double o_dist(double x1, double y1, double x2, double y2) {
  return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
}

/* ---------------------------------------------------------------------------
 * Gaussian taps for one output coordinate (lsd.cpp:461-487 gaussian_kernel,
 * called from lsd.cpp:588,620).  Host libm in both flavours.                */
void oracle_gauss_taps(double sigma, double mean, int n, double *k) {
  double sum = 0.0;
  int i;
  for (i = 0; i < n; i++) {
    double val = ((double)i - mean) / sigma;
    k[i] = exp(-0.5 * val * val);
    sum += k[i];
  }
  if (sum >= 0.0)
    for (i = 0; i < n; i++) k[i] /= sum;
}

/* lsd.cpp:529-646 gaussian_sampler.  in: X*Y doubles; out: N*M (malloc'd).   */
static double *o_gaussian_sampler(const double *in, int X, int Y, double scale,
                                  double sigma_scale, int *Nout, int *Mout) {
  int N = (int)(unsigned int)floor(X * scale);
  int M = (int)(unsigned int)floor(Y * scale);
  double *aux = (double *)malloc(sizeof(double) * (size_t)N * Y);
  double *out = (double *)malloc(sizeof(double) * (size_t)N * M);
  double sigma = scale < 1.0 ? sigma_scale / scale : sigma_scale;
  double prec = 3.0;
  int h = (int)(unsigned int)ceil(sigma * sqrt(2.0 * prec * log(10.0)));
  int n = 1 + 2 * h;
  double *k = (double *)malloc(sizeof(double) * n);
  int dX = 2 * X, dY = 2 * Y;
  int x, y, i;
  for (x = 0; x < N; x++) {
    double xx = (double)x / scale;
    int xc = (int)floor(xx + 0.5);
    oracle_gauss_taps(sigma, (double)h + xx - (double)xc, n, k);
    for (y = 0; y < Y; y++) {
      double sum = 0.0;
      for (i = 0; i < n; i++) {
        int j = xc - h + i;
        while (j < 0) j += dX;
        while (j >= dX) j -= dX;
        if (j >= X) j = dX - 1 - j;
        sum += in[j + y * X] * k[i];
      }
      aux[x + y * N] = sum;
    }
  }
  for (y = 0; y < M; y++) {
    double yy = (double)y / scale;
    int yc = (int)floor(yy + 0.5);
    oracle_gauss_taps(sigma, (double)h + yy - (double)yc, n, k);
    for (x = 0; x < N; x++) {
      double sum = 0.0;
      for (i = 0; i < n; i++) {
        int j = yc - h + i;
        while (j < 0) j += dY;
        while (j >= dY) j -= dY;
        if (j >= Y) j = dY - 1 - j;
        sum += aux[x + j * N] * k[i];
      }
      out[x + y * N] = sum;
    }
  }
  free(k);
  free(aux);
  *Nout = N;
  *Mout = M;
  return out;
}

/* lsd.cpp:670-794 ll_angle.  Produces angles, modgrad and the pseudo-ordered
 * seed list as an array of pixel addresses (x + y*p), highest bin first, and
 * inside a bin in the reference's scan order (x outer, y inner, :723-724).    */
static void o_ll_angle(const double *in, int p, int n, double threshold, int n_bins,
                       double max_grad, double *g, double *modgrad, int *seeds,
                       int *n_seeds) {
  int x, y, i;
  int *bin_of = (int *)malloc(sizeof(int) * (size_t)p * n); /* -1: not listed */
  int *count = (int *)calloc((size_t)n_bins, sizeof(int));
  int *start = (int *)calloc((size_t)n_bins, sizeof(int));
  int top, total = 0;
  for (i = 0; i < p * n; i++) { bin_of[i] = -1; modgrad[i] = 0.0; g[i] = 0.0; }
  /* new_image_double does not initialise; the last row/col of modgrad are
     never read by the reference either (angles there are NOTDEF).          */
  for (x = 0; x < p; x++) g[(n - 1) * p + x] = O_NOTDEF;
  for (y = 0; y < n; y++) g[p * y + p - 1] = O_NOTDEF;
  for (x = 0; x < p - 1; x++)
    for (y = 0; y < n - 1; y++) {
      int adr = y * p + x;
      double com1 = in[adr + p + 1] - in[adr];
      double com2 = in[adr + 1] - in[adr + p];
      double gx = com1 + com2;
      double gy = com1 - com2;
      double norm2 = gx * gx + gy * gy;
      double norm = sqrt(norm2 / 4.0);
      modgrad[adr] = norm;
      if (norm <= threshold)
        g[adr] = O_NOTDEF;
      else {
        unsigned int b;
        g[adr] = D_ATAN2(gx, -gy);
        b = (unsigned int)(norm * (double)n_bins / max_grad);
        if (b >= (unsigned int)n_bins) b = (unsigned int)n_bins - 1;
        bin_of[adr] = (int)b;
        count[b]++;
      }
    }
  /* lsd.cpp:777-786: start at the highest non-empty bin (falls to bin 0 only
     if every other bin is empty), then append bins top-1 .. 1; bin 0 is never
     appended.                                                               */
  for (top = n_bins - 1; top > 0 && count[top] == 0; top--) ;
  if (count[top] != 0) {
    start[top] = 0;
    total = count[top];
    for (i = top - 1; i > 0; i--) { start[i] = total; total += count[i]; }
    if (top > 0) start[0] = -1; /* excluded */
  }
  {
    int *fill = (int *)calloc((size_t)n_bins, sizeof(int));
    for (x = 0; x < p - 1; x++)
      for (y = 0; y < n - 1; y++) {
        int adr = y * p + x, b = bin_of[adr];
        if (b < 0) continue;
        if (b == 0 && top > 0) continue;
        if (b > top) continue;
        seeds[start[b] + fill[b]++] = adr;
      }
    free(fill);
  }
  *n_seeds = total;
  free(bin_of);
  free(count);
  free(start);
}

/* lsd.cpp:799-832 isaligned (a == NOTDEF -> not aligned) */
static int o_isaligned(const octx *c, int x, int y, double theta, double prec) {
  double a = c->angles[x + y * c->X];
  if (a == O_NOTDEF) return 0;
  theta -= a;
  if (theta < 0.0) theta = -theta;
  if (theta > O_3_2_PI) {
    theta -= O_2PI;
    if (theta < 0.0) theta = -theta;
  }
  return theta < prec;
}

/* lsd.cpp:837-845 / 850-857 */
static double o_angle_diff(double a, double b) {
  a -= b;
  while (a <= -O_PI) a += O_2PI;
  while (a > O_PI) a -= O_2PI;
  if (a < 0.0) a = -a;
  return a;
}
static double o_angle_diff_signed(double a, double b) {
  a -= b;
  while (a <= -O_PI) a += O_2PI;
  while (a > O_PI) a -= O_2PI;
  return a;
}

/* lsd.cpp:886-902, 923-927, 934 : log-gamma (host libm in both flavours) */
static double o_log_gamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                              1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
  double b = 0.0;
  int n;
  for (n = 0; n < 7; n++) {
    a -= log(x + (double)n);
    b += q[n] * pow(x, (double)n);
  }
  return a + log(b);
}
static double o_log_gamma_windschitl(double x) {
  return 0.918938533204673 + (x - 0.5) * log(x) - x +
         0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
}
double oracle_log_gamma(double x) {
  return x > 15.0 ? o_log_gamma_windschitl(x) : o_log_gamma_lanczos(x);
}

/* lsd.cpp:980-1065 nfa */
static double o_nfa(int n, int k, double p, double logNT) {
  double tolerance = 0.1;
  double log1term, term, bin_term, mult_term, bin_tail, err, p_term;
  int i;
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * log10(p);
  p_term = p / (1.0 - p);
  log1term = oracle_log_gamma((double)n + 1.0) - oracle_log_gamma((double)k + 1.0) -
             oracle_log_gamma((double)(n - k) + 1.0) + (double)k * log(p) +
             (double)(n - k) * log(1.0 - p);
  term = D_EXP(log1term);
  if (o_double_equal(term, 0.0)) {
    if ((double)k > (double)n * p) return -log1term / O_LN10 - logNT;
    else return -logNT;
  }
  bin_tail = term;
  for (i = k + 1; i <= n; i++) {
    bin_term = (double)(n - i + 1) * (1.0 / (double)i); /* inv[] table == 1.0/i */
    mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      err = term * ((1.0 - D_POW(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * fabs(-D_LOG10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -D_LOG10(bin_tail) - logNT;
}

/* lsd.cpp:1183-1215 inter_low / inter_hi */
static double o_inter_low(double x, double x1, double y1, double x2, double y2) {
  if (o_double_equal(x1, x2) && y1 < y2) return y1;
  if (o_double_equal(x1, x2) && y1 > y2) return y2;
  return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}
static double o_inter_hi(double x, double x1, double y1, double x2, double y2) {
  if (o_double_equal(x1, x2) && y1 < y2) return y2;
  if (o_double_equal(x1, x2) && y1 > y2) return y1;
  return y1 + (x - x1) * (y2 - y1) / (x2 - x1);
}

/* lsd.cpp:1388-1410 rect_nfa with the rectangle iterator (ri_ini :1317-1383,
 * ri_inc :1247-1310, ri_end :1231-1240) unrolled into two nested loops: the
 * iterator visits columns x = ceil(vx[0]) .. while (double)x <= vx[2], and in a
 * column the pixels y = ceil(ys) .. while (double)y <= ye.                   */
static double o_rect_nfa(octx *c, const orect *r, double logNT) {
  double vx[4], vy[4], rx[4], ry[4];
  int offset, n, x, y, pts = 0, alg = 0;
  rx[0] = r->x1 - r->dy * r->width / 2.0;  ry[0] = r->y1 + r->dx * r->width / 2.0;
  rx[1] = r->x2 - r->dy * r->width / 2.0;  ry[1] = r->y2 + r->dx * r->width / 2.0;
  rx[2] = r->x2 + r->dy * r->width / 2.0;  ry[2] = r->y2 - r->dx * r->width / 2.0;
  rx[3] = r->x1 + r->dy * r->width / 2.0;  ry[3] = r->y1 - r->dx * r->width / 2.0;
  if (r->x1 < r->x2 && r->y1 <= r->y2) offset = 0;
  else if (r->x1 >= r->x2 && r->y1 < r->y2) offset = 1;
  else if (r->x1 > r->x2 && r->y1 >= r->y2) offset = 2;
  else offset = 3;
  for (n = 0; n < 4; n++) { vx[n] = rx[(offset + n) % 4]; vy[n] = ry[(offset + n) % 4]; }
  c->n_rect_nfa++;
  for (x = (int)ceil(vx[0]); !((double)x > vx[2]); x++) {
    double ys, ye;
    if ((double)x < vx[3]) ys = o_inter_low((double)x, vx[0], vy[0], vx[3], vy[3]);
    else ys = o_inter_low((double)x, vx[3], vy[3], vx[2], vy[2]);
    if ((double)x < vx[1]) ye = o_inter_hi((double)x, vx[0], vy[0], vx[1], vy[1]);
    else ye = o_inter_hi((double)x, vx[1], vy[1], vx[2], vy[2]);
    for (y = (int)ceil(ys); !((double)y > ye); y++) {
      c->n_rect_pixels++;
      if (x >= 0 && y >= 0 && x < c->X && y < c->Y) {
        ++pts;
        if (o_isaligned(c, x, y, r->theta, r->prec)) ++alg;
      }
    }
  }
  return o_nfa(pts, alg, r->p, logNT);
}

/* Optional trace of the three libm calls of region2rect / get_theta (the only place where LSD's output depends on
 * libm's last bit): rows of 6 doubles {atan2 y, atan2 x, atan2 result, theta after the +pi flip, cos, sin}.
 * tests/test_oracle_lsd.py compares them with correctly rounded values to name the calls the host libm misrounds.
 * Not thread-safe; NULL (the default) = off. */
static double *g_theta_trace = 0;
static int g_theta_trace_cap = 0, g_theta_trace_n = 0;
void oracle_lsd_theta_trace(double *buf, int cap) { g_theta_trace = buf; g_theta_trace_cap = cap; g_theta_trace_n = 0; }
int oracle_lsd_theta_trace_count(void) { return g_theta_trace_n; }

/* lsd.cpp:1474-1512 get_theta */
#ifdef ORACLE_LFMATH
typedef struct { double th, t0; int flipped; lf_dd s0, c0; } o_theta_aux;
#else
typedef struct { int unused; } o_theta_aux;
#endif
static double o_get_theta(const octx *c, int reg_size, double x, double y, double reg_angle,
                          double prec, o_theta_aux *aux) {
  double lambda, theta, weight, Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
  int i;
  for (i = 0; i < reg_size; i++) {
    weight = c->modgrad[c->regx[i] + c->regy[i] * c->X];
    Ixx += ((double)c->regy[i] - y) * ((double)c->regy[i] - y) * weight;
    Iyy += ((double)c->regx[i] - x) * ((double)c->regx[i] - x) * weight;
    Ixy -= ((double)c->regx[i] - x) * ((double)c->regy[i] - y) * weight;
  }
  /* the reference aborts on a null inertia matrix (:1496-1497); cannot occur
     for regions of >= 2 pixels with positive weights                        */
  lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
#ifdef ORACLE_LFMATH
  theta = fabs(Ixx) > fabs(Iyy) ? lf_atan2_cr_sc(lambda - Ixx, Ixy, &aux->t0, &aux->s0, &aux->c0)
                                : lf_atan2_cr_sc(Ixy, lambda - Iyy, &aux->t0, &aux->s0, &aux->c0);
  aux->th = theta; aux->flipped = 0;
  if (o_angle_diff(theta, reg_angle) > prec) { theta += O_PI; aux->flipped = 1; }
#else
  (void)aux;
  theta = fabs(Ixx) > fabs(Iyy) ? atan2(lambda - Ixx, Ixy) : atan2(Ixy, lambda - Iyy);
  if (g_theta_trace && g_theta_trace_n < g_theta_trace_cap) {
    double *t = g_theta_trace + 6 * g_theta_trace_n;
    t[0] = fabs(Ixx) > fabs(Iyy) ? lambda - Ixx : Ixy;
    t[1] = fabs(Ixx) > fabs(Iyy) ? Ixy : lambda - Iyy;
    t[2] = theta;
  }
  if (o_angle_diff(theta, reg_angle) > prec) theta += O_PI;
#endif
  return theta;
}

/* lsd.cpp:1517-1604 region2rect */
static void o_region2rect(const octx *c, int reg_size, double reg_angle, double prec, double p,
                          orect *rec) {
  double x, y, dx, dy, l, w, theta, weight, sum, l_min, l_max, w_min, w_max;
  int i;
  x = y = sum = 0.0;
  for (i = 0; i < reg_size; i++) {
    weight = c->modgrad[c->regx[i] + c->regy[i] * c->X];
    x += (double)c->regx[i] * weight;
    y += (double)c->regy[i] * weight;
    sum += weight;
  }
  x /= sum;
  y /= sum;
  o_theta_aux aux;
  theta = o_get_theta(c, reg_size, x, y, reg_angle, prec, &aux);
#ifdef ORACLE_LFMATH
  lf_sincos_cr_near(theta, aux.flipped, aux.th, aux.t0, aux.s0, aux.c0, &dy, &dx);
#else
  dx = cos(theta);
  dy = sin(theta);
  if (g_theta_trace && g_theta_trace_n < g_theta_trace_cap) {
    double *t = g_theta_trace + 6 * g_theta_trace_n++;
    t[3] = theta; t[4] = dx; t[5] = dy;
  }
#endif
  l_min = l_max = w_min = w_max = 0.0;
  for (i = 0; i < reg_size; i++) {
    l = ((double)c->regx[i] - x) * dx + ((double)c->regy[i] - y) * dy;
    w = -((double)c->regx[i] - x) * dy + ((double)c->regy[i] - y) * dx;
    if (l > l_max) l_max = l;
    if (l < l_min) l_min = l;
    if (w > w_max) w_max = w;
    if (w < w_min) w_min = w;
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
  if (rec->width < 1.0) rec->width = 1.0;
}

/* lsd.cpp:1610-1656 region_grow */
static void o_region_grow(octx *c, int x, int y, int *reg_size, double *reg_angle, double prec) {
  double sumdx, sumdy;
  int xx, yy, i;
  c->n_grow++;
  *reg_size = 1;
  c->regx[0] = x;
  c->regy[0] = y;
  *reg_angle = c->angles[x + y * c->X];
  sumdx = D_COS(*reg_angle);
  sumdy = D_SIN(*reg_angle);
  c->used[x + y * c->X] = 1;
  for (i = 0; i < *reg_size; i++)
    for (xx = c->regx[i] - 1; xx <= c->regx[i] + 1; xx++)
      for (yy = c->regy[i] - 1; yy <= c->regy[i] + 1; yy++)
        if (xx >= 0 && yy >= 0 && xx < c->X && yy < c->Y && c->used[xx + yy * c->X] != 1) {
          c->n_aligned_tests++;
          if (o_isaligned(c, xx, yy, *reg_angle, prec)) {
            c->used[xx + yy * c->X] = 1;
            c->regx[*reg_size] = xx;
            c->regy[*reg_size] = yy;
            ++(*reg_size);
            c->n_accepted_px++;
            sumdx += D_COS(c->angles[xx + yy * c->X]);
            sumdy += D_SIN(c->angles[xx + yy * c->X]);
            *reg_angle = D_ATAN2(sumdy, sumdx);
          }
        }
}

/* lsd.cpp:1662-1768 rect_improve */
static double o_rect_improve(octx *c, orect *rec, double logNT, double eps) {
  orect r;
  double log_nfa, log_nfa_new, delta = 0.5, delta_2 = delta / 2.0;
  int n;
  log_nfa = o_rect_nfa(c, rec, logNT);
  if (log_nfa > eps) return log_nfa;
  r = *rec; /* finer precisions */
  for (n = 0; n < 5; n++) {
    r.p /= 2.0;
    r.prec = r.p * O_PI;
    log_nfa_new = o_rect_nfa(c, &r, logNT);
    if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; *rec = r; }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec; /* reduce width */
  for (n = 0; n < 5; n++) {
    if ((r.width - delta) >= 0.5) {
      r.width -= delta;
      log_nfa_new = o_rect_nfa(c, &r, logNT);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec; /* reduce one side */
  for (n = 0; n < 5; n++) {
    if ((r.width - delta) >= 0.5) {
      r.x1 += -r.dy * delta_2;
      r.y1 += r.dx * delta_2;
      r.x2 += -r.dy * delta_2;
      r.y2 += r.dx * delta_2;
      r.width -= delta;
      log_nfa_new = o_rect_nfa(c, &r, logNT);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec; /* reduce the other side */
  for (n = 0; n < 5; n++) {
    if ((r.width - delta) >= 0.5) {
      r.x1 -= -r.dy * delta_2;
      r.y1 -= r.dx * delta_2;
      r.x2 -= -r.dy * delta_2;
      r.y2 -= r.dx * delta_2;
      r.width -= delta;
      log_nfa_new = o_rect_nfa(c, &r, logNT);
      if (log_nfa_new > log_nfa) { *rec = r; log_nfa = log_nfa_new; }
    }
  }
  if (log_nfa > eps) return log_nfa;
  r = *rec; /* even finer precisions */
  for (n = 0; n < 5; n++) {
    r.p /= 2.0;
    r.prec = r.p * O_PI;
    log_nfa_new = o_rect_nfa(c, &r, logNT);
    if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; *rec = r; }
  }
  return log_nfa;
}

/* lsd.cpp:1775-1841 reduce_region_radius */
static int o_reduce_region_radius(octx *c, int *reg_size, double reg_angle, double prec,
                                  double p, orect *rec, double density_th) {
  double density, rad1, rad2, rad, xc, yc;
  int i;
  density = (double)*reg_size / (o_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density >= density_th) return 1;
  xc = (double)c->regx[0];
  yc = (double)c->regy[0];
  rad1 = o_dist(xc, yc, rec->x1, rec->y1);
  rad2 = o_dist(xc, yc, rec->x2, rec->y2);
  rad = rad1 > rad2 ? rad1 : rad2;
  while (density < density_th) {
    rad *= 0.75;
    for (i = 0; i < *reg_size; i++)
      if (o_dist(xc, yc, (double)c->regx[i], (double)c->regy[i]) > rad) {
        c->used[c->regx[i] + c->regy[i] * c->X] = 0;
        c->regx[i] = c->regx[*reg_size - 1];
        c->regy[i] = c->regy[*reg_size - 1];
        --(*reg_size);
        --i;
      }
    if (*reg_size < 2) return 0;
    o_region2rect(c, *reg_size, reg_angle, prec, p, rec);
    density = (double)*reg_size / (o_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  }
  return 1;
}

/* lsd.cpp:1853-1921 refine */
static int o_refine(octx *c, int *reg_size, double reg_angle, double prec, double p, orect *rec,
                    double density_th) {
  double angle, ang_d, mean_angle, tau, density, xc, yc, ang_c, sum, s_sum;
  int i, n;
  density = (double)*reg_size / (o_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density >= density_th) return 1;
  xc = (double)c->regx[0];
  yc = (double)c->regy[0];
  ang_c = c->angles[c->regx[0] + c->regy[0] * c->X];
  sum = s_sum = 0.0;
  n = 0;
  for (i = 0; i < *reg_size; i++) {
    c->used[c->regx[i] + c->regy[i] * c->X] = 0;
    if (o_dist(xc, yc, (double)c->regx[i], (double)c->regy[i]) < rec->width) {
      angle = c->angles[c->regx[i] + c->regy[i] * c->X];
      ang_d = o_angle_diff_signed(angle, ang_c);
      sum += ang_d;
      s_sum += ang_d * ang_d;
      ++n;
    }
  }
  mean_angle = sum / (double)n;
  tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
  o_region_grow(c, c->regx[0], c->regy[0], reg_size, &reg_angle, tau);
  if (*reg_size < 2) return 0;
  o_region2rect(c, *reg_size, reg_angle, prec, p, rec);
  density = (double)*reg_size / (o_dist(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
  if (density < density_th)
    return o_reduce_region_radius(c, reg_size, reg_angle, prec, p, rec, density_th);
  return 1;
}

/* ---------------------------------------------------------------------------
 * LineSegmentDetection (lsd.cpp:1931-2065).
 *   img      : X*Y doubles, grey levels in [0,255]
 *   segs     : out, up to cap rows of {x1,y1,x2,y2,width}
 *   labels   : out or NULL, N*M int32 region labels (0 = none, k = k-th segment)
 *   dbg_*    : optional intermediate products (NULL to skip)
 * returns the number of segments (may exceed cap; only cap rows are written).
 */
int oracle_lsd(const double *img, int X, int Y, double scale, double sigma_scale, double quant,
               double ang_th, double eps, double density_th, int n_bins, double max_grad,
               double *segs, int cap, int32_t *labels, int *Nout, int *Mout,
               double *dbg_scaled, double *dbg_angles, double *dbg_modgrad, int32_t *dbg_seeds,
               int *dbg_nseeds, long *stats /* 5 longs or NULL */) {
  double prec = O_PI * ang_th / 180.0;
  double p = ang_th / 180.0;
  double rho = quant / sin(prec); /* host libm in both flavours (lsd.cpp:1965) */
  int N, M, n_seeds = 0, s, i, ls_count = 0, reg_size, min_reg_size;
  double *scaled, *angles, *modgrad, logNT, reg_angle, log_nfa;
  int *seeds;
  octx c;
  orect rec;
  if (scale != 1.0)
    scaled = o_gaussian_sampler(img, X, Y, scale, sigma_scale, &N, &M);
  else {
    N = X; M = Y;
    scaled = (double *)malloc(sizeof(double) * (size_t)N * M);
    memcpy(scaled, img, sizeof(double) * (size_t)N * M);
  }
  angles = (double *)malloc(sizeof(double) * (size_t)N * M);
  modgrad = (double *)malloc(sizeof(double) * (size_t)N * M);
  seeds = (int *)malloc(sizeof(int) * (size_t)N * M);
  o_ll_angle(scaled, N, M, rho, n_bins, max_grad, angles, modgrad, seeds, &n_seeds);
  logNT = 5.0 * (log10((double)N) + log10((double)M)) / 2.0;
  min_reg_size = (int)(-logNT / log10(p));
  memset(&c, 0, sizeof(c));
  c.X = N; c.Y = M; c.angles = angles; c.modgrad = modgrad;
  c.used = (unsigned char *)calloc((size_t)N * M, 1);
  c.regx = (int *)malloc(sizeof(int) * (size_t)N * M);
  c.regy = (int *)malloc(sizeof(int) * (size_t)N * M);
  if (labels) memset(labels, 0, sizeof(int32_t) * (size_t)N * M);
  for (s = 0; s < n_seeds; s++) {
    int sx = seeds[s] % N, sy = seeds[s] / N;
    if (c.used[seeds[s]] != 0 || angles[seeds[s]] == O_NOTDEF) continue;
    o_region_grow(&c, sx, sy, &reg_size, &reg_angle, prec);
    if (reg_size < min_reg_size) continue;
    o_region2rect(&c, reg_size, reg_angle, prec, p, &rec);
    if (!o_refine(&c, &reg_size, reg_angle, prec, p, &rec, density_th)) continue;
    log_nfa = o_rect_improve(&c, &rec, logNT, eps);
    if (log_nfa <= eps) continue;
    ++ls_count;
    rec.x1 += 0.5; rec.y1 += 0.5;
    rec.x2 += 0.5; rec.y2 += 0.5;
    if (scale != 1.0) {
      rec.x1 /= scale; rec.y1 /= scale;
      rec.x2 /= scale; rec.y2 /= scale;
      rec.width /= scale;
    }
    if (ls_count <= cap) {
      double *o = segs + 5 * (size_t)(ls_count - 1);
      o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width;
    }
    if (labels)
      for (i = 0; i < reg_size; i++) labels[c.regx[i] + c.regy[i] * N] = ls_count;
  }
  if (Nout) *Nout = N;
  if (Mout) *Mout = M;
  if (dbg_scaled) memcpy(dbg_scaled, scaled, sizeof(double) * (size_t)N * M);
  if (dbg_angles) memcpy(dbg_angles, angles, sizeof(double) * (size_t)N * M);
  if (dbg_modgrad) memcpy(dbg_modgrad, modgrad, sizeof(double) * (size_t)N * M);
  if (dbg_seeds) for (i = 0; i < n_seeds; i++) dbg_seeds[i] = seeds[i];
  if (dbg_nseeds) *dbg_nseeds = n_seeds;
  if (stats) {
    stats[0] = c.n_grow; stats[1] = c.n_aligned_tests; stats[2] = c.n_rect_nfa;
    stats[3] = c.n_rect_pixels; stats[4] = c.n_accepted_px;
  }
  free(scaled); free(angles); free(modgrad); free(seeds);
  free(c.used); free(c.regx); free(c.regy);
  return ls_count;
}

/* callLsd (src/line/utils.cpp:112-135) + lsd()/lsd_scale() fixed parameters
 * (lsd.cpp:2070-2100): u8 grey -> double, scale 0.8, sigma_scale 0.6, quant 2,
 * eps 0, 1024 bins, max_grad 255; ang_th / density_th come from sysPara.       */
int oracle_call_lsd_u8(const uint8_t *gray, int stride, int w, int h, double ang_th,
                       double density_th, double *segs, int cap, int32_t *labels, int *Nout,
                       int *Mout) {
  double *img = (double *)malloc(sizeof(double) * (size_t)w * h);
  int x, y, n;
  for (x = 0; x < w; ++x)
    for (y = 0; y < h; ++y) img[x + y * w] = (double)gray[x + (size_t)y * stride];
  n = oracle_lsd(img, w, h, 0.8, 0.6, 2.0, ang_th, 0.0, density_th, 1024, 255.0, segs, cap,
                 labels, Nout, Mout, NULL, NULL, NULL, NULL, NULL, NULL);
  free(img);
  return n;
}
