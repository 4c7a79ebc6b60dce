/* lf_math.h -- deterministic fp64 elementary functions for the line front end.
 *
 * Why this exists: LSD (reference external/lsd/lsd.cpp) calls atan2/sin/cos/exp/
 * log10/pow from the host libm.  glibc (CPU) and OCML (gfx950) disagree in the
 * last ulp, so neither can be used on both sides of a parity test.  These
 * functions use ONLY IEEE-754 +,-,*,/ and sqrt on doubles (no fma, no table
 * look-ups, no libm), so the same source compiled by gcc (-ffp-contract=off)
 * and by hipcc (--offload-arch=gfx950 -ffp-contract=off) returns bit-identical
 * results.  Accuracy, measured against the x87 long-double libm (tests/test_lf_math.py):
 * exp / log / sin / cos < 0.9 ulp, atan2 < 1.4, log10 < 1.8 (glibc's own: 1.6), acos < 2.5;
 * the *_cr variants are correctly rounded (0.500 ulp).  Far below anything that can flip an
 * LSD decision except at the one place the *_cr variants exist for.
 *
 * Polynomial coefficients are the classical minimax sets published with
 * FreeBSD msun / fdlibm (Sun Microsystems, "freely granted" licence); the
 * argument reductions below are written for this project (single-division
 * atan2, double-double pi/2 reduction).
 *
 * This header is product code.  oracle/ includes it only for its
 * "lfmath" build flavour (see oracle/README in DESIGN.md section 3).
 */
#ifndef LF_MATH_H
#define LF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LF_HD __host__ __device__ static inline
#else
#define LF_HD static inline
#endif

#define LF_PI      3.14159265358979323846   /* M_PI, as lsd.cpp:99-101 */
#define LF_PI_LO   1.2246467991473532e-16   /* pi - (double)pi */
#define LF_PIO2    1.5707963267948966
#define LF_LN10    2.30258509299404568402   /* M_LN10, lsd.cpp:94-96 */

LF_HD uint64_t lf_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
LF_HD double lf_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
LF_HD double lf_fabs(double x) { return lf_from_bits(lf_bits(x) & 0x7fffffffffffffffULL); }
LF_HD double lf_copysign(double m, double s) {
  return lf_from_bits((lf_bits(m) & 0x7fffffffffffffffULL) | (lf_bits(s) & 0x8000000000000000ULL));
}
LF_HD double lf_sqrt(double x) { return __builtin_sqrt(x); }  /* IEEE correctly rounded on both sides */
LF_HD double lf_pow2i(int k) { return lf_from_bits((uint64_t)(k + 1023) << 52); } /* -1022<=k<=1023 */

/* ------------------------------------------------------------------ atan2 */
/* atan(t) for the reduced argument t, |t| <= 7/16 (approximately).          */
LF_HD double lf_atan_poly(double t) {
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  double z = t * t, w = z * z;
  double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  return t * (s1 + s2); /* atan(t) = t - this */
}

/* atan2(y,x) with one division.  With u = |y|/|x| the breakpoints are those
 * of the classical 4-interval scheme (7/16, 11/16, 19/16, 39/16) but the
 * reduced argument (u-c)/(1+c u) is formed as (|y| - c|x|)/(|x| + c|y|). */
LF_HD double lf_atan2(double y, double x) {
  if (x != x || y != y) return x + y;
  double ax = lf_fabs(x), ay = lf_fabs(y);
  int xneg = (int)(lf_bits(x) >> 63);
  if (ay == 0.0) return xneg ? lf_copysign(LF_PI, y) : y;
  if (ax == 0.0) return lf_copysign(LF_PIO2, y);
  /* bring the larger magnitude near 1 so products cannot over/underflow */
  {
    double m = ax > ay ? ax : ay;
    int e = (int)((lf_bits(m) >> 52) & 0x7ff) - 1023;
    if (e > 500 || e < -500) {
      if (e == 1024) { /* infinities */
        ax = (ax > 1.7e308) ? 1.0 : 0.0;
        ay = (ay > 1.7e308) ? 1.0 : 0.0;
      } else {
        int h = -e / 2; double s = lf_pow2i(h);
        ax = ax * s * s; ay = ay * s * s; /* exact unless a denormal operand is lost: irrelevant at ratio 2^-500 */
        if (e & 1) { double s2 = lf_pow2i(-e - 2 * h); ax *= s2; ay *= s2; }
      }
      if (ax == 0.0) return lf_copysign(LF_PIO2, y);
      if (ay == 0.0 && !xneg) return lf_copysign(0.0, y);
    }
  }
  double num, den, hi, lo;
  if (16.0 * ay < 7.0 * ax)        { num = ay;              den = ax;              hi = 0.0;                    lo = 0.0; }
  else if (16.0 * ay < 11.0 * ax)  { num = 2.0 * ay - ax;   den = 2.0 * ax + ay;   hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (16.0 * ay < 19.0 * ax)  { num = ay - ax;         den = ax + ay;         hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (16.0 * ay < 39.0 * ax)  { num = ay - 1.5 * ax;   den = ax + 1.5 * ay;   hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else                             { num = -ax;             den = ay;              hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  double t = num / den;
  double z = hi - ((lf_atan_poly(t) - lo) - t);   /* atan(|y/x|) in [0, pi/2] */
  if (xneg) z = LF_PI - (z - LF_PI_LO);
  return lf_copysign(z, y);
}

/* acos(x) = 2 atan2(sqrt(1-x), sqrt(1+x)): 1-x and 1+x are exact or correctly rounded, so the result stays
 * within a few ulp on the whole of [-1,1] (also next to +-1); NaN outside, as libm. */
LF_HD double lf_acos(double x) {
  if (!(x >= -1.0 && x <= 1.0)) return lf_from_bits(0x7ff8000000000000ULL);
  return 2.0 * lf_atan2(lf_sqrt(1.0 - x), lf_sqrt(1.0 + x));
}

/* ---------------------------------------------------------------- sin/cos */
LF_HD double lf_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x, v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
LF_HD double lf_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x, w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z, u = 1.0 - hz;
  return u + (((1.0 - u) - hz) + (z * r - x * y));
}
/* sin and cos of x, valid for |x| < 2^20 (the front end needs |x| <= 2 pi). */
LF_HD void lf_sincos(double x, double *s, double *c) {
  const double INVPIO2 = 6.36619772367581382433e-01;
  const double P1 = 1.5707963267341256, P2 = 6.077100506303966e-11, P3 = 2.0222662487959506e-21;
  double q = x * INVPIO2;
  int n = (int)(q + (q < 0.0 ? -0.5 : 0.5));
  double fn = (double)n;
  double r0 = x - fn * P1;            /* exact: P1 has 33 significant bits */
  double w1 = fn * P2;                /* exact */
  double r1 = r0 - w1;                /* TwoDiff */
  double bb = r1 - r0;
  double e1 = (r0 - (r1 - bb)) - (w1 + bb);
  double w2 = fn * P3;
  double yh = r1 - w2;
  double yl = ((r1 - yh) - w2) + e1;
  double ks = lf_ksin(yh, yl), kc = lf_kcos(yh, yl);
  switch (n & 3) {
    case 0:  *s = ks;  *c = kc;  break;
    case 1:  *s = kc;  *c = -ks; break;
    case 2:  *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks;  break;
  }
}
LF_HD double lf_sin(double x) { double s, c; lf_sincos(x, &s, &c); return s; }
LF_HD double lf_cos(double x) { double s, c; lf_sincos(x, &s, &c); return c; }

/* ------------------------------------------- correctly rounded sin/cos/atan2 (double-double)
 * Used where LSD is sensitive to the last bit of libm (region2rect's rectangle angle and axes: every rectangle's
 * end pixel lies exactly on its end edge).  Error-free transformations without FMA (Dekker / Veltkamp); the results
 * are accurate to ~2^-100 before the final rounding, i.e. correctly rounded except for inputs whose exact value
 * lies within 2^-47 ulp of a rounding boundary.  glibc's sin/cos/atan2 are (almost always) correctly rounded too,
 * so these agree with the reference's libm where the 1-ulp versions above do not. */
typedef struct { double hi, lo; } lf_dd;
LF_HD lf_dd lf_dd_make(double hi, double lo) { lf_dd r; r.hi = hi; r.lo = lo; return r; }
LF_HD lf_dd lf_two_sum(double a, double b) { double s = a + b, bb = s - a; return lf_dd_make(s, (a - (s - bb)) + (b - bb)); }
LF_HD lf_dd lf_quick_two_sum(double a, double b) { double s = a + b; return lf_dd_make(s, b - (s - a)); }
LF_HD lf_dd lf_two_prod(double a, double b) {
  double p = a * b, ca = 134217729.0 * a, cb = 134217729.0 * b;
  double ah = ca - (ca - a), al = a - ah, bh = cb - (cb - b), bl = b - bh;
  return lf_dd_make(p, ((ah * bh - p) + ah * bl + al * bh) + al * bl);
}
LF_HD lf_dd lf_dd_add(lf_dd a, lf_dd b) {
  lf_dd s = lf_two_sum(a.hi, b.hi), t = lf_two_sum(a.lo, b.lo);
  s.lo += t.hi; s = lf_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo; return lf_quick_two_sum(s.hi, s.lo);
}
LF_HD lf_dd lf_dd_neg(lf_dd a) { return lf_dd_make(-a.hi, -a.lo); }
LF_HD lf_dd lf_dd_mul(lf_dd a, lf_dd b) {
  lf_dd p = lf_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return lf_quick_two_sum(p.hi, p.lo);
}
LF_HD lf_dd lf_dd_mul_d(lf_dd a, double b) {
  lf_dd p = lf_two_prod(a.hi, b);
  p.lo += a.lo * b;
  return lf_quick_two_sum(p.hi, p.lo);
}
/* sin and cos of the DOUBLE x as double-doubles, |x| < 2^20, in three steps so that a wavefront can run the two
 * polynomial chains in two lanes at once (lf_lsd.hip) with exactly the arithmetic of the sequential version:
 *   reduce : r = x - n pi/2 (double-double), quadrant n
 *   poly   : Horner chain in z = r^2 of the sine (sel = 0) or cosine (sel = 1) series
 *   finish : sin r = r + r (ps z), cos r = 1 + pc z, quadrant rotation                                        */
LF_HD lf_dd lf_sincos_dd_reduce(double x, int *n_out) {
  const double INVPIO2 = 6.36619772367581382433e-01;
  const double P1 = 1.5707963267341256, P2 = 6.077100506303966e-11, P3 = 2.0222662487959506e-21, P4 = 1.0085854035872483e-37;
  double q = x * INVPIO2;
  int n = (int)(q + (q < 0.0 ? -0.5 : 0.5));
  double fn = (double)n;
  /* r = x - n (P1 + P2 + P3 + P4): n P1 and n P2 are exact (33-bit constants) */
  lf_dd r = lf_two_sum(x - fn * P1, -(fn * P2));
  lf_dd w = lf_two_prod(fn, P3);
  r = lf_dd_add(r, lf_dd_neg(w));
  r = lf_dd_add(r, lf_dd_make(-(fn * P4), 0.0));
  *n_out = n;
  return r;
}
LF_HD lf_dd lf_sincos_dd_poly(lf_dd z, int sel) {
  /* (-1)^k / (2k+1)!  and  (-1)^k / (2k)!,  k = 1..13, as double-doubles */
  static const double SC[13][2] = {
      {-0.16666666666666666, -9.25185853854297e-18},   {0.008333333333333333, 1.1564823173178714e-19},
      {-0.0001984126984126984, -1.7209558293420705e-22}, {2.7557319223985893e-06, -1.858393274046472e-22},
      {-2.505210838544172e-08, 1.448814070935912e-24},  {1.6059043836821613e-10, 1.2585294588752098e-26},
      {-7.647163731819816e-13, -7.03872877733453e-30},  {2.8114572543455206e-15, 1.6508842730861433e-31},
      {-8.22063524662433e-18, -2.2141894119604265e-34}, {1.9572941063391263e-20, -1.3643503830087908e-36},
      {-3.868170170630684e-23, 8.843177655482344e-40},  {6.446950284384474e-26, -1.9330404233703465e-42},
      {-9.183689863795546e-29, -1.4303150396787322e-45}};
  static const double CC[13][2] = {
      {-0.5, 0.0},                                      {0.041666666666666664, 2.3129646346357427e-18},
      {-0.001388888888888889, 5.300543954373577e-20},   {2.48015873015873e-05, 2.1511947866775882e-23},
      {-2.755731922398589e-07, -2.3767714622250297e-23}, {2.08767569878681e-09, -1.20734505911326e-25},
      {-1.1470745597729725e-11, -2.0655512752830745e-28}, {4.779477332387385e-14, 4.399205485834081e-31},
      {-1.5619206968586225e-16, -1.1910679660273754e-32}, {4.110317623312165e-19, 1.4412973378659527e-36},
      {-8.896791392450574e-22, 7.911402614872376e-38},  {1.6117375710961184e-24, -3.6846573564509766e-41},
      {-2.4795962632247976e-27, 1.2953730964765229e-43}};
  lf_dd p = sel ? lf_dd_make(CC[12][0], CC[12][1]) : lf_dd_make(SC[12][0], SC[12][1]);
  int k;
  for (k = 11; k >= 0; k--) p = lf_dd_add(lf_dd_mul(p, z), sel ? lf_dd_make(CC[k][0], CC[k][1]) : lf_dd_make(SC[k][0], SC[k][1]));
  return p;
}
LF_HD void lf_sincos_dd_finish(lf_dd r, lf_dd z, lf_dd ps, lf_dd pc, int n, lf_dd *s, lf_dd *c) {
  lf_dd ks = lf_dd_add(r, lf_dd_mul(r, lf_dd_mul(ps, z)));
  lf_dd kc = lf_dd_add(lf_dd_make(1.0, 0.0), lf_dd_mul(pc, z));
  switch (n & 3) {
    case 0:  *s = ks;             *c = kc;             break;
    case 1:  *s = kc;             *c = lf_dd_neg(ks);  break;
    case 2:  *s = lf_dd_neg(ks);  *c = lf_dd_neg(kc);  break;
    default: *s = lf_dd_neg(kc);  *c = ks;             break;
  }
}
#if defined(LF_SINCOS_DD_LANES) && defined(__HIP_DEVICE_COMPILE__)
/* wavefront-uniform callers only (lf_lsd.hip): lane parity picks the series, the two chains run at once */
LF_HD double lf_bcast_lane(double v, int l) {
  int lo = __builtin_amdgcn_readlane((int)(lf_bits(v) & 0xffffffffu), l), hi = __builtin_amdgcn_readlane((int)(lf_bits(v) >> 32), l);
  return lf_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
LF_HD void lf_sincos_dd(double x, lf_dd *s, lf_dd *c) {
  int n;
  lf_dd r = lf_sincos_dd_reduce(x, &n);
  lf_dd z = lf_dd_mul(r, r);
  lf_dd p = lf_sincos_dd_poly(z, (int)(__builtin_amdgcn_mbcnt_lo(~0u, 0u) & 1u));
  lf_dd ps = lf_dd_make(lf_bcast_lane(p.hi, 0), lf_bcast_lane(p.lo, 0)), pc = lf_dd_make(lf_bcast_lane(p.hi, 1), lf_bcast_lane(p.lo, 1));
  lf_sincos_dd_finish(r, z, ps, pc, n, s, c);
}
#else
LF_HD void lf_sincos_dd(double x, lf_dd *s, lf_dd *c) {
  int n;
  lf_dd r = lf_sincos_dd_reduce(x, &n);
  lf_dd z = lf_dd_mul(r, r);
  lf_sincos_dd_finish(r, z, lf_sincos_dd_poly(z, 0), lf_sincos_dd_poly(z, 1), n, s, c);
}
#endif
#define LF_SINCOS_DD(x, s, c) lf_sincos_dd((x), (s), (c))
LF_HD void lf_sincos_cr(double x, double *s, double *c) {
  lf_dd sd, cd;
  LF_SINCOS_DD(x, &sd, &cd);
  *s = sd.hi + sd.lo;
  *c = cd.hi + cd.lo;
}
/* atan2: one Newton step on the 1-ulp result t0:  t = t0 + (y cos t0 - x sin t0) / (x cos t0 + y sin t0).
 * Optionally hands back t0 and its double-double sine / cosine for lf_sincos_cr_near below. */
LF_HD double lf_atan2_cr_sc(double y, double x, double *t0_out, lf_dd *s_out, lf_dd *c_out) {
  double t0 = lf_atan2(y, x);
  lf_dd s, c, num, den;
  *t0_out = t0;
  if (!(t0 == t0) || (x == 0.0 && y == 0.0) || !(lf_fabs(x) <= 1.7e308) || !(lf_fabs(y) <= 1.7e308)) {
    LF_SINCOS_DD(t0 == t0 ? t0 : 0.0, s_out, c_out);
    return t0;
  }
  {  /* scale so that the products below can neither overflow nor lose their low parts */
    double m = lf_fabs(x) > lf_fabs(y) ? lf_fabs(x) : lf_fabs(y);
    int e = (int)((lf_bits(m) >> 52) & 0x7ff) - 1023;
    if (e > 400 || e < -400) { double sc = lf_pow2i(e > 0 ? -400 : 400); x *= sc; y *= sc; if (e > 800 || e < -800) { x *= sc; y *= sc; } }
  }
  LF_SINCOS_DD(t0, &s, &c);
  *s_out = s; *c_out = c;
  num = lf_dd_add(lf_dd_mul_d(c, y), lf_dd_neg(lf_dd_mul_d(s, x)));
  den = lf_dd_add(lf_dd_mul_d(c, x), lf_dd_mul_d(s, y));
  return t0 + (num.hi + num.lo) / (den.hi + den.lo);
}
LF_HD double lf_atan2_cr(double y, double x) {
  double t0; lf_dd s, c;
  return lf_atan2_cr_sc(y, x, &t0, &s, &c);
}
/* Correctly rounded sin / cos of the double `theta`, where theta is either th (flipped = 0) or the rounded sum
 * th + LF_PI (flipped = 1) and th itself lies within a few ulp of t0, whose double-double sine s0 / cosine c0 are
 * known (lf_atan2_cr_sc): with D = theta - t0 - (flipped ? pi : 0) (|D| < 1e-15, formed exactly),
 * sin(theta) = +-(s0 + c0 D), cos(theta) = +-(c0 - s0 D); the neglected D^2/2 is below 2^-102.      */
LF_HD void lf_sincos_cr_near(double theta, int flipped, double th, double t0, lf_dd s0, lf_dd c0, double *s, double *c) {
  lf_dd D = lf_dd_make(th - t0, 0.0);            /* exact: neighbours in the same binade (or across one) */
  lf_dd sn, cn;
  if (flipped) {
    /* th + LF_PI = theta + err  and  LF_PI = pi - (PI_LO + PI_LO2):  theta = th + pi - (PI_LO + PI_LO2 + err) */
    lf_dd sum = lf_two_sum(th, LF_PI);           /* sum.hi == theta */
    lf_dd e = lf_two_sum(LF_PI_LO, sum.lo);
    e = lf_dd_add(e, lf_dd_make(-2.9947698097183397e-33, 0.0));
    D = lf_dd_add(D, lf_dd_neg(e));
    (void)theta;
  }
  sn = lf_dd_add(s0, lf_dd_mul(c0, D));
  cn = lf_dd_add(c0, lf_dd_neg(lf_dd_mul(s0, D)));
  if (flipped) { sn = lf_dd_neg(sn); cn = lf_dd_neg(cn); }
  *s = sn.hi + sn.lo;
  *c = cn.hi + cn.lo;
}

/* -------------------------------------------------------------------- exp */
LF_HD double lf_exp(double x) {
  const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10,
               INVLN2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return lf_from_bits(0x7ff0000000000000ULL);
  if (x < -745.2) return 0.0;
  double q = x * INVLN2;
  int k = (int)(q + (q < 0.0 ? -0.5 : 0.5));
  double fk = (double)k;
  double hi = x - fk * LN2HI, lo = fk * LN2LO;
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  int k1 = k / 2, k2 = k - k1;
  return (y * lf_pow2i(k1)) * lf_pow2i(k2);
}

/* -------------------------------------------------------------------- log */
/* returns k and f with x = 2^k (1+f), sqrt(2)/2 <= 1+f < sqrt(2) */
LF_HD double lf_log_reduce(double x, int *kout) {
  int k = 0;
  uint64_t u = lf_bits(x);
  if ((u >> 52) == 0) { x *= 18014398509481984.0; k -= 54; u = lf_bits(x); } /* subnormal */
  k += (int)(u >> 52) - 1023;
  u = (u & 0x000fffffffffffffULL);
  /* mantissa above sqrt(2) -> halve */
  if (u >= 0x6a09e667f3bcdULL) { k += 1; u |= 0x3fe0000000000000ULL; }
  else u |= 0x3ff0000000000000ULL;
  *kout = k;
  return lf_from_bits(u) - 1.0;
}
/* log(1+f) - f, i.e. the part after the leading term */
LF_HD double lf_log_tail(double f, double *hfsq_out) {
  const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
               L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
               L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
               L7 = 1.479819860511658591e-01;
  double s = f / (2.0 + f);
  double z = s * s, w = z * z;
  double t1 = w * (L2 + w * (L4 + w * L6));
  double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  *hfsq_out = hfsq;
  return s * (hfsq + R);
}
LF_HD double lf_log(double x) {
  const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  if (x != x) return x;
  if (x < 0.0) return lf_from_bits(0x7ff8000000000000ULL);
  if (x == 0.0) return lf_from_bits(0xfff0000000000000ULL);
  if (lf_bits(x) == 0x7ff0000000000000ULL) return x;
  int k; double f = lf_log_reduce(x, &k);
  double hfsq, t = lf_log_tail(f, &hfsq);
  double dk = (double)k;
  return dk * LN2HI - ((hfsq - (t + dk * LN2LO)) - f);
}
LF_HD double lf_log10(double x) {
  const double IVLN10 = 4.34294481903251816668e-01, LG2HI = 3.01029995663611771306e-01,
               LG2LO = 3.69423907715893078616e-13;
  if (x != x) return x;
  if (x < 0.0) return lf_from_bits(0x7ff8000000000000ULL);
  if (x == 0.0) return lf_from_bits(0xfff0000000000000ULL);
  if (lf_bits(x) == 0x7ff0000000000000ULL) return x;
  int k; double f = lf_log_reduce(x, &k);
  double hfsq, t = lf_log_tail(f, &hfsq);
  double lg = f - (hfsq - t);              /* log(1+f) */
  double dk = (double)k;
  return (dk * LG2LO + IVLN10 * lg) + dk * LG2HI;
}
/* x^y for x > 0 (LSD only needs a ratio in (0,1) to an integer-valued power,
 * inside an error bound).  Relative error ~ |y log x| * 1e-16. */
LF_HD double lf_pow(double x, double y) {
  if (y == 0.0) return 1.0;
  if (x == 1.0) return 1.0;
  return lf_exp(y * lf_log(x));
}

#endif /* LF_MATH_H */
