/* oracle/crlibm_quad.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Correctly rounded sin / cos / atan2 for ONE diagnostic build of the reference's LSD
 * (oracle/Makefile: _ref/liblsd_ref_crlibm.so = the reference's own lsd.c, untouched, from where it lies
 * under /root/reference, with these three symbols bound in front of the host libm by -Wl,-Bsymbolic).
 *
 * Why: LSD's output depends on the last bit of libm in region2rect (external/lsd/lsd.cpp:1474-1604: the
 * rectangle angle atan2(...) and its cos / sin; every rectangle's end pixel lies exactly on its end edge).
 * The reference was developed on glibc <= 2.27, whose IBM Accurate Mathematical Library sin / cos / atan2 are
 * correctly rounded; glibc 2.28+ (this image: 2.35) dropped the slow paths and returns the neighbouring double
 * for about 0.5 % of arguments (error 0.500x ulp).  The HIP path evaluates those calls correctly rounded
 * (csrc/lf_math.h lf_atan2_cr / lf_sincos_cr), so against liblsd_ref.so (host glibc) a few frames in a thousand
 * differ, and against THIS build none do -- tests/test_oracle_lsd.py, tests/test_fullsize_gpu.py.
 * The untouched build _ref/liblsd_ref.so stays the pin; this one shows that what is left is the host libm's
 * rounding and nothing else.
 *
 * Independent of the product: no csrc/ header.  x87 extended precision (64-bit significand) first; if that
 * result lies too close to a rounding boundary of double to decide, GCC's libquadmath (113 bits).           */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <string.h>

/* Is rounding the extended-precision value v (error <= 2 ulp of its 64-bit significand, generously) to double
 * safe?  The low 11 bits of the significand say how far v is from a double and from a tie between two. */
static int safe_to_round(long double v) {
  uint64_t m;
  unsigned low;
  memcpy(&m, &v, 8);                 /* x86 extended: explicit 64-bit significand in the first 8 bytes */
  low = (unsigned)(m & 0x7ff);       /* tie at 0x400 */
  return !(low >= 0x400 - 4 && low <= 0x400 + 4);
}

double sin(double x) {
  long double v = sinl((long double)x);
  if (safe_to_round(v)) return (double)v;
  return (double)sinq((__float128)x);
}
double cos(double x) {
  long double v = cosl((long double)x);
  if (safe_to_round(v)) return (double)v;
  return (double)cosq((__float128)x);
}
void sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
double atan2(double y, double x) {
  long double v = atan2l((long double)y, (long double)x);
  if (safe_to_round(v)) return (double)v;
  return (double)atan2q((__float128)y, (__float128)x);
}

/* for tests: the same functions under names that do not collide with libm */
double oracle_cr_sin(double x) { return sin(x); }
double oracle_cr_cos(double x) { return cos(x); }
double oracle_cr_atan2(double y, double x) { return atan2(y, x); }
