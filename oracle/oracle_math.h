/*
 * oracle_math.h — TEST INFRASTRUCTURE (part of oracle/; never linked into or
 * called by the product).
 *
 * "fspt-math": the deterministic float32 arithmetic both the oracle and the
 * HIP kernels implement INDEPENDENTLY from the same written spec (DESIGN.md
 * §fspt-math), so that CPU and GPU agree bit for bit.  GLSL ES 3.00 leaves the
 * precision of sin/cos/atan/asin/pow/normalize/..., the NaN behaviour of
 * min/max and FMA contraction implementation-defined (the reference runs on
 * whatever the browser's GPU does: tracer.fs:181,205-298,410-434); this spec
 * pins one valid choice:
 *
 *   +,-,*,/,sqrt     IEEE-754 binary32, round-to-nearest-even, no flush.
 *   fma              fused, ONLY where written (build with -ffp-contract=off).
 *   fmin/fmax        IEEE minNum/maxNum (NaN loses); sign of a zero result is
 *                    unspecified and never observable in a non-zero output.
 *   sin/cos          binary64 Cody-Waite reduction by pi/2 (k = rint(x*2/pi),
 *                    r = fma(-k,PIO2_HI,x), r = fma(-k,PIO2_LO,r)), r rounded
 *                    to binary32, Cephes sinf/cosf minimax polynomials
 *                    evaluated in binary32 Horner form with fma.
 *   atan2/asin/exp2  Cephes atanf/asinf/exp2f kernels, binary32, fma Horner.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float om_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t om_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float om_min(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float om_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float om_clamp(float x, float lo, float hi) { return om_min(om_max(x, lo), hi); }
static inline float om_abs(float x) { return om_bits2f(om_f2bits(x) & 0x7fffffffu); }
static inline float om_floor(float x) { return floorf(x); }
static inline float om_fract(float x) { return x - floorf(x); }
static inline float om_fma(float a, float b, float c) { return fmaf(a, b, c); }

/* ---- sin / cos ------------------------------------------------------- */
#define OM_TWO_OVER_PI 0.63661977236758134308
#define OM_PIO2_HI 1.57079632673412561417e+00 /* first 33 bits of pi/2 */
#define OM_PIO2_LO 6.07710050650619224932e-11 /* pi/2 - PIO2_HI        */

static inline float om_sin_poly(float r) {
  float z = r * r;
  float p = om_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = om_fma(p, z, -1.6666654611e-1f);
  return om_fma(p * z, r, r);
}
static inline float om_cos_poly(float r) {
  float z = r * r;
  float p = om_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = om_fma(p, z, 4.166664568298827e-2f);
  float q = om_fma(-0.5f, z, 1.0f);
  return om_fma(p * z, z, q);
}
static inline float om_reduce(float x, int *quadrant) {
  double xd = (double)x;
  double kd = rint(xd * OM_TWO_OVER_PI);
  double r = fma(-kd, OM_PIO2_HI, xd);
  r = fma(-kd, OM_PIO2_LO, r);
  /* quadrant = kd mod 4 in exact binary64 arithmetic (no int64 conversion) */
  double qd = kd - 4.0 * floor(kd * 0.25);
  *quadrant = (qd >= 0.0 && qd < 4.0) ? (int)qd : 0; /* NaN/inf input */
  return (float)r;
}
static inline float om_sin(float x) {
  int q; float r = om_reduce(x, &q);
  float s = (q & 1) ? om_cos_poly(r) : om_sin_poly(r);
  return (q & 2) ? -s : s;
}
static inline float om_cos(float x) {
  int q; float r = om_reduce(x, &q);
  float c = (q & 1) ? om_sin_poly(r) : om_cos_poly(r);
  return ((q + 1) & 2) ? -c : c;
}

/* ---- atan2 ------------------------------------------------------------ */
#define OM_PI_F 3.14159265358979323846f
#define OM_PIO2_F 1.57079632679489661923f
#define OM_PIO4_F 0.78539816339744830962f

static inline float om_atan_01(float a) { /* a in [0,1] */
  float y0 = 0.0f, x = a;
  if (a > 0.4142135623730950f) { /* tan(pi/8) */
    x = (a - 1.0f) / (a + 1.0f);
    y0 = OM_PIO4_F;
  }
  float z = x * x;
  float p = om_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = om_fma(p, z, 1.99777106478e-1f);
  p = om_fma(p, z, -3.33329491539e-1f);
  float r = om_fma(p * z, x, x);
  return y0 + r;
}
static inline float om_atan2(float y, float x) {
  float ax = om_abs(x), ay = om_abs(y);
  float mx = om_max(ax, ay), mn = om_min(ax, ay);
  float a = (mx == 0.0f) ? 0.0f : mn / mx;
  float r = om_atan_01(a);
  if (ay > ax) r = OM_PIO2_F - r;
  if (x < 0.0f) r = OM_PI_F - r;
  if (y < 0.0f) r = -r;
  return r;
}

/* ---- asin (input clamped to [-1,1]) ------------------------------------ */
static inline float om_asin(float x) {
  float a = om_min(om_abs(x), 1.0f);
  float z, s;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); s = sqrtf(z); }
  else { z = a * a; s = a; }
  float p = om_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = om_fma(p, z, 4.5470025998e-2f);
  p = om_fma(p, z, 7.4953002686e-2f);
  p = om_fma(p, z, 1.6666752422e-1f);
  float r = om_fma(p * z, s, s);
  if (a > 0.5f) r = OM_PIO2_F - (r + r);
  return (x < 0.0f) ? -r : r;
}

/* ---- exp2 -------------------------------------------------------------- */
static inline float om_exp2(float x) {
  x = om_clamp(x, -252.0f, 252.0f);
  float kf = om_floor(x + 0.5f);
  float f = x - kf; /* [-0.5, 0.5] */
  float p = om_fma(1.535336188319500e-4f, f, 1.339887440266574e-3f);
  p = om_fma(p, f, 9.618437357674640e-3f);
  p = om_fma(p, f, 5.550332471162809e-2f);
  p = om_fma(p, f, 2.402264791363012e-1f);
  p = om_fma(p, f, 6.931472028550421e-1f);
  p = om_fma(p, f, 1.0f);
  int k = (int)kf;
  int k1 = k / 2, k2 = k - k1; /* each in [-126,126] */
  float s1 = om_bits2f((uint32_t)(k1 + 127) << 23);
  float s2 = om_bits2f((uint32_t)(k2 + 127) << 23);
  return (p * s1) * s2;
}

/* ---- log2 (Cephes log2f: frexp + degree-9 minimax), pow(x,y) = exp2(y*log2(x)) ---- */
static inline float om_log2(float x) {
  if (!(x > 0.0f)) return (x == 0.0f) ? -INFINITY : NAN;
  uint32_t b = om_f2bits(x);
  int e = 0;
  if ((b >> 23) == 0u) { x = x * 8388608.0f; b = om_f2bits(x); e = -23; } /* denormal */
  e += (int)(b >> 23) - 126;
  float m = om_bits2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = om_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = om_fma(p, m, 1.1676998740e-1f);
  p = om_fma(p, m, -1.2420140846e-1f);
  p = om_fma(p, m, 1.4249322787e-1f);
  p = om_fma(p, m, -1.6668057665e-1f);
  p = om_fma(p, m, 2.0000714765e-1f);
  p = om_fma(p, m, -2.4999993993e-1f);
  p = om_fma(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  y = om_fma(-0.5f, z, y);
  const float LOG2EA = 0.44269504088896340735992f; /* log2(e) - 1 */
  float r = y * LOG2EA;
  r = om_fma(m, LOG2EA, r);
  r = r + y;
  r = r + m;
  return r + (float)e;
}
static inline float om_pow(float x, float y) {
  if (x == 0.0f) return 0.0f; /* only y > 0 is used (draw.fs:91) */
  return om_exp2(y * om_log2(x));
}

#endif
