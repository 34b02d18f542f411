// fspt_math.hpp — device-side "fspt-math" (DESIGN.md §fspt-math).
//
// The float32 arithmetic contract of the path tracer, written for gfx950.
// GLSL ES 3.00 leaves sin/cos/atan/asin/pow/normalize precision, min/max NaN
// behaviour and FMA contraction to the implementation (the reference inherits
// whatever the browser GPU does, tracer.fs:181,205-298,410-434).  This header
// pins one valid choice so that results are reproducible bit for bit:
//   * + - * / sqrt : IEEE binary32 RN (hipcc default: correctly rounded
//     divide/sqrt, denormals kept);  fma only where written (-ffp-contract=off);
//   * min/max      : v_min_f32 / v_max_f32 = IEEE minNum/maxNum;
//   * sin/cos      : binary64 Cody-Waite reduction by pi/2, binary32 minimax
//     polynomial (Cephes sinf/cosf coefficients), Horner with fma;
//   * atan2/asin/exp2 : Cephes atanf/asinf/exp2f kernels, Horner with fma.
// The CPU oracle implements the same spec separately (oracle/oracle_math.h);
// tests/test_math_parity.py compares the two bitwise through fspt_math_eval.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fm {

#define FM_DEV __device__ __forceinline__

struct V3 { float x, y, z; };

FM_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
FM_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
FM_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
FM_DEV float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
FM_DEV float abs_(float x) { return __builtin_fabsf(x); }
FM_DEV float floor_(float x) { return __builtin_floorf(x); }
FM_DEV float fract_(float x) { return x - __builtin_floorf(x); }
FM_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
FM_DEV float bits2f(uint32_t u) { return __uint_as_float(u); }
FM_DEV uint32_t f2bits(float f) { return __float_as_uint(f); }

// ---- sin / cos -----------------------------------------------------------
FM_DEV float sin_poly(float r) {
  float z = r * r;
  float p = fma_(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = fma_(p, z, -1.6666654611e-1f);
  return fma_(p * z, r, r);
}
FM_DEV float cos_poly(float r) {
  float z = r * r;
  float p = fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = fma_(p, z, 4.166664568298827e-2f);
  float q = fma_(-0.5f, z, 1.0f);
  return fma_(p * z, z, q);
}
// (Round 5 timed a binary32 four-constant reduction for |x| <= 2e5 in place of the binary64 one: primary launch 0.1274 vs
// 0.1275-0.130 ms per tick, nothing - the reduction is not what the shading waits for; profiles/r05/ab_carry_sin_c2_20.log.)
FM_DEV float reduce_pio2(float x, int &quadrant) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double PIO2_HI = 1.57079632673412561417e+00;
  const double PIO2_LO = 6.07710050650619224932e-11;
  double xd = (double)x;
  double kd = __builtin_rint(xd * TWO_OVER_PI);
  double r = __builtin_fma(-kd, PIO2_HI, xd);
  r = __builtin_fma(-kd, PIO2_LO, r);
  // kd mod 4.  |kd| < 2^31 (every |x| < 3.3e9): the two low bits of the integer, which is the same number as the
  // general form below - two VALU instructions instead of seven double-precision ones per sin / cos / rnd.  The general
  // form sits behind a wave-uniform branch (a lane-masked block would still be issued, with no lane enabled).
  quadrant = (int)kd & 3;
  const bool big = !(__builtin_fabs(kd) < 2147483648.0);
  if (__builtin_expect(__ballot(big) != 0ull, 0)) {
    if (big) {
      double qd = kd - 4.0 * __builtin_floor(kd * 0.25);
      quadrant = (qd >= 0.0 && qd < 4.0) ? (int)qd : 0; // (NaN / infinite x: 0)
    }
  }
  return (float)r;
}
FM_DEV float sin_(float x) {
  int q;
  float r = reduce_pio2(x, q);
  float s = (q & 1) ? cos_poly(r) : sin_poly(r);
  return (q & 2) ? -s : s;
}
FM_DEV float cos_(float x) {
  int q;
  float r = reduce_pio2(x, q);
  float c = (q & 1) ? sin_poly(r) : cos_poly(r);
  return ((q + 1) & 2) ? -c : c;
}
// sin and cos of the same angle share the reduction (bitwise equal to sin_/cos_)
FM_DEV void sincos_(float x, float &s, float &c) {
  int q;
  float r = reduce_pio2(x, q);
  float sp = sin_poly(r), cp = cos_poly(r);
  float sv = (q & 1) ? cp : sp;
  float cv = (q & 1) ? sp : cp;
  s = (q & 2) ? -sv : sv;
  c = ((q + 1) & 2) ? -cv : cv;
}

// ---- atan2 ------------------------------------------------------------------
FM_DEV float atan_01(float a) {
  float y0 = 0.0f, x = a;
  if (a > 0.4142135623730950f) {
    x = (a - 1.0f) / (a + 1.0f);
    y0 = 0.78539816339744830962f;
  }
  float z = x * x;
  float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fma_(p, z, 1.99777106478e-1f);
  p = fma_(p, z, -3.33329491539e-1f);
  float r = fma_(p * z, x, x);
  return y0 + r;
}
FM_DEV float atan2_(float y, float x) {
  float ax = abs_(x), ay = abs_(y);
  float mx = max_(ax, ay), mn = min_(ax, ay);
  float a = (mx == 0.0f) ? 0.0f : mn / mx;
  float r = atan_01(a);
  if (ay > ax) r = 1.57079632679489661923f - r;
  if (x < 0.0f) r = 3.14159265358979323846f - r;
  if (y < 0.0f) r = -r;
  return r;
}

// ---- asin (input clamped to [-1,1]) ---------------------------------------------
FM_DEV float asin_(float x) {
  float a = min_(abs_(x), 1.0f);
  float z, s;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); s = sqrt_(z); }
  else { z = a * a; s = a; }
  float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = fma_(p, z, 4.5470025998e-2f);
  p = fma_(p, z, 7.4953002686e-2f);
  p = fma_(p, z, 1.6666752422e-1f);
  float r = fma_(p * z, s, s);
  if (a > 0.5f) r = 1.57079632679489661923f - (r + r);
  return (x < 0.0f) ? -r : r;
}

// ---- exp2 ---------------------------------------------------------------------
FM_DEV float exp2_(float x) {
  x = clamp_(x, -252.0f, 252.0f);
  float kf = floor_(x + 0.5f);
  float f = x - kf;
  float p = fma_(1.535336188319500e-4f, f, 1.339887440266574e-3f);
  p = fma_(p, f, 9.618437357674640e-3f);
  p = fma_(p, f, 5.550332471162809e-2f);
  p = fma_(p, f, 2.402264791363012e-1f);
  p = fma_(p, f, 6.931472028550421e-1f);
  p = fma_(p, f, 1.0f);
  int k = (int)kf;
  int k1 = k / 2, k2 = k - k1;
  float s1 = bits2f((uint32_t)(k1 + 127) << 23);
  float s2 = bits2f((uint32_t)(k2 + 127) << 23);
  return (p * s1) * s2;
}

// ---- log2 (Cephes log2f), pow(x,y) = exp2(y*log2(x)) (draw.fs:91 only) -----------
FM_DEV float log2_(float x) {
  if (!(x > 0.0f)) return (x == 0.0f) ? -__builtin_inff() : __builtin_nanf("");
  uint32_t b = f2bits(x);
  int e = 0;
  if ((b >> 23) == 0u) { x = x * 8388608.0f; b = f2bits(x); e = -23; }
  e += (int)(b >> 23) - 126;
  float m = bits2f((b & 0x007fffffu) | 0x3f000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = fma_(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = fma_(p, m, 1.1676998740e-1f);
  p = fma_(p, m, -1.2420140846e-1f);
  p = fma_(p, m, 1.4249322787e-1f);
  p = fma_(p, m, -1.6668057665e-1f);
  p = fma_(p, m, 2.0000714765e-1f);
  p = fma_(p, m, -2.4999993993e-1f);
  p = fma_(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  y = fma_(-0.5f, z, y);
  const float LOG2EA = 0.44269504088896340735992f;
  float r = y * LOG2EA;
  r = fma_(m, LOG2EA, r);
  r = r + y;
  r = r + m;
  return r + (float)e;
}
FM_DEV float pow_(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return exp2_(y * log2_(x));
}

// ---- vec3 helpers (explicit fma placement is part of the spec) ------------------
FM_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
FM_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
FM_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
FM_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
FM_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
FM_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
FM_DEV V3 vfma(V3 a, float s, V3 b) { return v3(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)); }
FM_DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
FM_DEV V3 cross(V3 a, V3 b) {
  return v3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
FM_DEV V3 normalize(V3 a) {
  float inv = 1.0f / sqrt_(dot(a, a));
  return a * inv;
}
// w.x*a + w.y*b + w.z*c
FM_DEV V3 bary3(V3 w, V3 a, V3 b, V3 c) {
  return v3(fma_(w.z, c.x, fma_(w.y, b.x, w.x * a.x)), fma_(w.z, c.y, fma_(w.y, b.y, w.x * a.y)),
            fma_(w.z, c.z, fma_(w.y, b.z, w.x * a.z)));
}
FM_DEV float lerp_(float x, float y, float a) { return fma_(a, y - x, x); }

// rnd (tracer.fs:181, camera.fs:19)
FM_DEV float rnd(float &seed) {
  seed = seed + 0.211324865405187f;
  return fract_(sin_(seed) * 43758.5453123f);
}

}  // namespace fm
