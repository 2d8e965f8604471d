// sfw_math.h — device math for the social-force pair term, written for the
// gfx950 vector ALU: no special-case branches, hardware rcp/rsq seeds refined by
// one Newton step, short Horner polynomials.  Accuracy targets: double ~1e-14
// relative (the parity tests hold the whole path to 1e-9), float ~1e-7.
// Coefficients: tools/gen_poly.py.
#ifndef SFW_MATH_H_
#define SFW_MATH_H_

#include <hip/hip_runtime.h>

namespace sfwm {

// 2*atan(t) = t * P(t*t), t = tan(phi/2) in [0, tan(pi/8)]; max abs err 1.9e-14
__device__ constexpr double kAtanP[9] = {
    1.99999999999994338e+00, -6.66666666614657344e-01, 3.99999991952324219e-01,
    -2.85713802188508836e-01, 2.22207562443480111e-01, -1.81566154891948661e-01,
    1.51262571875266205e-01, -1.17449017943713832e-01, 6.12649577079956778e-02};
// exp(r), |r| <= ln2/2; max rel err 1.8e-14
__device__ constexpr double kExpP[10] = {
    1.00000000000001421e+00, 1.00000000000000777e+00, 4.99999999994189259e-01, 1.66666666665346158e-01,
    4.16666670498183830e-02, 8.33333339452756693e-03, 1.38888004009164279e-03, 1.98411575417668925e-04,
    2.48850574964862367e-05, 2.76457905540610794e-06};
// float 2*atan(t) = t * P(t*t); max abs err 1.4e-8 + float rounding
__device__ constexpr float kAtanPf[5] = {1.999999963e+00f, -6.666557114e-01f, 3.994815087e-01f, -2.769682950e-01f, 1.595208338e-01f};

// ---- double ---------------------------------------------------------------
// Polynomial coefficients pinned to SGPR pairs.  Left to itself hipcc keeps the
// 28 double coefficients in VGPRs and emits v_mov_b64 + v_fmac_f64 per Horner
// term (2 VALU issues, 56 VGPRs).  Laundering each literal through an empty asm
// with an "s" constraint makes it an opaque scalar value, so the Horner steps
// select the 3-operand v_fma_f64 with a scalar addend: 1 VALU issue per term and
// no VGPRs, while the code stays ordinary C++ for the scheduler.
__device__ __forceinline__ double sgpr_const(double c) {
  asm("" : "+s"(c));
  return c;
}
// The opposite pin: keep a wave-uniform value in a VGPR (see make_consts).
__device__ __forceinline__ double vgpr_const(double c) {
  asm("" : "+v"(c));
  return c;
}
__device__ __forceinline__ float vgpr_const(float c) {
  asm("" : "+v"(c));
  return c;
}
// A VOP3 instruction reads at most one SGPR operand, so the first Horner step fma(c_n, z, c_{n-1}) of a chain
// with both coefficients in SGPRs costs an extra v_mov_b64 per evaluation: the leading coefficients live in VGPRs.
struct poly_consts {
  double at[9], ex[10];
  __device__ __forceinline__ poly_consts() {
#pragma unroll
    for (int n = 0; n < 8; ++n) at[n] = sgpr_const(kAtanP[n]);
    at[8] = vgpr_const(kAtanP[8]);
#pragma unroll
    for (int n = 0; n < 9; ++n) ex[n] = sgpr_const(kExpP[n]);
    ex[9] = vgpr_const(kExpP[9]);
  }
};
// v_rsq_f64 / v_rcp_f64 deliver ~23 good bits; one Newton step gives ~46.
// x must be > 0 and finite (callers clamp with fmax).
__device__ __forceinline__ void rsqrt_sqrt(double x, double &rs, double &sq) {
  const double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  const double r = fma(-h, y * y, 1.5);  // Newton: y1 = y0 (1.5 - 0.5 x y0^2)
  rs = y * r;
  sq = x * rs;
}
__device__ __forceinline__ double rcp_nr(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(e, y, y);                // one Newton step: ~2^-46
}
// exp(x) for -1e9 < x < ~700 (the pair term has x <= 0 and is clamped by its caller, the
// obstacle term x <= radius/sigma): no overflow handling; underflows to 0 through ldexp.
// Round-to-nearest of x*log2(e) by the 1.5*2^52 shift: the shifted sum holds k in its low
// mantissa bits (two's complement in the low dword), so no rint and no f64->i32 convert.  One
// fma reduction step: the error of fl(ln2) reaches r as |k|*2.3e-17 (< 1e-14 relative up to
// |k| ~ 400, where the result is ~1e-120 and far below anything it is added to).
__device__ __forceinline__ double exp_fast(const poly_consts &pc, double x) {
  const double shift = 6755399441055744.0;  // 1.5 * 2^52
  const double t = fma(x, 1.4426950408889634074, shift);
  const double k = t - shift;
  const double r = fma(k, -6.93147180559945286227e-01, x);
  double p = pc.ex[9];
#pragma unroll
  for (int n = 8; n >= 0; --n) p = fma(p, r, pc.ex[n]);
  return __builtin_amdgcn_ldexp(p, __double2loint(t));
}
// |atan2(y, x)| for y >= 0, result in [0, pi]; hyp = sqrt(x*x + y*y) > 0 (the
// caller has it already).  Octant fold to phi in [0, pi/4], then the half-angle
// t = tan(phi/2) = min / (max + hyp) in [0, tan(pi/8)] keeps the polynomial short.
__device__ __forceinline__ double atan2_abs(const poly_consts &pc, double y, double x, double hyp) {
  const double ax = fabs(x);
  const double mn = fmin(y, ax), mx = fmax(y, ax);
  const double t = mn * rcp_nr(mx + hyp);
  const double z = t * t;
  double p = pc.at[8];
#pragma unroll
  for (int n = 7; n >= 0; --n) p = fma(p, z, pc.at[n]);
  double a = p * t;                                   // phi
  a = (y > ax) ? (1.57079632679489661923 - a) : a;    // octant fold
  a = (x < 0.0) ? (3.14159265358979323846 - a) : a;   // half-plane fold
  return a;
}

// ---- float ----------------------------------------------------------------
__device__ __forceinline__ void rsqrt_sqrt(float x, float &rs, float &sq) {
  rs = __builtin_amdgcn_rsqf(x);
  sq = x * rs;
}
__device__ __forceinline__ float rcp_nr(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp_fast(const poly_consts &, float x) {
  return __builtin_amdgcn_exp2f(fmaxf(x * 1.44269504088896340736f, -126.0f));
}
__device__ __forceinline__ float atan2_abs(const poly_consts &, float y, float x, float hyp) {
  const float ax = fabsf(x);
  const float mn = fminf(y, ax), mx = fmaxf(y, ax);
  const float t = mn * rcp_nr(mx + hyp);
  const float z = t * t;
  float p = kAtanPf[4];
#pragma unroll
  for (int n = 3; n >= 0; --n) p = fmaf(p, z, kAtanPf[n]);
  float a = p * t;
  a = (y > ax) ? (1.57079632679489661923f - a) : a;
  a = (x < 0.0f) ? (3.14159265358979323846f - a) : a;
  return a;
}

}  // namespace sfwm
#endif
