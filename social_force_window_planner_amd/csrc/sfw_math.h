// sfw_math.h — device math for the social-force pair term, written for the
// gfx950 vector ALU: no special-case branches, hardware rcp/rsq seeds refined by
// one Newton step, short Horner polynomials.  Accuracy targets: double ~1e-14
// relative (the parity tests hold the whole path to 1e-9), float ~1e-7.
// Coefficients: tools/gen_poly.py.
#ifndef SFW_MATH_H_
#define SFW_MATH_H_

#include <hip/hip_runtime.h>

namespace sfwm {

// atan(q) = q * P(q*q), q in [0,1]; max abs err 3.0e-14
__device__ constexpr double kAtanP[16] = {
    9.99999999999944933e-01, -3.33333333306710444e-01, 1.99999997789646333e-01, -1.42857068890334782e-01,
    1.11109791772467159e-01, -9.08946625885427295e-02, 7.68179614371398145e-02, -6.61281542712686132e-02,
    5.68091358338733698e-02, -4.69734484941000119e-02, 3.54058579341653967e-02, -2.27526081143878400e-02,
    1.15690371021628380e-02, -4.25852928310632706e-03, 9.93382185697555542e-04, -1.09195709228515625e-04};
// exp(r), |r| <= ln2/2; max rel err 1.8e-14
__device__ constexpr double kExpP[10] = {
    1.00000000000001421e+00, 1.00000000000000777e+00, 4.99999999994189259e-01, 1.66666666665346158e-01,
    4.16666670498183830e-02, 8.33333339452756693e-03, 1.38888004009164279e-03, 1.98411575417668925e-04,
    2.48850574964862367e-05, 2.76457905540610794e-06};
// float atan(q) = q * P(q*q); max abs err 6.4e-8
__device__ constexpr float kAtanPf[8] = {9.999998820e-01f, -3.333181266e-01f, 1.996696183e-01f, -1.400329018e-01f,
                                         9.868865458e-02f, -5.882975314e-02f, 2.378051860e-02f, -4.559791986e-03f};

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
struct poly_consts {
  double at[16], ex[10];
  __device__ __forceinline__ poly_consts() {
#pragma unroll
    for (int n = 0; n < 16; ++n) at[n] = sgpr_const(kAtanP[n]);
#pragma unroll
    for (int n = 0; n < 10; ++n) ex[n] = sgpr_const(kExpP[n]);
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
// exp(x) for x < ~700 (the pair term has x <= 0, the obstacle term x <= radius/sigma):
// no overflow handling; underflows to 0 through ldexp.
__device__ __forceinline__ double exp_fast(const poly_consts &pc, double x) {
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = pc.ex[9];
#pragma unroll
  for (int n = 8; n >= 0; --n) p = fma(p, r, pc.ex[n]);
  return __builtin_amdgcn_ldexp(p, static_cast<int>(k));  // v_cvt_i32_f64 saturates, ldexp flushes to 0
}
// |atan2(y, x)| for y >= 0, result in [0, pi].  (y, x) != (0, 0).
__device__ __forceinline__ double atan2_abs(const poly_consts &pc, double y, double x) {
  const double ax = fabs(x);
  const double mn = fmin(y, ax), mx = fmax(y, ax);
  const double q = mn * rcp_nr(fmax(mx, 1e-300));  // (0,0) -> 0, no NaN
  const double z = q * q;
  double p = pc.at[15];
#pragma unroll
  for (int n = 14; n >= 0; --n) p = fma(p, z, pc.at[n]);
  double a = p * q;                                   // atan(q), q in [0,1]
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
__device__ __forceinline__ float atan2_abs(const poly_consts &, float y, float x) {
  const float ax = fabsf(x);
  const float mn = fminf(y, ax), mx = fmaxf(y, ax);
  const float q = mn * rcp_nr(fmaxf(mx, 1e-30f));
  const float z = q * q;
  float p = kAtanPf[7];
#pragma unroll
  for (int n = 6; n >= 0; --n) p = fmaf(p, z, kAtanPf[n]);
  float a = p * q;
  a = (y > ax) ? (1.57079632679489661923f - a) : a;
  a = (x < 0.0f) ? (3.14159265358979323846f - a) : a;
  return a;
}

}  // namespace sfwm
#endif
