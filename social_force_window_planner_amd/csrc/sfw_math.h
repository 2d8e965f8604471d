// sfw_math.h — device math for the social-force pair term, written for the
// gfx950 vector ALU: no special-case branches, hardware rcp/rsq seeds refined by
// one Newton step, short Horner polynomials.  Accuracy targets: double ~1e-14
// relative (the parity tests hold the whole path to 1e-9), float ~1e-7.
// Coefficients: tools/gen_poly.py.
#ifndef SFW_MATH_H_
#define SFW_MATH_H_

#include <hip/hip_runtime.h>

namespace sfwm {

// atan(q) = q * P(q*q), q in [0,1]; max abs err 5.6e-15
__device__ constexpr double kAtanP[17] = {
    9.99999999999981348e-01,  -3.33333333325527892e-01, 1.99999999358193709e-01,  -1.42857120457493492e-01,
    1.11110683421310363e-01,  -9.09040046372013300e-02, 7.68823292769246491e-02,  -6.64350567821361121e-02,
    5.78548561483468726e-02,  -4.95689849024251034e-02, 4.01396125667346226e-02,  -2.90889119465990650e-02,
    1.77171911227863993e-02,  -8.46465692515396828e-03, 2.91643949953273836e-03,  -6.36404154247438851e-04,
    6.55251344172218128e-05};
// exp(r), |r| <= ln2/2; max rel err 1.9e-15
__device__ constexpr double kExpP[11] = {
    1.00000000000000044e+00, 1.00000000000000533e+00, 4.99999999999804712e-01, 1.66666666665838487e-01,
    4.16666666805781197e-02, 8.33333337440901947e-03, 1.38888853632852804e-03, 1.98411846695697449e-04,
    2.48051922108969186e-05, 2.76343033983400868e-06, 2.63067990309652005e-07};
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
  double at[17], ex[11];
  __device__ __forceinline__ poly_consts() {
#pragma unroll
    for (int n = 0; n < 17; ++n) at[n] = sgpr_const(kAtanP[n]);
#pragma unroll
    for (int n = 0; n < 11; ++n) ex[n] = sgpr_const(kExpP[n]);
  }
};
// v_rsq_f64 / v_rcp_f64 deliver ~23 good bits; one Newton step gives ~46.
// x must be > 0 and finite (callers clamp with fmax).
__device__ __forceinline__ void rsqrt_sqrt(double x, double &rs, double &sq) {
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y;             // ~sqrt(x)
  const double h = 0.5 * y;           // ~1/(2 sqrt(x))
  const double r = fma(-h, g, 0.5);   // residual
  sq = fma(g, r, g);
  const double h2 = fma(h, r, h);
  rs = h2 + h2;
}
__device__ __forceinline__ double rcp_nr(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(fma(e, e, e), y, y);     // two-term correction: ~2^-60
}
// exp(x) for x < ~700 (the pair term has x <= 0, the obstacle term x <= radius/sigma):
// no overflow handling; underflows to 0 through ldexp.
__device__ __forceinline__ double exp_fast(const poly_consts &pc, double x) {
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = pc.ex[10];
#pragma unroll
  for (int n = 9; n >= 0; --n) p = fma(p, r, pc.ex[n]);
  return __builtin_amdgcn_ldexp(p, static_cast<int>(k));  // v_cvt_i32_f64 saturates, ldexp flushes to 0
}
// |atan2(y, x)| for y >= 0, result in [0, pi].  (y, x) != (0, 0).
__device__ __forceinline__ double atan2_abs(const poly_consts &pc, double y, double x) {
  const double ax = fabs(x);
  const double mn = fmin(y, ax), mx = fmax(y, ax);
  const double q = mn * rcp_nr(fmax(mx, 1e-300));  // (0,0) -> 0, no NaN
  const double z = q * q;
  double p = pc.at[16];
#pragma unroll
  for (int n = 15; n >= 0; --n) p = fma(p, z, pc.at[n]);
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
