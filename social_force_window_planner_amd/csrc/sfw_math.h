// sfw_math.h — device math for the social-force pair term, written for the
// gfx950 vector ALU: no special-case branches, hardware rcp/rsq seeds refined by
// one Newton step, short Horner polynomials.  Accuracy targets: double ~5e-14
// relative (asin 7 / exp 9; the parity tests hold the whole path to 1e-9), float ~1e-7.
// Coefficients: tools/gen_poly.py.
#ifndef SFW_MATH_H_
#define SFW_MATH_H_

#include <hip/hip_runtime.h>

namespace sfwm {

// Degrees of the two f64 polynomials of the pair term (tuning knobs, csrc/Makefile EXTRA; tools/gen_poly.py prints the
// coefficient sets and their errors).  Every VALU instruction of the pair loop costs one 4-cycle issue slot, so a degree
// is an issue slot per evaluation.  asin 7 / exp 9 (round 5) put the pair term at ~5e-14 relative; with exp 8 (rounds 3-4,
// ~1e-12, 1.6 % less of K2) four random scenes in 6000 — chaotic 0.25 s-Euler crowds — deviated by more than 50 x the oracle's
// own conditioning, with degree 9 none (profiles/r04_sweep_degrees.txt, r05_parity_sweep.txt).  The parity tests hold the whole
// rollout to 1e-9, the north star asks for 1e-4.
#ifndef SFW_ASIN_DEG
#define SFW_ASIN_DEG 7
#endif
#ifndef SFW_EXP_DEG
#define SFW_EXP_DEG 9
#endif
// asin(n) = n * Q(n*n), |n| <= sin(pi/8)
#if SFW_ASIN_DEG == 8  // max abs err 2.3e-15
__device__ constexpr double kAsinQ[9] = {
    1.00000000000000266e+00, 1.66666666662883378e-01, 7.50000007366493221e-02, 4.46428039656598177e-02, 3.03838357891550621e-02,
    2.23348546026741132e-02, 1.77785509266840877e-02, 1.12009038561371455e-02, 2.06905183025598710e-02};
#elif SFW_ASIN_DEG == 7  // max abs err 5.4e-14
__device__ constexpr double kAsinQ[8] = {
    9.99999999999873768e-01, 1.66666666776879580e-01, 7.49999842884431500e-02, 4.46437065808175729e-02,
    3.03595339518422935e-02, 2.26899347940125971e-02, 1.49057938849694732e-02, 2.32986453124961607e-02};
#elif SFW_ASIN_DEG == 6  // max abs err 1.6e-12
__device__ constexpr double kAsinQ[7] = {
    1.00000000000385847e+00, 1.66666664088492733e-01, 7.50002798152002437e-02, 4.46315408500555302e-02,
    3.05977872550623718e-02, 2.02964541344591819e-02, 2.68224190000059641e-02};
#else
#error "SFW_ASIN_DEG must be 6, 7 or 8"
#endif
// 2^r, |r| <= 1/2.  Every exponential of the path is evaluated in base 2: the arguments arrive in log2 units (the host
// folds log2(e) into the force constants, sfw_derive), so the range reduction is k = rint(x), r = x - k — an exact
// subtraction, no product with a rounded ln 2 — and the laser-point term, whose exponent is a plain product, forms it
// inside the two fma of the reduction (exp2_scaled: one issue slot less per point than exp(fma(d, -1/sigma, c0))).
#if SFW_EXP_DEG == 9  // max rel err 1.8e-14
__device__ constexpr double kExp2P[10] = {
    1.00000000000001399e+00, 6.93147180559949172e-01, 2.40226506956318808e-01, 5.55041086644670403e-02,
    9.61812919591992048e-03, 1.33335582320688226e-03, 1.54034323372305410e-04, 1.52526540189686452e-05,
    1.32599492119502467e-06, 1.02095991152988666e-07};
#elif SFW_EXP_DEG == 8  // max rel err 1.1e-12
__device__ constexpr double kExp2P[9] = {
    9.99999999999999556e-01, 6.93147180545924502e-01, 2.40226506958204633e-01, 5.55041094124378298e-02, 9.61812915779665205e-03,
    1.33334505311817272e-03, 1.54034569411874581e-04, 1.53100893037719350e-05, 1.32549957111681631e-06};
#elif SFW_EXP_DEG == 7  // max rel err 5.5e-11
__device__ constexpr double kExp2P[8] = {
    9.99999999959529817e-01, 6.93147180556832998e-01, 2.40226512138058845e-01, 5.55041090633913992e-02,
    9.61802557216482473e-03, 1.33334784505842571e-03, 1.54697424034518652e-04, 1.53037088618789910e-05};
#else
#error "SFW_EXP_DEG must be 7, 8 or 9"
#endif
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
__device__ __forceinline__ float sgpr_const(float c) {
  asm("" : "+s"(c));
  return c;
}
// from an SGPR into a VGPR, HERE (volatile: not hoisted out of the loop it stands in): a wave-uniform value that lives in
// scalar registers across a rollout and in vector registers only where a pass needs it there
__device__ __forceinline__ double vgpr_copy_here(double c) {
  asm volatile("" : "+v"(c));
  return c;
}
__device__ __forceinline__ float vgpr_copy_here(float c) {
  asm volatile("" : "+v"(c));
  return c;
}
__device__ __forceinline__ double vgpr_const(double c) {
  asm("" : "+v"(c));
  return c;
}
__device__ __forceinline__ float vgpr_const(float c) {
  asm("" : "+v"(c));
  return c;
}
// ... materialised HERE, by the asm itself (two v_mov_b32 of literals): for a constant that a kernel rebuilds per step so
// that it is not held across the passes that do not use it.  (A volatile asm is not hoisted out of the loop it stands in —
// but handed the constant as a register INPUT, the compiler materialises that input in front of the rollout and, at the
// register budgets of the K2 kernels, keeps it in scratch.)
constexpr uint32_t f64_lo(double c) { return static_cast<uint32_t>(__builtin_bit_cast(uint64_t, c)); }
constexpr uint32_t f64_hi(double c) { return static_cast<uint32_t>(__builtin_bit_cast(uint64_t, c) >> 32); }
template <uint32_t LO, uint32_t HI> __device__ __forceinline__ double vgpr_literal_here() {
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "n"(LO), "n"(HI));
  return __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
}
// A VOP3 instruction reads at most one SGPR operand, so the first Horner step fma(c_n, z, c_{n-1}) of a chain
// with both coefficients in SGPRs costs an extra v_mov_b64 per evaluation: the leading coefficients live in VGPRs.
struct poly_consts {
  double as[SFW_ASIN_DEG + 1], ex[SFW_EXP_DEG + 1];
  __device__ __forceinline__ poly_consts() {
#pragma unroll
    for (int n = 0; n < SFW_ASIN_DEG; ++n) as[n] = sgpr_const(kAsinQ[n]);
#pragma unroll
    for (int n = 0; n < SFW_EXP_DEG; ++n) ex[n] = sgpr_const(kExp2P[n]);
    leading_here();
  }
  // The two leading coefficients, in VGPRs, materialised where this is called: a kernel that keeps the scalar part across
  // its rollout (built once: rebuilding it per step is ~70 s_mov, which a lone wave pays in full) calls this once per step,
  // so that the four vector registers are free during the passes that do not evaluate a polynomial.
  __device__ __forceinline__ void leading_here() {
    as[SFW_ASIN_DEG] = vgpr_literal_here<f64_lo(kAsinQ[SFW_ASIN_DEG]), f64_hi(kAsinQ[SFW_ASIN_DEG])>();
    ex[SFW_EXP_DEG] = vgpr_literal_here<f64_lo(kExp2P[SFW_EXP_DEG]), f64_hi(kExp2P[SFW_EXP_DEG])>();
  }
};
// The Newton step of the reciprocal square root needs (1 - x y^2) / 2.  The halving is the VOP3 output modifier of the fma
// (div:2), which the hardware applies only with FP64 denormals flushed and the IEEE bit of the MODE register off
// (measured on gfx950, tools/omod_test.hip: ignored otherwise): every kernel that reaches rsqrt_sqrt(double) calls
// fp_mode_for_omod() first, as its first statement (the fma is an ordinary asm — a volatile one keeps loops with a run-time
// trip count from being unrolled —: its operands derive from loads, and no load moves across the volatile mode switch; a
// kernel that forgot the call would be off by 1e-7 in every norm and fail every parity test).  The result is the bit
// pattern of fma(-x/2, y^2, 1/2) (scaling by two commutes with the rounding); what changes is that results below 2.2e-308
// (an exponential with an argument under -708) become 0 where they were denormal: no sum of forces can tell.
#ifndef SFW_OMOD
#define SFW_OMOD 1
#endif
__device__ __forceinline__ void fp_mode_for_omod() {
#if SFW_OMOD
  // MODE[7:6] = FP_DENORM of f64 / f16: 0 = flush inputs and outputs; MODE[9] = IEEE
  // "memory": no load moves across the switch, so nothing derived from a load — every norm's operand is — is computed in
  // front of it (the fma below is an ordinary asm and carries no dependency of its own on the mode; tools/isa_asm_hazards.py,
  // run by tests/test_kernel_resources.py, checks on the ISA that every kernel holding such an fma switches the mode first)
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0" ::: "memory");
#endif
}
// v_rsq_f64 / v_rcp_f64 deliver ~23 good bits; one Newton step gives ~46.
// x must be > 0 and finite (callers clamp with fmax).
__device__ __forceinline__ void rsqrt_sqrt(double x, double &rs, double &sq) {
  const double y = __builtin_amdgcn_rsq(x);
#if SFW_OMOD
  double e;  // (1 - x y^2) / 2 in one issue
  [[clang::noconvergent]] { asm("v_fma_f64 %0, -%1, %2, 1.0 div:2" : "=v"(e) : "v"(x), "v"(y * y)); }
#else
  const double h = 0.5 * x;
  const double e = fma(-h, y * y, 0.5);  // Newton: y1 = y0 (1.5 - 0.5 x y0^2) = y0 + y0 (0.5 - 0.5 x y0^2): the same four
#endif                                   // issues with 0.5 (an inline constant) instead of 1.5 (a literal in a VGPR pair)
  rs = fma(y, e, y);
  sq = x * rs;
}
__device__ __forceinline__ double rcp_nr(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(e, y, y);                // one Newton step: ~2^-46
}
// 2^x for -1e9 < x < ~1000 (the pair term has x <= log2 Fs and is clamped by its caller): no overflow handling;
// underflows to 0 through ldexp.  Round-to-nearest of x by the 1.5*2^52 shift: the shifted sum holds k in its low
// mantissa bits (two's complement in the low dword), so no rint and no f64->i32 convert; r = x - k is exact.
__device__ __forceinline__ double exp2_poly(const poly_consts &pc, double r) {
  double p = pc.ex[SFW_EXP_DEG];
#pragma unroll
  for (int n = SFW_EXP_DEG - 1; n >= 0; --n) p = fma(p, r, pc.ex[n]);
  return p;
}
__device__ __forceinline__ double exp2_fast(const poly_consts &pc, double x) {
  const double shift = 6755399441055744.0;  // 1.5 * 2^52
  const double t = x + shift;
  const double k = t - shift;
  return __builtin_amdgcn_ldexp(exp2_poly(pc, x - k), __double2loint(t));
}
// 2^(u c): the product is formed inside the two fma of the range reduction (t = u c + shift, r = u c - k, the second one
// with a single rounding), never on its own.  u c > -1e9.
__device__ __forceinline__ double exp2_scaled(const poly_consts &pc, double u, double c) {
  const double shift = 6755399441055744.0;  // 1.5 * 2^52
  const double t = fma(u, c, shift);
  const double k = t - shift;
  return __builtin_amdgcn_ldexp(exp2_poly(pc, fma(u, c, -k)), __double2loint(t));
}
// Two exponentials at once, their Horner chains interleaved: a lone wave per SIMD (shared-prefix levels, control-cycle
// grids) pays the ~8-cycle dependent-issue latency of every link of a chain, two independent chains hide each other's.
// Same operations on the same values as two exp2_fast calls: bit-identical.
__device__ __forceinline__ void exp2_fast2(const poly_consts &pc, double x1, double x2, double &e1, double &e2) {
  const double shift = 6755399441055744.0;  // 1.5 * 2^52
  const double t1 = x1 + shift, t2 = x2 + shift;
  const double k1 = t1 - shift, k2 = t2 - shift;
  const double r1 = x1 - k1, r2 = x2 - k2;
  double p1 = pc.ex[SFW_EXP_DEG], p2 = pc.ex[SFW_EXP_DEG];
#pragma unroll
  for (int n = SFW_EXP_DEG - 1; n >= 0; --n) {
    p1 = fma(p1, r1, pc.ex[n]);
    p2 = fma(p2, r2, pc.ex[n]);
  }
  e1 = __builtin_amdgcn_ldexp(p1, __double2loint(t1));
  e2 = __builtin_amdgcn_ldexp(p2, __double2loint(t2));
}
// The same with the second exponential exactly 0 in the lanes whose `cw` is zero: the integer exponent of its 2^k scaling is replaced by one far below the denormals
// (v_ldexp_f64 then returns +0) — one v_cndmask_b32 on the integer instead of two on the result.  The replacement is the bit
// pattern of -4.0f read as an integer (-1 065 353 216): an INLINE constant of the instruction, so it costs neither a VGPR
// (the register form sits exactly at its 80-VGPR budget) nor the constant bus (a literal next to the vcc mask does not
// assemble on gfx9); the asm keeps the compiler from materialising it in a register across the rollout all the same, and
// the compare writes vcc for the select right behind it (through an SGPR pair and s_mov the two sat on the dependency chain
// of a lone wave: a control cycle's pair loop ran 8-12 % longer).
__device__ __forceinline__ int gate_exponent(int k, double cw) {
  int r;
  asm("v_cmp_neq_f64 vcc, 0, %2\n\tv_cndmask_b32_e32 %0, -4.0, %1, vcc" : "=v"(r) : "v"(k), "v"(cw) : "vcc");
  return r;
}
__device__ __forceinline__ void exp2_fast2_gated(const poly_consts &pc, double x1, double x2, double cw, double &e1, double &e2) {
  const double shift = 6755399441055744.0;  // 1.5 * 2^52
  const double t1 = x1 + shift, t2 = x2 + shift;
  const double k1 = t1 - shift, k2 = t2 - shift;
  const double r1 = x1 - k1, r2 = x2 - k2;
  double p1 = pc.ex[SFW_EXP_DEG], p2 = pc.ex[SFW_EXP_DEG];
#pragma unroll
  for (int n = SFW_EXP_DEG - 1; n >= 0; --n) {
    p1 = fma(p1, r1, pc.ex[n]);
    p2 = fma(p2, r2, pc.ex[n]);
  }
  e1 = __builtin_amdgcn_ldexp(p1, __double2loint(t1));
  e2 = __builtin_amdgcn_ldexp(p2, gate_exponent(__double2loint(t2), cw));
}
// theta = |atan2(y, x)| in [0, pi] for y >= 0, from y, nx = -x and rh = 1 / sqrt(x*x + y*y) (the caller has the
// reciprocal norm already) — without a division and without a select:
//   * octant fold: phi = atan2(min, max) in [0, pi/4] of (min, max) = sorted (y, |x|);
//   * the point (max, min) is turned back by pi/8, so that sin(phi - pi/8) = (min cos(pi/8) - max sin(pi/8)) rh lies in
//     +-sin(pi/8) and asin is a short odd polynomial there: phi = pi/8 + asin(n).  With p = y + |x| and d = y - |x|
//     the sorted pair never has to be formed: min c - max s = p (c - s)/2 - |d| (c + s)/2;
//   * unfold by sign transfers (v_bfi_b32 on the high dword) instead of compare + subtract + select: e = phi - pi/4 is
//     <= 0, the octant fold maps it to -e exactly when y > |x| (sign of d); f = (that) - pi/4 is <= 0 again, the
//     half-plane fold maps it to -f exactly when x < 0 (sign of nx); theta = f + pi/2.  (Rounding can leave e or f a few
//     1e-17 on the wrong side of 0: the transfer then moves the angle by that much.)
// 19 issue slots at degree 7, against 29 for the half-angle form t = min / (max + hyp) with its v_rcp_f64 (4 slots) +
// Newton step and two compare/select folds.
__device__ __forceinline__ double angle_abs(const poly_consts &pc, double y, double nx, double rh, double /*hyp*/) {
  const double ax = fabs(nx);
  const double d = y - ax, p = y + ax;
  const double nu = fma(p, 2.70598050073098492e-01, -(fabs(d) * 6.53281482438188264e-01));  // (c -+ s)/2, c,s = cos,sin(pi/8)
  const double n = nu * rh;
  const double z = n * n;
  double q = pc.as[SFW_ASIN_DEG];
#pragma unroll
  for (int k = SFW_ASIN_DEG - 1; k >= 0; --k) q = fma(q, z, pc.as[k]);
  const double e = fma(n, q, sgpr_const(-3.92699081698724155e-01));  // phi - pi/4 = asin(n) - pi/8 (scalar addend: no v_mov + v_fmac)
  const double f = __builtin_copysign(e, d) - 7.85398163397448310e-01;  // (octant-folded angle) - pi/2
  return __builtin_copysign(f, nx) + 1.57079632679489661923;
}

// ---- float ----------------------------------------------------------------
__device__ __forceinline__ void rsqrt_sqrt(float x, float &rs, float &sq) {
  rs = __builtin_amdgcn_rsqf(x);
  sq = x * rs;
}
__device__ __forceinline__ float rcp_nr(float x) { return __builtin_amdgcn_rcpf(x); }
// 2^x in float: v_exp_f32; below -150 the result is 0 (clamped so that no denormal-handling mode decides)
__device__ __forceinline__ float exp2_fast(const poly_consts &, float x) { return __builtin_amdgcn_exp2f(fmaxf(x, -150.0f)); }
__device__ __forceinline__ float exp2_scaled(const poly_consts &pc, float u, float c) { return exp2_fast(pc, u * c); }
__device__ __forceinline__ void exp2_fast2(const poly_consts &pc, float x1, float x2, float &e1, float &e2) {
  e1 = exp2_fast(pc, x1);
  e2 = exp2_fast(pc, x2);
}
__device__ __forceinline__ void exp2_fast2_gated(const poly_consts &pc, float x1, float x2, double cw, float &e1, float &e2) {
  e1 = exp2_fast(pc, x1);
  const float e = exp2_fast(pc, x2);
  asm("v_cmp_neq_f64 vcc, 0, %2\n\tv_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(e2) : "v"(e), "v"(cw) : "vcc");
}
__device__ __forceinline__ float angle_abs(const poly_consts &, float y, float nx, float /*rh*/, float hyp) {
  const float x = -nx;
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
