// valu_ubench.hip — issue cost (shader cycles per wave64 instruction) of the vector-ALU instructions the K2 pair loop
// is made of, measured on the device it runs on: one wave per SIMD, eight independent dependency chains per test, the
// wave's own s_memtime around the loop.  The K2 kernels are VALU-issue bound, so these numbers are the
// cost model instruction-level changes are judged against.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_ubench.hip -o build/valu_ubench && build/valu_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITERS = 32768;

#define CHAIN8(OP)                                                                                                      \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

// d = double registers a[0..7], b (uniform multiplier), c (addend); f = float likewise
#define TEST_BODY(NAME, STMT)                                                                     \
  __global__ void __launch_bounds__(64) NAME(unsigned long long *out, double seed) {             \
    double a[8], b = seed * 1.0000001, c = seed * 0.5;                                           \
    float fa[8], fb = (float)b, fc = (float)c;                                                   \
    int ia[8];                                                                                   \
    for (int k = 0; k < 8; ++k) {                                                                \
      a[k] = seed + k * 1e-3 + threadIdx.x * 1e-6;                                               \
      fa[k] = (float)a[k];                                                                       \
      ia[k] = k + threadIdx.x;                                                                   \
    }                                                                                            \
    const unsigned long long t0 = __builtin_readcyclecounter();                                  \
    for (int it = 0; it < ITERS; ++it) {                                                         \
      STMT                                                                                       \
    }                                                                                            \
    const unsigned long long t1 = __builtin_readcyclecounter();                                  \
    double s = 0;                                                                                \
    for (int k = 0; k < 8; ++k) s += a[k] + fa[k] + ia[k];                                       \
    if (s == 12345.678) out[0] = 1;                                                              \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                             \
  }

#define OP_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define OP_MAX64(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define OP_RSQ64(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
#define OP_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define OP_SQRT64(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[k]));
#define OP_LDEXP64(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[k]) : "v"(ia[k]));
#define OP_CMP64(k) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(a[k]), "v"(b), "v"(ia[k]), "v"(ia[(k + 1) & 7]) : "vcc");
#define OP_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[k]) : "v"(ia[(k + 1) & 7]) : "vcc");
#define OP_CVT3264(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[k]) : "v"(a[k]));
#define OP_CVT6432(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(fa[k]));
#define OP_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[k]) : "v"(fb), "v"(fc));
#define OP_PKFMA32(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_PKMUL32(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_EXP32(k) asm volatile("v_exp_f32 %0, %0" : "+v"(fa[k]));
#define OP_RSQ32(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(fa[k]));
#define OP_RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa[k]));
#define OP_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) & 7]));
#define OP_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[k]) : "v"(ia[(k + 1) & 7]));
#define OP_BFI(k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(ia[k]) : "v"(ia[(k + 1) & 7]), "v"(ia[(k + 2) & 7]));
#define OP_FRACT64(k) asm volatile("v_fract_f64 %0, %0" : "+v"(a[k]));
#define OP_RNDNE64(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
#define OP_CVTI64(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[k]) : "v"(a[k]));
#define OP_NOP(k) asm volatile("s_nop 0");
// v_fma_f64 with an SGPR addend (the Horner form of the pair loop)
#define OP_FMA64S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(seed));
// dependent chain of fma f64 (latency)
#define OP_FMA64DEP(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
// mixed: alternating f64 fma and f32 fma (do they overlap?)
#define OP_MIX(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(a[k]), "+v"(fa[k]) : "v"(b), "v"(c), "v"(fb), "v"(fc));
// f64 fma next to a transcendental f64 (separate pipe?)
#define OP_MIXT(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %0, %0, %2, %3\n\tv_rsq_f64 %1, %1" : "+v"(a[k]), "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));

TEST_BODY(t_fma64, CHAIN8(OP_FMA64))
TEST_BODY(t_fma64s, CHAIN8(OP_FMA64S))
TEST_BODY(t_fma64dep, CHAIN8(OP_FMA64DEP))
TEST_BODY(t_mul64, CHAIN8(OP_MUL64))
TEST_BODY(t_add64, CHAIN8(OP_ADD64))
TEST_BODY(t_max64, CHAIN8(OP_MAX64))
TEST_BODY(t_rsq64, CHAIN8(OP_RSQ64))
TEST_BODY(t_rcp64, CHAIN8(OP_RCP64))
TEST_BODY(t_sqrt64, CHAIN8(OP_SQRT64))
TEST_BODY(t_ldexp64, CHAIN8(OP_LDEXP64))
TEST_BODY(t_cmp64_cnd, CHAIN8(OP_CMP64))
TEST_BODY(t_cnd, CHAIN8(OP_CND))
TEST_BODY(t_cvt_f32_f64, CHAIN8(OP_CVT3264))
TEST_BODY(t_cvt_f64_f32, CHAIN8(OP_CVT6432))
TEST_BODY(t_fma32, CHAIN8(OP_FMA32))
TEST_BODY(t_pk_fma32, CHAIN8(OP_PKFMA32))
TEST_BODY(t_pk_mul32, CHAIN8(OP_PKMUL32))
TEST_BODY(t_exp32, CHAIN8(OP_EXP32))
TEST_BODY(t_rsq32, CHAIN8(OP_RSQ32))
TEST_BODY(t_rcp32, CHAIN8(OP_RCP32))
TEST_BODY(t_mov64, CHAIN8(OP_MOV64))
TEST_BODY(t_xor, CHAIN8(OP_XOR))
TEST_BODY(t_bfi, CHAIN8(OP_BFI))
TEST_BODY(t_fract64, CHAIN8(OP_FRACT64))
TEST_BODY(t_rndne64, CHAIN8(OP_RNDNE64))
TEST_BODY(t_cvt_i32_f64, CHAIN8(OP_CVTI64))
TEST_BODY(t_nop, CHAIN8(OP_NOP))
TEST_BODY(t_mix_f64_f32, CHAIN8(OP_MIX))
TEST_BODY(t_mix_3fma_rsq, CHAIN8(OP_MIXT))

struct test {
  const char *name;
  void (*fn)(unsigned long long *, double);
  int instr_per_op;
};

int main(int argc, char **argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
  unsigned long long *d = nullptr;
  hipMalloc(&d, 16);
  const test tests[] = {
      {"v_fma_f64 (vgpr addend)", t_fma64, 1}, {"v_fma_f64 (sgpr addend)", t_fma64s, 1}, {"v_fma_f64 dependent chain", t_fma64dep, 1},
      {"v_mul_f64", t_mul64, 1}, {"v_add_f64", t_add64, 1}, {"v_max_f64", t_max64, 1}, {"v_rsq_f64", t_rsq64, 1},
      {"v_rcp_f64", t_rcp64, 1}, {"v_sqrt_f64", t_sqrt64, 1}, {"v_ldexp_f64", t_ldexp64, 1},
      {"v_cmp_gt_f64 + v_cndmask_b32 (pair)", t_cmp64_cnd, 1}, {"v_cndmask_b32", t_cnd, 1},
      {"v_cvt_f32_f64", t_cvt_f32_f64, 1}, {"v_cvt_f64_f32", t_cvt_f64_f32, 1}, {"v_fma_f32", t_fma32, 1},
      {"v_pk_fma_f32", t_pk_fma32, 1}, {"v_pk_mul_f32", t_pk_mul32, 1}, {"v_exp_f32", t_exp32, 1}, {"v_rsq_f32", t_rsq32, 1},
      {"v_rcp_f32", t_rcp32, 1}, {"v_mov_b64", t_mov64, 1}, {"v_xor_b32", t_xor, 1}, {"v_bfi_b32", t_bfi, 1},
      {"v_fract_f64", t_fract64, 1}, {"v_rndne_f64", t_rndne64, 1}, {"v_cvt_i32_f64", t_cvt_i32_f64, 1}, {"s_nop 0", t_nop, 1},
      {"v_fma_f64 + v_fma_f32 (pair)", t_mix_f64_f32, 1}, {"3 v_fma_f64 + v_rsq_f64 (group of 4)", t_mix_3fma_rsq, 1},
  };
  const int grid = 1024 * waves_per_simd;
  hipFree(d);
  hipMalloc(&d, sizeof(unsigned long long) * grid);
  std::vector<unsigned long long> v(grid);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("waves per SIMD: %d (grid %d blocks of 64); cycles per instruction as seen by a wave (mean / max over waves), and\n"
         "SIMD issue cost = kernel wall time x 2.4 GHz / instructions per SIMD (an upper bound: launch ramp included)\n", waves_per_simd, grid);
  for (const test &t : tests) {
    double best_mean = 1e30, best_max = 1e30, best_ms = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0, sizeof(unsigned long long) * grid);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), 0, 0, d, 1.0);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(v.data(), d, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
      double sum = 0, mx = 0;
      for (int i = 0; i < grid; ++i) {
        sum += double(v[i]);
        if (double(v[i]) > mx) mx = double(v[i]);
      }
      const double mean = sum / grid / (8.0 * ITERS);
      if (mean < best_mean) best_mean = mean;
      if (mx / (8.0 * ITERS) < best_max) best_max = mx / (8.0 * ITERS);
      if (ms < best_ms) best_ms = ms;
    }
    printf("%-40s wave %6.2f / %6.2f   SIMD issue <= %5.2f cycles @2.4GHz (%.1f us)\n", t.name, best_mean, best_max,
           best_ms * 1e-3 * 2.4e9 / (8.0 * ITERS * waves_per_simd), best_ms * 1e3);
  }
  hipFree(d);
  return 0;
}
