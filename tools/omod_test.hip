// does VOP3 omod (div:2) act on v_fma_f64 on gfx950, and under which MODE bits?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(double *out, const double *in, int variant) {
  const double x = in[0], yy = in[1];
  uint32_t mode0 = 0, mode1 = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 16)" : "=s"(mode0));
  if (variant == 1) {  // f64/f16 denormals flushed (FP_DENORM[3:2] = 0), IEEE off (bit 9)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
  } else if (variant == 2) {  // only IEEE off
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
  } else if (variant == 3) {  // only denorm flush
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");
  }
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 16)" : "=s"(mode1));
  double e, m, d;
  asm volatile("v_fma_f64 %0, -%1, %2, 1.0 div:2" : "=v"(e) : "v"(x), "v"(yy));
  asm volatile("v_mul_f64 %0, %1, %2 mul:2" : "=v"(m) : "v"(x), "v"(yy));
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(in[2]), "v"(in[3]));  // a denormal product
  out[0] = e; out[1] = m; out[2] = d;
  out[3] = static_cast<double>(mode0); out[4] = static_cast<double>(mode1);
}
int main() {
  double h_in[4] = {3.0, 0.25, 1e-160, 1e-155}, h_out[5];
  double *in, *out;
  hipMalloc(&in, sizeof h_in); hipMalloc(&out, sizeof h_out);
  hipMemcpy(in, h_in, sizeof h_in, hipMemcpyHostToDevice);
  for (int v = 0; v < 4; ++v) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in, v);
    hipMemcpy(h_out, out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("variant %d: fma(-3,.25,1) div:2 = %.17g (plain 0.25, halved 0.125); 3*.25 mul:2 = %g (plain .75); denormal product %g; MODE %#x -> %#x\n",
           v, h_out[0], h_out[1], h_out[2], (unsigned)h_out[3], (unsigned)h_out[4]);
  }
  return 0;
}
