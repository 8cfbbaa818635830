// throughput of v_fmac_f64_dpp (row_newbcast) against plain v_fmac_f64 on gfx950: 256 CUs x 8 waves, 16 independent chains
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  double a[16];
  const double f = 1.0000001, g = 0.9999999 + threadIdx.x * 1e-12;
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(g), "v"(f));
      if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(g), "v"(f));
      if (MODE == 2) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[i])); asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(t), "v"(f)); }
      if (MODE == 3) { int lo, hi; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(__double2loint(a[i]))); asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(hi) : "v"(__double2hiint(a[i]))); double t = __hiloint2double(hi, lo); asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(t), "v"(f)); }
    }
  }
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double* d) {
  const int iters = 20000, blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * 256 * iters * 16;
    if (rep == 2) printf("%s: %.3f ms, %.2f T fma-lane-ops/s (= %.1f TFLOP/s), %.2f cycles per wave-instr-group at 2.4 GHz\n", name, ms, fma / ms * 1e-9, 2 * fma / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * 8 /*waves per SIMD*/ * 2));
  }
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
  run<0>("v_fmac_f64", d); run<1>("v_fmac_f64_dpp row_newbcast", d); run<2>("v_mov_b64_dpp + v_fmac_f64", d); run<3>("2 x v_mov_b32_dpp + v_fmac_f64", d);
  return 0;
}
