// issue rate of v_fmac_f64 / v_fmac_f64_dpp for ONE wave per SIMD (the D = 8 direct solve's regime), gfx950:
// 64 accumulators a[4][16]; patterns: 0 plain fmac (independent), 1 fmac_dpp with an unrelated source register,
// 2 the d8_update pattern (source = the pivot row's own accumulator, written last), 3 = 2 with v_fma_f64 (VOP3, no DPP) as control,
// 4 = v_mov_b64_dpp of the pivot row's accumulator into a temporary once + 4 plain fmacs (one DPP per 4 FMAs)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, int iters) {
  double a[4][16];
  const double f = 1e-9 * (threadIdx.x + 1);
  double nf[4] = {f, 2 * f, 3 * f, 0.0};
  for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) a[m][i] = threadIdx.x + i + 16 * m;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[m][i]) : "v"(nf[(m + 1) & 3]), "v"(nf[m]));
        if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[m][i]) : "v"(nf[(m + 1) & 3]), "v"(nf[m]));
        if (MODE == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[m][i]) : "v"(a[3][i]), "v"(nf[m]));
        if (MODE == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[m][i]) : "v"(a[3][i]), "v"(nf[m]));
      }
      if (MODE == 4) {
        double t;
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[3][i]));
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[m][i]) : "v"(t), "v"(nf[m]));
      }
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  double s = 0;
  for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) s += a[m][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE> void run(const char* name, double* d, long long* c, int blocks) {
  const int iters = 2000;
  long long h[2];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;
  printf("%-44s blocks %5d: %.2f clock64 ticks, %.2f ns per FMA instruction\n", name, blocks, h[0] / n, h[1] * 10.0 / n);
}
int main() {
  double* d; long long* c;
  hipMalloc(&d, sizeof(double) * 64 * 8192); hipMalloc(&c, 16);
  for (int blocks : {1, 1024, 2048, 4096}) {
    run<0>("v_fmac_f64 independent", d, c, blocks);
    run<1>("v_fmac_f64_dpp, unrelated source", d, c, blocks);
    run<2>("v_fmac_f64_dpp, d8_update pattern", d, c, blocks);
    run<3>("v_fmac_f64 (no DPP), same registers", d, c, blocks);
    run<4>("v_mov_b64_dpp + 4 v_fmac_f64", d, c, blocks);
  }
  return 0;
}
