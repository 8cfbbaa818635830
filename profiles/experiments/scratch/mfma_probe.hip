// scratch micro-benchmark: f64 MFMA shapes on gfx950 (not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k16(double* out, int iters) {
  v4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  v4 s = c0 + c1 + c2 + c3;
  if (s[0] + s[1] + s[2] + s[3] == 123.456) out[0] = s[0];
}
__global__ __launch_bounds__(256) void k4(double* out, int iters) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c7, 0, 0, 0);
  }
  double s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void kmix(double* out, int iters) {   // MFMA + independent DFMA in one wave
  v4 c0 = {0,0,0,0}, c1 = c0;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  double f[8]; for (int i = 0; i < 8; ++i) f[i] = a + i;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = __builtin_fma(f[i], 1.0000001, 1e-7);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = __builtin_fma(f[i], 1.0000001, 1e-7);
  }
  v4 s = c0 + c1; double t = 0; for (int i = 0; i < 8; ++i) t += f[i];
  if (s[0] + s[1] + s[2] + s[3] + t == 123.456) out[0] = s[0];
}
int main() {
  double* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = p.multiProcessorCount * wps;
    float ms;
    k16<<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k16<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    double n16 = 4.0 * iters * 4 * blocks;
    printf("16x16x4 waves/simd=%d: %.1f TF, %.1f ns per MFMA per SIMD\n", wps, n16 * 2048 / (ms * 1e-3) * 1e-12, ms * 1e6 / (4.0 * iters * wps));
    k4<<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k4<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    double n4 = 8.0 * iters * 4 * blocks;
    printf("4x4x4_4b waves/simd=%d: %.1f TF, %.1f ns per MFMA per SIMD\n", wps, n4 * 512 / (ms * 1e-3) * 1e-12, ms * 1e6 / (8.0 * iters * wps));
    kmix<<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); kmix<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    double fl = (2.0 * 2048 + 16.0 * 64 * 2) * iters * 4 * blocks;
    printf("mix (2 MFMA + 16 DFMA) waves/simd=%d: %.1f TF total, %.1f ns per loop trip per wave-slot\n", wps, fl / (ms * 1e-3) * 1e-12, ms * 1e6 / (iters * (double)wps));
  }
  return 0;
}
