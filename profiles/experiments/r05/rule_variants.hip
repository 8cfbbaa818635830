// GPU experiment: the device build of qmps_roto_rule.h against scipy's recorded answers under different compile options.
// usage: rule_variants fits.bin n   (fits.bin: n x 7 doubles a b c d x f nfev)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "qmps_roto_rule.h"
__global__ void k(const double* F, int n, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = qmps::double_sinusoid_step(F[7 * i], F[7 * i + 1], F[7 * i + 2], F[7 * i + 3], 0);
}
int main(int argc, char** argv) {
  int n = atoi(argv[2]);
  std::vector<double> F(7 * n), out(n);
  FILE* f = fopen(argv[1], "rb");
  if (fread(F.data(), 8, 7 * n, f) != (size_t)7 * n) return 1;
  fclose(f);
  double *dF, *dO;
  hipMalloc(&dF, 56 * n); hipMalloc(&dO, 8 * n);
  hipMemcpy(dF, F.data(), 56 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), 0, 0, dF, n, dO);
  hipMemcpy(out.data(), dO, 8 * n, hipMemcpyDeviceToHost);
  int bad = 0; double mx = 0;
  for (int i = 0; i < n; ++i) { double d = fabs(out[i] - F[7 * i + 4]); if (d > 1e-9) ++bad; if (d > mx) mx = d; }
  // host build of the same source in the same binary
  int badh = 0;
  for (int i = 0; i < n; ++i) { double d = fabs(qmps::double_sinusoid_step(F[7 * i], F[7 * i + 1], F[7 * i + 2], F[7 * i + 3], 0) - F[7 * i + 4]); if (d > 1e-9) ++badh; }
  printf("device: %d of %d differ from scipy by > 1e-9 (max %.3g); host build in the same binary: %d\n", bad, n, mx, badh);
  return 0;
}
