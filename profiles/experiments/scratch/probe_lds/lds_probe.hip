// does a workgroup get more than 64 KiB of static LDS on gfx950?  (the Krylov fall-back wants ~150 KiB per candidate)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KB>
__global__ __launch_bounds__(256) void k(double* o) {
  __shared__ double s[KB * 128];
  for (int i = threadIdx.x; i < KB * 128; i += 256) s[i] = (double)i + o[0];
  __syncthreads();
  double t = 0;
  for (int i = threadIdx.x; i < KB * 128; i += 256) t += s[(i * 7 + 3) % (KB * 128)];
  o[1 + blockIdx.x * 256 + threadIdx.x] = t;
}
template <int KB> void run(double* d) {
  hipLaunchKernelGGL(k<KB>, dim3(512), dim3(256), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  double h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%d KiB: launch %s, sync %s, out %g\n", KB, hipGetErrorString(hipGetLastError()), hipGetErrorString(e), h[1]);
}
int main() {
  double* d;
  hipMalloc(&d, (1 + 512 * 256) * 8);
  hipMemset(d, 0, (1 + 512 * 256) * 8);
  run<48>(d); run<64>(d); run<96>(d); run<128>(d); run<150>(d); run<160>(d);
  return 0;
}
