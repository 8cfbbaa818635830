// scratch micro-benchmark: the complex 16 x 16 x 16 product of the D = 16 kernels as 48 v_mfma_f64_4x4x4_4b (three real products,
// fixed A operand in 48 distinct registers) against 12 v_mfma_f64_16x16x4, one wave per SIMD and more (not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void k444(const double* in, double* out, int iters) {
  double ps[4][4], pr[4][4], pi[4][4];
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int K = 0; K < 4; ++K) {
      pr[I][K] = in[(I * 4 + K) * 64 + l];
      pi[I][K] = in[1024 + (I * 4 + K) * 64 + l];
      ps[I][K] = pr[I][K] + pi[I][K];
    }
  v4 qre = {in[l], in[64 + l], in[128 + l], in[192 + l]}, qim = {in[256 + l], in[320 + l], in[384 + l], in[448 + l]};
  v4 cre = {0, 0, 0, 0}, cim = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    v4 k1 = {0, 0, 0, 0}, k2 = k1, k3 = k1;
#pragma unroll
    for (int K = 0; K < 4; ++K) {
      double qd, qs;
      if (VARIANT >= 1) { qd = qim[K] - qre[K]; qs = qre[K] + qim[K]; } else { qd = qim[K]; qs = qre[K]; }
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        k1[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(ps[I][K], qre[K], k1[I], 0, 0, 0);
        k2[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[I][K], qd, k2[I], 0, 0, 0);
        k3[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[I][K], qs, k3[I], 0, 0, 0);
      }
    }
    if (VARIANT >= 2) {        // the result feeds the next product (as in a power step)
      cre = k1 - k3;
      cim = k1 + k2;
      qre = cre * 0.25;
      qim = cim * 0.25;
    } else {
      cre += k1 - k3;
      cim += k1 + k2;
    }
  }
  double s = cre[0] + cre[1] + cre[2] + cre[3] + cim[0] + cim[1] + cim[2] + cim[3];
  if (s == 123.456) out[0] = s;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k16(const double* in, double* out, int iters) {
  double pr[4], pi[4];
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int K = 0; K < 4; ++K) { pr[K] = in[K * 64 + l]; pi[K] = in[1024 + K * 64 + l]; }
  v4 qre = {in[l], in[64 + l], in[128 + l], in[192 + l]}, qim = {in[256 + l], in[320 + l], in[384 + l], in[448 + l]};
  v4 cre = {0, 0, 0, 0}, cim = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    v4 k1 = {0, 0, 0, 0}, k2 = k1, k3 = k1;
#pragma unroll
    for (int K = 0; K < 4; ++K) {
      const double ps = pr[K] + pi[K], qd = qim[K] - qre[K], qs = qre[K] + qim[K];
      k1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ps, qre[K], k1, 0, 0, 0);
      k2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[K], qd, k2, 0, 0, 0);
      k3 = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[K], qs, k3, 0, 0, 0);
    }
    if (VARIANT >= 2) {
      cre = k1 - k3;
      cim = k1 + k2;
      qre = cre * 0.25;
      qim = cim * 0.25;
    } else {
      cre += k1 - k3;
      cim += k1 + k2;
    }
  }
  double s = cre[0] + cre[1] + cre[2] + cre[3] + cim[0] + cim[1] + cim[2] + cim[3];
  if (s == 123.456) out[0] = s;
}

template <class F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f(100);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventRecord(e0);
  f(4000);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double *d, *in;
  (void)hipMalloc(&d, 64);
  (void)hipMalloc(&in, 4096 * 8);
  double h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 1e-3 * ((i * 37) % 101 - 50);
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = p.multiProcessorCount * wps;
    float a0 = timeit([&](int n) { k444<0><<<blocks, 256>>>(in, d, n); });
    float a1 = timeit([&](int n) { k444<1><<<blocks, 256>>>(in, d, n); });
    float a2 = timeit([&](int n) { k444<2><<<blocks, 256>>>(in, d, n); });
    float b1 = timeit([&](int n) { k16<1><<<blocks, 256>>>(in, d, n); });
    float b2 = timeit([&](int n) { k16<2><<<blocks, 256>>>(in, d, n); });
    printf("waves/SIMD %d: complex 16^3 product, ns per product per wave slot: 4x4x4 pure %.0f, + B-side adds %.0f, + dependent chain %.0f | 16x16x4 accumulate %.0f, dependent chain %.0f\n",
           wps, a0 * 1e6 / 4000 / wps, a1 * 1e6 / 4000 / wps, a2 * 1e6 / 4000 / wps, b1 * 1e6 / 4000 / wps, b2 * 1e6 / 4000 / wps);
  }
  return 0;
}
