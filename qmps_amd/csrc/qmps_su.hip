// qmps_su.hip - SU(N) parameters -> unitary / state tensor on the device (gfx950 only), N = 2 D in {4, 8, 16, 32}.
//
// Reference: NonSparseFullEnergyOptimizer / NonSparseFullTwoSiteEnergyOptimizer build their state unitaries as
// `U = SU(u_params, 2 D)` / `U4(p)` (qmps/ground_state.py:245, 251-266, 299-307; scripts/bond_dimension.py:21-50 sweeps D = 2 .. 16:
// 1 023 parameters at D = 16) - a matrix exponential per evaluation on the host.  `SU` comes from xmps.spin, which is not in the
// reference tree: the convention here is the host mirror's (qmps_amd/ground_state.py:SU, documentation-pinned):
//     U = exp(-i/2 sum_k p_k G_k),  G_k = generalised Gell-Mann matrices of su(N) in the order
//     (a < b: symmetric, antisymmetric) for a = 0 .. N-1, b = a+1 .. N-1, then the N - 1 diagonal ones.
//
// One workgroup of N x N threads per evaluation (N = 4: four evaluations per wave), thread (i, j) owns entry [i][j]:
//   X = -i/2 sum_k p_k G_k  (entries read straight off the parameter vector), scaled by 2^-s so that ||X||_F <= 1/4,
//   T = Taylor polynomial of degree 13 by Horner's rule (12 products, truncation error < 1e-19), s squarings.
// Products through padded LDS tiles.  Output: the full unitary (qmps_su_unitaries, the two-site unit cell) or directly the
// state tensor A[s][i][j] = U[2 i + s][j], j < D (unitary_to_tensor, qmps/tools.py:151-154, fused).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"

namespace qmps {

namespace {

__device__ __forceinline__ void cfma(double2 a, double2 b, double2& c) {   // c += a b
  c.x = dfma(a.x, b.x, c.x);
  c.x = dfma(-a.y, b.y, c.x);
  c.y = dfma(a.x, b.y, c.y);
  c.y = dfma(a.y, b.x, c.y);
}

}  // namespace

// out_tensor != 0: A[b][2][N/2][N/2] (the first N/2 columns of U, rows split as 2 i + s);  else U[b][N][N]
template <int N>
__global__ __launch_bounds__((N * N < 64) ? 64 : N * N) void su_exp_kernel(const double* __restrict__ params, int64_t B, int stride_params,
                                                                        double2* __restrict__ out, int out_tensor) {
  constexpr int NN = N * N, P = N + 1, THREADS = NN < 64 ? 64 : NN, ITEMS = THREADS / NN, WAVES = THREADS / 64;
  __shared__ double2 sXm[ITEMS][N][P], sT[2][ITEMS][N][P];
  __shared__ double red[WAVES > 1 ? WAVES : 1];
  __shared__ int s_scale;
  const int tid = threadIdx.x, e = tid / NN, l = tid % NN, i = l / N, j = l % N;
  const int64_t b = (int64_t)blockIdx.x * ITEMS + e;
  const int64_t bb = b < B ? b : B - 1;           // surplus lanes of the last workgroup shadow a real evaluation
  const double* p = params + bb * stride_params;
  // ---- X[i][j] = -i/2 M[i][j],  M = sum_k p_k G_k
  double2 x;
  if (i != j) {
    const int a = i < j ? i : j, c = i < j ? j : i;
    const int pair = a * N - a * (a + 1) / 2 + (c - a - 1);
    const double ps = p[2 * pair], pa = p[2 * pair + 1];
    // M[a][c] = ps - i pa, M[c][a] = ps + i pa;  -i/2 (mr + i mi) = (mi - i mr)/2
    const double mr = ps, mi = i < j ? -pa : pa;
    x = make_double2(0.5 * mi, -0.5 * mr);
  } else {
    const double* pd = p + N * (N - 1);
    double d = 0.0;
    for (int k = 1; k < N; ++k) {
      const double ck = __builtin_sqrt(2.0 / ((double)k * (k + 1)));
      d += pd[k - 1] * ck * (i < k ? 1.0 : (i == k ? -(double)k : 0.0));
    }
    x = make_double2(0.0, -0.5 * d);
  }
  // ---- scaling: ||X||_F 2^-s <= 1/4 (the workgroup's largest s serves all of its evaluations)
  {
    double f2 = x.x * x.x + x.y * x.y;
    if constexpr (NN == 16) f2 = row16_sum(f2);
    else f2 = wave_sum(f2);
    if (tid == 0) s_scale = 0;
    __syncthreads();
    if constexpr (WAVES > 1) {
      if ((tid & 63) == 0) red[tid >> 6] = f2;
      __syncthreads();
      f2 = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) f2 += red[w];
    }
    const double fro = __builtin_sqrt(f2);
    int s = 0;
    if (fro > 0.25) s = (int)ceil(log2(fro * 4.0));
    if (s > 60) s = 60;
    if (l == 0) atomicMax(&s_scale, s);
    __syncthreads();
  }
  const int s = s_scale;
  const double sc = ldexp(1.0, -s);
  x.x *= sc;
  x.y *= sc;
  sXm[e][i][j] = x;
  // ---- Horner: T_13 = 1 + X/13;  T_k = 1 + (X T_{k+1})/k
  double2 t = make_double2((i == j ? 1.0 : 0.0) + x.x * (1.0 / 13.0), x.y * (1.0 / 13.0));
  int cur = 0;
  sT[cur][e][i][j] = t;
  __syncthreads();
  for (int k = 12; k >= 1; --k) {
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll 8
    for (int q = 0; q < N; ++q) cfma(sXm[e][i][q], sT[cur][e][q][j], acc);
    const double inv = 1.0 / (double)k;
    t = make_double2((i == j ? 1.0 : 0.0) + acc.x * inv, acc.y * inv);
    cur ^= 1;
    sT[cur][e][i][j] = t;
    __syncthreads();
  }
  for (int q2 = 0; q2 < s; ++q2) {
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll 8
    for (int q = 0; q < N; ++q) cfma(sT[cur][e][i][q], sT[cur][e][q][j], acc);
    t = acc;
    cur ^= 1;
    sT[cur][e][i][j] = t;
    __syncthreads();
  }
  if (b >= B) return;
  if (out_tensor) {
    constexpr int D = N / 2;
    if (j < D) out[b * (2 * D * D) + ((i & 1) * D + (i >> 1)) * D + j] = t;      // A[s][i'][j] = U[2 i' + s][j]
  } else {
    out[b * NN + l] = t;
  }
}

hipError_t launch_su_exp(int N, const double* params, int64_t B, int stride_params, void* out, int out_tensor, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  switch (N) {
    case 4: hipLaunchKernelGGL((su_exp_kernel<4>), dim3((unsigned)((B + 3) / 4)), dim3(64), 0, st, params, B, stride_params, (double2*)out, out_tensor); break;
    case 8: hipLaunchKernelGGL((su_exp_kernel<8>), dim3((unsigned)B), dim3(64), 0, st, params, B, stride_params, (double2*)out, out_tensor); break;
    case 16: hipLaunchKernelGGL((su_exp_kernel<16>), dim3((unsigned)B), dim3(256), 0, st, params, B, stride_params, (double2*)out, out_tensor); break;
    case 32: hipLaunchKernelGGL((su_exp_kernel<32>), dim3((unsigned)B), dim3(1024), 0, st, params, B, stride_params, (double2*)out, out_tensor); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
