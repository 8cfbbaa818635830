// qmps_overlap_amp.hip - the overlap circuit's amplitude for a GIVEN environment (the variational route of the reference).
//
// `get_overlap` (qmps/time_evolve_tools.py:95-131) and `obj_state` (qmps/new_time_evolve.py:223-247) do not solve for the fixed
// point of the mixed transfer map: they put a candidate environment q on the circuit's outer qubits (R = put_env_on_left_site(q),
// L = put_env_on_right_site(q^+)) and read psi[0] of the simulated register.  With q^ = q / ||q||_F that amplitude is the Rayleigh
// form of the map, psi[0] = 1/2 <q^, T(q^)>_F, T(x) = sum_s C_s x Bm_s^+, C = WW . merge(A, A), Bm = merge(B, B)
// (tests/test_oracle.py checks the identity against a state-vector pass of the circuit; for the exact fixed point it is eta / 2,
// SURVEY App. B-3).  One wave per candidate, any D in {2, 4, 8, 16}: five sweeps of D x D products through LDS, never the
// merged two-site tensors:   P_t = A_t q^,  X_{t1 t2} = A_{t1} P_{t2},  Y_s = sum_s' WW[s][s'] X_s',
//                            V_{s1} = sum_{s2} Y_{s1 s2} B_{s2}^+,  Z = sum_{s1} V_{s1} B_{s1}^+,  amplitude = 1/2 <q^, Z>.
// Not a hot path (a Nelder-Mead objective over 2 D^2 reals): no MFMA, operands straight from global memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"

namespace qmps {

namespace {
__device__ __forceinline__ double2 cmadd(double2 acc, double2 a, double2 b) {          // acc + a b
  return make_double2(acc.x + a.x * b.x - a.y * b.y, acc.y + a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmadd_conj(double2 acc, double2 a, double2 b) {     // acc + a conj(b)
  return make_double2(acc.x + a.x * b.x + a.y * b.y, acc.y + a.y * b.x - a.x * b.y);
}
}  // namespace

// p.x_in: the environments q [B][D][D]; p.eta: the amplitudes [B]; addressing of references / candidates as in the overlap launch
template <int D>
__global__ __launch_bounds__(64) void overlap_amplitude_kernel(OverlapArgs p) {
  constexpr int N = D * D, PER = N < 64 ? 1 : N / 64;
  __shared__ double2 sQ[N], sP[2][N], sX[4][N];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const double2* Ap = (const double2*)p.A + overlap_ref_index(p, b) * 2 * N;
  const double2* Bp = (const double2*)p.Bt + b * 2 * N;
  const double2* W = (const double2*)p.WW;
  const double2* q = (const double2*)p.x_in + b * N;
  double n2 = 0.0;
  for (int k = 0; k < PER; ++k) {
    const int e = lane + 64 * k;
    if (e < N) n2 += q[e].x * q[e].x + q[e].y * q[e].y;
  }
  n2 = wave_sum(n2);
  const double inv = n2 > 0.0 ? 1.0 / __builtin_sqrt(n2) : 0.0;
  for (int k = 0; k < PER; ++k) {
    const int e = lane + 64 * k;
    if (e < N) sQ[e] = make_double2(q[e].x * inv, q[e].y * inv);
  }
  __syncthreads();
  for (int k = 0; k < PER; ++k) {                    // P_t = A_t q^
    const int e = lane + 64 * k, i = e / D, j = e % D;
    if (e < N)
      for (int t = 0; t < 2; ++t) {
        double2 acc = make_double2(0.0, 0.0);
        for (int m = 0; m < D; ++m) acc = cmadd(acc, Ap[t * N + i * D + m], sQ[m * D + j]);
        sP[t][e] = acc;
      }
  }
  __syncthreads();
  for (int k = 0; k < PER; ++k) {                    // X_{t1 t2} = A_{t1} P_{t2}
    const int e = lane + 64 * k, i = e / D, j = e % D;
    if (e < N)
      for (int t = 0; t < 4; ++t) {
        double2 acc = make_double2(0.0, 0.0);
        for (int m = 0; m < D; ++m) acc = cmadd(acc, Ap[(t >> 1) * N + i * D + m], sP[t & 1][m * D + j]);
        sX[t][e] = acc;
      }
  }
  __syncthreads();
  for (int k = 0; k < PER; ++k) {                    // Y_s = sum_s' WW[s][s'] X_s' (entry by entry, in place)
    const int e = lane + 64 * k;
    if (e < N) {
      double2 x[4], y[4];
      for (int t = 0; t < 4; ++t) x[t] = sX[t][e];
      for (int s = 0; s < 4; ++s) {
        y[s] = make_double2(0.0, 0.0);
        for (int t = 0; t < 4; ++t) y[s] = cmadd(y[s], W[s * 4 + t], x[t]);
      }
      for (int s = 0; s < 4; ++s) sX[s][e] = y[s];
    }
  }
  __syncthreads();
  for (int k = 0; k < PER; ++k) {                    // V_{s1} = sum_{s2} Y_{s1 s2} B_{s2}^+
    const int e = lane + 64 * k, i = e / D, j = e % D;
    if (e < N)
      for (int s1 = 0; s1 < 2; ++s1) {
        double2 acc = make_double2(0.0, 0.0);
        for (int s2 = 0; s2 < 2; ++s2)
          for (int m = 0; m < D; ++m) acc = cmadd_conj(acc, sX[2 * s1 + s2][i * D + m], Bp[s2 * N + j * D + m]);
        sP[s1][e] = acc;
      }
  }
  __syncthreads();
  double ar = 0.0, ai = 0.0;
  for (int k = 0; k < PER; ++k) {                    // Z = sum_{s1} V_{s1} B_{s1}^+ and <q^, Z>
    const int e = lane + 64 * k, i = e / D, j = e % D;
    if (e < N) {
      double2 z = make_double2(0.0, 0.0);
      for (int s1 = 0; s1 < 2; ++s1)
        for (int m = 0; m < D; ++m) z = cmadd_conj(z, sP[s1][i * D + m], Bp[s1 * N + j * D + m]);
      const double2 c = sQ[e];
      ar += c.x * z.x + c.y * z.y;
      ai += c.x * z.y - c.y * z.x;
    }
  }
  ar = wave_sum(ar);
  ai = wave_sum(ai);
  if (lane == 0) ((double2*)p.eta)[b] = make_double2(0.5 * ar, 0.5 * ai);
}

hipError_t launch_overlap_amplitude(int D, const OverlapArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const dim3 grid((unsigned)a.B), block(64);
  switch (D) {
    case 2: hipLaunchKernelGGL(overlap_amplitude_kernel<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(overlap_amplitude_kernel<4>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(overlap_amplitude_kernel<8>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(overlap_amplitude_kernel<16>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
