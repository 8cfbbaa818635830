// qmps_evolve_d4.hip - the whole BFGS time evolution at bond dimension D = 4 in one launch (gfx950 only): ONE WORKGROUP owns a trajectory,
// its WAVES are the candidates of an evaluation pass.
//
// Reference: qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 (`minimize(obj, params, (A_, WW))` per time step).  The
// host-loop driver (qmps_evolve_bfgs) spends 15 BFGS iterations x ~0.1 ms of round trip per time step around ~0.05 ms of kernels
// at D = 4 (profiles/r03h_evolve_d4_t256.json: kernel share of wall 0.3).  Here, as in qmps_evolve_d2.hip, the optimiser never
// leaves the device: wave w of the workgroup builds the tensor of candidate w (wave 0 the point itself, 1 .. P  +h e_k,
// P + 1 .. 2P  -h e_k, then the backtracking points) - four lanes simulate the four columns of the 8 x 8 ansatz unitary - and
// eigen-solves its mixed transfer map, one complex 16 x 16 tile SQUARED on the matrix cores until it is rank one
// (qmps_overlap_d4.h: the code of overlap_square_d4_kernel); the 2 P + 1 + (n_alphas - 1) solves of a pass run side by side
// and cost the latency of one.  The optimiser loop itself (qmps_evolve_core.h) is the D = 2 kernel's, with workgroup barriers.
// Every candidate is eigen-solved (no two-sided first-order gradient as in the host driver: with a wave per candidate the 2 P
// neighbours cost nothing extra, and the gradient is the plain central difference scipy would form from exact objectives).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_circuit.h"
#include "qmps_overlap_d4.h"
#include "qmps_evolve_core.h"

namespace qmps {

namespace {
constexpr int NWMAX = 12;        // waves per workgroup = candidates per pass (12 waves = 3 per SIMD: 170 registers each, the squaring needs ~140)
}

template <int KIND>
__global__ __launch_bounds__(64 * NWMAX) void evolve_bfgs_d4_kernel(EvolveD2Args p) {
  constexpr int PMAX = kEvolvePMax;
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, P = p.P, G1 = 2 * P + 1;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64];
  __shared__ int sOK[64];
  __shared__ double2 sA[32], sB[NWMAX][32], sT[NWMAX][kSquareD4Scratch];
  __shared__ double sCnt[4];
  const double2* W = (const double2*)p.WW;
  const double tol2 = p.tol * p.tol;
  if (tid < 4) sCnt[tid] = 0.0;
  // the four columns of the ansatz unitary for the parameter vector par(l), by lanes 0 .. 3 of the calling wave, as the tensor [2][4][4]
  auto build_tensor = [&](double2* out, auto par) {
    if (lane < 4) {
      Reg<3> r;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        r.re[x] = (x == lane) ? 1.0 : 0.0;
        r.im[x] = 0.0;
      }
      ansatz_circuit<3, KIND>(r, par, P);
#pragma unroll
      for (int x = 0; x < 8; ++x) out[((x & 1) * 4 + (x >> 1)) * 4 + lane] = make_double2(r.re[x], r.im[x]);      // A[s][i][j] = amplitude[2 i + s] of column j
    }
    __builtin_amdgcn_wave_barrier();
  };
  // one evaluation pass.  coef finite: candidates 0 .. 2P are the central-difference columns of z = x + coef d, then n_ladder
  // backtracking points; coef NaN: the n_ladder backtracking points only (they keep their candidate numbers G1 ..)
  auto evaluate = [&](double coef, int n_ladder) {
    const bool with_grad = coef == coef;
    const int cand = with_grad ? wave : G1 + wave;
    const bool mine = with_grad ? wave < G1 + n_ladder : wave < n_ladder;
    if (mine) {
      const bool grad = cand < G1;
      const double a = grad ? coef : p.alphas[cand - G1 + 1];
      const int isel = (grad && cand > 0) ? (cand - 1) % P : -1;
      const double hs = cand <= P ? p.h : -p.h;
      build_tensor(sB[wave], [&](int l) {
        double v = a != 0.0 ? evolve_detail::add_rn(sX[l], evolve_detail::mul_rn(a, sD[l])) : sX[l];
        if (l == isel) v = evolve_detail::add_rn(v, hs);
        return v;
      });
      double eta_r, eta_i;
      int rounds, status;
      v4f64 mr, mi;
      overlap_square_d4_item(sA, sB[wave], W, sT[wave], p.max_rounds, tol2, eta_r, eta_i, rounds, status, mr, mi);
      if (lane == 0) {
        sF[cand] = -__builtin_sqrt(__builtin_sqrt(eta_r * eta_r + eta_i * eta_i));
        sOK[cand] = status == QMPS_ST_OK ? 1 : 0;
        atomicAdd(&sCnt[1], (double)rounds);
        if (status != QMPS_ST_OK) atomicAdd(&sCnt[2], 1.0);
      }
    }
    __syncthreads();
    if (tid == 0) sCnt[0] += (double)((with_grad ? G1 : 0) + n_ladder);
  };
  auto build_reference = [&]() {
    if (wave == 0) build_tensor(sA, [&](int l) { return sX[l]; });
    __syncthreads();
  };
  BfgsLds L;
  L.X = sX; L.G = sG; L.D = sD; L.S = sS; L.Gn = sGn; L.Hy = sHy; L.H = sH; L.F = sF; L.OK = sOK;
  __syncthreads();
  const bool ladder_in_pass = (int)(blockDim.x >> 6) >= G1 + (p.NA - 1);
  bfgs_time_evolution(p, t, (wave == 0 && lane < P) ? lane : -1, tid == 0, L, evaluate, build_reference, [] { __syncthreads(); }, ladder_in_pass);
  __syncthreads();
  if (tid == 0) {
    if (p.nfev != nullptr) p.nfev[t] = sCnt[0];
    if (p.rounds != nullptr) p.rounds[t] = sCnt[1];
    if (p.fail != nullptr) p.fail[t] = (int32_t)sCnt[2];
  }
}

hipError_t launch_evolve_bfgs_d4(int kind, const EvolveD2Args& a, hipStream_t st) {
  if (a.T <= 0) return hipSuccess;
  const int G1 = 2 * a.P + 1, G = a.NA - 1;
  if (a.P < 1 || a.P > kEvolvePMax || a.NA < 1 || a.NA > kEvolveMaxAlphas || G1 > NWMAX || G > NWMAX || G1 + G > 64) return hipErrorInvalidValue;
  // candidates of a pass = waves: the backtracking points ride along with the gradient pass where they fit into 16 waves
  const int waves = G1 + G <= NWMAX ? G1 + G : (G1 > G ? G1 : G);
  const dim3 grid((unsigned)a.T), block(64 * waves);
  switch (kind) {
    case 0: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<1>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<3>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
