// qmps_energy_block.hip - D = 8 (and the D = 16 fall-back) energy path: one evaluation per workgroup of D x D threads, tiles in
// LDS; at D = 8 with the direct fixed-point solve (qmps_direct_d8.h) in front, in the same launch.  Split out of qmps_kernels.hip in round 3.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_knobs.h"
#include "qmps_device.h"
#include "qmps_direct_d8.h"

namespace qmps {

// ------------------------------------------------------------------------------------------
// Kernel 2: D = 8, 16 - one evaluation per workgroup of D x D threads, tiles in LDS.
// Thread (i, j) owns r[i][j].  First correct version of the large-D path.
// ------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  // sum over the D*D threads of the workgroup; result broadcast to every thread
  constexpr int N = D * D;
  v = wave_sum(v);   // DPP row rotations + permlane swaps: VALU only, ~20 instructions
  if (N > 64) {
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = 0.0;
#pragma unroll
    for (int w = 0; w < N / 64; ++w) v += red[w];
  }
  return v;
}

// FUSED8 (D = 8, SOLVE): the direct fixed-point solve (qmps_direct_d8.h) runs in front, in the same wave - solve, acceptance
// step and energies in ONE launch (the small batches of BASELINE.json configs[3] are all launch latency).  A separate
// instantiation: the solve needs ~190 VGPRs, the plain block kernel runs four waves per SIMD.
template <int D, bool SOLVE, bool FUSED8 = false>
__global__ __launch_bounds__(D* D) void energy_block_kernel(LaneArgs p) {
  constexpr int N = D * D;
  constexpr int P = D + 1;  // padded row (in double2 units) against bank conflicts
  __shared__ double2 sA[2][D][P];
  __shared__ double2 sR[D][P];
  __shared__ double2 sX[2][D][P];
  __shared__ double2 sT[2][D][P];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int i = tid / D, j = tid % D;
  const int64_t b = blockIdx.x;
  if (b >= p.B) return;
  if (p.acc_zero != nullptr && blockIdx.x == 0) acc_clear(p.acc_zero, p.n_terms, tid, N);   // accumulator of a later step
  // Finishing pass behind the Krylov fall-back (D = 8, round 5; the protocol of energy_mfma_d16_kernel): only evaluations the fall-back solved
  // (status PENDING, fixed point in r_in); every workgroup that finds work pending takes an exit ticket, the last one clears the counters.
  int iters0 = 0;
  if (SOLVE && p.only_pending) {
    if (__hip_atomic_load(p.kry_counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;      // nothing to finish
    if (p.status[b] != QMPS_ST_PENDING) {
      if (tid == 0 && atomicAdd(p.kry_counter + 4, 1) == (int)gridDim.x - 1) {
        __hip_atomic_store(p.kry_counter + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.kry_counter + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    iters0 = p.iters[b];
  }

  {
    const double2* a = (const double2*)p.A + b * (2 * N);
    sA[0][i][j] = a[tid];
    sA[1][i][j] = a[N + tid];
  }
  // The Hamiltonian terms are fetched NOW, into registers, and parked in LDS just before the energy stage: fetched where they
  // are used they cost an HBM / L2 round trip (~1 us of a 16 us evaluation at D = 8) at the very end of the kernel.
  constexpr int kHMax = 16 * 16, HR = (kHMax + N - 1) / N;      // kMaxTerms (qmps_ctx.h) x 16 entries
  __shared__ double2 sH[kHMax];
  double2 hreg[HR];
#pragma unroll
  for (int u = 0; u < HR; ++u) {
    const int l = tid + u * N;
    hreg[u] = l < 16 * p.n_terms ? ((const double2*)p.h)[l] : make_double2(0.0, 0.0);
  }
  double2 r;
  if constexpr (FUSED8) {
    __shared__ double sM8[64][17];
    __shared__ double sT8[8][9];
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    r = env_direct_d8_solve(sA, sT8, sM8, tid);
  } else if (p.r_in != nullptr) {
    const double2* g = (const double2*)p.r_in + b * N;
    const double2 u = g[i * D + j], l = g[j * D + i];
    r = make_double2(0.5 * (u.x + l.x), (i == j) ? 0.0 : 0.5 * (u.y - l.y));
    const double tr = block_sum<D>(i == j ? r.x : 0.0, red, tid);
    if (tr > 1e-300 && tr < 1e300) {
      r.x /= tr;
      r.y /= tr;
    } else {      // (zeros / NaN where nobody stored an environment: no guess, the default start)
      r = make_double2(i == j ? 1.0 / D : 0.0, 0.0);
    }
  } else {
    r = make_double2(i == j ? 1.0 / D : 0.0, 0.0);
  }
  sR[i][j] = r;
  __syncthreads();

  // D = 8: row i of A (for X = A r) lives in registers; row j (for r' = X A^+), the r column and the X row
  // come from LDS (40 instead of 56 ds_read_b128 per step) - keeps the kernel at 4 waves per SIMD, which
  // matters more here: the step is latency-bound (two LDS round trips + two reductions per step).  D = 16 keeps them in LDS (register budget);
  // that instantiation is only the fallback behind the MFMA kernel.
  constexpr bool kRowsInRegs = (D == 8);
  constexpr int RD = kRowsInRegs ? D : 1;
  double2 ai_[2][RD];
  if constexpr (kRowsInRegs) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) ai_[s][k] = sA[s][i][k];
  }
  auto apply = [&](double2& out) {
    // X_s[i][j] = sum_k A_s[i][k] r[k][j]
    double2 rc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) rc[k] = sR[k][j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double2 a;
        if constexpr (kRowsInRegs) a = ai_[s][k]; else a = sA[s][i][k];
        const double2 rr = rc[k];
        xr = dfma(a.x, rr.x, xr);
        xr = dfma(-a.y, rr.y, xr);
        xi = dfma(a.x, rr.y, xi);
        xi = dfma(a.y, rr.x, xi);
      }
      sX[s][i][j] = make_double2(xr, xi);
    }
    __syncthreads();
    double nr = 0.0, ni = 0.0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 a = sA[s][j][k];
        const double2 x = sX[s][i][k];
        nr = dfma(x.x, a.x, nr);
        nr = dfma(x.y, a.y, nr);
        ni = dfma(x.y, a.x, ni);
        ni = dfma(-x.x, a.y, ni);
      }
    out = make_double2(nr, ni);
  };

  int iters = 0, status = QMPS_ST_OK;
  if (SOLVE) {
    status = QMPS_ST_NOT_CONVERGED;
    const double tol2 = p.tol * p.tol;
    int k_ref = 0;
    float l_ref = 0.0f;
    for (int k = 1; k <= p.max_iter; ++k) {
      double2 n;
      apply(n);
      // hermitise through LDS, normalise by the trace
      sT[0][i][j] = n;
      __syncthreads();
      const double2 m = sT[0][j][i];
      n = make_double2(0.5 * (n.x + m.x), (i == j) ? 0.0 : 0.5 * (n.y - m.y));
      const double tr = block_sum<D>(i == j ? n.x : 0.0, red, tid);
      const double inv = 1.0 / tr;
      n.x *= inv;
      n.y *= inv;
      const double dr = n.x - r.x, di = n.y - r.y;
      const double d2 = block_sum<D>(dr * dr + di * di, red, tid);
      r = n;
      sR[i][j] = r;
      __syncthreads();
      iters = k;
      if (d2 < tol2) {
        status = QMPS_ST_OK;
        break;
      }
      // a long tail ahead (|lambda_2| close to 1 and no accepted direct solve): hand the evaluation to the Krylov fall-back
      // (status PENDING, iterate in r_out) - d2 is uniform over the workgroup
      if (k < p.max_iter && p.kry_counter != nullptr && power_gives_up(k, d2, tol2, p.krylov_after, k_ref, l_ref)) {
        status = QMPS_ST_PENDING;
        break;
      }
    }
  }
  const bool given = SOLVE && status == QMPS_ST_PENDING;
  if (!SOLVE && p.check_pd) status = p.status[b];
  if (SOLVE || p.check_pd) {
    if (status == QMPS_ST_OK) {
      // Positive definiteness (the criterion of cholesky(r), qmps/tools.py:182): the pivots of LDL^H, all D^2 threads at
      // once - thread (i, j) owns the Schur-complement entry S[i][j]; per pivot one LDS round trip (column c and the pivot),
      // S[i][j] -= S[i][c] conj(S[j][c]) / S[c][c].  (A single thread walking the Cholesky recurrence through LDS took
      // ~6 us of the 22 us this kernel needs per evaluation at D = 8.)
      double2 S = r;
      bool ok = true;
      for (int c = 0; c < D; ++c) {
        __syncthreads();
        sT[1][i][j] = S;
        __syncthreads();
        const double pc = sT[1][c][c].x;
        ok = ok && (pc > 0.0);
        const double2 li = sT[1][i][c], lj = sT[1][j][c];
        const double inv = fast_rcp(pc);
        const double wr = (li.x * lj.x + li.y * lj.y) * inv, wi = (li.y * lj.x - li.x * lj.y) * inv;
        S.x -= wr;
        S.y -= wi;
      }
      if (!ok) status = QMPS_ST_NOT_PD;
      __syncthreads();
    }
  }

  // ---- energy: rho[tau][sigma] = tr(B_tau r B_sigma^+), B_(2 t1 + t2) = A_t1 A_t2 - the merged two-site tensor (qmps/tools.py:432-433).
  // Two stages: the four B_tau (tiles in sX, sT), then Y_tau = B_tau r and this thread's share Y_tau[i][j] conj(B_sigma[i][j]).
  // (The first version went A_t2 r -> (.) A_s2^+ -> A_t1 (.) per (t2, s2): 112 complex multiply-adds, ~240 LDS reads and nine
  // barriers per evaluation instead of 64, 56 and two; at one wave per SIMD an evaluation costs its instruction count.)
  const double trr = block_sum<D>(i == j ? r.x : 0.0, red, tid);
  double2 rho_loc[4][4];
  {
    double2 bt[4];
    {
      double2 acol[2][D];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < D; ++k) acol[s][k] = sA[s][k][j];
#pragma unroll
      for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          double br = 0.0, bi = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            double2 a;
            if constexpr (kRowsInRegs) a = ai_[t1][k]; else a = sA[t1][i][k];
            const double2 c = acol[t2][k];
            br = dfma(a.x, c.x, br);
            br = dfma(-a.y, c.y, br);
            bi = dfma(a.x, c.y, bi);
            bi = dfma(a.y, c.x, bi);
          }
          bt[2 * t1 + t2] = make_double2(br, bi);
        }
    }
    __syncthreads();                       // the tiles are free: acceptance step and LDL^H are done with them
#pragma unroll
    for (int u = 0; u < HR; ++u) sH[tid + u * N] = hreg[u];
    sX[0][i][j] = bt[0];
    sX[1][i][j] = bt[1];
    sT[0][i][j] = bt[2];
    sT[1][i][j] = bt[3];
    __syncthreads();
    double2 rc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) rc[k] = sR[k][j];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double yr = 0.0, yi = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 b2 = t < 2 ? sX[t][i][k] : sT[t - 2][i][k], rr = rc[k];
        yr = dfma(b2.x, rr.x, yr);
        yr = dfma(-b2.y, rr.y, yr);
        yi = dfma(b2.x, rr.y, yi);
        yi = dfma(b2.y, rr.x, yi);
      }
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        rho_loc[t][sg] = make_double2(yr * bt[sg].x + yi * bt[sg].y, yi * bt[sg].x - yr * bt[sg].y);
    }
  }
  const double inv_tr = 1.0 / trr;
  if (p.rho_out != nullptr) {
    // the density matrix itself is wanted: 16 complex sums over the workgroup
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double sr = block_sum<D>(rho_loc[t][s].x, red, tid);
        const double si = block_sum<D>(rho_loc[t][s].y, red, tid);
        rho_loc[t][s] = make_double2(sr * inv_tr, si * inv_tr);
      }
  }
  for (int q = 0; q < p.n_terms; ++q) {
    // E_q = Re sum h_q[s][t] rho[t][s] is linear in rho: combine the thread's shares first, ONE sum over the workgroup per
    // term instead of 32 (with the density matrix already summed every thread holds the total: no sum at all)
    const double2* h = sH + q * 16;
    double e = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double2 hv = h[s * 4 + t];
        e = dfma(hv.x, rho_loc[t][s].x, e);
        e = dfma(-hv.y, rho_loc[t][s].y, e);
      }
    if (p.rho_out == nullptr) e = block_sum<D>(e, red, tid) * inv_tr;
    if (tid == 0 && !given) {      // (a handed-over evaluation: energy and arrival at the accumulator come from the finishing pass)
      p.E[b * p.n_terms + q] = e;
      if (p.acc != nullptr) acc_arrive(p.acc, p.acc_shards, q, (unsigned)b, e, p.acc_bound, p.acc_scale);   // exact in-kernel cost
    }
  }
  if (tid == 0) {
    if (SOLVE) {
      p.iters[b] = iters0 + iters;
      p.status[b] = status;
      if (given) atomicAdd(p.kry_counter + 2, 1);
    } else if (p.check_pd) {
      p.status[b] = status;
    }
    if (p.rho_out != nullptr && !given) {
      double2* o = (double2*)p.rho_out + b * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) o[t * 4 + s] = rho_loc[t][s];
    }
  }
  if (p.r_out != nullptr && SOLVE) ((double2*)p.r_out)[b * N + tid] = r;
  if (SOLVE && p.only_pending) {       // exit ticket of a finishing pass (see the top of the kernel)
    if (tid == 0 && atomicAdd(p.kry_counter + 4, 1) == (int)gridDim.x - 1) {
      __hip_atomic_store(p.kry_counter + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.kry_counter + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int D>
static hipError_t launch_block(const LaneArgs& a, bool solve, hipStream_t st) {
  if (solve && D == 8 && a.direct != 0) {
    if constexpr (D == 8) hipLaunchKernelGGL((energy_block_kernel<8, true, true>), dim3((unsigned)a.B), dim3(64), 0, st, a);
  } else if (solve)
    hipLaunchKernelGGL((energy_block_kernel<D, true>), dim3((unsigned)a.B), dim3(D * D), 0, st, a);
  else
    hipLaunchKernelGGL((energy_block_kernel<D, false>), dim3((unsigned)a.B), dim3(D * D), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_energy_block(int D, const LaneArgs& a, bool solve, hipStream_t st) {
  switch (D) {
    case 8: return launch_block<8>(a, solve, st);
    case 16: return launch_block<16>(a, solve, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace qmps
