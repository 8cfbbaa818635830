// qmps_energy_d16.hip - D = 16 energy path on the matrix cores (gfx950 only): power iteration + LDL^H test + energy epilogue,
// one wave (energy_mfma_d16_kernel) or two waves (energy_mfma_d16x2_kernel, small batches) per evaluation; complex 16 x 16 x 16
// products as v_mfma_f64_16x16x4 chains, register to register.  Split out of qmps_kernels.hip in round 3.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_knobs.h"
#include "qmps_device.h"

namespace qmps {

// ------------------------------------------------------------------------------------------
// Kernel 1d: D = 16 power iteration on the matrix cores, ONE WAVE PER EVALUATION, no LDS traffic in
// the products.  A complex 16 x 16 x 16 product is 4 real v_mfma_f64_16x16x4_f64 chains; layouts:
//   A operand, slab kk : lane (g, c) holds M[row = c][k = 4 kk + g]      ("A-layout")
//   B operand, slab kk : lane (g, c) holds M[k = 4 kk + g][col = c]      ("B-layout")
//   C / D, register q  : lane (g, c) holds M[row = 4 q + g][col = c]     (== B-layout with kk = q)
// One power step r' = sum_s A_s r A_s^+ is done as  Y_s = r A_s^+ ,  r' += A_s Y_s :
//   * r lives in C-layout; because r is Hermitian its A-layout is conj(C-layout) - same registers;
//   * A_s lives ONCE, in A-layout; A_s^+ in B-layout is conj(A-layout of A_s) - same registers;
//   * Y_s comes out in C-layout == the B operand of the second product.
// So the whole iteration runs register-to-register: 64 MFMAs per step, plus one LDS transpose per
// step to re-hermitise r'.  status/iters semantics as in the other kernels.
// ------------------------------------------------------------------------------------------
struct C4 {   // a complex matrix in C-layout: 4 registers re, 4 registers im
  v4f64 re, im;
};

// C += P * Q with P given in A-layout (pa_re/pa_im[kk]) and Q in B-layout (C-layout registers)
__device__ __forceinline__ void cmma(const double (&pre)[4], const double (&pim)[4], const double (&pimn)[4],
                                     const v4f64& qre, const v4f64& qim, v4f64& cre, v4f64& cim) {
  // Three real products per k-slab instead of four (round 3, as in qmps_overlap.hip: K1 = (Pr + Pi) Qr, K2 = Pr (Qi - Qr),
  // K3 = Pi (Qr + Qi); Re = K1 - K3, Im = K1 + K2): 12 v_mfma_f64_16x16x4 per complex product instead of 16, in three
  // independent accumulator chains - the matrix pipe (~100 cycles per instruction on this part) bounds these kernels.
  (void)pimn;
  v4f64 k1 = {0, 0, 0, 0}, k2 = {0, 0, 0, 0}, k3 = {0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double ps = pre[kk] + pim[kk], qd = qim[kk] - qre[kk], qs = qre[kk] + qim[kk];
    k1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ps, qre[kk], k1, 0, 0, 0);
    k2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pre[kk], qd, k2, 0, 0, 0);
    k3 = __builtin_amdgcn_mfma_f64_16x16x4f64(pim[kk], qs, k3, 0, 0, 0);
  }
  cre += k1 - k3;
  cim += k1 + k2;
}


template <bool SOLVE>
__global__ __launch_bounds__(256) void energy_mfma_d16_kernel(LaneArgs p) {
  constexpr int D = 16, LD = 17, WAVES = 4;
  __shared__ double2 sT_all[WAVES][D * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double2* sT = sT_all[wave];
  const double tol2 = p.tol * p.tol;
  if (p.acc_zero != nullptr && blockIdx.x == 0) acc_clear(p.acc_zero, p.n_terms, threadIdx.x, 256);   // accumulator of a later step
  if (p.only_pending && __hip_atomic_load(p.kry_counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // finishing pass with nothing to finish
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wave; b < p.B; b += (int64_t)gridDim.x * WAVES) {
    if (p.only_pending && p.status[b] != QMPS_ST_PENDING) continue;      // finishing pass: only what the Krylov fall-back solved
    const int iters0 = p.only_pending ? p.iters[b] : 0;
    // A_s in A-layout (and the negated imaginary part, MFMA has no operand negation for f64)
    double are[2][4], aim[2][4], aimn[2][4];
    const double2* Ab = (const double2*)p.A + b * (2 * D * D);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 v = Ab[(s * D + c) * D + 4 * kk + g];
        are[s][kk] = v.x;
        aim[s][kk] = v.y;
        aimn[s][kk] = -v.y;
      }
    // r in C-layout
    C4 r;
    if (p.r_in != nullptr) {
      const double2* gi = (const double2*)p.r_in + b * (D * D);
      double tr = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 u = gi[(4 * q + g) * D + c], l = gi[c * D + 4 * q + g];
        r.re[q] = 0.5 * (u.x + l.x);
        r.im[q] = 0.5 * (u.y - l.y);
        tr += (c == 4 * q + g) ? r.re[q] : 0.0;
      }
      const double trs = wave_sum(tr);
      const bool usable = trs > 1e-300 && trs < 1e300;      // (zeros / NaN where nobody stored an environment: no guess, the default start)
      const double inv = usable ? 1.0 / trs : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r.re[q] = usable ? r.re[q] * inv : ((c == 4 * q + g) ? 1.0 / D : 0.0);
        r.im[q] = usable ? r.im[q] * inv : 0.0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r.re[q] = (c == 4 * q + g) ? 1.0 / D : 0.0;
        r.im[q] = 0.0;
      }
    }
    int iters = 0, status = SOLVE ? QMPS_ST_NOT_CONVERGED : QMPS_ST_OK, k_ref = 0;
    float l_ref = 0.0f;
    for (int k = 1; SOLVE && k <= p.max_iter; ++k) {
      C4 n;
      n.re = (v4f64){0, 0, 0, 0};
      n.im = (v4f64){0, 0, 0, 0};
      // A-layout of r = conj(C-layout): re as is, im negated; its negated imaginary part = + r.im
      double rre[4], rimn[4], rim[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rre[q] = r.re[q];
        rim[q] = -r.im[q];
        rimn[q] = r.im[q];
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // Y = r A_s^+ : B operand = conj(A-layout of A_s) = (are, -aim)
        C4 y;
        y.re = (v4f64){0, 0, 0, 0};
        y.im = (v4f64){0, 0, 0, 0};
        v4f64 bre = {are[s][0], are[s][1], are[s][2], are[s][3]};
        v4f64 bim = {aimn[s][0], aimn[s][1], aimn[s][2], aimn[s][3]};
        cmma(rre, rim, rimn, bre, bim, y.re, y.im);
        // r' += A_s Y
        cmma(are[s], aim[s], aimn[s], y.re, y.im, n.re, n.im);
      }
      // hermitise through a padded LDS transpose, trace-normalise, compare
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(n.re[q], n.im[q]);
      __builtin_amdgcn_wave_barrier();
      double tr = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 t = sT[c * LD + 4 * q + g];      // n[c][4 q + g]
        const bool diag = (c == 4 * q + g);
        n.re[q] = 0.5 * (n.re[q] + t.x);
        n.im[q] = diag ? 0.0 : 0.5 * (n.im[q] - t.y);
        tr += diag ? n.re[q] : 0.0;
      }
      const double inv = 1.0 / wave_sum(tr);
      double dpart = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        n.re[q] *= inv;
        n.im[q] *= inv;
        const double dr = n.re[q] - r.re[q], di = n.im[q] - r.im[q];
        dpart = dfma(dr, dr, dpart);
        dpart = dfma(di, di, dpart);
      }
      const double d2 = lane0(wave_sum(dpart));
      r = n;
      iters = k;
      if (d2 < tol2) {
        status = QMPS_ST_OK;
        break;
      }
      // a long tail ahead (|lambda_2| close to 1): hand the evaluation to the Krylov fall-back (status PENDING, r in r_out)
      if (k < p.max_iter && p.kry_counter != nullptr && power_gives_up(k, d2, tol2, p.krylov_after, k_ref, l_ref)) {
        status = QMPS_ST_PENDING;
        break;
      }
    }
    const bool given = SOLVE && status == QMPS_ST_PENDING;
    if (SOLVE && p.r_out != nullptr) {
      double2* ro = (double2*)p.r_out + b * (D * D);
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[(4 * q + g) * D + c] = make_double2(r.re[q], r.im[q]);
    }
    // ---- positive-definiteness (the reference's cholesky(r), qmps/tools.py:182): column Cholesky on the
    // LDS copy, one lane per row (lanes 0..15), pivot test by lane 0 semantics (wave-uniform result)
    if (!SOLVE && p.check_pd) status = p.status[b];
    if ((SOLVE || p.check_pd) && status == QMPS_ST_OK) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(r.re[q], r.im[q]);
      __builtin_amdgcn_wave_barrier();
      bool ok = true;
      // right-looking: after column j is scaled, rows i > j update their trailing entries; lane = row i
      for (int j = 0; j < D; ++j) {
        const double d = sT[j * LD + j].x;              // current pivot (already updated)
        if (!(d > 0.0)) { ok = false; break; }
        const double inv = 1.0 / __builtin_sqrt(d);
        __builtin_amdgcn_wave_barrier();
        double2 lij = make_double2(0.0, 0.0);
        if (lane < D && lane > j) {
          const double2 v = sT[lane * LD + j];
          lij = make_double2(v.x * inv, v.y * inv);
          sT[lane * LD + j] = lij;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < D && lane > j) {
          // row `lane`: a[lane][k] -= l[lane][j] conj(l[k][j]) for j < k <= lane
          for (int k = j + 1; k <= lane; ++k) {
            const double2 lkj = sT[k * LD + j];
            double2 a = sT[lane * LD + k];
            a.x -= lij.x * lkj.x + lij.y * lkj.y;
            a.y -= lij.y * lkj.x - lij.x * lkj.y;
            sT[lane * LD + k] = a;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (!ok) status = QMPS_ST_NOT_PD;
    }
    // ---- energy on the matrix cores: rho[(t1 t2)][(s1 s2)] = tr(A_t1 (A_t2 r A_s2^+) A_s1^+)/tr r
    //   Y_s2 = r A_s2^+ ;  R = A_t2 Y_s2 ;  Z = A_t1 R  (all C-layout) ;  rho = <A_s1, Z>_F  (wave reduction)
    double cre[2][4], cim[2][4];      // A_s in C-layout for the final inner products
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 v = Ab[(s * D + 4 * q + g) * D + c];
        cre[s][q] = v.x;
        cim[s][q] = v.y;
      }
    double trp = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) trp += (c == 4 * q + g) ? r.re[q] : 0.0;
    const double inv_tr = 1.0 / wave_sum(trp);
    double rre[4], rimn[4], rim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rre[q] = r.re[q];
      rim[q] = -r.im[q];
      rimn[q] = r.im[q];
    }
    __builtin_amdgcn_wave_barrier();      // sT is free again: rho[t][s] (wave-uniform) is parked in its first 16 slots
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      C4 y;
      y.re = (v4f64){0, 0, 0, 0};
      y.im = (v4f64){0, 0, 0, 0};
      v4f64 bre = {are[s2][0], are[s2][1], are[s2][2], are[s2][3]};
      v4f64 bim = {aimn[s2][0], aimn[s2][1], aimn[s2][2], aimn[s2][3]};
      cmma(rre, rim, rimn, bre, bim, y.re, y.im);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        C4 R;
        R.re = (v4f64){0, 0, 0, 0};
        R.im = (v4f64){0, 0, 0, 0};
        cmma(are[t2], aim[t2], aimn[t2], y.re, y.im, R.re, R.im);
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
          C4 Z;
          Z.re = (v4f64){0, 0, 0, 0};
          Z.im = (v4f64){0, 0, 0, 0};
          cmma(are[t1], aim[t1], aimn[t1], R.re, R.im, Z.re, Z.im);
#pragma unroll
          for (int s1 = 0; s1 < 2; ++s1) {
            double pr = 0.0, pi = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // Z conj(A_s1)
              pr = dfma(Z.re[q], cre[s1][q], pr);
              pr = dfma(Z.im[q], cim[s1][q], pr);
              pi = dfma(Z.im[q], cre[s1][q], pi);
              pi = dfma(-Z.re[q], cim[s1][q], pi);
            }
            const double sr = wave_sum(pr) * inv_tr, si = wave_sum(pi) * inv_tr;
            if (lane == 0) sT[(2 * t1 + t2) * 4 + 2 * s1 + s2] = make_double2(sr, si);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      for (int q = 0; q < p.n_terms; ++q) {
        const double2* h = (const double2*)p.h + q * 16;
        double e = 0.0;
        for (int s = 0; s < 4; ++s)
          for (int t = 0; t < 4; ++t) {
            const double2 hv = h[s * 4 + t], rv = sT[t * 4 + s];
            e += hv.x * rv.x - hv.y * rv.y;
          }
        if (given) continue;       // (its energy - and its arrival at the accumulator - come from the finishing pass)
        p.E[b * p.n_terms + q] = e;
        if (p.acc != nullptr) acc_arrive(p.acc, p.acc_shards, q, (unsigned)b, e, p.acc_bound, p.acc_scale);   // exact in-kernel cost
      }
      if (SOLVE) p.iters[b] = iters0 + iters;
      if (SOLVE || p.check_pd) p.status[b] = status;
      if (given) atomicAdd(p.kry_counter + 2, 1);
    }
    if (p.rho_out != nullptr && lane < 16 && !given) ((double2*)p.rho_out)[b * 16 + lane] = sT[lane];
    __builtin_amdgcn_wave_barrier();
  }
  if (p.only_pending) {        // the last workgroup of a finishing pass clears its counters (see OverlapArgs::kry_counter)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(p.kry_counter + 4, 1) == (int)gridDim.x - 1) {
      __hip_atomic_store(p.kry_counter + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.kry_counter + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 1d': the same with TWO WAVES PER EVALUATION (small batches: BASELINE.json configs[4] shards its trajectories, a GPU
// holds tens to hundreds of evaluations).  Wave w owns the physical index s = w: its half of the step is Y = r A_w^+,
// n_w = A_w Y (32 of the 64 MFMAs), the halves are summed through LDS in the same order by both waves (bit-identical
// iterates, identical decisions); the density matrix is split by s2 = w the same way.  One wave per evaluation leaves a
// quarter of the SIMDs idle at B = 768 and runs every dependent MFMA chain on one matrix pipe.
// ------------------------------------------------------------------------------------------
template <bool SOLVE>
__global__ __launch_bounds__(128) void energy_mfma_d16x2_kernel(LaneArgs p) {
  constexpr int D = 16, LD = 17, WAVES = 2;
  __shared__ double2 sT_all[WAVES][D * LD];          // wave-private transposes
  __shared__ double2 sX_all[WAVES][D * D];           // exchange: one C-layout matrix per wave, element (q, lane) at [q * 64 + lane]
  __shared__ double2 sRho[16];                        // rho[t][s], entries with s2 = w written by wave w
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double2* sT = sT_all[wave];
  const double tol2 = p.tol * p.tol;
  if (p.acc_zero != nullptr && blockIdx.x == 0) acc_clear(p.acc_zero, p.n_terms, threadIdx.x, 128);   // accumulator of a later step
  if (p.only_pending && __hip_atomic_load(p.kry_counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // finishing pass with nothing to finish
  for (int64_t b = blockIdx.x; b < p.B; b += gridDim.x) {
    if (p.only_pending && p.status[b] != QMPS_ST_PENDING) continue;      // (uniform over the workgroup; no LDS touched)
    const int iters0 = p.only_pending ? p.iters[b] : 0;
    // A_s in A-layout (and the negated imaginary part, MFMA has no operand negation for f64)
    double are[2][4], aim[2][4], aimn[2][4];
    const double2* Ab = (const double2*)p.A + b * (2 * D * D);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 v = Ab[(s * D + c) * D + 4 * kk + g];
        are[s][kk] = v.x;
        aim[s][kk] = v.y;
        aimn[s][kk] = -v.y;
      }
    // r in C-layout
    C4 r;
    if (p.r_in != nullptr) {
      const double2* gi = (const double2*)p.r_in + b * (D * D);
      double tr = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 u = gi[(4 * q + g) * D + c], l = gi[c * D + 4 * q + g];
        r.re[q] = 0.5 * (u.x + l.x);
        r.im[q] = 0.5 * (u.y - l.y);
        tr += (c == 4 * q + g) ? r.re[q] : 0.0;
      }
      const double trs = wave_sum(tr);
      const bool usable = trs > 1e-300 && trs < 1e300;      // (zeros / NaN where nobody stored an environment: no guess, the default start)
      const double inv = usable ? 1.0 / trs : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r.re[q] = usable ? r.re[q] * inv : ((c == 4 * q + g) ? 1.0 / D : 0.0);
        r.im[q] = usable ? r.im[q] * inv : 0.0;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r.re[q] = (c == 4 * q + g) ? 1.0 / D : 0.0;
        r.im[q] = 0.0;
      }
    }
    int iters = 0, status = SOLVE ? QMPS_ST_NOT_CONVERGED : QMPS_ST_OK, k_ref = 0;
    float l_ref = 0.0f;
    for (int k = 1; SOLVE && k <= p.max_iter; ++k) {
      C4 n;
      n.re = (v4f64){0, 0, 0, 0};
      n.im = (v4f64){0, 0, 0, 0};
      // A-layout of r = conj(C-layout): re as is, im negated; its negated imaginary part = + r.im
      double rre[4], rimn[4], rim[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rre[q] = r.re[q];
        rim[q] = -r.im[q];
        rimn[q] = r.im[q];
      }
      {
        // this wave's physical index s = wave:  Y = r A_s^+ (B operand = conj(A-layout of A_s) = (are, -aim)),  n_s = A_s Y
        const int s = wave;
        C4 y, part;
        y.re = (v4f64){0, 0, 0, 0};
        y.im = (v4f64){0, 0, 0, 0};
        part.re = (v4f64){0, 0, 0, 0};
        part.im = (v4f64){0, 0, 0, 0};
        v4f64 bre = {are[s][0], are[s][1], are[s][2], are[s][3]};
        v4f64 bim = {aimn[s][0], aimn[s][1], aimn[s][2], aimn[s][3]};
        cmma(rre, rim, rimn, bre, bim, y.re, y.im);
        cmma(are[s], aim[s], aimn[s], y.re, y.im, part.re, part.im);
        // r' = n_0 + n_1 through LDS, summed in the same order by both waves (bit-identical iterates, identical decisions)
#pragma unroll
        for (int q = 0; q < 4; ++q) sX_all[wave][q * 64 + lane] = make_double2(part.re[q], part.im[q]);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double2 v = sX_all[w][q * 64 + lane];
            n.re[q] += v.x;
            n.im[q] += v.y;
          }
        __syncthreads();
      }
      // hermitise through a padded LDS transpose, trace-normalise, compare
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(n.re[q], n.im[q]);
      __builtin_amdgcn_wave_barrier();
      double tr = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 t = sT[c * LD + 4 * q + g];      // n[c][4 q + g]
        const bool diag = (c == 4 * q + g);
        n.re[q] = 0.5 * (n.re[q] + t.x);
        n.im[q] = diag ? 0.0 : 0.5 * (n.im[q] - t.y);
        tr += diag ? n.re[q] : 0.0;
      }
      const double inv = 1.0 / wave_sum(tr);
      double dpart = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        n.re[q] *= inv;
        n.im[q] *= inv;
        const double dr = n.re[q] - r.re[q], di = n.im[q] - r.im[q];
        dpart = dfma(dr, dr, dpart);
        dpart = dfma(di, di, dpart);
      }
      const double d2 = lane0(wave_sum(dpart));
      r = n;
      iters = k;
      if (d2 < tol2) {
        status = QMPS_ST_OK;
        break;
      }
      // a long tail ahead (|lambda_2| close to 1): hand the evaluation to the Krylov fall-back (status PENDING, r in r_out)
      if (k < p.max_iter && p.kry_counter != nullptr && power_gives_up(k, d2, tol2, p.krylov_after, k_ref, l_ref)) {
        status = QMPS_ST_PENDING;
        break;
      }
    }
    const bool given = SOLVE && status == QMPS_ST_PENDING;
    if (SOLVE && p.r_out != nullptr && wave == 0) {
      double2* ro = (double2*)p.r_out + b * (D * D);
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[(4 * q + g) * D + c] = make_double2(r.re[q], r.im[q]);
    }
    // ---- positive-definiteness (the reference's cholesky(r), qmps/tools.py:182): column Cholesky on the
    // LDS copy, one lane per row (lanes 0..15), pivot test by lane 0 semantics (wave-uniform result)
    if (!SOLVE && p.check_pd) status = p.status[b];
    if ((SOLVE || p.check_pd) && status == QMPS_ST_OK && wave == 0) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(r.re[q], r.im[q]);
      __builtin_amdgcn_wave_barrier();
      bool ok = true;
      // right-looking: after column j is scaled, rows i > j update their trailing entries; lane = row i
      for (int j = 0; j < D; ++j) {
        const double d = sT[j * LD + j].x;              // current pivot (already updated)
        if (!(d > 0.0)) { ok = false; break; }
        const double inv = 1.0 / __builtin_sqrt(d);
        __builtin_amdgcn_wave_barrier();
        double2 lij = make_double2(0.0, 0.0);
        if (lane < D && lane > j) {
          const double2 v = sT[lane * LD + j];
          lij = make_double2(v.x * inv, v.y * inv);
          sT[lane * LD + j] = lij;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < D && lane > j) {
          // row `lane`: a[lane][k] -= l[lane][j] conj(l[k][j]) for j < k <= lane
          for (int k = j + 1; k <= lane; ++k) {
            const double2 lkj = sT[k * LD + j];
            double2 a = sT[lane * LD + k];
            a.x -= lij.x * lkj.x + lij.y * lkj.y;
            a.y -= lij.y * lkj.x - lij.x * lkj.y;
            sT[lane * LD + k] = a;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (!ok) status = QMPS_ST_NOT_PD;
    }
    // ---- energy on the matrix cores: rho[(t1 t2)][(s1 s2)] = tr(A_t1 (A_t2 r A_s2^+) A_s1^+)/tr r
    //   Y_s2 = r A_s2^+ ;  R = A_t2 Y_s2 ;  Z = A_t1 R  (all C-layout) ;  rho = <A_s1, Z>_F  (wave reduction)
    double cre[2][4], cim[2][4];      // A_s in C-layout for the final inner products
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 v = Ab[(s * D + 4 * q + g) * D + c];
        cre[s][q] = v.x;
        cim[s][q] = v.y;
      }
    double trp = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) trp += (c == 4 * q + g) ? r.re[q] : 0.0;
    const double inv_tr = 1.0 / wave_sum(trp);
    double rre[4], rimn[4], rim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rre[q] = r.re[q];
      rim[q] = -r.im[q];
      rimn[q] = r.im[q];
    }
    __builtin_amdgcn_wave_barrier();      // sT is free again: rho[t][s] (wave-uniform) is parked in its first 16 slots
    {
      const int s2 = wave;             // this wave's half of the density matrix
      C4 y;
      y.re = (v4f64){0, 0, 0, 0};
      y.im = (v4f64){0, 0, 0, 0};
      v4f64 bre = {are[s2][0], are[s2][1], are[s2][2], are[s2][3]};
      v4f64 bim = {aimn[s2][0], aimn[s2][1], aimn[s2][2], aimn[s2][3]};
      cmma(rre, rim, rimn, bre, bim, y.re, y.im);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        C4 R;
        R.re = (v4f64){0, 0, 0, 0};
        R.im = (v4f64){0, 0, 0, 0};
        cmma(are[t2], aim[t2], aimn[t2], y.re, y.im, R.re, R.im);
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1) {
          C4 Z;
          Z.re = (v4f64){0, 0, 0, 0};
          Z.im = (v4f64){0, 0, 0, 0};
          cmma(are[t1], aim[t1], aimn[t1], R.re, R.im, Z.re, Z.im);
#pragma unroll
          for (int s1 = 0; s1 < 2; ++s1) {
            double pr = 0.0, pi = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // Z conj(A_s1)
              pr = dfma(Z.re[q], cre[s1][q], pr);
              pr = dfma(Z.im[q], cim[s1][q], pr);
              pi = dfma(Z.im[q], cre[s1][q], pi);
              pi = dfma(-Z.re[q], cim[s1][q], pi);
            }
            const double sr = wave_sum(pr) * inv_tr, si = wave_sum(pi) * inv_tr;
            if (lane == 0) sRho[(2 * t1 + t2) * 4 + 2 * s1 + s2] = make_double2(sr, si);
          }
        }
      }
    }
    __syncthreads();
    if (wave == 0 && lane == 0) {
      for (int q = 0; q < p.n_terms; ++q) {
        const double2* h = (const double2*)p.h + q * 16;
        double e = 0.0;
        for (int s = 0; s < 4; ++s)
          for (int t = 0; t < 4; ++t) {
            const double2 hv = h[s * 4 + t], rv = sRho[t * 4 + s];
            e += hv.x * rv.x - hv.y * rv.y;
          }
        if (given) continue;       // (its energy - and its arrival at the accumulator - come from the finishing pass)
        p.E[b * p.n_terms + q] = e;
        if (p.acc != nullptr) acc_arrive(p.acc, p.acc_shards, q, (unsigned)b, e, p.acc_bound, p.acc_scale);   // exact in-kernel cost
      }
      if (SOLVE) p.iters[b] = iters0 + iters;
      if (SOLVE || p.check_pd) p.status[b] = status;
      if (given) atomicAdd(p.kry_counter + 2, 1);
    }
    if (p.rho_out != nullptr && wave == 0 && lane < 16 && !given) ((double2*)p.rho_out)[b * 16 + lane] = sRho[lane];
    __syncthreads();
  }
  if (p.only_pending) {        // the last workgroup of a finishing pass clears its counters
    if (threadIdx.x == 0 && atomicAdd(p.kry_counter + 4, 1) == (int)gridDim.x - 1) {
      __hip_atomic_store(p.kry_counter + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.kry_counter + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

hipError_t launch_energy_mfma(int D, const LaneArgs& a, bool solve, hipStream_t st) {
  if (D != 16) return hipErrorInvalidValue;
  if (a.B <= 0) return hipSuccess;
  // measured: B = 96: 0.210 ms against 0.287 with one wave per evaluation; B = 768: 0.325 against 0.318 (the exchange through LDS
  // and its two barriers per step cost what the shorter chain saves once every SIMD has a wave anyway)
  static const int64_t split_below = tuning_knob("QMPS_D16_SPLIT_BELOW") ? atoll(tuning_knob("QMPS_D16_SPLIT_BELOW")) : 512;   // A/B knob
  if (solve && a.only_pending) {
    // finishing pass behind the Krylov fall-back: a handful of evaluations at most; every workgroup looks at the pending count first
    hipLaunchKernelGGL(energy_mfma_d16x2_kernel<true>, dim3((unsigned)(a.B < 64 ? a.B : 64)), dim3(128), 0, st, a);
    return hipGetLastError();
  }
  if (solve && a.B <= split_below) {
    // few evaluations: two waves per evaluation (half the dependent MFMA chain per wave)
    hipLaunchKernelGGL(energy_mfma_d16x2_kernel<true>, dim3((unsigned)(a.B < 8192 ? a.B : 8192)), dim3(128), 0, st, a);
    return hipGetLastError();
  }
  int grid = (int)((a.B + 3) / 4);
  if (grid > 4096) grid = 4096;
  if (solve)
    hipLaunchKernelGGL(energy_mfma_d16_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(energy_mfma_d16_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace qmps
