// qmps_overlap.hip - time-evolution overlap objective at bond dimension D = 4, 8, 16 (gfx950 only).
//
// Reference: qmps/new_time_evolve.py:193-221, scripts/loschmidt.py:209-239 and qmps/time_evolve_tools.py:20-23 (`merge`,
// whose final reshape hard-codes D = 2 in the reference; the mathematics is bond-dimension agnostic):
//   T(x) = sum_{s=0..3} C_s x Bm_s^+ ,   C = WW . merge(A, A) ,   Bm = merge(B, B) ,   merge(A, A)[2 s1 + s2] = A_s1 A_s2
// eta = dominant eigenvalue of T (complex; T is not Hermitian).  The reference's circuit measures 2 |psi[0]| = |eta|
// and minimises -sqrt(|eta|), so the kernel is only the dominant-eigenvalue solve.
//
// Power method in operator form (never builds the D^2 x D^2 matrix): x <- T(x)/||T(x)||_F from x_0 = 1/sqrt(D),
// eta = <x, T x> (Rayleigh quotient, ||x||_F = 1), stop when ||T x - eta x||_F < tol.  status 1 = no unique dominant
// eigenvalue within max_steps.  The D = 2 path (overlap_lane_kernel, qmps_kernels.hip) squares the 4 x 4 matrix instead.
//
//   overlap_square_d4_kernel   D = 4: the map is ONE complex 16 x 16 tile - squared on the matrix cores until it is rank one
//                              (O(log) rounds whatever the spectral gap), one wave per evaluation.
//   overlap_block_kernel<D>    D = 8 (D = 4, 16 fall-backs): thread (i, j) of a D x D tile per evaluation (D = 4: four evaluations per
//                              wave, one per DPP row), tiles of C_s, Bm_s, x, Y_s = x Bm_s^+ in LDS.
//   overlap_mfma_d16_kernel    D = 16: ONE WAVE PER EVALUATION on the matrix cores.  A complex 16 x 16 x 16 product is
//                              4 real v_mfma_f64_16x16x4_f64 chains; per step Y_s = x Bm_s^+ and x' += C_s Y_s for the
//                              four s: 128 MFMAs, register to register but for ONE LDS transpose of x
//                              (layouts as in energy_mfma_d16_kernel: C-layout of a product == B-layout of the next;
//                              Bm_s^+ in B-layout == conj of Bm_s in A-layout).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_knobs.h"
#include "qmps_device.h"
#include "qmps_circuit_wave.h"
#include "qmps_overlap_d2.h"
#include "qmps_overlap_d4.h"

namespace qmps {


// ------------------------------------------------------------------------------------------
// generic tile kernel
// ------------------------------------------------------------------------------------------
// ADJ: power method on the adjoint map y -> sum_s C_s^+ y Bm_s (its dominant eigenvalue is conj(eta), its fixed point the LEFT
// eigenvector of T for the Frobenius pairing: <y, T x> = <T^+ y, x>); eta_out receives eta itself
// (the body as a device function: the workgroups of one launch may run different instantiations - overlap_block_pair_kernel)
template <int D, bool ADJ>
__device__ __forceinline__ void overlap_block_body(const OverlapArgs& p, int64_t block) {
  constexpr int N = D * D, P = D + 1;
  constexpr int THREADS = N < 64 ? 64 : N, ITEMS = THREADS / N, WAVES = THREADS / 64;
  __shared__ double2 sC[ITEMS][4][D][P], sB[ITEMS][4][D][P], sX[ITEMS][D][P], sY[ITEMS][4][D][P];
  __shared__ double red[4][WAVES > 1 ? WAVES : 1];
  const int tid = threadIdx.x, e = tid / N, l = tid % N, i = l / D, j = l % D;
  const int64_t b = block * ITEMS + e;
  if (ITEMS == 1 && b < p.B && overlap_skipped(p, b)) return;      // (one evaluation per workgroup: a uniform exit)
  const bool valid = b < p.B;
  const int64_t bb = valid ? b : p.B - 1;       // surplus lanes of the last workgroup shadow a real evaluation
  // sum over the N threads of one evaluation (up to four values at a time), result in every one of them
  auto group_sum4 = [&](double (&v)[4]) {
    if constexpr (N == 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = row16_sum(v[q]);
    } else if constexpr (N == 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
      __syncthreads();
      if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[q][tid >> 6] = v[q];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red[q][w];
        v[q] = t;
      }
    }
  };
  // ---- set-up: inputs through the Y tiles, then C_s = sum_t WW[s][t] A_t1 A_t2 and Bm_s = B_s1 B_s2
  {
    const double2* Ap = (const double2*)p.A + overlap_ref_index(p, bb) * (2 * N);
    const double2* Bp = (const double2*)p.Bt + bb * (2 * N);
    sY[e][0][i][j] = Ap[l];
    sY[e][1][i][j] = Ap[N + l];
    sY[e][2][i][j] = Bp[l];
    sY[e][3][i][j] = Bp[N + l];
  }
  __syncthreads();
  {
    double2 aa[4], bm[4];
#pragma unroll
    for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        double2 u = make_double2(0.0, 0.0), v = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < D; ++k) {
          cfma(sY[e][t1][i][k], sY[e][t2][k][j], u);
          cfma(sY[e][2 + t1][i][k], sY[e][2 + t2][k][j], v);
        }
        aa[2 * t1 + t2] = u;
        bm[2 * t1 + t2] = v;
      }
    const double2* W = (const double2*)p.WW;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double2 c = make_double2(0.0, 0.0);
#pragma unroll
      for (int t = 0; t < 4; ++t) cfma(W[s * 4 + t], aa[t], c);
      sC[e][s][i][j] = c;
      sB[e][s][i][j] = bm[s];
    }
  }
  double2 x_cold = overlap_cold_start(i, j, D);      // (generic: see overlap_cold_start)
  {
    double v[4] = {x_cold.x * x_cold.x + x_cold.y * x_cold.y, 0.0, 0.0, 0.0};
    group_sum4(v);
    const double inv = 1.0 / __builtin_sqrt(v[0]);
    x_cold = make_double2(x_cold.x * inv, x_cold.y * inv);
  }
  double2 x = x_cold;
  const int64_t slot_off = overlap_slot_offset(p);
  if (p.x_in != nullptr) {      // warm start: the resident fixed point of this candidate's slot (all zero: none yet)
    const double2 w = ((const double2*)((const char*)p.x_in + slot_off))[(p.x_in_group > 0 ? bb / p.x_in_group : bb) * N + l];
    double v[4] = {w.x * w.x + w.y * w.y, 0.0, 0.0, 0.0};
    group_sum4(v);
    if (v[0] > 1e-200 && v[0] < 1e200) {
      const double inv = 1.0 / __builtin_sqrt(v[0]);
      x = make_double2(w.x * inv, w.y * inv);
    }
  }
  double2 eta = make_double2(0.0, 0.0);
  int iters = 0, status = QMPS_ST_NOT_CONVERGED;
  bool active = true;
  const double tol2 = overlap_tol2(p, b < p.B ? b : p.B - 1);
  // deflation steps (below): one evaluation per wave only - the branch must be uniform over the lanes of a reduction
  constexpr bool kDeflate = D == 8;
  double2 rp = make_double2(0.0, 0.0), sg_prev = make_double2(0.0, 0.0);
  double w0_prev = 0.0, n_prev = 0.0, sig_max2 = 0.0;
  int last_deflation = 0;
  bool deflate_on = p.no_deflation == 0;
  // hand-over to the Krylov fall-back (one evaluation per workgroup only: the decision must be uniform)
  const int give_up_after = (ITEMS == 1 && p.r_out != nullptr && p.kry_counter != nullptr) ? p.krylov_after : 0;
  int k_ref = 0;
  float l_ref = 0.0f;
  __syncthreads();
  for (int k = 1; k <= p.max_rounds; ++k) {
    if (!__syncthreads_or(active ? 1 : 0)) break;
    sX[e][i][j] = x;
    __syncthreads();
    // Y_s = x Bm_s^+   (ADJ: x Bm_s)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double2 y = make_double2(0.0, 0.0);
#pragma unroll
      for (int kk = 0; kk < D; ++kk) {
        if constexpr (ADJ) cfma(sX[e][i][kk], sB[e][s][kk][j], y);
        else cfma_conj(sX[e][i][kk], sB[e][s][j][kk], y);
      }
      sY[e][s][i][j] = y;
    }
    __syncthreads();
    // x' = sum_s C_s Y_s   (ADJ: C_s^+ Y_s)
    double2 xn = make_double2(0.0, 0.0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kk = 0; kk < D; ++kk) {
        if constexpr (ADJ) cfma_conj(sY[e][s][kk][j], sC[e][s][kk][i], xn);
        else cfma(sC[e][s][i][kk], sY[e][s][kk][j], xn);
      }
    // eta = <x, x'>, ||x'||^2, then the residual ||x' - eta x||^2   (||x||_F = 1)
    double v[4] = {x.x * xn.x + x.y * xn.y, x.x * xn.y - x.y * xn.x, xn.x * xn.x + xn.y * xn.y, 0.0};
    group_sum4(v);
    const double2 et = make_double2(v[0], v[1]);
    const double nn = v[2];
    const double dr = xn.x - (et.x * x.x - et.y * x.y), di = xn.y - (et.x * x.y + et.y * x.x);
    // (the spare slots of this reduction: <r_{k-1}, r_k> for the estimate of the second eigenvalue, see below)
    double w[4] = {dr * dr + di * di, rp.x * dr + rp.y * di, rp.x * di - rp.y * dr, 0.0};
    group_sum4(w);
    if (active) {
      eta = et;
      iters = k;
      const double e2 = et.x * et.x + et.y * et.y;
      if (w[0] < tol2 && kDeflate && sig_max2 > 0.0 && !(e2 > sig_max2 * (1.0 + 1e-3))) {
        // converged - to an eigenvalue that does not dominate one of those a deflation step removed (a nearly degenerate
        // dominant pair: the shift took out the wrong one).  Start again, plain power method only.
        deflate_on = false;
        sig_max2 = 0.0;
        w0_prev = 0.0;
        k_ref = 0;
        rp = make_double2(0.0, 0.0);
        x = x_cold;
      } else if (w[0] < tol2) {
        status = QMPS_ST_OK;
        active = false;
      } else {
        bool deflated = false;
        if constexpr (kDeflate) {
          // Slow convergence = a second eigenvalue eta_2 close to eta_1 in modulus.  The residual r_k = (T - eta_k) x_k is then
          // dominated by that eigenvector, r_k ~ (eta_2 / n_{k-1}) r_{k-1} with n = ||T x||, so two successive residuals give
          // eta_2 for free; ONE step with the shifted map, x <- (T - sigma) x = T x - sigma x, removes that component (Wielandt).
          // It also amplifies every other component by |eta_j - sigma| / |eta_1 - sigma|, so it is taken only when the plain
          // iteration is slow, the estimate has settled, and not again before the plain steps have damped what it stirred up.
          // The result is still the power method's: convergence is declared by the same residual test on plain steps only.
          const double2 sg = w0_prev > 0.0 ? make_double2(n_prev * w[1] / w0_prev, n_prev * w[2] / w0_prev) : make_double2(0.0, 0.0);
          const double ds = (sg.x - sg_prev.x) * (sg.x - sg_prev.x) + (sg.y - sg_prev.y) * (sg.y - sg_prev.y), s2 = sg.x * sg.x + sg.y * sg.y;
          // (the estimate settled to 1e-4, and what it names is smaller in modulus than the current Rayleigh quotient - never
          // shift out something that may be the dominant eigenvalue; the check above is the safety net behind this one)
          if (deflate_on && k >= 24 && k - last_deflation >= 12 && w[0] > 0.64 * w0_prev && ds < 1e-8 * s2 && s2 < 0.998 * e2 && s2 > 0.25 * e2) {
            double2 xd = make_double2(xn.x - (sg.x * x.x - sg.y * x.y), xn.y - (sg.x * x.y + sg.y * x.x));
            double q[4] = {xd.x * xd.x + xd.y * xd.y, 0.0, 0.0, 0.0};
            group_sum4(q);
            if (q[0] > 1e-280) {
              const double inv = 1.0 / __builtin_sqrt(q[0]);
              x = make_double2(xd.x * inv, xd.y * inv);
              deflated = true;
              last_deflation = k;
              sig_max2 = s2 > sig_max2 ? s2 : sig_max2;
            }
          }
          sg_prev = sg;
          w0_prev = deflated ? 0.0 : w[0];
          n_prev = __builtin_sqrt(nn);
          rp = make_double2(dr, di);
        }
        if (!deflated) {
          const double inv = nn > 0.0 ? 1.0 / __builtin_sqrt(nn) : 0.0;
          x = make_double2(xn.x * inv, xn.y * inv);
        } else {
          k_ref = 0;
        }
        // a long tail ahead: stop here (status 1, k < max_rounds) - overlap_krylov_kernel takes the candidate over from x
        if (power_gives_up(k, w[0], tol2, give_up_after, k_ref, l_ref) && k < p.max_rounds) {
          active = false;
          if (tid == 0) atomicAdd(p.kry_counter + 2, 1);
        }
      }
    }
  }
  if (!valid) return;
  if (l == 0) overlap_store(p, b, eta.x, ADJ ? -eta.y : eta.y, iters, status);
  if (p.r_out != nullptr) ((double2*)((char*)p.r_out + slot_off))[b * N + l] = x;
}

template <int D, bool ADJ = false>
__global__ __launch_bounds__((D * D < 64) ? 64 : D * D) void overlap_block_kernel(OverlapArgs p) {
  overlap_block_body<D, ADJ>(p, blockIdx.x);
}

// RIGHT and LEFT fixed points in one launch (qmps_overlap_gradient at D = 8): workgroups [0, n_right) run the map of `pr`, the
// others the adjoint map of `pl` - the iteration chains are latency-bound, the left solve rides along on idle SIMDs
template <int D>
__global__ __launch_bounds__((D * D < 64) ? 64 : D * D) void overlap_block_pair_kernel(OverlapArgs pr, OverlapArgs pl, int n_right) {
  if ((int)blockIdx.x < n_right) overlap_block_body<D, false>(pr, blockIdx.x);
  else overlap_block_body<D, true>(pl, (int64_t)blockIdx.x - n_right);
}

// ------------------------------------------------------------------------------------------
// D = 16 on the matrix cores, one wave per evaluation
// ------------------------------------------------------------------------------------------
namespace {

// The power iterations below test convergence on every SECOND step only (and on the last one): the three wave-wide sums and the
// residual pass of a test are a third of a step's latency once the products take 24 instead of 32 matrix instructions; between
// tests the iterate is not normalised (it shrinks by |eta| ~ 0.5 .. 1 per step).  `rounds` counts every step.
__device__ __forceinline__ bool overlap_check_step(int k, int max_rounds) { return (k & 1) == 0 || k == max_rounds; }

}  // namespace

template <bool ADJ>
__global__ __launch_bounds__(256) void overlap_mfma_d16_kernel(OverlapArgs p) {
  constexpr int D = 16, LD = 17, WAVES = 4;
  __shared__ double2 sT_all[WAVES][D * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double2* sT = sT_all[wave];
  const double tol2 = p.tol * p.tol;
  // C-layout register q of a matrix: element [row = 4 q + g][col = c];  A-layout slab kk: element [row = c][k = 4 kk + g]
  auto to_a_layout = [&](const v4f64& re, const v4f64& im, double (&are)[4], double (&aim)[4]) {   // wave-private LDS transpose
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(re[q], im[q]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 t = sT[c * LD + 4 * kk + g];
      are[kk] = t.x;
      aim[kk] = t.y;
    }
  };
  const int64_t slot_off = overlap_slot_offset(p);
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wave; b < p.B; b += (int64_t)gridDim.x * WAVES) {
    if (overlap_skipped(p, b)) continue;
    const double2* Ap = (const double2*)p.A + overlap_ref_index(p, b) * (2 * D * D);
    const double2* Bp = (const double2*)p.Bt + b * (2 * D * D);
    const double2* W = (const double2*)p.WW;
    // ---- set-up: AA_t = A_t1 A_t2 and BB_t = B_t1 B_t2 (C-layout), C_s = sum_t WW[s][t] AA_t; both sets to A-layout
    double cre[4][4], cim[4][4];     // C_s in A-layout (the P operand of x' += C_s Y_s)
    double bre[4][4], bimn[4][4];    // conj(Bm_s) in A-layout == Bm_s^+ in B-layout (the Q operand of Y_s = x Bm_s^+)
    // ADJ (y -> sum_s C_s^+ y Bm_s): the Q operand of Y_s = y Bm_s is Bm_s in B-layout - the C-layout the product B_s1 B_s2 comes
    // out in - and the P operand of y' += C_s^+ Y_s is C_s^+ in A-layout = conj of C_s in C-layout: no LDS transpose in the set-up
    {
      double pa[2][4], pai[2][4], pb[2][4], pbi[2][4];   // A_s, B_s in A-layout
      v4f64 qa[2], qai[2], qb[2], qbi[2];                 // A_s, B_s in B-layout (= C-layout)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double2 va = Ap[(s * D + c) * D + 4 * kk + g], vb = Bp[(s * D + c) * D + 4 * kk + g];
          pa[s][kk] = va.x; pai[s][kk] = va.y;
          pb[s][kk] = vb.x; pbi[s][kk] = vb.y;
          const double2 wa = Ap[(s * D + 4 * kk + g) * D + c], wb = Bp[(s * D + 4 * kk + g) * D + c];
          qa[s][kk] = wa.x; qai[s][kk] = wa.y;
          qb[s][kk] = wb.x; qbi[s][kk] = wb.y;
        }
      v4f64 aar[4], aai[4];
#pragma unroll
      for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          v4f64 zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
          cmma16(pa[t1], pai[t1], qa[t2], qai[t2], zr, zi);
          aar[2 * t1 + t2] = zr;
          aai[2 * t1 + t2] = zi;
          v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
          cmma16(pb[t1], pbi[t1], qb[t2], qbi[t2], yr, yi);
          if constexpr (ADJ) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              bre[2 * t1 + t2][kk] = yr[kk];
              bimn[2 * t1 + t2][kk] = yi[kk];
            }
          } else {
            double tr[4], ti[4];
            to_a_layout(yr, yi, tr, ti);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              bre[2 * t1 + t2][kk] = tr[kk];
              bimn[2 * t1 + t2][kk] = -ti[kk];
            }
          }
        }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        v4f64 zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double2 w = W[s * 4 + t];
          zr += w.x * aar[t] - w.y * aai[t];
          zi += w.x * aai[t] + w.y * aar[t];
        }
        if constexpr (ADJ) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            cre[s][kk] = zr[kk];
            cim[s][kk] = -zi[kk];
          }
        } else {
          to_a_layout(zr, zi, cre[s], cim[s]);
        }
      }
    }
    // ---- power method, x in C-layout, ||x||_F = 1
    v4f64 xr, xi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 x0 = overlap_cold_start(4 * q + g, c, 16);      // (generic: see overlap_cold_start)
      xr[q] = x0.x;
      xi[q] = x0.y;
    }
    if (p.x_in != nullptr) {      // warm start: the resident fixed point of this candidate's slot (all zero: none yet)
      const double2* xi_ = (const double2*)((const char*)p.x_in + slot_off) + (p.x_in_group > 0 ? b / p.x_in_group : b) * (D * D);
      v4f64 wr, wi;
      double n2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 t = xi_[(4 * q + g) * D + c];
        wr[q] = t.x; wi[q] = t.y;
        n2 = dfma(t.x, t.x, dfma(t.y, t.y, n2));
      }
      n2 = lane0(wave_sum(n2));
      if (n2 > 1e-200 && n2 < 1e200) {
        const double inv = 1.0 / __builtin_sqrt(n2);
        xr = wr * inv;
        xi = wi * inv;
      }
    }
    double eta_r = 0.0, eta_i = 0.0;
    int iters = 0, status = QMPS_ST_NOT_CONVERGED;
    for (int k = 1; k <= p.max_rounds; ++k) {
      double xar[4], xai[4];
      to_a_layout(xr, xi, xar, xai);
      v4f64 nr = {0, 0, 0, 0}, ni = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
        const v4f64 qre = {bre[s][0], bre[s][1], bre[s][2], bre[s][3]};
        const v4f64 qim = {bimn[s][0], bimn[s][1], bimn[s][2], bimn[s][3]};
        cmma16_3m(xar, xai, qre, qim, yr, yi);           // Y_s = x Bm_s^+
        cmma16_3m(cre[s], cim[s], yr, yi, nr, ni);       // x' += C_s Y_s
      }
      iters = k;
      if (!overlap_check_step(k, p.max_rounds)) {        // no test on this step: carry the (unnormalised) iterate on
        xr = nr;
        xi = ni;
        continue;
      }
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a0 = dfma(xr[q], nr[q], a0);
        a0 = dfma(xi[q], ni[q], a0);
        a1 = dfma(xr[q], ni[q], a1);
        a1 = dfma(-xi[q], nr[q], a1);
        a2 = dfma(nr[q], nr[q], a2);
        a2 = dfma(ni[q], ni[q], a2);
        a3 = dfma(xr[q], xr[q], a3);
        a3 = dfma(xi[q], xi[q], a3);
      }
      const double xx = wave_sum(a3), ixx = xx > 0.0 ? 1.0 / xx : 0.0;
      eta_r = wave_sum(a0) * ixx;                        // eta = <x, T x> / <x, x>
      eta_i = wave_sum(a1) * ixx;
      const double nn = wave_sum(a2);
      double rs = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double dr = nr[q] - (eta_r * xr[q] - eta_i * xi[q]), di = ni[q] - (eta_r * xi[q] + eta_i * xr[q]);
        rs = dfma(dr, dr, rs);
        rs = dfma(di, di, rs);
      }
      const double res2 = lane0(wave_sum(rs)) * ixx;     // ||T x - eta x||^2 / ||x||^2
      if (res2 < tol2) {
        status = QMPS_ST_OK;
        const double inv = xx > 0.0 ? 1.0 / __builtin_sqrt(xx) : 0.0;      // the fixed point handed out has unit norm
        xr *= inv;
        xi *= inv;
        break;
      }
      const double inv = nn > 0.0 ? 1.0 / __builtin_sqrt(nn) : 0.0;
      xr = nr * inv;
      xi = ni * inv;
    }
    if (lane == 0) overlap_store(p, b, eta_r, ADJ ? -eta_i : eta_i, iters, status);
    if (p.r_out != nullptr) {
      double2* ro = (double2*)((char*)p.r_out + slot_off) + b * (D * D);
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[(4 * q + g) * D + c] = make_double2(xr[q], xi[q]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// D = 4 on the matrix cores: the mixed transfer map IS one complex 16 x 16 tile,
//   E[(i,i'),(j,j')] = sum_{s<4} C_s[i][j] conj(Bm_s[i'][j']),
// so the power method is taken 2^m steps at a time by SQUARING it (one wave per evaluation, 16 v_mfma_f64_16x16x4 per
// round, Frobenius-normalised): O(log) rounds whatever the spectral gap - the plain power method needed up to 27 386
// steps inside a batch of 65 536 time-step candidates.  M = E^(2^m)/scale tends to the rank-one u v^+; a rank-one
// matrix satisfies M M = tr(M) M, which is the convergence test (||M M - tr(M) M||_F < tol ||M M||_F, elementwise in
// the accumulator layout, no gather), and then eta = tr(M E)/tr(M) (= v^+ E u / v^+ u).  The right fixed point, when
// asked for, is the largest column of M.  rounds = squarings used; status 1 = not rank one within max_rounds (two
// dominant eigenvalues of equal modulus) or tr(M) = 0.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void overlap_square_d4_kernel(OverlapArgs p) {
  constexpr int WAVES = 4;
  __shared__ double2 sT_all[WAVES][kSquareD4Scratch];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double2* sT = sT_all[wave];
  const double tol2 = p.tol * p.tol;
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wave; b < p.B; b += (int64_t)gridDim.x * WAVES) {
    if (overlap_skipped(p, b)) continue;
    // (set-up, squaring rounds, eta: qmps_overlap_d4.h)
    double eta_r, eta_i;
    int rounds, status;
    v4f64 mr, mi;
    overlap_square_d4_item((const double2*)p.A + overlap_ref_index(p, b) * 32, (const double2*)p.Bt + b * 32, (const double2*)p.WW, sT, p.max_rounds, tol2,
                           eta_r, eta_i, rounds, status, mr, mi);
    if (lane == 0) overlap_store(p, b, eta_r, eta_i, rounds, status);
    if (p.l_out != nullptr && lane == 0) overlap_store(p, b + p.B, eta_r, eta_i, rounds, status);      // (the left solve's record)
    if ((p.r_out != nullptr && p.adjoint) || p.l_out != nullptr) {
      // LEFT fixed point (T^+ y = conj(eta) y): M -> u v^+, so every row of M is a multiple of v^+; y = conj of the largest row
      void* lo = p.l_out != nullptr ? p.l_out : p.r_out;
      double rn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rn[q] = row16_sum(dfma(mr[q], mr[q], mi[q] * mi[q]));      // |row 4 q + g|^2, in the 16 lanes of row group g
      __builtin_amdgcn_wave_barrier();
      if (c == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[4 * q + g] = make_double2(rn[q], 0.0);
      }
      __builtin_amdgcn_wave_barrier();
      int best = 0;
      double bn = -1.0;
      for (int k = 0; k < 16; ++k) {
        const double v = sT[k].x;
        if (v > bn) { bn = v; best = k; }
      }
      __builtin_amdgcn_wave_barrier();
      if (g == (best & 3)) {
        double vr = 0.0, vi = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q == (best >> 2)) { vr = mr[q]; vi = mi[q]; }
        sT[16 + c] = make_double2(vr, -vi);
      }
      __builtin_amdgcn_wave_barrier();
      const double inv = bn > 0.0 ? 1.0 / __builtin_sqrt(bn) : 0.0;
      if (lane < 16) {
        const double2 u = sT[16 + lane];
        ((double2*)((char*)lo + overlap_slot_offset(p)))[b * 16 + lane] = make_double2(u.x * inv, u.y * inv);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (p.r_out != nullptr && !p.adjoint) {
      // right fixed point = the largest column of M, unit Frobenius norm
      double cn = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) cn = dfma(mr[q], mr[q], dfma(mi[q], mi[q], cn));
      cn = group4_sum(cn);                       // norm^2 of column c, in every row group
      __builtin_amdgcn_wave_barrier();
      if (g == 0) sT[c] = make_double2(cn, 0.0);
      __builtin_amdgcn_wave_barrier();
      int best = 0;
      double bn = -1.0;
      for (int k = 0; k < 16; ++k) {
        const double v = sT[k].x;
        if (v > bn) { bn = v; best = k; }
      }
      __builtin_amdgcn_wave_barrier();
      if (c == best) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[16 + 4 * q + g] = make_double2(mr[q], mi[q]);
      }
      __builtin_amdgcn_wave_barrier();
      const double inv = bn > 0.0 ? 1.0 / __builtin_sqrt(bn) : 0.0;
      if (lane < 16) {
        const double2 u = sT[16 + lane];
        ((double2*)((char*)p.r_out + overlap_slot_offset(p)))[b * 16 + lane] = make_double2(u.x * inv, u.y * inv);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// D = 16, FOUR WAVES PER EVALUATION (small batches: BASELINE.json configs[4] shards its trajectories over the GPUs, a GPU
// holds tens to hundreds of candidates).  Wave w of the workgroup owns the physical index s = w: it keeps only C_w and
// Bm_w, computes n_w = C_w (x Bm_w^+) - 32 of the step's 128 MFMAs - and the four partial maps are summed through LDS (same
// order in every wave, so all four hold bit-identical x' and take the same decisions).  One wave per evaluation leaves a
// quarter of the SIMDs idle at B = 768 and, worse, lets the launch wait for its slowest candidate at one wave's pace; here
// the stragglers run on four SIMDs each.
// ------------------------------------------------------------------------------------------
// (the body as a device function: the workgroups of one launch may run different instantiations - overlap_mfma_d16x4_pair_kernel)
// DEFL: with deflation steps (cold starts - the instantiation for warm-started batches is the lean loop: the bookkeeping of the
// deflation steps costs 6 - 8 % of a step even where it never fires, measured on the evolve workload whose batches are all warm)
template <bool ADJ, bool DEFL>
__device__ __forceinline__ void overlap_mfma_d16x4_body(const OverlapArgs& p, int64_t b_first, int64_t b_stride, double2 (*sT_all)[16 * 17],
                                                        double2 (*sX_all)[16 * 16]) {
  constexpr int D = 16, LD = 17;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double2* sT = sT_all[wave];
  auto to_a_layout = [&](const v4f64& re, const v4f64& im, double (&are)[4], double (&aim)[4]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(re[q], im[q]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 t = sT[c * LD + 4 * kk + g];
      are[kk] = t.x;
      aim[kk] = t.y;
    }
  };
  // every wave publishes one C-layout matrix; afterwards fetch(w, ...) reads wave w's
  auto publish = [&](const v4f64& re, const v4f64& im) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sX_all[wave][q * 64 + lane] = make_double2(re[q], im[q]);
  };
  auto fetch = [&](int w, v4f64& re, v4f64& im) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 t = sX_all[w][q * 64 + lane];
      re[q] = t.x;
      im[q] = t.y;
    }
  };
  const int64_t slot_off = overlap_slot_offset(p);
  // p.queue != nullptr: the workgroups PULL their evaluations from a counter in HBM (dynamic: a workgroup that drew a slow
  // candidate - hundreds of power steps against tens - does not hold up the candidates a static stride would have queued behind
  // it); the counter value travels to the four waves through the (then idle) first exchange word
  for (int64_t b = b_first;; b += b_stride) {
    if (p.queue != nullptr) {
      if (threadIdx.x == 0) ((int*)&sX_all[0][0])[0] = atomicAdd(p.queue, 1);
      __syncthreads();
      b = ((const int*)&sX_all[0][0])[0];
      __syncthreads();
    }
    if (b >= p.B) break;
    if (overlap_skipped(p, b)) continue;          // (uniform over the workgroup: all four waves see the same b)
    const double tol2 = overlap_tol2(p, b);
    const double2* Ap = (const double2*)p.A + overlap_ref_index(p, b) * (2 * D * D);
    const double2* Bp = (const double2*)p.Bt + b * (2 * D * D);
    const double2* W = (const double2*)p.WW;
    const int t1 = wave >> 1, t2 = wave & 1;          // this wave's pair (s = 2 t1 + t2 = wave)
    // ---- set-up: wave t forms AA_t = A_t1 A_t2 (published) and Bm_t = B_t1 B_t2 (kept); then C_w = sum_t WW[w][t] AA_t
    double cre[4], cim[4];       // C_w in A-layout
    double bre[4], bimn[4];      // conj(Bm_w) in A-layout == Bm_w^+ in B-layout
    {
      double pa[4], pai[4], pb[4], pbi[4];
      v4f64 qa, qai, qb, qbi;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 va = Ap[(t1 * D + c) * D + 4 * kk + g], vb = Bp[(t1 * D + c) * D + 4 * kk + g];
        pa[kk] = va.x; pai[kk] = va.y;
        pb[kk] = vb.x; pbi[kk] = vb.y;
        const double2 wa = Ap[(t2 * D + 4 * kk + g) * D + c], wb = Bp[(t2 * D + 4 * kk + g) * D + c];
        qa[kk] = wa.x; qai[kk] = wa.y;
        qb[kk] = wb.x; qbi[kk] = wb.y;
      }
      v4f64 zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
      cmma16(pa, pai, qa, qai, zr, zi);
      publish(zr, zi);
      v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
      cmma16(pb, pbi, qb, qbi, yr, yi);
      if constexpr (ADJ) {      // Bm_w in B-layout as it comes out of the product (see overlap_mfma_d16_kernel)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bre[kk] = yr[kk];
          bimn[kk] = yi[kk];
        }
      } else {
        double tr[4], ti[4];
        to_a_layout(yr, yi, tr, ti);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bre[kk] = tr[kk];
          bimn[kk] = -ti[kk];
        }
      }
      __syncthreads();
      v4f64 sr = {0, 0, 0, 0}, si = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v4f64 ar, ai;
        fetch(t, ar, ai);
        const double2 w = W[wave * 4 + t];
        sr += w.x * ar - w.y * ai;
        si += w.x * ai + w.y * ar;
      }
      if constexpr (ADJ) {      // C_w^+ in A-layout = conj of C_w in C-layout
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          cre[kk] = sr[kk];
          cim[kk] = -si[kk];
        }
      } else {
        to_a_layout(sr, si, cre, cim);
      }
      __syncthreads();               // the exchange buffers are free again
    }
    // ---- power method, x in C-layout (alike in the four waves), ||x||_F = 1
    v4f64 xr, xi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 x0 = overlap_cold_start(4 * q + g, c, 16);      // (generic: see overlap_cold_start)
      xr[q] = x0.x;
      xi[q] = x0.y;
    }
    if (p.x_in != nullptr) {      // warm start: the resident fixed point of this candidate's slot (all zero: none yet)
      const double2* xi_ = (const double2*)((const char*)p.x_in + slot_off) + (p.x_in_group > 0 ? b / p.x_in_group : b) * (D * D);
      v4f64 wr, wi;
      double n2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 t = xi_[(4 * q + g) * D + c];
        wr[q] = t.x; wi[q] = t.y;
        n2 = dfma(t.x, t.x, dfma(t.y, t.y, n2));
      }
      n2 = lane0(wave_sum(n2));
      if (n2 > 1e-200 && n2 < 1e200) {
        const double inv = 1.0 / __builtin_sqrt(n2);
        xr = wr * inv;
        xi = wi * inv;
      }
    }
    double eta_r = 0.0, eta_i = 0.0;
    int iters = 0, status = QMPS_ST_NOT_CONVERGED;
    // Deflation steps (see overlap_block_kernel: a slowly decaying second eigenvector is shifted out, x <- T x - sigma x).  Here
    // the tests come every second step: a test at step k that finds the iteration slow keeps its residual r_k = n_k - eta x_k;
    // step k + 1 works on x_{k+1} = n_k / |n_k|, so its product gives T r_k = (n_{k+1} - eta x_{k+1}) |n_k| and with it
    // sigma = <r_k, T r_k> / <r_k, r_k> - and holds both vectors of the shifted step.  All four waves hold the same numbers.
    bool defl_on = DEFL && p.no_deflation == 0, pending = false, have_prev_sigma = false;
    v4f64 rr = {0, 0, 0, 0}, ri = {0, 0, 0, 0};
    double rs_chk = 0.0, inv_chk = 0.0, res_prev = 0.0, sgr_prev = 0.0, sgi_prev = 0.0, sig_max2 = 0.0;
    int last_deflation = 0;
    const int give_up_after = (p.r_out != nullptr && p.kry_counter != nullptr) ? p.krylov_after : 0;       // hand-over to the Krylov fall-back (see OverlapArgs)
    int k_ref = 0;
    float l_ref = 0.0f;
    for (int k = 1; k <= p.max_rounds; ++k) {
      double xar[4], xai[4];
      to_a_layout(xr, xi, xar, xai);
      v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0}, pr = {0, 0, 0, 0}, pi = {0, 0, 0, 0};
      const v4f64 qre = {bre[0], bre[1], bre[2], bre[3]};
      const v4f64 qim = {bimn[0], bimn[1], bimn[2], bimn[3]};
      cmma16_3m(xar, xai, qre, qim, yr, yi);           // Y_w = x Bm_w^+
      cmma16_3m(cre, cim, yr, yi, pr, pi);             // C_w Y_w
      // the four partial maps are exchanged through one of TWO sets of buffers, alternately: a wave may publish step k + 1
      // while another still reads step k, so a step needs ONE workgroup barrier (a set is reused two barriers later)
      double2 (*sXk)[16 * 16] = sX_all + 4 * (k & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) sXk[wave][q * 64 + lane] = make_double2(pr[q], pi[q]);
      __syncthreads();
      v4f64 nr = {0, 0, 0, 0}, ni = {0, 0, 0, 0};
#pragma unroll
      for (int w = 0; w < 4; ++w) {                 // the same order in every wave: bit-identical sums
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2 t = sXk[w][q * 64 + lane];
          nr[q] += t.x;
          ni[q] += t.y;
        }
      }
      iters = k;
      if (!overlap_check_step(k, p.max_rounds)) {   // no test on this step (see overlap_check_step)
        if (DEFL && pending) {
          pending = false;
          double b0 = 0.0, b1 = 0.0;
          v4f64 tr_, ti_;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            tr_[q] = nr[q] - (eta_r * xr[q] - eta_i * xi[q]);
            ti_[q] = ni[q] - (eta_r * xi[q] + eta_i * xr[q]);
            b0 = dfma(rr[q], tr_[q], dfma(ri[q], ti_[q], b0));        // conj(r) (T r)
            b1 = dfma(rr[q], ti_[q], dfma(-ri[q], tr_[q], b1));
          }
          const double den = inv_chk * rs_chk;
          const double sgr = den > 0.0 ? lane0(wave_sum(b0)) / den : 0.0, sgi = den > 0.0 ? lane0(wave_sum(b1)) / den : 0.0;
          const double s2 = sgr * sgr + sgi * sgi, e2 = eta_r * eta_r + eta_i * eta_i;
          const double ds = (sgr - sgr_prev) * (sgr - sgr_prev) + (sgi - sgi_prev) * (sgi - sgi_prev);
          const bool settled = have_prev_sigma && ds < 1e-8 * s2;
          sgr_prev = sgr;
          sgi_prev = sgi;
          have_prev_sigma = true;
          if (settled && s2 < 0.998 * e2 && s2 > 0.25 * e2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double ar = nr[q] - (sgr * xr[q] - sgi * xi[q]), ai = ni[q] - (sgr * xi[q] + sgi * xr[q]);
              xr[q] = ar;
              xi[q] = ai;
            }
            last_deflation = k;
            sig_max2 = s2 > sig_max2 ? s2 : sig_max2;
            have_prev_sigma = false;
            res_prev = 0.0;
            k_ref = 0;
            continue;
          }
        }
        xr = nr;
        xi = ni;
        continue;
      }
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a0 = dfma(xr[q], nr[q], a0);
        a0 = dfma(xi[q], ni[q], a0);
        a1 = dfma(xr[q], ni[q], a1);
        a1 = dfma(-xi[q], nr[q], a1);
        a2 = dfma(nr[q], nr[q], a2);
        a2 = dfma(ni[q], ni[q], a2);
        a3 = dfma(xr[q], xr[q], a3);
        a3 = dfma(xi[q], xi[q], a3);
      }
      const double xx = wave_sum(a3), ixx = xx > 0.0 ? 1.0 / xx : 0.0;
      eta_r = wave_sum(a0) * ixx;
      eta_i = wave_sum(a1) * ixx;
      const double nn = wave_sum(a2);
      double rs = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double dr = nr[q] - (eta_r * xr[q] - eta_i * xi[q]), di = ni[q] - (eta_r * xi[q] + eta_i * xr[q]);
        rs = dfma(dr, dr, rs);
        rs = dfma(di, di, rs);
      }
      const double rs_sum = lane0(wave_sum(rs));
      const double res2 = rs_sum * ixx;
      if (res2 < tol2) {
        const double e2 = eta_r * eta_r + eta_i * eta_i;
        if (DEFL && sig_max2 > 0.0 && !(e2 > sig_max2 * (1.0 + 1e-3))) {
          // converged - but not to an eigenvalue that dominates every shift applied (a nearly degenerate dominant pair: the shift
          // took out the wrong one): start again as a plain power method
          defl_on = false;
          sig_max2 = 0.0;
          pending = false;
          k_ref = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double2 x0 = overlap_cold_start(4 * q + g, c, 16);
            xr[q] = x0.x;
            xi[q] = x0.y;
          }
          continue;
        }
        status = QMPS_ST_OK;
        const double inv = xx > 0.0 ? 1.0 / __builtin_sqrt(xx) : 0.0;
        xr *= inv;
        xi *= inv;
        break;
      }
      const double inv = nn > 0.0 ? 1.0 / __builtin_sqrt(nn) : 0.0;
      // slow (the residual shrank by less than 0.8 per step since the last test)?  keep it for the estimate of the next step
      pending = DEFL && defl_on && k >= 24 && k - last_deflation >= 12 && res_prev > 0.0 && res2 > 0.41 * res_prev && k + 1 < p.max_rounds;
      if (DEFL && pending) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rr[q] = nr[q] - (eta_r * xr[q] - eta_i * xi[q]);
          ri[q] = ni[q] - (eta_r * xi[q] + eta_i * xr[q]);
        }
        rs_chk = rs_sum;
        inv_chk = inv;
      } else {
        have_prev_sigma = false;
      }
      res_prev = res2;
      xr = nr * inv;
      xi = ni * inv;
      // a long tail ahead: stop here (status 1, k < max_rounds) - overlap_krylov_kernel takes the candidate over from x
      if (k < p.max_rounds && power_gives_up(k, res2, tol2, give_up_after, k_ref, l_ref)) {
        if (threadIdx.x == 0) atomicAdd(p.kry_counter + 2, 1);
        break;
      }
    }
    if (wave == 0) {
      if (lane == 0) overlap_store(p, b, eta_r, ADJ ? -eta_i : eta_i, iters, status);
      if (p.r_out != nullptr) {
        double2* ro = (double2*)((char*)p.r_out + slot_off) + b * (D * D);
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[(4 * q + g) * D + c] = make_double2(xr[q], xi[q]);
      }
    }
    __syncthreads();
  }
}

template <bool ADJ, bool DEFL>
__global__ __launch_bounds__(256, 3) void overlap_mfma_d16x4_kernel(OverlapArgs p) {
  __shared__ double2 sT_all[4][16 * 17];          // wave-private transposes
  __shared__ double2 sX_all[8][16 * 16];          // exchange: two sets of one C-layout matrix per wave, element (q, lane) at [q * 64 + lane]
  overlap_mfma_d16x4_body<ADJ, DEFL>(p, blockIdx.x, gridDim.x, sT_all, sX_all);
}

// RIGHT and LEFT fixed points in one launch (qmps_overlap_gradient): workgroups [0, n_right) run the map of `pr`, the others the
// adjoint map of `pl` - twice the waves in flight for the same length of the (latency-bound) iteration chain
// Round 5: the workgroups behind the solves, [n_right + n_left, gridDim), BUILD THE CENTRAL-DIFFERENCE NEIGHBOURS' TENSORS of the same
// gradient evaluation (nb.params != nullptr; ShallowCNOT families: the wave-distributed circuit of qmps_circuit_wave.h, one wave
// per neighbour, two columns per pass).  The solves are a latency-bound chain on the matrix cores whose stragglers leave most of
// the chip idle; the builds are vector work with no dependence on them - in round 4 a kernel of their own on a second stream,
// which cost the critical path two cross-stream dependencies (~7 us each) per evaluation.  Solver workgroups come first in the
// grid, so they are dispatched first; the builders fill what is left and the tail.
template <bool DEFL>
__global__ __launch_bounds__(256, 3) void overlap_mfma_d16x4_pair_kernel(OverlapArgs pr, OverlapArgs pl, int n_right, int n_left, NeighbourBuildArgs nb) {
  __shared__ double2 sT_all[4][16 * 17];
  __shared__ double2 sX_all[8][16 * 16];
  if ((int)blockIdx.x < n_right) overlap_mfma_d16x4_body<false, DEFL>(pr, blockIdx.x, n_right, sT_all, sX_all);
  else if ((int)blockIdx.x < n_right + n_left) overlap_mfma_d16x4_body<true, DEFL>(pl, blockIdx.x - n_right, n_left, sT_all, sX_all);
  else {
    const int lane = threadIdx.x & 63, a = lane & 31;
    const int P = nb.n_params;
    const int64_t n_neigh = nb.rows * 2 * P;
    for (int64_t b = ((int64_t)blockIdx.x - n_right - n_left) * 4 + (threadIdx.x >> 6); b < n_neigh; b += ((int64_t)gridDim.x - n_right - n_left) * 4) {
      const int64_t row = b / (2 * P);
      if (nb.active != nullptr && nb.active[row] == 0) continue;       // (a skipped trajectory's neighbours are never read)
      const int k = (int)(b - row * 2 * P), isel = k % P;
      // one sincos per angle and wave: lane l takes angle l (P <= 64)
      double cn = 1.0, sn = 0.0;
      if (lane < P) {
        double v = nb.params[row * P + lane];
        if (lane == isel) v += k < P ? nb.h : -nb.h;
        sincos(0.5 * v, &sn, &cn);
      }
      double2* out = (double2*)nb.out + b * 512;
#pragma unroll 1
      for (int w = 0; w < 8; ++w) {
        const int j = 2 * w + (lane >> 5);
        double re, im;
        if (nb.kind == 3) shallow_cnot_wave_column_d16<3>(cn, sn, P, j, re, im);
        else shallow_cnot_wave_column_d16<0>(cn, sn, P, j, re, im);
        out[((a & 1) * 16 + (a >> 1)) * 16 + j] = make_double2(re, im);      // A[s][i][j] = amplitude[2 i + s] of column j
      }
    }
  }
}

// the Krylov fall-back behind a power launch (both solves of a pair launch): candidates given up are finished, the others untouched
static hipError_t launch_krylov_after(int D, const OverlapArgs& a, const OverlapArgs* second, hipStream_t st) {
  const bool first = a.krylov_after > 0 && a.kry_counter != nullptr;
  const bool both = first && second != nullptr && second->krylov_after > 0 && second->kry_counter != nullptr;
  if (both) return launch_overlap_krylov_pair(D, a, *second, st);
  if (first)
    if (hipError_t e = launch_overlap_krylov(D, a, a.kry_counter, st); e != hipSuccess) return e;
  if (second != nullptr && second->krylov_after > 0 && second->kry_counter != nullptr)
    if (hipError_t e = launch_overlap_krylov(D, *second, second->kry_counter, st); e != hipSuccess) return e;
  return hipSuccess;
}

hipError_t launch_overlap_pair_d8(const OverlapArgs& right, const OverlapArgs& left, hipStream_t st, bool krylov_now) {
  if (right.B <= 0) return hipSuccess;
  hipLaunchKernelGGL((overlap_block_pair_kernel<8>), dim3((unsigned)(right.B + left.B)), dim3(64), 0, st, right, left, (int)right.B);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  return krylov_now ? launch_krylov_after(8, right, &left, st) : hipSuccess;
}

hipError_t launch_overlap_pair_d16(const OverlapArgs& right, const OverlapArgs& left, hipStream_t st, bool krylov_now, const NeighbourBuildArgs* build) {
  if (right.B <= 0) return hipSuccess;
  const int nr = (int)(right.B < 2048 ? right.B : 2048), nl = (int)(left.B < 2048 ? left.B : 2048);
  NeighbourBuildArgs nb;
  memset(&nb, 0, sizeof(nb));
  int n_build = 0;
  if (build != nullptr && build->params != nullptr && build->rows > 0) {
    nb = *build;
    const int64_t n_neigh = nb.rows * 2 * nb.n_params;
    n_build = (int)((n_neigh + 3) / 4 < 4096 ? (n_neigh + 3) / 4 : 4096);      // one wave per neighbour (a stride beyond 16 384 of them)
  }
  // (deflation steps for cold starts only, see overlap_mfma_d16x4_body)
  if (right.x_in == nullptr && right.no_deflation == 0) hipLaunchKernelGGL(overlap_mfma_d16x4_pair_kernel<true>, dim3((unsigned)(nr + nl + n_build)), dim3(256), 0, st, right, left, nr, nl, nb);
  else hipLaunchKernelGGL(overlap_mfma_d16x4_pair_kernel<false>, dim3((unsigned)(nr + nl + n_build)), dim3(256), 0, st, right, left, nr, nl, nb);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  return krylov_now ? launch_krylov_after(16, right, &left, st) : hipSuccess;
}

hipError_t launch_overlap_d(int D, const OverlapArgs& a, bool mfma, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  if (a.adjoint && D == 4 && !mfma) {      // (the squaring kernel hands out left fixed points itself: the largest row of the squared map)
    hipLaunchKernelGGL((overlap_block_kernel<4, true>), dim3((unsigned)((a.B + 3) / 4)), dim3(64), 0, st, a);
    return hipGetLastError();
  }
  switch (D) {
    case 4:
      if (mfma) {
        int grid = (int)((a.B + 3) / 4);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(overlap_square_d4_kernel, dim3(grid), dim3(256), 0, st, a);
      } else {
        hipLaunchKernelGGL((overlap_block_kernel<4>), dim3((unsigned)((a.B + 3) / 4)), dim3(64), 0, st, a);
      }
      break;
    case 8:
      if (a.adjoint) hipLaunchKernelGGL((overlap_block_kernel<8, true>), dim3((unsigned)a.B), dim3(64), 0, st, a);
      else hipLaunchKernelGGL((overlap_block_kernel<8>), dim3((unsigned)a.B), dim3(64), 0, st, a);
      break;
    case 16:
      if (mfma) {
        // few candidates: four waves per evaluation (the launch waits for its slowest candidate - give it four SIMDs);
        // many: one wave per evaluation (no exchange through LDS, same MFMA work)
        // four waves per evaluation at every batch size (a queue in HBM hands out the evaluations when there are more of them than
        // workgroups); QMPS_D16_ONE_WAVE: round 2's one-wave-per-evaluation kernel for batches above 2 048
        if (a.B <= 2048 || a.queue != nullptr) {
          const dim3 grid((unsigned)(a.B < 2048 ? a.B : 2048));
          const bool defl = a.x_in == nullptr && a.no_deflation == 0;       // cold starts only (see overlap_mfma_d16x4_body)
          if (a.adjoint && defl) hipLaunchKernelGGL((overlap_mfma_d16x4_kernel<true, true>), grid, dim3(256), 0, st, a);
          else if (a.adjoint) hipLaunchKernelGGL((overlap_mfma_d16x4_kernel<true, false>), grid, dim3(256), 0, st, a);
          else if (defl) hipLaunchKernelGGL((overlap_mfma_d16x4_kernel<false, true>), grid, dim3(256), 0, st, a);
          else hipLaunchKernelGGL((overlap_mfma_d16x4_kernel<false, false>), grid, dim3(256), 0, st, a);
        } else {
          int grid = (int)((a.B + 3) / 4);
          if (grid > 4096) grid = 4096;
          if (a.adjoint) hipLaunchKernelGGL(overlap_mfma_d16_kernel<true>, dim3(grid), dim3(256), 0, st, a);
          else hipLaunchKernelGGL(overlap_mfma_d16_kernel<false>, dim3(grid), dim3(256), 0, st, a);
        }
      } else {
        if (a.adjoint) hipLaunchKernelGGL((overlap_block_kernel<16, true>), dim3((unsigned)a.B), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((overlap_block_kernel<16>), dim3((unsigned)a.B), dim3(256), 0, st, a);
      }
      break;
    default: return hipErrorInvalidValue;
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (D == 8 || (D == 16 && mfma && (a.B <= 2048 || a.queue != nullptr))) return launch_krylov_after(D, a, nullptr, st);
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// Kernel 3d: time-evolution overlap objective (qmps/new_time_evolve.py:193-221,
// scripts/loschmidt.py:209-239), D = 2, one evaluation per lane.
//   T(x) = sum_{s=0..3} C_s x Bm_s^+ ,  C = WW . merge(A, A) ,  Bm = merge(B, B)
// (qmps/time_evolve_tools.py:20-23); the 6-qubit circuit of the reference measures
// 2 |psi[0]| = |eta|, eta the dominant eigenvalue of T, and the objective is -sqrt(|eta|).
// eta is found by power iteration applied 2^m steps at a time: the 4 x 4 complex matrix of T is
// squared (Frobenius-normalised each round), its dominant right vector v read off the largest column,
// eta = <v, T v>/<v, v>, stop when ||T v - eta v|| < tol ||v||.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void overlap_lane_kernel(OverlapArgs p) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B || overlap_skipped(p, b)) return;
  const double2* Ap = (const double2*)p.A + overlap_ref_index(p, b) * 8;
  const double2* Bp = (const double2*)p.Bt + b * 8;
  OverlapLaneResult o;
  overlap_lane_solve([&](int k) { return Ap[k]; }, [&](int k) { return Bp[k]; }, (const double2*)p.WW, p.max_rounds, p.tol, o);
  overlap_store(p, b, o.eta_r, o.eta_i, o.rounds, o.status);
  if (p.r_out != nullptr) {
    double n2 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) n2 += o.vr[a] * o.vr[a] + o.vi[a] * o.vi[a];
    const double inv = n2 > 0.0 ? 1.0 / __builtin_sqrt(n2) : 0.0;
    double2* ro = (double2*)((char*)p.r_out + overlap_slot_offset(p));
#pragma unroll
    for (int a = 0; a < 4; ++a) ro[b * 4 + a] = make_double2(o.vr[a] * inv, o.vi[a] * inv);
  }
}

hipError_t launch_overlap(const OverlapArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  hipLaunchKernelGGL(overlap_lane_kernel, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, st, a);
  return hipGetLastError();
}


}  // namespace qmps
