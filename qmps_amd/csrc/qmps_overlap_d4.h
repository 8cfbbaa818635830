// qmps_overlap_d4.h - the D = 4 time-evolution overlap solve of ONE wave on the matrix cores (gfx950 only), shared by
// overlap_square_d4_kernel (qmps_overlap.hip) and the device-resident BFGS time evolution (qmps_evolve_d4.hip); also the small complex
// helpers and the complex 16 x 16 x 16 products on v_mfma_f64_16x16x4 of the overlap kernels.
// Reference: qmps/new_time_evolve.py:193-221, scripts/loschmidt.py:209-239, qmps/time_evolve_tools.py:20-23.
#pragma once
#include <hip/hip_runtime.h>

#include "qmps_kernels.h"
#include "qmps_device.h"

namespace qmps {

namespace {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ void cfma(double2 a, double2 b, double2& c) {   // c += a b
  c.x = dfma(a.x, b.x, c.x);
  c.x = dfma(-a.y, b.y, c.x);
  c.y = dfma(a.x, b.y, c.y);
  c.y = dfma(a.y, b.x, c.y);
}
__device__ __forceinline__ void cfma_conj(double2 a, double2 b, double2& c) {   // c += a conj(b)
  c.x = dfma(a.x, b.x, c.x);
  c.x = dfma(a.y, b.y, c.x);
  c.y = dfma(a.y, b.x, c.y);
  c.y = dfma(-a.x, b.y, c.y);
}

}  // namespace

namespace {

// C += P * Q, P in A-layout (pre/pim[kk] = P[row = c][k = 4 kk + g]), Q in B-layout (qre/qim[kk] = Q[k = 4 kk + g][col = c])
__device__ __forceinline__ void cmma16(const double (&pre)[4], const double (&pim)[4], const v4f64& qre, const v4f64& qim,
                                       v4f64& cre, v4f64& cim) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    cre = __builtin_amdgcn_mfma_f64_16x16x4f64(pre[kk], qre[kk], cre, 0, 0, 0);
    cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pre[kk], qim[kk], cim, 0, 0, 0);
    cre = __builtin_amdgcn_mfma_f64_16x16x4f64(-pim[kk], qim[kk], cre, 0, 0, 0);
    cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pim[kk], qre[kk], cim, 0, 0, 0);
  }
}

// The same product with THREE real products per k-slab instead of four (K1 = (Pr + Pi) Qr, K2 = Pr (Qi - Qr), K3 = Pi (Qr + Qi);
// Re = K1 - K3, Im = K1 + K2): 12 v_mfma_f64_16x16x4 per complex 16 x 16 x 16 product instead of 16, in three independent
// accumulator chains of four.  The matrix pipe is what bounds the power iteration (a v_mfma_f64_16x16x4 occupies it for ~100
// cycles on this part, profiles/EXPERIMENTS.md), the handful of extra additions run on the vector pipe beside it.  Rounding:
// norm-wise the same bound as the four-product form (|error| <= c eps |P| |Q|).
__device__ __forceinline__ void cmma16_3m(const double (&pre)[4], const double (&pim)[4], const v4f64& qre, const v4f64& qim,
                                          v4f64& cre, v4f64& cim) {
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


// The mixed transfer map of D = 4 IS one complex 16 x 16 tile, E[(i,i'),(j,j')] = sum_{s<4} C_s[i][j] conj(Bm_s[i'][j']); the power method
// is taken 2^m steps at a time by SQUARING it (16 v_mfma_f64_16x16x4 per round, Frobenius-normalised) until it is rank one
// (||M M - tr(M) M||_F < tol ||M M||_F), then eta = tr(M E)/tr(M); 30 - 44 rounds without one: TIED dominant eigenvalues, eta = their modulus
// (Gelfand), QMPS_ST_TIED.  One wave; Ap / Bp: reference / candidate tensors [2][4][4]
// (any address space); sT: the wave's LDS scratch [kSquareD4Scratch]; (mr, mi): the final M in the accumulator layout (its largest column
// / row are the right / left fixed points).  All results wave-uniform.  sT: kSquareD4Scratch entries per wave.
constexpr int kSquareD4Keep = 16 * 17, kSquareD4Scratch = kSquareD4Keep + 128;
__device__ __forceinline__ void overlap_square_d4_item(const double2* Ap, const double2* Bp, const double2* W, double2* sT, int max_rounds, double tol2,
                                                       double& eta_r_out, double& eta_i_out, int& rounds_out, int& status_out, v4f64& mr_out, v4f64& mi_out) {
  constexpr int LD = 17;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
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
  double trr = 0.0, tri = 0.0;
    // ---- set-up through LDS: inputs at sT[0..63], then C_s[i][j] at sT[64 + 16 s + 4 i + j], Bm_s at sT[128 + ...]
    {
      __builtin_amdgcn_wave_barrier();
      sT[lane] = lane < 32 ? Ap[lane] : Bp[lane - 32];
      __builtin_amdgcn_wave_barrier();
      const int s = lane >> 4, i = (lane >> 2) & 3, j = lane & 3, s1 = s >> 1, s2 = s & 1;
      double2 cs = make_double2(0.0, 0.0), bm = make_double2(0.0, 0.0);
#pragma unroll
      for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          double2 aa = make_double2(0.0, 0.0);
#pragma unroll
          for (int k = 0; k < 4; ++k) cfma(sT[(t1 * 4 + i) * 4 + k], sT[(t2 * 4 + k) * 4 + j], aa);
          cfma(W[s * 4 + 2 * t1 + t2], aa, cs);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) cfma(sT[32 + (s1 * 4 + i) * 4 + k], sT[32 + (s2 * 4 + k) * 4 + j], bm);
      __builtin_amdgcn_wave_barrier();
      sT[64 + lane] = cs;
      sT[128 + lane] = bm;
      sT[kSquareD4Keep + lane] = cs;            // (a copy the transposes of the squaring rounds do not overwrite: E^T is formed from it at the end)
      sT[kSquareD4Keep + 64 + lane] = bm;
      __builtin_amdgcn_wave_barrier();
    }
    // E and its transpose in C-layout: register q of lane (g, c) = element [row 4 q + g][col c], row = (i, i'), col = (j, j')
    v4f64 er, ei;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 e = make_double2(0.0, 0.0);
#pragma unroll
      for (int s = 0; s < 4; ++s) cfma_conj(sT[64 + 16 * s + 4 * q + (c >> 2)], sT[128 + 16 * s + 4 * g + (c & 3)], e);    // E[(q,g)][(c>>2,c&3)]
      er[q] = e.x; ei[q] = e.y;
    }
    // ---- squaring rounds
    v4f64 mr = er, mi = ei;
    double hist = 0.0;      // lane m: ||M_m M_m||^2 of round m, lane 63: ||E||^2 - what a tie's modulus is read from at the end (a select per round; through LDS it cost 2 %)
    {
      double n2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) n2 = dfma(mr[q], mr[q], dfma(mi[q], mi[q], n2));
      n2 = wave_sum(n2);
      const double inv = n2 > 0.0 ? fast_rsqrt(n2) : 0.0;        // (v_rsq_f64 + a Newton step: the scale only keeps the powers O(1))
      mr *= inv;
      mi *= inv;
      hist = lane == 63 ? n2 : hist;
    }
    int rounds = 0, status = QMPS_ST_NOT_CONVERGED;
    bool nilpotent = false, collapsed = false;
    for (int m = 0; m <= max_rounds; ++m) {
      double ar[4], ai[4];
      to_a_layout(mr, mi, ar, ai);
      v4f64 qr = {0, 0, 0, 0}, qi = {0, 0, 0, 0};
      cmma16_3m(ar, ai, mr, mi, qr, qi);      // (three real products per k-slab: 12 instead of 16 v_mfma_f64_16x16x4 per squaring)                     // Q = M M
      // ||M M||_F^2 every round (the next power is normalised by it, a collapsed power shows in it); the rank-one test - the trace, the residual:
      // three more wave reductions and their latencies - only from round kFirstTest on (a map of this objective needs 5 - 10 squarings at tol
      // 1e-12; one that would have passed earlier squares on to round kFirstTest, harmlessly) and at the last round of the budget
      constexpr int kFirstTest = 3;
      const bool test = m >= kFirstTest || m == max_rounds;
      double res = 1e300, q2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) q2 = dfma(qr[q], qr[q], dfma(qi[q], qi[q], q2));
      if (test) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c == 4 * q + g) { d0 = mr[q]; d1 = mi[q]; }
        trr = wave_sum(d0);
        tri = wave_sum(d1);
        res = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double dr = qr[q] - (trr * mr[q] - tri * mi[q]), di = qi[q] - (trr * mi[q] + tri * mr[q]);
          res = dfma(dr, dr, dfma(di, di, res));
        }
        res = lane0(wave_sum(res));
      }
      q2 = lane0(wave_sum(q2));
      rounds = m;
      hist = lane == m ? q2 : hist;      // (m <= kLastPass < 63)
      if (q2 < 1e-28) {
        // ||M^2|| < 1e-14 ||M||: the power of the map has collapsed to rounding noise - a NILPOTENT map (reference and candidate orthogonal:
        // every eigenvalue vanishes).  Normalising that noise and squaring on used to 'converge' to the dominant direction of a random
        // matrix and return |eta| ~ 0.05 with status 0 (profiles/experiments/r05/stress_overlap.py); the answer is eta = 0.
        // (only within the first rounds - a nilpotent 16 x 16 map vanishes at the sixteenth power; a LATE collapse is a defective dominant
        // eigenvalue losing its digits: no answer, status 1)
        nilpotent = m <= 8;
        status = nilpotent ? QMPS_ST_OK : QMPS_ST_NOT_CONVERGED;
        collapsed = true;
        break;
      }
      if (res < tol2 * q2) {
        status = QMPS_ST_OK;
        break;
      }
      // No rank-one power by round kLastPass: none is believed later.  Dominant eigenvalues TIED in modulus (a non-injective state on the special
      // grid of the ansatz: 1, 1, -1, -1) keep M = P_1 + P_2 + .. for ever in exact arithmetic; in floating point every squaring moves the unit
      // eigenvalues of M by ~1e-16 and DOUBLES what is there already - round ~53 breaks the tie on noise, the power turns rank one and
      // tr(M E) / tr(M) of that noise-picked direction came back with status 0 (|eta| = 1.0008, 0.54 where it is 1, 0.999;
      // profiles/experiments/r06/grid_starts_probe.py).  A genuine gap that needs 45 squarings is 1e-12: a tie by any standard.
      constexpr int kLastPass = 44;
      if (m == max_rounds || m >= kLastPass) break;
      const double inv = fast_rsqrt(q2);
      mr = qr * inv;
      mi = qi * inv;
    }
    // eta = tr(M E)/tr(M):  tr(M E) = sum_{a,b} M[a][b] E[b][a] = sum over lanes and registers of M * E^T (elementwise); E^T from the
    // kept copy of C_s, Bm_s (16 registers less across the squaring rounds than carrying it)
    double nr = 0.0, ni = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 t = make_double2(0.0, 0.0);
#pragma unroll
      for (int s = 0; s < 4; ++s) cfma_conj(sT[kSquareD4Keep + 16 * s + 4 * (c >> 2) + q], sT[kSquareD4Keep + 64 + 16 * s + 4 * (c & 3) + g], t);    // E[(c>>2,c&3)][(q,g)]
      nr = dfma(mr[q], t.x, dfma(-mi[q], t.y, nr));
      ni = dfma(mr[q], t.y, dfma(mi[q], t.x, ni));
    }
    nr = wave_sum(nr);
    ni = wave_sum(ni);
    {
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (c == 4 * q + g) { d0 = mr[q]; d1 = mi[q]; }
      trr = wave_sum(d0);
      tri = wave_sum(d1);
    }
    const double den = trr * trr + tri * tri;
    double eta_r = 0.0, eta_i = 0.0;
    if (collapsed) {
      // (nilpotent: eta = 0 exactly, status 0; a late collapse: status 1.  The fixed points handed out are whatever the last power held)
    } else if (status != QMPS_ST_OK && rounds >= 30) {
      // ---- TIED dominant eigenvalues (as the D = 2 solvers, qmps_overlap_d2.h): their common modulus by Gelfand's formula from the norms the
      // rounds left in `hist` - E = n M_0, M_(m+1) = M_m M_m / s_m with ||M_m|| = 1:  ||E^(2^k)||^(1/2^k) = n prod_(m<k) s_m^(2^-(m+1)), good to
      // 2^-k log(condition) (3e-9 after 30 rounds, 1e-13 after 44) - as NESTED SQUARE ROOTS, sqrt(s_0 sqrt(s_1 sqrt(s_2 ..))): logarithms here cost
      // the squaring kernels 15 - 40 registers (the overlap kernel its fourth wave per SIMD), a call out of line 2 % of the overlap workload.
      // eta = rho (real), QMPS_ST_TIED: the objective -sqrt|eta| is the reference's whichever member ARPACK returns; (mr, mi) are the last
      // power - a mixture, no fixed point.
      double acc = 1.0;
#pragma nounroll
      for (int m = rounds; m >= 0; --m) acc = __builtin_sqrt(__builtin_sqrt(__shfl(hist, m, 64)) * acc);
      const double rho = __builtin_sqrt(__shfl(hist, 63, 64)) * acc;
      if (rho > 0.0 && rho < 1e300) {      // (a NaN map, a zero map: status 1)
        eta_r = rho;
        eta_i = 0.0;
        status = QMPS_ST_TIED;
      }
    } else if (den > 1e-280) {      // (status 1 - the budget ended before a rank-one power, too early to call it a tie: the estimate of the last power)
      eta_r = (nr * trr + ni * tri) / den;
      eta_i = (ni * trr - nr * tri) / den;
    } else {
      status = QMPS_ST_NOT_CONVERGED;
    }
  eta_r_out = eta_r;
  eta_i_out = eta_i;
  rounds_out = rounds;
  status_out = status;
  mr_out = mr;
  mi_out = mi;
}

}  // namespace

}  // namespace qmps
