// qmps_direct_d8.h - the D = 8 direct fixed-point solve (gfx950 only) as a device function, shared by env_direct_d8_kernel
// (qmps_direct.hip: writes the environments) and energy_block_kernel<8, true, FUSED> (qmps_kernels.hip: solve, acceptance step
// and energies in ONE launch for the small batches of BASELINE.json configs[3]).  See env_direct_d8_kernel for the layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_device.h"
#include "qmps_direct_core.h"    // static_for

namespace qmps {

// One Gauss-Jordan update of the D = 8 cyclic layout (see env_direct_d8_kernel): column class entry T of the lane's four
// rows, pivot row = local row KQ of lane KC of the same 16-lane DPP row.  The pivot row's own local row index goes last:
// its registers are the DPP source of the other three.
template <int KC, int KQ, int T>
__device__ __forceinline__ void d8_update(double (&Mn)[4][16], const double (&nf)[4]) {
#pragma unroll
  for (int mm = 1; mm <= 4; ++mm) {
    const int m = (KQ + mm) & 3;
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(Mn[m][T]) : "v"(Mn[KQ][T]), "v"(nf[m]), "n"(KC));
  }
}

// The elimination itself, on a matrix that is already in the cyclic layout (lane (g, c) = (lane >> 4, lane & 15): rows c + 16 m,
// columns g + 4 t).  Returns r[i][i'] of the lane, lane = 8 i + i' (trace 1; see env_direct_d8_solve).
// (A four-wave variant - all waves build the matrix, one eliminates - was costed and not built: splitting a row's 28 column pairs
// over four lanes needs lane-dependent operand selection, the build would drop from 1.6 to ~0.9 us only.)
__device__ __forceinline__ double2 env_direct_d8_eliminate(double (&Mn)[4][16], double (*sT)[9], int lane, long long* prof = nullptr) {
  constexpr int D = 8, N = 64;
  const int i = lane >> 3, ip = lane & 7;
  auto tick = [&](int k) {
    if (prof) {
      __builtin_amdgcn_sched_barrier(0);
      prof[k] = wall_clock64();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  __builtin_amdgcn_sched_barrier(0);
  tick(2);
  double x;
  {
    const int c = lane & 15;
    double dinv[4] = {0.0, 0.0, 0.0, 0.0}, yv[4] = {0.0, 0.0, 0.0, 0.0};
    auto from_lane = [](double v, int src) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    // (a 4-pivot blocked variant - panel fetched once per block, the four steps' multipliers from in-register panel elimination;
    // bit-identical - was built and measured slower: 1 920 instead of 2 176 multiply-adds but ~1 500 panel instructions;
    // profiles/EXPERIMENTS.md, git history)
    // Software pipeline: step k first updates the column class that holds column k + 1, then fetches step k + 1's pivot
    // and multipliers (v_readlane + v_rcp_f64, ds_bpermute: ~150 cycles of latency) and only then the rest of its own
    // updates, which cover that latency (one wave per SIMD at small batches: nothing else would).
    double pinv = fast_rcp(from_lane(Mn[0][0], 0)), colk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) colk[m] = __shfl(Mn[m][0], c, 64);
    static_for<N>([&](auto K) {
      constexpr int k = decltype(K)::value, kc = k & 15, kq = k >> 4, tk = k >> 2;
      constexpr int k1 = k + 1, kc1 = k1 & 15, kq1 = (k1 >> 4) & 3, gk1 = k1 & 3, tk1 = (k1 >> 2) & 15;
      double nf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool piv = m == kq && c == kc;
        dinv[m] = piv ? pinv : dinv[m];
        nf[m] = piv ? 0.0 : -colk[m] * pinv;
      }
      // columns t >= k / 4: the lane's columns g + 4 t > k and up to three already eliminated ones (the pivot row holds
      // rounding-level residues there; those columns are never read again).  The pivot row's own local row index last: its
      // registers are the DPP source of the other three.
      // The reciprocal of the next pivot is a dependent chain (v_rcp_f64, two Newton multiply-adds: ~60 cycles of an in-order
      // wave doing nothing else); its three instructions are spread between the first column classes of the remaining updates.
      double praw = 0.0, px = 0.0, pe = 0.0;
      if constexpr (k1 < N) {
        d8_update<kc, kq, tk1>(Mn, nf);
        __builtin_amdgcn_sched_barrier(0);
        praw = from_lane(Mn[kq1][tk1], 16 * gk1 + kc1);
#pragma unroll
        for (int m = 0; m < 4; ++m) colk[m] = __shfl(Mn[m][tk1], 16 * gk1 + c, 64);
        px = __builtin_amdgcn_rcp(praw);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<16>([&](auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int idx = t == tk ? 0 : t - tk - 1;          // position among the classes still to update
        if constexpr (t >= tk && !(k1 < N && t == tk1)) {
          if constexpr (k1 < N && idx == 1) {
            __builtin_amdgcn_sched_barrier(0);
            pe = dfma(-praw, px, 1.0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (k1 < N && idx == 2) {
            __builtin_amdgcn_sched_barrier(0);
            pinv = dfma(pe, px, px);
            __builtin_amdgcn_sched_barrier(0);
          }
          d8_update<kc, kq, t>(Mn, nf);
        }
      });
      if constexpr (k1 < N) {
        // (late steps: fewer than three classes were left, finish the chain here)
        constexpr int n_left = 15 - tk;      // classes tk .. 15 without tk1
        if constexpr (n_left < 2) pe = dfma(-praw, px, 1.0);
        if constexpr (n_left < 3) pinv = dfma(pe, px, px);
      }
      if (k == N - 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) yv[m] = nf[m];
      }
    });
    tick(3);
    // coordinate a = c + 16 m: the four row groups hold the same values; row group 0 hands them out through LDS
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int a = c + 16 * m;
        sT[a >> 3][a & 7] = (a == N - 1 ? 1.0 : yv[m]) * dinv[m];
      }
    }
    __builtin_amdgcn_wave_barrier();
    x = sT[i][ip];
    // singular to rounding (a fixed point that is not unique, see DirectD4::kMaxInversePivot): let the power iteration of
    // the energy kernel decide, from its default start
    const bool tiny = fabs(dinv[0]) > 1e10 || fabs(dinv[1]) > 1e10 || fabs(dinv[2]) > 1e10 || fabs(dinv[3]) > 1e10;
    if (__any(tiny)) x = __builtin_nan("");
  }
  // coordinates -> complex r[i][i']: the transposed coordinate comes through LDS; trace 1; a non-finite solve -> 1/8
  __builtin_amdgcn_wave_barrier();
  sT[i][ip] = x;
  __builtin_amdgcn_wave_barrier();
  const double xt = sT[ip][i];
  const double tr = wave_sum(i == ip ? x : 0.0);
  const bool good = fabs(tr) > 1e-300 && fabs(tr) < 1e300;      // wave-uniform; NaN fails
  const double inv = good ? 1.0 / tr : 0.0;
  double re = i <= ip ? x : xt, im = i == ip ? 0.0 : (i < ip ? xt : -x);
  re *= inv;
  im *= inv;
  const bool fin = __all(fabs(re) < 1e300 && fabs(im) < 1e300) && good;
  if (!fin) {
    re = i == ip ? 1.0 / D : 0.0;
    im = 0.0;
  }
  tick(4);
  return make_double2(re, im);
}

// One wave, lane = 8 i + i'.  sA: the evaluation's tensor A_s[i][j] (padded rows of 9), sT / sM: scratch.  Returns r[i][i'] of
// the lane (trace 1); a non-finite or ill-conditioned solve returns the default start 1/8 (the caller's power iteration decides).
__device__ __forceinline__ double2 env_direct_d8_solve(const double2 (*sA)[8][9], double (*sT)[9], double (*sM)[17], int lane,
                                                        long long* prof = nullptr) {     // prof: phase clocks, scratch builds only
  constexpr int D = 8, N = 64;
  auto tick = [&](int k) {
    if (prof) {
      __builtin_amdgcn_sched_barrier(0);
      prof[k] = wall_clock64();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  tick(0);
  const int i = lane >> 3, ip = lane & 7;
  double M[N];
  {
    // row (i, i') of the real transfer matrix: P(j,j') = sum_s (gamma A_s[i][j]) conj(A_s[i'][j']), gamma = 1 (i <= i') | i (i > i')
    const bool rot = i > ip;
    double tr[2][D], ti[2][D], br[2][D], bi[2][D];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double2 a = sA[s][i][j], c = sA[s][ip][j];
        tr[s][j] = rot ? -a.y : a.x;
        ti[s][j] = rot ? a.x : a.y;
        br[s][j] = c.x;
        bi[s][j] = c.y;
      }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double v = tr[0][j] * br[0][j];
      v = dfma(ti[0][j], bi[0][j], v);
      v = dfma(tr[1][j], br[1][j], v);
      v = dfma(ti[1][j], bi[1][j], v);
      M[9 * j] = v;
    }
#pragma unroll
    for (int lo = 0; lo < D; ++lo)
#pragma unroll
      for (int hi = lo + 1; hi < D; ++hi) {
        double re = tr[0][lo] * br[0][hi], im = tr[0][lo] * bi[0][hi];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s > 0) {
            re = dfma(tr[s][lo], br[s][hi], re);
            im = dfma(tr[s][lo], bi[s][hi], im);
          }
          re = dfma(ti[s][lo], bi[s][hi], re);
          re = dfma(tr[s][hi], br[s][lo], re);
          re = dfma(ti[s][hi], bi[s][lo], re);
          im = dfma(-ti[s][lo], br[s][hi], im);
          im = dfma(ti[s][hi], br[s][lo], im);
          im = dfma(-tr[s][hi], bi[s][lo], im);
        }
        M[8 * lo + hi] = re;
        M[8 * hi + lo] = im;
      }
    // + trace functional on the last row; the identity (column = the lane's own index: a register index that depends on the
    // lane - 64 compare / select / subtract triples here) is subtracted in LDS during the layout change below
    const double w63 = lane == N - 1 ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) M[9 * j] += w63;
  }
  __builtin_amdgcn_sched_barrier(0);
  tick(1);
  // ---- elimination in a 2-D cyclic layout: lane (g, c) = (lane >> 4, lane & 15) holds rows c + 16 m (m < 4) x columns
  // g + 4 t (t < 16).  Step k: the pivot row's entries of the lane's column class sit in lane k % 16 OF THE SAME
  // 16-LANE DPP ROW, so the update is ONE instruction per entry - v_fmac_f64_dpp row_newbcast (gfx90a+ 64-bit DPP, full
  // FMA rate measured: profiles/experiments/scratch/dpp64_probe.hip) - with no separate broadcast (the row-per-lane layout spends two
  // v_readlane_b32 per FMA).  The multipliers (column k of the lane's four rows) come from the same c in row group
  // k % 4 (ds_bpermute), and are the same in all four row groups.  ~3 800 instead of ~7 200 instructions per evaluation.
  double Mn[4][16];
  {
    // row-per-lane -> cyclic layout through LDS, a quarter of the columns at a time (8.5 KB)
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    // (one pass through a 33 KB image instead of four: no faster - the 128 LDS instructions take ~1.1 us whatever the row
    // padding, 1 or 2 doubles: the reads are 2- to 4-way bank conflicts in either; paddings of 4 and 8: 1.9 and 3.7 us)
    for (int qt = 0; qt < 4; ++qt) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) sM[lane][jj] = M[16 * qt + jj];
      __builtin_amdgcn_wave_barrier();
      if ((lane >> 4) == qt) sM[lane][lane & 15] -= 1.0;      // - identity: the lane's own diagonal entry sits in this quarter
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Mn[m][4 * qt + tt] = sM[c + 16 * m][g + 4 * tt];
    }
  }
  return env_direct_d8_eliminate(Mn, sT, lane, prof);
}

}  // namespace qmps
