// qmps_kernels.hip - hand-written CDNA4 (gfx950) kernels for qmps's classical inner loop.
//
// Hot path (per evaluation; reference = fergusfinn/qmps, cited file:line):
//   A[2][D][D]  --(power iteration on r -> sum_s A_s r A_s^+ ; replaces the xmps eigen-solve behind
//                  get_env_exact, qmps/tools.py:176-182; `krylov`, Power Method.ipynb cells 5-6)-->  r
//   (A, r, h)   --(closed form of State + psi^+ (1 x h x 1) psi, qmps/represent.py:258-262,
//                  qmps/ground_state.py:159-167)-->  E
//
// Mapping (fp64 complex arithmetic, VALU-bound at D <= 8, MFMA at D = 16):
//   D = 2, 4 : ONE EVALUATION PER LANE.  The wave's 64 tensors are read from HBM as one
//              contiguous, fully coalesced 64 x 32 D^2-byte slab (16 B per lane per
//              instruction), transposed through a padded LDS tile, and from then on every
//              operand of every v_fma_f64 is a VGPR of the lane that needs it: no cross-lane
//              traffic, no LDS traffic, no barriers inside the power loop.  r is kept as a
//              packed Hermitian matrix (upper triangle), so one power step costs 12 D^3 - 2 D^2
//              FMAs instead of 16 D^3.
//   D = 8,16 : one evaluation per workgroup of D x D threads, A / r / X tiles in LDS
//              (first correct version; the tuned D = 16 path uses v_mfma_f64_16x16x4_f64).
//
// Power iteration (identical in oracle/qmps_oracle.c and oracle/qmps_oracle.py):
//   r_0 = 1/D (or the caller's warm start);  r' = herm(sum_s A_s r A_s^+);  r' /= tr r';
//   stop when ||r' - r||_F^2 < tol^2;  status 0 converged / 1 hit max_iter / 2 r not PD.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"

namespace qmps {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Packed Hermitian accessors: only entries with j >= i are stored; the diagonal is real.
template <int D>
__device__ __forceinline__ double h_re(const double (&re)[D][D], int i, int j) {
  return i <= j ? re[i][j] : re[j][i];
}
template <int D>
__device__ __forceinline__ double h_im(const double (&im)[D][D], int i, int j) {
  return i < j ? im[i][j] : -im[j][i];  // never called with i == j
}

// One power step, one evaluation per lane:  n = sum_s A_s r A_s^+  (upper triangle only).
template <int D>
__device__ __forceinline__ void power_step(const double (&are)[2][D][D], const double (&aim)[2][D][D],
                                           const double (&rre)[D][D], const double (&rim)[D][D],
                                           double (&nre)[D][D], double (&nim)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      nre[i][j] = 0.0;
      nim[i][j] = 0.0;
    }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      // row i of X_s = A_s r
      double xre[D], xim[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double xr = 0.0, xi = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double ar = are[s][i][k], ai = aim[s][i][k];
          const double rr = h_re<D>(rre, k, j);
          xr = dfma(ar, rr, xr);
          xi = dfma(ai, rr, xi);
          if (k != j) {
            const double ri = h_im<D>(rim, k, j);
            xr = dfma(-ai, ri, xr);
            xi = dfma(ar, ri, xi);
          }
        }
        xre[j] = xr;
        xim[j] = xi;
      }
      // n[i][j] += sum_k X[i][k] conj(A_s[j][k]),  j >= i
#pragma unroll
      for (int j = i; j < D; ++j) {
        double nr = nre[i][j], ni = nim[i][j];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          nr = dfma(xre[k], are[s][j][k], nr);
          nr = dfma(xim[k], aim[s][j][k], nr);
          if (j > i) {
            ni = dfma(xim[k], are[s][j][k], ni);
            ni = dfma(-xre[k], aim[s][j][k], ni);
          }
        }
        nre[i][j] = nr;
        nim[i][j] = ni;
      }
    }
  }
}

// Cholesky positive-definiteness test of a packed Hermitian matrix (LAPACK zpotrf criterion:
// a pivot that is not > 0 fails).  Mirrors cholesky(r) at qmps/tools.py:182.
template <int D>
__device__ __forceinline__ bool is_positive_definite(const double (&rre)[D][D], const double (&rim)[D][D]) {
  double lre[D][D], lim[D][D];  // lower factor, L[i][j], j <= i
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double d = rre[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= lre[j][k] * lre[j][k] + lim[j][k] * lim[j][k];
    ok = ok && (d > 0.0);
    const double ljj = __builtin_sqrt(d > 0.0 ? d : 1.0);
    const double inv = 1.0 / ljj;
    lre[j][j] = ljj;
    lim[j][j] = 0.0;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      // r[i][j] with i > j  = conj(r[j][i])
      double cr = rre[j][i], ci = -rim[j][i];
#pragma unroll
      for (int k = 0; k < j; ++k) {
        // L[i][k] * conj(L[j][k])
        cr -= lre[i][k] * lre[j][k] + lim[i][k] * lim[j][k];
        ci -= lim[i][k] * lre[j][k] - lre[i][k] * lim[j][k];
      }
      lre[i][j] = cr * inv;
      lim[i][j] = ci * inv;
    }
  }
  return ok;
}

// Two-site reduced density matrix, upper triangle (tau <= sigma), one evaluation per lane:
//   rho[tau][sigma] = tr(B_tau r B_sigma^+),  B_{2 s1 + s2} = A_s1 A_s2   (NOT yet divided by tr r)
// computed as  X_t2 = A_t2 r ;  R = X_t2 A_s2^+ ;  Z = A_t1 R ;  rho = sum_ik Z[i][k] conj(A_s1[i][k]).
// Generalised to a two-site unit cell: left-site tensor L (are/aim) and right-site tensor Rt (bre/bim):
//   rho[(t1 t2)][(s1 s2)] = tr(L_t1 Rt_t2 r Rt_s2^+ L_s1^+);  single-site cell: L == Rt.
template <int D>
__device__ __forceinline__ void two_site_rdm(const double (&are)[2][D][D], const double (&aim)[2][D][D],
                                             const double (&bre)[2][D][D], const double (&bim)[2][D][D],
                                             const double (&rre)[D][D], const double (&rim)[D][D],
                                             double (&pre)[4][4], double (&pim)[4][4]) {
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    // X = A_t2 r (full)
    double xre[D][D], xim[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double xr = 0.0, xi = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double ar = bre[t2][i][k], ai = bim[t2][i][k];
          const double rr = h_re<D>(rre, k, j);
          xr = dfma(ar, rr, xr);
          xi = dfma(ai, rr, xi);
          if (k != j) {
            const double ri = h_im<D>(rim, k, j);
            xr = dfma(-ai, ri, xr);
            xi = dfma(ar, ri, xi);
          }
        }
        xre[i][j] = xr;
        xim[i][j] = xi;
      }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // R = X A_s2^+ (full)
      double Rre[D][D], Rim[D][D];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          double cr = 0.0, ci = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            cr = dfma(xre[i][k], bre[s2][j][k], cr);
            cr = dfma(xim[i][k], bim[s2][j][k], cr);
            ci = dfma(xim[i][k], bre[s2][j][k], ci);
            ci = dfma(-xre[i][k], bim[s2][j][k], ci);
          }
          Rre[i][j] = cr;
          Rim[i][j] = ci;
        }
#pragma unroll
      for (int t1 = 0; t1 < 2; ++t1) {
        const int tau = 2 * t1 + t2;
        // is any (s1) with tau <= sigma ?
        if (tau > 2 + s2) continue;
        // rho[tau][sigma] = sum_i sum_k Z[i][k] conj(A_s1[i][k]),  Z = A_t1 R, one row at a time
        double acc_re[2] = {0.0, 0.0}, acc_im[2] = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
          for (int k = 0; k < D; ++k) {
            double zr = 0.0, zi = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
              const double ar = are[t1][i][j], ai = aim[t1][i][j];
              zr = dfma(ar, Rre[j][k], zr);
              zr = dfma(-ai, Rim[j][k], zr);
              zi = dfma(ar, Rim[j][k], zi);
              zi = dfma(ai, Rre[j][k], zi);
            }
#pragma unroll
            for (int s1 = 0; s1 < 2; ++s1) {
              const int sigma = 2 * s1 + s2;
              if (tau <= sigma) {
                acc_re[s1] = dfma(zr, are[s1][i][k], acc_re[s1]);
                acc_re[s1] = dfma(zi, aim[s1][i][k], acc_re[s1]);
                if (tau < sigma) {
                  acc_im[s1] = dfma(zi, are[s1][i][k], acc_im[s1]);
                  acc_im[s1] = dfma(-zr, aim[s1][i][k], acc_im[s1]);
                }
              }
            }
          }
        }
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1) {
          const int sigma = 2 * s1 + s2;
          if (tau <= sigma) {
            pre[tau][sigma] = acc_re[s1];
            pim[tau][sigma] = acc_im[s1];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 1: D = 2, 4 - one evaluation per lane, one wave per workgroup.
// ------------------------------------------------------------------------------------------
template <int D>
struct LaneCfg {
  static constexpr int kRowBytes = 32 * D * D;          // one tensor A[2][D][D] complex128
  static constexpr int kRowPad = kRowBytes + 16;        // +16 B: conflict-free ds_read_b128 by row
  static constexpr int kLdsBytes = 64 * kRowPad;        // one wave's slab
  static constexpr int kChunks = kRowBytes / 16;        // 16-B pieces per tensor == loads per lane
};

__device__ __forceinline__ double rdm_energy(const double2* h, const double (&pre)[4][4], const double (&pim)[4][4]) {
  double e = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double2 hv = h[s * 4 + t];
      const double rr = (t <= s) ? pre[t][s] : pre[s][t];
      e = dfma(hv.x, rr, e);
      if (t != s) {
        const double ri = (t < s) ? pim[t][s] : -pim[s][t];
        e = dfma(-hv.y, ri, e);
      }
    }
  return e;
}

// normalise a freshly computed power step in place and return ||n - r||_F^2
template <int D>
__device__ __forceinline__ double normalise_and_diff(double (&nre)[D][D], double (&nim)[D][D],
                                                     const double (&rre)[D][D], const double (&rim)[D][D]) {
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) tr += nre[i][i];
  const double inv = 1.0 / tr;
  double dd = 0.0, od = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      nre[i][j] *= inv;
      const double dr = nre[i][j] - rre[i][j];
      if (i == j) {
        nim[i][j] = 0.0;
        dd = dfma(dr, dr, dd);
      } else {
        nim[i][j] *= inv;
        const double di = nim[i][j] - rim[i][j];
        od = dfma(dr, dr, od);
        od = dfma(di, di, od);
      }
    }
  return dfma(2.0, od, dd);
}

// D = 2 only: repeated-squaring tail, one evaluation per lane.  P = T^(2^m) as a 4 x 4 complex matrix in
// registers (index (i,i') -> 2 i + i'), r_m = herm(P vec(r_C))/tr, stop at ||r_m - r_{m-1}||_F^2 < tol^2.
__device__ __forceinline__ void squaring_tail_d2(const double (&are)[2][2][2], const double (&aim)[2][2][2],
                                                 double (&rre)[2][2], double (&rim)[2][2], bool& active, int& iters,
                                                 int& status, int done, int max_iter, double tol2) {
  double pr[4][4], pi[4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ip = 0; ip < 2; ++ip)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          double er = 0.0, ei = 0.0;
#pragma unroll
          for (int s = 0; s < 2; ++s) {  // A_s[i][j] * conj(A_s[ip][jp])
            er = dfma(are[s][i][j], are[s][ip][jp], er);
            er = dfma(aim[s][i][j], aim[s][ip][jp], er);
            ei = dfma(aim[s][i][j], are[s][ip][jp], ei);
            ei = dfma(-are[s][i][j], aim[s][ip][jp], ei);
          }
          pr[2 * i + ip][2 * j + jp] = er;
          pi[2 * i + ip][2 * j + jp] = ei;
        }
  // vec(r_C): full Hermitian matrix, index 2 i + i'
  const double vr[4] = {rre[0][0], rre[0][1], rre[0][1], rre[1][1]};
  const double vi[4] = {0.0, rim[0][1], -rim[0][1], 0.0};
  int m = 0;
  while (done + (1 << (m + 1)) <= max_iter && m < 29) {
    if (!__any(active)) break;
    double qr[4][4], qi[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double xr = 0.0, xi = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xr = dfma(pr[a][k], pr[k][c], xr);
          xr = dfma(-pi[a][k], pi[k][c], xr);
          xi = dfma(pr[a][k], pi[k][c], xi);
          xi = dfma(pi[a][k], pr[k][c], xi);
        }
        qr[a][c] = xr;
        qi[a][c] = xi;
      }
    ++m;
    double yr[4], yi[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pr[a][k] = qr[a][k];
        pi[a][k] = qi[a][k];
        xr = dfma(qr[a][k], vr[k], xr);
        xr = dfma(-qi[a][k], vi[k], xr);
        xi = dfma(qr[a][k], vi[k], xi);
        xi = dfma(qi[a][k], vr[k], xi);
      }
      yr[a] = xr;
      yi[a] = xi;
    }
    // hermitise (entries 1 = (0,1), 2 = (1,0)), trace-normalise
    double nre[2][2], nim[2][2];
    nre[0][0] = yr[0];
    nre[1][1] = yr[3];
    nre[0][1] = 0.5 * (yr[1] + yr[2]);
    nim[0][1] = 0.5 * (yi[1] - yi[2]);
    nim[0][0] = nim[1][1] = 0.0;
    const double d2 = normalise_and_diff<2>(nre, nim, rre, rim);
    if (active) {
      rre[0][0] = nre[0][0]; rre[0][1] = nre[0][1]; rre[1][1] = nre[1][1]; rim[0][1] = nim[0][1];
      iters = done + (1 << m);
      if (d2 < tol2) {
        active = false;
        status = QMPS_ST_OK;
      }
    }
  }
}

template <int D, bool SOLVE>
__global__ __launch_bounds__(64) void energy_lane_kernel(LaneArgs p) {
  using Cfg = LaneCfg<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x;
  const int64_t wave_first = (int64_t)blockIdx.x * 64;
  int64_t b = wave_first + lane;
  bool valid = b < p.B;
  double are[2][D][D], aim[2][D][D];

  if (p.idx_list != nullptr) {
    // ---- list mode: evaluation ids come from a device-side worklist (gathered loads, few items)
    const int64_t n_list = *p.idx_count;
    if (wave_first >= n_list) return;
    valid = wave_first + lane < n_list;
    b = valid ? (int64_t)p.idx_list[wave_first + lane] : (int64_t)p.idx_list[wave_first];
    const double2* a = (const double2*)p.A + b * (2 * D * D);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const double2 v = a[(s * D + i) * D + j];
          are[s][i][j] = v.x;
          aim[s][i][j] = v.y;
        }
  } else {
    // ---- HBM -> LDS: the wave's 64 tensors are one contiguous slab; 16 B per lane per load
    {
      const unsigned char* slab = (const unsigned char*)p.A + wave_first * Cfg::kRowBytes;
      const int64_t slab_bytes = (p.B - wave_first < 64 ? p.B - wave_first : 64) * (int64_t)Cfg::kRowBytes;
#pragma unroll
      for (int c = 0; c < Cfg::kChunks; ++c) {
        const int off = c * 1024 + lane * 16;
        double2 v = make_double2(0.0, 0.0);
        if (off < slab_bytes) v = *(const double2*)(slab + off);
        const int e = off / Cfg::kRowBytes, w = off % Cfg::kRowBytes;
        *(double2*)(lds + e * Cfg::kRowPad + w) = v;
      }
    }
    __syncthreads();
    // ---- LDS -> VGPR: each lane takes its own tensor
    const unsigned char* row = lds + lane * Cfg::kRowPad;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const double2 v = *(const double2*)(row + ((s * D + i) * D + j) * 16);
          are[s][i][j] = v.x;
          aim[s][i][j] = v.y;
        }
  }

  // ---- environment: r0 = 1/D or the caller's guess (packed Hermitian, trace-normalised)
  double rre[D][D], rim[D][D];
  if (p.r_in != nullptr && valid) {
    const double2* g = (const double2*)p.r_in + b * (D * D);
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        const double2 u = g[i * D + j], l = g[j * D + i];
        rre[i][j] = 0.5 * (u.x + l.x);
        rim[i][j] = (i == j) ? 0.0 : 0.5 * (u.y - l.y);
        if (i == j) tr += rre[i][j];
      }
    const double inv = 1.0 / tr;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        rre[i][j] *= inv;
        rim[i][j] *= inv;
      }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        rre[i][j] = (i == j) ? 1.0 / D : 0.0;
        rim[i][j] = 0.0;
      }
  }

  int iters = 0, status = QMPS_ST_OK;
  bool handed_off = false;
  if (SOLVE) {
    status = QMPS_ST_NOT_CONVERGED;
    bool active = valid;
    const double tol2 = p.tol * p.tol;
    const bool hybrid = p.handoff > 0 && p.handoff < p.max_iter;
    const int plain = hybrid ? p.handoff : p.max_iter;
    // two steps per trip, ping-pong r -> n -> r: frozen (converged) lanes are simply masked off
    for (int k = 1; k <= plain; k += 2) {
      if (!__any(active)) break;
      double nre[D][D], nim[D][D];
      if (active) {
        power_step<D>(are, aim, rre, rim, nre, nim);
        const double d2 = normalise_and_diff<D>(nre, nim, rre, rim);
        iters = k;
        if (d2 < tol2 || k == plain) {
          if (d2 < tol2) status = QMPS_ST_OK;
          active = false;
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) {
              rre[i][j] = nre[i][j];
              rim[i][j] = nim[i][j];
            }
        }
      }
      if (active) {
        power_step<D>(are, aim, nre, nim, rre, rim);
        const double d2 = normalise_and_diff<D>(rre, rim, nre, nim);
        iters = k + 1;
        if (d2 < tol2) {
          status = QMPS_ST_OK;
          active = false;
        }
      }
    }
    if (hybrid) {
      active = valid && status == QMPS_ST_NOT_CONVERGED;
      if (D == 2) {
        if constexpr (D == 2) squaring_tail_d2(are, aim, rre, rim, active, iters, status, plain, p.max_iter, tol2);
      } else if (p.work_idx != nullptr) {
        // hand the slow items to the wave-per-item squaring kernel: wave-aggregated append
        const unsigned long long mask = __ballot(active);
        if (mask != 0ull) {
          int base = 0;
          if (lane == 0) base = atomicAdd(p.work_count, __popcll(mask));
          base = __shfl(base, 0, 64);
          if (active) {
            p.work_idx[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)b;
            handed_off = true;
          }
        }
      }
    }
    if (status == QMPS_ST_OK && !is_positive_definite<D>(rre, rim)) status = QMPS_ST_NOT_PD;
  } else if (p.check_pd) {
    if (valid) {
      status = p.status[b];
      if (status == QMPS_ST_OK && !is_positive_definite<D>(rre, rim)) status = QMPS_ST_NOT_PD;
    }
  }

  // ---- energy epilogue: rho (upper triangle) -> E_t = Re sum h_t[s][t] rho[t][s] / tr r
  double pre[4][4], pim[4][4];
  two_site_rdm<D>(are, aim, are, aim, rre, rim, pre, pim);
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) tr += rre[i][i];
  const double inv = 1.0 / tr;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = t; s < 4; ++s) {
      pre[t][s] *= inv;
      pim[t][s] = (t == s) ? 0.0 : pim[t][s] * inv;
    }
  if (!valid) return;

  for (int q = 0; q < p.n_terms; ++q) {
    const double2* h = (const double2*)p.h + q * 16;  // wave-uniform -> scalar loads
    p.E[b * p.n_terms + q] = rdm_energy(h, pre, pim);
  }
  if (SOLVE) {
    p.iters[b] = iters;
    p.status[b] = status;
  } else if (p.check_pd) {
    p.status[b] = status;
  }
  (void)handed_off;
  if (p.r_out != nullptr && SOLVE) {
    double2* o = (double2*)p.r_out + b * (D * D);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double re = h_re<D>(rre, i, j);
        const double im = (i == j) ? 0.0 : h_im<D>(rim, i, j);
        o[i * D + j] = make_double2(re, im);
      }
  }
  if (p.rho_out != nullptr) {
    double2* o = (double2*)p.rho_out + b * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double re = (t <= s) ? pre[t][s] : pre[s][t];
        const double im = (t == s) ? 0.0 : ((t < s) ? pim[t][s] : -pim[s][t]);
        o[t * 4 + s] = make_double2(re, im);
      }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 1b: two-site unit cell (qmps/ground_state.py:291-331, NonSparseFullTwoSiteEnergyOptimizer),
// one evaluation per lane.  Inputs are the two state UNITARIES U1, U2 [B][2D][2D]; the kernel applies
// unitary_to_tensor on load.  r12 = fixed point of r -> T_A1(T_A2(r)) (transfer map of
// merge(A1, A2), qmps/time_evolve_tools.py:20-23); r21 = T_A2(r12)/tr is the fixed point of the
// swapped cell, so ONE power iteration serves both energies:
//   E1 = sum h[s][t] tr(A1_t1 A2_t2 r12 A2_s2^+ A1_s1^+),  E2 = same with 1 <-> 2 and r21,  f = (E1+E2)/2.
// ------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void normalise_herm(double (&nre)[D][D], double (&nim)[D][D]) {
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) tr += nre[i][i];
  const double inv = 1.0 / tr;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      nre[i][j] *= inv;
      nim[i][j] = (i == j) ? 0.0 : nim[i][j] * inv;
    }
}

template <int D>
__global__ __launch_bounds__(64) void cell2_lane_kernel(Cell2Args p) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  double a1re[2][D][D], a1im[2][D][D], a2re[2][D][D], a2im[2][D][D];
  {
    const double2* u1 = (const double2*)p.U1 + b * (4 * D * D);
    const double2* u2 = (const double2*)p.U2 + b * (4 * D * D);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const double2 v1 = u1[(2 * i + s) * (2 * D) + j], v2 = u2[(2 * i + s) * (2 * D) + j];
          a1re[s][i][j] = v1.x; a1im[s][i][j] = v1.y;
          a2re[s][i][j] = v2.x; a2im[s][i][j] = v2.y;
        }
  }
  double rre[D][D], rim[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      rre[i][j] = (i == j) ? 1.0 / D : 0.0;
      rim[i][j] = 0.0;
    }
  int iters = 0, status = QMPS_ST_NOT_CONVERGED;
  const double tol2 = p.tol * p.tol;
  for (int k = 1; k <= p.max_iter; ++k) {
    double tre[D][D], tim[D][D], nre[D][D], nim[D][D];
    power_step<D>(a2re, a2im, rre, rim, tre, tim);
    power_step<D>(a1re, a1im, tre, tim, nre, nim);
    normalise_herm<D>(nre, nim);
    double d2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) {
        const double dr = nre[i][j] - rre[i][j], di = nim[i][j] - rim[i][j];
        d2 += (i == j) ? dr * dr : 2.0 * (dr * dr + di * di);
        rre[i][j] = nre[i][j];
        rim[i][j] = nim[i][j];
      }
    iters = k;
    if (d2 < tol2) { status = QMPS_ST_OK; break; }
  }
  // environment of the swapped cell
  double qre[D][D], qim[D][D];
  power_step<D>(a2re, a2im, rre, rim, qre, qim);
  normalise_herm<D>(qre, qim);
  if (status == QMPS_ST_OK && !(is_positive_definite<D>(rre, rim) && is_positive_definite<D>(qre, qim)))
    status = QMPS_ST_NOT_PD;
  double p1re[4][4], p1im[4][4], p2re[4][4], p2im[4][4];
  two_site_rdm<D>(a1re, a1im, a2re, a2im, rre, rim, p1re, p1im);
  two_site_rdm<D>(a2re, a2im, a1re, a1im, qre, qim, p2re, p2im);
  double tr1 = 0.0, tr2 = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) { tr1 += rre[i][i]; tr2 += qre[i][i]; }
  for (int q = 0; q < p.n_terms; ++q) {
    const double2* h = (const double2*)p.h + q * 16;
    const double e1 = rdm_energy(h, p1re, p1im) / tr1;
    const double e2 = rdm_energy(h, p2re, p2im) / tr2;
    p.E[b * p.n_terms + q] = 0.5 * (e1 + e2);
    if (p.E12 != nullptr) {
      p.E12[(b * p.n_terms + q) * 2 + 0] = e1;
      p.E12[(b * p.n_terms + q) * 2 + 1] = e2;
    }
  }
  p.iters[b] = iters;
  p.status[b] = status;
}

// ------------------------------------------------------------------------------------------
// Kernel 1c: D = 4 repeated-squaring tail, ONE WAVE PER ITEM, v_mfma_f64_16x16x4_f64.
// The transfer matrix of a D = 4 tensor is exactly one 16 x 16 complex MFMA tile:
//   E[(i,i'),(j,j')] = sum_s A_s[i][j] conj(A_s[i'][j']).
// P_m = E^(2^m) by squaring (4 real 16x16x16 products = 16 MFMAs per complex squaring); the
// accumulator layout (row = 4 reg + lane/16, col = lane%16) is the B-operand layout of the next
// product, the A-operand layout (row = lane%16, k = 4 kk + lane/16) comes from a padded LDS image.
// r_m = herm(P_m vec(r_C))/tr;  stop at ||r_m - r_{m-1}||_F^2 < tol^2;  iterations = done + 2^m.
// ------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void env_square_d4_kernel(SquareArgs p) {
  constexpr int D = 4, N = 16, LD = 17;  // LD: padded leading dimension (complex) of the LDS image
  __shared__ double2 sA[2 * N];           // the tensor
  __shared__ double2 sP[N * LD];          // P, row-major
  __shared__ double2 sY[N];               // P vec(r)
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int n_items = *p.work_count;
  const double tol2 = p.tol * p.tol;
  for (int w = blockIdx.x; w < n_items; w += gridDim.x) {
    const int64_t b = p.work_idx[w];
    __syncthreads();
    if (lane < 2 * N) sA[lane] = ((const double2*)p.A)[b * (2 * N) + lane];
    // v = vec(r_C): entry (j,j') = c
    const double2 v = ((const double2*)p.r)[b * N + c];
    // previous iterate, replicated in every lane (16 complex)
    double2 rprev[N];
#pragma unroll
    for (int e = 0; e < N; ++e) rprev[e] = ((const double2*)p.r)[b * N + e];
    __syncthreads();
    // E in accumulator layout: row = 4 reg + g -> (i, i') = (reg, g); col = c -> (j, j') = (c>>2, c&3)
    v4f64 Pre, Pim;
    {
      const int j = c >> 2, jp = c & 3;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double er = 0.0, ei = 0.0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const double2 x = sA[(s * D + reg) * D + j], y = sA[(s * D + g) * D + jp];
          er = dfma(x.x, y.x, er);
          er = dfma(x.y, y.y, er);
          ei = dfma(x.y, y.x, ei);
          ei = dfma(-x.x, y.y, ei);
        }
        Pre[reg] = er;
        Pim[reg] = ei;
      }
    }
    int m = 0, iters = p.done, status = QMPS_ST_NOT_CONVERGED;
    while (p.done + (1 << (m + 1)) <= p.max_iter && m < 29) {
      // LDS image of P for the A-operand fragments
      __syncthreads();
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sP[(4 * reg + g) * LD + c] = make_double2(Pre[reg], Pim[reg]);
      __syncthreads();
      v4f64 cre0 = {0, 0, 0, 0}, cre1 = {0, 0, 0, 0}, cim0 = {0, 0, 0, 0}, cim1 = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 a = sP[c * LD + 4 * kk + g];          // A[row = c][k = 4 kk + g]
        const double bre = Pre[kk], bim = Pim[kk];          // B[k = 4 kk + g][col = c]
        cre0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bre, cre0, 0, 0, 0);
        cre1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, bim, cre1, 0, 0, 0);
        cim0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bim, cim0, 0, 0, 0);
        cim1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bre, cim1, 0, 0, 0);
      }
      Pre = cre0 + cre1;
      Pim = cim0 + cim1;
      ++m;
      // y = P v : per-lane products, then a sum over the 16 lanes of the row group
      double yr[4], yi[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double tr_ = Pre[reg] * v.x - Pim[reg] * v.y;
        double ti_ = Pre[reg] * v.y + Pim[reg] * v.x;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
          tr_ += __shfl_xor(tr_, off, 64);
          ti_ += __shfl_xor(ti_, off, 64);
        }
        yr[reg] = tr_;
        yi[reg] = ti_;
      }
      __syncthreads();
      if (c == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sY[4 * reg + g] = make_double2(yr[reg], yi[reg]);  // entry (i,i') = (reg,g)
      }
      __syncthreads();
      // every lane: hermitise, normalise, compare (16 entries)
      double2 rn[N];
      double tr = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) tr += sY[i * D + i].x;
      const double inv = 1.0 / tr;
      double d2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const double2 u = sY[i * D + j], l = sY[j * D + i];
          const double re = 0.5 * (u.x + l.x) * inv;
          const double im = (i == j) ? 0.0 : 0.5 * (u.y - l.y) * inv;
          rn[i * D + j] = make_double2(re, im);
          const double dr = re - rprev[i * D + j].x, di = im - rprev[i * D + j].y;
          d2 = dfma(dr, dr, d2);
          d2 = dfma(di, di, d2);
        }
#pragma unroll
      for (int e = 0; e < N; ++e) rprev[e] = rn[e];
      iters = p.done + (1 << m);
      if (d2 < tol2) {
        status = QMPS_ST_OK;
        break;
      }
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < N; ++e) sY[e] = rprev[e];
    }
    __syncthreads();
    if (lane < N) ((double2*)p.r)[b * N + lane] = sY[lane];
    if (lane == 0) {
      p.iters[b] = iters;
      p.status[b] = status;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 2: D = 8, 16 - one evaluation per workgroup of D x D threads, tiles in LDS.
// Thread (i, j) owns r[i][j].  First correct version of the large-D path.
// ------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  // sum over the D*D threads of the workgroup; result broadcast to every thread
  constexpr int N = D * D;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
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

template <int D, bool SOLVE>
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

  {
    const double2* a = (const double2*)p.A + b * (2 * N);
    sA[0][i][j] = a[tid];
    sA[1][i][j] = a[N + tid];
  }
  double2 r;
  if (p.r_in != nullptr) {
    const double2* g = (const double2*)p.r_in + b * N;
    const double2 u = g[i * D + j], l = g[j * D + i];
    r = make_double2(0.5 * (u.x + l.x), (i == j) ? 0.0 : 0.5 * (u.y - l.y));
    const double tr = block_sum<D>(i == j ? r.x : 0.0, red, tid);
    r.x /= tr;
    r.y /= tr;
  } else {
    r = make_double2(i == j ? 1.0 / D : 0.0, 0.0);
  }
  sR[i][j] = r;
  __syncthreads();

  auto apply = [&](double2& out) {
    // X_s[i][j] = sum_k A_s[i][k] r[k][j]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 a = sA[s][i][k], rr = sR[k][j];
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
        const double2 x = sX[s][i][k], a = sA[s][j][k];
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
    }
    if (status == QMPS_ST_OK) {
      // Cholesky PD test by thread 0 on the LDS copy (D <= 16: <= 816 complex MACs)
      __shared__ int s_pd;
      if (tid == 0) {
        bool ok = true;
        for (int c = 0; c < D && ok; ++c) {
          double d = sR[c][c].x;
          for (int k = 0; k < c; ++k) d -= sT[1][c][k].x * sT[1][c][k].x + sT[1][c][k].y * sT[1][c][k].y;
          if (!(d > 0.0)) { ok = false; break; }
          const double ljj = __builtin_sqrt(d);
          sT[1][c][c] = make_double2(ljj, 0.0);
          for (int rI = c + 1; rI < D; ++rI) {
            double cr = sR[rI][c].x, ci = sR[rI][c].y;
            for (int k = 0; k < c; ++k) {
              const double2 a = sT[1][rI][k], bb = sT[1][c][k];
              cr -= a.x * bb.x + a.y * bb.y;
              ci -= a.y * bb.x - a.x * bb.y;
            }
            sT[1][rI][c] = make_double2(cr / ljj, ci / ljj);
          }
        }
        s_pd = ok ? 1 : 0;
      }
      __syncthreads();
      if (!s_pd) status = QMPS_ST_NOT_PD;
      __syncthreads();
    }
  }

  // ---- energy: rho[tau][sigma] = tr(A_t1 (A_t2 r A_s2^+) A_s1^+)
  const double trr = block_sum<D>(i == j ? r.x : 0.0, red, tid);
  double2 rho_loc[4][4];
  // X_t2 = A_t2 r  (both t2) -> sX
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    double xr = 0.0, xi = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double2 a = sA[s][i][k], rr = sR[k][j];
      xr = dfma(a.x, rr.x, xr);
      xr = dfma(-a.y, rr.y, xr);
      xi = dfma(a.x, rr.y, xi);
      xi = dfma(a.y, rr.x, xi);
    }
    sX[s][i][j] = make_double2(xr, xi);
  }
  __syncthreads();
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // R[i][j] = sum_k X_t2[i][k] conj(A_s2[j][k]) -> sT[0]
      double cr = 0.0, ci = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 x = sX[t2][i][k], a = sA[s2][j][k];
        cr = dfma(x.x, a.x, cr);
        cr = dfma(x.y, a.y, cr);
        ci = dfma(x.y, a.x, ci);
        ci = dfma(-x.x, a.y, ci);
      }
      __syncthreads();
      sT[0][i][j] = make_double2(cr, ci);
      __syncthreads();
#pragma unroll
      for (int t1 = 0; t1 < 2; ++t1) {
        // Z[i][j] = sum_k A_t1[i][k] R[k][j]
        double zr = 0.0, zi = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double2 a = sA[t1][i][k], rr = sT[0][k][j];
          zr = dfma(a.x, rr.x, zr);
          zr = dfma(-a.y, rr.y, zr);
          zi = dfma(a.x, rr.y, zi);
          zi = dfma(a.y, rr.x, zi);
        }
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1) {
          const double2 a = sA[s1][i][j];
          const double pr = zr * a.x + zi * a.y;
          const double pi = zi * a.x - zr * a.y;
          const double sr = block_sum<D>(pr, red, tid);
          const double si = block_sum<D>(pi, red, tid);
          rho_loc[2 * t1 + t2][2 * s1 + s2] = make_double2(sr / trr, si / trr);
        }
      }
    }
  if (tid == 0) {
    for (int q = 0; q < p.n_terms; ++q) {
      const double2* h = (const double2*)p.h + q * 16;
      double e = 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double2 hv = h[s * 4 + t];
          e += hv.x * rho_loc[t][s].x - hv.y * rho_loc[t][s].y;
        }
      p.E[b * p.n_terms + q] = e;
    }
    if (SOLVE) {
      p.iters[b] = iters;
      p.status[b] = status;
    }
    if (p.rho_out != nullptr) {
      double2* o = (double2*)p.rho_out + b * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) o[t * 4 + s] = rho_loc[t][s];
    }
  }
  if (p.r_out != nullptr && SOLVE) ((double2*)p.r_out)[b * N + tid] = r;
}

// ------------------------------------------------------------------------------------------
// Kernel 3: unitary_to_tensor (qmps/tools.py:151-154):  A[b][s][i][j] = U[b][2 i + s][j], j < D
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unitary_to_tensor_kernel(const double2* __restrict__ U,
                                                                double2* __restrict__ A, int D, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int n = D * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / (2 * n);
    const int e = (int)(t % (2 * n));
    const int s = e / n, i = (e / D) % D, j = e % D;
    A[t] = U[b * (4 * n) + (int64_t)(2 * i + s) * (2 * D) + j];
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 4: cost[t] = sum_b E[b][t]   (rotosolve's M(x) = np.sum(eps(...)), qmps/tools.py:432-433)
// Deterministic two-pass reduction: per-block partials, then one block sums the partials.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_partial_kernel(const double* __restrict__ E, int64_t B, int n_terms,
                                                          double* __restrict__ partial) {
  __shared__ double red[4];
  for (int q = 0; q < n_terms; ++q) {
    double v = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x)
      v += E[b * n_terms + q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)q * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ __launch_bounds__(256) void sum_final_kernel(const double* __restrict__ partial, int n_partial, int n_terms,
                                                        double* __restrict__ cost) {
  __shared__ double red[4];
  for (int q = 0; q < n_terms; ++q) {
    double v = 0.0;
    for (int k = threadIdx.x; k < n_partial; k += blockDim.x) v += partial[(int64_t)q * n_partial + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) cost[q] = red[0] + red[1] + red[2] + red[3];
  }
}

// ------------------------------------------------------------------------------------------
// Probes: FP64 FMA peak and HBM streaming rate, measured on the box the numbers are quoted on.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void probe_fp64_kernel(double* out, int iters) {
  double a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 1.0 + 1e-9 * (threadIdx.x + k);
  const double m = 1.0000001, c = 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = dfma(a[k], m, c);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k];
  if (s == 123.456) out[0] = s;  // keep the chain live without a store in the common case
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst,
                                                         int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) dst[t] = src[t];
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
template <int D>
static hipError_t launch_lane(const LaneArgs& a, bool solve, hipStream_t st) {
  const int grid = (int)((a.B + 63) / 64);
  const size_t lds = LaneCfg<D>::kLdsBytes;
  if (solve)
    hipLaunchKernelGGL((energy_lane_kernel<D, true>), dim3(grid), dim3(64), lds, st, a);
  else
    hipLaunchKernelGGL((energy_lane_kernel<D, false>), dim3(grid), dim3(64), lds, st, a);
  return hipGetLastError();
}

template <int D>
static hipError_t launch_block(const LaneArgs& a, bool solve, hipStream_t st) {
  if (solve)
    hipLaunchKernelGGL((energy_block_kernel<D, true>), dim3((unsigned)a.B), dim3(D * D), 0, st, a);
  else
    hipLaunchKernelGGL((energy_block_kernel<D, false>), dim3((unsigned)a.B), dim3(D * D), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_square_tail(int D, const SquareArgs& a, int grid, hipStream_t st) {
  if (D != 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(env_square_d4_kernel, dim3(grid), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_energy(int D, const LaneArgs& a, bool solve, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  switch (D) {
    case 2: return launch_lane<2>(a, solve, st);
    case 4: return launch_lane<4>(a, solve, st);
    case 8: return launch_block<8>(a, solve, st);
    case 16: return launch_block<16>(a, solve, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_cell2(int D, const Cell2Args& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  if (D != 2) return hipErrorInvalidValue;  // the reference path is D = 2 only (ground_state.py:276)
  hipLaunchKernelGGL((cell2_lane_kernel<2>), dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_unitary_to_tensor(const void* U, void* A, int D, int64_t B, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int64_t total = B * 2 * D * D;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(unitary_to_tensor_kernel, dim3(grid), dim3(256), 0, st, (const double2*)U, (double2*)A, D, total);
  return hipGetLastError();
}

hipError_t launch_sum(const double* E, int64_t B, int n_terms, double* partial, int n_partial, double* cost,
                      hipStream_t st) {
  hipLaunchKernelGGL(sum_partial_kernel, dim3(n_partial), dim3(256), 0, st, E, B, n_terms, partial);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, n_partial, n_terms, cost);
  return hipGetLastError();
}

hipError_t launch_probe_fp64(double* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(probe_fp64_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
  return hipGetLastError();
}

hipError_t launch_probe_copy(const void* src, void* dst, int64_t n16, hipStream_t st) {
  hipLaunchKernelGGL(probe_copy_kernel, dim3(2048), dim3(256), 0, st, (const double2*)src, (double2*)dst, n16);
  return hipGetLastError();
}

}  // namespace qmps
