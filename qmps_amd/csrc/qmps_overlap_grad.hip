// qmps_overlap_grad.hip - central-difference gradient of the time-evolution overlap objective from ONE pair of eigen-solves
// per trajectory (gfx950 only).
//
// The optimiser the reference runs per time step (scipy BFGS on `obj`, qmps/new_time_evolve.py:284, scripts/loschmidt.py:371)
// differentiates the objective by finite differences: 2 P more dominant-eigenvalue solves per iterate, each of a map that differs
// from the iterate's by O(h) = 1e-6.  With the right AND the left fixed point of the iterate's map T,
//     T(r) = eta r ,   T^+(y) = conj(eta) y     (<a, b> = tr(a^+ b)),
// the dominant eigenvalue of a neighbour T' = T + O(h) is, to SECOND order in h,
//     eta' = <y, T'(r)> / <y, r> ,
// (error O(h^2 |dT|^2 / gap) ~ 1e-12: the accuracy of the solves themselves; measured against dense eigen-solves in
// tests/test_evolve_gpu.py), so the 2 P neighbours cost one application of the map each instead of a power iteration each.
// With T'(x) = sum_s C_s x Bm'_s^+ only Bm' = merge(B', B') depends on the neighbour:
//     <y, T'(r)> = sum_s tr(Bm'_s^+ G_s) ,   G_s = y^+ C_s r   (per trajectory, overlap_g_kernel),
// an elementwise contraction per neighbour (overlap_probe_kernel).
//
// Thread (i, j) of a D x D tile per item, tiles in LDS (D = 2, 4: several items per wave).  These kernels are a few percent of
// a gradient evaluation (the two power iterations dominate), so they are written for clarity, one code path for every D.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_overlap_d4.h"     // cmma16_3m
#include "qmps_circuit_wave.h"

namespace qmps {

namespace {

// (cfma: c += a b - qmps_overlap_d4.h)
__device__ __forceinline__ void cfma_conj1(double2 a, double2 b, double2& c) {   // c += conj(a) b
  c.x = dfma(a.x, b.x, c.x);
  c.x = dfma(a.y, b.y, c.x);
  c.y = dfma(a.x, b.y, c.y);
  c.y = dfma(-a.y, b.x, c.y);
}

// sum of two values over the N = D * D threads of one item (consecutive lanes, or the whole workgroup at D = 16)
template <int N>
__device__ __forceinline__ void item_sum2(double& a, double& b, double* red, int tid) {
  if constexpr (N == 4) {
    a = quad_sum(a);
    b = quad_sum(b);
  } else if constexpr (N == 16) {
    a = row16_sum(a);
    b = row16_sum(b);
  } else if constexpr (N == 64) {
    a = wave_sum(a);
    b = wave_sum(b);
  } else {
    constexpr int WAVES = N / 64;
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((tid & 63) == 0) {
      red[tid >> 6] = a;
      red[WAVES + (tid >> 6)] = b;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      sa += red[w];
      sb += red[WAVES + w];
    }
    a = sa;
    b = sb;
  }
}

}  // namespace

// G_s = y^+ C_s r (s < 4) and <y, r> per trajectory
template <int D>
__global__ __launch_bounds__((D * D < 64) ? 64 : D * D) void overlap_g_kernel(OverlapGradArgs p) {
  constexpr int N = D * D, P = D + 1, THREADS = N < 64 ? 64 : N, ITEMS = THREADS / N;
  __shared__ double2 sA[ITEMS][2][D][P], sC[ITEMS][4][D][P], sR[ITEMS][D][P], sYv[ITEMS][D][P];
  __shared__ double red[16];
  const int tid = threadIdx.x, e = tid / N, l = tid % N, i = l / D, j = l % D;
  const int64_t t = (int64_t)blockIdx.x * ITEMS + e;
  if (ITEMS == 1 && p.active != nullptr && t < p.T && p.active[t] == 0) return;      // skipped trajectory (uniform exit)
  const int64_t tt = t < p.T ? t : p.T - 1;       // surplus lanes of the last workgroup shadow a real trajectory
  {
    const double2* Ap = (const double2*)p.A + tt * (2 * N);
    sA[e][0][i][j] = Ap[l];
    sA[e][1][i][j] = Ap[N + l];
    sR[e][i][j] = ((const double2*)p.r)[tt * N + l];
    sYv[e][i][j] = ((const double2*)p.y)[tt * N + l];
  }
  __syncthreads();
  {
    double2 aa[4];
#pragma unroll
    for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        double2 u = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < D; ++k) cfma(sA[e][t1][i][k], sA[e][t2][k][j], u);
        aa[2 * t1 + t2] = u;
      }
    const double2* W = (const double2*)p.WW;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double2 c = make_double2(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < 4; ++q) cfma(W[s * 4 + q], aa[q], c);
      sC[e][s][i][j] = c;
    }
  }
  __syncthreads();
  // Z_s = C_s r (kept in this thread's registers, then published in place of C_s)
  double2 z[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double2 v = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < D; ++k) cfma(sC[e][s][i][k], sR[e][k][j], v);
    z[s] = v;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) sC[e][s][i][j] = z[s];
  __syncthreads();
  // G_s = y^+ Z_s
  double2* Gp = (double2*)p.G + tt * (4 * N);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double2 g = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < D; ++k) cfma_conj1(sYv[e][k][i], sC[e][s][k][j], g);
    if (t < p.T && !(p.active != nullptr && p.active[t] == 0)) Gp[s * N + l] = g;
  }
  const double2 yv = sYv[e][i][j], rv = sR[e][i][j];
  double a = yv.x * rv.x + yv.y * rv.y, b = yv.x * rv.y - yv.y * rv.x;   // conj(y) r
  item_sum2<N>(a, b, red, tid);
  if (l == 0 && t < p.T && !(p.active != nullptr && p.active[t] == 0)) ((double2*)p.yr)[t] = make_double2(a, b);
}

// D = 16: the same on the matrix cores, FOUR WAVES per trajectory (round 4): wave w = (t1, t2) forms A_t1 A_t2 (published through
// LDS), then C_w = sum_t WW[w][t] A A_t, Z_w = C_w r and G_w = y^+ Z_w - three complex 16 x 16 x 16 products per wave, operands
// loaded from HBM in the MFMA layouts (the set-up of overlap_mfma_d16x4_body, qmps_overlap.hip, does the first half the same way).
// The generic kernel above (a thread per matrix element, three LDS stages) took 9.5 us for 256 trajectories on the critical path of a
// gradient batch (solves -> G -> probes).
__global__ __launch_bounds__(256) void overlap_g_d16_kernel(OverlapGradArgs p) {
  constexpr int D = 16, N = 256, LD = 17;
  __shared__ double2 sX[4][N];            // exchange: one accumulator-layout matrix per wave, element (q, lane) at [q * 64 + lane]
  __shared__ double2 sT[4][D * LD];       // wave-private transposes
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t t = blockIdx.x;
  if (p.active != nullptr && p.active[t] == 0) return;      // skipped trajectory (uniform over the workgroup)
  const double2* Ap = (const double2*)p.A + t * (2 * N);
  const int t1 = wave >> 1, t2 = wave & 1;
  v4f64 zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
  {
    double pa[4], pai[4];
    v4f64 qa, qai;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 va = Ap[(t1 * D + c) * D + 4 * kk + g], wa = Ap[(t2 * D + 4 * kk + g) * D + c];
      pa[kk] = va.x; pai[kk] = va.y;
      qa[kk] = wa.x; qai[kk] = wa.y;
    }
    cmma16(pa, pai, qa, qai, zr, zi);       // A_t1 A_t2
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sX[wave][q * 64 + lane] = make_double2(zr[q], zi[q]);
  // r in B-layout, y^+ in A-layout ((y^+)[c][4 kk + g] = conj(y[4 kk + g][c])): the same addresses
  v4f64 rr, ri;
  double yr_[4], yin[4];
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double2 rv = ((const double2*)p.r)[t * N + (4 * kk + g) * D + c], yv = ((const double2*)p.y)[t * N + (4 * kk + g) * D + c];
    rr[kk] = rv.x; ri[kk] = rv.y;
    yr_[kk] = yv.x; yin[kk] = -yv.y;
    a = dfma(yv.x, rv.x, dfma(yv.y, rv.y, a));        // conj(y) r
    b = dfma(yv.x, rv.y, dfma(-yv.y, rv.x, b));
  }
  __syncthreads();
  v4f64 sr = {0, 0, 0, 0}, si = {0, 0, 0, 0};
  {
    const double2* W = (const double2*)p.WW;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const double2 w = W[wave * 4 + tt];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 v = sX[tt][q * 64 + lane];
        sr[q] += w.x * v.x - w.y * v.y;
        si[q] += w.x * v.y + w.y * v.x;
      }
    }
  }
  // C_w into the A-layout (padded transpose through the wave's own LDS tile)
  double cre[4], cim[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) sT[wave][(4 * q + g) * LD + c] = make_double2(sr[q], si[q]);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double2 v = sT[wave][c * LD + 4 * kk + g];
    cre[kk] = v.x;
    cim[kk] = v.y;
  }
  v4f64 ur = {0, 0, 0, 0}, ui = {0, 0, 0, 0}, gr = {0, 0, 0, 0}, gi = {0, 0, 0, 0};
  cmma16_3m(cre, cim, rr, ri, ur, ui);          // Z_w = C_w r
  cmma16_3m(yr_, yin, ur, ui, gr, gi);          // G_w = y^+ Z_w
  double2* Gp = (double2*)p.G + t * (4 * N) + wave * N;
#pragma unroll
  for (int q = 0; q < 4; ++q) Gp[(4 * q + g) * D + c] = make_double2(gr[q], gi[q]);
  if (wave == 0) {
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) ((double2*)p.yr)[t] = make_double2(a, b);
  }
}

// eta' = sum_s tr(Bm'_s^+ G_s) / <y, r>,  f = -sqrt|eta'|  per central-difference neighbour
template <int D>
__global__ __launch_bounds__((D * D < 64) ? 64 : D * D) void overlap_probe_kernel(OverlapGradArgs p) {
  constexpr int N = D * D, P = D + 1, THREADS = N < 64 ? 64 : N, ITEMS = THREADS / N;
  __shared__ double2 sB[ITEMS][2][D][P];
  __shared__ double red[16];
  const int tid = threadIdx.x, e = tid / N, l = tid % N, i = l / D, j = l % D;
  // items [0, T G2P): the neighbours; with p.Bc set, items [T G2P, T G2P + T): the iterates themselves (same launch)
  const int64_t nn = p.T * p.G2P, nb = nn + (p.Bc != nullptr ? p.T : 0);
  const int64_t b = (int64_t)blockIdx.x * ITEMS + e;
  const int64_t bb = b < nb ? b : nb - 1;
  const bool centre = bb >= nn;
  const int64_t t = centre ? bb - nn : bb / p.G2P;
  if (ITEMS == 1 && p.active != nullptr && p.active[t] == 0) return;
  {
    const double2* Bp = centre ? (const double2*)p.Bc + t * (2 * N) : (const double2*)p.Bt + bb * (2 * N);
    sB[e][0][i][j] = Bp[l];
    sB[e][1][i][j] = Bp[N + l];
  }
  __syncthreads();
  const double2* Gp = (const double2*)p.G + t * (4 * N);
  double nr = 0.0, ni = 0.0;
#pragma unroll
  for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      double2 bm = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) cfma(sB[e][s1][i][k], sB[e][s2][k][j], bm);
      const double2 g = Gp[(2 * s1 + s2) * N + l];
      nr = dfma(bm.x, g.x, dfma(bm.y, g.y, nr));       // conj(bm) g
      ni = dfma(bm.x, g.y, dfma(-bm.y, g.x, ni));
    }
  item_sum2<N>(nr, ni, red, tid);
  if (l == 0 && b < nb && !(p.active != nullptr && p.active[t] == 0)) {
    const double2 d = ((const double2*)p.yr)[t];
    const double den = d.x * d.x + d.y * d.y;
    const double er = (nr * d.x + ni * d.y) / den, ei = (ni * d.x - nr * d.y) / den;
    const double f = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
    if (centre) p.fc_out[t] = f;
    else p.f_out[b] = f;
  }
}

// D = 16: ONE WAVE per neighbour on the matrix cores (round 4).  merge(B', B')_s = B'_s1 B'_s2 are four complex 16 x 16 x 16 products:
// 48 v_mfma_f64_16x16x4 in the three-product form (qmps_overlap_d4.h), operands loaded from HBM straight into the A- and
// B-layouts, the products left in the accumulator layout and contracted there with G_s (read in the same layout) - no LDS at all.
// The round-3 kernel did the products on the vector pipe from an LDS copy (2 x 2 register tile per lane, 1 024 complex
// multiply-adds per lane): 20.1 -> 18.4 us for the 4 352 probes of 256 iterates - the kernel is bound by the 35 MB of neighbour
// tensors it streams in (and 70 MB of G from L2), not by its arithmetic: 2 TB/s + the latency of a lone wave's loads.
// Tried and dropped in round 3: neighbours LINEARISED around the iterate (<y, T'(r)> = <y, T(r)> + sum_s <B'_s - B_s, M_s>, M once per
// iterate; the even orders cancel in the central difference) - an elementwise contraction per neighbour, but it reads three
// tiles per neighbour (B', B, M: memory-bound, 80 us) and the eight extra products for M cost the G kernel 50 us: slower in sum.
__global__ __launch_bounds__(64) void overlap_probe_d16_kernel(OverlapGradArgs p) {
  constexpr int D = 16, N = 256;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int64_t b = blockIdx.x, nn = p.T * p.G2P;
  const bool centre = b >= nn;                      // (with p.Bc set the launch carries T more items: the iterates themselves)
  const int64_t t = centre ? b - nn : b / p.G2P;
  if (p.active != nullptr && p.active[t] == 0) return;
  const double2* Bp = centre ? (const double2*)p.Bc + t * (2 * N) : (const double2*)p.Bt + b * (2 * N);
  double par[2][4], pai[2][4];      // B'_s in A-layout: lane (g, c) holds B'_s[c][4 kk + g]
  v4f64 qbr[2], qbi[2];             // B'_s in B-layout: lane (g, c) holds B'_s[4 kk + g][c]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 va = Bp[(s * D + c) * D + 4 * kk + g], vb = Bp[(s * D + 4 * kk + g) * D + c];
      par[s][kk] = va.x;
      pai[s][kk] = va.y;
      qbr[s][kk] = vb.x;
      qbi[s][kk] = vb.y;
    }
  const double2* Gp = (const double2*)p.G + t * (4 * N);
  double nr = 0.0, ni = 0.0;
#pragma unroll
  for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      v4f64 mr = {0, 0, 0, 0}, mi = {0, 0, 0, 0};
      cmma16_3m(par[s1], pai[s1], qbr[s2], qbi[s2], mr, mi);       // Bm'_s in the accumulator layout: register q, lane (g, c) = element [4 q + g][c]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 gv = Gp[(2 * s1 + s2) * N + (4 * q + g) * D + c];
        nr = dfma(mr[q], gv.x, dfma(mi[q], gv.y, nr));       // conj(bm) g
        ni = dfma(mr[q], gv.y, dfma(-mi[q], gv.x, ni));
      }
    }
  nr = wave_sum(nr);
  ni = wave_sum(ni);
  if (lane == 0) {
    const double2 d = ((const double2*)p.yr)[t];
    const double den = d.x * d.x + d.y * d.y;
    const double er = (nr * d.x + ni * d.y) / den, ei = (ni * d.x - nr * d.y) / den;
    const double f = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
    if (centre) p.fc_out[t] = f;
    else p.f_out[b] = f;
  }
}

// The same with the neighbour's tensor BUILT by the probing wave (round 5): parameters -> the five-qubit circuit distributed over
// the lanes (qmps_circuit_wave.h: two columns per pass, eight passes) -> the tensor in LDS (8 KB, padded rows) -> the MFMA operand
// layouts.  Round 4 built the 2 P T neighbour tensors in a kernel of their own on a second stream beside the eigen-solves (65 us
// for 4 096 tensors, one lane per column), wrote 35 MB to HBM and read them back here; that kernel, its two cross-stream
// dependencies (~7 us each on the critical path) and its contention with the solves are gone.  The iterates' own items (the
// two-sided objective) still read the tensor the solves used.
template <int KIND>
__global__ __launch_bounds__(64) void overlap_probe_build_d16_kernel(OverlapGradArgs p) {
  constexpr int D = 16, N = 256, LD = 17;
  __shared__ double2 sB[2 * D * LD];                // [s][i][j] with rows padded to 17
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int64_t b = blockIdx.x, nn = p.T * p.G2P;
  const bool centre = b >= nn;                      // (with p.Bc set the launch carries T more items: the iterates themselves)
  const int64_t t = centre ? b - nn : b / p.G2P;
  if (p.active != nullptr && p.active[t] == 0) return;
  double par[2][4], pai[2][4];      // B'_s in A-layout: lane (g, c) holds B'_s[c][4 kk + g]
  v4f64 qbr[2], qbi[2];             // B'_s in B-layout: lane (g, c) holds B'_s[4 kk + g][c]
  if (centre) {
    const double2* Bp = (const double2*)p.Bc + t * (2 * N);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 va = Bp[(s * D + c) * D + 4 * kk + g], vb = Bp[(s * D + 4 * kk + g) * D + c];
        par[s][kk] = va.x;
        pai[s][kk] = va.y;
        qbr[s][kk] = vb.x;
        qbi[s][kk] = vb.y;
      }
  } else {
    const int P = p.G2P / 2, k = (int)(b - t * p.G2P);
    const int isel = k % P;
    const double shift = k < P ? p.fd_h : -p.fd_h;
    // one sincos per angle and wave: lane l takes angle l (P <= 64)
    double cn = 1.0, sn = 0.0;
    if (lane < P) {
      double v = p.fd_params[t * P + lane];
      if (lane == isel) v += shift;
      sincos(0.5 * v, &sn, &cn);
    }
    const int a = lane & 31;
#pragma unroll 1
    for (int w = 0; w < 8; ++w) {
      const int j = 2 * w + (lane >> 5);
      double re, im;
      shallow_cnot_wave_column_d16<KIND>(cn, sn, P, j, re, im);
      sB[((a & 1) * D + (a >> 1)) * LD + j] = make_double2(re, im);      // A[s][i][j] = amplitude[2 i + s] of column j
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 va = sB[(s * D + c) * LD + 4 * kk + g], vb = sB[(s * D + 4 * kk + g) * LD + c];
        par[s][kk] = va.x;
        pai[s][kk] = va.y;
        qbr[s][kk] = vb.x;
        qbi[s][kk] = vb.y;
      }
  }
  const double2* Gp = (const double2*)p.G + t * (4 * N);
  double nr = 0.0, ni = 0.0;
#pragma unroll
  for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      v4f64 mr = {0, 0, 0, 0}, mi = {0, 0, 0, 0};
      cmma16_3m(par[s1], pai[s1], qbr[s2], qbi[s2], mr, mi);       // Bm'_s in the accumulator layout: register q, lane (g, c) = element [4 q + g][c]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 gv = Gp[(2 * s1 + s2) * N + (4 * q + g) * D + c];
        nr = dfma(mr[q], gv.x, dfma(mi[q], gv.y, nr));       // conj(bm) g
        ni = dfma(mr[q], gv.y, dfma(-mi[q], gv.x, ni));
      }
    }
  nr = wave_sum(nr);
  ni = wave_sum(ni);
  if (lane == 0) {
    const double2 d = ((const double2*)p.yr)[t];
    const double den = d.x * d.x + d.y * d.y;
    const double er = (nr * d.x + ni * d.y) / den, ei = (ni * d.x - nr * d.y) / den;
    const double f = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
    if (centre) p.fc_out[t] = f;
    else p.f_out[b] = f;
  }
}

template <int D>
static hipError_t launch_grad_d(const OverlapGradArgs& a, hipStream_t st) {
  constexpr int N = D * D, THREADS = N < 64 ? 64 : N, ITEMS = THREADS / N;
  if constexpr (D == 16) hipLaunchKernelGGL(overlap_g_d16_kernel, dim3((unsigned)a.T), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((overlap_g_kernel<D>), dim3((unsigned)((a.T + ITEMS - 1) / ITEMS)), dim3(THREADS), 0, st, a);
  const int64_t nb = a.T * a.G2P + (a.Bc != nullptr ? a.T : 0);      // neighbours (+ the iterates themselves: same launch)
  if constexpr (D == 16) {
    if (a.fd_params != nullptr && overlap_probe_fusable(16, a.kind, a.G2P / 2)) {
      if (a.kind == 0) hipLaunchKernelGGL(overlap_probe_build_d16_kernel<0>, dim3((unsigned)nb), dim3(64), 0, st, a);
      else hipLaunchKernelGGL(overlap_probe_build_d16_kernel<3>, dim3((unsigned)nb), dim3(64), 0, st, a);
    } else {
      hipLaunchKernelGGL(overlap_probe_d16_kernel, dim3((unsigned)nb), dim3(64), 0, st, a);
    }
  } else hipLaunchKernelGGL((overlap_probe_kernel<D>), dim3((unsigned)((nb + ITEMS - 1) / ITEMS)), dim3(THREADS), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_overlap_grad(int D, const OverlapGradArgs& a, hipStream_t st) {
  if (a.T <= 0 || a.G2P <= 0) return hipSuccess;
  switch (D) {
    case 2: return launch_grad_d<2>(a, st);
    case 4: return launch_grad_d<4>(a, st);
    case 8: return launch_grad_d<8>(a, st);
    case 16: return launch_grad_d<16>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace qmps
