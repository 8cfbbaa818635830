// qmps_roto_d8.hip - the WHOLE rotosolve run of a D = 8 ansatz in one kernel launch (gfx950 only).
//
// BASELINE.json configs[3] (Heisenberg XXZ, D = 8, depth 3, 256 restarts x 3 angle samples) is all latency: 768 evaluations
// per parameter update occupy three quarters of the chip's SIMDs for ~22 us, and every update used to be three kernels
// (shifted ansatz, solve + energy, update) with their launch gaps: 42 us.  Restarts are independent, so - as at D = 2
// (rotosolve_fused_d2_kernel) - the sequential loop over parameters and sweeps needs no grid-wide step: ONE WORKGROUP PER
// RESTART, one wave per shift (NSH = 3: {0, +pi/2, -pi/2}, qmps/rotosolve.py:154-181; NSH = 6: {0, pi, +-pi/2, +-pi/4},
// qmps/tools.py:422-457).  Per parameter update each wave
//   1. builds its shifted state tensor in LDS: the 8 columns of the 4-qubit ansatz circuit, DISTRIBUTED over the wave - lane
//      8 j + a holds the two amplitudes (q0 q1 q2) = a, q3 = 0 | 1 of column j; diagonal gates are local, rx / H on q0..q2 are
//      butterflies with the lane a xor 1 | 2 | 4 (ds_swizzle), the CNOT ladder is one gather (ds_bpermute) - ~300 instructions
//      instead of ~2 300 for a lane that simulates a whole column by itself;
//   2. solves the environment directly (env_direct_d8_solve, qmps_direct_d8.h), accepts it by one power step (or iterates),
//      tests positive definiteness and evaluates the energies - the arithmetic of energy_block_kernel<8, true, FUSED>, wave-local;
//   3. publishes its energy; wave 0 applies the update to the restart's parameter vector in LDS.
// No launch, no graph replay, no HBM traffic inside the run: a parameter update costs the latency of ONE evaluation.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_direct_d8.h"
#include "qmps_roto_math.h"
#include "qmps_circuit.h"     // roto_shift_value

namespace qmps {

namespace {

struct D8Work {
  double2 sA[2][8][9];
  double2 sR[8][9];
  double2 sX[2][8][9];
  double2 sT[2][8][9];
  double sM8[64][17];
  double sT8[8][9];
  double cs[64][2];        // cos / sin of the half angles of this wave's (shifted) parameter vector
};

__device__ __forceinline__ void wsync() { __builtin_amdgcn_wave_barrier(); }

template <int PATTERN>
__device__ __forceinline__ double swz(double v) {      // value of the lane (own index xor mask), within groups of 32 lanes
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
  return __hiloint2double(hi, lo);
}

struct Amp {
  double re[2], im[2];
};

// rx on the qubit whose bit is the lane-index bit selected by PATTERN: a' = c a - i s (partner's a), both local amplitudes
template <int PATTERN>
__device__ __forceinline__ void rx_cross(Amp& v, double c, double s) {
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const double pr = swz<PATTERN>(v.re[l]), pi = swz<PATTERN>(v.im[l]);
    const double nr = dfma(c, v.re[l], s * pi), ni = dfma(c, v.im[l], -s * pr);
    v.re[l] = nr;
    v.im[l] = ni;
  }
}

// ShallowCNOTStateTensor (KIND 0) / ShallowCNOTStateTensor3 (KIND 3) at D = 8 (qmps/represent.py:288-310, 334-354): column j of the
// 4-qubit circuit on |0>|j>, distributed over the lanes 8 j + a; writes A[s][i][j] = amplitude[2 i + s] into w.sA.
// par(l): angle l of this wave's (shifted) parameter vector.
template <int KIND, class Par>
__device__ __forceinline__ void build_tensor_d8(D8Work& w, Par par, int n_params, int lane) {
  constexpr int per = KIND == 3 ? 3 : 2;
  const int j = lane >> 3, a = lane & 7;
  // one sincos for the whole wave: lane l takes angle l
  wsync();
  if (lane < n_params) {
    double sn, cn;
    sincos(0.5 * par(lane), &sn, &cn);
    w.cs[lane][0] = cn;
    w.cs[lane][1] = sn;
  }
  wsync();
  Amp v;
  // |0>|j>: basis state x = j (q0 = 0), x = 2 a + l
  v.re[0] = (2 * a == j) ? 1.0 : 0.0;
  v.re[1] = (2 * a + 1 == j) ? 1.0 : 0.0;
  v.im[0] = v.im[1] = 0.0;
  const int pa = __builtin_popcount(a);
  auto rz_all = [&](double c, double s) {
    // prod_q rz(theta) = diag(exp(-i phi (4 - 2 popcount(x)))), phi = theta / 2, popcount(x) = popcount(a) + l
    const double c2 = dfma(c, c, -s * s), s2 = 2.0 * s * c, c4 = dfma(c2, c2, -s2 * s2), s4 = 2.0 * s2 * c2;
    // m = 4 - 2 pop:  exp(-i m phi) = (cos m phi, -sin m phi)
    auto phase = [&](int m, double& pr, double& pi) {
      pr = m == 0 ? 1.0 : ((m == 2 || m == -2) ? c2 : c4);
      pi = m == 0 ? 0.0 : (m == 2 ? -s2 : (m == -2 ? s2 : (m == 4 ? -s4 : s4)));
    };
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      double pr, pi;
      phase(4 - 2 * (pa + l), pr, pi);
      const double nr = dfma(v.re[l], pr, -v.im[l] * pi), ni = dfma(v.re[l], pi, v.im[l] * pr);
      v.re[l] = nr;
      v.im[l] = ni;
    }
  };
  for (int l0 = 0; l0 + per <= n_params; l0 += per) {
    rz_all(w.cs[l0][0], w.cs[l0][1]);
    {
      const double c = w.cs[l0 + 1][0], s = w.cs[l0 + 1][1];
      // rx on q3: local 2 x 2
      const double r0 = dfma(c, v.re[0], s * v.im[1]), i0 = dfma(c, v.im[0], -s * v.re[1]);
      const double r1 = dfma(c, v.re[1], s * v.im[0]), i1 = dfma(c, v.im[1], -s * v.re[0]);
      v.re[0] = r0; v.im[0] = i0; v.re[1] = r1; v.im[1] = i1;
      rx_cross<0x041F>(v, c, s);     // q2 <-> a bit 0
      rx_cross<0x081F>(v, c, s);     // q1 <-> a bit 1
      rx_cross<0x101F>(v, c, s);     // q0 <-> a bit 2
    }
    if (KIND == 3) rz_all(w.cs[l0 + 2][0], w.cs[l0 + 2][1]);
    {
      // H on q0 (a bit 2): (a + partner)/sqrt 2 on the 0 side, (partner - a)/sqrt 2 on the 1 side
      const double h = 0.70710678118654752, sg = (a & 4) ? -h : h;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const double pr = swz<0x101F>(v.re[l]), pi = swz<0x101F>(v.im[l]);
        v.re[l] = dfma(sg, v.re[l], h * pr);
        v.im[l] = dfma(sg, v.im[l], h * pi);
      }
    }
    {
      // CNOT(q2, q3), CNOT(q1, q2), CNOT(q0, q1): amplitude (b0 b1 b2 b3) moves to (b0, b1^b0, b2^b1, b3^b2).  Destination lane
      // a' = (b0, b1^b0, b2^b1) gathers from a = (b0, b1, b2) and swaps the two local amplitudes when b2 = 1.
      const int b0 = (a >> 2) & 1, b1 = ((a >> 1) & 1) ^ b0, b2 = (a & 1) ^ b1;
      const int src = (lane & ~7) | (b0 << 2) | (b1 << 1) | b2;
      const double r0 = __shfl(v.re[0], src, 64), i0 = __shfl(v.im[0], src, 64), r1 = __shfl(v.re[1], src, 64), i1 = __shfl(v.im[1], src, 64);
      v.re[0] = b2 ? r1 : r0; v.im[0] = b2 ? i1 : i0;
      v.re[1] = b2 ? r0 : r1; v.im[1] = b2 ? i0 : i1;
    }
  }
  wsync();
  w.sA[0][a][j] = make_double2(v.re[0], v.im[0]);
  w.sA[1][a][j] = make_double2(v.re[1], v.im[1]);
  wsync();
}

// Environment + energies of the tensor in w.sA, one wave (lane = 8 i + j owns r[i][j]): direct solve, power step(s) until
// ||r' - r||_F < tol, LDL^H pivots > 0, E = sum over the Hamiltonian terms (the reference's M(x) = np.sum(eps), qmps/tools.py:432-433):
// hsum is the SUM of the terms' 4 x 4 matrices, staged in LDS once per run (E is linear in h; fetching the terms from HBM
// inside every evaluation cost ~1 us of latency).
// The arithmetic of energy_block_kernel<8, true, FUSED> (qmps_kernels.hip) with wave-local synchronisation.
__device__ __forceinline__ double eval_d8(D8Work& w, const double2* hsum, int max_iter, double tol, int lane,
                                          int& status_out, long long* prof = nullptr, int* iters_out = nullptr) {
  constexpr int D = 8;
  auto tick = [&](int k) {
    if (prof) {
      __builtin_amdgcn_sched_barrier(0);
      prof[k] = wall_clock64();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int i = lane >> 3, j = lane & 7;
  wsync();
  double2 r = env_direct_d8_solve(w.sA, w.sT8, w.sM8, lane, prof);
  w.sR[i][j] = r;
  wsync();
  double2 ai_[2][D];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int k = 0; k < D; ++k) ai_[s][k] = w.sA[s][i][k];
  int status = QMPS_ST_NOT_CONVERGED;
  const double tol2 = tol * tol;
  for (int it = 1; it <= max_iter; ++it) {
    double2 rc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) rc[k] = w.sR[k][j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 a = ai_[s][k], rr = rc[k];
        xr = dfma(a.x, rr.x, xr);
        xr = dfma(-a.y, rr.y, xr);
        xi = dfma(a.x, rr.y, xi);
        xi = dfma(a.y, rr.x, xi);
      }
      w.sX[s][i][j] = make_double2(xr, xi);
    }
    wsync();
    double nr = 0.0, ni = 0.0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 a = w.sA[s][j][k], x = w.sX[s][i][k];
        nr = dfma(x.x, a.x, nr);
        nr = dfma(x.y, a.y, nr);
        ni = dfma(x.y, a.x, ni);
        ni = dfma(-x.x, a.y, ni);
      }
    // hermitise through LDS, normalise by the trace
    w.sT[0][i][j] = make_double2(nr, ni);
    wsync();
    const double2 m = w.sT[0][j][i];
    double2 n = make_double2(0.5 * (nr + m.x), (i == j) ? 0.0 : 0.5 * (ni - m.y));
    const double tr = wave_sum(i == j ? n.x : 0.0);
    const double inv = 1.0 / tr;
    n.x *= inv;
    n.y *= inv;
    const double dr = n.x - r.x, di = n.y - r.y;
    const double d2 = wave_sum(dr * dr + di * di);
    r = n;
    wsync();
    w.sR[i][j] = r;
    wsync();
    if (iters_out) *iters_out = it;
    if (d2 < tol2) {
      status = QMPS_ST_OK;
      break;
    }
  }
  tick(5);
  if (status == QMPS_ST_OK) {
    // positive definiteness: the pivots of LDL^H, all 64 lanes at once (thread (i, j) owns the Schur-complement entry)
    double2 S = r;
    bool ok = true;
    for (int c = 0; c < D; ++c) {
      wsync();
      w.sT[1][i][j] = S;
      wsync();
      const double pc = w.sT[1][c][c].x;
      ok = ok && (pc > 0.0);
      const double2 li = w.sT[1][i][c], lj = w.sT[1][j][c];
      const double inv = fast_rcp(pc);
      const double wr = (li.x * lj.x + li.y * lj.y) * inv, wi = (li.y * lj.x - li.x * lj.y) * inv;
      S.x -= wr;
      S.y -= wi;
    }
    if (!ok) status = QMPS_ST_NOT_PD;
    wsync();
  }
  tick(6);
  // ---- energy: rho[tau][sigma] = tr(B_tau r B_sigma^+), B_(2 t1 + t2) = A_t1 A_t2 (qmps/tools.py:432-433 through the merged
  // two-site tensor).  Two stages - the four B_tau, then Y_tau = B_tau r - and the lane's share Y_tau[i][j] conj(B_sigma[i][j]);
  // E is linear in rho: one wave sum.  (At one wave per SIMD an evaluation costs its instruction count - ~1.9 ns per issue
  // slot, LDS reads two - so this form, 64 complex multiply-adds and 56 LDS reads, replaced one with 112 and ~240.)
  const double trr = wave_sum(i == j ? r.x : 0.0);
  double2 (*sB)[8][9] = w.sX;                // B_0, B_1 in sX[0..1], B_2, B_3 in sT[0..1] (contiguous in D8Work)
  static_assert(offsetof(D8Work, sT) == offsetof(D8Work, sX) + sizeof(D8Work::sX), "B tiles span sX and sT");
  double2 bt[4];
  {
    double2 acol[2][D];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) acol[s][k] = w.sA[s][k][j];
#pragma unroll
    for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        double br = 0.0, bi = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double2 a = ai_[t1][k], c = acol[t2][k];
          br = dfma(a.x, c.x, br);
          br = dfma(-a.y, c.y, br);
          bi = dfma(a.x, c.y, bi);
          bi = dfma(a.y, c.x, bi);
        }
        bt[2 * t1 + t2] = make_double2(br, bi);
      }
  }
  wsync();
#pragma unroll
  for (int t = 0; t < 4; ++t) sB[t][i][j] = bt[t];
  wsync();
  double2 rho_loc[4][4];
  {
    double2 rc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) rc[k] = w.sR[k][j];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double yr = 0.0, yi = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double2 b = sB[t][i][k], rr = rc[k];
        yr = dfma(b.x, rr.x, yr);
        yr = dfma(-b.y, rr.y, yr);
        yi = dfma(b.x, rr.y, yi);
        yi = dfma(b.y, rr.x, yi);
      }
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        rho_loc[t][sg] = make_double2(yr * bt[sg].x + yi * bt[sg].y, yi * bt[sg].x - yr * bt[sg].y);
    }
  }
  double e = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double2 hv = hsum[s * 4 + t];
      e = dfma(hv.x, rho_loc[t][s].x, e);
      e = dfma(-hv.y, rho_loc[t][s].y, e);
    }
  e = wave_sum(e) / trr;
  tick(7);
  status_out = status;
  return e;
}

// The parameter update of one rotosolve step (qmps/tools.py:Rotosolve / DoubleRotosolve closed forms, as roto_update_kernel).
// Deliberately NOT inlined: inside the kernel's loop the compiler hoists the eighteen polynomial constants of atan2 out of
// the loop, runs out of registers around the 64 x 64 elimination and reloads them from scratch one dependent load at a time
// (measured: 4.3 us per update instead of 1.6).
__device__ __noinline__ double roto_new_angle3(double par, double e0, double e1, double e2) {
  const double theta = -1.5707963267948966 - atan2(2.0 * e0 - e1 - e2, e1 - e2);
  return wrap_pi(par + wrap_pi(theta));
}
__device__ __noinline__ double roto_new_angle6(double par, double e0, double e1, double e2, double e3, double e4, double e5, int rule) {
  const double A = e0 + e1, Bv = e0 - e1, C = e2 + e3, Dv = e2 - e3, Ev = e4 - e5;
  const double a = 0.25 * (2.0 * Ev - 1.4142135623730951 * Dv), b = 0.25 * (A - C), c = 0.5 * Dv, d = 0.5 * Bv;
  return par + double_sinusoid_step(a, b, c, d, rule);
}

}  // namespace

// Three waves per restart whatever NSH: with six shifts every wave evaluates two of them, one after the other (six waves would
// put two on one SIMD and halve each one's register file: the solve then spills - 60 us per update measured).
template <int KIND, int NSH>
__global__ __launch_bounds__(192) void rotosolve_fused_d8_kernel(RotoArgs p) {
  extern __shared__ double2 lds_dyn[];
  char* lds_raw = (char*)lds_dyn;
  D8Work* work = (D8Work*)lds_raw;
  double* s_par = (double*)(lds_raw + 3 * sizeof(D8Work));        // the restart's parameter vector [P]
  double* s_e = s_par + 64;                                        // [NSH] energies of the shifted evaluations
  int* s_st = (int*)(s_e + 8);                                     // [NSH]
  double2* s_h = (double2*)(s_st + 8);                             // [16] sum of the Hamiltonian terms
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x;
  if (r >= p.R) return;
  D8Work& w = work[wave];
  for (int l = threadIdx.x; l < p.P; l += blockDim.x) s_par[l] = p.base[(int64_t)r * p.P + l];
  if (threadIdx.x < 16) {
    const double2* h = (const double2*)p.h;
    double2 acc = h[threadIdx.x];
    for (int q = 1; q < p.n_terms; ++q) {
      acc.x += h[q * 16 + threadIdx.x].x;
      acc.y += h[q * 16 + threadIdx.x].y;
    }
    s_h[threadIdx.x] = acc;
  }
  __syncthreads();
#ifdef QMPS_D8_PROFILE      // scratch instrumentation (profiles/experiments/scratch/d8_profile.sh): phase clocks of restart 0 into the history buffer
  long long tp[6] = {0, 0, 0, 0, 0, 0}, fine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = wall_clock64();
  int prof_max_it = 0;
  long long prof_sum_it = 0;
#define QMPS_TICK(k) do { if (r == 0 && threadIdx.x == 0) tp[k] = wall_clock64(); } while (0)
#else
#define QMPS_TICK(k) do { } while (0)
#endif
  auto evaluate = [&](int i_sel, int shift_index) {
    const double shift = roto_shift_value(NSH, shift_index);
    QMPS_TICK(0);
    build_tensor_d8<KIND>(w, [&](int l) { return s_par[l] + (l == i_sel ? shift : 0.0); }, p.P, lane);
    QMPS_TICK(1);
    int st;
#ifdef QMPS_D8_PROFILE
    int its = 0;
    const double e = eval_d8(w, s_h, p.max_iter, p.tol, lane, st, fine, &its);
    prof_max_it = its > prof_max_it ? its : prof_max_it;
    prof_sum_it += its;
#else
    const double e = eval_d8(w, s_h, p.max_iter, p.tol, lane, st);
#endif
    QMPS_TICK(2);
    if (lane == 0) {
      s_e[shift_index] = e;
      s_st[shift_index] = st;
    }
  };
  // ONE call site for the evaluation (it is ~12 000 instructions: inlined once, LDS accesses stay ds_* instructions - through a
  // real function call they would become flat loads): the record of the last sweep - the unshifted evaluation of the final
  // vector, wave 0 - is one more trip round the same loop.
  const int n_updates = p.n_sweeps * p.P;
  for (int u = 0; u <= n_updates; ++u) {
    const bool last = u == n_updates;
    const int sw = u / p.P, i = u - sw * p.P;
    for (int k = wave; k < NSH; k += 3) evaluate(last ? -1 : i, k);
    __syncthreads();
    QMPS_TICK(3);
    if (threadIdx.x == 0) {
      // the shift-0 evaluation of a sweep's first parameter IS the evaluation of the vector the previous sweep left: its record
      if (i == 0 && sw > 0) p.hist[(int64_t)(sw - 1) * p.R + r] = s_e[0];
      bool ok = !last;
      for (int k = 0; k < NSH; ++k) ok = ok && s_st[k] == QMPS_ST_OK;
      if (ok) {          // (an evaluation without a valid environment leaves the parameter untouched)
        double e[6];
#pragma unroll
        for (int k = 0; k < NSH; ++k) e[k] = s_e[k];
        s_par[i] = NSH == 3 ? roto_new_angle3(s_par[i], e[0], e[1], e[2]) : roto_new_angle6(s_par[i], e[0], e[1], e[2], e[3], e[4], e[5], p.rule);
      }
    }
    __syncthreads();
#ifdef QMPS_D8_PROFILE
    if (r == 0 && threadIdx.x == 0 && sw == p.n_sweeps / 2 && i == 1) {
      tp[4] = wall_clock64();
      double* dbg = p.hist + (int64_t)p.n_sweeps * p.R;       // caller allocates 16 extra doubles
      dbg[0] = (double)(tp[1] - tp[0]); dbg[1] = (double)(tp[2] - tp[1]); dbg[2] = (double)(tp[3] - tp[2]); dbg[3] = (double)(tp[4] - tp[3]);
      dbg[4] = (double)(tp[4] - tp[0]);
      for (int q = 0; q < 7; ++q) dbg[8 + q] = (double)(fine[q + 1] - fine[q]);
    }
#endif
  }
  __syncthreads();
  for (int l = threadIdx.x; l < p.P; l += blockDim.x) p.base[(int64_t)r * p.P + l] = s_par[l];
#ifdef QMPS_D8_PROFILE      // whole-kernel ticks of the first, a middle and the last restart, and the start offset of the last
  if (threadIdx.x == 0 && (r == 0 || r == p.R / 2 || r == p.R - 1)) {
    double* dbg = p.hist + (int64_t)p.n_sweeps * p.R;
    dbg[r == 0 ? 5 : (r == p.R - 1 ? 7 : 6)] = (double)(wall_clock64() - t_begin);
  }
  if (threadIdx.x == 0 && r < 4096) {      // per restart: whole-run ticks, HW_ID, XCC_ID
    double* dbg = p.hist + (int64_t)p.n_sweeps * p.R + 16;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[3 * r] = (double)(wall_clock64() - t_begin);
    dbg[3 * r + 1] = (double)hw;
    dbg[3 * r + 2] = (double)xcc;
  }
  if (lane == 0) {      // largest / total number of power steps of any evaluation of the run (dbg[15] as two ints, zeroed by the caller)
    int* cnt = (int*)(p.hist + (int64_t)p.n_sweeps * p.R + 15);
    atomicMax(cnt, prof_max_it);
    atomicAdd(cnt + 1, (int)prof_sum_it);
  }
#endif
}

hipError_t launch_rotosolve_fused_d8(int kind, const RotoArgs& a, hipStream_t st) {
  if (a.R <= 0) return hipSuccess;
  const dim3 grid((unsigned)a.R);
  // (at least 84 KB: more than half of a CU's 160 KB, so that no CU hosts two restarts - two workgroups on one CU would put two
  // waves on one SIMD and make those restarts, and with them the whole launch, half as fast again)
  const size_t need = 3 * sizeof(D8Work) + (64 + 8) * sizeof(double) + 8 * sizeof(int) + 16 * sizeof(double2);
  const int lds = (int)(need > (size_t)84 * 1024 ? need : (size_t)84 * 1024);
  hipError_t e = hipErrorInvalidValue;
  auto go = [&](auto kernel) {
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) hipLaunchKernelGGL(kernel, grid, dim3(192), lds, st, a);
  };
  if (a.nsh == 3 && kind == 0) go(rotosolve_fused_d8_kernel<0, 3>);
  else if (a.nsh == 3 && kind == 3) go(rotosolve_fused_d8_kernel<3, 3>);
  else if (a.nsh == 6 && kind == 0) go(rotosolve_fused_d8_kernel<0, 6>);
  else if (a.nsh == 6 && kind == 3) go(rotosolve_fused_d8_kernel<3, 6>);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

}  // namespace qmps
