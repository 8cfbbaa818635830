// qmps_direct.hip - the DIRECT environment solve fused with the two-site energy (gfx950 only).
//
// energy_direct_d4_kernel: D = 4, ONE DPP QUAD (4 lanes) PER EVALUATION, 16 evaluations per wave.  Per evaluation:
//   tensor (512 B, read once from HBM as part of the wave's contiguous 8 KB slab -> padded LDS tile)
//   -> the real 16 x 16 transfer matrix R, four rows per lane
//   -> (R - 1 + e_15 t^T) u = e_15 by Gauss-Jordan elimination in registers, pivot rows handed round the quad by
//      v_mov_b32_dpp quad_perm (VALU only; no LDS, no cross-quad traffic)
//   -> one power step as acceptance test (||T(r) - r||_F < tol; else the power method 2^m steps at a time)
//   -> LDL^H positive-definiteness test, two-site density matrix, energies of all Hamiltonian terms
//   -> E (8 B per term), iterations + status (8 B), optionally r (256 B); per-wave partial sums of E.
// One read of A and one store of E per evaluation: the environment never travels through HBM.
// The mathematics lives in qmps_direct_core.h (shared with the CPU lock-step emulation the test-suite uses).
// Replaces qmps/tools.py:176-182 (get_env_exact: the reference does an exact eigen-solve here as well) +
// qmps/represent.py:258-262 + qmps/ground_state.py:159-167.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_hip.h"
#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_circuit.h"
#include "qmps_direct_core.h"
#include "qmps_direct_d8.h"

namespace qmps {

namespace {

struct QuadOps {
  using V = double;
  using P = bool;
  int q;                  // row index of this lane inside its quad
  const double2* row;     // the evaluation's tensor in LDS: A_s[i][j] = row[(4 s + i) 4 + j]
  double* gj;             // QMPS_GJ_LDS: the quad's 16-double strip for the pivot rows of the elimination
  __device__ __forceinline__ P q_eq(int i) const { return q == i; }
  __device__ __forceinline__ P q_gt(int i) const { return q > i; }
  template <int L>
  static __device__ __forceinline__ V bcast(V v) { return quad_bcast<L>(v); }
  template <int L, int K>
  __device__ __forceinline__ void row_bcast(const V (&r)[16], V (&out)[16]) const {
#ifdef QMPS_GJ_LDS
    // through LDS: the owner lane parks the rest of its row, every lane of the quad reads it back (one address per quad:
    // a broadcast read); the LDS pipe works beside the vector ALU, which the DPP moves (2 per double) keep busy
    constexpr int K0 = K & ~1;
    __builtin_amdgcn_wave_barrier();
    if (q == L) {
#pragma unroll
      for (int j = K0; j < 16; j += 2) *(double2*)&gj[j] = make_double2(r[j], r[j + 1]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = K0; j < 16; j += 2) {
      const double2 t = *(const double2*)&gj[j];
      if (j >= K) out[j] = t.x;
      out[j + 1] = t.y;
    }
#else
#pragma unroll
    for (int j = K; j < 16; ++j) out[j] = quad_bcast<L>(r[j]);
#endif
  }
  static __device__ __forceinline__ V qsum(V v) { return quad_sum(v); }
  static __device__ __forceinline__ V qmax(V v) {
    v = fmax(v, quad_perm<0xB1>(v));
    return fmax(v, quad_perm<0x4E>(v));
  }
  static __device__ __forceinline__ V vabs(V v) { return fabs(v); }
  static __device__ __forceinline__ V vmax(V a, V b) { return fmax(a, b); }
  static __device__ __forceinline__ V sel(P p, V a, V b) { return p ? a : b; }
  static __device__ __forceinline__ V splat(double x) { return x; }
  static __device__ __forceinline__ V fma(V a, V b, V c) { return dfma(a, b, c); }
  static __device__ __forceinline__ V rcp(V t) { return fast_rcp(t); }
  static __device__ __forceinline__ P lt(V a, V b) { return a < b; }
  static __device__ __forceinline__ P gt0(V a) { return a > 0.0; }
  static __device__ __forceinline__ P p_and(P a, P b) { return a && b; }
  static __device__ __forceinline__ P p_not(P a) { return !a; }
  static __device__ __forceinline__ bool any(P a) { return __any(a) != 0; }   // wave-uniform: every quad stays in step
  __device__ __forceinline__ void own(int s, int j, V& re, V& im) const {
    const double2 v = row[(4 * s + q) * 4 + j];
    re = v.x;
    im = v.y;
  }
  __device__ __forceinline__ void uni(int s, int i, int j, V& re, V& im) const {
    const double2 v = row[(4 * s + i) * 4 + j];
    re = v.x;
    im = v.y;
  }
};

// The rare path (tensors that are not isometries, degenerate transfer spectra): the power method 2^m steps at a time,
// ONE EVALUATION AT A TIME WITH THE WHOLE WAVE ON THE MATRIX CORES.  In orthonormal Hermitian coordinates
// (a = 4 i + i': r_ii | sqrt2 Re r_ii' (i < i') | sqrt2 Im r_i'i (i > i')) the transfer map is one real 16 x 16 tile;
// squaring it is 4 x v_mfma_f64_16x16x4 (accumulator layout = B-operand layout of the next round, A fragments from a
// padded LDS image).  The recurrence is DirectD4::squaring's (qmps_direct_core.h: z_m = R^(2^m) r_0 / tr from
// r_0 = 1/4, stop at ||z_m - z_(m-1)||_F < tol, the matrix rescaled by 1/tr every round), which the CPU emulation
// runs in the quad layout; held in registers per quad that formulation needs two 16 x 16 matrices per evaluation
// (256 VGPRs + spills), and a kernel that fills the register file cannot share a SIMD with any other resident wave
// (a polling cost_finish_kernel or an RCCL kernel then costs the step a straggler wave, +3 us measured).  Here the
// rare path needs ~40 registers and no scratch memory.
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kSqLD = 17;                              // padded row of the LDS image
constexpr int kSqDoubles = 16 * kSqLD + 16;            // image + a 16-double strip

// todo: lanes whose evaluation needs the fall-back (alike in the four lanes of a quad).  x / steps / left of those
// evaluations are replaced; tiles: the wave's padded tensor tiles (stride pad bytes); img: kSqDoubles doubles of LDS.
__device__ __forceinline__ void squaring_fallback(const unsigned char* tiles, int pad, double* img, int lane, bool todo,
                                                  int max_iter, double tol2, double (&x)[4], double& steps, bool& left) {
  constexpr int D = 4, LD = kSqLD;
  constexpr double RS2 = 0.70710678118654752, S2 = 1.4142135623730951;
  const int g = lane >> 4, c = lane & 15, j = c >> 2, jp = c & 3, q = lane & 3;
  double* strip = img + 16 * LD;
  // lane constants of the matrix build: column (j, j') combines conj(A_s[g][j']) and conj(A_s[g][j]) with
  // (1, 0) on the diagonal, (1, 1)/sqrt2 for a real-part column, (-i, i)/sqrt2 for an imaginary-part column
  const double cpx = j == jp ? 1.0 : (j < jp ? RS2 : 0.0), cpy = j > jp ? -RS2 : 0.0;
  const double cqx = j < jp ? RS2 : 0.0, cqy = j > jp ? RS2 : 0.0;
  const bool diag = j == jp;
  unsigned long long mask = __ballot(todo);
  while (mask != 0) {
    const int e = (__ffsll((long long)mask) - 1) >> 2;        // wave-uniform
    mask &= ~(0xFull << (4 * e));
    const double2* sA = (const double2*)(tiles + e * pad);     // A_s[i][k] = sA[(4 s + i) 4 + k]
    // R[a][b] = tr(H_a T(H_b)) in accumulator layout: lane (g, c) holds R[(reg, g)][(j, j')], reg = 0 .. 3
    v4f64 R;
    {
      double pr[2], pi[2], qr[2], qi[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const double2 a = sA[(s * D + g) * D + jp], b = sA[(s * D + g) * D + j];
        pr[s] = dfma(cpx, a.x, cpy * a.y);
        pi[s] = dfma(cpy, a.x, -cpx * a.y);
        qr[s] = dfma(cqx, b.x, cqy * b.y);
        qi[s] = dfma(cqy, b.x, -cqx * b.y);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double gr = 0.0, gi = 0.0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const double2 u = sA[(s * D + reg) * D + j], v = sA[(s * D + reg) * D + jp];
          gr = dfma(u.x, pr[s], gr);
          gr = dfma(-u.y, pi[s], gr);
          gr = dfma(v.x, qr[s], gr);
          gr = dfma(-v.y, qi[s], gr);
          gi = dfma(u.x, pi[s], gi);
          gi = dfma(u.y, pr[s], gi);
          gi = dfma(v.x, qi[s], gi);
          gi = dfma(v.y, qr[s], gi);
        }
        // Re G (reg == g), sqrt2 Re G (reg < g), -sqrt2 Im G (reg > g: the sorted pair is (g, reg))
        R[reg] = reg == g ? gr : (reg < g ? S2 * gr : -S2 * gi);
      }
    }
    // image of a matrix in LDS (wave-private, LDS is in order per wave) -> its A-operand fragments M[c][4 kk + g]
    double af[4];
    auto image = [&](const v4f64& M) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) img[(4 * reg + g) * LD + c] = M[reg];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) af[kk] = img[c * LD + 4 * kk + g];
    };
    image(R);
    double prev = diag ? 0.25 : 0.0, z = prev, st = 0.0;   // coordinate a = c of the iterate (alike in the four row groups)
    bool active = true;
    int m = 0;
    while (active && m < 29 && (2 << m) <= max_iter) {
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], R[kk], acc, 0, 0, 0);
      ++m;
      image(acc);
      // z = R^(2^m) r_0 with r_0 = 1/4: a quarter of the sum of the four diagonal-coordinate columns
      z = 0.25 * ((img[c * LD + 0] + img[c * LD + 5]) + (img[c * LD + 10] + img[c * LD + 15]));
      const double inv = fast_rcp(row16_sum(diag ? z : 0.0));
      z *= inv;
      R = acc * inv;                  // 1/tr ~ 1/lambda^(2^m): the squared matrix stays O(1) for tensors that are not isometries
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) af[kk] *= inv;
      const double d = z - prev;
      const double d2 = row16_sum(d * d);       // orthonormal coordinates: the Frobenius distance of the two matrices
      prev = z;
      st = (double)(1 << m);
      active = !(d2 < tol2);
    }
    // hand the result to the evaluation's quad in its own coordinates (plain: Re / Im parts without the sqrt2)
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) strip[lane] = diag ? z : RS2 * z;
    __builtin_amdgcn_wave_barrier();
    if ((lane >> 2) == e) {
      if (m > 0) {                      // (an iteration cap below 3 leaves no room for a round: x stays what the solve gave)
#pragma unroll
        for (int l = 0; l < 4; ++l) x[l] = strip[4 * q + l];
      }
      steps = 1.0 + st;
      left = active;
    }
  }
}

}  // namespace

#ifndef QMPS_DIRECT_MINWAVES
#define QMPS_DIRECT_MINWAVES 2
#endif
// ANS = -1: tensors from HBM;  ANS = QMPS_ANSATZ_* (0, 1, 3): the tensor is built in LDS from the evaluation's ansatz
// parameters - lane q of the quad simulates the three-qubit circuit on the basis state |0>|q>, i.e. column q of the
// unitary, which is all unitary_to_tensor keeps (qmps/tools.py:151-154) - 8 P bytes per evaluation instead of 512.
template <int ANS, bool WARM>
__global__ __launch_bounds__(64, QMPS_DIRECT_MINWAVES) void energy_direct_d4_kernel(LaneArgs p) {
  using Core = DirectD4<QuadOps>;
  constexpr int ROW = 512, PAD = ROW + 16, ITEMS = 16;
  // item stride 528 B = 4 banks: the four items of a ds_read_b128 lane group sit on disjoint banks both when a quad
  // reads one address (rows of the partner index) and when its lanes read their own rows (64 B apart)
  __shared__ __attribute__((aligned(16))) unsigned char lds[ITEMS * PAD];
  __shared__ double sq_img[kSqDoubles];      // the rare path's 16 x 16 image
#ifdef QMPS_GJ_LDS
  __shared__ __attribute__((aligned(16))) double gjbuf[ITEMS * 18];     // 144-byte stride: the 16 quads on disjoint banks
#endif
  const int lane = threadIdx.x, e = lane >> 2, q = lane & 3;
  const int64_t first = (int64_t)blockIdx.x * ITEMS;
  const int64_t b = first + e;
  const bool valid = b < p.B;
  if constexpr (ANS >= 0) {
    Reg<3> reg;
#pragma unroll
    for (int xx = 0; xx < 8; ++xx) {
      reg.re[xx] = (valid && xx == q) ? 1.0 : 0.0;     // an evaluation beyond the batch gets a zero tensor
      reg.im[xx] = 0.0;
    }
    const int nsh = p.ans_nsh;
    const int64_t row = valid ? b : 0;
    const int64_t ridx = nsh > 0 ? row / nsh : row;
    const int shift_k = nsh > 0 ? (int)(row - ridx * nsh) : 0;
    const int isel = nsh > 0 ? *p.ans_i : -1;
    const double* pp = p.ans_params + ridx * p.ans_P;
    // The four lanes of the quad need the same angles: lane q does the sincos of angle l + q (double-precision sincos is
    // ~150 instructions, the gates of a layer ~110), quad_perm hands the results round.  KIND 0 / 1: four angles = two
    // layers per trip; KIND 3: the three angles of one layer.
    constexpr int per = ANS == QMPS_ANSATZ_SHALLOW_CNOT3 ? 3 : 2, trip = ANS == QMPS_ANSATZ_SHALLOW_CNOT3 ? 3 : 4;
    const int P = p.ans_P;
    for (int l = 0; l + per <= P; l += trip) {
      const int mine = l + q;
      double ang = 0.0;
      if (q < trip && mine < P) {
        ang = pp[mine];
        if (mine == isel) ang += roto_shift_value(nsh, shift_k);
      }
      // position of the angle inside its layer: q % per
      const double scale = ansatz_angle_scale<ANS>(ANS == QMPS_ANSATZ_SHALLOW_CNOT3 ? q : (q & 1));
      double sn, cs;
      sincos(scale * ang, &sn, &cs);
      {
        const double c0[3] = {quad_bcast<0>(cs), quad_bcast<1>(cs), quad_bcast<2>(cs)};
        const double s0[3] = {quad_bcast<0>(sn), quad_bcast<1>(sn), quad_bcast<2>(sn)};
        ansatz_layer_cs<3, ANS>(reg, c0, s0);
      }
      if (trip == 4 && l + 2 + per <= P) {
        const double c1[2] = {quad_bcast<2>(cs), quad_bcast<3>(cs)};
        const double s1[2] = {quad_bcast<2>(sn), quad_bcast<3>(sn)};
        ansatz_layer_cs<3, ANS>(reg, c1, s1);
      }
    }
    // A[s][i][j = q] = amplitude[2 i + s]
    double2* tile = (double2*)(lds + e * PAD);
#pragma unroll
    for (int xx = 0; xx < 8; ++xx) tile[((xx & 1) * 4 + (xx >> 1)) * 4 + q] = make_double2(reg.re[xx], reg.im[xx]);
  } else {
    // HBM -> LDS: the wave's 16 tensors are one contiguous 8 KB slab, 16 B per lane per load
    const unsigned char* slab = (const unsigned char*)p.A + first * ROW;
    const int64_t slab_bytes = (p.B - first < ITEMS ? p.B - first : ITEMS) * (int64_t)ROW;
    double2 v[ITEMS * ROW / 1024];
#pragma unroll
    for (int c = 0; c < ITEMS * ROW / 1024; ++c) {
      const int off = c * 1024 + lane * 16;
      v[c] = make_double2(0.0, 0.0);
      if (off < slab_bytes) v[c] = *(const double2*)(slab + off);
    }
#pragma unroll
    for (int c = 0; c < ITEMS * ROW / 1024; ++c) {
      const int off = c * 1024 + lane * 16;
      *(double2*)(lds + (off / ROW) * PAD + (off % ROW)) = v[c];
    }
  }
  __syncthreads();
#ifdef QMPS_GJ_LDS
  const QuadOps o{q, (const double2*)(lds + e * PAD), gjbuf + e * 18};
#else
  const QuadOps o{q, (const double2*)(lds + e * PAD), nullptr};
#endif
  const double tol2 = p.tol * p.tol;

  // ---- environment: direct solve, accepted by one power step ----
  double x[4], us[16];
  double steps = 1.0;
  int status = QMPS_ST_OK;
  bool have = false;            // WARM: the evaluation's resident environment passed the acceptance test as it is
  if constexpr (WARM) {
    // Warm start (environments carried over between evaluations of unchanged / barely changed tensors): one
    // power step from r_in; what it moves by less than tol IS the fixed point to the solvers' criterion - no matrix build,
    // no elimination (7 of the 15 kflop).  The rest of the wave's evaluations go through the direct solve below.
    const double2* rin = (const double2*)p.r_in + (valid ? b : 0) * 16;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double2 rw = rin[q * 4 + l], cl = rin[l * 4 + q];       // r[q][l], r[l][q]
      x[l] = q <= l ? rw.x : cl.y;                                   // Re r[q][l] (q <= l) | Im r[l][q] (q > l)
    }
    Core::normalise(o, x);
    Core::gather(x, us);
    double y[4];
    const double d2 = Core::power_step(o, x, us, y);
    have = valid && d2 < tol2;
#pragma unroll
    for (int l = 0; l < 4; ++l) x[l] = have ? y[l] : x[l];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!WARM || __any(valid && !have)) {
    double Rc[4][16], xs[4], y[4];
    Core::build(o, Rc);
    __builtin_amdgcn_sched_barrier(0);     // phase by phase: keeps the operand reads of later phases out of the register file
    double pivmax;
    Core::solve(o, Rc, xs, pivmax);
    __builtin_amdgcn_sched_barrier(0);
    Core::normalise(o, xs);
    Core::gather(xs, us);
    const double d2 = Core::power_step(o, xs, us, y);
    __builtin_amdgcn_sched_barrier(0);
    const bool accepted = d2 < tol2 && pivmax < Core::kMaxInversePivot;     // NaN (a zero pivot) fails both
    if (!have) {
#pragma unroll
      for (int l = 0; l < 4; ++l) x[l] = xs[l];
      if (WARM) steps = 2.0;               // the rejected warm step + the acceptance step of the solve
    }
    const bool todo = valid && !have && !accepted;
    if (__any(todo)) {
      // rare: not an isometry / degenerate transfer spectrum.  Wave-uniform branch, one evaluation at a time.
      bool left = false;
      squaring_fallback(lds, PAD, sq_img, lane, todo, p.max_iter - 1, tol2, x, steps, left);
      if (__any(todo && left)) {
        // The budget ended the chain between two powers of two (it compares z_(2^m) with z_(2^(m-1)): under max_iter = 10 000 an evaluation the plain
        // method finishes in 4 097 .. 9 998 steps has its fixed point in z_8192 and no comparison left; round 5, as at D = 2): the plain method's own
        // test on the last iterate, one application of T itself, one more iteration.
        double yb[4];
        Core::gather(x, us);
        const double d2b = Core::power_step(o, x, us, yb);
        const bool fin = todo && left && d2b < tol2 && steps + 1.0 <= (double)p.max_iter;
#pragma unroll
        for (int l = 0; l < 4; ++l) x[l] = fin ? yb[l] : x[l];
        if (fin) {
          left = false;
          steps += 1.0;
        }
      }
      if (todo) {
        status = left ? QMPS_ST_NOT_CONVERGED : QMPS_ST_OK;
        if (WARM) steps += 1.0;
      }
    }
    if (WARM || __any(todo)) Core::gather(x, us);
  } else {
    Core::gather(x, us);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- positive definiteness, two-site density matrix, energies ----
  double pre[4][4], pim[4][4];
  const bool pd = Core::density(o, us, pre, pim);
  if (status == QMPS_ST_OK && !pd) status = QMPS_ST_NOT_PD;
  if (p.acc_zero != nullptr && blockIdx.x == 0) acc_clear(p.acc_zero, p.n_terms, lane, 64);   // accumulator of a later step
  for (int t = 0; t < p.n_terms; ++t) {
    const double en = quad_sum(Core::energy((const double*)p.h + 32 * t, pre, pim));
    if (valid && q == 0) p.E[b * p.n_terms + t] = en;
    if (p.partial != nullptr || p.acc != nullptr) {
      const double s = wave_sum((valid && q == 0) ? en : 0.0);
      if (lane == 0) {
        // first pass of the cost reduction: one partial per wave, fixed order
        if (p.partial != nullptr) p.partial[(int64_t)t * gridDim.x + blockIdx.x] = s;
        // ... or the whole reduction: an exact fixed-point sum (order-independent) + the arrival count, ONE atomic
        if (p.acc != nullptr) acc_arrive(p.acc, p.acc_shards, t, blockIdx.x, s, p.acc_bound, p.acc_scale);
      }
    }
  }
  if (p.rho_out != nullptr) {
    double fre[4][4], fim[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = t; s < 4; ++s) {
        fre[t][s] = quad_sum(pre[t][s]);
        fim[t][s] = t == s ? 0.0 : quad_sum(pim[t][s]);
      }
    if (valid && q == 0) {
      double2* o2 = (double2*)p.rho_out + b * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          o2[t * 4 + s] = t <= s ? make_double2(fre[t][s], fim[t][s]) : make_double2(fre[s][t], -fim[s][t]);
    }
  }
  if (!valid) return;
  if (q == 0) {
    p.iters[b] = (int32_t)steps;
    p.status[b] = status;
  }
  if (p.r_out != nullptr) {
    // row q of r: the lane owns u[(q,l)] = x[l]; the transposed coordinate u[(l,q)] comes out of the gathered copy
    double2* o2 = (double2*)p.r_out + b * 16 + q * 4;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double f = q == 0 ? us[4 * l] : (q == 1 ? us[4 * l + 1] : (q == 2 ? us[4 * l + 2] : us[4 * l + 3]));
      const double re = q <= l ? x[l] : f;
      const double im = q == l ? 0.0 : (q < l ? f : -x[l]);
      o2[l] = make_double2(re, im);
    }
  }
}

// ------------------------------------------------------------------------------------------
// energy_only_d4_kernel: the contraction chain A - Abar - h - A - Abar of the north star with the RESIDENT environment
// (qmps_energy_only_launch, qmps_get_rdm): no solve.  Same quad layout and the same density / energy code as the fused
// kernel (one wave = 16 evaluations = one contiguous 8 KB slab of tensors + 4 KB of environments), without the 16 x 16
// system: half the instructions of round 1's two-lanes-per-evaluation kernel.
// r is read as stored (r[i][j] complex), symmetrised by taking Re r[q][l] / Im r[l][q] of the upper triangle, and
// trace-normalised; check_pd: LDL^H pivots, status 0 -> 2.
// ------------------------------------------------------------------------------------------
// LEAN (no density matrix asked for, no positive-definiteness test, one or two Hamiltonian terms): the energy without rho
// (DirectD4::energy_lean: ~30 % fewer instructions for one term).  Measured over 9 rotating batches (432 MiB): 15.5 us per
// 65 536 evaluations = 3.3 TB/s on the 776 B actually read (41 % of the HBM peak); the rho route 18.3 us; round 1's
// two-lanes-per-evaluation kernel 27.9 us.  Three waves per SIMD (168 VGPRs) spill 28 bytes and are SLOWER (18.7 us: any
// scratch use costs more than the third wave brings), four spill 764 bytes (36.8 us): two it is.
#ifndef QMPS_ENERGY_ONLY_LEAN_WAVES
#define QMPS_ENERGY_ONLY_LEAN_WAVES 2
#endif
template <bool LEAN>
__global__ __launch_bounds__(64, LEAN ? QMPS_ENERGY_ONLY_LEAN_WAVES : 2) void energy_only_d4_kernel(LaneArgs p) {
  using Core = DirectD4<QuadOps>;
  constexpr int ROW = 512, PAD = ROW + 16, ITEMS = 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[ITEMS * PAD];
  const int lane = threadIdx.x, e = lane >> 2, q = lane & 3;
  const int64_t first = (int64_t)blockIdx.x * ITEMS;
  const int64_t b = first + e;
  const bool valid = b < p.B;
  if (p.acc_zero != nullptr && blockIdx.x == 0) acc_clear(p.acc_zero, p.n_terms, lane, 64);
  {
    const unsigned char* slab = (const unsigned char*)p.A + first * ROW;
    const int64_t slab_bytes = (p.B - first < ITEMS ? p.B - first : ITEMS) * (int64_t)ROW;
    double2 v[ITEMS * ROW / 1024];
#pragma unroll
    for (int c = 0; c < ITEMS * ROW / 1024; ++c) {
      const int off = c * 1024 + lane * 16;
      v[c] = make_double2(0.0, 0.0);
      if (off < slab_bytes) v[c] = *(const double2*)(slab + off);
    }
    // the environment: lane q takes Re r[q][l] (q <= l) | Im r[l][q] (q > l), l = 0 .. 3
    double x[4];
    {
      const double2* rin = (const double2*)p.r_in + (valid ? b : 0) * 16;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const double2 rw = rin[q * 4 + l], cl = rin[l * 4 + q];
        x[l] = q <= l ? rw.x : cl.y;
      }
    }
#pragma unroll
    for (int c = 0; c < ITEMS * ROW / 1024; ++c) {
      const int off = c * 1024 + lane * 16;
      *(double2*)(lds + (off / ROW) * PAD + (off % ROW)) = v[c];
    }
    __syncthreads();
    const QuadOps o{q, (const double2*)(lds + e * PAD), nullptr};
    Core::normalise(o, x);
    double us[16];
    double pre[4][4], pim[4][4], bre[4][4], bim[4][4];
    bool pd = true;
    if constexpr (LEAN) {
      Core::b_rows(o, bre, bim);       // (before the gather: only the lane's four coordinates of r are live meanwhile)
      Core::gather(x, us);
    } else {
      Core::gather(x, us);
      pd = Core::density(o, us, pre, pim);
    }
    int status = QMPS_ST_OK;
    if (p.check_pd) {
      status = valid ? p.status[b] : QMPS_ST_OK;
      if (status == QMPS_ST_OK && !pd) status = QMPS_ST_NOT_PD;
    }
    for (int t = 0; t < p.n_terms; ++t) {
      double en;
      if constexpr (LEAN) en = quad_sum(Core::energy_lean(bre, bim, us, (const double*)p.h + 32 * t));
      else en = quad_sum(Core::energy((const double*)p.h + 32 * t, pre, pim));
      if (valid && q == 0) p.E[b * p.n_terms + t] = en;
      if (p.partial != nullptr || p.acc != nullptr) {
        const double s = wave_sum((valid && q == 0) ? en : 0.0);
        if (lane == 0) {
          if (p.partial != nullptr) p.partial[(int64_t)t * gridDim.x + blockIdx.x] = s;
          if (p.acc != nullptr) acc_arrive(p.acc, p.acc_shards, t, blockIdx.x, s, p.acc_bound, p.acc_scale);
        }
      }
    }
    if (!LEAN && p.rho_out != nullptr) {
      double fre[4][4], fim[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s2 = t; s2 < 4; ++s2) {
          fre[t][s2] = quad_sum(pre[t][s2]);
          fim[t][s2] = t == s2 ? 0.0 : quad_sum(pim[t][s2]);
        }
      if (valid && q == 0) {
        double2* o2 = (double2*)p.rho_out + b * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
            o2[t * 4 + s2] = t <= s2 ? make_double2(fre[t][s2], fim[t][s2]) : make_double2(fre[s2][t], -fim[s2][t]);
      }
    }
    if (p.check_pd && valid && q == 0) p.status[b] = status;
  }
}

// ------------------------------------------------------------------------------------------
// env_power_d4_kernel: the PLAIN normalised power iteration r <- T(r) / tr T(r), T(r) = sum_s A_s r A_s^+ (QMPS_ENV_POWER: the classical
// statement of the reference's PowerCircuit / krylov route, qmps/represent.py:235-248, and the "power-iteration environment solve" of
// BASELINE.json configs[2]) at D = 4: a 16-LANE DPP ROW PER EVALUATION, four evaluations per wave, PERSISTENT WAVES that draw their
// evaluations from a counter in HBM.
//   * In the real coordinates u of the Hermitian matrix r (qmps_direct_core.h: a = 4 i + i') the map is a real 16 x 16 matrix R.  Lane
//     c of the row builds row c of R once per evaluation (100 multiply-adds, the tensor tile in LDS) and keeps it in registers; a power
//     step is then the mat-vec u' = R u - SIXTEEN v_fmac_f64_dpp row_newbcast (lane b's coordinate straight into the multiply-add: no
//     LDS, no shuffle instruction) - the trace and the distance ||u'/tr - u||_F as DPP row reductions: ~60 instructions per step of four
//     evaluations, 256 real multiply-adds per evaluation and step where the operator form A r A^+ costs 960.
//   * The iteration count of a Haar tensor is geometric-ish (mean 105, 99.9 % 396, max ~900 of 65 536).  A lane per evaluation
//     (energy_lane_kernel<4>, rounds 1-5) holds 63 lanes of a wave idle behind its slowest evaluation, and the whole launch behind ONE
//     (964 steps x 1.7 us of a lane's 1 000-instruction step: 1.65 ms per 65 536 evaluations, 0.12 of the FP64 roofline).  Here a row
//     whose evaluation has converged stores its environment and draws the next one (wave-aggregated: one atomic per refill round), and a
//     straggler costs its step latency (~0.1 us), not a lane's.
// Same iterate sequence as the lane kernel up to rounding (r_0 = 1/D or the caller's guess, symmetrised and trace-normalised; stop when
// ||r_k - r_(k-1)||_F < tol; iterations = k; status 1 at max_iter).  The energies, the positive-definiteness test and the cost sums
// follow in energy_only_d4_kernel (check_pd) on the stored environments.
// ------------------------------------------------------------------------------------------
#ifndef QMPS_POWER_D4_MINWAVES
#define QMPS_POWER_D4_MINWAVES 4
#endif
// acc += (u of lane B of the own 16-lane row) * m
// FRESH: u may have been written by the VALU instruction just before (a DPP read then wants two wait states; inline assembly is invisible to the
// compiler's hazard recogniser, so the s_nop travels inside the same statement)
template <int B, bool FRESH = false>
__device__ __forceinline__ void row_fmac(double& acc, double u, double m) {
  if constexpr (FRESH) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(m), "n"(B));
  else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(m), "n"(B));
}
template <bool GUESS>
__global__ __launch_bounds__(64, QMPS_POWER_D4_MINWAVES) void env_power_d4_kernel(LaneArgs p, int* __restrict__ counter, int first_block) {
  constexpr int PAD = 512 + 16, SLOTS = 4, kChunk = 8;
  __shared__ __attribute__((aligned(16))) unsigned char lds[SLOTS * PAD];
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15, i = c >> 2, ip = c & 3;
  const bool diag = i == ip, rot = i > ip;
  const double wt = diag ? 1.0 : 2.0;                                  // ||r||_F^2 = sum_diag u^2 + 2 sum_offdiag u^2
  const double tol2 = p.tol * p.tol;
  const unsigned long long below = (1ull << (lane & ~15)) - 1ull;      // the lanes of the rows in front of this one
  const double2* tile = (const double2*)(lds + g * PAD);               // A_s[a][j] = tile[(4 s + a) 4 + j]
  int64_t b = 0;
  // the wave's range (wave-uniform): its static first block, then chunks from the counter, which counts from dyn0 = gridDim.x first_block
  const int64_t dyn0 = (int64_t)gridDim.x * first_block;
  int64_t next = (int64_t)blockIdx.x * first_block, end = next + first_block;
  if (next > p.B) next = p.B;
  if (end > p.B) end = p.B;
  bool active = false, exhausted = dyn0 >= p.B;
  bool refill = true;                      // wave-uniform: a row is idle and there may be work for it
  int iters = 0;
  double u = 0.0, up = 0.0, R[16];        // iterate k, iterate k - 1, the row of the map
#pragma unroll
  for (int k = 0; k < 16; ++k) R[k] = 0.0;
  for (;;) {
    if (refill) {
      // ---- refill (the rare path: ~1 % of the trips): every idle row of the wave takes the next evaluation of the wave's own RANGE [next, end) -
      // at first a static block of `first_block` evaluations per wave, then chunks of kChunk drawn from the counter: same-address atomics pass
      // through the L2 at ~16 ns each, and one per refill round (a row at a time: ~60 000 per launch) WAS the launch: 0.83 ms whatever the occupancy
      const unsigned long long idle = __ballot(!active);
      if (idle != 0ull && !(exhausted && next >= end)) {
        if (next >= end && !exhausted) {
          int base = 0;
          if (lane == 0) base = atomicAdd(counter, kChunk);
          const int64_t first = dyn0 + (int64_t)__builtin_amdgcn_readfirstlane(base);
          next = first < p.B ? first : p.B;
          end = first + kChunk < p.B ? first + kChunk : p.B;
          exhausted = first + kChunk >= p.B;
        }
        const int64_t nb = next + (__popcll(idle & below) >> 4);
        const int64_t lim = end;
        {
          const int64_t take = (int64_t)(__popcll(idle) >> 4);
          next = next + take < end ? next + take : end;
        }
        if (!active && nb < lim) {
          b = nb;
          {
            const double2* src = (const double2*)p.A + b * 32;
            const double2 v0 = src[c], v1 = src[c + 16];
            double2* w = (double2*)(lds + g * PAD);
            w[c] = v0;
            w[c + 16] = v1;
          }
          __builtin_amdgcn_wave_barrier();          // (the tile is private to the row; LDS is in order per wave)
          // row c = (i, i') of the real transfer matrix (DirectD4::build for one row): with P(j,j') = sum_s (gamma A_s[i][j]) conj(A_s[i'][j']),
          // gamma = 1 (i <= i') | i (i > i'):  column (j,j): Re P(j,j);  (lo,hi): Re (P(lo,hi) + P(hi,lo));  (hi,lo): -Im (P(lo,hi) - P(hi,lo))
          double tr[2][4], ti[2][4], br[2][4], bi[2][4];
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const double2 a = tile[(4 * s + i) * 4 + j], cc = tile[(4 * s + ip) * 4 + j];
              tr[s][j] = rot ? -a.y : a.x;
              ti[s][j] = rot ? a.x : a.y;
              br[s][j] = cc.x;
              bi[s][j] = cc.y;
            }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            double v = tr[0][j] * br[0][j];
            v = dfma(ti[0][j], bi[0][j], v);
            v = dfma(tr[1][j], br[1][j], v);
            v = dfma(ti[1][j], bi[1][j], v);
            R[5 * j] = v;
          }
#pragma unroll
          for (int lo = 0; lo < 4; ++lo)
#pragma unroll
            for (int hi = lo + 1; hi < 4; ++hi) {
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
              R[4 * lo + hi] = re;
              R[4 * hi + lo] = im;
            }
          // start vector: 1/D, or the caller's guess (qmps_set_env_guess) symmetrised and trace-normalised - zeros / NaN / nothing stored: 1/D
          u = diag ? 0.25 : 0.0;
          if constexpr (GUESS) {
            const double2* rin = (const double2*)p.r_in + b * 16;
            const double2 rw = rin[4 * i + ip], cl = rin[4 * ip + i];       // r[i][i'], r[i'][i]
            const double ug = i <= ip ? 0.5 * (rw.x + cl.x) : 0.5 * (cl.y - rw.y);   // Re r[i][i'] (i <= i') | Im r[i'][i] (i > i')
            const double t = row16_sum(diag ? ug : 0.0);
            const bool usable = t > 1e-300 && t < 1e300;
            if (usable) u = ug * (1.0 / t);
          }
          up = u;
          iters = 0;
          active = true;
        }
      }
      const unsigned long long still = __ballot(!active);
      if (still == ~0ull) break;                                        // nothing left for this wave
      refill = still != 0ull && !(exhausted && next >= end);            // (a range that ran out in the middle of a round: the rest next trip)
    }
    // ---- one power step of every row - straight-line code: idle rows compute on stale registers and nothing of theirs is committed.  The
    // convergence test of iterate k (its distance to iterate k - 1: a DPP row reduction) is issued BESIDE the step that produces iterate
    // k + 1, not behind it (one surplus step per evaluation), and the only branch of a trip asks whether ANY row of the wave has finished:
    // with a branch per decision a lone wave's step was 0.24 us - ten VALU -> SGPR -> branch round trips, twice the arithmetic.
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
    row_fmac<0, true>(y0, u, R[0]); row_fmac<1>(y1, u, R[1]);   row_fmac<2>(y2, u, R[2]);   row_fmac<3>(y3, u, R[3]);
    row_fmac<4>(y0, u, R[4]);       row_fmac<5>(y1, u, R[5]);   row_fmac<6>(y2, u, R[6]);   row_fmac<7>(y3, u, R[7]);
    row_fmac<8>(y0, u, R[8]);       row_fmac<9>(y1, u, R[9]);   row_fmac<10>(y2, u, R[10]); row_fmac<11>(y3, u, R[11]);
    row_fmac<12>(y0, u, R[12]);     row_fmac<13>(y1, u, R[13]); row_fmac<14>(y2, u, R[14]); row_fmac<15>(y3, u, R[15]);
    const double d = u - up;
    const double d2 = row16_sum(wt * d * d);       // ||r_k - r_(k-1)||_F^2
    const double y = (y0 + y1) + (y2 + y3);
    double t0 = 0.0, t1 = 0.0;                     // the trace: the diagonal coordinates sit in lanes 0, 5, 10, 15 of the row
    row_fmac<0, true>(t0, y, 1.0); row_fmac<5>(t1, y, 1.0); row_fmac<10>(t0, y, 1.0); row_fmac<15>(t1, y, 1.0);
    const double yn = y * fast_rcp(t0 + t1);
    const bool conv = iters > 0 && d2 < tol2;
    const bool fin = active && (conv || iters >= p.max_iter);
    if (__any(fin)) {
      if (fin) {
        if (c == 0) {
          p.iters[b] = iters;
          p.status[b] = conv ? QMPS_ST_OK : QMPS_ST_NOT_CONVERGED;
        }
        // r[i][i'] from the coordinates (i, i') (own) and (i', i) (lane 4 i' + i of the row; the sixteen lanes leave together)
        const double ut = __shfl(u, (lane & ~15) + 4 * ip + i, 64);
        const double re = i <= ip ? u : ut;
        const double im = diag ? 0.0 : (i < ip ? ut : -u);
        ((double2*)p.r_out)[b * 16 + c] = make_double2(re, im);
        active = false;
      }
      refill = true;
    }
    iters = active ? iters + 1 : iters;
    up = active ? u : up;
    u = active ? yn : u;
  }
}

// waves: the persistent grid (waves of four evaluations; the caller sizes it to the chip); counter: an int in HBM, zero at launch
hipError_t launch_env_power_d4(const LaneArgs& a, int* counter, int waves, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  if (a.r_out == nullptr || counter == nullptr) return hipErrorInvalidValue;
  const int64_t need = (a.B + 3) / 4;
  const int64_t nw = need < waves ? need : (waves < 1 ? 1 : waves);
  const dim3 grid((unsigned)nw), block(64);
  // half of the batch as static first blocks (whole rounds of four rows), the rest in chunks from the counter
  // (swept on 65 536 Haar tensors - waves per SIMD 2 .. 8 x first block 4 / 8 / 16 x chunk 4 / 8 / 16: 0.35 .. 0.45 ms without a pattern: what differs
  // is when the launch's slowest evaluation happens to start; profiles/experiments/r06/power_d4_sched.py)
  int first_block = (int)(a.B / (2 * nw) / 4 * 4);
  if (first_block < 4) first_block = 4;
  if (a.r_in != nullptr) hipLaunchKernelGGL(env_power_d4_kernel<true>, grid, block, 0, st, a, counter, first_block);
  else hipLaunchKernelGGL(env_power_d4_kernel<false>, grid, block, 0, st, a, counter, first_block);
  return hipGetLastError();
}

hipError_t launch_energy_only_d4(const LaneArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const dim3 grid((unsigned)((a.B + 15) / 16)), block(64);
  if (a.rho_out == nullptr && !a.check_pd && a.n_terms <= 2) hipLaunchKernelGGL(energy_only_d4_kernel<true>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(energy_only_d4_kernel<false>, grid, block, 0, st, a);
  return hipGetLastError();
}

// acc (fixed point + arrival counts) -> cost[t].  One wave.  With expect > 0 the producer may still be running on
// another stream: sweep the shards (agent-scope loads) until `expect` waves per term have arrived, sleeping in between;
// a bounded number of sweeps, then NaN + *err (a reader never hangs the GPU).  The integer sum of the shards is exact.
__global__ __launch_bounds__(64) void cost_finish_kernel(const long long* __restrict__ acc, int n_shards, long long expect,
                                                         int max_polls, double inv_scale, int n_terms,
                                                         double* __restrict__ cost, int* __restrict__ err) {
  const int lane = threadIdx.x;
  for (int t = 0; t < n_terms; ++t) {
    // per-lane sums as doubles (exact: counts and the two halves of the fixed-point values stay far below 2^53), reduced
    // with the DPP / permlane wave sum: the kernel keeps to 32 VGPRs, so that it fits beside two waves of the energy
    // kernel on a SIMD (a resident wave that does not fit costs the energy kernel a wave slot and the step a straggler)
    double cnt = 0.0, hi = 0.0, lo = 0.0;
    for (int poll = 0;; ++poll) {
      long long c_i = 0, hi_i = 0, lo_i = 0;
      for (int i = lane; i < n_shards; i += 64) {
        const long long w = __hip_atomic_load(acc + t * kAccMaxShards + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long c, v;
        acc_decode(w, c, v);
        c_i += c;
        hi_i += v >> 20;                // split so that the sum over the shards stays exact
        lo_i += v & 0xFFFFF;
      }
      cnt = wave_sum((double)c_i);
      hi = wave_sum((double)hi_i);
      lo = wave_sum((double)lo_i);
      if (expect <= 0 || cnt >= (double)expect || poll >= max_polls) break;
      __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) {
      const bool ok = expect <= 0 || cnt == (double)expect;
      const double over = __hip_atomic_load((const double*)(acc + kAccOver) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cost[t] = ok ? (hi * 1048576.0 + lo) * inv_scale + over : __builtin_nan("");
      if (!ok && err != nullptr) *err = 1;
    }
  }
}

hipError_t launch_cost_finish(const long long* acc, int n_shards, long long expect, int max_polls, double inv_scale,
                              int n_terms, double* cost, int* err, hipStream_t st) {
  hipLaunchKernelGGL(cost_finish_kernel, dim3(1), dim3(64), 0, st, acc, n_shards, expect, max_polls, inv_scale, n_terms, cost, err);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// env_direct_d8_kernel: D = 8 direct fixed-point solve, ONE WAVE PER EVALUATION, one ROW of the real 64 x 64 system
// per lane.  Lane a = 8 i + i' owns coordinate u[(i,i')] (r_ii | Re r_ii' for i < i' | Im r_i'i for i > i') and row a
// of R - 1 (+ the trace functional on the last pivot row, right-hand side e_63, as at D = 4): 64 doubles in
// registers.  Gauss-Jordan step k: what is left of lane k's row is broadcast through scalar registers
// (v_readlane_b32) and every lane eliminates column k from its own row - 2080 FMAs per lane in all, no
// pivoting (measured on Haar tensors: smallest pivot 0.37, residual 1e-15).  The result is written as the
// environment r[8][8]; energy_block_kernel<8, true> then starts its power iteration from it, so its first step is
// the acceptance test (iterations = 1) and its loop the fall-back for anything the solve got wrong; a non-finite
// result (zero pivot) is replaced by the default start 1/8.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void env_direct_d8_kernel(const double2* __restrict__ A, double2* __restrict__ r_out, int64_t B) {
  constexpr int D = 8, N = 64, P = D + 1;
  __shared__ double2 sA[2][D][P];
  __shared__ double sT[D][P];
  __shared__ double sM[N][17];
  const int lane = threadIdx.x, i = lane >> 3, ip = lane & 7;
  const int64_t b = blockIdx.x;
  if (b >= B) return;
  {
    const double2* a = A + b * (2 * N);
    sA[0][i][ip] = a[lane];
    sA[1][i][ip] = a[N + lane];
  }
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  r_out[b * N + lane] = env_direct_d8_solve(sA, sT, sM, lane);
}

hipError_t launch_env_direct_d8(const void* A, void* r_out, int64_t B, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(env_direct_d8_kernel, dim3((unsigned)B), dim3(64), 0, st, (const double2*)A, (double2*)r_out, B);
  return hipGetLastError();
}

hipError_t launch_energy_direct_d4(const LaneArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const dim3 grid((unsigned)((a.B + 15) / 16)), block(64);
  const bool warm = a.r_in != nullptr;
  if (a.ans_params == nullptr) {
    if (warm) hipLaunchKernelGGL((energy_direct_d4_kernel<-1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((energy_direct_d4_kernel<-1, false>), grid, block, 0, st, a);
  } else if (warm) {
    switch (a.ans_kind) {
      case QMPS_ANSATZ_SHALLOW_CNOT: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_CNOT, true>), grid, block, 0, st, a); break;
      case QMPS_ANSATZ_SHALLOW_QAOA: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_QAOA, true>), grid, block, 0, st, a); break;
      case QMPS_ANSATZ_SHALLOW_CNOT3: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_CNOT3, true>), grid, block, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (a.ans_kind) {
      case QMPS_ANSATZ_SHALLOW_CNOT: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_CNOT, false>), grid, block, 0, st, a); break;
      case QMPS_ANSATZ_SHALLOW_QAOA: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_QAOA, false>), grid, block, 0, st, a); break;
      case QMPS_ANSATZ_SHALLOW_CNOT3: hipLaunchKernelGGL((energy_direct_d4_kernel<QMPS_ANSATZ_SHALLOW_CNOT3, false>), grid, block, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

}  // namespace qmps
