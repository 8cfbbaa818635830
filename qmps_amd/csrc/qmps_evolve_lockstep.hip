// qmps_evolve_lockstep.hip - the optimiser algebra of the lock-step BFGS time evolution at D = 8, 16 on the device (gfx950 only).
//
// What the reference runs per time step is `minimize(obj, params, (A_, WW))` (scripts/loschmidt.py:367-375,
// qmps/new_time_evolve.py:276-292): scipy BFGS, one trajectory, one scalar objective call at a time.  qmps_evolve_bfgs runs T
// trajectories in lock-step; at D = 8, 16 every iteration is one GRADIENT EVALUATION of all active trajectories (the two fixed
// points of each iterate + its 2 P neighbours by the two-sided quotient: qmps_capi_overlap.hip, enqueue_gradient_kernels).  Round 4
// kept the algebra between two evaluations on the host - directions, the Armijo test of the full step, the rank-two update of
// H^-1, the masks, the next batch's parameter rows - which cost a synchronisation, two staged copies and ~40 us of idle device per
// batch (28 % of a time step at 256 trajectories).  The kernel below does that algebra on device-resident x, g, H^-1, so that the
// host only ENQUEUES: evaluation -> step kernel -> evaluation -> step kernel ...  A control word in HBM tells every kernel of a
// chain of iterations whether there is anything left to do - no trajectory active, or a trajectory REJECTED the full step (its
// backtracking ladder is the rare path: the host takes over for that iteration) - so a chain enqueued blindly costs empty
// launches at its tail, not wrong work.
//
// ONE kernel per iteration: `lockstep_step_kernel` finishes iteration k from its evaluation (phase ACCEPT: Armijo test of the full
// step, the step, the BFGS update, the new active set) and, when the lock-step goes on, opens iteration k + 1 (phase DIRECTION:
// d = -H^-1 g, slope, the candidates x + alpha_0 d and the mask of the next evaluation).  MODE_BEGIN replaces the first phase after
// the first evaluation of a time step (f, g from the batch, the active set, H^-1 = 1 unless carried).
//
// The floating-point expressions are those of the host loop (evolve_bfgs_group, which remains the checker and the path of the
// ladder), every sum in the same order and WITHOUT contraction: given the same evaluations both take the same decisions bit for bit.
//
// Layout: one workgroup of 1024 threads; PL = 2^ceil(log2 P) <= 32 lanes per trajectory (lane a owns component a and row a of
// H^-1, kept in registers: the P x P work is O(P) per lane with coalesced rows), 1024 / PL trajectories per pass.  Sums over
// components run in index order in every lane of the group alike, the components fetched by wave shuffles - no LDS, no barrier
// inside a pass - and every load of a pass is issued before its first use: the kernel is a handful of memory round trips.
// First version: a thread per trajectory with the P x P loops inside - 68 us for the accept phase of 256 trajectories, more than the
// host algebra it replaced; second: LDS staging with a barrier per stage and loads where they were needed - 16 .. 33 us.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"

namespace qmps {

namespace {

constexpr int LS_THREADS = 256;       // per workgroup; LS_THREADS / PL trajectories per workgroup and pass
constexpr int LS_MAX_BLOCKS = 64;
constexpr int LS_MAXP = 32;

__device__ __forceinline__ bool st_ok(const LockstepArgs& p, int t) { return p.st[t] == QMPS_ST_OK && p.st[p.T + t] == QMPS_ST_OK; }

// block-wide sum of an int over LS_THREADS threads (result in every thread)
__device__ __forceinline__ int block_sum(int v, int* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const int r = red[0];
  __syncthreads();
  return r;
}

}  // namespace

template <int PL>
__global__ __launch_bounds__(LS_THREADS) void lockstep_step_kernel(LockstepArgs p) {
#pragma clang fp contract(off)
  constexpr int SLOTS = LS_THREADS / PL;
  __shared__ int red[LS_THREADS];
  const int slot = threadIdx.x / PL, a = threadIdx.x % PL;
  const int gbase = (threadIdx.x & 63) & ~(PL - 1);      // first lane of this trajectory's group inside the wave
  const int P = p.P;
  const bool lane_on = a < P;
  const int n_active0 = p.ctl[0], nit0 = p.ctl[2], stop0 = p.ctl[3];
  const int done_step = p.ctl[5];
  const bool speculative = p.mode == 4;
  const bool begin = p.mode == 1 || speculative, open_only = p.mode == 2, after_ladder = p.mode == 3;
  // all components of a per-trajectory vector held one component per lane, into registers: PL shuffles issued back to back (every
  // lane of the wave takes part: no divergence around it).  (First version: a shuffle where a component was needed, inside the
  // sequential sums - ~100 dependent LDS-crossbar round trips per pass, 6 us.)
  auto gather = [&](double v, double (&out)[PL]) {
#pragma unroll
    for (int k = 0; k < PL; ++k) out[k] = __shfl(v, gbase + k, 64);
  };
  // np.abs(g).max() >= gtol, NaN-propagating (false with any NaN) - every lane of the group computes it alike
  auto gmax_of = [&](const double (&gv)[PL], bool& isnan_) {
    double m = 0.0;
    isnan_ = false;
#pragma unroll
    for (int k = 0; k < PL; ++k)
      if (k < P) {
        const double gk = gv[k];
        if (gk != gk) isnan_ = true;
        const double ak = fabs(gk);
        m = ak > m ? ak : m;
      }
    return m;
  };
  auto gmax_ok = [&](const double (&gv)[PL]) {
    bool isnan_;
    const double m = gmax_of(gv, isnan_);
    return !isnan_ && m >= p.gtol;
  };
  // tolerance of the eigen-solves of a gradient whose size is about m (NaN: the tightest)
  auto tol_rule = [&](double m, bool isnan_) {
    const double t = p.tol_rel * m;
    return isnan_ ? p.tol_min : (t < p.tol_min ? p.tol_min : (t > p.tol_max ? p.tol_max : t));
  };
  __syncthreads();
  // ---------------------------------------------------------------------------------------------------------------------------
  // phase 1: ACCEPT (or BEGIN)
  // ---------------------------------------------------------------------------------------------------------------------------
  // mode 3 finishes an iteration that stopped on rejected full steps (their ladder and the gradient at the accepted points have run)
  // mode 4: the head of a time step enqueued behind the previous step's chain - live only if that step has finished
  const bool live = speculative ? done_step == p.step_id - 1 : (begin || open_only || (after_ladder ? stop0 != 0 : !(n_active0 == 0 || stop0 != 0 || nit0 >= p.maxiter)));
  // (nothing to finish: the mask of the next evaluation stays empty - cleared when the lock-step stopped; the launch still takes part
  // in the barrier below, whose arrival count the host's epoch relies on)
  int n_need = 0, n_active = 0;
  const double nan = __builtin_nan("");
  for (int base = blockIdx.x * SLOTS; base < p.T && !open_only && live; base += gridDim.x * SLOTS) {
    const int t = base + slot;
    const bool on = t < p.T;
    const int ts = on ? t : 0;
    const int as = lane_on ? a : 0;
    const int64_t tp = (int64_t)ts * P;
    double* H = p.H + tp * P;
    // ---- every load of the pass up front (independent: one memory round trip instead of one per use)
    const int st0 = p.st[ts], st1 = p.st[p.T + ts];
    const unsigned char act_b = begin ? 1 : p.active[ts];
    const unsigned char need_b = after_ladder ? p.need[ts] : 0;
    const double asel = after_ladder ? p.asel[ts] : 0.0;
    const double fbt = p.fb[ts];
    const double* fn = p.fb + p.T + (int64_t)ts * 2 * P;
    const double fna = fn[as], fnb = fn[P + as];
    const double Ft = begin ? 0.0 : p.F[ts], slp = begin ? 0.0 : p.slope[ts];
    const double dva = begin ? 0.0 : p.Dv[tp + as], gold = begin ? 0.0 : p.G[tp + as], xa = begin ? 0.0 : p.X[tp + as];
    double hrow[PL];
    if (!begin)
#pragma unroll
      for (int b = 0; b < PL; ++b) hrow[b] = (b < P) ? H[as * P + b] : 0.0;
    const bool ok = st0 == QMPS_ST_OK && st1 == QMPS_ST_OK;
    const double gnew = ok ? (fna - fnb) / (2.0 * p.h) : nan;
    if (begin) {
      // f, g of the iterates from the first evaluation of the time step; H^-1 = 1 unless carried
      if (on && lane_on) {
        p.G[tp + a] = gnew;
        if (p.reset_h)
          for (int b = 0; b < P; ++b) H[a * P + b] = a == b ? 1.0 : 0.0;
      }
      double gn_all[PL];
      gather(gnew, gn_all);
      const bool act = gmax_ok(gn_all);
      if (on && a == 0 && p.tol_next != nullptr) {
        bool isn;
        const double m = gmax_of(gn_all, isn);
        p.g0max[t] = isn ? 0.0 : m;                      // (0 -> tol_min)
      }
      if (on && a == 0) {
        const double f = ok ? fbt : nan;
        p.F[t] = f;
        if (p.fh_start) p.fh_start[t] = f;
        p.active[t] = act ? 1 : 0;
        p.need[t] = 0;
        n_active += act ? 1 : 0;
      }
      continue;
    }
    // ---- the Armijo test of the full step (every lane of the group alike)
    // mode 0: the trajectories of this iteration's evaluation; mode 3: the ones that went through the ladder (the others keep their state)
    const bool act_before = on && (after_ladder ? need_b != 0 : act_b != 0);
    const double fs = ok ? fbt : nan;
    const double Ft0 = (fs - fs == 0.0) ? fs : INFINITY;      // isfinite
    const bool accepted = act_before && (after_ladder || Ft0 <= Ft + p.c1 * p.alpha0 * slp);
    const bool need = act_before && !accepted;                 // rejected: the ladder and the gradient at the accepted point come first
    const bool moved = after_ladder ? (act_before && asel != 0.0) : (accepted && Ft0 < Ft);
    const double a0 = after_ladder ? asel : (moved ? p.alpha0 : 0.0);     // (mode 3: the step length the ladder chose, 0 = no rung decreased f)
    // ---- y = g_new - g, s = a0 d, the rank-two update (sums in index order, by every lane of the group alike)
    const double ya = gnew - gold, sa = a0 * dva;
    double y_all[PL], s_all[PL];
    gather(ya, y_all);
    gather(sa, s_all);
    double sy = 0.0, ss = 0.0, yy = 0.0;
#pragma unroll
    for (int k = 0; k < PL; ++k)
      if (k < P) {
        sy += s_all[k] * y_all[k];
        ss += s_all[k] * s_all[k];
        yy += y_all[k] * y_all[k];
      }
    const bool upd = moved && sy > 1e-12 * sqrt(ss * yy) && sy > 0.0;
    const double rho = 1.0 / sy;
    double hya = 0.0;
#pragma unroll
    for (int b = 0; b < PL; ++b)
      if (b < P) hya += hrow[b] * y_all[b];
    double hy_all[PL];
    gather(hya, hy_all);
    double yHy = 0.0;
#pragma unroll
    for (int k = 0; k < PL; ++k)
      if (k < P) yHy += y_all[k] * hy_all[k];
    const double coef = rho * (1.0 + rho * yHy);
    // H' = H - rho (s (Hy)^T + (Hy) s^T) + rho (1 + rho y^T H y) s s^T
    if (on && lane_on && accepted) {
      if (upd)
#pragma unroll
        for (int b = 0; b < PL; ++b)
          if (b < P) H[a * P + b] = hrow[b] - (rho * sa * hy_all[b] + rho * s_all[b] * hya) + coef * sa * s_all[b];
      if (moved) p.G[tp + a] = gnew;
      p.X[tp + a] = xa + sa;
    }
    double gk_all[PL];
    gather(moved ? gnew : gold, gk_all);
    const bool act = moved && gmax_ok(gk_all);
    if (after_ladder && on && a == 0 && !act_before) n_active += act_b != 0 ? 1 : 0;      // (not in the ladder: active as before)
    if (on && a == 0 && act_before) {
      if (need) {
        p.F0[t] = Ft0;                                   // rung 0 of its ladder
        p.need[t] = 1;
        n_need += 1;
        n_active += 1;                                   // (still active until the host has finished its iteration)
      } else {
        if (moved) p.F[t] = fs;                          // (mode 0: finite; mode 3: NaN if the solve at the accepted point failed)
        p.active[t] = act ? 1 : 0;
        n_active += act ? 1 : 0;
      }
    }
  }
  {
    const int both = block_sum((n_need << 16) | n_active, red);       // this workgroup's counts (T < 65 536: checked by the launcher)
    n_need = both >> 16;
    n_active = both & 0xffff;
  }
  // ---- grid barrier: the counts of all workgroups.  Accumulators ctl[8 + 2 e], ctl[9 + 2 e] of parity e = epoch & 1 (zeroed by the
  // previous launch), arrival counter ctl[12] (monotonic: this launch completes it to epoch * gridDim).  Every workgroup has read
  // the control word before it arrives, so workgroup 0 may rewrite it behind the barrier.  (The workgroups are few and small: all
  // resident at once, the spin cannot starve one of them.)
  if (gridDim.x > 1) {
    const int e = p.epoch & 1;
    if (threadIdx.x == 0) {
      if (n_need) atomicAdd(p.ctl + 8 + 2 * e, n_need);
      if (n_active) atomicAdd(p.ctl + 9 + 2 * e, n_active);
      __threadfence();
      atomicAdd(p.ctl + 12, 1);
      const int target = p.epoch * (int)gridDim.x;
      while (__hip_atomic_load(p.ctl + 12, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      red[0] = __hip_atomic_load(p.ctl + 8 + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      red[1] = __hip_atomic_load(p.ctl + 9 + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    n_need = red[0];
    n_active = red[1];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the other parity's accumulators, for the next launch
      p.ctl[8 + 2 * (e ^ 1)] = 0;
      p.ctl[9 + 2 * (e ^ 1)] = 0;
    }
  }
  if (open_only) n_active = n_active0;                   // (the host finished the iteration and wrote the control word)
  const int nit = open_only ? nit0 : (begin ? 0 : (n_need > 0 ? nit0 : nit0 + 1));           // (mode 3: n_need = 0, the iteration is complete)
  const int stop = open_only ? stop0 : (n_need > 0 ? 1 : 0);
  if (blockIdx.x == 0 && threadIdx.x == 0 && !open_only && live) {
    p.ctl[0] = n_active;
    p.ctl[1] = n_need;
    p.ctl[2] = nit;
    p.ctl[3] = stop;
    if (!begin && !after_ladder && nit0 <= p.maxiter) p.ctl[16 + p.hist_off + nit0] = n_need;      // the pattern of rejections of this time step
    if (!(n_active > 0 && stop == 0 && nit < p.maxiter) && stop == 0) {      // the time step ends here
      p.ctl[5] = p.step_id;
      p.ctl[6] = nit;
    }
  }
  // ---------------------------------------------------------------------------------------------------------------------------
  // phase 2: DIRECTION of the next iteration (or an empty mask when the lock-step stops here)
  // ---------------------------------------------------------------------------------------------------------------------------
  if (!live) return;
  const bool go_on = n_active > 0 && stop == 0 && nit < p.maxiter;
  const bool finished = !go_on && stop == 0;
  if (begin && blockIdx.x == 0)      // a new time step: its pattern of rejections starts empty
    for (int k = threadIdx.x; k <= p.maxiter + 1; k += blockDim.x) p.ctl[16 + p.hist_off + k] = 0;
  for (int base = blockIdx.x * SLOTS; base < p.T; base += gridDim.x * SLOTS) {
    const int t = base + slot;
    const bool on = t < p.T;
    const int ts = on ? t : 0;
    const int as = lane_on ? a : 0;
    const int64_t tp = (int64_t)ts * P;
    double* H = p.H + tp * P;
    const double xa = p.X[tp + as], ga = p.G[tp + as], Ft = p.F[ts];
    const bool act = on && p.active[ts] != 0;
    double hrow[PL];
    if (go_on)
#pragma unroll
      for (int b = 0; b < PL; ++b) hrow[b] = (b < P) ? H[as * P + b] : 0.0;
    // the record of the time step so far (final when the lock-step stops here; a trajectory waiting for the host's ladder is
    // recorded again by the mode-2 launch that follows the host's update)
    if (on && lane_on) p.ph[tp + a] = xa;
    if (on && a == 0) {
      p.fh_end[t] = Ft;
      p.head_mask[t] = finished ? 1 : 0;
    }
    if (p.tol_next != nullptr) {
      double gt_all[PL];
      gather(ga, gt_all);
      bool isn;
      const double m = gmax_of(gt_all, isn);
      if (on && a == 0) {
        if (go_on) p.tol_next[t] = tol_rule(m, isn);                       // the next evaluation of this time step
        else if (stop == 0) p.tol_next[t] = tol_rule(p.g0max[t], false);   // the first evaluation of the next time step
        // (stopped on rejected steps: the gradient at the accepted points keeps this iteration's tolerances)
      }
    }
    if (!go_on) {
      if (on && a == 0) p.eff[t] = 0;
      continue;
    }
    // (a trajectory that has stopped keeps g and H^-1: its direction test had its one possible effect in iteration 0)
    const bool compute = on && (act || nit == 0);
    double g_all[PL];
    gather(ga, g_all);
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < PL; ++b)
      if (b < P) acc += hrow[b] * g_all[b];
    double dv = -acc;
    double d_all[PL];
    gather(dv, d_all);
    double sl = 0.0, sl_sd = 0.0;                        // slope along d; slope of steepest descent
#pragma unroll
    for (int k = 0; k < PL; ++k)
      if (k < P) {
        sl += g_all[k] * d_all[k];
        sl_sd -= g_all[k] * g_all[k];
      }
    const bool restart = !(sl < 0.0);                    // not a descent direction: restart from steepest descent
    if (restart) {
      sl = sl_sd;
      dv = -ga;
    }
    if (compute && lane_on && restart)
      for (int b = 0; b < P; ++b) H[a * P + b] = a == b ? 1.0 : 0.0;
    if (compute && a == 0) p.slope[t] = sl;
    if (!compute || !act) dv = 0.0;
    if (on && lane_on) {
      p.Dv[tp + a] = dv;
      p.Xc[tp + a] = xa + p.alpha0 * dv;
    }
    if (on && a == 0) {
      p.eff[t] = act ? 1 : 0;
      p.need[t] = 0;
    }
  }
}

// The ladder of an iteration that stopped on rejected full steps.  Candidates x + alphas[r + 1] d of EVERY trajectory (the batch is
// trajectory-major; the solves are masked by `need`), as the host loop builds them.
__global__ __launch_bounds__(256) void lockstep_ladder_cand_kernel(LockstepArgs p) {
#pragma clang fp contract(off)
  if (p.ctl[3] == 0) return;                             // (enqueued blindly: nothing was rejected)
  const int G = p.NA - 1;
  const int64_t n = (int64_t)p.T * G * p.P;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
    const int k = (int)(q % p.P);
    const int64_t tr = q / p.P;
    const int r = (int)(tr % G);
    const int64_t t = tr / G;
    p.cand[q] = p.X[t * p.P + k] + p.alphas[r + 1] * p.Dv[t * p.P + k];
  }
}

// ... and its verdict: the first rung with the Armijo decrease, else the best one if it decreases f, else no step (the host loop's
// rule); the accepted point goes into the parameter rows of the gradient evaluation that follows.
__global__ __launch_bounds__(256) void lockstep_ladder_pick_kernel(LockstepArgs p, const double* __restrict__ fl, const int32_t* __restrict__ stl) {
#pragma clang fp contract(off)
  if (p.ctl[3] == 0) return;
  const int G = p.NA - 1;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < p.T; t += gridDim.x * 256) {
    double a = 0.0;
    if (p.need[t] != 0) {
      const double f = p.F[t], sl = p.slope[t];
      int first = -1, best = 0;
      double Fbest = p.F0[t], Ffirst = 0.0;
      for (int r = 0; r < p.NA; ++r) {
        double Fr;
        if (r == 0) Fr = p.F0[t];
        else {
          const double v = fl[(int64_t)t * G + r - 1];
          Fr = (overlap_usable(stl[(int64_t)t * G + r - 1]) && v - v == 0.0) ? v : INFINITY;
        }
        if (first < 0 && Fr <= f + p.c1 * p.alphas[r] * sl) { first = r; Ffirst = Fr; }
        if (Fr < Fbest) { best = r; Fbest = Fr; }
      }
      if (first < 0) { first = best; Ffirst = Fbest; }
      if (Ffirst < f) a = p.alphas[first];
    }
    p.asel[t] = a;
    for (int k = 0; k < p.P; ++k) p.Xc[(int64_t)t * p.P + k] = p.X[(int64_t)t * p.P + k] + a * p.Dv[(int64_t)t * p.P + k];
  }
}

hipError_t launch_lockstep_ladder_cand(const LockstepArgs& a, hipStream_t st) {
  const int64_t n = (int64_t)a.T * (a.NA - 1) * a.P;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(lockstep_ladder_cand_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_lockstep_ladder_pick(const LockstepArgs& a, const double* fl, const int32_t* stl, hipStream_t st) {
  hipLaunchKernelGGL(lockstep_ladder_pick_kernel, dim3((unsigned)((a.T + 255) / 256)), dim3(256), 0, st, a, fl, stl);
  return hipGetLastError();
}

hipError_t launch_lockstep_step(const LockstepArgs& a, hipStream_t st) {
  if (a.P < 1 || a.P > LS_MAXP || a.T < 1 || a.T > 65535 || a.epoch < 1) return hipErrorInvalidValue;
  const int PL = a.P <= 4 ? 4 : (a.P <= 8 ? 8 : (a.P <= 16 ? 16 : 32));
  const int slots = LS_THREADS / PL;
  int nb = (a.T + slots - 1) / slots;
  nb = nb > LS_MAX_BLOCKS ? LS_MAX_BLOCKS : nb;
  if (nb != a.blocks) return hipErrorInvalidValue;       // (the host's epoch arithmetic assumes this grid: lockstep_step_blocks)
  if (PL == 4) hipLaunchKernelGGL(lockstep_step_kernel<4>, dim3(nb), dim3(LS_THREADS), 0, st, a);
  else if (PL == 8) hipLaunchKernelGGL(lockstep_step_kernel<8>, dim3(nb), dim3(LS_THREADS), 0, st, a);
  else if (PL == 16) hipLaunchKernelGGL(lockstep_step_kernel<16>, dim3(nb), dim3(LS_THREADS), 0, st, a);
  else hipLaunchKernelGGL(lockstep_step_kernel<32>, dim3(nb), dim3(LS_THREADS), 0, st, a);
  return hipGetLastError();
}
int lockstep_step_blocks(int T, int P) {
  const int PL = P <= 4 ? 4 : (P <= 8 ? 8 : (P <= 16 ? 16 : 32));
  const int slots = LS_THREADS / PL;
  const int nb = (T + slots - 1) / slots;
  return nb > LS_MAX_BLOCKS ? LS_MAX_BLOCKS : nb;
}

}  // namespace qmps
