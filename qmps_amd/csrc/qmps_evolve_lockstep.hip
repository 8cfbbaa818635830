// qmps_evolve_lockstep.hip - the optimiser algebra of the lock-step BFGS time evolution at D = 8, 16 on the device (gfx950 only).
//
// What the reference runs per time step is `minimize(obj, params, (A_, WW))` (scripts/loschmidt.py:367-375,
// qmps/new_time_evolve.py:276-292): scipy BFGS, one trajectory, one scalar objective call at a time.  qmps_evolve_bfgs runs T
// trajectories in lock-step; at D = 8, 16 every iteration is one GRADIENT EVALUATION of all active trajectories (the two fixed
// points of each iterate + its 2 P neighbours by the two-sided quotient: qmps_capi_overlap.hip, enqueue_gradient_kernels).  Round 4
// kept the algebra between two evaluations on the host - directions, the Armijo test of the full step, the rank-two update of
// H^-1, the masks, the next batch's parameter rows - which cost a synchronisation, two staged copies and ~40 us of idle device per
// batch (28 % of a time step at 256 trajectories).  The three kernels below do that algebra on device-resident x, g, H^-1, so that
// the host only ENQUEUES: begin (after the first evaluation of a time step), then per iteration direction -> [evaluation] -> accept.
// A control word in HBM tells every kernel of a chain of iterations whether there is anything left to do - no trajectory active, or
// a trajectory REJECTED the full step (its backtracking ladder is the rare path: the host takes over for that iteration) - so a
// chain enqueued blindly costs empty launches at its tail, not wrong work.
//
// The floating-point expressions are those of the host loop (evolve_bfgs_group, which remains the checker and the path of the
// ladder), in the same order and WITHOUT contraction: given the same evaluations both take the same decisions bit for bit.
// One workgroup; thread t, t + 256, ... own a trajectory each (the algebra is O(T P^2): microseconds).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"

namespace qmps {

namespace {

__device__ __forceinline__ bool st_ok(const LockstepArgs& p, int t) { return p.st[t] == QMPS_ST_OK && p.st[p.T + t] == QMPS_ST_OK; }

// np.abs(g).max() >= bound, NaN-propagating (false with any NaN) - gmax_at_least of the host loop
__device__ __forceinline__ bool gmax_at_least(const double* g, int P, double bound) {
  double m = 0.0;
  for (int k = 0; k < P; ++k) {
    if (g[k] != g[k]) return false;
    const double a = fabs(g[k]);
    m = a > m ? a : m;
  }
  return m >= bound;
}

// block-wide sum of an int over the 256 threads (result in every thread)
__device__ __forceinline__ int block_sum(int v, int* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const int r = red[0];
  __syncthreads();
  return r;
}

}  // namespace

// After the FIRST evaluation of a time step (the iterates are the previous step's minimisers, the references their tensors):
// f, g from the batch, the active set, the control word, the record of the objective at the start of the step.
__global__ __launch_bounds__(256) void lockstep_begin_kernel(LockstepArgs p) {
#pragma clang fp contract(off)
  __shared__ int red[256];
  int n_active = 0;
  for (int t = threadIdx.x; t < p.T; t += 256) {
    const bool ok = st_ok(p, t);
    const double nan = __builtin_nan("");
    const double* fn = p.fb + p.T + (int64_t)t * 2 * p.P;
    double* g = p.G + (int64_t)t * p.P;
    const double f = ok ? p.fb[t] : nan;
    for (int k = 0; k < p.P; ++k) g[k] = ok ? (fn[k] - fn[p.P + k]) / (2.0 * p.h) : nan;
    p.F[t] = f;
    if (p.fh_start) p.fh_start[t] = f;
    if (p.reset_h) {
      double* H = p.H + (int64_t)t * p.P * p.P;
      for (int a = 0; a < p.P; ++a)
        for (int b = 0; b < p.P; ++b) H[a * p.P + b] = a == b ? 1.0 : 0.0;
    }
    const bool act = gmax_at_least(g, p.P, p.gtol);
    p.active[t] = act ? 1 : 0;
    p.eff[t] = 0;
    p.need[t] = 0;
    n_active += act ? 1 : 0;
  }
  n_active = block_sum(n_active, red);
  if (threadIdx.x == 0) {
    p.ctl[0] = n_active;
    p.ctl[1] = 0;
    p.ctl[2] = 0;
    p.ctl[3] = 0;
  }
}

// Start of an iteration: d = -H^-1 g (steepest descent and H^-1 = 1 where that is no descent direction), the slope, the candidate
// x + alpha_0 d of every active trajectory (the rows the evaluation builds its tensors from) and the evaluation's mask.
// Nothing to do (mask all zero, nothing touched) when no trajectory is active, the iteration cap is reached, or a rejected full
// step is waiting for the host's ladder.
__global__ __launch_bounds__(256) void lockstep_direction_kernel(LockstepArgs p) {
#pragma clang fp contract(off)
  const int n_active = p.ctl[0], nit = p.ctl[2], stop = p.ctl[3];
  const bool idle = n_active == 0 || stop != 0 || nit >= p.maxiter;
  for (int t = threadIdx.x; t < p.T; t += 256) {
    if (idle) {
      p.eff[t] = 0;
      continue;
    }
    const bool act = p.active[t] != 0;
    double* H = p.H + (int64_t)t * p.P * p.P;
    const double* g = p.G + (int64_t)t * p.P;
    double* d = p.Dv + (int64_t)t * p.P;
    const double* x = p.X + (int64_t)t * p.P;
    double* xc = p.Xc + (int64_t)t * p.P;
    // (a trajectory that has stopped keeps g and H^-1: its direction test below had its one possible effect in iteration 0)
    if (!act && nit > 0) {
      for (int a = 0; a < p.P; ++a) d[a] = 0.0;
    } else {
      double sl = 0.0;
      for (int a = 0; a < p.P; ++a) {
        double acc = 0.0;
        for (int b = 0; b < p.P; ++b) acc += H[a * p.P + b] * g[b];
        d[a] = -acc;
      }
      for (int a = 0; a < p.P; ++a) sl += g[a] * d[a];
      if (!(sl < 0.0)) {                                  // not a descent direction: restart from steepest descent
        for (int a = 0; a < p.P; ++a)
          for (int b = 0; b < p.P; ++b) H[a * p.P + b] = a == b ? 1.0 : 0.0;
        sl = 0.0;
        for (int a = 0; a < p.P; ++a) { d[a] = -g[a]; sl -= g[a] * g[a]; }
      }
      p.slope[t] = sl;
      if (!act) for (int a = 0; a < p.P; ++a) d[a] = 0.0;
    }
    for (int a = 0; a < p.P; ++a) xc[a] = x[a] + p.alpha0 * d[a];
    p.eff[t] = act ? 1 : 0;
    p.need[t] = 0;
  }
}

// End of an iteration whose evaluation was at the full steps: Armijo test; a trajectory that accepts takes the step and the BFGS
// update at once (its values are the batch's); one that rejects is flagged (`need`) and left as it is - then the control word says
// STOP and the host finishes the iteration for the flagged trajectories (ladder, gradient at the accepted point, update).
__global__ __launch_bounds__(256) void lockstep_accept_kernel(LockstepArgs p) {
#pragma clang fp contract(off)
  __shared__ int red[256];
  const int n_active0 = p.ctl[0], nit = p.ctl[2], stop = p.ctl[3];
  __syncthreads();
  if (n_active0 == 0 || stop != 0 || nit >= p.maxiter) return;
  int n_need = 0, n_active = 0;
  for (int t = threadIdx.x; t < p.T; t += 256) {
    if (p.active[t] == 0) continue;                      // (rows of skipped trajectories: their last values; moved = 0, still inactive)
    const bool ok = st_ok(p, t);
    const double nan = __builtin_nan("");
    const double fs = ok ? p.fb[t] : nan;
    const double Ft0 = (fs - fs == 0.0) ? fs : INFINITY;   // isfinite
    const double f = p.F[t];
    if (!(Ft0 <= f + p.c1 * p.alpha0 * p.slope[t])) {    // rejected: the ladder and the gradient at the accepted point are the host's
      p.need[t] = 1;
      n_need += 1;
      n_active += 1;                                      // (still active until the host has finished its iteration)
      continue;
    }
    double* H = p.H + (int64_t)t * p.P * p.P;
    double* g = p.G + (int64_t)t * p.P;
    double* gs = p.Gs + (int64_t)t * p.P;
    double* Hy = p.Hy + (int64_t)t * p.P;
    const double* d = p.Dv + (int64_t)t * p.P;
    double* x = p.X + (int64_t)t * p.P;
    const bool moved = Ft0 < f;
    const double a0 = moved ? p.alpha0 : 0.0;
    if (moved) {
      const double* fn = p.fb + p.T + (int64_t)t * 2 * p.P;
      for (int k = 0; k < p.P; ++k) gs[k] = ok ? (fn[k] - fn[p.P + k]) / (2.0 * p.h) : nan;
      double sy = 0.0, ss = 0.0, yy = 0.0;
      for (int k = 0; k < p.P; ++k) {
        const double s = a0 * d[k], y = gs[k] - g[k];
        sy += s * y;
        ss += s * s;
        yy += y * y;
      }
      if (sy > 1e-12 * sqrt(ss * yy) && sy > 0.0) {
        // H' = H - rho (s (Hy)^T + (Hy) s^T) + rho (1 + rho y^T H y) s s^T
        const double rho = 1.0 / sy;
        double yHy = 0.0;
        for (int a = 0; a < p.P; ++a) {
          double acc = 0.0;
          for (int b = 0; b < p.P; ++b) acc += H[a * p.P + b] * (gs[b] - g[b]);
          Hy[a] = acc;
        }
        for (int a = 0; a < p.P; ++a) yHy += (gs[a] - g[a]) * Hy[a];
        const double coef = rho * (1.0 + rho * yHy);
        for (int a = 0; a < p.P; ++a)
          for (int b = 0; b < p.P; ++b) {
            const double sa = a0 * d[a], sb = a0 * d[b];
            H[a * p.P + b] = H[a * p.P + b] - (rho * sa * Hy[b] + rho * sb * Hy[a]) + coef * sa * sb;
          }
      }
      p.F[t] = fs;
      for (int k = 0; k < p.P; ++k) g[k] = gs[k];
    }
    for (int k = 0; k < p.P; ++k) x[k] = x[k] + a0 * d[k];
    const bool act = moved && gmax_at_least(g, p.P, p.gtol);
    p.active[t] = act ? 1 : 0;
    n_active += act ? 1 : 0;
  }
  n_need = block_sum(n_need, red);
  n_active = block_sum(n_active, red);
  if (threadIdx.x == 0) {
    p.ctl[0] = n_active;
    p.ctl[1] = n_need;
    if (n_need > 0) p.ctl[3] = 1;       // the host completes this iteration (and counts it)
    else p.ctl[2] = nit + 1;
  }
}

hipError_t launch_lockstep_begin(const LockstepArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(lockstep_begin_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_lockstep_direction(const LockstepArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(lockstep_direction_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_lockstep_accept(const LockstepArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(lockstep_accept_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace qmps
