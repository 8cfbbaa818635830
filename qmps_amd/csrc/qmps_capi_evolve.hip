// qmps_capi_evolve.hip - the evolve drivers of the C-ABI (declared in include/qmps_hip.h): the lock-step BFGS time evolution in one C call
// (qmps_evolve_bfgs: host loop, or - D = 8, 16, round 5 - the optimiser algebra in kernels on device-resident state with the host
// enqueueing chains of iterations; lock-step groups), the per-trajectory device-resident optimisers of D = 2, 4
// (qmps_evolve_bfgs_device), the rotosolve time evolution (qmps_evolve_rotosolve), and the versioned option structs in front of them.
// Split out of qmps_capi_overlap.hip in round 5; the overlap launches they are built on: qmps_overlap_internal.h.
#include "qmps_ctx.h"
#include "qmps_overlap_internal.h"

#include <string>
#include <thread>

using namespace qmps_host;

namespace {
// One lock-step group of qmps_evolve_bfgs: T trajectories on context c.  The histories are rows of T_hist trajectories, this
// group's at column t_off (params / hinv already point at the group's rows).  May throw (std::vector): the callers catch.
int evolve_bfgs_group(qmps_ctx* c, int64_t T, int64_t T_hist, int64_t t_off, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                      double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                      double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out) {
  if (int rc = bind(c)) return rc;
  DisarmOneShots disarm{c};      // nothing armed by this driver outlives it, whichever way it ends
  if (!params || !WW || !f_hist || !alphas) return fail(QMPS_ERR_ARG, "null argument");
  if (flags & ~(QMPS_BFGS_CARRY_HESSIAN | QMPS_BFGS_WARM | QMPS_BFGS_TIGHT_GRADIENT | QMPS_BFGS_ADAPTIVE_GRADIENT | QMPS_BFGS_TIME_STEPS)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  const int P = n_params, NA = n_alphas;
  if (NA < 1 || NA > 64) return fail(QMPS_ERR_ARG, "n_alphas outside [1, 64]");
  const int64_t G = NA - 1;
  if (T < 1 || T * (1 + 2 * (int64_t)P) > c->max_batch || T * G > c->max_batch)
    return fail(QMPS_ERR_ARG, "T max(2 n_params + 1, n_alphas - 1) = %lld evaluations exceed max_batch = %lld",
                (long long)(T * ((1 + 2 * (int64_t)P) > G ? (1 + 2 * (int64_t)P) : G)), (long long)c->max_batch);
  if (n_steps < 1 || maxiter < 0 || !(gtol > 0.0) || !(h > 0.0)) return fail(QMPS_ERR_ARG, "bad n_steps / maxiter / gtol / h");
  if (int rc = check_ansatz(c, kind, P)) return rc;
  const bool carry = (flags & QMPS_BFGS_CARRY_HESSIAN) != 0;
  bool warm = (flags & QMPS_BFGS_WARM) != 0;
  const bool two_sided = c->D >= 4;       // D = 2: the 2 P + 1 central-difference candidates are eigen-solved themselves (a lane each)
  if (warm && two_sided && c->grad_warm_T != T) return fail(QMPS_ERR_STATE, "QMPS_BFGS_WARM: the resident fixed points belong to %lld trajectories, not %lld", (long long)c->grad_warm_T, (long long)T);
  const bool squaring = overlap_squares(c);
  const int ladder_rounds = squaring ? (max_rounds > 60 ? 60 : max_rounds) : max_rounds;
  const int grad_rounds = max_rounds > 100000 ? max_rounds : 100000;       // (as _GroupedObjective.value_and_grad)
  // objective by the two-sided quotient (error ~ residual^2): the gradient batches' solves stop at 1e-8 (see qmps_hip.h)
  double grad_tol = (flags & QMPS_BFGS_TIGHT_GRADIENT) ? tol : (tol > 1e-8 ? tol : 1e-8);
  if (const char* e = tuning_knob("QMPS_GRAD_TOL")) grad_tol = atof(e);      // (tuning builds: profiles/EXPERIMENTS.md round 5)
  // QMPS_BFGS_ADAPTIVE_GRADIENT (D = 8, 16): the solves of a trajectory's gradient stop at clamp(1e-3 max|g|, grad_tol, 1e-6), g the
  // trajectory's current gradient (first evaluation of a time step: the gradient the previous step's first evaluation found; first
  // step of a call: grad_tol).  The objective still comes from the two-sided quotient (error ~ residual^2 <= 1e-12, far inside the
  // Armijo margin c1 |slope|: 1e-6 |g|^2 against 1e-4 |g|^2); the gradient carries a relative error <= ~1e-3.
  const bool adaptive = (flags & QMPS_BFGS_ADAPTIVE_GRADIENT) != 0 && (flags & QMPS_BFGS_TIGHT_GRADIENT) == 0 && two_sided && (c->D == 8 || c->D == 16);
  const double tol_min = grad_tol, tol_max = grad_tol > 1e-6 ? grad_tol : 1e-6, tol_rel = 1e-3;
  std::vector<double> tolv(adaptive ? T : 0, tol_min), g0max_prev(adaptive ? T : 0, 0.0);
  auto tol_rule = [&](double m, bool isnan_) {
    const double t = tol_rel * m;
    return isnan_ ? tol_min : (t < tol_min ? tol_min : (t > tol_max ? tol_max : t));
  };
  auto gmax_of = [&](const double* gt, bool& isnan_) {
    double m = 0.0;
    isnan_ = false;
    for (int k = 0; k < P; ++k) {
      if (gt[k] != gt[k]) isnan_ = true;
      const double a = fabs(gt[k]);
      m = a > m ? a : m;
    }
    return m;
  };
  const size_t TP = (size_t)T * P;
  const double nan = __builtin_nan("");
  std::vector<double> X(params, params + TP), Hinv(TP * P), f(T), g(TP), d(TP), slope(T), fs(T), gs(TP), fn(T), gn(TP), Xc(TP), Xn(TP), s(TP), Fc((size_t)T * NA),
      cand, Fl, Hy(P);
  std::vector<int32_t> st(T), stl;
  std::vector<unsigned char> active(T), moved(T), need(T);
  // (a pair of event records around a batch costs the stream ~12 us: only when asked for; restored on EVERY way out of this function)
  Restore<int> period_guard(c->timing_period, counters_out ? 1 : 0);
  Restore<bool> stash_guard(c->stash_masks, true);          // (every batch below ends with a synchronisation)
  double n_grad = 0.0, n_ladder = 0.0, nfev = 0.0, grad_ms = 0.0;
  auto set_identity = [&](int64_t t) {
    double* Ht = &Hinv[(size_t)t * P * P];
    for (int a = 0; a < P; ++a)
      for (int b = 0; b < P; ++b) Ht[a * P + b] = a == b ? 1.0 : 0.0;
  };
  if (carry && warm && hinv) memcpy(Hinv.data(), hinv, TP * P * sizeof(double));
  else for (int64_t t = 0; t < T; ++t) set_identity(t);
  // objective + gradient of a batch of iterates; trajectories with a failed solve come back as NaN (tools.py / new_time_evolve.py)
  std::vector<double> fdc, fdf;
  std::vector<int32_t> fds;
  auto value_and_grad = [&](const double* Z, double* fo, double* go, const unsigned char* mask) -> int {
    if (!two_sided) {
      // tools.batched_fd_gradient: candidate t (2 P + 1) + 0 = the iterate, + 1 + k = +h e_k, + 1 + P + k = -h e_k
      const int64_t G1 = 2 * (int64_t)P + 1;
      fdc.resize((size_t)T * G1 * P);
      fdf.resize((size_t)T * G1);
      fds.resize((size_t)T * G1);
      for (int64_t t = 0; t < T; ++t)
        for (int64_t r = 0; r < G1; ++r)
          for (int k = 0; k < P; ++k)
            fdc[((size_t)t * G1 + r) * P + k] = Z[(size_t)t * P + k] + (r >= 1 && (r - 1) % P == k ? (r <= P ? h : -h) : 0.0);
      if (int e = qmps_overlap_set_group(c, G1)) return e;
      if (mask) { if (int e = qmps_overlap_set_active(c, T, mask)) return e; }
      int e = qmps_overlap_eval_ansatz(c, T * G1, kind, P, fdc.data(), ladder_rounds, tol, 0, fdf.data(), fds.data());
      (void)qmps_overlap_set_group(c, 0);
      if (e) return e;
      for (int64_t t = 0; t < T; ++t) {
        const double* F = &fdf[(size_t)t * G1];
        const int32_t* S = &fds[(size_t)t * G1];
        fo[t] = qmps::overlap_usable(S[0]) ? F[0] : nan;
        for (int k = 0; k < P; ++k)
          go[(size_t)t * P + k] = (qmps::overlap_usable(S[1 + k]) && qmps::overlap_usable(S[1 + P + k])) ? (F[1 + k] - F[1 + P + k]) / (2.0 * h) : nan;
      }
      n_grad += 1.0;
      nfev += (double)T * (2 * P + 1);
      if (counters_out) {
        float ms = 0.f;
        if (qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
      }
      return QMPS_OK;
    }
    if (mask) { if (int e = qmps_overlap_set_active(c, T, mask)) return e; }
    if (adaptive) {
      if (!c->d_tolarr) HIP_TRY(hipMalloc((void**)&c->d_tolarr, (size_t)c->max_batch * sizeof(double)));
      HIP_TRY(hipMemcpyAsync(c->d_tolarr, tolv.data(), (size_t)T * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      c->grad_tol_in = c->d_tolarr;
    }
    if (int e = qmps_overlap_gradient(c, T, kind, P, Z, h, grad_rounds, grad_tol, (warm ? QMPS_OVERLAP_WARM : 0) | QMPS_OVERLAP_TWO_SIDED_F, fo, go, st.data())) return e;
    warm = true;
    for (int64_t t = 0; t < T; ++t)
      if (!qmps::overlap_usable(st[t])) {
        fo[t] = nan;
        for (int k = 0; k < P; ++k) go[(size_t)t * P + k] = nan;
      }
    n_grad += 1.0;
    nfev += (double)T * (2 * P + 1);
    if (counters_out) {
      float ms = 0.f;
      if (qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
    }
    return QMPS_OK;
  };
  auto gmax_at_least = [&](const double* gt, double bound) {       // np.abs(g).max() >= bound, NaN-propagating: false with any NaN
    double m = 0.0;
    for (int k = 0; k < P; ++k) {
      if (gt[k] != gt[k]) return false;
      const double a = fabs(gt[k]);
      m = a > m ? a : m;
    }
    return m >= bound;
  };
  int rc = QMPS_OK;
  // ---- D = 8, 16: the algebra between two evaluations on the device (qmps_evolve_lockstep.hip) -------------------------------
  // x, g, H^-1, f, the masks and a control word live in HBM; the host enqueues [direction -> evaluation -> accept] chains and reads
  // the control word back once per chain.  The iteration in which a trajectory rejects the full step is finished by the host
  // code below (ladder, gradient at the accepted point, update) on a downloaded copy of the state - the same code, the same
  // decisions.  QMPS_EVOLVE_HOST_ALGEBRA selects the host loop for everything (the round-4 driver; the test-suite runs both).
  const bool dev_algebra = two_sided && (c->D == 8 || c->D == 16) && P <= 32 && T <= 65535 && maxiter <= 480 && documented_switch("QMPS_EVOLVE_HOST_ALGEBRA") == nullptr &&
                           documented_switch("QMPS_D16_BLOCK") == nullptr && documented_switch("QMPS_D16_ONE_WAVE") == nullptr;
  struct {
    double *X, *G, *H, *F, *Dv, *slope, *Xc, *fh, *ph, *F0, *asel, *alphas, *cand, *tolarr, *g0max;
    int* ctl;          // [0, 4) the control word; [16, 16 + maxiter + 1): trajectories that rejected the full step, per iteration of the time step
    unsigned char *active, *eff, *need, *head;
  } dv = {};
  // first chain of a time step: as many iterations as the previous step took (the lock-step count is steady along an evolution with
  // carried Hessians; an idle iteration at the tail of a chain costs ~40 us of empty launches, a chain too short a synchronisation
  // per further iteration); QMPS_EVOLVE_CHAIN (tuning builds) fixes it
  int chain_fixed = 0, nit_prev = 4;
  std::vector<unsigned char> rej_prev;      // iterations of the previous time step in which a full step was rejected
  if (const char* e = tuning_knob("QMPS_EVOLVE_CHAIN")) chain_fixed = atoi(e) > 0 ? atoi(e) : 0;
  if (dev_algebra) {
    const size_t n_ctl = 16 + 2 * ((size_t)maxiter + 2);
    const size_t n_dbl = 4 * TP + TP * P + 6 * (size_t)T + (size_t)n_steps * 2 * T + (size_t)n_steps * TP + (size_t)NA + (size_t)T * (G > 0 ? G : 1) * P;
    const size_t bytes = n_dbl * sizeof(double) + (n_ctl + (n_ctl & 1)) * sizeof(int) + 4 * (((size_t)T + 7) / 8 * 8) + 64;
    if (bytes > c->d_lock_bytes) {
      if (c->d_lock) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_lock)); }
      c->d_lock = nullptr; c->d_lock_bytes = 0;
      HIP_TRY(hipMalloc(&c->d_lock, bytes));
      c->d_lock_bytes = bytes;
    }
    double* q = (double*)c->d_lock;
    dv.X = q; q += TP; dv.G = q; q += TP; dv.Dv = q; q += TP; dv.Xc = q; q += TP;
    dv.H = q; q += TP * P; dv.F = q; q += T; dv.slope = q; q += T; dv.F0 = q; q += T; dv.asel = q; q += T; dv.tolarr = q; q += T; dv.g0max = q; q += T;
    dv.fh = q; q += (size_t)n_steps * 2 * T; dv.ph = q; q += (size_t)n_steps * TP; dv.alphas = q; q += NA; dv.cand = q; q += (size_t)T * (G > 0 ? G : 1) * P;
    dv.ctl = (int*)q;
    dv.active = (unsigned char*)(dv.ctl + n_ctl + (n_ctl & 1)); dv.eff = dv.active + ((size_t)T + 7) / 8 * 8; dv.need = dv.eff + ((size_t)T + 7) / 8 * 8; dv.head = dv.need + ((size_t)T + 7) / 8 * 8;
    if (!c->d_active) HIP_TRY(hipMalloc((void**)&c->d_active, ((size_t)c->max_batch + 7) / 8 * 8));
    if (!c->h_ctl) HIP_TRY(hipHostMalloc((void**)&c->h_ctl, 4096, hipHostMallocDefault));
    if ((rc = ensure_overlap_outputs(c))) return rc;
    if (!c->d_y) HIP_TRY(hipMalloc(&c->d_y, (size_t)c->max_batch * env_bytes(c)));
    { const size_t nD = (size_t)c->D * c->D; if ((rc = ensure_scratch(c, (size_t)T * (4 * nD + 1) * 16 + 256))) return rc; }
    if ((rc = ensure_refs(c, T))) return rc;
    if (!c->aux_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
    }
    HIP_TRY(hipMemcpyAsync(dv.X, X.data(), TP * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dv.H, Hinv.data(), TP * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dv.alphas, alphas, (size_t)NA * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(dv.ctl, 0, 16 * sizeof(int), c->stream));      // control word, barrier accumulators and arrival counter
    if (adaptive) {
      HIP_TRY(hipMemcpyAsync(dv.tolarr, tolv.data(), (size_t)T * sizeof(double), hipMemcpyHostToDevice, c->stream));      // (first evaluation: the tightest)
      HIP_TRY(hipMemsetAsync(dv.g0max, 0, (size_t)T * sizeof(double), c->stream));
    }
    if ((rc = set_ww(c, WW))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));       // (X, Hinv are pageable host vectors)
    c->window = 0;
    c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
    c->ans_have = false; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0; c->tensors_valid = false; c->n_states = 0;
  }
  const bool beside_dev = T <= 1024 && !c->one_stream;
  // (no second stream when the neighbours' tensors are built inside the pair launch or inside the probe kernel)
  const bool fused_probe_dev = (qmps::overlap_probe_fusable(c->D, kind, P) && documented_switch("QMPS_FUSED_PROBE") != nullptr) ||
                               (qmps::neighbour_build_in_pair(c->D, kind, P) && documented_switch("QMPS_NEIGHBOURS_BESIDE") == nullptr &&
                                documented_switch("QMPS_D16_BLOCK") == nullptr && documented_switch("QMPS_D16_ONE_WAVE") == nullptr);
  auto lock_args = [&](int step, bool reset_h, int mode) {
    qmps::LockstepArgs la;
    memset(&la, 0, sizeof(la));
    la.X = dv.X; la.G = dv.G; la.H = dv.H; la.F = dv.F; la.Dv = dv.Dv; la.slope = dv.slope; la.Xc = dv.Xc;
    la.F0 = dv.F0; la.asel = dv.asel; la.alphas = dv.alphas; la.cand = dv.cand; la.NA = NA;
    la.tol_next = adaptive ? dv.tolarr : nullptr; la.g0max = dv.g0max; la.tol_min = tol_min; la.tol_max = tol_max; la.tol_rel = tol_rel;
    la.fb = c->d_f; la.st = c->d_status; la.active = dv.active; la.eff = dv.eff; la.need = dv.need; la.ctl = dv.ctl;
    la.fh_start = dv.fh + (size_t)step * 2 * T; la.fh_end = dv.fh + ((size_t)step * 2 + 1) * T; la.ph = dv.ph + (size_t)step * TP; la.mode = mode;
    la.step_id = step + 1; la.head_mask = dv.head; la.hist_off = ((step + 1) & 1) * (maxiter + 2);
    la.T = (int)T; la.P = P; la.maxiter = maxiter; la.reset_h = reset_h ? 1 : 0; la.h = h; la.gtol = gtol; la.c1 = c1; la.alpha0 = alphas[0];
    return la;
  };
  // QMPS_BFGS_TIME_STEPS (with counters_out): no event pairs around the evaluations and no one-iteration chains - the run is the timed
  // region's - but ONE pair per time step, from its first kernel to the last one enqueued: counters_out[3] = the milliseconds the device
  // spent on this call's kernels (idle launches at a chain's tail included; the host's gap between two time steps not)
  const bool time_steps = dev_algebra && counters_out != nullptr && (flags & QMPS_BFGS_TIME_STEPS) != 0;
  const bool per_eval = counters_out != nullptr && !time_steps;
  if (time_steps && !c->step_ev0) {
    HIP_TRY(hipEventCreate(&c->step_ev0));
    HIP_TRY(hipEventCreate(&c->step_ev1));
  }
  int lock_epoch = 0;          // launches of the step kernel on this control word (its grid barrier counts arrivals against it)
  const int lock_blocks = dev_algebra ? qmps::lockstep_step_blocks((int)T, P) : 0;
  auto launch_step = [&](int step, bool reset_h, int mode) -> int {
    qmps::LockstepArgs a2 = lock_args(step, reset_h, mode);
    a2.epoch = ++lock_epoch;
    a2.blocks = lock_blocks;
    HIP_TRY(qmps::launch_lockstep_step(a2, c->stream));
    return QMPS_OK;
  };
  // one evaluation of the rows at d_src (iterate tensors, both fixed points, neighbours, probes), enqueued only
  auto dev_gradient = [&](const double* d_src, const unsigned char* mask) -> int {
    HIP_TRY(qmps::launch_ansatz(c->D, kind, d_src, P, c->d_A, T, c->stream));      // (every row: a masked-out row's tensor is never read)
    if (beside_dev && !fused_probe_dev) HIP_TRY(hipEventRecord(c->aux_fork, c->stream));      // (the second stream builds the neighbours' tensors)
    c->timed = per_eval;
    const int tslot = (int)(c->samples % qmps_ctx::kRing);
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[tslot], c->stream));
    c->dominant = c->D == 16 ? "overlap_mfma_d16_kernel + adjoint + neighbour probes" : "overlap solve + adjoint + neighbour probes";
    GradPass gp;
    if (int e = enqueue_gradient_kernels(c, T, kind, P, d_src, h, grad_rounds, grad_tol, warm, true, mask, beside_dev, false, gp, adaptive ? dv.tolarr : nullptr)) return e;
    if (c->timed) { HIP_TRY(hipEventRecord(c->kev1[tslot], c->stream)); c->samples++; }
    c->launches++;
    warm = true;
    c->grad_warm_T = T;
    return QMPS_OK;
  };
  bool head_done = false;      // the head of this time step (references, first evaluation, begin) already ran behind the previous step's chain
  for (int step = 0; step < n_steps && rc == QMPS_OK && dev_algebra; ++step) {
    const bool reset_h = !(carry && (step > 0 || ((flags & QMPS_BFGS_WARM) != 0 && hinv)));
    if (time_steps) HIP_TRY(hipEventRecord(c->step_ev0, c->stream));
    if (!head_done) {
      HIP_TRY(qmps::launch_ansatz(c->D, kind, dv.X, P, c->d_ref, T, c->stream));        // the step's references: A_t = tensor(current parameters)
      c->overlap_refs = T;
      c->overlap_group = 0;
      if ((rc = dev_gradient(dv.X, nullptr))) break;
      if ((rc = launch_step(step, reset_h, 1))) break;      // f, g, active set; the first direction
    }
    head_done = false;
    n_grad += 1.0;
    nfev += (double)T * (2 * P + 1);
    int nit = 0;
    const qmps::LockstepArgs la = lock_args(step, false, 0);
    // the ladder of an iteration that stopped on rejected full steps, and what follows it - enqueued only: candidates of every
    // trajectory, their solves masked by `need` and started from the rejected steps' fixed points, the verdict, the gradient at the
    // accepted points (masked alike), the update and the next direction (step kernel, mode 3).  Every kernel of it does nothing
    // when nothing was rejected, so it may be enqueued blindly where the previous time step had a rejection.
    auto enqueue_ladder = [&]() -> int {
      if (G <= 0) return fail(QMPS_ERR_ARG, "a rejected full step needs a ladder (n_alphas >= 2)");
      HIP_TRY(qmps::launch_lockstep_ladder_cand(la, c->stream));
      HIP_TRY(hipMemcpyAsync(c->d_active, dv.need, (size_t)T, hipMemcpyDeviceToDevice, c->stream));
      c->mask_stash_n = 0; c->mask_host = nullptr; c->active_n = T;
      c->ans_have = true; c->ans_kind = kind; c->ans_P = P; c->ans_src = dv.cand; c->ans_i = nullptr; c->ans_nsh = 0;
      c->tensors_valid = false; c->n_states = T * G; c->window = 0;
      c->overlap_group = G;
      c->warm_from_group = (c->grad_warm_T == T) ? G : 0;      // (resident: the fixed points of the rejected full steps)
      const int e = qmps_overlap_launch(c, T * G, ladder_rounds, tol, 0);
      c->overlap_group = 0;
      c->ans_have = false; c->ans_src = nullptr; c->tensors_valid = false; c->n_states = 0;
      if (e) return e;
      HIP_TRY(qmps::launch_lockstep_ladder_pick(la, c->d_f, c->d_status, c->stream));
      if (int e2 = dev_gradient(dv.Xc, dv.need)) return e2;
      return launch_step(step, false, 3);
    };
    bool first_chain = true;
    for (;;) {
      // with counters: one iteration per chain, so that every evaluation's event pair can be read (the timed region runs without)
      int K = per_eval ? 1 : (first_chain ? (chain_fixed > 0 ? chain_fixed : nit_prev) : 1);
      K = K < 1 ? 1 : K;
      K = K < maxiter - nit ? K : maxiter - nit;
      const bool spec_ok = first_chain && !counters_out && documented_switch("QMPS_EVOLVE_SPECULATIVE_HEAD") != nullptr;
      first_chain = false;
      for (int i = 0; i < K; ++i) {
        if ((rc = dev_gradient(dv.Xc, dv.eff))) break;
        if ((rc = launch_step(step, false, 0))) break;               // finish the iteration, open the next
        // the previous time step had a rejection at this iteration: its ladder rides along (empty launches if nothing is rejected now)
        if (!per_eval && (size_t)(nit + i) < rej_prev.size() && rej_prev[nit + i] && (rc = enqueue_ladder())) break;
      }
      if (rc) break;
      // (QMPS_EVOLVE_SPECULATIVE_HEAD; off by default: measured 0.598 against 0.602 ms per time step carried, 3.7 against 3.2 ms identity
      // start - the host is back and enqueueing before the device has drained the chain, so there is no gap to fill)
      // the NEXT time step's head behind this chain, masked by "this time step has finished" (head_mask / ctl[5], written by the step
      // kernel that ends it): when the chain was long enough - the rule - the device goes on without waiting for the host to find out;
      // otherwise every kernel of it returns at once
      const bool spec = spec_ok && step + 1 < n_steps;
      if (spec) {
        HIP_TRY(qmps::launch_ansatz_masked(c->D, kind, dv.X, P, c->d_ref, T, dv.head, c->stream));
        if ((rc = dev_gradient(dv.X, dv.head))) break;
        if ((rc = launch_step(step + 1, !carry, 4))) break;
      }
      if (time_steps) HIP_TRY(hipEventRecord(c->step_ev1, c->stream));
      const size_t n_read = 16 + 2 * ((size_t)maxiter + 2);
      HIP_TRY(hipMemcpyAsync(c->h_ctl, dv.ctl, n_read * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      const int* hist = c->h_ctl + 16 + ((step + 1) & 1) * (maxiter + 2);
      const bool finished = c->h_ctl[5] == step + 1;
      // (finished with a speculative head behind it: the control word already describes the NEXT time step)
      const int n_act = finished ? 0 : c->h_ctl[0], nit_dev = finished ? c->h_ctl[6] : c->h_ctl[2], stop = finished ? 0 : c->h_ctl[3];
      if (per_eval && K > 0) {
        float ms = 0.f;
        if ((nit_dev > nit || stop) && qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
      }
      if (per_eval) {       // (K = 1: exact counts, as the host loop's)
        n_grad += (double)(nit_dev - nit) + (stop ? 1.0 : 0.0);
        nfev += ((double)(nit_dev - nit) + (stop ? 1.0 : 0.0)) * (double)T * (2 * P + 1);
      }
      nit = nit_dev;
      if (finished) {
        head_done = spec;
        rej_prev.assign((size_t)nit, 0);
        for (int i = 0; i < nit; ++i) rej_prev[i] = hist[i] > 0 ? 1 : 0;
        break;
      }
      if (stop) {
        // some trajectories rejected the full step and no ladder was waiting: enqueue it now (no further synchronisation - the
        // next chain follows at once)
        if ((rc = enqueue_ladder())) break;
        if (time_steps) HIP_TRY(hipEventRecord(c->step_ev1, c->stream));
        if (per_eval) {
          HIP_TRY(hipStreamSynchronize(c->stream));
          float ms = 0.f;
          if (qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
          n_ladder += 1.0;
          n_grad += 1.0;
          nfev += (double)T * G + (double)T * (2 * P + 1);
        }
        nit += 1;            // (the step kernel of mode 3 counts it on the device; the next read-back finds the step finished or not)
        continue;
      }
      if (n_act == 0 || nit >= maxiter) break;
    }
    if (rc) break;
    if (time_steps) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, c->step_ev0, c->step_ev1));
      grad_ms += ms;
    }
    // (the step's record - objective at the end, parameters - was written by the last live step kernel; on the device until the call ends)
    nit_prev = nit > 0 ? nit : 1;
    if (nit_out) nit_out[step] = nit;
  }
  if (dev_algebra) {
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    std::vector<double> fh((size_t)n_steps * 2 * T), ph(params_hist ? (size_t)n_steps * TP : 0);
    HIP_TRY(hipMemcpyAsync(fh.data(), dv.fh, fh.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (params_hist) HIP_TRY(hipMemcpyAsync(ph.data(), dv.ph, ph.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(X.data(), dv.X, TP * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(Hinv.data(), dv.H, TP * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int step = 0; step < n_steps; ++step) {
      memcpy(f_hist + (size_t)step * 2 * T_hist + t_off, &fh[(size_t)step * 2 * T], (size_t)T * sizeof(double));
      memcpy(f_hist + ((size_t)step * 2 + 1) * T_hist + t_off, &fh[((size_t)step * 2 + 1) * T], (size_t)T * sizeof(double));
      if (params_hist) memcpy(params_hist + ((size_t)step * T_hist + t_off) * P, &ph[(size_t)step * TP], TP * sizeof(double));
    }
    c->window = 0;
    c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
  }
  for (int step = 0; step < n_steps && rc == QMPS_OK && !dev_algebra; ++step) {
    // the step's references: A_t = tensor(current parameters)
    {
      Restore<bool> deferred(c->defer_sync, true);
      if ((rc = qmps_overlap_set_refs_ansatz(c, T, kind, P, X.data(), WW))) break;
    }
    if (!(carry && (step > 0 || (warm && hinv))))
      for (int64_t t = 0; t < T; ++t) set_identity(t);
    if (adaptive)      // first evaluation of a time step: by the gradient the previous step's first evaluation found (first step: the tightest)
      for (int64_t t = 0; t < T; ++t) tolv[t] = tol_rule(g0max_prev[t], false);
    if ((rc = value_and_grad(X.data(), f.data(), g.data(), nullptr))) break;
    if (adaptive)
      for (int64_t t = 0; t < T; ++t) {
        bool isn;
        const double m = gmax_of(&g[(size_t)t * P], isn);
        g0max_prev[t] = isn ? 0.0 : m;
      }
    memcpy(f_hist + (size_t)step * 2 * T_hist + t_off, f.data(), (size_t)T * sizeof(double));          // objective at the start of the time step
    bool any_active = false;
    for (int64_t t = 0; t < T; ++t) { active[t] = gmax_at_least(&g[(size_t)t * P], gtol) ? 1 : 0; any_active |= active[t] != 0; }
    int nit = 0;
    while (nit < maxiter && any_active) {
      for (int64_t t = 0; t < T; ++t) {
        const double* Ht = &Hinv[(size_t)t * P * P];
        const double* gt = &g[(size_t)t * P];
        double* dt = &d[(size_t)t * P];
        // (a trajectory that has stopped keeps g and H^-1: its direction test below had its one possible effect in iteration 0)
        if (!active[t] && nit > 0) {
          for (int a = 0; a < P; ++a) dt[a] = 0.0;
          continue;
        }
        double sl = 0.0;
        for (int a = 0; a < P; ++a) {
          double acc = 0.0;
          for (int b = 0; b < P; ++b) acc += Ht[a * P + b] * gt[b];
          dt[a] = -acc;
        }
        for (int a = 0; a < P; ++a) sl += gt[a] * dt[a];
        if (!(sl < 0.0)) {                                  // not a descent direction: restart from steepest descent
          set_identity(t);
          sl = 0.0;
          for (int a = 0; a < P; ++a) { dt[a] = -gt[a]; sl -= gt[a] * gt[a]; }
        }
        slope[t] = sl;
        if (!active[t]) for (int a = 0; a < P; ++a) dt[a] = 0.0;
      }
      // the full step with its gradient, straight away
      for (size_t q = 0; q < TP; ++q) Xc[q] = X[q] + alphas[0] * d[q];
      if (adaptive)      // by the trajectory's current gradient (the ladder's gradient at the accepted point, below, uses the same)
        for (int64_t t = 0; t < T; ++t) {
          bool isn;
          const double m = gmax_of(&g[(size_t)t * P], isn);
          tolv[t] = tol_rule(m, isn);
        }
      if ((rc = value_and_grad(Xc.data(), fs.data(), gs.data(), active.data()))) break;
      bool all_accept = true;
      for (int64_t t = 0; t < T; ++t) {
        if (!active[t]) {                                   // (rows of skipped trajectories: their last values)
          fs[t] = f[t];
          memcpy(&gs[(size_t)t * P], &g[(size_t)t * P], P * sizeof(double));
        }
        double* Ft = &Fc[(size_t)t * NA];
        for (int r = 0; r < NA; ++r) Ft[r] = INFINITY;
        Ft[0] = std::isfinite(fs[t]) ? fs[t] : INFINITY;
        // need: the trajectories that rejected the full step - the ladder and the gradient at the accepted point are for them only
        need[t] = (active[t] && !(Ft[0] <= f[t] + c1 * alphas[0] * slope[t])) ? 1 : 0;
        if (need[t]) all_accept = false;
      }
      bool have_new = all_accept;
      if (all_accept) {
        fn = fs;
        gn = gs;
      } else if (G > 0) {
        cand.resize((size_t)T * G * P);
        Fl.resize((size_t)T * G);
        stl.resize((size_t)T * G);
        for (int64_t t = 0; t < T; ++t)
          for (int64_t r = 0; r < G; ++r)
            for (int k = 0; k < P; ++k) cand[((size_t)t * G + r) * P + k] = X[(size_t)t * P + k] + alphas[r + 1] * d[(size_t)t * P + k];
        if ((rc = qmps_overlap_set_group(c, G))) break;
        if ((rc = qmps_overlap_set_active(c, T, need.data()))) break;
        c->warm_from_group = (two_sided && c->grad_warm_T == T) ? G : 0;      // (resident: the fixed points of the rejected full steps)
        rc = qmps_overlap_eval_ansatz(c, T * G, kind, P, cand.data(), ladder_rounds, tol, 0, Fl.data(), stl.data());
        (void)qmps_overlap_set_group(c, 0);
        if (rc) break;
        n_ladder += 1.0;
        nfev += (double)T * G;
        for (int64_t t = 0; t < T; ++t)
          for (int64_t r = 0; r < G; ++r) {
            const double v = (need[t] && qmps::overlap_usable(stl[(size_t)t * G + r])) ? Fl[(size_t)t * G + r] : nan;
            Fc[(size_t)t * NA + r + 1] = std::isfinite(v) ? v : INFINITY;
          }
      }
      for (int64_t t = 0; t < T; ++t) {
        const double* Ft = &Fc[(size_t)t * NA];
        int first = -1, best = 0;
        for (int r = 0; r < NA; ++r) {
          if (first < 0 && Ft[r] <= f[t] + c1 * alphas[r] * slope[t]) first = r;
          if (Ft[r] < Ft[best]) best = r;
        }
        if (first < 0) first = best;
        moved[t] = (active[t] && Ft[first] < f[t]) ? 1 : 0;
        const double a = moved[t] ? alphas[first] : 0.0;
        for (int k = 0; k < P; ++k) {
          s[(size_t)t * P + k] = a * d[(size_t)t * P + k];
          Xn[(size_t)t * P + k] = X[(size_t)t * P + k] + s[(size_t)t * P + k];
        }
      }
      if (!have_new) {
        if ((rc = value_and_grad(Xn.data(), fn.data(), gn.data(), need.data()))) break;
        for (int64_t t = 0; t < T; ++t)
          if (!need[t]) {                                   // accepted the full step: its values are the speculative batch's
            fn[t] = fs[t];
            memcpy(&gn[(size_t)t * P], &gs[(size_t)t * P], P * sizeof(double));
          }
      }
      any_active = false;
      for (int64_t t = 0; t < T; ++t) {
        double* gt = &g[(size_t)t * P];
        const double* gnt = &gn[(size_t)t * P];
        const double* sv = &s[(size_t)t * P];
        if (moved[t]) {
          double sy = 0.0, ss = 0.0, yy = 0.0;
          for (int k = 0; k < P; ++k) { const double y = gnt[k] - gt[k]; sy += sv[k] * y; ss += sv[k] * sv[k]; yy += y * y; }
          if (sy > 1e-12 * sqrt(ss * yy) && sy > 0.0) {
            // H' = H - rho (s (Hy)^T + (Hy) s^T) + rho (1 + rho y^T H y) s s^T
            double* Ht = &Hinv[(size_t)t * P * P];
            const double rho = 1.0 / sy;
            double yHy = 0.0;
            for (int a = 0; a < P; ++a) {
              double acc = 0.0;
              for (int b = 0; b < P; ++b) acc += Ht[a * P + b] * (gnt[b] - gt[b]);
              Hy[a] = acc;
            }
            for (int a = 0; a < P; ++a) yHy += (gnt[a] - gt[a]) * Hy[a];
            const double coef = rho * (1.0 + rho * yHy);
            for (int a = 0; a < P; ++a)
              for (int b = 0; b < P; ++b) Ht[a * P + b] = Ht[a * P + b] - (rho * sv[a] * Hy[b] + rho * sv[b] * Hy[a]) + coef * sv[a] * sv[b];
          }
          f[t] = fn[t];
          memcpy(gt, gnt, P * sizeof(double));
        }
        active[t] = (active[t] && moved[t] && gmax_at_least(gt, gtol)) ? 1 : 0;
        any_active |= active[t] != 0;
      }
      X = Xn;
      ++nit;
    }
    if (rc) break;
    memcpy(f_hist + ((size_t)step * 2 + 1) * T_hist + t_off, f.data(), (size_t)T * sizeof(double));     // ... and at its end
    if (params_hist) memcpy(params_hist + ((size_t)step * T_hist + t_off) * P, X.data(), TP * sizeof(double));
    if (nit_out) nit_out[step] = nit;
  }
  if (rc) return rc;
  memcpy(params, X.data(), TP * sizeof(double));
  if (hinv) memcpy(hinv, Hinv.data(), TP * P * sizeof(double));
  if (counters_out) { counters_out[0] = n_grad; counters_out[1] = n_ladder; counters_out[2] = nfev; counters_out[3] = grad_ms; }
  return QMPS_OK;
}

// Lock-step groups (include/qmps_hip.h, qmps_set_evolve_groups).  Automatic: groups of >= 256 trajectories, at most four -
// measured at D = 16 (profiles/EXPERIMENTS.md, round 4): T = 1 024 / 2 048 / 4 096 gain 21 / 33 / 53 % with four groups, T = 256 nothing.
int evolve_group_count(const qmps_ctx* c, int64_t T) {
  int64_t K = c->evolve_groups;
  if (const char* e = tuning_knob("QMPS_EVOLVE_GROUPS")) K = atoll(e);
  if (K <= 0) K = T >= 512 ? (T / 256 < 4 ? T / 256 : 4) : 1;
  if (K > T) K = T;
  if (K > 16) K = 16;
  return (int)K;
}
void drop_groups(qmps_ctx* c) {
  for (qmps_ctx* g : c->lockstep) (void)qmps_destroy(g);
  c->lockstep.clear();
  c->lockstep_T = 0;
  c->lockstep_cap = 0;
}
}  // namespace

int qmps_set_evolve_groups(qmps_ctx* c, int groups) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (groups < 0 || groups > 16) return fail(QMPS_ERR_ARG, "groups outside [0, 16] (0 = automatic)");
  c->evolve_groups = groups;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_evolve_groups(qmps_ctx* c, int64_t T, int* groups) try {
  if (!c || !groups) return fail(QMPS_ERR_ARG, "null argument");
  *groups = T >= 2 ? evolve_group_count(c, T) : 1;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_evolve_bfgs(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                     double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                     double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out) try {
  if (int rc = bind(c)) return rc;
  const int K = (T >= 2 && n_params >= 1 && n_alphas >= 1 && n_steps >= 1) ? evolve_group_count(c, T) : 1;
  const bool warm = (flags & QMPS_BFGS_WARM) != 0;
  // a continued evolution goes where its resident fixed points are: the groups' contexts if the previous call was grouped the same way
  // (group 0 runs on THIS context, groups 1 .. K-1 on contexts of their own)
  const bool groups_warm = (int)c->lockstep.size() == K - 1 && c->lockstep_T == T;
  if (K <= 1 || (warm && !groups_warm))
    return evolve_bfgs_group(c, T, T, 0, kind, n_params, params, WW, n_steps, maxiter, gtol, h, c1, n_alphas, alphas, flags, max_rounds, tol, hinv, params_hist,
                             f_hist, nit_out, counters_out);
  // K independent lock-step groups: a context (its own stream and resident buffers) and a host thread each.  While one group's
  // host arithmetic runs, the other groups' kernels do; a straggler holds back its own group only.
  const int P = n_params;
  const int64_t G = n_alphas - 1, per = (1 + 2 * (int64_t)P) > G ? (1 + 2 * (int64_t)P) : G;
  std::vector<int64_t> off(K + 1);
  for (int k = 0; k <= K; ++k) off[k] = T * k / K;
  int64_t Tmax = 0;
  for (int k = 0; k < K; ++k) Tmax = off[k + 1] - off[k] > Tmax ? off[k + 1] - off[k] : Tmax;
  if ((int)c->lockstep.size() != K - 1 || c->lockstep_cap < Tmax * per) {      // (contexts that are large enough serve any T)
    drop_groups(c);
    for (int k = 1; k < K; ++k) {
      qmps_ctx* g = nullptr;
      if (int rc = qmps_create(c->device, c->D, Tmax * per, &g)) { drop_groups(c); return rc; }
      c->lockstep.push_back(g);
    }
    c->lockstep_cap = Tmax * per;
    if (warm) return fail(QMPS_ERR_STATE, "QMPS_BFGS_WARM: no resident fixed points for %lld trajectories in %d groups", (long long)T, K);
  }
  c->lockstep_T = T;
  Restore<bool> one_stream(c->one_stream, true);
  for (qmps_ctx* g : c->lockstep) {          // the solver settings of the parent
    g->one_stream = true;
    g->handoff = c->handoff; g->default_solver = c->default_solver; g->skip_rounds = c->skip_rounds; g->matvec_period = c->matvec_period;
  }
  std::vector<int> rcs(K, QMPS_OK);
  std::vector<std::string> msgs(K);
  std::vector<int32_t> nits((size_t)K * n_steps, 0);
  std::vector<double> cnts((size_t)K * 4, 0.0);
  auto run = [&](int k) {
    try {
      rcs[k] = evolve_bfgs_group(k == 0 ? c : c->lockstep[k - 1], off[k + 1] - off[k], T, off[k], kind, P, params ? params + off[k] * P : nullptr, WW, n_steps, maxiter, gtol, h, c1, n_alphas,
                                 alphas, flags, max_rounds, tol, hinv ? hinv + off[k] * P * P : nullptr, params_hist, f_hist, &nits[(size_t)k * n_steps],
                                 counters_out ? &cnts[(size_t)k * 4] : nullptr);
    } catch (const std::exception& ex) {
      rcs[k] = fail(QMPS_ERR_ARG, "C++ exception inside the library: %s", ex.what());
    } catch (...) {
      rcs[k] = fail(QMPS_ERR_ARG, "unknown C++ exception inside the library");
    }
    if (rcs[k]) msgs[k] = qmps_last_error();           // (the message is per thread)
  };
  {
    std::vector<std::thread> workers;
    struct Join { std::vector<std::thread>& w; ~Join() { for (auto& t : w) if (t.joinable()) t.join(); } } join{workers};
    for (int k = 1; k < K; ++k) workers.emplace_back(run, k);
    run(0);
  }
  (void)hipSetDevice(c->device);
  for (int k = 0; k < K; ++k)
    if (rcs[k]) return fail(rcs[k], "%s (lock-step group %d of %d)", msgs[k].c_str(), k, K);
  if (nit_out)
    for (int s = 0; s < n_steps; ++s) {
      int32_t m = 0;
      for (int k = 0; k < K; ++k) m = nits[(size_t)k * n_steps + s] > m ? nits[(size_t)k * n_steps + s] : m;
      nit_out[s] = m;                      // (the lock-step count: the slowest trajectory's)
    }
  if (counters_out)
    for (int q = 0; q < 4; ++q) {
      counters_out[q] = 0.0;
      for (int k = 0; k < K; ++k) counters_out[q] += cnts[(size_t)k * 4 + q];        // (kernel time: summed over the groups' streams - they overlap)
    }
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_evolve_bfgs_device(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                            double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                            double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out) try {
  if (int rc = bind(c)) return rc;
  if (!params || !WW || !f_hist || !alphas) return fail(QMPS_ERR_ARG, "null argument");
  if (c->D != 2 && c->D != 4 && c->D != 16) return fail(QMPS_ERR_ARG, "qmps_evolve_bfgs_device: D = 2, 4, 16 (D = 8: qmps_evolve_bfgs)");
  if (flags & ~(QMPS_BFGS_CARRY_HESSIAN | QMPS_BFGS_WARM | QMPS_BFGS_TIGHT_GRADIENT | (c->D == 16 ? QMPS_BFGS_ADAPTIVE_GRADIENT : 0))) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  const int P = n_params, NA = n_alphas;
  if (P < 1 || P > 16 || NA < 1 || NA > 16 || 2 * P + NA > 64) return fail(QMPS_ERR_ARG, "n_params <= 16, n_alphas <= 16 and 2 n_params + n_alphas <= 64 (one wave per trajectory)");
  if (c->D == 4 && (NA - 1 > 8 || (kind != QMPS_ANSATZ_SHALLOW_CNOT && kind != QMPS_ANSATZ_SHALLOW_QAOA && kind != QMPS_ANSATZ_SHALLOW_CNOT3)))
    return fail(QMPS_ERR_ARG, "qmps_evolve_bfgs_device at D = 4: eight waves per trajectory - n_alphas <= 9; ShallowCNOT / QAOA / CNOT3");
  if (c->D == 2 && kind != QMPS_ANSATZ_SHALLOW_CNOT && kind != QMPS_ANSATZ_SHALLOW_QAOA && kind != QMPS_ANSATZ_SHALLOW_FULL && kind != QMPS_ANSATZ_SHALLOW_CNOT3 &&
      kind != QMPS_ANSATZ_STATE_GATE)
    return fail(QMPS_ERR_ARG, "qmps_evolve_bfgs_device at D = 2: ansatz kind %d has no device-resident kernel (ShallowCNOT / QAOA / Full / CNOT3 / StateGate); use qmps_evolve_bfgs", kind);
  if (c->D == 16 && kind != QMPS_ANSATZ_SHALLOW_CNOT && kind != QMPS_ANSATZ_SHALLOW_CNOT3)
    return fail(QMPS_ERR_ARG, "qmps_evolve_bfgs_device at D = 16: the wave-distributed circuit covers ShallowCNOT / CNOT3; use qmps_evolve_bfgs");
  if (T < 1 || n_steps < 1 || maxiter < 0 || !(gtol > 0.0) || !(h > 0.0) || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad T / n_steps / maxiter / gtol / h / tol");
  if (c->D == 16) {
    if (max_rounds < 1 || max_rounds > (1 << 24)) return fail(QMPS_ERR_ARG, "max_rounds in [1, 2^24] (power steps of a backtracking point's solve)");
  } else if (max_rounds < 1 || max_rounds > 60) return fail(QMPS_ERR_ARG, "max_rounds in [1, 60] (squarings of the 4 x 4 map)");
  if (int rc = check_ansatz(c, kind, P)) return rc;
  const bool carry = (flags & QMPS_BFGS_CARRY_HESSIAN) != 0, carry_in = carry && (flags & QMPS_BFGS_WARM) != 0 && hinv != nullptr;
  // device arena: params | hinv | params_hist | f_hist | nfev | WW | nit | fail  (doubles first, then the two int arrays)
  const size_t TP = (size_t)T * P, nH = hinv ? TP * P : 0, nPH = params_hist ? (size_t)n_steps * TP : 0, nF = (size_t)n_steps * 2 * T;
  const size_t n_dbl = TP + nH + nPH + nF + 2 * (size_t)T + 32, n_int = (size_t)n_steps * T + (size_t)T;
  if (int rc = ensure_scratch(c, n_dbl * sizeof(double) + n_int * sizeof(int32_t) + 64)) return rc;
  double* d_params = (double*)c->d_scratch;
  double* d_hinv = d_params + TP;
  double* d_ph = d_hinv + nH;
  double* d_fh = d_ph + nPH;
  double* d_nfev = d_fh + nF;
  double* d_rounds = d_nfev + T;
  double* d_ww = d_rounds + T;
  int32_t* d_nit = (int32_t*)(d_ww + 32);
  int32_t* d_fail = d_nit + (size_t)n_steps * T;
  HIP_TRY(hipMemcpyAsync(d_params, params, TP * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_ww, WW, 256, hipMemcpyHostToDevice, c->stream));
  if (carry_in) HIP_TRY(hipMemcpyAsync(d_hinv, hinv, nH * sizeof(double), hipMemcpyHostToDevice, c->stream));
  qmps::EvolveD2Args a;
  memset(&a, 0, sizeof(a));
  a.params = d_params; a.WW = d_ww; a.hinv = hinv ? d_hinv : nullptr; a.params_hist = params_hist ? d_ph : nullptr; a.f_hist = d_fh; a.nit = d_nit;
  a.nfev = d_nfev; a.rounds = d_rounds; a.fail = d_fail; a.T = T; a.P = P; a.n_steps = n_steps; a.maxiter = maxiter; a.NA = NA; a.max_rounds = max_rounds;
  a.carry_in = carry_in ? 1 : 0; a.carry = carry ? 1 : 0; a.gtol = gtol; a.h = h; a.c1 = c1; a.tol = tol;
  for (int r = 0; r < NA; ++r) a.alphas[r] = alphas[r];
  if (const char* e = tuning_knob("QMPS_EVOLVE_PROBE")) a.probe = atoi(e);
  if (documented_switch("QMPS_EVOLVE_D2_SQUARING")) a.probe |= 4;      // D = 2: the eigen-solves of the device-resident driver by squaring (rounds 4-5) instead of the characteristic polynomial
  double* d_prof = nullptr;
  if ((c->D == 16 || c->D == 2 || c->D == 4) && tuning_knob("QMPS_EVOLVE_PROF")) {      // (tuning builds: phase timers of the D = 16 kernel - and with -DQMPS_D2_PHASES of the D = 2 one: total | sincos | solve | circuit | closing barrier | optimiser algebra | passes | squarings - printed below)
    HIP_TRY(hipMalloc((void**)&d_prof, (size_t)T * 8 * sizeof(double)));
    HIP_TRY(hipMemsetAsync(d_prof, 0, (size_t)T * 8 * sizeof(double), c->stream));
    a.prof = d_prof;
  }
  if (c->D == 16) {
    // the two solves of a gradient: as qmps_evolve_bfgs (two-sided objective; QMPS_BFGS_TIGHT_GRADIENT: to tol)
    a.grad_tol = (flags & QMPS_BFGS_TIGHT_GRADIENT) ? tol : (tol > 1e-8 ? tol : 1e-8);
    a.adaptive = ((flags & QMPS_BFGS_ADAPTIVE_GRADIENT) != 0 && (flags & QMPS_BFGS_TIGHT_GRADIENT) == 0) ? 1 : 0;
  }
  c->dominant = c->D == 2 ? "evolve_bfgs_d2_kernel" : (c->D == 4 ? "evolve_bfgs_d4_kernel" : "evolve_bfgs_d16_kernel");
  if (counters_out) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  if (c->D == 2) HIP_TRY(qmps::launch_evolve_bfgs_d2(kind, a, c->stream));
  else if (c->D == 4) HIP_TRY(qmps::launch_evolve_bfgs_d4(kind, a, c->stream));
  else HIP_TRY(qmps::launch_evolve_bfgs_d16(kind, a, c->stream));
  if (counters_out) HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipMemcpyAsync(params, d_params, TP * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (hinv) HIP_TRY(hipMemcpyAsync(hinv, d_hinv, nH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (params_hist) HIP_TRY(hipMemcpyAsync(params_hist, d_ph, nPH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(f_hist, d_fh, nF * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (nit_out) HIP_TRY(hipMemcpyAsync(nit_out, d_nit, (size_t)n_steps * T * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  std::vector<double> nfev;
  std::vector<int32_t> nfail;
  if (counters_out) {
    nfev.resize(2 * T);
    nfail.resize(T);
    HIP_TRY(hipMemcpyAsync(nfev.data(), d_nfev, (size_t)2 * T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(nfail.data(), d_fail, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (d_prof) {
    std::vector<double> pr((size_t)T * 8);
    HIP_TRY(hipMemcpy(pr.data(), d_prof, pr.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d_prof));
    static const char* names[8] = {"total_us", "tensor_us", "solves_us", "G+first_us", "neigh_us", "ladder_us", "grad_passes", "power_steps"};
    for (int q = 0; q < 8; ++q) {
      double m = 0.0, mx = 0.0, mn = 1e300;
      int64_t tmx = 0;
      for (int64_t t = 0; t < T; ++t) {
        const double v = pr[(size_t)t * 8 + q] * (q < 6 ? 0.01 : 1.0);
        m += v; mn = v < mn ? v : mn;
        if (v > mx) { mx = v; tmx = t; }
      }
      fprintf(stderr, "[evolve_d16 prof] %-12s mean %10.2f  min %10.2f  max %10.2f (trajectory %lld)\n", names[q], m / (double)T, mn, mx, (long long)tmx);
    }
  }
  if (counters_out) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double a0 = 0.0, a1 = 0.0, a3 = 0.0;
    for (int64_t t = 0; t < T; ++t) { a0 += nfev[t]; a1 += nfail[t]; a3 += nfev[T + t]; }
    counters_out[0] = a0; counters_out[1] = a1; counters_out[2] = ms; counters_out[3] = a3;
  }
  c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1; c->grad_warm_T = 0;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_evolve_rotosolve(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps,
                          int n_sweeps, int nsh, int max_rounds, double tol, double* params_hist, double* f_hist) try {
  if (int rc = bind(c)) return rc;
  if (!params || !WW || !f_hist) return fail(QMPS_ERR_ARG, "null argument");
  if (nsh != 3 && nsh != 6) return fail(QMPS_ERR_ARG, "nsh must be 3 (single-frequency) or 6 (double-frequency)");
  if (T < 1 || nsh * T > c->max_batch) return fail(QMPS_ERR_ARG, "%d T = %lld candidates exceed max_batch = %lld", nsh, (long long)(nsh * T), (long long)c->max_batch);
  if (n_steps < 1 || n_sweeps < 1) return fail(QMPS_ERR_ARG, "n_steps and n_sweeps must be >= 1");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  const bool squaring = overlap_squares(c);
  const int cap = squaring ? 60 : (1 << 24);
  if (max_rounds < 1 || max_rounds > cap || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol (D = %d: max_rounds in [1, %d])", c->D, cap);
  const int P = n_params;
  const int64_t n_rec = (int64_t)n_steps * n_sweeps;
  auto grow = [&](double*& buf, size_t& have, size_t need) -> int {
    if (need > have) {
      if (buf) HIP_TRY(hipFree(buf));
      buf = nullptr;
      have = 0;
      HIP_TRY(hipMalloc((void**)&buf, need));
      have = need;
    }
    return QMPS_OK;
  };
  if (int rc = grow(c->roto_base, c->roto_base_bytes, (size_t)T * P * sizeof(double))) return rc;
  if (int rc = grow(c->roto_hist, c->roto_hist_bytes, (size_t)T * n_rec * sizeof(double))) return rc;
  if (!c->roto_idx) HIP_TRY(hipMalloc((void**)&c->roto_idx, 4 * sizeof(int)));
  if (int rc = ensure_refs(c, T)) return rc;
  if (int rc = ensure_E(c, c->n_terms > 0 ? c->n_terms : 1)) return rc;
  if (int rc = ensure_overlap_outputs(c)) return rc;
  if (int rc = ensure_scratch(c, (size_t)n_steps * T * P * sizeof(double))) return rc;    // parameter history
  // fixed points of the power method (D = 8, 16), one set per parameter plus one for the unshifted evaluation of a sweep:
  // the candidates of parameter i come back to the same slot in the next sweep and in the next time step - by then the
  // parameters have moved by one sweep's updates, so the resident fixed point is the natural warm start
  const bool warm = !squaring;
  const size_t slot_bytes = (size_t)nsh * T * env_bytes(c);
  if (warm) {
    const size_t need = (size_t)(P + 1) * slot_bytes;
    if (need > c->xwarm_bytes) {
      if (c->d_xwarm) HIP_TRY(hipFree(c->d_xwarm));
      c->d_xwarm = nullptr;
      c->xwarm_bytes = 0;
      HIP_TRY(hipMalloc(&c->d_xwarm, need));
      c->xwarm_bytes = need;
    }
    HIP_TRY(hipMemsetAsync(c->d_xwarm, 0, need, c->stream));       // all zero = cold start
  }
  double *d_base = c->roto_base, *d_hist = c->roto_hist, *d_phist = (double*)c->d_scratch;
  int* d_idx = c->roto_idx;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int rc = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(d_base, params, (size_t)T * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int idx0[4] = {0, 0, 0, P};      // parameter index, arrival counter, finished sweeps, slot of the unshifted evaluation
    HIP_TRY(hipMemcpyAsync(d_idx, idx0, sizeof(idx0), hipMemcpyHostToDevice, c->stream));
    if (int e = set_ww(c, WW)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->window = 0;
    c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
    c->ans_have = false; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
    auto evaluate = [&](int shifts) -> int {      // shifts = nsh: the shifted batch of parameter *d_idx;  0: the T base vectors
      const int64_t n = shifts > 0 ? (int64_t)shifts * T : T;
      HIP_TRY(qmps::launch_ansatz_shifted(c->D, kind, d_base, P, c->d_A, n, shifts, d_idx, c->stream));
      qmps::OverlapArgs a;
      memset(&a, 0, sizeof(a));
      a.A = c->d_ref; a.Bt = c->d_A; a.WW = c->d_ww; a.eta = c->d_eta; a.f_out = c->d_E;
      a.iters = c->d_iters; a.status = c->d_status; a.B = n; a.group = shifts > 0 ? shifts : 1;
      a.max_rounds = max_rounds; a.tol = tol; a.stats = c->d_ostats;
      if (warm) {
        a.x_in = c->d_xwarm; a.r_out = c->d_xwarm;
        a.slot_ptr = shifts > 0 ? d_idx : d_idx + 3; a.slot_stride = (int64_t)slot_bytes;
      }
      return launch_overlap_kernels(c, a);
    };
    auto one_sweep = [&]() -> int {
      for (int i = 0; i < P; ++i) {
        if (int e = evaluate(nsh)) return e;
        HIP_TRY(qmps::launch_roto_update(d_base, c->d_E, c->d_status, (int)T, P, d_idx, 1, nsh, c->roto_rule, c->stream));
      }
      // the sweep's record: the objective of the updated vectors against this time step's reference states
      if (int e = evaluate(0)) return e;
      HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)T, 1, d_idx + 2, 1, c->stream));
      return QMPS_OK;
    };
    const bool use_graph = documented_switch("QMPS_NO_GRAPH") == nullptr && P <= 256;
    if (use_graph) {
      c->capturing = true;
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      const int e = one_sweep();
      const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
      c->capturing = false;
      if (e) return e;
      HIP_TRY(ce);
      HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    for (int step = 0; step < n_steps; ++step) {
      // the states the step starts from are the reference: A_t = tensor(params_t)  (new_time_evolve.py:281-283)
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, P, c->d_ref, T, c->stream));
      for (int sw = 0; sw < n_sweeps; ++sw) {
        if (use_graph) HIP_TRY(hipGraphLaunch(exec, c->stream));
        else if (int e = one_sweep()) return e;
      }
      HIP_TRY(hipMemcpyAsync(d_phist + (size_t)step * T * P, d_base, (size_t)T * P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(params, d_base, (size_t)T * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(f_hist, d_hist, (size_t)T * n_rec * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (params_hist) HIP_TRY(hipMemcpyAsync(params_hist, d_phist, (size_t)n_steps * T * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMPS_OK;
  }();
  c->capturing = false;
  (void)hipStreamSynchronize(c->stream);
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  // what the call leaves resident: the T final candidates (tensors, eta, objective, status) against the last step's references
  c->n_states = rc == QMPS_OK ? T : 0;
  c->tensors_valid = true;
  c->overlap_refs = rc == QMPS_OK ? T : 0;
  c->overlap_group = 0;
  return rc;
}
QMPS_API_CATCH

int qmps_overlap_batch(qmps_ctx* c, int64_t B, const double* A, int a_shared, const double* states, int kind,
                       int n_params, const double* WW, int max_rounds, double tol, double* eta_out, double* r_out,
                       int32_t* rounds_out, int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!A || !WW || !eta_out || (!states && B > 0)) return fail(QMPS_ERR_ARG, "null argument");
  // candidates -> d_A: tensors, unitaries or ansatz parameters
  if (kind == QMPS_INPUT_TENSOR || kind == QMPS_INPUT_UNITARY) {
    if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  } else if (kind >= QMPS_INPUT_ANSATZ_BASE && kind <= QMPS_INPUT_ANSATZ_BASE + 6) {
    if (int rc = qmps_set_states_ansatz(c, B, kind - QMPS_INPUT_ANSATZ_BASE, n_params, states)) return rc;
  } else {
    return fail(QMPS_ERR_ARG, "unknown input kind %d", kind);
  }
  if (B == 0) return QMPS_OK;
  if (int rc = qmps_overlap_set(c, a_shared ? 1 : B, A, WW)) return rc;
  if (int rc = qmps_overlap_launch(c, B, max_rounds, tol, r_out != nullptr ? QMPS_OVERLAP_WANT_R : 0)) return rc;
  return qmps_overlap_get(c, B, eta_out, r_out, rounds_out, status_out);
}
QMPS_API_CATCH

// ---- brick-wall (new_tdvp) contractions -------------------------------------------------------

// ---- the evolve drivers behind versioned option structs (include/qmps_hip.h) ------------------------------------------------------------
namespace {
const double kDefaultLadder[8] = {1.0, 0.5, 0.25, 0.125, 1.0 / 16, 1.0 / 64, 1.0 / 256, 1.0 / 4096};
// the caller's struct (its first field says how much of it there is) over the library's defaults
int read_evolve_structs(const qmps_evolve_opts* opts, const qmps_evolve_out* out, qmps_evolve_opts& o, qmps_evolve_out& r) {
  if (!opts || !out) return fail(QMPS_ERR_ARG, "null options / outputs");
  if (opts->size < 2 * sizeof(uint32_t) || opts->size > sizeof(qmps_evolve_opts))
    return fail(QMPS_ERR_ARG, "qmps_evolve_opts.size = %u: this library knows %zu bytes of it (set it to sizeof(qmps_evolve_opts) of YOUR header; a newer header needs a newer library)", opts->size, sizeof(qmps_evolve_opts));
  if (out->size < 2 * sizeof(uint32_t) || out->size > sizeof(qmps_evolve_out))
    return fail(QMPS_ERR_ARG, "qmps_evolve_out.size = %u: this library knows %zu bytes of it", out->size, sizeof(qmps_evolve_out));
  (void)qmps_evolve_opts_init(&o);
  memcpy(&o, opts, opts->size);
  o.size = sizeof(qmps_evolve_opts);
  memset(&r, 0, sizeof(r));
  memcpy(&r, out, out->size);
  if (o.n_alphas == 0 || o.alphas == nullptr) { o.n_alphas = 8; o.alphas = kDefaultLadder; }
  if (!r.f_hist) return fail(QMPS_ERR_ARG, "qmps_evolve_out.f_hist is required");
  return QMPS_OK;
}
}  // namespace

int qmps_evolve_opts_init(qmps_evolve_opts* opts) try {
  if (!opts) return fail(QMPS_ERR_ARG, "null options");
  memset(opts, 0, sizeof(*opts));
  opts->size = sizeof(qmps_evolve_opts);
  opts->n_steps = 1; opts->maxiter = 200; opts->n_alphas = 0; opts->flags = 0; opts->max_rounds = 0;
  opts->gtol = 1e-5; opts->h = 1e-6; opts->c1 = 1e-4; opts->tol = 1e-12; opts->alphas = nullptr;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_evolve_bfgs_opts(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, const qmps_evolve_opts* opts,
                          const qmps_evolve_out* out) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  qmps_evolve_opts o;
  qmps_evolve_out r;
  if (int rc = read_evolve_structs(opts, out, o, r)) return rc;
  const int rounds = o.max_rounds > 0 ? o.max_rounds : ((c->D == 2 || c->D == 4) ? 60 : 100000);
  return qmps_evolve_bfgs(c, T, kind, n_params, params, WW, o.n_steps, o.maxiter, o.gtol, o.h, o.c1, o.n_alphas, o.alphas, o.flags, rounds, o.tol, r.hinv,
                          r.params_hist, r.f_hist, r.nit, r.counters);
}
QMPS_API_CATCH

int qmps_evolve_bfgs_device_opts(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, const qmps_evolve_opts* opts,
                                 const qmps_evolve_out* out) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  qmps_evolve_opts o;
  qmps_evolve_out r;
  if (int rc = read_evolve_structs(opts, out, o, r)) return rc;
  // 0 = the default of include/qmps_hip.h: 60 squarings at D = 2, 4; at D = 16 max_rounds caps the POWER steps of a backtracking point's solve
  // (qmps_evolve_d16.hip), where 60 would reject nearly every rung
  const int rounds = o.max_rounds > 0 ? o.max_rounds : ((c->D == 2 || c->D == 4) ? 60 : 100000);
  return qmps_evolve_bfgs_device(c, T, kind, n_params, params, WW, o.n_steps, o.maxiter, o.gtol, o.h, o.c1, o.n_alphas, o.alphas, o.flags, rounds, o.tol, r.hinv,
                                 r.params_hist, r.f_hist, r.nit, r.counters);
}
QMPS_API_CATCH
