// qmps_capi_overlap.hip - the C-ABI of the time-evolution overlap objective (declared in include/qmps_hip.h): resident references and
// candidates, the batched objective and its two-sided gradient.  The evolve drivers built on them: qmps_capi_evolve.hip (round 5).
// Split out of qmps_capi.hip in round 3; shared context + helpers: qmps_ctx.h, qmps_overlap_internal.h.
#include "qmps_ctx.h"
#include "qmps_overlap_internal.h"

#include <string>
#include <thread>

using namespace qmps_host;

// (every entry point below is declared extern "C" in include/qmps_hip.h: the definitions inherit the linkage)

namespace qmps_host {
int ensure_refs(qmps_ctx* c, int64_t n_ref) {
  if (n_ref > c->ref_cap) {
    if (c->d_ref) HIP_TRY(hipFree(c->d_ref));
    c->d_ref = nullptr;
    c->ref_cap = 0;
    HIP_TRY(hipMalloc(&c->d_ref, (size_t)n_ref * tensor_bytes(c)));
    c->ref_cap = n_ref;
  }
  return QMPS_OK;
}
int set_ww(qmps_ctx* c, const double* WW) {
  if (!c->d_ww) HIP_TRY(hipMalloc(&c->d_ww, 256));
  HIP_TRY(hipMemcpyAsync(c->d_ww, WW, 256, hipMemcpyHostToDevice, c->stream));
  return QMPS_OK;
}
int ensure_overlap_outputs(qmps_ctx* c) {
  if (!c->d_eta) HIP_TRY(hipMalloc(&c->d_eta, (size_t)c->max_batch * 16));
  if (!c->d_f) HIP_TRY(hipMalloc((void**)&c->d_f, (size_t)c->max_batch * sizeof(double)));
  if (!c->d_ostats) {
    HIP_TRY(hipMalloc((void**)&c->d_ostats, (size_t)qmps::kOverlapStatShards * 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->d_ostats, 0, (size_t)qmps::kOverlapStatShards * 4 * sizeof(unsigned long long), c->stream));
  }
  // the Krylov fall-back's iterate scratch: allocated HERE, never lazily at a launch (a launch may sit inside a stream capture -
  // qmps_evolve_rotosolve - where hipMalloc is illegal); the work counters (d_queue) come from qmps_create
  if (c->D >= 8 && !c->d_kry) HIP_TRY(hipMalloc(&c->d_kry, (size_t)c->max_batch * env_bytes(c)));
  return QMPS_OK;
}
// D = 16 batches above 2 048 evaluations: the four-waves-per-evaluation kernel with a work queue (zeroed here, on the stream)
int arm_queue(qmps_ctx* c, qmps::OverlapArgs& a, int which) {
  a.queue = nullptr;
  if (c->D != 16 || a.B <= 2048 || documented_switch("QMPS_D16_ONE_WAVE") != nullptr || documented_switch("QMPS_D16_BLOCK") != nullptr) return QMPS_OK;
  if (!c->d_queue) return fail(QMPS_ERR_STATE, "overlap work counters not allocated (ensure_overlap_outputs)");
  HIP_TRY(hipMemsetAsync(c->d_queue + which, 0, sizeof(int), c->stream));
  a.queue = c->d_queue + which;
  return QMPS_OK;
}
// D = 8, 16: arm the Krylov fall-back of a power launch (include/qmps_hip.h "fixed-point solvers"): the power kernel hands a
// candidate over once its residual history predicts more than kKrylovAfter further steps.  QMPS_NO_KRYLOV: plain power method.
constexpr int kKrylovAfter = 256;
int arm_krylov(qmps_ctx* c, qmps::OverlapArgs& a, int which) {
  a.krylov_after = 0;
  a.kry_counter = nullptr;
  if (c->D < 8 || documented_switch("QMPS_NO_KRYLOV") != nullptr || documented_switch("QMPS_D16_ONE_WAVE") != nullptr || documented_switch("QMPS_D16_BLOCK") != nullptr) return QMPS_OK;
  if (!c->d_queue || !c->d_kry) return fail(QMPS_ERR_STATE, "Krylov fall-back buffers not allocated (ensure_overlap_outputs)");
  int after = kKrylovAfter;
  if (const char* e = tuning_knob("QMPS_KRYLOV_AFTER")) after = atoi(e);
  if (after <= 0) return QMPS_OK;
  if (a.r_out == nullptr) a.r_out = (char*)c->d_kry + (size_t)c->window * env_bytes(c);     // (the iterate travels through r_out)
  a.krylov_after = after;
  a.kry_counter = c->d_queue + 2 + 3 * which;
  return QMPS_OK;
}

// the kernel the overlap launch of this context runs (name for qmps_kernel_time) and whether it counts squarings
bool overlap_squares(const qmps_ctx* c) { return c->D == 2 || (c->D == 4 && !documented_switch("QMPS_OVERLAP_POWER")); }
int launch_overlap_kernels(qmps_ctx* c, const qmps::OverlapArgs& a_in) {
  qmps::OverlapArgs a = a_in;
  if (int rc = arm_queue(c, a, 0)) return rc;
  if (int rc = arm_krylov(c, a, 0)) return rc;
  const bool squaring = overlap_squares(c);
  c->dominant = c->D == 2 ? "overlap_lane_kernel" : (c->D == 4 && squaring ? "overlap_square_d4_kernel" :
                (c->D == 16 && !documented_switch("QMPS_D16_BLOCK") ? "overlap_mfma_d16_kernel" : "overlap_block_kernel<D>"));
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int slot = (int)(c->samples % qmps_ctx::kRing);
  if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
  a.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
  if (c->D == 2) HIP_TRY(qmps::launch_overlap(a, c->stream));
  else HIP_TRY(qmps::launch_overlap_d(c->D, a, c->D == 4 ? squaring : documented_switch("QMPS_D16_BLOCK") == nullptr, c->stream));
  if (c->timed) { HIP_TRY(hipEventRecord(c->kev1[slot], c->stream)); c->samples++; }
  if (!c->capturing) c->launches++;
  return QMPS_OK;
}
}  // namespace qmps_host

int qmps_overlap_set(qmps_ctx* c, int64_t n_ref, const double* A, const double* WW) try {
  if (int rc = bind(c)) return rc;
  if (!A || !WW) return fail(QMPS_ERR_ARG, "null argument");
  if (n_ref < 1 || n_ref > c->max_batch) return fail(QMPS_ERR_ARG, "n_ref=%lld outside [1, max_batch]", (long long)n_ref);
  // the reference state(s) live in their own buffer: nothing else in the context writes it
  if (int rc = ensure_refs(c, n_ref)) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_ref, A, (size_t)n_ref * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
  if (int rc = set_ww(c, WW)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->overlap_refs = n_ref;
  c->overlap_group = 0;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_set_refs_ansatz(qmps_ctx* c, int64_t n_ref, int kind, int n_params, const double* params, const double* WW) try {
  if (int rc = bind(c)) return rc;
  if (!params || !WW) return fail(QMPS_ERR_ARG, "null argument");
  if (n_ref < 1 || n_ref > c->max_batch) return fail(QMPS_ERR_ARG, "n_ref=%lld outside [1, max_batch]", (long long)n_ref);
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (int rc = ensure_refs(c, n_ref)) return rc;
  // parameter rows through the scratch arena, tensors built on the device straight into the reference buffer
  const size_t pb = (size_t)n_ref * n_params * sizeof(double);
  if (int rc = ensure_scratch(c, pb)) return rc;
  if (pb <= (3u << 20) - 256 && !c->capturing) {
    // parameters and W through pinned staging (its fourth quarter), moved by ONE kernel on the context stream: two hipMemcpyAsync from
    // pageable memory cost a time step of 256 trajectories ~30 us of copy-queue hops (rocprofv3 kernel trace), 4 % of it
    if (int rc = ensure_pinned(c, (16u << 20))) return rc;
    if (!c->d_ww) HIP_TRY(hipMalloc(&c->d_ww, 256));
    char* stage = c->h_pin + (12u << 20);
    memcpy(stage, params, pb);
    memcpy(stage + (3u << 20) - 256, WW, 256);
    HIP_TRY(qmps::launch_stage_copy2(stage, c->d_scratch, (int64_t)(pb / 8), stage + (3u << 20) - 256, c->d_ww, 32, c->stream));
    HIP_TRY(qmps::launch_ansatz(c->D, kind, (const double*)c->d_scratch, n_params, c->d_ref, n_ref, c->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(c->d_scratch, params, pb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(qmps::launch_ansatz(c->D, kind, (const double*)c->d_scratch, n_params, c->d_ref, n_ref, c->stream));
    if (int rc = set_ww(c, WW)) return rc;
  }
  // (the evolve drivers go straight on to a gradient batch on the same stream: they synchronise there)
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->overlap_refs = n_ref;
  c->overlap_group = 0;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_set_group(qmps_ctx* c, int64_t group) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (group < 0) return fail(QMPS_ERR_ARG, "group must be >= 0");
  c->overlap_group = group;
  return QMPS_OK;
}
QMPS_API_CATCH

namespace qmps_host {
// a stashed mask that no parameter upload has taken along: its own copy kernel
int flush_mask(qmps_ctx* c) {
  if (c->mask_stash_n > 0) {
    const int64_t n = c->mask_stash_n;
    c->mask_stash_n = 0;
    HIP_TRY(qmps::launch_stage_copy(c->mask_stash, c->d_active, (n + 7) / 8, c->stream));
  }
  return QMPS_OK;
}
}  // namespace qmps_host

int qmps_overlap_set_active(qmps_ctx* c, int64_t n, const unsigned char* active) try {
  if (int rc = bind(c)) return rc;
  if (n < 0 || n > c->max_batch) return fail(QMPS_ERR_ARG, "n=%lld outside [0, max_batch]", (long long)n);
  if (n == 0 || !active) {
    c->active_n = 0;
    c->mask_stash_n = 0;
    c->mask_host = nullptr;
    return QMPS_OK;
  }
  if (!c->d_active) HIP_TRY(hipMalloc((void**)&c->d_active, ((size_t)c->max_batch + 7) / 8 * 8));      // (copied in whole 8-byte words)
  // through pinned staging of its own (the parameter / result regions of h_pin are sized by the drivers and may grow or move), two
  // slots of 512 KiB used alternately: the copy kernel of the previous mask may still be in flight when the host writes the next one
  if ((size_t)n + 8 > qmps_ctx::kMaskSlot) return fail(QMPS_ERR_ARG, "mask longer than 2^19 - 8 entries");
  if (!c->h_mask) HIP_TRY(hipHostMalloc((void**)&c->h_mask, 3 * qmps_ctx::kMaskSlot, hipHostMallocDefault));
  unsigned char* stage = c->h_mask + (size_t)(c->active_stage ^= 1) * qmps_ctx::kMaskSlot;
  if (c->active_inflight[c->active_stage]) HIP_TRY(hipEventSynchronize(c->active_ev[c->active_stage]));
  memcpy(stage, active, (size_t)n);
  memset(stage + n, 1, (size_t)((8 - n % 8) % 8));
  if (c->stash_masks && !c->capturing) {       // (the driver synchronises after every batch: no copy of an earlier mask is in flight)
    c->mask_copy.assign(active, active + n);   // the host copy the launch consults after its read-back: not in any staging region
    c->mask_stash = stage;
    c->mask_host = c->mask_copy.data();
    c->mask_stash_n = n;
    c->active_n = n;
    return QMPS_OK;
  }
  HIP_TRY(qmps::launch_stage_copy(stage, c->d_active, (n + 7) / 8, c->stream));
  if (!c->capturing) {
    if (!c->active_ev[c->active_stage]) HIP_TRY(hipEventCreateWithFlags(&c->active_ev[c->active_stage], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->active_ev[c->active_stage], c->stream));
    c->active_inflight[c->active_stage] = true;
  }
  c->active_n = n;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_launch(qmps_ctx* c, int64_t B, int max_rounds, double tol, int flags) try {
  if (int rc = bind(c)) return rc;
  DisarmOneShots disarm{c};      // the mask of qmps_overlap_set_active is ONE-SHOT: spent on every way out
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->overlap_refs < 1) return fail(QMPS_ERR_STATE, "qmps_overlap_set has not been called");
  if (flags & ~(QMPS_OVERLAP_WANT_R | QMPS_OVERLAP_WARM)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  const bool want_r = (flags & QMPS_OVERLAP_WANT_R) != 0, warm = (flags & QMPS_OVERLAP_WARM) != 0;
  if (warm && !want_r) return fail(QMPS_ERR_ARG, "QMPS_OVERLAP_WARM needs QMPS_OVERLAP_WANT_R (the fixed points stay resident for the next launch)");
  if (warm && !c->have_overlap_x) return fail(QMPS_ERR_STATE, "QMPS_OVERLAP_WARM: no resident fixed points (run a launch with QMPS_OVERLAP_WANT_R first)");
  const int64_t group = c->overlap_group;
  if (group > 0) {
    if (c->overlap_refs * group < c->window + B) return fail(QMPS_ERR_STATE, "%lld reference tensors x group %lld for window end %lld", (long long)c->overlap_refs, (long long)group, (long long)(c->window + B));
  } else if (c->overlap_refs != 1 && c->overlap_refs < c->window + B) {
    return fail(QMPS_ERR_STATE, "%lld reference tensors for window end %lld", (long long)c->overlap_refs, (long long)(c->window + B));
  }
  const bool squaring = overlap_squares(c);   // these square the matrix of the map: rounds, not steps
  const int cap = squaring ? 60 : (1 << 24);
  if (max_rounds < 1 || max_rounds > cap || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol (D = %d: max_rounds in [1, %d])", c->D, cap);
  if (int rc = ensure_overlap_outputs(c)) return rc;
  qmps::OverlapArgs a;
  memset(&a, 0, sizeof(a));
  const bool shared = group == 0 && c->overlap_refs == 1;
  // (the window displaces the candidates; with a group the references are addressed by the candidate's GLOBAL index)
  a.A = (shared || group > 0) ? (char*)c->d_ref + (group > 0 ? (size_t)(c->window / group) * tensor_bytes(c) : 0)
                              : (char*)c->d_ref + (size_t)c->window * tensor_bytes(c);
  if (group > 0 && c->window % group) return fail(QMPS_ERR_ARG, "with a candidate group the window must start at a multiple of it");
  if (int rc = ensure_tensors(c)) return rc;
  a.Bt = win_A(c);
  a.WW = c->d_ww;
  a.eta = (char*)c->d_eta + (size_t)c->window * 16;
  a.f_out = c->d_f + c->window;
  a.r_out = want_r ? win_r(c) : nullptr;
  a.x_in = warm ? win_r(c) : nullptr;
  if (c->warm_from_group > 0) {        // the ladder of a lock-step BFGS iteration: every step length starts from its trajectory's fixed point
    if (!warm && !squaring && c->window == 0 && group == c->warm_from_group && c->grad_warm_T * group >= B) {
      a.x_in = c->d_r;
      a.x_in_group = (int)group;
    }
    c->warm_from_group = 0;
  }
  a.stats = c->d_ostats;
  a.group = (int)group;
  a.iters = win_iters(c); a.status = win_status(c); a.B = B; a.a_shared = shared ? 1 : 0; a.max_rounds = max_rounds; a.tol = tol;
  if (int rc = flush_mask(c)) return rc;
  if (c->active_n > 0) {           // one-shot mask of qmps_overlap_set_active: one entry per trajectory (candidate group)
    const int64_t need = group > 0 ? (c->window + B + group - 1) / group : c->window + B;
    if (c->active_n < need) return fail(QMPS_ERR_STATE, "qmps_overlap_set_active: %lld entries for %lld trajectories", (long long)c->active_n, (long long)need);
    a.active = c->d_active + (group > 0 ? c->window / group : c->window);
    c->active_n = 0;
  }
  if (int rc = launch_overlap_kernels(c, a)) return rc;
  c->have_env = false;
  c->have_guess = false;
  if (want_r) {                 // (a launch that keeps no fixed points leaves the resident ones alone)
    c->have_overlap_x = true;
    c->grad_warm_T = 0;
  }
  c->acc_pending = false;
  c->partials_B = -1;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_get(qmps_ctx* c, int64_t B, double* eta_out, double* r_out, int32_t* rounds_out, int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!eta_out) return fail(QMPS_ERR_ARG, "null eta_out");
  if (!c->d_eta) return fail(QMPS_ERR_STATE, "qmps_overlap_launch has not been called");
  HIP_TRY(hipMemcpyAsync(eta_out, (char*)c->d_eta + (size_t)c->window * 16, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  if (r_out) {
    if (!c->have_overlap_x) return fail(QMPS_ERR_STATE, "the last overlap launch did not keep the fixed points (QMPS_OVERLAP_WANT_R)");
    HIP_TRY(hipMemcpyAsync(r_out, win_r(c), (size_t)B * env_bytes(c), hipMemcpyDeviceToHost, c->stream));
  }
  if (rounds_out) HIP_TRY(hipMemcpyAsync(rounds_out, win_iters(c), (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, win_status(c), (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_get_objective(qmps_ctx* c, int64_t B, double* f_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!f_out) return fail(QMPS_ERR_ARG, "null f_out");
  if (!c->d_f) return fail(QMPS_ERR_STATE, "qmps_overlap_launch has not been called");
  HIP_TRY(hipMemcpyAsync(f_out, c->d_f + c->window, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_amplitude(qmps_ctx* c, int64_t B, const double* q, double* amp_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!q || !amp_out) return fail(QMPS_ERR_ARG, "null argument");
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->overlap_refs < 1) return fail(QMPS_ERR_STATE, "qmps_overlap_set has not been called");
  const int64_t group = c->overlap_group;
  if (group > 0) {
    if (c->window % group) return fail(QMPS_ERR_ARG, "with a candidate group the window must start at a multiple of it");
    if (c->overlap_refs * group < c->window + B) return fail(QMPS_ERR_STATE, "%lld reference tensors x group %lld for window end %lld", (long long)c->overlap_refs, (long long)group, (long long)(c->window + B));
  } else if (c->overlap_refs != 1 && c->overlap_refs < c->window + B) {
    return fail(QMPS_ERR_STATE, "%lld reference tensors for window end %lld", (long long)c->overlap_refs, (long long)(c->window + B));
  }
  if (int rc = ensure_tensors(c)) return rc;
  // environments in and amplitudes out through the scratch arena: neither the resident fixed points nor eta of a launch are touched
  const size_t qb = (size_t)B * env_bytes(c), ob = (size_t)B * 16;
  if (int rc = ensure_scratch(c, qb + ob)) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_scratch, q, qb, hipMemcpyHostToDevice, c->stream));
  qmps::OverlapArgs a;
  memset(&a, 0, sizeof(a));
  const bool shared = group == 0 && c->overlap_refs == 1;
  a.A = (shared || group > 0) ? (char*)c->d_ref + (group > 0 ? (size_t)(c->window / group) * tensor_bytes(c) : 0)
                              : (char*)c->d_ref + (size_t)c->window * tensor_bytes(c);
  a.Bt = win_A(c);
  a.WW = c->d_ww;
  a.x_in = c->d_scratch;
  a.eta = (char*)c->d_scratch + qb;
  a.B = B; a.a_shared = shared ? 1 : 0; a.group = (int)group;
  HIP_TRY(qmps::launch_overlap_amplitude(c->D, a, c->stream));
  HIP_TRY(hipMemcpyAsync(amp_out, a.eta, ob, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_stats(qmps_ctx* c, int64_t* evaluations, int64_t* rounds_sum, int64_t* rounds_max, int64_t* not_converged, int reset) try {
  if (int rc = bind(c)) return rc;
  unsigned long long h[4] = {0, 0, 0, 0};
  if (c->d_ostats) {
    std::vector<unsigned long long> sh((size_t)qmps::kOverlapStatShards * 4);
    HIP_TRY(hipMemcpyAsync(sh.data(), c->d_ostats, sh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (reset) HIP_TRY(hipMemsetAsync(c->d_ostats, 0, sh.size() * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < qmps::kOverlapStatShards; ++k) {
      h[0] += sh[4 * k];
      h[1] += sh[4 * k + 1];
      if (sh[4 * k + 2] > h[2]) h[2] = sh[4 * k + 2];
      h[3] += sh[4 * k + 3];
    }
  }
  for (qmps_ctx* g : c->lockstep) {            // the lock-step groups of qmps_evolve_bfgs solve on their own contexts
    int64_t e = 0, rs = 0, rm = 0, nc = 0;
    if (int rc = qmps_overlap_stats(g, &e, &rs, &rm, &nc, reset)) return rc;
    h[0] += (unsigned long long)e; h[1] += (unsigned long long)rs; h[3] += (unsigned long long)nc;
    if ((unsigned long long)rm > h[2]) h[2] = (unsigned long long)rm;
  }
  (void)hipSetDevice(c->device);
  if (evaluations) *evaluations = (int64_t)h[0];
  if (rounds_sum) *rounds_sum = (int64_t)h[1];
  if (rounds_max) *rounds_max = (int64_t)h[2];
  if (not_converged) *not_converged = (int64_t)h[3];
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_overlap_eval_ansatz(qmps_ctx* c, int64_t B, int kind, int n_params, const double* params, int max_rounds, double tol,
                             int flags, double* f_out, int32_t* status_out) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  DisarmOneShots disarm{c};      // (also when the parameter upload / tensor build below fails before qmps_overlap_launch is reached)
  if (!f_out) return fail(QMPS_ERR_ARG, "null f_out");
  // one round trip: parameters in, ansatz + overlap kernels, objective and status out - ONE synchronisation (the optimiser
  // drivers call this twice per iteration; three separate calls cost three synchronisations and two extra launch gaps)
  int rc;
  {
    Restore<bool> deferred(c->defer_sync, true);
    rc = qmps_set_states_ansatz(c, B, kind, n_params, params);
  }
  if (!rc) rc = qmps_overlap_launch(c, B, max_rounds, tol, flags);
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  const size_t fb = (size_t)B * sizeof(double), sb = (size_t)B * sizeof(int32_t);
  if (fb + sb <= (8u << 20) && c->h_pin_bytes >= (16u << 20)) {
    char* out = c->h_pin + (8u << 20);
    // (d_status has max_batch >= B + 1 entries or the tail is never read)
    HIP_TRY(qmps::launch_stage_copy2(c->d_f, out, B, c->d_status, out + fb, status_out ? (B + 1) / 2 : 0, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(f_out, out, fb);
    if (status_out) memcpy(status_out, out + fb, sb);
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(f_out, c->d_f, fb, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, c->d_status, sb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

namespace qmps_host {
// One gradient evaluation of T iterates, ENQUEUED on the context stream and nothing else: [right + left fixed points] beside [the
// 2 P central-difference neighbours' tensors], then [G + probes].  The iterates' tensors are in d_A[0, T), their parameter rows at
// d_src (device).  `beside`: the neighbour tensors are built on the second stream, which waits for c->aux_fork - the caller records
// it on c->stream once d_src is complete.  allow_lazy_krylov: the Krylov fall-back of the two solves is NOT launched behind the
// power kernels (the caller looks at the statuses on the host and asks for the second pass); otherwise it always runs (it costs an
// empty launch when nothing was handed over) and the statuses behind this function are final.
// Outputs: d_f[0, T) objective of the iterates, d_f[T, T + 2 P T) of the neighbours, d_status[0, T) / [T, 2 T) right / left solves,
// d_r / d_y the fixed points (the next call's warm start).  Shared by qmps_overlap_gradient (host loop) and the device-resident
// lock-step BFGS of qmps_evolve_bfgs.
int enqueue_gradient_kernels(qmps_ctx* c, int64_t T, int kind, int P, const double* d_src, double h, int max_rounds, double tol, bool warm, bool two_sided_f,
                             const unsigned char* mask, bool beside, bool allow_lazy_krylov, GradPass& gp, const double* tol_in) {
  const bool squaring = overlap_squares(c);       // D = 4: the right fixed point comes from the squaring kernel (largest column)
  qmps::OverlapArgs& a = gp.a;
  memset(&a, 0, sizeof(a));
  a.A = c->d_ref; a.Bt = c->d_A; a.WW = c->d_ww; a.eta = c->d_eta; a.f_out = c->d_f; a.r_out = c->d_r;
  a.x_in = (warm && !squaring) ? c->d_r : nullptr;
  a.stats = c->d_ostats; a.iters = c->d_iters; a.status = c->d_status; a.B = T; a.a_shared = 0;
  a.max_rounds = squaring && max_rounds > 60 ? 60 : max_rounds; a.tol = tol;
  a.active = mask;
  a.tol_in = (c->D == 8 || c->D == 16) ? tol_in : nullptr;      // per-trajectory tolerances (the lock-step BFGS; the left solve inherits them)
  // The Krylov fall-back of the two solves is launched only if a status asks for it (below): behind every batch, it cost a warm
  // batch of 256 iterates - which never hands anything over - an empty launch and a launch gap, ~10 us of ~200.
  bool lazy_krylov = false;
  const size_t nD = (size_t)c->D * c->D;
  // where the 2 P central-difference neighbours' tensors come from (D = 16, ShallowCNOT families; see include/qmps_hip.h)
  const bool fused_probe = qmps::overlap_probe_fusable(c->D, kind, P) && documented_switch("QMPS_FUSED_PROBE") != nullptr;
  const bool built_in_pair = !fused_probe && qmps::neighbour_build_in_pair(c->D, kind, P) && documented_switch("QMPS_NEIGHBOURS_BESIDE") == nullptr;
  bool neighbours_done = false;
  // the left fixed points: power method on the adjoint map; results behind the iterates' (eta, rounds, status at [T, 2T))
  qmps::OverlapArgs& l = gp.l;
  l = a;
  l.adjoint = 1; l.eta = (char*)c->d_eta + (size_t)T * 16; l.f_out = nullptr; l.r_out = c->d_y; l.x_in = warm ? c->d_y : nullptr;
  l.iters = c->d_iters + T; l.status = c->d_status + T; l.max_rounds = max_rounds;
  if (c->D == 16 && documented_switch("QMPS_D16_BLOCK") == nullptr && documented_switch("QMPS_D16_ONE_WAVE") == nullptr) {
    // both solves in ONE launch: the iteration chains are latency-bound, so the left solve rides along (more than 2 048
    // iterates: the workgroups draw them from two queues)
    a.no_deflation = l.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    if (int e = arm_queue(c, a, 0)) return e;
    if (int e = arm_queue(c, l, 1)) return e;
    if (int e = arm_krylov(c, a, 0)) return e;
    if (int e = arm_krylov(c, l, 1)) return e;
    lazy_krylov = allow_lazy_krylov && a.krylov_after > 0 && l.krylov_after > 0 && a.kry_counter != nullptr && l.kry_counter != nullptr;
    // ... and the central-difference neighbours' tensors by the surplus workgroups of the same launch (ShallowCNOT families;
    // QMPS_NEIGHBOURS_BESIDE: the round-4 kernel on the second stream, QMPS_FUSED_PROBE: inside the probe kernel)
    qmps::NeighbourBuildArgs nb;
    memset(&nb, 0, sizeof(nb));
    if (built_in_pair) {
      nb.params = d_src; nb.out = (char*)c->d_A + (size_t)T * tensor_bytes(c); nb.rows = T; nb.n_params = P; nb.kind = kind; nb.h = h; nb.active = mask;
    }
    HIP_TRY(qmps::launch_overlap_pair_d16(a, l, c->stream, !lazy_krylov, built_in_pair ? &nb : nullptr));
    neighbours_done = built_in_pair;
  } else if (c->D == 8) {
    // D = 8: the same - one launch, the left solves on the SIMDs the right ones leave idle
    a.no_deflation = l.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    if (int e = arm_krylov(c, a, 0)) return e;
    if (int e = arm_krylov(c, l, 1)) return e;
    lazy_krylov = allow_lazy_krylov && a.krylov_after > 0 && l.krylov_after > 0 && a.kry_counter != nullptr && l.kry_counter != nullptr;
    HIP_TRY(qmps::launch_overlap_pair_d8(a, l, c->stream, !lazy_krylov));
  } else if (c->D == 4 && squaring) {
    // D = 4: largest column AND largest row of the squared map in one launch (right and left fixed point, whatever the gap)
    a.l_out = c->d_y;
    HIP_TRY(qmps::launch_overlap_d(c->D, a, true, c->stream));
  } else {
    a.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    HIP_TRY(qmps::launch_overlap_d(c->D, a, c->D == 4 ? squaring : documented_switch("QMPS_D16_BLOCK") == nullptr, c->stream));
    // (D = 4: the squaring kernel again - the left fixed point is the largest row of the squared map, whatever the spectral gap)
    if (c->D == 4 && squaring) l.max_rounds = a.max_rounds;
    l.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    HIP_TRY(qmps::launch_overlap_d(c->D, l, c->D == 4 ? squaring : (c->D == 16 && documented_switch("QMPS_D16_BLOCK") == nullptr), c->stream));
  }
  if (fused_probe || neighbours_done) {
  } else if (beside) {
    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
    HIP_TRY(qmps::launch_ansatz_fd(c->D, kind, d_src, P, (char*)c->d_A + (size_t)T * tensor_bytes(c), T, h, c->aux_stream, mask));
    HIP_TRY(hipEventRecord(c->aux_join, c->aux_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->aux_join, 0));
  }
  else HIP_TRY(qmps::launch_ansatz_fd(c->D, kind, d_src, P, (char*)c->d_A + (size_t)T * tensor_bytes(c), T, h, c->stream, mask));
  qmps::OverlapGradArgs& g = gp.g;
  memset(&g, 0, sizeof(g));
  g.A = c->d_ref; g.WW = c->d_ww; g.r = c->d_r; g.y = c->d_y; g.G = c->d_scratch; g.yr = (char*)c->d_scratch + (size_t)T * 4 * nD * 16;
  g.Bt = (char*)c->d_A + (size_t)T * tensor_bytes(c); g.f_out = c->d_f + T; g.T = T; g.G2P = 2 * P; g.active = mask;
  if (two_sided_f) { g.Bc = c->d_A; g.fc_out = c->d_f; }       // (overwrites the right solve's own estimate)
  if (fused_probe) { g.fd_params = d_src; g.fd_h = h; g.kind = kind; }
  HIP_TRY(qmps::launch_overlap_grad(c->D, g, c->stream));
  gp.lazy_krylov = lazy_krylov;
  return QMPS_OK;
}
}  // namespace qmps_host

int qmps_overlap_gradient(qmps_ctx* c, int64_t T, int kind, int n_params, const double* params, double h, int max_rounds, double tol,
                          int flags, double* f_out, double* g_out, int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  DisarmOneShots disarm{c};      // (one-shot mask: spent on every way out)
  if (!params || !f_out || !g_out) return fail(QMPS_ERR_ARG, "null argument");
  if (c->D < 4) return fail(QMPS_ERR_ARG, "qmps_overlap_gradient: D = 4, 8, 16 (at D = 2 evaluate the central-difference neighbours themselves)");
  if (flags & ~(QMPS_OVERLAP_WARM | QMPS_OVERLAP_TWO_SIDED_F)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  if (!(h > 0.0)) return fail(QMPS_ERR_ARG, "h must be > 0");
  const int P = n_params;
  if (T < 1 || T * (1 + 2 * (int64_t)P) > c->max_batch) return fail(QMPS_ERR_ARG, "T (1 + 2 n_params) = %lld evaluations exceed max_batch = %lld", (long long)(T * (1 + 2 * (int64_t)P)), (long long)c->max_batch);
  if (c->overlap_refs < T) return fail(QMPS_ERR_STATE, "%lld reference tensors resident, %lld trajectories (qmps_overlap_set / qmps_overlap_set_refs_ansatz)", (long long)c->overlap_refs, (long long)T);
  if (max_rounds < 1 || max_rounds > (1 << 24) || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol");
  const bool warm = (flags & QMPS_OVERLAP_WARM) != 0;
  if (warm && c->grad_warm_T != T) return fail(QMPS_ERR_STATE, "QMPS_OVERLAP_WARM: the resident fixed points belong to %lld trajectories, not %lld", (long long)c->grad_warm_T, (long long)T);
  if (int rc = ensure_overlap_outputs(c)) return rc;
  if (!c->d_y) HIP_TRY(hipMalloc(&c->d_y, (size_t)c->max_batch * env_bytes(c)));
  const size_t nD = (size_t)c->D * c->D;
  if (int rc = ensure_scratch(c, (size_t)T * (4 * nD + 1) * 16 + 256)) return rc;
  {      // pinned staging for the results, sized BEFORE anything is in flight through it
    const size_t need = (size_t)T * (1 + 2 * P) * sizeof(double) + (size_t)2 * T * sizeof(int32_t) + (8u << 20);
    if (int e = ensure_pinned(c, need > (16u << 20) ? need : (16u << 20))) return e;
  }
  // the 2 P central-difference neighbours of every iterate (evaluated below to second order in h from (y, r)): their tensors
  // need the parameters only, so they are built on a second stream BESIDE the eigen-solves (at small T a gradient batch is the
  // latency of its slowest solve; the neighbour tensors were a fifth of it in front of the probes)
  if (!c->aux_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
  }
  // (beyond ~1 000 iterates the solves fill the chip by themselves: T = 2 048 measured 5 % slower with the second stream)
  // (a lock-step group of qmps_evolve_bfgs keeps to ONE stream: the other groups fill the chip, and two streams of one group that
  // land on the same hardware queue serialise - measured 5-10 % slower and erratic, profiles/EXPERIMENTS.md round 4)
  const bool no_second_stream = (qmps::overlap_probe_fusable(c->D, kind, P) && documented_switch("QMPS_FUSED_PROBE") != nullptr) ||
                                (qmps::neighbour_build_in_pair(c->D, kind, P) && documented_switch("QMPS_NEIGHBOURS_BESIDE") == nullptr &&
                                 documented_switch("QMPS_D16_BLOCK") == nullptr && documented_switch("QMPS_D16_ONE_WAVE") == nullptr);
  const bool beside = T <= 1024 && !c->one_stream && !no_second_stream;
  // the iterates: parameters -> tensors in d_A[0, T); the fork event sits between the parameter upload and the tensor build, and
  // the second stream is fed only AFTER the solves are submitted (the host calls of the fork used to hold the solves back ~15 us)
  int rc;
  {
    Restore<bool> deferred(c->defer_sync, true);
    if (beside) c->fork_after_copy = c->aux_fork;
    rc = qmps_set_states_ansatz(c, T, kind, P, params);
  }
  if (!rc) rc = ensure_tensors(c);
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  const unsigned char* mask = nullptr;
  if (int e = flush_mask(c)) return e;
  if (c->active_n > 0) {
    if (c->active_n < T) return fail(QMPS_ERR_STATE, "qmps_overlap_set_active: %lld entries for %lld trajectories", (long long)c->active_n, (long long)T);
    mask = c->d_active;
    c->active_n = 0;
  }
  // HIP events around the WHOLE gradient evaluation (right solve, left solve, neighbour tensors, G, probes): qmps_kernel_time
  c->dominant = c->D == 16 ? "overlap_mfma_d16_kernel + adjoint + neighbour probes" : "overlap solve + adjoint + neighbour probes";
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int tslot = (int)(c->samples % qmps_ctx::kRing);
  if (c->timed) HIP_TRY(hipEventRecord(c->kev0[tslot], c->stream));
  if (beside && c->fork_after_copy) {        // (the parameter upload took another path: it did not record the fork)
    c->fork_after_copy = nullptr;
    return fail(QMPS_ERR_STATE, "qmps_overlap_gradient: the parameter upload did not record the fork event");
  }
  // The Krylov fall-back of the two solves is launched only if a status asks for it (below): behind every batch, it cost a warm
  // batch of 256 iterates - which never hands anything over - an empty launch and a launch gap, ~10 us of ~200.
  GradPass gp;
  const double* tol_in = c->grad_tol_in;        // one-shot (qmps_evolve_bfgs, host loop): per-trajectory tolerances
  c->grad_tol_in = nullptr;
  if (int e = enqueue_gradient_kernels(c, T, kind, P, c->d_params, h, max_rounds, tol, warm, (flags & QMPS_OVERLAP_TWO_SIDED_F) != 0, mask, beside, true, gp, tol_in)) return e;
  const bool lazy_krylov = gp.lazy_krylov;
  qmps::OverlapArgs &a = gp.a, &l = gp.l;
  qmps::OverlapGradArgs& g = gp.g;
  if (c->timed) { HIP_TRY(hipEventRecord(c->kev1[tslot], c->stream)); c->samples++; }
  c->launches++;
  // f of the iterates and of their neighbours are contiguous in d_f: one copy; statuses of both solves: one copy (pinned staging)
  // (tried: the two kernels writing pinned host mirrors themselves instead of the copy kernel - 4 352 eight-byte writes over the
  // host link are slower than one coalesced burst: 0.78 - 0.80 -> 0.80 - 0.83 ms per time step at 256 trajectories)
  const size_t fbytes = (size_t)T * (1 + 2 * P) * sizeof(double), sbytes = (size_t)2 * T * sizeof(int32_t);
  double* fall = (double*)(c->h_pin + (8u << 20));
  int32_t* st = (int32_t*)(c->h_pin + (8u << 20) + fbytes);
  // (tried: a completion flag in pinned memory written by the copy kernel's last block and polled by the host instead of
  // hipStreamSynchronize - the wait shrinks by 7 us, the next submission grows by 11: no gain)
  HIP_TRY(qmps::launch_stage_copy2(c->d_f, fall, (int64_t)(fbytes / 8), c->d_status, st, (int64_t)(sbytes / 8), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (lazy_krylov) {
    // a solve that stopped short (handed over by the power kernel, or at its cap): the fall-back kernel, then G and the probes of
    // those trajectories once more (everything else keeps its values), then the read-back again
    // (skipped trajectories keep the statuses of earlier launches: they do not count)
    const unsigned char* hm = mask != nullptr ? c->mask_host : nullptr;
    bool any = false;
    for (int64_t t = 0; t < T && !any; ++t)
      any = (mask == nullptr || hm == nullptr || hm[t] != 0) && (st[t] == qmps::QMPS_ST_NOT_CONVERGED || st[T + t] == qmps::QMPS_ST_NOT_CONVERGED);
    if (any) {
      if (!c->d_active) HIP_TRY(hipMalloc((void**)&c->d_active, ((size_t)c->max_batch + 7) / 8 * 8));
      if ((size_t)T + 8 > qmps_ctx::kMaskSlot) return fail(QMPS_ERR_ARG, "qmps_overlap_gradient: fall-back mask for %lld trajectories exceeds the staging slot", (long long)T);
      if (!c->h_mask) HIP_TRY(hipHostMalloc((void**)&c->h_mask, 3 * qmps_ctx::kMaskSlot, hipHostMallocDefault));
      unsigned char* fix = c->h_mask + 2 * qmps_ctx::kMaskSlot;           // (third slot: never one of the two upload slots)
      std::vector<unsigned char> was(T, 1);
      if (hm != nullptr) was.assign(hm, hm + T);
      for (int64_t t = 0; t < T; ++t) fix[t] = (was[t] != 0 && (st[t] == qmps::QMPS_ST_NOT_CONVERGED || st[T + t] == qmps::QMPS_ST_NOT_CONVERGED)) ? 1 : 0;
      for (int64_t t = T; t < (T + 7) / 8 * 8; ++t) fix[t] = 0;
      HIP_TRY(qmps::launch_overlap_krylov_pair(c->D, a, l, c->stream));
      HIP_TRY(qmps::launch_stage_copy(fix, c->d_active, (T + 7) / 8, c->stream));
      g.active = c->d_active;
      HIP_TRY(qmps::launch_overlap_grad(c->D, g, c->stream));
      // the timed interval of this batch ends HERE when there was a second pass (Krylov pair, G, probes): kernel time, the evolve
      // drivers' gradient_ms and kernel_share_of_wall count the fall-back too (the first read-back sits inside the interval: the
      // cost of discovering the stragglers)
      if (c->timed) HIP_TRY(hipEventRecord(c->kev1[tslot], c->stream));
      HIP_TRY(qmps::launch_stage_copy2(c->d_f, fall, (int64_t)(fbytes / 8), c->d_status, st, (int64_t)(sbytes / 8), c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
  }
  memcpy(f_out, fall, (size_t)T * sizeof(double));
  const double* fn = fall + T;
  for (int64_t t = 0; t < T; ++t) {
    for (int k = 0; k < P; ++k) g_out[t * P + k] = (fn[(size_t)t * 2 * P + k] - fn[(size_t)t * 2 * P + P + k]) / (2.0 * h);
    if (status_out) status_out[t] = st[t] > st[T + t] ? st[t] : st[T + t];
  }
  c->window = 0;
  c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
  c->grad_warm_T = T;
  return QMPS_OK;
}
QMPS_API_CATCH
