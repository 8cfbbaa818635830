// qmps_overlap_internal.h - helpers of the overlap entry points (qmps_capi_overlap.hip) that the evolve drivers (qmps_capi_evolve.hip)
// build on.  Host side, library-internal.
#pragma once
#include "qmps_ctx.h"

namespace qmps_host {

int ensure_refs(qmps_ctx* c, int64_t n_ref);
int set_ww(qmps_ctx* c, const double* WW);
int ensure_overlap_outputs(qmps_ctx* c);
bool overlap_squares(const qmps_ctx* c);
int launch_overlap_kernels(qmps_ctx* c, const qmps::OverlapArgs& a_in);
int flush_mask(qmps_ctx* c);

// The one-shot fields armed for "the next launch" (qmps_overlap_set_active's mask, the lock-step driver's warm-from-group and fork
// requests, per-trajectory tolerances) are spent on EVERY way out of the call that was to consume them - also when an earlier stage
// of that call fails: a later, unrelated launch must never inherit them.
struct DisarmOneShots {
  qmps_ctx* c;
  ~DisarmOneShots() { c->active_n = 0; c->mask_stash_n = 0; c->mask_host = nullptr; c->fork_after_copy = nullptr; c->warm_from_group = 0; c->grad_tol_in = nullptr; }
};

// One gradient evaluation of T iterates, ENQUEUED on the context stream and nothing else (see qmps_capi_overlap.hip)
struct GradPass {
  qmps::OverlapArgs a, l;
  qmps::OverlapGradArgs g;
  bool lazy_krylov = false;
};
int enqueue_gradient_kernels(qmps_ctx* c, int64_t T, int kind, int P, const double* d_src, double h, int max_rounds, double tol, bool warm, bool two_sided_f,
                             const unsigned char* mask, bool beside, bool allow_lazy_krylov, GradPass& gp, const double* tol_in = nullptr);

}  // namespace qmps_host
