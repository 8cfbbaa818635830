// qmps_capi.hip - the C-ABI of libqmps_hip.so (declared in include/qmps_hip.h): library / device, context lifetime, states,
// the energy path, the rotosolve drivers, read-back, the summed-cost exchange (RCCL), probes.  The time-evolution overlap
// objective and the evolve drivers: qmps_capi_overlap.hip.  Shared context + helpers: qmps_ctx.h.
// Host-side runtime: context = one device + one HIP stream + HBM buffers; asynchronous launches;
// pinned staging for small results; native RCCL communicator for the summed-cost all-reduce.
#include "qmps_ctx.h"

#include <time.h>

using namespace qmps_host;

namespace {
thread_local char g_err[512] = "";
}  // namespace

namespace qmps_host {

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}


int bind(qmps_ctx* c) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return QMPS_OK;
}


int ensure_scratch(qmps_ctx* c, size_t bytes) {
  if (bytes > c->scratch_bytes) {
    if (c->d_scratch) HIP_TRY(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
  }
  return QMPS_OK;
}

int ensure_E(qmps_ctx* c, int n_terms) {
  const int64_t need = c->max_batch * n_terms;
  if (need > c->E_capacity) {
    if (c->d_E) HIP_TRY(hipFree(c->d_E));
    c->d_E = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_E, (size_t)need * sizeof(double)));
    c->E_capacity = need;
  }
  return QMPS_OK;
}

int check_B(const qmps_ctx* c, int64_t B) {
  if (B < 0 || B > c->max_batch) return fail(QMPS_ERR_ARG, "B=%lld outside [0, max_batch=%lld]", (long long)B, (long long)c->max_batch);
  return QMPS_OK;
}

// launch / read-back calls: the window [window, window + B) must lie inside the buffers
int check_window(const qmps_ctx* c, int64_t B) {
  if (B < 0 || c->window + B > c->max_batch)
    return fail(QMPS_ERR_ARG, "window [%lld, %lld) outside [0, max_batch=%lld]", (long long)c->window, (long long)(c->window + B), (long long)c->max_batch);
  return QMPS_OK;
}

// addresses of the window's first evaluation

// kinds the D = 4 direct kernel builds in front of the solve (three-qubit circuits with a per-layer gate list)
bool fusable_ansatz(const qmps_ctx* c, int kind) {
  static const bool off = documented_switch("QMPS_NO_FUSED_ANSATZ") != nullptr;   // A/B knob
  return !off && c->D == 4 && (kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_QAOA || kind == QMPS_ANSATZ_SHALLOW_CNOT3);
}

int ensure_pinned(qmps_ctx* c, size_t bytes) {
  if (bytes > c->h_pin_bytes) {
    // growth: a staged copy kernel of an earlier call may still be reading the old buffer, and a staged upload may be waiting in
    // it - drain the device, carry the contents over, then free (offsets into the buffer stay valid; nothing keeps raw pointers)
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    char* fresh = nullptr;
    HIP_TRY(hipHostMalloc((void**)&fresh, want, hipHostMallocDefault));
    if (c->h_pin) {
      const hipError_t se = hipDeviceSynchronize();
      if (se != hipSuccess) { (void)hipHostFree(fresh); HIP_TRY(se); }
      memcpy(fresh, c->h_pin, c->h_pin_bytes);
      (void)hipHostFree(c->h_pin);
    }
    c->h_pin = fresh;
    c->h_pin_bytes = want;
  }
  return QMPS_OK;
}

// (kind, n_params) of an ansatz the device builders know (qmps/represent.py:268-404)
int check_ansatz(const qmps_ctx* c, int kind, int n_params) {
  if (n_params < 1 || n_params > 4096) return fail(QMPS_ERR_ARG, "n_params=%d outside [1,4096]", n_params);
  if (kind < 0 || kind > 6) return fail(QMPS_ERR_ARG, "unknown ansatz kind %d", kind);
  if (kind == QMPS_ANSATZ_SHALLOW_FULL && (c->D != 2 || n_params != 15))
    return fail(QMPS_ERR_ARG, "ShallowFullStateTensor is a two-qubit gate: D = 2, 15 parameters");
  if (kind == QMPS_ANSATZ_STATE_GATE && (c->D != 2 || n_params < 6))
    return fail(QMPS_ERR_ARG, "StateGate is a two-qubit gate: D = 2, 6 parameters");
  if (kind == QMPS_ANSATZ_EXACT_AFTER4 && n_params % 6) return fail(QMPS_ERR_ARG, "ExactAfter4 takes six angles per layer");
  if (kind == QMPS_ANSATZ_SHALLOW_CNOT_NONUNIFORM) {
    int nq = 1;
    while ((1 << (nq - 1)) < c->D) ++nq;      // n + 1 qubits
    if (n_params % (2 * nq)) return fail(QMPS_ERR_ARG, "ShallowCNOTStateTensor_nonuniform takes %d angles per layer at D = %d", 2 * nq, c->D);
  }
  if ((kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_QAOA) && n_params % 2)
    return fail(QMPS_ERR_ARG, "this ansatz takes (beta, gamma) pairs");
  if (kind == QMPS_ANSATZ_SHALLOW_CNOT3 && n_params % 3) return fail(QMPS_ERR_ARG, "this ansatz takes (beta, gamma, omega) triples");
  return QMPS_OK;
}

// d_A <- tensors of the resident ansatz parameters, if nothing has built them yet
int ensure_tensors(qmps_ctx* c) {
  if (c->tensors_valid) return QMPS_OK;
  if (!c->ans_have || c->ans_nsh != 0) return fail(QMPS_ERR_STATE, "no resident states");
  HIP_TRY(qmps::launch_ansatz(c->D, c->ans_kind, c->ans_src ? c->ans_src : c->d_params, c->ans_P, c->d_A, c->n_states, c->stream));
  c->tensors_valid = true;
  return QMPS_OK;
}

qmps::LaneArgs make_args(qmps_ctx* c, int64_t B, int max_iter, double tol, bool solve) {
  qmps::LaneArgs a;
  memset(&a, 0, sizeof(a));
  a.A = win_A(c);
  a.h = c->d_h;
  a.r_in = solve ? (c->have_guess ? win_r(c) : nullptr) : win_r(c);
  a.r_out = solve ? win_r(c) : nullptr;
  a.rho_out = c->want_rho ? (char*)c->d_rho + (size_t)c->window * 256 : nullptr;
  a.E = win_E(c);
  a.iters = win_iters(c);
  a.status = win_status(c);
  a.B = B;
  a.n_terms = c->n_terms;
  a.max_iter = max_iter;
  a.tol = tol;
  return a;
}

int close_group(qmps_ctx* c);
// QMPS_FLAG_ACCUMULATE_COST: point the energy kernel at the accumulator of the ring position the following
// qmps_cost_launch will use.  adds = arrivals per term (waves or evaluations that add one word each), per_add =
// evaluations behind one arrival (bounds the partial sum: per_add ||h||_F).
int setup_accumulator(qmps_ctx* c, qmps::LaneArgs& a, int64_t B, int64_t adds, int per_add) {
  // The position the following qmps_cost_launch will use: its accumulator must be clear BEFORE the finish kernel of
  // this step starts to poll it on a communication stream (a stale word of the previous lap carries a full arrival
  // count).  Consecutive ring slots alternate between two communication streams, so every launch clears the same
  // position TWO slots ahead: the finish kernel of this step (same stream as that later slot's) completes only when
  // every wave of this kernel - the clearing one included - has arrived, and the later slot's finish kernel is queued
  // behind it.
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots), pos = c->group_fill;
  const int nslot = (int)((c->groups + 2) % qmps_ctx::kCostSlots), npos = pos;
  c->acc_after_event[slot][pos] = false;
  if (c->acc_dirty[slot][pos]) {
    // unusual call order (exchange period changed, a partly filled group, an accumulated cost that was dropped): clear
    // it now on the compute stream, and order this position's finish kernel behind that by an event
    HIP_TRY(hipMemsetAsync(c->acc_at(slot, pos), 0, qmps::kAccWords * sizeof(long long), c->stream));
    c->acc_dirty[slot][pos] = false;
    c->acc_after_event[slot][pos] = true;
  }
  // a slot of the ring is touched again only after its previous exchange has finished.  Asked on the HOST (that
  // exchange, kCostSlots - 2 groups ago, has normally finished long ago): a stream wait would put a barrier packet
  // on the compute stream in every step (+4 us measured), and the compute stream carries no event either
#ifdef QMPS_DEBUG_KNOBS
  static const bool dbg_nohostwait = getenv("QMPS_DBG_NOHOSTWAIT") != nullptr;   // timing dissection only (unsafe slot reuse)
#else
  constexpr bool dbg_nohostwait = false;
#endif
  if (c->comm && !dbg_nohostwait && c->groups + 2 >= qmps_ctx::kCostSlots) {
    c->slot_checks++;
    if (hipEventQuery(c->cost_reduced[nslot]) != hipSuccess) {
      (void)hipGetLastError();
      timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      HIP_TRY(hipEventSynchronize(c->cost_reduced[nslot]));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      c->slot_blocks++;
      c->slot_block_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
  }
  int shards = 32;
  while (shards * (int64_t)qmps::kAccMaxWavesPerShard < adds && shards < qmps::kAccMaxShards) shards *= 2;
  if (shards * (int64_t)qmps::kAccMaxWavesPerShard < adds)
    return fail(QMPS_ERR_ARG, "B=%lld too large for QMPS_FLAG_ACCUMULATE_COST (at most %lld evaluations per launch on this path)", (long long)B,
                (long long)qmps::kAccMaxShards * qmps::kAccMaxWavesPerShard * per_add);
  a.acc = c->acc_at(slot, pos);
  a.acc_zero = c->acc_dirty[nslot][npos] ? c->acc_at(nslot, npos) : nullptr;
  a.acc_shards = shards;
  // partial sums (per_add evaluations each) beyond per_add ||h||_F bypass the fixed-point sum; scale 2^k with bound 2^k <= 2^51
  const double hf = c->h_fro > 1e-300 ? c->h_fro : 1.0;
  a.acc_bound = (double)per_add * hf * (1.0 + 1e-6);
  int k = (int)floor((double)qmps::kAccOffsetBits - 1e-9 - log2(a.acc_bound));
  if (k > 1000) k = 1000;
  if (k < -1000) k = -1000;
  a.acc_scale = ldexp(1.0, k);
  c->acc_shards[slot][pos] = shards;
  c->acc_expect[slot][pos] = adds;
  c->acc_scale[slot][pos] = a.acc_scale;
  c->acc_dirty[slot][pos] = true;
  c->acc_dirty[nslot][npos] = false;
  c->acc_pending = true; c->acc_B = B; c->acc_window = c->window; c->acc_slot = slot; c->acc_pos = pos;
  c->partials_B = -1;
  return QMPS_OK;
}

}  // namespace qmps_host

extern "C" {

int qmps_abi_version(void) { return QMPS_ABI_VERSION; }
int qmps_abi_minor(void) { return QMPS_ABI_MINOR; }

const char* qmps_last_error(void) { return g_err; }

// test hook for the "nothing throws across the ABI" contract (tests/test_cabi.py; needs no device): raises the exception a host
// allocation of absurd size would raise, inside the same function-try-block every other entry point has
int qmps_selftest_exception(int kind) try {
  if (kind == 1) throw std::bad_alloc();
  if (kind == 2) { std::vector<double> v; v.resize(v.max_size() + 1); return (int)v.size(); }     // std::length_error
  if (kind == 3) throw 42;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_device_count(int* count) try {
  if (!count) return fail(QMPS_ERR_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_device_info(int device, char* name, int name_len, char* arch, int arch_len, int* compute_units,
                     int64_t* hbm_bytes) try {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name && name_len > 0) snprintf(name, name_len, "%s", prop.name);
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_create(int device, int D, int64_t max_batch, qmps_ctx** out) try {
  if (!out) return fail(QMPS_ERR_ARG, "null out");
  *out = nullptr;
  if (D != 2 && D != 4 && D != 8 && D != 16) return fail(QMPS_ERR_ARG, "bond dimension D=%d not in {2,4,8,16}", D);
  if (max_batch < 1) return fail(QMPS_ERR_ARG, "max_batch must be >= 1");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return fail(QMPS_ERR_NO_DEVICE, "no HIP device visible: libqmps_hip has no CPU fallback");
  }
  if (device < 0 || device >= n) return fail(QMPS_ERR_ARG, "device %d outside [0,%d)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(QMPS_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device,
                prop.gcnArchName);
  qmps_ctx* c = new (std::nothrow) qmps_ctx();
  if (!c) return fail(QMPS_ERR_ARG, "out of host memory");
  c->device = device;
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->D = D;
  c->max_batch = max_batch;
  int rc = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    for (int i = 0; i < qmps_ctx::kRing; ++i) {
      HIP_TRY(hipEventCreate(&c->kev0[i]));
      HIP_TRY(hipEventCreate(&c->kev1[i]));
    }
    HIP_TRY(hipMalloc(&c->d_A, (size_t)max_batch * tensor_bytes(c)));
    HIP_TRY(hipMalloc(&c->d_r, (size_t)max_batch * env_bytes(c)));
    HIP_TRY(hipMalloc(&c->d_h, (size_t)kMaxTerms * 256));
    HIP_TRY(hipMalloc((void**)&c->d_iters, (size_t)(max_batch + 2) * sizeof(int32_t)));     // (+ 2: read in 8-byte units by the staging copy)
    HIP_TRY(hipMalloc((void**)&c->d_status, (size_t)(max_batch + 2) * sizeof(int32_t)));
    c->partial_cap = (max_batch + 15) / 16 > kSumBlocks ? (max_batch + 15) / 16 : kSumBlocks;   // one partial per 16 (direct kernel), 32 (pair kernel) or 64 items
    HIP_TRY(hipMalloc((void**)&c->d_partial, (size_t)kMaxTerms * c->partial_cap * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->d_cost, kMaxTerms * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->d_cost_ring, (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * kMaxTerms * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->d_cost_ring, 0, (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * kMaxTerms * sizeof(double), c->stream));
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream2, hipStreamNonBlocking));
    for (int i = 0; i < qmps_ctx::kCostSlots; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&c->cost_ready[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->cost_reduced[i], hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc((void**)&c->h_cost, kMaxTerms * sizeof(double), hipHostMallocDefault));
    const size_t acc_bytes = (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * qmps::kAccWords * sizeof(long long);
    HIP_TRY(hipMalloc((void**)&c->d_acc, acc_bytes));
    HIP_TRY(hipMemsetAsync(c->d_acc, 0, acc_bytes, c->stream));
    HIP_TRY(hipHostMalloc((void**)&c->h_acc, qmps::kAccWords * sizeof(long long), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&c->d_acc_err, sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->d_acc_err, 0, sizeof(int), c->stream));
    // counters of the Krylov fall-backs (overlap solves: qmps_capi_overlap.hip; D = 16 environment: qmps_energy_launch) - allocated here,
    // never at a launch (a launch may sit inside a stream capture); they clear themselves after use
    HIP_TRY(hipMalloc((void**)&c->d_queue, 16 * sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->d_queue, 0, 16 * sizeof(int), c->stream));
    HIP_TRY(hipMalloc((void**)&c->d_work_count, sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&c->d_work_idx, (size_t)max_batch * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->d_work_count, 0, sizeof(int32_t), c->stream));
    c->handoff = 0;   // D = 2, 4: squaring from the start (fastest); D = 8, 16 have no squaring path
    c->default_solver = (D == 2 || D == 4 || D == 8) ? QMPS_ENV_DIRECT : QMPS_ENV_POWER_SQUARING;
    c->skip_rounds = (D == 2) ? QMPS_SKIP_ROUNDS_D2 : QMPS_SKIP_ROUNDS_D4;
    if (const char* e = tuning_knob("QMPS_SKIP_ROUNDS")) c->skip_rounds = atoi(e);   // tuning knob
    if (const char* e = tuning_knob("QMPS_MATVEC_PERIOD")) c->matvec_period = atoi(e);   // tuning knob
    c->no_pair = tuning_knob("QMPS_NO_PAIR") != nullptr;
    c->pair_in_step = tuning_knob("QMPS_LANE_IN_STEP") == nullptr;
    return QMPS_OK;
  }();
  if (rc != QMPS_OK) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", g_err);
    qmps_destroy(c);
    snprintf(g_err, sizeof(g_err), "%s", keep);
    return rc;
  }
  *out = c;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_destroy(qmps_ctx* c) try {
  if (!c) return QMPS_OK;
  (void)hipSetDevice(c->device);
  for (qmps_ctx* g : c->lockstep) (void)qmps_destroy(g);        // (lock-step groups of qmps_evolve_bfgs)
  c->lockstep.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamSynchronize(c->comm_stream2);
  if (c->roto_exec) (void)hipGraphExecDestroy(c->roto_exec);
  if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
  if (c->comm2) (void)ncclCommDestroy(c->comm2);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  for (int i = 0; i < qmps_ctx::kCostSlots; ++i) {
    if (c->cost_ready[i]) (void)hipEventDestroy(c->cost_ready[i]);
    if (c->cost_reduced[i]) (void)hipEventDestroy(c->cost_reduced[i]);
  }
  if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
  for (hipEvent_t e : c->active_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
  if (c->aux_join) (void)hipEventDestroy(c->aux_join);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamDestroy(c->comm_stream2);
  void* bufs[] = {c->d_A, c->d_U, c->d_U2, c->d_params, c->d_ww, c->d_eta, c->d_ref, c->d_f, c->d_ostats, c->d_xwarm, c->d_y, c->d_queue, c->d_kry, c->d_active, c->d_scratch, c->d_h, c->d_r, c->d_rho, c->d_E, c->d_iters, c->d_status, c->d_partial, c->d_cost, c->d_cost_ring, c->d_work_count, c->d_work_idx, c->d_acc, c->d_acc_err, c->roto_base, c->roto_hist, c->roto_idx};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->h_cost) (void)hipHostFree(c->h_cost);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_mask) (void)hipHostFree(c->h_mask);
  if (c->d_lock) (void)hipFree(c->d_lock);
  if (c->h_ctl) (void)hipHostFree(c->h_ctl);
  if (c->step_ev0) (void)hipEventDestroy(c->step_ev0);
  if (c->step_ev1) (void)hipEventDestroy(c->step_ev1);
  if (c->d_tolarr) (void)hipFree(c->d_tolarr);
  if (c->h_acc) (void)hipHostFree(c->h_acc);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < qmps_ctx::kRing; ++i) {
    if (c->kev0[i]) (void)hipEventDestroy(c->kev0[i]);
    if (c->kev1[i]) (void)hipEventDestroy(c->kev1[i]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_sync(qmps_ctx* c) try {
  if (int rc = bind(c)) return rc;
  if (int rc = close_group(c)) return rc;     // costs still waiting for their exchange go out now
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->comm_stream));
  HIP_TRY(hipStreamSynchronize(c->comm_stream2));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_states(qmps_ctx* c, int64_t B, const double* states, int kind) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!states && B > 0) return fail(QMPS_ERR_ARG, "null states");
  if (kind == QMPS_INPUT_TENSOR) {
    HIP_TRY(hipMemcpyAsync(c->d_A, states, (size_t)B * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
  } else if (kind == QMPS_INPUT_UNITARY) {
    if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * 2 * tensor_bytes(c)));
    HIP_TRY(hipMemcpyAsync(c->d_U, states, (size_t)B * 2 * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(qmps::launch_unitary_to_tensor(c->d_U, c->d_A, c->D, B, c->stream));
  } else {
    return fail(QMPS_ERR_ARG, "unknown input kind %d", kind);
  }
  // the caller's host buffer may be pageable and re-used right after the call returns
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_states = B;
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  c->ans_have = false;
  c->tensors_valid = true;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_window(qmps_ctx* c, int64_t first) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (first < 0 || first > c->n_states) return fail(QMPS_ERR_ARG, "window start %lld outside the %lld resident states", (long long)first, (long long)c->n_states);
  c->window = first;
  c->partials_B = -1;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_states_ansatz(qmps_ctx* c, int64_t B, int kind, int n_params, const double* params) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (n_params > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * n_params * sizeof(double)));
    c->params_cap = n_params;
  }
  {
    const size_t pb = (size_t)B * n_params * sizeof(double);
    if (c->defer_sync && pb <= (8u << 20)) {
      // one-round-trip callers: through pinned memory, moved by a kernel on the context stream (no copy-queue hop)
      if (int rc = ensure_pinned(c, (16u << 20))) return rc;
      memcpy(c->h_pin, params, pb);
      if (c->mask_stash_n > 0) {         // a mask of qmps_overlap_set_active waiting in its staging slot rides along
        const int64_t n = c->mask_stash_n;
        c->mask_stash_n = 0;
        HIP_TRY(qmps::launch_stage_copy2(c->h_pin, c->d_params, (int64_t)(pb / 8), c->mask_stash, c->d_active, (n + 7) / 8, c->stream));
      } else {
        HIP_TRY(qmps::launch_stage_copy(c->h_pin, c->d_params, (int64_t)(pb / 8), c->stream));
      }
    } else {
      HIP_TRY(hipMemcpyAsync(c->d_params, params, pb, hipMemcpyHostToDevice, c->stream));
    }
    if (c->fork_after_copy) {            // qmps_overlap_gradient: its second stream needs the parameters only
      HIP_TRY(hipEventRecord(c->fork_after_copy, c->stream));
      c->fork_after_copy = nullptr;
    }
  }
  c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
  c->tensors_valid = false;
  c->n_states = B;
  // D = 4: the direct kernel builds the tensors itself (8 P bytes per evaluation instead of 512); d_A is filled on demand
  if (!fusable_ansatz(c, kind))
    if (int rc = ensure_tensors(c)) return rc;
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_states_su(qmps_ctx* c, int64_t B, const double* params) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  const int N = 2 * c->D, np_ = N * N - 1;
  if (np_ > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * np_ * sizeof(double)));
    c->params_cap = np_;
  }
  HIP_TRY(hipMemcpyAsync(c->d_params, params, (size_t)B * np_ * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_su_exp(N, c->d_params, B, np_, c->d_A, 1, c->stream));
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_states = B;
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  c->ans_have = false;
  c->tensors_valid = true;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_energy_batch_su(qmps_ctx* c, int64_t B, const double* params, const double* h, int n_terms, int max_iter, double tol,
                         double* E_out, int32_t* iters_out, int32_t* status_out) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  int rc;
  {
    Restore<bool> deferred(c->defer_sync, true);
    rc = qmps_set_states_su(c, B, params);
    if (!rc) rc = qmps_set_hamiltonian(c, n_terms, h);
    if (!rc) rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver | QMPS_FLAG_KRYLOV_FALLBACK);
  }
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}
QMPS_API_CATCH

int qmps_su_unitaries(qmps_ctx* c, int64_t B, int N, const double* params, double* U_out) try {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !params || !U_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (N != 4 && N != 8 && N != 16 && N != 32) return fail(QMPS_ERR_ARG, "N=%d not in {4, 8, 16, 32}", N);
  const size_t pb = (size_t)B * (N * N - 1) * sizeof(double), ub = (size_t)B * N * N * 16;
  if (int rc = ensure_scratch(c, ((pb + 255) & ~(size_t)255) + ub + 256)) return rc;
  double* d_p = (double*)c->d_scratch;
  void* d_u = (char*)c->d_scratch + ((pb + 255) & ~(size_t)255);
  HIP_TRY(hipMemcpyAsync(d_p, params, pb, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_su_exp(N, d_p, B, N * N - 1, d_u, 0, c->stream));
  HIP_TRY(hipMemcpyAsync(U_out, d_u, ub, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

namespace {
int rotosolve_impl(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter, double tol,
                   double* E_hist, int nsh);
}

int qmps_rotosolve(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                   double tol, double* E_hist) try {
  return rotosolve_impl(c, R, kind, n_params, params, n_sweeps, max_iter, tol, E_hist, 3);
}
QMPS_API_CATCH

int qmps_double_rotosolve(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                          double tol, double* E_hist) try {
  return rotosolve_impl(c, R, kind, n_params, params, n_sweeps, max_iter, tol, E_hist, 6);
}
QMPS_API_CATCH

namespace {
int rotosolve_impl(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter, double tol,
                   double* E_hist, int nsh) {
  if (int rc = bind(c)) return rc;
  if (R < 1 || nsh * R > c->max_batch) return fail(QMPS_ERR_ARG, "%d R = %lld evaluations exceed max_batch = %lld", nsh, (long long)(nsh * R), (long long)c->max_batch);
  c->window = 0;
  if (!params || !E_hist) return fail(QMPS_ERR_ARG, "null argument");
  if (n_sweeps < 1) return fail(QMPS_ERR_ARG, "n_sweeps must be >= 1");
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (n_params > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * n_params * sizeof(double)));
    c->params_cap = n_params;
  }
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
  if (int rc = grow(c->roto_base, c->roto_base_bytes, (size_t)R * n_params * sizeof(double))) return rc;
  // The run's results (final parameters, energy history) come back through the context's pinned buffer when they fit: the
  // first LARGE copy into pageable memory makes the runtime set up its internal staging, ~8 ms once per process (measured
  // in the first 160-sweep call after an 8-sweep one: 27.7 instead of 19.5 us per parameter update at D = 8).
  auto download_results = [&](size_t hist_doubles) -> int {
    const size_t pb = (size_t)R * n_params * sizeof(double), hb = hist_doubles * sizeof(double);
    if (pb + hb <= (2u << 20)) {      // (small results only: a 7 MB history copied twice cost the D = 4 run of 21 845 restarts 16 %)
      if (int e = ensure_pinned(c, (16u << 20))) return e;
      HIP_TRY(hipMemcpyAsync(c->h_pin, c->roto_base, pb, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_pin + pb, c->roto_hist, hb, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      memcpy(params, c->h_pin, pb);
      memcpy(E_hist, c->h_pin + pb, hb);
      return QMPS_OK;
    }
    HIP_TRY(hipMemcpyAsync(params, c->roto_base, pb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(E_hist, c->roto_hist, hb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMPS_OK;
  };
#ifdef QMPS_D8_PROFILE        // scratch instrumentation build (profiles/experiments/scratch/d8_profile.py): 16 phase clocks behind the history
  constexpr size_t kHistExtra = 16 + 3 * 4096;
#else
  constexpr size_t kHistExtra = 0;
#endif
#ifdef QMPS_D8_PROFILE
  const auto g0 = std::chrono::steady_clock::now();
#endif
  if (int rc = grow(c->roto_hist, c->roto_hist_bytes, ((size_t)R * n_sweeps + kHistExtra) * sizeof(double))) return rc;
#ifdef QMPS_D8_PROFILE
  fprintf(stderr, "[d8 profile] history buffer: %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count());
#endif
  if (!c->roto_idx) HIP_TRY(hipMalloc((void**)&c->roto_idx, 4 * sizeof(int)));
  double *d_base = c->roto_base, *d_hist = c->roto_hist;
  int* d_idx = c->roto_idx;
  int rc = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(d_base, params, (size_t)R * n_params * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_idx, 0, 3 * sizeof(int), c->stream));   // parameter index, arrival counter, finished sweeps
    if (kHistExtra) HIP_TRY(hipMemsetAsync(d_hist + (size_t)R * n_sweeps, 0, kHistExtra * sizeof(double), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const bool saved_guess = c->have_guess;
    c->have_guess = false;
    // D = 2 with the library's default solver: the whole run is ONE launch (restarts are independent, see
    // rotosolve_fused_d2_kernel); afterwards one ordinary evaluation of the final parameters leaves the context's
    // resident tensors / energies / statuses exactly as the step-by-step path does.
    if ((nsh == 3 || nsh == 6) && c->D == 2 && c->handoff == 0 && (c->default_solver == QMPS_ENV_POWER_SQUARING || c->default_solver == QMPS_ENV_DIRECT) && n_params <= 64 &&
        documented_switch("QMPS_NO_FUSED_ROTO") == nullptr) {
      qmps::RotoArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.base = d_base; ra.h = c->d_h; ra.hist = d_hist;
      ra.R = (int)R; ra.P = n_params; ra.n_terms = c->n_terms; ra.n_sweeps = n_sweeps; ra.max_iter = max_iter;
      ra.skip = c->skip_rounds; ra.tol = tol; ra.direct = c->default_solver == QMPS_ENV_DIRECT ? 1 : 0; ra.nsh = nsh; ra.rule = c->roto_rule;
      HIP_TRY(qmps::launch_rotosolve_fused_d2(kind, ra, c->stream));
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, n_params, c->d_A, R, c->stream));
      c->n_states = R; c->ans_have = false; c->tensors_valid = true;
      if (int e = qmps_energy_launch(c, R, max_iter, tol, c->default_solver)) return e;
      c->have_guess = saved_guess;
      return download_results((size_t)R * n_sweeps);
    }
    // D = 8 (ShallowCNOT families, direct solver): the whole run in ONE launch as well - a workgroup per restart, a wave per
    // shift (qmps_roto_d8.hip); afterwards one ordinary evaluation of the final parameters, as above
    // (six shifts: every wave evaluates two of them in turn).  A restart occupies a CU for the whole run, so this is the path of
    // the SMALL runs (BASELINE.json configs[3]: 256 restarts): measured against the step-by-step path below, us per update,
    // three shifts: R = 256: 19.9 / 34, 512: 40.8 / 45.4, 1 024: 77 / 63, 21 845: 1 552 / 785; six shifts: R = 128: 40.6 / 35.6, 256: 41.0 / 44.1.
    const bool d8_fused_pays = nsh == 3 ? R <= 512 : (R <= 256 && 6 * R > 1024);
    if (c->D == 8 && (nsh == 3 || nsh == 6) && d8_fused_pays && c->default_solver == QMPS_ENV_DIRECT && (kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_CNOT3) &&
        n_params <= 64 && documented_switch("QMPS_NO_FUSED_ROTO") == nullptr) {
      qmps::RotoArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.base = d_base; ra.h = c->d_h; ra.hist = d_hist;
      ra.R = (int)R; ra.P = n_params; ra.n_terms = c->n_terms; ra.n_sweeps = n_sweeps; ra.max_iter = max_iter;
      ra.tol = tol; ra.direct = 1; ra.nsh = nsh; ra.rule = c->roto_rule;
#ifdef QMPS_D8_PROFILE
      auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      const double h0 = now();
#endif
      HIP_TRY(qmps::launch_rotosolve_fused_d8(kind, ra, c->stream));
#ifdef QMPS_D8_PROFILE
      const double h1 = now();
      HIP_TRY(hipStreamSynchronize(c->stream));
      const double h2 = now();
#endif
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, n_params, c->d_A, R, c->stream));
      c->n_states = R; c->ans_have = false; c->tensors_valid = true;
      if (int e = qmps_energy_launch(c, R, max_iter, tol, c->default_solver)) return e;
      c->have_guess = saved_guess;
      if (int e = download_results((size_t)R * n_sweeps + kHistExtra)) return e;
#ifdef QMPS_D8_PROFILE
      fprintf(stderr, "[d8 profile] launch call %.0f us, kernel until sync %.0f us, final evaluation + downloads %.0f us\n", h1 - h0, h2 - h1, now() - h2);
#endif
      return QMPS_OK;
    }
    // One parameter update = shift build -> ansatz -> environment + energy -> closed-form update.  The
    // parameter index lives in HBM and is advanced by the update kernel, so the sequence is captured ONCE
    // into a hipGraph and replayed n_params x n_sweeps times: the sweep is launch-bound at small R.
    // D = 4 with the direct solver: shift build and ansatz happen INSIDE the energy kernel (evaluation nsh r + k builds
    // the tensor of restart r with shift k on parameter *d_idx straight into LDS): two kernels per parameter update
    const bool fused = c->default_solver == QMPS_ENV_DIRECT && fusable_ansatz(c, kind);
    auto evaluate = [&](int shifts) -> int {      // shifts = nsh: the shifted batch;  0: the R base vectors
      const int64_t n = shifts > 0 ? (int64_t)shifts * R : R;
      if (fused) {
        c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = d_base; c->ans_i = d_idx; c->ans_nsh = shifts;
        c->tensors_valid = false;
      } else {
        // shifted tensors straight from the base vectors (the shift build is folded into the ansatz kernel)
        HIP_TRY(qmps::launch_ansatz_shifted(c->D, kind, d_base, n_params, c->d_A, n, shifts, d_idx, c->stream));
        c->ans_have = false; c->tensors_valid = true;
      }
      c->n_states = n;
      return qmps_energy_launch(c, n, max_iter, tol, c->default_solver);
    };
    auto one_update = [&](bool first_of_sweep) -> int {
      if (int e = evaluate(nsh)) return e;
      // the shift-0 row of a sweep's first batch is the evaluation of the vectors the PREVIOUS sweep left: its record
      if (first_of_sweep) HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)R, c->n_terms, d_idx + 2, nsh, c->stream));
      HIP_TRY(qmps::launch_roto_update(d_base, c->d_E, c->d_status, (int)R, n_params, d_idx, c->n_terms, nsh, c->roto_rule, c->stream));
      return QMPS_OK;
    };
    // One sweep = n_params updates (the first one also records the previous sweep from its shift-0 rows).  The parameter index and the
    // sweep counter live in HBM and are advanced by the update kernel, so the sweep is captured ONCE into a hipGraph and
    // replayed n_sweeps times (a graph launch costs ~15 us: per update it was a third of the time, per sweep it is noise)
    auto one_sweep = [&]() -> int {
      for (int i = 0; i < n_params; ++i)
        if (int e = one_update(i == 0)) return e;
      return QMPS_OK;
    };
    const bool use_graph = documented_switch("QMPS_NO_GRAPH") == nullptr && n_params <= 256;
    if (use_graph) {
      qmps_ctx::RotoKey key;
      key.R = R; key.kind = kind; key.P = n_params; key.nsh = nsh; key.max_iter = max_iter; key.n_terms = c->n_terms;
      key.solver = c->default_solver; key.handoff = c->handoff; key.rule = c->roto_rule; key.tol = tol; key.fused = fused;
      key.base = d_base; key.hist = d_hist; key.params = c->d_params; key.E = c->d_E;
      if (!(c->roto_exec && key == c->roto_key)) {
        if (c->roto_exec) (void)hipGraphExecDestroy(c->roto_exec);
        if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
        c->roto_exec = nullptr; c->roto_graph = nullptr;
        c->capturing = true;
        HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int e = one_sweep();
        const hipError_t ce = hipStreamEndCapture(c->stream, &c->roto_graph);
        c->capturing = false;
        if (e) return e;
        HIP_TRY(ce);
        HIP_TRY(hipGraphInstantiate(&c->roto_exec, c->roto_graph, nullptr, nullptr, 0));
        c->roto_key = key;
      }
    }
    for (int sw = 0; sw < n_sweeps; ++sw) {
      if (use_graph) HIP_TRY(hipGraphLaunch(c->roto_exec, c->stream));
      else if (int e = one_sweep()) return e;
    }
    // the last sweep's record, and the resident state the call leaves: one evaluation of the final vectors
    if (int e = evaluate(0)) return e;
    HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)R, c->n_terms, d_idx + 2, 1, c->stream));
    // The context's view of what is resident - a replayed graph runs no host code, so it is stated here, not inherited from
    // the capture: the R final parameter vectors, their energies / statuses / environments
    c->n_states = R;
    c->window = 0;
    c->have_env = true;
    c->partials_B = -1;
    c->acc_pending = false;
    if (fused) {
      // as a qmps_set_states_ansatz of the final parameters would leave it: rows resident in d_params, tensors on demand
      HIP_TRY(hipMemcpyAsync(c->d_params, d_base, (size_t)R * n_params * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
      c->tensors_valid = false;
    } else {
      c->ans_have = false; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
      c->tensors_valid = true;
    }
    c->have_guess = saved_guess;
    return download_results((size_t)R * n_sweeps);
  }();
  c->capturing = false;
  (void)hipStreamSynchronize(c->stream);
  if (c->ans_src != nullptr) {     // an error left the context pointing at the run's own buffers
    c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0; c->ans_have = false; c->tensors_valid = true; c->n_states = 0;
  }
  if (rc != QMPS_OK && c->roto_exec) {     // do not trust a sweep captured by a failed run
    (void)hipGraphExecDestroy(c->roto_exec);
    if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
    c->roto_exec = nullptr; c->roto_graph = nullptr;
  }
  return rc;
}
}  // namespace

int qmps_get_states(qmps_ctx* c, int64_t B, double* A) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!A) return fail(QMPS_ERR_ARG, "null A");
  if (B > c->n_states) return fail(QMPS_ERR_STATE, "only %lld states are resident", (long long)c->n_states);
  if (int rc = ensure_tensors(c)) return rc;
  HIP_TRY(hipMemcpyAsync(A, c->d_A, (size_t)B * tensor_bytes(c), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_hamiltonian(qmps_ctx* c, int n_terms, const double* h) try {
  if (int rc = bind(c)) return rc;
  if (n_terms < 1 || n_terms > kMaxTerms) return fail(QMPS_ERR_ARG, "n_terms=%d outside [1,%d]", n_terms, kMaxTerms);
  if (!h) return fail(QMPS_ERR_ARG, "null h");
  if (int rc = ensure_E(c, n_terms)) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_h, h, (size_t)n_terms * 256, hipMemcpyHostToDevice, c->stream));
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->h_fro = 0.0;
  for (int t = 0; t < n_terms; ++t) {
    double f = 0.0;
    for (int i = 0; i < 32; ++i) f += h[32 * t + i] * h[32 * t + i];
    f = sqrt(f);
    if (f > c->h_fro) c->h_fro = f;
  }
  if (n_terms != c->n_terms) {
    // The in-kernel clear of a cost accumulator covers the CURRENT number of terms only: after a change of that number a slot that counts as clean
    // may still hold the arrivals of a term it was last used with ("cost accumulator: 46 of 23 waves arrived" on the first accumulating launch after
    // going from one Hamiltonian term to two; found by profiles/experiments/r05/stress_api_state.py, round 5).  Every position of the ring is marked
    // dirty: setup_accumulator clears a dirty position completely before it is used.
    for (int sl = 0; sl < qmps_ctx::kCostSlots; ++sl)
      for (int ps = 0; ps < qmps_ctx::kMaxGroup; ++ps) c->acc_dirty[sl][ps] = true;
  }
  c->n_terms = n_terms;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_env_guess(qmps_ctx* c, int64_t B, const double* r0) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  c->window = 0;
  if (!r0) {
    c->have_guess = false;
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(c->d_r, r0, (size_t)B * env_bytes(c), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->have_guess = true;
  c->have_env = true;
  c->have_overlap_x = false;
  c->grad_warm_T = 0;          // d_r no longer holds the right fixed points of a gradient batch
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_energy_launch(qmps_ctx* c, int64_t B, int max_iter, double tol, int flags) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (max_iter < 1) return fail(QMPS_ERR_ARG, "max_iter must be >= 1");
  if (!(tol > 0.0)) return fail(QMPS_ERR_ARG, "tol must be > 0");
  int solver = flags & 0xff;
  if (solver != QMPS_ENV_POWER && solver != QMPS_ENV_POWER_SQUARING && solver != QMPS_ENV_DIRECT)
    return fail(QMPS_ERR_ARG, "unknown environment solver %d", solver);
  if ((flags & ~0xff) & ~(QMPS_FLAG_NO_ENV_OUT | QMPS_FLAG_ACCUMULATE_COST | QMPS_FLAG_WARM_RESIDENT | QMPS_FLAG_KRYLOV_FALLBACK)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags & ~0xff);
  const bool warm_resident = (flags & QMPS_FLAG_WARM_RESIDENT) != 0;
  if (warm_resident && !c->have_env) return fail(QMPS_ERR_STATE, "QMPS_FLAG_WARM_RESIDENT: no resident environments (run a launch that stores them, or qmps_set_env_guess)");
  const bool direct = solver == QMPS_ENV_DIRECT && c->D == 4;
  if ((flags & QMPS_FLAG_NO_ENV_OUT) && !direct) return fail(QMPS_ERR_ARG, "QMPS_FLAG_NO_ENV_OUT needs QMPS_ENV_DIRECT at D = 4");
  const bool accumulate = (flags & QMPS_FLAG_ACCUMULATE_COST) != 0;
  if (accumulate && c->D == 4 && !direct && solver == QMPS_ENV_POWER_SQUARING)
    return fail(QMPS_ERR_ARG, "QMPS_FLAG_ACCUMULATE_COST: at D = 4 use QMPS_ENV_DIRECT or QMPS_ENV_POWER");
  if ((flags & QMPS_FLAG_ACCUMULATE_COST) && c->acc_pending)
    return fail(QMPS_ERR_STATE, "the cost accumulated by the previous launch has not been consumed by qmps_cost_launch");
  if ((flags & QMPS_FLAG_ACCUMULATE_COST) && c->capturing) return fail(QMPS_ERR_STATE, "no cost accumulation inside a graph capture");
  const bool direct8 = solver == QMPS_ENV_DIRECT && c->D == 8;
  const bool direct2 = solver == QMPS_ENV_DIRECT && c->D == 2;     // 4 x 4 solve in the lane, in front of the squaring tail
  if (solver == QMPS_ENV_DIRECT && !direct) solver = QMPS_ENV_POWER_SQUARING;   // D = 2: the lane kernel's squaring path with the 4 x 4 solve in front (direct2); D = 16 iterates (documented)
  c->acc_pending = false;   // whatever an earlier launch accumulated no longer describes the resident energies
  c->have_overlap_x = false;   // d_r is about to hold environments, not overlap fixed points
  c->grad_warm_T = 0;
  const bool fused = direct && c->ans_have && fusable_ansatz(c, c->ans_kind);
  if (!fused)
    if (int rc = ensure_tensors(c)) return rc;
  qmps::LaneArgs a = make_args(c, B, max_iter, tol, true);
  if (warm_resident) a.r_in = win_r(c);
  if (direct8) {
    // D = 8: the direct solve (one wave per evaluation) hands its result to the power iteration of the block kernel: its
    // first step is the acceptance test, its loop the fall-back.  Small batches (all launch latency: BASELINE configs[3]
    // is 96 evaluations per GPU) run both in ONE launch; large ones keep two kernels - the block kernel alone runs four
    // waves per SIMD, the solve two.
    static const int64_t fuse_below = tuning_knob("QMPS_D8_FUSE_BELOW") ? atoll(tuning_knob("QMPS_D8_FUSE_BELOW")) : 4096;   // A/B knob
    if (B <= fuse_below) {
      a.direct = 1;
      a.r_in = nullptr;
    } else {
      HIP_TRY(qmps::launch_env_direct_d8(win_A(c), win_r(c), B, c->stream));
      a.r_in = win_r(c);
    }
  }
  const bool hybrid = solver == QMPS_ENV_POWER_SQUARING && c->D <= 4 && c->handoff < max_iter;
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int slot = (int)(c->samples % qmps_ctx::kRing);
  c->partials_B = -1;
  const int lane_waves = (int)((B + 63) / 64);
  if (direct) {
    // D = 4: direct fixed-point solve + acceptance power step + energies in ONE kernel (a DPP quad per evaluation);
    // one read of A, one store of E (and, unless switched off, of r) per evaluation
    // a.r_in (qmps_set_env_guess / QMPS_FLAG_WARM_RESIDENT): evaluations whose guess passes the acceptance test skip the solve
    a.r_out = (flags & QMPS_FLAG_NO_ENV_OUT) ? nullptr : win_r(c);
    if (fused) {
      const double* rows = c->ans_src ? c->ans_src : c->d_params;
      a.ans_params = c->ans_nsh > 0 ? rows : rows + (size_t)c->window * c->ans_P;
      a.ans_P = c->ans_P; a.ans_kind = c->ans_kind; a.ans_nsh = c->ans_nsh; a.ans_i = c->ans_i;
    }
    if (flags & QMPS_FLAG_ACCUMULATE_COST) {
      if (int rc = setup_accumulator(c, a, B, (B + 15) / 16, 16)) return rc;
    } else {
      a.partial = c->d_partial; c->partials_B = B; c->partials_n = (int)((B + 15) / 16);
    }
    c->dominant = "energy_direct_d4_kernel";
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy_direct_d4(a, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
    if (c->timed) c->samples++;
    if (!c->capturing) c->launches++;
    if (a.r_out != nullptr) c->have_env = true;
    else if (a.r_in == nullptr) c->have_env = false;      // (a warm launch that stores nothing leaves the resident guesses in place)
    return QMPS_OK;
  }
  if (c->D == 16 && !documented_switch("QMPS_D16_BLOCK")) {
    // D = 16: power iteration on the matrix cores (one wave per evaluation), then the energy pass
    c->dominant = "energy_mfma_d16_kernel<true>";
    if (accumulate) if (int rc = setup_accumulator(c, a, B, B, 1)) return rc;
    // Krylov fall-back of the environment solve (include/qmps_hip.h "fixed-point solvers"; QMPS_NO_KRYLOV: power iteration alone):
    // evaluations whose power iteration predicts a long tail (|lambda_2| -> 1: shallow circuits) are finished by the Arnoldi kernel
    // of qmps_overlap_krylov.hip on the environment map, then accepted - energy, Cholesky test, status - by a pass of the same energy kernel
    const bool krylov = documented_switch("QMPS_NO_KRYLOV") == nullptr && max_iter > 64;
    if (krylov) {
      a.krylov_after = 256;
      if (const char* e = tuning_knob("QMPS_KRYLOV_AFTER")) a.krylov_after = atoi(e);
      a.kry_counter = c->d_queue + 8;
    }
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy_mfma(c->D, a, true, c->stream));
    if (krylov && a.krylov_after > 0) {
      qmps::OverlapArgs k;
      memset(&k, 0, sizeof(k));
      k.Bt = a.A; k.r_out = a.r_out; k.iters = a.iters; k.status = a.status; k.B = B; k.max_rounds = max_iter; k.tol = tol;
      k.env_mode = 1; k.krylov_after = a.krylov_after; k.kry_counter = a.kry_counter;
      HIP_TRY(qmps::launch_overlap_krylov(16, k, k.kry_counter, c->stream));
      qmps::LaneArgs f = a;
      f.r_in = a.r_out; f.only_pending = 1; f.krylov_after = 0; f.acc_zero = nullptr;
      f.max_iter = 64;
      HIP_TRY(qmps::launch_energy_mfma(c->D, f, true, c->stream));
    }
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else if (!hybrid && c->D == 4 && solver == QMPS_ENV_POWER && c->d_queue != nullptr && documented_switch("QMPS_POWER_LANE") == nullptr) {
    // D = 4, plain power iteration (round 6): a 16-lane DPP row per evaluation (the map as a real 16 x 16 matrix in registers, a step = sixteen
    // v_fmac_f64_dpp) in persistent waves that draw their evaluations from a counter - env_power_d4_kernel (qmps_direct.hip) - then the energies, the Cholesky test and the cost sums on the stored environments
    // (energy_only_d4_kernel).  The lane-per-evaluation kernel of rounds 1-5 waited for the slowest of its 64 evaluations in every wave
    // and for ONE evaluation per launch (QMPS_POWER_LANE=1 selects it: same iterates, same iteration counts).
    int* counter = c->d_queue + 13;
    HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), c->stream));
    qmps::LaneArgs e = make_args(c, B, 1, 1.0, false);
    e.check_pd = 1;
    if (accumulate) {
      if (int rc = setup_accumulator(c, e, B, (B + 15) / 16, 16)) return rc;
    } else { e.partial = c->d_partial; c->partials_B = B; c->partials_n = (int)((B + 15) / 16); }
    int waves_per_simd = 5;          // (what fits: 96 registers)
    if (const char* w = tuning_knob("QMPS_POWER_WAVES")) waves_per_simd = atoi(w);
    c->dominant = "env_power_d4_kernel";
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_env_power_d4(a, counter, c->n_cus * 4 * waves_per_simd, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
    HIP_TRY(qmps::launch_energy_only_d4(e, c->stream));
  } else if (!hybrid) {
    c->dominant = c->D <= 4 ? "energy_lane_kernel<D,true>" : "energy_block_kernel<D,true>";
    if (accumulate) {
      if (int rc = setup_accumulator(c, a, B, c->D <= 4 ? lane_waves : B, c->D <= 4 ? 64 : 1)) return rc;
    } else if (c->D <= 4) { a.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves; }
    // D = 8 (round 5): the power loop behind a direct solve that was not accepted - an elimination without pivoting meets structural zeros at special
    // angles of the ansatz - is the only fall-back of the energy path whose cost grows with 1 / gap (D = 2, 4 square, D = 16 hands over): evaluations
    // whose residual history predicts a long tail go to the Arnoldi kernel on the environment map, then through a finishing pass of the block kernel,
    // exactly as at D = 16.  QMPS_ENV_POWER stays the plain iteration (a-13: the classical statement of PowerCircuit); QMPS_NO_KRYLOV switches it off.
    // On request only (QMPS_FLAG_KRYLOV_FALLBACK; the one-shot entry points set it): the two extra launches - nearly always empty - cost 3.1 - 3.4 us of a
    // 16 - 20 us step of resident tensors (B = 96 / 768, measured), nothing next to the round trips of a one-shot call.
    const bool krylov8 = c->D == 8 && (flags & QMPS_FLAG_KRYLOV_FALLBACK) != 0 && solver != QMPS_ENV_POWER && documented_switch("QMPS_NO_KRYLOV") == nullptr && max_iter > 64 && c->d_queue != nullptr && a.r_out != nullptr;
    if (krylov8) {
      a.krylov_after = 256;
      if (const char* e = tuning_knob("QMPS_KRYLOV_AFTER")) a.krylov_after = atoi(e);
      a.kry_counter = c->d_queue + 8;
    }
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
    if (krylov8 && a.krylov_after > 0) {
      qmps::OverlapArgs k;
      memset(&k, 0, sizeof(k));
      k.Bt = a.A; k.r_out = a.r_out; k.iters = a.iters; k.status = a.status; k.B = B; k.max_rounds = max_iter; k.tol = tol;
      k.env_mode = 1; k.krylov_after = a.krylov_after; k.kry_counter = a.kry_counter;
      HIP_TRY(qmps::launch_overlap_krylov(8, k, k.kry_counter, c->stream));
      qmps::LaneArgs f = a;
      f.r_in = a.r_out; f.only_pending = 1; f.krylov_after = 0; f.acc_zero = nullptr; f.direct = 0;
      f.max_iter = 64;
      HIP_TRY(qmps::launch_energy(c->D, f, true, c->stream));
    }
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else if (c->D == 2) {
    a.handoff = c->handoff;  // the squaring tail runs in-lane (real 4 x 4 transfer matrix in registers)
    a.hybrid = 1;
    a.direct = direct2 ? 1 : 0;
    a.skip = c->handoff == 0 ? c->skip_rounds : 0;
    if (accumulate) {
      if (int rc = setup_accumulator(c, a, B, lane_waves, 64)) return rc;
    } else { a.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves; }
    c->dominant = "energy_lane_kernel<2,true>";
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else {
    // D = 4: (1) lane kernel: `handoff` plain steps, slow items -> worklist (skipped when handoff == 0:
    // every item goes straight to the squaring kernel); (2) wave-per-item MFMA squaring over the
    // worklist; (3) energy-only pass over the worklist.  No host round trip: the later kernels read the
    // item count from HBM.
    qmps::SquareArgs q;
    memset(&q, 0, sizeof(q));
    q.A = win_A(c); q.r_out = win_r(c); q.iters = win_iters(c); q.status = win_status(c);
    q.B = B; q.done = c->handoff; q.max_iter = max_iter; q.tol = tol;
    q.skip = c->handoff == 0 ? c->skip_rounds : 0;
    q.period = c->matvec_period;
    qmps::LaneArgs e = make_args(c, B, 1, 1.0, false);
    e.check_pd = 1;
    if (c->handoff > 0) {
      HIP_TRY(hipMemsetAsync(c->d_work_count, 0, sizeof(int32_t), c->stream));
      a.handoff = c->handoff;
      a.hybrid = 1;
      a.work_count = c->d_work_count;
      a.work_idx = c->d_work_idx;
      c->dominant = "energy_lane_kernel<4,true>";
      if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
      HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
      if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
      q.r_in = win_r(c);
      q.work_count = c->d_work_count;
      q.work_idx = c->d_work_idx;
      e.idx_list = c->d_work_idx;
      e.idx_count = c->d_work_count;
    } else {
      q.r_in = c->have_guess ? win_r(c) : nullptr;
      e.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves;   // the energy pass covers every item
    }
    // grid-stride workgroups of 4 waves: whole generations of the resident capacity (5 workgroups per CU), at most three
    // (measured at B = 65536 with settled clocks: 1280 / 2560 / 3840 / 5120 workgroups -> 0.0876 / 0.0870 / 0.0859 / 0.0875 ms;
    // 2048 and 3072, which end in a partial generation, 0.0900 and 0.0878)
    int grid = (int)((B + 15) / 16);
    const int generation = c->n_cus * 5;
    if (grid > generation) {
      grid = (grid / generation) * generation;
      if (grid > 3 * generation) grid = 3 * generation;
    }
    if (const char* e = tuning_knob("QMPS_SQ_GRID")) grid = atoi(e) < grid ? atoi(e) : grid;   // tuning knob
    if (grid < 1) grid = 1;
    if (c->handoff == 0) {
      c->dominant = "env_square_d4_kernel";
      if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    }
    HIP_TRY(qmps::launch_square_tail(c->D, q, grid, c->stream));
    if (c->handoff == 0) if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
    // (with settled clocks the step is 0.9 % shorter with the pair kernel: 0.1112 against 0.1122 ms at B = 65536;
    // QMPS_LANE_IN_STEP keeps the one-lane pass)
    if (e.idx_list == nullptr && !c->no_pair && c->pair_in_step) {
      if (e.partial != nullptr) c->partials_n = (int)((B + 31) / 32);      // two lanes per evaluation: one partial per 32 items
      HIP_TRY(qmps::launch_energy_pair_d4(e, c->stream));
    } else {
      HIP_TRY(qmps::launch_energy(c->D, e, false, c->stream));
    }
  }
  if (c->timed) c->samples++;
  if (!c->capturing) c->launches++;
  c->have_env = true;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_kernel_timing_period(qmps_ctx* c, int period) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (period < 0) return fail(QMPS_ERR_ARG, "period must be >= 0");
  c->timing_period = period;
  c->samples = 0;          // earlier samples belong to another schedule
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_handoff(qmps_ctx* c, int handoff) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (handoff < 0) return fail(QMPS_ERR_ARG, "handoff must be >= 0");
  c->handoff = handoff;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_default_solver(qmps_ctx* c, int solver) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (solver != QMPS_ENV_POWER && solver != QMPS_ENV_POWER_SQUARING && solver != QMPS_ENV_DIRECT)
    return fail(QMPS_ERR_ARG, "unknown solver %d", solver);
  c->default_solver = solver;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_roto_rule(qmps_ctx* c, int rule) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (rule != QMPS_ROTO_REFERENCE && rule != QMPS_ROTO_GLOBAL_ARGMIN) return fail(QMPS_ERR_ARG, "unknown rotosolve rule %d", rule);
  c->roto_rule = rule;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_roto_rule_probe(qmps_ctx* c, int64_t n, const double* abcd, int rule, double* theta) try {
  if (int rc = bind(c)) return rc;
  if (!abcd || !theta || n < 1 || n > (1 << 24)) return fail(QMPS_ERR_ARG, "bad arguments");
  if (rule != QMPS_ROTO_REFERENCE && rule != QMPS_ROTO_GLOBAL_ARGMIN) return fail(QMPS_ERR_ARG, "unknown rotosolve rule %d", rule);
  if (int rc = ensure_scratch(c, (size_t)n * 5 * sizeof(double))) return rc;
  double* d_in = (double*)c->d_scratch;
  double* d_out = d_in + 4 * n;
  HIP_TRY(hipMemcpyAsync(d_in, abcd, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_roto_rule_probe(d_in, n, rule, d_out, c->stream));
  HIP_TRY(hipMemcpyAsync(theta, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_roto_rule(qmps_ctx* c, int* rule) try {
  if (!c || !rule) return fail(QMPS_ERR_ARG, "null argument");
  *rule = c->roto_rule;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_handoff(qmps_ctx* c, int* handoff) try {
  if (!c || !handoff) return fail(QMPS_ERR_ARG, "null argument");
  *handoff = c->handoff;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_squaring_schedule(qmps_ctx* c, int* skip_rounds, int* matvec_period) try {
  if (!c || !skip_rounds || !matvec_period) return fail(QMPS_ERR_ARG, "null argument");
  *skip_rounds = c->skip_rounds;
  *matvec_period = c->D == 4 ? c->matvec_period : 0;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_energy_only_launch(qmps_ctx* c, int64_t B) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment: run qmps_energy_launch or qmps_set_env_guess first");
  if (int rc = ensure_tensors(c)) return rc;
  qmps::LaneArgs a = make_args(c, B, 1, 1.0, false);
  c->partials_B = -1;
  if (c->D == 16 && !documented_switch("QMPS_D16_BLOCK"))
    HIP_TRY(qmps::launch_energy_mfma(c->D, a, false, c->stream));
  else if (c->D == 4 && tuning_knob("QMPS_ENERGY_PAIR") == nullptr)
    HIP_TRY(qmps::launch_energy_only_d4(a, c->stream));     // quad layout, 4+ waves per SIMD (round 1: two lanes per evaluation)
  else if (c->D == 4 && !c->no_pair)
    HIP_TRY(qmps::launch_energy_pair_d4(a, c->stream));
  else
    HIP_TRY(qmps::launch_energy(c->D, a, false, c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

static int sum_on_device(qmps_ctx* c, int64_t B) {
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  c->partials_B = -1;   // d_partial is about to be overwritten by the generic two-pass reduction
  HIP_TRY(qmps::launch_sum(win_E(c), B, c->n_terms, c->d_partial, kSumBlocks, c->d_cost, c->stream));
  return QMPS_OK;
}

int qmps_sum_energies(qmps_ctx* c, int64_t B, double* cost) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!cost) return fail(QMPS_ERR_ARG, "null cost");
  if (int rc = sum_on_device(c, B)) return rc;
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, c->n_terms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(cost, c->h_cost, c->n_terms * sizeof(double));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_energies(qmps_ctx* c, int64_t B, double* E, int32_t* iters, int32_t* status) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  if (E) HIP_TRY(hipMemcpyAsync(E, win_E(c), (size_t)B * c->n_terms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, win_iters(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (status) HIP_TRY(hipMemcpyAsync(status, win_status(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_status(qmps_ctx* c, int64_t B, int32_t* status) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!status) return fail(QMPS_ERR_ARG, "null status");
  HIP_TRY(hipMemcpyAsync(status, win_status(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_env(qmps_ctx* c, int64_t B, double* r) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!r) return fail(QMPS_ERR_ARG, "null r");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment");
  HIP_TRY(hipMemcpyAsync(r, win_r(c), (size_t)B * env_bytes(c), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_rdm(qmps_ctx* c, int64_t B, double* rho) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!rho) return fail(QMPS_ERR_ARG, "null rho");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment");
  if (!c->d_rho) HIP_TRY(hipMalloc(&c->d_rho, (size_t)c->max_batch * 256));
  // recompute from the resident (A, r): the energy-only kernel writes rho when asked to
  c->want_rho = true;
  int rc = qmps_energy_only_launch(c, B);
  c->want_rho = false;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(rho, (char*)c->d_rho + (size_t)c->window * 256, (size_t)B * 256, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_energy_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* h, int n_terms,
                      const double* r0, int max_iter, double tol, double* E_out, int32_t* iters_out,
                      int32_t* status_out) try {
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  if (int rc = qmps_set_env_guess(c, B, r0)) return rc;
  if (int rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver | QMPS_FLAG_KRYLOV_FALLBACK)) return rc;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}
QMPS_API_CATCH

int qmps_energy_batch_ansatz(qmps_ctx* c, int64_t B, int ansatz_kind, int n_params, const double* params, const double* h,
                             int n_terms, int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  // the host buffers stay the caller's until this function returns: the copies in may stay in flight until the ONE
  // synchronisation of the read-back (a scalar objective call is all latency: three round trips -> one)
  int rc;
  {
    Restore<bool> deferred(c->defer_sync, true);
    rc = qmps_set_states_ansatz(c, B, ansatz_kind, n_params, params);
    if (!rc) rc = qmps_set_hamiltonian(c, n_terms, h);
    if (!rc) rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver | QMPS_FLAG_KRYLOV_FALLBACK | ((c->D == 4 && c->default_solver == QMPS_ENV_DIRECT) ? QMPS_FLAG_NO_ENV_OUT : 0));
  }
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}
QMPS_API_CATCH

int qmps_env_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* r0, int max_iter, double tol,
                   double* r_out, int32_t* iters_out, int32_t* status_out) try {
  if (!r_out) return fail(QMPS_ERR_ARG, "null r_out");
  if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  if (c->n_terms < 1) {
    // the solve kernel always evaluates at least one Hamiltonian term; use h = 0
    double zero[32];
    memset(zero, 0, sizeof(zero));
    if (int rc = qmps_set_hamiltonian(c, 1, zero)) return rc;
  }
  if (int rc = qmps_set_env_guess(c, B, r0)) return rc;
  if (int rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver | QMPS_FLAG_KRYLOV_FALLBACK)) return rc;
  if (int rc = qmps_get_energies(c, B, nullptr, iters_out, status_out)) return rc;
  return qmps_get_env(c, B, r_out);
}
QMPS_API_CATCH

int qmps_cell2_energy_batch(qmps_ctx* c, int64_t B, const double* U1, const double* U2, const double* h, int n_terms,
                             int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (c->D != 2) return fail(QMPS_ERR_ARG, "the two-site unit cell path is D = 2 only (qmps/ground_state.py:276)");
  c->window = 0;      // a one-shot call: results at the start of the buffers, like the qmps_set_* calls
  if ((!U1 || !U2) && B > 0) return fail(QMPS_ERR_ARG, "null unitaries");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  const size_t ub = 2 * tensor_bytes(c);
  if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * ub));
  if (!c->d_U2) HIP_TRY(hipMalloc(&c->d_U2, (size_t)c->max_batch * ub));
  HIP_TRY(hipMemcpyAsync(c->d_U, U1, (size_t)B * ub, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_U2, U2, (size_t)B * ub, hipMemcpyHostToDevice, c->stream));
  qmps::Cell2Args a;
  a.U1 = c->d_U; a.U2 = c->d_U2; a.h = c->d_h; a.E = c->d_E; a.E12 = nullptr;
  a.iters = c->d_iters; a.status = c->d_status; a.B = B; a.n_terms = n_terms; a.max_iter = max_iter; a.tol = tol;
  c->partials_B = -1;
  HIP_TRY(qmps::launch_cell2(c->D, a, c->stream));
  c->n_states = 0;  // the resident single-site states (if any) are no longer what d_E refers to
  c->have_env = false;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}
QMPS_API_CATCH

int qmps_cell2_energy_batch_su(qmps_ctx* c, int64_t B, const double* params, const double* h, int n_terms, int max_iter, double tol,
                               double* E_out, int32_t* iters_out, int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (c->D != 2) return fail(QMPS_ERR_ARG, "the two-site unit cell path is D = 2 only (qmps/ground_state.py:276)");
  c->window = 0;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  const size_t ub = 2 * tensor_bytes(c);
  if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * ub));
  if (!c->d_U2) HIP_TRY(hipMalloc(&c->d_U2, (size_t)c->max_batch * ub));
  if (int rc = ensure_scratch(c, (size_t)B * 30 * sizeof(double) + 256)) return rc;
  double* d_p = (double*)c->d_scratch;
  HIP_TRY(hipMemcpyAsync(d_p, params, (size_t)B * 30 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // U1 = U4(p[:15]), U2 = U4(p[15:])  (qmps/ground_state.py:300-301), both built on the device
  HIP_TRY(qmps::launch_su_exp(4, d_p, B, 30, c->d_U, 0, c->stream));
  HIP_TRY(qmps::launch_su_exp(4, d_p + 15, B, 30, c->d_U2, 0, c->stream));
  qmps::Cell2Args a;
  a.U1 = c->d_U; a.U2 = c->d_U2; a.h = c->d_h; a.E = c->d_E; a.E12 = nullptr;
  a.iters = c->d_iters; a.status = c->d_status; a.B = B; a.n_terms = n_terms; a.max_iter = max_iter; a.tol = tol;
  c->partials_B = -1;
  HIP_TRY(qmps::launch_cell2(c->D, a, c->stream));
  c->n_states = 0;
  c->have_env = false;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}
QMPS_API_CATCH

int qmps_kernel_time(qmps_ctx* c, int n_last, float* avg_ms, char* name, int name_len) try {
  if (int rc = bind(c)) return rc;
  if (!avg_ms || n_last < 1) return fail(QMPS_ERR_ARG, "bad arguments");
  if (c->samples < 1) return fail(QMPS_ERR_STATE, "no timed energy launch yet (qmps_set_kernel_timing_period)");
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the timed launches among the last n_last ones
  int64_t n = c->timing_period > 0 ? (n_last + c->timing_period - 1) / c->timing_period : 1;
  if (n < 1) n = 1;
  if (n > c->samples) n = c->samples;
  if (n > qmps_ctx::kRing) n = qmps_ctx::kRing;
  double sum = 0.0;
  for (int64_t k = 0; k < n; ++k) {
    const int slot = (int)((c->samples - 1 - k) % qmps_ctx::kRing);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->kev0[slot], c->kev1[slot]));
    sum += ms;
  }
  *avg_ms = (float)(sum / (double)n);
  if (name && name_len > 0) snprintf(name, name_len, "%s", c->dominant);
  return QMPS_OK;
}
QMPS_API_CATCH

namespace {
// bump allocator over the scratch arena: copies a host array in, returns the device address
struct Arena {
  qmps_ctx* c;
  size_t off = 0;
  void* put(const void* host, size_t bytes, hipError_t* err) {
    void* d = (char*)c->d_scratch + off;
    off += (bytes + 255) & ~(size_t)255;
    if (host) *err = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream);
    return d;
  }
};
}  // namespace

int qmps_bw_expval(qmps_ctx* c, int64_t B, int sites, const double* U1, const double* U2, const double* O, int o_shared,
                   double* out) try {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !O || !out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (sites != 2 && sites != 4) return fail(QMPS_ERR_ARG, "sites must be 2 or 4");
  const size_t no = sites == 2 ? 16 : 256;
  const size_t ob = (o_shared ? 1 : (size_t)B) * no * 16;
  if (int rc = ensure_scratch(c, (size_t)B * (256 + 256 + 16 + 256) + ob + 4096)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.O = a.put(O, ob, &e); HIP_TRY(e);
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.B = B; k.o_shared = o_shared ? 1 : 0;
  HIP_TRY(qmps::launch_bw(sites == 2 ? 0 : 1, k, c->stream));
  HIP_TRY(hipMemcpyAsync(out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_bw_env(qmps_ctx* c, int64_t B, int side, const double* U1, const double* U2, const double* U1p,
                const double* U2p, int max_rounds, double tol, double* mat_out, double* eta_out, double* vec_out,
                int32_t* status_out) try {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !U1p || !U2p || !eta_out || !vec_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (side != 0 && side != 1) return fail(QMPS_ERR_ARG, "side must be 0 (right) or 1 (left)");
  if (max_rounds < 1 || max_rounds > 60 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol");
  if (int rc = ensure_scratch(c, (size_t)B * (4 * 256 + 256 + 16 + 64 + 16) + 8192)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.U1p = a.put(U1p, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2p = a.put(U2p, (size_t)B * 256, &e); HIP_TRY(e);
  k.mat_out = mat_out ? a.put(nullptr, (size_t)B * 256, &e) : nullptr;
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.vec_out = a.put(nullptr, (size_t)B * 64, &e);
  k.status = (int32_t*)a.put(nullptr, (size_t)B * 4, &e);
  k.B = B; k.side = side; k.max_rounds = max_rounds; k.tol = tol;
  HIP_TRY(qmps::launch_bw(2, k, c->stream));
  if (mat_out) HIP_TRY(hipMemcpyAsync(mat_out, k.mat_out, (size_t)B * 256, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(eta_out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(vec_out, k.vec_out, (size_t)B * 64, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, k.status, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_bw_manifold(qmps_ctx* c, int64_t B, const double* U1, const double* U2, const double* U1p, const double* U2p,
                     const double* Mr, const double* Ml, int m_shared, const double* W, int w_shared, double* out) try {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !U1p || !U2p || !Mr || !Ml || !W || !out) return fail(QMPS_ERR_ARG, "bad arguments");
  const size_t mb = (m_shared ? 1 : (size_t)B) * 64, wb = (w_shared ? 1 : (size_t)B) * 4096;
  if (int rc = ensure_scratch(c, (size_t)B * (4 * 256 + 16) + 2 * mb + wb + 8192)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.U1p = a.put(U1p, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2p = a.put(U2p, (size_t)B * 256, &e); HIP_TRY(e);
  k.Mr = a.put(Mr, mb, &e); HIP_TRY(e);
  k.Ml = a.put(Ml, mb, &e); HIP_TRY(e);
  k.O = a.put(W, wb, &e); HIP_TRY(e);
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.B = B; k.m_shared = m_shared ? 1 : 0; k.o_shared = w_shared ? 1 : 0;
  HIP_TRY(qmps::launch_bw(3, k, c->stream));
  HIP_TRY(hipMemcpyAsync(out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_opt_env_objective(qmps_ctx* c, int64_t B, const double* params, const double* h, double k, double* f_out,
                           double* parts_out) try {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !params || !h || !f_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (int rc = ensure_scratch(c, (size_t)B * (240 + 8 + 32) + 4096)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  const double* d_p = (const double*)a.put(params, (size_t)B * 240, &e); HIP_TRY(e);
  const void* d_h = a.put(h, 256, &e); HIP_TRY(e);
  double* d_f = (double*)a.put(nullptr, (size_t)B * 8, &e);
  double* d_parts = parts_out ? (double*)a.put(nullptr, (size_t)B * 32, &e) : nullptr;
  HIP_TRY(qmps::launch_opt_env(d_p, d_h, k, d_f, d_parts, B, c->stream));
  HIP_TRY(hipMemcpyAsync(f_out, d_f, (size_t)B * 8, hipMemcpyDeviceToHost, c->stream));
  if (parts_out) HIP_TRY(hipMemcpyAsync(parts_out, d_parts, (size_t)B * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_timer_begin(qmps_ctx* c) try {
  if (int rc = bind(c)) return rc;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_timer_end(qmps_ctx* c, float* ms) try {
  if (int rc = bind(c)) return rc;
  if (!ms) return fail(QMPS_ERR_ARG, "null ms");
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return QMPS_OK;
}
QMPS_API_CATCH

// ---- RCCL ---------------------------------------------------------------------------------
int qmps_comm_unique_id(char id[QMPS_UNIQUE_ID_BYTES]) try {
  if (!id) return fail(QMPS_ERR_ARG, "null id");
  static_assert(sizeof(ncclUniqueId) <= QMPS_UNIQUE_ID_BYTES, "ncclUniqueId larger than QMPS_UNIQUE_ID_BYTES");
  ncclUniqueId u;
  RCCL_TRY(ncclGetUniqueId(&u));
  memset(id, 0, QMPS_UNIQUE_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_comm_init(qmps_ctx* c, const char id[QMPS_UNIQUE_ID_BYTES], int rank, int nranks) try {
  if (int rc = bind(c)) return rc;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(QMPS_ERR_ARG, "bad communicator arguments");
  if (c->comm) return fail(QMPS_ERR_STATE, "communicator already initialised");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  RCCL_TRY(ncclCommInitRank(&c->comm, nranks, u, rank));
  if (!tuning_knob("QMPS_ONE_COMM")) {
    // second communicator over the same ranks (collective, like the init itself); without it everything runs on the first
    ncclResult_t r2 = ncclCommSplit(c->comm, 0, rank, &c->comm2, nullptr);
    if (r2 != ncclSuccess) c->comm2 = nullptr;
    // every rank must take the same decision (slot -> communicator): agree on min over ranks of "I have the second one"
    HIP_TRY(hipStreamSynchronize(c->stream));
    double* flag = c->d_cost;
    const double mine = c->comm2 ? 1.0 : 0.0;
    double all = 0.0;
    HIP_TRY(hipMemcpyAsync(flag, &mine, sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    RCCL_TRY(ncclAllReduce(flag, flag, 1, ncclDouble, ncclMin, c->comm, c->comm_stream));
    HIP_TRY(hipMemcpyAsync(&all, flag, sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (all < 0.5 && c->comm2) {
      (void)ncclCommDestroy(c->comm2);
      c->comm2 = nullptr;
    }
  }
  c->rank = rank;
  c->nranks = nranks;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_comm_destroy(qmps_ctx* c) try {
  if (int rc = bind(c)) return rc;
  if (c->comm) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream2));
    if (c->comm2) RCCL_TRY(ncclCommDestroy(c->comm2));
    c->comm2 = nullptr;
    RCCL_TRY(ncclCommDestroy(c->comm));
    c->comm = nullptr;
    c->nranks = 1;
    c->rank = 0;
  }
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_comm_count(qmps_ctx* c, int* nranks) try {
  if (int rc = bind(c)) return rc;
  if (!nranks) return fail(QMPS_ERR_ARG, "null nranks");
  *nranks = 1;
  if (c->comm) RCCL_TRY(ncclCommCount(c->comm, nranks));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_allreduce_sum(qmps_ctx* c, double* inout, int n) try {
  if (int rc = bind(c)) return rc;
  if (!inout || n < 1 || n > kMaxTerms) return fail(QMPS_ERR_ARG, "n=%d outside [1,%d]", n, kMaxTerms);
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  memcpy(c->h_cost, inout, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_cost, c->h_cost, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(ncclAllReduce(c->d_cost, c->d_cost, n, ncclDouble, ncclSum, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(inout, c->h_cost, n * sizeof(double));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_allreduce_min(qmps_ctx* c, double* inout, int n) try {
  if (int rc = bind(c)) return rc;
  if (!inout || n < 1 || n > kMaxTerms) return fail(QMPS_ERR_ARG, "n=%d outside [1,%d]", n, kMaxTerms);
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  memcpy(c->h_cost, inout, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_cost, c->h_cost, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(ncclAllReduce(c->d_cost, c->d_cost, n, ncclDouble, ncclMin, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(inout, c->h_cost, n * sizeof(double));
  return QMPS_OK;
}
QMPS_API_CATCH

}  // extern "C"
namespace qmps_host {
// close the current group: ONE ncclAllReduce of its `fill` x 16 doubles on the communication stream, ordered after the
// device-side sums by an event, so the exchange overlaps the next steps' kernels instead of stalling the compute stream
int close_group(qmps_ctx* c) {
  if (c->group_fill == 0) return QMPS_OK;
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots);
  double* base = c->d_cost_ring + (size_t)slot * qmps_ctx::kMaxGroup * kMaxTerms;
  if (c->comm) {
#ifdef QMPS_DEBUG_KNOBS   // timing dissections only (they produce WRONG costs): compiled in with -DQMPS_DEBUG_KNOBS, never in the shipped library
    static const bool dbg_noevent = getenv("QMPS_DBG_NOEVENT") != nullptr, dbg_noar = getenv("QMPS_DBG_NOAR") != nullptr,
                      dbg_nofinish = getenv("QMPS_DBG_NOFINISH") != nullptr, dbg_nopoll = getenv("QMPS_DBG_NOPOLL") != nullptr;
#else
    constexpr bool dbg_noevent = false, dbg_noar = false, dbg_nofinish = false, dbg_nopoll = false;
#endif
    // positions whose cost lives in a fixed-point accumulator need no ordering on the compute stream: their finish
    // kernel polls the arrival counts.  Only costs written by reduction kernels on the compute stream need the event.
    bool need_event = false;
    for (int pos = 0; pos < c->group_fill; ++pos) need_event = need_event || !c->acc_is[slot][pos] || c->acc_after_event[slot][pos];
    if (need_event && !dbg_noevent) {
      HIP_TRY(hipEventRecord(c->cost_ready[slot], c->stream));
      HIP_TRY(hipStreamWaitEvent(c->comm_stream_of(slot), c->cost_ready[slot], 0));
    }
    for (int pos = 0; pos < c->group_fill; ++pos)
      if (c->acc_is[slot][pos]) {   // fixed-point accumulators -> doubles, off the compute stream
        if (!dbg_nofinish)
          HIP_TRY(qmps::launch_cost_finish(c->acc_at(slot, pos), c->acc_shards[slot][pos], c->acc_expect[slot][pos], dbg_nopoll ? 0 : 1 << 22,
                                           1.0 / c->acc_scale[slot][pos], c->n_terms, base + (size_t)pos * kMaxTerms,
                                           c->d_acc_err, c->comm_stream_of(slot)));
        c->acc_is[slot][pos] = false;
      }
#ifdef QMPS_DEBUG_KNOBS
    // robustness drill for the exchange pipeline at world size 1, where the real all-reduce is instantaneous: a busy kernel in
    // front of it makes every exchange last QMPS_DBG_SLOW_AR probe iterations (~1300 = 40 us, longer than a step), so the ring
    // fills up, the host-side slot guard blocks and the finish kernels queue behind exchanges that are still in flight
    static const int slow_ar = getenv("QMPS_DBG_SLOW_AR") ? atoi(getenv("QMPS_DBG_SLOW_AR")) : 0;
    if (slow_ar > 0) HIP_TRY(qmps::launch_probe_fp64((double*)c->d_work_idx, 1, slow_ar, c->comm_stream_of(slot)));
#endif
    if (!dbg_noar)
      RCCL_TRY(ncclAllReduce(base, base, (size_t)c->group_fill * kMaxTerms, ncclDouble, ncclSum, c->comm_of(slot), c->comm_stream_of(slot)));
    HIP_TRY(hipEventRecord(c->cost_reduced[slot], c->comm_stream_of(slot)));
  }
  c->group_fill = 0;
  c->groups++;
  c->slot_waited = false;
  return QMPS_OK;
}
}  // namespace qmps_host
extern "C" {

int qmps_exchange_stats(qmps_ctx* c, int64_t* checks, int64_t* blocked, double* blocked_ms, int reset) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (checks) *checks = c->slot_checks;
  if (blocked) *blocked = c->slot_blocks;
  if (blocked_ms) *blocked_ms = c->slot_block_ms;
  if (reset) { c->slot_checks = 0; c->slot_blocks = 0; c->slot_block_ms = 0.0; }
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_set_exchange_period(qmps_ctx* c, int steps) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (steps < 1 || steps > qmps_ctx::kMaxGroup) return fail(QMPS_ERR_ARG, "exchange period must be in [1, %d]", qmps_ctx::kMaxGroup);
  if (int rc = bind(c)) return rc;
  if (int rc = close_group(c)) return rc;     // costs summed under the old period are exchanged now
  c->exchange_period = steps;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_cost_launch(qmps_ctx* c, int64_t B) try {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  // device-side sum into this step's place in the current group of the ring (main stream) ...
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots);
  double* dst = c->d_cost_ring + ((size_t)slot * qmps_ctx::kMaxGroup + c->group_fill) * kMaxTerms;
  c->acc_is[slot][c->group_fill] = false;
  const bool in_kernel = c->acc_pending && c->acc_B == B && c->acc_window == c->window && c->acc_slot == slot && c->acc_pos == c->group_fill;
  // A slot is reused only after its previous all-reduce has finished.  Costs written by a reduction kernel on the compute
  // stream need that as a stream dependency; a cost that lives in a fixed-point accumulator is converted on the slot's own
  // communication stream, behind that all-reduce, and puts nothing on the compute stream (no barrier packet per step).
  if (c->comm && !in_kernel && c->groups >= qmps_ctx::kCostSlots && !c->slot_waited) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->cost_reduced[slot], 0));
    c->slot_waited = true;
  }
  if (in_kernel) {
    // the energy kernel has summed the batch itself (exact fixed-point accumulator): nothing to launch
    c->acc_is[slot][c->group_fill] = true;
  } else if (c->partials_B == B)   // the energy kernel already left per-wave partial sums: only the final pass is needed
    HIP_TRY(qmps::launch_sum_final(c->d_partial, c->partials_n, c->n_terms, dst, c->stream));
  else {
    c->partials_B = -1;   // the generic two-pass reduction reuses d_partial
    HIP_TRY(qmps::launch_sum(win_E(c), B, c->n_terms, c->d_partial, kSumBlocks, dst, c->stream));
  }
  c->acc_pending = false;
  c->last_slot = slot;
  c->last_pos = c->group_fill;
  c->group_fill++;
  c->cost_launches++;
  // ... then, once per `exchange_period` steps, the exchange step
  if (c->group_fill >= c->exchange_period)
    if (int rc = close_group(c)) return rc;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_get_cost(qmps_ctx* c, double* cost) try {
  if (int rc = bind(c)) return rc;
  if (!cost) return fail(QMPS_ERR_ARG, "null cost");
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  if (c->cost_launches < 1) return fail(QMPS_ERR_STATE, "qmps_cost_launch has not been called");
  if (int rc = close_group(c)) return rc;     // a partly filled group is exchanged now
  hipStream_t st = c->comm ? c->comm_stream_of(c->last_slot) : c->stream;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->acc_is[c->last_slot][c->last_pos]) {
    // no communicator: the cost still lives in its fixed-point accumulator; sum the shards on the host (exact)
    HIP_TRY(hipMemcpyAsync(c->h_acc, c->acc_at(c->last_slot, c->last_pos), qmps::kAccWords * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double inv = 1.0 / c->acc_scale[c->last_slot][c->last_pos];
    for (int t = 0; t < c->n_terms; ++t) {
      long long cnt = 0, hi = 0, lo = 0;
      for (int sh = 0; sh < c->acc_shards[c->last_slot][c->last_pos]; ++sh) {
        long long k, v;
        qmps::acc_decode(c->h_acc[t * qmps::kAccMaxShards + sh], k, v);
        cnt += k;
        hi += v >> 20;
        lo += v & 0xFFFFF;
      }
      if (cnt != c->acc_expect[c->last_slot][c->last_pos])
        return fail(QMPS_ERR_STATE, "cost accumulator: %lld of %lld waves arrived", cnt, c->acc_expect[c->last_slot][c->last_pos]);
      cost[t] = ((double)hi * 1048576.0 + (double)lo) * inv + ((const double*)(c->h_acc + qmps::kAccOver))[t];
    }
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost_ring + ((size_t)c->last_slot * qmps_ctx::kMaxGroup + c->last_pos) * kMaxTerms,
                         c->n_terms * sizeof(double), hipMemcpyDeviceToHost, st));
  int acc_err = 0;
  if (c->comm) HIP_TRY(hipMemcpyAsync(&acc_err, c->d_acc_err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (acc_err) {
    // a finish kernel gave up waiting for its energy kernel's waves (bounded poll): the cost it wrote is NaN
    (void)hipMemsetAsync(c->d_acc_err, 0, sizeof(int), st);
    return fail(QMPS_ERR_STATE, "cost accumulator: a step's energy kernel did not arrive within the polling bound (was it launched?)");
  }
  memcpy(cost, c->h_cost, c->n_terms * sizeof(double));
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_allreduce_cost(qmps_ctx* c, int64_t B, double* cost) try {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  if (int rc = qmps_cost_launch(c, B)) return rc;
  return qmps_get_cost(c, cost);
}
QMPS_API_CATCH

// ---- probes -------------------------------------------------------------------------------
int qmps_probe_fp64_peak(qmps_ctx* c, double* tflops) try {
  if (int rc = bind(c)) return rc;
  if (!tflops) return fail(QMPS_ERR_ARG, "null tflops");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int blocks = prop.multiProcessorCount * 8;  // 2 waves per SIMD
  const int iters = 20000;
  HIP_TRY(qmps::launch_probe_fp64(c->d_cost, blocks, 200, c->stream));  // warm-up
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(qmps::launch_probe_fp64(c->d_cost, blocks, iters, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * 16.0 * iters * 256.0 * blocks;
  *tflops = flops / (best * 1e-3) * 1e-12;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_probe_fp64_mfma_peak(qmps_ctx* c, int waves_per_simd, double* tflops) try {
  if (int rc = bind(c)) return rc;
  if (!tflops || waves_per_simd < 1 || waves_per_simd > 8) return fail(QMPS_ERR_ARG, "bad arguments");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  const int iters = 20000;
  HIP_TRY(qmps::launch_probe_mfma_f64(c->d_cost, blocks, 200, c->stream));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(qmps::launch_probe_mfma_f64(c->d_cost, blocks, iters, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  const double flops = 4.0 * 2048.0 * iters * 4.0 * blocks;  // 4 MFMAs x 2048 flop, 4 waves per block
  *tflops = flops / (best * 1e-3) * 1e-12;
  return QMPS_OK;
}
QMPS_API_CATCH

int qmps_probe_hbm_peak(qmps_ctx* c, double* gbps) try {
  if (int rc = bind(c)) return rc;
  if (!gbps) return fail(QMPS_ERR_ARG, "null gbps");
  const size_t bytes = (size_t)1 << 30;  // 1 GiB each way: well past the 256 MiB Infinity Cache
  void *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc(&src, bytes));
  if (hipMalloc(&dst, bytes) != hipSuccess) {
    (void)hipFree(src);
    return fail(QMPS_ERR_HIP, "hipMalloc failed in the HBM probe");
  }
  int rc = [&]() -> int {
    HIP_TRY(hipMemsetAsync(src, 1, bytes, c->stream));
    HIP_TRY(qmps::launch_probe_copy(src, dst, (int64_t)(bytes / 16), c->stream));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      HIP_TRY(hipEventRecord(c->ev0, c->stream));
      HIP_TRY(qmps::launch_probe_copy(src, dst, (int64_t)(bytes / 16), c->stream));
      HIP_TRY(hipEventRecord(c->ev1, c->stream));
      HIP_TRY(hipEventSynchronize(c->ev1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
      if (ms < best) best = ms;
    }
    *gbps = 2.0 * (double)bytes / (best * 1e-3) * 1e-9;
    return QMPS_OK;
  }();
  (void)hipFree(src);
  (void)hipFree(dst);
  return rc;
}
QMPS_API_CATCH

}  // extern "C"
