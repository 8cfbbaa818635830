// cpuabi.cpp - TEST INFRASTRUCTURE ONLY (last bullet: "a CPU build of the same ABI must exist so the boundary is
// testable without a GPU").  Exports a SUBSET of include/qmps_hip.h - qmps_abi_version, qmps_last_error, qmps_create,
// qmps_destroy, qmps_energy_batch, qmps_env_batch - with the header's own prototypes, argument checks and error codes, at bond
// dimension D = 4 with tensor input, computing through the very mathematics the GPU kernel is built from
// (qmps_amd/csrc/qmps_direct_core.h, instantiated on the host by direct_emu.cpp).  tests/test_cabi.py drives the boundary end to end
// through it in the GPU-less container: marshalling, ownership, return codes, qmps_last_error().
// It is NEVER loaded by qmps_amd (the product has no CPU fallback: qmps_create of libqmps_hip.so fails without a gfx950 device) and is
// not linked against oracle/.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "qmps_hip.h"

extern "C" int direct_emu_d4(long B, const double* A, const double* h, int nt, int max_iter, double tol, double* E, double* r_out,
                             double* rho_out, int32_t* iters, int32_t* status, double* resid, double* E_lean);

struct qmps_ctx {
  int D;
  int64_t max_batch;
};

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check(const qmps_ctx* c, int64_t B, const double* states, int kind) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (B < 1 || B > c->max_batch) return fail(QMPS_ERR_ARG, "B=%lld outside [1, max_batch=%lld]", (long long)B, (long long)c->max_batch);
  if (!states) return fail(QMPS_ERR_ARG, "null states");
  if (kind != QMPS_INPUT_TENSOR) return fail(QMPS_ERR_ARG, "the CPU test build takes tensors only (QMPS_INPUT_TENSOR)");
  return QMPS_OK;
}
}  // namespace

extern "C" {

int qmps_abi_version(void) { return QMPS_ABI_VERSION; }
const char* qmps_last_error(void) { return g_err; }

int qmps_create(int device, int D, int64_t max_batch, qmps_ctx** out) try {
  if (!out) return fail(QMPS_ERR_ARG, "null out");
  *out = nullptr;
  if (device != 0) return fail(QMPS_ERR_NO_DEVICE, "the CPU test build has one pseudo-device (0)");
  if (D != 4) return fail(QMPS_ERR_ARG, "the CPU test build of the ABI covers D = 4 only (got %d)", D);
  if (max_batch < 1) return fail(QMPS_ERR_ARG, "max_batch must be >= 1");
  *out = new qmps_ctx{D, max_batch};
  return QMPS_OK;
} catch (const std::bad_alloc&) {
  return fail(QMPS_ERR_ARG, "out of host memory");
}

int qmps_destroy(qmps_ctx* c) {
  delete c;
  return QMPS_OK;
}

int qmps_energy_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* h, int n_terms, const double* r0,
                      int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out) try {
  if (int rc = check(c, B, states, kind)) return rc;
  if (!h || !E_out) return fail(QMPS_ERR_ARG, "null h / E_out");
  if (n_terms < 1 || n_terms > 16) return fail(QMPS_ERR_ARG, "n_terms=%d outside [1,16]", n_terms);
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  (void)r0;      // (a warm start changes the iteration count, never the fixed point)
  std::vector<int32_t> it(B), st(B);
  direct_emu_d4((long)B, states, h, n_terms, max_iter, tol, E_out, nullptr, nullptr, it.data(), st.data(), nullptr, nullptr);
  if (iters_out) memcpy(iters_out, it.data(), (size_t)B * sizeof(int32_t));
  if (status_out) memcpy(status_out, st.data(), (size_t)B * sizeof(int32_t));
  return QMPS_OK;
} catch (const std::bad_alloc&) {
  return fail(QMPS_ERR_ARG, "out of host memory (sizes too large?)");
}

int qmps_env_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* r0, int max_iter, double tol, double* r_out,
                   int32_t* iters_out, int32_t* status_out) try {
  if (int rc = check(c, B, states, kind)) return rc;
  if (!r_out) return fail(QMPS_ERR_ARG, "null r_out");
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  (void)r0;
  const double h0[32] = {0};
  std::vector<double> E(B);
  std::vector<int32_t> it(B), st(B);
  direct_emu_d4((long)B, states, h0, 1, max_iter, tol, E.data(), r_out, nullptr, it.data(), st.data(), nullptr, nullptr);
  if (iters_out) memcpy(iters_out, it.data(), (size_t)B * sizeof(int32_t));
  if (status_out) memcpy(status_out, st.data(), (size_t)B * sizeof(int32_t));
  return QMPS_OK;
} catch (const std::bad_alloc&) {
  return fail(QMPS_ERR_ARG, "out of host memory (sizes too large?)");
}

}  // extern "C"
