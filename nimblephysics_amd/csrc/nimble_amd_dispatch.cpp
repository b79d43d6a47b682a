// nimble_amd_dispatch.cpp — the entry points of include/nimble_amd.h over the instantiations of the library (abi_variants.h):
// a model is given, when it is created, to the 24-row build (max_contacts <= 8, <= 16 colliders, <= 32 collider pairs: every
// BASELINE config), to the 48-row build (up to 16 contacts, 32 colliders, 64 pairs per world) or to the GENERAL build (up to 64
// contacts = 192 LCP rows, 64 colliders, 512 pairs: rows looped over instead of mapped to lanes - there so that no legal world
// gets a truncated answer; 2 M world-steps/s on eight-contact worlds since round 6; a fourth instantiation is the same code with 384 rows); every later call goes to the build that owns the handle.  Host code only; no HIP call of its own.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "../../include/nimble_amd.h"
#define NBL_DISPATCHER
#include "abi_variants.h"

// the same signatures as include/nimble_amd.h with the handle as void* (C linkage: no type in the symbol)
#define NBL_DECLARE_VARIANT(S)                                                                                                              \
  const char* nbl_last_error##S(void);                                                                                                     \
  int32_t nbl_version##S(void);                                                                                                            \
  int32_t nbl_device_count##S(void);                                                                                                       \
  int32_t nbl_model_create##S(const nbl_model_desc*, int32_t, void**);                                                                     \
  void nbl_model_destroy##S(void*);                                                                                                        \
  int32_t nbl_model_num_dofs##S(const void*);                                                                                              \
  int32_t nbl_model_num_action##S(const void*);                                                                                            \
  int32_t nbl_model_lcp_rows##S(const void*);                                                                                              \
  int32_t nbl_model_max_contacts##S(const void*);                                                                                          \
  size_t nbl_workspace_bytes##S(const void*, int64_t);                                                                                     \
  size_t nbl_saved_bytes##S(const void*, int64_t);                                                                                         \
  int32_t nbl_step_forward##S(void*, int64_t, const double*, const double*, const double*, double*, double*, void*, uint32_t*, void*,      \
                              size_t, void*);                                                                                              \
  int32_t nbl_step_backward##S(void*, int64_t, const void*, const double*, double*, double*, void*, size_t, void*);                        \
  int32_t nbl_set_body_inertia##S(void*, int32_t, double, const double*, const double*);                                                   \
  int32_t nbl_set_body_inertias##S(void*, int32_t, const int32_t*, const double*, const double*, const double*, void*);                    \
  int32_t nbl_set_inertia_params##S(void*, int32_t, const int32_t*, const double*);                                                        \
  int32_t nbl_set_inertia_params_on##S(void*, int32_t, const int32_t*, const double*, void*);                                              \
  int32_t nbl_num_inertia_params##S(const void*);                                                                                          \
  int32_t nbl_backward_inertia##S(void*, int64_t, const void*, double*, int32_t, void*, size_t, void*);                                    \
  size_t nbl_rollout_workspace_bytes##S(const void*, int64_t);                                                                             \
  int32_t nbl_rollout_forward##S(void*, int64_t, int32_t, const double*, const double*, int64_t, double*, void*, uint32_t*, int32_t,       \
                                 void*, size_t, void*);                                                                                    \
  int32_t nbl_rollout_backward##S(void*, int64_t, int32_t, const void*, const double*, double*, double*, void*, size_t, void*);            \
  int32_t nbl_rollout_backward_inertia##S(void*, int64_t, int32_t, const void*, const double*, double*, double*, double*, void*, size_t,   \
                                          void*);                                                                                          \
  size_t nbl_rollout_checkpoint_bytes##S(const void*, int64_t, int32_t, int32_t);                                                          \
  int32_t nbl_rollout_forward_checkpointed##S(void*, int64_t, int32_t, int32_t, const double*, const double*, int64_t, double*, void*,     \
                                              void*, uint32_t*, int32_t, void*, size_t, void*);                                            \
  int32_t nbl_rollout_backward_checkpointed##S(void*, int64_t, int32_t, int32_t, double*, const double*, int64_t, void*, const void*,      \
                                               int32_t, const double*, double*, double*, double*, void*, size_t, void*);                   \
  int32_t nbl_selftest_lcp_dantzig_timed##S(int32_t, int32_t, const double*, const double*, const double*, const double*, const int32_t*,  \
                                            double*, int32_t*, int32_t, double*);                                                          \
  int32_t nbl_selftest_pinv_rows##S(int32_t, int32_t, const double*, const int32_t*, int32_t, double*, int32_t*, int32_t, double*);        \
  int32_t nbl_selftest_lcp_cascade##S(int32_t, int32_t, const double*, const double*, const double*, int32_t, const double*, const uint8_t*, double,   \
                                      double*, int32_t*, uint32_t*, double*);                                                              \
  int32_t nbl_transpose_to_soa##S(const double*, double*, int64_t, int32_t, void*);                                                        \
  int32_t nbl_transpose_from_soa##S(const double*, double*, int64_t, int32_t, void*);                                                      \
  int32_t nbl_set_launch_lanes##S(void*, int32_t, int32_t);                                                                                \
  int32_t nbl_set_slices##S(void*, int32_t);                                                                                               \
  int32_t nbl_slices_for##S(const void*, int64_t);                                                                                         \
  int32_t nbl_set_deferred_join##S(void*, int32_t);                                                                                        \
  int32_t nbl_slice_stream##S(void*, int64_t, int32_t, void**, int64_t*, int64_t*);                                                        \
  int32_t nbl_join_slices##S(void*, void*);                                                                                                \
  int32_t nbl_fork_slices##S(void*, void*);                                                                                                \
  int32_t nbl_set_timing##S(void*, int32_t);                                                                                               \
  int32_t nbl_get_timing##S(void*, double*, int64_t*, double*, int64_t*);                                                                  \
  int32_t nbl_kernel_count##S(void);                                                                                                       \
  const char* nbl_kernel_name##S(int32_t);                                                                                                 \
  int32_t nbl_kernel_timing##S(void*, int32_t, double*, int64_t*);

extern "C" {
NBL_DECLARE_VARIANT(_c8)
NBL_DECLARE_VARIANT(_c16)
NBL_DECLARE_VARIANT(_c64)
NBL_DECLARE_VARIANT(_c128)
}

// one table of entry points per instantiation
struct Variant {
  int id;   // 8, 16 or 64: the contact slots per world of the instantiation
  const char* (*last_error)(void);
  int32_t (*model_create)(const nbl_model_desc*, int32_t, void**);
  void (*model_destroy)(void*);
  int32_t (*model_num_dofs)(const void*);
  int32_t (*model_num_action)(const void*);
  int32_t (*model_lcp_rows)(const void*);
  int32_t (*model_max_contacts)(const void*);
  size_t (*workspace_bytes)(const void*, int64_t);
  size_t (*saved_bytes)(const void*, int64_t);
  int32_t (*step_forward)(void*, int64_t, const double*, const double*, const double*, double*, double*, void*, uint32_t*, void*, size_t, void*);
  int32_t (*step_backward)(void*, int64_t, const void*, const double*, double*, double*, void*, size_t, void*);
  int32_t (*set_body_inertia)(void*, int32_t, double, const double*, const double*);
  int32_t (*set_body_inertias)(void*, int32_t, const int32_t*, const double*, const double*, const double*, void*);
  int32_t (*set_inertia_params)(void*, int32_t, const int32_t*, const double*);
  int32_t (*set_inertia_params_on)(void*, int32_t, const int32_t*, const double*, void*);
  int32_t (*num_inertia_params)(const void*);
  int32_t (*backward_inertia)(void*, int64_t, const void*, double*, int32_t, void*, size_t, void*);
  size_t (*rollout_workspace_bytes)(const void*, int64_t);
  int32_t (*rollout_forward)(void*, int64_t, int32_t, const double*, const double*, int64_t, double*, void*, uint32_t*, int32_t, void*, size_t, void*);
  int32_t (*rollout_backward)(void*, int64_t, int32_t, const void*, const double*, double*, double*, void*, size_t, void*);
  int32_t (*rollout_backward_inertia)(void*, int64_t, int32_t, const void*, const double*, double*, double*, double*, void*, size_t, void*);
  size_t (*rollout_checkpoint_bytes)(const void*, int64_t, int32_t, int32_t);
  int32_t (*rollout_forward_checkpointed)(void*, int64_t, int32_t, int32_t, const double*, const double*, int64_t, double*, void*, void*, uint32_t*, int32_t,
                                          void*, size_t, void*);
  int32_t (*rollout_backward_checkpointed)(void*, int64_t, int32_t, int32_t, double*, const double*, int64_t, void*, const void*, int32_t, const double*,
                                           double*, double*, double*, void*, size_t, void*);
  int32_t (*selftest_lcp_dantzig_timed)(int32_t, int32_t, const double*, const double*, const double*, const double*, const int32_t*, double*, int32_t*,
                                        int32_t, double*);
  int32_t (*selftest_pinv_rows)(int32_t, int32_t, const double*, const int32_t*, int32_t, double*, int32_t*, int32_t, double*);
  int32_t (*set_launch_lanes)(void*, int32_t, int32_t);
  int32_t (*set_slices)(void*, int32_t);
  int32_t (*slices_for)(const void*, int64_t);
  int32_t (*set_deferred_join)(void*, int32_t);
  int32_t (*slice_stream)(void*, int64_t, int32_t, void**, int64_t*, int64_t*);
  int32_t (*join_slices)(void*, void*);
  int32_t (*fork_slices)(void*, void*);
  int32_t (*set_timing)(void*, int32_t);
  int32_t (*get_timing)(void*, double*, int64_t*, double*, int64_t*);
  int32_t (*kernel_timing)(void*, int32_t, double*, int64_t*);
};
#define NBL_VARIANT_TABLE(ID, S)                                                                                                              \
  {ID, nbl_last_error##S, nbl_model_create##S, nbl_model_destroy##S, nbl_model_num_dofs##S, nbl_model_num_action##S, nbl_model_lcp_rows##S,    \
   nbl_model_max_contacts##S, nbl_workspace_bytes##S, nbl_saved_bytes##S, nbl_step_forward##S, nbl_step_backward##S, nbl_set_body_inertia##S, \
   nbl_set_body_inertias##S, nbl_set_inertia_params##S, nbl_set_inertia_params_on##S, nbl_num_inertia_params##S, nbl_backward_inertia##S,     \
   nbl_rollout_workspace_bytes##S, nbl_rollout_forward##S, nbl_rollout_backward##S, nbl_rollout_backward_inertia##S,                          \
   nbl_rollout_checkpoint_bytes##S, nbl_rollout_forward_checkpointed##S, nbl_rollout_backward_checkpointed##S,                               \
   nbl_selftest_lcp_dantzig_timed##S, nbl_selftest_pinv_rows##S, nbl_set_launch_lanes##S, nbl_set_slices##S, nbl_slices_for##S,               \
   nbl_set_deferred_join##S, nbl_slice_stream##S, nbl_join_slices##S, nbl_fork_slices##S,                                                                         \
   nbl_set_timing##S, nbl_get_timing##S, nbl_kernel_timing##S}
constexpr int kNumVariants = 4;
static const Variant kVariants[kNumVariants] = {NBL_VARIANT_TABLE(8, _c8), NBL_VARIANT_TABLE(16, _c16), NBL_VARIANT_TABLE(64, _c64),
                                                     NBL_VARIANT_TABLE(128, _c128)};

struct nbl_model {
  const Variant* v;   // the instantiation that owns `impl`
  void* impl;
};

namespace {
thread_local const Variant* g_errVariant = nullptr;   // whose message nbl_last_error returns: nullptr = the dispatcher's own
thread_local std::string g_err;
int ownError(int code, const char* msg) {
  g_err = msg;
  g_errVariant = nullptr;
  return code;
}
int noted(const nbl_model* m, int rc) {   // remember which instantiation holds the text of a failure
  if (rc != NBL_OK) g_errVariant = m->v;
  return rc;
}
}  // namespace

// int-returning entry point on a handle: FN = the table entry, then its arguments after the implementation handle
#define NBL_FWD(m, FN, ...) (!(m) ? ownError(NBL_E_BADARG, "null model") : noted((m), (m)->v->FN((m)->impl, ##__VA_ARGS__)))
#define NBL_GET(m, FN, ...) (!(m) ? 0 : (m)->v->FN((m)->impl, ##__VA_ARGS__))

extern "C" {

const char* nbl_last_error(void) { return g_errVariant ? g_errVariant->last_error() : g_err.c_str(); }
int32_t nbl_version(void) { return nbl_version_c8(); }
int32_t nbl_device_count(void) { return nbl_device_count_c8(); }

int32_t nbl_model_create(const nbl_model_desc* d, int32_t device, nbl_model** out) {
  if (!d || !out) return ownError(NBL_E_BADARG, "null argument");
  *out = nullptr;
  void* impl = nullptr;
  // the smallest instantiation that holds the model: contact slots first (a model without colliders and without enforced joint limits has no
  // contact stage: the 24-row build whatever max_contacts says), then whatever an instantiation answers NBL_E_CAPACITY to (its collider /
  // collider-pair budget: 16 / 32, 32 / 64, 64 / 512, 64 / 512).  The two general instantiations differ in their row budget only (192 / 384 rows:
  // the size of the per-world scratch and record): 64 slots unless the model asks for more
  const bool contactStage = d->n_boxes > 0 || d->dof_limit_enforced != nullptr;
  int first = 0;
  if (contactStage && d->max_contacts > 64) first = 3;
  else if (contactStage && (d->max_contacts > 16 || d->n_boxes > 32)) first = 2;
  else if (contactStage && (d->max_contacts > 8 || d->n_boxes > 16)) first = 1;
  // developer knob: NBL_MIN_VARIANT=0..3 starts the search at a later instantiation (never an earlier one: that would truncate)
  if (const char* e = getenv("NBL_MIN_VARIANT")) { const int f = atoi(e); if (f > first && f < kNumVariants) first = f; }
  int32_t rc = NBL_E_CAPACITY;
  const Variant* v = nullptr;
  for (int k = first; k < kNumVariants && rc == NBL_E_CAPACITY; k++) {
    v = &kVariants[k];
    rc = v->model_create(d, device, &impl);
    g_errVariant = v;
  }
  if (rc == NBL_E_CAPACITY) rc = NBL_E_UNSUPPORTED;
  if (rc != NBL_OK) return rc;
  nbl_model* m = new nbl_model();
  m->v = v;
  m->impl = impl;
  *out = m;
  return NBL_OK;
}

void nbl_model_destroy(nbl_model* m) {
  if (!m) return;
  m->v->model_destroy(m->impl);
  delete m;
}

int32_t nbl_model_num_dofs(const nbl_model* m) { return NBL_GET(m, model_num_dofs); }
int32_t nbl_model_num_action(const nbl_model* m) { return NBL_GET(m, model_num_action); }
int32_t nbl_model_lcp_rows(const nbl_model* m) { return NBL_GET(m, model_lcp_rows); }
int32_t nbl_model_max_contacts(const nbl_model* m) { return NBL_GET(m, model_max_contacts); }
size_t nbl_workspace_bytes(const nbl_model* m, int64_t B) { return NBL_GET(m, workspace_bytes, B); }
size_t nbl_saved_bytes(const nbl_model* m, int64_t B) { return NBL_GET(m, saved_bytes, B); }

int32_t nbl_step_forward(nbl_model* m, int64_t B, const double* state, const double* action, const double* lcp_cache_in, double* next_state,
                         double* lcp_cache_out, void* saved, uint32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, step_forward, B, state, action, lcp_cache_in, next_state, lcp_cache_out, saved, status, workspace, workspace_bytes, stream);
}
int32_t nbl_step_backward(nbl_model* m, int64_t B, const void* saved, const double* grad_next_state, double* grad_state, double* grad_action,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, step_backward, B, saved, grad_next_state, grad_state, grad_action, workspace, workspace_bytes, stream);
}

int32_t nbl_set_body_inertia(nbl_model* m, int32_t body, double mass, const double* com, const double* inertia) {
  return NBL_FWD(m, set_body_inertia, body, mass, com, inertia);
}
int32_t nbl_set_body_inertias(nbl_model* m, int32_t count, const int32_t* bodies, const double* mass, const double* com, const double* inertia,
                              void* stream) {
  return NBL_FWD(m, set_body_inertias, count, bodies, mass, com, inertia, stream);
}
int32_t nbl_set_inertia_params(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG) { return NBL_FWD(m, set_inertia_params, count, bodies, dG); }
int32_t nbl_set_inertia_params_on(nbl_model* m, int32_t count, const int32_t* bodies, const double* dG, void* stream) {
  return NBL_FWD(m, set_inertia_params_on, count, bodies, dG, stream);
}
int32_t nbl_num_inertia_params(const nbl_model* m) { return NBL_GET(m, num_inertia_params); }
int32_t nbl_backward_inertia(nbl_model* m, int64_t B, const void* saved, double* grad_params, int32_t accumulate, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, backward_inertia, B, saved, grad_params, accumulate, workspace, workspace_bytes, stream);
}

size_t nbl_rollout_workspace_bytes(const nbl_model* m, int64_t B) { return NBL_GET(m, rollout_workspace_bytes, B); }
int32_t nbl_rollout_forward(nbl_model* m, int64_t B, int32_t T, const double* state0, const double* actions, int64_t action_stride, double* states,
                            void* saved, uint32_t* status, int32_t warm_start, void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, rollout_forward, B, T, state0, actions, action_stride, states, saved, status, warm_start, workspace, workspace_bytes, stream);
}
int32_t nbl_rollout_backward(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states, double* grad_state0,
                             double* grad_actions, void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, rollout_backward, B, T, saved, grad_states, grad_state0, grad_actions, workspace, workspace_bytes, stream);
}
int32_t nbl_rollout_backward_inertia(nbl_model* m, int64_t B, int32_t T, const void* saved, const double* grad_states, double* grad_state0,
                                     double* grad_actions, double* grad_params, void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, rollout_backward_inertia, B, T, saved, grad_states, grad_state0, grad_actions, grad_params, workspace, workspace_bytes, stream);
}
size_t nbl_rollout_checkpoint_bytes(const nbl_model* m, int64_t B, int32_t T, int32_t segment) { return NBL_GET(m, rollout_checkpoint_bytes, B, T, segment); }
int32_t nbl_rollout_forward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, const double* state0, const double* actions,
                                         int64_t action_stride, double* states, void* saved, void* checkpoints, uint32_t* status,
                                         int32_t warm_start, void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, rollout_forward_checkpointed, B, T, segment, state0, actions, action_stride, states, saved, checkpoints, status, warm_start, workspace,
                 workspace_bytes, stream);
}
int32_t nbl_rollout_backward_checkpointed(nbl_model* m, int64_t B, int32_t T, int32_t segment, double* states, const double* actions,
                                          int64_t action_stride, void* saved, const void* checkpoints, int32_t warm_start,
                                          const double* grad_states, double* grad_state0, double* grad_actions, double* grad_params,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  return NBL_FWD(m, rollout_backward_checkpointed, B, T, segment, states, actions, action_stride, saved, checkpoints, warm_start, grad_states, grad_state0,
                 grad_actions, grad_params, workspace, workspace_bytes, stream);
}

// ---- self-tests: the problem size picks the instantiation whose device code runs ----
int32_t nbl_selftest_lcp_dantzig_timed(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                       const int32_t* findex, double* x, int32_t* rc, int32_t reps, double* ms_per_launch) {
  const Variant* v = &kVariants[n > 192 ? 3 : (n > 48 ? 2 : (n > 24 ? 1 : 0))];     // (n > 48: the general driver, gen_dantzig_dev.hpp; n > 192: its 384-row build)
  g_errVariant = v;
  return v->selftest_lcp_dantzig_timed(count, n, A, b, lo, hi, findex, x, rc, reps, ms_per_launch);
}
int32_t nbl_selftest_lcp_dantzig(int32_t count, int32_t n, const double* A, const double* b, const double* lo, const double* hi,
                                 const int32_t* findex, double* x, int32_t* rc) {
  return nbl_selftest_lcp_dantzig_timed(count, n, A, b, lo, hi, findex, x, rc, 1, nullptr);
}
int32_t nbl_selftest_lcp_cascade(int32_t count, int32_t m, const double* A, const double* b, const double* mu, int32_t have_cache,
                                 const double* x_cache, const uint8_t* on, double fallback_cfm, double* x, int32_t* cls, uint32_t* st, double* cfm) {
  g_errVariant = &kVariants[2];       // (the general instantiation's code, whatever the size)
  return nbl_selftest_lcp_cascade_c64(count, m, A, b, mu, have_cache, x_cache, on, fallback_cfm, x, cls, st, cfm);
}
int32_t nbl_selftest_pinv_rows(int32_t count, int32_t rows, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank,
                               int32_t reps, double* ms_per_launch) {
  if (rows != 24 && rows != 48) return ownError(NBL_E_BADARG, "rows must be 24 or 48");
  const Variant* v = &kVariants[rows == 48 ? 1 : 0];
  g_errVariant = v;
  return v->selftest_pinv_rows(count, rows, Q, cTrue, route, P, rank, reps, ms_per_launch);
}
int32_t nbl_selftest_pinv(int32_t count, const double* Q, const int32_t* cTrue, int32_t route, double* P, int32_t* rank, int32_t reps,
                          double* ms_per_launch) {
  return nbl_selftest_pinv_rows(count, 24, Q, cTrue, route, P, rank, reps, ms_per_launch);
}

int32_t nbl_transpose_to_soa(const double* src_bd, double* dst_db, int64_t B, int32_t d, void* stream) {
  g_errVariant = &kVariants[0];
  return nbl_transpose_to_soa_c8(src_bd, dst_db, B, d, stream);
}
int32_t nbl_transpose_from_soa(const double* src_db, double* dst_bd, int64_t B, int32_t d, void* stream) {
  g_errVariant = &kVariants[0];
  return nbl_transpose_from_soa_c8(src_db, dst_bd, B, d, stream);
}

int32_t nbl_set_launch_lanes(nbl_model* m, int32_t tree_lanes, int32_t lcp_lanes) { return NBL_FWD(m, set_launch_lanes, tree_lanes, lcp_lanes); }
int32_t nbl_set_slices(nbl_model* m, int32_t slices) { return NBL_FWD(m, set_slices, slices); }
int32_t nbl_slices_for(const nbl_model* m, int64_t B) { return NBL_GET(m, slices_for, B); }
int32_t nbl_set_deferred_join(nbl_model* m, int32_t enabled) { return NBL_FWD(m, set_deferred_join, enabled); }
int32_t nbl_slice_stream(nbl_model* m, int64_t B, int32_t slice, void** stream, int64_t* first_world, int64_t* end_world) {
  return NBL_FWD(m, slice_stream, B, slice, stream, first_world, end_world);
}
int32_t nbl_join_slices(nbl_model* m, void* stream) { return NBL_FWD(m, join_slices, stream); }
int32_t nbl_fork_slices(nbl_model* m, void* stream) { return NBL_FWD(m, fork_slices, stream); }
int32_t nbl_set_timing(nbl_model* m, int32_t enabled) { return NBL_FWD(m, set_timing, enabled); }
int32_t nbl_get_timing(nbl_model* m, double* fwd_ms_sum, int64_t* fwd_count, double* bwd_ms_sum, int64_t* bwd_count) {
  return NBL_FWD(m, get_timing, fwd_ms_sum, fwd_count, bwd_ms_sum, bwd_count);
}
int32_t nbl_kernel_count(void) { return nbl_kernel_count_c8(); }
const char* nbl_kernel_name(int32_t i) { return nbl_kernel_name_c8(i); }
int32_t nbl_kernel_timing(nbl_model* m, int32_t i, double* ms_sum, int64_t* count) { return NBL_FWD(m, kernel_timing, i, ms_sum, count); }

}  // extern "C"
