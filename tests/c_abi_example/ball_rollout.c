/* A plain C program against the C ABI (include/nimble_amd.h): a body on a BALL joint (NBL_JOINT_BALL: one body, three DOFs in the
 * description; the library expands it internally) rolled out for T steps, once with all T backward records resident and once
 * checkpointed (nbl_rollout_*_checkpointed, `segment` records resident).  Prints, per world, the final state (6 numbers) and the gradient
 * of L = sum(states[T]) with respect to the initial state (6 numbers) for both runs: the test compares them with each other (bit for
 * bit) and with the CPU oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nimble_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nbl_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4;
  const int32_t T = argc > 2 ? atoi(argv[2]) : 6, segment = argc > 3 ? atoi(argv[3]) : 4;
  enum { N = 3, S = 6 };
  nbl_model_desc d;
  memset(&d, 0, sizeof(d));
  int32_t parent[1] = {-1}, jtype[1] = {NBL_JOINT_BALL}, dofoff[1] = {0}, amap[3] = {0, 1, 2};
  double Tpj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0.5, 0};
  double Tcj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.1, 0.3, -0.05};
  double axis[3] = {0, 0, 1}, mass[1] = {2.0}, com[3] = {0.02, -0.01, 0.03}, inertia[6] = {0.4, 0.5, 0.6, 0.01, -0.02, 0.03};
  double damping[3] = {0.3, 0.2, 0.1};
  d.n_bodies = 1; d.n_dofs = N;
  d.parent = parent; d.joint_type = jtype; d.dof_offset = dofoff;
  d.T_pj = Tpj; d.T_cj = Tcj; d.axis = axis; d.mass = mass; d.com = com; d.inertia = inertia; d.damping = damping;
  d.gravity[0] = 0; d.gravity[1] = -9.81; d.gravity[2] = 0; d.dt = 1e-3;
  d.n_action = 3; d.action_map = amap;
  d.contact_clipping_depth = 0.03; d.fallback_cfm = 1e-4;
  nbl_model* m = NULL;
  CHECK(nbl_model_create(&d, 0, &m));
  const size_t wsBytes = nbl_rollout_workspace_bytes(m, B), svBytes = nbl_saved_bytes(m, B);
  const size_t ckBytes = nbl_rollout_checkpoint_bytes(m, B, T, segment);
  double *state0, *action, *states, *gstates, *g0, *ga;
  void *ws, *saved, *ckpt;
  HIP(hipMalloc((void**)&state0, S * B * sizeof(double))); HIP(hipMalloc((void**)&action, N * B * sizeof(double)));
  HIP(hipMalloc((void**)&states, (size_t)(T + 1) * S * B * sizeof(double))); HIP(hipMalloc((void**)&gstates, (size_t)(T + 1) * S * B * sizeof(double)));
  HIP(hipMalloc((void**)&g0, S * B * sizeof(double))); HIP(hipMalloc((void**)&ga, (size_t)T * N * B * sizeof(double)));
  HIP(hipMalloc(&ws, wsBytes)); HIP(hipMalloc(&saved, (size_t)T * svBytes)); HIP(hipMalloc(&ckpt, ckBytes ? ckBytes : 256));
  double* h = (double*)malloc((size_t)(T + 1) * S * B * sizeof(double));
  for (int r = 0; r < S; r++)                         /* DOF-major [row][B]: rows 0-2 = q (exponential map), rows 3-5 = angular velocity */
    for (int64_t b = 0; b < B; b++) h[r * B + b] = (r < 3 ? 0.4 : 1.5) * ((double)((r * 7 + b * 3) % 11) / 5.0 - 1.0);
  HIP(hipMemcpy(state0, h, S * B * sizeof(double), hipMemcpyHostToDevice));
  for (int r = 0; r < N; r++) for (int64_t b = 0; b < B; b++) h[r * B + b] = 0.5 - 0.1 * (double)((r + b) % 7);
  HIP(hipMemcpy(action, h, N * B * sizeof(double), hipMemcpyHostToDevice));
  memset(h, 0, (size_t)(T + 1) * S * B * sizeof(double));
  for (int64_t i = 0; i < S * B; i++) h[(size_t)T * S * B + i] = 1.0;               /* L = sum(states[T]) */
  HIP(hipMemcpy(gstates, h, (size_t)(T + 1) * S * B * sizeof(double), hipMemcpyHostToDevice));
  double* out = (double*)malloc(4 * S * B * sizeof(double));
  for (int run = 0; run < 2; run++) {
    const int32_t seg = run == 0 ? 0 : segment;      /* one [k][B] action block for every step: action_stride = 0 */
    CHECK(nbl_rollout_forward_checkpointed(m, B, T, seg, state0, action, 0, states, saved, ckpt, NULL, 1, ws, wsBytes, NULL));
    CHECK(nbl_rollout_backward_checkpointed(m, B, T, seg, states, action, 0, saved, ckpt, 1, gstates, g0, ga, NULL, ws, wsBytes, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(out + (size_t)run * 2 * S * B, states + (size_t)T * S * B, S * B * sizeof(double), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(out + (size_t)run * 2 * S * B + S * B, g0, S * B * sizeof(double), hipMemcpyDeviceToHost));
  }
  for (int64_t b = 0; b < B; b++) {
    for (int k = 0; k < 4 * S; k++) printf("%.17g ", out[(size_t)(k / S) * S * B + (k % S) * B + b]);
    printf("\n");
  }
  nbl_model_destroy(m);
  return 0;
}
