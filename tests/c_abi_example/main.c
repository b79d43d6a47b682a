/* A plain C program against the C ABI (include/nimble_amd.h): no Python, no torch, no C++.
 * Single pendulum (data/skel/test/single_pendulum.skel: revolute z, m = 5, I = diag(1, 2, 3), damping 10), B worlds,
 * one forward and one backward step; prints next state and gradients for the test to compare with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nimble_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nbl_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4;
  nbl_model_desc d;
  memset(&d, 0, sizeof(d));
  int32_t parent[1] = {-1}, jtype[1] = {NBL_JOINT_REVOLUTE}, dofoff[1] = {0}, amap[1] = {0};
  double Tid[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  double Tcj[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, -0.1, 0, 0};   /* joint frame in the child body: R row-major, then p (the .skel's joint offset) */
  double axis[3] = {0, 0, 1}, mass[1] = {5.0}, com[3] = {0, 0, 0}, inertia[6] = {1, 2, 3, 0, 0, 0}, damping[1] = {10.0};
  d.n_bodies = 1; d.n_dofs = 1;
  d.parent = parent; d.joint_type = jtype; d.dof_offset = dofoff;
  d.T_pj = Tid; d.T_cj = Tcj; d.axis = axis; d.mass = mass; d.com = com; d.inertia = inertia; d.damping = damping;
  d.gravity[0] = 0; d.gravity[1] = -9.81; d.gravity[2] = 0; d.dt = 1e-3;
  d.n_action = 1; d.action_map = amap;
  d.contact_clipping_depth = 0.03; d.fallback_cfm = 1e-4;
  nbl_model* m = NULL;
  CHECK(nbl_model_create(&d, 0, &m));
  const size_t wsBytes = nbl_workspace_bytes(m, B), svBytes = nbl_saved_bytes(m, B);
  double *state, *action, *next, *gnext, *gstate, *gaction;
  void *ws, *saved;
  uint32_t* status;
  HIP(hipMalloc((void**)&state, 2 * B * sizeof(double))); HIP(hipMalloc((void**)&action, B * sizeof(double)));
  HIP(hipMalloc((void**)&next, 2 * B * sizeof(double))); HIP(hipMalloc((void**)&gnext, 2 * B * sizeof(double)));
  HIP(hipMalloc((void**)&gstate, 2 * B * sizeof(double))); HIP(hipMalloc((void**)&gaction, B * sizeof(double)));
  HIP(hipMalloc(&ws, wsBytes)); HIP(hipMalloc(&saved, svBytes)); HIP(hipMalloc((void**)&status, B * sizeof(uint32_t)));
  double* h = (double*)malloc(2 * B * sizeof(double));
  /* DOF-major layout [row][B]: row 0 = q, row 1 = v */
  for (int64_t b = 0; b < B; b++) { h[b] = 0.3 + 0.1 * (double)b; h[B + b] = -0.5 + 0.2 * (double)b; }
  HIP(hipMemcpy(state, h, 2 * B * sizeof(double), hipMemcpyHostToDevice));
  for (int64_t b = 0; b < B; b++) h[b] = 0.7 - 0.05 * (double)b;
  HIP(hipMemcpy(action, h, B * sizeof(double), hipMemcpyHostToDevice));
  for (int64_t b = 0; b < B; b++) { h[b] = 1.0; h[B + b] = -2.0; }
  HIP(hipMemcpy(gnext, h, 2 * B * sizeof(double), hipMemcpyHostToDevice));
  CHECK(nbl_step_forward(m, B, state, action, NULL, next, NULL, saved, status, ws, wsBytes, NULL));
  CHECK(nbl_step_backward(m, B, saved, gnext, gstate, gaction, ws, wsBytes, NULL));
  HIP(hipDeviceSynchronize());
  double* out = (double*)malloc(5 * B * sizeof(double));
  HIP(hipMemcpy(out, next, 2 * B * sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(out + 2 * B, gstate, 2 * B * sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(out + 4 * B, gaction, B * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t b = 0; b < B; b++)
    printf("%.17g %.17g %.17g %.17g %.17g\n", out[b], out[B + b], out[2 * B + b], out[3 * B + b], out[4 * B + b]);
  nbl_model_destroy(m);
  return 0;
}
