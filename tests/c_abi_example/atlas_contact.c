/* The CONTACT fields of the C ABI from a plain C program (include/nimble_amd.h; no Python, no torch, no C++): the metric model - Atlas with
 * 20 DOFs standing on the ground box, 3 box colliders, 8 frictional foot-corner contacts = 24 LCP rows - described through
 * nbl_model_desc's collider / skeleton / max_contacts fields (atlas20_ground_model.h: generated data), two chained steps of B worlds with
 * the LCP warm start handed from the first to the second (lcp_cache_out -> lcp_cache_in: BoxedLcpConstraintSolver's mX,
 * BoxedLcpConstraintSolver.cpp:176-187), the per-world status words, and the backward pass through both steps.
 * Prints, per world: status of step 1 and 2, the state after step 2, dL/d(state 0), dL/d(action) summed over the steps, and the warm
 * start leaving step 2 (row count + impulses) - tests/test_gpu_c_abi.py compares them with the oracle chain. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "nimble_amd.h"
#include "atlas20_ground_model.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nbl_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double* dalloc(size_t doubles) { void* p = NULL; return hipMalloc(&p, doubles * sizeof(double)) == hipSuccess ? (double*)p : NULL; }

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4;
  nbl_model_desc d;
  memset(&d, 0, sizeof(d));
  mdl_fill(&d);
  nbl_model* m = NULL;
  CHECK(nbl_model_create(&d, 0, &m));
  const int n = nbl_model_num_dofs(m), k = nbl_model_num_action(m), rows = nbl_model_lcp_rows(m);   /* rows = 3 * contacts + 1 (the row count) */
  if (n != MDL_N_DOFS || k != MDL_N_ACTION || rows != 3 * nbl_model_max_contacts(m) + 1 || nbl_model_max_contacts(m) < MDL_MAX_CONTACTS) {
    fprintf(stderr, "unexpected model sizes: n %d k %d lcp rows %d max contacts %d\n", n, k, rows, nbl_model_max_contacts(m));
    return 1;
  }
  const size_t wsBytes = nbl_workspace_bytes(m, B), svBytes = nbl_saved_bytes(m, B);
  double *s0 = dalloc(2 * n * B), *act = dalloc((size_t)k * B), *s1 = dalloc(2 * n * B), *s2 = dalloc(2 * n * B);
  double *c1 = dalloc((size_t)rows * B), *c2 = dalloc((size_t)rows * B);
  double *g2 = dalloc(2 * n * B), *g1 = dalloc(2 * n * B), *g0 = dalloc(2 * n * B), *ga1 = dalloc((size_t)k * B), *ga2 = dalloc((size_t)k * B);
  void *ws = NULL, *sv1 = NULL, *sv2 = NULL;
  uint32_t *st1 = NULL, *st2 = NULL;
  HIP(hipMalloc(&ws, wsBytes)); HIP(hipMalloc(&sv1, svBytes)); HIP(hipMalloc(&sv2, svBytes));
  HIP(hipMalloc((void**)&st1, B * sizeof(uint32_t))); HIP(hipMalloc((void**)&st2, B * sizeof(uint32_t)));
  if (!s0 || !act || !s1 || !s2 || !c1 || !c2 || !g2 || !g1 || !g0 || !ga1 || !ga2) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  double* h = (double*)malloc((size_t)(2 * n + rows) * B * sizeof(double));
  /* DOF-major layout [row][B]: rows 0..n-1 = q, n..2n-1 = v.  The standing pose of test_AtlasGradients.cpp:235-236 (q[0] = -pi/2,
   * q[4] = -0.01) with a small deterministic perturbation of the joints and velocities */
  for (int r = 0; r < 2 * n; r++)
    for (int64_t b = 0; b < B; b++) {
      double x = 0.0;
      if (r == 0) x = -1.5707963267948966;
      else if (r == 4) x = -0.01;
      else if (r >= 6 && r < n) x = 0.002 * sin(1.0 + 3.0 * (double)b + 7.0 * (double)r);
      else if (r >= n) x = 0.001 * cos(2.0 + 5.0 * (double)b + 11.0 * (double)(r - n));
      h[(size_t)r * B + b] = x;
    }
  HIP(hipMemcpy(s0, h, (size_t)2 * n * B * sizeof(double), hipMemcpyHostToDevice));
  for (int r = 0; r < k; r++) for (int64_t b = 0; b < B; b++) h[(size_t)r * B + b] = 0.1 * sin((double)b + (double)r);
  HIP(hipMemcpy(act, h, (size_t)k * B * sizeof(double), hipMemcpyHostToDevice));
  for (int r = 0; r < 2 * n; r++) for (int64_t b = 0; b < B; b++) h[(size_t)r * B + b] = cos((double)r + 2.0 * (double)b);
  HIP(hipMemcpy(g2, h, (size_t)2 * n * B * sizeof(double), hipMemcpyHostToDevice));
  /* step 1: cold start (no warm start handed in), step 2: from step 1's solution */
  CHECK(nbl_step_forward(m, B, s0, act, NULL, s1, c1, sv1, st1, ws, wsBytes, NULL));
  CHECK(nbl_step_forward(m, B, s1, act, c1, s2, c2, sv2, st2, ws, wsBytes, NULL));
  CHECK(nbl_step_backward(m, B, sv2, g2, g1, ga2, ws, wsBytes, NULL));
  CHECK(nbl_step_backward(m, B, sv1, g1, g0, ga1, ws, wsBytes, NULL));
  HIP(hipDeviceSynchronize());
  uint32_t* hs = (uint32_t*)malloc(2 * B * sizeof(uint32_t));
  double *o2 = (double*)malloc((size_t)2 * n * B * sizeof(double)), *og = (double*)malloc((size_t)2 * n * B * sizeof(double));
  double *oa1 = (double*)malloc((size_t)k * B * sizeof(double)), *oa2 = (double*)malloc((size_t)k * B * sizeof(double));
  HIP(hipMemcpy(hs, st1, B * sizeof(uint32_t), hipMemcpyDeviceToHost)); HIP(hipMemcpy(hs + B, st2, B * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(o2, s2, (size_t)2 * n * B * sizeof(double), hipMemcpyDeviceToHost)); HIP(hipMemcpy(og, g0, (size_t)2 * n * B * sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(oa1, ga1, (size_t)k * B * sizeof(double), hipMemcpyDeviceToHost)); HIP(hipMemcpy(oa2, ga2, (size_t)k * B * sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(h, c2, (size_t)rows * B * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t b = 0; b < B; b++) {
    printf("%u %u", hs[b], hs[B + b]);
    for (int r = 0; r < 2 * n; r++) printf(" %.17g", o2[(size_t)r * B + b]);
    for (int r = 0; r < 2 * n; r++) printf(" %.17g", og[(size_t)r * B + b]);
    for (int r = 0; r < k; r++) printf(" %.17g", oa1[(size_t)r * B + b] + oa2[(size_t)r * B + b]);
    printf(" %.17g", h[(size_t)(rows - 1) * B + b]);                                   /* LCP rows the warm start belongs to */
    for (int r = 0; r < rows - 1; r++) printf(" %.17g", h[(size_t)r * B + b]);
    printf("\n");
  }
  nbl_model_destroy(m);
  return 0;
}
