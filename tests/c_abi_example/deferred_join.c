/* The DEFERRED JOIN of the C ABI (include/nimble_amd.h, ABI minor 5) from a plain C program: ONE model handle, a batch the library cuts
 * into slices, K steps forward + backward without a join per call - the caller's per-slice work (here: the loss gradient g = the next
 * state, a strided device-to-device copy) is enqueued on the slice's own stream (nbl_slice_stream) - against the same K steps with
 * joined calls.  The two runs must agree BYTE FOR BYTE (next state, status, both gradients, the warm start): the deferred join changes
 * where the work is ordered, not what is computed.  Prints one line: slices, bytes compared, and "identical" or the first difference.
 * (What a cgo / JNI / N-API binding of the reference would do around its own training loop: tests/test_gpu_c_abi.py runs it.) */
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

typedef struct { double *next, *gs, *ga, *cache; uint32_t* status; } Result;

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int K = argc > 2 ? atoi(argv[2]) : 3;
  nbl_model_desc d;
  memset(&d, 0, sizeof(d));
  mdl_fill(&d);
  nbl_model* m = NULL;
  CHECK(nbl_model_create(&d, 0, &m));
  if ((nbl_version() & 0xffff) < 5) { fprintf(stderr, "ABI minor %d < 5\n", nbl_version() & 0xffff); return 1; }
  const int n = nbl_model_num_dofs(m), k = nbl_model_num_action(m), rows = nbl_model_lcp_rows(m);
  const size_t wsBytes = nbl_workspace_bytes(m, B), svBytes = nbl_saved_bytes(m, B);
  const size_t nS = (size_t)2 * n * B, nA = (size_t)k * B, nC = (size_t)rows * B;
  double *s0 = dalloc(nS), *act = dalloc(nA), *next = dalloc(nS), *g = dalloc(nS), *gs = dalloc(nS), *ga = dalloc(nA), *cache = dalloc(nC);
  void *ws = NULL, *sv = NULL;
  uint32_t* st = NULL;
  HIP(hipMalloc(&ws, wsBytes)); HIP(hipMalloc(&sv, svBytes)); HIP(hipMalloc((void**)&st, B * sizeof(uint32_t)));
  if (!s0 || !act || !next || !g || !gs || !ga || !cache) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  double* h = (double*)malloc(nS * sizeof(double));
  for (int r = 0; r < 2 * n; r++)          /* the standing pose with a deterministic spread of the joints: some worlds resolve at LCP stage 0, some do not */
    for (int64_t b = 0; b < B; b++) {
      double x = 0.0;
      if (r == 0) x = -1.5707963267948966;
      else if (r == 4) x = -0.01;
      else if (r >= 6 && r < n) x = 0.03 * sin(1.0 + 3.0 * (double)b + 7.0 * (double)r);
      else if (r >= n) x = 0.01 * cos(2.0 + 5.0 * (double)b + 11.0 * (double)(r - n));
      h[(size_t)r * B + b] = x;
    }
  HIP(hipMemcpy(s0, h, nS * sizeof(double), hipMemcpyHostToDevice));
  for (int r = 0; r < k; r++) for (int64_t b = 0; b < B; b++) h[(size_t)r * B + b] = 0.1 * sin((double)b + (double)r);
  HIP(hipMemcpy(act, h, nA * sizeof(double), hipMemcpyHostToDevice));
  hipStream_t stream = NULL;
  HIP(hipStreamCreate(&stream));
  Result res[2];
  int slices = 0;
  for (int pass = 0; pass < 2; pass++) {            /* pass 0: joined calls, pass 1: deferred join */
    CHECK(nbl_set_deferred_join(m, pass));
    slices = nbl_slices_for(m, B);
    HIP(hipMemsetAsync(next, 0, nS * sizeof(double), stream)); HIP(hipMemsetAsync(gs, 0, nS * sizeof(double), stream));
    HIP(hipMemsetAsync(ga, 0, nA * sizeof(double), stream)); HIP(hipMemsetAsync(cache, 0, nC * sizeof(double), stream));
    if (pass) CHECK(nbl_fork_slices(m, stream));    /* the slice streams start behind what `stream` holds (the inputs, the memsets) */
    for (int t = 0; t < K; t++) {
      CHECK(nbl_step_forward(m, B, s0, act, NULL, next, cache, sv, st, ws, wsBytes, stream));      /* cold LCP start; warm start out */
      if (!pass) HIP(hipMemcpyAsync(g, next, nS * sizeof(double), hipMemcpyDeviceToDevice, stream));  /* the "loss": g = next state */
      else
        for (int i = 0; i < slices; i++) {          /* ... per slice, on the slice's own stream: rows [0, 2n) x worlds [b0, b1) of a [row][B] array */
          void* si = NULL; int64_t b0 = 0, b1 = 0;
          CHECK(nbl_slice_stream(m, B, i, &si, &b0, &b1));
          HIP(hipMemcpy2DAsync(g + b0, (size_t)B * sizeof(double), next + b0, (size_t)B * sizeof(double), (size_t)(b1 - b0) * sizeof(double), (size_t)2 * n,
                               hipMemcpyDeviceToDevice, si ? (hipStream_t)si : stream));
        }
      CHECK(nbl_step_backward(m, B, sv, g, gs, ga, ws, wsBytes, stream));
    }
    if (pass) CHECK(nbl_join_slices(m, stream));    /* `stream` waits for every slice before the results are read */
    HIP(hipStreamSynchronize(stream));
    Result* r = &res[pass];
    r->next = (double*)malloc(nS * sizeof(double)); r->gs = (double*)malloc(nS * sizeof(double)); r->ga = (double*)malloc(nA * sizeof(double));
    r->cache = (double*)malloc(nC * sizeof(double)); r->status = (uint32_t*)malloc(B * sizeof(uint32_t));
    HIP(hipMemcpy(r->next, next, nS * sizeof(double), hipMemcpyDeviceToHost)); HIP(hipMemcpy(r->gs, gs, nS * sizeof(double), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r->ga, ga, nA * sizeof(double), hipMemcpyDeviceToHost)); HIP(hipMemcpy(r->cache, cache, nC * sizeof(double), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r->status, st, B * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  CHECK(nbl_set_deferred_join(m, 0));
  int64_t contact = 0, stage0 = 0, finite = 1;
  for (int64_t b = 0; b < B; b++) { contact += (res[1].status[b] & NBL_ST_CONTACT) != 0; stage0 += (res[1].status[b] & NBL_ST_LCP_STAGE0) != 0; }
  for (size_t i = 0; i < nS; i++) if (!isfinite(res[1].next[i]) || !isfinite(res[1].gs[i])) finite = 0;
  const char* what = NULL;
  if (memcmp(res[0].next, res[1].next, nS * sizeof(double))) what = "next state";
  else if (memcmp(res[0].gs, res[1].gs, nS * sizeof(double))) what = "state gradient";
  else if (memcmp(res[0].ga, res[1].ga, nA * sizeof(double))) what = "action gradient";
  else if (memcmp(res[0].cache, res[1].cache, nC * sizeof(double))) what = "warm start";
  else if (memcmp(res[0].status, res[1].status, B * sizeof(uint32_t))) what = "status";
  printf("worlds %lld steps %d slices %d contact %lld stage0 %lld finite %lld bytes %zu %s%s\n", (long long)B, K, slices, (long long)contact, (long long)stage0,
         (long long)finite, (2 * nS + nA + nC) * sizeof(double) + B * sizeof(uint32_t), what ? "DIFFERENT: " : "identical", what ? what : "");
  nbl_model_destroy(m);
  return what ? 2 : 0;
}
