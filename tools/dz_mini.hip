// tools/dz_mini.hip - developer harness: ONLY the cooperative Dantzig driver (csrc/coop_dantzig_dev.hpp) behind a tiny C entry point, so that
// a change to it rebuilds in seconds (the whole library takes minutes).  Same kernel body as k_selftest_dantzig / nbl_selftest_lcp_dantzig_timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I nimblephysics_amd/csrc tools/dz_mini.hip -o tools/dbg/libdz_mini.so
// driver: tools/dz_mini.py (problems of the metric distribution, bit-exactness against oracle/_ref, launch latency and throughput)
#include <hip/hip_runtime.h>
#include "coop_dev.hpp"
#include "coop_dantzig_dev.hpp"
#include "coop_wave_dev.hpp"

using namespace NBL_NS;

__global__ __launch_bounds__(64) void k_dz(int count, int nmax, const int32_t* __restrict__ ns, const double* __restrict__ A, const double* __restrict__ b, const double* __restrict__ lo,
                                           const double* __restrict__ hi, const int32_t* __restrict__ findex, double* __restrict__ x, int32_t* __restrict__ rc) {
  __shared__ CascadeLds C;
  const DevWave w;
  const int ln = w.lane();
  const int64_t pb = blockIdx.x;
  if (pb >= count) return;
  const int n = ns[pb];                                        // rows of this problem (<= nmax: the arrays are padded to nmax)
  A += pb * nmax * nmax; b += pb * nmax; lo += pb * nmax; hi += pb * nmax; findex += pb * nmax; x += pb * nmax;
  for (int i = ln; i < MAXR * CLD; i += 64) { C.A[i] = 0.0; C.L[i] = 0.0; }
  w.sync();
  if (ln < n) for (int j = 0; j < n; j++) C.A[ln * CLD + j] = A[ln * nmax + j];
  w.sync();
  CoopLcpRow row;
  const bool on = ln < n;
  row.x = 0.0; row.b = on ? b[ln] : 0.0; row.lo = on ? lo[ln] : 0.0; row.hi = on ? hi[ln] : 0.0;
  row.findex = on ? findex[ln] : -1;
  const int r = coopDantzig(w, C, n, row);
  if (on) x[ln] = row.x;
  if (ln == 0) rc[pb] = r;
}

extern "C" int dz_run(int count, int n, const int32_t* ns, const double* A, const double* b, const double* lo, const double* hi, const int32_t* findex, double* x,
                      int32_t* rc, int reps, double* ms_per_launch) {
  const size_t nv = (size_t)count * n, nm = nv * n;
  double *dA = nullptr, *dv = nullptr;
  int32_t* di = nullptr;
  int32_t* dn = nullptr;
  if (hipMalloc((void**)&dA, nm * 8) != hipSuccess || hipMalloc((void**)&dv, 4 * nv * 8) != hipSuccess || hipMalloc((void**)&di, (nv + count) * 4) != hipSuccess ||
      hipMalloc((void**)&dn, count * 4) != hipSuccess) return -1;
  hipMemcpy(dn, ns, count * 4, hipMemcpyHostToDevice);
  hipMemcpy(dA, A, nm * 8, hipMemcpyHostToDevice); hipMemcpy(dv, b, nv * 8, hipMemcpyHostToDevice);
  hipMemcpy(dv + nv, lo, nv * 8, hipMemcpyHostToDevice); hipMemcpy(dv + 2 * nv, hi, nv * 8, hipMemcpyHostToDevice);
  hipMemcpy(di, findex, nv * 4, hipMemcpyHostToDevice);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  hipLaunchKernelGGL(k_dz, dim3(count), dim3(64), 0, 0, count, n, dn, dA, dv, dv + nv, dv + 2 * nv, di, dv + 3 * nv, di + nv);
  hipEventRecord(t0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_dz, dim3(count), dim3(64), 0, 0, count, n, dn, dA, dv, dv + nv, dv + 2 * nv, di, dv + 3 * nv, di + nv);
  hipEventRecord(t1, 0);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  float ms = 0.f;
  hipEventElapsedTime(&ms, t0, t1);
  if (ms_per_launch) *ms_per_launch = (double)ms / reps;
  hipMemcpy(x, dv + 3 * nv, nv * 8, hipMemcpyDeviceToHost); hipMemcpy(rc, di + nv, count * 4, hipMemcpyDeviceToHost);
  hipEventDestroy(t0); hipEventDestroy(t1);
  hipFree(dA); hipFree(dv); hipFree(di); hipFree(dn);
  return 0;
}
