// Measured fp64 rates of one MI355X: v_fma_f64 (vector ALU) against v_mfma_f64_16x16x4_f64 (matrix core), both issued from
// registers with no memory traffic, 8 waves per SIMD.  Answers "would the dense 24 x 24 blocks of the contact stage run faster
// on MFMA?": only if the matrix core's fp64 rate exceeded the vector rate.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_rate.hip -o tools/dbg/fp64_rate && tools/dbg/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a, double b) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double av = a + threadIdx.x, bv = b;
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
  }
  const double4_t s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main() {
  const int blocks = 256 * 8, threads = 256, iters = 20000;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0); k_fma<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 8 * (double)iters * blocks * threads;
    if (rep) printf("{\"kernel\": \"v_fma_f64\", \"ms\": %.3f, \"TFLOPs\": %.2f}\n", ms, fl / ms * 1e-9);
    hipEventRecord(e0); k_mfma<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double fm = 2.0 * 16 * 16 * 4 * 4 * (double)iters * blocks * (threads / 64);
    if (rep) printf("{\"kernel\": \"v_mfma_f64_16x16x4_f64\", \"ms\": %.3f, \"TFLOPs\": %.2f}\n", ms, fm / ms * 1e-9);
  }
  return 0;
}
