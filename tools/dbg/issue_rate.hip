// Issue-rate probe (developer): how many instructions per cycle does ONE SIMD issue when 1, 2, 4 or 8 wavefronts share it?  One workgroup
// on one CU, 4 x W wavefronts (W per SIMD), every wavefront runs the same unrolled loop of INDEPENDENT v_add_f64; cycles by clock64.
// Measured (MI355X, round 6): 1 wavefront per SIMD 5.45 cycles per instruction; 2: 9.09 per wavefront = 0.220 instructions per cycle per
// SIMD; 4: 17.2 = 0.233 - a SIMD issues one f64 VALU instruction per ~4.3 cycles however many wavefronts share it, a lone wavefront
// gets one per 5.45.  (A scalar variant of the loop must declare the SCC clobber of s_add_u32: without it the loop control is corrupted.)
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/issue_rate.hip -o tools/dbg/issue_rate && tools/dbg/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed, int iters) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001;
  int s0 = 1, s1 = 2;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {        // VALU f64 only: 4 independent chains
      REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  out[threadIdx.x] = a0 + a1 + a2 + a3 + s0 + s1;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 2048 * sizeof(double)); hipMalloc(&cyc, 32 * sizeof(long long));
  const int iters = 256;
  const char* names[1] = {"v_add_f64 x 4 (independent)"};
  for (int mode = 0; mode < 1; mode++)
    for (int W : {1, 2, 4, 8}) {            // wavefronts per SIMD
      const int waves = 4 * W;
      if (waves * 64 > 1024) continue;
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0, iters);
        hipDeviceSynchronize();
      }
      long long h[32]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      long long mx = 0; for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
      const double instr = (double)iters * 16 * 4;              // per wavefront
      // clock64 = s_memtime counts at 100 MHz on this part: convert with the shader clock the box reports if needed; reported raw and per instruction
      printf("%-30s %d wave(s) per SIMD: %lld ticks for %.0f instr per wave -> %.3f ticks per instr per wave, %.3f instr per tick per SIMD\n", names[mode], W, mx, instr,
             mx / instr, W * instr / mx);
    }
  return 0;
}
