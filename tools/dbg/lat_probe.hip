// Latency probe (developer): cycles per link of the dependent chains the lane = row Dantzig / PGS drivers are made of, ONE wavefront on a CU.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/lat_probe.hip -o tools/dbg/lat_probe && tools/dbg/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)
__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x, b = 1.0000001, c = 0.5;
  long long t0, t1;
  int idx = 0;
#define RUN(NAME, ...)                                           \
  t0 = clock64();                                                \
  for (int i = 0; i < N / 16; i++) { REP16(__VA_ARGS__) }               \
  t1 = clock64();                                                \
  if (threadIdx.x == 0) cyc[idx] = t1 - t0;                      \
  idx++;
  // 0: dependent v_add_f64
  RUN("add", asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
  // 1: dependent v_mul_f64 + v_add_f64
  RUN("muladd", asm volatile("v_mul_f64 %2, %0, %1\n\tv_add_f64 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));)
  // 2: readlane x2 -> SGPR -> v_add (broadcast link)
  RUN("readlane-add", { const int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
                        const double s = __hiloint2double(hi, lo); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "s"(s)); })
  // 3: readlane x2 -> v_mul (SGPR operand) -> v_add : one substitution step's chain
  RUN("readlane-mul-add", { const int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
                            const double s = __hiloint2double(hi, lo); double t; asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1" : "+v"(a), "=&v"(t) : "v"(b), "s"(s)); })
  // 4: DPP quad_perm broadcast (lane 0 of each quad) x2 -> v_mul -> v_add
  RUN("dpp-mul-add", { const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x00, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x00, 0xf, 0xf, false);
                       const double s = __hiloint2double(hi, lo); double t; asm volatile("v_mul_f64 %1, %2, %3\n\tv_add_f64 %0, %0, -%1" : "+v"(a), "=&v"(t) : "v"(b), "v"(s)); })
  // 5: the round-5 substitution step: readlane x2, then 7 instructions with EXEC save / two masks / restore
  RUN("step-r5", { const int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
                   const double s = __hiloint2double(hi, lo); double t; unsigned long long sv;
                   asm volatile("s_mov_b64 %2, exec\n\tv_mul_f64 %1, %3, %4\n\ts_mov_b64 exec, 0xe\n\tv_add_f64 %0, %0, -%1\n\ts_mov_b64 exec, 0xfffff0\n\tv_add_f64 %0, %0, %1\n\ts_mov_b64 exec, %2"
                                : "+v"(a), "=&v"(t), "=&s"(sv) : "v"(b), "s"(s)); })
  // 6: ds_bpermute x2 -> v_add
  RUN("bpermute-add", { const int lo = __builtin_amdgcn_ds_bpermute(12, __double2loint(a)), hi = __builtin_amdgcn_ds_bpermute(12, __double2hiint(a));
                        const double s = __hiloint2double(hi, lo); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(s)); })
  // 7: v_max_f64 dependent
  RUN("max", asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
  // 8: v_fma_f64 dependent
  RUN("fma", asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
  // 9: independent adds (8 accumulators round robin would need more code: two here)
  { double a2 = a + 1.0;
    RUN("add-2chains", asm volatile("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2" : "+v"(a), "+v"(a2) : "v"(b));)
    a += a2; }
  // 10: s_mov exec pairs between dependent adds
  RUN("add+2smov", { unsigned long long sv; asm volatile("s_mov_b64 %1, exec\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, %1" : "+v"(a), "=&s"(sv) : "v"(b)); })
  // 11: dependent add + 4 s_nop 0
  RUN("add+4nop", asm volatile("v_add_f64 %0, %0, %1\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" : "+v"(a) : "v"(b));)
  // 12: v_cndmask pair dependent
  RUN("cndmask2", { int lo = __double2loint(a), hi = __double2hiint(a); asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc" : "+v"(lo), "+v"(hi) : : ); a = __hiloint2double(hi, lo); })
  // 13: f32 dependent add
  { float f = (float)a; RUN("add-f32", asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));) a += f; }
  out[threadIdx.x] = a;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8);
  hipMemset(cyc, 0, 32 * 8);
  const char* names[] = {"v_add_f64", "v_mul_f64 + v_add_f64", "readlane x2 -> v_add(sgpr)", "readlane x2 -> v_mul(sgpr) -> v_add", "dpp quad_perm x2 -> v_mul -> v_add",
                         "round-5 substitution step (2 readlane + 7 asm)", "ds_bpermute x2 -> v_add", "v_max_f64", "v_fma_f64", "two independent v_add_f64 (per pair)",
                         "v_add_f64 + 2 s_mov exec", "v_add_f64 + 4 s_nop", "2 dependent v_cndmask_b32", "v_add_f32"};
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
  }
  long long h[32];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 14; i++) printf("%-52s %7.1f cycles per link\n", names[i], (double)h[i] / N);
  return 0;
}
