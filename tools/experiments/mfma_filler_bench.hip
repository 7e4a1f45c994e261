// tools/experiments/mfma_filler_bench.hip -- what does a vector instruction cost in the shadow of v_mfma_f32_32x32x16_bf16?
// One wave per SIMD (4-wave workgroups, 256 of them), a chain of MFMAs with N filler instructions of one kind between
// neighbours; prints cycles (s_memtime) per MFMA.  Build: hipcc --offload-arch=gfx950 -O3 -o prof_build/mfma_filler_bench <this>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define FILL_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i & 7]) : "v"(c));
#define FILL_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i & 7]) : "v"(f[i & 7]), "v"(f[(i + 1) & 7]));
#define FILL_AND(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i & 7]));
#define FILL_SHL(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i & 7]));
#define FILL_SUB(i) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(f[i & 7]) : "v"(c));

template <int KIND, int N, bool ALT>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
  f16v acc0 = {0}, acc1 = {0};
  float f[8]; uint32_t u[8]; float c = 1.0001f;
  for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (ALT && (m & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (KIND == 0) { FILL_ADD(q) } else if (KIND == 1) { FILL_CVT(q) } else if (KIND == 2) { FILL_AND(q) } else if (KIND == 3) { FILL_SHL(q) } else { FILL_SUB(q) }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int N, bool ALT>
void run(const char *name, float *out, long long *cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N, ALT>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N, ALT>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
  printf("%-10s N=%d %s: %7.2f ticks/MFMA  %8.3f ms -> %6.1f ns/MFMA\n", name, N, ALT ? "alternating" : "one chain  ", avg / (iters * 12.0), ms, ms * 1e6 / (iters * 12.0));
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0, false>("none", out, cyc); run<0, 0, true>("none", out, cyc);
  run<0, 1, false>("v_add", out, cyc); run<0, 2, false>("v_add", out, cyc); run<0, 4, false>("v_add", out, cyc); run<0, 6, false>("v_add", out, cyc);
  run<0, 2, true>("v_add", out, cyc); run<0, 4, true>("v_add", out, cyc);
  run<1, 1, false>("cvt_pk", out, cyc); run<1, 2, false>("cvt_pk", out, cyc); run<1, 4, false>("cvt_pk", out, cyc); run<1, 2, true>("cvt_pk", out, cyc);
  run<2, 2, false>("v_and", out, cyc); run<2, 4, false>("v_and", out, cyc);
  run<3, 2, false>("v_lshl", out, cyc); run<3, 4, false>("v_lshl", out, cyc);
  run<4, 2, false>("v_sub_e64", out, cyc); run<4, 4, false>("v_sub_e64", out, cyc); run<4, 4, true>("v_sub_e64", out, cyc);
  return 0;
}
