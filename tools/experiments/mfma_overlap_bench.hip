// tools/experiments/mfma_overlap_bench.hip -- how well do matrix-pipe work and vector work of the SAME SIMD overlap, within one
// wave and between the two waves of a 256-register kernel?  The shape is the embedding passes' (leafnet.hip): per "step" NM
// MFMAs on four accumulator chains and NV independent vector instructions (a mixed stream: v_add_f32, v_and_b32, v_sub_f32,
// v_perm_b32 -- 4 cycles each per `profiles/r04_valu_issue.json`).
//   same   : every wave runs MFMAs and vector instructions interleaved (NV / NM after each MFMA)
//   split  : waves 0..3 of the workgroup (one per SIMD) run ONLY the MFMAs of both, waves 4..7 ONLY the vector work of both
//   phased : every wave alternates 8 steps of MFMAs only with 8 steps of vector work only; the two waves of a SIMD start in
//            opposite phases
// W = waves per SIMD (workgroup = 256 W threads, one workgroup per CU).  Prints ns per step and, at the measured clock, cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o prof_build/mfma_overlap_bench tools/experiments/mfma_overlap_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

enum { MODE_SAME, MODE_SPLIT, MODE_PHASED };

#define V4(i)                                                                                                                        \
  asm volatile("v_add_f32 %0, %0, %4\n v_and_b32 %1, 0xffff0000, %1\n v_sub_f32 %2, %2, %4\n v_perm_b32 %3, %3, %1, %5"          \
               : "+v"(f[(i) & 3]), "+v"(u[(i) & 3]), "+v"(f[4 + ((i) & 3)]), "+v"(u[4 + ((i) & 3)]) : "v"(c), "s"(0x07060302u));

template <int MF, int NM, int NV, int MODE, int W>
__global__ __launch_bounds__(256 * W) void k(float *out, int iters) {
  extern __shared__ float lds[]; // (sized by the host so that ONE workgroup fits a CU)
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
  const float af = threadIdx.x * 0.001f, bfv = 1.0f + (threadIdx.x & 3);
  f16v acc[4] = {{0}, {0}, {0}, {0}};
  float f[8]; uint32_t u[8]; const float c = 1.0001f;
  for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
  const int wave = threadIdx.x >> 6, second = wave >= 4; // waves 0..3 and 4..7 sit pairwise on the four SIMDs
  auto mfma = [&](int m) {
    if (MF == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfv, acc[m & 3], 0, 0, 0);
    else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
  };
  constexpr int VQ = NV / 4; // groups of four vector instructions per step
  for (int it = 0; it < iters; ++it) {
    if (MODE == MODE_SAME) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        mfma(m);
#pragma unroll
        for (int q = m * VQ / NM; q < (m + 1) * VQ / NM; ++q) { V4(q) }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == MODE_SPLIT) {
      if (W == 1 || !second) {
#pragma unroll
        for (int m = 0; m < NM * W; ++m) { mfma(m); __builtin_amdgcn_sched_barrier(0); }
      }
      if (W == 1 || second) {
#pragma unroll
        for (int q = 0; q < VQ * W; ++q) { V4(q) }
      }
    } else {
      const bool mfma_phase = (((it >> 3) & 1) != 0) == (second != 0);
      if (mfma_phase) {
#pragma unroll
        for (int m = 0; m < 2 * NM; ++m) { mfma(m); __builtin_amdgcn_sched_barrier(0); }
      } else {
#pragma unroll
        for (int q = 0; q < 2 * VQ; ++q) { V4(q) }
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i];
  out[blockIdx.x * 256 * W + threadIdx.x] = s + lds[threadIdx.x & 7];
}

static double g_clock_ghz = 2.4;

template <int MF, int NM, int NV, int MODE, int W>
void run(float *out) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = k<MF, NM, NV, MODE, W>;
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 100 * 1024, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 100 * 1024, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const char *mode = MODE == MODE_SAME ? "same  " : MODE == MODE_SPLIT ? "split " : "phased";
  const double ns = ms * 1e6 / iters; // per step of every wave (W waves per SIMD advance one step each)
  const double pipe = (MF == 0 ? 64.0 : 32.0) * NM * W, issue = (4.0 * NV + 8.0 * NM) * W;
  printf("{\"mfma\": \"%s\", \"mfma_per_step\": %d, \"valu_per_step\": %d, \"mode\": \"%s\", \"waves_per_simd\": %d, \"ns_per_step\": %.1f, "
         "\"cycles_per_step_at_%.1f_ghz\": %.0f, \"pipe_cycles\": %.0f, \"issue_cycles\": %.0f, \"over_max\": %.2f, \"over_sum\": %.2f},\n",
         MF == 0 ? "f32_32x32x2" : "bf16_32x32x16", NM, NV, mode, W, ns, g_clock_ghz, ns * g_clock_ghz, pipe, issue,
         ns * g_clock_ghz / (pipe > issue ? pipe : issue), ns * g_clock_ghz / (pipe + issue));
}

template <int MF, int NM, int NV>
void family(float *out) {
  run<MF, NM, NV, MODE_SAME, 1>(out);
  run<MF, NM, NV, MODE_SPLIT, 1>(out); // = the serial sum: all MFMAs, then all vector work
  run<MF, NM, NV, MODE_SAME, 2>(out);
  run<MF, NM, NV, MODE_SPLIT, 2>(out);
  run<MF, NM, NV, MODE_PHASED, 2>(out);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * 4);
  printf("[\n");
  family<0, 4, 48>(out);  // the party first layer's step: 4 transposition MFMAs, ~47 vector / LDS instructions
  family<0, 4, 16>(out);
  family<0, 4, 96>(out);
  family<1, 12, 44>(out); // the second layer's k-step at two output blocks: 12 bf16 MFMAs, 44 split instructions
  family<1, 12, 96>(out);
  family<1, 6, 44>(out);
  printf("{}]\n");
  return 0;
}
