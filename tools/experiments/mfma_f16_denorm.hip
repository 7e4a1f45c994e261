// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal INPUTS, and does v_cvt_f16_f32 produce them?  (round 5: the main net as fp16 pairs.)
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_f16_denorm.hip -o gpurun_out/mfma_f16_denorm && gpurun_out/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *in, float *out, float bval) {
  // A: 32 x 16 (row r = lane & 31, k = 8 (lane >> 5) + j); B: 16 x 32 (col = lane & 31).  A[r][k] = in[0] for k = 0 only; B[k][c] = bval for k = 0.
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
  if (lane < 32) { a[0] = (_Float16)in[0]; b[0] = (_Float16)bval; }
  f32x16 c;
  for (int q = 0; q < 16; ++q) c[q] = 0.0f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float *din, *dout;
  hipMalloc(&din, 4); hipMalloc(&dout, 8);
  const float vals[] = {1.0f, 6.2e-5f /* just normal */, 3.0e-5f /* subnormal */, 1.0e-6f, 6.0e-8f /* smallest subnormal */, 2.0e-8f};
  for (float v : vals)
    for (float bval : {1.0f, 1024.0f}) {
      hipMemcpy(din, &v, 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, bval);
      float o[2]; hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost);
      printf("a = %.6e  (as f16 -> %.6e)  x b = %g  -> mfma %.6e   %s\n", v, o[1], bval, o[0], (o[1] != 0.0f && o[0] == 0.0f) ? "FLUSHED" : "ok");
    }
  return 0;
}
