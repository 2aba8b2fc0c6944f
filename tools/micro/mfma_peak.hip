// Sustained fp32 MFMA rate of this GPU: every wave issues back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpb : {1, 2}) {      // blocks per CU multiplier: 256*wpb*... grid sizes
    for (int rep = 0; rep < 3; ++rep) {
      const int blocks = 256 * 2 * wpb, iters = 20000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 2.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 32 * 32 * 2 * 2;
      printf("blocks %d: %.3f ms  %.1f TFLOP/s fp32 (32x32x2)\n", blocks, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
