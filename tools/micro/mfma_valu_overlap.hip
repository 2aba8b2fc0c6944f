// Does vector-ALU work hide in the shadow of MFMAs? A wave runs two dependent chains of v_mfma_f32_16x16x4_f32 (8 passes each)
// with K independent v_fma_f32 per MFMA (a) in the same wave, (b) in sibling waves on the same SIMD that run only FMAs.
// Prints cycles per MFMA for 1 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int K>
__global__ void __launch_bounds__(256) same_wave(float* out, int iters, float a, float b, long long* cyc) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(k + 4) & 7]) : "v"(a), "v"(b));
    }
  }
  const long long t1 = clock64();
  float s = c0[0] + c1[1];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
// Workgroups alternate between MFMA-only and FMA-only by dispatch round ((blockIdx.x >> 3) >> 5: workgroup i goes to XCD i % 8 and,
// inside the XCD, round-robin to its 32 CUs), so that both kinds share every CU / SIMD.
template <int K>
__global__ void __launch_bounds__(256) other_wave(float* out, int iters, float a, float b, long long* cyc) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  const long long t0 = clock64();
  const int role = (blockIdx.x >> 8) & 1;
  if (role == 0) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      }
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
    }
  }
  const long long t1 = clock64();
  float s = c0[0] + c1[1];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((blockIdx.x == 0 || blockIdx.x == 256) && threadIdx.x == 0) cyc[1 + role] = t1 - t0;
}
template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();                                    // warm-up
  (void)hipEventRecord(e0);
  launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
// Wall-clock per configuration. same: every wave does MFMAs with K FMAs each. split: half the waves of every SIMD do only the
// MFMAs, the other half only the FMAs (same totals per SIMD as `same` with twice the waves). mfma / fma: the two halves alone.
template <int K>
__global__ void __launch_bounds__(256) fma_only(float* out, int iters, float a, float b) {
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K>
void run(float* d, long long* c) {
  const int iters = 4000;
  for (int bpc : {1, 3}) {
    const float same = time_ms([&] { hipLaunchKernelGGL(same_wave<K>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0f, 0.5f, c); });
    const float split = time_ms([&] { hipLaunchKernelGGL(other_wave<K>, dim3(256 * bpc * 2), dim3(256), 0, 0, d, iters, 1.0f, 0.5f, c); });
    const float mfma = time_ms([&] { hipLaunchKernelGGL(same_wave<0>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0f, 0.5f, c); });
    const float fma = time_ms([&] { hipLaunchKernelGGL(fma_only<K>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0f, 0.5f); });
    printf("K=%d FMAs per MFMA, %d waves/SIMD: MFMA only %.3f ms, FMA only %.3f ms, same wave %.3f ms, separate waves (%d+%d per SIMD) %.3f ms\n",
           K, bpc, mfma, fma, same, bpc, bpc, split);
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 4 * 256 * 8192);
  long long* c; (void)hipMalloc(&c, 64);
  run<2>(d, c); run<4>(d, c); run<8>(d, c);
  return 0;
}
