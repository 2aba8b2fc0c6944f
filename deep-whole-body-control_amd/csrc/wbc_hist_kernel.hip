// wbc_hist_kernel.hip -- forward of the StateHistoryEncoder (inference) for gfx950.
//
// Replaces, for the regulariser target of PPO.update (reference rsl_rl/algorithms/ppo.py:174-176: the history latent
// of every stored row, no gradient) the chain of rsl_rl/modules/actor_critic.py:39-84 for tsteps = 10:
//   per time step Linear(76 -> 30) + ELU; Conv1d(30 -> 20, k=4, s=2) + ELU; Conv1d(20 -> 10, k=2, s=1) + ELU;
//   channel-major flatten (30); Linear(30 -> 20) + ELU.
// 68 kFLOP per row: plain fp32 FMAs with all weights in LDS (23 KB) are enough -- the eager path spent its time in
// unfold copies, five small GEMM launches per 32768-row chunk and element-wise kernels, not in arithmetic.
// One workgroup = 24 rows: phase A one thread per (row, time step), then the two convolutions and the output layer
// with one thread per output element; activations stay in LDS. Measured bound: LDS bandwidth (every FMA of phase A takes
// its weight from LDS: 608 broadcast 16-byte reads per thread; conv1 another ~1100) -- ~80 k cycles per 24 rows; neither a
// persistent loop that prefetches the next rows' inputs (448.6 vs 450.8 us for 163840 rows) nor packed FMAs moved it.
// The next step for this kernel is MFMA for phase A and conv1 (operands in registers), not more of the same.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define H_T 10
#define H_NP 76
#define H_OBS 860
#define H_OFF 100           // obs[:, 100:860] = history [10][76], oldest first
#define H_C1 30
#define H_C2 20
#define H_C3 10
#define H_OUT 20
#define H_ROWS 24
#define H_THREADS 256

struct HistParams { const float *enc_w, *enc_b, *c1_w, *c1_b, *c2_w, *c2_b, *lin_w, *lin_b; };

static __device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

extern "C" __global__ void __launch_bounds__(H_THREADS) wbc_hist_latent_kernel(HistParams P, const float* __restrict__ obs, float* __restrict__ out,
                                                                              int rows) {
  __shared__ float w_enc[H_NP][H_C1 + 2];        // transposed [k][j]: thread reads 30 consecutive floats (broadcast)
  __shared__ float b_enc[H_C1];
  // row strides padded (120 -> 124, 40 -> 42 floats): lanes of a wave read the same column of different output channels'
  // rows; with the natural strides channels co and co + 8 share their banks
  __shared__ __attribute__((aligned(16))) float w_c1[H_C2][H_C1 * 4 + 4];
  __shared__ float b_c1[H_C2];
  __shared__ __attribute__((aligned(8))) float w_c2[H_C3][H_C2 * 2 + 2];
  __shared__ float b_c2[H_C3];
  __shared__ float w_lin[H_OUT][H_C1], b_lin[H_OUT];
  __shared__ float h1[H_ROWS][H_T][H_C1 + 1];
  __shared__ float h2[H_ROWS][4][H_C2 + 1];
  __shared__ float h3[H_ROWS][H_C1 + 1];
  const int tid = threadIdx.x, row0 = blockIdx.x * H_ROWS;
  for (int e = tid; e < H_C1 * H_NP; e += H_THREADS) { const int j = e / H_NP, k = e - j * H_NP; w_enc[k][j] = P.enc_w[e]; }
  for (int e = tid; e < H_C2 * H_C1 * 4; e += H_THREADS) w_c1[e / (H_C1 * 4)][e % (H_C1 * 4)] = P.c1_w[e];      // [co][ci][k] as stored
  for (int e = tid; e < H_C3 * H_C2 * 2; e += H_THREADS) w_c2[e / (H_C2 * 2)][e % (H_C2 * 2)] = P.c2_w[e];
  for (int e = tid; e < H_OUT * H_C1; e += H_THREADS) (&w_lin[0][0])[e] = P.lin_w[e];
  if (tid < H_C1) b_enc[tid] = P.enc_b[tid];
  if (tid < H_C2) b_c1[tid] = P.c1_b[tid];
  if (tid < H_C3) b_c2[tid] = P.c2_b[tid];
  if (tid < H_OUT) b_lin[tid] = P.lin_b[tid];
  // phase A inputs: thread (r, t) reads its 76 floats (19 float4: rows are 860 floats = 16-byte aligned, 100 and 76 too)
  const int r = tid / H_T, t = tid - r * H_T;
  const bool haveA = tid < H_ROWS * H_T && row0 + r < rows;
  float4 x4[H_NP / 4];
  if (haveA) {
    const float4* src = reinterpret_cast<const float4*>(obs + (size_t)(row0 + r) * H_OBS + H_OFF + t * H_NP);
#pragma unroll
    for (int q = 0; q < H_NP / 4; ++q) x4[q] = src[q];
  }
  __syncthreads();
  if (haveA) {
    float acc[H_C1];
#pragma unroll
    for (int j = 0; j < H_C1; ++j) acc[j] = b_enc[j];
#pragma unroll
    for (int q = 0; q < H_NP / 4; ++q) {
      const float xs[4] = {x4[q].x, x4[q].y, x4[q].z, x4[q].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wr = w_enc[4 * q + u];
#pragma unroll
        for (int j = 0; j < H_C1; ++j) acc[j] += xs[u] * wr[j];
      }
    }
#pragma unroll
    for (int j = 0; j < H_C1; ++j) h1[r][t][j] = elu1(acc[j]);
  }
  __syncthreads();
  // conv1: out[r][l][co] = b + sum_{ci,k} h1[r][2l+k][ci] * W1[co][ci][k]
  for (int e = tid; e < H_ROWS * 4 * H_C2; e += H_THREADS) {
    const int rr = e / (4 * H_C2), rem = e - rr * (4 * H_C2), l = rem / H_C2, co = rem - l * H_C2;
    float acc = b_c1[co];
    const float* wr = w_c1[co];
#pragma unroll 6
    for (int ci = 0; ci < H_C1; ++ci)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += h1[rr][2 * l + k][ci] * wr[ci * 4 + k];
    h2[rr][l][co] = elu1(acc);
  }
  __syncthreads();
  // conv2 + channel-major flatten: h3[r][co*3 + l]
  for (int e = tid; e < H_ROWS * 3 * H_C3; e += H_THREADS) {
    const int rr = e / (3 * H_C3), rem = e - rr * (3 * H_C3), l = rem / H_C3, co = rem - l * H_C3;
    float acc = b_c2[co];
    const float* wr = w_c2[co];
#pragma unroll 4
    for (int ci = 0; ci < H_C2; ++ci) acc += h2[rr][l][ci] * wr[ci * 2] + h2[rr][l + 1][ci] * wr[ci * 2 + 1];
    h3[rr][co * 3 + l] = elu1(acc);
  }
  __syncthreads();
  for (int e = tid; e < H_ROWS * H_OUT; e += H_THREADS) {
    const int rr = e / H_OUT, j = e - rr * H_OUT;
    if (row0 + rr < rows) {
      float acc = b_lin[j];
#pragma unroll 6
      for (int i = 0; i < H_C1; ++i) acc += h3[rr][i] * w_lin[j][i];
      out[(size_t)(row0 + rr) * H_OUT + j] = elu1(acc);
    }
  }
}

// C-ABI. params: 8 device pointers (encoder.0.weight [30,76], .bias, conv_layers.0.weight [20,30,4], .bias,
// conv_layers.2.weight [10,20,2], .bias, linear_output.0.weight [20,30], .bias); obs f32 [rows, 860]; out f32 [rows, 20].
extern "C" int wbc_hist_latent(const void* const* params, const float* obs, float* out, int rows, void* stream) {
  if (!params || !obs || !out || rows <= 0) return -1;
  HistParams P;
  const float** dst = reinterpret_cast<const float**>(&P);
  for (int i = 0; i < 8; ++i) { if (!params[i]) return -1; dst[i] = static_cast<const float*>(params[i]); }
  hipLaunchKernelGGL(wbc_hist_latent_kernel, dim3((rows + H_ROWS - 1) / H_ROWS), dim3(H_THREADS), 0, (hipStream_t)stream, P, obs, out, rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
