// wbc_hist_kernel.hip -- forward of the StateHistoryEncoder (inference) for gfx950.
//
// Replaces, for the regulariser target of PPO.update (reference rsl_rl/algorithms/ppo.py:174-176: the history latent
// of every stored row, no gradient) the chain of rsl_rl/modules/actor_critic.py:39-84 for tsteps = 10:
//   per time step Linear(76 -> 30) + ELU; Conv1d(30 -> 20, k=4, s=2) + ELU; Conv1d(20 -> 10, k=2, s=1) + ELU;
//   channel-major flatten (30); Linear(30 -> 20) + ELU.
// 68 kFLOP per row; the eager path spent its time in unfold copies, five small GEMM launches per 32768-row chunk and
// element-wise kernels. The first version of this kernel (plain FMAs, all weights in LDS) was LDS-bound: every FMA of the
// two big layers took its weight from LDS (~1700 16-byte reads per thread and 24 rows).
// Persistent workgroups walk over groups of 24 rows; the weights of the two big layers stay in registers, activations in
// LDS (45 KB). 450 -> 230 us for 163840 rows.
#include <hip/hip_runtime.h>
#include "wbc_stream_guard.h"
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

typedef float f32x4h __attribute__((ext_vector_type(4)));

// The two GEMM-shaped layers on v_mfma_f32_16x16x4_f32 (A: lane L supplies A[row = L & 15][k-slot g = L >> 4]; B: B[g][col = L & 15];
// D[row = 4 (L >> 4) + r][col = L & 15]), weights in registers, no weight traffic through LDS:
//   per-step projection: rows = (row, time step) pairs, 15 blocks of 16 per workgroup. A pair's 76 inputs are 19 aligned
//     float4 chunks; slot g takes the chunks g, g + 4, g + 8, g + 12, g + 16 (k-step ks of slot g = element ks % 4 of chunk
//     g + 4 (ks / 4); 20 k-steps, slot 3's fifth chunk does not exist: zero weights), so a lane reads its A operands straight
//     from obs with five 16-byte loads; columns = the 30 outputs in two halves.
//   conv1 (30 -> 20, k = 4, s = 2): rows = (row, output position) pairs, 6 blocks; slot g = kernel tap, k-step = input
//     channel: a lane's 30 A operands are one time row of h1 in LDS; columns = the 20 outputs.
// conv2 and the output layer (1.5 % of the multiply-adds) stay plain FMAs over LDS.
static __device__ __forceinline__ void hist_latent_body(const HistParams& P, const float* __restrict__ obs, float* __restrict__ out, int rows) {
  __shared__ float b_enc[H_C1], b_c1[H_C2], b_c2[H_C3];
  __shared__ __attribute__((aligned(8))) float w_c2[H_C3][H_C2 * 2 + 2];       // row stride padded: channels co, co + 8 would share banks
  __shared__ float w_lin[H_OUT][H_C1], b_lin[H_OUT];
  __shared__ float h1[H_ROWS][H_T][H_C1 + 1];
  __shared__ float h2[H_ROWS][4][H_C2 + 1];
  __shared__ float h3[H_ROWS][H_C1 + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
  for (int e = tid; e < H_C3 * H_C2 * 2; e += H_THREADS) w_c2[e / (H_C2 * 2)][e % (H_C2 * 2)] = P.c2_w[e];
  for (int e = tid; e < H_OUT * H_C1; e += H_THREADS) (&w_lin[0][0])[e] = P.lin_w[e];
  if (tid < H_C1) b_enc[tid] = P.enc_b[tid];
  if (tid < H_C2) b_c1[tid] = P.c1_b[tid];
  if (tid < H_C3) b_c2[tid] = P.c2_b[tid];
  if (tid < H_OUT) b_lin[tid] = P.lin_b[tid];
  // Register-resident weights of the two MFMA layers, loaded once: a workgroup walks over groups of 24 rows (grid-stride).
  float wA[40];                                       // [ks][half]: enc_w[col = p + 16 half][k(ks, g)] (0 past column 30 / chunk 19)
#pragma unroll
  for (int ks = 0; ks < 20; ++ks) {
    const int c = g + 4 * (ks >> 2), k = 4 * c + (ks & 3);
    wA[2 * ks] = (c < 19) ? P.enc_w[p * H_NP + k] : 0.f;
    wA[2 * ks + 1] = (c < 19 && p + 16 < H_C1) ? P.enc_w[(p + 16) * H_NP + k] : 0.f;
  }
  float wB[60];                                       // [ci][half]: c1_w[co = p + 16 half][ci][tap g] (0 past output 20)
#pragma unroll
  for (int ci = 0; ci < H_C1; ++ci) {
    wB[2 * ci] = P.c1_w[p * (H_C1 * 4) + ci * 4 + g];
    wB[2 * ci + 1] = (p + 16 < H_C2) ? P.c1_w[(p + 16) * (H_C1 * 4) + ci * 4 + g] : 0.f;
  }
  const int ngroups = (rows + H_ROWS - 1) / H_ROWS;
  auto fetch = [&](float4 (&a)[5], int row0, int mb) {
    const int q = 16 * mb + p, r = q / H_T, t = q - r * H_T;
    const float4* src = reinterpret_cast<const float4*>(obs + (size_t)min(row0 + r, rows - 1) * H_OBS + H_OFF + t * H_NP);
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] = src[min(g + 4 * j, 18)];           // (slot 3, j = 4: any finite chunk, its weights are zero)
  };
  float4 a0[5], a1[5];
  if ((int)blockIdx.x < ngroups) fetch(a0, blockIdx.x * H_ROWS, wave);    // wave w: blocks w, w + 4, w + 8, w + 12 (< 15)
  __syncthreads();                                    // biases / small weights are in LDS
#pragma unroll 1
  for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
    const int row0 = group * H_ROWS;
    // ---- per-step projection 76 -> 30 + ELU -> h1
#pragma unroll 1
    for (int mb = wave; mb < (H_ROWS * H_T) / 16; mb += 4) {
      if (mb + 4 < (H_ROWS * H_T) / 16) fetch(a1, row0, mb + 4);
      else if (group + (int)gridDim.x < ngroups) fetch(a1, row0 + (int)gridDim.x * H_ROWS, wave);    // the next group's first block
      f32x4h acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float av[4] = {a0[j].x, a0[j].y, a0[j].z, a0[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wA[2 * (4 * j + e)], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wA[2 * (4 * j + e) + 1], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int q = 16 * mb + 4 * g + r4, rr = q / H_T, tt = q - rr * H_T;
        h1[rr][tt][p] = elu1(acc0[r4] + b_enc[p]);
        if (p + 16 < H_C1) h1[rr][tt][p + 16] = elu1(acc1[r4] + b_enc[p + 16]);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) a0[j] = a1[j];
    }
    __syncthreads();
    // ---- conv1: h2[r][l][co] = ELU(b + sum_{ci,k} h1[r][2l+k][ci] * W1[co][ci][k])
#pragma unroll 1
    for (int mb = wave; mb < (H_ROWS * 4) / 16; mb += 4) {       // 6 blocks of 16 (row, position) pairs
      const int q = 16 * mb + p, rr = q >> 2, l = q & 3;
      const float* ap = &h1[rr][2 * l + g][0];
      f32x4h acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ci = 0; ci < H_C1; ++ci) {
        const float av = ap[ci];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wB[2 * ci], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wB[2 * ci + 1], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int q2 = 16 * mb + 4 * g + r4, r2 = q2 >> 2, l2 = q2 & 3;
        h2[r2][l2][p] = elu1(acc0[r4] + b_c1[p]);
        if (p + 16 < H_C2) h2[r2][l2][p + 16] = elu1(acc1[r4] + b_c1[p + 16]);
      }
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
    // (h1 is rewritten only after the barrier that followed conv1's reads, h2 after the one that followed conv2's, h3 after
    // the next group's two barriers: no extra barrier needed here)
  }
}

// Two register budgets: three workgroups per CU (168 VGPRs, a few spills) win for long inputs (233 vs 253 us at 163840 rows),
// two (184 VGPRs, none) for one env step's worth of rows (21.6 vs 23.9 us at 4096).
extern "C" __global__ void __launch_bounds__(H_THREADS, 3) wbc_hist_latent_kernel(HistParams P, const float* __restrict__ obs, float* __restrict__ out,
                                                                              int rows) {
  hist_latent_body(P, obs, out, rows);
}
extern "C" __global__ void __launch_bounds__(H_THREADS, 2) wbc_hist_latent_small_kernel(HistParams P, const float* __restrict__ obs,
                                                                                    float* __restrict__ out, int rows) {
  hist_latent_body(P, obs, out, rows);
}

// C-ABI. params: 8 device pointers (encoder.0.weight [30,76], .bias, conv_layers.0.weight [20,30,4], .bias,
// conv_layers.2.weight [10,20,2], .bias, linear_output.0.weight [20,30], .bias); obs f32 [rows, 860]; out f32 [rows, 20].
extern "C" int wbc_hist_latent(const void* const* params, const float* obs, float* out, int rows, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!params || !obs || !out || rows <= 0) return -1;
  HistParams P;
  const float** dst = reinterpret_cast<const float**>(&P);
  for (int i = 0; i < 8; ++i) { if (!params[i]) return -1; dst[i] = static_cast<const float*>(params[i]); }
  const int ngroups = (rows + H_ROWS - 1) / H_ROWS;
  if (ngroups <= 512) hipLaunchKernelGGL(wbc_hist_latent_small_kernel, dim3(ngroups), dim3(H_THREADS), 0, (hipStream_t)stream, P, obs, out, rows);
  else hipLaunchKernelGGL(wbc_hist_latent_kernel, dim3(768), dim3(H_THREADS), 0, (hipStream_t)stream, P, obs, out, rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
