// wbc_hist_train_kernel.hip -- one minibatch of PPO.update_dagger for gfx950: forward + loss + backward + weight gradients
// of the StateHistoryEncoder in one launch, then a fixed-order reduction and clip + Adam.
//
// Replaces (reference rsl_rl/algorithms/ppo.py:265-291, rsl_rl/modules/actor_critic.py:39-84), per minibatch:
//   priv = sg(priv_encoder(obs[:, 76:100]));  hist = history_encoder(obs[:, 100:860] as [10][76])
//   loss = mean_rows ||priv - hist||_2;  hist_encoder_optimizer.zero_grad(); loss.backward(); clip_grad_norm_; Adam.step()
// with the network of wbc_hist_kernel.hip: Linear(76 -> 30)+ELU per time step; Conv1d(30 -> 20, k4, s2)+ELU;
// Conv1d(20 -> 10, k2, s1)+ELU; channel-major flatten (30); Linear(30 -> 20)+ELU. 160 kFLOP per row (forward 34 k MAC,
// input gradients 11 k MAC, weight gradients 34 k MAC).
//
// Mapping: persistent workgroups of 256 lanes walk over groups of 24 gathered rows. Forward as in the inference kernel
// (the two GEMM-shaped layers on v_mfma_f32_16x16x4_f32 with register-resident weights, A operands straight from obs).
// Backward layer by layer in LDS, every activation buffer overwritten in place by its pre-activation gradient
// (ELU'(z) = h > 0 ? 1 : h + 1 needs only the output). Weight gradients never leave registers inside a launch: every thread
// owns a fixed set of entries of the small layers (plain FMAs over the 24 rows of a group), the 30 x 76 projection gradient
// is an MFMA GEMM over the 240 (row, time step) pairs of a group with B operands re-read from obs (L2). At the end each
// workgroup writes ONE partial gradient vector; hist_reduce_kernel sums them in workgroup order (deterministic).
#include <hip/hip_runtime.h>
#include "wbc_stream_guard.h"
#include <stdint.h>

#define H_T 10
#define H_NP 76
#define H_OBS 860
#define H_OFF 100
#define H_C1 30
#define H_C2 20
#define H_C3 10
#define H_OUT 20
#define G_ROWS 24
#define G_THREADS 256
#define G_PAIRS (G_ROWS * H_T)          // 240 (row, time step) pairs = 15 MFMA row blocks
// flat gradient layout = history_encoder.parameters() order
#define O_ENC_W 0
#define O_ENC_B (O_ENC_W + H_C1 * H_NP)            // 2280
#define O_C1_W (O_ENC_B + H_C1)                    // 2310
#define O_C1_B (O_C1_W + H_C2 * H_C1 * 4)          // 4710
#define O_C2_W (O_C1_B + H_C2)                     // 4730
#define O_C2_B (O_C2_W + H_C3 * H_C2 * 2)          // 5130
#define O_LIN_W (O_C2_B + H_C3)                    // 5140
#define O_LIN_B (O_LIN_W + H_OUT * H_C1)           // 5740
#define N_GRAD (O_LIN_B + H_OUT)                   // 5760
#define N_PART (N_GRAD + 1)                        // + loss sum
#define PRIV_OFF 76
#define PRIV_N 24
#define PRIV_H 64

struct HistParams { const float *enc_w, *enc_b, *c1_w, *c1_b, *c2_w, *c2_b, *lin_w, *lin_b; };

static __device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
static __device__ __forceinline__ float delu_from_out(float h) { return h > 0.f ? 1.f : h + 1.f; }

typedef float f32x4h __attribute__((ext_vector_type(4)));

struct TrainSmem {
  float h1[G_PAIRS][H_C1 + 1];          // [q = row*10 + t][c]; becomes dz1
  float h2[G_ROWS][4][H_C2 + 1];
  float dz2[G_ROWS][4][H_C2 + 1];       // conv1 pre-activation gradient (its own buffer: no register staging of 8 values per lane)
  float h3[G_ROWS][H_C1 + 1];           // [co*3 + l]; becomes dz3
  float y[G_ROWS][H_OUT];               // becomes dz4
  float w1[H_C2 * 4][32];               // conv1 weights as [co][tap k][ci] (ci contiguous, rows padded to 32: bank = ci, conflict-free)
  __attribute__((aligned(8))) float w_c2[H_C3][H_C2 * 2 + 2];
  float w_lin[H_OUT][H_C1];
  float b_enc[H_C1], b_c1[H_C2], b_c2[H_C3], b_lin[H_OUT];
  long long ridx[G_ROWS];               // gathered row of obs / target, -1 past the end of the minibatch
};

extern "C" __global__ void __launch_bounds__(G_THREADS, 2)
hist_train_kernel(HistParams P, const float* __restrict__ obs, const float* __restrict__ target, const long long* __restrict__ idx,
                  int rows, float* __restrict__ partial) {
  __shared__ TrainSmem s;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, p = lane & 15, g = lane >> 4;
  for (int e = tid; e < H_C3 * H_C2 * 2; e += G_THREADS) s.w_c2[e / (H_C2 * 2)][e % (H_C2 * 2)] = P.c2_w[e];
  for (int e = tid; e < H_OUT * H_C1; e += G_THREADS) (&s.w_lin[0][0])[e] = P.lin_w[e];
  for (int e = tid; e < H_C2 * H_C1 * 4; e += G_THREADS) { const int co = e / (H_C1 * 4), rem = e - co * (H_C1 * 4); s.w1[co * 4 + (rem & 3)][rem >> 2] = P.c1_w[e]; }
  if (tid < H_C1) s.b_enc[tid] = P.enc_b[tid];
  if (tid < H_C2) s.b_c1[tid] = P.c1_b[tid];
  if (tid < H_C3) s.b_c2[tid] = P.c2_b[tid];
  if (tid < H_OUT) s.b_lin[tid] = P.lin_b[tid];
  for (int e = tid; e < G_PAIRS; e += G_THREADS) s.h1[e][H_C1] = 0.f;       // pad column: read (masked) by the MFMA A operand
  // The forward weights of the two MFMA layers (layouts: wbc_hist_kernel.hip) are re-fetched per group (projection: L2,
  // conv1: LDS): held across the backward they would push the gradient accumulators into scratch.
  // ---- gradient accumulators (registers, whole launch) ----
  f32x4h accE[3];                       // projection weight tiles of this wave: (ct, jt) = (0, wave), (1, wave), and for wave < 2 (wave, 4)
#pragma unroll
  for (int i = 0; i < 3; ++i) accE[i] = (f32x4h){0.f, 0.f, 0.f, 0.f};
  float accL[3] = {0.f, 0.f, 0.f};      // linear_output weight entries tid + 256 i (< 600)
  float accC2[2] = {0.f, 0.f};          // conv2 weight entries tid + 256 i (< 400)
  float accC1[10];                      // conv1 weight entries tid + 256 i (< 2400)
#pragma unroll
  for (int i = 0; i < 10; ++i) accC1[i] = 0.f;
  float accB = 0.f;                     // bias entry tid (< 80): enc 0..29, conv1 30..49, conv2 50..59, linear 60..79
  float loss_acc = 0.f;
  const float inv_rows = 1.f / (float)rows;

  const int ngroups = (rows + G_ROWS - 1) / G_ROWS;
  auto fetch = [&](float4 (&a)[5], int row0, int mb) {
    const int q = 16 * mb + p, r = q / H_T, t = q - r * H_T;
    const long long gi = idx[min(row0 + r, rows - 1)];
    const float4* src = reinterpret_cast<const float4*>(obs + (size_t)gi * H_OBS + H_OFF + t * H_NP);
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] = src[min(g + 4 * j, 18)];
  };
  __syncthreads();
#pragma unroll 1
  for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
    const int row0 = group * G_ROWS;
    if (tid < G_ROWS) s.ridx[tid] = (row0 + tid < rows) ? idx[row0 + tid] : -1;
    int opaque = 0;
    asm volatile("" : "+v"(opaque));                    // keeps the weight fetches inside the loop (no hoisting)
    // ---- forward: per-step projection 76 -> 30 + ELU -> h1
    float4 a0[5], a1[5];
    fetch(a0, row0, wave);
    float wA[40];
    {
      const float* ew = P.enc_w + opaque;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const int c = g + 4 * (ks >> 2), k = 4 * c + (ks & 3);
        wA[2 * ks] = (c < 19) ? ew[p * H_NP + k] : 0.f;
        wA[2 * ks + 1] = (c < 19 && p + 16 < H_C1) ? ew[(p + 16) * H_NP + k] : 0.f;
      }
    }
#pragma unroll 1
    for (int mb = wave; mb < G_PAIRS / 16; mb += 4) {
      if (mb + 4 < G_PAIRS / 16) fetch(a1, row0, mb + 4);
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
        const int q = 16 * mb + 4 * g + r4;
        s.h1[q][p] = elu1(acc0[r4] + s.b_enc[p]);
        if (p + 16 < H_C1) s.h1[q][p + 16] = elu1(acc1[r4] + s.b_enc[p + 16]);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) a0[j] = a1[j];
    }
    __syncthreads();
    // ---- conv1 (30 -> 20, k4 s2) + ELU -> h2
    float wB[60];
    {
      const float* w1p = &s.w1[0][0] + opaque;
#pragma unroll
      for (int ci = 0; ci < H_C1; ++ci) {
        wB[2 * ci] = w1p[(p * 4 + g) * 32 + ci];
        wB[2 * ci + 1] = (p + 16 < H_C2) ? w1p[((p + 16) * 4 + g) * 32 + ci] : 0.f;
      }
    }
#pragma unroll 1
    for (int mb = wave; mb < (G_ROWS * 4) / 16; mb += 4) {
      const int q = 16 * mb + p, rr = q >> 2, l = q & 3;
      const float* ap = &s.h1[rr * H_T + 2 * l + g][0];
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
        s.h2[r2][l2][p] = elu1(acc0[r4] + s.b_c1[p]);
        if (p + 16 < H_C2) s.h2[r2][l2][p + 16] = elu1(acc1[r4] + s.b_c1[p + 16]);
      }
    }
    __syncthreads();
    // ---- conv2 (20 -> 10, k2 s1) + ELU, channel-major flatten -> h3
    for (int e = tid; e < G_ROWS * 3 * H_C3; e += G_THREADS) {
      const int rr = e / (3 * H_C3), rem = e - rr * (3 * H_C3), l = rem / H_C3, co = rem - l * H_C3;
      float acc = s.b_c2[co];
      const float* wr = s.w_c2[co];
#pragma unroll
      for (int ci = 0; ci < H_C2; ++ci) acc += s.h2[rr][l][ci] * wr[ci * 2] + s.h2[rr][l + 1][ci] * wr[ci * 2 + 1];
      s.h3[rr][co * 3 + l] = elu1(acc);
    }
    __syncthreads();
    // ---- linear_output (30 -> 20) + ELU -> y
    for (int e = tid; e < G_ROWS * H_OUT; e += G_THREADS) {
      const int rr = e / H_OUT, j = e - rr * H_OUT;
      float acc = s.b_lin[j];
#pragma unroll 15
      for (int i = 0; i < H_C1; ++i) acc += s.h3[rr][i] * s.w_lin[j][i];
      s.y[rr][j] = elu1(acc);
    }
    __syncthreads();
    // ---- loss and dz4 = dL/dy * ELU'(y): one lane per row (||priv - hist||_2, mean over the minibatch's rows)
    if (tid < G_ROWS) {
      const long long gi = s.ridx[tid];
      if (gi >= 0) {
        const float* tg = target + (size_t)gi * H_OUT;
        float d[H_OUT], n2 = 0.f;
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) { d[j] = s.y[tid][j] - tg[j]; n2 += d[j] * d[j]; }
        const float nrm = sqrtf(n2);
        loss_acc += nrm;
        const float sc = nrm > 0.f ? inv_rows / nrm : 0.f;
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) s.y[tid][j] = d[j] * sc * delu_from_out(s.y[tid][j]);
      } else {
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) s.y[tid][j] = 0.f;
      }
    }
    __syncthreads();
    // ---- backward linear_output: weight / bias gradients, dz3 (held in registers until h3 has been read by everyone)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + G_THREADS * i;
      if (e < H_OUT * H_C1) {
        const int j = e / H_C1, i3 = e - j * H_C1;
        float a = 0.f;
#pragma unroll 12
        for (int rr = 0; rr < G_ROWS; ++rr) a += s.y[rr][j] * s.h3[rr][i3];
        accL[i] += a;
      }
    }
    if (tid >= 60 && tid < 80) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.y[rr][tid - 60];
      accB += a;
    }
    float dz3r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + G_THREADS * i;
      dz3r[i] = 0.f;
      if (e < G_ROWS * H_C1) {
        const int rr = e / H_C1, i3 = e - rr * H_C1;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) a += s.w_lin[j][i3] * s.y[rr][j];
        dz3r[i] = a * delu_from_out(s.h3[rr][i3]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + G_THREADS * i;
      if (e < G_ROWS * H_C1) s.h3[e / H_C1][e % H_C1] = dz3r[i];
    }
    __syncthreads();
    // ---- backward conv2: weight / bias gradients, dz2
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + G_THREADS * i;
      if (e < H_C3 * H_C2 * 2) {
        const int co = e / (H_C2 * 2), rem = e - co * (H_C2 * 2), ci = rem >> 1, k = rem & 1;
        float a = 0.f;
#pragma unroll 6
        for (int rr = 0; rr < G_ROWS; ++rr) {
#pragma unroll
          for (int l = 0; l < 3; ++l) a += s.h3[rr][co * 3 + l] * s.h2[rr][l + k][ci];
        }
        accC2[i] += a;
      }
    }
    if (tid >= 50 && tid < 60) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.h3[rr][(tid - 50) * 3] + s.h3[rr][(tid - 50) * 3 + 1] + s.h3[rr][(tid - 50) * 3 + 2];
      accB += a;
    }
    for (int e = tid; e < G_ROWS * 4 * H_C2; e += G_THREADS) {
      const int rr = e / (4 * H_C2), rem = e - rr * (4 * H_C2), lp = rem / H_C2, ci = rem - lp * H_C2;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int l = lp - k;
        if (l >= 0 && l < 3) {
#pragma unroll
          for (int co = 0; co < H_C3; ++co) a += s.w_c2[co][ci * 2 + k] * s.h3[rr][co * 3 + l];
        }
      }
      s.dz2[rr][lp][ci] = a * delu_from_out(s.h2[rr][lp][ci]);
    }
    __syncthreads();
    // ---- backward conv1: weight / bias gradients (h1 still holds the activations)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int e = tid + G_THREADS * i;
      if (e < H_C2 * H_C1 * 4) {                       // thread entries in (co, tap k, ci) order: consecutive lanes = consecutive ci (LDS banks)
        const int co = e / (H_C1 * 4), rem = e - co * (H_C1 * 4), k = rem / H_C1, ci = rem - k * H_C1;
        float a = 0.f;
#pragma unroll 4
        for (int rr = 0; rr < G_ROWS; ++rr) {
#pragma unroll
          for (int l = 0; l < 4; ++l) a += s.dz2[rr][l][co] * s.h1[rr * H_T + 2 * l + k][ci];
        }
        accC1[i] += a;
      }
    }
    if (tid >= 30 && tid < 50) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.dz2[rr][0][tid - 30] + s.dz2[rr][1][tid - 30] + s.dz2[rr][2][tid - 30] + s.dz2[rr][3][tid - 30];
      accB += a;
    }
    __syncthreads();
    // ---- dz1 = (conv1^T dz2) * ELU'(h1), in place (each element read and written by one thread)
    for (int e = tid; e < G_PAIRS * H_C1; e += G_THREADS) {
      const int q = e / H_C1, ci = e - q * H_C1, rr = q / H_T, t = q - rr * H_T;
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int k = t - 2 * l;
        if (k >= 0 && k < 4) {
#pragma unroll
          for (int co = 0; co < H_C2; ++co) a += s.w1[co * 4 + k][ci] * s.dz2[rr][l][co];
        }
      }
      s.h1[q][ci] = a * delu_from_out(s.h1[q][ci]);
    }
    __syncthreads();
    // ---- projection weight gradient dW[c][j] += sum_q dz1[q][c] x[q][j]: MFMA, A = dz1 (LDS), B = x (obs, L2)
    {
      const int jt0 = wave;                                      // this wave's column tile(s): jt0 for both ct, plus (ct = wave, jt = 4) for wave < 2
      const bool third = wave < 2;
      const bool a1ok = (p + 16) < H_C1;
      const bool b4ok = (64 + p) < H_NP;
#pragma unroll 2
      for (int ks = 0; ks < G_PAIRS / 4; ++ks) {
        const int q = 4 * ks + g, rr = q / H_T, t = q - rr * H_T;
        const long long gi = s.ridx[rr];
        const float* xr = obs + (size_t)(gi < 0 ? 0 : gi) * H_OBS + H_OFF + t * H_NP;
        const float bx = xr[jt0 * 16 + p];
        const float b4 = (third && b4ok) ? xr[64 + p] : 0.f;
        const float av0 = s.h1[q][p];
        const float av1 = a1ok ? s.h1[q][p + 16] : 0.f;
        accE[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bx, accE[0], 0, 0, 0);
        accE[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bx, accE[1], 0, 0, 0);
        if (third) accE[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wave == 0 ? av0 : av1, b4, accE[2], 0, 0, 0);
      }
    }
    if (tid < H_C1) {
      float a = 0.f;
      for (int q = 0; q < G_PAIRS; ++q) a += s.h1[q][tid];
      accB += a;
    }
    __syncthreads();                                   // h1 / ridx are rewritten by the next group
  }
  // ---- write this workgroup's partial gradient vector
  float* out = partial + (size_t)blockIdx.x * N_PART;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int c0 = 4 * g + r4, c1 = 16 + 4 * g + r4, j0 = wave * 16 + p;       // D[row = 4g + r][col = p]
    out[O_ENC_W + c0 * H_NP + j0] = accE[0][r4];
    if (c1 < H_C1) out[O_ENC_W + c1 * H_NP + j0] = accE[1][r4];
    if (wave < 2 && 64 + p < H_NP) {
      const int c = wave * 16 + 4 * g + r4;
      if (c < H_C1) out[O_ENC_W + c * H_NP + 64 + p] = accE[2][r4];
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int e = tid + G_THREADS * i;
    if (e < H_C2 * H_C1 * 4) {
      const int co = e / (H_C1 * 4), rem = e - co * (H_C1 * 4), k = rem / H_C1, ci = rem - k * H_C1;
      out[O_C1_W + co * (H_C1 * 4) + ci * 4 + k] = accC1[i];              // back to conv_layers.0.weight's [co][ci][k]
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int e = tid + G_THREADS * i; if (e < H_C3 * H_C2 * 2) out[O_C2_W + e] = accC2[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { const int e = tid + G_THREADS * i; if (e < H_OUT * H_C1) out[O_LIN_W + e] = accL[i]; }
  if (tid < 30) out[O_ENC_B + tid] = accB;
  else if (tid < 50) out[O_C1_B + tid - 30] = accB;
  else if (tid < 60) out[O_C2_B + tid - 50] = accB;
  else if (tid < 80) out[O_LIN_B + tid - 60] = accB;
  // loss: lanes < 24 of wave 0 hold the partial sums
  if (wave == 0) {
    float v = loss_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) out[N_GRAD] = v;
  }
}

// grad[i] = sum over workgroups of partial[wg][i] in a fixed order: a block owns 16 columns, its 16 slices of lanes each sum every
// 16th workgroup, the 16 slice sums are added in slice order. sq[i] = grad[i]^2 (i < N_GRAD) for the norm.
#define RED_COLS 16
#define RED_SLICES 16
extern "C" __global__ void __launch_bounds__(256) hist_reduce_kernel(const float* __restrict__ partial, int nwg, float* __restrict__ grad,
                                                                    float* __restrict__ sq) {
  __shared__ float sh[RED_SLICES][RED_COLS + 1];
  const int c = threadIdx.x % RED_COLS, sl = threadIdx.x / RED_COLS;
  const int i = blockIdx.x * RED_COLS + c;
  float a = 0.f;
  if (i < N_PART) for (int w = sl; w < nwg; w += RED_SLICES) a += partial[(size_t)w * N_PART + i];
  sh[sl][c] = a;
  __syncthreads();
  if (sl == 0 && i < N_PART) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RED_SLICES; ++k) t += sh[k][c];
    grad[i] = t;
    if (i < N_GRAD) sq[i] = t * t;
  }
}

struct HistAdamTable { float* p[8]; int off[9]; };

// sq[i] = g[i]^2 for a gradient that went through an all-reduce (the squares hist_reduce_kernel left are of the LOCAL gradient).
// Its own launch: hist_adam_kernel rewrites g in place, so no block of that launch may read g for the norm (a late block would
// sum entries its peers have already scaled: a different clip coefficient per block, replicas drifting apart).
extern "C" __global__ void __launch_bounds__(256) hist_sq_kernel(const float* __restrict__ g, float* __restrict__ sq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N_GRAD) { const float t = g[i]; sq[i] = t * t; }
}

// clip_grad_norm_ + Adam.step on the 8 tensors (as ppo_adam_kernel). Every block derives the squared norm itself, in a fixed
// order, from sq[] (written by an EARLIER launch: hist_reduce_kernel, or hist_sq_kernel after an all-reduce changed g).
extern "C" __global__ void __launch_bounds__(256) hist_adam_kernel(HistAdamTable T, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                                  const float* __restrict__ sq, float max_norm, float beta1,
                                                                  float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale) {
  __shared__ float sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  for (int e = threadIdx.x; e < N_GRAD; e += 256) a += sq[e];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  const float tot = sh[0];
  if (i >= N_GRAD) return;
  float coef = grad_scale;
  if (max_norm > 0.f) coef = grad_scale * fminf(max_norm / (sqrtf(tot) * grad_scale + 1e-6f), 1.f);
  int lo = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j) if (T.off[j] <= i) lo = j;
  const float gi = g[i] * coef;
  g[i] = gi;
  const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
  const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float* pp = T.p[lo] + (i - T.off[lo]);
  *pp = *pp - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

// privileged latent of every row: Linear(24 -> 64)+ELU, Linear(64 -> 20)+ELU (actor_critic.py:129-141,219-221), one lane per row,
// weights broadcast from LDS
extern "C" __global__ void __launch_bounds__(256) priv_latent_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                                    const float* __restrict__ obs, float* __restrict__ out, int rows) {
  __shared__ float sw0[PRIV_H][PRIV_N], sb0[PRIV_H], sw1[H_OUT][PRIV_H], sb1[H_OUT];
  for (int e = threadIdx.x; e < PRIV_H * PRIV_N; e += 256) (&sw0[0][0])[e] = w0[e];
  for (int e = threadIdx.x; e < H_OUT * PRIV_H; e += 256) (&sw1[0][0])[e] = w1[e];
  if (threadIdx.x < PRIV_H) sb0[threadIdx.x] = b0[threadIdx.x];
  if (threadIdx.x < H_OUT) sb1[threadIdx.x] = b1[threadIdx.x];
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float x[PRIV_N];
  const float4* src = reinterpret_cast<const float4*>(obs + (size_t)r * H_OBS + PRIV_OFF);       // 76 floats in: 16-byte aligned
#pragma unroll
  for (int j = 0; j < PRIV_N / 4; ++j) { const float4 v = src[j]; x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w; }
  float o[H_OUT];
#pragma unroll
  for (int j = 0; j < H_OUT; ++j) o[j] = sb1[j];
#pragma unroll 4
  for (int h = 0; h < PRIV_H; ++h) {
    float a = sb0[h];
#pragma unroll
    for (int k = 0; k < PRIV_N; ++k) a += sw0[h][k] * x[k];
    a = elu1(a);
#pragma unroll
    for (int j = 0; j < H_OUT; ++j) o[j] += sw1[j][h] * a;
  }
  float4* dst = reinterpret_cast<float4*>(out + (size_t)r * H_OUT);
#pragma unroll
  for (int j = 0; j < H_OUT / 4; ++j) dst[j] = make_float4(elu1(o[4 * j]), elu1(o[4 * j + 1]), elu1(o[4 * j + 2]), elu1(o[4 * j + 3]));
}

// ---- C-ABI (include/wbc_sim.h) ------------------------------------------------------------------------------
#define HIST_TRAIN_MAX_WG 512
extern "C" int wbc_hist_train_grad_floats(void) { return N_PART; }
extern "C" size_t wbc_hist_train_workspace_floats(void) { return (size_t)HIST_TRAIN_MAX_WG * N_PART + N_PART; }

extern "C" int wbc_hist_train_grad(const void* const* params, const float* obs, const float* target, const long long* idx, int rows,
                                   float* workspace, float* grad, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!params || !obs || !target || !idx || !workspace || !grad || rows <= 0) return -1;
  HistParams P;
  const float** dst = reinterpret_cast<const float**>(&P);
  for (int i = 0; i < 8; ++i) { if (!params[i]) return -1; dst[i] = static_cast<const float*>(params[i]); }
  const int ngroups = (rows + G_ROWS - 1) / G_ROWS;
  const int nwg = ngroups < HIST_TRAIN_MAX_WG ? ngroups : HIST_TRAIN_MAX_WG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hist_train_kernel, dim3(nwg), dim3(G_THREADS), 0, st, P, obs, target, idx, rows, workspace);
  hipLaunchKernelGGL(hist_reduce_kernel, dim3((N_PART + RED_COLS - 1) / RED_COLS), dim3(256), 0, st, workspace, nwg, grad, workspace + (size_t)HIST_TRAIN_MAX_WG * N_PART);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_hist_clip_adam(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                                  float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, int grad_was_reduced,
                                  float* workspace, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!params || !grad || !exp_avg || !exp_avg_sq || !workspace || !(grad_scale > 0.f)) return -1;
  static const int sizes[8] = {H_C1 * H_NP, H_C1, H_C2 * H_C1 * 4, H_C2, H_C3 * H_C2 * 2, H_C3, H_OUT * H_C1, H_OUT};
  HistAdamTable T;
  int off = 0;
  for (int i = 0; i < 8; ++i) { if (!params[i]) return -1; T.p[i] = (float*)params[i]; T.off[i] = off; off += sizes[i]; }
  T.off[8] = off;
  float* sq = workspace + (size_t)HIST_TRAIN_MAX_WG * N_PART;
  if (grad_was_reduced) hipLaunchKernelGGL(hist_sq_kernel, dim3((N_GRAD + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad, sq);
  hipLaunchKernelGGL(hist_adam_kernel, dim3((N_GRAD + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, grad, exp_avg, exp_avg_sq,
                     sq, max_norm, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_priv_latent(const void* const* params, const float* obs, float* out, int rows, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!params || !obs || !out || rows <= 0) return -1;
  for (int i = 0; i < 4; ++i) if (!params[i]) return -1;
  hipLaunchKernelGGL(priv_latent_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)params[0], (const float*)params[1],
                     (const float*)params[2], (const float*)params[3], obs, out, rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
