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
// is an MFMA GEMM over the 240 (row, time step) pairs of a group with B operands re-read from obs (L2). Since round 6 the two largest
// backward stages of conv1 -- its weight gradient (a 20 x 96 x 30 GEMM per tap, one tap per wave) and its input gradient dz1 (one
// 120 x 40 x 30 GEMM per parity of the time step) -- run on the matrix pipe too: as plain FMAs with two LDS reads each they were
// three quarters of the kernel's LDS traffic (21 M LDS instructions per launch, MFMA pipe 19 % busy: profiles/r02_pmc_hist_train.txt). At the end each
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

// development aid (-DWBC_HIST_TIMING): cycle stamps of workgroup 0's first group at the phase boundaries (tools/time_hist_train.py)
__device__ long long* g_hist_dbg = nullptr;
#ifdef WBC_HIST_TIMING
#define HSTAMP(i) do { if (g_hist_dbg && blockIdx.x == 0 && threadIdx.x == 0 && group == 0) g_hist_dbg[i] = clock64(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif
extern "C" void wbc_debug_set_hist_timing(void* dev_buf) {
  long long* pp = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hist_dbg), &pp, sizeof(pp));
}

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
  // conv1 / conv2 weight gradients of the whole launch (every element owned by one lane of one wave: plain read-add-write per group;
  // as register tiles they were 20 more VGPRs held across the forward pass, which sits at the 256-register limit)
  float gW1[H_C2][4][H_C1];             // [co][tap][ci]
  float gW2[H_C3][H_C2 * 2];            // [co][(ci, tap)]
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
  for (int e = tid; e < H_C2 * 4 * H_C1; e += G_THREADS) (&s.gW1[0][0][0])[e] = 0.f;
  for (int e = tid; e < H_C3 * H_C2 * 2; e += G_THREADS) (&s.gW2[0][0])[e] = 0.f;
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
    int tl = tid;
    asm volatile("" : "+v"(tl));                        // the thread index likewise: the index arithmetic of the per-thread phases (e / 30, e % 30, LDS
                                                        // addresses ...) is redone per group instead of being held in ~100 registers across the loop (and spilled)
    const int wave = tl >> 6, p = tl & 15, g = (tl >> 4) & 3;   // (shadow the launch-wide copies inside the loop for the same reason)
    HSTAMP(0);
    // ---- forward: per-step projection 76 -> 30 + ELU -> h1
    float4 a0[5], a1[5], a2[5], a3[5];                // the rows of ALL of this wave's blocks (mb = wave, wave + 4, + 8, + 12) are requested up front:
    fetch(a0, row0, wave);                            // the gathered rows come from HBM, and a wave that asks block by block waits a full
    fetch(a1, row0, wave + 4);                        // round trip per block (the projection was 30 k of a group's 137 k cycles)
    fetch(a2, row0, wave + 8);
    if (wave + 12 < G_PAIRS / 16) fetch(a3, row0, wave + 12);
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
      for (int j = 0; j < 5; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; a2[j] = a3[j]; }
    }
    __syncthreads();
    HSTAMP(1);
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
    HSTAMP(2);
    // ---- conv2 (20 -> 10, k2 s1) + ELU, channel-major flatten -> h3
    // (matrix pipe since round 6: rows u = (row, l) -- 72, five tiles of 16 --, K = (ci, tap) = 40, columns co = 10)
#pragma unroll 1
    for (int tile = wave; tile < 5; tile += 4) {
      const int u = tile * 16 + p, urr = u / 3, ul = u - 3 * urr;
      const bool uok = u < G_ROWS * 3;
      f32x4h d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < (H_C2 * 2) / 4; ++st) {
        const int K = 4 * st + g, ci = K >> 1, k = K & 1;
        const float av = uok ? s.h2[urr][ul + k][ci] : 0.f;
        const float bv = p < H_C3 ? s.w_c2[p][K] : 0.f;
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int u2 = tile * 16 + 4 * g + r4;
        if (u2 < G_ROWS * 3 && p < H_C3) { const int rr = u2 / 3, l = u2 - 3 * rr; s.h3[rr][p * 3 + l] = elu1(d[r4] + s.b_c2[p]); }
      }
    }
    __syncthreads();
    HSTAMP(3);
    // ---- linear_output (30 -> 20) + ELU -> y
    for (int e = tl; e < G_ROWS * H_OUT; e += G_THREADS) {
      const int rr = e / H_OUT, j = e - rr * H_OUT;
      float acc = s.b_lin[j];
#pragma unroll 15
      for (int i = 0; i < H_C1; ++i) acc += s.h3[rr][i] * s.w_lin[j][i];
      s.y[rr][j] = elu1(acc);
    }
    __syncthreads();
    HSTAMP(4);
    // ---- loss and dz4 = dL/dy * ELU'(y): one lane per row (||priv - hist||_2, mean over the minibatch's rows)
    if (tl < G_ROWS) {
      const long long gi = s.ridx[tl];
      if (gi >= 0) {
        const float* tg = target + (size_t)gi * H_OUT;
        float d[H_OUT], n2 = 0.f;
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) { d[j] = s.y[tl][j] - tg[j]; n2 += d[j] * d[j]; }
        const float nrm = sqrtf(n2);
        loss_acc += nrm;
        const float sc = nrm > 0.f ? inv_rows / nrm : 0.f;
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) s.y[tl][j] = d[j] * sc * delu_from_out(s.y[tl][j]);
      } else {
#pragma unroll
        for (int j = 0; j < H_OUT; ++j) s.y[tl][j] = 0.f;
      }
    }
    __syncthreads();
    HSTAMP(5);
    // ---- backward linear_output: weight / bias gradients, dz3 (held in registers until h3 has been read by everyone)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tl + G_THREADS * i;
      if (e < H_OUT * H_C1) {
        const int j = e / H_C1, i3 = e - j * H_C1;
        float a = 0.f;
#pragma unroll 12
        for (int rr = 0; rr < G_ROWS; ++rr) a += s.y[rr][j] * s.h3[rr][i3];
        accL[i] += a;
      }
    }
    if (tl >= 60 && tl < 80) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.y[rr][tl - 60];
      accB += a;
    }
    float dz3r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tl + G_THREADS * i;
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
      const int e = tl + G_THREADS * i;
      if (e < G_ROWS * H_C1) s.h3[e / H_C1][e % H_C1] = dz3r[i];
    }
    __syncthreads();
    HSTAMP(6);
    // ---- backward conv2: weight / bias gradients, dz2
    // (matrix pipe since round 6) weight gradient dW2[co][(ci, k)] = sum over the 72 (row, l) pairs of dz3[row][co 3 + l] h2[row][l + k][ci]:
    // column tile nt = this wave (three tiles of the 40 (ci, k) columns), 18 steps of 4 pairs
    if (wave < 3) {
      const int Kc = wave * 16 + p, ci = Kc >> 1, k = Kc & 1;
      const bool bok = Kc < H_C2 * 2;
      f32x4h accW2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
      for (int st = 0; st < (G_ROWS * 3) / 4; ++st) {
        const int m = 4 * st + g, rr = m / 3, l = m - 3 * rr;
        const float av = p < H_C3 ? s.h3[rr][p * 3 + l] : 0.f;
        const float bv = bok ? s.h2[rr][l + k][ci] : 0.f;
        accW2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accW2, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int co = 4 * g + r4;
        if (co < H_C3 && bok) s.gW2[co][Kc] += accW2[r4];
      }
    }
    if (tl >= 50 && tl < 60) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.h3[rr][(tl - 50) * 3] + s.h3[rr][(tl - 50) * 3 + 1] + s.h3[rr][(tl - 50) * 3 + 2];
      accB += a;
    }
    // dz2 = (conv2^T dz3) * ELU'(h2) on the matrix pipe: rows u = (row, lp) -- 96, six tiles --, K = (co, tap) = 20, columns ci = 20 (two tiles)
#pragma unroll 1
    for (int tile = wave; tile < 6; tile += 4) {
      const int u = tile * 16 + p, urr = u >> 2, ulp = u & 3;
      f32x4h d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < (H_C3 * 2) / 4; ++st) {
        const int K = 4 * st + g, co = K >> 1, k = K & 1, l = ulp - k;
        const float av = (l >= 0 && l < 3) ? s.h3[urr][co * 3 + (l < 0 ? 0 : (l > 2 ? 2 : l))] : 0.f;
        const float bv0 = s.w_c2[co][p * 2 + k];
        const float bv1 = (p + 16 < H_C2) ? s.w_c2[co][(p + 16) * 2 + k] : 0.f;
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv0, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv1, d1, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int u2 = tile * 16 + 4 * g + r4, rr = u2 >> 2, lp = u2 & 3;
        s.dz2[rr][lp][p] = d0[r4] * delu_from_out(s.h2[rr][lp][p]);
        if (p + 16 < H_C2) s.dz2[rr][lp][p + 16] = d1[r4] * delu_from_out(s.h2[rr][lp][p + 16]);
      }
    }
    __syncthreads();
    HSTAMP(7);
    // ---- backward conv1: weight / bias gradients (h1 still holds the activations). The weight gradient runs on the matrix pipe
    // (round 6; as 2400 thread-owned entries x 96 plain FMAs with two LDS reads each it was the kernel's largest LDS-bound phase):
    // tap k = this wave, dW1_k[co][ci] = sum over the 96 (row, l) pairs m of dz2[m][co] h1[row * 10 + 2 l + k][ci] -- a 20 x 96 x 30
    // GEMM, A = dz2^T, B = the tap's rows of h1, 24 steps of 4 pairs, four 16 x 16 tiles.
    {
      const bool a1ok = (p + 16) < H_C2, b1ok = (p + 16) < H_C1;
      f32x4h accW1[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accW1[i][j] = (f32x4h){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int st = 0; st < (G_ROWS * 4) / 4; ++st) {
        const int m = 4 * st + g, rr = m >> 2, l = m & 3;
        const float* zr = s.dz2[rr][l];
        const float* hr = s.h1[rr * H_T + 2 * l + wave];
        const float av0 = zr[p], av1 = a1ok ? zr[p + 16] : 0.f;
        const float bv0 = hr[p], bv1 = b1ok ? hr[p + 16] : 0.f;
        accW1[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0, accW1[0][0], 0, 0, 0);
        accW1[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv1, accW1[0][1], 0, 0, 0);
        accW1[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv0, accW1[1][0], 0, 0, 0);
        accW1[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1, accW1[1][1], 0, 0, 0);
      }
#pragma unroll
      for (int cot = 0; cot < 2; ++cot)
#pragma unroll
        for (int cit = 0; cit < 2; ++cit)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int co = cot * 16 + 4 * g + r4, ci = cit * 16 + p;       // D[row = 4g + r][col = p]
            if (co < H_C2 && ci < H_C1) s.gW1[co][wave][ci] += accW1[cot][cit][r4];
          }
    }
    if (tl >= 30 && tl < 50) {
      float a = 0.f;
      for (int rr = 0; rr < G_ROWS; ++rr) a += s.dz2[rr][0][tl - 30] + s.dz2[rr][1][tl - 30] + s.dz2[rr][2][tl - 30] + s.dz2[rr][3][tl - 30];
      accB += a;
    }
    __syncthreads();
    HSTAMP(8);
    // ---- dz1 = (conv1^T dz2) * ELU'(h1), in place, on the matrix pipe (round 6): time step t = 2 m + par receives tap par from l = m
    // and tap par + 2 from l = m - 1, so each parity is ONE GEMM over K = (co, j) = 40: rows u = (row, m) (120, eight tiles of 16),
    // A[u][(co, j)] = dz2[row][m - j][co] (0 outside l = 0..3), B[(co, j)][ci] = w1[co][par + 2 j][ci]; every element of h1 is
    // read (its ELU') and written by one lane. A wave takes row tiles 2 wave, 2 wave + 1 of both parities.
    {
      const bool b1ok = (p + 16) < H_C1;
#pragma unroll 1
      for (int pt = 0; pt < 4; ++pt) {
        const int par = pt >> 1, rt = 2 * wave + (pt & 1);
        const int u = rt * 16 + p, urr = u / 5, um = u - 5 * urr;
        const bool uok = u < G_ROWS * 5;
        f32x4h d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
        for (int st = 0; st < (H_C2 * 2) / 4; ++st) {
          const int K = 4 * st + g, co = K >> 1, j = K & 1, tap = par + 2 * j, l = um - j;
          const float av = (uok && l >= 0 && l <= 3) ? s.dz2[uok ? urr : 0][l < 0 ? 0 : (l > 3 ? 3 : l)][co] : 0.f;
          const float* wr = s.w1[co * 4 + tap];
          const float bv0 = wr[p], bv1 = b1ok ? wr[p + 16] : 0.f;
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv0, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv1, d1, 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int u2 = rt * 16 + 4 * g + r4;
          if (u2 < G_ROWS * 5) {
            const int rr = u2 / 5, m = u2 - 5 * rr, q = rr * H_T + 2 * m + par;
            s.h1[q][p] = d0[r4] * delu_from_out(s.h1[q][p]);
            if (b1ok) s.h1[q][p + 16] = d1[r4] * delu_from_out(s.h1[q][p + 16]);
          }
        }
      }
    }
    __syncthreads();
    HSTAMP(9);
    // ---- projection weight gradient dW[c][j] += sum_q dz1[q][c] x[q][j]: MFMA, A = dz1 (LDS), B = x (obs, L2)
    {
      const int jt0 = wave;                                      // this wave's column tile(s): jt0 for both ct, plus (ct = wave, jt = 4) for wave < 2
      const bool third = wave < 2;
      const bool a1ok = (p + 16) < H_C1;
      const bool b4ok = (64 + p) < H_NP;
      // (the B operands come from obs through L2: requested twenty steps at a time BEFORE those steps' MFMAs -- one step at a time, each
      // step waited a full global round trip for its own row: 56 k of the group's 160 k cycles, round 6's stamps)
      constexpr int CH = 20;
#pragma unroll 1
      for (int c0 = 0; c0 < G_PAIRS / 4; c0 += CH) {
        float bxv[CH], b4v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int q = 4 * (c0 + i) + g, rr = q / H_T, t = q - rr * H_T;
          const long long gi = s.ridx[rr];
          const float* xr = obs + (size_t)(gi < 0 ? 0 : gi) * H_OBS + H_OFF + t * H_NP;
          bxv[i] = xr[jt0 * 16 + p];
          b4v[i] = (third && b4ok) ? xr[64 + p] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int q = 4 * (c0 + i) + g;
          const float av0 = s.h1[q][p];
          const float av1 = a1ok ? s.h1[q][p + 16] : 0.f;
          accE[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bxv[i], accE[0], 0, 0, 0);
          accE[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bxv[i], accE[1], 0, 0, 0);
          if (third) accE[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wave == 0 ? av0 : av1, b4v[i], accE[2], 0, 0, 0);
        }
      }
    }
    if (tl < H_C1) {
      float a = 0.f;
      for (int q = 0; q < G_PAIRS; ++q) a += s.h1[q][tl];
      accB += a;
    }
    __syncthreads();                                   // h1 / ridx are rewritten by the next group
    HSTAMP(10);
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
  for (int e = tid; e < H_C2 * H_C1 * 4; e += G_THREADS) {                   // conv_layers.0.weight's [co][ci][k]
    const int co = e / (H_C1 * 4), rem = e - co * (H_C1 * 4), ci = rem >> 2, k = rem & 3;
    out[O_C1_W + e] = s.gW1[co][k][ci];
  }
  for (int e = tid; e < H_C3 * H_C2 * 2; e += G_THREADS) out[O_C2_W + e] = (&s.gW2[0][0])[e];     // conv_layers.2.weight's [co][ci][k] = co * 40 + (ci, k)
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
