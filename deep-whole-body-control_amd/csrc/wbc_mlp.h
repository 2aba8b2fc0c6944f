// wbc_mlp.h -- shared pieces of the fused ActorCritic kernels (gfx950, fp32 MFMA 32x32x2):
// the parameter pointer table and the LDS-resident dense layer used by both the rollout inference
// kernel (wbc_policy_kernel.hip) and the PPO update kernels (wbc_ppo_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_ROWS 32          // envs per workgroup
#define PT_THREADS 256
#define LDA 129             // activation row stride (odd: conflict-free b32 fragment reads)
#define PT_NPROP 76
#define PT_NPRIV 24
#define PT_NOBS 860
#define PT_NLEG 12
#define PT_NARM 6

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct PolicyParams {       // device pointers to the torch parameters (weight [out,in] row-major, bias [out])
  const float *priv0_w, *priv0_b, *priv2_w, *priv2_b;
  const float *bb_w, *bb_b;
  const float *leg0_w, *leg0_b, *leg2_w, *leg2_b, *leg4_w, *leg4_b;
  const float *arm0_w, *arm0_b, *arm2_w, *arm2_b, *arm4_w, *arm4_b;
  const float *cbb_w, *cbb_b;
  const float *cleg0_w, *cleg0_b, *cleg2_w, *cleg2_b, *cleg4_w, *cleg4_b;
  const float *carm0_w, *carm0_b, *carm2_w, *carm2_b, *carm4_w, *carm4_b;
  const float* std;         // [18]
};

enum { ACT_NONE = 0, ACT_ELU = 1, ACT_TANH = 2 };

static __device__ __forceinline__ float apply_act(float x, int act) {
  if (act == ACT_ELU) return x > 0.f ? x : expm1f(x);
  if (act == ACT_TANH) return tanhf(x);
  return x;
}

// out[32, N] = act(in[32, K] * W[N, K]^T + b). `in`/`out` live in LDS with row stride LDA (in may have its
// own stride ldi); W is staged through `wl` (row stride K+1, odd). N <= 128, K <= 128, K % 2 == 0.
// All 256 threads call this; wave w owns output columns [32w, 32w+32).
// If `stash` is given, the activated outputs of valid rows are also written to stash[(row0+row)*lds + scol + col].
static __device__ void fused_layer(const float* in, int ldi, int K, const float* __restrict__ W, const float* __restrict__ b, int N,
                            float* out, int ldo, int col_off, float* wl, int act, float* __restrict__ stash = nullptr, int lds = 0,
                            int scol = 0, int row0 = 0, int num_rows = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldw = K + 1;
  // stage W (N x K) into LDS, coalesced along K; rows >= N of the 32-wide blocks in use are zero-filled
  const int n_pad = (N + 31) & ~31;
  for (int e = tid; e < n_pad * K; e += PT_THREADS) {
    const int n = e / K, k = e - n * K;
    wl[n * ldw + k] = (n < N) ? W[(size_t)n * K + k] : 0.f;
  }
  __syncthreads();
  if (wave * 32 < n_pad) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* ap = in + (lane & 31) * ldi + (lane >> 5);
    const float* bp = wl + (wave * 32 + (lane & 31)) * ldw + (lane >> 5);
#pragma unroll 8
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = wave * 32 + (lane & 31);
    if (col < N) {
      const float bias = b[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = apply_act(acc[r] + bias, act);
        out[row * ldo + col_off + col] = v;
        if (stash && row0 + row < num_rows) stash[(size_t)(row0 + row) * lds + scol + col] = v;
      }
    }
  }
  __syncthreads();
}

