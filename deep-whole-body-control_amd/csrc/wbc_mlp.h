// wbc_mlp.h -- shared pieces of the fused ActorCritic kernels (gfx950, fp32 MFMA 32x32x2): the
// parameter pointer table, the packed-weight layout and the LDS-resident dense layer used by both the
// rollout inference kernel (wbc_policy_kernel.hip) and the PPO update kernels (wbc_ppo_kernel.hip).
//
// Forward GEMMs read their B operand (W^T fragments) from a PRE-PACKED copy of the weights: for layer l,
// k-pair kb and 32-column block cb, the 64 floats that the 64 lanes of a wave feed to one
// v_mfma_f32_32x32x2_f32 are contiguous (lane L gets W[cb*32 + (L&31)][2*kb + (L>>5)]), so a wave's
// operand load is one coalesced 256-byte global load served by L2 (the packed table is 0.7 MB and shared
// by every workgroup) -- no staging through LDS, no barrier per layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_ROWS 32          // rows (envs / minibatch samples) per workgroup
#define PT_THREADS 256
#define LDA 129             // activation row stride in LDS (odd: conflict-free b32 fragment reads)
#define PT_NPROP 76
#define PT_NPRIV 24
#define PT_NOBS 860
#define PT_NLEG 12
#define PT_NARM 6
#define NLAYERS 16

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

// layer order = PolicyParams order
enum { L_PRIV0 = 0, L_PRIV2, L_BB, L_LEG0, L_LEG2, L_LEG4, L_ARM0, L_ARM2, L_ARM4, L_CBB, L_CLEG0, L_CLEG2, L_CLEG4, L_CARM0, L_CARM2, L_CARM4 };
__host__ __device__ constexpr int layer_out(int l) {
  return l == L_PRIV0 ? 64 : l == L_PRIV2 ? 20 : l == L_LEG4 ? 12 : l == L_ARM4 ? 6 : (l == L_CLEG4 || l == L_CARM4) ? 1 : 128;
}
__host__ __device__ constexpr int layer_in(int l) { return l == L_PRIV0 ? 24 : l == L_PRIV2 ? 64 : l == L_BB ? 96 : l == L_CBB ? 100 : 128; }
__host__ __device__ constexpr int layer_nblk(int l) { return (layer_out(l) + 31) / 32; }
__host__ __device__ constexpr int layer_pack_floats(int l) { return layer_in(l) / 2 * layer_nblk(l) * 64; }
__host__ __device__ constexpr int layer_pack_off(int l) { return l == 0 ? 0 : layer_pack_off(l - 1) + layer_pack_floats(l - 1); }
#define WPACK_FLOATS (layer_pack_off(NLAYERS - 1) + layer_pack_floats(NLAYERS - 1))

enum { ACT_NONE = 0, ACT_ELU = 1, ACT_TANH = 2 };

template <int ACT>
static __device__ __forceinline__ float apply_act(float x) {
  if (ACT == ACT_ELU) return x > 0.f ? x : __expf(x) - 1.f;   // abs error <= 1 ulp(1.0); the derivative uses the stored value
  if (ACT == ACT_TANH) return tanhf(x);
  return x;
}

// Pack all 16 weight matrices into the fragment order described above. grid = (blocks, NLAYERS).
static __global__ void wbc_pack_weights_kernel(PolicyParams P, float* __restrict__ wpack) {
  const int l = blockIdx.y;
  const float* const* wp = reinterpret_cast<const float* const*>(&P);
  const float* W = wp[2 * l];
  int N = 128, K = 128, off = 0;
  // constexpr tables evaluated per layer at run time
  for (int j = 0; j < NLAYERS; ++j) if (j == l) { N = layer_out(j); K = layer_in(j); off = layer_pack_off(j); }
  const int nblk = (N + 31) / 32, total = K / 2 * nblk * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63, frag = e >> 6;
    const int cb = frag % nblk, kb = frag / nblk;
    const int n = cb * 32 + (lane & 31), k = 2 * kb + (lane >> 5);
    wpack[off + e] = (n < N) ? W[(size_t)n * K + k] : 0.f;
  }
}

// x[32, 100] (LDS, row stride 101) <- src rows' first 100 floats. 800 float4 loads, <= 4 per thread, all in flight
// at once. `row_ptr(r)` returns the global pointer of tile row r (16-byte aligned: 860-float rows are) or nullptr.
template <typename F>
static __device__ __forceinline__ void load_x_tile(float* x, F row_ptr) {
  const int tid = threadIdx.x;
  float4 v[4];
  int rr[4], cc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e4 = tid + j * PT_THREADS;
    rr[j] = e4 / 25; cc[j] = (e4 - rr[j] * 25) * 4;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < PT_ROWS * 25) {
      const float* p = row_ptr(rr[j]);
      if (p) v[j] = *reinterpret_cast<const float4*>(p + cc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (tid + j * PT_THREADS < PT_ROWS * 25) {
      float* d = x + rr[j] * 101 + cc[j];
      d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
    }
  }
}

// ---- forward layer, split into an operand prefetch and the MFMA chain ---------------------------------
// load_frags<L>: the K/2 B-fragments of this wave's 32-column block of layer L, one coalesced 256-byte load
// each, all issued back to back (K/2 <= 64 loads in flight). Call it BEFORE running the previous layer so that
// the L2 latency hides under that layer's MFMAs.
template <int L>
static __device__ __forceinline__ void load_frags(float (&w)[64], const float* __restrict__ wpack) {
  constexpr int NBLK = layer_nblk(L), KB = layer_in(L) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NBLK) {
    const float* bp = wpack + layer_pack_off(L) + wave * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) w[kb] = bp[(size_t)kb * NBLK * 64];
  }
}

// out[32, N] = act(in[32, K] * W^T + b) for layer L with the fragments already in registers. `in`/`out` live in
// LDS (row strides ldi/ldo); wave w owns output columns [32w, 32w+32). With STASH the activated outputs of valid
// rows also go to stash[(row0+row)*lds + scol + col]. Ends with a barrier.
template <int L, int ACT, bool STASH>
static __device__ __forceinline__ void mma_layer(const float* in, int ldi, const float (&w)[64], const float* __restrict__ b, float* out,
                                                 int ldo, int col_off, float* __restrict__ stash = nullptr, int lds = 0, int scol = 0,
                                                 int row0 = 0, int num_rows = 0) {
  constexpr int N = layer_out(L), NBLK = layer_nblk(L), KB = layer_in(L) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NBLK) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* ap = in + (lane & 31) * ldi + (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kb], w[kb], acc, 0, 0, 0);
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = wave * 32 + (lane & 31);
    if (col < N) {
      const float bias = b[col];
      const bool full = row0 + PT_ROWS <= num_rows;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = apply_act<ACT>(acc[r] + bias);
        out[row * ldo + col_off + col] = v;
        if (STASH && (full || row0 + row < num_rows)) stash[(size_t)(row0 + row) * lds + scol + col] = v;
      }
    }
  }
  __syncthreads();
}

// one-shot form (operands requested and consumed in the same call)
template <int L, int ACT>
static __device__ __forceinline__ void fused_layer(const float* in, int ldi, const float* __restrict__ wpack, const float* __restrict__ b,
                                                   float* out, int ldo, int col_off, float* __restrict__ stash = nullptr, int lds = 0,
                                                   int scol = 0, int row0 = 0, int num_rows = 0) {
  float w[64];
  load_frags<L>(w, wpack);
  if (stash) mma_layer<L, ACT, true>(in, ldi, w, b, out, ldo, col_off, stash, lds, scol, row0, num_rows);
  else mma_layer<L, ACT, false>(in, ldi, w, b, out, ldo, col_off);
}
