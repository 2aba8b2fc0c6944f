// wbc_ppo_kernel.hip -- one PPO.update() minibatch step (reference rsl_rl/algorithms/ppo.py:163-246, teacher
// path) as three launches on gfx950 instead of ~350 eager ones:
//
//   1. ppo_fwd_bwd_kernel: a workgroup takes 32 minibatch rows (gathered through the permutation index),
//      runs actor + critic forward on fp32 MFMA with all activations of the tile in LDS, evaluates the loss
//      terms -- clipped surrogate with Advantage Mixing (PPO:199-206), clipped value loss (PPO:209-214),
//      Regularized-Online-Adaptation latent regulariser (PPO:174-179) -- and their output gradients in the
//      epilogue, then back-propagates through every layer (dA = dZ W on MFMA, activation derivatives from
//      the stashed post-activations). Post-activations and pre-activation gradients go to two global
//      stashes; per-tile column sums of the latter are the bias-gradient partials.
//   2. ppo_wgrad_kernel: dW_l = dZ_l^T A_{l-1} for all 16 layers, split 64..80 ways over the rows
//      (the reduction length is the minibatch, 40960), 32x32x2 MFMA with both operands staged through LDS.
//   3. ppo_reduce_kernel: fixed-order sums of the split partials into the flat gradient buffer.
//
// Everything is deterministic (no atomics). The history-encoder latent that the regulariser targets is an
// input (its weights do not change during update(), PPO:175-176).
#include "wbc_mlp.h"

// ---- stash layouts (floats per row) -------------------------------------------------------------
// every block starts at a multiple of 4 floats and the row strides are multiples of 4: float4 accesses stay aligned
enum { A_X = 0, A_H1 = 100, A_LAT = 164, A_BB = 184, A_L1 = 312, A_L2 = 440, A_LEG = 568, A_A1 = 580, A_A2 = 708, A_ARM = 836,
       A_CB = 844, A_CL1 = 972, A_CL2 = 1100, A_CA1 = 1228, A_CA2 = 1356, A_Z = 1484, A_LD = 1584 };
enum { D_H1 = 0, D_LAT = 64, D_BB = 84, D_L1 = 212, D_L2 = 340, D_LEG = 468, D_A1 = 480, D_A2 = 608, D_ARM = 736, D_CB = 744,
       D_CL1 = 872, D_CL2 = 1000, D_VLEG = 1128, D_CA1 = 1132, D_CA2 = 1260, D_VARM = 1388, D_LD = 1392 };

struct PpoBatch {                 // flat [T*N, ...] rollout tensors + the minibatch's row indices
  const float* obs;               // [TN, 860]
  const float* actions;           // [TN, 18]
  const float* old_values;        // [TN, 2]
  const float* advantages;        // [TN, 2]
  const float* returns;           // [TN, 2]
  const float* old_logp;          // [TN, 2]
  const float* hist_latent;       // [TN, 20]  history-encoder latent of every stored row
  const int64_t* idx;             // [B]
  int B;
  float clip, value_coef, mixing, roa_coef;
  int use_clipped_value_loss;
};

struct __align__(16) PpoSmem {
  float x[PT_ROWS * 101];
  float a0[PT_ROWS * LDA], a1[PT_ROWS * LDA], a2[PT_ROWS * LDA];
  float outv[PT_ROWS * 21];       // mean 18, values 2
  float g[PT_ROWS * 41];          // output grads: dmu 18, dv 2, dlat 20
};

// Backward GEMM out[32, IN] (+)= dz[32, OUT] * W[OUT, IN]. The B operand W[k][col] is read straight from global
// memory (a wave's fragment is two runs of 32 consecutive floats of W, i.e. coalesced as stored), prefetched into
// registers by load_wrows one stage ahead of the MFMA chain that consumes it.
template <int OUT, int IN>
static __device__ __forceinline__ void load_wrows(float (&w)[64], const float* __restrict__ W) {
  constexpr int NBLK = (IN + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NBLK) {
    const int col = wave * 32 + (lane & 31);
    const bool col_ok = col < IN;
    const float* bp = W + (size_t)(lane >> 5) * IN + (col_ok ? col : 0);
#pragma unroll
    for (int t = 0; t < OUT / 2; ++t) { const float v = bp[(size_t)(2 * t) * IN]; w[t] = col_ok ? v : 0.f; }
  }
}
template <int OUT, int IN, bool ACCUMULATE>
static __device__ __forceinline__ void bwd_mma(const float* dz, int ldz, const float (&w)[64], float* out, int ldo) {
  constexpr int NBLK = (IN + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NBLK) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int col = wave * 32 + (lane & 31);
    const float* ap = dz + (lane & 31) * ldz + (lane >> 5);
#pragma unroll
    for (int t = 0; t < OUT / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * t], w[t], acc, 0, 0, 0);
    if (col < IN) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float* o = &out[row * ldo + col];
        *o = ACCUMULATE ? (*o + acc[r]) : acc[r];
      }
    }
  }
  __syncthreads();
}

// buf[32, N] <- buf * act'(A) with A the stashed post-activation (elu' = a > 0 ? 1 : a + 1, tanh' = 1 - a^2);
// the result also goes to the DZ stash. float4 global accesses, all of a thread's loads in flight together.
template <int ACT>
static __device__ __forceinline__ float act_deriv(float a) {
  return (ACT == ACT_ELU) ? (a > 0.f ? 1.f : a + 1.f) : ((ACT == ACT_TANH) ? 1.f - a * a : 1.f);
}
template <int N, int ACT>
static __device__ __forceinline__ void act_grad(float* buf, int ld, const float* __restrict__ act_stash, int acol, float* __restrict__ dz_stash,
                                                int dcol, int row0, int num_rows) {
  const int tid = threadIdx.x;
  if (N % 4 == 0) {
    constexpr int Q = N / 4, TOT = PT_ROWS * Q, PER = (TOT + PT_THREADS - 1) / PT_THREADS;
    float4 a[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = tid + j * PT_THREADS, r = e / Q, c = (e - r * Q) * 4;
      a[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (e < TOT && row0 + r < num_rows) a[j] = *reinterpret_cast<const float4*>(act_stash + (size_t)(row0 + r) * A_LD + acol + c);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = tid + j * PT_THREADS, r = e / Q, c = (e - r * Q) * 4;
      if (e < TOT) {
        float* bp = buf + r * ld + c;
        const bool ok = row0 + r < num_rows;
        float4 v;
        v.x = ok ? bp[0] * act_deriv<ACT>(a[j].x) : 0.f; v.y = ok ? bp[1] * act_deriv<ACT>(a[j].y) : 0.f;
        v.z = ok ? bp[2] * act_deriv<ACT>(a[j].z) : 0.f; v.w = ok ? bp[3] * act_deriv<ACT>(a[j].w) : 0.f;
        bp[0] = v.x; bp[1] = v.y; bp[2] = v.z; bp[3] = v.w;
        if (ok) *reinterpret_cast<float4*>(dz_stash + (size_t)(row0 + r) * D_LD + dcol + c) = v;
      }
    }
  } else {
    for (int e = tid; e < PT_ROWS * N; e += PT_THREADS) {
      const int r = e / N, c = e - r * N;
      float v = 0.f;
      if (row0 + r < num_rows) {
        v = buf[r * ld + c] * act_deriv<ACT>(act_stash[(size_t)(row0 + r) * A_LD + acol + c]);
        dz_stash[(size_t)(row0 + r) * D_LD + dcol + c] = v;
      }
      buf[r * ld + c] = v;
    }
  }
  __syncthreads();
}

extern "C" __global__ void __launch_bounds__(PT_THREADS) ppo_fwd_bwd_kernel(PolicyParams P, const float* __restrict__ wpack, PpoBatch Bt,
                                                                           float* __restrict__ act_stash, float* __restrict__ dz_stash,
                                                                           float* __restrict__ dstd_partial, float* __restrict__ loss_partial) {
  __shared__ PpoSmem s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = blockIdx.x, row0 = tile * PT_ROWS, B = Bt.B;
  // gather obs[idx, :100] (float4 loads, all in flight) and stash it as the input of priv0 / critic backbone
  load_x_tile(s.x, [&](int r) { return (row0 + r < B) ? Bt.obs + (size_t)Bt.idx[row0 + r] * PT_NOBS : (const float*)nullptr; });
  __syncthreads();
  for (int e = tid; e < PT_ROWS * 25; e += PT_THREADS) {
    const int r = e / 25, c = (e - r * 25) * 4;
    if (row0 + r < B) {
      const float* xp = s.x + r * 101 + c;
      *reinterpret_cast<float4*>(act_stash + (size_t)(row0 + r) * A_LD + A_X + c) = make_float4(xp[0], xp[1], xp[2], xp[3]);
    }
  }
  // ---------------- forward (same chain as wbc_policy_act_kernel), post-activations stashed. Two register sets of
  // weight fragments alternate so that the next layer's operands are in flight during the current MFMA chain.
  float wa[64], wb[64];
  load_frags<L_PRIV0>(wa, wpack);
  load_frags<L_PRIV2>(wb, wpack);
  mma_layer<L_PRIV0, ACT_ELU, true>(s.x + PT_NPROP, 101, wa, P.priv0_b, s.a0, LDA, 0, act_stash, A_LD, A_H1, row0, B);
  load_frags<L_BB>(wa, wpack);
  mma_layer<L_PRIV2, ACT_ELU, true>(s.a0, LDA, wb, P.priv2_b, s.a1, LDA, PT_NPROP, act_stash, A_LD, A_LAT, row0, B);
  for (int e = tid; e < PT_ROWS * PT_NPROP; e += PT_THREADS) {
    const int r = e / PT_NPROP, c = e - r * PT_NPROP;
    s.a1[r * LDA + c] = s.x[r * 101 + c];
  }
  __syncthreads();
  for (int e = tid; e < PT_ROWS * 24; e += PT_THREADS) {          // z = [prop, latent], the backbone's input, for its weight gradient
    const int r = e / 24, c = (e - r * 24) * 4;
    if (row0 + r < B) {
      const float* zp = s.a1 + r * LDA + c;
      *reinterpret_cast<float4*>(act_stash + (size_t)(row0 + r) * A_LD + A_Z + c) = make_float4(zp[0], zp[1], zp[2], zp[3]);
    }
  }
  load_frags<L_LEG0>(wb, wpack);
  mma_layer<L_BB, ACT_ELU, true>(s.a1, LDA, wa, P.bb_b, s.a2, LDA, 0, act_stash, A_LD, A_BB, row0, B);
  load_frags<L_LEG2>(wa, wpack);
  mma_layer<L_LEG0, ACT_ELU, true>(s.a2, LDA, wb, P.leg0_b, s.a0, LDA, 0, act_stash, A_LD, A_L1, row0, B);
  load_frags<L_LEG4>(wb, wpack);
  mma_layer<L_LEG2, ACT_ELU, true>(s.a0, LDA, wa, P.leg2_b, s.a1, LDA, 0, act_stash, A_LD, A_L2, row0, B);
  load_frags<L_ARM0>(wa, wpack);
  mma_layer<L_LEG4, ACT_TANH, true>(s.a1, LDA, wb, P.leg4_b, s.outv, 21, 0, act_stash, A_LD, A_LEG, row0, B);
  load_frags<L_ARM2>(wb, wpack);
  mma_layer<L_ARM0, ACT_ELU, true>(s.a2, LDA, wa, P.arm0_b, s.a0, LDA, 0, act_stash, A_LD, A_A1, row0, B);
  load_frags<L_ARM4>(wa, wpack);
  mma_layer<L_ARM2, ACT_ELU, true>(s.a0, LDA, wb, P.arm2_b, s.a1, LDA, 0, act_stash, A_LD, A_A2, row0, B);
  load_frags<L_CBB>(wb, wpack);
  mma_layer<L_ARM4, ACT_TANH, true>(s.a1, LDA, wa, P.arm4_b, s.outv, 21, PT_NLEG, act_stash, A_LD, A_ARM, row0, B);
  load_frags<L_CLEG0>(wa, wpack);
  mma_layer<L_CBB, ACT_ELU, true>(s.x, 101, wb, P.cbb_b, s.a2, LDA, 0, act_stash, A_LD, A_CB, row0, B);
  load_frags<L_CLEG2>(wb, wpack);
  mma_layer<L_CLEG0, ACT_ELU, true>(s.a2, LDA, wa, P.cleg0_b, s.a0, LDA, 0, act_stash, A_LD, A_CL1, row0, B);
  load_frags<L_CLEG4>(wa, wpack);
  mma_layer<L_CLEG2, ACT_ELU, true>(s.a0, LDA, wb, P.cleg2_b, s.a1, LDA, 0, act_stash, A_LD, A_CL2, row0, B);
  load_frags<L_CARM0>(wb, wpack);
  mma_layer<L_CLEG4, ACT_NONE, false>(s.a1, LDA, wa, P.cleg4_b, s.outv, 21, 18);
  load_frags<L_CARM2>(wa, wpack);
  mma_layer<L_CARM0, ACT_ELU, true>(s.a2, LDA, wb, P.carm0_b, s.a0, LDA, 0, act_stash, A_LD, A_CA1, row0, B);
  load_frags<L_CARM4>(wb, wpack);
  mma_layer<L_CARM2, ACT_ELU, true>(s.a0, LDA, wa, P.carm2_b, s.a1, LDA, 0, act_stash, A_LD, A_CA2, row0, B);
  mma_layer<L_CARM4, ACT_NONE, false>(s.a1, LDA, wb, P.carm4_b, s.outv, 21, 19);
  // first backward operands: requested now, consumed after the loss epilogue
  load_wrows<128, 128>(wa, P.cleg2_w);
  // ---------------- losses and output gradients: one row per lane of wave 0
  if (tid < 64) {
    const int r = tid & 31;
    const bool valid = (tid < PT_ROWS) && (row0 + r < B);
    float surr = 0.f, vls = 0.f, preg = 0.f;
    float dsd[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) dsd[j] = 0.f;
    if (valid) {
      const size_t src = (size_t)Bt.idx[row0 + r];
      const float inv2B = 1.f / (2.f * (float)B), invB = 1.f / (float)B;
      float lp[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        const float mu = s.outv[r * 21 + j], sd = P.std[j];
        const float d = Bt.actions[src * 18 + j] - mu;
        lp[j < PT_NLEG ? 0 : 1] += -(d * d) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;
      }
      const float adv0 = Bt.advantages[src * 2], adv1 = Bt.advantages[src * 2 + 1];
      const float mixed[2] = {adv0 + Bt.mixing * adv1, adv1 + Bt.mixing * adv0};               // PPO:199-201
      float dlp[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float ratio = expf(lp[c] - Bt.old_logp[src * 2 + c]);                               // PPO:202
        const float rc = fminf(fmaxf(ratio, 1.f - Bt.clip), 1.f + Bt.clip);
        const float s1 = -mixed[c] * ratio, s2 = -mixed[c] * rc;                                  // PPO:203-205
        surr += fmaxf(s1, s2);
        const bool inside = (ratio >= 1.f - Bt.clip) && (ratio <= 1.f + Bt.clip);
        const float dr = (inside || s1 > s2) ? -mixed[c] : 0.f;                                   // d max(s1,s2) / d ratio
        dlp[c] = inv2B * dr * ratio;
        // value loss PPO:209-216
        const float v = s.outv[r * 21 + 18 + c], ov = Bt.old_values[src * 2 + c], R = Bt.returns[src * 2 + c];
        float dv;
        if (Bt.use_clipped_value_loss) {
          const float dlt = v - ov;
          const float vc = ov + fminf(fmaxf(dlt, -Bt.clip), Bt.clip);
          const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
          vls += fmaxf(l1, l2);
          const float m = (dlt >= -Bt.clip && dlt <= Bt.clip) ? 1.f : 0.f;
          const float g1 = 2.f * (v - R), g2 = 2.f * (vc - R) * m;
          dv = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
        } else {
          vls += (R - v) * (R - v);
          dv = 2.f * (v - R);
        }
        s.g[r * 41 + 18 + c] = Bt.value_coef * inv2B * dv;
      }
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        const float mu = s.outv[r * 21 + j], sd = P.std[j];
        const float d = Bt.actions[src * 18 + j] - mu;
        const float gl = dlp[j < PT_NLEG ? 0 : 1];
        s.g[r * 41 + j] = gl * d / (sd * sd);                                                    // d logp / d mu
        dsd[j] = gl * (d * d / (sd * sd * sd) - 1.f / sd);                                       // d logp / d sigma
      }
      // ROA regulariser PPO:174-179: mean_B || priv_latent - hist_latent ||_2
      float dl[20], nrm = 0.f;
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        dl[k] = act_stash[(size_t)(row0 + r) * A_LD + A_LAT + k] - Bt.hist_latent[src * 20 + k];
        nrm += dl[k] * dl[k];
      }
      nrm = sqrtf(nrm);
      preg = nrm;
      const float sc = (nrm > 0.f) ? Bt.roa_coef * invB / nrm : 0.f;
#pragma unroll
      for (int k = 0; k < 20; ++k) s.g[r * 41 + 20 + k] = sc * dl[k];
    } else if (tid < PT_ROWS) {
      for (int k = 0; k < 40; ++k) s.g[r * 41 + k] = 0.f;
    }
    // reduce the tile's loss sums and sigma gradients over the 32 rows (lanes 32..63 carry zeros)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      surr += __shfl_xor(surr, off); vls += __shfl_xor(vls, off); preg += __shfl_xor(preg, off);
#pragma unroll
      for (int j = 0; j < 18; ++j) dsd[j] += __shfl_xor(dsd[j], off);
    }
    if (tid == 0) {
      loss_partial[tile * 3 + 0] = surr; loss_partial[tile * 3 + 1] = vls; loss_partial[tile * 3 + 2] = preg;
#pragma unroll
      for (int j = 0; j < 18; ++j) dstd_partial[tile * 18 + j] = dsd[j];
    }
  }
  __syncthreads();
  (void)lane;
  // ---------------- backward: critic
  for (int e = tid; e < PT_ROWS * 128; e += PT_THREADS) {          // dA_cl2 = dZ_vleg (x) W_cleg4 (1 x 128)
    const int r = e >> 7, c = e & 127;
    s.a0[r * LDA + c] = s.g[r * 41 + 18] * P.cleg4_w[c];
  }
  if (tid < PT_ROWS && row0 + tid < B) {
    dz_stash[(size_t)(row0 + tid) * D_LD + D_VLEG] = s.g[tid * 41 + 18];
    dz_stash[(size_t)(row0 + tid) * D_LD + D_VARM] = s.g[tid * 41 + 19];
  }
  __syncthreads();
  act_grad<128, ACT_ELU>(s.a0, LDA, act_stash, A_CL2, dz_stash, D_CL2, row0, B);
  load_wrows<128, 128>(wb, P.cleg0_w);
  bwd_mma<128, 128, false>(s.a0, LDA, wa, s.a1, LDA);
  act_grad<128, ACT_ELU>(s.a1, LDA, act_stash, A_CL1, dz_stash, D_CL1, row0, B);
  load_wrows<128, 128>(wa, P.carm2_w);
  bwd_mma<128, 128, false>(s.a1, LDA, wb, s.a2, LDA);                                     // a2 = dA_cb (leg part)
  for (int e = tid; e < PT_ROWS * 128; e += PT_THREADS) {
    const int r = e >> 7, c = e & 127;
    s.a0[r * LDA + c] = s.g[r * 41 + 19] * P.carm4_w[c];
  }
  __syncthreads();
  act_grad<128, ACT_ELU>(s.a0, LDA, act_stash, A_CA2, dz_stash, D_CA2, row0, B);
  load_wrows<128, 128>(wb, P.carm0_w);
  bwd_mma<128, 128, false>(s.a0, LDA, wa, s.a1, LDA);
  act_grad<128, ACT_ELU>(s.a1, LDA, act_stash, A_CA1, dz_stash, D_CA1, row0, B);
  load_wrows<PT_NLEG, 128>(wa, P.leg4_w);
  bwd_mma<128, 128, true>(s.a1, LDA, wb, s.a2, LDA);                                      // a2 += arm part
  act_grad<128, ACT_ELU>(s.a2, LDA, act_stash, A_CB, dz_stash, D_CB, row0, B);
  // ---------------- backward: actor
  for (int e = tid; e < PT_ROWS * PT_NLEG; e += PT_THREADS) {
    const int r = e / PT_NLEG, c = e - r * PT_NLEG;
    s.a0[r * LDA + c] = s.g[r * 41 + c];
  }
  __syncthreads();
  act_grad<PT_NLEG, ACT_TANH>(s.a0, LDA, act_stash, A_LEG, dz_stash, D_LEG, row0, B);
  load_wrows<128, 128>(wb, P.leg2_w);
  bwd_mma<PT_NLEG, 128, false>(s.a0, LDA, wa, s.a1, LDA);
  act_grad<128, ACT_ELU>(s.a1, LDA, act_stash, A_L2, dz_stash, D_L2, row0, B);
  load_wrows<128, 128>(wa, P.leg0_w);
  bwd_mma<128, 128, false>(s.a1, LDA, wb, s.a0, LDA);
  act_grad<128, ACT_ELU>(s.a0, LDA, act_stash, A_L1, dz_stash, D_L1, row0, B);
  load_wrows<PT_NARM, 128>(wb, P.arm4_w);
  bwd_mma<128, 128, false>(s.a0, LDA, wa, s.a2, LDA);                                     // a2 = dA_bb (leg part)
  for (int e = tid; e < PT_ROWS * PT_NARM; e += PT_THREADS) {
    const int r = e / PT_NARM, c = e - r * PT_NARM;
    s.a0[r * LDA + c] = s.g[r * 41 + PT_NLEG + c];
  }
  __syncthreads();
  act_grad<PT_NARM, ACT_TANH>(s.a0, LDA, act_stash, A_ARM, dz_stash, D_ARM, row0, B);
  load_wrows<128, 128>(wa, P.arm2_w);
  bwd_mma<PT_NARM, 128, false>(s.a0, LDA, wb, s.a1, LDA);
  act_grad<128, ACT_ELU>(s.a1, LDA, act_stash, A_A2, dz_stash, D_A2, row0, B);
  load_wrows<128, 128>(wb, P.arm0_w);
  bwd_mma<128, 128, false>(s.a1, LDA, wa, s.a0, LDA);
  act_grad<128, ACT_ELU>(s.a0, LDA, act_stash, A_A1, dz_stash, D_A1, row0, B);
  load_wrows<128, 96>(wa, P.bb_w);
  bwd_mma<128, 128, true>(s.a0, LDA, wb, s.a2, LDA);                                      // a2 += arm part
  act_grad<128, ACT_ELU>(s.a2, LDA, act_stash, A_BB, dz_stash, D_BB, row0, B);
  load_wrows<20, 64>(wb, P.priv2_w);
  bwd_mma<128, 96, false>(s.a2, LDA, wa, s.a0, LDA);                                      // a0 = dA_z [32, 96]
  for (int e = tid; e < PT_ROWS * 20; e += PT_THREADS) {                                  // d latent = dA_z[:, 76:96] + ROA gradient
    const int r = e / 20, c = e - r * 20;
    s.a1[r * LDA + c] = s.a0[r * LDA + PT_NPROP + c] + s.g[r * 41 + 20 + c];
  }
  __syncthreads();
  act_grad<20, ACT_ELU>(s.a1, LDA, act_stash, A_LAT, dz_stash, D_LAT, row0, B);
  bwd_mma<20, 64, false>(s.a1, LDA, wb, s.a0, LDA);
  act_grad<64, ACT_ELU>(s.a0, LDA, act_stash, A_H1, dz_stash, D_H1, row0, B);
}

// ---- weight and bias gradients -------------------------------------------------------------------
// dW_l[o][i] = sum_rows dZ_l[row][o] * A_{l-1}[row][i], db_l[o] = sum_rows dZ_l[row][o]. grid = (splits, layers);
// a workgroup owns a row range of one layer, wave w the output rows [32w, 32w+32). Both MFMA operands are read
// straight from the stashes (a fragment = two runs of 32 consecutive floats: coalesced as stored), the bias
// gradient falls out of one extra MFMA block whose B operand is the constant 1. No LDS, no atomics.
struct WgradLayer { int out, in, dcol, acol, goff; };   // goff: offset of this layer's weight gradient in the flat buffer
struct WgradTable { WgradLayer l[NLAYERS]; };

template <int NIB>
static __device__ __forceinline__ void wgrad_body(const WgradLayer& L, const float* __restrict__ act_stash, const float* __restrict__ dz_stash,
                                                  float* __restrict__ dst, int r_begin, int r_end) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  f32x16 acc[NIB + 1];
#pragma unroll
  for (int b = 0; b <= NIB; ++b) acc[b] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int o = wave * 32 + (lane & 31);
  const bool o_ok = o < L.out;
  const float* ap = dz_stash + L.dcol + (o_ok ? o : 0);          // clamped address + select: loads stay unconditional
  const float* bp[NIB];
  bool c_ok[NIB];
#pragma unroll
  for (int b = 0; b < NIB; ++b) {
    const int c = b * 32 + (lane & 31);
    c_ok[b] = c < L.in;
    bp[b] = act_stash + L.acol + (c_ok[b] ? c : 0);
  }
  constexpr int U = 4;                                             // k-steps (row pairs) per unrolled iteration
  int r = r_begin;
  for (; r + 2 * U <= r_end; r += 2 * U) {
    float av[U], bv[U][NIB];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const size_t row = (size_t)(r + 2 * t + half);
      av[t] = ap[row * D_LD];
#pragma unroll
      for (int b = 0; b < NIB; ++b) bv[t][b] = bp[b][row * A_LD];
    }
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const float a = o_ok ? av[t] : 0.f;
#pragma unroll
      for (int b = 0; b < NIB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, c_ok[b] ? bv[t][b] : 0.f, acc[b], 0, 0, 0);
      acc[NIB] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, acc[NIB], 0, 0, 0);
    }
  }
  for (; r < r_end; r += 2) {                                      // ragged tail
    const int row = r + half;
    const bool r_ok = row < r_end;
    const size_t rc = (size_t)(r_ok ? row : r);
    const float a = (o_ok && r_ok) ? ap[rc * D_LD] : 0.f;
#pragma unroll
    for (int b = 0; b < NIB; ++b) {
      const float v = bp[b][rc * A_LD];
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, (c_ok[b] && r_ok) ? v : 0.f, acc[b], 0, 0, 0);
    }
    acc[NIB] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, r_ok ? 1.0f : 0.f, acc[NIB], 0, 0, 0);
  }
#pragma unroll
  for (int b = 0; b < NIB; ++b) {
    const int col = b * 32 + (lane & 31);
    if (col < L.in) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int oo = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (oo < L.out) dst[(size_t)oo * L.in + col] = acc[b][q];
      }
    }
  }
  if ((lane & 31) == 0) {
    float* dbias = dst + (size_t)L.out * L.in;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int oo = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (oo < L.out) dbias[oo] = acc[NIB][q];
    }
  }
}

extern "C" __global__ void __launch_bounds__(PT_THREADS) ppo_wgrad_kernel(WgradTable tab, const float* __restrict__ act_stash,
                                                                         const float* __restrict__ dz_stash, float* __restrict__ wpart,
                                                                         int B, int rows_per_split, int nparams) {
  const WgradLayer L = tab.l[blockIdx.y];
  const int wave = threadIdx.x >> 6;
  if (wave * 32 >= L.out) return;
  const int r_begin = blockIdx.x * rows_per_split, r_end = min(B, r_begin + rows_per_split);
  float* dst = wpart + (size_t)blockIdx.x * nparams + L.goff;
  const int nib = (L.in + 31) / 32;                                // uniform per workgroup
  if (nib == 4) wgrad_body<4>(L, act_stash, dz_stash, dst, r_begin, r_end);
  else if (nib == 3) wgrad_body<3>(L, act_stash, dz_stash, dst, r_begin, r_end);
  else if (nib == 2) wgrad_body<2>(L, act_stash, dz_stash, dst, r_begin, r_end);
  else wgrad_body<1>(L, act_stash, dz_stash, dst, r_begin, r_end);
}

// grad[p] = sum_s part[s][p] in a fixed order; `stride` floats between consecutive partials
extern "C" __global__ void __launch_bounds__(256) ppo_reduce_kernel(const float* __restrict__ part, int nparts, int stride, int n,
                                                                   float* __restrict__ grad) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= nparts; s += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(s + j) * stride + p];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; s < nparts; ++s) acc += part[(size_t)s * stride + p];
  grad[p] = acc;
}

// out[j] = sum_t part[t*width + j] for a few columns and many partials: one block per column, fixed-order tree
extern "C" __global__ void __launch_bounds__(256) ppo_column_reduce_kernel(const float* __restrict__ part, int nparts, int width,
                                                                          float* __restrict__ out) {
  __shared__ float sh[256];
  const int j = blockIdx.x;
  float acc = 0.f;
  for (int t = threadIdx.x; t < nparts; t += 256) acc += part[(size_t)t * width + j];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[j] = sh[0];
}

// ---- C-ABI ----------------------------------------------------------------------------------------
// Flat gradient layout: for layer l in PolicyParams order: weight [out*in] then bias [out]; then std [18];
// then 3 loss sums (surrogate, value, priv_reg; divide by 2B, 2B, B for the means).
static const int kDcol[NLAYERS] = {D_H1, D_LAT, D_BB, D_L1, D_L2, D_LEG, D_A1, D_A2, D_ARM, D_CB, D_CL1, D_CL2, D_VLEG, D_CA1, D_CA2, D_VARM};
static const int kAcol[NLAYERS] = {A_X + PT_NPROP, A_H1, A_Z, A_BB, A_L1, A_L2, A_BB, A_A1, A_A2, A_X, A_CB, A_CL1, A_CL2, A_CB, A_CA1, A_CA2};
#define PPO_NSPLIT 56

extern "C" int wbc_ppo_grad_floats(void) {
  int n = 0;
  for (int l = 0; l < NLAYERS; ++l) n += layer_out(l) * layer_in(l) + layer_out(l);
  return n + 18 + 3;
}
extern "C" int wbc_ppo_num_splits(void) { return PPO_NSPLIT; }
// floats of workspace for a minibatch of B rows
extern "C" size_t wbc_ppo_workspace_floats(int B) {
  const size_t tiles = (size_t)(B + PT_ROWS - 1) / PT_ROWS;
  return (size_t)B * (A_LD + D_LD) + tiles * (18 + 3) + (size_t)PPO_NSPLIT * (size_t)wbc_ppo_grad_floats() + (size_t)WPACK_FLOATS;
}

static int fill_params(const void* const* params, PolicyParams* P) {
  const float** dst = reinterpret_cast<const float**>(P);
  for (int i = 0; i < 33; ++i) {
    if (!params[i]) return -1;
    dst[i] = static_cast<const float*>(params[i]);
  }
  return 0;
}

// One minibatch: gradients of loss = surrogate + value_coef*value_loss + roa_coef*priv_reg (PPO:218-221, entropy term
// handled by the caller) w.r.t. the 16 layers' weights/biases and std, into `grad` (wbc_ppo_grad_floats() floats).
extern "C" int wbc_ppo_minibatch_grad(const void* const* params, const float* obs, const float* actions, const float* old_values,
                                      const float* advantages, const float* returns, const float* old_logp, const float* hist_latent,
                                      const int64_t* idx, int B, float clip, float value_coef, float mixing, float roa_coef,
                                      int use_clipped_value_loss, float* workspace, float* grad, void* stream) {
  PolicyParams P;
  if (!params || !obs || !actions || !old_values || !advantages || !returns || !old_logp || !hist_latent || !idx || !workspace || !grad ||
      B <= 0 || fill_params(params, &P))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (B + PT_ROWS - 1) / PT_ROWS;
  const int ng = wbc_ppo_grad_floats();
  float* act_stash = workspace;
  float* dz_stash = act_stash + (size_t)B * A_LD;
  float* dstd_partial = dz_stash + (size_t)B * D_LD;
  float* loss_partial = dstd_partial + (size_t)tiles * 18;
  float* wpart = loss_partial + (size_t)tiles * 3;
  float* wpack = wpart + (size_t)PPO_NSPLIT * ng;
  hipLaunchKernelGGL(wbc_pack_weights_kernel, dim3(16, NLAYERS), dim3(256), 0, st, P, wpack);
  PpoBatch Bt{obs, actions, old_values, advantages, returns, old_logp, hist_latent, idx, B, clip, value_coef, mixing, roa_coef, use_clipped_value_loss};
  hipLaunchKernelGGL(ppo_fwd_bwd_kernel, dim3(tiles), dim3(PT_THREADS), 0, st, P, wpack, Bt, act_stash, dz_stash, dstd_partial, loss_partial);
  WgradTable tab;
  int off = 0;
  for (int l = 0; l < NLAYERS; ++l) {
    tab.l[l] = WgradLayer{layer_out(l), layer_in(l), kDcol[l], kAcol[l], off};
    off += layer_out(l) * layer_in(l) + layer_out(l);
  }
  int rows_per_split = (B + PPO_NSPLIT - 1) / PPO_NSPLIT;
  rows_per_split = (rows_per_split + 7) / 8 * 8;
  hipLaunchKernelGGL(ppo_wgrad_kernel, dim3(PPO_NSPLIT, NLAYERS), dim3(PT_THREADS), 0, st, tab, act_stash, dz_stash, wpart, B, rows_per_split, ng);
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3((off + 255) / 256), dim3(256), 0, st, wpart, PPO_NSPLIT, ng, off, grad);
  hipLaunchKernelGGL(ppo_column_reduce_kernel, dim3(18), dim3(256), 0, st, dstd_partial, tiles, 18, grad + off);
  hipLaunchKernelGGL(ppo_column_reduce_kernel, dim3(3), dim3(256), 0, st, loss_partial, tiles, 3, grad + off + 18);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
