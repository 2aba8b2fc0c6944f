// wbc_policy_kernel.hip -- fused ActorCritic inference for the rollout (gfx950, fp32 MFMA).
//
// Replaces, per policy step, the ~60 eager launches of PPO.act (reference rsl_rl/algorithms/ppo.py:115-127:
// Actor.forward AC:204-217 with the privileged latent, Critic.forward AC:281-286, Normal.sample,
// get_actions_log_prob AC:341-345) by ONE launch: a workgroup takes 32 envs, keeps every activation of
// those 32 rows in LDS, streams each layer's weights [out,in] through LDS once, and runs the GEMMs on
// v_mfma_f32_32x32x2_f32 (exact fp32, one 32x32 output block per wave, 4 waves = 128 output features).
// The epilogue samples the action from pre-drawn standard normals and writes mean, action, the two
// log-probabilities (12 leg / 6 arm dims) and the two values.
#include "wbc_mlp.h"

struct __align__(16) PolicySmem {
  float x[PT_ROWS * 101];        // obs[:, :100], stride 101
  float a0[PT_ROWS * LDA], a1[PT_ROWS * LDA], a2[PT_ROWS * LDA];
  float wl[128 * LDA];           // staged weights
  float outv[PT_ROWS * 21];      // mean 18 + value 2 (stride 21)
};

extern "C" __global__ void __launch_bounds__(PT_THREADS) wbc_policy_act_kernel(PolicyParams P, const float* __restrict__ obs,
                                                                              const float* __restrict__ eps, float* __restrict__ actions,
                                                                              float* __restrict__ mean_out, float* __restrict__ logp_out,
                                                                              float* __restrict__ value_out, int num_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PolicySmem& s = *reinterpret_cast<PolicySmem*>(smem_raw);
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * PT_ROWS;
  // load obs[:, :100] of 32 rows (rows past the end are zero)
  for (int e = tid; e < PT_ROWS * 100; e += PT_THREADS) {
    const int r = e / 100, c = e - r * 100;
    s.x[r * 101 + c] = (row0 + r < num_rows) ? obs[(size_t)(row0 + r) * PT_NOBS + c] : 0.f;
  }
  __syncthreads();
  // ---- actor (AC:204-221): priv encoder 24 -> 64 -> 20, backbone [prop76 | latent20] -> 128, two heads
  fused_layer(s.x + PT_NPROP, 101, PT_NPRIV, P.priv0_w, P.priv0_b, 64, s.a0, LDA, 0, s.wl, ACT_ELU);
  // backbone input z = [prop, latent] assembled in a1: latent goes to columns 76..95
  fused_layer(s.a0, LDA, 64, P.priv2_w, P.priv2_b, 20, s.a1, LDA, PT_NPROP, s.wl, ACT_ELU);
  for (int e = tid; e < PT_ROWS * PT_NPROP; e += PT_THREADS) {
    const int r = e / PT_NPROP, c = e - r * PT_NPROP;
    s.a1[r * LDA + c] = s.x[r * 101 + c];
  }
  __syncthreads();
  fused_layer(s.a1, LDA, 96, P.bb_w, P.bb_b, 128, s.a2, LDA, 0, s.wl, ACT_ELU);          // a2 = backbone output (kept)
  fused_layer(s.a2, LDA, 128, P.leg0_w, P.leg0_b, 128, s.a0, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a0, LDA, 128, P.leg2_w, P.leg2_b, 128, s.a1, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a1, LDA, 128, P.leg4_w, P.leg4_b, PT_NLEG, s.outv, 21, 0, s.wl, ACT_TANH);
  fused_layer(s.a2, LDA, 128, P.arm0_w, P.arm0_b, 128, s.a0, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a0, LDA, 128, P.arm2_w, P.arm2_b, 128, s.a1, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a1, LDA, 128, P.arm4_w, P.arm4_b, PT_NARM, s.outv, 21, PT_NLEG, s.wl, ACT_TANH);
  // ---- critic (AC:281-286): obs[:, :100] -> 128 -> two heads 128 -> 128 -> 1
  fused_layer(s.x, 101, 100, P.cbb_w, P.cbb_b, 128, s.a2, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a2, LDA, 128, P.cleg0_w, P.cleg0_b, 128, s.a0, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a0, LDA, 128, P.cleg2_w, P.cleg2_b, 128, s.a1, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a1, LDA, 128, P.cleg4_w, P.cleg4_b, 1, s.outv, 21, 18, s.wl, ACT_NONE);
  fused_layer(s.a2, LDA, 128, P.carm0_w, P.carm0_b, 128, s.a0, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a0, LDA, 128, P.carm2_w, P.carm2_b, 128, s.a1, LDA, 0, s.wl, ACT_ELU);
  fused_layer(s.a1, LDA, 128, P.carm4_w, P.carm4_b, 1, s.outv, 21, 19, s.wl, ACT_NONE);
  // ---- epilogue: sample, log-probabilities (Normal.log_prob summed over leg / arm dims), outputs
  if (tid < PT_ROWS && row0 + tid < num_rows) {
    const int r = tid;
    const size_t g = (size_t)(row0 + r);
    float lp_leg = 0.f, lp_arm = 0.f;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const float mu = s.outv[r * 21 + j], sd = P.std[j];
      const float e = eps ? eps[g * 18 + j] : 0.f;
      const float a = mu + sd * e;
      const float d = a - mu;
      const float lp = -(d * d) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;
      if (j < PT_NLEG) lp_leg += lp; else lp_arm += lp;
      actions[g * 18 + j] = a;
      mean_out[g * 18 + j] = mu;
    }
    logp_out[g * 2] = lp_leg; logp_out[g * 2 + 1] = lp_arm;
    value_out[g * 2] = s.outv[r * 21 + 18]; value_out[g * 2 + 1] = s.outv[r * 21 + 19];
  }
}

// C-ABI: one fused ActorCritic inference over `num_rows` observations (device pointers).
// params: 33 device pointers in the order of struct PolicyParams.
extern "C" int wbc_policy_act(const void* const* params, const float* obs, const float* eps, float* actions, float* mean, float* logp,
                              float* values, int num_rows, void* stream) {
  if (!params || !obs || !actions || !mean || !logp || !values || num_rows <= 0) return -1;
  PolicyParams P;
  const float** dst = reinterpret_cast<const float**>(&P);
  for (int i = 0; i < 33; ++i) {
    if (!params[i]) return -1;
    dst[i] = static_cast<const float*>(params[i]);
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wbc_policy_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PolicySmem)) != hipSuccess) return -2;
    attr_set = true;
  }
  const int blocks = (num_rows + PT_ROWS - 1) / PT_ROWS;
  hipLaunchKernelGGL(wbc_policy_act_kernel, dim3(blocks), dim3(PT_THREADS), sizeof(PolicySmem), (hipStream_t)stream, P, obs, eps, actions, mean,
                     logp, values, num_rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
