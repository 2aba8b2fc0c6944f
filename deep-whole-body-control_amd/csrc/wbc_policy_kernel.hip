// wbc_policy_kernel.hip -- fused ActorCritic inference for the rollout (gfx950, fp32 MFMA).
//
// Replaces, per policy step, the ~60 eager launches of PPO.act (reference rsl_rl/algorithms/ppo.py:115-127:
// Actor.forward AC:204-217 with the privileged latent, Critic.forward AC:281-286, Normal.sample,
// get_actions_log_prob AC:341-345) by ONE launch: a workgroup takes 32 envs, keeps every activation of
// those 32 rows in LDS, reads each layer's weights as pre-packed MFMA fragments straight from L2 (wbc_mlp.h),
// and runs the GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32, one 32x32 output block per wave, 4 waves =
// 128 output features).
// The epilogue samples the action from pre-drawn standard normals and writes mean, action, the two
// log-probabilities (12 leg / 6 arm dims) and the two values.
#include "wbc_mlp.h"

struct __align__(16) PolicySmem {
  float x[PT_ROWS * 101];        // obs[:, :100], stride 101
  float a0[PT_ROWS * LDA], a1[PT_ROWS * LDA], a2[PT_ROWS * LDA];
  float outv[PT_ROWS * 21];      // mean 18 + value 2 (stride 21)
};

extern "C" __global__ void __launch_bounds__(PT_THREADS) wbc_policy_act_kernel(PolicyParams P, const float* __restrict__ wpack,
                                                                              const float* __restrict__ obs, const float* __restrict__ eps,
                                                                              float* __restrict__ actions, float* __restrict__ mean_out,
                                                                              float* __restrict__ logp_out, float* __restrict__ value_out,
                                                                              int num_rows) {
  __shared__ PolicySmem s;
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * PT_ROWS;
  // load obs[:, :100] of 32 rows (rows past the end are zero)
  load_x_tile(s.x, [&](int r) { return (row0 + r < num_rows) ? obs + (size_t)(row0 + r) * PT_NOBS : (const float*)nullptr; });
  __syncthreads();
  // Two register sets of weight fragments alternate: the next layer's operands are requested from L2 before the
  // current layer's MFMA chain starts.
  float wa[64], wb[64];
  load_frags<L_PRIV0>(wa, wpack);
  // ---- actor (AC:204-221): priv encoder 24 -> 64 -> 20, backbone [prop76 | latent20] -> 128, two heads
  load_frags<L_PRIV2>(wb, wpack);
  mma_layer<L_PRIV0, ACT_ELU, false>(s.x + PT_NPROP, 101, wa, P.priv0_b, s.a0, LDA, 0);
  load_frags<L_BB>(wa, wpack);
  mma_layer<L_PRIV2, ACT_ELU, false>(s.a0, LDA, wb, P.priv2_b, s.a1, LDA, PT_NPROP);     // latent -> a1[:, 76:96]
  for (int e = tid; e < PT_ROWS * PT_NPROP; e += PT_THREADS) {                           // a1[:, :76] = prop
    const int r = e / PT_NPROP, c = e - r * PT_NPROP;
    s.a1[r * LDA + c] = s.x[r * 101 + c];
  }
  __syncthreads();
  load_frags<L_LEG0>(wb, wpack);
  mma_layer<L_BB, ACT_ELU, false>(s.a1, LDA, wa, P.bb_b, s.a2, LDA, 0);                   // a2 = backbone output (kept)
  load_frags<L_LEG2>(wa, wpack);
  mma_layer<L_LEG0, ACT_ELU, false>(s.a2, LDA, wb, P.leg0_b, s.a0, LDA, 0);
  load_frags<L_LEG4>(wb, wpack);
  mma_layer<L_LEG2, ACT_ELU, false>(s.a0, LDA, wa, P.leg2_b, s.a1, LDA, 0);
  load_frags<L_ARM0>(wa, wpack);
  mma_layer<L_LEG4, ACT_TANH, false>(s.a1, LDA, wb, P.leg4_b, s.outv, 21, 0);
  load_frags<L_ARM2>(wb, wpack);
  mma_layer<L_ARM0, ACT_ELU, false>(s.a2, LDA, wa, P.arm0_b, s.a0, LDA, 0);
  load_frags<L_ARM4>(wa, wpack);
  mma_layer<L_ARM2, ACT_ELU, false>(s.a0, LDA, wb, P.arm2_b, s.a1, LDA, 0);
  load_frags<L_CBB>(wb, wpack);
  mma_layer<L_ARM4, ACT_TANH, false>(s.a1, LDA, wa, P.arm4_b, s.outv, 21, PT_NLEG);
  // ---- critic (AC:281-286): obs[:, :100] -> 128 -> two heads 128 -> 128 -> 1
  load_frags<L_CLEG0>(wa, wpack);
  mma_layer<L_CBB, ACT_ELU, false>(s.x, 101, wb, P.cbb_b, s.a2, LDA, 0);
  load_frags<L_CLEG2>(wb, wpack);
  mma_layer<L_CLEG0, ACT_ELU, false>(s.a2, LDA, wa, P.cleg0_b, s.a0, LDA, 0);
  load_frags<L_CLEG4>(wa, wpack);
  mma_layer<L_CLEG2, ACT_ELU, false>(s.a0, LDA, wb, P.cleg2_b, s.a1, LDA, 0);
  load_frags<L_CARM0>(wb, wpack);
  mma_layer<L_CLEG4, ACT_NONE, false>(s.a1, LDA, wa, P.cleg4_b, s.outv, 21, 18);
  load_frags<L_CARM2>(wa, wpack);
  mma_layer<L_CARM0, ACT_ELU, false>(s.a2, LDA, wb, P.carm0_b, s.a0, LDA, 0);
  load_frags<L_CARM4>(wb, wpack);
  mma_layer<L_CARM2, ACT_ELU, false>(s.a0, LDA, wa, P.carm2_b, s.a1, LDA, 0);
  mma_layer<L_CARM4, ACT_NONE, false>(s.a1, LDA, wb, P.carm4_b, s.outv, 21, 19);
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

static int fill_params(const void* const* params, PolicyParams* P) {
  const float** dst = reinterpret_cast<const float**>(P);
  for (int i = 0; i < 33; ++i) {
    if (!params[i]) return -1;
    dst[i] = static_cast<const float*>(params[i]);
  }
  return 0;
}

// C-ABI. params: 33 device pointers in the order of struct PolicyParams; wpack: wbc_policy_pack_floats() floats.
extern "C" int wbc_policy_pack_floats(void) { return WPACK_FLOATS; }

// Re-pack the weights into MFMA fragment order (call after the parameters changed).
extern "C" int wbc_policy_pack(const void* const* params, float* wpack, void* stream) {
  PolicyParams P;
  if (!params || !wpack || fill_params(params, &P)) return -1;
  hipLaunchKernelGGL(wbc_pack_weights_kernel, dim3(16, NLAYERS), dim3(256), 0, (hipStream_t)stream, P, wpack);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_policy_act(const void* const* params, const float* wpack, const float* obs, const float* eps, float* actions, float* mean,
                              float* logp, float* values, int num_rows, void* stream) {
  PolicyParams P;
  if (!params || !wpack || !obs || !actions || !mean || !logp || !values || num_rows <= 0 || fill_params(params, &P)) return -1;
  const int blocks = (num_rows + PT_ROWS - 1) / PT_ROWS;
  hipLaunchKernelGGL(wbc_policy_act_kernel, dim3(blocks), dim3(PT_THREADS), 0, (hipStream_t)stream, P, wpack, obs, eps, actions, mean, logp,
                     values, num_rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
