// wbc_policy_kernel.hip -- fused ActorCritic inference for the rollout (gfx950, fp32 MFMA).
//
// Replaces, per policy step, the ~60 eager launches of PPO.act (reference rsl_rl/algorithms/ppo.py:115-127:
// Actor.forward AC:204-217 with the privileged latent, Critic.forward AC:281-286, Normal.sample,
// get_actions_log_prob AC:341-345) by ONE launch: a workgroup takes 32 envs, keeps every activation of
// those 32 rows in LDS, reads each layer's weights as pre-packed MFMA fragments straight from L2 (wbc_mlp.h),
// and runs the GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32; 16-row tiles: 4096 envs = 512 workgroups).
// The epilogue samples the action from pre-drawn standard normals and writes mean, action, the two
// log-probabilities (12 leg / 6 arm dims) and the two values.
// Grid = (row tiles, 2): blockIdx.y = 0 runs the actor (9 layers), 1 the critic (7 layers), so that 4096 envs give
// 256 workgroups (one per CU) and the dependent layer chain per workgroup is half as long.
#include "wbc_mlp.h"
#include "wbc_stats.h"
#include "wbc_stream_guard.h"

// ==== 16-row tiles (v_mfma_f32_16x16x4_f32) =====================================================================
// 4096 envs would be only 128 tiles of 32 rows (the first version: v_mfma_f32_32x32x2_f32, one workgroup per CU, one wave per
// SIMD, nothing to hide the LDS / barrier / operand latencies of the 9-layer chain behind). With 16-row tiles the same batch
// gives 512 workgroups, each layer costs a wave 64 MFMAs of 32 cycles instead of 64 cycles and the epilogue handles 8 outputs
// per lane instead of 16 (helpers in wbc_mlp.h). Measured faster at every batch size (4096 rows: 33 vs 51 us, 40960: 211 vs
// 281 us), so the 32-row kernel is gone.
#define T_X 0
#define T_A0 (R16 * LD16)
#define T_A1 (T_A0 + R16 * LD16)
#define T_A2 (T_A1 + R16 * LD16)
#define T_OUTV (T_A2 + R16 * LD16)
#define T_END (T_OUTV + R16 * 21)

static Tab16 make_tab16() {
  Tab16 t;
  const int in_off[NLAYERS] = {T_X + PT_NPROP, T_A0, T_A1, T_A2, T_A0, T_A1, T_A2, T_A0, T_A1, T_X, T_A2, T_A0, T_A1, T_A2, T_A0, T_A1};
  const int out_off[NLAYERS] = {T_A0, T_A1 + PT_NPROP, T_A2, T_A0, T_A1, T_OUTV, T_A0, T_A1, T_OUTV + PT_NLEG, T_A2, T_A0, T_A1, T_OUTV + 18,
                                T_A0, T_A1, T_OUTV + 19};
  for (int l = 0; l < NLAYERS; ++l) t.l[l] = make_desc16(l, in_off[l], out_off[l], -1);
  return t;
}

// Development aid (-DWBC_PPO_TIMING builds): when set to a device buffer of >= 128 int64, workgroup (0, 0) records clock64() at the four
// stage boundaries of every layer (slots 32 + 4 layer + {0: start, 1: MFMA chain done, 2: epilogue done, 3: barrier passed}) and at the
// kernel's own stages (slots 0..3: entry, inputs in LDS, layers done, outputs written). tools/time_act_layers.py prints them.
#ifdef WBC_PPO_TIMING
extern "C" void wbc_debug_set_policy_timing(void* dev_buf) {
  long long* p = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_dbg), &p, sizeof(p));
}
#define KSTAMP(i) do { if (g_mlp_dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_mlp_dbg[i] = clock64(); } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif
#ifndef ACT16_OCC
#define ACT16_OCC 2
#endif
extern "C" __global__ void __launch_bounds__(PT_THREADS, ACT16_OCC) wbc_policy_act16_kernel(PolicyParams P, Tab16 T, const float* __restrict__ wpack16,
                                                                                const float* __restrict__ bias, const float* __restrict__ obs,
                                                                                const float* __restrict__ latent, const float* __restrict__ eps,
                                                                                float* __restrict__ actions, float* __restrict__ mean_out,
                                                                                float* __restrict__ logp_out, float* __restrict__ value_out, int num_rows,
                                                                                int tiles, wbc_side_job job) {
  // workgroups past the row tiles carry a side job (the env step's episode statistics: wbc_stats.h) next to the inference
  if ((int)blockIdx.x >= tiles) {
    if (blockIdx.y == 0) side_job_block<PT_THREADS>(job, (int)blockIdx.x - tiles);
    return;
  }
  __shared__ __attribute__((aligned(16))) float smem[T_END];
  const int tid = threadIdx.x;
  KSTAMP(0);
  const int row0 = blockIdx.x * R16;
  const bool critic = blockIdx.y != 0;
  // x[16][128] <- obs[:, :100] (float4 loads); rows past the end and the columns 100..127 (k padding of the first layers) zero
  for (int e4 = tid; e4 < R16 * 32; e4 += PT_THREADS) {
    const int r = e4 >> 5, c = (e4 & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < num_rows && c < 100) v = *reinterpret_cast<const float4*>(obs + (size_t)(row0 + r) * PT_NOBS + c);
    float* dst = smem + T_X + r * LD16 + c;
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  __syncthreads();
  if (!critic) {
    for (int e = tid; e < R16 * PT_NPROP; e += PT_THREADS) {
      const int r = e / PT_NPROP, c = e - r * PT_NPROP;
      smem[T_A1 + r * LD16 + c] = smem[T_X + r * LD16 + c];
    }
    if (latent)
      for (int e = tid; e < R16 * 20; e += PT_THREADS) {
        const int r = e / 20, c = e - r * 20;
        smem[T_A1 + r * LD16 + PT_NPROP + c] = (row0 + r < num_rows) ? latent[(size_t)(row0 + r) * 20 + c] : 0.f;
      }
    __syncthreads();
  }
  KSTAMP(1);
  const int lbeg = critic ? L_CBB : (latent ? L_BB : 0), lend = critic ? NLAYERS : L_CBB;
  {
    float wa[66], wb[66];
    load16(wa, T.l[lbeg], wpack16, bias);
#pragma unroll 1
    for (int l = lbeg; l < lend; l += 2) {
      const bool two = l + 1 < lend;
      if (two) load16(wb, T.l[l + 1], wpack16, bias);
      run16<1>(wa, T.l[l], smem, nullptr, row0, 0, T.l[l], false, NoHook(), l);
      if (two) {
        if (l + 2 < lend) load16(wa, T.l[l + 2], wpack16, bias);
        run16<1>(wb, T.l[l + 1], smem, nullptr, row0, 0, T.l[l], false, NoHook(), l + 1);
      }
    }
  }
  KSTAMP(2);
  const float* outv = smem + T_OUTV;
  if (critic) {
    if (tid < R16 && row0 + tid < num_rows) {
      const size_t g = (size_t)(row0 + tid);
      value_out[g * 2] = outv[tid * 21 + 18]; value_out[g * 2 + 1] = outv[tid * 21 + 19];
    }
    return;
  }
  // Sampling and log-probabilities: thread (r, c) of a 16 x 16 grid takes actions c and (c < 2) 16 + c of row r; the leg / arm
  // log-probability sums run along the 16 lanes of the row (xor butterfly) -- 18 x (division, log, 2 stores) on one lane per row
  // before, in the workgroups the launch ends with.
  {
    const int r = tid >> 4, c = tid & 15;
    const bool live = row0 + r < num_rows;
    const size_t g = (size_t)(row0 + (live ? r : 0));
    float lp_leg = 0.f, lp_arm = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = c + 16 * h;
      if (j < 18) {
        const float mu = outv[r * 21 + j], sd = P.std[j];
        const float e = (eps && live) ? eps[g * 18 + j] : 0.f;
        const float a = mu + sd * e;
        const float dd = a - mu;
        const float lp = -(dd * dd) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;
        if (j < PT_NLEG) lp_leg += lp; else lp_arm += lp;
        if (live) { actions[g * 18 + j] = a; mean_out[g * 18 + j] = mu; }
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { lp_leg += __shfl_xor(lp_leg, off); lp_arm += __shfl_xor(lp_arm, off); }
    if (live && c == 0) { logp_out[g * 2] = lp_leg; logp_out[g * 2 + 1] = lp_arm; }
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
extern "C" int wbc_policy_pack_floats(void) { return WPACK16_OFF + WPACK16_FLOATS; }

// Re-pack the weights into MFMA fragment order (call after the parameters changed).
extern "C" int wbc_policy_pack(const void* const* params, float* wpack, void* stream) {
  StreamDeviceGuard sdg(stream);
  PolicyParams P;
  if (!params || !wpack || (reinterpret_cast<uintptr_t>(wpack) & 15) || fill_params(params, &P)) return -1;
  hipLaunchKernelGGL(wbc_pack16_kernel, dim3(8, NLAYERS, 1), dim3(256), 0, (hipStream_t)stream, P, make_pack16_table(), wpack + WPACK16_OFF);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_policy_act_job(const void* const* params, const float* wpack, const float* obs, const float* latent, const float* eps,
                                  float* actions, float* mean, float* logp, float* values, int num_rows, const wbc_side_job* job, void* stream) {
  StreamDeviceGuard sdg(stream);
  PolicyParams P;
  if (!params || !wpack || !obs || !actions || !mean || !logp || !values || num_rows <= 0 || fill_params(params, &P)) return -1;
  static const Tab16 T16 = make_tab16();
  wbc_side_job J;
  if (job) J = *job; else { J = wbc_side_job(); J.nblocks = 0; }
  if (J.nblocks < 0 || (J.nblocks > 0 && !J.out)) return -1;
  const int tiles = (num_rows + R16 - 1) / R16;
  hipLaunchKernelGGL(wbc_policy_act16_kernel, dim3(tiles + J.nblocks, 2), dim3(PT_THREADS), 0, (hipStream_t)stream, P, T16,
                     wpack + WPACK16_OFF, wpack + WPACK16_OFF + WPACK16_BIAS_OFF, obs, latent, eps, actions, mean, logp, values, num_rows, tiles, J);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_policy_act(const void* const* params, const float* wpack, const float* obs, const float* latent, const float* eps,
                              float* actions, float* mean, float* logp, float* values, int num_rows, void* stream) {
  return wbc_policy_act_job(params, wpack, obs, latent, eps, actions, mean, logp, values, num_rows, nullptr, stream);
}
