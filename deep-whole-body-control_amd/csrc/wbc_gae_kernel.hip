// wbc_gae_kernel.hip -- RolloutStorage.compute_returns (reference rsl_rl/storage/rollout_storage.py:136-150)
// as two launches instead of T sequential eager steps: (1) reverse-time GAE scan, one lane per
// (env, reward channel), coalesced over the [N,2] minor dimensions, with per-block double-precision
// partial statistics; (2) a fixed-order reduction of the partials to (count, sum, sum of squares).
// The joint normalisation over both channels (RS:150, unbiased std) is a third, elementwise launch so
// that a multi-GPU learner can all-reduce the three statistics in between.
#include <hip/hip_runtime.h>
#include "wbc_stream_guard.h"
#include <stdint.h>

#include <string>

#define GAE_BLOCK 256

__global__ void __launch_bounds__(GAE_BLOCK) gae_scan_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                            const uint8_t* __restrict__ dones, const float* __restrict__ last_values,
                                                            float* __restrict__ returns, float* __restrict__ advantages,
                                                            double* __restrict__ partials, int T, int N, float gamma, float lam) {
  const int tid = blockIdx.x * GAE_BLOCK + threadIdx.x;   // (env, channel)
  const int M = 2 * N;
  double sum = 0.0, sq = 0.0;
  if (tid < M) {
    const int n = tid >> 1;
    float adv = 0.f;
    float next_v = last_values[tid];
    for (int t = T - 1; t >= 0; --t) {
      const size_t idx = (size_t)t * M + tid;
      const float v = values[idx];
      const float nt = 1.0f - (float)dones[(size_t)t * N + n];            // RS:143
      const float delta = rewards[idx] + nt * gamma * next_v - v;         // RS:144
      adv = delta + nt * gamma * lam * adv;                               // RS:145
      const float ret = adv + v;                                          // RS:146
      returns[idx] = ret;
      const float a = ret - v;                                            // RS:149
      advantages[idx] = a;
      sum += (double)a; sq += (double)a * (double)a;
      next_v = v;
    }
  }
  // block reduction in a fixed tree order
  __shared__ double ssum[GAE_BLOCK], ssq[GAE_BLOCK];
  ssum[threadIdx.x] = sum; ssq[threadIdx.x] = sq;
  __syncthreads();
  for (int off = GAE_BLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) { ssum[threadIdx.x] += ssum[threadIdx.x + off]; ssq[threadIdx.x] += ssq[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ssum[0]; partials[2 * blockIdx.x + 1] = ssq[0]; }
}

__global__ void gae_stats_kernel(const double* __restrict__ partials, int nblocks, double count, double* __restrict__ stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblocks; ++b) { s += partials[2 * b]; q += partials[2 * b + 1]; }
    stats[0] = count; stats[1] = s; stats[2] = q;
  }
}

__global__ void __launch_bounds__(GAE_BLOCK) gae_normalize_kernel(float* __restrict__ adv, const double* __restrict__ stats, int64_t total) {
  const double cnt = stats[0], s = stats[1], q = stats[2];
  const double mean = s / cnt;
  double var = (q - s * s / cnt) / (cnt - 1.0);          // unbiased, torch.std default (RS:150)
  var = var < 0.0 ? 0.0 : var;
  const float fmean = (float)mean, inv = (float)(1.0 / (sqrt(var) + 1e-8));
  for (int64_t i = (int64_t)blockIdx.x * GAE_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GAE_BLOCK) adv[i] = (adv[i] - fmean) * inv;
}

extern "C" int wbc_gae_workspace_doubles(int N) { return 3 + 2 * ((2 * N + GAE_BLOCK - 1) / GAE_BLOCK); }

extern "C" int wbc_gae_compute(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, float* returns,
                               float* advantages, double* stats_dev, int T, int N, float gamma, float lam, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!rewards || !values || !dones || !last_values || !returns || !advantages || !stats_dev || T <= 0 || N <= 0) return -1;
  const int nblocks = (2 * N + GAE_BLOCK - 1) / GAE_BLOCK;
  hipLaunchKernelGGL(gae_scan_kernel, dim3(nblocks), dim3(GAE_BLOCK), 0, (hipStream_t)stream, rewards, values, dones, last_values, returns,
                     advantages, stats_dev + 3, T, N, gamma, lam);
  hipLaunchKernelGGL(gae_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats_dev + 3, nblocks, (double)T * (double)N * 2.0, stats_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_gae_normalize(float* advantages, const double* stats_dev, int64_t total, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!advantages || !stats_dev || total <= 0) return -1;
  int64_t blocks = (total + GAE_BLOCK - 1) / GAE_BLOCK;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gae_normalize_kernel, dim3((unsigned)blocks), dim3(GAE_BLOCK), 0, (hipStream_t)stream, advantages, stats_dev, total);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- PPO.process_env_step's tensor work in one launch (rsl_rl/algorithms/ppo.py:129-141, rollout_storage.py:70-72) ----
// rewards[n] = (rew[n], arm_rew[n]) + gamma * values[n] * time_out[n] (bootstrap of both channels on time-outs),
// dones[n] = reset_buf[n] as uint8, written straight into the rollout-storage slot of this step.
extern "C" __global__ void __launch_bounds__(256) rollout_store_kernel(const float* __restrict__ rew, const float* __restrict__ arm_rew,
                                                                      const int64_t* __restrict__ dones, const uint8_t* __restrict__ time_outs,
                                                                      const float* __restrict__ values, float gamma,
                                                                      float* __restrict__ out_rewards, uint8_t* __restrict__ out_dones, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float to = time_outs ? (float)time_outs[i] : 0.f;
  float r0 = rew[i], r1 = arm_rew[i];
  if (time_outs) { r0 += gamma * (values[2 * i] * to); r1 += gamma * (values[2 * i + 1] * to); }
  out_rewards[2 * i] = r0; out_rewards[2 * i + 1] = r1;
  out_dones[i] = (uint8_t)(dones[i] != 0);
}

extern "C" int wbc_rollout_store(const float* rew, const float* arm_rew, const int64_t* dones, const uint8_t* time_outs, const float* values,
                                 float gamma, float* out_rewards, uint8_t* out_dones, int n, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!rew || !arm_rew || !dones || !values || !out_rewards || !out_dones || n <= 0) return -1;
  hipLaunchKernelGGL(rollout_store_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, arm_rew, dones, time_outs, values, gamma,
                     out_rewards, out_dones, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- OnPolicyRunner.learn's per-step episode bookkeeping on the device (rsl_rl/runners/on_policy_runner.py:140-154) ----
// The reference keeps cur_reward_sum / cur_arm_reward_sum / cur_episode_length [N] on the device but pulls the finished
// episodes to the host on EVERY env step (nonzero() + 3 x .cpu()) to extend three deque(maxlen=100) and one deque of the
// per-step done fraction. Here the deques are device rings the host reads once per iteration:
//   state (floats): cur[3 N] | ring[3 cap] | done_ring[cap] | header (4 ints: ring head, ring fill, step head, step fill)
// One block walks the envs in ascending order (the order `extend(...tolist())` appends in), so the ring holds exactly the
// last `cap` finished episodes the reference's deques would hold, whatever the number that finished in one step.
#include "wbc_track.h"
#define TRK_THREADS 1024
extern "C" __global__ void __launch_bounds__(TRK_THREADS) track_episodes_kernel(const float* __restrict__ rew, const float* __restrict__ arm_rew,
                                                                               const int64_t* __restrict__ dones, int n, int cap,
                                                                               float* __restrict__ state) {
  track_episodes_block<TRK_THREADS>(rew, arm_rew, dones, n, cap, state);
}

extern "C" size_t wbc_runner_track_state_floats(int n, int cap) { return 3 * (size_t)n + 4 * (size_t)cap + 4; }

extern "C" int wbc_runner_track_episodes(const float* rew, const float* arm_rew, const int64_t* dones, int n, int cap, float* state, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!rew || !arm_rew || !dones || !state || n <= 0 || cap <= 0) return -1;
  hipLaunchKernelGGL(track_episodes_kernel, dim3(1), dim3(TRK_THREADS), 0, (hipStream_t)stream, rew, arm_rew, dones, n, cap, state);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
