// wbc_stats.h -- the per-step episode statistics (extras["episode"] of reset_idx, widowGo1.py:743-754) and the runner's episode
// deques (wbc_track.h) as a SIDE JOB: wbc_side_job (include/wbc_sim.h) describes the work, side_job_block executes one
// workgroup's share of it. Callers: episode_stats_kernel (wbc_sim.hip: a launch of its own) and wbc_policy_act16_kernel
// (wbc_policy_kernel.hip: extra workgroups of the policy inference that follows every env step of a rollout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wbc_sim.h"
#include "wbc_track.h"

// Workgroup `blk` of the job: blk < WBC_NREW + WBC_NMETRIC: the mean over the envs that reset in the last step of column blk of
// their finished episode's reward sums [WBC_NREW] / metric sums [WBC_NMETRIC], times `scale` (1 / max_episode_length_s); no reset
// in this step -> the previously published value (prev), as the reference's extras["episode"] is only rebuilt inside reset_idx
// when env_ids is non-empty (WG:705-706, 742-750). blk == WBC_NREW + WBC_NMETRIC: the tracker. Fixed-order sums: deterministic
// for a given THREADS.
template <int THREADS>
__device__ __forceinline__ void side_job_block(const wbc_side_job& J, int blk) {
  if (blk == WBC_NREW + WBC_NMETRIC) {
    if (J.track_state) track_episodes_block<THREADS>(J.rew, J.arm_rew, J.reset_buf, J.n, J.track_cap, J.track_state);
    return;
  }
  if (blk > WBC_NREW + WBC_NMETRIC) return;
  // Thread t sums envs t, t + THREADS, ... in ascending order; flag and value are loaded together (the value unconditionally: a
  // dependent second load would double the number of memory round trips), 4 envs in flight per thread. Then a fixed butterfly
  // per wavefront and the wave sums in wave order.
  __shared__ float sh[THREADS / 64], shc[THREADS / 64];
  const int col = blk, tid = threadIdx.x, n = J.n;
  const float* src = col < WBC_NREW ? J.ep_done + col : J.met_done + (col - WBC_NREW);
  const int width = col < WBC_NREW ? WBC_NREW : WBC_NMETRIC;
  float acc = 0.f, cnt = 0.f;
  for (int base = 0; base < n; base += THREADS * 4) {
    float v[4];
    int d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + THREADS * u + tid, ic = i < n ? i : n - 1;
      d[u] = (i < n) && J.reset_buf[ic] != 0;
      v[u] = src[(size_t)ic * width];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc += d[u] ? v[u] : 0.f; cnt += d[u] ? 1.f : 0.f; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off); cnt += __shfl_xor(cnt, off); }
  if ((tid & 63) == 0) { sh[tid >> 6] = acc; shc[tid >> 6] = cnt; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { a += sh[w]; c += shc[w]; }
    J.out[col] = c > 0.f ? a / c * J.scale : (J.prev ? J.prev[col] : 0.f);
  }
}
