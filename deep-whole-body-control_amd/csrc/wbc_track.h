// wbc_track.h -- OnPolicyRunner.learn's per-step episode bookkeeping (rsl_rl/runners/on_policy_runner.py:140-154) as ONE workgroup's
// work: shared by the stand-alone launch (wbc_runner_track_episodes, wbc_gae_kernel.hip) and by the extra workgroup of the
// episode-statistics launch (wbc_sim_episode_stats_track, wbc_sim.hip), where it costs the loop no launch of its own.
//   state (floats): cur[3 n] | ring[3 cap] | done_ring[cap] | header (4 ints: ring head, ring fill, step head, step fill)
// The workgroup walks the envs in ascending order (the order `extend(...tolist())` appends in), chunk by chunk, so the ring ends up
// holding exactly the last `cap` finished episodes the reference's deque(maxlen=cap) would hold, whatever the number that
// finished in this step: inside a chunk only the last `cap` episodes seen so far are written (distinct slots), and a later chunk
// overwrites an earlier one in program order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int THREADS>
__device__ __forceinline__ void track_episodes_block(const float* __restrict__ rew, const float* __restrict__ arm_rew,
                                                     const int64_t* __restrict__ dones, int n, int cap, float* __restrict__ state) {
  __shared__ int wave_cnt[THREADS / 64];
  float* cur = state;
  float* ring = state + 3 * (size_t)n;
  float* done_ring = ring + 3 * (size_t)cap;
  int* hdr = reinterpret_cast<int*>(done_ring + cap);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int head = hdr[0];
  int before = 0;
  for (int base = 0; base < n; base += THREADS) {
    const int i = base + tid;
    const bool in = i < n;
    float r = 0.f, a = 0.f, l = 0.f;
    bool d = false;
    if (in) {
      r = cur[3 * (size_t)i] + rew[i]; a = cur[3 * (size_t)i + 1] + arm_rew[i]; l = cur[3 * (size_t)i + 2] + 1.f;
      d = dones[i] != 0;
    }
    const unsigned long long bal = __ballot(d);
    __syncthreads();                                   // the previous chunk's reads of wave_cnt (and its ring writes) are done
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int pre = 0, chunk = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { const int k = wave_cnt[w]; pre += w < wave ? k : 0; chunk += k; }
    if (d) {
      const int rank = before + pre + __popcll(bal & ((1ull << lane) - 1ull));
      if (rank >= before + chunk - cap) {
        const int slot = (head + rank) % cap;
        ring[3 * slot] = r; ring[3 * slot + 1] = a; ring[3 * slot + 2] = l;
      }
      r = 0.f; a = 0.f; l = 0.f;
    }
    if (in) { cur[3 * (size_t)i] = r; cur[3 * (size_t)i + 1] = a; cur[3 * (size_t)i + 2] = l; }
    before += chunk;
  }
  if (tid == 0) {
    const int total = before;
    hdr[0] = (head + total) % cap;
    const int fill = hdr[1] + total;
    hdr[1] = fill < cap ? fill : cap;
    done_ring[hdr[2]] = (float)total / (float)n;
    hdr[2] = (hdr[2] + 1) % cap;
    hdr[3] = hdr[3] + 1 < cap ? hdr[3] + 1 : cap;
  }
}
