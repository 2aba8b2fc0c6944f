// wbc_ppo_kernel.hip -- one PPO.update() minibatch step (reference rsl_rl/algorithms/ppo.py:163-246, teacher
// path) as three launches on gfx950 instead of ~350 eager ones:
//
//   1. chain_pack_kernel + ppo_chain_kernel (wbc_ppo_chain.h): one WAVEFRONT per (16 minibatch rows, actor | critic) runs its
//      part's layer chain forward on fp32 MFMA with every activation in registers, evaluates its loss terms -- clipped
//      surrogate with Advantage Mixing (PPO:199-206), clipped value loss (PPO:209-214), Regularized-Online-Adaptation latent
//      regulariser (PPO:174-179) -- and back-propagates (dA^T = W^T dZ^T on MFMA, activation derivatives from the stashed
//      post-activations). Post-activations and pre-activation gradients go to two global stashes.
//   2. ppo_wgrad_kernel: dW_l = dZ_l^T A_{l-1} for all 16 layers, split 64..80 ways over the rows
//      (the reduction length is the minibatch, 40960), 32x32x2 MFMA with both operands read straight from the stashes.
//   3. ppo_grad_reduce_kernel: fixed-order sums of the split partials into the flat gradient buffer, the std-gradient and
//      loss columns, and the partial sums of squares the gradient clip needs.
//
// Everything is deterministic (no atomics). The history-encoder latent that the regulariser targets is an
// input (its weights do not change during update(), PPO:175-176).
#include "wbc_mlp.h"
#include "wbc_stream_guard.h"

// ---- stash layouts (floats per row) -------------------------------------------------------------
// every block starts at a multiple of 4 floats and the row strides are multiples of 4: float4 accesses stay aligned
enum { A_X = 0, A_H1 = 100, A_LAT = 164, A_BB = 184, A_L1 = 312, A_L2 = 440, A_LEG = 568, A_A1 = 580, A_A2 = 708, A_ARM = 836,
       A_CB = 844, A_CL1 = 972, A_CL2 = 1100, A_CA1 = 1228, A_CA2 = 1356, A_Z = 1484, A_LD = 1584 };
enum { D_H1 = 0, D_LAT = 64, D_BB = 84, D_L1 = 212, D_L2 = 340, D_LEG = 468, D_A1 = 480, D_A2 = 608, D_ARM = 736, D_CB = 744,
       D_CL1 = 872, D_CL2 = 1000, D_VLEG = 1128, D_CA1 = 1132, D_CA2 = 1260, D_VARM = 1388, D_LD = 1392 };

// The stashes are SLAB-MAJOR: the columns [c0, c0 + w) that one layer produces / consumes form a contiguous [B, w] block
// at offset c0 * B (element (row, c0 + c) at c0 * B + row * w + c). A wave of the weight-gradient kernel then streams a
// contiguous block instead of 128-byte pieces 6 KB apart, and a tile's epilogue writes 32 adjacent rows.
static const int kAslabs[] = {A_X, A_H1, A_LAT, A_BB, A_L1, A_L2, A_LEG, A_A1, A_A2, A_ARM, A_CB, A_CL1, A_CL2, A_CA1, A_CA2, A_Z, A_LD};
static const int kDslabs[] = {D_H1, D_LAT, D_BB, D_L1, D_L2, D_LEG, D_A1, D_A2, D_ARM, D_CB, D_CL1, D_CL2, D_VLEG, D_CA1, D_CA2, D_VARM, D_LD};
static int slab_w(const int* slabs, int n, int c0) {
  for (int i = 0; i + 1 < n; ++i) if (slabs[i] == c0) return slabs[i + 1] - c0;
  return -1;
}
static int a_slab_w(int c0) { return slab_w(kAslabs, 17, c0); }
static int d_slab_w(int c0) { return slab_w(kDslabs, 17, c0); }
static __device__ __forceinline__ size_t sidx(int B, int c0, int w, int row, int c) { return (size_t)c0 * B + (size_t)row * w + c; }

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
  int Bs;                         // rows of a stash slab (>= B; the 16-row kernel pads to whole tiles and stores every tile row)
  float clip, value_coef, mixing, roa_coef;
  int use_clipped_value_loss;
};

// development aid (tools/time_ppo.py builds a variant with -DWBC_PPO_TIMING): clock64 stamps of workgroup 0
__device__ long long* g_ppo_dbg = nullptr;
extern "C" void wbc_debug_set_ppo_timing(void* dev_buf) {
  long long* p = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ppo_dbg), &p, sizeof(p));
#ifdef WBC_PPO_TIMING
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_dbg), &p, sizeof(p));
#endif
}

#include "wbc_ppo_chain.h"

// ---- weight and bias gradients -------------------------------------------------------------------
// dW_l[o][i] = sum_rows dZ_l[row][o] * A_{l-1}[row][i], db_l[o] = sum_rows dZ_l[row][o]. grid = (splits, layers);
// a workgroup owns a row range of one layer, wave w the output rows [32w, 32w+32). Both MFMA operands are read
// straight from the stashes (a fragment = two runs of 32 consecutive floats: coalesced as stored), the bias
// gradient falls out of one extra MFMA block whose B operand is the constant 1. No LDS, no atomics.
struct WgradLayer { int out, in, dcol, dw, acol, aw, aoff, goff, nsplit, rows, ldw, boff, bias; };   // ldw: row stride of dW (the layer's full input width), goff: first element of this column range, boff: the bias gradient (written iff bias)   // nsplit row ranges of `rows` rows: proportional to the layer's MFMAs per row pair, so every wave has the same work   // dZ slab (start, width), A slab (start, width, first column in it); goff: offset of this layer's weight gradient in the flat buffer
#ifndef WG_U
#define WG_U 4               // row pairs per batch of operand loads (WG_STAGES - 1 batches in flight per wave behind the one being computed)
#endif

// NIB = 32-column blocks of the layer's input. IL (NIB == 4 only): block b holds the input columns 4 j + b (j = lane & 31)
// instead of 32 b + j, so that ONE 16-byte load per lane and row pair feeds all four MFMAs. `wave` = 32-row block of the
// layer's outputs this wave accumulates.
template <int NIB, bool IL = false>
static __device__ __forceinline__ void wgrad_body(const WgradLayer& L, const float* __restrict__ act_stash, const float* __restrict__ dz_stash,
                                                  float* __restrict__ dst, int r_begin, int r_end, int B, int wave) {     // B = slab rows
  const int lane = threadIdx.x & 63, half = lane >> 5;
  f32x16 acc[NIB];
#pragma unroll
  for (int b = 0; b < NIB; ++b) acc[b] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;        // bias gradient: this lane's column of dZ summed over its rows (even rows on lanes 0..31, odd on 32..63)
  const int o = wave * 32 + (lane & 31);
  const bool o_ok = o < L.out;
  const float* ap = dz_stash + (size_t)L.dcol * B + (o_ok ? o : 0);          // (ragged tail) clamped address + select
  const size_t dstr = (size_t)L.dw, astr = (size_t)L.aw;
  const float* bp[NIB];
  bool c_ok[NIB];
#pragma unroll
  for (int b = 0; b < NIB; ++b) {
    const int c = IL ? 4 * (lane & 31) + b : b * 32 + (lane & 31);
    c_ok[b] = c < L.in;
    bp[b] = act_stash + (size_t)L.acol * B + L.aoff + (c_ok[b] ? c : 0);
  }
  constexpr int U = WG_U;
  // Main loop. On this GPU vector-ALU instructions do not overlap the MFMAs of other waves on the same SIMD (measured: the
  // kernel took the same 240 us with every operand load removed, and for 1..9 workgroups per CU), so everything that is
  // not an MFMA is paid in full: no per-operand selects (columns / output rows past the layer's width read finite
  // neighbours inside the workspace and only feed accumulator entries that are never stored), uniform base + 32-bit lane
  // offset addressing, one offset add per batch. Two operand sets: the loads of batch i+1 are issued before the MFMAs of
  // batch i (2 x U row pairs in flight: this wave is alone on its SIMD, nothing else hides the memory latency).
  const char* dbase = reinterpret_cast<const char*>(dz_stash + (size_t)L.dcol * B);
  const char* abase = reinterpret_cast<const char*>(act_stash + (size_t)L.acol * B + L.aoff);
  uint32_t doff = (uint32_t)(((r_begin + half) * L.dw + o) * 4);
  uint32_t aoff = (uint32_t)(((r_begin + half) * L.aw + (IL ? 4 * (lane & 31) : (lane & 31))) * 4);
  const uint32_t dstep = 2u * (uint32_t)L.dw * 4u, astep = 2u * (uint32_t)L.aw * 4u;      // bytes per row pair
  struct Ops { float av[U], bv[U][NIB]; };
  auto issue = [&](Ops& q) {
#pragma unroll
    for (int t = 0; t < U; ++t) {
      q.av[t] = *reinterpret_cast<const float*>(dbase + (doff + t * dstep));
      if (IL) {
        const float4 v = *reinterpret_cast<const float4*>(abase + (aoff + t * astep));
        q.bv[t][0] = v.x; q.bv[t][1 % NIB] = v.y; q.bv[t][2 % NIB] = v.z; q.bv[t][3 % NIB] = v.w;
      } else {
#pragma unroll
        for (int b = 0; b < NIB; ++b) q.bv[t][b] = *reinterpret_cast<const float*>(abase + (aoff + t * astep) + b * 128);
      }
    }
    doff += U * dstep; aoff += U * astep;
  };
  auto compute = [&](const Ops& q) {
#pragma unroll
    for (int t = 0; t < U; ++t) {
#pragma unroll
      for (int b = 0; b < NIB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(q.av[t], q.bv[t][b], acc[b], 0, 0, 0);
      bsum += q.av[t];
    }
  };
  int r = r_begin;
  const int nfull = (r_end - r_begin) / (2 * U);                   // full batches
#ifndef WG_STAGES
#define WG_STAGES 3        // with three waves per SIMD, 2 and 3 measured the same (168.8 vs 168.2 us); with two (round 5) 3 is better: 155 vs 165
#endif
  if (nfull > 0) {
#if WG_STAGES == 3
    // Three operand sets: while a batch's MFMAs run, the loads of the next TWO batches are in flight (2 x 1024 cycles of
    // latency hiding per wave instead of one; 60 operand registers).
    Ops A, Bq, Cq;
    issue(A);
    issue(Bq);                                  // (past the last batch these read the following rows, inside the workspace, unused)
    int i = 0;
#pragma unroll 1
    for (; i + 3 <= nfull; i += 3) {
      issue(Cq);
      __builtin_amdgcn_sched_barrier(0);        // keep the loads ahead of the MFMAs (the scheduler sinks them to their uses otherwise)
      compute(A);
      __builtin_amdgcn_sched_barrier(0);
      issue(A);
      __builtin_amdgcn_sched_barrier(0);
      compute(Bq);
      __builtin_amdgcn_sched_barrier(0);
      issue(Bq);
      __builtin_amdgcn_sched_barrier(0);
      compute(Cq);
      __builtin_amdgcn_sched_barrier(0);
      r += 6 * U;
    }
    if (i < nfull) { compute(A); r += 2 * U; ++i; }
    if (i < nfull) { compute(Bq); r += 2 * U; }
#else
    Ops A, Bq;
    issue(A);
    int i = 0;
#pragma unroll 1
    for (; i + 2 <= nfull; i += 2) {
      issue(Bq);
      __builtin_amdgcn_sched_barrier(0);        // keep the loads ahead of the MFMAs (the scheduler sinks them to their uses otherwise)
      compute(A);
      __builtin_amdgcn_sched_barrier(0);
      issue(A);                                 // past the last batch this reads the following rows (inside the workspace), unused
      __builtin_amdgcn_sched_barrier(0);
      compute(Bq);
      __builtin_amdgcn_sched_barrier(0);
      r += 4 * U;
    }
    if (i < nfull) { compute(A); r += 2 * U; }
#endif
  }
  for (; r < r_end; r += 2) {                                      // ragged tail
    const int row = r + half;
    const bool r_ok = row < r_end;
    const size_t rc = (size_t)(r_ok ? row : r);
    const float a = (o_ok && r_ok) ? ap[rc * dstr] : 0.f;
#pragma unroll
    for (int b = 0; b < NIB; ++b) {
      const float v = bp[b][rc * astr];
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, (c_ok[b] && r_ok) ? v : 0.f, acc[b], 0, 0, 0);
    }
    bsum += a;
  }
#pragma unroll
  for (int b = 0; b < NIB; ++b) {
    const int col = IL ? 4 * (lane & 31) + b : b * 32 + (lane & 31);
    if (col < L.in) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int oo = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (oo < L.out) dst[L.goff + (size_t)oo * L.ldw + col] = acc[b][q];
      }
    }
  }
  bsum += __shfl_xor(bsum, 32);                                    // even-row + odd-row halves
  if (lane < 32 && o_ok && L.bias) dst[L.boff + o] = bsum;
}

// The four output heads (12, 6, 1 and 1 rows of dW, 128 columns each) on v_mfma_f32_16x16x4_f32: a 16-row output block wastes a
// quarter to 15/16 of the matrix pipe where a 32-row one wastes 5/8 to 31/32, and costs half the cycles per minibatch row (8 MFMAs
// of 32 cycles per FOUR rows against 4 of 64 per TWO) -- the heads' workgroups therefore take row ranges twice as long (plan).
// Lane (g, n) = (lane >> 4, lane & 15): A operand = dZ[row 4 q + g][output n] (one dword), B operands of the eight 16-column
// blocks = act[row 4 q + g][8 n + b], b = 0..7 (two 16-byte loads: a lane group reads the whole 512-byte row); D of block b leaves
// dW[4 g + i][8 n + b], i = 0..3, in the lane. The bias gradient is the running sum of the A operands, summed over g at the end.
static __device__ __forceinline__ void wgrad_heads_body(const WgradLayer& L, const float* __restrict__ act_stash, const float* __restrict__ dz_stash,
                                                        float* __restrict__ dst, int r_begin, int r_end, int B) {
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  f32x4 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool o_ok = n < L.out;
  const char* dbase = reinterpret_cast<const char*>(dz_stash + (size_t)L.dcol * B);
  const char* abase = reinterpret_cast<const char*>(act_stash + (size_t)L.acol * B + L.aoff);
  uint32_t doff = (uint32_t)(((r_begin + g) * L.dw + (o_ok ? n : 0)) * 4);      // (outputs past the layer's width: a duplicate of output 0, never stored)
  uint32_t aoff = (uint32_t)(((r_begin + g) * L.aw + 8 * n) * 4);
  const uint32_t dstep = 4u * (uint32_t)L.dw * 4u, astep = 4u * (uint32_t)L.aw * 4u;           // bytes per group of four rows
  constexpr int U = WG_U;
  struct Ops { float av[U]; float4 b0[U], b1[U]; };
  auto issue = [&](Ops& q) {
#pragma unroll
    for (int t = 0; t < U; ++t) {
      q.av[t] = *reinterpret_cast<const float*>(dbase + (doff + t * dstep));
      q.b0[t] = *reinterpret_cast<const float4*>(abase + (aoff + t * astep));
      q.b1[t] = *reinterpret_cast<const float4*>(abase + (aoff + t * astep) + 16);
    }
    doff += U * dstep; aoff += U * astep;
  };
  auto mfmas = [&](float a, const float4& b0, const float4& b1) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.z, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.w, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.x, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.y, acc[5], 0, 0, 0);
    acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.z, acc[6], 0, 0, 0);
    acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.w, acc[7], 0, 0, 0);
    bsum += a;
  };
  auto compute = [&](const Ops& q) {
#pragma unroll
    for (int t = 0; t < U; ++t) mfmas(q.av[t], q.b0[t], q.b1[t]);
  };
  int r = r_begin;
  const int nfull = (r_end - r_begin) / (4 * U);                   // full batches of 4 U rows
  if (nfull > 0) {
    Ops A, Bq, Cq;
    issue(A);
    issue(Bq);                                  // (past the last batch these read the following rows, inside the workspace, unused)
    int i = 0;
#pragma unroll 1
    for (; i + 3 <= nfull; i += 3) {
      issue(Cq);
      __builtin_amdgcn_sched_barrier(0);
      compute(A);
      __builtin_amdgcn_sched_barrier(0);
      issue(A);
      __builtin_amdgcn_sched_barrier(0);
      compute(Bq);
      __builtin_amdgcn_sched_barrier(0);
      issue(Bq);
      __builtin_amdgcn_sched_barrier(0);
      compute(Cq);
      __builtin_amdgcn_sched_barrier(0);
      r += 12 * U;
    }
    if (i < nfull) { compute(A); r += 4 * U; ++i; }
    if (i < nfull) { compute(Bq); r += 4 * U; }
  }
  for (; r < r_end; r += 4) {                                      // ragged tail: groups of four rows, rows past the range contribute zeros
    const int row = r + g;
    const bool r_ok = row < r_end;
    const size_t rc = (size_t)(r_ok ? row : r);
    const float* dp = dz_stash + (size_t)L.dcol * B + rc * (size_t)L.dw + (o_ok ? n : 0);
    const float* bp = act_stash + (size_t)L.acol * B + L.aoff + rc * (size_t)L.aw + 8 * n;
    const float a = r_ok ? *dp : 0.f;
    float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
    if (!r_ok) { b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0; }
    mfmas(a, b0, b1);
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oo = 4 * g + i;
      if (oo < L.out) dst[L.goff + (size_t)oo * L.ldw + 8 * n + b] = acc[b][i];
    }
  }
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (lane < 16 && o_ok && L.bias) dst[L.boff + n] = bsum;
}

// Grid = (row ranges, layers), 4 waves per workgroup = the layer's four 32-row output blocks. Two workgroups per CU (three
// until round 5): co-resident waves do not overlap each other's vector-ALU work with MFMAs, but they do hide each other's memory
// latency (round 2: one workgroup per CU, 8 row pairs in flight per wave: 340 us; three: 180 us).
// Work decomposition. A 32-row output block x 32-column input block of a layer costs one MFMA per row pair; the layers
// have 2, 2, 12, 4 or 16 such blocks. With one workgroup per (layer, row range) all workgroups are resident at once (three
// per CU) and the CUs that happen to hold three 16-block workgroups set the kernel's duration (169 us at 54 % matrix-pipe
// use; dealing the workgroups out by work needs to know which workgroups share a CU and gave 2 %). Instead the layers are
// bundled into WG_NVL = 11 "virtual layers" of exactly 16 blocks each -- the nine 128 x 128-class layers; backbone (12) +
// priv0 (2) + priv2 (2, its two input blocks on different waves); the four heads (4 each) -- and every wave of every
// workgroup gets tasks worth 4 MFMAs per row pair: all workgroups are equal, whatever the placement. 168.6 -> 150.4 us.
// (Round 5: the heads' virtual layer runs wgrad_heads_body on 16-row blocks, half the cycles per row, over half as many ranges of
// twice the rows: still the same work per workgroup, 1/22 less of it in total.)
#ifndef PPO_NSPLIT
#define PPO_NSPLIT 48       // x 10 virtual layers + 24 double-length ranges of the heads' = 504 workgroups: one resident wave of TWO per CU (the workspace is sized for this many partials)
#endif
// Row ranges per layer for a minibatch of B rows. Round 5, same box, wgrad / reducer us per launch at B = 40960 | 20480 | 10240:
// 69 splits (three waves per SIMD, 2 stages) 163 / 12.3 | 92 / 11.7 | 50 / 10.4; 46 (two per SIMD, 3 stages x 4 row pairs) 155 / 10.0 |
// 84 / 9.2 | 46 / 8.4; 23 (one per SIMD) 174 / 8.5 | 88 / 6.8 | 46 / 6.3 -- a third wave buys nothing the deeper operand ring does not,
// and every split is another 0.67 MB of partials written, read back by the reducer and pushed through L2 under the next chain launch.
// With every operand load removed the kernel takes 127 us, without its stores 150 (of 157, the same box): the matrix pipe at the
// clock the chip sustains under it, not the memory system, is most of what is left. What the loads cost is not the CU's address
// unit either: two output blocks per wave with the row range halved between two wave pairs (24 bytes per lane and row pair for 8
// MFMAs instead of 20 for 4; sums handed over through LDS) measured 148.8 us against 149.5 -- not kept.
// Small minibatches (a strong-scaled shard) take half as many: the same weight-gradient time and the cheaper reduction.
// (48 since the heads' virtual layer needs half as many workgroups: 10 x 48 + 24 = 504 <= 512 resident, each 1/22 shorter than with 46.)
static int ppo_nsplit(int B) { return B <= 12288 ? PPO_NSPLIT / 2 : PPO_NSPLIT; }
struct WgradTask { int sub, ob; };                      // sub-layer (a layer or a column range of one), 32-row output block
#define WG_NVL 11
#define WG_NSUB (NLAYERS + 1)
struct WgradPlan { WgradLayer sub[WG_NSUB]; WgradTask task[WG_NVL][4][2]; int ntask[WG_NVL][4]; int nsplit, rows, nsplit_heads; };   // the heads' virtual layer (the last): nsplit_heads ranges of 2 x rows rows

#ifndef WG_OCC
#define WG_OCC 2            // resident workgroups per CU the register budget is set for
#endif
extern "C" __global__ void __launch_bounds__(PT_THREADS, WG_OCC) ppo_wgrad_kernel(WgradPlan plan, const float* __restrict__ act_stash,
                                                                         const float* __restrict__ dz_stash, float* __restrict__ wpart,
                                                                         int B, int Bs, int nparams) {
  const int vl = min((int)blockIdx.x / plan.nsplit, WG_NVL - 1), split = blockIdx.x - vl * plan.nsplit;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // uniform: the task and its layer live in SGPRs
  const int rows = vl == WG_NVL - 1 ? 2 * plan.rows : plan.rows;
  const int r_begin = min(B, split * rows), r_end = min(B, r_begin + rows);     // (an empty range still writes its zeros)
  float* dst = wpart + (size_t)split * nparams;
  if (vl == WG_NVL - 1) {                                                      // the four heads, one per wave
    wgrad_heads_body(plan.sub[plan.task[vl][wave][0].sub], act_stash, dz_stash, dst, r_begin, r_end, Bs);
    return;
  }
  const int nt = plan.ntask[vl][wave];
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const WgradTask k = plan.task[vl][wave][t];
    const WgradLayer L = plan.sub[k.sub];
    const int nib = (L.in + 31) / 32;
    if (nib == 4 && ((L.aw | L.aoff) & 3) == 0) wgrad_body<4, true>(L, act_stash, dz_stash, dst, r_begin, r_end, Bs, k.ob);   // 16-byte aligned rows
    else if (nib == 4) wgrad_body<4>(L, act_stash, dz_stash, dst, r_begin, r_end, Bs, k.ob);
    else if (nib == 3) wgrad_body<3>(L, act_stash, dz_stash, dst, r_begin, r_end, Bs, k.ob);
    else if (nib == 2) wgrad_body<2>(L, act_stash, dz_stash, dst, r_begin, r_end, Bs, k.ob);
    else wgrad_body<1>(L, act_stash, dz_stash, dst, r_begin, r_end, Bs, k.ob);
  }
}

struct RedLayer { int goff, count, nsplit; };           // nsplit: the row ranges this layer's partials come in
struct RedTable { RedLayer l[NLAYERS]; int nsplit; };
#define RED_BX ((128 * 128 + 128 + 255) / 256)            // blocks per grid row (the largest layer: 65)
#define PPO_SQ_PARTS ((NLAYERS + 1) * RED_BX)             // one partial sum of squares per block of ppo_grad_reduce_kernel
// (Round 5, tried and not kept: this reduction, the clip and the Adam step as ONE launch whose 1105 workgroups meet at a counter --
// bit-identical, but the meeting costs more than the launch it saves: with agent-scope release fences (a whole-L2 write-back per
// workgroup) +170 us per minibatch, with the partial sums stored write-through and relaxed polling +20 us at 4096 envs, +50 at 1024.)
// ONE reduction launch per minibatch. Grid rows 0 .. NLAYERS-1: grad[goff + i] = sum_{s < nsplit} part[s][goff + i] in a fixed
// order (the split-K weight-gradient partials). Grid row NLAYERS: block j < 18 + 3 sums the per-tile partials of column j of
// the std gradient (18) / the loss sums (3) -- fixed-order tree -- into out_cols[j]; the loss sums are also ADDED to
// loss_accum[0..2] (if given: the update's running totals). Every block leaves the sum of squares of the gradient entries it
// produced in sq[blockIdx.y * RED_BX + blockIdx.x] (0 if none): clip_grad_norm_'s norm without another pass over the gradient.
extern "C" __global__ void __launch_bounds__(256) ppo_grad_reduce_kernel(RedTable tab, const float* __restrict__ part, int stride, float* __restrict__ grad,
                                                                        const float* __restrict__ col_part, int nparts, const float* __restrict__ loss_part,
                                                                        float* __restrict__ out_cols, float* __restrict__ loss_accum,
                                                                        float* __restrict__ sq) {
  __shared__ float sh[256];
  float acc = 0.f;
  bool counts = false;                                     // does this thread's value belong to the clipped gradient?
  if (blockIdx.y < NLAYERS) {
    const RedLayer L = tab.l[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L.count) {
      const int p = L.goff + i;
      int s2 = 0;
      for (; s2 + 8 <= L.nsplit; s2 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(s2 + j) * stride + p];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
      for (; s2 < L.nsplit; ++s2) acc += part[(size_t)s2 * stride + p];
      grad[p] = acc;
      counts = true;
    }
    sh[threadIdx.x] = counts ? acc * acc : 0.f;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) sq[blockIdx.y * RED_BX + blockIdx.x] = sh[0];
    return;
  }
  const int j = blockIdx.x;
  if (j >= 18 + 3) { if (threadIdx.x == 0) sq[blockIdx.y * RED_BX + blockIdx.x] = 0.f; return; }
  const float* src = j < 18 ? col_part + j : loss_part + (j - 18);
  const int width = j < 18 ? 18 : 3;
  for (int t = threadIdx.x; t < nparts; t += 256) acc += src[(size_t)t * width];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_cols[j] = sh[0];
    if (j >= 18 && loss_accum) loss_accum[j - 18] += sh[0];
    sq[blockIdx.y * RED_BX + blockIdx.x] = j < 18 ? sh[0] * sh[0] : 0.f;
  }
}

// ---- gradient clip + Adam on the flat gradient (nn.utils.clip_grad_norm_ + optim.Adam.step, PPO:243-246) ----
// The partial sums of squares of a flat gradient with EXACTLY the block structure, element order and reduction tree
// ppo_grad_reduce_kernel uses for the ones it leaves behind: a gradient that went through an all-reduce (or had the entropy term
// added) gets its norm from this pass, and a 1-rank process group reproduces the group-less run bit for bit.
// grid = (RED_BX, NLAYERS + 1); sq: PPO_SQ_PARTS floats.
extern "C" __global__ void __launch_bounds__(256) ppo_sqnorm_kernel(RedTable tab, const float* __restrict__ grad, int std_off, float* __restrict__ sq) {
  __shared__ float sh[256];
  if (blockIdx.y < NLAYERS) {
    const RedLayer L = tab.l[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float g = i < L.count ? grad[L.goff + i] : 0.f;
    sh[threadIdx.x] = g * g;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) sq[blockIdx.y * RED_BX + blockIdx.x] = sh[0];
    return;
  }
  if (threadIdx.x == 0) {
    const int j = blockIdx.x;
    const float g = j < 18 ? grad[std_off + j] : 0.f;
    sq[blockIdx.y * RED_BX + blockIdx.x] = g * g;
  }
}

struct AdamTable {
  float* p[33];
  int off[34];                    // flat offset of parameter j; off[33] = total
};

// g <- s g (s = grad_scale, 1 / world_size after a SUM all-reduce), then g <- g * min(1, max_norm / (||g|| + 1e-6));
// m, v, p as torch.optim.Adam (no weight decay, no amsgrad):
// m += (g-m)(1-b1); v = v b2 + (1-b2) g g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// pack != nullptr: the new value also goes to its copies in the chain kernel's weight streams (tab: chain_scatter_table_kernel), so
// the next minibatch needs no pack launch.
extern "C" __global__ void __launch_bounds__(256) ppo_adam_kernel(AdamTable T, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                                 const float* __restrict__ part, int nparts, float max_norm, float beta1,
                                                                 float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                                                                 float* __restrict__ pack, const int* __restrict__ tab) {
  __shared__ float sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float coef = grad_scale;
  if (max_norm > 0.f) {                                    // block-uniform
    float a = 0.f;                                         // every block: the same strided partial sums, the same tree -> the same bits
    for (int b = threadIdx.x; b < nparts; b += 256) a += part[b];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    coef = grad_scale * fminf(max_norm / (sqrtf(sh[0]) * grad_scale + 1e-6f), 1.f);      // the norm of the SCALED gradient
  }
  if (i >= T.off[33]) return;
  int lo = 0, hi = 33;            // parameter j with off[j] <= i < off[j+1]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (T.off[mid] <= i) lo = mid; else hi = mid; }
  const float gi = g[i] * coef;
  g[i] = gi;
  const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
  const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float* pp = T.p[lo] + (i - T.off[lo]);
  const float pn = *pp - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  *pp = pn;
  if (pack) {
    const int d0 = tab[i], d1 = tab[T.off[33] + i];
    if (d0 >= 0) {
      pack[d0] = pn;
      if ((lo & 1) && lo < 2 * NLAYERS) {               // a bias: one copy per row lane of its lane group
#pragma unroll
        for (int mm = 1; mm < 16; ++mm) pack[d0 + 4 * mm] = pn;
      }
    }
    if (d1 >= 0) pack[d1] = pn;
  }
}

// ---- C-ABI ----------------------------------------------------------------------------------------
// Flat gradient layout: for layer l in PolicyParams order: weight [out*in] then bias [out]; then std [18];
// then 3 loss sums (surrogate, value, priv_reg; divide by 2B, 2B, B for the means).
static const int kDcol[NLAYERS] = {D_H1, D_LAT, D_BB, D_L1, D_L2, D_LEG, D_A1, D_A2, D_ARM, D_CB, D_CL1, D_CL2, D_VLEG, D_CA1, D_CA2, D_VARM};
static const int kAcol[NLAYERS] = {A_X + PT_NPROP, A_H1, A_Z, A_BB, A_L1, A_L2, A_BB, A_A1, A_A2, A_X, A_CB, A_CL1, A_CL2, A_CB, A_CA1, A_CA2};
#define PPO_WPACK_FLOATS CHAIN_PACK_FLOATS

// The equal-work plan of ppo_wgrad_kernel for this network (checked: every wave of every virtual layer gets 4 blocks).
static int make_wgrad_plan(WgradPlan& plan, RedTable& red, int B) {
  int goff[NLAYERS], off = 0;
  const int nsplit = ppo_nsplit(B), nsplit_heads = (nsplit + 1) / 2;
  for (int l = 0; l < NLAYERS; ++l) {
    const bool head = l == L_LEG4 || l == L_ARM4 || l == L_CLEG4 || l == L_CARM4;
    goff[l] = off; red.l[l] = RedLayer{off, layer_out(l) * layer_in(l) + layer_out(l), head ? nsplit_heads : nsplit}; off += layer_out(l) * layer_in(l) + layer_out(l);
  }
  auto sub = [&](int l, int col0, int ncol, int bias) {
    const int aslab = (l == L_PRIV0) ? A_X : kAcol[l];                    // priv0 reads columns 76.. of the x slab
    return WgradLayer{layer_out(l), ncol, kDcol[l], d_slab_w(kDcol[l]), aslab, a_slab_w(aslab), kAcol[l] - aslab + col0, goff[l] + col0, 0, 0,
                      layer_in(l), goff[l] + layer_out(l) * layer_in(l), bias};
  };
  for (int l = 0; l < NLAYERS; ++l) plan.sub[l] = sub(l, 0, layer_in(l), 1);
  plan.sub[L_PRIV2] = sub(L_PRIV2, 0, 32, 1);                              // priv2's two input blocks are separate tasks
  plan.sub[NLAYERS] = sub(L_PRIV2, 32, 32, 0);
  const int big[9] = {L_LEG0, L_LEG2, L_ARM0, L_ARM2, L_CBB, L_CLEG0, L_CLEG2, L_CARM0, L_CARM2};
  const int heads[4] = {L_LEG4, L_ARM4, L_CLEG4, L_CARM4};
  for (int v = 0; v < WG_NVL; ++v)
    for (int w = 0; w < 4; ++w) {
      plan.ntask[v][w] = 0;
      auto add = [&](int sb, int ob) { plan.task[v][w][plan.ntask[v][w]++] = WgradTask{sb, ob}; };
      if (v < 9) add(big[v], w);
      else if (v == 9) { add(L_BB, w); if (w < 2) add(L_PRIV0, w); else add(w == 2 ? L_PRIV2 : NLAYERS, 0); }
      else add(heads[w], 0);
      int units = 0;                                                       // 4 MFMAs per row pair for every wave
      for (int t = 0; t < plan.ntask[v][w]; ++t) {
        const WgradLayer& S = plan.sub[plan.task[v][w][t].sub];
        if (plan.task[v][w][t].ob * 32 >= S.out) return -1;
        units += (S.in + 31) / 32;
      }
      if (units != 4) return -1;
    }
  plan.nsplit = red.nsplit = nsplit;
  plan.nsplit_heads = nsplit_heads;
  plan.rows = ((B + plan.nsplit - 1) / plan.nsplit + 7) / 8 * 8;
  return off;
}

extern "C" int wbc_ppo_grad_floats(void) {
  int n = 0;
  for (int l = 0; l < NLAYERS; ++l) n += layer_out(l) * layer_in(l) + layer_out(l);
  return n + 18 + 3;
}
// rows of a stash slab: whole tiles (every tile row is stored), then a multiple of 64
static int ppo_slab_rows(int B) { return ((B + R16 - 1) / R16 * R16 + 63) & ~63; }
extern "C" int wbc_ppo_num_splits(void) { return PPO_NSPLIT; }
// floats of workspace for a minibatch of B rows
extern "C" size_t wbc_ppo_workspace_floats(int B) {
  const size_t tiles = (size_t)(B + R16 - 1) / R16;
  return (size_t)ppo_slab_rows(B) * (A_LD + D_LD) + tiles * (18 + 3) + (size_t)PPO_NSPLIT * (size_t)wbc_ppo_grad_floats() + (size_t)PPO_WPACK_FLOATS + 4 +
         PPO_SQ_PARTS;
}
// where wbc_ppo_minibatch_grad leaves the partial sums of squares of the gradient it produced (floats from the start of a workspace
// for B rows), for wbc_ppo_clip_adam's sq_partials argument
static size_t ppo_sq_offset(int B) { return wbc_ppo_workspace_floats(B) - PPO_SQ_PARTS; }
extern "C" size_t wbc_ppo_sq_partials_offset(int B) { return ppo_sq_offset(B); }

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
static float* ppo_wpack_of(float* workspace, int B) {
  const int ng = wbc_ppo_grad_floats();
  const size_t tiles16w = (size_t)(B + R16 - 1) / R16;
  float* wpart = workspace + (size_t)ppo_slab_rows(B) * (A_LD + D_LD) + tiles16w * (18 + 3);
  return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(wpart + (size_t)PPO_NSPLIT * ng) + 15) & ~(uintptr_t)15);   // float4 loads
}

// Which workspaces hold CURRENT weight streams: workspace -> (B, the parameter table's first pointer) as left by the last
// wbc_ppo_minibatch_grad (which packs) / wbc_ppo_clip_adam_packed (which keeps them current). wbc_ppo_minibatch_grad_packed trusts the
// streams only when its own (workspace, B, params[0]) matches the record; a different B, a reallocated workspace or another parameter
// set falls back to packing afresh instead of computing gradients against stale weights. (Writes to the parameters by anything but
// wbc_ppo_clip_adam_packed -- load_state_dict, a broadcast -- must be followed by a plain wbc_ppo_minibatch_grad or wbc_ppo_pack_invalidate.)
#include <mutex>
#include <unordered_map>
struct PackRecord { int B; const void* p0; };
static std::mutex g_pack_mu;
static std::unordered_map<const void*, PackRecord> g_pack_records;
static void pack_record_set(const void* ws, int B, const void* p0) { std::lock_guard<std::mutex> g(g_pack_mu); g_pack_records[ws] = PackRecord{B, p0}; }
static bool pack_record_ok(const void* ws, int B, const void* p0) {
  std::lock_guard<std::mutex> g(g_pack_mu);
  auto it = g_pack_records.find(ws);
  return it != g_pack_records.end() && it->second.B == B && it->second.p0 == p0;
}
static void pack_record_drop_params(const void* p0) {       // a step that moved these parameters without touching the streams
  std::lock_guard<std::mutex> g(g_pack_mu);
  for (auto it = g_pack_records.begin(); it != g_pack_records.end();) it = (it->second.p0 == p0) ? g_pack_records.erase(it) : std::next(it);
}
extern "C" int wbc_ppo_pack_invalidate(const float* workspace) {
  std::lock_guard<std::mutex> g(g_pack_mu);
  if (workspace) g_pack_records.erase(workspace); else g_pack_records.clear();
  return 0;
}

static int ppo_minibatch_grad_impl(const void* const* params, const float* obs, const float* actions, const float* old_values,
                                   const float* advantages, const float* returns, const float* old_logp, const float* hist_latent,
                                   const int64_t* idx, int B, float clip, float value_coef, float mixing, float roa_coef,
                                   int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream, bool weights_packed) {
  StreamDeviceGuard sdg(stream);
  PolicyParams P;
  if (!params || !obs || !actions || !old_values || !advantages || !returns || !old_logp || !hist_latent || !idx || !workspace || !grad ||
      B <= 0 || fill_params(params, &P))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const int ng = wbc_ppo_grad_floats();
  const int Bs = ppo_slab_rows(B);          // rows per stash slab
  if ((long long)Bs * A_LD * 4 >= (1ll << 31)) return -3;          // the update kernel addresses a stash with 32-bit buffer offsets (B < 338 000 rows)
  float* act_stash = workspace;
  float* dz_stash = act_stash + (size_t)Bs * A_LD;
  float* dstd_partial = dz_stash + (size_t)Bs * D_LD;
  const size_t tiles16w = (size_t)(B + R16 - 1) / R16;                // the workspace holds one partial per 16-row tile
  float* loss_partial = dstd_partial + tiles16w * 18;
  float* wpart = loss_partial + tiles16w * 3;
  float* wpack = ppo_wpack_of(workspace, B);
  PpoBatch Bt{obs, actions, old_values, advantages, returns, old_logp, hist_latent, idx, B, Bs, clip, value_coef, mixing, roa_coef, use_clipped_value_loss};
  const ChainStreams& S = chain_streams();
  const int tiles16 = (B + 15) / 16;
  if (weights_packed && !pack_record_ok(workspace, B, params[0])) weights_packed = false;     // stale or unknown streams: pack afresh
  pack_record_set(workspace, B, params[0]);
  if (!weights_packed)
    hipLaunchKernelGGL(chain_pack_kernel, dim3(((S.nelem[0] > S.nelem[1] ? S.nelem[0] : S.nelem[1]) * 64 + 255) / 256, 2), dim3(256), 0, st, P, S, wpack);
  hipLaunchKernelGGL(ppo_chain_kernel, dim3((2 * tiles16 + CH_WG / 64 - 1) / (CH_WG / 64)), dim3(CH_WG), 0, st, wpack, S.base[1], S.nelem[0] * 1024, S.nelem[1] * 1024, Bt,
                     P.std, act_stash, dz_stash, dstd_partial, loss_partial, tiles16);
  WgradPlan plan;
  RedTable red;
  const int off = make_wgrad_plan(plan, red, B);
  if (off < 0) return -2;
  hipLaunchKernelGGL(ppo_wgrad_kernel, dim3((WG_NVL - 1) * plan.nsplit + plan.nsplit_heads), dim3(PT_THREADS), 0, st, plan, act_stash, dz_stash, wpart, B, Bs, ng);
  hipLaunchKernelGGL(ppo_grad_reduce_kernel, dim3(RED_BX, NLAYERS + 1), dim3(256), 0, st, red, wpart, ng, grad, dstd_partial, tiles16, loss_partial,
                     grad + off, loss_accum, workspace + ppo_sq_offset(B));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_ppo_minibatch_grad(const void* const* params, const float* obs, const float* actions, const float* old_values,
                                      const float* advantages, const float* returns, const float* old_logp, const float* hist_latent,
                                      const int64_t* idx, int B, float clip, float value_coef, float mixing, float roa_coef,
                                      int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream) {
  return ppo_minibatch_grad_impl(params, obs, actions, old_values, advantages, returns, old_logp, hist_latent, idx, B, clip, value_coef, mixing, roa_coef,
                                 use_clipped_value_loss, workspace, grad, loss_accum, stream, false);
}
// The same call without the weight-pack launch: `workspace` still holds the weight streams a wbc_ppo_minibatch_grad call for the same
// B packed into it, and every change of the parameters since then was a wbc_ppo_clip_adam_packed(..., workspace, B) step.
extern "C" int wbc_ppo_minibatch_grad_packed(const void* const* params, const float* obs, const float* actions, const float* old_values,
                                             const float* advantages, const float* returns, const float* old_logp, const float* hist_latent,
                                             const int64_t* idx, int B, float clip, float value_coef, float mixing, float roa_coef,
                                             int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream) {
  return ppo_minibatch_grad_impl(params, obs, actions, old_values, advantages, returns, old_logp, hist_latent, idx, B, clip, value_coef, mixing, roa_coef,
                                 use_clipped_value_loss, workspace, grad, loss_accum, stream, true);
}

// clip_grad_norm_(params, max_norm) followed by Adam.step() for the 33 parameters of `params`, whose gradients are
// grad[0 : wbc_ppo_grad_floats()-3] in the layout above; exp_avg / exp_avg_sq: flat state in the same layout.
// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) (computed by the caller in double, as torch does).
// max_norm <= 0: no clipping. workspace: >= wbc_ppo_clip_adam_workspace_floats() floats.
extern "C" int wbc_ppo_clip_adam_workspace_floats(void) { return PPO_SQ_PARTS; }
// The process's scatter table (device memory of the current device; built on first use): where each parameter's copies sit in the
// chain kernel's weight streams.
static const int* ppo_scatter_table(hipStream_t st, int nparam) {
  // one table per device, built once under a lock and never freed while the process lives (a kernel in flight on another device
  // may still be reading its table)
  static std::mutex mu;
  static int* tabs[64] = {nullptr};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> g(mu);
  if (tabs[dev]) return tabs[dev];
  int* t = nullptr;
  int* clash = nullptr;
  if (hipMalloc(&t, sizeof(int) * 2 * (size_t)nparam) != hipSuccess || hipMalloc(&clash, sizeof(int)) != hipSuccess) return nullptr;
  (void)hipMemsetAsync(t, 0xFF, sizeof(int) * 2 * (size_t)nparam, st);
  (void)hipMemsetAsync(clash, 0, sizeof(int), st);
  ChainParamOffsets PO;
  int off = 0;
  for (int l = 0; l < NLAYERS; ++l) { PO.off[2 * l] = off; off += layer_out(l) * layer_in(l); PO.off[2 * l + 1] = off; off += layer_out(l); }
  const ChainStreams& S = chain_streams();
  hipLaunchKernelGGL(chain_scatter_table_kernel, dim3(((S.nelem[0] > S.nelem[1] ? S.nelem[0] : S.nelem[1]) * 64 + 255) / 256, 2), dim3(256), 0, st, S, PO, nparam, t, clash);
  int h = -1;
  const hipError_t e1 = hipMemcpyAsync(&h, clash, sizeof(int), hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  if (e1 != hipSuccess || e2 != hipSuccess || h != 0) {
    fprintf(stderr, "wbc_ppo: scatter table not built (copy %d, sync %d, parameters with two copies in one stream kind: %d)\n", (int)e1, (int)e2, h);
    (void)hipFree(t); (void)hipFree(clash);
    return nullptr;
  }
  (void)hipFree(clash);
  tabs[dev] = t;
  return t;
}

static int ppo_clip_adam_impl(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                              float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, const float* sq_partials, float* workspace,
                              void* stream, float* mb_workspace, int B) {
  StreamDeviceGuard sdg(stream);
  if (!params || !grad || !exp_avg || !exp_avg_sq || !workspace || !(grad_scale > 0.f)) return -1;
  AdamTable T;
  int off = 0, j = 0;
  for (int l = 0; l < NLAYERS; ++l) {
    if (!params[2 * l] || !params[2 * l + 1]) return -1;
    T.p[j] = (float*)params[2 * l]; T.off[j++] = off; off += layer_out(l) * layer_in(l);
    T.p[j] = (float*)params[2 * l + 1]; T.off[j++] = off; off += layer_out(l);
  }
  if (!params[32]) return -1;
  T.p[32] = (float*)params[32]; T.off[32] = off; off += 18; T.off[33] = off;
  hipStream_t st = (hipStream_t)stream;
  // the norm: from the partials wbc_ppo_minibatch_grad left (the gradient is still the one it produced), or the same partials
  // recomputed from grad
  const bool have = sq_partials != nullptr;
  if (max_norm > 0.f && !have) {
    RedTable red;
    int o = 0;
    for (int l = 0; l < NLAYERS; ++l) { red.l[l] = RedLayer{o, layer_out(l) * layer_in(l) + layer_out(l), 0}; o += layer_out(l) * layer_in(l) + layer_out(l); }
    red.nsplit = 0;
    hipLaunchKernelGGL(ppo_sqnorm_kernel, dim3(RED_BX, NLAYERS + 1), dim3(256), 0, st, red, grad, o, workspace);
  }
  float* pack = nullptr;
  const int* tab = nullptr;
  if (mb_workspace && pack_record_ok(mb_workspace, B, params[0])) {      // (streams nobody packed for these parameters are not written into)
    if (B <= 0) return -1;
    tab = ppo_scatter_table(st, off);
    if (!tab) return -4;
    pack = ppo_wpack_of(mb_workspace, B);
  } else {
    pack_record_drop_params(params[0]);                                  // this step moves the parameters past every stream packed from them
  }
  hipLaunchKernelGGL(ppo_adam_kernel, dim3((off + 255) / 256), dim3(256), 0, st, T, grad, exp_avg, exp_avg_sq, have ? sq_partials : workspace,
                     PPO_SQ_PARTS, max_norm, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, pack, tab);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int wbc_ppo_clip_adam(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                                 float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, const float* sq_partials, float* workspace,
                                 void* stream) {
  return ppo_clip_adam_impl(params, grad, exp_avg, exp_avg_sq, max_norm, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, sq_partials, workspace, stream, nullptr, 0);
}
// The same step, and the new weights also written to their places in the weight streams inside `mb_workspace` (the workspace of
// wbc_ppo_minibatch_grad for B rows): the next wbc_ppo_minibatch_grad_packed on that workspace needs no pack launch.
extern "C" int wbc_ppo_clip_adam_packed(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                                        float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, const float* sq_partials, float* workspace,
                                        float* mb_workspace, int B, void* stream) {
  if (!mb_workspace) return -1;
  return ppo_clip_adam_impl(params, grad, exp_avg, exp_avg_sq, max_norm, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, sq_partials, workspace, stream, mb_workspace, B);
}
