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
//   3. ppo_grad_reduce_kernel: fixed-order sums of the split partials into the flat gradient buffer, the std-gradient and
//      loss columns, and the partial sums of squares the gradient clip needs.
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
#ifdef WBC_PPO_TIMING
#define PSTAMP(i) do { if (g_ppo_dbg && blockIdx.x == 0 && threadIdx.x == 0) g_ppo_dbg[i] = clock64(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
extern "C" void wbc_debug_set_ppo_timing(void* dev_buf) {
  long long* p = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ppo_dbg), &p, sizeof(p));
#ifdef WBC_PPO_TIMING
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_dbg), &p, sizeof(p));
#endif
}

#include "wbc_ppo_chain.h"

// LDS plan of the update kernel: x[32][101] (re-used as g[32][41] -- output grads dmu 18, dv 2, dlat 20 -- once the forward
// is done), TWO activation buffers and outv: 48.6 KB, three workgroups per CU. The third buffer the chain would like (the
// backbone output feeds two heads) is replaced by a reload from the activation stash (forward) and by keeping the first
// head's contribution in the accumulator registers until the second head adds to it (backward).
#define Q_X 0
#define Q_A0 (PT_ROWS * 101)
#define Q_A1 (Q_A0 + PT_ROWS * LDA)
#define Q_OUTV (Q_A1 + PT_ROWS * LDA)
#define Q_END (Q_OUTV + PT_ROWS * 21)
#define Q_G Q_X
static_assert(PT_ROWS * 41 <= PT_ROWS * 101, "g fits in x");
static_assert(Q_END * 4 * 3 <= 160 * 1024, "three workgroups per CU");

// ---- backward stages ------------------------------------------------------------------------------------
// stage = [pre-step] ; buf <- buf * act'(A) (also to the DZ stash) ; [out (+)= buf * W on MFMA]
enum { PRE_NONE = 0, PRE_OUTER_V0, PRE_OUTER_V1, PRE_COPY_LEG, PRE_COPY_ARM, PRE_LATENT };
struct BwdDesc {
  int pre; const float* wvec; int src_off;       // pre-step; for PRE_OUTER_*: the [128] last-layer weight row; for PRE_LATENT: dA_z buffer
  int buf_off, n, act, acol, dcol, aw, dw;        // activation-derivative pass over buf[32, n]; stash slabs (start, width)
  int has_mma, woffT, nblkT, out_dim, in_dim, out_off;   // out[32, in_dim] = buf[32, out_dim] * W[out_dim, in_dim] (transposed pack)
  int save_out, add_saved;      // keep the product in the accumulator registers instead of LDS / start from the kept product
};
#define NBWD 14
struct BwdTable { BwdDesc s[NBWD]; };

static BwdTable make_bwd_table(const PolicyParams& P) {
  BwdTable t;
  int i = 0;
  // lw = layer whose weight the stage multiplies by (-1: no GEMM); mode: 0 product -> LDS `out`, 1 product kept in registers,
  // 2 kept product + this product -> LDS `out`
  auto add = [&](int pre, const float* wvec, int src, int buf, int n, int act, int acol, int dcol, int lw, int out, int mode) {
    t.s[i++] = BwdDesc{pre, wvec, src, buf, n, act, acol, dcol, a_slab_w(acol), d_slab_w(dcol), lw >= 0, lw >= 0 ? layer_packT_off(lw) : 0, lw >= 0 ? layer_nblkT(lw) : 0,
                       lw >= 0 ? layer_out(lw) : 0, lw >= 0 ? layer_in(lw) : 0, out, mode == 1, mode == 2};
  };
  // critic
  add(PRE_OUTER_V0, P.cleg4_w, 0, Q_A0, 128, ACT_ELU, A_CL2, D_CL2, L_CLEG2, Q_A1, 0);
  add(PRE_NONE, nullptr, 0, Q_A1, 128, ACT_ELU, A_CL1, D_CL1, L_CLEG0, 0, 1);
  add(PRE_OUTER_V1, P.carm4_w, 0, Q_A0, 128, ACT_ELU, A_CA2, D_CA2, L_CARM2, Q_A1, 0);
  add(PRE_NONE, nullptr, 0, Q_A1, 128, ACT_ELU, A_CA1, D_CA1, L_CARM0, Q_A0, 2);
  add(PRE_NONE, nullptr, 0, Q_A0, 128, ACT_ELU, A_CB, D_CB, -1, 0, 0);
  // actor
  add(PRE_COPY_LEG, nullptr, 0, Q_A0, PT_NLEG, ACT_TANH, A_LEG, D_LEG, L_LEG4, Q_A1, 0);
  add(PRE_NONE, nullptr, 0, Q_A1, 128, ACT_ELU, A_L2, D_L2, L_LEG2, Q_A0, 0);
  add(PRE_NONE, nullptr, 0, Q_A0, 128, ACT_ELU, A_L1, D_L1, L_LEG0, 0, 1);
  add(PRE_COPY_ARM, nullptr, 0, Q_A0, PT_NARM, ACT_TANH, A_ARM, D_ARM, L_ARM4, Q_A1, 0);
  add(PRE_NONE, nullptr, 0, Q_A1, 128, ACT_ELU, A_A2, D_A2, L_ARM2, Q_A0, 0);
  add(PRE_NONE, nullptr, 0, Q_A0, 128, ACT_ELU, A_A1, D_A1, L_ARM0, Q_A1, 2);
  add(PRE_NONE, nullptr, 0, Q_A1, 128, ACT_ELU, A_BB, D_BB, L_BB, Q_A0, 0);
  add(PRE_LATENT, nullptr, Q_A0, Q_A1, 20, ACT_ELU, A_LAT, D_LAT, L_PRIV2, Q_A0, 0);
  add(PRE_NONE, nullptr, 0, Q_A0, 64, ACT_ELU, A_H1, D_H1, -1, 0, 0);
  return t;
}

template <int ACTV>
static __device__ __forceinline__ float act_deriv(float a) {
  return (ACTV == ACT_ELU) ? (a > 0.f ? 1.f : a + 1.f) : 1.f - a * a;
}

// ==== 16-row tiles =======================================================================================================
// The same kernel on v_mfma_f32_16x16x4_f32 tiles (wbc_mlp.h). A workgroup takes PPO_MB blocks of 16 rows that share every
// operand set (weight traffic from L2, descriptor loads, barriers and address arithmetic are per layer and workgroup, not
// per row). LDS: three activation buffers and outv; x = obs[idx, :100] is gathered into the third buffer for the actor's
// first layer and gathered again (obs is an input: no synchronisation) for the critic's, and that buffer holds the output
// gradients g afterwards. Stash layouts, loss phase and stage order are those of the 32-row kernel.
#ifndef PPO_MB
#define PPO_MB 2
#endif
#define HROWS (16 * PPO_MB)
#define H_A0 0
#define H_A1 (H_A0 + HROWS * LD16)
#define H_A2 (H_A1 + HROWS * LD16)
#define H_OUTV (H_A2 + HROWS * LD16)
#define H_END (H_OUTV + HROWS * 21)
#define H_X H_A2
#define H_G H_A2
#ifndef PPO16_OCC
#define PPO16_OCC 3
#endif
#define PPO16_CBB_POS 9          // position of the critic backbone in the forward order: x is gathered again in front of it
static_assert(HROWS <= 64 && HROWS * 41 <= HROWS * LD16, "g fits in the x buffer; loss phase: at least 4 threads per row");
static_assert(H_END * 4 * PPO16_OCC <= 160 * 1024, "LDS");

// Forward order: with three activation buffers both backbone outputs stay in LDS for their second head (no stash reload)
// and the forward is a plain loop over the 16 layers; the proprio block is copied next to where priv2 puts the latent
// while x is loaded.
static Tab16 make_fwd_table16(const int* stash_cols) {
  Tab16 t;
  int i = 0;
  auto layer = [&](int l, int in_off, int out_off) {
    t.l[i++] = make_desc16(l, in_off, out_off, stash_cols[l], stash_cols[l] >= 0 ? a_slab_w(stash_cols[l]) : 0);
  };
  layer(L_PRIV0, H_X + PT_NPROP, H_A0);
  layer(L_PRIV2, H_A0, H_A1 + PT_NPROP);
  layer(L_BB, H_A1, H_A0);
  layer(L_LEG0, H_A0, H_A1);
  layer(L_LEG2, H_A1, H_A2);
  layer(L_LEG4, H_A2, H_OUTV);
  layer(L_ARM0, H_A0, H_A1);
  layer(L_ARM2, H_A1, H_A2);
  layer(L_ARM4, H_A2, H_OUTV + PT_NLEG);
  layer(L_CBB, H_X, H_A0);
  layer(L_CLEG0, H_A0, H_A1);
  layer(L_CLEG2, H_A1, H_A2);
  layer(L_CLEG4, H_A2, H_OUTV + 18);
  layer(L_CARM0, H_A0, H_A1);
  layer(L_CARM2, H_A1, H_A2);
  layer(L_CARM4, H_A2, H_OUTV + 19);
  return t;
}

// the 32-row stage table with the 16-row LDS offsets and transposed-pack16 offsets
static BwdTable make_bwd_table16(const PolicyParams& P) {
  BwdTable t = make_bwd_table(P);
  for (int i = 0; i < NBWD; ++i) {
    BwdDesc& d = t.s[i];
    auto map = [](int q) { return q == Q_A0 ? H_A0 : (q == Q_A1 ? H_A1 : q); };
    d.buf_off = map(d.buf_off); d.out_off = map(d.out_off); d.src_off = map(d.src_off);
  }
  const int lw[NBWD] = {L_CLEG2, L_CLEG0, L_CARM2, L_CARM0, -1, L_LEG4, L_LEG2, L_LEG0, L_ARM4, L_ARM2, L_ARM0, L_BB, L_PRIV2, -1};
  for (int i = 0; i < NBWD; ++i) t.s[i].woffT = lw[i] >= 0 ? WPACK16_FWD_FLOATS + l16T_sum(lw[i] - 1) : WPACK16_FWD_FLOATS;
  return t;
}

static __device__ __forceinline__ void bwd_load16(float (&w)[66], const BwdDesc& d, const float* __restrict__ blob) {
  const int wave = wave_role();
  const int nblk = d.has_mma ? d.nblkT : 1;
  load_ops16(w, blob + d.woffT, (d.out_dim + 31) >> 5, wave < nblk ? wave : 0);
}

// Thread t owns columns c, c+1 with c = 2 (t & 63) of rows (t >> 6) + 4 j, j < 4 PPO_MB. Lanes past the stage's width n
// work on its last column pair as well (same wave, same instruction, same values as the owning lane): no lane predicate, so
// the stash accesses are unconditional and the compiler's s_waitcnt vmcnt counts stay exact.
struct BwdFetch16 { float2 a[4 * PPO_MB]; float wv; };
static __device__ __forceinline__ void bwd_fetch16(BwdFetch16& f, const BwdDesc& d, const float* __restrict__ act_stash, int row0, int Bs) {
  const int tid = threadIdx.x;
  const int c = min((tid & 63) * 2, d.n - 2), rb = tid >> 6;
  const char* base = reinterpret_cast<const char*>(act_stash + (size_t)d.acol * Bs);          // scalar base + 32-bit lane offset (bytes)
  const uint32_t off = (uint32_t)((row0 + rb) * d.aw + c) * 4u, step = (uint32_t)(16 * d.aw);
#pragma unroll
  for (int j = 0; j < 4 * PPO_MB; ++j) f.a[j] = *reinterpret_cast<const float2*>(base + (off + j * step));    // padded slabs: every tile row exists
  const float* wv = (d.pre == PRE_OUTER_V0 || d.pre == PRE_OUTER_V1) ? d.wvec : act_stash;
  f.wv = wv[tid & 127];
}

static __device__ __forceinline__ void bwd_pre_act16(const BwdDesc& d, const BwdFetch16& f, float* smem, float* __restrict__ dz_stash,
                                                     int row0, int num_rows, int Bs) {
  const int tid = threadIdx.x;
  float* buf = smem + d.buf_off;
  const float* g = smem + H_G;
  if (d.pre == PRE_OUTER_V0 || d.pre == PRE_OUTER_V1) {
    const int gi = (d.pre == PRE_OUTER_V0) ? 18 : 19;
    const int c = tid & 127, r0 = tid >> 7;
#pragma unroll
    for (int k = 0; k < 8 * PPO_MB; ++k) buf[(r0 + 2 * k) * LD16 + c] = g[(r0 + 2 * k) * 41 + gi] * f.wv;
    LBAR();
  } else if (d.pre == PRE_COPY_LEG || d.pre == PRE_COPY_ARM) {
    // the GEMM's k runs over the head's outputs in chunks of 32: the columns up to 32 must be finite (zeros)
    const int n = d.n, go = (d.pre == PRE_COPY_LEG) ? 0 : PT_NLEG;
    const int c = tid & 15;                                          // 16 threads per row, n <= 12
#pragma unroll
    for (int m = 0; m < PPO_MB; ++m) {
      const int r = (tid >> 4) + 16 * m;
      buf[r * LD16 + c] = c < n ? g[r * 41 + go + c] : 0.f;
      buf[r * LD16 + 16 + c] = 0.f;
    }
    LBAR();
  } else if (d.pre == PRE_LATENT) {
    const float* src = smem + d.src_off;
    const int c = tid & 15;
#pragma unroll
    for (int m = 0; m < PPO_MB; ++m) {
      const int r = (tid >> 4) + 16 * m;
      buf[r * LD16 + c] = src[r * LD16 + PT_NPROP + c] + g[r * 41 + 20 + c];
      buf[r * LD16 + 16 + c] = c < 4 ? src[r * LD16 + PT_NPROP + 16 + c] + g[r * 41 + 36 + c] : 0.f;      // k padding: zeros up to column 32
    }
    LBAR();
  }
  {
    const int c = min((tid & 63) * 2, d.n - 2), rb = tid >> 6;
    float* bp = buf + rb * LD16 + c;
    char* dzp = reinterpret_cast<char*>(dz_stash + (size_t)d.dcol * Bs);
    const uint32_t off = (uint32_t)((row0 + rb) * d.dw + c) * 4u, step = (uint32_t)(16 * d.dw);
    const bool elu = d.act == ACT_ELU;
#pragma unroll
    for (int j = 0; j < 4 * PPO_MB; ++j) {
      const bool ok = row0 + rb + 4 * j < num_rows;
      const float dx = elu ? act_deriv<ACT_ELU>(f.a[j].x) : act_deriv<ACT_TANH>(f.a[j].x);
      const float dy = elu ? act_deriv<ACT_ELU>(f.a[j].y) : act_deriv<ACT_TANH>(f.a[j].y);
      float2 v = *reinterpret_cast<const float2*>(bp + 4 * j * LD16);
      v.x *= dx; v.y *= dy;
      if (!ok) v = make_float2(0.f, 0.f);
      *reinterpret_cast<float2*>(bp + 4 * j * LD16) = v;
      *reinterpret_cast<float2*>(dzp + (off + j * step)) = v;
    }
    LBAR();
  }
}

extern "C" __global__ void __launch_bounds__(PT_THREADS, PPO16_OCC) ppo_fwd_bwd16_kernel(PolicyParams P, Tab16 FT, BwdTable BT,
                                                                                       const float* __restrict__ blob, PpoBatch Bt,
                                                                                       float* __restrict__ act_stash, float* __restrict__ dz_stash,
                                                                                       float* __restrict__ dstd_partial, float* __restrict__ loss_partial) {
  __shared__ __attribute__((aligned(16))) float smem[H_END];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, row0 = tile * HROWS, B = Bt.B, Bs = Bt.Bs;
  const float* bias = blob + WPACK16_BIAS_OFF;
  PSTAMP(0);
  // gather obs[idx, :100] into x (and the x slab and the proprio part of the z slab of the stash); the columns 100..127
  // (k padding of the first layers) zero; first = false: the second gather, LDS only
  auto gather_x = [&](bool first) {
#pragma unroll
    for (int k = 0; k < 2 * PPO_MB; ++k) {
      const int e4 = tid + k * PT_THREADS;
      const int r = e4 >> 5, c = (e4 & 31) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < 100) {
        v = *reinterpret_cast<const float4*>(Bt.obs + (size_t)Bt.idx[min(row0 + r, B - 1)] * PT_NOBS + c);
        if (row0 + r >= B) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (first) {
          *reinterpret_cast<float4*>(act_stash + sidx(Bs, A_X, 100, row0 + r, c)) = v;
          if (c < PT_NPROP) *reinterpret_cast<float4*>(act_stash + sidx(Bs, A_Z, 100, row0 + r, c)) = v;
        }
      }
      *reinterpret_cast<float4*>(smem + H_X + r * LD16 + c) = v;
      if (first && c < PT_NPROP) *reinterpret_cast<float4*>(smem + H_A1 + r * LD16 + c) = v;      // proprio block of the backbone's input
    }
    LBAR();
  };
  gather_x(true);
  PSTAMP(1);
  {
    float w[66];
    load16(w, FT.l[0], blob, bias);
#pragma unroll 1
    for (int i = 0; i < NLAYERS; ++i) {
      const int nx = i + 1 < NLAYERS ? i + 1 : i;        // (the last layer re-requests its own operands: no conditional refill)
      if (i == PPO16_CBB_POS) gather_x(false);           // (the previous layer's barrier: nobody reads this buffer any more)
      const int pv = i > 0 ? i - 1 : 0;
      run16<PPO_MB>(w, FT.l[i], smem, act_stash, row0, Bs, FT.l[pv], i > 0 && FT.l[pv].scol >= 0, [&]() { load16(w, FT.l[nx], blob, bias); }, i);
    }
    if (FT.l[NLAYERS - 1].scol >= 0) stash_copy16<PPO_MB>(FT.l[NLAYERS - 1], smem, act_stash, row0, Bs);
  }
  PSTAMP(2);
  __threadfence_block();
  __syncthreads();
  // latent part of the z = [prop, latent] slab (the backbone's input, for its weight gradient)
  for (int e = tid; e < HROWS * 5; e += PT_THREADS) {
    const int r = e / 5, c = (e - r * 5) * 4;
    *reinterpret_cast<float4*>(act_stash + sidx(Bs, A_Z, 100, row0 + r, PT_NPROP + c)) =
        *reinterpret_cast<const float4*>(act_stash + sidx(Bs, A_LAT, 20, row0 + r, c));
  }
  LBAR();
  const float* outv = smem + H_OUTV;
  float* gbuf = smem + H_G;
  PSTAMP(3);
  // Losses and output gradients: TPR = 256 / rows threads per row (the action and latent dimensions are dealt out to
  // them), all four waves busy; per-tile sums through a [4 waves][24] scratch in a1 (not written before stage 0's GEMM).
  {
    constexpr int TPR = (PT_THREADS / HROWS >= 16) ? 16 : ((PT_THREADS / HROWS >= 8) ? 8 : 4);      // power of two: butterflies
    constexpr int NJ = (18 + TPR - 1) / TPR, NK = (20 + TPR - 1) / TPR;
    const int r = min(tid / TPR, HROWS - 1), q = tid % TPR, lane = tid & 63;
    const bool live = tid < HROWS * TPR;                       // (48-row tiles: 192 of the 256 threads)
    const bool valid = live && row0 + r < B;
    const size_t src = (size_t)Bt.idx[min(row0 + r, B - 1)];
    const float inv2B = 1.f / (2.f * (float)B), invB = 1.f / (float)B;
    float lp[2] = {0.f, 0.f}, dj[NJ], sdj[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const int j = q + TPR * t, jj = j < 18 ? j : 0;
      const float mu = outv[r * 21 + jj], sd = P.std[jj];
      const float d = Bt.actions[src * 18 + jj] - mu;
      const float term = -(d * d) / (2.f * sd * sd) - logf(sd) - 0.91893853320467274178f;
      dj[t] = d; sdj[t] = sd;
      if (j < PT_NLEG) lp[0] += term; else if (j < 18) lp[1] += term;
    }
#pragma unroll
    for (int off = 1; off < TPR; off <<= 1) { lp[0] += __shfl_xor(lp[0], off); lp[1] += __shfl_xor(lp[1], off); }
    const float adv0 = Bt.advantages[src * 2], adv1 = Bt.advantages[src * 2 + 1];
    const float mixed[2] = {adv0 + Bt.mixing * adv1, adv1 + Bt.mixing * adv0};               // PPO:199-201
    float dlp[2], surr = 0.f, vls = 0.f, preg = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {                                                               // (every thread of the row: cheap)
      const float ratio = expf(lp[c] - Bt.old_logp[src * 2 + c]);                               // PPO:202
      const float rc = fminf(fmaxf(ratio, 1.f - Bt.clip), 1.f + Bt.clip);
      const float s1 = -mixed[c] * ratio, s2 = -mixed[c] * rc;                                  // PPO:203-205
      surr += fmaxf(s1, s2);
      const bool inside = (ratio >= 1.f - Bt.clip) && (ratio <= 1.f + Bt.clip);
      const float dr = (inside || s1 > s2) ? -mixed[c] : 0.f;
      dlp[c] = valid ? inv2B * dr * ratio : 0.f;
      const float v = outv[r * 21 + 18 + c], ov = Bt.old_values[src * 2 + c], R = Bt.returns[src * 2 + c];    // PPO:209-216
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
      if (q == c && live) gbuf[r * 41 + 18 + c] = valid ? Bt.value_coef * inv2B * dv : 0.f;
    }
    if (q != 0 || !valid) { surr = 0.f; vls = 0.f; }                                            // one thread per row carries the row's loss terms
    float dsd[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
      const int j = q + TPR * t;
      const float gl = dlp[j < PT_NLEG ? 0 : 1], d = dj[t], sd = sdj[t];
      if (j < 18 && live) gbuf[r * 41 + j] = gl * d / (sd * sd);
      dsd[t] = (j < 18 && live) ? gl * (d * d / (sd * sd * sd) - 1.f / sd) : 0.f;
    }
    float dl[NK], nrm = 0.f;                                                                    // ROA regulariser PPO:174-179
#pragma unroll
    for (int t = 0; t < NK; ++t) {
      const int k = q + TPR * t, kk = k < 20 ? k : 0;
      dl[t] = act_stash[sidx(Bs, A_LAT, 20, row0 + r, kk)] - Bt.hist_latent[src * 20 + kk];
      if (k >= 20) dl[t] = 0.f;
      nrm += dl[t] * dl[t];
    }
#pragma unroll
    for (int off = 1; off < TPR; off <<= 1) nrm += __shfl_xor(nrm, off);
    nrm = sqrtf(nrm);
    if (q == 0 && valid) preg = nrm;
    const float sc = (nrm > 0.f && valid) ? Bt.roa_coef * invB / nrm : 0.f;
#pragma unroll
    for (int t = 0; t < NK; ++t) {
      const int k = q + TPR * t;
      if (k < 20 && live) gbuf[r * 41 + 20 + k] = sc * dl[t];
    }
    // sums over the wave's rows (lanes with the same q), then over the four waves
#pragma unroll
    for (int off = TPR; off < 64; off <<= 1) {
      surr += __shfl_xor(surr, off); vls += __shfl_xor(vls, off); preg += __shfl_xor(preg, off);
#pragma unroll
      for (int t = 0; t < NJ; ++t) dsd[t] += __shfl_xor(dsd[t], off);
    }
    float* red = smem + H_A1;
    if (lane < TPR) {
#pragma unroll
      for (int t = 0; t < NJ; ++t) if (q + TPR * t < 18) red[(tid >> 6) * 24 + q + TPR * t] = dsd[t];
      if (q == 0) { red[(tid >> 6) * 24 + 18] = surr; red[(tid >> 6) * 24 + 19] = vls; red[(tid >> 6) * 24 + 20] = preg; }
    }
    LBAR();
    if (tid < 21) {
      const float v = (red[tid] + red[24 + tid]) + (red[48 + tid] + red[72 + tid]);
      if (tid < 18) dstd_partial[tile * 18 + tid] = v; else loss_partial[tile * 3 + (tid - 18)] = v;
    }
  }
  if (tid < HROWS) {
    dz_stash[sidx(Bs, D_VLEG, 4, row0 + tid, 0)] = gbuf[tid * 41 + 18];
    dz_stash[sidx(Bs, D_VARM, 4, row0 + tid, 0)] = gbuf[tid * 41 + 19];
  }
  {
    float w[66];
    BwdFetch16 f;
    f32x4 saved[PPO_MB][2];
#pragma unroll
    for (int m = 0; m < PPO_MB; ++m) { saved[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; saved[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    PSTAMP(4);
    bwd_fetch16(f, BT.s[0], act_stash, row0, Bs);
    bwd_load16(w, BT.s[0], blob);
    const int lane = tid & 63, wave = wave_role();
#pragma unroll 1
    for (int st = 0; st < NBWD; ++st) {
      const BwdDesc& d = BT.s[st];
      bwd_pre_act16(d, f, smem, dz_stash, row0, B, Bs);
      const bool mma = d.has_mma && wave < d.nblkT;
      f32x4 acc[PPO_MB][2];
#pragma unroll
      for (int m = 0; m < PPO_MB; ++m) {
        acc[m][0] = d.add_saved ? saved[m][0] : (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = d.add_saved ? saved[m][1] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (mma) mfma_chain16<PPO_MB>(smem + d.buf_off + (lane & 15) * LD16 + 8 * (lane >> 4), w, (d.out_dim + 31) >> 5, acc);
      {
        const BwdDesc& dn = BT.s[st + 1 < NBWD ? st + 1 : st];
        bwd_fetch16(f, dn, act_stash, row0, Bs);          // first: loads return in order, and the next stage starts with these
        bwd_load16(w, dn, blob);
      }
      if (mma) {
        if (d.save_out) {
#pragma unroll
          for (int m = 0; m < PPO_MB; ++m) { saved[m][0] = acc[m][0]; saved[m][1] = acc[m][1]; }
        } else {
          const int col = wave * 32 + (lane & 15);
          float* out = smem + d.out_off + 4 * (lane >> 4) * LD16 + col;
#pragma unroll
          for (int m = 0; m < PPO_MB; ++m) {
            if (col < d.in_dim) {
#pragma unroll
              for (int r = 0; r < 4; ++r) out[(16 * m + r) * LD16] = acc[m][0][r];
            }
            if (col + 16 < d.in_dim) {
#pragma unroll
              for (int r = 0; r < 4; ++r) out[(16 * m + r) * LD16 + 16] = acc[m][1][r];
            }
          }
        }
      }
      if (d.has_mma) LBAR();
      PSTAMP(5 + st);
    }
  }
}

// ---- weight and bias gradients -------------------------------------------------------------------
// dW_l[o][i] = sum_rows dZ_l[row][o] * A_{l-1}[row][i], db_l[o] = sum_rows dZ_l[row][o]. grid = (splits, layers);
// a workgroup owns a row range of one layer, wave w the output rows [32w, 32w+32). Both MFMA operands are read
// straight from the stashes (a fragment = two runs of 32 consecutive floats: coalesced as stored), the bias
// gradient falls out of one extra MFMA block whose B operand is the constant 1. No LDS, no atomics.
struct WgradLayer { int out, in, dcol, dw, acol, aw, aoff, goff, nsplit, rows, ldw, boff, bias; };   // ldw: row stride of dW (the layer's full input width), goff: first element of this column range, boff: the bias gradient (written iff bias)   // nsplit row ranges of `rows` rows: proportional to the layer's MFMAs per row pair, so every wave has the same work   // dZ slab (start, width), A slab (start, width, first column in it); goff: offset of this layer's weight gradient in the flat buffer
#ifndef WG_U
#define WG_U 4               // row pairs per batch of operand loads (two batches in flight per wave)
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
#define WG_STAGES 2        // 3 (two batches in flight behind the one being computed) measured the same: 168.8 vs 168.2 us
#endif
  if (nfull > 0) {
#if WG_STAGES == 3
    // Three operand sets: while a batch's MFMAs run, the loads of the next TWO batches are in flight (2 x 1024 cycles of
    // latency hiding per wave instead of one; 60 operand registers, still three waves per SIMD).
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

// Grid = (row ranges, layers), 4 waves per workgroup = the layer's four 32-row output blocks. About three workgroups per
// CU: co-resident waves do not overlap each other's vector-ALU work with MFMAs, but they do hide each other's memory latency
// (one workgroup per CU, 8 row pairs in flight per wave: 340 us; three: 180 us).
// Work decomposition. A 32-row output block x 32-column input block of a layer costs one MFMA per row pair; the layers
// have 2, 2, 12, 4 or 16 such blocks. With one workgroup per (layer, row range) all workgroups are resident at once (three
// per CU) and the CUs that happen to hold three 16-block workgroups set the kernel's duration (169 us at 54 % matrix-pipe
// use; dealing the workgroups out by work needs to know which workgroups share a CU and gave 2 %). Instead the layers are
// bundled into WG_NVL = 11 "virtual layers" of exactly 16 blocks each -- the nine 128 x 128-class layers; backbone (12) +
// priv0 (2) + priv2 (2, its two input blocks on different waves); the four heads (4 each) -- and every wave of every
// workgroup gets tasks worth 4 MFMAs per row pair: all workgroups are equal, whatever the placement. 168.6 -> 150.4 us.
#ifndef PPO_NSPLIT
#define PPO_NSPLIT 69       // x 11 virtual layers = 759 workgroups: one resident wave of three per CU
#endif
struct WgradTask { int sub, ob; };                      // sub-layer (a layer or a column range of one), 32-row output block
#define WG_NVL 11
#define WG_NSUB (NLAYERS + 1)
struct WgradPlan { WgradLayer sub[WG_NSUB]; WgradTask task[WG_NVL][4][2]; int ntask[WG_NVL][4]; int nsplit, rows; };

extern "C" __global__ void __launch_bounds__(PT_THREADS, 3) ppo_wgrad_kernel(WgradPlan plan, const float* __restrict__ act_stash,
                                                                         const float* __restrict__ dz_stash, float* __restrict__ wpart,
                                                                         int B, int Bs, int nparams) {
  const int vl = blockIdx.x / plan.nsplit, split = blockIdx.x - vl * plan.nsplit;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // uniform: the task and its layer live in SGPRs
  const int r_begin = min(B, split * plan.rows), r_end = min(B, r_begin + plan.rows);     // (an empty range still writes its zeros)
  float* dst = wpart + (size_t)split * nparams;
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

struct RedLayer { int goff, count; };
struct RedTable { RedLayer l[NLAYERS]; int nsplit; };
#define RED_BX ((128 * 128 + 128 + 255) / 256)            // blocks per grid row (the largest layer: 65)
#define PPO_SQ_PARTS ((NLAYERS + 1) * RED_BX)             // one partial sum of squares per block of ppo_grad_reduce_kernel
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
      for (; s2 + 8 <= tab.nsplit; s2 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(s2 + j) * stride + p];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
      for (; s2 < tab.nsplit; ++s2) acc += part[(size_t)s2 * stride + p];
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
#define ADAM_NBLK 64
// partial sums of g^2: block b takes elements b*256+tid, +64*256, ... ; fixed-order tree inside the block
extern "C" __global__ void __launch_bounds__(256) ppo_sqnorm_kernel(const float* __restrict__ g, int n, float* __restrict__ part) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += ADAM_NBLK * 256) acc += g[i] * g[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

struct AdamTable {
  float* p[33];
  int off[34];                    // flat offset of parameter j; off[33] = total
};

// g <- s g (s = grad_scale, 1 / world_size after a SUM all-reduce), then g <- g * min(1, max_norm / (||g|| + 1e-6));
// m, v, p as torch.optim.Adam (no weight decay, no amsgrad):
// m += (g-m)(1-b1); v = v b2 + (1-b2) g g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
extern "C" __global__ void __launch_bounds__(256) ppo_adam_kernel(AdamTable T, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                                 const float* __restrict__ part, int nparts, float max_norm, float beta1,
                                                                 float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale) {
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
  *pp = *pp - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

// ---- C-ABI ----------------------------------------------------------------------------------------
// Flat gradient layout: for layer l in PolicyParams order: weight [out*in] then bias [out]; then std [18];
// then 3 loss sums (surrogate, value, priv_reg; divide by 2B, 2B, B for the means).
static const int kDcol[NLAYERS] = {D_H1, D_LAT, D_BB, D_L1, D_L2, D_LEG, D_A1, D_A2, D_ARM, D_CB, D_CL1, D_CL2, D_VLEG, D_CA1, D_CA2, D_VARM};
static const int kAcol[NLAYERS] = {A_X + PT_NPROP, A_H1, A_Z, A_BB, A_L1, A_L2, A_BB, A_A1, A_A2, A_X, A_CB, A_CL1, A_CL2, A_CB, A_CA1, A_CA2};
#define PPO_WPACK_FLOATS (WPACK16_FLOATS > CHAIN_PACK_FLOATS ? WPACK16_FLOATS : CHAIN_PACK_FLOATS)

// The equal-work plan of ppo_wgrad_kernel for this network (checked: every wave of every virtual layer gets 4 blocks).
static int make_wgrad_plan(WgradPlan& plan, RedTable& red, int B) {
  int goff[NLAYERS], off = 0;
  for (int l = 0; l < NLAYERS; ++l) { goff[l] = off; red.l[l] = RedLayer{off, layer_out(l) * layer_in(l) + layer_out(l)}; off += layer_out(l) * layer_in(l) + layer_out(l); }
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
  plan.nsplit = red.nsplit = PPO_NSPLIT;
  plan.rows = ((B + PPO_NSPLIT - 1) / PPO_NSPLIT + 7) / 8 * 8;
  return off;
}

extern "C" int wbc_ppo_grad_floats(void) {
  int n = 0;
  for (int l = 0; l < NLAYERS; ++l) n += layer_out(l) * layer_in(l) + layer_out(l);
  return n + 18 + 3;
}
// rows of a stash slab: whole tiles (every tile row is stored), then a multiple of 64
static int ppo_slab_rows(int B) { return ((B + HROWS - 1) / HROWS * HROWS + 63) & ~63; }
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
extern "C" int wbc_ppo_minibatch_grad(const void* const* params, const float* obs, const float* actions, const float* old_values,
                                      const float* advantages, const float* returns, const float* old_logp, const float* hist_latent,
                                      const int64_t* idx, int B, float clip, float value_coef, float mixing, float roa_coef,
                                      int use_clipped_value_loss, float* workspace, float* grad, float* loss_accum, void* stream) {
  PolicyParams P;
  if (!params || !obs || !actions || !old_values || !advantages || !returns || !old_logp || !hist_latent || !idx || !workspace || !grad ||
      B <= 0 || fill_params(params, &P))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const int tile_rows = HROWS;
  const int tiles = (B + tile_rows - 1) / tile_rows;
  const int ng = wbc_ppo_grad_floats();
  const int Bs = ppo_slab_rows(B);          // rows per stash slab
  float* act_stash = workspace;
  float* dz_stash = act_stash + (size_t)Bs * A_LD;
  float* dstd_partial = dz_stash + (size_t)Bs * D_LD;
  const size_t tiles16w = (size_t)(B + R16 - 1) / R16;                // the workspace holds one partial per 16-row tile
  float* loss_partial = dstd_partial + tiles16w * 18;
  float* wpart = loss_partial + tiles16w * 3;
  float* wpack = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(wpart + (size_t)PPO_NSPLIT * ng) + 15) & ~(uintptr_t)15);   // float4 loads
  static const bool use_old = getenv("WBC_PPO_OLD16") != nullptr;          // development A/B switch
  PpoBatch Bt{obs, actions, old_values, advantages, returns, old_logp, hist_latent, idx, B, Bs, clip, value_coef, mixing, roa_coef, use_clipped_value_loss};
  int nparts = tiles;
  if (use_old) {
    hipLaunchKernelGGL(wbc_pack16_kernel, dim3(8, NLAYERS, 2), dim3(256), 0, st, P, make_pack16_table(), wpack);
    static const int kStashCols[NLAYERS] = {A_H1, A_LAT, A_BB, A_L1, A_L2, A_LEG, A_A1, A_A2, A_ARM, A_CB, A_CL1, A_CL2, -1, A_CA1, A_CA2, -1};
    hipLaunchKernelGGL(ppo_fwd_bwd16_kernel, dim3(tiles), dim3(PT_THREADS), 0, st, P, make_fwd_table16(kStashCols), make_bwd_table16(P), wpack, Bt,
                       act_stash, dz_stash, dstd_partial, loss_partial);
  } else {
    const ChainStreams& S = chain_streams();
    const int tiles16 = (B + 15) / 16;
    nparts = tiles16;
    hipLaunchKernelGGL(chain_pack_kernel, dim3(96, 2), dim3(256), 0, st, P, S, wpack);
    hipLaunchKernelGGL(ppo_chain_kernel, dim3((2 * tiles16 + 3) / 4), dim3(PT_THREADS), 0, st, wpack, S.base[1], S.nelem[0] * 1024, S.nelem[1] * 1024, Bt, P.std, act_stash, dz_stash,
                       dstd_partial, loss_partial, tiles16);
  }
  WgradPlan plan;
  RedTable red;
  const int off = make_wgrad_plan(plan, red, B);
  if (off < 0) return -2;
  hipLaunchKernelGGL(ppo_wgrad_kernel, dim3(WG_NVL * PPO_NSPLIT), dim3(PT_THREADS), 0, st, plan, act_stash, dz_stash, wpart, B, Bs, ng);
  hipLaunchKernelGGL(ppo_grad_reduce_kernel, dim3(RED_BX, NLAYERS + 1), dim3(256), 0, st, red, wpart, ng, grad, dstd_partial, nparts, loss_partial,
                     grad + off, loss_accum, workspace + ppo_sq_offset(B));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// clip_grad_norm_(params, max_norm) followed by Adam.step() for the 33 parameters of `params`, whose gradients are
// grad[0 : wbc_ppo_grad_floats()-3] in the layout above; exp_avg / exp_avg_sq: flat state in the same layout.
// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) (computed by the caller in double, as torch does).
// max_norm <= 0: no clipping. workspace: >= 64 floats.
extern "C" int wbc_ppo_clip_adam(const void* const* params, float* grad, float* exp_avg, float* exp_avg_sq, float max_norm, float beta1,
                                 float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, const float* sq_partials, float* workspace,
                                 void* stream) {
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
  // the norm: from the partials wbc_ppo_minibatch_grad left (the gradient is still the one it produced), or a pass over grad
  const bool have = sq_partials != nullptr;
  if (max_norm > 0.f && !have) hipLaunchKernelGGL(ppo_sqnorm_kernel, dim3(ADAM_NBLK), dim3(256), 0, st, grad, off, workspace);
  hipLaunchKernelGGL(ppo_adam_kernel, dim3((off + 255) / 256), dim3(256), 0, st, T, grad, exp_avg, exp_avg_sq, have ? sq_partials : workspace,
                     have ? PPO_SQ_PARTS : ADAM_NBLK, max_norm, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
