// wbc_ppo_chain.h -- the fused forward + loss + backward pass of one PPO minibatch (reference rsl_rl/algorithms/ppo.py:163-246,
// modules/actor_critic.py:204-217,281-286) with every activation of a row tile in REGISTERS (gfx950, v_mfma_f32_16x16x4_f32).
// Included by wbc_ppo_kernel.hip after the stash layouts.
//
// Work unit = (16 minibatch rows, actor | critic): one WAVEFRONT runs the unit's whole layer chain forward, evaluates its
// loss terms and back-propagates, alone: no LDS, no barrier, nothing shared with the other waves of its workgroup.
// B = 40960 rows give 5120 equal units = exactly 5 per SIMD of the 256 CUs (the 32-row x 4-wave workgroups of the kernel
// this replaces met at ~60 barriers per tile and left a third of the SIMD time without an instruction to issue).
//
// The GEMMs are computed TRANSPOSED: out^T[feature, row] = W[feature, k] * in^T[k, row]. The MFMA's A operand (16 x 4) is a
// weight fragment, its B operand (4 x 16) the activations, and the result D[m = 4 g + r][n] (lane = 16 g + n, register r)
// leaves lane (g, n) with features 16 j + 4 g + r of row n for out-block j. The B operand wants lane (g, n) to supply
// in^T[k-slot g][row n]: register r of out-block j IS a valid B operand for the k-step that stands for the features
// {16 j + 4 g + r : g = 0..3} -- the reduction order is free, the weights are packed to match. A layer's output registers are
// the next layer's input operands as they are: bias + activation happen in place, element-wise.
// The backward chain is the same with W^T fragments (dIn^T = W^T dZ^T), the activation derivative applied to the result
// registers from the stashed post-activations (read back in the very layout they were stored in).
//
// Weights: ONE stream per part, in consumption order (forward layers, then the backward stages), element = 64 lanes x float4
// = the A operands of four consecutive MFMAs (or a bias quadruple): a wave walks its stream with a 9-element register ring
// (one out-block ahead), the address is a scalar pointer bumped by 1 KB per element. 717 (actor) + 666 (critic) KB, L2 resident.
#pragma once

#define CH_D 9                    // ring depth (stream elements in flight)
#ifndef CH_OCC
#define CH_OCC 3                  // waves per SIMD the register budget is set for
#endif

// ---- stream layout ----------------------------------------------------------------------------------------------------
enum { SEG_FWD = 0, SEG_BWD = 1, SEG_PAD = 2 };
struct ChainSeg { int start, layer, kind, nblk, ngrp; };       // SEG_FWD: per block ngrp weight elements + 1 bias element;
                                                                // SEG_BWD: per block ngrp weight elements; SEG_PAD: nblk zero elements
#define CH_MAXSEG 24
struct ChainStreams { ChainSeg s[2][CH_MAXSEG]; int nseg[2], nelem[2], base[2]; };      // base: float offset of the part's stream in the pack

static ChainStreams make_chain_streams() {
  ChainStreams t;
  for (int p = 0; p < 2; ++p) {
    int n = 0, pos = 0;
    auto seg = [&](int layer, int kind, int nblk, int ngrp) {
      t.s[p][n++] = ChainSeg{pos, layer, kind, nblk, ngrp};
      pos += kind == SEG_FWD ? nblk * (ngrp + 1) : (kind == SEG_BWD ? nblk * ngrp : nblk);
    };
    if (p == 0) {
      seg(L_PRIV0, SEG_FWD, 4, 3); seg(L_PRIV2, SEG_FWD, 2, 4); seg(-1, SEG_PAD, 1, 0);            // 16 + 10 + 1 = 27
      seg(L_BB, SEG_FWD, 8, 7); seg(-1, SEG_PAD, 8, 0);                                              // 64 + 8
      seg(L_LEG0, SEG_FWD, 8, 8); seg(L_LEG2, SEG_FWD, 8, 8); seg(L_LEG4, SEG_FWD, 1, 8);           // 153
      seg(L_ARM0, SEG_FWD, 8, 8); seg(L_ARM2, SEG_FWD, 8, 8); seg(L_ARM4, SEG_FWD, 1, 8);
      seg(L_LEG4, SEG_BWD, 8, 1); seg(L_LEG2, SEG_BWD, 8, 8); seg(L_LEG0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, 8, 0);    // 144
      seg(L_ARM4, SEG_BWD, 8, 1); seg(L_ARM2, SEG_BWD, 8, 8); seg(L_ARM0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, 8, 0);
      seg(L_BB, SEG_BWD, 2, 8); seg(L_PRIV2, SEG_BWD, 4, 2);                                         // 16 + 8
    } else {
      seg(L_CBB, SEG_FWD, 8, 7); seg(-1, SEG_PAD, 8, 0);
      seg(L_CLEG0, SEG_FWD, 8, 8); seg(L_CLEG2, SEG_FWD, 8, 8); seg(L_CLEG4, SEG_FWD, 1, 8);
      seg(L_CARM0, SEG_FWD, 8, 8); seg(L_CARM2, SEG_FWD, 8, 8); seg(L_CARM4, SEG_FWD, 1, 8);
      seg(L_CLEG4, SEG_BWD, 8, 1); seg(L_CLEG2, SEG_BWD, 8, 8); seg(L_CLEG0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, 8, 0);
      seg(L_CARM4, SEG_BWD, 8, 1); seg(L_CARM2, SEG_BWD, 8, 8); seg(L_CARM0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, 8, 0);
    }
    seg(-1, SEG_PAD, CH_D, 0);                    // the ring reads CH_D elements past the last one consumed
    t.nseg[p] = n; t.nelem[p] = pos;
  }
  t.base[0] = 0; t.base[1] = t.nelem[0] * 256;
  return t;
}
static const ChainStreams& chain_streams() { static const ChainStreams t = make_chain_streams(); return t; }
#define CHAIN_PACK_FLOATS ((chain_streams().nelem[0] + chain_streams().nelem[1]) * 256)

// which input column of W the k-step (group kq, component i) stands for on lane group g, forward layers (-1: none)
static __device__ __forceinline__ int chain_fwd_kcol(int layer, int kq, int g, int i) {
  const int t = 16 * kq + 4 * g + i;
  if (layer == L_PRIV0) { const int f = t + 64; return (f >= PT_NPROP && f < 100) ? f - PT_NPROP : -1; }      // x blocks 4..6: the privileged part
  if (layer == L_BB) { if (kq < 5) return t < PT_NPROP ? t : -1; const int u = t - 80; return u < 20 ? PT_NPROP + u : -1; }   // proprio from x, then the latent blocks
  return t < layer_in(layer) ? t : -1;
}

// grid = (blocks, 2 parts). One thread per float of the streams.
static __global__ void __launch_bounds__(256) chain_pack_kernel(PolicyParams P, ChainStreams S, float* __restrict__ pack) {
  const int p = blockIdx.y;
  const int total = S.nelem[p] * 256;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = e & 3, lane = (e >> 2) & 63, el = e >> 8;
    const int g = lane >> 4, m = lane & 15;
    int si = 0;
    while (si + 1 < S.nseg[p] && S.s[p][si + 1].start <= el) ++si;
    const ChainSeg sg = S.s[p][si];
    float v = 0.f;
    if (sg.kind != SEG_PAD) {
      const float* W = reinterpret_cast<const float* const*>(&P)[2 * sg.layer];
      const float* bsrc = reinterpret_cast<const float* const*>(&P)[2 * sg.layer + 1];
      const int N = layer_out(sg.layer), K = layer_in(sg.layer);
      const int rel = el - sg.start;
      if (sg.kind == SEG_FWD) {
        const int blk = rel / (sg.ngrp + 1), kq = rel - blk * (sg.ngrp + 1);
        if (kq == sg.ngrp) {                                   // bias quadruple of lane group g: features 16 blk + 4 g + i
          const int o = 16 * blk + 4 * g + i;
          v = o < N ? bsrc[o] : 0.f;
        } else {
          const int o = 16 * blk + m, c = chain_fwd_kcol(sg.layer, kq, g, i);
          v = (o < N && c >= 0) ? W[(size_t)o * K + c] : 0.f;
        }
      } else {                                                 // dIn^T = W^T dZ^T: rows = the layer's inputs, k over its outputs
        const int blk = rel / sg.ngrp, kq = rel - blk * sg.ngrp;
        const int o = 16 * kq + 4 * g + i;
        int c = 16 * blk + m;
        if (sg.layer == L_BB) c = c < 20 ? PT_NPROP + c : K;  // only the latent columns of the backbone's input need a gradient
        v = (o < N && c < K) ? W[(size_t)o * K + c] : 0.f;
      }
    }
    pack[S.base[p] + e] = v;
  }
}

// ---- the ring ---------------------------------------------------------------------------------------------------------
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define CH_RSRC_FLAGS 0x00020000          // raw buffer, 32-bit data format (gfx9 family)
// Ring loads are pinned where they are written (the scheduler otherwise sinks each load to its use, nine elements later)
#define CH_PIN() __builtin_amdgcn_sched_barrier(0)
struct ChainRing {
  f32x4 r[CH_D];
  __amdgpu_buffer_rsrc_t rs;  // the part's stream
  int so;                     // uniform: byte offset of the element the next load fetches
  int loff;                   // lane * 16
  __device__ __forceinline__ void init(const float* stream, int bytes, int lane) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, bytes, CH_RSRC_FLAGS);
    so = 0; loff = lane * 16;
#pragma unroll
    for (int s = 0; s < CH_D; ++s) { r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, so, 0)); so += 1024; }
    CH_PIN();
  }
  // the element at ring slot `slot` (a compile-time constant after unrolling); its slot is refilled from the stream head
  __device__ __forceinline__ f32x4 take(int slot) {
    const f32x4 v = r[slot];
    r[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, so, 0));
    so += 1024;
    CH_PIN();
    return v;
  }
};

// stash accesses: buffer resource of the whole stash, per-lane byte offset, uniform slab offset (bytes), small immediate
static __device__ __forceinline__ f32x4 ch_ld4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
static __device__ __forceinline__ void ch_st4(__amdgpu_buffer_rsrc_t rs, int voff, int soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff, 0);
}
static __device__ __forceinline__ f32x4 ch_mfma4(const f32x4 w, const f32x4 b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], b[3], acc, 0, 0, 0);
  return acc;
}
static __device__ __forceinline__ float ch_elu(float x) {       // as act16<ACT_ELU>
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.f;
  return x > 0.f ? x : e;
}
static __device__ __forceinline__ f32x4 ch_elu4(f32x4 a, f32x4 b) {
  return (f32x4){ch_elu(a[0] + b[0]), ch_elu(a[1] + b[1]), ch_elu(a[2] + b[2]), ch_elu(a[3] + b[3])};
}
// d * ELU'(a) from the stored post-activation a: ELU' = 1 (a > 0), a + 1 otherwise
static __device__ __forceinline__ f32x4 ch_delu4(f32x4 d, f32x4 a) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = d[r] * (a[r] > 0.f ? 1.f : a[r] + 1.f);
  return o;
}

// Forward layer: out[ob] = act(W in + b) for NOB out-blocks over NKG k-groups; POS0 = ring position (mod CH_D) of its first
// element. STASH: the output goes to the activation slab at (uniform byte offset soff) + (lane offset voff) + 64 ob. The
// epilogue of a block runs after the next block's MFMAs were issued (the matrix pipe's result latency is not waited for).
template <int NOB, int NKG, int POS0, bool ELU, bool STASH>
static __device__ __forceinline__ void chain_fwd(ChainRing& R, const f32x4* bin, f32x4* out, __amdgpu_buffer_rsrc_t ars, int voff, int soff) {
  f32x4 bprev = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) acc = ch_mfma4(R.take((POS0 + ob * (NKG + 1) + kq) % CH_D), bin[kq], acc);
    out[ob] = acc;
    const f32x4 b = R.take((POS0 + ob * (NKG + 1) + NKG) % CH_D);
    if (ob > 0) {
      out[ob - 1] = ELU ? ch_elu4(out[ob - 1], bprev) : out[ob - 1] + bprev;
      if (STASH) ch_st4(ars, voff + 64 * (ob - 1), soff, out[ob - 1]);
    }
    bprev = b;
  }
  out[NOB - 1] = ELU ? ch_elu4(out[NOB - 1], bprev) : out[NOB - 1] + bprev;
  if (STASH) ch_st4(ars, voff + 64 * (NOB - 1), soff, out[NOB - 1]);
}

// Backward stage: dout[ib] (+)= W^T dz over NKG k-groups for NIB in-blocks; then, DERIV: dout[ib] *= ELU'(a[ib]) with a read
// from the activation slab (asoff) one block ahead of its use, and the result stored to the dZ slab (dsoff), both at the lane
// offset voff + 64 ib. ACCUM: the accumulators start from dout (no derivative pass).
template <int NIB, int NKG, int POS0, bool DERIV, bool ACCUM>
static __device__ __forceinline__ void chain_bwd(ChainRing& R, const f32x4* dz, f32x4* dout, __amdgpu_buffer_rsrc_t ars, int asoff,
                                                 __amdgpu_buffer_rsrc_t drs, int dsoff, int voff) {
  f32x4 ap[NIB];
  if (DERIV) ap[0] = ch_ld4(ars, voff, asoff);
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib) {
    if (DERIV && ib + 1 < NIB) ap[ib + 1] = ch_ld4(ars, voff + 64 * (ib + 1), asoff);
    f32x4 acc = ACCUM ? dout[ib] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) acc = ch_mfma4(R.take((POS0 + ib * NKG + kq) % CH_D), dz[kq], acc);
    dout[ib] = acc;
    if (DERIV && ib > 0) {
      dout[ib - 1] = ch_delu4(dout[ib - 1], ap[ib - 1]);
      ch_st4(drs, voff + 64 * (ib - 1), dsoff, dout[ib - 1]);
    }
  }
  if (DERIV) {
    dout[NIB - 1] = ch_delu4(dout[NIB - 1], ap[NIB - 1]);
    ch_st4(drs, voff + 64 * (NIB - 1), dsoff, dout[NIB - 1]);
  }
}

static __device__ __forceinline__ float ch_sum_g(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }       // over the 4 lane groups of a row
static __device__ __forceinline__ float ch_sum_n(float v) {                                                                       // over the 16 rows of a lane group
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

// ntiles 16-row tiles; stream_bytes[p], off_critic: the two streams inside `pack`. Stashes: Bs rows per slab, slab of start
// column c0 at byte offset 4 c0 Bs (both stashes stay below 4 GB: 32-bit buffer offsets).
extern "C" __global__ void __launch_bounds__(PT_THREADS, CH_OCC) ppo_chain_kernel(const float* __restrict__ pack, int off_critic, int bytes_actor,
                                                                                 int bytes_critic, PpoBatch Bt, const float* __restrict__ stdp,
                                                                                 float* __restrict__ act_stash, float* __restrict__ dz_stash,
                                                                                 float* __restrict__ dstd_partial, float* __restrict__ loss_partial,
                                                                                 int ntiles) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * (PT_THREADS / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = unit >> 1;
  const bool critic = (unit & 1) != 0;                 // wave-uniform
  if (tile >= ntiles) return;
  const int g = lane >> 4, n = lane & 15;
  const int B = Bt.B, Bs = Bt.Bs;
  const int row0 = tile * 16, row = row0 + n;
  const bool valid = row < B;
  const size_t src = (size_t)Bt.idx[min(row, B - 1)];
  ChainRing R;
  R.init(pack + (critic ? off_critic : 0), critic ? bytes_critic : bytes_actor, lane);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(act_stash, 0, Bs * (A_LD * 4), CH_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dz_stash, 0, Bs * (D_LD * 4), CH_RSRC_FLAGS);
  const int slab = Bs * 4;                             // bytes per stash column
  // per-lane byte offsets into a slab of width w: row `row`, column 4 g
  const int v128 = (row * 128 + 4 * g) * 4, v100 = (row * 100 + 4 * g) * 4, v64 = (row * 64 + 4 * g) * 4, v20 = (row * 20 + 4 * g) * 4;
  // x = obs[src, :100]: block q holds the features 16 q + 4 g + r (the layout of a layer output)
  f32x4 zin[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int c = 16 * q + 4 * g;
    f32x4 v = *reinterpret_cast<const f32x4*>(Bt.obs + src * PT_NOBS + (c < 100 ? c : 0));
    if (c >= 100) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    zin[q] = v;
  }
  f32x4 hbb[8];
  if (!critic) {
    // the x slab and the proprio part of the z = [prop, latent] slab (inputs of the priv0 / cbb / bb weight gradients)
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int c = 16 * q + 4 * g;
      if (c < 100) ch_st4(ars, v100 + 64 * q, A_X * slab, zin[q]);
      if (c < PT_NPROP) ch_st4(ars, v100 + 64 * q, A_Z * slab, zin[q]);
    }
    f32x4 h1[4], lat[2];
    chain_fwd<4, 3, 0, true, true>(R, zin + 4, h1, ars, v64, A_H1 * slab);
    chain_fwd<2, 4, 16, true, false>(R, h1, lat, ars, 0, 0);
    (void)R.take(26 % CH_D);
    // latent: the 20-wide slab and the z slab's columns 76..95
    ch_st4(ars, v20, A_LAT * slab, lat[0]);
    ch_st4(ars, v100 + 4 * PT_NPROP, A_Z * slab, lat[0]);
    if (g == 0) { ch_st4(ars, v20 + 64, A_LAT * slab, lat[1]); ch_st4(ars, v100 + 4 * PT_NPROP + 64, A_Z * slab, lat[1]); }
    zin[5] = lat[0]; zin[6] = lat[1];
  }
  // backbone (actor: [prop, latent] -> 128; critic: x -> 128): the same code, the stream holds the part's weights
  chain_fwd<8, 7, 0, true, true>(R, zin, hbb, ars, v128, (critic ? A_CB : A_BB) * slab);
#pragma unroll
  for (int s = 0; s < 8; ++s) (void)R.take((64 + s) % CH_D);
  // the two heads of the part: 128 -> 128 -> 128 -> (12 | 6 | 1 | 1)
  f32x4 out0 = (f32x4){0.f, 0.f, 0.f, 0.f}, out1 = out0;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int c_l0 = critic ? (h ? A_CA1 : A_CL1) : (h ? A_A1 : A_L1), c_l2 = critic ? (h ? A_CA2 : A_CL2) : (h ? A_A2 : A_L2);
    f32x4 a1[8], a2[8], o[1];
    chain_fwd<8, 8, 0, true, true>(R, hbb, a1, ars, v128, c_l0 * slab);
    chain_fwd<8, 8, 0, true, true>(R, a1, a2, ars, v128, c_l2 * slab);
    chain_fwd<1, 8, 0, false, false>(R, a2, o, ars, 0, 0);
    if (!critic) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[0][r] = tanhf(o[0][r]);
      const int w = h ? 8 : PT_NLEG;
      if (4 * g < w) ch_st4(ars, (row * w + 4 * g) * 4, (h ? A_ARM : A_LEG) * slab, o[0]);
    }
    if (h == 0) out0 = o[0]; else out1 = o[0];
  }
  // ---- losses and output gradients (PPO:199-216) -------------------------------------------------------------------------
  const float inv2B = 1.f / (2.f * (float)B), invB = 1.f / (float)B;
  f32x4 dz0 = (f32x4){0.f, 0.f, 0.f, 0.f}, dz1 = dz0;
  if (!critic) {
    float lp[2] = {0.f, 0.f}, d[2][4], sd[2][4];
    bool ok[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 4 * g + r;
        ok[c][r] = c ? jj < PT_NARM : jj < PT_NLEG;
        const int j = ok[c][r] ? (c ? PT_NLEG + jj : jj) : 0;
        const float mu = c ? out1[r] : out0[r];
        sd[c][r] = stdp[j];
        d[c][r] = Bt.actions[src * 18 + j] - mu;
        const float term = -(d[c][r] * d[c][r]) / (2.f * sd[c][r] * sd[c][r]) - logf(sd[c][r]) - 0.91893853320467274178f;
        if (ok[c][r]) lp[c] += term;
      }
    lp[0] = ch_sum_g(lp[0]); lp[1] = ch_sum_g(lp[1]);
    const float adv0 = Bt.advantages[src * 2], adv1 = Bt.advantages[src * 2 + 1];
    const float mixed[2] = {adv0 + Bt.mixing * adv1, adv1 + Bt.mixing * adv0};               // PPO:199-201
    float dlp[2], surr = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float ratio = expf(lp[c] - Bt.old_logp[src * 2 + c]);                               // PPO:202
      const float rc = fminf(fmaxf(ratio, 1.f - Bt.clip), 1.f + Bt.clip);
      const float s1 = -mixed[c] * ratio, s2 = -mixed[c] * rc;                                  // PPO:203-205
      surr += fmaxf(s1, s2);
      const bool inside = (ratio >= 1.f - Bt.clip) && (ratio <= 1.f + Bt.clip);
      const float dr = (inside || s1 > s2) ? -mixed[c] : 0.f;
      dlp[c] = valid ? inv2B * dr * ratio : 0.f;
    }
    float dsd[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = sd[c][r], dd = d[c][r], mu = c ? out1[r] : out0[r];
        const float gmu = ok[c][r] ? dlp[c] * dd / (s * s) : 0.f;
        const float gz = gmu * (1.f - mu * mu);                                                 // through the tanh of the head
        if (c) dz1[r] = gz; else dz0[r] = gz;
        dsd[c][r] = ok[c][r] ? dlp[c] * (dd * dd / (s * s * s) - 1.f / s) : 0.f;
        dsd[c][r] = ch_sum_n(dsd[c][r]);
      }
    if (n == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ok[c][r]) dstd_partial[(size_t)tile * 18 + (c ? PT_NLEG : 0) + 4 * g + r] = dsd[c][r];
    }
    surr = (g == 0 && valid) ? surr : 0.f;
    surr = ch_sum_n(surr);
    if (lane == 0) loss_partial[(size_t)tile * 3] = surr;
  } else {
    float vls = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {                                                                 // PPO:209-216
      const float v = c ? out1[0] : out0[0], ov = Bt.old_values[src * 2 + c], Rt = Bt.returns[src * 2 + c];
      float dv;
      if (Bt.use_clipped_value_loss) {
        const float dlt = v - ov;
        const float vc = ov + fminf(fmaxf(dlt, -Bt.clip), Bt.clip);
        const float l1 = (v - Rt) * (v - Rt), l2 = (vc - Rt) * (vc - Rt);
        vls += fmaxf(l1, l2);
        const float m = (dlt >= -Bt.clip && dlt <= Bt.clip) ? 1.f : 0.f;
        const float g1 = 2.f * (v - Rt), g2 = 2.f * (vc - Rt) * m;
        dv = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
      } else {
        vls += (Rt - v) * (Rt - v);
        dv = 2.f * (v - Rt);
      }
      const float gv = (valid && g == 0) ? Bt.value_coef * inv2B * dv : 0.f;                      // the head's single output: feature 0 = (g 0, r 0)
      if (c) dz1[0] = gv; else dz0[0] = gv;
    }
    vls = (g == 0 && valid) ? vls : 0.f;
    vls = ch_sum_n(vls);
    if (lane == 0) loss_partial[(size_t)tile * 3 + 1] = vls;
  }
  // ---- backward ----------------------------------------------------------------------------------------------------------
  f32x4 dbb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int c_l0 = critic ? (h ? A_CA1 : A_CL1) : (h ? A_A1 : A_L1), c_l2 = critic ? (h ? A_CA2 : A_CL2) : (h ? A_A2 : A_L2);
    const int e_l0 = critic ? (h ? D_CA1 : D_CL1) : (h ? D_A1 : D_L1), e_l2 = critic ? (h ? D_CA2 : D_CL2) : (h ? D_A2 : D_L2);
    const int e_hd = critic ? (h ? D_VARM : D_VLEG) : (h ? D_ARM : D_LEG);
    const int w_hd = critic ? 4 : (h ? 8 : PT_NLEG);
    f32x4 dzo[1];
    dzo[0] = h ? dz1 : dz0;
    if (4 * g < w_hd) ch_st4(drs, (row * w_hd + 4 * g) * 4, e_hd * slab, dzo[0]);
    f32x4 d2[8], d1[8];
    chain_bwd<8, 1, 0, true, false>(R, dzo, d2, ars, c_l2 * slab, drs, e_l2 * slab, v128);
    chain_bwd<8, 8, 8, true, false>(R, d2, d1, ars, c_l0 * slab, drs, e_l0 * slab, v128);
    chain_bwd<8, 8, 72 % CH_D, false, true>(R, d1, dbb, ars, 0, drs, 0, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s) (void)R.take((136 + s) % CH_D);
  }
  {                                                      // through the backbone's ELU
    const int asoff = (critic ? A_CB : A_BB) * slab, dsoff = (critic ? D_CB : D_BB) * slab;
    f32x4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = ch_ld4(ars, v128 + 64 * j, asoff);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dbb[j] = ch_delu4(dbb[j], a[j]); ch_st4(drs, v128 + 64 * j, dsoff, dbb[j]); }
  }
  if (critic) return;
  // actor: latent gradient = backbone's input gradient (columns 76..95) + the ROA regulariser's (PPO:174-179), then priv2, priv0
  {
    f32x4 lat[2];
    lat[0] = ch_ld4(ars, v20, A_LAT * slab);
    lat[1] = ch_ld4(ars, v20 + (g == 0 ? 64 : 0), A_LAT * slab);
    const f32x4 hl0 = *reinterpret_cast<const f32x4*>(Bt.hist_latent + src * 20 + 4 * g);
    const f32x4 hl1 = *reinterpret_cast<const f32x4*>(Bt.hist_latent + src * 20 + (g == 0 ? 16 : 0));
    f32x4 dl0, dl1;
    float nrm = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dl0[r] = lat[0][r] - hl0[r];
      dl1[r] = g == 0 ? lat[1][r] - hl1[r] : 0.f;
      nrm += dl0[r] * dl0[r] + dl1[r] * dl1[r];
    }
    nrm = sqrtf(ch_sum_g(nrm));
    const float sc = (nrm > 0.f && valid) ? Bt.roa_coef * invB / nrm : 0.f;
    float preg = (g == 0 && valid) ? nrm : 0.f;
    preg = ch_sum_n(preg);
    if (lane == 0) loss_partial[(size_t)tile * 3 + 2] = preg;
    f32x4 dlat[2];
    chain_bwd<2, 8, 0, false, false>(R, dbb, dlat, ars, 0, drs, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dlat[0][r] += sc * dl0[r]; dlat[1][r] += sc * dl1[r]; }
    dlat[0] = ch_delu4(dlat[0], lat[0]);
    dlat[1] = ch_delu4(dlat[1], lat[1]);
    if (g != 0) dlat[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ch_st4(drs, v20, D_LAT * slab, dlat[0]);
    if (g == 0) ch_st4(drs, v20 + 64, D_LAT * slab, dlat[1]);
    f32x4 dh1[4];
    chain_bwd<4, 2, 16 % CH_D, true, false>(R, dlat, dh1, ars, A_H1 * slab, drs, D_H1 * slab, v64);
  }
}
