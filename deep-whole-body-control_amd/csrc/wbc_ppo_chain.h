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

#ifndef CH_D
#define CH_D 9                    // ring depth (stream elements in flight); must divide 153 (a head chain's forward elements)
#endif
// zero elements that bring the stream position back to a multiple of the ring depth in front of / behind the looped bodies
#define CH_PAD_PRIV ((CH_D - 26 % CH_D) % CH_D)
#define CH_PAD_BB ((CH_D - 64 % CH_D) % CH_D)
#define CH_PAD_BWD ((CH_D - 136 % CH_D) % CH_D)
static_assert(153 % CH_D == 0, "ring depth");
#ifndef CH_WG
#define CH_WG 256                 // threads per workgroup: CH_WG / 64 independent units (nothing is shared between its waves)
#endif
#ifndef CH_OCC
#define CH_OCC 2                  // waves per SIMD the register budget is set for
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
      seg(L_PRIV0, SEG_FWD, 4, 3); seg(L_PRIV2, SEG_FWD, 2, 4); seg(-1, SEG_PAD, CH_PAD_PRIV, 0);            // 16 + 10 + 1 = 27
      seg(L_BB, SEG_FWD, 8, 7); seg(-1, SEG_PAD, CH_PAD_BB, 0);                                              // 64 + 8
      seg(L_LEG0, SEG_FWD, 8, 8); seg(L_LEG2, SEG_FWD, 8, 8); seg(L_LEG4, SEG_FWD, 1, 8);           // 153
      seg(L_ARM0, SEG_FWD, 8, 8); seg(L_ARM2, SEG_FWD, 8, 8); seg(L_ARM4, SEG_FWD, 1, 8);
      seg(L_LEG4, SEG_BWD, 8, 1); seg(L_LEG2, SEG_BWD, 8, 8); seg(L_LEG0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, CH_PAD_BWD, 0);    // 144
      seg(L_ARM4, SEG_BWD, 8, 1); seg(L_ARM2, SEG_BWD, 8, 8); seg(L_ARM0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, CH_PAD_BWD, 0);
      seg(L_BB, SEG_BWD, 2, 8); seg(L_PRIV2, SEG_BWD, 4, 2);                                         // 16 + 8
    } else {
      seg(L_CBB, SEG_FWD, 8, 7); seg(-1, SEG_PAD, CH_PAD_BB, 0);
      seg(L_CLEG0, SEG_FWD, 8, 8); seg(L_CLEG2, SEG_FWD, 8, 8); seg(L_CLEG4, SEG_FWD, 1, 8);
      seg(L_CARM0, SEG_FWD, 8, 8); seg(L_CARM2, SEG_FWD, 8, 8); seg(L_CARM4, SEG_FWD, 1, 8);
      seg(L_CLEG4, SEG_BWD, 8, 1); seg(L_CLEG2, SEG_BWD, 8, 8); seg(L_CLEG0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, CH_PAD_BWD, 0);
      seg(L_CARM4, SEG_BWD, 8, 1); seg(L_CARM2, SEG_BWD, 8, 8); seg(L_CARM0, SEG_BWD, 8, 8); seg(-1, SEG_PAD, CH_PAD_BWD, 0);
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

// Source of float i of stream float4 (part p, index t = 64 el + lane): parameter 2 l (weight) / 2 l + 1 (bias) of layer l and the
// element inside it, or layer -1 for a zero of the padding. kind: SEG_FWD / SEG_BWD of its segment.
struct ChainSrc { int layer, bias, idx, kind; };
static __device__ __forceinline__ ChainSrc chain_slot_source(const ChainStreams& S, int p, int t, int i) {
  const int lane = t & 63, el = t >> 6;
  const int g = lane >> 4, m = lane & 15;
  int si = 0;
  while (si + 1 < S.nseg[p] && S.s[p][si + 1].start <= el) ++si;
  const ChainSeg sg = S.s[p][si];
  ChainSrc r{-1, 0, 0, sg.kind};
  if (sg.kind == SEG_PAD) return r;
  const int N = layer_out(sg.layer), K = layer_in(sg.layer);
  const int rel = el - sg.start;
  // blocks come in pairs (2 p, 2 p + 1) whose elements alternate: two independent accumulators per wave (a dependent
  // v_mfma_f32_16x16x4_f32 issues every 40 cycles, an independent one every 32); single-block layers (heads) keep k order
  if (sg.kind == SEG_FWD) {
    int blk, kq;
    if (sg.nblk == 1) { blk = 0; kq = rel; }
    else { const int pe = 2 * (sg.ngrp + 1), pr = rel / pe, r2 = rel - pr * pe; blk = 2 * pr + (r2 & 1); kq = r2 >> 1; }
    if (kq == sg.ngrp) {                                   // bias quadruple of lane group g: features 16 blk + 4 g + i
      const int o = 16 * blk + 4 * g + i;
      if (o < N) { r.layer = sg.layer; r.bias = 1; r.idx = o; }
    } else {
      const int o = 16 * blk + m, c = chain_fwd_kcol(sg.layer, kq, g, i);
      if (o < N && c >= 0) { r.layer = sg.layer; r.idx = o * K + c; }
    }
  } else {                                                   // dIn^T = W^T dZ^T: rows = the layer's inputs, k over its outputs
    const int pe = 2 * sg.ngrp, pr = rel / pe, r2 = rel - pr * pe;
    const int blk = 2 * pr + (r2 & 1), kq = r2 >> 1;
    int c = 16 * blk + m;
    if (sg.layer == L_BB) c = c < 20 ? PT_NPROP + c : K;    // only the latent columns of the backbone's input need a gradient
    const int o = 16 * kq + 4 * g + i;
    if (o < N && c < K) { r.layer = sg.layer; r.idx = o * K + c; }
  }
  return r;
}

// grid = (blocks, 2 parts), one thread per (element, lane): a float4 of the stream.
static __global__ void __launch_bounds__(256) chain_pack_kernel(PolicyParams P, ChainStreams S, float* __restrict__ pack) {
  const int p = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ((t >> 6) >= S.nelem[p]) return;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const ChainSrc r = chain_slot_source(S, p, t, i);
    v[i] = r.layer < 0 ? 0.f : reinterpret_cast<const float* const*>(&P)[2 * r.layer + r.bias][r.idx];
  }
  reinterpret_cast<float4*>(pack + S.base[p])[t] = make_float4(v[0], v[1], v[2], v[3]);
}

// The inverse of chain_pack_kernel, once per process: tab[kind][flat parameter index] = float offset of its copy in the pack (-1:
// none), kind 0 = the forward streams, 1 = the backward streams; poff[j] = flat offset of parameter j (the flat gradient's layout).
// A weight sits at most once in each kind, a bias 16 times in the forward streams (4 floats apart: the first copy is recorded);
// *clash counts violations.
struct ChainParamOffsets { int off[2 * NLAYERS]; };
static __global__ void __launch_bounds__(256) chain_scatter_table_kernel(ChainStreams S, ChainParamOffsets PO, int nparam, int* __restrict__ tab, int* clash) {
  const int p = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if ((t >> 6) >= S.nelem[p]) return;
  for (int i = 0; i < 4; ++i) {
    const ChainSrc r = chain_slot_source(S, p, t, i);
    if (r.layer < 0) continue;
    if (r.bias && (t & 15) != 0) continue;             // a bias sits in all 16 row lanes of its lane group: 16 copies 4 floats apart, the first is recorded
    const int src = PO.off[2 * r.layer + r.bias] + r.idx;
    const int kind = r.kind == SEG_BWD ? 1 : 0;
    if (atomicExch(&tab[kind * nparam + src], S.base[p] + 4 * t + i) != -1) atomicAdd(clash, 1);
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
// 8 MFMAs on two independent accumulators, strictly alternating (one asm statement: the compiler neither reorders them nor
// renames an accumulator in mid-chain, which costs hazard nops). A dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles,
// alternating chains every 32. FIRST: the accumulators start from 0. The compiler does not see the matrix pipe's result
// latency through an asm statement: results are read CH_MFMA_SETTLE() or >= 8 further MFMAs later.
template <bool FIRST>
static __device__ __forceinline__ void ch_mfma8(const f32x4 w0, const f32x4 w1, const f32x4 b0, const f32x4 b1, f32x4& a0, f32x4& a1) {
  if (FIRST)
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %10, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %6, %14, 0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %7, %15, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %8, %16, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %9, %17, %1"
                 : "=&v"(a0), "=&v"(a1)
                 : "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]),
                   "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
  else
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %6, %14, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %7, %15, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %8, %16, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %1, %9, %17, %1"
                 : "+v"(a0), "+v"(a1)
                 : "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]),
                   "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
}
// 16 wait states: an 8-pass MFMA result is readable by the vector ALU / as an MFMA A/B operand 11 states after its issue
#define CH_MFMA_SETTLE() asm volatile("s_nop 7\n\ts_nop 7" ::: "memory")
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
#define CH_Z4 ((f32x4){0.f, 0.f, 0.f, 0.f})

// Forward layer: out[ob] = ELU(W in + b) for NOB (even) out-blocks over NKG k-groups, two blocks at a time; POS0 = ring
// position (mod CH_D) of its first element. STASH: the output goes to the activation slab at (uniform byte offset soff) +
// (lane offset voff) + 64 ob. The epilogue of a pair runs after the next pair's MFMAs were issued (the matrix pipe's result
// latency is not waited for).
template <int NOB, int NKG, int POS0, bool STASH>
static __device__ __forceinline__ void chain_fwd(ChainRing& R, const f32x4* bin, f32x4* out, __amdgpu_buffer_rsrc_t ars, int voff, int soff) {
  static_assert(NOB % 2 == 0, "pairs of out-blocks");
  constexpr int PE = 2 * (NKG + 1);
  f32x4 bp0 = CH_Z4, bp1 = CH_Z4;
#pragma unroll
  for (int p = 0; p < NOB / 2; ++p) {
    f32x4 a0, a1;
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) {
      const f32x4 w0 = R.take((POS0 + p * PE + 2 * kq) % CH_D), w1 = R.take((POS0 + p * PE + 2 * kq + 1) % CH_D);
      if (kq == 0) ch_mfma8<true>(w0, w1, bin[kq], bin[kq], a0, a1);
      else ch_mfma8<false>(w0, w1, bin[kq], bin[kq], a0, a1);
    }
    out[2 * p] = a0; out[2 * p + 1] = a1;
    const f32x4 b0 = R.take((POS0 + p * PE + 2 * NKG) % CH_D), b1 = R.take((POS0 + p * PE + 2 * NKG + 1) % CH_D);
    if (p > 0) {
      out[2 * p - 2] = ch_elu4(out[2 * p - 2], bp0);
      out[2 * p - 1] = ch_elu4(out[2 * p - 1], bp1);
      if (STASH) { ch_st4(ars, voff + 64 * (2 * p - 2), soff, out[2 * p - 2]); ch_st4(ars, voff + 64 * (2 * p - 1), soff, out[2 * p - 1]); }
    }
    bp0 = b0; bp1 = b1;
  }
  CH_MFMA_SETTLE();
  out[NOB - 2] = ch_elu4(out[NOB - 2], bp0);
  out[NOB - 1] = ch_elu4(out[NOB - 1], bp1);
  if (STASH) { ch_st4(ars, voff + 64 * (NOB - 2), soff, out[NOB - 2]); ch_st4(ars, voff + 64 * (NOB - 1), soff, out[NOB - 1]); }
}

// Head layer (one out-block, no activation): the k-groups alternate between two accumulators. NKG even, NKG + 1 elements.
template <int NKG, int POS0>
static __device__ __forceinline__ f32x4 chain_head(ChainRing& R, const f32x4* bin) {
  static_assert(NKG % 2 == 0, "pairs of k-groups");
  f32x4 a0, a1;
#pragma unroll
  for (int kq = 0; kq < NKG; kq += 2) {
    const f32x4 w0 = R.take((POS0 + kq) % CH_D), w1 = R.take((POS0 + kq + 1) % CH_D);
    if (kq == 0) ch_mfma8<true>(w0, w1, bin[kq], bin[kq + 1], a0, a1);
    else ch_mfma8<false>(w0, w1, bin[kq], bin[kq + 1], a0, a1);
  }
  const f32x4 b = R.take((POS0 + NKG) % CH_D);
  CH_MFMA_SETTLE();
  return (a0 + a1) + b;
}

// Backward stages: dout[ib] = W^T dz over NKG k-groups for NIB (even) in-blocks, two blocks at a time.
// chain_bwd_deriv: dout[ib] *= ELU'(a[ib]) and the result goes to the dZ slab (dsoff), lane offset voff + 64 ib. The
// post-activations a come from the activation slab (asoff), requested when their pair starts and used after the NEXT pair's
// MFMAs were issued -- or, ALL, every block's at stage entry (a stage of 8 MFMAs per pair is too short for the former).
template <int NIB, int NKG, int POS0, bool ALL = false>
static __device__ __forceinline__ void chain_bwd_deriv(ChainRing& R, const f32x4* dz, f32x4* dout, __amdgpu_buffer_rsrc_t ars, int asoff,
                                                       __amdgpu_buffer_rsrc_t drs, int dsoff, int voff) {
  static_assert(NIB % 2 == 0, "pairs of in-blocks");
  constexpr int PE = 2 * NKG;
  f32x4 ap[NIB];
  if (ALL) {
#pragma unroll
    for (int j = 0; j < NIB; ++j) ap[j] = ch_ld4(ars, voff + 64 * j, asoff);
  }
#pragma unroll
  for (int p = 0; p < NIB / 2; ++p) {
    if (!ALL) { ap[2 * p] = ch_ld4(ars, voff + 64 * (2 * p), asoff); ap[2 * p + 1] = ch_ld4(ars, voff + 64 * (2 * p + 1), asoff); }
    f32x4 a0, a1;
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) {
      const f32x4 w0 = R.take((POS0 + p * PE + 2 * kq) % CH_D), w1 = R.take((POS0 + p * PE + 2 * kq + 1) % CH_D);
      if (kq == 0) ch_mfma8<true>(w0, w1, dz[kq], dz[kq], a0, a1);
      else ch_mfma8<false>(w0, w1, dz[kq], dz[kq], a0, a1);
    }
    dout[2 * p] = a0; dout[2 * p + 1] = a1;
    if (p > 0) {
      if (NKG == 1) CH_MFMA_SETTLE();                  // (a pair of this stage is 8 MFMAs: the previous pair's last result may still be in flight)
#pragma unroll
      for (int j = 2 * p - 2; j < 2 * p; ++j) { dout[j] = ch_delu4(dout[j], ap[j]); ch_st4(drs, voff + 64 * j, dsoff, dout[j]); }
    }
  }
  CH_MFMA_SETTLE();
#pragma unroll
  for (int j = NIB - 2; j < NIB; ++j) { dout[j] = ch_delu4(dout[j], ap[j]); ch_st4(drs, voff + 64 * j, dsoff, dout[j]); }
}

// chain_bwd_accreg: dacc += W^T dz, the accumulators held by the caller (the backbone-output gradient, summed over the two heads)
template <int NIB, int NKG, int POS0>
static __device__ __forceinline__ void chain_bwd_accreg(ChainRing& R, const f32x4* dz, f32x4* dacc) {
  constexpr int PE = 2 * NKG;
#pragma unroll
  for (int p = 0; p < NIB / 2; ++p) {
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) {
      const f32x4 w0 = R.take((POS0 + p * PE + 2 * kq) % CH_D), w1 = R.take((POS0 + p * PE + 2 * kq + 1) % CH_D);
      ch_mfma8<false>(w0, w1, dz[kq], dz[kq], dacc[2 * p], dacc[2 * p + 1]);
    }
  }
  CH_MFMA_SETTLE();
}

// chain_bwd_plain: no epilogue (the caller finishes the blocks)
template <int NIB, int NKG, int POS0>
static __device__ __forceinline__ void chain_bwd_plain(ChainRing& R, const f32x4* dz, f32x4* dout) {
  constexpr int PE = 2 * NKG;
#pragma unroll
  for (int p = 0; p < NIB / 2; ++p) {
    f32x4 a0, a1;
#pragma unroll
    for (int kq = 0; kq < NKG; ++kq) {
      const f32x4 w0 = R.take((POS0 + p * PE + 2 * kq) % CH_D), w1 = R.take((POS0 + p * PE + 2 * kq + 1) % CH_D);
      if (kq == 0) ch_mfma8<true>(w0, w1, dz[kq], dz[kq], a0, a1);
      else ch_mfma8<false>(w0, w1, dz[kq], dz[kq], a0, a1);
    }
    dout[2 * p] = a0; dout[2 * p + 1] = a1;
  }
  CH_MFMA_SETTLE();
}

static __device__ __forceinline__ float ch_sum_g(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }       // over the 4 lane groups of a row
static __device__ __forceinline__ float ch_sum_n(float v) {                                                                       // over the 16 rows of a lane group
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

// ntiles 16-row tiles; stream_bytes[p], off_critic: the two streams inside `pack`. Stashes: Bs rows per slab, slab of start
// column c0 at byte offset 4 c0 Bs (both stashes stay below 4 GB: 32-bit buffer offsets).
#ifdef WBC_PPO_TIMING      // development aid: clock stamps of the first actor / critic unit (slots 0.. / 64..)
#define CSTAMP(i) do { if (g_ppo_dbg && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 64)) g_ppo_dbg[(threadIdx.x == 64 ? 64 : 0) + (i)] = clock64(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
extern "C" __global__ void __launch_bounds__(CH_WG, CH_OCC) ppo_chain_kernel(const float* __restrict__ pack, int off_critic, int bytes_actor,
                                                                                 int bytes_critic, PpoBatch Bt, const float* __restrict__ stdp,
                                                                                 float* __restrict__ act_stash, float* __restrict__ dz_stash,
                                                                                 float* __restrict__ dstd_partial, float* __restrict__ loss_partial,
                                                                                 int ntiles) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * (CH_WG / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = unit >> 1;
  const bool critic = (unit & 1) != 0;                 // wave-uniform
  if (tile >= ntiles) return;
  const int g = lane >> 4, n = lane & 15;
  const int B = Bt.B, Bs = Bt.Bs;
  const int row0 = tile * 16, row = row0 + n;
  const bool valid = row < B;
  const size_t src = (size_t)Bt.idx[min(row, B - 1)];
  CSTAMP(0);
#ifdef WBC_PPO_TIMING      // per-unit trace: [128 + 4 unit + {0, 1, 2, 3}] = wall clock (100 MHz) at start / end, shader clock at start / end
  if (g_ppo_dbg && lane == 0) { g_ppo_dbg[128 + 4 * unit] = wall_clock64(); g_ppo_dbg[128 + 4 * unit + 2] = clock64(); }
#define CH_UNIT_END() do { if (g_ppo_dbg && lane == 0) { g_ppo_dbg[128 + 4 * unit + 1] = wall_clock64(); g_ppo_dbg[128 + 4 * unit + 3] = clock64(); } } while (0)
#else
#define CH_UNIT_END() do { } while (0)
#endif
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
  CSTAMP(1);
  if (!critic) {
    // the x slab and the proprio part of the z = [prop, latent] slab (inputs of the priv0 / cbb / bb weight gradients)
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int c = 16 * q + 4 * g;
      if (c < 100) ch_st4(ars, v100 + 64 * q, A_X * slab, zin[q]);
      if (c < PT_NPROP) ch_st4(ars, v100 + 64 * q, A_Z * slab, zin[q]);
    }
    f32x4 h1[4], lat[2];
    chain_fwd<4, 3, 0, true>(R, zin + 4, h1, ars, v64, A_H1 * slab);
    chain_fwd<2, 4, 16 % CH_D, false>(R, h1, lat, ars, 0, 0);
#pragma unroll
    for (int s = 0; s < CH_PAD_PRIV; ++s) (void)R.take((26 + s) % CH_D);
    // latent: the 20-wide slab and the z slab's columns 76..95
    ch_st4(ars, v20, A_LAT * slab, lat[0]);
    ch_st4(ars, v100 + 4 * PT_NPROP, A_Z * slab, lat[0]);
    if (g == 0) { ch_st4(ars, v20 + 64, A_LAT * slab, lat[1]); ch_st4(ars, v100 + 4 * PT_NPROP + 64, A_Z * slab, lat[1]); }
    zin[5] = lat[0]; zin[6] = lat[1];
  }
  CSTAMP(2);
  // backbone (actor: [prop, latent] -> 128; critic: x -> 128): the same code, the stream holds the part's weights
  const int bb_a = (critic ? A_CB : A_BB) * slab, bb_d = (critic ? D_CB : D_BB) * slab;
  chain_fwd<8, 7, 0, true>(R, zin, hbb, ars, v128, bb_a);
#pragma unroll
  for (int s = 0; s < CH_PAD_BB; ++s) (void)R.take((64 + s) % CH_D);
  CSTAMP(3);
  // per-row inputs of the loss terms, requested a whole head chain ahead of their use
  float ld_a[2][4], ld_p[2], ld_q[2];                 // actor: actions of this lane's features, old log-probs, advantages; critic: old values, returns
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 4 * g + r;
      const bool ok = c ? jj < PT_NARM : jj < PT_NLEG;
      ld_a[c][r] = critic ? 0.f : Bt.actions[src * 18 + (ok ? (c ? PT_NLEG + jj : jj) : 0)];
    }
    ld_p[c] = critic ? Bt.old_values[src * 2 + c] : Bt.old_logp[src * 2 + c];
    ld_q[c] = critic ? Bt.returns[src * 2 + c] : Bt.advantages[src * 2 + c];
  }
  // the two heads of the part: 128 -> 128 -> 128 -> (12 | 6 | 1 | 1)
  f32x4 out0 = CH_Z4, out1 = CH_Z4;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int c_l0 = critic ? (h ? A_CA1 : A_CL1) : (h ? A_A1 : A_L1), c_l2 = critic ? (h ? A_CA2 : A_CL2) : (h ? A_A2 : A_L2);
    f32x4 a1[8], a2[8];
    chain_fwd<8, 8, 0, true>(R, hbb, a1, ars, v128, c_l0 * slab);
    CSTAMP(4 + 3 * h);
    chain_fwd<8, 8, 72 % CH_D, true>(R, a1, a2, ars, v128, c_l2 * slab);
    CSTAMP(5 + 3 * h);
    f32x4 o = chain_head<8, 144 % CH_D>(R, a2);
    CSTAMP(6 + 3 * h);
    if (!critic) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = tanhf(o[r]);
      const int w = h ? 8 : PT_NLEG;
      if (4 * g < w) ch_st4(ars, (row * w + 4 * g) * 4, (h ? A_ARM : A_LEG) * slab, o);
    }
    if (h == 0) out0 = o; else out1 = o;
  }
  CSTAMP(10);
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
        d[c][r] = ld_a[c][r] - mu;
        const float term = -(d[c][r] * d[c][r]) / (2.f * sd[c][r] * sd[c][r]) - logf(sd[c][r]) - 0.91893853320467274178f;
        if (ok[c][r]) lp[c] += term;
      }
    lp[0] = ch_sum_g(lp[0]); lp[1] = ch_sum_g(lp[1]);
    const float adv0 = ld_q[0], adv1 = ld_q[1];
    const float mixed[2] = {adv0 + Bt.mixing * adv1, adv1 + Bt.mixing * adv0};               // PPO:199-201
    float dlp[2], surr = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float ratio = expf(lp[c] - ld_p[c]);                               // PPO:202
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
      const float v = c ? out1[0] : out0[0], ov = ld_p[c], Rt = ld_q[c];
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
  CSTAMP(11);
  f32x4 dbb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbb[j] = CH_Z4;
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
    chain_bwd_deriv<8, 1, 0, true>(R, dzo, d2, ars, c_l2 * slab, drs, e_l2 * slab, v128);
    CSTAMP(12 + 3 * h);
    chain_bwd_deriv<8, 8, 8 % CH_D>(R, d2, d1, ars, c_l0 * slab, drs, e_l0 * slab, v128);
    CSTAMP(13 + 3 * h);
    chain_bwd_accreg<8, 8, 72 % CH_D>(R, d1, dbb);
#pragma unroll
    for (int s = 0; s < CH_PAD_BWD; ++s) (void)R.take((136 + s) % CH_D);
    CSTAMP(14 + 3 * h);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { dbb[j] = ch_delu4(dbb[j], hbb[j]); ch_st4(drs, v128 + 64 * j, bb_d, dbb[j]); }      // through the backbone's ELU
  CSTAMP(18);
  if (critic) { CH_UNIT_END(); return; }
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
    chain_bwd_plain<2, 8, 0>(R, dbb, dlat);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dlat[0][r] += sc * dl0[r]; dlat[1][r] += sc * dl1[r]; }
    dlat[0] = ch_delu4(dlat[0], lat[0]);
    dlat[1] = ch_delu4(dlat[1], lat[1]);
    if (g != 0) dlat[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ch_st4(drs, v20, D_LAT * slab, dlat[0]);
    if (g == 0) ch_st4(drs, v20 + 64, D_LAT * slab, dlat[1]);
    f32x4 dh1[4];
    chain_bwd_deriv<4, 2, 16 % CH_D>(R, dlat, dh1, ars, A_H1 * slab, drs, D_H1 * slab, v64);
  }
  CSTAMP(19);
  CH_UNIT_END();
}
