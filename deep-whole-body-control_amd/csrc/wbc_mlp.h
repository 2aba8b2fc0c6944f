// wbc_mlp.h -- shared pieces of the fused ActorCritic kernels (gfx950, fp32 MFMA 32x32x2): the parameter
// pointer table, the packed-weight layout and the table-driven dense-layer loop used by both the rollout
// inference kernel (wbc_policy_kernel.hip) and the PPO update kernels (wbc_ppo_kernel.hip).
//
// Two measured facts shape this file (profiles/ and DESIGN.md section 5):
//  * GEMMs read their B operand from a PRE-PACKED copy of the weights in fragment order: for layer l, group kg of
//    four k-pairs and 32-column block cb, lane L's four B values (k-pairs 4kg..4kg+3 of one
//    v_mfma_f32_32x32x2_f32 each: W[cb*32 + (L&31)][2*kb + (L>>5)]) are one float4, and the 64 lanes' float4s are
//    contiguous: a layer's operands are 16 coalesced 1-KB loads per wave, requested one layer ahead of their use.
//    (64 single-dword loads per layer defeated the prefetch: s_waitcnt vmcnt counts at most 63 outstanding
//    operations, so waiting for the current layer's operands also waited for the next layer's.)
//    The backward (dgrad) GEMMs use a second, transposed pack of the same weights.
//  * The layer chain is a LOOP over a descriptor table, not 16 unrolled template instances: fully unrolled, the
//    inference kernel was 51 KB of straight-line code (the update kernel 123 KB) against a 64 KB instruction
//    cache, and ran at ~300 cycles per 64-byte instruction line -- 3.5x slower than its MFMA time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_ROWS 32          // rows (envs / minibatch samples) per workgroup
#define PT_THREADS 256
#define LDA 129             // activation row stride in LDS (odd: conflict-free b32 fragment reads)
#define PT_NPROP 76
#define PT_NPRIV 24
#define PT_NOBS 860
#define PT_NLEG 12
#define PT_NARM 6
#define NLAYERS 16

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct PolicyParams {       // device pointers to the torch parameters (weight [out,in] row-major, bias [out])
  const float *priv0_w, *priv0_b, *priv2_w, *priv2_b;
  const float *bb_w, *bb_b;
  const float *leg0_w, *leg0_b, *leg2_w, *leg2_b, *leg4_w, *leg4_b;
  const float *arm0_w, *arm0_b, *arm2_w, *arm2_b, *arm4_w, *arm4_b;
  const float *cbb_w, *cbb_b;
  const float *cleg0_w, *cleg0_b, *cleg2_w, *cleg2_b, *cleg4_w, *cleg4_b;
  const float *carm0_w, *carm0_b, *carm2_w, *carm2_b, *carm4_w, *carm4_b;
  const float* std;         // [18]
};

// layer order = PolicyParams order
enum { L_PRIV0 = 0, L_PRIV2, L_BB, L_LEG0, L_LEG2, L_LEG4, L_ARM0, L_ARM2, L_ARM4, L_CBB, L_CLEG0, L_CLEG2, L_CLEG4, L_CARM0, L_CARM2, L_CARM4 };
__host__ __device__ constexpr int layer_out(int l) {
  return l == L_PRIV0 ? 64 : l == L_PRIV2 ? 20 : l == L_LEG4 ? 12 : l == L_ARM4 ? 6 : (l == L_CLEG4 || l == L_CARM4) ? 1 : 128;
}
__host__ __device__ constexpr int layer_in(int l) { return l == L_PRIV0 ? 24 : l == L_PRIV2 ? 64 : l == L_BB ? 96 : l == L_CBB ? 100 : 128; }
__host__ __device__ constexpr int layer_nblk(int l) { return (layer_out(l) + 31) / 32; }
// forward pack: k over the inputs, 32-column blocks over the outputs; groups of 4 k-pairs (zero padded)
__host__ __device__ constexpr int layer_kg(int l) { return (layer_in(l) / 2 + 3) / 4; }
__host__ __device__ constexpr int layer_pack_floats(int l) { return layer_kg(l) * layer_nblk(l) * 256; }
__host__ __device__ constexpr int layer_pack_off(int l) { return l == 0 ? 0 : layer_pack_off(l - 1) + layer_pack_floats(l - 1); }
#define WPACK_FWD_FLOATS (layer_pack_off(NLAYERS - 1) + layer_pack_floats(NLAYERS - 1))
// transposed pack (dgrad: dIn = dOut * W): k over the outputs, 32-column blocks over the inputs
__host__ __device__ constexpr int layer_kgT(int l) { return ((layer_out(l) + 1) / 2 + 3) / 4; }
__host__ __device__ constexpr int layer_nblkT(int l) { return (layer_in(l) + 31) / 32; }
__host__ __device__ constexpr int layer_packT_floats(int l) { return layer_kgT(l) * layer_nblkT(l) * 256; }
__host__ __device__ constexpr int layer_packT_off(int l) { return l == 0 ? WPACK_FWD_FLOATS : layer_packT_off(l - 1) + layer_packT_floats(l - 1); }
__host__ __device__ constexpr int layer_bias_off(int l) { return l == 0 ? 0 : layer_bias_off(l - 1) + layer_out(l - 1); }
#define WPACK_WEIGHT_FLOATS (layer_packT_off(NLAYERS - 1) + layer_packT_floats(NLAYERS - 1))
#define WPACK_BIAS_FLOATS (layer_bias_off(NLAYERS - 1) + layer_out(NLAYERS - 1))
#define WPACK_FLOATS (WPACK_WEIGHT_FLOATS + WPACK_BIAS_FLOATS)      // packed weights, then all biases back to back

enum { ACT_NONE = 0, ACT_ELU = 1, ACT_TANH = 2 };







// ---- one MFMA operand set: 64 registers per lane ---------------------------------------------------------
// The B fragments of a wave's 32-column block, k-pairs 0..kb-1, as 16 float4 loads of 4 k-pairs each. `base` points
// at the layer's pack (uniform), `lane_off` is this lane's float4 within a k-group; consecutive k-groups are `stride4`
// float4s apart. All 16 loads are issued
// UNCONDITIONALLY (k-groups past kb re-read group 0): with loads under predicates the compiler cannot know how many
// are outstanding and its s_waitcnt for the current layer's operands also drains the next layer's prefetch.
template <int NW>
static __device__ __forceinline__ void load_operands(float (&w)[NW], const float4* __restrict__ ubase, int lane_off, int stride4, int kb, int kg0 = 0) {
#pragma unroll
  for (int kg = 0; kg < 16; ++kg) {
    const float4* pu = ubase + (kg * 4 < kb ? kg0 + kg : 0) * stride4;    // uniform part (scalar registers); lane_off is the per-lane part
    const float4 v = pu[lane_off];
    w[kg * 4] = v.x; w[kg * 4 + 1] = v.y; w[kg * 4 + 2] = v.z; w[kg * 4 + 3] = v.w;
  }
}


// development aid (-DWBC_PPO_TIMING): workgroup 0 / thread 0 stamps inside the forward layers, slots 32 + 4*layer + i
#ifdef WBC_PPO_TIMING
static __device__ long long* g_mlp_dbg = nullptr;
#define LSTAMP(l, i) do { if (g_mlp_dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_mlp_dbg[32 + 4 * (l) + (i)] = clock64(); } while (0)
#else
#define LSTAMP(l, i) do { } while (0)
#endif


// Run forward layer d with its fragments in `w`. smem = LDS base (floats). Outputs go to LDS and, if d.scol >= 0, to
// the slab-major stash (see FwdDesc::sw; `num_rows` = total rows of the stash) for valid rows. Ends with a barrier.
struct NoHook { __device__ __forceinline__ void operator()() const {} };


// ==== 16-row tiles: v_mfma_f32_16x16x4_f32 ===================================================================================
// A: lane L supplies A[row = L & 15][k-slot g = L >> 4]; B: B[g][col = L & 15]; C/D: 4 registers, D[row = 4 g + r][col = L & 15].
// A wave owns 32 output columns as two 16-column halves h. The reduction index is cut into chunks of 32 (zero-padded): in
// chunk c, k-step j (0..7) and slot g stand for k = 32 c + 8 g + j, so that a lane's 8 A operands of a chunk are 8 consecutive
// LDS floats (two ds_read_b128, conflict-free at row stride 132). Pack ("pack16"), per 32-column block cb contiguous:
// float4 [cb][kp][lane] = W[cb*32 + 16h + (L&15)][k(ks, g)] for (ks, h) = (2kp,0), (2kp,1), (2kp+1,0), (2kp+1,1); the transposed
// pack ("packT16", dgrad: k over the layer's outputs, columns over its inputs) holds W[k(ks, g)][cb*32 + 16h + (L&15)].
// Blob layout: [pack16 of all layers | packT16 of all layers | biases].
#define R16 16
#define LD16 132
// Workgroup barrier that orders LDS traffic only. __syncthreads() also waits for every outstanding global access
// (s_waitcnt vmcnt(0)): at each layer that drained the operand prefetch and stalled on the acknowledgement of the stash
// stores (measured: ~8 k cycles per 128x128 layer against a 2 k-cycle MFMA chain). Use __syncthreads() wherever another
// wave's GLOBAL writes are read back.
// Which 32-column block a wave works on. (Rotating this assignment with the workgroup's dispatch round, so that the extra
// work of the narrow layers -- heads, latent -- is not always on wave 0, changed nothing: 586 vs 577 us per minibatch.)
static __device__ __forceinline__ int wave_role() { return threadIdx.x >> 6; }
#define LBAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
using f32x4 = __attribute__((ext_vector_type(4))) float;

__host__ __device__ constexpr int l16_floats(int l) { return (layer_in(l) + 31) / 32 * 4 * layer_nblk(l) * 256; }
__host__ __device__ constexpr int l16T_floats(int l) { return (layer_out(l) + 31) / 32 * 4 * layer_nblkT(l) * 256; }
__host__ __device__ constexpr int l16_sum(int l) { return l < 0 ? 0 : l16_sum(l - 1) + l16_floats(l); }
__host__ __device__ constexpr int l16T_sum(int l) { return l < 0 ? 0 : l16T_sum(l - 1) + l16T_floats(l); }
constexpr int kWpack16Fwd = l16_sum(NLAYERS - 1);            // constexpr variables: a constexpr recursion evaluated in device code
constexpr int kWpack16Bias = kWpack16Fwd + l16T_sum(NLAYERS - 1);   // would be compiled as a recursive call
#define WPACK16_FWD_FLOATS kWpack16Fwd
#define WPACK16_BIAS_OFF kWpack16Bias
#define WPACK16_FLOATS (WPACK16_BIAS_OFF + WPACK_BIAS_FLOATS)
#define WPACK16_OFF ((WPACK_FLOATS + 3) / 4 * 4)          // where the 16-row blob sits inside a wbc_policy_pack buffer

struct Desc16 {               // one forward layer on a 16-row tile
  int woff, boff, nch, nblk, n, in_off, out_off, ldo, act;      // pack16 offset, bias offset, k chunks (of 32 inputs; the LDS columns up to
                                                                 // 32 nch must be finite), 32-col blocks, outputs, LDS offsets (in_off % 4 == 0)
  int scol, sw;                                                  // stash slab (start column, width) or scol < 0
};
struct Tab16 { Desc16 l[NLAYERS]; };
struct Pack16Table { int n[NLAYERS], k[NLAYERS], off[NLAYERS], offT[NLAYERS], boff[NLAYERS]; };

static inline Desc16 make_desc16(int l, int in_off, int out_off, int scol, int sw = 0) {
  const int act = (l == L_LEG4 || l == L_ARM4) ? ACT_TANH : ((l == L_CLEG4 || l == L_CARM4) ? ACT_NONE : ACT_ELU);
  const bool head = (l == L_LEG4 || l == L_ARM4 || l == L_CLEG4 || l == L_CARM4);       // heads write outv[., 21]
  return Desc16{l16_sum(l - 1), layer_bias_off(l), (layer_in(l) + 31) / 32, layer_nblk(l), layer_out(l), in_off, out_off, head ? 21 : LD16, act, scol, sw};
}
static inline Pack16Table make_pack16_table() {
  Pack16Table t;
  for (int l = 0; l < NLAYERS; ++l) {
    t.n[l] = layer_out(l); t.k[l] = layer_in(l); t.off[l] = l16_sum(l - 1); t.offT[l] = WPACK16_FWD_FLOATS + l16T_sum(l - 1); t.boff[l] = layer_bias_off(l);
  }
  return t;
}

// grid = (blocks, NLAYERS, 1 or 2): z = 0 forward pack (+ biases), z = 1 transposed pack. `blob` 16-byte aligned.
static __global__ void wbc_pack16_kernel(PolicyParams P, Pack16Table T, float* __restrict__ blob) {
  const int l = blockIdx.y;
  const bool tr = blockIdx.z != 0;
  const float* W = reinterpret_cast<const float* const*>(&P)[2 * l];
  const float* bsrc = reinterpret_cast<const float* const*>(&P)[2 * l + 1];
  const int N = T.n[l], K = T.k[l];
  const int kdim = tr ? N : K, cdim = tr ? K : N;                // k runs over kdim, columns over cdim
  const int nblk = (cdim + 31) / 32, nkp = (kdim + 31) / 32 * 4;
  const int off = tr ? T.offT[l] : T.off[l], total = nkp * nblk * 256;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int j = e & 3, lane = (e >> 2) & 63, frag = e >> 8;
    const int kp = frag % nkp, cb = frag / nkp;
    const int ks = 2 * kp + (j >> 1), h = j & 1;
    const int c = cb * 32 + 16 * h + (lane & 15), k = 32 * (ks >> 3) + 8 * (lane >> 4) + (ks & 7);
    float v = 0.f;
    if (c < cdim && k < kdim) v = tr ? W[(size_t)k * K + c] : W[(size_t)c * K + k];
    blob[off + e] = v;
  }
  if (blockIdx.x == 0 && !tr)
    for (int e = threadIdx.x; e < N; e += blockDim.x) blob[WPACK16_BIAS_OFF + T.boff[l] + e] = bsrc[e];
}

// 16 float4 of one operand set for column block `cb` (wave-uniform) of a pack with `nch` chunks: scalar base + lane offset
// addressing, all 16 loads unconditional (chunks past nch re-read chunk 0; see load_operands)
static __device__ __forceinline__ void load_ops16(float (&w)[66], const float* __restrict__ pack, int nch, int cb) {
  const int lane = threadIdx.x & 63;
  const int nkp = nch * 4;
  const float4* base = reinterpret_cast<const float4*>(pack) + __builtin_amdgcn_readfirstlane(cb * nkp * 64);
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) {
    const float4 v = (base + (kp < nkp ? kp * 64 : 0))[lane];
    w[4 * kp] = v.x; w[4 * kp + 1] = v.y; w[4 * kp + 2] = v.z; w[4 * kp + 3] = v.w;
  }
}

// operands of forward layer d for this wave + the two bias values of its column halves (w[64], w[65])
static __device__ __forceinline__ void load16(float (&w)[66], const Desc16& d, const float* __restrict__ pack16, const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63, wave = wave_role();
  load_ops16(w, pack16 + d.woff, d.nch, wave < d.nblk ? wave : 0);
  const int c0 = wave * 32 + (lane & 15);
  w[64] = bias[d.boff + (c0 < d.n ? c0 : 0)];
  w[65] = bias[d.boff + (c0 + 16 < d.n ? c0 + 16 : 0)];
}

// A tile has MB blocks of 16 rows ("M-blocks") that share one operand set: acc[m][h] += sum over nch chunks of A_m * w for
// both column halves h. ap = in + (L & 15) * LD16 + 8 (L >> 4) (16-byte aligned), M-block m at ap + 16 m LD16. The next
// chunk's A operands are read before this chunk's MFMAs are issued (a chunk past the end re-reads the last one).
// HALVES = 1: only the first 16 columns exist (the heads: 12, 6 or 1 outputs), the second half's MFMAs are skipped.
template <int MB, int HALVES = 2>
static __device__ __forceinline__ void mfma_chain16(const float* ap, const float (&w)[66], int nch, f32x4 (&acc)[MB][2]) {
  float4 a[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    a[m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * LD16);
    a[m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * LD16 + 4);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < nch) {
      const float* np = ap + 32 * (c + 1 < nch ? c + 1 : c);
      float4 n[MB][2];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        n[m][0] = *reinterpret_cast<const float4*>(np + m * 16 * LD16);
        n[m][1] = *reinterpret_cast<const float4*>(np + m * 16 * LD16 + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const float4 q = a[m][j >> 2];
          const float av = (j & 3) == 0 ? q.x : ((j & 3) == 1 ? q.y : ((j & 3) == 2 ? q.z : q.w));
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[2 * (c * 8 + j)], acc[m][0], 0, 0, 0);
          if (HALVES == 2) acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[2 * (c * 8 + j) + 1], acc[m][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MB; ++m) { a[m][0] = n[m][0]; a[m][1] = n[m][1]; }
    }
  }
}

template <int ACT>
static __device__ __forceinline__ float act16(float x) {
  if (ACT == ACT_ELU) {          // branch-free; exp via v_exp_f32 (abs error <= 1 ulp(1.0); the derivative uses the stored value)
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.f;     // (+inf for large x: not selected)
    return x > 0.f ? x : e;
  }
  if (ACT == ACT_TANH) return tanhf(x);
  return x;
}

// LDS addresses of a lane's outputs, computed BEFORE the operand prefetch is issued: address arithmetic after it (the
// compiler uses 64-bit multiply-adds whose unused high half may alias a register that a prefetch load is still writing) made
// the epilogue wait for every outstanding load. lds16: step between M-blocks.
struct Epi16 { uint32_t lds[4], lds16; };
static __device__ __forceinline__ Epi16 epilogue16_offsets(const Desc16& d) {
  const int lane = threadIdx.x & 63, wave = wave_role();
  const int c0 = wave * 32 + (lane & 15), rb = 4 * (lane >> 4);
  Epi16 e;
#pragma unroll
  for (int r = 0; r < 4; ++r) e.lds[r] = (uint32_t)(d.out_off + (rb + r) * d.ldo + c0);
  e.lds16 = (uint32_t)(16 * d.ldo);
  return e;
}

// The tile's output of layer d, LDS -> its stash slab, row-major with float4 (512 contiguous bytes per 32 lanes) instead
// of 4-byte stores straight from the MFMA result layout. Called one layer late -- after the next layer's MFMA chain and
// right before its operand prefetch: the output buffer is still intact (three-buffer rotation), and a wave's vector-memory
// operations always go [stores][loads], so the s_waitcnt vmcnt in front of an MFMA chain counts only loads (stores issued
// after the loads would be lane-predicated: the compiler has to assume they were not issued, and the chain's last waits
// turn into waits for store acknowledgements). Slabs are padded to whole tiles: every tile row is stored.
template <int MB>
static __device__ __forceinline__ void stash_copy16(const Desc16& d, const float* smem, float* __restrict__ stash, int row0, int slab_rows) {
  float* sbase = stash + (size_t)d.scol * slab_rows + (size_t)row0 * d.sw;      // [tile rows][sw]
  const int tid = threadIdx.x;
  if (d.sw == 128 && d.ldo == LD16) {
#pragma unroll
    for (int j = 0; j < 2 * MB; ++j) {                    // 16 MB rows x 32 float4 over 256 threads
      const int row = (tid >> 5) + 8 * j, c = (tid & 31) * 4;
      *reinterpret_cast<float4*>(sbase + row * 128 + c) = *reinterpret_cast<const float4*>(smem + d.out_off + row * LD16 + c);
    }
  } else {                                                // narrow layers (64, 20, 12, 6 columns): element-wise
    const int total = 16 * MB * d.n;
    for (int e = tid; e < total; e += blockDim.x) {
      const int row = e / d.n, c = e - row * d.n;
      sbase[row * d.sw + c] = smem[d.out_off + row * d.ldo + c];
    }
  }
}

// epilogue of one layer: act(acc + bias) to LDS. (Starting the accumulators from the bias instead saves the adds but moved
// the 20-step update test 3e-4 away from the eager path.)
template <int ACT, int MB>
static __device__ __forceinline__ void epilogue16(const Desc16& d, const Epi16& e, const f32x4 (&acc)[MB][2], float b0, float b1, float* smem,
                                                  bool ok0, bool ok1) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h ? ok1 : ok0) {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) smem[e.lds[r] + m * e.lds16 + 16 * h] = act16<ACT>(acc[m][h][r] + (h ? b1 : b0));
      }
    }
  }
}

// Forward layer d on a tile of MB x 16 rows: out = act(in W^T + b) to LDS. copy_prev: the output of the previous layer
// `dprev` still has to go to the stash (stash_copy16; the caller does it itself after the last layer). `after_mfma` runs
// once `w` is no longer read (one call site: the operand registers it refills keep their place). Ends with a barrier.
// (Descriptors by reference into the kernel-argument table: taking their address would copy the table to scratch.)
template <int MB, typename Hook = NoHook>
static __device__ __forceinline__ void run16(float (&w)[66], const Desc16& d, float* smem, float* __restrict__ stash, int row0, int slab_rows,
                                             const Desc16& dprev, bool copy_prev, Hook after_mfma = Hook(), int dbg_l = 0) {
  const int lane = threadIdx.x & 63, wave = wave_role();
  const bool active = wave < d.nblk;
  LSTAMP(dbg_l, 0);
  f32x4 acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  if (active) {
    const float* ap = smem + d.in_off + (lane & 15) * LD16 + 8 * (lane >> 4);
    if (d.n > 16) mfma_chain16<MB, 2>(ap, w, d.nch, acc);
    else mfma_chain16<MB, 1>(ap, w, d.nch, acc);
  }
  const float b0 = w[64], b1 = w[65];
  const Epi16 e = epilogue16_offsets(d);
  if (copy_prev) stash_copy16<MB>(dprev, smem, stash, row0, slab_rows);
  LSTAMP(dbg_l, 1);
  after_mfma();
  {
    const int c0 = wave * 32 + (lane & 15);
    const bool ok0 = active && c0 < d.n, ok1 = active && c0 + 16 < d.n;
    if (d.act == ACT_ELU) epilogue16<ACT_ELU, MB>(d, e, acc, b0, b1, smem, ok0, ok1);
    else if (d.act == ACT_TANH) epilogue16<ACT_TANH, MB>(d, e, acc, b0, b1, smem, ok0, ok1);
    else epilogue16<ACT_NONE, MB>(d, e, acc, b0, b1, smem, ok0, ok1);
  }
  LSTAMP(dbg_l, 2);
  LBAR();
  LSTAMP(dbg_l, 3);
}
