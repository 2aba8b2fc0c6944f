// wbc_sim.hip -- host side of the C-ABI declared in include/wbc_sim.h (libwbc_amd.so).
// Owns the device tensors of one sim (one per GPU / process rank), carves them out of a single
// arena (caller-provided, e.g. a torch allocation, or hipMalloc'ed here) and launches the
// kernels of wbc_step_kernel.hip / wbc_gae_kernel.hip on the caller's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "wbc_device.h"
#include "wbc_stats.h"
#include "wbc_stream_guard.h"

extern "C" __global__ void wbc_step_kernel(const DevTensors* __restrict__ Tp, const DevConst* __restrict__ C, const float* __restrict__ actions, int num_envs,
                                           uint64_t seed, uint64_t step, StepOut so, uint32_t deal);
extern "C" __global__ void wbc_reset_kernel(DevTensors T, const DevConst* __restrict__ C, int num_envs, uint64_t seed, uint64_t step);
extern "C" __global__ void wbc_simulate_kernel(DevTensors T, const DevConst* __restrict__ C, int num_envs);
extern "C" __global__ void wbc_fk_kernel(DevTensors T, const DevConst* __restrict__ C, int num_envs);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// Every entry point runs with the sim's device current and leaves the caller's current device as it found it (a caller that
// keeps the env on cuda:1 without torch.cuda.set_device(1) would otherwise launch into a stream of another device, and the
// setters would switch its device underneath it).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct TensorSpec { int dims[3]; int ndim; int dtype; };
static const TensorSpec kSpecs[WBC_T_COUNT] = {
    {{2, 13, 0}, 2, WBC_F32}, {{20, 2, 0}, 2, WBC_F32}, {{28, 3, 0}, 2, WBC_F32}, {{28, 13, 0}, 2, WBC_F32},
    {{4, 6, 0}, 2, WBC_F32},  {{20, 0, 0}, 1, WBC_F32}, {{860, 0, 0}, 1, WBC_F32}, {{10, 76, 0}, 2, WBC_F32},
    {{4, 18, 0}, 2, WBC_F32}, {{18, 0, 0}, 1, WBC_F32}, {{18, 0, 0}, 1, WBC_F32},  {{20, 0, 0}, 1, WBC_F32},
    {{6, 0, 0}, 1, WBC_F32},  {{3, 0, 0}, 1, WBC_F32},  {{24, 0, 0}, 1, WBC_F32},  {{0, 0, 0}, 0, WBC_F32},
    {{0, 0, 0}, 0, WBC_F32},  {{0, 0, 0}, 0, WBC_I64},  {{0, 0, 0}, 0, WBC_U8},    {{0, 0, 0}, 0, WBC_I64},
    {{WBC_NREW, 0, 0}, 1, WBC_F32}, {{WBC_NMETRIC, 0, 0}, 1, WBC_F32}, {{WBC_NREW, 0, 0}, 1, WBC_F32},  {{WBC_NMETRIC, 0, 0}, 1, WBC_F32},
    {{3, 0, 0}, 1, WBC_F32},  {{3, 0, 0}, 1, WBC_F32},  {{5, 0, 0}, 1, WBC_F32},   {{0, 0, 0}, 0, WBC_F32},
    {{18, 0, 0}, 1, WBC_F32}, {{3, 0, 0}, 1, WBC_F32},  {{0, 0, 0}, 0, WBC_F32},   {{20, 0, 0}, 1, WBC_F32},
    {{2, 0, 0}, 1, WBC_F32},  {{0, 0, 0}, 0, WBC_F32},  {{0, 0, 0}, 0, WBC_F32},  {{WBC_NFEET, 0, 0}, 1, WBC_F32},
    {{WBC_NFEET, 0, 0}, 1, WBC_F32}, {{0, 0, 0}, 0, WBC_F32}};

static size_t spec_elems(const TensorSpec& s) {
  size_t n = 1;
  for (int i = 0; i < s.ndim; ++i) n *= (size_t)s.dims[i];
  return n;
}
static size_t dtype_bytes(int dt) { return dt == WBC_F32 ? 4 : (dt == WBC_I64 ? 8 : 1); }
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct wbc_sim {
  int n = 0, device = 0;
  uint64_t seed = 0;
  int64_t step_counter = 0;
  DevConst hc;
  DevConst* dc = nullptr;
  DevTensors T;
  DevTensors* dT = nullptr;
  void* ptr[WBC_T_COUNT];
  char* arena = nullptr;
  bool own_arena = false;
  size_t arena_bytes = 0;
  int16_t* hf_dev = nullptr;
  // balanced dealing of the step kernel's workgroups (wbc_step_kernel): on when every robot is resident at once and an XCD's range is
  // whole flag words (N a multiple of 512, 2048 to 4096); WBC_NO_DEAL=1 in the environment switches it off (A/B runs)
  char* deal_mem = nullptr;
  bool deal_on = false;
  int64_t step_launches = 0;
};

extern "C" const char* wbc_last_error(void) { return g_err.c_str(); }

// Per-env shape / dtype of tensor `id` without a sim (and without a GPU): lets a binding check its own table.
extern "C" int wbc_tensor_spec(int id, int64_t* dims3, int* ndim, int* dtype) {
  if (id < 0 || id >= WBC_T_COUNT || !dims3 || !ndim || !dtype) return -1;
  for (int i = 0; i < 3; ++i) dims3[i] = kSpecs[id].dims[i];
  *ndim = kSpecs[id].ndim; *dtype = kSpecs[id].dtype;
  return 0;
}

extern "C" size_t wbc_sim_arena_bytes(int num_envs) {
  size_t off = 0;
  for (int t = 0; t < WBC_T_COUNT; ++t) off = align_up(off + spec_elems(kSpecs[t]) * dtype_bytes(kSpecs[t].dtype) * (size_t)num_envs, 256);
  return off + 256;
}

static int build_chains(DevConst& hc) {
  const wbc_model& m = hc.model;
  for (int c = 0; c < WBC_NCHAIN; ++c) { hc.chain_len[c] = 0; for (int d = 0; d < WBC_MAX_DEPTH; ++d) hc.chain_body[c][d] = -1; }
  hc.body_chain[0] = -1; hc.body_depth[0] = 0;
  int nchain = 0;
  for (int i = 1; i < WBC_NB; ++i) {
    const int p = m.parent[i];
    if (p < 0 || p >= i) return -1;
    if (p == 0) {
      if (nchain >= WBC_NCHAIN) return -1;
      hc.body_chain[i] = nchain++; hc.body_depth[i] = 1;
    } else {
      hc.body_chain[i] = hc.body_chain[p]; hc.body_depth[i] = hc.body_depth[p] + 1;
      // serial chain: the parent must be the current tip of its chain
      if (hc.chain_body[hc.body_chain[p]][hc.body_depth[p] - 1] != p || hc.chain_len[hc.body_chain[p]] != hc.body_depth[p]) return -1;
    }
    if (hc.body_depth[i] > WBC_MAX_DEPTH) return -1;
    hc.chain_body[hc.body_chain[i]][hc.body_depth[i] - 1] = i;
    hc.chain_len[hc.body_chain[i]] = hc.body_depth[i];
  }
  if (nchain != WBC_NCHAIN) return -1;
  for (int d = 0; d < WBC_MAX_DEPTH; ++d) {          // the kinematics walk's per-level axis
    int ax = -1;
    for (int c = 0; c < WBC_NCHAIN; ++c) if (hc.chain_body[c][d] >= 0) {
      const int a = m.axis[hc.chain_body[c][d]];
      ax = ax < 0 ? a : (ax == a ? ax : 3);
    }
    hc.lvl_ax[d] = ax < 0 ? 0 : ax;
  }
  {   // euler_from_quat (roll, pitch) of the reset pose
    const float x = hc.cfg.base_init_state[3], y = hc.cfg.base_init_state[4], z = hc.cfg.base_init_state[5], w = hc.cfg.base_init_state[6];
    float sp = 2 * (w * y - z * x);
    sp = sp < -1.f ? -1.f : (sp > 1.f ? 1.f : sp);
    hc.init_rp[0] = atan2f(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
    hc.init_rp[1] = asinf(sp);
  }
  for (int c = 0; c <= WBC_NCHAIN; ++c) for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
    const int b = c < WBC_NCHAIN ? hc.chain_body[c][d] : -1;
    const int dj = b >= 0 ? m.dof[b] : -1;
    hc.chain_arm[c][d] = (dj >= 0 && dj < WBC_NACT) ? hc.cfg.joint_armature[dj] : 0.f;
  }
  {   // sweep groups (DevConst::sweep_pack): chains deeper than three levels first, on the group pairs (0,1), (2,3), ...
    for (int g = 0; g < 8; ++g) hc.sweep_pack[g] = 0x7FFFu | 7u << 15;
    int g = 0;
    auto seg = [&](int c, int d0) {
      uint32_t b = 0;
      for (int l = 0; l < 3; ++l) b |= (uint32_t)((d0 + l < WBC_MAX_DEPTH && hc.chain_body[c][d0 + l] >= 0) ? hc.chain_body[c][d0 + l] : 31) << (5 * l);
      return b;
    };
    for (int c = 0; c < WBC_NCHAIN; ++c) if (hc.chain_len[c] > 3) {
      if (g + 2 > 8) return -1;
      hc.sweep_pack[g] = seg(c, 0) | (uint32_t)c << 15 | 1u << 19;
      hc.sweep_pack[g + 1] = seg(c, 3) | (uint32_t)c << 15 | 1u << 18;
      g += 2;
    }
    for (int c = 0; c < WBC_NCHAIN; ++c) if (hc.chain_len[c] <= 3) {
      if (g + 1 > 8) return -1;
      hc.sweep_pack[g++] = seg(c, 0) | (uint32_t)c << 15;
    }
    hc.sweep_pack[0] |= 1u << 20;       // (group 0 is never a deep half)
  }
  // collision set: every contact has a sphere on a moving body (or on the free box actor); a pair also a partner body
  static_assert(WBC_NCP <= 64 && WBC_NRB_ENV <= 32, "contact sets are 64-bit masks (one lane per contact, ballot of the active ones)");
  if (m.ncp < WBC_NFEET || m.ncp > WBC_NCP) return -1;
  for (int d = 0; d <= WBC_MAX_DEPTH; ++d) hc.depth_cp_mask[d] = 0;
  for (int r = 0; r < 32; ++r) hc.out_cp_mask[r] = hc.out_cp2_mask[r] = 0;
  for (int i = 0; i <= WBC_NB; ++i) hc.body_cp_mask[i] = hc.body_cp2_mask[i] = 0;
  for (int f = 0; f < WBC_NFEET; ++f) hc.foot_cp[f] = hc.foot_cp2[f] = -1;
  hc.box_corner_mask = hc.box_pair_mask = 0;
  hc.cand_self_mask = hc.cand_box_mask = hc.dyn_self_mask = hc.dyn_box_mask = 0;
  for (int i = 0; i < WBC_NSPH; ++i) hc.sph_slot[i] = -1;
  for (int k = 0; k < m.ncp; ++k) {
    const int b = m.cp_body[k], kind = m.cp_kind[k];
    const uint64_t bit = 1ull << k;
    if (kind == WBC_CP_NONE) continue;                          // unused slot
    if (kind == WBC_CP_DYNAMIC) {                               // a free slot of the dynamic pool: the box row's belong to the free box
      if (b == WBC_BOX_BODY) return -1;                         // (the step kernel derives `onbox` from cp_body: a dynamic slot must not claim the box)
      if (k >= 32 && k <= 47) { hc.dyn_box_mask |= bit; hc.box_pair_mask |= bit; } else hc.dyn_self_mask |= bit;
      continue;
    }
    if (b == WBC_BOX_BODY) hc.box_corner_mask |= bit;
    if (kind != WBC_CP_TERRAIN && m.cp_body2[k] == WBC_BOX_BODY) hc.box_pair_mask |= bit;
    // layout rule of the step kernel: the contacts that involve the free box fill the 16-lane row 32..47, and only they do
    if (((hc.box_corner_mask | hc.box_pair_mask) & bit) != 0 ? (k < 32 || k > 47) : (k >= 32 && k <= 47)) return -2;
    if (b < 0 || b > WBC_NB || m.cp_rb[k] < 0 || m.cp_rb[k] >= WBC_NRB_ENV) return -1;
    if ((b == WBC_BOX_BODY) != (m.cp_rb[k] == WBC_BOX_RB)) return -1;
    if (kind != WBC_CP_TERRAIN && kind != WBC_CP_BOX) return -1;
    if (kind == WBC_CP_TERRAIN && b != WBC_BOX_BODY) {          // a robot sphere: its centre is cached under its compact index
      const int si = m.cp_sph[k];
      if (si < 0 || si >= WBC_NSPH || hc.sph_slot[si] >= 0) return -1;
      hc.sph_slot[si] = k;
    }
    int depth = b == WBC_BOX_BODY ? 0 : hc.body_depth[b];       // the box is not part of the tree: its contacts need no sweep level
    hc.body_cp_mask[b] |= bit;
    hc.out_cp_mask[m.cp_rb[k]] |= bit;
    if (kind != WBC_CP_TERRAIN) {
      const int b2 = m.cp_body2[k];
      if (b2 < 0 || b2 > WBC_NB || b2 == b || m.cp_rb2[k] < 0 || m.cp_rb2[k] >= WBC_NRB_ENV) return -1;
      if ((b2 == WBC_BOX_BODY) != (m.cp_rb2[k] == WBC_BOX_RB)) return -1;
      if (b2 != WBC_BOX_BODY && b2 != 0) return -1;             // a static box rides on the root body (its centre cp_a is in frame F) or is the free box
      if (b2 != WBC_BOX_BODY) depth = hc.body_depth[b2] > depth ? hc.body_depth[b2] : depth;
      hc.body_cp2_mask[b2] |= bit;
      hc.out_cp2_mask[m.cp_rb2[k]] |= bit;
      if (b2 == WBC_BOX_BODY) for (int f = 0; f < WBC_NFEET; ++f) if (m.feet_rb[f] == m.cp_rb[k]) {
        if (hc.foot_cp2[f] >= 0) return -1;
        hc.foot_cp2[f] = k;                          // the foot sphere against the box: its sensor sees that force too
      }
    } else {
      for (int f = 0; f < WBC_NFEET; ++f) if (m.feet_rb[f] == m.cp_rb[k]) {
        if (hc.foot_cp[f] >= 0) return -1;          // one sphere per foot: the sensor reads that contact
        hc.foot_cp[f] = k;
      }
    }
    hc.depth_cp_mask[depth] |= bit;
  }
  for (int f = 0; f < WBC_NFEET; ++f) if (hc.foot_cp[f] < 0) return -1;
  // pair descriptors: every lane tests one in the broad phase
  if (m.nlimb < 0 || m.nlimb > WBC_NLIMB) return -1;
  for (int l = 0; l < m.nlimb; ++l) {
    const int s0 = m.limb_s0[l], s1 = m.limb_s1[l];
    if (s0 < 0 || s0 >= WBC_NSPH || s1 < 0 || s1 >= WBC_NSPH || hc.sph_slot[s0] < 0 || hc.sph_slot[s1] < 0) return -1;
    if (m.limb_body[l] < 1 || m.limb_body[l] >= WBC_NB) return -1;
    const int rbs[3] = {m.limb_rb[l], m.limb_rb0[l], m.limb_rb1[l]};
    for (int r = 0; r < 3; ++r) if (rbs[r] < 0 || rbs[r] >= WBC_NRB) return -1;
  }
  bool have_trunk = false;
  for (int j = 0; j < 3; ++j) hc.trunk_c[j] = hc.trunk_h[j] = 0.f;
  for (int k = 0; k < WBC_NCP; ++k) {
    const int pk = m.pr_kind[k], own = (k < m.ncp && m.cp_kind[k] == WBC_CP_TERRAIN && m.cp_body[k] != WBC_BOX_BODY) ? m.cp_sph[k] : 31;
    uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, bsel = 0;
    const uint64_t bit = 1ull << k;
    if (pk == WBC_PR_LIMBS) {
      const int la = m.pr_a[k], lb = m.pr_b[k];
      if (la < 0 || la >= m.nlimb || lb < 0 || lb >= m.nlimb || m.limb_body[la] == m.limb_body[lb]) return -1;
      a0 = m.limb_s0[la]; a1 = m.limb_s1[la]; b0 = m.limb_s0[lb]; b1 = m.limb_s1[lb];
      // the second stage of the broad phase tests the two shafts against WBC_LIMB_RSUM_MAX: a pair with a larger radius sum would
      // silently lose contacts there
      const float ra = fmaxf(m.limb_radius[la], fmaxf(m.limb_cap0[la], m.limb_cap1[la])), rb = fmaxf(m.limb_radius[lb], fmaxf(m.limb_cap0[lb], m.limb_cap1[lb]));
      if (ra + rb > WBC_LIMB_RSUM_MAX + 1e-6f) return -3;
      hc.cand_self_mask |= bit;
    } else if (pk == WBC_PR_SPHERE_BOX) {
      if (m.pr_a[k] < 0 || m.pr_a[k] >= WBC_NSPH || hc.sph_slot[m.pr_a[k]] < 0) return -1;
      a0 = a1 = m.pr_a[k]; bsel = 2;
      hc.cand_box_mask |= bit;
    } else if (pk == WBC_PR_STATIC) {
      if (k >= m.ncp || m.cp_kind[k] != WBC_CP_BOX || m.pr_a[k] < 0 || m.pr_a[k] >= WBC_NSPH || hc.sph_slot[m.pr_a[k]] < 0) return -1;
      const int ss = hc.sph_slot[m.pr_a[k]];                   // the pair's sphere must be that robot sphere
      if (m.cp_body[ss] != m.cp_body[k] || m.cp_rb[ss] != m.cp_rb[k]) return -1;
      a0 = a1 = m.pr_a[k]; bsel = m.cp_body2[k] == WBC_BOX_BODY ? 2 : 1;
      if (bsel == 1) {                                         // the box on the root body: one box for all such pairs (the trunk)
        for (int j = 0; j < 3; ++j) {
          if (have_trunk && (hc.trunk_c[j] != m.cp_a[k][j] || hc.trunk_h[j] != m.cp_b[k][j])) return -1;
          hc.trunk_c[j] = m.cp_a[k][j]; hc.trunk_h[j] = m.cp_b[k][j];
        }
        have_trunk = true;
      } else for (int j = 0; j < 3; ++j) if (m.cp_a[k][j] != 0.f || m.cp_b[k][j] != m.box_half) return -1;
    } else if (pk != WBC_PR_NONE) return -1;
    if (k < m.ncp && m.cp_kind[k] == WBC_CP_BOX && pk != WBC_PR_STATIC) return -1;
    for (int j = 0; j < 8; ++j) hc.cand_rad[k][j] = 0.f;
    hc.cand_rbs[k] = hc.cand_bodies[k] = 0;
    if (pk == WBC_PR_LIMBS) {
      const int la = m.pr_a[k], lb = m.pr_b[k];
      const float r[6] = {m.limb_radius[la], m.limb_cap0[la], m.limb_cap1[la], m.limb_radius[lb], m.limb_cap0[lb], m.limb_cap1[lb]};
      const int rbs[6] = {m.limb_rb[la], m.limb_rb0[la], m.limb_rb1[la], m.limb_rb[lb], m.limb_rb0[lb], m.limb_rb1[lb]};
      for (int j = 0; j < 6; ++j) { hc.cand_rad[k][j] = r[j]; hc.cand_rbs[k] |= (uint32_t)rbs[j] << (5 * j); }
      hc.cand_bodies[k] = (uint32_t)m.limb_body[la] | (uint32_t)m.limb_body[lb] << 8;
    } else if (pk == WBC_PR_SPHERE_BOX) {
      const int ss = hc.sph_slot[m.pr_a[k]];
      hc.cand_rad[k][0] = m.cp_radius[ss];
      hc.cand_rbs[k] = (uint32_t)m.cp_rb[ss];
      hc.cand_bodies[k] = (uint32_t)m.cp_body[ss] | (uint32_t)WBC_BOX_BODY << 8;
    }
    uint32_t rcode = 0;
    if (pk != WBC_PR_NONE) {
      const float q = m.pr_reach[k] / WBC_REACH_STEP;
      const int code = (int)(q + 0.5f);
      if (code < 1 || code > 8 || fabsf(q - (float)code) > 1e-3f) return -1;       // pr_reach is a multiple of WBC_REACH_STEP (abi.quantise_reach)
      rcode = (uint32_t)(code - 1);
    }
    hc.pr_pack[k] = (uint32_t)pk | (uint32_t)own << 2 | a0 << 7 | a1 << 12 | b0 << 17 | b1 << 22 | bsel << 27 | rcode << 29;
  }
  static_assert(WBC_NB <= 31 && WBC_NDOF <= 32 && WBC_MAX_DEPTH * 5 <= 32, "bit packing of the chain tables");
  for (int c = 0; c <= WBC_NCHAIN; ++c) {
    uint32_t pb = 0, pd = 0, pa = 0;
    for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
      const int i = (c < WBC_NCHAIN) ? hc.chain_body[c][d] : -1;
      const int ii = i < 0 ? 0 : i;
      pb |= (uint32_t)(i < 0 ? 31 : i) << (5 * d);
      pd |= (uint32_t)(m.dof[ii] < 0 ? 0 : m.dof[ii]) << (5 * d);
      pa |= (uint32_t)(m.axis[ii] < 0 ? 0 : m.axis[ii]) << (2 * d);
    }
    hc.chain_pack_body[c] = pb; hc.chain_pack_dof[c] = pd; hc.chain_pack_ax[c] = pa;
  }
  for (int i = 0; i < WBC_NB; ++i)
    hc.body_pack[i] = (uint32_t)(m.axis[i] < 0 ? 0 : m.axis[i]) | (uint32_t)(m.dof[i] < 0 ? 0 : m.dof[i]) << 2;
  return 0;
}

extern "C" int wbc_sim_create(const wbc_model* model, const wbc_task_cfg* cfg, int num_envs, int hip_device, uint64_t seed, void* arena,
                              size_t arena_bytes, wbc_sim** out) {
  if (!model || !cfg || !out || num_envs <= 0) return fail(-1, "wbc_sim_create: bad arguments");
  if (num_envs > (1 << 22)) return fail(-1, "wbc_sim_create: num_envs above 2^22 (the kernels index a tensor's rows with 32-bit element offsets)");
  if (model->ncp < WBC_NFEET || model->ncp > WBC_NCP) return fail(-1, "wbc_sim_create: model.ncp must be in [4, WBC_NCP]");
  if (!(model->box_half > 0.f) || !(model->box_mass > 0.f)) return fail(-1, "wbc_sim_create: the box actor needs a positive size and mass");
  DeviceGuard dg(hip_device);
  wbc_sim* s = new wbc_sim();
  s->n = num_envs; s->device = hip_device; s->seed = seed;
  memset(&s->hc, 0, sizeof(DevConst));
  s->hc.model = *model; s->hc.cfg = *cfg;
  if (const int rc = build_chains(s->hc)) {
    delete s;
    return fail(-1, rc == -3 ? "wbc_sim_create: a candidate limb pair's radii add up to more than WBC_LIMB_RSUM_MAX (the broad phase's second stage would miss its contacts)" :
                    rc == -2 ? "wbc_sim_create: contact slots 32..47 are the free box's row (its corners, robot spheres against it) and nothing else"
                             : "wbc_sim_create: topology must be a root with 5 serial chains of depth <= 6, contacts on valid bodies");
  }
  const size_t need = wbc_sim_arena_bytes(num_envs);
  if (arena) {
    if (arena_bytes < need) { delete s; return fail(-1, "wbc_sim_create: arena too small"); }
    s->arena = (char*)arena;
  } else {
    hipError_t e = hipMalloc((void**)&s->arena, need);
    if (e != hipSuccess) { delete s; return fail(-2, std::string("hipMalloc arena: ") + hipGetErrorString(e)); }
    s->own_arena = true;
  }
  s->arena_bytes = need;
  size_t off = (256 - ((uintptr_t)s->arena & 255)) & 255;
  for (int t = 0; t < WBC_T_COUNT; ++t) {
    s->ptr[t] = s->arena + off;
    off = align_up(off + spec_elems(kSpecs[t]) * dtype_bytes(kSpecs[t].dtype) * (size_t)num_envs, 256);
  }
  HIP_OK(hipMemset(s->arena, 0, need));
  DevTensors& T = s->T;
  T.root = (float*)s->ptr[WBC_T_ROOT_STATES]; T.dof = (float*)s->ptr[WBC_T_DOF_STATE]; T.contact = (float*)s->ptr[WBC_T_NET_CONTACT_FORCE];
  T.rb = (float*)s->ptr[WBC_T_RIGID_BODY_STATE]; T.sensor = (float*)s->ptr[WBC_T_FORCE_SENSOR]; T.torques = (float*)s->ptr[WBC_T_TORQUES];
  T.obs = (float*)s->ptr[WBC_T_OBS_BUF]; T.obs_hist = (float*)s->ptr[WBC_T_OBS_HISTORY]; T.act_hist = (float*)s->ptr[WBC_T_ACTION_HISTORY];
  T.actions = (float*)s->ptr[WBC_T_ACTIONS]; T.last_actions = (float*)s->ptr[WBC_T_LAST_ACTIONS]; T.last_dof_vel = (float*)s->ptr[WBC_T_LAST_DOF_VEL];
  T.last_root_vel = (float*)s->ptr[WBC_T_LAST_ROOT_VEL]; T.commands = (float*)s->ptr[WBC_T_COMMANDS]; T.goal = (float*)s->ptr[WBC_T_GOAL_STATE];
  T.rew = (float*)s->ptr[WBC_T_REW_BUF]; T.arm_rew = (float*)s->ptr[WBC_T_ARM_REW_BUF]; T.reset_buf = (int64_t*)s->ptr[WBC_T_RESET_BUF];
  T.time_out = (uint8_t*)s->ptr[WBC_T_TIME_OUT_BUF]; T.ep_len = (int64_t*)s->ptr[WBC_T_EPISODE_LENGTH]; T.ep_sums = (float*)s->ptr[WBC_T_EPISODE_SUMS];
  T.met_sums = (float*)s->ptr[WBC_T_METRIC_SUMS]; T.ep_sums_done = (float*)s->ptr[WBC_T_EPISODE_SUMS_DONE];
  T.met_sums_done = (float*)s->ptr[WBC_T_METRIC_SUMS_DONE]; T.base_lin_vel = (float*)s->ptr[WBC_T_BASE_LIN_VEL];
  T.base_ang_vel = (float*)s->ptr[WBC_T_BASE_ANG_VEL]; T.mass_params = (float*)s->ptr[WBC_T_MASS_PARAMS]; T.friction = (float*)s->ptr[WBC_T_FRICTION];
  T.motor = (float*)s->ptr[WBC_T_MOTOR_STRENGTH]; T.origins = (float*)s->ptr[WBC_T_ENV_ORIGINS]; T.box_dy = (float*)s->ptr[WBC_T_BOX_DELTA_Y];
  T.body_params = (float*)s->ptr[WBC_T_BODY_PARAMS]; T.reset_travel = (float*)s->ptr[WBC_T_RESET_TRAVEL];
  T.box_mass = (float*)s->ptr[WBC_T_BOX_MASS]; T.box_timer = (float*)s->ptr[WBC_T_BOX_SLEEP_TIMER];
  T.feet_air_time = (float*)s->ptr[WBC_T_FEET_AIR_TIME]; T.last_contacts = (float*)s->ptr[WBC_T_LAST_CONTACTS];
  T.dropped = (float*)s->ptr[WBC_T_DROPPED_HITS];
  {
    const size_t fl = 2 * 8 * 2 * WBC_DEAL_WORDS * sizeof(uint64_t);
    HIP_OK(hipMalloc((void**)&s->deal_mem, fl));
    HIP_OK(hipMemset(s->deal_mem, 0, fl));
    T.deal_flags = (uint64_t*)s->deal_mem;
    const char* off = getenv("WBC_NO_DEAL");
    // (from 2048 envs on: with one robot per SIMD -- 1024 envs -- there is nothing to balance, and the rank-select at the head of every
    // wave costs 0.7 us of the launch's dependent chain: 81.5 against 82.3 us, tools/r06_runs/r06_gpu32.sh)
    s->deal_on = num_envs % 512 == 0 && num_envs >= 2048 && num_envs / 512 <= WBC_DEAL_WORDS && !(off && off[0] == '1');
  }
  // defaults: identity quaternions, unit friction/motor strength, nominal inertias, sane goal timers
  {
    const int n = num_envs;
    std::vector<float> root((size_t)n * 26, 0.f), fr(n, 1.f), ms((size_t)n * WBC_NACT, 1.f), bp((size_t)n * 20), goal((size_t)n * 24, 0.f);
    std::vector<float> bm(n, model->box_mass);
    const int g = model->gripper_body;
    for (int i = 0; i < n; ++i) {
      root[(size_t)i * 26 + 6] = 1.f; root[(size_t)i * 26 + 19] = 1.f;
      float* b = &bp[(size_t)i * 20];
      b[0] = model->mass[0];
      for (int k = 0; k < 3; ++k) b[1 + k] = model->com[0][k];
      for (int k = 0; k < 6; ++k) b[4 + k] = model->inertia[0][k];
      b[10] = model->mass[g];
      for (int k = 0; k < 3; ++k) b[11 + k] = model->com[g][k];
      for (int k = 0; k < 6; ++k) b[14 + k] = model->inertia[g][k];
      goal[(size_t)i * 24 + 22] = 100.f; goal[(size_t)i * 24 + 23] = 150.f;
    }
    HIP_OK(hipMemcpy(T.root, root.data(), root.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(T.friction, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(T.motor, ms.data(), ms.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(T.body_params, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(T.goal, goal.data(), goal.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(T.box_mass, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMalloc((void**)&s->dT, sizeof(DevTensors)));
  HIP_OK(hipMemcpy(s->dT, &s->T, sizeof(DevTensors), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc((void**)&s->dc, sizeof(DevConst)));
  HIP_OK(hipMemcpy(s->dc, &s->hc, sizeof(DevConst), hipMemcpyHostToDevice));
  *out = s;
  return 0;
}

extern "C" int wbc_sim_destroy(wbc_sim* s) {
  if (!s) return 0;
  DeviceGuard dg(s->device);
  if (s->own_arena && s->arena) (void)hipFree(s->arena);
  if (s->dc) (void)hipFree(s->dc);
  if (s->deal_mem) (void)hipFree(s->deal_mem);
  if (s->dT) (void)hipFree(s->dT);
  if (s->hf_dev) (void)hipFree(s->hf_dev);
  delete s;
  return 0;
}

extern "C" int wbc_sim_get_tensor(wbc_sim* s, int id, void** dev_ptr, int64_t shape[4], int* ndim, int* dtype) {
  if (!s || id < 0 || id >= WBC_T_COUNT) return fail(-1, "wbc_sim_get_tensor: bad id");
  *dev_ptr = s->ptr[id];
  shape[0] = s->n;
  for (int i = 0; i < kSpecs[id].ndim; ++i) shape[1 + i] = kSpecs[id].dims[i];
  *ndim = 1 + kSpecs[id].ndim;
  *dtype = kSpecs[id].dtype;
  return 0;
}

// composite of two rigid bodies, all about their own coms (host, double)
static void merge2(double m1, const double* c1, const double* I1, double m2, const double* c2, const double* I2, double* m, double* c, double* I) {
  *m = m1 + m2;
  for (int k = 0; k < 3; ++k) c[k] = (m1 * c1[k] + m2 * c2[k]) / *m;
  auto shift = [&](double mm, const double* cc, const double* Ii, double* out) {
    const double d[3] = {cc[0] - c[0], cc[1] - c[1], cc[2] - c[2]};
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    out[0] += Ii[0] + mm * (dd - d[0] * d[0]); out[1] += Ii[1] + mm * (dd - d[1] * d[1]); out[2] += Ii[2] + mm * (dd - d[2] * d[2]);
    out[3] += Ii[3] - mm * d[0] * d[1]; out[4] += Ii[4] - mm * d[0] * d[2]; out[5] += Ii[5] - mm * d[1] * d[2];
  };
  for (int k = 0; k < 6; ++k) I[k] = 0;
  shift(m1, c1, I1, I);
  shift(m2, c2, I2, I);
}

extern "C" int wbc_sim_set_env_params(wbc_sim* s, const float* friction, const float* base_dmass, const float* base_dcom,
                                      const float* gripper_dmass, const float* motor_strength, const float* env_origins,
                                      const float* box_delta_y, const float* traj_timesteps, const float* traj_total_timesteps,
                                      const float* box_dmass) {
  if (!s) return fail(-1, "wbc_sim_set_env_params: null sim");
  DeviceGuard dg(s->device);
  const int n = s->n;
  const wbc_model& m = s->hc.model;
  if (friction) HIP_OK(hipMemcpy(s->T.friction, friction, (size_t)n * 4, hipMemcpyHostToDevice));
  if (motor_strength) HIP_OK(hipMemcpy(s->T.motor, motor_strength, (size_t)n * WBC_NACT * 4, hipMemcpyHostToDevice));
  if (env_origins) HIP_OK(hipMemcpy(s->T.origins, env_origins, (size_t)n * 3 * 4, hipMemcpyHostToDevice));
  if (box_delta_y) HIP_OK(hipMemcpy(s->T.box_dy, box_delta_y, (size_t)n * 4, hipMemcpyHostToDevice));
  if (box_dmass) {
    std::vector<float> bm(n);
    for (int i = 0; i < n; ++i) {
      bm[i] = m.box_mass + box_dmass[i];
      if (!(bm[i] > 0.f)) return fail(-1, "wbc_sim_set_env_params: box mass must stay positive");
    }
    HIP_OK(hipMemcpy(s->T.box_mass, bm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  if (base_dmass || base_dcom || gripper_dmass) {
    std::vector<float> bp((size_t)n * 20), mp((size_t)n * 5);
    for (int i = 0; i < n; ++i) {
      const double dm = base_dmass ? base_dmass[i] : 0.0, gm = gripper_dmass ? gripper_dmass[i] : 0.0;
      double dc[3] = {0, 0, 0};
      if (base_dcom) for (int k = 0; k < 3; ++k) dc[k] = base_dcom[(size_t)i * 3 + k];
      mp[(size_t)i * 5] = (float)dm; for (int k = 0; k < 3; ++k) mp[(size_t)i * 5 + 1 + k] = (float)dc[k]; mp[(size_t)i * 5 + 4] = (float)gm;
      double rc[3], rI[6], pc[3], pI[6], M, Cc[3], I[6];
      for (int k = 0; k < 3; ++k) { rc[k] = m.base_rest_com[k]; pc[k] = m.base_piece_com[k] + dc[k]; }
      for (int k = 0; k < 6; ++k) { rI[k] = m.base_rest_inertia[k]; pI[k] = m.base_piece_inertia[k]; }
      merge2(m.base_rest_mass, rc, rI, m.base_piece_mass + dm, pc, pI, &M, Cc, I);
      float* b = &bp[(size_t)i * 20];
      b[0] = (float)M; for (int k = 0; k < 3; ++k) b[1 + k] = (float)Cc[k]; for (int k = 0; k < 6; ++k) b[4 + k] = (float)I[k];
      for (int k = 0; k < 3; ++k) { rc[k] = m.grip_rest_com[k]; pc[k] = m.grip_piece_com[k]; }
      for (int k = 0; k < 6; ++k) { rI[k] = m.grip_rest_inertia[k]; pI[k] = m.grip_piece_inertia[k]; }
      merge2(m.grip_rest_mass, rc, rI, m.grip_piece_mass + gm, pc, pI, &M, Cc, I);
      b[10] = (float)M; for (int k = 0; k < 3; ++k) b[11 + k] = (float)Cc[k]; for (int k = 0; k < 6; ++k) b[14 + k] = (float)I[k];
    }
    HIP_OK(hipMemcpy(s->T.body_params, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(s->T.mass_params, mp.data(), mp.size() * 4, hipMemcpyHostToDevice));
  }
  if (traj_timesteps && traj_total_timesteps) {
    std::vector<float> goal((size_t)n * 24);
    HIP_OK(hipMemcpy(goal.data(), s->T.goal, goal.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { goal[(size_t)i * 24 + 22] = traj_timesteps[i]; goal[(size_t)i * 24 + 23] = traj_total_timesteps[i]; }
    HIP_OK(hipMemcpy(s->T.goal, goal.data(), goal.size() * 4, hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int wbc_sim_set_heightfield(wbc_sim* s, const int16_t* heights, int rows, int cols, float hs, float vs, float tx, float ty, float tz) {
  if (!s) return fail(-1, "wbc_sim_set_heightfield: null sim");
  DeviceGuard dg(s->device);
  HIP_OK(hipDeviceSynchronize());
  if (s->hf_dev) { (void)hipFree(s->hf_dev); s->hf_dev = nullptr; }
  s->hc.hf = nullptr;
  if (heights) {
    if (rows < 2 || cols < 2) return fail(-1, "wbc_sim_set_heightfield: need at least a 2x2 grid");
    HIP_OK(hipMalloc((void**)&s->hf_dev, (size_t)rows * cols * sizeof(int16_t)));
    HIP_OK(hipMemcpy(s->hf_dev, heights, (size_t)rows * cols * sizeof(int16_t), hipMemcpyHostToDevice));
    s->hc.hf = s->hf_dev; s->hc.hf_rows = rows; s->hc.hf_cols = cols; s->hc.hf_hs = hs; s->hc.hf_vs = vs;
    s->hc.hf_t[0] = tx; s->hc.hf_t[1] = ty; s->hc.hf_t[2] = tz;
  }
  HIP_OK(hipMemcpy(s->dc, &s->hc, sizeof(DevConst), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int wbc_sim_set_curriculum(wbc_sim* s, const wbc_curriculum* cur) {
  if (!s || !cur) return fail(-1, "wbc_sim_set_curriculum: bad arguments");
  DeviceGuard dg(s->device);
  s->hc.cur = *cur;
  // plain (synchronising) copy: the previous step's kernel must not see a half-written table
  HIP_OK(hipDeviceSynchronize());      // a step kernel still in flight on a non-blocking stream reads DevConst.cur
  HIP_OK(hipMemcpy((char*)s->dc + offsetof(DevConst, cur), &s->hc.cur, sizeof(wbc_curriculum), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int wbc_sim_step_rollout(wbc_sim* s, const float* actions_dev, float* obs_out_dev, const float* values_dev, float gamma,
                                    float* out_rewards_dev, uint8_t* out_dones_dev, void* stream) {
  if (!s || !actions_dev) return fail(-1, "wbc_sim_step: bad arguments");
  DeviceGuard dg(s->device);
  if ((out_rewards_dev != nullptr) != (out_dones_dev != nullptr) || (out_rewards_dev && !values_dev))
    return fail(-1, "wbc_sim_step_rollout: values, out_rewards and out_dones go together");
  s->step_counter += 1;
  const StepOut so{obs_out_dev, values_dev, out_rewards_dev, out_dones_dev, gamma};
  const uint32_t deal = s->deal_on ? (2u | (uint32_t)(s->step_launches & 1)) : 0u;
  hipLaunchKernelGGL(wbc_step_kernel, dim3(8 * ((s->n + 7) / 8)), dim3(64), 0, (hipStream_t)stream, s->dT, s->dc, actions_dev, s->n, s->seed, (uint64_t)s->step_counter, so, deal);
  HIP_OK(hipGetLastError());
  s->step_launches += 1;
  return 0;
}
extern "C" int wbc_sim_step_to(wbc_sim* s, const float* actions_dev, float* obs_out_dev, void* stream) {
  return wbc_sim_step_rollout(s, actions_dev, obs_out_dev, nullptr, 0.f, nullptr, nullptr, stream);
}
extern "C" int wbc_sim_step(wbc_sim* s, const float* actions_dev, void* stream) { return wbc_sim_step_to(s, actions_dev, nullptr, stream); }

extern "C" int wbc_sim_reset_all(wbc_sim* s, void* stream) {
  if (!s) return fail(-1, "wbc_sim_reset_all: null sim");
  DeviceGuard dg(s->device);
  hipLaunchKernelGGL(wbc_reset_kernel, dim3(s->n), dim3(64), 0, (hipStream_t)stream, s->T, s->dc, s->n, s->seed, (uint64_t)s->step_counter);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int wbc_sim_set_dof_forces(wbc_sim* s, const float* torques_dev, void* stream) {
  if (!s || !torques_dev) return fail(-1, "wbc_sim_set_dof_forces: bad arguments");
  DeviceGuard dg(s->device);
  if (torques_dev != s->T.torques)
    HIP_OK(hipMemcpyAsync(s->T.torques, torques_dev, (size_t)s->n * WBC_NDOF * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

extern "C" int wbc_sim_simulate(wbc_sim* s, void* stream) {
  if (!s) return fail(-1, "wbc_sim_simulate: null sim");
  DeviceGuard dg(s->device);
  hipLaunchKernelGGL(wbc_simulate_kernel, dim3(s->n), dim3(64), 0, (hipStream_t)stream, s->T, s->dc, s->n);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int wbc_sim_set_root_state(wbc_sim* s, const float* root_dev, void* stream) {
  if (!s || !root_dev) return fail(-1, "wbc_sim_set_root_state: bad arguments");
  DeviceGuard dg(s->device);
  if (root_dev != s->T.root) HIP_OK(hipMemcpyAsync(s->T.root, root_dev, (size_t)s->n * 26 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
extern "C" int wbc_sim_set_dof_state(wbc_sim* s, const float* dof_dev, void* stream) {
  if (!s || !dof_dev) return fail(-1, "wbc_sim_set_dof_state: bad arguments");
  DeviceGuard dg(s->device);
  if (dof_dev != s->T.dof) HIP_OK(hipMemcpyAsync(s->T.dof, dof_dev, (size_t)s->n * 40 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

__global__ void copy_rows_indexed(float* dst, const float* src, const int32_t* ids, int n_ids, int row, int num_envs) {
  const int i = blockIdx.x;
  if (i >= n_ids) return;
  const int e = ids[i];
  if (e < 0 || e >= num_envs) return;
  for (int k = threadIdx.x; k < row; k += blockDim.x) dst[(size_t)e * row + k] = src[(size_t)e * row + k];
}
extern "C" int wbc_sim_set_root_state_indexed(wbc_sim* s, const float* root_dev, const int32_t* env_ids_dev, int n, void* stream) {
  if (!s || !root_dev || !env_ids_dev) return fail(-1, "wbc_sim_set_root_state_indexed: bad arguments");
  DeviceGuard dg(s->device);
  if (n > 0 && root_dev != s->T.root) hipLaunchKernelGGL(copy_rows_indexed, dim3(n), dim3(64), 0, (hipStream_t)stream, s->T.root, root_dev, env_ids_dev, n, 26, s->n);
  HIP_OK(hipGetLastError());
  return 0;
}
extern "C" int wbc_sim_set_dof_state_indexed(wbc_sim* s, const float* dof_dev, const int32_t* env_ids_dev, int n, void* stream) {
  if (!s || !dof_dev || !env_ids_dev) return fail(-1, "wbc_sim_set_dof_state_indexed: bad arguments");
  DeviceGuard dg(s->device);
  if (n > 0 && dof_dev != s->T.dof) hipLaunchKernelGGL(copy_rows_indexed, dim3(n), dim3(64), 0, (hipStream_t)stream, s->T.dof, dof_dev, env_ids_dev, n, 40, s->n);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int wbc_sim_refresh_dof_state(wbc_sim*) { return 0; }
extern "C" int wbc_sim_refresh_root_state(wbc_sim*) { return 0; }
extern "C" int wbc_sim_refresh_net_contact_force(wbc_sim*) { return 0; }
extern "C" int wbc_sim_refresh_force_sensor(wbc_sim*) { return 0; }
extern "C" int wbc_sim_refresh_rigid_body_state(wbc_sim* s, void* stream) {
  if (!s) return fail(-1, "wbc_sim_refresh_rigid_body_state: null sim");
  DeviceGuard dg(s->device);
  hipLaunchKernelGGL(wbc_fk_kernel, dim3(s->n), dim3(64), 0, (hipStream_t)stream, s->T, s->dc, s->n);
  HIP_OK(hipGetLastError());
  return 0;
}

// extras["episode"] of reset_idx (WG:743-754) + the runner's episode deques as a side job (wbc_stats.h): one workgroup per
// statistics column (+ 1 for the tracker), here as a launch of its own.
#define STATS_THREADS 1024
static __global__ void __launch_bounds__(STATS_THREADS) episode_stats_kernel(wbc_side_job J) { side_job_block<STATS_THREADS>(J, blockIdx.x); }

extern "C" int wbc_sim_episode_stats_job(wbc_sim* s, float scale, const float* prev, float* out, float* track_state, int track_cap,
                                         wbc_side_job* job) {
  if (!s || !out || !job) return fail(-1, "wbc_sim_episode_stats_job: null argument");
  if (track_state && track_cap <= 0) return fail(-1, "wbc_sim_episode_stats_job: cap must be positive");
  job->ep_done = s->T.ep_sums_done; job->met_done = s->T.met_sums_done; job->reset_buf = s->T.reset_buf; job->prev = prev;
  job->rew = s->T.rew; job->arm_rew = s->T.arm_rew; job->out = out; job->track_state = track_state;
  job->n = s->n; job->track_cap = track_cap; job->nblocks = WBC_NREW + WBC_NMETRIC + (track_state ? 1 : 0); job->scale = scale;
  return 0;
}

extern "C" int wbc_side_job_run(const wbc_side_job* job, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!job || !job->out || job->nblocks <= 0) return fail(-1, "wbc_side_job_run: empty job");
  hipLaunchKernelGGL(episode_stats_kernel, dim3(job->nblocks), dim3(STATS_THREADS), 0, (hipStream_t)stream, *job);
  return hipGetLastError() == hipSuccess ? 0 : fail(-2, "episode_stats_kernel launch failed");
}

extern "C" int wbc_sim_episode_stats_track(wbc_sim* s, float scale, const float* prev, float* out, float* track_state, int track_cap,
                                           void* stream) {
  wbc_side_job job;
  const int rc = wbc_sim_episode_stats_job(s, scale, prev, out, track_state, track_cap, &job);
  if (rc) return rc;
  DeviceGuard dg(s->device);
  return wbc_side_job_run(&job, stream);
}

extern "C" int wbc_sim_episode_stats(wbc_sim* s, float scale, const float* prev, float* out, void* stream) {
  return wbc_sim_episode_stats_track(s, scale, prev, out, nullptr, 0, stream);
}

// internal (wbc_arm_kernel.hip): the tensors the arm-dynamics pre-pass reads
extern "C" int wbc_sim_internal_arm_inputs(wbc_sim* s, const DevConst** hc, const float** root, const float** dofs, const float** body_params,
                                           const float** mass_params, int* n) {
  if (!s) return -1;
  *hc = &s->hc; *root = s->T.root; *dofs = s->T.dof; *body_params = s->T.body_params; *mass_params = s->T.mass_params; *n = s->n;
  return 0;
}

extern "C" int wbc_sim_get_step_counter(wbc_sim* s, int64_t* out) { if (!s || !out) return fail(-1, "null"); *out = s->step_counter; return 0; }
extern "C" int wbc_sim_set_step_counter(wbc_sim* s, int64_t v) { if (!s) return fail(-1, "null"); s->step_counter = v; return 0; }

// ---- asset file (host only): model + task configuration + curricula + names, see include/wbc_sim.h --------------------------------
struct wbc_asset {
  wbc_model model;
  wbc_task_cfg cfg;
  wbc_curriculum cur[2];
  int ndof = 0, nrb = 0;
  std::vector<std::string> dof_names, rb_names;
};
static const char kAssetMagic[10] = "WBCASSET1";
extern "C" int wbc_asset_load(const char* path, wbc_asset** out) {
  if (!path || !out) return fail(-1, "wbc_asset_load: bad arguments");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(-2, std::string("wbc_asset_load: cannot open ") + path);
  char magic[10];
  uint32_t hdr[5];
  wbc_asset* a = new wbc_asset();
  bool ok = fread(magic, 1, 10, f) == 10 && memcmp(magic, kAssetMagic, 10) == 0 && fread(hdr, 4, 5, f) == 5;
  if (ok && (hdr[0] != sizeof(wbc_model) || hdr[1] != sizeof(wbc_task_cfg) || hdr[2] != sizeof(wbc_curriculum))) {
    fclose(f); delete a;
    return fail(-3, "wbc_asset_load: the file was written for another version of the wbc_model / wbc_task_cfg / wbc_curriculum structs");
  }
  ok = ok && hdr[3] == WBC_NDOF && hdr[4] == WBC_NRB;
  ok = ok && fread(&a->model, sizeof(wbc_model), 1, f) == 1 && fread(&a->cfg, sizeof(wbc_task_cfg), 1, f) == 1 &&
       fread(a->cur, sizeof(wbc_curriculum), 2, f) == 2;
  if (ok) {
    a->ndof = (int)hdr[3]; a->nrb = (int)hdr[4];
    char name[64];
    for (int i = 0; ok && i < a->ndof + a->nrb; ++i) {
      ok = fread(name, 1, 64, f) == 64;
      name[63] = 0;
      (i < a->ndof ? a->dof_names : a->rb_names).push_back(name);
    }
  }
  fclose(f);
  if (!ok) { delete a; return fail(-2, std::string("wbc_asset_load: not a wbc asset file: ") + path); }
  *out = a;
  return 0;
}
extern "C" void wbc_asset_free(wbc_asset* a) { delete a; }
extern "C" int wbc_asset_dof_count(const wbc_asset* a) { return a ? a->ndof : -1; }
extern "C" int wbc_asset_rigid_body_count(const wbc_asset* a) { return a ? a->nrb : -1; }
extern "C" const char* wbc_asset_dof_name(const wbc_asset* a, int i) { return (a && i >= 0 && i < a->ndof) ? a->dof_names[i].c_str() : nullptr; }
extern "C" const char* wbc_asset_rigid_body_name(const wbc_asset* a, int i) { return (a && i >= 0 && i < a->nrb) ? a->rb_names[i].c_str() : nullptr; }
extern "C" int wbc_asset_dof_properties(const wbc_asset* a, float* lower, float* upper, float* velocity, float* effort) {
  if (!a) return fail(-1, "null asset");
  for (int i = 0; i < a->ndof; ++i) {
    if (lower) lower[i] = a->model.q_lower[i];
    if (upper) upper[i] = a->model.q_upper[i];
    if (velocity) velocity[i] = a->model.qd_limit[i];
    if (effort) effort[i] = a->model.effort[i];
  }
  return 0;
}
extern "C" const wbc_model* wbc_asset_model(const wbc_asset* a) { return a ? &a->model : nullptr; }
extern "C" const wbc_task_cfg* wbc_asset_task_cfg(const wbc_asset* a) { return a ? &a->cfg : nullptr; }
extern "C" const wbc_curriculum* wbc_asset_curriculum(const wbc_asset* a, int which) { return (a && (which == 0 || which == 1)) ? &a->cur[which] : nullptr; }

// sizes of the ABI structs, checked against the ctypes mirrors by the CPU tests
extern "C" void wbc_abi_sizes(int* out) { out[0] = (int)sizeof(wbc_model); out[1] = (int)sizeof(wbc_task_cfg); out[2] = (int)sizeof(wbc_curriculum); }
