// wbc_device.h -- device-side constants and small math shared by the HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wbc_sim.h"

#define WBC_NCHAIN 5
#define WBC_REACH_STEP 0.04f

// Everything the kernels read that is constant over a launch, resident in HBM and read through
// the scalar/constant path (uniform indices) or L1 (lane-dependent indices).
struct DevConst {
  wbc_model model;
  wbc_task_cfg cfg;
  wbc_curriculum cur;
  int32_t chain_body[WBC_NCHAIN][WBC_MAX_DEPTH];   // moving body at (chain, depth-1), -1 if none
  int32_t chain_len[WBC_NCHAIN];
  int32_t body_chain[WBC_NB];                      // chain of each body (-1 root)
  int32_t body_depth[WBC_NB];                      // 0 root, 1.. along the chain
  int32_t foot_cp[WBC_NFEET];                      // the contacts whose forces foot sensor f reports: the foot sphere against the terrain
  int32_t foot_cp2[WBC_NFEET];                     // ... and against the free box (-1: none)
  uint64_t depth_cp_mask[WBC_MAX_DEPTH + 1];       // contacts whose deepest TREE body (sphere's or partner's) sits at chain depth d
  // bit-packed copies the step kernel keeps in registers / LDS (built by build_chains)
  uint32_t chain_pack_body[WBC_NCHAIN + 1];        // 6 x 5 bits: body at depth d (31 = none); row WBC_NCHAIN = idle lanes
  uint32_t chain_pack_dof[WBC_NCHAIN + 1];         // 6 x 5 bits: dof of that body
  uint32_t chain_pack_ax[WBC_NCHAIN + 1];          // 6 x 2 bits: joint axis
  // The solver sweeps' layout: eight groups of 8 lanes, each walks a segment of up to three levels of one chain with its operands in
  // registers. A chain deeper than three levels takes two groups of one 16-lane DPP row (even group: levels 1..3, odd group: 4..6;
  // the hand-over between them is one DPP row shift). 3 x 5 bits: the segment's bodies (31 = none) | chain << 15 (7 = unused group)
  // | 1 << 18: the deep half of its chain | 1 << 19: its chain has a deep half (in the next group) | 1 << 20: the group that adds the
  // root's own contact wrench to the sum over the groups and stores the root's response
  uint32_t sweep_pack[8];
  int32_t lvl_ax[WBC_MAX_DEPTH];                    // the joint axis every chain with a body at depth d+1 shares (0 / 1 / 2), 3 where they differ
  float init_rp[2];                                // roll, pitch of cfg.base_init_state's quaternion (what a freshly reset env observes)
  float chain_arm[WBC_NCHAIN + 1][WBC_MAX_DEPTH];  // cfg.joint_armature of the joint at (chain, depth-1); 0 where there is none (row WBC_NCHAIN: idle)
  uint64_t out_cp_mask[32];                        // [rb]: contacts whose force net_contact_force row rb receives ...
  uint64_t out_cp2_mask[32];                       // ... and those it receives with the opposite sign (partner of a pair)
  uint64_t body_cp_mask[WBC_NB + 1];               // the same two sets per moving body (the sweeps' wrench gather); entry
  uint64_t body_cp2_mask[WBC_NB + 1];              // WBC_BOX_BODY is the free box actor
  uint64_t box_corner_mask, box_pair_mask;         // the box's corner contacts; the robot spheres against the box (static pairs + its dynamic slots)
  // self-collision broad phase (wbc_model pr_*): lanes whose descriptor is a candidate limb pair / a candidate robot sphere against
  // the free box; the dynamic slots their hits are promoted into (outside / inside the box row)
  uint64_t cand_self_mask, cand_box_mask, dyn_self_mask, dyn_box_mask;
  // per lane: kind (2 bits) | this lane's own sphere, 31 = none (5) | sphere indices a0, a1, b0, b1 of the descriptor's two bounding
  // centres (4 x 5) | where the second centre comes from: 0 the spheres b0, b1; 1 the static pair's box centre cp_a (a box on the root
  // body: frame F); 2 the free box's centre (2 bits) | bounding reach in units of WBC_REACH_STEP, minus one (3 bits)
  uint32_t pr_pack[WBC_NCP];
  float trunk_c[3], trunk_h[3];                    // the static pairs' box on the root body (the trunk): centre and half extents, frame F
  int32_t sph_slot[WBC_NSPH];                      // the terrain slot of each robot sphere
  // what a promoted candidate needs, per candidate lane, so that the promoted lane fetches it in ONE round of independent loads (a launch
  // ends with its slowest wave): radii (limb a: shaft, end 0, end 1; limb b: the same | a sphere-vs-box candidate: [0] = the sphere's
  // radius), rigid bodies (6 x 5 bits in the same order | [0] = the sphere's), moving bodies (a | b << 8)
  float cand_rad[WBC_NCP][8];
  uint32_t cand_rbs[WBC_NCP], cand_bodies[WBC_NCP];
  uint32_t body_pack[WBC_NB];                      // axis | dof << 2
  // heightfield (optional)
  const int16_t* hf;
  int32_t hf_rows, hf_cols;
  float hf_hs, hf_vs, hf_t[3];
};

// Per-env device tensors (AoS per env: one wave reads an env's block with consecutive lanes).
// Optional extra outputs of one step (wbc_sim_step_rollout): where the observation rows go instead of obs_buf, and the rollout
// storage's reward / done slots of this transition (PPO.process_env_step, rsl_rl/algorithms/ppo.py:129-141: the time-out
// bootstrap rewards += gamma * values * time_outs needs the values PPO.act computed for this step).
struct StepOut { float* obs; const float* values; float* rewards; uint8_t* dones; float gamma; };

struct DevTensors {
  float* root;        // [N,2,13]
  float* dof;         // [N,20,2]
  float* contact;     // [N,28,3]
  float* rb;          // [N,28,13]
  float* sensor;      // [N,4,6]
  float* torques;     // [N,20]
  float* obs;         // [N,860]
  float* obs_hist;    // [N,10,76]
  float* act_hist;    // [N,4,18]
  float* actions;     // [N,18]
  float* last_actions;
  float* last_dof_vel;
  float* last_root_vel;
  float* commands;    // [N,3]
  float* goal;        // [N,24]
  float* rew;
  float* arm_rew;
  int64_t* reset_buf;
  uint8_t* time_out;
  int64_t* ep_len;
  float* ep_sums;     // [N,WBC_NREW]
  float* met_sums;    // [N,10]
  float* ep_sums_done;
  float* met_sums_done;
  float* base_lin_vel;
  float* base_ang_vel;
  float* mass_params; // [N,5]
  float* friction;    // [N]
  float* motor;       // [N,18]
  float* origins;     // [N,3]
  float* box_dy;      // [N]
  float* body_params; // [N,20]
  float* reset_travel; // [N,2]
  float* box_mass;    // [N]
  float* box_timer;   // [N]
  float* feet_air_time; // [N,4]
  float* last_contacts; // [N,4]
  float* dropped;       // [N]
  // balanced dealing of the step kernel's workgroups (wbc_step_kernel's header comment; allocated apart from the tensor arena)
  uint64_t* deal_flags; // [2][8][2][WBC_DEAL_WORDS]: per launch parity and XCD, two bit planes with one bit per env of the XCD's range: "expected in contact" / "expected to lie on the ground or to touch itself"
};
#define WBC_DEAL_WORDS 8

#define WBC_PI 3.14159265358979323846f

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// counter-based uniform in [0,1): bit-identical to oracle/wbc_oracle.c rng_u01
__device__ __forceinline__ float rng_u01(uint64_t seed, uint64_t env, uint64_t step, uint64_t slot) {
  uint64_t h = mix64(seed + env * 0x9E3779B97F4A7C15ULL);
  h = mix64(h + step * 0xD1B54A32D192ED03ULL + slot * 0x8CB92BA72F3D8DD7ULL);
  return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float urange(float lo, float hi, float u) { return (hi - lo) * u + lo; }
__device__ __forceinline__ float rng_range(float lo, float hi, uint64_t seed, uint64_t env, uint64_t step, uint64_t slot) {
  return urange(lo, hi, rng_u01(seed, env, step, slot));
}
enum {
  SLOT_GOAL_ORN = 0, SLOT_GOAL_SPHERE = 3, SLOT_CMD = 33, SLOT_PUSH = 35, SLOT_RESET_DOF = 37,
  SLOT_RESET_XY = 57, SLOT_RESET_VEL = 59, SLOT_RESET_CMD = 65, SLOT_RESET_GOAL_ORN = 67,
  SLOT_RESET_GOAL_SPHERE = 70
};

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// row-major 3x3 in memory
__device__ __forceinline__ f3 mat_mul(const float* M, f3 v) {
  return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z);
}
__device__ __forceinline__ f3 matT_mul(const float* M, f3 v) {
  return mk3(M[0] * v.x + M[3] * v.y + M[6] * v.z, M[1] * v.x + M[4] * v.y + M[7] * v.z, M[2] * v.x + M[5] * v.y + M[8] * v.z);
}
__device__ __forceinline__ float dot6(const float* a, const float* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
__device__ __forceinline__ void quat_to_mat(const float* q, float* R) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) {
  float x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  float y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  float z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
__device__ __forceinline__ f3 quat_rotate_inverse(const float* q, f3 v) {
  float w = q[3];
  f3 qv = mk3(q[0], q[1], q[2]);
  f3 a = v * (2 * w * w - 1);
  f3 b = cross(qv, v) * (w * 2);
  f3 c = qv * (2 * dot(qv, v));
  return a - b + c;
}
// atan2 with the cephes single-precision arctangent (two range reductions + a degree-4 polynomial in x^2: ~2 ulp), ~30
// instructions instead of libm's ~100: the scalar task logic of a step (base / gripper Euler angles, cart2sphere) evaluates eight
// inverse trigonometric functions on ONE lane, where every instruction costs a full wavefront issue slot.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx);                 // in [0, 1]
  t = (mx == 0.f) ? 0.f : t;
  const bool mid = t > 0.4142135623730950f;                 // tan(pi / 8)
  const float u = mid ? (t - 1.f) * __builtin_amdgcn_rcpf(t + 1.f) : t;
  const float z = u * u;
  float r = fmaf(fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f) * z, u, u);
  r += mid ? 0.78539816339744831f : 0.f;
  r = (ay > ax) ? 1.57079632679489662f - r : r;             // atan(mn / mx) -> atan(|y| / |x|)
  r = (x < 0.f) ? 3.14159265358979324f - r : r;
  return copysignf(r, y);
}
// asin(s) = atan2(s, sqrt((1 - s)(1 + s)))
__device__ __forceinline__ float fast_asinf(float s) {
  return fast_atan2f(s, __builtin_amdgcn_sqrtf(fmaxf((1.f - s) * (1.f + s), 0.f)));
}
// exp through the hardware exp2 (relative error ~1e-6 for the reward arguments, |x| < 20)
__device__ __forceinline__ float fast_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ f3 euler_from_quat(const float* q) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float sp = 2 * (w * y - z * x);
  sp = fminf(fmaxf(sp, -1.f), 1.f);
  return mk3(fast_atan2f(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), fast_asinf(sp), fast_atan2f(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)));
}
// sin and cos with Cody-Waite reduction by pi/2 and the cephes single-precision polynomials: ~1 ulp for |x| < 1e3
// (joint angles, goal angles), 25 instructions instead of libm's ~150 (its large-argument path is inlined everywhere)
__device__ __forceinline__ void fast_sincosf(float x, float* sp, float* cp) {
  const float k = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188e-8f, r);
  const float z = r * r;
  const float sn = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cs = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(z, -0.5f, 1.f));
  const int q = (int)k;
  const float s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
  *sp = (q & 2) ? -s0 : s0;
  *cp = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ f3 sphere2cart(f3 s) {
  float sy, cy, sz, cz;
  fast_sincosf(s.y, &sy, &cy);
  fast_sincosf(s.z, &sz, &cz);
  return mk3(s.x * cy * cz, s.x * cy * sz, s.x * sy);
}
__device__ __forceinline__ f3 cart2sphere(f3 c) {
  const float h2 = c.x * c.x + c.y * c.y;
  const float l = __builtin_amdgcn_sqrtf(h2 + c.z * c.z);
  return mk3(l, fast_atan2f(c.z, __builtin_amdgcn_sqrtf(h2)), fast_atan2f(c.y, c.x));    // asin(z / l) = atan2(z, |xy|)
}
__device__ __forceinline__ float wrap_to_pi(float a) {
  const float two_pi = 2 * WBC_PI;
  float t = a + WBC_PI;
  t = t - two_pi * floorf(t * 0.15915494309189535f);       // (a reciprocal multiply: the IEEE division is a 10-instruction sequence)
  return t - WBC_PI;
}
__device__ __forceinline__ float lerp_torch(float a, float b, float w) {
  return (w < 0.5f) ? a + w * (b - a) : b - (b - a) * (1 - w);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
