/*
 * wbc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the widowGo1 rollout step, one environment at a time, used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker
 * for the HIP kernels in deep-whole-body-control_amd/csrc/. Built twice from this file:
 * REAL=double (the spec) and REAL=float (rounding mirror of the fp32 kernels).
 *
 * What it follows (reference paths; WG = legged_gym/legged_gym/envs/widowGo1/widowGo1.py):
 *   - everything AROUND physics is a line-by-line restatement of WG: step WG:1156-1199,
 *     _compute_torques WG:1262-1295, post_physics_step WG:865-915, update_curr_ee_goal /
 *     _resample_ee_goal / collision_check WG:1303-1350, _post_physics_step_callback
 *     WG:917-935, _resample_commands WG:831-843, _push_robots WG:804-814, check_termination
 *     WG:937-963, compute_reward WG:170-205 + _reward_* WG:1352-1469, reset_idx WG:695-754,
 *     _reset_dofs WG:816-828, _reset_root_states WG:757-788, compute_observations WG:966-1001.
 *     PINNED to the reference: tests/golden/wg_reference_*.npz are trajectories of the reference's
 *     own WidowGo1.step (tools/make_golden_wg.py runs that class on the fake Isaac Gym of
 *     tools/ref_harness/, whose gym.simulate is physics_substep below); tests/test_wg_golden.py
 *     replays every recorded step through this file (fp64: 2e-5, masks bit-exact) and through the
 *     HIP kernel.
 *   - PHYSICS (what WG:1183-1187 hands to Isaac Gym / PhysX, closed source and absent) is this
 *     framework's own specification: Featherstone articulated-body algorithm for the floating
 *     base + 18 revolute joints, velocity-level contact impulses from per-body inverse
 *     articulated inertias (the URDF's collision geometry as spheres against the terrain plus
 *     sphere-vs-box / sphere-vs-capsule self-collision pairs acting on both bodies, damped
 *     block-Jacobi sweeps through the tree: DESIGN.md section 3), semi-implicit Euler. PARITY UNPINNED for physics: no reference
 *     output exists to pin it; tests/test_oracle_physics.py pins it
 *     instead against an independent composite-rigid-body/Newton-Euler formulation and
 *     conservation laws.
 *   - the six helpers the reference imports from an author-patched isaacgym.torch_utils
 *     (euler_from_quat, sphere2cart, cart2sphere, torch_wrap_to_pi_minuspi, ...) follow
 *     SURVEY.md Appendix D (their source is in neither tree); the harness uses the same
 *     reconstruction, so the fixtures pin how the reference USES them, not their definitions.
 *   - random draws: the reference draws from torch's global generator (torch_rand_float);
 *     here every draw is a counter-based hash of (seed, env, step, slot) so the kernels and
 *     this file produce bit-identical integers (the harness feeds the reference the same
 *     uniforms through its own torch_rand_float formula).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/wbc_sim.h"

#ifndef REAL
#define REAL double
#endif

#define PI_R ((REAL)3.14159265358979323846)

/* ------------------------------------------------------------------ RNG -- */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* uniform in [0,1) with 24 random bits: exactly representable in fp32 */
static inline REAL rng_u01(uint64_t seed, uint64_t env, uint64_t step, uint64_t slot) {
  uint64_t h = mix64(seed + env * 0x9E3779B97F4A7C15ULL);
  h = mix64(h + step * 0xD1B54A32D192ED03ULL + slot * 0x8CB92BA72F3D8DD7ULL);
  return (REAL)(uint32_t)(h >> 40) * (REAL)(1.0 / 16777216.0);
}
/* torch_rand_float(lo, hi, ...) = (hi-lo)*rand + lo   */
static inline REAL rng_range(REAL lo, REAL hi, uint64_t seed, uint64_t env, uint64_t step, uint64_t slot) {
  return (hi - lo) * rng_u01(seed, env, step, slot) + lo;
}
enum {
  SLOT_GOAL_ORN = 0, SLOT_GOAL_SPHERE = 3, SLOT_CMD = 33, SLOT_PUSH = 35, SLOT_RESET_DOF = 37,
  SLOT_RESET_XY = 57, SLOT_RESET_VEL = 59, SLOT_RESET_CMD = 65, SLOT_RESET_GOAL_ORN = 67,
  SLOT_RESET_GOAL_SPHERE = 70
};

/* ------------------------------------------------------------ small math -- */
static inline void cross3(const REAL* a, const REAL* b, REAL* o) {
  REAL x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline REAL dot3(const REAL* a, const REAL* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline REAL dot6(const REAL* a, const REAL* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
static inline void mat3_mul_vec(const REAL* M, const REAL* v, REAL* o) { /* row-major */
  REAL x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  REAL y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  REAL z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void mat3T_mul_vec(const REAL* M, const REAL* v, REAL* o) {
  REAL x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  REAL y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  REAL z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* xyzw quaternion -> rotation matrix (body -> world) */
static void quat_to_mat(const REAL* q, REAL* R) {
  REAL x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
static void quat_mul(const REAL* a, const REAL* b, REAL* o) { /* xyzw, o = a (x) b */
  REAL x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  REAL y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  REAL z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  REAL w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
/* isaacgym.torch_utils.quat_rotate_inverse (stock definition) */
static void quat_rotate_inverse(const REAL* q, const REAL* v, REAL* o) {
  REAL w = q[3];
  REAL a0 = v[0] * (2 * w * w - 1), a1 = v[1] * (2 * w * w - 1), a2 = v[2] * (2 * w * w - 1);
  REAL cr[3]; cross3(q, v, cr);
  REAL d = 2 * dot3(q, v);
  o[0] = a0 - cr[0] * w * 2 + q[0] * d;
  o[1] = a1 - cr[1] * w * 2 + q[1] * d;
  o[2] = a2 - cr[2] * w * 2 + q[2] * d;
}
/* euler_from_quat (App. D reconstruction): roll, pitch, yaw */
static void euler_from_quat(const REAL* q, REAL* rpy) {
  REAL x = q[0], y = q[1], z = q[2], w = q[3];
  rpy[0] = atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  REAL sp = 2 * (w * y - z * x);
  if (sp > 1) sp = 1;
  if (sp < -1) sp = -1;
  rpy[1] = asin(sp);
  rpy[2] = atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}
static void sphere2cart(const REAL* s, REAL* c) { /* WG:855-863 convention */
  REAL l = s[0], p = s[1], y = s[2];
  c[0] = l * cos(p) * cos(y); c[1] = l * cos(p) * sin(y); c[2] = l * sin(p);
}
static void cart2sphere(const REAL* c, REAL* s) {
  REAL l = sqrt(dot3(c, c));
  s[0] = l; s[1] = asin(c[2] / l); s[2] = atan2(c[1], c[0]);
}
static REAL wrap_to_pi(REAL a) { /* (a + pi) mod 2pi - pi, python-style mod */
  REAL two_pi = 2 * PI_R;
  REAL t = a + PI_R;
  t = t - two_pi * floor(t / two_pi);
  return t - PI_R;
}
static inline REAL lerp_torch(REAL a, REAL b, REAL w) { /* torch.lerp's two-branch formula */
  return (w < (REAL)0.5) ? a + w * (b - a) : b - (b - a) * (1 - w);
}
static inline REAL clampr(REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------ env state -- */
typedef struct {
  REAL root[2][13];          /* robot, box: pos3 quat4 linvel3 angvel3 (world) */
  REAL q[WBC_NDOF], qd[WBC_NDOF];
  REAL torques[WBC_NDOF];
  REAL rb_state[WBC_NRB_ENV][13];
  REAL contact_force[WBC_NRB_ENV][3];
  REAL force_sensor[WBC_NFEET][6];
  REAL obs[WBC_NOBS];
  REAL obs_hist[WBC_HIST][WBC_NPROP];
  REAL act_hist[WBC_ADELAY_LEN][WBC_NACT];
  REAL actions[WBC_NACT], last_actions[WBC_NACT];
  REAL last_dof_vel[WBC_NDOF], last_root_vel[6];
  REAL commands[3];
  REAL goal[24];
  REAL rew, arm_rew;
  int64_t reset_buf, episode_length;
  uint8_t time_out;
  REAL episode_sums[WBC_NREW], metric_sums[WBC_NMETRIC];
  REAL episode_sums_done[WBC_NREW], metric_sums_done[WBC_NMETRIC];
  REAL base_lin_vel[3], base_ang_vel[3];
  REAL mass_params[5], friction, motor_strength[WBC_NACT];
  REAL env_origin[3], box_delta_y;
  REAL body_params[20];      /* root (m, com3, I6), gripper (m, com3, I6) */
  REAL reset_travel[2];      /* ||root_xy - origin_xy||, ||commands[:2]|| at the moment of reset (LR:431-435) */
  REAL box_mass;             /* total mass of the box actor (WG:458-466) */
  REAL box_timer;            /* substeps the box has been at rest (asleep from box_sleep_time / sim_dt on) */
  REAL dropped_hits;         /* broad-phase hits that found no free dynamic slot (WBC_T_DROPPED_HITS: diagnostic, accumulated) */
  REAL feet_air_time[WBC_NFEET], last_contacts[WBC_NFEET];   /* LR:898-909 (WG:626,633) */
} ora_env;

/* goal[] slots */
enum { G_START = 0, G_GOAL = 3, G_GOAL_CART = 6, G_CURR = 9, G_CURR_CART = 12, G_DORN = 15, G_ORN = 18,
       G_TIMER = 21, G_TRAJ = 22, G_TOTAL = 23 };

typedef struct {
  wbc_model model;
  wbc_task_cfg cfg;
  wbc_curriculum cur;
  int n;
  uint64_t seed;
  int64_t step_counter;
  ora_env* env;
  /* heightfield (optional) */
  int16_t* hf; int hf_rows, hf_cols; REAL hf_hs, hf_vs, hf_t[3];
} ora_sim;

/* --------------------------------------------------------------- terrain -- */
/* Height and unit normal of the terrain under world (x,y). The triangle mesh Isaac Gym builds
 * from a height grid splits every cell along the (i,j)-(i+1,j+1) diagonal; indices truncate
 * toward zero and clip as at LR:816-829. */
static void terrain_query(const ora_sim* s, REAL x, REAL y, REAL* h, REAL* n) {
  if (!s->hf) { *h = (REAL)s->cfg.ground_z; n[0] = 0; n[1] = 0; n[2] = 1; return; }
  REAL fx = (x - s->hf_t[0]) / s->hf_hs, fy = (y - s->hf_t[1]) / s->hf_hs;
  int64_t ix = (int64_t)fx, iy = (int64_t)fy;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > s->hf_rows - 2) ix = s->hf_rows - 2;
  if (iy > s->hf_cols - 2) iy = s->hf_cols - 2;
  REAL u = clampr(fx - (REAL)ix, 0, 1), v = clampr(fy - (REAL)iy, 0, 1);
  REAL h00 = s->hf[ix * s->hf_cols + iy] * s->hf_vs, h10 = s->hf[(ix + 1) * s->hf_cols + iy] * s->hf_vs;
  REAL h01 = s->hf[ix * s->hf_cols + iy + 1] * s->hf_vs, h11 = s->hf[(ix + 1) * s->hf_cols + iy + 1] * s->hf_vs;
  REAL dhdx, dhdy, hh;
  if (u >= v) { dhdx = h10 - h00; dhdy = h11 - h10; hh = h00 + u * dhdx + v * dhdy; }
  else        { dhdy = h01 - h00; dhdx = h11 - h01; hh = h00 + u * dhdx + v * dhdy; }
  *h = hh + s->hf_t[2];
  REAL gx = dhdx / s->hf_hs, gy = dhdy / s->hf_hs;
  REAL inv = 1 / sqrt(gx * gx + gy * gy + 1);
  n[0] = -gx * inv; n[1] = -gy * inv; n[2] = inv;
}

/* --------------------------------------------------------------- physics -- */
typedef struct {
  REAL E[WBC_NB][9], pos[WBC_NB][3];
  REAL S[WBC_NB][6], v[WBC_NB][6], c[WBC_NB][6];
  REAL IA[WBC_NB][36], pA[WBC_NB][6];
  REAL U[WBC_NB][6], D[WBC_NB], u[WBC_NB];
  REAL a[WBC_NB][6], qdd[WBC_NB];
  REAL K[WBC_NB][36];
} aba_ws;

static void body_inertia_params(const ora_sim* s, const ora_env* e, int i, REAL* m, REAL* com, REAL* I6) {
  const wbc_model* md = &s->model;
  if (i == 0) { *m = e->body_params[0]; memcpy(com, e->body_params + 1, 3 * sizeof(REAL)); memcpy(I6, e->body_params + 4, 6 * sizeof(REAL)); }
  else if (i == md->gripper_body) { *m = e->body_params[10]; memcpy(com, e->body_params + 11, 3 * sizeof(REAL)); memcpy(I6, e->body_params + 14, 6 * sizeof(REAL)); }
  else { *m = md->mass[i]; for (int k = 0; k < 3; ++k) com[k] = md->com[i][k]; for (int k = 0; k < 6; ++k) I6[k] = md->inertia[i][k]; }
}

/* Forward kinematics in frame F (origin = base origin, axes = base axes). */
static void fk(const ora_sim* s, const REAL* q, aba_ws* w) {
  const wbc_model* md = &s->model;
  for (int k = 0; k < 9; ++k) w->E[0][k] = (k % 4 == 0) ? 1 : 0;
  w->pos[0][0] = w->pos[0][1] = w->pos[0][2] = 0;
  for (int i = 1; i < WBC_NB; ++i) {
    int p = md->parent[i], ax = md->axis[i];
    REAL r[3] = {md->joint_xyz[i][0], md->joint_xyz[i][1], md->joint_xyz[i][2]}, t[3];
    mat3_mul_vec(w->E[p], r, t);
    for (int k = 0; k < 3; ++k) w->pos[i][k] = w->pos[p][k] + t[k];
    REAL qi = q[md->dof[i]], cq = cos(qi), sq = sin(qi);
    int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;   /* rotation about ax mixes columns a1, a2 */
    for (int row = 0; row < 3; ++row) {
      REAL e0 = w->E[p][row * 3 + ax], e1 = w->E[p][row * 3 + a1], e2 = w->E[p][row * 3 + a2];
      w->E[i][row * 3 + ax] = e0;
      w->E[i][row * 3 + a1] = cq * e1 + sq * e2;
      w->E[i][row * 3 + a2] = -sq * e1 + cq * e2;
    }
  }
}

/* 6x6 SPD inverse by six Gauss-Jordan sweeps (no pivoting: the matrix is positive definite).
 * Written entry-wise so that the HIP kernel can run one entry per lane with the same arithmetic. */
static void spd6_inverse(const REAL* A, REAL* Ainv) {
  REAL M[36], Nw[36];
  memcpy(M, A, sizeof(M));
  for (int k = 0; k < 6; ++k) {
    REAL id = 1 / M[k * 6 + k];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
      REAL v;
      if (r == k && c == k) v = id;
      else if (r == k) v = M[k * 6 + c] * id;
      else if (c == k) v = -M[r * 6 + k] * id;
      else v = M[r * 6 + c] - M[r * 6 + k] * M[k * 6 + c] * id;
      Nw[r * 6 + c] = v;
    }
    memcpy(M, Nw, sizeof(M));
  }
  memcpy(Ainv, M, sizeof(M));
}

static void solve3(const REAL* W, const REAL* b, REAL* x) { /* symmetric 3x3, cofactors */
  REAL a = W[0], bb = W[1], c = W[2], d = W[4], e = W[5], f = W[8];
  REAL c00 = d * f - e * e, c01 = c * e - bb * f, c02 = bb * e - c * d;
  REAL det = a * c00 + bb * c01 + c * c02;
  REAL c11 = a * f - c * c, c12 = bb * c - a * e, c22 = a * d - bb * bb;
  REAL id = 1 / det;
  x[0] = (c00 * b[0] + c01 * b[1] + c02 * b[2]) * id;
  x[1] = (c01 * b[0] + c11 * b[1] + c12 * b[2]) * id;
  x[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) * id;
}

typedef struct {
  int active, nshare;      /* nshare: the larger of the numbers of active contacts acting on the contact's two bodies (>= 1 when active) */
  REAL xc[3], n[3], W[9], vfree[3], vn_tgt, mu, lam[3], gap;
} contact_t;

/* The free box actor (WG:321-325,384) in frame F: centre, axes, centre velocity, angular velocity; mass and the (isotropic: a
 * cube) rotational inertia about the centre. It is not part of the tree: its inverse inertia is closed-form about its own centre. */
typedef struct { REAL c[3], E[9], v[3], w[3], m, Ic; } box_ws;

/* Friction-cone solve of one contact given the velocity the point would have without its own
 * impulse: returns the total impulse. */
static void contact_solve(const contact_t* c, const REAL* vref, REAL* lam) {
  REAL vn = dot3(c->n, vref);
  lam[0] = lam[1] = lam[2] = 0;
  if (vn >= c->vn_tgt) return;
  REAL Wn[3]; mat3_mul_vec(c->W, c->n, Wn);
  REAL nWn = dot3(c->n, Wn);
  REAL lam_fl = (c->vn_tgt - vn) / nWn;            /* frictionless answer */
  REAL rhs[3], st[3];
  for (int k = 0; k < 3; ++k) rhs[k] = c->n[k] * c->vn_tgt - vref[k];
  solve3(c->W, rhs, st);                            /* stick */
  REAL ln = dot3(c->n, st);
  REAL lt[3] = {st[0] - ln * c->n[0], st[1] - ln * c->n[1], st[2] - ln * c->n[2]};
  REAL ltn = sqrt(dot3(lt, lt));
  if (ln > 0 && ltn <= c->mu * ln) { lam[0] = st[0]; lam[1] = st[1]; lam[2] = st[2]; return; }
  /* sliding: Coulomb friction of magnitude mu * (normal impulse) opposing the tangential velocity of the contact point; the
   * normal impulse is what the normal-velocity target needs in the presence of that friction. (The stick impulse's own tangential
   * direction is bent by the anisotropy of W, and its normal part may even be negative for a corner whose friction pitches the
   * body onto it: four corners of a sliding box then felt half the friction, tests/test_oracle_contact_physics.py.) A point
   * without tangential velocity slides where the stick impulse points. */
  REAL vt[3] = {vref[0] - vn * c->n[0], vref[1] - vn * c->n[1], vref[2] - vn * c->n[2]};
  REAL vtn = sqrt(dot3(vt, vt));
  REAL dir[3];
  if (vtn > (REAL)1e-6) { for (int k = 0; k < 3; ++k) dir[k] = c->n[k] - c->mu * vt[k] / vtn; }
  else if (ltn > (REAL)1e-12) { for (int k = 0; k < 3; ++k) dir[k] = c->n[k] + c->mu * lt[k] / ltn; }
  else { for (int k = 0; k < 3; ++k) lam[k] = c->n[k] * lam_fl; return; }
  REAL Wd[3]; mat3_mul_vec(c->W, dir, Wd);
  REAL den = dot3(c->n, Wd);
  if (den <= (REAL)0.05 * nWn) { for (int k = 0; k < 3; ++k) lam[k] = c->n[k] * lam_fl; return; }
  REAL l = (c->vn_tgt - vn) / den;                  /* > 0 */
  for (int k = 0; k < 3; ++k) lam[k] = dir[k] * l;
}

/* What contact slot k stands for in this substep: the model's static tables, or -- a dynamic slot -- the pair the broad phase promoted
 * into it. rad / rad2: radii of the touching features on the two sides (the force sensors' lever arms). */
typedef struct { int kind, body, body2, rb, rb2; REAL rad, rad2; } slot_t;

/* Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9), parameters s on a0..a1 and t on b0..b1; a segment
 * shorter than 1e-5 m counts as a point. */
static void seg_seg_closest(const REAL* a0, const REAL* a1, const REAL* b0, const REAL* b1, REAL* pa, REAL* pb) {
  REAL d1[3], d2[3], r[3];
  for (int j = 0; j < 3; ++j) { d1[j] = a1[j] - a0[j]; d2[j] = b1[j] - b0[j]; r[j] = a0[j] - b0[j]; }
  const REAL EPS = (REAL)1e-10;
  REAL a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), sp = 0, tp = 0;
  if (a <= EPS && e <= EPS) { sp = tp = 0; }
  else if (a <= EPS) { sp = 0; tp = clampr(f / e, 0, 1); }
  else {
    REAL c = dot3(d1, r);
    if (e <= EPS) { tp = 0; sp = clampr(-c / a, 0, 1); }
    else {
      REAL b = dot3(d1, d2), den = a * e - b * b;
      sp = den > (REAL)1e-7 * a * e ? clampr((b * f - c * e) / den, 0, 1) : 0;       /* (near-)parallel: any point of a will do */
      tp = (b * sp + f) / e;
      if (tp < 0) { tp = 0; sp = clampr(-c / a, 0, 1); }
      else if (tp > 1) { tp = 1; sp = clampr((b - c) / a, 0, 1); }
    }
  }
  for (int j = 0; j < 3; ++j) { pa[j] = a0[j] + d1[j] * sp; pb[j] = b0[j] + d2[j] * tp; }
}

/* Limb A against limb B, each the union of a capsule (segment x0..x1, radius r) and up to two end spheres (radii c0, c1 > 0 at x0, x1;
 * 0: none): the feature pair with the smallest gap wins -- shaft/shaft, A's end spheres against B's shaft, B's end spheres against
 * A's shaft, end sphere against end sphere, in this order (a tie keeps the earlier). Out: the gap (minus rest), the unit normal from B
 * to A, the contact point on B's surface, which feature touched on each side (0 shaft, 1 end 0, 2 end 1) and its radius. */
typedef struct { REAL gap, n[3], q[3], ra, rb; int fa, fb; } limb_hit;
static void limb_pair(const REAL* a0, const REAL* a1, REAL ra, REAL ca0, REAL ca1, const REAL* b0, const REAL* b1, REAL rb, REAL cb0, REAL cb1,
                      REAL rest, limb_hit* out) {
  const REAL* aend[2] = {a0, a1}; const REAL* bend[2] = {b0, b1};
  const REAL acap[2] = {ca0, ca1}, bcap[2] = {cb0, cb1};
  REAL best = 0, bpa[3] = {0, 0, 0}, bpb[3] = {0, 0, 0};
  int first = 1;
  out->fa = out->fb = 0; out->ra = ra; out->rb = rb;
  for (int f = 0; f < 9; ++f) {
    /* f = 0: shafts; 1, 2: A's end ea = f - 1 vs B's shaft; 3, 4: A's shaft vs B's end eb = f - 3; 5..8: ends (ea, eb) = ((f - 5) / 2, (f - 5) % 2) */
    int ea = f == 0 ? -1 : (f <= 2 ? f - 1 : (f <= 4 ? -1 : (f - 5) / 2));
    int eb = f <= 2 ? -1 : (f <= 4 ? f - 3 : (f - 5) % 2);
    if ((ea >= 0 && !(acap[ea] > 0)) || (eb >= 0 && !(bcap[eb] > 0))) continue;
    REAL pa[3], pb[3];
    seg_seg_closest(ea >= 0 ? aend[ea] : a0, ea >= 0 ? aend[ea] : a1, eb >= 0 ? bend[eb] : b0, eb >= 0 ? bend[eb] : b1, pa, pb);
    REAL fra = ea >= 0 ? acap[ea] : ra, frb = eb >= 0 ? bcap[eb] : rb;
    REAL d[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    REAL g = sqrt(dot3(d, d)) - fra - frb - rest;
    if (first || g < best) {
      first = 0; best = g;
      for (int j = 0; j < 3; ++j) { bpa[j] = pa[j]; bpb[j] = pb[j]; }
      out->fa = ea + 1; out->fb = eb + 1; out->ra = fra; out->rb = frb;
    }
  }
  REAL d[3] = {bpa[0] - bpb[0], bpa[1] - bpb[1], bpa[2] - bpb[2]};
  REAL dist = sqrt(dot3(d, d));
  if (dist > (REAL)1e-9) { for (int j = 0; j < 3; ++j) out->n[j] = d[j] / dist; }
  else { out->n[0] = 1; out->n[1] = 0; out->n[2] = 0; }
  for (int j = 0; j < 3; ++j) out->q[j] = bpb[j] + out->rb * out->n[j];
  out->gap = best;
}

/* diagnostics (tests only): when set, physics_substep copies its contact list here */
typedef struct { int active[WBC_NCP]; double lam[WBC_NCP][3], n[WBC_NCP][3], xc[WBC_NCP][3], resid[WBC_NCP]; int nshare[WBC_NCP]; } contact_dump;
static _Thread_local contact_dump* g_dump = NULL;

/* One physics substep (what gym.simulate does at WG:1184), torques already in e->torques. */
static void physics_substep(const ora_sim* s, ora_env* e) {
  const wbc_model* md = &s->model;
  const wbc_task_cfg* cf = &s->cfg;
  const REAL dt = (REAL)cf->sim_dt;
  aba_ws W_; aba_ws* w = &W_;
  REAL R[9];
  quat_to_mat(e->root[0] + 3, R);
  REAL wb[3], vb[3], gF[3], gw[3] = {cf->gravity[0], cf->gravity[1], cf->gravity[2]};
  mat3T_mul_vec(R, e->root[0] + 10, wb);
  mat3T_mul_vec(R, e->root[0] + 7, vb);
  mat3T_mul_vec(R, gw, gF);

  fk(s, e->q, w);
  /* pass 1: velocities, bias accelerations, spatial inertias, bias forces */
  for (int k = 0; k < 3; ++k) { w->v[0][k] = wb[k]; w->v[0][3 + k] = vb[k]; }
  for (int k = 0; k < 6; ++k) { w->S[0][k] = 0; w->c[0][k] = 0; }
  for (int i = 1; i < WBC_NB; ++i) {
    int p = md->parent[i], ax = md->axis[i];
    REAL sv[3] = {w->E[i][ax], w->E[i][3 + ax], w->E[i][6 + ax]}, lin[3];
    cross3(w->pos[i], sv, lin);
    for (int k = 0; k < 3; ++k) { w->S[i][k] = sv[k]; w->S[i][3 + k] = lin[k]; }
    REAL qd = e->qd[md->dof[i]], vJ[6];
    for (int k = 0; k < 6; ++k) { vJ[k] = w->S[i][k] * qd; w->v[i][k] = w->v[p][k] + vJ[k]; }
    REAL t1[3], t2[3], t3[3];
    cross3(w->v[i], vJ, t1);           /* w x vJa */
    cross3(w->v[i], vJ + 3, t2);       /* w x vJl */
    cross3(w->v[i] + 3, vJ, t3);       /* vl x vJa */
    for (int k = 0; k < 3; ++k) { w->c[i][k] = t1[k]; w->c[i][3 + k] = t2[k] + t3[k]; }
  }
  for (int i = 0; i < WBC_NB; ++i) {
    REAL m, com[3], I6[6];
    body_inertia_params(s, e, i, &m, com, I6);
    REAL C[3], t[3];
    mat3_mul_vec(w->E[i], com, t);
    for (int k = 0; k < 3; ++k) C[k] = w->pos[i][k] + t[k];
    REAL Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    REAL EI[9], Ibar[9];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      REAL a = 0; for (int k = 0; k < 3; ++k) a += w->E[i][r * 3 + k] * Ib[k * 3 + cc]; EI[r * 3 + cc] = a; }
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      REAL a = 0; for (int k = 0; k < 3; ++k) a += EI[r * 3 + k] * w->E[i][cc * 3 + k]; Ibar[r * 3 + cc] = a; }
    REAL CC = dot3(C, C);
    REAL* I = w->IA[i];
    REAL Cx[9] = {0, -C[2], C[1], C[2], 0, -C[0], -C[1], C[0], 0};
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      I[r * 6 + cc] = Ibar[r * 3 + cc] + m * ((r == cc ? CC : 0) - C[r] * C[cc]);
      I[r * 6 + 3 + cc] = m * Cx[r * 3 + cc];
      I[(3 + r) * 6 + cc] = m * Cx[cc * 3 + r];
      I[(3 + r) * 6 + 3 + cc] = (r == cc) ? m : 0;
    }
    REAL Iv[6];
    for (int r = 0; r < 6; ++r) Iv[r] = dot6(I + r * 6, w->v[i]);
    REAL t1[3], t2[3], t3[3];
    cross3(w->v[i], Iv, t1);           /* w x n */
    cross3(w->v[i] + 3, Iv + 3, t2);   /* vl x f */
    cross3(w->v[i], Iv + 3, t3);       /* w x f */
    for (int k = 0; k < 3; ++k) { w->pA[i][k] = t1[k] + t2[k]; w->pA[i][3 + k] = t3[k]; }
  }
  /* pass 2: articulated inertias, inward */
  for (int i = WBC_NB - 1; i >= 1; --i) {
    int p = md->parent[i], d = md->dof[i];
    REAL* I = w->IA[i];
    for (int r = 0; r < 6; ++r) w->U[i][r] = dot6(I + r * 6, w->S[i]);
    REAL D = dot6(w->S[i], w->U[i]) + (d < WBC_NACT ? (REAL)cf->joint_armature[d] : 0);
    w->D[i] = D;
    REAL tau = e->torques[d];
    /* joint-limit stop, scaled by the joint's articulated inertia */
    REAL lo = md->q_lower[d], hi = md->q_upper[d];
    if (lo < hi) {
      REAL qq = e->q[d], qdv = e->qd[d], viol = 0;
      if (qq > hi) viol = qq - hi; else if (qq < lo) viol = qq - lo;
      if (viol != 0) {
        REAL tl = -(REAL)cf->limit_kappa * D / (dt * dt) * viol;
        if (qdv * viol > 0) tl -= (REAL)cf->limit_delta * D / dt * qdv;
        tau += tl;
      }
    }
    REAL u = tau - dot6(w->S[i], w->pA[i]);
    w->u[i] = u;
    REAL invD = 1 / D;
    REAL Ia[36];
    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) Ia[r * 6 + cc] = I[r * 6 + cc] - w->U[i][r] * w->U[i][cc] * invD;
    for (int r = 0; r < 6; ++r) {
      REAL pa = w->pA[i][r] + dot6(Ia + r * 6, w->c[i]) + w->U[i][r] * u * invD;
      w->pA[p][r] += pa;
    }
    for (int k = 0; k < 36; ++k) w->IA[p][k] += Ia[k];
  }
  /* root */
  spd6_inverse(w->IA[0], w->K[0]);
  for (int r = 0; r < 6; ++r) w->a[0][r] = -dot6(w->K[0] + r * 6, w->pA[0]);
  /* pass 3 + inverse articulated inertias, outward */
  for (int i = 1; i < WBC_NB; ++i) {
    int p = md->parent[i];
    REAL ap[6];
    for (int k = 0; k < 6; ++k) ap[k] = w->a[p][k] + w->c[i][k];
    REAL invD = 1 / w->D[i];
    REAL qdd = (w->u[i] - dot6(w->U[i], ap)) * invD;
    w->qdd[i] = qdd;
    for (int k = 0; k < 6; ++k) w->a[i][k] = ap[k] + w->S[i][k] * qdd;
    REAL g[6];
    for (int r = 0; r < 6; ++r) g[r] = dot6(w->K[p] + r * 6, w->U[i]) * invD;
    REAL gam = dot6(w->U[i], g) * invD + invD;
    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc)
      w->K[i][r * 6 + cc] = w->K[p][r * 6 + cc] - g[r] * w->S[i][cc] - w->S[i][r] * g[cc] + gam * w->S[i][r] * w->S[i][cc];
  }
  /* the free box in frame F */
  box_ws bx;
  {
    REAL Rb[9], d[3];
    quat_to_mat(e->root[1] + 3, Rb);
    for (int j = 0; j < 3; ++j) d[j] = e->root[1][j] - e->root[0][j];
    mat3T_mul_vec(R, d, bx.c);
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      REAL acc = 0; for (int j = 0; j < 3; ++j) acc += R[j * 3 + r] * Rb[j * 3 + cc]; bx.E[r * 3 + cc] = acc; }
    mat3T_mul_vec(R, e->root[1] + 7, bx.v);
    mat3T_mul_vec(R, e->root[1] + 10, bx.w);
    bx.m = e->box_mass;
    bx.Ic = bx.m * (REAL)(2.0 / 3.0) * (REAL)md->box_half * (REAL)md->box_half;    /* m (2h)^2 / 6 */
  }
  /* contacts. (1) every sphere against the terrain (the robot's, the box's corners); the robot spheres' centres in F are cached.
   * (2) broad phase: every lane's pair descriptor (its own static pair, or a candidate of the self-collision set) against bounding
   * spheres. (3) candidates that pass are promoted into free dynamic slots. (4) exact tests: sphere vs box (the trunk's, the free
   * box), limb vs limb. Slot k of this substep is described by sl[k] (static slots: the model's tables; dynamic: what was promoted). */
  contact_t ct[WBC_NCP];
  slot_t sl[WBC_NCP];
  REAL sph[WBC_NSPH][3];
  int sph_slot[WBC_NSPH];
  REAL mu = (REAL)0.5 * (e->friction + (REAL)cf->terrain_friction);   /* PhysX default combine: average */
  if (mu < 0) mu = 0;
  REAL mu_self = e->friction < 0 ? 0 : e->friction;                   /* both shapes carry the robot's material */
  REAL mu_box_terrain = (REAL)0.5 * ((REAL)md->box_friction + (REAL)cf->terrain_friction);
  REAL mu_box_robot = (REAL)0.5 * ((REAL)md->box_friction + e->friction);
  if (mu_box_terrain < 0) mu_box_terrain = 0;
  if (mu_box_robot < 0) mu_box_robot = 0;
  int any = 0;
  for (int k = 0; k < WBC_NSPH; ++k) { sph_slot[k] = -1; sph[k][0] = sph[k][1] = sph[k][2] = 0; }
  for (int k = 0; k < WBC_NCP; ++k) {
    contact_t* c = &ct[k];
    c->lam[0] = c->lam[1] = c->lam[2] = 0;
    c->active = 0; c->gap = 0;
    slot_t* q = &sl[k];
    q->kind = k < md->ncp ? md->cp_kind[k] : WBC_CP_NONE;
    q->body = md->cp_body[k]; q->body2 = md->cp_body2[k]; q->rb = md->cp_rb[k]; q->rb2 = md->cp_rb2[k];
    q->rad = md->cp_radius[k]; q->rad2 = 0;
    if (q->kind != WBC_CP_TERRAIN) continue;
    int b = q->body;
    const REAL* Eb = (b == WBC_BOX_BODY) ? bx.E : w->E[b];
    const REAL* pb = (b == WBC_BOX_BODY) ? bx.c : w->pos[b];
    REAL o[3] = {md->cp_pos[k][0], md->cp_pos[k][1], md->cp_pos[k][2]}, xk[3], t[3];
    mat3_mul_vec(Eb, o, t);
    for (int j = 0; j < 3; ++j) xk[j] = pb[j] + t[j];
    if (md->cp_sph[k] >= 0) { sph_slot[md->cp_sph[k]] = k; for (int j = 0; j < 3; ++j) sph[md->cp_sph[k]][j] = xk[j]; }
    REAL rad = q->rad, Xw[3], h, nw[3];
    mat3_mul_vec(R, xk, t);
    for (int j = 0; j < 3; ++j) Xw[j] = e->root[0][j] + t[j];
    terrain_query(s, Xw[0], Xw[1], &h, nw);
    REAL gap = (Xw[2] - h) * nw[2] - rad;
    c->active = gap < (REAL)cf->contact_margin;
    c->gap = gap;
    if (!c->active) continue;
    mat3T_mul_vec(R, nw, c->n);
    for (int j = 0; j < 3; ++j) c->xc[j] = xk[j] - rad * c->n[j];
    c->mu = (b == WBC_BOX_BODY) ? mu_box_terrain : mu;
  }
  /* (2) broad phase */
  int near[WBC_NCP];
  for (int k = 0; k < WBC_NCP; ++k) {
    near[k] = 0;
    int pk = md->pr_kind[k];
    if (pk == WBC_PR_NONE) continue;
    REAL ca[3], cb[3];
    if (pk == WBC_PR_LIMBS) {
      int la = md->pr_a[k], lb = md->pr_b[k];
      for (int j = 0; j < 3; ++j) {
        ca[j] = (REAL)0.5 * (sph[md->limb_s0[la]][j] + sph[md->limb_s1[la]][j]);
        cb[j] = (REAL)0.5 * (sph[md->limb_s0[lb]][j] + sph[md->limb_s1[lb]][j]);
      }
    } else {
      for (int j = 0; j < 3; ++j) ca[j] = sph[md->pr_a[k]][j];
      int b2 = pk == WBC_PR_STATIC ? md->cp_body2[k] : WBC_BOX_BODY;
      if (b2 == WBC_BOX_BODY) { for (int j = 0; j < 3; ++j) cb[j] = bx.c[j]; }
      else {                                   /* a box fixed to a tree body: its centre in F */
        REAL A[3] = {md->cp_a[k][0], md->cp_a[k][1], md->cp_a[k][2]}, t[3];
        mat3_mul_vec(w->E[b2], A, t);
        for (int j = 0; j < 3; ++j) cb[j] = w->pos[b2][j] + t[j];
      }
    }
    REAL reach = (REAL)md->pr_reach[k] + (REAL)cf->contact_margin + (REAL)1e-3, d2 = 0;
    for (int j = 0; j < 3; ++j) d2 += (ca[j] - cb[j]) * (ca[j] - cb[j]);
    near[k] = d2 < reach * reach;
    if (near[k] && pk == WBC_PR_LIMBS) {       /* second stage: the two shafts' segments against a generous common radius */
      int la = md->pr_a[k], lb = md->pr_b[k];
      REAL pa[3], pb[3], dd = 0;
      seg_seg_closest(sph[md->limb_s0[la]], sph[md->limb_s1[la]], sph[md->limb_s0[lb]], sph[md->limb_s1[lb]], pa, pb);
      for (int j = 0; j < 3; ++j) dd += (pa[j] - pb[j]) * (pa[j] - pb[j]);
      REAL r2 = (REAL)WBC_LIMB_RSUM_MAX + (REAL)md->pair_rest_offset + (REAL)cf->contact_margin + (REAL)1e-3;
      near[k] = dd < r2 * r2;
    }
  }
  /* (3) promotion: robot-vs-robot hits into the dynamic slots outside the box row, robot-vs-box hits into those inside it, both in
   * ascending order of lane and slot; hits beyond the free slots are dropped */
  int src[WBC_NCP];
  for (int k = 0; k < WBC_NCP; ++k) src[k] = -1;
  {
    int ds = 0, db = 32;
    for (int k = 0; k < WBC_NCP; ++k) {
      if (!near[k] || md->pr_kind[k] == WBC_PR_STATIC) continue;
      if (md->pr_kind[k] == WBC_PR_LIMBS) {
        while (ds < md->ncp && !(md->cp_kind[ds] == WBC_CP_DYNAMIC && (ds < 32 || ds > 47))) ++ds;
        if (ds < md->ncp) src[ds++] = k; else e->dropped_hits += 1;
      } else {
        while (db < 48 && db < md->ncp && md->cp_kind[db] != WBC_CP_DYNAMIC) ++db;
        if (db < 48 && db < md->ncp) src[db++] = k; else e->dropped_hits += 1;
      }
    }
  }
  /* (4) exact tests */
  for (int k = 0; k < md->ncp; ++k) {
    contact_t* c = &ct[k];
    slot_t* q = &sl[k];
    int cand = -1;
    if (q->kind == WBC_CP_BOX) { if (!near[k]) continue; }
    else if (q->kind == WBC_CP_DYNAMIC) { if (src[k] < 0) continue; cand = src[k]; }
    else continue;
    REAL gap, nF[3], xcF[3];
    if (cand >= 0 && md->pr_kind[cand] == WBC_PR_LIMBS) {
      int la = md->pr_a[cand], lb = md->pr_b[cand];
      limb_hit hit;
      limb_pair(sph[md->limb_s0[la]], sph[md->limb_s1[la]], md->limb_radius[la], md->limb_cap0[la], md->limb_cap1[la],
                sph[md->limb_s0[lb]], sph[md->limb_s1[lb]], md->limb_radius[lb], md->limb_cap0[lb], md->limb_cap1[lb],
                (REAL)md->pair_rest_offset, &hit);
      gap = hit.gap;
      for (int j = 0; j < 3; ++j) { nF[j] = hit.n[j]; xcF[j] = hit.q[j]; }
      q->kind = WBC_CP_LIMBS;
      q->body = md->limb_body[la]; q->body2 = md->limb_body[lb];
      q->rb = hit.fa == 0 ? md->limb_rb[la] : (hit.fa == 1 ? md->limb_rb0[la] : md->limb_rb1[la]);
      q->rb2 = hit.fb == 0 ? md->limb_rb[lb] : (hit.fb == 1 ? md->limb_rb0[lb] : md->limb_rb1[lb]);
      q->rad = hit.ra; q->rad2 = hit.rb;
    } else {
      /* sphere against a box: the lane's own static pair, or a promoted robot sphere against the free box */
      int si = cand >= 0 ? md->pr_a[cand] : md->pr_a[k];
      int b2 = cand >= 0 ? WBC_BOX_BODY : q->body2;
      REAL A[3], B[3], rad;
      if (cand >= 0) {
        int ss = sph_slot[si];
        q->kind = WBC_CP_BOX; q->body = md->cp_body[ss]; q->rb = md->cp_rb[ss]; q->body2 = WBC_BOX_BODY; q->rb2 = WBC_BOX_RB;
        q->rad = md->cp_radius[ss];
        for (int j = 0; j < 3; ++j) { A[j] = 0; B[j] = (REAL)md->box_half; }
      } else for (int j = 0; j < 3; ++j) { A[j] = md->cp_a[k][j]; B[j] = md->cp_b[k][j]; }
      rad = q->rad;
      const REAL* E2 = (b2 == WBC_BOX_BODY) ? bx.E : w->E[b2];
      const REAL* p2 = (b2 == WBC_BOX_BODY) ? bx.c : w->pos[b2];
      REAL d[3], pl[3], ql[3], nl[3], dist, t[3];
      for (int j = 0; j < 3; ++j) d[j] = sph[si][j] - p2[j];
      mat3T_mul_vec(E2, d, pl);                /* sphere centre in the box's frame */
      int inside = 1;
      for (int j = 0; j < 3; ++j) {
        REAL r = pl[j] - A[j], hb = B[j];
        REAL cl = r < -hb ? -hb : (r > hb ? hb : r);
        if (cl != r) inside = 0;
        ql[j] = A[j] + cl;
      }
      if (!inside) {
        for (int j = 0; j < 3; ++j) nl[j] = pl[j] - ql[j];
        dist = sqrt(dot3(nl, nl));
        for (int j = 0; j < 3; ++j) nl[j] /= dist;
      } else {                            /* centre inside the box: leave through the nearest face */
        int ax = 0; REAL best = 0;
        for (int j = 0; j < 3; ++j) {
          REAL depth = B[j] - fabs(pl[j] - A[j]);
          if (j == 0 || depth < best) { best = depth; ax = j; }
        }
        for (int j = 0; j < 3; ++j) nl[j] = 0;
        nl[ax] = (pl[ax] - A[ax]) >= 0 ? 1 : -1;
        ql[ax] = A[ax] + nl[ax] * B[ax];
        dist = -best;
      }
      gap = dist - rad;
      mat3_mul_vec(E2, nl, nF);
      mat3_mul_vec(E2, ql, t);
      for (int j = 0; j < 3; ++j) xcF[j] = p2[j] + t[j];             /* on the box's surface */
    }
    c->gap = gap;
    c->active = gap < (REAL)cf->contact_margin;
    if (!c->active) { if (cand >= 0) q->kind = WBC_CP_DYNAMIC; continue; }
    for (int j = 0; j < 3; ++j) { c->n[j] = nF[j]; c->xc[j] = xcF[j]; }
    c->mu = (q->body2 == WBC_BOX_BODY) ? mu_box_robot : mu_self;
  }
  /* sleeping box (as PhysX puts resting actors to sleep): slow, supported by at least three corners, untouched by the robot ->
   * frozen for this substep: its corner contacts are dropped, its velocity is zero, gravity does not act on it */
  int box_asleep = 0;
  if (md->box_sleep_speed > 0) {
    const REAL vs2 = (REAL)md->box_sleep_speed * (REAL)md->box_sleep_speed, h2 = (REAL)md->box_half * (REAL)md->box_half;
    int ncorner = 0, touched = 0;
    for (int k = 0; k < md->ncp; ++k) if (ct[k].active) {
      if (sl[k].body == WBC_BOX_BODY) ncorner += 1;
      if (sl[k].kind > WBC_CP_TERRAIN && sl[k].body2 == WBC_BOX_BODY) touched = 1;
    }
    const int resting = dot3(e->root[1] + 7, e->root[1] + 7) < vs2 && dot3(e->root[1] + 10, e->root[1] + 10) * h2 < vs2 && ncorner >= 3 && !touched;
    const REAL nsleep = (REAL)(int)(md->box_sleep_time * (1.0f / cf->sim_dt) + 0.5f);   /* the timer counts substeps (exact in fp32) */
    box_asleep = resting && e->box_timer >= nsleep;
    e->box_timer = resting ? fmin(e->box_timer + 1, nsleep) : 0;
    if (box_asleep) for (int k = 0; k < md->ncp; ++k) if (sl[k].body == WBC_BOX_BODY) ct[k].active = 0;
  }
  for (int k = 0; k < md->ncp; ++k) {
    contact_t* c = &ct[k];
    if (!c->active) continue;
    int b = sl[k].body, kind = sl[k].kind, b2 = sl[k].body2;
    REAL gap = c->gap, t[3];
    any = 1;
    c->vn_tgt = (gap >= 0) ? -gap / dt : fmin((REAL)cf->contact_erp * (-gap) / dt, (REAL)cf->max_depenetration_vel);
    /* W = J K J^T, J = [-xc x, 1], summed over the two bodies of a pair (their cross coupling through the tree is left to the
     * block-Jacobi sweeps, as the coupling between different contacts is). The free box contributes the closed form of a rigid
     * body with isotropic inertia about its centre: 1/m + (|r|^2 1 - r r^T) / Ic, r = xc - centre. */
    REAL X[9] = {0, -c->xc[2], c->xc[1], c->xc[2], 0, -c->xc[0], -c->xc[1], c->xc[0], 0};
    REAL J[18];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) { J[r * 6 + cc] = -X[r * 3 + cc]; J[r * 6 + 3 + cc] = (r == cc) ? 1 : 0; }
    for (int r = 0; r < 9; ++r) c->W[r] = 0;
    for (int j = 0; j < 3; ++j) c->vfree[j] = 0;
    for (int side = 0; side < (kind == WBC_CP_TERRAIN ? 1 : 2); ++side) {
      int bb = side == 0 ? b : b2;
      REAL sgn = side == 0 ? 1 : -1;
      if (bb == WBC_BOX_BODY) {
        REAL r[3] = {c->xc[0] - bx.c[0], c->xc[1] - bx.c[1], c->xc[2] - bx.c[2]};
        REAL rr = dot3(r, r), im = 1 / bx.m, iI = 1 / bx.Ic;
        for (int a1 = 0; a1 < 3; ++a1) for (int a2 = 0; a2 < 3; ++a2)
          c->W[a1 * 3 + a2] += (a1 == a2 ? im + rr * iI : 0) - r[a1] * r[a2] * iI;
        /* free motion: the centre falls with gravity, the spin is constant (isotropic inertia) */
        REAL wr[3], vp[3], wwr[3];
        cross3(bx.w, r, wr);
        for (int j = 0; j < 3; ++j) vp[j] = bx.v[j] + wr[j];
        cross3(bx.w, wr, wwr);
        for (int j = 0; j < 3; ++j) c->vfree[j] += sgn * (vp[j] + dt * (gF[j] + wwr[j]));
        continue;
      }
      const REAL* K = w->K[bb];
      REAL KJt[18];
      for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc) KJt[r * 3 + cc] = dot6(K + r * 6, J + cc * 6);
      for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
        REAL a = 0; for (int j = 0; j < 6; ++j) a += J[r * 6 + j] * KJt[j * 3 + cc];
        c->W[r * 3 + cc] += a;
      }
      /* velocity of the contact point after the unconstrained step (relative to the partner body's point for a pair) */
      REAL vp[3], ab[6], apnt[3], t2[3];
      cross3(w->v[bb], c->xc, t);
      for (int j = 0; j < 3; ++j) vp[j] = w->v[bb][3 + j] + t[j];
      for (int j = 0; j < 3; ++j) { ab[j] = w->a[bb][j]; ab[3 + j] = w->a[bb][3 + j] + gF[j]; }
      cross3(ab, c->xc, t);
      cross3(w->v[bb], vp, t2);
      for (int j = 0; j < 3; ++j) { apnt[j] = ab[3 + j] + t[j] + t2[j]; c->vfree[j] += sgn * (vp[j] + dt * apnt[j]); }
    }
    /* the solver reads the upper triangle (exactly symmetric by construction of the product; the kernel stores six entries) */
    c->W[3] = c->W[1]; c->W[6] = c->W[2]; c->W[7] = c->W[5];
    for (int r = 0; r < 3; ++r) c->W[r * 3 + r] += (REAL)1e-6;
  }
  REAL qddD[WBC_NB] = {0}, aD0[6] = {0};
  REAL boxF[3] = {0, 0, 0}, boxN[3] = {0, 0, 0};       /* net contact force on the box, its moment about the box centre (frame F) */
  {
    /* damped block-Jacobi: how many active contacts act on each body (as the sphere's body or as the partner of a pair) */
    int cnt[WBC_NB + 1] = {0};
    for (int k = 0; k < md->ncp; ++k) if (ct[k].active) {
      cnt[sl[k].body] += 1;
      if (sl[k].kind != WBC_CP_TERRAIN) cnt[sl[k].body2] += 1;
    }
    for (int k = 0; k < md->ncp; ++k) {
      ct[k].nshare = 0;
      if (!ct[k].active) continue;
      ct[k].nshare = cnt[sl[k].body];
      if (sl[k].kind != WBC_CP_TERRAIN && cnt[sl[k].body2] > ct[k].nshare) ct[k].nshare = cnt[sl[k].body2];
    }
  }
  if (any) {
    REAL dv[WBC_NCP][3];
    memset(dv, 0, sizeof(dv));
    REAL aD[WBC_NB][6];
    for (int it = 0; it < cf->contact_iters; ++it) {
      for (int k = 0; k < md->ncp; ++k) {
        contact_t* c = &ct[k];
        if (!c->active) continue;
        REAL vref[3], own[3], ln[3];
        mat3_mul_vec(c->W, c->lam, own);
        for (int j = 0; j < 3; ++j) vref[j] = c->vfree[j] + dv[k][j] - own[j];
        contact_solve(c, vref, ln);
        /* damped block-Jacobi: the contacts acting on ONE body see (almost) the same inverse inertia, so each of the m
         * active ones would remove the whole approach velocity on its own; they share it instead (relaxation 1/m, m counted on
         * the busier of the contact's two bodies; a convex combination of two impulses inside the friction cone stays inside
         * it). m = 1 for a lone foot: plain block-Jacobi. */
        REAL om = 1 / (REAL)c->nshare;
        for (int j = 0; j < 3; ++j) c->lam[j] += om * (ln[j] - c->lam[j]);
      }
      /* response of the whole tree, and of the free box, to all contact impulses */
      REAL pD[WBC_NB][6], uD[WBC_NB];
      REAL bF[3] = {0, 0, 0}, bN[3] = {0, 0, 0};      /* on the box: force, moment about the box centre */
      memset(pD, 0, sizeof(pD));
      for (int k = 0; k < md->ncp; ++k) {
        contact_t* c = &ct[k];
        if (!c->active) continue;
        REAL f[3] = {c->lam[0] / dt, c->lam[1] / dt, c->lam[2] / dt}, mom[3], momb[3];
        REAL rb3[3] = {c->xc[0] - bx.c[0], c->xc[1] - bx.c[1], c->xc[2] - bx.c[2]};
        cross3(c->xc, f, mom);                            /* about F's origin (tree bodies) */
        cross3(rb3, f, momb);                             /* about the box centre */
        int b = sl[k].body, b2 = sl[k].body2;
        if (b == WBC_BOX_BODY) { for (int j = 0; j < 3; ++j) { bN[j] += momb[j]; bF[j] += f[j]; } }
        else for (int j = 0; j < 3; ++j) { pD[b][j] -= mom[j]; pD[b][3 + j] -= f[j]; }
        if (sl[k].kind != WBC_CP_TERRAIN) {          /* the partner body receives the opposite wrench */
          if (b2 == WBC_BOX_BODY) { for (int j = 0; j < 3; ++j) { bN[j] -= momb[j]; bF[j] -= f[j]; } }
          else for (int j = 0; j < 3; ++j) { pD[b2][j] += mom[j]; pD[b2][3 + j] += f[j]; }
        }
      }
      for (int i = WBC_NB - 1; i >= 1; --i) {
        int p = md->parent[i];
        REAL u = -dot6(w->S[i], pD[i]);
        uD[i] = u;
        REAL f = u / w->D[i];
        for (int r = 0; r < 6; ++r) pD[p][r] += pD[i][r] + w->U[i][r] * f;
      }
      for (int r = 0; r < 6; ++r) aD[0][r] = -dot6(w->K[0] + r * 6, pD[0]);
      for (int i = 1; i < WBC_NB; ++i) {
        int p = md->parent[i];
        REAL qdd = (uD[i] - dot6(w->U[i], aD[p])) / w->D[i];
        qddD[i] = qdd;
        for (int k = 0; k < 6; ++k) aD[i][k] = aD[p][k] + w->S[i][k] * qdd;
      }
      /* the box: centre acceleration F/m, angular acceleration n/Ic */
      REAL bal[3], baa[3];
      for (int j = 0; j < 3; ++j) { boxF[j] = bF[j]; boxN[j] = bN[j]; bal[j] = boxF[j] / bx.m; baa[j] = boxN[j] / bx.Ic; }
      for (int k = 0; k < md->ncp; ++k) {
        contact_t* c = &ct[k];
        if (!c->active) continue;
        int b = sl[k].body, b2 = sl[k].body2;
        REAL t[3];
        for (int side = 0; side < (sl[k].kind == WBC_CP_TERRAIN ? 1 : 2); ++side) {
          int bb = side == 0 ? b : b2;
          REAL sgn = side == 0 ? 1 : -1, resp[3];
          if (bb == WBC_BOX_BODY) {
            REAL r[3] = {c->xc[0] - bx.c[0], c->xc[1] - bx.c[1], c->xc[2] - bx.c[2]};
            cross3(baa, r, t);
            for (int j = 0; j < 3; ++j) resp[j] = bal[j] + t[j];
          } else {
            cross3(aD[bb], c->xc, t);
            for (int j = 0; j < 3; ++j) resp[j] = aD[bb][3 + j] + t[j];
          }
          for (int j = 0; j < 3; ++j) dv[k][j] = (side == 0 ? 0 : dv[k][j]) + sgn * dt * resp[j];
        }
      }
    }
    for (int r = 0; r < 6; ++r) aD0[r] = aD[0][r];
  }
  if (g_dump) for (int k = 0; k < WBC_NCP; ++k) {
    g_dump->active[k] = k < md->ncp && ct[k].active;
    g_dump->nshare[k] = k < md->ncp ? ct[k].nshare : 0;
    for (int j = 0; j < 3; ++j) {
      g_dump->lam[k][j] = g_dump->active[k] ? (double)ct[k].lam[j] : 0;
      g_dump->n[k][j] = g_dump->active[k] ? (double)ct[k].n[j] : 0;
      g_dump->xc[k][j] = g_dump->active[k] ? (double)ct[k].xc[j] : 0;
    }
  }
  /* contact force outputs: world-frame net force per rigid body, foot-frame wrench per sensor */
  memset(e->contact_force, 0, sizeof(e->contact_force));
  memset(e->force_sensor, 0, sizeof(e->force_sensor));
  for (int k = 0; k < md->ncp; ++k) {
    contact_t* c = &ct[k];
    if (!c->active) continue;
    REAL f[3] = {c->lam[0] / dt, c->lam[1] / dt, c->lam[2] / dt}, fw[3];
    mat3_mul_vec(R, f, fw);
    int rb = sl[k].rb;
    for (int j = 0; j < 3; ++j) e->contact_force[rb][j] += fw[j];
    if (sl[k].kind != WBC_CP_TERRAIN)                /* PhysX reports pair forces in net_contact_force too */
      for (int j = 0; j < 3; ++j) e->contact_force[sl[k].rb2][j] -= fw[j];
    /* a foot's sensor sees every contact of the foot sphere: the terrain's, the box's, another limb's (on either side of the pair) */
    for (int ft = 0; ft < WBC_NFEET; ++ft) for (int side = 0; side < (sl[k].kind == WBC_CP_TERRAIN ? 1 : 2); ++side) {
      if (md->feet_rb[ft] != (side == 0 ? rb : sl[k].rb2)) continue;
      int b = side == 0 ? sl[k].body : sl[k].body2;
      REAL sg = side == 0 ? 1 : -1, fs[3] = {sg * f[0], sg * f[1], sg * f[2]};
      REAL fl[3], arm[3], tq[3], tl[3];
      mat3T_mul_vec(w->E[b], fs, fl);
      for (int j = 0; j < 3; ++j) arm[j] = (side == 0 ? -sl[k].rad : sl[k].rad2) * c->n[j];     /* from the foot's centre to the contact point */
      cross3(arm, fs, tq);
      mat3T_mul_vec(w->E[b], tq, tl);
      for (int j = 0; j < 3; ++j) { e->force_sensor[ft][j] += fl[j]; e->force_sensor[ft][3 + j] += tl[j]; }
    }
  }
  /* integrate: semi-implicit Euler */
  for (int i = 1; i < WBC_NB; ++i) {
    int d = md->dof[i];
    REAL qd = e->qd[d] + dt * (w->qdd[i] + qddD[i]);
    REAL lim = md->qd_limit[d];
    if (lim > 0) qd = clampr(qd, -lim, lim);
    e->qd[d] = qd;
    e->q[d] += dt * qd;
  }
  REAL a0[6];
  for (int k = 0; k < 3; ++k) { a0[k] = w->a[0][k] + aD0[k]; a0[3 + k] = w->a[0][3 + k] + aD0[3 + k] + gF[k]; }
  REAL wxv[3], accF[3], t[3];
  cross3(wb, vb, wxv);
  for (int k = 0; k < 3; ++k) accF[k] = a0[3 + k] + wxv[k];
  mat3_mul_vec(R, accF, t);
  for (int k = 0; k < 3; ++k) e->root[0][7 + k] += dt * t[k];
  mat3_mul_vec(R, a0, t);
  for (int k = 0; k < 3; ++k) e->root[0][10 + k] += dt * t[k];
  for (int k = 0; k < 3; ++k) e->root[0][k] += dt * e->root[0][7 + k];
  REAL* qt = e->root[0] + 3;
  REAL om[4] = {e->root[0][10], e->root[0][11], e->root[0][12], 0}, dq[4];
  quat_mul(om, qt, dq);
  REAL nq[4], nn = 0;
  for (int k = 0; k < 4; ++k) { nq[k] = qt[k] + (REAL)0.5 * dt * dq[k]; nn += nq[k] * nq[k]; }
  nn = 1 / sqrt(nn);
  for (int k = 0; k < 4; ++k) qt[k] = nq[k] * nn;
  /* the box: gravity + its net contact force; same integrator (asleep: frozen) */
  if (box_asleep) {
    for (int k = 7; k < 13; ++k) e->root[1][k] = 0;
  } else {
    REAL accB[3], alB[3];
    for (int k = 0; k < 3; ++k) { accB[k] = gF[k] + boxF[k] / bx.m; alB[k] = boxN[k] / bx.Ic; }
    mat3_mul_vec(R, accB, t);
    for (int k = 0; k < 3; ++k) e->root[1][7 + k] += dt * t[k];
    mat3_mul_vec(R, alB, t);
    for (int k = 0; k < 3; ++k) e->root[1][10 + k] += dt * t[k];
    for (int k = 0; k < 3; ++k) e->root[1][k] += dt * e->root[1][7 + k];
    REAL* qb = e->root[1] + 3;
    REAL omb[4] = {e->root[1][10], e->root[1][11], e->root[1][12], 0}, dqb[4];
    quat_mul(omb, qb, dqb);
    REAL nb4[4], nnb = 0;
    for (int k = 0; k < 4; ++k) { nb4[k] = qb[k] + (REAL)0.5 * dt * dqb[k]; nnb += nb4[k] * nb4[k]; }
    nnb = 1 / sqrt(nnb);
    for (int k = 0; k < 4; ++k) qb[k] = nb4[k] * nnb;
  }
}

/* rigid_body_state of the 27 robot bodies + the box (WG:546-556), world frame. */
static void update_rigid_body_state(const ora_sim* s, ora_env* e) {
  const wbc_model* md = &s->model;
  aba_ws W_; aba_ws* w = &W_;
  fk(s, e->q, w);
  REAL R[9]; quat_to_mat(e->root[0] + 3, R);
  /* world-frame velocity of each moving body: (omega, v at body origin) */
  REAL om[WBC_NB][3], vo[WBC_NB][3], quat[WBC_NB][4];
  for (int k = 0; k < 3; ++k) { om[0][k] = e->root[0][10 + k]; vo[0][k] = e->root[0][7 + k]; }
  for (int k = 0; k < 4; ++k) quat[0][k] = e->root[0][3 + k];
  for (int i = 1; i < WBC_NB; ++i) {
    int p = md->parent[i], ax = md->axis[i];
    REAL rel[3], relw[3], t[3];
    for (int k = 0; k < 3; ++k) rel[k] = w->pos[i][k] - w->pos[p][k];
    mat3_mul_vec(R, rel, relw);
    cross3(om[p], relw, t);
    for (int k = 0; k < 3; ++k) vo[i][k] = vo[p][k] + t[k];
    REAL sF[3] = {w->E[i][ax], w->E[i][3 + ax], w->E[i][6 + ax]}, sw[3];
    mat3_mul_vec(R, sF, sw);
    REAL qd = e->qd[md->dof[i]];
    for (int k = 0; k < 3; ++k) om[i][k] = om[p][k] + sw[k] * qd;
    REAL h = (REAL)0.5 * e->q[md->dof[i]], qa[4] = {0, 0, 0, cos(h)};
    qa[ax] = sin(h);
    quat_mul(quat[p], qa, quat[i]);
  }
  for (int r = 0; r < WBC_NRB; ++r) {
    int b = md->rb_body[r];
    REAL off[3] = {md->rb_offset[r][0], md->rb_offset[r][1], md->rb_offset[r][2]}, t[3], pf[3], pw[3], t2[3];
    mat3_mul_vec(w->E[b], off, t);
    for (int k = 0; k < 3; ++k) pf[k] = w->pos[b][k] + t[k];
    mat3_mul_vec(R, pf, pw);
    REAL offw[3];
    mat3_mul_vec(R, t, offw);
    cross3(om[b], offw, t2);
    for (int k = 0; k < 3; ++k) {
      e->rb_state[r][k] = e->root[0][k] + pw[k];
      e->rb_state[r][7 + k] = vo[b][k] + t2[k];
      e->rb_state[r][10 + k] = om[b][k];
    }
    for (int k = 0; k < 4; ++k) e->rb_state[r][3 + k] = quat[b][k];
  }
  for (int k = 0; k < 13; ++k) e->rb_state[WBC_NRB][k] = e->root[1][k];
}

/* ------------------------------------------------------------ env logic -- */
static const int POLICY_PERM[WBC_NDOF] = {3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8, 12, 13, 14, 15, 16, 17, 18, 19};
static const int FEET_PERM[4] = {1, 0, 3, 2};   /* WG:1007 */

/* _compute_torques, WG:1262-1295 (adaptive_arm_gains False, torque_supervision False) */
static void compute_torques(const ora_sim* s, ora_env* e) {
  const wbc_task_cfg* cf = &s->cfg;
  for (int j = 0; j < WBC_NACT; ++j) {
    REAL a_s = e->actions[j] * e->motor_strength[j] * (REAL)cf->action_scale[j];           /* WG:1276 */
    REAL qw = e->q[j];
    if (j == WBC_NACT - 8) qw = wrap_to_pi(qw);     /* WG:1279: column -8 of the 18-wide view (quirk Q2) */
    REAL t = (REAL)cf->p_gains[j] * (a_s + (REAL)cf->default_dof_pos[j] - qw) - (REAL)cf->d_gains[j] * e->qd[j];  /* WG:1281 */
    e->torques[j] = clampr(t, -(REAL)cf->torque_limits[j], (REAL)cf->torque_limits[j]);   /* WG:1295 */
  }
  for (int j = WBC_NACT; j < WBC_NDOF; ++j) e->torques[j] = 0;                              /* WG:1290 */
}

/* _resample_commands for one env, WG:831-843 */
static void resample_commands(const ora_sim* s, ora_env* e, int env, uint64_t step, int slot) {
  const wbc_curriculum* cu = &s->cur;
  REAL cx = rng_range((REAL)cu->lin_vel_x_range[0], (REAL)cu->lin_vel_x_range[1], s->seed, env, step, slot);
  REAL cy = rng_range((REAL)cu->ang_vel_yaw_range[0], (REAL)cu->ang_vel_yaw_range[1], s->seed, env, step, slot + 1);
  int keep = (cx > (REAL)s->cfg.lin_vel_x_clip) || (fabs(cy) > (REAL)s->cfg.ang_vel_yaw_clip);
  e->commands[0] = keep ? cx : 0;
  e->commands[1] = 0;
  e->commands[2] = keep ? cy : 0;
}

/* collision_check for one env, WG:1337-1342 */
static int goal_collision(const ora_sim* s, const ora_env* e) {
  const wbc_task_cfg* cf = &s->cfg;
  int ns = cf->goal_collision_samples, hit = 0;
  for (int k = 0; k < ns; ++k) {
    REAL t = (ns > 1) ? (REAL)k / (REAL)(ns - 1) : 0;
    REAL sp[3], c[3];
    for (int j = 0; j < 3; ++j) sp[j] = lerp_torch(e->goal[G_START + j], e->goal[G_GOAL + j], t);
    sphere2cart(sp, c);
    int inside = 1;
    for (int j = 0; j < 3; ++j) inside &= (c[j] < (REAL)cf->goal_collision_upper[j]) && (c[j] > (REAL)cf->goal_collision_lower[j]);
    hit |= inside;
    hit |= c[2] < (REAL)cf->goal_underground_limit;
  }
  return hit;
}

/* _resample_ee_goal for one env, WG:1316-1332; base_yaw is the pre-reset yaw (WG:1313) */
static void resample_ee_goal(const ora_sim* s, ora_env* e, int env, uint64_t step, int slot_orn, int slot_sph, REAL base_yaw) {
  const wbc_task_cfg* cf = &s->cfg;
  const wbc_curriculum* cu = &s->cur;
  for (int j = 0; j < 3; ++j) {
    REAL d = rng_range((REAL)cf->goal_delta_orn_range[j][0], (REAL)cf->goal_delta_orn_range[j][1], s->seed, env, step, slot_orn + j);
    e->goal[G_DORN + j] = d;
    e->goal[G_ORN + j] = wrap_to_pi(d + (j == 2 ? base_yaw : 0));
  }
  for (int j = 0; j < 3; ++j) e->goal[G_START + j] = e->goal[G_GOAL + j];
  for (int r = 0; r < 10; ++r) {
    e->goal[G_GOAL + 0] = rng_range((REAL)cu->goal_l_range[0], (REAL)cu->goal_l_range[1], s->seed, env, step, slot_sph + 3 * r);
    e->goal[G_GOAL + 1] = rng_range((REAL)cu->goal_p_range[0], (REAL)cu->goal_p_range[1], s->seed, env, step, slot_sph + 3 * r + 1);
    e->goal[G_GOAL + 2] = rng_range((REAL)cu->goal_y_range[0], (REAL)cu->goal_y_range[1], s->seed, env, step, slot_sph + 3 * r + 2);
    if (!goal_collision(s, e)) break;
  }
  sphere2cart(e->goal + G_GOAL, e->goal + G_GOAL_CART);
  e->goal[G_TIMER] = 0;
}

/* reset_idx for one env, WG:695-754 */
static void reset_env(const ora_sim* s, ora_env* e, int env, uint64_t step, int start, REAL base_yaw) {
  const wbc_task_cfg* cf = &s->cfg;
  for (int j = 0; j < WBC_NDOF; ++j) {                                                   /* _reset_dofs WG:824-825 */
    e->q[j] = (REAL)cf->default_dof_pos[j] * rng_range((REAL)cf->dof_reset_lo, (REAL)cf->dof_reset_hi, s->seed, env, step, SLOT_RESET_DOF + j);
    e->qd[j] = 0;
  }
  {                                                                                       /* _update_terrain_curriculum's inputs, LR:431-435 */
    const REAL dx = e->root[0][0] - e->env_origin[0], dy = e->root[0][1] - e->env_origin[1];
    e->reset_travel[0] = sqrt(dx * dx + dy * dy);
    e->reset_travel[1] = sqrt(e->commands[0] * e->commands[0] + e->commands[1] * e->commands[1]);
  }
  for (int k = 0; k < 13; ++k) e->root[0][k] = (REAL)cf->base_init_state[k];              /* WG:765 */
  for (int k = 0; k < 3; ++k) e->root[0][k] += e->env_origin[k];                          /* WG:766 */
  for (int k = 0; k < 2; ++k)
    e->root[0][k] += rng_range(-(REAL)cf->origin_perturb_range, (REAL)cf->origin_perturb_range, s->seed, env, step, SLOT_RESET_XY + k);
  e->root[1][0] = (REAL)cf->box_origin_x;                                                 /* WG:769-771 */
  e->root[1][1] = e->root[0][1] + e->box_delta_y;
  e->root[1][2] = (REAL)cf->box_origin_z;
  for (int k = 0; k < 6; ++k)                                                             /* WG:774 */
    e->root[0][7 + k] = rng_range(-(REAL)cf->init_vel_perturb_range, (REAL)cf->init_vel_perturb_range, s->seed, env, step, SLOT_RESET_VEL + k);
  if (start || e->time_out) resample_commands(s, e, env, step, SLOT_RESET_CMD);           /* WG:723-727 */
  resample_ee_goal(s, e, env, step, SLOT_RESET_GOAL_ORN, SLOT_RESET_GOAL_SPHERE, base_yaw); /* WG:729 */
  memset(e->last_actions, 0, sizeof(e->last_actions));                                    /* WG:732-739 */
  memset(e->last_dof_vel, 0, sizeof(e->last_dof_vel));
  memset(e->feet_air_time, 0, sizeof(e->feet_air_time));                                  /* WG:734 */
  e->episode_length = 0;
  e->reset_buf = 1;
  memset(e->obs_hist, 0, sizeof(e->obs_hist));
  memset(e->act_hist, 0, sizeof(e->act_hist));
  e->goal[G_TIMER] = 0;
  memcpy(e->episode_sums_done, e->episode_sums, sizeof(e->episode_sums));                  /* WG:743-750 */
  memcpy(e->metric_sums_done, e->metric_sums, sizeof(e->metric_sums));
  memset(e->episode_sums, 0, sizeof(e->episode_sums));
  memset(e->metric_sums, 0, sizeof(e->metric_sums));
}

/* compute_reward, WG:170-205 and the _reward_* functions WG:1352-1469 */
static void compute_reward(const ora_sim* s, ora_env* e, const REAL* base_yaw_quat) {
  const wbc_task_cfg* cf = &s->cfg;
  const wbc_curriculum* cu = &s->cur;
  REAL term[WBC_NREW];
  int met_used[WBC_NMETRIC] = {0};
  REAL met_val[WBC_NMETRIC] = {0};
  const REAL* ee_pos = e->rb_state[s->model.gripper_rb];
  const REAL* ee_orn = e->rb_state[s->model.gripper_rb] + 3;
  REAL sq = 0, abs_sum = 0, sum = 0, arm_abs = 0, tq2 = 0, act_leg = 0;
  for (int j = 0; j < 12; ++j) {
    REAL p = e->torques[j] * e->qd[j];
    sq += p * p; abs_sum += fabs(p); sum += p; act_leg += e->actions[j] * e->actions[j];
  }
  for (int j = 12; j < WBC_NDOF - 2; ++j) arm_abs += fabs(e->torques[j] * e->qd[j]);
  for (int j = 0; j < WBC_NDOF; ++j) tq2 += e->torques[j] * e->torques[j];
  term[WBC_REW_ENERGY_SQUARE] = sq;                                                       /* WG:1466-1469 */
  term[WBC_REW_SURVIVE] = 1;                                                              /* WG:1452 */
  REAL ex = fabs(e->commands[0] - e->base_lin_vel[0]);
  term[WBC_REW_TRACKING_LIN_VEL_X_L1] = -ex + fabs(e->commands[0]);                       /* WG:1427-1430 */
  term[WBC_REW_TRACKING_LIN_VEL_X_EXP] = exp(-ex / (REAL)cf->tracking_sigma);             /* WG:1432-1435 */
  REAL eyaw = fabs(e->commands[2] - e->base_ang_vel[2]);
  term[WBC_REW_TRACKING_ANG_VEL_YAW_EXP] = exp(-eyaw / (REAL)cf->tracking_sigma);         /* WG:1441-1444 */
  term[WBC_REW_TRACKING_ANG_VEL_YAW_L1] = -eyaw + fabs(e->commands[2]);                   /* WG:1437-1439 */
  REAL hip = e->actions[0] * e->actions[0] + e->actions[3] * e->actions[3] + e->actions[6] * e->actions[6] + e->actions[9] * e->actions[9];
  term[WBC_REW_HIP_ACTION_L2] = hip;                                                      /* WG:1379-1382 */
  REAL fz = 0;
  for (int f = 0; f < 4; ++f) fz += e->force_sensor[f][2] * e->force_sensor[f][2];
  term[WBC_REW_FOOT_CONTACTS_Z] = fz;                                                     /* WG:1455-1458 */
  /* tracking_ee_sphere WG:1352-1358 */
  REAL rel[3] = {ee_pos[0] - e->root[0][0], ee_pos[1] - e->root[0][1], ee_pos[2] - (REAL)cf->z_invariant_offset}, loc[3], sph[3];
  quat_rotate_inverse(base_yaw_quat, rel, loc);
  cart2sphere(loc, sph);
  REAL es = 0;
  for (int j = 0; j < 3; ++j) es += fabs(sph[j] - e->goal[G_CURR + j]) * (REAL)cf->sphere_error_scale[j];
  term[WBC_REW_TRACKING_EE_SPHERE] = exp(-es / (REAL)cf->tracking_ee_sigma);
  /* tracking_ee_cart WG:1360-1366: target = [x,y,0.53] + quat_apply(yaw_quat, curr_cart) */
  REAL yq_inv[4] = {-base_yaw_quat[0], -base_yaw_quat[1], -base_yaw_quat[2], base_yaw_quat[3]}, tw[3];
  quat_rotate_inverse(yq_inv, e->goal + G_CURR_CART, tw);
  REAL ec = fabs(ee_pos[0] - (e->root[0][0] + tw[0])) + fabs(ee_pos[1] - (e->root[0][1] + tw[1])) + fabs(ee_pos[2] - ((REAL)cf->z_invariant_offset + tw[2]));
  term[WBC_REW_TRACKING_EE_CART] = exp(-ec / (REAL)cf->tracking_ee_sigma);
  /* tracking_ee_orn / _ry WG:1368-1393 */
  REAL eul[3], eo = 0, eo_ry = 0;
  euler_from_quat(ee_orn, eul);
  for (int j = 0; j < 3; ++j) {
    REAL d = wrap_to_pi(e->goal[G_ORN + j] - eul[j]);
    eo += fabs(d) * (REAL)cf->orn_error_scale[j];
    if (j != 1) eo_ry += fabs(d * (REAL)cf->orn_error_scale[j]);
  }
  term[WBC_REW_TRACKING_EE_ORN] = exp(-eo / (REAL)cf->tracking_ee_sigma);
  term[WBC_REW_TRACKING_EE_ORN_RY] = exp(-eo_ry / (REAL)cf->tracking_ee_sigma);
  term[WBC_REW_LEG_ENERGY_ABS_SUM] = abs_sum;                                             /* WG:1396-1399 */
  term[WBC_REW_LEG_ENERGY_SUM_ABS] = fabs(sum);                                           /* WG:1401-1403 */
  term[WBC_REW_LEG_ACTION_L2] = act_leg;                                                  /* WG:1405-1408 */
  term[WBC_REW_LEG_ENERGY] = sum;                                                         /* WG:1410-1412 */
  term[WBC_REW_ARM_ENERGY_ABS_SUM] = arm_abs;                                             /* WG:1414-1415 */
  REAL lv = (e->commands[0] - e->base_lin_vel[0]) * (e->commands[0] - e->base_lin_vel[0]) + (e->commands[1] - e->base_lin_vel[1]) * (e->commands[1] - e->base_lin_vel[1]);
  term[WBC_REW_TRACKING_LIN_VEL] = exp(-lv / (REAL)cf->tracking_sigma);                   /* WG:1422-1425 */
  term[WBC_REW_TRACKING_LIN_VEL_Y_L2] = (e->commands[1] - e->base_lin_vel[1]) * (e->commands[1] - e->base_lin_vel[1]); /* WG:1446-1447 */
  term[WBC_REW_TRACKING_LIN_VEL_Z_L2] = (e->commands[2] - e->base_lin_vel[2]) * (e->commands[2] - e->base_lin_vel[2]); /* WG:1449-1450 */
  term[WBC_REW_TORQUES] = tq2;                                                            /* WG:1460-1464 */
  REAL ncol = 0;                                                                          /* LR:865-867 (base class) */
  for (int rb = 0; rb < WBC_NRB; ++rb)
    if ((cf->penalize_contact_rb_mask >> rb) & 1u) ncol += (sqrt(dot3(e->contact_force[rb], e->contact_force[rb])) > (REAL)0.1) ? 1 : 0;
  term[WBC_REW_COLLISION] = ncol;
  /* ---- the base class's terms (LR = legged_gym/envs/base/legged_robot.py) ---- */
  const REAL dtp = (REAL)cf->sim_dt * (REAL)cf->decimation;                              /* self.dt, WG:80 */
  term[WBC_REW_LIN_VEL_Z] = e->base_lin_vel[2] * e->base_lin_vel[2];                      /* LR:833-835 */
  term[WBC_REW_ANG_VEL_XY] = e->base_ang_vel[0] * e->base_ang_vel[0] + e->base_ang_vel[1] * e->base_ang_vel[1];   /* LR:837-839 */
  REAL dv2 = 0, da2 = 0, ar2 = 0, plim = 0, vlim = 0, tlim = 0, still = 0;
  for (int j = 0; j < WBC_NDOF; ++j) {
    dv2 += e->qd[j] * e->qd[j];                                                           /* LR:854-856 */
    REAL acc = (e->last_dof_vel[j] - e->qd[j]) / dtp;                                     /* LR:858-860 */
    da2 += acc * acc;
    REAL below = e->q[j] - (REAL)cf->soft_dof_lower[j], above = e->q[j] - (REAL)cf->soft_dof_upper[j];   /* LR:873-877 */
    plim += -(below < 0 ? below : 0) + (above > 0 ? above : 0);
    vlim += clampr(fabs(e->qd[j]) - (REAL)cf->soft_dof_vel_limit[j], 0, 1);               /* LR:879-882 */
    REAL over = fabs(e->torques[j]) - (REAL)cf->soft_torque_limit[j];                     /* LR:884-886 */
    tlim += over > 0 ? over : 0;
    still += fabs(e->q[j] - (REAL)cf->default_dof_pos[j]);                                /* LR:916-918 */
  }
  for (int j = 0; j < WBC_NACT; ++j) ar2 += (e->last_actions[j] - e->actions[j]) * (e->last_actions[j] - e->actions[j]);   /* LR:862-864 */
  term[WBC_REW_DOF_VEL] = dv2; term[WBC_REW_DOF_ACC] = da2; term[WBC_REW_ACTION_RATE] = ar2;
  term[WBC_REW_TERMINATION] = (e->reset_buf && !e->time_out) ? 1 : 0;                     /* LR:869-871 */
  term[WBC_REW_DOF_POS_LIMITS] = plim; term[WBC_REW_DOF_VEL_LIMITS] = vlim; term[WBC_REW_TORQUE_LIMITS] = tlim;
  REAL eyaw2 = (e->commands[2] - e->base_ang_vel[2]) * (e->commands[2] - e->base_ang_vel[2]);
  term[WBC_REW_TRACKING_ANG_VEL] = exp(-eyaw2 / (REAL)cf->tracking_sigma);                /* LR:893-896 */
  const REAL cmd_xy = sqrt(e->commands[0] * e->commands[0] + e->commands[1] * e->commands[1]);
  term[WBC_REW_STAND_STILL] = still * (cmd_xy < (REAL)0.1 ? 1 : 0);
  REAL stumble = 0, fcf = 0;
  for (int f = 0; f < WBC_NFEET; ++f) {
    const REAL* cf3 = e->contact_force[s->model.feet_rb[f]];
    if (sqrt(cf3[0] * cf3[0] + cf3[1] * cf3[1]) > 5 * fabs(cf3[2])) stumble = 1;          /* LR:911-914 */
    REAL over = sqrt(dot3(cf3, cf3)) - (REAL)cf->max_contact_force;                       /* LR:920-922 */
    fcf += over > 0 ? over : 0;
  }
  term[WBC_REW_STUMBLE] = stumble; term[WBC_REW_FEET_CONTACT_FORCES] = fcf;
  { REAL bh = e->root[0][2] - (REAL)cf->base_height_target; term[WBC_REW_BASE_HEIGHT] = bh * bh; }   /* LR:845-848, measured_heights = 0 (WG:639) */
  /* feet_air_time (LR:898-909): its state advances only when the function is in a reward list */
  term[WBC_REW_FEET_AIR_TIME] = 0;
  if (((cu->leg_active_mask | cu->arm_active_mask) >> WBC_REW_FEET_AIR_TIME) & 1u) {
    REAL rew = 0;
    for (int f = 0; f < WBC_NFEET; ++f) {
      const int contact = e->contact_force[s->model.feet_rb[f]][2] > (REAL)1.0;
      const int filt = contact || (e->last_contacts[f] != 0);
      e->last_contacts[f] = contact;
      const int first = (e->feet_air_time[f] > 0) && filt;
      e->feet_air_time[f] += dtp;
      rew += (e->feet_air_time[f] - (REAL)0.5) * first;
      if (filt) e->feet_air_time[f] = 0;
    }
    term[WBC_REW_FEET_AIR_TIME] = rew * (cmd_xy > (REAL)0.1 ? 1 : 0);
  }
  /* metric side effects of the reward functions, applied once per ACTIVE call */
  static const int met_of[WBC_NREW] = {
    WBC_MET_ENERGY_SQUARE, -1, WBC_MET_TRACKING_LIN_VEL_X_L1, WBC_MET_TRACKING_ANG_VEL_YAW_EXP, WBC_MET_LEG_ACTION_L2,
    WBC_MET_FOOT_CONTACTS_Z, WBC_MET_TRACKING_EE_SPHERE, -1, WBC_MET_TRACKING_EE_CART, -1, WBC_MET_TRACKING_EE_ORN,
    WBC_MET_LEG_ENERGY_ABS_SUM, -1, WBC_MET_LEG_ACTION_L2, -1, -1, WBC_MET_TRACKING_LIN_VEL_X_L1, -1, -1, -1, WBC_MET_TORQUE, -1,
    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};          /* the base class's terms touch no metric */
  REAL met_src[WBC_NREW] = {0};
  met_src[WBC_REW_ENERGY_SQUARE] = sq; met_src[WBC_REW_TRACKING_LIN_VEL_X_L1] = ex; met_src[WBC_REW_TRACKING_LIN_VEL_X_EXP] = ex;
  met_src[WBC_REW_TRACKING_ANG_VEL_YAW_EXP] = eyaw; met_src[WBC_REW_HIP_ACTION_L2] = hip; met_src[WBC_REW_LEG_ACTION_L2] = act_leg;
  met_src[WBC_REW_FOOT_CONTACTS_Z] = fz; met_src[WBC_REW_TRACKING_EE_SPHERE] = es; met_src[WBC_REW_TRACKING_EE_CART] = ec;
  met_src[WBC_REW_TRACKING_EE_ORN_RY] = eo_ry; met_src[WBC_REW_LEG_ENERGY_ABS_SUM] = abs_sum; met_src[WBC_REW_TORQUES] = tq2;
  (void)met_used; (void)met_val;
  REAL r = 0, ra = 0;
  for (int t = 0; t < WBC_NREW; ++t) {
    REAL sc = (REAL)cu->leg_reward_scale[t];
    if (t != WBC_REW_TERMINATION && ((cu->leg_active_mask >> t) & 1u)) {                  /* WG:176-180; list built at WG:128-142 */
      REAL v = term[t] * sc;
      r += v; e->episode_sums[t] += v;
      if (met_of[t] >= 0) e->metric_sums[met_of[t]] += met_src[t];
    }
  }
  if (cf->only_positive_rewards && r < 0) r = 0;                                          /* WG:181-182 */
  if ((cu->leg_active_mask >> WBC_REW_TERMINATION) & 1u) {                                /* after the clip, WG:184-188 */
    REAL v = term[WBC_REW_TERMINATION] * (REAL)cu->leg_reward_scale[WBC_REW_TERMINATION];
    r += v; e->episode_sums[WBC_REW_TERMINATION] += v;
  }
  e->rew = r / 100;                                                                       /* WG:189 */
  for (int t = 0; t < WBC_NREW; ++t) {
    REAL sc = (REAL)cu->arm_reward_scale[t];
    if (t != WBC_REW_TERMINATION && ((cu->arm_active_mask >> t) & 1u)) {                  /* WG:192-196; list built at WG:144-157 */
      REAL v = term[t] * sc;
      ra += v; e->episode_sums[t] += v;
      if (met_of[t] >= 0) e->metric_sums[met_of[t]] += met_src[t];
    }
  }
  if (cf->only_positive_rewards && ra < 0) ra = 0;
  if ((cu->arm_active_mask >> WBC_REW_TERMINATION) & 1u) {                                /* WG:200-203 */
    REAL v = term[WBC_REW_TERMINATION] * (REAL)cu->arm_reward_scale[WBC_REW_TERMINATION];
    ra += v; e->episode_sums[WBC_REW_TERMINATION] += v;
  }
  e->arm_rew = ra / 100;                                                                  /* WG:205 */
}

/* compute_observations, WG:966-1001 (index map: SURVEY.md Appendix A) */
static void compute_observations(const ora_sim* s, ora_env* e) {
  const wbc_task_cfg* cf = &s->cfg;
  REAL o[WBC_NPROP], rpy[3];
  euler_from_quat(e->root[0] + 3, rpy);                                                   /* WG:973,1101-1106 */
  o[0] = rpy[0]; o[1] = rpy[1];
  for (int k = 0; k < 3; ++k) o[2 + k] = e->base_ang_vel[k] * (REAL)cf->obs_scale_ang_vel; /* WG:974 */
  for (int j = 0; j < WBC_NDOF; ++j) {
    int sj = POLICY_PERM[j];
    REAL qw = e->q[sj];
    if (sj == WBC_NDOF - 8) qw = wrap_to_pi(qw);                                          /* WG:970 */
    o[5 + j] = (qw - (REAL)cf->default_dof_pos[sj]) * (REAL)cf->obs_scale_dof_pos;        /* WG:975 */
    o[25 + j] = e->qd[sj] * (REAL)cf->obs_scale_dof_vel;                                  /* WG:976 */
  }
  for (int j = 0; j < WBC_NACT; ++j) o[45 + j] = e->act_hist[WBC_ADELAY_LEN - 1][POLICY_PERM[j]]; /* WG:977 */
  for (int f = 0; f < 4; ++f) {                                                           /* WG:978,1095 */
    const REAL* fs = e->force_sensor[FEET_PERM[f]];
    REAL nrm = sqrt(dot6(fs, fs));
    o[63 + f] = nrm > (REAL)1.5 ? 1 : 0;
  }
  for (int k = 0; k < 3; ++k) o[67 + k] = e->commands[k] * (REAL)cf->commands_scale[k];   /* WG:979 */
  for (int k = 0; k < 3; ++k) o[70 + k] = e->goal[(cf->goal_command_cart ? G_CURR_CART : G_CURR) + k];   /* WG:980; curr_ee_goal per command_mode, WG:589-593 */
  for (int k = 0; k < 3; ++k) o[73 + k] = e->goal[G_DORN + k];                            /* WG:981 */
  REAL* ob = e->obs;
  for (int k = 0; k < WBC_NPROP; ++k) ob[k] = o[k];
  for (int k = 0; k < 5; ++k) ob[WBC_NPROP + k] = e->mass_params[k];                      /* WG:987-991 */
  ob[WBC_NPROP + 5] = e->friction;
  for (int k = 0; k < WBC_NACT; ++k) ob[WBC_NPROP + 6 + k] = e->motor_strength[k] - 1;
  for (int h = 0; h < WBC_HIST; ++h) for (int k = 0; k < WBC_NPROP; ++k) ob[WBC_NPROP + WBC_NPRIV + h * WBC_NPROP + k] = e->obs_hist[h][k];  /* WG:992 */
  if (e->episode_length <= 1) {                                                           /* WG:994-1001 */
    for (int h = 0; h < WBC_HIST; ++h) memcpy(e->obs_hist[h], o, sizeof(o));
  } else {
    memmove(e->obs_hist[0], e->obs_hist[1], sizeof(REAL) * WBC_NPROP * (WBC_HIST - 1));
    memcpy(e->obs_hist[WBC_HIST - 1], o, sizeof(o));
  }
  for (int k = 0; k < WBC_NOBS; ++k) ob[k] = clampr(ob[k], -(REAL)cf->clip_obs, (REAL)cf->clip_obs);   /* WG:1195-1196 */
}

/* WidowGo1.step for one env, WG:1156-1199. `step` = common_step_counter after increment. */
static void env_step(const ora_sim* s, ora_env* e, int env, const REAL* actions_policy, uint64_t step) {
  const wbc_task_cfg* cf = &s->cfg;
  REAL act[WBC_NACT];
  for (int j = 0; j < WBC_NACT; ++j)                                                      /* WG:1162-1163 */
    act[j] = clampr(actions_policy[POLICY_PERM[j]], -(REAL)cf->clip_actions, (REAL)cf->clip_actions);
  if (cf->action_delay != -1) {                                                           /* WG:1166-1168 */
    memmove(e->act_hist[0], e->act_hist[1], sizeof(REAL) * WBC_NACT * (WBC_ADELAY_LEN - 1));
    memcpy(e->act_hist[WBC_ADELAY_LEN - 1], act, sizeof(act));
    memcpy(e->actions, e->act_hist[WBC_ADELAY_LEN - cf->action_delay - 1], sizeof(act));
  } else {
    memcpy(e->actions, act, sizeof(act));
  }
  for (int t = 0; t < cf->decimation; ++t) {                                              /* WG:1175-1191 */
    compute_torques(s, e);
    physics_substep(s, e);
  }
  /* post_physics_step, WG:865-915 */
  update_rigid_body_state(s, e);                                                          /* WG:870-873 refresh */
  e->episode_length += 1;                                                                 /* WG:875 */
  quat_rotate_inverse(e->root[0] + 3, e->root[0] + 7, e->base_lin_vel);                   /* WG:880 */
  quat_rotate_inverse(e->root[0] + 3, e->root[0] + 10, e->base_ang_vel);                  /* WG:881 */
  REAL rpy[3];
  euler_from_quat(e->root[0] + 3, rpy);                                                   /* WG:882 */
  REAL base_yaw = rpy[2];
  REAL yq[4] = {0, 0, sin(base_yaw * (REAL)0.5), cos(base_yaw * (REAL)0.5)};              /* WG:884 */
  /* update_curr_ee_goal, WG:1344-1350 */
  REAL tt = clampr(e->goal[G_TIMER] / e->goal[G_TRAJ], 0, 1);
  for (int j = 0; j < 3; ++j) e->goal[G_CURR + j] = lerp_torch(e->goal[G_START + j], e->goal[G_GOAL + j], tt);
  sphere2cart(e->goal + G_CURR, e->goal + G_CURR_CART);
  e->goal[G_TIMER] += 1;
  if (e->goal[G_TIMER] > e->goal[G_TOTAL]) resample_ee_goal(s, e, env, step, SLOT_GOAL_ORN, SLOT_GOAL_SPHERE, base_yaw);
  /* _post_physics_step_callback, WG:917-935 */
  if (e->episode_length % cf->resample_interval == 0) resample_commands(s, e, env, step, SLOT_CMD);
  if (cf->push_interval > 0 && (step % (uint64_t)cf->push_interval) == 0) {               /* _push_robots WG:804-814 */
    REAL px = rng_range(-(REAL)cf->max_push_vel, (REAL)cf->max_push_vel, s->seed, env, step, SLOT_PUSH);
    REAL py = rng_range(-(REAL)cf->max_push_vel, (REAL)cf->max_push_vel, s->seed, env, step, SLOT_PUSH + 1);
    REAL k = ((e->commands[0] + e->commands[1] + e->commands[2]) == 0) ? (REAL)2.5 : 1;
    e->root[0][7] = px * k; e->root[0][8] = py * k;
  }
  /* check_termination, WG:937-963 (the shipped contact list is empty, widowGo1_config.py:179) */
  int c_term = 0;                                                                         /* WG:940 */
  for (int rb = 0; rb < WBC_NRB; ++rb)
    if ((cf->term_contact_rb_mask >> rb) & 1u) c_term |= sqrt(dot3(e->contact_force[rb], e->contact_force[rb])) > (REAL)1.0;
  REAL r = rpy[0], p = rpy[1], z = e->root[0][2], th = (REAL)cf->term_rp_threshold;
  const int gc = cf->goal_command_cart ? G_CURR_CART : G_CURR;                              /* curr_ee_goal (WG:589-593) */
  int r_term = ((r > th) && (e->goal[gc + 2] >= 0)) || ((r < -th) && (e->goal[gc + 2] <= 0));
  int p_term = ((p > th) && (e->goal[gc + 1] >= 0)) || ((p < -th) && (e->goal[gc + 1] <= 0));
  int z_term = z < (REAL)cf->term_z_threshold;
  e->time_out = e->episode_length > cf->max_episode_length;
  e->reset_buf = c_term | r_term | p_term | z_term | e->time_out;
  compute_reward(s, e, yq);                                                               /* WG:897 */
  if (e->reset_buf) reset_env(s, e, env, step, 0, base_yaw);                              /* WG:898-899 */
  compute_observations(s, e);                                                             /* WG:900 */
  memcpy(e->last_actions, e->actions, sizeof(e->actions));                                /* WG:908-910 */
  memcpy(e->last_dof_vel, e->qd, sizeof(e->qd));
  for (int k = 0; k < 6; ++k) e->last_root_vel[k] = e->root[0][7 + k];
}

/* --------------------------------------------------------------- C API --- */
#ifdef _WIN32
#define ORA_API
#else
#define ORA_API __attribute__((visibility("default")))
#endif

ORA_API int ora_real_bytes(void) { return (int)sizeof(REAL); }
/* threads ora_step spreads the envs over (bench.py reports it with the oracle's throughput) */
ORA_API int ora_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
/* sizeof of the ABI structs, so the ctypes mirrors in wbc_amd/abi.py can be checked */
ORA_API void ora_abi_sizes(int* out) { out[0] = (int)sizeof(wbc_model); out[1] = (int)sizeof(wbc_task_cfg); out[2] = (int)sizeof(wbc_curriculum); }

ORA_API ora_sim* ora_create(const wbc_model* model, const wbc_task_cfg* cfg, int n, uint64_t seed) {
  ora_sim* s = (ora_sim*)calloc(1, sizeof(ora_sim));
  s->model = *model; s->cfg = *cfg; s->n = n; s->seed = seed;
  s->env = (ora_env*)calloc((size_t)n, sizeof(ora_env));
  for (int i = 0; i < n; ++i) {
    ora_env* e = &s->env[i];
    e->root[0][6] = 1; e->root[1][6] = 1;
    for (int k = 0; k < WBC_NACT; ++k) e->motor_strength[k] = 1;
    e->friction = 1;
    e->goal[G_TRAJ] = 100; e->goal[G_TOTAL] = 150;
    e->box_mass = model->box_mass;
    e->body_params[0] = model->mass[0];
    for (int k = 0; k < 3; ++k) e->body_params[1 + k] = model->com[0][k];
    for (int k = 0; k < 6; ++k) e->body_params[4 + k] = model->inertia[0][k];
    int g = model->gripper_body;
    e->body_params[10] = model->mass[g];
    for (int k = 0; k < 3; ++k) e->body_params[11 + k] = model->com[g][k];
    for (int k = 0; k < 6; ++k) e->body_params[14 + k] = model->inertia[g][k];
  }
  return s;
}
ORA_API void ora_destroy(ora_sim* s) { if (s) { free(s->env); free(s->hf); free(s); } }
ORA_API void ora_set_curriculum(ora_sim* s, const wbc_curriculum* c) { s->cur = *c; }
ORA_API void ora_set_step_counter(ora_sim* s, int64_t v) { s->step_counter = v; }
ORA_API int64_t ora_get_step_counter(ora_sim* s) { return s->step_counter; }
ORA_API void ora_set_heightfield(ora_sim* s, const int16_t* h, int rows, int cols, double hs, double vs, double tx, double ty, double tz) {
  free(s->hf); s->hf = NULL;
  if (!h) return;
  s->hf = (int16_t*)malloc(sizeof(int16_t) * (size_t)rows * cols);
  memcpy(s->hf, h, sizeof(int16_t) * (size_t)rows * cols);
  s->hf_rows = rows; s->hf_cols = cols; s->hf_hs = (REAL)hs; s->hf_vs = (REAL)vs; s->hf_t[0] = (REAL)tx; s->hf_t[1] = (REAL)ty; s->hf_t[2] = (REAL)tz;
}

/* field access by wbc_tensor_id; data as double[], row-major, shapes of include/wbc_sim.h */
static int field_ptr(ora_env* e, int id, REAL** p, int* n) {
  switch (id) {
    case WBC_T_ROOT_STATES: *p = &e->root[0][0]; *n = 26; return 0;
    case WBC_T_NET_CONTACT_FORCE: *p = &e->contact_force[0][0]; *n = WBC_NRB_ENV * 3; return 0;
    case WBC_T_RIGID_BODY_STATE: *p = &e->rb_state[0][0]; *n = WBC_NRB_ENV * 13; return 0;
    case WBC_T_FORCE_SENSOR: *p = &e->force_sensor[0][0]; *n = 24; return 0;
    case WBC_T_TORQUES: *p = e->torques; *n = WBC_NDOF; return 0;
    case WBC_T_OBS_BUF: *p = e->obs; *n = WBC_NOBS; return 0;
    case WBC_T_OBS_HISTORY: *p = &e->obs_hist[0][0]; *n = WBC_HIST * WBC_NPROP; return 0;
    case WBC_T_ACTION_HISTORY: *p = &e->act_hist[0][0]; *n = WBC_ADELAY_LEN * WBC_NACT; return 0;
    case WBC_T_ACTIONS: *p = e->actions; *n = WBC_NACT; return 0;
    case WBC_T_LAST_ACTIONS: *p = e->last_actions; *n = WBC_NACT; return 0;
    case WBC_T_LAST_DOF_VEL: *p = e->last_dof_vel; *n = WBC_NDOF; return 0;
    case WBC_T_LAST_ROOT_VEL: *p = e->last_root_vel; *n = 6; return 0;
    case WBC_T_COMMANDS: *p = e->commands; *n = 3; return 0;
    case WBC_T_GOAL_STATE: *p = e->goal; *n = 24; return 0;
    case WBC_T_REW_BUF: *p = &e->rew; *n = 1; return 0;
    case WBC_T_ARM_REW_BUF: *p = &e->arm_rew; *n = 1; return 0;
    case WBC_T_EPISODE_SUMS: *p = e->episode_sums; *n = WBC_NREW; return 0;
    case WBC_T_METRIC_SUMS: *p = e->metric_sums; *n = WBC_NMETRIC; return 0;
    case WBC_T_EPISODE_SUMS_DONE: *p = e->episode_sums_done; *n = WBC_NREW; return 0;
    case WBC_T_METRIC_SUMS_DONE: *p = e->metric_sums_done; *n = WBC_NMETRIC; return 0;
    case WBC_T_BASE_LIN_VEL: *p = e->base_lin_vel; *n = 3; return 0;
    case WBC_T_BASE_ANG_VEL: *p = e->base_ang_vel; *n = 3; return 0;
    case WBC_T_MASS_PARAMS: *p = e->mass_params; *n = 5; return 0;
    case WBC_T_FRICTION: *p = &e->friction; *n = 1; return 0;
    case WBC_T_MOTOR_STRENGTH: *p = e->motor_strength; *n = WBC_NACT; return 0;
    case WBC_T_ENV_ORIGINS: *p = e->env_origin; *n = 3; return 0;
    case WBC_T_BOX_DELTA_Y: *p = &e->box_delta_y; *n = 1; return 0;
    case WBC_T_BODY_PARAMS: *p = e->body_params; *n = 20; return 0;
    case WBC_T_RESET_TRAVEL: *p = e->reset_travel; *n = 2; return 0;
    case WBC_T_BOX_MASS: *p = &e->box_mass; *n = 1; return 0;
    case WBC_T_BOX_SLEEP_TIMER: *p = &e->box_timer; *n = 1; return 0;
    case WBC_T_FEET_AIR_TIME: *p = e->feet_air_time; *n = WBC_NFEET; return 0;
    case WBC_T_LAST_CONTACTS: *p = e->last_contacts; *n = WBC_NFEET; return 0;
    case WBC_T_DROPPED_HITS: *p = &e->dropped_hits; *n = 1; return 0;
    default: return -1;
  }
}
ORA_API int ora_field_size(int id) {
  ora_env e; REAL* p; int n;
  if (id == WBC_T_DOF_STATE) return WBC_NDOF * 2;
  if (id == WBC_T_RESET_BUF || id == WBC_T_TIME_OUT_BUF || id == WBC_T_EPISODE_LENGTH) return 1;
  if (field_ptr(&e, id, &p, &n)) return -1;
  return n;
}
ORA_API int ora_get(ora_sim* s, int id, double* out) {
  int sz = ora_field_size(id);
  if (sz < 0) return -1;
  for (int i = 0; i < s->n; ++i) {
    ora_env* e = &s->env[i];
    double* o = out + (size_t)i * sz;
    if (id == WBC_T_DOF_STATE) { for (int j = 0; j < WBC_NDOF; ++j) { o[2 * j] = e->q[j]; o[2 * j + 1] = e->qd[j]; } }
    else if (id == WBC_T_RESET_BUF) o[0] = (double)e->reset_buf;
    else if (id == WBC_T_TIME_OUT_BUF) o[0] = (double)e->time_out;
    else if (id == WBC_T_EPISODE_LENGTH) o[0] = (double)e->episode_length;
    else { REAL* p; int n; field_ptr(e, id, &p, &n); for (int j = 0; j < n; ++j) o[j] = (double)p[j]; }
  }
  return 0;
}
ORA_API int ora_set(ora_sim* s, int id, const double* in) {
  int sz = ora_field_size(id);
  if (sz < 0) return -1;
  for (int i = 0; i < s->n; ++i) {
    ora_env* e = &s->env[i];
    const double* o = in + (size_t)i * sz;
    if (id == WBC_T_DOF_STATE) { for (int j = 0; j < WBC_NDOF; ++j) { e->q[j] = (REAL)o[2 * j]; e->qd[j] = (REAL)o[2 * j + 1]; } }
    else if (id == WBC_T_RESET_BUF) e->reset_buf = (int64_t)o[0];
    else if (id == WBC_T_TIME_OUT_BUF) e->time_out = (uint8_t)o[0];
    else if (id == WBC_T_EPISODE_LENGTH) e->episode_length = (int64_t)o[0];
    else { REAL* p; int n; field_ptr(e, id, &p, &n); for (int j = 0; j < n; ++j) p[j] = (REAL)o[j]; }
  }
  return 0;
}

/* WidowGo1.step over all envs; actions double [N,18] policy order */
ORA_API void ora_step(ora_sim* s, const double* actions) {
  s->step_counter += 1;                                                                   /* WG:876 */
  /* envs are independent (each writes only its own ora_env; draws are counter hashes of (seed, env, step, slot)): the host's
   * cores share the batch. Results do not depend on the thread count. */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < s->n; ++i) {
    REAL a[WBC_NACT];
    for (int j = 0; j < WBC_NACT; ++j) a[j] = (REAL)actions[(size_t)i * WBC_NACT + j];
    env_step(s, &s->env[i], i, a, (uint64_t)s->step_counter);
  }
}
/* reset_idx(all, start=True), BT:129 */
ORA_API void ora_reset_all(ora_sim* s) {
  for (int i = 0; i < s->n; ++i) {
    ora_env* e = &s->env[i];
    REAL rpy[3];
    euler_from_quat(e->root[0] + 3, rpy);
    reset_env(s, e, i, (uint64_t)s->step_counter, 1, rpy[2]);
    update_rigid_body_state(s, e);
  }
}
/* one physics substep with the torques currently stored (gym.simulate, WG:1184) */
ORA_API void ora_simulate(ora_sim* s) {
  for (int i = 0; i < s->n; ++i) physics_substep(s, &s->env[i]);
}
ORA_API void ora_refresh_rigid_body_state(ora_sim* s) {
  for (int i = 0; i < s->n; ++i) update_rigid_body_state(s, &s->env[i]);
}
/* _compute_torques over all envs with the currently stored self.actions */
ORA_API void ora_compute_torques(ora_sim* s) {
  for (int i = 0; i < s->n; ++i) compute_torques(s, &s->env[i]);
}

/* Diagnostics: the contact list of one substep of env i run on a COPY of its state (the env is not advanced):
 * active[NCP], nshare[NCP], lam[NCP][3] (impulses, frame F), n[NCP][3], xc[NCP][3]. */
ORA_API void ora_debug_contacts(ora_sim* s, int i, int* active, int* nshare, double* lam, double* n, double* xc) {
  ora_env copy = s->env[i];
  contact_dump d;
  memset(&d, 0, sizeof(d));
  g_dump = &d;
  physics_substep(s, &copy);
  g_dump = NULL;
  memcpy(active, d.active, sizeof(d.active)); memcpy(nshare, d.nshare, sizeof(d.nshare));
  memcpy(lam, d.lam, sizeof(d.lam)); memcpy(n, d.n, sizeof(d.n)); memcpy(xc, d.xc, sizeof(d.xc));
}

/* Diagnostics for tests/test_oracle_physics.py: joint accelerations and base spatial
 * acceleration (frame F, gravity included) of env i from a contact-free ABA evaluation. */
ORA_API void ora_debug_aba(ora_sim* s, int i, double* qdd_out, double* a0_out, double* K_foot_out) {
  ora_env copy = s->env[i];
  wbc_task_cfg saved = s->cfg;
  s->cfg.contact_margin = -1e30f;   /* no contacts */
  REAL q0[WBC_NDOF], qd0[WBC_NDOF], root0[13];
  memcpy(q0, copy.q, sizeof(q0)); memcpy(qd0, copy.qd, sizeof(qd0)); memcpy(root0, copy.root[0], sizeof(root0));
  /* disable velocity clamps for the finite-difference readout */
  wbc_model savedm = s->model;
  for (int j = 0; j < WBC_NDOF; ++j) s->model.qd_limit[j] = 0;
  physics_substep(s, &copy);
  REAL dt = (REAL)s->cfg.sim_dt;
  for (int j = 0; j < WBC_NDOF; ++j) qdd_out[j] = (double)((copy.qd[j] - qd0[j]) / dt);
  for (int k = 0; k < 6; ++k) a0_out[k] = (double)((copy.root[0][7 + k] - root0[7 + k]) / dt);   /* world: lin, ang */
  (void)K_foot_out;
  s->cfg = saved; s->model = savedm;
}
