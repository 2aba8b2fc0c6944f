// wbc_step_kernel.hip -- the fused widowGo1 rollout step for gfx950 (MI355X).
//
// One 64-lane wavefront per robot (blockDim = 64, gridDim = num_envs rounded up to 8: the XCD-aware env mapping). All per-robot link
// state (frames, joint screws, articulated inertias, inverse inertias, contact rows) lives in
// LDS for the whole policy step; HBM is touched once on entry (state in) and once on exit
// (state, observations, rewards out). Lanes are organised as 5 kinematic chains x 12 lanes
// (4 legs of depth 3, the arm of depth 6): the three ABA passes walk the chains level by
// level with the 6x6 algebra spread over a chain's 12 lanes (3 matrix entries per lane).
//
// What it replaces: WidowGo1.step (reference legged_gym/envs/widowGo1/widowGo1.py:1156-1199)
// including the 4x {_compute_torques WG:1262-1295, gym.simulate WG:1184} decimation loop and
// post_physics_step WG:865-915. The arithmetic mirrors oracle/wbc_oracle.c statement by
// statement (that file is the spec and cites the reference line for every step); comments here
// only describe the lane mapping. All waves of a launch are resident at once (16 robots per CU), so a launch ends with its
// slowest wave: what runs on one lane (the task logic, a reset) sits on the launch's critical path.
#include "wbc_device.h"

// The per-sim constant block is never written while a kernel runs: read through the CONSTANT address space, a wave-uniform load of
// it is a scalar load (s_load into SGPRs) whatever fences surround it; through a generic pointer every load after the first
// wavefront fence is a vector load plus a v_readfirstlane. The tensor table (wbc_step_kernel reads it from memory, the small
// kernels take it by value: TT = its type as the caller holds it) likewise. A tensor's base pointer is a GLOBAL pointer: with
// the address space spelled out its accesses are global_load / global_store with the base in SGPRs and a 32-bit lane offset
// instead of flat accesses on a 64-bit per-lane address. (The host pass of the compiler only parses the device code: plain types.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) DevConst* CP;
typedef __attribute__((address_space(4))) DevTensors DevTensorsK;
template <class X> __device__ __forceinline__ __attribute__((address_space(1))) X* G(X* p) { return (__attribute__((address_space(1))) X*)p; }
#else
typedef const DevConst* CP;
typedef DevTensors DevTensorsK;
template <class X> __device__ __forceinline__ X* G(X* p) { return p; }
#endif

// Row `env` of a per-env tensor: base + env * stride as a wave-uniform 32-bit product on the scalar unit (tensors stay below 2^32
// elements: wbc_sim_create bounds num_envs), so the lane's access is (SGPR base) + (32-bit lane offset).
#define ROW(p, env, stride) (G(p) + (uint32_t)(env) * (uint32_t)(stride))
#define LANES 64
static_assert(LANES == 64, "one wavefront per robot: cross-lane hand-overs rest on wavefront-scope ordering");

// development aid: phase timestamps of block 0 (tools/time_step.py builds a variant with -DWBC_STEP_TIMING); -DWBC_WAVE_TIMING: only
// every wave's own start / end / phase stamps (tools/wave_spread.py), without the phase stamps' pointer checks in the instruction stream
__device__ long long* g_step_dbg = nullptr;
__device__ long long* g_wave_dbg = nullptr;      // timing builds: per env {start, end, reset | HW_ID << 8} of the step kernel's wave
__device__ int g_phase_exit = -1;                // -DWBC_PHASE_EXIT builds: every wave ends at stamp g_phase_exit (tools/phase_counts.py)
#if defined(WBC_PHASE_EXIT)
// instruction counts per phase: a launch whose waves all end at stamp k, before any write-back (the state is left as it was), under
// rocprofv3 --pmc SQ_INSTS_*; the difference between the launches for consecutive stamps is the phase between them
#define STAMP(i) do { if (g_phase_exit == (i)) __builtin_amdgcn_endpgm(); } while (0)
#elif defined(WBC_STEP_TIMING)
#define STAMP(i) do { if (g_step_dbg && blockIdx.x == 0 && threadIdx.x == 0) g_step_dbg[i] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#if defined(WBC_XSTAMPS) || defined(WBC_PHASE_EXIT)
#define XSTAMP(i) STAMP(i)
#else
#define XSTAMP(i) do { } while (0)
#endif
#define CH_LANES 12

// A workgroup is ONE wavefront: its LDS operations execute in program order, so cross-lane hand-over through LDS
// needs no s_waitcnt/s_barrier, only that the compiler keeps the order (wavefront-scope fences emit no code).
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division sequence on the dependent chains
__device__ __forceinline__ float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2], VALU speed, no LDS
__device__ __forceinline__ float pair_swap(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

// LDS float add without a return value (ds_add_f32). One wavefront per workgroup: lanes that hit one address are served in lane order.
__device__ __forceinline__ void lds_add(float* p, float v) {
  // (measured in round 6 with plain stores in their place -- wrong physics, timing only: the contacts' adds cost nothing measurable,
  // 83.0 vs 83.6 us at 1024 envs, 117.7 vs 117.8 at 4096; six lanes adding to one address in EVERY level of pass 2 did: section 2.1)
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Sum over an aligned group of 8 lanes, every lane gets the total: three DPP adds at VALU speed, no LDS round trip.
// ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)), the same bits in all 8 lanes (each step adds the same two operands).
__device__ __forceinline__ float sum8(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, false));   // row_half_mirror
  return x;
}

// Sum over the eight lanes that share (lane & 7), one in every aligned group of 8, every lane gets the total: the other group of the
// 16-lane row by a DPP row rotation, the other rows by the two gfx950 row swaps (v_permlane16_swap / v_permlane32_swap). Vector-ALU
// speed, no LDS, a fixed tree (the same bits in all eight lanes).
__device__ __forceinline__ float sum_groups8(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, true));     // row_ror:8
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(x), false, false);     // {rows 0,0,2,2 | rows 1,1,3,3}
  x = __int_as_float(p[0]) + __int_as_float(p[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(x), false, false);     // {lower half twice | upper half twice}
  return __int_as_float(q[0]) + __int_as_float(q[1]);
}
// x of lane (lane ^ m) within the aligned group of 8, m = 1..7: quad permutations and the half-row mirror (lane -> 7 - lane)
template <int M> __device__ __forceinline__ float xor8(float x) {
  static_assert(M >= 1 && M <= 7, "within a group of 8");
  constexpr int Q = M & 3;
  if (M >= 4) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror: ^ 7
  constexpr int R = M >= 4 ? (Q ^ 3) : Q;          // what is left to apply inside the quad
  if (R == 1) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
  if (R == 2) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
  if (R == 3) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x1B, 0xF, 0xF, true));
  return x;
}

// Inclusive sum along a 16-lane row (DPP row_shr 1, 2, 4, 8 with zero fill): the row's last lane ends up with the row total.
// Fixed order, VALU speed, no LDS.
__device__ __forceinline__ float row_scan16(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xF, 0xF, true));    // row_shr:1
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xF, 0xF, true));    // row_shr:2
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xF, 0xF, true));    // row_shr:4
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xF, 0xF, true));    // row_shr:8
  return x;
}

struct PostBuf {                  // post-physics staging; shares LDS with IA (dead once the substeps are done)
  float out_rb[WBC_NRB_ENV][13];
  float quatB[WBC_NB][4];
  float o76[WBC_NPROP];
};
#define WBC_NREW_WG 22             // the terms WG itself defines (the only ones with a metric side effect)

struct __align__(16) Smem {
  union {
    float IA[WBC_NB][36];        // articulated inertia; reused as K (inverse articulated inertia)
    PostBuf post;
    struct {                     // contact iterations: K of the bodies 1.. is dead once every active contact has built its Delassus
      float K0[36];              // block (the sweeps only apply the root's K), so the per-contact data of the iterations lives there:
      float clam[WBC_NCP][3];    // the impulse (read by the force outputs; the contact point stays in the owning lane's registers)
    } ctc;
  };
  float E[WBC_NB][9];
  float pos[WBC_NB][3];
  float S[WBC_NB][6], v[WBC_NB][6], c[WBC_NB][6], pA[WBC_NB][6], a[WBC_NB][6];
  union {
    float U[WBC_NB][6];
    struct {                     // the step's contact outputs: written after the last substep's contact iterations (U is dead then:
      float out_contact[WBC_NRB_ENV][3];   // integration reads a, c, qdd, qddD only) and read by the post-physics phases
      float out_sensor[WBC_NFEET][6];
    };
  };
  float iD[WBC_NB], u[WBC_NB], qdd[WBC_NB], qddD[WBC_NB];   // iD = 1/D
  float pa1[WBC_NCHAIN][6];      // depth-1 contributions to the root, summed in fixed order
  float q[WBC_NDOF], qd[WBC_NDOF], tau[WBC_NDOF], act[WBC_NACT];
  float root[13], box[13], R[9], wb[3], vb[3], gF[3];
  float bp[20], motor[WBC_NACT];
  // the free box actor in frame F (oracle: box_ws): rotation of the box in the world, then centre / axes / centre velocity / spin in F,
  // 1/m and 1/Ic, and its response to the contact impulses of the current sweep (centre acceleration, angular acceleration)
  float bxRb[9], bxc[3], bxE[9], bxv[3], bxw[3], bxim, bxiI, bxa[6];
  int bxtimer;                   // substeps the box has been at rest (asleep from box_sleep_time / sim_dt on)
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  int dbg_ncon;                  // timing builds: active contacts | deepest active level << 8 of the last substep
#endif
  // contacts (one per lane): what OTHER lanes read -- the contact point and the impulse -- lives in ctc; normal, free velocity and
  // velocity target stay in the owning lane's registers; the set of active contacts is a wavefront ballot.
  // Centres of the robot's contact spheres in frame F, cached once per substep by the terrain lanes: the broad phase of every pair
  // descriptor (static pairs and the self-collision candidates) and the exact tests of the promoted ones read them.
  float sph[WBC_NSPH][3];
  uint32_t k_prk[WBC_NCP];       // DevConst::pr_pack: every lane's pair descriptor
  int dyn_dirty;                 // the per-body contact masks below carry bits of dynamic slots (restored before they are used again)
  int dropped;                   // broad-phase hits of this launch that found no free dynamic slot (WBC_T_DROPPED_HITS)
  int deal_hint;                 // last substep: how heavy the next step of this env is expected to be, 0..2 (the next launch's deal)
  int deal_word, deal_bit;       // where this env's hint bit goes (word index into DevTensors::deal_flags, -1: no dealing; bit 0..63)
  float goal[24], cmd[3], blv[3], bav[3];
  float act_last[WBC_NACT];      // newest (undelayed) action, sim order
  float ep_sums[WBC_NREW], met_sums[WBC_NMETRIC];
  float term[WBC_NREW], msrc[WBC_NREW_WG];   // raw reward terms / metric sources of this step (lane 0 -> lanes t)
  float rew, arm_rew, base_yaw, friction;
  float mu[4];                   // friction coefficients: robot-terrain, robot-robot, box-terrain, robot-box
  float rp[2];                   // base roll / pitch of the state the observation is built from (post-physics, or post-reset)
  float eul[6];                  // Euler angles of the base [0..2] and of the gripper [3..5] after the physics (euler_batch: six lanes, one atan2)
  int reset_flag, time_out, ep_len;
  // per DoF, written by joint_pre_pass as one 16-byte row: {sin q, cos q, joint-limit violation, the joint velocity if it moves further
  // out} (post-physics: {sin q, cos q, sin q/2, cos q/2}) -- the walk and pass 2 read a pair with ONE 8-byte LDS read (a lone wave pays
  // ~10 cycles per LDS instruction, whatever its width)
  alignas(16) float jt[WBC_NDOF][4];
  // model constants the dependent chains index per lane (LDS instead of a global load on the critical path)
  alignas(16) float k_jxyz[WBC_NB][4];   // joint origin in the parent frame (x, y, z, -): one 16-byte read
  uint32_t k_body[WBC_NB];               // DevConst::body_pack
  float k_qdlim[WBC_NDOF];
  float ct_arm[WBC_NCHAIN + 1][WBC_MAX_DEPTH];   // joint armature at (chain, depth); row WBC_NCHAIN is the idle row
  uint32_t k_qpack[3][8];                        // DevConst::chain_pack_body / _dof / _ax by quad of the kinematics walk (rows >= WBC_NCHAIN: idle)
  // DevConst::body_cp_mask / body_cp2_mask of the tree's bodies, contact slots 0..31 and 32..63 apart: the per-body loops walk the
  // two halves separately (32-bit mask arithmetic; the upper half -- mid-shanks, feet / gripper against the box -- is mostly idle).
  // The free box is not in these loops: its contacts fill one 16-lane row and are summed by a row reduction.
  uint2 k_gmlo[WBC_NB], k_gmhi[WBC_NB];           // .x: contacts whose sphere rides on the body, .y: contacts it is the partner of
};

// aliases: pD lives in pA, aD in c (both dead once pass 3 has run); g (pass 3 only) also lives in pA:
// pA is last read by the root solve, g is last read by the final K update, pD is first written by the contacts
#define PD(s) (s).pA
#define AD(s) (s).c
#define GG(s) (s).pA
// TT: one 6-vector of cross-lane terms per body; lives in a (free in pass 2; in pass 3 and the contact sweeps a[i] is
// either about to be written by the same lanes or dead: only a[0] and the contact set-up read it)
#define TT(s) (s).a

// This lane's chain, bit-packed into three registers (DevConst::chain_pack_*): body / dof / axis at each depth.
struct ChainRegs {
  uint32_t body, dof, ax;
  const float* arm;             // LDS row of joint armatures
};
#define CH_NONE 31
// (the unrolled passes decode a level's body / dof from an opaque copy of the packed registers, taken where the level starts: the
// compiler otherwise computes every level's LDS addresses at the top of the pass and holds them in ~30 registers)
__device__ __forceinline__ ChainRegs ch_here(const ChainRegs& cr) {
  ChainRegs c = cr;
  asm volatile("" : "+v"(c.body), "+v"(c.dof));
  return c;
}
__device__ __forceinline__ int ch_body(const ChainRegs& cr, int d) { return (cr.body >> (5 * d)) & 31; }
__device__ __forceinline__ int ch_par(const ChainRegs& cr, int d) { return d ? ((cr.body >> (5 * d - 5)) & 31) : 0; }
__device__ __forceinline__ int ch_dof(const ChainRegs& cr, int d) { return (cr.dof >> (5 * d)) & 31; }
__device__ __forceinline__ int ch_ax(const ChainRegs& cr, int d) { return (cr.ax >> (2 * d)) & 3; }

// value of lane (row + 1) % 3 / (row + 2) % 3 of this lane's quad (lane 3 of a quad reads itself): DPP quad_perm [1,2,0,3] / [2,0,1,3]
// (bound_ctrl set: the compiler folds the permutation into the consuming multiply instead of a v_mov pair)
__device__ __forceinline__ float quad_rot1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xC9, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_rot2(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xD2, 0xF, 0xF, true)); }

// Kinematics of the tree in frame F: ONE walk down every chain with the running frame in registers. A chain is a quad of lanes
// (q = lane >> 2 < 5), lane `row` = lane & 3 < 3 of it holds ROW `row` of the current body's rotation E (3 registers), component
// `row` of its origin and of its spatial velocity. E_i = E_p Rot(axis, q) touches only the lane's own row (a rotation about a
// coordinate axis mixes two columns), the origin needs the parent's row times the joint offset: a level of forward kinematics has no
// cross-lane step at all. The joint screw S = (axis; origin x axis) needs the other two rows' components (DPP rotations within the
// quad), the velocity v_i = v_p + S qd follows in the lane. Every operand of every level (sin / cos, joint offset, joint velocity) is
// requested before the walk starts; everything later phases read goes to LDS as it is produced (E, pos, S, v: the layouts they
// had); nothing is read back. (The velocity-product term c = v x (S qd) is formed with the spatial inertias, one body per lane.) (Rounds 1-5 ran this as four phases -- a 12-lane-per-chain forward
// kinematics with an LDS hand-over per level, then S, v and c as 108-entry maps with a hand-over between them: 6.7 k cycles of a
// wave's dependent chain per substep, 540 vector instructions.)
// POST (the rigid-body pass after the last substep): frames and velocities only. All 64 lanes run the same instructions (the DPP steps
// must not be masked); lanes without a body store nothing.
// one level's rotation with the joint axis known at compile time: no selects
template <int A>
__device__ __forceinline__ float rot_axis(float (&e)[3], float cq, float sq) {
  constexpr int a1 = (A + 1) % 3, a2 = (A + 2) % 3;
  const float n1 = cq * e[a1] + sq * e[a2], n2 = cq * e[a2] - sq * e[a1];
  e[a1] = n1; e[a2] = n2;
  return e[A];
}

template <bool POST>
__device__ __forceinline__ void kin_walk(Smem& s, CP C, int lane) {
  const int q = lane >> 2, row = lane & 3;
  const bool on = q < WBC_NCHAIN && row < 3;
  const uint32_t pb = s.k_qpack[0][q & 7], pd = s.k_qpack[1][q & 7], pa = s.k_qpack[2][q & 7];   // (rows >= WBC_NCHAIN: no bodies)
  const int r3 = row < 3 ? row : 0;
  float e[3] = {row == 0 ? 1.f : 0.f, row == 1 ? 1.f : 0.f, row == 2 ? 1.f : 0.f}, pos = 0.f, w = 0.f, vl = 0.f;
  float sq[WBC_MAX_DEPTH], cq[WBC_MAX_DEPTH], qd[WBC_MAX_DEPTH], jx[WBC_MAX_DEPTH], jy[WBC_MAX_DEPTH], jz[WBC_MAX_DEPTH];
#pragma unroll
  for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
    // (a level without a body: index 31 / dof 0 -- reads that stay inside this workgroup's LDS and feed lanes that store nothing)
    const int i = (pb >> (5 * d)) & 31, dj = (pd >> (5 * d)) & 31;
    const float2 sc = *reinterpret_cast<const float2*>(&s.jt[dj][0]);
    const float4 jo = *reinterpret_cast<const float4*>(&s.k_jxyz[i][0]);
    sq[d] = sc.x; cq[d] = sc.y; qd[d] = s.qd[dj];
    jx[d] = jo.x; jy[d] = jo.y; jz[d] = jo.z;
  }
  {
    // the root's velocity in F: component row of R^T omega, R^T v
    const float r0 = s.R[r3], r1 = s.R[3 + r3], r2 = s.R[6 + r3];
    w = r0 * s.root[10] + r1 * s.root[11] + r2 * s.root[12];
    vl = r0 * s.root[7] + r1 * s.root[8] + r2 * s.root[9];
  }
  if (!POST) XSTAMP(27);
  if (q == 0 && row < 3) {
    s.E[0][3 * row] = e[0]; s.E[0][3 * row + 1] = e[1]; s.E[0][3 * row + 2] = e[2]; s.pos[0][row] = 0.f;
    s.v[0][row] = w; s.v[0][3 + row] = vl;
    if (!POST) { s.S[0][row] = 0.f; s.S[0][3 + row] = 0.f; s.c[0][row] = 0.f; s.c[0][3 + row] = 0.f; }
  }
#pragma unroll
  for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
    const int i = (pb >> (5 * d)) & 31, ax = (pa >> (2 * d)) & 3;
    const bool act = on && i != CH_NONE;
    if (!POST && d == 3) XSTAMP(28);
    // origin: the parent's row times the joint offset
    pos = pos + e[0] * jx[d] + e[1] * jy[d] + e[2] * jz[d];
    // E_i = E_p Rot(ax, q): column ax stays, columns a1 = ax + 1, a2 = ax + 2 (mod 3) mix: new a1 = c a1 + s a2, new a2 = c a2 - s a1.
    // Where every chain that has a body at this level turns about the SAME axis (DevConst::lvl_ax: five of widowGo1's six levels) a
    // scalar branch picks the code for that axis -- register renaming instead of twelve selects on the runtime axis.
    float axw;                                                    // the joint axis in F, component row
    const int la = C->lvl_ax[d];
    if (la == 0) axw = rot_axis<0>(e, cq[d], sq[d]);
    else if (la == 1) axw = rot_axis<1>(e, cq[d], sq[d]);
    else if (la == 2) axw = rot_axis<2>(e, cq[d], sq[d]);
    else {
      const bool ax0 = ax == 0, ax1 = ax == 1;
      const float ea1 = ax0 ? e[1] : (ax1 ? e[2] : e[0]), ea2 = ax0 ? e[2] : (ax1 ? e[0] : e[1]);
      axw = ax0 ? e[0] : (ax1 ? e[1] : e[2]);
      const float n1 = cq[d] * ea1 + sq[d] * ea2, n2 = cq[d] * ea2 - sq[d] * ea1;
      const float f0 = ax0 ? e[0] : (ax1 ? n2 : n1), f1 = ax0 ? n1 : (ax1 ? e[1] : n2), f2 = ax0 ? n2 : (ax1 ? n1 : e[2]);
      e[0] = f0; e[1] = f1; e[2] = f2;
    }
    const float lin = quad_rot1(pos) * quad_rot2(axw) - quad_rot2(pos) * quad_rot1(axw);   // (origin x axis)[row]
    w += axw * qd[d]; vl += lin * qd[d];
    if (act) {
      s.E[i][3 * row] = e[0]; s.E[i][3 * row + 1] = e[1]; s.E[i][3 * row + 2] = e[2]; s.pos[i][row] = pos;
      s.v[i][row] = w; s.v[i][3 + row] = vl;
      if (!POST) { s.S[i][row] = axw; s.S[i][3 + row] = lin; }
    }
  }
}

// symmetric 3x3 (upper triangle 00 01 02 11 12 22) times vector
__device__ __forceinline__ f3 sym_mul(const float* W, f3 v) {
  return mk3(W[0] * v.x + W[1] * v.y + W[2] * v.z, W[1] * v.x + W[3] * v.y + W[4] * v.z, W[2] * v.x + W[4] * v.y + W[5] * v.z);
}
// contact_solve of the oracle, one contact per lane; W = the Delassus block's upper triangle
__device__ __forceinline__ void contact_solve(const float* W, f3 n, float vn_tgt, float mu, f3 vref, float* lam) {
  const float vn = dot(n, vref);
  lam[0] = lam[1] = lam[2] = 0.f;
  if (vn >= vn_tgt) return;
  const f3 Wn = sym_mul(W, n);
  const float nWn = dot(n, Wn);
  const float lam_fl = (vn_tgt - vn) * rcpf(nWn);
  const f3 rhs = n * vn_tgt - vref;
  // symmetric 3x3 solve by cofactors
  const float a = W[0], b = W[1], c = W[2], d = W[3], e = W[4], f = W[5];
  const float c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const float det = a * c00 + b * c01 + c * c02;
  const float c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
  const float id = rcpf(det);
  const f3 st = mk3((c00 * rhs.x + c01 * rhs.y + c02 * rhs.z) * id, (c01 * rhs.x + c11 * rhs.y + c12 * rhs.z) * id,
                    (c02 * rhs.x + c12 * rhs.y + c22 * rhs.z) * id);
  const float ln = dot(n, st);
  const f3 lt = st - n * ln;
  const float ltn = __builtin_amdgcn_sqrtf(dot(lt, lt));
  if (ln > 0.f && ltn <= mu * ln) { lam[0] = st.x; lam[1] = st.y; lam[2] = st.z; return; }
  // sliding: friction mu * (normal impulse) against the tangential velocity of the point (the stick impulse's direction if it has none)
  const f3 vt = vref - n * vn;
  const float vtn = __builtin_amdgcn_sqrtf(dot(vt, vt));
  f3 dir;
  if (vtn > 1e-6f) dir = n - vt * (mu * rcpf(vtn));
  else if (ltn > 1e-12f) dir = n + lt * (mu * rcpf(ltn));
  else { lam[0] = n.x * lam_fl; lam[1] = n.y * lam_fl; lam[2] = n.z * lam_fl; return; }
  const f3 Wd = sym_mul(W, dir);
  const float den = dot(n, Wd);
  if (den <= 0.05f * nWn) { lam[0] = n.x * lam_fl; lam[1] = n.y * lam_fl; lam[2] = n.z * lam_fl; return; }
  const float l = (vn_tgt - vn) * rcpf(den);
  lam[0] = dir.x * l; lam[1] = dir.y * l; lam[2] = dir.z * l;
}

__device__ __forceinline__ void terrain_query(CP C, float x, float y, float* h, f3* n) {
  if (!C->hf) { *h = C->cfg.ground_z; *n = mk3(0.f, 0.f, 1.f); return; }
  const float fx = (x - C->hf_t[0]) / C->hf_hs, fy = (y - C->hf_t[1]) / C->hf_hs;
  long long ix = (long long)fx, iy = (long long)fy;
  ix = ix < 0 ? 0 : ix; iy = iy < 0 ? 0 : iy;
  ix = ix > C->hf_rows - 2 ? C->hf_rows - 2 : ix;
  iy = iy > C->hf_cols - 2 ? C->hf_cols - 2 : iy;
  const float u = clampf(fx - (float)ix, 0.f, 1.f), v = clampf(fy - (float)iy, 0.f, 1.f);
  const int16_t* H = C->hf;
  const int cols = C->hf_cols;
  const float h00 = H[ix * cols + iy] * C->hf_vs, h10 = H[(ix + 1) * cols + iy] * C->hf_vs;
  const float h01 = H[ix * cols + iy + 1] * C->hf_vs, h11 = H[(ix + 1) * cols + iy + 1] * C->hf_vs;
  float dhdx, dhdy;
  if (u >= v) { dhdx = h10 - h00; dhdy = h11 - h10; } else { dhdy = h01 - h00; dhdx = h11 - h01; }
  *h = h00 + u * dhdx + v * dhdy + C->hf_t[2];
  const float gx = dhdx / C->hf_hs, gy = dhdy / C->hf_hs;
  const float inv = 1.f / sqrtf(gx * gx + gy * gy + 1.f);
  *n = mk3(-gx * inv, -gy * inv, inv);
}

// sin/cos of every joint angle and the joint-limit terms of this substep: lanes 32..51, one DoF each.
// HALF (post-physics): the limit slots receive sin/cos of the half angles instead (joint quaternions).
template <bool HALF>
__device__ __forceinline__ void joint_pre_pass(Smem& s, CP C) {
  int j = (int)threadIdx.x;
  asm volatile("" : "+v"(j));      // (laundered: lane - 32 is otherwise one more value held across the substep loop)
  j -= 32;
  if (j >= 0 && j < WBC_NDOF) {
    const float qq = s.q[j], qdv = s.qd[j];
    float sq, cq;
    fast_sincosf(qq, &sq, &cq);
    float2 lim;
    if (HALF) {
      float sh, ch;
      fast_sincosf(0.5f * qq, &sh, &ch);
      *reinterpret_cast<float4*>(&s.jt[j][0]) = make_float4(sq, cq, sh, ch);
      return;
    }
    const float lo = C->model.q_lower[j], hi = C->model.q_upper[j];
    float viol = 0.f;
    if (lo < hi) { if (qq > hi) viol = qq - hi; else if (qq < lo) viol = qq - lo; }
    lim = make_float2(viol, (qdv * viol > 0.f) ? qdv : 0.f);
    *reinterpret_cast<float4*>(&s.jt[j][0]) = make_float4(sq, cq, lim.x, lim.y);
  }
}

// Closest points of two segments (oracle: seg_seg_closest; Ericson 5.1.9): a segment shorter than 1e-5 m counts as a point.
__device__ __forceinline__ void seg_seg_closest(f3 a0, f3 a1, f3 b0, f3 b1, f3* pa, f3* pb) {
  const f3 d1 = a1 - a0, d2 = b1 - b0, r = a0 - b0;
  const float EPS = 1e-10f;
  const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
  const float ia = rcpf(fmaxf(a, EPS)), ie = rcpf(fmaxf(e, EPS));            // (v_rcp_f32: the stage runs on every near limb pair, every substep)
  float sp = 0.f, tp = 0.f;
  if (a <= EPS && e <= EPS) { sp = tp = 0.f; }
  else if (a <= EPS) { tp = clampf(f * ie, 0.f, 1.f); }
  else {
    const float c = dot(d1, r);
    if (e <= EPS) { sp = clampf(-c * ia, 0.f, 1.f); }
    else {
      const float b = dot(d1, d2), den = a * e - b * b;
      sp = den > 1e-7f * a * e ? clampf((b * f - c * e) * rcpf(den), 0.f, 1.f) : 0.f;
      tp = (b * sp + f) * ie;
      if (tp < 0.f) { tp = 0.f; sp = clampf(-c * ia, 0.f, 1.f); }
      else if (tp > 1.f) { tp = 1.f; sp = clampf((b - c) * ia, 0.f, 1.f); }
    }
  }
  *pa = a0 + d1 * sp; *pb = b0 + d2 * tp;
}
// Limb A against limb B (oracle: limb_pair): unions of a capsule and up to two end spheres; the feature pair with the smallest gap wins,
// in the oracle's order (shafts; A's end spheres against B's shaft; B's against A's shaft; end sphere against end sphere; a tie keeps
// the earlier). ck = the candidate's packed descriptor (sphere indices of the four ends), rad = its six radii (DevConst::cand_rad).
// Runs on the few lanes a candidate was promoted into, but a launch ends with its slowest wave: one general segment-segment test,
// then point-segment and point-point tests; the sphere centres are (re)read from LDS where they are used.
struct LimbHit { float gap, ra, rb; f3 n, q; int fa, fb; };
__device__ __forceinline__ f3 closest_on_segment(f3 p, f3 b0, f3 b1) {
  const f3 d = b1 - b0;
  const float e = dot(d, d);
  const float t = e <= 1e-10f ? 0.f : clampf(dot(p - b0, d) * rcpf(e), 0.f, 1.f);
  return b0 + d * t;
}
// The nine feature pairs of ONE limb pair on lanes 0..8 (feature f on lane f: 0 shaft-shaft, 1-2 an end sphere of A against B's shaft,
// 3-4 A's shaft against an end sphere of B, 5-8 end sphere against end sphere), every lane of the wavefront taking part; the result
// (the deepest pair; of equally deep ones the lowest f, as the oracle's ascending scan with a strict comparison finds it) on all lanes.
// (Round 5 ran the pairs as a nine-trip loop on the promoted lane: the ~20 waves per launch that promote were the slowest of the
// launch, 217-226 k cycles against 174 k.)
__device__ __forceinline__ void limb_pair_wave(const Smem& s, uint32_t ck, float r0, float r1, float r2, float r3, float r4, float r5, float rest, int lane, LimbHit& out) {
  const int f = lane;
  const int ea = f == 0 ? -1 : (f <= 2 ? f - 1 : (f <= 4 ? -1 : (f - 5) >> 1));
  const int eb = f <= 2 ? -1 : (f <= 4 ? f - 3 : (f - 5) & 1);
  const float fra = ea < 0 ? r0 : (ea == 0 ? r1 : r2);
  const float frb = eb < 0 ? r3 : (eb == 0 ? r4 : r5);
  const f3 a0 = ld3(s.sph[(ck >> 7) & 31]), a1 = ld3(s.sph[(ck >> 12) & 31]), b0 = ld3(s.sph[(ck >> 17) & 31]), b1 = ld3(s.sph[(ck >> 22) & 31]);
  f3 pa, pb;
  if (f == 0) seg_seg_closest(a0, a1, b0, b1, &pa, &pb);
  else if (ea >= 0) {
    pa = ea == 0 ? a0 : a1;
    pb = eb >= 0 ? (eb == 0 ? b0 : b1) : closest_on_segment(pa, b0, b1);
  } else {
    pb = eb == 0 ? b0 : b1;
    pa = closest_on_segment(pb, a0, a1);
  }
  const f3 dd = pa - pb;
  float g = __builtin_amdgcn_sqrtf(dot(dd, dd)) - fra - frb - rest;
  if (f > 8 || (f != 0 && (!(fra > 0.f) || !(frb > 0.f)))) g = 3.0e38f;      // (no such feature / no such end sphere)
  float m = g;
  m = fminf(m, __shfl_xor(m, 1)); m = fminf(m, __shfl_xor(m, 2)); m = fminf(m, __shfl_xor(m, 4)); m = fminf(m, __shfl_xor(m, 8));
  const int w = __ffsll((unsigned long long)(__ballot(g == m) & 0x1FFull)) - 1;          // (lanes 0..8 hold the minimum over lanes 0..15)
  const f3 bpa = mk3(__shfl(pa.x, w), __shfl(pa.y, w), __shfl(pa.z, w)), bpb = mk3(__shfl(pb.x, w), __shfl(pb.y, w), __shfl(pb.z, w));
  out.fa = __shfl(ea, w) + 1; out.fb = __shfl(eb, w) + 1; out.ra = __shfl(fra, w); out.rb = __shfl(frb, w);
  const f3 d = bpa - bpb;
  const float dist = __builtin_amdgcn_sqrtf(dot(d, d));
  out.n = dist > 1e-9f ? d * rcpf(dist) : mk3(1.f, 0.f, 0.f);
  out.q = bpb + out.n * out.rb;
  out.gap = __shfl(g, w);
}

// One physics substep on the LDS-resident state (oracle: physics_substep).
__device__ void physics_substep(Smem& s, CP C, const ChainRegs& cr, const int chain_, const int k_,
                                const bool want_outputs) {
  // the lane index as a value the optimiser cannot see through: everything derived from it (LDS / constant-table addresses of the
  // one-body / one-DoF / one-contact-per-lane phases) is then recomputed inside each substep (a few integer operations) instead
  // of being computed once before the substep loop and held in ~20 VGPRs across it (the kernel is compiled for 128)
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));
  // (the chain coordinates likewise: re-derived from the laundered lane, so that what depends on them -- the matrix entries a lane
  // owns in the passes, their LDS addresses -- is not held in registers across the substep loop either)
  const int chain = (int)((uint32_t)lane / CH_LANES), k = lane - chain * CH_LANES;
  (void)chain_; (void)k_;
  const float dt = C->cfg.sim_dt;
  const float idt = 1.f / dt;
  // constants of the inertia phase (lane = 3 body + row): issued here, consumed after the kinematics (latency hidden)
  const int ib = (int)((uint32_t)lane / 3u), ir = lane - 3 * ib;
  float bm = 0.f, bcom[3] = {0.f, 0.f, 0.f}, bI6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ib < WBC_NB) {
    bm = C->model.mass[ib];
#pragma unroll
    for (int j = 0; j < 3; ++j) bcom[j] = C->model.com[ib][j];
#pragma unroll
    for (int j = 0; j < 6; ++j) bI6[j] = C->model.inertia[ib][j];
  }
  // root-frame quantities (lane 1, in the same instructions: the rotation of the free box actor)
  if (lane < 2) quat_to_mat(lane == 0 ? &s.root[3] : &s.box[3], lane == 0 ? s.R : s.bxRb);
  joint_pre_pass<false>(s, C);
  WSYNC();
  STAMP(0);
  // angular / linear velocity of the base and gravity in frame F (wb, vb, gF: contiguous), one component per lane
  if (lane >= 20 && lane < 29) {
    const int e = lane - 20, which = e / 3, j = e - 3 * which;
    const int src = which == 0 ? 10 : 7;
    const float x0 = which == 2 ? C->cfg.gravity[0] : s.root[src], x1 = which == 2 ? C->cfg.gravity[1] : s.root[src + 1], x2 = which == 2 ? C->cfg.gravity[2] : s.root[src + 2];
    s.wb[e] = s.R[j] * x0 + s.R[3 + j] * x1 + s.R[6 + j] * x2;
  }
  // the free box in frame F (oracle: box_ws), 18 entries on otherwise idle lanes: out = sum_j R[j][r] x_j with x = a column of the
  // box's rotation (bxE = R^T Rb), its position relative to the base, its velocity, its spin
  if (lane >= 44 && lane < 62) {
    const int e = lane - 44;
    const bool isE = e < 9;
    const int r = isE ? e / 3 : (e - 9) % 3, which = isE ? 0 : (e - 9) / 3;
    const float* X = isE ? &s.bxRb[e % 3] : &s.box[which == 0 ? 0 : (which == 1 ? 7 : 10)];
    const int stp = isE ? 3 : 1;
    const bool rel = !isE && which == 0;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) acc += s.R[3 * j + r] * (X[j * stp] - (rel ? s.root[j] : 0.f));
    float* dst = isE ? &s.bxE[e] : (which == 0 ? &s.bxc[r] : (which == 1 ? &s.bxv[r] : &s.bxw[r]));
    *dst = acc;
  }
  // frames, joint screws, velocities and velocity-product terms: one walk per chain, in registers
  XSTAMP(25);
  kin_walk<false>(s, C, lane);
  XSTAMP(26);
  WSYNC();
  STAMP(1);
  STAMP(2);
  // spatial inertias in frame F, bias forces and velocity-product terms: one ROW of one body per lane (lane = 3 body + row, 57 lanes;
  // one body per lane -- 19 lanes -- cost 279 vector instructions per substep, a tenth of it). The lane works in the cyclically permuted
  // component order (r, r+1, r+2): a cyclic permutation keeps cross products, so it computes "component 0" of every quantity with one
  // instruction stream for all three rows; everything it reads from LDS it reads in that order (rows of E, components of pos, v, S), the
  // body-frame quantities (com, the inertia tensor) are not permuted. What it needs of the other two rows (its third inertia entry --
  // taken from the row that computes the mirrored one, so the matrix is exactly symmetric as before --, the other components of the
  // two intermediate vectors of the bias force) comes through the LDS crossbar (ds_bpermute).
  if (ib < WBC_NB) {
    const int i = ib;
    const int p0 = ir, p1 = ir == 2 ? 0 : ir + 1, p2 = ir == 0 ? 2 : ir - 1;
    float m = bm, com[3], I6[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) com[j] = bcom[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) I6[j] = bI6[j];
    if (i == 0 || i == C->model.gripper_body) {     // randomised per env (body_params)
      const float* bp = &s.bp[i == 0 ? 0 : 10];
      m = bp[0];
#pragma unroll
      for (int j = 0; j < 3; ++j) com[j] = bp[1 + j];
#pragma unroll
      for (int j = 0; j < 6; ++j) I6[j] = bp[4 + j];
    }
    const float* E = s.E[i];
    const f3 E0 = ld3(&E[3 * p0]), E1 = ld3(&E[3 * p1]), E2 = ld3(&E[3 * p2]);       // rows r, r+1, r+2
    const float* P = s.pos[i];
    const float C0 = P[p0] + (E0.x * com[0] + E0.y * com[1] + E0.z * com[2]);
    const float C1 = P[p1] + (E1.x * com[0] + E1.y * com[1] + E1.z * com[2]);
    const float C2 = P[p2] + (E2.x * com[0] + E2.y * com[1] + E2.z * com[2]);
    const float Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    float EI[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) EI[cc] = E0.x * Ib[cc] + E0.y * Ib[3 + cc] + E0.z * Ib[6 + cc];
    // rotational block about F's origin: E Ib E^T + m (|C|^2 1 - C C^T); coupling blocks +-[h]x with h = m C; lower-right m 1
    const float CC = C0 * C0 + C1 * C1 + C2 * C2;
    const float Ir00 = (EI[0] * E0.x + EI[1] * E0.y + EI[2] * E0.z) + m * (CC - C0 * C0);
    const float Ir01 = (EI[0] * E1.x + EI[1] * E1.y + EI[2] * E1.z) + m * (0.f - C0 * C1);
    const int lane1 = 3 * i + p1, lane2 = 3 * i + p2;
    const float Ir02 = __shfl(Ir01, lane2);            // row r+2 computes (r+2, r): the mirror of (r, r+2)
    const float h0 = C0 * m, h1 = C1 * m, h2 = C2 * m;
    (void)h0;
    float* I = s.IA[i];
    float* Ra = &I[p0 * 6];
    float* Rb = &I[(3 + p0) * 6];
    Ra[p0] = Ir00; Ra[p1] = Ir01; Ra[p2] = Ir02;
    Ra[3 + p0] = 0.f; Ra[3 + p1] = -h2; Ra[3 + p2] = h1;
    Rb[p0] = 0.f; Rb[p1] = h2; Rb[p2] = -h1;
    Rb[3 + p0] = m; Rb[3 + p1] = 0.f; Rb[3 + p2] = 0.f;
    // bias force v x* (I v) from the blocks: n = Irot w + h x vl, f = m vl - h x w
    const float* V = s.v[i];
    const float w0 = V[p0], w1 = V[p1], w2 = V[p2], vl0 = V[3 + p0], vl1 = V[3 + p1], vl2 = V[3 + p2];
    const float nn0 = (Ir00 * w0 + Ir01 * w1 + Ir02 * w2) + (h1 * vl2 - h2 * vl1);
    const float ff0 = vl0 * m - (h1 * w2 - h2 * w1);
    const float nn1 = __shfl(nn0, lane1), nn2 = __shfl(nn0, lane2), ff1 = __shfl(ff0, lane1), ff2 = __shfl(ff0, lane2);
    s.pA[i][p0] = (w1 * nn2 - w2 * nn1) + (vl1 * ff2 - vl2 * ff1);
    s.pA[i][3 + p0] = w1 * ff2 - w2 * ff1;
    // velocity-product acceleration c = v x (S qd) = (w x ja; w x jl + vl x ja), (ja; jl) = S qd (the root's is zero: kin_walk)
    if (i > 0) {
      const float qd = s.qd[(s.k_body[i] >> 2) & 31];
      const float* Sb = s.S[i];
      const float ja1 = Sb[p1] * qd, ja2 = Sb[p2] * qd, jl1 = Sb[3 + p1] * qd, jl2 = Sb[3 + p2] * qd;
      s.c[i][p0] = w1 * ja2 - w2 * ja1;
      s.c[i][3 + p0] = (w1 * jl2 - w2 * jl1) + (vl1 * ja2 - vl2 * ja1);
    }
  }
  WSYNC();
  STAMP(3);
  // pass 2, inward: level d+1 of every chain in parallel. Lane (chain,k) owns entries (mr, mc..mc+2) of the chain's
  // articulated inertia and the lane pair (2r,2r+1) component r of its bias force; the children's contributions
  // travel inward in registers (acc3, pacc), the body's own inertia / bias force are read from LDS (s.IA, s.pA, never
  // modified here). One LDS hand-over per level: U and the terms of S.pA out, then D, u and the rank-1 downdate.
  const float kap_dt2 = C->cfg.limit_kappa * idt * idt, del_dt = C->cfg.limit_delta * idt;
  const int mr = k >> 1, mc = (k & 1) * 3;
  {
    // The operands of a level that do not depend on the recursion (the body's own inertia entries, S, its bias force, torque, limit
    // terms, armature, c) are requested one level AHEAD, right after the previous level's hand-over: the pass never writes them, and
    // their LDS latency runs under that level's arithmetic instead of heading this one's dependent chain (levels unrolled: the
    // operand sets are plain registers).
    // Lanes whose chain has no body at a level (the legs below level 3, the four spare lanes) run the same arithmetic on whatever
    // their reads return (a "none" body index stays inside this workgroup's LDS): only the stores are masked.
    struct Lvl { float ia[3], s3[3], smr, pa; int i, dj; };
    auto fetch = [&](int d, Lvl& o) {
      const ChainRegs cl = ch_here(cr);
      o.i = ch_body(cl, d); o.dj = ch_dof(cl, d);
#pragma unroll
      for (int j = 0; j < 3; ++j) { o.ia[j] = s.IA[o.i][3 * k + j]; o.s3[j] = s.S[o.i][mc + j]; }
      o.smr = s.S[o.i][mr]; o.pa = s.pA[o.i][mr];
    };
    float acc3[3] = {0.f, 0.f, 0.f}, pacc = 0.f;
    Lvl cur, nxt;
    fetch(WBC_MAX_DEPTH - 1, cur);
#pragma unroll
    for (int d = WBC_MAX_DEPTH - 1; d >= 0; --d) {
      const int i = cur.i, dj = cur.dj;
      const bool act = i != CH_NONE;
      // what the second half of the level needs, requested at its start
      const float c30 = s.c[i][mc], c31 = s.c[i][mc + 1], c32 = s.c[i][mc + 2];
      const float2 jl = *reinterpret_cast<const float2*>(&s.jt[dj][2]);
      const float jtau = s.tau[dj], jviol = jl.x, jlimd = jl.y, jarm = cr.arm[d];
      float IA3[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) IA3[j] = cur.ia[j] + acc3[j];
      const float pAr = cur.pa + pacc;
      const float up = IA3[0] * cur.s3[0] + IA3[1] * cur.s3[1] + IA3[2] * cur.s3[2];
      const float Ur = up + pair_swap(up);          // U[mr] = row mr of IA times S
      if (act) {
        s.U[i][mr] = Ur;                            // both lanes of the pair write the same value
        TT(s)[i][mr] = cur.smr * pAr;
      }
      WSYNC();
      if (d > 0) fetch(d - 1, nxt);
      {
        float U3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) U3[j] = s.U[i][mc + j];
        // (the six S.pA terms: summed by every lane. Tried in round 6: one LDS float add per lane pair into u[i] and one read -- fewer
        // vector instructions, but six lanes adding to ONE address serialise in the LDS: 122 -> 135 us per launch)
        const float* t = TT(s)[i];
        const float tsum = ((((t[0] + t[1]) + t[2]) + t[3]) + t[4]) + t[5];
        const float dp = cur.s3[0] * U3[0] + cur.s3[1] * U3[1] + cur.s3[2] * U3[2];
        const float D = (dp + pair_swap(dp)) + jarm;
        // joint-limit stop (scaled by D): -kappa D/dt^2 viol - delta D/dt qd if moving further out
        const float tau = jtau - D * (kap_dt2 * jviol + del_dt * jlimd);
        const float u = tau - tsum;
        const float invD = rcpf(D);
        if (act && k == 0) { s.iD[i] = invD; s.u[i] = u; }
        float Ia3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) Ia3[j] = IA3[j] - Ur * U3[j] * invD;     // (U_r U_c)/D: stays exactly symmetric
        const float pp = Ia3[0] * c30 + Ia3[1] * c31 + Ia3[2] * c32;
        const float pa = pAr + (pp + pair_swap(pp)) + Ur * u * invD;         // component mr of pA + Ia c + U u/D
        if (d > 0) {
#pragma unroll
          for (int j = 0; j < 3; ++j) acc3[j] = act ? Ia3[j] : 0.f;
          pacc = act ? pa : 0.f;
        } else if (act) {
#pragma unroll
          for (int j = 0; j < 3; ++j) s.IA[i][3 * k + j] = Ia3[j];           // depth-1 contribution to the root
          s.pa1[chain][mr] = pa;
        }
      }
      cur = nxt;
    }
  }
  WSYNC();
  STAMP(4);
  // root: sum the five depth-1 contributions in fixed chain order
  if (lane < 36) {
    float acc = s.IA[0][lane];
#pragma unroll
    for (int ch = 0; ch < WBC_NCHAIN; ++ch) acc += s.IA[C->chain_body[ch][0]][lane];
    s.IA[0][lane] = acc;
  } else if (lane < 42) {
    const int r = lane - 36;
    float acc = s.pA[0][r];
#pragma unroll
    for (int ch = 0; ch < WBC_NCHAIN; ++ch) acc += s.pa1[ch][r];
    s.pA[0][r] = acc;
  }
  WSYNC();
  // K0 = IA0^-1 by six Gauss-Jordan sweeps, one matrix entry per lane. The matrix stays in registers: the pivot is a
  // v_readlane, the pivot row / column entries come through the LDS crossbar (ds_bpermute: no memory, no hand-over); one
  // store at the end. (Twelve LDS round trips before: 2.5 k of a substep's 48 k cycles.)
  if (lane < 36) {
    const int r = lane / 6, cc = lane % 6;
    float own = s.IA[0][lane];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      const float piv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own), kk * 7));
      const float rowv = __int_as_float(__builtin_amdgcn_ds_bpermute((kk * 6 + cc) * 4, __float_as_int(own)));
      const float colv = __int_as_float(__builtin_amdgcn_ds_bpermute((r * 6 + kk) * 4, __float_as_int(own)));
      const float id = rcpf(piv);
      const float on_row = (cc == kk) ? id : rowv * id;
      const float off_row = (cc == kk) ? -colv * id : own - colv * rowv * id;
      own = (r == kk) ? on_row : off_row;
    }
    s.IA[0][lane] = own;
  }
  WSYNC();
  if (lane < 6) s.a[0][lane] = -dot6(&s.IA[0][lane * 6], s.pA[0]);
  WSYNC();
  STAMP(5);
  // contact-sphere constants of this lane: issued before pass 3, consumed after it
  // (a laundered copy of C for the lane-indexed constant loads of the contact phase: left alone, the compiler computes their
  // 64-bit addresses once before the substep loop and keeps ~10 of them alive in VGPR pairs across it -- spilled to scratch)
  CP Cc = C;
  asm volatile("" : "+s"(Cc));
  int cpb = Cc->model.cp_body[lane], cpkind = Cc->model.cp_kind[lane];        // (WBC_NCP = 64: every lane has a slot)
  float cpp[3], cpr = Cc->model.cp_radius[lane];
#pragma unroll
  for (int j = 0; j < 3; ++j) cpp[j] = Cc->model.cp_pos[lane][j];
  int cpb2 = -1;                                                         // the partner of a pair; -1 for terrain contacts
  if (cpkind == WBC_CP_BOX) cpb2 = Cc->model.cp_body2[lane];
  // pass 3 and inverse articulated inertias, outward: the parent's K entries (K3) and acceleration component (apr)
  // travel down the chain in registers; one LDS hand-over per level (g = K_p U / D and the terms of U.(a_p + c)).
  {
    // (operands requested one level ahead, as in pass 2: U, 1/D, u, S and c of the next body are not written by this pass)
    struct Lvl { float u3[3], invD, cmr, umr; int i; };
    auto fetch = [&](int d, Lvl& o) {
      o.i = ch_body(ch_here(cr), d);
#pragma unroll
      for (int j = 0; j < 3; ++j) o.u3[j] = s.U[o.i][mc + j];
      o.invD = s.iD[o.i]; o.cmr = s.c[o.i][mr]; o.umr = s.U[o.i][mr];
    };
    float K3[3], apr = s.a[0][mr];
#pragma unroll
    for (int j = 0; j < 3; ++j) K3[j] = s.IA[0][3 * k + j];
    Lvl cur, nxt;
    fetch(0, cur);
#pragma unroll
    for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
      const int i = cur.i;
      const bool act = i != CH_NONE;
      // (lanes without a body at this level: the same arithmetic on whatever they read, only the stores are masked)
      const float s30 = s.S[i][mc], s31 = s.S[i][mc + 1], s32 = s.S[i][mc + 2], sr = s.S[i][mr], ui = s.u[i];
      const float gp = K3[0] * cur.u3[0] + K3[1] * cur.u3[1] + K3[2] * cur.u3[2];
      const float gr = (gp + pair_swap(gp)) * cur.invD;       // g[mr] = row mr of K_p times U / D
      const float apc = apr + cur.cmr;
      if (act) {
        GG(s)[i][mr] = gr;
        TT(s)[i][mr] = cur.umr * apc;
      }
      WSYNC();
      if (d + 1 < WBC_MAX_DEPTH) fetch(d + 1, nxt);
      {
        float g3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) g3[j] = GG(s)[i][mc + j];
        const float* t = TT(s)[i];
        const float qdd = (ui - (((((t[0] + t[1]) + t[2]) + t[3]) + t[4]) + t[5])) * cur.invD;
        const float ug = cur.u3[0] * g3[0] + cur.u3[1] * g3[1] + cur.u3[2] * g3[2];
        const float gam = (ug + pair_swap(ug)) * cur.invD + cur.invD;
        const float s3[3] = {s30, s31, s32};
#pragma unroll
        for (int j = 0; j < 3; ++j) K3[j] = K3[j] - gr * s3[j] - sr * g3[j] + gam * sr * s3[j];
        apr = apc + sr * qdd;
        if (act) {
#pragma unroll
          for (int j = 0; j < 3; ++j) s.IA[i][3 * k + j] = K3[j];
          s.a[i][mr] = apr;                         // after every lane's read of t (program order within the wavefront)
          if (k == 0) s.qdd[i] = qdd;
        }
      }
      cur = nxt;
    }
  }
  WSYNC();
  STAMP(6);
  // contacts: one contact per lane. (1) spheres against the terrain (the robot's, the free box's corners); the robot spheres' centres
  // are cached. (2) broad phase: EVERY lane tests its pair descriptor against bounding spheres in the same instructions -- a static
  // pair lane its own pair (arm vs trunk box, robot spheres vs the free box), every other lane a candidate of the self-collision set
  // (limb vs limb; more robot spheres vs the free box). (3) candidates that pass are promoted into free dynamic slots (rare). (4) exact
  // tests on the lanes that need them. Only lanes with an ACTIVE contact build a Delassus block.
  bool onbox = cpb == WBC_BOX_BODY;
  f3 cn = mk3(0.f, 0.f, 1.f), cxcr = mk3(0.f, 0.f, 0.f);
  float cgap = 1e30f;
  const uint32_t prk = s.k_prk[lane];                                   // this lane's pair descriptor (DevConst::pr_pack)
  if (cpkind == WBC_CP_TERRAIN) {
    const float* Eb = onbox ? s.bxE : s.E[onbox ? 0 : cpb];
    const float* pb = onbox ? s.bxc : s.pos[onbox ? 0 : cpb];
    const f3 xk = ld3(pb) + mat_mul(Eb, mk3(cpp[0], cpp[1], cpp[2]));
    const int own = (prk >> 2) & 31;
    if (own < WBC_NSPH) st3(s.sph[own], xk);
    if (!C->hf) {
      // the ground plane (a wave-uniform branch): only the height of the centre is needed, the normal is the world's z axis in F
      cn = mk3(s.R[6], s.R[7], s.R[8]);
      cgap = ((s.root[2] + dot(cn, xk)) - C->cfg.ground_z) - cpr;
    } else {
      const f3 Xw = ld3(&s.root[0]) + mat_mul(s.R, xk);
      float h; f3 nw;
      terrain_query(C, Xw.x, Xw.y, &h, &nw);
      cgap = (Xw.z - h) * nw.z - cpr;
      cn = matT_mul(s.R, nw);
    }
    cxcr = xk - cn * cpr;
  }
  WSYNC();
  // (2) broad phase: |centre A - centre B|^2 < (reach + margin + 1 mm)^2, centres = midpoints of two cached spheres (a sphere: twice the
  // same), the static pair's box centre (a box on the root body: frame F) or the free box's centre. A pair it rejects has a gap far above
  // the contact margin, so the exact test could not have found it active either.
  bool near = false;
  {
    const int pk = prk & 3;
    const f3 a0 = ld3(s.sph[(prk >> 7) & 31]), a1 = ld3(s.sph[(prk >> 12) & 31]), b0 = ld3(s.sph[(prk >> 17) & 31]), b1 = ld3(s.sph[(prk >> 22) & 31]);
    const f3 ca = (a0 + a1) * 0.5f;
    const f3 cbl = (b0 + b1) * 0.5f;
    const int bsel = (prk >> 27) & 3;
    const f3 cb = bsel == 0 ? cbl : (bsel == 1 ? mk3(C->trunk_c[0], C->trunk_c[1], C->trunk_c[2]) : ld3(s.bxc));
    const float reach = (float)((prk >> 29) + 1u) * WBC_REACH_STEP + C->cfg.contact_margin + 1e-3f;
    const f3 dc = ca - cb;
    const float dcc = dot(dc, dc);
    near = pk != WBC_PR_NONE && dcc < reach * reach;
    // second stage for the limb pairs that pass (two legs standing side by side always do): the two shafts' segments against a generous
    // common radius -- one segment-segment distance on those lanes instead of a promotion and nine feature pairs every substep. Before it,
    // in a dozen instructions: the separation of the two segments ALONG the line of their midpoints, |dc| - (|dc.hA| + |dc.hB|) / |dc|
    // with hA, hB the half segments, is a lower bound of their distance; parallel legs side by side (dc across the shafts) fail it at
    // once, and the segment-segment code runs only for the pairs it cannot exclude (it would have excluded the same ones: same radius).
    bool limbs = near && pk == WBC_PR_LIMBS;
    if (__ballot(limbs) != 0ull) {
      const float r2 = WBC_LIMB_RSUM_MAX + Cc->model.pair_rest_offset + C->cfg.contact_margin + 1e-3f;
      if (limbs) {
        const float sepn = dcc - (fabsf(dot(dc, a1 - ca)) + fabsf(dot(dc, b1 - cbl)));      // |dc| x the separation
        if (sepn > 0.f && sepn * sepn > 1.002f * (r2 * r2) * dcc) { limbs = false; near = false; }
      }
      if (__ballot(limbs) != 0ull && limbs) {
        f3 pa, pb;
        seg_seg_closest(a0, a1, b0, b1, &pa, &pb);
        const f3 dd = pa - pb;
        near = dot(dd, dd) < r2 * r2;
      }
    }
  }
  const uint64_t nearbits = __ballot(near);
  // (3) promotion: hits of the candidate lanes into the free dynamic slots, ascending lane -> ascending slot (robot-vs-robot hits outside
  // the box row, robot-vs-box hits inside it); hits beyond the free slots are dropped. Wave-uniform loops over the few set bits.
  int cand = -1;
  {
    uint64_t h = nearbits & Cc->cand_self_mask, pool = Cc->dyn_self_mask;
    while (h != 0ull && pool != 0ull) {
      const int c = __ffsll((unsigned long long)h) - 1, d = __ffsll((unsigned long long)pool) - 1;
      h &= h - 1; pool &= pool - 1;
      cand = (lane == d) ? c : cand;
    }
    int ndrop = __popcll(h);
    h = nearbits & Cc->cand_box_mask; pool = Cc->dyn_box_mask;
    while (h != 0ull && pool != 0ull) {
      const int c = __ffsll((unsigned long long)h) - 1, d = __ffsll((unsigned long long)pool) - 1;
      h &= h - 1; pool &= pool - 1;
      cand = (lane == d) ? c : cand;
    }
    ndrop += __popcll(h);
    if (ndrop != 0 && lane == 0) s.dropped += ndrop;                     // (scalar test: hits that found no free slot -- WBC_T_DROPPED_HITS)
  }
  // (a wave that promotes has the exact limb tests ahead of it -- the longer way to go: it takes the SIMD's issue slots first, as a
  // wave in contact does)
  if ((nearbits & (Cc->cand_self_mask | Cc->cand_box_mask)) != 0ull) __builtin_amdgcn_s_setprio(1);
  // (4) exact tests: the static pair lanes that passed, and the promoted candidates (their slot takes the pair's bodies, rigid bodies and
  // feature radii: drb / drb2 / drad2 are only meaningful on a promoted lane)
  int dpk = 0;                    // promoted lane: rigid body | partner's rigid body << 8 | partner's feature (0 shaft, 1 / 2 end sphere) << 16 | the candidate << 18
  const bool do_static = cpkind == WBC_CP_BOX && near;
  if (__ballot(do_static || cand >= 0) != 0ull) {
    bool sbox = do_static;
    int si = (prk >> 7) & 31;                                            // the sphere of a sphere-vs-box test
    // promoted limb pairs (wave-uniform loop over the few of them): the pair's nine feature pairs on nine lanes, the result to its lane
    LimbHit hit;
    hit.gap = 0.f; hit.ra = hit.rb = 0.f; hit.n = hit.q = mk3(0.f, 0.f, 0.f); hit.fa = hit.fb = 0;
    {
      uint64_t lb = __ballot(cand >= 0 && (s.k_prk[cand < 0 ? 0 : cand] & 3) == WBC_PR_LIMBS);
      while (lb != 0ull) {
        const int L = __ffsll((unsigned long long)lb) - 1;
        lb &= lb - 1;
        const int cL = __builtin_amdgcn_readlane(cand, L);
        const uint32_t ckL = s.k_prk[cL];
        const float* rl = Cc->cand_rad[cL];
        LimbHit h;
        limb_pair_wave(s, ckL, rl[0], rl[1], rl[2], rl[3], rl[4], rl[5], Cc->model.pair_rest_offset, lane, h);
        if (lane == L) hit = h;
      }
    }
    if (cand >= 0) {
      // everything the promoted pair needs in one round of independent loads: its descriptor (LDS), radii, rigid bodies, moving bodies
      const uint32_t ck = s.k_prk[cand], crb = Cc->cand_rbs[cand], cbd = Cc->cand_bodies[cand];
      float rad[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) rad[j] = Cc->cand_rad[cand][j];
      cpb = cbd & 255; cpb2 = (cbd >> 8) & 255;
      if ((ck & 3) == WBC_PR_LIMBS) {
        cgap = hit.gap; cn = hit.n; cxcr = hit.q;
        cpkind = WBC_CP_LIMBS;
        cpr = hit.ra;
        const int drb = (crb >> (5 * hit.fa)) & 31, drb2 = (crb >> (15 + 5 * hit.fb)) & 31;
        dpk = drb | drb2 << 8 | hit.fb << 16 | cand << 18;
      } else {                                                           // a robot sphere against the free box
        si = (ck >> 7) & 31;
        cpkind = WBC_CP_BOX;
        cpr = rad[0];
        dpk = (crb & 31) | WBC_BOX_RB << 8;
        sbox = true;
      }
    }
    if (sbox) {                                                          // sphere against a box: centre A, half extents B, the box's frame
      const bool fb = cpb2 == WBC_BOX_BODY;
      const float* E2 = fb ? s.bxE : s.E[fb ? 0 : cpb2];
      const float* p2 = fb ? s.bxc : s.pos[fb ? 0 : cpb2];
      const f3 pl = matT_mul(E2, ld3(s.sph[si]) - ld3(p2));               // sphere centre in the box's frame
      const float bh = Cc->model.box_half;                                // (the trunk box, or the free box: centre 0, half edge)
      const f3 A = fb ? mk3(0.f, 0.f, 0.f) : mk3(C->trunk_c[0], C->trunk_c[1], C->trunk_c[2]);
      const f3 B = fb ? mk3(bh, bh, bh) : mk3(C->trunk_h[0], C->trunk_h[1], C->trunk_h[2]);
      f3 ql, nl;
      float dist;
      const f3 r = pl - A;
      const f3 cl = mk3(clampf(r.x, -B.x, B.x), clampf(r.y, -B.y, B.y), clampf(r.z, -B.z, B.z));
      const bool inside = cl.x == r.x && cl.y == r.y && cl.z == r.z;
      if (!inside) {
        ql = A + cl;
        nl = r - cl;
        dist = __builtin_amdgcn_sqrtf(dot(nl, nl));
        nl = nl * (1.f / dist);
      } else {                                                           // leave through the nearest face
        const float dx = B.x - fabsf(r.x), dy = B.y - fabsf(r.y), dz = B.z - fabsf(r.z);
        int ax = 0; float best = dx;
        if (dy < best) { best = dy; ax = 1; }
        if (dz < best) { best = dz; ax = 2; }
        const float rv = ax == 0 ? r.x : (ax == 1 ? r.y : r.z), sg = rv >= 0.f ? 1.f : -1.f;
        nl = mk3(ax == 0 ? sg : 0.f, ax == 1 ? sg : 0.f, ax == 2 ? sg : 0.f);
        ql = A + mk3(ax == 0 ? sg * B.x : cl.x, ax == 1 ? sg * B.y : cl.y, ax == 2 ? sg * B.z : cl.z);
        dist = -best;
      }
      cgap = dist - cpr;
      cn = mat_mul(E2, nl);
      cxcr = ld3(p2) + mat_mul(E2, ql);                                  // on the box's surface
    }
  }
  const bool p2box = cpb2 == WBC_BOX_BODY;
  STAMP(18);
  bool cact = cgap < C->cfg.contact_margin;
  uint64_t abits = __ballot(cact);                                      // the active set (WBC_NCP <= 64 lanes)
  // sleeping box (oracle: box_asleep): at rest for box_sleep_time, on at least three corners, untouched -> frozen this substep
  bool asleep = false;
  {
    const float vs = Cc->model.box_sleep_speed;
    if (vs > 0.f) {
      const uint64_t bx_corner_mask = Cc->box_corner_mask, bx_pair_mask = Cc->box_pair_mask;
      const int bx_nsleep = (int)(Cc->model.box_sleep_time * idt + 0.5f);                      // substeps at rest before it sleeps
      // slow enough? (every lane evaluates the same broadcast reads of the box's velocities: no hand-over; kept as a scalar)
      const float bh = Cc->model.box_half;
      const f3 bv = ld3(&s.box[7]), bw = ld3(&s.box[10]);
      const int bxslow = __builtin_amdgcn_readfirstlane((int)(dot(bv, bv) < vs * vs && dot(bw, bw) * (bh * bh) < vs * vs));
      const bool resting = bxslow && __popcll(abits & bx_corner_mask) >= 3 && (abits & bx_pair_mask) == 0ull;
      const int boxtimer = __builtin_amdgcn_readfirstlane(s.bxtimer);      // (a register carried across the substeps spills the kernel)
      asleep = resting && boxtimer >= bx_nsleep;
      if (lane == 0) s.bxtimer = resting ? min(boxtimer + 1, bx_nsleep) : 0;
      if (asleep) { abits &= ~bx_corner_mask; cact = cact && !onbox; }
    }
  }
  const uint32_t ablo = (uint32_t)abits, abhi = (uint32_t)(abits >> 32);
  // Dynamic slots in contact (rare): the per-body contact masks of their two bodies (the gather loops, the relaxation counts) take
  // the slots' bits for this substep -- restored from the constant block first whenever an earlier substep left some behind -- and
  // the sweeps must reach their bodies' levels.
  const bool dyn_act = cact && cand >= 0;
  const uint64_t dynbits = __ballot(dyn_act);
  int ddmax = 0;
  if (dynbits != 0ull || __builtin_amdgcn_readfirstlane(s.dyn_dirty) != 0) {
    if (lane < WBC_NB) {
      const uint64_t m1 = C->body_cp_mask[lane], m2 = C->body_cp2_mask[lane];
      s.k_gmlo[lane] = make_uint2((uint32_t)m1, (uint32_t)m2); s.k_gmhi[lane] = make_uint2((uint32_t)(m1 >> 32), (uint32_t)(m2 >> 32));
    }
    if (lane == 0) s.dyn_dirty = dynbits != 0ull;
    WSYNC();
    int dd = 0;
    if (dyn_act) {
      const uint32_t bit = 1u << (lane & 31);
      uint2* gm = lane < 32 ? s.k_gmlo : s.k_gmhi;
      atomicOr(&gm[cpb].x, bit);                                          // the sphere-side body
      dd = Cc->body_depth[cpb];
      if (!p2box) { atomicOr(&gm[cpb2].y, bit); dd = max(dd, Cc->body_depth[cpb2]); }
    }
#pragma unroll
    for (int d = 2; d <= WBC_MAX_DEPTH; ++d) ddmax = (__ballot(dd >= d) != 0ull) ? d : ddmax;
    WSYNC();
  }
  // friction coefficient of this contact: robot-terrain, box-terrain, robot-robot, robot-box
  const float cmu = s.mu[cpkind == WBC_CP_TERRAIN ? (onbox ? 2 : 0) : (p2box ? 3 : 1)];
  f3 cvfree = mk3(0.f, 0.f, 0.f), clamr = mk3(0.f, 0.f, 0.f);
  float cvtgt = 0.f;
  float cW[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cact) {
    cvtgt = (cgap >= 0.f) ? -cgap * idt : fminf(C->cfg.contact_erp * (-cgap) * idt, C->cfg.max_depenetration_vel);
    const f3 xc = cxcr;
    // W = J K J^T with J = [-[xc]x  I] (point velocity = v + omega x xc): for any 6-vector (a; l), J-row products are
    // l + a x xc, so K J^T has rows Kl_r + Ka_r x xc and W's columns are L_c + U_c x xc (54 FMAs instead of 162). A
    // pair sums the blocks of its two bodies (their coupling through the tree is left to the sweeps). The free box contributes the
    // closed form of a rigid body with isotropic inertia about its centre: 1/m + (|r|^2 1 - r r^T) / Ic, r = xc - centre.
    const int nside = cpkind == WBC_CP_TERRAIN ? 1 : 2;
#pragma unroll 1
    for (int side = 0; side < nside; ++side) {
      const int b = side == 0 ? cpb : cpb2;
      f3 vf;
      if (b == WBC_BOX_BODY) {
        const f3 r = xc - ld3(s.bxc);
        const float rr = dot(r, r), im = s.bxim, iI = s.bxiI;
        cW[0] += (im + rr * iI) - r.x * r.x * iI; cW[1] -= r.x * r.y * iI; cW[2] -= r.x * r.z * iI;
        cW[3] += (im + rr * iI) - r.y * r.y * iI; cW[4] -= r.y * r.z * iI;
        cW[5] += (im + rr * iI) - r.z * r.z * iI;
        // free motion: the centre falls with gravity, the spin is constant (isotropic inertia)
        const f3 w = ld3(s.bxw);
        const f3 wr = cross(w, r);
        vf = (ld3(s.bxv) + wr) + (ld3(s.gF) + cross(w, wr)) * dt;
      } else {
        const float* K = s.IA[b];
        // rows of K J^T: kj_r = Kl_r + Ka_r x xc. W row r' = kj_{3+r'} + (column-wise) (kj_0, kj_1, kj_2) x xc, i.e.
        // W0 = kj3 + kj1 xc.z - kj2 xc.y, W1 = kj4 + kj2 xc.x - kj0 xc.z, W2 = kj5 + kj0 xc.y - kj1 xc.x; the upper triangle is kept
        const f3 k0 = ld3(&K[3]) + cross(ld3(&K[0]), xc), k1 = ld3(&K[9]) + cross(ld3(&K[6]), xc), k2 = ld3(&K[15]) + cross(ld3(&K[12]), xc);
        {
          const f3 k3 = ld3(&K[21]) + cross(ld3(&K[18]), xc);
          const f3 w0 = k3 + (k1 * xc.z - k2 * xc.y);
          cW[0] += w0.x; cW[1] += w0.y; cW[2] += w0.z;
        }
        {
          const f3 k4 = ld3(&K[27]) + cross(ld3(&K[24]), xc);
          const f3 w1 = k4 + (k2 * xc.x - k0 * xc.z);
          cW[3] += w1.y; cW[4] += w1.z;
        }
        {
          const f3 k5 = ld3(&K[33]) + cross(ld3(&K[30]), xc);
          const f3 w2 = k5 + (k0 * xc.y - k1 * xc.x);
          cW[5] += w2.z;
        }
        const f3 w = ld3(&s.v[b][0]);
        const f3 vp = ld3(&s.v[b][3]) + cross(w, xc);
        const f3 ab_a = ld3(&s.a[b][0]);
        const f3 ab_l = ld3(&s.a[b][3]) + ld3(s.gF);
        const f3 apnt = ab_l + cross(ab_a, xc) + cross(w, vp);
        vf = vp + apnt * dt;
      }
      cvfree = side == 0 ? vf : cvfree - vf;
    }
    cW[0] += 1e-6f; cW[3] += 1e-6f; cW[5] += 1e-6f;
  }
  // every lane's reads of K are done before the blocks go where K_1.. was (the two-sided loop above has a divergent trip count:
  // the barrier makes the order explicit instead of resting on reconvergence)
  WSYNC();
  if (cact) {
    s.ctc.clam[lane][0] = s.ctc.clam[lane][1] = s.ctc.clam[lane][2] = 0.f;
  }
  if (lane < WBC_NB) s.qddD[lane] = 0.f;
  if (lane < 6) { AD(s)[0][lane] = 0.f; s.bxa[lane] = 0.f; }
  // the per-body contact wrenches pD (every active contact ADDS its own; zeroed again by the sweep that consumed them)
  for (int t = lane; t < WBC_NB * 6; t += LANES) (&PD(s)[0][0])[t] = 0.f;
  WSYNC();
  STAMP(7);
  // dmax: deepest chain level that carries an active contact. Deeper levels see no contact wrench, so the inward
  // sweep skips them exactly; the outward sweep needs them only in the last iteration (joint accelerations).
  const int any = abits != 0ull;
  // A launch ends with its slowest wave, and a wave in contact has the longer way to go (the solver's sweeps: +17 k cycles per
  // substep over an airborne robot's): it takes the SIMD's issue slots first from here on (s_setprio; nothing changes while all
  // four waves of a SIMD are in contact).
  if (any) __builtin_amdgcn_s_setprio(1);
  // contacts that act on the tree (everything but the box's own corners): without one -- the robot in the air while the box lies
  // on the ground -- the tree's response is zero and its sweeps are skipped
  const bool any_tree = (abits & ~Cc->box_corner_mask) != 0ull;
  if (want_outputs) {
    // (wave-uniform) what the next launch's deal goes by. 1: in contact now, or a robot sphere that reaches the contact margin within
    // a step at the base's rate of descent; 2: five or more contacts on the tree (lying on the ground), or limbs touching each other
    // (the exact limb tests: the longest way a wave can go)
    const float soon_gap = C->cfg.contact_margin + 0.02f + fmaxf(-s.root[9], 0.f) * (6.f * dt);      // (root[9]: the base's vertical velocity)
    const uint64_t soon = __ballot(cpkind == WBC_CP_TERRAIN && !onbox && cgap < soon_gap);
    const uint64_t tree = abits & ~Cc->box_corner_mask;
    if (lane == 0) s.deal_hint = (__popcll(tree) >= 5 || (abits & Cc->dyn_self_mask) != 0ull) ? 2 : ((any_tree || soon != 0ull) ? 1 : 0);
  }
  int dmax = 1;
#pragma unroll
  for (int d = 2; d <= WBC_MAX_DEPTH; ++d) dmax = (abits & C->depth_cp_mask[d]) ? d : dmax;
  dmax = max(dmax, ddmax);
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  if (lane == 0) s.dbg_ncon = __popcll(abits) | (dmax << 8) | (min((int)__popcll(abits & Cc->dyn_self_mask), 15) << 16);
#endif
  // damped block-Jacobi: relaxation 1 / (number of active contacts acting on the busier of the contact's two bodies)
  float com = 1.f;
  if (cact) {
    // (the free box: every contact of its row acts on it; a scalar count)
    const int tb1 = onbox ? 0 : cpb, tb2 = (cpb2 < 0 || p2box) ? tb1 : cpb2;
    const uint2 l1 = s.k_gmlo[tb1], l2 = s.k_gmlo[tb2];
    int c1 = __popc(ablo & (l1.x | l1.y)), c2 = __popc(ablo & (l2.x | l2.y));
    if (abhi != 0u) {                                   // (scalar: the upper slots are mostly idle)
      const uint2 h1 = s.k_gmhi[tb1], h2 = s.k_gmhi[tb2];
      c1 += __popc(abhi & (h1.x | h1.y)); c2 += __popc(abhi & (h2.x | h2.y));
    }
    const int cbox = __popc(abhi & 0xFFFFu);
    c1 = onbox ? cbox : c1; c2 = p2box ? cbox : (cpb2 < 0 ? 0 : c2);
    const int cnt = max(c1, c2);
    com = 1.f / (float)cnt;
  }
  f3 cdvr = mk3(0.f, 0.f, 0.f);       // this contact's sweep response and its Delassus block (cW) stay in the owning lane's registers
  if (any) {
    const int iters = C->cfg.contact_iters;
    // The tree sweeps' operands stay in registers across the iterations. Sweep layout (DevConst::sweep_pack): eight groups of 8 lanes
    // (sg = lane >> 3, component sk = lane & 7 < 6), a group walks a SEGMENT of up to three levels of one chain: lane (sg, sk) holds
    // component sk of S and U and 1/D of the segment's bodies (zeros where there is none and on the two spare lanes of a group: the
    // sweeps then need no lane masks) and the S.p products of the inward sweep (uD) for the outward one. A chain deeper than three
    // levels (the arm) takes the two groups of one 16-lane row: its deep half hands its wrench to the shallow half -- and receives the
    // acceleration from it -- by one DPP row shift. A level of a sweep is a handful of dependent vector instructions (the 6-term
    // products are DPP sums) instead of two LDS round trips: the solver's sweeps are the dependent chain of the waves a launch ends
    // with (a robot in contact: 4 substeps x contact_iters of them).
    const int sk = lane & 7;
    uint32_t sbody;
    {
      const int sg = lane >> 3;
      sbody = (uint32_t)__builtin_amdgcn_readfirstlane(C->sweep_pack[7]);     // (scalar loads and a select chain)
#pragma unroll
      for (int g = 0; g < 7; ++g) sbody = (sg == g) ? (uint32_t)__builtin_amdgcn_readfirstlane(C->sweep_pack[g]) : sbody;
    }
    const bool sdeep = ((sbody >> 18) & 1u) != 0u, shasdeep = ((sbody >> 19) & 1u) != 0u;
    float sS[3] = {0.f, 0.f, 0.f}, sU[3] = {0.f, 0.f, 0.f}, siD[3] = {0.f, 0.f, 0.f}, suD[3] = {0.f, 0.f, 0.f};
    const bool sroot = ((sbody >> 20) & 1u) != 0u;    // the group that adds the root's own wrench and stores the root's response
    float k0p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // row sk of K0 in the order the group's lanes are gathered: K0[sk][sk ^ p]
    if (any_tree) {
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        const int i = (sbody >> (5 * l)) & 31;
        const bool act = i != CH_NONE && sk < 6;
        const int ii = act ? i : 0, kk = act ? sk : 0;
        const float vs = s.S[ii][kk], vu = s.U[ii][kk], vd = s.iD[ii];
        sS[l] = act ? vs : 0.f; sU[l] = act ? vu : 0.f; siD[l] = act ? vd : 0.f;
      }
#pragma unroll
      for (int p2 = 0; p2 < 8; ++p2) {
        const int c = sk ^ p2;
        const bool ok = sk < 6 && c < 6;
        const float v = s.ctc.K0[ok ? sk * 6 + c : 0];
        k0p[p2] = ok ? v : 0.f;
      }
    }
    // levels a sweep has to reach: the shallow segments' (nlA) and, for a contact below level 3, the deep segments' (nlB)
    const int nlA = min(dmax, 3), nlB = max(dmax - 3, 0);
    for (int it = 0; it < iters; ++it) {
      if (it == 1) XSTAMP(25);
      if (cact) {
        const f3 own = sym_mul(cW, clamr);
        const f3 vref = cvfree + cdvr - own;
        float lam[3];
        contact_solve(cW, cn, cvtgt, cmu, vref, lam);
        // damped block-Jacobi: the active contacts acting on one body share the correction
        clamr = clamr + (mk3(lam[0], lam[1], lam[2]) - clamr) * com;
        st3(s.ctc.clam[lane], clamr);
        // this contact's wrench about F's origin, added to its bodies' pD = -f_ext (the partner body of a pair receives the opposite
        // wrench; the free box is not part of the tree): LDS float adds from the contact's own lane -- the solver's dependent chain
        // goes solve -> sweep without a per-body gather loop and its second hand-over in between
        if (any_tree) {
          const f3 f = clamr * idt, mom = cross(cxcr, f);
          if (!onbox) {
            float* pd = PD(s)[cpb];
            lds_add(&pd[0], -mom.x); lds_add(&pd[1], -mom.y); lds_add(&pd[2], -mom.z);
            lds_add(&pd[3], -f.x); lds_add(&pd[4], -f.y); lds_add(&pd[5], -f.z);
          }
          if (cpb2 >= 0 && !p2box) {
            float* pd = PD(s)[cpb2];
            lds_add(&pd[0], mom.x); lds_add(&pd[1], mom.y); lds_add(&pd[2], mom.z);
            lds_add(&pd[3], f.x); lds_add(&pd[4], f.y); lds_add(&pd[5], f.z);
          }
        }
      }
      WSYNC();
      if (it == 0) STAMP(20);
      if (it == 1) XSTAMP(26);
      // the free box: its contacts fill the 16-lane row 32..47 (corners: + the impulse; a robot sphere against it: - the impulse);
      // every lane of the row forms its wrench about the box centre (3 m from F's origin: no cancellation in fp32), a row reduction
      // sums them, the row's last lane turns the sum into the box's response (angular acceleration n / Ic, centre acceleration F / m)
      if ((abhi & 0xFFFFu) != 0u && (lane >> 4) == 2) {
        const f3 f = cact ? clamr * (onbox ? idt : -idt) : mk3(0.f, 0.f, 0.f);
        const f3 mom = cross(cact ? cxcr - ld3(s.bxc) : mk3(0.f, 0.f, 0.f), f);
        const float nx = row_scan16(mom.x), ny = row_scan16(mom.y), nz = row_scan16(mom.z);
        const float fx = row_scan16(f.x), fy = row_scan16(f.y), fz = row_scan16(f.z);
        if (lane == 47) { st3(&s.bxa[0], mk3(nx, ny, nz) * s.bxiI); st3(&s.bxa[3], mk3(fx, fy, fz) * s.bxim); }
      }
      if (any_tree) {
        if (it == 0) STAMP(21);
        if (it == 1) XSTAMP(27);
        {   // inward: lane (sg, sk) carries component sk of the accumulated wrench in a register. The segments' own wrenches are
            // requested at once; levels that carry no contact are skipped by scalar tests (their uD stays 0).
          float pd[3];
          float pd0 = 0.f;                      // the root's own wrench (on the lanes of ONE group)
          {
            const float v = PD(s)[0][sk < 6 ? sk : 0];
            pd0 = (sroot && sk < 6) ? v : 0.f;
          }
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            pd[l] = 0.f;
            if (l < nlA) {
              const int i = (sbody >> (5 * l)) & 31;
              const bool act = i != CH_NONE && sk < 6;
              const float v = PD(s)[act ? i : 0][act ? sk : 0];
              pd[l] = act ? v : 0.f;
            }
          }
          float carry = 0.f;
#pragma unroll
          for (int l = 2; l >= 0; --l) {
            if (l < nlA) {
              const float pk = pd[l] + carry;
              const float uD = -sum8(sS[l] * pk);
              suD[l] = uD;
              carry = pk + sU[l] * (uD * siD[l]);
            }
          }
          if (nlB > 0) {
            // a contact below level 3: every group has walked its segment (nlA = 3); the deep halves' results stand, the shallow
            // halves walk theirs again from the wrench their deep half hands over (the other groups arrive at what they had)
            const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(carry), 0x108, 0xF, 0xF, true));   // row_shl:8
            carry = shasdeep ? up : 0.f;
#pragma unroll
            for (int l = 2; l >= 0; --l) {
              const float pk = pd[l] + carry;
              const float uD = -sum8(sS[l] * pk);
              suD[l] = uD;
              carry = pk + sU[l] * (uD * siD[l]);
            }
          }
          // root, still in registers: pD0 = its own wrench + the chains' depth-1 contributions (summed over the eight groups by row
          // rotation / row swaps: every lane gets component sk), a0 = -K0 pD0 (the group's lanes gather pD0, each forms its own row)
          const float p0 = sum_groups8((sdeep ? 0.f : carry) + pd0);
          float a0 = k0p[0] * p0;
          a0 += k0p[1] * xor8<1>(p0); a0 += k0p[2] * xor8<2>(p0); a0 += k0p[3] * xor8<3>(p0);
          a0 += k0p[4] * xor8<4>(p0); a0 += k0p[5] * xor8<5>(p0); a0 += k0p[6] * xor8<6>(p0); a0 += k0p[7] * xor8<7>(p0);
          a0 = -a0;
          if (sroot && sk < 6) AD(s)[0][sk] = a0;       // (read by the contact points on the root body and by the integrator)
          if (it == 0) STAMP(22);
          if (it == 1) XSTAMP(28);
          if (it == 0) STAMP(23);
          if (it == 1) XSTAMP(29);
          // outward: component sk of the parent's acceleration change travels in a register. The sweeps before the last one stop at
          // the deepest level that carries a contact (only the contact points' responses are read); the last one yields every joint's
          // acceleration change.
          const bool last = it == iters - 1;
          const int nA = last ? 3 : nlA, nB = last ? 3 : nlB;
          float adk = a0;
          if (!last) for (int t = lane; t < WBC_NB * 6; t += LANES) (&PD(s)[0][0])[t] = 0.f;     // (every read of pD is behind us: the next sweep's adds start from zero)
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            if (l < nA) {
              const int i = (sbody >> (5 * l)) & 31;
              const bool act = i != CH_NONE && sk < 6 && !sdeep;
              const float ut = sum8(sU[l] * adk);
              const float qdd = (suD[l] - ut) * siD[l];
              adk += sS[l] * qdd;
              if (act) {
                AD(s)[i][sk] = adk;
                if (last && sk == 0) s.qddD[i] = qdd;
              }
            }
          }
          if (nB > 0) {
            adk = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(adk), 0x118, 0xF, 0xF, true));               // row_shr:8
#pragma unroll
            for (int l = 0; l < 3; ++l) {
              if (l < nB) {
                const int i = (sbody >> (5 * l)) & 31;
                const bool act = i != CH_NONE && sk < 6 && sdeep;
                const float ut = sum8(sU[l] * adk);
                const float qdd = (suD[l] - ut) * siD[l];
                adk += sS[l] * qdd;
                if (act) {
                  AD(s)[i][sk] = adk;
                  if (last && sk == 0) s.qddD[i] = qdd;
                }
              }
            }
          }
          WSYNC();
        }
      } else {
        WSYNC();
      }
      if (it == 0) STAMP(24);
      if (it == 1) XSTAMP(30);
      if (it < iters - 1 && cact) {      // the last sweep's contact-point response is not used
        const f3 xc = cxcr;
        // response of a body at the contact point: (angular; linear) acceleration change, lever from F's origin (tree) / from the
        // box centre (free box)
        const float* ad1 = onbox ? s.bxa : AD(s)[onbox ? 0 : cpb];
        const f3 lv1 = onbox ? xc - ld3(s.bxc) : xc;
        f3 dvv = (ld3(ad1 + 3) + cross(ld3(ad1), lv1)) * dt;
        if (cpb2 >= 0) {
          const float* ad2 = p2box ? s.bxa : AD(s)[p2box ? 0 : cpb2];
          const f3 lv2 = p2box ? xc - ld3(s.bxc) : xc;
          dvv = dvv - (ld3(ad2 + 3) + cross(ld3(ad2), lv2)) * dt;
        }
        cdvr = dvv;
      }
      WSYNC();
      if (it == 1) XSTAMP(31);
    }
  }
  STAMP(8);
  // contact force outputs (world-frame net force per rigid body, foot-frame sensor wrench)
  if (want_outputs) {
    // a foot's sensor reads the normals of its sphere's contacts (the terrain's, the free box's) from their owning lanes; all lanes
    // take part in the shuffles
    const int ftl = lane - 32;
    const bool isft = ftl >= 0 && ftl < WBC_NFEET;
    const int kc1 = isft ? Cc->foot_cp[ftl] : lane;
    const int kc2 = isft ? Cc->foot_cp2[ftl] : -1;
    const int src2 = kc2 >= 0 ? kc2 : lane;
    const f3 n1 = mk3(__shfl(cn.x, kc1), __shfl(cn.y, kc1), __shfl(cn.z, kc1));
    const f3 n2 = mk3(__shfl(cn.x, src2), __shfl(cn.y, src2), __shfl(cn.z, src2));
    // lanes 0..27: net force on rigid body `lane` (+ as the sphere's body, - as the partner of a pair); row 27 is the box actor
    if (lane == WBC_BOX_RB) {                     // the box actor's row: the net force the last sweep's row reduction left (F = m a)
      st3(s.out_contact[lane], mat_mul(s.R, ld3(&s.bxa[3]) * (1.f / s.bxim)));
    } else if (lane < WBC_NRB) {
      const uint64_t m1 = Cc->out_cp_mask[lane], m2 = Cc->out_cp2_mask[lane];
      f3 acc = mk3(0.f, 0.f, 0.f);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half && abhi == 0u) break;
        const uint32_t h1 = (uint32_t)(m1 >> (32 * half));
        uint32_t mask = (h1 | (uint32_t)(m2 >> (32 * half))) & (half ? abhi : ablo);
        while (mask) {
          const int kb = __ffs(mask) - 1, kc = kb + 32 * half;
          mask &= mask - 1;
          const float sg = ((h1 >> kb) & 1u) ? idt : -idt;
          acc = acc + ld3(s.ctc.clam[kc]) * sg;
        }
      }
      st3(s.out_contact[lane], mat_mul(s.R, acc));
    } else if (isft) {
      f3 fa = mk3(0.f, 0.f, 0.f), ta = mk3(0.f, 0.f, 0.f);
      const int b = Cc->model.cp_body[kc1];
      const float rad = Cc->model.cp_radius[kc1];
      if ((abits >> kc1) & 1ull) {
        const f3 f = ld3(s.ctc.clam[kc1]) * idt;
        fa = matT_mul(s.E[b], f);
        ta = matT_mul(s.E[b], cross(n1 * (-rad), f));
      }
      if (kc2 >= 0 && ((abits >> kc2) & 1ull)) {
        const f3 f = ld3(s.ctc.clam[kc2]) * idt;
        fa = fa + matT_mul(s.E[b], f);
        ta = ta + matT_mul(s.E[b], cross(n2 * (-rad), f));
      }
      st3(&s.out_sensor[ftl][0], fa); st3(&s.out_sensor[ftl][3], ta);
    }
    // the dynamic slots in contact (rare): one after the other, the owning lane adds its force to the rows of its pair's two rigid
    // bodies (the free box's row already holds it: the row reduction) and, where a foot is one of them, to that foot's sensor
    if (dynbits != 0ull) {
      WSYNC();
      uint64_t db = dynbits;
      while (db != 0ull) {
        const int d = __ffsll((unsigned long long)db) - 1;
        db &= db - 1;
        if (lane == d) {
          const int drb = dpk & 255, drb2 = (dpk >> 8) & 255, dfb = (dpk >> 16) & 3, dcand = (dpk >> 18) & 63;
          const float drad2 = Cc->cand_rad[dcand][3 + dfb];               // the partner limb's touching feature
          const f3 f = clamr * idt, fw = mat_mul(s.R, f);
          st3(s.out_contact[drb], ld3(s.out_contact[drb]) + fw);
          if (drb2 != WBC_BOX_RB) st3(s.out_contact[drb2], ld3(s.out_contact[drb2]) - fw);
#pragma unroll
          for (int ft = 0; ft < WBC_NFEET; ++ft) {
            const int frb = Cc->model.feet_rb[ft];
            if (frb == drb) {
              st3(&s.out_sensor[ft][0], ld3(&s.out_sensor[ft][0]) + matT_mul(s.E[cpb], f));
              st3(&s.out_sensor[ft][3], ld3(&s.out_sensor[ft][3]) + matT_mul(s.E[cpb], cross(cn * (-cpr), f)));
            }
            if (frb == drb2) {
              const f3 fo = f * -1.f;
              st3(&s.out_sensor[ft][0], ld3(&s.out_sensor[ft][0]) + matT_mul(s.E[cpb2], fo));
              st3(&s.out_sensor[ft][3], ld3(&s.out_sensor[ft][3]) + matT_mul(s.E[cpb2], cross(cn * drad2, fo)));
            }
          }
        }
        WSYNC();
      }
    }
  }
  STAMP(9);
  // integrate (semi-implicit Euler)
  float a0[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) a0[j] = s.a[0][j] + AD(s)[0][j];
  WSYNC();
  if (lane >= 1 && lane < WBC_NB) {
    const int dj = (s.k_body[lane] >> 2) & 31;
    float qd = s.qd[dj] + dt * (s.qdd[lane] + s.qddD[lane]);
    const float lim = s.k_qdlim[dj];
    if (lim > 0.f) qd = fminf(fmaxf(qd, -lim), lim);
    s.qd[dj] = qd;
    s.q[dj] += dt * qd;
  } else if (lane == WBC_NB && asleep) {
    st3(&s.box[7], mk3(0.f, 0.f, 0.f)); st3(&s.box[10], mk3(0.f, 0.f, 0.f));      // frozen
  } else if (lane == 0 || lane == WBC_NB) {
    // the two free bodies, same instructions: lane 0 the robot's root, lane WBC_NB the box actor (gravity + its net contact force)
    const bool isb = lane != 0;
    float* st = isb ? s.box : s.root;
    const f3 wb = ld3(s.wb), vb = ld3(s.vb);
    const f3 accF = isb ? ld3(s.gF) + ld3(&s.bxa[3]) : mk3(a0[3], a0[4], a0[5]) + ld3(s.gF) + cross(wb, vb);
    const f3 alF = isb ? ld3(&s.bxa[0]) : mk3(a0[0], a0[1], a0[2]);
    const f3 vw = ld3(&st[7]) + mat_mul(s.R, accF) * dt;
    const f3 ww = ld3(&st[10]) + mat_mul(s.R, alF) * dt;
    st3(&st[7], vw); st3(&st[10], ww);
    st3(&st[0], ld3(&st[0]) + vw * dt);
    const float om[4] = {ww.x, ww.y, ww.z, 0.f};
    float dq[4], nq[4], nn = 0.f;
    quat_mul(om, &st[3], dq);
#pragma unroll
    for (int j = 0; j < 4; ++j) { nq[j] = st[3 + j] + 0.5f * dt * dq[j]; nn += nq[j] * nq[j]; }
    nn = __builtin_amdgcn_rsqf(nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) st[3 + j] = nq[j] * nn;
  }
  WSYNC();
  STAMP(10);
}

// rigid_body_state of the 27 robot bodies + box from the LDS state (oracle: update_rigid_body_state).
// Needs E/pos of the CURRENT q (fk_pass) and s.R.
__device__ void rigid_body_pass(Smem& s, CP C, const ChainRegs& cr, const int chain, const int k) {
  const int lane = threadIdx.x;
  if (lane == 0) {
    quat_to_mat(&s.root[3], s.R);
    for (int j = 0; j < 4; ++j) s.post.quatB[0][j] = s.root[3 + j];
  }
  joint_pre_pass<true>(s, C);
  WSYNC();
  // frames and spatial velocities (frame F: the base's axes, origin at the base) of every body: one walk per chain
  kin_walk<true>(s, C, lane);
  // orientations: one lane per chain multiplies the joint quaternions down its chain (half-angle sin / cos: joint_pre_pass<true>),
  // every operand requested before the walk
  // (the quaternions riding along the walk instead -- component `row` on the quad's four lanes, cos * own + (+-sin) * the partner one
  // quad permutation away, 4 instructions per level -- was built and measured on one box: 83.2 against 82.2 us at 1024 envs, 100.8
  // against 100.0 at 4096; not kept)
  if (k == 1) {
    float qp[4] = {s.root[3], s.root[4], s.root[5], s.root[6]};
    float sh[WBC_MAX_DEPTH], ch[WBC_MAX_DEPTH];
#pragma unroll
    for (int d = 0; d < WBC_MAX_DEPTH; ++d) { const float2 h = *reinterpret_cast<const float2*>(&s.jt[ch_dof(cr, d)][2]); sh[d] = h.x; ch[d] = h.y; }
#pragma unroll
    for (int d = 0; d < WBC_MAX_DEPTH; ++d) {
      const int i = ch_body(cr, d);
      if (i != CH_NONE) {
        const int ax = ch_ax(cr, d);
        const float qa[4] = {(ax == 0) ? sh[d] : 0.f, (ax == 1) ? sh[d] : 0.f, (ax == 2) ? sh[d] : 0.f, ch[d]};
        float qn[4];
        quat_mul(qp, qa, qn);
#pragma unroll
        for (int j = 0; j < 4; ++j) { qp[j] = qn[j]; s.post.quatB[i][j] = qn[j]; }
      }
    }
  }
  WSYNC();
  // rigid bodies: pose of the body's frame + the importer's offset; velocities from the body's spatial velocity (w; vl) in F: the point
  // p moves with vl + w x p, the world sees R times that (the base's own velocity is part of every body's: v_0 = R^T of the root's)
  if (lane < WBC_NRB) {
    const int r = lane, b = C->model.rb_body[r];
    const f3 t = mat_mul(s.E[b], ld3(C->model.rb_offset[r]));
    const f3 pF = ld3(s.pos[b]) + t;
    const f3 w = ld3(&s.v[b][0]), vl = ld3(&s.v[b][3]);
    st3(&s.post.out_rb[r][0], ld3(&s.root[0]) + mat_mul(s.R, pF));
    for (int j = 0; j < 4; ++j) s.post.out_rb[r][3 + j] = s.post.quatB[b][j];
    st3(&s.post.out_rb[r][7], mat_mul(s.R, vl + cross(w, pF)));
    st3(&s.post.out_rb[r][10], mat_mul(s.R, w));
  } else if (lane == WBC_NRB) {
    for (int j = 0; j < 13; ++j) s.post.out_rb[WBC_NRB][j] = s.box[j];
  }
  WSYNC();
}

// euler_from_quat of the base and of the gripper, the six angles on six lanes: roll and yaw are the same expression with the roles
// of x and z exchanged (atan2(2 (w p + y r), 1 - 2 (p p + y y)), p = x | z, r = z | x), the pitch's arcsine is an atan2 as well -- ONE
// evaluation of the 30-instruction arctangent instead of six on lane 0 (every instruction of a single-lane phase is a full
// wavefront issue slot). Same float expressions as euler_from_quat (wbc_device.h).
__device__ __forceinline__ void euler_batch(Smem& s, const float* q_root, const float* q_ee) {
  const int lane = threadIdx.x;
  if (lane < 6) {
    const int c = lane < 3 ? lane : lane - 3;
    const float* q = lane < 3 ? q_root : q_ee;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float p = c == 2 ? z : x, r = c == 2 ? x : z;
    float Y = 2 * (w * p + y * r), X = 1 - 2 * (p * p + y * y);
    if (c == 1) {
      float sp = 2 * (w * y - z * x);
      sp = fminf(fmaxf(sp, -1.f), 1.f);
      Y = sp; X = __builtin_amdgcn_sqrtf(fmaxf((1.f - sp) * (1.f + sp), 0.f));
    }
    s.eul[lane] = fast_atan2f(Y, X);
  }
}

enum { G_START = 0, G_GOAL = 3, G_GOAL_CART = 6, G_CURR = 9, G_CURR_CART = 12, G_DORN = 15, G_ORN = 18,
       G_TIMER = 21, G_TRAJ = 22, G_TOTAL = 23 };

// ---- random events of the task logic, wave-cooperative --------------------------------------------------------------------------
// The reference's samplers draw a handful of uniforms per event and _resample_ee_goal (WG:1303-1350) retries up to ten times until
// the straight line from the old goal to the candidate misses the robot's body; every draw is a counter hash of (seed, env, step,
// slot) -- two 64-bit mixes, ~80 dependent integer instructions. Done on lane 0, one reset was 30-40 dependent hashes plus up to ten
// 10-sample collision tests, in the wave the launch ends with (the slowest 1 % of the waves of a launch were all resetting ones:
// tools/wave_spread.py). Here lane j hashes slot j, the (candidate, sample) pairs of all ten candidates are tested at once, and lane
// 0 only applies the result. Same integers, same float expressions as the oracle's serial loop.
#define DRAWS(s) (&(s).pA[0][0])    // scratch: the bias-force array is dead once the substeps are done (114 floats)
enum { DR_PICK = 48, DR_MASK = 51 };  // [0..47] u01 draws of consecutive slots, [48..50] the picked EE goal (l, p, y), [51] (int) candidates that collide
static_assert(SLOT_PUSH + 2 <= DR_PICK && SLOT_RESET_GOAL_SPHERE + 30 - SLOT_RESET_XY <= DR_PICK && DR_MASK < WBC_NB * 6, "draw scratch layout");

// all lanes: DRAWS[j] = u01(slot0 + j), j < count
__device__ __forceinline__ void draw_block(Smem& s, uint64_t seed, int env, uint64_t step, int slot0, int count) {
  const int lane = threadIdx.x;
  if (lane < count) DRAWS(s)[lane] = rng_u01(seed, env, step, slot0 + lane);
  if (lane == 0) reinterpret_cast<int*>(DRAWS(s))[DR_MASK] = 0;
  WSYNC();
}

// all lanes: the rejection loop of _resample_ee_goal. DRAWS[sph0 + 3 r + j]: draw j of candidate r. Leaves the first candidate whose
// path from the current goal is free (the tenth if none is) in DRAWS[DR_PICK..].
__device__ void goal_pick(Smem& s, CP C, int sph0) {
  const int lane = threadIdx.x;
  const int ns = C->cfg.goal_collision_samples;
  const float* u = DRAWS(s) + sph0;
  int* mask = reinterpret_cast<int*>(DRAWS(s)) + DR_MASK;
  for (int p = lane; p < 10 * ns; p += LANES) {
    const int r = p / ns, kk = p - r * ns;
    const float t = (ns > 1) ? (float)kk / (float)(ns - 1) : 0.f;
    const f3 g = mk3(urange(C->cur.goal_l_range[0], C->cur.goal_l_range[1], u[3 * r]), urange(C->cur.goal_p_range[0], C->cur.goal_p_range[1], u[3 * r + 1]),
                     urange(C->cur.goal_y_range[0], C->cur.goal_y_range[1], u[3 * r + 2]));
    const f3 sp = mk3(lerp_torch(s.goal[G_GOAL], g.x, t), lerp_torch(s.goal[G_GOAL + 1], g.y, t), lerp_torch(s.goal[G_GOAL + 2], g.z, t));
    const f3 c = sphere2cart(sp);
    const float cv[3] = {c.x, c.y, c.z};
    int inside = 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) inside &= (cv[j] < C->cfg.goal_collision_upper[j]) && (cv[j] > C->cfg.goal_collision_lower[j]);
    if (inside | (c.z < C->cfg.goal_underground_limit)) __hip_atomic_fetch_or(mask, 1 << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  WSYNC();
  const int m = *mask & 1023;
  const int rs = (m == 1023) ? 9 : __ffs(~m) - 1;
  if (lane < 3) {
    const float lo = lane == 0 ? C->cur.goal_l_range[0] : (lane == 1 ? C->cur.goal_p_range[0] : C->cur.goal_y_range[0]);
    const float hi = lane == 0 ? C->cur.goal_l_range[1] : (lane == 1 ? C->cur.goal_p_range[1] : C->cur.goal_y_range[1]);
    DRAWS(s)[DR_PICK + lane] = urange(lo, hi, u[3 * rs + lane]);
  }
  WSYNC();
}

// lane 0: _resample_commands (WG:917-935) from its two draws
__device__ __forceinline__ void resample_commands(Smem& s, CP C, float u0, float u1) {
  const float cx = urange(C->cur.lin_vel_x_range[0], C->cur.lin_vel_x_range[1], u0);
  const float cy = urange(C->cur.ang_vel_yaw_range[0], C->cur.ang_vel_yaw_range[1], u1);
  const bool keep = (cx > C->cfg.lin_vel_x_clip) || (fabsf(cy) > C->cfg.ang_vel_yaw_clip);
  s.cmd[0] = keep ? cx : 0.f; s.cmd[1] = 0.f; s.cmd[2] = keep ? cy : 0.f;
}

// lane 0: _resample_ee_goal's bookkeeping once goal_pick has chosen; DRAWS[orn0 + j]: the orientation draws
__device__ void goal_apply(Smem& s, CP C, int orn0, float base_yaw) {
  for (int j = 0; j < 3; ++j) {
    const float d = urange(C->cfg.goal_delta_orn_range[j][0], C->cfg.goal_delta_orn_range[j][1], DRAWS(s)[orn0 + j]);
    s.goal[G_DORN + j] = d;
    s.goal[G_ORN + j] = wrap_to_pi(d + (j == 2 ? base_yaw : 0.f));
  }
  for (int j = 0; j < 3; ++j) { s.goal[G_START + j] = s.goal[G_GOAL + j]; s.goal[G_GOAL + j] = DRAWS(s)[DR_PICK + j]; }
  st3(&s.goal[G_GOAL_CART], sphere2cart(ld3(&s.goal[G_GOAL])));
  s.goal[G_TIMER] = 0.f;
}

__device__ const int8_t POLICY_PERM[WBC_NDOF] = {3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8, 12, 13, 14, 15, 16, 17, 18, 19};
// terms feeding each metric slot (-1 padded, ascending): a metric slot is owned by one lane
__device__ const int8_t MET_TERMS[WBC_NMETRIC][2] = {
    /* LEG_ENERGY_ABS_SUM */ {WBC_REW_LEG_ENERGY_ABS_SUM, -1},
    /* TRACKING_LIN_VEL_X_L1 */ {WBC_REW_TRACKING_LIN_VEL_X_L1, WBC_REW_TRACKING_LIN_VEL_X_EXP},
    /* TRACKING_ANG_VEL_YAW_EXP */ {WBC_REW_TRACKING_ANG_VEL_YAW_EXP, -1},
    /* TRACKING_EE_CART */ {WBC_REW_TRACKING_EE_CART, -1},
    /* TRACKING_EE_SPHERE */ {WBC_REW_TRACKING_EE_SPHERE, -1},
    /* TRACKING_EE_ORN */ {WBC_REW_TRACKING_EE_ORN_RY, -1},
    /* LEG_ACTION_L2 */ {WBC_REW_HIP_ACTION_L2, WBC_REW_LEG_ACTION_L2},
    /* TORQUE */ {WBC_REW_TORQUES, -1},
    /* ENERGY_SQUARE */ {WBC_REW_ENERGY_SQUARE, -1},
    /* FOOT_CONTACTS_Z */ {WBC_REW_FOOT_CONTACTS_Z, -1}};

// compute_reward of the oracle, executed by lane 0 on LDS state
// sums over the DoFs / actions that the base class's reward terms need (computed lane-parallel by base_reward_sums, all lanes)
struct BaseSums { float dv2, da2, ar2, plim, vlim, tlim, still; };

template <class TT> __device__ void compute_reward(Smem& s, const TT& T, CP C, const float* yq, const float ncol, const bool base_on,
                               const BaseSums& bs, int env) {
  const wbc_task_cfg& cf = C->cfg;
  const float inv_sig = rcpf(cf.tracking_sigma), inv_ee_sig = rcpf(cf.tracking_ee_sigma);
  float term[WBC_NREW_WG], met_src[WBC_NREW_WG];
#pragma unroll
  for (int t = 0; t < WBC_NREW_WG; ++t) met_src[t] = 0.f;
  const float* ee_pos = s.post.out_rb[C->model.gripper_rb];
  const float* ee_orn = ee_pos + 3;
  float sq = 0.f, abs_sum = 0.f, sum = 0.f, arm_abs = 0.f, tq2 = 0.f, act_leg = 0.f;
  for (int j = 0; j < 12; ++j) {
    const float p = s.tau[j] * s.qd[j];
    sq += p * p; abs_sum += fabsf(p); sum += p; act_leg += s.act[j] * s.act[j];
  }
  for (int j = 12; j < WBC_NDOF - 2; ++j) arm_abs += fabsf(s.tau[j] * s.qd[j]);
  for (int j = 0; j < WBC_NDOF; ++j) tq2 += s.tau[j] * s.tau[j];
  term[WBC_REW_ENERGY_SQUARE] = sq;
  term[WBC_REW_SURVIVE] = 1.f;
  const float ex = fabsf(s.cmd[0] - s.blv[0]);
  term[WBC_REW_TRACKING_LIN_VEL_X_L1] = -ex + fabsf(s.cmd[0]);
  term[WBC_REW_TRACKING_LIN_VEL_X_EXP] = fast_expf(-ex * inv_sig);
  const float eyaw = fabsf(s.cmd[2] - s.bav[2]);
  term[WBC_REW_TRACKING_ANG_VEL_YAW_EXP] = fast_expf(-eyaw * inv_sig);
  term[WBC_REW_TRACKING_ANG_VEL_YAW_L1] = -eyaw + fabsf(s.cmd[2]);
  const float hip = s.act[0] * s.act[0] + s.act[3] * s.act[3] + s.act[6] * s.act[6] + s.act[9] * s.act[9];
  term[WBC_REW_HIP_ACTION_L2] = hip;
  float fz = 0.f;
  for (int f = 0; f < 4; ++f) fz += s.out_sensor[f][2] * s.out_sensor[f][2];
  term[WBC_REW_FOOT_CONTACTS_Z] = fz;
  const f3 rel = mk3(ee_pos[0] - s.root[0], ee_pos[1] - s.root[1], ee_pos[2] - cf.z_invariant_offset);
  const f3 loc = quat_rotate_inverse(yq, rel);
  const f3 sph = cart2sphere(loc);
  const float es = fabsf(sph.x - s.goal[G_CURR]) * cf.sphere_error_scale[0] + fabsf(sph.y - s.goal[G_CURR + 1]) * cf.sphere_error_scale[1] +
                   fabsf(sph.z - s.goal[G_CURR + 2]) * cf.sphere_error_scale[2];
  term[WBC_REW_TRACKING_EE_SPHERE] = fast_expf(-es * inv_ee_sig);
  const float yq_inv[4] = {-yq[0], -yq[1], -yq[2], yq[3]};
  const f3 tw = quat_rotate_inverse(yq_inv, ld3(&s.goal[G_CURR_CART]));
  const float ec = fabsf(ee_pos[0] - (s.root[0] + tw.x)) + fabsf(ee_pos[1] - (s.root[1] + tw.y)) + fabsf(ee_pos[2] - (cf.z_invariant_offset + tw.z));
  term[WBC_REW_TRACKING_EE_CART] = fast_expf(-ec * inv_ee_sig);
  const f3 eul = ld3(&s.eul[3]);                    // euler_from_quat(ee_orn), computed by euler_batch
  const float eu[3] = {eul.x, eul.y, eul.z};
  float eo = 0.f, eo_ry = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float d = wrap_to_pi(s.goal[G_ORN + j] - eu[j]);
    eo += fabsf(d) * cf.orn_error_scale[j];
    if (j != 1) eo_ry += fabsf(d * cf.orn_error_scale[j]);
  }
  term[WBC_REW_TRACKING_EE_ORN] = fast_expf(-eo * inv_ee_sig);
  term[WBC_REW_TRACKING_EE_ORN_RY] = fast_expf(-eo_ry * inv_ee_sig);
  term[WBC_REW_LEG_ENERGY_ABS_SUM] = abs_sum;
  term[WBC_REW_LEG_ENERGY_SUM_ABS] = fabsf(sum);
  term[WBC_REW_LEG_ACTION_L2] = act_leg;
  term[WBC_REW_LEG_ENERGY] = sum;
  term[WBC_REW_ARM_ENERGY_ABS_SUM] = arm_abs;
  const float dx = s.cmd[0] - s.blv[0], dy = s.cmd[1] - s.blv[1], dz = s.cmd[2] - s.blv[2];
  term[WBC_REW_TRACKING_LIN_VEL] = fast_expf(-(dx * dx + dy * dy) * inv_sig);
  term[WBC_REW_TRACKING_LIN_VEL_Y_L2] = dy * dy;
  term[WBC_REW_TRACKING_LIN_VEL_Z_L2] = dz * dz;
  term[WBC_REW_TORQUES] = tq2;
  term[WBC_REW_COLLISION] = ncol;
  met_src[WBC_REW_ENERGY_SQUARE] = sq; met_src[WBC_REW_TRACKING_LIN_VEL_X_L1] = ex; met_src[WBC_REW_TRACKING_LIN_VEL_X_EXP] = ex;
  met_src[WBC_REW_TRACKING_ANG_VEL_YAW_EXP] = eyaw; met_src[WBC_REW_HIP_ACTION_L2] = hip; met_src[WBC_REW_LEG_ACTION_L2] = act_leg;
  met_src[WBC_REW_FOOT_CONTACTS_Z] = fz; met_src[WBC_REW_TRACKING_EE_SPHERE] = es; met_src[WBC_REW_TRACKING_EE_CART] = ec;
  met_src[WBC_REW_TRACKING_EE_ORN_RY] = eo_ry; met_src[WBC_REW_LEG_ENERGY_ABS_SUM] = abs_sum; met_src[WBC_REW_TORQUES] = tq2;
  // the per-term scaling / episode sums / metrics are done by lanes t < WBC_NREW (reward_accumulate)
#pragma unroll
  for (int t = 0; t < WBC_NREW_WG; ++t) { s.term[t] = term[t]; s.msrc[t] = met_src[t]; }
  // ---- the base class's terms (legged_robot.py:832-922; oracle: compute_reward), only when a config switches one on ----
  if (base_on) {
    s.term[WBC_REW_LIN_VEL_Z] = s.blv[2] * s.blv[2];
    s.term[WBC_REW_ANG_VEL_XY] = s.bav[0] * s.bav[0] + s.bav[1] * s.bav[1];
    s.term[WBC_REW_DOF_VEL] = bs.dv2; s.term[WBC_REW_DOF_ACC] = bs.da2; s.term[WBC_REW_ACTION_RATE] = bs.ar2;
    s.term[WBC_REW_TERMINATION] = (s.reset_flag && !s.time_out) ? 1.f : 0.f;
    s.term[WBC_REW_DOF_POS_LIMITS] = bs.plim; s.term[WBC_REW_DOF_VEL_LIMITS] = bs.vlim; s.term[WBC_REW_TORQUE_LIMITS] = bs.tlim;
    s.term[WBC_REW_TRACKING_ANG_VEL] = fast_expf(-(eyaw * eyaw) * inv_sig);
    const float cmd_xy = __builtin_amdgcn_sqrtf(s.cmd[0] * s.cmd[0] + s.cmd[1] * s.cmd[1]);
    s.term[WBC_REW_STAND_STILL] = bs.still * (cmd_xy < 0.1f ? 1.f : 0.f);
    float stumble = 0.f, fcf = 0.f, air = 0.f;
    const bool air_on = (((C->cur.leg_active_mask | C->cur.arm_active_mask) >> WBC_REW_FEET_AIR_TIME) & 1ull) != 0ull;
    const float dtp = cf.sim_dt * (float)cf.decimation;
    for (int f = 0; f < WBC_NFEET; ++f) {
      const f3 cf3 = ld3(s.out_contact[C->model.feet_rb[f]]);
      if (__builtin_amdgcn_sqrtf(cf3.x * cf3.x + cf3.y * cf3.y) > 5.f * fabsf(cf3.z)) stumble = 1.f;
      fcf += fmaxf(__builtin_amdgcn_sqrtf(dot(cf3, cf3)) - cf.max_contact_force, 0.f);
      if (air_on) {                                   // feet_air_time's state advances only while the function is in a reward list
        float at = ROW(T.feet_air_time, env, WBC_NFEET)[f];
        const bool contact = cf3.z > 1.f;
        const bool filt = contact || ROW(T.last_contacts, env, WBC_NFEET)[f] != 0.f;
        ROW(T.last_contacts, env, WBC_NFEET)[f] = contact ? 1.f : 0.f;
        const bool first = at > 0.f && filt;
        at += dtp;
        air += first ? at - 0.5f : 0.f;
        ROW(T.feet_air_time, env, WBC_NFEET)[f] = filt ? 0.f : at;
      }
    }
    s.term[WBC_REW_STUMBLE] = stumble; s.term[WBC_REW_FEET_CONTACT_FORCES] = fcf;
    s.term[WBC_REW_FEET_AIR_TIME] = air * (cmd_xy > 0.1f ? 1.f : 0.f);
    const float bh = s.root[2] - cf.base_height_target;
    s.term[WBC_REW_BASE_HEIGHT] = bh * bh;
  }
}

// The base class's per-DoF / per-action sums, one DoF per lane and a butterfly (all lanes; called only when a base term is on).
template <class TT> __device__ BaseSums base_reward_sums(const Smem& s, const TT& T, CP C, int env) {
  const int j = threadIdx.x;
  const wbc_task_cfg& cf = C->cfg;
  float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < WBC_NDOF) {
    const float q = s.q[j], qd = s.qd[j], tau = s.tau[j];
    const float acc = (ROW(T.last_dof_vel, env, WBC_NDOF)[j] - qd) / (cf.sim_dt * (float)cf.decimation);
    v[0] = qd * qd; v[1] = acc * acc;
    v[3] = -fminf(q - cf.soft_dof_lower[j], 0.f) + fmaxf(q - cf.soft_dof_upper[j], 0.f);
    v[4] = clampf(fabsf(qd) - cf.soft_dof_vel_limit[j], 0.f, 1.f);
    v[5] = fmaxf(fabsf(tau) - cf.soft_torque_limit[j], 0.f);
    v[6] = fabsf(q - cf.default_dof_pos[j]);
  }
  if (j < WBC_NACT) { const float d = ROW(T.last_actions, env, WBC_NACT)[j] - s.act[j]; v[2] = d * d; }
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
  // (the DoFs sit in lanes 0..19: after the xor 16 .. 1 butterfly every lane below 32 holds the sum over lanes 0..31)
  BaseSums b; b.dv2 = v[0]; b.da2 = v[1]; b.ar2 = v[2]; b.plim = v[3]; b.vlim = v[4]; b.tlim = v[5]; b.still = v[6];
  return b;
}

// rew_buf / arm_rew_buf, episode sums and metric sums from the raw terms: lane t owns term t, lane m metric slot m.
// lsc / asc = this lane's leg / arm reward scale (lanes >= WBC_NREW: 0). A term is evaluated when its function is in the list the
// reference builds at construction (cur.*_active_mask, WG:128-157), whatever its current (scheduled) scale. Per slot the
// order of additions is the reference's (leg channel then arm channel, terms ascending); the two reward totals are
// butterfly sums over the wavefront.
__device__ __forceinline__ void reward_accumulate(Smem& s, CP C, float lsc, float asc) {
  const int lane = threadIdx.x;
  const uint64_t lmask = C->cur.leg_active_mask, amask = C->cur.arm_active_mask;
  float vl = 0.f, va = 0.f;
  if (lane < WBC_NREW) {
    const float tm = s.term[lane];
    float e = s.ep_sums[lane];
    if ((lmask >> lane) & 1ull) { vl = tm * lsc; e += vl; }
    if ((amask >> lane) & 1ull) { va = tm * asc; e += va; }
    s.ep_sums[lane] = e;
    if (lane == WBC_REW_TERMINATION) { vl = 0.f; va = 0.f; }        // joins the totals after the only_positive_rewards clip (lane 0 below)
  }
  if (lane < WBC_NMETRIC) {
    float mt = s.met_sums[lane];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const uint64_t mask = ch == 0 ? lmask : amask;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = MET_TERMS[lane][j];
        if (t >= 0 && ((mask >> t) & 1ull)) mt += s.msrc[t];
      }
    }
    s.met_sums[lane] = mt;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { vl += __shfl_xor(vl, off); va += __shfl_xor(va, off); }
  if (lane == 0) {
    float r = vl, ra = va;
    if (C->cfg.only_positive_rewards && r < 0.f) r = 0.f;
    if (C->cfg.only_positive_rewards && ra < 0.f) ra = 0.f;
    if (((lmask | amask) >> WBC_REW_TERMINATION) & 1ull) {           // WG:184-188, 200-203
      const float tt = s.term[WBC_REW_TERMINATION];
      if ((lmask >> WBC_REW_TERMINATION) & 1ull) r += tt * C->cur.leg_reward_scale[WBC_REW_TERMINATION];
      if ((amask >> WBC_REW_TERMINATION) & 1ull) ra += tt * C->cur.arm_reward_scale[WBC_REW_TERMINATION];
    }
    s.rew = r / 100.f;
    s.arm_rew = ra / 100.f;
  }
}

// policy order -> simulator order of the 18 actions (POLICY_PERM for lane < 18, as arithmetic: the leg pairs FL<->FR, RL<->RR swap)
__device__ __forceinline__ int policy_perm18(int j) { return j < 12 ? ((j / 3) ^ 1) * 3 + j % 3 : j; }

// The kernel's prologue: the model constants the dependent chains index per lane (constant block -> LDS), this lane's chain registers,
// one env's state (HBM -> LDS; consecutive lanes read consecutive words) and -- STEP -- the action with its reorder, clip and delay
// FIFO (WG:1162-1168). EVERY load is issued before the first store: ~30 independent requests, one wait. (Written phase by phase --
// `if (lane < n) s.x[lane] = row[lane]` -- each masked region loads, waits and stores before the next one's pointer is even
// fetched: twenty global round trips in a row, 8.7 k cycles at the head of every wave's step.) Loads are unmasked with clamped
// indices; only the LDS stores are masked.
template <bool STEP, class TT>
__device__ __forceinline__ void prologue(Smem& s, ChainRegs& cr, const TT& T, CP C, int env, int chain, const float* __restrict__ actions) {
  // (a laundered lane index: the masks of this function's `lane < n` stores are otherwise kept for the identical conditions of the
  // write-back at the other end of the kernel -- ten SGPR pairs spilled through v_writelane / v_readlane; a compare is one instruction)
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));
  lane &= LANES - 1;
  // ---- loads: constants
  const float c_jxyz = (&C->model.joint_xyz[0][0])[min(lane, WBC_NB * 3 - 1)];
  const int lb = min(lane, WBC_NB - 1);
  const uint32_t c_body = C->body_pack[lb];
  const uint64_t c_m1 = C->body_cp_mask[lb], c_m2 = C->body_cp2_mask[lb];
  const uint32_t c_prk = C->pr_pack[lane];
  const float c_qdlim = C->model.qd_limit[min(lane, WBC_NDOF - 1)];
  const float c_arm = (&C->chain_arm[0][0])[min(lane, (WBC_NCHAIN + 1) * WBC_MAX_DEPTH - 1)];
  const int qt = min(lane >> 3, 2), qq = min(lane & 7, WBC_NCHAIN);
  const uint32_t c_qpack = (qt == 0 ? C->chain_pack_body : (qt == 1 ? C->chain_pack_dof : C->chain_pack_ax))[qq];
  const int ch = chain < WBC_NCHAIN ? chain : WBC_NCHAIN;
  cr.body = C->chain_pack_body[ch]; cr.dof = C->chain_pack_dof[ch]; cr.ax = C->chain_pack_ax[ch]; cr.arm = s.ct_arm[ch];
  // ---- loads: the env's state
  const float e_root = ROW(T.root, env, 26)[min(lane, 25)];
  const float e_dof = ROW(T.dof, env, 40)[min(lane, 39)];
  const float e_bp = ROW(T.body_params, env, 20)[min(lane, 19)];
  const float e_motor = ROW(T.motor, env, WBC_NACT)[min(lane, WBC_NACT - 1)];
  const float e_goal = ROW(T.goal, env, 24)[min(lane, 23)];
  const float e_cmd = ROW(T.commands, env, 3)[min(lane, 2)];
  const float e_eps = ROW(T.ep_sums, env, WBC_NREW)[min(lane, WBC_NREW - 1)];
  const float e_met = ROW(T.met_sums, env, WBC_NMETRIC)[min(lane, WBC_NMETRIC - 1)];
  const float e_fric = G(T.friction)[env], e_bxt = G(T.box_timer)[env], e_bxm = G(T.box_mass)[env];
  const int e_eplen = (int)G(T.ep_len)[env];
  // ---- loads: the action and its FIFO
  float a_new = 0.f, a_h[WBC_ADELAY_LEN];
  if (STEP) {
    const int la = min(lane, WBC_NACT - 1);
    a_new = ROW(actions, env, WBC_NACT)[policy_perm18(la)];
    auto ah = ROW(T.act_hist, env, (WBC_ADELAY_LEN * WBC_NACT));
#pragma unroll
    for (int r = 0; r < WBC_ADELAY_LEN - 1; ++r) a_h[r] = ah[(r + 1) * WBC_NACT + la];
  }
  // ---- stores
  if (lane < WBC_NB * 3) s.k_jxyz[lane / 3][lane % 3] = c_jxyz;
  if (lane < WBC_NB) {
    s.k_body[lane] = c_body;
    s.k_gmlo[lane] = make_uint2((uint32_t)c_m1, (uint32_t)c_m2); s.k_gmhi[lane] = make_uint2((uint32_t)(c_m1 >> 32), (uint32_t)(c_m2 >> 32));
  }
  s.k_prk[lane] = c_prk;
  if (lane < WBC_NDOF) s.k_qdlim[lane] = c_qdlim;
  if (lane < (WBC_NCHAIN + 1) * WBC_MAX_DEPTH) (&s.ct_arm[0][0])[lane] = c_arm;
  if (lane < 24) (&s.k_qpack[0][0])[lane] = c_qpack;
  if (lane < 26) (&s.root[0])[lane] = e_root;                    // (root[13], box[13]: contiguous)
  if (lane < 40) { if (lane & 1) s.qd[lane >> 1] = e_dof; else s.q[lane >> 1] = e_dof; }
  if (lane < 20) s.bp[lane] = e_bp;
  if (lane < WBC_NACT) s.motor[lane] = e_motor;
  if (lane < 24) s.goal[lane] = e_goal;
  if (lane < 3) s.cmd[lane] = e_cmd;
  if (lane < WBC_NREW) s.ep_sums[lane] = e_eps;
  if (lane < WBC_NMETRIC) s.met_sums[lane] = e_met;
  if (lane == 0) {
    s.dyn_dirty = 0; s.dropped = 0;
    s.friction = e_fric;
    const float tf = C->cfg.terrain_friction, bf = C->model.box_friction;      // PhysX default combine: the average, not below 0
    s.mu[0] = fmaxf(0.5f * (e_fric + tf), 0.f); s.mu[1] = fmaxf(e_fric, 0.f);
    s.mu[2] = fmaxf(0.5f * (bf + tf), 0.f); s.mu[3] = fmaxf(0.5f * (bf + e_fric), 0.f);
    s.bxtimer = (int)e_bxt;
    const float bh = C->model.box_half;
    s.bxim = 1.f / e_bxm; s.bxiI = 1.f / (e_bxm * (2.f / 3.f) * bh * bh);
    s.ep_len = e_eplen;
  }
  if (STEP && lane < WBC_NACT) {
    // action reorder, clip and delay FIFO (WG:1162-1168)
    const float clipa = C->cfg.clip_actions;
    const float a = clampf(a_new, -clipa, clipa);
    auto ah = ROW(T.act_hist, env, (WBC_ADELAY_LEN * WBC_NACT));
    float used = a;
    if (C->cfg.action_delay != -1) {
      a_h[WBC_ADELAY_LEN - 1] = a;
      const int sel = WBC_ADELAY_LEN - C->cfg.action_delay - 1;
      used = a_h[0];
#pragma unroll
      for (int r = 0; r < WBC_ADELAY_LEN; ++r) { ah[r * WBC_NACT + lane] = a_h[r]; used = (r == sel) ? a_h[r] : used; }
    }
    s.act[lane] = used;
    s.act_last[lane] = a;
  }
  WSYNC();
}

// _compute_torques (oracle: compute_torques): lanes 0..19
__device__ __forceinline__ void torque_pass(Smem& s, CP C) {
  const int j = threadIdx.x;
  if (j < WBC_NACT) {
    const float a_s = s.act[j] * s.motor[j] * C->cfg.action_scale[j];
    float qw = s.q[j];
    if (j == WBC_NACT - 8) qw = wrap_to_pi(qw);
    const float t = C->cfg.p_gains[j] * (a_s + C->cfg.default_dof_pos[j] - qw) - C->cfg.d_gains[j] * s.qd[j];
    const float lim = C->cfg.torque_limits[j];
    s.tau[j] = fminf(fmaxf(t, -lim), lim);
  } else if (j < WBC_NDOF) {
    s.tau[j] = 0.f;
  }
}

// reset_idx for this env (oracle: reset_env); all lanes enter, writes go to LDS
template <class TT> __device__ __forceinline__ void reset_env(Smem& s, const TT& T, CP C, uint64_t seed, int env, uint64_t step, int start, float base_yaw) {
  const int lane = threadIdx.x;
  if (lane < WBC_NDOF) {
    s.q[lane] = C->cfg.default_dof_pos[lane] * rng_range(C->cfg.dof_reset_lo, C->cfg.dof_reset_hi, seed, env, step, SLOT_RESET_DOF + lane);
    s.qd[lane] = 0.f;
  }
  if (lane < WBC_NFEET && (((C->cur.leg_active_mask | C->cur.arm_active_mask) >> WBC_REW_FEET_AIR_TIME) & 1ull))
    ROW(T.feet_air_time, env, WBC_NFEET)[lane] = 0.f;                              // WG:734 (the state only exists while the term is on)
  if (lane < WBC_NREW) { ROW(T.ep_sums_done, env, WBC_NREW)[lane] = s.ep_sums[lane]; }
  if (lane < WBC_NMETRIC) { ROW(T.met_sums_done, env, WBC_NMETRIC)[lane] = s.met_sums[lane]; }
  WSYNC();
  if (lane < WBC_NREW) s.ep_sums[lane] = 0.f;
  if (lane < WBC_NMETRIC) s.met_sums[lane] = 0.f;
  // the reset's draws (slots SLOT_RESET_XY ...: xy 0-1, velocity 2-7, commands 8-9, goal orientation 10-12, goal candidates 13-42) and the goal
  draw_block(s, seed, env, step, SLOT_RESET_XY, SLOT_RESET_GOAL_SPHERE + 30 - SLOT_RESET_XY);
  goal_pick(s, C, SLOT_RESET_GOAL_SPHERE - SLOT_RESET_XY);
  if (lane == 0) {
    const float* dr = DRAWS(s);
    {   // what _update_terrain_curriculum reads of the finished episode (LR:431-435), before root and commands are overwritten
      const float dx = s.root[0] - ROW(T.origins, env, 3)[0], dy = s.root[1] - ROW(T.origins, env, 3)[1];
      ROW(T.reset_travel, env, 2)[0] = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
      ROW(T.reset_travel, env, 2)[1] = __fsqrt_rn(__fadd_rn(__fmul_rn(s.cmd[0], s.cmd[0]), __fmul_rn(s.cmd[1], s.cmd[1])));
    }
    for (int j = 0; j < 13; ++j) s.root[j] = C->cfg.base_init_state[j];
    for (int j = 0; j < 3; ++j) s.root[j] += ROW(T.origins, env, 3)[j];
    for (int j = 0; j < 2; ++j) s.root[j] += urange(-C->cfg.origin_perturb_range, C->cfg.origin_perturb_range, dr[j]);
    s.box[0] = C->cfg.box_origin_x;
    s.box[1] = s.root[1] + G(T.box_dy)[env];
    s.box[2] = C->cfg.box_origin_z;
    for (int j = 0; j < 6; ++j) s.root[7 + j] = urange(-C->cfg.init_vel_perturb_range, C->cfg.init_vel_perturb_range, dr[SLOT_RESET_VEL - SLOT_RESET_XY + j]);
    s.rp[0] = C->init_rp[0]; s.rp[1] = C->init_rp[1];      // the observation of a reset env shows the new pose: roll / pitch of base_init_state (a constant: formed once on the host)
    if (start || s.time_out) resample_commands(s, C, dr[SLOT_RESET_CMD - SLOT_RESET_XY], dr[SLOT_RESET_CMD - SLOT_RESET_XY + 1]);
    goal_apply(s, C, SLOT_RESET_GOAL_ORN - SLOT_RESET_XY, base_yaw);
    s.ep_len = 0;
    s.reset_flag = 1;
    s.goal[G_TIMER] = 0.f;
  }
  WSYNC();
}

// The step kernel's copy stays out of line: 13 % of the waves take it, inlined it costs every wave its registers.
__device__ __attribute__((noinline)) void reset_env_call(Smem& s, const DevTensorsK& T, CP C, uint64_t seed, int env, uint64_t step, float base_yaw) {
  reset_env(s, T, C, seed, env, step, 0, base_yaw);
}

// compute_observations (oracle) + the HBM write-out of the step's results.
template <class TT> __device__ void observe_and_store(Smem& s, const TT& T, CP C, int env, bool was_reset, const StepOut& so,
                                  const float (&hist_in)[12]) {
  const int lane = threadIdx.x;
  const wbc_task_cfg& cf = C->cfg;
  // proprio vector o76: entry `lane` on every lane, entries 64..75 on the first twelve. (Two explicit parts instead of a two-trip loop
  // over one if-chain: every divergent branch of the chain is paid once per trip, and the second trip only has the contact flags, the
  // commands and the goal rows left. The policy order of the DoF entries is arithmetic, not a table load.)
  {
    const int e = lane;                               // 0..63
    float val;
    if (e < 2) val = s.rp[e];
    else if (e < 5) val = s.bav[e - 2] * cf.obs_scale_ang_vel;
    else if (e < 45) {
      const bool isq = e < 25;
      const int pj = isq ? e - 5 : e - 25, sj = pj < WBC_NACT ? policy_perm18(pj) : pj;
      float x = isq ? s.q[sj] : s.qd[sj];
      if (isq && sj == WBC_NDOF - 8) x = wrap_to_pi(x);
      val = isq ? (x - cf.default_dof_pos[sj]) * cf.obs_scale_dof_pos : x * cf.obs_scale_dof_vel;
    } else if (e < 63) val = was_reset ? 0.f : s.act_last[policy_perm18(e - 45)];
    else {
      const float* fs = s.out_sensor[1];              // entry 63: foot 0 reads sensor 1 ([1,0,3,2])
      val = dot6(fs, fs) > 2.25f ? 1.f : 0.f;         // |wrench| > 1.5
    }
    s.post.o76[e] = val;
  }
  if (lane < WBC_NPROP - LANES) {
    const int e = LANES + lane;                       // 64..75
    float val;
    if (e < 67) {
      const float* fs = s.out_sensor[(e - 63) ^ 1];
      val = dot6(fs, fs) > 2.25f ? 1.f : 0.f;
    } else if (e < 70) val = s.cmd[e - 67] * cf.commands_scale[e - 67];
    else val = s.goal[(e < 73 ? (cf.goal_command_cart ? G_CURR_CART : G_CURR) - 70 : G_DORN - 73) + e];      // (curr_ee_goal per command_mode, WG:589-593)
    s.post.o76[e] = val;
  }
  WSYNC();
  // obs_buf = [o76, priv24, old history]; history <- shifted / refilled
  const float clipv = cf.clip_obs;
  auto obs = ROW(so.obs ? so.obs : T.obs, env, WBC_NOBS);
  auto hist = ROW(T.obs_hist, env, (WBC_HIST * WBC_NPROP));
  const bool refill = s.ep_len <= 1;
  // the old history was requested right after the substeps (hist_in: its HBM latency ran under the rigid-body pass and the task
  // logic); every lane's reads were issued long before the first write of the in-place shift below
  float old[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) old[r] = was_reset ? 0.f : hist_in[r];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const int idx = lane + r * LANES;
    if (idx < WBC_HIST * WBC_NPROP) {
      obs[WBC_NPROP + WBC_NPRIV + idx] = clampf(old[r], -clipv, clipv);
      if (refill) hist[idx] = s.post.o76[idx % WBC_NPROP];
      else if (idx >= WBC_NPROP) hist[idx - WBC_NPROP] = old[r];
    }
  }
  if (!refill) for (int e = lane; e < WBC_NPROP; e += LANES) hist[(WBC_HIST - 1) * WBC_NPROP + e] = s.post.o76[e];
  for (int e = lane; e < WBC_NPROP + WBC_NPRIV; e += LANES) {
    float val;
    if (e < WBC_NPROP) val = s.post.o76[e];
    else if (e < WBC_NPROP + 5) val = ROW(T.mass_params, env, 5)[e - WBC_NPROP];
    else if (e == WBC_NPROP + 5) val = s.friction;
    else val = s.motor[e - WBC_NPROP - 6] - 1.f;
    obs[e] = clampf(val, -clipv, clipv);
  }
  // state write-back
  if (lane < 13) { ROW(T.root, env, 26)[lane] = s.root[lane]; ROW(T.root, env, 26)[13 + lane] = s.box[lane]; }
  {   // (a laundered lane index: left alone, the compiler keeps lane & 1 and lane >> 1 of load_env's dof rows alive across the whole
      // kernel for this store -- two VGPRs the substeps cannot spare)
    int l2 = threadIdx.x;
    asm volatile("" : "+v"(l2));
    l2 &= LANES - 1;
    if (l2 < 40) ROW(T.dof, env, 40)[l2] = (l2 & 1) ? s.qd[l2 >> 1] : s.q[l2 >> 1];
  }
  if (lane < WBC_NDOF) { ROW(T.torques, env, WBC_NDOF)[lane] = s.tau[lane]; ROW(T.last_dof_vel, env, WBC_NDOF)[lane] = s.qd[lane]; }
  if (lane < WBC_NACT) { ROW(T.actions, env, WBC_NACT)[lane] = s.act[lane]; ROW(T.last_actions, env, WBC_NACT)[lane] = s.act[lane]; }
  if (lane < 6) ROW(T.last_root_vel, env, 6)[lane] = s.root[7 + lane];
  if (lane < 24) ROW(T.goal, env, 24)[lane] = s.goal[lane];
  if (lane < 3) { ROW(T.commands, env, 3)[lane] = s.cmd[lane]; ROW(T.base_lin_vel, env, 3)[lane] = s.blv[lane]; ROW(T.base_ang_vel, env, 3)[lane] = s.bav[lane]; }
  if (lane < WBC_NREW) ROW(T.ep_sums, env, WBC_NREW)[lane] = s.ep_sums[lane];
  if (lane < WBC_NMETRIC) ROW(T.met_sums, env, WBC_NMETRIC)[lane] = s.met_sums[lane];
  for (int e = lane; e < WBC_NRB_ENV * 3; e += LANES) ROW(T.contact, env, (WBC_NRB_ENV * 3))[e] = (&s.out_contact[0][0])[e];
  if (lane < WBC_NFEET * 6) ROW(T.sensor, env, (WBC_NFEET * 6))[lane] = (&s.out_sensor[0][0])[lane];
  for (int e = lane; e < WBC_NRB_ENV * 13; e += LANES) ROW(T.rb, env, (WBC_NRB_ENV * 13))[e] = (&s.post.out_rb[0][0])[e];
  if (lane == 0) {
    G(T.rew)[env] = s.rew; G(T.arm_rew)[env] = s.arm_rew;
    G(T.reset_buf)[env] = s.reset_flag; G(T.time_out)[env] = (uint8_t)s.time_out; G(T.ep_len)[env] = s.ep_len;
    if (so.rewards) {                                 // the same arithmetic as rollout_store_kernel (csrc/wbc_gae_kernel.hip)
      const float to = s.time_out ? 1.f : 0.f;
      float r0 = s.rew, r1 = s.arm_rew;
      r0 += so.gamma * (ROW(so.values, env, 2)[0] * to); r1 += so.gamma * (ROW(so.values, env, 2)[1] * to);
      ROW(so.rewards, env, 2)[0] = r0; ROW(so.rewards, env, 2)[1] = r1;
      G(so.dones)[env] = (uint8_t)(s.reset_flag != 0);
    }
  }
}

// WidowGo1.step for one env per wave (oracle: env_step). `step` = common_step_counter after increment. so: optional
// extra outputs (observation rows into the rollout storage slot of the next transition, this transition's reward / done slots).
extern "C" __global__ void __launch_bounds__(LANES, 4) wbc_step_kernel(const DevTensors* __restrict__ Tp, const DevConst* __restrict__ Cg, const float* __restrict__ actions,
                                                                    int num_envs, uint64_t seed, uint64_t step, StepOut so, uint32_t deal) {
  CP C = (CP)Cg;
  __shared__ Smem s;
  // (the tensor table likewise: never written by a kernel -> constant address space, its pointers arrive by scalar loads on demand)
  const DevTensorsK& T = *(const DevTensorsK*)Tp;
  // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: XCD x takes the CONTIGUOUS env range
  // [x per, (x + 1) per), so the sub-64-B rows of neighbouring envs (13-float root rows, 3-float commands, ...) meet in one L2 and
  // leave it as whole lines instead of as byte-masked partial writes from two L2s (grid = 8 per workgroups, wbc_sim.hip).
  //
  // WHICH env of its XCD's range a workgroup takes is dealt per launch (deal & 2; N a multiple of 512 from 2048 to 4096: every robot resident
  // at once, 4 per SIMD). A launch ends with its busiest SIMD; left alone it waits for whichever SIMD happens to hold three or four
  // robots in contact (+15..30 % cycles each) while the mean SIMD holds one or two. The dispatcher hands the workgroups of an XCD to
  // its shader engines, their CUs and the CUs' SIMDs round-robin: any 64 consecutive workgroups of an XCD land on 64 different SIMDs,
  // any 128 on >= 114, any 256 on all 128 (measured inside the training loop, where WHICH SIMD a given workgroup gets changes from
  // launch to launch: tools/wave_bench_state.py). So the envs of the range are taken in the order "expected to lie on the ground or
  // to touch itself", "expected in contact", the rest (a stable partition by the two hint bits the previous launch left in deal_flags:
  // what its wave saw in its last substep, cleared by a reset): workgroup j of the XCD takes the j-th env of that order -- rank-select
  // on the range's <= 8 flag words per bit plane. (Running every other window of 128 backwards, so that the SIMDs that got the
  // heaviest robots get the lightest next: measured worse, 117-119 us against 114-116; a column-aware order -- position p and p + 128 taken
  // for the same SIMD, the heaviest class given the lightest mates -- likewise: 117.2 against 111.5 on one box. The plain order it is.) Any flag content
  // gives a bijection (the flags of this launch's parity are not written during it); only the balance depends on the hints. Results
  // do not depend on the deal: nothing but `env` is derived from blockIdx (tests/test_gpu_deal.py: bit for bit).
  const int per = (num_envs + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  int env = xcd * per + (blockIdx.x >> 3);
  if (deal & 2u) {
    const int nw = per >> 6;
    const auto fw = G(T.deal_flags) + (((deal & 1u) * 8u + (uint32_t)xcd) * (2 * WBC_DEAL_WORDS));
    uint32_t pos = blockIdx.x >> 3;
    uint64_t wv = 0ull;
    if ((int)threadIdx.x < 2 * WBC_DEAL_WORDS && (int)(threadIdx.x & (WBC_DEAL_WORDS - 1)) < nw) wv = fw[threadIdx.x];
    uint64_t wv_[WBC_DEAL_WORDS], wh_[WBC_DEAL_WORDS], wl_[WBC_DEAL_WORDS];
    int nV = 0, nH = 0;
#pragma unroll
    for (int i = 0; i < WBC_DEAL_WORDS; ++i) {
      const uint64_t a = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(wv >> 32), i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)wv, i);
      const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(wv >> 32), WBC_DEAL_WORDS + i) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)wv, WBC_DEAL_WORDS + i);
      wv_[i] = b; wh_[i] = a & ~b; wl_[i] = i < nw ? ~(a | b) : 0ull;
      nV += __popcll(b); nH += __popcll(a & ~b);
    }
    const int cls = (int)pos < nV ? 2 : ((int)pos < nV + nH ? 1 : 0);
    int kk = (int)pos - (cls == 2 ? 0 : (cls == 1 ? nV : nV + nH)), idx = 0;
    uint64_t W = 0ull;
    bool found = false;
#pragma unroll
    for (int i = 0; i < WBC_DEAL_WORDS; ++i) {
      const uint64_t wi = cls == 2 ? wv_[i] : (cls == 1 ? wh_[i] : wl_[i]);
      const int c = __popcll(wi);
      if (!found) {
        if (kk < c) { found = true; W = wi; idx = i; }
        else kk -= c;
      }
    }
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(W >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)W, 0u));
    const uint64_t hit = __ballot(((W >> threadIdx.x) & 1ull) != 0ull && (int)below == kk);
    const int ein = __builtin_amdgcn_readfirstlane(idx * 64 + (__ffsll((unsigned long long)hit) - 1));      // (wave-uniform by construction; said so)
    env = xcd * per + ein;
    // (raising the priority of the waves expected in contact from their first instruction on, instead of from their first active
    // contact: measured, no gain -- 115.2 / 113.8 / 115.7 us without, 117.1 / 116.5 / 116.6 with, in the bench loop)
    if (threadIdx.x == 0) {
      // (kept in LDS, not in scalar registers across the kernel) the hints go to the OTHER parity's words
      s.deal_word = (int)((((deal & 1u) ^ 1u) * 8u + (uint32_t)xcd) * (2 * WBC_DEAL_WORDS)) + (ein >> 6);
      s.deal_bit = ein & 63;
    }
  } else if (threadIdx.x == 0) s.deal_word = -1;
  if (env >= num_envs) return;
  const int lane = threadIdx.x;
  const int chain = lane / CH_LANES, k = lane % CH_LANES;
  ChainRegs cr;
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  const long long wave_t0 = clock64();
#endif
  STAMP(11);
  prologue<true>(s, cr, T, C, env, chain, actions);
  const int dec = C->cfg.decimation;
  STAMP(12);
  for (int t = 0; t < dec; ++t) {
    torque_pass(s, C);
    WSYNC();
    physics_substep(s, C, cr, chain, k, t == dec - 1);
  }
  if (lane == 0) {
    G(T.box_timer)[env] = (float)s.bxtimer;
    if (s.dropped != 0) G(T.dropped)[env] += (float)s.dropped;
  }
  STAMP(13);
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  const long long wave_t13 = clock64();
#endif
  // Everything after the substeps reads its tensor / constant pointers through laundered copies of the two kernel arguments: the
  // compiler otherwise hoists those (invariant) scalar loads above the substep loop, where the ~40 pointers and constants it
  // keeps alive across the loop exhaust the SGPR file and end up spilled to scratch memory through VGPRs.
  const DevTensorsK* Tq = (const DevTensorsK*)Tp;
  CP Cq = C;
  int envq = env;                                  // (likewise the env index: left alone, the byte offsets env * stride of a dozen tensor rows
  asm volatile("" : "+s"(Tq), "+s"(Cq), "+s"(envq));   // computed for the prologue's loads are kept for the write-back -- 20 SGPRs spilled through v_writelane / v_readlane)
  const DevTensorsK& T2 = *Tq;
  // the observation's history block (obs_history_buf before this step's update, WG:992): 12 coalesced loads per lane, consumed by
  // observe_and_store at the very end
  float hist_in[12];
  {
    auto hist = ROW(T2.obs_hist, envq, (WBC_HIST * WBC_NPROP));
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const int idx = lane + r * LANES;
      hist_in[r] = (idx < WBC_HIST * WBC_NPROP) ? hist[idx] : 0.f;
    }
  }
  // post_physics_step (WG:865-915)
  float rsc_leg = 0.f, rsc_arm = 0.f;             // this lane's reward scales (consumed after the lane-0 task logic)
  if (lane < WBC_NREW) { rsc_leg = Cq->cur.leg_reward_scale[lane]; rsc_arm = Cq->cur.arm_reward_scale[lane]; }
  rigid_body_pass(s, Cq, cr, chain, k);
  euler_batch(s, &s.root[3], &s.post.out_rb[Cq->model.gripper_rb][3]);      // (read by lane 0 below, after the wavefront fence before its block)
  STAMP(14);
  // check_termination's contact list (WG:940) and _reward_collision (LR:865-867): |net contact force| of every rigid body on its
  // own lane, the two conditions as wavefront ballots
  float cnrm = 0.f;
  if (lane < WBC_NRB) { const f3 f = ld3(s.out_contact[lane]); cnrm = sqrtf(dot(f, f)); }
  const uint32_t rb_bit = lane < WBC_NRB ? 1u << lane : 0u;
  const int c_term = __ballot((Cq->cfg.term_contact_rb_mask & rb_bit) != 0u && cnrm > 1.0f) != 0ull;
  const float ncol = (float)__popcll(__ballot((Cq->cfg.penalize_contact_rb_mask & rb_bit) != 0u && cnrm > 0.1f));
  // the base class's reward terms (zero-scaled in the shipped config): their per-DoF sums, lane-parallel, only when one is on
  const bool base_on = ((Cq->cur.leg_active_mask | Cq->cur.arm_active_mask) >> WBC_NREW_WG) != 0ull;
  BaseSums bsums = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (base_on) bsums = base_reward_sums(s, T2, Cq, envq);
  // this step's random events (wave-uniform conditions, evaluated as lane 0 will find them below): their draws and the new EE goal
  // are prepared by all lanes, lane 0 applies them
  {
    const int pi = Cq->cfg.push_interval;
    const int need_goal = __builtin_amdgcn_readfirstlane((int)((s.goal[G_TIMER] + 1.f) > s.goal[G_TOTAL]));
    const int need_cmd = __builtin_amdgcn_readfirstlane((int)(((s.ep_len + 1) % Cq->cfg.resample_interval) == 0));
    const int need_push = pi > 0 && (step % (uint64_t)pi) == 0;
    if (need_goal | need_cmd | need_push) {
      draw_block(s, seed, envq, step, SLOT_GOAL_ORN, SLOT_PUSH + 2);
      if (need_goal) goal_pick(s, Cq, SLOT_GOAL_SPHERE);
    }
  }
  float base_yaw = 0.f;
  WSYNC();
  if (lane == 0) {
    s.ep_len += 1;
    st3(s.blv, quat_rotate_inverse(&s.root[3], ld3(&s.root[7])));
    st3(s.bav, quat_rotate_inverse(&s.root[3], ld3(&s.root[10])));
    const f3 rpy = ld3(&s.eul[0]);                  // euler_from_quat(root), computed by euler_batch
    base_yaw = rpy.z;
    s.base_yaw = base_yaw;
    s.rp[0] = rpy.x; s.rp[1] = rpy.y;
    float sy, cy;
    fast_sincosf(0.5f * base_yaw, &sy, &cy);
    const float yq[4] = {0.f, 0.f, sy, cy};
    // update_curr_ee_goal
    const float tt = clampf(s.goal[G_TIMER] * rcpf(s.goal[G_TRAJ]), 0.f, 1.f);
    for (int j = 0; j < 3; ++j) s.goal[G_CURR + j] = lerp_torch(s.goal[G_START + j], s.goal[G_GOAL + j], tt);
    st3(&s.goal[G_CURR_CART], sphere2cart(ld3(&s.goal[G_CURR])));
    s.goal[G_TIMER] += 1.f;
    if (s.goal[G_TIMER] > s.goal[G_TOTAL]) goal_apply(s, Cq, SLOT_GOAL_ORN, base_yaw);
    if (s.ep_len % Cq->cfg.resample_interval == 0) resample_commands(s, Cq, DRAWS(s)[SLOT_CMD], DRAWS(s)[SLOT_CMD + 1]);
    if (Cq->cfg.push_interval > 0 && (step % (uint64_t)Cq->cfg.push_interval) == 0) {
      const float px = urange(-Cq->cfg.max_push_vel, Cq->cfg.max_push_vel, DRAWS(s)[SLOT_PUSH]);
      const float py = urange(-Cq->cfg.max_push_vel, Cq->cfg.max_push_vel, DRAWS(s)[SLOT_PUSH + 1]);
      const float kk = ((s.cmd[0] + s.cmd[1] + s.cmd[2]) == 0.f) ? 2.5f : 1.f;
      s.root[7] = px * kk; s.root[8] = py * kk;
    }
    const float r = rpy.x, p = rpy.y, z = s.root[2], th = Cq->cfg.term_rp_threshold;
    const int gc = Cq->cfg.goal_command_cart ? G_CURR_CART : G_CURR;                        // curr_ee_goal (WG:589-593)
    const int r_term = ((r > th) && (s.goal[gc + 2] >= 0.f)) || ((r < -th) && (s.goal[gc + 2] <= 0.f));
    const int p_term = ((p > th) && (s.goal[gc + 1] >= 0.f)) || ((p < -th) && (s.goal[gc + 1] <= 0.f));
    const int z_term = z < Cq->cfg.term_z_threshold;
    s.time_out = s.ep_len > Cq->cfg.max_episode_length;
    s.reset_flag = c_term | r_term | p_term | z_term | s.time_out;
    compute_reward(s, T2, Cq, yq, ncol, base_on, bsums, envq);
  }
  if (!base_on && lane >= WBC_NREW_WG && lane < WBC_NREW) s.term[lane] = 0.f;
  WSYNC();
  reward_accumulate(s, Cq, rsc_leg, rsc_arm);
  WSYNC();
  STAMP(15);
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  const long long wave_t15 = clock64();
#endif
  const bool do_reset = s.reset_flag != 0;
  if (do_reset) { __builtin_amdgcn_s_setprio(2); reset_env_call(s, T2, Cq, seed, envq, step, s.base_yaw); }
  if (do_reset && lane < WBC_ADELAY_LEN * WBC_NACT) {   // action_history_buf[env_ids] = 0 (WG:738)
    ROW(T2.act_hist, envq, (WBC_ADELAY_LEN * WBC_NACT))[lane] = 0.f;
    if (lane + LANES < WBC_ADELAY_LEN * WBC_NACT) ROW(T2.act_hist, envq, (WBC_ADELAY_LEN * WBC_NACT))[lane + LANES] = 0.f;
  }
  STAMP(16);
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  const long long wave_t16 = clock64();
#endif
  observe_and_store(s, T2, Cq, envq, do_reset, so, hist_in);
  STAMP(17);
  if (lane == 0 && s.deal_word >= 0) {
    // the next launch's deal: this envq's two hint bits (fire-and-forget atomics: 64 envs share a word)
    auto fo = G(T2.deal_flags) + s.deal_word;
    const uint64_t bit = 1ull << s.deal_bit;
    const int hint = do_reset ? 0 : s.deal_hint;
    if (hint >= 1) __hip_atomic_fetch_or(fo, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_and(fo, ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hint >= 2) __hip_atomic_fetch_or(fo + WBC_DEAL_WORDS, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_and(fo + WBC_DEAL_WORDS, ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#if defined(WBC_STEP_TIMING) || defined(WBC_WAVE_TIMING)
  if (g_wave_dbg && lane == 0) {
    g_wave_dbg[6 * (size_t)envq] = wave_t0; g_wave_dbg[6 * (size_t)envq + 1] = clock64();
    g_wave_dbg[6 * (size_t)envq + 3] = wave_t13; g_wave_dbg[6 * (size_t)envq + 4] = wave_t15; g_wave_dbg[6 * (size_t)envq + 5] = wave_t16;
    g_wave_dbg[6 * (size_t)envq + 2] = (long long)(do_reset ? 1 : 0) | (long long)(s.deal_hint != 0 ? 2 : 0) | (long long)(s.deal_hint == 2 ? 4 : 0) | ((long long)(s.dbg_ncon & 0xFFFF) << 8) | ((long long)((s.dbg_ncon >> 16) & 15) << 28) | ((long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 15) << 24) | ((long long)((unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11)) & 0xFFFFu) << 32) | ((long long)(blockIdx.x & 0xFFFFu) << 48);
  }
#endif
}

// reset_idx(all envs, start=True) (BT:129): one wave per env
extern "C" __global__ void __launch_bounds__(LANES) wbc_reset_kernel(DevTensors T, const DevConst* __restrict__ Cg, int num_envs, uint64_t seed, uint64_t step) {
  CP C = (CP)Cg;
  __shared__ Smem s;
  const int env = blockIdx.x;
  if (env >= num_envs) return;
  const int lane = threadIdx.x;
  const int chain = lane / CH_LANES, k = lane % CH_LANES;
  ChainRegs cr;
  prologue<false>(s, cr, T, C, env, chain, nullptr);
  if (lane == 0) { s.time_out = 0; s.reset_flag = 0; }
  WSYNC();
  float yaw = 0.f;
  if (lane == 0) { yaw = euler_from_quat(&s.root[3]).z; s.base_yaw = yaw; }
  WSYNC();
  reset_env(s, T, C, seed, env, step, 1, s.base_yaw);
  rigid_body_pass(s, C, cr, chain, k);
  // write back what a reset touches
  if (lane < 13) { ROW(T.root, env, 26)[lane] = s.root[lane]; ROW(T.root, env, 26)[13 + lane] = s.box[lane]; }
  {   // (a laundered lane index: left alone, the compiler keeps lane & 1 and lane >> 1 of load_env's dof rows alive across the whole
      // kernel for this store -- two VGPRs the substeps cannot spare)
    int l2 = threadIdx.x;
    asm volatile("" : "+v"(l2));
    l2 &= LANES - 1;
    if (l2 < 40) ROW(T.dof, env, 40)[l2] = (l2 & 1) ? s.qd[l2 >> 1] : s.q[l2 >> 1];
  }
  if (lane < 24) ROW(T.goal, env, 24)[lane] = s.goal[lane];
  if (lane < 3) ROW(T.commands, env, 3)[lane] = s.cmd[lane];
  if (lane < WBC_NREW) ROW(T.ep_sums, env, WBC_NREW)[lane] = 0.f;
  if (lane < WBC_NMETRIC) ROW(T.met_sums, env, WBC_NMETRIC)[lane] = 0.f;
  if (lane < WBC_NACT) ROW(T.last_actions, env, WBC_NACT)[lane] = 0.f;
  if (lane < WBC_NDOF) ROW(T.last_dof_vel, env, WBC_NDOF)[lane] = 0.f;
  for (int e = lane; e < WBC_HIST * WBC_NPROP; e += LANES) ROW(T.obs_hist, env, (WBC_HIST * WBC_NPROP))[e] = 0.f;
  for (int e = lane; e < WBC_ADELAY_LEN * WBC_NACT; e += LANES) ROW(T.act_hist, env, (WBC_ADELAY_LEN * WBC_NACT))[e] = 0.f;
  for (int e = lane; e < WBC_NRB_ENV * 13; e += LANES) ROW(T.rb, env, (WBC_NRB_ENV * 13))[e] = (&s.post.out_rb[0][0])[e];
  if (lane == 0) { G(T.reset_buf)[env] = 1; G(T.ep_len)[env] = 0; }
}

// gym.simulate: one substep with the torques tensor as set (no PD, no post-physics)
extern "C" __global__ void __launch_bounds__(LANES) wbc_simulate_kernel(DevTensors T, const DevConst* __restrict__ Cg, int num_envs) {
  CP C = (CP)Cg;
  __shared__ Smem s;
  const int env = blockIdx.x;
  if (env >= num_envs) return;
  const int lane = threadIdx.x;
  const int chain = lane / CH_LANES, k = lane % CH_LANES;
  ChainRegs cr;
  prologue<false>(s, cr, T, C, env, chain, nullptr);
  if (lane < WBC_NDOF) s.tau[lane] = ROW(T.torques, env, WBC_NDOF)[lane];
  WSYNC();
  physics_substep(s, C, cr, chain, k, true);
  if (lane < 13) { ROW(T.root, env, 26)[lane] = s.root[lane]; ROW(T.root, env, 26)[13 + lane] = s.box[lane]; }
  if (lane == 0) {
    G(T.box_timer)[env] = (float)s.bxtimer;
    if (s.dropped != 0) G(T.dropped)[env] += (float)s.dropped;
  }
  {   // (a laundered lane index: left alone, the compiler keeps lane & 1 and lane >> 1 of load_env's dof rows alive across the whole
      // kernel for this store -- two VGPRs the substeps cannot spare)
    int l2 = threadIdx.x;
    asm volatile("" : "+v"(l2));
    l2 &= LANES - 1;
    if (l2 < 40) ROW(T.dof, env, 40)[l2] = (l2 & 1) ? s.qd[l2 >> 1] : s.q[l2 >> 1];
  }
  for (int e = lane; e < WBC_NRB_ENV * 3; e += LANES) ROW(T.contact, env, (WBC_NRB_ENV * 3))[e] = (&s.out_contact[0][0])[e];
  if (lane < WBC_NFEET * 6) ROW(T.sensor, env, (WBC_NFEET * 6))[lane] = (&s.out_sensor[0][0])[lane];
}

// gym.refresh_rigid_body_state_tensor after a state write: forward kinematics only
extern "C" __global__ void __launch_bounds__(LANES) wbc_fk_kernel(DevTensors T, const DevConst* __restrict__ Cg, int num_envs) {
  CP C = (CP)Cg;
  __shared__ Smem s;
  const int env = blockIdx.x;
  if (env >= num_envs) return;
  const int lane = threadIdx.x;
  const int chain = lane / CH_LANES, k = lane % CH_LANES;
  ChainRegs cr;
  prologue<false>(s, cr, T, C, env, chain, nullptr);
  WSYNC();
  rigid_body_pass(s, C, cr, chain, k);
  for (int e = lane; e < WBC_NRB_ENV * 13; e += LANES) ROW(T.rb, env, (WBC_NRB_ENV * 13))[e] = (&s.post.out_rb[0][0])[e];
}

static_assert(sizeof(PostBuf) <= sizeof(float) * WBC_NB * 36, "post-physics staging must fit in the IA region");
static_assert(offsetof(Smem, vb) == offsetof(Smem, wb) + 12 && offsetof(Smem, gF) == offsetof(Smem, wb) + 24, "wb, vb, gF are written as one 9-float row");
static_assert(sizeof(float) * (36 + WBC_NCP * 3) <= sizeof(float) * WBC_NB * 36, "per-contact iteration data must fit in the IA region");
static_assert(sizeof(float) * (WBC_NRB_ENV * 3 + WBC_NFEET * 6) <= sizeof(float) * WBC_NB * 6, "contact outputs alias U");
static_assert(sizeof(Smem) <= 10240, "16 robots per CU (160 KB of LDS): all 4096 envs of the bench resident at once");
static_assert(offsetof(Smem, k_jxyz) + 32 * 16 <= sizeof(Smem), "a 'none' body index (31) must stay inside the workgroup's LDS where the passes read through it");

extern "C" void wbc_debug_set_wave_timing(void* dev_buf) {
  long long* p = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_dbg), &p, sizeof(p));
}

extern "C" void wbc_debug_set_phase_exit(int stamp) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_exit), &stamp, sizeof(stamp)); }

extern "C" void wbc_debug_set_step_timing(void* dev_buf) {
  long long* p = (long long*)dev_buf;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_dbg), &p, sizeof(p));
}
