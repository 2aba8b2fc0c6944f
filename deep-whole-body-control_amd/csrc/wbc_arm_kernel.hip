// wbc_arm_kernel.hip -- the quantities Isaac Gym hands to the torque-supervision path:
// reference widowGo1.py:550-558 wraps gym.acquire_mass_matrix_tensor / acquire_jacobian_tensor and uses
//   mm        = mass_matrix[:, -8:-2, -8:-2]          the arm's 6x6 joint-space inertia block          (WG:558)
//   ee_j_eef  = jacobian[:, gripper_idx, :6, -8:-2]    world-frame [linear; angular] Jacobian of the EE (WG:557)
//   g_torque  = sum over the last 9 rigid bodies of J_body^T (0, 0, 9.81 m_body, 0, 0, 0), arm columns  (WG:1201-1207)
// in get_arm_ee_control_torques (WG:1217-1242). Here they come from the same state tensors the step kernel keeps:
// one thread per env walks the 6-joint arm chain (forward kinematics in the base frame F, composite spatial inertias
// from the tip inwards, M_ij = S_i^T Ic_j S_j) -- an optional, once-per-step pre-pass (torque_supervision=False as shipped).
#include "wbc_device.h"
#include "wbc_stream_guard.h"

#define ARM_N 6
#define ARM_NLINK 9          // rigid bodies whose weight the reference compensates: the last 9 of the actor

struct ArmConst {
  int body[ARM_N], ax[ARM_N], dof[ARM_N];      // arm chain, root outwards
  int gripper_body_depth;                      // depth of the (randomised) gripper body in the chain, -1 if none
  float joint_xyz[ARM_N][3], mass[ARM_N], com[ARM_N][3], inertia[ARM_N][6];
  int ee_depth; float ee_off[3];               // EE rigid body: chain depth of its moving body (-1: not on the arm), offset in it
  int link_depth[ARM_NLINK]; float link_off[ARM_NLINK][3], link_mass[ARM_NLINK];
  int gripper_link;                            // index in the 9 links of the randomised rigid body (-1 none)
};

extern "C" __global__ void __launch_bounds__(64) wbc_arm_dynamics_kernel(ArmConst A, const float* __restrict__ root, const float* __restrict__ dofs,
                                                                        const float* __restrict__ body_params, const float* __restrict__ mass_params,
                                                                        int n, float* __restrict__ mm, float* __restrict__ jac,
                                                                        float* __restrict__ gtorque) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  float R[9];
  quat_to_mat(root + (size_t)env * 26 + 3, R);
  const f3 rp = ld3(root + (size_t)env * 26);
  // forward kinematics of the chain in F (base frame): E_d (columns = body axes), origin pos_d
  float E[ARM_N][9];
  f3 pos[ARM_N];
  {
    float Ep[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    f3 pp = mk3(0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d < ARM_N; ++d) {
      const float q = dofs[(size_t)env * 40 + 2 * A.dof[d]];
      float sq, cq;
      sincosf(q, &sq, &cq);
      const int ax = A.ax[d], a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
      pos[d] = pp + mat_mul(Ep, mk3(A.joint_xyz[d][0], A.joint_xyz[d][1], A.joint_xyz[d][2]));
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float e0 = Ep[r * 3 + ax], e1 = Ep[r * 3 + a1], e2 = Ep[r * 3 + a2];
        E[d][r * 3 + ax] = e0;
        E[d][r * 3 + a1] = cq * e1 + sq * e2;
        E[d][r * 3 + a2] = -sq * e1 + cq * e2;
      }
#pragma unroll
      for (int e = 0; e < 9; ++e) Ep[e] = E[d][e];
      pp = pos[d];
    }
  }
  // joint axes in F; levers are formed in F (base-relative, metres) and rotated to the world afterwards, so that no
  // world-scale coordinate (envs sit at |y| up to 115 m) enters a difference
  f3 axF[ARM_N];
#pragma unroll
  for (int d = 0; d < ARM_N; ++d) {
    const int ax = A.ax[d];
    axF[d] = mk3(E[d][ax], E[d][3 + ax], E[d][6 + ax]);
  }
  (void)rp;
  // EE Jacobian, world frame, rows [linear; angular], columns = arm joints
  {
    f3 pe = mk3(0.f, 0.f, 0.f);
    if (A.ee_depth >= 0) pe = pos[A.ee_depth] + mat_mul(E[A.ee_depth], mk3(A.ee_off[0], A.ee_off[1], A.ee_off[2]));
    float* J = jac + (size_t)env * 36;
#pragma unroll
    for (int d = 0; d < ARM_N; ++d) {
      const bool moves = A.ee_depth >= d;
      const f3 lin = moves ? mat_mul(R, cross(axF[d], pe - pos[d])) : mk3(0.f, 0.f, 0.f);
      const f3 ang = moves ? mat_mul(R, axF[d]) : mk3(0.f, 0.f, 0.f);
      J[0 * 6 + d] = lin.x; J[1 * 6 + d] = lin.y; J[2 * 6 + d] = lin.z;
      J[3 * 6 + d] = ang.x; J[4 * 6 + d] = ang.y; J[5 * 6 + d] = ang.z;
    }
  }
  // gravity compensation as the reference computes it: link ORIGINS (not centres of mass), the last 9 rigid bodies;
  // the link masses are those of env 0 after its randomisation (WG:664-670 reads env 0's properties: quirk kept)
  {
    float g[ARM_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < ARM_NLINK; ++k) {
      const int dk = A.link_depth[k];
      if (dk < 0) continue;
      float mk = A.link_mass[k];
      if (k == A.gripper_link) mk += mass_params[4];
      const f3 pk = pos[dk] + mat_mul(E[dk], mk3(A.link_off[k][0], A.link_off[k][1], A.link_off[k][2]));
#pragma unroll
      for (int d = 0; d < ARM_N; ++d)
        if (d <= dk) g[d] += mk * 9.81f * mat_mul(R, cross(axF[d], pk - pos[d])).z;
    }
#pragma unroll
    for (int d = 0; d < ARM_N; ++d) gtorque[(size_t)env * ARM_N + d] = g[d];
  }
  // joint-space inertia block: composite spatial inertias in F about F's origin, from the tip inwards
  {
    float Ic[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) Ic[e] = 0.f;
    float S[ARM_N][6];
#pragma unroll
    for (int d = 0; d < ARM_N; ++d) {
      const f3 l = cross(pos[d], axF[d]);
      S[d][0] = axF[d].x; S[d][1] = axF[d].y; S[d][2] = axF[d].z; S[d][3] = l.x; S[d][4] = l.y; S[d][5] = l.z;
    }
    float* M = mm + (size_t)env * 36;
#pragma unroll
    for (int d = ARM_N - 1; d >= 0; --d) {
      float m = A.mass[d], com[3], I6[6];
#pragma unroll
      for (int j = 0; j < 3; ++j) com[j] = A.com[d][j];
#pragma unroll
      for (int j = 0; j < 6; ++j) I6[j] = A.inertia[d][j];
      if (d == A.gripper_body_depth) {             // per-env randomised gripper body (body_params[10:20])
        const float* bp = body_params + (size_t)env * 20 + 10;
        m = bp[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) com[j] = bp[1 + j];
#pragma unroll
        for (int j = 0; j < 6; ++j) I6[j] = bp[4 + j];
      }
      const float* Ed = E[d];
      const f3 Cc = pos[d] + mat_mul(Ed, mk3(com[0], com[1], com[2]));
      const float Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
      float EI[9], Ibar[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) EI[r * 3 + c] = Ed[r * 3] * Ib[c] + Ed[r * 3 + 1] * Ib[3 + c] + Ed[r * 3 + 2] * Ib[6 + c];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Ibar[r * 3 + c] = EI[r * 3] * Ed[c * 3] + EI[r * 3 + 1] * Ed[c * 3 + 1] + EI[r * 3 + 2] * Ed[c * 3 + 2];
      const float CC = dot(Cc, Cc);
      const float Cv[3] = {Cc.x, Cc.y, Cc.z};
      const float Cx[9] = {0.f, -Cc.z, Cc.y, Cc.z, 0.f, -Cc.x, -Cc.y, Cc.x, 0.f};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          Ic[r * 6 + c] += Ibar[r * 3 + c] + m * ((r == c ? CC : 0.f) - Cv[r] * Cv[c]);
          Ic[r * 6 + 3 + c] += m * Cx[r * 3 + c];
          Ic[(3 + r) * 6 + c] += m * Cx[c * 3 + r];
          Ic[(3 + r) * 6 + 3 + c] += (r == c) ? m : 0.f;
        }
      float IS[6];                                 // Ic_d S_d
#pragma unroll
      for (int r = 0; r < 6; ++r) IS[r] = dot6(&Ic[r * 6], S[d]);
#pragma unroll
      for (int i = 0; i <= d; ++i) {
        const float v = dot6(S[i], IS);
        M[i * 6 + d] = v; M[d * 6 + i] = v;
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------
struct wbc_sim;
extern "C" int wbc_sim_internal_arm_inputs(wbc_sim* s, const DevConst** hc, const float** root, const float** dofs, const float** body_params,
                                           const float** mass_params, int* n);

// link_mass9: masses of the actor's last 9 rigid bodies (host pointer). Outputs (device): mm f32 [N,6,6], jac f32 [N,6,6]
// (rows linear xyz then angular xyz, world frame), gtorque f32 [N,6].
extern "C" int wbc_sim_arm_dynamics(wbc_sim* s, const int* link_rb9, const float* link_mass9, float* mm, float* jac, float* gtorque, void* stream) {
  StreamDeviceGuard sdg(stream);
  const DevConst* hc; const float *root, *dofs, *bp, *mp; int n;
  if (!s || !link_rb9 || !link_mass9 || !mm || !jac || !gtorque) return -1;
  if (wbc_sim_internal_arm_inputs(s, &hc, &root, &dofs, &bp, &mp, &n) != 0) return -1;
  const wbc_model& m = hc->model;
  ArmConst A;
  int arm = -1;
  for (int c = 0; c < WBC_NCHAIN; ++c) if (hc->chain_len[c] == ARM_N) arm = c;
  if (arm < 0) return -3;
  A.gripper_body_depth = -1;
  for (int d = 0; d < ARM_N; ++d) {
    const int b = hc->chain_body[arm][d];
    A.body[d] = b; A.ax[d] = m.axis[b]; A.dof[d] = m.dof[b]; A.mass[d] = m.mass[b];
    for (int j = 0; j < 3; ++j) { A.joint_xyz[d][j] = m.joint_xyz[b][j]; A.com[d][j] = m.com[b][j]; }
    for (int j = 0; j < 6; ++j) A.inertia[d][j] = m.inertia[b][j];
    if (b == m.gripper_body) A.gripper_body_depth = d;
  }
  auto depth_of = [&](int body) { for (int d = 0; d < ARM_N; ++d) if (A.body[d] == body) return d; return -1; };
  A.ee_depth = depth_of(m.rb_body[m.gripper_rb]);
  for (int j = 0; j < 3; ++j) A.ee_off[j] = m.rb_offset[m.gripper_rb][j];
  A.gripper_link = -1;
  for (int k = 0; k < ARM_NLINK; ++k) {
    const int rb = link_rb9[k];
    if (rb < 0 || rb >= WBC_NRB) return -1;
    A.link_depth[k] = depth_of(m.rb_body[rb]);
    for (int j = 0; j < 3; ++j) A.link_off[k][j] = m.rb_offset[rb][j];
    A.link_mass[k] = link_mass9[k];
    if (rb == m.gripper_rb) A.gripper_link = k;
  }
  hipLaunchKernelGGL(wbc_arm_dynamics_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, A, root, dofs, bp, mp, n, mm, jac, gtorque);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
