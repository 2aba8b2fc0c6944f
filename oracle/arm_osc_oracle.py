"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, fp64) restatement of the quantities Isaac Gym's mass-matrix and Jacobian
tensors supply to the torque-supervision path of the reference (legged_gym/envs/widowGo1/widowGo1.py:550-558,
1201-1242), used to check csrc/wbc_arm_kernel.hip. Nothing under wbc_amd imports this file.

PARITY UNPINNED against the reference itself (the tensors come from the closed-source simulator). The restatement is
deliberately built with DIFFERENT algebra than the kernel so that agreement means something:
  * mass matrix: sum over the arm's bodies of  m Jv^T Jv + Jw^T (R I R^T) Jw  with centre-of-mass Jacobians
    (the kernel: composite spatial inertias about the base origin, M_ij = S_i^T Ic_j S_j);
  * gravity torques: central finite differences of the potential  U(q) = sum_k m_k 9.81 z_origin_k(q)
    (the kernel: sum of J_k^T f_k);
  * end-effector Jacobian: analytic here and in the kernel; tests/test_arm_osc.py checks this one against finite
    differences of the forward kinematics below.
"""
import numpy as np


def quat_to_mat(q):            # xyzw
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_axis(ax, q):
    c, s = np.cos(q), np.sin(q)
    R = np.eye(3)
    a1, a2 = (ax + 1) % 3, (ax + 2) % 3
    R[a1, a1], R[a1, a2], R[a2, a1], R[a2, a2] = c, -s, s, c
    return R


def fk(model, root_pos, root_quat, q):
    """World rotation [nb,3,3] and origin [nb,3] of every moving body (URDF joint rpy are all zero)."""
    nb = model.nb
    R = np.zeros((nb, 3, 3)); p = np.zeros((nb, 3))
    R[0], p[0] = quat_to_mat(root_quat), np.asarray(root_pos, dtype=np.float64)
    for i in range(1, nb):
        par = model.parent[i]
        p[i] = p[par] + R[par] @ model.joint_xyz[i]
        R[i] = R[par] @ rot_axis(model.axis[i], q[model.body_dof[i]])
    return R, p


def arm_chain(model, gripper_body):
    chain, b = [], gripper_body
    while b != 0:
        chain.append(b)
        b = model.parent[b]
    return chain[::-1]


def _sym(I6):
    return np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])


def arm_quantities(model, root_pos, root_quat, q, ee_rb, link_rb, link_mass, gripper_params=None):
    """Returns (mm [6,6], ee Jacobian [6,6] rows linear/angular world frame, gravity torques [6]).
    gripper_params = (mass, com[3], I6[6]) of the randomised gripper body (None: the model's nominal composite)."""
    q = np.asarray(q, dtype=np.float64)
    gb = model.rb_body[ee_rb]
    chain = arm_chain(model, gb)
    assert len(chain) == 6
    R, p = fk(model, root_pos, root_quat, q)
    axes = [R[b][:, model.axis[b]] for b in chain]
    # mass matrix from centre-of-mass Jacobians
    M = np.zeros((6, 6))
    for bi, b in enumerate(chain):
        if b == gb and gripper_params is not None:
            m, com, I6 = gripper_params
        else:
            m, com, I6 = model.mass[b], model.com[b], model.inertia[b]
        c = p[b] + R[b] @ np.asarray(com)
        Jv, Jw = np.zeros((3, 6)), np.zeros((3, 6))
        for j in range(bi + 1):
            Jv[:, j] = np.cross(axes[j], c - p[chain[j]])
            Jw[:, j] = axes[j]
        Iw = R[b] @ _sym(I6) @ R[b].T
        M += m * Jv.T @ Jv + Jw.T @ Iw @ Jw
    # end-effector Jacobian
    pe = p[gb] + R[gb] @ model.rb_offset[ee_rb]
    J = np.zeros((6, 6))
    for j in range(6):
        J[:3, j] = np.cross(axes[j], pe - p[chain[j]])
        J[3:, j] = axes[j]

    # gravity torques: numeric gradient of the potential of the link ORIGINS (what the reference's Jacobians are about)
    def U(qq):
        Rq, pq = fk(model, root_pos, root_quat, qq)
        return sum(mk * 9.81 * (pq[model.rb_body[rb]] + Rq[model.rb_body[rb]] @ model.rb_offset[rb])[2] for rb, mk in zip(link_rb, link_mass))
    g = np.zeros(6)
    h = 1e-6
    for j, b in enumerate(chain):
        dq = np.zeros_like(q); dq[model.body_dof[b]] = h
        g[j] = (U(q + dq) - U(q - dq)) / (2 * h)
    return M, J, g


def ee_pose(model, root_pos, root_quat, q, ee_rb):
    R, p = fk(model, root_pos, root_quat, np.asarray(q, dtype=np.float64))
    gb = model.rb_body[ee_rb]
    return p[gb] + R[gb] @ model.rb_offset[ee_rb], R[gb]
