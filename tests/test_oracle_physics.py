"""Pins the physics half of the oracle (oracle/wbc_oracle.c), for which no reference output
exists (Isaac Gym is closed source and absent: SURVEY.md section 8c), against formulations that
share no code or algebra with it:

  * Kane's equations evaluated with world-frame kinematics and finite-difference partial
    velocities (no spatial vectors, no recursion) must be satisfied by the accelerations the
    articulated-body algorithm returns;
  * momentum / free-fall / static-equilibrium identities.
"""
import copy

import numpy as np
import pytest

from oracle import OracleSim, default_curriculum
from wbc_amd import abi

G = np.array([0.0, 0.0, -9.81])


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_axis(ax, a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3)
    i, j = (ax + 1) % 3, (ax + 2) % 3
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def body_twists(model, body_params, pos, quat, q, v, w, qd):
    """World-frame kinematics of every moving body: rotation, com position, com velocity, omega."""
    nb = model.nb
    Rb = quat_to_mat(quat / np.linalg.norm(quat))
    R, p, om, vo = [None] * nb, [None] * nb, [None] * nb, [None] * nb
    R[0], p[0], om[0], vo[0] = Rb, pos.copy(), w.copy(), v.copy()
    for i in range(1, nb):
        pa, ax, d = model.parent[i], model.axis[i], model.body_dof[i]
        p[i] = p[pa] + R[pa] @ model.joint_xyz[i]
        R[i] = R[pa] @ rot_axis(ax, q[d])
        om[i] = om[pa] + R[pa][:, ax] * qd[d]
        vo[i] = vo[pa] + np.cross(om[pa], p[i] - p[pa])
    out = []
    for i in range(nb):
        m, com, I6 = body_params[i]
        c = p[i] + R[i] @ com
        vc = vo[i] + np.cross(om[i], c - p[i])
        Ib = np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])
        out.append((m, R[i] @ Ib @ R[i].T, c, vc, om[i]))
    return out


def kane_residual(model, body_params, state, acc, tau):
    """Generalised active + inertia forces for each of the 6 + 18 generalised speeds."""
    pos, quat, q, v, w, qd = state
    a_lin, a_ang, qdd = acc
    nb = model.nb

    def twists(pos_, quat_, q_, v_, w_, qd_):
        return body_twists(model, body_params, pos_, quat_, q_, v_, w_, qd_)

    eps = 1e-6
    # time derivative of each body's (vc, omega) along the motion, by central differences
    def advance(h):
        wq = np.array([w[0], w[1], w[2], 0.0])
        x, y, z, ww = quat
        dq = 0.5 * np.array([wq[3] * x + wq[0] * ww + wq[1] * z - wq[2] * y,
                             wq[3] * y - wq[0] * z + wq[1] * ww + wq[2] * x,
                             wq[3] * z + wq[0] * y - wq[1] * x + wq[2] * ww,
                             wq[3] * ww - wq[0] * x - wq[1] * y - wq[2] * z])
        return twists(pos + h * v, quat + h * dq, q + h * qd, v + h * a_lin, w + h * a_ang, qd + h * qdd)
    tp, tm, t0 = advance(eps), advance(-eps), twists(pos, quat, q, v, w, qd)
    res = np.zeros(6 + model.num_dofs)
    gen_tau = np.zeros(6 + model.num_dofs)
    gen_tau[6:] = tau
    for k in range(6 + model.num_dofs):
        dv, dw, dqd = np.zeros(3), np.zeros(3), np.zeros(model.num_dofs)
        if k < 3:
            dv[k] = 1
        elif k < 6:
            dw[k - 3] = 1
        else:
            dqd[k - 6] = 1
        up = twists(pos, quat, q, v + eps * dv, w + eps * dw, qd + eps * dqd)
        um = twists(pos, quat, q, v - eps * dv, w - eps * dw, qd - eps * dqd)
        tot = 0.0
        for i in range(nb):
            m, I, c, vc, om = t0[i]
            acc_c = (tp[i][3] - tm[i][3]) / (2 * eps)
            alpha = (tp[i][4] - tm[i][4]) / (2 * eps)
            pv = (up[i][3] - um[i][3]) / (2 * eps)      # partial velocity of the com
            pw = (up[i][4] - um[i][4]) / (2 * eps)      # partial angular velocity
            Fstar = m * (acc_c - G)
            Tstar = I @ alpha + np.cross(om, I @ om)
            tot += Fstar @ pv + Tstar @ pw
        res[k] = tot - gen_tau[k]
    return res


def make_params(model, body_params_env):
    bp = []
    for i in range(model.nb):
        bp.append((model.mass[i], model.com[i], model.inertia[i]))
    bp[0] = (body_params_env[0], body_params_env[1:4], body_params_env[4:10])
    g = model.gripper_piece["body"]
    bp[g] = (body_params_env[10], body_params_env[11:14], body_params_env[14:20])
    return bp


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_aba_satisfies_kanes_equations(robot, seed):
    rng = np.random.default_rng(seed)
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    for j in range(18):
        tc.joint_armature[j] = 0.0      # the implicit-PD armature is a modelling term, not rigid-body dynamics
    o = OracleSim(robot["wmodel"], tc, 1)
    lo = np.where(model.dof_lower < model.dof_upper, model.dof_lower, -2.0)
    hi = np.where(model.dof_lower < model.dof_upper, model.dof_upper, 2.0)
    locked = np.array(model.dof_locked)
    lo, hi = np.where(locked, -1.0, lo), np.where(locked, 1.0, hi)
    q = rng.uniform(lo + 0.05, hi - 0.05)
    qd = rng.uniform(-3, 3, size=20)
    q[locked], qd[locked] = 0.0, 0.0
    tau = rng.uniform(-10, 10, size=20)
    tau[locked] = 0
    quat = rng.normal(size=4)
    quat /= np.linalg.norm(quat)
    pos = np.array([3.0, -2.0, 5.0])          # far above ground: no contacts
    v, w = rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)
    root = np.zeros((1, 2, 13))
    root[0, 0] = np.concatenate([pos, quat, v, w])
    root[0, 1, 6] = 1
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", np.stack([q, qd], -1)[None])
    o.set("TORQUES", tau[None])
    qdd, a0 = o.debug_aba(0)
    bp = make_params(model, o.get("BODY_PARAMS")[0])
    res = kane_residual(model, bp, (pos, quat, q, v, w, qd), (a0[:3], a0[3:], qdd), tau)
    scale = max(1.0, np.abs(tau).max())
    assert np.abs(res[~np.concatenate([np.zeros(6, bool), locked])]).max() < 2e-4 * scale, res


def test_free_fall_and_momentum(robot):
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    for j in range(18):
        tc.joint_armature[j] = 0.0
    o = OracleSim(robot["wmodel"], tc, 1)
    rng = np.random.default_rng(5)
    q = np.array(tc.default_dof_pos) + rng.uniform(-0.1, 0.1, 20)
    q[18:] = 0
    qd = rng.uniform(-0.5, 0.5, 20)      # below the URDF velocity limits: the (non-physical) clamp stays off
    qd[18:] = 0
    for j in range(20):
        model_lo, model_hi = model.dof_lower[j], model.dof_upper[j]
        assert not (model_lo < model_hi) or model_lo + 0.15 < q[j] < model_hi - 0.15 or j >= 18
    root = np.zeros((1, 2, 13))
    root[0, 0] = [0, 0, 50.0, 0, 0, 0, 1, 0.3, -0.2, 0.1, 0.5, -0.4, 0.3]
    root[0, 1, 6] = 1
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", np.stack([q, qd], -1)[None])
    o.set("TORQUES", np.zeros((1, 20)))
    bp = make_params(model, o.get("BODY_PARAMS")[0])
    mtot = sum(b[0] for b in bp)

    def momentum():
        r = o.get("ROOT_STATES")[0, 0]
        d = o.get("DOF_STATE")[0]
        tw = body_twists(model, bp, r[:3], r[3:7], d[:, 0], r[7:10], r[10:13], d[:, 1])
        P = sum(m * vc for m, I, c, vc, om in tw)
        cm = sum(m * c for m, I, c, vc, om in tw) / mtot
        L = sum(I @ om + m * np.cross(c - cm, vc) for m, I, c, vc, om in tw)
        return P, L
    P0, L0 = momentum()
    steps = 40
    for _ in range(steps):
        o.simulate()
    assert np.abs(o.get("DOF_STATE")[0, :, 1]).max() < 3.0
    P1, L1 = momentum()
    t = steps * tc.sim_dt
    np.testing.assert_allclose(P1 - P0, mtot * G * t, atol=1e-3 * mtot)       # impulse of gravity only
    np.testing.assert_allclose(L1, L0, atol=2e-3 * max(1.0, np.abs(L0).max()))  # no external torque about the com


def test_static_stance_supports_weight(robot):
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    tc.push_interval = 0
    o = OracleSim(robot["wmodel"], tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    root = np.zeros((1, 2, 13))
    root[0, 0, 2], root[0, 0, 6], root[0, 1, 6] = 0.34, 1, 1
    root[0, 1, :3] = [2.0, 0.0, 0.05]                         # the box actor rests out of reach
    o.set("ROOT_STATES", root)
    dof = np.zeros((1, 20, 2))
    dof[0, :, 0] = np.array(tc.default_dof_pos)
    o.set("DOF_STATE", dof)
    o.set("ACTIONS", np.zeros((1, 18)))
    for _ in range(3000):
        o.compute_torques()
        o.simulate()
    r = o.get("ROOT_STATES")[0, 0]
    f = o.get("NET_CONTACT_FORCE")[0, :27]
    mtot = sum(b[0] for b in make_params(model, o.get("BODY_PARAMS")[0]))
    assert np.abs(r[7:13]).max() < 5e-3                       # at rest
    assert 0.28 < r[2] < 0.34                                 # standing on its feet
    np.testing.assert_allclose(f[:, 2].sum(), mtot * 9.81, rtol=2e-3)
    np.testing.assert_allclose(f[:, :2].sum(0), 0, atol=0.5)
    feet = list(robot["wmodel"].feet_rb)
    assert (f[feet, 2] > 5).all()                             # every foot carries load
    # penetration stays within the contact margin
    o.refresh_rigid_body_state()
    rb = o.get("RIGID_BODY_STATE")[0]
    assert (rb[feet, 2] - 0.02 > -0.004).all()


def _arm_pose_in_self_contact(robot, tc, rng, want_thigh=False):
    """An arm pose whose gripper / wrist / elbow sphere starts inside the trunk box (or a front thigh capsule): found by trial in
    the air, where no terrain contact is possible."""
    model = robot["model"]
    n = 512
    o = OracleSim(robot["wmodel"], tc, n)
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 30.0
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = np.array(tc.default_dof_pos)[None]
    lo, hi = np.array(model.dof_lower[12:18]), np.array(model.dof_upper[12:18])
    lo[0], hi[0] = -1.5, 1.5
    dof[:, 12:18, 0] = rng.uniform(lo, hi, (n, 6))
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")
    hit = np.abs(f[:, [3, 7]]).sum((1, 2)) > 0 if want_thigh else (np.abs(f[:, 1]).sum(1) > 0)
    assert hit.any()
    return dof[np.nonzero(hit)[0][0], :, 0]


@pytest.mark.parametrize("want_thigh", [False, True])
def test_self_collision_impulses_are_internal(robot, want_thigh):
    """Self-collision pairs (arm spheres against the trunk box / the front thighs): the impulse acts on both bodies with opposite
    signs, so in free fall the robot's linear momentum changes by gravity's impulse only and its angular momentum about the
    centre of mass not at all; the two net_contact_force rows of a pair cancel; and the contact does separate the bodies."""
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    for j in range(18):
        tc.joint_armature[j] = 0.0
    rng = np.random.default_rng(11)
    q = _arm_pose_in_self_contact(robot, tc, rng, want_thigh)
    wm = type(robot["wmodel"]).from_buffer_copy(robot["wmodel"])
    for j in range(20):
        wm.qd_limit[j] = 0.0                 # the URDF velocity clamp (a non-physical projection) off: this is a conservation test
    o = OracleSim(wm, tc, 1)
    root = np.zeros((1, 2, 13))
    root[0, 0] = [0, 0, 30.0, 0, 0, 0, 1, 0.2, -0.1, 0.05, 0.3, -0.2, 0.1]
    root[0, 1, 6] = 1
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", np.stack([q, np.zeros(20)], -1)[None])
    o.set("TORQUES", np.zeros((1, 20)))
    bp = make_params(model, o.get("BODY_PARAMS")[0])
    mtot = sum(b[0] for b in bp)

    def momentum():
        r = o.get("ROOT_STATES")[0, 0]
        d = o.get("DOF_STATE")[0]
        tw = body_twists(model, bp, r[:3], r[3:7], d[:, 0], r[7:10], r[10:13], d[:, 1])
        P = sum(m * vc for m, I, c, vc, om in tw)
        cm = sum(m * c for m, I, c, vc, om in tw) / mtot
        L = sum(I @ om + m * np.cross(c - cm, vc) for m, I, c, vc, om in tw)
        return P, L
    P0, L0 = momentum()
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")[0, :27]                              # (row 27 is the box actor, resting on the ground far below)
    rows = np.nonzero(np.abs(f).sum(1) > 0)[0]
    assert len(rows) >= 2 and np.abs(f).max() > 0.5                     # a pair is pushing
    np.testing.assert_allclose(f.sum(0), 0, atol=1e-9 * np.abs(f).max())   # ... with equal and opposite forces
    steps = 1
    for _ in range(7):
        o.simulate(); steps += 1
    P1, L1 = momentum()
    t = steps * tc.sim_dt
    np.testing.assert_allclose(P1 - P0, mtot * G * t, atol=1e-3 * mtot)
    np.testing.assert_allclose(L1, L0, atol=2e-3 * max(1.0, np.abs(L0).max()))
    for _ in range(60):                                                  # depenetration at <= max_depenetration_velocity: it lets go
        o.simulate()
    assert np.abs(o.get("NET_CONTACT_FORCE")[0, :27]).max() < 1e-9 or np.abs(o.get("DOF_STATE")[0, 12:18, 1]).max() < 20.0


def test_trunk_and_thighs_rest_on_the_ground(robot):
    """The trunk box (corner spheres) and the thigh tops carry a robot whose legs are folded away: it comes to rest on them,
    net_contact_force reports the weight on the trunk / thigh rows (what penalize_contacts_on and terminate_after_contacts_on
    read), and the trunk does not sink through the ground."""
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    tc.push_interval = 0
    o = OracleSim(robot["wmodel"], tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    root = np.zeros((1, 2, 13))
    root[0, 0, 2], root[0, 0, 6], root[0, 1, 6] = 0.075, 1, 1
    root[0, 1, :3] = [2.0, 0.0, 0.05]                         # the box actor rests out of reach
    dof = np.zeros((1, 20, 2))
    dof[0, :, 0] = np.array(tc.default_dof_pos)
    for leg in range(4):
        dof[0, 3 * leg + 1, 0], dof[0, 3 * leg + 2, 0] = 2.9, -2.7          # legs folded up beside the body
    hold = dof[0, :18, 0] - np.array(tc.default_dof_pos)[:18]
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", dof)
    sc = np.array(tc.action_scale)
    act = np.where(sc > 0, hold / np.where(sc > 0, sc, 1.0), 0.0)
    o.set("ACTIONS", act[None])                                          # PD targets = the folded pose
    for _ in range(1500):
        o.compute_torques()
        o.simulate()
    r = o.get("ROOT_STATES")[0, 0]
    f = o.get("NET_CONTACT_FORCE")[0, :27]
    mtot = sum(b[0] for b in make_params(model, o.get("BODY_PARAMS")[0]))
    names = model.rb_names
    trunk, thighs = names.index("trunk"), [i for i, nm in enumerate(names) if "thigh" in nm]
    assert np.abs(r[7:13]).max() < 2e-2                                   # at rest
    assert 0.05 < r[2] < 0.08                                             # on its belly (box half height 0.057)
    np.testing.assert_allclose(f[:, 2].sum(), mtot * 9.81, rtol=5e-3)
    assert f[trunk, 2] + f[thighs, 2].sum() > 0.5 * mtot * 9.81           # trunk + thigh tops carry most of it
    assert f[trunk, 2] > 10.0


def test_self_collision_geometry_against_brute_force(robot):
    """The narrow phase of the self-collision pairs (closest point of the trunk box / of a thigh capsule to an arm sphere) against
    a brute-force search: for random arm poses of a robot at rest in free fall, a pair reports a contact force when the sphere
    penetrates a dense point sampling of the partner's surface (a separating velocity is demanded), none when it is clear of the
    contact margin (in between the contact is speculative: active, but bodies at relative rest need no impulse), and the force on
    the sphere's body points away from the partner (along the line from the nearest sampled surface point to the sphere centre)."""
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    wm = robot["wmodel"]
    n = 600
    rng = np.random.default_rng(23)
    o = OracleSim(wm, tc, n)
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 40.0
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = np.array(tc.default_dof_pos)[None]
    lo, hi = np.array(model.dof_lower[12:18]), np.array(model.dof_upper[12:18])
    lo[0], hi[0] = -1.5, 1.5
    dof[:, 12:18, 0] = rng.uniform(lo, hi, (n, 6))
    dof[:, [1, 4], 0] = rng.uniform(0.3, 1.4, (n, 2))                 # front thighs swung as well
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
    o.refresh_rigid_body_state()
    rb = o.get("RIGID_BODY_STATE")                                     # world poses of the rigid bodies BEFORE the substep
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")
    names = model.rb_names

    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    # surface samplings in the partner's frame
    hx, hy, hz = 0.3762 / 2, 0.0935 / 2, 0.114 / 2
    g = np.linspace(-1, 1, 41)
    u, v = np.meshgrid(g, g, indexing="ij")
    box = np.concatenate([np.stack([s * hx * np.ones_like(u), hy * u, hz * v], -1).reshape(-1, 3) for s in (-1, 1)] +
                         [np.stack([hx * u, s * hy * np.ones_like(u), hz * v], -1).reshape(-1, 3) for s in (-1, 1)] +
                         [np.stack([hx * u, hy * v, s * hz * np.ones_like(u)], -1).reshape(-1, 3) for s in (-1, 1)])
    ang = np.linspace(0, 2 * np.pi, 48, endpoint=False)
    zs = np.linspace(-0.213, 0.0, 60)
    cyl = np.stack([0.017 * np.cos(ang)[None] * np.ones((60, 1)), 0.017 * np.sin(ang)[None] * np.ones((60, 1)), zs[:, None] * np.ones((1, 48))], -1).reshape(-1, 3)
    th, ph = np.meshgrid(np.linspace(0, np.pi / 2, 12), ang, indexing="ij")
    cap = np.stack([0.017 * np.sin(th) * np.cos(ph), 0.017 * np.sin(th) * np.sin(ph), 0.017 * np.cos(th)], -1).reshape(-1, 3)
    caps = np.concatenate([cyl, cap, cap * [1, 1, -1] + [0, 0, -0.213]])
    checked = hits = 0
    pair_ids = [k for k in range(wm.ncp) if wm.cp_kind[k] != abi.CP_TERRAIN and wm.cp_body2[k] != abi.BOX_BODY]
    for rb1 in sorted({wm.cp_rb[k] for k in pair_ids}):               # per arm sphere: its net_contact_force row sums its pairs
        mine = [k for k in pair_ids if wm.cp_rb[k] == rb1]
        rad = wm.cp_radius[mine[0]]
        off1 = np.array(wm.cp_pos[mine[0]]) - np.array(model.rb_offset[rb1])
        for e in range(n):
            c = rb[e, rb1, :3] + rot(rb[e, rb1, 3:7]) @ off1
            gaps, aways, inside_box = [], [], False
            for k in mine:
                rb2 = wm.cp_rb2[k]
                R2 = rot(rb[e, rb2, 3:7])
                surf = box if names[rb2] == "trunk" else caps
                pts = rb[e, rb2, :3] + surf @ R2.T
                d = np.linalg.norm(pts - c, axis=1)
                gaps.append(d.min() - rad)
                aways.append((c - pts[d.argmin()]) / d.min())
                inside_box |= names[rb2] == "trunk" and bool(np.all(np.abs(R2.T @ (c - rb[e, rb2, :3])) < [hx, hy, hz]))
            pushing = np.abs(f[e, rb1]).sum() > 0
            if inside_box or min(gaps) < -4e-3:
                assert pushing, (names[rb1], e, gaps)
                hits += 1
                near = [i for i, gp in enumerate(gaps) if gp < tc.contact_margin + 4e-3]
                if not inside_box and len(near) == 1 and gaps[near[0]] > -0.5 * rad:
                    assert np.dot(f[e, rb1], aways[near[0]]) > 0.4 * np.linalg.norm(f[e, rb1]), (names[rb1], e)     # (normal + friction at mu = 1: within 66 deg)
            elif min(gaps) > tc.contact_margin + 4e-3:
                assert not pushing, (names[rb1], e, gaps)
            checked += 1
    assert checked > 1500 and hits > 20, (checked, hits)
