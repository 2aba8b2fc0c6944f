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
    """Self-collision as configured (asset.self_collisions = 0: every pair of non-adjacent links, widowGo1_config.py:180): the oracle's
    broad phase, promotion into dynamic slots and exact tests (sphere vs trunk box, limb vs limb as unions of capsule and end spheres)
    against brute-force geometry that shares no code with it (tests/self_collision_geometry.py). Robots at rest in free fall, every
    joint drawn uniformly inside its limits: a rigid body reports a contact force when one of its primitives penetrates another
    link's by more than 4 mm (bodies at relative rest: only a penetration demands an impulse), none when all of its pairs are clear
    of the contact margin by 4 mm; pair forces are internal (they cancel over the robot); with one penetrating pair on a body the
    force on an arm sphere points away from the trunk box."""
    import self_collision_geometry as G
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    wm = robot["wmodel"]
    n = 8000
    rng = np.random.default_rng(23)
    o = OracleSim(wm, tc, n)
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 40.0
    lo, hi = np.array(model.dof_lower, dtype=np.float64), np.array(model.dof_upper, dtype=np.float64)
    free = ~(lo < hi)
    lo[free], hi[free] = -np.pi, np.pi
    lo[18:], hi[18:] = 0.0, 0.0
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = rng.uniform(lo, hi, (n, 20))
    dof[: n // 2, :12, 0] = np.array(tc.default_dof_pos)[None, :12] + rng.uniform(-0.5, 0.5, (n // 2, 12))     # half of them: legs near the stance
    dof[:, :12, 0] = np.clip(dof[:, :12, 0], lo[:12], hi[:12])
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
    o.refresh_rigid_body_state()
    rb = o.get("RIGID_BODY_STATE")                                     # world poses of the rigid bodies BEFORE the substep
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")
    names = model.rb_names
    gaps = {k: g for k, g in G.all_pairs(rb, names).items() if not (k[0] in G.ARM_LIMBS and k[1] in G.ARM_LIMBS)}     # (the arm's own links do not collide with each other: the stated exception)
    # which rigid bodies report a primitive's contacts: the calf limb's shaft and knee on the calf, its foot sphere on the foot; the arm
    # capsules on their links (the elbow sphere shares its row with the forearm capsule, the wrist / gripper-tip spheres have their own)
    prim_rbs = {"trunk": ["trunk"], "elbow": ["wx250s/upper_forearm_link"], "wrist": ["wx250s/wrist_link"], "gripper": ["wx250s/ee_gripper_link"],
                "upper_arm": ["wx250s/upper_arm_link"], "forearm": ["wx250s/upper_forearm_link"], "hand": ["wx250s/gripper_link"]}
    for l in G.LEGS:
        prim_rbs[l + "_thigh"] = [l + "_thigh"]
        prim_rbs[l + "_calf"] = [l + "_calf", l + "_foot"]
    margin = float(tc.contact_margin)
    fnorm = np.linalg.norm(f[:, :27], axis=-1)
    np.testing.assert_allclose(f[:, :27].sum(1), 0.0, atol=1e-9)           # internal forces
    hits = clear = 0
    per_prim_pen = {p: np.zeros(n, dtype=int) for p in prim_rbs}
    per_prim_near = {p: np.zeros(n, dtype=bool) for p in prim_rbs}
    for (a, b), g in gaps.items():
        for p in (a, b):
            per_prim_pen[p] += g < -4e-3
            per_prim_near[p] |= g < margin + 4e-3
    # "must push" is asserted where the penetrating pair is the robot's ONLY pair inside the margin: with several contacts on one chain
    # another one's impulse may already be separating the pair (the solver then rightly gives it none)
    n_near = sum((g < margin + 4e-3).astype(int) for g in gaps.values())
    groups = {}                                                            # primitives by the rigid-body rows they report on
    for p, rbs in prim_rbs.items():
        groups.setdefault(tuple(rbs), []).append(p)
    for rbs, prims in groups.items():
        pushing = fnorm[:, [names.index(r) for r in rbs]].sum(1) > 0
        must = (sum(per_prim_pen[p] for p in prims) > 0) & (n_near == 1)
        mustnot = ~np.any([per_prim_near[p] for p in prims], axis=0)
        assert pushing[must].all(), (prims, np.nonzero(must & ~pushing)[0][:5])
        assert not pushing[mustnot].any(), (prims, np.nonzero(mustnot & pushing)[0][:5])
        hits += int(must.sum()); clear += int(mustnot.sum())
    assert hits > 300 and clear > 50000, (hits, clear)
    idx = {p: [names.index(r) for r in rbs] for p, rbs in prim_rbs.items()}
    # direction: an arm sphere whose ONLY near pair is the trunk is pushed away from the box (out through the nearest face when inside)
    _, arm, trunk = G.primitives(rb, names)
    Rt, pt, half = trunk
    checked = 0
    for sname, (c, r) in arm.items():
        g = gaps[(sname, "trunk")]
        loc = np.einsum("nji,nj->ni", Rt, c - pt)
        outside = np.any(np.abs(loc) > half, axis=1)
        cl = np.clip(loc, -half, half)
        away = np.einsum("nij,nj->ni", Rt, loc - cl)
        nrm = np.linalg.norm(away, axis=1)
        sel = (g < -4e-3) & outside & (nrm > 1e-6) & (n_near == 1)
        fa = f[:, idx[sname][0]]
        for e in np.nonzero(sel)[0]:
            assert np.dot(fa[e], away[e] / nrm[e]) > 0.4 * np.linalg.norm(fa[e]) > 0, (sname, e)     # (normal + friction at mu = 1: within 66 deg)
            checked += 1
    assert checked > 10, checked


def test_free_box_candidates_against_brute_force(robot):
    """The robot spheres that can meet the free box besides the five static pairs (knees, shins, the trunk's bottom corners: promoted
    into the box row's dynamic slots) and the static five (feet, gripper tip), against brute-force sphere-vs-cube geometry: boxes at
    random poses around robots in free fall, everything at rest. The box reports a contact force when a listed sphere penetrates it
    by more than 3 mm, none when all of them are clear of the margin by 3 mm; the robot's rows carry the opposite force."""
    model, tc = robot["model"], copy.copy(robot["tcfg"])
    wm = robot["wmodel"]
    n = 4000
    rng = np.random.default_rng(31)
    o = OracleSim(wm, tc, n)
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 40.0
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = np.array(tc.default_dof_pos)[None] + rng.uniform(-0.3, 0.3, (n, 20))
    dof[:, 18:, 0] = 0
    o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
    o.refresh_rigid_body_state()
    rb = o.get("RIGID_BODY_STATE")
    names = model.rb_names
    cps, _, cands = abi.collision_set(model)
    sph = {c["sph"]: c for c in cps if c["kind"] == abi.CP_TERRAIN and c["body"] != abi.BOX_BODY}
    listed = sorted({c["a"] for c in cands if c["kind"] == abi.PR_SPHERE_BOX} | {c["sph"] for c in cps if c["kind"] == abi.CP_BOX and c["body2"] == abi.BOX_BODY})
    assert len(listed) == 17

    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    centres = np.zeros((n, len(listed), 3))
    for e in range(n):
        for i, si in enumerate(listed):
            c = sph[si]
            centres[e, i] = rb[e, c["rb"], :3] + rot(rb[e, c["rb"], 3:7]) @ (np.asarray(c["pos"]) - np.asarray(model.rb_offset[c["rb"]]))
    # the box: next to one of the listed spheres (a random one per env), random orientation
    pick = rng.integers(0, len(listed), n)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    root[:, 1, :3] = centres[np.arange(n), pick] + d * rng.uniform(0.03, 0.11, (n, 1))
    root[:, 1, 3:7] = q
    o.set("ROOT_STATES", root)
    o.simulate()
    f = o.get("NET_CONTACT_FORCE")
    h = float(wm.box_half)
    gaps = np.zeros((n, len(listed)))
    for e in range(n):
        Rb = rot(q[e])
        for i, si in enumerate(listed):
            loc = Rb.T @ (centres[e, i] - root[e, 1, :3])
            cl = np.clip(loc, -h, h)
            out = np.linalg.norm(loc - cl)
            gaps[e, i] = (out if out > 0 else -(h - np.abs(loc)).min()) - sph[si]["radius"]
    box_push = np.abs(f[:, 27]).sum(-1) > 0
    pen, clear = (gaps < -3e-3).any(1), (gaps > float(tc.contact_margin) + 3e-3).all(1)
    few = (gaps < float(tc.contact_margin) + 3e-3).sum(1) <= 3                      # (the box row has three dynamic slots)
    assert box_push[pen & few].all() and not box_push[clear].any(), (np.nonzero(pen & few & ~box_push)[0][:5], np.nonzero(clear & box_push)[0][:5])
    assert (pen & few).sum() > 800 and clear.sum() > 300, (pen.sum(), clear.sum())
    np.testing.assert_allclose(f[:, :28].sum(1), 0.0, atol=1e-9)                    # the pair forces cancel over robot + box
    # candidates beyond the static five were exercised
    dyn = [i for i, si in enumerate(listed) if si not in (0, 1, 2, 3, 8)]
    assert ((gaps[:, dyn] < -3e-3).any(1) & box_push).sum() > 400
