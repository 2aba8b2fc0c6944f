"""Closed-form physics cases run on the ORACLE (oracle/wbc_oracle.c, fp64): the external evidence for the part of the
hot path no reference output can pin (gym.simulate = PhysX, closed source and absent).

Each function runs one experiment and returns the measured numbers next to their closed-form values;
tests/test_oracle_contact_physics.py asserts on them, tools/physics_evidence.py prints the table DESIGN.md section 3 quotes.
The HIP kernel inherits the evidence through the HIP <-> oracle parity suite (tests/test_gpu_sim_parity.py).
"""
import copy
import ctypes

import numpy as np

from oracle import OracleSim, default_curriculum
from wbc_amd import abi

G = 9.81


def clone_struct(s):
    return type(s).from_buffer_copy(s)


def _oracle_factory(wmodel, tcfg, n, seed=1):
    return OracleSim(wmodel, tcfg, n, seed=seed)


def quiet_task(tc):
    """Task logic out of the way for cases that advance by whole policy steps (`sim.step`): no pushes, no terminations, no time-outs."""
    tc.push_interval = 0
    tc.term_z_threshold = -1e3
    tc.term_rp_threshold = 1e3
    tc.max_episode_length = 10 ** 8
    tc.resample_interval = 10 ** 8
    return tc


POLICY_FROM_SIM = [3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8, 12, 13, 14, 15, 16, 17]


def to_policy_order(a_sim):
    """Actions in simulator joint order -> the policy's order (WG:1070-1088): policy[POLICY_FROM_SIM[j]] = sim[j]."""
    a = np.zeros_like(np.asarray(a_sim, dtype=np.float64))
    a[..., POLICY_FROM_SIM] = a_sim
    return a


def folded_pose(tc):
    """Default pose with the four legs folded up beside the body: the robot lies on its trunk box and thigh tops."""
    q = np.array(tc.default_dof_pos, dtype=np.float64)
    for leg in range(4):
        q[3 * leg + 1], q[3 * leg + 2] = 2.9, -2.7
    return q


def hold_actions(tc, q):
    """Actions whose PD targets are the pose q (the frozen arm joints keep their default targets)."""
    sc = np.array(tc.action_scale)
    hold = q[:18] - np.array(tc.default_dof_pos)[:18]
    return np.where(sc > 0, hold / np.where(sc > 0, sc, 1.0), 0.0)


def incline_heightfield(tan_theta, rows=400, cols=40, hs=0.05, vs=0.0005):
    """An exactly planar int16 height grid rising along +x with slope tan_theta (a multiple of vs / hs = 0.01), centred on x = 0."""
    per_cell = tan_theta * hs / vs
    assert abs(per_cell - round(per_cell)) < 1e-9, "slope must be a multiple of vs/hs"
    h = (np.arange(rows)[:, None] - rows // 2) * int(round(per_cell)) * np.ones((1, cols))
    assert np.abs(h).max() < 32767
    return h.astype(np.int16), hs, vs, -(rows // 2) * hs, -(cols // 2) * hs, 0.0


def _rot_y(a):
    """xyzw quaternion of a rotation by a about +y."""
    return np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)])


def robot_on_incline(robot, tan_theta, mu_env, terrain_friction=1.0, t_settle=0.3, t_measure=0.5, iters=None, kick=0.0, make_sim=_oracle_factory):
    """The robot on its trunk (legs folded, joints held by the task's PD law) on a plane of slope tan_theta, released at rest
    (or with a downhill velocity `kick`, so that the contacts are certainly sliding), advanced by whole policy steps. Returns the
    down-slope acceleration over the measurement window, the closed form g (sin - mu cos) if sliding else 0, mu, the final speed.
    make_sim: the backend (the fp64 oracle by default; tests/test_gpu_contact_physics.py passes the HIP kernel's)."""
    tc = quiet_task(clone_struct(robot["tcfg"]))
    tc.terrain_friction = terrain_friction
    tc.action_delay = -1                                                   # the hold pose acts from the first step
    if iters is not None:
        tc.contact_iters = iters
    o = make_sim(robot["wmodel"], tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    o.set("FRICTION", np.array([mu_env]))
    o.set_heightfield(*incline_heightfield(tan_theta))
    th = np.arctan(tan_theta)
    q = folded_pose(tc)
    root = np.zeros((1, 2, 13))
    root[0, :, 6] = 1
    root[0, 1, :3] = [0.0, 50.0, 0.05]                                  # the box: out of the way
    # trunk centre 0.057 + margin above the plane along its normal, pitched to lie in the plane (height rises with +x:
    # nose up = rotation by -theta about +y)
    nrm = np.array([-np.sin(th), 0.0, np.cos(th)])
    root[0, 0, :3] = 0.060 * nrm
    root[0, 0, 3:7] = _rot_y(-th)
    down = np.array([-np.cos(th), 0.0, -np.sin(th)])                     # unit vector down the slope
    root[0, 0, 7:10] = kick * down
    o.set("ROOT_STATES", root)
    dof = np.zeros((1, 20, 2))
    dof[0, :, 0] = q
    o.set("DOF_STATE", dof)
    act = to_policy_order(hold_actions(tc, q))[None]
    dtp = tc.sim_dt * tc.decimation

    def run(T):
        for _ in range(int(round(T / dtp))):
            o.step(act)
        return o.get("ROOT_STATES")[0, 0, 7:10] @ down
    v0 = run(t_settle)
    v1 = run(t_measure)
    mu = max(0.0, 0.5 * (mu_env + terrain_friction))
    sliding = tan_theta > mu or kick > 0
    expect = G * (np.sin(th) - mu * np.cos(th)) if sliding else 0.0
    return dict(acc=(v1 - v0) / t_measure, expect=expect, mu=mu, v_end=v1, sticks_expected=not sliding)


def box_on_incline(robot, tan_theta, terrain_friction, t_settle=0.2, t_measure=0.5, kick=0.0, make_sim=_oracle_factory):
    """The free box actor (a 0.1 m cube resting on four corner spheres) on the same incline: a single rigid body, so the closed
    form holds without any joint compliance. mu = (box 1.0 + terrain) / 2."""
    tc = clone_struct(robot["tcfg"])
    tc.terrain_friction = terrain_friction
    wm = robot["wmodel"]
    o = make_sim(wm, tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    o.set_heightfield(*incline_heightfield(tan_theta))
    th = np.arctan(tan_theta)
    nrm = np.array([-np.sin(th), 0.0, np.cos(th)])
    down = np.array([-np.cos(th), 0.0, -np.sin(th)])
    root = np.zeros((1, 2, 13))
    root[0, :, 6] = 1
    root[0, 0, :3] = [0.0, 0.0, 60.0]                                    # the robot: far above, in free fall
    root[0, 1, :3] = (wm.box_half + 0.001) * nrm
    root[0, 1, 3:7] = _rot_y(-th)
    root[0, 1, 7:10] = kick * down
    o.set("ROOT_STATES", root)
    dof = np.zeros((1, 20, 2))
    dof[0, :, 0] = np.array(tc.default_dof_pos)
    o.set("DOF_STATE", dof)
    o.set("TORQUES", np.zeros((1, 20)))
    dt = tc.sim_dt

    def run(T):
        for _ in range(int(round(T / dt))):
            o.simulate()
        return o.get("ROOT_STATES")[0, 1, 7:10] @ down
    v0 = run(t_settle)
    v1 = run(t_measure)
    mu = max(0.0, 0.5 * (wm.box_friction + terrain_friction))
    sliding = tan_theta > mu or kick > 0
    expect = G * (np.sin(th) - mu * np.cos(th)) if sliding else 0.0
    w_end = np.abs(o.get("ROOT_STATES")[0, 1, 10:13]).max()
    return dict(acc=(v1 - v0) / t_measure, expect=expect, mu=mu, v_end=v1, spin=w_end, sticks_expected=not sliding)


def robot_drop(robot, height=0.42, T=3.0, make_sim=_oracle_factory):
    """The default stance released from the spawn height (init_state.pos z, widowGo1_config.py:134), advanced by whole policy steps with
    zero actions: impact with restitution 0, rest penetration, settle time. The legs are PD springs, so the trunk rings; the FEET must
    not leave the ground again (sampled after every policy step)."""
    tc = quiet_task(clone_struct(robot["tcfg"]))
    wm = robot["wmodel"]
    o = make_sim(wm, tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    root = np.zeros((1, 2, 13))
    root[0, :, 6] = 1
    root[0, 0, 2] = height
    root[0, 1, :3] = [2.0, 0.0, 0.05]
    o.set("ROOT_STATES", root)
    dof = np.zeros((1, 20, 2))
    dof[0, :, 0] = np.array(tc.default_dof_pos)
    o.set("DOF_STATE", dof)
    feet = list(wm.feet_rb)
    dtp = tc.sim_dt * tc.decimation
    n = int(round(T / dtp))
    foot_z = np.zeros((n, 4)); foot_vz = np.zeros((n, 4)); speed = np.zeros(n); fz = np.zeros(n)
    zero = np.zeros((1, 18))
    for k in range(n):
        o.step(zero)
        rb = o.get("RIGID_BODY_STATE")[0]
        foot_z[k] = rb[feet, 2] - 0.02                                   # lowest point of the foot sphere
        foot_vz[k] = rb[feet, 9]
        r = o.get("ROOT_STATES")[0, 0]
        speed[k] = max(np.abs(r[7:13]).max(), np.abs(o.get("DOF_STATE")[0, :, 1]).max())
        fz[k] = o.get("NET_CONTACT_FORCE")[0, :27, 2].sum()
    touch = int(np.argmax((foot_z < tc.contact_margin).all(1)))           # first policy step with all four feet in the contact band
    after = slice(touch + 1, None)
    quiet = np.nonzero(speed > 2e-2)[0]
    settle = (quiet[-1] + 1) * dtp if len(quiet) else 0.0
    return dict(t_touch=(touch + 1) * dtp, rebound_height=float(foot_z[after].max()), rebound_vz=float(foot_vz[after].max()),
                impact_vz=float(foot_vz[max(touch - 1, 0)].min()), rest_penetration=float(-foot_z[-1].min()), settle_time=settle - touch * dtp,
                rest_force=float(fz[-1]), contact_offset=float(tc.contact_margin))


def box_drop(robot, height=None, T=1.5, make_sim=_oracle_factory):
    """The box actor released from its spawn height (box_env_origins_z = 0.21: 0.16 m of free fall, widowGo1_config.py:191)."""
    tc = clone_struct(robot["tcfg"])
    wm = robot["wmodel"]
    o = make_sim(wm, tc, 1)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    z0 = float(tc.box_origin_z) if height is None else height
    root = np.zeros((1, 2, 13))
    root[0, :, 6] = 1
    root[0, 0, :3] = [0.0, 0.0, 60.0]
    root[0, 1, :3] = [3.0, 0.2, z0]
    o.set("ROOT_STATES", root)
    dof = np.zeros((1, 20, 2)); dof[0, :, 0] = np.array(tc.default_dof_pos)
    o.set("DOF_STATE", dof)
    o.set("TORQUES", np.zeros((1, 20)))
    dt = tc.sim_dt
    n = int(round(T / dt))
    z = np.zeros(n); vz = np.zeros(n); f = np.zeros(n); tm = np.zeros(n)
    for k in range(n):
        o.simulate()
        b = o.get("ROOT_STATES")[0, 1]
        z[k], vz[k] = b[2] - wm.box_half, b[9]
        f[k] = o.get("NET_CONTACT_FORCE")[0, 27, 2]
        tm[k] = o.get("BOX_SLEEP_TIMER")[0]
    touch = int(np.argmax(z < tc.contact_margin))
    t_fall = np.sqrt(2 * (z0 - wm.box_half) / G)
    quiet = np.nonzero(np.abs(vz) > 1e-3)[0]
    awake = np.nonzero(f != 0)[0]                                        # asleep (frozen, PhysX-style) it reports no contact force
    asleep_from = (awake[-1] + 1) * dt if len(awake) and awake[-1] + 1 < n else float("nan")
    return dict(t_touch=touch * dt, t_touch_expected=t_fall, impact_vz=float(vz[touch - 1]), rebound_vz=float(vz[touch:].max()),
                rest_penetration=float(-z[-1]), rest_force=float(f[awake[-1]]), weight=float(o.get("BOX_MASS")[0] * G),
                asleep_from=float(asleep_from), final_speed=float(np.abs(o.get("ROOT_STATES")[0, 1, 7:13]).max()), timer=float(tm[-1]),
                settle_time=(quiet[-1] + 1) * dt - touch * dt, tilt=float(np.abs(o.get("ROOT_STATES")[0, 1, 3:5]).max()))


def joint_pd_step(robot, joint, step=0.2, T=0.6, armature=True):
    """ONE joint's PD response with everything else locked (base mass 1e7 kg, every other joint an armature of 1e6 kg m^2, no
    gravity, no contacts): released `step` rad away from its target, the simulated trajectory against the exact solution of
    I q'' = Kp (q* - q) - Kd q' (the reference's explicit law WG:1281, integrated exactly), I = the joint's inertia measured with
    a torque pulse. armature=False removes this framework's implicit-PD term dt Kd + dt^2 Kp from the joint
    -- what the explicit law does at dt = 5 ms. Returns inertia, armature, deviations in units of the step, rise times."""
    tc = clone_struct(robot["tcfg"])
    tc.push_interval = 0
    for k in range(3):
        tc.gravity[k] = 0.0
    tc.contact_margin = -1e30
    arm = float(tc.joint_armature[joint]) if armature else 0.0
    for j in range(18):
        tc.joint_armature[j] = 1e6
    wm = clone_struct(robot["wmodel"])
    for j in range(20):
        wm.qd_limit[j] = 0.0
        wm.q_lower[j] = wm.q_upper[j] = 0.0                               # joint-limit stops off
    kp, kd = float(tc.p_gains[joint]), float(tc.d_gains[joint])
    q0 = np.array(tc.default_dof_pos)

    def fresh(tcfg, dq):
        o = OracleSim(wm, tcfg, 1)
        o.set_curriculum(default_curriculum(robot["cfg"]))
        bp = o.get("BODY_PARAMS")
        bp[0, 0] = 1e7; bp[0, 4:7] = 1e7                                 # pinned base
        o.set("BODY_PARAMS", bp)
        root = np.zeros((1, 2, 13)); root[0, :, 6] = 1; root[0, 0, 2] = 50.0; root[0, 1, :3] = [9.0, 9.0, 0.05]
        o.set("ROOT_STATES", root)
        dof = np.zeros((1, 20, 2)); dof[0, :, 0] = q0; dof[0, joint, 0] += dq
        o.set("DOF_STATE", dof)
        o.set("ACTIONS", np.zeros((1, 18)))
        return o
    tz = clone_struct(tc)
    tz.joint_armature[joint] = 0.0
    o = fresh(tz, 0.0)
    tau = np.zeros((1, 20)); tau[0, joint] = 1e-3
    o.set("TORQUES", tau)
    o.simulate()
    inertia = 1e-3 / (o.get("DOF_STATE")[0, joint, 1] / tc.sim_dt)
    tc.joint_armature[joint] = arm
    o = fresh(tc, -step)
    dt = tc.sim_dt
    n = int(round(T / dt))
    traj = np.zeros(n + 1)
    for k in range(n):
        o.compute_torques()
        o.simulate()
        traj[k + 1] = o.get("DOF_STATE")[0, joint, 0] - (q0[joint] - step)
    t = np.arange(n + 1) * dt
    # exact: x = q - q*, I x'' + Kd x' + Kp x = 0, x(0) = -step, x'(0) = 0
    disc = kd * kd - 4 * kp * inertia
    if disc >= 0:
        r1, r2 = (-kd + np.sqrt(disc)) / (2 * inertia), (-kd - np.sqrt(disc)) / (2 * inertia)
        a = -step * r2 / (r2 - r1); b = -step - a
        x = a * np.exp(r1 * t) + b * np.exp(r2 * t)
    else:
        al, om = -kd / (2 * inertia), np.sqrt(-disc) / (2 * inertia)
        x = np.exp(al * t) * (-step * np.cos(om * t) + (step * al / om) * np.sin(om * t))
    exact = x + step
    stable = bool(np.isfinite(traj).all() and np.abs(traj).max() < 5 * step)
    dev = np.abs(traj - exact) if stable else np.full(n + 1, np.inf)

    def rise(y):
        i10, i90 = np.argmax(y >= 0.1 * step), np.argmax(y >= 0.9 * step)
        return (i90 - i10) * dt
    return dict(joint=joint, kp=kp, kd=kd, inertia=float(inertia), armature=arm, max_dev=float(dev.max() / step),
                t_max_dev=float(t[dev.argmax()]), final_dev=float(dev[-1] / step), rise_sim=float(rise(traj)) if stable else float("nan"),
                rise_exact=float(rise(exact)), overshoot_sim=float(traj.max() / step - 1) if stable else float("nan"),
                overshoot_exact=float(exact.max() / step - 1), stable=stable, gain_ratio=float(inertia / (inertia + arm)),
                explicit_margin=float(dt * kd / inertia))


def trot_actions(t, amp=0.35, freq=2.5):
    """Open-loop trot: diagonal leg pairs in phase, thigh / calf targets swung sinusoidally (sim order FL FR RL RR)."""
    a = np.zeros(18)
    for leg, ph in enumerate((0.0, np.pi, np.pi, 0.0)):
        s = np.sin(2 * np.pi * freq * t + ph)
        a[3 * leg + 1] = amp * s / 0.45
        a[3 * leg + 2] = -1.2 * amp * max(s, 0.0) / 0.45
    return a


def contact_states(robot, n_per_kind=128, seed=3):
    """States for the solver-convergence study: (stance) robots dropped and standing under small random actions, (trot) an
    open-loop trot, (trunk) robots lying on their trunks / sides, (self) arms driven into the trunk and the front thighs, (box)
    a foot pushing the box actor. Returns the oracle holding the 5 * n_per_kind states and the labels."""
    tc = clone_struct(robot["tcfg"])
    tc.push_interval = 0
    tc.term_z_threshold = -1.0
    tc.term_rp_threshold = 10.0
    tc.max_episode_length = 10 ** 6
    kinds = ["stance", "trot", "trunk", "self", "box"]
    n = n_per_kind * len(kinds)
    rng = np.random.default_rng(seed)
    o = OracleSim(robot["wmodel"], tc, n, seed=seed)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    o.set("FRICTION", rng.uniform(-0.5, 3.0, n))
    q0 = np.array(tc.default_dof_pos)
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1
    root[:, 0, 2] = 0.36
    root[:, 1, :3] = [50.0, 0.0, 0.05]
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = q0[None]
    label = np.repeat(np.arange(len(kinds)), n_per_kind)
    tr = label == 2
    root[tr, 0, 2] = 0.08
    dof[tr, :, 0] = folded_pose(tc)[None]
    side = tr & (rng.random(n) < 0.3)
    root[side, 0, 2] = 0.16
    root[side, 0, 3:7] = [0.7071068, 0.0, 0.0, 0.7071068]
    bx = label == 4
    # the box in front of the front-left foot (default stance: FL foot at about (+0.19, +0.13) of the base)
    root[bx, 1, 0] = 0.19 + 0.05 + rng.uniform(0.0, 0.03, bx.sum())
    root[bx, 1, 1] = 0.13 + rng.uniform(-0.03, 0.03, bx.sum())
    root[bx, 0, 7] = rng.uniform(0.3, 0.8, bx.sum())                     # walking into it
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", dof)
    lo, hi = np.array(robot["model"].dof_lower[12:18]), np.array(robot["model"].dof_upper[12:18])
    lo[0], hi[0] = -1.5, 1.5
    arm_tgt = rng.uniform(lo, hi, (n, 6)) - q0[12:18]
    steps = rng.integers(15, 60, n)
    act = np.zeros((n, 18))
    for k in range(int(steps.max())):
        t = k * tc.sim_dt * tc.decimation
        a = 0.1 * np.tanh(rng.standard_normal((n, 18)))
        a[label == 1] += trot_actions(t)[None]
        a[tr, :12] = hold_actions(tc, folded_pose(tc))[None, :12]
        sc = np.array(tc.action_scale)[12:18]
        a[label == 3, 12:18] = np.where(sc > 0, arm_tgt[label == 3] / np.where(sc > 0, sc, 1.0), 0.0) * min(1.0, (k + 1) / 10)
        live = k < steps
        act[live] = a[live]
        # envs that reached their step count are frozen by re-setting their state afterwards: simpler to step all and
        # snapshot per env
        if k == 0:
            snap_root, snap_dof, snap_act = root.copy(), dof.copy(), act.copy()
        pol = np.zeros((n, 18))
        pol[:, [3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8, 12, 13, 14, 15, 16, 17]] = a   # sim order -> policy order
        o.step(pol)
        done = (k + 1) == steps
        if done.any():
            snap_root[done] = o.get("ROOT_STATES")[done]
            snap_dof[done] = o.get("DOF_STATE")[done]
            snap_act[done] = o.get("ACTIONS")[done]
    return dict(root=snap_root, dof=snap_dof, actions=snap_act, label=label, kinds=kinds, friction=o.get("FRICTION"), tcfg=tc)


def solver_convergence(robot, states, iters_list=(2, 8, 64)):
    """One substep from identical states with contact_iters in iters_list: contact forces and post-step velocities."""
    out = {}
    n = states["root"].shape[0]
    for it in iters_list:
        tc = clone_struct(states["tcfg"])
        tc.contact_iters = it
        o = OracleSim(robot["wmodel"], tc, n)
        o.set_curriculum(default_curriculum(robot["cfg"]))
        o.set("FRICTION", states["friction"])
        o.set("ROOT_STATES", states["root"])
        o.set("DOF_STATE", states["dof"])
        o.set("ACTIONS", states["actions"])
        o.compute_torques()
        o.simulate()
        r = o.get("ROOT_STATES")
        out[it] = dict(f=o.get("NET_CONTACT_FORCE"), v=np.concatenate([r[:, 0, 7:13], o.get("DOF_STATE")[:, :18, 1], r[:, 1, 7:13]], 1))
    return out


def body_energy(model, bp, root, dof, twists_fn):
    tw = twists_fn(model, bp, root[:3], root[3:7], dof[:, 0], root[7:10], root[10:13], dof[:, 1])
    ke = sum(0.5 * m * vc @ vc + 0.5 * om @ I @ om for m, I, c, vc, om in tw)
    pe = sum(m * G * c[2] for m, I, c, vc, om in tw)
    return ke, pe


def free_flight_energy(robot, T=2.0, seed=0, gravity=True):
    """Passive swing: the robot tumbling in free flight with every joint swinging freely (no torques, no armature, no joint
    limits, no velocity clamp, no contacts). Total mechanical energy (kinetic + m g z) and the angular momentum about the centre
    of mass are constants of the motion; returns their drift over T seconds of semi-implicit Euler at dt = 5 ms."""
    from test_oracle_physics import body_twists, make_params
    model = robot["model"]
    tc = clone_struct(robot["tcfg"])
    for j in range(18):
        tc.joint_armature[j] = 0.0
    tc.contact_margin = -1e30
    if not gravity:
        for k in range(3):
            tc.gravity[k] = 0.0
    wm = clone_struct(robot["wmodel"])
    for j in range(20):
        wm.qd_limit[j] = 0.0
        wm.q_lower[j] = wm.q_upper[j] = 0.0
    o = OracleSim(wm, tc, 1)
    rng = np.random.default_rng(seed)
    q = np.array(tc.default_dof_pos) + rng.uniform(-0.2, 0.2, 20)
    qd = rng.uniform(-1.5, 1.5, 20)
    q[18:], qd[18:] = 0, 0
    root = np.zeros((1, 2, 13)); root[0, :, 6] = 1
    root[0, 0] = [0, 0, 100.0, 0, 0, 0, 1, 0.4, -0.3, 0.5, 1.0, -0.8, 0.6]
    root[0, 1, :3] = [9.0, 9.0, 0.05]
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", np.stack([q, qd], -1)[None])
    o.set("TORQUES", np.zeros((1, 20)))
    bp = make_params(model, o.get("BODY_PARAMS")[0])
    mtot = sum(b[0] for b in bp)
    g = G if gravity else 0.0

    def invariants():
        r = o.get("ROOT_STATES")[0, 0]
        d = o.get("DOF_STATE")[0]
        tw = body_twists(model, bp, r[:3], r[3:7], d[:, 0], r[7:10], r[10:13], d[:, 1])
        ke = sum(0.5 * m * vc @ vc + 0.5 * om @ I @ om for m, I, c, vc, om in tw)
        pe = sum(m * g * c[2] for m, I, c, vc, om in tw)
        cm = sum(m * c for m, I, c, vc, om in tw) / mtot
        vcm = sum(m * vc for m, I, c, vc, om in tw) / mtot
        L = sum(I @ om + m * np.cross(c - cm, vc - vcm) for m, I, c, vc, om in tw)
        ke_int = ke - 0.5 * mtot * vcm @ vcm                              # kinetic energy of the motion about the centre of mass
        return ke + pe, ke_int, L
    e0, k0, L0 = invariants()
    n = int(round(T / tc.sim_dt))
    es, ks = [e0], [k0]
    for _ in range(n):
        o.simulate()
        e, k, L = invariants()
        es.append(e); ks.append(k)
    es, ks = np.array(es), np.array(ks)
    return dict(e0=float(e0), ke_internal0=float(k0), drift_total=float(np.abs(es - e0).max()), drift_rel_internal=float(np.abs(ks - k0).max() / k0),
                drift_end_rel_internal=float((ks[-1] - k0) / k0), L_drift_rel=float(np.linalg.norm(L - L0) / np.linalg.norm(L0)),
                qd_max=float(np.abs(o.get("DOF_STATE")[0, :, 1]).max()))


def self_collision_momentum(robot, n=400, seed=0):
    """Self-collision is internal to the robot. Robots at rest in free flight (no gravity), every joint drawn uniformly inside its
    limits, so that limb pairs start out penetrating: ONE substep with the collision set on and one with self-collision off, from the
    same state. The contact impulses set joints and base in motion (momentum of the parts: `impulse`), but the robot's total linear
    momentum and its angular momentum about the centre of mass must end where the contact-free run ends (zero: nothing else acts).
    Returns, per env with a contact, the impulse scale and the two momentum differences."""
    from test_oracle_physics import body_twists, make_params
    model = robot["model"]
    tc = clone_struct(robot["tcfg"])
    for k in range(3):
        tc.gravity[k] = 0.0
    rng = np.random.default_rng(seed)
    lo, hi = np.array(model.dof_lower, dtype=np.float64), np.array(model.dof_upper, dtype=np.float64)
    free = ~(lo < hi)
    lo[free], hi[free] = -np.pi, np.pi
    lo[18:], hi[18:] = 0.0, 0.0
    root = np.zeros((n, 2, 13)); root[:, :, 6] = 1; root[:, 0, 2] = 50.0; root[:, 1, :3] = [9.0, 9.0, 0.05]
    dof = np.zeros((n, 20, 2)); dof[:, :, 0] = rng.uniform(lo, hi, (n, 20))
    out = []
    sims = []
    for wm in (clone_struct(robot["wmodel"]), abi.fill_model(model, self_collisions=False)):
        for j in range(20):
            wm.qd_limit[j] = 0.0           # (the URDF's joint-velocity clamp is not a force: it would eat momentum after a deep penetration's kick)
        o = OracleSim(wm, tc, n)
        o.set("ROOT_STATES", root); o.set("DOF_STATE", dof); o.set("TORQUES", np.zeros((n, 20)))
        o.simulate()
        sims.append(o)
    on, off = sims
    f = on.get("NET_CONTACT_FORCE")
    bp = make_params(model, on.get("BODY_PARAMS")[0])
    mtot = sum(b[0] for b in bp)

    def momenta(o, e):
        # (the velocities the substep produced, at the configuration it started from: the contact impulses act there, and the two
        # runs then differ in velocities only)
        r = o.get("ROOT_STATES")[e, 0]
        d = o.get("DOF_STATE")[e]
        tw = body_twists(model, bp, root[e, 0, :3], root[e, 0, 3:7], dof[e, :, 0], r[7:10], r[10:13], d[:, 1])
        cm = sum(m * c for m, I, c, vc, om in tw) / mtot
        vcm = sum(m * vc for m, I, c, vc, om in tw) / mtot
        L = sum(I @ om + m * np.cross(c - cm, vc - vcm) for m, I, c, vc, om in tw)
        parts = sum(m * np.linalg.norm(vc) for m, I, c, vc, om in tw)
        return mtot * vcm, L, parts
    for e in range(n):
        if np.abs(f[e, :27]).sum() == 0:
            continue
        P1, L1, parts = momenta(on, e)
        P0, L0, _ = momenta(off, e)
        out.append(dict(impulse=float(np.abs(f[e, :27]).sum() * 0.5 * tc.sim_dt), parts=float(parts), dP=float(np.linalg.norm(P1 - P0)),
                        dL=float(np.linalg.norm(L1 - L0)), pair_sum=float(np.abs(f[e, :27].sum(0)).max())))
    return out


def robot_kicks_box(robot, seed=0, make_sim=_oracle_factory):
    """A foot sphere started inside the box actor, both in free fall far above the ground (no terrain contact, no gravity): the pair
    impulse is internal to the robot + box system, so its total linear momentum and its angular momentum about the common centre
    of mass are conserved while the two separate."""
    from test_oracle_physics import body_twists, make_params
    model = robot["model"]
    tc = clone_struct(robot["tcfg"])
    for j in range(18):
        tc.joint_armature[j] = 0.0
    for k in range(3):
        tc.gravity[k] = 0.0
    wm = clone_struct(robot["wmodel"])
    for j in range(20):
        wm.qd_limit[j] = 0.0
    o = make_sim(wm, tc, 1)
    rng = np.random.default_rng(seed)
    q = np.array(tc.default_dof_pos)
    root = np.zeros((1, 2, 13)); root[0, :, 6] = 1
    root[0, 0] = [0, 0, 100.0, 0, 0, 0, 1, 0.3, 0.0, 0.0, 0.0, 0.0, 0.0]
    o.set("ROOT_STATES", root)
    o.set("DOF_STATE", np.stack([q, np.zeros(20)], -1)[None])
    o.refresh_rigid_body_state()
    foot = o.get("RIGID_BODY_STATE")[0, wm.feet_rb[0], :3]
    # the box just in front of the FL foot, overlapping its sphere by 5 mm, spinning and drifting towards the robot
    root[0, 1, :3] = foot + np.array([wm.box_half + 0.02 - 0.005, 0.01, 0.015])
    root[0, 1, 7:13] = [-0.4, 0.05, 0.1, 0.5, -1.0, 0.8]
    o.set("ROOT_STATES", root)
    o.set("TORQUES", np.zeros((1, 20)))
    bp = make_params(model, o.get("BODY_PARAMS")[0])
    mb = float(o.get("BOX_MASS")[0])
    Ib = mb * (2.0 / 3.0) * wm.box_half ** 2

    def momentum():
        r = o.get("ROOT_STATES")[0]
        d = o.get("DOF_STATE")[0]
        tw = body_twists(model, bp, r[0, :3], r[0, 3:7], d[:, 0], r[0, 7:10], r[0, 10:13], d[:, 1])
        tw = tw + [(mb, Ib * np.eye(3), r[1, :3], r[1, 7:10], r[1, 10:13])]
        mt = sum(t[0] for t in tw)
        P = sum(m * vc for m, I, c, vc, om in tw)
        cm = sum(m * c for m, I, c, vc, om in tw) / mt
        L = sum(I @ om + m * np.cross(c - cm, vc) for m, I, c, vc, om in tw)
        return P, L, sum(m * vc for m, I, c, vc, om in tw[:-1])
    P0, L0, Pr0 = momentum()
    fmax = 0.0
    pair_cancels = 0.0
    for _ in range(20):
        o.simulate()
        f = o.get("NET_CONTACT_FORCE")[0]
        fmax = max(fmax, np.abs(f[27]).max())
        pair_cancels = max(pair_cancels, np.abs(f.sum(0)).max())
    P1, L1, Pr1 = momentum()
    return dict(dP=float(np.abs(P1 - P0).max()), dL=float(np.abs(L1 - L0).max()), L_scale=float(max(1.0, np.abs(L0).max())),
                robot_dP=float(np.abs(Pr1 - Pr0).max()), box_force_max=float(fmax), pair_force_sum=float(pair_cancels),
                separated=bool(np.abs(o.get("NET_CONTACT_FORCE")[0, 27]).max() == 0.0))
