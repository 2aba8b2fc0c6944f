"""ctypes mirrors of the structs in include/wbc_sim.h and the host-side fillers that turn a
`RobotModel` + config objects into them. This is the Python end of the C-ABI; it contains no
torch types (tensors cross the boundary as raw device pointers)."""
from __future__ import annotations

import ctypes as C
import math
import json
import os
from typing import Optional

import numpy as np

from .urdf_model import RobotModel, merge_piece

NB, NJ, NDOF, NACT, NRB, NRB_ENV, NFEET, NCP = 19, 18, 20, 18, 27, 28, 4, 64
NSPH, NLIMB = 28, 11
BOX_BODY, BOX_RB = NB, NRB                                # the free box actor's pseudo body index / rigid-body row
CP_NONE, CP_TERRAIN, CP_BOX, CP_LIMBS, CP_DYNAMIC = -1, 0, 1, 2, 3
PR_NONE, PR_STATIC, PR_LIMBS, PR_SPHERE_BOX = 0, 1, 2, 3
NPROP, NPRIV, HIST, NOBS, ADELAY_LEN, NREW, NMETRIC = 76, 24, 10, 860, 4, 37, 10

f32, i32 = C.c_float, C.c_int32

REWARD_TERMS = [
    "energy_square", "survive", "tracking_lin_vel_x_l1", "tracking_ang_vel_yaw_exp", "hip_action_l2",
    "foot_contacts_z", "tracking_ee_sphere", "arm_energy_abs_sum", "tracking_ee_cart", "tracking_ee_orn",
    "tracking_ee_orn_ry", "leg_energy_abs_sum", "leg_energy_sum_abs", "leg_action_l2", "leg_energy",
    "tracking_lin_vel", "tracking_lin_vel_x_exp", "tracking_ang_vel_yaw_l1", "tracking_lin_vel_y_l2",
    "tracking_lin_vel_z_l2", "torques", "collision",
    # the base class's terms (legged_robot.py:832-922) that work in the widowGo1 task
    "lin_vel_z", "ang_vel_xy", "dof_vel", "dof_acc", "action_rate", "termination", "dof_pos_limits", "dof_vel_limits", "torque_limits",
    "tracking_ang_vel", "feet_air_time", "stumble", "stand_still", "feet_contact_forces", "base_height"]
assert len(REWARD_TERMS) == NREW
METRIC_NAMES = ["leg_energy_abs_sum", "tracking_lin_vel_x_l1", "tracking_ang_vel_yaw_exp", "tracking_ee_cart",
                "tracking_ee_sphere", "tracking_ee_orn", "leg_action_l2", "torque", "energy_square",
                "foot_contacts_z"]   # WG:165


class WbcModel(C.Structure):
    _fields_ = [
        ("parent", i32 * NB), ("axis", i32 * NB), ("dof", i32 * NB),
        ("joint_xyz", (f32 * 3) * NB), ("mass", f32 * NB), ("com", (f32 * 3) * NB), ("inertia", (f32 * 6) * NB),
        ("q_lower", f32 * NDOF), ("q_upper", f32 * NDOF), ("qd_limit", f32 * NDOF), ("effort", f32 * NDOF),
        ("rb_body", i32 * NRB), ("rb_offset", (f32 * 3) * NRB), ("feet_rb", i32 * NFEET), ("gripper_rb", i32),
        ("ncp", i32), ("cp_body", i32 * NCP), ("cp_pos", (f32 * 3) * NCP), ("cp_radius", f32 * NCP),
        ("cp_rb", i32 * NCP), ("cp_kind", i32 * NCP), ("cp_body2", i32 * NCP), ("cp_rb2", i32 * NCP),
        ("cp_a", (f32 * 3) * NCP), ("cp_b", (f32 * 3) * NCP), ("cp_radius2", f32 * NCP), ("cp_sph", i32 * NCP),
        ("pr_kind", i32 * NCP), ("pr_a", i32 * NCP), ("pr_b", i32 * NCP), ("pr_reach", f32 * NCP),
        ("nlimb", i32), ("limb_s0", i32 * NLIMB), ("limb_s1", i32 * NLIMB), ("limb_radius", f32 * NLIMB), ("limb_cap0", f32 * NLIMB),
        ("limb_cap1", f32 * NLIMB), ("limb_body", i32 * NLIMB), ("limb_rb", i32 * NLIMB), ("limb_rb0", i32 * NLIMB), ("limb_rb1", i32 * NLIMB),
        ("pair_rest_offset", f32),
        ("base_piece_mass", f32), ("base_piece_com", f32 * 3), ("base_piece_inertia", f32 * 6),
        ("base_rest_mass", f32), ("base_rest_com", f32 * 3), ("base_rest_inertia", f32 * 6),
        ("gripper_body", i32),
        ("grip_piece_mass", f32), ("grip_piece_com", f32 * 3), ("grip_piece_inertia", f32 * 6),
        ("grip_rest_mass", f32), ("grip_rest_com", f32 * 3), ("grip_rest_inertia", f32 * 6),
        ("box_half", f32), ("box_mass", f32), ("box_friction", f32), ("box_sleep_speed", f32), ("box_sleep_time", f32),
    ]


class WbcTaskCfg(C.Structure):
    _fields_ = [
        ("sim_dt", f32), ("decimation", i32), ("gravity", f32 * 3),
        ("contact_margin", f32), ("contact_erp", f32), ("max_depenetration_vel", f32), ("terrain_friction", f32),
        ("limit_kappa", f32), ("limit_delta", f32), ("contact_iters", i32), ("joint_armature", f32 * NACT),
        ("clip_actions", f32), ("action_scale", f32 * NACT), ("p_gains", f32 * NACT), ("d_gains", f32 * NACT),
        ("default_dof_pos", f32 * NDOF), ("torque_limits", f32 * NDOF), ("action_delay", i32),
        ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32), ("obs_scale_dof_vel", f32), ("clip_obs", f32),
        ("commands_scale", f32 * 3),
        ("max_episode_length", i32), ("term_rp_threshold", f32), ("term_z_threshold", f32),
        ("term_contact_rb_mask", C.c_uint32), ("penalize_contact_rb_mask", C.c_uint32),
        ("resample_interval", i32), ("push_interval", i32), ("max_push_vel", f32),
        ("lin_vel_x_clip", f32), ("ang_vel_yaw_clip", f32),
        ("goal_collision_lower", f32 * 3), ("goal_collision_upper", f32 * 3), ("goal_underground_limit", f32),
        ("goal_collision_samples", i32), ("goal_delta_orn_range", (f32 * 2) * 3),
        ("sphere_error_scale", f32 * 3), ("orn_error_scale", f32 * 3), ("z_invariant_offset", f32), ("goal_command_cart", i32),
        ("tracking_sigma", f32), ("tracking_ee_sigma", f32), ("only_positive_rewards", i32),
        ("soft_dof_lower", f32 * NDOF), ("soft_dof_upper", f32 * NDOF), ("soft_dof_vel_limit", f32 * NDOF), ("soft_torque_limit", f32 * NDOF),
        ("max_contact_force", f32), ("base_height_target", f32),
        ("base_init_state", f32 * 13), ("origin_perturb_range", f32), ("init_vel_perturb_range", f32),
        ("dof_reset_lo", f32), ("dof_reset_hi", f32), ("box_origin_x", f32), ("box_origin_z", f32),
        ("ground_z", f32),
    ]


class WbcCurriculum(C.Structure):
    _fields_ = [
        ("lin_vel_x_range", f32 * 2), ("ang_vel_yaw_range", f32 * 2),
        ("goal_l_range", f32 * 2), ("goal_p_range", f32 * 2), ("goal_y_range", f32 * 2),
        ("leg_reward_scale", f32 * NREW), ("arm_reward_scale", f32 * NREW),
        ("leg_active_mask", C.c_uint64), ("arm_active_mask", C.c_uint64),
    ]


class WbcSideJob(C.Structure):
    """wbc_side_job (include/wbc_sim.h): a per-step reduction another launch carries as extra workgroups."""
    _fields_ = [("ep_done", C.c_void_p), ("met_done", C.c_void_p), ("reset_buf", C.c_void_p), ("prev", C.c_void_p),
                ("rew", C.c_void_p), ("arm_rew", C.c_void_p), ("out", C.c_void_p), ("track_state", C.c_void_p),
                ("n", i32), ("track_cap", i32), ("nblocks", i32), ("scale", f32)]


# enum wbc_tensor_id, same order as the header
TENSOR_IDS = [
    "ROOT_STATES", "DOF_STATE", "NET_CONTACT_FORCE", "RIGID_BODY_STATE", "FORCE_SENSOR", "TORQUES", "OBS_BUF",
    "OBS_HISTORY", "ACTION_HISTORY", "ACTIONS", "LAST_ACTIONS", "LAST_DOF_VEL", "LAST_ROOT_VEL", "COMMANDS",
    "GOAL_STATE", "REW_BUF", "ARM_REW_BUF", "RESET_BUF", "TIME_OUT_BUF", "EPISODE_LENGTH", "EPISODE_SUMS",
    "METRIC_SUMS", "EPISODE_SUMS_DONE", "METRIC_SUMS_DONE", "BASE_LIN_VEL", "BASE_ANG_VEL", "MASS_PARAMS",
    "FRICTION", "MOTOR_STRENGTH", "ENV_ORIGINS", "BOX_DELTA_Y", "BODY_PARAMS", "RESET_TRAVEL", "BOX_MASS", "BOX_SLEEP_TIMER", "FEET_AIR_TIME", "LAST_CONTACTS", "DROPPED_HITS"]
T = {name: i for i, name in enumerate(TENSOR_IDS)}
# per-env shapes (without the leading N) and dtypes, as the header documents them
TENSOR_SHAPES = {
    "ROOT_STATES": (2, 13), "DOF_STATE": (20, 2), "NET_CONTACT_FORCE": (28, 3), "RIGID_BODY_STATE": (28, 13),
    "FORCE_SENSOR": (4, 6), "TORQUES": (20,), "OBS_BUF": (860,), "OBS_HISTORY": (10, 76),
    "ACTION_HISTORY": (4, 18), "ACTIONS": (18,), "LAST_ACTIONS": (18,), "LAST_DOF_VEL": (20,),
    "LAST_ROOT_VEL": (6,), "COMMANDS": (3,), "GOAL_STATE": (24,), "REW_BUF": (), "ARM_REW_BUF": (),
    "RESET_BUF": (), "TIME_OUT_BUF": (), "EPISODE_LENGTH": (), "EPISODE_SUMS": (NREW,), "METRIC_SUMS": (10,),
    "EPISODE_SUMS_DONE": (NREW,), "METRIC_SUMS_DONE": (10,), "BASE_LIN_VEL": (3,), "BASE_ANG_VEL": (3,),
    "MASS_PARAMS": (5,), "FRICTION": (), "MOTOR_STRENGTH": (18,), "ENV_ORIGINS": (3,), "BOX_DELTA_Y": (),
    "BODY_PARAMS": (20,), "RESET_TRAVEL": (2,), "BOX_MASS": (), "BOX_SLEEP_TIMER": (), "FEET_AIR_TIME": (4,), "LAST_CONTACTS": (4,), "DROPPED_HITS": ()}
TENSOR_DTYPES = {name: "f32" for name in TENSOR_IDS}
TENSOR_DTYPES.update(RESET_BUF="i64", EPISODE_LENGTH="i64", TIME_OUT_BUF="u8")

DEFAULT_MODEL_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "widowgo1_model.json")


def load_default_model() -> RobotModel:
    with open(DEFAULT_MODEL_JSON) as f:
        return RobotModel.from_json(f.read())


def _set(arr, values):
    """Copy `values` into a (possibly nested) ctypes array of floats."""
    view = np.ctypeslib.as_array(arr)
    view[...] = np.asarray(values, dtype=np.float64).reshape(view.shape)


def _seti(arr, values):
    view = np.ctypeslib.as_array(arr)
    view[...] = np.asarray(values, dtype=np.int64).reshape(view.shape)


# The URDF's <collision> blocks (legged_gym/resources/robots/widowGo1/urdf/widowGo1.urdf) as this framework's primitives.
TRUNK_HALF = (0.3762 / 2, 0.0935 / 2, 0.114 / 2)        # <link name="trunk"> box
THIGH_LEN, THIGH_RADIUS = 0.213, 0.017                   # thigh box 0.213 x 0.0245 x 0.034, long axis -z of the thigh frame
CORNER_RADIUS = 0.01
CALF_LEN, CALF_RADIUS = 0.213, 0.008                     # calf box 0.213 x 0.016 x 0.016 (urdf:981), long axis -z of the calf frame
BOX_CORNER_RADIUS = 0.005
BOX_DENSITY, BOX_FRICTION = 1000.0, 1.0                  # asset_options.density (WG:322); Isaac Gym's default shape friction
BOX_SLEEP_SPEED, BOX_SLEEP_TIME = 0.01, 0.4                # m/s, s: a box at rest for 0.4 s is frozen, as PhysX puts resting actors to sleep


KNEE_RADIUS, FOOT_RADIUS, ELBOW_RADIUS, WRIST_RADIUS, GRIP_RADIUS = 0.02, 0.02, 0.025, 0.025, 0.012
HAND_RADIUS = 0.02                                        # gripper body + fingers (a 4 cm bar) as a capsule from the wrist to the tip
UPPER_ARM_LEN = math.hypot(0.25, 0.04975)                 # shoulder joint .. elbow: the L-shaped upper-arm link (urdf:512-530) as the straight capsule between its joints
FOREARM_LEN, HAND_LEN = 0.25, 0.1586                       # elbow .. wrist (0.175 + 0.075) and wrist .. gripper tip (0.065 + 0.0936) along the arm (urdf:531-700)
LIMB_RSUM_MAX = 0.060     # WBC_LIMB_RSUM_MAX (include/wbc_sim.h): bounds the radius sums of all candidate limb pairs


def _arm_limb_fit():
    """Shaft / end-sphere radii of the arm's three limbs, FITTED to the convex hulls of the STL meshes the URDF names as collision
    geometry (widowGo1.urdf:504-819; tools/fit_arm_primitives.py --apply writes assets/arm_primitives.json): the radii that minimise
    max(under-, over-approximation) of capsule + end spheres against each hull -- 28.7 / 13.7 / 24.2 mm for upper arm / forearm / hand
    (a straight capsule against the cone-like hull of an L-shaped link; the file also holds the `under <= 10 mm` variant and its
    46 / 19 / 39 mm over-approximation). Rounds 3-5 had 25 / 25 / 20 mm typed in by hand (42 mm under on the upper arm)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "arm_primitives.json")) as f:
        d = json.load(f)["limbs"]
    return {k: tuple(round(float(d[k]["balanced"][q]), 4) for q in ("radius", "cap0", "cap1")) for k in ("upper_arm", "forearm", "hand")}


ARM_LIMB_FIT = _arm_limb_fit()
STATIC_SELF_SLOT0, BOX_ROW, SHANK0 = 23, 32, 48
# dynamic slots: what the self-collision broad phase promotes its hits into (robot-vs-robot: outside the box row; robot-vs-box: inside)
SHOULDER_SLOT = 26
DYN_SELF_SLOTS = list(range(27, 32)) + list(range(52, 64))
DYN_BOX_SLOTS = [45, 46, 47]
LEGS = ("FL", "FR", "RL", "RR")


def collision_set(m: RobotModel, foot_name: str = "foot", gripper_name: str = "wx250s/ee_gripper_link", self_collisions: bool = True,
                  box_half: float = 0.05):
    """The contact list of this framework's physics spec: (slots, limbs, candidates).
    `slots`: dicts with the wbc_model cp_* fields and the contact SLOT (= wavefront lane) each one occupies. Sphere-swept stand-ins
    for the URDF's <collision> blocks (legged_gym/resources/robots/widowGo1/urdf/widowGo1.urdf):
      terrain contacts -- 4 feet (the URDF's r = 0.02 spheres), 4 knees (calf box upper ends = thigh box lower ends), gripper tip,
        elbow, wrist (the arm-link meshes), 4 thigh tops (thigh box upper ends), 8 trunk-box corners, 4 mid-shanks (the middle of
        the calf boxes: what touches a stair edge between knee and foot), the 8 corners of the free box actor;
      static pairs -- gripper / wrist / elbow spheres against the trunk box (what bounds the arm's workspace: 3-5 % of uniformly drawn
        joint configurations, tools/self_collision_reach.py), the 4 foot spheres and the gripper tip against the free box (created
        with the robot's collision filter, WG:384);
      dynamic slots -- free lanes that the broad phase promotes its hits into.
    `limbs`: the capsules of the self-collision broad phase (thighs r 0.017, calves r 0.008 with knee / foot end spheres r 0.02, the
    upper arm shoulder .. elbow and the forearm elbow .. wrist r 0.025, the hand wrist .. gripper tip r 0.02), ends given as sphere indices. `candidates`: every pair of
    them on non-adjacent links that can touch inside the joint limits (asset.self_collisions = 0 means enabled for all of them,
    widowGo1_config.py:180) -- calf-calf x 6, thigh-calf of different legs x 12, left-right thigh pairs x 2, upper arm, forearm and hand against
    the 8 leg limbs x 24 -- and the robot spheres that can meet the free box besides the static five (knees, shins, the trunk's bottom
    corners), most frequent first.
    Slots (= wavefront lanes): 0..22 the robot's spheres against the terrain, 23..25 arm vs trunk, 26 the shoulder sphere, 27..31 dynamic; 32..47 everything
    that involves the free box (corners 32..39, static pairs 40..44, dynamic 45..47: one 16-lane row, summed by a row reduction);
    48..51 the mid-shanks; 52..63 dynamic."""
    rbn = m.rb_names
    feet = [i for i, n in enumerate(rbn) if foot_name in n]
    grip_rb = rbn.index(gripper_name)
    cps = []

    def sphere(rb, pos, rad, slot=None, sph=None):
        cps.append(dict(body=int(m.rb_body[rb]), pos=np.asarray(m.rb_offset[rb], dtype=np.float64) + np.asarray(pos, dtype=np.float64),
                        radius=rad, rb=rb, kind=CP_TERRAIN, body2=-1, rb2=-1, a=np.zeros(3), b=np.zeros(3), radius2=0.0,
                        slot=len(cps) if slot is None else slot, sph=len(cps) if sph is None else sph))
        return len(cps) - 1
    for rb in feet:
        sphere(rb, np.zeros(3), FOOT_RADIUS)
    for rb in feet:                                                         # knees = calf origins
        sphere(rb - 1, np.zeros(3), KNEE_RADIUS)
    k_grip = sphere(grip_rb, np.zeros(3), GRIP_RADIUS)
    k_elbow = sphere(rbn.index("wx250s/upper_forearm_link"), np.zeros(3), ELBOW_RADIUS)
    k_wrist = sphere(rbn.index("wx250s/wrist_link"), np.zeros(3), WRIST_RADIUS)
    thighs = [rb - 2 for rb in feet]
    for rb in thighs:
        assert "thigh" in rbn[rb]
        sphere(rb, np.zeros(3), THIGH_RADIUS)
    trunk_rb = rbn.index("trunk")
    hx, hy, hz = (h - CORNER_RADIUS for h in TRUNK_HALF)
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (-1, 1):
                sphere(trunk_rb, np.array([sx * hx, sy * hy, sz * hz]), CORNER_RADIUS)
    assert len(cps) == STATIC_SELF_SLOT0
    for i, rb in enumerate(feet):                                           # mid-shank: calf rigid body, half way down the calf box
        assert "calf" in rbn[rb - 1]
        sphere(rb - 1, np.array([0.0, 0.0, -CALF_LEN / 2]), CALF_RADIUS, slot=SHANK0 + i, sph=STATIC_SELF_SLOT0 + i)
    k_shoulder = sphere(rbn.index("wx250s/upper_arm_link"), np.zeros(3), ELBOW_RADIUS, slot=SHOULDER_SLOT, sph=NSPH - 1)   # the shoulder joint (wx250s_2_shoulder / 3_upper_arm meshes)
    assert len(cps) == NSPH
    hb = box_half - BOX_CORNER_RADIUS                                       # the free box: corner spheres inset so that the surface is the cube's
    hi = BOX_ROW
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (-1, 1):
                cps.append(dict(body=BOX_BODY, pos=np.array([sx * hb, sy * hb, sz * hb]), radius=BOX_CORNER_RADIUS, rb=BOX_RB,
                                kind=CP_TERRAIN, body2=-1, rb2=-1, a=np.zeros(3), b=np.zeros(3), radius2=0.0, slot=hi, sph=-1))
                hi += 1
    limbs, cands = [], []
    if self_collisions:
        lo = STATIC_SELF_SLOT0

        def pair(k, body2, rb2, a, b, slot):
            c = dict(cps[k])
            c.update(kind=CP_BOX, body2=body2, rb2=rb2, a=np.asarray(a, dtype=np.float64), b=np.asarray(b, dtype=np.float64), radius2=0.0, slot=slot)
            cps.append(c)
        for k in (k_grip, k_wrist, k_elbow):
            pair(k, int(m.rb_body[trunk_rb]), trunk_rb, np.asarray(m.rb_offset[trunk_rb]), np.array(TRUNK_HALF), lo)
            lo += 1
        for k in list(range(len(feet))) + [k_grip]:                           # the foot spheres and the gripper tip against the free box
            pair(k, BOX_BODY, BOX_RB, np.zeros(3), np.full(3, box_half), hi)
            hi += 1
        for slot in DYN_SELF_SLOTS + DYN_BOX_SLOTS:
            cps.append(dict(body=0, pos=np.zeros(3), radius=0.0, rb=0, kind=CP_DYNAMIC, body2=-1, rb2=-1, a=np.zeros(3), b=np.zeros(3), radius2=0.0,
                            slot=slot, sph=-1))
        # limbs: ends = compact sphere indices (thigh: thigh top .. knee; calf: knee .. foot)
        for i, leg in enumerate(LEGS):
            assert rbn[feet[i]].startswith(leg)
        for i in range(4):
            limbs.append(dict(name=LEGS[i] + "_thigh", s0=11 + i, s1=4 + i, radius=THIGH_RADIUS, cap0=0.0, cap1=0.0, length=THIGH_LEN,
                              body=int(m.rb_body[thighs[i]]), rb=thighs[i], rb0=thighs[i], rb1=thighs[i]))
        for i in range(4):
            limbs.append(dict(name=LEGS[i] + "_calf", s0=4 + i, s1=i, radius=CALF_RADIUS, cap0=KNEE_RADIUS, cap1=FOOT_RADIUS, length=CALF_LEN,
                              body=int(m.rb_body[feet[i] - 1]), rb=feet[i] - 1, rb0=feet[i] - 1, rb1=feet[i]))
        # the arm against the legs: its links BETWEEN the spheres collide too -- the upper arm (shoulder joint .. elbow), the forearm (elbow ..
        # wrist: upper + lower forearm link, the roll joint turns about this very axis) and the hand (wrist .. gripper tip: wrist, gripper
        # and finger links, about the wrist-rotate axis) as capsules. (The arm's own links do not collide with each other: the one pair
        # that could, hand vs upper arm, is the stated exception.)
        rbi = {n: i for i, n in enumerate(rbn)}
        fit = ARM_LIMB_FIT
        limbs.append(dict(name="upper_arm", s0=NSPH - 1, s1=k_elbow, radius=fit["upper_arm"][0], cap0=fit["upper_arm"][1], cap1=fit["upper_arm"][2], length=UPPER_ARM_LEN,
                          body=cps[k_shoulder]["body"], rb=cps[k_shoulder]["rb"], rb0=cps[k_shoulder]["rb"], rb1=cps[k_shoulder]["rb"]))
        limbs.append(dict(name="forearm", s0=k_elbow, s1=k_wrist, radius=fit["forearm"][0], cap0=fit["forearm"][1], cap1=fit["forearm"][2], length=FOREARM_LEN,
                          body=cps[k_elbow]["body"], rb=cps[k_elbow]["rb"], rb0=cps[k_elbow]["rb"], rb1=cps[k_wrist]["rb"]))
        limbs.append(dict(name="hand", s0=k_wrist, s1=k_grip, radius=fit["hand"][0], cap0=fit["hand"][1], cap1=fit["hand"][2], length=HAND_LEN,
                          body=cps[k_grip]["body"], rb=rbi["wx250s/gripper_link"], rb0=rbi["wx250s/gripper_link"], rb1=rbi["wx250s/gripper_link"]))
        assert len(limbs) <= NLIMB
        L = {l["name"]: i for i, l in enumerate(limbs)}

        def bound(l):
            return 0.5 * l["length"] + max(l["radius"], l["cap0"], l["cap1"])

        def limb_pair(a, b):
            cands.append(dict(kind=PR_LIMBS, a=L[a], b=L[b], reach=bound(limbs[L[a]]) + bound(limbs[L[b]])))
        side = [("FL", "RL"), ("FR", "RR")]
        lr = [("FL", "FR"), ("RL", "RR")]
        diag = [("FL", "RR"), ("FR", "RL")]
        # most frequent first (tools/self_collision_reach.py): neighbouring calves, a calf against the neighbouring leg's thigh, the
        # left-right thigh pairs, then the arm against the legs, the diagonal pairs last
        for a, b in side + lr:
            limb_pair(a + "_calf", b + "_calf")
        for a, b in side + lr:
            limb_pair(a + "_calf", b + "_thigh")
            limb_pair(b + "_calf", a + "_thigh")
        for a, b in lr:
            limb_pair(a + "_thigh", b + "_thigh")
        for arm in ("upper_arm", "forearm", "hand"):
            for leg in LEGS:
                for part in ("_thigh", "_calf"):
                    limb_pair(arm, leg + part)
        for a, b in diag:
            limb_pair(a + "_calf", b + "_calf")
            limb_pair(a + "_calf", b + "_thigh")
            limb_pair(b + "_calf", a + "_thigh")
        assert len(cands) == 44
        assert max(max(limbs[c["a"]][k] for k in ("radius", "cap0", "cap1")) + max(limbs[c["b"]][k] for k in ("radius", "cap0", "cap1")) for c in cands) <= LIMB_RSUM_MAX
        # robot spheres that can meet the free box besides the five static pairs: knees, shins, the trunk box's bottom corners
        bottom = [k for k in range(15, 23) if cps[k]["pos"][2] < m.rb_offset[trunk_rb][2]]
        for k in list(range(4, 8)) + [STATIC_SELF_SLOT0 + i for i in range(4)] + bottom:
            sp = next(c for c in cps if c["sph"] == k and c["kind"] == CP_TERRAIN)
            cands.append(dict(kind=PR_SPHERE_BOX, a=k, b=0, reach=box_half * math.sqrt(3.0) + sp["radius"]))
        assert len(cands) == 56
    assert hi <= BOX_ROW + 16
    assert len({c["slot"] for c in cps}) == len(cps) and max(c["slot"] for c in cps) < NCP
    return sorted(cps, key=lambda c: c["slot"]), limbs, cands


REACH_STEP = 0.04      # pr_reach is a multiple of this (3 bits of the kernel's packed descriptor), rounded up: a bounding radius may only grow


def quantise_reach(r: float) -> float:
    code = int(math.ceil(r / REACH_STEP - 1e-9))
    assert 1 <= code <= 8, f"bounding reach {r} m does not fit the descriptor's 3 bits"
    return code * REACH_STEP


def fill_model(m: RobotModel, foot_name: str = "foot",
               gripper_name: str = "wx250s/ee_gripper_link", self_collisions: bool = True, box_size: float = 0.1,
               rest_offset: float = 0.0) -> WbcModel:
    """RobotModel -> wbc_model, with the collision set of this framework's physics spec (collision_set). `rest_offset`
    (sim.physx.rest_offset, LRC:194): the separation at which two shapes rest -- every static contact has a sphere on one side,
    so holding the surfaces `rest_offset` apart is that sphere grown by it (fill_task_cfg shrinks the contact margin by the same);
    limb pairs subtract it from their gap (pair_rest_offset)."""
    assert m.nb == NB and m.num_dofs == NDOF and m.num_rigid_bodies == NRB
    out = WbcModel()
    _seti(out.parent, m.parent)
    _seti(out.axis, m.axis)
    _seti(out.dof, m.body_dof)
    _set(out.joint_xyz, m.joint_xyz)
    _set(out.mass, m.mass)
    _set(out.com, m.com)
    _set(out.inertia, m.inertia)
    lo, hi = m.dof_lower.copy(), m.dof_upper.copy()
    _set(out.q_lower, lo)
    _set(out.q_upper, hi)
    _set(out.qd_limit, m.dof_velocity)
    _set(out.effort, m.dof_effort)
    _seti(out.rb_body, m.rb_body)
    _set(out.rb_offset, m.rb_offset)
    feet = [i for i, n in enumerate(m.rb_names) if foot_name in n]          # WG:297
    assert len(feet) == NFEET
    _seti(out.feet_rb, feet)
    out.gripper_rb = m.rb_names.index(gripper_name)                         # WG:318
    cps, limbs, cands = collision_set(m, foot_name, gripper_name, self_collisions, box_half=0.5 * box_size)
    out.box_half, out.box_mass, out.box_friction = 0.5 * box_size, BOX_DENSITY * box_size ** 3, BOX_FRICTION
    out.box_sleep_speed, out.box_sleep_time = BOX_SLEEP_SPEED, BOX_SLEEP_TIME
    out.ncp = max(c["slot"] for c in cps) + 1                              # slots in use: 0 .. ncp-1, unused ones marked kind = -1
    for k in range(NCP):
        out.cp_body2[k] = out.cp_rb2[k] = out.cp_sph[k] = -1
        out.cp_kind[k] = CP_NONE
        out.pr_kind[k] = PR_NONE
    for c in cps:
        k = c["slot"]
        out.cp_body[k], out.cp_rb[k], out.cp_kind[k] = c["body"], c["rb"], c["kind"]
        out.cp_body2[k], out.cp_rb2[k] = c["body2"], c["rb2"]
        out.cp_radius[k], out.cp_radius2[k] = c["radius"] + (float(rest_offset) if c["kind"] != CP_DYNAMIC else 0.0), c["radius2"]
        out.cp_sph[k] = c["sph"]
        for j in range(3):
            out.cp_pos[k][j], out.cp_a[k][j], out.cp_b[k][j] = float(c["pos"][j]), float(c["a"][j]), float(c["b"][j])
        if c["kind"] == CP_BOX:                                            # the lane's own pair is what it tests in the broad phase
            out.pr_kind[k], out.pr_a[k], out.pr_b[k] = PR_STATIC, c["sph"], 0
            out.pr_reach[k] = quantise_reach(float(np.linalg.norm(c["b"])) + out.cp_radius[k])
    out.nlimb = len(limbs)
    for i, l in enumerate(limbs):
        out.limb_s0[i], out.limb_s1[i], out.limb_radius[i], out.limb_cap0[i], out.limb_cap1[i] = l["s0"], l["s1"], l["radius"], l["cap0"], l["cap1"]
        out.limb_body[i], out.limb_rb[i], out.limb_rb0[i], out.limb_rb1[i] = l["body"], l["rb"], l["rb0"], l["rb1"]
    out.pair_rest_offset = float(rest_offset)
    lanes = [k for k in range(NCP) if out.pr_kind[k] != PR_STATIC]          # every lane without a static pair of its own tests a candidate
    assert len(cands) <= len(lanes)
    for k, c in zip(lanes, cands):
        out.pr_kind[k], out.pr_a[k], out.pr_b[k] = c["kind"], c["a"], c["b"]
        out.pr_reach[k] = quantise_reach(c["reach"] + float(rest_offset))
    bp, gp = m.base_piece, m.gripper_piece
    out.base_piece_mass = bp["mass"]
    _set(out.base_piece_com, bp["com"])
    _set(out.base_piece_inertia, bp["inertia"])
    out.base_rest_mass = bp["rest_mass"]
    _set(out.base_rest_com, bp["rest_com"])
    _set(out.base_rest_inertia, bp["rest_inertia"])
    out.gripper_body = int(gp["body"])
    out.grip_piece_mass = gp["mass"]
    _set(out.grip_piece_com, gp["com"])
    _set(out.grip_piece_inertia, gp["inertia"])
    out.grip_rest_mass = gp["rest_mass"]
    _set(out.grip_rest_com, gp["rest_com"])
    _set(out.grip_rest_inertia, gp["rest_inertia"])
    return out


def _get(cfg, path, default=None):
    o = cfg
    for name in path.split("."):
        if not hasattr(o, name):
            return default
        o = getattr(o, name)
    return o


# Config switches the reference READS (WG = envs/widowGo1/widowGo1.py) whose non-default value this framework does not implement:
# (path, supported value(s), why). Flipping one raises instead of being silently ignored (tests/test_host_logic.py flips each).
UNSUPPORTED_SWITCHES = [
    ("control.adaptive_arm_gains", (False,), "dead in the reference as well: step() reorders the actions through an 18-index table (WG:1162, "
     "raisim2ig_wo_gripper) which drops the six gain columns, then _compute_torques slices actions[:, 18:] = [N, 0] and adds it to the six "
     "arm gains (WG:1274,1285: a shape error); action_history_buf is [N, 4, 18] (WG:540)"),
    ("env.reorder_dofs", (True,), "the fused step hard-wires the policy <-> simulator joint order of WG:1003-1088"),
    ("domain_rand.observe_priv", (True,), "dead in the reference as well: obs_buf is only ever assigned inside `if observe_priv` (WG:986-992), so "
     "without it the policy sees the zero-initialised buffer forever (profiles/r04_reference_switches.txt)"),
    ("asset.fix_base_link", (False,), "the physics spec is a floating base (DESIGN.md section 3)"),
    ("asset.disable_gravity", (False,), "not modelled; set sim.gravity instead"),
    ("asset.collapse_fixed_joints", (True,), "the rigid-body list (27 bodies, quirk Q1) is the collapsed one"),
    ("asset.default_dof_drive_mode", (3,), "the task drives joints by effort (WG:1183); position / velocity drive modes are not modelled"),
    ("asset.linear_damping", (0.0,), "body damping is not modelled"),
    ("asset.angular_damping", (0.0,), "body damping is not modelled"),
    ("terrain.restitution", (0.0,), "contacts are inelastic (restitution 0, tests/test_oracle_contact_physics.py)"),
    ("sim.substeps", (1,), "one solver step per gym.simulate (LRC:184)"),
    ("sim.up_axis", (1,), "z is up (LRC:186)"),
    ("terrain.selected", (False,), "a single generator chosen by name: broken in the reference too (utils/terrain.py:160-173 reads attributes the class never sets)"),
    ("sim.physx.solver_type", (1,), "the contact solver is this framework's own damped block-Jacobi (DESIGN.md section 3): the shipped value (1 = TGS, "
     "LRC:190) selects it, PhysX's PGS (0) is not offered"),
    ("sim.physx.num_velocity_iterations", (0,), "the solver has one kind of sweep (sim.physx.num_position_iterations of them); extra velocity-only iterations "
     "are not modelled (the reference ships 0, LRC:192)"),
    ("env.num_privileged_obs", (None,), "the reference hands the critic an all-zero privileged_obs_buf then (BT:77-80: allocated, never written by WG); "
     "the fused rollout has no such buffer"),
]
# Fields that are compiled into the kernels (observation layout of WG:966-1001, the 18-action / 3-command interface): a config that
# says otherwise would make the policy built from it (OPR:59-70) and the fused step disagree -- ValueError, not a silent mismatch.
KERNEL_CONSTANTS = {"env.num_proprio": NPROP, "env.num_priv": NPRIV, "env.history_len": HIST, "env.num_observations": NOBS,
                    "env.num_actions": NACT, "env.num_torques": NACT, "commands.num_commands": 3}
# Switches the reference reads but that have NO effect in its widowGo1 task either (the code that would use them is commented out or
# overridden): ignoring them is the faithful behaviour. (path, where the reference drops it)
REFERENCE_NO_OPS = [
    ("noise.add_noise", "WG:63 stores it, WG:616 builds noise_scale_vec, compute_observations (WG:966-1001) never applies either"),
    ("commands.heading_command", "WG:927-930 commented out"),
    ("commands.curriculum", "WG:711-712 commented out (update_command_curriculum is called by the runner instead, OPR:126)"),
    ("domain_rand.randomize_arm_ema", "WG:410-413,1170 commented out"),
    ("control.control_type", "WidowGo1._compute_torques (WG:1262-1295) is a PD law whatever the value; only the base class reads it"),
    ("box.box_pos_obs_range", "update_target_ee_base commented out (WG:1244-1259)"),
    ("arm.grasp_offset", "update_target_ee_base commented out (WG:1244-1259)"),
    ("arm.init_target_ee_base", "update_target_ee_base commented out (WG:1244-1259)"),
    ("termination.r_threshold", "check_termination hard-codes 0.2 rad (WG:945-946)"),
    ("termination.p_threshold", "check_termination hard-codes 0.2 rad (WG:945-946)"),
    ("asset.flip_visual_attachments", "visual only"),
    ("asset.max_linear_velocity", "1000 m/s: never reached"),
    ("asset.max_angular_velocity", "1000 rad/s: never reached (URDF joint velocity limits are enforced, DESIGN.md section 3)"),
    ("asset.thickness", "PhysX shape-thickness hint of the importer; the contact offset is sim.physx.contact_offset"),
    ("asset.density", "the URDF gives every link its mass"),
    ("asset.replace_cylinder_with_capsule", "the collision set uses sphere-swept primitives throughout"),
    ("env.env_spacing", "custom origins (WG:207-224)"),
    ("viewer.pos", "headless"), ("viewer.lookat", "headless"), ("viewer.ref_env", "headless"),
    ("noise.noise_level", "WG:65-75 folds it into noise_scale_vec, which compute_observations (WG:966-1001) never applies"),
    ("noise.noise_scales.dof_pos", "as noise.noise_level"), ("noise.noise_scales.dof_vel", "as noise.noise_level"),
    ("noise.noise_scales.lin_vel", "as noise.noise_level"), ("noise.noise_scales.ang_vel", "as noise.noise_level"),
    ("noise.noise_scales.gravity", "as noise.noise_level"), ("noise.noise_scales.height_measurements", "as noise.noise_level"),
    ("normalization.obs_scales.height_measurements", "only the noise vector (WG:75, never applied) and the base class's overridden compute_observations read it"),
    ("commands.max_curriculum", "the base class's update_command_curriculum (LR:443-452) is overridden (WG:678-692)"),
    ("commands.ranges.lin_vel_x", "the base class's ranges: WG:85-90 reads only the init_* / final_* keys of commands.ranges"),
    ("commands.ranges.lin_vel_y", "as commands.ranges.lin_vel_x (cmd_y is always 0, WG:838)"),
    ("commands.ranges.ang_vel_yaw", "as commands.ranges.lin_vel_x"), ("commands.ranges.heading", "as commands.ranges.lin_vel_x"),
    ("goal_ee.num_commands", "never read (WG:630 commented out)"),
    ("goal_ee.init_ranges.pos_l", "WG:1322 commented out"), ("goal_ee.init_ranges.pos_p", "WG:1322 commented out"), ("goal_ee.init_ranges.pos_y", "WG:1322 commented out"),
    ("domain_rand.cube_y_range", "never read (the box's y offset comes from box.box_env_origins_y_range, WG:226-227)"),
    ("terrain.terrain_kwargs", "only read under terrain.selected (utils/terrain.py:158-169), which is refused"),
    ("terrain.slope_treshold", "only moves vertices of the TRIMESH (utils/terrain.py:60,136: cells steeper than the threshold become vertical walls); contact here is "
     "against the height grid itself, where such a cell stays a one-cell ramp -- a stated deviation (INTEGRATION.md section 4); the shipped widowGo1 value "
     "(1e8, WGC:310) never triggers it"),
    ("terrain.add_slopes", "never read (utils/terrain.py:40-99)"), ("terrain.slope_incline", "never read (utils/terrain.py:40-99)"),
    # settings that size PhysX's buffers / pick its threads / its contact reporting: no counterpart, no numerical effect
    ("sim.physx.num_threads", "PhysX CPU worker threads (LRC:189)"), ("sim.physx.max_gpu_contact_pairs", "PhysX buffer size (LRC:197)"),
    ("sim.physx.default_buffer_size_multiplier", "PhysX buffer size (LRC:198)"),
    ("sim.physx.contact_collection", "when PhysX refreshes its contact report (LRC:199: 2 = every substep); net_contact_force here is the last substep's, as with 2"),
    ("sim.physx.bounce_threshold_velocity", "the speed above which PhysX applies restitution: every material here has restitution 0 (terrain.restitution "
     "other than 0 is refused), so nothing ever bounces at any threshold"),
]


def unsupported_switches(cfg):
    """[(path, value, reason)] for every switch of UNSUPPORTED_SWITCHES that `cfg` flips away from what is implemented."""
    bad = []
    for path, ok, why in UNSUPPORTED_SWITCHES:
        v = _get(cfg, path, ok[0])
        if ok[0] is None:
            if v is not None:
                bad.append((path, v, why))
            continue
        if isinstance(v, bool) or isinstance(ok[0], bool):
            v = bool(v)
        if v not in ok:
            bad.append((path, v, why))
    t = getattr(cfg, "terrain", None)
    if t is not None and float(getattr(t, "dynamic_friction", getattr(t, "static_friction", 1.0))) != float(getattr(t, "static_friction", 1.0)):
        bad.append(("terrain.dynamic_friction", t.dynamic_friction, "one Coulomb coefficient per contact (static = dynamic), DESIGN.md section 3"))
    return bad


def constant_mismatches(cfg):
    """[(path, value, compiled-in value)] for every field of KERNEL_CONSTANTS that `cfg` sets to something else."""
    return [(p, _get(cfg, p, want), want) for p, want in KERNEL_CONSTANTS.items() if _get(cfg, p, want) != want]


def set_soft_limits(out: "WbcTaskCfg", m: RobotModel, soft_pos: float, soft_vel: float, soft_torque: float, max_contact_force: float,
                    base_height_target: float) -> None:
    """What the base class's limit rewards compare against: LR:294-304 (_process_dof_props: centre +- half range * soft_dof_pos_limit),
    LR:882 (velocity limit * soft_dof_vel_limit), LR:886 (torque limit * soft_torque_limit), LR:922, LR:848."""
    for i in range(NDOF):
        lo, hi = float(m.dof_lower[i]), float(m.dof_upper[i])
        mid, rng = 0.5 * (lo + hi), hi - lo
        out.soft_dof_lower[i] = mid - 0.5 * rng * soft_pos
        out.soft_dof_upper[i] = mid + 0.5 * rng * soft_pos
        out.soft_dof_vel_limit[i] = float(m.dof_velocity[i]) * soft_vel
        out.soft_torque_limit[i] = float(m.dof_effort[i]) * soft_torque
    out.max_contact_force, out.base_height_target = max_contact_force, base_height_target


def fill_task_cfg(cfg, m: RobotModel, sim_dt: Optional[float] = None, check: bool = True) -> WbcTaskCfg:
    """WidowGo1RoughCfg (+ LeggedRobotCfg.sim) -> wbc_task_cfg, resolving names to numbers the way
    WidowGo1._parse_cfg / _init_buffers do (WG:78-121, 498-672). Raises NotImplementedError for a switch the reference reads
    and this framework does not implement (UNSUPPORTED_SWITCHES): nothing is silently ignored."""
    bad = unsupported_switches(cfg) if check else []      # (check=False: the reference harness, whose physics backend ignores the task switches)
    if bad:
        raise NotImplementedError("config switches this framework does not implement: " +
                                  "; ".join(f"{p} = {v!r} ({why})" for p, v, why in bad))
    wrong = constant_mismatches(cfg) if check else []
    if wrong:
        raise ValueError("config fields that are compiled into the fused step and the policy kernels (observation layout WG:966-1001, 18 actions, "
                         "3 commands): " + "; ".join(f"{p} = {v!r}, must be {want}" for p, v, want in wrong))
    out = WbcTaskCfg()
    dt = float(cfg.sim.dt if sim_dt is None else sim_dt)
    out.sim_dt = dt
    out.decimation = int(cfg.control.decimation)
    _set(out.gravity, cfg.sim.gravity)
    px = cfg.sim.physx
    # sim.physx.rest_offset (LRC:194): shapes rest that far apart -- fill_model grows every contact sphere by it, the margin within
    # which a contact is generated (contact_offset, measured between the shapes) shrinks by the same
    rest = float(_get(px, "rest_offset", 0.0))
    if not float(px.contact_offset) > rest:
        raise ValueError(f"sim.physx.contact_offset ({px.contact_offset}) must exceed sim.physx.rest_offset ({rest}), as PhysX requires")
    out.contact_margin = float(px.contact_offset) - rest
    out.contact_erp = 0.2
    out.max_depenetration_vel = float(px.max_depenetration_velocity)
    out.terrain_friction = float(cfg.terrain.static_friction)
    out.limit_kappa, out.limit_delta = 0.25, 0.5
    # sim.physx.num_position_iterations (LRC:191, the reference ships 4): sweeps of the contact solver per substep
    out.contact_iters = int(_get(px, "num_position_iterations", 4))
    if out.contact_iters < 1:
        raise ValueError("sim.physx.num_position_iterations must be >= 1")
    out.clip_actions = float(cfg.normalization.clip_actions)
    _set(out.action_scale, cfg.control.action_scale)
    names = m.dof_names
    for i in range(NACT):                                                    # WG:648-660
        kp = kd = 0.0
        for key in cfg.control.stiffness.keys():
            if key in names[i]:
                kp, kd = cfg.control.stiffness[key], cfg.control.damping[key]
        out.p_gains[i], out.d_gains[i] = kp, kd
        out.joint_armature[i] = dt * kd + dt * dt * kp + float(_get(cfg, "asset.armature", 0.0))   # + the importer's armature (LRC:115)
    for i in range(NDOF):                                                    # WG:642-646
        out.default_dof_pos[i] = float(cfg.init_state.default_joint_angles[names[i]])
        out.torque_limits[i] = float(m.dof_effort[i])                        # LR:294-299
    out.action_delay = int(cfg.env.action_delay)
    sc = cfg.normalization.obs_scales
    out.obs_scale_ang_vel, out.obs_scale_dof_pos, out.obs_scale_dof_vel = sc.ang_vel, sc.dof_pos, sc.dof_vel
    out.clip_obs = float(cfg.normalization.clip_observations)
    _set(out.commands_scale, [sc.lin_vel, sc.lin_vel, sc.ang_vel])           # WG:628
    policy_dt = out.decimation * dt                                          # WG:80
    out.max_episode_length = int(math.ceil(cfg.env.episode_length_s / policy_dt))   # WG:117-118
    out.term_rp_threshold = 0.2                                              # WG:945-946 (hard-coded)
    out.term_z_threshold = float(cfg.termination.z_threshold)
    def rb_mask(names):                                                       # WG:299-306: substring match on the body names
        mask = 0
        for name in names:
            for i, rbn in enumerate(m.rb_names):
                if name in rbn:
                    mask |= 1 << i
        return mask
    out.term_contact_rb_mask = rb_mask(cfg.asset.terminate_after_contacts_on)
    out.penalize_contact_rb_mask = rb_mask(cfg.asset.penalize_contacts_on)
    out.resample_interval = int(cfg.commands.resampling_time / policy_dt)    # WG:922
    out.push_interval = int(math.ceil(cfg.domain_rand.push_interval_s / policy_dt)) if cfg.domain_rand.push_robots else 0
    out.max_push_vel = float(cfg.domain_rand.max_push_vel_xy)
    out.lin_vel_x_clip = float(cfg.commands.lin_vel_x_clip)
    out.ang_vel_yaw_clip = float(cfg.commands.ang_vel_yaw_clip)
    g = cfg.goal_ee
    _set(out.goal_collision_lower, g.collision_lower_limits)
    _set(out.goal_collision_upper, g.collision_upper_limits)
    out.goal_underground_limit = float(g.underground_limit)
    out.goal_collision_samples = int(g.num_collision_check_samples)
    _set(out.goal_delta_orn_range, g.ranges.final_delta_orn)
    _set(out.sphere_error_scale, g.sphere_error_scale)
    _set(out.orn_error_scale, g.orn_error_scale)
    out.z_invariant_offset = 0.53                                            # WG:597
    mode = getattr(cfg.goal_ee, "command_mode", "sphere")
    if mode not in ("cart", "sphere"):                                       # WG:589 asserts the same
        raise ValueError(f"goal_ee.command_mode = {mode!r}: 'cart' or 'sphere'")
    out.goal_command_cart = 1 if mode == "cart" else 0                       # which goal curr_ee_goal is bound to, WG:589-593
    out.tracking_sigma = float(cfg.rewards.tracking_sigma)
    out.tracking_ee_sigma = float(cfg.rewards.tracking_ee_sigma)
    out.only_positive_rewards = int(bool(cfg.rewards.only_positive_rewards))
    rw = cfg.rewards
    set_soft_limits(out, m, float(_get(rw, "soft_dof_pos_limit", 1.0)), float(_get(rw, "soft_dof_vel_limit", 1.0)),
                    float(_get(rw, "soft_torque_limit", 1.0)), float(_get(rw, "max_contact_force", 100.0)), float(_get(rw, "base_height_target", 0.25)))
    st = cfg.init_state
    _set(out.base_init_state, list(st.pos) + list(st.rot) + list(st.lin_vel) + list(st.ang_vel))   # WG:341
    out.origin_perturb_range = float(cfg.terrain.origin_perturb_range)
    out.init_vel_perturb_range = float(cfg.terrain.init_vel_perturb_range)
    out.dof_reset_lo, out.dof_reset_hi = 0.8, 1.2                            # WG:824
    out.box_origin_x = float(cfg.box.box_env_origins_x)
    out.box_origin_z = float(cfg.box.box_env_origins_z)
    out.ground_z = 0.0
    return out


def body_params_from_randomisation(m: RobotModel, base_dmass, base_dcom, gripper_dmass) -> np.ndarray:
    """Per-env composite (mass, com3, inertia6) of moving body 0 and of the gripper body after the
    mass randomisation of WG:431-456. Spec decision (Isaac Gym's recomputeInertia semantics are not
    observable here): the added mass is a point mass at the piece's (shifted) centre of mass and the
    piece's rotational inertia about its centre of mass is unchanged."""
    base_dmass = np.asarray(base_dmass, dtype=np.float64)
    n = base_dmass.shape[0]
    bp, gp = m.base_piece, m.gripper_piece
    M0, C0, I0 = merge_piece(bp["rest_mass"], bp["rest_com"], bp["rest_inertia"],
                             bp["mass"] + base_dmass, bp["com"][None, :] + np.asarray(base_dcom, dtype=np.float64),
                             np.broadcast_to(bp["inertia"], (n, 6)))
    gd = np.asarray(gripper_dmass, dtype=np.float64)
    M1, C1, I1 = merge_piece(gp["rest_mass"], gp["rest_com"], gp["rest_inertia"],
                             gp["mass"] + gd, np.broadcast_to(gp["com"], (n, 3)),
                             np.broadcast_to(gp["inertia"], (n, 6)))
    out = np.concatenate([M0[:, None], C0, I0, M1[:, None], C1, I1], axis=1)
    return out.astype(np.float32)


DEFAULT_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "widowgo1_default.wbcasset")
ASSET_MAGIC = b"WBCASSET1\0"


def asset_bytes(m: RobotModel, cfg) -> bytes:
    """The binary asset wbc_asset_load reads (include/wbc_sim.h): the structs this module builds from a RobotModel and a config --
    wbc_model, wbc_task_cfg, wbc_curriculum at update counter 0 and 1 -- and the DoF / rigid-body names: everything a binding that
    is not this package needs to create a sim (tools/make_asset.py writes the packaged widowGo1 one)."""
    import struct
    from .curriculum import make_curriculum
    wm = fill_model(m, foot_name=cfg.asset.foot_name, self_collisions=int(_get(cfg, "asset.self_collisions", 0)) == 0, box_size=float(cfg.box.box_size),
                    rest_offset=float(_get(cfg, "sim.physx.rest_offset", 0.0)))
    tc = fill_task_cfg(cfg, m)
    parts = [ASSET_MAGIC, struct.pack("<5I", C.sizeof(WbcModel), C.sizeof(WbcTaskCfg), C.sizeof(WbcCurriculum), NDOF, NRB),
             bytes(wm), bytes(tc), bytes(make_curriculum(cfg, 0)), bytes(make_curriculum(cfg, 1))]
    for name in list(m.dof_names) + list(m.rb_names):
        b = name.encode()
        assert len(b) < 64
        parts.append(b.ljust(64, b"\0"))
    return b"".join(parts)
