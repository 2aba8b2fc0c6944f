"""Brute-force geometry of the robot's self-collision primitives (numpy, vectorised over poses; no code shared with the oracle or the
kernel): world poses of the rigid bodies -> signed distances of every pair of primitives on non-adjacent links. Primitives
(abi.collision_set's stand-ins for the URDF's <collision> blocks): trunk box; thigh capsules (r 0.017, hip-side end to knee); calf
capsules (r 0.008, knee to foot) with their end spheres (knee r 0.02, foot r 0.02); arm spheres (elbow 0.025, wrist 0.025, gripper tip
0.012). Used by tests/test_oracle_physics.py (the oracle's broad phase + promotion + exact tests against it) and by
tools/self_collision_reach.py (which pairs can touch inside the joint limits)."""
import numpy as np

from wbc_amd import abi

LEGS = ("FL", "FR", "RL", "RR")


def rotm(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def seg_seg(a0, a1, b0, b1):
    """closest distance between segments (degenerate ones included), vectorised"""
    d1, d2, r = a1 - a0, b1 - b0, a0 - b0
    a, e, f = (d1 * d1).sum(-1), (d2 * d2).sum(-1), (d2 * r).sum(-1)
    c, b = (d1 * r).sum(-1), (d1 * d2).sum(-1)
    den = a * e - b * b
    sa, se = np.maximum(a, 1e-30), np.maximum(e, 1e-30)
    s = np.where(den > 1e-12, np.clip((b * f - c * e) / np.maximum(den, 1e-30), 0, 1), 0.0)
    t = (b * s + f) / se
    s = np.where(t < 0, np.clip(-c / sa, 0, 1), np.where(t > 1, np.clip((b - c) / sa, 0, 1), s))
    s = np.where(a < 1e-12, 0.0, s)
    t = np.clip(np.where(e < 1e-12, 0.0, (b * s + f) / se), 0, 1)
    return np.linalg.norm((a0 + d1 * s[:, None]) - (b0 + d2 * t[:, None]), axis=-1)


def pt_seg(p, a0, a1):
    return seg_seg(p, p, a0, a1)


def primitives(rb, names):
    """limbs {name: (end0, end1, shaft r, cap r at end0, cap r at end1)}, arm spheres {name: (centre, r)}, trunk (R, p, half)."""
    P = {n: rb[:, i, :3] for i, n in enumerate(names)}
    limbs = {}
    for l in LEGS:
        limbs[l + "_thigh"] = (P[l + "_thigh"], P[l + "_calf"], abi.THIGH_RADIUS, 0.0, 0.0)
        limbs[l + "_calf"] = (P[l + "_calf"], P[l + "_foot"], abi.CALF_RADIUS, abi.KNEE_RADIUS, abi.FOOT_RADIUS)
    arm = {"elbow": (P["wx250s/upper_forearm_link"], abi.ELBOW_RADIUS), "wrist": (P["wx250s/wrist_link"], abi.WRIST_RADIUS),
           "gripper": (P["wx250s/ee_gripper_link"], abi.GRIP_RADIUS)}
    # the arm's links between those spheres, as the legs meet them: upper arm (shoulder joint .. elbow: the L-shaped link as the straight
    # capsule between its two joints), forearm and hand capsules
    # (the arm limbs' radii: fitted to the URDF meshes' convex hulls, tools/fit_arm_primitives.py -> abi.ARM_LIMB_FIT)
    limbs["upper_arm"] = (P["wx250s/upper_arm_link"], P["wx250s/upper_forearm_link"]) + abi.ARM_LIMB_FIT["upper_arm"]
    limbs["forearm"] = (P["wx250s/upper_forearm_link"], P["wx250s/wrist_link"]) + abi.ARM_LIMB_FIT["forearm"]
    limbs["hand"] = (P["wx250s/wrist_link"], P["wx250s/ee_gripper_link"]) + abi.ARM_LIMB_FIT["hand"]
    ti = names.index("trunk")
    return limbs, arm, (rotm(rb[:, ti, 3:7]), rb[:, ti, :3], np.array(abi.TRUNK_HALF))


def limb_limb(A, B):
    a0, a1, ra, ca0, ca1 = A
    b0, b1, rb_, cb0, cb1 = B
    g = seg_seg(a0, a1, b0, b1) - ra - rb_
    for p, c in ((a0, ca0), (a1, ca1)):
        if c > 0:
            g = np.minimum(g, pt_seg(p, b0, b1) - c - rb_)
            for q, c2 in ((b0, cb0), (b1, cb1)):
                if c2 > 0:
                    g = np.minimum(g, np.linalg.norm(p - q, axis=-1) - c - c2)
    for q, c2 in ((b0, cb0), (b1, cb1)):
        if c2 > 0:
            g = np.minimum(g, pt_seg(q, a0, a1) - c2 - ra)
    return g


def sphere_limb(c, r, L):
    return limb_limb((c, c, r, 0.0, 0.0), L)


def sphere_box(c, r, trunk):
    Rt, pt, half = trunk
    loc = np.einsum("nji,nj->ni", Rt, c - pt)
    cl = np.clip(loc, -half, half)
    out = np.linalg.norm(loc - cl, axis=-1)
    inside = np.all(np.abs(loc) < half, axis=-1)
    depth = (half - np.abs(loc)).min(-1)
    return np.where(inside, -depth, out) - r


def seg_box(a0, a1, r, trunk, k=9):
    """capsule against the trunk box: the minimum over k spheres along the segment (a bound good to a few mm for these lengths)"""
    return np.min([sphere_box(a0 + (a1 - a0) * t, r, trunk) for t in np.linspace(0, 1, k)], axis=0)


ARM_LIMBS = ("upper_arm", "forearm", "hand")


def all_pairs(rb, names):
    """{(prim a, prim b): gaps [n]} for every pair of primitives on non-adjacent links (limb names 'FL_thigh', ..., 'upper_arm',
    'forearm', 'hand'; the arm spheres against 'trunk')."""
    limbs, arm, trunk = primitives(rb, names)
    out = {}
    ln = list(limbs)
    arm_limbs = ARM_LIMBS
    adjacent = {frozenset(("upper_arm", "forearm")), frozenset(("forearm", "hand"))}
    for i in range(len(ln)):
        for j in range(i + 1, len(ln)):
            if (ln[i][:2] == ln[j][:2] and ln[i] not in arm_limbs) or frozenset((ln[i], ln[j])) in adjacent:
                continue                                                      # thigh and calf of one leg, neighbouring arm links: adjacent (filtered)
            out[(ln[i], ln[j])] = limb_limb(limbs[ln[i]], limbs[ln[j]])
    for s, (c, r) in arm.items():
        out[(s, "trunk")] = sphere_box(c, r, trunk)
    for k in ln:
        if k in arm_limbs:
            continue
        a0, a1, rr, c0, c1 = limbs[k]
        g = seg_box(a0, a1, rr, trunk)
        for q, cc in ((a0, c0), (a1, c1)):
            if cc > 0:
                g = np.minimum(g, sphere_box(q, cc, trunk))
        out[(k, "trunk")] = g
    return out
