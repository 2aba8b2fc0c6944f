"""URDF -> flat articulated-body model for the widowGo1 hot path.

Replaces what the reference gets from Isaac Gym's asset importer
(`gym.load_asset` + `get_asset_*`, reference
legged_gym/envs/widowGo1/widowGo1.py:285-294) with a host-side loader that
produces the flat arrays the HIP kernels and the C oracle consume.

Conventions reproduced from the reference's use of the importer (SURVEY.md
section 8a, quirk Q1):
  * children are visited depth-first in alphabetical order of the child link
    name, which yields the DoF order [FL,FR,RL,RR]x(hip,thigh,calf), waist,
    shoulder, elbow, forearm_roll, wrist_angle, wrist_rotate, left_finger,
    right_finger that widowGo1.py:529,557,1004-1005 relies on;
  * `collapse_fixed_joints=True` merges fixed-joint children into their parent
    unless the joint carries `dont_collapse="true"`; that decides the rigid
    body list (27 bodies: base, trunk, 4x(hip,thigh,calf,foot), 6 arm links,
    ee_gripper_link, 2 fingers).

Dynamics model (this framework's physics spec, not Isaac Gym's): every link
that is rigidly attached to a moving link (fixed joint, collapsed or not, and
the two prismatic finger joints whose URDF friction of 1000 N locks them) is
merged into one composite rigid body, giving 19 moving bodies: the floating
root plus 18 revolute joints.
"""
from __future__ import annotations

import json
import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

AXIS_NAMES = "xyz"


def _floats(s: Optional[str], n: int = 3) -> np.ndarray:
    if s is None:
        return np.zeros(n)
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def rpy_to_mat(rpy: np.ndarray) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


@dataclass
class _Link:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about com, link axes


@dataclass
class _Joint:
    name: str
    jtype: str
    parent: str
    child: str
    xyz: np.ndarray
    rpy: np.ndarray
    axis: np.ndarray
    lower: float
    upper: float
    velocity: float
    effort: float
    friction: float
    dont_collapse: bool


def parse_urdf(path: str):
    root = ET.parse(path).getroot()
    links: Dict[str, _Link] = {}
    for le in root.findall("link"):
        lk = _Link(le.attrib["name"])
        ie = le.find("inertial")
        if ie is not None:
            lk.mass = float(ie.find("mass").attrib["value"])
            oe = ie.find("origin")
            xyz = _floats(oe.attrib.get("xyz") if oe is not None else None)
            rpy = _floats(oe.attrib.get("rpy") if oe is not None else None)
            a = ie.find("inertia").attrib
            I = np.array([[float(a["ixx"]), float(a["ixy"]), float(a["ixz"])],
                          [float(a["ixy"]), float(a["iyy"]), float(a["iyz"])],
                          [float(a["ixz"]), float(a["iyz"]), float(a["izz"])]])
            R = rpy_to_mat(rpy)
            lk.com = xyz
            lk.inertia = R @ I @ R.T
        links[lk.name] = lk
    joints: List[_Joint] = []
    for je in root.findall("joint"):
        oe = je.find("origin")
        ae = je.find("axis")
        lim = je.find("limit")
        dyn = je.find("dynamics")
        joints.append(_Joint(
            name=je.attrib["name"], jtype=je.attrib["type"],
            parent=je.find("parent").attrib["link"], child=je.find("child").attrib["link"],
            xyz=_floats(oe.attrib.get("xyz") if oe is not None else None),
            rpy=_floats(oe.attrib.get("rpy") if oe is not None else None),
            axis=_floats(ae.attrib["xyz"]) if ae is not None else np.array([1.0, 0, 0]),
            lower=float(lim.attrib.get("lower", "0")) if lim is not None else 0.0,
            upper=float(lim.attrib.get("upper", "0")) if lim is not None else 0.0,
            velocity=float(lim.attrib.get("velocity", "0")) if lim is not None else 0.0,
            effort=float(lim.attrib.get("effort", "0")) if lim is not None else 0.0,
            friction=float(dyn.attrib.get("friction", "0")) if dyn is not None else 0.0,
            dont_collapse=je.attrib.get("dont_collapse", "false") == "true",
        ))
    return links, joints


def _merge(m1, c1, I1, m2, c2, I2):
    """Composite of two rigid bodies (mass, com, inertia-about-com) in one frame."""
    m = m1 + m2
    if m <= 0.0:
        return 0.0, np.zeros(3), I1 + I2
    c = (m1 * c1 + m2 * c2) / m

    def shift(I, mm, d):
        return I + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, shift(I1, m1, c1 - c) + shift(I2, m2, c2 - c)


@dataclass
class RobotModel:
    """Flat arrays. Moving bodies are indexed 0..nb-1 (0 = floating root);
    body i>0 is attached to parent[i] through revolute joint i about a
    coordinate axis of the (unrotated) parent frame."""
    nb: int
    parent: List[int]
    axis: List[int]                 # 0/1/2 = x/y/z, -1 for the root
    joint_xyz: np.ndarray           # [nb,3] joint origin in the parent body frame
    mass: np.ndarray                # [nb]
    com: np.ndarray                 # [nb,3] body frame
    inertia: np.ndarray             # [nb,6] xx,yy,zz,xy,xz,yz about com, body axes
    body_dof: List[int]             # [nb] DoF index (simulator order) driven by joint i, -1 root
    dof_names: List[str]            # simulator DoF order (20 for widowGo1)
    dof_lower: np.ndarray
    dof_upper: np.ndarray
    dof_velocity: np.ndarray
    dof_effort: np.ndarray
    dof_friction: np.ndarray
    dof_locked: List[bool]
    body_names: List[str]           # moving body names
    # rigid-body list as Isaac Gym would expose it after collapse_fixed_joints
    rb_names: List[str]
    rb_body: List[int]              # moving body each rigid body rides on
    rb_offset: np.ndarray           # [nrb,3] rigid-body frame origin in that moving body's frame
    rb_mass: np.ndarray             # [nrb] mass of each rigid body (its link + the links collapsed into it)
    # un-merged pieces needed for per-env mass randomisation (widowGo1.py:431-456)
    base_piece: dict
    gripper_piece: dict

    @property
    def num_dofs(self) -> int:
        return len(self.dof_names)

    @property
    def num_rigid_bodies(self) -> int:
        return len(self.rb_names)

    def to_json(self) -> str:
        d = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                d[k] = v.tolist()
            elif isinstance(v, dict):
                d[k] = {kk: (vv.tolist() if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
            else:
                d[k] = v
        return json.dumps(d, indent=1)

    @staticmethod
    def from_json(s: str) -> "RobotModel":
        d = json.loads(s)
        arr = ["joint_xyz", "mass", "com", "inertia", "dof_lower", "dof_upper", "dof_velocity",
               "dof_effort", "dof_friction", "rb_offset", "rb_mass"]
        for k in arr:
            d[k] = np.array(d[k], dtype=np.float64)
        for piece in ("base_piece", "gripper_piece"):
            d[piece] = {kk: (np.array(vv, dtype=np.float64) if isinstance(vv, list) else vv)
                        for kk, vv in d[piece].items()}
        return RobotModel(**d)


def _sym6(I: np.ndarray) -> np.ndarray:
    return np.array([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])


def build_model(urdf_path: str, root_link: str = "base", lock_friction_above: float = 100.0,
                randomized_base_link: str = "base", gripper_link: str = "wx250s/ee_gripper_link") -> RobotModel:
    links, joints = parse_urdf(urdf_path)
    children: Dict[str, List[_Joint]] = {}
    for j in joints:
        children.setdefault(j.parent, []).append(j)
    for k in children:
        children[k].sort(key=lambda jj: jj.child)   # importer's alphabetical child order

    def is_locked(j: _Joint) -> bool:
        return j.jtype == "fixed" or (j.jtype == "prismatic" and j.friction >= lock_friction_above)

    parent: List[int] = []
    axis: List[int] = []
    joint_xyz: List[np.ndarray] = []
    body_names: List[str] = []
    comp: List[tuple] = []           # (m, c, I) per moving body
    body_dof: List[int] = []
    dof_names: List[str] = []
    dof_meta: List[_Joint] = []
    dof_locked: List[bool] = []
    rb_names: List[str] = []
    rb_body: List[int] = []
    rb_offset: List[np.ndarray] = []
    rb_mass: List[float] = []
    pieces: Dict[str, dict] = {}

    def add_piece(tag, body, lk_mass, lk_com, lk_I):
        pieces[tag] = dict(body=body, mass=lk_mass, com=lk_com.copy(), inertia=_sym6(lk_I))

    def visit(link_name: str, body: int, offset: np.ndarray, rb_index: int):
        """link rides on moving body `body` at translation `offset` (all URDF joint rpy are zero)."""
        lk = links[link_name]
        m, c, I = comp[body]
        comp[body] = _merge(m, c, I, lk.mass, lk.com + offset, lk.inertia)
        while len(rb_mass) < len(rb_names):
            rb_mass.append(0.0)
        rb_mass[rb_index] += lk.mass
        # rigid-body bookkeeping: a collapsed link adds to rigid body rb_index, which for
        # the two randomised links we keep as separate pieces
        for j in children.get(link_name, []):
            assert np.allclose(j.rpy, 0.0), f"joint {j.name}: non-zero rpy not supported by the kernels"
            if j.jtype == "fixed":
                new_rb = rb_index
                if j.dont_collapse:
                    rb_names.append(j.child)
                    rb_body.append(body)
                    rb_offset.append(offset + j.xyz)
                    new_rb = len(rb_names) - 1
                visit(j.child, body, offset + j.xyz, new_rb)
            elif is_locked(j):
                # prismatic finger: a DoF in the simulator's tensors, rigid in the dynamics
                dof_names.append(j.name)
                dof_meta.append(j)
                dof_locked.append(True)
                rb_names.append(j.child)
                rb_body.append(body)
                rb_offset.append(offset + j.xyz)
                visit(j.child, body, offset + j.xyz, len(rb_names) - 1)
            else:
                assert j.jtype == "revolute", f"joint {j.name}: type {j.jtype} unsupported"
                ax = int(np.argmax(np.abs(j.axis)))
                assert np.allclose(np.abs(j.axis), np.eye(3)[ax]) and j.axis[ax] > 0, \
                    f"joint {j.name}: axis must be +x/+y/+z"
                nb_new = len(parent)
                parent.append(body)
                axis.append(ax)
                joint_xyz.append(offset + j.xyz)
                body_names.append(j.child)
                comp.append((0.0, np.zeros(3), np.zeros((3, 3))))
                body_dof.append(len(dof_names))
                dof_names.append(j.name)
                dof_meta.append(j)
                dof_locked.append(False)
                rb_names.append(j.child)
                rb_body.append(nb_new)
                rb_offset.append(np.zeros(3))
                visit(j.child, nb_new, np.zeros(3), len(rb_names) - 1)

    parent.append(-1)
    axis.append(-1)
    joint_xyz.append(np.zeros(3))
    body_names.append(root_link)
    comp.append((0.0, np.zeros(3), np.zeros((3, 3))))
    body_dof.append(-1)
    rb_names.append(root_link)
    rb_body.append(0)
    rb_offset.append(np.zeros(3))
    visit(root_link, 0, np.zeros(3), 0)

    nb = len(parent)
    mass = np.array([c[0] for c in comp])
    com = np.array([c[1] for c in comp])
    inertia = np.array([_sym6(c[2]) for c in comp])

    # Pieces for per-env mass randomisation. Rigid body 0 of the importer ("base") is the
    # root link plus everything collapsed into it (here wx250s/base_link through the
    # widow_mount fixed joint); the rest of moving body 0 is the `trunk` rigid body.
    def collapsed_piece(start_link, body_offset):
        m, c, I = 0.0, np.zeros(3), np.zeros((3, 3))
        stack = [(start_link, body_offset)]
        while stack:
            ln, off = stack.pop()
            lk = links[ln]
            m, c, I = _merge(m, c, I, lk.mass, lk.com + off, lk.inertia)
            for j in children.get(ln, []):
                if j.jtype == "fixed" and not j.dont_collapse:
                    stack.append((j.child, off + j.xyz))
        return m, c, I

    bm, bc, bI = collapsed_piece(randomized_base_link, np.zeros(3))
    rest = _unmerge(comp[0], (bm, bc, bI))
    base_piece = dict(body=0, mass=bm, com=bc, inertia=_sym6(bI),
                      rest_mass=rest[0], rest_com=rest[1], rest_inertia=_sym6(rest[2]))
    gi = rb_names.index(gripper_link)
    gb = rb_body[gi]
    gm, gc, gI = collapsed_piece(gripper_link, rb_offset[gi])
    rest = _unmerge(comp[gb], (gm, gc, gI))
    gripper_piece = dict(body=gb, mass=gm, com=gc, inertia=_sym6(gI),
                         rest_mass=rest[0], rest_com=rest[1], rest_inertia=_sym6(rest[2]))

    return RobotModel(
        nb=nb, parent=parent, axis=axis, joint_xyz=np.array(joint_xyz), mass=mass, com=com, inertia=inertia,
        body_dof=body_dof, dof_names=dof_names,
        dof_lower=np.array([j.lower for j in dof_meta]), dof_upper=np.array([j.upper for j in dof_meta]),
        dof_velocity=np.array([j.velocity for j in dof_meta]), dof_effort=np.array([j.effort for j in dof_meta]),
        dof_friction=np.array([j.friction for j in dof_meta]), dof_locked=dof_locked,
        body_names=body_names, rb_names=rb_names, rb_body=rb_body, rb_offset=np.array(rb_offset), rb_mass=np.array(rb_mass),
        base_piece=base_piece, gripper_piece=gripper_piece)


def _unmerge(total, piece):
    """Remove `piece` from composite `total`; both are (m, c, I-about-own-com)."""
    mt, ct, It = total
    mp, cp, Ip = piece
    mr = mt - mp
    if mr <= 1e-12:
        return 0.0, ct.copy(), np.zeros((3, 3))
    cr = (mt * ct - mp * cp) / mr

    def shift(I, mm, d):
        return I + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    # It = shift(Ip, mp, cp-ct) + shift(Ir, mr, cr-ct)
    Ir = It - shift(Ip, mp, cp - ct) - mr * (np.dot(cr - ct, cr - ct) * np.eye(3) - np.outer(cr - ct, cr - ct))
    return mr, cr, Ir


def merge_piece(rest_m, rest_c, rest_I6, m, c, I6):
    """Vectorised composite of `rest` with a (possibly per-env) piece.
    Inputs broadcast over a leading env dimension. Returns (mass, com, inertia6)."""
    rest_c = np.asarray(rest_c, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    m = np.asarray(m, dtype=np.float64)
    M = rest_m + m
    C = (rest_m * rest_c + m[..., None] * c) / M[..., None]

    def shift6(I6_, mm, d):
        dd = np.sum(d * d, axis=-1)
        out = np.empty(d.shape[:-1] + (6,))
        out[..., 0] = I6_[..., 0] + mm * (dd - d[..., 0] * d[..., 0])
        out[..., 1] = I6_[..., 1] + mm * (dd - d[..., 1] * d[..., 1])
        out[..., 2] = I6_[..., 2] + mm * (dd - d[..., 2] * d[..., 2])
        out[..., 3] = I6_[..., 3] - mm * d[..., 0] * d[..., 1]
        out[..., 4] = I6_[..., 4] - mm * d[..., 0] * d[..., 2]
        out[..., 5] = I6_[..., 5] - mm * d[..., 1] * d[..., 2]
        return out
    I = shift6(np.asarray(rest_I6, dtype=np.float64), rest_m, rest_c - C) + shift6(np.asarray(I6, dtype=np.float64), m, c - C)
    return M, C, I
