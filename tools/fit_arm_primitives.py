"""Fit this framework's collision primitives of the WidowX arm (capsules between joint centres with end spheres, contact spheres) to the
convex hulls of the STL meshes the URDF names as <collision> geometry (widowGo1.urdf:424-819; Isaac Gym / PhysX collide convex hulls of
them), and print how far each primitive under- and over-approximates its hull.

    python tools/fit_arm_primitives.py [--urdf PATH] [--apply]

Reads /root/reference (read-only); nothing of it is copied: the output is the table below and, with --apply, the radii written to
deep-whole-body-control_amd/wbc_amd/assets/arm_primitives.json (which abi.collision_set reads). Under-approximation = the largest
distance of a hull point OUTSIDE the primitive's surface; over-approximation = the largest distance of a primitive surface point
outside the hull. Limbs span two links joined by a roll joint about the limb's own axis, so the radial extent does not depend on
the roll angle; the fingers are taken at their locked default opening."""
import argparse, json, os, sys
import xml.etree.ElementTree as ET
import numpy as np
from scipy.spatial import ConvexHull

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
DEFAULT_URDF = "/root/reference/legged_gym/resources/robots/widowGo1/urdf/widowGo1.urdf"
UNDER_TARGET = 0.010          # the review's target: no hull point more than 10 mm outside its primitive


def rpy_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def read_stl(path):
    """vertices [n, 3, 3] of a binary STL"""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[80:84], dtype="<u4")[0])
    rec = np.frombuffer(raw[84:84 + 50 * n], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return rec["v"].astype(np.float64)


def floats(s, d=(0, 0, 0)):
    return np.array([float(x) for x in s.split()]) if s else np.array(d, dtype=np.float64)


def load(urdf):
    root = ET.parse(urdf).getroot()
    links, joints = {}, {}
    for le in root.findall("link"):
        ce = le.find("collision")
        mesh = None
        if ce is not None and ce.find("geometry/mesh") is not None:
            me, oe = ce.find("geometry/mesh"), ce.find("origin")
            mesh = dict(file=os.path.normpath(os.path.join(os.path.dirname(urdf), me.attrib["filename"])), scale=floats(me.attrib.get("scale"), (1, 1, 1)),
                        xyz=floats(oe.attrib.get("xyz") if oe is not None else None), rpy=floats(oe.attrib.get("rpy") if oe is not None else None))
        links[le.attrib["name"]] = mesh
    for je in root.findall("joint"):
        oe = je.find("origin")
        joints[je.find("child").attrib["link"]] = dict(name=je.attrib["name"], type=je.attrib["type"], parent=je.find("parent").attrib["link"],
                                                       xyz=floats(oe.attrib.get("xyz") if oe is not None else None), rpy=floats(oe.attrib.get("rpy") if oe is not None else None),
                                                       axis=floats(je.find("axis").attrib["xyz"]) if je.find("axis") is not None else np.zeros(3))
    return links, joints


def link_points(links, name):
    m = links.get(name)
    if not m:
        return np.zeros((0, 3))
    v = read_stl(m["file"]).reshape(-1, 3) * m["scale"]
    return v @ rpy_mat(m["rpy"]).T + m["xyz"]


def subtree_points(links, joints, root_link, finger_q):
    """Mesh vertices of `root_link` and of everything fixed (or finger-locked) to it, in root_link's frame (all joint rpy of the arm are 0)."""
    pts = [link_points(links, root_link)]
    for child, j in joints.items():
        if j["parent"] != root_link:
            continue
        if j["type"] == "fixed" or j["type"] == "prismatic":
            off = j["xyz"] + (j["axis"] * finger_q.get(j["name"], 0.0) if j["type"] == "prismatic" else 0.0)
            assert np.allclose(j["rpy"], 0)
            pts.append(subtree_points(links, joints, child, finger_q) + off)
    return np.concatenate(pts, 0)


def hull_samples(pts, per_facet=6, seed=0):
    h = ConvexHull(pts)
    rng = np.random.default_rng(seed)
    tri = pts[h.simplices]                                   # [f, 3, 3]
    w = rng.dirichlet(np.ones(3), size=(len(tri), per_facet))
    return np.concatenate([pts[h.vertices], np.einsum("fkj,fjd->fkd", w, tri).reshape(-1, 3)], 0), h


def seg_dist(p, a, b):
    d = b - a
    t = np.clip(((p - a) @ d) / max(d @ d, 1e-12), 0, 1)
    return np.linalg.norm(p - (a + t[:, None] * d), axis=1)


def capsule_surface(a, b, r, n=4000, seed=1):
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    t = rng.uniform(0, 1, n)
    d = b - a
    L = np.linalg.norm(d)
    ax = d / max(L, 1e-12)
    # points on the cylinder and on the two caps
    side = u - np.outer(u @ ax, ax); side /= np.linalg.norm(side, axis=1, keepdims=True)
    cyl = a + np.outer(t, d) + r * side
    cap = np.where((u @ ax)[:, None] > 0, b + r * u, a + r * u)
    return np.concatenate([cyl, cap], 0)


def sphere_surface(c, r, n=1500, seed=2):
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    return c + r * u


def limb_metrics(smp, h, a, b, rs, r0, r1):
    """(under, over) of the union capsule(a, b, rs) + sphere(a, r0) + sphere(b, r1) against the hull `h` sampled by `smp`"""
    inside = seg_dist(smp, a, b) - rs
    if r0 > 0: inside = np.minimum(inside, np.linalg.norm(smp - a, axis=1) - r0)
    if r1 > 0: inside = np.minimum(inside, np.linalg.norm(smp - b, axis=1) - r1)
    under = max(inside.max(), 0.0)
    surf = [capsule_surface(a, b, rs, n=1500)]
    if r0 > rs: surf.append(sphere_surface(a, r0))
    if r1 > rs: surf.append(sphere_surface(b, r1))
    q = np.concatenate(surf, 0)
    # keep the points on the union's outer surface
    dseg, da, db = seg_dist(q, a, b), np.linalg.norm(q - a, axis=1), np.linalg.norm(q - b, axis=1)
    outer = (dseg >= rs - 1e-9) & ((r0 <= 0) | (da >= r0 - 1e-9)) & ((r1 <= 0) | (db >= r1 - 1e-9))
    over = max(outside_hull(h, q[outer]).max(), 0.0)
    return under, over


def fit_limb(pts, a, b, under_cap=None):
    """shaft radius and end-sphere radii (0 = none) that minimise max(under, over) -- or, with under_cap, the over-approximation subject
    to under <= under_cap -- over a 2.5 mm grid"""
    smp, h = hull_samples(pts)
    grid = np.arange(0.015, 0.0801, 0.0025)
    best = None
    for rs in grid:
        for r0 in [0.0] + [x for x in grid if x > rs]:
            for r1 in [0.0] + [x for x in grid if x > rs]:
                u, o = limb_metrics(smp, h, a, b, rs, r0, r1)
                if under_cap is not None and u > under_cap:
                    continue
                key = o if under_cap is not None else max(u, o)
                if best is None or key < best[0] - 1e-6:
                    best = (key, rs, r0, r1, u, o)
    return best[1:]


def outside_hull(h, q):
    """signed distance bound: max over facets of (n.q + d); > 0 = outside"""
    return (q @ h.equations[:, :3].T + h.equations[:, 3]).max(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--urdf", default=DEFAULT_URDF)
    ap.add_argument("--apply", action="store_true")
    args = ap.parse_args()
    from wbc_amd import abi
    links, joints = load(args.urdf)
    fq = {"widow_left_finger": 0.037, "widow_right_finger": -0.037}     # widowGo1_config.py default_joint_angles (locked prismatic joints)
    J = lambda child: joints[child]["xyz"]                               # noqa: E731
    out, rows, limb_geoms = {}, [], []

    def report(name, pts_frames, a, b, r_now, spheres=()):
        """pts in the frame of the limb's first link; capsule a -> b with radius r_now (+ end spheres (centre, radius))"""
        smp, h = hull_samples(pts_frames)
        d = seg_dist(smp, a, b)
        # a hull point is inside the primitive if it is inside the capsule or inside one of the end spheres
        out_cap = d - r_now
        for c, rs in spheres:
            out_cap = np.minimum(out_cap, np.linalg.norm(smp - c, axis=1) - rs)
        under = max(out_cap.max(), 0.0)
        surf = capsule_surface(a, b, r_now)
        over = max(outside_hull(h, surf).max(), 0.0)
        r_fit = max(np.ceil((d.max() - UNDER_TARGET) * 1000) / 1000, 0.005)
        over_fit = max(outside_hull(h, capsule_surface(a, b, r_fit)).max(), 0.0)
        rows.append((name, d.max(), r_now, under, over, r_fit, max(d.max() - r_fit, 0.0), over_fit))
        return r_fit

    # upper arm: shoulder joint .. elbow joint (one link, L-shaped)
    p = subtree_points(links, joints, "wx250s/upper_arm_link", fq)
    limb_geoms.append(("upper_arm", p, np.zeros(3), J("wx250s/upper_forearm_link")))
    out["upper_arm_radius"] = report("upper arm (shoulder joint .. elbow)", p, np.zeros(3), J("wx250s/upper_forearm_link"), abi.ARM_LIMB_FIT["upper_arm"][0])
    # forearm: elbow .. wrist = upper_forearm_link + lower_forearm_link (roll about the forearm axis)
    p = np.concatenate([subtree_points(links, joints, "wx250s/upper_forearm_link", fq), subtree_points(links, joints, "wx250s/lower_forearm_link", fq) + J("wx250s/lower_forearm_link")], 0)
    limb_geoms.append(("forearm", p, np.zeros(3), J("wx250s/lower_forearm_link") + J("wx250s/wrist_link")))
    out["forearm_radius"] = report("forearm (elbow .. wrist)", p, np.zeros(3), J("wx250s/lower_forearm_link") + J("wx250s/wrist_link"), abi.ARM_LIMB_FIT["forearm"][0])
    # hand: wrist .. gripper tip = wrist_link + gripper_link and what is fixed to it (prop, bar, fingers)
    tip = J("wx250s/gripper_link") + J("wx250s/ee_arm_link") + J("wx250s/gripper_bar_link") + J("wx250s/fingers_link") + J("wx250s/ee_gripper_link")
    p = np.concatenate([subtree_points(links, joints, "wx250s/wrist_link", fq), subtree_points(links, joints, "wx250s/gripper_link", fq) + J("wx250s/gripper_link")], 0)
    limb_geoms.append(("hand", p, np.zeros(3), tip))
    out["hand_radius"] = report("hand (wrist .. gripper tip)", p, np.zeros(3), tip, abi.ARM_LIMB_FIT["hand"][0])
    out["hand_len"] = float(np.linalg.norm(tip))
    # shoulder link (waist joint .. shoulder joint): the sphere at the shoulder joint stands for it
    p = subtree_points(links, joints, "wx250s/shoulder_link", fq)
    out["shoulder_radius"] = report("shoulder link (sphere at the shoulder joint)", p, J("wx250s/upper_arm_link"), J("wx250s/upper_arm_link"), abi.ELBOW_RADIUS)
    # the arm's base: rigid on the trunk after collapse_fixed_joints
    p = subtree_points(links, joints, "wx250s/base_link", fq) + J("wx250s/base_link")
    lo, hi = p.min(0), p.max(0)
    out["base_box_centre"], out["base_box_half"] = ((lo + hi) / 2).tolist(), ((hi - lo) / 2).tolist()
    # the limbs of the self-collision set: capsule + end spheres, fitted (a) to the smallest max(under, over) and (b) to under <= 10 mm
    fits = {}
    for name, p, a, b in limb_geoms:
        bal, capd = fit_limb(p, a, b), fit_limb(p, a, b, under_cap=UNDER_TARGET)
        fits[name] = dict(balanced=dict(radius=bal[0], cap0=bal[1], cap1=bal[2], under=bal[3], over=bal[4]),
                          under10=dict(radius=capd[0], cap0=capd[1], cap1=capd[2], under=capd[3], over=capd[4]))
        print(f"limb {name:10s} balanced: shaft {bal[0]:.4f} end spheres {bal[1]:.4f} / {bal[2]:.4f} -> under {bal[3] * 1e3:.1f} mm, over {bal[4] * 1e3:.1f} mm | "
              f"under <= 10 mm: shaft {capd[0]:.4f} end spheres {capd[1]:.4f} / {capd[2]:.4f} -> under {capd[3] * 1e3:.1f} mm, over {capd[4] * 1e3:.1f} mm")
    out["limbs"] = fits
    print(f"{'primitive':46s} {'hull reach':>10s} {'r now':>7s} {'under':>7s} {'over':>7s} | {'r fit':>7s} {'under':>7s} {'over':>7s}   (metres; fit = within {UNDER_TARGET * 1e3:.0f} mm under)")
    for name, reach, r_now, under, over, r_fit, ufit, ofit in rows:
        print(f"{name:46s} {reach:10.4f} {r_now:7.3f} {under:7.4f} {over:7.4f} | {r_fit:7.3f} {ufit:7.4f} {ofit:7.4f}")
    print("arm base (wx250s/base_link on the trunk), box in the base frame: centre", np.round(out["base_box_centre"], 4), "half extents", np.round(out["base_box_half"], 4))
    if args.apply:
        path = os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "assets", "arm_primitives.json")
        json.dump(json.loads(json.dumps(out, default=float)), open(path, "w"), indent=1)
        print("wrote", path)
        # hull samples + facets as a test fixture (data: points on the hulls of the reference's meshes, in limb frames)
        fx = {}
        for name, p, a, b in limb_geoms:
            smp, h = hull_samples(p)
            fx[name + "_samples"], fx[name + "_facets"], fx[name + "_a"], fx[name + "_b"] = smp.astype(np.float32), h.equations.astype(np.float32), a, b
        gpath = os.path.join(ROOT, "tests", "golden", "arm_hull_samples.npz")
        np.savez_compressed(gpath, **fx)
        print("wrote", gpath)


if __name__ == "__main__":
    main()


def profile(name, pts, a, b, nb=10):
    """radial reach of the hull along the limb axis (bins of the axial coordinate, beyond the ends included)"""
    smp, _ = hull_samples(pts)
    d = b - a; L = np.linalg.norm(d); ax = d / L
    t = (smp - a) @ ax
    rad = np.linalg.norm(smp - a - np.outer(t, ax), axis=1)
    edges = np.linspace(min(t.min(), 0) - 1e-9, max(t.max(), L) + 1e-9, nb + 1)
    print(name, f"axis length {L:.3f}; axial span of the hull {t.min():.3f} .. {t.max():.3f}")
    for i in range(nb):
        sel = (t >= edges[i]) & (t < edges[i + 1])
        if sel.any():
            print(f"   t in [{edges[i]:+.3f}, {edges[i + 1]:+.3f}): max radial {rad[sel].max():.4f}")
