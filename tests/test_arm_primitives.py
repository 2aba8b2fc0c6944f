"""The arm's collision primitives against the convex hulls of the meshes the URDF names (widowGo1.urdf:504-819; PhysX collides those
hulls): tests/golden/arm_hull_samples.npz holds points on the hulls and the hulls' facet planes, in limb frames
(tools/fit_arm_primitives.py --apply, from the reference's STL files); the capsule radii abi.collision_set uses
(assets/arm_primitives.json -> abi.ARM_LIMB_FIT) must be the fitted ones and deviate from the hulls by what INTEGRATION.md section 4
states. Brute force: point-to-segment distances and plane tests, no code shared with the fitting tool."""
import json
import os

import numpy as np

from wbc_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = np.load(os.path.join(ROOT, "tests", "golden", "arm_hull_samples.npz"))
STATED_MM = {"upper_arm": 29.0, "forearm": 14.0, "hand": 24.5}           # max(under, over), INTEGRATION.md section 4


def _seg_dist(p, a, b):
    d = b - a
    t = np.clip(((p - a) @ d) / (d @ d), 0.0, 1.0)
    return np.linalg.norm(p - (a + t[:, None] * d), axis=1)


def _deviation(name, radius):
    smp, fac, a, b = (FX[name + k].astype(np.float64) for k in ("_samples", "_facets", "_a", "_b"))
    under = max((_seg_dist(smp, a, b) - radius).max(), 0.0)            # hull points outside the capsule
    # capsule surface points outside the hull: a dense deterministic sample of the surface
    ax = (b - a) / np.linalg.norm(b - a)
    e1 = np.cross(ax, [0.0, 1.0, 0.0] if abs(ax[0]) > 0.9 else [1.0, 0.0, 0.0]); e1 /= np.linalg.norm(e1); e2 = np.cross(ax, e1)
    th = np.linspace(0, 2 * np.pi, 48, endpoint=False)
    ring = np.cos(th)[:, None] * e1 + np.sin(th)[:, None] * e2
    pts = [a + t * (b - a) + radius * ring for t in np.linspace(0, 1, 40)]
    for ph in np.linspace(0, np.pi / 2, 12):                              # the two caps
        pts.append(b + radius * (np.cos(ph) * ring + np.sin(ph) * ax))
        pts.append(a + radius * (np.cos(ph) * ring - np.sin(ph) * ax))
    q = np.concatenate(pts, 0)
    over = max((q @ fac[:, :3].T + fac[:, 3]).max(1).max(), 0.0)
    return under, over


def test_arm_limb_radii_are_the_fitted_ones_and_deviate_as_stated():
    fit = json.load(open(os.path.join(ROOT, "deep-whole-body-control_amd", "wbc_amd", "assets", "arm_primitives.json")))["limbs"]
    m = abi.load_default_model()
    _, limbs, _ = abi.collision_set(m)
    by_name = {l["name"]: l for l in limbs}
    for name in ("upper_arm", "forearm", "hand"):
        r = by_name[name]["radius"]
        assert abs(r - fit[name]["balanced"]["radius"]) < 1e-4 and by_name[name]["cap0"] == 0.0 and by_name[name]["cap1"] == 0.0
        under, over = _deviation(name, r)
        print(f"{name}: radius {r * 1e3:.1f} mm, hull sticks out {under * 1e3:.1f} mm, capsule sticks out {over * 1e3:.1f} mm")
        assert abs(under - fit[name]["balanced"]["under"]) < 1.5e-3 and abs(over - fit[name]["balanced"]["over"]) < 2.5e-3   # the tool's own (random) sampling
        assert max(under, over) * 1e3 <= STATED_MM[name]
        # the hand-typed radii of rounds 3-5 (25 / 25 / 20 mm) were worse on the two limbs that changed
        old = {"upper_arm": 0.025, "forearm": 0.025, "hand": 0.020}[name]
        assert max(_deviation(name, old)) >= max(under, over) - 1e-4
    # every candidate limb pair stays inside the broad phase's radius bound
    assert max(l["radius"] for l in limbs) + 0.02 <= abi.LIMB_RSUM_MAX + 1e-9
