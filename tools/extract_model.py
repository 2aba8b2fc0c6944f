#!/usr/bin/env python3
"""Regenerate wbc_amd/assets/widowgo1_model.json from the robot's URDF.

The URDF lives in the reference tree (legged_gym/resources/robots/widowGo1/urdf/widowGo1.urdf)
and is not copied into this repo; the JSON holds only the derived flat arrays (masses, composite
inertias, joint origins, limits) that the kernels need, so the framework runs where the
reference tree is absent (the GPU box)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "deep-whole-body-control_amd"))
from wbc_amd.urdf_model import build_model  # noqa: E402

urdf = sys.argv[1] if len(sys.argv) > 1 else \
    "/root/reference/legged_gym/resources/robots/widowGo1/urdf/widowGo1.urdf"
out = os.path.join(HERE, "..", "deep-whole-body-control_amd", "wbc_amd", "assets", "widowgo1_model.json")
m = build_model(urdf)
with open(out, "w") as f:
    f.write(m.to_json())
print("wrote", out, "bodies", m.nb, "dofs", m.num_dofs, "rigid bodies", m.num_rigid_bodies, "mass", m.mass.sum())
