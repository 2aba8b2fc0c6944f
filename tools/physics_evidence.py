#!/usr/bin/env python3
"""Print the closed-form physics evidence table (tests/physics_cases.py run on the fp64 oracle): what DESIGN.md section 3 quotes
and tests/test_oracle_contact_physics.py asserts. CPU only:  python tools/physics_evidence.py > profiles/r04_physics_evidence.txt"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("tests", "oracle", "deep-whole-body-control_amd", "."):
    sys.path.insert(0, os.path.join(HERE, "..", p))
import numpy as np

import oracle
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg

oracle.build()
import physics_cases as pc  # noqa: E402
import test_oracle_contact_physics as T  # noqa: E402

m = abi.load_default_model()
cfg = WidowGo1RoughCfg()
robot = dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=abi.fill_task_cfg(cfg, m))

print("(a) incline: down-slope acceleration [m/s^2], measured vs g (sin - mu cos) (0 when tan < mu)")
for tf, tt in [(1.0, 0.2), (1.0, 0.7), (0.2, 0.5), (0.2, 0.7), (0.2, 0.9), (-0.6, 0.1), (-0.6, 0.3), (-0.6, 0.6), (-1.0, 0.2), (-1.0, 0.7)]:
    r = pc.box_on_incline(robot, tt, tf)
    print(f"  box   mu {r['mu']:.2f} tan {tt:.2f}: {r['acc']:8.4f} vs {r['expect']:8.4f}   v_end {r['v_end']:.4f} spin {r['spin']:.4f}")
for me, tf, tt in [(-0.5, 1.0, 0.1), (-0.5, 1.0, 0.2), (-0.5, 1.0, 0.4), (-0.5, 1.0, 0.6), (0.0, 1.0, 0.4), (0.0, 1.0, 0.6), (1.0, 1.0, 0.6),
                   (3.0, 1.0, 0.6), (0.6, 0.0, 0.5), (-0.5, 0.0, 0.1), (-0.5, 0.0, 0.4)]:
    r = pc.robot_on_incline(robot, tt, me, tf, t_settle=1.2, t_measure=0.4)
    print(f"  robot mu_env {me:4.1f} terrain {tf:.1f} -> mu {r['mu']:.2f} tan {tt:.2f}: {r['acc']:8.4f} vs {r['expect']:8.4f}   v_end {r['v_end']:.4f}")
print("(b) drops")
for k, v in pc.robot_drop(robot).items():
    print(f"  robot {k}: {v:.5g}")
for k, v in pc.box_drop(robot).items():
    print(f"  box   {k}: {v:.5g}")
print("(c) single-joint PD step (0.2 rad; everything else locked): deviation from the exact response of WG:1281 in units of the step")
print("  joint               Kp   Kd   inertia  armature  I/(I+A)  max_dev(shipped)  max_dev(explicit)  rise sim/exact [s]  dt*Kd/I")
for j in (0, 1, 2, 12, 13, 14, 15, 16, 17):
    a, b = pc.joint_pd_step(robot, j, armature=True), pc.joint_pd_step(robot, j, armature=False)
    print(f"  {m.dof_names[j]:20s}{a['kp']:4.0f} {a['kd']:4.1f}  {a['inertia']:.5f}  {a['armature']:.5f}   {a['gain_ratio']:.3f}    {a['max_dev']:.4f}"
          f"            {b['max_dev'] if b['stable'] else float('nan'):.4f}             {a['rise_sim']:.3f} / {a['rise_exact']:.3f}       {a['explicit_margin']:.3f}")
print("(d) solver convergence: one substep from 640 staged states, contact_iters vs 1024 sweeps (robot rows of net_contact_force, post-step velocities)")
st = pc.contact_states(robot, 128)
res = pc.solver_convergence(robot, st, (1, 2, 3, 4, 8, 64, 1024))
for it in (1, 2, 3, 4, 8, 64):
    e = T._class_errors(st, res, it)
    for name, r in e.items():
        print(f"  iters {it:3d} {name:7s} n={r['n']:3d} (unsolvable {r['skipped']}): force rel. error median {r['f_median']:.4f} p90 {r['f_p90']:.4f};"
              f" velocity error median {r['dv_median']:.5f} p90 {r['dv_p90']:.5f}")
print("(e) passive swing, 2 s")
for s in (0, 1, 2):
    print("  ", {k: round(v, 5) for k, v in pc.free_flight_energy(robot, seed=s).items()})
print("(f) foot against the box")
print("  ", pc.robot_kicks_box(robot))
