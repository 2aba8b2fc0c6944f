#!/usr/bin/env python3
"""Generate tests/golden/terrain_reference.npz from the REFERENCE's own terrain code
(/root/reference/legged_gym/legged_gym/utils/terrain.py and envs/base/legged_robot.py, imported read-only
under the fake isaacgym of tools/ref_harness/igstub.py):

  * Terrain_Perlin (terrain.py:40-99) on a small grid, np.random.rand patched to a seeded Generator so that this
    framework's TerrainPerlin can be fed the same uniforms;
  * Terrain (terrain.py:101-227), the base class's sub-terrain grid with curriculum, on a reduced LeggedRobotCfg.terrain
    (isaacgym.terrain_utils = this framework's restatement wbc_amd/terrain_utils.py, np.random seeded);
  * LeggedRobot._init_height_points / _get_heights (legged_robot.py:777-829) and _update_terrain_curriculum
    (legged_robot.py:421-441) called UNBOUND on a hand-built namespace with seeded tensors.

Run in the build container only: python tools/make_golden_terrain.py"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
import numpy as np
import torch

import harness

harness._import_reference()
from legged_gym.utils.terrain import Terrain, Terrain_Perlin          # noqa: E402  (the reference's)
from legged_gym.envs.base.legged_robot import LeggedRobot            # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
out = {}

# ---- Terrain_Perlin -------------------------------------------------------------------------------------------------
pcfg = types.SimpleNamespace(horizontal_scale=0.025, vertical_scale=1e-5, tot_cols=240, tot_rows=160, zScale=0.15, slope_treshold=1e8)
rng = np.random.default_rng(5)
orig = np.random.rand
np.random.rand = lambda *shape: rng.random(shape)
try:
    with np.errstate(invalid="ignore"):
        ref = Terrain_Perlin(pcfg)
finally:
    np.random.rand = orig
out["perlin_heightsamples"] = ref.heightsamples
out["perlin_cfg"] = np.array([pcfg.horizontal_scale, pcfg.vertical_scale, pcfg.tot_cols, pcfg.tot_rows, pcfg.zScale, 5])

# ---- Terrain (sub-terrain grid) -------------------------------------------------------------------------------------
from legged_gym.envs.base.legged_robot_config import LeggedRobotCfg   # noqa: E402
tcfg = LeggedRobotCfg().terrain
tcfg.num_rows, tcfg.num_cols, tcfg.border_size = 4, 10, 5           # LRC:43-66 otherwise (8 m tiles, proportions, scales)
np.random.seed(123)
ter = Terrain(tcfg, 64)
out["grid_height_field"] = ter.height_field_raw
out["grid_env_origins"] = ter.env_origins
out["grid_cfg"] = np.array([tcfg.num_rows, tcfg.num_cols, tcfg.border_size, 123])
tcfg2 = LeggedRobotCfg().terrain
tcfg2.num_rows, tcfg2.num_cols, tcfg2.border_size, tcfg2.curriculum = 3, 5, 2, False
np.random.seed(321)
ter2 = Terrain(tcfg2, 64)
out["grid_random_height_field"] = ter2.height_field_raw
out["grid_random_env_origins"] = ter2.env_origins
out["grid_random_cfg"] = np.array([tcfg2.num_rows, tcfg2.num_cols, tcfg2.border_size, 321])

# ---- _get_heights / _init_height_points ------------------------------------------------------------------------------
n = 48
g = torch.Generator().manual_seed(7)
hcfg = types.SimpleNamespace(terrain=types.SimpleNamespace(
    mesh_type="trimesh", measured_points_x=LeggedRobotCfg.terrain.measured_points_x, measured_points_y=LeggedRobotCfg.terrain.measured_points_y,
    border_size=float(tcfg.border_size), horizontal_scale=tcfg.horizontal_scale, vertical_scale=tcfg.vertical_scale))
ns = types.SimpleNamespace(cfg=hcfg, num_envs=n, device="cpu", terrain=types.SimpleNamespace(cfg=hcfg.terrain))
ns.height_points = LeggedRobot._init_height_points(ns)
ns.height_samples = torch.tensor(ter.heightsamples).view(ter.tot_rows, ter.tot_cols)
quat = torch.randn(n, 4, generator=g)
quat = quat / quat.norm(dim=1, keepdim=True)
root = torch.zeros(n, 13)
root[:, 0] = torch.rand(n, generator=g) * 36 - 2          # some off the grid: indices clip
root[:, 1] = torch.rand(n, generator=g) * 84 - 2
root[:8, :2] = torch.tensor([[-100.0, 3.0], [3.0, -100.0], [500.0, 20.0], [20.0, 500.0], [0.05, 0.05], [8.0, 8.0], [-5.0, -5.0], [16.0, 24.0]])
root[:, 2] = 0.4
root[:, 3:7] = quat
ns.base_quat, ns.root_states = root[:, 3:7], root
heights = LeggedRobot._get_heights(ns)
out["heights_points"] = ns.height_points.numpy()
out["heights_root"] = root.numpy()
out["heights_out"] = heights.numpy()
out["heights_num_points"] = np.int64(ns.num_height_points)

# ---- _update_terrain_curriculum --------------------------------------------------------------------------------------
n = 256
cns = types.SimpleNamespace(init_done=True, device="cpu", max_episode_length_s=10.0, max_terrain_level=tcfg.num_rows,
                            terrain=types.SimpleNamespace(env_length=8.0))
cns.terrain_origins = torch.from_numpy(ter.env_origins).float()
cns.terrain_levels = torch.randint(0, tcfg.num_rows, (n,), generator=g)
cns.terrain_types = torch.randint(0, tcfg.num_cols, (n,), generator=g)
cns.env_origins = cns.terrain_origins[cns.terrain_levels, cns.terrain_types].clone()
cns.root_states = torch.zeros(n, 13)
cns.root_states[:, :2] = cns.env_origins[:, :2] + (torch.rand(n, 2, generator=g) - 0.5) * 12
cns.commands = torch.zeros(n, 3)
cns.commands[:, 0] = torch.rand(n, generator=g) * 0.9
env_ids = torch.nonzero(torch.rand(n, generator=g) < 0.5).flatten()
out["cur_levels_before"] = cns.terrain_levels.numpy().copy()
out["cur_types"] = cns.terrain_types.numpy()
out["cur_root_xy"] = cns.root_states[:, :2].numpy().copy()
out["cur_origins_before"] = cns.env_origins.numpy().copy()
out["cur_commands"] = cns.commands.numpy().copy()
out["cur_env_ids"] = env_ids.numpy()
torch.manual_seed(99)                                       # the randint_like of "solved the last level"
LeggedRobot._update_terrain_curriculum(cns, env_ids)
out["cur_levels_after"] = cns.terrain_levels.numpy().copy()
out["cur_origins_after"] = cns.env_origins.numpy().copy()
out["cur_terrain_origins"] = cns.terrain_origins.numpy()

path = os.path.join(GOLD, "terrain_reference.npz")
np.savez_compressed(path, **out)
print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB", {k: np.asarray(v).shape for k, v in out.items()})
