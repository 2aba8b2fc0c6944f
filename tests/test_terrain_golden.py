"""Terrain code against the REFERENCE's own (tests/golden/terrain_reference.npz, written by tools/make_golden_terrain.py from
/root/reference/legged_gym/legged_gym/utils/terrain.py and envs/base/legged_robot.py):
Terrain_Perlin, the base class's sub-terrain grid `Terrain`, LeggedRobot._get_heights / _init_height_points (integer
cell indexing: bit-exact) and LeggedRobot._update_terrain_curriculum."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import terrain_oracle as to  # noqa: E402
from wbc_amd.terrain import Terrain, TerrainPerlin  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "terrain_reference.npz")))


def test_perlin_matches_reference():
    """utils/terrain.py:40-99 fed the same uniforms: the identical int16 grid, every sample (the float64 field is evaluated in
    the reference's operation order)."""
    hs, vs, cols, rows, zs, seed = G["perlin_cfg"]
    cfg = types.SimpleNamespace(horizontal_scale=float(hs), vertical_scale=float(vs), tot_cols=int(cols), tot_rows=int(rows), zScale=float(zs),
                                transform_x=0.0, transform_y=0.0, transform_z=0.0)
    ours = TerrainPerlin(cfg, seed=int(seed)).heightsamples
    ref = G["perlin_heightsamples"]
    assert ours.shape == ref.shape and ours.dtype == ref.dtype == np.int16
    np.testing.assert_array_equal(ours, ref)
    assert np.abs(ref[int(cols) // 2 - 100:]).max() == 0 and ref.max() > 10000       # quirk Q3's flat part, real relief elsewhere


def _grid_cfg(num_rows, num_cols, border, curriculum):
    from wbc_amd.config import LeggedRobotCfg
    t = LeggedRobotCfg().terrain
    t.num_rows, t.num_cols, t.border_size, t.curriculum = int(num_rows), int(num_cols), int(border), curriculum
    return t


@pytest.mark.parametrize("key,curriculum", [("grid", True), ("grid_random", False)])
def test_subterrain_grid_matches_reference(key, curriculum):
    """terrain.py:101-227 with LeggedRobotCfg.terrain (LRC:43-66: 8 m tiles, proportions [.1,.1,.35,.25,.2], 0.1 m / 5 mm
    scales): the int16 height grid and the platform origins, same np.random seed, bit for bit."""
    rows, cols, border, seed = G[key + "_cfg"]
    np.random.seed(int(seed))
    t = Terrain(_grid_cfg(rows, cols, border, curriculum), 64)
    np.testing.assert_array_equal(t.height_field_raw, G[key + "_height_field"])
    np.testing.assert_allclose(t.env_origins, G[key + "_env_origins"], rtol=0, atol=1e-12)
    assert t.vertices.shape == (t.tot_rows * t.tot_cols, 3) and t.triangles.shape == (2 * (t.tot_rows - 1) * (t.tot_cols - 1), 3)
    kinds = {int(np.ptp(t.height_field_raw[t.border + 80 * i: t.border + 80 * (i + 1), t.border + 80 * j: t.border + 80 * (j + 1)]) > 0)
             for i in range(int(rows)) for j in range(int(cols))}
    assert 1 in kinds


def test_get_heights_oracle_matches_reference_bit_exact():
    """LR:777-829 (called unbound on seeded tensors in the build container) vs oracle/terrain_oracle.py: every index, every bit."""
    from wbc_amd.config import LeggedRobotCfg
    pts = to.init_height_points(LeggedRobotCfg.terrain.measured_points_x, LeggedRobotCfg.terrain.measured_points_y, G["heights_root"].shape[0])
    np.testing.assert_array_equal(pts, G["heights_points"])
    assert pts.shape[1] == int(G["heights_num_points"]) == 187
    root = G["heights_root"]
    rows, cols, border, _ = G["grid_cfg"]
    got = to.get_heights(root[:, 3:7], root[:, :3], pts, G["grid_height_field"], float(border), 0.1, 0.005)
    np.testing.assert_array_equal(got, G["heights_out"])
    assert len(np.unique(got)) > 50


def test_update_terrain_curriculum_matches_reference():
    """LR:421-441 vs this package's WidowGo1._update_terrain_curriculum (host rule over the travel / command norm the fused
    step records) and vs the oracle's restatement."""
    from wbc_amd.envs import WidowGo1
    ids = torch.from_numpy(G["cur_env_ids"])
    n = G["cur_levels_before"].shape[0]
    travel = np.zeros((n, 2), np.float32)
    d = G["cur_root_xy"] - G["cur_origins_before"][:, :2]
    travel[:, 0] = np.sqrt((d * d).sum(1))
    travel[:, 1] = np.sqrt((G["cur_commands"][:, :2] ** 2).sum(1))
    ns = types.SimpleNamespace(init_done=True, max_episode_length_s=10.0, max_terrain_level=int(G["grid_cfg"][0]),
                               terrain=types.SimpleNamespace(env_length=8.0), _reset_travel=torch.from_numpy(travel),
                               terrain_levels=torch.from_numpy(G["cur_levels_before"].copy()), terrain_types=torch.from_numpy(G["cur_types"]),
                               terrain_origins=torch.from_numpy(G["cur_terrain_origins"]), env_origins=torch.from_numpy(G["cur_origins_before"].copy()))
    torch.manual_seed(99)
    WidowGo1._update_terrain_curriculum(ns, ids)
    np.testing.assert_array_equal(ns.terrain_levels.numpy(), G["cur_levels_after"])
    np.testing.assert_array_equal(ns.env_origins.numpy(), G["cur_origins_after"])
    moved = (G["cur_levels_after"] != G["cur_levels_before"]).sum()
    assert moved > 20
    # the oracle's restatement, given the reference's random levels where they were used
    e = G["cur_env_ids"]
    lv, org = to.update_terrain_curriculum(G["cur_root_xy"][e], G["cur_origins_before"][e, :2], G["cur_commands"][e, :2], G["cur_levels_before"][e],
                                           G["cur_types"][e], G["cur_terrain_origins"], 8.0, 10.0, int(G["grid_cfg"][0]), G["cur_levels_after"][e])
    np.testing.assert_array_equal(lv, G["cur_levels_after"][e])
    np.testing.assert_array_equal(org, G["cur_origins_after"][e])


@pytest.mark.gpu
def test_get_heights_kernel_matches_reference_bit_exact():
    """wbc_get_heights (csrc/wbc_terrain_kernel.hip) against the reference's LeggedRobot._get_heights output."""
    from wbc_amd.native import check, lib
    root = np.ascontiguousarray(G["heights_root"])
    pts = np.ascontiguousarray(G["heights_points"])
    H = np.ascontiguousarray(G["grid_height_field"])
    q, p, b, h = (torch.from_numpy(x).cuda() for x in (np.ascontiguousarray(root[:, 3:7]), root, pts, H))
    out = torch.full((pts.shape[0], pts.shape[1]), float("nan"), device="cuda")
    check(lib().wbc_get_heights(q.data_ptr(), q.stride(0), p.data_ptr(), p.stride(0), b.data_ptr(), h.data_ptr(), H.shape[0], H.shape[1],
                                float(G["grid_cfg"][2]), 0.1, 0.005, out.data_ptr(), pts.shape[0], pts.shape[1],
                                torch.cuda.current_stream().cuda_stream), "wbc_get_heights")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), G["heights_out"])
