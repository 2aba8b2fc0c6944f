"""wbc_get_heights (csrc/wbc_terrain_kernel.hip) bit for bit against oracle/terrain_oracle.py, the env's measure_heights path,
and the base class's terrain-level curriculum (legged_robot.py:421-441, 717-731, 793-829) on the fused step."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
pytestmark = pytest.mark.gpu
import terrain_oracle as to  # noqa: E402


def _gpu_heights(quat, pos, pts, H, border, hs, vs):
    import torch
    from wbc_amd.native import check, lib
    q, p, b, h = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (quat, pos, pts, H))
    out = torch.full((pts.shape[0], pts.shape[1]), float("nan"), device="cuda")
    check(lib().wbc_get_heights(q.data_ptr(), q.stride(0), p.data_ptr(), p.stride(0), b.data_ptr(), h.data_ptr(), H.shape[0], H.shape[1],
                                float(border), float(hs), float(vs), out.data_ptr(), pts.shape[0], pts.shape[1],
                                torch.cuda.current_stream().cuda_stream), "wbc_get_heights")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("hs,vs,border", [(0.025, 1e-5, 0.0), (0.1, 0.005, 25.0), (0.25, 0.01, 0.0)])
def test_get_heights_bit_exact(hs, vs, border):
    """Random general quaternions, positions from far below to far beyond the grid, and positions that sit exactly on cell
    boundaries (multiples of the scale): the fp32 index arithmetic must agree in every bit."""
    rng = np.random.default_rng(int(hs * 1000))
    n, rows, cols = 333, 120, 90
    H = rng.integers(-32768, 32767, size=(rows, cols), endpoint=True).astype(np.int16)
    quat = rng.normal(size=(n, 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    quat[0] = (0, 0, 0, 1)
    quat[1] = (0.6, 0.8, 0, 0)                       # degenerate yaw part: normalize() clamps the norm at 1e-9
    pos = np.zeros((n, 13), np.float32)
    pos[:, :2] = rng.uniform(-3.0 - border, rows * hs + 3.0 - border, size=(n, 2))
    pos[: n // 3, :2] = (rng.integers(-4, rows + 4, size=(n // 3, 2)) * np.float32(hs)).astype(np.float32) - np.float32(border)
    pts = to.init_height_points(np.arange(-0.8, 0.81, 0.1), np.arange(-0.5, 0.51, 0.1), n)      # 17 x 11 = 187 points
    pts[n // 2:] = (np.round(pts[n // 2:] / hs) * hs).astype(np.float32)                          # points on the grid lines too
    ref = to.get_heights(quat, pos, pts, H, border, hs, vs)
    got = _gpu_heights(quat, pos, pts, H, border, hs, vs)
    assert ref.shape == got.shape == (n, 187)
    np.testing.assert_array_equal(got, ref)
    assert len(np.unique(ref)) > 1000                 # the lookup really moved over the grid


def _cfg(n):
    from wbc_amd.config import WidowGo1RoughCfg
    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = n
    cfg.terrain.tot_rows = 2000                      # 50 m strip: same generator, faster test
    cfg.terrain.transform_y = -cfg.terrain.tot_rows * cfg.terrain.horizontal_scale / 2
    return cfg


def test_env_measure_heights_matches_oracle():
    import torch
    from wbc_amd.envs import WidowGo1
    cfg = _cfg(96)
    cfg.terrain.measure_heights = True
    env = WidowGo1(cfg, sim_device="cuda:0", seed=4)
    assert env.height_points.shape == (96, 187, 3) and env.num_height_points == 187               # WG:637-638, LR:777-791
    env.reset()
    for _ in range(5):
        env.step(0.3 * torch.randn(96, 18, device="cuda"))
    t = cfg.terrain
    ref = to.get_heights(env.base_quat.cpu().numpy(), env.root_states.cpu().numpy(), env.height_points.cpu().numpy(),
                         env.height_samples.cpu().numpy(), t.border_size, t.horizontal_scale, t.vertical_scale)
    assert env.measured_heights.shape == (96, 187)                                                # WG:932-933
    np.testing.assert_array_equal(env._get_heights().cpu().numpy(), ref)
    ids = [5, 17, 60]
    np.testing.assert_array_equal(env._get_heights(ids).cpu().numpy(), ref[ids])                  # LR:811-812
    cfg2 = _cfg(8)
    cfg2.terrain.mesh_type = "plane"
    cfg2.terrain.measure_heights = True
    flat = WidowGo1(cfg2, sim_device="cuda:0", seed=4)
    assert flat._get_heights().abs().max() == 0 and flat._get_heights().shape == (8, 187)         # LR:806-807


def test_terrain_level_curriculum_on_the_fused_step():
    """terrain.curriculum=True: levels / types / origins as LR:717-731; on every reset the level moves by the rule of
    LR:431-441 evaluated on the finished episode's travel (oracle), env_origins follow, and the robot is re-placed on its
    new platform."""
    import torch
    from wbc_amd.envs import WidowGo1
    n = 240
    cfg = _cfg(n)
    cfg.terrain.curriculum = True
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.max_init_terrain_level = 4, 6, 2
    cfg.env.episode_length_s = 0.4                                      # 20 steps: plenty of resets
    env = WidowGo1(cfg, sim_device="cuda:0", seed=6)
    assert env.terrain_origins.shape == (4, 6, 3) and env.max_terrain_level == 4
    assert int(env.terrain_levels.max()) <= 2 and int(env.terrain_levels.min()) >= 0              # max_init_terrain_level
    np.testing.assert_array_equal(env.terrain_types.cpu().numpy(), np.floor(np.arange(n) / (n / 6)).astype(np.int64))
    origins_tab = env.terrain_origins.cpu().numpy()
    np.testing.assert_array_equal(env.env_origins.cpu().numpy(), origins_tab[env.terrain_levels.cpu().numpy(), env.terrain_types.cpu().numpy()])
    env.reset()
    torch.manual_seed(0)
    changed = ups = downs = 0
    for step in range(70):
        lv0 = env.terrain_levels.cpu().numpy().copy()
        a = torch.zeros(n, 18, device="cuda") if step % 2 else 0.5 * torch.randn(n, 18, device="cuda")
        env.step(a)
        m = env.reset_buf.cpu().numpy().astype(bool)
        lv1 = env.terrain_levels.cpu().numpy()
        np.testing.assert_array_equal(lv1[~m], lv0[~m])                 # only resetting envs move
        if m.any():
            tr = env.sim.tensor("RESET_TRAVEL").cpu().numpy()[m]
            dummy_xy = np.zeros((m.sum(), 2), np.float32)
            exp, _ = to.update_terrain_curriculum(np.stack([tr[:, 0], 0 * tr[:, 0]], 1), dummy_xy, np.stack([tr[:, 1], 0 * tr[:, 1]], 1),
                                                  lv0[m], env.terrain_types.cpu().numpy()[m], origins_tab, env.terrain.env_length,
                                                  env.max_episode_length_s, 4, random_levels=np.full(m.sum(), -1))
            solved = exp == -1                                           # solved the last level: any level is allowed
            np.testing.assert_array_equal(lv1[m][~solved], exp[~solved])
            assert ((lv1[m][solved] >= 0) & (lv1[m][solved] < 4)).all()
            ups += int((lv1[m] > lv0[m]).sum()); downs += int((lv1[m] < lv0[m]).sum()); changed += int(m.sum())
        o = env.env_origins.cpu().numpy()
        np.testing.assert_array_equal(o, origins_tab[lv1, env.terrain_types.cpu().numpy()])
        np.testing.assert_array_equal(env.sim.tensor("ENV_ORIGINS").cpu().numpy(), o)
        if m.any():                                                      # re-placed on the new platform (WG:765-767)
            root = env.root_states.cpu().numpy()[m]
            assert np.abs(root[:, :2] - o[m, :2]).max() <= cfg.terrain.origin_perturb_range + 1e-5
            np.testing.assert_allclose(root[:, 2], cfg.init_state.pos[2] + o[m, 2], atol=1e-5)
    assert changed > 200 and ups + downs > 0
    assert torch.isfinite(env.obs_buf).all()


def test_subterrain_grid_env_with_curriculum():
    """BASELINE.json configs[2] as surveyed: the base class's Terrain grid (levels x types of
    8 m tiles, LRC:43-66) under the widowGo1 task with terrain.curriculum=True: placement on the platforms (LR:717-731),
    heightfield contact on the stairs / slopes, levels moving by LR:421-441 on resets, robots re-placed on the new platform."""
    import torch
    from wbc_amd.config import WidowGo1RoughCfg, use_grid_terrain
    from wbc_amd.envs import WidowGo1
    n = 360
    cfg = use_grid_terrain(WidowGo1RoughCfg(), num_rows=4, num_cols=6)
    cfg.env.num_envs = n
    cfg.terrain.max_init_terrain_level = 2
    cfg.env.episode_length_s = 0.5
    env = WidowGo1(cfg, sim_device="cuda:0", seed=8)
    env.update_command_curriculum()                                      # counter 1: the final command ranges (v_x up to 0.9 m/s)
    t = env.terrain
    assert t.heightsamples.shape == (4 * 80 + 500, 6 * 80 + 500) and env.terrain_origins.shape == (4, 6, 3)
    assert env.terrain.env_length == 8.0 and env.max_terrain_level == 4
    tab = env.terrain_origins.cpu().numpy()
    np.testing.assert_allclose(tab[..., 0], (np.arange(4)[:, None] + 0.5) * 8.0 * np.ones((1, 6)))
    env.reset()
    z0 = env.root_states[:, 2].cpu().numpy() - env.env_origins[:, 2].cpu().numpy()
    assert np.all(z0 > 0.25) and np.all(z0 < 0.5)                        # standing on their platforms, whatever the tile's height
    torch.manual_seed(0)
    moved = 0
    for step in range(60):
        lv0 = env.terrain_levels.clone()
        env.step(0.4 * torch.randn(n, 18, device="cuda"))
        m = env.reset_buf.bool()
        assert torch.equal(env.terrain_levels[~m], lv0[~m])
        moved += int((env.terrain_levels != lv0).sum())
        o = env.env_origins.cpu().numpy()
        np.testing.assert_array_equal(o, tab[env.terrain_levels.cpu().numpy(), env.terrain_types.cpu().numpy()])
    assert moved > 0 and torch.isfinite(env.obs_buf).all()
    assert (env.root_states[:, 2] > -1.5).all()                           # nobody fell through the terrain
