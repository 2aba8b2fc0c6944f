"""The base class's terrain bookkeeping (legged_robot.py:421-441, 717-731, 777-829): the oracle restatement on hand-computed
cases (CPU). The HIP kernel is compared bit for bit with it in tests/test_gpu_terrain.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import terrain_oracle as to  # noqa: E402


def test_get_heights_indexing_hand_cases():
    """LR:816-829: += border, / scale, .long() truncates TOWARD ZERO, clip to [0, dim-2], min of three corner samples."""
    H = (np.arange(20, dtype=np.int16).reshape(4, 5) * 100)
    q = np.array([[0, 0, 0, 1]], np.float32)
    pos = np.array([[0.26, 0.51, 0.3]], np.float32)
    pts = np.array([[[0, 0, 0], [-1, 0, 0], [10, 10, 0], [0, -0.6, 0]]], np.float32)
    out = to.get_heights(q, pos, pts, H, 0.0, 0.25, 0.01)
    # (0.26, 0.51)/0.25 -> (1, 2): min(700, 1200, 800); x = -0.74 -> -2.96 -> -2 -> clipped 0: min(200, 700, 300);
    # far outside -> (rows-2, cols-2) = (2, 3): min(1300, 1800, 1400); y = -0.09 -> -0.36 -> 0 (toward zero, not -1): min(500, 1000, 600)
    np.testing.assert_array_equal(out, np.array([[7.0, 2.0, 13.0, 5.0]], np.float32))
    # border shifts the lookup: +0.25 in x and y moves (1, 2) to (2, 3)
    np.testing.assert_array_equal(to.get_heights(q, pos, pts[:, :1], H, 0.25, 0.25, 0.01), np.array([[13.0]], np.float32))


def test_get_heights_uses_yaw_only():
    """quat_apply_yaw (utils/math.py:38-42): roll and pitch of the base do not move the sample points."""
    H = np.random.default_rng(0).integers(-3000, 3000, size=(40, 50)).astype(np.int16)
    pos = np.array([[2.0, 2.5, 0.4]], np.float32)
    pts = to.init_height_points(np.linspace(-0.8, 0.8, 5), np.linspace(-0.5, 0.5, 3), 1)
    yaw = 0.7
    qy = np.array([[0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]], np.float32)
    # the same yaw with 0.3 rad of pitch on top: q = q_yaw * q_pitch
    sp, cp = np.sin(0.15), np.cos(0.15)
    qyp = np.array([[-qy[0, 2] * sp, qy[0, 3] * sp, qy[0, 2] * cp, qy[0, 3] * cp]], np.float32)
    a = to.get_heights(qy, pos, pts, H, 0.0, 0.1, 0.005)
    b = to.get_heights(qyp, pos, pts, H, 0.0, 0.1, 0.005)
    np.testing.assert_array_equal(a, b)
    # and the points are really rotated: the first point (-0.8, -0.5) lands at pos + R(yaw) p
    c, s_ = np.cos(yaw), np.sin(yaw)
    p = pos[0, :2] + np.array([c * -0.8 - s_ * -0.5, s_ * -0.8 + c * -0.5])
    ix, iy = int(p[0] / 0.1), int(p[1] / 0.1)
    assert a[0, 0] == np.float32(min(H[ix, iy], H[ix + 1, iy], H[ix, iy + 1])) * np.float32(0.005)


def test_init_height_points_order():
    pts = to.init_height_points([-0.1, 0.1], [0.0, 0.2, 0.4], 3)             # LR:783-790: meshgrid(x, y), x-major
    assert pts.shape == (3, 6, 3)
    np.testing.assert_allclose(pts[1, :, 0], [-0.1, -0.1, -0.1, 0.1, 0.1, 0.1])
    np.testing.assert_allclose(pts[1, :, 1], [0.0, 0.2, 0.4, 0.0, 0.2, 0.4])
    assert (pts[:, :, 2] == 0).all()


def test_update_terrain_curriculum_rule():
    """LR:431-441: up if walked more than half a platform, down if less than half the commanded distance (and not up),
    solved-last-level -> the random draw, never below 0."""
    origins = np.zeros((3, 2, 3), np.float32)
    origins[:, :, 0] = np.arange(3)[:, None] * 8.0
    origins[:, :, 1] = np.arange(2)[None, :] * 8.0
    env_o = np.array([[0, 0], [8, 8], [16, 0], [0, 8], [8, 0]], np.float32)
    root = env_o + np.array([[5, 0], [0.5, 0], [4.5, 0], [0.2, 0], [3, 0]], np.float32)     # travelled 5, 0.5, 4.5, 0.2, 3
    cmd = np.array([[1, 0], [1, 0], [0, 0], [0.5, 0], [0.2, 0]], np.float32)                # x 10 s x 0.5 -> 5, 5, 0, 2.5, 1
    levels = np.array([0, 1, 2, 0, 1])
    types = np.array([0, 1, 0, 1, 0])
    lv, new_o = to.update_terrain_curriculum(root, env_o, cmd, levels, types, origins, env_length=8.0, max_episode_length_s=10.0,
                                             max_terrain_level=3, random_levels=np.array([1, 1, 1, 1, 1]))
    # env0: 5 > 4 up -> 1; env1: 0.5 < 5 down -> 0; env2: 4.5 > 4 up -> 3 = max -> random 1; env3: 0.2 < 2.5 down -> clip(-1) = 0;
    # env4: 3 neither (3 < 4, 3 > 1) -> stays 1
    np.testing.assert_array_equal(lv, [1, 0, 1, 0, 1])
    np.testing.assert_array_equal(new_o, origins[lv, types])


def test_level_grid_of_the_perlin_field():
    from wbc_amd.config import WidowGo1RoughCfg
    from wbc_amd.terrain import TerrainPerlin
    cfg = WidowGo1RoughCfg()
    cfg.terrain.tot_rows = 2000
    cfg.terrain.transform_y = -cfg.terrain.tot_rows * cfg.terrain.horizontal_scale / 2
    t = TerrainPerlin(cfg.terrain, seed=3)
    o = t.level_grid(4, 5)
    assert o.shape == (4, 5, 3) and t.env_length == t.flat_beyond_row * t.horizontal_scale / 4
    x0, y0 = t.transform[0], t.transform[1]
    assert (o[:, :, 0] > x0).all() and (o[:, :, 0] < x0 + t.flat_beyond_row * t.horizontal_scale).all()
    assert (np.diff(o[:, 0, 0]) > 0).all() and (np.diff(o[0, :, 1]) > 0).all()
    hmax = t.heightsamples[:t.flat_beyond_row].max() * t.vertical_scale
    assert (o[:, :, 2] <= hmax + 1e-9).all() and (o[:, :, 2] >= t.heightsamples.min() * t.vertical_scale).all()
