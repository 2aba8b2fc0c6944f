"""TEST INFRASTRUCTURE (oracle): CPU restatement of the base class's terrain bookkeeping -- never imported by the product.

    init_height_points           legged_gym/envs/base/legged_robot.py:777-791
    get_heights                  legged_robot.py:793-829 (+ utils/math.py:38-42 quat_apply_yaw; isaacgym.torch_utils
                                 quat_apply / normalize, un-vendored: restated from their published definitions)
    get_env_origins_levels       legged_robot.py:717-731 (the terrain-level branch)
    update_terrain_curriculum    legged_robot.py:421-441

Parity: the reference needs Isaac Gym to instantiate LeggedRobot, so these cannot be run here ("parity unpinned" against
reference outputs); each function follows the cited lines operation by operation. get_heights is integer / index work and
is evaluated one rounded fp32 operation at a time (numpy float32 scalars' arithmetic is IEEE single, no contraction), with
`p / horizontal_scale` as PyTorch's CUDA/HIP kernel evaluates a division by a Python scalar: p * fp32(1 / horizontal_scale)
(aten/src/ATen/native/cuda/BinaryDivTrueKernel.cu). The HIP kernel (csrc/wbc_terrain_kernel.hip) must match bit for bit."""
import numpy as np

f32 = np.float32


def init_height_points(measured_points_x, measured_points_y, num_envs):
    """LR:777-791: meshgrid(x, y) (indexing 'ij'), flattened; z = 0. -> [N, P, 3] float32"""
    gx, gy = np.meshgrid(np.asarray(measured_points_x, f32), np.asarray(measured_points_y, f32), indexing="ij")
    pts = np.zeros((num_envs, gx.size, 3), f32)
    pts[:, :, 0] = gx.ravel()
    pts[:, :, 1] = gy.ravel()
    return pts


def get_heights(base_quat, root_pos, height_points, height_samples, border_size, horizontal_scale, vertical_scale):
    """LR:793-829 for mesh types heightfield / trimesh. base_quat [N,4] xyzw, root_pos [N,>=2], height_points [N,P,3],
    height_samples [rows, cols] int16 -> [N, P] float32."""
    q = np.asarray(base_quat, f32)
    pos = np.asarray(root_pos, f32)
    b = np.asarray(height_points, f32)
    H = np.asarray(height_samples)
    rows, cols = H.shape
    z, w = q[:, 2], q[:, 3]
    n = np.sqrt(z * z + w * w)                       # float32 arrays: every * and + is one rounded fp32 operation
    n = np.maximum(n, f32(1e-9))
    qz, qw = (z / n)[:, None], (w / n)[:, None]
    bx, by = b[:, :, 0], b[:, :, 1]
    tx = -(qz * by) * f32(2)
    ty = (qz * bx) * f32(2)
    rx = (bx + qw * tx) + (-(qz * ty))
    ry = (by + qw * ty) + (qz * tx)
    px = (rx + pos[:, 0:1]) + f32(border_size)
    py = (ry + pos[:, 1:2]) + f32(border_size)
    inv = f32(1.0) / f32(horizontal_scale)
    ix = np.trunc(px * inv).astype(np.int64)         # .long(): toward zero
    iy = np.trunc(py * inv).astype(np.int64)
    ix = np.clip(ix, 0, rows - 2)
    iy = np.clip(iy, 0, cols - 2)
    h = np.minimum(np.minimum(H[ix, iy], H[ix + 1, iy]), H[ix, iy + 1])
    return h.astype(f32) * f32(vertical_scale)


def get_env_origins_levels(num_envs, num_rows, num_cols, max_init_terrain_level, curriculum, terrain_env_origins, rng):
    """LR:717-731. terrain_env_origins [num_rows, num_cols, 3]. -> (env_origins [N,3] f32, levels i64, types i64)"""
    max_init = max_init_terrain_level if curriculum else num_rows - 1
    levels = rng.integers(0, max_init + 1, size=num_envs, dtype=np.int64)
    types = np.floor(np.arange(num_envs) / (num_envs / num_cols)).astype(np.int64)
    origins = np.asarray(terrain_env_origins, f32)[levels, types]
    return origins, levels, types


def update_terrain_curriculum(root_xy, env_origins_xy, commands_xy, levels, types, terrain_env_origins, env_length, max_episode_length_s,
                              max_terrain_level, random_levels):
    """LR:421-441 for the envs being reset (all arrays already restricted to them). random_levels: the draw of
    torch.randint_like(levels, max_terrain_level). -> (new levels i64, new env_origins [n,3] f32)"""
    d = np.asarray(root_xy, f32) - np.asarray(env_origins_xy, f32)
    distance = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
    move_up = distance > f32(env_length / 2)
    c = np.asarray(commands_xy, f32)
    cn = np.sqrt(c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1])
    move_down = (distance < (cn * f32(max_episode_length_s)) * f32(0.5)) & ~move_up
    lv = np.asarray(levels, np.int64) + move_up.astype(np.int64) - move_down.astype(np.int64)
    lv = np.where(lv >= max_terrain_level, np.asarray(random_levels, np.int64), np.clip(lv, 0, None))
    return lv, np.asarray(terrain_env_origins, f32)[lv, np.asarray(types, np.int64)]
