"""Host-side terrain generation for the widowGo1 task: the fractal-Perlin height grid of
legged_gym/utils/terrain.py:40-99 (`Terrain_Perlin`), produced as the int16 sample grid that
`wbc_sim_set_heightfield` consumes (the kernels collide against the same triangulation Isaac Gym's
`convert_heightfield_to_trimesh` would build from it, so no vertex/triangle arrays are needed).

Reproduced semantics: 2 octaves of gradient noise, base frequency 10 cells per metre of the SHORT
side convention of the reference (`xScale = frequency * xSize`), lacunarity 2, gain 0.25, each octave
mapped to [0, 1] and scaled by zScale; samples stored as round-toward-zero(h / vertical_scale) int16.
Quirk Q3: the reference adds 100 km to rows >= tot_cols//2 - 100 and casts to int16; on
x86-64 numpy that out-of-range cast yields 0, i.e. the far part of the map is FLAT at z = 0. That is
what `flat_beyond_row` reproduces, explicitly instead of through an overflow.
The random stream is numpy's Generator seeded by the caller, not the reference's global np.random.
"""
from __future__ import annotations

import numpy as np


def _fade(t):
    return t * t * t * (t * (6.0 * t - 15.0) + 10.0)


def perlin_2d(shape, periods, rng) -> np.ndarray:
    """Gradient noise on a (shape[0], shape[1]) grid with periods[0] x periods[1] lattice cells, in [0, 1]."""
    nx, ny = shape
    px, py = periods
    assert nx % px == 0 and ny % py == 0, "grid must be divisible by the lattice (terrain.py:45 has the same assert)"
    ang = 2.0 * np.pi * rng.random((px + 1, py + 1))
    gx, gy = np.cos(ang), np.sin(ang)
    u = (np.arange(nx) * (px / nx))
    v = (np.arange(ny) * (py / ny))
    ix, iy = u.astype(np.int64), v.astype(np.int64)
    fx, fy = (u - ix)[:, None], (v - iy)[None, :]
    ix, iy = ix[:, None], iy[None, :]

    def corner(dx, dy):
        return gx[ix + dx, iy + dy] * (fx - dx) + gy[ix + dx, iy + dy] * (fy - dy)
    sx, sy = _fade(fx), _fade(fy)
    n0 = corner(0, 0) * (1 - sx) + corner(1, 0) * sx
    n1 = corner(0, 1) * (1 - sx) + corner(1, 1) * sx
    return np.sqrt(2.0) * (n0 * (1 - sy) + n1 * sy) * 0.5 + 0.5


def fractal_noise(x_size, y_size, x_samples, y_samples, rng, frequency=10, octaves=2, lacunarity=2.0, gain=0.25, z_scale=0.23):
    px, py = frequency * x_size, frequency * y_size
    amp = 1.0
    out = np.zeros((x_samples, y_samples))
    for _ in range(octaves):
        out += amp * z_scale * perlin_2d((x_samples, y_samples), (px, py), rng)
        amp *= gain
        px, py = int(lacunarity * px), int(lacunarity * py)
    return out


class TerrainPerlin:
    """Attributes mirror the reference object: tot_cols, tot_rows, heightsamples (int16 [tot_cols, tot_rows]),
    heightsamples_float; plus the placement the task uses (transform_x/y/z, horizontal/vertical scale)."""

    def __init__(self, cfg, seed: int = 0):
        self.cfg = cfg
        self.tot_cols, self.tot_rows = int(cfg.tot_cols), int(cfg.tot_rows)
        x_size, y_size = cfg.horizontal_scale * self.tot_cols, cfg.horizontal_scale * self.tot_rows
        assert x_size == int(x_size) and y_size == int(y_size)
        rng = np.random.default_rng(seed)
        self.heightsamples_float = fractal_noise(int(x_size), int(y_size), self.tot_cols, self.tot_rows, rng, z_scale=cfg.zScale)
        self.flat_beyond_row = self.tot_cols // 2 - 100
        hs = np.trunc(self.heightsamples_float / cfg.vertical_scale)
        hs[self.flat_beyond_row:, :] = 0            # quirk Q3, see module docstring
        self.heightsamples = np.clip(hs, -32768, 32767).astype(np.int16)
        self.horizontal_scale, self.vertical_scale = float(cfg.horizontal_scale), float(cfg.vertical_scale)
        self.transform = (float(cfg.transform_x), float(cfg.transform_y), float(cfg.transform_z))

    def level_grid(self, num_rows: int, num_cols: int):
        """What the base class's terrain curriculum needs from a terrain object -- `env_origins` [num_rows, num_cols, 3],
        `env_length`, `env_width` (utils/terrain.py Terrain: one platform per (level, type), origin z = highest sample of
        the 2 m x 2 m patch around the platform centre) -- for this field: its walkable part (x below the flattened rows,
        quirk Q3; all of y) cut into num_rows levels along x and num_cols types along y. The reference's widowGo1 cannot
        run with terrain.curriculum=True (its _get_env_origins override never creates terrain_levels, and its Perlin terrain
        has no env_length), so this grid is this framework's choice; the curriculum rule itself is LR:421-441."""
        hs, vs = self.horizontal_scale, self.vertical_scale
        nx, ny = self.flat_beyond_row, self.heightsamples.shape[1]
        self.env_length, self.env_width = nx * hs / num_rows, ny * hs / num_cols
        self.env_origins = np.zeros((num_rows, num_cols, 3))
        half = int(round(1.0 / hs))
        for i in range(num_rows):
            for j in range(num_cols):
                cx, cy = (i + 0.5) * self.env_length, (j + 0.5) * self.env_width
                ix, iy = int(cx / hs), int(cy / hs)
                patch = self.heightsamples[max(ix - half, 0):min(ix + half, nx), max(iy - half, 0):min(iy + half, ny)]
                self.env_origins[i, j] = (cx + self.transform[0], cy + self.transform[1], float(patch.max()) * vs + self.transform[2])
        return self.env_origins
