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
The random stream is numpy's Generator seeded by the caller, not the reference's global np.random; fed the same
uniforms (tools/make_golden_terrain.py patches the reference's np.random.rand with the same Generator) the two int16
grids are identical: the float64 field is evaluated in the reference's operation order (tests/test_terrain_golden.py).
"""
from __future__ import annotations

import numpy as np


def _fade(t):
    # the reference's polynomial AS WRITTEN (terrain.py:66: numpy powers, this operation order): the Horner form differs from it
    # by an ulp here and there, which moves a sample across an int16 truncation boundary on 1 cell in 1000
    return 6 * t**5 - 15 * t**4 + 10 * t**3


def perlin_2d(shape, periods, rng) -> np.ndarray:
    """Gradient noise on a (shape[0], shape[1]) grid with periods[0] x periods[1] lattice cells, in [0, 1]: terrain.py:64-89
    operation by operation (same fractional coordinates i * delta % 1, lattice cell = i // (samples per cell) as its
    `repeat`, same ramp / interpolation order), so that the same uniforms give the same float64 field bit for bit. Broadcast
    1-D coordinate vectors stand in for the reference's [nx, ny, 2] mgrid."""
    nx, ny = shape
    px, py = periods
    assert nx % px == 0 and ny % py == 0, "grid must be divisible by the lattice (terrain.py:45 has the same assert)"
    ang = 2 * np.pi * rng.random((px + 1, py + 1))
    gx, gy = np.cos(ang), np.sin(ang)
    fx = ((np.arange(nx) * (px / nx)) % 1)[:, None]          # np.mgrid[0:res:delta] = index * delta (+ 0)
    fy = ((np.arange(ny) * (py / ny)) % 1)[None, :]
    ix = (np.arange(nx) // (nx // px))[:, None]               # gradients[...].repeat(d, axis)
    iy = (np.arange(ny) // (ny // py))[None, :]

    def corner(dx, dy):                                      # np.sum(dstack((grid0 - dx, grid1 - dy)) * g, 2)
        return (fx - dx if dx else fx) * gx[ix + dx, iy + dy] + (fy - dy if dy else fy) * gy[ix + dx, iy + dy]
    tx, ty = _fade(fx), _fade(fy)
    n0 = corner(0, 0) * (1 - tx) + tx * corner(1, 0)
    n1 = corner(0, 1) * (1 - tx) + tx * corner(1, 1)
    return np.sqrt(2) * ((1 - ty) * n0 + ty * n1) * 0.5 + 0.5


def fractal_noise(x_size, y_size, x_samples, y_samples, rng, frequency=10, octaves=2, lacunarity=2.0, gain=0.25, z_scale=0.23):
    px, py = frequency * x_size, frequency * y_size
    amp = 1
    out = np.zeros((x_samples, y_samples))
    for _ in range(octaves):
        out += amp * perlin_2d((x_samples, y_samples), (px, py), rng) * z_scale      # terrain.py:96: (amplitude * noise) * zScale
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
        hs = np.trunc(self.heightsamples_float * (1 / cfg.vertical_scale))       # terrain.py:51: h * (1 / vertical_scale), cast = truncation
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


# ---------------------------------------------------------------------------------------------------------------------
# The base class's terrain: a grid of num_rows (difficulty levels) x num_cols (terrain types) square tiles
# (legged_gym/utils/terrain.py:101-250, used with LeggedRobotCfg.terrain, legged_robot_config.py:43-66).
def _gap(tile, gap_size, platform_size=1.0):                                       # terrain.py:229-241
    gap, plat = int(gap_size / tile.horizontal_scale), int(platform_size / tile.horizontal_scale)
    cx, cy = tile.length // 2, tile.width // 2
    x1, y1 = (tile.length - plat) // 2, (tile.width - plat) // 2
    x2, y2 = x1 + gap, y1 + gap
    tile.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    tile.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def _pit(tile, depth, platform_size=1.0):                                          # terrain.py:243-250
    d, half = int(depth / tile.vertical_scale), int(platform_size / tile.horizontal_scale / 2)
    x1, x2 = tile.length // 2 - half, tile.length // 2 + half
    y1, y2 = tile.width // 2 - half, tile.width // 2 + half
    tile.height_field_raw[x1:x2, y1:y2] = -d


class Terrain:
    """Attributes as the reference object: env_length / env_width, proportions (cumulative), border (cells), tot_rows /
    tot_cols, height_field_raw = heightsamples (int16 [tot_rows, tot_cols]), env_origins [num_rows, num_cols, 3]
    (tile centre, z = highest sample of the 2 m x 2 m patch around it), and for mesh_type 'trimesh' vertices / triangles.
    Tile (i, j) has difficulty i / num_rows and type j / num_cols + 0.001 under cfg.curriculum; otherwise a random type and
    one of three difficulties per tile (np.random, same call order as the reference)."""

    def __init__(self, cfg, num_robots) -> None:
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        from . import terrain_utils
        self._tu = terrain_utils
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        hs = cfg.horizontal_scale
        self.width_per_env_pixels, self.length_per_env_pixels = int(self.env_width / hs), int(self.env_length / hs)
        self.border = int(cfg.border_size / hs)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        tiles = [(i, j) for j in range(cfg.num_cols) for i in range(cfg.num_rows)] if cfg.curriculum else \
                [tuple(np.unravel_index(k, (cfg.num_rows, cfg.num_cols))) for k in range(cfg.num_sub_terrains)]
        if not cfg.curriculum and getattr(cfg, "selected", False):
            raise NotImplementedError("terrain.selected (a single generator by name) is broken in the reference too (terrain.py:160-173 "
                                      "reads attributes the class never sets)")
        for i, j in tiles:
            if cfg.curriculum:
                difficulty, choice = i / cfg.num_rows, j / cfg.num_cols + 0.001
            else:
                choice = np.random.uniform(0, 1)
                difficulty = np.random.choice([0.5, 0.75, 0.9])
            self._place(self.make_terrain(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw
        self.horizontal_scale, self.vertical_scale = float(hs), float(cfg.vertical_scale)
        self.transform = (-float(cfg.border_size), -float(cfg.border_size), 0.0)           # LR:_create_trimesh / _create_heightfield
        if self.type == "trimesh":
            self.vertices, self.triangles = terrain_utils.convert_heightfield_to_trimesh(self.height_field_raw, hs, cfg.vertical_scale,
                                                                                         cfg.slope_treshold)

    def make_terrain(self, choice, difficulty):
        tu, cfg = self._tu, self.cfg
        tile = tu.SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                             vertical_scale=cfg.vertical_scale, horizontal_scale=cfg.horizontal_scale)
        slope, step_h = difficulty * 0.4, 0.05 + 0.18 * difficulty
        P = list(self.proportions) + [np.inf] * 8                   # types beyond the configured proportions are never chosen
        if choice < P[0]:                                           # smooth slope (lower half of the band: inverted)
            tu.pyramid_sloped_terrain(tile, slope=-slope if choice < P[0] / 2 else slope, platform_size=3.)
        elif choice < P[1]:                                         # rough slope
            tu.pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
            tu.random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < P[3]:                                         # stairs: below P[2] descending, else ascending
            tu.pyramid_stairs_terrain(tile, step_width=0.31, step_height=-step_h if choice < P[2] else step_h, platform_size=3.)
        elif choice < P[4]:                                         # discrete obstacles
            tu.discrete_obstacles_terrain(tile, 0.05 + difficulty * 0.2, 1., 2., 20, platform_size=3.)
        elif choice < P[5]:
            tu.stepping_stones_terrain(tile, stone_size=1.5 * (1.05 - difficulty), stone_distance=0.05 if difficulty == 0 else 0.1,
                                       max_height=0., platform_size=4.)
        elif choice < P[6]:
            _gap(tile, gap_size=1. * difficulty, platform_size=3.)
        else:
            _pit(tile, depth=1. * difficulty, platform_size=4.)
        return tile

    def _place(self, tile, i, j):
        x0, y0 = self.border + i * self.length_per_env_pixels, self.border + j * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = tile.height_field_raw
        hs = tile.horizontal_scale
        x1, x2 = int((self.env_length / 2. - 1) / hs), int((self.env_length / 2. + 1) / hs)
        y1, y2 = int((self.env_width / 2. - 1) / hs), int((self.env_width / 2. + 1) / hs)
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width,
                                  np.max(tile.height_field_raw[x1:x2, y1:y2]) * tile.vertical_scale]

    add_terrain_to_map = _place
