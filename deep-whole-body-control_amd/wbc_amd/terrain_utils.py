"""Height-field generators of `isaacgym.terrain_utils` that the base-class terrain of the reference calls
(legged_gym/utils/terrain.py:133-202: SubTerrain, pyramid_sloped_terrain, random_uniform_terrain,
pyramid_stairs_terrain, discrete_obstacles_terrain, stepping_stones_terrain, convert_heightfield_to_trimesh).

`isaacgym` (Isaac Gym Preview 3, legged_gym/README.md:17) is a proprietary wheel that is NOT in the reference tree;
what follows restates the published behaviour of those generators (integer height grids in units of
`vertical_scale`, sizes converted with int() truncation, numpy's global generator in the documented call order).
Parity for these functions themselves is UNPINNED (no reference output exists here); what is pinned
(tests/test_terrain_golden.py) is the reference's own `Terrain` class running on top of them.

All generators add to / overwrite `terrain.height_field_raw` (int16 [width, length]) in place and return the terrain."""
from __future__ import annotations

import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _bilinear_resample(z, out_rows, out_cols):
    """Bilinear interpolation of a regular grid onto out_rows x out_cols points spanning the same extent (end points
    included) -- what scipy's interp2d(kind='linear') did for gridded data."""
    r = np.linspace(0.0, z.shape[0] - 1.0, out_rows)
    c = np.linspace(0.0, z.shape[1] - 1.0, out_cols)
    r0 = np.clip(np.floor(r).astype(np.int64), 0, max(z.shape[0] - 2, 0))
    c0 = np.clip(np.floor(c).astype(np.int64), 0, max(z.shape[1] - 2, 0))
    fr, fc = (r - r0)[:, None], (c - c0)[None, :]
    r1, c1 = np.minimum(r0 + 1, z.shape[0] - 1), np.minimum(c0 + 1, z.shape[1] - 1)
    z = z.astype(np.float64)
    top = z[r0][:, c0] * (1 - fc) + z[r0][:, c1] * fc
    bot = z[r1][:, c0] * (1 - fc) + z[r1][:, c1] * fc
    return top * (1 - fr) + bot * fr


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Uniform noise drawn on a coarse grid (`downsampled_scale` metres per sample), bilinearly up-sampled, rounded, ADDED."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo = int(min_height / terrain.vertical_scale)
    hi = int(max_height / terrain.vertical_scale)
    st = int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    coarse = np.random.choice(levels, (int(terrain.width * terrain.horizontal_scale / downsampled_scale),
                                       int(terrain.length * terrain.horizontal_scale / downsampled_scale)))
    fine = np.rint(_bilinear_resample(coarse, terrain.width, terrain.length))
    terrain.height_field_raw += fine.astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """Pyramid of the given slope (negative: a pit) with a flat platform of `platform_size` metres at the centre."""
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    rx = ((cx - np.abs(cx - np.arange(terrain.width))) / cx).reshape(terrain.width, 1)
    ry = ((cy - np.abs(cy - np.arange(terrain.length))) / cy).reshape(1, terrain.length)
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (peak * rx * ry).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = terrain.width // 2 - half, terrain.length // 2 - half
    edge = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps rising (or, step_height < 0, descending) towards a central platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    h, x0, x1, y0, y1 = 0, 0, terrain.width, 0, terrain.length
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw
        x1 -= sw
        y0 += sw
        y1 -= sw
        h += sh
        terrain.height_field_raw[x0:x1, y0:y1] = h
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """`num_rects` random rectangles of height in {-h, -h/2, h/2, h}; flat platform in the middle."""
    mh = int(max_height / terrain.vertical_scale)
    smin = int(min_size / terrain.horizontal_scale)
    smax = int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    ni, nj = terrain.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(smin, smax, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes)
        ln = np.random.choice(sizes)
        i0 = np.random.choice(range(0, ni - w, 4))
        j0 = np.random.choice(range(0, nj - ln, 4))
        terrain.height_field_raw[i0:i0 + w, j0:j0 + ln] = np.random.choice(heights)
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """Square stones of random height over a pit of `depth` metres; flat platform in the middle."""
    ss = int(stone_size / terrain.horizontal_scale)
    sd = int(stone_distance / terrain.horizontal_scale)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    heights = np.arange(-mh - 1, mh, step=1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        y0 = 0
        while y0 < terrain.length:
            y1 = min(terrain.length, y0 + ss)
            x0 = np.random.randint(0, ss)
            terrain.height_field_raw[0:max(0, x0 - sd), y0:y1] = np.random.choice(heights)      # the cut stone at the border
            while x0 < terrain.width:
                x1 = min(terrain.width, x0 + ss)
                terrain.height_field_raw[x0:x1, y0:y1] = np.random.choice(heights)
                x0 += ss + sd
            y0 += ss + sd
    else:
        x0 = 0
        while x0 < terrain.width:
            x1 = min(terrain.width, x0 + ss)
            y0 = np.random.randint(0, ss)
            terrain.height_field_raw[x0:x1, 0:max(0, y0 - sd)] = np.random.choice(heights)
            while y0 < terrain.length:
                y1 = min(terrain.length, y0 + ss)
                terrain.height_field_raw[x0:x1, y0:y1] = np.random.choice(heights)
                y0 += ss + sd
            x0 += ss + sd
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Vertices [rows*cols, 3] f32 and triangles [2 (rows-1)(cols-1), 3] u32 of the grid, every cell split along its
    (i, j)-(i+1, j+1) diagonal -- the triangulation the contact kernel evaluates analytically from the height grid.
    With `slope_threshold`, vertices next to steps steeper than it are shifted by one cell so that the step becomes a
    vertical wall (the contact kernel works on the un-shifted grid: DESIGN.md section 3)."""
    hf = height_field_raw
    rows, cols = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (cols - 1) * horizontal_scale, cols), np.linspace(0, (rows - 1) * horizontal_scale, rows))
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        hf64 = hf.astype(np.int64)
        mx, my, mc = np.zeros((rows, cols)), np.zeros((rows, cols)), np.zeros((rows, cols))
        mx[:rows - 1, :] += hf64[1:, :] - hf64[:rows - 1, :] > thr
        mx[1:, :] -= hf64[:rows - 1, :] - hf64[1:, :] > thr
        my[:, :cols - 1] += hf64[:, 1:] - hf64[:, :cols - 1] > thr
        my[:, 1:] -= hf64[:, :cols - 1] - hf64[:, 1:] > thr
        mc[:rows - 1, :cols - 1] += hf64[1:, 1:] - hf64[:rows - 1, :cols - 1] > thr
        mc[1:, 1:] -= hf64[:rows - 1, :cols - 1] - hf64[1:, 1:] > thr
        xx += (mx + mc * (mx == 0)) * horizontal_scale
        yy += (my + mc * (my == 0)) * horizontal_scale
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0], vertices[:, 1], vertices[:, 2] = xx.flatten(), yy.flatten(), hf.flatten() * vertical_scale
    i0 = (np.arange(rows - 1)[:, None] * cols + np.arange(cols - 1)[None, :]).reshape(-1)
    tri = np.empty((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    tri[0::2] = np.stack([i0, i0 + cols + 1, i0 + 1], 1)
    tri[1::2] = np.stack([i0, i0 + cols, i0 + cols + 1], 1)
    return vertices, tri
