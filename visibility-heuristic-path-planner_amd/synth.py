"""Seeded synthetic inputs for tests and bench.py (pure numpy; no reference code, no oracle).

The random-rectangle recipe is the reference generator's (src/environment.cpp:57-78)
driven by a local 64-bit LCG instead of glibc rand(), so the maps are identical
on every machine.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class Lcg:
    """Knuth MMIX LCG; deterministic across platforms."""

    def __init__(self, seed):
        self.s = (int(seed) * 2862933555777941757 + 3037000493) & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self.s >> 33

    def below(self, n):
        return self.next() % n


def random_rect_map(nx, ny, nb, min_w, max_w, min_h, max_h, seed):
    """occ[y, x] uint8, 1 = free.  Same clamping and half-open fill as the reference."""
    rng = Lcg(seed)
    occ = np.ones((ny, nx), np.uint8)
    for _ in range(nb):
        c1 = 1 + rng.below(nx + 1)
        c2 = c1 + min_w + rng.below(max_w - min_w + 1)
        c1, c2 = min(c1, nx - 1), min(c2, nx - 1)
        r1 = 1 + rng.below(ny + 1)
        r2 = r1 + min_h + rng.below(max_h - min_h + 1)
        r1, r2 = min(r1, ny - 1), min(r2, ny - 1)
        occ[r1:r2, c1:c2] = 0
    return occ


def free_sources(occ, n, seed):
    """n distinct-ish seeded sources on free cells, int32 [n, 2] as (x, y)."""
    rng = Lcg(seed)
    ny, nx = occ.shape
    out = []
    while len(out) < n:
        x, y = rng.below(nx), rng.below(ny)
        if occ[y, x]:
            out.append((x, y))
    return np.array(out, np.int32).reshape(n, 2)


def config_c3(n_sources=256):
    """BASELINE config 3: 1000x1000, 50 obstacles (20..100 wide/high), seeded sources."""
    occ = random_rect_map(1000, 1000, 50, 20, 100, 20, 100, seed=1)
    return occ, free_sources(occ, n_sources, seed=7)


def config_c5(n_sources=128):
    """BASELINE config 5 (per-GPU share): 4096x4096, 50 obstacles scaled x4."""
    occ = random_rect_map(4096, 4096, 50, 80, 400, 80, 400, seed=1)
    return occ, free_sources(occ, n_sources, seed=11)


def c1_rnd1_mask():
    """BASELINE config 1: the 101x101 mask derived from MATLAB_code/rnd_1.mat (tests/golden/make_fixtures.py)."""
    z = np.load(os.path.join(GOLDEN, "c1_rnd1_mask.npz"))
    nx, ny = int(z["nx"]), int(z["ny"])
    return np.unpackbits(z["packed"], axis=1)[:, :nx].astype(np.uint8).reshape(ny, nx)


def maze_6():
    z = np.load(os.path.join(GOLDEN, "maze_6.npz"))
    nx, ny = int(z["nx"]), int(z["ny"])
    return np.unpackbits(z["packed"], axis=1)[:, :nx].astype(np.uint8).reshape(ny, nx)


def samples_1000():
    """The reference's published 1000 x 1000 sample outputs (Samples/SFML{standAlone,rayCasting}Visibility.png) as
    committed by tests/golden/make_fixtures.py.  Returns dict(sweep_rgb / ray_rgb: the images' own pixels [row, x, 3];
    occ [y, x] uint8 (blocked = the pure-red pixels; field row y = 999 - image row; row y = 0 is never drawn and taken
    as free -- nothing reads it, quadrants march away from the source); source (x, y); ball_radius; sweep_grey / ray_grey
    uint8 [y, x] = uint8(255 v); comparable = the cells whose grey level the images show)."""
    z = np.load(os.path.join(GOLDEN, "samples_1000.npz"))
    sweep, ray = z["sweep_rgb"], z["ray_rgb"]
    red = (sweep[..., 0] == 255) & (sweep[..., 1] == 0) & (sweep[..., 2] == 0)
    occ = np.ascontiguousarray((~red)[::-1]).astype(np.uint8)
    ny, nx = occ.shape
    sx, sy = (int(v) for v in z["source"])
    r = int(z["ball_radius"])
    yy, xx = np.mgrid[0:ny, 0:nx]
    comparable = (yy >= 1) & (occ == 1) & ((xx - sx) ** 2 + (yy - sy) ** 2 > (r + 1) ** 2)
    return dict(occ=occ, source=(sx, sy), ball_radius=r, sweep_rgb=sweep, ray_rgb=ray,
                sweep_grey=np.ascontiguousarray(sweep[::-1, :, 2]), ray_grey=np.ascontiguousarray(ray[::-1, :, 2]),
                comparable=comparable)
