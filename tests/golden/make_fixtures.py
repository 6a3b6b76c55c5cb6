"""Generates tests/golden/*.npz from DATA the reference ships (never from its code).

Run in the build container only (needs /root/reference and Pillow):
    python tests/golden/make_fixtures.py

maze_6.npz : images/maze_6.png reduced exactly as the reference's loadImage does
             (src/environment.cpp:195-203: free <=> red channel == 255; field(x,y) =
             pixel(x,y), y = 0 at the top), bit-packed.  Input of BASELINE config 4.
benchmark_results.npz : Samples/benchmark_results.txt (the reference's only
             published numbers for this path), as a float array, for BASELINE.md checks.
c1_rnd1_mask.npz : BASELINE config 1's 101x101 occupancy mask.  MATLAB_code/rnd_1.mat is DATA: a
             MATLAB `rng` struct (Type 'twister', 625 x uint32 = the MT19937 state + position).
             numpy's RandomState accepts that state, and random_sample() is the same 53-bit
             construction as MATLAB's rand.  The obstacle recipe restated below is that of
             MATLAB_code/f_comparison_to_a_star.m:54-78 (25 rectangles, sides 2..20, start/end
             kept free), with randi(n) taken as floor(n*rand)+1 -- MATLAB itself is not
             available here, so whether its randi draws exactly like that is unverified: the
             committed MASK is the fixture, not the generator.
samples_1000.npz : the reference's own published outputs of this path, Samples/SFMLstandAloneVisibility.png and
             Samples/SFMLrayCastingVisibility.png (README.md:27-33): two 1000 x 1000 images written by the real
             reference's saveStandAloneVisibility / saveRayCastingVisibility (src/visibilityBasedSolver.cpp:898-955,
             960-1017) from one map and the source (500, 500).  Decoded to DATA: image row r <-> field y = 999 - r
             (row y = 0 is never drawn, :903); blocked cells are exactly the pure-red pixels (:943-949; the same
             set in both files); grey level = uint8(255 * v) (:906-908); the yellow disc of radius 15 (+ black ring
             of radius 16, :918-941) hides the cells around the source.  Stored: the RGB pixels of both images as they
             are (image orientation), the source, the ball radius; synth.samples_1000() derives the occupancy and
             the grey planes.  Cells whose value the images show = y >= 1, not blocked, farther than 16 from the
             source (951 360 of them).
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    img = np.array(Image.open(os.path.join(REF, "images", "maze_6.png")))
    occ = (img[..., 0] == 255).astype(np.uint8)  # [y, x], 1 = free
    ny, nx = occ.shape
    np.savez_compressed(os.path.join(OUT, "maze_6.npz"), packed=np.packbits(occ, axis=1), nx=nx, ny=ny)
    rows = np.loadtxt(os.path.join(REF, "Samples", "benchmark_results.txt"))
    np.savez_compressed(os.path.join(OUT, "benchmark_results.npz"), rows=rows.astype(np.float32))
    print("maze_6", occ.shape, "free cells", int(occ.sum()), "| benchmark rows", rows.shape)
    c1 = c1_mask_from_rnd1()
    np.savez_compressed(os.path.join(OUT, "c1_rnd1_mask.npz"), packed=np.packbits(c1, axis=1), nx=c1.shape[1], ny=c1.shape[0])
    print("c1_rnd1_mask", c1.shape, "free cells", int(c1.sum()))
    samples_1000()


def samples_1000():
    def pixels(name):
        im = np.array(Image.open(os.path.join(REF, "Samples", name)))
        assert im.shape == (1000, 1000, 4) and (im[..., 3] == 255).all()
        return im[..., :3]                # image orientation: row r is field row y = 999 - r
    sweep, ray = pixels("SFMLstandAloneVisibility.png"), pixels("SFMLrayCastingVisibility.png")
    is_red = lambda im: (im[..., 0] == 255) & (im[..., 1] == 0) & (im[..., 2] == 0)
    is_yellow = lambda im: (im[..., 0] == 255) & (im[..., 1] == 255) & (im[..., 2] == 0)
    assert (is_red(sweep) == is_red(ray)).all() and (is_yellow(sweep) == is_yellow(ray)).all()
    rows, xs = np.nonzero(is_yellow(sweep))
    src = (int(round(xs.mean())), 999 - int(round(rows.mean())))
    assert src == (500, 500) and len(xs) == 709          # a whole disc of radius 15: nothing blocked under it
    for im in (sweep, ray):                               # everything that is neither red nor yellow is a grey level
        rest = ~is_red(im) & ~is_yellow(im)
        assert (im[..., 0][rest] == im[..., 1][rest]).all() and (im[..., 1][rest] == im[..., 2][rest]).all()
        assert not im[999].any()                          # field row y = 0 is never drawn (:903): the constructor's black
    np.savez_compressed(os.path.join(OUT, "samples_1000.npz"), sweep_rgb=sweep, ray_rgb=ray,
                        source=np.array(src, np.int32), ball_radius=15)
    print("samples_1000: blocked", int(is_red(sweep).sum()), "source", src)


def c1_mask_from_rnd1(n=101):
    """occ[y, x] uint8 (1 = free) on an n x n grid from the twister state in rnd_1.mat."""
    import scipy.io
    st = scipy.io.loadmat(os.path.join(REF, "MATLAB_code", "rnd_1.mat"), squeeze_me=True, struct_as_record=False)["rngstate"]
    assert st.Type == "twister" and st.State.shape == (625,)
    rs = np.random.RandomState()
    rs.set_state(("MT19937", st.State[:624].astype(np.uint32), int(st.State[624])))
    randi = lambda k: int(np.floor(k * rs.random_sample())) + 1  # MATLAB randi(k): uniform on 1..k
    nx = ny = n
    sp, ep = (5, 5), (95, 95)                         # f_comparison_to_a_star.m:30-31 (1-based row, col)
    obstacle = np.zeros((nx + 1, ny + 1), bool)        # 1-based indexing, row/col 0 unused
    min_w, max_w, min_h, max_h = 2, 20, 2, 20          # :57-58
    for _ in range(25):                                # :56,60
        row_1 = 1 + randi(nx)
        row_2 = row_1 + min_w + randi(max_w - min_w)
        row_1, row_2 = min(row_1, nx - 1), min(row_2, nx - 1)
        col_1 = 1 + randi(ny)
        col_2 = col_1 + min_h + randi(max_h - min_h)
        col_1, col_2 = min(col_1, ny - 1), min(col_2, ny - 1)
        c1 = row_1 <= sp[0] <= row_2
        c2 = col_1 <= sp[1] <= col_2
        c3 = row_1 <= ep[0] <= row_2
        c4 = col_1 <= ep[1] <= col_2
        if not ((c1 and c2) or (c3 and c4)):
            obstacle[row_1:row_2 + 1, col_1:col_2 + 1] = True   # MATLAB ranges are inclusive
    # MATLAB (row, col) 1-based -> field (y, x) 0-based
    return (~obstacle[1:, 1:]).astype(np.uint8)


if __name__ == "__main__":
    main()
