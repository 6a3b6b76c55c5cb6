"""Generates tests/golden/*.npz from DATA the reference ships (never from its code).

Run in the build container only (needs /root/reference and Pillow):
    python tests/golden/make_fixtures.py

maze_6.npz : images/maze_6.png reduced exactly as the reference's loadImage does
             (src/environment.cpp:195-203: free <=> red channel == 255; field(x,y) =
             pixel(x,y), y = 0 at the top), bit-packed.  Input of BASELINE config 4.
benchmark_results.npz : Samples/benchmark_results.txt (the reference's only
             published numbers for this path), as a float array, for BASELINE.md checks.
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


if __name__ == "__main__":
    main()
