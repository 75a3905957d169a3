"""Generates tests/golden/alpha_golden.npz: planes softened by the reference's own
zimt::convolve (binomial 1 4 6 4 1 / 16, REFLECT, both axes: the call of
environment.h:833-843), through oracle/_ref/libref_zimt.so. Build container only:

    make -C oracle ref && python tests/golden/make_alpha_golden.py

Data only: input planes and the reference's output planes."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refz  # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    for k, (w, h) in enumerate([(37, 23), (64, 48), (5, 9), (3, 2), (1, 7), (130, 4)]):
        plane = (rng.random((h, w)) > 0.35).astype(np.float32)       # a 0/1 mask, as masks are
        if k % 2:
            plane = rng.random((h, w), dtype=np.float32)             # and arbitrary values
        out[f"in_{k}"] = plane
        out[f"out_{k}"] = refz.binomial_alpha(plane)
    np.savez_compressed(os.path.join(HERE, "alpha_golden.npz"), **out)
    print("wrote alpha_golden.npz:", len(out) // 2, "planes")


if __name__ == "__main__":
    main()
