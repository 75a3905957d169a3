"""Inputs of the masking_t / alpha_masking_t comparison (tests/test_mask_pinned.py, tests/golden/make_mask_golden.py)."""
import numpy as np


def cases():
    rng = np.random.default_rng(20261005)
    out = []
    for k, (w, h, nch, deg) in enumerate([(24, 12, 2, 1), (24, 12, 4, 1), (40, 17, 4, 3), (16, 9, 2, 3), (33, 20, 4, 2)]):
        core = rng.random((h, w, nch), dtype=np.float32)
        core[..., nch - 1] = (rng.random((h, w)) > 0.3).astype(np.float32) * rng.random((h, w), dtype=np.float32)
        crd = np.stack([rng.uniform(-1.0, w, 200), rng.uniform(-1.0, h, 200)], 1).astype(np.float32)
        out.append((f"case{k}", core, deg, crd))
    return out
