"""CPU check of the input-pipeline oracle (oracle/augment_oracle.py) against an independent formulation."""
import numpy as np

from oracle.augment_oracle import augment


def test_augment_oracle_against_index_formula():
    rng = np.random.default_rng(0)
    N, H, C, pad = 9, 16, 3, 2
    imgs = rng.integers(0, 256, size=(N, H, H, C), dtype=np.uint8)
    flip = rng.integers(0, 2, size=N).astype(bool)
    dy = rng.integers(0, 2 * pad + 1, size=N)
    dx = rng.integers(0, 2 * pad + 1, size=N)
    out = augment(imgs, flip, dy, dx, pad)
    for n in range(N):
        for y in (0, 3, H - 1):
            for x in (0, 5, H - 1):
                ys = min(max(y + dy[n] - pad, 0), H - 1)
                xs = min(max(x + dx[n] - pad, 0), H - 1)
                if flip[n]:
                    xs = H - 1 - xs
                assert np.array_equal(out[n, :, y, x], imgs[n, ys, xs, :].astype(np.float32) / np.float32(255))
    assert out.min() >= 0 and out.max() <= 1
