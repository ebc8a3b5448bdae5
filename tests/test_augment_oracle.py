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


def test_augment_oracle_against_torch_functional_ops():
    """Pin to a THIRD-PARTY implementation: torchvision is not installed, but its tensor code path for this pipeline is
    made of exactly these torch calls (torchvision/transforms/_functional_tensor.py: hflip = x.flip(-1); pad(mode='edge')
    = torch.nn.functional.pad(x.float(), mode='replicate'); crop = slicing; to_tensor = x.to(float32).div(255)),
    applied in the reference's order flip -> crop -> ToTensor (utils/torch_load.py:405-426).  Bit-exact on random uint8."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    for (N, H, W, C, pad) in ((7, 32, 32, 3, 4), (5, 16, 16, 1, 2), (3, 8, 12, 3, 1), (4, 28, 28, 1, 0)):
        imgs = rng.integers(0, 256, size=(N, H, W, C), dtype=np.uint8)
        flip = rng.integers(0, 2, size=N).astype(bool)
        dy = rng.integers(0, 2 * pad + 1, size=N)
        dx = rng.integers(0, 2 * pad + 1, size=N)
        out = augment(imgs, flip, dy, dx, pad)
        for n in range(N):
            t = torch.from_numpy(imgs[n]).permute(2, 0, 1)                 # CHW uint8
            if flip[n]:
                t = t.flip(-1)
            if pad:
                t = F.pad(t.float().unsqueeze(0), (pad, pad, pad, pad), mode='replicate').squeeze(0)
                t = t[:, dy[n]:dy[n] + H, dx[n]:dx[n] + W]
            ref = t.to(torch.float32).div(255)
            assert np.array_equal(out[n], ref.numpy()), (N, H, W, C, pad, n)
