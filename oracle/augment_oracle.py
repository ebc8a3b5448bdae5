"""TEST INFRASTRUCTURE — numpy restatement of the reference's training-time input transforms
(utils/torch_load.py:405-426 with data_augmentation=['flip','crop'], transformer='simple'):
RandomHorizontalFlip -> RandomCrop(size, padding=size//8, padding_mode='edge') -> ToTensor (uint8 HWC -> float CHW / 255),
with the random decisions passed in.  torchvision 0.x semantics (torchvision.transforms.functional.hflip / pad(mode='edge')
/ crop / to_tensor).  torchvision is NOT installed in this image; the restatement is pinned (tests/test_augment_oracle.py)
to the torch calls torchvision's tensor code path is made of - x.flip(-1), torch.nn.functional.pad(mode='replicate'),
slicing, .to(float32).div(255) - bit for bit on random uint8 batches.
"""
import numpy as np


def augment(images_u8_nhwc, flip, dy, dx, pad):
    """images (N,H,W,C) uint8; flip (N,) bool; dy, dx (N,) ints in [0, 2*pad] -> (N,C,H,W) float32."""
    N, H, W, C = images_u8_nhwc.shape
    out = np.empty((N, C, H, W), dtype=np.float32)
    for n in range(N):
        img = images_u8_nhwc[n]
        if flip is not None and flip[n]:
            img = img[:, ::-1, :]                                     # hflip
        if pad:
            img = np.pad(img, ((pad, pad), (pad, pad), (0, 0)), mode='edge')
            oy = int(dy[n]) if dy is not None else pad
            ox = int(dx[n]) if dx is not None else pad
            img = img[oy:oy + H, ox:ox + W, :]
        out[n] = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)   # to_tensor
    return out
