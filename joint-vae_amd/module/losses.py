"""Per-sample loss terms on HIP kernels (public surface of the reference's module/losses.py:8-86).

    mse_loss(x_output, x_target, ndim, batch_mean)        reconstruction mean-square over the image dims
    x_loss(y_target, logits, batch_mean)                  cross entropy of the L(+1) sampled logits
    categorical_loss(x_output, x_target, ndim, batch_mean) 256-way pixel cross entropy, summed over the image
"""
import torch

from jvae_hip import ops


def mse_loss(x_output, x_target, ndim=3, batch_mean=True):
    """x_target (N1..Ng, D1..Dt); x_output (L, N1..Ng, D1..Dt) -> (L, N1..Ng) mean squares (or their mean).

    Runs the reconstruction kernel with sigma = 1 on every row of x_output (ops.mse_rows: no padded copy).
    """
    lead = x_output.shape[:x_output.dim() - x_target.dim()]
    L_ = 1
    for s in lead:
        L_ *= s
    batch = x_target.shape[:x_target.dim() - ndim]
    n = 1
    for s in batch:
        n *= s
    D = x_target.numel() // max(n, 1)
    wmse = ops.mse_rows(x_output.reshape(L_, n, D), x_target.reshape(n, D)).reshape(*lead, *batch)
    return wmse.mean() if batch_mean else wmse


def x_loss(y_target, logits, batch_mean=True):
    """Cross entropy between labels y (N1..Ng) and logits (L, N1..Ng, C), averaged over L.

    y_target None: all-class evaluation -> -log(softmax + 1e-6) averaged over the sampled rows, class-major
    (evaluation path)."""
    if y_target is None:
        log_p = (logits.softmax(dim=-1) + 1e-6).log()
        perm = [-1] + list(range(log_p.dim() - 2))
        rows = log_p[1:].mean(0) if log_p.shape[0] > 1 else log_p[0]
        return -rows.permute(perm)
    ce = ops.cross_entropy_rows(logits, y_target.reshape(-1))        # (L, N1..Ng)
    return ce.mean() if batch_mean else ce.mean(0)


def categorical_loss(x_output, x_target, ndim=3, batch_mean=True):
    """256-way pixel cross entropy (reference module/losses.py:30-49): x_target (N1..Ng, D1..Dt) in [0, 1],
    x_output (N1..Ng, 256, D1..Dt) logits with the class axis in front of the image axes (the `Reshape((256, C, h, w))`
    at the end of a categorical decoder, conv.py:228-230); target class = floor(255 x).  Sum over the image, per sample.
    The cross-entropy rows run on the HIP kernel (class axis moved last by a permuted copy)."""
    expanded_shape = (*x_output.shape[:-ndim - 1], *x_target.shape[-ndim:])
    x_target = x_target.expand(*expanded_shape)
    batch_shape = x_target.shape[:-ndim]
    image_shape = x_target.shape[-ndim:]
    pixels = 1
    for s_ in image_shape:
        pixels *= s_
    target = (x_target * 255).long().reshape(-1)
    logits = x_output.reshape(-1, 256, pixels).permute(0, 2, 1).reshape(1, -1, 256)
    ce = ops.cross_entropy_rows(logits, target).reshape(*batch_shape, pixels).sum(-1)
    return ce.mean() if batch_mean else ce
