"""TEST INFRASTRUCTURE — deterministic weights / inputs shared by the golden generator and the tests.

Nothing under oracle/ is product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.

Every tensor is a pure function of (key, shape, seed) so that the reference (in the build container), the
CPU oracle and the HIP path can all regenerate identical parameters without shipping megabytes of weights.
"""
import zlib
import numpy as np
import torch


def _rng(key, seed):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def det_tensor(key, shape, seed=0):
    """Deterministic fp32 tensor for state_dict entry `key` (layout-agnostic: depends on key/shape only)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _rng(key, seed).uniform(-1.0, 1.0, size=n).astype(np.float64)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'running_mean':
        v = 0.1 * u
    elif leaf == 'running_var':
        v = 1.0 + 0.1 * np.abs(u)
    elif leaf == '_var_parameter':
        v = 1.0 + 0.2 * u                      # whitening factor: keep it positive and != 1
        if len(shape) == 3:                    # full: near-identity lower-triangular factor per class
            v = 0.05 * u
            v = v.reshape(shape)
            for c in range(shape[0]):
                v[c] += np.eye(shape[1])
            v = v.reshape(-1)
    elif leaf == 'mean' and len(shape) == 2:   # prior dictionary (C, K)
        v = 0.5 * u
    elif len(shape) >= 2:                      # conv / linear weights: U(-1,1)/sqrt(fan_in) * 1.7
        fan_in = int(np.prod(shape[1:]))
        v = 1.7 * u / np.sqrt(fan_in)
    elif leaf == 'weight':                     # 1-D weight = BatchNorm gamma
        v = 1.0 + 0.1 * u
    else:                                      # biases / BN beta
        v = 0.1 * u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def load_det_state(module, seed=0, skip=('sigma',)):
    """Overwrite every floating parameter/buffer of `module` in place with det_tensor(key, shape, seed)."""
    with torch.no_grad():
        for key, t in module.state_dict().items():
            if not t.dtype.is_floating_point or key in skip:
                continue
            t.copy_(det_tensor(key, t.shape, seed).to(t.device))
    return module


def det_inputs(N, input_shape, C, L, K, seed=1234, uniform_eps=False):
    """x ~ U[0,1) (seed), y ~ randint(C) (seed+1), eps (L+1,N,K) with eps[0]=0 (seed+2)."""
    gx = torch.Generator().manual_seed(seed)
    gy = torch.Generator().manual_seed(seed + 1)
    ge = torch.Generator().manual_seed(seed + 2)
    x = torch.rand((N, *input_shape), generator=gx, dtype=torch.float32)
    y = torch.randint(0, C, (N,), generator=gy, dtype=torch.int64)
    if uniform_eps:
        eps = (torch.rand((L + 1, N, K), generator=ge, dtype=torch.float32) - 0.5) * float(np.sqrt(12))
    else:
        eps = torch.randn((L + 1, N, K), generator=ge, dtype=torch.float32)
    eps[0] = 0
    return x, y, eps
