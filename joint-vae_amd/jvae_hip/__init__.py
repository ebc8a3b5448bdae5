"""jvae_hip — Python side of the C-ABI extension libjvae_hip.so (MI355X / gfx950 HIP kernels)."""
from .lib import JvaeHipError, load, LIB_PATH, exported_symbols  # noqa: F401
from . import ops  # noqa: F401
