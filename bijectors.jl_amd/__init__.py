"""bijectors.jl_amd — MI355X (gfx950) implementation of Bijectors.jl's batched
transform + log-abs-det-Jacobian hot path.

    import bijectors_amd as bj                      # root-level shim for this dotted directory
    b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)      # `@` is Julia's `∘`
    y, logjac = bj.with_logabsdet_jacobian(b, x)    # x: (dim, batch) column-major ROCm tensor

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/bjx.h),
_lib.py (ctypes binding) and interface.py (host mirror of src/interface.jl + src/bijectors/*.jl).
"""
from . import _lib, shard, vector
from .interface import *  # noqa: F401,F403
from .interface import __all__ as _iface_all

__all__ = list(_iface_all) + ["_lib", "shard", "vector"]
__version__ = "0.1.0"
