"""throttlecrab_amd -- MI355X-native batched GCRA rate-limit engine.

Drop-in for ONE hot path of lazureykis/throttlecrab:
`RateLimiter<AdaptiveStore>::rate_limit` (throttlecrab/src/core/rate_limiter.rs:102-250).
The product is libtcgpu.so (HIP kernels behind the C ABI of include/tcgpu.h);
this package is the thin host-side mirror used by tests and the bench.
There is no CPU fallback.
"""
from ._lib import LIB_PATH, load  # noqa: F401
from .engine import BatchResult, Engine, TcError  # noqa: F401

__all__ = ["Engine", "BatchResult", "TcError", "load", "LIB_PATH"]
