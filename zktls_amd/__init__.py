"""zktls_amd -- MI355X (gfx950) shard-prove hot path for zkTLS behind a C ABI (libzkhip.so).

The package holds only what the hot path needs: csrc/ (HIP kernels + C ABI + host
orchestration), the ctypes binding (_lib, device) and the host-side mirror of the
reference's prover plug point (prover).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import P, Params, ZkHipError, device_count, from_monty, to_monty  # noqa: F401

__all__ = ["P", "Params", "ZkHipError", "device_count", "from_monty", "to_monty"]
