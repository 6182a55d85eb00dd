"""Point the Python binding at the A/B build of the library (zktls_amd/libzkhip_ab.so, `make -C zktls_amd/csrc ab`): the
same sources with the timing / tuning hooks compiled in (-DZKHIP_AB_HOOKS: environment knobs ZKHIP_NTT_*, ZKHIP_FRI_*).
Import this module BEFORE anything loads the library.  The shipped libzkhip.so has none of these hooks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import zktls_amd._lib as _L  # noqa: E402

_AB = os.path.join(ROOT, "zktls_amd", "libzkhip_ab.so")
if not os.path.exists(_AB):
    raise ImportError("A/B library missing: run `make -C zktls_amd/csrc ab`")
if _L._LIB is not None:
    raise ImportError("tools/_ab.py must be imported before the library is loaded")
_L.LIB_PATH = _AB
