"""scmomer_amd: MI355X-native Performer RNA-encoder path of scMomer.

Host side mirrors the reference's module API (model/performer/performer_pytorch.py); all
arithmetic runs in hand-written gfx950 kernels behind the C ABI of libscmomer_hip.so.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing: there is no fallback)

__all__ = ["_lib"]
