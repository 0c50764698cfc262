"""easyav1_amd -- MI355X-native hot path of easyav1 (AV1 reconstruction + Vorbis synthesis).

The product is libeasyav1_amd.so (hand-written HIP for gfx950 behind the C ABI in include/);
this package is the thin Python host side used by the tests and bench.py: ctypes bindings plus
torch plumbing for device memory, streams and torch.distributed.
"""
from ._lib import Ea1Error, library_path, load  # noqa: F401

__all__ = ["Ea1Error", "library_path", "load"]
