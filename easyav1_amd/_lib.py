"""ctypes binding of libeasyav1_amd.so (the C ABI declared in include/*.h).

There is no CPU fallback: if the shared object is missing or does not load, importing the
compute wrappers raises.  `load(build=True)` compiles it in-tree first (hipcc cross-compiles).
"""
import ctypes
import os

from . import build as _build

_LIB = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i32p = ctypes.POINTER(ctypes.c_int)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_vp = ctypes.c_void_p


class Ea1Error(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().ea1_last_error().decode("utf-8", "replace")
        super().__init__(f"{where} failed with code {code}: {msg}")


def _declare_audio(L):
    L.ea1_last_error.restype = ctypes.c_char_p
    L.ea1_last_error.argtypes = []
    L.ea1_device_count.restype = ctypes.c_int
    L.ea1_vorbis_synth_create.argtypes = [ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ea1_vorbis_synth_destroy.argtypes = [c_vp]
    L.ea1_vorbis_synth_destroy.restype = None
    L.ea1_vorbis_synth_restart.argtypes = [c_vp]
    L.ea1_vorbis_synth_decode.argtypes = [c_vp, c_vp, c_u8p, ctypes.c_int, c_vp, ctypes.c_size_t, c_i32p, c_i64p, c_vp]
    L.ea1_vorbis_synth_count.argtypes = [c_vp, c_u8p, ctypes.c_int]
    L.ea1_vorbis_synth_count.restype = ctypes.c_int64
    L.ea1_mdct_backward_batch.argtypes = [ctypes.c_int, c_vp, ctypes.c_size_t, c_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, c_vp]
    L.ea1_audio_ring_create.argtypes = [ctypes.POINTER(c_vp), ctypes.c_int, ctypes.c_int]
    L.ea1_audio_ring_destroy.argtypes = [c_vp]
    L.ea1_audio_ring_destroy.restype = None
    L.ea1_audio_ring_begin_packet.argtypes = [c_vp]
    L.ea1_audio_ring_begin_packet.restype = None
    L.ea1_audio_ring_push.argtypes = [c_vp, c_f32p, ctypes.c_size_t, ctypes.c_int]
    L.ea1_audio_ring_get.argtypes = [c_vp, ctypes.POINTER(c_f32p), ctypes.POINTER(ctypes.c_size_t)]
    L.ea1_audio_ring_get.restype = ctypes.c_uint
    L.ea1_audio_ring_is_filled.argtypes = [c_vp]


def library_path():
    return _build.LIB


def load(build=False):
    """Load (optionally build first) the shared object; raises if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if build:
        _build.build_library()
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    _declare_audio(L)
    try:
        from . import _lib_video
        _lib_video.declare(L)
    except ImportError:
        pass
    _LIB = L
    return L


def lib():
    return load()


_SYNTH = None


def synth_lib():
    """tests/native/libea1_synth.so: the seeded parsed-frame generator (tests/native/synth_stream.cpp) -- TEST / BENCH INPUT, built beside the tests and
    not part of libeasyav1_amd.so (round-4 verdict).  The harness modules (ir.py, grain.py) take their synthetic frames from here."""
    global _SYNTH
    if _SYNTH is None:
        load()   # (the generator reports errors through the library's ea1_last_error)
        path = _build.SYNTH_LIB
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (the synthetic-stream generator is test infrastructure)")
        _SYNTH = ctypes.CDLL(path)
    return _SYNTH


def last_error():
    """text of the calling thread's last error (ea1_last_error)"""
    L = lib()
    L.ea1_last_error.restype = ctypes.c_char_p
    return (L.ea1_last_error() or b"").decode(errors="replace")


def check(rc, where):
    if rc < 0:
        raise Ea1Error(rc, where)
    return rc
