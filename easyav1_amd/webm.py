"""ctypes mirror of include/easyav1_amd_webm.h (tests / tools only)."""
import ctypes

from . import _lib

MAX_FRAMES = 256


class Track(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("codec", ctypes.c_int), ("number", ctypes.c_uint64), ("default_duration", ctypes.c_uint64),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("rate", ctypes.c_double), ("channels", ctypes.c_uint32),
                ("codec_data_count", ctypes.c_uint32), ("codec_delay", ctypes.c_uint64)]


class Packet(ctypes.Structure):
    _fields_ = [("track", ctypes.c_uint32), ("tstamp", ctypes.c_uint64), ("duration", ctypes.c_uint64), ("keyframe", ctypes.c_int),
                ("count", ctypes.c_uint32), ("data", ctypes.c_void_p * MAX_FRAMES), ("size", ctypes.c_size_t * MAX_FRAMES)]


def _bind():
    L = _lib.lib()
    if not getattr(L, "_webm_bound", False):
        vp = ctypes.c_void_p
        L.ea1_webm_open_memory.argtypes = [ctypes.POINTER(vp), vp, ctypes.c_size_t]
        L.ea1_webm_destroy.argtypes = [vp]
        L.ea1_webm_destroy.restype = None
        L.ea1_webm_track_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
        L.ea1_webm_track_info.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(Track)]
        L.ea1_webm_codec_data.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        L.ea1_webm_duration.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.ea1_webm_tstamp_scale.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.ea1_webm_read_packet.argtypes = [vp, ctypes.POINTER(Packet)]
        L.ea1_webm_rewind.argtypes = [vp]
        L._webm_bound = True
    return L


class WebmReader:
    def __init__(self, data):
        L = _bind()
        self._buf = ctypes.create_string_buffer(bytes(data), len(data))   # borrowed by the reader: keep it alive
        self._h = ctypes.c_void_p()
        _lib.check(L.ea1_webm_open_memory(ctypes.byref(self._h), ctypes.addressof(self._buf), len(data)), "ea1_webm_open_memory")
        n = ctypes.c_uint32()
        _lib.check(L.ea1_webm_track_count(self._h, ctypes.byref(n)), "ea1_webm_track_count")
        self.tracks = []
        for i in range(n.value):
            t = Track()
            _lib.check(L.ea1_webm_track_info(self._h, i, ctypes.byref(t)), "ea1_webm_track_info")
            self.tracks.append(t)

    def codec_data(self, track, item):
        L = _bind()
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(L.ea1_webm_codec_data(self._h, track, item, ctypes.byref(p), ctypes.byref(n)), "ea1_webm_codec_data")
        return ctypes.string_at(p.value, n.value)

    def duration(self):
        L = _bind()
        v = ctypes.c_uint64()
        return v.value if L.ea1_webm_duration(self._h, ctypes.byref(v)) == 0 else None

    def tstamp_scale(self):
        v = ctypes.c_uint64()
        _lib.check(_bind().ea1_webm_tstamp_scale(self._h, ctypes.byref(v)), "ea1_webm_tstamp_scale")
        return v.value

    def read_packet(self):
        """-> dict or None at the end of the stream; raises Ea1Error on a malformed block"""
        L = _bind()
        pk = Packet()
        rc = L.ea1_webm_read_packet(self._h, ctypes.byref(pk))
        if rc == 0:
            return None
        if rc != 1:
            _lib.check(rc, "ea1_webm_read_packet")
        return dict(track=pk.track, tstamp=pk.tstamp, duration=pk.duration, keyframe=pk.keyframe,
                    frames=[ctypes.string_at(pk.data[i], pk.size[i]) for i in range(pk.count)])

    def rewind(self):
        _lib.check(_bind().ea1_webm_rewind(self._h), "ea1_webm_rewind")

    def close(self):
        if self._h:
            _bind().ea1_webm_destroy(self._h)
            self._h = ctypes.c_void_p()
