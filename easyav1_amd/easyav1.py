"""ctypes mirror of include/easyav1.h over libeasyav1.so (tests / tools harness only): the same 34
entry points and struct layouts an application linking the reference's library sees
(/root/reference/src/easyav1.h:449-822)."""
import ctypes
import os

import numpy as np

from . import _lib
from . import build as _build

STATUS_ERROR, STATUS_OK, STATUS_FINISHED = 0, 1, 2
LOG_ERROR, LOG_WARNING, LOG_INFO = 0, 1, 2
AUDIO_BUFFER_SIZE = 4096


class VideoProperties(ctypes.Structure):
    _fields_ = [("pixel_layout", ctypes.c_int), ("bits_per_color", ctypes.c_int), ("color_space", ctypes.c_int),
                ("color_primaries", ctypes.c_int), ("transfer_characteristics", ctypes.c_int),
                ("matrix_coefficients", ctypes.c_int), ("chroma_sample_position", ctypes.c_int),
                ("width", ctypes.c_uint), ("height", ctypes.c_uint)]


class VideoFrame(ctypes.Structure):
    _fields_ = [("properties", VideoProperties), ("timestamp", ctypes.c_uint64), ("data", ctypes.c_void_p * 3),
                ("stride", ctypes.c_size_t * 3)]


class AudioFrame(ctypes.Structure):
    _fields_ = [("channels", ctypes.c_uint), ("samples", ctypes.c_uint), ("timestamp", ctypes.c_uint64),
                ("bytes", ctypes.c_size_t), ("pcm", ctypes.c_void_p)]


VIDEO_CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(VideoFrame), ctypes.c_void_p)
AUDIO_CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(AudioFrame), ctypes.c_void_p)


class Callbacks(ctypes.Structure):
    _fields_ = [("video", VIDEO_CB), ("audio", AUDIO_CB), ("userdata", ctypes.c_void_p)]


class Settings(ctypes.Structure):
    _fields_ = [("enable_video", ctypes.c_int), ("enable_audio", ctypes.c_int), ("skip_unprocessed_frames", ctypes.c_int),
                ("interlace_audio", ctypes.c_int), ("close_handle_on_destroy", ctypes.c_int), ("callbacks", Callbacks),
                ("video_track", ctypes.c_uint), ("audio_track", ctypes.c_uint), ("use_fast_seeking", ctypes.c_int),
                ("audio_offset_time", ctypes.c_int64), ("log_level", ctypes.c_int)]


READ_FUNC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
SEEK_FUNC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)
TELL_FUNC = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p)


class Stream(ctypes.Structure):
    _fields_ = [("read_func", READ_FUNC), ("seek_func", SEEK_FUNC), ("tell_func", TELL_FUNC), ("userdata", ctypes.c_void_p)]


EXPORTS = ["easyav1_default_settings", "easyav1_init_from_filename", "easyav1_init_from_file", "easyav1_init_from_memory",
           "easyav1_init_from_custom_stream", "easyav1_decode_next", "easyav1_decode_until", "easyav1_decode_for",
           "easyav1_play", "easyav1_stop", "easyav1_seek_forward", "easyav1_seek_backward", "easyav1_seek_to_timestamp",
           "easyav1_get_status", "easyav1_get_current_timestamp", "easyav1_has_video_track", "easyav1_has_audio_track",
           "easyav1_get_total_video_tracks", "easyav1_get_total_audio_tracks", "easyav1_get_video_width",
           "easyav1_get_video_height", "easyav1_get_video_fps", "easyav1_get_audio_channels",
           "easyav1_get_audio_sample_rate", "easyav1_has_video_frame", "easyav1_get_video_frame",
           "easyav1_get_total_video_frames_processed", "easyav1_is_audio_buffer_filled", "easyav1_get_audio_frame",
           "easyav1_get_duration", "easyav1_get_current_settings", "easyav1_update_settings", "easyav1_is_finished",
           "easyav1_destroy"]

_API = None


def library_path():
    return _build.API_LIB


def api():
    """libeasyav1.so (libeasyav1_amd.so is loaded first: same directory, found through $ORIGIN)"""
    global _API
    if _API is not None:
        return _API
    _lib.load()
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = ctypes.CDLL(path)
    vp, u64 = ctypes.c_void_p, ctypes.c_uint64
    sp = ctypes.POINTER(Settings)
    L.easyav1_default_settings.restype = Settings
    L.easyav1_init_from_filename.argtypes, L.easyav1_init_from_filename.restype = [ctypes.c_char_p, sp], vp
    L.easyav1_init_from_file.argtypes, L.easyav1_init_from_file.restype = [vp, sp], vp
    L.easyav1_init_from_memory.argtypes, L.easyav1_init_from_memory.restype = [vp, ctypes.c_size_t, sp], vp
    L.easyav1_init_from_custom_stream.argtypes, L.easyav1_init_from_custom_stream.restype = [ctypes.POINTER(Stream), sp], vp
    for name in ("decode_next", "play", "get_status"):
        getattr(L, "easyav1_" + name).argtypes = [vp]
    for name in ("decode_until", "decode_for", "seek_forward", "seek_backward", "seek_to_timestamp"):
        getattr(L, "easyav1_" + name).argtypes = [vp, u64]
    L.easyav1_stop.argtypes, L.easyav1_stop.restype = [vp], None
    for name in ("get_current_timestamp", "get_duration", "get_total_video_frames_processed"):
        f = getattr(L, "easyav1_" + name)
        f.argtypes, f.restype = [vp], u64
    for name in ("has_video_track", "has_audio_track", "has_video_frame", "is_audio_buffer_filled", "is_finished"):
        getattr(L, "easyav1_" + name).argtypes = [vp]
    for name in ("get_total_video_tracks", "get_total_audio_tracks", "get_video_width", "get_video_height", "get_video_fps",
                 "get_audio_channels", "get_audio_sample_rate"):
        f = getattr(L, "easyav1_" + name)
        f.argtypes, f.restype = [vp], ctypes.c_uint
    L.easyav1_get_video_frame.argtypes, L.easyav1_get_video_frame.restype = [vp], ctypes.POINTER(VideoFrame)
    L.easyav1_get_audio_frame.argtypes, L.easyav1_get_audio_frame.restype = [vp], ctypes.POINTER(AudioFrame)
    L.easyav1_get_current_settings.argtypes, L.easyav1_get_current_settings.restype = [vp], Settings
    L.easyav1_update_settings.argtypes = [vp, sp]
    L.easyav1_destroy.argtypes, L.easyav1_destroy.restype = [ctypes.POINTER(vp)], None
    _API = L
    return L


class Player:
    """An easyav1_t opened from bytes (easyav1_init_from_memory); thin: one method per entry point used by the tests."""

    def __init__(self, data, **settings):
        self.L = api()
        self._data = bytes(data)          # borrowed by the library: kept alive here
        self.settings = self.L.easyav1_default_settings()
        for k, v in settings.items():
            setattr(self.settings, k, v)
        self.h = ctypes.c_void_p(self.L.easyav1_init_from_memory(self._data, len(self._data), ctypes.byref(self.settings)))
        if not self.h:
            raise RuntimeError("easyav1_init_from_memory returned NULL")

    def __getattr__(self, name):
        f = getattr(self.L, "easyav1_" + name)
        return lambda *a: f(self.h, *a)

    def audio_frame(self):
        """-> (timestamp, pcm [channels, samples] float32 copy) or None"""
        p = self.L.easyav1_get_audio_frame(self.h)
        if not p:
            return None
        f = p.contents
        ch, n = f.channels, f.samples
        if self.settings.interlace_audio:
            assert f.bytes == 4 * ch * n
            a = np.ctypeslib.as_array(ctypes.cast(f.pcm, ctypes.POINTER(ctypes.c_float)), shape=(n, ch)).T.copy()
        else:
            assert f.bytes == 4 * n
            planes = ctypes.cast(f.pcm, ctypes.POINTER(ctypes.POINTER(ctypes.c_float)))
            a = np.stack([np.ctypeslib.as_array(planes[c], shape=(n,)).copy() for c in range(ch)])
        return f.timestamp, a

    def video_frame(self):
        """-> (timestamp, [y, u, v] uint8 copies at the visible size, properties) or None"""
        p = self.L.easyav1_get_video_frame(self.h)
        if not p:
            return None
        f = p.contents
        w, h = f.properties.width, f.properties.height
        planes = []
        for pl in range(3):
            # chroma planes by the picture's layout (src/easyav1.h:183-188: 2 = YUV420, 3 = YUV422, 4 = YUV444)
            sx, sy = (0 if f.properties.pixel_layout == 4 else 1), (1 if f.properties.pixel_layout in (1, 2) else 0)
            pw, ph = (w, h) if pl == 0 else ((w + sx) >> sx, (h + sy) >> sy)
            if not f.data[pl]:     # 4:0:0: no chroma planes
                planes.append(None)
                continue
            buf = np.ctypeslib.as_array(ctypes.cast(f.data[pl], ctypes.POINTER(ctypes.c_uint8)), shape=(ph, f.stride[pl]))
            planes.append(buf[:, :pw].copy())
        return f.timestamp, planes, f.properties

    def close(self):
        if self.h:
            self.L.easyav1_destroy(ctypes.byref(self.h))
            assert not self.h.value      # easyav1_destroy NULLs the caller's handle

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
