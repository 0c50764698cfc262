"""ctypes mirror of include/easyav1_amd_vorbis.h (tests / bench harness only)."""
import ctypes

import numpy as np

from . import _lib

_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.ea1_vorbis_decoder_create.argtypes = [ctypes.POINTER(vp), vp, sz, vp, sz, vp, sz, ctypes.c_int]
        L.ea1_vorbis_decoder_destroy.argtypes = [vp]
        L.ea1_vorbis_decoder_destroy.restype = None
        L.ea1_vorbis_decoder_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long),
                                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.ea1_vorbis_decoder_spectra.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_int)]
        L.ea1_vorbis_decoder_decode.argtypes = [vp, vp, vp, ctypes.c_int, vp, sz, vp, ctypes.POINTER(ctypes.c_int64), vp]
        L.ea1_vorbis_decoder_restart.argtypes = [vp]
        L.ea1_vorbis_floor1_db_table.restype = ctypes.POINTER(ctypes.c_float)
        _bound = True
    return L


def floor1_db_table():
    L = _bind()
    return np.ctypeslib.as_array(L.ea1_vorbis_floor1_db_table(), shape=(256,)).copy()


class VorbisDecoder:
    """Vorbis packets -> spectra (host) -> PCM (GPU synthesis).  device=-1: host stage only."""

    def __init__(self, id_header, comment_header, setup_header, device=0):
        L = _bind()
        self._h = ctypes.c_void_p()
        self._hdr = [bytes(id_header), bytes(comment_header), bytes(setup_header)]
        _lib.check(L.ea1_vorbis_decoder_create(ctypes.byref(self._h), self._hdr[0], len(self._hdr[0]), self._hdr[1],
                                               len(self._hdr[1]), self._hdr[2], len(self._hdr[2]), device),
                   "ea1_vorbis_decoder_create")
        ch, rate, b0, b1 = ctypes.c_int(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int()
        _lib.check(L.ea1_vorbis_decoder_info(self._h, ctypes.byref(ch), ctypes.byref(rate), ctypes.byref(b0),
                                             ctypes.byref(b1)), "ea1_vorbis_decoder_info")
        self.channels, self.rate, self.blocksizes = ch.value, rate.value, (b0.value, b1.value)

    def spectra(self, packet):
        """-> (spectra [channels, n/2] float32, long_block flag) or None for an undecodable packet"""
        L = _bind()
        half1 = self.blocksizes[1] // 2
        out = np.zeros((self.channels, half1), np.float32)
        W = ctypes.c_int()
        pkt = bytes(packet)
        if L.ea1_vorbis_decoder_spectra(self._h, pkt, len(pkt), out.ctypes.data, half1, ctypes.byref(W)) != 0:
            return None
        return out[:, :self.blocksizes[W.value] // 2].copy(), W.value

    def decode(self, packets, stream=None):
        """-> (pcm [channels, total] float32, counts per packet)"""
        L = _bind()
        n = len(packets)
        bufs = [bytes(p) for p in packets]
        ptrs = (ctypes.c_char_p * n)(*bufs)
        sizes = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        cap = max(n * self.blocksizes[1] // 2, 1)
        pcm = np.zeros((self.channels, cap), np.float32)
        counts = (ctypes.c_int * n)()
        total = ctypes.c_int64()
        _lib.check(L.ea1_vorbis_decoder_decode(self._h, ptrs, sizes, n, pcm.ctypes.data, cap, counts, ctypes.byref(total),
                                               stream), "ea1_vorbis_decoder_decode")
        return pcm[:, :total.value].copy(), list(counts)

    def restart(self):
        _lib.check(_bind().ea1_vorbis_decoder_restart(self._h), "ea1_vorbis_decoder_restart")

    def close(self):
        if self._h:
            _bind().ea1_vorbis_decoder_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
