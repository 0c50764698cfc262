"""Host-side mirror of the libvorbis synthesis interface easyav1 uses, on top of the C ABI.

Names follow the reference (ext/minivorbis/minivorbis.h): a `VorbisSynth` is the decode half of a
`vorbis_dsp_state`; `decode()` = mdct_backward (:7054) + vorbis_synthesis_blockin (:9195) +
pcmout/read (:9411/:9426) for a batch of packets; `restart()` = vorbis_synthesis_restart (:9159).
torch is only used for device buffers and the current stream.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class VorbisSynth:
    def __init__(self, channels, blocksize0, blocksize1, device=0):
        L = _lib.lib()
        self._h = ctypes.c_void_p()
        self.channels, self.bs = channels, (blocksize0, blocksize1)
        self.device = torch.device("cuda", device)
        check(L.ea1_vorbis_synth_create(ctypes.byref(self._h), channels, blocksize0, blocksize1, device),
              "ea1_vorbis_synth_create")

    def close(self):
        if self._h:
            _lib.lib().ea1_vorbis_synth_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def restart(self):
        check(_lib.lib().ea1_vorbis_synth_restart(self._h), "ea1_vorbis_synth_restart")

    def count(self, flags):
        flags = np.ascontiguousarray(flags, np.uint8)
        return check(_lib.lib().ea1_vorbis_synth_count(self._h, flags.ctypes.data_as(_lib.c_u8p), len(flags)),
                     "ea1_vorbis_synth_count")

    def decode(self, spectra, flags, out=None):
        """spectra: cuda float32 [nblocks, channels, bs1/2]; flags: W per block.

        Returns (pcm [channels, total] cuda tensor view, counts per block)."""
        flags = np.ascontiguousarray(flags, np.uint8)
        nb = len(flags)
        assert spectra.is_cuda and spectra.dtype == torch.float32 and spectra.is_contiguous()
        assert tuple(spectra.shape) == (nb, self.channels, self.bs[1] // 2)
        total = self.count(flags)
        if out is None:
            out = torch.empty((self.channels, max(total, 1)), dtype=torch.float32, device=self.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.stride(1) == 1 and out.shape[0] == self.channels
        counts = np.zeros(nb, np.int32)
        tot = ctypes.c_int64(0)
        check(_lib.lib().ea1_vorbis_synth_decode(
            self._h, ctypes.c_void_p(spectra.data_ptr()), flags.ctypes.data_as(_lib.c_u8p), nb,
            ctypes.c_void_p(out.data_ptr()), out.stride(0), counts.ctypes.data_as(_lib.c_i32p),
            ctypes.byref(tot), _stream_ptr(self.device)), "ea1_vorbis_synth_decode")
        return out[:, : tot.value], counts


def decode_batch(synths, spectra, flags, outs=None):
    """ea1_vorbis_synth_decode_batch: one launch per 16 streams.  synths / spectra / flags: one entry per stream.
    -> list of pcm tensors [channels, total]"""
    n = len(synths)
    fl = [np.ascontiguousarray(f, np.uint8) for f in flags]
    totals_expected = [s.count(f) for s, f in zip(synths, fl)]
    if outs is None:
        outs = [torch.empty((s.channels, max(t, 1)), dtype=torch.float32, device=s.device) for s, t in zip(synths, totals_expected)]
    vp = ctypes.c_void_p
    a_syn = (vp * n)(*[s._h for s in synths])
    a_sp = (vp * n)(*[sp.data_ptr() for sp in spectra])
    a_fl = (vp * n)(*[f.ctypes.data for f in fl])
    a_nb = (ctypes.c_int * n)(*[len(f) for f in fl])
    a_pcm = (vp * n)(*[o.data_ptr() for o in outs])
    a_st = (ctypes.c_size_t * n)(*[o.stride(0) for o in outs])
    tot = (ctypes.c_int64 * n)()
    L = _lib.lib()
    L.ea1_vorbis_synth_decode_batch.argtypes = [ctypes.c_int] + [vp] * 8
    check(L.ea1_vorbis_synth_decode_batch(n, a_syn, a_sp, a_fl, a_nb, a_pcm, a_st, tot, _stream_ptr(synths[0].device)), "ea1_vorbis_synth_decode_batch")
    return [o[:, :tot[k]] for k, o in enumerate(outs)]


def mdct_backward(spectra, n):
    """Batched mdct_backward: cuda float32 [..., n/2] -> [..., n]."""
    assert spectra.is_cuda and spectra.dtype == torch.float32
    lead = spectra.shape[:-1]
    flat = spectra.reshape(-1, n // 2).contiguous()
    out = torch.empty((flat.shape[0], n), dtype=torch.float32, device=spectra.device)
    check(_lib.lib().ea1_mdct_backward_batch(
        n, ctypes.c_void_p(flat.data_ptr()), n // 2, ctypes.c_void_p(out.data_ptr()), n, flat.shape[0],
        spectra.device.index or 0, _stream_ptr(spectra.device)), "ea1_mdct_backward_batch")
    return out.reshape(*lead, n)


class AudioRing:
    """easyav1's 4096-sample PCM hand-off buffer (src/easyav1.c:2471-2532, :3604-3630)."""

    def __init__(self, channels, interlace=True):
        self._h = ctypes.c_void_p()
        self.channels, self.interlace = channels, bool(interlace)
        check(_lib.lib().ea1_audio_ring_create(ctypes.byref(self._h), channels, int(interlace)), "ea1_audio_ring_create")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().ea1_audio_ring_destroy(self._h)
        except Exception:
            pass

    def begin_packet(self):
        _lib.lib().ea1_audio_ring_begin_packet(self._h)

    def push(self, pcm):
        """pcm: host float32 [channels, samples]"""
        pcm = np.ascontiguousarray(pcm, np.float32)
        check(_lib.lib().ea1_audio_ring_push(self._h, pcm.ctypes.data_as(_lib.c_f32p), pcm.shape[1] if pcm.ndim == 2 else 0,
                                            pcm.shape[1]), "ea1_audio_ring_push")

    def get(self):
        """Returns None or a copy of the frame: [samples, channels] if interlaced else [channels, samples]."""
        buf = _lib.c_f32p()
        nbytes = ctypes.c_size_t(0)
        n = _lib.lib().ea1_audio_ring_get(self._h, ctypes.byref(buf), ctypes.byref(nbytes))
        if n == 0:
            return None
        if self.interlace:
            return np.ctypeslib.as_array(buf, shape=(n, self.channels)).copy()
        full = np.ctypeslib.as_array(buf, shape=(self.channels, 4096))
        return full[:, :n].copy()

    def is_filled(self):
        return bool(_lib.lib().ea1_audio_ring_is_filled(self._h))
