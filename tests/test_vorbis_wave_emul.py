"""The lane program of the Vorbis synthesis kernels (easyav1_amd/csrc/vorbis_synth_wave.h) emulated on the host -- 64 threads for the
64 lanes, a barrier for the wave fence (tests/native/vorbis_wave_emul.cpp) -- against the oracle, bit for bit.  Checks the schedule
(radix passes, skewed LDS layout, bit-reversed hand-over, fused window / overlap-add, carried right halves) without a GPU; the GPU
tests (tests/test_audio_gpu.py) run the same header as device code."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as ol
from tests.test_audio_gpu import assert_same, make_stream, oracle_pcm, spectra_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "vorbis_wave_emul.cpp")
HDR = os.path.join(ROOT, "easyav1_amd", "csrc", "vorbis_synth_wave.h")
LIB = os.path.join(ROOT, "tests", "native", "libea1_vorbis_emul.so")
pytestmark = []   # CPU tests (the module imported above marks its own tests gpu)

META = np.dtype([("out_off", "<i8"), ("count", "<i4"), ("W", "u1"), ("lW", "u1"), ("has_prev", "u1"), ("pad", "u1")])
_f = ctypes.POINTER(ctypes.c_float)


def fptr(a):
    return a.ctypes.data_as(_f)


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-ffp-contract=off", "-I", os.path.dirname(HDR), SRC, "-o", LIB,
                        "-lpthread"], check=True)
    L = ctypes.CDLL(LIB)
    L.ea1emul_mdct_backward.argtypes = [ctypes.c_int, _f, _f, ctypes.c_size_t, _f, ctypes.c_size_t, ctypes.c_int]
    L.ea1emul_decode.argtypes = [ctypes.c_int, ctypes.c_int, _f, _f, _f, _f, _f, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _f, ctypes.c_size_t, _f, _f]
    assert L.ea1emul_block_meta_size() == META.itemsize
    return L


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_transform_alone(emul, n):
    rng = np.random.default_rng(100 + n)
    spec = spectra_cases(n, rng, 12)
    m = ol.OracleMdct(n)
    with np.errstate(all="ignore"):
        want = m.backward(spec)
    got = np.zeros_like(want)
    emul.ea1emul_mdct_backward(n, fptr(m.trig), fptr(spec), n // 2, fptr(got), n, len(spec))
    assert_same(want, got, f"n={n}")


def plan(bs0, bs1, flags, W, has_prev):
    meta = np.zeros(len(flags), META)
    lW, off = W, 0
    bs = (bs0, bs1)
    for k, f in enumerate(flags):
        cnt = bs[lW] // 4 + bs[f] // 4 if has_prev else 0
        meta[k] = (off, cnt, f, lW, int(has_prev), 0)
        off += cnt
        lW, has_prev = f, True
    return meta, off


class EmulSynth:
    def __init__(self, L, ch, bs0, bs1):
        self.L, self.ch, self.bs = L, ch, (bs0, bs1)
        self.m = [ol.OracleMdct(bs0), ol.OracleMdct(bs1)]
        self.win = [ol.oracle_window(bs0), ol.oracle_window(bs1)]
        self.tail = [np.zeros((ch, bs1 // 2), np.float32), np.zeros((ch, bs1 // 2), np.float32)]
        self.cur, self.W, self.has_prev = 0, 0, False

    def decode(self, spectra, flags):
        meta, total = plan(*self.bs, flags, self.W, self.has_prev)
        pcm = np.full((self.ch, max(total, 1)), np.float32(-777.0))
        spectra = np.ascontiguousarray(spectra, np.float32)
        self.L.ea1emul_decode(self.bs[0], self.bs[1], fptr(self.m[0].trig), fptr(self.win[0]), fptr(self.m[1].trig), fptr(self.win[1]), fptr(spectra),
                              meta.ctypes.data, len(flags), self.ch, fptr(pcm), pcm.shape[1], fptr(self.tail[self.cur]), fptr(self.tail[self.cur ^ 1]))
        self.cur ^= 1
        self.W, self.has_prev = flags[-1], True
        return pcm[:, :total], meta["count"].copy()


@pytest.mark.parametrize("ch,bs0,bs1", [(2, 256, 2048), (1, 64, 64), (2, 128, 8192), (2, 512, 512), (1, 64, 1024), (1, 2048, 4096)])
def test_decode_runs(emul, ch, bs0, bs1):
    rng = np.random.default_rng(ch * 1000 + bs1)
    nb = 21  # not a multiple of the run length
    flags = [int(rng.random() < 0.6) for _ in range(nb)]
    flags[:6] = [1, 1, 0, 0, 1, 0]  # all four (lW, W) cases early on
    spectra = make_stream(rng, ch, bs0, bs1, flags)
    want = np.concatenate(oracle_pcm(ch, bs0, bs1, flags, spectra), axis=1)
    syn = EmulSynth(emul, ch, bs0, bs1)
    got, counts = syn.decode(spectra, flags)
    assert got.shape == want.shape
    assert_same(want, got)
    assert counts[0] == 0 and counts.sum() == want.shape[1]


def test_decode_split_calls_and_restart(emul):
    ch, bs0, bs1 = 2, 256, 2048
    rng = np.random.default_rng(5)
    flags = [1, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 1, 0, 1, 1]
    spectra = make_stream(rng, ch, bs0, bs1, flags)
    restart_at = (7,)
    want = oracle_pcm(ch, bs0, bs1, flags, spectra, restart_at)
    syn = EmulSynth(emul, ch, bs0, bs1)
    outs = []
    for lo, hi in ((0, 3), (3, 4), (4, 7), (7, 15), (15, 19)):
        if lo in restart_at:
            syn.has_prev = False
        pcm, _ = syn.decode(spectra[lo:hi], flags[lo:hi])
        outs.append(pcm)
    assert_same(np.concatenate(want, axis=1), np.concatenate(outs, axis=1))


def test_unaligned_spectra_take_the_scalar_loads(emul):
    n = 512
    rng = np.random.default_rng(9)
    m = ol.OracleMdct(n)
    buf = np.zeros(3 * (n // 2 + 1) + 1, np.float32)
    spec = buf[1:].reshape(3, n // 2 + 1)[:, : n // 2]   # rows 4 bytes off a 16-byte boundary, odd stride
    spec[...] = rng.laplace(scale=20.0, size=spec.shape)
    want = m.backward(np.ascontiguousarray(spec))
    got = np.zeros_like(want)
    emul.ea1emul_mdct_backward(n, fptr(m.trig), ctypes.cast(buf.ctypes.data + 4, _f), n // 2 + 1, fptr(got), n, 3)
    assert_same(want, got)
