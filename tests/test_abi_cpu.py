"""CPU-only checks of the C ABI: the library builds, loads and exports every symbol the headers
declare; host-only entry points (the PCM ring) behave like the reference; GPU entry points fail
loudly (no silent CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import easyav1_amd
    return easyav1_amd.load(build=True)


def declared_symbols():
    names = []
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if not f.endswith(".h"):
            continue
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(ea1_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(L):
    syms = declared_symbols()
    assert len(syms) >= 14
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_no_device_is_a_loud_error(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = L.ea1_vorbis_synth_create(ctypes.byref(h), 2, 256, 2048, 0)
    assert rc < 0 and not h.value
    assert b"device" in L.ea1_last_error().lower() or b"hip" in L.ea1_last_error().lower()


def test_bad_arguments(L):
    h = ctypes.c_void_p()
    assert L.ea1_vorbis_synth_create(ctypes.byref(h), 2, 100, 2048, 0) < 0   # not a power of two
    assert L.ea1_vorbis_synth_create(ctypes.byref(h), 2, 2048, 256, 0) < 0   # bs1 < bs0
    assert L.ea1_vorbis_synth_create(ctypes.byref(h), 0, 256, 2048, 0) < 0
    assert L.ea1_vorbis_synth_restart(None) < 0


@pytest.mark.parametrize("interlace", [True, False])
def test_audio_ring_matches_oracle(L, interlace):
    """Random push/get traffic: the product ring and the oracle ring (restated from
    src/easyav1.c:2471-2532) must hand out identical frames."""
    from easyav1_amd.audio import AudioRing
    ch = 3
    rng = np.random.default_rng(11 + interlace)
    ring = AudioRing(ch, interlace)
    O = ol.oracle()
    oh = O.ea1o_ring_new(ch, int(interlace))
    for step in range(400):
        ring.begin_packet()
        O.ea1o_ring_begin_packet(oh)
        for _ in range(rng.integers(1, 3)):
            n = int(rng.choice([0, 128, 576, 1024, 2048, 4096]))
            pcm = rng.standard_normal((ch, max(n, 1))).astype(np.float32)
            ring.push(pcm[:, :n] if n else np.zeros((ch, 0), np.float32))
            O.ea1o_ring_push(oh, ol.fptr(pcm), pcm.shape[1], n)
        if rng.random() < 0.4:
            got = ring.get()
            buf = ctypes.POINTER(ctypes.c_float)()
            nbytes = ctypes.c_size_t()
            cnt = O.ea1o_ring_get(oh, ctypes.byref(buf), ctypes.byref(nbytes))
            if cnt == 0:
                assert got is None
            else:
                if interlace:
                    want = np.ctypeslib.as_array(buf, shape=(cnt, ch))
                    assert nbytes.value == cnt * 4 * ch
                else:
                    want = np.ctypeslib.as_array(buf, shape=(ch, 4096))[:, :cnt]
                    assert nbytes.value == cnt * 4
                assert got is not None and np.array_equal(got, want)
    O.ea1o_ring_free(oh)


def test_pmc_traffic_summary_is_consistent():
    """profiles/pmc_traffic.json (bench.py's roofline.traffic) holds, per kernel, the raw counters and the
    two bounds derived from them; the dominant kernel of the default configuration is there and its
    corrected traffic is at least its algorithmic bytes (128 frames of 1920x1080 luma read + written)"""
    import json
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path))
    entry = [e for e in doc["entries"] if e["workload"] == "1080p" and e["chains_per_launch"] == 128][0]
    for name, k in entry["kernels"].items():
        assert abs(k["traffic_low_bytes"] - (k["fetch_size_raw_bytes"] + k["write_size_bytes"])) <= 2, name
        assert abs(k["traffic_bytes"] - (2 * k["fetch_size_raw_bytes"] + k["write_size_bytes"])) <= 2, name
    dom = entry["kernels"]["cdef_filter_kernel<8>"]
    alg = 2 * 128 * 1920 * 1080
    assert alg <= dom["traffic_bytes"] < 1.2 * alg
