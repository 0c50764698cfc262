"""Pins the audio oracle (oracle/vorbis_synth.c) against the reference.

* `ref`-marked tests run the reference's own minivorbis code (oracle/_ref, built from
  /root/reference by oracle/Makefile) side by side with the restatement -- available in the
  build container only.
* the golden tests use tests/golden/audio_golden.npz, generated from that same reference build
  by tests/golden/make_audio_golden.py, and run everywhere (including the GPU box).
"""
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "audio_golden.npz")

ref = ol.ref_minivorbis()
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (no /root/reference here)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def lcg_spectrum(n, st=12345):
    """The probe spectrum of SURVEY.md Appendix B."""
    out = np.zeros(n // 2, np.float32)
    for i in range(n // 2):
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
        v = (st >> 8) % 2001 - 1000
        out[i] = np.float32(v) / np.float32(1000.0)
    return out


def test_survey_probe_hashes():
    # hashes measured on the reference build (no FMA)
    for n, want in ((256, 0x0A098A87), (2048, 0xF4BA0481)):
        out = ol.OracleMdct(n).backward(lcg_spectrum(n))
        assert ol.fnv1a32(out) == want


@needs_ref
@pytest.mark.parametrize("n", SIZES)
def test_tables_match_reference(n):
    m = ol.OracleMdct(n)
    h = ref.ref_mdct_new(n)
    trig = np.ctypeslib.as_array(ref.ref_mdct_trig(h), shape=(n + n // 4,)).copy()
    brev = np.ctypeslib.as_array(ref.ref_mdct_bitrev(h), shape=(n // 4,)).copy()
    ref.ref_mdct_free(h)
    assert np.array_equal(bits(trig), bits(m.trig))
    assert np.array_equal(brev, m.bitrev)
    idx = SIZES.index(n)
    w = np.ctypeslib.as_array(ref.ref_window(idx), shape=(n // 2,)).copy()
    assert np.array_equal(bits(w), bits(ol.oracle_window(n)))


def special_spectra(n, rng):
    n2 = n // 2
    yield np.zeros(n2, np.float32)
    for k in (0, 1, n2 // 2, n2 - 1):
        e = np.zeros(n2, np.float32)
        e[k] = 1.0
        yield e
    yield np.full(n2, 3.0e38, np.float32) * np.where(np.arange(n2) % 2, -1, 1).astype(np.float32)
    yield (rng.standard_normal(n2) * 1e-40).astype(np.float32)  # denormals
    for _ in range(4):
        yield rng.laplace(scale=20.0, size=n2).astype(np.float32)


@needs_ref
@pytest.mark.parametrize("n", SIZES)
def test_mdct_backward_matches_reference(n):
    rng = np.random.default_rng(n)
    m = ol.OracleMdct(n)
    h = ref.ref_mdct_new(n)
    with np.errstate(all="ignore"):
        for spec in special_spectra(n, rng):
            buf = np.zeros(n, np.float32)
            buf[: n // 2] = spec
            ref.ref_mdct_backward(h, ol.fptr(buf), ol.fptr(buf))  # in place, like :17825
            got = m.backward(spec)
            assert np.array_equal(bits(buf), bits(got))
    ref.ref_mdct_free(h)


def block_sequence(rng, bs0, bs1, channels, nblocks, pattern=None):
    flags = pattern if pattern is not None else [int(rng.random() < 0.6) for _ in range(nblocks)]
    blocks = [rng.standard_normal((channels, (bs0, bs1)[W])).astype(np.float32) for W in flags]
    return list(flags), blocks


def ref_synth_stream(channels, bs0, bs1, flags, blocks, restart_at=()):
    h = ref.ref_synth_new(channels, bs0, bs1)
    outs = []
    buf = np.zeros((channels, bs1), np.float32)
    for k, (W, blk) in enumerate(zip(flags, blocks)):
        if k in restart_at:
            ref.ref_synth_restart(h)
        blk = np.ascontiguousarray(blk)
        assert ref.ref_synth_blockin(h, W, ol.fptr(blk)) == 0
        n = ref.ref_synth_pcmout(h, ol.fptr(buf), bs1)
        assert n >= 0
        outs.append(buf[:, :n].copy())
        ref.ref_synth_read(h, n)
    ref.ref_synth_free(h)
    return outs


@needs_ref
@pytest.mark.parametrize("bs0,bs1,ch", [(256, 2048, 2), (64, 64, 1), (128, 8192, 3), (256, 2048, 6), (512, 512, 2)])
def test_blockin_matches_reference(bs0, bs1, ch):
    rng = np.random.default_rng(bs0 * 31 + bs1)
    # all four (lW, W) cases, first block long and short, plus restarts mid-stream
    for pattern in ([0, 0, 1, 1, 0, 1, 0, 0], [1, 1, 0, 0, 1], None):
        flags, blocks = block_sequence(rng, bs0, bs1, ch, 24, pattern)
        for restart_at in ((), (3,), (1, 4)):
            want = ref_synth_stream(ch, bs0, bs1, flags, blocks, restart_at)
            got = ol.oracle_synth_stream(ch, bs0, bs1, flags, blocks, restart_at)
            assert len(want) == len(got)
            for a, b in zip(want, got):
                assert a.shape == b.shape
                assert np.array_equal(bits(a), bits(b))


def test_first_block_returns_nothing_and_counts():
    rng = np.random.default_rng(7)
    flags, blocks = block_sequence(rng, 256, 2048, 2, 0, [1, 1, 0, 0, 1])
    outs = ol.oracle_synth_stream(2, 256, 2048, flags, blocks)
    assert [o.shape[1] for o in outs] == [0, 1024, 576, 128, 576]


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="golden file missing")
def test_golden_vectors():
    g = np.load(GOLDEN)
    for n in SIZES:
        spec = g[f"mdct_in_{n}"]
        want = g[f"mdct_out_{n}"]
        got = ol.OracleMdct(n).backward(spec)
        assert np.array_equal(bits(want), bits(got)), n
        assert np.array_equal(bits(g[f"window_{n}"]), bits(ol.oracle_window(n)))
    for key in [k for k in g.files if k.startswith("ola_flags_")]:
        tag = key[len("ola_flags_"):]
        ch, bs0, bs1 = (int(v) for v in g[f"ola_cfg_{tag}"])
        flags = [int(v) for v in g[key]]
        spectra = g[f"ola_spec_{tag}"]  # (nblocks, ch, bs1/2) zero padded
        restart = [int(v) for v in g[f"ola_restart_{tag}"]]
        m = {0: ol.OracleMdct(bs0), 1: ol.OracleMdct(bs1)}
        blocks = [m[W].backward(spectra[k, :, : (bs0, bs1)[W] // 2]) for k, W in enumerate(flags)]
        outs = ol.oracle_synth_stream(ch, bs0, bs1, flags, blocks, restart)
        pcm = np.concatenate(outs, axis=1)
        assert np.array_equal(bits(g[f"ola_pcm_{tag}"]), bits(pcm)), tag
