"""WebM reader (easyav1_amd/csrc/webm_reader.cpp) against the reference's demuxer (nestegg compiled
where it lies into oracle/_ref/libref_nestegg.so) on synthetic files from tests/webm_writer.py, and
the audio path end to end: WebM -> packets -> Vorbis front end -> GPU synthesis -> PCM."""
import ctypes
import os

import numpy as np
import pytest

from tests import vorbis_stream_gen as vgen
from tests import webm_writer as ww

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def nestegg():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_nestegg.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_nestegg.so not built")
    L = ctypes.CDLL(path)
    vp, u = ctypes.c_void_p, ctypes.c_uint
    L.refn_open.argtypes = [vp, ctypes.c_long]
    L.refn_open.restype = vp
    L.refn_close.argtypes = [vp]
    for f in ("refn_track_count",):
        getattr(L, f).argtypes = [vp]
    for f in ("refn_track_type", "refn_track_codec", "refn_codec_data_count"):
        getattr(L, f).argtypes = [vp, u]
    L.refn_codec_data.argtypes = [vp, u, u, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    L.refn_video_params.argtypes = [vp, u, ctypes.POINTER(u), ctypes.POINTER(u)]
    L.refn_audio_params.argtypes = [vp, u, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u)]
    L.refn_default_duration.argtypes = [vp, u, ctypes.POINTER(ctypes.c_uint64)]
    L.refn_duration.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.refn_tstamp_scale.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.refn_read_packet.argtypes = [vp, ctypes.POINTER(u), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                   ctypes.POINTER(ctypes.c_int), ctypes.POINTER(u)]
    L.refn_packet_data.argtypes = [vp, u, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    return L


def reference_view(data):
    """everything easyav1 asks nestegg about a file -> plain python structures"""
    L = nestegg()
    buf = ctypes.create_string_buffer(bytes(data), len(data))
    h = L.refn_open(ctypes.addressof(buf), len(data))
    if not h:
        return None
    out = {"tracks": [], "packets": []}
    v64 = ctypes.c_uint64()
    out["duration"] = v64.value if L.refn_duration(h, ctypes.byref(v64)) == 0 else None
    L.refn_tstamp_scale(h, ctypes.byref(v64))
    out["scale"] = v64.value
    for t in range(L.refn_track_count(h)):
        tr = {"type": L.refn_track_type(h, t), "codec": L.refn_track_codec(h, t), "private": []}
        for item in range(max(L.refn_codec_data_count(h, t), 0)):
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            assert L.refn_codec_data(h, t, item, ctypes.byref(p), ctypes.byref(n)) == 0
            tr["private"].append(ctypes.string_at(p.value, n.value))
        w, hh, ch, rate = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint(), ctypes.c_double()
        if tr["type"] == 0 and L.refn_video_params(h, t, ctypes.byref(w), ctypes.byref(hh)) == 0:
            tr["size"] = (w.value, hh.value)
        if tr["type"] == 1 and L.refn_audio_params(h, t, ctypes.byref(rate), ctypes.byref(ch)) == 0:
            tr["audio"] = (rate.value, ch.value)
            L.refn_codec_delay.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64)]
            tr["codec_delay"] = v64.value if L.refn_codec_delay(h, t, ctypes.byref(v64)) == 0 else None
        tr["default_duration"] = v64.value if L.refn_default_duration(h, t, ctypes.byref(v64)) == 0 else 0
        out["tracks"].append(tr)
    while True:
        tk, ts, du, kf, cnt = ctypes.c_uint(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int(), ctypes.c_uint()
        rc = L.refn_read_packet(h, ctypes.byref(tk), ctypes.byref(ts), ctypes.byref(du), ctypes.byref(kf), ctypes.byref(cnt))
        if rc != 1:
            out["end"] = rc
            break
        frames = []
        for i in range(cnt.value):
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            assert L.refn_packet_data(h, i, ctypes.byref(p), ctypes.byref(n)) == 0
            frames.append(ctypes.string_at(p.value, n.value))
        out["packets"].append(dict(track=tk.value, tstamp=ts.value, duration=du.value, keyframe=kf.value, frames=frames))
    L.refn_close(h)
    return out


def our_view(data):
    from easyav1_amd import webm, Ea1Error
    try:
        r = webm.WebmReader(data)
    except Ea1Error:
        return None
    out = {"tracks": [], "packets": [], "duration": r.duration(), "scale": r.tstamp_scale()}
    for i, t in enumerate(r.tracks):
        tr = {"type": t.type, "codec": t.codec, "private": [r.codec_data(i, k) for k in range(t.codec_data_count)],
              "default_duration": t.default_duration}
        if t.type == 0:
            tr["size"] = (t.width, t.height)
        if t.type == 1:
            tr["audio"] = (t.rate, t.channels)
            tr["codec_delay"] = t.codec_delay
        out["tracks"].append(tr)
    try:
        while True:
            p = r.read_packet()
            if p is None:
                out["end"] = 0
                break
            out["packets"].append(p)
    except Ea1Error:
        out["end"] = -1
    r.close()
    return out


def synthetic_file(seed, unknown_sizes=False, extras=True, tc_scale=1000000):
    rng = np.random.default_rng(seed)
    hdr = vgen.make_stream(seed, channels=2)
    tracks = [dict(number=1, type=1, codec="V_AV1", private=bytes(rng.integers(0, 256, 20, dtype=np.uint8)), default_duration=33366700,
                   width=640, height=360),
              dict(number=2, type=2, codec="A_VORBIS", private=ww.vorbis_private(list(hdr[:3])), rate=48000.0, channels=2,
                   codec_delay=int(rng.integers(1, 80)) * 1000000 if seed & 1 else 0)]   # CodecDelay (nestegg_audio_params.codec_delay) on every other file
    apk = vgen.random_packets(seed, hdr[3], count=40)
    blocks, tc, ai = [], 0, 0
    for k in range(60):
        if rng.random() < 0.5 and ai < len(apk):
            lacing = int(rng.integers(0, 4))
            n = 1 if lacing == 0 else int(rng.integers(2, 5))
            frames = apk[ai:ai + n]
            if len(frames) < n:
                lacing, n, frames = 0, 1, apk[ai:ai + 1]
            if lacing == 2:                                  # fixed lacing needs equal sizes
                m = min(len(f) for f in frames)
                frames = [f[:m] for f in frames]
            ai += n
            blocks.append(dict(track=2, tc=tc, key=True, frames=frames, lacing=lacing, group=bool(rng.integers(2)),
                               duration=int(rng.integers(1, 50)) if rng.random() < 0.5 else None))
        else:
            blocks.append(dict(track=1, tc=tc, key=k % 10 == 0, frames=[bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))],
                               lacing=0, group=bool(rng.random() < 0.2), duration=None))
        tc += int(rng.integers(0, 40))
    data = ww.make_webm(seed, tracks, blocks, tc_scale=tc_scale, duration=float(tc), unknown_sizes=unknown_sizes, extras=extras)
    return data, hdr, apk


def assert_same_view(a, b):
    assert (a is None) == (b is None)
    if a is None:
        return
    assert a["scale"] == b["scale"] and a["duration"] == b["duration"]
    assert len(a["tracks"]) == len(b["tracks"])
    for x, y in zip(a["tracks"], b["tracks"]):
        if x["codec"] not in (1, 3):   # nestegg exposes codec private data for Vorbis and Opus only
            x, y = dict(x, private=None), dict(y, private=None)
        assert x == y
    assert len(a["packets"]) == len(b["packets"])
    for x, y in zip(a["packets"], b["packets"]):
        assert x == y
    assert a["end"] == b["end"]


@pytest.mark.parametrize("seed,unknown,extras,scale", [(1, False, True, 1000000), (2, True, True, 1000000), (3, False, False, 500000),
                                                        (4, True, False, 1000000), (5, False, True, 2000000), (6, True, True, 1)])
def test_reader_matches_nestegg(seed, unknown, extras, scale):
    data, _, _ = synthetic_file(seed, unknown, extras, scale)
    ref, ours = reference_view(data), our_view(data)
    assert ref is not None and len(ref["packets"]) == 60 and len(ref["tracks"]) == 2
    assert ref["tracks"][1]["codec"] == 1 and len(ref["tracks"][1]["private"]) == 3
    assert_same_view(ref, ours)


def test_truncated_and_foreign_files():
    data, _, _ = synthetic_file(7)
    for cut in list(range(1, 1200, 7)) + list(range(1200, len(data), 211)):
        assert_same_view(reference_view(data[:cut]), our_view(data[:cut]))
    data2, _, _ = synthetic_file(8, unknown_sizes=True)
    for cut in list(range(900, 1000)) + list(range(1000, len(data2), 401)):
        assert_same_view(reference_view(data2[:cut]), our_view(data2[:cut]))
    assert our_view(b"OggS" + bytes(100)) is None and reference_view(b"OggS" + bytes(100)) is None
    not_webm = data.replace(b"webm", b"avi!", 1)
    assert_same_view(reference_view(not_webm), our_view(not_webm))


@pytest.mark.gpu
def test_webm_audio_end_to_end():
    """file -> demux -> Vorbis packets -> GPU synthesis, against nestegg + libvorbis on the same bytes"""
    from easyav1_amd import vorbis, webm
    from tests import test_vorbis_front as tv
    data, hdr, apk = synthetic_file(11)
    r = webm.WebmReader(data)
    at = [i for i, t in enumerate(r.tracks) if t.codec == 1][0]
    headers = [r.codec_data(at, k) for k in range(3)]
    packets = []
    while True:
        p = r.read_packet()
        if p is None:
            break
        if p["track"] == at:
            packets += p["frames"]
    r.close()
    ref = reference_view(data)
    ref_packets = [f for p in ref["packets"] if p["track"] == at for f in p["frames"]]
    assert packets == ref_packets and headers == ref["tracks"][at]["private"]
    want, counts, ch, bs = tv.ref_decode(headers + packets)
    dec = vorbis.VorbisDecoder(headers[0], headers[1], headers[2], device=0)
    got, _ = dec.decode(packets)
    dec.close()
    assert want.shape[1] > 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
