"""REAL AV1 and Vorbis in ONE WebM file through the public interface (BASELINE configs[0] and configs[4]), and what easyav1_update_settings
does to a playing file (/root/reference/src/easyav1.c:3795-3949).

  video   committed libaom streams (tests/golden/av1_stream_*.npz): every shown picture == dav1d 1.5.3's digest
  audio   generated Vorbis tracks (tests/streamgen.py): the PCM handed out == the reference's own minivorbis compiled from
          /root/reference (oracle/_ref/libref_minivorbis.so), bit for bit
  order   packets of the two tracks are taken as src/easyav1.c:2026-2131 takes them (video first on a timestamp tie)

The settings tests restate the reference's text (its library cannot be built here: no dav1d sources): a change of the audio track / of
interlace_audio / of audio_offset_time / of enable_video ends in a SLOW seek to the current position (:3937-3945) -- the demuxer goes back, the
audio decoder restarts (vorbis_synthesis_restart, :3077; the packets before the position are only tracked, :2447-2457), the first audio packet at or
after the position comes out like the first packet of a stream (no samples), the video is decoded forward from the last key frame and the
packet AT the position is decoded again (:3127 breaks before it is taken)."""
import ctypes
import os

import numpy as np
import pytest

from tests import vorbis_stream_gen as vgen
from tests import test_real_av1 as tr
from tests import webm_writer as ww
from tests.test_easyav1_abi import merge, reference_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ea():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import easyav1
    easyav1.api()
    return easyav1


def video_track(name, repeat=1, frame_ms=40):
    """a committed libaom stream as a V_AV1 track -> (track, blocks, fixture)"""
    import os
    fx = tr.load_fixture(os.path.join(tr.HERE, "golden", "av1_stream_%s.npz" % name))
    units, seq = [], None
    for tu in fx["units"]:
        body = b""
        for typ, off, size in tr.split_obus(tu):
            if typ == 2:
                continue
            if typ == 1 and seq is None:
                seq = tu[off:off + size]
            body += tu[off:off + size]
        units.append(body)
    keys = [any(typ == 1 for typ, off, size in tr.split_obus(tu)) for tu in fx["units"]]
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq, default_duration=frame_ms * 1000000, width=fx["width"], height=fx["height"])
    n = len(units)
    blocks = [dict(track=1, tc=frame_ms * k, key=keys[k % n], frames=[units[k % n]], lacing=0, group=False, duration=None) for k in range(n * repeat)]
    return track, blocks, fx


def audio_track(seed, channels, count, number=2, step_ms=21, bs=(256, 2048), codec_delay=0):
    hdr = vgen.make_stream(seed, channels, bs)
    packets = vgen.random_packets(seed, hdr[3], count=count)
    track = dict(number=number, type=2, codec="A_VORBIS", private=ww.vorbis_private(list(hdr[:3])), rate=48000.0, channels=channels, codec_delay=codec_delay)
    blocks = [dict(track=number, tc=k * step_ms, key=True, frames=[p], lacing=0, group=False, duration=None) for k, p in enumerate(packets)]
    return track, blocks, hdr, packets


def drain(p, ea, fx, shown0=0, until=None):
    """decode_next until the end (or `until` calls): -> (pictures' timestamps, their digests checked against dav1d's starting at shown0, audio frames [(ts, pcm)])"""
    stamps, audio, shown, calls = [], [], shown0, 0
    mono = fx["layout"] == 0
    n = len(fx["shown_sha"])
    while (until is None or calls < until) and p.decode_next() == ea.STATUS_OK:
        calls += 1
        if p.has_video_frame():
            ts, planes, props = p.video_frame()
            assert np.array_equal(tr.sha([x for x in planes if x is not None], mono), fx["shown_sha"][shown % n]), (fx["name"], shown)
            stamps.append(ts)
            shown += 1
        f = p.audio_frame()
        if f is not None:
            audio.append(f)
    return stamps, shown, audio


@pytest.mark.parametrize("name,repeat,channels,packets,interlace", [
    ("objects_640x360", 1, 2, 60, 1),      # BASELINE configs[0]: 640x360 + 48 kHz stereo
    ("tiles_4cols_640x360", 4, 2, 40, 0),
    ("bench_hd_gop16", 3, 6, 110, 0),      # BASELINE configs[4]: 1080p + 48 kHz 5.1; three GOPs, so that several GOP chains are in flight beside the audio look-ahead
    ("hd_1920x1080", 2, 6, 30, 1),
])
def test_real_av1_and_vorbis_in_one_file(ea, name, repeat, channels, packets, interlace, monkeypatch):
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    vt, vb, fx = video_track(name, repeat)
    at, ab, hdr, pk = audio_track(1000 + channels, channels, packets)
    want, counts, ch, _ = reference_pcm(hdr, pk)
    data = ww.make_webm(21, [vt, at], merge(vb, ab), duration=float(max(vb[-1]["tc"] + 40, ab[-1]["tc"] + 21)))
    p = ea.Player(data, interlace_audio=interlace, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    assert p.has_video_track() and p.has_audio_track() and p.get_audio_channels() == channels and p.get_audio_sample_rate() == 48000
    assert (p.get_video_width(), p.get_video_height()) == (fx["width"], fx["height"])
    stamps, shown, audio = drain(p, ea, fx)
    assert p.is_finished() and p.get_status() != ea.STATUS_ERROR
    # video: every shown picture of every repeat (hidden frames and show_existing_frame units included in the stream), in order
    per_pass = len(fx["shown_sha"])
    assert shown == per_pass * repeat and stamps == sorted(stamps) and len(set(stamps)) == len(stamps)
    # audio: the reference build's PCM, bit for bit; a frame's timestamp is its first packet's
    got = np.concatenate([a for _, a in audio], axis=1)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert [t for t, _ in audio] == [21 * k for k, c in enumerate(counts) if c > 0]
    p.close()


def test_both_callbacks_on_a_real_file(ea, monkeypatch):
    """the callback form of the same (settings.callbacks, src/easyav1.c:2586-2616): pictures and PCM arrive in the order the packets are taken"""
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    vt, vb, fx = video_track("slow_640x360", 2)
    at, ab, hdr, pk = audio_track(77, 2, 40)
    want, _, _, _ = reference_pcm(hdr, pk)
    data = ww.make_webm(22, [vt, at], merge(vb, ab), duration=1000.0)
    events, pcm = [], []

    def on_video(frame, _):
        f = frame.contents
        planes = []
        for pl in range(3):
            w, h = (f.properties.width, f.properties.height) if pl == 0 else ((f.properties.width + 1) // 2, (f.properties.height + 1) // 2)
            planes.append(np.ctypeslib.as_array(ctypes.cast(f.data[pl], ctypes.POINTER(ctypes.c_uint8)), shape=(h, f.stride[pl]))[:, :w].copy())
        events.append(("v", f.timestamp, tr.sha(planes)))

    def on_audio(frame, _):
        f = frame.contents
        events.append(("a", f.timestamp, None))
        pcm.append(np.ctypeslib.as_array(ctypes.cast(f.pcm, ctypes.POINTER(ctypes.c_float)), shape=(f.samples, f.channels)).T.copy())
    L = ea.api()
    s = L.easyav1_default_settings()
    s.log_level, s.skip_unprocessed_frames = ea.LOG_ERROR, 0
    vcb, acb = ea.VIDEO_CB(on_video), ea.AUDIO_CB(on_audio)
    s.callbacks.video, s.callbacks.audio = vcb, acb
    h = ctypes.c_void_p(L.easyav1_init_from_memory(data, len(data), ctypes.byref(s)))
    assert h
    while L.easyav1_decode_next(h) == ea.STATUS_OK:
        pass
    assert L.easyav1_is_finished(h)
    L.easyav1_destroy(ctypes.byref(h))
    vs = [e for e in events if e[0] == "v"]
    n = len(fx["shown_sha"])
    assert len(vs) == 2 * n
    for k, (_, ts, digest) in enumerate(vs):
        assert np.array_equal(digest, fx["shown_sha"][k % n]), k
    stamps = [e[1] for e in events]
    assert stamps == sorted(stamps)          # one clock for both tracks
    got = np.concatenate(pcm, axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------------------------------------ easyav1_update_settings (N3)
def two_audio_file(name="objects_640x360", frame_ms=40, nb=70, codec_delay=0):
    vt, vb, fx = video_track(name, 1, frame_ms)
    a1, b1, h1, p1 = audio_track(201, 2, nb, number=2)
    a2, b2, h2, p2 = audio_track(202, 6, nb, number=3, codec_delay=codec_delay)
    data = ww.make_webm(23, [vt, a1, a2], merge(vb, b1, b2), duration=float(max(vb[-1]["tc"] + frame_ms, 21 * nb)))
    return data, fx, (h1, p1), (h2, p2)


def pcm_from(hdr, packets, first):
    """what the reference's decoder hands out when it restarts in front of packets[first] (a fresh decoder: the packets before are tracked only)"""
    want, counts, _, _ = reference_pcm(hdr, packets[first:])
    return want, counts


def update(p, ea, **changes):
    s = p.L.easyav1_get_current_settings(p.h)
    for k, v in changes.items():
        setattr(s, k, v)
    rc = p.L.easyav1_update_settings(p.h, ctypes.byref(s))
    p.settings = s
    return rc


def test_switching_the_audio_track_while_the_file_plays(ea, monkeypatch):
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fx, (h1, p1), (h2, p2) = two_audio_file()
    p = ea.Player(data, interlace_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    assert p.get_total_audio_tracks() == 2 and p.get_audio_channels() == 2
    stamps, shown, audio = drain(p, ea, fx, until=31)
    pos = p.get_current_timestamp()
    want1, counts1 = pcm_from(h1, p1, 0)
    got1 = np.concatenate([a for _, a in audio], axis=1)
    assert np.array_equal(got1.view(np.uint32), want1[:, :got1.shape[1]].view(np.uint32)) and got1.shape[1] > 0
    assert update(p, ea, audio_track=1) == ea.STATUS_OK
    assert p.get_audio_channels() == 6 and p.get_current_timestamp() == pos
    # the second track from the first packet at or after the position, like the start of a stream
    first = next(k for k in range(len(p2)) if 21 * k >= pos)
    want2, counts2 = pcm_from(h2, p2, first)
    stamps2, shown2, audio2 = drain(p, ea, fx, shown0=(pos + 39) // 40)   # (one shown picture per temporal unit, 40 ms apart)
    assert p.is_finished()
    got2 = np.concatenate([a for _, a in audio2], axis=1)
    assert got2.shape == want2.shape and np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
    assert [t for t, _ in audio2] == [21 * (first + k) for k, c in enumerate(counts2) if c > 0]
    # the pictures go on where they were (the one AT the position comes again: the reference's seek loop stops in front of that packet)
    assert stamps2 and stamps2[0] >= pos and shown2 == len(fx["shown_sha"]) and stamps2 == sorted(stamps2)
    p.close()


def test_toggling_interlace_audio_and_the_audio_offset(ea, monkeypatch):
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fx, (h1, p1), _ = two_audio_file()
    p = ea.Player(data, interlace_audio=1, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    drain(p, ea, fx, until=20)
    pos = p.get_current_timestamp()
    assert update(p, ea, interlace_audio=0) == ea.STATUS_OK and p.get_current_timestamp() == pos
    _, _, audio = drain(p, ea, fx, shown0=(pos + 39) // 40, until=25)     # planar frames now (Player.audio_frame reads the layout the settings name)
    first = next(k for k in range(len(p1)) if 21 * k >= pos)
    want, counts = pcm_from(h1, p1, first)
    got = np.concatenate([a for _, a in audio], axis=1)
    assert got.shape[1] > 0 and np.array_equal(got.view(np.uint32), want[:, :got.shape[1]].view(np.uint32))
    # audio 100 ms later than the container says: the packets' timestamps move, the samples do not
    pos = p.get_current_timestamp()
    assert update(p, ea, audio_offset_time=100) == ea.STATUS_OK and p.get_current_timestamp() == pos
    _, _, audio = drain(p, ea, fx, shown0=(pos + 39) // 40)
    assert p.is_finished()
    first = next(k for k in range(len(p1)) if 21 * k + 100 >= pos)
    want, counts = pcm_from(h1, p1, first)
    got = np.concatenate([a for _, a in audio], axis=1)
    # (packets whose shifted timestamp lies beyond the file's duration are dropped, src/easyav1.c:1970-1974)
    keep = [k for k in range(first, len(p1)) if 21 * k + 100 <= int(p.get_duration())]
    assert got.shape[1] > 0 and np.array_equal(got.view(np.uint32), want[:, :got.shape[1]].view(np.uint32))
    assert len(audio) > 3 and [t for t, _ in audio] == [21 * (first + k) + 100 for k, c in enumerate(counts) if c > 0 and first + k in keep][:len(audio)]
    p.close()


def test_disabling_the_video_mid_stream_and_the_switched_off_kind_stays_off(ea, monkeypatch):
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fx, (h1, p1), _ = two_audio_file()
    p = ea.Player(data, interlace_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    stamps, shown, _ = drain(p, ea, fx, until=16)
    assert shown > 0
    pos = p.get_current_timestamp()
    assert update(p, ea, enable_video=0) == ea.STATUS_OK
    assert not p.has_video_frame()
    _, _, audio = drain(p, ea, fx, until=12)
    first = next(k for k in range(len(p1)) if 21 * k >= pos)
    want, _ = pcm_from(h1, p1, first)
    got = np.concatenate([a for _, a in audio], axis=1)
    assert got.shape[1] > 0 and np.array_equal(got.view(np.uint32), want[:, :got.shape[1]].view(np.uint32))
    assert not p.has_video_frame()
    # enable_video again WITHOUT naming another track: change_track is not called (src/easyav1.c:3924-3927), the video stays off
    assert update(p, ea, enable_video=1) == ea.STATUS_OK
    stamps3, shown3, audio3 = drain(p, ea, fx, shown0=0)
    assert p.is_finished() and shown3 == 0 and stamps3 == [] and len(audio3) > 0
    p.close()


def test_codec_delay_moves_the_audio_timestamps(ea, monkeypatch):
    """CodecDelay of the audio track is added to the audio offset (src/easyav1.c:1495-1496)"""
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fx, _, (h2, p2) = two_audio_file(nb=30, codec_delay=60 * 1000000)
    p = ea.Player(data, interlace_audio=0, audio_track=1, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    _, _, audio = drain(p, ea, fx)
    want, counts = pcm_from(h2, p2, 0)
    got = np.concatenate([a for _, a in audio], axis=1)
    assert np.array_equal(got.view(np.uint32), want[:, :got.shape[1]].view(np.uint32)) and got.shape[1] > 0
    assert [t for t, _ in audio] == [21 * k + 60 for k, c in enumerate(counts) if c > 0][:len(audio)]
    p.close()


# ------------------------------------------------------------------------------------------------ two video tracks
def two_video_file(names=("pyramid_200x136", "c444_screen_200x136")):
    """two V_AV1 tracks of different content (and chroma layout) in one file, blocks interleaved by time"""
    tracks, blocks, fxs = [], [], []
    for k, name in enumerate(names):
        fx = tr.load_fixture(os.path.join(tr.HERE, "golden", "av1_stream_%s.npz" % name))
        vt, vb, _ = video_track(name, 3)
        vt = dict(vt, number=k + 1)
        tracks.append(vt)
        blocks.append([dict(b, track=k + 1) for b in vb])
        fxs.append(fx)
    end = max(b[-1]["tc"] for b in blocks) + 40
    return ww.make_webm(29, tracks, merge(*blocks), duration=float(end)), fxs


@pytest.mark.parametrize("names", [("pyramid_200x136", "c444_screen_200x136"), ("mono_136x104", "deltalf_q30_264x200")], ids=["420_and_444", "400_and_420"])
def test_the_video_track_is_the_one_the_settings_name(ea, names, monkeypatch):
    """settings.video_track picks among the file's video tracks (src/easyav1.c:1279-1400: the n-th track of the kind); each track gets the decoder context of ITS
    sequence header -- size, chroma layout -- and its pictures are dav1d's"""
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fxs = two_video_file(names)
    for k, fx in enumerate(fxs):
        p = ea.Player(data, enable_audio=0, video_track=k, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
        assert p.get_total_video_tracks() == 2 and (p.get_video_width(), p.get_video_height()) == (fx["width"], fx["height"])
        stamps, shown, _ = drain(p, ea, fx)
        assert p.is_finished() and shown == 3 * len(fx["shown_sha"]) and stamps == sorted(stamps)
        p.close()


def test_switching_the_video_track_while_the_file_plays(ea, monkeypatch):
    """easyav1_update_settings with another video_track (src/easyav1.c:3852-3949, change_track): the decoder of the old track is closed, the new one opened from its
    own sequence header (another chroma layout here: another context), and the file goes on from the position -- from the last key frame of the NEW track at or before
    it, pictures in between reconstructed and dropped: what comes out is dav1d's"""
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    data, fxs = two_video_file(("two_gops_104x72", "c444_objects_264x200"))     # 24 pictures each; GOPs of 4 / of 8
    p = ea.Player(data, enable_audio=0, video_track=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    stamps, shown, _ = drain(p, ea, fxs[0], until=11)
    assert shown > 0
    pos = p.get_current_timestamp()
    assert update(p, ea, video_track=1) == ea.STATUS_OK and p.get_current_timestamp() == pos
    assert (p.get_video_width(), p.get_video_height()) == (fxs[1]["width"], fxs[1]["height"])
    fx = fxs[1]
    n = len(fx["units"])
    assert len(fx["shown_sha"]) == n          # (one shown picture per temporal unit: the index of a picture follows from its timestamp)
    got = []
    while p.decode_next() == ea.STATUS_OK:
        if p.has_video_frame():
            ts, planes, props = p.video_frame()
            assert props.pixel_layout == 4 and planes[1].shape == planes[0].shape
            assert ts % 40 == 0 and np.array_equal(tr.sha(planes), fx["shown_sha"][(ts // 40) % n]), ts
            got.append(ts)
    assert p.is_finished() and got and got[0] >= pos and got == sorted(got) and got[-1] == 40 * (3 * n - 1)
    p.close()
