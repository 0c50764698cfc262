"""libeasyav1.so: the reference's public interface (include/easyav1.h == /root/reference/src/easyav1.h
in names, values and layouts) on top of this repository's C ABI.

CPU part: exported symbols, struct layouts against the reference header (compiled side by side, build
container only), the reference's own tools/easyav1_benchmark.c compiled IN PLACE against this header and
library, handle / error conventions, a video-only file (track facts, packet order, end of stream).
GPU part (-m gpu): audio of a WebM file pulled through easyav1_decode_next / easyav1_get_audio_frame,
bit-exact with the reference build's PCM; parsed-frame video through easyav1_get_video_frame, bit-exact
with the oracle; the reference's benchmark tool run against a generated file."""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from tests import av1_obu_writer as wr
from tests import test_frame_header as tfh
from tests import vorbis_stream_gen as vgen
from tests import webm_writer as ww

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "easyav1_benchmark")   # built by oracle/Makefile from the reference's tool source
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def ea():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import easyav1
    easyav1.api()
    return easyav1


def has_gpu():
    from easyav1_amd import _lib
    return _lib.lib().ea1_device_count() > 0


# ------------------------------------------------------------------------------------------ symbols / layouts
def test_exports_the_34_entry_points(ea):
    out = subprocess.check_output(["nm", "-D", "--defined-only", ea.library_path()], text=True)
    # EVERY defined dynamic symbol, whatever its name (round-4 verdict: the sink's C++ functions used to leak out beside the 34)
    got = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) >= 3)
    assert got == sorted(ea.EXPORTS) and len(got) == 34, [g for g in got if g not in ea.EXPORTS]
    assert "easyav1_has_audio_frame" not in got   # mentioned by the reference's README, never existed


LAYOUT_PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "easyav1.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
    S(easyav1_timestamp); S(easyav1_bool); S(easyav1_status); S(easyav1_stream); S(easyav1_video_frame); S(easyav1_audio_frame);
    S(easyav1_settings); S(easyav1_log_level_t); S(easyav1_pixel_layout);
    O(easyav1_stream, read_func); O(easyav1_stream, seek_func); O(easyav1_stream, tell_func); O(easyav1_stream, userdata);
    O(easyav1_video_frame, properties.pixel_layout); O(easyav1_video_frame, properties.bits_per_color); O(easyav1_video_frame, properties.color_space);
    O(easyav1_video_frame, properties.color_primaries); O(easyav1_video_frame, properties.transfer_characteristics);
    O(easyav1_video_frame, properties.matrix_coefficients); O(easyav1_video_frame, properties.chroma_sample_position);
    O(easyav1_video_frame, properties.width); O(easyav1_video_frame, properties.height); O(easyav1_video_frame, timestamp);
    O(easyav1_video_frame, data); O(easyav1_video_frame, stride);
    O(easyav1_audio_frame, channels); O(easyav1_audio_frame, samples); O(easyav1_audio_frame, timestamp); O(easyav1_audio_frame, bytes); O(easyav1_audio_frame, pcm);
    O(easyav1_settings, enable_video); O(easyav1_settings, enable_audio); O(easyav1_settings, skip_unprocessed_frames); O(easyav1_settings, interlace_audio);
    O(easyav1_settings, close_handle_on_destroy); O(easyav1_settings, callbacks.video); O(easyav1_settings, callbacks.audio); O(easyav1_settings, callbacks.userdata);
    O(easyav1_settings, video_track); O(easyav1_settings, audio_track); O(easyav1_settings, use_fast_seeking); O(easyav1_settings, audio_offset_time);
    O(easyav1_settings, log_level);
    printf("enums %d %d %d %d %d %d %d %d %d %d %d\n", EASYAV1_STATUS_ERROR, EASYAV1_STATUS_FINISHED, EASYAV1_STATUS_UNSUPPORTED, EASYAV1_PIXEL_LAYOUT_YUV444,
           EASYAV1_COLOR_SPACE_FULL, EASYAV1_BITS_PER_COLOR_12, EASYAV1_COLOR_PRIMARIES_EBU3213, EASYAV1_TRANSFER_CHARACTERISTICS_HLG,
           EASYAV1_MATRIX_COEFFICIENTS_ICTCP, EASYAV1_CHROMA_SAMPLE_POSITION_COLOCATED, EASYAV1_LOG_LEVEL_INFO);
    printf("enums2 %d %d %d %d %d %d\n", EASYAV1_STATUS_INVALID_ARGUMENT, EASYAV1_STATUS_OUT_OF_MEMORY, EASYAV1_STATUS_IO_ERROR, EASYAV1_STATUS_DECODER_ERROR,
           EASYAV1_STATUS_NOT_IMPLEMENTED, EASYAV1_STATUS_INVALID_STATE);
    printf("enums3 %d %d %d %d %d %d %d\n", EASYAV1_STATUS_INVALID_DATA, EASYAV1_COLOR_PRIMARIES_BT2020, EASYAV1_TRANSFER_CHARACTERISTICS_SRGB,
           EASYAV1_MATRIX_COEFFICIENTS_BT2020_CL, EASYAV1_TRANSFER_CHARACTERISTICS_SMPTE2084, EASYAV1_COLOR_PRIMARIES_XYZ, EASYAV1_MATRIX_COEFFICIENTS_SMPTE2085);
    return 0;
}
"""


@needs_ref
def test_struct_layouts_and_enumerators_equal_the_reference_header(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text(LAYOUT_PROBE)
    outs = []
    for inc in (os.path.join(REF, "src"), os.path.join(ROOT, "include")):
        exe = tmp_path / ("probe_" + ("ref" if inc.startswith(REF) else "own"))
        subprocess.check_call(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)])
        outs.append(subprocess.check_output([str(exe)], text=True))
    assert outs[0] == outs[1] and "easyav1_settings" in outs[0]


@needs_ref
def test_every_reference_prototype_is_declared_identically(tmp_path):
    """both headers in ONE translation unit would clash on the typedefs, so the prototypes are compared as text:
    each `easyav1_*(...)` declaration of the reference header, whitespace-normalised, occurs in ours"""
    import re

    def protos(path):
        text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        return {re.sub(r"\s+", " ", m).replace("( ", "(").strip() for m in re.findall(r"\n[a-z_0-9A-Z ]+\*?\s*\*?easyav1_[a-z_0-9]+\([^;{]*\);", text)}
    ref = protos(os.path.join(REF, "src", "easyav1.h"))
    own = protos(os.path.join(ROOT, "include", "easyav1.h"))
    norm = lambda s: {re.sub(r"\b(easyav1)\b(?=[,)])", "h", re.sub(r"\s*\*\s*", " *", p)) for p in s}
    assert len(ref) == 34 and norm(ref) == norm(own)


@needs_ref
def test_reference_benchmark_tool_compiles_in_place_against_this_library(ea, tmp_path):
    """tools/easyav1_benchmark.c of the reference, untouched and where it lies, against include/easyav1.h + libeasyav1.so"""
    exe = tmp_path / "easyav1_benchmark"
    libdir = os.path.dirname(ea.library_path())
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(REF, "tools", "easyav1_benchmark.c"), "-o", str(exe),
                           "-L", libdir, "-leasyav1", "-leasyav1_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([str(exe), str(tmp_path / "missing.webm")], capture_output=True, text=True)
    assert r.returncode == 2 and "Failed to initialize easyav1" in r.stdout


# ------------------------------------------------------------------------------------------ files for the tests
def av1_video_track(width=640, height=360, frames=12, frame_ms=33):
    """-> (track dict, blocks): a V_AV1 track of temporal units with real OBU framing and sequence headers
    (tile data random: nothing here can decode it, and nothing tries to)"""
    rng = np.random.default_rng(7)
    seq = tfh.simple_seq(max_width=width, max_height=height)
    seq_obu = wr.obu(1, wr.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=frame_ms * 1000000 + 366700,
                 width=width, height=height)
    blocks = []
    for k in range(frames):
        tu = wr.obu(2, b"") + (seq_obu if k % 6 == 0 else b"") + wr.obu(15, bytes(rng.integers(0, 256, 20, dtype=np.uint8)))   # padding OBU stands for the frame
        blocks.append(dict(track=1, tc=k * frame_ms, key=k % 6 == 0, frames=[tu], lacing=0, group=False, duration=None))
    return track, blocks


def vorbis_audio_track(seed, channels, bs, count, step_ms=21):
    hdr = vgen.make_stream(seed, channels, bs)
    packets = vgen.random_packets(seed, hdr[3], count=count)
    track = dict(number=2, type=2, codec="A_VORBIS", private=ww.vorbis_private(list(hdr[:3])), rate=48000.0, channels=channels)
    blocks = [dict(track=2, tc=k * step_ms, key=True, frames=[p], lacing=0, group=False, duration=None) for k, p in enumerate(packets)]
    return track, blocks, hdr, packets


def merge(*block_lists):
    return sorted((b for bl in block_lists for b in bl), key=lambda b: (b["tc"], b["track"]))


# ------------------------------------------------------------------------------------------ conventions (CPU)
def test_null_handles_and_constructors(ea):
    L = ea.api()
    s = L.easyav1_default_settings()
    assert (s.enable_video, s.enable_audio, s.skip_unprocessed_frames, s.interlace_audio, s.close_handle_on_destroy) == (1, 1, 1, 1, 0)
    assert (s.video_track, s.audio_track, s.use_fast_seeking, s.audio_offset_time, s.log_level) == (0, 0, 0, 0, ea.LOG_WARNING)
    s.log_level = ea.LOG_ERROR
    null = ctypes.c_void_p()
    assert L.easyav1_decode_next(null) == ea.STATUS_ERROR and L.easyav1_decode_until(null, 5) == ea.STATUS_ERROR
    assert L.easyav1_decode_for(null, 5) == ea.STATUS_ERROR and L.easyav1_play(null) == ea.STATUS_ERROR
    assert L.easyav1_seek_to_timestamp(null, 1) == ea.STATUS_ERROR and L.easyav1_get_status(null) == ea.STATUS_ERROR
    assert L.easyav1_get_current_timestamp(null) == 0 and L.easyav1_get_duration(null) == 0
    assert not L.easyav1_has_video_track(null) and not L.easyav1_has_video_frame(null) and not L.easyav1_is_finished(null)
    assert not L.easyav1_get_video_frame(null) and not L.easyav1_get_audio_frame(null) and L.easyav1_get_video_width(null) == 0
    L.easyav1_stop(null)
    L.easyav1_destroy(None)
    L.easyav1_destroy(ctypes.byref(null))
    assert not L.easyav1_init_from_memory(None, 10, ctypes.byref(s)) and not L.easyav1_init_from_memory(b"abc", 0, ctypes.byref(s))
    assert not L.easyav1_init_from_memory(b"not a webm file at all", 22, ctypes.byref(s))
    assert not L.easyav1_init_from_filename(None, ctypes.byref(s)) and not L.easyav1_init_from_filename(b"/nonexistent/x.webm", ctypes.byref(s))
    assert not L.easyav1_init_from_file(None, ctypes.byref(s)) and not L.easyav1_init_from_custom_stream(None, ctypes.byref(s))


def test_video_only_file_facts_order_and_end_of_stream(ea, tmp_path):
    vt, vb = av1_video_track(frames=12)
    at, ab, _, _ = vorbis_audio_track(5, 2, (256, 2048), 10)
    data = ww.make_webm(3, [vt, at], merge(vb, ab), duration=400.0)
    p = ea.Player(data, enable_audio=0, log_level=ea.LOG_ERROR)
    assert p.has_video_track() and not p.has_audio_track()
    assert (p.get_total_video_tracks(), p.get_total_audio_tracks()) == (1, 1)
    assert (p.get_video_width(), p.get_video_height(), p.get_video_fps()) == (640, 360, 29)     # 1e9 / 33366700 ns, integer division
    assert p.get_audio_channels() == 0 and p.get_audio_sample_rate() == 0 and p.get_duration() == 400
    assert p.get_status() == ea.STATUS_OK and p.get_current_timestamp() == 0 and not p.is_finished()
    stamps = []
    while True:
        st = p.decode_next()
        if st != ea.STATUS_OK:
            break
        stamps.append(p.get_current_timestamp())
        assert not p.has_video_frame() and not p.get_video_frame()      # headers only: no picture without the symbol parser
    assert st == ea.STATUS_FINISHED and p.is_finished() and p.get_status() == ea.STATUS_FINISHED
    assert stamps == [33 * k for k in range(12)]
    assert p.decode_next() == ea.STATUS_FINISHED and p.get_total_video_frames_processed() == 0
    # seeking back re-opens the stream; decode_until stops at the first packet at or after the target
    assert p.seek_to_timestamp(100) == ea.STATUS_OK and p.get_current_timestamp() == 100 and not p.is_finished()
    assert p.decode_next() == ea.STATUS_OK and p.get_current_timestamp() == 132
    assert p.decode_until(200) == ea.STATUS_OK and p.get_current_timestamp() == 200
    assert p.decode_for(50) == ea.STATUS_OK and p.get_current_timestamp() == 250
    assert p.seek_backward(1000) == ea.STATUS_OK and p.get_current_timestamp() == 0
    assert p.seek_forward(70) == ea.STATUS_OK and p.get_current_timestamp() == 70
    cur = p.get_current_settings()
    assert cur.enable_audio == 0 and cur.enable_video == 1
    p.close()
    # the same through a file name, a FILE* the library must not close, and a custom stream
    path = tmp_path / "v.webm"
    path.write_bytes(data)
    L = ea.api()
    s = L.easyav1_default_settings()
    s.enable_audio, s.log_level = 0, ea.LOG_ERROR
    h = ctypes.c_void_p(L.easyav1_init_from_filename(str(path).encode(), ctypes.byref(s)))
    assert h and L.easyav1_get_video_width(h) == 640
    n = 0
    while L.easyav1_decode_next(h) == ea.STATUS_OK:
        n += 1
    assert n == 12 and L.easyav1_is_finished(h)
    L.easyav1_destroy(ctypes.byref(h))
    assert not h.value
    pos = [0]
    blob = data

    def rd(buf, size, _):
        if pos[0] + size > len(blob):
            return 0
        ctypes.memmove(buf, blob[pos[0]:pos[0] + size], size)
        pos[0] += size
        return 1

    def sk(off, origin, _):
        base = {0: 0, 1: pos[0], 2: len(blob)}[origin]
        pos[0] = base + off
        return 0
    st = ea.Stream(ea.READ_FUNC(rd), ea.SEEK_FUNC(sk), ea.TELL_FUNC(lambda _: pos[0]), None)
    h = ctypes.c_void_p(L.easyav1_init_from_custom_stream(ctypes.byref(st), ctypes.byref(s)))
    assert h and L.easyav1_get_duration(h) == 400
    L.easyav1_destroy(ctypes.byref(h))


def test_malformed_video_packet_is_a_sticky_decoder_error(ea):
    vt, vb = av1_video_track(frames=4)
    vb[2]["frames"] = [b"\x12\x7f\x00"]     # an OBU whose size field runs past the packet
    p = ea.Player(ww.make_webm(4, [vt], vb, duration=200.0), enable_audio=0, log_level=-1)
    # packets are handed to the parser as they are prefetched, so the error is there at once and stays
    assert p.decode_next() == ea.STATUS_ERROR and p.get_status() == ea.STATUS_ERROR
    assert p.decode_next() == ea.STATUS_ERROR and p.decode_until(1000) == ea.STATUS_ERROR and not p.is_finished()
    p.close()


def test_track_without_frame_duration_is_refused_like_the_reference(ea):
    vt, vb = av1_video_track(frames=2)
    vt["default_duration"] = None
    with pytest.raises(RuntimeError):
        ea.Player(ww.make_webm(5, [vt], vb, duration=100.0), enable_audio=0, log_level=-1)


def test_audio_needs_the_gpu_and_says_so(ea):
    if has_gpu():
        pytest.skip("a device is present")
    at, ab, _, _ = vorbis_audio_track(5, 2, (256, 2048), 4)
    with pytest.raises(RuntimeError):     # no CPU fallback: the constructor fails, it does not decode on the host
        ea.Player(ww.make_webm(6, [at], ab, duration=100.0), log_level=-1)


# ------------------------------------------------------------------------------------------ GPU: audio
def reference_pcm(hdr, packets):
    from tests import test_vorbis_front as tv
    return tv.ref_decode(list(hdr[:3]) + packets)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,channels,bs,interlace", [(31, 2, (256, 2048), 1), (32, 6, (256, 2048), 0), (33, 1, (64, 8192), 1), (34, 3, (128, 1024), 0)])
def test_audio_frames_match_the_reference_build(ea, seed, channels, bs, interlace):
    """every audio packet through easyav1_decode_next, every frame through easyav1_get_audio_frame: the concatenation is
    the reference decoder's PCM of the same packets bit for bit, frame timestamps are the packets' timestamps"""
    at, ab, hdr, packets = vorbis_audio_track(seed, channels, bs, 70)
    vt, vb = av1_video_track(frames=20)
    want, counts, ch, _ = reference_pcm(hdr, packets)
    p = ea.Player(ww.make_webm(seed, [vt, at], merge(vb, ab), duration=1500.0), interlace_audio=interlace, log_level=ea.LOG_ERROR)
    assert p.has_audio_track() and p.has_video_track() and p.get_audio_channels() == channels and p.get_audio_sample_rate() == 48000
    got, stamps, n_ok = [], [], 0
    while p.decode_next() == ea.STATUS_OK:
        n_ok += 1
        f = p.audio_frame()
        if f is not None:
            stamps.append(f[0])
            got.append(f[1])
        assert p.audio_frame() is None            # a frame is handed out once
    assert p.is_finished() and n_ok == 90
    got = np.concatenate(got, axis=1)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    produced = [21 * k for k, c in enumerate(counts) if c > 0]
    assert stamps == produced
    p.close()


@pytest.mark.gpu
def test_audio_buffer_keeps_the_newest_4096_samples(ea):
    at, ab, hdr, packets = vorbis_audio_track(41, 2, (256, 2048), 40)
    want, counts, _, _ = reference_pcm(hdr, packets)
    p = ea.Player(ww.make_webm(41, [at], ab, duration=900.0), log_level=ea.LOG_ERROR)
    total, filled_seen = 0, False
    for k in range(25):
        assert p.decode_next() == ea.STATUS_OK
        total += max(counts[k], 0)
        filled_seen |= bool(p.is_audio_buffer_filled())
        assert bool(p.is_audio_buffer_filled()) == (total >= ea.AUDIO_BUFFER_SIZE)
    assert total > ea.AUDIO_BUFFER_SIZE and filled_seen
    ts, pcm = p.audio_frame()
    first = next(k for k, c in enumerate(counts) if c > 0)
    assert pcm.shape == (2, ea.AUDIO_BUFFER_SIZE) and ts == 21 * first   # the packet that put the first samples in since the last hand-out (src/easyav1.c:2441-2444)
    assert np.array_equal(pcm.view(np.uint32), want[:, total - ea.AUDIO_BUFFER_SIZE:total].view(np.uint32))
    assert not p.is_audio_buffer_filled()
    # decode_until takes several packets at once; seeking restarts the synthesis like the start of a stream
    assert p.decode_until(21 * 30) == ea.STATUS_OK and p.get_current_timestamp() == 630
    more = sum(max(c, 0) for c in counts[25:30])
    ts, pcm = p.audio_frame()
    assert ts == 21 * 25 and pcm.shape[1] == more and np.array_equal(pcm.view(np.uint32), want[:, total:total + more].view(np.uint32))
    assert p.seek_to_timestamp(0) == ea.STATUS_OK
    again = []
    while p.decode_next() == ea.STATUS_OK:
        f = p.audio_frame()
        if f:
            again.append(f[1])
    again = np.concatenate(again, axis=1)
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
    p.close()


@pytest.mark.gpu
def test_audio_callback_and_playback_thread(ea):
    at, ab, hdr, packets = vorbis_audio_track(43, 2, (256, 2048), 30)
    want, _, _, _ = reference_pcm(hdr, packets)
    chunks = []

    def on_audio(frame, _):
        f = frame.contents
        chunks.append(np.ctypeslib.as_array(ctypes.cast(f.pcm, ctypes.POINTER(ctypes.c_float)), shape=(f.samples, f.channels)).T.copy())
    cb = ea.AUDIO_CB(on_audio)
    L = ea.api()
    s = L.easyav1_default_settings()
    s.log_level = ea.LOG_ERROR
    s.callbacks.audio = cb
    data = ww.make_webm(43, [at], ab, duration=700.0)
    h = ctypes.c_void_p(L.easyav1_init_from_memory(data, len(data), ctypes.byref(s)))
    assert h and L.easyav1_play(h) == ea.STATUS_OK
    assert L.easyav1_decode_next(h) == ea.STATUS_OK          # refused politely while the playback thread owns the decoder
    import time
    t0 = time.time()
    while not L.easyav1_is_finished(h) and time.time() - t0 < 20:
        time.sleep(0.02)
    L.easyav1_stop(h)
    assert L.easyav1_is_finished(h) and time.time() - t0 >= 0.5   # paced by the wall clock: 30 packets x 21 ms
    got = np.concatenate(chunks, axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    L.easyav1_destroy(ctypes.byref(h))


# ------------------------------------------------------------------------------------------ GPU: parsed-frame video
def parsed_frame_file(width, height, nframes, seed, frame_ms=40, features=None, audio=None, gop=None):
    """a V_AV1 track whose temporal units carry the frames already parsed (include/easyav1_amd_ir_packet.h)
    -> (file bytes, [ParsedFrame]); gop: a key frame every `gop` frames (default: only the first)"""
    from easyav1_amd import ir
    gop = gop or max(nframes, 2)
    prm = ir.synth_params(width, height, 1, seed=seed, gop=gop)
    if features is not None:
        prm.features = features
    seq = tfh.simple_seq(max_width=width, max_height=height)
    seq_obu = wr.obu(1, wr.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=frame_ms * 1000000, width=width, height=height)
    frames, blocks = [], []
    for k in range(nframes):
        f = ir.synth_frame(prm, k)
        frames.append(f)
        tu = wr.obu(2, b"") + (seq_obu if k == 0 else b"") + wr.obu(5, ir.pack_frame(f))
        blocks.append(dict(track=1, tc=k * frame_ms, key=k % gop == 0, frames=[tu], lacing=0, group=False, duration=None))
    tracks = [track]
    if audio is not None:
        tracks.append(audio[0])
        blocks = merge(blocks, audio[1])
    return ww.make_webm(seed, tracks, blocks, duration=float(nframes * frame_ms), extras=False), frames


@pytest.mark.gpu
def test_parsed_frames_come_out_of_get_video_frame_bit_exact(ea):
    from tests import oracle_lib as ol
    W, H, N = 192, 128, 8
    data, frames = parsed_frame_file(W, H, N, seed=9)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    refs = [None] * 8
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        f = frames[shown]
        want = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = want
        assert ts == 40 * shown and (props.width, props.height) == (W, H)
        assert (props.pixel_layout, props.bits_per_color) == (2, 1)      # YUV420, 8 bit: from the sequence header
        for pl in range(3):
            assert np.array_equal(planes[pl], want[pl]), (shown, pl)
        assert not p.has_video_frame() and p.video_frame() is None
        shown += 1
    assert shown == N and p.is_finished() and p.get_total_video_frames_processed() == N
    p.close()


@pytest.mark.gpu
def test_resolution_switch_and_scaled_references_through_the_public_interface(ea):
    """a 192x128 track whose frames switch to 128x96 and back WITHOUT key frames: the first frame of either size predicts
    from references of the other (scaled references); every picture comes out at its own size, bit-exact"""
    from easyav1_amd import ir
    from tests import oracle_lib as ol
    feats = ir.SYN_RECT | ir.SYN_TX_SPLIT | ir.SYN_COMPOUND | ir.SYN_OBMC | ir.SYN_SUB8 | ir.SYN_DIAGONAL_COEFS
    big, small = ir.synth_params(192, 128, 1, seed=21, gop=16, features=feats), ir.synth_params(128, 96, 1, seed=22, gop=16, features=feats)
    plan = [big, big, big, small, small, big, small, big]   # frame k of the plan's generator: k = 0 the key frame, then inter frames
    frames = [ir.synth_frame(prm, k) for k, prm in enumerate(plan)]
    seq = tfh.simple_seq(max_width=192, max_height=128)
    seq_obu = wr.obu(1, wr.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=192, height=128)
    blocks = [dict(track=1, tc=k * 40, key=k == 0, frames=[wr.obu(2, b"") + (seq_obu if k == 0 else b"") + wr.obu(5, ir.pack_frame(f))], lacing=0, group=False, duration=None)
              for k, f in enumerate(frames)]
    data = ww.make_webm(5, [track], blocks, duration=float(len(frames) * 40), extras=False)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    refs = [None] * 8
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        f = frames[shown]
        want = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = want
        assert (props.width, props.height) == (f.desc.width, f.desc.height), shown
        for pl in range(3):
            assert np.array_equal(planes[pl], want[pl]), (shown, pl)
        shown += 1
    assert shown == len(frames) and p.is_finished()
    p.close()


@pytest.mark.gpu
def test_parsed_frames_and_audio_together_with_callbacks(ea):
    """configs[0] shape: video + Vorbis stereo in one file, both through the public interface"""
    from tests import oracle_lib as ol
    at, ab, hdr, packets = vorbis_audio_track(51, 2, (256, 2048), 20, step_ms=23)
    want_pcm, _, _, _ = reference_pcm(hdr, packets)
    data, frames = parsed_frame_file(128, 64, 10, seed=4, audio=(at, ab))
    seen_v, seen_a = [], []

    def on_video(frame, _):
        f = frame.contents
        y = np.ctypeslib.as_array(ctypes.cast(f.data[0], ctypes.POINTER(ctypes.c_uint8)), shape=(f.properties.height, f.stride[0]))[:, :f.properties.width]
        seen_v.append((f.timestamp, y.copy()))

    def on_audio(frame, _):
        f = frame.contents
        seen_a.append(np.ctypeslib.as_array(ctypes.cast(f.pcm, ctypes.POINTER(ctypes.c_float)), shape=(f.samples, f.channels)).T.copy())
    L = ea.api()
    s = L.easyav1_default_settings()
    s.log_level = ea.LOG_ERROR
    vcb, acb = ea.VIDEO_CB(on_video), ea.AUDIO_CB(on_audio)
    s.callbacks.video, s.callbacks.audio = vcb, acb
    h = ctypes.c_void_p(L.easyav1_init_from_memory(data, len(data), ctypes.byref(s)))
    assert h
    while L.easyav1_decode_next(h) == ea.STATUS_OK:
        pass
    assert L.easyav1_is_finished(h)
    L.easyav1_destroy(ctypes.byref(h))
    refs = [None] * 8
    assert [t for t, _ in seen_v] == [40 * k for k in range(10)]
    for (t, y), f in zip(seen_v, frames):
        want = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = want
        assert np.array_equal(y, want[0])
    got = np.concatenate(seen_a, axis=1)
    assert np.array_equal(got.view(np.uint32), want_pcm.view(np.uint32))


@pytest.mark.gpu
def test_corrupt_parsed_frame_is_an_error_not_a_fault(ea):
    data, frames = parsed_frame_file(128, 64, 3, seed=5)
    raw = bytearray(data)
    key = b"EA1F"
    at = raw.find(key, raw.find(key) + 1)          # second frame: claim more blocks than the payload holds
    desc_off = at + 4 + 4 + 16 + 16
    from easyav1_amd import ir
    nb_off = desc_off + ir.FrameDesc.num_blocks.offset
    raw[nb_off:nb_off + 4] = (10 ** 6).to_bytes(4, "little")
    p = ea.Player(bytes(raw), enable_audio=0, log_level=-1)
    st = p.decode_next()
    while st == ea.STATUS_OK:
        st = p.decode_next()
    assert st == ea.STATUS_ERROR and p.get_status() == ea.STATUS_ERROR
    p.close()


@pytest.mark.gpu
def test_reference_benchmark_tool_runs_on_a_generated_file(ea, tmp_path):
    """the reference's tools/easyav1_benchmark.c (compiled in place in the build container into oracle/_ref/) decoding
    a parsed-frame file through libeasyav1.so on the GPU: prints its own result lines"""
    if not os.path.exists(REF_TOOL):
        pytest.skip("oracle/_ref/easyav1_benchmark not built")
    data, frames = parsed_frame_file(640, 384, 30, seed=2)
    path = tmp_path / "bench.webm"
    path.write_bytes(data)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(ea.library_path()) + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([REF_TOOL, str(path)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Video size: 640x384, 25 FPS." in r.stdout and "Decoded 30 frames in" in r.stdout and "Slowest frame" in r.stdout


@pytest.mark.gpu
def test_seek_into_the_middle_of_a_gop_decodes_from_its_key_frame(ea):
    """seeking to a timestamp between key frames: the frames from the key frame on are reconstructed (and thrown away) inside
    the seek, so the first picture after it is the right one -- not a decoder error for want of references"""
    from tests import oracle_lib as ol
    W, H, N = 192, 128, 10
    data, frames = parsed_frame_file(W, H, N, seed=31)
    want, refs = [], [None] * 8
    for f in frames:
        out = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = out
        want.append(out)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    for target, first in ((170, 5), (40, 1), (390, None), (0, 0)):
        assert p.seek_to_timestamp(target) == ea.STATUS_OK and p.get_current_timestamp() == target
        assert p.get_status() == (ea.STATUS_OK if first is not None else ea.STATUS_FINISHED)    # beyond the last frame: nothing is left
        assert not p.has_video_frame()                       # what was decoded on the way to the target is not shown
        if first is None:
            assert p.decode_next() == ea.STATUS_FINISHED
            continue
        for k in range(first, min(first + 3, N)):
            assert p.decode_next() == ea.STATUS_OK and p.has_video_frame()
            ts, planes, props = p.video_frame()
            assert ts == 40 * k
            for pl in range(3):
                assert np.array_equal(planes[pl], want[k][pl]), (target, k, pl)
    p.close()


@pytest.mark.gpu
def test_polling_the_getters_while_the_playback_thread_decodes(ea):
    """the reference's player pattern (README, tools/easyav1_player.c): easyav1_play, then poll has_video_frame /
    get_video_frame / get_audio_frame from the caller's thread.  Pictures must be whole and in order, PCM complete."""
    from tests import oracle_lib as ol
    import time
    at, ab, hdr, packets = vorbis_audio_track(52, 2, (256, 2048), 30, step_ms=21)
    want_pcm, _, _, _ = reference_pcm(hdr, packets)
    data, frames = parsed_frame_file(128, 64, 16, seed=6, audio=(at, ab))
    want, refs = {}, [None] * 8
    for k, f in enumerate(frames):
        out = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = out
        want[40 * k] = out
    p = ea.Player(data, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    L = ea.api()
    assert L.easyav1_play(p.h) == ea.STATUS_OK
    seen, pcm = [], []
    t0 = time.time()
    while time.time() - t0 < 30:
        done = p.is_finished()      # read BEFORE the last poll: what was queued when the stream ended is still fetched below
        if p.has_video_frame():
            got = p.video_frame()
            if got is not None:
                ts, planes, props = got
                seen.append(ts)
                for pl in range(3):
                    assert np.array_equal(planes[pl], want[ts][pl]), (ts, pl)
        a = p.audio_frame()
        if a is not None:
            pcm.append(a[1])
        if done:
            break
        time.sleep(0.001)
    L.easyav1_stop(p.h)
    assert p.is_finished()
    assert seen == sorted(seen) and len(seen) >= 8 and set(seen) <= set(want)     # a slow poller may miss pictures (the ring drops), never sees a torn one
    if pcm:
        got = np.concatenate(pcm, axis=1)
        assert got.shape[1] <= want_pcm.shape[1]
    p.close()


@pytest.mark.gpu
def test_a_stream_longer_than_the_prefetch_and_a_caller_that_takes_no_pictures(ea):
    """40 frames through decode_next: more than the 10-packet prefetch and the 11-picture ring.  A caller that never asks for
    pictures must not stall or leak (the oldest waiting picture is dropped); one that asks gets every picture bit-exact"""
    from tests import oracle_lib as ol
    W, H, N = 128, 64, 40
    data, frames = parsed_frame_file(W, H, N, seed=12)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    n = 0
    while p.decode_next() == ea.STATUS_OK:
        n += 1
    assert n == N and p.is_finished() and p.get_total_video_frames_processed() == N
    p.close()
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    refs, shown = [None] * 8, 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        f = frames[shown]
        want = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = want
        assert ts == 40 * shown
        for pl in range(3):
            assert np.array_equal(planes[pl], want[pl]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gops_env", [None, "1", "3"])
def test_several_gops_in_flight_deliver_every_picture_in_order(ea, gops_env, monkeypatch):
    """a stream of 6 closed GOPs of 7 frames: the sink decodes several GOPs at once (one frame of each per launch, a key-frame
    packet starts the next chain in turn) and still hands the pictures out in stream order, each bit-exact; also with the
    look-ahead switched off (EASYAV1_AMD_GOPS=1) and with a chain count that does not divide the GOP count"""
    from tests import oracle_lib as ol
    if gops_env is None:
        monkeypatch.delenv("EASYAV1_AMD_GOPS", raising=False)
    else:
        monkeypatch.setenv("EASYAV1_AMD_GOPS", gops_env)
    W, H, N, G = 128, 64, 42, 7
    data, frames = parsed_frame_file(W, H, N, seed=23, gop=G)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    refs, shown = [None] * 8, 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        f = frames[shown]
        if shown % G == 0:
            refs = [None] * 8
        want = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = want
        assert ts == 40 * shown
        for pl in range(3):
            assert np.array_equal(planes[pl], want[pl]), (shown, pl)
        shown += 1
    assert shown == N and p.is_finished() and p.get_total_video_frames_processed() == N
    # a seek into the fourth GOP while later GOPs are in flight, then on to the end
    assert p.seek_to_timestamp(40 * 24 + 5) == ea.STATUS_OK
    k = 25
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        assert ts == 40 * k
        k += 1
    assert k == N
    p.close()


@pytest.mark.gpu
def test_fast_seeking_and_skipping_ahead_land_on_key_frames(ea):
    """settings.use_fast_seeking: a seek resumes AT the last key frame before the target (src/easyav1.c:3106-3111); decode_until with
    skip_unprocessed_frames more than a second behind jumps to the last key frame before its target instead of decoding everything
    in between (:2693-2713).  Pictures after either are the stream's own, bit-exact."""
    from tests import oracle_lib as ol
    W, H, N, G = 128, 64, 56, 7
    data, frames = parsed_frame_file(W, H, N, seed=29, gop=G)
    want, refs = [], [None] * 8
    for k, f in enumerate(frames):
        if k % G == 0:
            refs = [None] * 8
        out = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = out
        want.append(out)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, use_fast_seeking=1, log_level=ea.LOG_ERROR)
    assert p.seek_to_timestamp(40 * 24 + 5) == ea.STATUS_OK and p.get_current_timestamp() == 40 * 21     # frame 21 starts the GOP of frame 24
    for k in (21, 22, 23):
        assert p.decode_next() == ea.STATUS_OK and p.has_video_frame()
        ts, planes, _ = p.video_frame()
        assert ts == 40 * k
        for pl in range(3):
            assert np.array_equal(planes[pl], want[k][pl]), (k, pl)
    p.close()
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=1, log_level=ea.LOG_ERROR)
    assert p.decode_until(40 * 45 + 1) == ea.STATUS_OK and p.get_current_timestamp() == 40 * 45 + 1     # 1.8 s ahead: from key frame 42 on
    assert p.get_total_video_frames_processed() <= 5          # frames 42 .. 45, not 0 .. 45
    seen = []
    while p.has_video_frame():
        ts, planes, _ = p.video_frame()
        k = ts // 40
        seen.append(k)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[k][pl]), (k, pl)
    assert seen and seen[-1] == 45 and seen[0] >= 42
    p.close()
