"""Vorbis packet front end (easyav1_amd/csrc/vorbis_front.cpp) against the reference's own decoder
(oracle/_ref/libref_minivorbis.so = /root/reference/ext/minivorbis compiled where it lies).

Real Vorbis data: there is no encoder in this environment, so the test streams are the Ogg Vorbis
files that ship inside Python packages of the image (third-party assets: they are read in place and
only hashes of their decoded PCM are kept in the repository).  Tests skip when no file is found."""
import ctypes
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

CANDIDATES = sorted(set(glob.glob("/usr/local/lib/python3*/dist-packages/**/*.ogg", recursive=True) +
                        glob.glob("/usr/lib/python3/dist-packages/**/*.ogg", recursive=True)))


def ogg_packets(path):
    """Ogg pages -> packets of the first logical stream"""
    data = open(path, "rb").read()
    pos, packets, cur, serial = 0, [], b"", None
    while pos + 27 <= len(data) and data[pos:pos + 4] == b"OggS":
        nseg = data[pos + 26]
        ser = data[pos + 14:pos + 18]
        table = data[pos + 27:pos + 27 + nseg]
        body = pos + 27 + nseg
        if serial is None:
            serial = ser
        for s in table:
            if ser == serial:
                cur += data[body:body + s]
                if s < 255:
                    packets.append(cur)
                    cur = b""
            body += s
        pos = body
    return packets


def vorbis_files():
    out = []
    for p in CANDIDATES:
        try:
            pk = ogg_packets(p)
        except OSError:
            continue
        if len(pk) > 3 and pk[0][:7] == b"\x01vorbis" and pk[2][:7] == b"\x05vorbis":
            out.append(p)
    return out


FILES = vorbis_files()
needs_file = pytest.mark.skipif(not FILES, reason="no Ogg Vorbis file in the image")


def ref_lib():
    L = ol.ref_minivorbis()
    if L is None:
        pytest.skip("oracle/_ref/libref_minivorbis.so not built")
    if not getattr(L, "_stream_bound", False):
        vp = ctypes.c_void_p
        L.ref_stream_new.argtypes = [ctypes.c_char_p, ctypes.c_long] * 3
        L.ref_stream_new.restype = vp
        L.ref_stream_free.argtypes = [vp]
        L.ref_stream_channels.argtypes = [vp]
        L.ref_stream_blocksize.argtypes = [vp, ctypes.c_int]
        L.ref_stream_decode.argtypes = [vp, ctypes.c_char_p, ctypes.c_long, vp, ctypes.c_int]
        L.ref_floor1_db_table.restype = ctypes.POINTER(ctypes.c_float)
        L._stream_bound = True
    return L


def ref_decode(packets):
    """-> (pcm [channels, total], counts per audio packet (-1 = rejected), channels, (bs0, bs1))"""
    L = ref_lib()
    h = L.ref_stream_new(packets[0], len(packets[0]), packets[1], len(packets[1]), packets[2], len(packets[2]))
    assert h, "reference rejected the headers"
    ch = L.ref_stream_channels(h)
    bs = (L.ref_stream_blocksize(h, 0), L.ref_stream_blocksize(h, 1))
    cap = bs[1]
    chunks, counts = [], []
    buf = np.zeros((ch, cap), np.float32)
    for p in packets[3:]:
        n = L.ref_stream_decode(h, p, len(p), buf.ctypes.data, cap)
        counts.append(n)
        if n > 0:
            chunks.append(buf[:, :n].copy())
    L.ref_stream_free(h)
    pcm = np.concatenate(chunks, axis=1) if chunks else np.zeros((ch, 0), np.float32)
    return pcm, counts, ch, bs


def test_floor1_db_table_matches_reference():
    from easyav1_amd import vorbis
    L = ref_lib()
    want = np.ctypeslib.as_array(L.ref_floor1_db_table(), shape=(256,))
    got = vorbis.floor1_db_table()
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


@needs_file
@pytest.mark.parametrize("path", FILES[:4])
def test_front_end_spectra_give_reference_pcm(path):
    """host stage only: packets -> spectra through the product front end, IMDCT + overlap-add through
    the CPU oracle (itself pinned against the reference), PCM bit-exact with the reference decoding
    the same packets end to end"""
    from easyav1_amd import vorbis
    packets = ogg_packets(path)
    want, counts, ch, bs = ref_decode(packets)
    dec = vorbis.VorbisDecoder(packets[0], packets[1], packets[2], device=-1)
    assert dec.channels == ch and dec.blocksizes == bs
    flags, blocks = [], []
    for p, c in zip(packets[3:], counts):
        r = dec.spectra(p)
        assert (r is None) == (c < 0)
        if r is None:
            continue
        spec, W = r
        full = np.zeros((ch, bs[W]), np.float32)      # the oracle takes the mdct_backward input layout
        full[:, :bs[W] // 2] = spec
        flags.append(W)
        blocks.append(full)
    dec.close()
    got = ol.oracle_synth_from_spectra(ch, bs[0], bs[1], flags, blocks)
    assert got.shape == want.shape and want.shape[1] > 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert any(flags) and not all(flags) or len(flags) < 8   # both block sizes occur in any real stream


@needs_file
def test_truncated_and_foreign_packets():
    """end-of-packet behaviour: a truncated audio packet decodes what it holds (the reference does
    the same); header / garbage packets are rejected by both"""
    from easyav1_amd import vorbis
    packets = ogg_packets(FILES[0])
    L = ref_lib()
    dec = vorbis.VorbisDecoder(packets[0], packets[1], packets[2], device=-1)
    h = L.ref_stream_new(packets[0], len(packets[0]), packets[1], len(packets[1]), packets[2], len(packets[2]))
    cap = dec.blocksizes[1]
    buf = np.zeros((dec.channels, cap), np.float32)
    audio = packets[3:]
    victims = [audio[len(audio) // 2][:k] for k in (1, 2, 5, 17, len(audio[len(audio) // 2]) // 2)] + [packets[0], b"", b"\xff" * 40]
    for p in victims:
        n = L.ref_stream_decode(h, p, len(p), buf.ctypes.data, cap)
        r = dec.spectra(p) if len(p) else None
        assert (r is None) == (n < 0), (len(p), n)
    L.ref_stream_free(h)
    dec.close()


@needs_file
@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES[:4])
def test_packets_to_pcm_on_gpu(path):
    """the whole audio path on real data: packets -> host front end -> GPU IMDCT / window / overlap-add,
    in one call and split over several calls, bit-exact with the reference"""
    from easyav1_amd import vorbis
    packets = ogg_packets(path)
    want, counts, ch, bs = ref_decode(packets)
    dec = vorbis.VorbisDecoder(packets[0], packets[1], packets[2], device=0)
    got, got_counts = dec.decode(packets[3:])
    assert got_counts == [max(c, 0) for c in counts]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    dec.restart()
    audio = packets[3:]
    parts = [dec.decode(audio[i:i + 3])[0] for i in range(0, len(audio), 3)]
    again = np.concatenate(parts, axis=1)
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
    dec.close()
    # the decoded PCM of this third-party file is pinned by hash only (tests/golden/vorbis_pcm_hashes.json)
    key = "%s:%d" % (os.path.basename(path), os.path.getsize(path))
    known = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vorbis_pcm_hashes.json")))
    if key in known:
        assert hashlib.sha256(want.tobytes()).hexdigest() == known[key]


# ---------------------------------------------------------------- synthetic setups, random packets
def _decode_both_cpu(hdr, packets):
    from easyav1_amd import vorbis
    all_packets = list(hdr) + packets
    want, counts, ch, bs = ref_decode(all_packets)
    dec = vorbis.VorbisDecoder(hdr[0], hdr[1], hdr[2], device=-1)
    flags, blocks = [], []
    for p, c in zip(packets, counts):
        r = dec.spectra(p)
        assert (r is None) == (c < 0)
        if r is None:
            continue
        spec, W = r
        full = np.zeros((ch, bs[W]), np.float32)
        full[:, :bs[W] // 2] = spec
        flags.append(W)
        blocks.append(full)
    dec.close()
    got = ol.oracle_synth_from_spectra(ch, bs[0], bs[1], flags, blocks)
    return want, got, flags


@pytest.mark.parametrize("seed,channels,bs", [(1, 2, (256, 2048)), (2, 1, (64, 512)), (3, 6, (256, 2048)), (4, 2, (128, 128)),
                                               (5, 3, (512, 4096)), (6, 2, (256, 1024)), (7, 4, (64, 8192)), (8, 2, (256, 2048)),
                                               (9, 5, (128, 2048)), (10, 2, (256, 2048)), (11, 6, (256, 2048)), (12, 2, (1024, 2048))])
def test_random_setups_match_reference(seed, channels, bs):
    """random valid setup headers (all codebook forms, floor-1 layouts, residue types 0/1/2 with
    cascades, coupling, submaps) + random-bit audio packets: same PCM as the reference, bit for bit"""
    from tests import vorbis_stream_gen as gen
    hdr = gen.make_stream(seed, channels, bs)
    packets = gen.random_packets(seed, hdr[3], count=30)
    want, got, flags = _decode_both_cpu(hdr[:3], packets)
    assert got.shape == want.shape and want.shape[1] > 0
    assert np.isfinite(want).all()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,channels,bs", [(21, 2, (256, 2048)), (22, 6, (256, 2048)), (23, 1, (64, 8192)), (24, 3, (128, 1024))])
def test_random_setups_on_gpu(seed, channels, bs):
    from easyav1_amd import vorbis
    from tests import vorbis_stream_gen as gen
    hdr = gen.make_stream(seed, channels, bs)
    packets = gen.random_packets(seed, hdr[3], count=60)
    want, counts, ch, _ = ref_decode(list(hdr[:3]) + packets)
    dec = vorbis.VorbisDecoder(hdr[0], hdr[1], hdr[2], device=0)
    got, got_counts = dec.decode(packets)
    dec.close()
    assert got_counts == [max(c, 0) for c in counts]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------- codebook rules (Vorbis I 3.2.1, 9.2.3)
def _setup_with_book(write_book0):
    """smallest legal stream (1 channel, 64-sample blocks, one floor / residue / mapping / mode) whose
    codebook 0 is written by the caller; codebook 1 is a complete two-entry scalar book used by
    the floor and as residue class book.  -> (id, comment, setup)"""
    from tests import vorbis_stream_gen as gen
    idw = gen.BitWriter()
    idw.write(0, 32)
    idw.write(1, 8)
    idw.write(48000, 32)
    for _ in range(3):
        idw.write(0, 32)
    idw.write(6, 4)
    idw.write(6, 4)
    idw.write(1, 1)
    cw = gen.BitWriter()
    cw.write(0, 32)
    cw.write(0, 32)
    cw.write(1, 1)
    w = gen.BitWriter()
    w.write(1, 8)                      # two codebooks
    write_book0(w)
    w.write(0x564342, 24); w.write(1, 16); w.write(2, 24); w.write(0, 1); w.write(0, 1)
    w.write(0, 5); w.write(0, 5); w.write(0, 4)                 # lengths 1, 1; no value mapping
    w.write(0, 6); w.write(0, 16)      # time domain placeholder
    w.write(0, 6); w.write(1, 16)      # one floor of type 1
    w.write(1, 5); w.write(0, 4)       # one partition of class 0
    w.write(0, 3); w.write(0, 2); w.write(2, 8)                 # class 0: one dimension, no subclasses, book 1
    w.write(0, 2); w.write(4, 4); w.write(5, 4)                 # multiplier 1, 4 range bits, x = 5
    w.write(0, 6); w.write(0, 16)      # one residue of type 0
    w.write(0, 24); w.write(8, 24); w.write(3, 24); w.write(0, 6); w.write(1, 8)
    w.write(0, 3); w.write(0, 1)       # no cascade stages
    w.write(0, 6); w.write(0, 16)      # one mapping
    w.write(0, 1); w.write(0, 1); w.write(0, 2)
    w.write(0, 8); w.write(0, 8); w.write(0, 8)
    w.write(0, 6)                      # one mode
    w.write(0, 1); w.write(0, 16); w.write(0, 16); w.write(0, 8)
    w.write(1, 1)
    return b"\x01vorbis" + idw.bytes(), b"\x03vorbis" + cw.bytes(), b"\x05vorbis" + w.bytes()


def _both_accept(hdr):
    """-> (reference accepts, product accepts)"""
    from easyav1_amd import vorbis
    from easyav1_amd._lib import Ea1Error
    L = ref_lib()
    h = L.ref_stream_new(hdr[0], len(hdr[0]), hdr[1], len(hdr[1]), hdr[2], len(hdr[2]))
    if h:
        L.ref_stream_free(h)
    try:
        vorbis.VorbisDecoder(hdr[0], hdr[1], hdr[2], device=-1).close()
        ours = True
    except Ea1Error:
        ours = False
    return bool(h), ours


def _plain_book(lengths, dim=1):
    def write(w):
        w.write(0x564342, 24)
        w.write(dim, 16)
        w.write(len(lengths), 24)
        w.write(0, 1)
        w.write(1, 1)                  # sparse: a length of 0 marks an unused entry
        for L in lengths:
            w.write(1 if L else 0, 1)
            if L:
                w.write(L - 1, 5)
        w.write(0, 4)
    return write


@pytest.mark.parametrize("lengths,accept", [
    ([1, 1], True), ([1, 2, 2], True), ([2, 2, 2, 2], True), ([3, 1, 3, 2], True), ([2, 0, 1, 0, 2], True),
    ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 32], True),
    ([1, 1, 1], False), ([1, 2, 2, 2], False), ([2, 2, 2, 2, 5], False), ([1, 2, 3, 3, 1], False),   # overspecified
    ([1, 2], False), ([2, 2, 2], False), ([3, 3], False), ([1, 3, 3, 3], False), ([32, 32], False),    # underspecified
    ([1], True), ([0, 0, 1, 0], True),     # the one-entry book: codeword "0"
    ([2], False), ([5], False), ([0, 3, 0], False),   # one entry of any other length leaves positions over
    ([0, 0, 0], True), ([], True),         # nothing to decode from: legal until used
])
def test_codeword_trees_accepted_like_the_reference(lengths, accept):
    """Vorbis I 3.2.1: over- and underspecified codeword trees are errors; the reference decoder's
    tolerated exception is the single-entry book (minivorbis.h:18478-18488)"""
    ref, ours = _both_accept(_setup_with_book(_plain_book(lengths)))
    assert ref == accept, "expectation about the reference is wrong"
    assert ours == ref


@pytest.mark.parametrize("entries,dim", [(1, 1), (1, 5), (2, 1), (7, 3), (8, 3), (9, 3), (26, 3), (27, 3), (28, 3), (4095, 2), (4096, 2), (4096, 3),
                                          (4097, 3), (6560, 8), (6561, 8), (6562, 8), (59048, 10), (59049, 10), (65535, 16), (65536, 16),
                                          (100000, 5), (99999, 5), (16, 65535 // 4096), (255, 255)])
def test_lookup1_values_consumes_the_same_bits_as_the_reference(entries, dim):
    """a lookup-type-1 book stores lookup1_values(entries, dim) quantised values (Vorbis I 9.2.3); a
    wrong count shifts every later bit of the setup header, which then fails to parse"""
    vals = 0
    while (vals + 1) ** dim <= entries:
        vals += 1
    # a complete tree over `entries` entries: entries - 2^k pairs one level below 2^k
    k = entries.bit_length() - 1
    deep = 2 * (entries - (1 << k))
    lengths = [k + 1] * deep + [k] * (entries - deep) if entries > 1 else [1]

    def write(w):
        w.write(0x564342, 24)
        w.write(dim, 16)
        w.write(entries, 24)
        w.write(0, 1)
        w.write(0, 1)
        for L in lengths:
            w.write(L - 1, 5)
        w.write(1, 4)
        w.write(0, 32); w.write(0, 32); w.write(3, 4); w.write(0, 1)   # minimum, delta, 4 value bits, no sequence
        for i in range(vals):
            w.write(i & 15, 4)
    ref, ours = _both_accept(_setup_with_book(write))
    assert ref and ours
