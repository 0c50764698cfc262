"""OBU framing + sequence header parser (SURVEY.md 8(f) N1, table-free slice) against an independent
writer (tests/av1_obu_writer.py).  Parity with dav1d is UNPINNED: dav1d is absent and the reference
tree holds no AV1 stream; both sides follow AV1 specification 5.3 / 5.5."""
import ctypes

import numpy as np
import pytest

from tests import av1_obu_writer as wr


@pytest.fixture(scope="module")
def obu():
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import obu
    return obu


def test_struct_layout_matches_header(obu):
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "easyav1_amd_obu.h"\nint main(){printf("%zu %zu %zu %zu", sizeof(ea1_av1_sequence_header), offsetof(ea1_av1_sequence_header, max_width), offsetof(ea1_av1_sequence_header, operating_point_idc), sizeof(ea1_obu));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    H = obu.SequenceHeader
    assert sizes == [ctypes.sizeof(H), H.max_width.offset, H.operating_point_idc.offset, ctypes.sizeof(obu.Obu)]


def test_random_sequence_headers_round_trip(obu):
    rng = np.random.default_rng(11)
    seen = dict(reduced=0, timing=0, model=0, many_ops=0, mono=0, i444=0, i422=0, hbd12=0)
    for trial in range(400):
        h = wr.random_sequence_header(rng)
        payload = wr.write_sequence_header(h)
        data = wr.obu(2, b"") + wr.obu(1, payload, extension=(int(rng.integers(8)), int(rng.integers(4))) if trial % 5 == 0 else None,
                                       size_pad=int(rng.integers(0, 4)))
        if trial % 3 == 0:
            data += wr.obu(6, bytes(rng.integers(0, 256, 20, dtype=np.uint8)))
        rc, got = obu.parse_sequence_header(data)
        assert rc == 0, (trial, h)
        for k, v in got.items():
            want = h.get(k, 0)
            assert v == want, (trial, k, v, want)
        seen["reduced"] += h["reduced_still_picture_header"]
        seen["timing"] += h.get("timing_info_present", 0)
        seen["model"] += h.get("decoder_model_info_present", 0)
        seen["many_ops"] += h["num_operating_points"] > 8
        seen["mono"] += h["monochrome"]
        seen["i444"] += h["layout"] == 3
        seen["i422"] += h["layout"] == 2
        seen["hbd12"] += h["bit_depth"] == 12
    assert all(v > 5 for v in seen.values()), seen


def test_obu_framing(obu):
    rng = np.random.default_rng(2)
    body = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    # size field (also over-long leb128), extension, no size field (runs to the end)
    for kw in (dict(), dict(size_pad=3), dict(extension=(5, 2)), dict(has_size=False), dict(has_size=False, extension=(1, 1))):
        data = wr.obu(6, body, **kw)
        rc, o, payload = obu.next_obu(data + (b"" if not kw.get("has_size", True) else b"\x12\x00"))
        assert rc == 0 and payload == body and o.type == 6
        assert o.total_size == len(data) and o.has_size_field == int(kw.get("has_size", True))
        if "extension" in kw:
            assert (o.temporal_id, o.spatial_id) == kw["extension"] and o.has_extension == 1
    # walking a temporal unit
    tu = wr.obu(2, b"") + wr.obu(1, wr.write_sequence_header(wr.random_sequence_header(rng))) + wr.obu(6, body) + wr.obu(15, b"\0" * 7)
    pos, types = 0, []
    while pos < len(tu):
        rc, o, _ = obu.next_obu(tu[pos:])
        assert rc == 0
        types.append(o.type)
        pos += o.total_size
    assert types == [2, 1, 6, 15] and pos == len(tu)


def test_errors_and_absence(obu):
    import easyav1_amd
    L = easyav1_amd.load()
    L.ea1_last_error.restype = ctypes.c_char_p
    rng = np.random.default_rng(4)
    h = wr.random_sequence_header(rng)
    good = wr.obu(1, wr.write_sequence_header(h))
    ENOENT, EINVAL = -7, -1
    assert obu.parse_sequence_header(wr.obu(2, b"") + wr.obu(6, b"\x01\x02\x03"))[0] == ENOENT
    assert obu.parse_sequence_header(b"")[0] == ENOENT
    assert obu.parse_sequence_header(bytes([0x80 | good[0]]) + good[1:])[0] == EINVAL          # forbidden bit
    assert obu.parse_sequence_header(good[:-3])[0] == EINVAL                                   # payload shorter than its size field
    assert obu.parse_sequence_header(wr.obu(1, wr.write_sequence_header(h)[:2]))[0] == EINVAL   # truncated header
    bad = bytearray(wr.write_sequence_header(h))
    bad[0] |= 0xE0                                                                             # seq_profile 7
    assert obu.parse_sequence_header(wr.obu(1, bytes(bad)))[0] == EINVAL
    assert b"seq_profile" in L.ea1_last_error()
    # every truncation of a valid header is rejected or -- if only trailing zero bytes went -- still parses the same
    payload = wr.write_sequence_header(h)
    for n in range(len(payload)):
        rc, got = obu.parse_sequence_header(wr.obu(1, payload[:n]))
        assert rc == EINVAL or (rc == 0 and payload[n:].strip(b"\0") == b"")
    # random garbage never crashes
    for _ in range(300):
        junk = bytes(rng.integers(0, 256, int(rng.integers(1, 80)), dtype=np.uint8))
        assert obu.parse_sequence_header(junk)[0] in (0, ENOENT, EINVAL)
