"""The block-level AV1 parser (easyav1_amd/csrc/av1_tile_syntax.h) with ARBITRARY tables:

  round trip   a generator that walks the same syntax with a symbol WRITER (tests/native/av1_streamgen.cpp) produces plain AV1
               temporal units -- sequence header, frame headers from the writer of test_frame_header.py, tile groups -- and the
               records it built while writing; the parser must turn the bytes back into the very same records (blocks,
               transform blocks, coefficients, CDEF indices, restoration units), frame after frame, which also checks the CDF
               adaptation, the saving / loading of contexts with the reference slots and the per-tile contexts.
  validity     what comes out is accepted by the hand-over's validation and reconstructs on the oracle (CPU).

Parity with dav1d is UNPINNED (see the header of av1_tile_syntax.h): these tests prove self-consistency for any tables, not
conformance."""
import numpy as np
import pytest

import easyav1_amd
from tests import av1_frame_header_writer as fw
from tests import av1_obu_writer as ow
from tests import test_frame_header as tfh

KEY, INTER, INTRA_ONLY = 0, 1, 2


@pytest.fixture(scope="module")
def env():
    easyav1_amd.load(build=True)
    from easyav1_amd import av1parse
    from tests import av1_tables_gen as tg
    from tests import streamgen
    tables = av1parse.Tables(tg.make_blob(5))
    return av1parse, streamgen, tables


def block_header(rng, seq, slots, ft, frame_no, **fixed):
    """a random frame header inside the block parser's scope: no segmentation, no quantiser matrices, one frame size"""
    fixed.setdefault("show_frame", 1)
    fixed.setdefault("frame_size_override_flag", 0)
    h = tfh.random_header(rng, seq, slots, ft, frame_no, fixed)
    h.update(segmentation_enabled=0, segmentation_update_map=0, segmentation_temporal_update=0, segmentation_update_data=0, last_active_seg_id=0,
             seg_id_pre_skip=0, using_qmatrix=0, qm_y=0, qm_u=0, qm_v=0)
    h["feature_enabled"], h["feature_data"] = [[0] * 8 for _ in range(8)], [[0] * 8 for _ in range(8)]
    h["coded_feature_value"] = [[0] * 8 for _ in range(8)]
    deltas = any(h[k] for k in ("delta_q_y_dc", "delta_q_u_dc", "delta_q_u_ac", "delta_q_v_dc", "delta_q_v_ac"))
    if not h["coded_lossless"] and h["base_q_idx"] == 0 and not deltas:
        h["base_q_idx"] = int(rng.integers(1, 256))     # it was a segment feature that made the frame lossy
    if "base_q_idx" in fixed:
        h["base_q_idx"] = fixed["base_q_idx"]
    h["film_grain"]["apply_grain"] = 0
    h["tile_size_bytes"] = 4        # the generated tiles have any size
    return h


def make_stream(env, seed, sizes, nframes, seq_kw=None, policy=None, hdr_kw=None):
    """-> (list of temporal units, list of the generator's own parsed frames, list of header dicts)"""
    av1parse, streamgen, tables = env
    rng = np.random.default_rng(seed)
    kw = dict(max_width=sizes[0], max_height=sizes[1], enable_ref_frame_mvs=0, film_grain_params_present=0, enable_superres=0)
    kw.update(seq_kw or {})
    seq = tfh.simple_seq(**kw)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    gen = streamgen.Generator(tables, seq_obu)
    if policy:
        gen.policy(**policy)
    slots = fw.Slots()
    units, frames, headers = [], [], []
    for n in range(nframes):
        ft = KEY if n == 0 else INTER
        h = block_header(rng, seq, slots, ft, n, **(hdr_kw or {}))
        if ft == KEY:
            h["refresh_frame_flags"] = 0xFF
        elif h["refresh_frame_flags"] == 0:
            h["refresh_frame_flags"] = 1 << int(rng.integers(8))
        hdr_bytes, nbits = fw.write_frame_header(seq, h, slots)
        rc, tiles, frame = gen.frame(hdr_bytes, seed * 100 + n)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        tu = ow.obu(2, b"") + (seq_obu if n == 0 else b"") + streamgen.frame_obu(h, hdr_bytes, tiles)
        units.append(tu)
        frames.append(frame)
        headers.append(h)
        slots.refresh(h)
    counts = gen.element_counts()
    gen.close()
    return units, frames, headers, counts


def same_frame(a, b):
    assert bytes(a.desc) == bytes(b.desc)
    for name in ("blocks", "txs", "coefs", "aux", "cdef_idx"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), name
    for pl in range(3):
        assert a.lr[pl].tobytes() == b.lr[pl].tobytes(), ("lr", pl)


@pytest.mark.parametrize("seed,size", [(1, (64, 64)), (2, (128, 72)), (3, (200, 136)), (4, (72, 200))])
def test_round_trip_of_generated_streams(env, seed, size):
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, size, 5)
    parser = av1parse.FrameParser(tables)
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        assert info.show_frame == 1
    parser.close()
