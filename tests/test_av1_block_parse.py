"""The block-level AV1 parser (easyav1_amd/csrc/av1_tile_syntax.h) with ARBITRARY tables:

  round trip   a generator that walks the same syntax with a symbol WRITER (tests/native/av1_streamgen.cpp) produces plain AV1
               temporal units -- sequence header, frame headers from the writer of test_frame_header.py, tile groups -- and the
               records it built while writing; the parser must turn the bytes back into the very same records (blocks,
               transform blocks, coefficients, CDEF indices, restoration units), frame after frame, which also checks the CDF
               adaptation, the saving / loading of contexts with the reference slots and the per-tile contexts.
  validity     what comes out is accepted by the hand-over's validation and reconstructs on the oracle (CPU).

Conformance is tests/test_real_av1.py's business (libaom streams beside dav1d 1.5.3, round 4); these tests prove self-consistency for any
tables and cover what libaom does not emit (super-resolution, frames of changing size, delta_lf, switch frames)."""
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
    """a random frame header inside the block parser's scope: no quantiser matrices, one frame size (segmentation=False: none of that either)"""
    fixed.setdefault("show_frame", 1)
    fixed.setdefault("frame_size_override_flag", 0)
    h = tfh.random_header(rng, seq, slots, ft, frame_no, fixed)
    if not fixed.get("segmentation", True):
        h.update(segmentation_enabled=0, segmentation_update_map=0, segmentation_temporal_update=0, segmentation_update_data=0, last_active_seg_id=0,
                 seg_id_pre_skip=0)
        h["feature_enabled"], h["feature_data"] = [[0] * 8 for _ in range(8)], [[0] * 8 for _ in range(8)]
        h["coded_feature_value"] = [[0] * 8 for _ in range(8)]
        deltas = any(h[k] for k in ("delta_q_y_dc", "delta_q_u_dc", "delta_q_u_ac", "delta_q_v_dc", "delta_q_v_ac"))
        if not h["coded_lossless"] and h["base_q_idx"] == 0 and not deltas:
            h["base_q_idx"] = int(rng.integers(1, 256))     # it was a segment feature that made the frame lossy
    if fixed.get("base_q_idx"):
        h["base_q_idx"] = fixed["base_q_idx"]           # (the fields a lossless frame leaves out are in the dictionary with their defaults)
        h["coded_lossless"] = h["all_lossless"] = 0
        h["feature_enabled"] = [[0 if j == 0 else v for j, v in enumerate(row)] for row in h["feature_enabled"]]   # no segment may end up lossless
    if fixed.get("one_tile"):     # what an encoder does with a frame that fits one tile (up to 4096 columns and the area of 4096 x 2304)
        h.update(uniform_tile_spacing_flag=1, tile_cols_log2=0, tile_rows_log2=0, tile_cols=1, tile_rows=1, context_update_tile_id=0)
        h["mi_col_starts"], h["mi_row_starts"] = [0, h["mi_cols"]], [0, h["mi_rows"]]
    if fixed.get("lf_levels") and not (h["coded_lossless"] or h["allow_intrabc"]):
        # the two luma frame levels as given (e.g. 0 and non-0) with per-block deltas on top: a block's level may rise above a frame level of 0
        h["loop_filter_level"][0], h["loop_filter_level"][1] = fixed["lf_levels"]
        h["loop_filter_level"][2], h["loop_filter_level"][3] = 9, 17
        if h["base_q_idx"] == 0:
            h["base_q_idx"] = 60
        h.update(delta_q_present=1, delta_q_res=1, delta_lf_present=1, delta_lf_res=2, delta_lf_multi=int(fixed.get("lf_multi", 0)))
    if not fixed.get("grain"):
        h["film_grain"]["apply_grain"] = 0
    h["tile_size_bytes"] = 4        # the generated tiles have any size
    return h


def references_within_limits(h, slots):
    """AV1 spec 5.9.7 / 7.20: a frame may be at most twice as small and sixteen times as large as each reference it names (per direction, on the upscaled width);
    a decoder answers a frame outside it with an error (dav1d: EINVAL)"""
    if h["frame_type"] not in (INTER, tfh.SWITCH):
        return True
    for idx in h["ref_frame_idx"]:
        r = slots.s[idx]
        if r is not None and (2 * h["frame_width"] < r["upscaled_width"] or 2 * h["frame_height"] < r["frame_height"] or
                              h["frame_width"] > 16 * r["upscaled_width"] or h["frame_height"] > 16 * r["frame_height"]):
            return False
    return True


def make_stream(env, seed, sizes, nframes, seq_kw=None, policy=None, hdr_kw=None, gop=None, switch_every=None, split_obus=False, hdr_adjust=None, legal_scaling=False):
    """-> (list of temporal units, list of the generator's own parsed frames, list of header dicts); hdr_adjust(h): last word on every random header"""
    av1parse, streamgen, tables = env
    rng = np.random.default_rng(seed)
    kw = dict(max_width=sizes[0], max_height=sizes[1], enable_ref_frame_mvs=1, film_grain_params_present=0, enable_superres=0,
              seq_force_screen_content_tools=2)   # screen content tools (palette) chosen per frame
    kw.update(seq_kw or {})
    seq = tfh.simple_seq(**kw)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    gen = streamgen.Generator(tables, seq_obu)
    if policy:
        gen.policy(**policy)
    slots = fw.Slots()
    units, frames, headers = [], [], []
    for n in range(nframes):
        ft = KEY if n == 0 or (gop and n % gop == 0) else (INTRA_ONLY if rng.integers(7) == 0 else INTER)
        kw_n = dict(hdr_kw or {})
        if switch_every and ft != KEY and n % switch_every == switch_every - 1:
            # a switch frame: an inter frame that is error resilient, refreshes every slot and carries its own size
            ft = tfh.SWITCH
            kw_n.setdefault("frame_size_range", (sizes[0] * 5 // 8, sizes[1] * 5 // 8))
        h = block_header(rng, seq, slots, ft, n, **kw_n)
        for _ in range(40 if legal_scaling else 0):   # (streams for a decoder that checks conformance: another draw until the frame's size suits its references)
            if references_within_limits(h, slots):
                break
            h = block_header(rng, seq, slots, ft, n, **kw_n)
        if hdr_adjust:
            hdr_adjust(h)
        if ft == KEY:
            h["refresh_frame_flags"] = 0xFF
        elif h["refresh_frame_flags"] == 0:
            h["refresh_frame_flags"] = 1 << int(rng.integers(8))
        hdr_bytes, nbits = fw.write_frame_header(seq, h, slots)
        rc, tiles, frame = gen.frame(hdr_bytes, seed * 100 + n)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        if split_obus:   # OBU_FRAME_HEADER + several OBU_TILE_GROUPs instead of one OBU_FRAME
            body, ngroups = streamgen.frame_obus_split(h, hdr_bytes, nbits, tiles, rng)
        else:
            body = streamgen.frame_obu(h, hdr_bytes, tiles)
        tu = ow.obu(2, b"") + (seq_obu if n == 0 else b"") + body
        units.append(tu)
        frames.append(frame)
        headers.append(h)
        slots.refresh(h)
    counts = gen.element_counts() + [gen.temporal_candidates()]
    gen.close()
    return units, frames, headers, counts


def same_frame(a, b):
    assert bytes(a.desc) == bytes(b.desc)
    for name in ("blocks", "txs", "coefs", "aux", "cdef_idx"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), name
    for pl in range(3):
        assert a.lr[pl].tobytes() == b.lr[pl].tobytes(), ("lr", pl)


@pytest.mark.parametrize("seed,size", [(1, (64, 64)), (2, (128, 72)), (3, (200, 136)), (4, (72, 200)), (5, (168, 104)), (6, (232, 152))])
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


@pytest.mark.parametrize("seed,size", [(31, (200, 136)), (32, (296, 216)), (33, (392, 200))])
def test_round_trip_with_the_tiles_of_a_frame_parsed_side_by_side(env, seed, size):
    """ea1_av1_frame_parser_set_threads(3): random tile layouts (up to 255 tiles), palettes and warp matrices in the byte arena, contexts saved from
    the context tile -- the records must be the generator's, which writes tile after tile"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, size, 6)
    assert max(h["tile_cols"] * h["tile_rows"] for h in headers) > 1
    parser = av1parse.FrameParser(tables)
    parser.set_threads(3)
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
    parser.close()


LF_ZERO_NONZERO = [dict(hdr_kw=dict(lf_levels=(0, 29), segmentation=False)), dict(hdr_kw=dict(lf_levels=(41, 0), lf_multi=1, segmentation=False)),
                   dict(hdr_kw=dict(lf_levels=(0, 13), lf_multi=1, segmentation=False))]


@pytest.mark.parametrize("seed,kw", [(801, LF_ZERO_NONZERO[0]), (802, LF_ZERO_NONZERO[1]), (803, LF_ZERO_NONZERO[2])])
def test_one_luma_frame_level_zero_with_per_block_deltas(env, seed, kw):
    """round-3 audit finding 3, untested until now: with delta_lf a block's level may rise above a frame level of 0, so only BOTH luma
    levels 0 switch deblocking off (dav1d calc_lf_values).  Generated streams with frame levels (0, non-0) / (non-0, 0) and single / multi deltas:
    the parser reproduces the generator's records, blocks really carry non-zero levels in the direction whose frame level is 0, and the oracle
    filters there (the picture differs from the one with deblocking off).  libaom does not emit delta_lf through the interface at hand
    (`deltalf-mode` is refused), so this stays a self-consistency test."""
    from easyav1_amd import ir
    from tests import oracle_lib as ol
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (200, 136), 4, **kw)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    zero_dir = 0 if kw["hdr_kw"]["lf_levels"][0] == 0 else 1
    hit = 0
    slots = [None] * ir.NUM_REF_SLOTS
    for f, h in zip(parsed, headers):
        if h["coded_lossless"] or h["allow_intrabc"]:
            slots[f.desc.dst_slot] = ol.oracle_decode_frame(f, slots)["final"]
            continue
        assert f.desc.lf_per_block and f.desc.lf_level[zero_dir] == 0 and f.desc.lf_level[1 - zero_dir] != 0
        raised = int((f.blocks["lf_level"][:, zero_dir] > 0).sum())
        out = ol.oracle_decode_frame(f, slots, ir.STAGE_RECON | ir.STAGE_DEBLOCK)
        if raised:
            # the same frame with that direction's levels cleared: the deblocked picture must differ, i.e. the pass DID run
            g = ir.ParsedFrame(f.desc.copy(), f.blocks.copy(), f.txs, f.coefs, f.aux, f.cdef_idx, f.lr)
            g.blocks["lf_level"][:, zero_dir] = 0
            off = ol.oracle_decode_frame(g, slots, ir.STAGE_RECON | ir.STAGE_DEBLOCK)
            hit += int(any(not np.array_equal(a, b) for a, b in zip(out["final"], off["final"])))
        slots[f.desc.dst_slot] = ol.oracle_decode_frame(f, slots)["final"]
    assert hit >= 1


def blocks_by_size(frames):
    import collections
    hist = collections.Counter()
    for f in frames:
        for b in f.blocks[:f.desc.num_blocks]:
            hist[(1 << int(b["bwl"]), 1 << int(b["bhl"]))] += 1
    return hist


SB128 = dict(seq_kw=dict(use_128x128_superblock=1), policy=dict(split_bias=10))
# sequences with tools switched off: the syntax elements that depend on them must not be read
NO_ORDER_HINT = dict(seq_kw=dict(enable_order_hint=0, enable_jnt_comp=0, enable_ref_frame_mvs=0, order_hint_bits=0))
TOOLS_OFF = dict(seq_kw=dict(enable_dual_filter=0, enable_warped_motion=0, enable_interintra_compound=0, enable_masked_compound=0, enable_jnt_comp=0,
                             enable_filter_intra=0, enable_intra_edge_filter=0, enable_cdef=0, enable_restoration=0, enable_ref_frame_mvs=0))
INTEGER_MV = dict(seq_kw=dict(seq_force_integer_mv=1))


@pytest.mark.parametrize("seed,kw", [(301, NO_ORDER_HINT), (302, TOOLS_OFF), (303, INTEGER_MV)])
def test_round_trip_with_sequence_tools_switched_off(env, seed, kw):
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (200, 136), 6, **kw)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    assert len({p[0].tobytes() for p in oracle_pictures(parsed)}) > 1


@pytest.mark.parametrize("seed,size", [(71, (256, 128)), (72, (200, 136)), (73, (320, 264))])
def test_round_trip_with_128x128_superblocks_and_128_sample_blocks(env, seed, size):
    """superblocks of 128x128: blocks of 128x128, 128x64 and 64x128 (coded in chunks of 64x64 luma samples, every plane of a chunk before
    the next; CDEF index and transform sizes per 64x64 area), the hand-over's frame padded to whole superblocks (up to 15 units of 8);
    the oracle reconstructs what comes out"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, size, 5, **SB128)
    hist = blocks_by_size(frames)
    assert hist[(128, 128)] and hist[(128, 64)] and hist[(64, 128)], hist
    assert frames[0].desc.width % 128 == 0 and frames[0].desc.height % 128 == 0
    if size == (200, 136):
        assert frames[0].desc.pad == 0xF7     # 256 - 200 = 7 units, 256 - 136 = 15 units
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    pics = oracle_pictures(parsed)
    assert len({p[0].tobytes() for p in pics}) > 1


SUPERRES = dict(seq_kw=dict(enable_superres=1))
MONO = dict(seq_kw=dict(monochrome=1, layout=0))


@pytest.mark.parametrize("seed", [501, 502])
def test_round_trip_of_monochrome_streams(env, seed):
    """mono_chrome (4:0:0): no chroma syntax at all (modes, palette, coefficients, two loop filter deltas less); the hand-over still lists
    the chroma transform blocks, without coefficients, so that the frame goes through the 4:2:0 reconstruction with chroma planes nothing
    is coded into (they are not shown)"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (200, 136), 6, **MONO)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
        t = got.txs[:got.desc.num_tx]
        assert int((t["plane"] > 0).sum()) > 0 and not np.any((t["plane"] > 0) & (t["coef_off"] != 0xFFFFFFFF))
        b = got.blocks[:got.desc.num_blocks]
        assert not np.any(b["flags"] & 0x0400) and np.all(b["pal_size"][:, 1] == 0)     # no chroma palette
    parser.close()
    pics = oracle_pictures(parsed)
    assert len({p[0].tobytes() for p in pics}) > 1


def test_upscaling_known_answers():
    """7.16 on its own: positions (with taps that only pick the centre sample the output is the source at floor((x + 0.5) down / up - 0.5
    + 1/128), derived by hand for 2 : 1 and 9 : 8), a constant stays constant with any taps that sum to 128, and an independent
    restatement of the fixed-point arithmetic"""
    import ctypes
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    L = ol.oracle()
    L.ea1o_upscale_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.ea1o_upscale_rows.restype = None

    def run(src, up_w, taps, last_x=None):
        ol.oracle_set_upscale_filter(taps)
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((src.shape[0], up_w), np.uint8)
        L.ea1o_upscale_rows(src.ctypes.data, src.shape[1], src.shape[1], src.shape[1] - 1 if last_x is None else last_x, src.shape[0], dst.ctypes.data, up_w, up_w)
        return dst

    pick = np.zeros((64, 8), np.int16)
    pick[:, 3] = 128          # tap 3 is the sample at floor(position)
    ramp = np.arange(32, dtype=np.uint8)[None, :] * 5
    # 2 : 1 -- output x sits at (x + 0.5) / 2 - 0.5 = x / 2 - 0.25 (+ 1/128): floor = -1, 0, 0, 1, 1, 2, ... (the first clamps to 0)
    out = run(ramp, 64, pick)
    assert out[0].tolist() == [5 * max(0, (2 * x - 1) // 4) for x in range(64)]
    # 9 : 8 -- 27 outputs from 24 samples: position (x + 0.5) 8 / 9 - 0.5
    out = run(ramp[:, :24], 27, pick)
    want = [int(np.floor((x + 0.5) * 24 / 27 - 0.5 + 1 / 128)) for x in range(27)]
    assert out[0].tolist() == [5 * min(23, max(0, v)) for v in want]
    # any taps that sum to 128 keep a constant
    assert np.all(run(np.full((3, 40), 77, np.uint8), 73, tg.upscale_filter()) == 77)
    # the arithmetic restated: step, error, initial position, phase, Round2(., 7), edge clamp at last_x
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (2, 37), dtype=np.uint8)
    taps = tg.upscale_filter()
    for up_w in (41, 50, 61, 74):
        got = run(src, up_w, taps)
        down = 37
        step = ((down << 14) + up_w // 2) // up_w
        err = up_w * step - (down << 14)
        num = -((up_w - down) << 13) + up_w // 2
        initial = (int(np.trunc(num / up_w)) + 128 - int(np.trunc(err / 2))) & 16383
        for x in range(up_w):
            sx = -(1 << 14) + initial + x * step
            px, ph = sx >> 14, (sx & 16383) >> 8
            acc = sum(int(taps[ph][k]) * int(src[1, min(36, max(0, px + k - 3))]) for k in range(8))
            assert got[1, x] == min(255, max(0, (acc + 64) >> 7)), (up_w, x)
    ol.oracle_set_upscale_filter(None)


@pytest.mark.parametrize("seed", [401, 402, 403])
def test_round_trip_of_streams_with_super_resolution(env, seed):
    """use_superres: the frame is coded at the downscaled width (restoration units counted on the upscaled one, 5.11.57), references are
    kept upscaled (is_scaled() compares their width with the downscaled one); the oracle reconstructs, upscales both filtered pictures and
    restores the upscaled pair"""
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (264, 200), 6, **SUPERRES)
    assert any(f.desc.upscaled_width for f in frames)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    ol.oracle_set_upscale_filter(tg.upscale_filter())
    try:
        pics = oracle_pictures(parsed)
    finally:
        ol.oracle_set_upscale_filter(None)
    for f, p in zip(frames, pics):
        assert p[0].shape[1] == (((f.desc.upscaled_width + 7) & ~7) if f.desc.upscaled_width else f.desc.width)


RESIZED = dict(hdr_kw=dict(frame_size_override_flag=1, frame_size_range=(150, 110)))   # within 2 : 1 of the largest frame (264 x 200)


@pytest.mark.parametrize("seed", [801, 802, 803])
def test_frames_as_a_header_obu_and_several_tile_groups(env, seed):
    """OBU_FRAME_HEADER + OBU_TILE_GROUPs with tile ranges (tile_start_and_end_present_flag) instead of one OBU_FRAME: the same records"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (264, 200), 5, split_obus=True)
    assert any(h["tile_cols"] * h["tile_rows"] > 2 for h in headers)
    parser = av1parse.FrameParser(tables)
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
    parser.close()


def test_packets_without_a_temporal_delimiter_and_with_other_obus(env):
    """a Matroska block carries no temporal delimiter; metadata (ITU-T T.35) and padding OBUs beside the frame are passed over"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, 901, (200, 136), 4)
    td = ow.obu(2, b"")
    parser = av1parse.FrameParser(tables)
    for n, (tu, want) in enumerate(zip(units, frames)):
        assert tu.startswith(td)
        rc, got, info = parser.parse(ow.obu(5, bytes([4, 0xB5, 0x00, 0x3C, 0x80])) + tu[len(td):] + ow.obu(15, b"\x00\x00\x80"))
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
    parser.close()


@pytest.mark.parametrize("seed", [601, 602])
def test_round_trip_of_streams_with_switch_frames(env, seed):
    """S-frames: inter frames with error_resilient_mode, no primary reference (default CDFs, default deltas), all slots refreshed, a size of
    their own (frame_size_override_flag without found_ref)"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (264, 200), 7, switch_every=3)
    assert sum(1 for h in headers if h["frame_type"] == tfh.SWITCH) >= 2
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    oracle_pictures(parsed)


@pytest.mark.parametrize("seed", [211, 212, 213])
def test_round_trip_of_streams_that_change_their_frame_size(env, seed):
    """frame_size_override_flag: every frame has its own size (or takes a reference's, found_ref), predictions come from references of
    other sizes (7.11.3.3), motion vector projection and the segment map only use references of the frame's own size in 4x4 units;
    the oracle reconstructs what comes out (every reference with its own size, crop and padding)"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (264, 200), 6, **RESIZED)
    assert len({(f.desc.width, f.desc.height, f.desc.pad) for f in frames}) > 1
    parser = av1parse.FrameParser(tables)
    parsed = []
    for n, (tu, want) in enumerate(zip(units, frames)):
        rc, got, info = parser.parse(tu)
        assert rc == 0, (n, easyav1_amd._lib.last_error())
        same_frame(got, want)
        parsed.append(got)
    parser.close()
    pics = oracle_pictures(parsed)
    assert [p[0].shape for p in pics] == [(f.desc.height, f.desc.width) for f in frames]


def coded_part(desc, planes):
    """the planes cut to the frame's TRUE coded size (beyond it -- the padding of ea1_frame_desc.pad -- a picture is undefined)"""
    w, h = desc.width - 8 * (desc.pad & 15), desc.height - 8 * (desc.pad >> 4)
    return [planes[0][:h, :w], planes[1][:h // 2, :w // 2], planes[2][:h // 2, :w // 2]]


def oracle_pictures(frames):
    """reconstruct parsed frames in order on the oracle (pictures kept per destination slot) -> list of final pictures"""
    from easyav1_amd import ir
    from tests import oracle_lib as ol
    slots, crops, out = [None] * 8, [0] * 8, []
    for f in frames:
        pic = ol.oracle_decode_frame(f, slots, ir.STAGE_ALL, ref_crops=crops)["final"]   # (per slot: crop | pad << 8 of the frame that made its picture)
        slots[f.desc.dst_slot] = pic
        crops[f.desc.dst_slot] = f.desc.crop | (f.desc.pad << 8)
        if f.desc.upscaled_width:    # kept upscaled: upscaled_width visible samples in whole 8-sample units, no padding to the right
            up = f.desc.upscaled_width
            crops[f.desc.dst_slot] = (f.desc.crop & 0xF0) | (((up + 7) & ~7) - up) | ((f.desc.pad & 0xF0) << 8)
        out.append(pic)
    return out


def visible_part(desc, planes):
    """the planes cut to the picture that is shown (super-resolution: upscaled_width samples wide)"""
    w = desc.upscaled_width if desc.upscaled_width else desc.width - 8 * (desc.pad & 15) - (desc.crop & 15)
    h = desc.height - 8 * (desc.pad >> 4) - (desc.crop >> 4)
    return [planes[0][:h, :w], planes[1][:(h + 1) // 2, :(w + 1) // 2], planes[2][:(h + 1) // 2, :(w + 1) // 2]]


def test_every_syntax_element_and_the_motion_field_are_exercised(env):
    av1parse, streamgen, tables = env
    total = None
    for seed in range(100, 112):
        units, frames, headers, counts = make_stream(env, seed, [(128, 72), (200, 136)][seed & 1], 6)
        total = counts if total is None else [a + b for a, b in zip(total, counts)]
        parser = av1parse.FrameParser(tables)
        for tu, want in zip(units, frames):
            rc, got, info = parser.parse(tu)
            assert rc == 0, easyav1_amd._lib.last_error()
            same_frame(got, want)
        parser.close()
    assert all(c > 0 for c in total), total   # every element of the syntax (enum Elem) + temporal motion vector candidates


@pytest.mark.parametrize("seed,size", [(11, (128, 72)), (12, (200, 136)), (13, (168, 104))])
def test_generated_streams_reconstruct_on_the_oracle(env, seed, size):
    """what the parser hands over is a coherent frame: the oracle reconstructs it (its own checks abort on an impossible record),
    and the pictures are not trivial (the residual reaches them)"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, size, 4)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0
        parsed.append(f)
    parser.close()
    pics = oracle_pictures(parsed)
    assert all(p[0].shape == (frames[0].desc.height, frames[0].desc.width) for p in pics)
    assert len({p[0].tobytes() for p in pics}) > 1


def test_refused_and_malformed_input(env):
    """what lies outside the parser's scope is refused with EA1_ENOTSUP and a reason; truncated tile data never crashes"""
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, 21, (128, 72), 3)
    # a 10-bit sequence: the reconstruction path is 8-bit 4:2:0
    rng = np.random.default_rng(5)
    seq = tfh.simple_seq(max_width=128, max_height=72, enable_ref_frame_mvs=1, film_grain_params_present=0, bit_depth=10, hbd=1)
    slots = fw.Slots()
    h = block_header(rng, seq, slots, KEY, 0)
    h["refresh_frame_flags"] = 0xFF
    hdr_bytes, nbits = fw.write_frame_header(seq, h, slots)
    tu = ow.obu(2, b"") + ow.obu(1, ow.write_sequence_header(seq)) + ow.obu(6, hdr_bytes + b"\x00" * 16)
    p = av1parse.FrameParser(tables)
    rc, f, info = p.parse(tu)
    assert rc == -8 and "8-bit 4:2:0" in easyav1_amd._lib.last_error(), (rc, easyav1_amd._lib.last_error())
    p.close()
    # truncations and bit flips of a good stream: an error or a (different) frame, never a crash
    good = units[0]
    for cut in range(len(good) - 40, len(good), 3):
        p = av1parse.FrameParser(tables)
        p.parse(good[:cut])
        p.close()
    for k in range(60):
        bad = bytearray(good)
        pos = int(rng.integers(len(good) // 2, len(good)))
        bad[pos] ^= 1 << int(rng.integers(8))
        p = av1parse.FrameParser(tables)
        rc, f, info = p.parse(bytes(bad))
        assert rc in (0, -1, -8), rc
        p.close()


def test_table_blob_checks(env):
    av1parse, streamgen, tables = env
    from tests import av1_tables_gen as tg
    blob = tg.make_blob(3)
    secs = av1parse.blob_sections()
    assert {s["kind"] for s in secs} == {0, 1, 2, 3, 4, 5} and sum(s["count"] for s in secs) * 2 + 16 == av1parse.blob_size()
    names = [s["name"] for s in secs]
    assert "CoeffBase_q3" in names and "Default_Scan_32x32" in names and "Mcol_Scan_16x4" in names and "Ac_Qlookup_8bit" in names
    # the super-resolution taps: a row that does not sum to 128 is refused
    bad = blob.copy()
    s5 = next(x for x in secs if x["kind"] == 5)
    bad.view(np.uint16)[s5["offset"] // 2 + 8 * 17] += 1
    with pytest.raises(Exception):
        av1parse.Tables(bad)
    for kind, poke in ((0, 40000), (1, 5000), (2, 0)):
        s = next(x for x in secs if x["kind"] == kind)
        bad = blob.copy()
        bad.view(np.uint16)[s["offset"] // 2 + 1] = poke
        with pytest.raises(Exception):
            av1parse.Tables(bad)
    with pytest.raises(Exception):
        av1parse.Tables(blob[:-2])


def make_hidden_frame_stream(env, seed, size, hidden_key=False, superres=0):
    """key frame; then temporal units of [hidden frame H into slot 5 (show_frame = 0, showable), shown inter frame S], a plain inter frame,
    and a unit that holds only a show_existing_frame header for slot 5.  -> (units, decoded frames in order, what each unit shows:
    index into the decoded frames)"""
    av1parse, streamgen, tables = env
    rng = np.random.default_rng(seed)
    seq = tfh.simple_seq(max_width=size[0], max_height=size[1], enable_ref_frame_mvs=1, film_grain_params_present=0, enable_superres=superres)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    gen = streamgen.Generator(tables, seq_obu)
    slots = fw.Slots()
    units, frames, shows = [], [], []

    def coded(ft, n, **kw):
        h = block_header(rng, seq, slots, ft, n, **kw)
        if ft == KEY and h["show_frame"]:
            h["refresh_frame_flags"] = 0xFF
        if not h["show_frame"]:
            h["showable_frame"] = 1
        hdr_bytes, nbits = fw.write_frame_header(seq, h, slots)
        rc, tiles, frame = gen.frame(hdr_bytes, seed * 100 + n)
        assert rc == 0, easyav1_amd._lib.last_error()
        slots.refresh(h)
        frames.append(frame)
        return streamgen.frame_obu(h, hdr_bytes, tiles)
    units.append(ow.obu(2, b"") + seq_obu + coded(KEY, 0))
    shows.append(0)
    n = 1
    for rep in range(2):
        hidden = coded(KEY if hidden_key else INTER, n, show_frame=0, refresh_frame_flags=1 << 5, error_resilient_mode=1 if hidden_key else 0)
        # (the hidden frame must stay showable: the writer emitted the flag it drew -- read it back from the slot mirror)
        hidden_index = len(frames) - 1
        showable = slots.s[5]["showable_frame"]
        assert showable
        shown = coded(INTER, n + 1, refresh_frame_flags=1 << (rep + 1))
        units.append(ow.obu(2, b"") + hidden + shown)
        shows.append(len(frames) - 1)
        units.append(ow.obu(2, b"") + coded(INTER, n + 2, refresh_frame_flags=1 << 3))
        shows.append(len(frames) - 1)
        if showable:
            w = fw.Bits()
            w.f(1, 1)
            w.f(3, 5)
            while len(w.bits) % 8:
                w.bits.append(0)
            hdr = w.bytes()
            assert gen.show_existing(hdr) == 0, easyav1_amd._lib.last_error()
            if hidden_key:   # 7.21: the shown key frame takes every slot
                for i in range(8):
                    slots.s[i] = slots.s[5]
            units.append(ow.obu(2, b"") + ow.obu(3, hdr))
            shows.append(hidden_index)
        n += 3
    gen.close()
    return units, frames, shows


@pytest.mark.parametrize("hidden_key", [False, True])
def test_hidden_frames_and_show_existing_frame(env, hidden_key):
    """temporal units with a hidden frame in front of the shown one, and show_existing_frame: every frame parses like the generator's
    records, the unit shows the right picture"""
    av1parse, streamgen, tables = env
    units, frames, shows = make_hidden_frame_stream(env, 51 + hidden_key, (128, 72), hidden_key)
    parser = av1parse.FrameParser(tables)
    k = 0
    decoded = []
    for tu, show in zip(units, shows):
        got = parser.parse_unit(tu)
        assert got, "no frame in the unit"
        shown = None
        for f, info in got:
            if info.show_existing:
                assert f is None
                shown = ("slot", info.frame.desc.dst_slot)
            else:
                same_frame(f, frames[k])
                decoded.append(f)
                if info.show_frame:
                    shown = ("frame", k)
                k += 1
        assert shown is not None
        if shown[0] == "frame":
            assert shown[1] == show
        else:   # the picture that slot holds is the hidden frame's
            assert frames[show].desc.dst_slot == shown[1]
    assert k == len(frames)
    parser.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size,kw", [(31, (128, 72), {}), (32, (200, 136), {}), (33, (320, 192), {}), (34, (168, 104), {}), (35, (232, 152), {}), (36, (104, 168), {}),
                                          (37, (296, 216), {}), (74, (256, 128), SB128), (75, (200, 136), SB128), (76, (320, 264), SB128), (77, (392, 200), SB128),
                                          (301, (200, 136), NO_ORDER_HINT), (302, (200, 136), TOOLS_OFF), (303, (200, 136), INTEGER_MV),
                                          (501, (200, 136), MONO), (502, (264, 200), MONO),
                                          (801, (200, 136), LF_ZERO_NONZERO[0]), (802, (200, 136), LF_ZERO_NONZERO[1]), (803, (200, 136), LF_ZERO_NONZERO[2])])
def test_generated_streams_reconstruct_on_the_gpu_like_the_oracle(env, seed, size, kw):
    """tile bytes -> block parser -> hand-over -> HIP reconstruction == the oracle on the same records (the last four: superblocks of
    128x128 with blocks of 128 samples)"""
    from easyav1_amd import video
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, size, 5, **kw)
    if kw is SB128:
        hist = blocks_by_size(frames)
        assert hist[(128, 128)] + hist[(128, 64)] + hist[(64, 128)] > 3, hist
    parser = av1parse.FrameParser(tables)
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0
        parsed.append(f)
    parser.close()
    want = oracle_pictures(parsed)
    ctx = video.ReconContext(parsed[0].desc.width, parsed[0].desc.height)
    for n, f in enumerate(parsed):
        job = ctx.make_job(f)
        ctx.submit(job)
        got = coded_part(f.desc, ctx.download(f.desc.dst_slot))
        job.close()
        exp = coded_part(f.desc, want[n])
        for pl in range(3):
            assert np.array_equal(got[pl], exp[pl]), (n, pl)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [211, 212, 213, 214, 215, 601, 602])
def test_streams_that_change_their_frame_size_reconstruct_on_the_gpu_like_the_oracle(env, seed):
    """every frame with its own size, references of other sizes (scaled prediction): block parser -> hand-over -> HIP == the oracle
    (seeds from 601: streams with switch frames instead)"""
    from easyav1_amd import video
    av1parse, streamgen, tables = env
    if seed >= 600:
        units, frames, headers, counts = make_stream(env, seed, (264, 200), 7, switch_every=3)
    else:
        units, frames, headers, counts = make_stream(env, seed, (264, 200), 6, **RESIZED)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0
        parsed.append(f)
    parser.close()
    want = oracle_pictures(parsed)
    ctx = video.ReconContext(max(f.desc.width for f in parsed), max(f.desc.height for f in parsed))
    for n, f in enumerate(parsed):
        job = ctx.make_job(f)
        ctx.submit(job)
        got = coded_part(f.desc, ctx.download(f.desc.dst_slot))
        job.close()
        exp = coded_part(f.desc, want[n])
        for pl in range(3):
            assert got[pl].shape == exp[pl].shape and np.array_equal(got[pl], exp[pl]), (n, pl)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [401, 402, 403, 404, 405, 406])
def test_streams_with_super_resolution_reconstruct_on_the_gpu_like_the_oracle(env, seed):
    """use_superres: reconstruction, deblocking and CDEF at the downscaled width, both pictures upscaled (7.16), restoration on the
    upscaled pair, the reference kept upscaled and predicted from with the scaled path: block parser -> hand-over -> HIP == the oracle"""
    from easyav1_amd import video
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    av1parse, streamgen, tables = env
    units, frames, headers, counts = make_stream(env, seed, (264, 200), 6, **SUPERRES)
    parser = av1parse.FrameParser(tables)
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0
        parsed.append(f)
    parser.close()
    ol.oracle_set_upscale_filter(tg.upscale_filter())
    try:
        want = oracle_pictures(parsed)
    finally:
        ol.oracle_set_upscale_filter(None)
    ctx = video.ReconContext(max(max(f.desc.width, (f.desc.upscaled_width + 7) & ~7) for f in parsed), max(f.desc.height for f in parsed))
    ctx.set_upscale_filter(tg.upscale_filter())
    # the tile-column split refuses such frames (EA1_ENOTSUP): a strip of the downscaled frame is not a strip of the upscaled picture
    up = next(f for f in parsed if f.desc.upscaled_width)
    with pytest.raises(easyav1_amd._lib.Ea1Error) as err:
        video.Job.strip(ctx, up, 0, 64)
    assert err.value.code == -8
    for n, f in enumerate(parsed):
        job = ctx.make_job(f)
        ctx.submit(job)
        got = visible_part(f.desc, ctx.download(f.desc.dst_slot))
        job.close()
        exp = visible_part(f.desc, want[n])
        for pl in range(3):
            assert got[pl].shape == exp[pl].shape and np.array_equal(got[pl], exp[pl]), (n, pl, f.desc.upscaled_width)
    ctx.close()


@pytest.mark.gpu
def test_a_monochrome_plain_av1_file_through_the_public_interface(env, tmp_path, monkeypatch):
    """4:0:0: the pictures come out with the luma plane only (pixel layout YUV400, data[1] = data[2] = NULL as dav1d hands them on)"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N, G = 200, 136, 8, 4
    units, frames, headers, counts = make_stream(env, 503, (W, H), N, gop=G, **MONO)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0, monochrome=1, layout=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k % G == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    monkeypatch.setenv("EASYAV1_AMD_GOPS", "2")
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        assert ts == 40 * shown and props.pixel_layout == 1 and planes[1] is None and planes[2] is None
        assert planes[0].shape == (H, W) and np.array_equal(planes[0], want[shown][0][:H, :W]), shown
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
def test_a_plain_av1_file_with_super_resolution_through_the_public_interface(env, tmp_path, monkeypatch):
    """the same kind of stream as a WebM file: the Upscale_Filter comes with the table blob, every picture comes out upscaled_width wide"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N, G = 264, 200, 12, 4
    units, frames, headers, counts = make_stream(env, 407, (W, H), N, gop=G, **SUPERRES)
    assert sum(1 for f in frames if f.desc.upscaled_width) >= 3
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0, enable_superres=1)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k % G == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    ol.oracle_set_upscale_filter(tg.upscale_filter())
    try:
        want = oracle_pictures(frames)
    finally:
        ol.oracle_set_upscale_filter(None)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    monkeypatch.setenv("EASYAV1_AMD_GOPS", "3")
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        h = headers[shown]
        assert ts == 40 * shown and (props.width, props.height) == (h["upscaled_width"], h["frame_height"]), (shown, props.width, props.height)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
def test_a_plain_av1_file_that_changes_its_frame_size_through_the_public_interface(env, tmp_path, monkeypatch):
    """the same kind of stream as a WebM file, closed GOPs on several chains: every picture comes out with its frame's own size"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N, G = 264, 200, 12, 4
    units, frames, headers, counts = make_stream(env, 216, (W, H), N, gop=G, **RESIZED)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k % G == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    monkeypatch.setenv("EASYAV1_AMD_GOPS", "3")
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        h = headers[shown]
        assert ts == 40 * shown and (props.width, props.height) == (h["upscaled_width"], h["frame_height"]), (shown, props.width, props.height)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
def test_plain_av1_webm_through_the_public_interface(env, tmp_path, monkeypatch):
    """a generated plain-AV1 WebM file (no private metadata) comes out of easyav1_get_video_frame bit-exact against the oracle when
    the tables are supplied (EASYAV1_AMD_AV1_TABLES); without them the track yields no pictures and a warning"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N = 192, 136, 6
    units, frames, headers, counts = make_stream(env, 41, (W, H), N)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        assert ts == 40 * shown and (props.width, props.height) == (W, H)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N and p.is_finished()
    p.close()
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES")
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    while p.decode_next() == ea.STATUS_OK:
        assert not p.has_video_frame()
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hidden_key,superres", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_hidden_frames_and_show_existing_through_the_public_interface(env, tmp_path, monkeypatch, hidden_key, superres):
    """packets that hold a hidden frame + the shown one, and packets that are one show_existing_frame header: one picture per packet,
    the right one, bit-exact against the oracle (also with super-resolution: a shown existing frame is the upscaled picture)"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H = 192, 136
    units, frames, shows = make_hidden_frame_stream(env, 61 + hidden_key + 10 * superres, (W, H), hidden_key, superres)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0, enable_superres=superres)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(len(units))]
    data = ww.make_webm(7, [track], blocks, duration=float(len(units) * 40), extras=False)
    ol.oracle_set_upscale_filter(tg.upscale_filter())
    try:
        pics = oracle_pictures(frames)
    finally:
        ol.oracle_set_upscale_filter(None)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame(), shown
        ts, planes, props = p.video_frame()
        assert ts == 40 * shown
        want = pics[shows[shown]]
        fd = frames[shows[shown]].desc
        if fd.upscaled_width:
            assert props.width == fd.upscaled_width, (shown, props.width, fd.upscaled_width)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == len(units) and p.is_finished()
    p.close()


@pytest.mark.gpu
def test_film_grain_of_a_plain_av1_stream_through_the_public_interface(env, tmp_path, monkeypatch):
    """film_grain_params in the frame headers + the Gaussian sequence of the table blob: the pictures easyav1_get_video_frame hands out
    carry the grain (oracle: reconstruction, then oracle/av1_grain.c with the same sequence), the references stay clean"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import oracle_lib as ol
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N = 192, 136, 6
    units, frames, headers, counts = make_stream(env, 71, (W, H), N, seq_kw=dict(film_grain_params_present=1), hdr_kw=dict(grain=True))
    assert sum(h["film_grain"]["apply_grain"] for h in headers) >= 2
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=1)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    # expected: parse (for the grain parameters as the library derives them), reconstruct, grain
    blob = tg.make_blob(5)
    parser = av1parse.FrameParser(tables)
    clean = oracle_pictures(frames)
    want, grained = [], 0
    ol.oracle_set_gaussian_sequence(tg.gaussian_of(blob))
    try:
        for n, tu in enumerate(units):
            rc, f, info = parser.parse(tu)
            assert rc == 0
            if info.has_film_grain:
                want.append(ol.oracle_film_grain(info.film_grain, clean[n]))
                grained += any(not np.array_equal(a, b) for a, b in zip(want[-1], clean[n]))
            else:
                want.append(clean[n])
    finally:
        from tests.test_video_kats import grain_stand_in
        ol.oracle_set_gaussian_sequence(grain_stand_in())
    parser.close()
    assert grained >= 1
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(blob.tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gops,split_obus", [("1", False), ("3", False), ("3", True)])
def test_plain_av1_with_several_gops_in_flight(env, tmp_path, monkeypatch, gops, split_obus):
    """closed GOPs of a plain AV1 stream on several chains: one parser per chain (their symbol parses run side by side), the sequence
    header only in the first packet; pictures in presentation order, bit-exact"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N, G = (264, 200, 12, 3) if split_obus else (128, 72, 12, 3)    # (the larger frames have several tiles: several tile groups)
    units, frames, headers, counts = make_stream(env, 81, (W, H), N, gop=G, split_obus=split_obus)
    if split_obus:   # (and, as a Matroska muxer does, without the temporal delimiters)
        td = ow.obu(2, b"")
        units = [u[len(td):] if u.startswith(td) else u for u in units]
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k % G == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    monkeypatch.setenv("EASYAV1_AMD_GOPS", gops)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        assert ts == 40 * shown
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gops", ["1", "4"])
def test_seeks_in_a_plain_av1_file(env, tmp_path, monkeypatch, gops):
    """seeking in a plain AV1 file (forwards into the middle of a GOP, backwards, to a key frame, beyond the end): the chains' parsers start
    again at the key frame before the target, what is decoded on the way is not shown, the pictures from the target on are the oracle's"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N, G = 128, 72, 16, 4
    units, frames, headers, counts = make_stream(env, 83, (W, H), N, gop=G)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k % G == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    monkeypatch.setenv("EASYAV1_AMD_GOPS", gops)
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    for k in range(3):      # a few pictures first, so that the seek meets a pipeline in flight
        assert p.decode_next() == ea.STATUS_OK and p.has_video_frame()
        p.video_frame()
    for target, first in ((40 * 9 + 10, 10), (40 * 2, 2), (40 * 12, 12), (40 * N + 40, None), (40 * 5 + 39, 6), (0, 0)):
        assert p.seek_to_timestamp(target) == ea.STATUS_OK
        assert not p.has_video_frame()
        if first is None:
            assert p.decode_next() == ea.STATUS_FINISHED
            continue
        for k in range(first, min(first + 5, N)):
            assert p.decode_next() == ea.STATUS_OK and p.has_video_frame(), (target, k)
            ts, planes, props = p.video_frame()
            assert ts == 40 * k, (target, k, ts)
            for pl in range(3):
                assert np.array_equal(planes[pl], want[k][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (target, k, pl)
    p.close()


def streams_with(env, want, count, size=(200, 136)):
    """generated streams (key frame every other frame) whose frames hold blocks that satisfy `want(blocks)`"""
    found = []
    for seed in range(300, 400):
        units, frames, headers, counts = make_stream(env, seed, size, 4, gop=2)
        if any(want(f.blocks) for f in frames):
            found.append((units, frames))
            if len(found) == count:
                break
    assert len(found) == count
    return found


def narrow_palette_blocks(b):
    """blocks of 4x16 / 16x4 samples with a palette (the chroma palette of one belongs to the chroma block of the whole 8-sample-wide area)"""
    from easyav1_amd import ir
    pal = (b["flags"] & (ir.BLK_PALETTE_Y | ir.BLK_PALETTE_UV)).astype(bool)
    return int((pal & ((b["bwl"] == 2) | (b["bhl"] == 2))).sum())


def test_palettes_on_blocks_with_a_4_sample_side_round_trip_and_reconstruct_on_the_oracle(env):
    av1parse, streamgen, tables = env
    from easyav1_amd import ir
    seen_uv = 0
    for units, frames in streams_with(env, lambda b: narrow_palette_blocks(b) >= 1, 3):
        parser = av1parse.FrameParser(tables)
        parsed = []
        for tu, want in zip(units, frames):
            rc, f, info = parser.parse(tu)
            assert rc == 0, easyav1_amd._lib.last_error()
            same_frame(f, want)
            parsed.append(f)
            b = f.blocks[:f.desc.num_blocks]
            seen_uv += int((((b["flags"] & ir.BLK_PALETTE_UV) != 0) & ((b["bwl"] == 2) | (b["bhl"] == 2))).sum())
        parser.close()
        oracle_pictures(parsed)
    assert seen_uv > 0


def narrow_intrabc_blocks(b):
    from easyav1_amd import ir
    return int(((b["mode_y"] == ir.INTRABC_PRED) & ((b["bwl"] == 2) | (b["bhl"] == 2))).sum())


def test_intra_block_copy_of_blocks_with_a_4_sample_side_round_trips_and_reconstructs_on_the_oracle(env):
    """4x4 .. 16x4 blocks copied by a block vector: luma block by block, the chroma block of the 8x8 area by its last block's vector"""
    av1parse, streamgen, tables = env
    for units, frames in streams_with(env, lambda b: narrow_intrabc_blocks(b) >= 4, 3):
        parser = av1parse.FrameParser(tables)
        parsed = []
        for tu, want in zip(units, frames):
            rc, f, info = parser.parse(tu)
            assert rc == 0, easyav1_amd._lib.last_error()
            same_frame(f, want)
            parsed.append(f)
        parser.close()
        oracle_pictures(parsed)


@pytest.mark.gpu
@pytest.mark.parametrize("tool", ["intrabc", "palette", "narrow_palette", "narrow_intrabc"])
def test_screen_content_streams_reconstruct_on_the_gpu_like_the_oracle(env, tool):
    """streams with intra block copy / palette blocks: parser -> hand-over (whose validation checks every block vector and colour index)
    -> HIP reconstruction == the oracle"""
    from easyav1_amd import ir, video
    av1parse, streamgen, tables = env
    want = (lambda b: bool((b["mode_y"] == ir.INTRABC_PRED).sum() >= 3)) if tool == "intrabc" else (lambda b: bool((b["flags"] & (ir.BLK_PALETTE_Y | ir.BLK_PALETTE_UV)).astype(bool).sum() >= 3))
    if tool == "narrow_palette":
        want = lambda b: narrow_palette_blocks(b) >= 1
    if tool == "narrow_intrabc":
        want = lambda b: narrow_intrabc_blocks(b) >= 4
    for units, frames in streams_with(env, want, 3):
        parser = av1parse.FrameParser(tables)
        parsed = []
        for tu in units:
            rc, f, info = parser.parse(tu)
            assert rc == 0
            parsed.append(f)
        parser.close()
        ref = oracle_pictures(parsed)
        ctx = video.ReconContext(parsed[0].desc.width, parsed[0].desc.height)
        for n, f in enumerate(parsed):
            job = ctx.make_job(f)
            ctx.submit(job)
            got = coded_part(f.desc, ctx.download(f.desc.dst_slot))
            job.close()
            exp = coded_part(f.desc, ref[n])
            for pl in range(3):
                assert np.array_equal(got[pl], exp[pl]), (n, pl)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sb128", [False, True])
def test_blocks_beyond_the_coded_frame_through_the_public_interface(env, tmp_path, monkeypatch, sb128):
    """a frame size that is not a multiple of its blocks (168 x 104: 40 samples into the last superblock column and row, as 1080 rows are
    56 into theirs): blocks reach beyond the coded frame, the hand-over's frame is padded for them; the pictures that come out have the
    frame's size and equal the oracle's.  The same with superblocks of 128x128 (padded to 256 x 128) and blocks of 128 samples"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N = 166, 102, 6      # visible 166 x 102 inside the coded 168 x 104 inside the padded 192 x 128
    units, frames, headers, counts = make_stream(env, 91, (W, H), N, **(SB128 if sb128 else {}))
    assert frames[0].desc.pad == (0x3B if sb128 else 0x33) and frames[0].desc.crop == 0x22
    d = frames[0].desc
    b = frames[0].blocks
    assert int((((b["x4"].astype(int) * 4 < 168) & (b["x4"].astype(int) * 4 + (1 << b["bwl"].astype(int)) > 168))).sum()) > 0
    if sb128:
        hist = blocks_by_size(frames)
        assert hist[(128, 128)] + hist[(128, 64)] + hist[(64, 128)] > 0, hist
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0, use_128x128_superblock=int(sb128))
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blocks = [dict(track=1, tc=40 * k, key=k == 0, frames=[units[k]], lacing=0, group=False, duration=None) for k in range(N)]
    data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
    want = oracle_pictures(frames)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        assert p.has_video_frame()
        ts, planes, props = p.video_frame()
        assert (props.width, props.height) == (W, H)
        assert planes[0].shape == (H, W) and planes[1].shape == ((H + 1) // 2, (W + 1) // 2)
        for pl in range(3):
            assert np.array_equal(planes[pl], want[shown][pl][:planes[pl].shape[0], :planes[pl].shape[1]]), (shown, pl)
        shown += 1
    assert shown == N
    p.close()


@pytest.mark.gpu
def test_damaged_plain_av1_packets_end_in_an_error_status_not_in_a_crash(env, tmp_path, monkeypatch):
    """packets of a plain AV1 file damaged in different ways: the handle either reports EASYAV1_STATUS_ERROR (sticky) or keeps decoding
    something else, never crashes, and can always be destroyed"""
    from easyav1_amd import easyav1 as ea
    from tests import av1_tables_gen as tg
    from tests import webm_writer as ww
    av1parse, streamgen, tables = env
    ea.api()
    W, H, N = 128, 72, 8
    units, frames, headers, counts = make_stream(env, 95, (W, H), N)
    seq = tfh.simple_seq(max_width=W, max_height=H, enable_ref_frame_mvs=1, film_grain_params_present=0)
    seq_obu = ow.obu(1, ow.write_sequence_header(seq))
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu, default_duration=40 * 1000000, width=W, height=H)
    blob_path = tmp_path / "tables.bin"
    blob_path.write_bytes(tg.make_blob(5).tobytes())
    monkeypatch.setenv("EASYAV1_AMD_AV1_TABLES", str(blob_path))
    rng = np.random.default_rng(3)
    outcomes = set()
    for trial in range(12):
        bad = [bytearray(u) for u in units]
        k = int(rng.integers(2, N))
        if trial % 3 == 0:
            bad[k] = bad[k][:len(bad[k]) // 2]                      # truncated
        else:
            for _ in range(int(rng.integers(1, 6))):
                bad[k][int(rng.integers(len(bad[k]) // 3, len(bad[k])))] ^= 1 << int(rng.integers(8))
        blocks = [dict(track=1, tc=40 * i, key=i == 0, frames=[bytes(bad[i])], lacing=0, group=False, duration=None) for i in range(N)]
        data = ww.make_webm(7, [track], blocks, duration=float(N * 40), extras=False)
        p = ea.Player(data, enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
        shown = 0
        status = p.decode_next()
        while status == ea.STATUS_OK:
            if p.has_video_frame():
                p.video_frame()
                shown += 1
            status = p.decode_next()
        assert shown <= N      # (packets are read ahead: a damaged one may end the stream before the frames in front of it were asked for)
        outcomes.add("error" if status == ea.STATUS_ERROR else "finished")
        p.close()
    assert "error" in outcomes
