"""REAL AV1 against the REFERENCE DECODER (the pin of the whole video path, round 4).

tests/golden/av1_stream_*.npz hold streams written by a real encoder (libaom 3.13.2) and the sha256 of every picture dav1d 1.5.3 -- the
decoder /root/reference/cmake/dav1d.cmake:18-21 pins and src/easyav1.c:2390,2399 calls -- makes of them, per in-loop filter set and
with film grain (tests/golden/make_video_fixtures.py; the GPU box needs only the .npz files).  Here:

  CPU   bytes -> this repository's OBU / block-level parser with the tables of easyav1_amd/data/av1_spec_tables.bin -> the oracle
        (oracle/av1_recon.c, av1_grain.c) == dav1d, every frame incl. the hidden ones, after reconstruction / deblocking / CDEF /
        restoration (one chain of references per filter set, like dav1d's inloop_filters setting), and the shown pictures with grain
  GPU   the same bytes -> parser -> hand-over -> HIP kernels == dav1d (IR level, every frame), and muxed into WebM through
        easyav1_get_video_frame == dav1d (what an application sees; default tables beside the library, no environment variable)
  live  when Pillow's libavif is importable (build container): fresh libaom encodes beside dav1d, so the fixtures cannot go stale

A mismatch here is a bug of this repository by definition."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import easyav1_amd

HERE = os.path.dirname(os.path.abspath(__file__))
ALL_FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "av1_stream_*.npz")))
# (4:4:4 / 4:2:2 streams -- av1_stream_c444_* / _c422_* -- are parser + oracle territory: tests/test_real_av1_chroma.py)
CHROMA_FIXTURES = [p for p in ALL_FIXTURES if os.path.basename(p).startswith(("av1_stream_c444_", "av1_stream_c422_"))]
FIXTURES = [p for p in ALL_FIXTURES if p not in CHROMA_FIXTURES]
STAGE_BITS = {"recon": 1, "deblock": 3, "cdef": 7, "final": 15}
BIG = 1920 * 1080


def fixture_id(path):
    return os.path.basename(path)[len("av1_stream_"):-4]


def load_fixture(path):
    z = np.load(path, allow_pickle=False)
    assert int(z["format"][0]) == 10
    sizes = z["unit_sizes"].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    raw = z["units"].tobytes()
    fx = dict(name=fixture_id(path), width=int(z["size"][0]), height=int(z["size"][1]), layout=int(z["layout"][0]),
              units=[raw[offs[i]:offs[i + 1]] for i in range(len(sizes))], shown_sha=z["shown_sha"],
              stage_sha={s: z["all_%s_sha" % s] for s in STAGE_BITS}, meta=json.loads(z["meta"].tobytes().decode()))
    fx["planes"] = {}
    i = 0
    while "p%d_y" % i in z:
        fx["planes"][i] = [z["p%d_%s" % (i, n)] if "p%d_%s" % (i, n) in z else None for n in "yuv"]
        i += 1
    return fx


def sha(planes, mono=False):
    h = hashlib.sha256()
    for p in (planes[:1] if mono else planes):
        h.update(np.ascontiguousarray(p).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def visible(desc, planes, w, h):
    """the planes cut to the shown picture"""
    return [planes[0][:h, :w], planes[1][:(h + 1) // 2, :(w + 1) // 2], planes[2][:(h + 1) // 2, :(w + 1) // 2]]


@pytest.fixture(scope="module")
def env():
    easyav1_amd.load(build=True)
    from easyav1_amd import av1parse
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    tables = av1parse.Tables(blob)
    sec = {s["kind"]: s for s in av1parse.blob_sections() if s["kind"] in (3, 5)}
    gauss = np.ascontiguousarray(blob.view(np.uint16)[sec[3]["offset"] // 2:][:2048].view(np.int16))
    upscale = np.ascontiguousarray(blob.view(np.uint16)[sec[5]["offset"] // 2:][:512].view(np.int16).reshape(64, 8))
    return av1parse, tables, gauss, upscale


def parse_all(av1parse, tables, units):
    """-> per temporal unit the list of (ParsedFrame or None for show_existing_frame, info)"""
    parser = av1parse.FrameParser(tables)
    out = [parser.parse_unit(tu) for tu in units]
    parser.close()
    return out


def oracle_stream_check(fx, parsed, gauss, upscale, stages=("recon", "deblock", "cdef", "final")):
    """every frame through one oracle chain per filter set; asserts the digests of the fixture"""
    from easyav1_amd import ir
    from tests import oracle_lib as ol
    from tests.test_video_kats import grain_stand_in
    mono = fx["layout"] == 0
    w, h = fx["width"], fx["height"]
    ol.oracle_set_upscale_filter(upscale)
    ol.oracle_set_gaussian_sequence(gauss)
    try:
        chains = {s: ([None] * ir.NUM_REF_SLOTS, [0] * ir.NUM_REF_SLOTS) for s in stages}
        k_all, k_shown = 0, 0
        for ti, frames in enumerate(parsed):
            for f, info in frames:
                if f is None:                       # show_existing_frame: the picture of a slot is shown again (dav1d's all-frames output has it again too)
                    slot = info.frame.desc.dst_slot
                    for s in stages:
                        assert np.array_equal(sha(visible(None, chains[s][0][slot], w, h), mono), fx["stage_sha"][s][k_all]), (fx["name"], "show_existing", ti, s)
                    k_all += 1
                    if "final" in chains:
                        pic = visible(None, chains["final"][0][slot], w, h)
                        if info.has_film_grain:
                            pic = ol.oracle_film_grain(info.film_grain, pic)
                        assert np.array_equal(sha(pic, mono), fx["shown_sha"][k_shown]), (fx["name"], "show_existing", ti)
                    k_shown += 1
                    continue
                d = f.desc
                for s in stages:
                    slots, crops = chains[s]
                    out = ol.oracle_decode_frame(f, slots, STAGE_BITS[s], ref_crops=crops)["final"]
                    slots[d.dst_slot] = out
                    crops[d.dst_slot] = d.crop | (d.pad << 8)
                    if d.upscaled_width:
                        up = d.upscaled_width
                        crops[d.dst_slot] = (d.crop & 0xF0) | (((up + 7) & ~7) - up) | ((d.pad & 0xF0) << 8)
                    got = sha(visible(d, out, w, h), mono)
                    assert np.array_equal(got, fx["stage_sha"][s][k_all]), (fx["name"], "frame %d of unit %d" % (k_all, ti), s)
                k_all += 1
                if info.show_frame and "final" in chains:
                    pic = visible(d, chains["final"][0][d.dst_slot], w, h)
                    if info.has_film_grain:
                        pic = ol.oracle_film_grain(info.film_grain, pic)
                    assert np.array_equal(sha(pic, mono), fx["shown_sha"][k_shown]), (fx["name"], "shown picture %d" % k_shown)
                    if k_shown in fx["planes"]:
                        for pl in range(1 if mono else 3):
                            assert np.array_equal(pic[pl], fx["planes"][k_shown][pl])
                    k_shown += 1
        assert k_all == len(fx["stage_sha"]["final"]) and ("final" not in chains or k_shown == len(fx["shown_sha"]))
    finally:
        ol.oracle_set_upscale_filter(None)
        ol.oracle_set_gaussian_sequence(grain_stand_in())


def test_fixtures_exist_and_name_the_pinned_decoder():
    assert len(FIXTURES) >= 12
    for p in FIXTURES:
        fx = load_fixture(p)
        assert "dav1d [dec]:1.5.3" in fx["meta"]["versions"] and "aom [enc]" in fx["meta"]["versions"], fx["meta"]["versions"]
        assert len(fx["shown_sha"]) >= 1 and len(fx["stage_sha"]["final"]) >= 1


@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_parser_and_oracle_reproduce_dav1d(env, path):
    av1parse, tables, gauss, upscale = env
    fx = load_fixture(path)
    parsed = parse_all(av1parse, tables, fx["units"])
    big = fx["width"] * fx["height"] >= BIG
    oracle_stream_check(fx, parsed, gauss, upscale, stages=("final",) if big else ("recon", "deblock", "cdef", "final"))


@pytest.mark.parametrize("name", ["tiles_2x2_352x288", "tiles_4cols_640x360", "uhd_4cols_3840x2160", "uhd_8cols_3840x2160", "bench_uhd_4cols_gop16"])
@pytest.mark.parametrize("threads", [2, 5])
def test_tiles_parsed_side_by_side_give_the_same_records(env, name, threads):
    """ea1_av1_frame_parser_set_threads: the tiles of a frame on several threads, merged in tile order == the tile-after-tile parse, byte for byte"""
    av1parse, tables, gauss, upscale = env
    fx = load_fixture(os.path.join(HERE, "golden", "av1_stream_%s.npz" % name))
    one = parse_all(av1parse, tables, fx["units"])
    parser = av1parse.FrameParser(tables)
    parser.set_threads(threads)
    many = [parser.parse_unit(tu) for tu in fx["units"]]
    parser.close()
    for fa, fb in zip(one, many):
        assert len(fa) == len(fb)
        for (a, ia), (b, ib) in zip(fa, fb):
            assert (a is None) == (b is None)
            if a is None:
                continue
            assert bytes(a.desc) == bytes(b.desc)
            for k in ("blocks", "txs", "coefs", "aux", "cdef_idx"):
                assert np.array_equal(getattr(a, k), getattr(b, k)), (name, k)
            for pl in range(3):
                assert np.array_equal(a.lr[pl], b.lr[pl])


def tool_census(parsed):
    """how many blocks of the parsed frames reach each tool"""
    from easyav1_amd import ir
    c = dict(frames=0, hidden=0, show_existing=0, blocks=0, intra=0, inter=0, compound=0, obmc=0, warp=0, interintra=0, wedge=0, diffwtd=0, dist=0,
             palette=0, intrabc=0, cfl=0, filter_intra=0, directional=0, smooth=0, paeth=0, tx64=0, tx_1d_or_flip=0, identity=0, wht=0, sgr=0, wiener=0,
             b128=0, sub8=0, dual_filter=0)
    for frames in parsed:
        for f, info in frames:
            if f is None:
                c["show_existing"] += 1
                continue
            c["frames"] += 1
            c["hidden"] += 0 if info.show_frame else 1
            b = f.blocks
            inter = (b["flags"] & ir.BLK_INTER) != 0
            c["blocks"] += len(b)
            c["inter"] += int(inter.sum())
            c["intra"] += int((~inter & (b["mode_y"] != ir.INTRABC_PRED)).sum())
            c["compound"] += int(((b["flags"] & ir.BLK_COMPOUND) != 0).sum())
            c["obmc"] += int(((b["flags"] & ir.BLK_OBMC) != 0).sum())
            c["warp"] += int(((b["flags"] & ir.BLK_WARP) != 0).sum())
            c["interintra"] += int(((b["flags"] & ir.BLK_INTERINTRA) != 0).sum())
            comp = (b["flags"] & ir.BLK_COMPOUND) != 0
            c["wedge"] += int((comp & (b["comp_type"] == ir.COMP_WEDGE)).sum())
            c["diffwtd"] += int((comp & (b["comp_type"] == ir.COMP_DIFFWTD)).sum())
            c["dist"] += int((comp & (b["comp_type"] == ir.COMP_DIST)).sum())
            c["palette"] += int(((b["flags"] & (ir.BLK_PALETTE_Y | ir.BLK_PALETTE_UV)) != 0).sum())
            c["intrabc"] += int((~inter & (b["mode_y"] == ir.INTRABC_PRED)).sum())
            c["cfl"] += int((~inter & (b["mode_uv"] == ir.UV_CFL_PRED)).sum())
            c["filter_intra"] += int((~inter & (b["filter_intra"] != 0)).sum())
            c["directional"] += int((~inter & (b["mode_y"] >= ir.V_PRED) & (b["mode_y"] <= ir.D67_PRED)).sum())
            c["smooth"] += int((~inter & (b["mode_y"] >= ir.SMOOTH_PRED) & (b["mode_y"] <= ir.SMOOTH_H_PRED)).sum())
            c["paeth"] += int((~inter & (b["mode_y"] == ir.PAETH_PRED)).sum())
            c["b128"] += int(((b["bwl"] == 7) | (b["bhl"] == 7)).sum())
            c["sub8"] += int(((b["bwl"] == 2) | (b["bhl"] == 2)).sum())
            c["dual_filter"] += int((inter & ((b["mode_y"] & 3) != ((b["mode_y"] >> 2) & 3))).sum())
            t = f.txs
            coded = t["coef_off"] != ir.NO_COEFS
            c["tx64"] += int((((t["size"] & 15) == 6) | ((t["size"] >> 4) == 6)).sum())
            c["wht"] += int((coded & (t["txtp"] == ir.TXTP_WHT)).sum())
            kinds = t["txtp"] & 15
            c["identity"] += int((coded & (t["txtp"] != ir.TXTP_WHT) & (((kinds & 3) == ir.TX_IDENTITY) | ((kinds >> 2) == ir.TX_IDENTITY))).sum())
            c["tx_1d_or_flip"] += int((coded & (t["txtp"] != ir.TXTP_WHT) & (((kinds & 3) == ir.TX_FLIPADST) | ((kinds >> 2) == ir.TX_FLIPADST))).sum())
            for pl in range(3):
                c["sgr"] += int((f.lr[pl]["type"] == ir.LR_SGRPROJ).sum())
                c["wiener"] += int((f.lr[pl]["type"] == ir.LR_WIENER).sum())
    return c


def test_the_fixture_set_reaches_the_tools(env):
    """"pinned" quantified: blocks of the committed libaom streams per tool (DESIGN.md 2 quotes this table); every tool libaom emits must
    be reached by at least one block"""
    av1parse, tables, gauss, upscale = env
    total = None
    for p in FIXTURES:
        fx = load_fixture(p)
        c = tool_census(parse_all(av1parse, tables, fx["units"]))
        total = c if total is None else {k: total[k] + v for k, v in c.items()}
    print("\ntool census of the committed libaom streams:", json.dumps(total, sort_keys=True))
    for k in ("hidden", "show_existing", "intra", "inter", "compound", "obmc", "warp", "interintra", "wedge", "diffwtd", "dist", "palette", "intrabc", "cfl",
              "filter_intra", "directional", "smooth", "paeth", "tx64", "tx_1d_or_flip", "identity", "wht", "sgr", "wiener", "b128", "sub8", "dual_filter"):
        assert total[k] > 0, (k, total)


# ---------------------------------------------------------------------------------------------------------------- live (build container)
def _live_available():
    try:
        from tests.golden import av1_ref
        return av1_ref.available()
    except Exception:
        return False


@pytest.mark.skipif(not _live_available(), reason="Pillow's libavif (libaom + dav1d) is not importable on this machine")
@pytest.mark.parametrize("seed,kw", [(901, dict(w=120, h=88, n=5)), (902, dict(w=168, h=120, n=6, speed=2, quality=50)),
                                     (903, dict(w=200, h=136, n=4, kind="screen", advanced={"tune-content": "screen"}))])
def test_fresh_libaom_streams_decode_like_dav1d(env, seed, kw):
    """not from a committed file: encode now, decode with dav1d now, compare with parser + oracle"""
    from tests.golden import av1_ref
    from tests.golden.make_video_fixtures import STAGE_BITS as SB
    from tests.golden.make_video_fixtures import sha as fsha
    av1parse, tables, gauss, upscale = env
    kw = dict(kw)
    w, h, n, kind = kw.pop("w"), kw.pop("h"), kw.pop("n"), kw.pop("kind", "natural")
    units = av1_ref.encode_sequence(av1_ref.synthetic_frames(seed, w, h, n, kind), **kw)
    fx = dict(name="live %d" % seed, width=w, height=h, layout=1, units=units, planes={},
              shown_sha=np.stack([fsha(p) for p in av1_ref.dav1d_pictures(units)]),
              stage_sha={s: np.stack([fsha(p) for p in av1_ref.dav1d_pictures(units, inloop=b, apply_grain=0, all_frames=True)]) for s, b in SB})
    oracle_stream_check(fx, parse_all(av1parse, tables, units), gauss, upscale)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_hip_path_reproduces_dav1d(env, path):
    """bytes -> parser -> hand-over -> HIP kernels: every frame (hidden ones too) equals dav1d's picture before film grain"""
    from easyav1_amd import video
    av1parse, tables, gauss, upscale = env
    fx = load_fixture(path)
    mono = fx["layout"] == 0
    parsed = parse_all(av1parse, tables, fx["units"])
    frames = [f for fr in parsed for f, info in fr if f is not None]
    ctx = video.ReconContext(max(max(f.desc.width, (f.desc.upscaled_width + 7) & ~7) for f in frames), max(f.desc.height for f in frames))
    ctx.set_upscale_filter(upscale)
    k = 0
    for fr in parsed:
        for f, info in fr:
            if f is None:      # show_existing_frame: the slot's picture again (dav1d's all-frames output has it again too)
                got = visible(None, ctx.download(info.frame.desc.dst_slot), fx["width"], fx["height"])
            else:
                job = ctx.make_job(f)
                ctx.submit(job)
                got = visible(f.desc, ctx.download(f.desc.dst_slot), fx["width"], fx["height"])
                job.close()
            assert np.array_equal(sha(got, mono), fx["stage_sha"]["final"][k]), (fx["name"], k)
            k += 1
    assert k == len(fx["stage_sha"]["final"])
    ctx.close()


def split_obus(tu):
    """[(type, offset, total size)] of the OBUs of a temporal unit (every OBU with a size field, as libaom writes them)"""
    out, pos = [], 0
    while pos < len(tu):
        b0 = tu[pos]
        ext, has_size = (b0 >> 2) & 1, (b0 >> 1) & 1
        assert has_size
        q = pos + 1 + ext
        size, shift = 0, 0
        while True:
            v = tu[q]
            q += 1
            size |= (v & 0x7F) << shift
            shift += 7
            if not v & 0x80:
                break
        out.append(((b0 >> 3) & 15, pos, q + size - pos))
        pos = q + size
    return out


def mux_webm(fx, fps=25, key_flags="true", repeat=1):
    """the fixture's temporal units as a WebM file (V_AV1; the temporal delimiters stripped as a Matroska muxer does)"""
    from tests import webm_writer as ww
    units = []
    seq = None
    for tu in fx["units"]:
        body = b""
        for typ, off, size in split_obus(tu):
            if typ == 2:
                continue
            if typ == 1 and seq is None:
                seq = tu[off:off + size]
            body += tu[off:off + size]
        units.append(body)
    dur = 1000 // fps
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seq, default_duration=dur * 1000000, width=fx["width"], height=fx["height"])
    keys = [any(typ == 1 for typ, off, size in split_obus(tu)) for tu in fx["units"]]   # libaom writes a sequence header with every key frame
    if key_flags == "all":        # a muxer that flags every block as a key frame (BlockGroups without ReferenceBlock)
        keys = [True] * len(keys)
    elif key_flags == "first":    # ... or only the first
        keys = [k == 0 for k in range(len(keys))]
    n = len(units)
    blocks = [dict(track=1, tc=dur * k, key=keys[k % n], frames=[units[k % n]], lacing=0, group=False, duration=None) for k in range(n * repeat)]
    return ww.make_webm(11, [track], blocks, duration=float(n * repeat * dur), extras=False)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_real_files_through_the_public_interface(env, path, monkeypatch):
    """a WebM file of a real libaom stream -> easyav1_init_from_memory / easyav1_decode_next / easyav1_get_video_frame (the calls of
    tools/easyav1_benchmark.c:79-187) == dav1d's shown pictures, film grain included; the tables are the library's own (no environment)"""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    fx = load_fixture(path)
    mono = fx["layout"] == 0
    ea.api()
    p = ea.Player(mux_webm(fx), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        assert planes[0].shape == (fx["height"], fx["width"])
        assert np.array_equal(sha([x for x in planes if x is not None], mono), fx["shown_sha"][shown]), (fx["name"], shown)
        shown += 1
    assert p.get_status() != ea.STATUS_ERROR
    assert shown == len(fx["shown_sha"]), (shown, len(fx["shown_sha"]))
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["two_gops_104x72", "pyramid_200x136"])
@pytest.mark.parametrize("key_flags", ["all", "first"])
@pytest.mark.parametrize("gops", ["1", "4"])
def test_the_containers_key_flags_are_not_trusted(env, name, key_flags, gops, monkeypatch):
    """GOP chains start where the BITSTREAM has a shown key frame, whatever the WebM blocks claim (dav1d behind the reference's decode_video,
    src/easyav1.c:2381-2428, never sees the container flag): inter packets flagged as key frames must not land on a fresh chain, key
    packets without the flag must still start one"""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    monkeypatch.setenv("EASYAV1_AMD_GOPS", gops)
    fx = load_fixture(os.path.join(HERE, "golden", "av1_stream_%s.npz" % name))
    ea.api()
    p = ea.Player(mux_webm(fx, key_flags=key_flags), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        assert np.array_equal(sha(planes), fx["shown_sha"][shown]), (name, key_flags, shown)
        shown += 1
    assert p.get_status() != ea.STATUS_ERROR and shown == len(fx["shown_sha"])
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead,gops", [("10", "4"), ("12", "16")])
def test_gops_longer_than_the_look_ahead(env, lookahead, gops, monkeypatch):
    """GOPs of 24 frames against a look-ahead of 10 / 12 packets: the shim reads on until every chain has a GOP (want_more_packets), the
    chains then decode GOPs that lie far apart in the stream -- same pictures, in order"""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    monkeypatch.setenv("EASYAV1_AMD_LOOKAHEAD", lookahead)
    monkeypatch.setenv("EASYAV1_AMD_GOPS", gops)
    fx = load_fixture(os.path.join(HERE, "golden", "av1_stream_long_136x104.npz"))
    repeat, n = 9, len(fx["shown_sha"])
    ea.api()
    p = ea.Player(mux_webm(fx, repeat=repeat), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        assert np.array_equal(sha(planes), fx["shown_sha"][shown % n]), (lookahead, gops, shown)
        shown += 1
    assert p.get_status() != ea.STATUS_ERROR and shown == repeat * n
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,repeat", [("two_gops_104x72", 3), ("inter_104x72", 7), ("pyramid_200x136", 4), ("bench_hd_gop16", 5)])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_gop_chains_across_device_contexts(env, name, repeat, devices, monkeypatch):
    """EASYAV1_AMD_DEVICES: one reconstruction context + stage thread + helpers per entry, GOPs dealt round-robin over them (closed GOPs need no
    collective), pictures handed out in stream order.  On the one-GPU box the entries name the same device twice / three
    times -- the dealing, the per-context pipelines and the ordering are what is under test; bit-exact against dav1d's pictures."""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    monkeypatch.setenv("EASYAV1_AMD_DEVICES", devices)
    fx = load_fixture(os.path.join(HERE, "golden", "av1_stream_%s.npz" % name))
    want = list(fx["shown_sha"]) * repeat
    ea.api()
    p = ea.Player(mux_webm(fx, repeat=repeat), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown, last_ts = 0, -1
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        assert ts > last_ts, (name, shown)
        last_ts = ts
        assert np.array_equal(sha(planes), want[shown]), (name, devices, shown)
        shown += 1
    assert p.get_status() != ea.STATUS_ERROR and shown == len(want)
    p.close()


def mux_webm_of(fxs, track_of=0, fps=25):
    """several fixtures one after the other in ONE video track (each with its own sequence header at its key frames), the track's
    size and CodecPrivate those of fxs[track_of]"""
    from tests import webm_writer as ww
    dur = 1000 // fps
    blocks, seqs, k = [], [], 0
    for fx in fxs:
        seq = None
        for tu in fx["units"]:
            body, key = b"", False
            for typ, off, size in split_obus(tu):
                if typ == 2:
                    continue
                if typ == 1:
                    key = True
                    seq = seq or tu[off:off + size]
                body += tu[off:off + size]
            blocks.append(dict(track=1, tc=dur * k, key=key, frames=[body], lacing=0, group=False, duration=None))
            k += 1
        seqs.append(seq)
    f0 = fxs[track_of]
    track = dict(number=1, type=1, codec="V_AV1", private=bytes([0x81, 0x00, 0x0C, 0x00]) + seqs[0], default_duration=dur * 1000000, width=f0["width"], height=f0["height"])
    return ww.make_webm(11, [track], blocks, duration=float(k * dur), extras=False)


def named(name):
    return load_fixture([p for p in ALL_FIXTURES if p.endswith("av1_stream_%s.npz" % name)][0])


def check_pictures(p, ea, want, names):
    """every remaining packet through easyav1_decode_next; each picture == dav1d's, described from its own sequence header"""
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        assert shown < len(want), "more pictures than the stream holds"
        fx, i = want[shown]
        mono = fx["layout"] == 0
        assert planes[0].shape == (fx["height"], fx["width"]), (names, shown)
        # (easyav1_pixel_layout: 1 = YUV400 ... 4 = YUV444 = dav1d's layout number + 1, src/easyav1.h:183-188)
        assert props.pixel_layout == int(fx["layout"]) + 1 and (planes[1] is None) == mono and (planes[2] is None) == mono, (names, shown, props.pixel_layout)
        assert np.array_equal(sha([x for x in planes if x is not None], mono), fx["shown_sha"][i]), (names, shown)
        shown += 1
    assert shown == len(want), (names, shown, len(want))
    assert p.get_status() != ea.STATUS_ERROR, names


@pytest.mark.gpu
@pytest.mark.parametrize("names,track_of", [
    (["deltalf_q30_264x200", "pyramid_200x136"], 0),        # a smaller sequence follows
    (["pyramid_200x136", "deltalf_q30_264x200"], 1),        # a larger one, inside the track's size
    (["pyramid_200x136", "mono_136x104", "inter_104x72"], 0),   # 4:2:0, 4:0:0, 4:2:0: the chroma planes come and go with the PICTURE's sequence
    (["film_grain_mono_130x98", "odd_130x98"], 0),
    (["pyramid_200x136", "c444_screen_200x136"], 0),        # another chroma layout: a context of its own from that key frame on
    (["pyramid_200x136", "deltalf_q30_264x200"], 0),        # larger than the track says: likewise
    (["pyramid_200x136", "c444_screen_200x136", "odd_130x98", "c422_slow_136x104", "deltalf_q30_264x200"], 0),   # there and back again
], ids=["smaller", "larger", "mono_between", "mono_first", "another_layout", "beyond_the_track", "there_and_back"])
def test_a_track_that_changes_its_sequence_header(env, names, track_of, monkeypatch):
    """sequence headers may change inside a track (at key frames).  The pipeline decodes many GOPs ahead of the picture that is shown, so every picture is
    described from ITS OWN sequence header -- size, layout, 4:0:0 without chroma planes (dav1d's pictures carry theirs: src/easyav1.c:3287-3504, :3557-3563)
    -- and each == dav1d's.  A sequence the decoder context cannot hold (another chroma layout, frames beyond the context's size: dav1d re-allocates its
    pictures there) waits until its packet's turn has come, then gets a context of its own; the pictures in front of it are handed out as ever"""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    fxs = [named(n) for n in names]
    want = [(fx, i) for fx in fxs for i in range(len(fx["shown_sha"]))]
    ea.api()
    p = ea.Player(mux_webm_of(fxs, track_of), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    check_pictures(p, ea, want, names)
    p.close()


@pytest.mark.gpu
def test_seeking_across_a_change_of_context(env, monkeypatch):
    """4:2:0 then 4:4:4 in one track: after the whole file a seek back to the start meets the first sequence with the context of the second (a new one is made
    again, whatever was queued is dropped), a seek into the second sequence from the first lands on its key frame; pictures == dav1d's from there on"""
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    names = ["pyramid_200x136", "c444_screen_200x136"]
    fxs = [named(n) for n in names]
    want = [(fx, i) for fx in fxs for i in range(len(fx["shown_sha"]))]
    n0 = len(fxs[0]["shown_sha"])
    assert len(fxs[0]["units"]) == n0 and len(fxs[1]["units"]) == len(fxs[1]["shown_sha"]), "one picture per packet is what the timestamps below assume"
    ea.api()
    p = ea.Player(mux_webm_of(fxs), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    check_pictures(p, ea, want, names)
    assert p.seek_to_timestamp(0) == ea.STATUS_OK
    check_pictures(p, ea, want, names)
    assert p.seek_to_timestamp(0) == ea.STATUS_OK
    for _ in range(3):
        assert p.decode_next() == ea.STATUS_OK
    # into the second sequence (25 pictures a second): its key frame is decoded inside the seek -- with a context of its own -- and thrown away
    assert p.seek_to_timestamp(40 * n0 + 5) == ea.STATUS_OK
    check_pictures(p, ea, want[n0 + 1:], names)
    p.close()


@pytest.mark.gpu
@pytest.mark.skipif(not _live_available(), reason="Pillow's libavif (libaom + dav1d) is not importable on this machine")
@pytest.mark.parametrize("sub,layout", [("4:2:0", 1), ("4:4:4", 3), ("4:2:2", 2), ("4:0:0", 0)])
def test_pictures_of_unusual_sizes_through_the_public_interface(env, sub, layout, monkeypatch):
    """1x1, 3x5, 64x1, 4x128, 130x2, 129x129: encoded by libaom now, every chroma layout, through easyav1_* == dav1d's pictures (frames smaller than one 8x8
    block, one sample wide or high, odd in both directions: the padding of the hand-over, chroma planes of (w + 1) >> 1 samples; the long sweep over 16 sizes:
    tools/dbg/tiny_sizes_gpu.py, profiles/r05_y_tiny_sizes.log)"""
    from tests.golden import av1_ref
    from easyav1_amd import easyav1 as ea
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    ea.api()
    rng = np.random.default_rng(17)
    for w, h in [(1, 1), (3, 5), (64, 1), (4, 128), (130, 2), (129, 129)]:
        frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(3)]
        units = av1_ref.encode_sequence(frames, subsampling=sub, quality=70)
        shown = av1_ref.dav1d_pictures(units)
        p = ea.Player(mux_webm(dict(name="tiny", width=w, height=h, layout=layout, units=units)), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
        k = 0
        while p.decode_next() == ea.STATUS_OK:
            if not p.has_video_frame():
                continue
            ts, planes, props = p.video_frame()
            got = [x for x in planes if x is not None]
            assert len(got) == (1 if layout == 0 else 3), (w, h, sub)
            for g, q in zip(got, shown[k]):
                assert g.shape == np.asarray(q).shape and np.array_equal(g, q), (w, h, sub, k)
            k += 1
        assert k == len(shown) and p.get_status() != ea.STATUS_ERROR, (w, h, sub, k)
        p.close()
