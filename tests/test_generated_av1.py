"""GENERATED AV1 streams against the reference decoder (round 5).

tests/golden/av1_gen_*.npz (tests/golden/make_generated_fixtures.py): bitstreams written by this repository's test generator -- the parser's TileSyntax with a
symbol writer -- decoded by dav1d 1.5.3, whose pictures after reconstruction / deblocking / CDEF / restoration are kept as digests per frame.  They pin what libaom
through Pillow does not emit (tests/golden/README.md): super-resolution, frames of changing size predicted from scaled references, switch frames, 128x128
superblocks in mass, sequences without order hints or with tools switched off, whole-sample motion vectors beside global motion, GLOBAL_GLOBALMV pairs, chroma
restoration units of 32 samples, OBU_FRAME_HEADER + OBU_TILE_GROUP splits -- each also in 4:4:4 / 4:2:2 -- against the REFERENCE instead of against the round trip
generator -> parser and the self-comparison oracle == kernels.  Three conformance bugs fell out of the first run.

CPU: bytes -> block parser -> CPU oracle == dav1d, every frame, every stage.  GPU (`-m gpu`): bytes -> parser -> hand-over -> HIP kernels == dav1d likewise, and
as WebM files through easyav1_* (every picture at its own size)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import easyav1_amd
from tests import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "av1_gen_*.npz")))
STAGE_BITS = {"recon": 1, "deblock": 3, "cdef": 7, "final": 15}
SUB = {0: (1, 1), 1: (1, 1), 2: (1, 0), 3: (0, 0)}   # dav1d's layout number -> subsampling_x, subsampling_y


def fixture_id(path):
    return os.path.basename(path)[len("av1_gen_"):-4]


def load(path):
    z = np.load(path, allow_pickle=False)
    assert int(z["format"][0]) == 11
    offs = np.concatenate([[0], np.cumsum(z["unit_sizes"].astype(np.int64))])
    raw = z["units"].tobytes()
    return dict(name=fixture_id(path), layout=int(z["layout"][0]), units=[raw[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], sizes=z["sizes"].astype(int),
                stage_sha={s: z["all_%s_sha" % s] for s in STAGE_BITS}, shown_sha=z["shown_sha"], meta=json.loads(z["meta"].tobytes().decode()))


def sha(planes, mono):
    h = hashlib.sha256()
    for p in (planes[:1] if mono else planes):
        h.update(np.ascontiguousarray(p).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def cut(planes, w, h, ssx, ssy):
    return [planes[0][:h, :w], planes[1][:(h + ssy) >> ssy, :(w + ssx) >> ssx], planes[2][:(h + ssy) >> ssy, :(w + ssx) >> ssx]]


@pytest.fixture(scope="module")
def env():
    easyav1_amd.load(build=True)
    from easyav1_amd import av1parse
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    sec = {s["kind"]: s for s in av1parse.blob_sections() if s["kind"] == 5}
    upscale = np.ascontiguousarray(blob.view(np.uint16)[sec[5]["offset"] // 2:][:512].view(np.int16).reshape(64, 8))
    sec = {s["kind"]: s for s in av1parse.blob_sections() if s["kind"] == 3}
    gauss = np.ascontiguousarray(blob.view(np.uint16)[sec[3]["offset"] // 2:][:2048].view(np.int16))
    return av1parse, av1parse.Tables(blob), upscale, gauss


def parse(av1parse, tables, fx):
    parser = av1parse.FrameParser(tables)
    out = []
    for tu in fx["units"]:
        rc, f, info = parser.parse(tu)
        assert rc == 0, (fx["name"], easyav1_amd._lib.last_error())
        out.append((f, info))
    parser.close()
    return out


def ref_crop(d):
    """ea1_frame_desc.crop | pad << 8 of the picture a frame leaves in its slot (super-resolution: the upscaled picture, nothing padded to the right)"""
    if d.upscaled_width:
        up = d.upscaled_width
        return (d.crop & 0xF0) | (((up + 7) & ~7) - up) | ((d.pad & 0xF0) << 8)
    return d.crop | (d.pad << 8)


def test_the_set_covers_what_it_is_for():
    assert len(FIXTURES) >= 20
    metas = [load(p)["meta"] for p in FIXTURES]
    assert sum(m["census"]["superres"] for m in metas) >= 8 and sum(m["census"]["switch"] for m in metas) >= 4
    assert sum(1 for m in metas if m["census"]["sizes"] > 1) >= 6 and {load(p)["layout"] for p in FIXTURES} == {0, 1, 2, 3}
    assert all("1.5.3" in json.dumps(m["versions"]) for m in metas)


@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_parser_and_oracle_reproduce_dav1d(env, path):
    from easyav1_amd import ir
    av1parse, tables, upscale, gauss = env
    fx = load(path)
    ssx, ssy = SUB[fx["layout"]]
    mono = fx["layout"] == 0
    ol.oracle_set_upscale_filter(upscale)
    ol.oracle_set_chroma_format(ssx, ssy)
    ol.oracle_set_gaussian_sequence(gauss)
    try:
        chains = {s: ([None] * ir.NUM_REF_SLOTS, [0] * ir.NUM_REF_SLOTS) for s in STAGE_BITS}
        for k, (f, info) in enumerate(parse(av1parse, tables, fx)):
            d = f.desc
            w, h = fx["sizes"][k]
            for s, bits in STAGE_BITS.items():
                slots, crops = chains[s]
                out = ol.oracle_decode_frame(f, slots, bits, ref_crops=crops)["final"]
                slots[d.dst_slot], crops[d.dst_slot] = out, ref_crop(d)
                assert np.array_equal(sha(cut(out, w, h, ssx, ssy), mono), fx["stage_sha"][s][k]), (fx["name"], "frame %d" % k, s)
            pic = cut(chains["final"][0][d.dst_slot], w, h, ssx, ssy)   # as shown: with film grain where the frame has it
            if info.has_film_grain:
                pic = ol.oracle_film_grain(info.film_grain, pic)
            assert np.array_equal(sha(pic, mono), fx["shown_sha"][k]), (fx["name"], "shown picture %d" % k, "film grain" if info.has_film_grain else "")
        assert k + 1 == len(fx["sizes"])
    finally:
        from tests.test_video_kats import grain_stand_in
        ol.oracle_set_chroma_format(1, 1)
        ol.oracle_set_gaussian_sequence(grain_stand_in())


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_hip_path_reproduces_dav1d(env, path):
    """one context per filter set (each predicts from its own output, like dav1d's): every frame equals dav1d's picture at every stage"""
    from easyav1_amd import video, ir, grain
    av1parse, tables, upscale, gauss = env
    fx = load(path)
    ssx, ssy = SUB[fx["layout"]]
    mono = fx["layout"] == 0
    layout = {2: ir.LAYOUT_422, 3: ir.LAYOUT_444}.get(fx["layout"], ir.LAYOUT_420)
    parsed = parse(av1parse, tables, fx)
    W = max(max(f.desc.width, (f.desc.upscaled_width + 7) & ~7) for f, info in parsed)
    H = max(f.desc.height for f, info in parsed)
    ctxs = {s: video.ReconContext(W, H, layout=layout) for s in STAGE_BITS}
    try:
        for c in ctxs.values():
            c.set_upscale_filter(upscale)
        grain.set_gaussian_sequence(ctxs["final"], gauss)
        for k, (f, info) in enumerate(parsed):
            w, h = fx["sizes"][k]
            assert f.desc.layout == layout
            for s, bits in STAGE_BITS.items():
                job = ctxs[s].make_job(f)
                ctxs[s].submit(job, stages=bits)
                got = cut(ctxs[s].download(f.desc.dst_slot), w, h, ssx, ssy)
                job.close()
                assert np.array_equal(sha(got, mono), fx["stage_sha"][s][k]), (fx["name"], "frame %d (%dx%d, upscaled to %d)" % (k, f.desc.width, f.desc.height, f.desc.upscaled_width), s)
            if info.has_film_grain:   # the picture as shown
                grain.film_grain(ctxs["final"], f.desc.dst_slot, info.film_grain)
                got = cut(grain.download_display(ctxs["final"]), w, h, ssx, ssy)
            assert np.array_equal(sha(got, mono), fx["shown_sha"][k]), (fx["name"], "shown picture %d" % k, "film grain" if info.has_film_grain else "")
    finally:
        for c in ctxs.values():
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=fixture_id)
def test_through_the_public_interface(env, path, monkeypatch):
    """the streams as WebM files through easyav1_init_from_memory / easyav1_decode_next / easyav1_get_video_frame: dav1d's pictures as shown (film grain
    included), every picture at its own size (frames of changing size, switch frames: `p.w` / `p.h` of the picture, src/easyav1.c:3557-3568)"""
    from easyav1_amd import easyav1 as ea
    from tests import test_real_av1 as tr
    monkeypatch.delenv("EASYAV1_AMD_AV1_TABLES", raising=False)
    fx = load(path)
    w0, h0 = (int(v) for v in fx["sizes"].max(axis=0))
    ea.api()
    p = ea.Player(tr.mux_webm(dict(units=fx["units"], width=w0, height=h0)), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
    shown = 0
    while p.decode_next() == ea.STATUS_OK:
        if not p.has_video_frame():
            continue
        ts, planes, props = p.video_frame()
        w, h = (int(v) for v in fx["sizes"][shown])
        assert planes[0].shape == (h, w), (fx["name"], shown, planes[0].shape, (h, w))
        assert np.array_equal(sha([x for x in planes if x is not None], fx["layout"] == 0), fx["shown_sha"][shown]), (fx["name"], shown)
        shown += 1
    assert p.get_status() != ea.STATUS_ERROR
    assert shown == len(fx["sizes"])   # (the generator's frames are all shown)
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c444_200x136", "c422_200x136", "c444_sb128_small_blocks_320x192", "c422_switch_frames_264x200"])
def test_mutated_frames_of_the_other_layouts_match_the_oracle(env, name):
    """differential fuzzing in 4:4:4 / 4:2:2 contexts (as tests/test_video_gpu.py does in 4:2:0): single-field corruptions of a parsed frame that the hand-over
    still ACCEPTS are frames in their own right -- the kernels must decode them exactly like the oracle, and must not fault on what validation lets through"""
    import ctypes
    from easyav1_amd import video, ir, Ea1Error
    av1parse, tables, upscale, gauss = env
    fx = load([p for p in FIXTURES if p.endswith("av1_gen_%s.npz" % name)][0])
    ssx, ssy = SUB[fx["layout"]]
    layout = {2: ir.LAYOUT_422, 3: ir.LAYOUT_444}[fx["layout"]]
    parsed = [f for f, info in parse(av1parse, tables, fx)]
    W, H = max(f.desc.width for f in parsed), max(f.desc.height for f in parsed)
    rng = np.random.default_rng(int(os.environ.get("EA1_DIFF_SEED", "7")))
    want_checks = int(os.environ.get("EA1_DIFF_FUZZ", "40"))
    ctx = video.ReconContext(W, H, layout=layout)
    ctx.set_upscale_filter(upscale)
    ol.oracle_set_upscale_filter(upscale)
    ol.oracle_set_chroma_format(ssx, ssy)
    try:
        # the valid frames up to the one that gets mutated: references on the device and for the oracle
        victim = next(k for k, f in enumerate(parsed) if k >= 2 and f.desc.is_inter)
        slots, crops = [None] * ir.NUM_REF_SLOTS, [0] * ir.NUM_REF_SLOTS
        for f in parsed[:victim]:
            job = ctx.make_job(f)
            ctx.submit(job)
            job.close()
            slots[f.desc.dst_slot] = ol.oracle_decode_frame(f, slots, ref_crops=crops)["final"]
            crops[f.desc.dst_slot] = ref_crop(f.desc)
        f = parsed[victim]
        free = [s for s in range(ir.NUM_REF_SLOTS) if slots[s] is None] or [f.desc.dst_slot]
        checked = 0
        for trial in range(60 * want_checks):
            if checked >= want_checks:
                break
            blocks, txs, coefs, aux = f.blocks.copy(), f.txs.copy(), f.coefs.copy(), f.aux.copy()
            cdef_idx, lr, desc = f.cdef_idx.copy(), [u.copy() for u in f.lr], f.desc.copy()
            desc.dst_slot = free[0]   # (the references stay as they are: every trial starts from the same state)
            kind = trial % 4
            if kind == 0:
                raw = blocks.view(np.uint8).reshape(len(blocks), -1)
                raw[rng.integers(len(blocks)), rng.integers(raw.shape[1])] = rng.integers(256)
            elif kind == 1:
                raw = txs.view(np.uint8).reshape(len(txs), -1)
                raw[rng.integers(len(txs)), rng.integers(raw.shape[1])] = rng.integers(256)
            elif kind == 2:
                if len(cdef_idx):
                    cdef_idx[rng.integers(len(cdef_idx))] = rng.integers(256)
                for u in lr:
                    if len(u):
                        u.view(np.uint8)[rng.integers(u.nbytes)] = rng.integers(256)
            else:
                raw = (ctypes.c_uint8 * ctypes.sizeof(desc)).from_address(ctypes.addressof(desc))
                at = 8 + int(rng.integers(36))
                if at != 9:   # (not dst_slot)
                    raw[at] = int(rng.integers(256))
            bad = ir.ParsedFrame(desc, blocks, txs, coefs, aux, cdef_idx, lr)
            if bad.desc.upscaled_width:
                continue
            try:
                j = ctx.make_job(bad)
            except Ea1Error:
                continue
            try:
                ctx.submit(j)
            except Ea1Error:
                j.close()
                continue
            got = ctx.download(bad.desc.dst_slot)
            j.close()
            want = ol.oracle_decode_frame(bad, slots, ref_crops=crops)["final"]
            # (the picture beyond the coded size the frame declares is undefined: ea1_frame_desc.pad)
            cw, ch = bad.desc.width - 8 * (bad.desc.pad & 15), bad.desc.height - 8 * (bad.desc.pad >> 4)
            for pl in range(3):
                pw_, ph_ = (cw, ch) if pl == 0 else (cw >> ssx, ch >> ssy)
                assert np.array_equal(got[pl][:ph_, :pw_], want[pl][:ph_, :pw_]), (name, "trial %d kind %d plane %d" % (trial, kind, pl))
            checked += 1
        assert checked >= min(want_checks, 30), checked
    finally:
        ol.oracle_set_chroma_format(1, 1)
        ctx.close()
