"""Build container only: fixtures of GENERATED AV1 streams against the reference decoder (round 5).

The test generator (tests/native/av1_streamgen.cpp: the parser's own TileSyntax with a symbol writer; headers by tests/av1_frame_header_writer.py) writes real AV1
bitstreams, and dav1d 1.5.3 (Pillow's libavif, tests/golden/av1_ref.py) decodes them.  That pins what libaom through Pillow never emits -- super-resolution, frames
of changing size / scaled references, switch frames, whole-sample motion vectors beside global motion, chroma restoration units of 32, GLOBAL_GLOBALMV pairs of
different global motion types, tools in combinations no encoder picks, 4:4:4 / 4:2:2 with all of it -- against the REFERENCE instead of against this repository's
own round trip.  Three conformance bugs were found this way.  Streams stay conformant by construction: a fine quantiser and small coefficient levels
(random levels overflow the inverse transforms' value range -- AV1 spec 7.13.3 -- where decoders may differ), reference sizes inside the 2 : 1 / 1 : 16 limits.

Format of tests/golden/av1_gen_<name>.npz: units / unit_sizes (temporal units), layout (dav1d's numbering), per frame in decoding order (hidden ones too)
`sizes` = the picture dav1d hands out (w, h) and all_<stage>_sha = SHA-256 of its planes after reconstruction / + deblocking / + CDEF / + restoration, meta.

    python tests/golden/make_generated_fixtures.py [name ...]"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TAME = dict(br_stop=92, golomb=97)
RICH = dict(TAME, skip=5, all_zero=10)
SMALL = dict(TAME, split_bias=70, max_mv_class=8, skip=15, all_zero=20)
S444 = dict(profile=1, subsampling_x=0, subsampling_y=0, layout=3)
S422 = dict(profile=2, subsampling_x=1, subsampling_y=0, layout=2)
SB128 = dict(use_128x128_superblock=1)
GRAIN = dict(film_grain_params_present=1)
RESIZE = lambda w, h: dict(frame_size_override_flag=1, frame_size_range=(w * 5 // 8, h * 5 // 8))   # noqa: E731
# name -> (seed, size, frames, keyword arguments of make_stream, environment, header fix-ups)
FIXTURES = {
    "plain_200x136": (101, (200, 136), 6, dict(policy=RICH), {}, {}),
    "gop3_small_blocks_264x200": (102, (264, 200), 6, dict(policy=SMALL, gop=3), {}, {}),
    "sb128_320x192": (103, (320, 192), 5, dict(seq_kw=SB128, policy=dict(RICH, split_bias=10)), {}, {}),
    "superres_200x136": (104, (200, 136), 6, dict(seq_kw=dict(enable_superres=1), policy=RICH), {}, {}),
    "superres_sb128_392x264": (105, (392, 264), 5, dict(seq_kw=dict(SB128, enable_superres=1), policy=dict(TAME, split_bias=10)), {}, {}),
    "resize_166x102": (106, (166, 102), 6, dict(hdr_kw=RESIZE(166, 102), policy=RICH), {}, {}),
    "resize_sb128_320x192": (107, (320, 192), 6, dict(seq_kw=SB128, hdr_kw=RESIZE(320, 192), policy=dict(TAME, split_bias=10)), {}, {}),
    "switch_frames_264x200": (108, (264, 200), 7, dict(switch_every=3, policy=RICH), {}, {}),
    "no_order_hint_200x136": (109, (200, 136), 6, dict(seq_kw=dict(enable_order_hint=0, enable_jnt_comp=0, enable_ref_frame_mvs=0, order_hint_bits=0), policy=RICH), {}, {}),
    "tools_off_166x102": (110, (166, 102), 6, dict(seq_kw="TOOLS_OFF", policy=RICH), {}, {}),
    "integer_mv_global_motion_200x136": (111, (200, 136), 6, dict(seq_kw=dict(seq_force_integer_mv=1), policy=dict(TAME, inter=95)), {}, {}),
    "mono_166x102": (112, (166, 102), 5, dict(seq_kw=dict(monochrome=1, layout=0), policy=RICH), {}, {}),
    "global_globalmv_264x200": (113, (264, 200), 6, dict(policy=dict(RICH, inter=95)), {"EA1_SG_FORCE": "38:6"}, {}),
    "lr_chroma_units_of_32_200x136": (114, (200, 136), 5, dict(policy=RICH), {}, dict(lr_unit_shift=0, lr_uv_shift=1)),
    "split_obus_130x98": (115, (130, 98), 6, dict(split_obus=True, gop=3, policy=RICH), {}, {}),
    "c444_200x136": (216, (200, 136), 6, dict(seq_kw=S444, policy=RICH), {}, {}),
    "c444_superres_resize_264x200": (123, (264, 200), 6, dict(seq_kw=dict(S444, enable_superres=1), hdr_kw=RESIZE(264, 200), policy=TAME), {}, {}),
    "c444_sb128_small_blocks_320x192": (118, (320, 192), 5, dict(seq_kw=dict(S444, **SB128), policy=dict(SMALL, split_bias=30)), {}, {}),
    "c422_200x136": (119, (200, 136), 6, dict(seq_kw=S422, policy=RICH), {}, {}),
    "c422_superres_166x102": (120, (166, 102), 6, dict(seq_kw=dict(S422, enable_superres=1), policy=RICH), {}, {}),
    "c422_switch_frames_264x200": (121, (264, 200), 7, dict(seq_kw=S422, switch_every=3, policy=TAME), {}, {}),
    "c422_global_globalmv_200x136": (122, (200, 136), 6, dict(seq_kw=S422, policy=dict(RICH, inter=95)), {"EA1_SG_FORCE": "38:6"}, {}),
    # two frames in a row that keep the previous segmentation map (segmentation_update_map = 0): the second one reads what the first one hands on
    "segmentation_map_kept_twice_200x136": (124, (200, 136), 6, dict(policy=dict(RICH, inter=90)), {}, {"require": "kept_twice"}),
    "c444_segmentation_map_kept_twice_166x102": (125, (166, 102), 6, dict(seq_kw=S444, policy=dict(RICH, inter=90)), {}, {"require": "kept_twice"}),
    # film grain with RANDOM parameters (libaom's sixteen test vectors are all its streams ever carry), pictures of odd size, parameters inherited from a reference
    "film_grain_random_params_166x102": (126, (166, 102), 6, dict(seq_kw=GRAIN, hdr_kw=dict(grain=True), policy=RICH), {}, {"require": "grain"}),
    "film_grain_resize_odd_sizes_200x136": (127, (200, 136), 6, dict(seq_kw=GRAIN, hdr_kw=dict(RESIZE(200, 136), grain=True), policy=TAME), {}, {"require": "grain"}),
    "c444_film_grain_random_params_130x98": (128, (130, 98), 5, dict(seq_kw=dict(S444, **GRAIN), hdr_kw=dict(grain=True), policy=RICH), {}, {"require": "grain"}),
    "c422_film_grain_random_params_166x102": (129, (166, 102), 5, dict(seq_kw=dict(S422, **GRAIN), hdr_kw=dict(grain=True), policy=RICH), {}, {"require": "grain"}),
}


def kept_twice(headers):
    keeps = [bool(h["segmentation_enabled"] and not h["segmentation_update_map"]) for h in headers]
    feats = [any(any(r) for r in h["feature_enabled"]) for h in headers]
    return any(keeps[i] and keeps[i + 1] and feats[i + 1] for i in range(len(keeps) - 1))


def grain_thrice(headers):
    g_ = [h["film_grain"] for h in headers if h["film_grain"]["apply_grain"]]
    return len(g_) >= 3 and any(not x["update_grain"] for x in g_) and any(x["num_cb_points"] for x in g_) and any(x["ar_coeff_lag"] >= 2 for x in g_)


REQUIRE = {"kept_twice": kept_twice, "grain": grain_thrice}
STAGE_BITS = (("recon", 0), ("deblock", 1), ("cdef", 3), ("final", 7))


def sha(planes):
    h = hashlib.sha256()
    for p in planes:
        if p is not None:
            h.update(np.ascontiguousarray(p).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def make(name):
    import easyav1_amd
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse
    from tests import streamgen
    from tests import test_av1_block_parse as T
    from tests.golden import av1_ref
    from tools.dbg import gen_vs_dav1d as g
    seed, size, nframes, kw, env, fix = FIXTURES[name]
    kw, fix = dict(kw), dict(fix)
    require = REQUIRE.get(fix.pop("require", None))
    if kw.get("seq_kw") == "TOOLS_OFF":
        kw["seq_kw"] = dict(T.TOOLS_OFF["seq_kw"])
    kw["hdr_kw"] = dict(kw.get("hdr_kw") or {}, base_q_idx=[12, 20, 32, 48][seed % 4])
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    tables = (av1parse, streamgen, av1parse.Tables(blob))

    def adjust(h):
        g.tame_header(h)
        if fix and any(h["frame_restoration_type"]) and not (h["coded_lossless"] or h["allow_intrabc"]):
            h.update(fix)
            h["loop_restoration_size"] = [64, 32, 32]
    for attempt in range(400):   # (a required header pattern: the first seed of seed, seed + 1000, ... that shows it)
        units, frames, headers, counts = T.make_stream(tables, seed + 1000 * attempt, size, nframes, hdr_adjust=adjust, **kw)
        if require is None or (require(headers) and g.scaling_violation(headers) is None):
            seed += 1000 * attempt
            break
    else:
        raise SystemExit("no seed shows the required pattern")
    assert g.scaling_violation(headers) is None, "reference sizes leave the 2 : 1 / 1 : 16 limits: pick another seed"
    # ... and the inverse transforms stay inside the value range of a conformant stream (the oracle counts what leaves it: oracle/av1_itx.c)
    import ctypes
    from easyav1_amd import ir
    from tests import oracle_lib as ol
    sec = {x["kind"]: x for x in av1parse.blob_sections() if x["kind"] == 5}
    ol.oracle_set_upscale_filter(np.ascontiguousarray(blob.view(np.uint16)[sec[5]["offset"] // 2:][:512].view(np.int16).reshape(64, 8)))
    lay = frames[0].desc.layout
    ol.oracle_set_chroma_format(0 if lay == 1 else 1, 1 if lay == 0 else 0)
    ol.oracle().ea1o_itx_range_violations.restype = ctypes.c_ulong
    ol.oracle().ea1o_itx_range_violations(1)
    slots, crops = [None] * ir.NUM_REF_SLOTS, [0] * ir.NUM_REF_SLOTS
    for f in frames:
        d = f.desc
        slots[d.dst_slot] = ol.oracle_decode_frame(f, slots, 15, ref_crops=crops)["final"]
        crops[d.dst_slot] = d.crop | (d.pad << 8)
        if d.upscaled_width:
            crops[d.dst_slot] = (d.crop & 0xF0) | (((d.upscaled_width + 7) & ~7) - d.upscaled_width) | ((d.pad & 0xF0) << 8)
    assert ol.oracle().ea1o_itx_range_violations(1) == 0, "a transform leaves the value range of a conformant stream: pick another seed"
    a = {"format": np.array([11], np.uint32), "units": np.frombuffer(b"".join(units), np.uint8), "unit_sizes": np.array([len(u) for u in units], np.uint32)}
    n = None
    for stage, bits in STAGE_BITS:
        pics = av1_ref.dav1d_pictures(units, inloop=bits, apply_grain=0, all_frames=True)
        a["all_%s_sha" % stage] = np.stack([sha(p) for p in pics])
        assert n in (None, len(pics))
        n = len(pics)
    a["sizes"] = np.array([[p[0].shape[1], p[0].shape[0]] for p in pics], np.uint32)
    shown = av1_ref.dav1d_pictures(units, inloop=7, apply_grain=1, all_frames=False)
    assert len(shown) == n   # (the generator shows every frame)
    a["shown_sha"] = np.stack([sha(p) for p in shown])   # with film grain, where the stream has it
    layout = 0 if pics[0][1] is None else (3 if pics[0][1].shape == pics[0][0].shape else (2 if pics[0][1].shape[0] == pics[0][0].shape[0] else 1))
    a["layout"] = np.array([layout], np.uint32)
    census = dict(frames=len(frames), grain=sum(1 for h in headers if h["film_grain"]["apply_grain"]), superres=sum(1 for f in frames if f.desc.upscaled_width), sizes=len({(f.desc.width, f.desc.height, f.desc.crop, f.desc.pad) for f in frames}),
                  switch=sum(1 for h in headers if h["frame_type"] == 3), blocks=int(sum(len(f.blocks) for f in frames)))
    meta = dict(name=name, seed=seed, size=list(size), frames=nframes, make_stream={k: (v if not callable(v) else "") for k, v in kw.items()}, environment=env, header_fixups=fix,
                census=census, versions=av1_ref.versions())
    a["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True, default=str).encode(), np.uint8)
    path = os.path.join(HERE, "av1_gen_%s.npz" % name)
    np.savez_compressed(path, **a)
    print("%-36s %d units, %6d bytes of AV1, %d frames, layout %d, %s, file %d bytes" % (name, len(units), sum(len(u) for u in units), n, layout, census, os.path.getsize(path)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        make(sys.argv[2])
        sys.exit(0)
    for nm in sys.argv[1:] or list(FIXTURES):
        env = dict(os.environ, **FIXTURES[nm][4])   # (the generator reads its debugging knobs once per process)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", nm], env=env)
        if r.returncode:
            print(nm, "FAILED", flush=True)
