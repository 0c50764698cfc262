"""Writes tests/golden/video_regression.json: sha256 of the ORACLE's output pictures for a few small
synthetic streams.  These are regression digests of our own CPU restatement (the reference has no
vectors for the video path and dav1d cannot be built here) -- they pin the oracle against
accidental change and across machines, they do NOT pin it against dav1d."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from easyav1_amd import ir  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402
from tests.test_oracle_video import digest  # noqa: E402

cases = [dict(w=192, h=128, tiles=1, seed=3, features=0, n=4), dict(w=448, h=256, tiles=4, seed=5, features=ir.SYN_SMOOTH, n=3),
         dict(w=640, h=360, tiles=1, seed=1, features=0, n=3), dict(w=320, h=192, tiles=2, seed=7, features=ir.SYN_ALL, n=3),
         dict(w=320, h=192, tiles=2, seed=8, features=ir.SYN_ALL | ir.SYN_WARP, n=3),   # warp: with the STAND-IN filter table
         dict(w=320, h=192, tiles=2, seed=9, features=ir.SYN_INTRABC | ir.SYN_RECT | ir.SYN_SMOOTH, n=2)]
out = {"note": "oracle regression digests (not reference output)", "cases": []}
for c in cases:
    p = ir.synth_params(c["w"], c["h"], c["tiles"], seed=c["seed"], features=c["features"])
    refs = [None] * 8
    ds = []
    if c["features"] & ir.SYN_WARP:
        ol.oracle_set_warp_filter(ir.synth_warp_filter())
    for fi in range(c["n"]):
        f = ir.synth_frame(p, fi)
        o = ol.oracle_decode_frame(f, refs)
        refs[f.desc.dst_slot] = o["final"]
        ds.append(digest(o["final"]))
    out["cases"].append(dict(w=c["w"], h=c["h"], tiles=c["tiles"], seed=c["seed"], features=c["features"], digests=ds))
json.dump(out, open(os.path.join(HERE, "video_regression.json"), "w"), indent=1)
print("wrote video_regression.json")
