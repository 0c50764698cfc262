"""Build container only: the streams of the round-5 robustness sweep (tools/real_stream_sweep.py) -- the bench content (tests/golden/make_video_fixtures.py
"bench_uhd_4cols_gop16": 3840x2160, 4 tile columns, libaom speed 10) at other qualities and GOP lengths, each ONE closed GOP in the fixture format of
tests/golden (units + dav1d's shown-picture digests; no per-stage digests).  Written to tools/bin/sweep/ (not committed: 30 MB; this script remakes them).
    python tools/make_sweep_streams.py [name ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden import av1_ref  # noqa: E402
from tests.golden.make_video_fixtures import sha  # noqa: E402

OUT = os.path.join(ROOT, "tools", "bin", "sweep")
BASE = dict(w=3840, h=2160, seed=28, speed=10, tile_cols_log2=2, kind="tiled_smooth", threads=8)
STREAMS = {
    "uhd_q60_gop16": dict(BASE, n=16, quality=60),
    "uhd_q80_gop16": dict(BASE, n=16, quality=80),
    "uhd_q40_gop60": dict(BASE, n=60, quality=40),
    "uhd_q40_gop120": dict(BASE, n=120, quality=40),
    "uhd_q40_gop250": dict(BASE, n=250, quality=40),
    "uhd_noisy_q40_gop60": dict(BASE, n=60, quality=40, kind="tiled"),      # sensor-like noise: inter frames that carry bits
    "uhd_noisy_q60_gop60": dict(BASE, n=60, quality=60, kind="tiled"),
    # the other chroma layouts (round 5): 1080p, the content and options of the 1080p bench stream
    "hd444_q45_gop16": dict(w=1920, h=1080, seed=29, speed=9, kind="tiled_smooth", threads=8, n=16, quality=45, subsampling="4:4:4"),
    "hd422_q45_gop16": dict(w=1920, h=1080, seed=29, speed=9, kind="tiled_smooth", threads=8, n=16, quality=45, subsampling="4:2:2"),
    "hd420_q45_gop16": dict(w=1920, h=1080, seed=29, speed=9, kind="tiled_smooth", threads=8, n=16, quality=45),
}


def make(name, cfg):
    cfg = dict(cfg)
    w, h, n, seed, kind = cfg.pop("w"), cfg.pop("h"), cfg.pop("n"), cfg.pop("seed"), cfg.pop("kind")
    t0 = time.time()
    frames = av1_ref.synthetic_frames(seed, w, h, n, kind)
    units = av1_ref.encode_sequence(frames, **cfg)
    del frames
    shown = av1_ref.dav1d_pictures(units)
    a = {"format": np.array([10], np.uint32), "size": np.array([w, h], np.uint32), "layout": np.array([{"4:4:4": 3, "4:2:2": 2}.get(cfg.get("subsampling"), 1)], np.uint32),
         "units": np.frombuffer(b"".join(units), np.uint8), "unit_sizes": np.array([len(u) for u in units], np.uint32),
         "shown_sha": np.stack([sha(p) for p in shown])}
    for stage in ("recon", "deblock", "cdef", "final"):   # (the loader wants the keys)
        a["all_%s_sha" % stage] = np.zeros((0, 32), np.uint8)
    a["meta"] = np.frombuffer(json.dumps(dict(name=name, width=w, height=h, frames_in=n, content=kind, seed=seed, encoder_options={k: str(v) for k, v in cfg.items()},
                                              shown=len(shown), versions=av1_ref.versions()), sort_keys=True).encode(), np.uint8)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "av1_stream_%s.npz" % name)
    np.savez_compressed(path, **a)
    print("%-24s %3d units, %9d bytes of AV1 (%7d bits / frame), %d shown, %.0f s" % (name, len(units), sum(len(u) for u in units), 8 * sum(len(u) for u in units) // len(units), len(shown), time.time() - t0), flush=True)


if __name__ == "__main__":
    for nm in sys.argv[1:] or list(STREAMS):
        make(nm, STREAMS[nm])
