"""Host rate of the block-level AV1 parser (symbol decode + context derivation + hand-over records) on ONE thread:
  python tools/av1_parse_rate.py [--size 1920x1080] [--frames 8] [--seed 1] [--skip 55 --all-zero 60]
The stream comes from the generator of the tests (tests/native/av1_streamgen.cpp: the same syntax walked with a symbol writer) with
arbitrary tables, so its statistics are the policy's, not a camera's: the figure to read is symbols (and bits) per second, the
frame rate follows from how dense a stream is.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(width, height, nframes, seed, policy, repeats=3):
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import av1parse
    from tests import av1_tables_gen as tg
    from tests import streamgen
    from tests import test_av1_block_parse as T
    tables = av1parse.Tables(tg.make_blob(5))
    env = (av1parse, streamgen, tables)
    units, frames, headers, counts = T.make_stream(env, seed, (width, height), nframes, policy=policy,
                                                   hdr_kw=dict(base_q_idx=96, one_tile=True))
    symbols = int(sum(counts))
    nbytes = sum(len(u) for u in units)
    best = None
    for _ in range(repeats):
        p = av1parse.FrameParser(tables)
        secs = 0.0
        t0 = time.perf_counter()
        for tu in units:
            rc, f, info = p.parse(tu)
            assert rc == 0
            secs += info.parse_seconds
        wall = time.perf_counter() - t0
        p.close()
        best = secs if best is None else min(best, secs)
    blocks = sum(len(f.blocks) for f in frames)
    coefs = sum(int((f.coefs != 0).sum()) for f in frames)
    return dict(workload=f"{width}x{height}", frames=nframes, stream_bytes=nbytes, bits_per_frame=round(8 * nbytes / nframes), syntax_elements=symbols,
                blocks_per_frame=round(blocks / nframes), nonzero_coefs_per_frame=round(coefs / nframes), parse_seconds=round(best, 6),
                frames_per_second_per_thread=round(nframes / best, 2), elements_per_second=round(symbols / best), mbit_per_second=round(8e-6 * nbytes / best, 2),
                python_wall_seconds=round(wall, 4), tiles_per_frame=1, policy=policy, tables="arbitrary (tests/av1_tables_gen.py)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--skip", type=int, default=55)
    ap.add_argument("--all-zero", type=int, default=60)
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    print(json.dumps(measure(w, h, a.frames, a.seed, dict(skip=a.skip, all_zero=a.all_zero, inter=80, max_mv_class=4, split_bias=25))))
