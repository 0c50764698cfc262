"""Generated streams that combine the late additions -- superblocks of 128x128, frames of changing size, super-resolution, sequences with
tools switched off, hidden frames -- over many seeds: tile bytes -> block parser -> hand-over -> HIP reconstruction against the oracle.
   python tools/dbg/feature_sweep.py [streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import easyav1_amd  # noqa: E402

easyav1_amd.load(build=False)
from easyav1_amd import av1parse, video  # noqa: E402
from tests import av1_tables_gen as tg  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402
from tests import streamgen  # noqa: E402
from tests import test_av1_block_parse as T  # noqa: E402

env = (av1parse, streamgen, av1parse.Tables(tg.make_blob(5)))
taps = tg.upscale_filter()
ol.oracle_set_upscale_filter(taps)
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sizes = [(264, 200), (200, 136), (392, 264), (166, 102), (320, 192), (520, 280)]
bad = 0
totals = dict(sr=0, resized=0, big=0)
for k in range(nstreams):
    seq_kw, hdr_kw, policy = {}, {}, {}
    if k & 1:
        seq_kw["use_128x128_superblock"] = 1
        policy["split_bias"] = 10
    if k & 2:
        seq_kw["enable_superres"] = 1
    size = sizes[k % len(sizes)]
    if k & 4:
        hdr_kw.update(frame_size_override_flag=1, frame_size_range=(size[0] * 5 // 8, size[1] * 5 // 8))
    if k % 5 == 4:
        seq_kw.update(T.TOOLS_OFF["seq_kw"])
    if k % 7 == 6:
        seq_kw.update(T.NO_ORDER_HINT["seq_kw"])
    units, frames, headers, counts = T.make_stream(env, 2000 + k, size, 6, seq_kw=seq_kw, hdr_kw=hdr_kw, policy=policy or None, gop=3 if k % 3 == 0 else None)
    hist = T.blocks_by_size(frames)
    totals["big"] += hist[(128, 128)] + hist[(128, 64)] + hist[(64, 128)]
    totals["sr"] += sum(1 for f in frames if f.desc.upscaled_width)
    totals["resized"] += len({(f.desc.width, f.desc.height, f.desc.pad, f.desc.crop) for f in frames}) > 1
    parser = av1parse.FrameParser(env[2])
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0, easyav1_amd._lib.last_error()
        parsed.append(f)
    parser.close()
    want = T.oracle_pictures(parsed)
    ctx = video.ReconContext(max(max(f.desc.width, (f.desc.upscaled_width + 7) & ~7) for f in parsed), max(f.desc.height for f in parsed))
    ctx.set_upscale_filter(taps)
    ok = True
    for n, f in enumerate(parsed):
        try:
            job = ctx.make_job(f)
            ctx.submit(job)
        except Exception as e:   # a legal combination the reconstruction does not cover is reported, not hidden
            if "beyond the 2 : 1" in str(e):   # (the generator does not keep to the limits of scaled references: such a stream is not a legal one)
                print("stream", 2000 + k, "frame", n, "leaves the 2 : 1 / 1 : 16 limits of scaled references: skipped from here", flush=True)
                break
            print("REFUSED stream", 2000 + k, "frame", n, str(e)[:160], flush=True)
            ok = False
            break
        got = T.visible_part(f.desc, ctx.download(f.desc.dst_slot))
        job.close()
        exp = T.visible_part(f.desc, want[n])
        for pl in range(3):
            if got[pl].shape != exp[pl].shape or not np.array_equal(got[pl], exp[pl]):
                ok = False
                print("MISMATCH stream", 2000 + k, size, seq_kw, hdr_kw, "frame", n, "plane", pl, flush=True)
    ctx.close()
    bad += not ok
    print(f"stream {2000 + k} {size} sb128={k & 1} superres={(k >> 1) & 1} resized={(k >> 2) & 1}: {'ok' if ok else 'BAD'}", flush=True)
print("streams", nstreams, "bad", bad, totals)
sys.exit(1 if bad else 0)
