"""What the GPU does with REAL 4K frames (the committed libaom GOP): per frame the intra wavefront's shape (levels, widest level, items) and per
kernel the time of the GOP on 1 chain and on 16 chains (the sink's batch shape): python tools/dbg/real_levels.py [fixture name]   (GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
import easyav1_amd  # noqa: E402

easyav1_amd.load(build=False)
from easyav1_amd import av1parse, video  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bench_uhd_4cols_gop16"
z = np.load(os.path.join(ROOT, "tests", "golden", "av1_stream_%s.npz" % name))
raw, sizes = z["units"].tobytes(), z["unit_sizes"]
units, p = [], 0
for s in sizes:
    units.append(raw[p:p + int(s)])
    p += int(s)
blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
tables = av1parse.Tables(blob)
parser = av1parse.FrameParser(tables)
frames = []
for tu in units:
    for f, info in parser.parse_unit(tu):
        if f is not None:
            frames.append(f)
W = max(f.desc.width for f in frames)
H = max(f.desc.height for f in frames)
print("%s: %d frames of %dx%d" % (name, len(frames), W, H))
for S in (1, 16):
    ctx = video.ReconContext(W, H, 0, nchains=S)
    jobs = [[ctx.make_job(f) for f in frames] for _ in range(S)]   # chain c decodes the same GOP
    if S == 1:
        for i, j in enumerate(jobs[0]):
            si, st = j.schedule_info(), j.stats()
            print("  frame %2d: type %d  levels %4d  widest level %5d  intra items %6d  inter tiles %s  coefficients %d" % (
                i, frames[i].desc.frame_type if hasattr(frames[i].desc, "frame_type") else -1, si["num_levels"], si["max_level_items"], si["intra_items"], si["mc_tiles"], st["coef_count"]))
    def gop():
        for i in range(len(frames)):
            ctx.submit_batch([jobs[c][i] for c in range(S)])
    gop()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        gop()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%d chain(s): %.2f ms per GOP of %d frames = %.0f frames/s" % (S, dt * 1e3, len(frames), S * len(frames) / dt))
    ctx.profile(True)
    gop()
    prof = ctx.profile_read()
    tot = sum(e["ms_total"] for e in prof)
    print("   kernel time by the library's events (one GOP, %d chains): %.2f ms" % (S, tot))
    for e in sorted(prof, key=lambda e: -e["ms_total"])[:14]:
        print("     %-48s %8.3f ms  %5d launches" % (e["name"][:48], e["ms_total"], e["launches"]))
    ctx.profile(False)
    if S == 16:   # what a drifted sink submits: ONE chain at its key frame, the others at inter frames
        mixed = [jobs[0][0]] + [jobs[c][1 + c % 15] for c in range(1, S)]
        ctx.submit_batch(mixed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.submit_batch(mixed)
        torch.cuda.synchronize()
        print("   one key frame + 15 inter frames in a batch: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
        ctx.profile(True)
        ctx.submit_batch(mixed)
        prof = ctx.profile_read()
        for e in sorted(prof, key=lambda e: -e["ms_total"])[:6]:
            print("     %-48s %8.3f ms  %5d launches" % (e["name"][:48], e["ms_total"], e["launches"]))
        ctx.profile(False)
        inter = [jobs[c][1 + c % 15] for c in range(S)]
        ctx.submit_batch(inter)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.submit_batch(inter)
        torch.cuda.synchronize()
        print("   16 inter frames in a batch: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
        ctx.profile(True)
        ctx.submit_batch(inter)
        prof = ctx.profile_read()
        for e in sorted(prof, key=lambda e: -e["ms_total"])[:6]:
            print("     %-48s %8.3f ms  %5d launches" % (e["name"][:48], e["ms_total"], e["launches"]))
        ctx.profile(False)
    for js in jobs:
        for j in js:
            j.close()
    ctx.close()
