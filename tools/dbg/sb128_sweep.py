"""Generated streams with superblocks of 128x128 (blocks of 128 samples) over many seeds, sizes and generator policies:
tile bytes -> block parser -> hand-over -> HIP reconstruction against the oracle.   python tools/dbg/sb128_sweep.py [seeds]"""
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
from tests import streamgen  # noqa: E402
from tests import test_av1_block_parse as T  # noqa: E402

env = (av1parse, streamgen, av1parse.Tables(tg.make_blob(5)))
sizes = [(256, 128), (200, 136), (320, 264), (392, 200), (166, 102), (520, 392)]
policies = [dict(split_bias=5), dict(split_bias=10), dict(split_bias=25), dict(split_bias=8, inter=90), dict(split_bias=8, inter=20), dict(split_bias=10, skip=70)]
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 36
bad = 0
big = 0
for k in range(nseeds):
    size, pol = sizes[k % len(sizes)], policies[(k // len(sizes)) % len(policies)]
    units, frames, headers, counts = T.make_stream(env, 1000 + k, size, 5, seq_kw=dict(use_128x128_superblock=1), policy=pol)
    hist = T.blocks_by_size(frames)
    n128 = hist[(128, 128)] + hist[(128, 64)] + hist[(64, 128)]
    big += n128
    parser = av1parse.FrameParser(env[2])
    parsed = []
    for tu in units:
        rc, f, info = parser.parse(tu)
        assert rc == 0, easyav1_amd._lib.last_error()
        parsed.append(f)
    parser.close()
    want = T.oracle_pictures(parsed)
    ctx = video.ReconContext(parsed[0].desc.width, parsed[0].desc.height)
    ok = True
    for n, f in enumerate(parsed):
        job = ctx.make_job(f)
        ctx.submit(job)
        got = T.coded_part(f.desc, ctx.download(f.desc.dst_slot))
        job.close()
        exp = T.coded_part(f.desc, want[n])
        for pl in range(3):
            if not np.array_equal(got[pl], exp[pl]):
                ok = False
                print("MISMATCH seed", 1000 + k, size, pol, "frame", n, "plane", pl, int((got[pl] != exp[pl]).sum()), flush=True)
    ctx.close()
    bad += not ok
    print(f"seed {1000 + k} {size} {pol}: {n128} blocks of 128 samples, {'ok' if ok else 'BAD'}", flush=True)
print("streams", nseeds, "bad", bad, "blocks of 128 samples", big)
sys.exit(1 if bad else 0)
