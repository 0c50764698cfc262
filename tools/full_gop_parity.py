"""One-off long parity run: whole GOPs at BASELINE sizes, every frame of every chain against the
oracle (errors would accumulate through the reference slots).  ~5 s of oracle per 1080p frame, so
this is not part of the routine test suite; the result is quoted in DESIGN.md."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402  (initialises the device through torch first, see tests/conftest.py)

assert torch.cuda.is_available()
from easyav1_amd import ir, video  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402


def run(W, H, tiles, nchains, nframes, features, seed0):
    t0 = time.time()
    ctx = video.ReconContext(W, H, nchains=nchains)
    params = [ir.synth_params(W, H, tiles, seed=seed0 + c, features=features) for c in range(nchains)]
    refs = [[None] * 8 for _ in range(nchains)]
    bad = 0
    for fi in range(nframes):
        frames = [ir.synth_frame(params[c], fi) for c in range(nchains)]
        jobs = [ctx.make_job(f) for f in frames]
        ctx.submit_batch(jobs)
        want = [None] * nchains

        def work(c):
            want[c] = ol.oracle_decode_frame(frames[c], refs[c])["final"]
        th = [threading.Thread(target=work, args=(c,)) for c in range(nchains)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in range(nchains):
            got = ctx.download(frames[c].desc.dst_slot, chain=c)
            ok = all(np.array_equal(want[c][pl], got[pl]) for pl in range(3))
            bad += 0 if ok else 1
            refs[c][frames[c].desc.dst_slot] = want[c]
        for j in jobs:
            j.close()
        print(f"{W}x{H} frame {fi:3d}: {'ok' if not bad else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
    ctx.close()
    return bad


if __name__ == "__main__":
    total = 0
    total += run(1920, 1080, 1, 4, 60, 0, 1000)          # graded set, 4 chains x one whole GOP
    total += run(1920, 1080, 2, 2, 24, 0x7FFF, 2000)     # every tool, 2 chains x 24 frames
    total += run(3840, 2160, 8, 2, 12, 0, 3000)          # 4K, 8 tile columns
    print("mismatching frames:", total)
    sys.exit(1 if total else 0)
