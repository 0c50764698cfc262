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
ol.oracle()   # built / loaded once, before the worker threads call into it


def run(W, H, tiles, nchains, nframes, features, seed0, grain_every=0, **overrides):
    t0 = time.time()
    ctx = video.ReconContext(W, H, nchains=nchains)
    if features & ir.SYN_WARP:
        ctx.set_warp_filter(ir.synth_warp_filter())
        ol.oracle_set_warp_filter(ir.synth_warp_filter())
    if grain_every:
        from easyav1_amd import grain
        grain.set_gaussian_sequence(ctx, grain.synth_gaussian_sequence())
        ol.oracle_set_gaussian_sequence(grain.synth_gaussian_sequence())
        grng = np.random.default_rng(seed0)
    params = [ir.synth_params(W, H, tiles, seed=seed0 + c, features=features, **overrides) for c in range(nchains)]
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
        if grain_every and fi % grain_every == 0:   # film grain on the output copies (oracle: one picture at a time, it keeps static state)
            fgs = [grain.random_params(grng) for _ in range(nchains)]
            grain.film_grain_batch(ctx, [f.desc.dst_slot for f in frames], fgs)
            for c in range(nchains):
                g_want = ol.oracle_film_grain(fgs[c], want[c])
                g_got = grain.download_display(ctx, chain=c)
                bad += 0 if all(np.array_equal(g_want[pl], g_got[pl]) for pl in range(3)) else 1
        for j in jobs:
            j.close()
        print(f"{W}x{H} frame {fi:3d}: {'ok' if not bad else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
    ctx.close()
    return bad


if __name__ == "__main__":
    total = 0
    only_new = len(sys.argv) > 1 and sys.argv[1] == "--new-tools"
    if not only_new:
        total += run(1920, 1080, 1, 4, 60, 0, 1000)          # graded set, 4 chains x one whole GOP
        total += run(1920, 1080, 2, 2, 24, 0x7FFF, 2000)     # every tool, 2 chains x 24 frames
        total += run(3840, 2160, 8, 2, 12, 0, 3000)          # 4K, 8 tile columns
    # every tool incl. warped motion, intra block copy (an intra frame every 4th) and film grain on every 3rd output
    total += run(1920, 1080, 2, 2, 16, 0x7FFF | ir.SYN_WARP | ir.SYN_INTRABC, 4000, grain_every=3, gop=4)
    total += run(3840, 2160, 4, 1, 4, 0x7FFF | ir.SYN_WARP | ir.SYN_INTRABC, 5000, grain_every=2, gop=2)
    print("mismatching frames:", total)
    sys.exit(1 if total else 0)
