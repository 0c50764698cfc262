"""Host cost of the hand-over of one parsed frame (ea1_job_create): ms per frame and thread, and the rate T threads
sustain.  Needs the GPU (the call uploads).   python tools/time_job_create.py [1080p|4k] [threads]"""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, '.')
import torch  # noqa: F401  (device first through torch, see tests/conftest.py)

torch.cuda.init()
from easyav1_amd import ir, video

which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H, tiles = {"1080p": (1920, 1080, 1), "4k": (3840, 2160, 4)}[which]
ctx = video.ReconContext(W, H)
p = ir.synth_params(W, H, tiles, seed=1)
frames = [ir.synth_frame(p, fi) for fi in range(8)]
for f in frames:            # warm the pool
    ctx.make_job(f).close()
per = []
for rep in range(3):
    for f in frames:
        t = time.perf_counter()
        j = ctx.make_job(f)
        per.append((time.perf_counter() - t, f.desc.is_inter))
        j.close()
per.sort()
print("%s: ea1_job_create, one thread: median %.3f ms, min %.3f ms, max %.3f ms per frame (%d blocks, %d transform blocks, %.1f MB of coefficients)" % (
    which, per[len(per) // 2][0] * 1e3, per[0][0] * 1e3, per[-1][0] * 1e3, len(frames[1].blocks), len(frames[1].txs), frames[1].coefs.nbytes / 1e6))
N = 40 * T
with ThreadPoolExecutor(T) as ex:
    def one(k):
        ctx.make_job(frames[k % len(frames)]).close()
    list(ex.map(one, range(T)))
    t = time.perf_counter()
    list(ex.map(one, range(N)))
    dt = time.perf_counter() - t
print("%s: %d threads: %.0f frames/s handed over (%.3f ms per frame and thread)" % (which, T, N / dt, dt * T / N * 1e3))
torch.cuda.synchronize()
ctx.close()
