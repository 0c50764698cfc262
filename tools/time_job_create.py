"""Host cost of the hand-over of one parsed frame (ea1_job_create): ms per frame and thread, and the rate T threads
of the library sustain (ea1_job_create_batch), coefficients in pageable and in pinned host memory.
Needs the GPU (the call uploads).   python tools/time_job_create.py [1080p|4k] [threads]"""
import sys
import time

sys.path.insert(0, '.')
import numpy as np
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


def batch_rate(tag, fr):
    n = 4 * T
    for _ in range(2):
        for j in ctx.make_jobs([fr[k % len(fr)] for k in range(n)], T):
            j.close()
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        for j in ctx.make_jobs([fr[k % len(fr)] for k in range(n)], T):
            j.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%s: %d library threads, coefficients in %s memory: %.0f frames/s handed over (%.3f ms per frame and thread, %.1f GB/s to the device)" % (
        which, T, tag, n * reps / dt, dt * T / (n * reps) * 1e3, n * reps / dt * fr[1].coefs.nbytes / 1e9))


batch_rate("pageable", frames)
pins = []
for f in frames:
    pb = video.PinnedBuffer(f.coefs.nbytes)
    a = np.frombuffer(pb.array(), dtype=np.int16, count=len(f.coefs))
    a[:] = f.coefs
    f.coefs = a
    f._ir_struct = None
    pins.append(pb)
batch_rate("pinned", frames)
for T2 in (1, 4, 8):
    if T2 < T:
        T_save, T = T, T2
        batch_rate("pinned", frames)
        T = T_save
ctx.close()
