import sys, time
sys.path.insert(0, '.')
from easyav1_amd import ir, video
W, H = 1920, 1080
ctx = video.ReconContext(W, H)
p = ir.synth_params(W, H, 1, seed=1)
frames = [ir.synth_frame(p, fi) for fi in (0, 1, 2, 3)]
for rep in range(2):
    for f in frames:
        t = time.perf_counter()
        j = ctx.make_job(f)
        dt = time.perf_counter() - t
        print("frame inter=%d blocks=%d txs=%d: job_create %.2f ms" % (f.desc.is_inter, len(f.blocks), len(f.txs), dt * 1e3))
        j.close()
