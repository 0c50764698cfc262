import sys
sys.path.insert(0, '.')
from easyav1_amd import ir, video
W, H = 1920, 1080
ctx = video.ReconContext(W, H)
p = ir.synth_params(W, H, 1, seed=1)
for fi in (0, 1, 2, 30):
    j = ctx.make_job(ir.synth_frame(p, fi))
    print(fi, j.schedule_info())
