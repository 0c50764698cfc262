import sys, numpy as np
sys.path.insert(0, '.')
import torch
from easyav1_amd import ir, video
from tests import oracle_lib as ol
W, H = 192, 128
p = ir.synth_params(W, H, 1, seed=3)
f = ir.synth_frame(p, 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "asis"
if mode == "hident":
    f.lr[0]["coef"][:, 1, :] = 0
if mode == "both":
    f.lr[0]["coef"][:] = 0
if mode == "vident":
    f.lr[0]["coef"][:, 0, :] = 0
want = ol.oracle_decode_frame(f, None)
ctx = video.ReconContext(W, H)
job = ctx.make_job(f)
ctx.submit(job)
got = ctx.download(f.desc.dst_slot)
d = got[0].astype(int) - want["final"][0].astype(int)
print(mode, "mismatch", (d != 0).sum(), "by col%4", [(d[:, c::4] != 0).sum() for c in range(4)], "vals", np.unique(d))
print(f.lr[0][:4])
ys, xs = np.nonzero(d)
print("got-cdef at mismatches:", np.unique((got[0].astype(int) - want["cdef"][0].astype(int))[d != 0])[:20])
for y, x in list(zip(ys, xs))[:3]:
    print(y, x, want["final"][0][y, x], got[0][y, x], want["cdef"][0][max(0,y-1):y+2, max(0,x-3):x+4])
