import sys, numpy as np
sys.path.insert(0, '.')
from easyav1_amd import ir, video
from tests import oracle_lib as ol
W, H = 3840, 2160
p = ir.synth_params(W, H, 4, seed=21)
ctx = video.ReconContext(W, H)
f0 = ir.synth_frame(p, 0)
j0 = ctx.make_job(f0); ctx.submit(j0)
w0 = ol.oracle_decode_frame(f0, [None]*8)
g0 = ctx.download(f0.desc.dst_slot)
print("frame0 equal", all(np.array_equal(a, b) for a, b in zip(w0["final"], g0)))
refs = [None]*8; refs[f0.desc.dst_slot] = w0["final"]
f1 = ir.synth_frame(p, 1)
w1 = ol.oracle_decode_frame(f1, refs)
j1 = ctx.make_job(f1)
for stages, name in ((1, "recon"), (3, "deblock"), (7, "cdef"), (15, "final")):
    ctx.set_ref(f0.desc.dst_slot, w0["final"])
    ctx.submit(j1, stages)
    g = ctx.download(f1.desc.dst_slot)
    for pl in range(3):
        d = g[pl].astype(int) - w1[name][pl].astype(int)
        if (d != 0).any():
            ys, xs = np.nonzero(d)
            print(name, "plane", pl, "mismatch", len(ys), list(zip(ys[:6].tolist(), xs[:6].tolist())), d[d != 0][:6])
print("done")
y, x = 1051, 1790
for name in ("recon", "deblock", "cdef", "final"):
    print(name); print(w1[name][2][y-3:y+4, x-5:x+4])
# block info around chroma (1790,1051) -> luma (3580, 2102)
b = f1.blocks
for blk in b:
    bx, by, n = int(blk["x4"])*4, int(blk["y4"])*4, 1 << int(blk["bwl"])
    if bx <= 3580 < bx+n and by <= 2102 < by+n:
        print("block", bx, by, n, blk)
