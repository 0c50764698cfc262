import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, '.')
import easyav1_amd
easyav1_amd.load(build=True)
from easyav1_amd import av1parse, ir, video
from tests import av1_tables_gen as tg, streamgen, oracle_lib as ol
from tests import test_av1_block_parse as T
tables = av1parse.Tables(tg.make_blob(5))
env = (av1parse, streamgen, tables)
seed, size = int(sys.argv[1]), (int(sys.argv[2]), int(sys.argv[3]))
units, frames, headers, counts = T.make_stream(env, seed, size, 2)
f = frames[0]
d = f.desc
cw, ch = d.width - 8 * (d.pad & 15), d.height - 8 * (d.pad >> 4)
print("desc", d.width, d.height, hex(d.pad), hex(d.crop), "coded", cw, ch)
outs = ol.oracle_decode_frame(f, [None] * 8, ir.STAGE_ALL)
ctx = video.ReconContext(d.width, d.height)
for name, stages in (("recon", ir.STAGE_RECON), ("deblock", ir.STAGE_RECON | ir.STAGE_DEBLOCK), ("cdef", ir.STAGE_RECON | ir.STAGE_DEBLOCK | ir.STAGE_CDEF), ("final", ir.STAGE_ALL)):
    job = ctx.make_job(f)
    ctx.submit(job, stages)
    got = ctx.download(d.dst_slot)
    job.close()
    for pl in range(3):
        a = got[pl][:ch >> (pl > 0), :cw >> (pl > 0)].astype(int)
        b = outs[name][pl][:ch >> (pl > 0), :cw >> (pl > 0)].astype(int)
        bad = np.argwhere(a != b)
        if len(bad):
            y, x = bad[0]
            print(name, "plane", pl, "mismatches", len(bad), "first at", (int(x), int(y)), "gpu", a[y, x], "oracle", b[y, x], "bbox x", bad[:, 1].min(), bad[:, 1].max(), "y", bad[:, 0].min(), bad[:, 0].max())
            # the block covering it
            ss = 1 if pl else 0
            lx, ly = int(x) << ss, int(y) << ss
            for bi, blk in enumerate(f.blocks):
                bx, by = int(blk["x4"]) * 4, int(blk["y4"]) * 4
                if bx <= lx < bx + (1 << int(blk["bwl"])) and by <= ly < by + (1 << int(blk["bhl"])):
                    print("   block", bi, blk)
                    break
        else:
            print(name, "plane", pl, "ok")
