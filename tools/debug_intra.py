"""Find the first transform blocks whose reconstruction differs from the oracle (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from easyav1_amd import ir, video
from tests import oracle_lib as ol

W, H, feats, seed = 320, 192, int(sys.argv[1]) if len(sys.argv) > 1 else 251, int(sys.argv[2]) if len(sys.argv) > 2 else 40 + (251 & 63)
p = ir.synth_params(W, H, 2, seed=seed, gop=2, features=feats)
ctx = video.ReconContext(W, H)
f = ir.synth_frame(p, 0)
want = ol.oracle_decode_frame(f, [None] * 8)
job = ctx.make_job(f)
ctx.submit(job, 1)
got = ctx.download(f.desc.dst_slot)
seen = set()
for pl in range(3):
    bad = np.argwhere(want["recon"][pl] != got[pl])
    print("plane", pl, "bad", len(bad))
    for y, x in bad[:4000]:
        for ti, t in enumerate(f.txs):
            if t["plane"] != pl:
                continue
            tw, th = 1 << (int(t["size"]) & 15), 1 << (int(t["size"]) >> 4)
            if t["x4"] * 4 <= x < t["x4"] * 4 + tw and t["y4"] * 4 <= y < t["y4"] * 4 + th:
                if ti in seen:
                    break
                seen.add(ti)
                bi = [i for i, b in enumerate(f.blocks) if b["first_tx"] <= ti < b["first_tx"] + b["n_tx"]][0]
                b = f.blocks[bi]
                if len(seen) < 12:
                    print(f"tx {ti} pl {pl} at ({t['x4']*4},{t['y4']*4}) {tw}x{th} first bad ({x},{y}) | block {bi} at ({b['x4']*4},{b['y4']*4}) "
                          f"{1<<b['bwl']}x{1<<b['bhl']} flags {int(b['flags']):#x} mode_y {b['mode_y']} mode_uv {b['mode_uv']} "
                          f"delta {b['angle_delta']} fi {b['filter_intra']} pal {b['pal_size']} alpha {b['cfl_alpha']}")
                break
