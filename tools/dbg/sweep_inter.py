import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden import av1_ref
from tools.dbg import real_av1
OFF = {"enable-obmc": 0, "enable-warped-motion": 0, "enable-global-motion": 0, "enable-ref-frame-mvs": 0, "enable-interintra-comp": 0,
       "enable-masked-comp": 0, "enable-dist-wtd-comp": 0, "enable-dual-filter": 0, "enable-onesided-comp": 0, "enable-diff-wtd-comp": 0,
       "enable-interinter-wedge": 0, "enable-interintra-wedge": 0, "enable-smooth-interintra": 0, "enable-filter-intra": 0, "enable-cfl-intra": 0,
       "enable-palette": 0, "enable-intrabc": 0, "enable-angle-delta": 0, "enable-tx64": 0, "enable-flip-idtx": 0, "enable-rect-tx": 0, "enable-rect-partitions": 0,
       "enable-ab-partitions": 0, "enable-1to4-partitions": 0, "enable-intra-edge-filter": 0, "enable-smooth-intra": 0, "enable-paeth-intra": 0,
       "enable-cdef": 0, "enable-restoration": 0, "enable-tpl-model": 0, "deltaq-mode": 0, "enable-qm": 0, "reduced-tx-type-set": 1, "max-reference-frames": 3,
       "enable-order-hint": 1, "enable-chroma-deltaq": 0}
def trial(name, adv, w=104, h=72, n=3, seed=1, **kw):
    frames = av1_ref.synthetic_frames(seed, w, h, n, kw.pop("kind", "natural"))
    try:
        units = av1_ref.encode_sequence(frames, advanced=adv, **kw)
    except Exception as e:
        print("%-40s ENCODE FAILED %s" % (name, e)); return None
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok = real_av1.run(units, verbose=False, only=("recon", "final"))
    print("%-40s %s" % (name, "OK" if ok else "DIFF"))
    if not ok and os.environ.get("V"):
        print(buf.getvalue())
    return ok
if __name__ == "__main__":
    base = dict(OFF)
    trial("all off", base)
    for k in OFF:
        adv = dict(base)
        adv[k] = 1 - OFF[k] if k not in ("max-reference-frames",) else 7
        trial(k + "=" + str(adv[k]), adv)
