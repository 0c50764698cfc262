"""many libaom configurations through parser + oracle vs dav1d (build container only); prints one line per configuration"""
import sys, os, io, contextlib, itertools, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def one(cfg):
    from tests.golden import av1_ref
    from tools.dbg import real_av1
    name, kw = cfg
    kw = dict(kw)
    w, h, n, seed, kind = kw.pop("w", 104), kw.pop("h", 72), kw.pop("n", 4), kw.pop("seed", 1), kw.pop("kind", "natural")
    frames = av1_ref.synthetic_frames(seed, w, h, n, kind)
    try:
        units = av1_ref.encode_sequence(frames, **kw)
    except Exception as e:
        return name, "ENCODE FAILED %s" % e, ""
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            ok = real_av1.run(units, verbose=False)
    except Exception as e:
        return name, "EXC %r" % e, buf.getvalue()
    return name, "OK" if ok else "DIFF", "" if ok else buf.getvalue()

def configs():
    out = []
    for i, (w, h) in enumerate([(64, 64), (72, 56), (104, 72), (130, 98), (200, 136), (256, 144), (352, 288), (322, 242)]):
        for q in (30, 60, 90):
            out.append(("size %dx%d q%d" % (w, h, q), dict(w=w, h=h, n=5, seed=10 + i, quality=q)))
    for sp in (0, 2, 4, 8, 10):
        out.append(("speed %d" % sp, dict(w=136, h=104, n=5, seed=30 + sp, speed=sp)))
    for i in range(6):
        out.append(("slow %d" % i, dict(w=[104, 136, 200][i % 3], h=[72, 104, 136][i % 3], n=8, seed=70 + i, speed=i % 3, quality=[35, 55, 75, 90][i % 4])))
    for i in range(4):
        out.append(("slowscreen %d" % i, dict(w=200, h=136, n=6, seed=80 + i, speed=i % 3, kind="screen", quality=[40, 70][i % 2], advanced={"tune-content": "screen"})))
    out.append(("screen", dict(w=200, h=136, n=5, seed=40, kind="screen", advanced={"tune-content": "screen"})))
    out.append(("screen intrabc still", dict(w=200, h=136, n=1, seed=41, kind="screen", advanced={"tune-content": "screen", "enable-intrabc": 1})))
    out.append(("screen q30", dict(w=256, h=144, n=4, seed=42, kind="screen", quality=30, advanced={"tune-content": "screen"})))
    out.append(("mono", dict(w=136, h=104, n=4, seed=43, subsampling="4:0:0")))
    out.append(("lossless", dict(w=104, h=72, n=3, seed=44, quality=100)))
    out.append(("lossless adv", dict(w=104, h=72, n=3, seed=44, advanced={"lossless": 1})))
    out.append(("tiles 2x2", dict(w=352, h=288, n=4, seed=45, tile_cols_log2=1, tile_rows_log2=1)))
    out.append(("tiles 4 cols", dict(w=640, h=360, n=3, seed=46, tile_cols_log2=2)))
    out.append(("qm", dict(w=200, h=136, n=4, seed=47, advanced={"enable-qm": 1, "qm-min": 2, "qm-max": 10})))
    out.append(("deltaq", dict(w=200, h=136, n=4, seed=48, advanced={"deltaq-mode": 1})))
    out.append(("deltaq lf", dict(w=200, h=136, n=4, seed=49, advanced={"deltaq-mode": 1, "deltalf-mode": 1})))
    out.append(("aq-mode 1", dict(w=200, h=136, n=4, seed=50, advanced={"aq-mode": 1})))
    out.append(("aq-mode 2", dict(w=200, h=136, n=4, seed=51, advanced={"aq-mode": 2})))
    out.append(("aq-mode 3", dict(w=200, h=136, n=4, seed=52, advanced={"aq-mode": 3})))
    out.append(("sb128", dict(w=352, h=288, n=3, seed=53, advanced={"sb-size": "128"})))
    out.append(("sb64", dict(w=352, h=288, n=3, seed=54, advanced={"sb-size": "64"})))
    out.append(("superres fixed", dict(w=200, h=136, n=3, seed=55, advanced={"superres-mode": 1, "superres-denominator": 12, "superres-kf-denominator": 12})))
    out.append(("superres random", dict(w=200, h=136, n=4, seed=56, advanced={"superres-mode": 2})))
    out.append(("resize", dict(w=200, h=136, n=4, seed=57, advanced={"resize-mode": 1, "resize-denominator": 12})))
    out.append(("error resilient", dict(w=136, h=104, n=4, seed=58, advanced={"error-resilient-mode": 1})))
    out.append(("cdf update off", dict(w=136, h=104, n=4, seed=59, advanced={"cdf-update-mode": 0})))
    out.append(("reduced refs", dict(w=136, h=104, n=6, seed=60, advanced={"reduced-reference-set": 1})))
    out.append(("long 16", dict(w=136, h=104, n=16, seed=61)))
    out.append(("long 24 q80", dict(w=200, h=136, n=24, seed=62, quality=80)))
    return out

if __name__ == "__main__":
    cfgs = configs()
    if len(sys.argv) > 1:
        cfgs = [c for c in cfgs if any(a in c[0] for a in sys.argv[1:])]
    with mp.Pool(7) as pool:
        for name, res, log in pool.imap(one, cfgs):
            print("%-28s %s" % (name, res), flush=True)
            if log and os.environ.get("V"):
                print(log)
