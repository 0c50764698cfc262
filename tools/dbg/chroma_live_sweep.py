"""Fresh libaom streams in 4:4:4 and 4:2:2 through the product's GPU path against dav1d (where both Pillow's libavif and a GPU are at hand: the GPU box).

    python tools/dbg/chroma_live_sweep.py encode DIR [name filter ...] && python tools/dbg/chroma_live_sweep.py check DIR

Every configuration of tools/dbg/stress_real.py (bar the ones its encoder refuses) plus a few of its own, once per layout: bytes -> block parser -> hand-over ->
a context of that layout -> HIP kernels (-> film grain) == dav1d 1.5.3, every frame including the hidden ones, and the pictures as shown.  Encoding and
dav1d run in worker processes (CPU), the GPU is used by this process alone."""
import sys, os, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

SUB = {"4:4:4": (0, 0), "4:2:2": (1, 0), "4:2:0": (1, 1)}


def configs():
    from tools.dbg import stress_real
    out = []
    for name, kw in stress_real.configs():
        if kw.get("subsampling") or name in ("deltaq lf", "superres fixed", "superres random", "resize", "error resilient"):
            continue
        out.append((name, kw))
    out.append(("objects slow", dict(w=264, h=200, n=10, seed=90, kind="objects", speed=1, quality=55)))
    out.append(("objects q80", dict(w=264, h=200, n=8, seed=91, kind="objects", speed=2, quality=80)))
    out.append(("restoration", dict(w=352, h=288, n=4, seed=92, speed=1, quality=45, advanced={"enable-restoration": 1})))
    out.append(("grain 25", dict(w=136, h=104, n=4, seed=93, advanced={"film-grain-test": 3})))
    out.append(("grain 50 odd", dict(w=130, h=98, n=4, seed=94, advanced={"film-grain-test": 8})))
    out.append(("grain table 11", dict(w=200, h=136, n=3, seed=95, advanced={"film-grain-test": 11})))
    out.append(("deltalf", dict(w=200, h=136, n=4, seed=96, advanced={"deltaq-mode": 1, "delta-lf-mode": 1})))
    out.append(("error resilient", dict(w=136, h=104, n=4, seed=97, advanced={"error-resilient": 1})))
    out.append(("screen palette intrabc", dict(w=264, h=200, n=3, seed=98, kind="screen", speed=1, advanced={"tune-content": "screen", "enable-intrabc": 1, "enable-palette": 1})))
    out.append(("hd 640x360 objects", dict(w=640, h=360, n=6, seed=99, kind="objects", speed=3)))
    return [("%s %s" % (sub, name), dict(kw, subsampling=sub)) for sub in ("4:4:4", "4:2:2") for name, kw in out]


def encode_and_reference(cfg):
    from tests.golden import av1_ref
    name, kw = cfg
    kw = dict(kw)
    w, h, n, seed, kind = kw.pop("w", 104), kw.pop("h", 72), kw.pop("n", 4), kw.pop("seed", 1), kw.pop("kind", "natural")
    try:
        units = av1_ref.encode_sequence(av1_ref.synthetic_frames(seed, w, h, n, kind), **kw)
    except Exception as e:
        return name, None, "ENCODE FAILED %s" % e
    final = av1_ref.dav1d_pictures(units, inloop=7, apply_grain=0, all_frames=True)
    shown = av1_ref.dav1d_pictures(units, inloop=7, apply_grain=1, all_frames=False)
    return name, dict(units=units, w=w, h=h, sub=kw["subsampling"], final=final, shown=shown), ""


def gpu_check(fx, av1parse, tables, gauss):
    from easyav1_amd import video, ir, grain
    ssx, ssy = SUB[fx["sub"]]
    layout = {"4:4:4": ir.LAYOUT_444, "4:2:2": ir.LAYOUT_422, "4:2:0": ir.LAYOUT_420}[fx["sub"]]
    w, h = fx["w"], fx["h"]
    vis = lambda planes: [planes[0][:h, :w], planes[1][:(h + ssy) >> ssy, :(w + ssx) >> ssx], planes[2][:(h + ssy) >> ssy, :(w + ssx) >> ssx]]
    parser = av1parse.FrameParser(tables)
    parsed = [parser.parse_unit(tu) for tu in fx["units"]]
    parser.close()
    frames = [f for fr in parsed for f, info in fr if f is not None]
    ctx = video.ReconContext(max(f.desc.width for f in frames), max(f.desc.height for f in frames), layout=layout)
    grain.set_gaussian_sequence(ctx, gauss)
    k_all = k_shown = 0
    census = dict(frames=0, grain=0)
    try:
        for ti, fr in enumerate(parsed):
            for f, info in fr:
                if f is None:
                    slot, shown = info.frame.desc.dst_slot, True
                else:
                    job = ctx.make_job(f)
                    ctx.submit(job)
                    job.close()
                    slot, shown = f.desc.dst_slot, bool(info.show_frame)
                    census["frames"] += 1
                got = vis(ctx.download(slot))
                want = fx["final"][k_all]
                for pl in range(3):
                    if not np.array_equal(got[pl], want[pl]):
                        d = np.argwhere(got[pl] != want[pl])
                        return "DIFF frame %d (unit %d) plane %d: %d samples, first (y=%d,x=%d) got %d want %d" % (k_all, ti, pl, len(d), d[0][0], d[0][1], got[pl][d[0][0], d[0][1]], want[pl][d[0][0], d[0][1]]), census
                k_all += 1
                if shown:
                    pic = got
                    if info.has_film_grain:
                        grain.film_grain(ctx, slot, info.film_grain)
                        pic = vis(grain.download_display(ctx))
                        census["grain"] += 1
                    for pl in range(3):
                        if not np.array_equal(pic[pl], fx["shown"][k_shown][pl]):
                            return "DIFF shown picture %d plane %d%s" % (k_shown, pl, " (film grain)" if info.has_film_grain else ""), census
                    k_shown += 1
        if k_all != len(fx["final"]) or k_shown != len(fx["shown"]):
            return "COUNT %d/%d frames, %d/%d shown" % (k_all, len(fx["final"]), k_shown, len(fx["shown"])), census
        return "OK", census
    finally:
        ctx.close()


if __name__ == "__main__":
    # two phases, two processes: `encode DIR [filter ...]` (CPU workers: libaom + dav1d, nothing touches the GPU) writes one file per configuration,
    # `check DIR` (one process, the GPU) reads them
    import pickle
    mode, outdir = sys.argv[1], sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    if mode == "encode":
        cfgs = configs()
        if len(sys.argv) > 3:
            cfgs = [c for c in cfgs if any(a in c[0] for a in sys.argv[3:])]
        with mp.Pool(int(os.environ.get("EA1_SWEEP_WORKERS", "12"))) as pool:
            for k, (name, fx, msg) in enumerate(pool.imap(encode_and_reference, cfgs)):
                with open(os.path.join(outdir, "%04d.pkl" % k), "wb") as fp:
                    pickle.dump((name, fx, msg), fp)
        print("encoded %d configurations" % len(cfgs))
        sys.exit(0)
    import torch
    torch.zeros(1, device="cuda")   # (torch's runtime first, as in the other tools of this directory)
    import easyav1_amd
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    tables = av1parse.Tables(blob)
    sec = {s_["kind"]: s_ for s_ in av1parse.blob_sections() if s_["kind"] == 3}
    gauss = np.ascontiguousarray(blob.view(np.uint16)[sec[3]["offset"] // 2:][:2048].view(np.int16))
    bad = ok = refused = 0
    total = dict(frames=0, grain=0)
    files = sorted(f for f in os.listdir(outdir) if f.endswith(".pkl"))
    for fn in files:
        with open(os.path.join(outdir, fn), "rb") as fp:
            name, fx, msg = pickle.load(fp)   # (written by the encode phase of this very script)
        if fx is None:
            refused += 1
            print("%-40s %s" % (name, msg), flush=True)
            continue
        try:
            res, census = gpu_check(fx, av1parse, tables, gauss)
        except Exception as e:
            res, census = "EXC %r" % e, dict(frames=0, grain=0)
        for k in total:
            total[k] += census[k]
        ok += res == "OK"
        bad += res != "OK"
        print("%-40s %s" % (name, res), flush=True)
    print("configurations %d ok %d bad %d refused by the encoder %d; frames %d, pictures with film grain %d" % (len(files), ok, bad, refused, total["frames"], total["grain"]))
    sys.exit(1 if bad else 0)
