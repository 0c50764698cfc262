"""Generated streams through the product's GPU path against dav1d (the GPU box: Pillow's libavif carries dav1d there too).

    python tools/dbg/gen_gpu_sweep.py encode DIR [streams] [first] && python tools/dbg/gen_gpu_sweep.py check DIR

encode (CPU workers): the configurations of tools/dbg/gen_vs_dav1d.py -- generator -> bytes; dav1d's pictures after all filters, every frame; streams that are
not conformant (reference sizes beyond the 2 : 1 / 1 : 16 limits: dav1d answers EINVAL; transforms that leave the value range: the oracle's monitor) are set aside.
check (one process, the GPU): bytes -> block parser -> hand-over -> a context of the stream's layout -> HIP kernels == dav1d, frame by frame."""
import sys, os, pickle, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def encode(k):
    import io, contextlib
    import easyav1_amd
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse
    from tests import streamgen
    from tests import test_av1_block_parse as T
    from tests.golden import av1_ref
    from tools.dbg import gen_vs_dav1d as g, real_av1
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    env = (av1parse, streamgen, av1parse.Tables(blob))
    size, kw = g.config(k)
    units, frames, headers, counts = T.make_stream(env, 3000 + k, size, 6, hdr_adjust=g.tame_header, legal_scaling=True, **kw)
    if g.scaling_violation(headers) is not None:
        return k, None, "not conformant (reference scaling limits)"
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ok = real_av1.run(units, verbose=False)     # parser + oracle == dav1d (and the conformance monitor)
        final = av1_ref.dav1d_pictures(units, inloop=7, apply_grain=0, all_frames=True)
    except Exception as e:
        return k, None, "dav1d refuses: %r" % e
    if ok == "range":
        return k, None, "not conformant (transform range)"
    return k, dict(units=units, final=final, oracle_ok=bool(ok), layout=frames[0].desc.layout), ""


if __name__ == "__main__":
    mode, outdir = sys.argv[1], sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    if mode == "encode":
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 336
        first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        with mp.Pool(int(os.environ.get("EA1_SWEEP_WORKERS", "12"))) as pool:
            for k, fx, msg in pool.imap(encode, range(first, first + n)):
                with open(os.path.join(outdir, "%05d.pkl" % k), "wb") as fp:
                    pickle.dump((k, fx, msg), fp)
        print("encoded", n)
        sys.exit(0)
    import torch
    torch.zeros(1, device="cuda")
    import easyav1_amd
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse, video, ir
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    tables = av1parse.Tables(blob)
    sec = {s["kind"]: s for s in av1parse.blob_sections() if s["kind"] == 5}
    upscale = np.ascontiguousarray(blob.view(np.uint16)[sec[5]["offset"] // 2:][:512].view(np.int16).reshape(64, 8))
    ok = bad = aside = frames_n = 0
    why = {}
    for fn in sorted(os.listdir(outdir)):
        with open(os.path.join(outdir, fn), "rb") as fp:
            k, fx, msg = pickle.load(fp)   # (written by the encode phase of this script)
        if fx is None:
            aside += 1
            why[msg.split(":")[0]] = why.get(msg.split(":")[0], 0) + 1
            continue
        ssx, ssy = (0 if fx["layout"] == 1 else 1), (1 if fx["layout"] == 0 else 0)
        parser = av1parse.FrameParser(tables)
        parsed = []
        for tu in fx["units"]:
            rc, f, info = parser.parse(tu)
            assert rc == 0
            parsed.append(f)
        parser.close()
        ctx = video.ReconContext(max(max(f.desc.width, (f.desc.upscaled_width + 7) & ~7) for f in parsed), max(f.desc.height for f in parsed), layout=fx["layout"])
        ctx.set_upscale_filter(upscale)
        res = "OK"
        try:
            for n, f in enumerate(parsed):
                job = ctx.make_job(f)
                ctx.submit(job)
                got = ctx.download(f.desc.dst_slot)
                job.close()
                want = fx["final"][n]
                for pl in range(3):
                    if want[pl] is None:
                        continue
                    g_ = got[pl][:want[pl].shape[0], :want[pl].shape[1]]
                    if not np.array_equal(g_, want[pl]):
                        d = np.argwhere(g_ != want[pl])
                        res = "DIFF frame %d plane %d: %d samples, first (y=%d, x=%d)" % (n, pl, len(d), d[0][0], d[0][1])
                        break
                if res != "OK":
                    break
                frames_n += 1
        except Exception as e:
            res = "EXC %r" % e
        ctx.close()
        ok += res == "OK"
        bad += res != "OK"
        if res != "OK" or not fx["oracle_ok"]:
            print("stream %d: %s (parser + oracle == dav1d: %s)" % (3000 + k, res, fx["oracle_ok"]), flush=True)
    print("streams through the GPU path %d: equal to dav1d %d, not %d; %d frames; set aside %d %s" % (ok + bad, ok, bad, frames_n, aside, why))
    sys.exit(1 if bad else 0)
