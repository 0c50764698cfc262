"""The test generator's streams against the REFERENCE DECODER (build container: Pillow's libavif carries dav1d 1.5.3).

The generator (tests/native/av1_streamgen.cpp) writes real AV1 bitstreams -- the same TileSyntax as the parser with a symbol WRITER -- and reaches what libaom
through Pillow never emits: super-resolution, frames of changing size / scaled references, switch frames, whole-sample motion vectors with global motion, chroma
restoration units of 32, tools in combinations no encoder picks.  Until round 5 those were pinned by the round trip generator -> parser and by parser + oracle ==
kernels only; dav1d decodes the same bytes.  Here: stream -> parser -> oracle, stage by stage (reconstruction / + deblocking / + CDEF / + restoration) == dav1d.

    python tools/dbg/gen_vs_dav1d.py [streams] [first seed]        V=1: the per-stage / per-block report of every failing stream"""
import sys, os, io, contextlib, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SIZES = [(200, 136), (264, 200), (130, 98), (392, 264), (166, 102), (320, 192), (72, 56), (520, 280)]


def config(k):
    from tests import test_av1_block_parse as T
    seq_kw, hdr_kw, policy, extra = {}, {}, {}, {}
    if k & 1:
        seq_kw["use_128x128_superblock"] = 1
        policy["split_bias"] = 10
    if k & 2:
        seq_kw["enable_superres"] = 1
    size = SIZES[k % len(SIZES)]
    if k & 4:
        hdr_kw.update(frame_size_override_flag=1, frame_size_range=(size[0] * 5 // 8, size[1] * 5 // 8))
    if k % 5 == 4:
        seq_kw.update(T.TOOLS_OFF["seq_kw"])
    if k % 7 == 6:
        seq_kw.update(T.NO_ORDER_HINT["seq_kw"])
    if k % 11 == 5:
        seq_kw.update(T.INTEGER_MV["seq_kw"])
    if k % 13 == 7:
        seq_kw.update(T.MONO["seq_kw"])
    # a fine quantiser: the generator's coefficient levels are random, and at a coarse one their inverse transforms leave the range a conformant stream
    # keeps its intermediate values in (AV1 spec 7.13.2.1 / 7.13.3: "it is a requirement of bitstream conformance that all values stored ... are
    # representable by a signed integer using 8 + BitDepth bits") -- where decoders are free to differ, and dav1d's clamps do differ from the specification's
    hdr_kw["base_q_idx"] = [12, 20, 32, 48][k % 4]
    if k % 6 == 3:
        extra["switch_every"] = 3
    if k % 4 == 2:
        extra["split_obus"] = True
    policy.update(br_stop=92, golomb=97)   # ... and small coefficient levels, for the same reason
    # from stream 56 on: other symbol policies (rich residuals make every prediction visible; small blocks and long vectors; nearly all inter) ...
    variant = (k // 56) % 4
    if variant == 1:
        policy.update(skip=5, all_zero=10)
    elif variant == 2:
        policy.update(split_bias=70, max_mv_class=8, skip=15, all_zero=20)
    elif variant == 3:
        policy.update(inter=95, skip=15, all_zero=15)
    # ... and from stream 224 on the other chroma layouts (4:4:4 is profile 1, 8-bit 4:2:2 profile 2)
    layout = (k // 224) % 3
    if layout and not seq_kw.get("monochrome"):
        seq_kw.update(dict(profile=1, subsampling_x=0, subsampling_y=0, layout=3) if layout == 1 else dict(profile=2, subsampling_x=1, subsampling_y=0, layout=2))
    if k % 3 == 1:   # film grain with random parameters on the shown pictures
        seq_kw["film_grain_params_present"] = 1
        hdr_kw["grain"] = True
    return size, dict(seq_kw=seq_kw, hdr_kw=hdr_kw, policy=policy or None, gop=3 if k % 3 == 0 else None, **extra)


def tame_header(h):
    """the strong quantiser matrices multiply high frequencies by up to 8: with random levels the inverse transforms overflow again (see config())"""
    for key in ("qm_y", "qm_u", "qm_v"):
        h[key] = max(int(h.get(key, 15)), 9)


def scaling_violation(headers):
    """AV1 spec 7.20 / 5.9.7 (reference frame scaling): 2 * FrameWidth >= RefUpscaledWidth, 2 * FrameHeight >= RefFrameHeight, FrameWidth <= 16 * RefUpscaledWidth,
    FrameHeight <= 16 * RefFrameHeight for every reference an inter frame names.  The generator does not keep to it; dav1d answers EINVAL.  -> frame number or None"""
    slots = [None] * 8
    for n, h in enumerate(headers):
        if h["frame_type"] in (1, 3):   # inter, switch
            for idx in h["ref_frame_idx"]:
                r = slots[idx]
                if r is None:
                    continue
                if 2 * h["frame_width"] < r[0] or 2 * h["frame_height"] < r[1] or h["frame_width"] > 16 * r[0] or h["frame_height"] > 16 * r[1]:
                    return n
        for i in range(8):
            if (h["refresh_frame_flags"] >> i) & 1:
                slots[i] = (h["upscaled_width"], h["frame_height"])
    return None


def one(k):
    import numpy as np
    import easyav1_amd
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse
    from tests import streamgen
    from tests import test_av1_block_parse as T
    from tools.dbg import real_av1
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    env = (av1parse, streamgen, av1parse.Tables(blob))
    size, kw = config(k)
    buf = io.StringIO()
    try:
        units, frames, headers, counts = T.make_stream(env, 3000 + k, size, 6, hdr_adjust=tame_header, legal_scaling=True, **kw)
        with contextlib.redirect_stdout(buf):
            ok = real_av1.run(units, verbose=False)
    except Exception as e:
        if "dav1d_send_data: -22" in repr(e):
            n = scaling_violation(headers)
            if n is not None:
                return k, size, kw, "not conformant (reference scaling limits, frame %d): dav1d answers EINVAL" % n, buf.getvalue()
        return k, size, kw, "EXC %r" % e, buf.getvalue()
    if ok == "range":
        return k, size, kw, "not conformant (transform range)", buf.getvalue()
    return k, size, kw, "OK" if ok else "DIFF", "" if ok else buf.getvalue()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    with mp.Pool(int(os.environ.get("EA1_SWEEP_WORKERS", "7"))) as pool:
        for k, size, kw, res, log in pool.imap(one, range(first, first + n)):
            bad += not (res == "OK" or res.startswith("not conformant"))
            print("stream %d %s %s: %s" % (3000 + k, size, {a: b for a, b in kw.items() if b}, res), flush=True)
            if log and os.environ.get("V"):
                print(log)
    print("streams %d bad %d" % (n, bad))
    sys.exit(1 if bad else 0)
