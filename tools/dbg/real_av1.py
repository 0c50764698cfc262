"""debug driver: libaom stream -> (repo parser + oracle) vs dav1d, stage by stage (build container only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import easyav1_amd
from tests.golden import av1_ref

STAGES = [("recon", 0), ("deblock", 1), ("cdef", 3), ("final", 7)]


def first_diff(a, b):
    d = np.argwhere(a != b)
    return None if len(d) == 0 else (int(d[0][0]), int(d[0][1]), int(a[d[0][0], d[0][1]]), int(b[d[0][0], d[0][1]]), len(d))


def run(units, verbose=True, all_frames=True, stop_on_diff=True, only=None, show_blocks=6):
    from tests import oracle_lib as ol_
    try:
        return run_(units, verbose, all_frames, stop_on_diff, only, show_blocks)
    finally:
        ol_.oracle_set_chroma_format(1, 1)


def run_(units, verbose=True, all_frames=True, stop_on_diff=True, only=None, show_blocks=6):
    easyav1_amd.load(build=False)
    from easyav1_amd import av1parse, ir
    from tests import oracle_lib as ol
    blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
    tables = av1parse.Tables(blob)
    from tests import av1_tables_gen as tg
    ol.oracle_set_upscale_filter(np.ascontiguousarray(blob.view(np.uint16)[[s for s in av1parse.blob_sections() if s["kind"] == 5][0]["offset"] // 2:][:512].view(np.int16).reshape(64, 8)))
    parser = av1parse.FrameParser(tables)
    refs = {name: av1_ref.Dav1d(inloop=bits, apply_grain=0, all_frames=all_frames) for name, bits in STAGES}
    grain_ref = av1_ref.Dav1d(inloop=7, apply_grain=1, all_frames=all_frames)
    gs = [x for x in av1parse.blob_sections() if x["kind"] == 3][0]
    ol.oracle_set_gaussian_sequence(np.ascontiguousarray(blob.view(np.uint16)[gs["offset"] // 2:][:2048].view(np.int16)))
    chains = {name: ([None] * 9, [0] * 9) for name, _ in STAGES}   # one oracle chain per filter set, references from its own output (like dav1d's)
    stage_bits = {"recon": 1, "deblock": 3, "cdef": 7, "final": 15}
    ok = True
    nfr = 0
    import ctypes
    ol.oracle().ea1o_itx_range_violations.restype = ctypes.c_ulong
    ol.oracle().ea1o_itx_range_violations(1)
    for ti, tu in enumerate(units):
        want = {name: d.decode(tu) for name, d in refs.items()}
        want_grain = grain_ref.decode(tu)
        try:
            frames = parser.parse_unit(tu)
        except Exception as e:
            print("TU %d: parse error: %s" % (ti, e))
            return False
        k = 0
        for f, info in frames:
            if f is None:
                if verbose: print("TU %d: show_existing" % ti)
                if not all_frames or True:
                    k += 1 if info.show_frame or info.show_existing else 0
                continue
            d = f.desc
            ssx, ssy = (0 if d.layout == 1 else 1), (1 if d.layout == 0 else 0)   # ea1_frame_desc.layout: 0 = 4:2:0, 1 = 4:4:4, 2 = 4:2:2
            ol.oracle_set_chroma_format(ssx, ssy)
            outs = {}
            for name, _ in STAGES:
                slots, crops = chains[name]
                if only and name not in only:
                    continue
                outs[name] = ol.oracle_decode_frame(f, slots, stage_bits[name], ref_crops=crops)["final"]
                slots[d.dst_slot] = outs[name]
                crops[d.dst_slot] = d.crop | (d.pad << 8)
                if d.upscaled_width:
                    up = d.upscaled_width
                    crops[d.dst_slot] = (d.crop & 0xF0) | (((up + 7) & ~7) - up) | ((d.pad & 0xF0) << 8)
            shown = bool(info.show_frame)
            if not shown and not all_frames:
                continue
            msg = []
            for name, _ in STAGES:
                if name not in outs:
                    continue
                if k >= len(want[name]):
                    msg.append("%s: dav1d gave no picture" % name)
                    ok = False
                    continue
                wy, wu, wv = want[name][k]
                h, w = wy.shape
                got = outs[name]
                for pl, (g, r) in enumerate(zip(got, (wy, wu, wv))):
                    if r is None: continue
                    g = g[:r.shape[0], :r.shape[1]]
                    fd = first_diff(g, r)
                    if fd:
                        msg.append("%s plane %d first diff at (y=%d,x=%d) got %d want %d (%d samples differ)" % (name, pl, *fd))
                        ok = False
            if info.has_film_grain and "final" in outs and k < len(want_grain):
                gh, gw = want_grain[k][0].shape
                mono = want_grain[k][1] is None   # 4:0:0: our chroma planes are grey and stay unseen
                vis = [outs["final"][0][:gh, :gw]] + [outs["final"][pl][:(gh + ssy) >> ssy, :(gw + ssx) >> ssx] for pl in (1, 2)]
                g = ol.oracle_film_grain(info.film_grain, vis)
                for pl in range(1 if mono else 3):
                    fd = first_diff(g[pl], want_grain[k][pl])
                    if fd:
                        msg.append("GRAIN plane %d first diff at (y=%d,x=%d) got %d want %d (%d samples differ)" % (pl, *fd))
                        ok = False
                if verbose: print("   (film grain applied: %d luma samples changed)" % int((g[0] != vis[0]).sum()))
            if msg and "recon" in outs and k < len(want["recon"]):
                # the blocks (in decoding order) whose recon differs
                wy, wu, wv = want["recon"][k]
                bad = 0
                for bi, b in enumerate(f.blocks):
                    x, y, bw, bh = int(b["x4"]) * 4, int(b["y4"]) * 4, 1 << int(b["bwl"]), 1 << int(b["bhl"])
                    dy = outs["recon"][0][y:y + bh, x:x + bw][:max(0, wy.shape[0] - y), :max(0, wy.shape[1] - x)] != wy[y:y + bh, x:x + bw]
                    cy0, cy1, cx0, cx1 = y >> ssy, (y + bh + ssy) >> ssy, x >> ssx, (x + bw + ssx) >> ssx
                    if wu is None:   # 4:0:0
                        du = np.zeros((1, 1), bool)
                    else:
                        du = outs["recon"][1][cy0:cy1, cx0:cx1][:max(0, wu.shape[0] - cy0), :max(0, wu.shape[1] - cx0)] != wu[cy0:cy1, cx0:cx1]
                    if dy.any() or du.any():
                        bad += 1
                        if bad <= show_blocks:
                            print("    block %d at (%d,%d) %dx%d luma diff %d chroma diff %d: flags %#x mode_y %d mode_uv %d ref %s mv %s comp %d/%d ii %d/%d fi %d pal %s n_tx %d angle %s cfl %s" % (
                                bi, x, y, bw, bh, int(dy.sum()), int(du.sum()), int(b["flags"]), int(b["mode_y"]), int(b["mode_uv"]), b["ref"].tolist(), b["mv"].tolist(),
                                int(b["comp_type"]), int(b["comp_arg"]), int(b["ii_mode"]), int(b["ii_arg"]), int(b["filter_intra"]), b["pal_size"].tolist(), int(b["n_tx"]), b["angle_delta"].tolist(), b["cfl_alpha"].tolist()))
                print("    %d of %d blocks differ in recon" % (bad, len(f.blocks)))
            viol = int(ol.oracle().ea1o_itx_range_violations(1))
            if msg and viol:
                # not a conformant stream (AV1 spec 7.13.3: the inverse transforms left the value range a conformant stream keeps): decoders may differ
                print("TU %d frame %d: NOT CONFORMANT: %d transform values beyond 8 + BitDepth bits; differences set aside: %s" % (ti, nfr, viol, msg[0]))
                return "range"
            if verbose or msg:
                print("TU %d frame %d (%dx%d type %d show %d, %d blocks, %d tx): %s" % (ti, nfr, d.width, d.height, info.header.frame_type, info.show_frame, d.num_blocks, d.num_tx, "OK" if not msg else ""))
                for m in msg: print("    " + m)
            k += 1
            nfr += 1
            if msg and stop_on_diff:
                return False
    return ok


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=64); ap.add_argument("--h", type=int, default=64); ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1); ap.add_argument("--q", type=int, default=60); ap.add_argument("--speed", type=int, default=6)
    ap.add_argument("--kind", default="natural"); ap.add_argument("--tc", type=int, default=0); ap.add_argument("--tr", type=int, default=0)
    ap.add_argument("--sub", default="4:2:0")
    ap.add_argument("-o", action="append", default=[], help="libaom option key=value")
    a = ap.parse_args()
    adv = dict(x.split("=", 1) for x in a.o)
    frames = av1_ref.synthetic_frames(a.seed, a.w, a.h, a.n, a.kind)
    units = av1_ref.encode_sequence(frames, quality=a.q, speed=a.speed, subsampling=a.sub, tile_cols_log2=a.tc, tile_rows_log2=a.tr, advanced=adv)
    print("units:", [len(u) for u in units])
    print("RESULT", run(units))
