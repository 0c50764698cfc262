"""Build-container-only: assemble the AV1 specification's constant tables (default CDFs, coefficient scans, Dc/Ac_Qlookup,
Quantizer_Matrix, Gaussian_Sequence, Upscale_Filter) into the table blob of include/easyav1_amd_obu.h.

The numbers are DATA of the AV1 specification that neither /root/reference (dav1d is an empty submodule there) nor this
repository's authors can write down from memory to the last digit.  Two independent witnesses of them sit in this container:
Pillow's bundled libavif (pillow.libs/libavif-*.so) links libaom 3.13.2 (encoder) and dav1d 1.5.3 -- the decoder the reference pins
(/root/reference/cmake/dav1d.cmake:18-21) -- and both carry the tables as constant data.  This script

  1. locates each table in the read-only data of that file BY SHAPE AND BY RECALLED ANCHOR VALUES (no symbol names, no offsets
     hard-coded): a libaom CDF array is rows of `32768 - cdf` values, strictly decreasing, then zeros up to a fixed stride;
  2. takes the value from the libaom copy where the compiler kept the array, and the recalled values below (checked for presence in
     dav1d's copy) for the few small arrays libaom's build folded into immediates;
  3. cross-checks every multi-symbol CDF row against a SECOND copy elsewhere in the file (dav1d's);
  4. writes easyav1_amd/data/av1_spec_tables.bin (product data: the library's default tables).

The acceptance test is not provenance but pictures: tests/test_real_av1.py decodes libaom-made streams with these tables and
compares every picture with dav1d 1.5.3's, bit for bit (tests/golden/make_video_fixtures.py).

Orientation notes (found while doing this, confirmed by decoding): libaom >= 3.6 and dav1d keep coefficients TRANSPOSED relative to
the specification, so their square default scans and their rectangular quantiser matrices are the transposes of the specification's;
rectangular default scans are a transposition-symmetric family (the scan of w x h transposed is the scan of h x w).

Run:  python tests/golden/make_av1_tables.py            (needs Pillow with AVIF support; nothing is installed or downloaded)
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "easyav1_amd", "data", "av1_spec_tables.bin")


def libavif_path():
    import PIL
    hits = glob.glob(os.path.join(os.path.dirname(PIL.__file__), "..", "pillow.libs", "libavif-*.so*"))
    if not hits:
        raise SystemExit("Pillow's bundled libavif not found")
    return hits[0]


class RoData:
    """the .rodata section of the library as bytes / uint16 / int16"""

    def __init__(self, path):
        raw = open(path, "rb").read()
        # ELF64 section headers: find .rodata
        shoff = int.from_bytes(raw[0x28:0x30], "little")
        shentsize = int.from_bytes(raw[0x3A:0x3C], "little")
        shnum = int.from_bytes(raw[0x3C:0x3E], "little")
        shstrndx = int.from_bytes(raw[0x3E:0x40], "little")
        secs = []
        for i in range(shnum):
            h = raw[shoff + i * shentsize: shoff + (i + 1) * shentsize]
            secs.append((int.from_bytes(h[0:4], "little"), int.from_bytes(h[0x18:0x20], "little"), int.from_bytes(h[0x20:0x28], "little")))
        stro = secs[shstrndx][1]
        self.bytes = None
        for name, off, size in secs:
            end = raw.index(b"\0", stro + name)
            if raw[stro + name:end] == b".rodata":
                self.bytes = raw[off:off + size]
        if self.bytes is None:
            raise SystemExit("no .rodata")
        n = len(self.bytes) // 2 * 2
        self.u16 = np.frombuffer(self.bytes[:n], np.uint16).astype(np.int64)
        self.s16 = np.frombuffer(self.bytes[:n], np.int16).astype(np.int64)
        self.u8 = np.frombuffer(self.bytes, np.uint8)

    def find_bytes(self, pattern):
        out, i = [], 0
        while True:
            i = self.bytes.find(pattern, i)
            if i < 0:
                return out
            out.append(i)
            i += 1


def row_ok(u, i, stride, n):
    r = u[i:i + stride]
    if len(r) < stride:
        return False
    if n == 0:
        return not r.any()
    k = n - 1
    if k:
        if r[0] >= 32768 or r[k - 1] <= 0:
            return False
        if k > 1 and not np.all(np.diff(r[:k]) < 0):
            return False
    return not r[k:].any()


def find_cdf_array(ro, stride, ns, anchor=None, align=16, probe=None):
    """u16 indices where an inverse-CDF array with `stride` entries per row and ns[r] symbols in row r (0 = all-zero row) starts;
    anchor = leading cumulative values of the first non-zero row; probe = (row, value): first cumulative value of another row"""
    u = ro.u16
    first = next(j for j, n in enumerate(ns) if n)
    want = None if anchor is None else [32768 - a for a in anchor]
    hits = []
    if want is not None:
        # jump straight to candidates through a byte search of the anchor
        pat = np.array(want, np.uint16).tobytes()
        cands = [p // 2 - first * stride for p in ro.find_bytes(pat) if p % 2 == 0]
    else:
        cands = range(0, len(u) - stride * len(ns), align)
    for i in cands:
        if i < 0 or i % align:
            continue
        if all(row_ok(u, i + j * stride, stride, n) for j, n in enumerate(ns)):
            if probe is not None and 32768 - u[i + probe[0] * stride] != probe[1]:
                continue
            hits.append(i)
    return hits


def take(ro, name, stride, ns, anchor=None, align=16, probe=None):
    h = find_cdf_array(ro, stride, ns, anchor, align, probe)
    if not h:
        raise SystemExit("%s: no candidate" % name)
    got = []
    for i in h:
        rows = []
        for j, n in enumerate(ns):
            r = ro.u16[i + j * stride: i + j * stride + stride]
            rows.append([int(32768 - v) for v in r[:n - 1]] + [32768] if n else None)
        got.append(rows)
    if any(g != got[0] for g in got[1:]):   # several places of the same shape are fine when they hold the same numbers (dav1d's copy)
        raise SystemExit("%s: %d different candidates %s" % (name, len(h), [hex(2 * x) for x in h[:8]]))
    return got[0]


# ---- recalled small arrays (libaom's build folded them into immediates); every one is checked for presence in dav1d's data ----
RECALLED = {
    "FilterIntraMode": [[8949, 12776, 17211, 29558]],
    "CflSign": [[1418, 2123, 13340, 18405, 26972, 28343, 32294]],
    "Skip": [[31671], [16515], [4576]],
    "SkipMode": [[32621], [20708], [8127]],
    "IsInter": [[806], [16662], [20186], [26538]],
    "SegmentId": [[5622, 7893, 16093, 18233, 27809, 28373, 32533], [14274, 18230, 22557, 24935, 29980, 30851, 32344],
                  [27527, 28487, 28723, 28890, 32397, 32647, 32679]],
    "SegmentIdPredicted": [[16384]] * 3,
    "DeltaQ": [[28160, 32120, 32677]],
    "DeltaLF": [[28160, 32120, 32677]],
    "DeltaLFMulti": [[28160, 32120, 32677]] * 4,
    "CompMode": [[26828], [24035], [12031], [10640], [2901]],
    "CompRefType": [[1198], [2070], [9166], [7499], [22475]],
    "NewMv": [[24035], [16630], [15339], [8386], [12222], [4676]],
    "ZeroMv": [[2175], [1054]],
    "RefMv": [[23974], [24188], [17848], [28622], [24312], [19923]],
    "DrlMode": [[13104], [24560], [18945]],
    "InterIntra": [[26887], [27597], [30237]],
    "CompGroupIdx": [[26607], [22891], [18840], [24594], [19934], [22674]],
    "CompoundIdx": [[18244], [12865], [7053], [13259], [9334], [4644]],
    "Intrabc": [[30531]],
    "PaletteUVMode": [[32461], [21488]],
    "UseWiener": [[11570]],
    "UseSgrproj": [[16855]],
    "RestorationType": [[9413, 22581]],
    # motion vector contexts (identical for both components and both MvCtx values)
    "MvJoint": [[4096, 11264, 19328]] * 2,
    "MvSign": [[16384]] * 4,
    "MvClass": [[28672, 30976, 31858, 32320, 32551, 32656, 32740, 32757, 32762, 32767]] * 4,
    "MvClass0Bit": [[27648]] * 4,
    "MvClass0Fr": [[16384, 24576, 26624], [12288, 21248, 24128]] * 4,
    "MvClass0Hp": [[20480]] * 4,
    "MvBit": [[17408], [17920], [18944], [20480], [22528], [24576], [28672], [29952], [29952], [30720]] * 4,
    "MvFr": [[8192, 17408, 21248]] * 4,
    "MvHp": [[16384]] * 4,
}


def second_copy(ro, row, skip_at=None):
    """is the inverse-CDF prefix of `row` (cumulative values without the final 32768) present somewhere else in the file?"""
    pat = np.array([32768 - v for v in row], np.uint16).tobytes()
    hits = [p for p in ro.find_bytes(pat) if p % 2 == 0]
    return len(hits) >= (2 if skip_at is None else 1)


def build_cdfs(ro):
    out = {}
    P = lambda rows: [r[:-1] for r in rows]   # drop the trailing 32768: sections are filled below with it appended again
    out["IntraFrameYMode"] = P(take(ro, "kf_y_mode", 14, [13] * 25, [15588, 17027]))
    out["YMode"] = P(take(ro, "if_y_mode", 14, [13] * 4, [22801, 23489]))
    uv = take(ro, "uv_mode", 15, [13] * 13 + [14] * 13, [22631, 24152])
    out["UVModeCflNotAllowed"], out["UVModeCflAllowed"] = P(uv[:13]), P(uv[13:])
    out["AngleDelta"] = P(take(ro, "angle_delta", 8, [7] * 8, [2180, 5032]))
    out["FilterIntra"] = P(take(ro, "filter_intra", 3, [2] * 22, [4621]))
    out["CflAlpha"] = P(take(ro, "cfl_alpha", 17, [16] * 6, [7637, 20719]))
    part = take(ro, "partition", 11, [4] * 4 + [10] * 12 + [8] * 4, [19132, 25510])
    out["Partition8"], out["Partition16"], out["Partition32"], out["Partition64"], out["Partition128"] = (
        P(part[0:4]), P(part[4:8]), P(part[8:12]), P(part[12:16]), P(part[16:20]))
    tx = take(ro, "tx_size", 4, [2] * 3 + [3] * 9, [19968])
    out["Tx8x8"], out["Tx16x16"], out["Tx32x32"], out["Tx64x64"] = P(tx[0:3]), P(tx[3:6]), P(tx[6:9]), P(tx[9:12])
    out["TxfmSplit"] = P(take(ro, "txfm_partition", 3, [2] * 21, [28581]))
    # [set][size: 4x4, 8x8, 16x16, 32x32][intra mode]; set 0 is never coded (zeros); the 7-type set is coded for 4x4 and 8x8, the 5-type
    # set up to 16x16 (sizes it is not coded for hold uniform distributions)
    itx = take(ro, "intra_ext_tx", 17, [0] * 52 + [7] * 52 + [5] * 52, None, probe=(52, 1535))
    out["IntraTxTypeSet1"] = P(itx[52:52 + 26])
    out["IntraTxTypeSet2"] = P(itx[104:104 + 39])
    etx = take(ro, "inter_ext_tx", 17, [0] * 4 + [16] * 4 + [12] * 4 + [2] * 4, None)
    out["InterTxTypeSet1"] = P(etx[4:6])
    out["InterTxTypeSet2"] = P(etx[10:11])
    out["InterTxTypeSet3"] = P(etx[12:16])
    out["UniCompRef"] = P(take(ro, "uni_comp_ref", 3, [2] * 9, [5284]))
    out["CompRef"] = P(take(ro, "comp_ref", 3, [2] * 9, [4946]))
    out["CompBwdRef"] = P(take(ro, "comp_bwdref", 3, [2] * 6, [2235]))
    out["SingleRef"] = P(take(ro, "single_ref", 3, [2] * 18, [4897]))
    out["CompoundMode"] = P(take(ro, "inter_compound_mode", 9, [8] * 8, [7760, 13823]))
    out["InterpFilter"] = P(take(ro, "switchable_interp", 4, [3] * 16, [31935, 32720]))
    out["MotionMode"] = P(take(ro, "motion_mode", 4, [3] * 22, [10923, 21845], probe=(3, 7651)))
    # three more [22] binary arrays share a shape: told apart by the BLOCK_8X8 entry
    out["UseObmc"] = P(take(ro, "obmc", 3, [2] * 22, [16384], probe=(3, 10437)))
    out["WedgeInterIntra"] = P(take(ro, "wedge_interintra", 3, [2] * 22, [16384], probe=(3, 20036)))
    out["CompoundType"] = P(take(ro, "compound_type", 3, [2] * 22, [16384], probe=(3, 23431)))
    out["WedgeIndex"] = P(take(ro, "wedge_idx", 17, [16] * 22, [2048, 4096]))
    iim = take(ro, "interintra_mode", 5, [4] * 4, [8192, 16384, 24576], probe=(1, 1875))
    out["InterIntraMode"] = P(iim[1:4])
    out["PaletteYMode"] = P(take(ro, "palette_y_mode", 3, [2] * 21, [31676]))
    out["PaletteYSize"] = P(take(ro, "palette_y_size", 8, [7] * 7, [7952, 13000]))
    out["PaletteUVSize"] = P(take(ro, "palette_uv_size", 8, [7] * 7, [8713, 19979]))
    sizes = sum([[n] * 5 for n in range(2, 9)], [])
    py = take(ro, "palette_y_color", 9, sizes, [28710])
    puv = take(ro, "palette_uv_color", 9, sizes, [29089])
    for n in range(2, 9):
        out["PaletteSize%dYColor" % n] = P(py[(n - 2) * 5:(n - 1) * 5])
        out["PaletteSize%dUVColor" % n] = P(puv[(n - 2) * 5:(n - 1) * 5])
    out.update({k: [list(r) for r in v] for k, v in RECALLED.items()})

    coef = [dict() for _ in range(4)]
    tsk = take(ro, "txb_skip", 3, [2] * 260, [31849])
    e16 = take(ro, "eob_multi16", 6, [5] * 16, [840, 1039])
    e32 = take(ro, "eob_multi32", 7, [6] * 16, [400, 520])
    e64 = take(ro, "eob_multi64", 8, [7] * 16, [329, 498])
    e128 = take(ro, "eob_multi128", 9, [8] * 16, [219, 482])
    e256 = take(ro, "eob_multi256", 10, [9] * 16, [310, 584])
    e512 = take(ro, "eob_multi512", 11, [10] * 16, [641, 983])
    e1024 = take(ro, "eob_multi1024", 12, [11] * 16, [393, 421])
    eex = take(ro, "eob_extra", 3, [2] * 360, [16961])
    dcs = take(ro, "dc_sign", 3, [2] * 24, [16000])
    cbe = take(ro, "coeff_base_eob", 4, [3] * 160, [17837, 29055])
    cb = take(ro, "coeff_base", 5, [4] * 1680, [4034, 8930, 12727])
    cbr = take(ro, "coeff_lps", 5, [4] * 840, [14298, 20718, 24174])
    for q in range(4):
        c = coef[q]
        c["TxbSkip"] = P(tsk[q * 65:(q + 1) * 65])
        for nm, arr in (("EobPt16", e16), ("EobPt32", e32), ("EobPt64", e64), ("EobPt128", e128), ("EobPt256", e256)):
            c[nm] = P(arr[q * 4:(q + 1) * 4])                       # [plane][2-D class or not]
        c["EobPt512"] = P([e512[q * 4 + 0], e512[q * 4 + 2]])        # [plane], the 2-D class only
        c["EobPt1024"] = P([e1024[q * 4 + 0], e1024[q * 4 + 2]])
        c["EobExtra"] = P(eex[q * 90:(q + 1) * 90])
        c["DcSign"] = P(dcs[q * 6:(q + 1) * 6])
        c["CoeffBaseEob"] = P(cbe[q * 40:(q + 1) * 40])
        c["CoeffBase"] = P(cb[q * 420:(q + 1) * 420])
        c["CoeffBr"] = P(cbr[q * 210:q * 210 + 168])                # transform size classes 0..3 (the fifth repeats the fourth's role)
    return out, coef


def is_diag_scan(a, w, h):
    n = w * h
    if a[0] != 0 or a.min() != 0 or a.max() != n - 1 or len(set(a.tolist())) != n:
        return False
    d = a // w + a % w
    return bool(np.all(np.diff(d) >= 0))


def build_scans(ro):
    """default scans per (w, h): located by shape (a permutation that walks the anti-diagonals in order); squares are transposed back to
    the specification's orientation"""
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8)]
    out = {}
    s = ro.s16
    for w, h in shapes:
        n = w * h
        found = []
        for i in range(0, len(s) - n, 8):
            if s[i] == 0 and s[i + 1] in (1, w) and is_diag_scan(s[i:i + n], w, h):
                a = s[i:i + n].copy()
                if not any(np.array_equal(a, f) for f in found):
                    found.append(a)
        if w == h:
            # both orientations of a square zig-zag may be present; the AV1 copies (libaom >= 3.6, dav1d) start downwards ([0, w, 1, ...]): the
            # specification's is their transpose, starting to the right
            down = [f for f in found if f[1] == w]
            if len(down) != 1:
                raise SystemExit("scan %dx%d: %d candidates" % (w, h, len(down)))
            a = down[0]
            a = (a % w) * w + a // w
        else:
            if len(found) != 1:
                raise SystemExit("scan %dx%d: %d candidates" % (w, h, len(found)))
            a = found[0]
            assert a[1] == (1 if w < h else w)
        out[(w, h)] = a
    return out


def build_quant(ro):
    dc = ro.find_bytes(np.array([4, 8, 8, 9, 10, 11, 12, 12, 13, 14, 15, 16, 17, 18, 19, 19], np.int16).tobytes())
    ac = ro.find_bytes(np.array([4, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22], np.int16).tobytes())
    if len(dc) != 1 or len(ac) != 1:
        raise SystemExit("quantiser look-ups: %d / %d candidates" % (len(dc), len(ac)))
    dcq = ro.s16[dc[0] // 2: dc[0] // 2 + 256]
    acq = ro.s16[ac[0] // 2: ac[0] // 2 + 256]
    assert np.all(np.diff(dcq) >= 0) and np.all(np.diff(acq) >= 0) and dcq[255] == 1336 and acq[255] == 1828, (dcq[255], acq[255])
    # second witness: dav1d keeps {dc, ac} pairs per index
    pair = np.stack([dcq, acq], 1).astype(np.uint16).tobytes()
    if not ro.find_bytes(pair):
        raise SystemExit("quantiser look-ups: no interleaved second copy")
    return dcq, acq


QM_SHAPES = [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8)]


def build_qm(ro):
    """Quantizer_Matrix[15][2][3344] from libaom's copy, rectangular sizes swapped back (see the module comment)"""
    h = ro.find_bytes(bytes([32, 43, 73, 97, 43, 67, 94, 110, 73, 94, 137, 150, 97, 110, 150, 200]))
    if len(h) != 1:
        raise SystemExit("quantiser matrices: %d candidates" % len(h))
    raw = ro.u8[h[0]: h[0] + 15 * 2 * 3344].astype(np.int64).reshape(15, 2, 3344)
    assert raw.min() >= 1
    out = np.zeros_like(raw)
    offs, o = {}, 0
    for w, hh in QM_SHAPES:
        offs[(w, hh)] = o
        o += w * hh
    for lvl in range(15):
        for pl in range(2):
            for w, hh in QM_SHAPES:
                src = raw[lvl, pl, offs[(hh, w)]: offs[(hh, w)] + w * hh]   # libaom's slot of the transposed size holds this size, row-major
                m = src.reshape(hh, w)
                if w == hh:
                    assert np.array_equal(m, m.T), "square quantiser matrix not symmetric"
                out[lvl, pl, offs[(w, hh)]: offs[(w, hh)] + w * hh] = src
            # orientation check on the steepest level: a 4-wide, 8-high matrix changes faster along its rows
        m = out[lvl, 0, offs[(4, 8)]: offs[(4, 8)] + 32].reshape(8, 4)
        if lvl < 8:
            assert m[0, 3] - m[0, 0] > m[3, 0] - m[0, 0], "4x8 quantiser matrix orientation"
    return out


def build_gaussian(ro):
    head = [56, 568, -180, 172, 124, -84, 172, -64, -900, 24, 820, 224, 1248, 996, 272, -8, -916, -388, -732, -104, -188, 800, 112, -652,
            -320, -376, 140, -252, 492, -168, 44, -788, 588, -584, 500, -228, 12, 680, 272, -476, 972, -100, 652, 368, 432, -196, -720, -192]
    h = ro.find_bytes(np.array(head, np.int16).tobytes())
    if not h:
        raise SystemExit("Gaussian_Sequence not found")
    g = ro.s16[h[0] // 2: h[0] // 2 + 2048]
    for x in h[1:]:
        assert np.array_equal(g, ro.s16[x // 2: x // 2 + 2048])
    assert g.min() >= -2048 and g.max() <= 2047 and np.all(g % 4 == 0) and abs(int(g.sum())) < 40000
    return g


def build_upscale(ro):
    head = [0, 0, 0, 128, 0, 0, 0, 0, 0, 0, -1, 128, 2, -1, 0, 0, 0, 1, -3, 127, 4, -2, 1, 0]
    h = ro.find_bytes(np.array(head, np.int16).tobytes())
    if len(h) != 1:
        raise SystemExit("Upscale_Filter: %d candidates" % len(h))
    f = ro.s16[h[0] // 2: h[0] // 2 + 512].reshape(64, 8)
    assert np.all(f.sum(1) == 128)
    # dav1d's copy: int8, negated
    if not ro.find_bytes((-f).astype(np.int8).tobytes()):
        raise SystemExit("Upscale_Filter: no second copy")
    return f


def main():
    from easyav1_amd import av1parse
    ro = RoData(libavif_path())
    mode, coef = build_cdfs(ro)
    scans = build_scans(ro)
    dcq, acq = build_quant(ro)
    qm = build_qm(ro)
    gauss = build_gaussian(ro)
    upscale = build_upscale(ro)

    # second-witness check of every CDF row with at least two free values; binary rows: the value must occur in the file's other copy
    unchecked = 0
    checked = 0
    for name, rows in list(mode.items()) + [("%s_q%d" % (k, q), v) for q in range(4) for k, v in coef[q].items()]:
        for r in rows:
            if len(r) >= 2 and len(set(np.diff([0] + r + [32768]).tolist())) > 2:   # uniform rows of never-coded contexts are not in dav1d
                if not second_copy(ro, r, skip_at=True if name in RECALLED else None):
                    raise SystemExit("%s row %s has no second copy in the file" % (name, r))
                checked += 1
            else:
                unchecked += 1
    for name, rows in RECALLED.items():
        for r in rows:
            pat = np.array([32768 - v for v in r], np.uint16).tobytes()
            if not [p for p in ro.find_bytes(pat) if p % 2 == 0]:
                raise SystemExit("recalled %s %s not present in the file" % (name, r))
    print("CDF rows with a second witness: %d; binary / uniform rows (value presence only): %d" % (checked, unchecked))

    blob = av1parse.new_blob()
    u16 = blob.view(np.uint16)
    qm_off = {}
    o = 0
    for w, h in QM_SHAPES:
        qm_off[(w, h)] = o
        o += w * h
    for s in av1parse.blob_sections():
        at = s["offset"] // 2
        name = s["name"]
        if s["kind"] == 0:
            if "_q" in name and name.rsplit("_q", 1)[1].isdigit():
                base, q = name.rsplit("_q", 1)
                rows = coef[int(q)][base]
            else:
                rows = mode[name]
            n = s["symbols"]
            assert len(rows) * n == s["count"], (name, len(rows), n, s["count"])
            flat = []
            for r in rows:
                assert len(r) == n - 1, (name, r)
                flat += list(r) + [32768]
            u16[at:at + s["count"]] = np.array(flat, np.uint16)
        elif s["kind"] == 1:
            w, h = s["width"], s["height"]
            if name.startswith("Default"):
                a = scans[(w, h)]
            elif name.startswith("Mrow"):
                a = np.arange(w * h)
            else:
                a = np.array([r * w + c for c in range(w) for r in range(h)])
            u16[at:at + s["count"]] = a.astype(np.uint16)
        elif s["kind"] == 2:
            u16[at:at + 256] = (dcq if name.startswith("Dc") else acq).astype(np.uint16)
        elif s["kind"] == 4:
            level = int(name.split("_L")[1].split("_")[0])
            pl = 1 if "_chroma_" in name else 0
            w, h = s["width"], s["height"]
            u16[at:at + s["count"]] = qm[level, pl, qm_off[(w, h)]: qm_off[(w, h)] + w * h].astype(np.uint16)
        elif s["kind"] == 3:
            u16[at:at + 2048] = gauss.astype(np.int16).view(np.uint16)
        elif s["kind"] == 5:
            u16[at:at + 512] = upscale.reshape(-1).astype(np.int16).view(np.uint16)
        else:
            raise SystemExit("unknown section kind %d" % s["kind"])
    t = av1parse.Tables(blob)   # the library's own checks
    t.close()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    blob.tofile(OUT)
    import hashlib
    print("wrote %s: %d bytes, sha256 %s" % (OUT, len(blob), hashlib.sha256(blob.tobytes()).hexdigest()[:16]))


if __name__ == "__main__":
    main()
