"""Known-answer tests of the RECONSTRUCTION stage (inter prediction, inverse transforms, intra edges) whose expected values
are hand-derived from the AV1 specification's formulas for simple inputs -- arithmetic written out here, not produced by
the oracle.  Same purpose and harness as tests/test_video_kats.py (in-loop filters): they check the oracle on the CPU and
the HIP kernels with -m gpu, sharing code with neither.

What is derived by hand:
* inter prediction: whole-sample vectors copy (with the reference clamped at the picture edge, 7.11.3.3); every sub-pel
  filter is 8 taps summing to 128 and the half-sample ones are symmetric, so a constant stays a constant and a ramp
  a + 2x read half a sample to the right is a + 2x + 1; compound predictions keep 4 more bits: two constants A, B average
  to (A + B + 1) >> 1 and blend with distance weights w : 16 - w to (A w + B (16 - w) + 8) >> 4 (7.11.3.14)
* inverse DCT of a lone DC coefficient c (7.13.2.3, 7.13.3): every 1-D pass turns it into Round2(c * 2896, 12) on all
  outputs (cos(pi / 4) = 2896 / 4096 is the only factor a DC input meets); blocks with a 2 : 1 side ratio scale the input by
  the same factor first; between the passes Round2(., Transform_Row_Shift[size]), at the end Round2(., 4)
* intra edges (7.11.2): a block without neighbours predicts DC 128, V from a row of 127, H from a column of 129, Paeth from
  those and a corner of 128; with only a left (top) neighbour the missing row (column) is filled with the neighbour's
  sample next to the block's corner, not with 127 (129)."""
import numpy as np
import pytest

from easyav1_amd import ir
from tests import oracle_lib as ol
from tests.test_video_kats import BACKENDS

W, H = 128, 64


def round2(x, n):
    return (x + (1 << (n - 1))) >> n if n else x


# Transform_Row_Shift in the order of ir.TX_SIZES: 4x4 8x8 16x16 32x32 64x64 | 4x8 8x4 8x16 16x8 16x32 32x16 32x64 64x32 |
# 4x16 16x4 8x32 32x8 16x64 64x16
ROW_SHIFT = [0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2]


def dc_residual(w, h, c):
    """what a DCT_DCT block of w x h with the single coefficient c at (0, 0) adds to every sample"""
    k = ir.TX_SIZES.index((w, h))
    if max(w, h) == 2 * min(w, h):
        c = round2(c * 2896, 12)
    c = round2(c * 2896, 12)        # row pass
    c = round2(c, ROW_SHIFT[k])
    c = round2(c * 2896, 12)        # column pass
    return round2(c, 4)


def frame_of(blocks_spec, is_inter, nz_hint=0):
    """blocks_spec: list of dicts(x, y, wl, hl, flags, mode_y, mode_uv, ref, mv, comp_type, comp_arg, txs) with
    txs = list of (plane, x, y, wl, hl, txtp, coefs or None) in plane coordinates, coefs a dense (min(h,32), min(w,32)) array"""
    blocks, txs, coefs, aux = [], [], [], []
    for s in blocks_spec:
        blk = np.zeros(1, ir.BLOCK_DTYPE)[0]
        blk["x4"], blk["y4"], blk["bwl"], blk["bhl"] = s["x"] // 4, s["y"] // 4, s["wl"], s["hl"]
        blk["flags"], blk["mode_y"], blk["mode_uv"] = s["flags"], s.get("mode_y", 0), s.get("mode_uv", 0)
        blk["ref"] = s.get("ref", (0, 0))
        blk["mv"] = s.get("mv", ((0, 0), (0, 0)))
        blk["comp_type"], blk["comp_arg"] = s.get("comp_type", 0), s.get("comp_arg", 0)
        blk["cfl_alpha"] = s.get("cfl_alpha", (0, 0))
        blk["angle_delta"] = s.get("angle_delta", (0, 0))
        if "palette" in s:   # (luma colours, chroma U colours, chroma V colours, luma index map, chroma index map)
            py, pu, pv, my, mc = s["palette"]
            blk["pal_size"] = (len(py), len(pu))
            blk["aux_off"] = sum(len(a) for a in aux)
            aux.append(np.concatenate([np.asarray(py, np.uint8), np.asarray(pu, np.uint8), np.asarray(pv, np.uint8),
                                       np.asarray(my, np.uint8).ravel(), np.asarray(mc, np.uint8).ravel()]))
        blk["first_tx"] = len(txs)
        for (pl, x, y, wl, hl, tp, cf) in s["txs"]:
            off = ir.NO_COEFS
            if cf is not None:
                off = sum(len(c) for c in coefs)
                coefs.append(np.asarray(cf, np.int16).ravel())
            txs.append((x // 4, y // 4, pl, wl | (hl << 4), tp, nz_hint if cf is not None else 0, off))
        blk["n_tx"] = len(txs) - blk["first_tx"]
        blocks.append(blk)
    d = ir.FrameDesc()
    d.width, d.height, d.is_inter, d.dst_slot = W, H, 1 if is_inter else 0, 1
    d.cdef_damping = 3
    lr = [np.zeros(0, ir.LR_UNIT_DTYPE)] * 3
    allc = np.concatenate(coefs) if coefs else np.zeros(0, np.int16)
    sbs = ((W + 63) // 64) * ((H + 63) // 64)
    allaux = np.concatenate(aux) if aux else np.zeros(0, np.uint8)
    return ir.ParsedFrame(d, np.array(blocks, ir.BLOCK_DTYPE), np.array(txs, ir.TXBLOCK_DTYPE), allc, allaux, np.zeros(sbs, np.uint8), lr)


def decode(backend, f, refs):
    if backend == "oracle":
        return ol.oracle_decode_frame(f, list(refs) + [None] * (8 - len(refs)), ir.STAGE_RECON)["recon"]
    from easyav1_amd import video
    ctx = video.ReconContext(W, H)
    for k, r in enumerate(refs):
        if r is not None:
            ctx.set_ref(k, r)
    job = ctx.make_job(f)
    ctx.submit(job, ir.STAGE_RECON)
    out = ctx.download(1)
    job.close()
    ctx.close()
    return out


def uniform_grid(b_log2, make_block):
    """W x H tiled by (1 << b_log2)^2 blocks in raster order; make_block(x, y, flags) -> spec without position / size"""
    b = 1 << b_log2
    spec = []
    for y in range(0, H, b):
        for x in range(0, W, b):
            fl = (ir.BLK_HAVE_LEFT if x else 0) | (ir.BLK_HAVE_TOP if y else 0)
            s = dict(x=x, y=y, wl=b_log2, hl=b_log2)
            s.update(make_block(x, y, fl))
            spec.append(s)
    return spec


def whole_block_txs(x, y, b_log2, luma=None, chroma=None, tp=0):
    """one transform block per plane of the block's own size (64: still 64, chroma 32)"""
    c = b_log2 - 1
    return [(0, x, y, b_log2, b_log2, tp, luma), (1, x // 2, y // 2, c, c, tp, chroma), (2, x // 2, y // 2, c, c, tp, chroma)]


# ------------------------------------------------------------------------------------------------- inter prediction
def ramp_picture():
    yy, xx = np.mgrid[0:H, 0:W]
    y = (2 * xx + (yy & 1)).astype(np.uint8)               # 2x (+1 on odd rows: rows differ, columns are a ramp)
    cy, cx = np.mgrid[0:H // 2, 0:W // 2]
    c = (3 * cx + 10).astype(np.uint8)                      # <= 3 * 63 + 10 = 199
    return [y, c.copy(), (c + 20).astype(np.uint8)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("b_log2", [3, 4, 5, 6])
@pytest.mark.parametrize("dy,dx", [(0, 0), (2, -6), (-4, 10), (70, -200)])
def test_whole_sample_vectors_copy_the_reference_clamped_at_its_edges(backend, b_log2, dy, dx):
    """even whole-sample luma vectors (chroma then lands on whole samples too): every sample is the reference sample at the
    displaced position, coordinates clamped into the picture (7.11.3.3: Clip3(0, lastX, .))"""
    rng = np.random.default_rng(5)
    ref = [rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8)]
    spec = uniform_grid(b_log2, lambda x, y, fl: dict(flags=ir.BLK_INTER | fl, ref=(0, 0), mv=((8 * dy, 8 * dx), (0, 0)), txs=whole_block_txs(x, y, b_log2)))
    got = decode(backend, frame_of(spec, True), [ref])
    for pl in range(3):
        ss = 1 if pl else 0
        h, w = ref[pl].shape
        yy, xx = np.mgrid[0:h, 0:w]
        want = ref[pl][np.clip(yy + (dy >> ss), 0, h - 1), np.clip(xx + (dx >> ss), 0, w - 1)]
        assert np.array_equal(got[pl], want), (pl, np.argwhere(got[pl] != want)[:4].tolist())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("filt", [ir.FILTER_REGULAR, ir.FILTER_SMOOTH, ir.FILTER_SHARP, ir.FILTER_BILINEAR])
def test_half_sample_positions_on_a_ramp(backend, filt):
    """luma vector (0, +1/2): a + 2x becomes a + 2x + 1 wherever the 8 taps stay inside the picture; luma vector (0, +1):
    chroma sits half a sample to the right, 3x + 10 becomes Round2(3x + 11.5) -- the two middle samples average with a
    symmetric filter: (3x + 10 + 3x + 13 + 1) >> 1 = 3x + 12"""
    ref = ramp_picture()
    for mvx, plane_checks in ((4, [(0, lambda v: v + 1)]), (8, [(1, None), (2, None)])):
        spec = uniform_grid(4, lambda x, y, fl: dict(flags=ir.BLK_INTER | fl, mode_y=filt | (filt << 2), ref=(0, 0), mv=((0, mvx), (0, 0)), txs=whole_block_txs(x, y, 4)))
        got = decode(backend, frame_of(spec, True), [ref])
        for pl, fn in plane_checks:
            r = ref[pl].astype(np.int32)
            inner = slice(4, r.shape[1] - 5)
            if pl == 0:
                want = r + 1
            else:
                want = (r[:, :-1] + r[:, 1:] + 1) >> 1          # between x and x + 1
                want = np.concatenate([want, want[:, -1:]], axis=1)
            assert np.array_equal(got[pl][:, inner], want[:, inner]), (filt, mvx, pl, got[pl][2, 2:12].tolist(), want[2, 2:12].tolist())
        if mvx == 8:   # and luma moved by exactly one sample
            assert np.array_equal(got[0][:, :-1], ref[0][:, 1:])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("A,B", [(100, 101), (0, 255), (17, 200), (255, 255)])
def test_compound_predictions_of_two_flat_references(backend, A, B):
    refs = [[np.full((H, W), v, np.uint8), np.full((H // 2, W // 2), v // 2, np.uint8), np.full((H // 2, W // 2), 255 - v // 2, np.uint8)] for v in (A, B)]
    # sub-pel vectors on purpose: a flat picture stays flat through every filter phase (taps sum to 128)
    base = dict(ref=(0, 1), mv=((3, -5), (-9, 13)))
    spec = uniform_grid(4, lambda x, y, fl: dict(flags=ir.BLK_INTER | ir.BLK_COMPOUND | fl, comp_type=ir.COMP_AVG, txs=whole_block_txs(x, y, 4), **base))
    got = decode(backend, frame_of(spec, True), refs)
    for pl in range(3):
        a, b = int(refs[0][pl][0, 0]), int(refs[1][pl][0, 0])
        assert (got[pl] == (a + b + 1) >> 1).all(), (pl, int(got[pl][0, 0]), (a + b + 1) >> 1)
    for w in (1, 5, 8, 13, 15):   # distance weights: w sixteenths of the first prediction
        spec = uniform_grid(4, lambda x, y, fl: dict(flags=ir.BLK_INTER | ir.BLK_COMPOUND | fl, comp_type=ir.COMP_DIST, comp_arg=w, txs=whole_block_txs(x, y, 4), **base))
        got = decode(backend, frame_of(spec, True), refs)
        for pl in range(3):
            a, b = int(refs[0][pl][0, 0]), int(refs[1][pl][0, 0])
            assert (got[pl] == (a * w + b * (16 - w) + 8) >> 4).all(), (w, pl, int(got[pl][0, 0]), (a * w + b * (16 - w) + 8) >> 4)


# ------------------------------------------------------------------------------------------------- inverse DCT, DC only
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("w,h", ir.TX_SIZES)
def test_dc_only_dct_blocks_add_the_closed_form_to_every_sample(backend, w, h):
    """inter blocks with zero motion over a flat reference of 120, each with ONE luma transform block per w x h cell holding
    a lone DC coefficient; chroma stays uncoded"""
    wl, hl = w.bit_length() - 1, h.bit_length() - 1
    b_log2 = max(wl, hl, 3)
    b = 1 << b_log2
    ref = [np.full((H, W), 120, np.uint8), np.full((H // 2, W // 2), 128, np.uint8), np.full((H // 2, W // 2), 128, np.uint8)]
    cvals = [64, -64, 1000, -1000, 2047, -2048, 7, -7, 300, 5000, -5000, 90]
    want = np.full((H, W), 120, np.int32)
    counter = [0]

    def make(x, y, fl):
        txs = []
        for ty in range(y, y + b, h):
            for tx in range(x, x + b, w):
                c = cvals[counter[0] % len(cvals)]
                counter[0] += 1
                cf = np.zeros((min(h, 32), min(w, 32)), np.int16)
                cf[0, 0] = c
                txs.append((0, tx, ty, wl, hl, 0, cf))
                want[ty:ty + h, tx:tx + w] += dc_residual(w, h, c)
        cl = b_log2 - 1
        txs += [(1, x // 2, y // 2, cl, cl, 0, None), (2, x // 2, y // 2, cl, cl, 0, None)]
        return dict(flags=ir.BLK_INTER | fl, ref=(0, 0), txs=txs)
    spec = uniform_grid(b_log2, make)
    got = decode(backend, frame_of(spec, True), [ref])
    assert np.array_equal(got[0], np.clip(want, 0, 255)), (w, h, got[0][0, :8].tolist(), np.clip(want, 0, 255)[0, :8].tolist())
    assert (got[1] == 128).all() and (got[2] == 128).all()


# ------------------------------------------------------------------------------------------------- intra edges
def intra_corner_frame(mode_a, mode_b, mode_c, mode_d, ra=0, rb=0, rc=0):
    """a key frame of 16x16 blocks, all DC except the four at the picture's top-left corner:
         A (0,0) no neighbours | B (16,0) left = A
         C (0,16) top = A      | D (16,16) top = B, left = C, corner = A
    ra / rb / rc: DC coefficients of A / B / C (luma 16x16, chroma 8x8 uncoded)"""
    special = {(0, 0): (mode_a, ra), (16, 0): (mode_b, rb), (0, 16): (mode_c, rc), (16, 16): (mode_d, 0)}

    def make(x, y, fl):
        mode, r = special.get((x, y), (ir.DC_PRED, 0))
        # top-right exists for blocks in the first column pair of a row whose upper-right neighbour is already decoded:
        # leave it off everywhere (none of the modes used here reads it); chroma flags follow the luma ones for >= 8x8
        cf = None
        if r:
            cf = np.zeros((16, 16), np.int16)
            cf[0, 0] = r
        return dict(flags=fl, mode_y=mode, mode_uv=ir.DC_PRED, txs=whole_block_txs(x, y, 4, luma=cf))
    return frame_of(uniform_grid(4, make), False)


def paeth(top, left, corner):
    base = top + left - corner
    pl_, pt, pc = abs(base - left), abs(base - top), abs(base - corner)
    return left if (pl_ <= pt and pl_ <= pc) else (top if pt <= pc else corner)


@pytest.mark.parametrize("backend", BACKENDS)
def test_intra_blocks_without_neighbours_use_the_fixed_edges(backend):
    for mode, want in ((ir.DC_PRED, 128), (ir.V_PRED, 127), (ir.H_PRED, 129), (ir.PAETH_PRED, paeth(127, 129, 128)), (ir.SMOOTH_V_PRED, None), (ir.SMOOTH_H_PRED, None)):
        got = decode(backend, intra_corner_frame(mode, ir.DC_PRED, ir.DC_PRED, ir.DC_PRED), [])
        a = got[0][0:16, 0:16]
        if want is not None:
            assert (a == want).all(), (mode, int(a[0, 0]), want)
        elif mode == ir.SMOOTH_V_PRED:   # every row blends the top row (127) with the bottom-left sample (129): columns equal, rows within [127, 129]
            assert (a == a[:, :1]).all() and a.min() >= 127 and a.max() <= 129 and a[0, 0] < a[15, 0]
        else:
            assert (a == a[:1, :]).all() and a.min() >= 127 and a.max() <= 129 and a[0, 0] > a[0, 15]
        # chroma of the same block: DC without neighbours
        assert (got[1][0:8, 0:8] == 128).all() and (got[2][0:8, 0:8] == 128).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_missing_intra_edges_are_filled_from_the_existing_neighbour(backend):
    ra = 1600
    a = 128 + dc_residual(16, 16, ra)                      # A is flat: DC 128 + residual
    assert a != 127 and a != 129 and a != 128
    # B has a left neighbour only: the missing top row is A's sample next to B's corner -> V, H, Paeth, DC all give a
    for mode in (ir.V_PRED, ir.H_PRED, ir.PAETH_PRED, ir.DC_PRED, ir.SMOOTH_PRED):
        got = decode(backend, intra_corner_frame(ir.DC_PRED, mode, ir.DC_PRED, ir.DC_PRED, ra=ra), [])
        assert (got[0][0:16, 0:16] == a).all()
        assert (got[0][0:16, 16:32] == a).all(), (mode, int(got[0][0, 16]), a)
    # C has a top neighbour only: same from above
    for mode in (ir.V_PRED, ir.H_PRED, ir.PAETH_PRED, ir.DC_PRED, ir.SMOOTH_PRED):
        got = decode(backend, intra_corner_frame(ir.DC_PRED, ir.DC_PRED, mode, ir.DC_PRED, ra=ra), [])
        assert (got[0][16:32, 0:16] == a).all(), (mode, int(got[0][16, 0]), a)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ra,rb,rc", [(1600, 800, -800), (-1600, 2400, 800), (800, -2400, -1600), (0, 1600, 1600)])
def test_intra_block_with_both_neighbours(backend, ra, rb, rc):
    """D sees a flat top edge (B), a flat left edge (C) and the corner sample of A: V copies the top, H the left, DC is
    their mean (16 t + 16 l + 16) >> 5, Paeth picks by its rule, the smooth modes blend between t and l"""
    a = 128 + dc_residual(16, 16, ra)
    t = a + dc_residual(16, 16, rb)                        # B: DC from its left edge (a everywhere) + residual
    l = a + dc_residual(16, 16, rc)                        # C: DC from its top edge + residual
    assert 0 < min(a, t, l) and max(a, t, l) < 255
    for mode, want in ((ir.V_PRED, t), (ir.H_PRED, l), (ir.DC_PRED, (16 * t + 16 * l + 16) >> 5), (ir.PAETH_PRED, paeth(t, l, a))):
        got = decode(backend, intra_corner_frame(ir.DC_PRED, ir.DC_PRED, ir.DC_PRED, mode, ra=ra, rb=rb, rc=rc), [])
        assert (got[0][0:16, 16:32] == t).all() and (got[0][16:32, 0:16] == l).all()
        assert (got[0][16:32, 16:32] == want).all(), (mode, int(got[0][16, 16]), want)
    got = decode(backend, intra_corner_frame(ir.DC_PRED, ir.DC_PRED, ir.DC_PRED, ir.SMOOTH_PRED, ra=ra, rb=rb, rc=rc), [])
    d = got[0][16:32, 16:32].astype(int)
    assert d.min() >= min(t, l) and d.max() <= max(t, l)
    if t != l:   # nearer the top edge it is nearer t, nearer the left edge nearer l
        assert abs(int(d[0, 15]) - t) < abs(int(d[15, 0]) - t)


# ------------------------------------------------------------------------------------------------- identity transforms
def identity_1d(n, x):
    """7.13.2.15: identity4 = Round2(x * 5793, 12), identity8 = 2x, identity16 = Round2(x * 11586, 12), identity32 = 4x"""
    return {4: round2(x * 5793, 12), 8: 2 * x, 16: round2(x * 11586, 12), 32: 4 * x}[n]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("w,h", [s for s in ir.TX_SIZES if max(s) <= 32])
def test_identity_transform_of_a_lone_coefficient_touches_one_sample(backend, w, h):
    """IDTX scales every coefficient on its own: a lone c at (0, 0) changes sample (0, 0) of the block and nothing else"""
    wl, hl = w.bit_length() - 1, h.bit_length() - 1
    b_log2 = max(wl, hl, 3)
    b = 1 << b_log2
    k = ir.TX_SIZES.index((w, h))
    ref = [np.full((H, W), 90, np.uint8), np.full((H // 2, W // 2), 128, np.uint8), np.full((H // 2, W // 2), 128, np.uint8)]
    want = np.full((H, W), 90, np.int32)
    cvals = [640, -640, 1999, -1999, 77, -77, 3000, -3000]
    counter = [0]

    def make(x, y, fl):
        txs = []
        for ty in range(y, y + b, h):
            for tx in range(x, x + b, w):
                c = cvals[counter[0] % len(cvals)]
                counter[0] += 1
                cf = np.zeros((h, w), np.int16)
                cf[0, 0] = c
                txs.append((0, tx, ty, wl, hl, ir.txtp(ir.TX_IDENTITY, ir.TX_IDENTITY), cf))
                v = round2(c * 2896, 12) if max(w, h) == 2 * min(w, h) else c
                v = round2(identity_1d(w, v), ROW_SHIFT[k])
                want[ty, tx] += round2(identity_1d(h, v), 4)
        cl = b_log2 - 1
        txs += [(1, x // 2, y // 2, cl, cl, 0, None), (2, x // 2, y // 2, cl, cl, 0, None)]
        return dict(flags=ir.BLK_INTER | fl, ref=(0, 0), txs=txs)
    got = decode(backend, frame_of(uniform_grid(b_log2, make), True), [ref])
    assert np.array_equal(got[0], np.clip(want, 0, 255)), (w, h, got[0][0, :4].tolist(), np.clip(want, 0, 255)[0, :4].tolist())


# ------------------------------------------------------------------------------------------------- chroma from luma
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("r1,r2,au,av", [(1600, 800, 8, -8), (-800, 1600, 16, 3), (1600, -2400, -13, 16), (800, 0, 5, -5)])
def test_chroma_from_luma_on_a_two_level_luma_block(backend, r1, r2, au, av):
    """block A (no neighbours), luma as two 8x16 transform blocks: the left one is DC 128 + res(r1) = a1, the right one
    predicts DC from its left column (a1) and adds res(r2): a2.  Chroma 8x8 with CfL (7.11.5): each chroma sample's luma
    value is the 2x2 sum << 1 = 8 a; their mean over the block is 4 (a1 + a2); pred = 128 + Round2Signed(alpha * (8 a - mean), 6)"""
    a1 = 128 + dc_residual(8, 16, r1)
    a2 = a1 + dc_residual(8, 16, r2)
    assert 0 <= min(a1, a2) and max(a1, a2) <= 255

    def cf(r):
        c = np.zeros((16, 8), np.int16)
        c[0, 0] = r
        return c

    def make(x, y, fl):
        if (x, y) != (0, 0):
            return dict(flags=fl, mode_y=ir.DC_PRED, mode_uv=ir.DC_PRED, txs=whole_block_txs(x, y, 4))
        txs = [(0, 0, 0, 3, 4, 0, cf(r1) if r1 else None), (0, 8, 0, 3, 4, 0, cf(r2) if r2 else None), (1, 0, 0, 3, 3, 0, None), (2, 0, 0, 3, 3, 0, None)]
        return dict(flags=fl, mode_y=ir.DC_PRED, mode_uv=ir.UV_CFL_PRED, cfl_alpha=(au, av), txs=txs)
    got = decode(backend, frame_of(uniform_grid(4, make), False), [])
    assert (got[0][0:16, 0:8] == a1).all() and (got[0][0:16, 8:16] == a2).all()

    def r2s(x):
        return ((abs(x) + 32) >> 6) * (1 if x >= 0 else -1)
    for pl, alpha in ((1, au), (2, av)):
        left = min(255, max(0, 128 + r2s(alpha * 4 * (a1 - a2))))
        right = min(255, max(0, 128 + r2s(alpha * 4 * (a2 - a1))))
        assert (got[pl][0:8, 0:4] == left).all() and (got[pl][0:8, 4:8] == right).all(), (pl, int(got[pl][0, 0]), left, int(got[pl][0, 7]), right)


# ------------------------------------------------------------------------------------------------- palette
@pytest.mark.parametrize("backend", BACKENDS)
def test_palette_blocks_are_colour_lookups(backend):
    rng = np.random.default_rng(3)
    py, pu, pv = [12, 80, 140, 200, 255], [30, 220], [5, 250]
    my = rng.integers(0, len(py), (16, 16))
    mc = rng.integers(0, len(pu), (8, 8))

    def make(x, y, fl):
        if (x, y) != (32, 16):
            return dict(flags=fl, mode_y=ir.DC_PRED, mode_uv=ir.DC_PRED, txs=whole_block_txs(x, y, 4))
        return dict(flags=fl | ir.BLK_PALETTE_Y | ir.BLK_PALETTE_UV, mode_y=ir.DC_PRED, mode_uv=ir.DC_PRED, palette=(py, pu, pv, my, mc), txs=whole_block_txs(x, y, 4))
    got = decode(backend, frame_of(uniform_grid(4, make), False), [])
    assert np.array_equal(got[0][16:32, 32:48], np.asarray(py)[my])
    assert np.array_equal(got[1][8:16, 16:24], np.asarray(pu)[mc]) and np.array_equal(got[2][8:16, 16:24], np.asarray(pv)[mc])
    # its neighbours predict from the palette block's samples like from any others: the block to the right, H_PRED-free DC,
    # sees the column of luma colours my[:, 15]
    col = np.asarray(py)[my][:, 15].astype(int)
    top = got[0][15, 48:64].astype(int)
    assert (got[0][16:32, 48:64] == (col.sum() + top.sum() + 16) >> 5).all()


# ------------------------------------------------------------------------------------------------- directional modes
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ra,rb,rc", [(1600, 800, -800), (-800, -1600, 2400)])
def test_directional_modes_between_flat_edges(backend, ra, rb, rc):
    """block D between a flat top edge t and a flat left edge l (edge filter off; no top-right / bottom-left: the last edge
    sample is repeated): modes pointing up-right (D45, D67) read only the top row -> t; the one pointing down-left (D203)
    reads only the left column -> l; the ones in between read both and the corner a: every sample is one of the three or
    lies between them"""
    a = 128 + dc_residual(16, 16, ra)
    t = a + dc_residual(16, 16, rb)
    l = a + dc_residual(16, 16, rc)
    for mode, want in ((ir.D45_PRED, t), (ir.D67_PRED, t), (ir.D203_PRED, l)):
        got = decode(backend, intra_corner_frame(ir.DC_PRED, ir.DC_PRED, ir.DC_PRED, mode, ra=ra, rb=rb, rc=rc), [])
        assert (got[0][16:32, 16:32] == want).all(), (mode, int(got[0][16, 16]), want)
    lo, hi = min(a, t, l), max(a, t, l)
    for mode in (ir.D113_PRED, ir.D135_PRED, ir.D157_PRED):
        got = decode(backend, intra_corner_frame(ir.DC_PRED, ir.DC_PRED, ir.DC_PRED, mode, ra=ra, rb=rb, rc=rc), [])
        d = got[0][16:32, 16:32]
        assert d.min() >= lo and d.max() <= hi
        assert d[0, 15] == t and d[15, 0] == l   # far from the corner along the edges the nearer edge alone is read
