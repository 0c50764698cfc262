"""Known-answer tests of the in-loop filters whose expected values are HAND-DERIVED from the closed forms of the AV1
specification for simple inputs (flat fields, step edges, stripes, impulses) -- written out below as arithmetic on the
input values, not produced by the oracle.  They check the oracle on the CPU and the HIP kernels with -m gpu, and they
are the only video checks here that do not share code with either.

How a filter gets a chosen input: an inter frame of skipped (or coefficient-less) blocks with zero motion copies a
crafted reference picture sample by sample, so the picture in front of the filters IS that reference."""
import numpy as np
import pytest

from easyav1_amd import ir
from tests import oracle_lib as ol

W, H = 128, 64


def copy_frame(block_log2, tx_log2, skip=False, **desc_fields):
    """an inter frame of (1 << block_log2)^2 blocks, each with a uniform luma transform grid of (1 << tx_log2)^2 and the
    co-located chroma grid, no coefficients, zero motion from slot 0, finished picture into slot 1"""
    b, t = 1 << block_log2, 1 << tx_log2
    tc_log2 = max(tx_log2 - 1, 2)
    tc = 1 << tc_log2
    blocks, txs = [], []
    for y in range(0, H, b):
        for x in range(0, W, b):
            blk = np.zeros(1, ir.BLOCK_DTYPE)[0]
            blk["x4"], blk["y4"], blk["bwl"], blk["bhl"] = x // 4, y // 4, block_log2, block_log2
            blk["flags"] = ir.BLK_INTER | (ir.BLK_SKIP if skip else 0) | (ir.BLK_HAVE_LEFT if x else 0) | (ir.BLK_HAVE_TOP if y else 0)
            blk["first_tx"] = len(txs)
            for ty in range(y, y + b, t):
                for tx in range(x, x + b, t):
                    txs.append((tx // 4, ty // 4, 0, tx_log2 | (tx_log2 << 4), 0, 0, ir.NO_COEFS))
            for pl in (1, 2):
                for ty in range(y // 2, (y + b) // 2, tc):
                    for tx in range(x // 2, (x + b) // 2, tc):
                        txs.append((tx // 4, ty // 4, pl, tc_log2 | (tc_log2 << 4), 0, 0, ir.NO_COEFS))
            blk["n_tx"] = len(txs) - blk["first_tx"]
            blocks.append(blk)
    d = ir.FrameDesc()
    d.width, d.height, d.is_inter, d.dst_slot = W, H, 1, 1
    d.cdef_damping = 3
    for k, v in desc_fields.items():
        if isinstance(v, (list, tuple)):
            for i, e in enumerate(v):
                getattr(d, k)[i] = e
        else:
            setattr(d, k, v)
    lr = [np.zeros(0, ir.LR_UNIT_DTYPE)] * 3
    return d, np.array(blocks, ir.BLOCK_DTYPE), np.array(txs, ir.TXBLOCK_DTYPE), lr


def run(backend, desc, blocks, txs, lr, ref, cdef_idx=None, stages=ir.STAGE_ALL):
    sbs = ((W + 63) // 64) * ((H + 63) // 64)
    f = ir.ParsedFrame(desc, blocks, txs, np.zeros(0, np.int16), np.zeros(0, np.uint8), np.zeros(sbs, np.uint8) if cdef_idx is None else cdef_idx, lr)
    if backend == "oracle":
        return ol.oracle_decode_frame(f, [ref] + [None] * 7, stages)["final"]
    from easyav1_amd import video
    ctx = video.ReconContext(W, H)
    ctx.set_ref(0, ref, pad=desc.pad, crop=desc.crop if desc.pad else 0)
    job = ctx.make_job(f)
    ctx.submit(job, stages)
    out = ctx.download(1)
    job.close()
    ctx.close()
    return out


BACKENDS = ["oracle", pytest.param("gpu", marks=pytest.mark.gpu)]


def step_picture(a, b, ca=None, cb=None):
    """luma a | b split at x = W / 2, chroma ca | cb at the co-located column; constant along y"""
    ca, cb = a if ca is None else ca, b if cb is None else cb
    y = np.full((H, W), a, np.uint8)
    y[:, W // 2:] = b
    c = np.full((H // 2, W // 2), ca, np.uint8)
    c[:, W // 4:] = cb
    return [y, c.copy(), c.copy()]


# ---- deblocking, level 20, sharpness 0 (AV1 spec 7.14.6): limit = 20, blimit = 2 * (20 + 2) + 20 = 64, thresh = 20 >> 4 = 1.
# A step a | b between two flat sides passes the filter mask when 2 * |a - b| + (|a - b| >> 1) <= 64, is never "high edge
# variance" (|p1 - p0| = 0) and is "flat" on both sides (all |p_i - p_0| = 0 <= 1).  The row of samples across the edge is
# then: narrow filter (transform 4): f = 3 * (b - a), f1 = (f + 4) >> 3, f2 = (f + 3) >> 3, f3 = (f1 + 1) >> 1:
#       p1 + f3, p0 + f2 | q0 - f1, q1 - f3;  8-sample filter: p2..q2 = Round2(7a + b, 3), (6a + 2b), (5a + 3b) | (3a + 5b), (2a + 6b), (a + 7b);
# 14-sample filter (13 taps, the two nearest the output doubled): the b-side weights out of 16 are 1, 2, 3, 4, 5, 7 for p5 .. p0 and
# 9, 11, 12, 13, 14, 15 for q0 .. q5: Round2((16 - k) * a + k * b, 4);  chroma 6-sample filter: p1, p0 | q0, q1 =
# Round2(7a + b, 3), Round2(5a + 3b, 3) | Round2(3a + 5b, 3), Round2(a + 7b, 3).
def expected_row(a, b, kind):
    d = b - a
    if kind == 4:
        f = 3 * d
        f1, f2 = (f + 4) >> 3, (f + 3) >> 3
        f3 = (f1 + 1) >> 1
        return [a + f3, a + f2], [b - f1, b - f3]
    if kind == 6:
        return [(7 * a + b + 4) >> 3, (5 * a + 3 * b + 4) >> 3], [(3 * a + 5 * b + 4) >> 3, (a + 7 * b + 4) >> 3]
    if kind == 8:
        return [(7 * a + b + 4) >> 3, (6 * a + 2 * b + 4) >> 3, (5 * a + 3 * b + 4) >> 3], [(3 * a + 5 * b + 4) >> 3, (2 * a + 6 * b + 4) >> 3, (a + 7 * b + 4) >> 3]
    return [((16 - k) * a + k * b + 8) >> 4 for k in (1, 2, 3, 4, 5, 7)], [((16 - k) * a + k * b + 8) >> 4 for k in (9, 11, 12, 13, 14, 15)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("tx_log2,luma_kind,chroma_kind", [(2, 4, 4), (3, 8, 4), (4, 14, 6), (5, 14, 6), (6, 14, 6)])
@pytest.mark.parametrize("a,b", [(100, 110), (200, 180), (17, 18), (90, 115)])
def test_deblock_step_edges_follow_the_closed_forms(backend, tx_log2, luma_kind, chroma_kind, a, b):
    block_log2 = max(tx_log2, 3)
    desc, blocks, txs, lr = copy_frame(block_log2, tx_log2, lf_level=[20, 20, 20, 20])
    ref = step_picture(a, b)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_DEBLOCK)
    for pl, kind in ((0, luma_kind), (1, chroma_kind), (2, chroma_kind)):
        want = ref[pl].astype(np.int32)
        edge = want.shape[1] // 2
        p, q = expected_row(a, b, kind)
        want[:, edge - len(p):edge] = p
        want[:, edge:edge + len(q)] = q
        assert np.array_equal(got[pl], want), (pl, kind, got[pl][3, edge - 8:edge + 8].tolist(), want[3, edge - 8:edge + 8].tolist())


@pytest.mark.parametrize("backend", BACKENDS)
def test_deblock_leaves_real_edges_and_level_zero_alone(backend):
    # 2 * 26 + 13 = 65 > blimit 64: the step is image content, not a blocking artefact
    desc, blocks, txs, lr = copy_frame(4, 4, lf_level=[20, 20, 20, 20])
    ref = step_picture(100, 126)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_DEBLOCK)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))
    # levels of 0 switch the planes off one by one (7.14.1): luma vertical edges off -> nothing at all is filtered
    desc, blocks, txs, lr = copy_frame(4, 4, lf_level=[0, 0, 20, 20])
    ref = step_picture(100, 110)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_DEBLOCK)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref))
    # inside a skipped inter block the transform edges are not filtered, its outer edge is (7.14.2): 64x64 blocks of
    # 16x16 transforms, the step at x = 48 lies inside a block, the one at x = 64 between two
    desc, blocks, txs, lr = copy_frame(6, 4, skip=True, lf_level=[20, 20, 20, 20])
    for edge, filtered in ((48, False), (64, True)):
        y = np.full((H, W), 100, np.uint8)
        y[:, edge:] = 110
        ref = [y, np.full((H // 2, W // 2), 128, np.uint8), np.full((H // 2, W // 2), 128, np.uint8)]
        got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_DEBLOCK)
        assert np.array_equal(got[0], ref[0]) != filtered


# ---- CDEF (7.15).  With secondary strength 0 only the two primary taps on either side along the block's direction act;
# a picture that is constant along that direction gives every tap the centre value: differences 0, output = input.
# Directions (Cdef_Directions, (dy, dx) of the first tap): 0 up-right diagonal (-1, +1), 2 horizontal (0, +1),
# 4 down-right diagonal (+1, +1), 6 vertical (+1, 0).  A direction found wrong moves samples: stripes along another
# direction are the control.
def stripes(dy, dx, period=8, lo=100, hi=104):
    """low-contrast stripes (a step of 4 passes constrain() unchanged at these strengths; a strong edge would be left alone by
    ANY tap and prove nothing), constant along the vector (dy, dx)"""
    yy, xx = np.mgrid[0:H, 0:W]
    t = (xx * dy - yy * dx) % period
    return np.where(t < period // 2, lo, hi).astype(np.uint8)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("direction,dy,dx", [(0, -1, 1), (2, 0, 1), (4, 1, 1), (6, 1, 0)])
def test_cdef_primary_taps_run_along_the_detected_direction(backend, direction, dy, dx):
    flat = np.full((H // 2, W // 2), 128, np.uint8)
    ref = [stripes(dy, dx), flat, flat.copy()]
    inner = (slice(8, -8), slice(8, -8))   # at the picture border taps are missing and the 8x8 blocks see clipped stripes
    # primary taps only: they lie along the stripes, every difference is 0
    desc, blocks, txs, lr = copy_frame(3, 3, cdef_enable=1, cdef_damping=6, cdef_y_strength=[15 << 2] * 8, cdef_uv_strength=[0] * 8)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_CDEF)
    assert np.array_equal(got[0][inner], ref[0][inner]), direction
    # secondary taps only: they lie 45 degrees off the stripes and cross them -- the same picture does change, i.e. the
    # filter is on and it was the direction that kept the primary taps quiet
    desc, blocks, txs, lr = copy_frame(3, 3, cdef_enable=1, cdef_damping=6, cdef_y_strength=[3] * 8, cdef_uv_strength=[0] * 8)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_CDEF)
    assert not np.array_equal(got[0][inner], ref[0][inner]), direction
    assert np.array_equal(got[1], flat) and np.array_equal(got[2], flat)


# A single sample x + e in a flat chroma plane (luma flat: direction 0, variance 0).  Chroma: damping - 1, primary
# strength unscaled.  constrain(d, s, D) = sign(d) * min(|d|, max(0, s - (|d| >> (D - floor(log2 s))))).
# The sample itself sees 12 taps of difference -e: primary weights (even strength) 4 + 2 on both sides, secondary 2 + 1 on both
# sides of two directions: sum = -(12 * cp + 12 * cs), y = x + e + ((8 + sum - 1) >> 4).  A neighbour that has the sample as
# its first primary tap (one step along direction 0: up-right / down-left) sees +e once with weight 4: y = x + ((8 + 4 * cp) >> 4).
def constrain(d, s, damping):
    if s == 0:
        return 0
    mag = abs(d)
    v = min(mag, max(0, s - (mag >> max(0, damping - (s.bit_length() - 1)))))
    return v if d > 0 else -v


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("pri,sec,damping,e", [(4, 2, 4, 6), (8, 1, 5, 9), (2, 4, 3, 5), (12, 0, 6, 20), (0, 4, 5, 7)])
def test_cdef_impulse_in_flat_chroma_matches_the_tap_arithmetic(backend, pri, sec, damping, e):
    sec_code = 3 if sec == 4 else sec
    desc, blocks, txs, lr = copy_frame(3, 3, cdef_enable=1, cdef_damping=damping, cdef_y_strength=[0] * 8, cdef_uv_strength=[(pri << 2) | sec_code] * 8)
    x = 100
    c = np.full((H // 2, W // 2), x, np.uint8)
    cy, cx = 17, 21
    c[cy, cx] = x + e
    ref = [np.full((H, W), 128, np.uint8), c, np.full((H // 2, W // 2), x, np.uint8)]
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_CDEF)
    D = damping - 1
    cp, cs = constrain(-e, pri, D), constrain(-e, sec, D)
    w0, w1 = (4, 2) if pri % 2 == 0 else (3, 3)
    total = 2 * (w0 + w1) * cp + 2 * 2 * (2 + 1) * cs
    assert got[1][cy, cx] == x + e + ((8 + total - (1 if total < 0 else 0)) >> 4)
    near = w0 * constrain(e, pri, D)               # the sample one step along direction 0 from the impulse
    assert got[1][cy + 1, cx - 1] == got[1][cy - 1, cx + 1] == x + ((8 + near) >> 4)
    far = w1 * constrain(e, pri, D)
    assert got[1][cy + 2, cx - 2] == got[1][cy - 2, cx + 2] == x + ((8 + far) >> 4)
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[0], ref[0])     # flat planes stay flat


# ---- deblocking at the edge of a frame whose true size is not a multiple of 8 (7.14.2 "onScreen"; dav1d lf_apply: columns / rows up
# to (size + 3) >> 2): a 4x4 unit that STARTS at or beyond the true size is not filtered -- and CDEF of visible samples reads there.
# A lone DC coefficient of 1024 in a 4x4 DCT block adds Round2(Round2(Round2(1024 * 2896, 12) * 2896, 12), 4) = Round2(512, 4) = 32
# to its 16 samples (724 after the row pass, 512 after the column pass), so blocks with it stand 32 above a flat 100: across such a
# step the narrow filter of level 28 (2 * 32 + 32 / 2 = 80 <= blimit 2 * 30 + 28 = 88; no high edge variance) gives f = 96, f1 = f2 = 12, f3 = 6:
# 100 100 | 132 132 -> 106 112 | 120 126.
def dc_step_frame(crop, first_unit, horizontal):
    desc, blocks, txs, lr = copy_frame(4, 2, lf_level=[28, 28, 0, 0], crop=crop)
    coefs = []
    for t in txs:
        pos = t["y4"] if horizontal else t["x4"]
        if t["plane"] == 0 and pos >= first_unit:
            t["coef_off"] = len(coefs)
            coefs += [1024] + [0] * 15
    return desc, blocks, txs, lr, np.array(coefs, np.int16)


def run_with_coefs(backend, desc, blocks, txs, lr, coefs, ref, stages):
    f = ir.ParsedFrame(desc, blocks, txs, coefs, np.zeros(0, np.uint8), np.zeros(((W + 63) // 64) * ((H + 63) // 64), np.uint8), lr)
    if backend == "oracle":
        return ol.oracle_decode_frame(f, [ref] + [None] * 7, stages)["final"]
    from easyav1_amd import video
    ctx = video.ReconContext(W, H)
    ctx.set_ref(0, ref, crop=desc.crop)
    job = ctx.make_job(f)
    ctx.submit(job, stages)
    out = ctx.download(1)
    job.close()
    ctx.close()
    return out


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("horizontal", [False, True])
def test_deblocking_stops_at_the_true_frame_size(backend, horizontal):
    ref = [np.full((H, W), 100, np.uint8), np.full((H // 2, W // 2), 50, np.uint8), np.full((H // 2, W // 2), 60, np.uint8)]
    size = H if horizontal else W
    line = lambda pic: (pic[0][:, 5] if horizontal else pic[0][5, :]).astype(int).tolist()
    stages = ir.STAGE_RECON | ir.STAGE_DEBLOCK
    # true size = coded size - 4: the last 4x4 unit column / row starts AT the true size -> its edge stays
    crop = (4 << 4) if horizontal else 4
    got = line(run_with_coefs(backend, *dc_step_frame(crop, size // 4 - 1, horizontal), ref, stages))
    assert got[size - 8:] == [100, 100, 100, 100, 132, 132, 132, 132], got[size - 8:]
    # the same step one unit further in: filtered (and the off-screen unit beside it left alone)
    got = line(run_with_coefs(backend, *dc_step_frame(crop, size // 4 - 2, horizontal), ref, stages))
    assert got[size - 12:] == [100, 100, 106, 112, 120, 126, 132, 132, 132, 132, 132, 132], got[size - 12:]
    # true size = coded size - 3: the last unit starts inside the frame -> filtered
    crop = (3 << 4) if horizontal else 3
    got = line(run_with_coefs(backend, *dc_step_frame(crop, size // 4 - 1, horizontal), ref, stages))
    assert got[size - 8:] == [100, 100, 106, 112, 120, 126, 132, 132], got[size - 8:]


# ---- loop restoration (7.17).  Wiener: taps c0 c1 c2 of the stream, c3 = 128 - 2 (c0 + c1 + c2); for a picture that is constant
# along y the vertical pass has unit gain, so a sample is Round2(Round2(sum_k h_k * x_k, 3), 4) with the 7 horizontal taps
# (for 8-bit video the intermediate offsets cancel and no clamp is reached).  Across a step a | b the sample j places right of the
# edge (j = 0 is the first b sample) sees 3 + j + 1 taps on the b side for j < 3.
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("taps", [(3, -7, 15), (0, 0, 0), (-5, 10, 20), (1, 2, 3)])
def test_wiener_horizontal_step_response(backend, taps):
    a, b = 80, 144
    desc, blocks, txs, _ = copy_frame(4, 4, lr_type=[1, 1, 1], lr_unit_log2=[6, 5, 5])
    c0, c1, c2 = taps
    h = [c0, c1, c2, 128 - 2 * (c0 + c1 + c2), c2, c1, c0]
    lr = []
    for pl in range(3):
        ss = 1 if pl else 0
        n = max(((H >> ss) + (32 >> ss)) // (64 >> ss), 1) * max(((W >> ss) + (32 >> ss)) // (64 >> ss), 1)
        u = np.zeros(n, ir.LR_UNIT_DTYPE)
        u["type"] = ir.LR_WIENER
        u["coef"][:, 0, :] = (0, 0, 0)             # vertical: identity (its centre tap becomes 128)
        u["coef"][:, 1, :] = taps if pl == 0 else (0, c1, c2)   # chroma filters have 5 taps: c0 = 0
        lr.append(u)
    ref = step_picture(a, b)
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_LR)
    for pl in range(3):
        hh = h if pl == 0 else [0, c1, c2, 128 - 2 * (c1 + c2), c2, c1, 0]
        row = ref[pl][0].astype(np.int64)
        padded = np.concatenate([[row[0]] * 3, row, [row[-1]] * 3])
        want = np.array([(((sum(hh[k] * padded[i + k] for k in range(7)) + 4) >> 3) + 8) >> 4 for i in range(len(row))])
        want = np.clip(want, 0, 255)
        assert np.array_equal(got[pl], np.tile(want.astype(np.uint8), (ref[pl].shape[0], 1))), (pl, got[pl][0, W // (4 if pl else 2) - 5:W // (4 if pl else 2) + 5].tolist())


# The intermediate clamp of 7.17.4: offset = 1 << (BitDepth + FILTER_BITS - InterRound0 - 1) = 1 << (8 + 7 - 3 - 1) = 2048,
# limit = (1 << (BitDepth + 1 + FILTER_BITS - InterRound0)) - 1 = 8191, intermediate = Clip3(-offset, limit - offset, v) =
# [-2048, 6143].  The extreme legal horizontal taps (-5, -23, -17) give a centre tap of 128 + 2 * 45 = 218 and reach both ends:
#   * a lone 255 on 0: Round2(218 * 255, 3) = (55590 + 4) >> 3 = 6949 -> 6143
#   * a lone 0 on 255: Round2(128 * 255 - 218 * 255, 3) = (-22950 + 4) >> 3 = -2869 -> -2048, the rows around it Round2(32640, 3) = 4080
# Seen through the vertical taps (10, 8, 46) (centre 128 - 2 * 64 = 0) in the impulse's column, d rows away (weight c2, c1, c0
# for d = 1, 2, 3; every other row of the first picture is 0, of the second 4080):
#   first : Round2(46 * 6143, 11) = 138, Round2(8 * 6143, 11) = 24, Round2(10 * 6143, 11) = 30   (an unclamped 6949 would give 156, 27, 34)
#   second: Round2(82 * 4080 - 46 * 2048, 11) = 117, Round2(120 * 4080 - 8 * 2048, 11) = 231, Round2(118 * 4080 - 10 * 2048, 11) = 225
#           (an unclamped -2869 would give 99, 228, 221)
# Chroma (5 taps: c0 = 0): horizontal (0, -23, -17), centre 208: Round2(208 * 255, 3) = 6630 -> 6143 and Round2(-80 * 255, 3) = -2550 -> -2048;
# vertical (0, 8, 46), centre 20:  first: d = 0: Round2(20 * 6143, 11) = 60, d = 1: 138, d = 2: 24;
#   second: d = 0: Round2(108 * 4080 - 20 * 2048, 11) = 195, d = 1: 117, d = 2: 231.
def wiener_expected(pic, hc, vc, chroma):
    """7.17.4 written out with numpy on a whole plane (edges replicated; the test keeps the impulse inside one stripe)"""
    h = [0 if chroma else hc[0], hc[1], hc[2]]
    v = [0 if chroma else vc[0], vc[1], vc[2]]
    hf = h + [128 - 2 * sum(h)] + h[::-1]
    vf = v + [128 - 2 * sum(v)] + v[::-1]
    x = np.pad(pic.astype(np.int64), 3, mode="edge")
    hh, ww = pic.shape
    mid = sum(hf[t] * x[:, t:t + ww] for t in range(7))
    mid = np.clip((mid + 4) >> 3, -2048, 6143)
    out = sum(vf[t] * mid[t:t + hh, :] for t in range(7))
    return np.clip((out + 1024) >> 11, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("backend", BACKENDS)
def test_wiener_intermediate_clamp_both_ends(backend):
    desc, blocks, txs, _ = copy_frame(4, 4, lr_type=[1, 1, 1], lr_unit_log2=[6, 5, 5])
    hc, vc = (-5, -23, -17), (10, 8, 46)
    lr = []
    for pl in range(3):
        ss = 1 if pl else 0
        n = max(((H >> ss) + (32 >> ss)) // (64 >> ss), 1) * max(((W >> ss) + (32 >> ss)) // (64 >> ss), 1)
        u = np.zeros(n, ir.LR_UNIT_DTYPE)
        u["type"] = ir.LR_WIENER
        u["coef"][:, 0, :] = vc if pl == 0 else (0, vc[1], vc[2])
        u["coef"][:, 1, :] = hc if pl == 0 else (0, hc[1], hc[2])
        lr.append(u)
    yx, yy, cx, cy = 40, 24, 20, 12       # inside the first stripe of each plane (luma rows 0 .. 55, chroma rows 0 .. 27), away from the edges
    for bg, fg, luma_d, chroma_d in ((0, 255, {1: 138, 2: 24, 3: 30}, {0: 60, 1: 138, 2: 24}),
                                     (255, 0, {1: 117, 2: 231, 3: 225}, {0: 195, 1: 117, 2: 231})):
        ref = [np.full((H, W), bg, np.uint8), np.full((H // 2, W // 2), bg, np.uint8), np.full((H // 2, W // 2), bg, np.uint8)]
        ref[0][yy, yx] = fg
        ref[1][cy, cx] = ref[2][cy, cx] = fg
        got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_LR)
        for d, want in luma_d.items():
            assert got[0][yy + d, yx] == want and got[0][yy - d, yx] == want, (bg, d, int(got[0][yy + d, yx]), want)
        for pl in (1, 2):
            for d, want in chroma_d.items():
                assert got[pl][cy + d, cx] == want and got[pl][cy - d, cx] == want, (bg, pl, d, int(got[pl][cy + d, cx]), want)
        for pl in range(3):
            assert np.array_equal(got[pl], wiener_expected(ref[pl], hc, vc, pl > 0)), (bg, pl)


# Self-guided filter on a flat field of value v (7.17.3): every box has sum n v and sum of squares n v^2, so p = 0, z = 0,
# a = x_by_xplus1[0] = 1 and b = Round2((256 - 1) * (n v) * one_by_n, 12) with one_by_n = Round(4096 / n) (164 for the 5x5 box,
# 455 for the 3x3 box); the weights of the 3x3 neighbourhood add up to 32, so filtered = Round2(32 * a * v + 32 * b, 9 resp. 8 + 1)
# = (v + b + 8) >> 4 per pass, and the projection with w0 + w1 + w2 = 128 returns Round2(w0 f0 + w1 (v << 4) + w2 f1, 11) with
# f = filtered value.  b is within +-1 of 16 v - v, so f is 16 v (+-1) and the output is v for every legal weight pair.
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("sgr_set,w0,w1", [(0, 30, 40), (5, -20, 90), (9, 31, -32), (11, 0, 100), (14, 50, 78), (15, -96, 127)])
def test_self_guided_filter_keeps_flat_fields(backend, sgr_set, w0, w1):
    desc, blocks, txs, _ = copy_frame(4, 4, lr_type=[1, 1, 1], lr_unit_log2=[6, 5, 5])
    lr = []
    for pl in range(3):
        ss = 1 if pl else 0
        n = max(((H >> ss) + (32 >> ss)) // (64 >> ss), 1) * max(((W >> ss) + (32 >> ss)) // (64 >> ss), 1)
        u = np.zeros(n, ir.LR_UNIT_DTYPE)
        u["type"] = ir.LR_SGRPROJ
        u["sgr_set"] = sgr_set
        u["coef"][:, 0, 0], u["coef"][:, 0, 1] = w0, w1
        lr.append(u)
    for v in (0, 1, 77, 128, 254, 255):
        ref = [np.full((H, W), v, np.uint8), np.full((H // 2, W // 2), 255 - v, np.uint8), np.full((H // 2, W // 2), v // 2, np.uint8)]
        # the arithmetic of the comment above, spelled out for this v
        for val in (v, 255 - v, v // 2):
            outs = []
            for n_box, one_by_n in ((25, 164), (9, 455)):
                bb = (255 * (n_box * val) * one_by_n + 2048) >> 12
                outs.append((32 * 1 * val + 32 * bb + 256) >> 9)
            r0, r1 = ((2, 1) if sgr_set < 10 else ((0, 1) if sgr_set < 14 else (2, 0)))
            f0 = outs[0] if r0 else val << 4
            f1 = outs[1] if r1 else val << 4
            assert ((w0 * f0 + w1 * (val << 4) + (128 - w0 - w1) * f1 + 1024) >> 11) == val or abs(f0 - (val << 4)) <= 1
        got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_LR)
        for pl in range(3):
            diff = np.abs(got[pl].astype(int) - ref[pl].astype(int)).max()
            assert diff <= (1 if abs(w0) + abs(128 - w0 - w1) > 128 else 0), (v, pl, diff)


# ---- CDEF: the clip to the range of the taps (7.15.2: CdefFrame = Clip3(min, max, x + ...)).  A chroma sample 2 ABOVE a flat plane
# with primary and secondary strength 4 at damping 6 (chroma: 5; shift 5 - 2 = 3, so constrain(-2) = -min(2, 4 - (2 >> 3)) = -2 for
# all 12 taps): sum = -(4 + 4 + 2 + 2) * 2 - (2 * 4 + 1 * 4) * 2 = -48, x + 2 + ((8 - 48 - 1) >> 4) = x + 2 - 3 = x - 1: BELOW every tap
# (all of them x) and below the sample itself -> clipped to min = x.  The mirror image (2 below) is clipped to max = x.
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("e", [2, -2])
def test_cdef_result_is_clipped_to_the_range_of_its_taps(backend, e):
    desc, blocks, txs, lr = copy_frame(3, 3, cdef_enable=1, cdef_damping=6, cdef_y_strength=[0] * 8, cdef_uv_strength=[(4 << 2) | 3] * 8)
    x = 100
    c = np.full((H // 2, W // 2), x, np.uint8)
    cy, cx = 17, 21
    c[cy, cx] = x + e
    ref = [np.full((H, W), 128, np.uint8), c, np.full((H // 2, W // 2), x, np.uint8)]
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_CDEF)
    total = -48 if e > 0 else 48
    unclipped = x + e + ((8 + total - (1 if total < 0 else 0)) >> 4)
    assert unclipped == (x - 1 if e > 0 else x + 1)         # the arithmetic alone overshoots the flat level by one
    assert got[1][cy, cx] == x, (int(got[1][cy, cx]), unclipped)


# ---- deblocking with per-block levels (7.14.4): an edge takes the level of the block on its q side (the current block), or of the
# block across the edge when that is 0; both 0: not filtered.  Two 16-wide columns of blocks meet at x = 64 with a step of 32
# (a DC coefficient of 1024 in every 4x4 transform of the right half, see above): with level 28 on either side the narrow filter
# gives 106 112 | 120 126, with 0 on both the step stays.
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("left_lvl,right_lvl,filtered", [(28, 28, True), (0, 28, True), (28, 0, True), (0, 0, False)])
def test_deblock_level_of_the_other_block_when_the_own_one_is_zero(backend, left_lvl, right_lvl, filtered):
    desc, blocks, txs, lr = copy_frame(4, 2, lf_level=[28, 28, 0, 0], lf_per_block=1)
    coefs = []
    for t in txs:
        if t["plane"] == 0 and t["x4"] >= W // 8:
            t["coef_off"] = len(coefs)
            coefs += [1024] + [0] * 15
    for b in blocks:
        lvl = right_lvl if b["x4"] >= W // 8 else left_lvl
        b["lf_level"] = (lvl, lvl, 0, 0)
    ref = [np.full((H, W), 100, np.uint8), np.full((H // 2, W // 2), 50, np.uint8), np.full((H // 2, W // 2), 60, np.uint8)]
    got = run_with_coefs(backend, desc, blocks, txs, lr, np.array(coefs, np.int16), ref, ir.STAGE_RECON | ir.STAGE_DEBLOCK)
    row = got[0][9, W // 2 - 4:W // 2 + 4].astype(int).tolist()
    assert row == ([100, 100, 106, 112, 120, 126, 132, 132] if filtered else [100, 100, 100, 100, 132, 132, 132, 132]), row


# ---- film grain (7.18.3) with a CONSTANT Gaussian sequence (every entry 1024; the table is caller-supplied data, so a test may pick
# it) and lag 0: the random offsets cannot matter, every step has a closed form.
#   grain templates    Round2(1024, 12 - 8 + grain_scale_shift 0) = 64 everywhere in luma.  Chroma at lag 0 has ONE coefficient, on the
#                      co-located luma average (64): Cb 64 + Round2(64 * 8, 6) = 72; Cr 64 + Round2(64 * -128, 6) = 64 + ((-8192 + 32) >> 6)
#                      = 64 - 128 = -64  (Round2 of a negative number floors: -127.5 -> -128)
#   overlap            luma columns 32k, 32k+1 (k >= 1) blend 27/17 and 17/27 (both sum 44): (64 * 44 + 16) >> 5 = 88; rows 32n, 32n+1 the
#                      same on top: plain columns 88, overlapped ones (88 * 44 + 16) >> 5 = 121.  Chroma column / row 16k blends 23/22:
#                      Cb (72 * 45 + 16) >> 5 = 101, crossing (101 * 45 + 16) >> 5 = 142 -> clipped to 127; Cr (-64 * 45 + 16) >> 5 = -90
#                      (-89.5 floors), crossing (-90 * 45 + 16) >> 5 = -127 (-126.06 floors)
#   scaling            one point per plane = a constant function: luma 80, Cb 64, Cr 128, scaling_shift 8:
#                      luma Round2(80 * {64, 88, 121}, 8) = {20, 28, 38}; Cb Round2(64 * {72, 101, 127}, 8) = {18, 25, 32};
#                      Cr Round2(128 * {-64, -90, -127}, 8) = {-32 (-31.5 floors), -45 (-44.5 floors), -63 (-16128 / 256 exactly, + .5 floors)}
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("overlap", [0, 1])
def test_film_grain_closed_form_with_a_constant_gaussian_sequence(backend, overlap):
    from easyav1_amd import grain
    w, h = 96, 64
    fg = grain.FilmGrain()
    fg.grain_seed = 0x1234
    fg.num_y_points, fg.point_y[0][0], fg.point_y[0][1] = 1, 7, 80
    for pl, (x, s) in enumerate([(99, 64), (0, 128)]):
        fg.num_uv_points[pl], fg.point_uv[pl][0][0], fg.point_uv[pl][0][1] = 1, x, s
    fg.scaling_shift, fg.ar_coeff_lag, fg.ar_coeff_shift, fg.grain_scale_shift = 8, 0, 6, 0
    fg.ar_coeffs_uv[0][0], fg.ar_coeffs_uv[1][0] = 8, -128
    for pl in range(2):
        fg.uv_mult[pl], fg.uv_luma_mult[pl], fg.uv_offset[pl] = 128, 192, 256
    fg.overlap_flag = overlap
    seq = np.full(2048, 1024, np.int16)
    pic = [np.full((h, w), 100, np.uint8), np.full((h // 2, w // 2), 90, np.uint8), np.full((h // 2, w // 2), 160, np.uint8)]
    try:
        if backend == "oracle":
            ol.oracle_set_gaussian_sequence(seq)
            got = ol.oracle_film_grain(fg, pic)
        else:
            from easyav1_amd import video
            ctx = video.ReconContext(w, h)
            grain.set_gaussian_sequence(ctx, seq)
            ctx.set_ref(0, pic)
            grain.film_grain(ctx, 0, fg)
            got = [np.array(p) for p in grain.download_display(ctx)]
            ctx.close()
    finally:
        ol.oracle_set_gaussian_sequence(grain_stand_in())

    def expect(base, size, period, ncols, plain, once, twice):
        e = np.full(size, base + plain, int)
        if overlap:
            col = np.zeros(size[1], bool)
            row = np.zeros(size[0], bool)
            for k in range(period, size[1], period):
                col[k:k + ncols] = True
            for k in range(period, size[0], period):
                row[k:k + ncols] = True
            e[:, col] = base + once
            e[row, :] = base + once
            e[np.ix_(row, col)] = base + twice
        return e
    np.testing.assert_array_equal(got[0].astype(int), expect(100, (h, w), 32, 2, 20, 28, 38))
    np.testing.assert_array_equal(got[1].astype(int), expect(90, (h // 2, w // 2), 16, 1, 18, 25, 32))
    np.testing.assert_array_equal(got[2].astype(int), expect(160, (h // 2, w // 2), 16, 1, -32, -45, -63))


def grain_stand_in():
    """the stand-in sequence (include/easyav1_amd_synth.h) the other film grain tests run with"""
    from easyav1_amd import grain
    return grain.synth_gaussian_sequence()


# ---- CDEF at the edge of a frame whose blocks reach beyond it (ea1_frame_desc.pad): the frame is 128 x 64 with a TRUE coded size of
# 128 x 40 (three 8-row units of padding), chroma 64 x 20.  An impulse x + e on the LAST coded chroma row (19) of a flat plane x,
# direction 0 (flat luma): of the primary taps (-1, +1), (-2, +2) and their opposites only the two that point up are available
# (CdefAvailable is false beyond the coded frame, 7.15.2), of the secondary taps the four on the row and the two above ((-1, 0), (-2, 0)):
#     sum = (w0 + w1) * constrain(-e, pri) + (2 + 2 + 1 + 1 + 2 + 1) * constrain(-e, sec)
# An implementation that took the padding for picture would find x at (20, 20) and (21, 19) -- the inter copy repeats row 19 of the
# reference down there, and those columns hold x -- and add two more primary taps.
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("pri,sec,damping,e", [(4, 2, 4, 6), (8, 1, 5, 9), (12, 0, 6, 20)])
def test_cdef_takes_samples_beyond_the_coded_frame_for_unavailable(backend, pri, sec, damping, e):
    desc, blocks, txs, lr = copy_frame(3, 3, cdef_enable=1, cdef_damping=damping, cdef_y_strength=[0] * 8, cdef_uv_strength=[(pri << 2) | sec] * 8, pad=0x30)
    x = 100
    c = np.full((H // 2, W // 2), x, np.uint8)
    cy, cx = 19, 21
    c[cy, cx] = x + e
    ref = [np.full((H, W), 128, np.uint8), c, np.full((H // 2, W // 2), x, np.uint8)]
    got = run(backend, desc, blocks, txs, lr, ref, stages=ir.STAGE_RECON | ir.STAGE_CDEF)
    D = damping - 1
    cp, cs = constrain(-e, pri, D), constrain(-e, sec, D)
    w0, w1 = (4, 2) if pri % 2 == 0 else (3, 3)
    total = (w0 + w1) * cp + 9 * cs
    assert got[1][cy, cx] == x + e + ((8 + total - (1 if total < 0 else 0)) >> 4)
    assert got[1][cy - 1, cx + 1] == x + ((8 + w0 * constrain(e, pri, D)) >> 4)      # sees the impulse one step down-left of it
    assert got[1][cy, cx + 1] == x + ((8 + 2 * constrain(e, sec, D)) >> 4)           # ... and on its row, through a secondary tap
