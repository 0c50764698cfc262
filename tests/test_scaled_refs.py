"""Frames smaller than their context and scaled references (AV1 spec 7.11.3.3 / 7.11.3.4): the oracle against closed
forms on the CPU, the HIP path against the oracle with -m gpu."""
import numpy as np
import pytest

from easyav1_amd import ir
from tests import oracle_lib as ol

GPU = pytest.mark.gpu


def plain_inter_features():
    return ir.SYN_RECT | ir.SYN_TX_SPLIT | ir.SYN_ALL_TXTP | ir.SYN_SMOOTH | ir.SYN_DIRECTIONAL | ir.SYN_BILINEAR | ir.SYN_LR_MIX | ir.SYN_DIAGONAL_COEFS


def zero_motion_frame(W, H, b_log2=4, mv=(0, 0), ref=0):
    """an inter frame of (1 << b_log2)^2 blocks, one vector for all, nothing coded, no filters"""
    b = 1 << b_log2
    blocks, txs = [], []
    for y in range(0, H, b):
        for x in range(0, W, b):
            blk = np.zeros(1, ir.BLOCK_DTYPE)[0]
            blk["x4"], blk["y4"], blk["bwl"], blk["bhl"] = x // 4, y // 4, b_log2, b_log2
            blk["flags"] = ir.BLK_INTER | (ir.BLK_HAVE_LEFT if x else 0) | (ir.BLK_HAVE_TOP if y else 0)
            blk["ref"] = (ref, ref)
            blk["mv"] = (mv, (0, 0))
            blk["first_tx"] = len(txs)
            c = b_log2 - 1
            txs += [(x // 4, y // 4, 0, b_log2 | (b_log2 << 4), 0, 0, ir.NO_COEFS), (x // 8, y // 8, 1, c | (c << 4), 0, 0, ir.NO_COEFS), (x // 8, y // 8, 2, c | (c << 4), 0, 0, ir.NO_COEFS)]
            blk["n_tx"] = 3
            blocks.append(blk)
    d = ir.FrameDesc()
    d.width, d.height, d.is_inter, d.dst_slot = W, H, 1, 1
    d.cdef_damping = 3
    sbs = ((W + 63) // 64) * ((H + 63) // 64)
    return ir.ParsedFrame(d, np.array(blocks, ir.BLOCK_DTYPE), np.array(txs, ir.TXBLOCK_DTYPE), np.zeros(0, np.int16), np.zeros(0, np.uint8),
                          np.zeros(sbs, np.uint8), [np.zeros(0, ir.LR_UNIT_DTYPE)] * 3)


# ---------------------------------------------------------------------------------------------- closed forms (CPU)
def test_flat_references_stay_flat_at_any_scale():
    f = zero_motion_frame(128, 64, mv=(5, -11))
    for rw, rh in ((64, 32), (96, 48), (256, 128), (200, 64), (128, 120), (8, 8)):
        ref = [np.full((rh, rw), 77, np.uint8), np.full((rh // 2, rw // 2), 30, np.uint8), np.full((rh // 2, rw // 2), 200, np.uint8)]
        got = ol.oracle_decode_frame(f, [ref], ir.STAGE_RECON)["recon"]
        assert (got[0] == 77).all() and (got[1] == 30).all() and (got[2] == 200).all(), (rw, rh)


def test_two_to_one_reference_is_read_at_the_centres_of_sample_pairs():
    """reference twice as wide: xScale = 2 << 14, xStep = 2048; with zero motion sample c of the frame sits at reference
    position 2 c + 1/2 (the centre of the pair it stands for): a ramp ref[x] = x gives (2c + 2c + 1 + 1) >> 1 = 2c + 1 through
    the symmetric half-sample filter wherever the taps stay inside; rows are copied (height unscaled)"""
    W, H = 64, 32
    f = zero_motion_frame(W, H)
    yy, xx = np.mgrid[0:H, 0:2 * W]
    ref_y = (xx + (yy & 3)).astype(np.uint8)
    cy, cx = np.mgrid[0:H // 2, 0:W]
    ref_c = (2 * cx + 5).astype(np.uint8)
    got = ol.oracle_decode_frame(f, [[ref_y, ref_c, ref_c.copy()]], ir.STAGE_RECON)["recon"]
    c = np.arange(W)
    want = (2 * c + 1)[None, :] + (np.arange(H) & 3)[:, None]
    assert np.array_equal(got[0][:, 2:-2], want[:, 2:-2])
    cc = np.arange(W // 2)
    wc = ((2 * (2 * cc) + 5) + (2 * (2 * cc + 1) + 5) + 1) >> 1   # chroma ramp 2x + 5 between x = 2c and 2c + 1: 4c + 6
    assert np.array_equal(got[1][:, 2:-2], np.broadcast_to(wc, (H // 2, W // 2))[:, 2:-2])


def test_one_to_two_reference_alternates_between_quarter_positions():
    """reference half as wide: xStep = 512; frame sample c sits at reference position c / 2 - 1/4: a flat-in-x, stepped-in-y
    picture keeps its rows (the vertical scale is 1: rows are copied)"""
    W, H = 64, 32
    f = zero_motion_frame(W, H)
    ref_y = np.repeat((np.arange(H) * 5 % 251).astype(np.uint8)[:, None], W // 2, axis=1)
    ref_c = np.full((H // 2, W // 4), 90, np.uint8)
    got = ol.oracle_decode_frame(f, [[ref_y, ref_c, ref_c.copy()]], ir.STAGE_RECON)["recon"]
    assert np.array_equal(got[0], np.repeat(ref_y[:, :1], W, axis=1))


def cropped_two_to_one_case():
    """frame 64 x 66 visible (coded 64 x 72, crop 6 rows) from a reference of 128 x 132 visible (coded 128 x 136, crop 4 rows):
    yScale = ((132 << 14) + 33) / 66 = 2 << 14 EXACTLY on the true sizes (on the coded sizes it would be ((136 << 14) + 36) / 72 =
    30947); the reference is a vertical ramp ref[y][x] = y (chroma 2 y + 3)"""
    W, H, crop = 64, 72, 6 << 4
    f = zero_motion_frame(W, H, b_log2=3)
    f.desc.crop = crop
    yy, _ = np.mgrid[0:136, 0:128]
    cy, _ = np.mgrid[0:68, 0:64]
    ref = [yy.astype(np.uint8), (2 * cy + 3).astype(np.uint8), (2 * cy + 3).astype(np.uint8)]
    return f, ref, 4 << 4


def check_cropped_two_to_one(got):
    # luma row r of the frame sits at reference row 2 r + 1/2: (2r + 2r + 1 + 1) >> 1 = 2r + 1 while the 8 taps stay inside the
    # reference's VISIBLE rows 0 .. 131; chroma row r: ramp 2y + 3 at y = 2r + 1/2 -> 4r + 4
    r = np.arange(2, 62)
    assert np.array_equal(got[0][2:62, 5], 2 * r + 1), got[0][:, 5].tolist()
    rc = np.arange(2, 30)
    assert np.array_equal(got[1][2:30, 3], 4 * rc + 4), got[1][:, 3].tolist()


def test_scale_factors_use_the_true_sizes_not_the_coded_ones():
    f, ref, rcrop = cropped_two_to_one_case()
    got = ol.oracle_decode_frame(f, [ref], ir.STAGE_RECON, ref_crops=[rcrop] * 8)["recon"]
    check_cropped_two_to_one(got)


def test_same_coded_size_but_another_visible_size_is_a_scaled_reference():
    """a 64 x 64 reference without crop read by a 64 x 60-visible frame (coded 64 x 64 too): yScale = ((64 << 14) + 30) / 60 != 1 << 14,
    so a flat-in-x picture with distinct rows is NOT copied row for row"""
    f = zero_motion_frame(64, 64)
    f.desc.crop = 4 << 4
    rows = (np.arange(64) * 3).astype(np.uint8)
    ref = [np.repeat(rows[:, None], 64, axis=1), np.full((32, 32), 9, np.uint8), np.full((32, 32), 9, np.uint8)]
    same = ol.oracle_decode_frame(f, [ref], ir.STAGE_RECON, ref_crops=[4 << 4] * 8)["recon"]
    assert np.array_equal(same[0][:60], ref[0][:60])             # reference with the frame's own true size: plain copy of the visible rows
    assert (same[0][60:] == ref[0][59]).all()                    # beyond them: the last visible row (lastY clamp)
    scaled = ol.oracle_decode_frame(f, [ref], ir.STAGE_RECON, ref_crops=[0] * 8)["recon"]
    assert not np.array_equal(scaled[0][:60], ref[0][:60])
    # row r reads reference position (r + 1/2) * 64 / 60 - 1/2: row 30 of the frame -> 32.03: between rows 32 and 33 of the ramp 3y
    assert 96 <= scaled[0][30, 8] <= 99


# ---------------------------------------------------------------------------------------------- GPU vs oracle
def decode_gpu(ctx, f):
    job = ctx.make_job(f)
    ctx.submit(job)
    out = ctx.download(f.desc.dst_slot)
    job.close()
    return out


@GPU
@pytest.mark.parametrize("w,h", [(128, 96), (72, 40), (256, 192)])
def test_frames_smaller_than_the_context(w, h):
    """a 256x192 context decodes a whole GOP of w x h frames (every tool): identical to the oracle at that size"""
    from easyav1_amd import video
    from tests.test_video_gpu import assert_planes_equal
    ctx = video.ReconContext(256, 192)
    ctx.set_warp_filter(ir.synth_warp_filter())
    ol.oracle_set_warp_filter(ir.synth_warp_filter())
    p = ir.synth_params(w, h, 1, seed=w + h, features=0x7FFF | ir.SYN_WARP | ir.SYN_DIAGONAL_COEFS)
    refs = [None] * 8
    for k in range(4):
        f = ir.synth_frame(p, k)
        want = ol.oracle_decode_frame(f, refs)["final"]
        assert_planes_equal(want, decode_gpu(ctx, f), f"{w}x{h} frame {k}")
        refs[f.desc.dst_slot] = want
    ctx.close()


@GPU
@pytest.mark.parametrize("rw,rh", [(128, 96), (192, 144), (256, 192), (64, 48), (96, 192), (256, 96), (16, 16)])
@pytest.mark.parametrize("fw,fh", [(128, 96), (256, 192)])
def test_scaled_references_match_the_oracle(rw, rh, fw, fh):
    """an inter frame of fw x fh (translated blocks of every shape from 4x4 up, sub-pel vectors of +-64 samples, single
    reference and every compound kind, OBMC, inter-intra) predicts from random pictures of rw x rh in its slots"""
    if 2 * fw < rw or 2 * fh < rh or fw > 16 * rw or fh > 16 * rh:
        pytest.skip("outside the limits of scaled references")
    from easyav1_amd import video
    from tests.test_video_gpu import assert_planes_equal
    rng = np.random.default_rng(rw * 7 + rh + fw)
    ctx = video.ReconContext(256, 192)
    p = ir.synth_params(fw, fh, 2, seed=rw + rh, features=plain_inter_features() | ir.SYN_COMPOUND | ir.SYN_OBMC | ir.SYN_INTERINTRA | ir.SYN_SUB8)
    for k in (1, 2):   # frame 1 reads slot 0, frame 2 slots 1 and 0 (compound blocks: one prediction from each)
        f = ir.synth_frame(p, k)
        if k == 2:
            assert (f.blocks["flags"] & ir.BLK_COMPOUND).any()
        refs = [None] * 8
        for slot in range(8):
            if f.blocks["flags"].any() and slot < 3:
                sw, sh = (rw, rh) if slot != 1 else (fw, fh)     # slot 1 keeps the frame's size: scaled and unscaled references in one frame
                pic = [rng.integers(0, 256, (sh, sw), dtype=np.uint8), rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8), rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8)]
                refs[slot] = pic
                ctx.set_ref(slot, pic)
        want = ol.oracle_decode_frame(f, refs)["final"]
        assert_planes_equal(want, decode_gpu(ctx, f), f"ref {rw}x{rh} frame {fw}x{fh} #{k}")
    ctx.close()


@GPU
def test_warped_motion_with_a_scaled_reference_is_refused():
    from easyav1_amd import video, Ea1Error
    ctx = video.ReconContext(128, 96)
    ctx.set_warp_filter(ir.synth_warp_filter())
    p = ir.synth_params(128, 96, 1, seed=4, features=ir.SYN_WARP)
    f = ir.synth_frame(p, 1)
    assert (f.blocks["flags"] & ir.BLK_WARP).any()
    small = [np.zeros((48, 64), np.uint8), np.zeros((24, 32), np.uint8), np.zeros((24, 32), np.uint8)]
    for slot in range(8):
        ctx.set_ref(slot, small)
    job = ctx.make_job(f)
    with pytest.raises(Ea1Error):
        ctx.submit(job)
    big = [np.zeros((96, 128), np.uint8), np.zeros((48, 64), np.uint8), np.zeros((48, 64), np.uint8)]
    for slot in range(8):
        ctx.set_ref(slot, big)
    ctx.submit(job)   # the same job with references of its own size decodes
    job.close()
    ctx.close()


@GPU
def test_scaled_references_with_crop_on_both_sides():
    """the closed form above on the GPU, then random pictures: frame 128 x 90 visible (coded 96, crop 6) from 256 x 180 visible
    (coded 184, crop 4), against the oracle"""
    from easyav1_amd import video
    from tests.test_video_gpu import assert_planes_equal
    f, ref, rcrop = cropped_two_to_one_case()
    ctx = video.ReconContext(256, 192)
    ctx.set_ref(0, ref, crop=rcrop)
    job = ctx.make_job(f)
    ctx.submit(job, ir.STAGE_RECON)
    got = ctx.download(1)
    job.close()
    check_cropped_two_to_one(got)
    rng = np.random.default_rng(5)
    p = ir.synth_params(128, 96, 2, seed=77, features=plain_inter_features() | ir.SYN_COMPOUND | ir.SYN_OBMC | ir.SYN_INTERINTRA | ir.SYN_SUB8)
    for k in (1, 2):
        fr = ir.synth_frame(p, k)
        fr.desc.crop = 6 << 4
        refs, crops = [None] * 8, [0] * 8
        for slot in range(3):
            sw, sh, cr = (256, 184, 4 << 4) if slot != 1 else (128, 96, 6 << 4)
            pic = [rng.integers(0, 256, (sh, sw), dtype=np.uint8), rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8), rng.integers(0, 256, (sh // 2, sw // 2), dtype=np.uint8)]
            refs[slot], crops[slot] = pic, cr
            ctx.set_ref(slot, pic, crop=cr)
        want = ol.oracle_decode_frame(fr, refs, ref_crops=crops)["final"]
        assert_planes_equal(want, decode_gpu(ctx, fr), f"cropped scaled references, frame {k}")
    ctx.close()
