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
