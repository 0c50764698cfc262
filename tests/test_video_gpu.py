"""GPU parity of the AV1 reconstruction path: HIP kernels (through the C ABI) vs the CPU oracle,
bit-exact on every stage (recon, deblock, CDEF, loop restoration)."""
import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

# synthetic-stream feature bits (easyav1_amd.ir.SYN_*)
(SYN_RECT, SYN_TX_SPLIT, SYN_ALL_TXTP, SYN_SMOOTH, SYN_DIRECTIONAL, SYN_CFL, SYN_FILTER_INTRA, SYN_PALETTE,
 SYN_COMPOUND, SYN_OBMC, SYN_INTERINTRA, SYN_BILINEAR, SYN_LF_DELTA, SYN_SUB8, SYN_LR_MIX) = (1 << i for i in range(15))

SYN_WARP = 0x10000
SYN_INTRABC = 0x20000

ITX_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16),
             (32, 64), (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def kinds_for(n):
    k = [0]
    if n <= 16:
        k += [1, 2]
    if n <= 32:
        k += [3]
    return k


def assert_planes_equal(want, got, what):
    for pl in range(3):
        if not np.array_equal(want[pl], got[pl]):
            bad = np.argwhere(want[pl] != got[pl])
            y, x = bad[0]
            raise AssertionError(f"{what}: plane {pl}: {len(bad)} differing pixels, first at (x={x}, y={y}): "
                                 f"want {want[pl][y, x]} got {got[pl][y, x]}; rows {sorted(set(bad[:, 0].tolist()))[:8]} "
                                 f"cols {sorted(set(bad[:, 1].tolist()))[:8]}")


@pytest.mark.parametrize("w,h", ITX_SIZES)
def test_itx_add_all_sizes(torch_cuda, w, h):
    torch = torch_cuda
    from easyav1_amd import video
    rng = np.random.default_rng(w * 100 + h)
    types = [(c << 2) | r for c in kinds_for(h) for r in kinds_for(w)]
    sw, sh = min(w, 32), min(h, 32)
    count = len(types) * 3
    bw = 4  # blocks per row in the test plane
    rows = (count + bw - 1) // bw
    plane = rng.integers(0, 256, (rows * h, bw * w), dtype=np.uint8)
    coefs = np.zeros((count, sw * sh), np.int16)
    pos = np.zeros((count, 2), np.uint32)
    txtp = np.zeros(count, np.uint8)
    for i in range(count):
        txtp[i] = types[i % len(types)]
        pos[i] = ((i % bw) * w, (i // bw) * h)
        variant = i // len(types)
        if variant == 0:    # sparse low-frequency, like real streams
            n = max(1, sw * sh // 8)
            idx = rng.choice(sw * sh, n, replace=False)
            coefs[i, idx] = np.clip(np.round(rng.laplace(scale=40, size=n)), -32768, 32767)
            coefs[i, 0] = rng.integers(-2000, 2000)
        elif variant == 1:  # DC only
            coefs[i, 0] = rng.integers(-4000, 4000)
        else:               # dense, large: exercises the int16 clamps
            coefs[i] = rng.integers(-3000, 3000, sw * sh)
    want = plane.copy()
    for i in range(count):
        x, y = int(pos[i, 0]), int(pos[i, 1])
        want[y:y + h, x:x + w] = ol.oracle_itx_add(want[y:y + h, x:x + w], w, h, int(txtp[i]), coefs[i])
    d_plane = torch.from_numpy(plane).cuda()
    video.dsp_itx_add(d_plane, w, h, torch.from_numpy(coefs).cuda(), pos, txtp)
    got = d_plane.cpu().numpy()
    if not np.array_equal(want, got):
        bad = [i for i in range(count) if not np.array_equal(
            want[pos[i, 1]:pos[i, 1] + h, pos[i, 0]:pos[i, 0] + w], got[pos[i, 1]:pos[i, 1] + h, pos[i, 0]:pos[i, 0] + w])]
        raise AssertionError(f"{w}x{h}: blocks {bad[:10]} differ (txtp {[int(txtp[i]) for i in bad[:10]]})")


def decode_chain(torch, W, H, nframes, tile_cols=1, seed=1, check_stages=True, warp_table=None, **overrides):
    """warp_table: None = the table the library and the oracle start with (include/ea1_av1_tables.inc); else a 193 x 8 table that goes to
    the library (ea1_recon_set_warp_filter) and to the checker alike"""
    from easyav1_amd import ir, video
    p = ir.synth_params(W, H, tile_cols, seed=seed, **overrides)
    ctx = video.ReconContext(W, H)
    ol.oracle_set_warp_filter(warp_table)
    if warp_table is not None:
        ctx.set_warp_filter(warp_table)
    refs = [None] * 8
    for fi in range(nframes):
        f = ir.synth_frame(p, fi)
        want = ol.oracle_decode_frame(f, refs)
        job = ctx.make_job(f)
        slot = f.desc.dst_slot
        if check_stages:
            for stages, name in ((1, "recon"), (3, "deblock"), (7, "cdef")):
                ctx.submit(job, stages)
                assert_planes_equal(want[name], ctx.download(slot), f"frame {fi} stage {name}")
        ctx.submit(job)  # all stages: leaves the final picture in the slot for the next frames
        got = ctx.download(slot)
        assert_planes_equal(want["final"], got, f"frame {fi} final")
        refs[slot] = want["final"]
        job.close()
    ctx.close()


def test_small_gop_every_stage(torch_cuda):
    decode_chain(torch_cuda, 192, 128, 4, seed=3)


def test_c1_plumbing_640x360(torch_cuda):
    decode_chain(torch_cuda, 640, 360, 3, seed=1)


def test_smooth_modes_and_tiles(torch_cuda):
    decode_chain(torch_cuda, 448, 256, 3, tile_cols=4, seed=5, features=SYN_SMOOTH)


def test_rect_blocks_and_transform_grids(torch_cuda):
    """all partition shapes (2:1, 4:1, T shapes), transform sizes below the block size (4x4 ..),
    all 1-D transform kind pairs, bilinear filter: every stage against the oracle"""
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=31, features=SYN_RECT)
    decode_chain(torch_cuda, 320, 192, 3, seed=32, features=SYN_TX_SPLIT | SYN_ALL_TXTP)
    decode_chain(torch_cuda, 448, 320, 3, tile_cols=2, seed=33,
                 features=SYN_RECT | SYN_TX_SPLIT | SYN_ALL_TXTP | SYN_SMOOTH | SYN_BILINEAR)


@pytest.mark.parametrize("features", [SYN_DIRECTIONAL, SYN_CFL, SYN_FILTER_INTRA, SYN_PALETTE,
                                      SYN_DIRECTIONAL | SYN_CFL | SYN_FILTER_INTRA | SYN_PALETTE | SYN_SMOOTH | SYN_RECT | SYN_TX_SPLIT])
def test_intra_predictors(torch_cuda, features):
    """directional modes (angle deltas, edge filter, upsampling), chroma from luma, recursive
    filter intra, palette -- alone and together on all block shapes / transform grids; gop=2 makes
    every other frame a key frame"""
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=40 + (features & 63), gop=2, features=features, check_stages=False)
    decode_chain(torch_cuda, 192, 128, 2, seed=41, gop=1, features=features, check_stages=True)


@pytest.mark.parametrize("features", [SYN_COMPOUND, SYN_OBMC, SYN_INTERINTRA,
                                      SYN_COMPOUND | SYN_OBMC | SYN_INTERINTRA | SYN_BILINEAR | SYN_RECT | SYN_TX_SPLIT, 0x7FFF])
def test_inter_tools(torch_cuda, features):
    """distance-weighted / wedge / difference-weighted compound, overlapped block motion
    compensation, inter-intra (smooth and wedge), alone, together and with every other tool"""
    decode_chain(torch_cuda, 320, 192, 4, tile_cols=2, seed=60 + (features & 63), features=features, check_stages=False)
    decode_chain(torch_cuda, 192, 128, 3, seed=61, features=features, check_stages=True)


@pytest.mark.parametrize("features", [SYN_WARP, SYN_WARP | SYN_RECT | SYN_COMPOUND | SYN_OBMC | SYN_INTERINTRA, SYN_WARP | 0x7FFF])
def test_warped_motion(torch_cuda, features):
    """warped predictions (AV1 spec 7.11.3.5) on every block shape from 8x8 up: single and compound
    (either or both predictions warped), matrices that fail the shear test (plain translation
    instead), translated chroma of blocks whose chroma is smaller than 8x8, OBMC / inter-intra on
    top; positions far outside the picture (edge clamping)"""
    decode_chain(torch_cuda, 320, 192, 4, tile_cols=2, seed=90 + (features & 63), features=features, check_stages=False)
    decode_chain(torch_cuda, 192, 128, 3, seed=91, features=features, check_stages=True)
    if features == SYN_WARP:   # a caller's own table instead of the built-in one (here: the generator's stand-in, another table altogether)
        from easyav1_amd import ir
        decode_chain(torch_cuda, 192, 128, 3, seed=92, features=features, check_stages=False, warp_table=ir.synth_warp_filter())


SYN_DIAGONAL_COEFS = 0x100000


@pytest.mark.parametrize("features", [SYN_DIAGONAL_COEFS, SYN_DIAGONAL_COEFS | 0x7FFF, SYN_DIAGONAL_COEFS | 0x7FFF | SYN_WARP])
def test_diagonal_coefficient_layout(torch_cuda, features):
    """EA1_COEFS_DIAGONAL (packed anti-diagonals, expanded to the corner layout on the device at hand-over) with every
    transform size and tool, tiles, inter-intra blocks (zero-fill items beside expanded ones), lossless 4x4 blocks"""
    decode_chain(torch_cuda, 320, 192, 4, tile_cols=2, seed=60 + (features & 63), features=features, check_stages=False)
    decode_chain(torch_cuda, 136, 72, 3, seed=61, features=features, check_stages=True)
    if features == SYN_DIAGONAL_COEFS:
        decode_chain(torch_cuda, 128, 64, 2, seed=62, features=features | 0x8000 | SYN_RECT | SYN_SUB8, check_stages=False)


def test_warp_works_out_of_the_box_and_an_own_table_is_validated(torch_cuda):
    from easyav1_amd import ir, video
    from easyav1_amd._lib import Ea1Error
    p = ir.synth_params(192, 128, 1, seed=5, features=SYN_WARP)
    ctx = video.ReconContext(192, 128)
    f = ir.synth_frame(p, 1)
    assert (f.blocks["flags"] & ir.BLK_WARP).any()
    ctx.make_job(f).close()           # a caller that supplies nothing: the library's own Warped_Filters table
    bad = ir.synth_warp_filter()
    bad[17, 3] += 1
    with pytest.raises(Ea1Error, match="sums to"):
        ctx.set_warp_filter(bad)
    ctx.set_warp_filter(ir.synth_warp_filter())
    ctx.make_job(f).close()
    ctx.close()


@pytest.mark.parametrize("features", [SYN_INTRABC, SYN_INTRABC | SYN_RECT | SYN_TX_SPLIT | SYN_ALL_TXTP, SYN_INTRABC | 0x7FFF])
def test_intra_block_copy(torch_cuda, features):
    """intra block copy inside the intra wavefront: whole-sample luma copies, half-sample chroma,
    copies of copies (chains of dependency levels), next to every other intra tool; gop=2 makes every
    other frame an intra frame (without in-loop filters) that the inter frames then reference"""
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=110 + (features & 63), gop=2, features=features, check_stages=False)
    decode_chain(torch_cuda, 192, 128, 2, seed=111, gop=1, features=features, check_stages=True)
    decode_chain(torch_cuda, 640, 384, 1, seed=112, gop=1, features=features, check_stages=False)


def test_intra_block_copy_validation(torch_cuda):
    from easyav1_amd import ir, video
    from easyav1_amd._lib import Ea1Error
    W, H = 192, 128
    p = ir.synth_params(W, H, 1, seed=7, features=SYN_INTRABC)
    f = ir.synth_frame(p, 0)
    idx = np.flatnonzero(f.blocks["mode_y"] == ir.INTRABC_PRED)
    assert len(idx) > 3
    ctx = video.ReconContext(W, H)
    ctx.make_job(f).close()

    def broken(edit):
        g = ir.ParsedFrame(f.desc.copy(), f.blocks.copy(), f.txs.copy(), f.coefs.copy(), f.aux.copy(), f.cdef_idx.copy(), [u.copy() for u in f.lr])
        edit(g)
        return g
    cases = {
        "not decoded yet": lambda g: g.blocks["mv"].__setitem__((idx[0], 0), [8 * 8, 8 * 8]),      # forward / downward
        "block vector": lambda g: g.blocks["mv"].__setitem__((idx[1], 0), [g.blocks["mv"][idx[1], 0, 0] + 4, g.blocks["mv"][idx[1], 0, 1]]),
        "leaves the frame": lambda g: g.blocks["mv"].__setitem__((idx[2], 0), [-8 * 200, 0]),
        "mode_y == mode_uv": lambda g: g.blocks["mode_uv"].__setitem__(idx[0], 0),
        "in-loop filters": lambda g: setattr(g.desc, "cdef_enable", 1),
    }
    for text, edit in cases.items():
        with pytest.raises(Ea1Error, match=text):
            ctx.make_job(broken(edit))
    ctx.close()


def test_blocks_with_4_sample_sides(torch_cuda):
    """4x4, 4x8, 8x4, 4x16, 16x4 luma blocks: shared chroma (owned by the last block of the 8x8
    area), piecewise chroma prediction of all-inter areas, 4-sample deblock grid"""
    decode_chain(torch_cuda, 192, 128, 4, seed=81, features=SYN_SUB8)
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=82, features=SYN_SUB8 | SYN_RECT | SYN_TX_SPLIT | SYN_LF_DELTA)
    decode_chain(torch_cuda, 256, 192, 3, seed=83, gop=2, check_stages=False,
                 features=SYN_SUB8 | SYN_RECT | SYN_DIRECTIONAL | SYN_CFL | SYN_FILTER_INTRA | SYN_SMOOTH)


def test_restoration_types_per_unit(torch_cuda):
    """none / Wiener / self-guided drawn per unit in all three planes, all 16 self-guided parameter
    sets (radius 2 only, radius 1 only, both), unit sizes 64 / 128 / 256"""
    decode_chain(torch_cuda, 448, 320, 2, seed=91, features=SYN_LR_MIX, check_stages=False)
    decode_chain(torch_cuda, 640, 384, 2, seed=92, features=SYN_LR_MIX, lr_unit_log2=7, check_stages=False)
    decode_chain(torch_cuda, 768, 512, 2, seed=93, features=SYN_LR_MIX, lr_unit_log2=8, check_stages=False)


def test_tile_rows_and_lossless_frames(torch_cuda):
    """tile rows (no top edge across a tile-row boundary, above-right / below-left limited by the tile)
    and lossless frames (4x4 Walsh-Hadamard everywhere, in-loop filters off)"""
    decode_chain(torch_cuda, 320, 256, 3, tile_cols=2, seed=111, tile_rows=2, features=0x7FFF, check_stages=False)
    decode_chain(torch_cuda, 448, 320, 2, tile_cols=4, seed=112, tile_rows=4, features=SYN_RECT | SYN_DIRECTIONAL | SYN_SUB8)
    decode_chain(torch_cuda, 256, 192, 3, seed=113, features=0x8000 | SYN_RECT | SYN_SUB8 | SYN_DIRECTIONAL | SYN_COMPOUND)


def test_per_block_deblock_levels(torch_cuda):
    """delta_lf-style per-block levels, including level 0 blocks that inherit across the edge"""
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=71, features=SYN_LF_DELTA)
    decode_chain(torch_cuda, 256, 192, 3, seed=72, features=SYN_LF_DELTA | SYN_RECT | SYN_TX_SPLIT)


def test_wht_lossless_4x4(torch_cuda):
    torch = torch_cuda
    from easyav1_amd import video
    rng = np.random.default_rng(5)
    count = 64
    plane = rng.integers(0, 256, (4 * 8, 4 * 8), dtype=np.uint8)
    coefs = rng.integers(-400, 400, (count, 16)).astype(np.int16)
    pos = np.array([((i % 8) * 4, (i // 8) * 4) for i in range(count)], np.uint32)
    txtp = np.full(count, 16, np.uint8)
    want = plane.copy()
    for i in range(count):
        x, y = int(pos[i, 0]), int(pos[i, 1])
        want[y:y + 4, x:x + 4] = ol.oracle_itx_add(want[y:y + 4, x:x + 4], 4, 4, 16, coefs[i])
    d_plane = torch.from_numpy(plane).cuda()
    video.dsp_itx_add(d_plane, 4, 4, torch.from_numpy(coefs).cuda(), pos, txtp)
    assert np.array_equal(want, d_plane.cpu().numpy())


def test_stage_switches(torch_cuda):
    # frames with some post-filters disabled in the frame header
    decode_chain(torch_cuda, 256, 192, 2, seed=7, enable_cdef=0)
    decode_chain(torch_cuda, 256, 192, 2, seed=8, enable_lr=0, enable_deblock=0)


def test_luma_lr_unit_sizes(torch_cuda):
    decode_chain(torch_cuda, 384, 320, 2, seed=9, lr_unit_log2=7, check_stages=False)
    decode_chain(torch_cuda, 640, 360, 2, seed=10, lr_unit_log2=8, check_stages=False)


def test_full_size_1080p_properties(torch_cuda):
    """C2-sized frames (1920x1080): determinism (two contexts agree bit for bit) and a full
    oracle comparison of ONE key + ONE inter frame (the oracle needs a few seconds per frame)."""
    torch = torch_cuda
    from easyav1_amd import ir, video
    W, H = 1920, 1080
    p = ir.synth_params(W, H, 1, seed=2)
    a, b = video.ReconContext(W, H), video.ReconContext(W, H)
    refs = [None] * 8
    for fi in range(2):
        f = ir.synth_frame(p, fi)
        ja, jb = a.make_job(f), b.make_job(f)
        a.submit(ja)
        b.submit(jb)
        pa, pb = a.download(f.desc.dst_slot), b.download(f.desc.dst_slot)
        assert_planes_equal(pa, pb, f"determinism frame {fi}")
        want = ol.oracle_decode_frame(f, refs)
        assert_planes_equal(want["final"], pa, f"1080p frame {fi}")
        refs[f.desc.dst_slot] = want["final"]
        st = ja.stats()
        assert st["algorithmic_bytes"] > 0 and (fi == 0 or st["refs_per_pixel"] > 0.5)
    a.close()
    b.close()


def test_whole_gop_1080p_batched(torch_cuda):
    """BASELINE configs[1] at full length: a whole 60-frame GOP of two chains decoded through
    submit_batch, every frame of every chain against the oracle (a single wrong sample would spread
    through the reference slots)"""
    import threading
    from easyav1_amd import ir, video
    W, H, G, NCH = 1920, 1080, 60, 2
    ctx = video.ReconContext(W, H, nchains=NCH)
    params = [ir.synth_params(W, H, 1, seed=300 + c) for c in range(NCH)]
    refs = [[None] * 8 for _ in range(NCH)]
    for fi in range(G):
        frames = [ir.synth_frame(params[c], fi) for c in range(NCH)]
        jobs = [ctx.make_job(f) for f in frames]
        ctx.submit_batch(jobs)
        want = [None] * NCH

        def work(c):
            want[c] = ol.oracle_decode_frame(frames[c], refs[c])["final"]
        th = [threading.Thread(target=work, args=(c,)) for c in range(NCH)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in range(NCH):
            assert_planes_equal(want[c], ctx.download(frames[c].desc.dst_slot, chain=c), f"chain {c} frame {fi}")
            refs[c][frames[c].desc.dst_slot] = want[c]
        for j in jobs:
            j.close()
    ctx.close()


def test_full_size_4k_tiles(torch_cuda):
    """C3/C4-sized frames (3840x2160, 4 and 8 tile columns): key + inter frame against the oracle."""
    from easyav1_amd import ir, video
    W, H = 3840, 2160
    for tiles, seed in ((4, 21), (8, 22)):
        p = ir.synth_params(W, H, tiles, seed=seed)
        ctx = video.ReconContext(W, H)
        refs = [None] * 8
        for fi in range(2):
            f = ir.synth_frame(p, fi)
            job = ctx.make_job(f)
            ctx.submit(job)
            want = ol.oracle_decode_frame(f, refs)
            assert_planes_equal(want["final"], ctx.download(f.desc.dst_slot), f"4K {tiles} tiles frame {fi}")
            refs[f.desc.dst_slot] = want["final"]
            job.close()
        ctx.close()


def test_batched_chains_match_oracle(torch_cuda):
    """Five chains with different seeds, GOP lengths and (partly) disabled filters decoded through
    submit_batch (one launch serves all chains) must each equal the oracle; chains of different
    length leave idle (None) entries."""
    from easyav1_amd import ir, video
    W, H = 256, 192
    cfgs = [dict(seed=11), dict(seed=12, features=SYN_SMOOTH | SYN_RECT | SYN_TX_SPLIT), dict(seed=13, enable_cdef=0), dict(seed=14, enable_lr=0),
            dict(seed=15, enable_deblock=0, gop=2)]
    nfr = [4, 3, 4, 2, 4]
    ctx = video.ReconContext(W, H, nchains=len(cfgs))
    params = [ir.synth_params(W, H, 1, **c) for c in cfgs]
    refs = [[None] * 8 for _ in cfgs]
    for fi in range(max(nfr)):
        frames = [ir.synth_frame(params[c], fi) if fi < nfr[c] else None for c in range(len(cfgs))]
        jobs = [ctx.make_job(f) if f is not None else None for f in frames]
        ctx.submit_batch(jobs)
        for c, f in enumerate(frames):
            if f is None:
                continue
            want = ol.oracle_decode_frame(f, refs[c])
            assert_planes_equal(want["final"], ctx.download(f.desc.dst_slot, chain=c), f"chain {c} frame {fi}")
            refs[c][f.desc.dst_slot] = want["final"]
        for j in jobs:
            if j is not None:
                j.close()
    ctx.close()


def test_job_validation(torch_cuda):
    from easyav1_amd import ir, video, Ea1Error
    p = ir.synth_params(128, 128, 1, seed=1)
    f = ir.synth_frame(p, 0)
    ctx = video.ReconContext(128, 128)
    bad = ir.ParsedFrame(f.desc.copy(), f.blocks.copy(), f.txs, f.coefs, f.aux, f.cdef_idx, f.lr)
    bad.blocks["x4"][0] = 1  # misaligned
    with pytest.raises(Ea1Error):
        ctx.make_job(bad)
    bad2 = ir.ParsedFrame(f.desc.copy(), f.blocks[:-1].copy(), f.txs, f.coefs, f.aux, f.cdef_idx, f.lr)  # hole in the frame
    with pytest.raises(Ea1Error):
        ctx.make_job(bad2)
    bad3 = ir.ParsedFrame(f.desc.copy(), f.blocks.copy(), f.txs.copy(), f.coefs, f.aux, f.cdef_idx, f.lr)
    bad3.txs["size"][0] = 0x22 if bad3.txs["size"][0] != 0x22 else 0x33   # no longer tiles its block
    with pytest.raises(Ea1Error):
        ctx.make_job(bad3)
    with pytest.raises(Ea1Error):
        video.ReconContext(100, 64)
    ctx.close()


@pytest.mark.parametrize("w,h", [(8, 8), (16, 8), (8, 64), (64, 8), (24, 40), (72, 72), (136, 8), (200, 120)])
def test_tiny_and_ragged_frame_sizes(torch_cuda, w, h):
    """frames smaller than one superblock / one filter tile / one restoration unit, and sizes that are
    not multiples of 64: every tool, every stage"""
    decode_chain(torch_cuda, w, h, 3, seed=100 + w + h, features=0x7FFF, check_stages=True)
    decode_chain(torch_cuda, w, h, 2, seed=200 + w + h, features=0, check_stages=False)


def test_concurrent_hand_over_from_threads(torch_cuda):
    """ea1_job_create is called from eight host threads at once (a real shim parses and hands over
    frames of different chains on different threads); the frames then decode bit-exactly"""
    import threading
    from easyav1_amd import ir, video
    W, H, NCH = 256, 192, 8
    ctx = video.ReconContext(W, H, nchains=NCH)
    params = [ir.synth_params(W, H, 1, seed=400 + c, features=0x7FFF) for c in range(NCH)]
    refs = [[None] * 8 for _ in range(NCH)]
    for fi in range(3):
        frames = [ir.synth_frame(params[c], fi) for c in range(NCH)]
        jobs, errs = [None] * NCH, []

        def work(c):
            try:
                jobs[c] = ctx.make_job(frames[c])
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(c,)) for c in range(NCH)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        ctx.submit_batch(jobs)
        for c in range(NCH):
            want = ol.oracle_decode_frame(frames[c], refs[c])["final"]
            assert_planes_equal(want, ctx.download(frames[c].desc.dst_slot, chain=c), f"chain {c} frame {fi}")
            refs[c][frames[c].desc.dst_slot] = want
        for j in jobs:
            j.close()
    ctx.close()


def test_fuzzed_jobs_are_rejected_or_decoded(torch_cuda):
    """Random single-field corruptions of a valid parsed frame: ea1_job_create must either reject the
    frame (Ea1Error) or accept one the kernels can run safely -- the GPU must neither fault nor hang,
    and a valid frame must still decode bit-exactly afterwards."""
    import os
    from easyav1_amd import ir, video, Ea1Error
    W, H = 192, 128
    rng = np.random.default_rng(1234)
    ctx = video.ReconContext(W, H)
    ctx.set_warp_filter(ir.synth_warp_filter())
    p = ir.synth_params(W, H, 1, seed=5, features=0x7FFF | SYN_WARP | int(os.environ.get("EA1_FUZZ_LAYOUT", "0"), 0))
    key = ir.synth_frame(p, 0)
    job = ctx.make_job(key)
    ctx.submit(job)                       # slot 0 now holds a reference for the inter frame
    job.close()
    f = ir.synth_frame(p, 1)
    rejected = accepted = 0
    for trial in range(int(os.environ.get("EA1_FUZZ_TRIALS", "300"))):   # a long one-off run: EA1_FUZZ_TRIALS=6000
        blocks, txs, coefs, aux = f.blocks.copy(), f.txs.copy(), f.coefs.copy(), f.aux.copy()
        cdef_idx = f.cdef_idx.copy()
        lr = [u.copy() for u in f.lr]
        desc = f.desc.copy()
        kind = trial % 6
        if kind == 0:      # a random byte of a random block record
            raw = blocks.view(np.uint8).reshape(len(blocks), -1)
            raw[rng.integers(len(blocks)), rng.integers(raw.shape[1])] = rng.integers(256)
        elif kind == 1:    # a random byte of a transform-block record
            raw = txs.view(np.uint8).reshape(len(txs), -1)
            raw[rng.integers(len(txs)), rng.integers(raw.shape[1])] = rng.integers(256)
        elif kind == 2:    # extreme coefficients
            coefs[rng.integers(len(coefs), size=64)] = rng.choice([-32768, 32767], 64)
        elif kind == 3:    # palette / index bytes, warp matrices
            if len(aux):
                aux[rng.integers(len(aux), size=16)] = rng.integers(256, size=16)
        elif kind == 4:    # filter side data
            cdef_idx[rng.integers(len(cdef_idx))] = rng.integers(256)
            for u in lr:
                if len(u):
                    u.view(np.uint8)[rng.integers(u.nbytes)] = rng.integers(256)
        else:              # frame header bytes (not the sizes / counts)
            import ctypes
            raw = (ctypes.c_uint8 * ctypes.sizeof(desc)).from_address(ctypes.addressof(desc))
            raw[8 + int(rng.integers(36))] = int(rng.integers(256))
        bad = ir.ParsedFrame(desc, blocks, txs, coefs, aux, cdef_idx, lr)
        try:
            j = ctx.make_job(bad)
        except Ea1Error:
            rejected += 1
            continue
        accepted += 1
        try:
            ctx.submit(j)
        except Ea1Error:      # e.g. a corrupted reference index that points at a slot without a picture
            j.close()
            continue
        ctx.download(bad.desc.dst_slot if bad.desc.dst_slot < 8 else 0)   # synchronises: a fault would surface here
        j.close()
    assert rejected > 20 and accepted > 20, (rejected, accepted)
    # the context is still healthy: the valid frames decode bit-exactly
    ctx.close()
    decode_chain(torch_cuda, W, H, 2, seed=5, features=0x7FFF | SYN_WARP, check_stages=False)


# ---------------------------------------------------------------- film grain
def _grain_ctx(W, H, nchains=1):
    from easyav1_amd import grain, video
    ctx = video.ReconContext(W, H, nchains=nchains)
    seq = grain.synth_gaussian_sequence()
    grain.set_gaussian_sequence(ctx, seq)
    ol.oracle_set_gaussian_sequence(seq)
    return ctx, grain


@pytest.mark.parametrize("w,h", [(192, 128), (200, 120), (72, 40), (32, 32), (8, 8), (640, 360)])
def test_film_grain_matches_oracle(torch_cuda, w, h):
    """random legal film_grain_params (all lags, shifts, point counts, overlap on / off, chroma scaled
    from luma or by its own points, restricted range) on pictures whose size is / is not a multiple of
    the 32-sample block: GPU closed form vs the oracle's literal noise-stripe construction"""
    from easyav1_amd import ir
    ctx, grain = _grain_ctx(w, h)
    rng = np.random.default_rng(w * 1000 + h)
    for trial in range(12):
        pic = ir.synth_picture(w, h, 100 + trial)
        if trial % 4 == 3:   # extremes: saturated planes
            pic = [np.full_like(pic[0], 255 * (trial & 4 == 4)), np.full_like(pic[1], 255), np.zeros_like(pic[2])]
        ctx.set_ref(trial % 8, pic)
        fixed = {}
        if trial == 0:
            fixed = dict(overlap_flag=1, ar_coeff_lag=3, chroma_scaling_from_luma=0)
        if trial == 1:
            fixed = dict(overlap_flag=1, num_y_points=0)
        fg = grain.random_params(rng, **fixed)
        grain.film_grain(ctx, trial % 8, fg)
        got = grain.download_display(ctx)
        want = ol.oracle_film_grain(fg, pic)
        assert_planes_equal(want, got, f"film grain trial {trial} ({w}x{h})")
    ctx.close()


def test_film_grain_batch_and_errors(torch_cuda):
    from easyav1_amd import ir, video
    from easyav1_amd import grain as gr
    from easyav1_amd._lib import Ea1Error
    W, H = 320, 192
    bare = video.ReconContext(W, H)
    bare.set_ref(0, ir.synth_picture(W, H, 1))
    with pytest.raises(Ea1Error, match="Gaussian"):
        gr.film_grain(bare, 0, gr.random_params(np.random.default_rng(0)))
    bare.close()
    ctx, grain = _grain_ctx(W, H, nchains=5)
    rng = np.random.default_rng(77)
    pics = [ir.synth_picture(W, H, 10 + c) for c in range(5)]
    params = [grain.random_params(rng) for _ in range(5)]
    for c in range(5):
        ctx.set_ref(c, pics[c], chain=c)
    slots = [0, 1, -1, 3, 4]
    params[2] = None
    grain.film_grain_batch(ctx, slots, params)
    for c in (0, 1, 3, 4):
        assert_planes_equal(ol.oracle_film_grain(params[c], pics[c]), grain.download_display(ctx, chain=c), f"chain {c}")
    with pytest.raises(Ea1Error):
        grain.download_display(ctx, chain=2)            # never written
    with pytest.raises(Ea1Error, match="holds no picture"):
        grain.film_grain(ctx, 5, params[0], chain=0)
    bad = grain.random_params(rng, scaling_shift=12)
    with pytest.raises(Ea1Error, match="out of range"):
        grain.film_grain(ctx, 0, bad, chain=0)
    bad = grain.random_params(rng, num_y_points=2)
    bad.point_y[0][0], bad.point_y[1][0] = 50, 50
    with pytest.raises(Ea1Error, match="increase"):
        grain.film_grain(ctx, 0, bad, chain=0)
    # the reference slot itself stays free of grain
    assert_planes_equal(pics[0], ctx.download(0, chain=0), "reference slot after film grain")
    ctx.close()


def test_async_batch_download_matches_sync_download(torch_cuda):
    """the delivery path (one device-layout copy per chain into pinned memory, no synchronisation in the
    call) hands out the same pictures as the synchronous per-plane download"""
    from easyav1_amd import ir, video
    torch = torch_cuda
    W, H, S = 320, 192, 3
    ctx = video.ReconContext(W, H, nchains=S)
    pics = [ir.synth_picture(W, H, 40 + c) for c in range(S)]
    for c in range(S):
        ctx.set_ref(2, pics[c], chain=c)
    off, stride, nbytes = ctx.picture_layout()
    assert off[0] == 0 and stride[0] >= W and stride[1] >= W // 2 and nbytes >= off[2] + stride[2] * (H // 2)
    buf = video.PinnedBuffer(S * nbytes)
    ctx.download_batch_async([2, -1, 2], [buf.ptr.value + c * nbytes for c in range(S)], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = buf.array()
    for c in (0, 2):
        blob = host[c * nbytes:(c + 1) * nbytes]
        for pl in range(3):
            pw, ph = (W, H) if pl == 0 else (W // 2, H // 2)
            plane = blob[off[pl]: off[pl] + stride[pl] * ph].reshape(ph, stride[pl])[:, :pw]
            assert np.array_equal(plane, pics[c][pl]), (c, pl)
    from easyav1_amd._lib import Ea1Error
    with pytest.raises(Ea1Error, match="holds no picture"):
        ctx.download_batch_async([5, -1, -1], [buf.ptr.value] * S, torch.cuda.current_stream().cuda_stream)
    buf.close()
    ctx.close()


@pytest.mark.parametrize("crop", [(1, 0), (0, 1), (3, 5), (7, 7), (4, 2), (6, 3)])
def test_frame_sizes_that_are_not_multiples_of_8(torch_cuda, crop):
    """visible size = coded size minus 0..7 samples: reference reads clamp at the visible edge (also for odd
    chroma widths), restoration runs over the visible plane only (unit counts, replicated edges, dwords that
    straddle the edge) and leaves the CDEF output beyond it; every tool, every stage"""
    cr, cb = crop
    decode_chain(torch_cuda, 200, 120, 4, seed=130 + cr, features=0x7FFF, crop_right=cr, crop_bottom=cb, check_stages=True)
    decode_chain(torch_cuda, 320, 192, 3, tile_cols=2, seed=140 + cb, features=0x7FFF | SYN_WARP, crop_right=cr, crop_bottom=cb, check_stages=False)
    decode_chain(torch_cuda, 64, 64, 3, seed=150, crop_right=cr, crop_bottom=cb, check_stages=False)


def test_film_grain_on_cropped_pictures(torch_cuda):
    from easyav1_amd import ir
    ctx, grain = _grain_ctx(200, 120)
    rng = np.random.default_rng(5)
    for trial in range(16):
        pic = ir.synth_picture(200, 120, 300 + trial)
        ctx.set_ref(0, pic)
        fg = grain.random_params(rng, crop=int(rng.integers(8)) | (int(rng.integers(8)) << 4))
        if trial < 3:
            fg.overlap_flag, fg.num_y_points, fg.chroma_scaling_from_luma = 1, max(fg.num_y_points, 2), 0
        grain.film_grain(ctx, 0, fg)
        assert_planes_equal(ol.oracle_film_grain(fg, pic), grain.download_display(ctx), f"cropped film grain trial {trial} crop {fg.crop:#x}")
    ctx.close()


def test_inter_frame_without_its_reference_is_refused(torch_cuda):
    from easyav1_amd import ir, video
    from easyav1_amd._lib import Ea1Error
    p = ir.synth_params(192, 128, 1, seed=2)
    ctx = video.ReconContext(192, 128)
    inter = ctx.make_job(ir.synth_frame(p, 1))
    with pytest.raises(Ea1Error, match="holds no picture"):
        ctx.submit(inter)                      # nothing decoded yet
    key = ctx.make_job(ir.synth_frame(p, 0))
    ctx.submit(key)
    ctx.submit(inter)                          # its reference exists now
    ctx.flush()
    with pytest.raises(Ea1Error, match="holds no picture"):
        ctx.submit(inter)                      # a seek empties the slots
    ctx.close()


@pytest.mark.parametrize("scenario", ["inter", "intra", "tiles", "small", "scaled"])
def test_mutated_frames_that_pass_validation_match_the_oracle(torch_cuda, scenario):
    """differential fuzzing: single-field corruptions of a valid inter frame that ea1_job_create still ACCEPTS are
    frames in their own right -- the GPU must decode them exactly like the oracle (unusual flag / mode /
    level combinations no generator would produce)"""
    import os
    from easyav1_amd import ir, video, Ea1Error
    W, H = {"tiles": (320, 192), "small": (72, 56)}.get(scenario, (192, 128))
    rng = np.random.default_rng(int(os.environ.get("EA1_DIFF_SEED", "99")))
    ctx = video.ReconContext(W, H)
    ctx.set_warp_filter(ir.synth_warp_filter())
    ol.oracle_set_warp_filter(ir.synth_warp_filter())
    layout = int(os.environ.get("EA1_FUZZ_LAYOUT", "0"), 0)   # a long one-off run per layout: EA1_FUZZ_LAYOUT=0x80000 / 0x100000
    if scenario == "tiles":   # 2 x 2 tiles, a visible size that is not a multiple of 8, restoration type per unit; packed diagonals
        p = ir.synth_params(W, H, 2, seed=16, features=0x7FFF | SYN_WARP | (layout or SYN_DIAGONAL_COEFS), tile_rows=2, crop_right=5, crop_bottom=2)
        key = ir.synth_frame(p, 0)
        ref = ol.oracle_decode_frame(key, [None] * 8)["final"]
        f = ir.synth_frame(p, 1)
    elif scenario == "scaled":   # the reference in slot 0 has another size than the frame: every prediction through the scaled path
        p = ir.synth_params(W, H, 1, seed=26, features=(0x7FFF & ~SYN_LR_MIX) | (layout or SYN_DIAGONAL_COEFS))
        srng = np.random.default_rng(3)
        ref = [srng.integers(0, 256, (96, 128), dtype=np.uint8), srng.integers(0, 256, (48, 64), dtype=np.uint8), srng.integers(0, 256, (48, 64), dtype=np.uint8)]
        f = ir.synth_frame(p, 1)
    elif scenario in ("inter", "small"):   # "small": a frame barely larger than a superblock, every block near an edge
        p = ir.synth_params(W, H, 1, seed=6, features=0x7FFF | SYN_WARP | layout)
        key = ir.synth_frame(p, 0)
        ref = ol.oracle_decode_frame(key, [None] * 8)["final"]
        f = ir.synth_frame(p, 1)
    else:   # an intra frame with intra block copy (no in-loop filters) and every intra tool; nothing is referenced
        p = ir.synth_params(W, H, 2, seed=8, features=0x7FFF | SYN_INTRABC)
        f = ir.synth_frame(p, 0)
        ref = ir.synth_picture(W, H, 1)
    refs = [ref] + [None] * 7
    ref_crops = [f.desc.crop if scenario != "scaled" else 0] + [0] * 7     # true size of the picture in every slot
    checked = 0
    want_checks = int(os.environ.get("EA1_DIFF_FUZZ", "120" if scenario == "inter" else "40"))
    for trial in range(40 * want_checks):
        if checked >= want_checks:
            break
        blocks, txs, coefs, aux = f.blocks.copy(), f.txs.copy(), f.coefs.copy(), f.aux.copy()
        cdef_idx, lr, desc = f.cdef_idx.copy(), [u.copy() for u in f.lr], f.desc.copy()
        kind = trial % 4
        if kind == 0:
            raw = blocks.view(np.uint8).reshape(len(blocks), -1)
            raw[rng.integers(len(blocks)), rng.integers(raw.shape[1])] = rng.integers(256)
        elif kind == 1:
            raw = txs.view(np.uint8).reshape(len(txs), -1)
            raw[rng.integers(len(txs)), rng.integers(raw.shape[1])] = rng.integers(256)
        elif kind == 2:
            cdef_idx[rng.integers(len(cdef_idx))] = rng.integers(256)
            for u in lr:
                if len(u):
                    u.view(np.uint8)[rng.integers(u.nbytes)] = rng.integers(256)
        else:
            import ctypes
            raw = (ctypes.c_uint8 * ctypes.sizeof(desc)).from_address(ctypes.addressof(desc))
            raw[8 + int(rng.integers(36))] = int(rng.integers(256))
        bad = ir.ParsedFrame(desc, blocks, txs, coefs, aux, cdef_idx, lr)
        if bad.desc.dst_slot >= 8 or (bad.desc.dst_slot == 0 and scenario != "intra"):
            continue   # keep the reference in slot 0 (the intra frame references nothing)
        try:
            j = ctx.make_job(bad)
        except Ea1Error:
            continue
        if scenario != "intra":
            # the reference has the TRUE size of the frame it came from (the "tiles" scenario has a crop: without it the picture would be a
            # scaled reference).  A mutation of the crop byte itself makes it one: decoded through the scaled path, by the oracle too
            ctx.set_ref(0, ref, crop=f.desc.crop if scenario != "scaled" else 0)
        try:
            ctx.submit(j)
        except Ea1Error:
            j.close()
            continue
        got = ctx.download(bad.desc.dst_slot)
        want = ol.oracle_decode_frame(bad, refs, ref_crops=ref_crops)["final"]
        if bad.desc.pad:   # (a mutated `pad`: the picture beyond the coded size it declares is undefined, ea1_frame_desc.pad)
            cw, ch = bad.desc.width - 8 * (bad.desc.pad & 15), bad.desc.height - 8 * (bad.desc.pad >> 4)
            want = [want[0][:ch, :cw], want[1][:ch // 2, :cw // 2], want[2][:ch // 2, :cw // 2]]
            got = [got[0][:ch, :cw], got[1][:ch // 2, :cw // 2], got[2][:ch // 2, :cw // 2]]
        assert_planes_equal(want, got, f"mutated frame, trial {trial} kind {kind}")
        refs[bad.desc.dst_slot] = want      # the slot now holds this picture on the device too
        ref_crops[bad.desc.dst_slot] = bad.desc.crop
        j.close()
        checked += 1
    assert checked >= min(want_checks, 60)
    ctx.close()


def test_batched_mutated_frames_match_the_oracle(torch_cuda):
    """the batched launch path under differential fuzzing: four chains, each holding its own reference, get a random
    mix of idle slots, the valid frame and accepted mutations of it in ONE submit; every chain must match the
    oracle (grids are sized by the largest chain, per-chain counts bound the work inside the kernels)"""
    import os
    from easyav1_amd import ir, video, Ea1Error
    W, H, S = 192, 128, 4
    rng = np.random.default_rng(int(os.environ.get("EA1_DIFF_SEED", "5")))
    ctx = video.ReconContext(W, H, nchains=S)
    ctx.set_warp_filter(ir.synth_warp_filter())
    ol.oracle_set_warp_filter(ir.synth_warp_filter())
    params = [ir.synth_params(W, H, 1, seed=20 + c, features=0x7FFF | SYN_WARP) for c in range(S)]
    refs = []
    for c in range(S):
        ref = ol.oracle_decode_frame(ir.synth_frame(params[c], 0), [None] * 8)["final"]
        ctx.set_ref(0, ref, chain=c)
        refs.append([ref] + [None] * 7)     # per chain: what every slot holds (oracle side)
    frames = [ir.synth_frame(params[c], 1) for c in range(S)]
    rounds = int(os.environ.get("EA1_DIFF_FUZZ", "25"))
    checked = 0
    for it in range(rounds):
        batch, jobs = [], []
        for c in range(S):
            f = frames[c]
            r = rng.integers(4)
            if r == 0:
                batch.append(None)
                jobs.append(None)
                continue
            cand = f
            if r >= 2:   # a mutation of the chain's frame, if validation lets it through
                blocks, txs = f.blocks.copy(), f.txs.copy()
                raw = (blocks if r == 2 else txs).view(np.uint8)
                raw = raw.reshape(len(blocks if r == 2 else txs), -1)
                raw[rng.integers(raw.shape[0]), rng.integers(raw.shape[1])] = rng.integers(256)
                cand = ir.ParsedFrame(f.desc.copy(), blocks, txs, f.coefs.copy(), f.aux.copy(), f.cdef_idx.copy(), [u.copy() for u in f.lr])
            try:
                job = ctx.make_job(cand)
            except Ea1Error:
                cand, job = f, ctx.make_job(f)
            if cand.desc.dst_slot >= 8:
                job.close()
                cand, job = f, ctx.make_job(f)
            batch.append(cand)
            jobs.append(job)
        try:
            ctx.submit_batch(jobs)
        except Ea1Error:      # a corrupted reference index pointing at an empty slot: the whole submit is refused
            for j in jobs:
                if j is not None:
                    j.close()
            continue
        for c in range(S):
            if batch[c] is None:
                continue
            got = ctx.download(batch[c].desc.dst_slot, chain=c)
            want = ol.oracle_decode_frame(batch[c], refs[c])["final"]
            assert_planes_equal(want, got, f"round {it} chain {c}")
            refs[c][batch[c].desc.dst_slot] = want
            checked += 1
        for j in jobs:
            if j is not None:
                j.close()
    assert checked >= rounds
    ctx.close()
