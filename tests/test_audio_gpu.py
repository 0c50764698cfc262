"""GPU parity of the Vorbis synthesis path: HIP (through the C ABI) vs the oracle, bit-exact."""
import os

import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "audio_golden.npz")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same(want, got, what=""):
    """Bit-exact comparison; NaNs compare equal whatever their payload/sign (x86 and gfx950
    generate different default NaNs for inf-inf; no finite spectrum produces one)."""
    want = np.ascontiguousarray(want, np.float32)
    got = np.ascontiguousarray(got, np.float32)
    assert want.shape == got.shape, (what, want.shape, got.shape)
    bad = (bits(want) != bits(got)) & ~(np.isnan(want) & np.isnan(got))
    if bad.any():
        idx = np.argwhere(bad)
        rows = sorted(set(int(i[0]) for i in idx))
        first = tuple(idx[0])
        raise AssertionError(f"{what}: {bad.sum()} mismatching values, rows {rows[:10]}, first at {first}: "
                             f"want {want[first]!r} ({bits(want)[first]:#x}) got {got[first]!r} ({bits(got)[first]:#x})")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def spectra_cases(n, rng, count):
    n2 = n // 2
    rows = [np.zeros(n2, np.float32)]
    for k in (0, 1, n2 - 1):
        e = np.zeros(n2, np.float32)
        e[k] = 1.0
        rows.append(e)
    rows.append((np.full(n2, 3.0e38) * np.where(np.arange(n2) % 2, -1, 1)).astype(np.float32))
    rows.append((rng.standard_normal(n2) * 1e-40).astype(np.float32))  # denormal inputs
    rows.append((rng.standard_normal(n2) * 1e-30).astype(np.float32))  # denormal products
    while len(rows) < count:
        rows.append(rng.laplace(scale=20.0, size=n2).astype(np.float32))
    return np.stack(rows)


@pytest.mark.parametrize("n", SIZES)
def test_mdct_backward_bit_exact(torch_cuda, n):
    torch = torch_cuda
    from easyav1_amd import audio
    rng = np.random.default_rng(100 + n)
    spec = spectra_cases(n, rng, 40)
    with np.errstate(all="ignore"):
        want = ol.OracleMdct(n).backward(spec)
    got = audio.mdct_backward(torch.from_numpy(spec).cuda(), n).cpu().numpy()
    assert_same(want, got)


def test_mdct_golden(torch_cuda):
    torch = torch_cuda
    from easyav1_amd import audio
    g = np.load(GOLDEN)
    for n in SIZES:
        got = audio.mdct_backward(torch.from_numpy(g[f"mdct_in_{n}"]).cuda(), n).cpu().numpy()
        assert_same(g[f"mdct_out_{n}"], got, f"golden mdct {n}")


def make_stream(rng, ch, bs0, bs1, flags):
    spectra = np.zeros((len(flags), ch, bs1 // 2), np.float32)
    for k, W in enumerate(flags):
        n2 = (bs0, bs1)[W] // 2
        s = rng.laplace(scale=1.0, size=(ch, n2)) / (1.0 + np.arange(n2) / 8.0)
        s[:, int(0.8 * n2):] = 0
        spectra[k, :, :n2] = s
        spectra[k, :, n2:] = 777.0  # must be ignored for short blocks
    return spectra


def oracle_pcm(ch, bs0, bs1, flags, spectra, restart_at=()):
    m = {0: ol.OracleMdct(bs0), 1: ol.OracleMdct(bs1)}
    blocks = [m[W].backward(spectra[k, :, : (bs0, bs1)[W] // 2]) for k, W in enumerate(flags)]
    outs = ol.oracle_synth_stream(ch, bs0, bs1, flags, blocks, restart_at)
    return outs


@pytest.mark.parametrize("ch,bs0,bs1", [(2, 256, 2048), (6, 256, 2048), (1, 64, 64), (3, 128, 8192), (2, 512, 512)])
def test_decode_stream_bit_exact(torch_cuda, ch, bs0, bs1):
    torch = torch_cuda
    from easyav1_amd import audio
    rng = np.random.default_rng(ch * 1000 + bs1)
    nb = 61  # not a multiple of the kernel's run length
    flags = [int(rng.random() < 0.6) for _ in range(nb)]
    flags[:6] = [1, 1, 0, 0, 1, 0]  # all four (lW, W) cases early on
    spectra = make_stream(rng, ch, bs0, bs1, flags)
    want = np.concatenate(oracle_pcm(ch, bs0, bs1, flags, spectra), axis=1)
    syn = audio.VorbisSynth(ch, bs0, bs1)
    pcm, counts = syn.decode(torch.from_numpy(spectra).cuda(), flags)
    got = pcm.cpu().numpy()
    assert got.shape == want.shape
    assert_same(want, got)
    assert counts[0] == 0 and counts.sum() == want.shape[1]
    syn.close()


def test_decode_split_calls_and_restart(torch_cuda):
    """N calls of one block == one call of N blocks == reference block by block (state carry)."""
    torch = torch_cuda
    from easyav1_amd import audio
    ch, bs0, bs1 = 2, 256, 2048
    rng = np.random.default_rng(5)
    flags = [1, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 1, 0, 1, 1]
    spectra = make_stream(rng, ch, bs0, bs1, flags)
    restart_at = (7, 12)
    want = oracle_pcm(ch, bs0, bs1, flags, spectra, restart_at)
    dspec = torch.from_numpy(spectra).cuda()
    syn = audio.VorbisSynth(ch, bs0, bs1)
    # irregular batches with restarts between calls
    cuts = [0, 1, 2, 7, 12, 13, len(flags)]
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a in restart_at:
            syn.restart()
        pcm, counts = syn.decode(dspec[a:b].contiguous(), flags[a:b])
        pcm = pcm.cpu().numpy()
        pos = 0
        for c in counts:
            got.append(pcm[:, pos:pos + c])
            pos += c
    assert len(got) == len(want)
    for k, (w, g_) in enumerate(zip(want, got)):
        assert w.shape == g_.shape, k
        assert_same(w, g_, f"block {k}")
    syn.close()


def test_decode_golden(torch_cuda):
    torch = torch_cuda
    from easyav1_amd import audio
    g = np.load(GOLDEN)
    for key in [k for k in g.files if k.startswith("ola_flags_")]:
        tag = key[len("ola_flags_"):]
        ch, bs0, bs1 = (int(v) for v in g[f"ola_cfg_{tag}"])
        flags = [int(v) for v in g[key]]
        restart = [int(v) for v in g[f"ola_restart_{tag}"]]
        spectra = torch.from_numpy(g[f"ola_spec_{tag}"]).cuda()
        syn = audio.VorbisSynth(ch, bs0, bs1)
        cuts = sorted(set([0, len(flags)] + restart))
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if a in restart:
                syn.restart()
            pcm, _ = syn.decode(spectra[a:b].contiguous(), flags[a:b])
            parts.append(pcm.cpu().numpy())
        got = np.concatenate(parts, axis=1)
        assert_same(g[f"ola_pcm_{tag}"], got, tag)
        syn.close()


def test_full_size_properties(torch_cuda):
    """C5-sized run (6 ch, 256/2048, ~60 s): linearity-free checks that do not need the oracle at
    full size: sample count, determinism, and a sampled bit-exact comparison of a window."""
    torch = torch_cuda
    from easyav1_amd import audio
    ch, bs0, bs1 = 6, 256, 2048
    rng = np.random.default_rng(99)
    nb = 3000
    flags = np.ones(nb, np.uint8)
    i = 50
    while i < nb - 8:
        if rng.random() < 0.15:
            flags[i:i + 8] = 0
            i += 8
        i += 10
    spectra = torch.randn((nb, ch, bs1 // 2), device="cuda") * (1.0 / (1.0 + torch.arange(bs1 // 2, device="cuda") / 8.0))
    syn = audio.VorbisSynth(ch, bs0, bs1)
    pcm1, counts = syn.decode(spectra, flags)
    pcm1 = pcm1.clone()
    expect = sum((bs0, bs1)[flags[k - 1]] // 4 + (bs0, bs1)[flags[k]] // 4 for k in range(1, nb))
    assert pcm1.shape[1] == expect == counts.sum()
    syn2 = audio.VorbisSynth(ch, bs0, bs1)
    pcm2, _ = syn2.decode(spectra, flags)
    assert torch.equal(pcm1.view(torch.int32), pcm2.view(torch.int32))
    # oracle on a window of 40 blocks in the middle (the block before provides the overlap)
    a, b = 1200, 1240
    sp = spectra[a - 1:b].cpu().numpy()
    fl = [int(v) for v in flags[a - 1:b]]
    want = np.concatenate(oracle_pcm(ch, bs0, bs1, fl, sp), axis=1)
    off = int(counts[:a].sum())
    got = pcm1[:, off:off + want.shape[1]].cpu().numpy()
    assert_same(want, got)
    syn.close()
    syn2.close()


def test_capacity_error(torch_cuda):
    torch = torch_cuda
    from easyav1_amd import audio, Ea1Error
    syn = audio.VorbisSynth(1, 64, 64)
    spectra = torch.zeros((4, 1, 32), device="cuda")
    small = torch.zeros((1, 8), device="cuda")
    with pytest.raises(Ea1Error):
        syn.decode(spectra, [0, 0, 0, 0], out=small)
    syn.close()


@pytest.mark.gpu
def test_streams_batched_into_one_launch_equal_separate_decodes():
    """ea1_vorbis_synth_decode_batch: streams with different channel counts, block sizes and lengths in one launch,
    two calls in a row (carried overlap state), against one ea1_vorbis_synth_decode per stream"""
    import torch
    from easyav1_amd import audio
    rng = np.random.default_rng(5)
    cfgs = [(6, 256, 2048, 40), (2, 256, 2048, 7), (1, 64, 512, 90), (3, 128, 128, 25)] + [(2, 256, 2048, 30 + k) for k in range(15)]   # 19 streams: two launches
    a = [audio.VorbisSynth(ch, b0, b1) for ch, b0, b1, _ in cfgs]
    b = [audio.VorbisSynth(ch, b0, b1) for ch, b0, b1, _ in cfgs]
    for _ in range(2):
        spectra, flags = [], []
        for ch, b0, b1, nb in cfgs:
            fl = rng.integers(0, 2, nb).astype(np.uint8)
            sp = torch.randn((nb, ch, b1 // 2), device="cuda")
            spectra.append(sp)
            flags.append(fl)
        got = audio.decode_batch(a, spectra, flags)
        for k, syn in enumerate(b):
            want, _ = syn.decode(spectra[k], flags[k])
            assert got[k].shape == want.shape and torch.equal(got[k].view(torch.int32), want.view(torch.int32)), k
    for syn in a + b:
        syn.close()


@pytest.mark.gpu
def test_audio_synthesis_runs_beside_video_reconstruction():
    """BASELINE configs[4]: IMDCT 256 / 2048 of a 5.1 stream on its own HIP stream WHILE batched 1080p chains reconstruct
    on another: both results are those of the runs alone (which other tests pin against the oracles)"""
    import torch
    from easyav1_amd import audio, ir, video
    W, H, S, N = 1920, 1080, 4, 4
    ctx = video.ReconContext(W, H, nchains=S)
    chains = []
    for s in range(S):
        p = ir.synth_params(W, H, 1, seed=60 + s, gop=N)
        chains.append([ctx.make_job(ir.synth_frame(p, k)) for k in range(N)])
    rng = np.random.default_rng(9)
    nb, ch = 400, 6
    flags = (rng.random(nb) < 0.85).astype(np.uint8)
    spectra = torch.randn((nb, ch, 1024), device="cuda")
    syn = audio.VorbisSynth(ch, 256, 2048)

    def video_pass():
        for k in range(N):
            ctx.submit_batch([chains[s][k] for s in range(S)])

    # alone
    video_pass()
    torch.cuda.synchronize()
    want_v = [ctx.download((N - 1) % 8, chain=s) for s in range(S)]
    want_a = syn.decode(spectra, flags)[0].clone()
    torch.cuda.synchronize()
    # together, several rounds so that the kernels really overlap
    sv, sa = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        syn.restart()
        with torch.cuda.stream(sa):
            got_a = syn.decode(spectra, flags)[0]
        with torch.cuda.stream(sv):
            video_pass()
        torch.cuda.synchronize()
        assert torch.equal(got_a.view(torch.int32), want_a.view(torch.int32))
        for s in range(S):
            got_v = ctx.download((N - 1) % 8, chain=s)
            for pl in range(3):
                assert np.array_equal(got_v[pl], want_v[s][pl])
    # one chain against the oracle, to tie "alone" to the checker in this test too
    from tests import oracle_lib as ol
    p = ir.synth_params(W, H, 1, seed=60, gop=N)
    refs = [None] * 8
    for k in range(N):
        f = ir.synth_frame(p, k)
        fin = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = fin
    for pl in range(3):
        assert np.array_equal(want_v[0][pl], fin[pl])
    for c in chains:
        for j in c:
            j.close()
    ctx.close()
    syn.close()
