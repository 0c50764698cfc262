"""GPU tests of the batched hand-over (ea1_job_create_batch: host halves on library threads, one issuing thread,
device-made work lists): parity with the oracle, all-or-nothing errors, and the alternative copy path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as ol
from tests.test_video_gpu import assert_planes_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_batch_hand_over_matches_oracle(torch_cuda, threads):
    """six chains (every tool, different seeds), three frames each, every frame of a step handed over in ONE call on
    1 / 3 / 16 library threads; jobs destroyed right after their submit (the arenas are recycled behind events)"""
    from easyav1_amd import ir, video
    W, H, NCH = 256, 192, 6
    ctx = video.ReconContext(W, H, nchains=NCH)
    params = [ir.synth_params(W, H, 1 + c % 2, seed=700 + c, features=0x7FFF if c % 3 else 0) for c in range(NCH)]
    refs = [[None] * 8 for _ in range(NCH)]
    prev = []
    for fi in range(3):
        frames = [ir.synth_frame(params[c], fi) for c in range(NCH)]
        jobs = ctx.make_jobs(frames, threads)
        ctx.submit_batch(jobs)
        for j in prev:
            j.close()
        prev = jobs
        for c in range(NCH):
            want = ol.oracle_decode_frame(frames[c], refs[c])["final"]
            assert_planes_equal(want, ctx.download(frames[c].desc.dst_slot, chain=c), f"threads {threads} chain {c} frame {fi}")
            refs[c][frames[c].desc.dst_slot] = want
    for j in prev:
        j.close()
    ctx.close()


def test_batch_is_all_or_nothing(torch_cuda):
    """one bad frame among five: the call fails, names the frame, leaves no job behind, and the context goes on working"""
    from easyav1_amd import ir, video, Ea1Error
    W, H = 192, 128
    ctx = video.ReconContext(W, H, nchains=5)
    p = ir.synth_params(W, H, 1, seed=9)
    good = [ir.synth_frame(p, 0) for _ in range(5)]
    bad = ir.ParsedFrame(good[0].desc.copy(), good[0].blocks.copy(), good[0].txs, good[0].coefs, good[0].aux, good[0].cdef_idx, good[0].lr)
    bad.blocks["bwl"][3] = 9
    with pytest.raises(Ea1Error) as e:
        ctx.make_jobs(good[:3] + [bad] + good[4:], 4)
    assert "frame 3" in str(e.value)
    jobs = ctx.make_jobs(good, 4)
    ctx.submit_batch(jobs)
    want = ol.oracle_decode_frame(good[0], [None] * 8)["final"]
    for c in range(5):
        assert_planes_equal(want, ctx.download(good[0].desc.dst_slot, chain=c), f"chain {c}")
    for j in jobs:
        j.close()
    ctx.close()


def test_lists_without_parser_hints(torch_cuda):
    """frames without the parser's corner hint (dense coefficients, nz_log2 = 0): the device finds the coded corner
    from the uploaded coefficients itself; with inter-intra blocks (zero-fill items) and warped motion (host-made tiles
    in front of the device-made ones)"""
    from easyav1_amd import ir, video
    W, H = 256, 192
    ctx = video.ReconContext(W, H, nchains=2)
    ctx.set_warp_filter(ir.synth_warp_filter())
    ol.oracle_set_warp_filter(ir.synth_warp_filter())
    params = [ir.synth_params(W, H, 1, seed=31 + c, features=(0x7FFF | 0x10000 | ir.SYN_NO_NZ_HINT) & ~ir.SYN_COMPACT_COEFS) for c in range(2)]
    refs = [[None] * 8 for _ in range(2)]
    for fi in range(3):
        frames = [ir.synth_frame(params[c], fi) for c in range(2)]
        assert frames[0].desc.coef_layout == ir.COEFS_DENSE and not frames[0].txs["nz_log2"].any()
        jobs = ctx.make_jobs(frames, 2)
        ctx.submit_batch(jobs)
        for c in range(2):
            want = ol.oracle_decode_frame(frames[c], refs[c])["final"]
            assert_planes_equal(want, ctx.download(frames[c].desc.dst_slot, chain=c), f"chain {c} frame {fi}")
            refs[c][frames[c].desc.dst_slot] = want
        for j in jobs:
            j.close()
    ctx.close()


_PULL_SNIPPET = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
torch.cuda.init()
from easyav1_amd import ir, video
from tests import oracle_lib as ol
W, H = 256, 192
ctx = video.ReconContext(W, H, nchains=3)
params = [ir.synth_params(W, H, 1, seed=50 + c, features=0x7FFF) for c in range(3)]
refs = [[None] * 8 for _ in range(3)]
pins = []
for fi in range(3):
    frames = [ir.synth_frame(params[c], fi) for c in range(3)]
    for f in frames:   # coefficients in pinned memory of the library: the source a kernel may pull from
        pb = video.PinnedBuffer(max(f.coefs.nbytes, 16))
        a = np.frombuffer(pb.array(), dtype=np.int16, count=len(f.coefs))
        a[:] = f.coefs
        f.coefs = a
        pins.append(pb)
    jobs = ctx.make_jobs(frames, 3)
    ctx.submit_batch(jobs)
    for c in range(3):
        want = ol.oracle_decode_frame(frames[c], refs[c])["final"]
        got = ctx.download(frames[c].desc.dst_slot, chain=c)
        for pl in range(3):
            assert np.array_equal(want[pl], got[pl]), (fi, c, pl)
        refs[c][frames[c].desc.dst_slot] = want
    for j in jobs:
        j.close()
print("PULL-OK")
"""


@pytest.mark.parametrize("env", [{"EA1_PULL": "1"}, {"EA1_COPY_STREAMS": "2"}])
def test_alternative_copy_paths_are_equivalent(torch_cuda, env):
    """the knobs of the hand-over's copy path (records and coefficients pulled by a kernel; a second copy stream) read
    their environment once per process: a child process decodes three chains with the knob set and checks the oracle"""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _PULL_SNIPPET % ROOT], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PULL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_threads_hand_batches_over_at_once(torch_cuda):
    """two caller threads, each with batches for its own chains of ONE context: the library runs one batch at a time,
    both decode exactly"""
    import threading
    from easyav1_amd import ir, video
    W, H, NCH = 192, 128, 4
    ctx = video.ReconContext(W, H, nchains=NCH)
    params = [ir.synth_params(W, H, 1, seed=900 + c, features=0x7FFF) for c in range(NCH)]
    frames = [[ir.synth_frame(params[c], fi) for c in range(NCH)] for fi in range(3)]
    for fi in range(3):
        halves, errs = [None, None], []

        def work(k):
            try:
                halves[k] = ctx.make_jobs(frames[fi][2 * k:2 * k + 2], 3)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        jobs = halves[0] + halves[1]
        ctx.submit_batch(jobs)
        for c in range(NCH):
            refs = [None] * 8
            for fj in range(fi + 1):   # the oracle's chain up to this frame
                want = ol.oracle_decode_frame(frames[fj][c], refs)["final"]
                refs[frames[fj][c].desc.dst_slot] = want
            assert_planes_equal(want, ctx.download(frames[fi][c].desc.dst_slot, chain=c), f"chain {c} frame {fi}")
        for j in jobs:
            j.close()
    ctx.close()
