"""The tile-column split's NATIVE driver (include/easyav1_amd_split.h, easyav1_amd/csrc/av1_split_native.cpp): what tests/test_tile_split.py proves
for the Python driver -- every rank's assembled pictures equal the single-rank decode -- for ea1_split_decode, with the all-gather as a callback
(ranks = threads of one process, each with its own context on the one device) and as RCCL (one rank: the only communicator a one-GPU box can make;
the several-rank RCCL form is the same calls with world > 1 and has not run on hardware)."""
import threading

import numpy as np
import pytest
import torch

from easyav1_amd import ir, parallel
from tests.test_tile_split import reference_chain, run_ranks


def test_the_library_exports_the_split():
    import ctypes
    import easyav1_amd
    L = easyav1_amd.load()
    for name in ("ea1_split_create", "ea1_split_create_rccl", "ea1_rccl_unique_id", "ea1_split_job_create", "ea1_split_decode", "ea1_split_traffic", "ea1_split_destroy"):
        assert hasattr(L, name), name
    # argument checks need no device
    h = ctypes.c_void_p()
    L.ea1_split_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert L.ea1_split_create(ctypes.byref(h), None, None, 2, 0, None, None) < 0 and not h.value


class ThreadTable:
    """all-gather between `world` threads through host memory (the transport of the test, not of the product)"""

    def __init__(self, world):
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world, timeout=120)

    def for_rank(self, rank):
        def gather(send, recv):
            torch.cuda.synchronize()
            self.slots[rank] = send.cpu()
            self.barrier.wait()
            for q in range(self.world):
                recv[q].copy_(self.slots[q])
            torch.cuda.synchronize()
            self.barrier.wait()
        return gather


@pytest.fixture(scope="module")
def gpu():
    import easyav1_amd
    easyav1_amd.load()
    from easyav1_amd import video
    return video


@pytest.mark.gpu
@pytest.mark.parametrize("width,height,tile_cols,world,features,nframes,seed,chains", [
    (256, 128, 2, 2, None, 4, 1, 1),
    (512, 192, 4, 4, ir.SYN_ALL, 4, 2, 1),
    (448, 128, 4, 3, ir.SYN_ALL, 3, 3, 2),
    (1920, 1080, 4, 4, None, 3, 7, 1),
    (3840, 192, 8, 8, ir.SYN_ALL, 2, 9, 2),        # configs[3]'s split: 8 ranks, strips of 7 x 512 + 256 columns
    (3840, 2160, 8, 8, None, 2, 10, 1),            # ... at its full size
])
def test_native_split_assembles_the_oracle_frames(gpu, width, height, tile_cols, world, features, nframes, seed, chains):
    streams = []
    for c in range(chains):
        prm = ir.synth_params(width, height, tile_cols, seed=seed + 10 * c, gop=nframes)
        if features is not None:
            prm.features = features
        streams.append([ir.synth_frame(prm, k) for k in range(nframes)])
    want = [reference_chain(s) for s in streams]
    bounds = parallel.strip_bounds(parallel.uniform_tile_starts(width, tile_cols), world)
    table = ThreadTable(world)

    def rank_body(r):
        ctx = gpu.ReconContext(width, height, 0, nchains=chains)
        ts = parallel.NativeTileSplit(ctx, bounds, r, all_gather=table.for_rank(r))
        t = ts.traffic()
        assert t["halo_received"] == world * t["halo_sent"] and t["strip_received"] == world * t["strip_sent"] and t["strip_sent"] >= ctx.columns_bytes(bounds[r][1] - bounds[r][0])
        jobs = []
        for k in range(nframes):
            js = [ts.make_job(streams[c][k]) for c in range(chains)]
            jobs += js
            ts.decode(js, streams[0][k].desc.dst_slot)
            for c in range(chains):
                got = ctx.download(streams[c][k].desc.dst_slot, chain=c)
                for pl in range(3):
                    assert np.array_equal(got[pl], want[c][k][pl]), (r, c, k, pl)
        for j in jobs:
            j.close()
        ts.close()
        ctx.close()
    run_ranks(world, rank_body)


@pytest.mark.gpu
def test_native_split_over_rccl_with_one_rank(gpu):
    """librccl.so is loaded at run time, a communicator is made from a unique id, both all-gathers of every frame go through ncclAllGather: with the
    one rank a one-GPU box allows, the strip is the frame and what comes back is what went out -- the pictures are the plain decode's"""
    W, H, n = 512, 192, 4
    prm = ir.synth_params(W, H, 4, seed=5, gop=n)
    prm.features = ir.SYN_ALL
    frames = [ir.synth_frame(prm, k) for k in range(n)]
    want = reference_chain(frames)
    uid = parallel.rccl_unique_id()
    assert len(uid) == parallel.RCCL_UNIQUE_ID_BYTES and any(uid)
    ctx = gpu.ReconContext(W, H, 0, nchains=1)
    ts = parallel.NativeTileSplit(ctx, [(0, W)], 0, rccl_id=uid)
    jobs = []
    for k in range(n):
        j = ts.make_job(frames[k])
        jobs.append(j)
        ts.decode([j], frames[k].desc.dst_slot)
        got = ctx.download(frames[k].desc.dst_slot)
        for pl in range(3):
            assert np.array_equal(got[pl], want[k][pl]), (k, pl)
    for j in jobs:
        j.close()
    ts.close()
    ctx.close()
