"""Tile-column split of one stream over several ranks (BASELINE configs[3]; design in
include/easyav1_amd_video.h, driver easyav1_amd/parallel.py::TileSplit).

CPU: strip layout; the split algorithm itself -- which columns a rank has to receive for its strip to come out right --
on the CPU oracle, with every column a rank must not depend on filled with noise; the same through a real gloo
all-gather in two processes.  GPU (-m gpu): the HIP kernels' column windows against the oracle with several ranks
sharing the one device (threads in one process; 2 and 4 processes over gloo on the BASELINE configs[3] frame size)."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

from easyav1_amd import ir, parallel
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_strip_layout_of_the_baseline_configs():
    ts = parallel.uniform_tile_starts(3840, 8)          # configs[3]: 60 superblock columns -> 7 x 8 + 4
    assert ts == [0, 512, 1024, 1536, 2048, 2560, 3072, 3584, 3840]
    assert parallel.strip_bounds(ts, 8) == [(ts[k], ts[k + 1]) for k in range(8)]
    assert parallel.strip_bounds(ts, 4) == [(0, 1024), (1024, 2048), (2048, 3072), (3072, 3840)]
    assert parallel.strip_bounds(ts, 2) == [(0, 2048), (2048, 3840)]
    assert parallel.strip_bounds(ts, 1) == [(0, 3840)]
    assert parallel.uniform_tile_starts(3840, 4) == [0, 960, 1920, 2880, 3840]      # configs[2]
    assert parallel.uniform_tile_starts(1920, 1) == [0, 1920]
    for world in range(1, 9):
        b = parallel.strip_bounds(ts, world)
        assert b[0][0] == 0 and b[-1][1] == 3840 and all(b[k][1] == b[k + 1][0] for k in range(world - 1)) and all(e > a for a, e in b)
    with pytest.raises(ValueError):
        parallel.strip_bounds(ts, 9)


class OracleStripBackend:
    """TileSplit compute on the CPU oracle.  Every column this rank has not been GIVEN holds noise of its own:
    a result that equals the single-rank decode proves that the exchanged columns are all the strip depends on."""
    PIC_WORK = -1

    def __init__(self, width, height, nchains, noise_seed):
        self.W, self.H, self.n = width, height, nchains
        self.rng = np.random.default_rng(noise_seed)
        self.work = [self.noise() for _ in range(nchains)]
        self.slots = [[None] * 8 for _ in range(nchains)]

    def noise(self):
        W, H = self.W, self.H
        return [self.rng.integers(0, 256, s, dtype=np.uint8) for s in ((H, W), (H // 2, W // 2), (H // 2, W // 2))]

    def make_job(self, frame, x0, x1):
        return (frame, x0, x1)

    def recon(self, jobs):
        for c, (frame, x0, x1) in enumerate(jobs):
            self.work[c] = self.noise()              # nothing survives from the previous frame
            ol.oracle_strip_recon(frame, self.slots[c], x0, x1, self.work[c])

    def filter(self, jobs):
        for c, (frame, x0, x1) in enumerate(jobs):
            full = ol.oracle_strip_filter(frame, self.work[c])
            pic = self.noise()
            for pl in range(3):
                ss = 1 if pl else 0
                pic[pl][:, x0 >> ss:x1 >> ss] = full[pl][:, x0 >> ss:x1 >> ss]
            self.slots[c][frame.desc.dst_slot] = pic

    def columns_bytes(self, w):
        return w * self.H * 3 // 2

    def new_buffer(self, nbytes):
        return torch.zeros((self.n, nbytes), dtype=torch.uint8)

    def _planes(self, pic, c):
        return self.work[c] if pic == self.PIC_WORK else self.slots[c][pic]

    def get_columns(self, pic, x, w, out):
        for c in range(self.n):
            p = self._planes(pic, c)
            flat = np.concatenate([p[0][:, x:x + w].ravel(), p[1][:, x // 2:(x + w) // 2].ravel(), p[2][:, x // 2:(x + w) // 2].ravel()])
            out[c, :flat.size] = torch.from_numpy(flat)

    def put_columns(self, pic, x, w, buf):
        H = self.H
        for c in range(self.n):
            p = self._planes(pic, c)
            a = buf[c].numpy()
            ny, nc = w * H, (w // 2) * (H // 2)
            p[0][:, x:x + w] = a[:ny].reshape(H, w)
            p[1][:, x // 2:(x + w) // 2] = a[ny:ny + nc].reshape(H // 2, w // 2)
            p[2][:, x // 2:(x + w) // 2] = a[ny + nc:ny + 2 * nc].reshape(H // 2, w // 2)


class ThreadGather:
    """all_gather between `world` threads of one process"""

    def __init__(self, world):
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def for_rank(self, rank):
        def gather(t):
            self.slots[rank] = t.clone()
            self.barrier.wait()
            out = [s.clone() for s in self.slots]
            self.barrier.wait()
            return out
        return gather


def run_ranks(world, fn):
    """fn(rank) in `world` threads; re-raises the first failure"""
    errors = []

    def body(r):
        try:
            fn(r)
        except BaseException as e:   # noqa: BLE001
            errors.append(e)
            raise
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def reference_chain(frames):
    """single-rank oracle decode of a frame sequence -> list of final pictures"""
    refs, out = [None] * 8, []
    for f in frames:
        fin = ol.oracle_decode_frame(f, refs)["final"]
        refs[f.desc.dst_slot] = fin
        out.append(fin)
    return out


@pytest.mark.parametrize("width,height,tile_cols,world,features,nframes,seed", [
    (256, 128, 2, 2, None, 4, 1),                      # graded tool set
    (512, 192, 4, 4, ir.SYN_ALL, 4, 2),                # every tool, one tile column per rank, strips of 64 and 128
    (448, 128, 4, 3, ir.SYN_ALL, 3, 3),                # uneven strips (2 + 1 + 1 tile columns)
    (512, 136, 8, 2, ir.SYN_ALL & ~ir.SYN_SUB8, 3, 4),  # height not a multiple of 64, 4 tile columns per rank
    (320, 128, 4, 2, ir.SYN_ALL | ir.SYN_WARP, 3, 5),
    # BASELINE configs[3] at its full width: 8 tile columns over 8 ranks, strips of 7 x 512 + 256 columns (the last rank's strip is half
    # as wide: its right halo does not exist, its left neighbour's right halo is its whole left quarter)
    (3840, 128, 8, 8, None, 3, 6),
    (3840, 192, 8, 8, ir.SYN_ALL, 2, 7),
])
def test_split_decode_equals_single_rank_decode_on_the_oracle(width, height, tile_cols, world, features, nframes, seed):
    prm = ir.synth_params(width, height, tile_cols, seed=seed, gop=nframes)
    if features is not None:
        prm.features = features
    if features is not None and features & ir.SYN_WARP:
        ol.oracle_set_warp_filter(ir.synth_warp_filter())
    frames = [ir.synth_frame(prm, k) for k in range(nframes)]
    want = reference_chain(frames)
    bounds = parallel.strip_bounds(parallel.uniform_tile_starts(width, tile_cols), world)
    tg = ThreadGather(world)
    got = [None] * world

    def rank_body(r):
        be = OracleStripBackend(width, height, 1, noise_seed=100 + r)
        ts = parallel.TileSplit(be, bounds, r, tg.for_rank(r))
        for f in frames:
            ts.decode([ts.make_job(f)], f.desc.dst_slot)
        got[r] = [be.slots[0][f.desc.dst_slot] for f in frames[-2:]]
        for k, f in enumerate(frames):   # checked below for the last frames; every frame's strip entered the next one's prediction
            pass
    run_ranks(world, rank_body)
    for r in range(world):
        for k in (-2, -1):
            for pl in range(3):
                assert np.array_equal(got[r][k][pl], want[k][pl]), (r, k, pl)


def test_halo_narrower_than_16_columns_is_not_enough():
    """the test above has teeth: with an 8-column halo the assembled frames differ"""
    width, height, world = 256, 128, 2
    prm = ir.synth_params(width, height, 2, seed=1, gop=3)
    frames = [ir.synth_frame(prm, k) for k in range(2)]
    want = reference_chain(frames)
    bounds = parallel.strip_bounds(parallel.uniform_tile_starts(width, 2), world)
    tg = ThreadGather(world)
    bad = []
    old = parallel.HALO
    parallel.HALO = 8
    try:
        def rank_body(r):
            be = OracleStripBackend(width, height, 1, noise_seed=7 + r)
            ts = parallel.TileSplit(be, bounds, r, tg.for_rank(r))
            for f in frames:
                ts.decode([ts.make_job(f)], f.desc.dst_slot)
            bad.append(any(not np.array_equal(be.slots[0][frames[-1].desc.dst_slot][pl], want[-1][pl]) for pl in range(3)))
        run_ranks(world, rank_body)
    finally:
        parallel.HALO = old
    assert any(bad)


# ---------------------------------------------------------------------------------------------- processes + gloo
def _gloo_rank(rank, world, port, width, height, tile_cols, nframes, seed, backend_kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    if backend_kind == "gpu":
        torch.cuda.init()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyav1_amd import ir as ir_, parallel as par
        prm = ir_.synth_params(width, height, tile_cols, seed=seed, gop=nframes)
        frames = [ir_.synth_frame(prm, k) for k in range(nframes)]
        bounds = par.strip_bounds(par.uniform_tile_starts(width, tile_cols), world)
        if backend_kind == "gpu":
            from easyav1_amd import video
            ctx = video.ReconContext(width, height, 0, nchains=1)
            be = par.GpuStripBackend(ctx)
            gather = par.torch_all_gather(stage_through_host=True)
        else:
            be = OracleStripBackend(width, height, 1, noise_seed=50 + rank)
            gather = par.torch_all_gather(stage_through_host=True)
        ts = par.TileSplit(be, bounds, rank, gather)
        digests = []
        import hashlib
        jobs = []
        for f in frames:
            job = ts.make_job(f)
            jobs.append(job)
            ts.decode([job], f.desc.dst_slot)
            planes = ctx.download(f.desc.dst_slot) if backend_kind == "gpu" else be.slots[0][f.desc.dst_slot]
            digests.append(hashlib.sha256(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest())
        q.put((rank, digests))
        dist.barrier()
        if backend_kind == "gpu":
            for j in jobs:
                j.close()
            ctx.close()
    finally:
        dist.destroy_process_group()


def run_gloo(world, width, height, tile_cols, nframes, seed, backend_kind):
    import torch.multiprocessing as mp
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    port = 29500 + (os.getpid() * 7 + world) % 2000
    procs = [ctxmp.Process(target=_gloo_rank, args=(r, world, port, width, height, tile_cols, nframes, seed, backend_kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, d = q.get(timeout=900)
        results[r] = d
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def digests_of(pictures):
    import hashlib
    return [hashlib.sha256(b"".join(np.ascontiguousarray(p).tobytes() for p in planes)).hexdigest() for planes in pictures]


def test_two_processes_over_gloo_on_the_oracle():
    """world_size 2, real collective: every rank ends up with every whole frame of the single-rank decode"""
    W, H, tiles, N, seed = 256, 128, 2, 3, 6
    prm = ir.synth_params(W, H, tiles, seed=seed, gop=N)
    want = digests_of(reference_chain([ir.synth_frame(prm, k) for k in range(N)]))
    res = run_gloo(2, W, H, tiles, N, seed, "oracle")
    assert res[0] == want and res[1] == want


def test_eight_processes_over_gloo_on_the_oracle_at_the_width_of_configs3():
    """world_size 8 with a real collective, 3840 columns in 8 tile columns (7 x 512 + 256): the two all-gathers per frame with
    strips of unequal width padded to the widest"""
    W, H, tiles, N, seed = 3840, 64, 8, 3, 9
    prm = ir.synth_params(W, H, tiles, seed=seed, gop=N)
    want = digests_of(reference_chain([ir.synth_frame(prm, k) for k in range(N)]))
    res = run_gloo(8, W, H, tiles, N, seed, "oracle")
    for r in range(8):
        assert res[r] == want, r


# ---------------------------------------------------------------------------------------------- GPU
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
    (512, 136, 8, 2, ir.SYN_ALL & ~ir.SYN_SUB8, 3, 4, 1),
    (1920, 1080, 4, 4, None, 3, 7, 1),
    (3840, 256, 8, 8, None, 3, 8, 1),              # configs[3]'s split: 8 ranks, strips of 7 x 512 + 256 columns
    (3840, 192, 8, 8, ir.SYN_ALL, 2, 9, 2),
    (3840, 2160, 8, 8, None, 2, 10, 1),            # ... at its full size
])
def test_gpu_strips_assemble_to_the_oracle_frames(gpu, width, height, tile_cols, world, features, nframes, seed, chains):
    """several ranks as threads of one process, each with its own context on the one device"""
    streams = []
    for c in range(chains):
        prm = ir.synth_params(width, height, tile_cols, seed=seed + 10 * c, gop=nframes)
        if features is not None:
            prm.features = features
        streams.append([ir.synth_frame(prm, k) for k in range(nframes)])
    want = [reference_chain(s) for s in streams]
    bounds = parallel.strip_bounds(parallel.uniform_tile_starts(width, tile_cols), world)
    tg = ThreadGather(world)
    lock = threading.Lock()     # the ranks share one HIP stream pool: keep their submissions whole

    def rank_body(r):
        ctx = gpu.ReconContext(width, height, 0, nchains=chains)
        be = parallel.GpuStripBackend(ctx)

        def gather(t):
            torch.cuda.synchronize()
            return [x.cuda() for x in tg.for_rank(r)(t.cpu())]
        ts = parallel.TileSplit(be, bounds, r, gather)
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
        ctx.close()
    run_ranks(world, rank_body)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_gpu_4k_eight_tile_columns_over_gloo_processes(gpu, world):
    """BASELINE configs[3] frame (3840x2160, 8 tile columns), `world` processes sharing the device, strips staged
    through host memory by gloo: 8 frames, every rank's assembled pictures equal the single-rank GPU decode and
    the oracle"""
    W, H, tiles, N, seed = 3840, 2160, 8, 8, 3
    prm = ir.synth_params(W, H, tiles, seed=seed, gop=N)
    frames = [ir.synth_frame(prm, k) for k in range(N)]
    ctx = gpu.ReconContext(W, H)
    single = []
    for f in frames:
        job = ctx.make_job(f)
        ctx.submit(job)
        single.append(ctx.download(f.desc.dst_slot))
        job.close()
    ctx.close()
    want = digests_of(single)
    if world == 2:
        assert want == digests_of(reference_chain(frames))     # the oracle, once (seconds per 4K frame)
    res = run_gloo(world, W, H, tiles, N, seed, "gpu")
    for r in range(world):
        assert res[r] == want, r
