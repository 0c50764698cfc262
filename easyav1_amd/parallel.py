"""One process per GPU, two ways to shard the path:

(i)  independent GOP chains: closed GOPs are independent units, NO data-path collective;
     torch.distributed only carries the start/stop barrier and the max-over-ranks timing of bench.py.
(ii) tile columns of ONE stream (BASELINE.json configs[3]): rank g owns a contiguous range of tile columns of every
     frame; per frame two collectives -- an all-gather of 16-column halos between reconstruction and the in-loop
     filters, and an all-gather of the finished strips so that every rank holds the whole reference picture
     (`TileSplit`).  The reference is a single process (src/easyav1.c:2271); the split is this design's own.
"""
import ctypes
import os


def dist_env():
    """(rank, world, local_rank) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def chain_ids(rank, world, chains_per_rank):
    """Global ids of the GOP chains rank `rank` decodes (weak scaling: every rank owns the same
    number of chains; ids are disjoint and cover range(world * chains_per_rank))."""
    if not (0 <= rank < world) or chains_per_rank < 1:
        raise ValueError("bad rank/world/chains")
    return [rank * chains_per_rank + i for i in range(chains_per_rank)]


def chain_seed(base_seed, chain_id):
    """Seed of a chain's synthetic stream: depends only on the GLOBAL chain id, so the set of
    streams decoded by N ranks x C chains equals that of 1 rank x N*C chains."""
    return base_seed + 1000 * (chain_id // 1000) + chain_id % 1000 if chain_id >= 1000 else base_seed + chain_id


def owner_of_gop(gop_index, world, chains_per_rank):
    """(rank, local chain) that decodes GOP number `gop_index` of ONE long stream when its GOPs are
    dealt round-robin over all chains of the job."""
    g = gop_index % (world * chains_per_rank)
    return g // chains_per_rank, g % chains_per_rank


def global_max(value, device=None):
    """max over ranks of a python float (bench.py's timing rule); identity without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


# ------------------------------------------------------------------------------------------------------------
# (ii) tile-column split
HALO = 16   # luma columns of reconstruction a rank needs from each neighbour (include/easyav1_amd_video.h explains why)


def uniform_tile_starts(width, tile_cols):
    """luma x of every tile column start + [width]: AV1 uniform tile spacing on 64x64 superblocks
    (tile width = ceil(sb_cols / 2^ceil(log2(tile_cols))) superblocks, the last tile takes the rest)"""
    sb_cols = (width + 63) // 64
    log2 = max(tile_cols - 1, 0).bit_length()
    tile_w = (sb_cols + (1 << log2) - 1) >> log2
    return [sx * 64 for sx in range(0, sb_cols, tile_w)] + [width]


def strip_bounds(tile_starts, world):
    """[(x_begin, x_end)] per rank: contiguous runs of tile columns, boundaries at the tile starts nearest to
    equal shares of the width; every rank gets at least one tile column"""
    ntiles = len(tile_starts) - 1
    if not (1 <= world <= ntiles):
        raise ValueError(f"{world} ranks cannot share {ntiles} tile columns")
    width = tile_starts[-1]
    cuts = [0]
    for r in range(1, world):
        lo, hi = cuts[-1] + 1, ntiles - (world - r)          # leave a tile for every rank before and after
        cuts.append(min(range(lo, hi + 1), key=lambda t: abs(tile_starts[t] - r * width / world)))
    cuts.append(ntiles)
    return [(tile_starts[cuts[r]], tile_starts[cuts[r + 1]]) for r in range(world)]


class TileSplit:
    """Per-frame driver of one rank.  `backend` supplies the compute (GpuStripBackend below; the tests also run it on
    the CPU oracle), `all_gather(t)` the collective: a uint8 tensor -> list of `world` tensors of the same shape.

    Bytes per frame and rank (4K, 8 ranks, strips of 512 columns): halo all-gather 2 x 16 columns x 2160 rows x 1.5 =
    104 KB contributed, 830 KB received; strip all-gather 1.66 MB contributed, 13.3 MB received (the whole picture)."""

    def __init__(self, backend, bounds, rank, all_gather):
        self.b, self.bounds, self.rank, self.world, self.all_gather = backend, bounds, rank, len(bounds), all_gather
        self.x0, self.x1 = bounds[rank]
        self.max_w = max(b - a for a, b in bounds)

    def make_job(self, frame):
        return self.b.make_job(frame, self.x0, self.x1)

    def decode(self, jobs, dst_slot):
        """jobs: one per chain (all strips of this rank, same frame position of each stream)"""
        b, r = self.b, self.rank
        b.recon(jobs)
        if self.world > 1:
            mine = b.new_buffer(2 * b.columns_bytes(HALO))
            n = b.columns_bytes(HALO)
            b.get_columns(b.PIC_WORK, self.x0, HALO, mine[:, :n])
            b.get_columns(b.PIC_WORK, self.x1 - HALO, HALO, mine[:, n:])
            everyone = self.all_gather(mine)
            if r > 0:
                b.put_columns(b.PIC_WORK, self.x0 - HALO, HALO, everyone[r - 1][:, n:])      # the left neighbour's right edge
            if r < self.world - 1:
                b.put_columns(b.PIC_WORK, self.x1, HALO, everyone[r + 1][:, :n])             # the right neighbour's left edge
        b.filter(jobs)
        if self.world > 1:
            strip = b.new_buffer(b.columns_bytes(self.max_w))                                # padded to the widest strip
            b.get_columns(dst_slot, self.x0, self.x1 - self.x0, strip[:, :b.columns_bytes(self.x1 - self.x0)])
            everyone = self.all_gather(strip)
            for q, (a, e) in enumerate(self.bounds):
                if q != r:
                    b.put_columns(dst_slot, a, e - a, everyone[q][:, :b.columns_bytes(e - a)])


class GpuStripBackend:
    """TileSplit compute on a ReconContext (nchains streams, every one split the same way)"""
    PIC_WORK = -1

    def __init__(self, ctx):
        import torch
        self.ctx, self.torch = ctx, torch

    def make_job(self, frame, x0, x1):
        from . import video
        return video.Job.strip(self.ctx, frame, x0, x1)

    def recon(self, jobs):
        self.ctx.submit_phase(jobs, 1)

    def filter(self, jobs):
        self.ctx.submit_phase(jobs, 2)

    def columns_bytes(self, w):
        return self.ctx.columns_bytes(w)

    def new_buffer(self, nbytes):
        return self.torch.empty((self.ctx.nchains, nbytes), dtype=self.torch.uint8, device=self.ctx.device)

    def get_columns(self, pic, x, w, out):
        for c in range(self.ctx.nchains):
            row = out[c]
            if not row.is_contiguous() or row.data_ptr() % 4:
                tmp = self.torch.empty(row.numel(), dtype=self.torch.uint8, device=row.device)
                self.ctx.get_columns(pic, x, w, tmp, chain=c)
                row.copy_(tmp)
            else:
                self.ctx.get_columns(pic, x, w, row, chain=c)

    def put_columns(self, pic, x, w, buf):
        for c in range(self.ctx.nchains):
            row = buf[c]
            if not row.is_contiguous() or row.data_ptr() % 4:
                row = row.contiguous().clone()
            self.ctx.put_columns(pic, x, w, row, chain=c)


def torch_all_gather(group=None, stage_through_host=False):
    """all_gather for TileSplit on torch.distributed: RCCL on device tensors, or (gloo: tests / rehearsal with several
    ranks on one GPU) staged through host memory"""
    import torch
    import torch.distributed as dist

    def gather(t):
        world = dist.get_world_size(group)
        if stage_through_host:
            h = t.cpu()
            outs = [torch.empty_like(h) for _ in range(world)]
            dist.all_gather(outs, h, group=group)
            return [o.to(t.device) for o in outs]
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return [out[q] for q in range(world)]
    return gather


# ---------------------------------------------------------------------------------------------------------------- the native driver
GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
RCCL_UNIQUE_ID_BYTES = 128


class _DevBytes:
    """a raw device pointer as something torch.as_tensor() understands (__cuda_array_interface__)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, device):
    import torch
    return torch.as_tensor(_DevBytes(ptr, nbytes), device=device)


def rccl_unique_id():
    """rank 0: the 128 bytes every rank passes to NativeTileSplit(..., rccl_id=...) (ea1_rccl_unique_id = ncclGetUniqueId)"""
    from . import _lib
    buf = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
    L = _lib.lib()
    L.ea1_rccl_unique_id.argtypes = [ctypes.c_void_p]
    _lib.check(L.ea1_rccl_unique_id(buf), "ea1_rccl_unique_id")
    return bytes(buf)


class NativeTileSplit:
    """ea1_split_* (include/easyav1_amd_split.h): the per-frame sequence of TileSplit above inside the library, with its collective --
    RCCL when `rccl_id` is given (the communicator is the split's own: ncclCommInitRank from the 128 bytes of rank 0's rccl_unique_id()),
    else `all_gather(send, recv_list)`: a Python transport for tests and rehearsals (cuda uint8 tensor of this rank's bytes -> list of `world`
    cuda uint8 tensors to fill, the own one included)."""

    def __init__(self, ctx, bounds, rank, all_gather=None, rccl_id=None):
        from . import _lib
        self.ctx, self.bounds, self.rank, self.world = ctx, bounds, rank, len(bounds)
        self.x0, self.x1 = bounds[rank]
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        xb = (ctypes.c_uint32 * (self.world + 1))(*([b[0] for b in bounds] + [bounds[-1][1]]))
        if rccl_id is not None:
            L.ea1_split_create_rccl.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
            _lib.check(L.ea1_split_create_rccl(ctypes.byref(self._h), ctx._h, xb, self.world, rank, rccl_id), "ea1_split_create_rccl")
            self._cb = None
        else:
            device, world = ctx.device, self.world

            def cb(_ud, d_send, d_recv, nbytes, _stream):
                try:
                    send = device_bytes(d_send, nbytes, device)
                    recv = device_bytes(d_recv, nbytes * world, device)
                    all_gather(send, [recv[q * nbytes:(q + 1) * nbytes] for q in range(world)])
                    return 0
                except BaseException:   # noqa: BLE001  (an exception must not unwind through the C caller)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = GATHER_FN(cb) if (all_gather is not None or self.world > 1) else None
            L.ea1_split_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            _lib.check(L.ea1_split_create(ctypes.byref(self._h), ctx._h, xb, self.world, rank, self._cb if all_gather is not None else None, None), "ea1_split_create")

    def make_job(self, frame):
        from . import _lib, ir, video
        job = video.Job.__new__(video.Job)
        job._h = ctypes.c_void_p()
        job.desc = frame.desc
        f = ir.ir_frame_struct(frame)
        L = _lib.lib()
        L.ea1_split_job_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(L.ea1_split_job_create(self._h, ctypes.byref(f), ctypes.byref(job._h)), "ea1_split_job_create")
        return job

    def decode(self, jobs, dst_slot):
        from . import _lib, video
        assert len(jobs) == self.ctx.nchains
        arr = (ctypes.c_void_p * len(jobs))(*[j._h if j is not None else None for j in jobs])
        L = _lib.lib()
        L.ea1_split_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib.check(L.ea1_split_decode(self._h, arr, int(dst_slot), video._stream_ptr(self.ctx.device)), "ea1_split_decode")

    def traffic(self):
        from . import _lib
        v = [ctypes.c_size_t() for _ in range(4)]
        L = _lib.lib()
        L.ea1_split_traffic.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_size_t)] * 4
        L.ea1_split_traffic.restype = None
        L.ea1_split_traffic(self._h, *[ctypes.byref(x) for x in v])
        return dict(zip(("halo_sent", "halo_received", "strip_sent", "strip_received"), [x.value for x in v]))

    def close(self):
        from . import _lib
        if self._h:
            L = _lib.lib()
            L.ea1_split_destroy.argtypes = [ctypes.c_void_p]
            L.ea1_split_destroy.restype = None
            L.ea1_split_destroy(self._h)
            self._h = ctypes.c_void_p()
