"""One-process-per-GPU sharding of independent GOP chains (closed GOPs are
independent units, so the path shards with NO data-path collective; torch.distributed is used
only for the start/stop barrier and the max-over-ranks timing that bench.py reports)."""
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
