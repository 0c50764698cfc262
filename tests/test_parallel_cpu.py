"""N > 1 path on CPU: the GOP-chain sharding used by bench.py, exercised with world_size-2 gloo
process groups (the decoder itself is replaced by the CPU oracle here -- test infrastructure)."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chain_partition_is_disjoint_and_complete():
    from easyav1_amd import parallel
    for world in (1, 2, 4, 8):
        for cpr in (1, 3, 32):
            ids = [parallel.chain_ids(r, world, cpr) for r in range(world)]
            flat = [i for part in ids for i in part]
            assert sorted(flat) == list(range(world * cpr))
            for g in range(3 * world * cpr):
                r, c = parallel.owner_of_gop(g, world, cpr)
                assert ids[r][c] == g % (world * cpr)
    with pytest.raises(ValueError):
        parallel.chain_ids(2, 2, 1)
    # seeds depend on the global chain id only: 2 ranks x 2 chains == 1 rank x 4 chains
    a = [parallel.chain_seed(7, i) for r in range(2) for i in parallel.chain_ids(r, 2, 2)]
    b = [parallel.chain_seed(7, i) for i in parallel.chain_ids(0, 1, 4)]
    assert a == b and len(set(a)) == 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, cpr, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    from easyav1_amd import ir, parallel
    from tests import oracle_lib as ol
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert parallel.dist_env()[:2] == (rank, world)
    digests = {}
    for cid in parallel.chain_ids(rank, world, cpr):
        p = ir.synth_params(64, 64, 1, seed=parallel.chain_seed(5, cid), gop=2)
        refs = [None] * 8
        h = hashlib.sha256()
        for fi in range(2):
            f = ir.synth_frame(p, fi)
            out = ol.oracle_decode_frame(f, refs)
            refs[f.desc.dst_slot] = out["final"]
            for pl in out["final"]:
                h.update(pl.tobytes())
        digests[cid] = h.hexdigest()
    parallel.barrier()
    # bench.py's timing rule: max over ranks
    t = parallel.global_max(1.0 + rank)
    assert t == float(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        np.save(os.path.join(out_dir, "digests.npy"), np.array(sorted(merged.items()), dtype=object), allow_pickle=True)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    world, cpr = 2, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, cpr, str(tmp_path)), nprocs=world, join=True)
    got = dict(np.load(os.path.join(tmp_path, "digests.npy"), allow_pickle=True).tolist())
    # single-process decode of all world*cpr chains
    sys.path.insert(0, ROOT)
    from easyav1_amd import ir, parallel
    from tests import oracle_lib as ol
    assert sorted(got) == list(range(world * cpr))
    for cid in range(world * cpr):
        p = ir.synth_params(64, 64, 1, seed=parallel.chain_seed(5, cid), gop=2)
        refs = [None] * 8
        h = hashlib.sha256()
        for fi in range(2):
            f = ir.synth_frame(p, fi)
            out = ol.oracle_decode_frame(f, refs)
            refs[f.desc.dst_slot] = out["final"]
            for pl in out["final"]:
                h.update(pl.tobytes())
        assert got[cid] == h.hexdigest()
