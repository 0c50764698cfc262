"""Loader / writer of the video fixture layout documented in tests/golden/README.md: parsed frames + expected
pictures as plain arrays, so that vectors dumped from a real AV1 decoder pin the oracle and the HIP path the day they
exist.  numpy.load(allow_pickle=False): nothing from a fixture file is ever executed."""
import ctypes

import numpy as np

from easyav1_amd import ir

FORMAT = 1
STAGES = ("recon", "deblock", "cdef")


def save_fixture(path, width, height, frames, outputs, ref_slots=None, warp_filter=None, stages=None):
    """frames: [ParsedFrame]; outputs: [[y, u, v]] expected finished pictures; stages: optional [{name: [y, u, v]}]"""
    a = {"format": np.array([FORMAT], np.uint32), "size": np.array([width, height], np.uint32)}
    if warp_filter is not None:
        a["warp_filter"] = np.ascontiguousarray(warp_filter, np.int8)
    if ref_slots:
        a["ref_slots"] = np.array(sorted(ref_slots), np.uint8)
        for s, planes in ref_slots.items():
            for n, p in zip("yuv", planes):
                a[f"ref_{s}_{n}"] = np.ascontiguousarray(p, np.uint8)
    for i, (f, out) in enumerate(zip(frames, outputs)):
        a[f"f{i}_desc"] = np.frombuffer(bytes(f.desc), np.uint8).copy()
        a[f"f{i}_blocks"] = f.blocks.view(np.uint8).reshape(len(f.blocks), -1)
        a[f"f{i}_txs"] = f.txs.view(np.uint8).reshape(len(f.txs), -1)
        a[f"f{i}_coefs"] = f.coefs
        a[f"f{i}_aux"] = f.aux
        a[f"f{i}_cdef_idx"] = f.cdef_idx
        for p in range(3):
            if len(f.lr[p]):
                a[f"f{i}_lr{p}"] = f.lr[p].view(np.uint8).reshape(len(f.lr[p]), -1)
        for n, pl in zip("yuv", out):
            a[f"f{i}_out_{n}"] = np.ascontiguousarray(pl, np.uint8)
        if stages and stages[i]:
            for st, planes in stages[i].items():
                for n, pl in zip("yuv", planes):
                    a[f"f{i}_stage_{st}_{n}"] = np.ascontiguousarray(pl, np.uint8)
    np.savez_compressed(path, **a)


def load_fixture(path):
    """-> dict(width, height, warp_filter or None, refs: {slot: [y, u, v]}, frames: [(ParsedFrame, expected [y, u, v], {stage: [y, u, v]})])"""
    z = np.load(path, allow_pickle=False)
    assert int(z["format"][0]) == FORMAT, "unknown fixture format"
    W, H = (int(v) for v in z["size"])
    refs = {}
    if "ref_slots" in z:
        for s in z["ref_slots"]:
            refs[int(s)] = [z[f"ref_{int(s)}_{n}"] for n in "yuv"]
    frames = []
    i = 0
    while f"f{i}_desc" in z:
        raw = z[f"f{i}_desc"].tobytes()
        assert len(raw) == ctypes.sizeof(ir.FrameDesc), "ea1_frame_desc size differs from this build"
        desc = ir.FrameDesc.from_buffer_copy(raw)
        assert desc.width <= W and desc.height <= H, "a frame larger than the fixture's `size` (the largest coded size of the stream)"
        blocks = np.ascontiguousarray(z[f"f{i}_blocks"]).reshape(-1).view(ir.BLOCK_DTYPE)
        txs = np.ascontiguousarray(z[f"f{i}_txs"]).reshape(-1).view(ir.TXBLOCK_DTYPE)
        lr = [np.ascontiguousarray(z[f"f{i}_lr{p}"]).reshape(-1).view(ir.LR_UNIT_DTYPE) if f"f{i}_lr{p}" in z else np.zeros(0, ir.LR_UNIT_DTYPE) for p in range(3)]
        f = ir.ParsedFrame(desc, blocks, txs, z[f"f{i}_coefs"], z[f"f{i}_aux"], z[f"f{i}_cdef_idx"], lr)
        out = [z[f"f{i}_out_{n}"] for n in "yuv"]
        st = {s: [z[f"f{i}_stage_{s}_{n}"] for n in "yuv"] for s in STAGES if f"f{i}_stage_{s}_y" in z}
        frames.append((f, out, st))
        i += 1
    return dict(width=W, height=H, warp_filter=z["warp_filter"] if "warp_filter" in z else None, refs=refs, frames=frames)


def check_with_oracle(fx):
    """every frame of a loaded fixture through the CPU oracle; raises AssertionError on the first differing sample"""
    from tests import oracle_lib as ol
    if fx["warp_filter"] is not None:
        ol.oracle_set_warp_filter(fx["warp_filter"])
    refs = [fx["refs"].get(s) for s in range(8)]
    for k, (f, want, stages) in enumerate(fx["frames"]):
        got = ol.oracle_decode_frame(f, refs)
        for pl in range(3):
            assert np.array_equal(got["final"][pl], want[pl]), f"oracle: frame {k} plane {pl} differs from the fixture"
            for s, planes in stages.items():
                assert np.array_equal(got[s][pl], planes[pl]), f"oracle: frame {k} stage {s} plane {pl}"
        refs[f.desc.dst_slot] = want        # continue from the fixture's own pictures


def check_with_gpu(fx):
    """the same through libeasyav1_amd.so"""
    from easyav1_amd import video
    stage_mask = {"recon": ir.STAGE_RECON, "deblock": ir.STAGE_RECON | ir.STAGE_DEBLOCK, "cdef": ir.STAGE_RECON | ir.STAGE_DEBLOCK | ir.STAGE_CDEF}
    ctx = video.ReconContext(fx["width"], fx["height"])
    if fx["warp_filter"] is not None:
        ctx.set_warp_filter(fx["warp_filter"])
    for s, planes in fx["refs"].items():
        ctx.set_ref(s, planes)
    for k, (f, want, stages) in enumerate(fx["frames"]):
        job = ctx.make_job(f)
        for s, planes in stages.items():      # intermediate pictures: partial runs into the destination slot
            ctx.submit(job, stage_mask[s])
            got = ctx.download(f.desc.dst_slot)
            for pl in range(3):
                assert np.array_equal(got[pl], planes[pl]), f"gpu: frame {k} stage {s} plane {pl}"
        ctx.submit(job)
        got = ctx.download(f.desc.dst_slot)
        for pl in range(3):
            assert np.array_equal(got[pl], want[pl]), f"gpu: frame {k} plane {pl} differs from the fixture"
        ctx.set_ref(f.desc.dst_slot, want)
        job.close()
    ctx.close()
