#!/usr/bin/env python3
"""Headline benchmark: decoded frames/s of the MI355X AV1 reconstruction path.

Metric (BASELINE.json): decoded frames/sec, 8-bit 4:2:0, YUV bit-exact (parity is asserted by
tests/, not here).  Workload at N=1: configs[1] = 1920x1080, 1 tile column, seeded synthetic
parsed-frame streams (GOP 60: 1 key + 59 inter frames), reconstruction + deblock + CDEF + loop
restoration, everything (parsed frames, references) resident in HBM before the timed region.

A "step" = every decoding chain of this rank reconstructs one whole GOP.  Each rank owns
`--chains` independent GOP chains (closed GOPs are independent units) inside one
batched context: every kernel launch serves one frame of every chain.  Ranks never exchange data
(no collective on the data path) -> weak scaling.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, algorithmic bytes / measured
launch time, HIP events on the launch stream) and `cpu_baseline` (the oracle timed on host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

AUDIO_WORKLOAD = dict(channels=6, bs0=256, bs1=2048, seconds=60, rate=48000,
                      name="Vorbis synthesis, 48 kHz 5.1, blocksizes 256/2048, 60 s per stream (BASELINE configs[4] audio part)")

WORKLOADS = {
    "1080p": dict(width=1920, height=1080, tile_cols=1, name="1920x1080 8-bit yuv420, 1 tile column, GOP 60 (BASELINE configs[1])"),
    "4k": dict(width=3840, height=2160, tile_cols=4, name="3840x2160 8-bit yuv420, 4 tile columns, GOP 60 (BASELINE configs[2])"),
    "360p": dict(width=640, height=360, tile_cols=1, name="640x360 8-bit yuv420, 1 tile (BASELINE configs[0] video part)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS) + ["audio"])
    ap.add_argument("--chains", type=int, default=0, help="independent GOP chains per GPU, batched into every launch (0 = per-workload default)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="batched contexts per GPU, each on its own HIP stream (2 x 64 chains overlap complementary kernels and "
                         "give ~7 %% more frames/s than 1 x 128, but per-kernel durations then include the sharing: the default "
                         "keeps one context so that the roofline entry is the kernel's own time and agrees with rocprofv3)")
    ap.add_argument("--gop", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real thing) or gloo: rehearses the multi-rank code path with "
                    "several ranks sharing one GPU (ranks are then mapped to devices modulo the device count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-d2h", action="store_true", help="skip the PCIe-inclusive pass (reported beside `value`, never as `value`)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames per thread of the CPU sample (0 = sized for ~15 s)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (box share for one GPU: 16)")
    return ap.parse_args()


def cpu_baseline(wl, gop, seed, nframes, threads):
    """The oracle (oracle/av1_recon.c, a literal C restatement -- NOT dav1d) timed on this box's
    host cores: `threads` independent chains (the same sharding the GPU path uses), each decoding the
    first `nframes` frames of its stream in its own thread (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from easyav1_amd import ir, parallel
    from tests import oracle_lib as ol
    ol.oracle_decode_frame(ir.synth_frame(ir.synth_params(64, 64, 1, seed=1), 0), None)  # load + warm the library
    streams = []
    for t in range(threads):
        p = ir.synth_params(wl["width"], wl["height"], wl["tile_cols"], seed=parallel.chain_seed(seed, t), gop=gop)
        streams.append([ir.synth_frame(p, fi) for fi in range(nframes)])

    def run(frames):
        refs = [None] * 8
        for f in frames:
            out = ol.oracle_decode_frame(f, refs)
            refs[f.desc.dst_slot] = out["final"]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, streams))
    dt = time.perf_counter() - t0
    return threads * nframes / dt, dt


def audio_main(args):
    """Secondary workload: IMDCT + window + overlap-add of 60 s of 5.1 audio per stream.  A step =
    every stream of the rank decodes its 60 s (one launch per stream); value = transforms/s
    (one transform = one (block, channel) IMDCT; 4*N algorithmic bytes each)."""
    import numpy as np
    import torch
    import easyav1_amd
    easyav1_amd.load()
    from easyav1_amd import audio
    from tests import oracle_lib as ol
    wl = AUDIO_WORKLOAD
    ch, bs0, bs1 = wl["channels"], wl["bs0"], wl["bs1"]
    torch.cuda.set_device(0)
    rng = np.random.default_rng(args.seed)
    # block-type sequence: 85 % long, shorts in runs of 8
    flags = []
    samples = 0
    while samples < wl["seconds"] * wl["rate"]:
        run = [0] * 8 if flags and rng.random() < 0.0216 else [1]   # ~15 % of the blocks are short, in runs of 8
        for W in run:
            if flags:
                samples += (bs0, bs1)[flags[-1]] // 4 + (bs0, bs1)[W] // 4
            flags.append(W)
    flags = np.array(flags, np.uint8)
    nb = len(flags)
    nstreams = args.chains or 16
    env = 1.0 / (1.0 + torch.arange(bs1 // 2, device="cuda") / 8.0)
    spectra = [torch.randn((nb, ch, bs1 // 2), device="cuda") * env for _ in range(nstreams)]
    for sp in spectra:
        sp[:, :, int(0.8 * bs1 // 2):] = 0
    synths = [audio.VorbisSynth(ch, bs0, bs1) for _ in range(nstreams)]
    total = synths[0].count(flags)
    outs = [torch.empty((ch, total), dtype=torch.float32, device="cuda") for _ in range(nstreams)]

    def step():
        for s_, sp, o in zip(synths, spectra, outs):
            s_.restart()
            s_.decode(sp, flags, out=o)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / (args.steps * nstreams)   # one decode kernel per stream and step
    transforms = nb * ch
    alg_bytes = sum(4 * (bs0, bs1)[W] for W in flags.tolist()) * ch
    value = transforms * nstreams * args.steps / dt
    cpu = None
    if not args.no_cpu_baseline:
        m = {0: ol.OracleMdct(bs0), 1: ol.OracleMdct(bs1)}
        nsample = nb
        sp = spectra[0][:nsample].cpu().numpy()
        c0 = time.perf_counter()
        blocks = [m[int(W)].backward(sp[k, :, : (bs0, bs1)[int(W)] // 2]) for k, W in enumerate(flags[:nsample])]
        ol.oracle_synth_stream(ch, bs0, bs1, [int(v) for v in flags[:nsample]], blocks)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(nsample * ch / cdt, 1), "unit": "transforms/s", "cores": 1, "kind": "port",
               "sample": f"first {nsample} blocks x {ch} channels of one stream, {cdt:.1f} s, oracle/vorbis_synth.c through ctypes (one call per transform)"}
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "Vorbis synthesis transforms/sec (IMDCT + window + overlap-add, sample-exact)", "value": round(value, 1),
        "unit": "transforms/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "streams": nstreams, "blocks_per_stream": nb, "pcm_samples_per_channel": int(total)},
        "roofline": {"bound": "hbm", "kernel": "synth_decode_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_us": round(kern_ms * 1e3, 2),
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "note": "launch time from torch events on the launch stream around the decode calls (one kernel each)"},
        "cpu_baseline": cpu}), flush=True)


def main():
    args = parse_args()
    if args.workload == "audio":
        return audio_main(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.dist_backend != "nccl":   # rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if distributed:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    n_gpus = world
    assert n_gpus == args.gpus or not distributed, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    import easyav1_amd
    easyav1_amd.load()  # raises when the HIP library is missing: no fallback
    from easyav1_amd import ir, parallel, video

    wl = WORKLOADS[args.workload]
    W, H = wl["width"], wl["height"]
    # chains per launch: as many as the context allows at 1080p; 4K frames are 4x the work each
    S = args.chains or {"1080p": 128, "4k": 32, "360p": 128}[args.workload]
    G = args.gop

    # ---- build the inputs: NC contexts x S independent GOP chains per rank, parsed frames resident in HBM ----
    NC = args.contexts
    ctxs, streams, jobs = [], [], []  # jobs[context][chain][frame]
    alg_bytes_gop = 0
    coef_total = 0
    for k in range(NC):
        ctx = video.ReconContext(W, H, local_rank, nchains=S)
        cj = []
        for s in range(S):
            seed = parallel.chain_seed(args.seed, parallel.chain_ids(rank, world, NC * S)[k * S + s])
            p = ir.synth_params(W, H, wl["tile_cols"], seed=seed, gop=G)
            chain = []
            for fi in range(G):
                job = ctx.make_job(ir.synth_frame(p, fi))
                if s == 0 and k == 0:
                    st = job.stats()
                    alg_bytes_gop += st["algorithmic_bytes"]
                    coef_total += st["coef_count"]
                chain.append(job)
            cj.append(chain)
        ctxs.append(ctx)
        jobs.append(cj)
        streams.append(torch.cuda.Stream(device=local_rank))
    ctx = ctxs[0]

    def step():
        for fi in range(G):
            for k in range(NC):
                with torch.cuda.stream(streams[k]):
                    ctxs[k].submit_batch([jobs[k][s][fi] for s in range(S)])

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_total = n_gpus * NC * S * G * args.steps
    fps = frames_total / dt
    ms_per_step = dt / args.steps * 1e3

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one more pass of the same step (one GOP of every chain of
        # context 0; with the default single context that IS the timed workload), every kernel launch
        # bracketed by HIP events on the launch stream.  Kept out of the timed region because the
        # event records would perturb `value`. ----
        ctx.profile(True)
        with torch.cuda.stream(streams[0]):
            for fi in range(G):
                ctx.submit_batch([jobs[0][s][fi] for s in range(S)])
        prof = ctx.profile_read()
        ctx.profile(False)
        prof.sort(key=lambda e: -e["ms_total"])
        total_ms = sum(e["ms_total"] for e in prof)
        dom = prof[0]
        avg_s = dom["ms_total"] / dom["launches"] * 1e-3
        bytes_per_launch = dom["algorithmic_bytes"] / dom["launches"]
        achieved = bytes_per_launch / avg_s / 1e9
        roofline = {
            "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "share_of_gpu_time": round(dom["ms_total"] / total_ms, 3),
            "kernels": [{"name": e["name"], "launches": e["launches"], "avg_us": round(e["ms_total"] / e["launches"] * 1e3, 2),
                         "GBps": round(e["algorithmic_bytes"] / max(e["ms_total"], 1e-9) / 1e6, 1)} for e in prof],
            "whole_path": {"algorithmic_bytes_per_frame": int(alg_bytes_gop / G),
                           "achieved_GBps": round(alg_bytes_gop / G * fps / n_gpus / 1e9, 2),
                           "frac": round(alg_bytes_gop / G * fps / n_gpus / 1e9 / HBM_PEAK_GBS, 5)},
        }
        # HBM-side traffic of the dominant kernel per launch: PMC counters cannot be read from inside
        # this process (rocprofv3 collects them in separate passes of this very command), so the
        # figure comes from the committed summary of those passes when it covers this configuration
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if NC == 1 and os.path.exists(tpath):
            for e in json.load(open(tpath))["entries"]:
                k = e["kernels"].get(dom["name"].replace(" ", ""))
                if e["workload"] == args.workload and e["chains_per_launch"] == S and k:
                    roofline["traffic"] = k["traffic_bytes"]
                    roofline["traffic_detail"] = {"fetch_size_raw_bytes": k["fetch_size_raw_bytes"], "write_size_bytes": k["write_size_bytes"],
                                                  "traffic_low_bytes": k["traffic_low_bytes"], "source": e["source"],
                                                  "correction": "2 * FETCH_SIZE (128-byte requests tallied at 64 B on gfx950) + WRITE_SIZE"}
        # ---- PCIe-inclusive rate (never `value`): the same step with every finished picture copied to
        # pinned host memory, one device-layout copy per chain and frame on a second stream that
        # trails the submits by events; submits wait for the copy of the picture whose slot they
        # are about to overwrite (8 reference slots).  Rank 0, N = 1 only.
        d2h = None
        if n_gpus == 1 and NC == 1 and not args.no_d2h:
            _, _, pic_bytes = ctx.picture_layout()
            bufs = [video.PinnedBuffer(S * pic_bytes) for _ in range(2)]
            copy_stream = torch.cuda.Stream(device=local_rank)
            ev_done = [torch.cuda.Event() for _ in range(G)]
            ev_copied = [torch.cuda.Event() for _ in range(G)]

            primed = [False]

            def step_d2h():
                for fi in range(G):
                    with torch.cuda.stream(streams[0]):
                        if fi >= 7 or primed[0]:   # the picture 8 frames back (previous step included) has left its slot
                            streams[0].wait_event(ev_copied[(fi - 7) % G])
                        ctx.submit_batch([jobs[0][s][fi] for s in range(S)])
                        ev_done[fi].record(streams[0])
                    copy_stream.wait_event(ev_done[fi])
                    base = bufs[fi & 1].ptr.value
                    ctx.download_batch_async([fi % 8] * S, [base + s * pic_bytes for s in range(S)], copy_stream.cuda_stream)
                    ev_copied[fi].record(copy_stream)
                primed[0] = True
            step_d2h()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_d2h()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            fps1 = S * G * args.steps / dt1
            d2h = {"value": round(fps1, 2), "unit": "frames/s", "GB_per_s_to_host": round(fps1 * pic_bytes / 1e9, 2),
                   "bytes_per_picture": int(pic_bytes), "note": "reconstruction + one device-layout D2H copy per picture into pinned host memory"}
            for b in bufs:
                b.close()
        cpu = None
        if not args.no_cpu_baseline and n_gpus == 1:   # rank 0 at N = 1 only
            nfr = args.cpu_frames or {"1080p": 48, "4k": 12, "360p": 240}[args.workload]
            nth = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            cfps, cdt = cpu_baseline(wl, G, args.seed, nfr, nth)
            cpu = {"value": round(cfps, 3), "unit": "frames/s", "cores": nth, "kind": "port",
                   "sample": f"{nth} independent streams x first {nfr} frames (1 key + {nfr - 1} inter each) of the same workload, "
                             f"{cdt:.1f} s wall, oracle/av1_recon.c one thread per stream ({os.cpu_count()} host cores visible); "
                             "literal C restatement, NOT dav1d (dav1d is absent from the reference tree)"}
        result = {
            "metric": "decoded frames/sec (easyav1_benchmark) 1080p/4K 8-bit AV1; YUV bit-exact",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "frames_per_step_per_gpu": NC * S * G, "chains_per_gpu": NC * S, "contexts_per_gpu": NC,
                       "chains_per_launch": S, "gop": G,
                       "stages": "inter-pred + itx + intra + deblock + CDEF + loop-restoration",
                       "input": "seeded parsed-frame IR resident in HBM (host entropy parse not included)",
                       "coded_coefficients_per_frame": int(coef_total / G)},
            "roofline": roofline, "cpu_baseline": cpu, "with_d2h": d2h,
        }
    sync_all()
    for cj in jobs:
        for chain in cj:
            for j in chain:
                j.close()
    for c in ctxs:
        c.close()
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
