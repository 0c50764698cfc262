"""Round-5 robustness sweep (verdict item 5): the bench content at other qualities and GOP lengths -- dav1d 1.5.3 (all cores it sees, and 16 threads) beside
this repository's path through libeasyav1.so, one table.  Streams: tests/golden/av1_stream_bench_uhd_4cols_gop16.npz + tools/bin/sweep/*.npz
(tools/make_sweep_streams.py).      python tools/real_stream_sweep.py > gpurun_out/<name>.txt"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

streams = [os.path.join(ROOT, "tests", "golden", "av1_stream_bench_uhd_4cols_gop16.npz")] + sorted(glob.glob(os.path.join(ROOT, "tools", "bin", "sweep", "*.npz")))
only = sys.argv[1:]
print("%-34s %9s %7s %7s | %8s %8s %8s | %6s %6s" % ("stream", "kbit/frm", "dav1d", "dav1d16", "ours", "steady", "frames", "x all", "x 16t"), flush=True)
for path in streams:
    name = os.path.basename(path)[len("av1_stream_"):-4]
    if only and not any(o in name for o in only):
        continue
    import numpy as np
    z = np.load(path, allow_pickle=False)
    gop = len(z["unit_sizes"])
    bits = 8 * int(z["unit_sizes"].sum()) // gop
    repeats = max(24, 3072 // gop)    # ~3 000 frames each; at least 24 GOPs so that every chain gets several
    r = bench.real_av1_pass(path, repeats_override=repeats, dav1d_seconds=8.0)
    d, d16, o = r.get("dav1d") or {}, r.get("dav1d_16_threads") or {}, r.get("ours") or {}
    if "error" in o:
        print(name, "FAILED", o["error"][-300:], flush=True)
        continue
    print("%-34s %9.1f %7.1f %7.1f | %8.1f %8.1f %8d | %6.2f %6.2f" % (name + " GOP %d" % gop, bits / 1e3, d.get("value", 0), d16.get("value", 0), o.get("value", 0), o.get("steady_state", 0),
                                                                   o.get("frames", 0), o.get("value", 0) / max(d.get("value", 1e-9), 1e-9), o.get("value", 0) / max(d16.get("value", 1e-9), 1e-9)), flush=True)
    print("    " + json.dumps({k: r[k] for k in r if k in ("dav1d_unavailable", "dav1d_16_threads_unavailable")}), flush=True) if (not d or not d16) else None
