"""Single-stream rate through libeasyav1.so (tools/bin/easyav1_bench on a generated file) against the sink's knobs:
GOP chains in flight (EASYAV1_AMD_GOPS), helper threads (EASYAV1_AMD_THREADS).   python tools/api_sweep.py 1080p|4k [repeats] [skip]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_stream  # noqa: E402
import easyav1_amd  # noqa: E402
easyav1_amd.load(build=False)
which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 20
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 240
W, H, tc = make_stream.SIZES[which]
tool = os.path.join(ROOT, "tools", "bin", "easyav1_bench")
tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
with tempfile.TemporaryDirectory(dir=tmpdir) as tmp:
    path = os.path.join(tmp, "s.webm")
    n, nbytes = make_stream.write_parsed_stream(path, W, H, tc, 60, repeats, 1)
    print(f"{which}: {n} frames, {nbytes / 1e6:.0f} MB; the first {skip} displayed frames are not counted", flush=True)
    configs = ((1, 8, {}), (2, 8, {}), (4, 8, {}), (8, 8, {}), (4, 16, {}), (4, 4, {}), (4, 8, {"EASYAV1_AMD_STAGE_THREADS": "3"}),
                             (8, 8, {"EASYAV1_AMD_STAGE_THREADS": "3"}), (4, 8, {"EASYAV1_AMD_PREPARE_AHEAD": "12"}), (8, 8, {"EASYAV1_AMD_PREPARE_AHEAD": "12"}),
                             (8, 8, {"EASYAV1_AMD_PREPARE_AHEAD": "3"}))
    if len(sys.argv) > 4:   # e.g. "8:8,12:12,16:16": chains:threads pairs instead of the standard list
        configs = tuple((int(a.split(":")[0]), int(a.split(":")[1]), {}) for a in sys.argv[4].split(","))
    for gops, thr, extra in configs:
        env = dict(os.environ, EASYAV1_AMD_GOPS=str(gops), EASYAV1_AMD_THREADS=str(thr), **extra,
                   LD_LIBRARY_PATH=os.path.join(ROOT, "easyav1_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([tool, path, "--json", "--skip", str(skip)], capture_output=True, text=True, env=env, timeout=600)
        if r.returncode:
            print(f"GOPS={gops} THREADS={thr}: exit code {r.returncode} {r.stderr[-300:]}")
            continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"GOPS={gops} THREADS={thr:2d} {' '.join(k[12:] + '=' + v for k, v in extra.items()):18s}: {j['fps_wall_all']:8.1f} frames/s over the whole file ({j['frames_all']} frames, start-up included), {j['fps_wall']:8.1f} behind the first {skip} "
              f"(slowest frame {j['slowest_ms']:.2f} ms, open {j['open_ms']:.0f} ms)", flush=True)
