"""the committed real 4K / 1080p GOPs through libeasyav1.so (tools/bin/easyav1_bench) under several thread settings: one table
    python tools/real_av1_sweep.py [uhd_4cols_gop16|hd_gop16] > gpurun_out/<name>.txt"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_stream  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "uhd_4cols_gop16"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 96
fixture = os.path.join(ROOT, "tests", "golden", "av1_stream_bench_%s.npz" % which)
tool = os.path.join(ROOT, "tools", "bin", "easyav1_bench")
variants = [dict(), dict(EASYAV1_AMD_TILE_THREADS="1"), dict(EASYAV1_AMD_TILE_THREADS="4", EASYAV1_AMD_THREADS="32"), dict(EASYAV1_AMD_THREADS="32", EASYAV1_AMD_TILE_THREADS="1"),
            dict(EASYAV1_AMD_GOPS="1"), dict(EASYAV1_AMD_GOPS="1", EASYAV1_AMD_TILE_THREADS="1"), dict(EASYAV1_AMD_GOPS="32", EASYAV1_AMD_THREADS="32"),
            dict(EASYAV1_AMD_GOPS="4"), dict(EASYAV1_AMD_GOPS="8", EASYAV1_AMD_THREADS="8")]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    path = os.path.join(tmp, "real.webm")
    n, nbytes, W, H, gop = make_stream.write_real_av1_stream(path, fixture, repeats)
    print("%s: %dx%d, GOP %d x %d, %d bytes" % (which, W, H, gop, repeats, nbytes), flush=True)
    for v in variants:
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "easyav1_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        env.update(v)
        r = subprocess.run([tool, path, "--json", "--repeat", "2", "--skip", str(16 * gop)], capture_output=True, text=True, env=env, timeout=900)
        if r.returncode:
            print(v, "FAILED", (r.stdout + r.stderr)[-300:], flush=True)
            continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-70s whole file %8.1f frames/s, behind the first %d frames %8.1f" % (json.dumps(v), j["fps_wall_all"], 16 * gop, j["fps_wall"]), flush=True)
