"""The reference's own tools/easyav1_benchmark.c (compiled in place against include/easyav1.h + libeasyav1.so into
oracle/_ref/easyav1_benchmark by oracle/Makefile) on generated files: ONE stream through the public interface --
file read, demux, parsed-frame packets, hand-over, reconstruction of one frame per launch, copy to host memory.
    python tools/run_reference_benchmark.py [1080p|4k] [frames]
Needs the GPU.  The number it prints is the BASELINE metric's own tool; it is a single-stream latency figure, not the
batched throughput of bench.py."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_easyav1_abi as abi  # noqa: E402  (file builders)

which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W, H = {"1080p": (1920, 1080), "4k": (3840, 2160), "360p": (640, 360)}[which]
tool = os.path.join(ROOT, "oracle", "_ref", "easyav1_benchmark")
assert os.path.exists(tool), "oracle/_ref/easyav1_benchmark is built in the container that has the reference tree (make -C oracle ref)"
import easyav1_amd  # noqa: E402
easyav1_amd.load(build=False)
libdir = os.path.join(ROOT, "easyav1_amd")
env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
# The tool's own clock counts whole milliseconds per displayed frame (tools/easyav1_benchmark.c:55-64, :124-125 of the
# reference): frames that take less than a millisecond add nothing, so its "fps average" is far too high here.  The
# honest single-stream figure is the slope of the PROCESS wall time between a short and a long file of the same stream.
walls = []
with tempfile.TemporaryDirectory() as tmp:
    for n in (nframes, 3 * nframes):
        t0 = time.perf_counter()
        data, frames = abi.parsed_frame_file(W, H, n, seed=1, frame_ms=17)
        path = os.path.join(tmp, "stream%d.webm" % n)
        with open(path, "wb") as f:
            f.write(data)
        print(f"{which}: {n} frames, file of {len(data) / 1e6:.1f} MB made in {time.perf_counter() - t0:.1f} s", flush=True)
        del data, frames
        best = None
        for rep in range(3):
            t1 = time.perf_counter()
            r = subprocess.run([tool, path], capture_output=True, text=True, env=env, timeout=900)
            wall = time.perf_counter() - t1
            if r.returncode:
                print("exit code", r.returncode, r.stdout[-1000:], r.stderr[-2000:])
                sys.exit(1)
            best = wall if best is None else min(best, wall)
        print(r.stdout.strip().splitlines()[-2])
        print(f"   process wall time (best of 3): {best * 1e3:.0f} ms")
        walls.append((n, best))
(n0, w0), (n1, w1) = walls
print(f"{which}: single stream through easyav1_decode_next / easyav1_get_video_frame: {(n1 - n0) / (w1 - w0):.0f} frames/s "
      f"({(w1 - w0) / (n1 - n0) * 1e3:.2f} ms per frame; start-up {(w0 - n0 * (w1 - w0) / (n1 - n0)) * 1e3:.0f} ms)")
