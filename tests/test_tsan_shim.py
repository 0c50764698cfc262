"""ThreadSanitizer on the host shim (the reference builds Debug with -fsanitize=thread,undefined:
/root/reference/CMakeLists.txt:50-58).  tools/tsan/run.sh compiles easyav1_api.cpp, easyav1_video_sink.cpp and every parser with
-fsanitize=thread against a device-less stand-in for the GPU half (tools/tsan/device_stub.cpp: grey pictures) and runs the player
patterns of tools/easyav1_player.c on real plain-AV1 WebM files: decode loop, play + polling getters, a seek storm while playing,
update_settings while playing, destroy while playing.  Zero reports, or the test fails.  (Round 4: the first run found Item::discard
written outside the sink's lock.)"""
import os
import shutil
import subprocess

import pytest

from tests.test_real_av1 import load_fixture, mux_webm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("name", ["long_136x104", "pyramid_200x136", "tiles_4cols_640x360"])
def test_the_host_shim_under_threadsanitizer(tmp_path, name):
    fx = load_fixture(os.path.join(ROOT, "tests", "golden", "av1_stream_%s.npz" % name))
    path = tmp_path / "stream.webm"
    path.write_bytes(mux_webm(fx))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    env.pop("EASYAV1_AMD_AV1_TABLES", None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan", "run.sh"), str(path)], capture_output=True, text=True, env=env, timeout=900)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
    assert r.returncode == 0 and "all scenarios done" in out, out[-3000:]
