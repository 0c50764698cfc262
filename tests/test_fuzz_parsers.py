"""Short run of the sanitizer fuzzer over the host-side parsers of untrusted input (tools/fuzz/):
ASan + UBSan build on the CPU, seed corpus from the test-side writers, a few thousand mutated inputs.
The long run (2.28 million inputs, clean) is quoted in DESIGN.md."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
def test_parsers_survive_mutated_inputs_under_sanitizers(tmp_path):
    import easyav1_amd
    easyav1_amd.load(build=True)
    corpus = str(tmp_path / "corpus")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fuzz", "make_corpus.py"), corpus], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz", "run.sh"), corpus, "40"], env=env, capture_output=True, text=True, timeout=600)
    if "cannot find -lasan" in out.stderr or "cannot find -lubsan" in out.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert out.returncode == 0 and "no sanitizer report" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
def test_job_creation_survives_corrupted_frames_under_sanitizers(tmp_path):
    """the host part of ea1_job_create (validation + schedule derivation) on corrupted frames of the synthetic
    generator, ASan + UBSan; the long runs (1.5 million frames on the CPU, 200 000 through the GPU) are in DESIGN.md"""
    import easyav1_amd
    easyav1_amd.load(build=True)
    env = dict(os.environ, TMPDIR=str(tmp_path))
    # (round 5: frames of generated 4:4:4 / 4:2:2 streams beside the synthetic 4:2:0 ones, each in a context of its layout and of the wrong one)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fuzz", "make_layout_seeds.py"), str(tmp_path)], stdout=subprocess.DEVNULL)
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz", "run.sh"), str(tmp_path), "400", "job"], env=env, capture_output=True, text=True,
                         timeout=900)
    if "cannot find -lasan" in out.stderr or "cannot find -lubsan" in out.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert out.returncode == 0 and "no sanitizer report" in out.stdout and "4:4:4 / 4:2:2 contexts" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
def test_block_parser_survives_mutated_streams_under_sanitizers(tmp_path):
    """the block-level AV1 parser (symbol decoder, syntax, contexts, frame driver) on mutated generated streams -- frame headers and tile
    bytes alike -- and mutated table blobs, ASan + UBSan; the long run is in DESIGN.md"""
    import easyav1_amd
    easyav1_amd.load(build=True)
    corpus = str(tmp_path / "av1_corpus")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fuzz", "make_av1_corpus.py"), corpus], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz", "run.sh"), corpus, "25", "av1"], env=env, capture_output=True, text=True, timeout=900)
    if "cannot find -lasan" in out.stderr or "cannot find -lubsan" in out.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert out.returncode == 0 and "mutated runs" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
