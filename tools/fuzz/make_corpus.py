"""Seed corpus for tools/fuzz/fuzz_parsers.cpp, made with the test-side writers (there are no media
files in this environment):  python tools/fuzz/make_corpus.py /tmp/ea1_corpus"""
import ctypes
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from easyav1_amd import obu  # noqa: E402
from tests import av1_frame_header_writer as fw  # noqa: E402
from tests import av1_obu_writer as wr  # noqa: E402
from tests import test_frame_header as tfh  # noqa: E402
from tests import test_webm_reader as twr  # noqa: E402
from tests import vorbis_stream_gen as vgen  # noqa: E402

out = sys.argv[1]
for d in ("seq", "frame", "webm", "vorbis"):
    os.makedirs(os.path.join(out, d), exist_ok=True)
rng = np.random.default_rng(1)
for i in range(40):      # temporal units with a sequence header
    h = wr.random_sequence_header(rng)
    data = wr.obu(2, b"") + wr.obu(1, wr.write_sequence_header(h)) + wr.obu(6, bytes(rng.integers(0, 256, 30, dtype=np.uint8)))
    open(os.path.join(out, "seq", f"{i:03d}.bin"), "wb").write(data)
n = 0
for stream in range(25):  # frame headers, each with the sequence header struct it belongs to and a tile group behind it
    seq = tfh.make_seq(rng)
    seq_bytes = bytes(obu.sequence_header_struct(seq))
    slots = fw.Slots()
    frames = []
    for frame_no in range(int(rng.integers(2, 7))):
        ft = 0 if not all(s is not None for s in slots.s) else int(rng.choice([0, 1, 1, 1, 2, 3]))
        hdr = tfh.random_header(rng, seq, slots, ft, frame_no)
        payload, nbits = fw.write_frame_header(seq, hdr, slots)
        slots.refresh(hdr)
        frames.append(payload + bytes(rng.integers(0, 256, int(rng.integers(4, 200)), dtype=np.uint8)))
    blob = struct.pack("<I", len(seq_bytes)) + seq_bytes + struct.pack("<I", len(frames))
    for f in frames:
        blob += struct.pack("<I", len(f)) + f
    open(os.path.join(out, "frame", f"{n:03d}.bin"), "wb").write(blob)
    n += 1
for seed in range(6):
    data, hdr, apk = twr.synthetic_file(seed, unknown_sizes=bool(seed & 1), extras=bool(seed & 2))
    open(os.path.join(out, "webm", f"{seed:03d}.webm"), "wb").write(data)
for seed, ch, bs in ((1, 1, (64, 512)), (2, 2, (256, 2048)), (3, 6, (256, 2048)), (4, 2, (128, 128)), (5, 3, (512, 8192))):
    hdr = vgen.make_stream(seed, channels=ch, bs=bs)
    pk = vgen.random_packets(seed, hdr[3], count=12)
    items = list(hdr[:3]) + pk
    blob = struct.pack("<I", len(items))
    for it in items:
        blob += struct.pack("<I", len(it)) + it
    open(os.path.join(out, "vorbis", f"{seed:03d}.bin"), "wb").write(blob)
print("corpus in", out)
