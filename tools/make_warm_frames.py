"""easyav1_amd/data/warm_frames.bin: the three parsed frames ea1_recon_warm() decodes at start-up so that the HIP runtime has set every kernel of a first
frame's path up before the first real picture is asked for (easyav1_amd/csrc/av1_recon.cpp).  A key frame and two inter frames, 256 x 192, every
coding tool of the seeded generator (tests/native/synth_stream.cpp: test infrastructure, NOT linked into the product -- this file is how the product gets
its warm-up input without it).  Layout: per frame a 32-bit little-endian size and the packet of include/easyav1_amd_ir_packet.h.
    python tools/make_warm_frames.py"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import easyav1_amd  # noqa: E402
from easyav1_amd import ir  # noqa: E402

easyav1_amd.load()
p = ir.synth_params(256, 192, 1, seed=0x5EED, gop=3)
p.features |= ir.SYN_ALL
out = b""
for fi in range(3):
    pk = ir.pack_frame(ir.synth_frame(p, fi))
    out += struct.pack("<I", len(pk)) + pk
path = os.path.join(ROOT, "easyav1_amd", "data", "warm_frames.bin")
open(path, "wb").write(out)
print(path, len(out), "bytes")
