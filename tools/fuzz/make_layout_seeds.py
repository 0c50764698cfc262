"""Seeds for the job fuzzer in the chroma layouts the synthetic generator does not write: a few parsed frames of the committed generated fixtures (4:4:4 and 4:2:2
-- key and inter frames with every tool the generator draws) as hand-over packets (include/easyav1_amd_ir_packet.h), size-prefixed, in <dir>/layout_frames.bin.
    python tools/fuzz/make_layout_seeds.py <dir>"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import easyav1_amd  # noqa: E402

easyav1_amd.load()
from easyav1_amd import av1parse, ir  # noqa: E402
from tests import test_generated_av1 as tg  # noqa: E402

blob = np.fromfile(os.path.join(os.path.dirname(easyav1_amd.__file__), "data", "av1_spec_tables.bin"), np.uint8)
tables = av1parse.Tables(blob)
out = b""
n = 0
for name in ("c444_200x136", "c422_200x136", "c444_sb128_small_blocks_320x192", "c422_switch_frames_264x200"):
    fx = tg.load([p for p in tg.FIXTURES if p.endswith("av1_gen_%s.npz" % name)][0])
    parser = av1parse.FrameParser(tables)
    for k, tu in enumerate(fx["units"][:3]):
        rc, f, info = parser.parse(tu)
        assert rc == 0
        pk = ir.pack_frame(f)
        out += struct.pack("<I", len(pk)) + pk
        n += 1
    parser.close()
os.makedirs(sys.argv[1], exist_ok=True)
path = os.path.join(sys.argv[1], "layout_frames.bin")
open(path, "wb").write(out)
print(path, n, "frames,", len(out), "bytes")
