"""Parsed frames of a committed real stream as hand-over packets (include/easyav1_amd_ir_packet.h), size-prefixed: input of time_job_prepare.
    python tools/host_prepare/make_packets.py <fixture name, e.g. bench_uhd_4cols_gop16> <frames> <out.bin>"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import easyav1_amd  # noqa: E402

easyav1_amd.load()
from easyav1_amd import av1parse, ir  # noqa: E402

blob = np.fromfile(os.path.join(ROOT, "easyav1_amd", "data", "av1_spec_tables.bin"), np.uint8)
tables = av1parse.Tables(blob)
d = np.load(os.path.join(ROOT, "tests", "golden", "av1_stream_%s.npz" % sys.argv[1]))
units, sizes = d["units"], d["unit_sizes"]
parser = av1parse.FrameParser(tables)
out, pos = b"", 0
for k, sz in enumerate(sizes[:int(sys.argv[2])]):
    tu = bytes(units[pos:pos + sz])
    pos += sz
    rc, f, info = parser.parse(tu)
    assert rc == 0
    pk = ir.pack_frame(f)
    print("frame %d: %d bytes of stream, %d blocks, %d transform blocks, packet %d bytes" % (k, sz, f.desc.num_blocks, f.desc.num_tx, len(pk)))
    out += struct.pack("<I", len(pk)) + pk
open(sys.argv[3], "wb").write(out)
