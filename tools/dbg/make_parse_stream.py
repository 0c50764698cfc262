"""stream file for tools/dbg/parse_prof.cpp / job_prof_real.cpp: [u32 n][table blob] then [u32 n][temporal unit] ... from a committed fixture
    python tools/dbg/make_parse_stream.py tests/golden/av1_stream_bench_uhd_4cols_gop16.npz /tmp/bench4k.stream"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = np.load(sys.argv[1], allow_pickle=False)
sizes = z["unit_sizes"].astype(np.int64)
offs = np.concatenate([[0], np.cumsum(sizes)])
raw = z["units"].tobytes()
blob = open(os.path.join(ROOT, "easyav1_amd", "data", "av1_spec_tables.bin"), "rb").read()
with open(sys.argv[2], "wb") as fp:
    fp.write(struct.pack("<I", len(blob)) + blob)
    for i in range(len(sizes)):
        u = raw[offs[i]:offs[i + 1]]
        fp.write(struct.pack("<I", len(u)) + u)
print(len(sizes), "units ->", sys.argv[2])
