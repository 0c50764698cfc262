"""Seed corpus for tools/fuzz/fuzz_av1_blocks.cpp: generated plain-AV1 streams (tests/native/av1_streamgen.cpp) with the arbitrary table blob they
were written with:  python tools/fuzz/make_av1_corpus.py /tmp/ea1_av1_corpus"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import easyav1_amd  # noqa: E402

easyav1_amd.load(build=True)
from easyav1_amd import av1parse  # noqa: E402
from tests import av1_tables_gen as tg  # noqa: E402
from tests import streamgen  # noqa: E402
from tests import test_av1_block_parse as T  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
blob = tg.make_blob(5)
tables = av1parse.Tables(blob)
env = (av1parse, streamgen, tables)
n = 0


def write(units):
    global n
    with open(os.path.join(out, f"{n:03d}.bin"), "wb") as f:
        b = blob.tobytes()
        f.write(struct.pack("<I", len(b)) + b)
        for u in units:
            f.write(struct.pack("<I", len(u)) + u)
    n += 1


for seed, size in [(1, (64, 64)), (2, (128, 72)), (3, (168, 104)), (4, (232, 152)), (5, (104, 168)), (6, (320, 192)), (7, (72, 200)), (8, (296, 216))]:
    units, frames, headers, counts = T.make_stream(env, seed, size, 5)
    write(units)
    units, frames, headers, counts = T.make_stream(env, 100 + seed, size, 4, gop=2)
    write(units)
for seed, size in [(21, (256, 128)), (22, (200, 136)), (23, (320, 264)), (24, (166, 102))]:   # superblocks of 128x128, blocks of 128 samples
    units, frames, headers, counts = T.make_stream(env, seed, size, 4, **T.SB128)
    write(units)
for seed in (31, 32):   # every frame with its own size, references of other sizes
    units, frames, headers, counts = T.make_stream(env, seed, (264, 200), 5, **T.RESIZED)
    write(units)
for seed in (41, 42):   # super-resolution
    units, frames, headers, counts = T.make_stream(env, seed, (264, 200), 5, **T.SUPERRES)
    write(units)
for seed in (51,):   # mono_chrome
    units, frames, headers, counts = T.make_stream(env, seed, (200, 136), 5, **T.MONO)
    write(units)
S444 = dict(profile=1, subsampling_x=0, subsampling_y=0, layout=3)
S422 = dict(profile=2, subsampling_x=1, subsampling_y=0, layout=2)
for seed, size, sub in [(61, (200, 136), S444), (62, (166, 102), S422), (63, (264, 200), S444), (64, (320, 192), S422)]:   # 4:4:4 / 4:2:2 (round 5)
    units, frames, headers, counts = T.make_stream(env, seed, size, 4, seq_kw=sub)
    write(units)
    units, frames, headers, counts = T.make_stream(env, 100 + seed, size, 4, seq_kw=dict(sub, use_128x128_superblock=1, enable_superres=1), policy=dict(split_bias=10))
    write(units)
for hk in (False, True):
    units, frames, shows = T.make_hidden_frame_stream(env, 50 + hk, (168, 104), hk)
    write(units)
print(n, "streams in", out)
