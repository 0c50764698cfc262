"""WebM files for the public-interface benchmarks: a V_AV1 track whose temporal units carry the frames already parsed
(include/easyav1_amd_ir_packet.h), i.e. what libeasyav1.so reconstructs on the GPU without the block-level symbol
parser.  One seeded GOP of the graded synthetic stream is generated and written `repeats` times with
consecutive timestamps -- every repetition starts with its key frame, so the file is a valid stream of closed GOPs.
Written straight to disk, one cluster per frame (a 4K frame is ~4 MB: the general writer of tests/webm_writer.py
concatenates in memory).

    python tools/make_stream.py out.webm 4k|1080p|360p [gop] [repeats]
"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES = {"4k": (3840, 2160, 4), "4k8": (3840, 2160, 8), "1080p": (1920, 1080, 1), "360p": (640, 360, 1)}


def write_plain_av1_stream(path, tables_path, width, height, gop=12, repeats=1, seed=1, frame_ms=17, policy=None):
    """A PLAIN AV1 track (no private metadata): one closed GOP from the bit-exact generator of the block-parser tests
    (tests/native/av1_streamgen.cpp, arbitrary tables written to `tables_path` for EASYAV1_AMD_AV1_TABLES), written `repeats` times.
    -> (frames in the file, bytes written)"""
    import easyav1_amd
    easyav1_amd.load(build=True)
    from easyav1_amd import av1parse
    from tests import av1_tables_gen as tg
    from tests import streamgen
    from tests import test_av1_block_parse as T
    from tests import test_frame_header as tfh
    from tests import av1_obu_writer as wr
    from tests import webm_writer as ww
    blob = tg.make_blob(5)
    with open(tables_path, "wb") as f:
        f.write(blob.tobytes())
    tables = av1parse.Tables(blob)
    units, frames, headers, counts = T.make_stream((av1parse, streamgen, tables), seed, (width, height), gop, policy=policy, hdr_kw=dict(base_q_idx=96, one_tile=True))
    seq_obu = wr.obu(1, wr.write_sequence_header(tfh.simple_seq(max_width=width, max_height=height, enable_ref_frame_mvs=1, film_grain_params_present=0, enable_superres=0)))
    n = gop * repeats
    hdr = ww.el(ww.EBML, ww.el(0x4286, b"\x01") + ww.el(0x42F7, b"\x01") + ww.el(0x42F2, b"\x04") + ww.el(0x42F3, b"\x08") + ww.el(0x4282, b"webm") +
                ww.el(0x4287, b"\x02") + ww.el(0x4285, b"\x02"))
    info = ww.uint(0x2AD7B1, 1000000) + ww.el(0x4489, struct.pack(">d", float(n * frame_ms))) + ww.el(0x4D80, b"easyav1-amd bench") + ww.el(0x5741, b"easyav1-amd bench")
    track = (ww.uint(0xD7, 1) + ww.uint(0x73C5, 1001) + ww.uint(0x83, 1) + ww.el(0x86, b"V_AV1") + ww.el(0x63A2, bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu) +
             ww.uint(0x23E383, frame_ms * 1000000) + ww.el(0xE0, ww.uint(0xB0, width) + ww.uint(0xBA, height)))
    head = ww.el(ww.INFO, info) + ww.el(ww.TRACKS, ww.el(0xAE, track))
    body = []
    for i in range(n):
        tu = units[i % gop]
        block_head = ww.enc_size(1) + struct.pack(">h", 0) + bytes([0x80 if i % gop == 0 else 0])
        simple = ww.enc_id(0xA3) + ww.enc_size(len(block_head) + len(tu)) + block_head
        tc = ww.uint(0xE7, i * frame_ms)
        body.append(ww.enc_id(ww.CLUSTER) + ww.enc_size(len(tc) + len(simple) + len(tu)) + tc + simple + tu)
    body_len = len(head) + sum(len(b) for b in body)
    data = hdr + ww.enc_id(ww.SEGMENT) + ww.enc_size(body_len, 8) + head + b"".join(body)
    with open(path, "wb") as out:
        out.write(data)
    return n, len(data), sum(len(u) for u in units) * 8 // gop


def write_real_av1_stream(path, fixture_path, repeats=1, frame_ms=17):
    """A REAL AV1 track: the temporal units of a committed libaom stream (tests/golden/av1_stream_*.npz: one closed GOP, the pictures dav1d 1.5.3
    makes of it are the fixture's digests) written `repeats` times with consecutive timestamps, temporal delimiters stripped as a Matroska
    muxer does.  -> (frames in the file, bytes written, width, height, GOP units)"""
    import numpy as np
    from tests import webm_writer as ww
    z = np.load(fixture_path, allow_pickle=False)
    sizes = z["unit_sizes"].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    raw = z["units"].tobytes()
    width, height = int(z["size"][0]), int(z["size"][1])
    units, keys, seq_obu = [], [], None
    for i in range(len(sizes)):
        tu, body, key, pos = raw[offs[i]:offs[i + 1]], b"", False, 0
        while pos < len(tu):
            b0 = tu[pos]
            q = pos + 1 + ((b0 >> 2) & 1)
            size, shift = 0, 0
            while True:
                v = tu[q]
                q += 1
                size |= (v & 0x7F) << shift
                shift += 7
                if not v & 0x80:
                    break
            typ = (b0 >> 3) & 15
            if typ == 1:
                key = True
                seq_obu = seq_obu or tu[pos:q + size]
            if typ != 2:
                body += tu[pos:q + size]
            pos = q + size
        units.append(body)
        keys.append(key)
    gop = len(units)
    n = gop * repeats
    hdr = ww.el(ww.EBML, ww.el(0x4286, b"\x01") + ww.el(0x42F7, b"\x01") + ww.el(0x42F2, b"\x04") + ww.el(0x42F3, b"\x08") + ww.el(0x4282, b"webm") +
                ww.el(0x4287, b"\x02") + ww.el(0x4285, b"\x02"))
    info = ww.uint(0x2AD7B1, 1000000) + ww.el(0x4489, struct.pack(">d", float(n * frame_ms))) + ww.el(0x4D80, b"easyav1-amd bench") + ww.el(0x5741, b"easyav1-amd bench")
    track = (ww.uint(0xD7, 1) + ww.uint(0x73C5, 1001) + ww.uint(0x83, 1) + ww.el(0x86, b"V_AV1") + ww.el(0x63A2, bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu) +
             ww.uint(0x23E383, frame_ms * 1000000) + ww.el(0xE0, ww.uint(0xB0, width) + ww.uint(0xBA, height)))
    head = ww.el(ww.INFO, info) + ww.el(ww.TRACKS, ww.el(0xAE, track))
    body = []
    for i in range(n):
        tu = units[i % gop]
        block_head = ww.enc_size(1) + struct.pack(">h", 0) + bytes([0x80 if keys[i % gop] else 0])
        simple = ww.enc_id(0xA3) + ww.enc_size(len(block_head) + len(tu)) + block_head
        tc = ww.uint(0xE7, i * frame_ms)
        body.append(ww.enc_id(ww.CLUSTER) + ww.enc_size(len(tc) + len(simple) + len(tu)) + tc + simple + tu)
    body_len = len(head) + sum(len(b) for b in body)
    data = hdr + ww.enc_id(ww.SEGMENT) + ww.enc_size(body_len, 8) + head + b"".join(body)
    with open(path, "wb") as out:
        out.write(data)
    return n, len(data), width, height, gop


def write_parsed_stream(path, width, height, tile_cols=1, gop=60, repeats=1, seed=1, frame_ms=17):
    """-> (frames in the file, bytes written)"""
    from easyav1_amd import ir
    from tests import test_frame_header as tfh
    from tests import av1_obu_writer as wr
    from tests import webm_writer as ww
    prm = ir.synth_params(width, height, tile_cols, seed=seed, gop=gop)
    seq_obu = wr.obu(1, wr.write_sequence_header(tfh.simple_seq(max_width=width, max_height=height)))
    units = []
    for k in range(gop):
        f = ir.synth_frame(prm, k)
        units.append(wr.obu(2, b"") + (seq_obu if k == 0 else b"") + wr.obu(5, ir.pack_frame(f)))
        del f
    n = gop * repeats
    hdr = ww.el(ww.EBML, ww.el(0x4286, b"\x01") + ww.el(0x42F7, b"\x01") + ww.el(0x42F2, b"\x04") + ww.el(0x42F3, b"\x08") + ww.el(0x4282, b"webm") +
                ww.el(0x4287, b"\x02") + ww.el(0x4285, b"\x02"))
    info = ww.uint(0x2AD7B1, 1000000) + ww.el(0x4489, struct.pack(">d", float(n * frame_ms))) + ww.el(0x4D80, b"easyav1-amd bench") + ww.el(0x5741, b"easyav1-amd bench")
    track = (ww.uint(0xD7, 1) + ww.uint(0x73C5, 1001) + ww.uint(0x83, 1) + ww.el(0x86, b"V_AV1") + ww.el(0x63A2, bytes([0x81, 0x00, 0x0C, 0x00]) + seq_obu) +
             ww.uint(0x23E383, frame_ms * 1000000) + ww.el(0xE0, ww.uint(0xB0, width) + ww.uint(0xBA, height)))
    head = ww.el(ww.INFO, info) + ww.el(ww.TRACKS, ww.el(0xAE, track))

    def cluster_parts(i):
        tu = units[i % gop]
        key = i % gop == 0
        block_head = ww.enc_size(1) + struct.pack(">h", 0) + bytes([0x80 if key else 0])
        simple = ww.enc_id(0xA3) + ww.enc_size(len(block_head) + len(tu)) + block_head
        tc = ww.uint(0xE7, i * frame_ms)
        return [ww.enc_id(ww.CLUSTER) + ww.enc_size(len(tc) + len(simple) + len(tu)) + tc + simple, tu]

    body_len = len(head) + sum(sum(len(p) for p in cluster_parts(i)) for i in range(n))
    total = 0
    with open(path, "wb") as out:
        for part in (hdr, ww.enc_id(ww.SEGMENT) + ww.enc_size(body_len, 8), head):
            out.write(part)
            total += len(part)
        for i in range(n):
            for part in cluster_parts(i):
                out.write(part)
                total += len(part)
    return n, total


if __name__ == "__main__":
    which = sys.argv[2] if len(sys.argv) > 2 else "1080p"
    W, H, tc = SIZES[which]
    import easyav1_amd
    easyav1_amd.load(build=False)
    frames, nbytes = write_parsed_stream(sys.argv[1], W, H, tc, int(sys.argv[3]) if len(sys.argv) > 3 else 60, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    print(f"{sys.argv[1]}: {frames} frames of {W}x{H}, {nbytes / 1e6:.1f} MB")
