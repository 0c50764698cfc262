"""Minimal WebM (Matroska) writer for the reader tests: there is no muxer in this environment.
Produces the element forms a reader has to cope with: SimpleBlock / BlockGroup, no / Xiph / EBML /
fixed lacing, known- and unknown-size Segments and Clusters, Void and CRC-32 elements, Cues at the
end, 1..8-byte sizes."""
import struct

import numpy as np

EBML, SEGMENT, INFO, TRACKS, CLUSTER, CUES = 0x1A45DFA3, 0x18538067, 0x1549A966, 0x1654AE6B, 0x1F43B675, 0x1C53BB6B


def enc_id(i):
    n = (i.bit_length() + 7) // 8
    return i.to_bytes(n, "big")


def enc_size(v, length=None):
    if v is None:                       # unknown size
        return b"\x01" + b"\xff" * 7
    n = length or 1
    while v >= (1 << (7 * n)) - 1:
        n += 1
    return ((1 << (7 * n)) | v).to_bytes(n, "big")


def el(i, payload, size_len=None):
    return enc_id(i) + enc_size(len(payload), size_len) + payload


def uint(i, v, n=None):
    n = n or max(1, (v.bit_length() + 7) // 8)
    return el(i, v.to_bytes(n, "big"))


def xiph_lace(sizes):
    out = b""
    for s in sizes:
        out += b"\xff" * (s // 255) + bytes([s % 255])
    return out


def vorbis_private(headers):
    return bytes([len(headers) - 1]) + xiph_lace([len(h) for h in headers[:-1]]) + b"".join(headers)


def block_body(track, rel_tc, flags, frames, lacing):
    head = enc_size(track) + struct.pack(">h", rel_tc)
    if lacing == 0 or len(frames) == 1:
        assert len(frames) == 1
        return head + bytes([flags]) + frames[0]
    n = len(frames)
    if lacing == 1:     # Xiph
        return head + bytes([flags | 0x02, n - 1]) + xiph_lace([len(f) for f in frames[:-1]]) + b"".join(frames)
    if lacing == 2:     # fixed
        assert len({len(f) for f in frames}) == 1
        return head + bytes([flags | 0x04, n - 1]) + b"".join(frames)
    sizes = b""         # EBML: first size, then signed differences
    for i, f in enumerate(frames[:-1]):
        if i == 0:
            sizes += enc_size(len(f))
        else:
            d = len(f) - len(frames[i - 1])
            ln = 1
            while not (-(1 << (7 * ln - 1)) + 1 <= d <= (1 << (7 * ln - 1)) - 1):
                ln += 1
            sizes += ((1 << (7 * ln)) | (d + (1 << (7 * ln - 1)) - 1)).to_bytes(ln, "big")
    return head + bytes([flags | 0x06, n - 1]) + sizes + b"".join(frames)


def make_webm(seed, tracks, blocks, tc_scale=1000000, duration=None, unknown_sizes=False, extras=True):
    """tracks: list of dicts (number, type 1/2, codec, private bytes or header list, default_duration,
    width/height or rate/channels); blocks: list of dicts (track number, tc (absolute, in timecode
    units), key, frames [bytes], lacing 0..3, group bool, duration or None) in file order"""
    rng = np.random.default_rng(seed)
    hdr = el(EBML, el(0x4286, b"\x01") + el(0x42F7, b"\x01") + el(0x42F2, b"\x04") + el(0x42F3, b"\x08") + el(0x4282, b"webm") +
             el(0x4287, b"\x02") + el(0x4285, b"\x02"))
    info = uint(0x2AD7B1, tc_scale)
    if duration is not None:
        info += el(0x4489, struct.pack(">d", float(duration)))
    info += el(0x4D80, b"easyav1-amd tests") + el(0x5741, b"easyav1-amd tests")
    tr = b""
    for t in tracks:
        e = uint(0xD7, t["number"]) + uint(0x73C5, 1000 + t["number"]) + uint(0x83, t["type"]) + el(0x86, t["codec"].encode())
        if t.get("private") is not None:
            e += el(0x63A2, t["private"])
        if t.get("default_duration"):
            e += uint(0x23E383, t["default_duration"])
        if t.get("codec_delay"):
            e += uint(0x56AA, t["codec_delay"])
        if t["type"] == 1:
            e += el(0xE0, uint(0xB0, t["width"]) + uint(0xBA, t["height"]))
        else:
            e += el(0xE1, el(0xB5, struct.pack(">d", float(t["rate"]))) + uint(0x9F, t["channels"]))
        tr += el(0xAE, e)
    body = el(INFO, info)
    if extras:
        body += el(0xEC, bytes(int(rng.integers(0, 40))))            # Void
    body += el(TRACKS, tr, size_len=int(rng.integers(1, 5)))
    # clusters of a few blocks each
    i = 0
    while i < len(blocks):
        n = int(rng.integers(1, 6))
        group = blocks[i:i + n]
        i += n
        base = min(b["tc"] for b in group)
        c = b""
        if extras and rng.random() < 0.3:
            c += el(0xBF, b"\x00\x00\x00\x00")                        # CRC-32 before the Timecode
        c += uint(0xE7, base)
        for b in group:
            flags = 0x80 if (b["key"] and not b["group"]) else 0
            bb = block_body(b["track"], b["tc"] - base, flags, b["frames"], b["lacing"])
            if b["group"]:
                g = el(0xA1, bb)
                if b.get("duration") is not None:
                    g += uint(0x9B, b["duration"])
                if not b["key"]:
                    g += el(0xFB, b"\xff")                            # ReferenceBlock -1
                c += el(0xA0, g)
            else:
                c += el(0xA3, bb, size_len=int(rng.integers(1, 4)))
            if extras and rng.random() < 0.1:
                c += el(0xEC, bytes(int(rng.integers(0, 8))))
        if unknown_sizes:
            body += enc_id(CLUSTER) + enc_size(None) + c
        else:
            body += el(CLUSTER, c)
    if extras:
        body += el(CUES, el(0xBB, uint(0xB3, 0) + el(0xB7, uint(0xF7, 1) + uint(0xF1, 0))))
    seg = enc_id(SEGMENT) + (enc_size(None) if unknown_sizes else enc_size(len(body), 8)) + body
    return hdr + seg
