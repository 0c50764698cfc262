"""ctypes binding of tests/native/libea1_streamgen.so (TEST INFRASTRUCTURE: the bit-exact AV1 tile generator, see av1_streamgen.cpp)
and the assembly of its tiles into OBUs."""
import ctypes

import easyav1_amd
from easyav1_amd import av1parse, build
from tests import av1_obu_writer as ow

_lib = None


def lib():
    global _lib
    if _lib is None:
        easyav1_amd.load(build=True)
        _lib = ctypes.CDLL(build.build_streamgen())
        _lib.ea1sg_tile_size.restype = ctypes.c_size_t
        _lib.ea1sg_tile_size.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.ea1sg_tile_data.restype = ctypes.c_void_p
        _lib.ea1sg_tile_data.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.ea1sg_frame_ir.restype = ctypes.POINTER(av1parse.ir._IrFrame)
        _lib.ea1sg_frame_ir.argtypes = [ctypes.c_void_p]
        _lib.ea1sg_element_count.restype = ctypes.c_uint64
        _lib.ea1sg_element_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.ea1sg_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.ea1sg_destroy.argtypes = [ctypes.c_void_p]
        _lib.ea1sg_destroy.restype = None
        _lib.ea1sg_set_sequence_header.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib.ea1sg_frame.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_void_p]
        _lib.ea1sg_temporal_candidates.restype = ctypes.c_uint64
        _lib.ea1sg_temporal_candidates.argtypes = [ctypes.c_void_p]
        _lib.ea1sg_show_existing.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib.ea1sg_policy.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
        _lib.ea1sg_policy.restype = None
    return _lib


class Generator:
    def __init__(self, tables, seq_obu):
        L = lib()
        self._tables = tables
        self._h = ctypes.c_void_p()
        assert L.ea1sg_create(tables._h, ctypes.byref(self._h)) == 0
        rc = L.ea1sg_set_sequence_header(self._h, seq_obu, len(seq_obu))
        assert rc == 0, easyav1_amd._lib.last_error()

    def policy(self, skip=30, inter=75, all_zero=40, max_mv_class=5, split_bias=40, br_stop=None, golomb=None):
        lib().ea1sg_policy(self._h, skip, inter, all_zero, max_mv_class, split_bias)
        if br_stop is not None or golomb is not None:   # coefficient magnitudes (tame ones for streams that go to a decoder which checks conformance)
            L = lib()
            L.ea1sg_policy_coefs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            L.ea1sg_policy_coefs.restype = None
            L.ea1sg_policy_coefs(self._h, 60 if br_stop is None else br_stop, 60 if golomb is None else golomb)

    def frame(self, header_bytes, seed):
        """uncompressed_header bytes -> (rc, [tile bytes], ParsedFrame the generator itself built)"""
        L = lib()
        n = ctypes.c_int()
        rc = L.ea1sg_frame(self._h, header_bytes, len(header_bytes), seed, ctypes.byref(n))
        if rc != 0:
            return rc, None, None
        tiles = [ctypes.string_at(L.ea1sg_tile_data(self._h, t), L.ea1sg_tile_size(self._h, t)) for t in range(n.value)]
        return 0, tiles, av1parse.parsed_frame_from_struct(L.ea1sg_frame_ir(self._h).contents)

    def show_existing(self, header_bytes):
        return lib().ea1sg_show_existing(self._h, header_bytes, len(header_bytes))

    def temporal_candidates(self):
        return lib().ea1sg_temporal_candidates(self._h)

    def element_counts(self):
        L = lib()
        return [L.ea1sg_element_count(self._h, e) for e in range(L.ea1sg_num_elements())]

    def close(self):
        if self._h:
            lib().ea1sg_destroy(self._h)
            self._h = ctypes.c_void_p()


def frame_obu(hdr, hdr_bytes, tiles):
    """OBU_FRAME: the header, byte alignment, one tile group with every tile (tile_start_and_end_present_flag = 0)"""
    data = b""
    if len(tiles) > 1:
        data = b"\x00"      # the flag + byte alignment
    for t, body in enumerate(tiles):
        if t != len(tiles) - 1:
            data += (len(body) - 1).to_bytes(hdr["tile_size_bytes"], "little")
        data += body
    return ow.obu(6, hdr_bytes + data)


def frame_obus_split(hdr, hdr_bytes, nbits, tiles, rng):
    """the same frame as OBU_FRAME_HEADER (with its trailing bits) followed by one to three OBU_TILE_GROUPs that carry
    tile_start_and_end_present_flag = 1 and their tile ranges (what encoders that do not use OBU_FRAME write) -> (bytes, number of groups)"""
    from tests import av1_frame_header_writer as fw
    hb = bytearray(hdr_bytes[: (nbits + 7) // 8])
    if nbits % 8 == 0:
        hb.append(0x80)
    else:
        hb[-1] |= 0x80 >> (nbits % 8)
    out = ow.obu(3, bytes(hb))
    n = len(tiles)
    bits = hdr["tile_cols_log2"] + hdr["tile_rows_log2"]
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, size=min(2, n - 1))])) if n > 1 else [0, 1]
    for a, b in zip(cuts[:-1], cuts[1:]):
        data = b""
        if n > 1:
            w = fw.Bits()
            w.f(1, 1)
            w.f(bits, a)
            w.f(bits, b - 1)
            while len(w.bits) % 8:
                w.bits.append(0)
            data = w.bytes()
        for t in range(a, b):
            if t != b - 1:
                data += (len(tiles[t]) - 1).to_bytes(hdr["tile_size_bytes"], "little")
            data += tiles[t]
        out += ow.obu(4, data)
    return out, len(cuts) - 1

