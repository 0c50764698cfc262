"""Block-level AV1 parser through the C ABI (include/easyav1_amd_obu.h: ea1_av1_tables_*, ea1_av1_frame_parser_*): harness-side wrapper."""
import ctypes

import numpy as np

from . import _lib, ir, obu
from .grain import FilmGrain


class SectionInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("kind", ctypes.c_int), ("offset", ctypes.c_size_t), ("count", ctypes.c_uint32),
                ("symbols", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int)]


def blob_sections():
    """the layout of the table blob as the library describes it: list of dicts"""
    L = _lib.lib()
    L.ea1_av1_tables_section.argtypes = [ctypes.c_int, ctypes.c_void_p]
    out = []
    for i in range(L.ea1_av1_tables_num_sections()):
        s = SectionInfo()
        _lib.check(L.ea1_av1_tables_section(i, ctypes.byref(s)), "ea1_av1_tables_section")
        out.append(dict(name=s.name.decode(), kind=s.kind, offset=s.offset, count=s.count, symbols=s.symbols, width=s.width, height=s.height))
    return out


def blob_size():
    L = _lib.lib()
    L.ea1_av1_tables_blob_size.restype = ctypes.c_size_t
    return L.ea1_av1_tables_blob_size()


def new_blob():
    """a zeroed blob with its header written"""
    L = _lib.lib()
    b = np.zeros(blob_size(), np.uint8)
    L.ea1_av1_tables_blob_header.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _lib.check(L.ea1_av1_tables_blob_header(b.ctypes.data, len(b)), "ea1_av1_tables_blob_header")
    return b


class Tables:
    """ea1_av1_tables"""

    def __init__(self, blob):
        L = _lib.lib()
        self._h = ctypes.c_void_p()
        self._blob = np.ascontiguousarray(blob, np.uint8)
        L.ea1_av1_tables_load.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _lib.check(L.ea1_av1_tables_load(self._blob.ctypes.data, len(self._blob), ctypes.byref(self._h)), "ea1_av1_tables_load")

    def close(self):
        if self._h:
            L = _lib.lib()
            L.ea1_av1_tables_destroy.argtypes = [ctypes.c_void_p]
            L.ea1_av1_tables_destroy.restype = None
            L.ea1_av1_tables_destroy(self._h)
            self._h = ctypes.c_void_p()


class Parsed(ctypes.Structure):
    """ea1_av1_parsed"""
    _fields_ = [("frame", ir._IrFrame), ("header", obu.FrameHeader), ("show_frame", ctypes.c_uint8), ("show_existing", ctypes.c_uint8),
                ("has_film_grain", ctypes.c_uint8), ("film_grain", FilmGrain), ("parse_seconds", ctypes.c_double), ("consumed", ctypes.c_size_t)]


def parsed_frame_from_struct(f):
    """ea1_ir_frame (ctypes) -> ir.ParsedFrame with numpy copies of its arrays"""
    desc = f.desc.copy()

    def arr(ptr, dtype, n):
        if n == 0 or not ptr:
            return np.zeros(0, dtype)
        buf = (ctypes.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()
    return ir.ParsedFrame(desc, arr(f.blocks, ir.BLOCK_DTYPE, desc.num_blocks), arr(f.txs, ir.TXBLOCK_DTYPE, desc.num_tx), arr(f.coefs, np.int16, desc.num_coefs),
                          arr(f.aux, np.uint8, desc.num_aux), arr(f.cdef_idx, np.uint8, f.num_sb),
                          [arr(f.lr[pl], ir.LR_UNIT_DTYPE, f.lr_count[pl]) for pl in range(3)])


class FrameParser:
    """ea1_av1_frame_parser"""

    def __init__(self, tables):
        L = _lib.lib()
        self._tables = tables
        self._h = ctypes.c_void_p()
        L.ea1_av1_frame_parser_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.check(L.ea1_av1_frame_parser_create(tables._h, ctypes.byref(self._h)), "ea1_av1_frame_parser_create")

    def set_threads(self, n):
        """parse the tiles of a frame on n threads"""
        L = _lib.lib()
        L.ea1_av1_frame_parser_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.check(L.ea1_av1_frame_parser_set_threads(self._h, n), "ea1_av1_frame_parser_set_threads")

    def allow_chroma_formats(self, on=True):
        """hand out frames of 4:2:2 / 4:4:4 streams too (their records need a consumer that knows the layout: the CPU oracle of the tests)"""
        L = _lib.lib()
        L.ea1_av1_frame_parser_allow_chroma_formats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.check(L.ea1_av1_frame_parser_allow_chroma_formats(self._h, 1 if on else 0), "ea1_av1_frame_parser_allow_chroma_formats")

    def parse(self, data):
        """one temporal unit -> (rc, ParsedFrame or None, Parsed struct)"""
        L = _lib.lib()
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        out = Parsed()
        L.ea1_av1_frame_parser_parse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        rc = L.ea1_av1_frame_parser_parse(self._h, buf, len(data), ctypes.byref(out))
        return rc, (parsed_frame_from_struct(out.frame) if rc == 0 and not out.show_existing else None), out

    def parse_unit(self, data):
        """every frame of a temporal unit -> list of (ParsedFrame or None for show_existing_frame, Parsed struct); raises on an error"""
        frames, pos = [], 0
        while pos < len(data):
            rc, f, info = self.parse(data[pos:])
            if rc == -7:   # EA1_ENOENT: nothing but non-frame OBUs left
                break
            _lib.check(rc, "ea1_av1_frame_parser_parse")
            frames.append((f, info))
            pos += info.consumed
        return frames

    def reset(self):
        L = _lib.lib()
        L.ea1_av1_frame_parser_reset.argtypes = [ctypes.c_void_p]
        L.ea1_av1_frame_parser_reset.restype = None
        L.ea1_av1_frame_parser_reset(self._h)

    def close(self):
        if self._h:
            L = _lib.lib()
            L.ea1_av1_frame_parser_destroy.argtypes = [ctypes.c_void_p]
            L.ea1_av1_frame_parser_destroy.restype = None
            L.ea1_av1_frame_parser_destroy(self._h)
            self._h = ctypes.c_void_p()
