"""Python mirror of the ReconJob IR structs of include/easyav1_amd_video.h / _synth.h."""
import ctypes

import numpy as np

NUM_REF_SLOTS = 9

TX_DCT, TX_ADST, TX_FLIPADST, TX_IDENTITY = 0, 1, 2, 3
(DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED, SMOOTH_PRED,
 SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, UV_CFL_PRED) = range(14)
FILTER_REGULAR, FILTER_SMOOTH, FILTER_SHARP, FILTER_BILINEAR = range(4)
COMP_AVG, COMP_DIST, COMP_WEDGE, COMP_DIFFWTD = range(4)
II_DC, II_V, II_H, II_SMOOTH, II_WEDGE = 0, 1, 2, 3, 0x80
BLK_INTER, BLK_COMPOUND, BLK_SKIP, BLK_HAVE_LEFT, BLK_HAVE_TOP = 1, 2, 4, 8, 16
BLK_HAVE_TOP_RIGHT, BLK_HAVE_BOTTOM_LEFT, BLK_OBMC, BLK_INTERINTRA = 0x20, 0x40, 0x80, 0x100
BLK_PALETTE_Y, BLK_PALETTE_UV = 0x200, 0x400
BLK_CHROMA_HAVE_LEFT, BLK_CHROMA_HAVE_TOP, BLK_CHROMA_HAVE_TOP_RIGHT, BLK_CHROMA_HAVE_BOTTOM_LEFT = 0x800, 0x1000, 0x2000, 0x4000
TXTP_WHT = 16
NO_COEFS = 0xFFFFFFFF
LR_NONE, LR_WIENER, LR_SGRPROJ = 0, 1, 2
STAGE_RECON, STAGE_DEBLOCK, STAGE_CDEF, STAGE_LR, STAGE_ALL = 1, 2, 4, 8, 15
LAYOUT_420, LAYOUT_444, LAYOUT_422 = 0, 1, 2   # EA1_LAYOUT_*: ea1_frame_desc.layout, ea1_recon_create_layout
# synthetic-stream feature bits (tests/native/easyav1_amd_synth.h)
(SYN_RECT, SYN_TX_SPLIT, SYN_ALL_TXTP, SYN_SMOOTH, SYN_DIRECTIONAL, SYN_CFL, SYN_FILTER_INTRA, SYN_PALETTE,
 SYN_COMPOUND, SYN_OBMC, SYN_INTERINTRA, SYN_BILINEAR, SYN_LF_DELTA, SYN_SUB8, SYN_LR_MIX) = (1 << i for i in range(15))
SYN_ALL = 0x7FFF
SYN_LOSSLESS = 0x8000
SYN_WARP = 0x10000
SYN_INTRABC = 0x20000
SYN_NO_NZ_HINT = 0x40000
SYN_COMPACT_COEFS = 0x80000
SYN_DIAGONAL_COEFS = 0x100000
COEFS_DENSE, COEFS_CORNER, COEFS_DIAGONAL = 0, 1, 2
INTRABC_PRED = 14
BLK_WARP = 0x8000
# transform sizes in the order of ea1_tx_size_index()
TX_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16),
            (32, 64), (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def txtp(col_kind, row_kind):
    return (col_kind << 2) | row_kind


BLOCK_DTYPE = np.dtype({
    "names": ["x4", "y4", "bwl", "bhl", "flags", "mode_y", "mode_uv", "ref", "angle_delta", "cfl_alpha", "mv",
              "comp_type", "comp_arg", "ii_mode", "ii_arg", "filter_intra", "pal_size", "warp_refs", "first_tx", "n_tx",
              "pad1", "aux_off", "lf_level"],
    "formats": ["<u2", "<u2", "u1", "u1", "<u2", "u1", "u1", ("u1", 2), ("i1", 2), ("i1", 2), ("<i2", (2, 2)),
                "u1", "u1", "u1", "u1", "u1", ("u1", 2), "u1", "<u4", "<u2", "<u2", "<u4", ("u1", 4)],
    "offsets": [0, 2, 4, 5, 6, 8, 9, 10, 12, 14, 16, 24, 25, 26, 27, 28, 29, 31, 32, 36, 38, 40, 44],
    "itemsize": 48,
})

TXBLOCK_DTYPE = np.dtype({
    "names": ["x4", "y4", "plane", "size", "txtp", "nz_log2", "coef_off"],
    "formats": ["<u2", "<u2", "u1", "u1", "u1", "u1", "<u4"],
    "offsets": [0, 2, 4, 5, 6, 7, 8],
    "itemsize": 12,
})

LR_UNIT_DTYPE = np.dtype({
    "names": ["type", "sgr_set", "coef"],
    "formats": ["u1", "u1", ("i1", (2, 3))],
    "offsets": [0, 1, 2],
    "itemsize": 8,
})


class FrameDesc(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
        ("is_inter", ctypes.c_uint8), ("dst_slot", ctypes.c_uint8),
        ("lf_level", ctypes.c_uint8 * 4), ("lf_sharpness", ctypes.c_uint8),
        ("cdef_enable", ctypes.c_uint8), ("cdef_damping", ctypes.c_uint8),
        ("cdef_y_strength", ctypes.c_uint8 * 8), ("cdef_uv_strength", ctypes.c_uint8 * 8),
        ("lr_type", ctypes.c_uint8 * 3), ("lr_unit_log2", ctypes.c_uint8 * 3), ("intra_edge_filter", ctypes.c_uint8),
        ("lf_per_block", ctypes.c_uint8), ("crop", ctypes.c_uint8), ("coef_layout", ctypes.c_uint8), ("pad", ctypes.c_uint8),
        ("num_blocks", ctypes.c_uint32), ("num_coefs", ctypes.c_uint32), ("num_tx", ctypes.c_uint32),
        ("num_aux", ctypes.c_uint32), ("upscaled_width", ctypes.c_uint32),
        ("layout", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 3),
    ]

    def copy(self):
        d = FrameDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(self), ctypes.sizeof(self))
        return d


class SynthParams(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("tile_cols", ctypes.c_uint32),
        ("gop", ctypes.c_uint32), ("seed", ctypes.c_uint64), ("features", ctypes.c_uint32),
        ("lr_unit_log2", ctypes.c_uint8), ("enable_deblock", ctypes.c_uint8), ("enable_cdef", ctypes.c_uint8),
        ("enable_lr", ctypes.c_uint8), ("tile_rows", ctypes.c_uint32),
        ("crop_right", ctypes.c_uint8), ("crop_bottom", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 2),
    ]


class _SynthFrame(ctypes.Structure):
    _fields_ = [
        ("desc", FrameDesc), ("blocks", ctypes.c_void_p), ("txs", ctypes.c_void_p), ("coefs", ctypes.c_void_p),
        ("aux", ctypes.c_void_p), ("cdef_idx", ctypes.c_void_p),
        ("lr", ctypes.c_void_p * 3), ("lr_count", ctypes.c_uint32 * 3), ("num_sb", ctypes.c_uint32),
    ]


class Picture(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p * 3), ("stride", ctypes.c_size_t * 3), ("width", ctypes.c_uint32),
                ("height", ctypes.c_uint32)]


def lr_unit_counts(desc):
    out = []
    for pl in range(3):
        ss = 1 if pl else 0
        usz = 1 << desc.lr_unit_log2[pl]

        def units(size):
            return max((size + (usz >> 1)) // usz, 1)
        vis_w, vis_h = desc.width - 8 * (desc.pad & 15) - (desc.crop & 15), desc.height - 8 * (desc.pad >> 4) - (desc.crop >> 4)
        out.append((units((vis_h + ss) >> ss), units((vis_w + ss) >> ss)))
    return out


class ParsedFrame:
    """Host copy of one parsed frame (what ea1_job_create takes)."""

    def __init__(self, desc, blocks, txs, coefs, aux, cdef_idx, lr):
        self.desc = desc
        self.blocks = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        self.txs = np.ascontiguousarray(txs, TXBLOCK_DTYPE)
        self.coefs = np.ascontiguousarray(coefs, np.int16)
        self.aux = np.ascontiguousarray(aux, np.uint8)
        self.cdef_idx = np.ascontiguousarray(cdef_idx, np.uint8)
        self.lr = [np.ascontiguousarray(u, LR_UNIT_DTYPE) for u in lr]
        self.desc.num_blocks = len(self.blocks)
        self.desc.num_coefs = len(self.coefs)
        self.desc.num_tx = len(self.txs)
        self.desc.num_aux = len(self.aux)

    def lr_pointers(self):
        arr = (ctypes.c_void_p * 3)()
        for pl in range(3):
            arr[pl] = self.lr[pl].ctypes.data if len(self.lr[pl]) else None
        return arr


def _bind_synth(L):
    if getattr(L, "_ea1_synth_bound", False):
        return
    L.ea1_synth_default_params.argtypes = [ctypes.POINTER(SynthParams), ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_uint64]
    L.ea1_synth_default_params.restype = None
    L.ea1_synth_frame_create.argtypes = [ctypes.POINTER(SynthParams), ctypes.c_uint32,
                                         ctypes.POINTER(ctypes.POINTER(_SynthFrame))]
    L.ea1_synth_frame_destroy.argtypes = [ctypes.POINTER(_SynthFrame)]
    L.ea1_synth_frame_destroy.restype = None
    L.ea1_synth_fill_picture.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]
    L.ea1_synth_fill_picture.restype = None
    L.ea1_synth_warp_filter.argtypes = [ctypes.c_void_p]
    L.ea1_synth_warp_filter.restype = None
    L._ea1_synth_bound = True


def synth_params(width, height, tile_cols=1, seed=1, **overrides):
    from . import _lib
    L = _lib.synth_lib()
    _bind_synth(L)
    p = SynthParams()
    L.ea1_synth_default_params(ctypes.byref(p), width, height, tile_cols, seed)
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def synth_frame(params, frame_index):
    """Seeded parsed frame -> ParsedFrame (numpy copies; host only, no GPU involved)."""
    from . import _lib
    L = _lib.synth_lib()
    _bind_synth(L)
    fp = ctypes.POINTER(_SynthFrame)()
    _lib.check(L.ea1_synth_frame_create(ctypes.byref(params), frame_index, ctypes.byref(fp)), "ea1_synth_frame_create")
    f = fp.contents
    try:
        desc = f.desc.copy()

        def arr(ptr, dtype, n):
            if n == 0:
                return np.zeros(0, dtype)
            buf = (ctypes.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n).copy()
        blocks = arr(f.blocks, BLOCK_DTYPE, desc.num_blocks)
        txs = arr(f.txs, TXBLOCK_DTYPE, desc.num_tx)
        coefs = arr(f.coefs, np.int16, desc.num_coefs)
        aux = arr(f.aux, np.uint8, desc.num_aux)
        cdef_idx = arr(f.cdef_idx, np.uint8, f.num_sb)
        lr = [arr(f.lr[pl], LR_UNIT_DTYPE, f.lr_count[pl]) for pl in range(3)]
    finally:
        L.ea1_synth_frame_destroy(fp)
    return ParsedFrame(desc, blocks, txs, coefs, aux, cdef_idx, lr)


def synth_picture(width, height, seed):
    from . import _lib
    L = _lib.synth_lib()
    _bind_synth(L)
    y = np.zeros((height, width), np.uint8)
    u = np.zeros((height // 2, width // 2), np.uint8)
    v = np.zeros((height // 2, width // 2), np.uint8)
    L.ea1_synth_fill_picture(width, height, seed, y.ctypes.data, u.ctypes.data, v.ctypes.data)
    return [y, u, v]


def synth_warp_filter():
    """The stand-in 193 x 8 warp filter table of tests/native/easyav1_amd_synth.h (NOT the AV1 table)."""
    from . import _lib
    L = _lib.synth_lib()
    _bind_synth(L)
    t = np.zeros((193, 8), np.int8)
    L.ea1_synth_warp_filter(t.ctypes.data)
    return t


class _IrFrame(ctypes.Structure):
    """ea1_ir_frame of include/easyav1_amd_ir_packet.h"""
    _fields_ = [("desc", FrameDesc), ("blocks", ctypes.c_void_p), ("txs", ctypes.c_void_p), ("coefs", ctypes.c_void_p),
                ("aux", ctypes.c_void_p), ("cdef_idx", ctypes.c_void_p), ("num_sb", ctypes.c_uint32),
                ("lr", ctypes.c_void_p * 3), ("lr_count", ctypes.c_uint32 * 3)]


def ir_frame_struct(frame):
    """ea1_ir_frame view of a ParsedFrame (borrows its arrays; cached on the frame)"""
    f = getattr(frame, "_ir_struct", None)
    if f is not None and f._coefs_ptr == (frame.coefs.ctypes.data if len(frame.coefs) else None):
        return f
    f = _IrFrame()
    f.desc = frame.desc.copy()
    f.blocks = frame.blocks.ctypes.data if len(frame.blocks) else None
    f.txs = frame.txs.ctypes.data if len(frame.txs) else None
    f.coefs = frame.coefs.ctypes.data if len(frame.coefs) else None
    f.aux = frame.aux.ctypes.data if len(frame.aux) else None
    f.cdef_idx = frame.cdef_idx.ctypes.data if len(frame.cdef_idx) else None
    f.num_sb = len(frame.cdef_idx)
    for pl in range(3):
        f.lr[pl] = frame.lr[pl].ctypes.data if len(frame.lr[pl]) else None
        f.lr_count[pl] = len(frame.lr[pl])
    f._coefs_ptr = f.coefs
    frame._ir_struct = f
    return f


def pack_frame(frame):
    """ParsedFrame -> payload of the private OBU_METADATA that carries it through a container
    (ea1_ir_packet_write); wrap it with obu type 5 to place it in a temporal unit"""
    from . import _lib
    L = _lib.lib()
    L.ea1_ir_packet_bytes.argtypes = [ctypes.c_void_p]
    L.ea1_ir_packet_bytes.restype = ctypes.c_size_t
    L.ea1_ir_packet_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    f = ir_frame_struct(frame)
    n = L.ea1_ir_packet_bytes(ctypes.byref(f))
    out = np.empty(n, np.uint8)
    _lib.check(L.ea1_ir_packet_write(ctypes.byref(f), out.ctypes.data, n), "ea1_ir_packet_write")
    return out.tobytes()
