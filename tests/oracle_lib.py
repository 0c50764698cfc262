"""ctypes loaders for the CPU checkers (oracle/liboracle.so and oracle/_ref/*.so).

Test infrastructure only: nothing under easyav1_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        L = ctypes.CDLL(build_oracle())
        L.ea1o_mdct_tables.argtypes = [ctypes.c_int, _f32p, _i32p]
        L.ea1o_window.argtypes = [ctypes.c_int, _f32p]
        L.ea1o_mdct_backward.argtypes = [ctypes.c_int, _f32p, _i32p, _f32p, _f32p]
        L.ea1o_synth_new.restype = ctypes.c_void_p
        L.ea1o_synth_new.argtypes = [ctypes.c_int] * 3
        L.ea1o_synth_free.argtypes = [ctypes.c_void_p]
        L.ea1o_synth_restart.argtypes = [ctypes.c_void_p]
        L.ea1o_synth_blockin.argtypes = [ctypes.c_void_p, ctypes.c_int, _f32p]
        L.ea1o_synth_pcmout.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int]
        L.ea1o_synth_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ea1o_ring_new.restype = ctypes.c_void_p
        L.ea1o_ring_new.argtypes = [ctypes.c_int, ctypes.c_int]
        L.ea1o_ring_free.argtypes = [ctypes.c_void_p]
        L.ea1o_ring_push.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, ctypes.c_int]
        L.ea1o_ring_begin_packet.argtypes = [ctypes.c_void_p]
        L.ea1o_ring_get.restype = ctypes.c_uint
        L.ea1o_ring_get.argtypes = [ctypes.c_void_p, ctypes.POINTER(_f32p), ctypes.POINTER(ctypes.c_size_t)]
        _oracle = L
    return _oracle


def ref_minivorbis():
    """The reference's own code (oracle/_ref), or None when it has not been built."""
    so = os.path.join(ORACLE_DIR, "_ref", "libref_minivorbis.so")
    if not os.path.exists(so):
        return None
    L = ctypes.CDLL(so)
    L.ref_mdct_new.restype = ctypes.c_void_p
    L.ref_mdct_new.argtypes = [ctypes.c_int]
    L.ref_mdct_free.argtypes = [ctypes.c_void_p]
    L.ref_mdct_backward.argtypes = [ctypes.c_void_p, _f32p, _f32p]
    L.ref_mdct_trig.restype = _f32p
    L.ref_mdct_trig.argtypes = [ctypes.c_void_p]
    L.ref_mdct_bitrev.restype = _i32p
    L.ref_mdct_bitrev.argtypes = [ctypes.c_void_p]
    L.ref_window.restype = _f32p
    L.ref_window.argtypes = [ctypes.c_int]
    L.ref_synth_new.restype = ctypes.c_void_p
    L.ref_synth_new.argtypes = [ctypes.c_int] * 3
    L.ref_synth_free.argtypes = [ctypes.c_void_p]
    L.ref_synth_blockin.argtypes = [ctypes.c_void_p, ctypes.c_int, _f32p]
    L.ref_synth_pcmout.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int]
    L.ref_synth_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ref_synth_restart.argtypes = [ctypes.c_void_p]
    return L


# ---------------------------------------------------------------- oracle convenience wrappers
class OracleMdct:
    def __init__(self, n):
        self.n = n
        self.trig = np.zeros(n + n // 4, np.float32)
        self.bitrev = np.zeros(n // 4, np.int32)
        oracle().ea1o_mdct_tables(n, fptr(self.trig), self.bitrev.ctypes.data_as(_i32p))

    def backward(self, spec):
        """spec: (..., n/2) float32 -> (..., n) float32"""
        spec = np.ascontiguousarray(spec, np.float32)
        lead = spec.shape[:-1]
        flat = spec.reshape(-1, self.n // 2)
        out = np.zeros((flat.shape[0], self.n), np.float32)
        L = oracle()
        for i in range(flat.shape[0]):
            L.ea1o_mdct_backward(self.n, fptr(self.trig), self.bitrev.ctypes.data_as(_i32p),
                                 fptr(flat[i]), fptr(out[i]))
        return out.reshape(*lead, self.n)


def oracle_window(n):
    w = np.zeros(n // 2, np.float32)
    oracle().ea1o_window(n, fptr(w))
    return w


def oracle_synth_stream(channels, bs0, bs1, flags, blocks, restart_at=()):
    """Run the blockin/pcmout/read state machine over a block sequence.

    flags: list of W; blocks: list of (channels, bs[W]) float32 arrays (IMDCT outputs).
    Returns list of per-block (channels, count) arrays."""
    L = oracle()
    h = L.ea1o_synth_new(channels, bs0, bs1)
    outs = []
    buf = np.zeros((channels, bs1), np.float32)
    for k, (W, blk) in enumerate(zip(flags, blocks)):
        if k in restart_at:
            L.ea1o_synth_restart(h)
        blk = np.ascontiguousarray(blk, np.float32)
        assert L.ea1o_synth_blockin(h, W, fptr(blk)) == 0
        n = L.ea1o_synth_pcmout(h, fptr(buf), bs1)
        assert n >= 0
        outs.append(buf[:, :n].copy())
        L.ea1o_synth_read(h, n)
    L.ea1o_synth_free(h)
    return outs


def fnv1a32(arr):
    """FNV-1a-32 folded over 32-bit words (the hash SURVEY.md Appendix B quotes for its probes)."""
    h = 0x811C9DC5
    for w in np.ascontiguousarray(arr).view(np.uint32).ravel().tolist():
        h = ((h ^ w) * 0x01000193) & 0xFFFFFFFF
    return h


# ---------------------------------------------------------------- video oracle wrappers
def _bind_video(L):
    if getattr(L, "_video_bound", False):
        return
    L.ea1o_inv_txfm_1d.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
    L.ea1o_itx_add.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.ea1o_decode_frame.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_uint] + [ctypes.c_void_p] * 4
    L.ea1o_wedge_mask.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    L.ea1o_wedge_mask.restype = None
    L.ea1o_strip_recon.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.ea1o_strip_filter.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    L._video_bound = True


def oracle_inv_txfm_1d(kind, n, vec):
    L = oracle()
    _bind_video(L)
    a = np.array(vec, np.int32)
    rc = L.ea1o_inv_txfm_1d(kind, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return a if rc == 0 else None


def oracle_itx_add(dst, w, h, txtp, coef):
    """dst: 2-D uint8 array (modified copy returned); coef: min(w,32)*min(h,32) int16 row-major."""
    L = oracle()
    _bind_video(L)
    out = np.ascontiguousarray(dst, np.uint8).copy()
    coef = np.ascontiguousarray(coef, np.int16)
    assert coef.size == min(w, 32) * min(h, 32)
    rc = L.ea1o_itx_add(out.ctypes.data, out.shape[1], w, h, txtp, coef.ctypes.data)
    assert rc == 0
    return out


def oracle_set_upscale_filter(taps):
    """taps: int16 array [64][8] (the Upscale_Filter the super-resolution of the oracle uses), None = all zero"""
    L = oracle()
    L.ea1o_set_upscale_filter.argtypes = [ctypes.c_void_p]
    L.ea1o_set_upscale_filter.restype = None
    if taps is None:
        L.ea1o_set_upscale_filter(None)
    else:
        a = np.ascontiguousarray(taps, np.int16)
        assert a.shape == (64, 8)
        L.ea1o_set_upscale_filter(a.ctypes.data)


def oracle_set_chroma_format(ssx=1, ssy=1):
    """subsampling of the chroma planes for the next oracle_decode_frame calls of this thread (4:2:0 = the default 1, 1; 4:2:2: 1, 0; 4:4:4: 0, 0)"""
    L = oracle()
    L.ea1o_set_chroma_format.argtypes = [ctypes.c_int, ctypes.c_int]
    L.ea1o_set_chroma_format.restype = None
    L.ea1o_set_chroma_format(ssx, ssy)
    L.ea1o_grain_set_chroma_format.argtypes = [ctypes.c_int, ctypes.c_int]
    L.ea1o_grain_set_chroma_format.restype = None
    L.ea1o_grain_set_chroma_format(ssx, ssy)
    global _CHROMA
    _CHROMA = (ssx, ssy)


_CHROMA = (1, 1)


def oracle_decode_frame(frame, refs, stages=15, ref_crops=None):
    """frame: easyav1_amd.ir.ParsedFrame; refs: list of 8 entries, each None or [y, u, v] uint8 arrays; ref_crops: per slot the
    crop byte (ea1_frame_desc.crop layout) of the picture it holds, None = references have the frame's own crop.

    Returns dict stage -> [y, u, v] for 'recon', 'deblock', 'cdef', 'final'."""
    L = oracle()
    _bind_video(L)
    d = frame.desc
    W, H = d.width, d.height
    from easyav1_amd.ir import NUM_REF_SLOTS as NS
    ref_arr = (ctypes.c_void_p * (3 * NS))()
    keep = []
    sizes = (ctypes.c_uint32 * (2 * NS))()   # references of another size than the frame (scaled references)
    any_sized = False
    for i in range(NS):
        if refs is not None and i < len(refs) and refs[i] is not None:
            planes = [np.ascontiguousarray(p, np.uint8) for p in refs[i]]
            rh, rw = planes[0].shape
            assert planes[1].shape == (rh >> _CHROMA[1], rw >> _CHROMA[0]) and planes[2].shape == planes[1].shape
            if (rh, rw) != (H, W):
                sizes[2 * i], sizes[2 * i + 1] = rw, rh
                any_sized = True
            keep.append(planes)
            for pl in range(3):
                ref_arr[i * 3 + pl] = planes[pl].ctypes.data
    L.ea1o_set_ref_sizes.argtypes = [ctypes.c_void_p]
    L.ea1o_set_ref_sizes.restype = None
    L.ea1o_set_ref_sizes(ctypes.addressof(sizes) if any_sized else None)
    L.ea1o_set_ref_crops.argtypes = [ctypes.c_void_p]
    L.ea1o_set_ref_crops.restype = None
    if ref_crops is not None:
        crops = (ctypes.c_int * NS)(*[int(c) for c in (list(ref_crops) + [0] * NS)[:NS]])
        L.ea1o_set_ref_crops(ctypes.addressof(crops))
    outs = {}
    ptrs = []
    for name in ("recon", "deblock", "cdef", "final"):
        # (super-resolution: the finished picture is (upscaled_width + 7) & ~7 samples wide)
        Wn = ((d.upscaled_width + 7) & ~7) if (name == "final" and d.upscaled_width) else W
        planes = [np.zeros((H, Wn), np.uint8), np.zeros((H >> _CHROMA[1], Wn >> _CHROMA[0]), np.uint8), np.zeros((H >> _CHROMA[1], Wn >> _CHROMA[0]), np.uint8)]
        outs[name] = planes
        arr = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in planes])
        ptrs.append(arr)
    lr = frame.lr_pointers()
    rc = L.ea1o_decode_frame(ctypes.addressof(d), frame.blocks.ctypes.data, frame.txs.ctypes.data,
                             frame.coefs.ctypes.data, frame.aux.ctypes.data if len(frame.aux) else None,
                             frame.cdef_idx.ctypes.data, ctypes.addressof(lr), ctypes.addressof(ref_arr), stages,
                             ctypes.addressof(ptrs[0]), ctypes.addressof(ptrs[1]), ctypes.addressof(ptrs[2]),
                             ctypes.addressof(ptrs[3]))
    L.ea1o_set_ref_sizes(None)
    L.ea1o_set_ref_crops(None)
    assert rc == 0
    return outs


def _ref_array(refs, W, H):
    from easyav1_amd.ir import NUM_REF_SLOTS as NS
    ref_arr = (ctypes.c_void_p * (3 * NS))()
    keep = []
    for i in range(NS):
        if refs is not None and i < len(refs) and refs[i] is not None:
            planes = [np.ascontiguousarray(p, np.uint8) for p in refs[i]]
            assert planes[0].shape == (H, W) and planes[1].shape == (H // 2, W // 2)
            keep.append(planes)
            for pl in range(3):
                ref_arr[i * 3 + pl] = planes[pl].ctypes.data
    return ref_arr, keep


def oracle_strip_recon(frame, refs, x0, x1, work):
    """tile-column split, phase 1 of one rank on the CPU: reconstructs the blocks of luma columns [x0, x1) INTO `work`
    ([y, u, v] C-contiguous uint8 arrays of the whole frame); all other columns of `work` stay as they are"""
    L = oracle()
    _bind_video(L)
    d = frame.desc
    ref_arr, keep = _ref_array(refs, d.width, d.height)
    assert all(p.flags.c_contiguous and p.dtype == np.uint8 for p in work)
    wp = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in work])
    rc = L.ea1o_strip_recon(ctypes.addressof(d), frame.blocks.ctypes.data, frame.txs.ctypes.data, frame.coefs.ctypes.data,
                            frame.aux.ctypes.data if len(frame.aux) else None, ctypes.addressof(ref_arr), x0, x1, ctypes.addressof(wp))
    assert rc == 0


def oracle_strip_filter(frame, work, stages=15):
    """phase 2: the in-loop filters over the whole of `work` (reconstruction incl. halos; anything elsewhere) -> [y, u, v]"""
    L = oracle()
    _bind_video(L)
    d = frame.desc
    W, H = d.width, d.height
    out = [np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)]
    wp = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in work])
    op = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in out])
    lr = frame.lr_pointers()
    rc = L.ea1o_strip_filter(ctypes.addressof(d), frame.blocks.ctypes.data, frame.txs.ctypes.data, frame.cdef_idx.ctypes.data,
                             ctypes.addressof(lr), stages, ctypes.addressof(wp), ctypes.addressof(op))
    assert rc == 0
    return out


def oracle_wedge_mask(w, h, index, sign):
    L = oracle()
    _bind_video(L)
    out = np.zeros((h, w), np.uint8)
    L.ea1o_wedge_mask(w, h, index, sign, out.ctypes.data)
    return out


def oracle_synth_from_spectra(channels, bs0, bs1, flags, blocks):
    """spectra blocks ((channels, bs[W]) arrays whose first half holds the coefficients, the layout
    mapping0_inverse hands to mdct_backward) -> PCM (channels, total) through the CPU oracle"""
    mdct = {0: OracleMdct(bs0), 1: OracleMdct(bs1)}
    bs = (bs0, bs1)
    time = [mdct[W].backward(np.ascontiguousarray(b[:, :bs[W] // 2])) for W, b in zip(flags, blocks)]
    outs = oracle_synth_stream(channels, bs0, bs1, flags, time)
    return np.concatenate(outs, axis=1) if outs else np.zeros((channels, 0), np.float32)


def oracle_set_warp_filter(table):
    """Hand the checker the 193 x 8 warp filter table (the same one the library gets)."""
    L = oracle()
    _bind_video(L)
    L.ea1o_set_warp_filter.argtypes = [ctypes.c_void_p]
    L.ea1o_set_warp_filter.restype = None
    if table is None:           # back to the built-in table of include/ea1_av1_tables.inc
        L.ea1o_set_warp_filter(None)
        return
    t = np.ascontiguousarray(table, np.int8)
    assert t.shape == (193, 8)
    L.ea1o_set_warp_filter(t.ctypes.data)


def oracle_setup_shear(mat):
    """-> (valid, [alpha, beta, gamma, delta]) of a 6-entry warp matrix (AV1 spec 7.11.3.6)."""
    L = oracle()
    _bind_video(L)
    m = (ctypes.c_int32 * 6)(*[int(v) for v in mat])
    sh = (ctypes.c_int * 4)()
    L.ea1o_setup_shear.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    valid = L.ea1o_setup_shear(m, sh)
    return bool(valid), list(sh)


def oracle_set_gaussian_sequence(seq):
    L = oracle()
    s = np.ascontiguousarray(seq, np.int16)
    assert s.shape == (2048,)
    L.ea1o_set_gaussian_sequence.argtypes = [ctypes.c_void_p]
    L.ea1o_set_gaussian_sequence.restype = None
    L.ea1o_set_gaussian_sequence(s.ctypes.data)


def oracle_film_grain(params, planes):
    """params: easyav1_amd.grain.FilmGrain; planes: [y, u, v] uint8 -> grained [y, u, v]"""
    L = oracle()
    src = [np.ascontiguousarray(p, np.uint8) for p in planes]
    H, W = src[0].shape
    out = [np.zeros_like(p) for p in src]
    ip = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in src])
    op = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in out])
    L.ea1o_film_grain.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert L.ea1o_film_grain(ctypes.byref(params), W, H, ip, op) == 0
    return out
