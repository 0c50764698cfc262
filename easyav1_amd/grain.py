"""Film grain synthesis through the C ABI (include/easyav1_amd_grain.h): harness-side wrapper."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check


class FilmGrain(ctypes.Structure):
    """ea1_film_grain"""
    _fields_ = [("grain_seed", ctypes.c_uint16), ("num_y_points", ctypes.c_uint8), ("point_y", (ctypes.c_uint8 * 2) * 14),
                ("chroma_scaling_from_luma", ctypes.c_uint8), ("num_uv_points", ctypes.c_uint8 * 2),
                ("point_uv", ((ctypes.c_uint8 * 2) * 10) * 2), ("scaling_shift", ctypes.c_uint8), ("ar_coeff_lag", ctypes.c_uint8),
                ("ar_coeffs_y", ctypes.c_int8 * 24), ("ar_coeffs_uv", (ctypes.c_int8 * 25) * 2), ("ar_coeff_shift", ctypes.c_uint8),
                ("grain_scale_shift", ctypes.c_uint8), ("uv_mult", ctypes.c_uint8 * 2), ("uv_luma_mult", ctypes.c_uint8 * 2),
                ("uv_offset", ctypes.c_uint16 * 2), ("overlap_flag", ctypes.c_uint8), ("clip_to_restricted_range", ctypes.c_uint8),
                ("matrix_is_identity", ctypes.c_uint8), ("crop", ctypes.c_uint8)]


def random_params(rng, **fixed):
    """Random but legal film_grain_params() (numpy Generator); keyword arguments pin fields."""
    fg = FilmGrain()
    fg.grain_seed = int(rng.integers(1 << 16))

    def points(n, arr):
        vals = np.sort(rng.choice(256, size=n, replace=False))
        for i in range(n):
            arr[i][0] = int(vals[i])
            arr[i][1] = int(rng.integers(256))
    fg.num_y_points = int(rng.integers(0, 15))
    points(fg.num_y_points, fg.point_y)
    fg.chroma_scaling_from_luma = int(rng.integers(2))
    for p in range(2):
        fg.num_uv_points[p] = 0 if fg.chroma_scaling_from_luma else int(rng.integers(0, 11))
        points(fg.num_uv_points[p], fg.point_uv[p])
    fg.scaling_shift = int(rng.integers(8, 12))
    fg.ar_coeff_lag = int(rng.integers(4))
    for i in range(24):
        fg.ar_coeffs_y[i] = int(rng.integers(-128, 128))
    for p in range(2):
        for i in range(25):
            fg.ar_coeffs_uv[p][i] = int(rng.integers(-128, 128))
    fg.ar_coeff_shift = int(rng.integers(6, 10))
    fg.grain_scale_shift = int(rng.integers(4))
    for p in range(2):
        fg.uv_mult[p] = int(rng.integers(256))
        fg.uv_luma_mult[p] = int(rng.integers(256))
        fg.uv_offset[p] = int(rng.integers(512))
    fg.overlap_flag = int(rng.integers(2))
    fg.clip_to_restricted_range = int(rng.integers(2))
    fg.matrix_is_identity = int(rng.integers(2))
    for k, v in fixed.items():
        setattr(fg, k, v)
    return fg


def synth_gaussian_sequence():
    """the stand-in Gaussian sequence of tests/native/easyav1_amd_synth.h (NOT the AV1 table)"""
    L = _lib.synth_lib()
    t = np.zeros(2048, np.int16)
    L.ea1_synth_gaussian_sequence.argtypes = [ctypes.c_void_p]
    L.ea1_synth_gaussian_sequence.restype = None
    L.ea1_synth_gaussian_sequence(t.ctypes.data)
    return t


def set_gaussian_sequence(ctx, seq):
    L = _lib.lib()
    s = np.ascontiguousarray(seq, np.int16)
    assert s.shape == (2048,)
    L.ea1_recon_set_gaussian_sequence.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    check(L.ea1_recon_set_gaussian_sequence(ctx._h, s.ctypes.data), "ea1_recon_set_gaussian_sequence")


def film_grain_batch(ctx, slots, params, stream=None):
    """slots[c] (-1: idle) / params[c] (FilmGrain or None) per chain: one launch for all chains"""
    L = _lib.lib()
    n = ctx.nchains
    assert len(slots) == n and len(params) == n
    sl = (ctypes.c_int * n)(*[int(s) for s in slots])
    pp = (ctypes.c_void_p * n)(*[ctypes.addressof(p) if p is not None else None for p in params])
    L.ea1_recon_film_grain_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    check(L.ea1_recon_film_grain_batch(ctx._h, sl, pp, stream), "ea1_recon_film_grain_batch")


def film_grain(ctx, slot, params, chain=0, stream=None):
    L = _lib.lib()
    L.ea1_recon_film_grain_chain.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    check(L.ea1_recon_film_grain_chain(ctx._h, chain, slot, ctypes.byref(params), stream), "ea1_recon_film_grain_chain")


def download_display(ctx, chain=0, stream=None):
    L = _lib.lib()
    W, H = ctx.width, ctx.height
    y = np.zeros((H, W), np.uint8)
    ssx, ssy = getattr(ctx, "ssx", 1), getattr(ctx, "ssy", 1)
    u = np.zeros((H >> ssy, W >> ssx), np.uint8)
    v = np.zeros((H >> ssy, W >> ssx), np.uint8)
    L.ea1_recon_download_display_chain.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    check(L.ea1_recon_download_display_chain(ctx._h, chain, y.ctypes.data, W, u.ctypes.data, v.ctypes.data, W >> ssx, stream),
          "ea1_recon_download_display_chain")
    return [y, u, v]
