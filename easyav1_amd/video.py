"""Host-side mirror of the decoder interface easyav1 drives (dav1d_open / send_data / get_picture /
flush / close, reference src/easyav1.c:1405, :2390-2403, :3059, :3969) on top of the C ABI.

A `ReconContext` is one decoding chain (its reference slots live in HBM); `submit(job)`
reconstructs one parsed frame into its destination slot.  torch is used only for the current
stream and for device scratch in the DSP-level helper.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ir
from ._lib import check


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Job:
    """One parsed frame resident in HBM (ea1_job)."""

    def __init__(self, ctx, frame):
        self._h = ctypes.c_void_p()
        self.desc = frame.desc
        L = _lib.lib()
        lr = frame.lr_pointers()
        check(L.ea1_job_create(ctx._h, ctypes.byref(frame.desc), frame.blocks.ctypes.data if len(frame.blocks) else None,
                               frame.txs.ctypes.data if len(frame.txs) else None,
                               frame.coefs.ctypes.data if len(frame.coefs) else None,
                               frame.aux.ctypes.data if len(frame.aux) else None,
                               frame.cdef_idx.ctypes.data if len(frame.cdef_idx) else None, ctypes.addressof(lr),
                               ctypes.byref(self._h)), "ea1_job_create")

    @classmethod
    def strip(cls, ctx, frame, x_begin, x_end):
        """the job of ONE rank of a tile-column split: reconstructs only the blocks in luma columns [x_begin, x_end)"""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.desc = frame.desc
        L = _lib.lib()
        L.ea1_job_create_strip.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        lr = frame.lr_pointers()
        check(L.ea1_job_create_strip(ctx._h, ctypes.addressof(frame.desc), frame.blocks.ctypes.data if len(frame.blocks) else None,
                                     frame.txs.ctypes.data if len(frame.txs) else None,
                                     frame.coefs.ctypes.data if len(frame.coefs) else None,
                                     frame.aux.ctypes.data if len(frame.aux) else None,
                                     frame.cdef_idx.ctypes.data if len(frame.cdef_idx) else None, ctypes.addressof(lr),
                                     x_begin, x_end, ctypes.byref(self._h)), "ea1_job_create_strip")
        return self

    def stats(self):
        r = ctypes.c_double()
        c = ctypes.c_uint64()
        m = ctypes.c_uint64()
        b = ctypes.c_uint64()
        check(_lib.lib().ea1_job_stats(self._h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(m), ctypes.byref(b)), "ea1_job_stats")
        return {"refs_per_pixel": r.value, "coef_count": c.value, "block_bytes": m.value, "algorithmic_bytes": b.value}

    def schedule_info(self):
        nl, mi, ni = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        mc = (ctypes.c_int * 3)()
        tx = (ctypes.c_int * 19)()
        check(_lib.lib().ea1_job_schedule_info(self._h, ctypes.byref(nl), ctypes.byref(mi), ctypes.byref(ni), mc, tx),
              "ea1_job_schedule_info")
        return {"num_levels": nl.value, "max_level_items": mi.value, "intra_items": ni.value, "mc_tiles": list(mc),
                "tx_blocks": list(tx)}

    def close(self):
        if self._h:
            _lib.lib().ea1_job_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReconContext:
    """`nchains` independent decoding chains; submit_batch() reconstructs one frame of every chain
    per kernel launch (closed GOPs are independent: GOP k goes to chain k % nchains)."""

    def __init__(self, width, height, device=0, nchains=1, layout=ir.LAYOUT_420):
        """layout: ir.LAYOUT_420 / _444 / _422, the chroma layout of every picture of the context (ea1_frame_desc.layout of its frames)"""
        self._h = ctypes.c_void_p()
        self.width, self.height, self.nchains, self.layout = width, height, nchains, layout
        self.ssx, self.ssy = (0 if layout == ir.LAYOUT_444 else 1), (1 if layout == ir.LAYOUT_420 else 0)
        self.device = torch.device("cuda", device)
        check(_lib.lib().ea1_recon_create_layout(ctypes.byref(self._h), width, height, nchains, device, layout), "ea1_recon_create_layout")

    def close(self):
        if self._h:
            _lib.lib().ea1_recon_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_warp_filter(self, table):
        """The 193 x 8 int8 warp filter table (ea1_recon_set_warp_filter): needed before jobs with warped blocks."""
        t = np.ascontiguousarray(table, dtype=np.int8)
        assert t.shape == (193, 8)
        L = _lib.lib()
        L.ea1_recon_set_warp_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(L.ea1_recon_set_warp_filter(self._h, t.ctypes.data), "ea1_recon_set_warp_filter")

    def flush(self):
        check(_lib.lib().ea1_recon_flush(self._h), "ea1_recon_flush")

    def set_ref(self, slot, planes, chain=0, crop=0, pad=0):
        """load a picture into a reference slot; its size may be any multiple of 8 up to the context's (scaled references);
        crop: ea1_frame_desc.crop layout, the picture's visible size below its coded size"""
        y, u, v = (np.ascontiguousarray(p, np.uint8) for p in planes)
        h, w = y.shape
        assert u.shape == (h >> self.ssy, w >> self.ssx) == v.shape
        L = _lib.lib()
        L.ea1_recon_set_ref_sized.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        check(L.ea1_recon_set_ref_sized(self._h, chain, slot, w, h, y.ctypes.data, y.shape[1], u.ctypes.data, v.ctypes.data, u.shape[1]), "ea1_recon_set_ref")
        if pad:
            L.ea1_recon_set_ref_pad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint8, ctypes.c_uint8]
            check(L.ea1_recon_set_ref_pad(self._h, chain, slot, pad, crop), "ea1_recon_set_ref_pad")
        elif crop:
            L.ea1_recon_set_ref_crop.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint8]
            check(L.ea1_recon_set_ref_crop(self._h, chain, slot, crop), "ea1_recon_set_ref_crop")

    def set_upscale_filter(self, taps):
        """taps: int16 [64][8], the Upscale_Filter of super-resolution (the library does not own the table)"""
        a = np.ascontiguousarray(taps, np.int16)
        assert a.shape == (64, 8)
        L = _lib.lib()
        L.ea1_recon_set_upscale_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ea1_recon_set_upscale_filter.restype = ctypes.c_int
        check(L.ea1_recon_set_upscale_filter(self._h, a.ctypes.data), "ea1_recon_set_upscale_filter")

    def make_job(self, frame):
        return Job(self, frame)

    def make_jobs(self, frames, threads=1):
        """ea1_job_create_batch: the hand-over of several parsed frames on `threads` host threads of the library"""
        n = len(frames)
        structs = [ir.ir_frame_struct(f) for f in frames]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.addressof(s) for s in structs])
        out = (ctypes.c_void_p * n)()
        L = _lib.lib()
        L.ea1_job_create_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        check(L.ea1_job_create_batch(self._h, ptrs, n, threads, out), "ea1_job_create_batch")
        jobs = []
        for k in range(n):
            j = Job.__new__(Job)
            j._h = ctypes.c_void_p(out[k])
            j.desc = frames[k].desc
            jobs.append(j)
        return jobs

    def submit(self, job, stages=ir.STAGE_ALL):
        check(_lib.lib().ea1_recon_submit_stages(self._h, job._h, stages, _stream_ptr(self.device)), "ea1_recon_submit")

    def submit_batch(self, jobs, stages=ir.STAGE_ALL):
        """jobs: one Job (or None = idle) per chain."""
        assert len(jobs) == self.nchains
        arr = (ctypes.c_void_p * self.nchains)(*[j._h if j is not None else None for j in jobs])
        check(_lib.lib().ea1_recon_submit_batch(self._h, arr, stages, _stream_ptr(self.device)), "ea1_recon_submit_batch")

    # ---- tile-column split (include/easyav1_amd_video.h): phases + column exchange buffers ----
    PIC_WORK = -1

    def submit_phase(self, jobs, phase):
        """phase 1: reconstruction of the jobs' strip; phase 2: in-loop filters of the strip (after the halo exchange)"""
        assert len(jobs) == self.nchains
        arr = (ctypes.c_void_p * self.nchains)(*[j._h if j is not None else None for j in jobs])
        L = _lib.lib()
        L.ea1_recon_submit_phase.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
        check(L.ea1_recon_submit_phase(self._h, arr, phase, _stream_ptr(self.device)), "ea1_recon_submit_phase")

    def columns_bytes(self, w):
        L = _lib.lib()
        L.ea1_recon_columns_bytes.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.ea1_recon_columns_bytes.restype = ctypes.c_size_t
        return L.ea1_recon_columns_bytes(self._h, w)

    def get_columns(self, pic, x, w, out, chain=0):
        """luma columns [x, x + w) + their chroma of a picture (PIC_WORK or a reference slot) -> `out`, a cuda uint8 tensor"""
        assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= self.columns_bytes(w)
        L = _lib.lib()
        L.ea1_recon_get_columns.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        check(L.ea1_recon_get_columns(self._h, chain, pic, x, w, out.data_ptr(), _stream_ptr(self.device)), "ea1_recon_get_columns")
        return out

    def put_columns(self, pic, x, w, buf, chain=0):
        assert buf.is_cuda and buf.dtype == torch.uint8 and buf.is_contiguous() and buf.numel() >= self.columns_bytes(w)
        L = _lib.lib()
        L.ea1_recon_put_columns.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        check(L.ea1_recon_put_columns(self._h, chain, pic, x, w, buf.data_ptr(), _stream_ptr(self.device)), "ea1_recon_put_columns")

    def picture_layout(self):
        """-> (plane offsets, plane strides, bytes) of a picture in device layout"""
        off = (ctypes.c_size_t * 3)()
        st = (ctypes.c_size_t * 3)()
        n = ctypes.c_size_t()
        L = _lib.lib()
        L.ea1_recon_picture_layout.argtypes = [ctypes.c_void_p] * 4
        check(L.ea1_recon_picture_layout(self._h, off, st, ctypes.byref(n)), "ea1_recon_picture_layout")
        return list(off), list(st), n.value

    def download_batch_async(self, slots, host_ptrs, stream):
        """enqueue one device-layout copy per chain (slots[c] < 0: skip) to the host addresses; no synchronisation"""
        n = self.nchains
        sl = (ctypes.c_int * n)(*[int(s) for s in slots])
        pp = (ctypes.c_void_p * n)(*host_ptrs)
        L = _lib.lib()
        L.ea1_recon_download_batch_async.argtypes = [ctypes.c_void_p] * 4
        check(L.ea1_recon_download_batch_async(self._h, sl, pp, stream), "ea1_recon_download_batch_async")

    def profile(self, on=True):
        """Bracket every kernel launch of later submits with HIP events (see profile_read)."""
        check(_lib.lib().ea1_recon_profile_enable(self._h, int(on)), "ea1_recon_profile_enable")

    def profile_read(self):
        """-> list of dicts {name, ms_total, launches, algorithmic_bytes}; synchronises and resets."""
        from ._lib_video import ProfEntry
        arr = (ProfEntry * 64)()
        n = ctypes.c_int(0)
        check(_lib.lib().ea1_recon_profile_read(self._h, arr, 64, ctypes.byref(n)), "ea1_recon_profile_read")
        return [{"name": arr[i].name.decode(), "ms_total": arr[i].ms_total, "launches": int(arr[i].launches),
                 "algorithmic_bytes": int(arr[i].algorithmic_bytes)} for i in range(n.value)]

    def picture(self, slot, chain=0):
        p = ir.Picture()
        check(_lib.lib().ea1_recon_get_picture_chain(self._h, chain, slot, ctypes.byref(p)), "ea1_recon_get_picture")
        return p

    def slot_size(self, slot, chain=0):
        """-> (width, height) of the picture the slot holds (frames may be smaller than the context)"""
        L = _lib.lib()
        L.ea1_recon_slot_size.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        w, h = ctypes.c_uint32(0), ctypes.c_uint32(0)
        check(L.ea1_recon_slot_size(self._h, chain, slot, ctypes.byref(w), ctypes.byref(h)), "ea1_recon_slot_size")
        return w.value, h.value

    def download(self, slot, chain=0):
        w, h = self.slot_size(slot, chain)
        y = np.empty((h, w), np.uint8)
        u = np.empty((h >> self.ssy, w >> self.ssx), np.uint8)
        v = np.empty_like(u)
        check(_lib.lib().ea1_recon_download_chain(self._h, chain, slot, y.ctypes.data, y.shape[1], u.ctypes.data,
                                                  v.ctypes.data, u.shape[1], _stream_ptr(self.device)), "ea1_recon_download")
        return [y, u, v]


def dsp_itx_add(plane, w, h, coefs, pos, txtp):
    """dav1d's itxfm_add in batch form on a cuda uint8 2-D tensor (modified in place).

    coefs: cuda int16 [count, min(w,32)*min(h,32)]; pos: (count, 2) x,y; txtp: (count,)"""
    assert plane.is_cuda and plane.dtype == torch.uint8 and plane.dim() == 2 and plane.stride(1) == 1
    assert coefs.is_cuda and coefs.dtype == torch.int16 and coefs.is_contiguous()
    pos = np.ascontiguousarray(pos, np.uint32)
    txtp = np.ascontiguousarray(txtp, np.uint8)
    check(_lib.lib().ea1_dsp_itx_add(ctypes.c_void_p(plane.data_ptr()), plane.stride(0), w, h,
                                     ctypes.c_void_p(coefs.data_ptr()), pos.ctypes.data, txtp.ctypes.data, len(txtp),
                                     plane.device.index or 0, _stream_ptr(plane.device)), "ea1_dsp_itx_add")
    return plane


class PinnedBuffer:
    """pinned host memory from the library's HIP runtime (ea1_host_alloc)"""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.nbytes = nbytes
        L = _lib.lib()
        L.ea1_host_alloc.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        check(L.ea1_host_alloc(nbytes, ctypes.byref(self.ptr)), "ea1_host_alloc")

    def array(self):
        return np.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self.ptr.value))

    def close(self):
        if self.ptr:
            L = _lib.lib()
            L.ea1_host_free.argtypes = [ctypes.c_void_p]
            L.ea1_host_free.restype = None
            L.ea1_host_free(self.ptr)
            self.ptr = ctypes.c_void_p()
