"""ctypes prototypes of include/easyav1_amd_video.h."""
import ctypes

from .ir import FrameDesc, Picture

c_vp = ctypes.c_void_p


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 56), ("ms_total", ctypes.c_double), ("launches", ctypes.c_uint64),
                ("algorithmic_bytes", ctypes.c_uint64)]


def declare(L):
    L.ea1_recon_create.argtypes = [ctypes.POINTER(c_vp), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    L.ea1_recon_destroy.argtypes = [c_vp]
    L.ea1_recon_destroy.restype = None
    L.ea1_recon_flush.argtypes = [c_vp]
    L.ea1_recon_set_ref.argtypes = [c_vp, ctypes.c_int, c_vp, ctypes.c_size_t, c_vp, c_vp, ctypes.c_size_t]
    L.ea1_job_create.argtypes = [c_vp, ctypes.POINTER(FrameDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_vp)]
    L.ea1_tx_size_index.argtypes = [ctypes.c_int, ctypes.c_int]
    L.ea1_job_destroy.argtypes = [c_vp]
    L.ea1_job_destroy.restype = None
    L.ea1_job_stats.argtypes = [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.ea1_recon_submit.argtypes = [c_vp, c_vp, c_vp]
    L.ea1_recon_submit_stages.argtypes = [c_vp, c_vp, ctypes.c_uint, c_vp]
    L.ea1_recon_get_picture.argtypes = [c_vp, ctypes.c_int, ctypes.POINTER(Picture)]
    L.ea1_recon_download.argtypes = [c_vp, ctypes.c_int, c_vp, ctypes.c_size_t, c_vp, c_vp, ctypes.c_size_t, c_vp]
    L.ea1_dsp_itx_add.argtypes = [c_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int,
                                  ctypes.c_int, c_vp]
    L.ea1_recon_profile_enable.argtypes = [c_vp, ctypes.c_int]
    L.ea1_recon_profile_read.argtypes = [c_vp, ctypes.POINTER(ProfEntry), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.ea1_recon_create_batch.argtypes = [ctypes.POINTER(c_vp), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
    L.ea1_recon_num_chains.argtypes = [c_vp]
    L.ea1_recon_set_ref_chain.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_size_t, c_vp, c_vp, ctypes.c_size_t]
    L.ea1_recon_submit_batch.argtypes = [c_vp, ctypes.POINTER(c_vp), ctypes.c_uint, c_vp]
    L.ea1_recon_get_picture_chain.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Picture)]
    L.ea1_recon_download_chain.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_size_t, c_vp, c_vp,
                                           ctypes.c_size_t, c_vp]
    L.ea1_job_schedule_info.argtypes = [c_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
