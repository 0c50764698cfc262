"""Generates tests/golden/audio_golden.npz from the REFERENCE build (oracle/_ref).

Run in the build container (needs /root/reference):  make -C oracle && python tests/golden/make_audio_golden.py
The vectors are inputs + the reference's outputs only (data, no reference source):
  mdct_in_N / mdct_out_N   seeded spectra and mdct_backward (minivorbis.h:7054) results, 8 sizes
  window_N                 _vorbis_window_get tables (minivorbis.h:11992)
  ola_*                    block sequences pushed through mdct_backward + vorbis_synthesis_blockin
                           /pcmout/read (+restart) exactly as easyav1's decode_audio does
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import oracle_lib as ol  # noqa: E402

SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]


def main():
    ref = ol.ref_minivorbis()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    out = {}
    rng = np.random.default_rng(20261003)
    for idx, n in enumerate(SIZES):
        n2 = n // 2
        rows = [np.zeros(n2, np.float32)]
        for k in (0, n2 - 1):
            e = np.zeros(n2, np.float32)
            e[k] = 1.0
            rows.append(e)
        rows.append((rng.standard_normal(n2) * 1e-39).astype(np.float32))
        env = 1.0 / (1.0 + np.arange(n2) / 8.0)
        for _ in range(4):
            s = rng.laplace(scale=1.0, size=n2) * env
            s[int(0.8 * n2):] = 0
            rows.append(s.astype(np.float32))
        spec = np.stack(rows)
        res = np.zeros((spec.shape[0], n), np.float32)
        h = ref.ref_mdct_new(n)
        for i in range(spec.shape[0]):
            res[i, :n2] = spec[i]
            ref.ref_mdct_backward(h, ol.fptr(res[i]), ol.fptr(res[i]))
        ref.ref_mdct_free(h)
        out[f"mdct_in_{n}"] = spec
        out[f"mdct_out_{n}"] = res
        out[f"window_{n}"] = np.ctypeslib.as_array(ref.ref_window(idx), shape=(n2,)).copy()

    cases = {
        "stereo_256_2048": (2, 256, 2048, [1, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1], []),
        "six_256_2048_restart": (6, 256, 2048, [0, 1, 1, 0, 1, 1, 0, 0, 1, 1], [4]),
        "mono_64_64": (1, 64, 64, [0] * 9, []),
        "tri_128_1024": (3, 128, 1024, [1, 0, 1, 0, 0, 1, 1], [2, 5]),
    }
    for tag, (ch, bs0, bs1, flags, restart) in cases.items():
        spectra = np.zeros((len(flags), ch, bs1 // 2), np.float32)
        h = ref.ref_synth_new(ch, bs0, bs1)
        m = {0: ref.ref_mdct_new(bs0), 1: ref.ref_mdct_new(bs1)}
        buf = np.zeros((ch, bs1), np.float32)
        pcm = []
        for k, W in enumerate(flags):
            n = (bs0, bs1)[W]
            s = (rng.laplace(scale=1.0, size=(ch, n // 2)) / (1.0 + np.arange(n // 2) / 8.0)).astype(np.float32)
            spectra[k, :, : n // 2] = s
            blk = np.zeros((ch, n), np.float32)
            blk[:, : n // 2] = s
            for c in range(ch):
                ref.ref_mdct_backward(m[W], ol.fptr(blk[c]), ol.fptr(blk[c]))
            if k in restart:
                ref.ref_synth_restart(h)
            assert ref.ref_synth_blockin(h, W, ol.fptr(blk)) == 0
            cnt = ref.ref_synth_pcmout(h, ol.fptr(buf), bs1)
            pcm.append(buf[:, :cnt].copy())
            ref.ref_synth_read(h, cnt)
        ref.ref_synth_free(h)
        for v in m.values():
            ref.ref_mdct_free(v)
        out[f"ola_cfg_{tag}"] = np.array([ch, bs0, bs1], np.int32)
        out[f"ola_flags_{tag}"] = np.array(flags, np.int32)
        out[f"ola_restart_{tag}"] = np.array(restart, np.int32)
        out[f"ola_spec_{tag}"] = spectra
        out[f"ola_pcm_{tag}"] = np.concatenate(pcm, axis=1)
    path = os.path.join(HERE, "audio_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
