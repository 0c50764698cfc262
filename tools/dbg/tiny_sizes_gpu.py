"""Pictures of unusual sizes (1x1 ... 130x2, every chroma layout) encoded by libaom now, through easyav1_* on the GPU, against dav1d's pictures.
    python tools/dbg/tiny_sizes_gpu.py"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
torch.zeros(1, device="cuda")
from tests import test_real_av1 as T
from tests.golden import av1_ref
from easyav1_amd import easyav1 as ea
ea.api()
rng = np.random.default_rng(5)
sizes = [(1, 1), (2, 2), (3, 5), (7, 5), (5, 16), (64, 1), (1, 64), (4, 128), (130, 2), (8, 8), (9, 9), (15, 17), (65, 65), (127, 3), (129, 129), (255, 9)]
bad = 0
for (w, h) in sizes:
    for sub in ("4:2:0", "4:4:4", "4:2:2", "4:0:0"):
        frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(4)]
        if (w * h) % 3 == 0:      # smoother content now and then
            frames = [np.clip(np.add.outer(np.arange(h) * 3, np.arange(w) * 5)[:, :, None] + 20 * k + np.zeros(3), 0, 255).astype(np.uint8) for k in range(4)]
        try:
            units = av1_ref.encode_sequence(frames, subsampling=sub, quality=70)
            shown = av1_ref.dav1d_pictures(units)
        except Exception as ex:
            print(w, h, sub, "encode / dav1d failed:", repr(ex)[:120])
            continue
        layout = {"4:0:0": 0, "4:2:0": 1, "4:2:2": 2, "4:4:4": 3}[sub]
        fx = dict(name="tiny", width=w, height=h, layout=layout, units=units)
        try:
            p = ea.Player(T.mux_webm(fx), enable_audio=0, skip_unprocessed_frames=0, log_level=ea.LOG_ERROR)
            k, wrong = 0, 0
            while p.decode_next() == ea.STATUS_OK:
                if not p.has_video_frame():
                    continue
                ts, planes, props = p.video_frame()
                want = shown[k]
                got = [x for x in planes if x is not None]
                ok = len(got) == (1 if layout == 0 else 3) and all(g.shape == np.asarray(q).shape and np.array_equal(g, q) for g, q in zip(got, want))
                wrong += not ok
                k += 1
            st = p.get_status()
            p.close()
            res = "OK" if (wrong == 0 and k == len(shown) and st != ea.STATUS_ERROR) else "FAILED"
            bad += res != "OK"
            print(w, h, sub, res, "pictures", k, "of", len(shown), "wrong", wrong, "status", st, flush=True)
        except Exception as ex:
            bad += 1
            print(w, h, sub, "EXCEPTION", repr(ex)[:200], flush=True)
print("failures:", bad)
