"""Film grain pass on the bench workload: 128 chains x 1920x1080, one launch for all chains.
Prints the kernel time from the in-library HIP events (algorithmic bytes = picture read + written =
3 bytes per luma sample... i.e. 2 * 1.5 * W * H per picture) and the host-side cost of the call
(template generation + upload), which is NOT part of the kernel figure."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
torch.cuda.init()
from easyav1_amd import grain, ir, video  # noqa: E402

W, H, S = 1920, 1080 // 8 * 8, int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = video.ReconContext(W, H, nchains=S)
grain.set_gaussian_sequence(ctx, grain.synth_gaussian_sequence())
rng = np.random.default_rng(1)
pic = ir.synth_picture(W, H, 3)
for c in range(S):
    ctx.set_ref(0, pic, chain=c)
params = [grain.random_params(rng, num_y_points=4, overlap_flag=1, chroma_scaling_from_luma=0) for _ in range(S)]
for p in params:
    p.num_uv_points[0] = p.num_uv_points[1] = 3
    for k in range(4):
        p.point_y[k][0] = 40 * k + 10
    for q in range(2):
        for k in range(3):
            p.point_uv[q][k][0] = 60 * k + 20
slots = [0] * S
grain.film_grain_batch(ctx, slots, params)
torch.cuda.synchronize()
ctx.profile(True)
t0 = time.perf_counter()
N = 5
for _ in range(N):
    grain.film_grain_batch(ctx, slots, params)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
prof = [e for e in ctx.profile_read() if e["name"] == "film_grain_kernel"][0]
avg_s = prof["ms_total"] / prof["launches"] * 1e-3
bytes_per_launch = prof["algorithmic_bytes"] / prof["launches"]
print(json.dumps({"kernel": "film_grain_kernel", "pictures_per_launch": S, "size": f"{W}x{H}", "avg_launch_us": round(avg_s * 1e6, 1),
                  "algorithmic_bytes_per_launch": int(bytes_per_launch), "achieved_GBps": round(bytes_per_launch / avg_s / 1e9, 1),
                  "frac_of_8TBps": round(bytes_per_launch / avg_s / 8e12, 4), "pictures_per_s_kernel_only": round(S / avg_s, 1),
                  "host_ms_per_call": round(wall * 1e3, 2), "host_ms_per_picture": round(wall * 1e3 / S, 3)}))
