"""What the host link of this box delivers: pinned host <-> device copies of several sizes (torch streams)."""
import time
import torch
torch.cuda.init()
for mb in (0.5, 2, 8, 64):
    n = int(mb * 1e6)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = max(8, int(2e9 / n))
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{name} {mb:5.1f} MB x {reps}: {n * reps / dt / 1e9:6.2f} GB/s")
# two directions at once, and two streams in one direction
n = int(16e6)
h1, h2 = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
d1, d2 = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for tag, f1, f2 in (("H2D + H2D on two streams", lambda: d1.copy_(h1, non_blocking=True), lambda: d2.copy_(h2, non_blocking=True)),
                    ("H2D + D2H on two streams", lambda: d1.copy_(h1, non_blocking=True), lambda: h2.copy_(d2, non_blocking=True))):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        with torch.cuda.stream(s1):
            f1()
        with torch.cuda.stream(s2):
            f2()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{tag}: {2 * n * 100 / dt / 1e9:6.2f} GB/s in total")
# large working sets: 32 distinct 12.5 MB pictures per round into an 800 MB pinned area (what with_d2h does), alone and
# beside a compute kernel that keeps the device busy
n = 12_503_040
host = torch.empty(64 * n, dtype=torch.uint8).pin_memory()
dev = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
busy = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
cs, ks = torch.cuda.Stream(), torch.cuda.Stream()
for beside in (False, True):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for rep in range(6):
        with torch.cuda.stream(cs):
            for k in range(32):
                host[((rep & 1) * 32 + k) * n:((rep & 1) * 32 + k + 1) * n].copy_(dev[k * n:(k + 1) * n], non_blocking=True)
        if beside:
            with torch.cuda.stream(ks):
                for _ in range(20):
                    busy.mul_(1.0001)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"D2H 32 x 12.5 MB pictures per round into an 800 MB pinned area{' beside a streaming kernel' if beside else ''}: {6 * 32 * n / dt / 1e9:6.2f} GB/s")
