"""Summary of a `rocprofv3 --kernel-trace --memory-copy-trace` run of the sink (tools/r05_timeline.sh): how busy the copy engines and the
compute queues were, how long a picture's copy took, and what lay between two copies.
    python tools/timeline_summary.py <rocprof output dir> [picture bytes] > gpurun_out/<name>.txt
Reads every *_memory_copy_trace.csv / *_kernel_trace.csv below the directory; the first quarter of the run (start-up) is left out."""
import csv
import glob
import os
import sys


def load(pattern, root):
    rows = []
    for path in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        with open(path, newline="") as fp:
            for r in csv.DictReader(fp):
                rows.append(r)
    return rows


def col(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return r[n]
    return None


def union(iv):
    """total length of the union of [a, b) intervals"""
    iv = sorted(iv)
    busy, end = 0, None
    for a, b in iv:
        if end is None or a > end:
            busy += b - a
            end = b
        elif b > end:
            busy += b - end
            end = b
    return busy


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))] if v else 0


def main():
    root = sys.argv[1]
    pic = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    copies = load("*memory_copy_trace.csv", root)
    kernels = load("*kernel_trace.csv", root)
    if not copies and not kernels:
        print("no traces under", root)
        return 1
    ks = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name") or "?") for r in kernels]
    cs = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Direction") or "?") for r in copies]
    t0 = min([a for a, _, _ in ks] + [a for a, _, _ in cs])
    t1 = max([b for _, b, _ in ks] + [b for _, b, _ in cs])
    w0 = t0 + (t1 - t0) // 4
    span = t1 - w0
    print("trace: %d kernel launches, %d copies over %.1f ms; window = the last three quarters (%.1f ms)" % (len(ks), len(cs), (t1 - t0) / 1e6, span / 1e6))
    kin = [(max(a, w0), b) for a, b, _ in ks if b > w0]
    print("compute: any kernel running %.1f %% of the window (sum of kernel durations %.1f %% -- above 100 %% means kernels of several streams overlap)"
          % (100.0 * union(kin) / span, 100.0 * sum(b - a for a, b in kin) / span))
    by = {}
    for a, b, d in cs:
        if b > w0:
            by.setdefault(d, []).append((max(a, w0), b))
    for d, iv in sorted(by.items()):
        dur = [b - a for a, b in iv]
        print("copies %-28s %6d, engine busy %5.1f %% of the window, duration p10 / p50 / p90 / max = %.0f / %.0f / %.0f / %.0f us"
              % (d + ":", len(iv), 100.0 * union(iv) / span, pct(dur, 0.1) / 1e3, pct(dur, 0.5) / 1e3, pct(dur, 0.9) / 1e3, max(dur) / 1e3))
    # the pictures' way out: the large device-to-host copies
    d2h = sorted((a, b) for a, b, d in cs if b > w0 and "DEVICE_TO_HOST" in d.upper().replace(" ", "_"))
    if d2h:
        med = pct([b - a for a, b in d2h], 0.5)
        big = [(a, b) for a, b in d2h if b - a > med / 4]   # (small read-backs aside)
        dur = [b - a for a, b in big]
        gaps = [big[i + 1][0] - big[i][1] for i in range(len(big) - 1)]
        print("picture copies (device to host, longer than a quarter of the median): %d in %.1f ms = %.0f pictures/s" % (len(big), span / 1e6, len(big) / (span / 1e9)))
        print("  duration p10 / p50 / p90 = %.0f / %.0f / %.0f us" % (pct(dur, 0.1) / 1e3, pct(dur, 0.5) / 1e3, pct(dur, 0.9) / 1e3), end="")
        if pic:
            print("  -> %.1f / %.1f / %.1f GB/s for %.2f MB" % (pic / pct(dur, 0.1), pic / pct(dur, 0.5), pic / pct(dur, 0.9), pic / 1e6))
        else:
            print()
        print("  busy %.1f %% of the window; gap to the next copy p10 / p50 / p90 = %.0f / %.0f / %.0f us; gaps above 200 us: %d (%.1f ms in all = %.1f %% of the window)"
              % (100.0 * union(big) / span, pct(gaps, 0.1) / 1e3, pct(gaps, 0.5) / 1e3, pct(gaps, 0.9) / 1e3, sum(1 for g in gaps if g > 200000),
                 sum(g for g in gaps if g > 200000) / 1e6, 100.0 * sum(g for g in gaps if g > 200000) / span))
        # does a copy run slower while kernels run? split copies by how much of their duration overlapped any kernel
        kiv = sorted(kin)
        starts = [a for a, _ in kiv]
        import bisect
        merged = []
        for a, b in kiv:
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        ms = [m[0] for m in merged]
        lo, hi = [], []
        for a, b in big:
            i = max(0, bisect.bisect_right(ms, a) - 1)
            ov = 0
            while i < len(merged) and merged[i][0] < b:
                ov += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
                i += 1
            (hi if ov > 0.5 * (b - a) else lo).append(b - a)
        if pic:
            for name, v in (("mostly beside kernels", hi), ("mostly alone", lo)):
                if v:
                    print("  %-22s %5d copies, median %.0f us = %.1f GB/s" % (name + ":", len(v), pct(v, 0.5) / 1e3, pic / pct(v, 0.5)))
    # the ten kernels that hold the most time
    tot = {}
    for a, b, n in ks:
        if b > w0:
            tot[n] = tot.get(n, 0) + b - max(a, w0)
    print("kernels by time in the window:")
    for n, t in sorted(tot.items(), key=lambda x: -x[1])[:12]:
        print("  %5.1f %%  %s" % (100.0 * t / span, n[:110]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
