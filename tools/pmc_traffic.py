"""Build profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from two separate
rocprofv3 PMC passes of the same bench command (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
fit one pass):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  python tools/pmc_traffic.py F W --workload 1080p --chains 128 --source profiles/<name> [--merge profiles/pmc_traffic.json]

Units and corrections (guide, section HBM, + our own calibration tools/traffic_calib.hip,
profiles/r01_k_traffic_calibration.txt): both counters are KiB per dispatch.  WRITE_SIZE is exact for
full-line stores (and counts 32 B for every partial 8-byte segment).  FETCH_SIZE tallies one 64 B
unit per fabric read request: exact when the L2 asks for 64-byte sectors, HALF the bytes when it
asks for whole 128-byte lines (every wide streaming read).  Traffic per launch is therefore bounded
by  fetch_raw + write  <=  traffic  <=  2 * fetch_raw + write;  `traffic_bytes` is the upper bound
(the right figure whenever neighbouring tiles share an L2, which the XCD-aware tile order arranges)."""
import argparse
import collections
import csv
import glob
import json
import os
import re


def per_kernel(directory, counter):
    path = glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter or "ea1::" not in row["Kernel_Name"]:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("ea1::", "").replace(" ", "")
        acc[name].append(float(row["Counter_Value"]) * 1024.0)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--chains", type=int, required=True)
    ap.add_argument("--source", required=True, help="where the raw summaries are kept (profiles/...)")
    ap.add_argument("--merge", default=None, help="existing pmc_traffic.json to update")
    ap.add_argument("--out", default="profiles/pmc_traffic.json")
    args = ap.parse_args()
    f, w = per_kernel(args.fetch_dir, "FETCH_SIZE"), per_kernel(args.write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(f):
        fr, n = f[name]
        wr = w.get(name, (0.0, 0))[0]
        kernels[name] = {"dispatches": n, "fetch_size_raw_bytes": int(fr), "write_size_bytes": int(wr),
                         "traffic_low_bytes": int(fr + wr), "traffic_bytes": int(2 * fr + wr)}
    entry = {"workload": args.workload, "chains_per_launch": args.chains, "contexts": 1, "source": args.source, "kernels": kernels}
    doc = {"note": __doc__.split("Units and corrections")[1].strip(), "entries": []}
    if args.merge and os.path.exists(args.merge):
        doc = json.load(open(args.merge))
        doc["entries"] = [e for e in doc["entries"] if (e["workload"], e["chains_per_launch"]) != (args.workload, args.chains)]
    doc["entries"].append(entry)
    json.dump(doc, open(args.out, "w"), indent=1)
    print("wrote", args.out, "with", len(kernels), "kernels")


if __name__ == "__main__":
    main()
