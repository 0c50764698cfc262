"""VALU-issue roofline per kernel, next to the HBM one: from a profile set of tools/collect_profiles.sh
  python tools/valu_roofline.py profiles/r03_k_4k_default [--out profiles/valu_roofline.json]
For every kernel of the default workload: vector instructions per launch (SQ_INSTS_VALU, wave-instructions, mean per dispatch of
the --steps 1 counter pass = the same launches as the timed ones), the time the chip needs just to ISSUE them
  t_issue = instructions x c / (1024 SIMDs x 2.4 GHz),   c = 2.6 cycles (full-rate op, measured: profiles/r02_a_valu_issue_rates.txt)
                                                          .. 4.4 cycles (packed / 3-operand integer ops: half rate on gfx950)
the measured launch duration (kernel_stats.csv) and VALUBusy.  frac = t_issue / duration is how close a kernel is to the vector
issue roof (the bound of every kernel here whose HBM fraction is low); lane-instructions per sample where the sample domain is plain."""
import argparse
import csv
import json
import re


def read_counters(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = re.sub(r"^void ", "", line.split(" dispatches")[0]).replace("ea1::", "").replace(" ", "")
            out[name] = {}
        else:
            parts = line.split()   # "counter  mean  [(min .., max .. over n dispatches)]"
            out[name][parts[0]] = float(parts[1])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix")
    ap.add_argument("--out", default="profiles/valu_roofline.json")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--chains", type=int, default=32)
    a = ap.parse_args()
    sq = read_counters(a.prefix + "_sq_counters.txt")
    util = read_counters(a.prefix + "_valubusy_mfmautil.txt")
    dur = {}
    for row in csv.DictReader(open(a.prefix + "_kernel_stats.csv")):
        name = re.sub(r"\(.*", "", row["Name"]).replace("void ", "").replace("ea1::", "").replace(" ", "")
        dur[name] = float(row["AverageNs"]) * 1e-3
    luma = a.width * a.height * a.chains
    # sample domains per launch (samples the kernel produces): plain cases only
    domain = {"cdef_sorted_kernel<8,128>": luma, "cdef_filter_kernel<4>": luma // 2, "cdef_filter_kernel<4,1,1>": luma // 2, "cdef_dir_kernel": luma, "deblock_luma_v_kernel": luma,
              "deblock_luma_h_kernel": luma, "deblock_chroma_v_kernel": luma // 2, "deblock_chroma_h_kernel": luma // 2,
              "lr_stripe_kernel<64,64,256,1,false,false>": luma, "lr_stripe_kernel<32,32,256,2,true,false>": luma // 2}
    simds, clk = 1024, 2.4e9
    rows = []
    for name, c in sq.items():
        if name not in dur or "SQ_INSTS_VALU" not in c or name.startswith("__amd") or name.startswith("derive") or name in ("pull_kernel", "expand_coefs_kernel"):
            continue
        n = c["SQ_INSTS_VALU"]
        lo, hi = n * 2.6 / (simds * clk) * 1e6, n * 4.4 / (simds * clk) * 1e6
        r = {"kernel": name, "avg_us": round(dur[name], 1), "valu_wave_instructions": int(n), "issue_floor_us": [round(lo, 1), round(hi, 1)],
             "frac_of_duration": [round(lo / dur[name], 3), round(hi / dur[name], 3)], "valu_busy_percent": util.get(name, {}).get("VALUBusy")}
        if name in domain:
            r["lane_instructions_per_sample"] = round(n * 64 / domain[name], 1)
        rows.append(r)
    rows.sort(key=lambda r: -r["avg_us"])
    doc = {"source": a.prefix + "_{sq_counters,valubusy_mfmautil,kernel_stats}", "workload": f"{a.width}x{a.height}, {a.chains} chains per launch",
           "note": __doc__.split("For every kernel")[1].strip(), "kernels": rows}
    json.dump(doc, open(a.out, "w"), indent=1)
    print("| kernel | µs / launch | VALU wave-instr | issue floor µs (2.6 .. 4.4 cyc) | floor / duration | VALUBusy % | lane-instr / sample |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r['kernel']}` | {r['avg_us']} | {r['valu_wave_instructions'] / 1e6:.1f} M | {r['issue_floor_us'][0]} .. {r['issue_floor_us'][1]} | "
              f"{r['frac_of_duration'][0]} .. {r['frac_of_duration'][1]} | {r['valu_busy_percent']} | {r.get('lane_instructions_per_sample', '')} |")


if __name__ == "__main__":
    main()
