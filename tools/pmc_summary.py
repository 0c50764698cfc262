"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "", 1).split("(")[0][:60]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    n = max(len(v) for v in ctrs.values())
    print(name, "dispatches", n)
    for c, v in sorted(ctrs.items()):
        print("   %-28s %14.1f" % (c, sum(v) / len(v)) + ("" if len(v) == 1 or min(v) == max(v) else "   (min %.1f, max %.1f over %d dispatches)" % (min(v), max(v), len(v))))
