"""Summarise gpurun_out/<tag>/ from tools/prof_scan.sh: per-kernel average duration and counter totals per dispatch."""
import csv, glob, os, sys, collections
tag = sys.argv[1]
base = tag if os.path.isdir(tag) else os.path.join("gpurun_out", tag)
short = lambda n: n.split("(")[0].replace("void ", "")[:40]
for f in glob.glob(os.path.join(base, "kt", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("amp::", "k_def", "k_call", "k_event")):
            print("%-42s calls %3s avg %9.1f us" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(base, "pmc_*", "*counter_collection.csv")) + glob.glob(os.path.join(base, "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if any(k in n for k in ("amp::", "k_def")):
            acc[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)[len(v) // 2:] if len(v) > 2 else v        # (the first launch of a run is cold)
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
