"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>/) into the small files kept under profiles/:
   python3 tools/profile_summary.py <tag> <name>   ->  profiles/<name>_kernel_stats_{pipelined,serial}.csv (rocprofv3 --stats as is),
   profiles/<name>_kernels.json (durations of the launches of the bench workload only: the stats files average over every
   launch, the small batches of the end-to-end legs included), profiles/<name>_pmc.json (counter means per kernel and launch,
   HBM-side traffic corrected as profiles/<name>_fetch_calibration.json shows)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
base = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
KEYS = ("amp::k_fast", "k_gcompact", "amp::k_tile", "k_deferred", "k_call", "k_event", "k_reads_lane", "amp::k_trim", "amp::k_scan")
short = lambda n: n.split("(")[0].replace("void ", "")

kern = {}
for mode in ("pipelined", "serial"):
    st = glob.glob(os.path.join(base, "kt_" + mode, "*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join("profiles", "%s_kernel_stats_%s.csv" % (name, mode)))
    tr = glob.glob(os.path.join(base, "kt_" + mode, "*kernel_trace.csv"))
    if not tr:
        continue
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    lead = [r for r in rows if "k_fast" in r["Kernel_Name"] or "amp::k_tile" in r["Kernel_Name"]]
    gmax = max(int(r["Grid_Size_X"]) for r in lead if "k_fast" in r["Kernel_Name"]) if any("k_fast" in r["Kernel_Name"] for r in lead) else None
    acc, take = collections.defaultdict(list), gmax is None
    for r in rows:
        n = r["Kernel_Name"]
        if "k_fast" in n and gmax is not None:
            take = int(r["Grid_Size_X"]) == gmax          # a launch of the bench workload starts here
        if take and any(k in n for k in KEYS):
            acc[short(n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    kern[mode] = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "max_us": round(max(v), 2)}
                  for k, v in acc.items()}
kern["_note"] = ("kernel durations from rocprofv3 --kernel-trace of `bench.py` (pipelined: the default run, steps dealt over several HIP streams, so "
                 "kernels of consecutive steps overlap and their own durations stretch; serial: --no-pipeline, one step at a time), "
                 "launches of the bench workload only (k_fast with its full grid and the kernels behind it)")
json.dump(kern, open(os.path.join("profiles", name + "_kernels.json"), "w"), indent=1)

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(base, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in KEYS):
            pmc[short(n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in pmc.items():
    # keep the launches of the bench workload: the largest values of a counter belong to the full-size batches
    o = {}
    for c, v in d.items():
        v = sorted(v)
        top = [x for x in v if x >= 0.5 * v[-1]] if v[-1] > 0 else v
        o[c] = round(sum(top) / len(top), 1)
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes_per_launch"] = int(o["FETCH_SIZE"] * 1024 * 2 + o["WRITE_SIZE"] * 1024)
    out[k] = o
out["_note"] = ("means per launch over the full-size launches of `bench.py --no-pipeline` (separate --pmc passes); SQ_* cycle counters "
                "count in units of 4 cycles; hbm_bytes_per_launch = FETCH_SIZE [KiB] * 1024 * 2 + WRITE_SIZE [KiB] * 1024 (gfx950 reports "
                "half of the fetched bytes: *_fetch_calibration.json).  FETCH_SIZE / WRITE_SIZE come from the L2's memory-side request counters; by "
                "/opt/skills/guides/MI355X_MICROARCH.md requests that hit the 256 MB Infinity Cache appear to be counted too, so the figure "
                "is an upper bound of the bytes HBM itself moved")
json.dump(out, open(os.path.join("profiles", name + "_pmc.json"), "w"), indent=1)

cal = {}
for f in glob.glob(os.path.join(base, "calib_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        cal.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if cal:
    json.dump({"note": "tools/micro/fetch_calib.hip: every kernel streams exactly 1 GiB (1,048,576 KiB) once", "kernels":
               {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in cal.items() if k.startswith("k_")}},
              open(os.path.join("profiles", name + "_fetch_calibration.json"), "w"), indent=1)
print(json.dumps(kern, indent=1)[:1500])
print({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)})
