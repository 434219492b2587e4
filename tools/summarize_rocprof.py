"""Condenses rocprofv3 output directories into the small summaries kept under profiles/.

  python tools/summarize_rocprof.py stats <dir> <out.csv>      # --kernel-trace --stats: per-kernel totals / averages
  python tools/summarize_rocprof.py pmc <dir> <out.json>       # --pmc: per-kernel mean counter values per launch
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:160]


def stats(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        rows = list(csv.DictReader(open(files[0])))
        with open(out, "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
                            r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")])
        return
    # fall back to aggregating the kernel trace
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values()) or 1
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)])


def pmc(d, out):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(list))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            agg[short(r.get("Kernel_Name", ""))][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
    res = {}
    for k, cs in agg.items():
        res[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
