"""Summarise rocprofv3 output dirs (kernel stats + PMC) into a small text/JSON report."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    return name.split("(")[0][:70]


print("== kernel stats (trace pass) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    for r in rows[:14]:
        print(f"{short(r['Name']):70s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:10.3f} "
              f"avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r.get('Percentage','')}")

summary = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{tag}/**/*counter_collection.csv"):
        agg = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
        print(f"== {ctr} per launch (KB as reported; gfx950 FETCH_SIZE under-reports wide streaming reads 2x) ==")
        for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"{k:70s} launches={n:5d} avg={v/n:14.1f}")
            summary.setdefault(k, {})[ctr] = v / n
for f in find("pmc_sq/**/*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    print("== SQ counters per launch ==")
    for k, cs in agg.items():
        if "k_scan" in k or "k_merge" in k or "k_assign" in k:
            print(k, {c: round(v / n, 1) for c, (v, n) in cs.items()})
            summary.setdefault(k, {}).update({c: v / n for c, (v, n) in cs.items()})
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
