"""Digest one scripts/gpu_profile.sh output directory into the small files kept under profiles/:
    python scripts/make_pmc_json.py <gpurun_out/tag/prof_dir> <kernel substring> <profiles/out prefix> "<command>"
writes <prefix>_kernel_stats.csv (rocprofv3 --stats of the trace pass), <prefix>_rocprof_summary.txt and <prefix>_pmc.json
(HBM traffic of the dominant kernel per launch: FETCH_SIZE (KB, as reported) x 1024 x 2 -- gfx950 tallies the 128-B requests of
wide streaming reads at 64 B, MI355X_MICROARCH.md "HBM" -- next to the algorithmic bytes bench.py printed in the same run)."""
import csv
import glob
import json
import os
import shutil
import sys

src, kernel, prefix, cmd = sys.argv[1:5]
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[0], prefix + "_kernel_stats.csv")
shutil.copy(os.path.join(src, "summary.txt"), prefix + "_rocprof_summary.txt")
summ = json.load(open(os.path.join(src, "summary.json")))
key = next(k for k in summ if kernel in k and "FETCH_SIZE" in summ[k])
bench = json.loads(open(os.path.join(src, "trace_stdout.log")).read().strip().splitlines()[-1])
row = next(r for r in csv.DictReader(open(stats[0])) if kernel in r["Name"])
fetch_kb = summ[key]["FETCH_SIZE"]
out = {
    "kernel": key.replace("void ", ""),
    "command": cmd,
    "nvec": bench["config"]["nvec"], "dim": bench["config"]["dim"], "k": bench["config"]["k"], "nprobe": bench["config"]["nprobe"],
    "workload": bench["config"]["workload"],
    "FETCH_SIZE_KB_reported_avg": round(fetch_kb, 1),
    "correction": "gfx950 FETCH_SIZE counts 128-B requests of wide coalesced streaming reads at 64 B: x2 (MI355X_MICROARCH.md, HBM)",
    "traffic_bytes_per_launch": int(fetch_kb * 1024 * 2),
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "sq": {k: round(v, 1) for k, v in summ[key].items() if k != "FETCH_SIZE" and k != "WRITE_SIZE"},
    "trace_pass": {
        "kernel_avg_us_rocprof": round(float(row["AverageNs"]) / 1e3, 2), "launches": int(row["Calls"]),
        "kernel_avg_ms_bench_hip_events_same_run": bench["roofline"]["kernel_ms_avg"],
        "note": "rocprof average over every launch of the process (sweep, settle, warmup, timed, phase pass); bench.py's "
                "HIP-event mean over the timed steps of the same profiled run",
    },
}
out["traffic_over_algorithmic"] = round(out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"], 3)
json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
