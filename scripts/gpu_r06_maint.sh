#!/bin/bash
# round 6: the hot dynamic workload under variants of the scenario and of the policy (scripts/dynamic_workload.py, DW_* knobs)
# usage: scripts/gpu_r06_maint.sh <n> <tag> [VAR=value ...]   -> gpurun_out/r06/dyn_<tag>.json (+ .err with the cProfile when DW_PROFILE=1)
n=$1; tag=$2; shift 2
ops=60; for kv in "$@"; do case $kv in DW_OPS=*) ops=${kv#DW_OPS=};; esac; done
mkdir -p gpurun_out/r06
env "$@" timeout 1500 python scripts/dynamic_workload.py $n 128 $ops hot > gpurun_out/r06/dyn_$tag.json 2> gpurun_out/r06/dyn_$tag.err
echo "== $tag rc=$?"
python - <<PY
import json
try:
    j = json.load(open("gpurun_out/r06/dyn_$tag.json"))
    for name, r in j["results"].items():
        print(name, {k: r[k] for k in ("n_list_first_last", "n_splits", "n_deletes", "max_list_size_first_last", "query_batch_ms_p50_second_half",
                                       "query_scan_ms_p50_second_half", "query_call_ms_p50_second_half", "query_device_ms_p50_second_half", "pair_rows_p50_second_half", "unique_rows_p50_second_half",
                                       "maintenance_ms_mean", "maintenance_ms_p50", "maintenance_ms_mean_second_half", "window_size", "maintenance_ms_max", "delete_ms_max", "query_recall_at_10")})
except Exception as e:
    print("no result:", e)
PY
