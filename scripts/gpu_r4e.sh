#!/bin/bash
# row-per-lane scan: share and granularity of the dynamically claimed tail (probe build)
O=gpurun_out/r4e; mkdir -p $O
run() { name=$1; shift; args=$1; shift
  env "$@" timeout 600 python bench.py $args --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}.json 2> $O/b_${name}.err
}
for cfg in "25 64" "40 64" "25 32" "40 32" "50 32" "40 16"; do
  set -- $cfg
  run np8_p$1_c$2 "--nprobe 8" QK_SCAN_RL_DYN_PCT=$1 QK_SCAN_RL_DYN_CHUNK=$2
  run np16_p$1_c$2 "--nprobe 16" QK_SCAN_RL_DYN_PCT=$1 QK_SCAN_RL_DYN_CHUNK=$2
  run hard_p$1_c$2 "--manifold 10" QK_SCAN_RL_DYN_PCT=$1 QK_SCAN_RL_DYN_CHUNK=$2
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e)
PY
