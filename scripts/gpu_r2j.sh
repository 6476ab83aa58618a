#!/bin/bash
O=gpurun_out/r2j; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --nprobe 1 --no-extra --no-cpu --steps 100 > $O/b_${name}.json 2> $O/b_${name}.err
}
run old QK_SCAN_RL=0
run d40c64 QK_SCAN_RL_DYN_PCT=40 QK_SCAN_RL_DYN_CHUNK=64
run d25c64 QK_SCAN_RL_DYN_PCT=25 QK_SCAN_RL_DYN_CHUNK=64
run d25c128 QK_SCAN_RL_DYN_PCT=25 QK_SCAN_RL_DYN_CHUNK=128
run d40c128 QK_SCAN_RL_DYN_PCT=40 QK_SCAN_RL_DYN_CHUNK=128
run d40c256 QK_SCAN_RL_DYN_PCT=40 QK_SCAN_RL_DYN_CHUNK=256
run d15c64 QK_SCAN_RL_DYN_PCT=15 QK_SCAN_RL_DYN_CHUNK=64
run d0 QK_SCAN_RL_DYN_PCT=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2j/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms'])
    except Exception as e: print(f,'ERR',e)
PY
