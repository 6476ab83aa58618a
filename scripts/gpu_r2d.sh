#!/bin/bash
# round-2 probe: small-batch kernel tests + latency; row-per-lane scan with cost model v2 / dynamic tail
O=gpurun_out/r2d; mkdir -p $O
(timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_index_gpu.py tests/test_bindings_gpu.py tests/test_aps_gpu.py tests/test_maintenance_gpu.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
(QK_SCAN_RL=1 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py -m gpu -x -q -k "not configs2") > $O/pytest_rl1.log 2>&1; tail -3 $O/pytest_rl1.log
python scripts/latency_probe.py > $O/latency.log 2>&1; tail -12 $O/latency.log
QK_SMALL=0 python scripts/latency_probe.py > $O/latency_nosmall.log 2>&1; tail -6 $O/latency_nosmall.log
run() { # name, env...
  name=$1; shift
  for np in 8 32; do
    env QK_SCAN_RL=1 "$@" timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_${name}_np${np}.json 2> $O/b_${name}_np${np}.err
  done
}
run base
run nodyn QK_SCAN_RL_DYN_PCT=0
run dyn40 QK_SCAN_RL_DYN_PCT=40
run dyn60c32 QK_SCAN_RL_DYN_PCT=60 QK_SCAN_RL_DYN_CHUNK=32
run m5 QK_SCAN_RL_M=5
run h16 QK_SCAN_RL_H0=16
run h8 QK_SCAN_RL_H0=8
QK_SCAN_RL=1 timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --steps 50 --settle 50 > $O/b_base_hard.json 2> $O/b_base_hard.err
QK_SCAN_RL=1 timeout 600 python bench.py --nprobe 1 --no-extra --no-cpu --steps 50 --settle 50 > $O/b_base_np1.json 2> $O/b_base_np1.err
for np in 8 32; do QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 2 --warmup 1 --settle 2 > $O/clock_np${np}.json 2> $O/clock_np${np}.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2d/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e)
PY
for np in 8 32; do grep -E "k_scan waves|decile" $O/clock_np${np}.err | tail -11; done
