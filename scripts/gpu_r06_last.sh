#!/bin/bash
# round 6, last: the whole -m gpu suite, smoke() and the driver's bench command on the final tree
cd $GRAFT_REPO_ROOT
M=gpurun_out/r6l; mkdir -p $M
(time python -m pytest tests -m gpu -x -q --durations=10) > $M/r06_pytest_gpu_last.log 2>&1; tail -n 5 $M/r06_pytest_gpu_last.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $M/r06_bench_driver_command_last.json 2> $M/bench.err; echo "bench rc=$?"
python - <<PY
import json
b=json.loads(open("$M/r06_bench_driver_command_last.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["build"]["last_iteration"]["update"]["frac"], b["host_api"]["value"] if isinstance(b.get("host_api"), dict) else b.get("host_api"))
PY
