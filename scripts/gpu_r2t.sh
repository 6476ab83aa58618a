#!/bin/bash
# round-2 evidence run on the product build: GPU suite, default bench, configs[2] bench, latency probe, dynamic workload, 2-rank gloo run
O=gpurun_out/r2t; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
(time python bench.py) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python bench.py --dim 768 --metric ip --k 100 --no-extra > $O/bench_c3.json 2> $O/bench_c3.err
python scripts/latency_probe.py > $O/latency.json 2> $O/latency.err; tail -c 700 $O/latency.json
python scripts/dynamic_workload.py 2000000 128 60 > $O/dynamic.json 2> $O/dynamic.err; tail -c 900 $O/dynamic.json
QUAKE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --nvec-sharded 6250000 --nlist-sharded 4096 --batch-sharded 512 --steps 50 --warmup 5 --settle 20 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
python - <<'PY'
import json
for f in ('bench','bench_c3','bench_2rank_gloo'):
    try:
        r=json.loads(open(f'gpurun_out/r2t/{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r['config']['nprobe'], r['config']['recall_at_k'], r['roofline']['frac'], r['phases_ms'])
        cb=r.get('cpu_baseline')
        if cb: print('   cpu', cb['value'], cb['cores'], cb.get('effective_cores_measured'), cb['single_thread_qps'], cb['threads_speedup'])
        for k,v in (r.get('workloads') or {}).items():
            print('  ', k, v.get('value'), v.get('ms_per_step'), v.get('config',{}).get('nprobe'), v.get('config',{}).get('recall_at_k'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'), v.get('latency_us_synchronised'), (v.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f,'ERR',e)
PY
