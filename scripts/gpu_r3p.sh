#!/bin/bash
# round-2 final evidence (second pass, after the flat merge / k_coarse_small / store changes) on the product build: GPU suite, rocprofv3 passes (headline + hard), default bench, configs[2] bench,
# latency probe, coarse probe, dynamic workload, 2-rank functional run of the configs[3] path over gloo
O=gpurun_out/r3p; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
(time python bench.py) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
bash scripts/gpu_profile.sh r3p/prof_headline --no-extra --steps 100 > $O/prof_headline.log 2>&1
bash scripts/gpu_profile.sh r3p/prof_hard --no-extra --steps 100 --manifold 10 > $O/prof_hard.log 2>&1
grep -E "k_scan|k_merge|k_dense|k_seed|k_group|k_prep|fillBuffer" $O/prof_headline/summary.txt | head -14
grep -E "k_scan|k_merge" $O/prof_hard/summary.txt | head -8
python bench.py --dim 768 --metric ip --k 100 --no-extra > $O/bench_c3.json 2> $O/bench_c3.err
for np in 2 4 8 16 32; do timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_np${np}.json 2> $O/b_np${np}.err; done
python scripts/latency_probe.py > $O/latency.json 2> $O/latency.err
python scripts/coarse_probe.py > $O/coarse.jsonl 2> $O/coarse.err
python scripts/phase_probe.py > $O/phase.jsonl 2> $O/phase.err
python scripts/dynamic_workload.py 2000000 128 60 > $O/dynamic.json 2> $O/dynamic.err
python scripts/dynamic_workload.py 10000000 128 60 > $O/dynamic_10M.json 2> $O/dynamic_10M.err
python scripts/add_probe.py > $O/add_probe.jsonl 2> $O/add_probe.err
python scripts/aps_probe.py > $O/aps.jsonl 2> $O/aps.err
QUAKE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --nvec-sharded 6250000 --nlist-sharded 4096 --batch-sharded 512 --steps 50 --warmup 5 --settle 20 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
python - <<'PY'
import json,glob
for f in ['bench','bench_c3','bench_2rank_gloo']+['b_np%d'%n for n in (2,4,8,16,32)]:
    try:
        r=json.loads(open(f'gpurun_out/r3p/{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r['config']['nprobe'], r['config']['recall_at_k'], r['roofline']['kernel'], r['roofline']['frac'], r['phases_ms'])
        cb=r.get('cpu_baseline')
        if cb: print('   cpu', cb['value'], cb['cores'], cb.get('effective_cores_measured'), cb['single_thread_qps'], cb['threads_speedup'])
        for k,v in (r.get('workloads') or {}).items():
            print('  ', k, v.get('value'), v.get('ms_per_step'), v.get('config',{}).get('nprobe'), v.get('config',{}).get('recall_at_k'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'), v.get('latency_us_synchronised'), (v.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f,'ERR',e)
PY
