#!/bin/bash
# round-2 milestone run: the whole GPU suite, the default bench, configs[2], rocprofv3 passes for the headline and the hard workload
O=gpurun_out/r2i; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
(time python bench.py) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python bench.py --dim 768 --metric ip --k 100 --no-extra > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err
bash scripts/gpu_profile.sh r2i/prof_headline --no-extra --steps 100 > $O/prof_headline.log 2>&1
bash scripts/gpu_profile.sh r2i/prof_hard --no-extra --steps 100 --manifold 10 > $O/prof_hard.log 2>&1
tail -30 $O/prof_headline.log
python - <<'PY'
import json
for f in ('bench','bench_c3'):
    try:
        r=json.load(open(f'gpurun_out/r2i/{f}.json'))
        print(f, r['value'], r['ms_per_step'], r['config']['nprobe'], r['config']['recall_at_k'], r['roofline']['frac'], r.get('cpu_baseline'))
        for k,v in (r.get('workloads') or {}).items():
            print('  ', k, v.get('value'), v.get('ms_per_step'), v.get('config',{}).get('nprobe'), v.get('config',{}).get('recall_at_k'), (v.get('roofline') or {}).get('frac'), v.get('cpu_baseline'), v.get('latency_us_synchronised'))
    except Exception as e: print(f,'ERR',e)
PY
