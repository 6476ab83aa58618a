#!/bin/bash
# last pass of round 2 on the final product build: GPU suite, default bench, nprobe sweep, coarse / phase probes
O=gpurun_out/r3z; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err
for np in 2 4 8 16 32; do timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_np${np}.json 2> $O/b_np${np}.err; done
python scripts/coarse_probe.py 65536,32768,16384,8192,4096 > $O/coarse.jsonl 2> $O/coarse.err
python scripts/phase_probe.py > $O/phase.jsonl 2> $O/phase.err
python - <<'PY'
import json
for f in ['bench']+['b_np%d'%n for n in (2,4,8,16,32)]:
    r=json.loads(open(f'gpurun_out/r3z/{f}.json').read().strip().splitlines()[-1])
    print(f, r['value'], r['ms_per_step'], r['config']['nprobe'], r['roofline']['kernel'], r['roofline']['bound'], r['roofline']['frac'], r['phases_ms']['coarse'], r['phases_ms']['group'], r['phases_ms']['merge'])
    for k,v in (r.get('workloads') or {}).items(): print('  ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'))
PY
