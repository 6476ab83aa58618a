#!/bin/bash
# the bench step on shapes beside the configs (run on the GPU box): one line each -- d, metric, k, nprobe -> queries/s, kernel, fraction
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-extra --no-pmc --steps 20 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']; r = d['roofline']
print(json.dumps({'dim': c['dim'], 'metric': c['metric_type'], 'k': c['k'], 'nprobe': c['nprobe'], 'recall': c.get('recall_at_k'), 'qps': round(d['value']), 'ms_per_step': d['ms_per_step'],
                  'kernel': r.get('kernel'), 'kernel_ms': r.get('kernel_ms_avg'), 'frac': r.get('frac'), 'phases_ms': d.get('phases_ms')}))"; }
run --dim 768 --metric ip --k 100 --nprobe 4
run --dim 768 --metric ip --k 100 --nprobe 16
run --dim 768 --metric l2 --k 10 --nprobe 8
run --dim 256 --metric l2 --k 10 --nprobe 1
run --dim 256 --metric l2 --k 10 --nprobe 8
run --dim 64 --metric l2 --k 10 --nprobe 1
run --dim 64 --metric l2 --k 10 --nprobe 16
run --dim 128 --metric l2 --k 100 --nprobe 8
run --dim 128 --metric ip --k 10 --nprobe 8
run --dim 96 --metric l2 --k 10 --nprobe 8
