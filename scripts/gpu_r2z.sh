#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pr in 0 1 2 3; do
QK_SCAN_RL=1 QK_SCAN_RL_TEAM=1 QK_SCAN_RL_PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p$pr -o team -- python $GRAFT_REPO_ROOT/bench.py --nprobe 16 --no-extra --no-cpu --inflight 1 --steps 50 --settle 20 > $O/p$pr.json 2> $O/p$pr.err
find /tmp/prof_p$pr -name "*kernel_stats.csv" -exec cp {} $O/p${pr}_kernel_stats.csv \;
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for pr in (0,1,2,3):
    for r in csv.DictReader(open(f'gpurun_out/r2z/p{pr}_kernel_stats.csv')):
        if 'k_scan_rl' in r['Name']: print('probe', pr, r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
