#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for np in 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np$np -o t -- python $GRAFT_REPO_ROOT/bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 100 --settle 20 > $O/np$np.json 2> $O/np$np.err
find /tmp/prof_np$np -name "*kernel_stats.csv" -exec cp {} $O/np${np}_kernel_stats.csv \;
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for np in (8,):
    for r in csv.DictReader(open(f'gpurun_out/r3d/np{np}_kernel_stats.csv')):
        if 100 <= int(r['Calls']) <= 400: print(np, r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
