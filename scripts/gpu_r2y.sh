#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for np in 8 16; do
QK_SCAN_RL=1 QK_SCAN_RL_TEAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np$np -o team -- python $GRAFT_REPO_ROOT/bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 20 > $O/np$np.json 2> $O/np$np.err
find /tmp/prof_np$np -name "*kernel_stats.csv" -exec cp {} $O/np${np}_kernel_stats.csv \;
tail -3 $O/np$np.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for np in (8,16):
    f=glob.glob(f'gpurun_out/r2y/np{np}_kernel_stats.csv')
    if not f: print('no stats', np); continue
    for r in csv.DictReader(open(f[0])):
        if 'k_scan_rl' in r['Name'] or 'k_merge' in r['Name'] or 'k_group' in r['Name']:
            print(np, r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['MinNs'], r['MaxNs'])
PY
