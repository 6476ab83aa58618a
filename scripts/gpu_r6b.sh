#!/bin/bash
for v in "" _nq8 _nq12; do
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip$v.so
echo "== lib $v"
python scripts/coarse_probe.py 4096,16384,65536 8,32 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['nlist'], r['nprobe'], r['coarse_us'])"
done
