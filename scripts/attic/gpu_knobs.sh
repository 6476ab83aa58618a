#!/bin/bash
# one box visit: the nprobe sweep of the probe build under several settings of its environment switches
# usage: bash scripts/gpu_knobs.sh "<nprobes>[:corpus]" "NAME1 K=V K=V" "NAME2 K=V" ...
export QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_probe.so
NP=${1%%:*}; CORPUS=${1#*:}; [ "$CORPUS" = "$1" ] && CORPUS=mixture; shift
for cfg in "$@"; do
  name=${cfg%% *}; envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs python scripts/nprobe_sweep.py --nprobes $NP --corpus $CORPUS --steps 30 --tag $name 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['tag'], j['corpus'], j['nprobe'], j['kernel'], 'scan_ms', j['scan_ms'], 'step', j['step_ms'])"
done
