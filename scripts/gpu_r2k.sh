#!/bin/bash
O=gpurun_out/r2k; mkdir -p $O
for rl in 0 1; do
QK_SCAN_RL=$rl QK_SCAN_WAVE_CLOCK=1 python bench.py --nprobe 1 --no-extra --no-cpu --steps 2 --warmup 1 --settle 1 > $O/clock_rl$rl.json 2> $O/clock_rl$rl.err
grep -E "k_scan launch|k_scan_rl\]|k_scan params" $O/clock_rl$rl.err | tail -3
done
QK_SCAN_RL=0 python -c "
import os,ctypes
print('py env', os.environ.get('QK_SCAN_RL'))
libc=ctypes.CDLL(None); libc.getenv.restype=ctypes.c_char_p; print('c env', libc.getenv(b'QK_SCAN_RL'))"
for rl in 0 1; do QK_SCAN_RL=$rl python bench.py --nprobe 1 --no-extra --no-cpu --steps 100 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('rl=$rl', r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['phases_ms'])"; done
