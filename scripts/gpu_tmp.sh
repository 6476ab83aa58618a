export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5m
run() { tag=$1; shift; env "$@" QK_SCAN_HOT_UNIT=300 QK_SCAN_RL=1 python scripts/nprobe_sweep.py --corpus $CORP --nprobes $NPS --steps 30 --tag $tag > gpurun_out/r5m/$tag.jsonl 2> gpurun_out/r5m/$tag.err; }
CORP=mixture; NPS=2,4,8,16,32,64
for mn in 5 7 9 13; do run mix_min$mn QK_SCAN_HOT_MIN=$mn; done
run mix_nohot QK_SCAN_HOT_MIN=0
CORP=hard; NPS=4,8,16,32,64
for mn in 7 13; do run hard_min$mn QK_SCAN_HOT_MIN=$mn; done
cat gpurun_out/r5m/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], r['kernel'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'roof', r['frac_of_binding_roof'])
"
