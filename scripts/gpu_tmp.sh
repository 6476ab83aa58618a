export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r7c
run() { tag=$1; shift; env "$@" python scripts/nprobe_sweep.py --corpus $CORP --nprobes 8,16,32 --steps 30 --tag $tag > gpurun_out/r7c/${CORP}_$tag.jsonl 2> gpurun_out/r7c/${CORP}_$tag.err; }
for CORP in mixture hard; do
run base
run r1w1 QK_SEED_RANKS=1
run r1w2 QK_SEED_RANKS=1 QK_SEED_WAVES=2
run r1w4 QK_SEED_RANKS=1 QK_SEED_WAVES=4
run r3w1 QK_SEED_RANKS=3
done
cat gpurun_out/r7c/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['corpus'], r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'step', r['step_ms'], 'group', r['phases_ms']['group'])
"
