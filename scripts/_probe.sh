run() { python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'], j['config']['recall_at_k'])
"; }
for r in 0 1 0 1; do echo "refresh=$r"; if [ $r = 1 ]; then QK_SCAN_TAU_REFRESH=1 run; else run; fi; done
