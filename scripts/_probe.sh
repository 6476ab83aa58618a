python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['phases_ms'])
"; }
run --dim 768 --metric ip --k 100
run
python scripts/latency_probe.py 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print({k:(v['mean_us'] if isinstance(v,dict) else v) for k,v in j.items()})
"
