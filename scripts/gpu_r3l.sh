#!/bin/bash
# k_coarse_small: parity sweep + phase probe of the mid-sized batches
O=gpurun_out/r3l; mkdir -p $O
(QK_RANDOM_SHAPES=500 timeout 1500 python -m pytest tests/test_random_shapes_gpu.py tests/test_scan_gpu.py tests/test_index_gpu.py tests/test_bench_parity_gpu.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
python scripts/phase_probe.py > $O/phase.jsonl 2> $O/phase.err; cat $O/phase.jsonl
