#!/bin/bash
# round 6, late: arena compaction in place (bounce buffer) -- the store's dynamic tests with the new form, the same files with the
# replacing form (QK_COMPACT_FRESH=1), the refine probe (its 1.2 s call was a compaction), then the 50M hot replay
R=$GRAFT_REPO_ROOT; M=$R/gpurun_out/r6c; mkdir -p $M
cd $R
timeout 900 python -m pytest tests/test_store_dynamic_gpu.py tests/test_maintenance_gpu.py tests/test_index_gpu.py tests/test_random_index_streams_gpu.py -m gpu -x -q 2>&1 | tail -n 4 | tee $M/r06_compact_pytest.log
QK_COMPACT_FRESH=1 timeout 900 python -m pytest tests/test_store_dynamic_gpu.py -m gpu -x -q 2>&1 | tail -n 2 | tee $M/r06_compact_pytest_fresh_form.log
timeout 600 python scripts/refine_probe.py 2>/dev/null | tee $M/r06_refine_probe_in_place.jsonl | cut -c1-300
if [ "$1" = "replay" ]; then bash scripts/gpu_r06_maint.sh 50000000 r6c_50M DW_EXT=1 DW_OPS=90 > $M/maint50.log 2>&1; tail -n 5 $M/maint50.log | cut -c1-600; fi
