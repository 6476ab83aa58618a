#!/bin/bash
# round 5: the driver's sequence on one box -- the whole -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r05_pytest_gpu.log 2>&1; tail -5 gpurun_out/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err); echo "bench rc=$?"; tail -3 gpurun_out/r05_bench.err
