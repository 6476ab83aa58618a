#!/bin/bash
# sharded maintenance on the GPU (world 1 in process, world 2 as two processes over gloo), then the whole GPU suite
O=gpurun_out/r2o; mkdir -p $O
(time timeout 900 python -m pytest tests/test_sharded_maintenance_gpu.py -x -q) > $O/pytest_sm.log 2>&1; tail -30 $O/pytest_sm.log
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
