#!/bin/bash
# round 6: the whole -m gpu suite (a) with form feedback forced off for every context (QK_FORM_FEEDBACK=0), (b) under rocprofv3
# --kernel-trace, which perturbs every timing -- no assertion of the suite may notice either.  Then the dynamic replays.
cd $GRAFT_REPO_ROOT
M=gpurun_out/r6v
mkdir -p $M
(time QK_FORM_FEEDBACK=0 python -m pytest tests -m gpu -x -q) > $M/r06_pytest_gpu_feedback_off.log 2>&1; tail -3 $M/r06_pytest_gpu_feedback_off.log
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && time timeout 2000 rocprofv3 --kernel-trace --output-format csv -d /tmp/suite_trace -- python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --rootdir $R) > $M/r06_pytest_gpu_under_rocprofv3.log 2>&1; tail -4 $M/r06_pytest_gpu_under_rocprofv3.log
du -sh /tmp/suite_trace 2>/dev/null | tail -1 >> $M/r06_pytest_gpu_under_rocprofv3.log; rm -rf /tmp/suite_trace
scripts/gpu_r06_maint.sh 10000000 m_10M DW_PROFILE=1
scripts/gpu_r06_maint.sh 50000000 n_50M DW_PROFILE=1 DW_OPS=120
