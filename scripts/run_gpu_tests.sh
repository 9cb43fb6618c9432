#!/bin/bash
# full GPU suite with the log under gpurun_out/ (so a long run never looks silent)
mkdir -p gpurun_out
T=${1:-t}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/$T.log 2>&1
rc=$?
tail -15 gpurun_out/$T.log
exit $rc
