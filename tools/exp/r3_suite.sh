#!/bin/bash
# The whole GPU suite (run on the GPU box), output kept under gpurun_out/r3_suite.
set -o pipefail
O=gpurun_out/r3_suite
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1
rc=$?
tail -40 $O/tests.log
exit $rc
