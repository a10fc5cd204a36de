#!/bin/bash
# Run on the GPU box from the repository root: counters of the clipped-area kernel (tools/bench_partial.py).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof/rows
rm -rf $OUT
mkdir -p $OUT
ARGS="tools/bench_partial.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ARGS > $OUT/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
echo "rows done"
