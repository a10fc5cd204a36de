#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics, HBM traffic and issue counters of the
# display-list kernel replaying the golden LVGL scene (1649 descriptors) over 4096 frames.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof/list
mkdir -p $OUT
ARGS="tools/bench_list.py --only-list --scene tests/golden/scene2_f0_swap1.npz --steps 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ARGS > $OUT/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
echo "list done"
