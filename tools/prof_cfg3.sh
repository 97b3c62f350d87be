#!/bin/bash
OUT=$PWD/gpurun_out/prof_cfg3
mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 $PWD/tools/bench_cfg.py --cfg 3 --steps 6"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $CMD > /dev/null 2> $OUT/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > /dev/null 2>> $OUT/err.txt
