#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box via gpurun).
# usage: tools/prof.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --no-cpu-baseline --no-placement-ab"   # the default run (20 steps, 3 warmup) minus the CPU leg and minus the
# placement-off repetition (a second context without the arena runs the same kernels slower: its launches would be averaged into the
# kernel-trace figures of the product path)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > /dev/null 2> $OUT/pmc_sq2.err
find $OUT -name "*.csv" | head -50
