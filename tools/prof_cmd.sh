#!/bin/bash
# rocprofv3 passes over ANY python tool of this repo (run on the GPU box via gpurun), condensed into one table per kernel:
# kernel-trace + stats, FETCH_SIZE, WRITE_SIZE (separate passes: the TCC block cannot hold both), one SQ pass.
# The program sits directly after `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
# usage: tools/prof_cmd.sh <tag> <name> <script.py> [args...]
#   -> gpurun_out/prof_<tag>_<name>/...  and  gpurun_out/<tag>_<name>.txt  (copy the .txt into profiles/)
set -u
TAG=$1; NAME=$2; shift 2
OUT=$PWD/gpurun_out/prof_${TAG}_${NAME}
mkdir -p $OUT
export TMPDIR=/tmp
SCRIPT=$PWD/$1; shift
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT "$@" > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $SCRIPT "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $SCRIPT "$@" > /dev/null 2> $OUT/pmc_write.err
if [ "${KNOX_PROF_SQ:-1}" = "1" ]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $SCRIPT "$@" > /dev/null 2> $OUT/pmc_sq.err
fi
cd $ROOT
python3 tools/prof_kernels.py $OUT "$(basename $SCRIPT) $*" > gpurun_out/${TAG}_${NAME}.txt
cat $OUT/trace.out >> gpurun_out/${TAG}_${NAME}.txt
tail -n 60 gpurun_out/${TAG}_${NAME}.txt
