#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the narrow-type and aggregate-heavy scans (tools/bench_narrow.py): run on the GPU box via gpurun.
# usage: tools/prof_bloom.sh <tag>  -> gpurun_out/prof_narrow_<tag>/ and gpurun_out/<tag>_narrow_pmc.txt (copy into profiles/)
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_narrow_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/bench_narrow.py"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
cd - > /dev/null
python3 tools/prof_summary.py $OUT > gpurun_out/${TAG}_narrow_pmc.txt
cat $OUT/trace.out >> gpurun_out/${TAG}_narrow_pmc.txt
tail -n 30 gpurun_out/${TAG}_narrow_pmc.txt
