#!/bin/bash
# L2 / memory-interface counters of the headline scan with the bitset in ANOTHER placement class (fast) and in the column's OWN
# class (slow): tools/placement_counters.py under separate rocprofv3 --pmc passes.  usage: tools/prof_placement.sh <tag>
# -> gpurun_out/<tag>_placement_counters.txt (copy into profiles/)
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_placement_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/placement_counters.py"
cd /tmp
$CMD > $OUT/plain.out 2> $OUT/plain.err
i=0
for SET in "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TA_TA_BUSY_sum" \
           "TCC_BUSY TCC_CYCLE TCC_TAG_STALL TCC_IB_STALL" \
           "TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL TCC_BUBBLE TCC_REQ" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.out 2> $OUT/pmc$i.err
done
cd - > /dev/null
python3 tools/placement_counters_summary.py $OUT > gpurun_out/${TAG}_placement_counters.txt
cat gpurun_out/${TAG}_placement_counters.txt
