#!/bin/bash
OUT=$PWD/gpurun_out/pmc_place
mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 $PWD/tools/placement_pmc.py 12 8"
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.txt 2> $OUT/p1.err
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.txt 2> $OUT/p2.err
rocprofv3 --pmc TCC_EA0_RDREQ_GMI_32B TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_IO_32B TCC_TAG_STALL --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.txt 2> $OUT/p3.err
rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.txt 2> $OUT/p4.err
