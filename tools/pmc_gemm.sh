#!/bin/bash
# SQ counters of the GEMM kernels at the north-star micro-batch shape: tools/pmc_gemm.sh <outdir under gpurun_out>
# (run on the GPU box; two rocprofv3 --pmc passes, condensed by tools/pmc_kernels.py)
set -e
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p1 -o p1 --output-format csv -- python $REPO/tools/run_gemms.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o p2 --output-format csv -- python $REPO/tools/run_gemms.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/p3 -o p3 --output-format csv -- python $REPO/tools/run_gemms.py > $OUT/p3.log 2>&1 || true
cd $REPO
for p in p1 p2 p3; do f=$(find $OUT/$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_kernels.py $f gemm; done > $OUT/summary.txt
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/summary.txt
