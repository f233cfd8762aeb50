#!/bin/bash
# usage: tools/prof.sh <workload> <outdir>   (run on the GPU box through gpurun)
# kernel trace + PMC passes (counters in their own runs, as the pool requires)
W=$1; OUT=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/prof_fir.py $W 5 > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 tools/prof_fir.py $W 3 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -- python3 tools/prof_fir.py $W 3 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 tools/prof_fir.py $W 3 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 tools/prof_fir.py $W 3 > $OUT/pmc4.log 2>&1
python3 tools/prof_summary.py $OUT "$3" | tee $OUT/summary.txt
