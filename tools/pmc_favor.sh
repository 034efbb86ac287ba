#!/bin/bash
# Matrix-core / vector-ALU / LDS counters of one kbench group alone (tools/kbench.py FILTER, default favor).
# usage: tools/pmc_favor.sh TAG [FILTER]
TAG=$1; FILT=${2:-favor}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pf_$TAG
rm -rf $OUT && mkdir -p $OUT
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1 -o p -- python tools/kbench.py $FILT > $OUT/out1.txt 2> $OUT/err1.txt || { echo "sq1 failed"; tail -5 $OUT/err1.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/sq2 -o p -- python tools/kbench.py $FILT > $OUT/out2.txt 2> $OUT/err2.txt || { echo "sq2 failed"; tail -5 $OUT/err2.txt; exit 1; }
python tools/pmc_sq_summary.py $OUT/sq1 $OUT/sq2 > $OUT/summary.txt
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
head -9 $OUT/summary.txt | cut -c1-190
