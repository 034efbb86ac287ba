#!/bin/bash
# Counter pass over the GEMM micro-benchmarks: usage tools/pmc_kbench.sh TAG COUNTERS FILTER
TAG=$1; CTR=$2; FILT=$3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pk_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc -o p -- python tools/kbench.py "$FILT" > $OUT/out.txt 2> $OUT/err.txt || { echo "pmc pass failed"; tail -5 $OUT/err.txt; exit 1; }
python tools/pmc_summary.py $OUT/pmc
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
