#!/bin/bash
# kernel trace of one bench mode: tools/prof_mode.sh MODE [bench args]  -> gpurun_out/prof_MODE/kernel_stats_summary.txt
MODE=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$MODE
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --mode $MODE --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
python tools/prof_summary.py $OUT/trace 1 70 > $OUT/kernel_stats_summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
head -60 $OUT/kernel_stats_summary.txt
