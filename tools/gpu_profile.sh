#!/bin/bash
# GPU-box profile: rocprofv3 kernel trace + stats of the default bench command, then two counter passes
# (memory-side read and write bytes) of a short eager run.  usage: tools/gpu_profile.sh TAG
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace run failed"; tail -5 $OUT/trace.err; exit 1; }
echo "trace ok"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err || { echo "fetch pass failed"; tail -5 $OUT/fetch.err; exit 1; }
echo "fetch ok"
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err || { echo "write pass failed"; tail -5 $OUT/write.err; exit 1; }
echo "write ok"
python tools/prof_summary.py $OUT/trace 30 60 > $OUT/kernel_stats_summary.txt
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
# keep only the summaries and the stats csv (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
head -30 $OUT/kernel_stats_summary.txt
