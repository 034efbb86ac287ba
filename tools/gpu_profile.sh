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
# matrix-core / vector-ALU / LDS counters: two SQ passes of 8 counters each (+ GRBM_GUI_ACTIVE for the kernel's cycles)
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
timeout -s KILL 400 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/pmc_sq1 -o s1 -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $OUT/bench_sq1.json 2> $OUT/sq1.err || { echo "sq1 pass failed"; tail -5 $OUT/sq1.err; exit 1; }
echo "sq1 ok"
timeout -s KILL 400 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/pmc_sq2 -o s2 -- python bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline > $OUT/bench_sq2.json 2> $OUT/sq2.err || { echo "sq2 pass failed"; tail -5 $OUT/sq2.err; exit 1; }
echo "sq2 ok"
python tools/pmc_sq_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_mfma_valu_summary.txt
# the extraction configuration: read and write bytes of its kernels
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_x -o f -- python bench.py --mode extract --cells 48 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch_x.json 2> $OUT/fetch_x.err || { echo "extract fetch pass failed"; tail -5 $OUT/fetch_x.err; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_x -o w -- python bench.py --mode extract --cells 48 --warmup 1 --no-cpu-baseline > $OUT/bench_write_x.json 2> $OUT/write_x.err || { echo "extract write pass failed"; tail -5 $OUT/write_x.err; exit 1; }
echo "extract passes ok"
python tools/pmc_summary.py $OUT/pmc_fetch_x $OUT/pmc_write_x > $OUT/pmc_extract_summary.txt
python tools/prof_summary.py $OUT/trace 31 60 > $OUT/kernel_stats_summary.txt
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
# keep only the summaries and the stats csv (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
head -30 $OUT/kernel_stats_summary.txt
