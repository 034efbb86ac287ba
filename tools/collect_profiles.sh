#!/bin/bash
# copy the summaries of gpurun_out/prof_TAG (written by tools/gpu_profile.sh TAG on the GPU box) into profiles/
TAG=$1
S=gpurun_out/prof_$TAG
cp $S/kernel_stats_summary.txt profiles/${TAG}_kernel_stats_summary.txt
cp $S/trace/bench_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats.csv
cp $S/pmc_summary.txt profiles/${TAG}_pmc_fetch_write_summary.txt
cp $S/pmc_extract_summary.txt profiles/${TAG}_pmc_fetch_write_extract_summary.txt
cp $S/pmc_mfma_valu_summary.txt profiles/${TAG}_pmc_mfma_valu_summary.txt
cp $S/bench_trace.json profiles/${TAG}_bench_under_rocprof.json
ls -la profiles/${TAG}_*
