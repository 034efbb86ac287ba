#!/bin/bash
# extraction throughput (bench.py --mode extract) against cells per batch
mkdir -p gpurun_out
for b in ${BATCHES:-24 48 96 192}; do
  timeout -k 10 300 python bench.py --mode extract --batch $b --cells ${CELLS:-1536} --warmup 2 --no-cpu-baseline > gpurun_out/extract_b$b.json 2> gpurun_out/extract_b$b.err || { tail -5 gpurun_out/extract_b$b.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/extract_b$b.json"))
print("cells/batch", $b, "cells/s", round(d["value"],1), "ms/batch", round(d["ms_per_step"],2), "mfma frac", round(d["roofline"]["whole_step_frac_of_mfma_peak"],4), "dominant", d["roofline"]["kernel"], flush=True)
PY
done
