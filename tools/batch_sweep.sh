#!/bin/bash
# train-mode throughput against cells per launch chain (one key stabiliser over the whole batch)
mkdir -p gpurun_out
for b in ${BATCHES:-3 6 15 30}; do
  timeout -k 10 240 python bench.py --batch $b --grad-acc 1 --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${EXTRA} > gpurun_out/sweep_b$b.json 2> gpurun_out/sweep_b$b.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_b$b.json"))
print("batch", $b, "cells/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), flush=True)
PY
done
