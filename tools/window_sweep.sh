#!/bin/bash
# S1 throughput against micro-steps per launch chain (bench.py --window), grad_acc = max(5, window)
mkdir -p gpurun_out
for w in ${WINDOWS:-1 5 10}; do
  ga=$(( w > 5 ? w : 5 ))
  timeout -k 10 300 python bench.py --window $w --grad-acc $ga --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${EXTRA} > gpurun_out/window_$w.json 2> gpurun_out/window_$w.err || { tail -5 gpurun_out/window_$w.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/window_$w.json"))
print("window", $w, "grad_acc", $ga, "cells/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "dominant", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), flush=True)
PY
done
