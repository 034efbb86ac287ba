#!/bin/bash
# cost of the optimizer step (average, clip, Adam, weight refresh): the same launch chains with the step every 5 / every 20 micro-steps
for ga in 5 20 5 20; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --window 5 --grad-acc $ga | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('window 5, grad_acc $ga: %.1f cells/s %.3f ms per launch chain' % (d['value'], d['ms_per_step']))" || exit 1
done
