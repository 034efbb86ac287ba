#!/bin/bash
# same-box comparison of an older tree (_old_tree, built from its commit) with this tree
for i in 1 2 3; do
  (cd _old_tree && timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('previous tree (micro-step of 3 cells) %.1f cells/s %.3f ms per micro-step' % (d['value'], d['ms_per_step']))") || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --window 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('this tree, --window 1                 %.1f cells/s %.3f ms per micro-step' % (d['value'], d['ms_per_step']))" || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('this tree, default (window 5)         %.1f cells/s %.3f ms per micro-step' % (d['value'], d['ms_per_step']/5))" || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --window 20 --grad-acc 20 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('this tree, --window 20 --grad-acc 20  %.1f cells/s %.3f ms per micro-step' % (d['value'], d['ms_per_step']/20))" || exit 1
