for ga in 60 5 1 60 5 1; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --grad-acc $ga --steps 120 2>/dev/null | tail -1 > /tmp/l.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/l.json').read()); print('grad_acc', $ga, round(d['value'],1), round(d['ms_per_step'],3))"
done
