#!/bin/bash
# same-box A/B of the product library against a variant built by tools/build_variant.sh:
#   tools/ab_variant.sh NAME [bench args]  -> value and the per-kernel table rows of the attention passes for both
NAME=$1; shift
for i in 1 2; do
  for lib in "" "$PWD/scmomer_amd/libscmomer_hip_$NAME.so"; do
    SCMOMER_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 "$@" > /tmp/ab.json || exit 1
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').readline())
ks = d.get('kernels', {})
import os
pat = os.environ.get('AB_KERNELS', 'favor').split(',')
names = [k for k in ks if any(p_ in k for p_ in pat)]
print('%-10s %.1f %s  ' % ('variant' if sys.argv[1] else 'product', d['value'], d['unit']),
      ' '.join('%s=%.3fms' % (k, ks[k]['ms_per_step']) for k in sorted(names)))
PY
  done
done
