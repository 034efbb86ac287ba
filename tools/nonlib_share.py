#!/usr/bin/env python3
"""Share of a traced run's GPU time spent in kernels that are not this library's, and the wall span against the
summed kernel time.  usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o s -- python bench.py ...;
python tools/nonlib_share.py DIR"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[len(rows) // 3:]                      # skip start-up (weight preparation, first-call work)
lib = lambda n: ('scm' in n) or any(k in n for k in ('favor', 'gemm', 'wgrad', 'reduce_batch', 's1_', 'ce_', 'head_', 'embed_', 'cast_', 'ffn_chain', 'bin_'))
tot = sum(e - s for s, e, _ in rows)
non = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if not lib(n):
        k = n.split('(')[0][-60:]
        non[k][0] += 1; non[k][1] += e - s
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels, span {span / 1e6:.2f} ms, kernel time {tot / 1e6:.2f} ms, non-library {sum(v[1] for v in non.values()) / 1e6:.2f} ms "
      f"({100 * sum(v[1] for v in non.values()) / tot:.1f} % of kernel time), idle {100 * (span - tot) / span:.1f} % of the span")
for k, (c, t) in sorted(non.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {t / 1e3:9.1f} us in {c:5d} launches  {k}")
