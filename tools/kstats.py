#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 kernel trace (second half of the run, to skip start-up).
usage: python tools/kstats.py DIR [top]"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[len(rows) // 2:]
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split('(')[0][-70:]
    agg[k][0] += 1; agg[k][1] += e - s
tot = sum(v[1] for v in agg.values())
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{100 * t / tot:5.1f} %  {t / c / 1e3:8.1f} us x {c:5d}  {k}")
