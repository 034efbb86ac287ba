#!/usr/bin/env python3
"""Kernels of the last traced micro-step in launch order, showing the ones that are not this library's (torch fills,
copies, foreach ops) and the library kernel each follows.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o s -- python bench.py --steps 2 --warmup 1 --eager
       python tools/kernel_sequence.py DIR"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 400 kernels ~ the last eager micro-step
names = [r['Kernel_Name'] for r in rows]
# find the last s1_mask_kernel occurrence = start of last micro-step
idx = max(i for i, n in enumerate(names) if 's1_mask' in n)
prev = None
for r in rows[idx - 6: idx + 330]:
    n = r['Kernel_Name']
    short = n.split('(')[0][-70:]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    lib = ('scm' in n) or any(k in n for k in ('favor', 'gemm', 'wgrad', 'reduce_batch', 's1_', 'ce_', 'head_', 'embed_', 'cast_'))
    if not lib:
        print(f"   after {prev[-40:] if prev else None!s:42s} -> {d:6.1f} us  {short}")
    else:
        prev = short
