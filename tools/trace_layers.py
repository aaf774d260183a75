#!/usr/bin/env python
"""Per-(kernel, grid) breakdown of a rocprofv3 kernel trace csv:
python tools/trace_layers.py <dir with *_kernel_trace.csv> [n_steps] [name filter regex]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    k = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    if flt and not flt.search(k):
        continue
    key = (k, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print('%-44s grid %-20s calls %4d avg %9.1f us  per step %8.2f ms' % (
        k[0][:44], ','.join(k[1:]), len(v), sum(v) / len(v), sum(v) / steps / 1e3))
print('total per step %.2f ms' % (tot / steps / 1e3))
