"""The kernel SEQUENCE of the last complete period of a rocprofv3 --kernel-trace CSV
(a training step repeated N times): python tools/dbg/step_sequence.py <dir> <n_steps_traced> [min_us]
prints every dispatch of the last step in order (consecutive repeats merged) and the
per-kernel totals of that step."""
import collections
import csv
import glob
import re
import sys

d, nsteps = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:70]


ker = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
ker.sort()
# the marker of a step: the LAST adam_kernel dispatch pair; steps are separated by
# runs of adam kernels (one per network)
adam = [i for i, k in enumerate(ker) if k[2].startswith('adam')]
# group consecutive adam indices
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 1]
# a training step has two optimizer applications (gen, disc)
per = 2
last_end = ends[-1]
first = ends[-1 - per] + 1
step = ker[first:last_end + 1]
span = step[-1][1] - step[0][0]
busy = sum(e - s for s, e, _ in step)
print(f'last step: {len(step)} dispatches, span {span / 1e6:.3f} ms, sum of kernels {busy / 1e6:.3f} ms')
prev, cnt, tot, t0 = None, 0, 0, 0
for s, e, n in step + [(0, 0, None)]:
    if n != prev:
        if prev is not None and tot / 1e3 >= min_us:
            print(f'{(t0 - step[0][0]) / 1e6:8.3f} ms  {cnt:3d} x {tot / cnt / 1e3:8.1f} us = {tot / 1e6:7.3f} ms  {prev}')
        prev, cnt, tot, t0 = n, 0, 0, s
    cnt += 1
    tot += e - s
agg = collections.Counter()
num = collections.Counter()
for s, e, n in step:
    agg[n] += e - s
    num[n] += 1
print('--- totals of the step')
for n, t in agg.most_common(60):
    print(f'{t / 1e6:8.3f} ms {num[n]:4d} calls avg {t / num[n] / 1e3:8.1f} us  {n}')
