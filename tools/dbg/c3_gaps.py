"""Idle time between the kernels of the C3 executor (rocprofv3 --kernel-trace
CSV directory): where the GPU waits for the host.
    python tools/dbg/c3_gaps.py <dir>"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     k.split('(')[0].replace('void ', '')[:60]))
rows.sort()
rows = rows[int(len(rows) * 0.4):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f'span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms '
      f'({100 * busy / span:.1f} %), {len(rows)} kernels')
gaps = collections.defaultdict(lambda: [0, 0])
big = []
end = rows[0][1]
prev = rows[0][2]
for s, e, k in rows[1:]:
    g = s - end
    if g > 0:
        key = prev + ' -> ' + k
        gaps[key][0] += g
        gaps[key][1] += 1
        if g > 50000:
            big.append((g, key))
    if e > end:
        end, prev = e, k
for key, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'{t / 1e3:10.1f} us  x{n:4d}  avg {t / n / 1e3:7.1f} us  {key}')
print('gaps > 50 us:', len(big), 'total %.2f ms' % (sum(g for g, _ in big) / 1e6))
