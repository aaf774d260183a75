"""GPU-busy account of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV
directory: busy time (union of kernels), gaps by the kernel that FOLLOWS them,
top kernels.  python tools/dbg/timeline_gaps.py <dir> [skip_fraction]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:56]


ker, cop = [], []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cop.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', 'copy')))
ker.sort()
cop.sort()
t_lo = ker[int(len(ker) * skip)][0]
ker = [k for k in ker if k[0] >= t_lo]
cop = [c for c in cop if c[0] >= t_lo]
span = ker[-1][1] - ker[0][0]
busy, cur_s, cur_e = 0, None, None
gaps = collections.Counter()
gapn = collections.Counter()
big = []
for s, e, n in ker:
    if cur_e is None:
        cur_s, cur_e = s, e
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps[n] += s - cur_e
        gapn[n] += 1
        if s - cur_e > 50000:
            big.append(((s - ker[0][0]) / 1e6, (s - cur_e) / 1e3, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms = {busy / span:.3f}, '
      f'{len(ker)} dispatches, idle {(span - busy) / 1e6:.2f} ms')
print('idle time in front of (top 12):')
for n, t in gaps.most_common(12):
    print(f'  {t / 1e6:8.3f} ms in {gapn[n]:5d} gaps (avg {t / gapn[n] / 1e3:7.1f} us)  {n}')
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n in ker:
    tot[n] += e - s
    cnt[n] += 1
print('gaps > 50 us (at ms, us, before):', [(round(a, 2), round(b), c[:24]) for a, b, c in big][:40])
print('kernel time (top 14):')
for n, t in tot.most_common(14):
    print(f'  {t / 1e6:8.3f} ms {cnt[n]:6d} calls avg {t / cnt[n] / 1e3:8.1f} us  {n}')
ct = collections.Counter()
cn = collections.Counter()
for s, e, n in cop:
    ct[n] += e - s
    cn[n] += 1
for n, t in ct.items():
    print(f'  copies {n}: {t / 1e6:.3f} ms in {cn[n]} ({t / cn[n] / 1e3:.1f} us avg)')
