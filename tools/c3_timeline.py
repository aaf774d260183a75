"""Timeline account of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV
directory of the C3 executor: how much of the device -> host delivery runs
UNDER the forward kernels.  python tools/c3_timeline.py <dir> [label]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:48]


rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     short(r['Kernel_Name']), 'k'))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     r.get('Direction', 'copy'), 'c'))
rows.sort()
if not rows:
    sys.exit('no trace rows under ' + d)


def is_copy(r):
    return 'd2h_stream' in r[2] or 'copyBuffer' in r[2] or \
        'DEVICE_TO_HOST' in r[2]


# the steady state: the last 60 % of the trace
t0 = rows[int(len(rows) * 0.4)][0]
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
comp = [r for r in rows if not is_copy(r) and 'HOST_TO_DEVICE' not in r[2]]
cop = [r for r in rows if is_copy(r)]


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap(a, b):
    """time in union(a) that is also in union(b)"""
    return union(a) + union(b) - union(a + b)


ci = [(s, e) for s, e, _, _ in comp]
di = [(s, e) for s, e, _, _ in cop]
print(f'== {label}: span {span / 1e6:.1f} ms, compute busy '
      f'{union(ci) / 1e6:.1f} ms, delivery busy {union(di) / 1e6:.1f} ms, '
      f'delivery under compute {overlap(ci, di) / 1e6:.1f} ms, idle '
      f'{(span - union(ci + di)) / 1e6:.1f} ms')
# compute kernels that overlap a delivery vs those that do not
by = {}
for s, e, name, _ in comp:
    ov = any(s < de and ds < e for ds, de in di)
    k = (name, ov)
    a = by.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for (name, ov), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f'  {name:50s} {"beside delivery" if ov else "alone          "} '
          f'calls {n:5d} avg {t / n / 1e3:9.1f} us')
for s, e, name, _ in cop[:6]:
    print(f'  delivery {name[:40]:40s} {(e - s) / 1e3:9.1f} us')
