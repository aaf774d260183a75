"""The kernels of the LAST training step of a rocprofv3 --kernel-trace CSV, in
launch order, runs of one kernel collapsed.
    python tools/dbg/step_sequence.py <dir> <steps in trace incl. warm-up>"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     k.split('(')[0].replace('void ', '')[:58],
                     r['Grid_Size_X']))
rows.sort()
# a step starts at each adam_kernel pair's end: take the kernels after the
# third-last adam launch up to the last one (gen step + disc step)
adam = [i for i, r in enumerate(rows) if r[2].startswith('adam')]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = adam[-per - 1] + 1, adam[-1] + 1
seq = rows[lo:hi]
print(f'{len(seq)} kernels, span {(seq[-1][1] - seq[0][0]) / 1e6:.3f} ms, '
      f'busy {sum(e - s for s, e, _, _ in seq) / 1e6:.3f} ms')
i = 0
t = 0.0
while i < len(seq):
    j = i
    while j + 1 < len(seq) and seq[j + 1][2] == seq[i][2] and \
            seq[j + 1][3] == seq[i][3]:
        j += 1
    d = [e - s for s, e, _, _ in seq[i:j + 1]]
    t += sum(d) / 1e3
    print(f'{t:9.1f} us  {j - i + 1:3d} x {sum(d) / len(d) / 1e3:8.1f} us  '
          f'{seq[i][2]} [{seq[i][3]}]')
    i = j + 1
