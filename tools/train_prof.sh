#!/bin/bash
# per-kernel breakdown of one training configuration under rocprofv3
# Usage on the GPU box: bash tools/train_prof.sh gpurun_out/prof_train [train_probe args]
OUT=${1:-gpurun_out/prof_train}; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
python $ROOTD/tools/train_probe.py "$@" > "$ROOTD/$OUT/plain.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/$OUT/stats" -- python $ROOTD/tools/train_probe.py "$@" > "$ROOTD/$OUT/stats.log" 2>&1
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:64]
with open(out + '/kernel_stats.txt', 'w') as fo:
    for r in rows[:70]:
        fo.write('%-66s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%\n' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, r['Percentage']))
print(open(out + '/kernel_stats.txt').read())
print(open(out + '/plain.log').read())
PY
