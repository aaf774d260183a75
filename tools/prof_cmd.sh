#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command; top kernels to <out>/kernel_stats.txt
# Usage on the GPU box: bash tools/prof_cmd.sh gpurun_out/prof_x -- python tools/dbg/x.py args
OUT=$1; shift; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; CMD=("$@")
for i in "${!CMD[@]}"; do [[ -e "$ROOTD/${CMD[$i]}" ]] && CMD[$i]="$ROOTD/${CMD[$i]}"; done
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/$OUT/stats" -- "${CMD[@]}" > "$ROOTD/$OUT/stats.log" 2>&1
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:72]
with open(out + '/kernel_stats.txt', 'w') as fo:
    for r in rows[:30]:
        fo.write('%-74s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%\n' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, r['Percentage']))
PY
