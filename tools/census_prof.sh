#!/bin/bash
# per-kernel breakdown (rocprofv3 --kernel-trace --stats) of one census entry
# Usage on the GPU box: bash tools/census_prof.sh gpurun_out/prof_2d "spatial/gen_2x_2f" [census args]
OUT=${1:-gpurun_out/prof_census}; ONLY=${2:-spatial/gen_2x_2f}; shift; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/$OUT/stats" -- python $ROOTD/tools/config_census.py --only "$ONLY" --precisions bf16 --out "$ROOTD/$OUT/census.md" "$@" > "$ROOTD/$OUT/stats.log" 2>&1
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
print(open(out + '/kernel_stats.txt').read())
PY
