#!/bin/bash
# GPU-side account of the C3 executor: kernel / copy time per chunk vs wall time
# Usage on the GPU box: bash tools/c3_gpu_busy.sh gpurun_out/c3_busy [steps]
OUT=${1:-gpurun_out/c3_busy}; STEPS=${2:-12}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
python $ROOTD/bench.py --mode c3 --steps $STEPS --warmup 2 > "$ROOTD/$OUT/plain.json" 2> "$ROOTD/$OUT/plain.err"
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$ROOTD/$OUT/stats" -- python $ROOTD/bench.py --mode c3 --steps $STEPS --warmup 2 > "$ROOTD/$OUT/prof.json" 2> "$ROOTD/$OUT/prof.err"
python - "$ROOTD/$OUT" <<'PY'
import csv, glob, sys, re, json
out = sys.argv[1]
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:64]
ks = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(ks[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
with open(out + '/summary.txt', 'w') as fo:
    fo.write('kernel time total %.1f ms\n' % tot)
    for r in rows[:14]:
        fo.write('%-66s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%\n' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, r['Percentage']))
    for f in glob.glob(out + '/stats/**/*memory_copy_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            fo.write('copy %s\n' % dict(r))
    for name in ('plain.json', 'prof.json'):
        try:
            j = json.loads(open(out + '/' + name).read().strip().splitlines()[-1])
            fo.write('%s: %s %s, %.2f ms per step\n' % (name, j['value'], j['unit'], j['ms_per_step']))
        except Exception as e:
            fo.write('%s: %r\n' % (name, e))
print(open(out + '/summary.txt').read())
PY
