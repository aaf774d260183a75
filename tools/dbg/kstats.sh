#!/bin/bash
# top kernels of a command under rocprofv3 --stats:  bash tools/dbg/kstats.sh <n_rows> -- <command>
N=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d /tmp/ks_XXXX)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- "$@" > $D/log 2>&1
python3 - "$D" "$N" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.2f ms in %d dispatches' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
for r in rows[:int(sys.argv[2])]:
    print('%-84s calls %6s total %9.2f ms avg %8.1f us' % (re.sub(r'\(anonymous namespace\)::', '', r['Name']).replace('void ', '')[:84], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
rm -rf $D
