#!/bin/bash
# average duration of the kernels whose name matches a pattern, under rocprofv3 --stats
# bash tools/dbg/kernel_time.sh '<grep -E pattern>' -- <command>
PAT=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
D=$(mktemp -d /tmp/kt_XXXX)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- "$@" > $D/log 2>&1
python3 - "$D" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if re.search(sys.argv[2], r['Name']):
        print('%-90s calls %5s avg %9.1f us' % (re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
rm -rf $D
