#!/bin/bash
# bash tools/c3_timeline.sh <outdir> <mode,...>   (GPU box; modes of tools/c3_probe.py)
OUT=${1:-gpurun_out/c3_tl}; MODES=${2:-copy,k16}
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; cd /tmp; export TMPDIR=/tmp
for m in ${MODES//,/ }; do
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$ROOTD/$OUT/$m" -- \
    python $ROOTD/tools/c3_probe.py 6 $m > "$ROOTD/$OUT/$m.json" 2> "$ROOTD/$OUT/$m.err"
  python $ROOTD/tools/c3_timeline.py "$ROOTD/$OUT/$m" $m >> "$ROOTD/$OUT/summary.txt" 2>&1
done
cat "$ROOTD/$OUT/summary.txt"
