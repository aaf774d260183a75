#!/bin/bash
# Sample GPU clock / power while a command runs:  tools/sample_clocks.sh out.log -- cmd...
# (rocm-smi polling; used to show the trunk kernel runs power-limited)
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.2; done ) > "$out" &
SAMPLER=$!
"$@"
rc=$?
kill $SAMPLER
exit $rc
