#!/bin/bash
# per-kernel totals of one C2 training step, this build vs SUP3R_AMD_LIB=$1 (same box)
# usage (GPU box): bash tools/dbg/ab_step.sh sup3r_amd/lib/libsup3r_hip_base.so [regex]
R=$(cd "$(dirname "$0")/../.." && pwd); BASE=$1; RX=${2:-.}
C2="--gen gen_5x_12x_2f.json --disc disc_st.json --lr-shape 8,16,16,24,4 --precision bf16 --iters 3"
mkdir -p $R/gpurun_out/ab; cd /tmp; export TMPDIR=/tmp
for v in new base; do
  rm -rf $R/gpurun_out/ab/$v
  if [ $v = base ]; then export SUP3R_AMD_LIB=$R/$BASE; else unset SUP3R_AMD_LIB; fi
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab/$v -- python $R/tools/train_probe.py $C2 > $R/gpurun_out/ab/$v.log 2>&1
  python $R/tools/dbg/step_sequence.py $R/gpurun_out/ab/$v > $R/gpurun_out/ab/$v.seq
done
unset SUP3R_AMD_LIB
python - "$R" "$RX" <<'PY'
import re, collections, sys
R, rx = sys.argv[1], re.compile(sys.argv[2])
def load(f):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for l in open(f).read().splitlines()[1:]:
        m = re.match(r'\s*([\d.]+) us\s+(\d+) x\s+([\d.]+) us\s+(\S.*?) \[', l)
        tot[m.group(4)][0] += int(m.group(2)); tot[m.group(4)][1] += int(m.group(2)) * float(m.group(3))
    return tot
a, b = load(R + '/gpurun_out/ab/base.seq'), load(R + '/gpurun_out/ab/new.seq')
rows = sorted((b.get(k, [0, 0])[1] - a.get(k, [0, 0])[1], k) for k in set(a) | set(b))
for d, k in rows:
    if abs(d) > 15 and rx.search(k):
        print(f'{d:9.1f} us  {k[:60]:60s} {a.get(k,[0,0])[0]:3d} x {a.get(k,[0,1e-9])[1]/max(1,a.get(k,[0,0])[0]):7.1f} -> {b.get(k,[0,0])[0]:3d} x {b.get(k,[0,0])[1]/max(1,b.get(k,[0,0])[0]):7.1f}')
print('busy base %.1f us  new %.1f us' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
PY
find $R/gpurun_out/ab -name "*.csv" -delete
