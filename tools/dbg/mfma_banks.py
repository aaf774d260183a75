#!/usr/bin/env python
"""histogram of (A, B, C) start-register residues mod 4 over the v_mfma instructions of one kernel:
python tools/dbg/mfma_banks.py <file.hip> <mangled-name prefix>   (compiles to ISA with hipcc)"""
import collections
import os
import re
import subprocess
import sys
import tempfile
src, key = sys.argv[1], sys.argv[2]
out = tempfile.mktemp(suffix='.s')
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only',
                os.path.abspath(src), '-o', out], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
s = open(out).read()
# the function's text: from its label to its .Lfunc_end marker (an early exit has its own s_endpgm)
m = re.search(r'^(' + re.escape(key) + r'[^\s:]*):', s, re.M)
f = s[m.end():]
f = f[:f.index('.Lfunc_end')]
c = collections.Counter()
n = 0
for x in f.split('\n'):
    if 'v_mfma' in x:
        d, a, b, cc = [int(t) for t in re.findall(r'v\[(\d+):\d+\]', x)[:4]]
        c[(a % 4, b % 4, cc % 4)] += 1
        n += 1
print(n, 'MFMAs;', sorted(c.items()), '; scratch ops:', f.count('scratch_'))
