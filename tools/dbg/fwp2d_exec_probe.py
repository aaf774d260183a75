#!/usr/bin/env python
"""bench.py's fwd2d executor leg on its own (for rocprofv3 timelines and host
profiles): python tools/dbg/fwp2d_exec_probe.py [batch] [reps] [--cprofile]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
batch = int(argv[0]) if argv else 4
reps = int(argv[1]) if len(argv) > 1 else 3
sys.argv = sys.argv[:1]
import bench  # noqa: E402

if '--cprofile' in os.environ.get('PROBE_FLAGS', ''):
    import cProfile
    import pstats
    bench.fwp2d_executor_leg(batch=batch, reps=1)
    pr = cProfile.Profile()
    pr.enable()
    out = bench.fwp2d_executor_leg(batch=batch, reps=reps)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
else:
    out = bench.fwp2d_executor_leg(batch=batch, reps=reps)
print({k: v for k, v in out.items() if k != 'workload'})
