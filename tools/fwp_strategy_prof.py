"""host profile of the C3 strategy path (bench.py --mode c3): where the time
of one chunk goes between init_chunk and the delivered hi-res array"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    import torch
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    n, el, _ = bench.c3_leg(8, 4, 2, 1, 0, entry='strategy')
    pr.disable()
    print('chunks', n, 'seconds', el, 'chunks/s', n / el)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
